import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import graphblas_amd as gb
gb.init()
from graphblas_amd import device
rng = np.random.default_rng(0)
m, n = 300, 400
r = rng.integers(0, m, 2000); c = rng.integers(0, n, 2000)
key = np.unique(r * n + c); r, c = key // n, key % n
v = rng.integers(1, 9, r.size).astype(np.float32)
def T(label, f, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); out = f(); device.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(label, [round(x, 2) for x in ts]); return out
A = T("from_coo", lambda: gb.Matrix.from_coo(r, c, v, dtype="FP32", nrows=m, ncols=n))
u = T("vec from_coo", lambda: gb.Vector.from_coo(np.arange(n), np.ones(n, np.float32), dtype="FP32", size=n))
w = T("mxv", lambda: A.mxv(u, gb.semiring.min_plus).new())
T("to_coo", lambda: w.to_coo())
T("vxm", lambda: u.vxm(A.T, gb.semiring.min_plus).new())
T("reduce", lambda: w.reduce(gb.monoid.plus).new() if hasattr(w.reduce(gb.monoid.plus), "new") else None)
