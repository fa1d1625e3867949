"""Times the SURVEY 8f-4 callers on the GPU: reduce_rowwise / reduce_columnwise with agg.count (the plus_pair mat-vec with a
dense iso operand) on the scale-24 graph, and A.power(n) (repeated squaring over the SpGEMM path) on a scale-14 graph.
Prints one JSON line per measurement."""
import json
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

import graphblas_amd as gb
from graphblas_amd import _lib, device, synthetic

gb.init()


def sync():
    _lib.lib.GrX_synchronize()


def timed(fn, reps):
    fn()
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << scale
indptr, col = synthetic.rmat_csr(scale, device="cuda")
w8 = torch.randint(1, 200, (col.numel(),), dtype=torch.float32, device="cuda")
A = device.matrix_from_device_csr(indptr, col, w8, n, n, "FP32")
device.cache_transpose(A)
nnz = int(col.numel())
w = gb.Vector("INT64", n)
for name, fn in [("reduce_rowwise(agg.count)", lambda: w << A.reduce_rowwise(gb.agg.count)),
                 ("reduce_columnwise(agg.count)", lambda: w << A.reduce_columnwise(gb.agg.count)),
                 ("reduce_rowwise(monoid.plus)", None)]:
    if fn is None:
        wf = gb.Vector("FP32", n)
        fn = lambda: wf << A.reduce_rowwise(gb.monoid.plus)  # noqa: E731
    t = timed(fn, 20)
    print(json.dumps({"what": name, "scale": scale, "nnz": nnz, "ms": 1e3 * t, "G_entries_per_s": nnz / t / 1e9}))
del A, w8, indptr, col

ps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = 1 << ps
indptr, col = synthetic.rmat_csr(ps, device="cuda")
one = torch.ones(1, dtype=torch.int64, device="cuda")
B = device.matrix_from_device_csr(indptr, col, one, n, n, "INT64", iso=True)
for p in (2, 3, 4):
    C = gb.Matrix("INT64", n, n)
    t = timed(lambda: C << B.power(p), 3)
    print(json.dumps({"what": f"power({p}) plus_times INT64", "scale": ps, "nnz_A": int(col.numel()), "nnz_C": int(C.nvals), "ms": 1e3 * t}))
