import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
import graphblas_amd as gb
from graphblas_amd import device, synthetic, _lib
gb.init()
scale=24; n=1<<scale
indptr, col = synthetic.rmat_csr(scale, device="cuda")
one = torch.ones(1, dtype=torch.bool, device="cuda")
A = device.matrix_from_device_csr(indptr, col, one, n, n, "BOOL", iso=True)
device.cache_transpose(A)
deg = indptr[1:]-indptr[:-1]; src=int(torch.argmax(deg).item())
def sync(): _lib.lib.GrX_synchronize()
for rep in range(3):
    v = gb.Vector("INT32", n); q = gb.Vector(bool, n); q[src] << True; succ = gb.Scalar(bool); d=0
    sync(); T0=time.perf_counter()
    while True:
        d+=1
        t0=time.perf_counter(); v[:](mask=q.V) << d; sync(); t1=time.perf_counter()
        q(~v.S, replace=True) << q.vxm(A, gb.semiring.lor_land); sync(); t2=time.perf_counter()
        st=device.last_stats()
        succ << q.reduce(gb.monoid.lor, allow_empty=False); t3=time.perf_counter()
        if rep==2: print(f"level {d}: assign {1e3*(t1-t0):.3f} ms  vxm {1e3*(t2-t1):.3f} ms (method {st['method']}, launches {st['kernel_launches']})  reduce {1e3*(t3-t2):.3f} ms")
        if not succ: break
    if rep==2: print("total", 1e3*(time.perf_counter()-T0))
