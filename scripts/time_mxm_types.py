"""SpGEMM A @ A on R-MAT with real (non-iso) values in several types and semirings: ms per product (GPU box).
usage: python scripts/time_mxm_types.py [scale]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.backend import bind  # noqa: E402

gb = bind("gpu")
from graphblas_amd import device, synthetic  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << scale
indptr, col = synthetic.rmat_csr(scale, device="cuda")
gen = torch.Generator(device="cuda")
gen.manual_seed(5)
for tname, tt, sr in (("INT64", torch.int64, "plus_times"), ("FP64", torch.float64, "plus_times"), ("FP32", torch.float32, "plus_times"),
                      ("FP32", torch.float32, "min_plus"), ("INT32", torch.int32, "plus_times"), ("BOOL", torch.bool, "lor_land"),
                      ("FP64", torch.float64, "max_times"), ("INT64", torch.int64, "any_pair")):
    if tt == torch.bool:
        vals = torch.ones(col.numel(), dtype=torch.bool, device="cuda")
    elif tt.is_floating_point:
        vals = torch.rand(col.numel(), generator=gen, device="cuda", dtype=tt)
    else:
        vals = torch.randint(1, 5, (col.numel(),), generator=gen, device="cuda").to(tt)
    A = device.matrix_from_device_csr(indptr, col, vals, n, n, tname)
    s = getattr(gb.semiring, sr)
    C = A.mxm(A, s).new()
    del C
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        C = A.mxm(A, s).new()
        nv = C.nvals
        del C
    torch.cuda.synchronize()
    print(f"scale {scale} {tname} {sr}: {(time.perf_counter() - t) / 2 * 1e3:.1f} ms  nnz(C) {nv}")
