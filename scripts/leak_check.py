"""Device-memory growth over repeated products (GPU box): free memory before / after loops of mxv, mxm (plain, masked, streamed)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from tests.backend import bind  # noqa: E402

gb = bind("gpu")
from graphblas_amd import _lib, device, synthetic  # noqa: E402


def free_mb():
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    return f / 2**20


scale = 16
n = 1 << scale
ip, col = synthetic.rmat_csr(scale, device="cuda")
one = torch.ones(1, dtype=torch.int64, device="cuda")
A = device.matrix_from_device_csr(ip, col, one, n, n, "INT64", iso=True)
F = device.matrix_from_device_csr(ip, col, synthetic.edge_weights(col, scale), n, n, "FP32")
u = device.vector_from_device(torch.rand(n, device="cuda"))
for name, fn, reps in (
    ("mxv", lambda: F.mxv(u, gb.semiring.min_plus).new(), 300),
    ("mxm", lambda: A.mxm(A, gb.semiring.plus_times).new(), 30),
    ("mxm masked", lambda: A.mxm(A, gb.semiring.plus_times).new(mask=A.S), 30),
):
    fn()
    before = free_mb()
    for _ in range(reps):
        r = fn()
        del r
    after = free_mb()
    print(f"{name}: free before {before:.0f} MiB, after {reps} calls {after:.0f} MiB, delta {before - after:.1f} MiB")
L = _lib.lib
nv, cs, fl, nb = (ctypes.c_uint64() for _ in range(4))
before = free_mb()
for it in range(15):
    if it % 5 == 0 and it:
        print(f"mxm streamed: after {it} runs delta {before - free_mb():.1f} MiB (freed blocks are kept in the library's size-class cache)")
    rc = L.GrX_mxm_streamed(gb.semiring.plus_times["INT64"]._carg, A._carg, A._carg, ctypes.c_uint64(1 << 30), ctypes.byref(nv), ctypes.byref(cs),
                            ctypes.byref(fl), ctypes.byref(nb))
    assert rc == 0
print(f"mxm streamed ({nb.value} batches): delta {before - free_mb():.1f} MiB")

# round 6: the cached layouts of a large matrix (popularity-ordered twin, strips with 16-bit slots, packed row tiles, cold tiles; a row block with
# the library's column ranking) are released with the matrix: free memory after five build / drop cycles = free memory after the first
import gc  # noqa: E402


def cycle(block):
    s = 22
    nn = 1 << s
    if block:
        ipb, colb = synthetic.rmat_csr(s, device="cuda", row_range=(0, nn // 4))
        M = device.matrix_from_device_csr(ipb, colb, synthetic.edge_weights(colb, s), nn // 4, nn, "FP32")
        device.matrix_shard_setup(M, torch.bincount(colb.long(), minlength=nn).to(torch.int32))
        w = device.vector_from_device(torch.rand(nn // 4, device="cuda"))
    else:
        ipb, colb = synthetic.rmat_csr(s, device="cuda")
        M = device.matrix_from_device_csr(ipb, colb, synthetic.edge_weights(colb, s), nn, nn, "FP32")
        w = device.vector_from_device(torch.rand(nn, device="cuda"))
    x = device.vector_from_device(torch.rand(nn, device="cuda"))
    for _ in range(3):
        w(accum=gb.binary.min) << M.mxv(x, gb.semiring.min_plus)
    st = device.last_stats()
    del M, w, x, ipb, colb
    gc.collect()
    device.trim_memory()
    torch.cuda.empty_cache()
    return st["ordered"]


for block in (False, True):
    ordered = cycle(block)
    base = free_mb()
    for _ in range(4):
        cycle(block)
    print(f"layouts of a scale-22 {'row block (shard set-up)' if block else 'matrix'}: ordered {ordered}, free after cycle 1 {base:.0f} MiB, after cycle 5 {free_mb():.0f} MiB, delta {base - free_mb():.1f} MiB")
