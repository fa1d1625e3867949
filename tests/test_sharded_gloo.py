"""N > 1 path on CPU: two gloo ranks, each driving the SIMT-emulator build of the kernels on its row
block, all-gathering w into the replicated u -- compared with the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, scale, iters, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GRB_EMU_PREBUILT="1")
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.backend import bind

    gb = bind("emu")
    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    lo, hi = sharded.row_block(n, rank, world)
    indptr, col = synthetic.rmat_csr(scale, device="cpu", row_range=(lo, hi))
    w_all = synthetic.edge_weights(torch.zeros(0, dtype=torch.int32), scale)  # (generator warm-up parity is irrelevant)
    full_ip, full_col = synthetic.rmat_csr(scale, device="cpu")
    wts_full = synthetic.edge_weights(full_col, scale)
    wts = wts_full[full_ip[lo]: full_ip[hi]].contiguous()
    A = device.matrix_from_device_csr(indptr, col, wts, hi - lo, n, "FP32", copy=True)
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32)
    present = torch.rand(n, generator=g) < 0.4
    visited = torch.rand(n, generator=g) < 0.5
    idx = torch.nonzero(present).flatten().numpy()
    u = gb.Vector.from_coo(idx, dist0.numpy()[idx], dtype="FP32", size=n)
    loc = idx[(idx >= lo) & (idx < hi)]
    w = gb.Vector.from_coo(loc - lo, dist0.numpy()[loc], dtype="FP32", size=hi - lo)
    vloc = torch.nonzero(visited[lo:hi]).flatten().numpy()
    vis = gb.Vector.from_coo(vloc, np.ones(vloc.size, bool), dtype="BOOL", size=hi - lo)
    for _ in range(iters):
        w(~vis.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
        sharded.allgather_into(u, w, device="cpu")
    ui, uv = u.to_coo()
    if rank == 0:
        q.put((ui.tolist(), uv.tolist(), full_ip.numpy(), full_col.numpy(), wts_full.numpy(), idx, dist0.numpy(), visited.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_mxv_two_ranks():
    import subprocess

    import torch.multiprocessing as mp

    from oracle import grb_oracle as O

    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    scale, iters, world = 10, 3, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, scale, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    ui, uv, ip, col, wts, idx, dist0, visited = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 1 << scale
    oa = O.OMat(n, n, ip, col.astype(np.int64), wts, "FP32")
    ou = O.OVec(n, idx, dist0[idx], "FP32")
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    for _ in range(iters):
        ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    assert ui == ou.idx.tolist()
    assert uv == ou.vals.tolist()
