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


# ---------------------------------------------------------------------------------------------------------------------
# the other exchanges of SURVEY.md section 8e: monoid all-reduce of partial products (vxm with the rows sharded), entry-balanced
# row cuts with an all-gather-v, and the row-sharded mxm with flop-balanced cuts
# ---------------------------------------------------------------------------------------------------------------------
def _spawn(worker, world, args, timeout=600):
    import subprocess

    import torch.multiprocessing as mp

    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return dict(out)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GRB_EMU_PREBUILT="1")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.backend import bind

    return bind("emu"), dist


def _vxm_worker(rank, world, port, q, scale, case):
    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    cuts = sharded.balanced_cuts(ip.numpy(), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    tname, srname, accum = case
    wts = synthetic.edge_weights(col, scale, dtype=torch.int64 if tname == "INT64" else None)
    e0, e1 = int(ip[lo]), int(ip[hi])
    vals = torch.ones(1, dtype=torch.bool) if tname == "BOOL" else wts[e0:e1].contiguous()
    A = device.matrix_from_device_csr((ip[lo:hi + 1] - ip[lo]).contiguous(), col[e0:e1].contiguous(), vals, hi - lo, n, tname,
                                      copy=True, iso=(tname == "BOOL"))
    rng = np.random.default_rng(5)
    ui = np.flatnonzero(rng.random(n) < 0.3)
    uv = (np.ones(ui.size, bool) if tname == "BOOL" else rng.integers(1, 50, ui.size)).astype({"BOOL": bool, "INT64": np.int64, "FP32": np.float32}[tname])
    sel = (ui >= lo) & (ui < hi)
    u_loc = gb.Vector.from_coo(ui[sel] - lo, uv[sel], dtype=tname, size=hi - lo)
    wi = np.flatnonzero(rng.random(n) < 0.5)
    wv = (np.ones(wi.size, bool) if tname == "BOOL" else rng.integers(1, 2000, wi.size)).astype(uv.dtype)
    mi = np.flatnonzero(rng.random(n) < 0.5)
    w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
    mk = gb.Vector.from_coo(mi, np.ones(mi.size, bool), dtype="BOOL", size=n)
    sharded.sharded_vxm(gb, w, u_loc, A, getattr(gb.semiring, srname), mask=~mk.S, accum=accum, replace=(accum is None), device="cpu")
    gi, gv = w.to_coo()
    q.put((rank, (gi.tolist(), gv.tolist(), cuts)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [("FP32", "min_plus", "min"), ("INT64", "plus_times", "plus"), ("BOOL", "lor_land", None)])
def test_row_sharded_vxm_monoid_allreduce(case):
    """vxm with the rows of A (and u) sharded over two ranks: each rank's partial product over all columns, combined by the
    monoid all-reduce (min / sum / max on the values, bit-or on the presence words), then the write rule on the replicated
    result -- equal to the single-process oracle on both ranks."""
    import torch

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, world = 9, 2
    tname, srname, accum = case
    res = _spawn(_vxm_worker, world, (scale, case))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    wts = synthetic.edge_weights(col, scale, dtype=torch.int64 if tname == "INT64" else None).numpy()
    np_t = {"BOOL": bool, "INT64": np.int64, "FP32": np.float32}[tname]
    oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), np.ones(col.numel(), bool) if tname == "BOOL" else wts.astype(np_t), tname)
    rng = np.random.default_rng(5)
    ui = np.flatnonzero(rng.random(n) < 0.3)
    uv = (np.ones(ui.size, bool) if tname == "BOOL" else rng.integers(1, 50, ui.size)).astype(np_t)
    wi = np.flatnonzero(rng.random(n) < 0.5)
    wv = (np.ones(wi.size, bool) if tname == "BOOL" else rng.integers(1, 2000, wi.size)).astype(np_t)
    mi = np.flatnonzero(rng.random(n) < 0.5)
    exp = O.vxm(O.OVec(n, ui, uv, tname), oa, srname, w=O.OVec(n, wi, wv, tname), mask=O.OVec(n, mi, np.ones(mi.size, bool), "BOOL"),
                mask_comp=True, mask_struct=True, accum=accum, replace=(accum is None))
    cuts = res[0][2]
    assert cuts[0] == 0 and cuts[-1] == n and all(c % 64 == 0 for c in cuts)
    for r in range(world):
        assert res[r][0] == exp.idx.tolist() and res[r][1] == exp.vals.tolist()


def _mxv_cuts_worker(rank, world, port, q, scale, iters):
    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    wts = synthetic.edge_weights(col, scale)
    cuts = sharded.balanced_cuts(ip.numpy(), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    e0, e1 = int(ip[lo]), int(ip[hi])
    A = device.matrix_from_device_csr((ip[lo:hi + 1] - ip[lo]).contiguous(), col[e0:e1].contiguous(), wts[e0:e1].contiguous(), hi - lo, n,
                                      "FP32", copy=True)
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    present = (torch.rand(n, generator=g) < 0.4).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    idx = np.flatnonzero(present)
    # two replicas of u: the exchange of step k lands in the one step k does not read (what lets it overlap with compute)
    u = [gb.Vector.from_coo(idx, dist0[idx], dtype="FP32", size=n) for _ in range(2)]
    loc = idx[(idx >= lo) & (idx < hi)]
    w = gb.Vector.from_coo(loc - lo, dist0[loc], dtype="FP32", size=hi - lo)
    vloc = np.flatnonzero(visited[lo:hi])
    vis = gb.Vector.from_coo(vloc, np.ones(vloc.size, bool), dtype="BOOL", size=hi - lo)
    for k in range(iters):
        w(~vis.S, accum=gb.binary.min) << A.mxv(u[k & 1], gb.semiring.min_plus)
        works = sharded.allgatherv_into(u[(k + 1) & 1], w, cuts, device="cpu", async_op=True)
        for wk in works:
            wk.wait()
    ui, uv = u[iters & 1].to_coo()
    q.put((rank, (ui.tolist(), uv.tolist(), cuts, int(e1 - e0))))
    dist.barrier()
    dist.destroy_process_group()


def test_entry_balanced_cuts_and_allgatherv():
    """Three ranks, row cuts balanced by entries (not rows), the w slices exchanged by an all-gather-v into the other replica
    of u: three relaxation steps equal the single-process oracle; the blocks' entry counts are within 25 % of each other while
    their row counts are not equal."""
    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, iters, world = 10, 3, 3
    res = _spawn(_mxv_cuts_worker, world, (scale, iters))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    wts = synthetic.edge_weights(col, scale).numpy()
    import torch

    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    present = (torch.rand(n, generator=g) < 0.4).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    idx = np.flatnonzero(present)
    oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), wts, "FP32")
    ou = O.OVec(n, idx, dist0[idx], "FP32")
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    for _ in range(iters):
        ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    for r in range(world):
        assert res[r][0] == ou.idx.tolist() and res[r][1] == ou.vals.tolist()
    cuts = res[0][2]
    nnz = [res[r][3] for r in range(world)]
    rows = [cuts[r + 1] - cuts[r] for r in range(world)]
    assert max(nnz) <= 1.25 * min(nnz), (nnz, rows)
    assert len(set(rows)) > 1


def _mxv_delta_worker(rank, world, port, q, scale, iters):
    """The relaxation loop of _mxv_cuts_worker with ONE replica of u and the sparse exchange: after its product every rank sends
    only what changed in its slice."""
    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    wts = synthetic.edge_weights(col, scale)
    cuts = sharded.balanced_cuts(ip.numpy(), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    e0, e1 = int(ip[lo]), int(ip[hi])
    A = device.matrix_from_device_csr((ip[lo:hi + 1] - ip[lo]).contiguous(), col[e0:e1].contiguous(), wts[e0:e1].contiguous(), hi - lo, n,
                                      "FP32", copy=True)
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    present = (torch.rand(n, generator=g) < 0.4).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    idx = np.flatnonzero(present)
    u = gb.Vector.from_coo(idx, dist0[idx], dtype="FP32", size=n)
    loc = idx[(idx >= lo) & (idx < hi)]
    w = gb.Vector.from_coo(loc - lo, dist0[loc], dtype="FP32", size=hi - lo)
    vloc = np.flatnonzero(visited[lo:hi])
    vis = gb.Vector.from_coo(vloc, np.ones(vloc.size, bool), dtype="BOOL", size=hi - lo)
    sent = []
    for k in range(iters):
        w(~vis.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
        # (step 0 changes about a quarter of the entries: with the default limit it takes the dense exchange; later steps are sparse)
        sent.append(sharded.allgather_delta_into(u, w, cuts, device="cpu", dense_above=0.125 if k == 0 else 1.0))
    ui, uv = u.to_coo()
    q.put((rank, (ui.tolist(), uv.tolist(), sent)))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_delta_exchange():
    """Three ranks, entry-balanced cuts, the changed entries exchanged as (index, value) pairs and changed presence words
    (sharded.allgather_delta_into): five relaxation steps equal the single-process oracle on every rank; the first step takes
    the dense fallback, the later ones send fewer and fewer pairs."""
    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, iters, world = 10, 5, 3
    res = _spawn(_mxv_delta_worker, world, (scale, iters))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    wts = synthetic.edge_weights(col, scale).numpy()
    import torch

    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    present = (torch.rand(n, generator=g) < 0.4).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    idx = np.flatnonzero(present)
    oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), wts, "FP32")
    ou = O.OVec(n, idx, dist0[idx], "FP32")
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    for _ in range(iters):
        ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    for r in range(world):
        assert res[r][0] == ou.idx.tolist() and res[r][1] == ou.vals.tolist()
        assert res[r][2] == res[0][2]  # (every rank saw the same counts)
    sent = res[0][2]
    assert sent[0] == -1, sent  # the dense fallback
    assert all(x >= 0 for x in sent[1:]) and sent[-1] <= sent[1], sent


def _mxm_worker(rank, world, port, q, scale):
    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    one = torch.ones(1, dtype=torch.int64)
    B = device.matrix_from_device_csr(ip, col, one, n, n, "INT64", copy=True, iso=True)
    fp = sharded.flops_prefix(ip, col, ip[1:] - ip[:-1])
    cuts = sharded.balanced_cuts(fp, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    e0, e1 = int(ip[lo]), int(ip[hi])
    A = device.matrix_from_device_csr((ip[lo:hi + 1] - ip[lo]).contiguous(), col[e0:e1].contiguous(), one, hi - lo, n, "INT64", copy=True,
                                      iso=True)
    C = A.mxm(B, gb.semiring.plus_times).new()
    flops = device.last_stats()["flops"]
    M = gb.Matrix("INT64", hi - lo, n)
    M(A.S) << A.mxm(B, gb.semiring.plus_times)
    cp, cj, cx = C.to_csr()
    mp_, mj, mx = M.to_csr()
    q.put((rank, (cuts, int(flops), cp.tolist(), cj.tolist(), cx.tolist(), mp_.tolist(), mj.tolist(), mx.tolist())))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_mxm_flop_balanced():
    """C = A (+.x) A and C<A.S> = A (+.x) A with the rows of the left operand sharded over two ranks by flop-balanced cuts and B
    replicated: the stacked row blocks equal scipy's product; the ranks' flop counts are within 20 % of each other."""
    import scipy.sparse as sp

    from graphblas_amd import synthetic

    scale, world = 8, 2
    res = _spawn(_mxm_worker, world, (scale,))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    S = sp.csr_matrix((np.ones(col.numel(), np.int64), col.numpy(), ip.numpy()), shape=(n, n))
    ref = (S @ S).tocsr()
    ref.sort_indices()
    refm = ref.multiply(S.astype(bool)).tocsr()
    refm.sort_indices()
    cuts = res[0][0]
    flops = [res[r][1] for r in range(world)]
    assert max(flops) <= 1.2 * min(flops), flops
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        blk = ref[lo:hi]
        assert res[r][2] == blk.indptr.tolist() and res[r][3] == blk.indices.tolist() and res[r][4] == blk.data.tolist()
        blkm = refm[lo:hi]
        assert res[r][5] == blkm.indptr.tolist() and res[r][6] == blkm.indices.tolist() and res[r][7] == blkm.data.tolist()


def _mxm_streamed_worker(rank, world, port, q, scale, budget):
    """What bench.py's sharded scale-22 line does per rank: GrX_mxm_streamed over the rank's flop-balanced row block in row batches
    under a byte budget (small here: several batches), then the ranks' entry counts, checksums and multiply counts are summed."""
    import ctypes

    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import _lib, device, sharded, synthetic

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    one = torch.ones(1, dtype=torch.int64)
    B = device.matrix_from_device_csr(ip, col, one, n, n, "INT64", copy=True, iso=True)
    cuts = sharded.balanced_cuts(sharded.flops_prefix(ip, col, ip[1:] - ip[:-1]), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    e0, e1 = int(ip[lo]), int(ip[hi])
    A = device.matrix_from_device_csr((ip[lo:hi + 1] - ip[lo]).contiguous(), col[e0:e1].contiguous(), one, hi - lo, n, "INT64", copy=True, iso=True)
    nv, cs, fl, nb = (ctypes.c_uint64(0) for _ in range(4))
    rc = _lib.lib.GrX_mxm_streamed(gb.semiring.plus_times["INT64"]._carg, A._carg, B._carg, ctypes.c_uint64(budget), ctypes.byref(nv), ctypes.byref(cs),
                                   ctypes.byref(fl), ctypes.byref(nb))
    assert rc == 0
    t = torch.tensor([nv.value, cs.value, fl.value], dtype=torch.int64)
    dist.all_reduce(t)  # (the only exchange of the sharded product: three counters)
    q.put((rank, (t.tolist(), int(nb.value), int(nv.value))))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_mxm_streamed_counts():
    """The sharded form of the scale-22 SpGEMM line (the product streamed in row batches per rank, only counters leave): summed over
    the ranks, the entry count is nnz(A A), the checksum and the multiply count are the number of multiplies -- against scipy; every
    rank really ran several batches."""
    import scipy.sparse as sp

    from graphblas_amd import synthetic

    scale, world = 8, 2
    res = _spawn(_mxm_streamed_worker, world, (scale, 1 << 16))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    S = sp.csr_matrix((np.ones(col.numel(), np.int64), col.numpy(), ip.numpy()), shape=(n, n))
    ref = (S @ S).tocsr()
    flops = int(np.diff(ip.numpy())[col.numpy()].sum())
    for r in range(world):
        assert res[r][0] == [ref.nnz, flops, flops], (res[r], ref.nnz, flops)
        assert res[r][1] >= 2  # (batches)
    assert sum(res[r][2] for r in range(world)) == ref.nnz


# ---------------------------------------------------------------------------------------------------------------------
# the overlapped step bench.py --gpus N runs (sharded.OverlappedMxv): every rank owns `chunks` row blocks, two replicas of u,
# the all-gather of block c is issued asynchronously while block c + 1 is computed
# ---------------------------------------------------------------------------------------------------------------------
def _overlap_worker(rank, world, port, q, scale, iters, chunks, kind, exchange="dense"):
    gb, dist = _init(rank, world, port)
    import torch

    from graphblas_amd import device, sharded, synthetic

    n = 1 << scale
    ranges = sharded.chunk_blocks(n, rank, world, chunks)
    graphs = synthetic.rmat_csr(scale, device="cpu", row_ranges=ranges)
    full_ip, full_col = synthetic.rmat_csr(scale, device="cpu")
    wts_full = synthetic.edge_weights(full_col, scale)
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    frontier = (torch.rand(n, generator=g) < 0.3).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    As, ws, masks = [], [], []
    for (lo, hi), (ip, col) in zip(ranges, graphs):
        assert torch.equal(col, full_col[full_ip[lo]: full_ip[hi]])
        vloc = np.flatnonzero(visited[lo:hi])
        masks.append(gb.Vector.from_coo(vloc, np.ones(vloc.size, bool), dtype="BOOL", size=hi - lo))
        if kind == "min_plus":
            As.append(device.matrix_from_device_csr(ip, col, wts_full[full_ip[lo]: full_ip[hi]].contiguous(), hi - lo, n, "FP32", copy=True))
            ws.append(gb.Vector.from_coo(np.arange(hi - lo), dist0[lo:hi], dtype="FP32", size=hi - lo))
        else:
            As.append(device.matrix_from_device_csr(ip, col, torch.ones(1, dtype=torch.bool), hi - lo, n, "BOOL", copy=True, iso=True))
            floc = np.flatnonzero(frontier[lo:hi])
            ws.append(gb.Vector.from_coo(floc, np.ones(floc.size, bool), dtype="BOOL", size=hi - lo))
    if kind == "min_plus":
        us = [gb.Vector.from_coo(np.arange(n), dist0, dtype="FP32", size=n) for _ in range(2)]
        ov = sharded.OverlappedMxv(As, ws, masks, us, gb.semiring.min_plus["FP32"], accum=gb.binary.min["FP32"], desc_name="GrB_DESC_SC",
                                   presence=False, device="cpu", exchange=exchange, delta_below=0.2)
    else:
        fi = np.flatnonzero(frontier)
        us = [gb.Vector.from_coo(fi, np.ones(fi.size, bool), dtype="BOOL", size=n) for _ in range(2)]
        ov = sharded.OverlappedMxv(As, ws, masks, us, gb.semiring.lor_land["BOOL"], desc_name="GrB_DESC_RSC", presence=True, device="cpu",
                                   exchange=exchange)
    ov.probe_exchange()
    for _ in range(iters):
        ov.step()
    ui, uv = ov.current_u().to_coo()
    q.put((rank, (ui.tolist(), uv.tolist(), ranges, ov.staged, ov.log)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,chunks,world,exchange", [("min_plus", 2, 2, "auto"), ("lor_land", 2, 2, "delta"), ("min_plus", 1, 2, "delta")])
def test_sharded_step_exchanges_what_changed(kind, chunks, world, exchange):
    """Round 6: the N-rank step sends what CHANGED when little does (sharded.OverlappedMxv(exchange="auto" | "delta")): (global row, value)
    pairs of the entries that differ from the target replica, the changed presence bytes of a BFS step.  Six steps of the masked
    relaxation -- it converges after a few, so an "auto" run starts dense and ends on deltas -- and of the BFS level step with the delta
    exchange forced equal the single-process oracle on every rank; the log says which method every step took, identically on every rank."""
    import torch

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, iters = 10, 6
    res = _spawn(_overlap_worker, world, (scale, iters, chunks, kind, exchange))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    frontier = (torch.rand(n, generator=g) < 0.3).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    if kind == "min_plus":
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), synthetic.edge_weights(col, scale).numpy(), "FP32")
        ou = O.OVec(n, np.arange(n), dist0, "FP32")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    else:
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), np.ones(col.numel(), bool), "BOOL")
        fi = np.flatnonzero(frontier)
        ou = O.OVec(n, fi, np.ones(fi.size, bool), "BOOL")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "lor_land", w=ou, mask=ovis, mask_comp=True, mask_struct=True, replace=True)
    for r in range(world):
        assert res[r][0] == ou.idx.tolist() and res[r][1] == ou.vals.tolist(), (kind, r)
        methods = [m for m, _b, _c in res[r][4]]
        assert methods == [m for m, _b, _c in res[0][4]]  # (the same choice on every rank)
        assert methods[:2] == ["dense", "dense"] and "delta" in methods, methods
        if exchange == "delta":
            assert methods[2:] == ["delta"] * (iters - 2)
        # a delta step sends less than the dense one
        dense_b = max(b for m, b, _c in res[r][4] if m == "dense")
        assert all(b < dense_b for m, b, _c in res[r][4][-2:] if m == "delta") or kind == "lor_land"


@pytest.mark.parametrize("kind,chunks,world", [("min_plus", 2, 2), ("lor_land", 2, 2), ("min_plus", 1, 2), ("lor_land", 4, 2)])
def test_overlapped_sharded_step(kind, chunks, world):
    """The step of `bench.py --gpus N`: `chunks` row blocks per rank, asynchronous all-gather of every block's slices into the
    other replica of u while the next block is computed.  Three steps of the masked min_plus relaxation (values travel) and of the
    BFS level step (values and presence words travel) equal the single-process oracle on every rank."""
    import torch

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, iters = 10, 3
    res = _spawn(_overlap_worker, world, (scale, iters, chunks, kind))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu")
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    frontier = (torch.rand(n, generator=g) < 0.3).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    if kind == "min_plus":
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), synthetic.edge_weights(col, scale).numpy(), "FP32")
        ou = O.OVec(n, np.arange(n), dist0, "FP32")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    else:
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), np.ones(col.numel(), bool), "BOOL")
        fi = np.flatnonzero(frontier)
        ou = O.OVec(n, fi, np.ones(fi.size, bool), "BOOL")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "lor_land", w=ou, mask=ovis, mask_comp=True, mask_struct=True, replace=True)
        assert ou.idx.size > 0
    blocks = sorted(b for r in range(world) for b in res[r][2])
    assert blocks[0][0] == 0 and blocks[-1][1] == n and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))  # the blocks tile the rows
    for r in range(world):
        assert res[r][0] == ou.idx.tolist() and res[r][1] == ou.vals.tolist()
        assert res[r][3] is False


# ---------------------------------------------------------------------------------------------------------------------
# the block-cyclic step of `bench.py --gpus N --ranked` (round 5): labels ranked by popularity up front, rows dealt in stripes of B rows,
# every block matrix carrying GrX_Matrix_hint_ranked (ordered layouts, no permutation), the gathered slices dealt back into u
# ---------------------------------------------------------------------------------------------------------------------
_CYCLIC_OPTS = ((b"order_min_nnz", 1), (b"lean_min_nnz", 1), (b"split_min_nnz", 1), (b"split_min_len", 8), (b"hot_min_cols", 8), (b"lazy_layout", 0),
                (b"vec_pad_min_bytes", 0), (b"rows_head_min_groups", 1), (b"rtile_entries", 700), (b"hub_min_len", 100))


def _cyclic_worker(rank, world, port, q, scale, iters, chunks, kind, B):
    gb, dist = _init(rank, world, port)
    import scipy.sparse as sp
    import torch

    from graphblas_amd import _lib, device, sharded, synthetic

    for name, val in _CYCLIC_OPTS:
        assert _lib.lib.GrX_option_set(name, val) == 0
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu", relabel="popularity", stripes=(B, world, rank))
    gid = synthetic.stripe_rows(n, B, world, rank, device="cpu").numpy()
    full_ip, full_col = synthetic.rmat_csr(scale, device="cpu", relabel="popularity")
    wts_full = synthetic.edge_weights(full_col, scale).numpy()
    M = sp.csr_matrix((wts_full, full_col.numpy(), full_ip.numpy()), shape=(n, n))[gid]
    assert np.array_equal(M.indptr, ip.numpy()) and np.array_equal(M.indices, col.numpy())  # the generator's stripes = those rows of the whole graph
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    frontier = (torch.rand(n, generator=g) < 0.3).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    h = n // (world * chunks)
    As, ws, masks, ordered = [], [], [], []
    for c in range(chunks):
        sel = sharded.cyclic_chunk_rows(n // world, B, chunks, c, device="cpu").numpy()  # (local stripes c, c + chunks, ...)
        blk = M[sel]
        rows_g = gid[sel]
        vloc = np.flatnonzero(visited[rows_g])
        masks.append(gb.Vector.from_coo(vloc, np.ones(vloc.size, bool), dtype="BOOL", size=h))
        bip, bcol = torch.from_numpy(blk.indptr.astype(np.int64)), torch.from_numpy(blk.indices.astype(np.int32))
        if kind == "min_plus":
            As.append(device.matrix_from_device_csr(bip, bcol, torch.from_numpy(blk.data.astype(np.float32)), h, n, "FP32", copy=True))
            ws.append(gb.Vector.from_coo(np.arange(h), dist0[rows_g], dtype="FP32", size=h))
        else:
            As.append(device.matrix_from_device_csr(bip, bcol, torch.ones(1, dtype=torch.bool), h, n, "BOOL", copy=True, iso=True))
            floc = np.flatnonzero(frontier[rows_g])
            ws.append(gb.Vector.from_coo(floc, np.ones(floc.size, bool), dtype="BOOL", size=h))
        device.matrix_hint_ranked(As[-1])
    if kind == "min_plus":
        us = [gb.Vector.from_coo(np.arange(n), dist0, dtype="FP32", size=n) for _ in range(2)]
        ov = sharded.OverlappedMxv(As, ws, masks, us, gb.semiring.min_plus["FP32"], accum=gb.binary.min["FP32"], desc_name="GrB_DESC_SC",
                                   presence=False, device="cpu", cyclic=B)
    else:
        fi = np.flatnonzero(frontier)
        us = [gb.Vector.from_coo(fi, np.ones(fi.size, bool), dtype="BOOL", size=n) for _ in range(2)]
        ov = sharded.OverlappedMxv(As, ws, masks, us, gb.semiring.lor_land["BOOL"], desc_name="GrB_DESC_RSC", presence=True, device="cpu", cyclic=B)
    ov.probe_exchange()
    for _ in range(iters):
        ov.step()
    ui, uv = ov.current_u().to_coo()
    # which layouts the blocks ran on: one more product of every block into a scratch output (every chunk carries the same mix of rows)
    for c in range(chunks):
        t = gb.Vector(ws[c].dtype, size=h)
        t << As[c].mxv(ov.current_u(), gb.semiring.min_plus if kind == "min_plus" else gb.semiring.lor_land)
        st = device.last_stats()
        ordered.append((int(st["ordered"]), int(st["reorders"])))
    q.put((rank, (ui.tolist(), uv.tolist(), ordered, ov.staged)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,chunks,world,B", [("min_plus", 2, 2, 8), ("lor_land", 2, 2, 8), ("min_plus", 1, 2, 64), ("lor_land", 1, 2, 16)])
def test_block_cyclic_ranked_step(kind, chunks, world, B):
    """The step of `bench.py --gpus N --ranked`: the graph relabelled by popularity, every rank holding the stripes r, r + world, ... of B rows
    (in `chunks` blocks of local rows), the matrices hinted as ranked (ordered layouts in the caller's labels, nothing permuted), the slices
    gathered rank-major and dealt back stripe by stripe.  Three steps equal the single-process oracle on the whole relabelled graph."""
    import torch

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    scale, iters = 10, 3
    res = _spawn(_cyclic_worker, world, (scale, iters, chunks, kind, B))
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cpu", relabel="popularity")
    indeg = np.bincount(col.numpy(), minlength=n)
    # the labels are popularity ranks of the edge stream (duplicates counted): near-monotone column counts after the dedupe
    assert indeg[: n // 8].sum() > 0.6 * indeg.sum() and indeg[: n // 64].min() >= indeg[n // 2:].max()
    g = torch.Generator().manual_seed(11)
    dist0 = torch.randint(0, 1000, (n,), generator=g).to(torch.float32).numpy()
    frontier = (torch.rand(n, generator=g) < 0.3).numpy()
    visited = (torch.rand(n, generator=g) < 0.5).numpy()
    ovis = O.OVec(n, np.flatnonzero(visited), np.ones(int(visited.sum()), bool), "BOOL")
    if kind == "min_plus":
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), synthetic.edge_weights(col, scale).numpy(), "FP32")
        ou = O.OVec(n, np.arange(n), dist0, "FP32")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "min_plus", w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum="min")
    else:
        oa = O.OMat(n, n, ip.numpy(), col.numpy().astype(np.int64), np.ones(col.numel(), bool), "BOOL")
        fi = np.flatnonzero(frontier)
        ou = O.OVec(n, fi, np.ones(fi.size, bool), "BOOL")
        for _ in range(iters):
            ou = O.mxv(oa, ou, "lor_land", w=ou, mask=ovis, mask_comp=True, mask_struct=True, replace=True)
        assert ou.idx.size > 0
    for r in range(world):
        assert res[r][0] == ou.idx.tolist() and res[r][1] == ou.vals.tolist()
        assert all(o == (1, 0) for o in res[r][2]), res[r][2]  # ordered layouts on every chunk, no vector converted
        assert res[r][3] is False


def _any_worker(rank, world, port, q):
    gb, dist = _init(rank, world, port)
    from graphblas_amd import sharded

    n = 192
    # rank 0 has entries at 0..63 (negative), rank 1 at 32..127 (negative): 32..63 on both, 128.. on none
    lo, hi = (0, 64) if rank == 0 else (32, 128)
    idx = np.arange(lo, hi)
    vals = (-(idx + 1) * (rank + 2)).astype(np.int64)
    t = gb.Vector.from_coo(idx, vals, dtype="INT64", size=n)
    sharded.allreduce_monoid(t, "any", 0, device="cpu")
    gi, gv = t.to_coo()
    tf = gb.Vector.from_coo(idx, vals.astype(np.float32), dtype="FP32", size=n)
    sharded.allreduce_monoid(tf, "any", 0, device="cpu")
    fi, fv = tf.to_coo()
    q.put((rank, (gi.tolist(), gv.tolist(), fi.tolist(), fv.tolist())))
    dist.barrier()
    dist.destroy_process_group()


def test_any_monoid_allreduce_with_negative_values():
    """ANY has no identity: a rank that lacks an entry must not contribute a value.  With negative products the result of the
    all-reduce has to be one of the ranks' values (it used to be 0 = the filler of the absent side)."""
    res = _spawn(_any_worker, 2, ())
    for r in range(2):
        gi, gv, fi, fv = res[r]
        assert gi == list(range(128)) == fi
        for i, v, f in zip(gi, gv, fv):
            cands = set()
            if i < 64:
                cands.add(-(i + 1) * 2)
            if i >= 32:
                cands.add(-(i + 1) * 3)
            assert v in cands and int(f) in cands and v < 0
