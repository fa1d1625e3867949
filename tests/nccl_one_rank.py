"""Run by tests/test_gpu_scale.py::test_rccl_one_rank_world in a process of its own (GPU box): a ONE-rank `nccl` (= RCCL) world that
drives every collective of python-graphblas_amd/sharded.py on the library's own device memory -- all_gather_into_tensor into the u
replicas (values and presence words, asynchronous), the probe exchange, the monoid all-reduce of the row-sharded vxm (MIN / SUM /
MAX + bit-or of the presence words) -- and checks every result against the same product computed without the exchange.  With one
rank a collective is a self-copy, but RCCL initialises, registers the buffers and runs its kernels on them: what a multi-GPU run
does first.  Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    seen = torch.ones(1, device="cuda")
    dist.all_reduce(seen)
    assert int(seen.item()) == 1
    import graphblas_amd as gb
    from graphblas_amd import device, sharded, synthetic

    gb.init()
    out = {"backend": dist.get_backend(), "checks": {}}
    scale = 18
    n = 1 << scale
    # ---- the overlapped pull step: 2 row blocks, two replicas of u, all_gather_into_tensor per block, values (+ presence words) ----
    for name, presence in (("min_plus", False), ("lor_land", True)):
        chunks = 2
        ranges = sharded.chunk_blocks(n, 0, 1, chunks)
        graphs = synthetic.rmat_csr(scale, device="cuda", row_ranges=ranges)
        ip_f, col_f = synthetic.rmat_csr(scale, device="cuda")
        gen = torch.Generator(device="cuda")
        gen.manual_seed(5)
        visited = torch.rand(n, generator=gen, device="cuda") < 0.5
        if name == "min_plus":
            d0 = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
            us = [device.vector_from_device(d0) for _ in range(2)]
            sr, accum, desc, tname = gb.semiring.min_plus["FP32"], gb.binary.min["FP32"], "GrB_DESC_SC", "FP32"
        else:
            fr = torch.rand(n, generator=gen, device="cuda") < 0.3
            us = [device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=fr) for _ in range(2)]
            sr, accum, desc, tname = gb.semiring.lor_land["BOOL"], None, "GrB_DESC_RSC", "BOOL"
        As, ws, masks = [], [], []
        for (lo, hi), (ip, col) in zip(ranges, graphs):
            if name == "min_plus":
                vals = synthetic.edge_weights(col_f, scale)[ip_f[lo]: ip_f[hi]].contiguous()
                As.append(device.matrix_from_device_csr(ip, col, vals, hi - lo, n, "FP32"))
                ws.append(device.vector_from_device(d0[lo:hi].contiguous()))
            else:
                As.append(device.matrix_from_device_csr(ip, col, torch.ones(1, dtype=torch.bool, device="cuda"), hi - lo, n, "BOOL", iso=True))
                ws.append(device.vector_from_device(torch.ones(hi - lo, dtype=torch.bool, device="cuda"), present=fr[lo:hi].contiguous()))
            masks.append(device.vector_from_device(torch.ones(hi - lo, dtype=torch.bool, device="cuda"), present=visited[lo:hi].contiguous()))
        ov = sharded.OverlappedMxv(As, ws, masks, us, sr, accum=accum, desc_name=desc, presence=presence)
        ov.probe_exchange()
        torch.cuda.synchronize()
        steps = 3
        for _ in range(steps):
            ov.step()
        torch.cuda.synchronize()
        # the same steps on the whole graph in one piece (no exchange: w becomes the next u by a copy)
        if name == "min_plus":
            A = device.matrix_from_device_csr(ip_f, col_f, synthetic.edge_weights(col_f, scale), n, n, "FP32")
            u = device.vector_from_device(d0)
        else:
            A = device.matrix_from_device_csr(ip_f, col_f, torch.ones(1, dtype=torch.bool, device="cuda"), n, n, "BOOL", iso=True)
            u = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=fr)
        vis = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=visited)
        for _ in range(steps):
            w = u.dup()
            if name == "min_plus":
                w(~vis.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
            else:
                w(~vis.S, replace=True) << A.mxv(u, gb.semiring.lor_land)
            u = w
        gi, gv = ov.current_u().to_coo()
        ei, ev = u.to_coo()
        out["checks"][f"overlapped_{name}"] = bool(np.array_equal(gi, ei) and np.array_equal(gv, ev))
        out["staged_through_torch_buffers"] = bool(ov.staged)
        del ov, As, ws, masks, us, A, u, w, vis
    # ---- the monoid all-reduce of the row-sharded vxm (one rank: the "partial" product is the whole one) ----
    scale = 14
    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cuda")
    rng = np.random.default_rng(3)
    for tname, srn in (("FP32", "min_plus"), ("INT64", "plus_times"), ("BOOL", "lor_land"), ("FP32", "any_pair")):
        np_t = {"FP32": np.float32, "INT64": np.int64, "BOOL": bool}[tname]
        vals = torch.from_numpy((rng.integers(1, 9, col.numel())).astype(np_t) if tname != "BOOL" else rng.random(col.numel()) < 0.8).cuda()
        A = device.matrix_from_device_csr(ip, col, vals, n, n, tname)
        ui = np.flatnonzero(rng.random(n) < 0.3)
        uv = rng.integers(1, 9, ui.size).astype(np_t) if tname != "BOOL" else rng.random(ui.size) < 0.8
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector(tname, size=n)
        sharded.sharded_vxm(gb, w, u, A, getattr(gb.semiring, srn))
        e = u.vxm(A, getattr(gb.semiring, srn)).new()
        gi, gv = w.to_coo()
        ei, ev = e.to_coo()
        same_vals = np.array_equal(gv, ev) if srn != "any_pair" else True
        out["checks"][f"sharded_vxm_{tname}_{srn}"] = bool(np.array_equal(gi, ei) and same_vals)
    out["ok"] = all(out["checks"].values())
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
