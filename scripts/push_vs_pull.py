#!/usr/bin/env python3
"""GPU measurement (SURVEY.md section 8d grids): one BFS level step  q<~visited.S, replace> = q lor.land A  on R-MAT at frontier densities x
visited densities, the pull direction (over the cached transpose) against the push direction (SpMSpV over A's rows; thin path of round 4),
plus the frontier of THREE vertices and the scale-20 SSSP step with a sparse u (density 1e-3).  One JSON line per point.
    python scripts/push_vs_pull.py [scale]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import graphblas_amd as gb
from graphblas_amd import _lib, device, synthetic
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gb.init()
L = _lib.lib


def time_call(fn, reps=5):
    outs = []
    for _ in range(reps):
        torch.cuda.synchronize()
        device.timer_start()
        fn()
        outs.append(device.timer_stop())
    return min(outs)


def bfs_grid(scale):
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, torch.ones(1, dtype=torch.bool, device="cuda"), n, n, "BOOL", iso=True)
    device.cache_transpose(A)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    sr, desc = gb.semiring.lor_land["BOOL"], _lib.handle("GrB_DESC_RSC")
    # (the three random vertices of the first point are, on an R-MAT graph, usually isolated ones -- the point of rounds 1-3, kept for the
    #  comparison: pure per-call overhead; the last point takes three vertices that HAVE edges)
    points = [(3.0 / n, 0.0)] + [(fd, vd) for fd in (1e-4, 1e-2, 0.3) for vd in (0.0, 0.5, 0.9)] + [(-3.0 / n, 0.0)]
    deg = indptr[1:] - indptr[:-1]
    for fd, vd in points:
        visited = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=torch.rand(n, generator=gen, device="cuda") < vd)
        present = torch.rand(n, generator=gen, device="cuda") < fd
        with_edges = fd < 0
        fd = abs(fd)
        if fd * n < 10:
            present = torch.zeros(n, dtype=torch.bool, device="cuda")
            if with_edges:
                cand = torch.nonzero(deg > 0).flatten()
                present[cand[torch.randint(0, cand.numel(), (3,), generator=gen, device="cuda")]] = True
            else:
                present[torch.randint(0, n, (3,), generator=gen, device="cuda")] = True
        q = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=present)
        res = {}
        for mode, name in ((0, "pull"), (2, "push"), (1, "auto")):
            L.GrX_option_set(b"push_mode", mode)
            keep = {}

            def call():
                keep["w"] = w = gb.Vector("BOOL", n)
                assert L.GrB_vxm(w._carg, visited._carg, None, sr._carg, q._carg, A._carg, desc) == 0

            for _ in range(3):
                call()  # (layouts; operands settle in their order)
            res[name + "_ms"] = time_call(call)
            res[name + "_nvals"] = keep["w"].nvals
            st = device.last_stats()
            res[name + "_method"] = st["method"]
            if name == "push":
                res["work"] = st["flops"]
                res["one_workgroup"] = st["long_kernel"] == -2
        assert res["pull_nvals"] == res["push_nvals"] == res["auto_nvals"], res
        print(json.dumps({"workload": "bfs_level_step lor_land", "scale": scale, "frontier_density": fd, "frontier_with_edges": bool(with_edges or fd * n >= 10), "visited_density": vd,
                          "frontier_nvals": int(present.sum().item()), **res}), flush=True)
    L.GrX_option_set(b"push_mode", 1)


def sssp_sparse_u(scale=20, dens=1e-3):
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, synthetic.edge_weights(col, scale), n, n, "FP32")
    device.cache_transpose(A)
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    d = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
    present = torch.rand(n, generator=gen, device="cuda") < dens
    u = device.vector_from_device(d, present=present)
    sr, acc = gb.semiring.min_plus["FP32"], gb.binary.min["FP32"]
    res = {}
    for mode, name in ((0, "pull"), (2, "push"), (1, "auto")):
        L.GrX_option_set(b"push_mode", mode)
        keep = {}

        def call():
            keep["w"] = w = device.vector_from_device(d)
            assert L.GrB_mxv(w._carg, None, acc._carg, sr._carg, A._carg, u._carg, None) == 0

        for _ in range(3):
            call()
        res[name + "_ms"] = time_call(call)
        res[name + "_method"] = device.last_stats()["method"]
        # (compared entry for entry with the pull result: a float sum over a million values depends on the order they are stored in)
        if name == "pull":
            ref = keep["w"]
        else:
            assert keep["w"].isequal(ref), (name, res)
    print(json.dumps({"workload": "sssp_step w(min) << A min.+ u, sparse u", "scale": scale, "u_density": dens, "u_nvals": int(present.sum().item()), **res}), flush=True)
    L.GrX_option_set(b"push_mode", 1)


if scale != 24:
    bfs_grid(scale)
else:
    bfs_grid(22)
    sssp_sparse_u(20)
    bfs_grid(24)
