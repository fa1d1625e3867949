#!/usr/bin/env python3
"""GPU measurement: one BFS-style level step  q<~visited.S, replace> = q lor.land A  at several frontier
densities, pull (merge-path SpMV over the cached transpose) vs push (SpMSpV over A's rows)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import graphblas_amd as gb
from graphblas_amd import _lib, device, synthetic
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
gb.init()
n = 1 << scale
indptr, col = synthetic.rmat_csr(scale, device="cuda")
one = torch.ones(1, dtype=torch.bool, device="cuda")
A = device.matrix_from_device_csr(indptr, col, one, n, n, "BOOL", iso=True)
device.cache_transpose(A)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
visited = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=torch.rand(n, generator=gen, device="cuda") < 0.3)
L = _lib.lib
sr = gb.semiring.lor_land["BOOL"]
desc = _lib.handle("GrB_DESC_RSC")
for dens in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
    present = torch.rand(n, generator=gen, device="cuda") < dens
    q = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=present)
    res = {}
    for mode, name in ((0, "pull"), (2, "push")):
        L.GrX_option_set(b"push_mode", mode)
        outs = []
        for rep in range(4):
            w = gb.Vector("BOOL", n)
            torch.cuda.synchronize()
            device.timer_start()
            rc = L.GrB_vxm(w._carg, visited._carg, None, sr._carg, q._carg, A._carg, desc)
            ms = device.timer_stop()
            assert rc == 0
            outs.append(ms)
        res[name] = min(outs)
        res[name + "_nvals"] = w.nvals
        res[name + "_method"] = device.last_stats()["method"]
    assert res["pull_nvals"] == res["push_nvals"]
    print(json.dumps({"scale": scale, "frontier_density": dens, "frontier_nvals": int(present.sum().item()), **res}), flush=True)
L.GrX_option_set(b"push_mode", 1)
