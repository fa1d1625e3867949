#!/usr/bin/env python3
"""GPU diagnostics: time one rank's share of the N-way row-sharded bench graph on a single GPU (the compute part of a
`bench.py --gpus N` step, without the all-gather): python scripts/block_time.py [scale] [world]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import graphblas_amd as gb
from graphblas_amd import device, synthetic

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gb.init()
n = 1 << scale
for world in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 2, 4, 8]):
    rows = n // world
    indptr, col = synthetic.rmat_csr(scale, device="cuda", row_range=(0, rows) if world > 1 else None)
    vals = synthetic.edge_weights(col, scale)
    A = device.matrix_from_device_csr(indptr, col, vals, rows, n, "FP32")
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    visited = (torch.rand(n, generator=gen, device="cuda") < 0.5)[:rows].contiguous()
    active = int((indptr[1:] - indptr[:-1])[~visited].sum().item())
    dist = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
    u = device.vector_from_device(dist)
    w = device.vector_from_device(dist[:rows].contiguous())
    mask = device.vector_from_device(torch.ones(rows, dtype=torch.bool, device="cuda"), present=visited)
    for _ in range(3):
        w(~mask.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
    device.synchronize(); device.timer_start()
    for _ in range(20):
        w(~mask.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
    ms = device.timer_stop() / 20
    print(json.dumps({"scale": scale, "world": world, "rows": rows, "nnz": int(col.numel()), "active_entries": active, "ms": round(ms, 4),
                      "GTEPS_this_rank": round(active / ms / 1e6, 1), "ideal_ms": None, "stats": device.last_stats()}), flush=True)
    del A, u, w, mask, indptr, col, vals
