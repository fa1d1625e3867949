#!/usr/bin/env python3
"""GPU diagnostics: hot-table size sweep on the bench graph."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import graphblas_amd as gb
from graphblas_amd import _lib, device
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gb.init()
for name, sr, vis in (("min_plus_masked", "min_plus", 0.5), ("min_plus_unmasked", "min_plus", 0.0), ("lor_land_masked", "lor_land", 0.5)):
    for hot_k in (1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
        _lib.lib.GrX_option_set(b"hot_k", hot_k if hot_k else 0)
        _lib.lib.GrX_option_set(b"hot_min_cols", (1 << 20) if hot_k else (1 << 40))
        wl = bench.MxvWorkload(gb, torch, scale, 0, 1, sr, vis)   # new matrix object => table rebuilt with this K
        for dbg in (0,):
            _lib.lib.GrX_option_set(b"debug_flags", dbg)
            for _ in range(3):
                wl.step()
            torch.cuda.synchronize()
            device.timer_start()
            for _ in range(10):
                wl.step()
            ms = device.timer_stop() / 10
            print(json.dumps({"workload": name, "hot_k": hot_k, "ms": round(ms, 4),
                              "GTEPS": round(wl.nnz_active_local / ms / 1e6, 1), "stats_hot_k": device.last_stats()["hot_k"]}), flush=True)
        _lib.lib.GrX_option_set(b"debug_flags", 0)
        del wl
