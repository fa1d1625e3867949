#!/usr/bin/env python3
"""GPU diagnostics: short rows of the split by the row-group kernel (1) vs the merge-path kernel (0)."""
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
    wl = bench.MxvWorkload(gb, torch, scale, 0, 1, sr, vis)
    for sk in (0, 1, 0, 1):
        _lib.lib.GrX_option_set(b"short_kernel", sk)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        device.timer_start()
        for _ in range(20):
            wl.step()
        ms = device.timer_stop() / 20
        print(json.dumps({"workload": name, "short_kernel": sk, "ms": round(ms, 4),
                          "GTEPS": round(wl.nnz_active_local / ms / 1e6, 1), "launches": device.last_stats()["kernel_launches"]}), flush=True)
    del wl
