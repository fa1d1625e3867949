#!/usr/bin/env python3
"""GPU diagnostics: long/short row split threshold sweep on the bench graph."""
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
    for min_len in (64, 128, 192, 256, 384, 512):
        _lib.lib.GrX_option_set(b"split_min_nnz", (1 << 22) if min_len else (1 << 60))
        _lib.lib.GrX_option_set(b"split_min_len", max(min_len, 1))
        wl = bench.MxvWorkload(gb, torch, scale, 0, 1, sr, vis)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        device.timer_start()
        for _ in range(10):
            wl.step()
        ms = device.timer_stop() / 10
        print(json.dumps({"workload": name, "split_min_len": min_len, "ms": round(ms, 4), "GTEPS": round(wl.nnz_active_local / ms / 1e6, 1),
                          "launches": device.last_stats()["kernel_launches"], "tiles": device.last_stats()["tiles"]}), flush=True)
        del wl
