#!/usr/bin/env python3
"""Is part of the layout-building call a once-per-process cost?  Two matrices of the headline graph, one after the other in ONE process:
first call (plain path), second call (builds the layouts), both timed as bench.py times them.
    python scripts/build_twice.py [scale]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import graphblas_amd as gb

gb.init()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out = []
for k in range(3):
    wl = bench.MxvWorkload(gb, torch, scale, 0, 1, "min_plus", 0.5)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        wl.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    out.append({"matrix": k, "first_call_ms": round(ts[0], 3), "layout_build_call_ms": round(ts[1], 3), "third_call_ms": round(ts[2], 3)})
    print(json.dumps(out[-1]), flush=True)
    del wl
