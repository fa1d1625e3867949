#!/usr/bin/env python3
"""GPU diagnostics: time the pull SpMV kernel variants on the bench graph (one process, one graph)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import graphblas_amd as gb  # noqa: E402
from graphblas_amd import _lib, device  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
IPTS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '0,4,16').split(',')]
DBGS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0,1,2,3,4,7').split(',')]
gb.init()
out = []
for wl_name, sr, vis in (("min_plus_masked", "min_plus", 0.5), ("min_plus_unmasked", "min_plus", 0.0),
                         ("lor_land_masked", "lor_land", 0.5)):
    wl = bench.MxvWorkload(gb, torch, scale, 0, 1, sr, vis)
    for ipt in IPTS:
        for dbg in DBGS:
            _lib.lib.GrX_option_set(b'debug_flags', dbg); _lib.lib.GrX_option_set(b'pull_ipt', ipt)
            for _ in range(2):
                wl.step()
            torch.cuda.synchronize()
            device.timer_start()
            for _ in range(10):
                wl.step()
            ms = device.timer_stop() / 10
            rec = {"workload": wl_name, "ipt": ipt or 8, "dbg": dbg, "ms": round(ms, 4),
                   "GTEPS": round(wl.nnz_active_local / ms / 1e6, 1), "GBs_alg": round(wl.bytes_per_step() / ms / 1e6, 1)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    _lib.lib.GrX_option_set(b'debug_flags', 0); _lib.lib.GrX_option_set(b'pull_ipt', 0)
    del wl
