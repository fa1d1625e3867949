#!/usr/bin/env python3
"""GPU diagnostics: per-phase cycle stamps of k_mxv_pull (GRB_DEBUG_FLAGS bit 8) on the bench graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import graphblas_amd as gb
from graphblas_amd import _lib, device
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gb.init()
for name, sr, vis in (("min_plus_masked", "min_plus", 0.5), ("lor_land_masked", "lor_land", 0.5)):
    wl = bench.MxvWorkload(gb, torch, scale, 0, 1, sr, vis)
    for ipt in (0, 4):
        for dbg in (8, 15):
            _lib.lib.GrX_option_set(b"pull_ipt", ipt)
            _lib.lib.GrX_option_set(b"debug_flags", 0)
            for _ in range(2):
                wl.step()
            torch.cuda.synchronize()
            print(f"== {name} ipt={ipt or 8} dbg={dbg}", file=sys.stderr, flush=True)
            _lib.lib.GrX_option_set(b"debug_flags", dbg)
            wl.step()
            torch.cuda.synchronize()
    _lib.lib.GrX_option_set(b"debug_flags", 0); _lib.lib.GrX_option_set(b"pull_ipt", 0)
    del wl
