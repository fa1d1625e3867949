#!/usr/bin/env python3
"""GPU (or emulator: pass `emu`) stress: the random parity tests with seeds beyond the ones pytest runs."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.backend import bind
import tests.test_random_parity as T

dev = sys.argv[1] if len(sys.argv) > 1 else "gpu"
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 140)
gb = bind(dev)
names = ["test_mxv_random", "test_vxm_and_transposes_random", "test_hot_column_table", "test_push_direction",
         "test_long_short_row_split", "test_mxm_random", "test_mxm_mask_driven", "test_mxm_units_random", "test_vector_assign_reduce_random",
         "test_vector_ewise_random", "test_pair_over_full_operand", "test_sell_short_rows", "test_reductions_over_split_matrices",
         "test_mixed_types_unread_operands"]
fails = 0
for name in names:
    f = getattr(T, name)
    f = getattr(f, "__wrapped__", f)
    for seed in range(lo, hi):
        try:
            f(gb, seed)
        except BaseException as e:
            if type(e).__name__ == "Skipped":  # pytest.skip inside a test (no such semiring for the drawn types)
                continue
            fails += 1
            print("FAIL", name, seed)
            traceback.print_exc(limit=3)
print("stress done:", len(names) * (hi - lo), "cases,", fails, "failures")
