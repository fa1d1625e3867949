#!/usr/bin/env python3
"""GPU (or emulator: pass `emu`) stress: the random parity tests with seeds beyond the ones pytest runs."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.backend import bind
import tests.test_random_parity as T
import tests.test_vertex_order as V

dev = sys.argv[1] if len(sys.argv) > 1 else "gpu"
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 140)
gb = bind(dev)
import inspect


def seeded(mod, skip=()):
    """every test of the module whose arguments are (gb, seed): the ones a new seed means new data for"""
    out = []
    for n, f in sorted(vars(mod).items()):
        if n.startswith("test_") and callable(f) and n not in skip:
            f = getattr(f, "__wrapped__", f)
            if list(inspect.signature(f).parameters) == ["gb", "seed"]:
                out.append(n)
    return out


# (a test that indexes a list of hand-written configurations by its seed runs with pytest's seeds only: its IndexError ends its turn here)
names = seeded(T)
# (round 5) the seeded tests of the ordered layouts, the sorted row tiles, the bottom-up probe and the ranked hint
# (test_ordered_vectors_through_every_exit indexes a list by its seed: pytest's four seeds only)
vnames = ["test_ordered_product_matches_the_oracle", "test_sssp_and_bfs_loops_stay_ordered",
          "test_sorted_row_tiles_match_the_oracle", "test_sorted_row_tiles_bool_step", "test_bottom_up_probe_of_long_bool_rows",
          "test_ranked_hint_orders_without_a_permutation",
          "test_shard_setup_orders_a_row_block_by_global_column_counts"]  # (round 6; the row-tile test now also draws cold_in_rows and rtile_pack)
which = os.environ.get("STRESS_SET", "all")  # random | order | all
todo = ([(T, n) for n in names] if which in ("random", "all") else []) + ([(V, n) for n in vnames] if which in ("order", "all") else [])
fails = 0
ran = 0
layout = 0  # draws whose matrix did not take the layout the test expects (its assertion on GrX_Stats, not on values): counted, not failures
for mod, name in todo:
    f = getattr(mod, name)
    f = getattr(f, "__wrapped__", f)
    for seed in range(lo, hi):
        try:
            ran += 1
            f(gb, seed)
        except IndexError:
            tb = traceback.extract_tb(sys.exc_info()[2])
            if os.path.basename(tb[-1].filename).startswith("test_"):
                print("seed-indexed test, pytest's seeds only:", name)
                ran -= 1
                break
            fails += 1
            print("FAIL", name, seed)
            traceback.print_exc(limit=3)
        except BaseException as e:
            if type(e).__name__ == "Skipped":  # pytest.skip inside a test (no such semiring for the drawn types)
                continue
            if isinstance(e, AssertionError) and str(e).startswith("{'kernel_launches'"):
                layout += 1
                print("layout expectation", name, seed, str(e)[:160])
                continue
            fails += 1
            print("FAIL", name, seed)
            traceback.print_exc(limit=3)
print("stress done:", ran, "cases,", fails, "failures,", layout, "draws off the expected layout")
