"""CPU tier: the body of ``__graft_entry__.smoke()`` on the emulator build of the kernel sources -- the same small masked products of
every hot semiring through the default path, the round-3 split kernels and the popularity-ordered layouts with the sorted row tiles
(k_mxv_rtile, k_mxv_rtile_bool, the bottom-up probe), checked against the oracle, with the assertions about WHICH kernels ran.
What the driver runs on the GPU box at round end has then run here first (smoke() forced the round-3 kernels only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_smoke_body_on_the_emulator():
    from tests.backend import bind

    gb = bind("emu")
    import __graft_entry__ as entry

    entry._smoke_body(gb, "cpu")
