"""Selects which build of the kernel sources a test process drives.

* ``gpu``: the product library python-graphblas_amd/csrc/libgrb_mi355x.so on a real MI355X.
* ``emu``: tests/emu/libgrb_emu.so -- the SAME kernel sources compiled for the CPU wave64 SIMT
  emulator (test infrastructure; lets the CPU-only tier check kernel logic against the oracle).
One process can bind only one of them (the Python host holds a single library handle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "libgrb_emu.so")
_bound = None


def bind(dev):
    global _bound
    import graphblas_amd as gb

    if _bound is None:
        if dev == "emu":
            if not os.environ.get("GRB_EMU_PREBUILT"):  # (worker processes of a multi-rank test reuse the parent's build)
                import fcntl

                # (pytest-xdist workers start together: one of them rebuilds a stale library, the others wait instead of loading it half-linked)
                with open(os.path.join(ROOT, "tests", "emu", ".build.lock"), "w") as lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
            gb.init(lib_path=EMU_SO)
        else:
            gb.init()
        _bound = dev
    elif _bound != dev:
        pytest.skip(f"this process is already bound to the {_bound!r} library")
    return gb


DEVICES = [pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)]
