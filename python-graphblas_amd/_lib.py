"""ctypes binding of libgrb_mi355x.so -- the counterpart of the reference's cffi ``lib``/``ffi``
pair published by graphblas/__init__.py:143-199 and looked up per call by
graphblas/core/utils.py:11-23 (``libget``).

The library is loaded from the package's ``csrc/`` directory (built in-tree by
``__graft_entry__.build()``).  There is no CPU implementation behind it: ``GrB_init`` returns
GrB_PANIC when no gfx950 device is present and :func:`init` raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libgrb_mi355x.so")

lib = None
_handles = {}

c_void_p = ctypes.c_void_p
c_u64 = ctypes.c_uint64
c_int = ctypes.c_int

TYPE_NAMES = ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"]

# GrB_Info values (include/grb_mi355x.h)
GrB_SUCCESS = 0
GrB_NO_VALUE = 1


class GrX_Stats(ctypes.Structure):
    _fields_ = [("kernel_launches", ctypes.c_int64), ("tiles", ctypes.c_int64), ("flops", ctypes.c_int64),
                ("out_nvals", ctypes.c_int64), ("method", ctypes.c_int32), ("fused_epilogue", ctypes.c_int32),
                ("hot_k", ctypes.c_int64), ("long_entries", ctypes.c_int64), ("long_segments", ctypes.c_int64),
                ("long_kernel", ctypes.c_int32), ("reorders", ctypes.c_int32), ("ordered", ctypes.c_int64), ("value_dict", ctypes.c_int64), ("fill_absent", ctypes.c_int64),
                ("long_probe", ctypes.c_int64), ("long_tails", ctypes.c_int64), ("pinned_natural", ctypes.c_int64)]


def load(path: str | None = None):
    """Open the shared library (idempotent).  ``path`` overrides the in-tree location; the CPU test
    tier uses that to point at the SIMT-emulator build of the same sources (tests/emu)."""
    global lib
    if lib is not None:
        return lib
    path = path or os.environ.get("GRB_MI355X_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no pure-Python or CPU fallback)"
        )
    if os.path.basename(path) == os.path.basename(DEFAULT_LIB):  # (the CPU emulator build has no HIP runtime)
        _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    _declare(lib)
    return lib


def _share_hip_runtime_with_torch():
    """torch wheels bundle their own libamdhip64.so.7; a process must not end up with the system copy
    loaded first and torch's second (torch then reports "no ROCm-capable device").  When torch is
    installed, load ITS runtime before our library so both resolve the SONAME to the same copy."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _declare(L):
    P = ctypes.POINTER
    L.GrB_init.argtypes = [c_int]
    L.GrB_finalize.argtypes = []
    for name in ("GrB_mxm", "GrB_mxv", "GrB_vxm"):
        getattr(L, name).argtypes = [c_void_p] * 7
    L.GrB_Matrix_new.argtypes = [P(c_void_p), c_void_p, c_u64, c_u64]
    L.GrB_Matrix_dup.argtypes = [P(c_void_p), c_void_p]
    L.GrB_Matrix_free.argtypes = [P(c_void_p)]
    L.GrB_Matrix_clear.argtypes = [c_void_p]
    L.GrB_Matrix_resize.argtypes = [c_void_p, c_u64, c_u64]
    L.GrB_Matrix_reduce_Monoid.argtypes = [c_void_p] * 6
    for name in ("GrB_Matrix_nrows", "GrB_Matrix_ncols", "GrB_Matrix_nvals", "GrB_Vector_size", "GrB_Vector_nvals"):
        getattr(L, name).argtypes = [P(c_u64), c_void_p]
    L.GrB_Matrix_wait.argtypes = [c_void_p, c_int]
    L.GrB_Vector_wait.argtypes = [c_void_p, c_int]
    L.GrB_Matrix_error.argtypes = [P(ctypes.c_char_p), c_void_p]
    L.GrB_Vector_error.argtypes = [P(ctypes.c_char_p), c_void_p]
    L.GrB_Matrix_exportSize.argtypes = [P(c_u64), P(c_u64), P(c_u64), c_int, c_void_p]
    # (the reference's zero-copy ingress names, core/ss/matrix.py:1316-1333: array cells by address, sizes in bytes, iso, jumbled, desc)
    L.GxB_Matrix_import_CSR.argtypes = [P(c_void_p), c_void_p, c_u64, c_u64, P(c_void_p), P(c_void_p), P(c_void_p), c_u64, c_u64, c_u64,
                                        ctypes.c_bool, ctypes.c_bool, c_void_p]
    L.GxB_Matrix_pack_CSR.argtypes = [c_void_p, P(c_void_p), P(c_void_p), P(c_void_p), c_u64, c_u64, c_u64, ctypes.c_bool, ctypes.c_bool, c_void_p]
    L.GrB_transpose.argtypes = [c_void_p] * 5
    L.GrB_Vector_new.argtypes = [P(c_void_p), c_void_p, c_u64]
    L.GrB_Vector_dup.argtypes = [P(c_void_p), c_void_p]
    L.GrB_Vector_free.argtypes = [P(c_void_p)]
    L.GrB_Vector_clear.argtypes = [c_void_p]
    L.GrB_Vector_removeElement.argtypes = [c_void_p, c_u64]
    for name in ("GrB_Vector_eWiseAdd_BinaryOp", "GrB_Vector_eWiseAdd_Monoid", "GrB_Vector_eWiseMult_BinaryOp", "GrB_Vector_eWiseMult_Monoid"):
        getattr(L, name).argtypes = [c_void_p] * 7
    for t, ct in (("BOOL", ctypes.c_bool), ("INT8", ctypes.c_int8), ("INT16", ctypes.c_int16), ("INT32", ctypes.c_int32),
                  ("INT64", ctypes.c_int64), ("UINT8", ctypes.c_uint8), ("UINT16", ctypes.c_uint16), ("UINT32", ctypes.c_uint32),
                  ("UINT64", ctypes.c_uint64), ("FP32", ctypes.c_float), ("FP64", ctypes.c_double)):
        getattr(L, f"GrB_Vector_setElement_{t}").argtypes = [c_void_p, ct, c_u64]
        getattr(L, f"GrB_Vector_extractElement_{t}").argtypes = [P(ct), c_void_p, c_u64]
        getattr(L, f"GrB_Vector_assign_{t}").argtypes = [c_void_p, c_void_p, c_void_p, ct, c_void_p, c_u64, c_void_p]
        getattr(L, f"GrB_Vector_reduce_{t}").argtypes = [P(ct), c_void_p, c_void_p, c_void_p, c_void_p]
    L.GrB_Vector_resize.argtypes = [c_void_p, c_u64]
    for t in TYPE_NAMES:
        getattr(L, f"GrB_Matrix_build_{t}").argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_void_p]
        getattr(L, f"GrB_Matrix_extractTuples_{t}").argtypes = [c_void_p, c_void_p, c_void_p, P(c_u64), c_void_p]
        getattr(L, f"GrB_Matrix_import_{t}").argtypes = [P(c_void_p), c_void_p, c_u64, c_u64, c_void_p, c_void_p,
                                                          c_void_p, c_u64, c_u64, c_u64, c_int]
        getattr(L, f"GrB_Matrix_export_{t}").argtypes = [c_void_p, c_void_p, c_void_p, P(c_u64), P(c_u64), P(c_u64),
                                                          c_int, c_void_p]
        getattr(L, f"GrB_Vector_build_{t}").argtypes = [c_void_p, c_void_p, c_void_p, c_u64, c_void_p]
        getattr(L, f"GrB_Vector_extractTuples_{t}").argtypes = [c_void_p, c_void_p, P(c_u64), c_void_p]
    L.GrX_Matrix_import_CSR_device.argtypes = [P(c_void_p), c_void_p, c_u64, c_u64, c_void_p, c_void_p, c_void_p,
                                                c_u64, c_int, c_int]
    L.GrX_Matrix_export_CSR_device.argtypes = [P(c_void_p), P(c_void_p), P(c_void_p), P(c_u64), P(c_int), c_void_p]
    L.GrX_Matrix_cache_transpose.argtypes = [c_void_p]
    L.GrX_Matrix_hint_ranked.argtypes = [c_void_p, c_int]
    L.GrX_Matrix_shard_setup.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    L.GrX_Vector_import_dense_device.argtypes = [P(c_void_p), c_void_p, c_u64, c_void_p, c_void_p]
    L.GrX_Vector_export_dense_device.argtypes = [P(c_void_p), P(c_void_p), c_void_p]
    L.GrX_Vector_modified.argtypes = [c_void_p]
    L.GrX_Vector_pin_natural.argtypes = [c_void_p, ctypes.c_int]
    L.GrX_Vector_fill_absent.argtypes = [c_void_p, c_void_p]
    L.GrX_get_stream.argtypes = [P(c_void_p)]
    L.GrB_Vector_assign.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_void_p]
    L.GrB_Vector_extract.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_void_p]
    L.GrX_option_set.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    L.GrX_set_stream.argtypes = [c_void_p]
    L.GrX_synchronize.argtypes = []
    L.GrX_trim_memory.argtypes = []
    L.GrX_timer_start.argtypes = []
    L.GrX_timer_stop.argtypes = [P(ctypes.c_float)]
    L.GrX_last_stats.argtypes = [P(GrX_Stats)]
    L.GrX_Matrix_cache_bytes.argtypes = [c_void_p, P(c_u64)]
    L.GrX_Matrix_isclose.argtypes = [P(ctypes.c_bool), c_void_p, c_void_p, ctypes.c_double, ctypes.c_double]
    L.GrX_Matrix_dup_as.argtypes = [P(c_void_p), c_void_p, c_void_p]
    L.GrX_Vector_dup_as.argtypes = [P(c_void_p), c_void_p, c_void_p]
    L.GrX_mxm_streamed.argtypes = [c_void_p, c_void_p, c_void_p, c_u64, P(c_u64), P(c_u64), P(c_u64), P(c_u64)]
    L.GrX_version_string.restype = ctypes.c_char_p


def handle(name: str) -> int:
    """Address held by the global handle symbol ``name`` (e.g. ``GrB_MIN_PLUS_SEMIRING_FP32``) --
    what ``getattr(lib, name)`` yields in the reference (core/operator/base.py:868-893)."""
    h = _handles.get(name)
    if h is None:
        h = c_void_p.in_dll(load(), name).value
        _handles[name] = h
    return h


def has_symbol(name: str) -> bool:
    try:
        handle(name)
        return True
    except ValueError:
        return False


def all_indices():
    """The value of the library's ``GrB_ALL`` pointer (``v[:]``; reference core/expr.py:14)."""
    return ctypes.c_void_p(ctypes.c_void_p.in_dll(lib, "GrB_ALL").value)
