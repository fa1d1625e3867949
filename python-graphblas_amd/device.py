"""Zero-copy bridges between torch device tensors and the library's HBM-resident objects (GrX_*
extensions of include/grb_mi355x.h).  torch is used for allocation / RNG / collectives only; every
GraphBLAS operation runs in the library's own HIP kernels."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .base import call_on
from .dtypes import lookup_dtype
from .matrix import Matrix
from .vector import Vector, _name_counter


def _dptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def matrix_from_device_csr(indptr, col_indices, values, nrows, ncols, dtype=None, *, iso=False, copy=False, name=None):
    """Adopt a CSR held in torch CUDA tensors: int64 indptr[nrows+1], int32 sorted col_indices[nnz],
    values[nnz] (or one value when ``iso``).  With copy=False the tensors are kept alive by the Matrix."""
    import torch

    assert indptr.dtype == torch.int64 and col_indices.dtype == torch.int32
    dtype = lookup_dtype(dtype if dtype is not None else str(values.dtype).replace("torch.", "").replace("float32", "FP32").replace("float64", "FP64"))
    A = Matrix.__new__(Matrix)
    A.dtype, A._nrows, A._ncols, A.name = dtype, int(nrows), int(ncols), name or f"M_{next(_name_counter)}"
    A._handle = ctypes.c_void_p()
    _lib.load()
    nnz = int(col_indices.numel())
    call_on(None, "GrX_Matrix_import_CSR_device",
            [ctypes.byref(A._handle), dtype._carg, A._nrows, A._ncols, _dptr(indptr), _dptr(col_indices), _dptr(values),
             nnz, 1 if iso else 0, 1 if copy else 0])
    if not copy:
        A._keepalive = (indptr, col_indices, values)
    return A


def pack_bits(present):
    """bool tensor [n] -> int32 presence words (bit i of word i>>5), padded to whole 64-bit words."""
    import torch

    n = present.numel()
    pad = (-n) % 64
    p = torch.nn.functional.pad(present.to(torch.int64), (0, pad)).view(-1, 64)
    w = (p << torch.arange(64, device=present.device, dtype=torch.int64)).sum(1)  # wraps into the sign bit as intended
    return w.view(torch.int32)


def vector_from_device(values, present=None, dtype=None, *, name=None):
    """Dense-with-presence vector from a torch CUDA tensor of values (+ optional bool presence)."""
    import torch

    dtype = lookup_dtype(dtype if dtype is not None else {torch.float32: "FP32", torch.float64: "FP64", torch.int64: "INT64",
                                                          torch.int32: "INT32", torch.bool: "BOOL", torch.uint8: "UINT8",
                                                          torch.int8: "INT8", torch.int16: "INT16"}[values.dtype])
    v = Vector.__new__(Vector)
    v.dtype, v._size, v.name = dtype, int(values.numel()), name or f"v_{next(_name_counter)}"
    v._handle = ctypes.c_void_p()
    _lib.load()
    bits = pack_bits(present) if present is not None else None
    call_on(None, "GrX_Vector_import_dense_device",
            [ctypes.byref(v._handle), dtype._carg, v._size, _dptr(values.contiguous()), _dptr(bits)])
    return v


class _CudaView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def vector_pin_natural(v, pinned=True):
    """Keep ``v`` in natural index order for good: its HBM image is aliased outside the library for longer than one call (collective
    buffers).  Products that involve a pinned vector run on the natural-order layouts of their matrix."""
    call_on(v, "GrX_Vector_pin_natural", [v._handle, 1 if pinned else 0])


def vector_release_views(v):
    """The caller no longer uses the tensors :func:`vector_device_views` returned for ``v``: the vector may live in a matrix's
    vertex order again (GrX_Vector_pin_natural(v, 0))."""
    vector_pin_natural(v, False)


def vector_device_views(v, device="cuda", *, pin=True):
    """(values, presence_words) torch tensors ALIASING the vector's HBM image (valid until the vector is resized, cleared or
    freed).  The image is in natural index order when this returns, and with ``pin=True`` (the default: the library pins a vector
    whose pointers it hands out) it stays so -- products that involve the vector then run on their matrix's natural-order layouts.
    ``pin=False`` is for TRANSIENT views: the caller uses them before its next library call on ``v`` and fetches them again
    afterwards (a product with a large square matrix may leave ``v`` in that matrix's vertex order, inside the same buffers);
    :func:`vector_release_views` does the same for views that were pinned.  ``device="cpu"`` is for the CPU test tier, where the
    emulator build keeps the image in host memory."""
    import torch

    dv, db = ctypes.c_void_p(), ctypes.c_void_p()
    call_on(v, "GrX_Vector_export_dense_device", [ctypes.byref(dv), ctypes.byref(db), v._handle])
    if not pin:
        vector_pin_natural(v, False)
    n = v._size
    if device == "cpu":
        nb = n * np.dtype(v.dtype.np_type).itemsize
        vals = np.ctypeslib.as_array((ctypes.c_uint8 * max(nb, 1)).from_address(dv.value))[:nb].view(v.dtype.np_type)
        nw = ((n + 63) // 64) * 2
        words = np.ctypeslib.as_array((ctypes.c_uint8 * max(nw * 4, 1)).from_address(db.value))[: nw * 4].view(np.int32)
        if v.dtype.np_type == np.dtype(bool):
            return torch.from_numpy(vals.view(np.uint8)).view(torch.bool), torch.from_numpy(words)
        return torch.from_numpy(vals), torch.from_numpy(words)
    vals = torch.as_tensor(_CudaView(dv.value, (n,), np.dtype(v.dtype.np_type).str), device="cuda")
    words = torch.as_tensor(_CudaView(db.value, (((n + 63) // 64) * 2,), "<i4"), device="cuda")
    return vals, words


def matrix_device_views(A):
    """(indptr int64[nrows+1], col int32[nvals], values[nvals] or [1] when iso, iso) torch CUDA tensors ALIASING A's CSR in HBM
    (valid until A is modified or freed) -- for checks that stay on the device."""
    import torch

    dp, dj, dx = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    nv, iso = ctypes.c_uint64(), ctypes.c_int()
    call_on(A, "GrX_Matrix_export_CSR_device", [ctypes.byref(dp), ctypes.byref(dj), ctypes.byref(dx), ctypes.byref(nv), ctypes.byref(iso), A._handle])
    nnz = int(nv.value)
    indptr = torch.as_tensor(_CudaView(dp.value, (A._nrows + 1,), "<i8"), device="cuda")
    col = torch.as_tensor(_CudaView(dj.value, (max(nnz, 1),), "<i4"), device="cuda")[:nnz]
    nx = 1 if iso.value else nnz
    vals = torch.as_tensor(_CudaView(dx.value, (max(nx, 1),), np.dtype(A.dtype.np_type).str), device="cuda")[:nx]
    return indptr, col, vals, bool(iso.value)


def trim_memory():
    """Hand the library's cached device memory back to the driver (between bench phases with different footprints)."""
    call_on(None, "GrX_trim_memory", [])


def synchronize():
    call_on(None, "GrX_synchronize", [])


def timer_start():
    call_on(None, "GrX_timer_start", [])


def timer_stop():
    ms = ctypes.c_float()
    call_on(None, "GrX_timer_stop", [ctypes.byref(ms)])
    return float(ms.value)


def last_stats():
    s = _lib.GrX_Stats()
    _lib.lib.GrX_last_stats(ctypes.byref(s))
    return {k: getattr(s, k) for k, _ in s._fields_}


def matrix_hint_ranked(A, ranked=True):
    """The labels of ``A`` are popularity ranks already (column 0 the most referred-to, heavy rows first): the library builds its
    popularity-ordered layouts in the caller's own index order -- no permutation, no vector conversion, any shape (GrX_Matrix_hint_ranked).
    A performance hint only."""
    call_on(A, "GrX_Matrix_hint_ranked", [A._handle, 1 if ranked else 0])


def matrix_shard_setup(A, col_counts=None, *, like=None):
    """``A`` is a row block of a sharded graph: hand the library the GLOBAL reference counts of its columns (a uint32 / int32 torch tensor of
    ``ncols`` elements on the library's device, a CPU tensor or a numpy array -- identical on every rank: the host layer all-reduces the ranks'
    column histograms once) and it ranks the columns itself and builds the block's popularity-ordered layouts in that column order, rows as
    they are (GrX_Matrix_shard_setup).  ``like``: another set-up block of the same width whose order object ``A`` shares (the chunks of one
    rank: an operand is then converted once for all of them).  A performance hint only."""
    if like is not None:
        call_on(A, "GrX_Matrix_shard_setup", [A._handle, None, 0, like._handle])
        return
    import numpy as np

    on_device = 0
    keep = None
    if hasattr(col_counts, "data_ptr"):  # a torch tensor
        import torch

        keep = col_counts.to(torch.int32).contiguous() if col_counts.dtype != torch.int32 else col_counts.contiguous()
        if keep.numel() != A.ncols:
            raise ValueError("matrix_shard_setup: one count per column")
        on_device = 1 if keep.is_cuda else 0
        ptr = ctypes.c_void_p(keep.data_ptr())
    else:
        keep = np.ascontiguousarray(col_counts, dtype=np.uint32)
        if keep.size != A.ncols:
            raise ValueError("matrix_shard_setup: one count per column")
        ptr = keep.ctypes.data_as(ctypes.c_void_p)
    call_on(A, "GrX_Matrix_shard_setup", [A._handle, ptr, on_device, None])


def cache_transpose(A):
    call_on(A, "GrX_Matrix_cache_transpose", [A._handle])


def vector_modified(v):
    """Call after writing into the tensors returned by :func:`vector_device_views`."""
    call_on(v, "GrX_Vector_modified", [v._handle])
