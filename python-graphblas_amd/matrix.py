"""Matrix / TransposedMatrix: the host-side mirror of graphblas/core/matrix.py for the path --
constructor :190-203, ``__del__`` :218-225, ``build`` :627-681, ``from_coo`` :818-894, ``_from_csx``
:992-1068, ``to_coo`` :525-594, ``_to_csx`` :1601-1645, ``isequal`` :373-415, ``mxv`` :2203-2262,
``mxm`` :2264-2331, ``TransposedMatrix`` :3900-3960."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .base import BaseType, Expression, InfixMatMul, call, call_on
from .dtypes import lookup_dtype
from .exceptions import DimensionMismatch
from .operators import get_typed_op, semiring as _semiring
from .vector import Vector, _name_counter, _ptr

GrB_CSR_FORMAT, GrB_CSC_FORMAT = 0, 1


def _owned_copy(arr):
    """A malloc'd copy of a numpy array: the GxB import / pack entries take OWNERSHIP of their host arrays and release them with
    the library's deallocator (the reference hands over numpy buffers it has unclaimed, core/ss/matrix.py:1300-1349)."""
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    nbytes = max(int(arr.nbytes), 1)
    p = libc.malloc(nbytes)
    if not p:
        raise MemoryError
    if arr.nbytes:
        ctypes.memmove(p, arr.ctypes.data, arr.nbytes)
    return ctypes.c_void_p(p), int(arr.nbytes)


class _MatrixSS:
    """``Matrix.ss`` / ``A.ss``: the reference's SuiteSparse namespace (graphblas/core/ss/matrix.py), reduced to the ingress the
    hot path's callers use -- ``import_csr`` (:1139-1237) and ``pack_csr`` (:1239-1277), both through ``_import_csr``
    (:1279-1349) -> ``GxB_Matrix_{import,pack}_CSR``.  Same keyword arguments; ``take_ownership`` is accepted and means nothing
    here (the matrix lives in HBM: the host arrays are copied once either way, and the caller's numpy arrays are left alone)."""

    def __init__(self, parent=None):
        self._parent = parent

    @staticmethod
    def _csr_args(indptr, values, col_indices, dtype, is_iso, fmt):
        if fmt is not None and fmt.lower() != "csr":
            raise ValueError(f"Invalid format: {fmt!r}.  Must be None or 'csr'.")
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        col_indices = np.ascontiguousarray(col_indices, dtype=np.uint64)
        values = np.asarray(values)
        dtype = lookup_dtype(dtype if dtype is not None else values.dtype)
        values = np.ascontiguousarray(np.atleast_1d(values).astype(dtype.np_type, copy=False))
        if is_iso:
            values = values[:1]
        ptrs = [_owned_copy(a) for a in (indptr, col_indices, values)]
        cells = [ctypes.c_void_p(p.value) for p, _ in ptrs]
        return dtype, cells, [nb for _, nb in ptrs]

    @staticmethod
    def _release_leftovers(cells):
        libc = ctypes.CDLL(None)
        libc.free.argtypes = [ctypes.c_void_p]
        for c in cells:  # (a failed call leaves the arrays with the caller: *Ap & co. are still set)
            if c.value:
                libc.free(c)

    def import_csr(self, *, nrows, ncols, indptr, values, col_indices, is_iso=False, sorted_cols=False, take_ownership=False,
                   dtype=None, format=None, name=None, **opts):
        dtype, cells, sizes = self._csr_args(indptr, values, col_indices, dtype, is_iso, format)
        A = Matrix.__new__(Matrix)
        A.dtype, A._nrows, A._ncols, A.name = dtype, int(nrows), int(ncols), name or f"M_{next(_name_counter)}"
        A._handle = ctypes.c_void_p()
        _lib.load()
        try:
            call_on(None, "GxB_Matrix_import_CSR",
                    [ctypes.byref(A._handle), dtype._carg, A._nrows, A._ncols, ctypes.byref(cells[0]), ctypes.byref(cells[1]),
                     ctypes.byref(cells[2]), sizes[0], sizes[1], sizes[2], bool(is_iso), not sorted_cols, None])
        finally:
            self._release_leftovers(cells)
        return A

    def pack_csr(self, *, indptr, values, col_indices, is_iso=False, sorted_cols=False, take_ownership=False, format=None,
                 nrows=None, ncols=None, dtype=None, name=None, **opts):
        A = self._parent
        if A is None:
            raise TypeError("pack_csr packs into an existing Matrix: call it on an instance (A.ss.pack_csr(...))")
        _dt, cells, sizes = self._csr_args(indptr, values, col_indices, A.dtype, is_iso, format)
        try:
            call_on(A, "GxB_Matrix_pack_CSR",
                    [A._handle, ctypes.byref(cells[0]), ctypes.byref(cells[1]), ctypes.byref(cells[2]), sizes[0], sizes[1], sizes[2],
                     bool(is_iso), not sorted_cols, None])
        finally:
            self._release_leftovers(cells)
        return A


class _SSAccessor:
    def __get__(self, obj, objtype=None):
        return _MatrixSS(obj)


class Matrix(BaseType):
    _grb_kind = "Matrix"
    ndim = 2
    _is_transposed = False
    ss = _SSAccessor()  # Matrix.ss.import_csr(...), A.ss.pack_csr(...): reference core/ss/matrix.py:1139-1349

    def __init__(self, dtype=float, nrows=0, ncols=0, *, name=None):
        self.dtype = lookup_dtype(dtype)
        self._nrows, self._ncols = int(nrows), int(ncols)
        self.name = name or f"M_{next(_name_counter)}"
        self._handle = ctypes.c_void_p()
        _lib.load()
        call_on(None, "GrB_Matrix_new", [ctypes.byref(self._handle), self.dtype._carg, self._nrows, self._ncols])

    @property
    def _carg(self):
        return self._handle

    @property
    def _matrix(self):
        return self

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and _lib is not None and _lib.lib is not None:
            try:
                _lib.lib.GrB_Matrix_free(ctypes.byref(h))
            except Exception:  # interpreter shutdown
                pass

    nrows = property(lambda self: self._nrows)
    ncols = property(lambda self: self._ncols)
    shape = property(lambda self: (self._nrows, self._ncols))

    @property
    def nvals(self):
        n = ctypes.c_uint64()
        call_on(self, "GrB_Matrix_nvals", [ctypes.byref(n), self._handle])
        return int(n.value)

    @property
    def T(self):
        return TransposedMatrix(self)

    def __repr__(self):
        return f"Matrix<{self.dtype}, {self._nrows}x{self._ncols}, name={self.name}>"

    # ---- ingress / egress ---------------------------------------------------------------------------------
    def build(self, rows, columns, values, *, dup_op=None, clear=False):
        """reference core/matrix.py:627-681: uint64 indices (:637-638), default dup_op=plus with
        duplicate detection by nvals < n (:657-681)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        columns = np.ascontiguousarray(columns, dtype=np.uint64)
        values = np.asarray(values)
        if values.ndim == 0:
            values = np.broadcast_to(values, rows.shape)
        values = np.ascontiguousarray(values.astype(self.dtype.np_type, copy=False))
        n = values.size
        if rows.size != n or columns.size != n:
            raise ValueError(f"`rows` and `columns` and `values` lengths must match: {rows.size}, {columns.size}, {n}")
        if clear:
            self.clear()
        if n == 0:
            return
        dup_orig = dup_op
        if dup_op is None:
            dup_op = "plus"
        dup_t = get_typed_op(dup_op, self.dtype, kind="binary")
        if dup_t.opclass == "Monoid":
            dup_t = dup_t.binaryop
        call(f"GrB_Matrix_build_{self.dtype.name}", [self, _ptr(rows), _ptr(columns), _ptr(values), n, dup_t])
        if dup_orig is None and self.nvals < n:
            self.clear()
            raise ValueError("Duplicate indices found, must provide `dup_op` BinaryOp")

    @classmethod
    def from_coo(cls, rows, columns, values=1.0, dtype=None, *, nrows=None, ncols=None, dup_op=None, name=None):
        """reference core/matrix.py:818-894: shape inferred as max index + 1 when not given."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        columns = np.ascontiguousarray(columns, dtype=np.uint64)
        values = np.asarray(values)
        if dtype is None:
            dtype = lookup_dtype(values.dtype) if values.dtype.kind in "biuf" else lookup_dtype(float)
        if nrows is None:
            if rows.size == 0:
                raise ValueError("No row indices provided. Unable to infer nrows.")
            nrows = int(rows.max()) + 1
        if ncols is None:
            if columns.size == 0:
                raise ValueError("No column indices provided. Unable to infer ncols.")
            ncols = int(columns.max()) + 1
        C = cls(dtype, nrows, ncols, name=name)
        C.build(rows, columns, values, dup_op=dup_op)
        return C

    @classmethod
    def _from_csx(cls, fmt, indptr, indices, values, dtype, num, name):
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        values = np.asarray(values)
        if dtype is None:
            dtype = lookup_dtype(values.dtype)
        dtype = lookup_dtype(dtype)
        if values.ndim == 0:
            values = np.broadcast_to(values, indices.shape)
        values = np.ascontiguousarray(values.astype(dtype.np_type, copy=False))
        if num is None:
            num = int(indices.max()) + 1 if indices.size else 0
        nrows, ncols = (indptr.size - 1, num) if fmt == GrB_CSR_FORMAT else (num, indptr.size - 1)
        A = cls.__new__(cls)
        A.dtype, A._nrows, A._ncols, A.name = dtype, nrows, ncols, name or f"M_{next(_name_counter)}"
        A._handle = ctypes.c_void_p()
        _lib.load()
        call_on(None, f"GrB_Matrix_import_{dtype.name}",
                [ctypes.byref(A._handle), dtype._carg, nrows, ncols, _ptr(indptr), _ptr(indices), _ptr(values),
                 indptr.size, indices.size, values.size, fmt])
        return A

    @classmethod
    def from_csr(cls, indptr, col_indices, values=1.0, dtype=None, *, ncols=None, name=None):
        return cls._from_csx(GrB_CSR_FORMAT, indptr, col_indices, values, dtype, ncols, name)

    @classmethod
    def from_csc(cls, indptr, row_indices, values=1.0, dtype=None, *, nrows=None, name=None):
        return cls._from_csx(GrB_CSC_FORMAT, indptr, row_indices, values, dtype, nrows, name)

    def to_coo(self, dtype=None, *, rows=True, columns=True, values=True, sort=True):
        """reference core/matrix.py:525-594 (row-major sorted)."""
        n = self.nvals
        I = np.empty(n, np.uint64) if rows else None
        J = np.empty(n, np.uint64) if columns else None
        X = np.empty(n, self.dtype.np_type) if values else None
        cnt = ctypes.c_uint64(n)
        call_on(self, f"GrB_Matrix_extractTuples_{self.dtype.name}",
                [_ptr(I) if n else None, _ptr(J) if n else None, _ptr(X) if n else None, ctypes.byref(cnt), self._handle])
        if X is not None and dtype is not None:
            X = X.astype(lookup_dtype(dtype).np_type)
        return I, J, X

    def _to_csx(self, fmt):
        ap, ai, ax = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        call_on(self, "GrB_Matrix_exportSize", [ctypes.byref(ap), ctypes.byref(ai), ctypes.byref(ax), fmt, self._handle])
        Ap = np.empty(ap.value, np.uint64)
        Ai = np.empty(ai.value, np.uint64)
        Ax = np.empty(ax.value, self.dtype.np_type)
        call_on(self, f"GrB_Matrix_export_{self.dtype.name}",
                [_ptr(Ap), Ai.ctypes.data_as(ctypes.c_void_p), Ax.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ap),
                 ctypes.byref(ai), ctypes.byref(ax), fmt, self._handle])
        return Ap[: ap.value], Ai[: ai.value], Ax[: ax.value]

    def to_csr(self):
        return self._to_csx(GrB_CSR_FORMAT)

    def to_csc(self):
        return self._to_csx(GrB_CSC_FORMAT)

    def dup(self, dtype=None, *, clear=False, mask=None, name=None):
        """A copy; ``dtype``: with the values cast (on the device); ``clear``: same shape and dtype, no entries; ``mask``: only the
        entries the mask lets through (reference core/matrix.py:469-497; tests/test_matrix.py:66-96)."""
        dt = self.dtype if dtype is None else lookup_dtype(dtype)
        if clear:
            return Matrix(dt, self._nrows, self._ncols, name=name)
        if mask is not None:
            C = Matrix(dt, self._nrows, self._ncols, name=name)
            C(mask=mask) << self
            return C
        C = Matrix.__new__(Matrix)
        C.dtype, C._nrows, C._ncols, C.name = dt, self._nrows, self._ncols, name or f"M_{next(_name_counter)}"
        C._handle = ctypes.c_void_p()
        if dt is self.dtype:
            call_on(self, "GrB_Matrix_dup", [ctypes.byref(C._handle), self._handle])
        else:  # the typecast copy is made on the device
            call_on(self, "GrX_Matrix_dup_as", [ctypes.byref(C._handle), dt._carg, self._handle])
        return C

    def clear(self):
        call("GrB_Matrix_clear", [self])

    def resize(self, nrows, ncols):
        """In place: growing adds empty rows / columns, shrinking drops the entries beyond the new bounds
        (reference core/matrix.py:512-523)."""
        nrows, ncols = int(nrows), int(ncols)
        if nrows < 0 or ncols < 0:
            raise ValueError("nrows and ncols must be non-negative")
        call("GrB_Matrix_resize", [self, nrows, ncols])
        self._nrows, self._ncols = nrows, ncols

    def isequal(self, other, *, check_dtype=False):
        """Same shape, structure and values (reference core/matrix.py:373-415)."""
        if not isinstance(other, Matrix):
            raise TypeError(f"Expected type: Matrix; got {type(other).__name__}")
        if check_dtype and self.dtype is not other.dtype:
            return False
        if self.shape != other.shape or self.nvals != other.nvals:
            return False
        return self._compare_on_device(other, 0.0, 0.0)

    def _compare_on_device(self, other, rel_tol, abs_tol):
        """pattern and values compared by the library on the device: only the verdict comes back"""
        eq = ctypes.c_bool(False)
        call_on(self, "GrX_Matrix_isclose", [ctypes.byref(eq), self._handle, other._handle, ctypes.c_double(rel_tol), ctypes.c_double(abs_tol)])
        return bool(eq.value)

    def isclose(self, other, *, rel_tol=1e-7, abs_tol=0.0, check_dtype=False):
        if check_dtype and self.dtype is not other.dtype:
            return False
        if self.shape != other.shape or self.nvals != other.nvals:
            return False
        return self._compare_on_device(other, float(rel_tol), float(abs_tol))

    # ---- the hot path ---------------------------------------------------------------------------------------
    def mxv(self, other, op=_semiring.plus_times):
        """``w << A.mxv(v, semiring)``  (reference core/matrix.py:2203-2262 -> C ``GrB_mxv``)."""
        return _mxv(self, other, op)

    def mxm(self, other, op=_semiring.plus_times):
        """``C << A.mxm(B, semiring)``  (reference core/matrix.py:2264-2331 -> C ``GrB_mxm``)."""
        return _mxm(self, other, op)

    def reduce_rowwise(self, op="plus"):
        return _reduce_vector(self, op, "reduce_rowwise", False)

    def reduce_columnwise(self, op="plus"):
        return _reduce_vector(self, op, "reduce_columnwise", True)

    def reduce_scalar(self, op="plus", *, allow_empty=True):
        """``s << A.reduce_scalar(monoid)`` (reference core/matrix.py:2712-2770): here the row reduction on the pull SpMV path
        followed by the vector reduction -- Matrix -> Vector -> Scalar, as the reference's aggregators do (core/operator/agg.py:
        304-330); ``agg.count`` / ``agg.exists`` are the number of entries / whether there is one."""
        return _reduce_scalar(self, op, allow_empty)

    def power(self, n, op=_semiring.plus_times):
        """``C << A.power(n, semiring)`` by repeated squaring (reference core/matrix.py:2840-2905)."""
        return _matrix_power(self, n, op)

    def __matmul__(self, other):
        return InfixMatMul(self, other)


class TransposedMatrix:
    """``A.T``: a view that only flips the descriptor's T0/T1 (reference core/matrix.py:3900-3960)."""

    _is_transposed = True
    ndim = 2

    def __init__(self, matrix):
        self._matrix = matrix
        self._nrows, self._ncols = matrix._ncols, matrix._nrows

    dtype = property(lambda self: self._matrix.dtype)
    nrows = property(lambda self: self._nrows)
    ncols = property(lambda self: self._ncols)
    shape = property(lambda self: (self._nrows, self._ncols))
    name = property(lambda self: f"{self._matrix.name}.T")

    @property
    def T(self):
        return self._matrix

    def mxv(self, other, op=_semiring.plus_times):
        return _mxv(self, other, op)

    def mxm(self, other, op=_semiring.plus_times):
        return _mxm(self, other, op)

    def reduce_rowwise(self, op="plus"):
        return _reduce_vector(self, op, "reduce_rowwise", False)

    def reduce_columnwise(self, op="plus"):
        return _reduce_vector(self, op, "reduce_columnwise", True)

    def reduce_scalar(self, op="plus", *, allow_empty=True):
        return _reduce_scalar(self, op, allow_empty)

    def power(self, n, op=_semiring.plus_times):
        return _matrix_power(self, n, op)

    def __matmul__(self, other):
        return InfixMatMul(self, other)

    def new(self, dtype=None, *, name=None):
        C = Matrix(dtype or self.dtype, self._nrows, self._ncols, name=name)
        call("GrB_transpose", [C, None, None, self._matrix, None])
        return C


def _reduce_vector(A, op, method_name, transpose):
    """``w << A.reduce_rowwise(monoid)`` / ``reduce_columnwise`` (reference core/matrix.py:2636-2710 -> ``GrB_Matrix_reduce_Monoid``;
    column-wise = row-wise over the transpose, descriptor T0).  An aggregator is either its monoid or, for ``count`` /
    ``exists``, the mat-vec ``semiring(A @ init)`` with a dense iso vector (reference core/operator/agg.py:264-283)."""
    from .operators import Aggregator
    from .operators import monoid as _monoid_ns

    at = A._is_transposed != transpose
    base = A._matrix
    if isinstance(op, Aggregator):
        if op.semiring is not None:
            view = base.T if at else base
            init = Vector(op.any_dtype, view._ncols, name="init")
            init[:] = 1
            expr = _mxv(view, init, op.semiring[op.any_dtype])
            expr.method_name = method_name
            return expr
        op = op.monoid
    op = get_typed_op(op, A.dtype, kind="binary")
    if op.opclass == "BinaryOp":
        if not hasattr(_monoid_ns, op.name):
            raise TypeError(f"Expected type: Monoid; got BinaryOp `{op!r}`")
        op = getattr(_monoid_ns, op.name)[A.dtype]
    out_size = base._ncols if at else base._nrows
    return Expression(method_name, "GrB_Matrix_reduce_Monoid", [base], op=op, output_type=Vector, shape=(out_size,), at=at)


def _reduce_scalar(A, op, allow_empty):
    from .base import ScalarExpression
    from .operators import Aggregator

    base = A._matrix
    if isinstance(op, Aggregator) and op.semiring is not None:
        count = op.name == "count"

        def compute_agg():
            nv = base.nvals
            if nv == 0 and (allow_empty or not count):
                return None if allow_empty else 0
            return int(nv) if count else 1

        return ScalarExpression(compute_agg, op.any_dtype)
    mon = op.monoid if isinstance(op, Aggregator) else op
    rows = _reduce_vector(base, mon, "reduce_scalar", False)

    def compute():
        return rows.new().reduce(mon, allow_empty=allow_empty).value

    return ScalarExpression(compute, rows.dtype)


def _power(updater, A, n, op):
    """Repeated squaring over the mxm path (reference core/matrix.py:99-155): A^2, A^4, ... are formed in ``square`` and
    multiplied into ``result`` for each set bit of ``n``; the last product goes through the caller's updater so that its
    mask / accum / replace apply to the final multiplication only."""
    opts = updater.opts
    if n == 0:
        ident = op.parent.binaryop.monoid.identity(A.dtype)
        idx = np.arange(A._nrows, dtype=np.uint64)
        D = Matrix.from_coo(idx, idx, np.full(A._nrows, ident), dtype=A.dtype, nrows=A._nrows, ncols=A._ncols, name="M_diag")
        updater << PowerCopy(D)
        return
    if n == 1:
        updater << PowerCopy(A)
        return
    result = square = None  # square = A^(2^k) once k >= 1; result = product of the squares of the bits seen so far
    owned = False           # result is a matrix of ours (not A itself)
    k = 0
    while True:
        n, bit = divmod(n, 2)
        if k:
            src = A if square is None else square
            squaring = _mxm(src, src, op)
            if n == 0 and result is None:  # n was a power of two: the last squaring is the answer
                updater << squaring
                return
            if square is None:
                square = squaring.new(name="Squares", **opts)
            else:
                square(**opts) << squaring
        k += 1
        if not bit:
            continue
        cur = A if square is None else square
        if result is None:
            result, owned = (A, False) if square is None else (square.dup(name="Power"), True)
        elif n == 0:
            updater << _mxm(result, cur, op)
            return
        elif not owned:
            result, owned = _mxm(result, cur, op).new(name="Power", **opts), True
        else:
            result(**opts) << _mxm(result, cur, op)


class PowerCopy:
    """``C << A`` for the degenerate powers (n = 0, 1): a copy through ``GrB_transpose`` with descriptor T0."""

    def __init__(self, matrix):
        self.matrix = matrix


class PowerExpression:
    """``A.power(n, op)`` (reference core/matrix.py:2840-2905): evaluated by ``_power`` when assigned."""

    def __init__(self, A, n, op):
        self.A, self.n, self.op = A, n, op
        self.shape = (A._nrows, A._ncols)
        self.dtype = op.return_type

    def new(self, dtype=None, *, mask=None, name=None, **opts):
        out = Matrix(dtype or self.dtype, *self.shape, name=name)
        if mask is None:
            out(**opts) << self
        else:
            out(mask=mask, **opts) << self
        return out


def _matrix_power(A, n, op):
    from numbers import Integral

    if A._nrows != A._ncols:
        raise DimensionMismatch(f"power only works for square Matrix; shape is {A.shape}")
    if isinstance(n, bool) or not (isinstance(n, (Integral, np.integer)) or (isinstance(n, float) and n.is_integer())):
        raise TypeError(f"n must be a nonnegative integer; got bad type: {type(n)}")
    N = int(n)
    if N < 0:
        raise ValueError(f"n must be a nonnegative integer; got: {N}")
    op = get_typed_op(op, A.dtype, kind="semiring")
    if N == 0 and op.parent.binaryop.monoid is None:
        raise ValueError(
            f"Binary operator of {op} semiring does not have a monoid with an identity. "
            "When n=0, the result is a diagonal matrix with values equal to the "
            "identity of the binaryop, so the binaryop must be associated with a monoid."
        )
    return PowerExpression(A, N, op)


def _mxv(A, v, op):
    if not isinstance(v, Vector):
        raise TypeError(f"Expected type: Vector; got {type(v).__name__}")
    op = get_typed_op(op, A.dtype, v.dtype, kind="semiring")
    expr = Expression("mxv", "GrB_mxv", [A._matrix, v], op=op, output_type=Vector, shape=(A._nrows,),
                      at=A._is_transposed)
    if A._ncols != v._size:
        expr._force_library_error()
    return expr


def _mxm(A, B, op):
    if not isinstance(B, (Matrix, TransposedMatrix)):
        raise TypeError(f"Expected type: Matrix; got {type(B).__name__}")
    op = get_typed_op(op, A.dtype, B.dtype, kind="semiring")
    expr = Expression("mxm", "GrB_mxm", [A._matrix, B._matrix], op=op, output_type=Matrix,
                      shape=(A._nrows, B._ncols), at=A._is_transposed, bt=B._is_transposed)
    if A._ncols != B._nrows:
        expr._force_library_error()
    return expr
