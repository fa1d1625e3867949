"""Vector: the host-side mirror of graphblas/core/vector.py for the path -- constructor :159-170,
``__del__`` :179-191, ``build`` :522-568, ``from_coo`` (:~700), ``from_dense`` :849-901,
``to_dense`` :903-955, ``to_coo`` (:~470), ``isequal`` :340-379 and ``vxm`` :1309-1378."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .base import AllIndexAssigner, BaseType, Expression, InfixMatMul, Scalar, ScalarExpression, call, call_on
from .dtypes import lookup_dtype
from .operators import Monoid, get_typed_op, monoid as _monoid, semiring as _semiring

_name_counter = iter(range(1 << 62))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


_EMPTY_INDEX = np.zeros(1, np.uint64)


def _iptr(idx):
    """pointer to an index list (never NULL: NULL is not a legal list, an empty one is)"""
    return (idx if idx.size else _EMPTY_INDEX).ctypes.data_as(ctypes.c_void_p)


class Vector(BaseType):
    _grb_kind = "Vector"
    ndim = 1

    def __init__(self, dtype=float, size=0, *, name=None):
        self.dtype = lookup_dtype(dtype)
        self._size = int(size)
        self.name = name or f"v_{next(_name_counter)}"
        self._handle = ctypes.c_void_p()
        _lib.load()
        call_on(None, "GrB_Vector_new", [ctypes.byref(self._handle), self.dtype._carg, self._size])

    @property
    def _carg(self):
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and _lib is not None and _lib.lib is not None:
            try:
                _lib.lib.GrB_Vector_free(ctypes.byref(h))
            except Exception:  # interpreter shutdown
                pass

    # ---- properties ---------------------------------------------------------------------------------
    @property
    def size(self):
        return self._size

    @property
    def shape(self):
        return (self._size,)

    @property
    def nvals(self):
        n = ctypes.c_uint64()
        call_on(self, "GrB_Vector_nvals", [ctypes.byref(n), self._handle])
        return int(n.value)

    def __repr__(self):
        return f"Vector<{self.dtype}, size={self._size}, name={self.name}>"

    # ---- ingress / egress ---------------------------------------------------------------------------------
    def build(self, indices, values, *, dup_op=None, clear=False, size=None):
        """reference core/vector.py:522-568: indices as uint64; duplicates need ``dup_op``."""
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        values = np.asarray(values)
        if values.ndim == 0:
            values = np.broadcast_to(values, indices.shape)
        values = np.ascontiguousarray(values.astype(self.dtype.np_type, copy=False))
        n = indices.size
        if values.size != n:
            raise ValueError(f"`indices` and `values` lengths must match: {n} != {values.size}")
        if clear:
            self.clear()
        if n == 0:
            return
        dup_orig = dup_op
        if dup_op is None:
            dup_op = "plus"  # duplicates are detected below, exactly like the reference (:548-566)
        dup_t = get_typed_op(dup_op, self.dtype, kind="binary")
        if dup_t.opclass == "Monoid":
            dup_t = dup_t.binaryop
        call(f"GrB_Vector_build_{self.dtype.name}", [self, _ptr(indices), _ptr(values), n, dup_t])
        if dup_orig is None and self.nvals < n:
            self.clear()
            raise ValueError("Duplicate indices found, must provide `dup_op` BinaryOp")

    @classmethod
    def from_coo(cls, indices, values=1.0, dtype=None, *, size=None, dup_op=None, name=None):
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        values = np.asarray(values)
        if dtype is None:
            dtype = lookup_dtype(values.dtype) if values.dtype.kind in "biuf" else lookup_dtype(float)
        if size is None:
            if indices.size == 0:
                raise ValueError("No indices provided. Unable to infer size.")
            size = int(indices.max()) + 1
        w = cls(dtype, size, name=name)
        w.build(indices, values, dup_op=dup_op)
        return w

    @classmethod
    def from_dense(cls, values, missing_value=None, *, dtype=None, name=None):
        """reference core/vector.py:849-901 (vanilla path: from_coo(arange(n), values))."""
        values = np.asarray(values)
        if dtype is None:
            dtype = lookup_dtype(values.dtype)
        idx = np.arange(values.size, dtype=np.uint64)
        if missing_value is not None:
            keep = values != missing_value
            idx, values = idx[keep], values[keep]
        w = cls(dtype, np.asarray(values).size if missing_value is None else int(keep.size), name=name)
        w.build(idx, values)
        return w

    def to_coo(self, dtype=None, *, indices=True, values=True, sort=True):
        n = self.nvals
        I = np.empty(n, np.uint64) if indices else None
        X = np.empty(n, self.dtype.np_type) if values else None
        cnt = ctypes.c_uint64(n)
        call_on(self, f"GrB_Vector_extractTuples_{self.dtype.name}", [_ptr(I) if n else None, _ptr(X) if n else None,
                                                                      ctypes.byref(cnt), self._handle])
        if X is not None and dtype is not None:
            X = X.astype(lookup_dtype(dtype).np_type)
        return I, X

    def to_dense(self, fill_value=None, dtype=None):
        I, X = self.to_coo()
        if fill_value is None and I.size < self._size:
            raise TypeError("fill_value must be given for vectors with missing elements")
        out = np.full(self._size, fill_value if fill_value is not None else 0,
                      (lookup_dtype(dtype) if dtype is not None else self.dtype).np_type)
        out[I.astype(np.int64)] = X
        return out

    def dup(self, dtype=None, *, clear=False, mask=None, name=None):
        """A copy; ``dtype``: with the values cast (on the device); ``clear``: same size and dtype, no entries; ``mask``: only the
        entries the mask lets through (reference core/vector.py:392-420; tests/test_vector.py:58-92)."""
        dt = self.dtype if dtype is None else lookup_dtype(dtype)
        if clear:
            return Vector(dt, self._size, name=name)
        if mask is not None:
            w = Vector(dt, self._size, name=name)
            w(mask=mask) << self
            return w
        w = Vector.__new__(Vector)
        w.dtype, w._size, w.name = dt, self._size, name or f"v_{next(_name_counter)}"
        w._handle = ctypes.c_void_p()
        if dt is self.dtype:
            call_on(self, "GrB_Vector_dup", [ctypes.byref(w._handle), self._handle])
        else:  # the typecast copy is made on the device
            call_on(self, "GrX_Vector_dup_as", [ctypes.byref(w._handle), dt._carg, self._handle])
        return w

    def clear(self):
        call("GrB_Vector_clear", [self])

    def resize(self, size):
        """In place: growing adds empty positions, shrinking drops the entries at or above the new size
        (reference core/vector.py:455-463)."""
        size = int(size)
        if size < 0:
            raise ValueError("size must be non-negative")
        call("GrB_Vector_resize", [self, size])
        self._size = size

    def isequal(self, other, *, check_dtype=False):
        """Same size, structure and values, decided on the device exactly as the reference does (core/vector.py:340-379):
        equal ``nvals``, ``ewise_mult(eq)`` keeps every entry, and the ``land`` of the comparisons is true."""
        if not isinstance(other, Vector):
            raise TypeError(f"Expected type: Vector; got {type(other).__name__}")
        if check_dtype and self.dtype is not other.dtype:
            return False
        if self._size != other._size:
            return False
        nv = self.nvals
        if nv != other.nvals:
            return False
        if nv == 0:
            return True
        from .operators import binary as _binary

        matches = self.ewise_mult(other, _binary.eq).new()
        if matches.nvals != nv:
            return False
        return bool(matches.reduce(_monoid.land, allow_empty=False).new().value)

    def isclose(self, other, *, rel_tol=1e-7, abs_tol=0.0, check_dtype=False):
        if check_dtype and self.dtype is not other.dtype:
            return False
        if self._size != other._size or self.nvals != other.nvals:
            return False
        (i1, x1), (i2, x2) = self.to_coo(), other.to_coo()
        return bool(np.array_equal(i1, i2) and np.allclose(x1.astype(float), x2.astype(float), rtol=rel_tol, atol=abs_tol))

    # ---- element access, scalar assign, reduce: what the BFS / SSSP loops need around the hot path ----------------
    def _index(self, key):
        if isinstance(key, (bool, np.bool_)) or not isinstance(key, (int, np.integer)):
            raise TypeError(f"Invalid type for index: {type(key).__name__}; only integers and [:] are supported")
        i = int(key)
        if i < 0:
            i += self._size
        if not 0 <= i < self._size:
            raise IndexError(f"Index out of range: index={key}, size={self._size}")
        return i

    def _index_list(self, key):
        """Slices, lists and integer arrays as an explicit uint64 index array (reference core/expr.py IndexerResolver)."""
        if isinstance(key, slice):
            return np.arange(*key.indices(self._size), dtype=np.uint64)
        a = np.asarray(key)
        if a.ndim != 1 or a.dtype.kind not in "iu":
            raise TypeError(f"Invalid type for index: {type(key).__name__}; integers, slices and integer lists are supported")
        a = a.astype(np.int64)
        a = np.where(a < 0, a + self._size, a)
        if a.size and (a.min() < 0 or a.max() >= self._size):
            raise IndexError(f"Index out of range: size={self._size}")
        return np.ascontiguousarray(a, dtype=np.uint64)

    @staticmethod
    def _is_scalar_index(key):
        return isinstance(key, (int, np.integer)) and not isinstance(key, (bool, np.bool_))

    def __getitem__(self, key):
        """``v[i]`` -> scalar expression (reference core/vector.py:1840-1866); ``v[:]`` -> assign-to-everything target; ``v[I]``
        with a slice or an index list -> extract expression / assign target (reference core/vector.py:1906-2035)."""
        if isinstance(key, slice) and key == slice(None):
            return AllIndexAssigner(self)
        if self._is_scalar_index(key) or isinstance(key, (bool, np.bool_)):
            return _ElementExpr(self, self._index(key))
        return IndexedVector(self, self._index_list(key))

    def __setitem__(self, key, value):
        if isinstance(key, slice) and key == slice(None):
            AllIndexAssigner(self) << value
        elif self._is_scalar_index(key):
            self._element_assigner(key) << value
        else:
            IndexedVector(self, self._index_list(key)) << value

    def _indexed_assigner(self, key, *, mask=None, accum=None, replace=False, opts=None):
        return IndexedVector(self, self._index_list(key), mask=mask, accum=accum, replace=replace, opts=opts)

    def _assign_indexed(self, value, idx, mask=None, accum=None, replace=False, *, opts):
        """``w(mask, accum, replace)[I] << u`` -> GrB_Vector_assign; ``... << scalar`` -> GrB_Vector_assign_<T> with the index list
        (reference core/vector.py:1979-2035)."""
        from .base import InfixMatMul, _check_mask
        from .descriptor import lookup as descriptor_lookup

        if mask is None:
            complement = structure = False
        else:
            mask = _check_mask(mask, self)
            complement, structure = mask.complement, mask.structure
        if accum is not None:
            accum = get_typed_op(accum, self.dtype, kind="binary")
            if hasattr(accum, "opclass") and accum.opclass == "Monoid":
                accum = accum.binaryop
        desc = descriptor_lookup(mask_complement=complement, mask_structure=structure, output_replace=replace, **(opts or {}))
        if isinstance(value, (Expression, InfixMatMul)):
            value = value.new()
        if isinstance(value, Vector):
            call("GrB_Vector_assign", [self, mask, accum, value, _iptr(idx), ctypes.c_uint64(idx.size), desc])
            return
        x = self._scalar_carg(value)
        ctype = np.ctypeslib.as_ctypes_type(self.dtype.np_type)
        call(f"GrB_Vector_assign_{self.dtype.name}", [self, mask, accum, ctype(x.item()), _iptr(idx), ctypes.c_uint64(idx.size), desc])

    def __delitem__(self, key):
        """reference core/vector.py:1916-1930"""
        call("GrB_Vector_removeElement", [self, self._index(key)])

    def _element_assigner(self, key, accum=None):
        return _ElementAssigner(self, self._index(key), accum)

    def _scalar_carg(self, value):
        if isinstance(value, Scalar):
            if value.is_empty:
                raise ValueError("cannot assign an empty Scalar")
            value = value.value
        if not isinstance(value, (bool, int, float, np.generic)):
            raise TypeError(f"Bad type for arg: {type(value).__name__}")
        return self.dtype.np_type.type(value)

    def _assign_scalar_all(self, value, mask=None, accum=None, replace=False, *, opts):
        """``w(mask, accum, replace)[:] << scalar`` -> ``GrB_Vector_assign_<T>`` over ``GrB_ALL``
        (reference core/vector.py:1979-2035)."""
        from .base import _check_mask
        from .descriptor import lookup as descriptor_lookup

        x = self._scalar_carg(value)
        if mask is None:
            complement = structure = False
        else:
            mask = _check_mask(mask, self)
            complement, structure = mask.complement, mask.structure
        desc = descriptor_lookup(mask_complement=complement, mask_structure=structure, output_replace=replace, **opts)
        ctype = np.ctypeslib.as_ctypes_type(self.dtype.np_type)
        call(f"GrB_Vector_assign_{self.dtype.name}", [self, mask, accum, ctype(x.item()), _lib.all_indices(), self._size, desc])

    def reduce(self, op=_monoid.plus, *, allow_empty=True):
        """``s << v.reduce(monoid)`` (reference core/vector.py:1635-1684).  ``allow_empty=False``: an empty vector gives the
        monoid identity instead of an empty scalar.  Aggregators: the monoid ones reduce with their monoid; ``agg.count`` and
        ``agg.exists`` are the plus_pair / any_pair products with a dense iso operand (reference core/operator/agg.py:284-303),
        i.e. the number of entries and whether there is one."""
        from .operators import Aggregator

        if isinstance(op, Aggregator):
            if op.semiring is not None:
                count = op.name == "count"

                def compute_agg():
                    nv = self.nvals
                    if nv == 0 and (allow_empty or not count):
                        return None if allow_empty else 0
                    return int(nv) if count else 1

                return ScalarExpression(compute_agg, op.any_dtype)
            op = op.monoid
        op = get_typed_op(op, self.dtype, kind="binary")
        if op.opclass == "BinaryOp":
            if not hasattr(_monoid, op.name):
                raise TypeError(f"Expected type: Monoid; got BinaryOp `{op!r}`")
            op = getattr(_monoid, op.name)[self.dtype]
        rtype = op.return_type

        def compute():
            if allow_empty and self.nvals == 0:
                return None
            out = np.ctypeslib.as_ctypes_type(rtype.np_type)()
            call_on(self, f"GrB_Vector_reduce_{rtype.name}", [ctypes.byref(out), None, op._carg, self._handle, None])
            return rtype.np_type.type(out.value).item()

        return ScalarExpression(compute, rtype)

    def _ewise(self, other, op, method_name, cfunc):
        if not isinstance(other, Vector):
            raise TypeError(f"Expected type: Vector; got {type(other).__name__}")
        op = get_typed_op(op, self.dtype, other.dtype, kind="binary")
        expr = Expression(method_name, f"{cfunc}_{op.opclass}", [self, other], op=op, output_type=Vector, shape=(self._size,))
        if self._size != other._size:
            expr._force_library_error()
        return expr

    def ewise_add(self, other, op=_monoid.plus):
        """Union of the patterns; ``op`` where both have an entry (reference core/vector.py:960-1060)."""
        return self._ewise(other, op, "ewise_add", "GrB_Vector_eWiseAdd")

    def ewise_mult(self, other, op=None):
        """Intersection of the patterns (reference core/vector.py:1062-1150)."""
        from .operators import binary as _binary

        return self._ewise(other, op if op is not None else _binary.times, "ewise_mult", "GrB_Vector_eWiseMult")

    # ---- the hot path ---------------------------------------------------------------------------------------
    def vxm(self, other, op=_semiring.plus_times):
        """``w << u.vxm(A, semiring)``  (reference core/vector.py:1309-1378 -> C ``GrB_vxm``)."""
        from .matrix import Matrix, TransposedMatrix

        if not isinstance(other, (Matrix, TransposedMatrix)):
            raise TypeError(f"Expected type: Matrix; got {type(other).__name__}")
        op = get_typed_op(op, self.dtype, other.dtype, kind="semiring")
        expr = Expression("vxm", "GrB_vxm", [self, other._matrix], op=op, output_type=Vector, shape=(other._ncols,),
                          bt=other._is_transposed)
        if self._size != other._nrows:
            expr._force_library_error()
        return expr

    def __matmul__(self, other):
        return InfixMatMul(self, other)


class IndexedVector(Expression):
    """``v[I]`` with a slice or an index list: on the right of ``<<`` (or with ``.new()``) it extracts -- GrB_Vector_extract --,
    on the left it is an assign target -- GrB_Vector_assign / GrB_Vector_assign_<T> (the reference's AmbiguousAssignOrExtract,
    core/expr.py:240-400)."""

    def __init__(self, parent, idx, *, mask=None, accum=None, replace=False, opts=None):
        self.parent, self.idx = parent, idx
        self.mask, self.accum, self.replace, self.opts = mask, accum, replace, opts or {}
        super().__init__("extract", "GrB_Vector_extract", [parent, _iptr(idx), ctypes.c_uint64(idx.size)], op=None,
                         output_type=Vector, shape=(int(idx.size),))

    @property
    def dtype(self):
        return self.parent.dtype

    def __call__(self, *args, **kwargs):
        up = self.parent(*args, **kwargs)
        return IndexedVector(self.parent, self.idx, mask=up.mask, accum=up.accum, replace=up.replace, opts=up.opts)

    def __lshift__(self, value):
        self.parent._assign_indexed(value, self.idx, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)

    update = __lshift__


class _ElementExpr(ScalarExpression):
    """``v[i]``: read with ``.new()`` / ``.value``; ``v[i] << x`` assigns (reference core/vector.py:1840-1910)."""

    def __init__(self, parent, index):
        self.parent, self.index = parent, index
        super().__init__(self._read, parent.dtype)

    def _read(self):
        v = self.parent
        out = np.ctypeslib.as_ctypes_type(v.dtype.np_type)()
        info = _lib.lib.__getattr__(f"GrB_Vector_extractElement_{v.dtype.name}")(ctypes.byref(out), v._handle, self.index)
        if info == 1:  # GrB_NO_VALUE
            return None
        from .exceptions import check_status

        check_status(info, v)
        return v.dtype.np_type.type(out.value).item()

    def __lshift__(self, value):
        _ElementAssigner(self.parent, self.index, None) << value

    def __call__(self, *, accum=None):
        return _ElementAssigner(self.parent, self.index, accum)


class _ElementAssigner:
    def __init__(self, parent, index, accum):
        self.parent, self.index, self.accum = parent, index, accum

    def __lshift__(self, value):
        v = self.parent
        x = v._scalar_carg(value)
        if self.accum is not None:
            old = _ElementExpr(v, self.index).value
            if old is not None:
                from .base import _apply_host_binary

                op = get_typed_op(self.accum, v.dtype, kind="binary")
                x = v.dtype.np_type.type(_apply_host_binary(op.name, v.dtype.np_type.type(old), x))
        ctype = np.ctypeslib.as_ctypes_type(v.dtype.np_type)
        call(f"GrB_Vector_setElement_{v.dtype.name}", [v, ctype(x.item()), self.index])

    update = __lshift__
