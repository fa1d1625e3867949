"""Dispatcher, delayed expressions, the ``w(mask, accum, replace) << expr`` updater and mask wrappers.

Mirrors, for the mxm/mxv/vxm path, the reference's
  * ``call(cfunc_name, args)``                      graphblas/core/base.py:23-54
  * ``BaseType.__call__`` (output-argument parsing) graphblas/core/base.py:192-263
  * ``BaseType._update`` (descriptor pick, 7-argument marshalling) core/base.py:338-514 (:468-475, :496-503)
  * ``BaseExpression.new``                           core/base.py:583-616
  * ``Updater``                                      graphblas/core/expr.py:404-481
  * Mask family                                      graphblas/core/mask.py:9-35,133-200
"""
from __future__ import annotations

from . import _lib
from .descriptor import lookup as descriptor_lookup
from .dtypes import BOOL, lookup_dtype
from .exceptions import check_status
from .operators import Monoid, TypedOp, get_typed_op, semiring

_recorder = None  # optional list collecting C-call strings (reference core/recorder.py)


def record_calls(target: list | None):
    global _recorder
    _recorder = target


def _cname(x):
    if x is None:
        return "NULL"
    if isinstance(x, TypedOp):
        return x.gb_name
    return getattr(x, "name", None) or repr(x)


def call(cfunc_name, args):
    """reference core/base.py:23-54: NULL for absent args, symbol lookup, status check on args[0]."""
    call_args = [getattr(x, "_carg", x) if x is not None else None for x in args]
    cfunc = getattr(_lib.lib, cfunc_name)
    err_code = cfunc(*call_args)
    if _recorder is not None:
        _recorder.append(f"{cfunc_name}({', '.join(_cname(x) for x in args)});")
    return check_status(err_code, args)


def call_on(obj, cfunc_name, cargs):
    """Direct call whose error text belongs to ``obj`` (calls whose first C argument is an output
    pointer: GrB_*_new, _nvals, _extractTuples ...; reference core/matrix.py:225,493)."""
    err_code = getattr(_lib.lib, cfunc_name)(*cargs)
    if _recorder is not None:
        _recorder.append(f"{cfunc_name}(...);")
    return check_status(err_code, obj)


# ---- masks --------------------------------------------------------------------------------------------
class Mask:
    complement = False
    structure = False
    value = False

    def __init__(self, parent):
        self.parent = parent

    @property
    def _carg(self):
        return self.parent._carg

    def __repr__(self):
        return f"{'~' if self.complement else ''}{self.parent.name}.{'S' if self.structure else 'V'}"


class StructuralMask(Mask):
    structure = True

    def __invert__(self):
        return ComplementedStructuralMask(self.parent)


class ValueMask(Mask):
    value = True

    def __invert__(self):
        return ComplementedValueMask(self.parent)


class ComplementedStructuralMask(Mask):
    complement = True
    structure = True

    def __invert__(self):
        return StructuralMask(self.parent)


class ComplementedValueMask(Mask):
    complement = True
    value = True

    def __invert__(self):
        return ValueMask(self.parent)


def _check_mask(mask, output=None):
    """reference core/base.py:169-183: bare BOOL collections become value masks; anything else must
    spell .S or .V."""
    if isinstance(mask, Mask):
        if output is not None and type(mask.parent) is not type(output):
            raise TypeError(f"Mask object must be type {type(output).__name__}")
        return mask
    if isinstance(mask, BaseType):
        if mask.dtype is not BOOL:
            raise TypeError(f"Mask must be from a {type(mask).__name__} with BOOL dtype, or use .S or .V to create a mask")
        return ValueMask(mask)
    raise TypeError(f"Invalid mask: {type(mask)}")


# ---- collections' common base ----------------------------------------------------------------------------
def _output_param_kind(item):
    """What a positional output parameter is: a mask (a Mask view or a collection used as one), the ``replace`` marker, or an
    accumulator (an operator, or its name); anything else is rejected with the reference's message."""
    if item is _replace_singleton:
        return "replace"
    if isinstance(item, (Mask, BaseType)):
        return "mask"
    if isinstance(item, (str, TypedOp)) or hasattr(item, "opclass"):
        return "accum"
    raise TypeError(f"Invalid item found in output params: {type(item)}")


class BaseType:
    _grb_kind = None
    _is_transposed = False

    @property
    def S(self):
        return StructuralMask(self)

    @property
    def V(self):
        return ValueMask(self)

    def __call__(self, *args, mask=None, accum=None, replace=False, **opts):
        """The output parameters of an update, ``w(mask, accum, replace) << expr``: positional and keyword spellings in any order
        (behaviour pinned by the reference's tests, tests/test_vector.py:350-368, tests/test_matrix.py:373-374; counterpart of
        core/base.py:192-263)."""
        given = {"mask": [mask] if mask is not None else [], "accum": [accum] if accum is not None else []}
        for item in args:
            kind = _output_param_kind(item)
            if kind == "replace":
                replace = True
            else:
                given[kind].append(item)
        for name, found in given.items():
            if len(found) > 1:
                raise TypeError(f"Got multiple values for argument '{name}'")
        mask = _check_mask(given["mask"][0]) if given["mask"] else None
        if mask is None and replace:
            raise TypeError("'replace' argument may only be True if a mask is provided")
        accum = None
        if given["accum"]:
            accum = get_typed_op(given["accum"][0], self.dtype, kind="binary")
            if accum.opclass == "Monoid":  # (a monoid given as the accumulator means its binary operator)
                accum = accum.binaryop
        return Updater(self, mask=mask, accum=accum, replace=replace, opts=opts)

    def __lshift__(self, expr):
        return self._update(expr, opts={})

    def update(self, expr):
        return self._update(expr, opts={})

    def _update(self, expr, mask=None, accum=None, replace=False, *, opts):
        """reference core/base.py:338-514."""
        if isinstance(expr, InfixMatMul):
            expr = expr.with_op(semiring.plus_times)
        if isinstance(expr, Scalar) or _is_python_scalar(expr):
            # ``w(mask) << 5``: scalar assign over every index (reference core/base.py:352-372 -> Updater[...] << scalar)
            return self._assign_scalar_all(expr, mask=mask, accum=accum, replace=replace, opts=opts)
        if self._grb_kind == "Vector" and type(expr) is type(self):
            # ``w(mask, accum, replace) << u``: a copy under the write rule (reference core/base.py:373-399 applies the identity
            # operator).  Here: first(u, u) over the union of u with itself -- one element-wise pass on the device.
            from .operators import binary as _binary

            expr = expr.ewise_add(expr, _binary.first)
        if self._grb_kind == "Matrix":
            from .matrix import Matrix, PowerCopy, PowerExpression, TransposedMatrix, _power

            if type(expr) is Matrix:
                # ``C(mask, accum, replace) << A``: GrB_transpose with the input transposed back by the descriptor (T0)
                complement = structure = False
                if mask is not None:
                    mask = _check_mask(mask, self)
                    complement, structure = mask.complement, mask.structure
                desc = descriptor_lookup(transpose_first=True, mask_complement=complement, mask_structure=structure,
                                         output_replace=replace, **opts)
                return call("GrB_transpose", [self, mask, accum, expr, desc])

            if isinstance(expr, TransposedMatrix):
                # ``C(mask, accum, replace) << A.T`` -> GrB_transpose (reference core/base.py:401-411)
                complement = structure = False
                if mask is not None:
                    mask = _check_mask(mask, self)
                    complement, structure = mask.complement, mask.structure
                desc = descriptor_lookup(mask_complement=complement, mask_structure=structure, output_replace=replace, **opts)
                return call("GrB_transpose", [self, mask, accum, expr._matrix, desc])

            if isinstance(expr, PowerExpression):
                # evaluated by repeated squaring; (mask, accum, replace) apply to the last product (reference core/matrix.py:99-155)
                return _power(Updater(self, mask=mask, accum=accum, replace=replace, opts=opts), expr.A, expr.n, expr.op)
            if isinstance(expr, PowerCopy):
                complement = structure = False
                if mask is not None:
                    mask = _check_mask(mask, self)
                    complement, structure = mask.complement, mask.structure
                src = expr.matrix  # (A.T.power(1) is a real transpose: no T0)
                desc = descriptor_lookup(transpose_first=not src._is_transposed, mask_complement=complement,
                                         mask_structure=structure, output_replace=replace, **opts)
                return call("GrB_transpose", [self, mask, accum, src._matrix, desc])
        if not isinstance(expr, Expression):
            raise TypeError(f"Assignment value must be a valid expression; got {type(expr).__name__}")
        if type(self) is not expr.output_type:
            raise TypeError(f"Bad type when updating: output is {type(self).__name__}, expression yields {expr.output_type.__name__}")
        if mask is None:
            complement = structure = False
        else:
            mask = _check_mask(mask, self)
            complement, structure = mask.complement, mask.structure
        desc = descriptor_lookup(transpose_first=expr.at, transpose_second=expr.bt, mask_complement=complement,
                                 mask_structure=structure, output_replace=replace, **opts)
        # (extract has no operator argument: GrB_Vector_extract(w, mask, accum, u, I, ni, desc))
        args = [self, mask, accum, *([expr.op] if expr.op is not None else []), *expr.args, desc]
        call(expr.cfunc_name, args)

    def _assign_scalar_all(self, value, mask=None, accum=None, replace=False, *, opts):
        raise TypeError(f"Scalar assignment is not supported for {type(self).__name__}")

    def wait(self, how="materialize"):
        call(f"GrB_{self._grb_kind}_wait", [self, 1 if how == "materialize" else 0])
        return self


class Updater:
    """reference graphblas/core/expr.py:404-481."""

    def __init__(self, parent, *, mask=None, accum=None, replace=False, opts=None):
        self.parent, self.mask, self.accum, self.replace = parent, mask, accum, replace
        self.opts = opts or {}

    def __lshift__(self, expr):
        self.parent._update(expr, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)

    def update(self, expr):
        self.parent._update(expr, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)

    def __getitem__(self, key):
        if isinstance(key, slice) and key == slice(None):
            return AllIndexAssigner(self.parent, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)
        if hasattr(self.parent, "_indexed_assigner") and not self.parent._is_scalar_index(key) and not isinstance(key, bool):
            # ``w(mask, accum, replace)[I] << value`` with an index list / slice (reference core/expr.py:404-481, 484-560)
            return self.parent._indexed_assigner(key, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)
        if self.mask is not None:
            raise TypeError("Single element assign does not accept a submask")
        return self.parent._element_assigner(key, accum=self.accum)

    def __setitem__(self, key, value):
        self[key] << value


class AllIndexAssigner:
    """``w[:]`` / ``w(mask)[:]`` on the left of ``<<``: assign a scalar (or the collection's own kind of expression) to every
    index under the updater's (mask, accum, replace) -- reference core/expr.py:484-560 restricted to ``[:]``."""

    def __init__(self, parent, *, mask=None, accum=None, replace=False, opts=None):
        self.parent, self.mask, self.accum, self.replace, self.opts = parent, mask, accum, replace, opts or {}

    def __call__(self, *args, **kwargs):
        if self.mask is not None or self.accum is not None or self.replace:
            raise TypeError("mask, accum and replace were already given")
        up = self.parent(*args, **kwargs)
        return AllIndexAssigner(self.parent, mask=up.mask, accum=up.accum, replace=up.replace, opts=up.opts)

    def __lshift__(self, value):
        if not (isinstance(value, (Scalar, Expression, InfixMatMul)) or _is_python_scalar(value)):
            raise TypeError(f"Bad type for arg `value` in {type(self.parent).__name__}[:] = ...: {type(value).__name__}")
        self.parent._update(value, mask=self.mask, accum=self.accum, replace=self.replace, opts=self.opts)

    update = __lshift__


def _is_python_scalar(x):
    import numpy as np

    return isinstance(x, (bool, int, float, np.generic))


class Scalar:
    """A host-side scalar that may be empty (reference core/scalar.py, the parts the BFS / SSSP loops use: ``Scalar(dtype)``,
    ``from_value``, ``s << v.reduce(...)``, ``s(accum=op) << ...``, ``.value``, ``is_empty``, truthiness, comparisons)."""

    def __init__(self, dtype=float, *, name=None):
        self.dtype = lookup_dtype(dtype)
        self._value = None
        self.name = name or "s"

    @classmethod
    def from_value(cls, value, dtype=None, *, name=None):
        import numpy as np

        if dtype is None:
            dtype = lookup_dtype(np.asarray(value).dtype) if not isinstance(value, (bool, int, float)) else \
                lookup_dtype(bool if isinstance(value, bool) else (np.int64 if isinstance(value, int) else np.float64))
        s = cls(dtype, name=name)
        s._value = s.dtype.np_type.type(value) if value is not None else None
        return s

    @property
    def value(self):
        return None if self._value is None else self._value.item()

    @value.setter
    def value(self, v):
        self._value = None if v is None else self.dtype.np_type.type(v)

    @property
    def is_empty(self):
        return self._value is None

    @property
    def nvals(self):
        return 0 if self._value is None else 1

    def clear(self):
        self._value = None

    def dup(self, dtype=None, *, name=None):
        out = Scalar(dtype if dtype is not None else self.dtype, name=name)
        out.value = self.value
        return out

    def new(self, dtype=None, *, name=None):
        return self.dup(dtype, name=name)

    def __bool__(self):
        return bool(self._value) if self._value is not None else False

    def __eq__(self, other):
        other = other.value if isinstance(other, (Scalar, ScalarExpression)) else other
        return self.value == other

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"Scalar<{self.dtype}, value={self.value}>"

    def isequal(self, other, *, check_dtype=False):
        if isinstance(other, Scalar) and check_dtype and other.dtype is not self.dtype:
            return False
        return self == other

    def __call__(self, *, accum=None):
        return _ScalarUpdater(self, accum)

    def __lshift__(self, expr):
        _ScalarUpdater(self, None) << expr

    def update(self, expr):
        _ScalarUpdater(self, None) << expr


class _ScalarUpdater:
    def __init__(self, parent, accum):
        self.parent, self.accum = parent, accum

    def __lshift__(self, expr):
        import numpy as np

        s = self.parent
        new = expr.value if isinstance(expr, (Scalar, ScalarExpression)) else expr
        if new is not None and self.accum is not None and s._value is not None:
            op = get_typed_op(self.accum, s.dtype, kind="binary")
            new = _apply_host_binary(op.name, s.dtype.np_type.type(s.value), s.dtype.np_type.type(new))
        elif new is None and self.accum is not None:
            return  # nothing to accumulate
        s._value = None if new is None else s.dtype.np_type.type(new)

    update = __lshift__


def _apply_host_binary(name, a, b):
    import numpy as np

    with np.errstate(all="ignore"):
        return {"plus": lambda: a + b, "times": lambda: a * b, "min": lambda: min(a, b), "max": lambda: max(a, b),
                "first": lambda: a, "second": lambda: b, "any": lambda: b, "minus": lambda: a - b,
                "lor": lambda: bool(a) or bool(b), "land": lambda: bool(a) and bool(b), "lxor": lambda: bool(a) != bool(b),
                "pair": lambda: type(a)(1), "oneb": lambda: type(a)(1)}[name]()


class ScalarExpression:
    """A delayed scalar result (``v.reduce(...)``, ``v[i]``): ``.new()`` -> :class:`Scalar`; ``.value`` runs it."""

    def __init__(self, compute, dtype):
        self._compute, self.dtype = compute, lookup_dtype(dtype)

    @property
    def value(self):
        return self._compute()

    def new(self, dtype=None, *, name=None):
        s = Scalar(dtype if dtype is not None else self.dtype, name=name)
        s.value = self._compute()
        return s

    def __eq__(self, other):
        return self.new() == other

    def __hash__(self):
        return id(self)

    def __bool__(self):
        return bool(self.new())


class Expression:
    """A delayed C call: which function, which operator, which operands (reference
    VectorExpression/MatrixExpression, core/vector.py:2166+, core/matrix.py:3621+)."""

    def __init__(self, method_name, cfunc_name, args, *, op, output_type, shape, at=False, bt=False):
        self.method_name, self.cfunc_name, self.args, self.op = method_name, cfunc_name, args, op
        self.output_type, self.shape, self.at, self.bt = output_type, shape, at, bt

    @property
    def dtype(self):
        return self.op.return_type

    def new(self, dtype=None, *, mask=None, name=None, **opts):
        """Allocate the output, then the same update path (reference core/base.py:583-616)."""
        dtype = lookup_dtype(dtype) if dtype is not None else self.dtype
        out = self.output_type(dtype, *self.shape, name=name)
        if mask is None:
            out._update(self, opts=opts)
        else:
            out(mask=mask, **opts).update(self)
        return out

    def _force_library_error(self):
        # shape mismatches are raised BY THE LIBRARY (reference core/matrix.py:2260-2261)
        self.new()


class InfixMatMul:
    """``A @ B`` before a semiring is chosen (reference core/infix.py:546-596)."""

    def __init__(self, left, right):
        self.left, self.right = left, right

    def with_op(self, sr):
        from .matrix import Matrix, TransposedMatrix
        from .vector import Vector

        if isinstance(self.left, Vector):
            return self.left.vxm(self.right, sr)
        if isinstance(self.right, Vector):
            return self.left.mxv(self.right, sr)
        return self.left.mxm(self.right, sr)

    def new(self, dtype=None, *, mask=None, name=None, **opts):
        return self.with_op(semiring.plus_times).new(dtype, mask=mask, name=name, **opts)


class _Replace:
    def __repr__(self):
        return "replace"


_replace_singleton = _Replace()
