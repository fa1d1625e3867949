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
        """Parse ``(mask, accum, replace)`` in any order (reference core/base.py:192-263)."""
        mask_arg = accum_arg = None
        for arg in args:
            if isinstance(arg, (Mask, BaseType)):
                if mask_arg is not None:
                    raise TypeError("Got multiple values for argument 'mask'")
                mask_arg = arg
            elif arg is _replace_singleton:
                replace = True
            else:
                if accum_arg is not None:
                    raise TypeError("Got multiple values for argument 'accum'")
                if not isinstance(arg, (str, TypedOp)) and not hasattr(arg, "opclass"):
                    raise TypeError(f"Invalid item found in output params: {type(arg)}")
                accum_arg = arg
        if mask_arg is not None and mask is not None:
            raise TypeError("Got multiple values for argument 'mask'")
        if mask_arg is not None:
            mask = mask_arg
        if mask is None:
            if replace:
                raise TypeError("'replace' argument may only be True if a mask is provided")
        else:
            mask = _check_mask(mask)
        if accum_arg is not None:
            if accum is not None:
                raise TypeError("Got multiple values for argument 'accum'")
            accum = accum_arg
        if accum is not None:
            accum = get_typed_op(accum, self.dtype, kind="binary")
            if accum.opclass == "Monoid":
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
        args = [self, mask, accum, expr.op, *expr.args, desc]
        call(expr.cfunc_name, args)

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
