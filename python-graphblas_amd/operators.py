"""Builtin BinaryOp / Monoid / Semiring objects and typed-handle lookup.

Mirrors, for the builtin operators on the mxm/mxv/vxm path only, the reference's
``graphblas/core/operator/`` package: typed-op selection ``get_typed_op``
(core/operator/utils.py:60-157), handle naming (core/operator/semiring.py:185-219: ``GrB_<MONOID>_
<BINARY>_SEMIRING_<TYPE>`` / ``GxB_<MONOID>_<BINARY>_<TYPE>``), BOOL coercions
(semiring.py:538-548, 568-587) and string spellings such as ``"min_plus"`` / ``"+"``
(tests/test_vector.py:362-368, docs/user_guide/operations.rst:44).  Of the aggregators, the ones that are a monoid
reduction or one semiring mat-vec (``agg.sum/prod/min/max/all/any/any_value/count/exists``, core/operator/agg.py:356-378) are
here; user-defined functions, UDTs, composite aggregators and positional ops need a JIT and are outside the path
(SURVEY.md section 2 #6).
"""
from __future__ import annotations

from . import _lib
from .dtypes import BOOL, DataType, lookup_dtype, unify

_BINARY_NAMES = ["first", "second", "pair", "oneb", "plus", "minus", "times", "min", "max", "any", "lor", "land",
                 "lxor", "lxnor", "eq", "ne", "gt", "lt", "ge", "le"]
_COMPARISONS = {"eq", "ne", "gt", "lt", "ge", "le"}  # T x T -> BOOL: element-wise vector operations only
_MONOID_NAMES = ["plus", "times", "min", "max", "any", "lor", "land", "lxor", "lxnor"]
_BOOL_ONLY = {"lor", "land", "lxor", "lxnor"}
# symbol spellings per operator kind (reference core/operator/utils.py:347-370: "==" is binary.eq among the binary
# operators and monoid.eq -- the BOOL-only lxnor -- among the monoids; the NAME "eq" always means binary.eq for a BinaryOp)
_BINARY_SYMBOLS = {"+": "plus", "*": "times", "-": "minus", "|": "lor", "&": "land", "^": "lxor", "==": "eq", "!=": "ne",
                   "<": "lt", ">": "gt", "<=": "le", ">=": "ge"}
_MONOID_SYMBOLS = {"+": "plus", "*": "times", "|": "lor", "&": "land", "^": "lxor", "==": "lxnor", "eq": "lxnor"}
# BOOL renames (SuiteSparse convention; reference semiring.py:568-587)
_BOOL_RENAME = {"plus": "lor", "times": "land", "min": "land", "max": "lor", "minus": "lxor"}


class TypedOp:
    """An operator bound to a dtype: ``.gb_obj`` is the library handle (reference:
    TypedBuiltinSemiring, core/operator/semiring.py:39-89)."""

    __slots__ = ("parent", "name", "type", "return_type", "gb_name", "opclass")

    def __init__(self, parent, name, type_, return_type, gb_name, opclass):
        self.parent, self.name, self.type, self.return_type = parent, name, type_, return_type
        self.gb_name, self.opclass = gb_name, opclass

    @property
    def gb_obj(self):
        return _lib.handle(self.gb_name)

    _carg = gb_obj

    @property
    def binaryop(self):  # Monoid -> BinaryOp (reference core/base.py:257-258)
        return self.parent.binaryop[self.type]

    def __repr__(self):
        return f"{self.opclass.lower()}.{self.name}[{self.type}]"


class _OpBase:
    opclass = "Op"

    def __init__(self, name):
        self.name = name
        self._typed = {}

    def _gb_candidates(self, dtype):
        raise NotImplementedError

    def _coerce(self, dtype):
        return dtype

    def __getitem__(self, dtype):
        dtype = self._coerce(lookup_dtype(dtype))
        t = self._typed.get(dtype)
        if t is None:
            for gb_name in self._gb_candidates(dtype):
                if _lib.has_symbol(gb_name):
                    t = TypedOp(self, self.name, dtype, BOOL if self.name in _COMPARISONS else dtype, gb_name, self.opclass)
                    break
            else:
                raise KeyError(f"{self.name} does not work with {dtype}")
            self._typed[dtype] = t
        return t

    def __contains__(self, dtype):
        try:
            self[dtype]
            return True
        except (KeyError, ValueError, TypeError):
            return False

    def __repr__(self):
        return f"{self.opclass.lower()}.{self.name}"


def _canon(name, dtype):
    return _BOOL_RENAME.get(name, name) if dtype is BOOL else name


class BinaryOp(_OpBase):
    opclass = "BinaryOp"

    def _coerce(self, dtype):
        return BOOL if self.name in _BOOL_ONLY else dtype

    def _gb_candidates(self, dtype):
        n = {"oneb": "ONEB", "pair": "ONEB"}.get(self.name, self.name.upper())
        t = dtype.name
        if self.name in _BOOL_ONLY:
            return [f"GrB_{n}", f"GxB_{n}_{t}"]
        return [f"GrB_{n}_{t}", f"GxB_{n}_{t}"]

    @property
    def monoid(self):
        """The monoid built on this operator, or None (reference: ``BinaryOp.monoid``, core/operator/binary.py)."""
        return getattr(monoid, self.name, None)


class Monoid(_OpBase):
    opclass = "Monoid"

    def _coerce(self, dtype):
        return BOOL if self.name in _BOOL_ONLY else dtype

    def _gb_candidates(self, dtype):
        n = _canon(self.name, dtype).upper()
        t = dtype.name
        return [f"GrB_{n}_MONOID_{t}", f"GxB_{n}_{t}_MONOID"]

    @property
    def binaryop(self):
        return getattr(binary, self.name)

    def identity(self, dtype):
        """The monoid's identity as a numpy scalar of ``dtype`` (reference: ``TypedBuiltinMonoid.identity``,
        core/operator/monoid.py:34-60)."""
        import numpy as np

        dtype = self._coerce(lookup_dtype(dtype))
        t = dtype.np_type
        name = _canon(self.name, dtype)
        if name in ("plus", "lor", "lxor", "any"):
            return t.type(0)
        if name in ("times", "land", "lxnor"):
            return t.type(1)
        if name not in ("min", "max"):
            raise ValueError(f"{self!r} has no identity for {dtype}")
        if np.issubdtype(t, np.floating):
            return t.type(np.inf if name == "min" else -np.inf)
        info = np.iinfo(t)
        return t.type(info.max if name == "min" else info.min)


class Aggregator:
    """The aggregators that are one monoid reduction or one semiring mat-vec with a dense iso vector
    (reference core/operator/agg.py:264-283 builds ``semiring(A @ init)``; the table at :356-378 names them)."""

    opclass = "Aggregator"

    def __init__(self, name, *, monoid=None, semiring=None, any_dtype=None):
        self.name, self.monoid, self.semiring, self.any_dtype = name, monoid, semiring, any_dtype

    def __repr__(self):
        return f"agg.{self.name}"


class Semiring(_OpBase):
    opclass = "Semiring"

    def __init__(self, name, monoid, binaryop):
        super().__init__(name)
        self.monoid, self.binaryop = monoid, binaryop

    def _coerce(self, dtype):
        # lor_land & co. on non-bool inputs run in BOOL (reference semiring.py:538-548)
        if self.monoid.name in _BOOL_ONLY:
            return BOOL
        return dtype

    def _gb_candidates(self, dtype):
        m, b = _canon(self.monoid.name, dtype).upper(), _canon(self.binaryop.name, dtype).upper()
        if b == "ONEB":
            b = "PAIR"
        t = dtype.name
        return [f"GrB_{m}_{b}_SEMIRING_{t}", f"GxB_{m}_{b}_{t}"]

    def __call__(self, expr):
        """Functional style: ``semiring.min_plus(v @ G)`` (docs/user_guide/operations.rst:45)."""
        from .base import InfixMatMul

        if not isinstance(expr, InfixMatMul):
            raise TypeError(f"Expected an `A @ B` expression, got {type(expr).__name__}")
        return expr.with_op(self)


class _Namespace:
    def __init__(self, kind):
        self._kind = kind

    def __repr__(self):
        return f"<namespace {self._kind}>"


binary = _Namespace("binary")
monoid = _Namespace("monoid")
semiring = _Namespace("semiring")
op = _Namespace("op")
agg = _Namespace("agg")

for _n in _BINARY_NAMES:
    setattr(binary, _n, BinaryOp(_n))
    setattr(op, _n, getattr(binary, _n))
for _n in _MONOID_NAMES:
    setattr(monoid, _n, Monoid(_n))
for _m in _MONOID_NAMES:
    for _b in _BINARY_NAMES:
        if _b == "oneb" or _b in _COMPARISONS:
            continue
        _s = Semiring(f"{_m}_{_b}", getattr(monoid, _m), getattr(binary, _b))
        setattr(semiring, _s.name, _s)
        setattr(op, _s.name, _s)


from .dtypes import INT64 as _INT64  # noqa: E402

for _a, _m in [("sum", "plus"), ("prod", "times"), ("all", "land"), ("any", "lor"), ("min", "min"), ("max", "max"),
               ("any_value", "any")]:
    setattr(agg, _a, Aggregator(_a, monoid=getattr(monoid, _m)))
agg.count = Aggregator("count", semiring=semiring.plus_pair, any_dtype=_INT64)
agg.exists = Aggregator("exists", semiring=semiring.any_pair, any_dtype=_INT64)


def _from_string(string, kind):
    s = string.strip()
    if kind == "semiring":
        for sep in ("_", "."):
            if sep in s:
                left, right = s.split(sep, 1)
                left, right = _MONOID_SYMBOLS.get(left, left), _BINARY_SYMBOLS.get(right, right)
                if hasattr(monoid, left) and hasattr(binary, right) and hasattr(semiring, f"{left}_{right}"):
                    return getattr(semiring, f"{left}_{right}")
        raise ValueError(f"Unknown semiring string: {string!r}")
    name = (_MONOID_SYMBOLS if kind == "monoid" else _BINARY_SYMBOLS).get(s, s)
    ns = monoid if kind == "monoid" else binary
    if not hasattr(ns, name):
        raise ValueError(f"Unknown {kind} string: {string!r}")
    return getattr(ns, name)


def get_typed_op(op_, dtype, dtype2=None, *, kind=None):
    """reference core/operator/utils.py:60-157 restricted to builtin ops.  ``kind`` is "binary"
    (accum: BinaryOp or Monoid, tests/test_vector.py:350-368) or "semiring"."""
    if isinstance(op_, TypedOp):
        return op_
    if isinstance(op_, str):
        op_ = _from_string(op_, "semiring" if kind == "semiring" else "binary")
    if not isinstance(op_, _OpBase):
        raise TypeError(f"Unable to get typed operator from object with type {type(op_)}")
    if kind == "semiring" and not isinstance(op_, Semiring):
        raise TypeError(f"Expected type: Semiring; got {op_.opclass} `{op_!r}`")
    if kind == "binary" and isinstance(op_, Semiring):
        raise TypeError(f"Expected type: BinaryOp, Monoid; got Semiring `{op_!r}`")
    dtype = lookup_dtype(dtype)
    if dtype2 is not None:
        dtype = unify(dtype, lookup_dtype(dtype2))
    try:
        return op_[dtype]
    except KeyError:
        raise TypeError(f"Operator {op_!r} does not work with {dtype}") from None
