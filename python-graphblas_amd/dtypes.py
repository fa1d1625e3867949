"""Builtin data types; mirrors the builtin part of graphblas/core/dtypes.py (table :329-420,
``lookup_dtype`` :527-549, ``unify`` :552-568).  UDTs and complex types are outside the path."""
import numpy as np

from . import _lib


class DataType:
    __slots__ = ("name", "np_type", "_carg_name")

    def __init__(self, name, np_type):
        self.name = name
        self.np_type = np.dtype(np_type)
        self._carg_name = f"GrB_{name}"

    @property
    def _carg(self):
        return _lib.handle(self._carg_name)

    gb_obj = _carg

    def __repr__(self):
        return self.name

    def __eq__(self, other):
        try:
            return lookup_dtype(other) is self
        except (ValueError, TypeError):
            return False

    def __hash__(self):
        return hash(self.name)


BOOL = DataType("BOOL", np.bool_)
INT8 = DataType("INT8", np.int8)
INT16 = DataType("INT16", np.int16)
INT32 = DataType("INT32", np.int32)
INT64 = DataType("INT64", np.int64)
UINT8 = DataType("UINT8", np.uint8)
UINT16 = DataType("UINT16", np.uint16)
UINT32 = DataType("UINT32", np.uint32)
UINT64 = DataType("UINT64", np.uint64)
FP32 = DataType("FP32", np.float32)
FP64 = DataType("FP64", np.float64)
_ALL = [BOOL, INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64, FP32, FP64]

_registry = {}
for _dt in _ALL:
    _registry[_dt.name] = _dt
    _registry[_dt.name.lower()] = _dt
    _registry[_dt.np_type] = _dt
    _registry[_dt.np_type.name] = _dt
    _registry[_dt.np_type.type] = _dt
_registry[bool] = BOOL
_registry[int] = INT64
_registry[float] = FP64
_registry["bool"] = BOOL
_registry["int"] = INT64
_registry["float"] = FP64  # 'float' follows numpy/Python (reference dtypes.py:521)


def lookup_dtype(key, value=None):
    if type(key) is DataType:
        return key
    try:
        return _registry[key]
    except (KeyError, TypeError):
        pass
    if value is not None and hasattr(value, "dtype") and value.dtype in _registry:
        return _registry[value.dtype]
    if key is None:
        raise TypeError("Bad dtype: None.  A valid dtype must be provided.")
    try:
        return _registry[np.dtype(key)]
    except (KeyError, TypeError):
        raise ValueError(f"Unknown dtype: {key} of type {type(key)}") from None


def unify(type1, type2):
    """A type that can hold both (numpy promote_types): unify(INT8, UINT16) -> INT32,
    unify(FP32, INT32) -> FP64 (reference dtypes.py:552-568)."""
    if type1 is type2:
        return type1
    return lookup_dtype(np.promote_types(type1.np_type, type2.np_type))
