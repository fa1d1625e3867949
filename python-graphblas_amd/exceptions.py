"""GrB_Info -> Python exception mapping; mirrors graphblas/exceptions.py:1-189 of the reference
(same class names, raised from the library's GrB_<Type>_error text)."""
import ctypes

from . import _lib


class GraphblasException(Exception):
    pass


class NoValue(GraphblasException): pass
class UninitializedObject(GraphblasException): pass
class InvalidObject(GraphblasException): pass
class NullPointer(GraphblasException): pass
class InvalidValue(GraphblasException): pass
class InvalidIndex(GraphblasException): pass
class DomainMismatch(GraphblasException): pass
class DimensionMismatch(GraphblasException): pass
class OutputNotEmpty(GraphblasException): pass
class EmptyObject(GraphblasException): pass
class OutOfMemory(GraphblasException): pass
class InsufficientSpace(GraphblasException): pass
class IndexOutOfBound(GraphblasException): pass
class Panic(GraphblasException): pass
class NotImplementedException(GraphblasException): pass
class UdfParseError(GraphblasException): pass


# numeric values: include/grb_mi355x.h (GraphBLAS C API 2.0)
_error_code_lookup = {
    1: NoValue,
    -1: UninitializedObject,
    -2: NullPointer,
    -3: InvalidValue,
    -4: InvalidIndex,
    -5: DomainMismatch,
    -6: DimensionMismatch,
    -7: OutputNotEmpty,
    -8: NotImplementedException,
    -101: Panic,
    -102: OutOfMemory,
    -103: InsufficientSpace,
    -104: InvalidObject,
    -105: IndexOutOfBound,
    -106: EmptyObject,
}


def check_status(response_code, args):
    """reference: exceptions.py:153-168 -- error text comes from GrB_<Type>_error on the FIRST argument."""
    if response_code == 0:
        return None
    if response_code == 1:
        return NoValue
    arg = args[0] if isinstance(args, (list, tuple)) else args
    text = ""
    kind = getattr(arg, "_grb_kind", None)
    carg = getattr(arg, "_carg", None)
    if kind in ("Matrix", "Vector") and carg:
        s = ctypes.c_char_p()
        getattr(_lib.lib, f"GrB_{kind}_error")(ctypes.byref(s), carg)
        text = (s.value or b"").decode()
    raise _error_code_lookup.get(response_code, Panic)(text)
