"""graphblas_amd -- the python-graphblas user surface for the mxm / mxv / vxm hot path, dispatching to
hand-written gfx950 HIP kernels in ``csrc/libgrb_mi355x.so`` through the GraphBLAS C API.

Reference counterparts: package entry / backend init ``graphblas/__init__.py:107-199``; the user
syntax ``C(mask, accum, replace) << A.mxm(B, semiring)`` documented at
``docs/user_guide/operations.rst:1-175``.

    import graphblas_amd as gb
    gb.init()                       # needs an MI355X; there is no CPU fallback
    A = gb.Matrix.from_coo(rows, cols, vals)
    v = gb.Vector.from_coo([0], [0.0], size=A.nrows)
    v(gb.op.min) << gb.semiring.min_plus(v @ A)
"""
from . import _lib, descriptor, dtypes, exceptions
from .base import _replace_singleton as replace
from .base import record_calls
from .operators import agg, binary, monoid, op, semiring

_initialized = False
backend = "mi355x"
MAX_SIZE = 1 << 60  # largest Vector / Matrix dimension: GrB_INDEX_MAX + 1 (reference graphblas/__init__.py:42,210; tests/test_core.py:95-96)


def init(backend="mi355x", blocking=False, *, lib_path=None):
    """Bind the native library and initialise it (reference graphblas/__init__.py:107-199).
    Raises if no MI355X-class HIP device is available: the product has no CPU path."""
    global _initialized
    if backend != "mi355x":
        raise ValueError(f'Bad backend name.  Must be "mi355x".  Got: {backend!r}')
    if _initialized:
        return
    L = _lib.load(lib_path)
    rc = L.GrB_init(1 if blocking else 0)
    if rc != 0:
        raise exceptions._error_code_lookup.get(rc, exceptions.Panic)(
            "GrB_init failed: libgrb_mi355x.so needs a HIP device (gfx950); no CPU fallback exists"
        )
    _initialized = True


def is_initialized():
    return _initialized


from .matrix import Matrix, TransposedMatrix  # noqa: E402
from .vector import Vector  # noqa: E402
from .base import Scalar  # noqa: E402

__all__ = ["init", "MAX_SIZE", "Matrix", "Vector", "Scalar", "semiring", "binary", "monoid", "op", "agg", "dtypes", "replace", "exceptions"]
