"""CPU tier: libgrb_mi355x.so loads without a GPU and exports every function and every global handle that
include/grb_mi355x.h declares (the header is preprocessed with gcc -E, so the macro-generated per-type entry
points and operator handles are checked too).  No compute call is made: GrB_init needs a device."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "grb_mi355x.h")
LIB = os.path.join(ROOT, "python-graphblas_amd", "csrc", "libgrb_mi355x.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libgrb_mi355x.so not built (run __graft_entry__.build())")
    return ctypes.CDLL(LIB)


@pytest.fixture(scope="module")
def declared():
    text = subprocess.check_output(["gcc", "-E", "-P", HEADER], text=True)
    funcs = set(re.findall(r"\b(?:GrB_Info|const char \*)\s*(G[rx][BX]_\w+)\s*\(", text))
    data = set()
    for m in re.finditer(r"\bextern\s+(?:const\s+)?\w+\s*\*?\s*([^;]+);", text):
        for name in m.group(1).split(","):
            name = name.strip().lstrip("*").strip()
            if re.fullmatch(r"G[rx]B_\w+", name):
                data.add(name)
    return funcs, data


def test_exports_every_declared_symbol(lib, declared):
    funcs, data = declared
    assert len(funcs) >= 100 and len(data) >= 500, (len(funcs), len(data))  # 11 types x 6 typed entry points, 32 descriptors, ...
    missing = [f for f in sorted(funcs) if not hasattr(lib, f)]
    for name in sorted(data):
        try:
            if not ctypes.c_void_p.in_dll(lib, name).value:
                missing.append(name + " (NULL handle)")
        except ValueError:
            missing.append(name)
    assert not missing, missing[:20]


def test_hot_path_and_discovery_names(lib, declared):
    funcs, data = declared
    for f in ("GrB_mxm", "GrB_mxv", "GrB_vxm", "GrB_Matrix_build_FP32", "GrB_Matrix_import_INT64", "GrB_Vector_extractTuples_BOOL",
              "GrB_Matrix_error", "GrB_Descriptor_set", "GrX_Matrix_import_CSR_device",
              "GxB_Matrix_import_CSR", "GxB_Matrix_pack_CSR", "GxB_init"):  # (the reference's zero-copy ingress names, core/ss/matrix.py:1316)
        assert f in funcs
    # the names python-graphblas discovers by regex over dir(lib) (core/operator/semiring.py:185-219, descriptor.py:51-84)
    for d in ("GrB_PLUS_TIMES_SEMIRING_FP64", "GrB_MIN_PLUS_SEMIRING_FP32", "GrB_LOR_LAND_SEMIRING_BOOL", "GxB_ANY_PAIR_INT64",
              "GrB_PLUS_TIMES_SEMIRING_INT64", "GrB_DESC_RSC", "GrB_DESC_T0T1", "GrB_BOOL", "GrB_FP64", "GrB_MIN_FP32",
              "GrB_PLUS_MONOID_INT64", "GrB_ALL"):
        assert d in data


def test_init_needs_a_device(lib):
    """No CPU fallback: without a HIP device GrB_init must fail loudly (GrB_PANIC = -101)."""
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    lib.GrB_init.argtypes = [ctypes.c_int]
    assert lib.GrB_init(0) == -101


def test_import_time_surface(lib, declared):
    """Handle types, builtin unary / index-unary operator handles and the entry points of the operations outside the path exist
    (python-graphblas resolves them while it is imported: graphblas/core/mask.py:1-5, core/operator/base.py:803-893); GrB_Scalar
    is real and needs no device."""
    funcs, data = declared
    for d in ("GrB_VALUENE_INT64", "GxB_ONE_BOOL", "GrB_IDENTITY_FP64", "GrB_LNOT", "GrB_TRIL", "GrB_ROWINDEX_INT64", "GrB_DIV_FP32",
              "GxB_RMINUS_INT32"):
        assert d in data and ctypes.c_void_p.in_dll(lib, d).value
    for f in ("GrB_Matrix_apply", "GrB_Vector_select_INT64", "GrB_Matrix_eWiseAdd_BinaryOp", "GrB_Matrix_assign_FP64", "GrB_Matrix_extract",
              "GrB_Matrix_kronecker_Semiring", "GrB_Scalar_new", "GrB_Vector_assign", "GrB_Vector_extract", "GrX_mxm_streamed"):
        assert f in funcs and hasattr(lib, f)
    s = ctypes.c_void_p()
    assert lib.GrB_Scalar_new(ctypes.byref(s), ctypes.c_void_p.in_dll(lib, "GrB_FP64")) == 0
    n = ctypes.c_uint64(7)
    assert lib.GrB_Scalar_nvals(ctypes.byref(n), s) == 0 and n.value == 0
    x = ctypes.c_double(0)
    assert lib.GrB_Scalar_extractElement_FP64(ctypes.byref(x), s) == 1  # GrB_NO_VALUE
    lib.GrB_Scalar_setElement_INT32.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    assert lib.GrB_Scalar_setElement_INT32(s, -5) == 0
    assert lib.GrB_Scalar_extractElement_FP64(ctypes.byref(x), s) == 0 and x.value == -5.0
    assert lib.GrB_Scalar_nvals(ctypes.byref(n), s) == 0 and n.value == 1
    assert lib.GrB_Scalar_clear(s) == 0 and lib.GrB_Scalar_nvals(ctypes.byref(n), s) == 0 and n.value == 0
    assert lib.GrB_Scalar_free(ctypes.byref(s)) == 0 and s.value is None
    # an entry point outside the path answers GrB_NOT_IMPLEMENTED (-8) even on NULL arguments
    lib.GrB_Matrix_apply.argtypes = [ctypes.c_void_p] * 6
    assert lib.GrB_Matrix_apply(None, None, None, None, None, None) == -8
