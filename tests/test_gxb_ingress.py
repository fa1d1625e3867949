"""The reference's zero-copy ingress under the names it binds: ``Matrix.ss.import_csr`` / ``A.ss.pack_csr`` ->
``GxB_Matrix_import_CSR`` / ``GxB_Matrix_pack_CSR`` (reference graphblas/core/ss/matrix.py:1139-1349).  Both tiers:
the HIP library on the GPU box, the same sources under the CPU emulator here.  Checked: the round trip against the inputs, the
ownership protocol of the raw C entry (the library releases the host arrays and NULLs the caller's cells; a failed call leaves
them alone), iso and jumbled inputs, pack into an existing matrix, and a product on an imported matrix against the oracle."""
import ctypes

import numpy as np
import pytest

from oracle import grb_oracle as O
from tests.backend import DEVICES, bind
from tests.test_random_parity import same_vec


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


def random_csr(rng, nrows, ncols, density, dtype):
    mask = rng.random((nrows, ncols)) < density
    indptr = np.concatenate([[0], np.cumsum(mask.sum(1))]).astype(np.uint64)
    cols = np.nonzero(mask)[1].astype(np.uint64)
    vals = rng.integers(1, 50, cols.size).astype(dtype)
    return indptr, cols, vals


@pytest.mark.parametrize("tname,np_t", [("FP32", np.float32), ("INT64", np.int64), ("BOOL", np.bool_), ("FP64", np.float64)])
def test_import_csr_round_trip_and_product(gb, tname, np_t):
    rng = np.random.default_rng(77)
    nrows, ncols = 37, 53
    indptr, cols, vals = random_csr(rng, nrows, ncols, 0.2, np_t)
    A = gb.Matrix.ss.import_csr(nrows=nrows, ncols=ncols, indptr=indptr, values=vals, col_indices=cols, sorted_cols=True, name="A")
    assert A.shape == (nrows, ncols) and A.dtype.name == tname and A.nvals == cols.size
    r, c, v = A.to_coo()
    exp_rows = np.repeat(np.arange(nrows, dtype=np.uint64), np.diff(indptr).astype(np.int64))
    assert r.tolist() == exp_rows.tolist() and c.tolist() == cols.tolist() and v.tolist() == vals.tolist()
    # the caller's numpy arrays are untouched (the host mirror hands over malloc'd copies)
    assert int(indptr[-1]) == cols.size
    # a product on it, against the oracle
    sr = "lor_land" if tname == "BOOL" else "min_plus"
    ui = np.arange(ncols)
    uv = rng.integers(1, 9, ncols).astype(np_t)
    u = gb.Vector.from_coo(ui, uv, dtype=tname, size=ncols)
    oa = O.OMat(nrows, ncols, indptr.astype(np.int64), cols.astype(np.int64), vals, tname)
    same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), O.mxv(oa, O.OVec(ncols, ui, uv, tname), sr))


def test_import_csr_iso_and_jumbled(gb):
    rng = np.random.default_rng(78)
    nrows, ncols = 20, 30
    indptr, cols, _ = random_csr(rng, nrows, ncols, 0.3, np.int64)
    # jumbled: reverse the columns inside every row; iso: one value for every entry
    jumbled = cols.copy()
    for i in range(nrows):
        a, b = int(indptr[i]), int(indptr[i + 1])
        jumbled[a:b] = cols[a:b][::-1]
    A = gb.Matrix.ss.import_csr(nrows=nrows, ncols=ncols, indptr=indptr, values=np.array([7], np.int64), col_indices=jumbled, is_iso=True,
                                sorted_cols=False)
    r, c, v = A.to_coo()
    assert c.tolist() == cols.tolist() and set(v.tolist()) <= {7} and v.size == cols.size
    with pytest.raises(ValueError):
        gb.Matrix.ss.import_csr(nrows=nrows, ncols=ncols, indptr=indptr, values=np.array([7], np.int64), col_indices=cols, is_iso=True, format="csc")


def test_pack_csr_replaces_the_content(gb):
    rng = np.random.default_rng(79)
    nrows, ncols = 16, 16
    A = gb.Matrix.from_coo([0, 3, 5], [1, 2, 7], [1.0, 2.0, 3.0], dtype="FP32", nrows=nrows, ncols=ncols)
    indptr, cols, vals = random_csr(rng, nrows, ncols, 0.25, np.float32)
    same = A.ss.pack_csr(indptr=indptr, values=vals, col_indices=cols, sorted_cols=True)
    assert same is A and A.nvals == cols.size
    r, c, v = A.to_coo()
    assert c.tolist() == cols.tolist() and v.tolist() == vals.tolist()
    with pytest.raises(TypeError):
        gb.Matrix.ss.pack_csr(indptr=indptr, values=vals, col_indices=cols)


def test_raw_entry_takes_ownership_and_nulls_the_cells(gb):
    """The C protocol itself (what the reference's cffi wrapper relies on, core/ss/matrix.py:1316-1349): arrays from malloc, sizes in
    bytes; on success *Ap = *Aj = *Ax = NULL and the library has released them; on failure they stay the caller's."""
    from graphblas_amd import _lib

    L = _lib.lib
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]

    def owned(a):
        p = libc.malloc(max(a.nbytes, 1))
        ctypes.memmove(p, a.ctypes.data, a.nbytes)
        return ctypes.c_void_p(p)

    indptr = np.array([0, 2, 2, 3], np.uint64)
    cols = np.array([0, 2, 1], np.uint64)
    vals = np.array([5, 6, 7], np.int32)
    ap, aj, ax = owned(indptr), owned(cols), owned(vals)
    h = ctypes.c_void_p()
    rc = L.GxB_Matrix_import_CSR(ctypes.byref(h), ctypes.c_void_p(_lib.handle("GrB_INT32")), 3, 3, ctypes.byref(ap), ctypes.byref(aj), ctypes.byref(ax),
                                 indptr.nbytes, cols.nbytes, vals.nbytes, False, False, None)
    assert rc == 0 and h.value and ap.value is None and aj.value is None and ax.value is None
    nv = ctypes.c_uint64()
    assert L.GrB_Matrix_nvals(ctypes.byref(nv), h) == 0 and nv.value == 3
    L.GrB_Matrix_free(ctypes.byref(h))
    # a failing call (Ap_size too small): the cells keep their arrays, which the caller then frees
    ap, aj, ax = owned(indptr), owned(cols), owned(vals)
    h = ctypes.c_void_p()
    rc = L.GxB_Matrix_import_CSR(ctypes.byref(h), ctypes.c_void_p(_lib.handle("GrB_INT32")), 3, 3, ctypes.byref(ap), ctypes.byref(aj), ctypes.byref(ax),
                                 8, cols.nbytes, vals.nbytes, False, False, None)
    assert rc != 0 and not h.value and ap.value and aj.value and ax.value
    for cell in (ap, aj, ax):
        libc.free(cell)


@pytest.mark.parametrize("count", [300, 600, 150_000, 1_200_000])
def test_host_copies_staged_and_direct(gb, count):
    """The caller's tuples cross the boundary through the context's page-locked blocks (up to 4 KiB / 8 MiB per array: csrc/grb_context.hip
    h2d / d2h) or straight from / to the caller's memory (larger): 2.4 KB, 4.8 KB, 1.2 MB and 9.6 MB per array, in and out, values intact."""
    import tests.backend as backend

    if count > 200_000 and backend._bound == "emu":
        pytest.skip("the direct path (arrays above 8 MiB) on the GPU tier only: 17 s of sorting under the emulator")
    rng = np.random.default_rng(count)
    n = 4 * count
    idx = np.sort(rng.choice(n, size=count, replace=False)).astype(np.uint64)
    vals = rng.integers(-(1 << 40), 1 << 40, size=count)
    v = gb.Vector.from_coo(idx, vals, dtype="INT64", size=n)
    assert v.nvals == count
    gi, gv = v.to_coo()
    assert np.array_equal(gi, idx) and np.array_equal(gv, vals)
    w = gb.Vector.from_coo(idx[::-1].copy(), vals[::-1].copy(), dtype="INT64", size=n)  # (unsorted input: the device sort sees the same bytes)
    assert w.isequal(v)
