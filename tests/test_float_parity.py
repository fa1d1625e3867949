"""Floating point with REAL values (every other random test feeds small integers so that results are exact): the tolerance the
north star states -- 1e-6 relative for FP32 / FP64 -- against a float64 evaluation of the same products.

* configs[0] of BASELINE.json: 4096 x 4096, 1 % uniform density, FP64 U[0,1) values, plus_times mxv (the reference runs it on
  the CPU through SuiteSparse; reference call graphblas/core/matrix.py:2203-2262, docs/user_guide/operations.rst:67-75);
* the same in FP32, vxm and mxm, masks and accumulators;
* a hub row of 100 000 entries (GPU; 20 000 on the emulator): FP32 sums of many terms cut into wavefront partials, class
  strips and atomics -- the order of summation differs from a sequential loop, the error has to stay inside the tolerance.

The float64 reference is scipy / numpy (independent of the oracle and of the library); the C oracle is checked against it too.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import grb_oracle as O
from tests.backend import DEVICES, bind

RTOL = 1e-6  # BASELINE.json north_star: "within 1e-6 rel for fp64/fp32"


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


def _uniform(n, density, seed, np_dtype):
    from graphblas_amd import synthetic

    r, c, v = synthetic.uniform_coo(n, n, density, seed, np_dtype)
    return r, c, v


def _close(got, want64, rtol=RTOL):
    got = np.asarray(got, np.float64)
    scale = np.maximum(np.abs(want64), 1e-300)
    err = np.abs(got - want64) / scale
    assert err.max(initial=0.0) <= rtol, f"max relative error {err.max():.3e} > {rtol:g}"


@pytest.mark.parametrize("tname", ["FP64", "FP32"])
def test_config1_uniform_plus_times(gb, tname, request):
    """configs[0]: 4096 x 4096 at 1 % (167 k entries), U[0,1) values, dense U[0,1) operand: mxv, vxm, masked + accumulated mxv."""
    np_t = O.NP_OF[tname]
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    n = 4096 if on_gpu else 1024
    r, c, v = _uniform(n, 0.01, 1, np_t)
    rng = np.random.default_rng(2)
    x = rng.random(n).astype(np_t)
    S = sp.csr_matrix((v.astype(np.float64), (r, c)), shape=(n, n))
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=n, ncols=n)
    u = gb.Vector.from_coo(np.arange(n), x, dtype=tname, size=n)
    # mxv
    w = A.mxv(u, gb.semiring.plus_times).new()
    wi, wv = w.to_coo()
    want = S @ x.astype(np.float64)
    has = np.flatnonzero(np.diff(S.indptr) > 0)
    assert np.array_equal(wi.astype(np.int64), has)
    assert wv.dtype == np_t
    _close(wv, want[has])
    # vxm (the cached transpose)
    t = u.vxm(A, gb.semiring.plus_times).new()
    ti, tv = t.to_coo()
    want_t = S.T @ x.astype(np.float64)
    has_t = np.flatnonzero(np.diff(S.T.tocsr().indptr) > 0)
    assert np.array_equal(ti.astype(np.int64), has_t)
    _close(tv, want_t[has_t])
    # w<m> += A x  with a valued mask and the plus accumulator
    mi = np.flatnonzero(rng.random(n) < 0.5)
    mk = gb.Vector.from_coo(mi, np.ones(mi.size, bool), dtype="BOOL", size=n)
    w0 = rng.random(n).astype(np_t)
    wm = gb.Vector.from_coo(np.arange(n), w0, dtype=tname, size=n)
    wm(mk.V, accum=gb.binary.plus) << A.mxv(u, gb.semiring.plus_times)
    gi, gv = wm.to_coo()
    exp = w0.astype(np.float64).copy()
    sel = np.zeros(n, bool)
    sel[mi] = True
    sel &= np.diff(S.indptr) > 0
    exp[sel] += want[sel]
    assert np.array_equal(gi, np.arange(n, dtype=gi.dtype))
    _close(gv, exp)
    # the C oracle agrees with the float64 evaluation as well (it is the checker of every other test)
    oa = O.OMat.from_coo(r, c, v, n, n, tname)
    ow = O.mxv(oa, O.OVec(n, np.arange(n), x, tname), "plus_times")
    assert np.array_equal(ow.idx, has)
    _close(ow.vals, want[has])


@pytest.mark.parametrize("tname", ["FP64", "FP32"])
def test_real_valued_mxm(gb, tname, request):
    """C = A (+.x) A and C<A.S> = A (+.x) A with U[0,1) values against scipy in float64."""
    np_t = O.NP_OF[tname]
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    n = 2048 if on_gpu else 256
    r, c, v = _uniform(n, 0.02, 3, np_t)
    S = sp.csr_matrix((v.astype(np.float64), (r, c)), shape=(n, n))
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=n, ncols=n)
    C = A.mxm(A, gb.semiring.plus_times).new()
    ci, cj, cv = C.to_coo()
    # structure: the pattern product (scipy drops nothing here: all values are positive)
    P = (S @ S).tocsr()
    P.sort_indices()
    pr = np.repeat(np.arange(n), np.diff(P.indptr))
    assert np.array_equal(ci.astype(np.int64), pr) and np.array_equal(cj.astype(np.int64), P.indices)
    _close(cv, P.data)
    Cm = gb.Matrix(tname, n, n)
    Cm(A.S) << A.mxm(A, gb.semiring.plus_times)
    mi, mj, mv = Cm.to_coo()
    M = P.multiply(S.astype(bool)).tocsr()
    M.sort_indices()
    assert np.array_equal(mj.astype(np.int64), M.indices)
    _close(mv, M.data)


@pytest.mark.parametrize("tname,sr", [("FP32", "plus_times"), ("FP64", "plus_times"), ("FP32", "min_plus"), ("FP32", "max_times")])
def test_hub_rows_real_values(gb, tname, sr, request):
    """A power-law-ish matrix with hub rows (the longest: 100 000 entries on the GPU) and real values, forced through the long /
    short row split with the hot-column table: FP32 plus over 10^5 terms summed as wavefront partials + atomics stays within
    1e-6 of the float64 sum; min / max are exact up to the rounding of the single product they select."""
    from graphblas_amd import _lib

    np_t = O.NP_OF[tname]
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    m, n = (3000, 400_000) if on_gpu else (300, 30_000)
    hub = 100_000 if on_gpu else 20_000
    rng = np.random.default_rng(77)
    deg = rng.integers(0, 12, m)
    for k, ln in enumerate((hub, hub // 3, 5000, 1025, 513, 300, 64, 65)):
        deg[7 * k + 1] = min(ln, n)
    rows = np.repeat(np.arange(m), deg)
    cols = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg])
    vals = rng.random(rows.size).astype(np_t) + np_t(0.5 if sr == "max_times" else 0.0)
    x = (rng.random(n) * 4).astype(np_t)
    S = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(m, n))
    x64 = x.astype(np.float64)
    if sr == "plus_times":
        want = S @ x64
    else:
        want = np.full(m, np.inf if sr == "min_plus" else -np.inf)
        prod = (vals.astype(np.float64) + x64[cols]) if sr == "min_plus" else (vals.astype(np.float64) * x64[cols])
        (np.minimum if sr == "min_plus" else np.maximum).at(want, rows, prod)
    has = np.flatnonzero(deg > 0)
    try:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1)
        _lib.lib.GrX_option_set(b"hot_min_cols", 8)
        _lib.lib.GrX_option_set(b"push_mode", 0)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(np.arange(n), x, dtype=tname, size=n)
        for _ in range(2):  # (the second call runs on the cached layout)
            w = A.mxv(u, getattr(gb.semiring, sr)).new()
            wi, wv = w.to_coo()
            assert np.array_equal(wi.astype(np.int64), has)
            # the product of two FP32 numbers rounds once; min / max then pick one of them exactly
            _close(wv, want[has], rtol=RTOL)
    finally:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1 << 22)
        _lib.lib.GrX_option_set(b"hot_min_cols", 1 << 20)
        _lib.lib.GrX_option_set(b"push_mode", 1)
