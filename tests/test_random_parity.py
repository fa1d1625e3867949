"""Seeded random parity: the HIP path (GPU tier) / the same kernels under the SIMT emulator (CPU tier)
against the CPU oracle, bit-exact for integer, boolean and min/max semirings; floating-point values are
small integers, so plus_times is exact as well and the comparison stays bit-exact.

Covers: empty and hub rows (longer than a merge tile, so the tile seams are exercised), ragged sizes,
all mask/complement/structure/replace combinations, accumulators, w aliased with u, a mask aliased with w,
vxm (transpose cache), desc T0/T1, operands whose type differs from the semiring type."""
import numpy as np
import pytest

from oracle import grb_oracle as O
from tests.backend import DEVICES, bind

TYPES = ["INT64", "FP32", "BOOL", "FP64", "INT8", "UINT16", "INT32"]
import os

DEFAULT_SHORT_KERNEL = int(os.environ.get("GRB_SHORT_KERNEL", "6"))
DEFAULT_LONG_KERNEL = int(os.environ.get("GRB_LONG_KERNEL", "5"))  # what tests restore after forcing a long-row kernel


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


def rand_coo(rng, m, n, tname, long_rows=0):
    np_t = O.NP_OF[tname]
    deg = rng.integers(0, 6, m)
    deg[rng.random(m) < 0.3] = 0
    deg = np.minimum(deg, n)
    for _ in range(long_rows):
        deg[rng.integers(0, m)] = rng.integers(n // 2, n + 1)
    rows = np.repeat(np.arange(m), deg)
    cols = np.concatenate([rng.choice(n, d, replace=False) for d in deg]) if deg.sum() else np.zeros(0, np.int64)
    return rows, cols, rand_vals(rng, len(rows), tname)


def rand_vals(rng, k, tname):
    np_t = O.NP_OF[tname]
    if tname == "BOOL":
        return rng.random(k) < 0.8
    if tname.startswith("FP"):
        return rng.integers(1, 9, k).astype(np_t)
    return rng.integers(0, 100, k).astype(np_t)


def rand_vec(rng, n, dens, tname):
    idx = np.flatnonzero(rng.random(n) < dens)
    return idx, rand_vals(rng, len(idx), tname)


def semirings_for(tname):
    if tname == "BOOL":
        return ["lor_land", "any_pair", "land_lor", "lxor_land"]
    return ["plus_times", "min_plus", "any_pair", "max_plus", "plus_plus", "min_second", "max_first", "plus_pair"]


def same_vec(got, exp):
    gi, gv = got.to_coo()
    assert gi.tolist() == exp.idx.tolist()
    assert gv.tolist() == exp.vals.tolist()


@pytest.mark.parametrize("seed", range(40))
def test_mxv_random(gb, seed):
    rng = np.random.default_rng(seed)
    tname = TYPES[seed % 7]
    srs = semirings_for(tname)
    sr = srs[rng.integers(len(srs))]
    m, n = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    if seed % 5 == 0:
        m, n = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    r, c, v = rand_coo(rng, m, n, tname, long_rows=int(rng.integers(0, 3)))
    ui, uv = rand_vec(rng, n, [1.0, 0.5, 0.05][seed % 3], tname)
    wi, wv = rand_vec(rng, m, 0.4, tname)
    mi, mv = rand_vec(rng, m, 0.5, "INT8")
    use_mask = seed % 4 != 0
    comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
    accum = [None, "plus", "min", "second", "max"][rng.integers(5)]
    exp = O.mxv(O.OMat.from_coo(r, c, v, m, n, tname), O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, tname),
                mask=O.OVec(m, mi, mv, "INT8") if use_mask else None, mask_comp=comp and use_mask,
                mask_struct=struct, accum=accum, replace=repl and use_mask)
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=m, ncols=n)
    u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
    w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
    kw = {}
    if use_mask:
        mk = gb.Vector.from_coo(mi, mv, dtype="INT8", size=m)
        mm = mk.S if struct else mk.V
        kw = dict(mask=~mm if comp else mm, replace=repl)
    if accum:
        kw["accum"] = accum
    w(**kw) << A.mxv(u, getattr(gb.semiring, sr))
    same_vec(w, exp)


@pytest.mark.parametrize("seed", range(16))
def test_vxm_and_transposes_random(gb, seed):
    rng = np.random.default_rng(100 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[seed % 4]
    m, n = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
    r, c, v = rand_coo(rng, m, n, tname, long_rows=1)
    oa = O.OMat.from_coo(r, c, v, m, n, tname)
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=m, ncols=n)
    ui, uv = rand_vec(rng, m, 0.3, tname)
    xi, xv = rand_vec(rng, n, 0.3, tname)
    u, ou = gb.Vector.from_coo(ui, uv, dtype=tname, size=m), O.OVec(m, ui, uv, tname)
    x, ox = gb.Vector.from_coo(xi, xv, dtype=tname, size=n), O.OVec(n, xi, xv, tname)
    S = getattr(gb.semiring, sr)
    same_vec(u.vxm(A, S).new(), O.vxm(ou, oa, sr))                        # u' A
    same_vec(x.vxm(A.T, S).new(), O.vxm(ox, oa, sr, transpose_b=True))    # x' A'  (desc T1)
    same_vec(A.T.mxv(u, S).new(), O.mxv(oa, ou, sr, transpose_a=True))    # A' u   (desc T0)
    same_vec(A.mxv(x, S).new(), O.mxv(oa, ox, sr))


@pytest.mark.parametrize("seed", range(8))
def test_aliasing(gb, seed):
    """w is also u (BFS style  q(~v.S, replace) << A.mxv(q)); mask is also w."""
    rng = np.random.default_rng(200 + seed)
    tname = ["BOOL", "INT64", "FP32", "FP64"][seed % 4]
    sr = "lor_land" if tname == "BOOL" else ["min_plus", "plus_times"][seed % 2]
    n = int(rng.integers(2, 2500))
    r, c, v = rand_coo(rng, n, n, tname, long_rows=1)
    oa = O.OMat.from_coo(r, c, v, n, n, tname)
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=n, ncols=n)
    qi, qv = rand_vec(rng, n, 0.2, tname)
    vi, vv = rand_vec(rng, n, 0.5, "BOOL")
    q, oq = gb.Vector.from_coo(qi, qv, dtype=tname, size=n), O.OVec(n, qi, qv, tname)
    vis, ovis = gb.Vector.from_coo(vi, vv, dtype="BOOL", size=n), O.OVec(n, vi, vv, "BOOL")
    S = getattr(gb.semiring, sr)
    q(~vis.S, replace=True) << A.mxv(q, S)
    same_vec(q, O.mxv(oa, oq, sr, w=oq, mask=ovis, mask_comp=True, mask_struct=True, replace=True))
    # mask aliased with the output
    wi, wv = rand_vec(rng, n, 0.5, tname)
    w, ow = gb.Vector.from_coo(wi, wv, dtype=tname, size=n), O.OVec(n, wi, wv, tname)
    xi, xv = rand_vec(rng, n, 0.7, tname)
    x, ox = gb.Vector.from_coo(xi, xv, dtype=tname, size=n), O.OVec(n, xi, xv, tname)
    w(w.S, replace=bool(seed & 1)) << A.mxv(x, S)
    same_vec(w, O.mxv(oa, ox, sr, w=ow, mask=ow, mask_struct=True, replace=bool(seed & 1)))


@pytest.mark.parametrize("seed", range(8))
def test_mixed_types(gb, seed):
    """Operand / output types that differ from the semiring type (typecasts, unfused write rule)."""
    rng = np.random.default_rng(300 + seed)
    ta, tu, tw = [("INT8", "UINT16", "INT64"), ("FP32", "INT32", "FP64"), ("INT64", "INT64", "FP32"),
                  ("BOOL", "INT32", "INT32"), ("INT32", "FP64", "INT16"), ("UINT8", "UINT8", "FP64"),
                  ("INT64", "FP32", "INT64"), ("INT16", "INT16", "BOOL")][seed]
    m, n = int(rng.integers(1, 800)), int(rng.integers(1, 800))
    r, c, v = rand_coo(rng, m, n, ta, long_rows=1)
    ui, uv = rand_vec(rng, n, 0.6, tu)
    wi, wv = rand_vec(rng, m, 0.4, tw)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    sr = ["plus_times", "min_plus", "max_plus", "plus_plus"][seed % 4]
    accum = [None, "plus"][seed % 2]
    exp = O.mxv(O.OMat.from_coo(r, c, v, m, n, ta), O.OVec(n, ui, uv, tu), sr, w=O.OVec(m, wi, wv, tw),
                mask=O.OVec(m, mi, mv, "BOOL"), accum=accum)
    A = gb.Matrix.from_coo(r, c, v, dtype=ta, nrows=m, ncols=n)
    u = gb.Vector.from_coo(ui, uv, dtype=tu, size=n)
    w = gb.Vector.from_coo(wi, wv, dtype=tw, size=m)
    mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
    w(mk.V, accum=accum) << A.mxv(u, getattr(gb.semiring, sr))
    same_vec(w, exp)


def test_edge_shapes(gb):
    S = gb.semiring
    # completely empty matrix: result is empty, and replaces the old content of w when unmasked
    A = gb.Matrix(int, 5, 4)
    u = gb.Vector.from_coo([0, 1], [1, 2], size=4)
    w = gb.Vector.from_coo([1, 2], [7, 8], size=5)
    w << A.mxv(u, S.plus_times)
    assert w.nvals == 0
    # empty u
    A = gb.Matrix.from_coo([0, 1], [1, 0], [3, 4], nrows=2, ncols=2)
    w = A.mxv(gb.Vector(int, 2), S.min_plus).new()
    assert w.nvals == 0
    # 1x1
    A = gb.Matrix.from_coo([0], [0], [5])
    w = A.mxv(gb.Vector.from_coo([0], [6]), S.plus_times).new()
    assert w.to_coo()[1].tolist() == [30]
    # a single row longer than several tiles, everything else empty
    n = 9000
    A = gb.Matrix.from_coo(np.full(n, 3), np.arange(n), np.ones(n, np.int64), nrows=7, ncols=n)
    w = A.mxv(gb.Vector.from_dense(np.arange(n)), S.plus_times).new()
    assert w.to_coo()[0].tolist() == [3] and w.to_coo()[1].tolist() == [n * (n - 1) // 2]
    w = A.mxv(gb.Vector.from_dense(np.arange(n) + 5), S.min_plus).new()
    assert w.to_coo()[1].tolist() == [6]
    # many empty rows around few entries (tiles made only of row ends)
    m = 20000
    A = gb.Matrix.from_coo([5, m - 1], [0, 1], [2.0, 3.0], nrows=m, ncols=2)
    w = gb.Vector.from_dense(np.ones(m))
    w(accum=gb.binary.plus) << A.mxv(gb.Vector.from_dense(np.array([10.0, 20.0])), S.plus_times)
    d = w.to_dense()
    assert d[5] == 21.0 and d[m - 1] == 61.0 and d.sum() == m + 20 + 60


def same_mat(got, exp):
    I, J, X = got.to_coo()
    er, ec, ev = exp.to_coo()
    assert got.shape == (exp.nrows, exp.ncols)
    assert I.tolist() == er.tolist() and J.tolist() == ec.tolist()
    assert X.tolist() == ev.tolist()


@pytest.mark.parametrize("seed", range(24))
def test_mxm_random(gb, seed):
    """Hash SpGEMM (all LDS table sizes + the dense-accumulator path for hub rows) and the matrix write rule."""
    rng = np.random.default_rng(500 + seed)
    tname = TYPES[seed % 7]
    srs = semirings_for(tname)
    sr = srs[rng.integers(len(srs))]
    m, k, n = (int(x) for x in rng.integers(1, 400, 3))
    if seed % 6 == 0:
        m, k, n = (int(x) for x in rng.integers(1, 12, 3))
    big = seed % 4 == 1  # rows whose product exceeds the LDS tables (ub > 16384, nnz(T_i) > 4096)
    if big:
        m, k, n = 40, 700, 9000
        if seed % 8 == 5:  # more than 2^18 columns: the 1024-thread whole-row bitmap kernel and 19 numeric windows
            n = 300000
    ar, ac, av = rand_coo(rng, m, k, tname, long_rows=2 if big else int(rng.integers(0, 2)))
    br, bc, bv = rand_coo(rng, k, n, tname, long_rows=int(rng.integers(0, 3)))
    if big:  # dense-ish B rows so that one A row yields thousands of distinct columns
        deg = rng.integers(20, 60, k)
        br = np.repeat(np.arange(k), deg)
        bc = np.concatenate([rng.choice(n, d, replace=False) for d in deg])
        bv = rand_vals(rng, br.size, tname)
    cr, cc, cv = rand_coo(rng, m, n, tname)
    mr, mc, mv = rand_coo(rng, m, n, "INT8", long_rows=1)
    use_mask = seed % 3 != 0
    comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
    accum = [None, "plus", "min", "second"][rng.integers(4)]
    use_c = bool(rng.integers(2)) or accum is not None
    oa, ob = O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname)
    oc = O.OMat.from_coo(cr, cc, cv, m, n, tname) if use_c else None
    om = O.OMat.from_coo(mr, mc, mv, m, n, "INT8")
    exp = O.mxm(oa, ob, sr, C=oc, mask=om if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct,
                accum=accum, replace=repl and use_mask)
    A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
    C = gb.Matrix.from_coo(cr, cc, cv, dtype=tname, nrows=m, ncols=n) if use_c else gb.Matrix(tname, m, n)
    kw = {}
    if use_mask:
        M = gb.Matrix.from_coo(mr, mc, mv, dtype="INT8", nrows=m, ncols=n)
        mm = M.S if struct else M.V
        kw = dict(mask=~mm if comp else mm, replace=repl)
    if accum:
        kw["accum"] = accum
    C(**kw) << A.mxm(B, getattr(gb.semiring, sr))
    same_mat(C, exp)


@pytest.mark.parametrize("seed", range(16))
def test_mxm_mask_driven(gb, seed):
    """Mask-driven SpGEMM (forced on): non-complemented structural and valued masks, every hash table size of the mask
    rows, mask rows longer than 4096 entries (LDS column windows, one and many windows), accumulators, replace, empty
    rows of A / of the mask, C aliased with the mask."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(1500 + seed)
    tname = TYPES[seed % 7]
    srs = semirings_for(tname)
    sr = srs[rng.integers(len(srs))]
    m, k, n = (int(x) for x in rng.integers(1, 300, 3))
    heavy = seed % 4 == 1
    if heavy:
        m, k, n = 30, 500, (9000 if seed % 8 == 1 else 200000)
    ar, ac, av = rand_coo(rng, m, k, tname, long_rows=2 if heavy else int(rng.integers(0, 2)))
    br, bc, bv = rand_coo(rng, k, n, tname, long_rows=int(rng.integers(0, 3)))
    if heavy:
        deg = rng.integers(20, 60, k)
        br = np.repeat(np.arange(k), deg)
        bc = np.concatenate([rng.choice(n, d, replace=False) for d in deg])
        bv = rand_vals(rng, br.size, tname)
    mr, mc, mv = rand_coo(rng, m, n, "INT8", long_rows=2)
    if heavy:  # mask rows of every size class: 5000+ entries (windows), ~2000, ~500, ~50 and empty ones
        lens = [min(n, x) for x in (6000, 5000, 2000, 500, 50, 0, 1)]
        rows_ = rng.choice(m, len(lens), replace=False)
        mr = np.concatenate([np.full(ln, r) for r, ln in zip(rows_, lens)])
        mc = np.concatenate([rng.choice(n, ln, replace=False) for ln in lens])
        mv = rng.integers(0, 2, mr.size).astype(np.int8)
    cr, cc, cv = rand_coo(rng, m, n, tname)
    struct, repl = bool(seed & 1), bool(seed & 2)
    accum = [None, "plus", "min", "second"][seed % 4] if seed % 3 else None
    use_c = bool(rng.integers(2)) or accum is not None
    oa, ob = O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname)
    oc = O.OMat.from_coo(cr, cc, cv, m, n, tname) if use_c else None
    om = O.OMat.from_coo(mr, mc, mv, m, n, "INT8")
    exp = O.mxm(oa, ob, sr, C=oc, mask=om, mask_struct=struct, accum=accum, replace=repl)
    try:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 2)
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 0 if seed % 2 else 64 << 20)  # (row, window) units / the row kernels
        A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
        B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
        C = gb.Matrix.from_coo(cr, cc, cv, dtype=tname, nrows=m, ncols=n) if use_c else gb.Matrix(tname, m, n)
        M = gb.Matrix.from_coo(mr, mc, mv, dtype="INT8", nrows=m, ncols=n)
        kw = dict(mask=M.S if struct else M.V, replace=repl)
        if accum:
            kw["accum"] = accum
        C(**kw) << A.mxm(B, getattr(gb.semiring, sr))
        assert device.last_stats()["method"] == 4  # the mask-driven path really ran
        same_mat(C, exp)
        if m == n and not accum:  # the mask is the output itself (C<C.S> = A B)
            D = gb.Matrix.from_coo(mr, mc, mv.astype(np.int64), dtype="INT64", nrows=m, ncols=n)
            od = O.OMat.from_coo(mr, mc, mv.astype(np.int64), m, n, "INT64")
            exp2 = O.mxm(O.OMat.from_coo(ar, ac, av, m, k, tname), ob, sr, C=od, mask=od, mask_struct=True)
            D(D.S) << A.mxm(B, getattr(gb.semiring, sr))
            same_mat(D, exp2)
    finally:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 1)
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 64 << 20)


@pytest.mark.parametrize("seed", range(6))
def test_mxm_transposes_and_types(gb, seed):
    rng = np.random.default_rng(600 + seed)
    ta, tb, tc = [("INT64", "INT64", "INT64"), ("INT8", "UINT16", "INT64"), ("FP32", "INT32", "FP64"),
                  ("BOOL", "BOOL", "INT32"), ("INT32", "INT32", "FP32"), ("FP64", "FP64", "FP64")][seed]
    sr = ["plus_times", "min_plus", "max_plus", "lor_land", "plus_pair", "min_second"][seed]
    m, k, n = (int(x) for x in rng.integers(1, 200, 3))
    ta_, tb_ = bool(seed & 1), bool(seed & 2)
    ar, ac, av = rand_coo(rng, *((k, m) if ta_ else (m, k)), ta, long_rows=1)
    br, bc, bv = rand_coo(rng, *((n, k) if tb_ else (k, n)), tb, long_rows=1)
    oa = O.OMat.from_coo(ar, ac, av, *((k, m) if ta_ else (m, k)), ta)
    ob = O.OMat.from_coo(br, bc, bv, *((n, k) if tb_ else (k, n)), tb)
    exp = O.mxm(oa, ob, sr, out_type=tc, transpose_a=ta_, transpose_b=tb_)
    A = gb.Matrix.from_coo(ar, ac, av, dtype=ta, nrows=oa.nrows, ncols=oa.ncols)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tb, nrows=ob.nrows, ncols=ob.ncols)
    C = (A.T if ta_ else A).mxm(B.T if tb_ else B, getattr(gb.semiring, sr)).new(dtype=tc)
    assert C.dtype == tc
    same_mat(C, exp)


@pytest.mark.parametrize("seed", range(12))
def test_hot_column_table(gb, seed):
    """Force the hot-column table of the pull SpMV on (tiny thresholds) and compare with the oracle:
    skewed column degrees, bitmap and full u, masks, vxm over the (separately analysed) transpose."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(700 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[seed % 4]
    m, n = int(rng.integers(200, 1500)), int(rng.integers(300, 2000))
    # power-law-ish columns: most entries land in a few columns, labels scrambled
    nnz = int(rng.integers(2000, 20000))
    hot = rng.permutation(n)[: max(4, n // 20)]
    cols = np.where(rng.random(nnz) < 0.8, hot[rng.integers(0, hot.size, nnz)], rng.integers(0, n, nnz))
    rows = rng.integers(0, m, nnz)
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    vals = rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.6, 0.1][seed % 3], tname)
    wi, wv = rand_vec(rng, m, 0.4, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    accum = [None, "plus", "min"][seed % 3]
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"),
                mask_comp=bool(seed & 1), accum=accum)
    xi, xv = rand_vec(rng, m, 0.5, tname)
    exp_t = O.vxm(O.OVec(m, xi, xv, tname), oa, sr)
    try:
        _lib.lib.GrX_option_set(b"hot_min_cols", 8)
        _lib.lib.GrX_option_set(b"hot_k", 64)
        # even seeds: vectors carry the front pad, the table is gathered in front of u's own values (no copy of u)
        _lib.lib.GrX_option_set(b"vec_pad_min_bytes", 0 if seed % 2 == 0 else 1 << 20)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w(~mk.V if seed & 1 else mk.V, accum=accum) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w, exp)
        ui2, uv2 = u.to_coo()
        assert np.array_equal(ui2, np.sort(ui)) and u.nvals == len(ui)  # u itself is untouched by the table in its pad
        from graphblas_amd import device
        needs_x = not (sr.endswith(("_pair", "_first")) and seed % 3 == 0)  # pair/first on a full u never reads x
        assert device.last_stats()["hot_k"] == (64 if needs_x else 0)  # the table really was in use
        # second call reuses the cached table
        w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        w2(~mk.V if seed & 1 else mk.V, accum=accum) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w2, exp)
        x = gb.Vector.from_coo(xi, xv, dtype=tname, size=m)
        same_vec(x.vxm(A, getattr(gb.semiring, sr)).new(), exp_t)
        # in place (w aliased with u) over a square matrix with the table
        k = min(m, n)
        sel = (rows < k) & (cols < k)
        B = gb.Matrix.from_coo(rows[sel], cols[sel], vals[sel], dtype=tname, nrows=k, ncols=k)
        ob = O.OMat.from_coo(rows[sel], cols[sel], vals[sel], k, k, tname)
        qi, qv = rand_vec(rng, k, 0.5, tname)
        q, oq = gb.Vector.from_coo(qi, qv, dtype=tname, size=k), O.OVec(k, qi, qv, tname)
        q(accum=getattr(gb.binary, accum) if accum else None) << B.mxv(q, getattr(gb.semiring, sr))
        same_vec(q, O.mxv(ob, oq, sr, w=oq, accum=accum))
    finally:
        _lib.lib.GrX_option_set(b"hot_min_cols", 1 << 20)
        _lib.lib.GrX_option_set(b"hot_k", 0)
        _lib.lib.GrX_option_set(b"vec_pad_min_bytes", 1 << 20)


@pytest.mark.parametrize("seed", range(16))
def test_push_direction(gb, seed):
    """Push-direction SpMSpV (forced on): vxm over A's own rows, mxv over the cached transpose; sparse and dense u,
    masks, accumulators, w aliased with u (BFS style), typecasts."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(900 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[seed % 4]
    m, n = int(rng.integers(1, 1200)), int(rng.integers(1, 1200))
    r, c, v = rand_coo(rng, m, n, tname, long_rows=2)
    oa = O.OMat.from_coo(r, c, v, m, n, tname)
    ui, uv = rand_vec(rng, m, [0.02, 0.3, 1.0][seed % 3], tname)
    wi, wv = rand_vec(rng, n, 0.4, tname)
    mi, mv = rand_vec(rng, n, 0.5, "INT8")
    use_mask = seed % 4 != 0
    comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
    accum = [None, "plus", "min", "second"][rng.integers(4)]
    exp = O.vxm(O.OVec(m, ui, uv, tname), oa, sr, w=O.OVec(n, wi, wv, tname), mask=O.OVec(n, mi, mv, "INT8") if use_mask else None,
                mask_comp=comp and use_mask, mask_struct=struct, accum=accum, replace=repl and use_mask)
    xi, xv = rand_vec(rng, n, 0.05, tname)
    exp_mxv = O.mxv(oa, O.OVec(n, xi, xv, tname), sr)
    try:
        _lib.lib.GrX_option_set(b"push_mode", 2)
        A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=m)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        kw = {}
        if use_mask:
            mk = gb.Vector.from_coo(mi, mv, dtype="INT8", size=n)
            mm = mk.S if struct else mk.V
            kw = dict(mask=~mm if comp else mm, replace=repl)
        if accum:
            kw["accum"] = accum
        w(**kw) << u.vxm(A, getattr(gb.semiring, sr))
        assert device.last_stats()["method"] == (2 if ui.size and r.size else 6)  # (an empty operand: the write rule alone)
        same_vec(w, exp)
        # mxv pushes over the transpose once it is cached
        x = gb.Vector.from_coo(xi, xv, dtype=tname, size=n)
        device.cache_transpose(A)
        y = A.mxv(x, getattr(gb.semiring, sr)).new()
        assert device.last_stats()["method"] == (2 if xi.size and r.size else 6)
        same_vec(y, exp_mxv)
        if m == n:
            pass
        # BFS-style aliasing on a square matrix
        k = min(m, n)
        r2, c2, v2 = rand_coo(rng, k, k, tname, long_rows=1)
        B = gb.Matrix.from_coo(r2, c2, v2, dtype=tname, nrows=k, ncols=k)
        ob = O.OMat.from_coo(r2, c2, v2, k, k, tname)
        qi, qv = rand_vec(rng, k, 0.1, tname)
        vi, vv = rand_vec(rng, k, 0.5, "BOOL")
        q, oq = gb.Vector.from_coo(qi, qv, dtype=tname, size=k), O.OVec(k, qi, qv, tname)
        vis, ovis = gb.Vector.from_coo(vi, vv, dtype="BOOL", size=k), O.OVec(k, vi, vv, "BOOL")
        q(~vis.S, replace=True) << q.vxm(B, getattr(gb.semiring, sr))
        same_vec(q, O.vxm(oq, ob, sr, w=oq, mask=ovis, mask_comp=True, mask_struct=True, replace=True))
    finally:
        _lib.lib.GrX_option_set(b"push_mode", 1)


@pytest.mark.parametrize("seed", range(16))
def test_long_short_row_split(gb, seed):
    """Force the long/short row split of the pull SpMV on (tiny thresholds), with and without the hot-column table:
    rows of every length around the threshold and the chunk size, masks that switch long rows off, accumulators,
    bitmap and full u, w aliased with u, vxm over the transpose (its own split)."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(1100 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[seed % 4]
    m, n = int(rng.integers(50, 600)), int(rng.integers(2100, 6000))
    deg = rng.integers(0, 6, m)
    deg[rng.random(m) < 0.3] = 0
    for ln in (7, 8, 9, 63, 64, 65, 511, 513, 2047, 2048, 2049, int(rng.integers(2100, n))):
        deg[rng.integers(0, m)] = min(ln, n)
    rows = np.repeat(np.arange(m), deg)
    cols = np.concatenate([rng.choice(n, d, replace=False) for d in deg])
    vals = rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.5, 0.05][seed % 3], tname)
    wi, wv = rand_vec(rng, m, 0.5, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    accum = [None, "plus", "min"][seed % 3]
    comp = bool(seed & 1)
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"),
                mask_comp=comp, accum=accum, replace=bool(seed & 2))
    xi, xv = rand_vec(rng, m, 0.6, tname)
    exp_t = O.vxm(O.OVec(m, xi, xv, tname), oa, sr)
    try:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1)
        _lib.lib.GrX_option_set(b"split_min_len", 8)
        _lib.lib.GrX_option_set(b"push_mode", 0)
        # merge-path kernel / row-group kernel / sliced-ELLPACK kernel for the short rows
        # sliced ELLPACK / persistent row groups with an LDS head / merge path / row groups
        # (seeds 0, 6, 12: a lane per row over entries staged in LDS -- 4-byte types with a full operand; the rest falls back to row groups)
        # (5: tagged row groups -- the row of every entry stored with it)
        _lib.lib.GrX_option_set(b"short_kernel", [1, 5, 1, 5, 6, 5, 1, 5, 0, 0, 5, 1, 6, 5, 5, 1][seed])
        if seed & 4:
            _lib.lib.GrX_option_set(b"hot_min_cols", 8)
            _lib.lib.GrX_option_set(b"hot_k", 64)
        _lib.lib.GrX_option_set(b"long_sub", 1 + seed % 5)
        _lib.lib.GrX_option_set(b"long_sub_min_len", 64 if seed & 1 else 8)
        # long rows: by matrix type (strips, items for BOOL) / strips for every type incl. BOOL / items for every type
        # ... / hot + cold strips for every type / by type with the hot + cold strips
        forced_long = [DEFAULT_LONG_KERNEL, 2, 1, 4, 3][seed % 5]
        _lib.lib.GrX_option_set(b"long_kernel", forced_long)
        _lib.lib.GrX_option_set(b"long_classes", [16, 8, 32, 16, 64][seed % 5])
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w(~mk.V if comp else mk.V, accum=accum, replace=bool(seed & 2)) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()  # init + long rows + short rows (with the write rule of every row); PAIR over a full u reads no rows
        assert st["kernel_launches"] >= 3 or st["method"] == 5
        # (5 = by type and size: a matrix this small gets the mixed class strips; the hot / cold strips start at lean_min_nnz entries)
        want = {3: 1 if tname == "BOOL" else 2, 5: 1 if tname == "BOOL" else 2}.get(forced_long, forced_long)
        assert st["method"] == 5 or (st["long_kernel"] == want and st["long_entries"] > 0)
        same_vec(w, exp)
        # a product that needs A's values in another type cannot take the split (whose re-coded column copy of the whole matrix
        # was released when the split was built): it runs on the plain arrays -- and the split serves the next same-typed call
        other = "FP64" if tname not in ("FP64", "BOOL") else "INT32"
        u2 = gb.Vector.from_coo(ui, uv.astype(O.NP_OF[other]), dtype=other, size=n)
        sr2 = "plus_times" if tname != "BOOL" else "plus_plus"
        got2 = A.mxv(u2, getattr(gb.semiring, sr2)).new()
        exp2 = O.mxv(oa, O.OVec(n, ui, uv.astype(O.NP_OF[other]), other), sr2)
        gi2, gv2 = got2.to_coo()
        assert gi2.tolist() == exp2.idx.tolist() and np.allclose(gv2.astype(np.float64), exp2.vals.astype(np.float64), rtol=1e-6)
        w3 = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        w3(~mk.V if comp else mk.V, accum=accum, replace=bool(seed & 2)) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w3, exp)
        x = gb.Vector.from_coo(xi, xv, dtype=tname, size=m)
        same_vec(x.vxm(A, getattr(gb.semiring, sr)).new(), exp_t)
        # square, w aliased with u
        k = m
        r2 = np.repeat(np.arange(k), np.minimum(deg, k))
        c2 = np.concatenate([rng.choice(k, min(d, k), replace=False) for d in deg])
        v2 = rand_vals(rng, r2.size, tname)
        B = gb.Matrix.from_coo(r2, c2, v2, dtype=tname, nrows=k, ncols=k)
        ob = O.OMat.from_coo(r2, c2, v2, k, k, tname)
        qi, qv = rand_vec(rng, k, 0.3, tname)
        q, oq = gb.Vector.from_coo(qi, qv, dtype=tname, size=k), O.OVec(k, qi, qv, tname)
        q(~mk.S, replace=True) << B.mxv(q, getattr(gb.semiring, sr))
        same_vec(q, O.mxv(ob, oq, sr, w=oq, mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=True, mask_struct=True, replace=True))
    finally:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1 << 22)
        _lib.lib.GrX_option_set(b"split_min_len", 0)
        _lib.lib.GrX_option_set(b"push_mode", 1)
        _lib.lib.GrX_option_set(b"short_kernel", DEFAULT_SHORT_KERNEL)
        _lib.lib.GrX_option_set(b"hot_min_cols", 1 << 20)
        _lib.lib.GrX_option_set(b"hot_k", 0)
        _lib.lib.GrX_option_set(b"long_sub", 0)
        _lib.lib.GrX_option_set(b"long_sub_min_len", 0)
        _lib.lib.GrX_option_set(b"long_kernel", DEFAULT_LONG_KERNEL)
        _lib.lib.GrX_option_set(b"long_classes", 16)


@pytest.mark.parametrize("seed", range(21))
def test_hot_cold_strips(gb, seed):
    """long_kernel = 4: the long rows as HOT strips (entries whose column code is LDS-resident in its class: lane records of 8
    16-bit LDS slots + 8 values, 32-bit slots for BOOL) plus COLD strips (image gathers, 8 contiguous column ranges).  Skewed
    columns so that both parts carry entries, every type, table sizes from a few lines to everything, 8 .. 64 classes, full and
    sparse operands (the presence lookups of resident codes go to the image), masks that switch long rows off, accumulators,
    replace, iso matrices, vxm over the transpose -- against the oracle."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(8800 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[(seed // 7) % 4]
    m, n = int(rng.integers(60, 400)), int(rng.integers(2100, 5000))
    deg = rng.integers(0, 6, m)
    deg[rng.random(m) < 0.3] = 0
    for ln in (8, 9, 15, 16, 17, 63, 64, 65, 511, 513, 1030, int(rng.integers(1500, 2000))):
        deg[rng.integers(0, m)] = ln
    hot = rng.permutation(n)[: n // 10]
    rows = np.repeat(np.arange(m), deg)
    cols = np.where(rng.random(rows.size) < 0.75, hot[rng.integers(0, hot.size, rows.size)], rng.integers(0, n, rows.size))
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    iso = seed % 5 == 4
    vals = np.full(rows.size, rand_vals(rng, 1, tname)[0]) if iso else rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.5, 0.05][seed % 3], tname)
    wi, wv = rand_vec(rng, m, 0.5, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    accum = [None, "plus", "min"][seed % 3] if tname != "BOOL" else [None, "lor", "land"][seed % 3]
    comp, repl = bool(seed & 1), bool(seed & 2)
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"),
                mask_comp=comp, accum=accum, replace=repl)
    exp_plain = O.mxv(oa, O.OVec(n, ui, uv, tname), sr)
    xi, xv = rand_vec(rng, m, 0.6, tname)
    exp_t = O.vxm(O.OVec(m, xi, xv, tname), oa, sr)
    try:
        for name, val in ((b"split_min_nnz", 1), (b"split_min_len", 8), (b"push_mode", 0), (b"hot_min_cols", 8),
                          (b"hot_k", [64, 256, 2048, 1 << 20][seed % 4]), (b"long_kernel", 4), (b"long_classes", [16, 8, 32, 64][(seed // 2) % 4]),
                          (b"short_kernel", 5 if seed % 3 else DEFAULT_SHORT_KERNEL),
                          (b"vec_pad_min_bytes", 0 if seed % 2 else 1 << 20)):
            _lib.lib.GrX_option_set(name, val)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w(~mk.V if comp else mk.V, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        # (a 64-entry table covers too few references to be built: those seeds run the cold strips alone)
        assert st["method"] == 5 or (st["long_kernel"] == 4 and st["long_entries"] > 0 and (st["hot_k"] > 0 or seed % 4 == 0)), st
        same_vec(w, exp)
        same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), exp_plain)
        x = gb.Vector.from_coo(xi, xv, dtype=tname, size=m)
        same_vec(x.vxm(A, getattr(gb.semiring, sr)).new(), exp_t)
        if tname != "BOOL":
            # a FULL operand under the semirings the fast hot-strip kernel takes for 4- and 8-byte types (k_mxv_hstrip: padding by an
            # absorbing LDS word, scalar scan masks, byte mask probes), with and without a mask that switches long rows off
            fv = rand_vals(rng, n, tname)
            uf, ouf = gb.Vector.from_coo(np.arange(n), fv, dtype=tname, size=n), O.OVec(n, np.arange(n), fv, tname)
            for sr2 in ("min_plus", "max_plus", "plus_times"):
                same_vec(A.mxv(uf, getattr(gb.semiring, sr2)).new(), O.mxv(oa, ouf, sr2))
                w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
                w2(~mk.S if comp else mk.S, accum=accum, replace=repl) << A.mxv(uf, getattr(gb.semiring, sr2))
                same_vec(w2, O.mxv(oa, ouf, sr2, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=comp, mask_struct=True,
                                   accum=accum, replace=repl))
    finally:
        for name, val in ((b"split_min_nnz", 1 << 22), (b"split_min_len", 0), (b"push_mode", 1), (b"hot_min_cols", 1 << 20), (b"hot_k", 0),
                          (b"long_kernel", DEFAULT_LONG_KERNEL), (b"long_classes", 16), (b"vec_pad_min_bytes", 1 << 20),
                          (b"short_kernel", DEFAULT_SHORT_KERNEL)):
            _lib.lib.GrX_option_set(name, val)


@pytest.mark.parametrize("seed", range(4))
def test_cold_tiles_many_long_rows(gb, seed):
    """The cold entries of a hot / cold layout as tagged tiles (k_mxv_ctile) with MORE long rows than one tile holds (8192): the
    threshold at 2 entries makes every non-empty row long, so tiles of several row blocks and column ranges carry entries; a small
    hot table keeps most entries cold.  Full and sparse operands, masks, an accumulator -- against the oracle."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(9900 + seed)
    tname = ["FP32", "INT64", "BOOL", "FP64"][seed]
    sr = {"FP32": "min_plus", "INT64": "plus_times", "BOOL": "lor_land", "FP64": "max_plus"}[tname]
    m, n = 21000, 3000
    deg = rng.integers(2, 5, m)
    deg[rng.random(m) < 0.1] = 0
    hot = rng.permutation(n)[: n // 10]
    rows = np.repeat(np.arange(m), deg)
    cols = np.where(rng.random(rows.size) < 0.5, hot[rng.integers(0, hot.size, rows.size)], rng.integers(0, n, rows.size))
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    vals = rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.5][seed & 1], tname)
    wi, wv = rand_vec(rng, m, 0.5, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    accum = None if tname == "BOOL" else "min"
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=True,
                mask_struct=True, accum=accum)
    exp_plain = O.mxv(oa, O.OVec(n, ui, uv, tname), sr)
    try:
        for name, val in ((b"split_min_nnz", 1), (b"split_min_len", 2), (b"push_mode", 0), (b"hot_min_cols", 8), (b"hot_k", 128),
                          (b"long_kernel", 4), (b"long_sub", 1 + seed), (b"vec_pad_min_bytes", 0)):
            _lib.lib.GrX_option_set(name, val)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w(~mk.S, accum=accum) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        assert st["long_kernel"] == 4 and st["long_entries"] > 30000, st
        same_vec(w, exp)
        same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), exp_plain)
    finally:
        for name, val in ((b"split_min_nnz", 1 << 22), (b"split_min_len", 0), (b"push_mode", 1), (b"hot_min_cols", 1 << 20), (b"hot_k", 0),
                          (b"long_kernel", DEFAULT_LONG_KERNEL), (b"long_sub", 0), (b"vec_pad_min_bytes", 1 << 20)):
            _lib.lib.GrX_option_set(name, val)


@pytest.mark.parametrize("seed", range(4))
def test_split_survives_mixed_calls_and_option_changes(gb, seed):
    """A product builds the hot-coded split and releases the re-coded column copy; then a row reduction (a call that reads no
    column), the same product again, a product that gathers with the ORIGINAL column indices (a typecast of the values cannot take
    the split), and a change of the class count (a rebuild in the hot coding, which needs the re-coded columns back): none of
    them may rebuild from a released array, and the reduction must not throw the hot-coded split away."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(4400 + seed)
    tname = ["FP32", "INT64", "FP64", "INT32"][seed]
    m, n = int(rng.integers(80, 300)), int(rng.integers(2100, 4000))
    deg = rng.integers(0, 6, m)
    for ln in (9, 64, 65, 513, 2049):
        deg[rng.integers(0, m)] = ln
    # skewed columns (the hot-column table is only built for those): most entries land in a twentieth of the columns
    hot = rng.permutation(n)[: n // 20]
    rows = np.repeat(np.arange(m), deg)
    cols = np.where(rng.random(rows.size) < 0.8, hot[rng.integers(0, hot.size, rows.size)], rng.integers(0, n, rows.size))
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    deg = np.bincount(rows, minlength=m)
    vals = rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.5][seed & 1], tname)
    oa, ou = O.OMat.from_coo(rows, cols, vals, m, n, tname), O.OVec(n, ui, uv, tname)
    exp = O.mxv(oa, ou, "min_plus")
    try:
        for name, val in ((b"split_min_nnz", 1), (b"split_min_len", 8), (b"push_mode", 0), (b"hot_min_cols", 8), (b"hot_k", 64),
                          (b"long_kernel", 2), (b"long_classes", 16), (b"vec_pad_min_bytes", 0)):
            _lib.lib.GrX_option_set(name, val)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        same_vec(A.mxv(u, gb.semiring.min_plus).new(), exp)
        st = device.last_stats()
        assert st["hot_k"] > 0 and st["long_kernel"] == 2 and st["long_entries"] > 0
        red = A.reduce_rowwise(gb.monoid.plus).new()
        assert device.last_stats()["long_kernel"] == 2  # (took the hot-coded split as it is)
        ri, rv = red.to_coo()
        want = np.zeros(m, vals.dtype)
        np.add.at(want, rows, vals)
        assert ri.tolist() == np.flatnonzero(deg).tolist() and rv.tolist() == want[deg > 0].tolist()
        same_vec(A.mxv(u, gb.semiring.min_plus).new(), exp)
        assert device.last_stats()["long_kernel"] == 2
        other = "FP64" if tname != "FP64" else "INT32"
        u2 = gb.Vector.from_coo(ui, uv.astype(O.NP_OF[other]), dtype=other, size=n)
        g2 = A.mxv(u2, gb.semiring.plus_times).new()
        e2 = O.mxv(oa, O.OVec(n, ui, uv.astype(O.NP_OF[other]), other), "plus_times")
        gi2, gv2 = g2.to_coo()
        assert gi2.tolist() == e2.idx.tolist() and np.allclose(gv2.astype(np.float64), e2.vals.astype(np.float64), rtol=1e-6)
        for ncls, kernel in ((32, 2), (32, 1), (16, 4), (64, 4), (16, 2)):
            _lib.lib.GrX_option_set(b"long_classes", ncls)
            _lib.lib.GrX_option_set(b"long_kernel", kernel)
            same_vec(A.mxv(u, gb.semiring.min_plus).new(), exp)
            assert device.last_stats()["long_kernel"] == kernel
            red2 = A.reduce_rowwise(gb.monoid.plus).new()
            assert red2.isequal(red)
    finally:
        for name, val in ((b"split_min_nnz", 1 << 22), (b"split_min_len", 0), (b"push_mode", 1), (b"hot_min_cols", 1 << 20), (b"hot_k", 0),
                          (b"long_kernel", DEFAULT_LONG_KERNEL), (b"long_classes", 16), (b"vec_pad_min_bytes", 1 << 20)):
            _lib.lib.GrX_option_set(name, val)


@pytest.mark.parametrize("seed", range(6))
def test_long_rows_many_chunks(gb, seed, request):
    """Every non-empty row is 'long' (threshold 2): each wavefront of the long-row kernel walks many chunks of its
    descriptor batch -- the software pipeline crosses chunk boundaries, skips masked-out chunks, and (seed >= 4) a
    wavefront needs more than one batch of 64 descriptors."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(1300 + seed)
    tname = ["FP32", "INT32", "BOOL", "FP64", "INT64", "UINT8"][seed]
    sr = semirings_for(tname)[seed % 4]
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    m = 700 if seed < 4 else (450000 if on_gpu else 6000)  # seed >= 4: more than 64 chunks per wavefront
    n = 900
    deg = rng.integers(0, 5, m)
    long_at = rng.choice(m, 6, replace=False)
    deg[long_at] = 0
    rows = np.repeat(np.arange(m), deg)
    first = np.cumsum(deg) - deg
    cols = (np.repeat(rng.integers(0, n, m), deg) + 7 * (np.arange(rows.size) - np.repeat(first, deg))) % n  # distinct within a row
    for r, ln in zip(long_at, (513, 1024, 1025, 2048, 2050, 4097)):
        ln = min(ln, n)
        rows = np.concatenate([rows, np.full(ln, r)])
        cols = np.concatenate([cols, rng.choice(n, ln, replace=False)])
    vals = rand_vals(rng, rows.size, tname)
    ui, uv = rand_vec(rng, n, [1.0, 0.4][seed % 2], tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=bool(seed & 1))
    exp_nomask = O.mxv(oa, O.OVec(n, ui, uv, tname), sr)
    try:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1)
        _lib.lib.GrX_option_set(b"split_min_len", 2)
        _lib.lib.GrX_option_set(b"push_mode", 0)
        _lib.lib.GrX_option_set(b"long_kernel", [2, 0, 1, 4, 4, 0][seed])  # class strips / chunk kernel / item kernel / hot + cold strips
        _lib.lib.GrX_option_set(b"long_sub", [2, 1, 4, 3, 16, 1][seed])  # sub-ranges per class of the cold columns
        _lib.lib.GrX_option_set(b"long_sub_min_len", [2, 2, 600, 1025, 2, 2][seed])
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w = gb.Vector(tname, size=m)
        w(~mk.V if seed & 1 else mk.V) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w, exp)
        same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), exp_nomask)
    finally:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1 << 22)
        _lib.lib.GrX_option_set(b"split_min_len", 0)
        _lib.lib.GrX_option_set(b"push_mode", 1)
        _lib.lib.GrX_option_set(b"long_kernel", DEFAULT_LONG_KERNEL)
        _lib.lib.GrX_option_set(b"long_sub", 0)
        _lib.lib.GrX_option_set(b"long_sub_min_len", 0)


@pytest.mark.parametrize("seed", range(22))
def test_vector_assign_reduce_random(gb, seed):
    """Scalar assign over all indices (mask forms, accumulators, replace, a mask aliasing the output) and monoid reduce,
    every type, against the oracle's write rule / numpy folds."""
    rng = np.random.default_rng(1700 + seed)
    tname = ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"][seed % 11]
    n = int(rng.integers(1, 700)) if seed % 5 else [1, 63, 64, 65, 128][seed % 4]
    wi, wv = rand_vec(rng, n, 0.5, tname)
    mi, mv = rand_vec(rng, n, 0.6, "INT8")
    comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
    use_mask = seed % 4 != 0
    accum = [None, "plus", "min", "second", "max"][seed % 5]
    value = rand_vals(rng, 1, tname)[0]
    ow = O.OVec(n, wi, wv, tname)
    om = O.OVec(n, mi, mv, "INT8")
    exp = O.vec_assign_scalar(ow, value, mask=om if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct,
                              accum=accum, replace=repl and use_mask)
    w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
    kw = {}
    if use_mask:
        M = gb.Vector.from_coo(mi, mv, dtype="INT8", size=n)
        mm = M.S if struct else M.V
        kw = dict(mask=~mm if comp else mm, replace=repl)
    if accum:
        kw["accum"] = accum
    w(**kw)[:] << value.item()
    same_vec(w, exp)
    assert w.nvals == exp.idx.size
    # the output as its own (structural) mask
    w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
    w2(w2.S)[:] << value.item()
    same_vec(w2, O.vec_assign_scalar(ow, value, mask=ow, mask_struct=True))
    # reduce
    for mon in (["lor", "land", "lxor"] if tname == "BOOL" else ["plus", "times", "min", "max"]):
        got = w.reduce(getattr(gb.monoid, mon)).new().value
        ref = O.vec_reduce(exp, mon)
        if ref is None:
            assert got is None
        elif tname in ("FP32", "FP64"):
            assert np.isclose(got, ref, rtol=1e-5 if tname == "FP32" else 1e-12) or (np.isinf(got) and np.isinf(ref)) \
                or (np.isnan(got) and np.isnan(ref)), (mon, got, ref)
        else:
            assert got == ref, (mon, got, ref)
    e = gb.Vector(tname, n)
    assert e.reduce(getattr(gb.monoid, "lor" if tname == "BOOL" else "plus")).new().value is None


@pytest.mark.parametrize("seed", range(22))
def test_vector_ewise_random(gb, seed):
    """eWiseAdd / eWiseMult against a numpy restatement (union / intersection, then the oracle's write rule): every type,
    masks, accumulators, replace, output aliased with an input, mixed input types."""
    rng = np.random.default_rng(1900 + seed)
    tname = ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"][seed % 11]
    n = int(rng.integers(1, 500)) if seed % 5 else [1, 64, 65, 128][seed % 4]
    ops = ["lor", "land", "lxor", "first", "second"] if tname == "BOOL" else ["plus", "times", "min", "max", "minus", "first", "second"]
    opname = ops[seed % len(ops)]
    ui, uv = rand_vec(rng, n, 0.5, tname)
    vi, vv = rand_vec(rng, n, 0.5, tname)
    wi, wv = rand_vec(rng, n, 0.4, tname)
    mi, mv = rand_vec(rng, n, 0.6, "INT8")
    np_t = O.NP_OF[tname]

    def dense(i, x):
        h = np.zeros(n, bool); d = np.zeros(n, np_t); h[i] = True; d[i] = x
        return h, d

    hu, du = dense(ui, uv); hv, dv = dense(vi, vv)
    with np.errstate(over="ignore"):
        if tname == "BOOL":
            f = {"lor": np.logical_or, "land": np.logical_and, "lxor": np.logical_xor, "first": lambda a, b: a, "second": lambda a, b: b}[opname]
        else:
            f = {"plus": np.add, "times": np.multiply, "min": np.minimum, "max": np.maximum, "minus": np.subtract,
                 "first": lambda a, b: a, "second": lambda a, b: b}[opname]
        both = np.asarray(f(du, dv)).astype(np_t)
    for is_add in (True, False):
        t_has = (hu | hv) if is_add else (hu & hv)
        t_val = np.where(hu & hv, both, np.where(hu, du, dv)).astype(np_t)
        comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
        use_mask = seed % 3 != 0
        accum = [None, "plus", "min", "second"][seed % 4] if tname != "BOOL" else [None, "lor", "second"][seed % 3]
        ow = O.OVec(n, wi, wv, tname)
        om = O.OVec(n, mi, mv, "INT8")
        # expected: the oracle's write rule with T = (t_has, t_val)
        w_has, w_val = ow.dense()
        mt = O._dense_mask(om if use_mask else None, struct, n)
        import ctypes
        rc = O.lib().grbo_vec_write(O.TYPE_CODES[tname], ctypes.c_int64(n), O._p(w_has), O._p(w_val), O._p(t_has.astype(np.uint8)),
                                    O._p(np.ascontiguousarray(t_val)), O._p(mt), int(comp and use_mask),
                                    O.OP_CODES[accum] if accum else -1, int(repl and use_mask))
        assert rc == 0
        exp = O.OVec.from_dense(w_has, w_val, tname)
        U = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        V = gb.Vector.from_coo(vi, vv, dtype=tname, size=n)
        W = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        kw = {}
        if use_mask:
            M = gb.Vector.from_coo(mi, mv, dtype="INT8", size=n)
            mm = M.S if struct else M.V
            kw = dict(mask=~mm if comp else mm, replace=repl)
        if accum:
            kw["accum"] = accum
        op = getattr(gb.binary, opname)
        W(**kw) << (U.ewise_add(V, op) if is_add else U.ewise_mult(V, op))
        same_vec(W, exp)
    # output aliased with an input: u = u (op) v
    U = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
    V = gb.Vector.from_coo(vi, vv, dtype=tname, size=n)
    U << U.ewise_add(V, getattr(gb.binary, opname))
    t_has = hu | hv
    t_val = np.where(hu & hv, both, np.where(hu, du, dv)).astype(np_t)
    same_vec(U, O.OVec.from_dense(t_has.astype(np.uint8), t_val, tname))


@pytest.mark.parametrize("seed", range(18))
def test_pair_over_full_operand(gb, seed):
    """(monoid, PAIR) with a full operand is computed from the row lengths alone (GrX_Stats.method 5): every monoid, every
    type, all mask forms / accumulators / replace, an output of another type, the output aliasing the operand -- against the
    oracle, and against the general pull kernels (debug flag 65536 turns the shortcut off)."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(4100 + seed)
    tname = TYPES[seed % 7]
    monoids = ["lor", "land", "lxor", "any"] if tname == "BOOL" else ["plus", "min", "max", "any", "times"]
    sr = f"{monoids[seed % len(monoids)]}_pair"
    if not hasattr(gb.semiring, sr) or tname not in getattr(gb.semiring, sr):
        sr = "any_pair"
    m, n = int(rng.integers(1, 2500)), int(rng.integers(1, 2500))
    square = seed % 6 == 5
    if square:
        n = m
    r, c, v = rand_coo(rng, m, n, tname, long_rows=int(rng.integers(0, 3)))
    ui, uv = np.arange(n), rand_vals(rng, n, tname)
    wname = tname if seed % 3 else ["INT64", "FP64", "INT32"][seed % 3]
    wi, wv = rand_vec(rng, m, 0.4, wname)
    mi, mv = rand_vec(rng, m, 0.5, "INT8")
    use_mask = seed % 4 != 0
    comp, struct, repl = (bool(x) for x in rng.integers(0, 2, 3))
    accum = [None, "plus", "min", "second"][rng.integers(4)]
    oa = O.OMat.from_coo(r, c, v, m, n, tname)
    A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=m, ncols=n)
    kw = {}
    if use_mask:
        mk = gb.Vector.from_coo(mi, mv, dtype="INT8", size=m)
        mm = mk.S if struct else mk.V
        kw = dict(mask=~mm if comp else mm, replace=repl)
    if accum:
        kw["accum"] = accum
    okw = dict(mask=O.OVec(m, mi, mv, "INT8") if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct,
               accum=accum, replace=repl and use_mask)
    if square and wname == tname:  # w aliases u
        exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(n, ui, uv, tname), **okw)
    else:
        exp = O.mxv(oa, O.OVec(n, ui, uv, tname), sr, w=O.OVec(m, wi, wv, wname), **okw)
    for flags in (0, 65536):
        _lib.lib.GrX_option_set(b"debug_flags", flags)
        try:
            u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
            w = u if (square and wname == tname) else gb.Vector.from_coo(wi, wv, dtype=wname, size=m)
            w(**kw) << A.mxv(u, getattr(gb.semiring, sr))
            assert device.last_stats()["method"] == (5 if flags == 0 else 1)
            same_vec(w, exp)
            # vxm over the transpose view is the same product
            w2 = gb.Vector.from_coo(wi, wv, dtype=wname, size=m) if not (square and wname == tname) else None
            if w2 is not None:
                w2(**kw) << u.vxm(A.T, getattr(gb.semiring, sr))
                same_vec(w2, exp)
        finally:
            _lib.lib.GrX_option_set(b"debug_flags", 0)


@pytest.mark.parametrize("seed", range(6))
def test_long_rows_terminal_monoids(gb, seed):
    """LOR / LAND / ANY stop early in the long-row kernel: items of a row whose accumulator already holds the terminal value
    (or, for ANY, any product) are dropped before their entries are read.  Rows with several items per class, so that later
    quads find finished rows; checked against the oracle with and without the early exit (debug flag 262144)."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(5200 + seed)
    sr = ["lor_land", "any_pair", "land_lor", "lor_land", "any_pair", "land_land"][seed]
    m, n, rl = 40, 30000, 9000
    rows = np.repeat(np.arange(m), rl)
    cols = np.concatenate([rng.choice(n, rl, replace=False) for _ in range(m)])
    vals = rng.random(rows.size) < (0.5 if "land_l" in sr and sr.startswith("land") else 0.9)
    dens = [0.3, 0.5, 1.0, 0.002, 0.4, 0.7][seed]
    ui = np.flatnonzero(rng.random(n) < dens)
    uv = rng.random(ui.size) < (0.5 if sr.startswith("land") else 0.8)
    mi, mv = rand_vec(rng, m, 0.7, "BOOL")
    oa = O.OMat.from_coo(rows, cols, vals, m, n, "BOOL")
    ou = O.OVec(n, ui, uv, "BOOL")
    exp = O.mxv(oa, ou, sr)
    exp_m = O.mxv(oa, ou, sr, mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=True, mask_struct=True, replace=True)
    try:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1)
        _lib.lib.GrX_option_set(b"push_mode", 0)
        # (the item kernel BOOL matrices get by default / the hot + cold strips / the class strips: each has its own early exit)
        _lib.lib.GrX_option_set(b"long_kernel", [DEFAULT_LONG_KERNEL, 4, 2, DEFAULT_LONG_KERNEL, 4, DEFAULT_LONG_KERNEL][seed])
        if seed in (1, 4):
            _lib.lib.GrX_option_set(b"hot_min_cols", 8)
            _lib.lib.GrX_option_set(b"hot_k", 4096)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype="BOOL", nrows=m, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype="BOOL", size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        for flags in (0, 262144):
            _lib.lib.GrX_option_set(b"debug_flags", flags)
            same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), exp)
            w = gb.Vector("BOOL", size=m)
            w(~mk.S, replace=True) << A.mxv(u, getattr(gb.semiring, sr))
            same_vec(w, exp_m)
    finally:
        _lib.lib.GrX_option_set(b"debug_flags", 0)
        _lib.lib.GrX_option_set(b"split_min_nnz", 1 << 22)
        _lib.lib.GrX_option_set(b"push_mode", 1)
        _lib.lib.GrX_option_set(b"long_kernel", DEFAULT_LONG_KERNEL)
        _lib.lib.GrX_option_set(b"hot_min_cols", 1 << 20)
        _lib.lib.GrX_option_set(b"hot_k", 0)


@pytest.mark.parametrize("seed", range(8))
def test_reductions_over_split_matrices(gb, seed):
    """Row / column reductions and (monoid, FIRST) products with a full operand on matrices with the long / short row split:
    the kernels then read no column index at all (debug flag 524288 turns that off) -- against the oracle."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(6100 + seed)
    tname = TYPES[seed % 7]
    mon = (["lor", "land", "lxor"] if tname == "BOOL" else ["plus", "min", "max", "times"])[seed % (3 if tname == "BOOL" else 4)]
    m, n = int(rng.integers(80, 400)), int(rng.integers(2100, 5000))
    deg = rng.integers(0, 6, m)
    deg[rng.random(m) < 0.3] = 0
    for ln in (8, 9, 63, 65, 511, 1025, 2050, int(rng.integers(1500, n))):
        deg[rng.integers(0, m)] = min(ln, n)
    rows = np.repeat(np.arange(m), deg)
    cols = np.concatenate([rng.choice(n, d, replace=False) for d in deg])
    vals = rand_vals(rng, rows.size, tname)
    if mon == "times":
        vals = (vals % 3).astype(vals.dtype)
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    try:
        _lib.lib.GrX_option_set(b"split_min_nnz", 1)
        _lib.lib.GrX_option_set(b"split_min_len", 8)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        ui, uv = np.arange(n), rand_vals(rng, n, tname)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        for flags in (0, 524288):
            _lib.lib.GrX_option_set(b"debug_flags", flags)
            same_vec(A.reduce_rowwise(getattr(gb.monoid, mon)).new(), O.mat_reduce_rows(oa, mon))
            same_vec(A.reduce_columnwise(getattr(gb.monoid, mon)).new(), O.mat_reduce_rows(oa, mon, columns=True))
            sr = f"{mon}_first"
            if tname != "BOOL" and hasattr(gb.semiring, sr) and tname in getattr(gb.semiring, sr):
                w = gb.Vector(tname, size=m)
                w(mk.V) << A.mxv(u, getattr(gb.semiring, sr))
                same_vec(w, O.mxv(oa, O.OVec(n, ui, uv, tname), sr, mask=O.OVec(m, mi, mv, "BOOL")))
    finally:
        _lib.lib.GrX_option_set(b"debug_flags", 0)
        _lib.lib.GrX_option_set(b"split_min_nnz", 1 << 22)
        _lib.lib.GrX_option_set(b"split_min_len", 0)


@pytest.mark.parametrize("seed", range(4))
def test_mxm_streamed_row_batches(gb, seed, request):
    """GrX_mxm_streamed: the product in row batches under a byte budget (down to one row per batch), only count and value
    checksum kept -- equal to the materialised product's (scipy, exact in INT64 / small-integer FP64)."""
    import ctypes

    import scipy.sparse as sp

    from graphblas_amd import _lib

    rng = np.random.default_rng(9100 + seed)
    tname = ["INT64", "FP64", "INT32", "INT64"][seed]
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    m, k, n = int(rng.integers(40, 200) if on_gpu else rng.integers(12, 30)), int(rng.integers(40, 200)), int(rng.integers(40, 300))
    ar, ac, av = rand_coo(rng, m, k, tname, long_rows=2)
    br, bc, bv = rand_coo(rng, k, n, tname, long_rows=1)
    A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
    Sa = sp.csr_matrix((av.astype(np.int64), (ar, ac)), shape=(m, k))
    Sb = sp.csr_matrix((bv.astype(np.int64), (br, bc)), shape=(k, n))
    pat = (sp.csr_matrix((np.ones(ar.size, np.int64), (ar, ac)), shape=(m, k)) @ sp.csr_matrix((np.ones(br.size, np.int64), (br, bc)), shape=(k, n))).tocsr()
    ref = (Sa @ Sb).tocsr()
    want_sum = int(ref.data.astype(np.int64).sum()) & 0xFFFFFFFFFFFFFFFF
    sr = gb.semiring.plus_times[tname]
    sums = {}
    try:
        for sum_pass in (0, 1):  # (the checksum folded into the numeric kernels' stores -- round 5 -- and by a pass of its own over the product)
            assert _lib.lib.GrX_option_set(b"mxm_checksum_pass", sum_pass) == 0
            for budget in (1, 4096, 1 << 30):
                nv, cs, fl, nb = (ctypes.c_uint64(0) for _ in range(4))
                rc = _lib.lib.GrX_mxm_streamed(sr._carg, A._carg, B._carg, budget, ctypes.byref(nv), ctypes.byref(cs), ctypes.byref(fl), ctypes.byref(nb))
                assert rc == 0
                assert nv.value == pat.nnz
                assert fl.value == int((np.diff(Sb.indptr)[ac]).sum())
                if tname != "INT32":  # (INT32 entries are summed after sign extension: not comparable with the int64 product's sum)
                    assert cs.value == want_sum
                sums.setdefault(budget, []).append(cs.value)
                assert nb.value >= 1 and (budget > 1 or nb.value > m // 4) and (budget < (1 << 30) or nb.value == 1)
        assert all(a == b for a, b in sums.values())
    finally:
        _lib.lib.GrX_option_set(b"mxm_checksum_pass", 0)


@pytest.mark.parametrize("groups", [0, 4])
def test_mxm_streamed_checksum_through_the_unit_kernels(gb, groups, request):
    """The folded checksum on the heavy-row kernels: rows of A with hundreds of entries times a B of nine column windows -- units of
    every compact class, dense units, window groups -- in row batches; count and checksum against scipy."""
    import ctypes

    import scipy.sparse as sp

    from graphblas_amd import _lib

    rng = np.random.default_rng(515)
    on_gpu = request.node.callspec.params["gb"] == "gpu"
    k, n, m = 150, 9 * 16384 - 77, 6
    br, bc = [], []
    for r in range(k):
        d = int(rng.integers(50, 400 if on_gpu else 150))
        cols = np.unique(np.minimum((rng.random(d) ** 2 * n).astype(np.int64), n - 1))  # (the first windows are dense, the last ones thin)
        br.append(np.full(cols.size, r))
        bc.append(cols)
    br, bc = np.concatenate(br), np.concatenate(bc)
    deg = np.array([120, 3, 150, 0, 90, 40])
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg])
    av = rng.integers(1, 4, ar.size).astype(np.int64)
    bv = rng.integers(1, 4, br.size).astype(np.int64)
    ref = (sp.csr_matrix((av, (ar, ac)), shape=(m, k)) @ sp.csr_matrix((bv, (br, bc)), shape=(k, n))).tocsr()
    A = gb.Matrix.from_coo(ar, ac, av, dtype="INT64", nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype="INT64", nrows=k, ncols=n)
    sr = gb.semiring.plus_times["INT64"]
    try:
        for name, val in dict(mxm_unit_min_flops=128, mxm_window_groups=groups, mxm_unit_dense=1500).items():
            assert _lib.lib.GrX_option_set(name.encode(), val) == 0
        for budget in (1 << 16, 1 << 30):
            nv, cs, fl, nb = (ctypes.c_uint64(0) for _ in range(4))
            assert _lib.lib.GrX_mxm_streamed(sr._carg, A._carg, B._carg, budget, ctypes.byref(nv), ctypes.byref(cs), ctypes.byref(fl), ctypes.byref(nb)) == 0
            assert nv.value == ref.nnz and cs.value == int(ref.data.sum())
    finally:
        for name, val in dict(mxm_unit_min_flops=1024, mxm_window_groups=0, mxm_unit_dense=4096).items():
            _lib.lib.GrX_option_set(name.encode(), val)


def test_mxm_wide_heavy_rows(gb):
    """A product with more than 2^20 columns and rows whose multiply count exceeds the LDS hash table: the symbolic pass covers
    the column range in passes of 2^20 columns of LDS bitmap (two passes here), the numeric pass walks column windows."""
    import scipy.sparse as sp

    rng = np.random.default_rng(4242)
    m, k, n = 6, 400, (1 << 20) + 70_000
    deg = np.array([300, 0, 5, 350, 1, 280])
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([rng.choice(k, d, replace=False) for d in deg])
    av = rng.integers(1, 5, ar.size).astype(np.int64)
    bdeg = rng.integers(40, 120, k)
    br = np.repeat(np.arange(k), bdeg)
    bc = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in bdeg])
    bc[:50] = np.sort(rng.choice(np.arange(n - 60_000, n), 50, replace=False))  # make sure the second pass has work
    key = np.unique(br * n + bc)
    br, bc = key // n, key % n
    bv = rng.integers(1, 5, br.size).astype(np.int64)
    A = gb.Matrix.from_coo(ar, ac, av, dtype="INT64", nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype="INT64", nrows=k, ncols=n)
    C = A.mxm(B, gb.semiring.plus_times).new()
    ref = (sp.csr_matrix((av, (ar, ac)), shape=(m, k)) @ sp.csr_matrix((bv, (br, bc)), shape=(k, n))).tocsr()
    ref.sort_indices()
    cp, cj, cx = C.to_csr()
    assert np.array_equal(cp.astype(np.int64), ref.indptr) and np.array_equal(cj.astype(np.int64), ref.indices)
    assert np.array_equal(cx, ref.data)
    assert int(np.diff(ref.indptr).max()) > 16384  # (heavy rows really took the LDS-bitmap / window path)


@pytest.mark.parametrize("sr,tname", [("plus_times", "INT64"), ("min_plus", "FP64"), ("max_second", "INT32"), ("any_pair", "BOOL"), ("plus_pair", "UINT16")])
@pytest.mark.parametrize("pool", [None, 0, 3])
def test_mxm_unit_classes(gb, sr, tname, pool):
    """Heavy rows as (row, column window) units: three windows whose entry counts fall into the three classes of the numeric pass
    -- dense (> 4096 of the window's 16384 columns: one accumulator per column), medium (compact accumulators, four
    wavefronts), small (<= 512: one wavefront) --, light rows, empty rows.  ``pool``: the number of unit bitmaps the symbolic
    pass may keep for the numeric pass (None: no limit; 0: none, the numeric pass recomputes all; 3: the pool runs out)."""
    import scipy.sparse as sp

    from graphblas_amd import _lib

    rng = np.random.default_rng(99)
    m, k, n = 9, 220, 40_000
    np_t = O.NP_OF[tname]
    bc, br = [], []
    for r in range(k):
        cols = np.concatenate([rng.choice(16384, 150, replace=False), 16384 + rng.choice(16384, 40, replace=False),
                               32768 + rng.choice(n - 32768, 3, replace=False)])
        bc.append(np.sort(cols))
        br.append(np.full(cols.size, r))
    br, bc = np.concatenate(br), np.concatenate(bc)
    deg = np.array([100, 0, 25, 3, 180, 1, 60, 0, 26])
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg])
    if tname == "BOOL":
        av, bv = np.ones(ar.size, bool), np.ones(br.size, bool)
    else:
        av, bv = rng.integers(1, 6, ar.size).astype(np_t), rng.integers(1, 6, br.size).astype(np_t)
    A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
    try:
        if pool is not None:
            _lib.lib.GrX_option_set(b"mxm_bitmap_pool_cap", pool)
        C = A.mxm(B, getattr(gb.semiring, sr)).new()
    finally:
        _lib.lib.GrX_option_set(b"mxm_bitmap_pool_cap", (1 << 31) - 1)
    oc = O.mxm(O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname), sr)
    cp, cj, cx = C.to_csr()
    assert np.array_equal(cp.astype(np.int64), oc.indptr) and np.array_equal(cj.astype(np.int64), oc.indices)
    if sr.startswith("any_") and not sr.endswith("pair"):
        return
    assert np.array_equal(cx, oc.values)
    # the classes the rows were built for
    P = (sp.csr_matrix((np.ones(ar.size), (ar, ac)), shape=(m, k)) @ sp.csr_matrix((np.ones(br.size), (br, bc)), shape=(k, n))).tocsr()
    w0 = np.diff(P[:, :16384].tocsr().indptr)
    w1 = np.diff(P[:, 16384:32768].tocsr().indptr)
    w2 = np.diff(P[:, 32768:].tocsr().indptr)
    assert w0[0] > 4096 and 512 < w1[0] <= 4096 and 0 < w2[0] <= 512 and w0[4] > 4096
    assert np.diff(P.indptr)[2] > 4096 and deg[2] * 193 <= 16384


def test_mxm_very_wide(gb):
    """More than 2047 column windows (40 M columns): the window offsets come from the search kernel instead of the LDS histogram;
    heavy rows are still walked as units (2 442 windows each), plain and under a structural mask."""
    import scipy.sparse as sp

    from graphblas_amd import _lib

    rng = np.random.default_rng(4243)
    m, k, n = 4, 50, 40_000_000
    deg = np.array([30, 0, 45, 2])
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg])
    av = rng.integers(1, 5, ar.size).astype(np.int64)
    bdeg = rng.integers(60, 140, k)
    br = np.repeat(np.arange(k), bdeg)
    bc = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in bdeg])
    bv = rng.integers(1, 5, br.size).astype(np.int64)
    A = gb.Matrix.from_coo(ar, ac, av, dtype="INT64", nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype="INT64", nrows=k, ncols=n)
    C = A.mxm(B, gb.semiring.plus_times).new()
    ref = (sp.csr_matrix((av, (ar, ac)), shape=(m, k)) @ sp.csr_matrix((bv, (br, bc)), shape=(k, n))).tocsr()
    ref.sort_indices()
    cp, cj, cx = C.to_csr()
    assert np.array_equal(cp.astype(np.int64), ref.indptr) and np.array_equal(cj.astype(np.int64), ref.indices)
    assert np.array_equal(cx, ref.data) and int(np.diff(ref.indptr).max()) > 2000
    # mask: half of the product's own pattern plus entries elsewhere
    rr = np.repeat(np.arange(m), np.diff(ref.indptr))
    keep = rng.random(rr.size) < 0.5
    mr = np.concatenate([rr[keep], rng.integers(0, m, 500)])
    mc = np.concatenate([ref.indices[keep], rng.integers(0, n, 500)])
    key = np.unique(mr * n + mc)
    mr, mc = key // n, key % n
    M = gb.Matrix.from_coo(mr, mc, np.ones(mr.size, bool), dtype="BOOL", nrows=m, ncols=n)
    try:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 2)
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 0)
        D = A.mxm(B, gb.semiring.plus_times).new(mask=M.S)
    finally:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 1)
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 64 << 20)
    want = ref.multiply(sp.csr_matrix((np.ones(mr.size), (mr, mc)), shape=(m, n)).astype(bool)).tocsr()
    want.sort_indices()
    dp, dj, dx = D.to_csr()
    assert np.array_equal(dp.astype(np.int64), want.indptr) and np.array_equal(dj.astype(np.int64), want.indices)
    assert np.array_equal(dx, want.data.astype(np.int64))


@pytest.mark.parametrize("seed", range(20))
def test_mxm_units_random(gb, seed, request=None):
    on_gpu = request is None or request.node.callspec.params["gb"] == "gpu"  # (scripts/stress_parity.py calls without a request)
    _mxm_units_case(gb, seed, on_gpu)


@pytest.mark.parametrize("groups", [2, 4, 8])
@pytest.mark.parametrize("seed", range(12))
def test_mxm_units_window_groups(gb, seed, groups, request):
    """Round 5: the units walk GROUPS of 2 / 4 / 8 column windows (option mxm_window_groups; by default a function of the width of B).
    The same random cases as test_mxm_units_random on matrices of up to 12 windows: groups that fit the compact classes run as one
    unit, denser ones fall back to their windows; the last group of a row is short; kept and recomputed bitmaps."""
    _mxm_units_case(gb, 300 + seed, request.node.callspec.params["gb"] == "gpu", nwin_hi=13, groups=groups)


def _mxm_units_case(gb, seed, on_gpu, nwin_hi=6, groups=0):
    """(row, column window) units on random shapes: the columns of B drawn from a skewed window distribution (units of every
    class: one wavefront / four wavefronts with 1024 and 4096 accumulators / an accumulator per column, several windows, empty
    windows), rows of A from a few to thousands of entries, every type and semiring of the random tests, iso and valued
    operands; plain, with an accumulator into an existing C, under a structural / valued mask (full product + write rule, and
    mask-driven with the units keyed by the mask row); random class limits and bitmap-pool sizes."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(9100 + seed)
    tname = TYPES[seed % 7]
    srs = semirings_for(tname)
    sr = srs[int(rng.integers(len(srs)))]
    nwin = int(rng.integers(1, nwin_hi))
    n = int(rng.integers((nwin - 1) * 16384 + 1, nwin * 16384 + 1))
    k = int(rng.integers(40, 400 if on_gpu else 120))
    m = int(rng.integers(3, 40 if on_gpu else 10))
    # rows of B: 20 .. 600 entries, windows drawn with skewed weights (some windows get nothing)
    wts = rng.random(nwin) ** 3
    wts[rng.random(nwin) < 0.25] = 0
    if wts.sum() == 0:
        wts[0] = 1
    wts /= wts.sum()
    br, bc = [], []
    for r in range(k):
        d = int(rng.integers(20, 600 if on_gpu else 200))
        win = rng.choice(nwin, d, p=wts)
        cols = np.unique(np.minimum(win * 16384 + rng.integers(0, 16384, d), n - 1))
        br.append(np.full(cols.size, r))
        bc.append(cols)
    br, bc = np.concatenate(br), np.concatenate(bc)
    deg = rng.integers(0, 12, m)
    heavy = rng.random(m) < 0.5
    deg[heavy] = rng.integers(max(1, k // 4), k + 1, int(heavy.sum()))
    deg = np.minimum(deg, k)
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg]) if deg.sum() else np.zeros(0, np.int64)
    iso = seed % 5 == 0
    av = rand_vals(rng, ar.size, tname)
    bv = rand_vals(rng, br.size, tname)
    if iso and av.size and bv.size:
        av[:] = av[0]
        bv[:] = bv[0]
    oa, ob = O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname)
    mode = seed % 4  # 0 plain, 1 accumulate into C, 2 structural mask (mask-driven), 3 valued mask + replace (full product)
    cr, cc, cv = rand_coo(rng, m, n, tname, long_rows=1)
    mr = np.repeat(np.arange(m), rng.integers(0, 3000 if on_gpu else 600, m))
    mc = rng.integers(0, n, mr.size)
    key = np.unique(mr * n + mc)
    mr, mc = key // n, key % n
    mv = rng.integers(0, 2, mr.size).astype(np.int8)
    oc = O.OMat.from_coo(cr, cc, cv, m, n, tname) if mode in (1, 3) else None
    om = O.OMat.from_coo(mr, mc, mv, m, n, "INT8")
    accum = "plus" if mode == 1 and tname != "BOOL" else None
    if mode == 0 or mode == 1:
        exp = O.mxm(oa, ob, sr, C=oc, accum=accum)
    elif mode == 2:
        exp = O.mxm(oa, ob, sr, mask=om, mask_struct=True)
    else:
        exp = O.mxm(oa, ob, sr, C=oc, mask=om, mask_struct=False, replace=True)
    opts = dict(mxm_unit_small=int(rng.choice([64, 512])), mxm_unit_mid=int(rng.choice([300, 1024])),
                mxm_unit_dense=int(rng.choice([1500, 4096])), mxm_bitmap_pool_cap=int(rng.choice([0, 2, (1 << 31) - 1])),
                mxm_unit_min_flops=int(rng.choice([128, 1024])), mxm_masked_units_min_flops=0, mxm_mask_mode=2 if mode == 2 else 0,
                mxm_sym_windows=int(rng.choice([1, 2, 4, 64])), mxm_window_groups=groups)
    try:
        for name, val in opts.items():
            assert _lib.lib.GrX_option_set(name.encode(), val) == 0, name
        A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
        B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
        C = gb.Matrix.from_coo(cr, cc, cv, dtype=tname, nrows=m, ncols=n) if mode in (1, 3) else gb.Matrix(tname, m, n)
        M = gb.Matrix.from_coo(mr, mc, mv, dtype="INT8", nrows=m, ncols=n)
        if mode == 0:
            C << A.mxm(B, getattr(gb.semiring, sr))
        elif mode == 1:
            if accum:
                C(accum=getattr(gb.binary, accum)) << A.mxm(B, getattr(gb.semiring, sr))
            else:
                C << A.mxm(B, getattr(gb.semiring, sr))
                exp = O.mxm(oa, ob, sr)
        elif mode == 2:
            C(M.S) << A.mxm(B, getattr(gb.semiring, sr))
        else:
            C(M.V, replace=True) << A.mxm(B, getattr(gb.semiring, sr))
        if sr.startswith("any_") and not sr.endswith("pair"):  # (any: one of the products, not a fixed one -- the pattern is checked)
            I, J, _ = C.to_coo()
            er, ec, _ = exp.to_coo()
            assert I.tolist() == er.tolist() and J.tolist() == ec.tolist()
        else:
            same_mat(C, exp)
    finally:
        for name, val in dict(mxm_unit_small=512, mxm_unit_mid=1024, mxm_unit_dense=4096, mxm_bitmap_pool_cap=(1 << 31) - 1,
                              mxm_unit_min_flops=1024, mxm_masked_units_min_flops=64 << 20, mxm_mask_mode=1, mxm_sym_windows=8,
                              mxm_window_groups=0).items():
            _lib.lib.GrX_option_set(name.encode(), val)


@pytest.mark.parametrize("seed", range(28))
def test_mxm_complemented_mask_fused(gb, seed, request):
    """C<!M> = A (+.x) B with the complemented mask fused into the product (the forbidden positions never enter T): rows of
    every size class -- LDS hash tables (forbidden columns pre-inserted), (row, column window) units of every class with and
    without kept bitmaps, the dense unit, the 1024-thread window walk --, structural and valued masks whose entries cover a large
    part of the product's pattern (and positions outside it, and whole rows of it), with / without an old C, accumulators,
    replace.  Checked against the oracle, and against the unfused path (full product + write rule) of the library itself."""
    from graphblas_amd import _lib, device

    on_gpu = request.node.callspec.params["gb"] == "gpu"
    if not on_gpu and seed >= 20:
        pytest.skip("the emulator tier runs the first 20 cases (a few seconds each); the GPU tier runs all")
    rng = np.random.default_rng(9900 + seed)
    tname = TYPES[seed % 7]
    srs = semirings_for(tname)
    sr = srs[int(rng.integers(len(srs)))]
    small = seed % 4 == 3  # hash-table rows only
    nwin = 1 if small else int(rng.integers(1, 5))
    n = int(rng.integers(20, 400)) if small else int(rng.integers((nwin - 1) * 16384 + 1, nwin * 16384 + 1))
    k = int(rng.integers(5, 80)) if small else int(rng.integers(40, 300 if on_gpu else 70))
    m = int(rng.integers(3, 60)) if small else int(rng.integers(3, 30 if on_gpu else 7))
    wts = rng.random(nwin) ** 2
    wts /= wts.sum()
    br, bc = [], []
    for r in range(k):
        d = int(rng.integers(1, 30)) if small else int(rng.integers(20, 500 if on_gpu else 110))
        win = rng.choice(nwin, d, p=wts)
        cols = np.unique(np.minimum(win * 16384 + rng.integers(0, min(n, 16384), d), n - 1))
        br.append(np.full(cols.size, r))
        bc.append(cols)
    br, bc = np.concatenate(br), np.concatenate(bc)
    deg = rng.integers(0, 12, m)
    heavy = rng.random(m) < (0.2 if small else 0.6)
    deg[heavy] = rng.integers(max(1, k // 4), k + 1, int(heavy.sum()))
    deg = np.minimum(deg, k)
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg]) if deg.sum() else np.zeros(0, np.int64)
    av, bv = rand_vals(rng, ar.size, tname), rand_vals(rng, br.size, tname)
    if seed % 5 == 0 and av.size and bv.size:
        av[:] = av[0]
        bv[:] = bv[0]
    oa, ob = O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname)
    # the mask: a fraction of the product's own pattern (per row: none / some / most / all of it) + positions outside it
    tr, tc, _ = O.mxm(oa, ob, "any_pair" if tname == "BOOL" else sr).to_coo()
    tr, tc = np.asarray(tr, np.int64), np.asarray(tc, np.int64)
    frac = rng.choice([0.0, 0.1, 0.6, 1.0], m)
    keep = rng.random(tr.size) < frac[tr] if tr.size else np.zeros(0, bool)
    xr = rng.integers(0, m, 50 if small else 400)
    xc = rng.integers(0, n, xr.size)
    key = np.unique(np.concatenate([tr[keep] * n + tc[keep], xr * n + xc]))
    mr, mc = key // n, key % n
    mv = rng.integers(0, 2, mr.size).astype(np.int8)
    struct, repl = bool(seed & 1), bool(seed & 2)
    accum = [None, "plus", "second", None][seed % 4] if seed % 3 == 0 and tname != "BOOL" else None
    use_c = accum is not None or bool(rng.integers(2))
    cr, cc, cv = rand_coo(rng, m, n, tname, long_rows=1)
    oc = O.OMat.from_coo(cr, cc, cv, m, n, tname) if use_c else None
    om = O.OMat.from_coo(mr, mc, mv, m, n, "INT8")
    exp = O.mxm(oa, ob, sr, C=oc, mask=om, mask_comp=True, mask_struct=struct, accum=accum, replace=repl)
    opts = dict(mxm_unit_small=int(rng.choice([64, 512])), mxm_unit_mid=int(rng.choice([300, 1024])),
                mxm_unit_dense=int(rng.choice([1500, 4096])), mxm_bitmap_pool_cap=int(rng.choice([0, 2, (1 << 31) - 1])),
                mxm_unit_min_flops=int(rng.choice([128, 1024, 1 << 30])),  # (2^30: no units -- the window walk takes the heavy rows)
                mxm_sym_windows=int(rng.choice([1, 3, 4, 64])))
    any_values = sr.startswith("any_") and not sr.endswith("pair")
    try:
        for name, val in opts.items():
            _lib.lib.GrX_option_set(name.encode(), val)
        A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
        B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
        M = gb.Matrix.from_coo(mr, mc, mv, dtype="INT8", nrows=m, ncols=n)
        got = []
        for mask_mode in (1, 0):  # fused, then the full product + write rule
            _lib.lib.GrX_option_set(b"mxm_mask_mode", mask_mode)
            C = gb.Matrix.from_coo(cr, cc, cv, dtype=tname, nrows=m, ncols=n) if use_c else gb.Matrix(tname, m, n)
            kw = dict(mask=~(M.S if struct else M.V), replace=repl)
            if accum:
                kw["accum"] = getattr(gb.binary, accum)
            C(**kw) << A.mxm(B, getattr(gb.semiring, sr))
            if ar.size and br.size and mr.size:
                assert device.last_stats()["method"] == (7 if mask_mode == 1 else 3)
            got.append(C)
        for C in got:
            if any_values:  # (any: one of the products, not a fixed one -- the pattern is checked)
                I, J, _ = C.to_coo()
                er, ec, _ = exp.to_coo()
                assert I.tolist() == er.tolist() and J.tolist() == ec.tolist()
            else:
                same_mat(C, exp)
    finally:
        for name, val in dict(mxm_unit_small=512, mxm_unit_mid=1024, mxm_unit_dense=4096, mxm_bitmap_pool_cap=(1 << 31) - 1,
                              mxm_unit_min_flops=1024, mxm_mask_mode=1, mxm_sym_windows=8).items():
            _lib.lib.GrX_option_set(name.encode(), val)


@pytest.mark.parametrize("seed", range(24))
def test_matrix_write_rule_wavefront_merge(gb, seed, request):
    """The write rule C<M, replace> = accum(C, T) of matrix results (k_mat_write_wave): rows of C_old, T and M of every length
    against each other -- empty, a few entries, several 64-entry chunks, rows cut into column pieces (more than 8192 entries on
    more than 16384 columns), a full row --, the mask denser than C_old and T (the merge step bounded by the mask chunk), false
    mask values, complement, replace, accumulators, iso C_old.  T is given exactly (A = identity, so T = B), the result is
    checked against the oracle and against the thread-per-row kernel of rounds 1-2."""
    from graphblas_amd import _lib

    rng = np.random.default_rng(7700 + seed)
    tname = ["INT64", "FP64", "INT8", "BOOL", "FP32", "UINT16"][seed % 6]
    sr = "lor_land" if tname == "BOOL" else "plus_times"
    m = 7
    on_gpu = request.node.callspec.params["gb"] == "gpu"  # (the emulator tier: the same shapes, shorter rows)
    n = int(rng.integers(300, 3000)) if seed % 4 == 3 else int(rng.integers(17000, 70000 if on_gpu else 36000))
    shapes = [0, 3, 70, min(n, 200), min(n, 9000 if on_gpu else 8400), min(n, 20000 if on_gpu else 11000), n]

    def rows_with(lengths, vals_type):
        lengths = rng.permutation(lengths)
        r = np.concatenate([np.full(ln, i) for i, ln in enumerate(lengths)]).astype(np.int64)
        c = np.concatenate([np.sort(rng.choice(n, ln, replace=False)) for ln in lengths]).astype(np.int64)
        return r, c, rand_vals(rng, r.size, vals_type)

    br, bc, bv = rows_with(shapes, tname)
    cr, cc, cv = rows_with(shapes, tname)
    mr, mc, mv = rows_with([0, 1, 40, min(n, 500), min(n, 15000 if on_gpu else 9000), min(n, 30000 if on_gpu else 14000), n], "INT8")
    if seed % 5 == 1 and cv.size:
        cv[:] = cv[0]
    eye = np.arange(m)
    one = np.ones(m, dtype=bv.dtype)
    oa, ob = O.OMat.from_coo(eye, eye, one, m, m, tname), O.OMat.from_coo(br, bc, bv, m, n, tname)
    oc, om = O.OMat.from_coo(cr, cc, cv, m, n, tname), O.OMat.from_coo(mr, mc, mv, m, n, "INT8")
    mask_kind = seed % 3  # 0 none, 1 structural, 2 valued
    comp, repl = bool(seed & 1) and mask_kind > 0, bool(seed & 2) and mask_kind > 0
    accum = [None, "plus", "second", "min"][seed % 4] if tname != "BOOL" else [None, "lor"][seed % 2]
    if mask_kind == 0 and accum is None:
        accum = "lor" if tname == "BOOL" else "plus"  # (no mask and no accum: C = T, no write rule at all)
    exp = O.mxm(oa, ob, sr, C=oc, mask=om if mask_kind else None, mask_comp=comp, mask_struct=mask_kind == 1, accum=accum, replace=repl)
    A = gb.Matrix.from_coo(eye, eye, one, dtype=tname, nrows=m, ncols=m)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=m, ncols=n)
    M = gb.Matrix.from_coo(mr, mc, mv, dtype="INT8", nrows=m, ncols=n)
    try:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 0)  # (the full product, then the write rule -- the kernel under test)
        for kernel in (1, 0):
            _lib.lib.GrX_option_set(b"mat_write_kernel", kernel)
            C = gb.Matrix.from_coo(cr, cc, cv, dtype=tname, nrows=m, ncols=n)
            kw = {}
            if mask_kind:
                mm = M.S if mask_kind == 1 else M.V
                kw = dict(mask=~mm if comp else mm, replace=repl)
            if accum:
                kw["accum"] = getattr(gb.binary, accum)
            C(**kw) << A.mxm(B, getattr(gb.semiring, sr))
            same_mat(C, exp)
    finally:
        _lib.lib.GrX_option_set(b"mxm_mask_mode", 1)
        _lib.lib.GrX_option_set(b"mat_write_kernel", 1)


@pytest.mark.parametrize("sr,tname", [("plus_times", "INT64"), ("min_plus", "FP64"), ("any_pair", "BOOL"), ("plus_pair", "UINT16")])
def test_mxm_masked_unit_classes(gb, sr, tname):
    """C<M.S> = A (+.x) B, mask-driven, with the heavy rows walked as (row, column window) units: the bitmap of a unit is the mask
    row's part inside the window, the accumulators are keyed by the mask entry.  Mask rows with > 4096 entries in a window (several
    passes of 4096 accumulators), 512 .. 4096 (four wavefronts), a few (one wavefront), none; a heavy row without mask entries;
    light rows (LDS hash keyed by the mask row)."""
    rng = np.random.default_rng(199)
    m, k, n = 9, 220, 40_000
    np_t = O.NP_OF[tname]
    bc, br = [], []
    for r in range(k):
        cols = np.concatenate([rng.choice(16384, 150, replace=False), 16384 + rng.choice(16384, 40, replace=False),
                               32768 + rng.choice(n - 32768, 3, replace=False)])
        bc.append(np.sort(cols))
        br.append(np.full(cols.size, r))
    br, bc = np.concatenate(br), np.concatenate(bc)
    deg = np.array([100, 0, 25, 3, 180, 1, 60, 0, 26])
    ar = np.repeat(np.arange(m), deg)
    ac = np.concatenate([np.sort(rng.choice(k, d, replace=False)) for d in deg])
    # the mask: per row, how many entries in each of the three windows
    per_win = {0: (6000, 800, 100), 2: (20, 5, 0), 3: (10, 0, 2), 4: (3000, 1500, 0), 5: (4, 4, 4), 8: (600, 30, 7000)}
    mr, mc = [], []
    for r, (n0, n1, n2) in per_win.items():
        cols = np.concatenate([rng.choice(16384, n0, replace=False), 16384 + rng.choice(16384, n1, replace=False),
                               32768 + rng.choice(n - 32768, n2, replace=False)])
        mr.append(np.full(cols.size, r))
        mc.append(np.sort(cols))
    mr, mc = np.concatenate(mr), np.concatenate(mc)
    if tname == "BOOL":
        av, bv = np.ones(ar.size, bool), np.ones(br.size, bool)
    else:
        av, bv = rng.integers(1, 6, ar.size).astype(np_t), rng.integers(1, 6, br.size).astype(np_t)
    A = gb.Matrix.from_coo(ar, ac, av, dtype=tname, nrows=m, ncols=k)
    B = gb.Matrix.from_coo(br, bc, bv, dtype=tname, nrows=k, ncols=n)
    M = gb.Matrix.from_coo(mr, mc, np.ones(mr.size, bool), dtype="BOOL", nrows=m, ncols=n)
    C = gb.Matrix(tname, m, n)
    from graphblas_amd import _lib

    try:
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 0)  # (small products keep the row kernels by default)
        C(M.S) << A.mxm(B, getattr(gb.semiring, sr))
    finally:
        _lib.lib.GrX_option_set(b"mxm_masked_units_min_flops", 64 << 20)
    om = O.OMat.from_coo(mr, mc, np.ones(mr.size, bool), m, n, "BOOL")
    oc = O.mxm(O.OMat.from_coo(ar, ac, av, m, k, tname), O.OMat.from_coo(br, bc, bv, k, n, tname), sr, mask=om, mask_struct=True)
    cp, cj, cx = C.to_csr()
    assert np.array_equal(cp.astype(np.int64), oc.indptr) and np.array_equal(cj.astype(np.int64), oc.indices)
    assert np.array_equal(cx, oc.values)
    assert oc.nvals > 3000  # (the mask really met the product)


@pytest.mark.parametrize("seed", range(24))
def test_vector_assign_extract_random(gb, seed):
    """GrB_Vector_assign (vector and scalar sources) and GrB_Vector_extract with random index lists (no duplicates for assign),
    every mask form, accumulators, replace, typecasts between source and output -- against the oracle's restatement."""
    rng = np.random.default_rng(7700 + seed)
    tname = TYPES[seed % 7]
    src_t = TYPES[(seed // 2) % 7] if seed % 3 == 0 and tname != "BOOL" else tname
    if src_t == "BOOL" and tname != "BOOL":
        src_t = tname
    n = int(rng.integers(1, 400))
    ni = int(rng.integers(0, n + 1))
    I = rng.choice(n, ni, replace=False)
    if seed % 4 == 0:
        I = np.sort(I)
    wi, wv = rand_vec(rng, n, 0.5, tname)
    ui, uv = rand_vec(rng, ni, 0.6, src_t)
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    accum = [None, "plus", "min", None][seed % 4] if tname != "BOOL" else [None, "lor"][seed % 2]
    comp, struct, repl = bool(seed & 1), bool(seed & 2), bool(seed & 4)
    use_mask = seed % 5 != 0
    ow, ou, om = O.OVec(n, wi, wv, tname), O.OVec(ni, ui, uv, src_t), O.OVec(n, mi, mv, "BOOL")

    def target(w):
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        if not use_mask:
            return w(accum=accum)
        m = mk.S if struct else mk.V
        return w(~m if comp else m, accum=accum, replace=repl)

    kw = dict(mask=om if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct, accum=accum, replace=repl and use_mask)
    # vector source
    w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
    u = gb.Vector.from_coo(ui, uv, dtype=src_t, size=ni)
    target(w)[I] << u
    same_vec(w, O.vec_assign(ow, ou, I, **kw))
    # scalar source
    w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
    sval = rand_vals(rng, 1, tname)[0]
    target(w)[I] << sval
    same_vec(w, O.vec_assign(ow, sval, I, **kw))
    # extract (duplicates allowed): output of size nj from u2 of size n
    nj = int(rng.integers(1, 300))
    J = rng.integers(0, n, nj)
    xi, xv = rand_vec(rng, nj, 0.5, tname)
    m2i, m2v = rand_vec(rng, nj, 0.5, "BOOL")
    x = gb.Vector.from_coo(xi, xv, dtype=tname, size=nj)
    src = gb.Vector.from_coo(wi, wv, dtype=tname if src_t == tname else src_t, size=n) if False else gb.Vector.from_coo(wi, cast_to(wv, src_t), dtype=src_t, size=n)
    mk2 = gb.Vector.from_coo(m2i, m2v, dtype="BOOL", size=nj)
    if use_mask:
        m = mk2.S if struct else mk2.V
        x(~m if comp else m, accum=accum, replace=repl) << src[J]
    else:
        x(accum=accum) << src[J]
    exp = O.vec_extract(O.OVec(nj, xi, xv, tname), O.OVec(n, wi, cast_to(wv, src_t), src_t), J, mask=O.OVec(nj, m2i, m2v, "BOOL") if use_mask else None,
                        mask_comp=comp and use_mask, mask_struct=struct, accum=accum, replace=repl and use_mask)
    same_vec(x, exp)


def cast_to(vals, tname):
    return O.cast(np.asarray(vals), tname)


@pytest.mark.parametrize("seed", range(12))
def test_push_small_frontiers_in_one_workgroup(gb, seed):
    """A pushed frontier of a few vertices (at most 64 work items of 1024 entries) runs the three push passes as phases of ONE
    workgroup (k_push_small: grb_mxv_push.inc) when what the write rule deletes from w is nothing -- w empty, or an accumulator
    without replace -- or sits on the frontier list (w IS the frontier: the level step of a BFS).  Every flavour against the oracle,
    and against the pass-per-kernel form (option "push_small" 0); an output with content of its own under a deleting rule takes
    that form by itself.  Rows of 0, 1, 1023, 1024, 1025 and 3000 entries in the frontier: the cuts of the work items."""
    from graphblas_amd import _lib, device

    rng = np.random.default_rng(7700 + seed)
    tname = TYPES[seed % 7]
    sr = semirings_for(tname)[(seed // 2) % 4]
    n = int(rng.integers(3100, 5000))
    deg = rng.integers(0, 5, n)
    special = rng.permutation(n)[:6]
    for v, ln in zip(special, (0, 1, 1023, 1024, 1025, 3000)):
        deg[v] = ln
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([rng.choice(n, d, replace=False) for d in deg if d] or [np.zeros(0, np.int64)])
    vals = rand_vals(rng, rows.size, tname)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, tname)
    # the frontier: the special rows (some seeds) plus a few others
    fi = np.unique(np.concatenate([special[: 1 + seed % 6], rng.integers(0, n, 1 + seed % 4)]))
    fv = rand_vals(rng, fi.size, tname)
    of = O.OVec(n, fi, fv, tname)
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    om = O.OVec(n, mi, mv, "BOOL")
    wi, wv = rand_vec(rng, n, 0.3, tname)
    accum = [None, "plus", "min", "second"][seed % 4]
    try:
        _lib.lib.GrX_option_set(b"push_mode", 2)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        semi = getattr(gb.semiring, sr)
        for small in (1, 0):
            assert _lib.lib.GrX_option_set(b"push_small", small) == 0
            ran_small = -2 if small else -1
            # (1) w empty, complemented structural mask + replace: one level step into a fresh vector
            f = gb.Vector.from_coo(fi, fv, dtype=tname, size=n)
            w = gb.Vector(tname, n)
            w(~mk.S, replace=True) << f.vxm(A, semi)
            st = device.last_stats()
            assert st["method"] == 2 and st["long_kernel"] == ran_small, st
            same_vec(w, O.vxm(of, oa, sr, mask=om, mask_comp=True, mask_struct=True, replace=True))
            # (2) w IS the frontier: q<!m.S, replace> = q A, then once more without a mask
            q = gb.Vector.from_coo(fi, fv, dtype=tname, size=n)
            q(~mk.S, replace=True) << q.vxm(A, semi)
            assert device.last_stats()["long_kernel"] == ran_small
            same_vec(q, O.vxm(of, oa, sr, w=of, mask=om, mask_comp=True, mask_struct=True, replace=True))
            q2 = gb.Vector.from_coo(fi, fv, dtype=tname, size=n)
            if accum:
                q2(mk.V, accum=accum) << q2.vxm(A, semi)
                exp2 = O.vxm(of, oa, sr, w=of, mask=om, accum=accum)
            else:
                q2 << q2.vxm(A, semi)
                exp2 = O.vxm(of, oa, sr, w=of)
            assert device.last_stats()["long_kernel"] == ran_small
            same_vec(q2, exp2)
            # (3) w with content of its own: an accumulator without replace deletes nothing (one workgroup); any other rule walks w's words
            w3 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
            w3(mk.S, accum=accum or "plus") << f.vxm(A, semi)
            assert device.last_stats()["long_kernel"] == ran_small
            same_vec(w3, O.vxm(of, oa, sr, w=O.OVec(n, wi, wv, tname), mask=om, mask_struct=True, accum=accum or "plus"))
            w4 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
            w4(mk.S, accum=accum, replace=True) << f.vxm(A, semi)
            st = device.last_stats()
            assert st["method"] == 2 and st["long_kernel"] == -1, st
            same_vec(w4, O.vxm(of, oa, sr, w=O.OVec(n, wi, wv, tname), mask=om, mask_struct=True, accum=accum, replace=True))
            # (4) the counters are left clean: a second small call right after a large one, and after a pulled one
            big_i = np.unique(rng.integers(0, n, n // 3))
            big = gb.Vector.from_coo(big_i, rand_vals(rng, big_i.size, tname), dtype=tname, size=n)
            _ = big.vxm(A, semi).new()
            w5 = gb.Vector(tname, n)
            w5 << f.vxm(A, semi)
            assert device.last_stats()["long_kernel"] == ran_small
            same_vec(w5, O.vxm(of, oa, sr))
            # (5) a frontier whose rows are all empty: no work item at all -- w stays empty, a frontier that is its own output is emptied
            z = gb.Vector.from_coo(special[:1], rand_vals(rng, 1, tname), dtype=tname, size=n)
            w6 = gb.Vector(tname, n)
            w6(~mk.S, replace=True) << z.vxm(A, semi)
            st = device.last_stats()
            assert st["method"] == 2 and st["flops"] == 0 and st["long_kernel"] == ran_small, st
            assert w6.nvals == 0
            z(~mk.S, replace=True) << z.vxm(A, semi)
            assert z.nvals == 0
            w7 = gb.Vector(tname, n)
            w7 << f.vxm(A, semi)  # (and the counters were left clean)
            same_vec(w7, O.vxm(of, oa, sr))
    finally:
        _lib.lib.GrX_option_set(b"push_mode", 1)
        _lib.lib.GrX_option_set(b"push_small", 1)
