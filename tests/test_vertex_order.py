"""Popularity-ordered layouts (python-graphblas_amd/csrc/grb_mxv_order.inc): a large square matrix is laid out a second time in a
vertex order of its own and the vectors of its space are KEPT in that order between calls.  The order is a storage format: every
result must be the one of the natural order, element for element -- checked here against the oracle with the size thresholds lowered
so that small graphs take the path the scale-24 bench graph takes (GPU tier: the HIP kernels; CPU tier: the same sources under the
wave64 emulator).

Covers: the masked / accumulated product, the plain product, vxm over the transpose (which shares the order), products that cannot
take the ordered layouts in between (another type, the row-length path), operands converted on first use and reused, every way a
vector leaves the library (to_coo, element access, reduce, isequal, device views), element-wise operations between ordered and
natural vectors, scalar assignment under a mask, the SSSP loop of the reference's primer and a level BFS written with the
reference's calls."""
import numpy as np
import pytest

from oracle import grb_oracle as O
from tests.backend import DEVICES, bind
from tests.test_random_parity import rand_vals, rand_vec, same_vec

ORDER_OPTS = ((b"order_min_nnz", 1), (b"lean_min_nnz", 1), (b"split_min_nnz", 1), (b"split_min_len", 8), (b"push_mode", 0), (b"hot_min_cols", 8),
              (b"lazy_layout", 0), (b"vec_pad_min_bytes", 0), (b"rows_head_min_groups", 1))
RESTORE = ((b"order_min_nnz", 24 << 20), (b"lean_min_nnz", 48 << 20), (b"split_min_nnz", 1 << 22), (b"split_min_len", 0), (b"push_mode", 1),
           (b"hot_min_cols", 1 << 20), (b"hot_k", 0), (b"lazy_layout", 1), (b"vec_pad_min_bytes", 1 << 20), (b"long_classes", 16), (b"order_mode", 1), (b"hub_min_len", 1024), (b"rows_head_min_groups", 16384), (b"rows_head", 1),
           (b"rows_tile", 1), (b"rtile_rows", 8192), (b"rtile_entries", 49152), (b"bool_probe", 8), (b"stream_nt_min_nnz", 48 << 20), (b"cold_in_rows", 0), (b"rtile_pack", 1), (b"strip_slot16", 1), (b"ctile_pack", 0), (b"lazy_tagged", 1))


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


def skewed_square(rng, n, tname, hub_rows=6):
    """A square power-law-ish graph: most references go to a tenth of the columns, a few rows are long, many are empty."""
    deg = rng.integers(0, 7, n)
    deg[rng.random(n) < 0.4] = 0
    for ln in (8, 9, 17, 63, 64, 65, 300, 513, int(rng.integers(700, n // 2))) + tuple(int(x) for x in rng.integers(100, 600, hub_rows + n // 150)):
        deg[rng.integers(0, n)] = ln  # (the long rows must hold a good share of the entries, or the split declines)
    hot = rng.permutation(n)[: n // 10]
    rows = np.repeat(np.arange(n), deg)
    cols = np.where(rng.random(rows.size) < 0.7, hot[rng.integers(0, hot.size, rows.size)], rng.integers(0, n, rows.size))
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    return rows, cols, rand_vals(rng, rows.size, tname)


def set_opts(opts):
    from graphblas_amd import _lib

    for name, val in opts:
        assert _lib.lib.GrX_option_set(name, val) == 0, name


@pytest.mark.parametrize("seed", range(21))
def test_ordered_product_matches_the_oracle(gb, seed):
    from graphblas_amd import device

    rng = np.random.default_rng(4100 + seed)
    tname = ["FP32", "INT64", "FP64", "INT32", "UINT16", "INT8", "BOOL"][seed % 7]
    sr = ["min_plus", "plus_times", "max_plus", "any_pair", "min_second", "plus_plus"][(seed // 2) % 6] if tname != "BOOL" else ["lor_land", "any_pair", "lxor_land"][(seed // 7) % 3]
    n = int(rng.integers(2200, 5200))
    rows, cols, vals = skewed_square(rng, n, tname)
    iso = seed % 5 == 3 or (tname == "BOOL" and seed % 2 == 0)
    if iso:
        vals = np.full(rows.size, vals[0])
    ui, uv = rand_vec(rng, n, [1.0, 0.5, 0.05][seed % 3], tname)
    wi, wv = rand_vec(rng, n, 0.5, tname)
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    accum = [None, "plus", "min"][seed % 3] if tname != "BOOL" else [None, "lor", "land"][seed % 3]
    comp, repl, struct = bool(seed & 1), bool(seed & 2), bool(seed & 4)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, tname)
    ou, ow, om = O.OVec(n, ui, uv, tname), O.OVec(n, wi, wv, tname), O.OVec(n, mi, mv, "BOOL")
    exp = O.mxv(oa, ou, sr, w=ow, mask=om, mask_comp=comp, mask_struct=struct, accum=accum, replace=repl)
    try:
        # (hub_min_len: rows from this many entries are dealt to 64 classes -- a second level of hot strips; 0 switches it off)
        set_opts(ORDER_OPTS + ((b"hot_k", [64, 256, 1 << 20][seed % 3]), (b"long_classes", [16, 8, 32][seed % 3]), (b"hub_min_len", [100, 0, 300, 1024][seed % 4]),
                             (b"rows_head", 0 if seed == 13 else 1),  # (BOOL: the short rows with the LDS head of the hottest columns -- seeds 6, 20 -- and without)
                             (b"cold_in_rows", [0, 1024][(seed // 3) % 2]),  # (round 6: the cold entries of the long rows below the hub level with the short rows)
                             (b"strip_slot16", [1, 0][(seed // 2) % 2]),  # (the strips' lane slots as 16-bit offsets from the chunk's base / as 32-bit numbers)
                             (b"ctile_pack", [1, 2, 0][seed % 3])))  # (cold tiles: slot and column in one word / + one-byte value codes / three streams)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        m_arg = (mk.S if struct else mk.V)
        w(~m_arg if comp else m_arg, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        by_rowlen = sr in ("any_pair",) and len(ui) == n
        assert by_rowlen or (st["ordered"] == 1 and st["long_kernel"] == (1 if tname == "BOOL" else 4) and st["reorders"] >= 1), st
        if tname != "BOOL" or seed == 13:  # (the LDS head of the short-row kernel serves BOOL operands given as presence / value pairs)
            assert by_rowlen or st["fused_epilogue"] in (1, 3), st  # (3: the short rows ran as sorted row tiles -- full operand, specialised semiring)
        same_vec(w, exp)  # (to_coo brings w back to the natural order)
        # the same call again: the operands that stayed in the library are still in the matrix's order
        w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        w2(~m_arg if comp else m_arg, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        # (only the fresh w2 -- and not even that when nothing of it survives the call: no accumulator and replace)
        assert by_rowlen or (st["ordered"] == 1 and st["reorders"] == (0 if accum is None and repl else 1)), st
        same_vec(w2, exp)
        # plain product into a new vector, vxm over the transpose (same order), a product in another type (natural path)
        same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), O.mxv(oa, ou, sr))
        xi, xv = rand_vec(rng, n, 0.6, tname)
        x = gb.Vector.from_coo(xi, xv, dtype=tname, size=n)
        same_vec(x.vxm(A, getattr(gb.semiring, sr)).new(), O.vxm(O.OVec(n, xi, xv, tname), oa, sr))
        same_vec(A.T.mxv(x, getattr(gb.semiring, sr)).new(), O.mxv(oa, O.OVec(n, xi, xv, tname), sr, transpose_a=True))
        if tname not in ("FP64", "BOOL"):  # (a semiring of another type than the matrix: typecast copies, natural order)
            same_vec(A.mxv(u, gb.semiring.plus_times["FP64"]).new(),
                     O.mxv(O.OMat.from_coo(rows, cols, vals.astype(np.float64), n, n, "FP64"), O.OVec(n, ui, uv.astype(np.float64), "FP64"), "plus_times"))
            assert device.last_stats()["ordered"] == 0
        # the operand itself is unchanged by all of this
        same_vec(u, ou)
        same_vec(mk, om)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(4))
def test_ordered_vectors_through_every_exit(gb, seed):
    """A vector that the library keeps in a matrix's order must look like any other vector from outside."""
    from graphblas_amd import device

    rng = np.random.default_rng(4300 + seed)
    tname = ["FP32", "INT64", "FP64", "INT32"][seed]
    n = int(rng.integers(2100, 3000))
    rows, cols, vals = skewed_square(rng, n, tname)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, tname)
    ui, uv = rand_vec(rng, n, 0.7, tname)
    ou = O.OVec(n, ui, uv, tname)
    try:
        set_opts(ORDER_OPTS + ((b"hot_k", 256),))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = A.mxv(u, gb.semiring.min_plus).new()
        assert device.last_stats()["ordered"] == 1
        exp = O.mxv(oa, ou, "min_plus")
        dense = np.zeros(n, O.NP_OF[tname])
        dense[exp.idx] = exp.vals
        has = np.zeros(n, bool)
        has[exp.idx] = True
        # element access on the ordered vector (no conversion: the position is looked up)
        for i in list(exp.idx[:3]) + ([int(np.flatnonzero(~has)[0])] if (~has).any() else []):
            got = w[int(i)].value
            assert (got is None and not has[i]) or got == dense[i], (i, got)
        assert w.nvals == exp.idx.size
        # reduce, dup, isequal, element-wise with a NATURAL vector, scalar assignment under the ordered vector as a mask
        assert w.reduce(gb.monoid.min).new().value == (exp.vals.min() if exp.idx.size else None)
        d = w.dup()
        assert d.isequal(w)
        same_vec(d, exp)
        yi, yv = rand_vec(rng, n, 0.5, tname)
        y = gb.Vector.from_coo(yi, yv, dtype=tname, size=n)
        w2 = A.mxv(u, gb.semiring.min_plus).new()  # ordered again
        z = w2.ewise_add(y, gb.binary.plus).new()
        oz = O.OVec(n, *ewise_np(exp, O.OVec(n, yi, yv, tname), n, tname, add=True), tname)
        same_vec(z, oz)
        z2 = w2.ewise_mult(y, gb.binary.min).new()
        same_vec(z2, O.OVec(n, *ewise_np(exp, O.OVec(n, yi, yv, tname), n, tname, add=False), tname))
        t = gb.Vector(tname, size=n)
        t(mask=w2.S) << 7
        ti, tv = t.to_coo()
        assert ti.tolist() == exp.idx.tolist() and set(tv.tolist()) <= {7}
        # setElement / removeElement on the ordered vector, then the whole content
        w3 = A.mxv(u, gb.semiring.min_plus).new()
        w3[5] = 3
        del w3[int(exp.idx[0])]
        dense2, has2 = dense.copy(), has.copy()
        dense2[5], has2[5] = 3, True
        has2[exp.idx[0]] = False
        gi, gv = w3.to_coo()
        assert gi.tolist() == np.flatnonzero(has2).tolist() and gv.tolist() == dense2[has2].tolist()
        # a product with an ordered OUTPUT that is then an INPUT (the loop case): w4 = A min.+ (A min.+ u)
        w4 = A.mxv(w2, gb.semiring.min_plus).new()
        st = device.last_stats()
        assert st["ordered"] == 1 and st["reorders"] == 0, st
        same_vec(w4, O.mxv(oa, exp, "min_plus"))
        # the HBM image handed out to torch / RCCL is the natural one
        vals_t, words_t = device.vector_device_views(w2, "cuda" if _on_gpu() else "cpu")
        words = np.asarray(words_t.cpu()).view(np.uint32)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)
        assert bits.tolist() == has.tolist()
        assert np.asarray(vals_t.cpu())[has].tolist() == dense[has].tolist()
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("tname", ["FP32", "FP64"])
def test_fp_plus_reduction_does_not_depend_on_the_stored_order(gb, tname):
    """a floating-point PLUS / TIMES reduction rounds differently in another fold order, and whether an earlier product left the
    vector in a matrix's vertex order is hidden state -- such reductions see the natural order.  The sum of a vector a product has just left
    ordered equals, bit for bit, the sum of the same entries built as a fresh (natural) vector."""
    from graphblas_amd import device

    rng = np.random.default_rng(991)
    n = 2900
    rows, cols, _ = skewed_square(rng, n, tname)
    vals = (rng.random(rows.size) * 10.0 ** rng.integers(-3, 4, rows.size)).astype(O.NP_OF[tname])
    uv = (rng.random(n) * 10.0 ** rng.integers(-3, 4, n)).astype(O.NP_OF[tname])
    try:
        set_opts(ORDER_OPTS + ((b"hot_k", 256),))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n, dup_op=gb.binary.min)
        u = gb.Vector.from_coo(np.arange(n), uv, dtype=tname, size=n)
        for mon in (gb.monoid.plus, gb.monoid.times):
            w = A.mxv(u, gb.semiring.min_plus).new()
            assert device.last_stats()["ordered"] == 1
            got = w.reduce(mon).new().value  # (w is in the matrix's order when the reduction is called)
            wi, wv = w.to_coo()
            fresh = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
            assert got == fresh.reduce(mon).new().value
    finally:
        set_opts(RESTORE)


def _on_gpu():
    from tests import backend

    return backend._bound == "gpu"


def ewise_np(a, b, n, tname, add):
    da, db = np.zeros(n, O.NP_OF[tname]), np.zeros(n, O.NP_OF[tname])
    ha, hb = np.zeros(n, bool), np.zeros(n, bool)
    da[a.idx], ha[a.idx] = a.vals, True
    db[b.idx], hb[b.idx] = b.vals, True
    if add:
        out = np.where(ha & hb, da + db, np.where(ha, da, db))
        has = ha | hb
    else:
        out = np.minimum(da, db)
        has = ha & hb
    return np.flatnonzero(has), out[has].astype(O.NP_OF[tname])


@pytest.mark.parametrize("seed", range(6))
def test_sssp_and_bfs_loops_stay_ordered(gb, seed):
    """The reference's traversal loops (docs/getting_started/primer.rst:236-246, notebooks Example B.1) on an ordered matrix: after the first
    sweep no vector is converted again -- also when thin frontiers take the push direction, whose kernels translate positions through the
    order's maps (seeds 3-5: even sweeps forced to push, odd ones to pull) -- and the results are the oracle's."""
    from graphblas_amd import device

    rng = np.random.default_rng(4500 + seed)
    n = int(rng.integers(2100, 2600))
    rows, cols, vals = skewed_square(rng, n, "FP32")
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "FP32")
    src = int(rows[np.argmax(np.bincount(rows, minlength=n)[rows])])
    push = seed >= 3
    try:
        set_opts(ORDER_OPTS + ((b"hot_k", 256), (b"push_mode", 1 if push else 0)))
        G = gb.Matrix.from_coo(rows, cols, vals, dtype="FP32", nrows=n, ncols=n)
        # ---- SSSP (Bellman-Ford sweeps until nothing changes)
        v = gb.Vector("FP32", size=n)
        v[src] = 0
        ov = O.OVec(n, np.array([src]), np.array([0], np.float32), "FP32")
        conversions, methods = [], []
        for it in range(n):
            if push:  # (even sweeps pushed, odd ones pulled: the operands change direction without changing order)
                set_opts(((b"push_mode", 2 if it % 2 == 0 else 0),))
            w = v.dup()
            v(gb.binary.min) << v.vxm(G, gb.semiring.min_plus)
            st = device.last_stats()
            conversions.append(st["reorders"])
            methods.append(st["method"])
            ov = O.vxm(ov, oa, "min_plus", w=ov, accum="min")
            if v.isequal(w):
                break
        same_vec(v, ov)
        first_pull = methods.index(1)
        assert it >= 2 and sum(conversions[first_pull + 1:]) == 0, (conversions, methods)  # (the first pull converts v; after that everything stays)
        if push:
            assert methods[0] == 2 and methods[1] == 1 and methods[2] == 2, methods
        # ---- level BFS: levels[:](mask=q.V) << level; q(~levels.S, replace) << q.vxm(G, lor_land) -- on the BOOL view of the graph
        Gb = gb.Matrix.from_coo(rows, cols, np.ones(rows.size, bool), dtype="BOOL", nrows=n, ncols=n)
        ob = O.OMat.from_coo(rows, cols, np.ones(rows.size, bool), n, n, "BOOL")
        q = gb.Vector("BOOL", size=n)
        q[src] = True
        levels = gb.Vector("INT64", size=n)
        oq = O.OVec(n, np.array([src]), np.array([True]), "BOOL")
        olev_i, olev_v = [], []
        conv = []
        for level in range(n):
            if q.nvals == 0:
                break
            if push:
                set_opts(((b"push_mode", 2 if level % 2 == 0 else 0),))
            levels(mask=q.V)[:] << level
            olev_i += oq.idx.tolist()
            olev_v += [level] * oq.idx.size
            q(~levels.S, replace=True) << q.vxm(Gb, gb.semiring.lor_land)
            conv.append((device.last_stats()["method"], device.last_stats()["reorders"], device.last_stats()["fused_epilogue"]))
            seen = O.OVec(n, np.array(sorted(olev_i)), np.ones(len(olev_i), bool), "BOOL")
            oq = O.vxm(oq, ob, "lor_land", w=oq, mask=seen, mask_comp=True, mask_struct=True, replace=True)
        li, lv = levels.to_coo()
        order = np.argsort(olev_i)
        assert li.tolist() == np.asarray(olev_i)[order].tolist() and lv.tolist() == np.asarray(olev_v)[order].tolist()
        assert level >= 2, conv
        # (the pulled levels of the BOOL graph: the frontier's presence / value pairs in the LDS head of the short-row kernel -- also
        #  though q is its own output)
        assert any(m == 1 and fe in (2, 3) for m, _, fe in conv), conv  # (2: the LDS head of the row groups; 3: the sorted row tiles, round 5)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("case", ["256_values", "257_values", "with_inf", "dict_off"])
def test_value_dictionary_boundaries(gb, case):
    """Hot-strip records carry one-byte value codes when a 4-byte matrix holds at most 256 distinct finite values (grb_mxv_vdict.inc):
    exactly 256 (coded), 257 (not coded), an infinite value among few (not coded: the fast kernel's padding needs finite values), and
    the option switched off -- the same results either way, against the oracle, through the fast kernel (full operand) and the generic one
    (sparse operand), ordered layouts."""
    from graphblas_amd import device

    rng = np.random.default_rng(77)
    n = 3000
    rows, cols, _ = skewed_square(rng, n, "FP32")
    k = {"256_values": 256, "257_values": 257, "with_inf": 9, "dict_off": 9}[case]
    pool = (np.arange(1, k + 1) * 0.5).astype(np.float32)
    if case == "with_inf":
        pool[3] = np.inf
    vals = pool[rng.integers(0, k, rows.size)]
    vals[:k] = pool  # (every value of the pool occurs)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "FP32")
    try:
        set_opts(ORDER_OPTS + ((b"hot_k", 1 << 20), (b"value_dict", 0 if case == "dict_off" else 1), (b"hub_min_len", 200)))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype="FP32", nrows=n, ncols=n)
        for dens in (1.0, 0.4):
            ui, uv = rand_vec(rng, n, dens, "FP32")
            u = gb.Vector.from_coo(ui, uv, dtype="FP32", size=n)
            for sr in ("min_plus", "plus_times", "max_plus"):
                got = A.mxv(u, getattr(gb.semiring, sr)).new()
                st = device.last_stats()
                assert st["ordered"] == 1 and st["value_dict"] == (256 if case == "256_values" else 0), st
                exp = O.mxv(oa, O.OVec(n, ui, uv, "FP32"), sr)
                gi, gv = got.to_coo()
                assert gi.tolist() == exp.idx.tolist()
                np.testing.assert_array_equal(gv, exp.vals)
    finally:
        set_opts(RESTORE + ((b"value_dict", 1),))


def test_ordered_vectors_in_unusual_flows(gb):
    """Vectors kept in a matrix's order through the entry points that are NOT order-aware, and through the life cycle of the matrix: build
    into / resize / clear / indexed assign and extract on an ordered vector, a mask that aliases the output, an operand that aliases the
    output, two matrices with different orders sharing vectors, the matrix freed or rebuilt while vectors still carry its order, a pinned
    vector.  Everything against the oracle."""
    from graphblas_amd import device

    rng = np.random.default_rng(911)
    n = 2600
    tname = "INT64"
    rows, cols, vals = skewed_square(rng, n, tname)
    rows2, cols2, vals2 = skewed_square(np.random.default_rng(912), n, tname)
    oa, ob = O.OMat.from_coo(rows, cols, vals, n, n, tname), O.OMat.from_coo(rows2, cols2, vals2, n, n, tname)
    ui, uv = rand_vec(rng, n, 0.6, tname)
    ou = O.OVec(n, ui, uv, tname)
    try:
        set_opts(ORDER_OPTS + ((b"hot_k", 256), (b"hub_min_len", 200)))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        B = gb.Matrix.from_coo(rows2, cols2, vals2, dtype=tname, nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        # ---- an operand that is also the output, a mask that is also the output
        w = u.dup()
        w << A.mxv(w, gb.semiring.plus_times)
        assert device.last_stats()["ordered"] == 1
        e1 = O.mxv(oa, ou, "plus_times")
        same_vec(w, e1)
        w2 = A.mxv(u, gb.semiring.min_plus).new()
        w2(w2.S, accum=gb.binary.plus) << A.mxv(u, gb.semiring.min_plus)
        e2 = O.mxv(oa, ou, "min_plus")
        e2b = O.mxv(oa, ou, "min_plus", w=e2, mask=e2, mask_struct=True, accum="plus")
        same_vec(w2, e2b)
        # ---- two matrices, two orders: the vector goes from one to the other and back
        x = A.mxv(u, gb.semiring.min_plus).new()          # in A's order
        y = B.mxv(x, gb.semiring.min_plus).new()          # x converted to B's
        z = A.mxv(y, gb.semiring.min_plus).new()
        ey = O.mxv(ob, e2, "min_plus")
        same_vec(z, O.mxv(oa, ey, "min_plus"))
        same_vec(y, ey)
        # ---- build into, indexed assign / extract, resize, clear on ordered vectors
        t = A.mxv(u, gb.semiring.min_plus).new()          # ordered
        idx = np.array([3, 77, 1500, 2599])
        got = t[idx].new()
        dense = {int(i): int(v) for i, v in zip(e2.idx, e2.vals)}
        gi, gv = got.to_coo()
        assert gi.tolist() == [k for k, i in enumerate(idx) if int(i) in dense] and gv.tolist() == [dense[int(i)] for i in idx if int(i) in dense]
        t2 = A.mxv(u, gb.semiring.min_plus).new()
        t2[idx] << 5
        d2 = dict(dense)
        for i in idx:
            d2[int(i)] = 5
        gi, gv = t2.to_coo()
        assert gi.tolist() == sorted(d2) and gv.tolist() == [d2[i] for i in sorted(d2)]
        t3 = A.mxv(u, gb.semiring.min_plus).new()
        t3.resize(n - 100)
        gi, gv = t3.to_coo()
        keep = [i for i in sorted(dense) if i < n - 100]
        assert gi.tolist() == keep and gv.tolist() == [dense[i] for i in keep]
        t4 = A.mxv(u, gb.semiring.min_plus).new()
        t4.clear()
        assert t4.nvals == 0
        t4.build([1, 2], [10, 20])
        gi, gv = t4.to_coo()
        assert gi.tolist() == [1, 2] and gv.tolist() == [10, 20]
        # ---- the matrix goes away (or is modified) while a vector still carries its order
        keepv = A.mxv(u, gb.semiring.min_plus).new()
        del A
        import gc

        gc.collect()
        same_vec(keepv, e2)
        k2 = B.mxv(u, gb.semiring.min_plus).new()
        B.resize(n, n)                                     # (drops B's cached layouts and its order; k2 still carries it)
        k3 = B.mxv(k2, gb.semiring.min_plus).new()
        same_vec(k3, O.mxv(ob, O.mxv(ob, ou, "min_plus"), "min_plus"))
        # ---- a pinned vector keeps the products it takes part in on the natural layouts
        A2 = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        p = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        device.vector_pin_natural(p)
        r = A2.mxv(p, gb.semiring.min_plus).new()
        assert device.last_stats()["ordered"] == 0 and device.last_stats()["pinned_natural"] == 1  # (the statistics say why: ADVICE r05)
        same_vec(r, e2)
        assert A2.mxv(u, gb.semiring.min_plus).new().isequal(r) and device.last_stats()["pinned_natural"] == 0
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("sr", ["min_plus", "max_plus"])
def test_sparse_operand_run_as_full_with_absorbing_fill(gb, sr):
    """A sparse operand of a floating-point min_plus / max_plus product on an ordered matrix runs the full-operand kernels on an image with
    +-inf under its absent entries (the pattern of the product = the rows whose accumulator left the identity).  Exact only while every
    present value is finite: an operand that HOLDS an infinity must take the general path -- a row whose only partner is that entry has a
    product (value inf), which the fill would lose.  Both against the oracle, with the option on and off."""
    from graphblas_amd import device

    rng = np.random.default_rng(31)
    n = 2400
    rows, cols, vals = skewed_square(rng, n, "FP32")
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "FP32")
    ui, uv = rand_vec(rng, n, 0.3, "FP32")
    wi, wv = rand_vec(rng, n, 0.4, "FP32")
    bad = float("inf") if sr == "min_plus" else -float("inf")
    uv_inf = uv.copy()
    uv_inf[: max(1, ui.size // 50)] = bad
    #  finite values whose sum with a matrix value overflows: a product of value +-inf that EXISTS -- the fill would take it for
    # "no product"; the bound on the operand's magnitude sends such a call down the general path
    uv_big = uv.copy()
    uv_big[: max(1, ui.size // 50)] = np.float32(3.4e38 if sr == "min_plus" else -3.4e38)
    vals = vals.copy()
    vals[:: 7] = np.float32(1e37 if sr == "min_plus" else -1e37)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "FP32")
    try:
        for fill in (1, 0):
            set_opts(ORDER_OPTS + ((b"hot_k", 256), (b"hub_min_len", 200), (b"fill_absent", fill)))
            A = gb.Matrix.from_coo(rows, cols, vals, dtype="FP32", nrows=n, ncols=n)
            for values in (uv, uv_inf, uv_big):
                u = gb.Vector.from_coo(ui, values, dtype="FP32", size=n)
                ou = O.OVec(n, ui, values, "FP32")
                same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), O.mxv(oa, ou, sr))
                assert device.last_stats()["fill_absent"] == (1 if (fill and values is uv) else 0), device.last_stats()
                w = gb.Vector.from_coo(wi, wv, dtype="FP32", size=n)
                acc = "min" if sr == "min_plus" else "max"
                w(accum=getattr(gb.binary, acc)) << A.mxv(u, getattr(gb.semiring, sr))
                same_vec(w, O.mxv(oa, ou, sr, w=O.OVec(n, wi, wv, "FP32"), accum=acc))
    finally:
        set_opts(RESTORE + ((b"fill_absent", 1),))


def test_view_held_across_an_ordered_product_stays_natural(gb):
    """GrX_Vector_export_dense_device pins -- a torch view obtained BEFORE a product with an ordered matrix must still
    read natural-order data after it (before, the product converted the vector to vertex order inside the aliased buffers), data
    written through it and announced with GrX_Vector_modified must arrive, and a released vector may be ordered again."""
    from graphblas_amd import device

    rng = np.random.default_rng(9090)
    tname, n = "FP32", 3000
    rows, cols, vals = skewed_square(rng, n, tname)
    ui, uv = rand_vec(rng, n, 1.0, tname)
    oa, ou = O.OMat.from_coo(rows, cols, vals, n, n, tname), O.OVec(n, ui, uv, tname)
    dev = "cuda" if _on_gpu() else "cpu"
    try:
        set_opts(ORDER_OPTS)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        held_vals, held_words = device.vector_device_views(u, dev)  # (pins u)
        before = np.asarray(held_vals.cpu()).copy()
        assert before.tolist() == uv.tolist()
        w = A.mxv(u, gb.semiring.min_plus).new()
        st = device.last_stats()
        assert st["ordered"] == 0, st  # (a pinned operand: natural-order layouts)
        assert np.asarray(held_vals.cpu()).tolist() == before.tolist()  # the stale-view hazard: still natural, element for element
        same_vec(w, O.mxv(oa, ou, "min_plus"))
        # write through the held view, announce it, multiply again
        uv2 = (uv + 1).astype(uv.dtype)
        held_vals.copy_(held_vals.new_tensor(uv2))
        device.vector_modified(u)
        same_vec(A.mxv(u, gb.semiring.min_plus).new(), O.mxv(oa, O.OVec(n, ui, uv2, tname), "min_plus"))
        # released: the next product may keep u in the matrix's order ...
        device.vector_release_views(u)
        w3 = A.mxv(u, gb.semiring.min_plus).new()
        assert device.last_stats()["ordered"] == 1
        same_vec(w3, O.mxv(oa, O.OVec(n, ui, uv2, tname), "min_plus"))
        # ... and a TRANSIENT view fetched afterwards reads natural data again and leaves the vector free
        tv, _tw = device.vector_device_views(u, dev, pin=False)
        assert np.asarray(tv.cpu()).tolist() == uv2.tolist()
        A.mxv(u, gb.semiring.min_plus).new()
        assert device.last_stats()["ordered"] == 1
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(18))
def test_sorted_row_tiles_match_the_oracle(gb, seed):
    """Round 5 (grb_mxv_rtile.inc): the short rows of an ordered matrix as sorted row tiles -- tiles cut by rows and by entries (small
    limits: many tiles, dense and thin ones, tiles of one group), entries sorted by column code in lane-transposed blocks, dictionary-coded
    and full values, 4- and 8-byte types, every mask / accumulator / replace combination the kernel takes, long rows inside the tiles,
    a sparse operand run as a full one (absorbing fill), and the calls it must NOT take (sparse operand of plus_times, another semiring),
    which fall back to the tagged row groups.  Against the oracle, element for element."""
    from graphblas_amd import device

    rng = np.random.default_rng(7700 + seed)
    tname = ["FP32", "INT64", "FP64", "FP32", "INT32", "FP32"][seed % 6]
    sr = ["min_plus", "plus_times", "min_plus", "plus_times", "min_plus", "max_plus"][seed % 6]
    n = int(rng.integers(2500, 6000))
    rows, cols, vals = skewed_square(rng, n, tname)
    if seed % 4 == 1:  # (more than 256 distinct values: no dictionary)
        vals = (rng.integers(1, 4000, rows.size)).astype(vals.dtype)
    full = seed % 5 != 4
    ui, uv = rand_vec(rng, n, 1.0 if full else 0.4, tname)
    if tname in ("FP32", "FP64"):
        vals, uv = np.abs(vals) + 1, np.abs(uv)
    wi, wv = rand_vec(rng, n, [1.0, 0.6, 0.0][seed % 3], tname)
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    accum = [None, "min", "plus"][(seed // 2) % 3]
    use_mask, comp, repl, struct = seed % 7 != 6, bool(seed & 1), bool(seed & 2) and seed % 7 != 6, bool(seed & 4)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, tname)
    ou, ow, om = O.OVec(n, ui, uv, tname), O.OVec(n, wi, wv, tname), O.OVec(n, mi, mv, "BOOL")
    exp = O.mxv(oa, ou, sr, w=ow, mask=om if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct, accum=accum, replace=repl)
    try:
        # (seeds 12 ..: the natural-order layouts -- order_mode 0, what the row blocks of a sharded run take: the columns re-coded through the
        #  hot table, the operand image [table | u] built per call -- carry the tiles too)
        natural = seed >= 12
        set_opts(ORDER_OPTS + ((b"rtile_rows", [8192, 16384][seed % 2]), (b"rtile_entries", [256, 700, 5000][seed % 3]), (b"hub_min_len", [100, 0][seed % 2]),
                               (b"order_mode", 0 if natural else 1), (b"hot_k", 256 if natural else 0), (b"rows_tile", 2 if natural else 1),
                               (b"stream_nt_min_nnz", [1, 48 << 20][(seed >> 1) & 1]),  # (1: the streams of the row tiles and cold tiles are read non-temporal)
                               # (round 6: the cold entries of the long rows below the hub level with the short rows -- 1024: all of them, 40: of the
                               #  rows below 40 entries, 0: none, i.e. the cold tiles of rounds 3-5)
                               (b"cold_in_rows", [1024, 0, 40][(seed // 2) % 3]),
                               (b"rtile_pack", [1, 0][(seed // 3) % 2]),  # (column and value code of a dictionary-coded entry in one word / in two streams)
                               (b"ctile_pack", [2, 1, 0][(seed // 2) % 3]),
                               # (the tagged row groups' entries at layout build / from the tiles at the first call the tiles do not take)
                               (b"lazy_tagged", [1, 1, 0][seed % 3])))
        cold_in_rows = [1024, 0, 40][(seed // 2) % 3]
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        m_arg = (mk.S if struct else mk.V)
        kw = dict(accum=accum, replace=repl) if use_mask else dict(accum=accum)
        target = w(~m_arg if comp else m_arg, **kw) if use_mask else w(**kw)
        target << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        specialised = (tname, sr) in (("FP32", "min_plus"), ("FP64", "min_plus"), ("INT64", "min_plus"), ("FP32", "plus_times"), ("FP64", "plus_times"), ("INT64", "plus_times"))
        takes = specialised and (full or (st["fill_absent"] == 1))
        assert st["ordered"] == (0 if natural else 1) and st["fused_epilogue"] == (3 if takes else 1), (st, takes)
        # (the short-row kernel -- row tiles or, for the calls they do not take, tagged row groups -- merged the strips' accumulators into rows
        #  whose cold entries it folded itself: split_min_len 8 < cold_in_rows)
        assert st["long_tails"] == (1 if (not natural and cold_in_rows > 8) else 0), st
        same_vec(w, exp)
        # a call the tiles do NOT take on the same matrix (a semiring without a compiled tile kernel): the tagged row groups, whose entries are
        # laid out from the tiles now when lazy_tagged is on -- against the oracle
        if tname in ("FP32", "FP64", "INT64", "INT32"):
            w9 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
            (w9(~m_arg if comp else m_arg, **kw) if use_mask else w9(**kw)) << A.mxv(u, gb.semiring.max_second)
            assert device.last_stats()["fused_epilogue"] == 1, device.last_stats()
            same_vec(w9, O.mxv(oa, ou, "max_second", w=ow, mask=om if use_mask else None, mask_comp=comp and use_mask, mask_struct=struct, accum=accum, replace=repl))
        # once more on the converted operands (nothing is reordered any more), and without the tiles: the same result
        w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        (w2(~m_arg if comp else m_arg, **kw) if use_mask else w2(**kw)) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w2, exp)
        set_opts(((b"rows_tile", 0),))
        w3 = gb.Vector.from_coo(wi, wv, dtype=tname, size=n)
        (w3(~m_arg if comp else m_arg, **kw) if use_mask else w3(**kw)) << A.mxv(u, getattr(gb.semiring, sr))
        assert device.last_stats()["fused_epilogue"] == 1
        same_vec(w3, exp)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(12))
def test_sorted_row_tiles_bool_step(gb, seed):
    """k_mxv_rtile_bool: the BFS level step on an iso BOOL matrix -- q<!visited.S, replace> = A lor.land q, and any.pair -- with the operand
    as presence / value pairs, accumulators of two bits per row; false operand values (present but false: no true product), masks of every
    kind, accumulators, outputs that are their own operand (fresh buffers: NOT taken by the tiles).  Against the oracle."""
    from graphblas_amd import device

    rng = np.random.default_rng(8800 + seed)
    n = int(rng.integers(2500, 6000))
    rows, cols, _ = skewed_square(rng, n, "BOOL")
    vals = np.ones(rows.size, bool)
    sr = ["lor_land", "any_pair"][seed % 2]
    ui = np.flatnonzero(rng.random(n) < [0.3, 0.05, 0.7][seed % 3])
    uv = rng.random(ui.size) < (0.6 if seed % 4 == 1 else 1.1)  # (seed 1, 5, 9: some present entries are false)
    wi, wv = rand_vec(rng, n, 0.5, "BOOL")
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    accum = [None, "lor", None][seed % 3]
    comp, repl, struct = bool(seed & 1) or seed % 4 == 0, bool(seed & 2), bool(seed & 4)
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "BOOL")
    ou, ow, om = O.OVec(n, ui, uv, "BOOL"), O.OVec(n, wi, wv, "BOOL"), O.OVec(n, mi, mv, "BOOL")
    exp = O.mxv(oa, ou, sr, w=ow, mask=om, mask_comp=comp, mask_struct=struct, accum=accum, replace=repl)
    try:
        set_opts(ORDER_OPTS + ((b"rtile_rows", [8192, 16384][seed % 2]), (b"rtile_entries", [256, 900, 6000][seed % 3]),
                             (b"stream_nt_min_nnz", [1, 48 << 20][(seed >> 1) & 1])))  # (1: the tiles' streams are read non-temporal)
        A = gb.Matrix.from_coo(rows, cols, vals, dtype="BOOL", nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype="BOOL", size=n)
        w = gb.Vector.from_coo(wi, wv, dtype="BOOL", size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        m_arg = (mk.S if struct else mk.V)
        w(~m_arg if comp else m_arg, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        assert st["ordered"] == 1 and st["fused_epilogue"] == 3, st
        same_vec(w, exp)
        # the frontier as its own output (the notebook's q(~v.S, replace=True) << q.vxm(A, lor_land) shape, here with mxv): fresh buffers
        q = gb.Vector.from_coo(ui, uv, dtype="BOOL", size=n)
        q(~m_arg, replace=True) << A.mxv(q, getattr(gb.semiring, sr))
        same_vec(q, O.mxv(oa, ou, sr, w=ou, mask=om, mask_comp=True, mask_struct=struct, replace=True))
        set_opts(((b"rows_tile", 0),))
        w3 = gb.Vector.from_coo(wi, wv, dtype="BOOL", size=n)
        w3(~m_arg if comp else m_arg, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        assert device.last_stats()["fused_epilogue"] in (1, 2)
        same_vec(w3, exp)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(12))
def test_bottom_up_probe_of_long_bool_rows(gb, seed):
    """The bottom-up probe (k_long_init, round 5): a BOOL product under a terminal monoid tests the first entries of every admitted long row
    before the item kernels run; a row that finds a present (and, for lor.land, true) partner there is decided.  Frontiers from 0.1 % to
    90 %, present-but-false entries, an iso matrix value of FALSE (no row may be decided), ordered and natural-order layouts, every probe
    depth -- against the oracle, and the same with the probe switched off."""
    from graphblas_amd import device

    rng = np.random.default_rng(9900 + seed)
    n = int(rng.integers(2500, 5000))
    rows, cols, _ = skewed_square(rng, n, "BOOL")
    iso_false = seed == 7
    vals = np.zeros(rows.size, bool) if iso_false else np.ones(rows.size, bool)
    sr = "lor_land" if iso_false else ["lor_land", "any_pair"][seed % 2]
    ui = np.flatnonzero(rng.random(n) < [0.3, 0.02, 0.9, 0.001][seed % 4])
    uv = rng.random(ui.size) < (0.5 if seed % 3 == 2 else 1.1)  # (seed 2, 5, 8, 11: half of the present entries are false)
    mi, mv = rand_vec(rng, n, 0.5, "BOOL")
    oa = O.OMat.from_coo(rows, cols, vals, n, n, "BOOL")
    ou, om = O.OVec(n, ui, uv, "BOOL"), O.OVec(n, mi, mv, "BOOL")
    exp = O.mxv(oa, ou, sr, mask=om, mask_comp=True, mask_struct=True, replace=True)
    exp_nomask = O.mxv(oa, ou, sr)
    try:
        set_opts(ORDER_OPTS + ((b"order_mode", 0 if seed % 5 == 4 else 1),))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype="BOOL", nrows=n, ncols=n)
        u = gb.Vector.from_coo(ui, uv, dtype="BOOL", size=n)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=n)
        for depth in (8, 0, 16, 1):
            set_opts(((b"bool_probe", depth),))
            w = gb.Vector("BOOL", size=n)
            w(~mk.S, replace=True) << A.mxv(u, getattr(gb.semiring, sr))
            st = device.last_stats()
            assert st["long_kernel"] == 1 and st["long_probe"] == depth, st
            same_vec(w, exp)
            same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), exp_nomask)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(10))
def test_ranked_hint_orders_without_a_permutation(gb, seed):
    """GrX_Matrix_hint_ranked (round 5): a matrix whose labels the caller has ranked by popularity takes the ordered layouts in its own index
    order -- no permutation, no vector converted (pinned vectors included), square matrices and NON-square row blocks (what a rank of a
    sharded run holds after the graph was relabelled once).  Results against the oracle; the hint is a performance hint: a matrix that is NOT
    ranked but says so must give the same results."""
    from graphblas_amd import device

    rng = np.random.default_rng(6600 + seed)
    tname = ["FP32", "INT64", "BOOL", "FP64"][seed % 4]
    sr = {"FP32": "min_plus", "INT64": "plus_times", "BOOL": "lor_land", "FP64": "min_plus"}[tname]
    n = int(rng.integers(3000, 6000))
    rows, cols, vals = skewed_square(rng, n, tname)
    if seed % 5 != 4:  # relabel by falling column count (ties: row length) -- seed 4, 9: the hint on labels that are NOT ranked
        cnt = np.bincount(cols, minlength=n) * (n + 1) + np.bincount(rows, minlength=n)
        rank = np.empty(n, np.int64)
        rank[np.argsort(-cnt, kind="stable")] = np.arange(n)
        rows, cols = rank[rows], rank[cols]
    if tname == "BOOL":
        vals = np.ones(rows.size, bool)
    # a row block (block-cyclic: every third stripe of 256 rows) for the odd seeds, the square matrix for the even ones
    if seed & 1:
        keep = (rows // 256) % 3 == 1
        local = (rows[keep] // 256 // 3) * 256 + rows[keep] % 256
        m = int(local.max()) + 1
        rows, cols, vals = local, cols[keep], vals[keep]
    else:
        m = n
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    ui, uv = rand_vec(rng, n, [1.0, 0.4][seed % 2] if tname != "BOOL" else 0.3, tname)
    wi, wv = rand_vec(rng, m, 0.7, tname)
    mi, mv = rand_vec(rng, m, 0.5, "BOOL")
    accum = {"FP32": "min", "INT64": "plus", "BOOL": None, "FP64": "min"}[tname]
    repl = tname == "BOOL"
    oa = O.OMat.from_coo(rows, cols, vals, m, n, tname)
    ou, ow, om = O.OVec(n, ui, uv, tname), O.OVec(m, wi, wv, tname), O.OVec(m, mi, mv, "BOOL")
    exp = O.mxv(oa, ou, sr, w=ow, mask=om, mask_comp=True, mask_struct=True, accum=accum, replace=repl)
    try:
        set_opts(ORDER_OPTS + ((b"hub_min_len", [100, 0][seed % 2]), (b"rtile_entries", 700)))
        A = gb.Matrix.from_coo(rows, cols, vals, dtype=tname, nrows=m, ncols=n)
        device.matrix_hint_ranked(A)
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        if seed % 3 == 0:
            device.vector_device_views(u, "cuda" if _on_gpu() else "cpu")  # (a pinned operand: the ranked layouts take it all the same)
        w(~mk.S, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        st = device.last_stats()
        assert st["ordered"] == 1 and st["reorders"] == 0 and st["long_kernel"] == (1 if tname == "BOOL" else 4), st
        same_vec(w, exp)
        same_vec(A.mxv(u, getattr(gb.semiring, sr)).new(), O.mxv(oa, ou, sr))
        if m == n:  # vxm over the cached transpose, which shares the labels
            xi, xv = rand_vec(rng, n, 0.6, tname)
            x = gb.Vector.from_coo(xi, xv, dtype=tname, size=n)
            same_vec(x.vxm(A, getattr(gb.semiring, sr)).new(), O.vxm(O.OVec(n, xi, xv, tname), oa, sr))
            assert device.last_stats()["reorders"] == 0
        device.matrix_hint_ranked(A, False)  # the hint taken back: the matrix's own order (square) or the natural-order layouts (a block)
        w2 = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        w2(~mk.S, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        same_vec(w2, exp)
    finally:
        set_opts(RESTORE)


@pytest.mark.parametrize("seed", range(8))
def test_shard_setup_orders_a_row_block_by_global_column_counts(gb, seed):
    """GrX_Matrix_shard_setup (round 6): a NON-square row block on labels nobody ranked is handed the global reference counts of its columns;
    the library ranks the columns itself and runs the block on its popularity-ordered layouts, the operand kept in that column order, the
    output and the mask (the block's rows) natural.  Two blocks of one rank share one order object (`like`): the operand is converted once
    for both.  A pinned operand (an RCCL buffer) cannot be converted: the product runs the natural-order layouts and gives the same result.
    Everything against the oracle."""
    from graphblas_amd import device

    rng = np.random.default_rng(9900 + seed)
    tname = ["FP32", "INT64", "BOOL", "FP64"][seed % 4]
    sr = {"FP32": "min_plus", "INT64": "plus_times", "BOOL": "lor_land", "FP64": "min_plus"}[tname]
    n = int(rng.integers(3000, 6000))
    rows, cols, vals = skewed_square(rng, n, tname)
    if tname == "BOOL":
        vals = np.ones(rows.size, bool)
    counts = np.bincount(cols, minlength=n)  # (what the ranks' all-reduce of their histograms yields)
    cut = [0, n // 3 // 64 * 64, n]  # two row blocks of the square graph
    accum = {"FP32": "min", "INT64": "plus", "BOOL": None, "FP64": "min"}[tname]
    repl = tname == "BOOL"
    ui, uv = rand_vec(rng, n, [1.0, 0.4][seed % 2] if tname != "BOOL" else 0.3, tname)
    ou = O.OVec(n, ui, uv, tname)
    try:
        set_opts(ORDER_OPTS + ((b"hub_min_len", [100, 0][seed % 2]), (b"rtile_entries", 700), (b"hot_k", 256)))
        u = gb.Vector.from_coo(ui, uv, dtype=tname, size=n)
        blocks, expected = [], []
        for b in range(2):
            lo, hi = cut[b], cut[b + 1]
            keep = (rows >= lo) & (rows < hi)
            r, c, v = rows[keep] - lo, cols[keep], vals[keep]
            m = hi - lo
            A = gb.Matrix.from_coo(r, c, v, dtype=tname, nrows=m, ncols=n)
            if b == 0:
                device.matrix_shard_setup(A, counts if seed % 2 else counts.astype(np.uint32))
            else:
                device.matrix_shard_setup(A, like=blocks[0][0])
            wi, wv = rand_vec(rng, m, 0.7, tname)
            mi, mv = rand_vec(rng, m, 0.5, "BOOL")
            oa = O.OMat.from_coo(r, c, v, m, n, tname)
            exp = O.mxv(oa, ou, sr, w=O.OVec(m, wi, wv, tname), mask=O.OVec(m, mi, mv, "BOOL"), mask_comp=True, mask_struct=True, accum=accum, replace=repl)
            blocks.append((A, oa, (wi, wv), (mi, mv), m))
            expected.append(exp)
        total_reorders = 0
        for rep in range(2):
            for (A, oa, (wi, wv), (mi, mv), m), exp in zip(blocks, expected):
                w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
                mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
                w(~mk.S, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
                st = device.last_stats()
                assert st["ordered"] == 1 and st["long_kernel"] == (1 if tname == "BOOL" else 4), st
                total_reorders += st["reorders"]
                same_vec(w, exp)
        assert total_reorders == 1  # (the operand was converted once, for both blocks and both rounds)
        # the operand read back is the natural one; a plain product too
        gi, gv = u.to_coo()
        assert gi.tolist() == ui.tolist() and gv.tolist() == uv.tolist()
        same_vec(blocks[1][0].mxv(u, getattr(gb.semiring, sr)).new(), O.mxv(blocks[1][1], ou, sr))
        # a pinned operand: natural-order layouts, same result
        device.vector_device_views(u, "cuda" if _on_gpu() else "cpu")
        A, oa, (wi, wv), (mi, mv), m = blocks[0]
        w = gb.Vector.from_coo(wi, wv, dtype=tname, size=m)
        mk = gb.Vector.from_coo(mi, mv, dtype="BOOL", size=m)
        w(~mk.S, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        assert device.last_stats()["ordered"] == 0 and device.last_stats()["pinned_natural"] == 1
        same_vec(w, expected[0])
    finally:
        set_opts(RESTORE)


def test_shard_setup_argument_errors(gb):
    """GrX_Matrix_shard_setup rejects what it cannot use and leaves the matrix as it was: no counts and no `like`, counts of the wrong length
    (host layer), a `like` that carries no column order or another width; a matrix that was set up and is then modified forgets the order."""
    import ctypes

    from graphblas_amd import _lib, device, exceptions

    rng = np.random.default_rng(77)
    n = 2000
    rows, cols, vals = skewed_square(rng, n, "FP32")
    A = gb.Matrix.from_coo(rows, cols, vals, dtype="FP32", nrows=n, ncols=n)
    B = gb.Matrix.from_coo(rows, cols, vals, dtype="FP32", nrows=n, ncols=n)
    C = gb.Matrix.from_coo([0, 1], [1, 0], [1.0, 2.0], dtype="FP32", nrows=100, ncols=100)
    assert _lib.lib.GrX_Matrix_shard_setup(A._carg, None, 0, None) != 0  # (GrB_NULL_POINTER: neither counts nor `like`)
    with pytest.raises(ValueError, match="one count per column"):
        device.matrix_shard_setup(A, np.ones(n - 1, np.uint32))
    with pytest.raises(exceptions.InvalidValue):  # (B was never set up: it carries no column order)
        device.matrix_shard_setup(A, like=B)
    device.matrix_shard_setup(B, np.bincount(cols, minlength=n).astype(np.uint32))
    with pytest.raises(exceptions.InvalidValue):  # (another width)
        device.matrix_shard_setup(C, like=B)
    device.matrix_shard_setup(A, like=B)  # (and this one is fine)
    ui, uv = rand_vec(rng, n, 1.0, "FP32")
    u = gb.Vector.from_coo(ui, uv, dtype="FP32", size=n)
    exp = O.mxv(O.OMat.from_coo(rows, cols, vals, n, n, "FP32"), O.OVec(n, ui, uv, "FP32"), "min_plus")
    same_vec(A.mxv(u, gb.semiring.min_plus).new(), exp)
    A.resize(n, n)  # (a modification drops A's cached layouts and its share of the order; B keeps its own)
    same_vec(A.mxv(u, gb.semiring.min_plus).new(), exp)
    same_vec(B.mxv(u, gb.semiring.min_plus).new(), exp)
