"""GPU tier, BASELINE.json sizes: the HIP path on R-MAT graphs against (a) the CPU oracle at sizes the
oracle finishes in seconds and (b) size-independent properties at full scale:
  * linearity in the mask: the masked result equals the unmasked result restricted to the mask;
  * mxv(A, u) == vxm(u, A') (transpose cache) ;
  * idempotence of a min.+ relaxation with accum=min once converged (SSSP fixed point = Dijkstra via scipy);
  * BFS by lor_land level steps reaches exactly the vertices scipy's BFS reaches, level by level.
All through the C ABI (the Python host is a thin ctypes layer)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gb():
    from tests.backend import bind

    return bind("gpu")


def _graph(gb, scale, kind):
    import torch

    from graphblas_amd import device, synthetic

    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    n = 1 << scale
    if kind == "FP32":
        vals = synthetic.edge_weights(col, scale)
        A = device.matrix_from_device_csr(indptr, col, vals, n, n, "FP32")
    elif kind == "INT64":
        vals = synthetic.edge_weights(col, scale, dtype=torch.int64)
        A = device.matrix_from_device_csr(indptr, col, vals, n, n, "INT64")
    else:
        vals = torch.ones(1, dtype=torch.bool, device="cuda")
        A = device.matrix_from_device_csr(indptr, col, vals, n, n, "BOOL", iso=True)
    return A, indptr, col, vals


@pytest.mark.parametrize("scale,kind,sr", [(16, "FP32", "min_plus"), (17, "BOOL", "lor_land"), (16, "INT64", "plus_times"),
                                           (16, "FP32", "plus_times"), (16, "BOOL", "any_pair"), (20, "FP32", "min_plus")])
def test_rmat_vs_oracle(gb, scale, kind, sr):
    """configs[1]-style graphs at oracle-friendly sizes -- and configs[1] itself (R-MAT scale 20, FP32 min_plus: 16 M entries, the
    oracle takes a second per call) -- bit-exact against the CPU oracle."""
    import torch

    from graphblas_amd import device
    from oracle import grb_oracle as O

    A, indptr, col, vals = _graph(gb, scale, kind)
    n = 1 << scale
    ip, cj = indptr.cpu().numpy(), col.cpu().numpy().astype(np.int64)
    nnz = cj.size
    hv = np.ones(nnz, bool) if kind == "BOOL" else vals.cpu().numpy()
    oa = O.OMat(n, n, ip, cj, hv, kind)
    rng = np.random.default_rng(scale)
    for dens, vis_d, accum, repl in ((1.0, 0.5, "min" if sr == "min_plus" else None, False), (0.01, 0.9, None, True),
                                     (0.3, 0.0, None, False)):
        ui = np.flatnonzero(rng.random(n) < dens)
        # small integer values: every partial sum stays below 2^24, so even fp32 plus_times is exact in any order
        uv = (rng.random(ui.size) < 0.9) if kind == "BOOL" else rng.integers(1, 3 if kind == "FP32" else 50, ui.size).astype(O.NP_OF[kind])
        vi = np.flatnonzero(rng.random(n) < vis_d)
        u, ou = gb.Vector.from_coo(ui, uv, dtype=kind, size=n), O.OVec(n, ui, uv, kind)
        vis, ovis = gb.Vector.from_coo(vi, np.ones(vi.size, bool), dtype="BOOL", size=n), O.OVec(n, vi, np.ones(vi.size, bool), "BOOL")
        w = u.dup()
        w(~vis.S, accum=accum, replace=repl) << A.mxv(u, getattr(gb.semiring, sr))
        exp = O.mxv(oa, ou, sr, w=ou, mask=ovis, mask_comp=True, mask_struct=True, accum=accum, replace=repl)
        gi, gv = w.to_coo()
        assert np.array_equal(gi.astype(np.int64), exp.idx)
        if kind == "FP32" and sr == "plus_times":
            np.testing.assert_allclose(gv, exp.vals, rtol=1e-6)  # fp plus: order of summation differs (north-star tolerance)
        else:
            assert np.array_equal(gv, exp.vals)
        # vxm over the cached transpose must agree with mxv
        w2 = u.dup()
        w2(~vis.S, accum=accum, replace=repl) << u.vxm(A.T, getattr(gb.semiring, sr))
        assert w2.isclose(w, rel_tol=1e-6) if (kind == "FP32" and sr == "plus_times") else w2.isequal(w)


def test_sssp_fixed_point_matches_dijkstra(gb):
    """primer.rst:221-251 loop at R-MAT scale 15: v(min) << min_plus(v @ G) until fixed point == scipy Dijkstra."""
    sp = pytest.importorskip("scipy.sparse")
    from scipy.sparse.csgraph import dijkstra

    scale = 15
    A, indptr, col, vals = _graph(gb, scale, "FP32")
    n = 1 << scale
    G = sp.csr_matrix((vals.cpu().numpy().astype(np.float64), col.cpu().numpy(), indptr.cpu().numpy()), shape=(n, n))
    src = int(np.argmax(np.diff(indptr.cpu().numpy())))
    ref = dijkstra(G, directed=True, indices=src)
    v = gb.Vector.from_coo([src], [0.0], dtype="FP32", size=n)
    for it in range(200):
        before = v.nvals, None
        prev = v.dup()
        v(gb.op.min) << gb.semiring.min_plus(v @ A)
        if v.isequal(prev):
            break
    assert it < 199
    idx, val = v.to_coo()
    reach = np.flatnonzero(np.isfinite(ref))
    assert np.array_equal(idx.astype(np.int64), reach)
    assert np.array_equal(val.astype(np.float64), ref[reach])  # integer weights: exact in fp32
    # idempotence once converged
    again = v.dup()
    again(gb.op.min) << gb.semiring.min_plus(again @ A)
    assert again.isequal(v)


def test_bfs_levels_match_scipy(gb):
    """Level BFS with  q(~visited.S, replace) << q.vxm(A, lor_land)  (notebooks/Example B.1) at scale 16."""
    sp = pytest.importorskip("scipy.sparse")
    from scipy.sparse.csgraph import breadth_first_order

    scale = 16
    A, indptr, col, _ = _graph(gb, scale, "BOOL")
    n = 1 << scale
    ip, cj = indptr.cpu().numpy(), col.cpu().numpy()
    G = sp.csr_matrix((np.ones(cj.size, np.int8), cj, ip), shape=(n, n))
    src = int(np.argmax(np.diff(ip)))
    order, pred = breadth_first_order(G, src, directed=True, return_predecessors=True)
    level = np.full(n, -1)
    level[src] = 0
    for x in order[1:]:
        level[x] = level[pred[x]] + 1
    # the loop of notebooks/Example B.1 -- Level BFS.ipynb, entirely on the device: the level vector doubles as the visited set
    v = gb.Vector("INT32", n)
    q = gb.Vector(bool, n)
    q[src] << True
    succ = gb.Scalar(bool)
    d = 0
    while True:
        d += 1
        v[:](mask=q.V) << d
        q(~v.S, replace=True) << q.vxm(A, gb.semiring.lor_land)
        succ << q.reduce(gb.monoid.lor, allow_empty=False)
        if not succ:
            break
    idx, lev = v.to_coo()
    assert np.array_equal(idx.astype(np.int64), np.flatnonzero(level >= 0))
    assert np.array_equal(lev.astype(np.int64) - 1, level[level >= 0])
    assert d - 1 == level.max()


def test_full_scale_properties(gb):
    """R-MAT scale-22 (configs-sized rows; scale-24 runs in bench.py): mask restriction + transpose agreement."""
    import torch

    from graphblas_amd import device

    scale = 22
    A, indptr, col, vals = _graph(gb, scale, "FP32")
    n = 1 << scale
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    dist = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
    present = torch.rand(n, generator=gen, device="cuda") < 0.5
    visited = torch.rand(n, generator=gen, device="cuda") < 0.5
    u = device.vector_from_device(dist, present=present)
    vis = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=visited)
    full = A.mxv(u, gb.semiring.min_plus).new()
    masked = A.mxv(u, gb.semiring.min_plus).new(mask=~vis.S)
    fv, fb = device.vector_device_views(full)
    mv, mb = device.vector_device_views(masked)
    fbits = torch.from_numpy(np.unpackbits(fb.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)).cuda()
    mbits = torch.from_numpy(np.unpackbits(mb.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)).cuda()
    assert torch.equal(mbits, fbits & ~visited)
    assert torch.equal(mv[mbits], fv[mbits])
    # independent check of the unmasked product with torch segment reductions (same fp32 adds, min is exact)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), indptr[1:] - indptr[:-1])
    cl = col.long()
    ok = present[cl]
    cand = torch.where(ok, vals + dist[cl], torch.full_like(vals, float("inf")))
    ref = torch.full((n,), float("inf"), device="cuda").scatter_reduce(0, rows, cand, "amin")
    ref_has = torch.zeros(n, dtype=torch.bool, device="cuda").index_put_((rows[ok],), torch.tensor(True, device="cuda"))
    assert torch.equal(fbits, ref_has)
    assert torch.equal(fv[fbits], ref[fbits])
    # transpose agreement: (A')' u via vxm
    t = u.vxm(A.T, gb.semiring.min_plus).new()
    tv, tb = device.vector_device_views(t)
    assert torch.equal(tb, fb) and torch.equal(tv[fbits], fv[fbits])


@pytest.mark.parametrize("scale", [10, 13])
def test_mxm_rmat_vs_scipy(gb, scale):
    """configs[3]-style A@A (plus_times INT64, ones) against scipy's csr @ csr -- integer, bit-exact -- plus the
    masked (triangle-count style) C<A.S> = A@A against the oracle's write rule."""
    sp = pytest.importorskip("scipy.sparse")
    import torch

    from graphblas_amd import device, synthetic
    from oracle import grb_oracle as O

    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, one, n, n, "INT64", iso=True)
    ip, cj = indptr.cpu().numpy(), col.cpu().numpy()
    S = sp.csr_matrix((np.ones(cj.size, np.int64), cj, ip), shape=(n, n))
    ref = (S @ S).tocsr()
    ref.sort_indices()
    C = A.mxm(A, gb.semiring.plus_times).new()
    st = device.last_stats()
    assert st["out_nvals"] == ref.nnz
    Cp, Cj, Cx = C.to_csr()
    assert np.array_equal(Cp.astype(np.int64), ref.indptr) and np.array_equal(Cj.astype(np.int64), ref.indices)
    assert np.array_equal(Cx, ref.data)
    M = A.mxm(A, gb.semiring.plus_times).new(mask=A.S)
    oa = O.OMat(n, n, ip, cj.astype(np.int64), np.ones(cj.size, np.int64), "INT64")
    exp = O.mxm(oa, oa, "plus_times", mask=oa, mask_struct=True)
    Mp, Mj, Mx = M.to_csr()
    assert np.array_equal(Mp.astype(np.int64), exp.indptr) and np.array_equal(Mj.astype(np.int64), exp.indices)
    assert np.array_equal(Mx, exp.values)


@pytest.mark.parametrize("scale,weighted", [(18, False), (17, True)])
def test_mxm_unit_kernels_against_row_kernels_at_scale(gb, scale, weighted):
    """configs[3] sizes beyond what scipy multiplies in seconds: the (row, column window) unit kernels of round 2 against the
    1024-thread row kernels of round 1 (two independent implementations of the heavy rows), with other class limits and without
    the bitmap pool -- identical matrices, compared on the device; for a matrix of ones the sum of the product's values equals
    the number of multiplies (GrX_mxm_streamed's checksum, one batch and many)."""
    import ctypes

    import torch

    from graphblas_amd import _lib, device, synthetic

    L = _lib.lib
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    if weighted:
        vals = synthetic.edge_weights(col, scale, dtype=torch.int64)
        A = device.matrix_from_device_csr(indptr, col, vals, n, n, "INT64")
    else:
        A = device.matrix_from_device_csr(indptr, col, torch.ones(1, dtype=torch.int64, device="cuda"), n, n, "INT64", iso=True)
    sr = gb.semiring.plus_times if not weighted else gb.semiring.min_plus

    def product(**opts):
        try:
            for k, v in opts.items():
                L.GrX_option_set(k.encode(), v)
            C = A.mxm(A, sr).new()
            return C, device.last_stats()
        finally:
            for k, v in dict(mxm_heavy_kernel=1, mxm_unit_small=512, mxm_unit_mid=1024, mxm_unit_dense=4096,
                             mxm_bitmap_pool_cap=(1 << 31) - 1, mxm_unit_min_flops=1024).items():
                L.GrX_option_set(k.encode(), v)

    C1, st1 = product()
    C0, st0 = product(mxm_heavy_kernel=0)
    assert st0["out_nvals"] == st1["out_nvals"] and st0["flops"] == st1["flops"] and st1["out_nvals"] > 100 * n
    assert C1.isequal(C0)
    del C0
    C2, _ = product(mxm_unit_small=64, mxm_unit_mid=300, mxm_unit_dense=1500, mxm_bitmap_pool_cap=5000, mxm_unit_min_flops=128)
    assert C1.isequal(C2)
    del C2
    C3, _ = product(mxm_bitmap_pool_cap=0, mxm_unit_min_flops=16384)
    assert C1.isequal(C3)
    del C3
    if not weighted:
        for budget in (1 << 40, 2 << 30):
            nv, cs, fl, nb = (ctypes.c_uint64() for _ in range(4))
            rc = L.GrX_mxm_streamed(gb.semiring.plus_times["INT64"]._carg, A._carg, A._carg, ctypes.c_uint64(budget), ctypes.byref(nv),
                                    ctypes.byref(cs), ctypes.byref(fl), ctypes.byref(nb))
            assert rc == 0
            assert nv.value == st1["out_nvals"] and fl.value == st1["flops"] and cs.value == st1["flops"]
            assert (nb.value == 1) == (budget == 1 << 40)
        # the mask-driven product: units keyed by the mask row against the row kernels, and against the full product under the mask
        M1 = A.mxm(A, sr).new(mask=A.S)
        try:
            L.GrX_option_set(b"mxm_heavy_kernel", 0)
            M0 = A.mxm(A, sr).new(mask=A.S)
        finally:
            L.GrX_option_set(b"mxm_heavy_kernel", 1)
        assert M1.isequal(M0)
        try:
            L.GrX_option_set(b"mxm_mask_mode", 0)  # (the full product, then the write rule)
            M2 = A.mxm(A, sr).new(mask=A.S)
        finally:
            L.GrX_option_set(b"mxm_mask_mode", 1)
        assert M1.isequal(M2) and M1.nvals > n


@pytest.mark.gpu
def test_reduce_rowwise_is_the_degree_vector(gb):
    """A.reduce_rowwise(plus) over an iso-ones R-MAT matrix = the out-degrees, reduce_columnwise = the in-degrees (scale 18):
    the pull SpMV over (plus, first) with an operand that is never read."""
    import torch

    from graphblas_amd import device, synthetic

    scale = 18
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, one, n, n, "INT64", iso=True)
    out_deg = (indptr[1:] - indptr[:-1]).cpu().numpy()
    in_deg = torch.bincount(col.long(), minlength=n).cpu().numpy()
    w = A.reduce_rowwise(gb.monoid.plus).new()
    i, x = w.to_coo()
    assert np.array_equal(i.astype(np.int64), np.flatnonzero(out_deg)) and np.array_equal(x, out_deg[out_deg > 0])
    w = A.reduce_columnwise(gb.monoid.plus).new()
    i, x = w.to_coo()
    assert np.array_equal(i.astype(np.int64), np.flatnonzero(in_deg)) and np.array_equal(x, in_deg[in_deg > 0])
    assert w.reduce(gb.monoid.plus).new() == int(col.numel())


@pytest.mark.gpu
def test_count_aggregator_and_power_at_scale(gb):
    """``agg.count`` (the plus_pair mat-vec with a dense iso operand) = the degree vectors at scale 18; ``A.power(5)`` over
    plus_times INT64 on a scale-9 R-MAT graph against scipy's repeated product (integer, bit-exact)."""
    sp = pytest.importorskip("scipy.sparse")
    import torch

    from graphblas_amd import device, synthetic

    scale = 18
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    w8 = torch.randint(1, 200, (col.numel(),), dtype=torch.float32, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, w8, n, n, "FP32")
    out_deg = (indptr[1:] - indptr[:-1]).cpu().numpy()
    in_deg = torch.bincount(col.long(), minlength=n).cpu().numpy()
    w = A.reduce_rowwise(gb.agg.count).new()
    assert w.dtype == gb.dtypes.INT64
    i, x = w.to_coo()
    assert np.array_equal(i.astype(np.int64), np.flatnonzero(out_deg)) and np.array_equal(x, out_deg[out_deg > 0])
    i, x = A.reduce_columnwise(gb.agg.count).new().to_coo()
    assert np.array_equal(i.astype(np.int64), np.flatnonzero(in_deg)) and np.array_equal(x, in_deg[in_deg > 0])
    i, x = A.reduce_columnwise(gb.agg.exists).new().to_coo()
    assert np.array_equal(i.astype(np.int64), np.flatnonzero(in_deg)) and np.all(x == 1)

    scale = 9
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    B = device.matrix_from_device_csr(indptr, col, one, n, n, "INT64", iso=True)
    ip, cj = indptr.cpu().numpy(), col.cpu().numpy()
    S = sp.csr_matrix((np.ones(cj.size, np.int64), cj, ip), shape=(n, n))
    ref = S
    for _ in range(4):
        ref = (ref @ S).tocsr()
    ref.sort_indices()
    Cp, Cj, Cx = B.power(5).new().to_csr()
    assert np.array_equal(Cp.astype(np.int64), ref.indptr) and np.array_equal(Cj.astype(np.int64), ref.indices)
    assert np.array_equal(Cx, ref.data)


@pytest.mark.gpu
@pytest.mark.parametrize("scale,world,rank", [(22, 8, 5), (24, 16, 11)])
def test_row_block_of_a_sharded_graph(gb, scale, world, rank):
    """(scale 24: a 16th of the headline graph -- 1 Mi x 16 Mi, 16 M entries, the layout of the bench call with 2 sub-ranges per
    class -- against the CPU oracle.)  One rank's share of the 8-way row-sharded scale-22 graph (a 524 288 x 4 194 304 block: hot-column table, long / short row
    split, sub-ranged long rows -- the shapes `bench.py --gpus 8` runs): the masked min_plus and lor_land steps of the block are
    bit-exact against the CPU oracle and equal to the block's rows of the single-GPU product."""
    import torch

    from graphblas_amd import device, synthetic
    from oracle import grb_oracle as O

    n = 1 << scale
    rows = n // world
    lo, hi = rank * rows, (rank + 1) * rows
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    vals = synthetic.edge_weights(col, scale)
    e0, e1 = int(indptr[lo].item()), int(indptr[hi].item())
    ip_b = (indptr[lo:hi + 1] - indptr[lo]).contiguous()
    col_b, val_b = col[e0:e1].contiguous(), vals[e0:e1].contiguous()
    A = device.matrix_from_device_csr(indptr, col, vals, n, n, "FP32")
    B = device.matrix_from_device_csr(ip_b, col_b, val_b, rows, n, "FP32")
    rng = np.random.default_rng(scale)
    uv = rng.integers(0, 1000, n).astype(np.float32)
    vi = np.flatnonzero(rng.random(n) < 0.5)
    u = gb.Vector.from_coo(np.arange(n), uv, dtype="FP32", size=n)
    vis = gb.Vector.from_coo(vi, np.ones(vi.size, bool), dtype="BOOL", size=n)
    w_full = u.dup()
    w_full(~vis.S, accum="min") << A.mxv(u, gb.semiring.min_plus)
    vb = vi[(vi >= lo) & (vi < hi)] - lo
    vis_b = gb.Vector.from_coo(vb, np.ones(vb.size, bool), dtype="BOOL", size=rows)
    fi, fv = w_full.to_coo()
    for call in range(2):  # the first product of a large matrix runs on the plain CSR arrays, the second on the cached layouts
        w_b = gb.Vector.from_coo(np.arange(rows), uv[lo:hi], dtype="FP32", size=rows)
        w_b(~vis_b.S, accum="min") << B.mxv(u, gb.semiring.min_plus)
        assert (device.last_stats()["hot_k"] > 0) == (call == 1)
        gi, gv = w_b.to_coo()
        assert np.array_equal(gi, np.arange(rows, dtype=gi.dtype)) and np.array_equal(gv, fv[lo:hi])
    ob = O.OMat(rows, n, ip_b.cpu().numpy(), col_b.cpu().numpy().astype(np.int64), val_b.cpu().numpy(), "FP32")
    exp = O.mxv(ob, O.OVec(n, np.arange(n), uv, "FP32"), "min_plus", w=O.OVec(rows, np.arange(rows), uv[lo:hi], "FP32"),
                mask=O.OVec(rows, vb, np.ones(vb.size, bool), "BOOL"), mask_comp=True, mask_struct=True, accum="min")
    assert np.array_equal(gv, exp.vals)
    # the BFS step on the same block (iso-True BOOL, frontier of density 0.3, replace)
    one = torch.ones(1, dtype=torch.bool, device="cuda")
    Bb = device.matrix_from_device_csr(ip_b, col_b, one, rows, n, "BOOL", iso=True)
    qi = np.flatnonzero(rng.random(n) < 0.3)
    q = gb.Vector.from_coo(qi, np.ones(qi.size, bool), dtype="BOOL", size=n)
    for call in range(2):
        nxt = gb.Vector("BOOL", size=rows)
        nxt(~vis_b.S, replace=True) << Bb.mxv(q, gb.semiring.lor_land)
        if call == 0:
            first_bfs = nxt.to_coo()
    assert np.array_equal(first_bfs[0], nxt.to_coo()[0])
    obb = O.OMat(rows, n, ip_b.cpu().numpy(), col_b.cpu().numpy().astype(np.int64), np.ones(col_b.numel(), bool), "BOOL")
    expb = O.mxv(obb, O.OVec(n, qi, np.ones(qi.size, bool), "BOOL"), "lor_land", mask=O.OVec(rows, vb, np.ones(vb.size, bool), "BOOL"),
                 mask_comp=True, mask_struct=True, replace=True)
    bi, bv = nxt.to_coo()
    assert np.array_equal(bi.astype(np.int64), expb.idx) and np.array_equal(bv, expb.vals)



@pytest.mark.gpu
def test_scale24_headline_calls(gb):
    """The two calls bench.py times, at full size (R-MAT scale 24, 263 M entries): the masked SSSP relaxation
    w<~visited.S> = min(w, A min.+ u) and the BFS level step q<~visited.S, replace> = A lor.land q, checked against torch
    segment reductions of the same products (min over integer-valued fp32 sums is exact: bit for bit); then the relaxation
    once more with real-valued U[0,1) weights and distances (fp32 sums of two terms, min exact: still bit for bit)."""
    import torch

    from graphblas_amd import device, synthetic
    from oracle import grb_oracle as O

    scale = 24
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), indptr[1:] - indptr[:-1])
    cl = col.long()
    ip_np, cj_np = indptr.cpu().numpy(), col.cpu().numpy().astype(np.int64)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    visited = torch.rand(n, generator=gen, device="cuda") < 0.5
    vis = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=visited)

    def bits(words):
        return torch.from_numpy(np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)).cuda()

    for weights in ("integer", "real"):
        if weights == "integer":
            vals = synthetic.edge_weights(col, scale)
            dist = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
        else:
            vals = torch.rand(col.numel(), generator=gen, device="cuda", dtype=torch.float32)
            dist = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32) * 8
        A = device.matrix_from_device_csr(indptr, col, vals, n, n, "FP32")
        u = device.vector_from_device(dist)
        ref = torch.full((n,), float("inf"), device="cuda").scatter_reduce(0, rows, vals + dist[cl], "amin")
        exp = torch.where(~visited, torch.minimum(dist, ref), dist)
        # the C oracle on the same operands (one full-size pass, a fraction of a second on the box's cores): the checker the small-size
        # parity tests use, here at the size the headline is quoted on (the full-size call was only put against torch)
        O.use_all_threads()
        t_has = np.zeros(n, np.uint8)
        t_val = np.zeros(n, np.float32)
        act = (~visited).cpu().numpy().astype(np.uint8)
        rc = O.lib().grbo_mxv(O.TYPE_CODES["FP32"], O.OP_CODES["min"], O.OP_CODES["plus"], ctypes.c_int64(n), O._p(ip_np), O._p(cj_np),
                              O._p(np.ascontiguousarray(vals.cpu().numpy())), 0, O._p(np.ones(n, np.uint8)), O._p(np.ascontiguousarray(dist.cpu().numpy())),
                              O._p(act), O._p(t_has), O._p(t_val))
        assert rc == 0
        d_np = dist.cpu().numpy()
        exp_c = np.where((act != 0) & (t_has != 0), np.minimum(d_np, t_val), d_np)
        assert np.array_equal(exp_c, exp.cpu().numpy()), "the C oracle and the torch reduction disagree"
        for call in range(3):  # first product: plain CSR arrays; second and third: the cached layouts (the third is what bench.py times)
            w = device.vector_from_device(dist.clone())
            w(~vis.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
            st = device.last_stats()
            if call >= 1:
                assert st["long_kernel"] == 4 and st["long_entries"] > 100_000_000 and st["hot_k"] > 0
                # ... on the popularity-ordered twin, the short rows as sorted row tiles with the write rule fused (k_mxv_rtile): the kernels
                # of the bench line are the ones that produced the result checked below
                assert st["ordered"] == 1 and st["fused_epilogue"] == 3, st
                assert st["value_dict"] == (255 if weights == "integer" else 0), st
            else:
                assert st["long_kernel"] == -1 and st["hot_k"] == 0
            wv, wb = device.vector_device_views(w, pin=False)
            assert bool(bits(wb).all()) and torch.equal(wv, exp)
            assert np.array_equal(wv.cpu().numpy(), exp_c)
        del A, u, w, ref, exp
    one = torch.ones(1, dtype=torch.bool, device="cuda")
    Ab = device.matrix_from_device_csr(indptr, col, one, n, n, "BOOL", iso=True)
    frontier = torch.rand(n, generator=gen, device="cuda") < 0.3
    q = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=frontier)
    hit = torch.zeros(n, dtype=torch.bool, device="cuda")
    hit.index_put_((rows[frontier[cl]],), torch.tensor(True, device="cuda"))
    for call in range(2):
        nxt = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=frontier)
        nxt(~vis.S, replace=True) << Ab.mxv(q, gb.semiring.lor_land)
        assert device.last_stats()["long_kernel"] == (1 if call else -1)
        nv, nb = device.vector_device_views(nxt)
        got = bits(nb)
        assert torch.equal(got, hit & ~visited) and bool(nv[got].all())


@pytest.mark.gpu
def test_scale24_ranked_hint_call(gb):
    """The headline call on a graph whose application relabelled its vertices by popularity once (synthetic.rmat_csr(relabel="popularity"))
    and says so (GrX_Matrix_hint_ranked, `bench.py --ranked`), at full size against the C oracle: the ordered layouts are built in the
    caller's own labels -- no permutation, an operand whose pointers were handed out (pinned) is taken as it is."""
    import torch

    from graphblas_amd import device, synthetic
    from oracle import grb_oracle as O

    scale = 24
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda", relabel="popularity")
    vals = synthetic.edge_weights(col, scale)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    visited = torch.rand(n, generator=gen, device="cuda") < 0.5
    dist = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
    vis = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=visited)
    A = device.matrix_from_device_csr(indptr, col, vals, n, n, "FP32")
    device.matrix_hint_ranked(A)
    u = device.vector_from_device(dist)
    device.vector_device_views(u)  # (pinned natural: what a vector that RCCL gathers into looks like)
    O.use_all_threads()
    t_has, t_val = np.zeros(n, np.uint8), np.zeros(n, np.float32)
    act = (~visited).cpu().numpy().astype(np.uint8)
    d_np = dist.cpu().numpy()
    rc = O.lib().grbo_mxv(O.TYPE_CODES["FP32"], O.OP_CODES["min"], O.OP_CODES["plus"], ctypes.c_int64(n), O._p(indptr.cpu().numpy()),
                          O._p(col.cpu().numpy().astype(np.int64)), O._p(np.ascontiguousarray(vals.cpu().numpy())), 0, O._p(np.ones(n, np.uint8)),
                          O._p(d_np), O._p(act), O._p(t_has), O._p(t_val))
    assert rc == 0
    exp = np.where((act != 0) & (t_has != 0), np.minimum(d_np, t_val), d_np)
    for call in range(3):
        w = device.vector_from_device(dist.clone())
        w(~vis.S, accum=gb.binary.min) << A.mxv(u, gb.semiring.min_plus)
        st = device.last_stats()
        if call >= 1:
            assert st["long_kernel"] == 4 and st["fused_epilogue"] == 3 and st["reorders"] == 0, st
        wv, _wb = device.vector_device_views(w, pin=False)
        assert np.array_equal(wv.cpu().numpy(), exp)


@pytest.mark.gpu
def test_kron26_rank_block(gb):
    """configs[4] (Kronecker scale 26, min_plus FP32, 8 GPUs): rank 0's row block of the 8-way run -- 8 388 608 x 67 108 864,
    ~134 M entries, generated from the same edge stream every rank draws -- against the CPU oracle, bit for bit: the first call on
    the plain CSR arrays, the second on the cached layouts (hot-column table over a 256 MiB operand, class strips), unmasked as
    `bench.py --workload kron26` runs it and under a complemented structural mask; then the BFS step of the same block."""
    import torch

    from graphblas_amd import device, synthetic
    from oracle import grb_oracle as O

    scale, world, rank = 26, 8, 0
    n = 1 << scale
    rows = n // world
    lo, hi = rank * rows, (rank + 1) * rows
    ip_b, col_b = synthetic.rmat_csr(scale, device="cuda", row_range=(lo, hi))
    assert ip_b.numel() == rows + 1 and 100_000_000 < col_b.numel() < 180_000_000
    val_b = synthetic.edge_weights(col_b, scale)
    B = device.matrix_from_device_csr(ip_b, col_b, val_b, rows, n, "FP32")
    rng = np.random.default_rng(scale)
    uv = rng.integers(0, 1000, n).astype(np.float32)
    vb = np.flatnonzero(rng.random(rows) < 0.5)
    u = device.vector_from_device(torch.from_numpy(uv).cuda())
    vis_b = gb.Vector.from_coo(vb, np.ones(vb.size, bool), dtype="BOOL", size=rows)
    ob = O.OMat(rows, n, ip_b.cpu().numpy(), col_b.cpu().numpy().astype(np.int64), val_b.cpu().numpy(), "FP32")
    ou = O.OVec(n, np.arange(n), uv, "FP32")
    ow = O.OVec(rows, np.arange(rows), uv[lo:hi], "FP32")
    exp_plain = O.mxv(ob, ou, "min_plus", w=ow, accum="min")
    exp_masked = O.mxv(ob, ou, "min_plus", w=ow, mask=O.OVec(rows, vb, np.ones(vb.size, bool), "BOOL"), mask_comp=True, mask_struct=True,
                       accum="min")
    for call in range(2):
        w_b = device.vector_from_device(torch.from_numpy(uv[lo:hi].copy()).cuda())
        w_b(accum="min") << B.mxv(u, gb.semiring.min_plus)
        st = device.last_stats()
        assert (st["hot_k"] > 0) == (call == 1) and (st["long_kernel"] == 4) == (call == 1)
        wv, _ = device.vector_device_views(w_b)
        assert np.array_equal(wv.cpu().numpy(), exp_plain.vals)
    w_b = device.vector_from_device(torch.from_numpy(uv[lo:hi].copy()).cuda())
    w_b(~vis_b.S, accum="min") << B.mxv(u, gb.semiring.min_plus)
    wv, _ = device.vector_device_views(w_b)
    assert np.array_equal(wv.cpu().numpy(), exp_masked.vals) and w_b.nvals == rows
    del B, w_b
    one = torch.ones(1, dtype=torch.bool, device="cuda")
    Bb = device.matrix_from_device_csr(ip_b, col_b, one, rows, n, "BOOL", iso=True)
    qi = np.flatnonzero(rng.random(n) < 0.3)
    q = gb.Vector.from_coo(qi, np.ones(qi.size, bool), dtype="BOOL", size=n)
    for call in range(2):
        nxt = gb.Vector("BOOL", size=rows)
        nxt(~vis_b.S, replace=True) << Bb.mxv(q, gb.semiring.lor_land)
    obb = O.OMat(rows, n, ob.indptr, ob.indices, np.ones(col_b.numel(), bool), "BOOL")
    expb = O.mxv(obb, O.OVec(n, qi, np.ones(qi.size, bool), "BOOL"), "lor_land", mask=O.OVec(rows, vb, np.ones(vb.size, bool), "BOOL"),
                 mask_comp=True, mask_struct=True, replace=True)
    bi, bv = nxt.to_coo()
    assert np.array_equal(bi.astype(np.int64), expb.idx) and np.array_equal(bv, expb.vals)


@pytest.mark.gpu
def test_mxm_scale16_vs_scipy_on_the_host(gb):
    """SpGEMM at a size where the (row, column window) unit kernels, the dense units and the hash kernels all run (R-MAT scale 16:
    0.96 M entries, 0.4 G multiplies, 165 M entries in C), against scipy's csr @ csr computed on the host -- an implementation that
    shares nothing with this library; INT64 ones, so the comparison is bit for bit over the whole CSR."""
    sp = pytest.importorskip("scipy.sparse")
    import torch

    from graphblas_amd import device, synthetic

    scale = 16
    n = 1 << scale
    indptr, col = synthetic.rmat_csr(scale, device="cuda")
    A = device.matrix_from_device_csr(indptr, col, torch.ones(1, dtype=torch.int64, device="cuda"), n, n, "INT64", iso=True)
    ip, cj = indptr.cpu().numpy(), col.cpu().numpy()
    S = sp.csr_matrix((np.ones(cj.size, np.int64), cj, ip), shape=(n, n))
    ref = (S @ S).tocsr()
    ref.sort_indices()
    C = A.mxm(A, gb.semiring.plus_times).new()
    st = device.last_stats()
    assert st["out_nvals"] == ref.nnz and st["flops"] == int(np.diff(ip)[cj].sum())
    cp, cjd, cx, _ = device.matrix_device_views(C)
    assert np.array_equal(cp.cpu().numpy(), ref.indptr)
    assert np.array_equal(cjd.cpu().numpy(), ref.indices)
    assert np.array_equal(cx.cpu().numpy(), ref.data)
    # under the complemented structural mask (the reference pins C<~M.S> at tests/test_matrix.py:359-366 on its 7 x 7 literal)
    M = A.mxm(A, gb.semiring.plus_times).new(mask=~A.S)
    assert device.last_stats()["method"] == 7  # the complemented mask was fused into the product (no second pass over T)
    pat = sp.csr_matrix((np.ones(cj.size, np.int64), cj, ip), shape=(n, n))
    inside = ref.multiply(pat).tocsr()  # product restricted to A's pattern (values kept)
    diff = (ref - inside).tocsr()
    diff.eliminate_zeros()  # (the product's values are >= 1, so a zero here is exactly an entry inside the pattern)
    diff.sort_indices()
    mp, mj, mx, _ = device.matrix_device_views(M)
    assert np.array_equal(mp.cpu().numpy(), diff.indptr) and np.array_equal(mj.cpu().numpy(), diff.indices)
    assert np.array_equal(mx.cpu().numpy(), diff.data)


@pytest.mark.gpu
@pytest.mark.parametrize("scale,workload", [(18, "mxm_plus_times"), (20, "mxm_plus_times"), (22, "mxm_plus_times"), (22, "mxm_plus_times_masked"),
                                            (18, "mxm_plus_times_cmask"), (20, "mxm_plus_times_cmask")])
def test_mxm_bench_sizes_with_independent_checks(gb, scale, workload):
    """The SpGEMM lines bench.py prints (scale 20: the materialised product, 9.7 G entries; scale 22: row batches with the output
    streamed; scale 22 under the structural mask A; scale 18 / 20 under the complemented structural mask A, fused into the product),
    each with the checks of bench.verify_mxm: the multiply count against the row
    pointers, sum of the values = multiply count, nnz(C) = last row pointer, sampled rows (columns, order and values) against
    torch.unique over the gathered rows of B, and for the streamed pipeline a row block against its materialised product."""
    import argparse

    import torch

    import bench
    from graphblas_amd import device

    args = argparse.Namespace(streamed=False, stream_budget_gb=64.0, no_cpu_baseline=True)
    line = bench.run_mxm(args, gb, torch, device, 0, 1, None, torch.cuda.synchronize, scale=scale, workload=workload, steps=1, warmup=0,
                         want_cpu=False)
    assert line["verified"] is True
    cfg = line["config"]
    assert cfg["flops"] > 10 * cfg["nnz_A"] and cfg["nnz_C"] > 0
    if workload == "mxm_plus_times":
        assert cfg["nnz_C"] > 50 * cfg["nnz_A"]
        if scale >= 21:
            assert cfg["batches"] > 1 and cfg["checksum"] == cfg["flops"]
    elif workload == "mxm_plus_times_cmask":
        assert line["stats"]["method"] == 7 and cfg["nnz_C"] > 50 * cfg["nnz_A"]
    else:
        assert cfg["nnz_C"] <= cfg["nnz_A"]
    device.trim_memory()


@pytest.mark.gpu
def test_rccl_one_rank_world():
    """RCCL runs on the library's own device memory: a one-rank `nccl` world in a process of its own (tests/nccl_one_rank.py) drives the
    overlapped all-gather step (values, values + presence words), the probe exchange and the monoid all-reduce of the row-sharded vxm,
    each checked against the same product without the exchange; then bench.py's N-rank path with one rank (--force-dist): the masked
    relaxation through sharded.OverlappedMxv and the row-sharded SpGEMM, both `verified`."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "nccl_one_rank.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])  # (RCCL prints its banner to stdout as well)
    assert out["backend"] == "nccl" and out["ok"], out
    for args in (["--scale", "20", "--steps", "3", "--warmup", "3", "--no-extra", "--no-cpu-baseline"],
                 ["--workload", "mxm_plus_times", "--scale", "17", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--backend", "nccl"] + args, capture_output=True, text=True,
                           timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]  # (bench.py keeps stdout to its one JSON line)
        line = json.loads(r.stdout)
        assert line["verified"] is True, line
        if "exchange" in line:
            assert line["exchange"]["collective"].startswith("all_gather_into_tensor") and "rehearsal" in line
