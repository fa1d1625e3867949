"""Behaviour of the python-graphblas-shaped host API around the mxm / mxv / vxm path -- updater spellings, accumulators, descriptors,
recorder text, error classes, element access, the traversal loops, aggregators, `power`, indexed assign / extract, `dup` -- through the
C-ABI library (HIP kernels on the GPU tier, the same kernel sources under the CPU SIMT emulator on the CPU tier).  Each test cites the
reference code it follows (paths relative to /root/reference).

The reference's VALUE-LEVEL literals are not restated here: tests/test_library_golden.py runs them through the library from
tests/golden/reference_literals.json -- mxm / mxv / vxm (plain, masks, transposes, accum, the docs' tables, the primer's SSSP) and, since
the last session of round 5, the operations around the path (element access, index lists, resize, transpose, reductions, aggregators).
What stays here are BEHAVIOUR tests: spellings compared with each other, error classes, recorder text, results compared with numpy on the
same data, the equivalences the host API promises (power / masked transpose against explicit products)."""
import numpy as np
import pytest

from tests.backend import DEVICES, bind



def heq(a, b):
    """Host-side equality of two Vectors / Matrices: same class, shape, dtype, and identical to_coo() tuples.  The literal tests
    assert through this, not through the library's own device-side isequal (which has its own test below)."""
    if type(a) is not type(b) or a.dtype != b.dtype:
        return False
    if getattr(a, "shape", None) != getattr(b, "shape", None) or getattr(a, "size", None) != getattr(b, "size", None):
        return False
    ta, tb = a.to_coo(), b.to_coo()
    return len(ta) == len(tb) and all(x.dtype == y.dtype and x.tolist() == y.tolist() for x, y in zip(ta, tb))


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


@pytest.fixture
def A(gb):
    # graphblas/tests/test_matrix.py:34-49
    data = [
        [3, 0, 3, 5, 6, 0, 6, 1, 6, 2, 4, 1],
        [0, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5, 6],
        [3, 2, 3, 1, 5, 3, 7, 8, 3, 1, 7, 4],
    ]
    return gb.Matrix.from_coo(*data)


@pytest.fixture
def v(gb):
    # graphblas/tests/test_matrix.py:52-55
    return gb.Vector.from_coo([1, 3, 4, 6], [1, 1, 2, 0])


def test_vxm_accum(gb, A, v):
    # graphblas/tests/test_vector.py:350-368 -- five spellings of the accumulator
    Vector, semiring, binary, monoid = gb.Vector, gb.semiring, gb.binary, gb.monoid
    w1 = v.dup()
    w1(binary.plus) << v.vxm(A, semiring.plus_times)
    result = Vector.from_coo([0, 1, 2, 3, 4, 5, 6], [3, 1, 3, 1, 10, 14, 4], size=7)
    assert heq(w1, result)
    w2 = v.dup()
    w2(monoid.plus) << v.vxm(A, semiring.plus_times)
    assert heq(w2, result)
    w3 = v.dup()
    w3(accum=monoid.plus) << v.vxm(A, semiring.plus_times)
    assert heq(w3, result)
    w4 = v.dup()
    w4("+") << v.vxm(A, semiring.plus_times)
    assert heq(w4, result)
    w5 = v.dup()
    w5(accum="plus") << v.vxm(A, semiring.plus_times)
    assert heq(w5, result)


def test_semiring_handles(gb):
    # graphblas/tests/test_op.py:96-100: semiring.min_plus["INT32"].gb_obj == lib.GrB_MIN_PLUS_SEMIRING_INT32
    from graphblas_amd import _lib

    assert gb.semiring.min_plus["INT32"].gb_obj == _lib.handle("GrB_MIN_PLUS_SEMIRING_INT32")
    assert gb.semiring.plus_times["FP64"].gb_name == "GrB_PLUS_TIMES_SEMIRING_FP64"
    assert gb.semiring.lor_land["BOOL"].gb_name == "GrB_LOR_LAND_SEMIRING_BOOL"
    assert gb.semiring.lor_land["INT64"].gb_name == "GrB_LOR_LAND_SEMIRING_BOOL"  # coerced (semiring.py:538-548)
    assert gb.semiring.any_pair["FP32"].gb_name == "GxB_ANY_PAIR_FP32"
    assert gb.dtypes.unify(gb.dtypes.INT8, gb.dtypes.UINT16) is gb.dtypes.INT32
    assert gb.dtypes.unify(gb.dtypes.FP32, gb.dtypes.INT32) is gb.dtypes.FP64


def test_recorder_call_text(gb, A, v):
    # graphblas/tests/test_recorder.py:31-37: argument order (C, mask, accum, semiring, A, B, desc)
    calls = []
    gb.record_calls(calls)
    try:
        w = gb.Vector(int, 7, name="w")
        w << A.mxv(v, gb.semiring.min_plus)
        w2 = gb.Vector(int, 7, name="w2")
        w2 << v.vxm(A.T, gb.semiring.min_plus)
    finally:
        gb.record_calls(None)
    mx = [c for c in calls if c.startswith(("GrB_mxv", "GrB_vxm"))]
    assert mx[0] == f"GrB_mxv(w, NULL, NULL, GrB_MIN_PLUS_SEMIRING_INT64, {A.name}, {v.name}, NULL);"
    assert mx[1] == f"GrB_vxm(w2, NULL, NULL, GrB_MIN_PLUS_SEMIRING_INT64, {v.name}, {A.name}, GrB_DESC_T1);"


def test_errors(gb, A, v):
    # shape errors come from the library (graphblas/tests/test_matrix.py:1920-1927, test_vector.py:1131-1136)
    with pytest.raises(gb.exceptions.DimensionMismatch):
        A.mxv(gb.Vector.from_coo([0], [1], size=3), gb.semiring.plus_times)
    with pytest.raises(gb.exceptions.DimensionMismatch):
        gb.Vector(int, 3) << A.mxv(v)
    with pytest.raises(TypeError, match="Mask must be"):
        A.mxv(v).new(mask=v)  # non-bool collection used bare as a mask (test_matrix.py:373-374)
    with pytest.raises(TypeError, match="replace"):
        v(replace=True)
    # duplicates need a dup_op (graphblas/tests/test_matrix.py:99-120)
    with pytest.raises(ValueError, match="Duplicate indices found"):
        gb.Matrix.from_coo([0, 0], [1, 1], [1, 2])
    M = gb.Matrix.from_coo([0, 0], [1, 1], [1, 2], dup_op=gb.binary.plus)
    assert M.to_coo()[2].tolist() == [3]
    with pytest.raises(gb.exceptions.OutputNotEmpty):
        M.build([0], [0], [1])
    with pytest.raises(gb.exceptions.IndexOutOfBound):
        gb.Matrix.from_coo([5], [0], [1], nrows=2, ncols=2)


def test_roundtrips(gb):
    rng = np.random.default_rng(3)
    r, c = np.nonzero(rng.random((40, 33)) < 0.2)
    x = rng.integers(-50, 50, r.size)
    x[::7] = 0  # explicit zeros survive
    M = gb.Matrix.from_coo(r, c, x, nrows=40, ncols=33)
    I, J, X = M.to_coo()
    order = np.lexsort((c, r))
    assert I.tolist() == r[order].tolist() and J.tolist() == c[order].tolist() and X.tolist() == x[order].tolist()
    Ap, Aj, Ax = M.to_csr()
    M2 = gb.Matrix.from_csr(Ap, Aj, Ax, ncols=33)
    assert heq(M2, M)
    Cp, Ci, Cx = M.to_csc()
    M3 = gb.Matrix.from_csc(Cp, Ci, Cx, nrows=40)
    assert heq(M3, M)
    assert heq(M.T.new().T.new(), M)
    d = rng.random(50)
    vd = gb.Vector.from_dense(d)
    assert vd.nvals == 50 and np.array_equal(vd.to_dense(), d)
    assert gb.Vector(float, 5).nvals == 0


def test_resize_random(gb):
    """resize against a numpy restatement (drop what lies beyond the new bounds), then use the result in mxv: caches of the
    old shape must not survive."""
    rng = np.random.default_rng(8)
    for trial in range(6):
        m, n = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        r, c = np.nonzero(rng.random((m, n)) < 0.25)
        x = rng.integers(1, 9, r.size).astype(np.int64) if trial % 2 else np.ones(r.size, np.int64)  # (iso when all ones)
        M = gb.Matrix.from_coo(r, c, x, nrows=m, ncols=n)
        u = gb.Vector.from_coo(np.arange(n), np.arange(n) + 1, size=n)
        M.mxv(u, gb.semiring.plus_times).new()  # builds whatever the library caches for this shape
        m2, n2 = int(rng.integers(0, 120)), int(rng.integers(0, 120))
        M.resize(m2, n2)
        keep = (r < m2) & (c < n2)
        I, J, X = M.to_coo()
        order = np.lexsort((c[keep], r[keep]))
        assert (M.nrows, M.ncols, M.nvals) == (m2, n2, int(keep.sum()))
        assert I.tolist() == r[keep][order].tolist() and J.tolist() == c[keep][order].tolist() and X.tolist() == x[keep][order].tolist()
        u.resize(n2)
        assert u.nvals == min(n, n2)
        w = M.mxv(u, gb.semiring.plus_times).new()
        dense = np.zeros((m2, n2), np.int64)
        dense[r[keep], c[keep]] = x[keep]
        uu = np.zeros(n2, np.int64)
        uu[: min(n, n2)] = np.arange(min(n, n2)) + 1
        pres = np.zeros((m2, n2), bool)
        pres[r[keep], c[keep]] = True
        pres[:, min(n, n2):] = False
        wi, wv = w.to_coo()
        assert wi.tolist() == np.nonzero(pres.any(axis=1))[0].tolist()
        assert wv.tolist() == (dense @ uu)[pres.any(axis=1)].tolist()
    big = gb.Vector.from_coo([0, 5, 70, 128, 129], [1, 2, 3, 4, 5], size=130)
    big.resize(129)
    assert big.to_coo()[0].tolist() == [0, 5, 70, 128]
    big.resize(64)
    assert big.to_coo()[0].tolist() == [0, 5]
    big.resize(0)
    assert big.nvals == 0 and big.size == 0


# ---- the vector operations around the path -----------------------------------------------------
def test_element_access_behaviour(gb, v):
    # graphblas/tests/test_vector.py:267-297 -- the error classes and the Scalar on the left-hand side (the element VALUES of these tests:
    # tests/golden/reference_literals.json extract_element_v / set_element_v / remove_element_v)
    with pytest.raises(IndexError):
        v[100]
    with pytest.raises(TypeError, match="Invalid type for index"):
        v[object()]
    s = gb.Scalar(int)
    s << v[1]
    assert s == v[1].new()
    assert v[0].new().value is None
    nv = v.nvals
    del v[0]  # removing an absent element changes nothing
    assert v.nvals == nv


def test_assign_scalar_all_and_mask(gb, v):
    # graphblas/tests/test_vector.py:528-533, 544-562, 616-628
    w = gb.Vector.from_coo([0, 1, 2], [1, 1, 1])
    w[:] = gb.Scalar.from_value(9)
    assert heq(w, gb.Vector.from_coo([0, 1, 2], [9, 9, 9]))
    with pytest.raises(TypeError, match="Bad type for arg"):
        w[:] = object()
    w << 2
    assert heq(w, gb.Vector.from_coo([0, 1, 2], [2, 2, 2]))
    mask = gb.Vector.from_coo([1, 2, 5, 6], [0, 0, 1, 0])
    result = gb.Vector.from_coo([1, 3, 4, 5, 6], [1, 1, 2, 5, 0])
    for form in range(3):
        w = v.dup()
        if form == 0:
            w[:](mask.V) << 5
        elif form == 1:
            w(mask.V) << 5
        else:
            w(mask.V)[:] << 5
        assert heq(w, result)
    result2 = gb.Vector.from_coo([0, 1, 2, 3, 4, 6], [5, 5, 5, 5, 5, 5])
    w = v.dup()
    w[:](~mask.V) << 5
    assert heq(w, result2)
    w = v.dup()
    w(~mask.V) << 5
    assert heq(w, result2)
    x = gb.Vector.from_coo([0, 1, 2], [1, 2, 3])
    m = gb.Vector.from_coo([0, 2], [False, True])
    x(m.V)[:] << 100
    assert heq(x, gb.Vector.from_coo([0, 1, 2], [1, 2, 100]))
    x(m.V, accum=gb.binary.plus)[:] << 1000
    assert heq(x, gb.Vector.from_coo([0, 1, 2], [1, 2, 1100]))


def test_reduce(gb, v):
    # graphblas/tests/test_vector.py:866-882
    s = v.reduce(gb.monoid.plus).new()
    assert s == 4
    assert s.dtype.name == "INT64"
    assert v.reduce(gb.binary.plus).new() == 4
    with pytest.raises(TypeError, match="Expected type: Monoid"):
        v.reduce(gb.binary.minus)
    s(accum=gb.binary.times) << v.reduce(gb.monoid.plus)
    assert s == 16
    assert v.reduce().new() == 4
    e = gb.Vector(int, 5)
    assert e.reduce(gb.monoid.plus).new().value is None
    assert e.reduce(gb.monoid.plus, allow_empty=False).new() == 0
    assert e.reduce(gb.monoid.min, allow_empty=False).new() == np.iinfo(np.int64).max
    assert v.reduce(gb.monoid.max).new() == 2 and v.reduce(gb.monoid.min).new() == 0


def test_notebook_level_bfs(gb):
    """notebooks/Example B.1 -- Level BFS.ipynb, cell by cell, entirely through the API (no host-side bookkeeping)."""
    edges = [[3, 0, 3, 5, 6, 0, 6, 1, 6, 2, 4, 1], [0, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5, 6]]
    A = gb.Matrix.from_coo(edges[0], edges[1], [True for _ in edges[0]])
    s = 1
    n = A.nrows
    v = gb.Vector("INT32", n)
    q = gb.Vector(bool, n)
    q[s] << True
    succ = gb.Scalar(bool)
    d = 0
    while True:
        d += 1
        v[:](mask=q.V) << d
        q(~v.S, replace=True) << q.vxm(A, gb.semiring.lor_land)
        succ << q.reduce(gb.monoid.lor, allow_empty=False)
        if not succ:
            break
    # levels from vertex 1 in this digraph: 1 -> {4, 6} -> {5, 2, 3} -> {0}
    idx, lev = v.to_coo()
    assert idx.tolist() == [0, 1, 2, 3, 4, 5, 6] and lev.tolist() == [4, 1, 3, 3, 2, 3, 2]


def test_primer_sssp_loop_on_device(gb):
    # docs/getting_started/primer.rst:221-251, with the loop's isequal through the API
    G = gb.Matrix.from_coo([0, 0, 1, 1, 2], [1, 2, 2, 3, 3], [2.0, 5.0, 1.5, 4.25, 0.5], nrows=4, ncols=4)
    v = gb.Vector.from_coo([0], [0.0], size=4)
    while True:
        w = v.dup()
        v(gb.op.min) << gb.semiring.min_plus(v @ G)
        if v.isequal(w):
            break
    assert v.to_coo()[1].tolist() == [0.0, 2.0, 3.5, 4.0]


def test_ewise_mult_and_add(gb, v):
    # graphblas/tests/test_vector.py:371-380, 402-418 -- operator spellings and the default operator (the literal results:
    # reference_literals.json ewise_mult_times / ewise_add_max)
    v2 = gb.Vector.from_coo([0, 3, 5, 6], [2, 3, 2, 1])
    w = v.ewise_mult(v2, gb.binary.times).new()
    w2 = gb.Vector(w.dtype, w.size)
    w2 << v.ewise_mult(v2, gb.monoid.times)
    assert heq(w, w2)
    assert [x.tolist() for x in w.to_coo()] == [[3, 6], [3, 0]]  # (by hand: the indices both hold -- 1 * 3 and 0 * 1, the explicit zero kept)
    with pytest.raises(TypeError, match="Expected type: BinaryOp, Monoid"):
        v.ewise_mult(v2, gb.semiring.plus_times)
    w = v.ewise_add(v2, gb.binary.max).new()
    w2.update(v.ewise_add(v2, gb.monoid.max))
    assert heq(w, w2)
    assert [x.tolist() for x in w.to_coo()] == [[0, 1, 3, 4, 5, 6], [2, 1, 3, 2, 2, 1]]  # (by hand: the union, max where both hold an entry)
    with pytest.raises(TypeError, match="Expected type: BinaryOp, Monoid"):
        v.ewise_add(v2, gb.semiring.max_times)
    assert heq(v.ewise_add(v2).new(), v.ewise_add(v2, gb.monoid.plus).new())  # default is plus


def test_comparisons_and_isequal_on_device(gb, v):
    """binary.eq & co. (T x T -> BOOL) through eWiseMult / eWiseAdd, and Vector.isequal the way the reference computes it
    (graphblas/core/vector.py:340-379: nvals, ewise_mult(eq), reduce(land))."""
    v2 = gb.Vector.from_coo([1, 3, 4, 6], [1, 2, 2, 0])
    m = v.ewise_mult(v2, gb.binary.eq).new()
    assert m.dtype.name == "BOOL"
    assert [x.tolist() for x in m.to_coo()] == [[1, 3, 4, 6], [True, False, True, True]]
    assert v.ewise_mult(v2, gb.binary.gt).new().to_coo()[1].tolist() == [False, False, False, False]
    assert v.ewise_mult(v2, gb.binary.le).new().to_coo()[1].tolist() == [True, True, True, True]
    assert not v.isequal(v2)
    assert v.isequal(v.dup())
    assert not v.isequal(gb.Vector.from_coo([1, 3, 4, 5], [1, 1, 2, 0], size=7))  # same nvals, other pattern
    assert not v.isequal(gb.Vector.from_coo([1, 3, 4, 6], [1, 1, 2, 0], size=8))  # other size
    assert gb.Vector(int, 3).isequal(gb.Vector(int, 3))
    f = gb.Vector.from_coo([0, 2], [1.5, float("inf")])
    assert f.isequal(gb.Vector.from_coo([0, 2], [1.5, float("inf")]))
    with pytest.raises(Exception):
        v(accum=gb.binary.eq) << v.ewise_mult(v2, gb.binary.plus)  # a comparison is not an accumulator


def test_operator_strings_match_objects(gb):
    """String spellings resolve like the reference's (core/operator/utils.py:347-370): "==" / "eq" name binary.eq for a
    BinaryOp (T x T -> BOOL), the BOOL-only lxnor only for a monoid; the symbol and the object give the same result."""
    u = gb.Vector.from_coo([0, 1], [2, 3])
    w = gb.Vector.from_coo([0, 1], [3, 3])
    want = u.ewise_mult(w, gb.binary.eq).new()
    assert [x.tolist() for x in want.to_coo()] == [[0, 1], [False, True]]
    for spelling in ("eq", "=="):
        assert heq(u.ewise_mult(w, spelling).new(), want)
    for sym, name in (("+", "plus"), ("*", "times"), ("-", "minus"), ("<", "lt"), (">=", "ge"), ("!=", "ne")):
        assert heq(u.ewise_mult(w, sym).new(), u.ewise_mult(w, getattr(gb.binary, name)).new())
    from graphblas_amd import operators

    assert operators._from_string("==", "monoid") is gb.monoid.lxnor
    assert operators._from_string("min.+", "semiring") is gb.semiring.min_plus


def test_reduce_rowwise_columnwise(gb, A):
    # graphblas/tests/test_matrix.py:1355-1360, 1648-1653 -- operator spellings and the transposed forms against each other (the literal
    # vectors: reference_literals.json reduce_rowwise_plus / reduce_columnwise_plus)
    I0, J0, X0 = (np.asarray(a) for a in A.to_coo())
    result = A.reduce_rowwise(gb.monoid.plus).new()
    assert [x.tolist() for x in result.to_coo()] == [list(range(7)), np.bincount(I0.astype(int), weights=X0).astype(int).tolist()]  # (numpy on the same tuples)
    assert result.to_coo()[1].tolist() == [5, 12, 1, 6, 7, 1, 15]  # (by hand from the fixture)
    assert heq(A.reduce_rowwise(gb.binary.plus).new(), result)
    assert heq(A.T.reduce_columnwise(gb.monoid.plus).new(), result)
    result = A.reduce_columnwise(gb.monoid.plus).new()
    assert [x.tolist() for x in result.to_coo()] == [list(range(7)), np.bincount(J0.astype(int), weights=X0).astype(int).tolist()]
    assert heq(A.T.reduce_rowwise(gb.binary.plus).new(), result)
    with pytest.raises(TypeError, match="Expected type: Monoid"):
        A.reduce_rowwise(gb.binary.minus)
    # max / min and a masked, accumulated form
    I, J, X = A.to_coo()
    dense = np.full((7, 7), -1)
    dense[I.astype(int), J.astype(int)] = X
    assert A.reduce_rowwise(gb.monoid.max).new().to_coo()[1].tolist() == dense.max(axis=1).tolist()
    w = gb.Vector.from_coo([0, 1, 2], [100, 100, 100], size=7)
    m = gb.Vector.from_coo([0, 2, 4], [True, True, True], size=7)
    w(m.S, accum=gb.binary.plus) << A.reduce_rowwise(gb.monoid.plus)
    assert [x.tolist() for x in w.to_coo()] == [[0, 1, 2, 4], [105, 100, 101, 7]]


def test_reduce_agg(gb, A):
    # graphblas/tests/test_matrix.py:1364-1417 (the aggregators that are a monoid or one semiring mat-vec).  The literal vectors of agg.sum
    # and agg.exists: reference_literals.json agg_sum_* / agg_exists_*; the reference derives its counts (A.dup(bool) reduced with
    # plus[int]) -- numpy's bincount here
    rows_, cols_, _ = A.to_coo()
    assert heq(A.T.reduce_columnwise(gb.agg.sum).new(), A.reduce_rowwise(gb.agg.sum).new())
    counts = gb.Vector.from_coo(np.arange(7), np.bincount(rows_, minlength=7))
    w3 = A.reduce_rowwise(gb.agg.count).new()
    assert w3.dtype == gb.dtypes.INT64
    assert heq(w3, counts)
    assert heq(A.T.reduce_columnwise(gb.agg.count).new(), counts)
    assert heq(A.T.reduce_rowwise(gb.agg.sum).new(), A.reduce_columnwise(gb.agg.sum).new())
    counts = gb.Vector.from_coo(np.arange(7), np.bincount(cols_, minlength=7))
    assert heq(A.reduce_columnwise(gb.agg.count).new(), counts)
    assert heq(A.T.reduce_rowwise(gb.agg.count).new(), counts)
    # rows without entries get no count (the product has no entry there)
    B = gb.Matrix.from_coo([0, 0, 3], [1, 2, 0], [1.5, 2.5, 3.5], nrows=5, ncols=4)
    assert heq(B.reduce_rowwise(gb.agg.count).new(), gb.Vector.from_coo([0, 3], [2, 1], size=5))
    assert heq(B.reduce_columnwise(gb.agg.min).new(), gb.Vector.from_coo([0, 1, 2], [3.5, 1.5, 2.5], size=4))


def test_reduce_scalar(gb, A):
    # graphblas/tests/test_matrix.py:1419-1429 -- the spellings against each other and numpy (the literals 47 / 1270080 / 12 / 1:
    # reference_literals.json reduce_scalar_*), :1448-1451 (empty)
    vals_ = A.to_coo()[2]
    assert A.reduce_scalar(gb.agg.sum).new() == A.reduce_scalar(gb.monoid.plus).new() == A.reduce_scalar().new() == int(vals_.sum())
    assert A.T.reduce_scalar(gb.agg.prod).new() == A.reduce_scalar(gb.monoid.times).new()
    assert A.reduce_scalar(gb.agg.count).new() == A.nvals
    assert A.reduce_scalar(gb.agg.min).new() == int(vals_.min())
    assert A.reduce_scalar(gb.agg.max).new() == int(vals_.max())
    B = gb.Matrix(int, 3, 4)
    assert B.reduce_scalar(gb.agg.sum, allow_empty=True).new().is_empty
    assert B.reduce_scalar(gb.agg.sum, allow_empty=False).new() == 0


def test_vector_reduce_agg(gb, v):
    # graphblas/tests/test_vector.py:919-935, 1047-1052 (the same subset; literals: reference_literals.json reduce_plus / vector_reduce_*)
    s = gb.Scalar(int)
    s << v.reduce(gb.agg.sum)
    assert s == v.reduce(gb.monoid.plus).new() and s.value == 4  # (1 + 1 + 2 + 0)
    s << v.reduce(gb.agg.count)
    assert s == v.nvals and s.value == 4
    s << v.reduce(gb.agg.max)
    assert s == v.reduce(gb.monoid.max).new() and s.value == 2
    empty = gb.Vector(int, size=3)
    s << empty.reduce(gb.agg.count)
    assert s.is_empty


def test_power(gb, A):
    # graphblas/tests/test_matrix.py:4379-4415
    expected = A.dup()
    for i in range(1, 50):
        result = A.power(i).new()
        assert result.isequal(expected), i
        expected << A @ expected
    expected = A.T.new()
    for i in range(1, 10):
        result = A.T.power(i).new()
        assert result.isequal(expected), i
        expected << A.T @ expected
    expected = A.dup()
    for i in range(1, 10):
        result = A.power(i, gb.semiring.min_plus).new()
        assert result.isequal(expected), i
        expected << gb.semiring.min_plus(A @ expected)
    result = A.power(0).new()
    idx = list(range(7))
    assert heq(result, gb.Matrix.from_coo(idx, idx, [1] * 7))
    result = A.power(0, gb.semiring.plus_min).new()
    identity = gb.monoid.min.identity(A.dtype)
    assert identity != 1
    assert heq(result, gb.Matrix.from_coo(idx, idx, [identity] * 7, dtype=A.dtype))
    with pytest.raises(TypeError, match="must be a nonnegative integer"):
        A.power(1.5)
    with pytest.raises(ValueError, match="must be a nonnegative integer"):
        A.power(-1)
    with pytest.raises(ValueError, match="binaryop must be associated with a monoid"):
        A.power(0, gb.semiring.min_first)
    B = gb.Matrix.from_coo([0, 1], [2, 0], [1, 2], nrows=2, ncols=3)
    with pytest.raises(gb.exceptions.DimensionMismatch):
        B.power(2)


def test_power_masked_final_product(gb, A):
    # the updater's (mask, accum) apply to the last multiplication (reference core/matrix.py:99-155)
    for n in (2, 3, 4, 5, 6, 7, 12):
        C = A.dup()
        C(A.S, gb.binary.plus) << A.power(n)
        D = A.dup()
        D(A.S, gb.binary.plus) << A.power(n - 1).new() @ A
        assert C.isequal(D), n
        C = A.dup()
        C(~A.S, replace=True) << A.power(n, gb.semiring.min_plus)
        D = A.dup()
        D(~A.S, replace=True) << gb.semiring.min_plus(A.power(n - 1, gb.semiring.min_plus).new() @ A)
        assert C.isequal(D), n
    # the degenerate powers take the write rule too (a copy through GrB_transpose): equal to the product with the identity matrix
    eye = gb.Matrix.from_coo(range(7), range(7), [1] * 7, dtype=A.dtype)
    M = gb.Matrix.from_coo([0, 1, 3, 6, 6], [1, 4, 0, 2, 3], [True, False, True, True, True], nrows=7, ncols=7)
    for n in (0, 1):
        want = eye if n == 0 else A
        for upd in (lambda X: X(M.S), lambda X: X(M.V, gb.binary.plus), lambda X: X(~M.S, replace=True), lambda X: X(accum=gb.binary.min)):
            C = A.dup()
            upd(C) << A.power(n)
            D = A.dup()
            upd(D) << eye.mxm(want, gb.semiring.plus_times)
            assert heq(C, D), n


def test_masked_transpose(gb, A):
    """``C(mask, accum, replace) << A.T`` -> GrB_transpose with the write rule (reference core/base.py:401-411): equal to the
    product of the identity matrix with the transposed operand under the same mask / accumulator (mxm's own write rule)."""
    eye = gb.Matrix.from_coo(range(7), range(7), [1] * 7, dtype=A.dtype)
    M = gb.Matrix.from_coo([0, 1, 3, 3, 6, 6, 5], [1, 4, 0, 2, 2, 3, 2], [True, False, True, True, True, False, True], nrows=7, ncols=7)
    # (the plain transpose's literal, graphblas/tests/test_matrix.py:1700-1707: reference_literals.json transpose)
    for upd in (lambda X: X(M.S), lambda X: X(M.V), lambda X: X(~M.V, replace=True), lambda X: X(M.S, gb.binary.plus),
                lambda X: X(accum=gb.binary.times), lambda X: X(~M.S, gb.binary.min, replace=True)):
        C = A.dup()
        upd(C) << A.T
        D = A.dup()
        upd(D) << eye.mxm(A.T, gb.semiring.plus_times)
        assert heq(C, D)
    C = A.dup()
    C(C.S) << C.T  # output aliased with the input and the mask
    D = A.dup()
    D(A.S) << eye.mxm(A.T, gb.semiring.plus_times)
    assert heq(C, D)


def test_index_max(gb):
    # graphblas/tests/test_core.py:95-96; objects of that size exist as long as they hold no entries (GrB_INDEX_MAX)
    assert gb.MAX_SIZE == 2**60
    v = gb.Vector(int, gb.MAX_SIZE)
    assert v.size == 2**60 and v.nvals == 0
    A = gb.Matrix(float, gb.MAX_SIZE, gb.MAX_SIZE)
    assert A.shape == (2**60, 2**60) and A.nvals == 0
    with pytest.raises(gb.exceptions.InvalidValue):
        gb.Vector(int, gb.MAX_SIZE + 1)
    with pytest.raises(gb.exceptions.InvalidValue):
        gb.Matrix(int, 1, gb.MAX_SIZE + 1)
    B = gb.Matrix(float, 3, gb.MAX_SIZE)
    u = gb.Vector(float, gb.MAX_SIZE)
    w = B.mxv(u, gb.semiring.plus_times).new()  # a product of empty operands of the largest inner dimension
    assert w.size == 3 and w.nvals == 0
    # ... also under a mask with replace and an accumulator: only the write rule acts
    w = gb.Vector.from_coo([0, 2], [1.5, 2.5], size=3)
    mk = gb.Vector.from_coo([0, 1], [True, True], size=3)
    w(mk.S, gb.binary.plus, replace=True) << B.mxv(u, gb.semiring.plus_times)
    assert heq(w, gb.Vector.from_coo([0], [1.5], size=3))
    x = gb.Vector(float, 3)
    y = x.vxm(B, gb.semiring.min_plus).new()
    assert y.size == gb.MAX_SIZE and y.nvals == 0
    C = gb.Matrix(float, gb.MAX_SIZE, 4)
    P = B.mxm(C, gb.semiring.plus_times).new()
    assert P.shape == (3, 4) and P.nvals == 0
    B.resize(3, 5)
    assert B.shape == (3, 5)


def test_extract_and_assign_with_index_lists(gb, v):
    # graphblas/tests/test_vector.py:428-443 (extract), :505-517 (assign), :520-541 (assign a scalar) -- a slice, a list and an array name
    # the same indices (the literal results: reference_literals.json extract_index_list / assign_vector_index_list / assign_scalar_index_list)
    result = gb.Vector(v.dtype, 3)
    result << v[[1, 3, 5]]
    w = gb.Vector(v.dtype, 3)
    w() << v[1::2]
    assert heq(w, result)
    assert [x.tolist() for x in result.to_coo()] == [[0, 1], [1, 1]]  # (by hand: v[1], v[3]; v[5] is absent)
    assert heq(v[1::2].new(), w)
    u = gb.Vector.from_coo([0, 2], [9, 8])
    result = v.dup()
    result[[0, 2, 4]] = u
    w = v.dup()
    w[:5:2] << u
    assert heq(w, result)
    assert [x.tolist() for x in result.to_coo()] == [[0, 1, 3, 4, 6], [9, 1, 1, 8, 0]]  # (by hand: w[0] = u[0], w[2] takes u's absent entry, w[4] = u[2])
    with pytest.raises(TypeError, match="Invalid type for index"):
        w[w] = 1
    result = v.dup()
    result[[1, 3, 5]] = 9
    w = v.dup()
    w[1::2] = 9
    assert heq(w, result)
    assert [x.tolist() for x in result.to_coo()] == [[1, 3, 4, 5, 6], [9, 9, 2, 9, 0]]
    with pytest.raises(IndexError):
        v[[-v.size - 1]]
    # accumulate into a sub-vector, and a mask of the output's size with replace (C API 2.0 GrB_assign)
    w = v.dup()
    w(gb.binary.plus)[[1, 2, 4]] << gb.Vector.from_coo([0, 1], [10, 20], size=3)
    assert heq(w, gb.Vector.from_coo([1, 2, 3, 4, 6], [11, 20, 1, 2, 0]))
    m = gb.Vector.from_coo([1, 2, 3], [True, True, False], size=7)
    w = v.dup()
    w(m.V, replace=True)[[1, 2, 4]] << gb.Vector.from_coo([0, 1], [10, 20], size=3)
    assert heq(w, gb.Vector.from_coo([1, 2], [10, 20], size=7))


def test_matrix_isequal_isclose_on_device(gb, A):
    """Matrix.isequal / isclose compare pattern and values in the library (GrX_Matrix_isclose), across types and iso storage."""
    assert A.isequal(A.dup())
    rows, cols, vals = A.to_coo()
    B = gb.Matrix.from_coo(rows, cols, vals.astype(np.float64), nrows=7, ncols=7)
    assert A.isequal(B) and not A.isequal(B, check_dtype=True)
    v2 = vals.copy()
    v2[3] += 1
    assert not A.isequal(gb.Matrix.from_coo(rows, cols, v2, nrows=7, ncols=7))
    c2 = cols.copy()
    c2[0] = (c2[0] + 1) % 7 if (rows[0], (c2[0] + 1) % 7) not in set(zip(rows.tolist(), cols.tolist())) else c2[0]
    assert np.array_equal(c2, cols) or not A.isequal(gb.Matrix.from_coo(rows, c2, vals, nrows=7, ncols=7))
    assert not A.isequal(gb.Matrix.from_coo(rows, cols, vals, nrows=8, ncols=7))
    near = gb.Matrix.from_coo(rows, cols, vals.astype(np.float64) * (1 + 1e-9), nrows=7, ncols=7)
    assert B.isclose(near) and not B.isequal(near) and not B.isclose(near, rel_tol=1e-12)
    ones = gb.Matrix.from_coo(rows, cols, np.ones(rows.size, np.int64), nrows=7, ncols=7)  # stored as one value (iso)
    assert ones.isequal(gb.Matrix.from_coo(rows, cols, np.ones(rows.size), nrows=7, ncols=7))
    assert gb.Matrix(int, 3, 4).isequal(gb.Matrix(float, 3, 4))


def test_dup_dtype_mask_clear(gb, A, v):
    """dup with its extended arguments -- literals of graphblas/tests/test_matrix.py:66-96 and tests/test_vector.py:69-96; the
    typecast copy is made in the library (GrX_Matrix_dup_as / GrX_Vector_dup_as), the masked one is `rv(mask) << self`."""
    C = A.dup()
    assert C is not A and C.dtype == A.dtype and C.nvals == A.nvals and C.nrows == A.nrows and C.ncols == A.ncols
    keep = A.dup()
    A.clear()  # (not the same backend object; Matrix element assignment is outside the path)
    assert C.nvals == 12 and heq(C, keep)
    A = keep
    D = gb.Matrix.from_coo([0, 1], [0, 1], [0, 2.5], dtype="FP64")
    assert heq(D.dup(dtype="INT64"), gb.Matrix.from_coo([0, 1], [0, 1], [0, 2], dtype="INT64"))
    assert heq(D.dup(mask=D.V), gb.Matrix.from_coo([1], [1], [2.5], dtype="FP64"))
    assert heq(D.dup(dtype="INT64", mask=D.V), gb.Matrix.from_coo([1], [1], [2], dtype="INT64"))
    E = A.dup(clear=True)
    assert E.dtype == A.dtype and E.nvals == 0 and E.shape == A.shape
    E = A.dup("INT8", clear=True)
    assert E.dtype.name == "INT8" and E.nvals == 0 and E.shape == A.shape
    # vectors
    u = v.dup()
    assert u is not v and u.dtype == v.dtype and u.nvals == v.nvals and u.size == v.size
    v[0] = 1000
    assert u[0].new() != 1000
    w = gb.Vector.from_coo([0, 1], [0, 2.5], dtype="FP64")
    assert heq(w.dup(dtype="INT64"), gb.Vector.from_coo([0, 1], [0, 2], dtype="INT64"))
    assert heq(w.dup(mask=w.V), gb.Vector.from_coo([1], [2.5], dtype="FP64"))
    assert heq(w.dup(dtype="INT64", mask=w.V), gb.Vector.from_coo([1], [2], dtype="INT64"))
    x = v.dup(clear=True)
    assert x.dtype == v.dtype and x.nvals == 0 and x.size == v.size
    x = v.dup("INT8", clear=True)
    assert x.dtype.name == "INT8" and x.nvals == 0 and x.size == v.size
    # a plain object on the right-hand side of an update: a copy under the write rule, with an accumulator too
    y = gb.Vector.from_coo([0, 1, 3], [5, 6, 7], size=7)
    y(gb.binary.plus) << v
    assert heq(y, gb.Vector.from_coo([0, 1, 3, 4, 6], [1005, 7, 8, 2, 0], size=7))
    F = gb.Matrix.from_coo([0, 3], [0, 0], [1, 10], nrows=7, ncols=7)
    F(gb.binary.plus) << A
    fr, fc, fv = F.to_coo()
    ar, ac, av = A.to_coo()
    exp = {(int(r), int(c)): int(x) for r, c, x in zip(ar, ac, av)}
    exp[(0, 0)] = exp.get((0, 0), 0) + 1
    exp[(3, 0)] = exp.get((3, 0), 0) + 10
    assert {(int(r), int(c)): int(x) for r, c, x in zip(fr, fc, fv)} == exp
