"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/grb_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import
this module.  The product (python-graphblas_amd/) never does.

The oracle restates the GraphBLAS C-API semantics of the three calls the reference
dispatches on the hot path (graphblas/core/matrix.py:2254 ``GrB_mxv``, :2321
``GrB_mxm``; graphblas/core/vector.py:1370 ``GrB_vxm``; argument marshalling
graphblas/core/base.py:496-503).  The arithmetic itself is in SuiteSparse:GraphBLAS
(third-party, absent here); parity is pinned on the reference's own test literals
(tests/golden/reference_literals.json).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TYPE_CODES = {
    "BOOL": 0, "INT8": 1, "INT16": 2, "INT32": 3, "INT64": 4,
    "UINT8": 5, "UINT16": 6, "UINT32": 7, "UINT64": 8, "FP32": 9, "FP64": 10,
}
NP_OF = {
    "BOOL": np.bool_, "INT8": np.int8, "INT16": np.int16, "INT32": np.int32, "INT64": np.int64,
    "UINT8": np.uint8, "UINT16": np.uint16, "UINT32": np.uint32, "UINT64": np.uint64,
    "FP32": np.float32, "FP64": np.float64,
}
NAME_OF_NP = {np.dtype(v): k for k, v in NP_OF.items()}
OP_CODES = {
    "first": 0, "second": 1, "pair": 2, "plus": 3, "minus": 4, "times": 5,
    "min": 6, "max": 7, "lor": 8, "land": 9, "lxor": 10, "lxnor": 11, "any": 12,
    "oneb": 2, "eq_bool": 11,
}


def build():
    """Compile oracle/libgrb_oracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "libgrb_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("grb_oracle.c", "grb_oracle_typed.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libgrb_oracle.so"], stdout=subprocess.DEVNULL)
    return so


_MAX_THREADS = None


def cpu_budget():
    """(logical CPUs this process may run on, CPUs' worth of time its cgroup grants it or None).  A container on a 256-thread host with
    ``cpu.max = 1600000 100000`` gets 16 CPUs of time however many threads it starts: a team of 128 is throttled, not faster (measured on
    the round-6 GPU box: 128 / 64 / 32 / 16 threads = 200.7 / 191.9 / 120.7 / 140.1 ms per headline pass)."""
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        logical = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return logical, quota


def lib():
    global _LIB, _MAX_THREADS
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.grbo_num_threads.restype = ctypes.c_int
        _MAX_THREADS = int(_LIB.grbo_num_threads())
        if "OMP_NUM_THREADS" not in os.environ:  # (an explicit setting wins)
            logical, quota = cpu_budget()
            # two threads per granted CPU: the measured optimum under a quota (a throttled thread leaves its CPU to its twin)
            cap = logical if quota is None else max(1, min(logical, int(2 * quota + 0.5)))
            _MAX_THREADS = max(1, min(_MAX_THREADS, cap))
    return _LIB


def use_threads(work: int):
    """Size the OpenMP team to the work of the next call (one thread per 64 Ki units): waking a team of hundreds of threads for a
    50-entry test matrix costs 0.2 s per call on a large host -- 300 calls of it dominated the GPU test tier."""
    L = lib()
    L.grbo_set_num_threads(int(max(1, min(_MAX_THREADS, 1 + int(work) // 65536))))


def use_all_threads():
    lib().grbo_set_num_threads(int(_MAX_THREADS))


def type_name(dtype) -> str:
    if isinstance(dtype, str) and dtype.upper() in TYPE_CODES:
        return dtype.upper()
    return NAME_OF_NP[np.dtype(dtype)]


def unify(t1: str, t2: str) -> str:
    """graphblas/core/dtypes.py:552-568 -- numpy promote_types."""
    if t1 == t2:
        return t1
    return NAME_OF_NP[np.dtype(np.promote_types(NP_OF[t1], NP_OF[t2]))]


def cast(values, tname: str):
    """GraphBLAS typecast: to BOOL is (x != 0); int<->int wraps like C; float->int truncates
    (saturating, NaN -> 0)."""
    values = np.asarray(values)
    dst = np.dtype(NP_OF[tname])
    if values.dtype == dst:
        return values
    if tname == "BOOL":
        return values != 0
    if values.dtype.kind == "f" and dst.kind in "iu":
        info = np.iinfo(dst)
        v = np.where(np.isnan(values), 0.0, values)
        v = np.clip(np.trunc(v), float(info.min), float(info.max))
        out = v.astype(dst)
        out[v >= float(info.max)] = info.max
        return out
    with np.errstate(over="ignore", invalid="ignore"):
        return values.astype(dst)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class OVec:
    """Sparse vector: sorted unique indices + values."""

    def __init__(self, size, idx, vals, tname=None):
        idx = np.asarray(idx, dtype=np.int64)
        vals = np.asarray(vals)
        if tname is None:
            tname = type_name(vals.dtype)
        vals = cast(vals, tname)
        order = np.argsort(idx, kind="stable")
        self.size = int(size)
        self.idx = np.ascontiguousarray(idx[order])
        self.vals = np.ascontiguousarray(vals[order])
        self.tname = tname

    @classmethod
    def empty(cls, size, tname):
        return cls(size, np.zeros(0, np.int64), np.zeros(0, NP_OF[tname]), tname)

    def dense(self, tname=None):
        tname = tname or self.tname
        has = np.zeros(self.size, np.uint8)
        val = np.zeros(self.size, NP_OF[tname])
        has[self.idx] = 1
        val[self.idx] = cast(self.vals, tname)
        return has, val

    @classmethod
    def from_dense(cls, has, val, tname):
        idx = np.flatnonzero(has).astype(np.int64)
        return cls(len(has), idx, np.asarray(val)[idx], tname)

    def __repr__(self):
        return f"OVec({self.tname}, size={self.size}, {dict(zip(self.idx.tolist(), self.vals.tolist()))})"


class OMat:
    """CSR matrix with sorted columns; duplicates combined with dup_op at construction."""

    def __init__(self, nrows, ncols, indptr, indices, values, tname):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int64)
        self.values = np.ascontiguousarray(cast(values, tname))
        self.tname = tname

    @property
    def nvals(self):
        return int(self.indptr[-1])

    @classmethod
    def from_coo(cls, rows, cols, vals, nrows=None, ncols=None, tname=None, dup_op="plus"):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals)
        if np.isscalar(vals) or vals.ndim == 0:
            vals = np.full(len(rows), vals)
        if tname is None:
            tname = type_name(vals.dtype)
        vals = cast(vals, tname)
        if nrows is None:
            nrows = int(rows.max()) + 1 if len(rows) else 0
        if ncols is None:
            ncols = int(cols.max()) + 1 if len(cols) else 0
        order = np.lexsort((cols, rows))
        rows, cols, vals = rows[order], cols[order], vals[order]
        if len(rows):
            head = np.ones(len(rows), bool)
            head[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
            if not head.all():
                seg = np.flatnonzero(head)
                if dup_op == "plus":
                    if tname == "BOOL":
                        vals = np.logical_or.reduceat(vals, seg)
                    else:
                        with np.errstate(over="ignore"):
                            vals = np.add.reduceat(vals, seg).astype(NP_OF[tname])
                elif dup_op == "min":
                    vals = np.minimum.reduceat(vals, seg)
                elif dup_op == "max":
                    vals = np.maximum.reduceat(vals, seg)
                elif dup_op == "first":
                    vals = vals[seg]
                elif dup_op in ("second", "any"):
                    last = np.append(seg[1:], len(rows)) - 1
                    vals = vals[last]
                elif dup_op in ("lor",):
                    vals = np.logical_or.reduceat(vals != 0, seg)
                else:
                    raise ValueError(dup_op)
                rows, cols = rows[seg], cols[seg]
        indptr = np.zeros(nrows + 1, np.int64)
        np.add.at(indptr, rows + 1, 1)
        indptr = np.cumsum(indptr)
        return cls(nrows, ncols, indptr, cols, vals, tname)

    def to_coo(self):
        rows = np.repeat(np.arange(self.nrows, dtype=np.int64), np.diff(self.indptr))
        return rows, self.indices.copy(), self.values.copy()

    def transpose(self):
        Bp = np.zeros(self.ncols + 1, np.int64)
        Bj = np.zeros(max(self.nvals, 1), np.int64)
        Bx = np.zeros(max(self.nvals, 1), self.values.dtype)
        vals = self.values if self.nvals else np.zeros(1, self.values.dtype)
        lib().grbo_transpose(ctypes.c_int64(self.nrows), ctypes.c_int64(self.ncols), _p(self.indptr),
                             _p(self.indices), _p(vals), ctypes.c_int64(self.values.dtype.itemsize),
                             _p(Bp), _p(Bj), _p(Bx))
        return OMat(self.ncols, self.nrows, Bp, Bj[: self.nvals], Bx[: self.nvals], self.tname)

    def astype(self, tname):
        if tname == self.tname:
            return self
        return OMat(self.nrows, self.ncols, self.indptr, self.indices, cast(self.values, tname), tname)

    def empty_like(self, tname=None):
        return OMat(self.nrows, self.ncols, np.zeros(self.nrows + 1, np.int64), np.zeros(0, np.int64),
                    np.zeros(0, NP_OF[tname or self.tname]), tname or self.tname)


def semiring_type(name: str, t_in: str) -> str:
    """Input/compute dtype of a builtin semiring for unified operand dtype ``t_in``.
    lor/land/lxor/lxnor(eq) monoid semirings are BOOL-only: non-bool operands coerce to BOOL
    (graphblas/core/operator/semiring.py:538-548)."""
    monoid, _ = name.split("_", 1)
    if monoid in ("lor", "land", "lxor", "lxnor", "eq"):
        return "BOOL"
    return t_in


def _dense_mask(mask, structure, n):
    if mask is None:
        return None
    mt = np.zeros(n, np.uint8)
    if structure:
        mt[mask.idx] = 1
    else:
        mt[mask.idx] = (cast(mask.vals, "BOOL")).astype(np.uint8)
    return mt


def mxv(A: OMat, u: OVec, semiring="plus_times", *, w: OVec | None = None, out_type=None,
        mask: OVec | None = None, mask_comp=False, mask_struct=False, accum=None, replace=False,
        transpose_a=False, skip_masked_rows=False) -> OVec:
    """w<mask, replace> = accum(w, A (+.x) u).  Returns the new w (a fresh OVec)."""
    if transpose_a:
        A = A.transpose()
    if A.ncols != u.size:
        raise ValueError("DimensionMismatch")
    monoid, mult = semiring.split("_", 1)
    st = semiring_type(semiring, unify(A.tname, u.tname))
    A = A.astype(st)
    u_has, u_val = u.dense(st)
    t_has = np.zeros(A.nrows, np.uint8)
    t_val = np.zeros(A.nrows, NP_OF[st])
    Ax = A.values if A.nvals else np.zeros(1, NP_OF[st])
    mt = _dense_mask(mask, mask_struct, A.nrows)
    row_active = None
    if skip_masked_rows and mt is not None:
        row_active = (mt == 0).astype(np.uint8) if mask_comp else mt.copy()
    use_threads(A.indices.size + A.nrows)
    rc = lib().grbo_mxv(TYPE_CODES[st], OP_CODES[monoid], OP_CODES[mult], ctypes.c_int64(A.nrows),
                        _p(A.indptr), _p(A.indices), _p(Ax), 0, _p(u_has), _p(u_val), _p(row_active),
                        _p(t_has), _p(t_val))
    assert rc == 0
    if w is None:
        w = OVec.empty(A.nrows, out_type or st)
    if w.size != A.nrows or (mask is not None and mask.size != A.nrows):
        raise ValueError("DimensionMismatch")
    wt = w.tname
    w_has, w_val = w.dense()
    t_val = np.ascontiguousarray(cast(t_val, wt))
    rc = lib().grbo_vec_write(TYPE_CODES[wt], ctypes.c_int64(A.nrows), _p(w_has), _p(w_val), _p(t_has),
                              _p(t_val), _p(mt), int(mask_comp), OP_CODES[accum] if accum else -1,
                              int(replace))
    assert rc == 0
    return OVec.from_dense(w_has, w_val, wt)


def vxm(u: OVec, A: OMat, semiring="plus_times", *, transpose_b=False, **kw) -> OVec:
    """w = u (+.x) A  ==  A' (+.x) u with the multiply operands swapped."""
    monoid, mult = semiring.split("_", 1)
    swapped = {"first": "second", "second": "first", "minus": None}.get(mult, mult)
    if swapped is None:
        raise NotImplementedError("non-commutative multiply in vxm oracle")
    return mxv(A, u, f"{monoid}_{swapped}", transpose_a=not transpose_b, **kw)


def _take_csr(ptr_p, ptr_j, ptr_x, nrows, np_t):
    Np = np.ctypeslib.as_array(ptr_p, shape=(nrows + 1,)).copy()
    nnz = int(Np[-1])
    if nnz:
        Nj = np.ctypeslib.as_array(ptr_j, shape=(nnz,)).copy()
        Nx = np.ctypeslib.as_array(ctypes.cast(ptr_x, ctypes.POINTER(np.ctypeslib.as_ctypes_type(np_t))),
                                   shape=(nnz,)).copy()
    else:
        Nj = np.zeros(0, np.int64)
        Nx = np.zeros(0, np_t)
    for p in (ptr_p, ptr_j, ptr_x):
        lib().grbo_free(ctypes.cast(p, ctypes.c_void_p))
    return Np, Nj, Nx


def mxm_product(A: OMat, B: OMat, semiring="plus_times", filt: OMat | None = None, filt_comp=False):
    """T = A (+.x) B as OMat (semiring dtype), optionally restricted to a structural filter."""
    if A.ncols != B.nrows:
        raise ValueError("DimensionMismatch")
    monoid, mult = semiring.split("_", 1)
    st = semiring_type(semiring, unify(A.tname, B.tname))
    A, B = A.astype(st), B.astype(st)
    Ax = A.values if A.nvals else np.zeros(1, NP_OF[st])
    Bx = B.values if B.nvals else np.zeros(1, NP_OF[st])
    Tp = ctypes.POINTER(ctypes.c_int64)()
    Tj = ctypes.POINTER(ctypes.c_int64)()
    Tx = ctypes.c_void_p()
    use_threads(16 * (A.indices.size + B.indices.size) + A.nrows)
    rc = lib().grbo_mxm(TYPE_CODES[st], OP_CODES[monoid], OP_CODES[mult], ctypes.c_int64(A.nrows),
                        ctypes.c_int64(B.ncols), _p(A.indptr), _p(A.indices), _p(Ax), 0,
                        _p(B.indptr), _p(B.indices), _p(Bx), 0,
                        _p(filt.indptr) if filt is not None else None,
                        _p(filt.indices) if filt is not None else None, int(filt_comp),
                        ctypes.byref(Tp), ctypes.byref(Tj), ctypes.byref(Tx))
    assert rc == 0
    p, j, x = _take_csr(Tp, Tj, Tx, A.nrows, NP_OF[st])
    return OMat(A.nrows, B.ncols, p, j, x, st)


def mxm(A: OMat, B: OMat, semiring="plus_times", *, C: OMat | None = None, out_type=None,
        mask: OMat | None = None, mask_comp=False, mask_struct=False, accum=None, replace=False,
        transpose_a=False, transpose_b=False) -> OMat:
    if transpose_a:
        A = A.transpose()
    if transpose_b:
        B = B.transpose()
    T = mxm_product(A, B, semiring)
    if C is None:
        C = OMat(T.nrows, T.ncols, np.zeros(T.nrows + 1, np.int64), [], np.zeros(0, NP_OF[out_type or T.tname]),
                 out_type or T.tname)
    if (C.nrows, C.ncols) != (T.nrows, T.ncols) or (mask is not None and (mask.nrows, mask.ncols) != (T.nrows, T.ncols)):
        raise ValueError("DimensionMismatch")
    ct = C.tname
    T = T.astype(ct)
    M = None
    if mask is not None:
        if mask_struct:
            M = mask
        else:
            r, c, v = mask.to_coo()
            keep = cast(v, "BOOL")
            M = OMat.from_coo(r[keep], c[keep], np.ones(int(keep.sum()), np.bool_), mask.nrows, mask.ncols, "BOOL")
    Cx = C.values if C.nvals else np.zeros(1, NP_OF[ct])
    Tx = T.values if T.nvals else np.zeros(1, NP_OF[ct])
    Np = ctypes.POINTER(ctypes.c_int64)()
    Nj = ctypes.POINTER(ctypes.c_int64)()
    Nx = ctypes.c_void_p()
    rc = lib().grbo_mat_write(TYPE_CODES[ct], ctypes.c_int64(C.nrows), _p(C.indptr), _p(C.indices), _p(Cx),
                              _p(T.indptr), _p(T.indices), _p(Tx),
                              _p(M.indptr) if M is not None else None,
                              _p(M.indices) if M is not None else None,
                              int(M is not None), int(mask_comp), OP_CODES[accum] if accum else -1,
                              int(replace), ctypes.byref(Np), ctypes.byref(Nj), ctypes.byref(Nx))
    assert rc == 0
    p, j, x = _take_csr(Np, Nj, Nx, C.nrows, NP_OF[ct])
    return OMat(C.nrows, C.ncols, p, j, x, ct)


def vec_assign_scalar(w: OVec, value, *, mask: OVec | None = None, mask_comp=False, mask_struct=False, accum=None,
                      replace=False) -> OVec:
    """w<mask, replace>[:] = accum(w, value): GraphBLAS C API 2.0 GrB_Vector_assign with a scalar and GrB_ALL (the
    reference's ``w(mask)[:] << s``, core/vector.py:1979-2035) -- T is ``value`` at every index, then the write rule."""
    use_threads(w.size)
    n = w.size
    if mask is not None and mask.size != n:
        raise ValueError("DimensionMismatch")
    wt = w.tname
    w_has, w_val = w.dense()
    t_has = np.ones(n, np.uint8)
    t_val = np.ascontiguousarray(cast(np.full(n, value), wt))
    mt = _dense_mask(mask, mask_struct, n)
    if n:
        rc = lib().grbo_vec_write(TYPE_CODES[wt], ctypes.c_int64(n), _p(w_has), _p(w_val), _p(t_has), _p(t_val), _p(mt),
                                  int(mask_comp), OP_CODES[accum] if accum else -1, int(replace))
        assert rc == 0
    return OVec.from_dense(w_has, w_val, wt)


_NP_BINOP = {"plus": np.add, "times": np.multiply, "min": np.minimum, "max": np.maximum, "minus": np.subtract,
             "first": lambda a, b: a, "second": lambda a, b: b, "lor": np.logical_or, "land": np.logical_and, "lxor": np.logical_xor}


def _write(w: OVec, t_has, t_val, mask, mask_comp, mask_struct, accum, replace) -> OVec:
    """w<mask, replace> = accum(w, T) through the C oracle's write rule (T dense, in w's type)."""
    n, wt = w.size, w.tname
    w_has, w_val = w.dense()
    mt = _dense_mask(mask, mask_struct, n)
    if n:
        rc = lib().grbo_vec_write(TYPE_CODES[wt], ctypes.c_int64(n), _p(w_has), _p(w_val), _p(np.ascontiguousarray(t_has, np.uint8)),
                                  _p(np.ascontiguousarray(cast(t_val, wt))), _p(mt), int(mask_comp), OP_CODES[accum] if accum else -1,
                                  int(replace))
        assert rc == 0
    return OVec.from_dense(w_has, w_val, wt)


def vec_assign(w: OVec, u, indices, *, mask: OVec | None = None, mask_comp=False, mask_struct=False, accum=None, replace=False) -> OVec:
    """w<mask, replace>(I) = accum(w(I), u) -- GraphBLAS C API 2.0 section 4.3.7.1 (GrB_Vector_assign; the mask has w's size) and
    4.3.7.5 (u a scalar); the reference's ``w(mask, accum)[I] << u`` (core/vector.py:1906-2035).  Restated: Z = w; for every k,
    i = I[k]: without an accumulator Z(i) = u(k) (an absent u(k) deletes Z(i)); with one Z(i) = accum(w(i), u(k)) where both are
    present, u(k) where only it is, w(i) otherwise.  Then w<mask, replace> = Z.  Duplicate indices are undefined in the
    specification and not exercised."""
    use_threads(w.size)
    n, wt = w.size, w.tname
    I = np.asarray(indices, np.int64)
    if I.size and (I.min() < 0 or I.max() >= n):
        raise IndexError("IndexOutOfBound")
    z_has, z_val = w.dense()
    if isinstance(u, OVec):
        if u.size != I.size:
            raise ValueError("DimensionMismatch")
        u_has, u_val = u.dense(wt)
        u_has = u_has.astype(bool)
    else:
        u_has, u_val = np.ones(I.size, bool), cast(np.full(I.size, u), wt)
    if accum:
        both = u_has & z_has[I].astype(bool)
        with np.errstate(over="ignore"):
            comb = np.asarray(_NP_BINOP[accum](z_val[I], u_val)).astype(NP_OF[wt])
        z_val[I[both]] = comb[both]
        only_u = u_has & ~both
        z_val[I[only_u]] = u_val[only_u]
        z_has[I[u_has]] = 1
    else:
        z_val[I[u_has]] = u_val[u_has]
        z_has[I] = u_has.astype(np.uint8)
    if mask is None and not mask_comp:
        return OVec.from_dense(z_has, z_val, wt)
    return _write(w, z_has, z_val, mask, mask_comp, mask_struct, None, replace)


def vec_extract(w: OVec, u: OVec, indices, *, mask: OVec | None = None, mask_comp=False, mask_struct=False, accum=None,
                replace=False) -> OVec:
    """w<mask, replace> = accum(w, u(I)) -- GraphBLAS C API 2.0 section 4.3.6.1 (GrB_Vector_extract); the reference's
    ``w(mask, accum) << u[I]`` (core/vector.py:1906-1975): T(k) = u(I[k]), then the write rule in w's type."""
    use_threads(w.size)
    I = np.asarray(indices, np.int64)
    if I.size != w.size:
        raise ValueError("DimensionMismatch")
    if I.size and (I.min() < 0 or I.max() >= u.size):
        raise IndexError("IndexOutOfBound")
    u_has, u_val = u.dense()
    return _write(w, u_has[I], cast(u_val[I], w.tname), mask, mask_comp, mask_struct, accum, replace)


def vec_reduce(u: OVec, monoid: str):
    """Fold of the stored values with ``monoid`` in u's type, ``None`` when u is empty (GrB_Vector_reduce; reference
    core/vector.py:1635-1684).  Integers wrap like the C types; floating-point sums are left-to-right."""
    use_threads(u.size)
    if u.idx.size == 0:
        return None
    v = u.vals
    if u.tname == "BOOL":
        f = {"lor": np.logical_or, "plus": np.logical_or, "max": np.logical_or, "land": np.logical_and, "times": np.logical_and,
             "min": np.logical_and, "lxor": np.logical_xor, "lxnor": lambda a, b: ~np.logical_xor(a, b)}[monoid]
        return bool(f.reduce(v)) if hasattr(f, "reduce") else bool(__import__("functools").reduce(f, v))
    with np.errstate(over="ignore"):
        if monoid == "plus":
            return np.add.reduce(v, dtype=v.dtype).item()
        if monoid == "times":
            return np.multiply.reduce(v, dtype=v.dtype).item()
        if monoid == "min":
            return v.min().item()
        if monoid == "max":
            return v.max().item()
    raise NotImplementedError(monoid)


def vec_ewise(u: OVec, v: OVec, binop: str, *, union: bool) -> OVec:
    """T = u (op) v on the union (eWiseAdd: single entries pass through) or the intersection (eWiseMult) of the patterns, in
    the unified type (GraphBLAS C API 2.0 eWiseAdd / eWiseMult; reference core/vector.py:960-1150)."""
    use_threads(u.size)
    if u.size != v.size:
        raise ValueError("DimensionMismatch")
    t = unify(u.tname, v.tname)
    hu, du = u.dense(t)
    hv, dv = v.dense(t)
    hu, hv = hu.astype(bool), hv.astype(bool)
    f = {"plus": np.add, "times": np.multiply, "min": np.minimum, "max": np.maximum, "minus": np.subtract,
         "first": lambda a, b: a, "second": lambda a, b: b, "lor": np.logical_or, "land": np.logical_and,
         "lxor": np.logical_xor}[binop]
    with np.errstate(over="ignore"):
        both = np.asarray(f(du, dv)).astype(NP_OF[t])
    has = (hu | hv) if union else (hu & hv)
    val = np.where(hu & hv, both, np.where(hu, du, dv)).astype(NP_OF[t])
    return OVec.from_dense(has.astype(np.uint8), val, t)


def mat_reduce_rows(A: OMat, monoid: str, *, columns=False) -> OVec:
    """w(i) = fold of row i of A with the monoid (rows without entries stay empty) -- GrB_Matrix_reduce_Monoid; as the
    library does, stated as the mxv over (monoid, first) with a full operand."""
    if columns:
        A = A.transpose()
    ones = OVec(A.ncols, np.arange(A.ncols), np.ones(A.ncols, NP_OF[A.tname]), A.tname)
    return mxv(A, ones, f"{monoid}_first")


def num_threads() -> int:
    """threads of a full-size call (the host's OpenMP default)"""
    lib()
    return int(_MAX_THREADS)
