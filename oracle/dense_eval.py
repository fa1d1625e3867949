"""TEST INFRASTRUCTURE ONLY -- an independent, deliberately naive restatement.

Dense brute-force evaluation of ``C<M,replace> = accum(C, A (+.x) B)`` with Python loops
over (value, present) pairs.  It shares no code with oracle/grb_oracle.c; tests use it
to cross-check the C oracle on small random cases -- in particular for ``lor_land`` and
``any_pair``, for which the reference holds no value-level literal.

Semantics follow the GraphBLAS C API 2.0 description of GrB_mxm / GrB_mxv / GrB_vxm, the
calls named at graphblas/core/matrix.py:2254,2321 and graphblas/core/vector.py:1370.
"""
from __future__ import annotations

import numpy as np


def _wrap(x, np_t):
    """Bring a Python number into dtype np_t with C-style wrap-around for integers."""
    dt = np.dtype(np_t)
    if dt.kind == "b":
        return bool(x)
    if dt.kind in "iu":
        bits = dt.itemsize * 8
        x = int(x) & ((1 << bits) - 1)
        if dt.kind == "i" and x >= 1 << (bits - 1):
            x -= 1 << bits
        return x
    return dt.type(x)


def binop(name, a, b, np_t):
    dt = np.dtype(np_t)
    if dt.kind == "b":
        a, b = bool(a), bool(b)
        table = {
            "first": a, "second": b, "pair": True, "plus": a or b, "max": a or b, "lor": a or b,
            "times": a and b, "min": a and b, "land": a and b, "minus": a != b, "lxor": a != b,
            "lxnor": a == b, "any": a,
        }
        return table[name]
    if name == "first":
        return a
    if name == "second":
        return b
    if name == "pair":
        return _wrap(1, np_t)
    if name == "any":
        return a
    if name in ("lor", "land", "lxor", "lxnor"):
        x, y = a != 0, b != 0
        r = {"lor": x or y, "land": x and y, "lxor": x != y, "lxnor": x == y}[name]
        return _wrap(int(r), np_t)
    if dt.kind in "iu":
        a, b = int(a), int(b)
        r = {"plus": a + b, "minus": a - b, "times": a * b, "min": min(a, b), "max": max(a, b)}[name]
        return _wrap(r, np_t)
    a, b = dt.type(a), dt.type(b)
    with np.errstate(all="ignore"):
        if name == "plus":
            return a + b
        if name == "minus":
            return a - b
        if name == "times":
            return a * b
        if name == "min":
            return np.fmin(a, b)
        if name == "max":
            return np.fmax(a, b)
    raise KeyError(name)


def matmul(Aval, Ahas, Bval, Bhas, monoid, mult, np_t):
    """Dense (value, present) product.  A: m x k, B: k x n."""
    m, k = Ahas.shape
    k2, n = Bhas.shape
    assert k == k2
    Tval = np.zeros((m, n), np_t)
    Thas = np.zeros((m, n), bool)
    for i in range(m):
        for j in range(n):
            acc = None
            for kk in range(k):
                if Ahas[i, kk] and Bhas[kk, j]:
                    p = binop(mult, Aval[i, kk], Bval[kk, j], np_t)
                    if acc is None:
                        acc = p
                    elif monoid != "any":
                        acc = binop(monoid, acc, p, np_t)
            if acc is not None:
                Thas[i, j] = True
                Tval[i, j] = acc
    return Tval, Thas


def write(Cval, Chas, Tval, Thas, Mval=None, Mhas=None, *, comp=False, struct=False, accum=None,
          replace=False, np_t=None):
    """The C<M,replace> = accum(C,T) rule on dense (value, present) arrays; returns new (val, has)."""
    np_t = np_t or Cval.dtype
    Nval = np.zeros_like(Cval)
    Nhas = np.zeros_like(Chas)
    it = np.ndindex(Chas.shape)
    for ix in it:
        if Mhas is None:
            m = True
        else:
            m = bool(Mhas[ix]) and (True if struct else bool(Mval[ix] != 0))
            if comp:
                m = not m
        if accum is not None:
            if Chas[ix] and Thas[ix]:
                zh, zv = True, binop(accum, Cval[ix], Tval[ix], np_t)
            elif Chas[ix]:
                zh, zv = True, Cval[ix]
            elif Thas[ix]:
                zh, zv = True, Tval[ix]
            else:
                zh, zv = False, 0
        else:
            zh, zv = bool(Thas[ix]), Tval[ix]
        if m:
            Nhas[ix], Nval[ix] = zh, (zv if zh else 0)
        elif not replace:
            Nhas[ix], Nval[ix] = Chas[ix], Cval[ix]
    return Nval, Nhas
