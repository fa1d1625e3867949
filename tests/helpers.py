"""Shared helpers: run one golden/random case through the oracle."""
import numpy as np

from oracle import grb_oracle as O


def o_obj(spec):
    if spec["kind"] == "matrix":
        return O.OMat.from_coo(spec["rows"], spec["cols"], np.asarray(spec["vals"], O.NP_OF[spec["dtype"]]),
                               spec["nrows"], spec["ncols"], spec["dtype"])
    return O.OVec(spec["size"], spec["idx"], np.asarray(spec["vals"], O.NP_OF[spec["dtype"]]), spec["dtype"])


def oracle_case(case, inputs):
    """Evaluate one case dict (see tests/golden/make_reference_literals.py) with the oracle."""
    g = lambda k: o_obj(inputs[case[k]]) if case.get(k) else None
    kw = dict(mask=g("mask"), mask_comp=case.get("mask_comp", False), mask_struct=case.get("mask_struct", False),
              accum=case.get("accum"), replace=case.get("replace", False))
    if case["op"] == "mxm":
        return O.mxm(g("A"), g("B"), case["semiring"], C=g("C"), transpose_a=case.get("ta", False),
                     transpose_b=case.get("tb", False), **kw)
    if case["op"] == "mxv":
        return O.mxv(g("A"), g("u"), case["semiring"], w=g("C"), transpose_a=case.get("ta", False), **kw)
    if case["op"] == "vxm":
        return O.vxm(g("u"), g("A"), case["semiring"], w=g("C"), transpose_b=case.get("tb", False), **kw)
    raise ValueError(case["op"])


def same(got, exp_spec, rtol=0.0):
    exp = o_obj(exp_spec)
    if isinstance(exp, O.OVec):
        assert got.size == exp.size
        assert got.idx.tolist() == exp.idx.tolist(), (got, exp)
        gv, ev = got.vals, exp.vals
    else:
        assert (got.nrows, got.ncols) == (exp.nrows, exp.ncols)
        assert got.indptr.tolist() == exp.indptr.tolist()
        assert got.indices.tolist() == exp.indices.tolist()
        gv, ev = got.values, exp.values
    assert got.tname == exp.tname, (got.tname, exp.tname)
    if rtol:
        np.testing.assert_allclose(gv, ev, rtol=rtol)
    else:
        assert gv.tolist() == ev.tolist(), (gv, ev)
