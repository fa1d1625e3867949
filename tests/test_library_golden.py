"""Every known-answer literal the reference's own tests and docs hold for the path (tests/golden/reference_literals.json: 25 mxm / mxv /
vxm cases, the primer's SSSP loop, 30 cases of the operations around the path), run THROUGH THE LIBRARY from the data file -- the HIP
kernels on the GPU tier, the same sources under the wave64 emulator on the CPU tier.  The fixtures are data (inputs and expected outputs
with the reference line each comes from); this file holds no transcription of the reference's test bodies: one generic runner turns a
case record into calls of the host API.  (tests/test_oracle_golden.py pins the ORACLE on the same records.)"""
import json
import os

import numpy as np
import pytest

from tests.backend import DEVICES, bind

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))
_NP = {"BOOL": bool, "INT8": np.int8, "INT16": np.int16, "INT32": np.int32, "INT64": np.int64, "UINT8": np.uint8, "UINT16": np.uint16,
       "UINT32": np.uint32, "UINT64": np.uint64, "FP32": np.float32, "FP64": np.float64}


@pytest.fixture(params=DEVICES)
def gb(request):
    return bind(request.param)


def obj(gb, spec):
    spec = _G["inputs"][spec] if isinstance(spec, str) else spec
    if spec["kind"] == "matrix":
        return gb.Matrix.from_coo(spec["rows"], spec["cols"], np.asarray(spec["vals"], _NP[spec["dtype"]]), dtype=spec["dtype"], nrows=spec["nrows"],
                                  ncols=spec["ncols"])
    return gb.Vector.from_coo(spec["idx"], np.asarray(spec["vals"], _NP[spec["dtype"]]), dtype=spec["dtype"], size=spec["size"])


def check(got, exp):
    assert got.dtype.name == exp["dtype"], (got.dtype.name, exp["dtype"])
    if exp["kind"] == "matrix":
        assert (got.nrows, got.ncols) == (exp["nrows"], exp["ncols"])
        r, c, v = got.to_coo()
        order = np.lexsort((exp["cols"], exp["rows"]))
        assert r.tolist() == np.asarray(exp["rows"])[order].tolist() and c.tolist() == np.asarray(exp["cols"])[order].tolist()
        ev = np.asarray(exp["vals"], _NP[exp["dtype"]])[order]
    else:
        assert got.size == exp["size"]
        i, v = got.to_coo()
        assert i.tolist() == list(exp["idx"])
        ev = np.asarray(exp["vals"], _NP[exp["dtype"]])
    if exp["dtype"].startswith("FP"):
        np.testing.assert_allclose(v, ev, rtol=1e-6)
    else:
        assert v.tolist() == ev.tolist()


def mask_of(gb, case):
    if not case.get("mask"):
        return None
    m = obj(gb, case["mask"])
    m = m.S if case.get("mask_struct") else m.V
    return ~m if case.get("mask_comp") else m


@pytest.mark.parametrize("case", _G["cases"], ids=[c["name"] for c in _G["cases"]])
def test_reference_literal_through_the_library(gb, case):
    sr = getattr(gb.semiring, case["semiring"])
    A = obj(gb, case["A"]) if case.get("A") else None
    if case["op"] == "mxm":
        B = obj(gb, case["B"])
        left, right = (A.T if case.get("ta") else A), (B.T if case.get("tb") else B)
        expr = left.mxm(right, sr)
    elif case["op"] == "mxv":
        expr = (A.T if case.get("ta") else A).mxv(obj(gb, case["u"]), sr)
    else:
        expr = obj(gb, case["u"]).vxm(A.T if case.get("tb") else A, sr)
    if "expect_shape" in case:
        got = expr.new()
        assert [got.nrows, got.ncols] == case["expect_shape"]
        return
    mask, accum = mask_of(gb, case), (getattr(gb.binary, case["accum"]) if case.get("accum") else None)
    if case.get("C"):
        C = obj(gb, case["C"])
        if mask is not None:
            C(mask, accum=accum, replace=case.get("replace", False)) << expr
        elif accum is not None:
            C(accum=accum) << expr
        else:
            C << expr
        got = C
    else:
        got = expr.new(mask=mask) if mask is not None else expr.new()
    check(got, case["expect"])


def test_primer_sssp_through_the_library(gb):
    s = _G["sssp"]
    G, v = obj(gb, s["G"]), obj(gb, s["start"])
    for _ in range(10):
        w = v.dup()
        v(gb.binary.min) << v.vxm(G, gb.semiring.min_plus)
        if v.isequal(w):
            break
    check(v, s["expect"])


@pytest.mark.parametrize("case", _G["vector_ops"], ids=lambda c: c["name"])
def test_operations_around_the_path_through_the_library(gb, case):
    op = case["op"]
    if op == "reduce":
        assert obj(gb, case["u"]).reduce(getattr(gb.monoid, case["monoid"])).new().value == case["expect_scalar"]
    elif op in ("ewise_mult", "ewise_add"):
        u, v = obj(gb, case["u"]), obj(gb, case["v"])
        got = (u.ewise_add if op == "ewise_add" else u.ewise_mult)(v, getattr(gb.binary, case["binop"])).new()
        check(got, case["expect"])
    elif op in ("reduce_rowwise", "reduce_columnwise"):
        A = obj(gb, case["A"])
        how = getattr(gb.agg, case["agg"]) if "agg" in case else getattr(gb.monoid, case["monoid"])  # (an aggregator or the monoid behind it)
        got = (A.reduce_columnwise if op == "reduce_columnwise" else A.reduce_rowwise)(how).new()
        check(got, case["expect"])
    elif op == "reduce_scalar":
        assert obj(gb, case["A"]).reduce_scalar(getattr(gb.monoid, case["monoid"])).new().value == case["expect_scalar"]
    elif op == "resize":
        x = obj(gb, case["A"] if "A" in case else case["w"])
        x.resize(*case["to"])
        check(x, case["expect"])
    elif op == "agg_matvec":  # the exists / count aggregators: one semiring mat-vec inside the host (agg.py)
        A = obj(gb, case["A"])
        ag = {"any_pair": gb.agg.exists, "plus_pair": gb.agg.count}[case["semiring"]]
        check((A.reduce_columnwise if case.get("columns") else A.reduce_rowwise)(ag).new(), case["expect"])
    elif op == "agg_matvec_scalar":
        ag = {"any_pair": gb.agg.exists, "plus_pair": gb.agg.count}[case["semiring"]]
        assert obj(gb, case["A"]).reduce_scalar(ag).new().value == case["expect_scalar"]
    elif op == "agg_exists_scalar":
        got = obj(gb, case["A"]).reduce_scalar(gb.agg.exists).new() if "A" in case else obj(gb, case["u"]).reduce(gb.agg.exists).new()
        assert got.value == case["expect_scalar"]
    elif op == "extract_element":
        u = obj(gb, case["u"])
        for i, want in case["probes"]:
            assert u[i].new().value == want
    elif op == "set_element":
        w = obj(gb, case["w"])
        for k, (i, x) in enumerate(case["sets"]):
            if k % 2:
                w[i] << x  # (both spellings of the reference's test)
            else:
                w[i] = x
        check(w, case["expect"])
    elif op == "remove_element":
        w = obj(gb, case["w"])
        del w[case["index"]]
        check(w, case["expect"])
    elif op == "extract":
        u = obj(gb, case["u"])
        w = gb.Vector(u.dtype, len(case["indices"]))
        w << u[case["indices"]]
        check(w, case["expect"])
        check(u[np.array(case["indices"])].new(), case["expect"])
    elif op == "assign_vector":
        w = obj(gb, case["w"])
        w[case["indices"]] = obj(gb, case["u"])
        check(w, case["expect"])
    elif op == "assign_scalar_at":
        w = obj(gb, case["w"])
        w[case["indices"]] = case["value"]
        check(w, case["expect"])
    elif op == "transpose":
        A = obj(gb, case["A"])
        check(A.T.new(), case["expect"])
        C = gb.Matrix(A.dtype, A.ncols, A.nrows)
        C << A.T
        check(C, case["expect"])
    else:
        assert op == "assign_scalar", op
        w = obj(gb, case["w"])
        if "mask" in case:
            m = obj(gb, case["mask"]).V
            w(mask=~m if case.get("mask_comp") else m)[:] << case["value"]
        else:
            w[:] << case["value"]
        check(w, case["expect"])
