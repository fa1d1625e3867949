#!/usr/bin/env python3
"""Which C symbols would an unmodified python-graphblas ask of `lib` -- and which of them libgrb_mi355x.so exports.

Runs ONLY in the build container (it reads /root/reference; nothing of it is copied): collects
  (a) attributes read directly:            ``lib.NAME``
  (b) C function names passed as strings:  ``call("GrB_Matrix_new", ...)``, ``cfunc_name="GrB_mxv"``, f-strings such as
      ``f"GrB_Matrix_build_{dtype.name}"`` (expanded over the 11 builtin type names)
  (c) builtin operator handles the import-time code names explicitly (``binary.pair``, ``unary.one``, ``select.valuene``, ...)
from the reference's non-test, non-SuiteSparse-only (core/ss/*) sources, splits them into the GraphBLAS C API 2.0 names
(GrB_*) and SuiteSparse extensions (GxB_*), and diffs them against ``nm -D`` of the built library.

    python scripts/lib_surface_report.py [--markdown INTEGRATION_surface.md]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/graphblas"
LIB = os.path.join(ROOT, "python-graphblas_amd", "csrc", "libgrb_mi355x.so")
TYPES = ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"]


def sources(include_ss):
    for base, _dirs, files in os.walk(REF):
        if "/tests" in base or base.endswith("tests"):
            continue
        if not include_ss and ("/core/ss" in base or base.endswith("/ss")):
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def collect(include_ss=False):
    attrs, strings, templ = set(), set(), set()
    for path in sources(include_ss):
        text = open(path, encoding="utf-8").read()
        attrs.update(re.findall(r"\blib\.(G[rx]B_[A-Za-z0-9_]+)", text))
        for m in re.finditer(r"[\"'](G[rx]B_[A-Za-z0-9_{}.]+)[\"']", text):
            s = m.group(1)
            if "{" in s:
                templ.add(s)
            else:
                strings.add(s)
    expanded = set()
    for t in templ:
        expanded.update(expand(t))
    return attrs, strings | expanded, templ


OPCLASSES = ["BinaryOp", "Monoid", "Semiring"]
APPLY_METHODS = ["BinaryOp1st", "BinaryOp2nd", "IndexOp"]


def expand(t):
    """The names an f-string template stands for: {...name} / {output_dtype} -> the 11 builtin types, {op.opclass} -> the operator
    classes, {cfunc_method} -> the apply flavours, {type_name} -> Matrix / Vector / Scalar; anything else stays unexpanded."""
    m = re.search(r"\{([^}]*)\}", t)
    if not m:
        return [t]
    key, out = m.group(1), []
    if key.endswith("opclass"):
        subs = OPCLASSES
    elif key == "cfunc_method":
        subs = APPLY_METHODS
    elif key == "type_name":
        subs = ["Matrix", "Vector", "Scalar"]
    elif key.endswith("name") or key == "output_dtype":
        subs = TYPES
    else:
        return []
    for sub in subs:
        out.extend(expand(t[: m.start()] + sub + t[m.end():]))
    return out


def exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    syms = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith(("GrB_", "GxB_", "GrX_"))}
    # type names and enumeration constants reach cffi's `lib` through the cdef (the header), not through the symbol table
    hdr = subprocess.check_output(["gcc", "-E", "-P", os.path.join(ROOT, "include", "grb_mi355x.h")], text=True)
    decl = set(re.findall(r"\b(G[rx]B_[A-Za-z0-9_]+)\b", hdr)) - syms
    return syms, decl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--markdown", default=None)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("the reference is not on this machine: this report is generated in the build container only")
    syms, decl = exported()
    have = syms | decl
    attrs, funcs, templ = collect(include_ss=False)
    ss_attrs, ss_funcs, _ = collect(include_ss=True)
    want = attrs | funcs
    grb = sorted(n for n in want if n.startswith("GrB_"))
    gxb = sorted(n for n in want if n.startswith("GxB_"))
    miss_grb = [n for n in grb if n not in have]
    miss_gxb = [n for n in gxb if n not in have]
    ss_only = sorted((ss_attrs | ss_funcs) - want)
    lines = [
        "| group | named by the reference | exported by libgrb_mi355x.so | missing |", "|---|---|---|---|",
        f"| GraphBLAS C API names (`GrB_*`) outside core/ss | {len(grb)} | {len(grb) - len(miss_grb)} | {len(miss_grb)} |",
        f"| SuiteSparse extensions (`GxB_*`) outside core/ss | {len(gxb)} | {len(gxb) - len(miss_gxb)} | {len(miss_gxb)} |",
        f"| names only core/ss/* touches (SuiteSparse-specific layer, not loaded by the vanilla backend) | {len(ss_only)} | "
        f"{len([n for n in ss_only if n in have])} | {len([n for n in ss_only if n not in have])} |",
        f"| library exports in total (+ names the header declares as types / enum constants) | | {len(syms)} (+ {len(decl)}) | |",
    ]
    print("\n".join(lines))
    print("\nmissing GrB_*:", " ".join(miss_grb) or "(none)")
    print("\nmissing GxB_* (outside core/ss):", " ".join(miss_gxb) or "(none)")
    print("\nunexpanded templates:", " ".join(sorted(t for t in templ if not expand(t))))
    if args.markdown:
        with open(args.markdown, "w") as f:
            f.write("\n".join(lines) + "\n\nMissing `GrB_*`: " + (", ".join(f"`{n}`" for n in miss_grb) or "none") +
                    "\n\nMissing `GxB_*` named outside core/ss: " + (", ".join(f"`{n}`" for n in miss_gxb) or "none") + "\n")


if __name__ == "__main__":
    main()
