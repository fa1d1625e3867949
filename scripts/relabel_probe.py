"""Round-4 probe: how much of the pull SpMV's time is label locality?  Relabels the R-MAT graph (rows and columns with the SAME
permutation, so it is the same graph) before the library sees it and runs bench.py's headline on it.
    GRB_RELABEL=none|degree|colcount|longfirst python scripts/relabel_probe.py [bench.py arguments]
degree: vertices by falling (row length + column count); colcount: by falling column count; longfirst: rows of >= 64 entries first,
then by falling column count."""
import os, sys, runpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from graphblas_amd import synthetic

mode = os.environ.get("GRB_RELABEL", "degree")
orig = synthetic.rmat_csr


def relabeled(scale, *a, **kw):
    out = orig(scale, *a, **kw)
    if mode == "none" or kw.get("row_range") is not None or kw.get("row_ranges") is not None:
        return out
    indptr, col = out
    n = 1 << scale
    rl = indptr[1:] - indptr[:-1]
    cc = torch.bincount(col.long(), minlength=n)
    if mode == "degree":
        key = rl + cc
    elif mode == "colcount":
        key = cc * (1 << 24) + torch.clamp(rl, max=(1 << 24) - 1)
    else:
        key = (rl >= 64).long() * (1 << 40) + cc * (1 << 16) + torch.clamp(rl, max=65535)
    order = torch.argsort(-key, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=order.device)
    del order, key, cc
    rows = torch.repeat_interleave(torch.arange(n, device=col.device), rl)
    k = rank[rows] * n + rank[col.long()]
    del rows, col, indptr, rank, rl
    k = torch.sort(k).values
    row = torch.div(k, n, rounding_mode="floor")
    col2 = (k - row * n).to(torch.int32)
    del k
    counts = torch.bincount(row, minlength=n)
    del row
    ip = torch.zeros(n + 1, dtype=torch.int64, device=col2.device)
    ip[1:] = torch.cumsum(counts, 0)
    torch.cuda.empty_cache()
    return ip, col2


synthetic.rmat_csr = relabeled
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
