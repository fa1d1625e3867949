"""Round 6: where do the references of the bench graph go?  For the popularity-ordered scale-S R-MAT graph: the share of the column
references below code K, by row class (short < 64 entries, mid 64..1023, hub >= 1024), and the entries by row-length bucket.
Feeds the sizing of LDS heads (grb_mxv_rtile.inc) and of the class levels.  Usage: python scripts/ref_shares.py --scale 24"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scale", type=int, default=24)
    p.add_argument("--device", default="cuda")
    a = p.parse_args()
    import torch

    from graphblas_amd import synthetic

    n = 1 << a.scale
    indptr, col = synthetic.rmat_csr(a.scale, device=a.device)
    rowlen = indptr[1:] - indptr[:-1]
    colcnt = torch.bincount(col.long(), minlength=n)
    order = torch.argsort(colcnt * (int(rowlen.max().item()) + 1) + rowlen, descending=True, stable=True)
    rank = torch.empty(n, dtype=torch.int64, device=a.device)
    rank[order] = torch.arange(n, device=a.device)
    code = rank[col.long()]
    rl = torch.repeat_interleave(rowlen, rowlen)
    out = {"scale": a.scale, "nnz": int(col.numel()), "shares_below_code": {}, "entries_by_rowlen_bucket": {}}
    Ks = [1 << k for k in range(8, a.scale + 1)] + [10240, 20480, 40960, 163840, 638976, 2555904]
    for name, sel in (("short", rl < 64), ("mid", (rl >= 64) & (rl < 1024)), ("hub", rl >= 1024), ("all", rl >= 0)):
        c = code[sel]
        tot = int(c.numel())
        out["shares_below_code"][name] = {"entries": tot, **{str(K): round(int((c < K).sum().item()) / max(tot, 1), 4) for K in sorted(Ks)}}
    edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 40]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (rowlen >= lo) & (rowlen < hi)
        out["entries_by_rowlen_bucket"][f"{lo}-{hi - 1 if hi < (1 << 40) else 'max'}"] = {"rows": int(m.sum().item()), "entries": int(rowlen[m].sum().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
