"""Synthetic inputs of BASELINE.md section 3: Graph500 R-MAT / Kronecker graphs (a,b,c,d =
0.57,0.19,0.19,0.05, edge factor 16, vertex labels scrambled by a fixed permutation, duplicates
merged, nnz reported after dedupe) and uniform-random sparse matrices.

Generation runs in torch (on the GPU for the bench, on the CPU for small test graphs) -- it is input
plumbing, not part of the measured path."""
from __future__ import annotations

import numpy as np


def rmat_csr(scale, edge_factor=16, *, seed=None, device="cuda", abcd=(0.57, 0.19, 0.19, 0.05), row_range=None,
             row_ranges=None, chunk_edges=1 << 27):
    """Returns (indptr int64[n_rows+1], col int32[nnz]) torch tensors on ``device``; sorted, deduped.

    ``row_range=(lo, hi)`` keeps only rows lo..hi-1 (1-D row sharding: every rank draws the same
    edge stream from the same seed and keeps its block), re-based so that row lo becomes row 0.
    ``row_ranges=[(lo, hi), ...]`` does the same for several blocks from ONE pass over the edge stream
    and returns a list of (indptr, col) pairs (a rank that owns several row blocks: sharded.OverlappedMxv).
    """
    import torch

    n = 1 << scale
    n_edges = edge_factor * n
    seed = scale if seed is None else seed
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + seed)
    perm = torch.randperm(n, generator=gen, device=device)
    a, b, c, _ = abcd
    ranges = list(row_ranges) if row_ranges is not None else [row_range if row_range is not None else (0, n)]
    whole = row_ranges is None and row_range is None
    keys = [[] for _ in ranges]
    done = 0
    while done < n_edges:
        m = min(chunk_edges, n_edges - done)
        src = torch.zeros(m, dtype=torch.int64, device=device)
        dst = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            r = torch.rand(m, generator=gen, device=device)
            src_bit = r >= (a + b)
            dst_bit = ((r >= a) & (r < a + b)) | (r >= (a + b + c))
            src = (src << 1) | src_bit
            dst = (dst << 1) | dst_bit
        src = perm[src]
        dst = perm[dst]
        for k, (lo, hi) in enumerate(ranges):
            if whole:
                keys[k].append(torch.unique(src * n + dst))
            else:
                keep = (src >= lo) & (src < hi)
                keys[k].append(torch.unique((src[keep] - lo) * n + dst[keep]))
        del src, dst
        done += m
    out = []
    for k, (lo, hi) in enumerate(ranges):
        key = torch.unique(torch.cat(keys[k])) if len(keys[k]) > 1 else keys[k][0]
        keys[k] = None
        rows = hi - lo
        row = torch.div(key, n, rounding_mode="floor")
        col = (key - row * n).to(torch.int32)
        counts = torch.bincount(row, minlength=rows)
        indptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
        indptr[1:] = torch.cumsum(counts, 0)
        out.append((indptr, col))
    return out if row_ranges is not None else out[0]


def edge_weights(col, seed, dtype=None, device=None):
    """Integer-valued weights U{1..255} (exact in fp32 => min_plus / plus_times on them are bit-exact)."""
    import torch

    gen = torch.Generator(device=col.device)
    gen.manual_seed(7000 + seed)
    w = torch.randint(1, 256, (col.numel(),), generator=gen, device=col.device)
    return w.to(dtype or torch.float32)


def edge_weights_real(col, seed, dtype=None):
    """Real-valued weights U[0,1) (FP32): more distinct values than any dictionary holds; min_plus stays bit-exact against a reference
    that forms the same single-rounding sums a + x and takes their minimum."""
    import torch

    gen = torch.Generator(device=col.device)
    gen.manual_seed(9000 + seed)
    return torch.rand(col.numel(), generator=gen, device=col.device, dtype=dtype or torch.float32)


def uniform_coo(nrows, ncols, density, seed, np_dtype=np.float64):
    """configs[0]: each entry present with probability ``density``; values U[0,1) (numpy, host)."""
    rng = np.random.default_rng(seed)
    mask = rng.random((nrows, ncols)) < density
    r, c = np.nonzero(mask)
    return r.astype(np.int64), c.astype(np.int64), rng.random(r.size).astype(np_dtype)
