"""Synthetic inputs of BASELINE.md section 3: Graph500 R-MAT / Kronecker graphs (a,b,c,d =
0.57,0.19,0.19,0.05, edge factor 16, vertex labels scrambled by a fixed permutation, duplicates
merged, nnz reported after dedupe) and uniform-random sparse matrices.

Generation runs in torch (on the GPU for the bench, on the CPU for small test graphs) -- it is input
plumbing, not part of the measured path."""
from __future__ import annotations

import numpy as np


def rmat_csr(scale, edge_factor=16, *, seed=None, device="cuda", abcd=(0.57, 0.19, 0.19, 0.05), row_range=None,
             row_ranges=None, chunk_edges=1 << 27, relabel=None, stripes=None, col_counts_only=False):
    """Returns (indptr int64[n_rows+1], col int32[nnz]) torch tensors on ``device``; sorted, deduped.

    ``row_range=(lo, hi)`` keeps only rows lo..hi-1 (1-D row sharding: every rank draws the same
    edge stream from the same seed and keeps its block), re-based so that row lo becomes row 0.
    ``row_ranges=[(lo, hi), ...]`` does the same for several blocks from ONE pass over the edge stream
    and returns a list of (indptr, col) pairs (a rank that owns several row blocks: sharded.OverlappedMxv).

    ``relabel="popularity"`` (round 5): instead of the scrambled labels the vertices are numbered by falling in-degree (ties: falling
    out-degree) of the whole edge stream -- what an application does ONCE before it shards a graph whose matrix it will multiply with many
    times; every rank derives the same numbering from the same stream.  ``stripes=(B, w, r)``: keep the rows of the stripes of ``B``
    consecutive rows numbered r, r + w, r + 2 w, ... (block-cyclic 1-D sharding: with ranked labels a contiguous block would give rank 0
    all the hubs), re-based so that local row = (stripe // w) * B + row % B.
    """
    import torch

    n = 1 << scale
    n_edges = edge_factor * n
    seed = scale if seed is None else seed
    a, b, c, _ = abcd

    def stream():
        """(perm, generator of (src, dst) chunks before the labels are applied) -- the same sequence at every call."""
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + seed)
        perm = torch.randperm(n, generator=gen, device=device)

        def chunks():
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
                yield src, dst
                done += m

        return perm, chunks()

    perm, chunks = stream()
    if relabel == "popularity":
        indeg = torch.zeros(n, dtype=torch.int64, device=device)
        outdeg = torch.zeros(n, dtype=torch.int64, device=device)
        for src, dst in chunks:
            indeg += torch.bincount(perm[dst], minlength=n)
            outdeg += torch.bincount(perm[src], minlength=n)
        order = torch.argsort(indeg * (n_edges + 1) + outdeg, descending=True, stable=True)
        rank = torch.empty(n, dtype=torch.int64, device=device)
        rank[order] = torch.arange(n, device=device)
        del indeg, outdeg, order
        perm0, chunks = stream()
        perm = rank[perm0]
        del rank, perm0
    elif relabel is not None:
        raise ValueError(f"relabel={relabel!r}")
    if col_counts_only:
        # (the column histogram of the raw edge stream -- duplicates counted: a ranking signal, one pass, no keys kept)
        counts = torch.zeros(n, dtype=torch.int64, device=device)
        for _src, dst in chunks:
            counts += torch.bincount(perm[dst], minlength=n)
        return counts
    if stripes is not None:
        B, sw, sr = stripes
        assert row_range is None and row_ranges is None and n % (B * sw) == 0
        ranges = [(0, n // sw)]
    else:
        ranges = list(row_ranges) if row_ranges is not None else [row_range if row_range is not None else (0, n)]
    whole = row_ranges is None and row_range is None and stripes is None
    keys = [[] for _ in ranges]
    for src, dst in chunks:
        src = perm[src]
        dst = perm[dst]
        if stripes is not None:
            stripe = torch.div(src, B, rounding_mode="floor")
            keep = (stripe % sw) == sr
            local = torch.div(stripe[keep], sw, rounding_mode="floor") * B + src[keep] % B
            keys[0].append(torch.unique(local * n + dst[keep]))
            del stripe, keep, local
        else:
            for k, (lo, hi) in enumerate(ranges):
                if whole:
                    keys[k].append(torch.unique(src * n + dst))
                else:
                    keep = (src >= lo) & (src < hi)
                    keys[k].append(torch.unique((src[keep] - lo) * n + dst[keep]))
        del src, dst
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


def rmat_col_counts(scale, edge_factor=16, *, seed=None, device="cuda", abcd=(0.57, 0.19, 0.19, 0.05), chunk_edges=1 << 27):
    """Reference counts of the columns of the WHOLE edge stream :func:`rmat_csr` draws (duplicates counted), int64[n] on ``device`` -- the
    ranking signal the ranks of a sharded run obtain by all-reducing the histograms of their blocks; a process that holds one block only
    (``bench.py --block``) draws it from the same stream instead."""
    return rmat_csr(scale, edge_factor, seed=seed, device=device, abcd=abcd, chunk_edges=chunk_edges, col_counts_only=True)


def stripe_rows(n, B, w, r, device="cuda"):
    """Global row numbers of the local rows of rank r in the block-cyclic sharding ``stripes=(B, w, r)`` of :func:`rmat_csr`."""
    import torch

    local = torch.arange(n // w, device=device)
    return (torch.div(local, B, rounding_mode="floor") * w + r) * B + local % B


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
