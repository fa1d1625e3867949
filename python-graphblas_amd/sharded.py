"""1-D row sharding of mxv / vxm / mxm over the GPUs of one node.  One process per GPU; collectives
go through torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests):

* **mxv (pull)**: rank r owns the row block [cuts[r], cuts[r+1]) of A and a full replica of u; after the local product the
  ranks exchange their w slices into the next u -- an all-gather with per-rank sizes (``allgatherv_into``; equal blocks take
  the single-collective fast path ``allgather_into``).  Slices are 64-row aligned, so values AND bit-packed presence words
  land directly in the replicated vector's HBM image, with no staging copy.
  When few entries change from one step to the next (the tail of an SSSP or BFS run), ``allgather_delta_into`` sends the changed
  (index, value) pairs and presence words only and falls back to the dense exchange above a changed fraction.
* **vxm (push) / mxv with T0**: rank r owns rows [cuts[r], cuts[r+1]) of A and the matching slice of u; its product is a
  PARTIAL result over all n columns; the partials are combined with the semiring's monoid by an all-reduce
  (``allreduce_monoid``: ncclMin / ncclMax / ncclSum on the values with absent entries set to the identity, bit-or on the
  presence words), then the write rule (mask, accumulator, replace) is applied to the replicated result.
* **mxm**: rows of A sharded, B replicated: no collective in the product (the result stays row-sharded).

Row cuts are balanced by the work they carry -- entries for mxv, flops (sum over the entries of A of the length of the B row
they select) for mxm -- not by row count: on a power-law graph equal row blocks differ by 2x and more in entries.
"""
from __future__ import annotations

import numpy as np


def row_block(n, rank, world):
    """Equal 64-row-aligned blocks (the single-collective all-gather needs them)."""
    if n % (64 * world):
        raise ValueError("n must be a multiple of 64 * world_size for zero-copy all-gather")
    rows = n // world
    return rank * rows, (rank + 1) * rows


def chunk_blocks(n, rank, world, chunks):
    """Row blocks of ``rank`` when every rank owns ``chunks`` blocks instead of one: block c = rows
    [c * n / chunks + rank * h, ... + h) with h = n / (chunks * world), 64-row aligned.  The slices of chunk c of all ranks are
    contiguous in u, in rank order: ONE all-gather per chunk lands them in place, and the exchange of chunk c can run while
    chunk c + 1 is being computed (``OverlappedMxv``)."""
    if n % (64 * world * chunks):
        raise ValueError("n must be a multiple of 64 * world_size * chunks")
    h = n // (world * chunks)
    return [(c * (n // chunks) + rank * h, c * (n // chunks) + (rank + 1) * h) for c in range(chunks)]


def balanced_cuts(weight_prefix, world, align=64):
    """Row cuts [c_0 = 0, c_1, ..., c_world = m] such that every block carries about the same weight.

    ``weight_prefix``: array of m+1 non-decreasing numbers, prefix sums of the per-row work (the CSR row pointer for
    entry-balanced mxv blocks; ``flops_prefix`` for mxm).  Cuts are multiples of ``align`` rows (64: one presence word), except
    the last one; a block may be empty when one row outweighs a whole share."""
    wp = np.asarray(weight_prefix)
    m = wp.size - 1
    total = float(wp[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(wp, target, side="left"))
        c = min(m, max(cuts[-1], int(round(c / align)) * align))
        cuts.append(c)
    cuts.append(m)
    return cuts


def flops_prefix(indptr_a, col_a, rowlen_b):
    """Prefix sums (m+1) of the per-row multiply counts of A @ B: sum over the entries (i, k) of row i of nnz(B(k, :))."""
    import torch

    ip = torch.as_tensor(indptr_a)
    per_entry = torch.as_tensor(rowlen_b)[torch.as_tensor(col_a).long()].to(torch.int64)
    csum = torch.zeros(per_entry.numel() + 1, dtype=torch.int64, device=per_entry.device)
    csum[1:] = torch.cumsum(per_entry, 0)
    return csum[ip.long()].cpu().numpy()


def allgather_into(u_full, w_local, *, device="cuda", values=True, presence=True):
    """u_full[lo_r:hi_r] = w_local of rank r, for every r -- equal blocks, one collective per array.

    Side effect, here and in every function of this module that hands a vector's HBM image to a collective (``allgatherv_into``,
    ``allgather_delta_into``, ``allreduce_monoid``, ``OverlappedMxv``): the vectors are PINNED to natural index order for good
    (``GrX_Vector_pin_natural``) -- the collective's buffers alias them, and a row-sharded run works on non-square blocks that never take
    a vertex order anyway.  A caller that goes on to use such a vector with a large square matrix on one GPU and wants its ordered
    layouts back releases it with ``device.vector_release_views(v)``."""
    import torch.distributed as dist

    from . import device as dev

    u_vals, u_words = dev.vector_device_views(u_full, device, pin=True)
    w_vals, w_words = dev.vector_device_views(w_local, device, pin=True)
    if values:
        _gather(dist, u_vals, w_vals)
    if presence:
        _gather(dist, u_words, w_words)
        dev.vector_modified(u_full)


def allgatherv_into(u_full, w_local, cuts, *, device="cuda", values=True, presence=True, async_op=False):
    """u_full[cuts[r]:cuts[r+1]] = w_local of rank r, for every r -- blocks of any (64-row aligned) sizes: rank r broadcasts its
    slice in place (an all-gather-v as grouped broadcasts: what RCCL does for uneven all-gathers as well).  With ``async_op``
    returns the work handles (wait on them before reading u_full); the caller must not hand u_full to a product meanwhile."""
    import torch.distributed as dist

    from . import device as dev

    rank, world = dist.get_rank(), dist.get_world_size()
    u_vals, u_words = dev.vector_device_views(u_full, device, pin=True)
    w_vals, w_words = dev.vector_device_views(w_local, device, pin=True)
    lo, hi = cuts[rank], cuts[rank + 1]
    if hi > lo:
        if values:
            u_vals[lo:hi].copy_(w_vals[: hi - lo])
        if presence:
            u_words[lo // 32: (hi + 31) // 32].copy_(w_words[: (hi - lo + 31) // 32])
    works = []
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        if b <= a:
            continue
        if values:
            works.append(dist.broadcast(u_vals[a:b], src=r, async_op=async_op))
        if presence:
            works.append(dist.broadcast(u_words[a // 32: (b + 31) // 32], src=r, async_op=async_op))
    if presence:
        dev.vector_modified(u_full)
    return [w for w in works if w is not None] if async_op else None


def allgather_delta_into(u_full, w_local, cuts, *, device="cuda", dense_above=0.125):
    """The same exchange as ``allgatherv_into`` when little changes between steps (late SSSP / BFS iterations): rank r compares
    its new slice w_local with what u_full still holds for its rows (the previous iterate), and the ranks exchange only the
    changed VALUES as (index, value) pairs and the changed PRESENCE WORDS as (word index, word) pairs -- three small collectives
    (counts, padded indices, padded payloads) instead of n / N values per rank.  When any rank has more than ``dense_above`` of its
    slice changed, every rank takes the dense exchange (the decision is made from the gathered counts, so it is the same
    everywhere).  Returns the number of changed values over all ranks (-1: the dense exchange ran)."""
    import torch
    import torch.distributed as dist

    from . import device as dev

    rank, world = dist.get_rank(), dist.get_world_size()
    u_vals, u_words = dev.vector_device_views(u_full, device, pin=True)
    w_vals, w_words = dev.vector_device_views(w_local, device, pin=True)
    lo, hi = cuts[rank], cuts[rank + 1]
    rows = hi - lo
    nw = (rows + 31) // 32
    as_bits = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[u_vals.element_size()]
    old_v, new_v = u_vals[lo:hi].view(as_bits), w_vals[:rows].view(as_bits)  # (bit patterns: NaN-safe, -0.0 != 0.0 is harmless)
    old_w, new_w = u_words[lo // 32: lo // 32 + nw], w_words[:nw]
    vi = torch.nonzero(old_v != new_v).flatten()
    wi = torch.nonzero(old_w != new_w).flatten()
    counts = torch.tensor([vi.numel(), wi.numel(), rows], dtype=torch.int64, device=u_vals.device)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = torch.stack(all_counts).cpu()
    if bool(((all_counts[:, 0].double() > dense_above * all_counts[:, 2].double().clamp(min=1)) & (all_counts[:, 0] > 0)).any()):
        allgatherv_into(u_full, w_local, cuts, device=device)
        return -1
    mv, mw = int(all_counts[:, 0].max()), int(all_counts[:, 1].max())
    if mv:
        send_i = torch.zeros(mv, dtype=torch.int64, device=u_vals.device)
        send_x = torch.zeros(mv, dtype=as_bits, device=u_vals.device)
        send_i[: vi.numel()] = vi + lo
        send_x[: vi.numel()] = new_v[vi]
        got_i = [torch.empty_like(send_i) for _ in range(world)]
        got_x = [torch.empty_like(send_x) for _ in range(world)]
        dist.all_gather(got_i, send_i)
        dist.all_gather(got_x, send_x)
        flat = u_vals.view(as_bits)
        for r in range(world):
            c = int(all_counts[r, 0])
            if c:
                flat[got_i[r][:c]] = got_x[r][:c]
    if mw:
        send_i = torch.zeros(mw, dtype=torch.int64, device=u_words.device)
        send_x = torch.zeros(mw, dtype=u_words.dtype, device=u_words.device)
        send_i[: wi.numel()] = wi + lo // 32
        send_x[: wi.numel()] = new_w[wi]
        got_i = [torch.empty_like(send_i) for _ in range(world)]
        got_x = [torch.empty_like(send_x) for _ in range(world)]
        dist.all_gather(got_i, send_i)
        dist.all_gather(got_x, send_x)
        for r in range(world):
            c = int(all_counts[r, 1])
            if c:
                u_words[got_i[r][:c]] = got_x[r][:c]
        dev.vector_modified(u_full)
    return int(all_counts[:, 0].sum())


def shard_setup(A_blocks, col_of_blocks, ncols, *, device="cuda", dist=None):
    """The set-up step of a row-sharded run on labels the application did NOT rank (round 6): every rank histograms the
    column indices of its blocks (``col_of_blocks``: their int32 column arrays as torch tensors), the histograms are summed over the ranks
    by ONE all-reduce, and every block matrix is handed the global counts (``GrX_Matrix_shard_setup``): the library ranks the columns --
    the same permutation on every rank -- and lays the block out in that order.  The blocks of one rank share one order object.
    ``dist`` = torch.distributed (None: a single process, e.g. one rank's block run alone).  Returns the counts."""
    import torch

    from . import device as dev

    dev_t = "cpu" if device == "cpu" else "cuda"
    counts = torch.zeros(ncols, dtype=torch.int64, device=dev_t)
    for col in col_of_blocks:
        counts += torch.bincount(col.to(dev_t).long(), minlength=ncols)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = counts.cpu() if (dev_t == "cuda" and dist.get_backend() == "gloo") else counts
        dist.all_reduce(t)
        counts = t.to(dev_t)
    c32 = counts.clamp(max=(1 << 31) - 1).to(torch.int32)
    first = None
    for A in A_blocks:
        if first is None:
            dev.matrix_shard_setup(A, c32)
            first = A
        else:
            dev.matrix_shard_setup(A, like=first)
    return counts


def cyclic_chunk_rows(n_local, B, chunks, c, device="cuda"):
    """LOCAL row numbers (in the rank's stripe order, ``synthetic.stripe_rows``) of chunk c of a block-cyclic rank: the stripes
    c, c + chunks, c + 2 chunks, ... of B rows each."""
    import torch

    j = torch.arange(n_local // chunks, device=device)
    return (torch.div(j, B, rounding_mode="floor") * chunks + c) * B + j % B


def _gather_async(dist, out, part):
    """all-gather `part` of every rank into `out` (rank order), asynchronously; returns the work handle."""
    try:
        return dist.all_gather_into_tensor(out, part, async_op=True)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        return dist.all_gather(list(out.chunk(dist.get_world_size())), part, async_op=True)


class OverlappedMxv:
    """The row-sharded pull step  w<mask> = accum(w, A (+.x) u)  followed by the exchange of the w slices into the next u, with
    the exchange OVERLAPPED with compute (reference call graphblas/core/matrix.py:2203-2262).

    Every rank owns ``chunks`` row blocks (``chunk_blocks``) with their own matrix, output and mask objects, and there are TWO
    replicas of u: step k reads replica k % 2 and the slices land in replica (k + 1) % 2, so nothing a product still reads is
    overwritten.  Per step: product of chunk 0 -> its all-gather is issued asynchronously (on the GPU: RCCL's own stream, ordered
    behind the product by an event) -> product of chunk 1 runs meanwhile -> ... -> wait for all exchanges.  With one chunk this
    is the plain step (product, then all-gather).  Values always travel; presence words only when ``presence`` (an output that
    is not full: the BFS step -- min_plus relaxations with a full u keep u and w full).

    ``cyclic=B`` (round 5): the rows are dealt to the ranks BLOCK-CYCLICALLY in stripes of B rows (global row (k * world + r) * B + i is
    local row k * B + i of rank r; B a multiple of 8) -- the sharding of a graph whose vertices are numbered by popularity
    (``GrX_Matrix_hint_ranked``: every rank's rows then carry the same mix of hubs and leaves, and the library builds its ordered layouts in
    those labels).  The chunks of a rank are interleaved the same way (its stripe k belongs to chunk k mod chunks: ``cyclic_chunk_rows``), so
    that every chunk carries the same mix too and the exchange of one hides behind a product as long as its own.  The all-gather of a chunk
    lands in a staging buffer (rank-major) and one strided copy per chunk deals the stripes into the replica the next step reads, after the
    collective: 2 x 4 bytes per row more HBM traffic than the contiguous sharding.

    The products go straight to the C ABI (``GrB_mxv`` with pre-resolved handles).  ``device="cpu"``: the gloo tests over the
    emulator build, where the vectors' images live in host memory."""

    def __init__(self, A_blocks, w_blocks, mask_blocks, u_pair, semiring, *, accum=None, desc_name=None, presence=False,
                 device="cuda", cyclic=0, exchange="dense", delta_below=0.03):
        import ctypes

        import torch.distributed as dist

        from . import _lib
        from . import device as dev

        self.dist, self.dev, self.device = dist, dev, device
        self.chunks = len(A_blocks)
        self.A, self.w, self.mask, self.u = list(A_blocks), list(w_blocks), list(mask_blocks), list(u_pair)
        self.presence = presence
        self.k = 0
        self.staged = False
        # (gloo moves CPU tensors only: device images are exchanged through host copies -- the rehearsal mode of bench.py on a box with
        #  fewer GPUs than ranks; RCCL takes the device images as they are)
        self.host_staged = device != "cpu" and dist.get_backend() == "gloo"
        n = self.u[0]._size
        world = dist.get_world_size()
        if n % (64 * world * self.chunks):
            raise ValueError("n must be a multiple of 64 * world_size * chunks")
        self.n, self.world, self.h = n, world, n // (world * self.chunks)
        self.cyclic = int(cyclic)
        if self.cyclic and (self.cyclic % 8 or self.h % self.cyclic):
            raise ValueError("cyclic: the stripe must be a multiple of 8 rows and divide n / (world_size * chunks)")
        self._stage, self._deal = {}, []
        # round 6: what travels.  "dense" = every chunk's slice, every step (the all-gather SURVEY 8e prescribes).  "auto": a step whose
        # PREDECESSOR changed less than `delta_below` of the rows (the same number on every rank: it comes out of the exchange itself)
        # sends (global row, value) pairs of the entries that differ from what the target replica holds, and the changed presence words --
        # under `~visited.S` only unvisited rows can change, and the tail of an SSSP / BFS run changes a few thousand rows of millions.
        # The first two steps are dense (both replicas must have been written once).
        if exchange not in ("dense", "auto", "delta"):
            raise ValueError("exchange: 'dense', 'auto' or 'delta'")
        self.exchange, self.delta_below = exchange, float(delta_below)
        self.last_changed = None       # rows changed by the last step over all ranks (None: unknown -- the dense exchange does not count)
        self.log = []                  # per step: (method, bytes this rank sent, rows changed over all ranks or -1)
        self._gidx = {}
        if device != "cpu":
            # The asynchronous collective is ordered behind the product (and the next product behind the collective) through torch's
            # CURRENT stream: ProcessGroupNCCL records its events there.  That orders against the library only if the library launches on
            # the same stream -- both default to the null stream; after GrX_set_stream, or under torch.cuda.stream(...), they differ and
            # a gather could read a slice before its product has written it.
            import torch

            lib_stream = ctypes.c_void_p()
            _lib.lib.GrX_get_stream(ctypes.byref(lib_stream))
            if (lib_stream.value or 0) != (torch.cuda.current_stream().cuda_stream or 0):
                raise RuntimeError("OverlappedMxv: the library's launch stream (GrX_set_stream) is not torch's current stream: the exchange "
                                   "would not be ordered behind the products; construct and step it with the two streams equal")
        desc = ctypes.c_void_p(_lib.handle(desc_name)) if desc_name else None
        self._call = _lib.lib.GrB_mxv
        self._args = [[(self.w[c]._carg, self.mask[c]._carg if self.mask[c] is not None else None, accum._carg if accum is not None else None,
                        semiring._carg, self.A[c]._carg, self.u[r]._carg, desc) for c in range(self.chunks)] for r in range(2)]
        self._views()

    def _views(self):
        as_u8 = lambda t: t.view(self._torch().uint8) if t.dtype == self._torch().bool else t
        self.u_vals, self.u_words, self.w_vals, self.w_words = [], [], [], []
        for r in range(2):
            v, b = self.dev.vector_device_views(self.u[r], self.device, pin=True)
            self.u_vals.append(as_u8(v))
            self.u_words.append(b)
        for c in range(self.chunks):
            v, b = self.dev.vector_device_views(self.w[c], self.device, pin=True)
            self.w_vals.append(as_u8(v))
            self.w_words.append(b)

    @staticmethod
    def _torch():
        import torch

        return torch

    def current_u(self):
        """The replica the NEXT step reads."""
        return self.u[self.k & 1]

    def _exchange_cyclic(self, c, dst):
        """all-gather of chunk c into its staging buffer; the strided copies that deal the stripes into replica ``dst`` are queued in
        self._deal and run once the collective has completed (step)."""
        torch = self._torch()
        B, h, world = self.cyclic, self.h, self.world
        k_c = h // B
        pairs = [("v", self.u_vals[dst], self.w_vals[c][:h], B)]
        if self.presence:
            pairs.append(("b", self.u_words[dst].view(torch.uint8), self.w_words[c][: h // 32].view(torch.uint8), B // 8))
        works = []
        for tag, out_full, part, b in pairs:
            host = self.host_staged
            send = part.cpu() if host else (part.clone() if self.staged else part)
            G = self._stage.get((tag, c))
            if G is None:
                G = self._stage[(tag, c)] = torch.empty(world * part.numel(), dtype=send.dtype, device=send.device)
            if host:
                _gather(self.dist, G, send)
            else:
                works.append(_gather_async(self.dist, G, send))
            # u[((j * chunks + c) * world + r) * b + i]  <-  G[r][j * b + i]   for the stripes j of this chunk
            self._deal.append((out_full.view(k_c, self.chunks, world, b)[:, c], G.view(world, k_c, b).permute(1, 0, 2)))
        return works

    def _exchange(self, c, dst):
        if self.cyclic:
            return self._exchange_cyclic(c, dst)
        n_c, h = self.n // self.chunks, self.h
        out_v = self.u_vals[dst][c * n_c: (c + 1) * n_c]
        part_v = self.w_vals[c][:h]
        works = []
        if self.host_staged:
            torch = self._torch()
            pairs = [(out_v, part_v)]
            if self.presence:
                pairs.append((self.u_words[dst][c * n_c // 32: (c + 1) * n_c // 32], self.w_words[c][: h // 32]))
            for out_t, part_t in pairs:
                send = part_t.cpu()
                recv = torch.empty(out_t.numel(), dtype=send.dtype)
                _gather(self.dist, recv, send)
                out_t.copy_(recv)
            return works
        if self.staged:
            torch = self._torch()
            send = part_v.clone()
            recv = torch.empty(n_c, dtype=send.dtype, device=send.device)
            self.dist.all_gather_into_tensor(recv, send)
            out_v.copy_(recv)
        else:
            works.append(_gather_async(self.dist, out_v, part_v))
        if self.presence:
            out_b = self.u_words[dst][c * n_c // 32: (c + 1) * n_c // 32]
            part_b = self.w_words[c][: h // 32]
            if self.staged:
                torch = self._torch()
                send = part_b.clone()
                recv = torch.empty(n_c // 32, dtype=send.dtype, device=send.device)
                self.dist.all_gather_into_tensor(recv, send)
                out_b.copy_(recv)
            else:
                works.append(_gather_async(self.dist, out_b, part_b))
        return works

    def probe_exchange(self):
        """One untimed exchange of what the vectors hold now, to find out whether the collective accepts the library's own device
        memory (stream-ordered HIP pool); should it refuse, the exchanges are staged through torch-owned tensors from then on."""
        try:
            for wk in self._exchange(0, self.k & 1 ^ 1):
                wk.wait()
        except RuntimeError:
            self.staged = True
            for wk in self._exchange(0, self.k & 1 ^ 1):
                wk.wait()
        self._deal = []  # (a probe: nothing is dealt into the replica)

    def _check_stream(self):
        """The exchange is ordered behind the products through torch's current stream (see __init__): checked at every step, not only at
        construction -- a caller may have entered torch.cuda.stream(...) or called GrX_set_stream in between.  One ctypes
        call and one attribute read."""
        if self.device == "cpu":
            return
        import ctypes

        from . import _lib

        lib_stream = ctypes.c_void_p()
        _lib.lib.GrX_get_stream(ctypes.byref(lib_stream))
        if (lib_stream.value or 0) != (self._torch().cuda.current_stream().cuda_stream or 0):
            raise RuntimeError("OverlappedMxv.step: the library's launch stream is not torch's current stream: the exchange would not be "
                               "ordered behind the products")

    def _global_rows(self, c):
        """int64 tensor: the global row of every local row of chunk c (contiguous blocks: a range; cyclic: the stripes' rows)."""
        g = self._gidx.get(c)
        if g is None:
            torch = self._torch()
            rank = self.dist.get_rank()
            dev_t = self.u_vals[0].device
            j = torch.arange(self.h, device=dev_t, dtype=torch.int64)
            if self.cyclic:
                B = self.cyclic
                g = ((torch.div(j, B, rounding_mode="floor") * self.chunks + c) * self.world + rank) * B + j % B
            else:
                g = c * (self.n // self.chunks) + rank * self.h + j
            self._gidx[c] = g
        return g

    def _exchange_delta(self, c, dst):
        """The entries of chunk c's slice that differ from what replica ``dst`` holds for those rows travel as (global row, bit pattern)
        pairs, the presence words that differ as (word index, word) pairs: three small collectives.  Returns (rows changed over all
        ranks, bytes this rank sent).  Synchronous (the sizes are read on the host); used where little changes, i.e. where the
        products are short as well."""
        torch = self._torch()
        dist, world, h = self.dist, self.world, self.h
        uv, wv = self.u_vals[dst], self.w_vals[c][:h]
        as_bits = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[uv.element_size()]
        g = self._global_rows(c)
        old_bits, new_bits = uv.view(as_bits)[g], wv.view(as_bits)
        vi = torch.nonzero(old_bits != new_bits).flatten()
        wi = None
        cnt = [int(vi.numel()), 0]
        if self.presence:
            # (64-row aligned slices: local presence word q covers local rows 32 q .. 32 q + 31; cyclic stripes are multiples of 8 rows, so
            #  presence travels as BYTES there: byte b of the local image = local rows 8 b .. 8 b + 7 = global byte g[8 b] / 8)
            ub = self.u_words[dst].view(torch.uint8)
            wb = self.w_words[c][: h // 32].view(torch.uint8)
            gb = torch.div(g[::8], 8, rounding_mode="floor")
            wi = torch.nonzero(ub[gb] != wb).flatten()
            cnt[1] = int(wi.numel())
        counts = torch.tensor(cnt, dtype=torch.int64)
        allc = [torch.empty_like(counts) for _ in range(world)]
        host_counts = counts if (self.device == "cpu" or self.host_staged) else counts.to(uv.device)
        allh = [torch.empty_like(host_counts) for _ in range(world)]
        dist.all_gather(allh, host_counts)
        allc = torch.stack([t.cpu() for t in allh])
        mv, mw = int(allc[:, 0].max()), int(allc[:, 1].max())
        sent = 16

        def gather_pairs(idx, payload, m, scatter):
            si = torch.zeros(m, dtype=torch.int64, device=idx.device)
            sx = torch.zeros(m, dtype=payload.dtype, device=payload.device)
            si[: idx.numel()] = idx
            sx[: idx.numel()] = payload
            if self.host_staged:
                si, sx = si.cpu(), sx.cpu()
            gi = [torch.empty_like(si) for _ in range(world)]
            gx = [torch.empty_like(sx) for _ in range(world)]
            dist.all_gather(gi, si)
            dist.all_gather(gx, sx)
            return gi, gx

        if mv:
            gi, gx = gather_pairs(g[vi], new_bits[vi], mv, None)
            flat = uv.view(as_bits)
            for r in range(world):
                k = int(allc[r, 0])
                if k:
                    flat[gi[r][:k].to(flat.device)] = gx[r][:k].to(flat.device)
            sent += cnt[0] * (8 + uv.element_size())
        if self.presence and mw:
            gi, gx = gather_pairs(gb[wi], wb[wi], mw, None)
            for r in range(world):
                k = int(allc[r, 1])
                if k:
                    ub[gi[r][:k].to(ub.device)] = gx[r][:k].to(ub.device)
            sent += cnt[1] * 9
        return int(allc[:, 0].sum()), sent

    def step(self):
        self._check_stream()
        src, dst = self.k & 1, (self.k + 1) & 1
        rows_all = self.n
        use_delta = self.exchange == "delta" or (self.exchange == "auto" and self.k >= 2 and self.last_changed is not None and
                                                 self.last_changed <= self.delta_below * rows_all)
        if self.k < 2 and self.exchange != "dense":
            use_delta = False  # (a replica that was never written holds nothing to take a difference from)
        works = []
        changed, sent = 0, 0
        for c in range(self.chunks):
            rc = self._call(*self._args[src][c])
            if rc != 0:
                raise RuntimeError(f"GrB_mxv failed with GrB_Info {rc}")
            if use_delta:
                ch, by = self._exchange_delta(c, dst)
                changed += ch
                sent += by
            else:
                works += self._exchange(c, dst)
                sent += self.h * self.u_vals[dst].element_size() + (self.h // 8 if self.presence else 0)
        for wk in works:
            wk.wait()
        for out_t, src_t in self._deal:
            out_t.copy_(src_t)
        self._deal = []
        if self.presence:
            self.dev.vector_modified(self.u[dst])
        if self.exchange != "dense" and not use_delta:
            # a dense step of an "auto" run still has to know how much changed -- the next step chooses from it: one count per rank
            # (the rows of this rank's slices that differ between the two replicas), summed by a tiny all-reduce
            torch = self._torch()
            diff = 0
            for c in range(self.chunks):
                g = self._global_rows(c)
                a_bits = self.u_vals[dst].view(self.u_vals[dst].dtype)[g]
                b_bits = self.u_vals[src].view(self.u_vals[src].dtype)[g]
                diff += int((a_bits != b_bits).sum().item()) if a_bits.dtype != torch.float32 and a_bits.dtype != torch.float64 else \
                    int((a_bits.view(torch.int32 if a_bits.element_size() == 4 else torch.int64) != b_bits.view(torch.int32 if b_bits.element_size() == 4 else torch.int64)).sum().item())
            t = torch.tensor([diff], dtype=torch.int64)
            if not (self.device == "cpu" or self.host_staged):
                t = t.to(self.u_vals[dst].device)
            self.dist.all_reduce(t)
            changed = int(t.item())
        self.last_changed = changed if self.exchange != "dense" else None
        self.log.append(("delta" if use_delta else "dense", sent, changed if self.exchange != "dense" else -1))
        self.k += 1


_REDUCE_OF = {"min": "MIN", "max": "MAX", "plus": "SUM", "lor": "MAX", "land": "MIN", "any": "MAX"}


def allreduce_monoid(t, monoid_name, identity=None, *, device="cuda"):
    """Combine the ranks' PARTIAL products t (same size on every rank) with the monoid, in place: the library writes the monoid's
    identity under the entries a rank does not have (GrX_Vector_fill_absent: one pass over the vector, no n-element temporaries; for
    ANY, which has no identity, the smallest value of the type, so that a rank without the entry never wins the MAX), the value image
    is all-reduced with the matching collective operator (ncclMin / ncclMax / ncclSum; lor = max and land = min on 0/1), the
    presence words with a bit-or.  Returns nothing; t then holds the full product on every rank.  (``identity`` is kept for old
    callers: the library derives it from the monoid.)"""
    import torch
    import torch.distributed as dist

    from . import _lib, operators
    from . import device as dev
    from .base import call_on

    if monoid_name not in _REDUCE_OF:
        raise NotImplementedError(f"no collective operator for the monoid {monoid_name!r} (times / lxor / lxnor need a gather + local fold)")
    mon = getattr(operators.monoid, monoid_name)[t.dtype]
    vals, words = dev.vector_device_views(t, device, pin=True)
    call_on(t, "GrX_Vector_fill_absent", [t._handle, mon._carg])
    as_num = vals.view(torch.uint8) if vals.dtype == torch.bool else vals
    dist.all_reduce(as_num, op=getattr(dist.ReduceOp, _REDUCE_OF[monoid_name]))
    # presence: the bit-or of the ranks' words.  RCCL / NCCL have no bitwise reduction (ProcessGroupNCCL refuses ReduceOp.BOR -- found
    # by the first run on RCCL, profiles/r04/rccl_one_rank.txt): the packed words of all ranks are gathered (n / 8 bytes per rank) and
    # folded locally
    world = dist.get_world_size()
    if dist.get_backend() == "nccl" or world == 1:
        gathered = torch.empty(world * words.numel(), dtype=words.dtype, device=words.device)
        _gather(dist, gathered, words.contiguous())
        acc = gathered[: words.numel()].clone()
        for r in range(1, world):
            acc |= gathered[r * words.numel(): (r + 1) * words.numel()]
        words.copy_(acc)
    else:
        dist.all_reduce(words, op=dist.ReduceOp.BOR)
    dev.vector_modified(t)


def sharded_vxm(gb, w, u_local, A_local, semiring, *, mask=None, accum=None, replace=False, device="cuda"):
    """w<mask, replace> = accum(w, u' A) with the ROWS of A (and u) sharded: ``A_local`` holds this rank's rows (all n columns),
    ``u_local`` the matching slice of u; w, mask are replicated (size n).  Local partial product -> monoid all-reduce -> the
    write rule on the replicated result (reference call graphblas/core/vector.py:1309-1378)."""
    from . import operators

    sr = semiring if isinstance(semiring, operators.TypedOp) else operators.get_typed_op(semiring, u_local.dtype, A_local.dtype, kind="semiring")
    n = A_local.ncols
    t = gb.Vector(sr.return_type, size=n)
    t << u_local.vxm(A_local, sr)
    mon = operators._canon(sr.parent.monoid.name, sr.type)
    ident = sr.parent.monoid.identity(sr.type)
    _ensure_dense_storage(gb, t)
    allreduce_monoid(t, mon, ident, device=device)
    # the write rule over the replicated product: first(t, t) = t through eWiseAdd's mask / accumulator / replace handling
    out = w(accum=accum) if mask is None else w(mask, accum=accum, replace=replace)
    out << t.ewise_add(t, gb.binary.first)
    return w


def _ensure_dense_storage(gb, t):
    """A vector that never received an entry has no storage yet: give it one (an entry set and removed)."""
    if t.nvals == 0:
        t[0] << t.dtype.np_type.type(0)
        del t[0]


def _gather(dist, out, part):
    try:
        dist.all_gather_into_tensor(out, part)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        parts = list(out.chunk(dist.get_world_size()))
        dist.all_gather(parts, part)
