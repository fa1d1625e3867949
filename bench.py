#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark (BASELINE.json metric: GTEPS of mxv on R-MAT scale-24).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S] [--workload NAME]

One "step" = one pass of the hot path over the resident graph: the masked SSSP relaxation
    w<~visited.S> = min(w, A min.+ u)          (GrB_mxv, GrB_MIN_PLUS_SEMIRING_FP32, accum GrB_MIN_FP32)
on the R-MAT scale-24 graph (north-star target line), inputs already resident in HBM.  With N > 1 ranks
(one process per GPU, launched by torch.distributed.run) the matrix is 1-D row-sharded, u is replicated
and every step ends with an RCCL all-gather of the w slices into the next u (strong scaling).

Prints ONE JSON line on rank 0 (see the driver contract); `roofline` is computed from the algorithmic
bytes of DESIGN.md / SURVEY.md section 8d and the HIP-event duration of the timed region; `cpu_baseline`
times the C oracle (OpenMP, all host cores) on the same workload -- it is the checker doubling as the
reported CPU baseline, never the product.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROFILE_ROUND = 5  # PMC traffic profiles of kernels from earlier rounds do not describe this library: only profiles/r05 (and later) count
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured copy


def measured_traffic(workload, scale):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN/pmc_traffic.json,
    collected by scripts/gpu_final.sh with rocprofv3 --pmc in separate runs); None when no profile matches."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic*.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and rec.get("scale") == scale and int(rec.get("round", 0)) >= PROFILE_ROUND:
            best = rec  # (directories sort by round: the newest matching profile wins)
    return best["traffic_bytes_per_launch"] if best else None


def measured_traffic_band(workload, scale):
    """[uncorrected, FETCH_SIZE-doubled] bytes per launch of the same profile: the corrected figure alone over-counts the gather share
    (the doubling is calibrated for 16-byte streaming reads only) -- the truth lies inside the band."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic*.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and rec.get("scale") == scale and int(rec.get("round", 0)) >= PROFILE_ROUND:
            best = rec
    if not best:
        return None
    return best.get("traffic_band_bytes") or [best.get("traffic_bytes_per_launch_uncorrected"), best.get("traffic_bytes_per_launch")]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scale", type=int, default=24)
    p.add_argument("--visited", type=float, default=0.5, help="fraction of rows masked out (visited)")
    p.add_argument("--workload", default="mxv_min_plus_masked",
                   choices=["mxv_min_plus_masked", "mxv_lor_land_masked", "mxv_min_plus", "mxm_plus_times", "mxm_plus_times_masked",
                            "mxm_plus_times_cmask", "bfs", "sssp",
                            "uniform_fp64", "kron26"])
    p.add_argument("--block", default=None, metavar="R/W",
                   help="single GPU: run rank R's row block of a W-way sharded run (the compute part of one rank's step; no exchange)")
    p.add_argument("--ranked", action="store_true",
                   help="the graph's vertices numbered by falling in-degree up front (the application relabels once), GrX_Matrix_hint_ranked on every "
                        "matrix; with --gpus N (or --block r/w: one rank's share, alone) the rows are dealt block-cyclically, --stripe rows at a time (round 5)")
    p.add_argument("--stripe", type=int, default=8, help="--ranked: rows per stripe of the block-cyclic row dealing (a multiple of 8)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-warm-build", "--no-page-in", dest="no_warm_build", action="store_true",
                   help="skip the second, warm layout build on another handle of the same graph (layout_build_warm_ms)")
    p.add_argument("--no-shard-setup", action="store_true",
                   help="row blocks (--block / --gpus N) on scrambled labels: skip the set-up step that hands the library the global column "
                        "counts (sharded.shard_setup); the blocks then run the natural-order layouts of rounds 1-5")
    p.add_argument("--streamed", action="store_true", help="mxm: run the unmasked product in row batches with the output streamed")
    p.add_argument("--stream-budget-gb", type=float, default=64.0, help="mxm --streamed: device bytes one batch's product may take")
    p.add_argument("--extra", action="store_true", help="(kept for old command lines: the secondary workloads now run by default)")
    p.add_argument("--no-extra", action="store_true", help="headline workload only: skip the lines reported under 'extra'")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="N > 1: torch.distributed backend (nccl = RCCL; gloo only to rehearse the N-rank code path on a box with fewer GPUs)")
    p.add_argument("--share-gpus", action="store_true",
                   help="N > 1 rehearsal: ranks beyond the visible GPUs share devices (rank r on GPU r %% device_count; needs --backend gloo: "
                        "RCCL refuses two ranks on one device); the line is marked as a rehearsal and is no measurement")
    p.add_argument("--force-dist", action="store_true",
                   help="with ONE rank: initialise torch.distributed all the same and take the N-rank code path (sharded.OverlappedMxv with two "
                        "replicas of u, all_gather_into_tensor on the library's memory, the sharded SpGEMM extra) -- a one-rank RCCL world "
                        "proves on a single-GPU box that RCCL accepts the library's buffers; the numbers are no multi-GPU measurement")
    p.add_argument("--overlap-chunks", type=int, default=2,
                   help="N > 1: row blocks per rank; the all-gather of block c overlaps the product of block c + 1 (1 = no overlap)")
    return p.parse_args()


def algorithmic_bytes_mxv(nnz_active, m, n, v_a, v_u, v_w, accum, mask):
    """pull mxv bytes = nnz*(I+V_A) + (m+1)*P + n*V_u + m*V_w*(1+[accum]) + m*Mb*[mask],
    I = 4 B column index, P = 8 B row pointer, Mb = 1/8 B (bit-packed mask), x counted once."""
    return nnz_active * (4 + v_a) + (m + 1) * 8 + n * v_u + m * v_w * (1 + int(accum)) + (m / 8.0) * int(mask)


class MxvWorkload:
    """Holds the HBM-resident operands of one masked mxv and launches it through the C ABI directly
    (ctypes call with pre-resolved handles: no per-step Python marshalling beyond one FFI call).

    With N > 1 ranks every rank owns ``chunks`` row blocks (sharded.chunk_blocks) and two replicas of u: a step is the product
    of every block, each followed by the asynchronous all-gather of its w slices into the replica the NEXT step reads
    (sharded.OverlappedMxv) -- the exchange of block c runs while block c + 1 is computed."""

    def __init__(self, gb, torch, scale, rank, world, semiring, visited_frac, seed=0, block=None, chunks=1, force_dist=False, weights="int255",
                 fresh_outputs=0, ranked=False, stripe=8, shard_setup=True):
        from graphblas_amd import _lib, device, sharded, synthetic

        self.gb, self.torch, self.rank, self.world = gb, torch, rank, world
        sharded_path = (world > 1 or force_dist) and not block  # (--force-dist: the N-rank code path with one rank)
        n = 1 << scale
        # (--block r/w: this process computes rank r's rows of a w-way run, alone: no exchange)
        if block:
            assert n % (64 * block[1]) == 0
            rows = n // block[1]
            ranges = [(block[0] * rows, (block[0] + 1) * rows)]
        elif sharded_path:
            ranges = sharded.chunk_blocks(n, rank, world, chunks)
        else:
            ranges = [(0, n)]
        # --ranked (round 5): the graph's vertices numbered by falling in-degree ONCE, up front (synthetic.rmat_csr(relabel="popularity")), its
        # rows dealt to the ranks block-cyclically in stripes of --stripe rows (a contiguous block of a ranked graph would hold all the hubs), every
        # matrix carrying GrX_Matrix_hint_ranked: the library then builds its popularity-ordered layouts in these labels -- for row blocks too
        self.ranked, self._gid = bool(ranked), None
        if ranked and sharded_path:
            # N ranks: rank r holds the stripes r, r + world, ... (n / world local rows) in `chunks` blocks of consecutive LOCAL rows; the
            # exchange deals the gathered slices back stripe by stripe (sharded.OverlappedMxv(cyclic=stripe))
            h = n // (world * chunks)
            if stripe % 8 or h % stripe or n % (64 * world * chunks):
                raise SystemExit("bench.py --ranked: --stripe must be a multiple of 8 rows and divide n / (ranks * chunks)")
            gid_all = synthetic.stripe_rows(n, stripe, world, rank)
            # (the rank's stripes are dealt to its chunks cyclically too: chunk c = local stripes c, c + chunks, ...; _gid is chunk-major)
            self._chunk_rows = [sharded.cyclic_chunk_rows(n // world, stripe, chunks, c) for c in range(chunks)]
            self._gid = torch.cat([gid_all[sel] for sel in self._chunk_rows])
            ranges = [(c * h, (c + 1) * h) for c in range(chunks)]
        if ranked and block:
            self._gid = synthetic.stripe_rows(n, stripe, block[1], block[0])
            ranges = [(0, n // block[1])]
        self.ranges, self.n, self.block = ranges, n, block
        self.lo, self.hi = ranges[0]
        self.m = sum(hi - lo for lo, hi in ranges)
        if ranked:
            deal = (stripe, block[1], block[0]) if block else ((stripe, world, rank) if sharded_path else None)
            ip_l, col_l = synthetic.rmat_csr(scale, device="cuda", relabel="popularity", stripes=deal)
            graphs = [(ip_l, col_l)]
            if len(ranges) > 1:  # (the rank's local rows dealt to its chunks: a row selection of the local CSR)
                graphs = []
                for sel in self._chunk_rows:
                    lens = ip_l[sel + 1] - ip_l[sel]
                    ip_c = torch.zeros(sel.numel() + 1, dtype=torch.int64, device="cuda")
                    ip_c[1:] = torch.cumsum(lens, 0)
                    src = torch.repeat_interleave(ip_l[sel] - ip_c[:-1], lens) + torch.arange(int(ip_c[-1].item()), device="cuda")
                    graphs.append((ip_c, col_l[src].contiguous()))
                    del lens, src
            del ip_l, col_l
        else:
            graphs = synthetic.rmat_csr(scale, device="cuda", row_ranges=ranges) if (block or sharded_path) else [synthetic.rmat_csr(scale, device="cuda")]
        gen = torch.Generator(device="cuda")
        gen.manual_seed(4242 + seed)
        visited = torch.rand(n, generator=gen, device="cuda") < visited_frac
        self.semiring = semiring
        n_u = 2 if sharded_path else 1
        if semiring == "min_plus":
            dist = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
            self._dist = dist
            self.us = [device.vector_from_device(dist) for _ in range(n_u)]
            self.sr = gb.semiring.min_plus["FP32"]
            self.accum = gb.binary.min["FP32"]
            self.dtype_name, self.v_a, self.v_u, self.v_w = "f32", 4, 4, 4
        else:  # lor_land on an iso-True BOOL matrix: one BFS level step  q<~visited.S, replace> = A lor.land q
            frontier = torch.rand(n, generator=gen, device="cuda") < 0.3
            self._frontier = frontier
            self.us = [device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"), present=frontier) for _ in range(n_u)]
            self.sr = gb.semiring.lor_land["BOOL"]
            self.accum = None
            self.dtype_name, self.v_a, self.v_u, self.v_w = "bool", 0, 1, 1
        self.u = self.us[0]
        self.As, self.ws, self.masks, self.visited_c, self._keeps, self._valss = [], [], [], [], [], []
        self.nnz_local = self.nnz_active_local = 0
        self._rows_of = lambda X, lo, hi: (X[self._gid[lo:hi]] if self._gid is not None else X[lo:hi])  # (this rank's rows lo..hi-1 of a vector over all vertices)
        for (lo, hi), (indptr, col) in zip(ranges, graphs):
            rows = hi - lo
            vis_c = self._rows_of(visited, lo, hi).contiguous()
            rowlen = indptr[1:] - indptr[:-1]
            self.nnz_local += int(col.numel())
            self.nnz_active_local += int(rowlen[~vis_c].sum().item())
            if semiring == "min_plus":
                # (a sharded run never materialises the whole graph: every block draws its weights from the same stream, so the
                #  N-rank runs relax other weights than the single-GPU run; each run is checked against ITS OWN operands)
                vals = synthetic.edge_weights(col, scale) if weights == "int255" else synthetic.edge_weights_real(col, scale)
                self.As.append(device.matrix_from_device_csr(indptr, col, vals, rows, n, "FP32"))
                self.ws.append(device.vector_from_device(self._rows_of(self._dist, lo, hi).contiguous()))
                self._valss.append(vals)
            else:
                one = torch.ones(1, dtype=torch.bool, device="cuda")
                self.As.append(device.matrix_from_device_csr(indptr, col, one, rows, n, "BOOL", iso=True))
                self.ws.append(device.vector_from_device(torch.ones(rows, dtype=torch.bool, device="cuda"),
                                                         present=self._rows_of(self._frontier, lo, hi).contiguous()))
                self._valss.append(None)
            if ranked:
                device.matrix_hint_ranked(self.As[-1])
            self.masks.append(device.vector_from_device(torch.ones(rows, dtype=torch.bool, device="cuda"), present=vis_c))
            self.visited_c.append(vis_c)
            self._keeps.append((indptr, col))
        # round 6: row blocks on labels the application did not rank -- the ranks all-reduce their column histograms ONCE and hand every block
        # the global counts (sharded.shard_setup -> GrX_Matrix_shard_setup): the library ranks the columns, the same way on every rank, and
        # lays the block out in that order.  A process that holds one block only (--block) draws the counts from the same edge stream.
        self.shard_setup = False
        if shard_setup and not ranked and (block or sharded_path):
            if block:
                counts = synthetic.rmat_col_counts(scale, device="cuda").clamp(max=(1 << 31) - 1).to(torch.int32)
                for k, A in enumerate(self.As):
                    device.matrix_shard_setup(A, counts) if k == 0 else device.matrix_shard_setup(A, like=self.As[0])
                del counts
            else:
                import torch.distributed as tdist

                sharded.shard_setup(self.As, [k[1] for k in self._keeps], n, dist=tdist if tdist.is_initialized() else None)
            self.shard_setup = True
        # (single block: the names the CPU baseline and the single-GPU check use)
        self.A, self.w, self.mask, self.visited_local, self._keep, self._vals = (self.As[0], self.ws[0], self.masks[0], self.visited_c[0],
                                                                                 self._keeps[0], self._valss[0])
        desc_name = "GrB_DESC_SC" if semiring == "min_plus" else "GrB_DESC_RSC"
        L = _lib.lib
        self._call = L.GrB_mxv
        self._args = (self.w._carg, self.mask._carg, self.accum._carg if self.accum else None, self.sr._carg,
                      self.A._carg, self.u._carg, ctypes.c_void_p(_lib.handle(desc_name)))
        # "honest operands" variant: every timed step gets an output of its own holding the initial distances and the
        # operand alternates between two vectors, so no step runs at the relaxation's fixed point and the write rule stores every time
        self.fresh_ws, self.alt_us, self._dist2, self._k = [], [], None, 0
        if fresh_outputs and not sharded_path and not block and semiring == "min_plus":
            self._dist2 = torch.randint(0, 1000, (n,), generator=gen, device="cuda").to(torch.float32)
            self.alt_us = [self.u, device.vector_from_device(self._dist2)]
            self.fresh_ws = [device.vector_from_device(self._dist) for _ in range(fresh_outputs)]
            self._none = device.vector_from_device(torch.ones(n, dtype=torch.bool, device="cuda"))  # (a full mask: its complement admits no row)
        self.ov = None
        if n_u == 2:
            # the BFS step's output is not full: its presence words travel with the values
            self.ov = sharded.OverlappedMxv(self.As, self.ws, self.masks, self.us, self.sr, accum=self.accum, desc_name=desc_name,
                                            presence=(semiring != "min_plus"), cyclic=stripe if ranked else 0)
            self.ov.probe_exchange()
            torch.cuda.synchronize()

    def prepare_fresh(self):
        """Bring every output / operand of the honest-operands variant into the matrix's vertex order WITHOUT changing it: a product under a
        mask that admits no row converts the vectors it touches and stores nothing.  (Called after the layouts exist.)"""
        for v in self.fresh_ws:
            for uu in self.alt_us:
                rc = self._call(v._carg, self._none._carg, self.accum._carg, self.sr._carg, self.A._carg, uu._carg, self._args[6])
                assert rc == 0, rc
        self._k = 0

    def step(self):
        if self.ov is not None:
            return self.ov.step()
        if self.fresh_ws:
            k = self._k
            self._k += 1
            rc = self._call(self.fresh_ws[k % len(self.fresh_ws)]._carg, self.mask._carg, self.accum._carg, self.sr._carg, self.A._carg,
                            self.alt_us[k & 1]._carg, self._args[6])
            if rc != 0:
                raise RuntimeError(f"GrB_mxv failed with GrB_Info {rc}")
            return
        rc = self._call(*self._args)
        if rc != 0:
            raise RuntimeError(f"GrB_mxv failed with GrB_Info {rc}")

    # ---- checks --------------------------------------------------------------------------------------------------------------
    def _bits(self, words, m):
        import numpy as np

        return self.torch.from_numpy(np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")[:m].astype(bool)).cuda()

    def _expected_block(self, c, u_vals, u_has, w_vals_before, w_has_before):
        """torch evaluation of one step on block c from the given operand images: (expected presence, expected values)."""
        torch = self.torch
        indptr, col = self._keeps[c]
        lo, hi = self.ranges[c]
        m = hi - lo
        rows = torch.repeat_interleave(torch.arange(m, device="cuda"), indptr[1:] - indptr[:-1])
        active = ~self.visited_c[c]
        if self.semiring == "min_plus":
            cand = self._valss[c] + u_vals[col.long()]
            ref = torch.full((m,), float("inf"), device="cuda").scatter_reduce(0, rows, cand, "amin")
            del cand
            return w_has_before, torch.where(active, torch.minimum(w_vals_before, ref), w_vals_before)
        hit = torch.zeros(m, dtype=torch.bool, device="cuda")
        sel = u_has[col.long()] & u_vals[col.long()]
        hit.index_put_((rows[sel],), torch.tensor(True, device="cuda"))
        return hit & active, None

    def verify(self):
        """Check the output of the timed calls against a torch segment reduction of the same products.  Single GPU: u is fixed,
        so K steps of  w<mask> = min(w, A min.+ u)  leave what one step leaves; the BFS step has no accumulator at all.  N ranks:
        u changes with every step, so ONE MORE step is run from a snapshot of the operands and every rank checks (a) its blocks of
        that step against torch, (b) that the replica the next step reads holds exactly the ranks' slices (its own rows equal its
        w, and a checksum of the whole replica is the same on every rank); the flags are combined with a MIN all-reduce.
        min_plus over integer-valued fp32 operands is exact, so every comparison is bit for bit."""
        from graphblas_amd import device

        torch = self.torch
        if self.fresh_ws:
            # every timed step k relaxed ITS OWN copy of the initial distances with operand k & 1: check the last two (one per operand)
            ok = self._k >= 2
            for k in (self._k - 2, self._k - 1):
                if k < 0:
                    continue
                wv, wb = device.vector_device_views(self.fresh_ws[k % len(self.fresh_ws)], pin=False)
                uvals = self._dist if (k & 1) == 0 else self._dist2
                d0 = self._rows_of(self._dist, self.lo, self.hi)
                _h, exp = self._expected_block(0, uvals, None, d0, torch.ones(self.m, dtype=torch.bool, device="cuda"))
                ok = ok and bool(self._bits(wb, self.m).all().item()) and bool(torch.equal(wv, exp))
            return ok
        if self.ov is None:
            wv, wb = device.vector_device_views(self.w, pin=False)  # (transient views: fetched again after every call)
            if self.semiring == "min_plus":
                # (after the warm-up w is at the fixed point of the relaxation: a call that computed nothing would leave it right.
                #  Reset w to its initial values and run the call once more, on the layouts the timed calls ran on.)
                wv.copy_(self._rows_of(self._dist, self.lo, self.hi))
                self.step()
                torch.cuda.synchronize()
                # (the call may keep w in the matrix's vertex order between calls: the views are fetched again -- that brings it back)
                wv, wb = device.vector_device_views(self.w, pin=False)
            got_has = self._bits(wb, self.m)
            if self.semiring == "min_plus":
                d0 = self._rows_of(self._dist, self.lo, self.hi)
                exp_has, exp = self._expected_block(0, self._dist, None, d0, torch.ones(self.m, dtype=torch.bool, device="cuda"))
                return bool(got_has.all().item()) and bool(torch.equal(wv, exp))
            exp_has, _ = self._expected_block(0, torch.ones(self.n, dtype=torch.bool, device="cuda"), self._frontier, None, None)
            return bool(torch.equal(got_has, exp_has)) and bool(wv[exp_has].all().item())
        import torch.distributed as dist

        ov = self.ov
        src = ov.current_u()
        uv, ub = device.vector_device_views(src)
        if self.semiring == "min_plus":
            # (the relaxation has converged during the timed steps: start the checked step from the initial operands again)
            uv.copy_(self._dist)
            for c in range(ov.chunks):
                lo, hi = self.ranges[c]
                device.vector_device_views(self.ws[c])[0].copy_(self._rows_of(self._dist, lo, hi))
        u_vals, u_has = uv.clone(), self._bits(ub, self.n)
        before = []
        for c in range(ov.chunks):
            wv, wb = device.vector_device_views(self.ws[c])
            before.append((wv.clone(), self._bits(wb, self.ranges[c][1] - self.ranges[c][0])))
        ov.step()
        torch.cuda.synchronize()
        ok = True
        nv, nb = device.vector_device_views(ov.current_u())
        n_has = self._bits(nb, self.n)
        for c in range(ov.chunks):
            lo, hi = self.ranges[c]
            wv, wb = device.vector_device_views(self.ws[c])
            got_has = self._bits(wb, hi - lo)
            exp_has, exp = self._expected_block(c, u_vals, u_has, *before[c])
            if self.semiring == "min_plus":
                ok = ok and bool(got_has.all().item()) and bool(torch.equal(wv, exp)) and bool(torch.equal(self._rows_of(nv, lo, hi), wv))
            else:
                ok = ok and bool(torch.equal(got_has, exp_has)) and bool(wv[exp_has].all().item())
                ok = ok and bool(torch.equal(self._rows_of(n_has, lo, hi), got_has)) and bool(self._rows_of(nv, lo, hi)[got_has].all().item())
        # the replica every rank will read next must be the same everywhere: two order-independent checksums
        img = nv.view(torch.int32).to(torch.int64) if self.semiring == "min_plus" else (nv.to(torch.int64) & n_has.to(torch.int64))
        idx = torch.arange(1, self.n + 1, device="cuda", dtype=torch.int64)
        sums = torch.stack([img.sum(), (img * idx).sum(), n_has.to(torch.int64).sum()])
        hi_, lo_ = sums.clone(), sums.clone()
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        ok = ok and bool(torch.equal(hi_, lo_))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def bytes_per_step(self):
        return algorithmic_bytes_mxv(self.nnz_active_local, self.m, self.n, self.v_a, self.v_u, self.v_w,
                                     accum=self.accum is not None, mask=True)


def probe_suitesparse():
    """SURVEY.md section 8c / BASELINE.md 2.1: look for SuiteSparse:GraphBLAS on this machine (the library the reference binds,
    graphblas/__init__.py:143) before falling back to the repo's CPU restatement.  Returns (ctypes library | None, note)."""
    import ctypes.util

    try:
        import suitesparse_graphblas  # noqa: F401  (the reference's own binding; ships libgraphblas)

        return None, "python package suitesparse_graphblas importable (drive it through python-graphblas itself)"
    except Exception:
        pass
    name = ctypes.util.find_library("graphblas")
    if not name:
        return None, "not found: neither `import suitesparse_graphblas` nor ctypes.util.find_library('graphblas')"
    try:
        return ctypes.CDLL(name), f"found {name}"
    except OSError as e:
        return None, f"found {name} but it does not load: {e}"


def cpu_baseline_suitesparse(lib, wl, torch, reps=3):
    """The same masked min_plus mxv through SuiteSparse's C API on the host cores (only when probe_suitesparse() finds it)."""
    import numpy as np

    from graphblas_amd import device, synthetic

    P, c_u64, vp = ctypes.POINTER, ctypes.c_uint64, ctypes.c_void_p
    if lib.GrB_init(0) != 0:  # GrB_NONBLOCKING
        return None
    ip, cj = (t.cpu().numpy() for t in wl._keep)
    m, n = wl.m, wl.n
    rows = np.repeat(np.arange(m, dtype=np.uint64), np.diff(ip))
    cols = cj.astype(np.uint64)
    vals = synthetic.edge_weights(wl._keep[1], int(round(np.log2(n)))).cpu().numpy()
    h = lambda sym: vp.in_dll(lib, sym)
    A, u, w, mk = vp(), vp(), vp(), vp()
    ok = lib.GrB_Matrix_new(ctypes.byref(A), h("GrB_FP32"), c_u64(m), c_u64(n)) == 0
    ok &= lib.GrB_Matrix_build_FP32(A, rows.ctypes.data_as(vp), cols.ctypes.data_as(vp), vals.ctypes.data_as(vp), c_u64(rows.size), h("GrB_PLUS_FP32")) == 0
    uvals, _ = device.vector_device_views(wl.u, pin=False)
    uv = uvals.cpu().numpy()
    idx = np.arange(n, dtype=np.uint64)
    ok &= lib.GrB_Vector_new(ctypes.byref(u), h("GrB_FP32"), c_u64(n)) == 0
    ok &= lib.GrB_Vector_build_FP32(u, idx.ctypes.data_as(vp), uv.ctypes.data_as(vp), c_u64(n), h("GrB_PLUS_FP32")) == 0
    ok &= lib.GrB_Vector_new(ctypes.byref(w), h("GrB_FP32"), c_u64(m)) == 0
    ok &= lib.GrB_Vector_build_FP32(w, idx[:m].ctypes.data_as(vp), uv[wl.lo:wl.hi].copy().ctypes.data_as(vp), c_u64(m), h("GrB_PLUS_FP32")) == 0
    vis = np.flatnonzero(wl.visited_local.cpu().numpy()).astype(np.uint64)
    ones = np.ones(vis.size, np.bool_)
    ok &= lib.GrB_Vector_new(ctypes.byref(mk), h("GrB_BOOL"), c_u64(m)) == 0
    ok &= lib.GrB_Vector_build_BOOL(mk, vis.ctypes.data_as(vp), ones.ctypes.data_as(vp), c_u64(vis.size), h("GrB_LOR")) == 0
    if not ok:
        return None
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = lib.GrB_mxv(w, mk, h("GrB_MIN_FP32"), h("GrB_MIN_PLUS_SEMIRING_FP32"), A, u, h("GrB_DESC_SC"))
        lib.GrB_Vector_wait(w, 1)
        times.append(time.perf_counter() - t0)
        if rc != 0:
            return None
    best = float(np.median(times))
    return {"value": wl.nnz_active_local / best / 1e9, "unit": "GTEPS", "cores": os.cpu_count(), "kind": "reference",
            "sample": f"SuiteSparse:GraphBLAS C API on the host, same graph and operands, median of {reps} ({best * 1e3:.1f} ms)"}


def cpu_baseline_mxv(wl, torch, reps_budget_s=20.0):
    """Time the C oracle (OpenMP) on the same single-GPU workload: full graph, a few repetitions."""
    import numpy as np

    from graphblas_amd import device
    from oracle import grb_oracle as O

    ip, cj = wl._keep
    ip = ip.cpu().numpy()
    cj64 = cj.cpu().numpy().astype(np.int64)
    m, n = wl.m, wl.n
    row_active = (~wl.visited_local).cpu().numpy().astype(np.uint8)
    L = O.lib()
    O.use_all_threads()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    if wl.semiring == "min_plus":
        from graphblas_amd import synthetic

        vals = synthetic.edge_weights(cj, int(round(np.log2(n)))).cpu().numpy()
        uvals, _ = device.vector_device_views(wl.u, pin=False)
        u_val = uvals.cpu().numpy()
        u_has = None  # (the operand is full: the oracle skips the presence lookups, as the GPU kernels do)
        tcode, mon, mul, acc = O.TYPE_CODES["FP32"], O.OP_CODES["min"], O.OP_CODES["plus"], O.OP_CODES["min"]
        w_val = u_val[:m].copy()
        a_iso, replace = 0, 0
    else:
        vals = np.ones(1, np.uint8)
        uvals, uwords = device.vector_device_views(wl.u, pin=False)
        u_val = uvals.cpu().numpy().astype(np.uint8)
        u_has = np.unpackbits(uwords.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(np.uint8)
        tcode, mon, mul, acc = O.TYPE_CODES["BOOL"], O.OP_CODES["lor"], O.OP_CODES["land"], -1
        w_val = np.zeros(m, np.uint8)
        a_iso, replace = 1, 1
    w_has = np.ones(m, np.uint8)
    t_has = np.zeros(m, np.uint8)
    t_val = np.zeros(m, w_val.dtype)
    mask_true = wl.visited_local.cpu().numpy().astype(np.uint8)
    times = []
    t_begin = time.perf_counter()
    while len(times) < 3 or (len(times) < 10 and time.perf_counter() - t_begin < reps_budget_s):
        t0 = time.perf_counter()
        L.grbo_mxv(tcode, mon, mul, ctypes.c_int64(m), p(ip), p(cj64), p(vals), a_iso, p(u_has), p(u_val), p(row_active),
                   p(t_has), p(t_val))
        L.grbo_vec_write(tcode, ctypes.c_int64(m), p(w_has), p(w_val), p(t_has), p(t_val), p(mask_true), 1, acc, replace)
        times.append(time.perf_counter() - t0)
    best = float(np.median(times))
    logical, quota = O.cpu_budget()
    return {"value": wl.nnz_active_local / best / 1e9, "unit": "GTEPS", "cores": O.num_threads(), "kind": "port",
            "sample": f"same graph and operands, full pass, median of {len(times)} reps ({best * 1e3:.1f} ms each); "
                      "C oracle (oracle/grb_oracle.c, OpenMP) -- a CPU restatement, not SuiteSparse; "
                      f"host: {logical} logical CPUs visible, " + (f"cgroup quota {quota:.0f} CPUs of time -- the team is 2 threads per granted CPU "
                      "(more are throttled, not faster: profiles/r06/cpu_host.txt)" if quota else "no CPU quota")}


def _desc_s():
    from graphblas_amd import _lib
    return _lib.handle("GrB_DESC_S")


def verify_mxm(torch, device, gb, L, sr, A, B, ip_a, col_a, ip_b, col_b, n, masked, C, st, stream_out, budget, seed=5):
    """Independent checks of the SpGEMM line (operands are matrices of ones, INT64 plus_times):
      * the multiply count the symbolic pass reports = sum over the entries (i, k) of A of nnz(B(k, :)), computed by torch from the
        row pointers;
      * unmasked: sum of the values of C = that count (every multiply adds 1 somewhere), nnz(C) = the last row pointer, and, on
        a seeded sample of rows, the row's columns and values equal torch.unique(with counts) over the gathered rows of B
        (sorted, so the order inside the row is checked too); masked: the same rows restricted to the mask row's columns;
      * streamed (no C leaves the library): its checksum = the multiply count, and the same pipeline on a small row block equals
        the materialised product of that block, which is checked as above."""
    rowlen_b = ip_b[1:] - ip_b[:-1]
    flops_expected = int(rowlen_b[col_a.long()].sum().item())
    ok = int(st["flops"]) == flops_expected
    m = int(ip_a.numel()) - 1

    def check_rows(Cmat, ip_x, col_x, rows_total, mask_rows):
        cp, cj, cx, _iso = device.matrix_device_views(Cmat)
        good = int(cp[-1].item()) == int(cj.numel())
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        lens = (ip_x[1:] - ip_x[:-1]).cpu()
        cand = torch.nonzero(lens > 0).flatten()
        pick = cand[torch.randperm(cand.numel(), generator=g)[:24]].tolist() if cand.numel() else []
        for i in pick:
            ks = col_x[int(ip_x[i]): int(ip_x[i + 1])].long()
            starts, lens_k = ip_b[ks], ip_b[ks + 1] - ip_b[ks]
            offs = torch.cumsum(lens_k, 0) - lens_k
            total = int(lens_k.sum().item())
            pos = torch.arange(total, device="cuda") + (starts - offs).repeat_interleave(lens_k)  # every entry of every selected row of B
            cols, cnt = torch.unique(col_b[pos], return_counts=True)
            if mask_rows:
                keep = torch.isin(cols, ks.to(cols.dtype))
                if mask_rows == "comp":
                    keep = ~keep
                cols, cnt = cols[keep], cnt[keep]
            r0, r1 = int(cp[i].item()), int(cp[i + 1].item())
            good = good and r1 - r0 == int(cols.numel()) and bool(torch.equal(cj[r0:r1], cols.to(torch.int32))) \
                and bool(torch.equal(cx[r0:r1], cnt.to(torch.int64)))
        return good, cx

    if C is not None:
        good, cx = check_rows(C, ip_a, col_a, m, masked)
        # (stats.out_nvals counts the product the kernels wrote: C itself, except when a complemented mask was left to the write rule)
        ok = ok and good and (int(st["out_nvals"]) == int(cx.numel()) or (masked == "comp" and int(st["method"]) == 3))
        if not masked:
            ok = ok and int(cx.sum().item()) == flops_expected
        elif masked == "comp":
            # C<!A.S> and C<A.S> split the full product: their value sums add up to the multiply count (the mask-driven product
            # is another code path -- k_spgemm_mhash / k_spgemm_mwin / masked units -- and has its own checks above when it is the line)
            total_c = int(cx.sum().item())
            Cm = gb.Matrix("INT64", m, n)
            if L.GrB_mxm(Cm._carg, A._carg, None, sr._carg, A._carg, B._carg, ctypes.c_void_p(_desc_s())) == 0 and Cm.nvals > 0:
                good_m, cxm = check_rows(Cm, ip_a, col_a, m, True)
                ok = ok and good_m and total_c + int(cxm.sum().item()) == flops_expected
            else:
                ok = False
    else:
        ok = ok and stream_out.get("checksum") == flops_expected
        # the streamed pipeline on a row block against the materialised (and independently checked) product of the same block
        rows_blk = min(m, 2048)
        e1 = int(ip_a[rows_blk].item())
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        ip_k, col_k = ip_a[: rows_blk + 1].contiguous(), col_a[:e1].contiguous()
        Ablk = device.matrix_from_device_csr(ip_k, col_k, one, rows_blk, n, "INT64", iso=True)
        Cblk = gb.Matrix("INT64", rows_blk, n)
        ok = ok and L.GrB_mxm(Cblk._carg, None, None, sr._carg, Ablk._carg, B._carg, None) == 0
        good, cx = check_rows(Cblk, ip_k, col_k, rows_blk, False)
        nv, cs, fl, nb = (ctypes.c_uint64(0) for _ in range(4))
        ok = ok and good and L.GrX_mxm_streamed(sr._carg, Ablk._carg, B._carg, ctypes.c_uint64(max(budget >> 8, 1 << 26)), ctypes.byref(nv),
                                                ctypes.byref(cs), ctypes.byref(fl), ctypes.byref(nb)) == 0
        ok = ok and int(nv.value) == int(cx.numel()) and int(cs.value) == int(cx.sum().item()) == int(fl.value)
    return bool(ok)


def run_mxm(args, gb, torch, device, rank, world, dist, barrier, *, scale, workload, steps, warmup, want_cpu=True):
    """configs[3]: C = A (+.x) A, plus_times INT64 on R-MAT (ones), rows of A sharded over the ranks, B = A
    replicated; no collective in the timed loop (the product stays row-sharded).  value = nnz(C) / s.  Returns the line."""
    from graphblas_amd import _lib, sharded, synthetic

    n = 1 << scale
    ip_b, col_b = synthetic.rmat_csr(scale, device="cuda")
    # rows of A cut so that every rank carries the same number of multiplies (not the same number of rows): no collective
    # constrains the block sizes here
    sharded_path = world > 1 or getattr(args, "force_dist", False)  # (--force-dist: the flop-balanced cuts and the row view of A with one rank)
    blk = tuple(int(x) for x in args.block.split("/")) if (getattr(args, "block", None) and world == 1) else None
    if blk:  # (--block r/w: rank r's flop-balanced share of a w-way run, alone on this GPU: what one rank of the sharded product computes)
        sharded_path = True
        cuts = sharded.balanced_cuts(sharded.flops_prefix(ip_b, col_b, ip_b[1:] - ip_b[:-1]), blk[1])
        lo, hi = cuts[blk[0]], cuts[blk[0] + 1]
    elif sharded_path:
        cuts = sharded.balanced_cuts(sharded.flops_prefix(ip_b, col_b, ip_b[1:] - ip_b[:-1]), world)
        lo, hi = cuts[rank], cuts[rank + 1]
    else:
        lo, hi = 0, n
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    B = device.matrix_from_device_csr(ip_b, col_b, one, n, n, "INT64", iso=True)
    if sharded_path:
        ip_a = (ip_b[lo: hi + 1] - ip_b[lo]).contiguous()
        col_a = col_b[ip_b[lo]: ip_b[hi]].contiguous()
        A = device.matrix_from_device_csr(ip_a, col_a, one, hi - lo, n, "INT64", iso=True)
    else:
        A, ip_a, col_a = B, ip_b, col_b
    sr = gb.semiring.plus_times["INT64"]
    L = _lib.lib

    # triangle-count style: C<A.S> = A (+.x) A;  "comp": the two-hop pairs that are not neighbours, C<!A.S> = A (+.x) A
    masked = {"mxm_plus_times_masked": True, "mxm_plus_times_cmask": "comp"}.get(workload, False)
    desc_s = ctypes.c_void_p(_lib.handle("GrB_DESC_SC" if masked == "comp" else "GrB_DESC_S"))
    # the unmasked product of scale >= 21 does not fit one GPU (scale 22: 900 GB): it runs in row batches whose products fit
    # `budget`, every batch through the full symbolic + numeric pipeline; count and checksum leave the batch (GrX_mxm_streamed)
    streamed = (not masked) and (args.streamed or scale >= 21)  # (the same pipeline at every rank count: comparable lines)
    budget = int(args.stream_budget_gb * (1 << 30))
    stream_out = {}
    keep = {}

    def step():
        if streamed:
            nv, cs, fl, nb = (ctypes.c_uint64(0) for _ in range(4))
            rc = L.GrX_mxm_streamed(sr._carg, A._carg, B._carg, ctypes.c_uint64(budget), ctypes.byref(nv), ctypes.byref(cs),
                                    ctypes.byref(fl), ctypes.byref(nb))
            if rc != 0:
                raise RuntimeError(f"GrX_mxm_streamed failed with GrB_Info {rc}")
            stream_out.update(batches=int(nb.value), checksum=int(cs.value))
            return device.last_stats()
        keep.pop("C", None)  # (the previous product is released before the next one is allocated)
        C = gb.Matrix("INT64", hi - lo, n)
        if masked:
            rc = L.GrB_mxm(C._carg, A._carg, None, sr._carg, A._carg, B._carg, desc_s)
        else:
            rc = L.GrB_mxm(C._carg, None, None, sr._carg, A._carg, B._carg, None)
        if rc != 0:
            raise RuntimeError(f"GrB_mxm failed with GrB_Info {rc}")
        st = device.last_stats()
        keep["C"] = C
        return st

    for _ in range(warmup):
        st = step()
    barrier()
    t0 = time.perf_counter()
    device.timer_start()
    for _ in range(steps):
        st = step()
    ev_ms = device.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    verified = verify_mxm(torch, device, gb, L, sr, A, B, ip_a, col_a, ip_b, col_b, n, masked, keep.get("C"), st, stream_out, budget)
    keep.clear()
    t = torch.tensor([dt, float(st["out_nvals"]), float(st["flops"]), ev_ms, 1.0 if verified else 0.0], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        tmin = t.clone(); dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt, nnz_c, flops, ev_ms, verified = tmax[0].item(), tsum[1].item(), tsum[2].item(), tmax[3].item(), bool(tmin[4].item())
    else:
        nnz_c, flops = float(st["out_nvals"]), float(st["flops"])
    ms = dt / steps * 1e3
    nnz_a = float(col_b.numel())
    # nnz(A) (I + V_A) + flops (I + V_B) + nnz(C) (I + V_C) + 3 (n + 1) P with I = 4, P = 8; the operands are iso
    # (one stored value: V_A = V_B = 0), the product holds INT64 values (V_C = 8)
    alg_bytes = nnz_a * 4 + flops * 4 + nnz_c * 12 + 3 * (n + 1) * 8
    achieved = alg_bytes / world / (ev_ms / steps * 1e-3) / 1e9
    cpu = None
    if rank == 0 and want_cpu and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_mxm(ip_b, col_b, n, flops, masked)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = {"value": None, "unit": "nnz(C)/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    # (streamed products: the PMC pass of round 6 ran the streamed command itself, so its traffic describes this line too)
    traffic = measured_traffic(workload + ("_streamed" if streamed else ""), scale)
    kernel_s = ev_ms / steps * 1e-3
    line = {
        "metric": "SpGEMM nnz-out/s on R-MAT scale-%d" % scale, "value": nnz_c / (ms * 1e-3), "unit": "nnz(C)/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic", "verified": verified,
        "config": {"workload": f"rmat{scale} {workload}: " + ("C<!A.S> = A (+.x) A (complemented mask fused into the product)" if masked == "comp"
                                                            else "C<A.S> = A (+.x) A (mask-driven)" if masked else "C = A (+.x) A")
                   + ", INT64 ones" + (f"; row batches under {args.stream_budget_gb:g} GiB, output streamed (count + checksum)" if streamed else ""),
                   "nnz_A": nnz_a, "flops": flops, "nnz_C": nnz_c, "parallelism": (f"rank {blk[0]} of a {blk[1]}-way row shard (flop-balanced cuts), B replicated, compute only" if blk else
                                   f"row-shard x{world} (flop-balanced cuts), B replicated"), **stream_out},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "frac_moved": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "kernel": "k_spgemm_mhash / k_spgemm_mwin" if masked is True else "k_spgemm_unit (symbolic + numeric classes) / k_spgemm_unit_dense / k_spgemm_hash",
                     "kernel_ms_hip_events": ev_ms / steps, "algorithmic_bytes_per_launch": alg_bytes / world},
        "cpu_baseline": cpu, "stats": st}
    return line


def main_mxm(args, gb, torch, device, rank, world, dist, barrier):
    line = run_mxm(args, gb, torch, device, rank, world, dist, barrier, scale=args.scale, workload=args.workload, steps=args.steps,
                   warmup=args.warmup, want_cpu=(world == 1))
    if rank == 0:
        emit(line)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_mxm(indptr, col, n, flops_total, masked, target_flops=1.5e9):
    """The C oracle (OpenMP, all host cores) on a bounded sample of the same product: the first rows of A (as many as carry about
    `target_flops` multiplies) times the whole B; value = nnz(C_sample) / s."""
    import numpy as np

    from oracle import grb_oracle as O

    ip = indptr.cpu().numpy()
    cj = col.cpu().numpy().astype(np.int64)
    rowlen = np.diff(ip)
    fl_prefix = np.concatenate([[0], np.cumsum(rowlen[cj])])[ip]
    rows = int(min(n, max(64, np.searchsorted(fl_prefix, min(target_flops, float(fl_prefix[-1])), side="right"))))
    e1 = int(ip[rows])
    ones = np.ones(cj.size, np.int64)
    ob = O.OMat(n, n, ip, cj, ones, "INT64")
    oa = O.OMat(rows, n, ip[: rows + 1].copy(), cj[:e1].copy(), ones[:e1].copy(), "INT64")
    O.use_all_threads()
    t0 = time.perf_counter()
    T = (O.mxm(oa, ob, "plus_times", mask=oa, mask_struct=True, mask_comp=masked == "comp") if masked
         else O.mxm_product(oa, ob, "plus_times"))
    dt = time.perf_counter() - t0
    nnz_c = int(T.indices.size) if hasattr(T, "indices") else int(T[1].size)
    return {"value": nnz_c / dt, "unit": "nnz(C)/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"rows 0..{rows - 1} of A ({int(fl_prefix[rows]):,} of {int(flops_total):,} multiplies) times the whole B, one pass "
                      f"({dt:.2f} s); C oracle (oracle/grb_oracle.c, OpenMP) -- a CPU restatement, not SuiteSparse"}


def main_uniform(args, gb, torch, device, rank, world):
    """configs[0]: 4096 x 4096, 1 % uniform density, FP64 U[0,1) values, plus_times mxv with a dense U[0,1) operand -- the
    reference's CPU-runnable plumbing case, here through the same C ABI on the GPU, beside the CPU oracle and scipy."""
    import numpy as np
    import scipy.sparse as sp

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    assert world == 1
    n = 4096
    r, c, v = synthetic.uniform_coo(n, n, 0.01, 1, np.float64)
    x = np.random.default_rng(2).random(n)
    A = gb.Matrix.from_coo(r, c, v, dtype="FP64", nrows=n, ncols=n)
    u = gb.Vector.from_coo(np.arange(n), x, dtype="FP64", size=n)
    w = gb.Vector("FP64", size=n)
    sr = gb.semiring.plus_times
    for _ in range(args.warmup):
        w << A.mxv(u, sr)
    torch.cuda.synchronize()
    device.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w << A.mxv(u, sr)
    ev_ms = device.timer_stop()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    S = sp.csr_matrix((v, (r, c)), shape=(n, n))
    want = S @ x
    wi, wv = w.to_coo()
    has = np.flatnonzero(np.diff(S.indptr) > 0)
    verified = bool(np.array_equal(wi.astype(np.int64), has) and np.allclose(wv, want[has], rtol=1e-6, atol=0))
    oa = O.OMat.from_coo(r, c, v, n, n, "FP64")
    ou = O.OVec(n, np.arange(n), x, "FP64")
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        O.mxv(oa, ou, "plus_times")
        ts.append(time.perf_counter() - t1)
    t1 = time.perf_counter()
    for _ in range(20):
        S @ x
    t_scipy = (time.perf_counter() - t1) / 20
    nnz = int(r.size)
    alg = algorithmic_bytes_mxv(nnz, n, n, 8, 8, 8, accum=False, mask=False)
    kernel_ms = ev_ms / args.steps
    emit(({
        "metric": "GTEPS (mxv) on uniform 4096x4096 1% FP64", "value": nnz / dt / 1e9, "unit": "GTEPS", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "verified": verified,
        "config": {"workload": "uniform 4096x4096, density 0.01, FP64 U[0,1): w = A plus.times u (BASELINE.json configs[0])", "nnz": nnz},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_mxv_pull + k_mxv_seams",
                     "kernel_ms_hip_events": kernel_ms, "algorithmic_bytes_per_launch": alg,
                     "note": "2 MB of operands: launch-latency bound, not a roofline case"},
        "cpu_baseline": {"value": nnz / float(np.median(ts)) / 1e9, "unit": "GTEPS", "cores": O.num_threads(), "kind": "port",
                         "sample": f"same matrix and operand, median of 5 full passes of the C oracle ({np.median(ts) * 1e3:.3f} ms); "
                                   f"scipy.sparse csr @ x on one core: {t_scipy * 1e3:.3f} ms; not SuiteSparse"}}))


def cpu_baseline_loop(kind, scale=20):
    """The CPU column of the traversal lines: the same loop with the C oracle's mat-vec (OpenMP, all host cores) on the R-MAT graph of
    the same generator at scale 20 (a bounded sample: the scale-24 loop would take minutes on the host), from the largest-degree
    vertex.  value = Graph500-style TEPS of that run."""
    import numpy as np
    import torch

    from graphblas_amd import synthetic
    from oracle import grb_oracle as O

    n = 1 << scale
    ip, col = synthetic.rmat_csr(scale, device="cuda" if torch.cuda.is_available() else "cpu")
    vals = synthetic.edge_weights(col, scale).cpu().numpy() if kind == "sssp" else None
    ip, col = ip.cpu().numpy(), col.cpu().numpy().astype(np.int64)
    deg = np.diff(ip)
    src = int(np.argmax(deg))
    O.use_all_threads()
    if kind == "sssp":
        At = O.OMat(n, n, ip, col, vals, "FP32").transpose()  # (v @ G pulls over G')
        t0 = time.perf_counter()
        v = O.OVec(n, np.array([src]), np.array([0], np.float32), "FP32")
        its = 0
        while True:
            its += 1
            nv = O.mxv(At, v, "min_plus", w=v, accum="min")
            same = nv.idx.size == v.idx.size and np.array_equal(nv.idx, v.idx) and np.array_equal(nv.vals, v.vals)
            v = nv
            if same:
                break
        dt = time.perf_counter() - t0
        reached = v.idx
    else:
        At = O.OMat(n, n, ip, col, np.ones(col.size, bool), "BOOL").transpose()
        t0 = time.perf_counter()
        q = O.OVec(n, np.array([src]), np.array([True]), "BOOL")
        seen = np.zeros(n, bool)
        its = 0
        while q.idx.size:
            its += 1
            seen[q.idx] = True
            vis = O.OVec(n, np.flatnonzero(seen), np.ones(int(seen.sum()), bool), "BOOL")
            q = O.mxv(At, q, "lor_land", w=q, mask=vis, mask_comp=True, mask_struct=True, replace=True)
        dt = time.perf_counter() - t0
        reached = np.flatnonzero(seen)
    edges = int(deg[reached].sum())
    return {"value": edges / dt / 1e9, "unit": "GTEPS", "cores": O.num_threads(), "kind": "port",
            "sample": f"the same loop on the scale-{scale} graph of the same generator ({its} sweeps, {dt * 1e3:.0f} ms, {edges} edges counted) with the C "
                      "oracle's mat-vec (oracle/grb_oracle.c, OpenMP) -- a CPU restatement, not SuiteSparse"}


def main_bfs(args, gb, torch, device, rank, world):
    """A whole level-synchronous BFS on the device (notebooks/Example B.1 -- Level BFS.ipynb): per level one masked scalar
    assign (levels), one vxm over lor_land with the complemented structural mask + replace (push for thin frontiers, pull
    for thick ones) and one reduce that the host reads to decide whether to stop.  value = Graph500-style TEPS: edges
    incident to the visited vertices / time of the traversal; a step = one traversal from the largest-degree vertex."""
    from graphblas_amd import synthetic

    assert world == 1, "the BFS line is a single-GPU measurement"
    n = 1 << args.scale
    indptr, col = synthetic.rmat_csr(args.scale, device="cuda")
    one = torch.ones(1, dtype=torch.bool, device="cuda")
    deg = indptr[1:] - indptr[:-1]
    src = int(torch.argmax(deg).item())
    torch.cuda.synchronize()
    t_cold = time.perf_counter()  # (cold_ms: adoption of the CSR, the cached transpose, every layout the traversal builds, the first result)
    A = device.matrix_from_device_csr(indptr, col, one, n, n, "BOOL", iso=True)
    device.cache_transpose(A)  # vxm pulls over A' and pushes over A

    def traverse():
        v = gb.Vector("INT32", n)
        q = gb.Vector(bool, n)
        q[src] << True
        succ = gb.Scalar(bool)
        d = 0
        while True:
            d += 1
            v[:](mask=q.V) << d
            q(~v.S, replace=True) << q.vxm(A, gb.semiring.lor_land)
            succ << q.reduce(gb.monoid.lor, allow_empty=False)
            if not succ:
                break
        return v, d

    v, depth = traverse()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t_cold) * 1e3
    for _ in range(max(args.warmup - 1, 0)):
        v, depth = traverse()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v, depth = traverse()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # one more traversal, instrumented (not timed): per level the direction the library took and the SURVEY 8d bytes of that direction
    #   push: products x I + frontier x (I + 2 P) + reached x 1 B;  pull: entries of the unvisited rows x I (estimated from the share of
    #   unvisited VERTICES) + (n + 1) P + the frontier / visited / output bit vectors
    nnz_all = int(col.numel())
    alg_bytes, levels_log = 0.0, []
    vv = gb.Vector("INT32", n)
    qq = gb.Vector(bool, n)
    qq[src] << True
    dd = 0
    while True:
        dd += 1
        f_nv, seen_before = qq.nvals, vv.nvals
        vv[:](mask=qq.V) << dd
        qq(~vv.S, replace=True) << qq.vxm(A, gb.semiring.lor_land)
        st = device.last_stats()
        if st["method"] == 2:
            b = st["flops"] * 4 + f_nv * 20 + qq.nvals
        else:
            b = nnz_all * 4 * (1.0 - (seen_before + f_nv) / n) + (n + 1) * 8 + 3 * n / 8
        alg_bytes += b
        levels_log.append({"level": dd, "frontier": int(f_nv), "direction": "push" if st["method"] == 2 else "pull", "bytes": b})
        if qq.nvals == 0:
            break
    idx, _lev = v.to_coo()
    visited = torch.from_numpy(idx.astype("int64")).cuda()
    edges = int(deg[visited].sum().item())
    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_loop("bfs")
        except Exception as e:
            cpu = {"value": None, "unit": "GTEPS", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    emit(({
        "metric": "GTEPS (BFS traversal) on R-MAT scale-%d" % args.scale, "value": edges / dt / 1e9, "unit": "GTEPS", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bool", "data": "synthetic",
        "config": {"workload": f"rmat{args.scale} bfs: level BFS from the largest-degree vertex, whole loop through the C ABI",
                   "levels": depth, "visited_vertices": int(idx.size), "edges_counted_per_step": edges, "per_level": levels_log,
                   "cold_ms": cold_ms, "note": "ms_per_step / value are WARM-layout numbers (steady state of repeated traversals); cold_ms = from the adoption of "
                                               "the CSR to the first traversal's result, the cached transpose and every layout build included"},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / dt / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "whole traversal (wall clock incl. the host's per-level reads) over the sum of the levels' SURVEY 8d bytes in the direction taken"},
        "cpu_baseline": cpu}))


def main_sssp(args, gb, torch, device, rank, world):
    """The primer's SSSP loop (docs/getting_started/primer.rst:236-246) on the device: ``v(min) << min_plus(v @ G)`` until
    ``v.isequal(w)`` -- Bellman-Ford by vxm, fixed-point test by eWiseMult(eq) + reduce(land).  FP32 weights U{1..255}.
    value = Graph500-style TEPS: edges incident to the reached vertices / time of the whole loop."""
    from graphblas_amd import synthetic

    assert world == 1, "the SSSP line is a single-GPU measurement"
    n = 1 << args.scale
    indptr, col = synthetic.rmat_csr(args.scale, device="cuda")
    vals = synthetic.edge_weights(col, args.scale)
    deg = indptr[1:] - indptr[:-1]
    src = int(torch.argmax(deg).item())
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    G = device.matrix_from_device_csr(indptr, col, vals, n, n, "FP32")
    device.cache_transpose(G)

    def solve():
        v = gb.Vector("FP32", n)
        v[src] << 0.0
        its = 0
        while True:
            its += 1
            w = v.dup()
            v(gb.op.min) << gb.semiring.min_plus(v @ G)
            if v.isequal(w):
                break
        return v, its

    v, its = solve()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t_cold) * 1e3
    for _ in range(max(args.warmup - 1, 0)):
        v, its = solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v, its = solve()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # instrumented solve (not timed): bytes per sweep in the direction taken -- pull: all entries x (I + V) + (n + 1) P + the vectors (u, w
    # read and written); push: products x (I + V) + frontier x (I + V + 2 P) + touched outputs
    nnz_all = int(col.numel())
    alg_bytes, sweeps_log = 0.0, []
    vv = gb.Vector("FP32", n)
    vv[src] << 0.0
    while True:
        f_nv = vv.nvals
        ww = vv.dup()
        vv(gb.op.min) << gb.semiring.min_plus(vv @ G)
        st = device.last_stats()
        b = (st["flops"] * 8 + f_nv * 24 + vv.nvals * 8) if st["method"] == 2 else (nnz_all * 8 + (n + 1) * 8 + 3 * n * 4)
        alg_bytes += b + 2 * n * 4 * 2  # (+ the copy and the comparison of the fixed-point test)
        sweeps_log.append({"frontier": int(f_nv), "direction": "push" if st["method"] == 2 else "pull", "ordered": int(st["ordered"])})
        if vv.isequal(ww):
            break
    idx, _d = v.to_coo()
    reached = torch.from_numpy(idx.astype("int64")).cuda()
    edges = int(deg[reached].sum().item())
    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_loop("sssp")
        except Exception as e:
            cpu = {"value": None, "unit": "GTEPS", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    emit(({
        "metric": "GTEPS (SSSP) on R-MAT scale-%d" % args.scale, "value": edges / dt / 1e9, "unit": "GTEPS", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"rmat{args.scale} sssp: Bellman-Ford by vxm(min_plus) with accum min until isequal, whole loop through the C ABI",
                   "iterations": its, "ms_per_iteration": dt * 1e3 / its, "reached_vertices": int(idx.size), "edges_counted_per_step": edges,
                   "per_sweep": sweeps_log, "cold_ms": cold_ms,
                   "note": "ms_per_step / value are WARM-layout numbers; cold_ms = from the adoption of the CSR to the first solve's result, the cached "
                           "transpose and every layout build included"},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / dt / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "whole loop (wall clock incl. the host's fixed-point reads) over the sum of the sweeps' SURVEY 8d bytes in the direction taken"},
        "cpu_baseline": cpu}))


_REAL_STDOUT = None


def sssp_sharded_line(wl, gb, torch, device, dist, steps_cap=64):
    """The N-rank SSSP LOOP: Bellman-Ford relaxations  d = min(d, A min.+ d)  from one source over this run's row blocks,
    operands CHANGING from step to step, with the exchange chosen per step (sharded.OverlappedMxv(exchange="auto")): dense all-gathers while
    many distances move, (row, value) pairs once few do.  Unreached vertices hold a large finite distance (the library's full-operand
    kernels; 2^30 + w rounds the same way on both sides of the check).  Run twice -- "auto" and "dense" -- from the same start: the final
    replicas must agree bit for bit on every rank, and a checksum of the replica must be the same on every rank."""
    from graphblas_amd import sharded

    n, BIG = wl.n, float(1 << 30)
    src_vertex = 1
    out = {}
    finals = {}
    for mode in ("auto", "dense"):
        d0 = torch.full((n,), BIG, dtype=torch.float32, device="cuda")
        d0[src_vertex] = 0.0
        us = [device.vector_from_device(d0.clone()) for _ in range(2)]
        ws = [device.vector_from_device(wl._rows_of(d0, lo, hi).contiguous()) for lo, hi in wl.ranges]
        ov = sharded.OverlappedMxv(wl.As, ws, [None] * len(ws), us, wl.sr, accum=wl.accum, desc_name=None, presence=False,
                                   cyclic=wl.ov.cyclic, exchange=mode)
        ov.staged = wl.ov.staged
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        it = 0
        prev = None
        while it < steps_cap:
            ov.step()
            it += 1
            if mode == "auto":
                if ov.last_changed == 0 and it > 2:
                    break
            elif it >= out["auto"]["iterations"]:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        uv, _ub = device.vector_device_views(ov.current_u())
        finals[mode] = uv.clone()
        out[mode] = {"iterations": it, "ms": dt * 1e3, "bytes_sent_per_rank": int(sum(x[1] for x in ov.log)),
                     "methods": "".join("D" if x[0] == "dense" else "d" for x in ov.log),
                     "rows_changed_per_step": [x[2] for x in ov.log]}
        del ov, us, ws
    same = bool(torch.equal(finals["auto"], finals["dense"]))
    reached = int((finals["auto"] < BIG).sum().item())
    if dist is not None:
        img = finals["auto"].view(torch.int32).to(torch.int64)
        sums = torch.stack([img.sum(), (img * torch.arange(1, n + 1, device="cuda", dtype=torch.int64)).sum()])
        hi_, lo_ = sums.clone(), sums.clone()
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        same = same and bool(torch.equal(hi_, lo_))
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item())
    return {"workload": f"rmat{int(math.log2(n))} sssp_sharded: Bellman-Ford loop d = min(d, A min.+ d) from one source over this run's row blocks, "
                        "the exchange chosen per step from what the last step changed (auto) against the dense all-gather every step (dense)",
            "verified": same, "reached_vertices": reached, "auto": out["auto"], "dense": out["dense"],
            "exchange_bytes_ratio_dense_over_auto": out["dense"]["bytes_sent_per_rank"] / max(out["auto"]["bytes_sent_per_rank"], 1),
            "note": "methods: one letter per step, D = dense all-gather, d = (row, value) pairs; verified = both runs end on the same distances, "
                    "bit for bit, on every rank"}


def claim_stdout():
    """The driver reads ONE JSON line from stdout.  Libraries loaded later write there too (RCCL prints its version banner to the C
    stdout when a communicator comes up): file descriptor 1 is pointed at stderr for the rest of the run and the line goes out through a
    private copy of the original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def library_build_state():
    """Is the shared library the kernels were launched from up to date with the sources next to it?  (`make -q`: 0 = nothing to rebuild --
    the prebuilt .so that travelled with the tree is what ran; 1 = sources are newer than the library.)"""
    import subprocess

    csrc = os.path.join(ROOT, "python-graphblas_amd", "csrc")
    try:
        rc = subprocess.run(["make", "-q", "-C", csrc, "libgrb_mi355x.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60).returncode
    except Exception as e:
        return {"make_q": f"not run: {e!r}"}
    return {"make_q": rc, "up_to_date": rc == 0, "library": os.environ.get("GRB_MI355X_LIB") or os.path.join(csrc, "libgrb_mi355x.so")}


def respawn_under_launcher(args):
    """`python bench.py --gpus N` started without a launcher (no WORLD_SIZE in the environment): start N ranks of this very
    command under torch.distributed.run on this node -- or fail cleanly when the node has fewer than N GPUs."""
    import socket

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (args.share_gpus and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs on this node, found {have}; "
                         "not falling back to fewer ranks (the line would claim a rank count it did not run)\n")
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.workload == "kron26":  # configs[4]: Kronecker scale-26 min_plus mxv fp32 (8 GPUs; one rank's block with --block r/8)
        args.workload, args.scale = "mxv_min_plus", 26
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.block:
        respawn_under_launcher(args)
    claim_stdout()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    block = None
    if args.block and world == 1:
        block = tuple(int(x) for x in args.block.split("/"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and not block:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE=1: start it without a launcher (it spawns its own ranks) or with "
                             f"torch.distributed.run --nproc-per-node {args.gpus}\n")
            sys.exit(2)
        args.gpus = world
    if args.share_gpus:
        if args.backend != "gloo":
            sys.stderr.write("bench.py: --share-gpus needs --backend gloo (RCCL refuses two ranks on one device)\n")
            sys.exit(2)
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:  # (--force-dist without a launcher: a one-rank world on this process)
            import socket

            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            sock.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        seen = torch.ones(1, device="cuda")
        dist.all_reduce(seen)  # every rank adds one: the ranks really see each other over RCCL
        assert int(seen.item()) == world == dist.get_world_size(), (seen.item(), world)
    import graphblas_amd as gb
    from graphblas_amd import _lib as _lib_mod
    from graphblas_amd import device

    gb.init()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    warm_build = {"done": False, "ms": None}

    def warm_layout_build(sr, visited, scale):
        # Round 6: `layout_build_call_ms` is the wall time of the call that builds the cached layouts IN THIS PROCESS -- the first process on a fresh box
        # also pays for the allocator's first gigabytes from the driver and for code paged in from a cold file cache there (the same call: 44 ms in a
        # process that follows another on the box, 101-382 ms as the first).  `layout_build_warm_ms` is the same build once more in this process, on a
        # second handle of the same graph (pools grown, code resident): what the library's kernels and copies cost.
        if warm_build["done"] or world != 1 or block or args.no_warm_build:
            return
        warm_build["done"] = True
        w0 = MxvWorkload(gb, torch, scale, 0, 1, sr, visited)
        w0.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        w0.step()
        torch.cuda.synchronize()
        warm_build["ms"] = (time.perf_counter() - t) * 1e3
        del w0
        import gc

        gc.collect()
        device.trim_memory()

    def run(workload, scale, steps, warmup, weights="int255", fresh_outputs=False, ranked=None):
        sr = {"mxv_min_plus_masked": "min_plus", "mxv_lor_land_masked": "lor_land", "mxv_min_plus": "min_plus"}[workload]
        visited = 0.0 if workload == "mxv_min_plus" else args.visited
        wl = MxvWorkload(gb, torch, scale, rank, world, sr, visited, block=block, chunks=max(1, args.overlap_chunks), force_dist=args.force_dist,
                         weights=weights, fresh_outputs=(steps if fresh_outputs else 0), ranked=(args.ranked if ranked is None else ranked),
                         stripe=args.stripe, shard_setup=not args.no_shard_setup)
        # The first product of a matrix runs on its CSR arrays as they are; the second builds the cached layouts (hot-column coding,
        # long / short split, class strips).  Both are part of the warm-up and are timed apart (wall clock around a synchronised call).
        def timed_call():
            torch.cuda.synchronize()
            t = time.perf_counter()
            wl.step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) * 1e3

        first_call_ms = timed_call()
        second_call_ms = timed_call()
        cache_bytes = 0
        for A in wl.As:
            cb = ctypes.c_uint64(0)
            _lib_mod.lib.GrX_Matrix_cache_bytes(A._carg, ctypes.byref(cb))
            cache_bytes += int(cb.value)
        for _ in range(max(warmup - 2, 0)):
            wl.step()
        if wl.fresh_ws:
            wl.prepare_fresh()  # (every timed step works on an output of its own, already in the matrix's vertex order)
        barrier()
        t0 = time.perf_counter()
        device.timer_start()
        for _ in range(steps):
            wl.step()
        ev_ms = device.timer_stop()
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt, float(wl.nnz_active_local), ev_ms], dtype=torch.float64, device="cuda")
        if dist is not None:
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dt, edges, ev_ms = tmax[0].item(), tsum[1].item(), tmax[2].item()
        else:
            edges = float(wl.nnz_active_local)
        ms_per_step = dt / steps * 1e3
        kernel_ms = ev_ms / steps
        achieved = wl.bytes_per_step() / (kernel_ms * 1e-3) / 1e9
        stats = device.last_stats()
        verified = wl.verify()
        if scale >= 22 and workload == "mxv_min_plus_masked" and weights == "int255" and not fresh_outputs and not (args.ranked if ranked is None else ranked):
            warm_layout_build(sr, visited, scale)
        res = {
            "verified": verified,
            "first_call_ms": first_call_ms,
            "layout_build_call_ms": second_call_ms,
            "layout_build_warm_ms": warm_build["ms"],
            "preprocess_bytes": cache_bytes,
            "matrix_bytes": int(wl.nnz_local * (4 + wl.v_a) + (wl.m + len(wl.As)) * 8),
            "value": edges / (ms_per_step * 1e-3) / 1e9,
            "ms_per_step": ms_per_step,
            "dtype": wl.dtype_name,
            "edges_per_step": edges,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                         "traffic": measured_traffic(workload, scale) if (world == 1 and not block and wl.ov is None) else None,
                         "kernel": MXV_KERNELS_NOTE,
                         "kernel_ms_hip_events": kernel_ms, "algorithmic_bytes_per_launch": wl.bytes_per_step()},
            "stats": stats,
            "shard_setup": bool(wl.shard_setup),
        }
        if wl.ov is not None:
            res["roofline"]["note"] = ("per rank: this rank's algorithmic bytes over the HIP-event time of its step on the library's stream "
                                       "(products + waits for the exchanges), max over ranks")
            timed = wl.ov.log[-steps:] if steps else []
            res["exchange"] = {"chunks_per_rank": wl.ov.chunks, "replicas_of_u": 2, "staged_through_torch_buffers": wl.ov.staged,
                               "presence_words_travel": wl.ov.presence, "collective": "all_gather_into_tensor per chunk, async_op",
                               # what travelled in the timed steps: the dense all-gather SURVEY 8e prescribes is the headline's
                               # method; the sparse (row, value) exchange of a loop whose operand changes is the `sssp_sharded` line under extra
                               "method": wl.ov.exchange, "methods_of_timed_steps": {m: sum(1 for x in timed if x[0] == m) for m in ("dense", "delta")},
                               "bytes_per_step": (sum(x[1] for x in timed) / max(len(timed), 1)) if timed else None,
                               "bytes_per_step_note": "bytes THIS rank contributes per step; every rank receives (world - 1) times that"}
        if res["roofline"]["traffic"]:
            res["roofline"]["traffic_band"] = measured_traffic_band(workload, scale)
            # the PMC traffic (L2 misses x 128 B, measured in separate profiled runs of this workload) over this run's kernel time
            res["roofline"]["traffic_GBps"] = res["roofline"]["traffic"] / (kernel_ms * 1e-3) / 1e9
            # the second fraction: what the call MOVED (counter traffic) over its time and the peak, next to `frac`
            # (what the algorithm NEEDS over the same time).  frac_moved > frac: wasted re-reads; frac_moved < frac: the kernels read
            # fewer bytes than the formula counts (early exits of terminal monoids, dictionary-coded values)
            res["roofline"]["frac_moved"] = res["roofline"]["traffic_GBps"] / HBM_PEAK_GBS
        return wl, res

    def cpu_line(wl):
        ss_lib, ss_note = probe_suitesparse()
        cpu = None
        if ss_lib is not None and wl.semiring == "min_plus":
            try:
                cpu = cpu_baseline_suitesparse(ss_lib, wl, torch)
            except Exception as e:
                ss_note += f"; driving it failed: {e!r}"
        if cpu is None:
            try:
                cpu = cpu_baseline_mxv(wl, torch)
            except Exception as e:  # the baseline must never take the bench line down
                cpu = {"value": None, "unit": "GTEPS", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        cpu["suitesparse_probe"] = ss_note
        return cpu

    if args.workload == "uniform_fp64":
        return main_uniform(args, gb, torch, device, rank, world)
    if args.workload == "bfs":
        return main_bfs(args, gb, torch, device, rank, world)
    if args.workload == "sssp":
        return main_sssp(args, gb, torch, device, rank, world)
    if args.workload in ("mxm_plus_times", "mxm_plus_times_masked", "mxm_plus_times_cmask"):
        return main_mxm(args, gb, torch, device, rank, world, dist, barrier)
    wl, res = run(args.workload, args.scale, args.steps, args.warmup)
    cpu = cpu_line(wl) if (rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_dist) else None
    # ---- the rest of the BASELINE metric, in the same line under `extra`: the BFS level step (configs[2]) and SpGEMM A (+.x) A
    #      at scale 20 (one GPU: the materialised product) and scale 22 (configs[3]; row batches on one GPU, row-sharded on N) ----
    extra = []
    default_line = args.workload == "mxv_min_plus_masked" and args.scale == 24 and not block
    if default_line and not args.no_extra:
        if wl.ov is not None and args.workload == "mxv_min_plus_masked":
            # the N-rank SSSP loop on this run's blocks (operands that change; the exchange chosen per step)
            try:
                extra.append(sssp_sharded_line(wl, gb, torch, device, dist))
            except Exception as e:  # an extra line must never take the headline down
                extra.append({"workload": "sssp_sharded", "error": repr(e)})
        del wl

        def freed():
            import gc

            gc.collect()
            device.trim_memory()
            torch.cuda.empty_cache()

        freed()
        wl2, r = run("mxv_lor_land_masked", 24, args.steps, args.warmup)
        c2 = cpu_line(wl2) if (rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_dist) else None
        extra.append({"workload": "rmat24 mxv_lor_land_masked: q<~visited.S, replace> = A lor.land q, iso BOOL, frontier density 0.3 (configs[2])",
                      **{k: r[k] for k in ("value", "ms_per_step", "dtype", "roofline", "verified")}, "unit": "GTEPS", "cpu_baseline": c2})
        del wl2
        freed()
        # ---- the headline on inputs that lend it nothing: real-valued weights U[0,1) -- no value dictionary --,
        #      and every step on an output of its own with the operand alternating -- no step at the fixed point, every admitted row stores
        for label, kw in (("rmat24 mxv_min_plus_masked, FP32 weights U[0,1) instead of U{1..255}: the value dictionary cannot apply", dict(weights="real")),
                          ("rmat24 mxv_min_plus_masked, every timed step on its own copy of the initial distances, operand alternating between two "
                           "vectors: the write rule stores in every step (the default line repeats one call at its fixed point)", dict(fresh_outputs=True)),
                          ("rmat24 mxv_min_plus_masked on a graph whose vertices the APPLICATION numbered by falling in-degree once, up front (not the "
                           "scrambled labels of the headline), every matrix carrying GrX_Matrix_hint_ranked" +
                           (f"; rows dealt to the {world} ranks block-cyclically in stripes of {args.stripe}, the gathered slices dealt back the same way"
                            if (world > 1 or args.force_dist) else ""), dict(ranked=True))):
            if kw.get("fresh_outputs") and (world > 1 or args.force_dist):
                continue  # (the N-rank step feeds every output back as the next operand already)
            if kw.get("ranked") and args.ranked:
                continue
            try:
                wl3, r = run("mxv_min_plus_masked", 24, args.steps, args.warmup, **kw)
                extra.append({"workload": label, **{k: r[k] for k in ("value", "ms_per_step", "dtype", "roofline", "verified")}, "unit": "GTEPS",
                              "stats": {k: r["stats"].get(k) for k in ("ordered", "value_dict", "fused_epilogue", "reorders")}})
                del wl3
            except Exception as e:  # an extra line must never take the headline down
                extra.append({"workload": label, "error": repr(e)})
            freed()
        for scale, steps, warmup, wk in ((20, 3, 1, "mxm_plus_times"), (20, 2, 1, "mxm_plus_times_cmask"), (22, 2, 1, "mxm_plus_times")):
            if scale == 20 and (world > 1 or args.force_dist):
                continue  # (the sharded runs carry the scale-22 product, the size the north star quotes for 1 -> 8 GPUs)
            try:
                line = run_mxm(args, gb, torch, device, rank, world, dist, barrier, scale=scale, workload=wk, steps=steps,
                               warmup=warmup, want_cpu=(world == 1 and wk == "mxm_plus_times"))
            except Exception as e:  # an extra line must never take the headline down
                line = {"error": repr(e)}
            extra.append({"workload": f"rmat{scale} {wk} (configs[3]" + (")" if scale == 22 else " at scale 20: fits one GPU as an object"
                                                                          + (", under the complemented structural mask ~A.S" if wk.endswith("cmask") else "") + ")"),
                          **{k: line[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "verified", "config", "roofline", "cpu_baseline", "error") if k in line}})
            freed()
    if rank == 0:
        out = {
            "metric": "GTEPS (mxv) on R-MAT scale-%d" % args.scale,
            "value": res["value"],
            "unit": "GTEPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": res["dtype"],
            "data": "synthetic",
            "config": {"workload": (f"rmat{args.scale} {args.workload}: w<~visited.S> = min(w, A min.+ u), "
                                    f"edge factor 16, visited density {args.visited}, dense u, FP32 weights U{{1..255}} (BASELINE.md section 3)"
                                    + (" -- NOT the headline input: --ranked numbers the vertices by falling in-degree instead of scrambling them; "
                                       if args.ranked else "; ") +
                                    f"layouts: ordered={int(res['stats'].get('ordered', 0))} (popularity-ordered twin), "
                                    f"value_dict={int(res['stats'].get('value_dict', 0))} (distinct values coded in one byte; the same call on U[0,1) "
                                    "weights is under extra)")
                       if args.workload == "mxv_min_plus_masked" else f"rmat{args.scale} {args.workload}",
                       "ordered": int(res["stats"].get("ordered", 0)), "value_dict": int(res["stats"].get("value_dict", 0)),
                       "edges_counted_per_step": res["edges_per_step"],
                       "parallelism": ((f"rank {block[0]} of a {block[1]}-way row shard, compute only" + (f" (ranked labels, stripes of {args.stripe} rows dealt block-cyclically)" if args.ranked else "")) if block else
                                       f"row-shard x{world}" + (f", ranked labels, stripes of {args.stripe} rows dealt block-cyclically" if args.ranked else "") +
                                       (f" ({res['exchange']['chunks_per_rank']} row blocks per rank) + RCCL all-gather of the w slices into "
                                                                "the other replica of u, overlapped with the next block's product" if "exchange" in res else ""))},
            "verified": res["verified"],
            "shard_setup": ("the blocks were handed the global column counts (sharded.shard_setup / GrX_Matrix_shard_setup): the library ranked the "
                            "columns and laid the blocks out in that order" + ("; the replicas of u are RCCL buffers (pinned natural): the N-rank "
                            "products run the natural-order layouts all the same, compute-only block runs (--block) the ordered ones"
                            if "exchange" in res else "")) if res.get("shard_setup") else None,
            "first_call_ms": res["first_call_ms"],
            "layout_build_call_ms": res["layout_build_call_ms"],
            "layout_build_warm_ms": res.get("layout_build_warm_ms"),
            "preprocess_bytes": res["preprocess_bytes"],
            "matrix_bytes": res["matrix_bytes"],
            "roofline": res["roofline"],
            "cpu_baseline": cpu,
            "stats": res["stats"],
            "build": library_build_state(),
        }
        if "exchange" in res:
            out["exchange"] = res["exchange"]
        if args.share_gpus or args.backend != "nccl" or (args.force_dist and world == 1):
            out["rehearsal"] = (f"backend {args.backend}" + (", ranks share GPUs" if args.share_gpus else "") +
                                (", ONE rank with --force-dist (a one-rank world: RCCL initialised, the collectives ran on the library's buffers)"
                                 if args.force_dist and world == 1 else "") +
                                ": the N-rank code path was exercised; the numbers are NOT a multi-GPU measurement")
        if extra:
            out["extra"] = extra
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


MXV_KERNELS_NOTE = ("one GrB_mxv call: k_long_init (BOOL under a terminal monoid: with the bottom-up probe of the long rows) + k_mxv_hstrip + k_mxv_ctile (k_long_compact_* + k_mxv_long_grp "
                    "for BOOL matrices) + k_mxv_rtile (round 5: the short rows as "
                    "sorted row tiles; k_mxv_rtile_bool for BOOL operands given as presence / value pairs; k_mxv_rows_tag for the calls the tiles do not take) "
                    "(+ k_rows_tail when the write rule touches the empty tail) on the matrix's popularity-ordered layouts -- no per-call operand image; "
                    "k_x_image in front of them on the natural-order layouts (order_mode 0, row blocks of a sharded run); k_mxv_pull + k_mxv_seams below the split threshold")


if __name__ == "__main__":
    main()
