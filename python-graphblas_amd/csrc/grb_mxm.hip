// grb_mxm.hip -- GrB_mxm: two-pass (symbolic + numeric) row-wise Gustavson SpGEMM with LDS hash
// accumulators sized for gfx950 (160 KiB LDS per CU) and a dense-accumulator path for hub rows,
// followed by the GraphBLAS write rule C<M,replace> = accum(C, T).
//
// Reference call site: graphblas/core/matrix.py:2264-2331 (expression :2318-2328) -> C ``GrB_mxm``,
// arguments marshalled at core/base.py:496-503.  The arithmetic replaced is SuiteSparse:GraphBLAS's
// GrB_mxm (Gustavson / hash / dot methods named at core/ss/descriptor.py:77-83); it is not in /root/reference.
//
// Pipeline (DESIGN.md "SpGEMM"):
//   1. flops: f[p] = nnz(B(col[p], :)) per stored entry of A, exclusive scan -> per-row upper bound ub_i.
//   2. rows are binned by ub_i (symbolic) / by nnz(T_i) (numeric) with one stable radix sort of (bin,row).
//   3. symbolic: a 256-thread workgroup per row inserts column keys into an LDS hash table (256 / 2048 /
//      32768 keys); rows above that use a per-workgroup global bitmap.  Output: nnz(T_i).
//   4. exclusive scan -> row pointers of T; allocate T.
//   5. numeric: same hash insertion with LDS value accumulators (256 / 2048 / 8192 entries: the table is now
//      sized by the exact nnz(T_i)); keys are compacted, bitonic-sorted in LDS and written with their values,
//      so rows of T come out sorted.  Heavier rows accumulate into a dense per-workgroup accumulator in HBM
//      (values + presence bitmap) and are emitted by an ordered bitmap sweep.
//   6. write rule: no mask and no accum -> T becomes C; otherwise a row-merge of (C_old, T, Mask).
#include <algorithm>
#include <vector>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

namespace grb {

constexpr int MM_BLOCK = 256;

struct MxmArgs {
    int64_t m, n;  // T is m x n
    const int64_t *Ap;
    const int32_t *Aj;
    const void *Ax;
    int a_iso;
    const int64_t *Bp;
    const int32_t *Bj;
    const void *Bx;
    int b_iso;
    int monoid, mult;
    int need_a, need_b;
    // outputs
    int64_t *row_nnz;  // symbolic result (counts), later turned into Tp by a scan
    const int64_t *Tp;
    int32_t *Tj;
    void *Tx;
    // dense accumulators for hub rows (one slice per workgroup)
    uint64_t *spa_bits;
    void *spa_vals;
    int64_t spa_words;  // 64-bit words per slice
    // column windows of B for the LDS dense-window numeric kernel: woff[k*(n_win+1) + w] = first entry of B(k,:)
    // (relative to Bp[k]) whose column is >= w * MM_WIN
    const int32_t *woff;
    int n_win;
    // round 5: the (row, window) units of a product walk GROUPS of win_f consecutive windows (1, 2, 4 or 8: the unit kernels'
    // template parameter F) -- the window is a function of n, so that a row of a scale-22 matrix is cut into as many units as a row
    // of a scale-20 matrix.  The tables (woff, wcnt, wbm, cm_woff) stay per 16 Ki-column window: a group whose row holds more than
    // the densest compact class takes (mxm_unit_dense entries) is split into its windows again, each a unit as before.
    int win_f;
    // XCD-aware unit order (round 5): workgroups are dispatched round-robin over the 8 XCDs, each with an L2 of its own; the units of a
    // row -- its windows -- follow each other in the unit order and read the SAME rows of B.  With xcd_map every XCD takes a contiguous
    // eighth of the order (workgroup b works on position (b mod 8) grid / 8 + b / 8), so a row of B is fetched into one L2, not eight.
    int xcd_map;
    // streamed product (GrX_mxm_streamed): the wrapping sum of the values the numeric kernels store, folded into the store --
    // 1024 counters a 128-byte line apart (one address would serialise millions of atomics); nullptr: not wanted
    unsigned long long *csum;
    // (row, window) units (k_spgemm_unit): per row of the symbolic pass n_win + 1 numbers -- counts, then offsets inside the row;
    // wrow[row] = the row's slot in wcnt (-1: the symbolic pass counted the row with a hash kernel)
    int32_t *wcnt;
    int32_t *wrow;
    unsigned long long *class_count;         // symbolic pass: units per class of the numeric pass (device, MU_NCLS numbers)
    const unsigned long long *class_known;   // numeric pass: the same on the host
    // bitmaps of the units the symbolic pass found beyond bm_min_cnt entries, kept for the numeric pass (which then skips its own
    // pass A): a pool of bm_cap bitmaps of MM_WIN bits handed out by an atomic cursor, wbm[slot * n_win + w] = the unit's
    // bitmap or -1 (small unit, or the pool ran out: the numeric pass recomputes)
    unsigned long long *bm_pool;
    unsigned long long *bm_cursor;
    int64_t bm_cap;  // (per sub-pool)
    int bm_pools;    // (a power of two)
    int32_t *wbm;
    int bm_min_cnt;
    // symbolic units: sym_wg consecutive windows of a row per unit (the prologue -- row, row bounds, entries of A, row pointers of B --
    // once per group); rows with more than sym_plen_max entries of A are skipped by such a launch (0: none is) -- their units would
    // run their windows' thousands of batches one after the other -- and walked one window per unit from the list sym_list
    int sym_wg, sym_plen_max;
    const int32_t *sym_list;  // positions (in `rows`) of the rows of a list launch, or nullptr: positions ridx0 ...
    int abl;  // -DGRB_ABLATE builds: timing switches of the unit kernels (bits 20.. of debug_flags); results are wrong on purpose
    // mask-driven product (T restricted to the pattern of a non-complemented mask): results land in the mask's own layout
    const int64_t *Mp;
    const int32_t *Mj;
    void *cap_val;           // per mask entry: the product's value ...
    unsigned char *cap_hit;  // ... and whether any product hit it
    // complemented mask fused into the product (C<!M>): the pattern of the mask entries that forbid a position (structural: all of
    // them, valued: the true ones), sorted rows; T never holds a forbidden position, and the symbolic counts are those of T
    // without them.  cm_woff = the window offsets of the forbidden rows (m x (n_win + 1), as woff for B) for the unit kernels.
    const int64_t *CMp;
    const int32_t *CMj;
    const int32_t *cm_woff;
};

// the hash kernels keep a forbidden column as the key -(j + 2): it occupies its slot (probing stays consistent), a product that
// finds it is dropped, and the compaction (keys >= 0) never sees it
__device__ __forceinline__ int forbidden_key(int j) { return -(j + 2); }

// inclusive sum over the 64 lanes of a wavefront in the vector ALU (DPP row shifts inside the rows of 16 lanes, then the two row
// broadcasts): six add instructions -- the __shfl_up form is six ds_bpermute round trips through the LDS pipeline, a dependent
// chain of ~600 cycles in kernels whose time is the length of that chain
__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// a load through a pointer that went through LDS: the compiler no longer knows it points to global memory and emits FLAT loads
// (counted on both memory counters, waited for together with the LDS operations) -- say so
__device__ __forceinline__ int load_global_i32(const int32_t *p) { return *(const __attribute__((address_space(1))) int32_t *)p; }

// streamed product: what a stored value adds to the wrapping checksum (floating-point values by their integer part, as the separate
// pass over the product did), and the wavefront's share into one of 1024 counters (GrX_mxm_streamed adds them up)
constexpr int MM_CSUM_SLOTS = 1024;
template <typename T>
__device__ __forceinline__ unsigned long long checksum_term(T v)
{
    if constexpr (std::is_floating_point<T>::value) return (unsigned long long)(long long)v;
    else return (unsigned long long)v;
}
__device__ __forceinline__ void checksum_commit(const MxmArgs &a, unsigned long long mine)
{
    if (!a.csum) return;  // (uniform)
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&a.csum[(size_t)(blockIdx.x & (MM_CSUM_SLOTS - 1)) * 16], mine);
}

// (grids of the unit kernels are multiples of 8 when xcd_map is set: every position is covered)
__device__ __forceinline__ int64_t xcd_block(const MxmArgs &a)
{
    if (!a.xcd_map) return (int64_t)blockIdx.x;
    const int64_t per = ((int64_t)gridDim.x + 7) >> 3;
    return (int64_t)(blockIdx.x & 7u) * per + (int64_t)(blockIdx.x >> 3);
}

__device__ __forceinline__ unsigned hash_col(int c, int table_mask) { return ((unsigned)c * 2654435761u) & (unsigned)table_mask; }

// ---- step 1 helpers ---------------------------------------------------------------------------------------
__global__ void k_nnz_flops(const int32_t *Aj, int64_t nnzA, const int64_t *Bp, int64_t *f)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnzA) {
        const int c = Aj[p];
        f[p] = Bp[c + 1] - Bp[c];
    } else if (p == nnzA) f[p] = 0;
}

__device__ __forceinline__ int bin_of(int64_t x, int64_t b1, int64_t b2, int64_t b3)
{
    return x == 0 ? 0 : (x <= b1 ? 1 : (x <= b2 ? 2 : (x <= b3 ? 3 : 4)));
}

// size[i] = F[Ap[i+1]] - F[Ap[i]]  (F == nullptr: size[i] is already in `size`);  key = bin, payload = row
// (wrow: rows with a slot in the unit tables -- wrow[i] >= 0 -- go to bin 4 whatever their size)
// (extra_ptr: row pointers whose row lengths count towards the bin of a non-empty row -- the forbidden columns of a fused
//  complemented mask sit in the hash tables beside the row's own)
__global__ void k_row_bins(const int64_t *Ap, const int64_t *F, int64_t m, int64_t *size, int64_t b1, int64_t b2,
                           int64_t b3, uint64_t *binkey, uint32_t *rowid, const int32_t *wrow, const int64_t *extra_ptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int64_t s = F ? F[Ap[i + 1]] - F[Ap[i]] : size[i];
    if (F) size[i] = s;
    if (extra_ptr && s > 0) s += extra_ptr[i + 1] - extra_ptr[i];
    binkey[i] = (wrow && wrow[i] >= 0 && s > 0) ? 4ull : (uint64_t)bin_of(s, b1, b2, b3);
    rowid[i] = (uint32_t)i;
}

// bin_start[b] = first position in sorted keys with key >= b, b = 0..5
__global__ void k_bin_starts(const uint64_t *keys, int64_t m, int64_t *bin_start)
{
    const int b = threadIdx.x;
    if (b > 5) return;
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint64_t)b) lo = mid + 1;
        else hi = mid;
    }
    bin_start[b] = lo;
}

// Visit every product A(row,k) * B(k,j) with the whole workgroup (every thread must call).  The row's entries come
// BLOCK at a time, one per thread (k, the bounds of B(k,:): dependent loads, but 256 of them in flight); a workgroup
// scan of the B row lengths numbers the products, and the threads take them round-robin -- product t belongs to the last
// entry whose first product number is <= t (binary search in LDS).  A hub column with 10^5 entries is thus shared by all
// threads instead of serialising one 16-lane group, and consecutive threads read consecutive entries of B.
template <int BLOCK = MM_BLOCK, typename F>
__device__ __forceinline__ void foreach_product(const MxmArgs &a, int64_t row, F &&f)
{
    __shared__ int s_fp_scan[BLOCK + 1];
    __shared__ int64_t s_fp_qb[BLOCK];
    __shared__ int s_fp_wave[BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t pend = a.Ap[row + 1];
    for (int64_t pc = a.Ap[row]; pc < pend; pc += BLOCK) {
        const int64_t p = pc + tid;
        int len = 0;
        int64_t qb = 0;
        if (p < pend) {
            const int k = a.Aj[p];
            qb = a.Bp[k];
            len = (int)(a.Bp[k + 1] - qb);
        }
        int incl = wave_inclusive_sum(len);
        if (lane == 63) s_fp_wave[wv] = incl;
        __syncthreads();
        int wave_off = 0, total = 0;
        for (int x = 0; x < BLOCK / 64; x++) {
            if (x < wv) wave_off += s_fp_wave[x];
            total += s_fp_wave[x];
        }
        s_fp_scan[tid] = wave_off + incl - len;
        s_fp_qb[tid] = qb;
        __syncthreads();
        for (int t = tid; t < total; t += BLOCK) {
            int lo = 0;  // the last entry whose first product number is <= t (BLOCK is a power of two; three VALU per step)
#pragma unroll
            for (int st = BLOCK / 2; st > 0; st >>= 1)
                if (s_fp_scan[lo + st] <= t) lo += st;
            const int64_t q = s_fp_qb[lo] + (t - s_fp_scan[lo]);
            f(a.Bj[q], pc + lo, q);
        }
        __syncthreads();
    }
}

// ---- LDS hash kernels -------------------------------------------------------------------------------------------
// One workgroup per row.  NUMERIC=false: count distinct columns.  NUMERIC=true: accumulate, sort, emit.
template <typename T, int TABLE, bool NUMERIC>
__global__ __launch_bounds__(MM_BLOCK) void k_spgemm_hash(const MxmArgs a, const uint32_t *rows)
{
    using W = typename Widen<T>::type;
    constexpr int NV = NUMERIC ? TABLE : 1;
    constexpr int NS = NUMERIC ? TABLE / 2 : 1;
    __shared__ int s_key[TABLE];
    __shared__ W s_val[NV];
    __shared__ int s_sorted[NS];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int64_t row = rows[blockIdx.x];
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    for (int k = tid; k < TABLE; k += MM_BLOCK) {
        s_key[k] = -1;
        if (NUMERIC) s_val[k] = monoid_identity<T, W>(monoid);
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (a.CMp) {  // fused complemented mask: the forbidden columns take their slots first
        const int64_t mhi = a.CMp[row + 1];
        for (int64_t p = a.CMp[row] + tid; p < mhi; p += MM_BLOCK) {
            const int j = a.CMj[p];
            unsigned h = hash_col(j, TABLE - 1);
            while (atomicCAS(&s_key[h], -1, forbidden_key(j)) != -1) h = (h + 1) & (TABLE - 1);
        }
        __syncthreads();
    }

    int my_new = 0;
    foreach_product(a, row, [&](int j, int64_t p, int64_t q) {
        unsigned h = hash_col(j, TABLE - 1);
        bool dropped = false;
        while (true) {
            const int old = atomicCAS(&s_key[h], -1, j);
            if (old == -1) { my_new++; break; }
            if (old == j) break;
            if (old == forbidden_key(j)) { dropped = true; break; }
            h = (h + 1) & (TABLE - 1);
        }
        if (NUMERIC && !dropped) {
            const T av = a.need_a ? Ax[a.a_iso ? 0 : p] : (T)0;
            const T bv = a.need_b ? Bx[a.b_iso ? 0 : q] : (T)0;
            const W prod = (W)apply_binop<T>(mult, av, bv);
            if (monoid == OP_ANY) s_val[h] = prod;
            else atomic_combine<W>(&s_val[h], prod, monoid);
        }
    });
    if (!NUMERIC) {
        if (my_new) atomicAdd(&s_cnt, my_new);
        __syncthreads();
        if (tid == 0) a.row_nnz[row] = s_cnt;
        return;
    }
    __syncthreads();
    // ---- compact the occupied keys, sort them, emit (col, value) in column order ------------------------
    for (int k = tid; k < TABLE; k += MM_BLOCK) {
        const int key = s_key[k];
        if (key >= 0) s_sorted[atomicAdd(&s_cnt, 1)] = key;
    }
    __syncthreads();
    const int cnt = s_cnt;
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    for (int k = cnt + tid; k < np2; k += MM_BLOCK) s_sorted[k] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += MM_BLOCK) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const int x = s_sorted[lo], y = s_sorted[hi];
                if ((x > y) == up) { s_sorted[lo] = y; s_sorted[hi] = x; }
            }
            __syncthreads();
        }
    }
    const int64_t base = a.Tp[row];
    T *Tx = (T *)a.Tx;
    unsigned long long csum_mine = 0;
    for (int k = tid; k < cnt; k += MM_BLOCK) {
        const int key = s_sorted[k];
        unsigned h = hash_col(key, TABLE - 1);
        while (s_key[h] != key) h = (h + 1) & (TABLE - 1);
        a.Tj[base + k] = key;
        const T v = from_acc<T, W>(s_val[h]);
        Tx[base + k] = v;
        csum_mine += checksum_term<T>(v);
    }
    checksum_commit(a, csum_mine);
}

// ---- dense-accumulator (SPA) kernels for hub rows ------------------------------------------------------------------
// Persistent workgroups: workgroup b owns slice b of spa_bits / spa_vals and walks rows b, b+G, ...
template <typename T, bool NUMERIC>
__global__ __launch_bounds__(MM_BLOCK) void k_spgemm_spa(const MxmArgs a, const uint32_t *rows, int64_t nrows_bin)
{
    using W = typename Widen<T>::type;
    __shared__ int s_cnt;
    __shared__ int s_wave[MM_BLOCK / 64];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    unsigned long long *bits = (unsigned long long *)(a.spa_bits + (int64_t)blockIdx.x * a.spa_words);
    W *vals = NUMERIC ? ((W *)a.spa_vals + (int64_t)blockIdx.x * a.spa_words * 64) : nullptr;
    unsigned long long csum_mine = 0;
    for (int64_t r = blockIdx.x; r < nrows_bin; r += gridDim.x) {
        const int64_t row = rows[r];
        if (tid == 0) { s_cnt = 0; s_base = 0; }
        __syncthreads();
        int my_new = 0;
        foreach_product(a, row, [&](int j, int64_t p, int64_t q) {
            const unsigned long long bit = 1ull << (j & 63);
            if (NUMERIC) {
                atomicOr(&bits[j >> 6], bit);
                const T av = a.need_a ? Ax[a.a_iso ? 0 : p] : (T)0;
                const T bv = a.need_b ? Bx[a.b_iso ? 0 : q] : (T)0;
                const W prod = (W)apply_binop<T>(mult, av, bv);
                if (monoid == OP_ANY) vals[j] = prod;
                else atomic_combine<W>(&vals[j], prod, monoid);
            } else {
                const unsigned long long old = atomicOr(&bits[j >> 6], bit);
                if (!(old & bit)) my_new++;
            }
        });
        if (!NUMERIC) {
            if (my_new) atomicAdd(&s_cnt, my_new);
            __syncthreads();
            if (tid == 0) a.row_nnz[row] = s_cnt;
            // clear the touched words again (second walk over the same products)
            foreach_product(a, row, [&](int j, int64_t, int64_t) { bits[j >> 6] = 0ull; });
            __syncthreads();
            continue;
        }
        __syncthreads();
        // ordered sweep of the presence words: emit sorted (col, value), restore identity / zero
        const int64_t out0 = a.Tp[row];
        T *Tx = (T *)a.Tx;
        const W ident = monoid_identity<T, W>(monoid);
        for (int64_t w0 = 0; w0 < a.spa_words; w0 += MM_BLOCK) {
            const int64_t w = w0 + tid;
            unsigned long long b = (w < a.spa_words) ? bits[w] : 0ull;
            const int c = __popcll(b);
            // workgroup exclusive scan of c: wave scan by shuffles, then the 4 wave totals through LDS
            int incl = wave_inclusive_sum(c);
            if (lane == 63) s_wave[wv] = incl;
            __syncthreads();
            int wave_off = 0;
            for (int x = 0; x < wv; x++) wave_off += s_wave[x];
            int total = 0;
            for (int x = 0; x < MM_BLOCK / 64; x++) total += s_wave[x];
            int64_t o = out0 + s_base + wave_off + (incl - c);
            if (b) {
                bits[w] = 0ull;
                while (b) {
                    const int t = __ffsll(b) - 1;
                    b &= b - 1;
                    const int64_t j = w * 64 + t;
                    a.Tj[o] = (int32_t)j;
                    const T v = from_acc<T, W>(vals[j]);
                    Tx[o] = v;
                    csum_mine += checksum_term<T>(v);
                    vals[j] = ident;
                    o++;
                }
            }
            __syncthreads();
            if (tid == 0) s_base += total;
            __syncthreads();
        }
    }
    if constexpr (NUMERIC) checksum_commit(a, csum_mine);
}

// ---- LDS dense-window numeric kernel for rows of T with more than 4096 entries --------------------------------
// A^2 of a power-law graph is nearly dense per row (R-MAT 18: 4 877 entries per row on average), too many for an LDS
// hash table; a dense accumulator per workgroup in HBM thrashes (2048 live 2 MB slices).  Here the columns are cut
// into windows of MM_WIN: the workgroup of a row walks the windows in order, accumulates window w in LDS by direct
// index (native LDS atomics, no probing), emits it with an ordered bitmap sweep (so T's rows come out sorted, at a
// running offset) and recycles the accumulator -- no global atomics.  B's rows are sorted, so the part of B(k,:)
// inside window w is the range [woff[k][w], woff[k][w+1]), cached with B.
constexpr int MM_WIN = 16384;   // 128 KiB of 8-byte accumulators: one 1024-thread workgroup per CU
constexpr int MM_WIN_BLOCK = 1024;


struct DealScratch {
    int scan[MM_WIN_BLOCK + 1];
    int64_t qb[MM_WIN_BLOCK];
    int wsum[MM_WIN_BLOCK / 64];
};
__device__ __forceinline__ DealScratch &deal_scratch()
{
    __shared__ DealScratch sc;
    return sc;
}

// Products A(row,k) * B(k,j) with j inside column window w, visited by the whole 1024-thread workgroup (every thread must
// call).  The row's entries come 1024 at a time, one per thread: the thread fetches the range of B(k,:) inside the window
// from the offset table; a workgroup scan of the range lengths numbers the products, and the threads take them round-robin
// (binary search of the product number in the scan) -- a hub column with thousands of entries in the window is shared by the
// whole workgroup.  f(p, q): p = position of A(row,k), q = position of B(k,j).
// the products of up to MM_WIN_BLOCK entries of a row -- thread t brings `len` products starting at position qb of B --
// dealt round-robin to the workgroup: scan of the lengths, binary search of the product number.  f(e, q): e = the entry
// (thread) the product belongs to, q = position of B(k,j).  Every thread must call.
template <typename F>
__device__ __forceinline__ void deal_products(int len, int64_t qb, F &&f)
{
    // (one scratch area for every instantiation: a static __shared__ array inside a function template is allocated once per
    //  instantiation, and the callers instantiate this one per semiring copy of their loops)
    DealScratch &sc = deal_scratch();
    int *s_scan = sc.scan;
    int64_t *s_qb = sc.qb;
    int *s_wsum = sc.wsum;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int incl = wave_inclusive_sum(len);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int x = 0; x < MM_WIN_BLOCK / 64; x++) {
        if (x < wv) wave_off += s_wsum[x];
        total += s_wsum[x];
    }
    s_scan[tid] = wave_off + incl - len;
    s_qb[tid] = qb;
    __syncthreads();
    for (int t = tid; t < total; t += MM_WIN_BLOCK) {
        int lo = 0;  // the last entry whose first product number is <= t
#pragma unroll
        for (int st = MM_WIN_BLOCK / 2; st > 0; st >>= 1)
            if (s_scan[lo + st] <= t) lo += st;
        f(lo, s_qb[lo] + (t - s_scan[lo]));
    }
    __syncthreads();
}

// the same with the products' loads and their use apart: d = load(e, q) for DP_ILP products of a thread back to back (product
// numbers clamped into the batch instead of guarded: no branch holds a load's wait), then apply(d).  Every thread must call.
constexpr int DP_ILP = 4;
template <typename FL, typename FA>
__device__ __forceinline__ void deal_products_2(int len, int64_t qb, FL &&load, FA &&apply)
{
    DealScratch &sc = deal_scratch();
    int *s_scan = sc.scan;
    int64_t *s_qb = sc.qb;
    int *s_wsum = sc.wsum;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int incl = wave_inclusive_sum(len);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int x = 0; x < MM_WIN_BLOCK / 64; x++) {
        if (x < wv) wave_off += s_wsum[x];
        total += s_wsum[x];
    }
    s_scan[tid] = wave_off + incl - len;
    s_qb[tid] = qb - (wave_off + incl - len);  // (product t of the batch is entry s_qb[e] + t of B)
    __syncthreads();
    for (int t0 = tid; t0 < total; t0 += MM_WIN_BLOCK * DP_ILP) {
        decltype(load(0, (int64_t)0)) d[DP_ILP];
#pragma unroll
        for (int u = 0; u < DP_ILP; u++) {
            const int t = t0 + MM_WIN_BLOCK * u < total ? t0 + MM_WIN_BLOCK * u : total - 1;
            int lo = 0;  // the last entry whose first product number is <= t
#pragma unroll
            for (int st = MM_WIN_BLOCK / 2; st > 0; st >>= 1)
                if (s_scan[lo + st] <= t) lo += st;
            d[u] = load(lo, s_qb[lo] + t);
        }
#pragma unroll
        for (int u = 0; u < DP_ILP; u++)
            if (t0 + MM_WIN_BLOCK * u < total) apply(d[u]);
    }
    __syncthreads();
}

// Walks the column windows of one row with a 1024-thread workgroup; body(w, visit) is called once per window (uniformly),
// and visit(f) delivers the window's products to f(p, q) (p = position of A(row,k), q = position of B(k,j)).  The part of
// B(k,:) inside window w is [woff[k][w], woff[k][w+1]).  Rows with at most 1024 entries (all but the hubs) keep one entry
// per thread in registers for the whole walk: k and Bp[k] are loaded once, and the only load a window adds -- woff[k][w+2],
// the end of the NEXT window's range -- is issued a window ahead, so a window's dependent chain is scan -> B loads ->
// LDS atomics instead of Aj -> woff/Bp -> scan -> B loads.  Hub rows re-read their entries per window, 1024 at a time.
template <typename Body>
__device__ __forceinline__ void walk_windows(const MxmArgs &a, int64_t row, Body &&body)
{
    const int tid = threadIdx.x;
    const int64_t pbeg = a.Ap[row], pend = a.Ap[row + 1];
    const int nwin = a.n_win;
    if (pend - pbeg <= MM_WIN_BLOCK) {
        const int64_t p = pbeg + tid;
        const bool mine = p < pend;
        int64_t bp = 0;
        const int32_t *wo = nullptr;
        int o_cur = 0, o_nxt = 0;
        if (mine) {
            const int k = a.Aj[p];
            bp = a.Bp[k];
            wo = a.woff + (int64_t)k * (nwin + 1);
            o_cur = wo[0];
            o_nxt = wo[1];
        }
        for (int w = 0; w < nwin; w++) {
            const int o_ahead = (mine && w + 2 <= nwin) ? wo[w + 2] : 0;
            const int len = o_nxt - o_cur;
            const int64_t qb = bp + o_cur;
            body(w, [&](auto &&f) { deal_products(len, qb, [&](int e, int64_t q) { f(pbeg + e, q); }); });
            o_cur = o_nxt;
            o_nxt = o_ahead;
        }
    } else {
        for (int w = 0; w < nwin; w++) {
            body(w, [&](auto &&f) {
                for (int64_t pc = pbeg; pc < pend; pc += MM_WIN_BLOCK) {
                    const int64_t p = pc + tid;
                    int len = 0;
                    int64_t qb = 0;
                    if (p < pend) {
                        const int k = a.Aj[p];
                        const int32_t *o = a.woff + (int64_t)k * (nwin + 1) + w;
                        const int o0 = o[0], o1 = o[1];
                        qb = a.Bp[k] + o0;
                        len = o1 - o0;
                    }
                    deal_products(len, qb, [&](int e, int64_t q) { f(pc + e, q); });
                }
            });
        }
    }
}

template <typename T>
__global__ __launch_bounds__(MM_WIN_BLOCK) void k_spgemm_win(const MxmArgs a, const uint32_t *rows)
{
    using W = typename Widen<T>::type;
    __shared__ W s_acc[MM_WIN];
    __shared__ unsigned long long s_bits[MM_WIN / 64];
    __shared__ int s_wave[MM_WIN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    const int64_t row = rows[blockIdx.x];
    if (a.wrow && a.wrow[row] >= 0) return;  // (the row's windows are units of k_spgemm_unit)
    const W ident = monoid_identity<T, W>(monoid);
    for (int k = tid; k < MM_WIN; k += MM_WIN_BLOCK) s_acc[k] = ident;
    if (tid < MM_WIN / 64) s_bits[tid] = 0ull;
    __syncthreads();
    int64_t out = a.Tp[row];
    T *Tx = (T *)a.Tx;
    unsigned long long csum_mine = 0;
    walk_windows(a, row, [&](int w, auto &&visit) {
        const int c0 = w * MM_WIN;
        visit([&](int64_t p, int64_t q) {
            const int j = a.Bj[q] - c0;
            const T av = a.need_a ? Ax[a.a_iso ? 0 : p] : (T)0;
            const T bv = a.need_b ? Bx[a.b_iso ? 0 : q] : (T)0;
            const W prod = (W)apply_binop<T>(mult, av, bv);
            if (monoid == OP_ANY) s_acc[j] = prod;
            else atomic_combine<W>(&s_acc[j], prod, monoid);
            atomicOr(&s_bits[j >> 6], 1ull << (j & 63));
        });
        __syncthreads();
        if (a.CMp) {  // fused complemented mask: forbidden columns give their accumulators back and are not emitted
            const int32_t *mo = a.cm_woff + row * (a.n_win + 1) + w;
            const int m0 = mo[0], m1 = mo[1];
            const int64_t mb = a.CMp[row];
            for (int i = m0 + tid; i < m1; i += MM_WIN_BLOCK) {
                const int j = a.CMj[mb + i] - c0;
                atomicAnd(&s_bits[j >> 6], ~(1ull << (j & 63)));
                s_acc[j] = ident;
            }
            __syncthreads();
        }
        // ordered sweep of the window's presence words (MM_WIN/64 = 256 words: threads 0..255 take one each)
        unsigned long long b = (tid < MM_WIN / 64) ? s_bits[tid] : 0ull;
        const int c = __popcll(b);
        int incl = wave_inclusive_sum(c);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int wave_off = 0, total = 0;
        for (int x = 0; x < MM_WIN_BLOCK / 64; x++) {
            if (x < wv) wave_off += s_wave[x];
            total += s_wave[x];
        }
        int64_t o = out + wave_off + (incl - c);
        if (b) {
            s_bits[tid] = 0ull;
            while (b) {
                const int t = __ffsll(b) - 1;
                b &= b - 1;
                const int j = tid * 64 + t;
                a.Tj[o] = c0 + j;
                const T v = from_acc<T, W>(s_acc[j]);
                Tx[o] = v;
                csum_mine += checksum_term<T>(v);
                s_acc[j] = ident;
                o++;
            }
        }
        out += total;
        __syncthreads();
    });
    checksum_commit(a, csum_mine);
}

// ---------------------------------------------------------------------------------------------------
// Heavy rows as (row, column window) WORK UNITS (round 2; replaces k_spgemm_sym_lds / k_spgemm_win where the window offsets
// are at hand).  The 1024-thread window kernel above walks the windows of a row one after the other, with workgroup barriers
// and scans around a few hundred products per window; here every (row, window) pair is a unit of its own, so the windows of
// one row run side by side, and nothing heavier than a wavefront synchronises for the small ones.  A unit:
//   pass A  every product sets its column's bit in the unit's 2 KiB LDS bitmap;
//   count   popcounts per word + a wavefront scan: the unit's entry count and, per word, how many set columns precede it.
//           The symbolic kernel stops here and stores the count (k_unit_prefix turns the counts of a row into offsets);
//   pass B  (numeric) every product finds its RANK among the unit's columns (word prefix + popcount below its bit) and
//           combines into a COMPACT accumulator array -- rank order is column order: no hash, no sort;  units with more
//           distinct columns than accumulators take ceil(count / CAP) passes over their products;
//   emit    columns from the bitmap, values from the accumulators, at the unit's offset inside the row: rows come out sorted.
// Products are dealt round-robin to the lanes (adjacent lanes read adjacent entries of a row of B) with a binary search in
// the wavefront's scan of the B-range lengths.  WPU = wavefronts per unit: 1 (four units per workgroup, 512 accumulators
// each) for units of up to 512 entries -- and every unit of the symbolic pass --, 4 (one unit per workgroup, 4096 accumulators,
// the entries of A dealt to the wavefronts 64 at a time) for the denser ones.
// ---------------------------------------------------------------------------------------------------
#ifdef GRB_ABLATE
#define MXM_ABL(a, bit) (((a).abl & (bit)) != 0)
#else
#define MXM_ABL(a, bit) false
#endif
#ifndef GRB_MU_ILP
#define GRB_MU_ILP 4
#endif
#ifndef GRB_MU_M2_WPU
#define GRB_MU_M2_WPU 4
#endif
#ifndef GRB_MU_SEG
#define GRB_MU_SEG 2048
#endif
constexpr int MU_SEG = GRB_MU_SEG;  // products per segment of the rank dealing (a multiple of 64 MU_ILP, at most 4096)
constexpr int MU_SYM_PLEN = 128;  // rows of A up to this long (the batches a wavefront keeps in registers) may share a symbolic unit between windows
constexpr int MU_ILP = GRB_MU_ILP;  // products a lane has in flight
constexpr int MU_SYMBOLIC = 0, MU_NUMERIC = 1, MU_MASKED = 2;  // what a unit kernel does
// a unit of the numeric / masked pass as the classification writes it: everything the kernel needs to start on the entries of
// A's row (one 32-byte load instead of the chain unit -> row -> slot -> offsets)
struct alignas(16) UnitRec {
    int64_t out;     // numeric: Tp[row] + the unit's offset inside its row;  masked: position of the unit's first mask entry
    int64_t pbeg;    // the row of A: first entry ...
    int32_t plen;    // ... and length
    uint32_t row;
    int32_t aux;     // numeric: the unit's bitmap in the pool or -1;  masked: mask entries inside the window
    int32_t w;       // the window
};
static_assert(sizeof(UnitRec) == 32, "UnitRec is two 16-byte loads");
constexpr int MU_POOLS = 1024;  // sub-pools of the bitmap pool
#ifndef GRB_MU_SMALL
#define GRB_MU_SMALL 512
#endif
constexpr int MU_SMALL = GRB_MU_SMALL;  // entries of a unit a single wavefront accumulates
// Units per wavefront of the numeric one-wavefront class (round 6, VERDICT r03-r05 "two independent units interleaved per wavefront").
// A unit is a chain of dependent round trips -- its record, the entries of A, their window offsets and row pointers of B, the
// bitmap, then the products -- and LDS holds the workgroups per CU at four, i.e. four chains per SIMD.  With GRB_MU_UPW = 2 a
// wavefront owns two consecutive units of the class list (neighbouring windows of one row, as a rule): both records and both
// offset chains are requested BEFORE the first unit's bitmap and products are touched, so the second unit's chain travels
// under the first unit's work; the units then run one after the other in the same LDS (no more LDS, +14 registers).
// MEASURED (profiles/r06/mxm_units_per_wavefront.txt, INT64 A (+.x) A, ms per product, UPW = 1 / 2 / 3): scale 20 (single windows)
// 130.0 / 134.2 / 133.3; scale 22 (window pairs) 1292.8 / 1270.7 / 1265.4, run-to-run noise +-0.5 %.  The offset chain is not what a unit
// waits for (round 3 found the same for the symbolic units); a unit's TRIPS are, and two units' trips in flight at once need two
// bitmaps and two accumulator sets -- the LDS that already holds the class at four workgroups per CU.  Default: 1 (2 from window
// groups on would buy 1.7 % at scale 22 and cost 3 % wherever single windows run).
#ifndef GRB_MU_UPW
#define GRB_MU_UPW 1
#endif
#ifndef GRB_MU_PIPE
#define GRB_MU_PIPE 0  // bit 0: the trips of a batch of the SEARCH dealing (one-wavefront numeric class) are software-pipelined (round 6), bit 1: the trips
                       // of a segment of the RANKED dealing (every other class); 0: load, wait, apply per trip (rounds 2-5).  See the measurement at the loops.
#endif
constexpr int mu_units_per_wave(int mode, int wpu) { return (mode == 1 && wpu == 1) ? GRB_MU_UPW : 1; }

__device__ __forceinline__ void mw_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// F (round 5) = windows per unit: the unit covers the columns [w F MM_WIN, (w + 1) F MM_WIN) -- `w` counts GROUPS of F windows --, its
// bitmap holds F MM_WIN bits, its ranges of B run from window offset w F to (w + 1) F of the per-window table.  The symbolic unit
// still reports one count per WINDOW (and keeps its bitmap as F consecutive window bitmaps of the pool): the classification sends
// a group to one numeric unit of the same F, or -- beyond the densest compact class -- its windows to F = 1 units, as before.
// (wavefronts per SIMD the register allocation must leave room for: the 1024 class of GROUP units (F = 2) stood at 99 registers = four
//  waves per SIMD; asked for five it compiles to 95 without spills: scale 22 1317 -> 1290 ms, profiles/r05/mxm_lds_occupancy.txt -- shipped
//  in round 6 after the whole GPU tier ran on it.  F = 4 spills 23 registers under the same hint: not hinted.)
constexpr int mu_min_waves(int mode, int wpu, int cap, int f) { return (mode == 1 && wpu == 4 && cap == 1024 && f == 2) ? 5 : 1; }
template <typename T, int MODE, int WPU, int CAP, int F = 1>
__global__ __launch_bounds__(64 * (WPU > 4 ? WPU : 4), mu_min_waves(MODE, WPU, CAP, F)) void k_spgemm_unit(const MxmArgs a, const uint32_t *rows, int64_t ridx0,
                                                                          int64_t nrows_here, const UnitRec *units, int64_t nunits)
{
    using W = typename Widen<T>::type;
    constexpr bool NUMERIC = MODE != MU_SYMBOLIC, MASKED = MODE == MU_MASKED;
    static_assert(F == 1 || F == 2 || F == 4 || F == 8, "window groups of 1, 2, 4 or 8");
    static_assert(F == 1 || !MASKED, "the mask-driven units walk single windows");
    constexpr int WIN = MM_WIN * F, WORDS = WIN / 64, WPL = WORDS / 64, FWORDS = MM_WIN / 64;
    constexpr int WAVES = WPU > 4 ? WPU : 4;  // wavefronts per workgroup
    constexpr int UPB = WAVES / WPU;          // units per workgroup
    __shared__ unsigned long long s_bits[UPB][WORDS];
    // (the words' prefix counts as 16-bit numbers where the unit's bitmap holds at most 65536 bits: what decides how many workgroups a CU holds
    //  is LDS -- with groups of two windows the one-wavefront class took 44032 bytes, three workgroups per CU; 39936 are four.  Last session
    //  of round 5, profiles/r05/mxm_lds_occupancy.txt)
    using WPre = typename std::conditional<(WIN <= 65536), unsigned short, int>::type;
    __shared__ WPre s_wpre[NUMERIC ? UPB : 1][NUMERIC ? WORDS : 1];
    // how a wavefront finds the entry of A a product belongs to: by rank in a bitmap of first product numbers, or -- the numeric
    // one-wavefront class, measured 3 ms faster with it -- by binary search in the scan of the range lengths (round 2)
    constexpr bool SEARCH_DEAL = NUMERIC && WPU == 1;
    __shared__ unsigned long long s_recm[WAVES][SEARCH_DEAL ? 1 : MU_SEG / 64];
    __shared__ short s_recb[WAVES][SEARCH_DEAL ? 1 : MU_SEG / 64];
    __shared__ const int32_t *s_cptr[WAVES][SEARCH_DEAL ? 1 : 64];
    __shared__ unsigned char s_clane[WAVES][SEARCH_DEAL ? 1 : 64];
    __shared__ int s_scan[WAVES][SEARCH_DEAL ? 64 : 1];
    __shared__ int64_t s_qb[WAVES][SEARCH_DEAL ? 64 : 1];
    __shared__ W s_acc[NUMERIC ? UPB : 1][NUMERIC ? CAP : 1];
    __shared__ unsigned long long s_hit[MASKED ? UPB : 1][MASKED ? (CAP + 63) / 64 : 1];  // (masked: which accumulators received a product)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int uib = WPU == 1 ? wave : 0, sub = WPU == 1 ? 0 : wave;  // unit inside the workgroup, wavefront inside the unit
    const int nwin = a.n_win;                   // (windows: the tables' unit)
    const int ngroups_w = (nwin + F - 1) / F;   // (groups of F windows: the units' unit)
    constexpr int UPW = mu_units_per_wave(MODE, WPU);  // units this wavefront owns (numeric one-wavefront class: GRB_MU_UPW)
    const int64_t unit = (xcd_block(a) * UPB + uib) * UPW;
    int64_t ridx = 0, row, out = 0, pbeg, pend;
    int w, w_end = 0, bslot = -1, mcnt = 0;
    constexpr int NB = 2;  // batches of entries of A a wavefront keeps (range of B inside the window) from pass A for pass B
    UnitRec urec[UPW];
    int pf_len[UPW][NB];     // (UPW > 1: the ranges of B of every owned unit's first NB batches, requested up front)
    int64_t pf_qb[UPW][NB];
    if constexpr (NUMERIC) {  // a unit of the class list
        if (unit >= nunits) return;  // (uniform over the unit's threads)
#pragma unroll
        for (int k = 0; k < UPW; k++) urec[k] = units[unit + k < nunits ? unit + k : nunits - 1];  // (past the list: the last unit again, never run)
        if constexpr (UPW > 1) {
            // both offset chains, stage by stage: all entries of A first, then everything that depends on them -- the loads of the
            // second unit leave with the first unit's instead of behind its products
            int pk[UPW][NB];
            bool pok[UPW][NB];
#pragma unroll
            for (int k = 0; k < UPW; k++)
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const int64_t pe = urec[k].pbeg + urec[k].plen, pp = urec[k].pbeg + (int64_t)b * 64 + (int64_t)lane;
                    pok[k][b] = pp < pe;
                    pk[k][b] = a.Aj[pok[k][b] ? pp : pe - 1];
                }
#pragma unroll
            for (int k = 0; k < UPW; k++)
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const int wk = urec[k].w;
                    const int fs = a.n_win - wk * F < F ? a.n_win - wk * F : F;
                    const int32_t *o = a.woff + (int64_t)pk[k][b] * (a.n_win + 1) + wk * F;
                    const int o0 = o[0], o1 = o[fs];
                    pf_qb[k][b] = a.Bp[pk[k][b]] + o0;
                    pf_len[k][b] = pok[k][b] ? o1 - o0 : 0;
                }
        }
        const UnitRec &r = urec[0];
        row = r.row;
        w = r.w;
        out = r.out;
        pbeg = r.pbeg;
        pend = pbeg + r.plen;
        if constexpr (MASKED) mcnt = r.aux;  // (the unit's mask entries are Mj[out .. out + mcnt))
        else bslot = r.aux;
    } else {  // every (row, window) of the rows [ridx0, ridx0 + nrows_here) of the bin
        const int wg = a.sym_wg, ngrp = (ngroups_w + wg - 1) / wg;
        if (unit >= nrows_here * ngrp) return;
        ridx = ridx0 + unit / ngrp;
        if (a.sym_list) ridx = a.sym_list[ridx];
        w = (int)(unit % ngrp) * wg;
        w_end = w + wg < ngroups_w ? w + wg : ngroups_w;
        row = rows[ridx];
        pbeg = a.Ap[row];
        pend = a.Ap[row + 1];
        if (a.sym_plen_max > 0 && pend - pbeg > a.sym_plen_max) return;  // (a row of the list launch)
    }
    auto usync = [&]() {
        if constexpr (WPU == 1) mw_sync();
        else __syncthreads();
    };
    unsigned long long *bits = s_bits[uib];
    WPre *wpre = s_wpre[NUMERIC ? uib : 0];
    unsigned long long *recm = s_recm[wave];
    short *recb = s_recb[wave];
    int *scan = s_scan[wave];
    int64_t *sqb = s_qb[wave];
    const int32_t **cptr = s_cptr[wave];
    unsigned char *clane = s_clane[wave];
    W *acc = s_acc[NUMERIC ? uib : 0];
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    const W ident = monoid_identity<T, W>(monoid);
    // (an iso operand stores one value: read once, not once per product)
    T a_iso_val = (T)0, b_iso_val = (T)0;
    if constexpr (NUMERIC) {
        if (a.need_a && a.a_iso) a_iso_val = Ax[0];
        if (a.need_b && a.b_iso) b_iso_val = Bx[0];
    }
    const int tiu = sub * 64 + lane;  // thread inside the unit
    unsigned long long csum_mine = 0;  // (streamed product: the values this thread stores)
    // (the units this wavefront owns, one instance of the body per unit -- `uk` is a compile-time constant: a run-time loop over them was
    //  left un-unrolled by the compiler, and the prefetched ranges, indexed by `uk`, went to scratch memory; UPW = 1 everywhere but in
    //  the numeric one-wavefront class)
    auto run_unit = [&](auto uk_c) {
    constexpr int uk = decltype(uk_c)::value;
    if constexpr (UPW > 1) {
        if (uk > 0) {
            if (unit + uk >= nunits) return;  // (uniform over the wavefront)
            usync();  // (the next unit clears / loads the bitmap and the accumulators the last one wrote its columns from)
            const UnitRec &r = urec[uk];
            row = r.row;
            w = r.w;
            out = r.out;
            pbeg = r.pbeg;
            pend = pbeg + r.plen;
            bslot = r.aux;
        }
    }
    for (;;) {  // (the windows of a symbolic unit; numeric and masked units: once)
    const int c0 = w * WIN;
    const int fspan = nwin - w * F < F ? nwin - w * F : F;  // windows of the group that exist (the last group of a row may be short)
    // the ranges of B inside the window for the first NB batches of the wavefront's entries of A: requested before anything
    // else (two dependent round trips that overlap the bitmap load / clear and the accumulator fill), kept for every pass
    int c_len[NB];
    int64_t c_qb[NB];
    // (unguarded: an entry past the row's end re-reads the last one and brings length 0 -- a guarded chain of loads is waited
    //  for inside its branch, and the NB chains below would run one after the other)
    auto fetch = [&](int64_t p, int &len, int64_t &qb) {
        const bool ok = p < pend;
        const int k = a.Aj[ok ? p : pend - 1];
        const int32_t *o = a.woff + (int64_t)k * (nwin + 1) + w * F;
        const int o0 = o[0], o1 = o[fspan];
        qb = a.Bp[k] + o0;
        len = ok ? o1 - o0 : 0;
    };
    if constexpr (UPW > 1) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            c_len[b] = pf_len[uk][b];
            c_qb[b] = pf_qb[uk][b];
        }
    } else {
#pragma unroll
        for (int b = 0; b < NB; b++) fetch(pbeg + sub + (int64_t)b * 64 * WPU + (int64_t)lane * WPU, c_len[b], c_qb[b]);
    }
    if (bslot >= 0) {  // (numeric pass: the symbolic pass kept the unit's bitmap)
        for (int k = tiu; k < WORDS; k += 64 * WPU) bits[k] = a.bm_pool[(int64_t)bslot * FWORDS + k];  // (F consecutive window bitmaps)
    } else {
        for (int k = tiu; k < WORDS; k += 64 * WPU) bits[k] = 0ull;
    }
    usync();
    if constexpr (MASKED) {  // the bitmap is the mask row's part inside the window: products outside it are dropped
        for (int i = tiu; i < mcnt; i += 64 * WPU) {
            const int j = a.Mj[out + i] - c0;
            atomicOr(&bits[j >> 6], 1ull << (j & 63));
        }
        usync();
    }
    // every product of the row inside the window: d = load(p, q), then apply(d) -- MU_ILP products per lane at a time, their loads
    // issued before the first apply (the LDS atomics would otherwise serialise the global load latencies).  The wavefronts of
    // a unit take the entries of A 64 at a time.
    const unsigned long long lane_le = (2ull << lane) - 1ull;  // bits 0 .. lane
    int64_t pc_of_batch = 0;  // the entry of A lane 0 holds in the batch being dealt (for `load`: entry of A = this + lane WPU)
    auto visit = [&](auto &&load, auto &&apply) {
        // one batch: lane l holds the range [qb, qb + len) of B for the entry pc + l WPU of A; a wavefront scan of the lengths
        // numbers the products, and the lanes take them round-robin (binary search of the product number in the scan).
        // (A search-free dealing -- ranges of 16+ entries walked by the whole wavefront four at a time, shorter ones by their own
        //  lane -- measured 20 % SLOWER, 275 against 225 ms at scale 20: fewer products in flight per round trip.)
        auto process = [&](int64_t pc, int len, int64_t qb) {
            constexpr int ILP = MU_ILP;
            static_assert(MU_SEG % (64 * ILP) == 0 && MU_SEG / 64 <= 64 && (MU_SEG & (MU_SEG - 1)) == 0, "segments hold whole trips, one word per lane");
            const int incl = wave_inclusive_sum(len);
            const int total = __builtin_amdgcn_readlane(incl, 63);  // (uniform: loop bounds stay scalar)
            if (total == 0 || MXM_ABL(a, 128)) return;  // (timing switch 128: no products at all)
            const int start = incl - len;  // the number of the entry's first product
            pc_of_batch = pc;
            if constexpr (SEARCH_DEAL) {
                scan[lane] = start;
                sqb[lane] = qb - start;  // (product t of the batch is entry sqb[e] + t of B, e = the lane that brought it)
                mw_sync();
                // ILP products per lane and TRIP: their searches interleave and their loads are issued back to back -- every product number
                // is clamped into the batch instead of being guarded by a branch (a guarded load is waited for inside its branch).
                // Round 6 (GRB_MU_PIPE): the trips of a batch are SOFTWARE-PIPELINED -- the loads of trip k + 1 are issued before the
                // products of trip k go to LDS, through two register sets with static roles (no copies: a copied register waits for its
                // load), and only trips that exist are ever requested (the loop runs while two more trips follow; the last one or two
                // are peeled).  A unit's wavefront used to wait one global round trip per trip with nothing else of its own in flight.
                if constexpr ((GRB_MU_PIPE & 1) != 0) {
                using D = decltype(load((const int32_t *)nullptr, 0u, 0));
                constexpr int TRIP = 64 * ILP;
                auto fetch_trip = [&](int base, D (&d)[ILP]) {
#pragma unroll
                    for (int u = 0; u < ILP; u++) {
                        const int tt = base + lane + 64 * u;
                        const int t = tt < total ? tt : total - 1;
                        int lo = 0;  // the last entry whose first product number is <= t (scan[0] = 0 <= t): six steps, three VALU each
#pragma unroll
                        for (int st = 32; st > 0; st >>= 1)
                            if (scan[lo + st] <= t) lo += st;
                        d[u] = load(a.Bj + sqb[lo], (unsigned)t, lo);
                    }
                };
                auto apply_trip = [&](int base, const D (&d)[ILP]) {
#pragma unroll
                    for (int u = 0; u < ILP; u++)
                        if (base + lane + 64 * u < total) apply(d[u]);
                };
                D da[ILP], db[ILP];
                int base = 0;
                fetch_trip(0, da);
                while (base + 2 * TRIP < total) {  // (uniform: two more trips follow the one in `da`)
                    fetch_trip(base + TRIP, db);
                    apply_trip(base, da);
                    fetch_trip(base + 2 * TRIP, da);
                    apply_trip(base + TRIP, db);
                    base += 2 * TRIP;
                }
                if (base + TRIP < total) {
                    fetch_trip(base + TRIP, db);
                    apply_trip(base, da);
                    apply_trip(base + TRIP, db);
                } else {
                    apply_trip(base, da);
                }
                } else {
                for (int t0 = lane; t0 < total; t0 += 64 * ILP) {  // (rounds 2-5: load, wait, apply per trip)
                    decltype(load((const int32_t *)nullptr, 0u, 0)) d[ILP];
#pragma unroll
                    for (int u = 0; u < ILP; u++) {
                        const int t = t0 + 64 * u < total ? t0 + 64 * u : total - 1;
                        int lo = 0;  // the last entry whose first product number is <= t (scan[0] = 0 <= t): six steps, three VALU each
#pragma unroll
                        for (int st = 32; st > 0; st >>= 1)
                            if (scan[lo + st] <= t) lo += st;
                        d[u] = load(a.Bj + sqb[lo], (unsigned)t, lo);
                    }
#pragma unroll
                    for (int u = 0; u < ILP; u++)
                        if (t0 + 64 * u < total) apply(d[u]);
                }
                }
                mw_sync();
                return;
            }
            // The owner of a product by RANK in a bitmap of the entries' first product numbers (the trick the unit's columns are
            // ranked with).  Round 2 searched the product number in the scan: six dependent LDS reads and 18 vector instructions
            // per product -- and the instruction counters of round 3 (profiles/r03/pmc_mxm_issue.txt) show what the units are
            // bound by: ONE instruction of any kind per SIMD and 4 cycles (23.5 G instructions x 4 / (1024 SIMDs x 2.4 GHz) = 38 ms
            // for a 37.8 ms symbolic pass), so the product loop is written for instruction count.  Per batch the non-empty entries
            // are compacted (cptr[c] = where product 0 of the batch would sit in B's column array if the entry's range started
            // there, clane[c] = the entry's lane); per segment of MU_SEG products every non-empty entry sets the bit of its first
            // product number (one LDS atomic per ENTRY), a scan of the words' popcounts gives every word the rank of the last
            // entry that starts before it; lane l of group g then takes product 64 g + l, which belongs to entry
            // rec[g].b + popcount(rec[g].m up to bit l) of the compacted list: one 16-byte LDS read with an immediate offset, two
            // ANDs, two popcounts, one pointer read, one 64-bit shift-add.
            const unsigned long long have = __ballot(len > 0);
            if (len > 0) {
                const int c = __popcll(have & (lane_le >> 1));
                cptr[c] = a.Bj + (qb - start);
                clane[c] = (unsigned char)lane;
            }
            for (int seg0 = 0; seg0 < total; seg0 += MU_SEG) {
                const int seg_n = total - seg0 < MU_SEG ? total - seg0 : MU_SEG;  // products of the segment
                if (lane < MU_SEG / 64) recm[lane] = 0ull;
                mw_sync();
                const int rel = start - seg0;
                if (len > 0 && rel >= 0 && rel < MU_SEG) atomicOr(&recm[rel >> 6], 1ull << (rel & 63));
                mw_sync();
                const int c = lane < MU_SEG / 64 ? __popcll(recm[lane]) : 0;
                const int ci = wave_inclusive_sum(c);
                const int before = __popcll(__ballot(len > 0 && rel < 0));  // entries that start before the segment
                // (words past the segment's last group hold no bit and the rank of the batch's last entry: a group number needs no clamp)
                if (lane < MU_SEG / 64) recb[lane] = (short)(before + ci - c - 1);
                mw_sync();
                // ILP groups per trip, their loads issued back to back: a product number past the batch is clamped to the last
                // product instead of being guarded by a branch (a guarded load is waited for inside its branch); `load` only loads,
                // whatever depends on the loaded values happens in `apply`, which is guarded.
                const int g_n = (seg_n + 63) >> 6;
                const unsigned long long *rpm = recm;
                const short *rpb = recb;
                unsigned t = (unsigned)(seg0 + lane);
                const unsigned t_last = (unsigned)(total - 1);
                // (round 6, GRB_MU_PIPE: the trips of a segment software-pipelined as in the search dealing above -- the loads of trip k + 1
                //  leave before the products of trip k go to LDS; two register sets with static roles; only trips that exist are requested)
                if constexpr ((GRB_MU_PIPE & 2) != 0) {
                using D = decltype(load((const int32_t *)nullptr, 0u, 0));
                auto fetch_grp = [&](int g0, D (&d)[ILP]) {
#pragma unroll
                    for (int u = 0; u < ILP; u++) {
                        const int rank = (int)rpb[g0 + u] + __popcll(rpm[g0 + u] & lane_le);
                        const unsigned tt = t + 64u * (unsigned)(g0 + u);
                        const unsigned tu = tt < t_last ? tt : t_last;
                        d[u] = load(cptr[rank], tu, rank);
                    }
                };
                auto apply_grp = [&](int g0, const D (&d)[ILP]) {
#pragma unroll
                    for (int u = 0; u < ILP; u++)
                        if (t + 64u * (unsigned)(g0 + u) <= t_last) apply(d[u]);
                };
                D da[ILP], db[ILP];
                int g0 = 0;
                fetch_grp(0, da);
                while (g0 + 2 * ILP < g_n) {  // (uniform: two more trips follow the one in `da`)
                    fetch_grp(g0 + ILP, db);
                    apply_grp(g0, da);
                    fetch_grp(g0 + 2 * ILP, da);
                    apply_grp(g0 + ILP, db);
                    g0 += 2 * ILP;
                }
                if (g0 + ILP < g_n) {
                    fetch_grp(g0 + ILP, db);
                    apply_grp(g0, da);
                    apply_grp(g0 + ILP, db);
                } else {
                    apply_grp(g0, da);
                }
                } else {
                for (int g0 = 0; g0 < g_n; g0 += ILP, rpm += ILP, rpb += ILP, t += 64 * ILP) {  // (rounds 3-5: load, wait, apply per trip)
                    decltype(load((const int32_t *)nullptr, 0u, 0)) d[ILP];
#pragma unroll
                    for (int u = 0; u < ILP; u++) {
                        const int rank = (int)rpb[u] + __popcll(rpm[u] & lane_le);
                        const unsigned tu = t + 64u * u < t_last ? t + 64u * u : t_last;
                        d[u] = load(cptr[rank], tu, rank);
                    }
#pragma unroll
                    for (int u = 0; u < ILP; u++)
                        if (t + 64u * u <= t_last) apply(d[u]);
                }
                }
                mw_sync();
            }
        };
        // wavefront s of the unit takes the entries s, s + WPU, s + 2 WPU, ... of the row (64 of them per batch): dealt in
        // blocks of 64, the first wavefront would own the smallest k -- on a graph numbered by degree, the hub rows of B
        int64_t pc = pbeg + sub;  // the batch's first entry (lane 0's)
#pragma unroll
        for (int b = 0; b < NB; b++, pc += 64 * WPU) {
            if (pc >= pend) return;  // (wave-uniform)
            process(pc, c_len[b], c_qb[b]);
        }
        for (; pc < pend; pc += 64 * WPU) {
            int len;
            int64_t qb;
            fetch(pc + (int64_t)lane * WPU, len, qb);
            process(pc, len, qb);
        }
    };
    // ---- pass A: which columns of the window does the row reach
    if (!MASKED && bslot < 0 && !(NUMERIC && MXM_ABL(a, 16)))
        visit([&](const int32_t *from, unsigned t, int) { return MXM_ABL(a, 256) ? c0 + (int)((t * 37u) & (WIN - 1)) : load_global_i32(from + t); },  // (256: no load of B)
              [&](int jraw) {
                  if (MXM_ABL(a, 64)) return;  // (64: no bitmap atomics)
                  const int j = jraw - c0;
                  unsigned *wp = (unsigned *)bits + (j >> 5);  // (32-bit words: one shift, no 64-bit mask to build)
                  const unsigned bit = 1u << (j & 31);
                  atomicOr(wp, bit);
              });
    // (the clamped duplicates of a trip's last products stay guarded although setting a bit twice changes nothing: unguarded, up to
    //  255 lanes hit the same bitmap word and the LDS atomics serialise -- symbolic pass 34.6 -> 41.5 ms)
    usync();
    // fused complemented mask: the forbidden columns inside the window leave the bitmap (a bitmap kept by the symbolic pass
    // is already without them); pass B drops the products that find their column's bit clear
    const bool cmask = !MASKED && a.CMp != nullptr;
    if (cmask && bslot < 0) {
        const int32_t *mo = a.cm_woff + (int64_t)row * (nwin + 1) + w * F;
        const int m0 = mo[0], m1 = mo[fspan];
        if (m1 > m0) {
            const int64_t mb = a.CMp[row];
            for (int i = m0 + tiu; i < m1; i += 64 * WPU) {
                const int j = a.CMj[mb + i] - c0;
                atomicAnd(&bits[j >> 6], ~(1ull << (j & 63)));
            }
        }
        usync();
    }
    // ---- counts: lane l looks at words 4 l .. 4 l + 3 (every wavefront of the unit computes the same numbers)
    unsigned long long mine[WPL];
    int c = 0;
#pragma unroll
    for (int x = 0; x < WPL; x++) {
        mine[x] = bits[lane * WPL + x];
        c += __popcll(mine[x]);
    }
    const int incl = wave_inclusive_sum(c);
    const int cnt = __builtin_amdgcn_readlane(incl, 63);
    if constexpr (!NUMERIC) {
        // one count per WINDOW of the group: window f's words belong to the lanes [f 64 / F, (f + 1) 64 / F)
        constexpr int LPF = 64 / F;
        int cf = 0;
#pragma unroll
        for (int f = 0; f < F; f++) {
            const int upto = __builtin_amdgcn_readlane(incl, (f + 1) * LPF - 1);
            const int before = f ? __builtin_amdgcn_readlane(incl, f ? f * LPF - 1 : 0) : 0;
            if (lane == f) cf = upto - before;
        }
        if (lane < fspan) a.wcnt[ridx * (nwin + 1) + w * F + lane] = cf;
        if (a.wbm) {
            int slot = -1;
            if (cnt > a.bm_min_cnt) {
                if (lane == 0) {  // (MU_POOLS sub-pools, a cursor each on its own 128-byte line: one cursor serialises millions of atomics)
                    const int sp = (int)(blockIdx.x & (a.bm_pools - 1));
                    const unsigned long long got = atomicAdd(&a.bm_cursor[sp * 16], (unsigned long long)F);
                    slot = got + F <= (unsigned long long)a.bm_cap ? (int)(sp * a.bm_cap + (int64_t)got) : -1;
                }
                slot = __shfl(slot, 0);
                if (slot >= 0) {  // (F window bitmaps, one after the other)
#pragma unroll
                    for (int x = 0; x < WPL; x++) a.bm_pool[(int64_t)slot * FWORDS + lane * WPL + x] = mine[x];
                }
            }
            if (lane < fspan) a.wbm[ridx * nwin + w * F + lane] = slot >= 0 ? slot + lane : -1;
        }
    } else {
        if (sub == 0) {
            int pre = incl - c;
#pragma unroll
            for (int x = 0; x < WPL; x++) {
                wpre[lane * WPL + x] = (WPre)pre;
                pre += __popcll(mine[x]);
            }
        }
        usync();
        // ---- pass B: values into the compact accumulators, CAP ranks at a time
        T *Tx = (T *)a.Tx;
        for (int r0 = 0; r0 < cnt; r0 += CAP) {
            const int here = cnt - r0 < CAP ? cnt - r0 : CAP;
            for (int i = tiu; i < here; i += 64 * WPU) acc[i] = ident;
            if constexpr (MASKED)
                for (int i = tiu; i < (CAP + 63) / 64; i += 64 * WPU) s_hit[uib][i] = 0ull;
            usync();
            struct Prod {  // what a product loads: the column of B's entry and the two values (the arithmetic waits until `apply`)
                int j;
                T av, bv;
            };
            // One copy of the product loop per common semiring (MU / MO = the multiply / monoid opcodes as constants, -1 = the
            // operands of the call): with the opcodes as run-time values every product step walked two trees of scalar compares
            // and branches (9 600 lines of ISA for this kernel, SALU instructions = half the VALU count in the PMC pass).
            auto pass_b = [&](auto mu_c, auto mo_c) {
                constexpr int MU = decltype(mu_c)::value, MO = decltype(mo_c)::value;
                const int mult_ = MU >= 0 ? MU : mult, monoid_ = MO >= 0 ? MO : monoid;
                auto load_b = [&](const int32_t *from, unsigned t, int rank) {
                    Prod r;
                    r.j = load_global_i32(from + t);
                    r.av = (T)0;
                    r.bv = (T)0;
                    if constexpr (MU != OP_PAIR) {
                        if (a.need_a) r.av = a.a_iso ? a_iso_val : Ax[pc_of_batch + (int64_t)(SEARCH_DEAL ? rank : (int)clane[rank]) * WPU];  // (search: the rank is the lane)
                        if (a.need_b) r.bv = a.b_iso ? b_iso_val : Bx[(from - a.Bj) + (int64_t)t];
                    }
                    return r;
                };
                auto apply_b = [&](const Prod &d) {
                    const int j = d.j - c0;
                    const unsigned long long word = bits[j >> 6];
                    if ((MASKED || cmask) && !((word >> (j & 63)) & 1ull)) return;
                    const int rank = (int)wpre[j >> 6] + __popcll(word & ((1ull << (j & 63)) - 1ull)) - r0;
                    if ((unsigned)rank < (unsigned)CAP && !MXM_ABL(a, 4)) {
                        const W v = (W)apply_binop<T>(mult_, d.av, d.bv);
                        if (monoid_ == OP_ANY) acc[rank] = v;
                        else atomic_combine<W>(&acc[rank], v, monoid_);
                        if constexpr (MASKED) atomicOr(&s_hit[uib][rank >> 6], 1ull << (rank & 63));
                    }
                };
                visit(load_b, apply_b);
            };
            if (!MXM_ABL(a, 8)) {
                using std::integral_constant;
                if (mult == OP_TIMES && monoid == OP_PLUS) pass_b(integral_constant<int, OP_TIMES>{}, integral_constant<int, OP_PLUS>{});
                else if (mult == OP_PLUS && monoid == OP_MIN) pass_b(integral_constant<int, OP_PLUS>{}, integral_constant<int, OP_MIN>{});
                else if (mult == OP_PAIR && monoid == OP_PLUS) pass_b(integral_constant<int, OP_PAIR>{}, integral_constant<int, OP_PLUS>{});
                else if (mult == OP_PAIR && monoid == OP_ANY) pass_b(integral_constant<int, OP_PAIR>{}, integral_constant<int, OP_ANY>{});
                else if (mult == OP_LAND && monoid == OP_LOR) pass_b(integral_constant<int, OP_LAND>{}, integral_constant<int, OP_LOR>{});
                else pass_b(integral_constant<int, -1>{}, integral_constant<int, -1>{});
            }
            usync();
            if constexpr (MASKED) {  // per mask entry: hit or not, and the value
                T *cv = (T *)a.cap_val;
                for (int i = tiu; i < here; i += 64 * WPU) {
                    const bool hit = (s_hit[uib][i >> 6] >> (i & 63)) & 1ull;
                    a.cap_hit[out + r0 + i] = hit ? 1 : 0;
                    if (hit) cv[out + r0 + i] = from_acc<T, W>(acc[i]);
                }
            } else if (!MXM_ABL(a, 2)) {
                for (int i = tiu; i < here; i += 64 * WPU) {
                    const T v = from_acc<T, W>(acc[i]);
                    Tx[out + r0 + i] = v;
                    csum_mine += checksum_term<T>(v);
                }
            }
            usync();
        }
        if constexpr (MASKED) return;
        // ---- columns, in order: the wavefronts of the unit share the (lane, word) pairs -- every wavefront holds all of them
        constexpr int LSPLIT = WPU > WPL ? WPU / WPL : 1;  // wavefronts per word index
        // A lane turns its words into column numbers one bit at a time; written straight to T every store instruction of the
        // wavefront touched 64 different cache lines (each lane its own run).  The numbers go through LDS instead -- the
        // accumulators are free by now -- and leave as consecutive dwords: 256 bytes per store instruction.  (A unit with more
        // columns than accumulators -- class limits set by hand -- keeps the direct stores.)
        const bool staged = cnt <= CAP * (int)(sizeof(W) / sizeof(int));
        int *cols = (int *)acc;
        int pre = incl - c;
#pragma unroll
        for (int x = 0; x < WPL; x++) {
            unsigned long long b = mine[x];
            const bool take = WPU == 1 || (WPU <= WPL ? (x % WPU == sub) : (x == sub % WPL && lane % LSPLIT == sub / WPL));
            if (take && !MXM_ABL(a, 1)) {
                const int cbase = c0 + (lane * WPL + x) * 64;
                if (staged) {
                    int o = pre;
                    while (b) {
                        const int t = __ffsll(b) - 1;
                        b &= b - 1;
                        cols[o++] = cbase + t;
                    }
                } else {
                    int64_t o = out + pre;
                    while (b) {
                        const int t = __ffsll(b) - 1;
                        b &= b - 1;
                        a.Tj[o++] = cbase + t;
                    }
                }
            }
            pre += __popcll(mine[x]);
        }
        if (staged && !MXM_ABL(a, 1)) {
            usync();
            for (int i = tiu; i < cnt; i += 64 * WPU) a.Tj[out + i] = cols[i];
        }
    }
    if constexpr (NUMERIC) break;
    else {
        if (++w >= w_end) break;
        usync();  // (the next window clears the bitmap this one was counted from)
    }
    }  // for (;;)
    };  // run_unit
    run_unit(std::integral_constant<int, 0>{});
    if constexpr (UPW > 1) run_unit(std::integral_constant<int, 1>{});
    if constexpr (UPW > 2) run_unit(std::integral_constant<int, 2>{});
    if constexpr (UPW > 3) run_unit(std::integral_constant<int, 3>{});
    static_assert(UPW >= 1 && UPW <= 4, "one to four units per wavefront");
    if constexpr (NUMERIC && !MASKED) checksum_commit(a, csum_mine);
}

// a DENSE unit (more than MU_DENSE of the window's MM_WIN columns): compact accumulators would take several passes over the
// products; a 1024-thread workgroup with one accumulator per column of the window (k_spgemm_win's, 128 KiB of LDS for 8-byte
// values) takes one, the bitmap filled on the way.  The offset of the unit inside its row is known from the symbolic pass.
constexpr int MU_DENSE = 4096;

template <typename T>
__global__ __launch_bounds__(MM_WIN_BLOCK) void k_spgemm_unit_dense(const MxmArgs a, const UnitRec *units, int64_t nunits)
{
    const int64_t upos = xcd_block(a);
    if (upos >= nunits) return;  // (uniform over the workgroup)
    using W = typename Widen<T>::type;
    __shared__ W s_acc[MM_WIN];
    __shared__ unsigned long long s_bits[MM_WIN / 64];
    __shared__ int s_wave[MM_WIN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwin = a.n_win;
    const UnitRec r = units[upos];
    const int w = r.w;
    const int bslot = r.aux;  // (the symbolic pass kept the bitmap: no atomics on it here)
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    const W ident = monoid_identity<T, W>(monoid);
    unsigned long long csum_mine = 0;
    for (int k = tid; k < MM_WIN; k += MM_WIN_BLOCK) s_acc[k] = ident;
    if (tid < MM_WIN / 64) s_bits[tid] = bslot >= 0 ? a.bm_pool[(int64_t)bslot * (MM_WIN / 64) + tid] : 0ull;
    __syncthreads();
    const int64_t pbeg = r.pbeg, pend = pbeg + r.plen;
    const int c0 = w * MM_WIN;
    T a_iso_val = (T)0, b_iso_val = (T)0;  // (an iso operand stores one value: read once, not once per product)
    if (a.need_a && a.a_iso) a_iso_val = Ax[0];
    if (a.need_b && a.b_iso) b_iso_val = Bx[0];
    // (one copy of the loop per common semiring: see k_spgemm_unit's pass B)
    auto run = [&](auto mu_c, auto mo_c) {
        constexpr int MU = decltype(mu_c)::value, MO = decltype(mo_c)::value;
        const int mult_ = MU >= 0 ? MU : mult, monoid_ = MO >= 0 ? MO : monoid;
        for (int64_t pc = pbeg; pc < pend; pc += MM_WIN_BLOCK) {
            const int64_t p = pc + tid;
            int len = 0;
            int64_t qb = 0;
            if (p < pend) {
                const int k = a.Aj[p];
                const int32_t *o = a.woff + (int64_t)k * (nwin + 1) + w;
                const int b0 = o[0], b1 = o[1];
                qb = a.Bp[k] + b0;
                len = b1 - b0;
            }
            struct Prod {
                int j;
                T av, bv;
            };
            if (!MXM_ABL(a, 8))
                deal_products_2(
                    len, qb,
                    [&](int e, int64_t q) {
                        Prod r;
                        r.j = a.Bj[q];
                        r.av = (T)0;
                        r.bv = (T)0;
                        if constexpr (MU != OP_PAIR) {
                            if (a.need_a) r.av = a.a_iso ? a_iso_val : Ax[pc + e];
                            if (a.need_b) r.bv = a.b_iso ? b_iso_val : Bx[q];
                        }
                        return r;
                    },
                    [&](const Prod &d) {
                        if (MXM_ABL(a, 4)) return;
                        const int j = d.j - c0;
                        const W prod = (W)apply_binop<T>(mult_, d.av, d.bv);
                        if (monoid_ == OP_ANY) s_acc[j] = prod;
                        else atomic_combine<W>(&s_acc[j], prod, monoid_);
                        if (bslot < 0) atomicOr(&s_bits[j >> 6], 1ull << (j & 63));
                    });
        }
    };
    {
        using std::integral_constant;
        if (mult == OP_TIMES && monoid == OP_PLUS) run(integral_constant<int, OP_TIMES>{}, integral_constant<int, OP_PLUS>{});
        else if (mult == OP_PLUS && monoid == OP_MIN) run(integral_constant<int, OP_PLUS>{}, integral_constant<int, OP_MIN>{});
        else if (mult == OP_PAIR && monoid == OP_PLUS) run(integral_constant<int, OP_PAIR>{}, integral_constant<int, OP_PLUS>{});
        else run(integral_constant<int, -1>{}, integral_constant<int, -1>{});
    }
    __syncthreads();
    if (a.CMp && bslot < 0) {  // fused complemented mask: forbidden columns are not emitted (the kept bitmaps are already without them)
        const int32_t *mo = a.cm_woff + (int64_t)r.row * (nwin + 1) + w;
        const int m0 = mo[0], m1 = mo[1];
        const int64_t mb = a.CMp[r.row];
        for (int i = m0 + tid; i < m1; i += MM_WIN_BLOCK) {
            const int j = a.CMj[mb + i] - c0;
            atomicAnd(&s_bits[j >> 6], ~(1ull << (j & 63)));
        }
        __syncthreads();
    }
    // emit: thread t takes 16 columns (a quarter of word t / 4)
    const unsigned long long word = s_bits[tid >> 2];
    const int q4 = tid & 3;
    unsigned b16 = (unsigned)(word >> (16 * q4)) & 0xFFFFu;
    const int c = __popc(b16);
    int incl = wave_inclusive_sum(c);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int x = 0; x < wv; x++) wave_off += s_wave[x];
    int64_t o = r.out + wave_off + (incl - c);
    T *Tx = (T *)a.Tx;
    while (b16) {
        const int t = __ffs(b16) - 1;
        b16 &= b16 - 1;
        const int j = tid * 16 + t;
        if (!MXM_ABL(a, 1)) a.Tj[o] = c0 + j;
        const T v = from_acc<T, W>(s_acc[j]);
        if (!MXM_ABL(a, 2)) Tx[o] = v;
        csum_mine += checksum_term<T>(v);
        o++;
    }
    checksum_commit(a, csum_mine);
}

// the units of the rows of the numeric bin, by class (the first class whose limit the entry count does not exceed, MU_NCLS - 1
// = dense beyond the last limit; empty units are dropped): counts (FILL = false) or the lists themselves, (row << 16) |
// window, class c from cursor[c] on.  One wavefront per row, lanes over the windows, one atomic per wavefront, class and batch
// of 64 windows.
// Round 5: classes 0 .. 2 = GROUP units (F > 1 consecutive windows of a row whose entries together fit the compact classes: up to
// lim[0] / lim[1] / lim[2] entries), 3 .. 5 = single-window units of up to lim[0] / lim[1] / lim[2] entries, 6 = dense single-window units.
// With F = 1 (and in the mask-driven product) every unit is a single window.
constexpr int MU_NCLS = 7, MU_CLS_WINDOW = 3;
struct UnitLimits {
    int lim[3];
};
// the class of the unit lane `lane` LEADS (-1: none), cnt = the entries of the lane's window; the lanes of a wavefront hold consecutive
// windows starting at a multiple of 64, so an aligned run of F lanes is a group.  All 64 lanes must call this.
__device__ __forceinline__ int unit_class_of(int cnt, int F, const UnitLimits &L, int lane, bool *grouped_out = nullptr)
{
    int tot = cnt;
    for (int d = 1; d < F; d <<= 1) tot += __shfl_xor(tot, d);
    const bool grouped = F > 1 && tot <= L.lim[2];
    if (grouped_out) *grouped_out = grouped;
    if (grouped) {
        if ((lane & (F - 1)) != 0 || tot == 0) return -1;
        return tot <= L.lim[0] ? 0 : (tot <= L.lim[1] ? 1 : 2);
    }
    if (cnt <= 0) return -1;
    return MU_CLS_WINDOW + (cnt <= L.lim[0] ? 0 : (cnt <= L.lim[1] ? 1 : (cnt <= L.lim[2] ? 2 : 3)));
}
// class counters and list cursors are kept per SLOT of the row ((row / 4) mod MU_CSLOTS): millions of atomics on one address
// serialise (21 ms per pass over the 4 M rows of scale 22 with a single counter per class)
constexpr int MU_CSLOTS = 256;
__device__ __forceinline__ int class_slot(int64_t row) { return (int)((row >> 2) & (MU_CSLOTS - 1)); }

template <bool FILL>
__global__ __launch_bounds__(256) void k_unit_classify(const int32_t *wcnt, const int32_t *wrow, int nwin, const uint32_t *rows, int64_t nrows_bin,
                                                       unsigned long long *cursor, UnitRec *lists, UnitLimits L, const int64_t *Ap,
                                                       const int64_t *base_ptr, const int32_t *wbm, int masked, int F)
{
    const int lane = threadIdx.x & 63;
    const int64_t ridx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= nrows_bin) return;
    const int64_t row = rows[ridx];
    const int slot = wrow[row];
    if (slot < 0) return;
    const int32_t *wc = wcnt + (int64_t)slot * (nwin + 1);
    unsigned long long *cur = cursor + class_slot(row) * MU_NCLS;
    unsigned mine[MU_NCLS] = {0};
    for (int b = 0; b < nwin; b += 64) {
        const int w = b + lane;
        const int cnt = w < nwin ? wc[w + 1] - wc[w] : 0;
        const int cls = unit_class_of(cnt, F, L, lane);
        for (int c = 0; c < MU_NCLS; c++) {
            const unsigned long long mk = __ballot(cls == c);
            if (mk == 0) continue;
            if constexpr (!FILL) {
                mine[c] += (unsigned)__popcll(mk);
            } else {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&cur[c], (unsigned long long)__popcll(mk));
                base = __shfl(base, 0);
                if (cls == c) {
                    UnitRec r;
                    r.out = base_ptr[row] + wc[w];  // (base_ptr: the row pointers of T, or of the mask; a group starts where its first window starts)
                    r.pbeg = Ap[row];
                    r.plen = (int32_t)(Ap[row + 1] - r.pbeg);
                    r.row = (uint32_t)row;
                    r.aux = masked ? cnt : (wbm ? wbm[(int64_t)slot * nwin + w] : -1);  // (a group's bitmap: F window bitmaps from its first window's slot on)
                    r.w = c < MU_CLS_WINDOW ? w / F : w;  // (group units count groups)
                    lists[base + __popcll(mk & ((1ull << lane) - 1ull))] = r;
                }
            }
        }
    }
    if constexpr (!FILL) {
        if (lane == 0)
            for (int c = 0; c < MU_NCLS; c++)
                if (mine[c]) atomicAdd(&cur[c], (unsigned long long)mine[c]);
    }
}

// k_window_offsets with a WAVEFRONT per row: lane l searches the row for the first entry of windows l, l + 64, ... (the writes
// are coalesced; a thread per row wrote 4.3 GB of offsets at scale 22 with a 1 KiB stride between lanes: 18 ms).  With
// class_count: the rows with urow[row] >= 0 also have their windows classified by entry count (the counting pass of
// k_unit_classify for the mask-driven product, whose "counts" are these offsets of the MASK rows).
__global__ __launch_bounds__(256) void k_window_offsets_wave(const int64_t *Bp, const int32_t *Bj, int64_t nrowsB, int n_win, int32_t *woff,
                                                             const int32_t *urow, unsigned long long *class_count, UnitLimits L)
{
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nrowsB) return;
    const int64_t b = Bp[k], e = Bp[k + 1];
    int32_t *o = woff + k * (int64_t)(n_win + 1);
    const bool classify = class_count && (!urow || urow[k] >= 0) && e > b;
    unsigned ccount[MU_NCLS] = {0};
    for (int w0 = 0; w0 <= n_win; w0 += 64) {
        const int w = w0 + lane;
        int first = 0;
        if (w <= n_win) {
            const int64_t target = (int64_t)w * MM_WIN;
            int64_t lo = b, hi = e;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (Bj[mid] < target) lo = mid + 1;
                else hi = mid;
            }
            first = (int)(lo - b);
            o[w] = first;
        }
        if (classify) {
            // the window's count = the next window's first entry - this one's (the last lane of a batch searches one more)
            int next = __shfl_down(first, 1);
            if (lane == 63 || w == n_win) {
                next = first;
                if (w < n_win) {
                    const int64_t target = (int64_t)(w + 1) * MM_WIN;
                    int64_t lo = b + first, hi = e;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (Bj[mid] < target) lo = mid + 1;
                        else hi = mid;
                    }
                    next = (int)(lo - b);
                }
            }
            const int cnt = w < n_win ? next - first : 0;
            const int cls = unit_class_of(cnt, 1, L, lane);  // (the mask-driven units walk single windows)
            for (int c = 0; c < MU_NCLS; c++) ccount[c] += (unsigned)__popcll(__ballot(cls == c));
        }
    }
    if (classify && lane == 0)
        for (int c = 0; c < MU_NCLS; c++)
            if (ccount[c]) atomicAdd(&class_count[class_slot(k) * MU_NCLS + c], (unsigned long long)ccount[c]);
}

// The same without any search, for up to WO_MAX_WIN windows: the wavefront histograms the window indices of the row's entries in
// LDS (run lengths of the sorted row, no atomics), a wavefront scan of the histogram gives the offsets, written coalesced; the histogram is the
// per-window entry count the classification needs.  (The search version costs a binary search per (row, window): 10^9 of them
// for the 4 M rows x 257 windows of scale 22, 21 ms; this one 4 ms.)
constexpr int WO_MAX_WIN = 2047;

__global__ __launch_bounds__(256) void k_window_offsets_hist(const int64_t *Bp, const int32_t *Bj, int64_t nrowsB, int n_win, int32_t *woff,
                                                             const int32_t *urow, unsigned long long *class_count, UnitLimits L)
{
    __shared__ int s_hist[4][WO_MAX_WIN + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 4 + wave;
    if (k >= nrowsB) return;
    int *hist = s_hist[wave];
    const int64_t b = Bp[k], e = Bp[k + 1];
    int32_t *o = woff + k * (int64_t)(n_win + 1);
    const bool classify = class_count && (!urow || urow[k] >= 0) && e > b;
    for (int w = lane; w < n_win; w += 64) hist[w] = 0;
    mw_sync();
    // (the row is sorted: equal window indices are contiguous -- the first lane of every run adds the run's length, no atomics:
    //  with one LDS atomic per entry the hub rows, thousands of entries in window 0, serialised the kernel to the speed of the
    //  search version)
    for (int64_t p0 = b; p0 < e; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool valid = p < e;
        const int w = valid ? Bj[p] / MM_WIN : -1;
        const int prev = __shfl_up(w, 1);
        const bool head = valid && (lane == 0 || prev != w);
        const unsigned long long heads = __ballot(head), valids = __ballot(valid);
        if (head) {
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int run = above ? __ffsll(above) : __popcll(valids) - lane;  // to the next run's first lane, or to the end of the chunk
            hist[w] += run;
        }
        mw_sync();  // (a run may continue in the next chunk: same wavefront, same counter)
    }
    int carry = 0;
    unsigned ccount[MU_NCLS] = {0};
    for (int w0 = 0; w0 <= n_win; w0 += 64) {
        const int w = w0 + lane;
        const int cnt = w < n_win ? hist[w] : 0;
        int incl = wave_inclusive_sum(cnt);
        if (w <= n_win) o[w] = carry + incl - cnt;  // entries of the row before window w
        carry += __builtin_amdgcn_readlane(incl, 63);
        if (classify) {
            const int cls = unit_class_of(cnt, 1, L, lane);
            for (int c = 0; c < MU_NCLS; c++) ccount[c] += (unsigned)__popcll(__ballot(cls == c));
        }
    }
    if (classify && lane == 0)
        for (int c = 0; c < MU_NCLS; c++)
            if (ccount[c]) atomicAdd(&class_count[class_slot(k) * MU_NCLS + c], (unsigned long long)ccount[c]);
}

static void launch_window_offsets(const int64_t *Bp, const int32_t *Bj, int64_t nrowsB, int n_win, int32_t *woff, const int32_t *urow,
                                  unsigned long long *class_count, UnitLimits L)
{
    if (n_win <= WO_MAX_WIN)
        hipLaunchKernelGGL(k_window_offsets_hist, dim3((unsigned)ceil_div(nrowsB, 4)), dim3(256), 0, ctx().stream, Bp, Bj, nrowsB, n_win, woff,
                           urow, class_count, L);
    else
        hipLaunchKernelGGL(k_window_offsets_wave, dim3((unsigned)ceil_div(nrowsB, 4)), dim3(256), 0, ctx().stream, Bp, Bj, nrowsB, n_win, woff,
                           urow, class_count, L);
}

// positions (in the bin's row list) of the rows with more than plen_max entries of A, in no particular order
__global__ void k_sym_long_rows(const uint32_t *rows, int64_t nrows_bin, const int64_t *Ap, int plen_max, unsigned long long *cursor, int32_t *list)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows_bin) return;
    const int64_t row = rows[i];
    if (Ap[row + 1] - Ap[row] > plen_max) list[atomicAdd(cursor, 1ull)] = (int32_t)i;
}

// the per-window counts of the rows of the symbolic unit pass -> offsets inside the row (exclusive scan in place, n_win + 1
// numbers per row), the row's entry count, and the row -> slot map the numeric pass finds its units' offsets with.
// One wavefront per row.
// ... and, on the way, how many units each class of the numeric pass will hold (class_count[c]; k_unit_classify's counting pass)
__global__ __launch_bounds__(256) void k_unit_prefix(int32_t *wcnt, int nwin, const uint32_t *rows, int64_t nrows_bin, int64_t *row_nnz,
                                                     int32_t *wrow, unsigned long long *class_count, UnitLimits L, int F)
{
    const int lane = threadIdx.x & 63;
    const int64_t ridx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ridx >= nrows_bin) return;
    int32_t *wc = wcnt + ridx * (nwin + 1);
    int carry = 0;
    unsigned ccount[MU_NCLS] = {0};
    for (int b = 0; b < nwin; b += 64) {
        const int v = b + lane < nwin ? wc[b + lane] : 0;
        const int cls = unit_class_of(v, F, L, lane);
        for (int c = 0; c < MU_NCLS; c++) ccount[c] += (unsigned)__popcll(__ballot(cls == c));
        int incl = wave_inclusive_sum(v);
        if (b + lane < nwin) wc[b + lane] = carry + incl - v;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) {
        wc[nwin] = carry;
        const uint32_t row = rows[ridx];
        row_nnz[row] = carry;
        wrow[row] = (int32_t)ridx;
        for (int c = 0; c < MU_NCLS; c++)
            if (ccount[c]) atomicAdd(&class_count[class_slot(row) * MU_NCLS + c], (unsigned long long)ccount[c]);
    }
}

// ---- mask-driven product: C<M> = A (+.x) B with a non-complemented mask only needs the entries of T inside M's pattern
//      (the write rule never looks at the others), so the accumulators are keyed by the MASK row: no symbolic pass, no
//      sort (M's rows are sorted), output size bounded by nnz(M) however dense A*B is -------------------------------------
// rows with at most TABLE/2 mask entries: LDS hash pre-loaded with the mask row's columns; products only look up
template <typename T, int TABLE>
__global__ __launch_bounds__(MM_BLOCK) void k_spgemm_mhash(const MxmArgs a, const uint32_t *rows)
{
    using W = typename Widen<T>::type;
    __shared__ int s_key[TABLE];
    __shared__ W s_val[TABLE];
    __shared__ unsigned char s_hit[TABLE];
    const int tid = threadIdx.x;
    const int64_t row = rows[blockIdx.x];
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    const int64_t mlo = a.Mp[row], mhi = a.Mp[row + 1];
    for (int k = tid; k < TABLE; k += MM_BLOCK) {
        s_key[k] = -1;
        s_val[k] = monoid_identity<T, W>(monoid);
        s_hit[k] = 0;
    }
    __syncthreads();
    for (int64_t p = mlo + tid; p < mhi; p += MM_BLOCK) {
        const int j = a.Mj[p];
        unsigned h = hash_col(j, TABLE - 1);
        while (true) {
            const int old = atomicCAS(&s_key[h], -1, j);
            if (old == -1 || old == j) break;
            h = (h + 1) & (TABLE - 1);
        }
    }
    __syncthreads();
    foreach_product(a, row, [&](int j, int64_t p, int64_t q) {
        unsigned h = hash_col(j, TABLE - 1);
        while (true) {
            const int key = s_key[h];
            if (key == j) {
                const T av = a.need_a ? Ax[a.a_iso ? 0 : p] : (T)0;
                const T bv = a.need_b ? Bx[a.b_iso ? 0 : q] : (T)0;
                const W prod = (W)apply_binop<T>(mult, av, bv);
                if (monoid == OP_ANY) s_val[h] = prod;
                else atomic_combine<W>(&s_val[h], prod, monoid);
                s_hit[h] = 1;
                break;
            }
            if (key == -1) break;  // not in the mask row
            h = (h + 1) & (TABLE - 1);
        }
    });
    __syncthreads();
    T *cv = (T *)a.cap_val;
    for (int64_t p = mlo + tid; p < mhi; p += MM_BLOCK) {
        const int j = a.Mj[p];
        unsigned h = hash_col(j, TABLE - 1);
        while (s_key[h] != j) h = (h + 1) & (TABLE - 1);
        const unsigned char hit = s_hit[h];
        a.cap_hit[p] = hit;
        if (hit) cv[p] = from_acc<T, W>(s_val[h]);
    }
}

// longer mask rows: LDS column windows as k_spgemm_win, plus a bitmap of the mask row inside the window; windows without
// mask entries are skipped
template <typename T>
__global__ __launch_bounds__(MM_WIN_BLOCK) void k_spgemm_mwin(const MxmArgs a, const uint32_t *rows)
{
    using W = typename Widen<T>::type;
    __shared__ W s_acc[MM_WIN];
    __shared__ unsigned long long s_bits[MM_WIN / 64];
    __shared__ unsigned long long s_mbits[MM_WIN / 64];
    __shared__ int64_t s_mrange[2];
    const int tid = threadIdx.x;
    const int monoid = a.monoid, mult = a.mult;
    const T *Ax = (const T *)a.Ax, *Bx = (const T *)a.Bx;
    const int64_t row = rows[blockIdx.x];
    if (a.wrow && a.wrow[row] >= 0) return;  // (the row's windows are units of k_spgemm_unit)
    const W ident = monoid_identity<T, W>(monoid);
    for (int k = tid; k < MM_WIN; k += MM_WIN_BLOCK) s_acc[k] = ident;
    if (tid < MM_WIN / 64) { s_bits[tid] = 0ull; s_mbits[tid] = 0ull; }
    const int64_t mlo = a.Mp[row], mhi = a.Mp[row + 1];
    T *cv = (T *)a.cap_val;
    int64_t mpos = mlo;  // mask entries before it lie in earlier windows
    __syncthreads();
    walk_windows(a, row, [&](int w, auto &&visit) {
        if (mpos >= mhi) return;  // (uniform) the mask row is exhausted
        const int c0 = w * MM_WIN;
        if (tid == 0) {  // the mask row's entries inside the window: [mpos, first entry with column >= c0 + MM_WIN)
            int64_t lo = mpos, hi = mhi;
            const int64_t lim = (int64_t)c0 + MM_WIN;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)a.Mj[mid] < lim) lo = mid + 1;
                else hi = mid;
            }
            s_mrange[0] = mpos;
            s_mrange[1] = lo;
        }
        __syncthreads();
        const int64_t wlo = s_mrange[0], whi = s_mrange[1];
        __syncthreads();
        mpos = whi;
        if (wlo == whi) return;  // (uniform) no mask entry in this window
        for (int64_t p = wlo + tid; p < whi; p += MM_WIN_BLOCK) {
            const int j = a.Mj[p] - c0;
            atomicOr(&s_mbits[j >> 6], 1ull << (j & 63));
        }
        __syncthreads();
        visit([&](int64_t p, int64_t q) {
            const int j = a.Bj[q] - c0;
            if ((s_mbits[j >> 6] >> (j & 63)) & 1ull) {
                const T av = a.need_a ? Ax[a.a_iso ? 0 : p] : (T)0;
                const T bv = a.need_b ? Bx[a.b_iso ? 0 : q] : (T)0;
                const W prod = (W)apply_binop<T>(mult, av, bv);
                if (monoid == OP_ANY) s_acc[j] = prod;
                else atomic_combine<W>(&s_acc[j], prod, monoid);
                atomicOr(&s_bits[j >> 6], 1ull << (j & 63));
            }
        });
        __syncthreads();
        for (int64_t p = wlo + tid; p < whi; p += MM_WIN_BLOCK) {
            const int j = a.Mj[p] - c0;
            const bool hit = (s_bits[j >> 6] >> (j & 63)) & 1ull;
            a.cap_hit[p] = hit ? 1 : 0;
            if (hit) {
                cv[p] = from_acc<T, W>(s_acc[j]);
                s_acc[j] = ident;
            }
        }
        __syncthreads();
        if (tid < MM_WIN / 64) { s_bits[tid] = 0ull; s_mbits[tid] = 0ull; }
        __syncthreads();
    });
}

// compaction of the mask-layout results into CSR
// urow[i] = i for the rows of a masked product that are walked as units (more than min_flops products and a mask row), else -1
__global__ void k_mask_unit_rows(const int64_t *Ap, const int64_t *F, const int64_t *Mp, int64_t m, int64_t min_flops, int32_t *urow)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) urow[i] = (F[Ap[i + 1]] - F[Ap[i]] > min_flops && Mp[i + 1] > Mp[i]) ? (int32_t)i : -1;
}

__global__ void k_mask_sizes(const int64_t *Ap, const int64_t *Mp, int64_t m, int64_t *size)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) size[i] = (Ap[i + 1] > Ap[i]) ? Mp[i + 1] - Mp[i] : 0;
}
__global__ void k_hits_to_i64(const unsigned char *hit, int64_t n, int64_t *out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= n) out[p] = p < n ? (int64_t)hit[p] : 0;
}
__global__ void k_cap_rowptr(const int64_t *Mp, const int64_t *pos, int64_t m, int64_t *Tp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= m) Tp[i] = pos[Mp[i]];
}
template <typename T>
__global__ void k_cap_scatter(const unsigned char *hit, const int64_t *pos, const int32_t *Mj, const T *cap_val, int64_t n, int32_t *Tj, T *Tx)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && hit[p]) {
        Tj[pos[p]] = Mj[p];
        Tx[pos[p]] = cap_val[p];
    }
}

// ---- symbolic pass for rows whose upper bound exceeds the LDS hash table: the presence bitmap of the WHOLE column
//      range lives in LDS (n <= 2^20 columns = 128 KiB), filled by LDS atomicOr, counted by a popcount sweep ---------
template <int WORDS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_spgemm_sym_lds(const MxmArgs a, const uint32_t *rows)
{
    __shared__ unsigned long long s_bits[WORDS];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int64_t row = rows[blockIdx.x];
    if (tid == 0) s_cnt = 0;
    // wider matrices: the column range is covered in passes of WORDS * 64 columns (every pass visits all products and keeps the
    // ones of its range -- a pass in LDS is still far cheaper than one round of global atomics on a bitmap in HBM)
    const int passes = (int)((a.n + (int64_t)WORDS * 64 - 1) / ((int64_t)WORDS * 64));
    for (int pass = 0; pass < passes; pass++) {
        for (int k = tid; k < WORDS; k += BLOCK) s_bits[k] = 0ull;
        __syncthreads();
        const int jlo = pass * WORDS * 64;
        foreach_product<BLOCK>(a, row, [&](int j, int64_t, int64_t) {
            const unsigned r = (unsigned)(j - jlo);
            if (r < (unsigned)(WORDS * 64)) atomicOr(&s_bits[r >> 6], 1ull << (r & 63));
        });
        __syncthreads();
        int c = 0;
        for (int k = tid; k < WORDS; k += BLOCK) c += __popcll(s_bits[k]);
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if ((tid & 63) == 0 && c) atomicAdd(&s_cnt, c);
        __syncthreads();
    }
    if (tid == 0) a.row_nnz[row] = s_cnt;
}

template <typename W>
__global__ void k_fill_ident(W *p, int64_t n, W v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- write rule on sorted CSR rows: one thread per row, two passes (count / fill) -------------------------------------
template <typename TM>
__device__ __forceinline__ bool mask_true_at(const TM *Mx, int m_iso, int64_t p) { return Mx[m_iso ? 0 : p] != (TM)0; }

template <typename T, bool FILL>
__global__ void k_mat_write(int64_t m, const int64_t *Cp, const int32_t *Cj, const T *Cx, int c_iso, const int64_t *Tp,
                            const int32_t *Tj, const T *Tx, const int64_t *Mp, const int32_t *Mj, const void *Mx, int m_type,
                            int m_iso, int has_mask, int m_struct, int m_comp, int accum, int replace, int64_t *Ncount,
                            const int64_t *Np, int32_t *Nj, T *Nx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int64_t pc = Cp ? Cp[i] : 0, ec = Cp ? Cp[i + 1] : 0, pt = Tp[i], et = Tp[i + 1];
    int64_t pm = has_mask ? Mp[i] : 0, em = has_mask ? Mp[i + 1] : 0;
    int64_t o = FILL ? Np[i] : 0, cnt = 0;
    while (pc < ec || pt < et) {
        const int jc = pc < ec ? Cj[pc] : 0x7fffffff, jt = pt < et ? Tj[pt] : 0x7fffffff;
        const int j = jc < jt ? jc : jt;
        const bool hc = (jc == j), ht = (jt == j);
        bool mk = true;
        if (has_mask) {
            while (pm < em && Mj[pm] < j) pm++;
            mk = (pm < em && Mj[pm] == j);
            if (mk && !m_struct) {
                switch (m_type) {
                case TC_BOOL: case TC_INT8: case TC_UINT8: mk = mask_true_at<uint8_t>((const uint8_t *)Mx, m_iso, pm); break;
                case TC_INT16: case TC_UINT16: mk = mask_true_at<uint16_t>((const uint16_t *)Mx, m_iso, pm); break;
                case TC_INT32: case TC_UINT32: mk = mask_true_at<uint32_t>((const uint32_t *)Mx, m_iso, pm); break;
                case TC_INT64: case TC_UINT64: mk = mask_true_at<uint64_t>((const uint64_t *)Mx, m_iso, pm); break;
                case TC_FP32: mk = mask_true_at<float>((const float *)Mx, m_iso, pm); break;
                default: mk = mask_true_at<double>((const double *)Mx, m_iso, pm); break;
                }
            }
            if (m_comp) mk = !mk;
        }
        bool zh = false;
        T zv = (T)0;
        if (mk) {
            if (accum >= 0) {
                const T cv = hc ? Cx[c_iso ? 0 : pc] : (T)0;
                if (hc && ht) { zh = true; zv = apply_binop<T>(accum, cv, Tx[pt]); }
                else if (hc) { zh = true; zv = cv; }
                else { zh = true; zv = Tx[pt]; }
            } else if (ht) { zh = true; zv = Tx[pt]; }
        } else if (!replace && hc) { zh = true; zv = Cx[c_iso ? 0 : pc]; }
        if (zh) {
            if (FILL) { Nj[o + cnt] = j; Nx[o + cnt] = zv; }
            cnt++;
        }
        if (hc) pc++;
        if (ht) pt++;
    }
    if (!FILL) Ncount[i] = cnt;
}

#include "grb_mxm_write.inc"

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct RowBins {
    DevBuf<uint32_t> rows;
    int64_t start[6];
    explicit RowBins(int64_t m) : rows(m) {}
    int64_t count(int b) const { return start[b + 1] - start[b]; }
    const uint32_t *ptr(int b) const { return rows.p + start[b]; }
};

// stable sort of rows by bin(size): rows inside a bin stay in increasing order
static void make_bins(RowBins &rb, const int64_t *Ap, const int64_t *F, int64_t m, int64_t *size, int64_t b1, int64_t b2,
                      int64_t b3, const int32_t *wrow = nullptr, const int64_t *extra_ptr = nullptr)
{
    DevBuf<uint64_t> key(m), key2(m);
    DevBuf<uint32_t> rid(m);
    hipLaunchKernelGGL(k_row_bins, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream, Ap, F, m, size, b1, b2, b3,
                       key.p, rid.p, wrow, extra_ptr);
    prim_sort_pairs_u64_u32(key.p, key2.p, rid.p, rb.rows.p, m, 3);
    DevBuf<int64_t> bs(6);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(64), 0, ctx().stream, key2.p, m, bs.p);
    d2h(rb.start, bs.p, sizeof(int64_t) * 6);
    ctx().stats.kernel_launches += 2;
}

// the units of the rows `rows` (those with a slot: wrow[row] >= 0) by class, then one launch per class (and 2^21 units: a grid holds
// fewer than 2^32 threads).  MODE = MU_NUMERIC: classes by the entry counts of the symbolic pass; MU_MASKED: by the number of mask
// entries inside the window (a.wcnt = the window offsets of the mask rows), the densest class takes several passes.
static UnitLimits unit_limits(bool masked)
{
    UnitLimits L;
    L.lim[0] = std::min(MU_SMALL, ctx().mxm_unit_small);
    L.lim[1] = std::max(L.lim[0], ctx().mxm_unit_mid);
    L.lim[2] = masked ? INT32_MAX : std::max(L.lim[1], ctx().mxm_unit_dense);
    return L;
}

// (known: the per-slot class counts on the host, MU_CSLOTS x MU_NCLS numbers, when an earlier pass already took them --
//  k_unit_prefix of the symbolic pass, the mask's k_window_offsets)
template <typename T, int MODE>
static void launch_unit_classes(MxmArgs &a, const uint32_t *rows, int64_t nrows, const unsigned long long *known = nullptr)
{
    constexpr int NC = MU_CSLOTS * MU_NCLS;
    DevBuf<unsigned long long> cur(NC, true);
    const UnitLimits L = unit_limits(MODE == MU_MASKED);
    const int FG = MODE == MU_MASKED ? 1 : std::max(1, a.win_f);  // windows per group unit
    std::vector<unsigned long long> slot_cnt(NC);
    if (known) {
        std::copy(known, known + NC, slot_cnt.begin());
    } else {
        hipLaunchKernelGGL((k_unit_classify<false>), dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, ctx().stream, (const int32_t *)a.wcnt,
                           (const int32_t *)a.wrow, a.n_win, rows, nrows, cur.p, (UnitRec *)nullptr, L, a.Ap, (const int64_t *)nullptr,
                           (const int32_t *)nullptr, 0, FG);
        d2h(slot_cnt.data(), cur.p, sizeof(unsigned long long) * NC);
    }
    // class c's list = the slots' sub-lists one after the other: the fill pass's cursors start at the sub-lists' beginnings
    unsigned long long cnt[MU_NCLS] = {0}, start[MU_NCLS + 1] = {0};
    for (int sl = 0; sl < MU_CSLOTS; sl++)
        for (int c = 0; c < MU_NCLS; c++) cnt[c] += slot_cnt[(size_t)sl * MU_NCLS + c];
    for (int c = 0; c < MU_NCLS; c++) start[c + 1] = start[c] + cnt[c];
    std::vector<unsigned long long> cursors(NC);
    {
        unsigned long long run[MU_NCLS];
        for (int c = 0; c < MU_NCLS; c++) run[c] = start[c];
        for (int sl = 0; sl < MU_CSLOTS; sl++)
            for (int c = 0; c < MU_NCLS; c++) {
                cursors[(size_t)sl * MU_NCLS + c] = run[c];
                run[c] += slot_cnt[(size_t)sl * MU_NCLS + c];
            }
    }
    if (getenv("GRB_MXM_TRACE"))
        fprintf(stderr, "[mxm] %s units: %lld rows x %d windows, groups of %d; group classes %llu %llu %llu, window classes %llu %llu %llu, dense %llu\n",
                MODE == MU_MASKED ? "masked" : "numeric", (long long)nrows, a.n_win, FG, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6]);
    DevBuf<UnitRec> lists((size_t)start[MU_NCLS]);
    h2d(cur.p, cursors.data(), sizeof(unsigned long long) * NC);
    hipLaunchKernelGGL((k_unit_classify<true>), dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, ctx().stream, (const int32_t *)a.wcnt,
                       (const int32_t *)a.wrow, a.n_win, rows, nrows, cur.p, lists.p, L, a.Ap, MODE == MU_MASKED ? a.Mp : a.Tp,
                       (const int32_t *)a.wbm, MODE == MU_MASKED ? 1 : 0, FG);
    constexpr int64_t PER = 1ll << 21;
    auto per_class = [&](int c, auto &&launch) {
        for (int64_t u0 = 0; u0 < (int64_t)cnt[c]; u0 += PER)
            launch(lists.p + start[c] + u0, std::min<int64_t>(PER, (int64_t)cnt[c] - u0));
    };
    // the three compact classes of units of F windows (F = 1: single windows)
    auto grid8 = [&](int64_t g) { return (unsigned)(a.xcd_map ? ceil_div(g, (int64_t)8) * 8 : g); };  // (xcd_block covers a multiple of 8)
    auto compact_classes = [&](int c0, auto f_c) {
        constexpr int F = decltype(f_c)::value;
        per_class(c0, [&](const UnitRec *u, int64_t nu) {
            hipLaunchKernelGGL((k_spgemm_unit<T, MODE, 1, MU_SMALL, F>), dim3(grid8(ceil_div(nu, (int64_t)4 * mu_units_per_wave(MODE, 1)))), dim3(256), 0, ctx().stream, a, rows, 0, 0, u, nu);
        });
        per_class(c0 + 1, [&](const UnitRec *u, int64_t nu) {
            hipLaunchKernelGGL((k_spgemm_unit<T, MODE, 4, 1024, F>), dim3(grid8(nu)), dim3(256), 0, ctx().stream, a, rows, 0, 0, u, nu);
        });
        per_class(c0 + 2, [&](const UnitRec *u, int64_t nu) {
            // (group units: 128 accumulators fewer -- 40448 bytes of LDS, four workgroups per CU instead of three; a unit of more than
            //  3968 entries takes a second pass over its products)
            hipLaunchKernelGGL((k_spgemm_unit<T, MODE, GRB_MU_M2_WPU, (F == 1 ? 4096 : 3968), F>), dim3(grid8(nu)), dim3(64 * GRB_MU_M2_WPU), 0, ctx().stream, a, rows, 0, 0, u, nu);
        });
    };
    if constexpr (MODE == MU_NUMERIC) {
        using std::integral_constant;
        if (FG == 2) compact_classes(0, integral_constant<int, 2>{});
        else if (FG == 4) compact_classes(0, integral_constant<int, 4>{});
        else if (FG == 8) compact_classes(0, integral_constant<int, 8>{});
    }
    compact_classes(MU_CLS_WINDOW, std::integral_constant<int, 1>{});
    if constexpr (MODE == MU_NUMERIC)
        per_class(MU_CLS_WINDOW + 3, [&](const UnitRec *u, int64_t nu) {
            hipLaunchKernelGGL((k_spgemm_unit_dense<T>), dim3(grid8(nu)), dim3(MM_WIN_BLOCK), 0, ctx().stream, a, u, nu);
        });
    ctx().stats.kernel_launches += 6;
    sync_stream();  // (the lists are freed at the end of this scope)
}

template <typename T, bool NUMERIC>
static void run_bins(MxmArgs &a, const RowBins &rb)
{
    using W = typename Widen<T>::type;
    constexpr int T1 = 256, T2 = 2048, T3 = NUMERIC ? 8192 : 32768;
    if (rb.count(1)) hipLaunchKernelGGL((k_spgemm_hash<T, T1, NUMERIC>), dim3((unsigned)rb.count(1)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(1));
    if (rb.count(2)) hipLaunchKernelGGL((k_spgemm_hash<T, T2, NUMERIC>), dim3((unsigned)rb.count(2)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(2));
    if (rb.count(3)) hipLaunchKernelGGL((k_spgemm_hash<T, T3, NUMERIC>), dim3((unsigned)rb.count(3)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(3));
    ctx().stats.kernel_launches += 3;
    if (rb.count(4) && a.woff && a.wrow && a.wcnt) {
        if constexpr (NUMERIC) {
            launch_unit_classes<T, MU_NUMERIC>(a, rb.ptr(4), rb.count(4), a.class_known);
            // rows of the bin the symbolic pass counted with a hash kernel (few products, but more entries than the numeric
            // hash table holds): the 1024-thread window walk
            hipLaunchKernelGGL((k_spgemm_win<T>), dim3((unsigned)rb.count(4)), dim3(MM_WIN_BLOCK), 0, ctx().stream, a, rb.ptr(4));
            ctx().stats.kernel_launches += 1;
        } else {
            // rows with at most MU_SYM_PLEN entries of A (94 % of the unit rows of an R-MAT product): MU_SYM_WG windows per unit;
            // the others from a list, one window per unit
            const int FG = std::max(1, a.win_f);
            const int64_t n_groups = ceil_div((int64_t)a.n_win, (int64_t)FG);  // (units walk groups of FG windows)
            const int wg = (int)std::max<int64_t>(1, std::min<int64_t>(ctx().mxm_sym_windows, n_groups));
            DevBuf<int32_t> long_list(wg > 1 ? rb.count(4) : 0);
            int64_t n_long_rows = 0;
            if (wg > 1) {
                DevBuf<unsigned long long> cur(1, true);
                hipLaunchKernelGGL(k_sym_long_rows, dim3((unsigned)ceil_div(rb.count(4), 256)), dim3(256), 0, ctx().stream, rb.ptr(4), rb.count(4), a.Ap,
                                   MU_SYM_PLEN, cur.p, long_list.p);
                unsigned long long got = 0;
                d2h(&got, cur.p, sizeof(got));
                n_long_rows = (int64_t)got;
            }
            auto launch_sym = [&](int64_t nrows_total, int wg_here, int plen_max, const int32_t *list) {
                MxmArgs as = a;
                as.sym_wg = wg_here;
                as.sym_plen_max = plen_max;
                as.sym_list = list;
                const int64_t ngrp = ceil_div(n_groups, (int64_t)wg_here);
                const int64_t rows_per_launch = std::max<int64_t>(1, (1ll << 22) / ngrp);
                for (int64_t r0 = 0; r0 < nrows_total; r0 += rows_per_launch) {
                    const int64_t nr = std::min(rows_per_launch, nrows_total - r0);
                    const dim3 grid((unsigned)(as.xcd_map ? ceil_div(ceil_div(nr * ngrp, 4), (int64_t)8) * 8 : ceil_div(nr * ngrp, 4)));
                    const UnitRec *none = nullptr;
                    if (FG == 2) hipLaunchKernelGGL((k_spgemm_unit<T, MU_SYMBOLIC, 1, 1, 2>), grid, dim3(256), 0, ctx().stream, as, rb.ptr(4), r0, nr, none, 0);
                    else if (FG == 4) hipLaunchKernelGGL((k_spgemm_unit<T, MU_SYMBOLIC, 1, 1, 4>), grid, dim3(256), 0, ctx().stream, as, rb.ptr(4), r0, nr, none, 0);
                    else if (FG == 8) hipLaunchKernelGGL((k_spgemm_unit<T, MU_SYMBOLIC, 1, 1, 8>), grid, dim3(256), 0, ctx().stream, as, rb.ptr(4), r0, nr, none, 0);
                    else hipLaunchKernelGGL((k_spgemm_unit<T, MU_SYMBOLIC, 1, 1, 1>), grid, dim3(256), 0, ctx().stream, as, rb.ptr(4), r0, nr, none, 0);
                    ctx().stats.kernel_launches += 1;
                }
            };
            if (wg > 1) {
                launch_sym(rb.count(4), wg, MU_SYM_PLEN, nullptr);
                if (n_long_rows) launch_sym(n_long_rows, 1, 0, long_list.p);
                sync_stream();  // (the list is released at the end of this scope)
            } else {
                launch_sym(rb.count(4), 1, 0, nullptr);
            }
            hipLaunchKernelGGL(k_unit_prefix, dim3((unsigned)ceil_div(rb.count(4), 4)), dim3(256), 0, ctx().stream, a.wcnt, a.n_win, rb.ptr(4),
                               rb.count(4), a.row_nnz, a.wrow, a.class_count, unit_limits(false), FG);
            ctx().stats.kernel_launches += 1;
        }
    } else if (rb.count(4) && !NUMERIC && a.n <= (1 << 24) && !(ctx().debug_flags & 256)) {
        if (a.n <= (1 << 18)) hipLaunchKernelGGL((k_spgemm_sym_lds<4096, MM_BLOCK>), dim3((unsigned)rb.count(4)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(4));
        else  // (128 KiB of bitmap: one workgroup per CU, so make it a big one)
            hipLaunchKernelGGL((k_spgemm_sym_lds<16384, 1024>), dim3((unsigned)rb.count(4)), dim3(1024), 0, ctx().stream, a, rb.ptr(4));
        ctx().stats.kernel_launches += 1;
    } else if (rb.count(4) && NUMERIC && a.woff) {
        hipLaunchKernelGGL((k_spgemm_win<T>), dim3((unsigned)rb.count(4)), dim3(MM_WIN_BLOCK), 0, ctx().stream, a, rb.ptr(4));
        ctx().stats.kernel_launches += 1;
    } else if (rb.count(4)) {
        // dense accumulators in HBM: one slice per resident workgroup (8 per CU), as many as fit in ~32 GiB
        const int64_t words = (int64_t)bits_words64((uint64_t)a.n);
        const int64_t slice_bytes = words * 8 + (NUMERIC ? words * 64 * (int64_t)sizeof(W) : 0);
        int64_t G = std::min<int64_t>(std::min<int64_t>((int64_t)ctx().num_cus * 8, rb.count(4)), std::max<int64_t>(1, (32ll << 30) / slice_bytes));
        DevBuf<uint64_t> bits((size_t)(G * words), true);
        DevBuf<W> vals(NUMERIC ? (size_t)(G * words * 64) : 1);
        if (NUMERIC) {
            const int64_t nv = G * words * 64;
            hipLaunchKernelGGL((k_fill_ident<W>), dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0, ctx().stream, vals.p, nv,
                               monoid_identity<T, W>(a.monoid));
        }
        a.spa_bits = bits.p;
        a.spa_vals = vals.p;
        a.spa_words = words;
        hipLaunchKernelGGL((k_spgemm_spa<T, NUMERIC>), dim3((unsigned)G), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(4), rb.count(4));
        ctx().stats.kernel_launches += 2;
        sync_stream();  // slices are freed when this scope ends
    }
    GRB_HIP(hipGetLastError());
}

// GrX_mxm_streamed multiplies row batches of A by the same B: the window offset table of B (19 ms to build at scale 22) is
// built by the first batch that needs it and kept until the loop ends.
struct WoffKeep {
    bool on = false;
    const void *Bp = nullptr;
    int32_t *woff = nullptr;
};
static WoffKeep g_woff_keep;

// T = A (+.x) B in the semiring's type; returns a fresh matrix (sorted rows)
// forbidden: the pattern of a complemented mask to fuse (positions T must not hold), *fused tells whether it was honoured --
// the fused path needs the window offset tables (hash / unit / window kernels only); when it is not taken T is the full product
// and the caller's write rule applies the mask.
struct ForbiddenPattern {
    const int64_t *p = nullptr;
    const int32_t *j = nullptr;
};
template <typename T>
static GB_Matrix_opaque *spgemm(GB_Matrix_opaque *A, const void *Ax, GB_Matrix_opaque *B, const void *Bx, int st, int monoid,
                                int mult, const ForbiddenPattern *forbidden = nullptr, bool *fused = nullptr,
                                unsigned long long *csum_slots = nullptr)
{
    if (fused) *fused = false;
    GB_Matrix_opaque *Tm = matrix_new(type_of_code(st), A->nrows, B->ncols);
    if (A->nvals == 0 || B->nvals == 0) return Tm;
    try {
        const int64_t m = (int64_t)A->nrows, nnzA = A->nvals;
        MxmArgs a{};
        a.m = m;
        a.n = (int64_t)B->ncols;
        a.Ap = A->d_ptr; a.Aj = A->d_col; a.Ax = Ax; a.a_iso = A->iso ? 1 : 0;
        a.Bp = B->d_ptr; a.Bj = B->d_col; a.Bx = Bx; a.b_iso = B->iso ? 1 : 0;
        a.monoid = monoid;
        a.mult = mult;
        a.need_a = !(mult == OP_PAIR || mult == OP_SECOND);
        a.need_b = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
        a.csum = csum_slots;  // (streamed product: the numeric kernels add the values they store)
        a.xcd_map = ctx().mxm_xcd_map;
#ifdef GRB_ABLATE
        a.abl = (ctx().debug_flags >> 20) & 0x7FF;
#endif
        // 1. flops per stored entry of A, scanned
        DevBuf<int64_t> F(nnzA + 1);
        hipLaunchKernelGGL(k_nnz_flops, dim3((unsigned)ceil_div(nnzA + 1, 256)), dim3(256), 0, ctx().stream, A->d_col, nnzA,
                           B->d_ptr, F.p);
        prim_exclusive_sum_i64(F.p, F.p, nnzA + 1);
        int64_t flops = 0;
        d2h(&flops, F.p + nnzA, sizeof(int64_t));
        ctx().stats.flops = flops;
        // 2./3. symbolic
        DevBuf<int64_t> rownnz(m + 1, true);
        a.row_nnz = rownnz.p;
        // column-window offsets of B's rows (heavy rows walk the windows in both passes): n_B x (windows + 1) int32, per call
        DevBuf<int32_t> woff(0), wcnt(0), wrow(0), wbm(0);
        DevBuf<unsigned long long> bm_pool(0), bm_cur(16 * MU_POOLS), class_cnt(MU_CSLOTS * MU_NCLS);
        std::vector<unsigned long long> class_host(MU_CSLOTS * MU_NCLS);
        const int64_t n_win = ceil_div((int64_t)B->ncols, MM_WIN);
        const int64_t woff_entries = (int64_t)B->nrows * (n_win + 1);
        DevBuf<int32_t> cm_woff(0);
        bool fuse = forbidden && forbidden->p && !(ctx().debug_flags & 256) && ctx().mxm_heavy_kernel == 1 && n_win < 65536 &&
                    woff_entries * 4 <= (8ll << 30) && m * (n_win + 1) * 4 <= (8ll << 30);
        auto ensure_woff = [&]() {
            if (a.woff || (ctx().debug_flags & 256) || woff_entries * 4 > (8ll << 30)) return;
            if (g_woff_keep.on && g_woff_keep.Bp == (const void *)B->d_ptr && g_woff_keep.woff) {  // (a row-batched product: B is the same in every batch)
                a.woff = g_woff_keep.woff;
                a.n_win = (int)n_win;
                return;
            }
            dev_free(woff.p);
            woff.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)woff_entries);
            launch_window_offsets((const int64_t *)B->d_ptr, (const int32_t *)B->d_col, (int64_t)B->nrows, (int)n_win, woff.p,
                                  (const int32_t *)nullptr, (unsigned long long *)nullptr, UnitLimits{});
            a.woff = woff.p;
            a.n_win = (int)n_win;
            if (g_woff_keep.on) {  // hand the table to the batch loop, which frees it at its end
                g_woff_keep.Bp = (const void *)B->d_ptr;
                g_woff_keep.woff = woff.release();
            }
        };
        {
            RowBins rb(m);
            // rows beyond the LDS hash tables are walked as (row, window) units; with the unit kernels at hand the hash kernels
            // only keep the rows of up to max(4096, 32 per window) products (units of a handful of products do not pay)
            const bool units_ok = ctx().mxm_heavy_kernel == 1 && !(ctx().debug_flags & 256) && woff_entries * 4 <= (8ll << 30) &&
                                  n_win < 65536;
            // (never above 4096: a row the hash kernels count must fit the numeric hash table, nnz <= flops <= 4096 -- or it falls to the
            //  1024-thread window walk, a quarter of the scale-22 run while the limit was 32 x 256 windows = 8192)
            // round 5: the units walk groups of win_f windows (option mxm_window_groups: 1 / 2 / 4 / 8; 0 = 2 beyond 64 windows per row).
            // Measured (profiles/r05/mxm_window_groups.txt; ms per product, groups 1 / 2 / 4 / 8): scale 20 (64 windows) 132.9 / 145.6 /
            // 158.1 / 214.5, scale 22 (256 windows) 1565 / 1432 / 1505 / 2079, scale 21 456 / 447 -- the units of a wider matrix do gain from
            // wider windows, but a group's bitmap and its per-lane words cost occupancy (symbolic unit: 8 / 8 / 4 / 2 wavefronts per SIMD,
            // the one-wavefront numeric class 5 / 3 / 2 / 1) faster than they save units: pairs beyond 64 windows, single windows below.
            int win_f = 1;
            if (units_ok) {
                const int64_t want = ctx().mxm_window_groups;
                if (want > 0) win_f = (int)want;
                else win_f = n_win <= 64 ? 1 : 2;
            }
            a.win_f = win_f;
            const int64_t n_groups = ceil_div(n_win, (int64_t)win_f);
            const int64_t sym_b3 = units_ok ? std::min<int64_t>(4096, std::max<int64_t>(ctx().mxm_unit_min_flops, ctx().mxm_unit_min_per_window * n_groups)) : 16384;
            make_bins(rb, A->d_ptr, F.p, m, rownnz.p, 128, 1024, sym_b3, nullptr, fuse ? forbidden->p : nullptr);
            GRB_HIP(hipMemsetAsync(rownnz.p, 0, sizeof(int64_t) * (m + 1), ctx().stream));
            if (rb.count(4) && units_ok && rb.count(4) * (n_win + 1) * 4 <= (8ll << 30)) {
                ensure_woff();
                if (a.woff) {
                    dev_free(wcnt.p);
                    wcnt.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(rb.count(4) * (n_win + 1)));
                    dev_free(wrow.p);
                    wrow.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)m);
                    GRB_HIP(hipMemsetAsync(wrow.p, 0xFF, sizeof(int32_t) * (size_t)m, ctx().stream));
                    a.wcnt = wcnt.p;
                    a.wrow = wrow.p;
                    GRB_HIP(hipMemsetAsync(class_cnt.p, 0, sizeof(unsigned long long) * MU_CSLOTS * MU_NCLS, ctx().stream));
                    a.class_count = class_cnt.p;
                    // the bitmap pool: as many bitmaps as the option allows and a quarter of the free memory holds
                    size_t free_b = 0, total_b = 0;
                    GRB_HIP(hipMemGetInfo(&free_b, &total_b));
                    const int64_t bm_bytes = MM_WIN / 8;
                    const int64_t cap = std::min<int64_t>({rb.count(4) * n_win, (int64_t)(ctx().mxm_bitmap_pool_mb << 20) / bm_bytes,
                                                           (int64_t)(free_b / 4) / bm_bytes, ctx().mxm_bitmap_pool_cap});
                    if (cap > 0) {
                        dev_free(wbm.p);
                        wbm.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(rb.count(4) * n_win));
                        dev_free(bm_pool.p);
                        bm_pool.p = (unsigned long long *)dev_alloc((size_t)(cap * bm_bytes));
                        GRB_HIP(hipMemsetAsync(bm_cur.p, 0, sizeof(unsigned long long) * 16 * MU_POOLS, ctx().stream));
                        a.wbm = wbm.p;
                        a.bm_pool = bm_pool.p;
                        a.bm_cursor = bm_cur.p;
                        a.bm_pools = cap >= 64 * MU_POOLS ? MU_POOLS : 1;
                        a.bm_cap = cap / a.bm_pools;
                        a.bm_min_cnt = ctx().mxm_bitmap_min_cnt;
                    }
                }
            }
            // the fused complemented mask: every kernel this product will run must know it (the dense-accumulator fall-backs do not)
            fuse = fuse && (rb.count(4) == 0 || (a.woff && a.wrow && a.wcnt));
            if (fuse) {
                dev_free(cm_woff.p);
                cm_woff.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(m * (n_win + 1)));
                launch_window_offsets(forbidden->p, forbidden->j, m, (int)n_win, cm_woff.p, (const int32_t *)nullptr,
                                      (unsigned long long *)nullptr, UnitLimits{});
                a.CMp = forbidden->p;
                a.CMj = forbidden->j;
                a.cm_woff = cm_woff.p;
                a.n_win = (int)n_win;
                if (fused) *fused = true;
            }
            run_bins<T, false>(a, rb);
        }
        // 4. row pointers of T (counts stay in rownnz for the numeric binning)
        int64_t *Tp = (int64_t *)dev_alloc(sizeof(int64_t) * (m + 1));
        prim_exclusive_sum_i64(rownnz.p, Tp, m + 1);
        int64_t nnzT = 0;
        d2h(&nnzT, Tp + m, sizeof(int64_t));
        if (a.class_count) {  // (the symbolic pass walked units: their class counts come with the same synchronisation)
            d2h(class_host.data(), class_cnt.p, sizeof(unsigned long long) * MU_CSLOTS * MU_NCLS);
            a.class_known = class_host.data();
        }
        Tm->d_ptr = Tp;
        ctx().stats.out_nvals = nnzT;
        if (nnzT == 0) {
            dev_free(Tm->d_ptr);
            Tm->d_ptr = nullptr;
            return Tm;
        }
        Tm->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnzT);
        Tm->d_val = dev_alloc(sizeof(T) * (size_t)nnzT);
        Tm->nvals = nnzT;
        a.Tp = Tp;
        a.Tj = Tm->d_col;
        a.Tx = Tm->d_val;
        // 5. numeric, binned by the exact row sizes; rows above the LDS hash limit use LDS column windows when the
        //    offset table (n_B x (windows+1) int32) is affordable, else dense accumulators in HBM
        {
            RowBins rb(m);
            make_bins(rb, A->d_ptr, nullptr, m, rownnz.p, 128, 1024, 4096, a.wrow, fuse ? forbidden->p : nullptr);
            if (rb.count(4)) ensure_woff();
            run_bins<T, true>(a, rb);
            sync_stream();  // woff is released at the end of this scope
        }
    } catch (...) {
        matrix_free(Tm);
        throw;
    }
    return Tm;
}

// T = (A (+.x) B) restricted to the pattern of Mask (non-complemented): mask-driven, see k_spgemm_mhash / k_spgemm_mwin.
// Returns nullptr when the heavy-row path is unaffordable (offset table too large): the caller then takes the full product.
template <typename T>
static GB_Matrix_opaque *spgemm_masked(GB_Matrix_opaque *A, const void *Ax, GB_Matrix_opaque *B, const void *Bx, int st, int monoid,
                                       int mult, GB_Matrix_opaque *Mask, int64_t flops_total)
{
    GB_Matrix_opaque *Tm = matrix_new(type_of_code(st), A->nrows, B->ncols);
    if (A->nvals == 0 || B->nvals == 0 || Mask->nvals == 0) return Tm;
    try {
        const int64_t m = (int64_t)A->nrows, nnzM = Mask->nvals;
        MxmArgs a{};
        a.m = m;
        a.n = (int64_t)B->ncols;
        a.Ap = A->d_ptr; a.Aj = A->d_col; a.Ax = Ax; a.a_iso = A->iso ? 1 : 0;
        a.Bp = B->d_ptr; a.Bj = B->d_col; a.Bx = Bx; a.b_iso = B->iso ? 1 : 0;
        a.monoid = monoid;
        a.mult = mult;
        a.need_a = !(mult == OP_PAIR || mult == OP_SECOND);
        a.need_b = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
        a.Mp = matrix_rowptr(Mask);
        a.Mj = Mask->d_col;
        a.xcd_map = ctx().mxm_xcd_map;
        DevBuf<T> cap_val(nnzM);
        DevBuf<unsigned char> cap_hit(nnzM, true);
        a.cap_val = cap_val.p;
        a.cap_hit = cap_hit.p;
        // rows binned by the length of their mask row (rows of A without entries produce nothing); rows with many products
        // (and a mask row) are walked as (row, window) units whatever the length of their mask row: urow[row] = row for them
        DevBuf<int64_t> size(m + 1);
        hipLaunchKernelGGL(k_mask_sizes, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr,
                           a.Mp, m, size.p);
        const int64_t n_win = ceil_div((int64_t)B->ncols, MM_WIN);
        const int64_t woff_entries = (int64_t)B->nrows * (n_win + 1), mwoff_entries = m * (n_win + 1);
        // (small products: the classification's two host round trips cost more than the units save -- scale 12: 0.74 against 0.47 ms)
        const bool units_ok = ctx().mxm_heavy_kernel == 1 && !(ctx().debug_flags & 256) && woff_entries * 4 <= (8ll << 30) &&
                              mwoff_entries * 4 <= (8ll << 30) && n_win < 65536 && Mask->ncols == B->ncols &&
                              flops_total >= ctx().mxm_masked_units_min_flops;
        DevBuf<int32_t> urow(units_ok ? m : 0), mwoff(0);
        if (units_ok) {
            const int64_t nnzA = A->nvals;
            DevBuf<int64_t> F(nnzA + 1);
            hipLaunchKernelGGL(k_nnz_flops, dim3((unsigned)ceil_div(nnzA + 1, 256)), dim3(256), 0, ctx().stream, A->d_col, nnzA, B->d_ptr, F.p);
            prim_exclusive_sum_i64(F.p, F.p, nnzA + 1);
            hipLaunchKernelGGL(k_mask_unit_rows, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr,
                               (const int64_t *)F.p, a.Mp, m, std::max<int64_t>(ctx().mxm_unit_min_flops, ctx().mxm_unit_min_per_window * n_win), urow.p);
            sync_stream();  // (F is freed at the end of this scope)
        }
        RowBins rb(m);
        make_bins(rb, A->d_ptr, nullptr, m, size.p, 128, 1024, 4096, units_ok ? urow.p : nullptr);
        DevBuf<int32_t> woff(0);
        if (rb.count(4)) {
            if (woff_entries * 4 > (8ll << 30)) {
                matrix_free(Tm);
                return nullptr;
            }
            dev_free(woff.p);
            woff.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)woff_entries);
            launch_window_offsets((const int64_t *)B->d_ptr, (const int32_t *)B->d_col, (int64_t)B->nrows, (int)n_win, woff.p,
                                  (const int32_t *)nullptr, (unsigned long long *)nullptr, UnitLimits{});
            a.woff = woff.p;
            a.n_win = (int)n_win;
            if (units_ok) {  // the mask rows' window offsets play the part of the symbolic pass's counts, classes counted on the way
                dev_free(mwoff.p);
                mwoff.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)mwoff_entries);
                DevBuf<unsigned long long> ccnt(MU_CSLOTS * MU_NCLS, true);
                launch_window_offsets(a.Mp, a.Mj, m, (int)n_win, mwoff.p, (const int32_t *)urow.p, ccnt.p, unit_limits(true));
                std::vector<unsigned long long> known(MU_CSLOTS * MU_NCLS);
                d2h(known.data(), ccnt.p, sizeof(unsigned long long) * MU_CSLOTS * MU_NCLS);
                a.wcnt = mwoff.p;
                a.wrow = urow.p;
                launch_unit_classes<T, MU_MASKED>(a, rb.ptr(4), rb.count(4), known.data());
            }
        }
        if (rb.count(1)) hipLaunchKernelGGL((k_spgemm_mhash<T, 256>), dim3((unsigned)rb.count(1)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(1));
        if (rb.count(2)) hipLaunchKernelGGL((k_spgemm_mhash<T, 2048>), dim3((unsigned)rb.count(2)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(2));
        if (rb.count(3)) hipLaunchKernelGGL((k_spgemm_mhash<T, 8192>), dim3((unsigned)rb.count(3)), dim3(MM_BLOCK), 0, ctx().stream, a, rb.ptr(3));
        if (rb.count(4)) hipLaunchKernelGGL((k_spgemm_mwin<T>), dim3((unsigned)rb.count(4)), dim3(MM_WIN_BLOCK), 0, ctx().stream, a, rb.ptr(4));
        GRB_HIP(hipGetLastError());
        ctx().stats.kernel_launches += 5;
        // compaction: position of every hit, row pointers, scatter
        DevBuf<int64_t> pos(nnzM + 1);
        hipLaunchKernelGGL(k_hits_to_i64, dim3((unsigned)ceil_div(nnzM + 1, 256)), dim3(256), 0, ctx().stream,
                           (const unsigned char *)cap_hit.p, nnzM, pos.p);
        prim_exclusive_sum_i64(pos.p, pos.p, nnzM + 1);
        int64_t nnzT = 0;
        d2h(&nnzT, pos.p + nnzM, sizeof(int64_t));
        ctx().stats.out_nvals = nnzT;
        if (nnzT > 0) {
            Tm->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (m + 1));
            hipLaunchKernelGGL(k_cap_rowptr, dim3((unsigned)ceil_div(m + 1, 256)), dim3(256), 0, ctx().stream, a.Mp,
                               (const int64_t *)pos.p, m, Tm->d_ptr);
            Tm->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnzT);
            Tm->d_val = dev_alloc(sizeof(T) * (size_t)nnzT);
            Tm->nvals = nnzT;
            hipLaunchKernelGGL((k_cap_scatter<T>), dim3((unsigned)ceil_div(nnzM, 256)), dim3(256), 0, ctx().stream,
                               (const unsigned char *)cap_hit.p, (const int64_t *)pos.p, a.Mj, (const T *)cap_val.p, nnzM, Tm->d_col,
                               (T *)Tm->d_val);
        }
        sync_stream();  // the temporaries are released at the end of this scope
    } catch (...) {
        matrix_free(Tm);
        throw;
    }
    return Tm;
}

// ---- the forbidden pattern of a VALUED complemented mask: its true entries, compacted (a structural mask forbids its whole
//      pattern, no copy) ------------------------------------------------------------------------------------------------------
template <typename TM>
__global__ void k_mask_truth(const TM *Mx, int m_iso, int64_t n, int64_t *flag)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) flag[p] = Mx[m_iso ? 0 : p] != (TM)0 ? 1 : 0;
    else if (p == n) flag[p] = 0;
}
__global__ void k_true_cols(const int64_t *pos, const int32_t *Mj, int64_t n, int32_t *out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && pos[p + 1] != pos[p]) out[pos[p]] = Mj[p];
}
static void true_pattern(GB_Matrix_opaque *Mask, DevBuf<int64_t> &ptr, DevBuf<int32_t> &col)
{
    const int64_t n = Mask->nvals, m = (int64_t)Mask->nrows;
    DevBuf<int64_t> pos(n + 1);
    const dim3 grid((unsigned)ceil_div(n + 1, 256)), block(256);
    const int iso = Mask->iso ? 1 : 0;
    switch (type_size(Mask->type->code)) {
    case 1: hipLaunchKernelGGL((k_mask_truth<uint8_t>), grid, block, 0, ctx().stream, (const uint8_t *)Mask->d_val, iso, n, pos.p); break;
    case 2: hipLaunchKernelGGL((k_mask_truth<uint16_t>), grid, block, 0, ctx().stream, (const uint16_t *)Mask->d_val, iso, n, pos.p); break;
    case 4:
        if (Mask->type->code == TC_FP32) hipLaunchKernelGGL((k_mask_truth<float>), grid, block, 0, ctx().stream, (const float *)Mask->d_val, iso, n, pos.p);
        else hipLaunchKernelGGL((k_mask_truth<uint32_t>), grid, block, 0, ctx().stream, (const uint32_t *)Mask->d_val, iso, n, pos.p);
        break;
    default:
        if (Mask->type->code == TC_FP64) hipLaunchKernelGGL((k_mask_truth<double>), grid, block, 0, ctx().stream, (const double *)Mask->d_val, iso, n, pos.p);
        else hipLaunchKernelGGL((k_mask_truth<uint64_t>), grid, block, 0, ctx().stream, (const uint64_t *)Mask->d_val, iso, n, pos.p);
        break;
    }
    prim_exclusive_sum_i64(pos.p, pos.p, n + 1);
    int64_t n_true = 0;
    d2h(&n_true, pos.p + n, sizeof(int64_t));
    dev_free(ptr.p);
    ptr.p = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(m + 1));
    hipLaunchKernelGGL(k_cap_rowptr, dim3((unsigned)ceil_div(m + 1, 256)), dim3(256), 0, ctx().stream, matrix_rowptr(Mask),
                       (const int64_t *)pos.p, m, ptr.p);
    dev_free(col.p);
    col.p = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(1, n_true));
    hipLaunchKernelGGL(k_true_cols, grid, block, 0, ctx().stream, (const int64_t *)pos.p, (const int32_t *)Mask->d_col, n, col.p);
    ctx().stats.kernel_launches += 3;
    sync_stream();  // (pos is released at the end of this scope)
}

// flops of A (+.x) B = sum over the entries A(i,k) of nnz(B(k,:))
static int64_t product_flops(GB_Matrix_opaque *A, GB_Matrix_opaque *B)
{
    if (A->nvals == 0 || B->nvals == 0) return 0;
    DevBuf<int64_t> F(A->nvals + 1);
    hipLaunchKernelGGL(k_nnz_flops, dim3((unsigned)ceil_div(A->nvals + 1, 256)), dim3(256), 0, ctx().stream, A->d_col, A->nvals, B->d_ptr,
                       F.p);
    prim_exclusive_sum_i64(F.p, F.p, A->nvals + 1);
    int64_t flops = 0;
    d2h(&flops, F.p + A->nvals, sizeof(int64_t));
    return flops;
}

struct MDesc {
    bool replace = false, comp = false, structure = false, t0 = false, t1 = false;
};

static void take_storage(GB_Matrix_opaque *C, GB_Matrix_opaque *src)
{
    matrix_release_storage(C);
    C->d_ptr = src->d_ptr; C->d_col = src->d_col; C->d_val = src->d_val;
    C->nvals = src->nvals; C->iso = src->iso; C->owns = true;
    src->d_ptr = nullptr; src->d_col = nullptr; src->d_val = nullptr; src->nvals = 0;
}

// C<Mask, replace> = accum(C, T) for a product / copy T already in C's type (sorted rows); T's storage is taken when nothing masks
// or accumulates.  Shared by GrB_mxm and GrB_transpose.
void matrix_apply_write_rule(GB_Matrix_opaque *C, GB_Matrix_opaque *Mask, const GB_BinaryOp_opaque *accum, GB_Matrix_opaque *Tm,
                             bool replace, bool comp, bool structure)
{
    struct { bool replace, comp, structure; } f{replace, comp, structure};
    if (!Mask && !accum) {
        take_storage(C, Tm);  // C = T (aliasing with A/B is safe: T is a fresh object)
    } else {
        const int64_t m = (int64_t)C->nrows;
        const int64_t *Tp = matrix_rowptr(Tm);
        const int acc_op = accum ? canonical_op(C->type->code, accum->op) : -1;
        const int64_t *Mp = Mask ? matrix_rowptr(Mask) : nullptr;
        GB_Matrix_opaque *N = matrix_new(C->type, C->nrows, C->ncols);
        if (ctx().mat_write_kernel == 1) {  // a wavefront per row / per column piece of a long row (grb_mxm_write.inc)
            try {
                WriteArgs wa{};
                wa.m = m;
                wa.Cp = (const int64_t *)C->d_ptr; wa.Cj = (const int32_t *)C->d_col; wa.Cx = C->d_val; wa.c_iso = C->iso ? 1 : 0;
                wa.Tp = Tp; wa.Tj = (const int32_t *)Tm->d_col; wa.Tx = Tm->d_val;
                wa.Mp = Mp; wa.Mj = Mask ? (const int32_t *)Mask->d_col : nullptr; wa.Mx = Mask ? (const void *)Mask->d_val : nullptr;
                wa.m_type = Mask ? Mask->type->code : 0; wa.m_iso = Mask && Mask->iso ? 1 : 0; wa.has_mask = Mask ? 1 : 0;
                wa.m_struct = f.structure ? 1 : 0; wa.m_comp = f.comp ? 1 : 0; wa.accum = acc_op; wa.replace = f.replace ? 1 : 0;
                const int64_t ncols = (int64_t)C->ncols;
                const int pieces = (int)std::min<int64_t>(WR_MAX_PIECES, std::max<int64_t>(1, ceil_div(ncols, (int64_t)16384)));
                wa.piece_cols = (int)std::min<int64_t>(0x7fffffff, ceil_div(ncols, (int64_t)pieces));
                DevBuf<int64_t> ubase(m + 1);
                hipLaunchKernelGGL(k_write_units_per_row, dim3((unsigned)ceil_div(m + 1, 256)), dim3(256), 0, ctx().stream, wa.Cp, Tp, m, pieces, ubase.p);
                prim_exclusive_sum_i64(ubase.p, ubase.p, m + 1);
                int64_t n_units = 0, nnzN = 0;
                d2h(&n_units, ubase.p + m, sizeof(int64_t));
                if (n_units > 0) {
                    DevBuf<int32_t> urow(n_units);
                    hipLaunchKernelGGL(k_write_unit_rows, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream, (const int64_t *)ubase.p, m, urow.p);
                    DevBuf<int64_t> uoff(n_units + 1, true);
                    wa.ubase = ubase.p; wa.urow = urow.p; wa.n_units = n_units; wa.ucount = uoff.p;
                    const dim3 grid((unsigned)ceil_div(n_units, 4)), block(256);
                    GRB_DISPATCH_TYPE(C->type->code, TW, { hipLaunchKernelGGL((k_mat_write_wave<TW, false>), grid, block, 0, ctx().stream, wa); })
                    prim_exclusive_sum_i64(uoff.p, uoff.p, n_units + 1);
                    d2h(&nnzN, uoff.p + n_units, sizeof(int64_t));
                    if (nnzN) {
                        N->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (m + 1));
                        hipLaunchKernelGGL(k_write_rowptr, dim3((unsigned)ceil_div(m + 1, 256)), dim3(256), 0, ctx().stream, (const int64_t *)ubase.p,
                                           (const int64_t *)uoff.p, m, N->d_ptr);
                        N->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnzN);
                        N->d_val = dev_alloc(C->type->size * (size_t)nnzN);
                        wa.uoff = uoff.p; wa.Nj = N->d_col; wa.Nx = N->d_val;
                        GRB_DISPATCH_TYPE(C->type->code, TW, { hipLaunchKernelGGL((k_mat_write_wave<TW, true>), grid, block, 0, ctx().stream, wa); })
                    }
                    ctx().stats.kernel_launches += 5;
                    sync_stream();  // (the unit tables are released at the end of this scope)
                }
                N->nvals = nnzN;
                GRB_HIP(hipGetLastError());
                take_storage(C, N);
            } catch (...) {
                matrix_free(N);
                throw;
            }
            matrix_free(N);
            return;
        }
        try {
            DevBuf<int64_t> cnt(m + 1, true);
            int64_t *Np = (int64_t *)dev_alloc(sizeof(int64_t) * (m + 1));
            N->d_ptr = Np;
            int64_t nnzN = 0;
            GRB_DISPATCH_TYPE(C->type->code, TW, {
                const dim3 grid((unsigned)ceil_div(m, 256)), block(256);
                hipLaunchKernelGGL((k_mat_write<TW, false>), grid, block, 0, ctx().stream, m, (const int64_t *)C->d_ptr,
                                   (const int32_t *)C->d_col, (const TW *)C->d_val, C->iso ? 1 : 0, Tp,
                                   (const int32_t *)Tm->d_col, (const TW *)Tm->d_val, Mp,
                                   Mask ? (const int32_t *)Mask->d_col : nullptr, Mask ? (const void *)Mask->d_val : nullptr,
                                   Mask ? Mask->type->code : 0, Mask && Mask->iso ? 1 : 0, Mask ? 1 : 0,
                                   f.structure ? 1 : 0, f.comp ? 1 : 0, acc_op, f.replace ? 1 : 0, cnt.p,
                                   (const int64_t *)nullptr, (int32_t *)nullptr, (TW *)nullptr);
                prim_exclusive_sum_i64(cnt.p, Np, m + 1);
                d2h(&nnzN, Np + m, sizeof(int64_t));
                if (nnzN) {
                    N->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnzN);
                    N->d_val = dev_alloc(sizeof(TW) * (size_t)nnzN);
                    hipLaunchKernelGGL((k_mat_write<TW, true>), grid, block, 0, ctx().stream, m, (const int64_t *)C->d_ptr,
                                       (const int32_t *)C->d_col, (const TW *)C->d_val, C->iso ? 1 : 0, Tp,
                                       (const int32_t *)Tm->d_col, (const TW *)Tm->d_val, Mp,
                                       Mask ? (const int32_t *)Mask->d_col : nullptr,
                                       Mask ? (const void *)Mask->d_val : nullptr, Mask ? Mask->type->code : 0,
                                       Mask && Mask->iso ? 1 : 0, Mask ? 1 : 0, f.structure ? 1 : 0, f.comp ? 1 : 0, acc_op,
                                       f.replace ? 1 : 0, (int64_t *)nullptr, (const int64_t *)Np, N->d_col, (TW *)N->d_val);
                }
            })
            N->nvals = nnzN;
            ctx().stats.kernel_launches += 2;
            if (nnzN == 0) {
                dev_free(N->d_ptr);
                N->d_ptr = nullptr;
            }
            sync_stream();
            take_storage(C, N);
        } catch (...) {
            matrix_free(N);
            throw;
        }
        matrix_free(N);
    }
}

static void mxm_core(GB_Matrix_opaque *C, GB_Matrix_opaque *Mask, const GB_BinaryOp_opaque *accum,
                     const GB_Semiring_opaque *sr, GB_Matrix_opaque *A, GB_Matrix_opaque *B, MDesc f)
{
    GB_Matrix_opaque *Ae = f.t0 ? matrix_transpose_cached(A) : A;
    GB_Matrix_opaque *Be = f.t1 ? matrix_transpose_cached(B) : B;
    if (Ae->ncols != Be->nrows) fail(GrB_DIMENSION_MISMATCH, "mxm: inner dimensions " + std::to_string(Ae->ncols) + " and " + std::to_string(Be->nrows) + " differ");
    if (C->nrows != Ae->nrows || C->ncols != Be->ncols) fail(GrB_DIMENSION_MISMATCH, "mxm: output is " + std::to_string(C->nrows) + "x" + std::to_string(C->ncols) + ", product is " + std::to_string(Ae->nrows) + "x" + std::to_string(Be->ncols));
    if (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols)) fail(GrB_DIMENSION_MISMATCH, "mxm: mask shape does not match the output");
    if (accum && (accum->type != C->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "mxm: accum operator type must equal the output type");
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 3;
    if (!Mask && f.comp) {
        if (f.replace) matrix_release_storage(C);
        return;
    }
    const int st = sr->type;
    const int monoid = canonical_op(st, sr->monoid), mult = canonical_op(st, sr->mult);
    // operands in the semiring's type
    DevBuf<char> a_cast(0), b_cast(0);
    const void *Ax = Ae->d_val, *Bx = Be->d_val;
    if (Ae->nvals && Ae->type->code != st) {
        const int64_t nv = Ae->iso ? 1 : Ae->nvals;
        dev_free(a_cast.p);
        a_cast.p = (char *)dev_alloc(type_size(st) * (size_t)nv);
        cast_array(st, a_cast.p, Ae->type->code, Ae->d_val, nv);
        Ax = a_cast.p;
    }
    if (Be->nvals && Be->type->code != st) {
        const int64_t nv = Be->iso ? 1 : Be->nvals;
        dev_free(b_cast.p);
        b_cast.p = (char *)dev_alloc(type_size(st) * (size_t)nv);
        cast_array(st, b_cast.p, Be->type->code, Be->d_val, nv);
        Bx = b_cast.p;
    }
    GB_Matrix_opaque *Tm = nullptr;
    // A non-complemented mask bounds the useful part of the product by its own pattern: take the mask-driven path when the
    // full product would cost clearly more than walking it with the mask rows in LDS (always / never: option mxm_mask_mode)
    if (Mask && !f.comp && ctx().mxm_mask_mode != 0 && Mask->nvals > 0 && Mask->nvals < 0x7fffffffll * 8) {
        const int64_t flops = product_flops(Ae, Be);
        if (ctx().mxm_mask_mode == 2 || flops > 4 * (Mask->nvals + Ae->nvals)) {
            GRB_DISPATCH_TYPE(st, T, { Tm = spgemm_masked<T>(Ae, Ax, Be, Bx, st, monoid, mult, Mask, flops); })
            if (Tm) {
                ctx().stats.method = 4;
                ctx().stats.flops = flops;
            }
        }
    }
    // A complemented mask is fused into the product: the positions it forbids never enter T (they would only be dropped by the
    // write rule, after a second pass over all of T)
    bool fused = false;
    DevBuf<int64_t> fp_ptr(0);
    DevBuf<int32_t> fp_col(0);
    if (!Tm && Mask && f.comp && ctx().mxm_mask_mode != 0 && Mask->nvals > 0) {
        ForbiddenPattern fp;
        if (f.structure) {
            fp.p = matrix_rowptr(Mask);
            fp.j = Mask->d_col;
        } else {
            true_pattern(Mask, fp_ptr, fp_col);
            fp.p = fp_ptr.p;
            fp.j = fp_col.p;
        }
        GRB_DISPATCH_TYPE(st, T, { Tm = spgemm<T>(Ae, Ax, Be, Bx, st, monoid, mult, &fp, &fused); })
        if (fused) ctx().stats.method = 7;
    }
    if (!Tm) GRB_DISPATCH_TYPE(st, T, { Tm = spgemm<T>(Ae, Ax, Be, Bx, st, monoid, mult); })
    try {
        // T in the output type
        if (Tm->nvals && Tm->type->code != C->type->code) {
            void *cv = dev_alloc(C->type->size * (size_t)Tm->nvals);
            cast_array(C->type->code, cv, st, Tm->d_val, Tm->nvals);
            dev_free(Tm->d_val);
            Tm->d_val = cv;
        }
        Tm->type = C->type;
        // a mask-driven product lies inside the mask's pattern: under a structural mask, with nothing to accumulate into and
        // nothing of C to keep (C empty, or replace), the write rule is C = T
        // (the same for a fused complemented mask, valued or structural: T holds no forbidden position, and what C held at the
        //  forbidden ones is either nothing or deleted by replace)
        if (ctx().stats.method == 4 && !accum && f.structure && !f.comp && (C->nvals == 0 || f.replace)) take_storage(C, Tm);
        else if (fused && !accum && (C->nvals == 0 || f.replace)) take_storage(C, Tm);
        else matrix_apply_write_rule(C, Mask, accum, Tm, f.replace, f.comp, f.structure);
    } catch (...) {
        matrix_free(Tm);
        throw;
    }
    matrix_free(Tm);
    if (ctx().blocking) sync_stream();
}

// ---- row-batched product with the output streamed through a bounded buffer ---------------------------------------------------
// C = A (+.x) B of a power-law graph can outgrow any one GPU (R-MAT scale 22: 7.5e10 entries, 900 GB): the product is then run
// over row batches of A whose products fit `budget_bytes`, each batch through the full two-pass pipeline (symbolic + numeric,
// sorted rows), and what leaves a batch is its entry count and a checksum of its values -- the single-GPU denominator of the
// row-sharded multi-GPU product, where every rank holds only its rows of C.
__global__ void k_rebase_rowptr(const int64_t *Ap, int64_t r0, int64_t rows, int64_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= rows) out[i] = Ap[r0 + i] - Ap[r0];
}
template <typename T>
__global__ void k_value_checksum(const T *x, int64_t n, unsigned long long *sum)
{
    // (16-byte loads: the sum re-reads every value of a batch's product -- 575 GB over the scale-22 run; with 8-byte loads it ran at 3.9 TB/s)
    unsigned long long s = 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(T) == 8) {
        const int64_t head = (((uintptr_t)x & 15u) && n > 0) ? 1 : 0;  // (one element up to the 16-byte boundary)
        const int64_t pairs = (n - head) / 2;
        const ulonglong2 *x2 = (const ulonglong2 *)(x + head);
        for (int64_t i = tid; i < pairs; i += nth) {
            const ulonglong2 v = x2[i];
            if constexpr (std::is_floating_point<T>::value) s += (unsigned long long)(long long)__builtin_bit_cast(T, v.x) + (unsigned long long)(long long)__builtin_bit_cast(T, v.y);
            else s += v.x + v.y;
        }
        if (tid == 0) {
            if (head) s += std::is_floating_point<T>::value ? (unsigned long long)(long long)x[0] : (unsigned long long)x[0];
            for (int64_t i = head + 2 * pairs; i < n; i++) s += std::is_floating_point<T>::value ? (unsigned long long)(long long)x[i] : (unsigned long long)x[i];
        }
    } else {
        for (int64_t i = tid; i < n; i += nth) {
            if constexpr (std::is_floating_point<T>::value) s += (unsigned long long)(long long)x[i];
            else s += (unsigned long long)x[i];
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(sum, s);
}
// first row r1 > r0 such that flops of rows [r0, r1) stay within the share (at least one row)
__global__ void k_batch_end(const int64_t *Ap, const int64_t *F, int64_t m, int64_t r0, int64_t max_flops, int64_t *out)
{
    if (threadIdx.x || blockIdx.x) return;
    const int64_t base = F[Ap[r0]];
    int64_t lo = r0 + 1, hi = m;
    while (lo < hi) {  // largest r1 with F[Ap[r1]] - base <= max_flops
        const int64_t mid = (lo + hi + 1) >> 1;
        if (F[Ap[mid]] - base <= max_flops) lo = mid;
        else hi = mid - 1;
    }
    out[0] = lo;
    out[1] = F[Ap[lo]] - base;
}

}  // namespace grb

using namespace grb;

extern "C" GrB_Info GrX_mxm_streamed(const GrB_Semiring semiring, const GrB_Matrix A, const GrB_Matrix B, uint64_t budget_bytes,
                                     uint64_t *nvals_out, uint64_t *checksum_out, uint64_t *flops_out, uint64_t *batches_out)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    check_matrix(B, "B");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    if (A->ncols != B->nrows) fail(GrB_DIMENSION_MISMATCH, "mxm: inner dimensions differ");
    const int st = semiring->type;
    if (A->type->code != st || B->type->code != st) fail(GrB_DOMAIN_MISMATCH, "GrX_mxm_streamed: operands must have the semiring's type");
    const int monoid = canonical_op(st, semiring->monoid), mult = canonical_op(st, semiring->mult);
    uint64_t nv = 0, fl = 0, nb = 0;
    // the checksum of the product's values: folded into the numeric kernels' stores (round 5; MM_CSUM_SLOTS counters, a 128-byte line
    // apart) -- or, with mxm_checksum_pass = 1, by a pass of its own over every batch's product (round 4: 7 % of the scale-22 run)
    const bool sum_pass = ctx().mxm_checksum_pass != 0;
    DevBuf<unsigned long long> csum((size_t)MM_CSUM_SLOTS * 16, true);
    const int64_t m = (int64_t)A->nrows;
    if (A->nvals && B->nvals) {
        DevBuf<int64_t> F(A->nvals + 1);
        hipLaunchKernelGGL(k_nnz_flops, dim3((unsigned)ceil_div(A->nvals + 1, 256)), dim3(256), 0, ctx().stream, A->d_col, A->nvals, B->d_ptr, F.p);
        prim_exclusive_sum_i64(F.p, F.p, A->nvals + 1);
        // a product entry costs 4 + sizeof(T) bytes, and a row of T never has more entries than multiplies
        const int64_t per_entry = 4 + (int64_t)type_size(st);
        const int64_t max_flops = std::max<int64_t>(1, (int64_t)(budget_bytes / (uint64_t)per_entry));
        DevBuf<int64_t> cut(2);
        struct KeepGuard {  // B's window offsets live from the first batch that builds them to the end of the loop
            KeepGuard() { g_woff_keep = WoffKeep{}; g_woff_keep.on = true; }
            ~KeepGuard()
            {
                (void)hipStreamSynchronize(ctx().stream);
                dev_free(g_woff_keep.woff);
                g_woff_keep = WoffKeep{};
            }
        } keep_guard;
        int64_t r0 = 0;
        while (r0 < m) {
            int64_t h[2];
            hipLaunchKernelGGL(k_batch_end, dim3(1), dim3(64), 0, ctx().stream, (const int64_t *)A->d_ptr, (const int64_t *)F.p, m, r0, max_flops, cut.p);
            d2h(h, cut.p, sizeof(h));
            const int64_t r1 = h[0], rows = r1 - r0;
            int64_t e0 = 0, e1 = 0;
            d2h(&e0, A->d_ptr + r0, 8);
            d2h(&e1, A->d_ptr + r1, 8);
            if (e1 > e0) {
                GB_Matrix_opaque *V = matrix_new(A->type, (uint64_t)rows, A->ncols);  // rows [r0, r1) of A: a view on its arrays
                GB_Matrix_opaque *Tm = nullptr;
                try {
                    V->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(rows + 1));
                    hipLaunchKernelGGL(k_rebase_rowptr, dim3((unsigned)ceil_div(rows + 1, 256)), dim3(256), 0, ctx().stream,
                                       (const int64_t *)A->d_ptr, r0, rows, V->d_ptr);
                    V->d_col = A->d_col + e0;
                    V->d_val = A->iso ? A->d_val : (void *)((char *)A->d_val + (size_t)e0 * A->type->size);
                    V->iso = A->iso;
                    V->nvals = e1 - e0;
                    V->owns = false;
                    GRB_DISPATCH_TYPE(st, T, {
                        Tm = spgemm<T>(V, V->d_val, B, B->d_val, st, monoid, mult, nullptr, nullptr, sum_pass ? nullptr : csum.p);
                        if (Tm->nvals && sum_pass) {
                            hipLaunchKernelGGL((k_value_checksum<T>), dim3((unsigned)(ctx().num_cus * 16)), dim3(256), 0, ctx().stream, (const T *)Tm->d_val, Tm->nvals,
                                               csum.p);
                        }
                    })
                    nv += (uint64_t)Tm->nvals;
                    fl += (uint64_t)h[1];
                    nb++;
                    sync_stream();
                } catch (...) {
                    if (Tm) matrix_free(Tm);
                    dev_free(V->d_ptr);
                    V->d_ptr = nullptr; V->d_col = nullptr; V->d_val = nullptr;
                    matrix_free(V);
                    throw;
                }
                matrix_free(Tm);
                dev_free(V->d_ptr);
                V->d_ptr = nullptr; V->d_col = nullptr; V->d_val = nullptr;
                matrix_free(V);
            }
            r0 = r1;
        }
    }
    unsigned long long hsum = 0;
    {
        std::vector<unsigned long long> slots((size_t)MM_CSUM_SLOTS * 16);
        d2h(slots.data(), csum.p, sizeof(unsigned long long) * slots.size());
        for (size_t i = 0; i < slots.size(); i += (sum_pass ? slots.size() : 16)) hsum += slots[i];
    }
    if (nvals_out) *nvals_out = nv;
    if (checksum_out) *checksum_out = hsum;
    if (flops_out) *flops_out = fl;
    if (batches_out) *batches_out = nb;
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 3;
    ctx().stats.flops = (int64_t)fl;
    ctx().stats.out_nvals = (int64_t)nv;
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                            const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_matrix(C, "C");
    if (Mask) check_matrix(Mask, "Mask");
    check_matrix(A, "A");
    check_matrix(B, "B");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    MDesc f;
    if (desc) { f.replace = desc->replace; f.comp = desc->comp; f.structure = desc->structure; f.t0 = desc->t0; f.t1 = desc->t1; }
    mxm_core(C, Mask, accum, semiring, A, B, f);
    GRB_CATCH(errp(C))
}

namespace grb {
void preload_mxm() { hipFuncAttributes at; (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_window_offsets_wave)); (void)hipGetLastError(); }
}  // namespace grb
