// grb_mxv.hip -- GrB_mxv / GrB_vxm: merge-path pull SpMV over a semiring with the GraphBLAS write
// rule (mask, accumulator, replace) fused into the kernel epilogue.
//
// Reference call sites (paths relative to /root/reference):
//   GrB_mxv  graphblas/core/matrix.py:2203-2262 (expression :2252-2259), dispatched core/base.py:496-503
//   GrB_vxm  graphblas/core/vector.py:1309-1378 (expression :1367-1375)
// The arithmetic replaced is SuiteSparse:GraphBLAS's GrB_mxv/GrB_vxm (not in /root/reference).
//
// Kernel design:
//   * Work = the merge of the row-end list (m items) with the nnz list (nnz items); every 256-thread
//     workgroup owns TILE = 256*IPT consecutive merge items, so tiles are balanced no matter how
//     skewed the degree distribution is (R-MAT hubs, empty rows).  Tile start rows are cached with
//     the matrix (they only depend on the row pointers).
//   * All HBM loads of a tile are issued up front: each thread's IPT consecutive entries arrive by 16-byte
//     buffer loads in the registers of the thread that consumes them; LDS holds only per-row state.
//   * Each non-empty row marks the entry where it starts (LDS u16 array); a wavefront max-scan by __shfl_up
//     gives every thread the row of its first entry -- no per-thread merge walk.
//   * x gathers are buffer loads over ONE image [hot-column table | u]; "no column" (-1: masked-out row, tile
//     tail) is out of range and reads nothing.  IPT independent gathers are in flight per lane.
//   * Straight-line segmented fold; a finished segment is one native LDS atomic into its row's accumulator;
//     rows cut by tile boundaries leave per-tile carries that k_mxv_seams folds (one wavefront per seam).
//   * Epilogue: each wavefront takes 64 consecutive output rows, applies mask / accum / replace
//     against the old w, writes values coalesced and the presence word with one __ballot.
//   * Few entries in u: push direction (SpMSpV, k_push) over the rows selected by u.
#include <algorithm>
#include <cstring>
#include <vector>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

namespace grb {

// The kernels live in grb_mxv_*.inc (one translation unit, so that the templated launchers below instantiate them):
#include "grb_mxv_common.inc"
#include "grb_mxv_pull.inc"
#include "grb_mxv_long.inc"
#include "grb_mxv_vdict.inc"
#include "grb_mxv_strip.inc"
#include "grb_mxv_rows.inc"
#include "grb_mxv_rows_tag.inc"
#include "grb_mxv_ctile.inc"
#include "grb_mxv_rtile.inc"
#include "grb_mxv_split_build.inc"
#include "grb_mxv_write.inc"
#include "grb_mxv_push.inc"
#include "grb_mxv_hot.inc"
#include "grb_mxv_order.inc"

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// entries of the hot-column table of a matrix with n columns whose operands have value_bytes per element (0: no table)
static int64_t hot_table_size(int64_t n, size_t value_bytes)
{
    int64_t k = ctx().hot_k > 0 ? ctx().hot_k : (int64_t)((2u << 20) / (value_bytes ? value_bytes : 1));
    // (the class strips keep the hottest codes of every class in LDS: the table has to hold at least those)
    if (ctx().hot_k <= 0 && value_bytes > 1 && ctx().long_kernel >= 2) {
        const int ncls = (ctx().long_classes == 16 || ctx().long_classes == 32 || ctx().long_classes == 64) ? ctx().long_classes : 8;
        k = std::max<int64_t>(k, long_lds_codes((int)value_bytes, false, LONG_LDS_WORDS) / 8 * ncls);
    }
    k = std::min<int64_t>(k, n / 4);
    k &= ~(int64_t)63;
    return k < 64 ? 0 : k;
}

static void ensure_hot(GB_Matrix_opaque *A, size_t value_bytes)
{
    if (A->hot_state != 0) return;
    A->hot_state = -1;
    const int64_t n = (int64_t)A->ncols, nnz = A->nvals;
    if (n < ctx().hot_min_cols || nnz == 0 || n + (int64_t)(1 << 22) > 0x7fffffff) return;
    const int64_t k = hot_table_size(n, value_bytes);
    if (k < 64) return;
    DevBuf<unsigned int> cnt(n, true);
    const int hstride = nnz >= ((int64_t)1 << 26) ? 8 : 1;
    hipLaunchKernelGGL(k_hot_hist, dim3((unsigned)ceil_div(nnz, 256 * hstride)), dim3(256), 0, ctx().stream, A->d_col, nnz, cnt.p, hstride);
    DevBuf<uint64_t> keys(n), keys2(n);
    DevBuf<uint32_t> ids(n), ids2(n);
    hipLaunchKernelGGL(k_hot_keys, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, cnt.p, n, keys.p, ids.p);
    prim_sort_pairs_u64_u32(keys.p, keys2.p, ids.p, ids2.p, n, 32);
    DevBuf<int32_t> rank(n);
    GRB_HIP(hipMemsetAsync(rank.p, 0xff, sizeof(int32_t) * (size_t)n, ctx().stream));
    int32_t *hot_cols = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)k);
    DevBuf<unsigned long long> covered(1, true);
    hipLaunchKernelGGL(k_hot_rank, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, ctx().stream, ids2.p, keys2.p, k, rank.p,
                       hot_cols, covered.p);
    unsigned long long cov = 0;
    d2h(&cov, covered.p, sizeof(cov));
    if ((double)cov * hstride < 0.25 * (double)nnz) {  // flat degree distribution: the table would not pay for itself
        dev_free(hot_cols);
        return;
    }
    A->d_col_hot = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnz);
    hipLaunchKernelGGL(k_hot_recode, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, ctx().stream, A->d_col, nnz, rank.p,
                       (int)k, A->d_col_hot);
    A->d_hot_cols = hot_cols;
    A->hot_k = k;
    A->hot_state = 1;
}

// The re-coded copy of the whole column array is released once the split is built from it (ensure_split); a later rebuild of the
// split in the hot coding (another long-row kernel or class count was selected) needs it again: re-code from the hot-column list.
static void restore_hot_cols(GB_Matrix_opaque *A)
{
    if (A->hot_identity) fail(GrB_PANIC, "pull SpMV: an ordered layout cannot be rebuilt in place (internal error)");
    if (A->d_col_hot || A->hot_state != 1 || !A->d_hot_cols) return;
    const int64_t n = (int64_t)A->ncols, nnz = A->nvals;
    DevBuf<int32_t> rank(n);
    GRB_HIP(hipMemsetAsync(rank.p, 0xff, sizeof(int32_t) * (size_t)n, ctx().stream));
    hipLaunchKernelGGL(k_hot_rank_scatter, dim3((unsigned)ceil_div(A->hot_k, 256)), dim3(256), 0, ctx().stream, (const int32_t *)A->d_hot_cols,
                       A->hot_k, rank.p);
    A->d_col_hot = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    hipLaunchKernelGGL(k_hot_recode, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, ctx().stream, A->d_col, nnz, (const int32_t *)rank.p,
                       (int)A->hot_k, A->d_col_hot);
    sync_stream();  // (rank is released at the end of this scope)
    A->hot_cols_dropped = false;
}

template <typename T> struct PullIPT { static constexpr int value = sizeof(T) >= 8 ? 4 : 8; };

static void ensure_tile_table(GB_Matrix_opaque *A, int tile_items)
{
    if (A->d_tile_row && A->tile_items == tile_items) return;
    dev_free(A->d_tile_row);
    A->d_tile_row = nullptr;
    const int64_t total = (int64_t)A->nrows + A->nvals;
    const int64_t n_tiles = ceil_div(total, tile_items);
    int64_t *tab = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(n_tiles + 1));
    const int64_t nt1 = n_tiles + 1;
    hipLaunchKernelGGL(k_tile_table, dim3((unsigned)ceil_div(nt1, 256)), dim3(256), 0, ctx().stream, matrix_rowptr(A),
                       (int64_t)A->nrows, A->nvals, tile_items, n_tiles, tab);
    A->d_tile_row = tab;
    A->n_tiles = n_tiles;
    A->tile_items = tile_items;
}

// diagnostics (GRB_DEBUG_FLAGS & 8): mean cycles between the phase stamps of k_mxv_pull
static void report_phase_times(const long long *d_times, int64_t n_tiles)
{
    std::vector<long long> h((size_t)n_tiles * 10);
    d2h(h.data(), d_times, sizeof(long long) * h.size());
    double sum[9] = {0};
    long long tmin = h[0], tmax = h[0];
    for (int64_t t = 0; t < n_tiles; t++) {
        for (int i = 0; i < 7; i++)
            if (h[t * 10 + i + 1] && h[t * 10 + i]) sum[i] += (double)(h[t * 10 + i + 1] - h[t * 10 + i]);
        tmin = std::min(tmin, h[t * 10]);
        tmax = std::max(tmax, h[t * 10 + 7]);
    }
    fprintf(stderr, "[k_mxv_pull phases, mean clock64 ticks/tile over %lld tiles] LDS fill->sync %.0f | mask+heads+issue next->sync %.0f | "
            "head scan->sync %.0f | classify+gather %.0f | fold->sync %.0f | epilogue %.0f | seams %.0f | kernel span %lld\n",
            (long long)n_tiles, sum[0] / n_tiles, sum[1] / n_tiles, sum[2] / n_tiles, sum[3] / n_tiles, sum[4] / n_tiles,
            sum[5] / n_tiles, sum[6] / n_tiles, tmax - tmin);
}

// value dictionary of a matrix (grb_mxv_vdict.inc): at most 256 distinct finite values of a 4-byte type -> one-byte codes in the lane
// records of the hot strips, the cold tiles and the tagged row groups.  Sets A->vdict_n (0: none).
static void ensure_vdict(GB_Matrix_opaque *A, bool wanted)
{
    dev_free(A->d_vdict); dev_free(A->d_vd_table); dev_free(A->d_vd_codes);
    A->d_vdict = nullptr; A->d_vd_table = nullptr; A->d_vd_codes = nullptr;
    A->vdict_n = 0;
    A->vals_finite = false;
    const int64_t nnz = A->nvals;
    if (!wanted || !ctx().value_dict || A->iso || A->type->size != 4 || A->type->code == TC_BOOL || nnz == 0) return;
    DevBuf<unsigned long long> vd_table(VDICT_SLOTS, true);
    DevBuf<unsigned int> vd_count(1, true);
    hipLaunchKernelGGL(k_vdict_collect, dim3((unsigned)std::min<int64_t>(ceil_div(nnz, 256), (int64_t)ctx().num_cus * 16)), dim3(256), 0, ctx().stream,
                       (const uint32_t *)A->d_val, nnz, vd_table.p, vd_count.p);
    unsigned int h_count = 0;
    d2h(&h_count, vd_count.p, sizeof(h_count));
    if (h_count < 1 || h_count > 256) return;
    std::vector<unsigned long long> h_table(VDICT_SLOTS);
    d2h(h_table.data(), vd_table.p, sizeof(unsigned long long) * VDICT_SLOTS);
    std::vector<uint32_t> dict(256, 0u);
    std::vector<unsigned char> h_codes(VDICT_SLOTS, 0);
    int next = 0;
    for (int sl = 0; sl < VDICT_SLOTS; sl++) {
        if (!h_table[(size_t)sl]) continue;
        const uint32_t bits = (uint32_t)h_table[(size_t)sl];
        if (A->type->code == TC_FP32 && ((bits >> 23) & 0xffu) == 0xffu) return;  // (inf / NaN: the padding trick of the fast kernel needs finite values)
        if (next < 256) {
            dict[(size_t)next] = bits;
            h_codes[(size_t)sl] = (unsigned char)next;
        }
        next++;
    }
    if (next != (int)h_count) return;
    A->d_vdict = dev_alloc(256 * sizeof(uint32_t));
    h2d(A->d_vdict, dict.data(), 256 * sizeof(uint32_t));
    A->d_vd_table = (unsigned long long *)dev_alloc(sizeof(unsigned long long) * VDICT_SLOTS);
    h2d(A->d_vd_table, h_table.data(), sizeof(unsigned long long) * VDICT_SLOTS);
    A->d_vd_codes = (unsigned char *)dev_alloc(VDICT_SLOTS);
    h2d(A->d_vd_codes, h_codes.data(), VDICT_SLOTS);
    A->vdict_n = next;
    A->vals_finite = true;  // (every distinct value was looked at above)
    A->vals_absmax = 0.0;
    if (A->type->code == TC_FP32)
        for (int i = 0; i < next; i++) {
            float f;
            memcpy(&f, &dict[(size_t)i], 4);
            A->vals_absmax = std::max(A->vals_absmax, (double)(f < 0 ? -f : f));
        }
}

// Analyse (once) whether the rows split usefully into long and short ones and build the two parts.
// `col_src` is the column array the kernels will index (hot-coded or original).
static void ensure_tagged_index(GB_Matrix_opaque *A);
static void ensure_tagged(GB_Matrix_opaque *A);
static void ensure_rtile(GB_Matrix_opaque *A);
// the short-row kernel of a matrix: option 6 (default) = tagged row groups for large matrices, the row-group kernel below
static int short_kernel_for(const GB_Matrix_opaque *A)
{
    const int sk = ctx().short_kernel;
    return sk == 6 ? ((A->hot_identity || A->nvals >= ctx().lean_min_nnz) ? 5 : 1) : sk;  // (an ordered twin: always the lean layouts)
}
static void ensure_split(GB_Matrix_opaque *A, const int32_t *col_src, bool hot)
{
    const int ncls_opt = (ctx().long_classes == 16 || ctx().long_classes == 32 || ctx().long_classes == 64) ? ctx().long_classes : 8;
    // which long-row kernel this matrix is laid out for: option 3 (default) = class strips, except for BOOL matrices -- their
    // products are terminal monoids (the BFS step), where the item kernel's per-row early exit wins
    // (5: the same by type with the hot / cold strips -- 4 -- for non-BOOL matrices)
    const int lk = ctx().long_kernel;
    const bool big = A->hot_identity || A->nvals >= ctx().lean_min_nnz;  // (the hot / cold strips and the tagged row groups pay from ~50 M entries in natural order; an ordered twin always takes them)
    const int kind = lk == 3 ? (A->type->code == TC_BOOL ? 1 : 2) : (lk == 5 ? (A->type->code == TC_BOOL ? 1 : (big ? 4 : 2)) : lk);
    // (the short part holds its entries either as CSR arrays or, for the tagged row groups, in that layout alone: another short-row
    //  kernel than the one the split was built for rebuilds it)
    // (the tagged groups number their units with 31 bits and their entries with 33: beyond that the short part keeps its CSR arrays
    //  and the row-group kernel -- the same guard as where short_tagged_only is set, or the split would be rebuilt at every call)
    const bool want_tagged_only = short_kernel_for(A) == 5 && A->nvals < 0x1ffffffffll;
    if (A->split_state != 0 && (A->split_state < 0 || (A->split_hot == hot && A->split_kind == kind && A->short_tagged_only == want_tagged_only &&
                                                       ((A->split_kind != 2 && A->split_kind != 4) || (A->strip_nseg == 0 && A->ct_units == 0) || A->strip_ncls == ncls_opt)))) return;
    if (A->split_state == 1) {  // built against the other column coding (or for another long-row kernel): rebuild
        matrix_free(A->short_part);
        A->short_part = nullptr;
        dev_free(A->d_long_bits); dev_free(A->d_long_rows); dev_free(A->d_chunk_slot); dev_free(A->d_chunk_start); dev_free(A->d_chunk_len); dev_free(A->d_long_prefix);
        dev_free(A->d_probe); A->d_probe = nullptr; A->probe_k = 0;
        dev_free(A->d_lcol); dev_free(A->d_lval); dev_free(A->d_it_start); dev_free(A->d_it_len); dev_free(A->d_it_slot);
        dev_free(A->d_item_begin);
        // (layouts derived from the short part go with it)
        dev_free(A->d_tg_off); dev_free(A->d_tg_col); dev_free(A->d_tg_val); dev_free(A->d_tg_tag); dev_free(A->d_tg_nonempty);
        A->d_tg_off = nullptr; A->d_tg_col = nullptr; A->d_tg_val = nullptr; A->d_tg_tag = nullptr; A->d_tg_nonempty = nullptr; A->tg_state = 0;
        dev_free(A->d_rt_col); dev_free(A->d_rt_tag); dev_free(A->d_rt_val); dev_free(A->d_rt_tiles); dev_free(A->d_rt_order); dev_free(A->d_rt_counter);
        A->d_rt_col = nullptr; A->d_rt_tag = nullptr; A->d_rt_val = nullptr; A->d_rt_tiles = nullptr; A->d_rt_order = nullptr; A->d_rt_counter = nullptr;
        A->rt_state = 0; A->rt_units = 0; A->rt_ntiles = 0;
        A->tails_max_len = 0;
        dev_free(A->d_sstart); dev_free(A->d_sslot); dev_free(A->d_hrec); dev_free(A->d_sslot16); dev_free(A->d_sslot_base);
        A->d_sstart = nullptr; A->d_sslot = nullptr; A->d_hrec = nullptr; A->strip_nseg = 0; A->hub_ncls = 0;
        A->d_sslot16 = nullptr; A->d_sslot_base = nullptr;
        dev_free(A->d_ct_col); dev_free(A->d_ct_val); dev_free(A->d_ct_loc); dev_free(A->d_ct_tiles); dev_free(A->d_ct_order);
        A->d_ct_col = nullptr; A->d_ct_val = nullptr; A->d_ct_loc = nullptr; A->d_ct_tiles = nullptr; A->d_ct_order = nullptr; A->ct_units = 0; A->ct_ntiles = 0;
        A->d_lcol = nullptr; A->d_lval = nullptr; A->d_it_start = nullptr; A->d_it_len = nullptr; A->d_it_slot = nullptr;
        A->d_item_begin = nullptr; A->long_nnz = 0; A->n_items = 0;
        A->d_long_prefix = nullptr;
        A->d_long_bits = nullptr; A->d_long_rows = nullptr; A->d_chunk_slot = nullptr; A->d_chunk_start = nullptr; A->d_chunk_len = nullptr;
    }
    A->split_state = -1;
    if (hot && A->hot_cols_dropped) {  // (a rebuild in the hot coding after the re-coded columns were released)
        restore_hot_cols(A);
        col_src = A->d_col_hot;
    }
    if (!col_src) fail(GrB_PANIC, "pull SpMV: the split cannot be built without a column source (internal error)");
    const int64_t m = (int64_t)A->nrows, nnz = A->nvals;
    if (nnz < ctx().split_min_nnz || m == 0 || (ctx().debug_flags & 128)) return;
    const int min_len = ctx().split_min_len > 0 ? ctx().split_min_len : ((kind == 2 || kind == 4) ? 64 : 256);  // (measured optima, scripts/gpu_r02_nc.sh)
    // Round 6: the COLD entries of the long rows below `tail_max` entries -- column codes that are not LDS-resident in their class: codes from
    // `tail_lim` on -- join the short part (Context::cold_in_rows): on an ordered twin the short rows' sorted row tiles stream them at 7 bytes
    // an entry into accumulators they hold in LDS anyway, where the cold tiles (10 bytes, a flush by global atomics, a launch of their own)
    // cost twice as much per entry.  Hub rows keep their cold entries as cold tiles (64 of them would fill a row tile a hundred times).
    // (the limits below are the ones the class keys use further down: strip_classes / lds_lim4 / hub_ncls -- checked there)
    int64_t tail_max = 0;
    unsigned tail_lim = 0u;
    if (kind == 4 && A->hot_identity && hot && want_tagged_only && ctx().rows_tile && ctx().cold_in_rows > min_len && nnz < 0xf0000000ll &&
        A->d_cold_bounds && A->ct_ncr > 0 && !A->iso && (A->type->size == 4 || A->type->size == 8)) {
        const int ncls_t = (ctx().long_classes == 16 || ctx().long_classes == 32 || ctx().long_classes == 64) ? ctx().long_classes : 8;
        const int64_t lim_t = std::min<int64_t>(A->hot_k, long_lds_codes((int)A->type->size, false, LONG_LDS_WORDS) / 8 * ncls_t);
        const bool hub_t = ctx().hub_min_len > 0 && ncls_t < 64;
        tail_max = hub_t ? std::min<int64_t>(ctx().cold_in_rows, ctx().hub_min_len) : (int64_t)ctx().cold_in_rows;
        tail_lim = (unsigned)lim_t;
        if (tail_max <= min_len) tail_max = 0;
    }
    DevBuf<uint64_t> lbits(bits_words64((uint64_t)m));
    DevBuf<int64_t> slen(m + 1), lflag(m + 1), nchunk(m + 1);
    DevBuf<unsigned long long> tails_total(1, true);
    hipLaunchKernelGGL(k_split_classify, dim3((unsigned)ceil_div((int64_t)bits_words64((uint64_t)m) * 64 + 1, 256)), dim3(256), 0,
                       ctx().stream, (const int64_t *)A->d_ptr, m, min_len, lbits.p, slen.p, lflag.p, nchunk.p, col_src, tail_max, tail_lim, tails_total.p);
    prim_exclusive_sum_i64(slen.p, slen.p, m + 1);
    prim_exclusive_sum_i64(lflag.p, lflag.p, m + 1);
    prim_exclusive_sum_i64(nchunk.p, nchunk.p, m + 1);
    int64_t nnz_short = 0, nl = 0, nc = 0;
    d2h(&nnz_short, slen.p + m, 8);
    d2h(&nl, lflag.p + m, 8);
    d2h(&nc, nchunk.p + m, 8);
    unsigned long long n_tails = 0;
    if (tail_max > 0) d2h(&n_tails, tails_total.p, 8);
    if (nl == 0 || (double)(nnz - nnz_short + (int64_t)n_tails) < 0.3 * (double)nnz) return;  // too few entries in long rows to pay off
    ensure_vdict(A, kind == 4);
    GB_Matrix_opaque *S = matrix_new(A->type, A->nrows, A->ncols);
    try {
        S->d_ptr = slen.release();
        S->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(nnz_short ? nnz_short : 1));
        S->d_val = dev_alloc(A->type->size * (size_t)(A->iso ? 1 : (nnz_short ? nnz_short : 1)));
        if (A->iso) d2d(S->d_val, A->d_val, A->type->size);
        S->iso = A->iso;
        S->nvals = nnz_short;
        A->d_long_rows = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nl);
        A->d_chunk_slot = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nc);
        A->d_chunk_start = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)nc);
        A->d_chunk_len = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nc);
        A->d_long_prefix = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)bits_words64((uint64_t)m));
        GRB_DISPATCH_TYPE(A->type->code, T, {
            hipLaunchKernelGGL((k_split_fill<T>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream,
                               (const int64_t *)A->d_ptr, col_src, (const T *)A->d_val, A->iso ? 1 : 0, m, min_len,
                               (const int64_t *)S->d_ptr, (const int64_t *)lflag.p, (const int64_t *)nchunk.p, S->d_col,
                               (T *)S->d_val, A->d_long_rows, A->d_chunk_slot, A->d_chunk_start, A->d_chunk_len, A->d_long_prefix);
            if (tail_max > 0)
                hipLaunchKernelGGL((k_split_fill_tails<T>), dim3((unsigned)ceil_div(nl, 4)), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr, col_src,
                                   (const T *)A->d_val, A->iso ? 1 : 0, (const int32_t *)A->d_long_rows, nl, (const int64_t *)S->d_ptr, S->d_col, (T *)S->d_val,
                                   tail_max, tail_lim);
        })
        // the bottom-up probe of a BOOL matrix (k_long_init): the first 16 column codes of every long row
        dev_free(A->d_probe);
        A->d_probe = nullptr;
        A->probe_k = 0;
        if (A->type->code == TC_BOOL && kind == 1 && nl * 16 < 0x7fffffffll) {
            A->d_probe = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(nl * 16));
            hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)ceil_div(nl * 16, 256)), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr, col_src,
                               (const int32_t *)A->d_long_rows, nl, 16, A->d_probe);
            A->probe_k = 16;
        }
        // class-partitioned copy of the long rows: sort the entries by (class, slot), cut the runs into items, sort the
        // items by (class, falling length), lay them out with 4-entry aligned starts
        // (tail_max > 0: the keys cover ALL entries of the long rows -- those that moved to the short part get the discard class and sort behind
        //  everything; nnz_long counts the entries the long-row kernels keep)
        const int64_t nnz_long = nnz - nnz_short;
        DevBuf<int64_t> lpre(tail_max > 0 ? nl + 1 : 0);
        int64_t nnz_keyed = nnz_long;
        if (tail_max > 0) {
            hipLaunchKernelGGL(k_long_lens, dim3((unsigned)ceil_div(nl + 1, 256)), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr, (const int32_t *)A->d_long_rows, nl, lpre.p);
            prim_exclusive_sum_i64(lpre.p, lpre.p, nl + 1);
            d2h(&nnz_keyed, lpre.p + nl, 8);
        }
        A->long_nnz = 0;
        A->n_items = 0;
        if (nnz_keyed < 0xf0000000ll && nnz < 0xffffffffll) {
            // sub-ranges per class: sized so that one sub-range of the operand image is ~2 MiB (half an XCD's L2; BOOL images are
            // bit-packed and fit as they are); measured on R-MAT scale 24 fp32: 4 sub-ranges -3 % per call, and rows need ~512
            // entries per sub-range or their items get too small (sub 8 from 2048 entries: +3 %)
            const bool strips = kind == 2 || kind == 4;
            const int ncls = (strips && (ctx().long_classes == 16 || ctx().long_classes == 32 || ctx().long_classes == 64))
                                 ? ctx().long_classes : 8;
            // hot / cold strips: the codes that are LDS-resident in their class, and the contiguous column ranges of the others
            const int COLD_CLS = 8;
            const int64_t lds_lim4 = std::min<int64_t>(hot ? A->hot_k : 0, long_lds_codes((int)A->type->size, A->type->code == TC_BOOL, LONG_LDS_WORDS) / 8 * ncls);
            const int64_t codes_total = (int64_t)A->ncols + ((hot && !A->hot_identity) ? A->hot_k : 0);
            // (kind 4: the cold entries go to tagged tiles of 8 * `sub` column ranges of ~3 MiB of the operand image each, every row cut)
            const int range_cls = kind == 4 ? COLD_CLS : ncls;
            unsigned sub = 1;
            if (ctx().long_sub > 0) sub = (unsigned)std::min(kind == 4 ? 8 : 16, ctx().long_sub);
            else if (A->type->code != TC_BOOL || kind == 4) {
                const int64_t image_bytes = A->type->code == TC_BOOL ? (int64_t)A->ncols / 8 : (int64_t)A->ncols * (int64_t)A->type->size;
                while (sub < (kind == 4 ? 8u : 16u) && image_bytes > (int64_t)sub * range_cls * (3ll << 20)) sub *= 2;
            }
            const int64_t sub_min_len = kind == 4 ? 0 : (ctx().long_sub_min_len > 0 ? ctx().long_sub_min_len : 512 * (int64_t)sub);
            const int64_t nv = (int64_t)ncls * (int64_t)sub * nl;
            int bits = 1;
            while (((int64_t)1 << bits) < nv) bits++;
            DevBuf<uint64_t> keys(nnz_keyed), keys2(nnz_keyed);
            DevBuf<uint32_t> idx(nnz_keyed), idx2(nnz_keyed);
            // (a matrix in its popularity order carries its own column ranges: about equal reference counts, at most ~2 MiB of operand)
            const bool own_ranges = kind == 4 && A->hot_identity && A->d_cold_bounds && A->ct_ncr > 0;
            // ... and deals its hub rows to 64 classes (GB_Matrix_opaque::hub_ncls): only there are the LDS heads lines of the operand
            // itself -- with a per-call hot table four times the codes would cost four times the image
            const int hub_ncls = (own_ranges && ctx().hub_min_len > 0 && ncls < 64) ? 64 : 0;
            const int64_t hub_lim = hub_ncls ? std::min<int64_t>((int64_t)A->ncols, long_lds_codes((int)A->type->size, A->type->code == TC_BOOL, LONG_LDS_WORDS) / 8 * hub_ncls) : 0;
            const int64_t hub_len = hub_ncls ? (int64_t)ctx().hub_min_len : 0;
            if (tail_max > 0 && (!own_ranges || (int64_t)tail_lim != lds_lim4 || (hub_ncls && tail_max > hub_len)))
                fail(GrB_PANIC, "long / short split: the cold entries of the long rows were cut at another limit than the class keys use (internal error)");
            const unsigned discard_cls = (unsigned)(ncls + hub_ncls + (own_ranges ? A->ct_ncr : COLD_CLS * (int)sub));  // (behind the last cold range)
            if (kind == 4)
                hipLaunchKernelGGL(k_long_keys, dim3((unsigned)nl), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr,
                                   (const int64_t *)S->d_ptr, (const int32_t *)A->d_long_rows, col_src, nl, keys.p, idx.p,
                                   (unsigned)lds_lim4, (unsigned)std::max<int64_t>(1, ceil_div(codes_total - std::max<int64_t>(lds_lim4, hot ? A->hot_k : 0), (int64_t)COLD_CLS * (int64_t)sub)),
                                   sub, sub_min_len, (unsigned)ncls, 2, (unsigned)(hot ? A->hot_k : 0),
                                   own_ranges ? (const int32_t *)A->d_cold_bounds : (const int32_t *)nullptr, own_ranges ? A->ct_ncr + 1 : 0,
                                   hub_len, (unsigned)hub_lim, (unsigned)hub_ncls, tail_max, discard_cls, tail_max > 0 ? (const int64_t *)lpre.p : (const int64_t *)nullptr);
            else
            hipLaunchKernelGGL(k_long_keys, dim3((unsigned)nl), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr,
                               (const int64_t *)S->d_ptr, (const int32_t *)A->d_long_rows, col_src, nl, keys.p, idx.p,
                               (unsigned)(hot ? A->hot_k : 0), (unsigned)std::max<int64_t>(1, ceil_div((int64_t)A->ncols, (int64_t)ncls * (int64_t)sub)),
                               sub, sub_min_len, (unsigned)ncls, kind == 2 ? 1 : 0, 0u, (const int32_t *)nullptr, 0, (int64_t)0, 0u, 0u, (int64_t)0, 0u, (const int64_t *)nullptr);
            // (virtual classes: kind 2 ncls * sub; kind 4 ncls hot classes + COLD_CLS cold ranges, with `sub` = 1 in the segment tables)
            const int nvc = ncls + hub_ncls;                                            // classes of the strips (chunk ranges)
            const int n_cr = own_ranges ? A->ct_ncr : COLD_CLS * (int)sub;             // kind 4: column ranges of the cold tiles
            const int nvirt = kind == 4 ? nvc + n_cr + (tail_max > 0 ? 1 : 0) : ncls * (int)sub;  // virtual classes (sort keys; + the discard class)
            const int hot_cls = 0;
            const unsigned strip_sub = kind == 4 ? 1u : sub;                            // virtual classes per strip class
            if (strips) {
                int vbits = 1;
                while ((1 << vbits) < nvirt) vbits++;
                prim_sort_pairs_u64_u32_bits(keys.p, keys2.p, idx.p, idx2.p, nnz_keyed, 32, 32 + vbits);
            } else {
                prim_sort_pairs_u64_u32(keys.p, keys2.p, idx.p, idx2.p, nnz_keyed, bits);
            }
            // kind 4: the sorted entries are [hot: virtual classes 0 .. ncls - 1 | cold: column ranges ncls .. ncls + n_cr - 1]; the hot
            // part becomes strips (lane records), the cold part tagged tiles (grb_mxv_ctile.inc)
            // Round 6 (Context::ctile_pack): the cold tiles of an ordered matrix whose column ranges are at most 2^19 codes wide keep column (offset in the
            // range) and row slot of an entry in one word -- slot << 19 | column: tiles of 8192 slots --, mode 2 also one-byte value codes
            int ct_mode = 0;
            std::vector<int32_t> h_bounds;
            if (kind == 4 && own_ranges && ctx().ctile_pack > 0) {
                h_bounds.resize((size_t)A->ct_ncr + 1);
                d2h(h_bounds.data(), A->d_cold_bounds, sizeof(int32_t) * h_bounds.size());
                int64_t widest = 0;
                for (int r = 0; r < A->ct_ncr; r++) widest = std::max<int64_t>(widest, (int64_t)h_bounds[(size_t)r + 1] - (int64_t)h_bounds[(size_t)r]);
                if (widest < ((int64_t)1 << 19) - 1) ct_mode = (ctx().ctile_pack == 2 && A->vdict_n > 0 && A->type->size == 4 && !A->iso) ? 2 : 1;
            }
            const int ct_slots = ct_mode ? CT_PACK_SLOTS : (A->type->size > 4 ? CT_SLOTS / 2 : CT_SLOTS);  // (CtSlots<W>: 8-byte accumulators take half the slots)
            const int n_sb = (int)ceil_div(nl, (int64_t)ct_slots);
            const int64_t n_tiles = kind == 4 ? (int64_t)n_cr * n_sb : 0;
            std::vector<int64_t> h_first((size_t)n_tiles + 1, 0);
            DevBuf<int64_t> tile_first(n_tiles + 1);
            int64_t n_strip = nnz_long;
            if (kind == 4) {
                hipLaunchKernelGGL(k_ctile_first, dim3((unsigned)ceil_div(n_tiles + 1, 256)), dim3(256), 0, ctx().stream, (const uint64_t *)keys2.p, nnz_keyed,
                                   n_tiles, n_sb, (unsigned)nvc, tile_first.p, ct_slots);
                d2h(h_first.data(), tile_first.p, sizeof(int64_t) * (size_t)(n_tiles + 1));
                n_strip = h_first[0];
                // (h_first[n_tiles] = the first key of the discard class: everything in front of it is what the long-row kernels keep)
                if (h_first[(size_t)n_tiles] != nnz_long)
                    fail(GrB_PANIC, "long / short split: the short part and the class keys disagree about the cold entries that moved (internal error)");
            }
            if (kind == 4) {
                A->long_nnz = nnz_long;
                A->strip_ncls = ncls;
                A->hub_ncls = hub_ncls;
                A->hub_lds_lim = (int)hub_lim;
                A->strip_nseg = 0;
                A->cls_lds_lim = (int)lds_lim4;
                for (int c = 0; c <= nvc; c++) A->strip_cb[c] = 0;
            }
            if (strips && n_strip > 0) {
                // flat class strips (grb_mxv_strip.inc): segments = runs of equal keys; every segment padded to a multiple of 8
                // entries, every class to whole chunks of 512.  Kind 4: the hot entries only, as lane records in d_hrec
                const int64_t nblk = ceil_div(n_strip, STRIP_CH);
                DevBuf<int64_t> blk(nblk + 1);
                hipLaunchKernelGGL(k_strip_count, dim3((unsigned)nblk), dim3(STRIP_CH), 0, ctx().stream, (const uint64_t *)keys2.p, n_strip, blk.p);
                prim_exclusive_sum_i64(blk.p, blk.p, nblk + 1);
                int64_t nseg = 0;
                d2h(&nseg, blk.p + nblk, 8);
                if (nseg >= 0x7ffffff0ll) fail(GrB_NOT_IMPLEMENTED, "class strips: too many segments for 32-bit numbering");
                DevBuf<int64_t> seg_first(nseg + 1), off(nseg + 1);
                hipLaunchKernelGGL(k_strip_seg_fill, dim3((unsigned)nblk), dim3(STRIP_CH), 0, ctx().stream, (const uint64_t *)keys2.p, n_strip,
                                   (const int64_t *)blk.p, seg_first.p);
                hipLaunchKernelGGL(k_strip_plen, dim3((unsigned)ceil_div(nseg + 1, 256)), dim3(256), 0, ctx().stream, (const int64_t *)seg_first.p,
                                   nseg, off.p);
                prim_exclusive_sum_i64(off.p, off.p, nseg + 1);
                constexpr int MAXC = 160;
                DevBuf<int64_t> sc(MAXC + 1), raw(MAXC + 1), cshift(MAXC);
                hipLaunchKernelGGL(k_strip_class_bounds, dim3(1), dim3(128), 0, ctx().stream, (const uint64_t *)keys2.p, (const int64_t *)seg_first.p,
                                   nseg, nl, strip_sub, (const int64_t *)off.p, sc.p, raw.p, nvc, hot_cls);
                int64_t h_raw[MAXC + 1], h_shift[MAXC], h_cb[MAXC + 1];
                d2h(h_raw, raw.p, sizeof(int64_t) * (size_t)(nvc + 1));
                int64_t base = 0;
                for (int c = 0; c < nvc; c++) {
                    h_cb[c] = base / STRIP_CH;
                    h_shift[c] = base - h_raw[c];
                    base += ceil_div(h_raw[c + 1] - h_raw[c], STRIP_CH) * STRIP_CH;
                }
                h_cb[nvc] = base / STRIP_CH;
                const int64_t padded = base, nch = padded / STRIP_CH;
                h2d(cshift.p, h_shift, sizeof(int64_t) * (size_t)nvc);
                A->cls_lds_lim = (int)lds_lim4;
                const int64_t hot_chunks = kind == 4 ? h_cb[nvc] : 0;    // chunks of the hot strips (kind 4: all of them)
                const int64_t flat_entries = kind == 4 ? 0 : padded;     // entries held by d_lcol / d_lval
                const int code_bytes = A->type->code == TC_BOOL ? 32 : 16;
                const bool use_dict = A->vdict_n > 0;
                const int val_bytes = A->iso ? 0 : (use_dict ? 8 : (int)std::max<size_t>(16, 8 * A->type->size));
                if (padded > 0 && padded < 0x7fffffff0ll) {
                    A->d_lcol = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(flat_entries, 1));
                    A->d_lval = A->iso ? nullptr : dev_alloc(A->type->size * (size_t)std::max<int64_t>(flat_entries, 1));
                    A->d_sstart = (unsigned long long *)dev_alloc(sizeof(unsigned long long) * (size_t)nch);
                    A->d_sslot = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(padded / 8));
                    GRB_HIP(hipMemsetAsync(A->d_sslot, 0xff, sizeof(int32_t) * (size_t)(padded / 8), ctx().stream));
                    GRB_HIP(hipMemsetAsync(A->d_lcol, 0xff, sizeof(int32_t) * (size_t)std::max<int64_t>(flat_entries, 1), ctx().stream));
                    if (A->d_lval) GRB_HIP(hipMemsetAsync(A->d_lval, 0, A->type->size * (size_t)std::max<int64_t>(flat_entries, 1), ctx().stream));
                    GRB_HIP(hipMemsetAsync(A->d_sstart, 0, sizeof(unsigned long long) * (size_t)nch, ctx().stream));
                    A->hrec_bytes = 0;
                    if (kind == 4) {
                        A->hrec_bytes = code_bytes + val_bytes;
                        const size_t hbytes = (size_t)std::max<int64_t>(hot_chunks, 1) * 64 * (size_t)A->hrec_bytes;
                        A->d_hrec = (char *)dev_alloc(hbytes);
                        if (A->type->code == TC_BOOL) GRB_HIP(hipMemsetAsync(A->d_hrec, 0xff, hbytes, ctx().stream));  // (code -1 = padding)
                        else {
                            const int64_t n_rec = std::max<int64_t>(hot_chunks, 1) * 64;
                            GRB_DISPATCH_TYPE(A->type->code, T, {
                                hipLaunchKernelGGL((k_hrec_init<T>), dim3((unsigned)ceil_div(n_rec, 256)), dim3(256), 0, ctx().stream, A->d_hrec, n_rec,
                                                   A->hrec_bytes, hot_pad_code<T>(LONG_LDS_WORDS));
                            })
                        }
                    }
                    GRB_DISPATCH_TYPE(A->type->code, T, {
                        hipLaunchKernelGGL((k_strip_place<T>), dim3((unsigned)nblk), dim3(STRIP_CH), 0, ctx().stream, (const uint64_t *)keys2.p,
                                           (const uint32_t *)idx2.p, n_strip, (const int64_t *)blk.p, (const int64_t *)seg_first.p,
                                           (const int64_t *)off.p, (const int64_t *)cshift.p, nl, strip_sub, col_src, (const T *)A->d_val,
                                           A->iso ? 1 : 0, A->cls_lds_lim, ncls, A->d_lcol, (T *)A->d_lval, A->d_sstart, A->d_sslot,
                                           A->d_hrec, A->hrec_bytes, hot_chunks * STRIP_CH, hot_cls,
                                           use_dict ? (const unsigned long long *)A->d_vd_table : (const unsigned long long *)nullptr,
                                           use_dict ? (const unsigned char *)A->d_vd_codes : (const unsigned char *)nullptr, hub_ncls, (int)hub_lim);
                    })
                    {
                        int64_t h_end[MAXC];
                        for (int c = 0; c < nvc; c++) h_end[c] = h_cb[c + 1] * STRIP_CH;
                        DevBuf<int64_t> cend(MAXC);
                        h2d(cend.p, h_end, sizeof(int64_t) * (size_t)nvc);
                        hipLaunchKernelGGL(k_strip_pad_starts, dim3(1), dim3(128), 0, ctx().stream, (const int64_t *)sc.p, (const int64_t *)off.p,
                                           (const int64_t *)cshift.p, (const int64_t *)cend.p, nvc, A->d_sstart);
                        sync_stream();
                    }
                    // round 6: the slots as 16-bit offsets from each chunk's smallest slot (half the slot stream of the strip kernels)
                    dev_free(A->d_sslot16); dev_free(A->d_sslot_base);
                    A->d_sslot16 = nullptr; A->d_sslot_base = nullptr;
                    if (ctx().strip_slot16 && nch > 0) {
                        DevBuf<unsigned int> too_wide(1, true);
                        uint16_t *s16 = (uint16_t *)dev_alloc(sizeof(uint16_t) * (size_t)(padded / 8));
                        int32_t *sb = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nch);
                        hipLaunchKernelGGL(k_strip_slot16, dim3((unsigned)ceil_div(nch, 4)), dim3(256), 0, ctx().stream, (const int32_t *)A->d_sslot, nch, s16, sb, too_wide.p);
                        unsigned int wide = 0;
                        d2h(&wide, too_wide.p, sizeof(wide));
                        if (wide) {  // (a chunk whose rows lie 65535 slots apart: the 32-bit slots stay)
                            dev_free(s16);
                            dev_free(sb);
                        } else {
                            A->d_sslot16 = s16;
                            A->d_sslot_base = sb;
                            dev_free(A->d_sslot);
                            A->d_sslot = nullptr;
                        }
                    }
                    for (int c = 0; c <= nvc; c++) A->strip_cb[c] = h_cb[c];
                    dev_free(A->d_wg_tab); dev_free(A->d_strip_cb);
                    A->d_wg_tab = nullptr; A->d_strip_cb = nullptr; A->wg_tab_g = 0;
                    if (hub_ncls > 0) {
                        // one launch of the fast strip kernel over both levels: the persistent workgroups (one per CU) are dealt to the nvc
                        // classes by chunk count -- every class with chunks gets one, the rest go to whichever class has the most chunks
                        // per workgroup left
                        const int G = std::max(nvc, ctx().strip_wgs > 0 ? ctx().strip_wgs : ctx().num_cus);
                        std::vector<int> wgs((size_t)nvc, 0);
                        int used = 0;
                        for (int c = 0; c < nvc; c++)
                            if (h_cb[c + 1] > h_cb[c]) { wgs[(size_t)c] = 1; used++; }
                        while (used < G) {
                            int best = -1;
                            double best_load = -1.0;
                            for (int c = 0; c < nvc; c++) {
                                if (!wgs[(size_t)c]) continue;
                                const double load = (double)(h_cb[c + 1] - h_cb[c]) / wgs[(size_t)c];
                                if (load > best_load) { best_load = load; best = c; }
                            }
                            if (best < 0) break;
                            wgs[(size_t)best]++;
                            used++;
                        }
                        std::vector<int32_t> tab((size_t)G * 5, 0);
                        int b = 0;
                        // (workgroups of one class are spread over the launch order, not adjacent: consecutive ones land on different XCDs)
                        std::vector<int> next((size_t)nvc, 0);
                        for (int round = 0; b < G; round++) {
                            bool any = false;
                            for (int c = 0; c < nvc && b < G; c++) {
                                if (next[(size_t)c] >= wgs[(size_t)c]) continue;
                                any = true;
                                int32_t *t = &tab[(size_t)b * 5];
                                t[0] = c;
                                t[1] = c < ncls ? c : c - ncls;
                                t[2] = c < ncls ? ncls : hub_ncls;
                                t[3] = next[(size_t)c]++;
                                t[4] = wgs[(size_t)c];
                                b++;
                            }
                            if (!any) break;
                        }
                        for (; b < G; b++) {  // (no class left: an idle workgroup of class 0 behind its last chunk)
                            int32_t *t = &tab[(size_t)b * 5];
                            t[0] = 0; t[1] = 0; t[2] = ncls; t[3] = 1 << 28; t[4] = 1;
                        }
                        A->d_wg_tab = (int32_t *)dev_alloc(sizeof(int32_t) * tab.size());
                        h2d(A->d_wg_tab, tab.data(), sizeof(int32_t) * tab.size());
                        A->d_strip_cb = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(nvc + 1));
                        h2d(A->d_strip_cb, h_cb, sizeof(int64_t) * (size_t)(nvc + 1));
                        A->wg_tab_g = G;
                    }
                    if (getenv("GRB_PRINT_STRIPS")) {  // (diagnostic: chunks per class -- a persistent class with more chunks than the others sets the kernel's time)
                        fprintf(stderr, "[strips kind %d] %d classes (%d of them the hub level's), chunks per class:", kind, nvc, hub_ncls);
                        for (int c = 0; c < nvc; c++) fprintf(stderr, " %lld", (long long)(h_cb[c + 1] - h_cb[c]));
                        fprintf(stderr, "  segments %lld\n", (long long)nseg);
                    }
                    A->strip_ncls = ncls;
                    A->strip_nseg = nseg;
                    A->long_nnz = nnz_long;
                    sync_stream();  // (the temporaries above are released at the end of this scope)
                }
            }
            // (measured, profiles/r04/value_dict.txt: one-byte codes make the tagged row groups 5 % faster -- 9 -> 6 bytes per entry of a
            //  kernel that streams them all -- and the cold tiles 7 % SLOWER: their stream is a third of their time, the code load + LDS
            //  lookup per entry costs more than the bytes it saves.  The tiles keep full values.)
            const bool use_dict_all = ct_mode == 2;  // (round 6: behind the packed words the codes were measured again -- profiles/r06/ctile_pack.txt)
            A->ct_mode = 0;
            if (kind == 4 && nnz_long > n_strip) {
                // the cold entries as tagged tiles: a (column range, slot block) pair is one tile, or several when it holds more than
                // CT_MAX_ENTRIES entries (the hub rows: the sorted run is cut into equal pieces -- a piece still lies inside the
                // pair's 4096 slots); per tile its entries padded to a multiple of 4, in (column range, row) order
                std::vector<CTile> h_tiles;
                std::vector<int64_t> h_efirst;
                std::vector<int> tile_range;
                std::vector<int64_t> range_cnt((size_t)n_cr, 0);
                int64_t units = 0;
                for (int64_t t = 0; t < n_tiles; t++) {
                    const int64_t cnt = h_first[t + 1] - h_first[t];
                    if (cnt == 0) continue;
                    const int64_t pieces = ceil_div(cnt, CT_MAX_ENTRIES), per = ceil_div(cnt, pieces);
                    for (int64_t q = 0; q < pieces; q++) {
                        const int64_t e0 = h_first[t] + q * per, e1 = std::min<int64_t>(h_first[t] + (q + 1) * per, h_first[t + 1]);
                        CTile ct;
                        ct.u0 = units;
                        ct.n_units = (int32_t)ceil_div(e1 - e0, (int64_t)CT_EPL);
                        ct.base = (int32_t)((t % n_sb) * (int64_t)ct_slots);
                        units += ct.n_units;
                        h_tiles.push_back(ct);
                        h_efirst.push_back(e0);
                        tile_range.push_back((int)(t / n_sb));
                    }
                    range_cnt[(size_t)(t / n_sb)] += cnt;
                }
                const int64_t nt = (int64_t)h_tiles.size();
                h_efirst.push_back(nnz_long);
                // column ranges to XCDs: round-robin (the natural order's equal-width ranges), or -- ranges of a popularity order,
                // whose counts differ -- the largest remaining range to the XCD with the least work; an XCD walks its ranges one after
                // the other, so its 32 CUs gather from ONE range at a time and it stays in their L2
                std::vector<int> range_xcd((size_t)n_cr);
                if (own_ranges) {
                    std::vector<int> by_cnt((size_t)n_cr);
                    for (int r = 0; r < n_cr; r++) by_cnt[(size_t)r] = r;
                    std::stable_sort(by_cnt.begin(), by_cnt.end(), [&](int x, int y) { return range_cnt[(size_t)x] > range_cnt[(size_t)y]; });
                    int64_t load[8] = {0};
                    for (int r : by_cnt) {
                        int best = 0;
                        for (int x = 1; x < 8; x++)
                            if (load[x] < load[best]) best = x;
                        range_xcd[(size_t)r] = best;
                        load[best] += range_cnt[(size_t)r];
                    }
                } else {
                    for (int r = 0; r < n_cr; r++) range_xcd[(size_t)r] = r % 8;
                }
                std::vector<int32_t> h_order;
                h_order.reserve((size_t)nt);
                for (int x = 0; x < 8; x++) {
                    A->ct_xoff[x] = (int64_t)h_order.size();
                    for (int64_t t = 0; t < nt; t++)
                        if (range_xcd[(size_t)tile_range[(size_t)t]] == x) h_order.push_back((int32_t)t);
                }
                A->ct_xoff[8] = (int64_t)h_order.size();
                const size_t ents = (size_t)std::max<int64_t>(units, 1) * CT_EPL;
                // (packed: behind the nt tile records, the first column code of every tile's range -- int32[nt])
                A->d_ct_tiles = dev_alloc((sizeof(CTile) + sizeof(int32_t)) * (size_t)std::max<int64_t>(nt, 1));
                h2d(A->d_ct_tiles, h_tiles.data(), sizeof(CTile) * (size_t)nt);
                if (ct_mode) {
                    std::vector<int32_t> h_cbase((size_t)nt);
                    for (int64_t t = 0; t < nt; t++) h_cbase[(size_t)t] = h_bounds[(size_t)tile_range[(size_t)t]];
                    h2d((char *)A->d_ct_tiles + sizeof(CTile) * (size_t)std::max<int64_t>(nt, 1), h_cbase.data(), sizeof(int32_t) * (size_t)nt);
                }
                dev_free(A->d_ct_order);
                A->d_ct_order = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nt, 1));
                h2d(A->d_ct_order, h_order.data(), sizeof(int32_t) * (size_t)nt);
                DevBuf<int64_t> efirst(nt + 1);
                h2d(efirst.p, h_efirst.data(), sizeof(int64_t) * (size_t)(nt + 1));
                A->d_ct_col = (int32_t *)dev_alloc(sizeof(int32_t) * ents);
                const size_t ct_vb = use_dict_all ? 1 : A->type->size;  // (dictionary-coded matrices: one byte per value)
                A->d_ct_val = A->iso ? nullptr : dev_alloc(ct_vb * ents);
                A->d_ct_loc = ct_mode ? nullptr : (uint16_t *)dev_alloc(sizeof(uint16_t) * ents);
                GRB_HIP(hipMemsetAsync(A->d_ct_col, 0xff, sizeof(int32_t) * ents, ctx().stream));  // (padding: column -1 / the packed word 0xffffffff)
                if (A->d_ct_val) GRB_HIP(hipMemsetAsync(A->d_ct_val, 0, ct_vb * ents, ctx().stream));
                if (A->d_ct_loc) GRB_HIP(hipMemsetAsync(A->d_ct_loc, 0, sizeof(uint16_t) * ents, ctx().stream));
                const int64_t n_cold = nnz_long - n_strip;
                {
                    // inside a tile the entries go by column code: neighbouring lanes gather from the same lines (grb_mxv_ctile.inc)
                    DevBuf<uint64_t> key2(n_cold), key2s(n_cold);
                    DevBuf<uint32_t> pay(n_cold), pays(n_cold);
                    hipLaunchKernelGGL(k_ctile_keys2, dim3((unsigned)ceil_div(n_cold, 256)), dim3(256), 0, ctx().stream, (const uint32_t *)idx2.p, n_strip, n_cold,
                                       (const int64_t *)efirst.p, nt, col_src, key2.p, pay.p);
                    int tbits = 1;
                    while (((int64_t)1 << tbits) < nt) tbits++;
                    prim_sort_pairs_u64_u32(key2.p, key2s.p, pay.p, pays.p, n_cold, 32 + tbits);
                    GRB_DISPATCH_TYPE(A->type->code, T, {
                        hipLaunchKernelGGL((k_ctile_place<T>), dim3((unsigned)ceil_div(n_cold, 256)), dim3(256), 0, ctx().stream, (const uint64_t *)keys2.p,
                                           (const uint32_t *)idx2.p, n_strip, n_cold, (const uint64_t *)key2s.p, (const uint32_t *)pays.p, (const int64_t *)efirst.p,
                                           (const CTile *)A->d_ct_tiles, (const T *)A->d_val, A->iso ? 1 : 0, A->d_ct_col, (T *)A->d_ct_val, A->d_ct_loc,
                                           use_dict_all ? (const unsigned long long *)A->d_vd_table : (const unsigned long long *)nullptr,
                                           use_dict_all ? (const unsigned char *)A->d_vd_codes : (const unsigned char *)nullptr,
                                           ct_mode ? (const int32_t *)((const char *)A->d_ct_tiles + sizeof(CTile) * (size_t)std::max<int64_t>(nt, 1)) : (const int32_t *)nullptr);
                    })
                    sync_stream();
                }
                A->ct_mode = ct_mode;
                A->ct_nsb = n_sb;
                A->ct_ncr = n_cr;
                A->ct_ntiles = nt;
                A->ct_units = units;
                if (getenv("GRB_PRINT_STRIPS")) {
                    fprintf(stderr, "[cold tiles] mode %d (slots %d), %d column ranges, %lld tiles, entries per range:", ct_mode, ct_slots, n_cr, (long long)nt);
                    for (int r = 0; r < n_cr; r++) fprintf(stderr, " %lld(x%d)", (long long)range_cnt[(size_t)r], range_xcd[(size_t)r]);
                    fprintf(stderr, "\n");
                }
                sync_stream();
            }
            DevBuf<int64_t> vptr(strips ? 0 : nv + 1), icnt(strips ? 0 : nv + 1);
            if (!strips) {
            hipLaunchKernelGGL(k_long_vptr, dim3((unsigned)ceil_div(nv + 1, 256)), dim3(256), 0, ctx().stream,
                               (const uint64_t *)keys2.p, nnz_long, nv, vptr.p);
            hipLaunchKernelGGL(k_long_item_count, dim3((unsigned)ceil_div(nv + 1, 256)), dim3(256), 0, ctx().stream,
                               (const int64_t *)vptr.p, nv, icnt.p);
            prim_exclusive_sum_i64(icnt.p, icnt.p, nv + 1);
            int64_t ni = 0;
            d2h(&ni, icnt.p + nv, 8);
            if (ni > 0 && ni < 0x7fffffffll) {
                DevBuf<uint64_t> ikey(ni), ikey2(ni);
                DevBuf<uint32_t> iid(ni), iorder(ni);
                DevBuf<int64_t> isrc(ni), osrc(ni), len4(ni + 1);
                DevBuf<int32_t> ilen(ni), islot(ni);
                hipLaunchKernelGGL(k_long_item_fill, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0, ctx().stream,
                                   (const int64_t *)vptr.p, nv, nl, (const int64_t *)icnt.p, ikey.p, iid.p, isrc.p, ilen.p, islot.p);
                prim_sort_pairs_u64_u32(ikey.p, ikey2.p, iid.p, iorder.p, ni, 19);  // 11 bits of length under the virtual class (< 128)
                A->d_it_len = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)ni);
                A->d_it_slot = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)ni);
                A->d_it_start = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(ni + 1));
                hipLaunchKernelGGL(k_long_item_order, dim3((unsigned)ceil_div(ni + 1, 256)), dim3(256), 0, ctx().stream,
                                   (const uint32_t *)iorder.p, ni, (const int64_t *)isrc.p, (const int32_t *)ilen.p,
                                   (const int32_t *)islot.p, osrc.p, A->d_it_len, A->d_it_slot, len4.p);
                prim_exclusive_sum_i64(len4.p, A->d_it_start, ni + 1);
                int64_t padded = 0;
                d2h(&padded, A->d_it_start + ni, 8);
                A->d_item_begin = (int64_t *)dev_alloc(sizeof(int64_t) * 9);
                hipLaunchKernelGGL(k_long_class_bounds, dim3(1), dim3(64), 0, ctx().stream, (const uint64_t *)ikey2.p, ni, A->d_item_begin, sub);
                d2h(A->item_begin, A->d_item_begin, sizeof(int64_t) * 9);
                // (+2048 entries: a group's steps run to the longest item of its quad, i.e. past its own entries)
                // only codes of the hot table are classed by line (k_long_keys): those may live in LDS
                A->cls_lds_lim = (int)std::min<int64_t>(hot ? A->hot_k : 0,
                                                        long_lds_codes((int)A->type->size, A->type->code == TC_BOOL, LONG_LDS_WORDS));
                A->d_lcol = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(padded + 2048));
                A->d_lval = A->iso ? nullptr : dev_alloc(A->type->size * (size_t)(padded + 2048));
                GRB_DISPATCH_TYPE(A->type->code, T, {
                    hipLaunchKernelGGL((k_long_place<T>), dim3((unsigned)ni), dim3(64), 0, ctx().stream, (const int64_t *)A->d_it_start,
                                       (const int64_t *)osrc.p, (const int32_t *)A->d_it_len, (const uint32_t *)idx2.p, col_src,
                                       (const T *)A->d_val, A->iso ? 1 : 0, A->d_lcol, (T *)A->d_lval, A->cls_lds_lim);
                })
                GRB_HIP(hipMemsetAsync(A->d_lcol + padded, 0xff, sizeof(int32_t) * 2048, ctx().stream));
                if (A->d_lval) GRB_HIP(hipMemsetAsync((char *)A->d_lval + A->type->size * (size_t)padded, 0, A->type->size * 2048, ctx().stream));
                A->n_items = ni;
                A->long_nnz = nnz_long;
                sync_stream();  // (the temporaries above are released at the end of this scope)
            }
            }
        }
        sync_stream();
    } catch (...) {
        matrix_free(S);
        throw;
    }
    A->short_part = S;
    A->d_long_bits = lbits.release();
    A->n_long = nl;
    A->n_chunks = nc;
    A->split_hot = hot;
    A->split_kind = kind;
    A->split_state = 1;
    A->tails_max_len = tail_max;
    A->short_tagged_only = false;
    if (want_tagged_only && S->nrows == A->nrows && S->nvals < 0x1ffffffffll) {
        // the tagged row groups are the only form of the short rows' entries the kernels read: the CSR copy they were built from is
        // released (0.42 GB of the cached layouts at scale 24); the row pointers stay (row lengths, accounting)
        // (round 6, lazy_tagged: with row tiles the groups' ENTRIES wait for the first call that needs them -- ensure_tagged then builds them from the tiles)
        ensure_tagged_index(A);
        // (an ordered twin: its short rows as sorted row tiles too, from the same CSR copy.  rows_tile = 2 builds them for the natural-order
        //  layouts of a hot-coded matrix as well -- row blocks of a sharded run, order_mode 0; the kernels take them there too, tested, but
        //  MEASURED SLOWER than the tagged row groups: blocks 0/2, 0/4 of the scale-24 graph 0.335 -> 0.352, 0.197 -> 0.214 ms, the
        //  Kronecker-26 block 0.636 -> 0.652 (profiles/r05/final_run_summary_tiles_on_natural_layouts.txt) -- behind a 2 MiB hot table the
        //  codes of the other columns are original labels: sorting by them gathers nothing together)
        if (ctx().rows_tile && (A->hot_identity || (hot && ctx().rows_tile == 2))) ensure_rtile(A);
        if (A->rt_state != 1 || !ctx().lazy_tagged) ensure_tagged(A);
        dev_free(S->d_col);
        S->d_col = nullptr;
        if (!S->iso) {
            dev_free(S->d_val);
            S->d_val = nullptr;
        }
        A->short_tagged_only = true;
    }
    // the short part and the strips / items carry their own re-coded columns: the re-coded copy of the whole array is only read
    // again by a product that cannot take the split (a typecast of the values), which falls back to the plain arrays
    if (hot && ctx().drop_hot_cols && nc > 0 && A->long_nnz > 0 && (kind == 1 || kind == 2 || kind == 4) && A->d_col_hot) {
        dev_free(A->d_col_hot);
        A->d_col_hot = nullptr;
        if (A->hot_identity) A->d_col = nullptr;  // (the same array)
        A->hot_cols_dropped = true;
    }
}

// ---------------------------------------------------------------------------------------------------
// popularity-ordered layouts (grb_mxv_order.inc)
// ---------------------------------------------------------------------------------------------------
// the layout options an ordered twin depends on: a change rebuilds it from the matrix (its own CSR arrays are gone by then)
static uint64_t order_signature()
{
    const Context &c = ctx();
    uint64_t h = 1469598103934665603ull;
    const int64_t v[] = {c.long_kernel, c.short_kernel, c.long_classes, c.split_min_len, c.long_sub, c.long_sub_min_len, c.lean_min_nnz,
                         c.split_min_nnz, c.hot_k, c.drop_hot_cols, c.hub_min_len, c.value_dict, c.rows_tile, c.rtile_rows, c.rtile_entries, c.cold_in_rows, c.rtile_pack, c.strip_slot16, c.ctile_pack, c.lazy_tagged};
    for (int64_t x : v) h = (h ^ (uint64_t)x) * 1099511628211ull;
    return h;
}

// The vertex order of S's space: its own, or the one its transpose partner already has (both directions of a square matrix share
// one order, so a vector never has to change between A and A').  Built from S: vertices by falling column count (sampled like the
// hot table's histogram), ties by falling row length; the first HOT_MIX ranks are dealt over the lines of the hot table.
static GB_Perm *ensure_perm(GB_Matrix_opaque *S, int64_t k_hot, DevBuf<unsigned int> &poscnt)
{
    const int64_t n = (int64_t)S->ncols, nnz = S->nvals;
    GB_Matrix_opaque *partner = S->tr ? S->tr : S->tr_of;
    const bool adopt = !S->perm && partner && partner->perm;
    if (adopt) {
        S->perm = partner->perm;
        perm_retain(S->perm);
    }
    // (the reference counts by position steer the column ranges of the cold tiles: needed with an adopted order as well)
    DevBuf<unsigned int> cnt(n, true);
    const int hstride = nnz >= ((int64_t)1 << 26) ? 8 : 1;
    hipLaunchKernelGGL(k_hot_hist, dim3((unsigned)ceil_div(nnz, 256 * hstride)), dim3(256), 0, ctx().stream, S->d_col, nnz, cnt.p, hstride);
    if (S->perm) {
        hipLaunchKernelGGL(k_order_poscnt, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, (const unsigned int *)cnt.p,
                           (const int32_t *)S->perm->d_inv, n, poscnt.p);
        sync_stream();
        return S->perm;
    }
    DevBuf<uint64_t> keys(n), keys2(n);
    DevBuf<uint32_t> ids(n), ids2(n);
    hipLaunchKernelGGL(k_order_keys, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, (const unsigned int *)cnt.p, matrix_rowptr(S), n, keys.p, ids.p);
    prim_sort_pairs_u64_u32(keys.p, keys2.p, ids.p, ids2.p, n, 40);
    GB_Perm *P = new GB_Perm();
    P->n = (uint64_t)n;
    try {
        P->d_rank = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)n);
        P->d_inv = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)n);
        DevBuf<unsigned long long> live(2, true);
        hipLaunchKernelGGL(k_order_place, dim3((unsigned)ceil_div(n, 1024)), dim3(1024), 0, ctx().stream, (const uint32_t *)ids2.p, (const uint64_t *)keys2.p, n, k_hot,
                           P->d_rank, P->d_inv, poscnt.p, live.p);
        unsigned long long h_live[2] = {0, 0};
        d2h(h_live, live.p, sizeof(h_live));
        P->n_live_rows = (int64_t)h_live[0];
        P->n_live_cols = (int64_t)h_live[1];
    } catch (...) {
        perm_release(P);
        throw;
    }
    S->perm = P;
    if (partner && !partner->perm) {
        partner->perm = P;
        perm_retain(P);
    }
    return P;
}

// ord = P S P' with its pull layouts; S->ord_state = 1, or -1 when the matrix does not take them (no skew: the split declines)
static void ensure_ordered(GB_Matrix_opaque *S)
{
    const uint64_t sig = order_signature();
    if (S->ord_state == 1 && S->ord && S->ord_sig == sig) return;
    if (S->ord_state == -1 && S->ord_sig == sig) return;
    if (S->ord) {  // (built under other layout options)
        matrix_free(S->ord);
        S->ord = nullptr;
    }
    S->ord_state = -1;
    S->ord_sig = sig;
    const int64_t n = (int64_t)S->ncols, nnz = S->nvals;
    if (n + (int64_t)(1 << 22) > 0x7fffffff || nnz >= 0xf0000000ll) return;
    const size_t vb = S->type->size;
    const int64_t k_hot = hot_table_size(n, vb);
    if (k_hot < 64 || n < ctx().hot_min_cols) return;
    DevBuf<unsigned int> poscnt(n);
    // A matrix whose labels the caller has ranked already (GrX_Matrix_hint_ranked): the twin is the matrix itself, copied -- its layouts are
    // built from the copy, which is released afterwards like any twin's arrays --, positions are the caller's indices, no vector is ever
    // converted, and nothing has to be square: m rows, n columns.
    const bool ranked = S->ranked;
    const int64_t m_rows = (int64_t)S->nrows;
    // A row block set up for a sharded run (GrX_Matrix_shard_setup, round 6): the column order came from the GLOBAL reference counts -- the
    // same on every rank --, the rows stay as they are: the twin is the block with its columns renamed, any shape.
    if (S->col_order_only && !S->perm) S->col_order_only = false;
    const bool col_only = !ranked && S->col_order_only;
    if (!ranked && !col_only && S->nrows != S->ncols) return;  // (an order of the library's own needs one vertex space)
    GB_Perm *P = ranked ? nullptr : (col_only ? S->perm : ensure_perm(S, k_hot, poscnt));
    GB_Matrix_opaque *R = matrix_new(S->type, S->nrows, S->ncols);
    try {
      if (col_only) {
        // reference counts of THIS block by position (they steer the column ranges of its cold tiles)
        DevBuf<unsigned int> cnt(n, true);
        const int hstride = nnz >= ((int64_t)1 << 26) ? 8 : 1;
        hipLaunchKernelGGL(k_hot_hist, dim3((unsigned)ceil_div(nnz, 256 * hstride)), dim3(256), 0, ctx().stream, S->d_col, nnz, cnt.p, hstride);
        hipLaunchKernelGGL(k_order_poscnt, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, (const unsigned int *)cnt.p,
                           (const int32_t *)P->d_inv, n, poscnt.p);
        R->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(m_rows + 1));
        d2d(R->d_ptr, matrix_rowptr(S), sizeof(int64_t) * (size_t)(m_rows + 1));
        R->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnz);
        hipLaunchKernelGGL(k_cols_rename, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, ctx().stream, (const int32_t *)S->d_col, (const int32_t *)P->d_rank, nnz, R->d_col);
        R->d_val = dev_alloc(vb * (size_t)(S->iso ? 1 : nnz));
        d2d(R->d_val, S->d_val, vb * (size_t)(S->iso ? 1 : nnz));
        R->iso = S->iso;
        R->nvals = nnz;
        sync_stream();  // (cnt is released at the end of this scope)
      } else if (ranked) {
        GRB_HIP(hipMemsetAsync(poscnt.p, 0, sizeof(unsigned int) * (size_t)n, ctx().stream));
        const int hstride = nnz >= ((int64_t)1 << 26) ? 8 : 1;
        hipLaunchKernelGGL(k_hot_hist, dim3((unsigned)ceil_div(nnz, 256 * hstride)), dim3(256), 0, ctx().stream, S->d_col, nnz, poscnt.p, hstride);
        R->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(m_rows + 1));
        d2d(R->d_ptr, matrix_rowptr(S), sizeof(int64_t) * (size_t)(m_rows + 1));
        R->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnz);
        d2d(R->d_col, S->d_col, sizeof(int32_t) * (size_t)nnz);
        R->d_val = dev_alloc(vb * (size_t)(S->iso ? 1 : nnz));
        d2d(R->d_val, S->d_val, vb * (size_t)(S->iso ? 1 : nnz));
        R->iso = S->iso;
        R->nvals = nnz;
      } else {
        // ---- the matrix in the new order: row p = row d_inv[p] of S with its columns renamed by d_rank (unsorted inside a row: no
        //      layout below needs them sorted)
        DevBuf<int64_t> len(n + 1);
        hipLaunchKernelGGL(k_twin_lengths, dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, ctx().stream, matrix_rowptr(S), (const int32_t *)P->d_inv, n, len.p);
        R->d_ptr = (int64_t *)dev_alloc(sizeof(int64_t) * (size_t)(n + 1));
        prim_exclusive_sum_i64(len.p, R->d_ptr, n + 1);
        R->d_col = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)nnz);
        R->d_val = dev_alloc(vb * (size_t)(S->iso ? 1 : nnz));
        R->iso = S->iso;
        R->nvals = nnz;
        if (S->iso) d2d(R->d_val, S->d_val, vb);
        DevBuf<unsigned long long> span(1, true);
        hipLaunchKernelGGL(k_twin_huge_span, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, (const int64_t *)R->d_ptr, n, span.p);
        unsigned long long h_span = 0;
        d2h(&h_span, span.p, sizeof(h_span));
        GRB_DISPATCH_TYPE(S->type->code, T, {
            hipLaunchKernelGGL((k_twin_rows<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, matrix_rowptr(S), (const int32_t *)S->d_col,
                               (const T *)S->d_val, S->iso ? 1 : 0, (const int32_t *)P->d_inv, (const int32_t *)P->d_rank, n, (const int64_t *)R->d_ptr,
                               R->d_col, (T *)R->d_val);
            if (h_span > 0)
                hipLaunchKernelGGL((k_twin_huge<T>), dim3((unsigned)h_span), dim3(256), 0, ctx().stream, matrix_rowptr(S), (const int32_t *)S->d_col,
                                   (const T *)S->d_val, S->iso ? 1 : 0, (const int32_t *)P->d_inv, (const int32_t *)P->d_rank, (const int64_t *)R->d_ptr,
                                   R->d_col, (T *)R->d_val);
        })
      }
        {
            DevBuf<unsigned long long> last(1, true);
            hipLaunchKernelGGL(k_twin_live_rows, dim3((unsigned)ceil_div(m_rows, 1024)), dim3(1024), 0, ctx().stream, (const int64_t *)R->d_ptr, m_rows, last.p);
            unsigned long long h_last = 0;
            d2h(&h_last, last.p, sizeof(h_last));
            R->ord_live_rows = (int64_t)h_last;
        }
        // ---- its column codes ARE the hot codes: positions below k_hot are the table, the operand is the image
        R->hot_state = 1;
        R->hot_k = k_hot;
        R->hot_identity = true;
        R->d_col_hot = R->d_col;
        // ---- column ranges of the cold tiles: behind the LDS-resident codes, about equal reference counts, at most ~2 MiB of operand
        {
            const int ncls = (ctx().long_classes == 16 || ctx().long_classes == 32 || ctx().long_classes == 64) ? ctx().long_classes : 8;
            const int64_t lim = std::min<int64_t>(k_hot, long_lds_codes((int)vb, S->type->code == TC_BOOL, LONG_LDS_WORDS) / 8 * ncls);
            const int g = 4096;
            const int64_t nblk = ceil_div(std::max<int64_t>(n - lim, 1), g);
            DevBuf<int64_t> blk(nblk);
            hipLaunchKernelGGL(k_order_coarse, dim3((unsigned)ceil_div(nblk, 4)), dim3(256), 0, ctx().stream, (const unsigned int *)poscnt.p, n, lim, g, nblk, blk.p);
            std::vector<int64_t> h_blk((size_t)nblk);
            d2h(h_blk.data(), blk.p, sizeof(int64_t) * (size_t)nblk);
            int64_t total = 0;
            for (int64_t b = 0; b < nblk; b++) total += h_blk[(size_t)b];
            // (packed cold tiles need EVERY range below 2^19 codes: no last range that takes whatever is left)
            const int R_TARGET = getenv("GRB_ORD_RANGES") ? std::max(1, std::min(48, atoi(getenv("GRB_ORD_RANGES")))) : 32, R_MAX = ctx().ctile_pack ? 128 : 48;
            const int64_t cap_bytes = getenv("GRB_ORD_RANGE_KB") ? (int64_t)atoi(getenv("GRB_ORD_RANGE_KB")) << 10 : (int64_t)2 << 20;
            int64_t cap_codes = std::max<int64_t>(g, cap_bytes / (int64_t)std::max<size_t>(vb, 1));
            // (round 6, packed cold tiles: a column's offset in its range takes 19 bits and the all-ones word is the padding -- ranges stay below 2^19 codes)
            if (ctx().ctile_pack) cap_codes = std::min<int64_t>(cap_codes, ((int64_t)1 << 19) - g);
            const int64_t target = std::max<int64_t>(1, total / R_TARGET);
            std::vector<int32_t> bounds;
            bounds.push_back((int32_t)lim);
            int64_t acc = 0, width = 0;
            for (int64_t b = 0; b < nblk; b++) {
                acc += h_blk[(size_t)b];
                width += g;
                const int64_t end = std::min<int64_t>(lim + (b + 1) * g, n);
                const bool last_ref = (lim + (b + 1) * g >= (P ? P->n_live_cols : n));  // (behind it: columns nobody counted a reference to)
                if ((int)bounds.size() < R_MAX && end < n && (acc >= target || width >= cap_codes || (last_ref && acc > 0))) {
                    bounds.push_back((int32_t)end);
                    acc = 0;
                    width = 0;
                }
            }
            bounds.push_back((int32_t)n);
            R->ct_ncr = (int)bounds.size() - 1;
            R->d_cold_bounds = (int32_t *)dev_alloc(sizeof(int32_t) * bounds.size());
            h2d(R->d_cold_bounds, bounds.data(), sizeof(int32_t) * bounds.size());
        }
        sync_stream();
        // ---- the layouts (hot strips, cold tiles, tagged row groups) in that order
        ensure_split(R, R->d_col_hot, true);
        // (hot strips + cold tiles, or -- BOOL matrices -- class items; the short rows as tagged row groups either way)
        const bool usable = R->split_state == 1 && R->long_nnz > 0 && R->short_tagged_only &&
                            ((R->split_kind == 4 && (R->strip_nseg > 0 || R->ct_units > 0)) || (R->split_kind == 1 && R->n_items > 0));
        if (!usable) {
            matrix_free(R);
            return;
        }
        // (nothing reads the CSR arrays of the twin again: the layouts carry their own columns and values)
        if (R->d_col) {
            dev_free(R->d_col);
            R->d_col = nullptr;
            R->d_col_hot = nullptr;
            R->hot_cols_dropped = true;
        }
        if (!R->iso) {
            dev_free(R->d_val);
            R->d_val = nullptr;
        }
    } catch (...) {
        matrix_free(R);
        throw;
    }
    S->ord = R;
    S->ord_state = 1;
}

// tagged row groups of the short part S of A (once per matrix; see grb_mxv_rows_tag.inc)
// offsets of the row groups (units of TAG_EPL entries) and their "row has an entry" words: from the short part's row pointers alone
static void ensure_tagged_index(GB_Matrix_opaque *A)
{
    if (A->tg_state != 0) return;
    GB_Matrix_opaque *S = A->short_part;
    const int64_t m = (int64_t)S->nrows, ngroups = ceil_div(m, 64);
    const int64_t *sptr = matrix_rowptr(S);
    DevBuf<int64_t> cnt(ngroups + 1);
    A->d_tg_nonempty = (uint64_t *)dev_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(ngroups, 1));
    hipLaunchKernelGGL(k_tag_count, dim3((unsigned)ceil_div((ngroups + 1) * 64, 256)), dim3(256), 0, ctx().stream, sptr, m, ngroups, cnt.p,
                       A->d_tg_nonempty);
    prim_exclusive_sum_i64(cnt.p, cnt.p, ngroups + 1);
    int64_t units = 0;
    d2h(&units, cnt.p + ngroups, 8);
    if (units >= 0x7ffffff0ll) fail(GrB_NOT_IMPLEMENTED, "tagged row groups: too many entries for 32-bit group offsets");  // (unreachable below 2^33 entries: ensure_split's guard)
    A->d_tg_off = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)(ngroups + 1));
    hipLaunchKernelGGL(k_tag_off32, dim3((unsigned)ceil_div(ngroups + 1, 256)), dim3(256), 0, ctx().stream, (const int64_t *)cnt.p, ngroups + 1,
                       A->d_tg_off);
    sync_stream();  // (cnt is released at the end of this scope)
    A->tg_units = units;
    A->tg_state = 2;
}
// ... and their entries: from the short part's CSR arrays (layout build), or -- those are gone and the short rows have sorted row tiles: round 6,
// Context::lazy_tagged -- from the tiles, at the first call the tiles do not take
static void ensure_tagged(GB_Matrix_opaque *A)
{
    if (A->tg_state == 1) return;
    ensure_tagged_index(A);
    GB_Matrix_opaque *S = A->short_part;
    const bool from_csr = S->d_col != nullptr || S->nvals == 0;
    if (!from_csr && A->rt_state != 1) fail(GrB_PANIC, "tagged row groups: the short part's CSR arrays were released and it has no row tiles (internal error)");
    const int64_t m = (int64_t)S->nrows, ngroups = ceil_div(m, 64);
    const int64_t *sptr = matrix_rowptr(S);
    const int64_t units = A->tg_units;
    const size_t ents = (size_t)std::max<int64_t>(units, 1) * TAG_EPL;
    A->d_tg_col = (int32_t *)dev_alloc(sizeof(int32_t) * ents);
    const bool dict = A->vdict_n > 0;  // (dictionary-coded matrices: one byte per value)
    const size_t tg_vb = dict ? 1 : S->type->size;
    A->d_tg_val = S->iso ? nullptr : dev_alloc(tg_vb * ents);
    A->d_tg_tag = (unsigned char *)dev_alloc(ents);
    GRB_HIP(hipMemsetAsync(A->d_tg_col, 0xff, sizeof(int32_t) * ents, ctx().stream));
    if (A->d_tg_val) GRB_HIP(hipMemsetAsync(A->d_tg_val, 0, tg_vb * ents, ctx().stream));
    GRB_HIP(hipMemsetAsync(A->d_tg_tag, 0x40, ents, ctx().stream));
    if (from_csr) {
        GRB_DISPATCH_TYPE(S->type->code, T, {
            hipLaunchKernelGGL((k_tag_fill<T>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx().stream, sptr, (const int32_t *)S->d_col,
                               (const T *)S->d_val, S->iso ? 1 : 0, m, (const int32_t *)A->d_tg_off, A->d_tg_col, (T *)A->d_tg_val, A->d_tg_tag,
                               dict ? (const unsigned long long *)A->d_vd_table : (const unsigned long long *)nullptr,
                               dict ? (const unsigned char *)A->d_vd_codes : (const unsigned char *)nullptr);
        })
    } else {
        // (the tiles hold every entry of the short part: column code, row inside the tile, value or value code)
        const bool is_bool = S->type->code == TC_BOOL;
        const int rows4 = A->rt_rows4 == 16384 ? 16384 : 8192;
        const int rows_cap = (S->type->size > 4 && !is_bool) ? rows4 / 2 : rows4;
        const int vmode = S->iso ? 0 : (dict ? (A->d_rt_val ? 1 : -1) : (int)S->type->size);
        DevBuf<int32_t> gcnt(ngroups + 1, true);
        GRB_DISPATCH_TYPE(S->type->code, T, {
            hipLaunchKernelGGL((k_tag_from_tiles<T>), dim3((unsigned)A->rt_ntiles), dim3(1024), 0, ctx().stream, (const RTile *)A->d_rt_tiles, (const int32_t *)A->d_rt_col,
                               (const uint16_t *)A->d_rt_tag, (const void *)A->d_rt_val, vmode, rows_cap, (const int32_t *)A->d_tg_off, gcnt.p, A->d_tg_col, A->d_tg_val,
                               A->d_tg_tag);
        })
        sync_stream();  // (gcnt is released at the end of this scope)
    }
    sync_stream();
    A->tg_state = 1;
}

// sorted row tiles of the short part S of an ordered twin A (once per matrix; see grb_mxv_rtile.inc).  Needs the tagged row groups'
// per-group "row has an entry" words (ensure_tagged) and S's CSR arrays.
// can k_mxv_rtile address every stream of `units` lane units with its 32-bit byte offsets?  (columns 4 B, tags 2 B, values vb B per entry, RT_EPL entries
// per unit; the sentinel unit 0x0fffffff must stay OUT of range of every descriptor, so the largest stream must end below it as well)
static bool rtile_units_addressable(int64_t units, size_t vb)
{
    const uint64_t per_unit = (uint64_t)RT_EPL * (uint64_t)std::max<size_t>(4, vb);
    return units > 0 && (uint64_t)units * per_unit < 0xffffff00ull && (uint64_t)units < 0x0ffffff0ull;
}

static void ensure_rtile(GB_Matrix_opaque *A)
{
    if (A->rt_state != 0) return;
    A->rt_state = -1;
    GB_Matrix_opaque *S = A->short_part;
    const size_t vs = S->type->size;
    // (4- and 8-byte types with their values; BOOL matrices when they are iso -- the adjacency matrices of the BFS step: k_mxv_rtile_bool)
    const bool is_bool = S->type->code == TC_BOOL;
    if (!S->d_col || S->nvals == 0 || S->nvals >= 0xf0000000ll || A->tg_state == 0) return;
    if (is_bool ? !S->iso : (S->iso || (vs != 4 && vs != 8))) return;
    const int64_t m = (int64_t)S->nrows;
    const int64_t live_rows = A->hot_identity ? std::min<int64_t>(m, std::max<int64_t>(A->ord_live_rows, 1)) : m;
    const int64_t G = ceil_div(live_rows, 64);  // (the groups the row kernels cover; the rows behind them are empty)
    const int64_t rows_end = std::min<int64_t>(G * 64, m);
    const int rows4 = ctx().rtile_rows == 16384 ? 16384 : 8192;
    const int rows_cap = (vs > 4 && !is_bool) ? rows4 / 2 : rows4;
    const int64_t *sptr = matrix_rowptr(S);
    const int64_t nnz = S->nvals;
    DevBuf<int64_t> flag(G + 1), tidx(G + 1);
    // (BOOL tiles: two bits of accumulator per row and 6-byte entries -- twice the entries per tile measured best on the BFS step: 0.399 -> 0.383 ms)
    const int64_t ents_per_tile = std::max<int64_t>(256, ctx().rtile_entries) * (is_bool ? 2 : 1);
    hipLaunchKernelGGL(k_rtile_heads, dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, ctx().stream, sptr, G, ents_per_tile,
                       rows_cap / 64, flag.p);
    GRB_HIP(hipMemsetAsync(flag.p + G, 0, sizeof(int64_t), ctx().stream));
    prim_exclusive_sum_i64(flag.p, tidx.p, G + 1);
    int64_t n_tiles = 0;
    d2h(&n_tiles, tidx.p + G, sizeof(int64_t));
    if (n_tiles <= 0 || n_tiles > (1 << 24)) return;
    DevBuf<int32_t> g0(n_tiles);
    DevBuf<int64_t> e0(n_tiles), units(n_tiles + 1);
    hipLaunchKernelGGL(k_rtile_starts, dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, ctx().stream, sptr, (const int64_t *)flag.p, (const int64_t *)tidx.p, G, g0.p, e0.p);
    hipLaunchKernelGGL(k_rtile_units, dim3((unsigned)ceil_div(n_tiles + 1, 256)), dim3(256), 0, ctx().stream, (const int64_t *)e0.p, n_tiles, nnz, units.p);
    prim_exclusive_sum_i64(units.p, units.p, n_tiles + 1);
    int64_t total_units = 0;
    d2h(&total_units, units.p + n_tiles, sizeof(int64_t));
    // (unit numbers are 28-bit offsets of the kernel's buffer loads -- and the kernel addresses a unit's values at byte u * RT_EPL * sizeof(value)
    //  with 32-bit arithmetic: 8-byte values wrap from 2^27 units on, inside a descriptor clamped to 0xfffffff0 bytes.  The
    //  tagged row groups run when the layout is not built.)
    if (total_units <= 0 || total_units >= 0x0ffffff0ll) return;
    {
        const size_t vb_chk = (A->vdict_n > 0 && vs == 4) ? 1 : (is_bool ? 1 : vs);
        if (!rtile_units_addressable(total_units, vb_chk)) return;
    }
    RTile *tiles = (RTile *)dev_alloc(sizeof(RTile) * (size_t)n_tiles);
    A->d_rt_tiles = tiles;
    hipLaunchKernelGGL(k_rtile_table, dim3((unsigned)ceil_div(n_tiles, 256)), dim3(256), 0, ctx().stream, (const int32_t *)g0.p, (const int64_t *)units.p, n_tiles, G, tiles);
    const size_t ents = (size_t)total_units * RT_EPL;
    const bool dict = A->vdict_n > 0 && vs == 4;
    const size_t vb = dict ? 1 : vs;
    // (round 6: a dictionary-coded matrix with at most 2^24 columns keeps the value code in the top byte of the column word -- no value stream)
    // (the codes of a hot-coded matrix in natural order -- rows_tile = 2 -- index the image [table | u]: ncols + hot_k of them)
    const int64_t n_codes = (int64_t)S->ncols + (A->hot_identity ? 0 : std::max<int64_t>(A->hot_k, 0));
    const bool pack = dict && !is_bool && ctx().rtile_pack && n_codes <= ((int64_t)1 << 24);
    A->d_rt_col = (int32_t *)dev_alloc(sizeof(int32_t) * ents);
    A->d_rt_tag = (uint16_t *)dev_alloc(sizeof(uint16_t) * ents);
    A->d_rt_val = (is_bool || pack) ? nullptr : dev_alloc(vb * ents);
    A->d_rt_counter = (unsigned int *)dev_alloc_zero(64);
    GRB_HIP(hipMemsetAsync(A->d_rt_col, 0xff, sizeof(int32_t) * ents, ctx().stream));
    if (A->d_rt_val) GRB_HIP(hipMemsetAsync(A->d_rt_val, 0, vb * ents, ctx().stream));
    hipLaunchKernelGGL(k_rtile_pad_tags, dim3((unsigned)ceil_div((int64_t)ents, 256)), dim3(256), 0, ctx().stream, A->d_rt_tag, (int64_t)ents, (uint16_t)rows_cap);
    {
        DevBuf<uint64_t> key(nnz), key2(nnz);
        DevBuf<uint32_t> pay(nnz), pay2(nnz);
        DevBuf<uint16_t> rtag(nnz);
        hipLaunchKernelGGL(k_rtile_keys, dim3((unsigned)ceil_div(rows_end, 256)), dim3(256), 0, ctx().stream, sptr, (const int32_t *)S->d_col, rows_end,
                           (const int64_t *)flag.p, (const int64_t *)tidx.p, (const int32_t *)g0.p, key.p, pay.p, rtag.p);
        int tbits = 1;
        while ((1ll << tbits) < n_tiles) tbits++;
        prim_sort_pairs_u64_u32(key.p, key2.p, pay.p, pay2.p, nnz, 32 + tbits);
        GRB_DISPATCH_TYPE(S->type->code, T, {
            if constexpr (sizeof(T) == 4 || sizeof(T) == 8 || std::is_same<T, bool>::value) {
                hipLaunchKernelGGL((k_rtile_place<T>), dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, ctx().stream, (const uint64_t *)key2.p, (const uint32_t *)pay2.p, nnz,
                                   (const int64_t *)e0.p, (const RTile *)tiles, (const uint16_t *)rtag.p, (const T *)S->d_val, A->d_rt_col, A->d_rt_tag, (T *)A->d_rt_val,
                                   dict ? (const unsigned long long *)A->d_vd_table : (const unsigned long long *)nullptr,
                                   dict ? (const unsigned char *)A->d_vd_codes : (const unsigned char *)nullptr, is_bool ? 1 : 0, pack ? 1 : 0);
            }
        })
        sync_stream();  // (the temporaries are released at the end of this scope)
    }
    // hand-out order: heaviest tiles first
    std::vector<RTile> h_tiles((size_t)n_tiles);
    d2h(h_tiles.data(), tiles, sizeof(RTile) * (size_t)n_tiles);
    std::vector<int32_t> order((size_t)n_tiles);
    for (int64_t t = 0; t < n_tiles; t++) order[(size_t)t] = (int32_t)t;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return h_tiles[(size_t)x].n_units > h_tiles[(size_t)y].n_units; });
    A->d_rt_order = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)n_tiles);
    h2d(A->d_rt_order, order.data(), sizeof(int32_t) * (size_t)n_tiles);
    sync_stream();
    A->rt_units = total_units;
    A->rt_ntiles = (int)n_tiles;
    A->rt_rows4 = rows4;
    A->rt_state = 1;
}

template <typename T, int MON, int MUL, int IPT>
static void launch_pull_ipt(GB_Matrix_opaque *A, PullArgs &a)
{
    using W = typename Widen<T>::type;
    // long rows first (lean wavefront-per-chunk kernel straight from A's arrays), then the merge-path kernel over
    // the short part -- when the split exists and the operands are exactly A's own arrays (no typecast copy)
    const bool use_split = A->split_state == 1 && a.aval == A->d_val &&
                           a.col == (A->split_hot ? A->d_col_hot : A->d_col) && a.n_chunks > 0;
    if (use_split) {
        GB_Matrix_opaque *S = A->short_part;
        // (ablation build GRB_STRIP_ABL & 32: one scratch slot per lane of every chunk)
        DevBuf<W> tl_val((GRB_STRIP_ABL & 32) ? std::max<int64_t>(a.n_long, (A->strip_cb[A->strip_ncls > 0 ? A->strip_ncls : 0] + 1) * 64) : a.n_long);
        DevBuf<unsigned char> tl_has(a.n_long);
        const bool by_strip = (A->split_kind == 2 || A->split_kind == 4) && A->long_nnz > 0 && (A->strip_nseg > 0 || A->ct_units > 0);
        ctx().stats.long_entries = A->nvals - S->nvals;
        ctx().stats.long_segments = by_strip ? A->strip_nseg : 0;
        const bool by_class = A->split_kind == 1 && A->long_nnz > 0 && A->n_items > 0;
        ctx().stats.long_kernel = by_strip ? A->split_kind : (by_class ? 1 : 0);
        // (round 6: the short part holds the cold entries of the long rows below tails_max_len entries -- the short-row kernels merge)
        a.long_tails = A->tails_max_len > 0 ? 1 : 0;
        ctx().stats.long_tails = a.long_tails;
        DevBuf<uint32_t> long_act((size_t)ceil_div(a.n_long, 64) * 2);
        a.long_act = long_act.p;
        // the fast kernel of the hot strips (k_mxv_hstrip): a specialised semiring over a full operand whose values are read
        bool hot_fast = false;
        // (not for iso matrices: their padding entries would carry the one stored value instead of 0, and value (+) padding word must
        //  be the identity -- INT64_MAX + v wraps)
        if constexpr (MON >= 0) hot_fast = hstrip_fast_semiring<T>(MON, MUL) && A->split_kind == 4 && by_strip && a.u_full && a.need_uval && a.need_aval && !a.a_iso;
        if (hot_fast) {
            // (the fast kernel reads a class's records through a buffer descriptor with 32-bit offsets: every class below 2 GiB, records of
            //  the layout it is compiled for)
            const int rec_ct = A->vdict_n > 0 ? 24 : 16 + 8 * (int)sizeof(T);
            int64_t widest = 0;
            for (int c = 0; c < A->strip_ncls + A->hub_ncls; c++) widest = std::max<int64_t>(widest, A->strip_cb[c + 1] - A->strip_cb[c]);
            if (A->hrec_bytes != rec_ct || widest * 64 * (int64_t)A->hrec_bytes >= (1ll << 31)) hot_fast = false;
        }
        DevBuf<unsigned char> long_act8(hot_fast && a.has_mask ? (size_t)a.n_long : 1);
        a.long_act8 = long_act8.p;
        // BOOL, a terminal monoid, the operand as pairs or presence bits: the first entries of every admitted long row are tested here
        LongProbe pb{nullptr, 0, nullptr, nullptr, nullptr, 0u};
        if constexpr (std::is_same<T, bool>::value && ((MON == OP_LOR && MUL == OP_LAND) || (MON == OP_ANY && MUL == OP_PAIR))) {
            if (by_class && A->d_probe && ctx().bool_probe > 0 && !a.u_full && (a.u_pv != nullptr || !a.need_uval) && (a.a_iso || !a.need_aval) &&
                a.col == (A->split_hot ? A->d_col_hot : A->d_col)) {
                pb.probe = A->d_probe;
                pb.k = std::min(ctx().bool_probe, A->probe_k);
                pb.pv = a.u_pv;
                pb.pbits = a.u_pv ? nullptr : (const uint32_t *)a.u_bits;
                pb.iso_val = a.need_aval ? (const bool *)a.aval : nullptr;
                pb.want = MON == OP_LOR ? 2u : 1u;
                ctx().stats.long_probe = pb.k;
            }
        }
        hipLaunchKernelGGL((k_long_init<W>), dim3((unsigned)ceil_div(a.n_long, 256)), dim3(256), 0, ctx().stream, tl_val.p, tl_has.p,
                           a.n_long, monoid_identity<T, W>(a.monoid), a.long_rows, a.m_bits, a.has_mask, a.m_comp, long_act.p,
                           ((by_class || by_strip) && a.u_full) ? 1 : 0, (by_strip && acc_is_ordered<W>(a.monoid)) ? 1 : 0,
                           (hot_fast && a.has_mask) ? long_act8.p : nullptr, A->rt_state == 1 ? A->d_rt_counter : nullptr, pb);
        a.tl_ord = (by_strip && acc_is_ordered<W>(a.monoid)) ? 1 : 0;
        a.long_has_known = ((by_class || by_strip) && a.u_full) ? 1 : 0;
        a.tl_val = tl_val.p;
        a.tl_has = tl_has.p;
        a.dbg = ctx().debug_flags;
        const int64_t ncb = ceil_div(A->n_items, (int64_t)COMPACT_BLOCK);
        // (the items of the rows that are switched off -- by the mask, or decided by the bottom-up probe -- are compacted away per call)
        const bool compact = by_class && (a.has_mask || pb.probe != nullptr);
        DevBuf<int64_t> act_start(compact ? (size_t)A->n_items : 1), block_cnt(compact ? (size_t)ncb + 1 : 1), class_off(9);
        DevBuf<int32_t> act_len(compact ? (size_t)A->n_items : 1), act_slot(compact ? (size_t)A->n_items : 1);
        if (by_strip) {
            a.lcol = A->d_lcol;
            a.lval = A->d_lval;
            a.cls_lds_lim = A->cls_lds_lim;
            a.strip_start = A->d_sstart;
            a.strip_slot = A->d_sslot;
            a.strip_slot16 = A->d_sslot16;  // (round 6: 16-bit offsets + a base per chunk; d_sslot is then gone)
            a.strip_slot_base = A->d_sslot_base;
            a.strip_nseg = A->strip_nseg;
            for (int c = 0; c <= A->strip_ncls; c++) a.strip_cb[c] = A->strip_cb[c];
            a.strip_ncls = A->strip_ncls;
            const int64_t G = std::max<int64_t>(A->strip_ncls, (int64_t)(ctx().num_cus / A->strip_ncls) * A->strip_ncls);
            if (A->split_kind == 4) {
                // hot strips (lane records, LDS gathers only), then the cold strips (image gathers) with a token LDS array
                a.hrec = A->d_hrec;
                a.hrec_bytes = A->hrec_bytes;
                a.vdict = A->vdict_n > 0 ? A->d_vdict : nullptr;
                ctx().stats.value_dict = A->vdict_n;
                // the cold entries (tagged tiles, column range by column range per XCD) gather through the texture path and leave HBM idle; the hot
                // strips stream from HBM and leave the texture path idle; both only ADD to the long rows' accumulators (atomics), so their order
                // does not matter: round 6 runs the tiles on the auxiliary stream next to the strips (mxv_overlap).  The short-row kernel that
                // follows reads the accumulators: the main stream waits for the tiles (ev_join) before it.
                auto launch_ctile = [&](hipStream_t on) {
                    if (A->ct_units <= 0) return;
                    a.ct_col = A->d_ct_col;
                    a.ct_val = A->d_ct_val;
                    a.ct_loc = A->d_ct_loc;
                    a.ct_tiles = (const CTile *)A->d_ct_tiles;
                    a.ct_mode = A->ct_mode;
                    a.ct_cbase = A->ct_mode ? (const int32_t *)((const char *)A->d_ct_tiles + sizeof(CTile) * (size_t)std::max<int64_t>(A->ct_ntiles, 1)) : nullptr;
                    if (A->ct_mode == 2) a.vdict = A->d_vdict;  // (the cold tiles' values are dictionary codes)
                    a.ct_order = A->d_ct_order;
                    for (int x = 0; x <= 8; x++) a.ct_xoff[x] = A->ct_xoff[x];
                    const int64_t Gt = std::max<int64_t>(8, (int64_t)(ctx().num_cus * CT_WGS_PER_CU / 8) * 8);
                    hipLaunchKernelGGL((k_mxv_ctile<T, MON, MUL>), dim3((unsigned)Gt), dim3(CT_BLOCK), 0, on, a);
                    ctx().stats.kernel_launches += 1;
                };
                const bool forked = ctx().mxv_overlap && ctx().aux_stream && A->ct_units > 0;
                if (forked) {
                    GRB_HIP(hipEventRecord(ctx().ev_fork, ctx().stream));
                    GRB_HIP(hipStreamWaitEvent(ctx().aux_stream, ctx().ev_fork, 0));
                    launch_ctile(ctx().aux_stream);
                    GRB_HIP(hipEventRecord(ctx().ev_join, ctx().aux_stream));
                }
                // the fast kernel takes both levels of an ordered matrix in ONE launch (workgroups dealt to the classes by chunk count)
                bool merged = false;
                if constexpr (MON >= 0) {
                    if constexpr (hstrip_fast_semiring<T>(MON, MUL)) {
                        if (hot_fast && A->hub_ncls > 0 && A->d_wg_tab && A->wg_tab_g > 0 && A->strip_cb[A->strip_ncls + A->hub_ncls] > 0) {
                            PullArgs al = a;
                            al.wg_tab = A->d_wg_tab;
                            al.strip_cb_dev = A->d_strip_cb;
                            al.strip_chunks = A->strip_cb[A->strip_ncls + A->hub_ncls];
                            bool dict_launched = false;
                            const bool s16 = al.strip_slot16 != nullptr;
                            if constexpr (sizeof(T) == 4) {
                                if (al.vdict) {
                                    if (s16) hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, true, true>), dim3((unsigned)A->wg_tab_g), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                    else hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, true>), dim3((unsigned)A->wg_tab_g), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                    dict_launched = true;
                                }
                            }
                            if (!dict_launched) {
                                if (s16) hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, false, true>), dim3((unsigned)A->wg_tab_g), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                else hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS>), dim3((unsigned)A->wg_tab_g), dim3(LONG_BLOCK), 0, ctx().stream, al);
                            }
                            merged = true;
                        }
                    }
                }
                // otherwise level 0: the classes of all long rows; level 1 (an ordered matrix): the 64 classes of its hub rows, strip_cb[strip_ncls ..]
                for (int level = 0; !merged && level < (A->hub_ncls > 0 ? 2 : 1); level++) {
                    const int c0 = level ? A->strip_ncls : 0, nc = level ? A->hub_ncls : A->strip_ncls;
                    if (A->strip_cb[c0 + nc] == A->strip_cb[c0]) continue;  // (no chunk at this level)
                    PullArgs al = a;
                    al.strip_ncls = nc;
                    for (int c = 0; c <= nc; c++) al.strip_cb[c] = A->strip_cb[c0 + c];
                    al.cls_lds_lim = level ? A->hub_lds_lim : A->cls_lds_lim;
                    const int64_t Gl = std::max<int64_t>(nc, (int64_t)(ctx().num_cus / nc) * nc);
                    bool launched = false;
                    if constexpr (MON >= 0) {
                        if constexpr (hstrip_fast_semiring<T>(MON, MUL)) {
                            if (hot_fast) {
                                bool dict_launched = false;
                                const bool s16 = al.strip_slot16 != nullptr;
                                if constexpr (sizeof(T) == 4) {
                                    if (al.vdict) {
                                        if (s16) hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, true, true>), dim3((unsigned)Gl), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                        else hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, true>), dim3((unsigned)Gl), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                        dict_launched = true;
                                    }
                                }
                                if (!dict_launched) {
                                    if (s16) hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS, false, true>), dim3((unsigned)Gl), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                    else hipLaunchKernelGGL((k_mxv_hstrip<T, MON, MUL, LONG_LDS_WORDS>), dim3((unsigned)Gl), dim3(LONG_BLOCK), 0, ctx().stream, al);
                                }
                                launched = true;
                            }
                        }
                    }
                    if (!launched)
                        hipLaunchKernelGGL((k_mxv_strip<T, MON, MUL, LONG_LDS_WORDS, 1>), dim3((unsigned)Gl), dim3(LONG_BLOCK), 0, ctx().stream, al);
                    if (level) ctx().stats.kernel_launches += 1;
                }
                if (forked) GRB_HIP(hipStreamWaitEvent(ctx().stream, ctx().ev_join, 0));
                else launch_ctile(ctx().stream);
            } else
            hipLaunchKernelGGL((k_mxv_strip<T, MON, MUL, LONG_LDS_WORDS>), dim3((unsigned)G), dim3(LONG_BLOCK), 0, ctx().stream, a);
        } else if (by_class) {
            a.lcol = A->d_lcol;
            a.lval = A->d_lval;
            a.it_start = A->d_it_start;
            a.it_len = A->d_it_len;
            a.it_slot = A->d_it_slot;
            for (int c = 0; c < 9; c++) a.item_begin[c] = A->item_begin[c];
            a.cls_lds_lim = A->cls_lds_lim;
            a.class_off = nullptr;
            if (compact) {
                hipLaunchKernelGGL(k_long_compact_count, dim3((unsigned)ncb), dim3(256), 0, ctx().stream, (const int32_t *)A->d_it_slot,
                                   A->n_items, (const uint32_t *)long_act.p, block_cnt.p);
                prim_exclusive_sum_i64(block_cnt.p, block_cnt.p, ncb + 1);
                hipLaunchKernelGGL(k_long_compact_write, dim3((unsigned)ncb), dim3(256), 0, ctx().stream, (const int32_t *)A->d_it_slot,
                                   A->n_items, (const uint32_t *)long_act.p, (const int64_t *)block_cnt.p,
                                   (const int64_t *)A->d_item_begin, (const int64_t *)A->d_it_start, (const int32_t *)A->d_it_len,
                                   act_start.p, act_len.p, act_slot.p, class_off.p);
                a.it_start = act_start.p;
                a.it_len = act_len.p;
                a.it_slot = act_slot.p;
                a.class_off = class_off.p;
                ctx().stats.kernel_launches += 2;
            }
            // one persistent 1024-thread workgroup per CU; block b works on column class b % 8 (= the XCD it runs on)
            const int64_t G = std::max<int64_t>(8, (int64_t)(ctx().num_cus / 8) * 8);
            hipLaunchKernelGGL((k_mxv_long_grp<T, MON, MUL, LONG_LDS_WORDS>), dim3((unsigned)G), dim3(LONG_BLOCK), 0, ctx().stream, a);
        } else if (a.long_tails) {
            // (neither strips nor cold tiles were built.  With cold entries of long rows in the short part that means EVERY entry of every long row
            //  is there -- hub rows, which keep theirs, always produce strips or tiles -- and the chunk kernel, which walks whole rows of A, must not run)
            if (A->long_nnz > 0) fail(GrB_PANIC, "pull SpMV: long rows with entries in the short part but no strips (internal error)");
        } else {
            const int64_t want = ceil_div(a.n_chunks, LONG_BLOCK / 64);
            const int64_t G = std::min<int64_t>(want, (int64_t)ctx().num_cus);  // persistent: one 1024-thread workgroup per CU
            hipLaunchKernelGGL((k_mxv_long<T, MON, MUL, LONG_LDS_WORDS>), dim3((unsigned)G), dim3(LONG_BLOCK), 0, ctx().stream, a);
        }
        ctx().stats.kernel_launches += 2;
        PullArgs b = a;
        b.rowptr = matrix_rowptr(S);
        b.col = S->d_col;
        b.aval = S->d_val;
        b.nnz = S->nvals;
        b.long_bits = A->d_long_bits;
        b.n_chunks = 0;
        b.n_long_epi = a.n_long;
        const int sk = short_kernel_for(A);
        if ((sk == 5 || A->short_tagged_only) && S->nrows == A->nrows && S->nvals < 0x1ffffffffll) {
            // short rows as tagged row groups: the row of every entry is stored with it (no marks, no scan, no segmented fold)
            // (round 6: the groups' offsets and non-empty words now, their entries only if the call ends up with k_mxv_rows_tag -- lazy_tagged)
            ensure_tagged_index(A);
            b.long_prefix = A->d_long_prefix;
            b.tg_off = A->d_tg_off;
            b.vdict = A->vdict_n > 0 ? A->d_vdict : nullptr;
            b.tg_nonempty = A->d_tg_nonempty;
            int64_t groups = ceil_div(b.m, 64);
            b.tg_groups = 0;
            b.tg_stride = 0;
            if (A->hot_identity) {
                // a matrix in its popularity order: rows sorted by falling weight, the empty ones at the end
                const int64_t live_groups = std::max<int64_t>(1, ceil_div(A->ord_live_rows, 64));
                // (an output written into fresh buffers keeps its tail in the kernel: the values move too -- unless nothing of the tail survives
                //  the call: replace without an accumulator, the level step of a BFS whose frontier is its own output)
                const bool tail_dies = b.replace && b.accum < 0;
                if (live_groups < groups && (!b.fresh || tail_dies)) {
                    if (!(b.accum >= 0 && !b.replace)) {
                        const int64_t tail = groups - live_groups;
                        hipLaunchKernelGGL(k_rows_tail, dim3((unsigned)ceil_div(tail, 256)), dim3(256), 0, ctx().stream, b, live_groups);
                        ctx().stats.kernel_launches += 1;
                    }
                    groups = live_groups;
                    b.tg_groups = live_groups;
                }
                b.tg_stride = ceil_div(groups, TAG_K);
                if (b.tg_groups == 0) b.tg_groups = groups;
            }
            // an ordered BOOL matrix, a specialised semiring, the operand as presence / value pairs: two persistent workgroups per CU with the
            // pairs of the 276 Ki hottest columns in LDS (k_mxv_rows_tag<..., HEAD>; measured: 266 -> 238 us on the level step of scale 24.
            // For 4-byte values the head holds 17 Ki columns, a third of the references, and the kernel is 13 % SLOWER than without)
            // the sorted row tiles of an ordered matrix (round 5, grb_mxv_rtile.inc): a specialised semiring over a full operand (or an image
            // with the absorbing value under its absent entries) whose values are read, values of the matrix read, output in place
            if constexpr (MON >= 0 && !std::is_same<T, bool>::value && (sizeof(T) == 4 || sizeof(T) == 8)) {
                if (ctx().rows_tile && A->rt_state == 1 && b.u_full && b.need_uval && b.need_aval && !b.a_iso &&
                    (A->hot_identity ? (b.tg_groups > 0 && b.tg_groups == ceil_div(std::min<int64_t>(b.m, std::max<int64_t>(A->ord_live_rows, 1)), 64)) : b.tg_groups == 0) &&
                    b.m * (int64_t)sizeof(T) < 0xfffffff0ll) {  // (the old values of a tile's rows are read through a buffer descriptor)
                    b.rt_col = A->d_rt_col;
                    b.rt_tag = A->d_rt_tag;
                    b.rt_val = A->d_rt_val;
                    b.rt_tiles = (const RTile *)A->d_rt_tiles;
                    b.rt_order = A->d_rt_order;
                    b.rt_counter = A->d_rt_counter;
                    b.rt_units = A->rt_units;
                    b.rt_ntiles = A->rt_ntiles;
                    const bool dict = b.vdict != nullptr && sizeof(T) == 4;
                    const bool pack = dict && A->d_rt_val == nullptr;  // (value codes in the top byte of the column words: ensure_rtile)
                    const bool tall = A->rt_rows4 == 16384;
                    const int64_t G = std::min<int64_t>(A->rt_ntiles, (int64_t)ctx().num_cus * (tall ? 2 : 4));
                    bool launched = false;
                    if constexpr (sizeof(T) == 4) {
                        if (dict && pack) {
                            if (tall) hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 16384, true, true>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                            else hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 8192, true, true>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                            launched = true;
                        } else if (dict) {
                            if (tall) hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 16384, true>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                            else hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 8192, true>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                            launched = true;
                        }
                    }
                    if (!launched) {
                        if (tall) hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 16384, false>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                        else hipLaunchKernelGGL((k_mxv_rtile<T, MON, MUL, 8192, false>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                    }
                    GRB_HIP(hipGetLastError());
                    ctx().stats.kernel_launches += 1;
                    ctx().stats.fused_epilogue = 3;  // (bookkeeping: 3 = fused, by the sorted row tiles)
                    ctx().stats.tiles = A->rt_ntiles;
                    return;
                }
            }
            // ... and BOOL products over an operand given as presence / value pairs (the BFS level step): k_mxv_rtile_bool
            if constexpr (std::is_same<T, bool>::value && ((MON == OP_LOR && MUL == OP_LAND) || (MON == OP_ANY && MUL == OP_PAIR))) {
                if (ctx().rows_tile && A->rt_state == 1 && !b.u_full && (b.u_pv != nullptr || !b.need_uval) && (b.a_iso || !b.need_aval) &&
                    (A->hot_identity ? (b.tg_groups > 0 && b.tg_groups == ceil_div(std::min<int64_t>(b.m, std::max<int64_t>(A->ord_live_rows, 1)), 64)) : b.tg_groups == 0) &&
                    b.m < 0xfffffff0ll) {
                    b.rt_col = A->d_rt_col;
                    b.rt_tag = A->d_rt_tag;
                    b.rt_val = nullptr;
                    b.rt_tiles = (const RTile *)A->d_rt_tiles;
                    b.rt_order = A->d_rt_order;
                    b.rt_counter = A->d_rt_counter;
                    b.rt_units = A->rt_units;
                    b.rt_ntiles = A->rt_ntiles;
                    const int64_t G = std::min<int64_t>(A->rt_ntiles, (int64_t)ctx().num_cus * 4);
                    if (A->rt_rows4 == 16384) hipLaunchKernelGGL((k_mxv_rtile_bool<MON, MUL, 16384>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                    else hipLaunchKernelGGL((k_mxv_rtile_bool<MON, MUL, 8192>), dim3((unsigned)G), dim3(RT_BLOCK), 0, ctx().stream, b);
                    GRB_HIP(hipGetLastError());
                    ctx().stats.kernel_launches += 1;
                    ctx().stats.fused_epilogue = 3;
                    ctx().stats.tiles = A->rt_ntiles;
                    return;
                }
            }
            // (neither kind of row tiles took the call: the tagged row groups -- their entries laid out now if they were not yet)
            ensure_tagged(A);
            b.tg_col = A->d_tg_col;
            b.tg_val = A->d_tg_val;
            b.tg_tag = A->d_tg_tag;
            bool head = false;
            if constexpr (MON >= 0 && std::is_same<T, bool>::value) {
                head = ctx().rows_head && A->hot_identity && b.tg_stride > 0 && b.u_pv != nullptr && groups >= ctx().rows_head_min_groups;
                if (head) {
                    const int64_t G = std::min<int64_t>(2 * (int64_t)ctx().num_cus, ceil_div(groups, (int64_t)(TAG_HEAD_BLOCK / 64)));
                    hipLaunchKernelGGL((k_mxv_rows_tag<T, MON, MUL, true>), dim3((unsigned)G), dim3(TAG_HEAD_BLOCK), 0, ctx().stream, b);
                    ctx().stats.fused_epilogue = 2;  // (bookkeeping: 2 = fused, by the kernel with the LDS head)
                }
            }
            if (!head)
                hipLaunchKernelGGL((k_mxv_rows_tag<T, MON, MUL>), dim3((unsigned)ceil_div(ceil_div(groups, TAG_K), ROWS_BLOCK / 64)), dim3(ROWS_BLOCK), 0,
                                   ctx().stream, b);
            GRB_HIP(hipGetLastError());
            ctx().stats.kernel_launches += 1;
            ctx().stats.tiles = ceil_div(b.m, 64);
            return;
        }
        if (a.long_tails) fail(GrB_PANIC, "pull SpMV: the short part holds entries of long rows, which only the tagged row groups and the row tiles merge (internal error)");
        if (sk != 0 && S->nrows == A->nrows) {
            // short rows: one wavefront per 64 consecutive rows, which also applies the write rule of the long rows
            b.long_prefix = A->d_long_prefix;
            // (persistent variants -- static strides with the next group prefetched, or an LDS work counter per workgroup --
            //  measured 3-10 % slower than one group per wavefront)
            hipLaunchKernelGGL((k_mxv_rows<T, MON, MUL>), dim3((unsigned)ceil_div(ceil_div(b.m, 64), ROWS_BLOCK / 64)), dim3(ROWS_BLOCK), 0,
                               ctx().stream, b);
            GRB_HIP(hipGetLastError());
            ctx().stats.kernel_launches += 1;
            ctx().stats.tiles = ceil_div(b.m, 64);
            return;
        }
        launch_pull_ipt<T, MON, MUL, IPT>(S, b);  // S has no split of its own: takes the plain path below
        return;
    }
    if (!a.col && a.nnz > 0) fail(GrB_PANIC, "pull SpMV: the plain path was reached with the re-coded columns released (internal error)");
    constexpr int TILE = PULL_BLOCK * IPT;
    a.dbg = ctx().debug_flags;
    ensure_tile_table(A, TILE);
    a.tile_row = A->d_tile_row;
    a.n_tiles = A->n_tiles;
    if (a.n_tiles > 0x7fffffff) fail(GrB_NOT_IMPLEMENTED, "too many tiles for one launch");
    DevBuf<W> carry_val(a.n_tiles), first_val(a.n_tiles);
    DevBuf<uint8_t> carry_has(a.n_tiles), first_has(a.n_tiles);
    a.carry_val = carry_val.p;
    a.first_val = first_val.p;
    a.carry_has = carry_has.p;
    a.first_has = first_has.p;
    DevBuf<long long> dbg_times((ctx().debug_flags & 8) ? (size_t)a.n_tiles * 10 : 1, true);
    a.dbg_times = (ctx().debug_flags & 8) ? dbg_times.p : nullptr;
    hipLaunchKernelGGL((k_mxv_pull<T, MON, MUL, IPT>), dim3((unsigned)a.n_tiles), dim3(PULL_BLOCK), 0, ctx().stream, a);
    if (a.dbg_times) report_phase_times(dbg_times.p, a.n_tiles);
    hipLaunchKernelGGL((k_mxv_seams<T, TILE>),
                       dim3((unsigned)(ceil_div(a.n_tiles, PULL_BLOCK / 64) + ceil_div(a.n_long_epi, PULL_BLOCK))), dim3(PULL_BLOCK), 0,
                       ctx().stream, a);
    GRB_HIP(hipGetLastError());
    ctx().stats.kernel_launches += 2;
    ctx().stats.tiles = a.n_tiles;
}

template <typename T, int MON, int MUL>
static void launch_pull(GB_Matrix_opaque *A, PullArgs &a)
{
    // GRB_PULL_IPT (tuning knob) is honoured by the fully specialised kernels only
    if constexpr (MON >= 0) {
        const int want = ctx().tune_pull_ipt;
        if (want == 4 && sizeof(T) <= 4) return launch_pull_ipt<T, MON, MUL, 4>(A, a);
        if (want == 16 && sizeof(T) <= 4) return launch_pull_ipt<T, MON, MUL, 16>(A, a);
    }
    launch_pull_ipt<T, MON, MUL, PullIPT<T>::value>(A, a);
}

static void pull_dispatch(GB_Matrix_opaque *A, int type, PullArgs &a)
{
    const int mon = a.monoid, mul = a.mult;
    if (mul == OP_PAIR && a.u_full && !(ctx().debug_flags & 65536)) {
        GRB_DISPATCH_TYPE(type, T, {
            const int64_t nthreads = (int64_t)bits_words64((uint64_t)a.m) * 64;
            hipLaunchKernelGGL((k_mxv_rowlen<T>), dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 0, ctx().stream, a);
        })
        ctx().stats.kernel_launches += 1;
        ctx().stats.method = 5;
        return;
    }
    // hot semirings get fully specialised kernels; everything else runs the runtime-operator kernel
#define SPECIAL(TC, CT, MON, MUL)                                \
    if (type == TC && mon == MON && mul == MUL) {                \
        launch_pull<CT, MON, MUL>(A, a);                         \
        return;                                                  \
    }
    SPECIAL(TC_FP32, float, OP_MIN, OP_PLUS)
    SPECIAL(TC_FP64, double, OP_MIN, OP_PLUS)
    SPECIAL(TC_INT64, int64_t, OP_MIN, OP_PLUS)
    SPECIAL(TC_FP32, float, OP_PLUS, OP_TIMES)
    SPECIAL(TC_FP64, double, OP_PLUS, OP_TIMES)
    SPECIAL(TC_INT64, int64_t, OP_PLUS, OP_TIMES)
    SPECIAL(TC_BOOL, bool, OP_LOR, OP_LAND)
    SPECIAL(TC_BOOL, bool, OP_ANY, OP_PAIR)
    SPECIAL(TC_INT64, int64_t, OP_ANY, OP_PAIR)
    SPECIAL(TC_FP32, float, OP_ANY, OP_PAIR)
#undef SPECIAL
    GRB_DISPATCH_TYPE(type, T, { launch_pull<T, -1, -1>(A, a); })
}

struct DescFlags {
    bool replace = false, comp = false, structure = false, t0 = false, t1 = false;
};
static DescFlags flags_of(const GB_Descriptor_opaque *d)
{
    DescFlags f;
    if (d) { f.replace = d->replace; f.comp = d->comp; f.structure = d->structure; f.t0 = d->t0; f.t1 = d->t1; }
    return f;
}

// w<mask> = accum(w, S (+.x) u) where S is the CSR to pull over (A or its cached transpose);
// `flip` evaluates mult(u_k, S_ik) instead of mult(S_ik, u_k)  (vxm).
static void mxv_core(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum,
                     const GB_Semiring_opaque *sr, GB_Matrix_opaque *S, GB_Vector_opaque *u, bool flip, DescFlags f)
{
    if (S->ncols != u->n) fail(GrB_DIMENSION_MISMATCH, "mxv/vxm: matrix inner dimension " + std::to_string(S->ncols) + " does not match vector size " + std::to_string(u->n));
    if (w->n != S->nrows) fail(GrB_DIMENSION_MISMATCH, "mxv/vxm: output size " + std::to_string(w->n) + " does not match matrix dimension " + std::to_string(S->nrows));
    if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "mxv/vxm: mask size does not match output size");
    if (accum && (accum->type != w->type->code || op_is_comparison(accum->op))) fail(GrB_DOMAIN_MISMATCH, "mxv/vxm: accum operator type must equal the output type");
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 1;
    ctx().stats.long_kernel = -1;
    ctx().stats.flops = S->nvals;
    ctx().stats.out_nvals = -1;
    const int64_t m = (int64_t)w->n;
    if (m == 0) return;
    if (!mask && f.comp) {  // complement of "no mask": nothing may be written
        if (f.replace) vector_release_storage(w);
        return;
    }
    const int st = sr->type;
    const int monoid = canonical_op(st, sr->monoid);
    int mult = canonical_op(st, sr->mult);
    if (flip) mult = flip_op(mult);

    // ---- mask bits ------------------------------------------------------------------------------------------
    DevBuf<uint64_t> mbits_tmp(0);
    const uint64_t *m_bits = nullptr;
    if (mask) {
        // (a mask that aliases w is snapshotted: tiles update w's presence words while others still read them)
        if (f.structure && mask->d_val && mask != w) m_bits = mask->d_bits;
        else {
            dev_free(mbits_tmp.p);
            mbits_tmp.p = (uint64_t *)dev_alloc(bits_words64(mask->n) * 8);
            vector_mask_bits(mask, f.structure, mbits_tmp.p);
            m_bits = mbits_tmp.p;
        }
    }

    // ---- an operand without entries: T is empty, only the write rule remains (and nothing of the size of u is touched: empty
    //      objects may be as large as GrB_INDEX_MAX + 1) ---------------------------------------------------------------------
    if (S->nvals == 0 || u->nvals == 0) {
        ctx().stats.method = 6;
        if (w->nvals == 0) return;  // nothing to keep, nothing to write
        if (!mask && !accum) {      // w = T = empty
            vector_release_storage(w);
            return;
        }
        vector_ensure_storage(w);
        DevBuf<uint64_t> t_bits(bits_words64(w->n), true);
        const int acc_op = accum ? canonical_op(w->type->code, accum->op) : -1;
        vector_write_rule(w, w->d_val, t_bits.p, m_bits, f.comp, acc_op, f.replace);
        ctx().stats.kernel_launches += 1;
        ctx().stats.method = 6;
        w->nvals = -1;
        if (ctx().blocking) sync_stream();
        return;
    }

    // ---- operands in the semiring's type (only the ones the multiply operator reads) ------------------------------------
    bool need_aval = !(mult == OP_PAIR || mult == OP_SECOND) && S->nvals > 0;
    const bool need_uval = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
    if (mult == OP_ANY) need_aval = S->nvals > 0;
    DevBuf<char> a_cast(0), u_cast(0);
    const void *aval = S->d_val;
    if (need_aval && S->type->code != st) {
        const int64_t nv = S->iso ? 1 : S->nvals;
        dev_free(a_cast.p);
        a_cast.p = (char *)dev_alloc(type_size(st) * (size_t)nv);
        cast_array(st, a_cast.p, S->type->code, S->d_val, nv);
        aval = a_cast.p;
    }
    vector_ensure_storage(u);
    const void *uval = u->d_val;
    // (values nobody reads still travel with the presence image [hot table | u] that is built when u is not full, and that
    //  image is laid out in the semiring's type: only a full operand whose values are unused skips the cast)
    const bool u_full_now = u->nvals == (int64_t)u->n;
    if ((need_uval || !u_full_now) && u->type->code != st) {
        dev_free(u_cast.p);
        u_cast.p = (char *)dev_alloc(type_size(st) * (size_t)u->n);
        cast_array(st, u_cast.p, u->type->code, u->d_val, (int64_t)u->n);
        uval = u_cast.p;
    }

    PullArgs a{};
    a.m = m;
    a.nnz = S->nvals;
    a.stream_nt = S->nvals >= ctx().stream_nt_min_nnz ? 1 : 0;  // (a matrix much larger than the 256 MB infinity cache: its streams bypass the caches)
    a.rowptr = matrix_rowptr(S);
    a.col = S->d_col;
    a.aval = aval;
    a.a_iso = S->iso ? 1 : 0;
    a.u_val = uval;
    a.u_bits = (const uint32_t *)u->d_bits;
    a.u_full = (u->nvals == (int64_t)u->n) ? 1 : 0;
    a.monoid = monoid;
    a.mult = mult;
    a.need_aval = need_aval;
    a.need_uval = need_uval;
    // PAIR over a full operand: every entry of a row contributes the same 1, so the product is a function of the row length
    // alone (k_mxv_rowlen) -- the aggregators count / exists are this case (reference core/operator/agg.py:264-283, :360-378)
    const bool by_rowlen = (mult == OP_PAIR && a.u_full && !(ctx().debug_flags & 65536));
    a.x_len = (int64_t)u->n;
    // (an operand that is never read -- full, values unused: the row reductions -- is not subject to the 32-bit buffer range)
    if ((a.need_uval || !a.u_full) && (uint64_t)u->n * type_size(st) >= 0xff000000ull)
        fail(GrB_NOT_IMPLEMENTED, "mxv/vxm: input vectors of 4 GiB or more are not supported by the pull kernel yet");
    // hot-column table (wide matrices with a skewed column-degree distribution): the kernel indexes ONE image
    // [ K hot entries | the n entries of u ] with the re-coded column indices (hot rank, or K + col)
    DevBuf<char> xcat_val(0);
    DevBuf<uint64_t> xcat_bits(0);
    // the first pull over a large matrix runs on the CSR arrays as they are; the layouts below (tens of milliseconds to build at
    // scale 24) are built when the matrix comes back for a second product
    const bool lazy = ctx().lazy_layout && S->nvals >= ctx().lazy_min_nnz && S->pull_calls == 0 && S->hot_state == 0 && S->split_state == 0;
    S->pull_calls++;
    if (S->hot_identity) {
        // a matrix in its popularity order (grb_mxv_order.inc): the column codes are positions of the operand itself -- the first hot_k of
        // them are the hot table, nothing is gathered per call
        a.col = S->d_col_hot;
        ctx().stats.hot_k = S->hot_k;
    } else if ((a.need_uval || !a.u_full) && !lazy) {
        ensure_hot(S, type_size(st));
        // (once the split is built from the re-coded columns, the re-coded copy of the WHOLE column array is released --
        //  1.05 GB of the 3.7 GB of layouts at scale 24: a call that cannot take the split then runs on the plain arrays)
        const bool split_usable = S->nvals && (S->type->code == st || !need_aval) && !by_rowlen;
        if (S->hot_state == 1 && (!S->hot_cols_dropped || split_usable) && (uint64_t)(u->n + S->hot_k) * type_size(st) < 0xff000000ull) {
            const int k = (int)S->hot_k;  // multiple of 64
            const size_t vb = type_size(st);
            char *img_val;
            uint64_t *img_bits;
            if (u->padded && uval == u->d_val && vb * (size_t)k <= VEC_VAL_PAD && (size_t)k / 8 <= VEC_BITS_PAD) {
                // u's own allocation has room in front of its values and presence words: the table is gathered there and
                // [table | u] is one image without copying u
                img_val = (char *)u->d_val - vb * (size_t)k;
                img_bits = u->d_bits - k / 64;
                GRB_DISPATCH_TYPE(st, T, {
                    hipLaunchKernelGGL((k_x_image<T>), dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, ctx().stream,
                                       (const int32_t *)S->d_hot_cols, k, (const T *)uval, (const uint32_t *)u->d_bits, a.u_full,
                                       (T *)img_val, img_bits, (int64_t)u->n);
                })
            } else {
                dev_free(xcat_val.p);
                xcat_val.p = (char *)dev_alloc(vb * (size_t)(k + u->n));
                dev_free(xcat_bits.p);
                xcat_bits.p = (uint64_t *)dev_alloc((size_t)(k / 64 + bits_words64(u->n)) * 8);
                img_val = xcat_val.p;
                img_bits = xcat_bits.p;
                const int64_t copy_threads = std::max<int64_t>(((int64_t)u->n * (int64_t)vb >> 4) + 1, (int64_t)bits_words64(u->n));
                if (((uintptr_t)uval & 15u) == 0) {
                    GRB_DISPATCH_TYPE(st, T, {
                        hipLaunchKernelGGL((k_x_image<T>), dim3((unsigned)(ceil_div(k, 256) + ceil_div(copy_threads, 256))), dim3(256), 0,
                                           ctx().stream, (const int32_t *)S->d_hot_cols, k, (const T *)uval, (const uint32_t *)u->d_bits,
                                           a.u_full, (T *)img_val, img_bits, (int64_t)u->n);
                    })
                } else {  // (a typecast copy of u need not be 16-byte aligned)
                    GRB_DISPATCH_TYPE(st, T, {
                        hipLaunchKernelGGL((k_x_image<T>), dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, ctx().stream,
                                           (const int32_t *)S->d_hot_cols, k, (const T *)uval, (const uint32_t *)u->d_bits, a.u_full,
                                           (T *)img_val, img_bits, (int64_t)u->n);
                    })
                    d2d(img_val + vb * (size_t)k, uval, vb * (size_t)u->n);
                    if (!a.u_full) d2d(img_bits + k / 64, u->d_bits, bits_words64(u->n) * 8);
                }
            }
            ctx().stats.kernel_launches += 1;
            ctx().stats.hot_k = k;
            a.col = S->d_col_hot;
            a.u_val = img_val;
            a.u_bits = (const uint32_t *)img_bits;
            a.x_len = (int64_t)u->n + k;
        }
    }
    // long/short row split (large matrices whose long rows hold a good share of the entries)
    if (S->nvals && (S->type->code == st || !need_aval) && !by_rowlen && !lazy) {
        bool hot = (a.col == S->d_col_hot);
        // An existing split is never rebuilt because a call comes with the other column coding (mixing row reductions and
        // products on one matrix used to rebuild the layouts -- tens of milliseconds at scale 24 -- at every call, and a
        // rebuild after the re-coded columns were released read a null column array): a call that reads no column at all (a
        // full operand whose values the multiply ignores: the row reductions) takes the split as it is; a call that gathers
        // with the original column indices while the split holds table codes runs on the plain arrays.  Only a split in the
        // original coding is replaced when a call can use the hot table.
        const bool reads_cols = a.need_uval || !a.u_full;
        if (S->split_state == 1 && S->split_hot != hot && !reads_cols) {
            hot = S->split_hot;
            a.col = hot ? S->d_col_hot : S->d_col;
        } else if (S->split_state == 1 && S->split_hot && !hot) {
            // (plain path below: a.col is the original column array, which use_split does not match)
        } else {
            ensure_split(S, a.col, hot);
        }
        if (hot && S->hot_cols_dropped) a.col = S->d_col_hot;  // (released when the split was built: the tag of the re-coded columns is now nullptr)
        if (S->split_state == 1 && S->split_hot == hot) {
            a.long_rows = S->d_long_rows;
            a.chunk_slot = S->d_chunk_slot;
            a.chunk_start = S->d_chunk_start;
            a.chunk_len = S->d_chunk_len;
            a.n_chunks = S->n_chunks;
            a.n_long = S->n_long;
        }
    }
    // A sparse operand of a floating-point min_plus / max_plus product on an ordered matrix: run it as a FULL operand whose absent entries
    // hold the multiply's absorbing value (+inf: a + inf = inf = the identity of min; -inf for max).  With every matrix value finite (known
    // from the dictionary's scan) and every present operand value finite (checked by the pass that builds the image: one host read) a row's
    // accumulator leaves the identity iff the row meets a present entry -- the product's pattern is exact -- and the kernels skip the
    // presence gathers and take the fast hot-strip path: the sweeps of an SSSP loop, whose distance vector never becomes full, went from
    // 1.53 to ~0.8 ms (section 4.1.10)
    DevBuf<char> fill_img(0);
    // (vals_finite comes with the dictionary, i.e. with FP32 today; the bound on the operand keeps every product finite)
    const double fill_limit = (st == TC_FP32 ? 3.4028234663852886e38 : 1.7976931348623157e308) - S->vals_absmax;
    if (S->hot_identity && !a.u_full && a.need_uval && ctx().fill_absent && S->vals_finite && fill_limit > 0 && !S->iso && (st == TC_FP32 || st == TC_FP64) &&
        mult == OP_PLUS && (monoid == OP_MIN || monoid == OP_MAX) && S->split_state == 1 && S->split_kind == 4) {
        dev_free(fill_img.p);
        fill_img.p = (char *)dev_alloc(type_size(st) * (size_t)u->n);
        DevBuf<int> flag(1, true);
        if (st == TC_FP32)
            hipLaunchKernelGGL((k_fill_image<float>), dim3((unsigned)ceil_div((int64_t)u->n, 256)), dim3(256), 0, ctx().stream, (const float *)a.u_val,
                               (const uint64_t *)a.u_bits, (int64_t)u->n, monoid == OP_MIN ? __builtin_huge_valf() : -__builtin_huge_valf(), (float *)fill_img.p, flag.p, (float)(fill_limit * (1.0 - 1e-6)));
        else
            hipLaunchKernelGGL((k_fill_image<double>), dim3((unsigned)ceil_div((int64_t)u->n, 256)), dim3(256), 0, ctx().stream, (const double *)a.u_val,
                               (const uint64_t *)a.u_bits, (int64_t)u->n, monoid == OP_MIN ? __builtin_huge_val() : -__builtin_huge_val(), (double *)fill_img.p, flag.p, fill_limit * (1.0 - 1e-12));
        int h_flag = 0;
        d2h(&h_flag, flag.p, sizeof(h_flag));
        ctx().stats.kernel_launches += 1;
        if (!h_flag) {
            a.u_val = fill_img.p;
            a.u_full = 1;
            a.has_by_value = 1;
            ctx().stats.fill_absent = 1;
        }
    }
    // BOOL: pack the values of the image the kernel indexes ([hot | u] or u) into bits
    DevBuf<uint64_t> valbits(0);
    // (u not full: presence and value share one word per 16 codes, one gather per entry instead of two -- every pull kernel
    //  but the chunk kernel of the long rows reads that form)
    const bool chunk_kernel = S->split_state == 1 && !((S->split_kind == 1 || S->split_kind == 2 || S->split_kind == 4) && S->long_nnz > 0);
    if (st == TC_BOOL && a.need_uval && !a.u_full && !chunk_kernel && !(ctx().debug_flags & 2048)) {
        const int64_t len = a.x_len;
        dev_free(valbits.p);
        valbits.p = (uint64_t *)dev_alloc((size_t)((len + 15) >> 4) * 4 + 8);
        pack_bool_pv((const uint64_t *)a.u_bits, (const bool *)a.u_val, len, (uint32_t *)valbits.p);
        a.u_pv = (const uint32_t *)valbits.p;
        ctx().stats.kernel_launches += 1;
    } else if (st == TC_BOOL && a.need_uval) {
        const int64_t len = a.x_len;
        dev_free(valbits.p);
        valbits.p = (uint64_t *)dev_alloc(bits_words64((uint64_t)len) * 8);
        DevBuf<uint64_t> allp(0);
        const uint64_t *pres = (const uint64_t *)a.u_bits;
        if (a.u_full) {  // no presence image in this case: every entry is present
            dev_free(allp.p);
            allp.p = (uint64_t *)dev_alloc(bits_words64((uint64_t)len) * 8);
            GRB_HIP(hipMemsetAsync(allp.p, 0xff, bits_words64((uint64_t)len) * 8, ctx().stream));
            pres = allp.p;
        }
        pack_bool_values(pres, (const bool *)a.u_val, len, valbits.p);
        a.u_valbits = (const uint32_t *)valbits.p;
        ctx().stats.kernel_launches += 1;
    }
    a.m_bits = m_bits;
    a.has_mask = mask ? 1 : 0;
    a.m_comp = f.comp ? 1 : 0;
    a.replace = f.replace ? 1 : 0;

    const bool fused = (w->type->code == st);
    vector_ensure_storage(w);
    if (fused) {
        a.accum = accum ? canonical_op(st, accum->op) : -1;
        const bool fresh = (w == u);
        void *new_val = w->d_val;
        uint64_t *new_bits = w->d_bits;
        if (fresh) {
            vector_alloc_pair(w, w->padded, false, &new_val, &new_bits);
        }
        a.w_old_val = w->d_val;
        a.w_old_bits = w->d_bits;
        a.w_new_val = new_val;
        a.w_new_bits = new_bits;
        a.fresh = fresh ? 1 : 0;
        pull_dispatch(S, st, a);
        if (fresh) {
            vector_free_pair(w->padded, w->d_val, w->d_bits);
            w->d_val = new_val;
            w->d_bits = new_bits;
        }
        if (ctx().stats.fused_epilogue < 2) ctx().stats.fused_epilogue = 1;  // (2 / 3: which fused kernel took the short rows)
    } else {
        // product into a temporary of the semiring type, then the general write rule with a typecast
        GB_Vector_opaque *t = vector_new(type_of_code(st), w->n);
        try {
            vector_ensure_storage(t);
            a.accum = -1;
            a.has_mask = 0;
            a.replace = 0;
            a.w_old_val = t->d_val;
            a.w_old_bits = t->d_bits;
            a.w_new_val = t->d_val;
            a.w_new_bits = t->d_bits;
            a.fresh = 0;
            pull_dispatch(S, st, a);
            DevBuf<char> tc((size_t)w->n * w->type->size);
            cast_array(w->type->code, tc.p, st, t->d_val, (int64_t)w->n);
            const int acc_op = accum ? canonical_op(w->type->code, accum->op) : -1;
            GRB_DISPATCH_TYPE(w->type->code, TW, {
                const int64_t nthreads = (int64_t)bits_words64(w->n) * 64;
                hipLaunchKernelGGL((k_vec_write<TW>), dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 0, ctx().stream,
                                   (int64_t)w->n, (const TW *)w->d_val, (const uint64_t *)w->d_bits, (TW *)w->d_val,
                                   w->d_bits, (const TW *)tc.p, (const uint64_t *)t->d_bits, m_bits, mask ? 1 : 0,
                                   f.comp ? 1 : 0, acc_op, f.replace ? 1 : 0, 0);
            })
            ctx().stats.kernel_launches += 1;
        } catch (...) {
            vector_free(t);
            throw;
        }
        vector_free(t);
    }
    w->nvals = -1;
    if (ctx().blocking) sync_stream();
}

static int widened_type_code(int st)
{
    switch (st) {
    case TC_BOOL: case TC_INT8: case TC_INT16: return TC_INT32;
    case TC_UINT8: case TC_UINT16: return TC_UINT32;
    default: return st;
    }
}

// w<mask> = accum(w, u (+.x) P) with P's rows indexed like u.  `flip`: multiply evaluates mult(P_kj, u_k).
// Returns false (nothing done) when the frontier's rows hold so many entries that the pull direction is cheaper: push costs
// one atomic per entry of the frontier's rows (~20 G/s measured), pull streams all of S once (~270 G entries/s) -- the level
// after a hub of a power-law graph has few vertices but a large share of the edges.
static bool push_core(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum,
                      const GB_Semiring_opaque *sr, GB_Matrix_opaque *P, GB_Vector_opaque *u, bool flip, DescFlags f)
{
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 2;
    ctx().stats.out_nvals = -1;
    const int64_t n_out = (int64_t)w->n;
    const int st = sr->type;
    const int monoid = canonical_op(st, sr->monoid);
    int mult = canonical_op(st, sr->mult);
    if (flip) mult = flip_op(mult);  // the kernel evaluates mult(u_k, P_kj)
    // operands in the semiring's type
    DevBuf<char> a_cast(0), u_cast(0);
    const void *aval = P->d_val;
    if (P->nvals && P->type->code != st) {
        const int64_t nv = P->iso ? 1 : P->nvals;
        dev_free(a_cast.p);
        a_cast.p = (char *)dev_alloc(type_size(st) * (size_t)nv);
        cast_array(st, a_cast.p, P->type->code, P->d_val, nv);
        aval = a_cast.p;
    }
    const void *uval = u->d_val;
    if (u->type->code != st) {
        dev_free(u_cast.p);
        u_cast.p = (char *)dev_alloc(type_size(st) * (size_t)u->n);
        cast_array(st, u_cast.p, u->type->code, u->d_val, (int64_t)u->n);
        uval = u_cast.p;
    }
    DevBuf<uint64_t> mbits_tmp(0);
    const uint64_t *m_bits = nullptr;
    if (mask) {
        if (f.structure && mask->d_val && mask != w) m_bits = mask->d_bits;
        else {
            dev_free(mbits_tmp.p);
            mbits_tmp.p = (uint64_t *)dev_alloc(bits_words64(mask->n) * 8);
            vector_mask_bits(mask, f.structure, mbits_tmp.p);
            m_bits = mbits_tmp.p;
        }
    }
    // frontier list, its rows' lengths, prefix sums
    uint64_t *d_idx = nullptr;
    const int64_t fcount = vector_index_list(u, &d_idx);
    DevBuf<uint64_t> idx_hold(0);
    dev_free(idx_hold.p);
    idx_hold.p = d_idx;
    int64_t work = 0;
    DevBuf<int64_t> pre(fcount + 1);
    if (fcount > 0 && P->nvals > 0) {
        hipLaunchKernelGGL(k_push_degrees, dim3((unsigned)ceil_div(fcount + 1, 256)), dim3(256), 0, ctx().stream, d_idx, fcount,
                           matrix_rowptr(P), pre.p);
        prim_exclusive_sum_i64(pre.p, pre.p, fcount + 1);
        d2h(&work, pre.p + fcount, sizeof(int64_t));
    }
    if (ctx().push_mode == 1 && work * 128 > P->nvals) return false;
    const int wt = widened_type_code(st);
    const size_t wbytes = type_size(wt);
    // dense accumulator of the product (semiring type, widened) + presence
    DevBuf<char> t_val((size_t)n_out * wbytes);
    DevBuf<uint64_t> t_bits(bits_words64((uint64_t)n_out), true);
    GRB_DISPATCH_TYPE(st, T, {
        using W = typename Widen<T>::type;
        hipLaunchKernelGGL((k_fill_w<W>), dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, ctx().stream, (W *)t_val.p, n_out,
                           monoid_identity<T, W>(monoid));
        if (work > 0) {
            const int64_t nthreads = ceil_div(work, PUSH_CHUNK);
            const int need_a = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
            const int need_u = !(mult == OP_PAIR || mult == OP_SECOND);
            hipLaunchKernelGGL((k_push<T>), dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 0, ctx().stream, d_idx, fcount,
                               pre.p, work, matrix_rowptr(P), P->d_col, (const T *)aval, P->iso ? 1 : 0, (const T *)uval,
                               monoid, mult, need_a, need_u, m_bits, mask ? 1 : 0, f.comp ? 1 : 0, (W *)t_val.p,
                               (unsigned long long *)t_bits.p);
        }
    })
    ctx().stats.flops = work;
    ctx().stats.kernel_launches += 4;
    // write rule, in place on w (w never aliases the dense accumulator); u == w is safe: u was read above
    vector_ensure_storage(w);
    DevBuf<char> tc((size_t)n_out * w->type->size);
    cast_array(w->type->code, tc.p, wt, t_val.p, n_out);
    const int acc_op = accum ? canonical_op(w->type->code, accum->op) : -1;
    GRB_DISPATCH_TYPE(w->type->code, TW, {
        const int64_t nthreads = (int64_t)bits_words64(w->n) * 64;
        hipLaunchKernelGGL((k_vec_write<TW>), dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 0, ctx().stream, (int64_t)w->n,
                           (const TW *)w->d_val, (const uint64_t *)w->d_bits, (TW *)w->d_val, w->d_bits, (const TW *)tc.p,
                           (const uint64_t *)t_bits.p, m_bits, mask ? 1 : 0, f.comp ? 1 : 0, acc_op, f.replace ? 1 : 0, 0);
    })
    w->nvals = -1;
    if (ctx().blocking) sync_stream();
    return true;
}

// choose the direction: push when u has few entries and the matrix whose rows are indexed like u is at hand
static bool want_push(GB_Vector_opaque *u, GB_Matrix_opaque *P_or_null)
{
    if (!P_or_null) return false;
    if (ctx().push_mode == 0) return false;
    if (u->nvals == 0 || P_or_null->nvals == 0) return false;  // (nothing to push: the pull entry applies the write rule alone)
    if (ctx().push_mode == 2) return true;
    const int64_t nv = vector_nvals(u);
    return nv * 64 < (int64_t)u->n;  // fewer than n/64 entries
}

// The push direction for a THIN frontier (grb_mxv_push.inc, second half): returns 1 when the product was computed, 0 when the pull
// direction should run (u has too many entries, or its rows hold too many: decided from ONE host read), -1 when this path does not
// apply (typecasts, a frontier beyond the queue) and the dense push path should be tried.
static int push_thin(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, const GB_Semiring_opaque *sr,
                     GB_Matrix_opaque *P, GB_Vector_opaque *u, bool flip, DescFlags f)
{
    const int st = sr->type;
    const int monoid = canonical_op(st, sr->monoid);
    int mult = canonical_op(st, sr->mult);
    if (flip) mult = flip_op(mult);  // the kernel evaluates mult(u_k, P_kj)
    const int need_a = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
    const int need_u = !(mult == OP_PAIR || mult == OP_SECOND);
    if ((need_a && P->type->code != st) || (need_u && u->type->code != st) || w->type->code != st) return -1;
    if (u->n > 0xffffffffull || w->n > 0x7fffffffull || !u->d_val) return -1;
    const int64_t n_in = (int64_t)u->n, n_out = (int64_t)w->n;
    if (ctx().push_mode == 1 && u->nvals >= 0 && u->nvals * 64 >= n_in) return 0;
    // (a frontier vertex has at least the mean degree: with this many of them the products pass the direction threshold below for sure --
    //  no attempt, no host read)
    if (ctx().push_mode == 1 && u->nvals >= 0 && n_in > 0 && (double)u->nvals * ((double)P->nvals / (double)n_in) * 128.0 > (double)P->nvals) return 0;
    // the operands in one order: natural, or the vertex order they already share (a square matrix's: the maps translate)
    GB_Vector_opaque *vs[3] = {u, w, mask};
    GB_Perm *ord = (u->n == w->n) ? vectors_common_order(vs, 3) : nullptr;
    if (u->n != w->n)
        for (GB_Vector_opaque *v : vs)
            if (v) vector_set_order(v, nullptr);
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 2;
    ctx().stats.out_nvals = -1;
    ctx().stats.long_kernel = -1;
    PushThin a{};
    a.u_bits = u->d_bits;
    a.n_in = n_in;
    a.n_out = n_out;
    a.in_map = ord ? ord->d_inv : nullptr;
    a.out_map = ord ? ord->d_rank : nullptr;
    a.rowptr = matrix_rowptr(P);
    a.col = P->d_col;
    a.aval = P->d_val;
    a.a_iso = P->iso ? 1 : 0;
    a.u_val = u->d_val;
    a.monoid = monoid;
    a.mult = mult;
    a.need_a = need_a;
    a.need_u = need_u;
    a.f_cap = std::max<int64_t>(n_in / 64 + 64, (int64_t)1 << 16);
    a.c_cap = a.f_cap + P->nvals / PUSH_Q + 64;
    DevBuf<uint32_t> f_list(a.f_cap);
    DevBuf<uint64_t> chunks(a.c_cap);
    // (two sets of counters live with the context: a call counts in one and its frontier kernel zeroes the other for the next call -- the
    //  host read its own set before that call was launched; no memset in front of a call)
    if (!ctx().push_counters) {
        ctx().push_counters = (unsigned long long *)dev_alloc(8 * sizeof(unsigned long long));
        GRB_HIP(hipMemsetAsync(ctx().push_counters, 0, 8 * sizeof(unsigned long long), ctx().stream));
        ctx().push_parity = 0;
    }
    a.f_list = f_list.p;
    a.chunks = chunks.p;
    a.counters = ctx().push_counters + 4 * ctx().push_parity;
    a.counters_next = ctx().push_counters + 4 * (ctx().push_parity ^ 1);
    hipLaunchKernelGGL(k_push_frontier, dim3((unsigned)ceil_div((int64_t)bits_words64(u->n), 256)), dim3(256), 0, ctx().stream, a);
    ctx().push_parity ^= 1;
    unsigned long long h[3] = {0, 0, 0};
    d2h(h, a.counters, sizeof(h));
    const int64_t fcount = (int64_t)h[0], work = (int64_t)h[1], n_chunks = (int64_t)h[2];
    u->nvals = fcount;
    ctx().stats.flops = work;
    ctx().stats.kernel_launches = 1;
    if (fcount == 0) return 0;  // (nothing to push: the pull entry applies the write rule alone)
    // (direction: a masked pull over the ordered layouts costs 0.3-0.75 ms at scale 24 whatever the frontier, the push passes ~0.3 ms per
    //  million products: profiles/r04/push_pull_grid*.jsonl -- the frontier of density 1e-2, 2.7 M products, is pulled, 1e-4 pushed)
    if (ctx().push_mode == 1 && (fcount * 64 >= n_in || work * 128 > P->nvals)) return 0;
    if (fcount > a.f_cap || n_chunks > a.c_cap) return -1;
    if (ctx().push_mode == 1 && work * 8 > n_out) return -1;  // (the dense accumulator pays from here)
    // ---- mask bits (a mask that aliases w is snapshotted) ----
    DevBuf<uint64_t> mbits_tmp(0);
    if (mask) {
        if (f.structure && mask->d_val && mask != w) a.m_bits = mask->d_bits;
        else {
            dev_free(mbits_tmp.p);
            mbits_tmp.p = (uint64_t *)dev_alloc(bits_words64(mask->n) * 8);
            vector_mask_bits(mask, f.structure, mbits_tmp.p);
            a.m_bits = mbits_tmp.p;
        }
    }
    a.has_mask = mask ? 1 : 0;
    a.m_comp = f.comp ? 1 : 0;
    if (n_chunks == 0 && !w->d_val && w != u) {  // (a frontier of isolated vertices into a vector without storage: w stays as empty as it is)
        ctx().stats.long_kernel = ctx().push_small ? -2 : -1;
        return 1;
    }
    vector_ensure_storage(w);
    const size_t wbytes = type_size(widened_type_code(st));
    DevBuf<char> t_val((size_t)n_out * wbytes);  // (never filled: only positions a product marks are read)
    const bool rule_deletes = !(accum && !f.replace);
    // a handful of vertices: one workgroup runs the three passes (k_push_small) -- when what the rule deletes from w is nothing (w empty)
    // or sits on the frontier list (w IS the frontier)
    const bool small = n_chunks <= PUSH_SMALL_CHUNKS && fcount <= 4096 && ctx().push_small &&
                       (!rule_deletes || w == u || w->nvals == 0);
    DevBuf<unsigned long long> done(bits_words64((uint64_t)n_out), !small);
    a.t_val = t_val.p;
    a.done_bits = done.p;
    a.w_val = w->d_val;
    a.w_bits = (unsigned long long *)w->d_bits;
    a.accum = accum ? canonical_op(st, accum->op) : -1;
    if (small) {
        const bool clear_frontier = rule_deletes && w == u;
        ctx().stats.long_kernel = -2;  // (bookkeeping: the one-workgroup form)
        if (n_chunks > 0 || clear_frontier) {  // (no work item and nothing to delete: w is what it was, nothing is launched)
            GRB_DISPATCH_TYPE(st, T, {
                hipLaunchKernelGGL((k_push_small<T>), dim3(1), dim3(PUSH_SMALL_BLOCK), 0, ctx().stream, a, n_chunks, fcount, clear_frontier ? 1 : 0,
                                   f.replace ? 1 : 0);
            })
            ctx().stats.kernel_launches += 1;
            w->nvals = -1;
        }
        if (ctx().blocking) sync_stream();
        return 1;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_chunks, 4), (int64_t)ctx().num_cus * 8));
    GRB_DISPATCH_TYPE(st, T, {
        if (n_chunks > 0) {
            hipLaunchKernelGGL((k_push_pass<T, 0>), dim3(grid), dim3(256), 0, ctx().stream, a, n_chunks);
            hipLaunchKernelGGL((k_push_pass<T, 1>), dim3(grid), dim3(256), 0, ctx().stream, a, n_chunks);
        }
        // what the rule deletes from w goes first (u was read above: w may alias it) ...
        if (rule_deletes)
            hipLaunchKernelGGL(k_push_words, dim3((unsigned)ceil_div((int64_t)bits_words64(w->n), 256)), dim3(256), 0, ctx().stream,
                               (unsigned long long *)w->d_bits, a.m_bits, a.has_mask, a.m_comp, a.accum, f.replace ? 1 : 0, (int64_t)bits_words64(w->n));
        // ... then the products are applied
        if (n_chunks > 0) hipLaunchKernelGGL((k_push_pass<T, 2>), dim3(grid), dim3(256), 0, ctx().stream, a, n_chunks);
    })
    ctx().stats.kernel_launches += 4;
    w->nvals = -1;
    if (ctx().blocking) sync_stream();
    return 1;
}

// the push direction walks the rows of a matrix in natural order: its operands come back to it first
static bool push_natural(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, const GB_Semiring_opaque *sr,
                         GB_Matrix_opaque *P, GB_Vector_opaque *u, bool flip, DescFlags f)
{
    vector_set_order(u, nullptr);
    vector_set_order(w, nullptr);
    if (mask) vector_set_order(mask, nullptr);
    return push_core(w, mask, accum, sr, P, u, flip, f);
}

// the push direction, thin path first; false: the pull direction runs
static bool push_any(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, const GB_Semiring_opaque *sr,
                     GB_Matrix_opaque *P, GB_Vector_opaque *u, bool flip, DescFlags f)
{
    if (!P || ctx().push_mode == 0 || u->nvals == 0 || P->nvals == 0 || !u->d_val) return false;
    const int thin = push_thin(w, mask, accum, sr, P, u, flip, f);
    if (thin >= 0) return thin == 1;
    if (!want_push(u, P)) return false;
    return push_natural(w, mask, accum, sr, P, u, flip, f);
}

// One product  w<mask> = accum(w, S (+.x) u): on the popularity-ordered twin of S with the operands kept in its vertex order when the
// matrix has (or now gets) one and the call can use it, otherwise on S with the operands in natural order.
// (`u_token`: u stands for "present everywhere, values never read" -- the row reductions -- and has no storage to convert)
static void mxv_any_order(GB_Vector_opaque *w, GB_Vector_opaque *mask, const GB_BinaryOp_opaque *accum, const GB_Semiring_opaque *sr,
                          GB_Matrix_opaque *S, GB_Vector_opaque *u, bool flip, DescFlags f, bool u_token = false)
{
    const int64_t reorders0 = ctx().reorder_count;
    GB_Vector_opaque *vs[3] = {w, u_token ? nullptr : u, mask};
    bool ordered = false;
    const bool shapes_ok = S->ncols == u->n && w->n == S->nrows && (!mask || mask->n == w->n) && !(!mask && f.comp) && w->n > 0;
    if (shapes_ok && (S->nvals == 0 || u->nvals == 0 || !u->d_val)) {
        // only the write rule runs (element-wise): any order the operands share will do
        GB_Vector_opaque *ws[2] = {w, mask};
        (void)vectors_common_order(ws, 2);
        mxv_core(w, mask, accum, sr, S, u, flip, f);
        ctx().stats.reorders = (int32_t)(ctx().reorder_count - reorders0);
        return;
    }
    // (a matrix with ranked labels, GrX_Matrix_hint_ranked: the ordered layouts without an order -- any shape, pinned vectors welcome)
    // (a row block with a column order, GrX_Matrix_shard_setup: only the operand is converted -- it must be free to: not pinned, not the output)
    const bool col_only = S->col_order_only && S->perm && !S->ranked;
    if (shapes_ok && ctx().order_mode && S->nvals >= ctx().order_min_nnz && sr->type == S->type->code && S->d_col &&
        (S->ranked || (col_only ? (!u->pinned && u != w && u != mask && !u_token) : (S->nrows == S->ncols && !w->pinned && !u->pinned && !(mask && mask->pinned))))) {
        int mult = canonical_op(sr->type, sr->mult);
        if (flip) mult = flip_op(mult);
        const bool by_rowlen = mult == OP_PAIR && u->nvals == (int64_t)u->n && !(ctx().debug_flags & 65536);
        // (the first product of a large matrix runs on its arrays as they are -- lazy layouts, section 4.1.6 -- and counts itself)
        const bool first = ctx().lazy_layout && S->nvals >= ctx().lazy_min_nnz && S->pull_calls == 0 && S->ord_state == 0;
        if (!by_rowlen && !first) {
            ensure_ordered(S);
            ordered = S->ord_state == 1;
        }
    }
    if (ordered && col_only) {
        // the operand in the column order, the output and the mask (the block's ROW space) natural
        vector_set_order(w, nullptr);
        if (mask) vector_set_order(mask, nullptr);
        vector_set_order(u, S->perm);
        mxv_core(w, mask, accum, sr, S->ord, u, flip, f);
        ctx().stats.ordered = 1;
    } else if (ordered) {
        GB_Perm *P = S->ranked ? nullptr : S->perm;  // (ranked labels: the twin is in the caller's own order -- the vectors stay natural)
        // an output that keeps nothing of its old content needs no conversion: it is emptied and takes the order
        const bool w_dead = !accum && (!mask || f.replace) && w != u && w != mask;
        if (w_dead && w->order != P && w->d_val) {
            GRB_HIP(hipMemsetAsync(w->d_bits, 0, bits_words64(w->n) * 8, ctx().stream));
            w->nvals = 0;
        }
        for (GB_Vector_opaque *v : vs)
            if (v) vector_set_order(v, P);
        mxv_core(w, mask, accum, sr, S->ord, u, flip, f);
        ctx().stats.ordered = 1;
    } else {
        for (GB_Vector_opaque *v : vs)
            if (v) vector_set_order(v, nullptr);
        mxv_core(w, mask, accum, sr, S, u, flip, f);
    }
    ctx().stats.reorders = (int32_t)(ctx().reorder_count - reorders0);
    // (a caller can see WHY a product that would have taken the ordered layouts did not -- one of its vectors is pinned to the natural
    //  order: a device view was exported from it, GrX_Vector_export_dense_device / pin_natural, e.g. the RCCL buffers of a sharded run)
    if (!ordered && shapes_ok && ctx().order_mode && S->nvals >= ctx().order_min_nnz && sr->type == S->type->code && S->d_col && !S->ranked) {
        const bool pinned = col_only ? u->pinned : (S->nrows == S->ncols && (w->pinned || u->pinned || (mask && mask->pinned)));
        if (pinned) ctx().stats.pinned_natural = 1;
    }
}

}  // namespace grb

using namespace grb;

extern "C" GrB_Info GrB_mxv(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                            const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector_any(w, "w");
    if (mask) check_vector_any(mask, "mask");
    check_matrix(A, "A");
    check_vector_any(u, "u");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    DescFlags f = flags_of(desc);
    // pull over S = A (or A' with T0); push needs the matrix whose ROWS are indexed like u: S' -- only when cached
    GB_Matrix_opaque *P = f.t0 ? A : A->tr;
    const bool dims_ok = (f.t0 ? A->nrows : A->ncols) == u->n && (f.t0 ? A->ncols : A->nrows) == w->n && (!mask || mask->n == w->n);
    if (dims_ok && (!accum || (accum->type == w->type->code && !op_is_comparison(accum->op))) && !(!mask && f.comp) && w->n > 0 && push_any(w, mask, accum, semiring, P, u, /*flip=*/true, f)) {
    } else {
        GB_Matrix_opaque *S = f.t0 ? matrix_transpose_cached(A) : A;
        mxv_any_order(w, mask, accum, semiring, S, u, /*flip=*/false, f);
    }
    GRB_CATCH(errp(w))
}

extern "C" GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                            const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector_any(w, "w");
    if (mask) check_vector_any(mask, "mask");
    check_matrix(A, "A");
    check_vector_any(u, "u");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    DescFlags f = flags_of(desc);
    // w' = u' A  <=>  w = A' u with the multiply operands swapped; desc T1 transposes A
    // push walks the rows of P = A (or A' with T1, when cached) selected by u; pull gathers over S = P'
    GB_Matrix_opaque *P = f.t1 ? A->tr : A;
    const bool dims_ok = (f.t1 ? A->ncols : A->nrows) == u->n && (f.t1 ? A->nrows : A->ncols) == w->n && (!mask || mask->n == w->n);
    if (dims_ok && (!accum || (accum->type == w->type->code && !op_is_comparison(accum->op))) && !(!mask && f.comp) && w->n > 0 && push_any(w, mask, accum, semiring, P, u, /*flip=*/false, f)) {
    } else {
        GB_Matrix_opaque *S = f.t1 ? A : matrix_transpose_cached(A);
        mxv_any_order(w, mask, accum, semiring, S, u, /*flip=*/true, f);
    }
    GRB_CATCH(errp(w))
}

// Device bytes of the layouts the pull SpMV caches with a matrix (hot-coded columns, short part, long-row strips / items,
// tile table, transpose not included), for the bench line's bookkeeping.
// The caller's labels are popularity ranks (round 5): see GB_Matrix_opaque::ranked.  Setting or clearing the hint drops the ordered twin.
extern "C" GrB_Info GrX_Matrix_hint_ranked(GrB_Matrix A, int ranked)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    if (A->ranked != (ranked != 0)) {
        if (A->ord) {
            matrix_free(A->ord);
            A->ord = nullptr;
        }
        A->ord_state = 0;
        A->ranked = ranked != 0;
        if (A->tr) {  // (the cached transpose shares the labels)
            if (A->tr->ord) {
                matrix_free(A->tr->ord);
                A->tr->ord = nullptr;
            }
            A->tr->ord_state = 0;
            A->tr->ranked = A->ranked;
        }
    }
    GRB_CATCH(errp(A))
}

// A row block of a sharded graph (round 6): every rank passes the SAME reference counts of the n columns
// (its host layer all-reduced the ranks' column histograms once, at set-up), the library ranks the columns by them -- falling count, ties
// by index: the same permutation on every rank -- and builds the popularity-ordered layouts of THIS block in that column order, rows as
// they are, any shape.  `like`: take the order of another block that was set up (the chunks of one rank share one order object, so an
// operand is converted once for all of them); `col_counts` is then ignored.  A performance hint only: results are those of the natural
// order.  An operand whose image is pinned (RCCL buffers) cannot be converted in place: such a product runs the natural-order layouts.
extern "C" GrB_Info GrX_Matrix_shard_setup(GrB_Matrix A, const uint32_t *col_counts, int on_device, const GrB_Matrix like)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    GB_Perm *P = nullptr;
    const int64_t n = (int64_t)A->ncols;
    if (like) {
        check_matrix(like, "like");
        if (!like->perm || !like->col_order_only || like->ncols != A->ncols) fail(GrB_INVALID_VALUE, "GrX_Matrix_shard_setup: `like` carries no column order of this width");
        P = like->perm;
        perm_retain(P);
    } else {
        if (!col_counts) fail(GrB_NULL_POINTER, "GrX_Matrix_shard_setup: col_counts is NULL");
        if (n <= 0 || n + (int64_t)(1 << 22) > 0x7fffffff) fail(GrB_NOT_IMPLEMENTED, "GrX_Matrix_shard_setup: column count out of range");
        DevBuf<unsigned int> cnt(n);
        if (on_device) d2d(cnt.p, col_counts, sizeof(unsigned int) * (size_t)n);
        else h2d(cnt.p, col_counts, sizeof(unsigned int) * (size_t)n);
        DevBuf<uint64_t> keys(n), keys2(n);
        DevBuf<uint32_t> ids(n), ids2(n);
        DevBuf<unsigned int> poscnt(n);
        hipLaunchKernelGGL(k_order_keys, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, (const unsigned int *)cnt.p, (const int64_t *)nullptr, n, keys.p, ids.p);
        prim_sort_pairs_u64_u32(keys.p, keys2.p, ids.p, ids2.p, n, 40);
        P = new GB_Perm();
        P->n = (uint64_t)n;
        try {
            P->d_rank = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)n);
            P->d_inv = (int32_t *)dev_alloc(sizeof(int32_t) * (size_t)n);
            DevBuf<unsigned long long> live(2, true);
            const int64_t k_hot = hot_table_size(n, A->type->size);
            hipLaunchKernelGGL(k_order_place, dim3((unsigned)ceil_div(n, 1024)), dim3(1024), 0, ctx().stream, (const uint32_t *)ids2.p, (const uint64_t *)keys2.p, n, k_hot,
                               P->d_rank, P->d_inv, poscnt.p, live.p);
            unsigned long long h_live[2] = {0, 0};
            d2h(h_live, live.p, sizeof(h_live));
            P->n_live_rows = 0;
            P->n_live_cols = (int64_t)h_live[1];
        } catch (...) {
            perm_release(P);
            throw;
        }
        sync_stream();
    }
    if (A->ord) {
        matrix_free(A->ord);
        A->ord = nullptr;
    }
    A->ord_state = 0;
    perm_release(A->perm);
    A->perm = P;
    A->col_order_only = true;
    GRB_CATCH(errp(A))
}

extern "C" GrB_Info GrX_Matrix_cache_bytes(const GrB_Matrix A, uint64_t *bytes)
{
    GRB_TRY
    require_init();
    check_matrix(A, "A");
    if (!bytes) fail(GrB_NULL_POINTER, "bytes is NULL");
    uint64_t b = 0;
    const uint64_t vs = A->type->size;
    if (A->d_tile_row) b += 8ull * (uint64_t)(A->n_tiles + 1);
    if (A->d_col_hot) b += 4ull * (uint64_t)A->nvals;
    if (A->d_hot_cols) b += 4ull * (uint64_t)A->hot_k;
    if (A->split_state == 1) {
        const GB_Matrix_opaque *S = A->short_part;
        b += 8ull * (A->nrows + 1) + (A->short_tagged_only ? 0 : 4ull * (uint64_t)S->nvals + (S->iso ? vs : vs * (uint64_t)S->nvals));
        b += bits_words64(A->nrows) * 8 + 4ull * (uint64_t)A->n_long + 4ull * bits_words64(A->nrows) + 16ull * (uint64_t)A->n_chunks;
        if (A->d_probe) b += 4ull * (uint64_t)A->probe_k * (uint64_t)A->n_long;
        if (A->rt_state == 1) b += (uint64_t)A->rt_units * RT_EPL * (6 + (A->d_rt_val ? (A->vdict_n > 0 && vs == 4 ? 1 : vs) : 0)) + 36ull * (uint64_t)A->rt_ntiles;
        if (A->tg_state == 1) b += (uint64_t)A->tg_units * TAG_EPL * (5 + (A->d_tg_val ? (A->vdict_n > 0 ? 1 : vs) : 0)) + 12ull * ((A->nrows + 63) / 64);
        else if (A->tg_state == 2) b += 12ull * ((A->nrows + 63) / 64);  // (offsets and non-empty words only: the entries are built when a call needs them)
        if (A->split_kind == 4 && (A->strip_nseg > 0 || A->ct_units > 0)) {
            const uint64_t hot_lanes = (uint64_t)A->strip_cb[A->strip_ncls + A->hub_ncls] * 64;
            const uint64_t cold = (uint64_t)A->ct_units * CT_EPL;
            b += hot_lanes * (uint64_t)A->hrec_bytes + (A->d_sslot16 ? hot_lanes * 2 + hot_lanes / 16 : hot_lanes * 4) + hot_lanes / 8 + cold * ((A->d_ct_loc ? 6 : 4) + (A->d_ct_val ? (A->ct_mode == 2 ? 1 : vs) : 0)) + 20ull * (uint64_t)A->ct_ntiles;
        } else if (A->split_kind == 2 && A->strip_nseg > 0) {
            const uint64_t padded = (uint64_t)A->strip_cb[A->strip_ncls] * STRIP_CH;
            b += padded * 4 + (A->d_lval ? padded * vs : 0) + padded / 2 + padded / STRIP_CH * 8;
        } else if (A->n_items > 0) {
            const uint64_t padded = (uint64_t)A->long_nnz + 4ull * (uint64_t)A->n_items;
            b += padded * 4 + (A->d_lval ? padded * vs : 0) + 16ull * (uint64_t)A->n_items;
        }
    }
    if (A->ord) {  // the popularity-ordered twin: its layouts, its row pointers, the two maps of the order
        uint64_t tb = 0;
        (void)GrX_Matrix_cache_bytes(A->ord, &tb);
        b += tb + 8ull * (A->nrows + 1) + 8ull * A->nrows;
    }
    *bytes = b;
    GRB_CATCH(errp(A))
}

// w<mask, replace> = accum(w, reduce of the rows of A with a monoid)  (reference Matrix.reduce_rowwise / reduce_columnwise,
// core/matrix.py:2636-2710 -> GrB_Matrix_reduce_Monoid).  A row reduction is the pull SpMV over the semiring (monoid, FIRST)
// with an operand that is present everywhere and never read: the kernels then stream A once and gather nothing.
extern "C" GrB_Info GrB_Matrix_reduce_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid monoid,
                                             const GrB_Matrix A, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector_any(w, "w");
    if (mask) check_vector_any(mask, "mask");
    check_matrix(A, "A");
    if (!monoid) fail(GrB_NULL_POINTER, "monoid is NULL");
    DescFlags f = flags_of(desc);
    GB_Matrix_opaque *S = f.t0 ? matrix_transpose_cached(A) : A;
    f.t0 = false;
    const GB_Semiring_opaque sr = {monoid->op, OP_FIRST, monoid->type, "reduce"};
    GB_Vector_opaque *ones = vector_new(type_of_code(monoid->type), S->ncols);
    // "full"; FIRST never reads its values and a full operand needs no presence lookups: token storage only
    ones->d_val = dev_alloc(16);
    ones->d_bits = (uint64_t *)dev_alloc(16);
    ones->nvals = (int64_t)S->ncols;
    try {
        mxv_any_order(w, mask, accum, &sr, S, ones, /*flip=*/false, f, /*u_token=*/true);
    } catch (...) {
        vector_free(ones);
        throw;
    }
    vector_free(ones);
    GRB_CATCH(errp(w))
}

// GrB_init: load this file's code object now instead of at the first product (see preload_code_objects)
namespace grb {
void preload_mxv() { hipFuncAttributes at; (void)hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_hot_hist)); (void)hipGetLastError(); }
}  // namespace grb
