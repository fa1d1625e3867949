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
#include <vector>

#include "grb_internal.hpp"
#include "grb_ops.hpp"

namespace grb {

constexpr int PULL_BLOCK = 256;

struct PullArgs {
    int64_t m, nnz;
    const int64_t *rowptr;
    const int32_t *col;
    const void *aval;
    int a_iso;
    const void *u_val;
    const uint32_t *u_bits;
    int u_full;
    int monoid, mult;
    int need_aval, need_uval;
    const int64_t *tile_row;
    int64_t n_tiles;
    // write rule
    const uint64_t *m_bits;
    int has_mask, m_comp;
    int accum, replace;
    const void *w_old_val;
    const uint64_t *w_old_bits;
    void *w_new_val;
    uint64_t *w_new_bits;
    int fresh;  // w_new_* are different buffers from w_old_*: every kept entry must be copied
    // seams between tiles
    void *carry_val;
    uint8_t *carry_has;
    void *first_val;
    uint8_t *first_has;  // bit0: has a partial, bit1: the tile's first row started in an earlier tile
    const uint32_t *u_valbits;  // BOOL semirings: values of the u image, bit-packed (bit = present and true)
    const uint32_t *u_pv;       // BOOL semirings, u not full: presence AND value of the image in one word per 16 codes (bits 2k, 2k+1
                                // of word c >> 4 for code c, k = c & 15): one gather per entry instead of two; replaces u_valbits
    int64_t x_len;       // entries of the u image the column codes index ([hot table | u] when a hot table is in use)
    const uint64_t *long_bits;  // rows the merge-path kernel does NOT own (handled by k_mxv_long), or nullptr
    // long-row kernel (k_mxv_long / k_mxv_long_epilogue)
    const int32_t *long_rows;
    const int32_t *chunk_slot;
    const int64_t *chunk_start;
    const int32_t *chunk_len;
    int64_t n_chunks, n_long;
    int64_t n_long_epi;  // long rows whose write rule the seams launch applies (extra workgroups behind the seam ones)
    const int32_t *long_prefix;  // per 64-row group: long rows before it (slot of a long row = prefix + rank in its word)
    // class-partitioned long rows (k_mxv_long_grp)
    const int32_t *lcol;
    const void *lval;
    const int64_t *it_start;
    const int32_t *it_len;
    const int32_t *it_slot;
    int64_t item_begin[9];
    int cls_lds_lim;           // codes below it are LDS-resident in their class's workgroups (and pre-translated in lcol)
    const int64_t *class_off;  // per call with a mask: the admitted items of class c are [class_off[c], class_off[c+1]) of a
                               // compacted copy of the item fields (it_start / it_len / it_slot then point to it); else nullptr
    int long_has_known;        // u is full: an admitted long row certainly has a product (tl_has is preset, not stored per item)
    uint32_t *long_act;        // per call: bit s = the mask admits long row s
    void *tl_val;           // per long row: product accumulator (identity-initialised)
    unsigned char *tl_has;  // per long row: any product present
    long long *dbg_times;  // GRB_DEBUG_FLAGS & 8: 10 phase timestamps per tile (thread 0)
    int dbg;             // ablation switches (GRB_DEBUG): 1 = no x gathers, 2 = no A staging loads, 4 = no epilogue
};

// rows consumed by the merge path at diagonal `diag` (row-end list vs nnz list)
__global__ void k_tile_table(const int64_t *rowptr, int64_t m, int64_t nnz, int tile, int64_t n_tiles, int64_t *tile_row)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    const int64_t total = m + nnz;
    int64_t diag = t * (int64_t)tile;
    if (diag > total) diag = total;
    int64_t lo = diag - nnz > 0 ? diag - nnz : 0, hi = diag < m ? diag : m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid + 1] <= diag - mid - 1) lo = mid + 1;
        else hi = mid;
    }
    tile_row[t] = lo;
}

// The write rule for one output row.  Returns the new presence; stores the value when present.
// `old_val` is w_old[row] (only read by the caller when old_has and (accum or fresh)).
template <typename T>
__device__ __forceinline__ bool write_rule_row(const PullArgs &a, int64_t row, bool mact, bool old_has, T old_val, bool t_has,
                                               T t_val)
{
    T *w_new = (T *)a.w_new_val;
    if (!mact) {
        const bool keep = a.replace ? false : old_has;
        if (keep && a.fresh) w_new[row] = old_val;
        return keep;
    }
    if (a.accum >= 0) {
        if (old_has && t_has) { w_new[row] = apply_binop<T>(a.accum, old_val, t_val); return true; }
        if (old_has) { if (a.fresh) w_new[row] = old_val; return true; }
        if (t_has) { w_new[row] = t_val; return true; }
        return false;
    }
    if (t_has) w_new[row] = t_val;
    return t_has;
}

#define EARLY_EXIT(level) do { if (((a.dbg >> 8) & 15) == (level)) return; } while (0)
#define PHASE_STAMP(i) do { if (a.dbg_times && threadIdx.x == 0) a.dbg_times[blockIdx.x * 10 + (i)] = clock64(); } while (0)

// ---- buffer-descriptor loads: the hardware range check returns 0 for out-of-range offsets, so gathers of
//      "no column" (index -1) and tile tails need neither a branch nor an exec-mask dance -------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int64_t bytes)
{
    const int64_t lim = bytes < 0 ? 0 : (bytes > 0xfffffff0ll ? 0xfffffff0ll : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)lim, 0x00020000);
}
// streaming variant (aux 2 = non-temporal): for data read once, so that it does not displace the x image from the caches
template <typename T>
__device__ __forceinline__ T buf_load_nt(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    if constexpr (sizeof(T) == 1) return __builtin_bit_cast(T, (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 2));
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 2));
    else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 2));
    else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 2);
        const unsigned long long u = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
        return __builtin_bit_cast(T, u);
    }
}
template <typename T>
__device__ __forceinline__ T buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    if constexpr (sizeof(T) == 1) return __builtin_bit_cast(T, (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 0));
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0));
    else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
    else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
        const unsigned long long u = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
        return __builtin_bit_cast(T, u);
    }
}

// BOOL operand whose presence and value share a word (PullArgs::u_pv): xp / xv of N codes from one gather each; a code of
// -1 (nothing to gather) is out of range and reads 0 = absent
template <int N>
__device__ __forceinline__ void bool_pv_gather(__amdgpu_buffer_rsrc_t pv_rs, const int (&cc)[N], bool (&xp)[N], bool (&xv)[N])
{
    uint32_t pw[N];
#pragma unroll
    for (int i = 0; i < N; i++) pw[i] = buf_load<uint32_t>(pv_rs, (unsigned)(cc[i] >> 4) * 4u);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int sh = (cc[i] & 15) * 2;
        xp[i] = (pw[i] >> sh) & 1u;
        xv[i] = (pw[i] >> (sh + 1)) & 1u;
    }
}

template <typename T, int MONOID_CT, int MULT_CT, int IPT>
__global__ __launch_bounds__(PULL_BLOCK) void k_mxv_pull(const PullArgs a)
{
    using W = typename Widen<T>::type;
    constexpr int TILE = PULL_BLOCK * IPT;
    static_assert(IPT % 4 == 0, "IPT must be a multiple of 4 (16-byte loads)");
    // LDS holds only per-ROW state (row-start marks, row accumulators): the tile's column indices and values
    // go straight from HBM into the registers of the thread that consumes them (IPT consecutive entries).
    __shared__ __attribute__((aligned(16))) unsigned short s_head[TILE + 8];
    // (+64: per-lane scratch slots that absorb the "nothing to emit" case of the branch-free fold)
    __shared__ W s_tval[TILE + 1 + 64];
    __shared__ unsigned char s_thas[TILE + 1 + 64];
    __shared__ unsigned int s_act[TILE / 32 + 3];
    __shared__ int s_any;
    __shared__ int s_wave_last[PULL_BLOCK / 64];

    const int monoid = MONOID_CT >= 0 ? MONOID_CT : a.monoid;
    const int mult = MULT_CT >= 0 ? MULT_CT : a.mult;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x;
    const T *aval = (const T *)a.aval;
    const bool need_aval = a.need_aval != 0, need_uval = a.need_uval != 0;
    const bool has_mask = a.has_mask != 0;
    const bool stage_vals = need_aval && !a.a_iso;
    const bool need_old = (a.accum >= 0) || a.fresh;
    PHASE_STAMP(0);

    // ---- tile coordinates -------------------------------------------------------------------------------------
    const int64_t i0 = a.tile_row[tile], i1 = a.tile_row[tile + 1];
    const int64_t total = a.m + a.nnz;
    const int64_t d0 = tile * (int64_t)TILE;
    const int64_t d1 = d0 + TILE < total ? d0 + TILE : total;
    const int64_t j0 = d0 - i0;
    const int nrows_t = (int)(i1 - i0);  // rows whose end falls inside this tile (slot nrows_t = the row still open)
    const int nnz_t = (int)((d1 - i1) - j0);
    const int base = tid * IPT;  // my IPT consecutive entries of the tile
    EARLY_EXIT(1);

    // ---- issue every HBM load of the tile now: my entries (16-byte buffer loads; the descriptor ends at the
    //      end of the arrays, out-of-range parts read 0), my row's bounds, mask word, old w of my first rows ----
    const int64_t left = a.nnz - j0;
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.col + j0, left * 4);
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(aval + (a.a_iso ? 0 : j0), stage_vals ? left * (int64_t)sizeof(T) : 0);
    const bool whole = left >= TILE + 4;  // 16-byte loads never straddle the end of the arrays
    int creg[IPT];
    T vreg[IPT];
#pragma unroll
    for (int q = 0; q < IPT / 4; q++) {
        const unsigned k = (unsigned)(base + q * 4);
        if (whole && !(a.dbg & 2)) {
            // (non-temporal streaming, aux 2, measured 2-3 % slower than default-policy loads: debug flag 64 selects it)
            const auto c4 = (a.dbg & 64) ? __builtin_amdgcn_raw_buffer_load_b128(crs, k * 4u, 0, 2)
                                         : __builtin_amdgcn_raw_buffer_load_b128(crs, k * 4u, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) creg[q * 4 + i] = (int)c4[i];
            if constexpr (sizeof(T) == 4) {
                const auto v4 = (a.dbg & 64) ? __builtin_amdgcn_raw_buffer_load_b128(vrs, k * 4u, 0, 2)
                                             : __builtin_amdgcn_raw_buffer_load_b128(vrs, k * 4u, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) vreg[q * 4 + i] = __builtin_bit_cast(T, (unsigned int)v4[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) vreg[q * 4 + i] = buf_load<T>(vrs, (k + i) * (unsigned)sizeof(T));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                creg[q * 4 + i] = (a.dbg & 2) ? (int)((k + i) & 1023) : buf_load<int>(crs, (k + i) * 4u);
                vreg[q * 4 + i] = (a.dbg & 2) ? (T)1 : buf_load<T>(vrs, (k + i) * (unsigned)sizeof(T));
            }
        }
    }
    int64_t my_rs = 0, my_re = 0;
    if (tid <= nrows_t && i0 + tid < a.m) {
        my_rs = a.rowptr[i0 + tid];
        my_re = a.rowptr[i0 + tid + 1];
    }
    const int64_t rs0_64 = (i0 < a.m ? a.rowptr[i0] : a.nnz) - j0;  // start of row i0 relative to the tile
    const int abase = (int)(i0 & 31);  // bit of row i0 inside s_act[0]
    const int64_t last_row = i1 < a.m ? i1 : a.m - 1;
    const int nw = has_mask ? (int)((last_row >> 5) - (i0 >> 5)) + 1 : 0;
    const uint32_t mword = (tid < nw) ? ((const uint32_t *)a.m_bits)[(i0 >> 5) + tid] : 0u;
    const int64_t pre_g = (i0 >> 6) + wave;
    const int64_t pre_row = (pre_g << 6) + lane;
    uint64_t pre_word = 0;
    T pre_val = (T)0;
    if ((pre_g << 6) < a.m) {
        pre_word = a.w_old_bits[pre_g];
        if (need_old && pre_row < a.m) pre_val = ((const T *)a.w_old_val)[pre_row];
    }
    const T iso_v = (a.a_iso && need_aval) ? aval[0] : (T)0;
    const __amdgpu_buffer_rsrc_t xval_rs = make_rsrc(a.u_val, a.x_len * (int64_t)sizeof(T));
    const __amdgpu_buffer_rsrc_t xbits_rs = make_rsrc(a.u_bits, a.u_full ? 0 : ((a.x_len + 63) >> 6) * 8);
    const __amdgpu_buffer_rsrc_t xvbits_rs = make_rsrc(a.u_valbits, a.u_valbits ? ((a.x_len + 63) >> 6) * 8 : 0);
    const __amdgpu_buffer_rsrc_t xpv_rs = make_rsrc(a.u_pv, a.u_pv ? ((a.x_len + 15) >> 4) * 4 : 0);
    PHASE_STAMP(1);
    EARLY_EXIT(2);

    // ---- LDS: row accumulators at the monoid identity, row-start marks cleared -----------------------------------
    for (int k = tid; k <= nrows_t; k += PULL_BLOCK) {
        s_tval[k] = monoid_identity<T, W>(monoid);
        s_thas[k] = 0;
    }
    for (int k = tid * 8; k < TILE + 8; k += PULL_BLOCK * 8) *(uint4 *)&s_head[k] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) s_any = has_mask ? 0 : 1;
    __syncthreads();
    EARLY_EXIT(3);
    // ---- active-row words for rows i0 .. min(i1, m-1); mark the entry at which each non-empty row starts ----------
    for (int k = tid; k < nw; k += PULL_BLOCK) {
        uint32_t w = (k == tid) ? mword : ((const uint32_t *)a.m_bits)[(i0 >> 5) + k];
        if (a.m_comp) w = ~w;
        s_act[k] = w;
        const int64_t base_row = ((i0 >> 5) + k) << 5;
        uint32_t in = 0xffffffffu;  // restrict to [i0, last_row] for the "anything to do" test
        if (base_row < i0) in &= 0xffffffffu << (int)(i0 - base_row);
        if (base_row + 31 > last_row) in &= 0xffffffffu >> (int)(base_row + 31 - last_row);
        if (w & in) s_any = 1;
    }
    for (int k = tid; k <= nrows_t && i0 + k < a.m; k += PULL_BLOCK) {
        const int64_t rs = (k == tid) ? my_rs : a.rowptr[i0 + k];
        const int64_t re = (k == tid) ? my_re : a.rowptr[i0 + k + 1];
        const int64_t start = rs - j0;
        if (start >= 0 && start < nnz_t && rs < re) s_head[start] = (unsigned short)(k + 1);  // local row + 1
    }
    __syncthreads();
    PHASE_STAMP(2);
    EARLY_EXIT(4);
    const bool any_active = s_any != 0;

#define ROW_ACTIVE(r) (!has_mask || ((s_act[(abase + (r)) >> 5] >> ((abase + (r)) & 31)) & 1u))

    if (any_active) {
        // ---- local row (encoded k+1) of each of my entries: row starts inside my chunk, else the last start
        //      seen by earlier lanes (wavefront max-scan by shuffles) / earlier wavefronts (LDS) -------------------
        int h[IPT];
        if constexpr (IPT == 8) {
            const uint4 v = *(const uint4 *)&s_head[base];
            h[0] = v.x & 0xffff; h[1] = v.x >> 16; h[2] = v.y & 0xffff; h[3] = v.y >> 16;
            h[4] = v.z & 0xffff; h[5] = v.z >> 16; h[6] = v.w & 0xffff; h[7] = v.w >> 16;
        } else {
#pragma unroll
            for (int i = 0; i < IPT; i++) h[i] = s_head[base + i];
        }
        int lastk = 0;
#pragma unroll
        for (int i = 0; i < IPT; i++) lastk = h[i] ? h[i] : lastk;  // marks increase along the tile: last = max
        int incl = lastk;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl = incl > t ? incl : t;
        }
        int excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 0;
        if (lane == 63) s_wave_last[wave] = incl;
        __syncthreads();
        PHASE_STAMP(3);
        EARLY_EXIT(5);
        int e = 1;  // the tile's first entries belong to row i0
        for (int x = 0; x < wave; x++) e = e > s_wave_last[x] ? e : s_wave_last[x];
        e = e > excl ? e : excl;

        // ---- classify: column to gather, or -1 (past the tile end / masked-out row: the gather reads nothing) --------
        int ek[IPT], cc[IPT];
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            e = h[i] ? h[i] : e;
            ek[i] = e;
            cc[i] = (base + i < nnz_t && ROW_ACTIVE(e - 1)) ? creg[i] : -1;
        }
        if (a.dbg & (16384 | 32768)) {  // diagnostic: fold every gather into the first 2^19 / 2^15 entries of the image
            const int gmask = (a.dbg & 16384) ? 0x7ffff : 0x7fff;
#pragma unroll
            for (int i = 0; i < IPT; i++) cc[i] = cc[i] >= 0 ? (cc[i] & gmask) : -1;
        }
        // ---- gathers: presence words, then values -- IPT independent random accesses in flight per lane --------------
        bool xp[IPT];
        T xv[IPT];
        bool pv_done = false;
        if constexpr (std::is_same<T, bool>::value) {
            if (a.u_pv && !(a.dbg & 1)) {  // presence and value from one word
                bool_pv_gather<IPT>(xpv_rs, cc, xp, xv);
                pv_done = true;
            }
        }
        if (pv_done) {
        } else if (a.u_full || (a.dbg & 1)) {
#pragma unroll
            for (int i = 0; i < IPT; i++) xp[i] = cc[i] >= 0;
        } else {
            uint32_t bw[IPT];
#pragma unroll
            for (int i = 0; i < IPT; i++) bw[i] = buf_load<uint32_t>(xbits_rs, (unsigned)(cc[i] >> 5) * 4u);
#pragma unroll
            for (int i = 0; i < IPT; i++) xp[i] = (bw[i] >> (cc[i] & 31)) & 1u;
        }
        if (pv_done) {
        } else if (need_uval && !(a.dbg & 1)) {
            if constexpr (std::is_same<T, bool>::value) {
                // BOOL values travel bit-packed (2 MiB at scale 24, its hot head L1-resident) instead of one byte each
                uint32_t vw[IPT];
#pragma unroll
                for (int i = 0; i < IPT; i++) vw[i] = buf_load<uint32_t>(xvbits_rs, xp[i] ? (unsigned)(cc[i] >> 5) * 4u : 0xfffffff8u);
#pragma unroll
                for (int i = 0; i < IPT; i++) xv[i] = (vw[i] >> (cc[i] & 31)) & 1u;
            } else {
#pragma unroll
                for (int i = 0; i < IPT; i++) xv[i] = buf_load<T>(xval_rs, xp[i] ? (unsigned)cc[i] * (unsigned)sizeof(T) : 0xfffffff8u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < IPT; i++) xv[i] = (T)(cc[i] & 7);
        }
        PHASE_STAMP(4);

        // ---- segmented fold of my chunk, straight-line: a segment ends where the next entry starts a row (or at the
        //      end of the chunk) and is emitted with ONE LDS atomic into its row's accumulator ------------------------
        T acc = (T)0;
        bool has = false;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const T av = need_aval ? (a.a_iso ? iso_v : vreg[i]) : (T)0;
            const T prod = apply_binop<T>(mult, av, xv[i]);
            const bool keep = has && (i > 0) && (h[i] == 0);
            acc = xp[i] ? (keep ? apply_binop<T>(monoid, acc, prod) : prod) : (keep ? acc : (T)0);
            has = xp[i] || keep;
            const bool seg_end = (i == IPT - 1) ? true : (h[i + 1] != 0);
            // emit without a branch: a finished segment goes to its row's accumulator, anything else to this
            // lane's scratch slot (accumulators start at the monoid identity, so one atomic is right whether or
            // not other threads share the row; "any" just stores)
            const int k = (seg_end && has) ? ek[i] - 1 : TILE + 1 + lane;
            if (monoid == OP_ANY || (a.dbg & 16)) s_tval[k] = (W)acc;
            else atomic_combine<W>(&s_tval[k], (W)acc, monoid);
            s_thas[k] = 1;
        }
    }
    __syncthreads();
    PHASE_STAMP(5);
    EARLY_EXIT(6);

    // ---- epilogue: rows this tile owns, 64 consecutive rows per wavefront -----------------------------------------
    const bool started_earlier = (i0 < a.m) && (rs0_64 < 0);
    const int own_lo = (started_earlier && nrows_t > 0) ? 1 : 0;
    const int64_t row_lo = i0 + own_lo, row_hi = i1;  // [row_lo, row_hi)
    if (row_lo < row_hi && !(a.dbg & 4)) {
        const int64_t g_first = row_lo >> 6, g_last = (row_hi - 1) >> 6;
        for (int64_t g = g_first + wave; g <= g_last; g += PULL_BLOCK / 64) {
            const int64_t row = (g << 6) + lane;
            bool owned = row >= row_lo && row < row_hi;
            if (a.long_bits) owned = owned && !((a.long_bits[g] >> lane) & 1ull);  // long rows belong to k_mxv_long
            const uint64_t oldw = (g == pre_g) ? pre_word : a.w_old_bits[g];
            const bool old_has = (oldw >> lane) & 1ull;
            bool new_has = false;
            if (owned) {
                const int k = (int)(row - i0);
                const bool mact = ROW_ACTIVE(k);
                T old_val = pre_val;
                if (g != pre_g && need_old && old_has) old_val = ((const T *)a.w_old_val)[row];
                new_has = write_rule_row<T>(a, row, mact, old_has, old_val, s_thas[k] != 0, from_acc<T, W>(s_tval[k]));
            }
            const unsigned long long nb = __ballot(owned && new_has);
            const unsigned long long om = __ballot(owned);
            if (lane == 0) {
                if (om == ~0ull) a.w_new_bits[g] = nb;
                else {
                    atomicAnd((unsigned long long *)&a.w_new_bits[g], ~om);
                    if (nb) atomicOr((unsigned long long *)&a.w_new_bits[g], nb);
                }
            }
        }
    }
#undef ROW_ACTIVE
    PHASE_STAMP(6);

    // ---- seams: the row still open at the tile end, and a first row that began in an earlier tile ------------------
    if (tid == 0) {
        a.carry_has[tile] = s_thas[nrows_t];
        ((W *)a.carry_val)[tile] = s_tval[nrows_t];
        const bool se = started_earlier && nrows_t > 0;
        a.first_has[tile] = se ? (unsigned char)(2 | (s_thas[0] ? 1 : 0)) : (unsigned char)0;
        ((W *)a.first_val)[tile] = s_tval[0];
    }
    PHASE_STAMP(7);
}

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wavefront execute in order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------
// Long rows (>= split_min_len entries; on power-law graphs 3 % of the rows hold 80 % of the entries): one wavefront
// per chunk of at most PULL_CHUNK entries of ONE row.  No row bookkeeping at all -- 8 consecutive entries per lane
// (16-byte buffer loads), 8 gathers in flight, a lane-local fold, one wavefront reduction by __shfl_down, one
// atomic into the row's accumulator.  A masked-out row returns before touching its entries.
// ---------------------------------------------------------------------------------------------------
constexpr int PULL_CHUNK = 2048;
constexpr int LONG_BLOCK = 1024;        // 16 wavefronts; one persistent workgroup per CU
constexpr int LONG_LDS_WORDS = 32768;   // 128 KiB: the head of the x image ([hot | u]: hottest columns first)

template <typename T, int MONOID_CT, int MULT_CT, int LDS_WORDS>
__global__ __launch_bounds__(LONG_BLOCK) void k_mxv_long(const PullArgs a)
{
    using W = typename Widen<T>::type;
    constexpr int EPL = 8;  // entries per lane per step
    // Every gather that misses L1 moves a 128-byte line from L2 for 4 useful bytes; the workgroup therefore keeps
    // the head of the x image in LDS for its whole life (hot-coded columns: the most referenced come first --
    // 32 Ki fp32 entries receive ~45 % of the references of an R-MAT graph; BOOL values are bit-packed: 1 Mi entries).
    __shared__ uint32_t s_x[LDS_WORDS];
    const int monoid = MONOID_CT >= 0 ? MONOID_CT : a.monoid;
    const int mult = MULT_CT >= 0 ? MULT_CT : a.mult;
    const int lane = threadIdx.x & 63;
    const bool need_aval = a.need_aval != 0, need_uval = a.need_uval != 0;
    const bool stage_vals = need_aval && !a.a_iso;
    const T *aval = (const T *)a.aval;
    const T iso_v = (a.a_iso && need_aval) ? aval[0] : (T)0;
    const __amdgpu_buffer_rsrc_t xval_rs = make_rsrc(a.u_val, a.x_len * (int64_t)sizeof(T));
    const __amdgpu_buffer_rsrc_t xbits_rs = make_rsrc(a.u_bits, a.u_full ? 0 : ((a.x_len + 63) >> 6) * 8);
    const __amdgpu_buffer_rsrc_t xvbits_rs = make_rsrc(a.u_valbits, a.u_valbits ? ((a.x_len + 63) >> 6) * 8 : 0);
    const __amdgpu_buffer_rsrc_t xpv_rs = make_rsrc(a.u_pv, a.u_pv ? ((a.x_len + 15) >> 4) * 4 : 0);
    constexpr bool IS_BOOL = std::is_same<T, bool>::value;
    // entries of the image resident in LDS
    constexpr int64_t LDS_CAP = IS_BOOL ? (int64_t)LDS_WORDS * 32 : (int64_t)LDS_WORDS * 4 / (int64_t)(sizeof(T) < 4 ? 4 : sizeof(T));
    const int lds_n = (need_uval && !(a.dbg & 4096)) ? (int)(a.x_len < LDS_CAP ? a.x_len : LDS_CAP) : 0;  // (debug flag 4096: no LDS residency)
    if (need_uval) {
        if constexpr (IS_BOOL) {
            const int words = (lds_n + 31) >> 5;
            for (int k = threadIdx.x; k < words; k += LONG_BLOCK) s_x[k] = buf_load<uint32_t>(xvbits_rs, (unsigned)k * 4u);
        } else if constexpr (sizeof(T) == 8) {
            for (int k = threadIdx.x; k < lds_n; k += LONG_BLOCK) ((T *)s_x)[k] = buf_load<T>(xval_rs, (unsigned)k * 8u);
        } else if constexpr (sizeof(T) == 4) {
            for (int k = threadIdx.x; k < lds_n; k += LONG_BLOCK) s_x[k] = __builtin_bit_cast(uint32_t, buf_load<T>(xval_rs, (unsigned)k * 4u));
        } else {  // 1- and 2-byte values: one per 32-bit LDS word
            for (int k = threadIdx.x; k < lds_n; k += LONG_BLOCK) s_x[k] = (uint32_t)buf_load<T>(xval_rs, (unsigned)k * (unsigned)sizeof(T));
        }
    }
    __syncthreads();

    const int64_t wave0 = (int64_t)blockIdx.x * (LONG_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (LONG_BLOCK / 64);
    constexpr int STEP = 64 * EPL;
    const bool nt = (a.dbg & 64) != 0;  // diagnostic: non-temporal entry loads
    // The wavefront owns chunks wave0, wave0 + nwaves, ...  Their descriptors (slot -> row -> mask bit -> start, len: a
    // chain of dependent loads) are fetched once, 64 chunks at a time with one chunk per lane, and broadcast from the
    // lanes as needed; masked-out chunks drop out of the ballot and nothing of their rows is read.
    for (int64_t first = wave0; first < a.n_chunks; first += 64 * nwaves) {
        const int64_t my_chunk = first + (int64_t)lane * nwaves;
        int m_len = 0, m_slot = 0, m_start_lo = 0, m_start_hi = 0;
        if (my_chunk < a.n_chunks) {
            m_slot = a.chunk_slot[my_chunk];
            bool act = true;
            if (a.has_mask) {
                const int64_t row = a.long_rows[m_slot];
                act = (((const uint32_t *)a.m_bits)[row >> 5] >> (row & 31)) & 1u;
                if (a.m_comp) act = !act;
            }
            if (act) {
                const int64_t st = a.chunk_start[my_chunk];
                m_start_lo = (int)(uint32_t)st;
                m_start_hi = (int)(st >> 32);
                m_len = a.chunk_len[my_chunk];
            }
        }
        unsigned long long todo = __ballot(m_len > 0);
        if (!todo) continue;
        // software pipeline over the steps (64 * EPL entries) of the active chunks: the entries of step s+1 are in
        // flight while the gathers of step s are
        int cur = __ffsll(todo) - 1;
        todo &= todo - 1;
        int base = 0;
        int c_len = __builtin_amdgcn_readfirstlane(__shfl(m_len, cur));
        int c_slot = __builtin_amdgcn_readfirstlane(__shfl(m_slot, cur));
        int64_t c_start = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(__shfl(m_start_hi, cur)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane(__shfl(m_start_lo, cur)));
        int cc_n[EPL];
        T av_n[EPL];
        {
            const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.col + c_start, (int64_t)c_len * 4);
            const __amdgpu_buffer_rsrc_t vrs = make_rsrc(aval + (a.a_iso ? 0 : c_start), stage_vals ? (int64_t)c_len * (int64_t)sizeof(T) : 0);
#pragma unroll
            for (int i = 0; i < EPL; i++) {
                const int e = i * 64 + lane;  // coalesced 4-byte loads; entries past the end of the chunk read 0 and are discarded
                const int c = nt ? buf_load_nt<int>(crs, (unsigned)e * 4u) : buf_load<int>(crs, (unsigned)e * 4u);
                cc_n[i] = (e < c_len) ? c : -1;
                av_n[i] = stage_vals ? (nt ? buf_load_nt<T>(vrs, (unsigned)e * (unsigned)sizeof(T)) : buf_load<T>(vrs, (unsigned)e * (unsigned)sizeof(T))) : iso_v;
            }
        }
        T acc = (T)0;
        bool has = false;
        while (true) {
            int cc[EPL];
            T av[EPL];
#pragma unroll
            for (int i = 0; i < EPL; i++) { cc[i] = cc_n[i]; av[i] = av_n[i]; }
            if (a.dbg & (16384 | 32768)) {  // diagnostic: fold every gather into the first 2^19 / 2^15 entries of the image
                const int gmask = (a.dbg & 16384) ? 0x7ffff : 0x7fff;
#pragma unroll
                for (int i = 0; i < EPL; i++) cc[i] = cc[i] >= 0 ? (cc[i] & gmask) : -1;
            }
            // where the next step lies
            int n_cur = cur, n_base = base + STEP, n_len = c_len, n_slot = c_slot;
            int64_t n_start = c_start;
            bool more = true;
            if (n_base >= c_len) {
                n_base = 0;
                if (todo) {
                    n_cur = __ffsll(todo) - 1;
                    todo &= todo - 1;
                    n_len = __builtin_amdgcn_readfirstlane(__shfl(m_len, n_cur));
                    n_slot = __builtin_amdgcn_readfirstlane(__shfl(m_slot, n_cur));
                    n_start = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(__shfl(m_start_hi, n_cur)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane(__shfl(m_start_lo, n_cur)));
                } else {
                    more = false;
                    n_len = 0;
                }
            }
            // gathers of this step
            bool xp[EPL];
            T xg[EPL];
            if (a.u_full) {
#pragma unroll
                for (int i = 0; i < EPL; i++) xp[i] = cc[i] >= 0;
            } else {
                uint32_t bw[EPL];
#pragma unroll
                for (int i = 0; i < EPL; i++) bw[i] = buf_load<uint32_t>(xbits_rs, (unsigned)(cc[i] >> 5) * 4u);
#pragma unroll
                for (int i = 0; i < EPL; i++) xp[i] = (bw[i] >> (cc[i] & 31)) & 1u;
            }
            uint32_t vw[EPL];
            if (need_uval && !(a.dbg & 8192)) {
                // resident entries come from LDS; the others from the image in HBM (an out-of-range offset reads nothing)
                if constexpr (IS_BOOL) {
#pragma unroll
                    for (int i = 0; i < EPL; i++) vw[i] = buf_load<uint32_t>(xvbits_rs, (xp[i] && cc[i] >= lds_n) ? (unsigned)(cc[i] >> 5) * 4u : 0xfffffff8u);
                } else {
#pragma unroll
                    for (int i = 0; i < EPL; i++) xg[i] = buf_load<T>(xval_rs, (xp[i] && cc[i] >= lds_n) ? (unsigned)cc[i] * (unsigned)sizeof(T) : 0xfffffff8u);
                }
            }
            // entries of the next step (issued behind the gathers, consumed one iteration later)
            {
                const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.col + n_start, (int64_t)n_len * 4);
                const __amdgpu_buffer_rsrc_t vrs = make_rsrc(aval + (a.a_iso ? 0 : n_start), stage_vals ? (int64_t)n_len * (int64_t)sizeof(T) : 0);
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const int e = n_base + i * 64 + lane;
                    const int c = nt ? buf_load_nt<int>(crs, (unsigned)e * 4u) : buf_load<int>(crs, (unsigned)e * 4u);
                    cc_n[i] = (e < n_len) ? c : -1;
                    av_n[i] = stage_vals ? (nt ? buf_load_nt<T>(vrs, (unsigned)e * (unsigned)sizeof(T)) : buf_load<T>(vrs, (unsigned)e * (unsigned)sizeof(T))) : iso_v;
                }
            }
            T xv[EPL];
            if (need_uval && !(a.dbg & 8192)) {
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const bool in_lds = xp[i] && cc[i] < lds_n;
                    if constexpr (IS_BOOL) {
                        const uint32_t wv = in_lds ? s_x[cc[i] >> 5] : vw[i];
                        xv[i] = (wv >> (cc[i] & 31)) & 1u;
                    } else {
                        const int li = in_lds ? cc[i] : 0;
                        T xl;
                        if constexpr (sizeof(T) == 8) xl = ((const T *)s_x)[li];
                        else if constexpr (sizeof(T) == 4) xl = __builtin_bit_cast(T, s_x[li]);
                        else xl = (T)s_x[li];
                        xv[i] = in_lds ? xl : xg[i];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPL; i++) xv[i] = (T)(need_uval ? 1 : 0);
            }
#pragma unroll
            for (int i = 0; i < EPL; i++) {
                const T prod = apply_binop<T>(mult, need_aval ? av[i] : (T)0, xv[i]);
                acc = xp[i] ? (has ? apply_binop<T>(monoid, acc, prod) : prod) : acc;
                has = has || xp[i];
            }
            if (n_cur != cur || !more) {
                // the chunk is complete: wavefront reduction (value + presence), one atomic into the row's accumulator
                int hasi = has ? 1 : 0;
                for (int off = 32; off > 0; off >>= 1) {
                    const T o = __shfl_down(acc, off);
                    const int oh = __shfl_down(hasi, off);
                    if (oh) {
                        acc = hasi ? apply_binop<T>(monoid, acc, o) : o;
                        hasi = 1;
                    }
                }
                if (lane == 0 && hasi) {
                    W *tl = (W *)a.tl_val;
                    if (monoid == OP_ANY) tl[c_slot] = (W)acc;
                    else atomic_combine<W>(&tl[c_slot], (W)acc, monoid);
                    a.tl_has[c_slot] = 1;
                }
                acc = (T)0;
                has = false;
            }
            if (!more) break;
            cur = n_cur; base = n_base; c_len = n_len; c_slot = n_slot; c_start = n_start;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Long rows, class-partitioned.  The gathers of the long rows dominate the SpMV, and a gather costs nothing when it
// hits LDS, one L2 request when it hits L2 and ~3x that when it misses.  So the entries of the long rows are stored a
// second time grouped by COLUMN CLASS = (code >> 5) & 7 (32 consecutive codes = one 128-byte line of 4-byte values),
// and a workgroup only ever works on the class of its own index: blocks are dealt round-robin to the 8 XCDs, so the
// L2 of an XCD sees one eighth of the x image (its hot part fits), and the 128 KiB of LDS a workgroup spends on the
// image head hold the 32 Ki hottest codes OF ITS CLASS -- 256 Ki distinct codes across the chip instead of 32 Ki.
// (Placement only changes speed: any workgroup computes any class correctly.)
//
// The unit of work is an ITEM: at most LONG_ITEM entries of one (class, long row), stored contiguously, starts aligned
// to 4 entries, items of a class sorted by falling length.  16 lanes take one item (four items per wavefront, of about
// the same length): 8 consecutive entries per lane and step straight into registers, gathers, a fold in registers,
// one butterfly reduction over the 16 lanes and ONE atomic into the row's accumulator.  With a mask the admitted items
// are compacted per call (k_long_compact), so masked-out rows are neither streamed nor scheduled.
// ---------------------------------------------------------------------------------------------------
constexpr int LONG_ITEM = 1024;
// codes the image head in LDS can hold per class x 8 classes: BOOL operands are bit-packed (32 values per word, or 16
// (presence, value) pairs per word when u is not full -- the limit is set by the latter), the others take one LDS slot of
// max(4, sizeof) bytes per code
constexpr int64_t long_lds_codes(int type_size_bytes, bool is_bool, int lds_words)
{
    return is_bool ? (int64_t)lds_words * 128 : ((int64_t)lds_words * 4 / (type_size_bytes < 4 ? 4 : type_size_bytes)) * 8;
}
// A column code of the class-partitioned copy as the kernel reads it: c itself when it is gathered from the image, or
// -2 - slot when it is resident in LDS (slot = ((c >> 8) << 5) | (c & 31) within its class; for BOOL the bit slot & 31
// of LDS word slot >> 5); -1 = padding.  A negative code times the value size is out of range for the buffer gather.
__device__ __forceinline__ int long_tcode(int c, int lds_lim) { return c < lds_lim ? -2 - (((c >> 8) << 5) | (c & 31)) : c; }

template <typename T, int MONOID_CT, int MULT_CT, int LDS_WORDS>
__global__ __launch_bounds__(LONG_BLOCK) void k_mxv_long_grp(const PullArgs a)
{
    using W = typename Widen<T>::type;
    constexpr int EPL = 8, GL = 16, STEP = GL * EPL, NWV = LONG_BLOCK / 64;
    __shared__ uint32_t s_x[LDS_WORDS];
    const int monoid = MONOID_CT >= 0 ? MONOID_CT : a.monoid;
    const int mult = MULT_CT >= 0 ? MULT_CT : a.mult;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 4, gl = lane & 15;
    const int cls = blockIdx.x & 7;
    const bool need_aval = a.need_aval != 0, need_uval = a.need_uval != 0;
    const bool stage_vals = need_aval && !a.a_iso;
    const T *lval = (const T *)a.lval;
    const T iso_v = (a.a_iso && need_aval) ? ((const T *)a.aval)[0] : (T)0;
    const __amdgpu_buffer_rsrc_t xval_rs = make_rsrc(a.u_val, a.x_len * (int64_t)sizeof(T));
    const __amdgpu_buffer_rsrc_t xbits_rs = make_rsrc(a.u_bits, a.u_full ? 0 : ((a.x_len + 63) >> 6) * 8);
    const __amdgpu_buffer_rsrc_t xvbits_rs = make_rsrc(a.u_valbits, a.u_valbits ? ((a.x_len + 63) >> 6) * 8 : 0);
    const __amdgpu_buffer_rsrc_t xpv_rs = make_rsrc(a.u_pv, a.u_pv ? ((a.x_len + 15) >> 4) * 4 : 0);
    constexpr bool IS_BOOL = std::is_same<T, bool>::value;
    // LDS residency of my class (the entries carry pre-translated codes, see long_tcode)
    constexpr int LDS_SLOTS = IS_BOOL ? LDS_WORDS : (int)((int64_t)LDS_WORDS * 4 / (int64_t)(sizeof(T) < 4 ? 4 : sizeof(T)));
    const bool use_lds = need_uval && a.cls_lds_lim > 0;
    if (use_lds) {
        if constexpr (IS_BOOL) {
            if (a.u_pv) {  // LDS word w = (presence, value) pairs of my class's slots 16 w .. 16 w + 15 = 16 consecutive codes
                for (int w = threadIdx.x; w < LDS_SLOTS; w += LONG_BLOCK) {
                    const unsigned s0 = (unsigned)w << 4;
                    const unsigned c0 = ((s0 >> 5) << 8) | ((unsigned)cls << 5) | (s0 & 31u);
                    s_x[w] = buf_load<uint32_t>(xpv_rs, (c0 >> 4) * 4u);
                }
            } else {
                for (int w = threadIdx.x; w < LDS_SLOTS; w += LONG_BLOCK) s_x[w] = buf_load<uint32_t>(xvbits_rs, (unsigned)((w << 3) | cls) * 4u);
            }
        } else {
            for (int k = threadIdx.x; k < LDS_SLOTS; k += LONG_BLOCK) {
                const unsigned c = ((unsigned)(k >> 5) << 8) | ((unsigned)cls << 5) | (unsigned)(k & 31);
                const T v = buf_load<T>(xval_rs, c * (unsigned)sizeof(T));  // (past the image: 0, never looked up)
                if constexpr (sizeof(T) == 8) ((T *)s_x)[k] = v;
                else if constexpr (sizeof(T) == 4) s_x[k] = __builtin_bit_cast(uint32_t, v);
                else s_x[k] = (uint32_t)v;
            }
        }
    }
    __syncthreads();

    // my class's items: [ib, ib + n_it) of the field arrays (with a mask: of their compacted copy)
    const int64_t ib = a.class_off ? a.class_off[cls] : a.item_begin[cls];
    const int64_t n_it = (a.class_off ? a.class_off[cls + 1] : a.item_begin[cls + 1]) - ib;
    const int64_t nblk = ((int64_t)gridDim.x - cls + 7) >> 3;  // workgroups of my class
    const int64_t wv = (int64_t)(blockIdx.x >> 3) * NWV + wave, nwv = nblk * NWV;
    W *tl = (W *)a.tl_val;

    // Software pipeline over the steps (4 items x 128 entries) of my quads: the item fields of quad q+1 and the entries of
    // the NEXT step (of this quad or the first of the next one) are requested before the gathers of the current step, so a
    // step waits for one memory round trip (its gathers), not three.
    int64_t f_st = 0, n_st = 0;
    int f_len = 0, f_slot = 0, n_len = 0, n_slot = 0;
#define GRP_ITEM_LOADS(Q)                                                  \
    do {                                                                   \
        const int64_t k_ = (Q) * 4 + grp;                                  \
        n_st = 0; n_len = 0; n_slot = 0;                                   \
        if (k_ < n_it) {                                                   \
            const int64_t it_ = ib + k_;                                   \
            n_st = a.it_start[it_];                                        \
            n_len = a.it_len[it_];                                         \
            n_slot = a.it_slot[it_];                                       \
        }                                                                  \
    } while (0)
    int n_c[EPL];
    T n_v[EPL];
#define GRP_ENTRY_LOADS(ST, BASE)                                                                   \
    do {                                                                                            \
        const int32_t *cp_ = a.lcol + ((a.dbg & 2) ? (int64_t)(lane * 8) : (ST) + gl * EPL + (BASE)); /* (2: diagnostic, no streaming) */ \
        _Pragma("unroll") for (int q_ = 0; q_ < EPL / 4; q_++) {                                    \
            const uint4 c4_ = *(const uint4 *)(cp_ + q_ * 4);                                       \
            n_c[q_ * 4 + 0] = (int)c4_.x; n_c[q_ * 4 + 1] = (int)c4_.y;                             \
            n_c[q_ * 4 + 2] = (int)c4_.z; n_c[q_ * 4 + 3] = (int)c4_.w;                             \
        }                                                                                           \
        if (stage_vals) {                                                                           \
            const T *vp_ = lval + ((a.dbg & 2) ? (int64_t)(lane * 8) : (ST) + gl * EPL + (BASE));   \
            if constexpr (sizeof(T) == 4) {                                                         \
                _Pragma("unroll") for (int q_ = 0; q_ < EPL / 4; q_++) {                            \
                    const uint4 v4_ = *(const uint4 *)(vp_ + q_ * 4);                               \
                    n_v[q_ * 4 + 0] = __builtin_bit_cast(T, v4_.x); n_v[q_ * 4 + 1] = __builtin_bit_cast(T, v4_.y); \
                    n_v[q_ * 4 + 2] = __builtin_bit_cast(T, v4_.z); n_v[q_ * 4 + 3] = __builtin_bit_cast(T, v4_.w); \
                }                                                                                   \
            } else {                                                                                \
                _Pragma("unroll") for (int i_ = 0; i_ < EPL; i_++) n_v[i_] = vp_[i_];               \
            }                                                                                       \
        } else {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < EPL; i_++) n_v[i_] = iso_v;                     \
        }                                                                                           \
    } while (0)
    const bool ident_fold = monoid == OP_MIN || monoid == OP_MAX || monoid == OP_LOR || monoid == OP_LAND;
    const T ident = from_acc<T, W>(monoid_identity<T, W>(monoid));
    int64_t q = wv;
    if (q * 4 >= n_it) return;
    GRP_ITEM_LOADS(q);
    f_st = n_st; f_len = n_len; f_slot = n_slot;
    if ((q + nwv) * 4 < n_it) GRP_ITEM_LOADS(q + nwv);
    GRP_ENTRY_LOADS(f_st, 0);
    for (;;) {
        const int64_t st = f_st;
        const int len = f_len, slot = f_slot;
        const bool more = (q + nwv) * 4 < n_it;
        int maxlen = len;
        {
            const int o1 = __shfl_xor(maxlen, 16);
            maxlen = maxlen > o1 ? maxlen : o1;
            const int o2 = __shfl_xor(maxlen, 32);
            maxlen = maxlen > o2 ? maxlen : o2;
            maxlen = __builtin_amdgcn_readfirstlane(maxlen);
        }
        T acc = ident_fold ? ident : (T)0;
        bool has = false;
        for (int base = 0; base < maxlen; base += STEP) {
            int cc[EPL];
            T av[EPL];
#pragma unroll
            for (int i = 0; i < EPL; i++) {
                cc[i] = (base + gl * EPL + i < len) ? n_c[i] : -1;  // (entries past my item belong to others: discarded)
                av[i] = n_v[i];
            }
            // the next step's entries travel behind this step's gathers
            if (base + STEP < maxlen) {
                GRP_ENTRY_LOADS(st, base + STEP);
            } else if (more) {
                f_st = n_st; f_len = n_len; f_slot = n_slot;  // (requested one quad ago)
                GRP_ENTRY_LOADS(f_st, 0);
                if ((q + 2 * nwv) * 4 < n_it) GRP_ITEM_LOADS(q + 2 * nwv);
            }
            if (a.dbg & (16384 | 32768)) {  // diagnostic: fold the image gathers into 2^19 / 2^14 entries
                const int gmask = (a.dbg & 16384) ? 0x7ffff : 0x3fff;
#pragma unroll
                for (int i = 0; i < EPL; i++) cc[i] = cc[i] >= 0 ? (cc[i] & gmask) : cc[i];
            }
            // cc: >= 0 a code to gather from the image, <= -2 an LDS slot, -1 nothing
            bool xp[EPL];
            T xv[EPL];
            bool pv_done = false;
            if constexpr (IS_BOOL) {
                if (a.u_pv && !(a.dbg & 8192)) {  // presence and value from one word (LDS for the resident codes)
                    uint32_t pw[EPL];
#pragma unroll
                    for (int i = 0; i < EPL; i++) pw[i] = buf_load<uint32_t>(xpv_rs, cc[i] >= 0 ? (unsigned)(cc[i] >> 4) * 4u : 0xfffffff8u);
#pragma unroll
                    for (int i = 0; i < EPL; i++) {
                        const bool in_lds = cc[i] < -1;
                        const int sl = in_lds ? -2 - cc[i] : cc[i];
                        const uint32_t wv32 = in_lds ? s_x[sl >> 4] : pw[i];
                        const int sh = (sl & 15) * 2;
                        xp[i] = (wv32 >> sh) & 1u;
                        xv[i] = (wv32 >> (sh + 1)) & 1u;
                    }
                    pv_done = true;
                }
            }
            if (pv_done) {
            } else if (a.u_full) {
#pragma unroll
                for (int i = 0; i < EPL; i++) xp[i] = cc[i] != -1;
            } else {
                // presence words come from the image for every entry; a resident code is translated back first (its class is
                // the workgroup's): c = ((slot >> 5) << 8) | (cls << 5) | (slot & 31)
                int co[EPL];
                uint32_t bw[EPL];
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const int sl = -2 - cc[i];
                    co[i] = cc[i] < -1 ? (((sl >> 5) << 8) | (cls << 5) | (sl & 31)) : cc[i];
                    bw[i] = buf_load<uint32_t>(xbits_rs, (unsigned)(co[i] >> 5) * 4u);  // (-1: out of range, reads 0)
                }
#pragma unroll
                for (int i = 0; i < EPL; i++) xp[i] = (bw[i] >> (co[i] & 31)) & 1u;
            }
            if (pv_done) {
            } else if (need_uval && !(a.dbg & 8192)) {
                if constexpr (IS_BOOL) {
                    uint32_t vw[EPL];
#pragma unroll
                    for (int i = 0; i < EPL; i++) vw[i] = buf_load<uint32_t>(xvbits_rs, xp[i] ? (unsigned)(cc[i] >> 5) * 4u : 0xfffffff8u);
#pragma unroll
                    for (int i = 0; i < EPL; i++) {
                        const bool in_lds = cc[i] < -1;
                        const int sl = in_lds ? -2 - cc[i] : 0;
                        const uint32_t wv32 = in_lds ? s_x[sl >> 5] : vw[i];
                        const int bit = in_lds ? (sl & 31) : (cc[i] & 31);
                        xv[i] = (wv32 >> bit) & 1u;
                    }
                } else {
                    T xg[EPL];
#pragma unroll
                    for (int i = 0; i < EPL; i++) xg[i] = buf_load<T>(xval_rs, (unsigned)cc[i] * (unsigned)sizeof(T));  // (negative: out of range)
#pragma unroll
                    for (int i = 0; i < EPL; i++) {
                        const bool in_lds = cc[i] < -1;
                        const int li = in_lds ? -2 - cc[i] : 0;
                        T xl;
                        if constexpr (sizeof(T) == 8) xl = ((const T *)s_x)[li];
                        else if constexpr (sizeof(T) == 4) xl = __builtin_bit_cast(T, s_x[li]);
                        else xl = (T)s_x[li];
                        xv[i] = in_lds ? xl : xg[i];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPL; i++) xv[i] = (T)(need_uval ? 1 : 0);
            }
            if (ident_fold) {
                // monoids whose identity absorbs "nothing here" exactly: no presence-dependent select in the chain
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const T prod = apply_binop<T>(mult, need_aval ? av[i] : (T)0, xv[i]);
                    acc = apply_binop<T>(monoid, acc, xp[i] ? prod : ident);
                    has = has || xp[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const T prod = apply_binop<T>(mult, need_aval ? av[i] : (T)0, xv[i]);
                    acc = xp[i] ? (has ? apply_binop<T>(monoid, acc, prod) : prod) : acc;
                    has = has || xp[i];
                }
            }
        }
        // butterfly over my 16 lanes (value + presence), one atomic per item
        int hasi = (has && !(a.dbg & 4)) ? 1 : 0;
#pragma unroll
        for (int off = GL / 2; off > 0; off >>= 1) {
            const T o = __shfl_xor(acc, off);
            const int oh = __shfl_xor(hasi, off);
            if (oh) {
                acc = hasi ? apply_binop<T>(monoid, acc, o) : o;
                hasi = 1;
            }
        }
        if (gl == 0 && hasi) {
            if (monoid == OP_ANY) tl[slot] = (W)acc;
            else atomic_combine<W>(&tl[slot], (W)acc, monoid);
            if (!a.long_has_known) a.tl_has[slot] = 1;
        }
        if (!more) break;
        q += nwv;
    }
#undef GRP_ENTRY_LOADS
#undef GRP_ITEM_LOADS
}

// per call with a mask: the (start, length, slot) of the admitted items, compacted in order (classes stay contiguous,
// lengths stay sorted): admitted items per 1024-item block -> scan -> write; class_off[c] = admitted items before class c
constexpr int COMPACT_BLOCK = 1024;
__global__ __launch_bounds__(256) void k_long_compact_count(const int32_t *it_slot, int64_t n_items, const uint32_t *long_act, int64_t *block_cnt)
{
    __shared__ int s_cnt[4];
    const int64_t b0 = (int64_t)blockIdx.x * COMPACT_BLOCK;
    int c = 0;
    for (int k = 0; k < COMPACT_BLOCK / 256; k++) {
        const int64_t i = b0 + k * 256 + threadIdx.x;
        bool act = false;
        if (i < n_items) {
            const int s = it_slot[i];
            act = (long_act[s >> 5] >> (s & 31)) & 1u;
        }
        c += __popcll(__ballot(act));
    }
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        block_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (blockIdx.x == gridDim.x - 1) block_cnt[gridDim.x] = 0;
    }
}
__global__ __launch_bounds__(256) void k_long_compact_write(const int32_t *it_slot, int64_t n_items, const uint32_t *long_act,
                                                            const int64_t *block_off, const int64_t *item_begin_dev,
                                                            const int64_t *it_start, const int32_t *it_len, int64_t *act_start,
                                                            int32_t *act_len, int32_t *act_slot, int64_t *class_off)
{
    __shared__ int s_cnt[4];
    const int64_t b0 = (int64_t)blockIdx.x * COMPACT_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t off = block_off[blockIdx.x];
    for (int k = 0; k < COMPACT_BLOCK / 256; k++) {
        const int64_t i = b0 + k * 256 + threadIdx.x;
        bool act = false;
        int s = 0;
        if (i < n_items) {
            s = it_slot[i];
            act = (long_act[s >> 5] >> (s & 31)) & 1u;
        }
        const unsigned long long m = __ballot(act);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += s_cnt[w];
        const int64_t rank = off + before + __popcll(m & ((1ull << lane) - 1ull));  // admitted items before item i
        if (act) {  // the kernel reads the fields of admitted item number `rank` without an indirection
            act_start[rank] = it_start[i];
            act_len[rank] = it_len[i];
            act_slot[rank] = s;
        }
        if (i < n_items) {
            for (int c = 0; c < 9; c++)
                if (item_begin_dev[c] == i) class_off[c] = rank;
        }
        off += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        for (int c = 0; c < 9; c++)
            if (item_begin_dev[c] >= n_items) class_off[c] = off;  // (classes that begin at the end of the list)
    }
}

// the write rule for one long row (run by the extra workgroups of the seams launch, after both kernels)
template <typename T>
__device__ __forceinline__ void long_row_write(const PullArgs &a, int64_t slot)
{
    using W = typename Widen<T>::type;
    const int64_t row = a.long_rows[slot];
    bool mact = true;
    if (a.has_mask) {
        mact = (((const uint32_t *)a.m_bits)[row >> 5] >> (row & 31)) & 1u;
        if (a.m_comp) mact = !mact;
    }
    const bool old_has = (a.w_old_bits[row >> 6] >> (row & 63)) & 1ull;
    const T old_val = old_has ? ((const T *)a.w_old_val)[row] : (T)0;
    const bool new_has = write_rule_row<T>(a, row, mact, old_has, old_val, a.tl_has[slot] != 0, from_acc<T, W>(((const W *)a.tl_val)[slot]));
    const unsigned long long bit = 1ull << (row & 63);
    if (new_has) atomicOr((unsigned long long *)&a.w_new_bits[row >> 6], bit);
    else atomicAnd((unsigned long long *)&a.w_new_bits[row >> 6], ~bit);
}

// per call: long-row accumulators at the monoid identity; bit s of long_act = the mask admits long row s
template <typename W>
__global__ void k_long_init(W *tl_val, unsigned char *tl_has, int64_t n, W identity, const int32_t *long_rows, const uint64_t *m_bits,
                            int has_mask, int m_comp, uint32_t *long_act, int has_known)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool act = false;
    if (i < n) {
        tl_val[i] = identity;
        act = true;
        if (has_mask) {
            const int64_t row = long_rows[i];
            act = (m_bits[row >> 6] >> (row & 63)) & 1ull;
            if (m_comp) act = !act;
        }
        // has_known: every entry of an admitted long row meets a present u entry, so the row has a product
        tl_has[i] = (has_known && act) ? 1 : 0;
    }
    const unsigned long long b = __ballot(act);
    const int lane = threadIdx.x & 63;
    if (long_act && lane == 0 && (i >> 6) < ((n + 63) >> 6)) {
        long_act[(i >> 6) * 2] = (uint32_t)b;
        long_act[(i >> 6) * 2 + 1] = (uint32_t)(b >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------
// Short rows of a split matrix (every row below split_min_len entries; the long rows are empty here): one WAVEFRONT per
// 64 consecutive rows = one word of the presence bitmap.  Nothing is shared between wavefronts: no workgroup barrier, no
// tile table, no seams.  The group's entries are contiguous in the CSR arrays and are consumed in windows of 64 x 8
// entries -- each lane takes 8 consecutive entries straight into registers (16-byte loads), the row of each entry
// comes from row-start marks in LDS plus a wavefront max-scan, products are folded per lane and every finished
// segment goes to its row's LDS accumulator with one LDS atomic (as in k_mxv_pull).  The epilogue applies the write
// rule to the 64 rows with coalesced loads and stores and writes the whole presence word; long rows of the group take
// their product from k_mxv_long's per-row accumulators.
// ---------------------------------------------------------------------------------------------------
constexpr int ROWS_BLOCK = 256;
constexpr int ROWS_EPL = 4;

// one group of 64 rows, by one wavefront; mark / acc / has are the wavefront's LDS scratch
template <typename T, int MONOID_CT, int MULT_CT>
__device__ __forceinline__ void rows_group(const PullArgs &a, int64_t g, int lane, unsigned char *mark,
                                           typename Widen<T>::type *acc_slots, unsigned char *has_slots)
{
    using W = typename Widen<T>::type;
    constexpr int EPL = ROWS_EPL, WIN = 64 * EPL;
    const int monoid = MONOID_CT >= 0 ? MONOID_CT : a.monoid;
    const int mult = MULT_CT >= 0 ? MULT_CT : a.mult;
    const T *aval = (const T *)a.aval;
    const bool need_aval = a.need_aval != 0, need_uval = a.need_uval != 0;
    const bool stage_vals = need_aval && !a.a_iso;
    const bool need_old = (a.accum >= 0) || a.fresh;
    const int64_t row = (g << 6) + lane;
    const bool in = row < a.m;

    // ---- everything the group needs from HBM is requested up front: row bounds, mask / presence / long-row words, old w
    const int64_t p0 = a.rowptr[in ? row : a.m], p1 = a.rowptr[in ? row + 1 : a.m];
    uint64_t actw = a.has_mask ? (a.m_comp ? ~a.m_bits[g] : a.m_bits[g]) : ~0ull;
    const uint64_t longw = a.long_bits ? a.long_bits[g] : 0ull;
    const uint64_t oldw = a.w_old_bits[g];
    const bool old_has = in && ((oldw >> lane) & 1ull);
    const T old_val = (need_old && in) ? ((const T *)a.w_old_val)[row] : (T)0;  // (not waiting for the presence word)
    const bool is_long = (longw >> lane) & 1ull;
    int long_slot = 0;
    if (longw) long_slot = a.long_prefix[g] + __popcll(longw & ((1ull << lane) - 1ull));
    bool t_has = false;
    W t_acc = monoid_identity<T, W>(monoid);
    if (is_long) {
        t_has = a.tl_has[long_slot] != 0;
        t_acc = ((const W *)a.tl_val)[long_slot];
    }
    const T iso_v = (a.a_iso && need_aval) ? aval[0] : (T)0;
    const __amdgpu_buffer_rsrc_t xval_rs = make_rsrc(a.u_val, a.x_len * (int64_t)sizeof(T));
    const __amdgpu_buffer_rsrc_t xbits_rs = make_rsrc(a.u_bits, a.u_full ? 0 : ((a.x_len + 63) >> 6) * 8);
    const __amdgpu_buffer_rsrc_t xvbits_rs = make_rsrc(a.u_valbits, a.u_valbits ? ((a.x_len + 63) >> 6) * 8 : 0);
    const __amdgpu_buffer_rsrc_t xpv_rs = make_rsrc(a.u_pv, a.u_pv ? ((a.x_len + 15) >> 4) * 4 : 0);

    const int p0_lo = __shfl((int)(uint32_t)p0, 0), p0_hi = __shfl((int)(p0 >> 32), 0);
    const int64_t gbase = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(p0_hi) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane(p0_lo));
    const int rel = (int)(p0 - gbase), len = (int)(p1 - p0);
    const int total = __builtin_amdgcn_readfirstlane(__shfl(rel + len, 63));  // entries of the group (< 64 * split_min_len)

    acc_slots[lane] = monoid_identity<T, W>(monoid);
    acc_slots[64 + lane] = monoid_identity<T, W>(monoid);
    has_slots[lane] = 0;
    has_slots[64 + lane] = 0;

    const int64_t left = a.nnz - gbase;  // entries from the group's first to the end of the arrays
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.col + gbase, left * 4);
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(aval + (a.a_iso ? 0 : gbase), stage_vals ? left * (int64_t)sizeof(T) : 0);
    const bool any_active = (actw & ~longw) != 0;  // (a group without an active short row reads none of its entries)

    for (int wbase = 0; wbase < total && any_active; wbase += WIN) {
        const int e0 = wbase + lane * EPL;  // my EPL consecutive entries of the group
        int creg[EPL];
        T vreg[EPL];
        const bool whole = left >= (int64_t)wbase + WIN + 4;  // 16-byte loads never straddle the end of the arrays
#pragma unroll
        for (int q = 0; q < EPL / 4; q++) {
            const unsigned k = (unsigned)(e0 + q * 4);
            if (whole) {
                const auto c4 = (a.dbg & 64) ? __builtin_amdgcn_raw_buffer_load_b128(crs, k * 4u, 0, 2)
                                             : __builtin_amdgcn_raw_buffer_load_b128(crs, k * 4u, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) creg[q * 4 + i] = (int)c4[i];
                if constexpr (sizeof(T) == 4) {
                    const auto v4 = (a.dbg & 64) ? __builtin_amdgcn_raw_buffer_load_b128(vrs, k * 4u, 0, 2)
                                                 : __builtin_amdgcn_raw_buffer_load_b128(vrs, k * 4u, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) vreg[q * 4 + i] = __builtin_bit_cast(T, (unsigned int)v4[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) vreg[q * 4 + i] = buf_load<T>(vrs, (k + i) * (unsigned)sizeof(T));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    creg[q * 4 + i] = buf_load<int>(crs, (k + i) * 4u);
                    vreg[q * 4 + i] = buf_load<T>(vrs, (k + i) * (unsigned)sizeof(T));
                }
            }
        }
        // ---- row of each entry: marks where rows start inside the window, max-scan across the wavefront -------------
        using MarkWord = typename std::conditional<EPL == 8, uint64_t, uint32_t>::type;  // my EPL marks in one LDS access
        *(MarkWord *)&mark[lane * EPL] = 0;
        wave_sync();
        if (len > 0 && rel >= wbase && rel < wbase + WIN) mark[rel - wbase] = (unsigned char)(lane + 1);
        const unsigned long long before = __ballot(len > 0 && rel <= wbase);  // rows begun at or before the window start
        const int carry_in = 64 - __clzll(before);                            // (1 + the last of them; never 0 inside a group)
        wave_sync();
        const uint64_t mk = *(const MarkWord *)&mark[lane * EPL];
        int h[EPL];
#pragma unroll
        for (int i = 0; i < EPL; i++) h[i] = (int)((mk >> (8 * i)) & 0xffu);
        int lastk = 0;
#pragma unroll
        for (int i = 0; i < EPL; i++) lastk = h[i] ? h[i] : lastk;  // marks increase along the window: last = max
        int incl = lastk;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl = incl > t ? incl : t;
        }
        int e = __shfl_up(incl, 1);
        if (lane == 0) e = 0;
        e = e > carry_in ? e : carry_in;
        // ---- classify: column to gather, or -1 (past the group's end / masked-out row: the gather reads nothing) -----
        int ek[EPL], cc[EPL];
#pragma unroll
        for (int i = 0; i < EPL; i++) {
            e = h[i] ? h[i] : e;
            ek[i] = e;
            cc[i] = (e0 + i < total && ((actw >> (e - 1)) & 1ull)) ? creg[i] : -1;
        }
        if (a.dbg & (16384 | 32768)) {  // diagnostic: fold every gather into the first 2^19 / 2^15 entries of the image
            const int gmask = (a.dbg & 16384) ? 0x7ffff : 0x7fff;
#pragma unroll
            for (int i = 0; i < EPL; i++) cc[i] = cc[i] >= 0 ? (cc[i] & gmask) : -1;
        }
        // ---- gathers: presence words, then values -- EPL independent random accesses in flight per lane --------------
        bool xp[EPL];
        T xv[EPL];
        bool pv_done = false;
        if constexpr (std::is_same<T, bool>::value) {
            if (a.u_pv && !(a.dbg & 1)) {  // presence and value from one word
                bool_pv_gather<EPL>(xpv_rs, cc, xp, xv);
                pv_done = true;
            }
        }
        if (pv_done) {
        } else if (a.u_full || (a.dbg & 1)) {
#pragma unroll
            for (int i = 0; i < EPL; i++) xp[i] = cc[i] >= 0;
        } else {
            uint32_t bw[EPL];
#pragma unroll
            for (int i = 0; i < EPL; i++) bw[i] = buf_load<uint32_t>(xbits_rs, (unsigned)(cc[i] >> 5) * 4u);
#pragma unroll
            for (int i = 0; i < EPL; i++) xp[i] = (bw[i] >> (cc[i] & 31)) & 1u;
        }
        if (pv_done) {
        } else if (need_uval && !(a.dbg & 1)) {
            if constexpr (std::is_same<T, bool>::value) {
                uint32_t vw[EPL];
#pragma unroll
                for (int i = 0; i < EPL; i++) vw[i] = buf_load<uint32_t>(xvbits_rs, xp[i] ? (unsigned)(cc[i] >> 5) * 4u : 0xfffffff8u);
#pragma unroll
                for (int i = 0; i < EPL; i++) xv[i] = (vw[i] >> (cc[i] & 31)) & 1u;
            } else {
#pragma unroll
                for (int i = 0; i < EPL; i++) xv[i] = buf_load<T>(xval_rs, xp[i] ? (unsigned)cc[i] * (unsigned)sizeof(T) : 0xfffffff8u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPL; i++) xv[i] = (T)(cc[i] & 7);
        }
        // ---- segmented fold of my entries, straight-line: a segment ends where the next entry starts a row (or at the
        //      end of my entries) and is emitted with ONE LDS atomic into its row's accumulator -------------------------
        T acc = (T)0;
        bool has = false;
#pragma unroll
        for (int i = 0; i < EPL; i++) {
            const T av = need_aval ? (a.a_iso ? iso_v : vreg[i]) : (T)0;
            const T prod = apply_binop<T>(mult, av, xv[i]);
            const bool keep = has && (i > 0) && (h[i] == 0);
            acc = xp[i] ? (keep ? apply_binop<T>(monoid, acc, prod) : prod) : (keep ? acc : (T)0);
            has = xp[i] || keep;
            const bool seg_end = (i == EPL - 1) ? true : (h[i + 1] != 0);
            const int k = (seg_end && has) ? ek[i] - 1 : 64 + lane;
            if (monoid == OP_ANY) acc_slots[k] = (W)acc;
            else atomic_combine<W>(&acc_slots[k], (W)acc, monoid);
            has_slots[k] = 1;
        }
    }
    wave_sync();

    // ---- write rule for my row; the wavefront owns the whole presence word ----------------------------------------------
    if (a.dbg & 4) return;
    if (!is_long) {
        t_has = has_slots[lane] != 0;
        t_acc = acc_slots[lane];
    }
    bool new_has = false;
    if (in) new_has = write_rule_row<T>(a, row, (actw >> lane) & 1ull, old_has, old_val, t_has, from_acc<T, W>(t_acc));
    const unsigned long long nb = __ballot(in && new_has);
    if (lane == 0) a.w_new_bits[g] = nb;
}


template <typename T, int MONOID_CT, int MULT_CT>
__global__ __launch_bounds__(ROWS_BLOCK) void k_mxv_rows(const PullArgs a)
{
    using W = typename Widen<T>::type;
    constexpr int WIN = 64 * ROWS_EPL, NW = ROWS_BLOCK / 64;
    __shared__ __attribute__((aligned(16))) unsigned char s_mark[NW][WIN];
    __shared__ W s_acc[NW][128];  // 64 rows + one scratch slot per lane ("nothing to emit" of the branch-free fold)
    __shared__ unsigned char s_has[NW][128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t g = (int64_t)blockIdx.x * NW + wave;
    if ((g << 6) >= a.m) return;  // wave-uniform; wavefronts never wait for each other
    rows_group<T, MONOID_CT, MULT_CT>(a, g, lane, s_mark[wave], s_acc[wave], s_has[wave]);
}

// ---- building the split (once per matrix) ---------------------------------------------------------------------
__global__ void k_split_classify(const int64_t *ptr, int64_t m, int min_len, uint64_t *long_bits, int64_t *slen, int64_t *lflag,
                                 int64_t *nchunk)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is_long = false;
    if (r < m) {
        const int64_t len = ptr[r + 1] - ptr[r];
        is_long = len >= min_len;
        slen[r] = is_long ? 0 : len;
        lflag[r] = is_long ? 1 : 0;
        nchunk[r] = is_long ? (len + PULL_CHUNK - 1) / PULL_CHUNK : 0;
    } else if (r == m) {
        slen[r] = 0; lflag[r] = 0; nchunk[r] = 0;
    }
    const unsigned long long b = __ballot(is_long);
    if ((threadIdx.x & 63) == 0 && (r >> 6) < ((m + 63) >> 6)) long_bits[r >> 6] = b;
}

template <typename T>
__global__ void k_split_fill(const int64_t *ptr, const int32_t *col, const T *val, int iso, int64_t m, int min_len,
                             const int64_t *sptr, const int64_t *lidx, const int64_t *cidx, int32_t *scol, T *sval,
                             int32_t *long_rows, int32_t *chunk_slot, int64_t *chunk_start, int32_t *chunk_len, int32_t *long_prefix)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    if ((r & 63) == 0) long_prefix[r >> 6] = (int32_t)lidx[r];
    const int64_t b = ptr[r], len = ptr[r + 1] - b;
    if (len >= min_len) {
        const int64_t slot = lidx[r];
        long_rows[slot] = (int32_t)r;
        int64_t c = cidx[r];
        for (int64_t off = 0; off < len; off += PULL_CHUNK, c++) {
            chunk_slot[c] = (int32_t)slot;
            chunk_start[c] = b + off;
            chunk_len[c] = (int32_t)(len - off < PULL_CHUNK ? len - off : PULL_CHUNK);
        }
    } else {
        const int64_t o = sptr[r];
        for (int64_t i = 0; i < len; i++) {
            scol[o + i] = col[b + i];
            if (!iso) sval[o + i] = val[b + i];
        }
    }
}

// class partition of the long rows (once per matrix): sort key = class * n_long + slot of every entry of a long row
__global__ void k_long_keys(const int64_t *ptr, const int64_t *sptr, const int32_t *long_rows, const int32_t *col, int64_t n_long,
                            uint64_t *keys, uint32_t *idx, unsigned hot_k, unsigned cold_per_class)
{
    const int64_t s = blockIdx.x;
    const int64_t row = long_rows[s];
    const int64_t b = ptr[row], len = ptr[row + 1] - b;
    const int64_t o = b - sptr[row];  // entries of long rows before this one
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) {
        const unsigned c = (unsigned)col[b + i];
        // class: codes of the hot table interleave by 128-byte line (every class gets the same heat, and its hottest codes
        // fill its workgroups' LDS); the columns behind it split into 8 contiguous ranges (one eighth of the address range
        // per XCD: fewer pages and L2 lines per class than interleaving them too)
        unsigned cls = c < hot_k ? ((c >> 5) & 7u) : (c - hot_k) / cold_per_class;
        cls = cls > 7u ? 7u : cls;
        keys[o + i] = (uint64_t)cls * (uint64_t)n_long + (uint64_t)s;
        idx[o + i] = (uint32_t)(b + i);
    }
}
template <typename T>
__global__ void k_long_permute(const uint32_t *idx, int64_t n, const int32_t *col, const T *val, int iso, int32_t *lcol, T *lval)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    lcol[i] = col[j];
    if (!iso) lval[i] = val[j];
}
// vptr[v] = first position whose key is >= v  (v = 0 .. nv; keys sorted)
__global__ void k_long_vptr(const uint64_t *keys, int64_t n, int64_t nv, int64_t *vptr)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nv) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint64_t)v) lo = mid + 1;
        else hi = mid;
    }
    vptr[v] = lo;
}
// items of virtual row v (class * n_long + slot): pieces of at most LONG_ITEM entries
__global__ void k_long_item_count(const int64_t *vptr, int64_t nv, int64_t *cnt)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nv) return;
    cnt[v] = v < nv ? (vptr[v + 1] - vptr[v] + LONG_ITEM - 1) / LONG_ITEM : 0;
}
// sort key of an item: class, then falling length
__global__ void k_long_item_fill(const int64_t *vptr, int64_t nv, int64_t n_long, const int64_t *ioff, uint64_t *key, uint32_t *id,
                                 int64_t *src, int32_t *len, int32_t *slot)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const int64_t b = vptr[v], total = vptr[v + 1] - b;
    int64_t it = ioff[v];
    for (int64_t off = 0; off < total; off += LONG_ITEM, it++) {
        const int l = (int)(total - off < LONG_ITEM ? total - off : LONG_ITEM);
        key[it] = ((uint64_t)(v / n_long) << 11) | (uint64_t)(LONG_ITEM - l);
        id[it] = (uint32_t)it;
        src[it] = b + off;
        len[it] = l;
        slot[it] = (int32_t)(v % n_long);
    }
}
// items in sorted order: fields + padded length (starts are multiples of 4 entries)
__global__ void k_long_item_order(const uint32_t *order, int64_t n_items, const int64_t *src, const int32_t *len, const int32_t *slot,
                                  int64_t *o_src, int32_t *o_len, int32_t *o_slot, int64_t *o_len4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_items) return;
    if (i == n_items) { o_len4[i] = 0; return; }
    const uint32_t j = order[i];
    o_src[i] = src[j];
    o_len[i] = len[j];
    o_slot[i] = slot[j];
    o_len4[i] = (len[j] + 3) & ~3;
}
// first sorted item of class c (c = 0 .. 8)
__global__ void k_long_class_bounds(const uint64_t *skeys, int64_t n_items, int64_t *bounds)
{
    const int c = threadIdx.x;
    if (c > 8) return;
    int64_t lo = 0, hi = n_items;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((skeys[mid] >> 11) < (uint64_t)c) lo = mid + 1;
        else hi = mid;
    }
    bounds[c] = lo;
}
// the entries of item i go to [it_start[i], it_start[i] + len4): one wavefront per item
template <typename T>
__global__ void k_long_place(const int64_t *it_start, const int64_t *it_src, const int32_t *it_len, const uint32_t *idx,
                             const int32_t *col, const T *val, int iso, int32_t *lcol, T *lval, int lds_lim)
{
    const int64_t i = blockIdx.x;
    const int64_t dst = it_start[i], src = it_src[i];
    const int len = it_len[i], len4 = (len + 3) & ~3;
    for (int e = threadIdx.x; e < len4; e += blockDim.x) {
        if (e < len) {
            const uint32_t j = idx[src + e];
            lcol[dst + e] = long_tcode(col[j], lds_lim);
            if (!iso) lval[dst + e] = val[j];
        } else {
            lcol[dst + e] = -1;
            if (!iso) lval[dst + e] = (T)0;
        }
    }
}

// One wavefront per tile whose first row began in earlier tiles: fold the carries of tiles
// [t_s, tile) with the tile's own first partial and apply the write rule for that row.
template <typename T, int TILE>
__global__ __launch_bounds__(PULL_BLOCK) void k_mxv_seams(const PullArgs a)
{
    using W = typename Widen<T>::type;
    const int lane = threadIdx.x & 63;
    const int64_t seam_blocks = (a.n_tiles + PULL_BLOCK / 64 - 1) / (PULL_BLOCK / 64);
    if ((int64_t)blockIdx.x >= seam_blocks) {  // the workgroups behind the seam ones: one thread per long row
        const int64_t slot = ((int64_t)blockIdx.x - seam_blocks) * PULL_BLOCK + threadIdx.x;
        if (slot < a.n_long_epi) long_row_write<T>(a, slot);
        return;
    }
    const int64_t tile = (int64_t)blockIdx.x * (PULL_BLOCK / 64) + (threadIdx.x >> 6);
    if (tile >= a.n_tiles) return;  // wave-uniform
    const unsigned char flag = a.first_has[tile];
    if (!(flag & 2)) return;  // wave-uniform
    const int monoid = a.monoid;
    const int64_t row = a.tile_row[tile];
    const int64_t t_s = (row + a.rowptr[row]) / TILE;
    const W *cv = (const W *)a.carry_val;
    W acc = monoid_identity<T, W>(monoid);
    int has = 0;
    for (int64_t t = t_s + lane; t < tile; t += 64) {
        if (a.carry_has[t]) {
            acc = has ? apply_binop<W>(monoid, acc, cv[t]) : cv[t];
            has = 1;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const W o = __shfl_down(acc, off);
        const int oh = __shfl_down(has, off);
        if (oh) {
            acc = has ? apply_binop<W>(monoid, acc, o) : o;
            has = 1;
        }
    }
    if (lane == 0) {
        if (flag & 1) {
            const W f = ((const W *)a.first_val)[tile];
            acc = has ? apply_binop<W>(monoid, acc, f) : f;
            has = 1;
        }
        bool mact = true;
        if (a.has_mask) {
            mact = (((const uint32_t *)a.m_bits)[row >> 5] >> (row & 31)) & 1u;
            if (a.m_comp) mact = !mact;
        }
        const bool old_has = (a.w_old_bits[row >> 6] >> (row & 63)) & 1ull;
        const T old_val = old_has ? ((const T *)a.w_old_val)[row] : (T)0;
        const bool new_has = write_rule_row<T>(a, row, mact, old_has, old_val, has != 0, from_acc<T, W>(acc));
        const unsigned long long bit = 1ull << (row & 63);
        if (new_has) atomicOr((unsigned long long *)&a.w_new_bits[row >> 6], bit);
        else atomicAnd((unsigned long long *)&a.w_new_bits[row >> 6], ~bit);
    }
}

// General (unfused) write rule: w<mask,replace> = accum(w, (TW) t), t of another type.
template <typename TW>
__global__ void k_vec_write(int64_t n, const TW *w_old_val, const uint64_t *w_old_bits, TW *w_new_val,
                            uint64_t *w_new_bits, const TW *t_val, const uint64_t *t_bits, const uint64_t *m_bits,
                            int has_mask, int m_comp, int accum, int replace, int fresh)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one wavefront = one presence word
    const int lane = threadIdx.x & 63;
    const int64_t g = row >> 6;
    const int64_t nwords = (n + 63) >> 6;
    bool new_has = false;
    if (row < n) {
        bool mact = true;
        if (has_mask) {
            mact = (m_bits[g] >> lane) & 1ull;
            if (m_comp) mact = !mact;
        }
        const bool old_has = (w_old_bits[g] >> lane) & 1ull;
        const bool t_has = (t_bits[g] >> lane) & 1ull;
        if (!mact) {
            new_has = replace ? false : old_has;
            if (new_has && fresh) w_new_val[row] = w_old_val[row];
        } else if (accum >= 0) {
            if (old_has && t_has) { w_new_val[row] = apply_binop<TW>(accum, w_old_val[row], t_val[row]); new_has = true; }
            else if (old_has) { if (fresh) w_new_val[row] = w_old_val[row]; new_has = true; }
            else if (t_has) { w_new_val[row] = t_val[row]; new_has = true; }
        } else if (t_has) { w_new_val[row] = t_val[row]; new_has = true; }
    }
    const unsigned long long nb = __ballot(new_has);
    if (lane == 0 && g < nwords) w_new_bits[g] = nb;
}

// w<m_bits (^comp), replace> = accum(w, t), in place, t of w's type (shared with the element-wise operations of grb_vecops.hip)
void vector_write_rule(GB_Vector_opaque *w, const void *t_val, const uint64_t *t_bits, const uint64_t *m_bits, bool comp, int accum,
                       bool replace)
{
    GRB_DISPATCH_TYPE(w->type->code, TW, {
        const int64_t nthreads = (int64_t)bits_words64(w->n) * 64;
        hipLaunchKernelGGL((k_vec_write<TW>), dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 0, ctx().stream, (int64_t)w->n,
                           (const TW *)w->d_val, (const uint64_t *)w->d_bits, (TW *)w->d_val, w->d_bits, (const TW *)t_val, t_bits,
                           m_bits, m_bits ? 1 : 0, comp ? 1 : 0, accum, replace ? 1 : 0, 0);
    })
}

// ---------------------------------------------------------------------------------------------------
// push direction (SpMSpV): few entries in u.  T(j) = (+)_{k in u} mult(u_k, P(k,j)) where P's ROWS are indexed
// like u (P = A for vxm, A' for mxv).  The work -- all entries of the rows selected by u -- is cut into equal
// chunks over the prefix sum of those rows' lengths (hub rows in the frontier are shared by many threads);
// products land in a dense accumulator by native global atomics, restricted to positions the mask admits.
// ---------------------------------------------------------------------------------------------------
constexpr int PUSH_CHUNK = 8;  // consecutive work items per thread

__global__ void k_push_degrees(const uint64_t *idx, int64_t f, const int64_t *rowptr, int64_t *deg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < f) {
        const int64_t k = (int64_t)idx[i];
        deg[i] = rowptr[k + 1] - rowptr[k];
    } else if (i == f) deg[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_push(const uint64_t *idx, int64_t f, const int64_t *pre /* f+1 */, int64_t work,
                                              const int64_t *rowptr, const int32_t *col, const T *aval, int a_iso,
                                              const T *u_val, int monoid, int mult, int need_a, int need_u,
                                              const uint64_t *m_bits, int has_mask, int m_comp,
                                              typename Widen<T>::type *t_val, unsigned long long *t_bits)
{
    using W = typename Widen<T>::type;
    const int64_t x0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * PUSH_CHUNK;
    if (x0 >= work) return;
    // the frontier entry owning work item x0: last i with pre[i] <= x0
    int64_t lo = 0, hi = f;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (pre[mid] <= x0) lo = mid;
        else hi = mid;
    }
    int64_t i = lo;
    int64_t k = (int64_t)idx[i];
    int64_t next = pre[i + 1];
    int64_t p = rowptr[k] + (x0 - pre[i]);
    T uk = need_u ? u_val[k] : (T)0;
    const int64_t x1 = x0 + PUSH_CHUNK < work ? x0 + PUSH_CHUNK : work;
    for (int64_t x = x0; x < x1; x++) {
        while (x >= next) {  // move to the next frontier entry (rows of length 0 are skipped)
            i++;
            k = (int64_t)idx[i];
            next = pre[i + 1];
            p = rowptr[k];
            uk = need_u ? u_val[k] : (T)0;
        }
        const int j = col[p];
        bool ok = true;
        if (has_mask) {
            ok = (m_bits[j >> 6] >> (j & 63)) & 1ull;
            if (m_comp) ok = !ok;
        }
        if (ok) {
            const T av = need_a ? aval[a_iso ? 0 : p] : (T)0;
            const W prod = (W)apply_binop<T>(mult, uk, av);
            if (monoid == OP_ANY) t_val[j] = prod;
            else atomic_combine<W>(&t_val[j], prod, monoid);
            atomicOr(&t_bits[j >> 6], 1ull << (j & 63));
        }
        p++;
    }
}

template <typename W>
__global__ void k_fill_w(W *p, int64_t n, W v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// ---- hot-column table ------------------------------------------------------------------------------------
// Power-law graphs send most gathers to a few columns (R-MAT scale 24: ~3 % of the columns receive ~80 % of
// the references) but vertex labels are scrambled, so every 128-byte line of x holds about one hot entry
// and nothing stays cached.  The K most referenced columns are therefore re-coded 0..K-1 in a cached copy
// of the column indices; per call their x entries are gathered into a K-entry table (~2 MiB: resident in
// every XCD's 4 MiB L2) and the kernel reads hot columns from the table, all others (coded K+col) from x.
__global__ void k_hot_hist(const int32_t *col, int64_t nnz, unsigned int *cnt)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) atomicAdd(&cnt[col[p]], 1u);
}
__global__ void k_hot_keys(const unsigned int *cnt, int64_t n, uint64_t *keys, uint32_t *ids)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) {
        keys[c] = (uint64_t)(0xffffffffu - cnt[c]);  // ascending sort = most referenced first
        ids[c] = (uint32_t)c;
    }
}
__global__ void k_hot_rank(const uint32_t *sorted_ids, const uint64_t *sorted_keys, int64_t k, int32_t *rank, int32_t *hot_cols,
                           unsigned long long *covered)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    if (r < k) {
        rank[sorted_ids[r]] = (int32_t)r;
        hot_cols[r] = (int32_t)sorted_ids[r];
        c = 0xffffffffull - sorted_keys[r];
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(covered, c);
}
__global__ void k_hot_recode(const int32_t *col, int64_t nnz, const int32_t *rank, int k, int32_t *col_hot)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) {
        const int c = col[p];
        const int r = rank[c];
        col_hot[p] = r >= 0 ? r : k + c;
    }
}
// per call, the image the kernels index: [ table[r] = u[hot_cols[r]], r < k | u ] -- the first workgroups gather the
// table (presence word by ballot), the others copy u's values (16 bytes per thread) and presence words behind it
template <typename T>
__global__ void k_x_image(const int32_t *hot_cols, int k, const T *u_val, const uint32_t *u_bits, int u_full, T *img_val,
                          uint64_t *img_bits, int64_t n)
{
    const int gather_blocks = (k + (int)blockDim.x - 1) / (int)blockDim.x;
    if ((int)blockIdx.x < gather_blocks) {
        const int r = blockIdx.x * blockDim.x + threadIdx.x;
        bool p = false;
        if (r < k) {
            const int c = hot_cols[r];
            p = u_full ? true : ((u_bits[c >> 5] >> (c & 31)) & 1u);
            if (p) img_val[r] = u_val[c];
        }
        const unsigned long long b = __ballot(p);
        if ((threadIdx.x & 63) == 0 && r < ((k + 63) / 64) * 64) img_bits[r >> 6] = b;
        return;
    }
    const int64_t t = ((int64_t)blockIdx.x - gather_blocks) * blockDim.x + threadIdx.x;
    const int64_t bytes = n * (int64_t)sizeof(T), n16 = bytes >> 4;
    const char *src = (const char *)u_val;
    char *dst = (char *)(img_val + k);
    if (t < n16) ((uint4 *)dst)[t] = ((const uint4 *)src)[t];
    else if (t == n16) {
        for (int64_t b = n16 << 4; b < bytes; b++) dst[b] = src[b];
    }
    if (!u_full) {
        const int64_t words = (n + 63) >> 6;
        if (t < words) img_bits[(k >> 6) + t] = ((const uint64_t *)u_bits)[t];
    }
}

static void ensure_hot(GB_Matrix_opaque *A, size_t value_bytes)
{
    if (A->hot_state != 0) return;
    A->hot_state = -1;
    const int64_t n = (int64_t)A->ncols, nnz = A->nvals;
    if (n < ctx().hot_min_cols || nnz == 0 || n + (int64_t)(1 << 22) > 0x7fffffff) return;
    int64_t k = ctx().hot_k > 0 ? ctx().hot_k : (int64_t)((2u << 20) / (value_bytes ? value_bytes : 1));
    k = std::min<int64_t>(k, n / 4);
    k &= ~(int64_t)63;
    if (k < 64) return;
    DevBuf<unsigned int> cnt(n, true);
    hipLaunchKernelGGL(k_hot_hist, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, ctx().stream, A->d_col, nnz, cnt.p);
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
    if ((double)cov < 0.25 * (double)nnz) {  // flat degree distribution: the table would not pay for itself
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

// Analyse (once) whether the rows split usefully into long and short ones and build the two parts.
// `col_src` is the column array the kernels will index (hot-coded or original).
static void ensure_split(GB_Matrix_opaque *A, const int32_t *col_src, bool hot)
{
    if (A->split_state != 0 && (A->split_state < 0 || A->split_hot == hot)) return;
    if (A->split_state == 1) {  // built against the other column coding: rebuild
        matrix_free(A->short_part);
        A->short_part = nullptr;
        dev_free(A->d_long_bits); dev_free(A->d_long_rows); dev_free(A->d_chunk_slot); dev_free(A->d_chunk_start); dev_free(A->d_chunk_len); dev_free(A->d_long_prefix);
        dev_free(A->d_lcol); dev_free(A->d_lval); dev_free(A->d_it_start); dev_free(A->d_it_len); dev_free(A->d_it_slot);
        dev_free(A->d_item_begin);
        A->d_lcol = nullptr; A->d_lval = nullptr; A->d_it_start = nullptr; A->d_it_len = nullptr; A->d_it_slot = nullptr;
        A->d_item_begin = nullptr; A->long_nnz = 0; A->n_items = 0;
        A->d_long_prefix = nullptr;
        A->d_long_bits = nullptr; A->d_long_rows = nullptr; A->d_chunk_slot = nullptr; A->d_chunk_start = nullptr; A->d_chunk_len = nullptr;
    }
    A->split_state = -1;
    const int64_t m = (int64_t)A->nrows, nnz = A->nvals;
    if (nnz < ctx().split_min_nnz || m == 0 || (ctx().debug_flags & 128)) return;
    const int min_len = ctx().split_min_len;
    DevBuf<uint64_t> lbits(bits_words64((uint64_t)m));
    DevBuf<int64_t> slen(m + 1), lflag(m + 1), nchunk(m + 1);
    hipLaunchKernelGGL(k_split_classify, dim3((unsigned)ceil_div((int64_t)bits_words64((uint64_t)m) * 64 + 1, 256)), dim3(256), 0,
                       ctx().stream, (const int64_t *)A->d_ptr, m, min_len, lbits.p, slen.p, lflag.p, nchunk.p);
    prim_exclusive_sum_i64(slen.p, slen.p, m + 1);
    prim_exclusive_sum_i64(lflag.p, lflag.p, m + 1);
    prim_exclusive_sum_i64(nchunk.p, nchunk.p, m + 1);
    int64_t nnz_short = 0, nl = 0, nc = 0;
    d2h(&nnz_short, slen.p + m, 8);
    d2h(&nl, lflag.p + m, 8);
    d2h(&nc, nchunk.p + m, 8);
    if (nl == 0 || (double)(nnz - nnz_short) < 0.3 * (double)nnz) return;  // too few entries in long rows to pay off
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
        })
        // class-partitioned copy of the long rows: sort the entries by (class, slot), cut the runs into items, sort the
        // items by (class, falling length), lay them out with 4-entry aligned starts
        const int64_t nnz_long = nnz - nnz_short;
        A->long_nnz = 0;
        A->n_items = 0;
        if (nnz_long < 0xf0000000ll && nnz < 0xffffffffll) {
            const int64_t nv = 8 * nl;
            int bits = 1;
            while (((int64_t)1 << bits) < nv) bits++;
            DevBuf<uint64_t> keys(nnz_long), keys2(nnz_long);
            DevBuf<uint32_t> idx(nnz_long), idx2(nnz_long);
            hipLaunchKernelGGL(k_long_keys, dim3((unsigned)nl), dim3(256), 0, ctx().stream, (const int64_t *)A->d_ptr,
                               (const int64_t *)S->d_ptr, (const int32_t *)A->d_long_rows, col_src, nl, keys.p, idx.p,
                               (unsigned)(hot ? A->hot_k : 0), (unsigned)std::max<int64_t>(1, ceil_div((int64_t)A->ncols, 8)));
            prim_sort_pairs_u64_u32(keys.p, keys2.p, idx.p, idx2.p, nnz_long, bits);
            DevBuf<int64_t> vptr(nv + 1), icnt(nv + 1);
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
                prim_sort_pairs_u64_u32(ikey.p, ikey2.p, iid.p, iorder.p, ni, 14);
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
                hipLaunchKernelGGL(k_long_class_bounds, dim3(1), dim3(64), 0, ctx().stream, (const uint64_t *)ikey2.p, ni, A->d_item_begin);
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
    A->split_state = 1;
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
        DevBuf<W> tl_val(a.n_long);
        DevBuf<unsigned char> tl_has(a.n_long);
        const bool by_class = ctx().long_kernel == 1 && A->long_nnz > 0;
        DevBuf<uint32_t> long_act((size_t)ceil_div(a.n_long, 64) * 2);
        a.long_act = long_act.p;
        hipLaunchKernelGGL((k_long_init<W>), dim3((unsigned)ceil_div(a.n_long, 256)), dim3(256), 0, ctx().stream, tl_val.p, tl_has.p,
                           a.n_long, monoid_identity<T, W>(a.monoid), a.long_rows, a.m_bits, a.has_mask, a.m_comp, long_act.p,
                           (by_class && a.u_full) ? 1 : 0);
        a.long_has_known = (by_class && a.u_full) ? 1 : 0;
        a.tl_val = tl_val.p;
        a.tl_has = tl_has.p;
        a.dbg = ctx().debug_flags;
        const int64_t ncb = ceil_div(A->n_items, (int64_t)COMPACT_BLOCK);
        const bool compact = by_class && a.has_mask;
        DevBuf<int64_t> act_start(compact ? (size_t)A->n_items : 1), block_cnt(compact ? (size_t)ncb + 1 : 1), class_off(9);
        DevBuf<int32_t> act_len(compact ? (size_t)A->n_items : 1), act_slot(compact ? (size_t)A->n_items : 1);
        if (by_class) {
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
        if (ctx().short_kernel == 1 && S->nrows == A->nrows) {
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
    if (accum && accum->type != w->type->code) fail(GrB_DOMAIN_MISMATCH, "mxv/vxm: accum operator type must equal the output type");
    ctx().stats = GrX_Stats{};
    ctx().stats.method = 1;
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

    // ---- operands in the semiring's type --------------------------------------------------------------
    DevBuf<char> a_cast(0), u_cast(0);
    const void *aval = S->d_val;
    if (S->nvals && S->type->code != st) {
        const int64_t nv = S->iso ? 1 : S->nvals;
        dev_free(a_cast.p);
        a_cast.p = (char *)dev_alloc(type_size(st) * (size_t)nv);
        cast_array(st, a_cast.p, S->type->code, S->d_val, nv);
        aval = a_cast.p;
    }
    vector_ensure_storage(u);
    const void *uval = u->d_val;
    if (u->type->code != st) {
        dev_free(u_cast.p);
        u_cast.p = (char *)dev_alloc(type_size(st) * (size_t)u->n);
        cast_array(st, u_cast.p, u->type->code, u->d_val, (int64_t)u->n);
        uval = u_cast.p;
    }

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

    PullArgs a{};
    a.m = m;
    a.nnz = S->nvals;
    a.rowptr = matrix_rowptr(S);
    a.col = S->d_col;
    a.aval = aval;
    a.a_iso = S->iso ? 1 : 0;
    a.u_val = uval;
    a.u_bits = (const uint32_t *)u->d_bits;
    a.u_full = (u->nvals == (int64_t)u->n) ? 1 : 0;
    a.monoid = monoid;
    a.mult = mult;
    a.need_aval = !(mult == OP_PAIR || mult == OP_SECOND) && S->nvals > 0;
    a.need_uval = !(mult == OP_PAIR || mult == OP_FIRST || mult == OP_ANY);
    if (mult == OP_ANY) a.need_aval = S->nvals > 0;
    a.x_len = (int64_t)u->n;
    if ((uint64_t)u->n * type_size(st) >= 0xff000000ull)
        fail(GrB_NOT_IMPLEMENTED, "mxv/vxm: input vectors of 4 GiB or more are not supported by the pull kernel yet");
    // hot-column table (wide matrices with a skewed column-degree distribution): the kernel indexes ONE image
    // [ K hot entries | the n entries of u ] with the re-coded column indices (hot rank, or K + col)
    DevBuf<char> xcat_val(0);
    DevBuf<uint64_t> xcat_bits(0);
    if (a.need_uval || !a.u_full) {
        ensure_hot(S, type_size(st));
        if (S->hot_state == 1 && (uint64_t)(u->n + S->hot_k) * type_size(st) < 0xff000000ull) {
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
    if (S->nvals && S->type->code == st) {
        const bool hot = (a.col == S->d_col_hot);
        ensure_split(S, a.col, hot);
        if (S->split_state == 1 && S->split_hot == hot) {
            a.long_rows = S->d_long_rows;
            a.chunk_slot = S->d_chunk_slot;
            a.chunk_start = S->d_chunk_start;
            a.chunk_len = S->d_chunk_len;
            a.n_chunks = S->n_chunks;
            a.n_long = S->n_long;
        }
    }
    // BOOL: pack the values of the image the kernel indexes ([hot | u] or u) into bits
    DevBuf<uint64_t> valbits(0);
    // (u not full: presence and value share one word per 16 codes, one gather per entry instead of two -- every pull kernel
    //  but the chunk kernel of the long rows reads that form)
    const bool chunk_kernel = S->split_state == 1 && !(ctx().long_kernel == 1 && S->long_nnz > 0);
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
        ctx().stats.fused_epilogue = 1;
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
    if (ctx().push_mode == 1 && work * 12 > P->nvals) return false;
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
    if (ctx().push_mode == 2) return true;
    const int64_t nv = vector_nvals(u);
    return nv * 64 < (int64_t)u->n;  // fewer than n/64 entries
}

}  // namespace grb

using namespace grb;

extern "C" GrB_Info GrB_mxv(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                            const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector(w, "w");
    if (mask) check_vector(mask, "mask");
    check_matrix(A, "A");
    check_vector(u, "u");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    DescFlags f = flags_of(desc);
    // pull over S = A (or A' with T0); push needs the matrix whose ROWS are indexed like u: S' -- only when cached
    GB_Matrix_opaque *P = f.t0 ? A : A->tr;
    const bool dims_ok = (f.t0 ? A->nrows : A->ncols) == u->n && (f.t0 ? A->ncols : A->nrows) == w->n && (!mask || mask->n == w->n);
    if (dims_ok && (!accum || accum->type == w->type->code) && !(!mask && f.comp) && w->n > 0 && want_push(u, P) &&
        push_core(w, mask, accum, semiring, P, u, /*flip=*/true, f)) {
    } else {
        GB_Matrix_opaque *S = f.t0 ? matrix_transpose_cached(A) : A;
        mxv_core(w, mask, accum, semiring, S, u, /*flip=*/false, f);
    }
    GRB_CATCH(errp(w))
}

extern "C" GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                            const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc)
{
    GRB_TRY
    require_init();
    check_vector(w, "w");
    if (mask) check_vector(mask, "mask");
    check_matrix(A, "A");
    check_vector(u, "u");
    if (!semiring) fail(GrB_NULL_POINTER, "semiring is NULL");
    DescFlags f = flags_of(desc);
    // w' = u' A  <=>  w = A' u with the multiply operands swapped; desc T1 transposes A
    // push walks the rows of P = A (or A' with T1, when cached) selected by u; pull gathers over S = P'
    GB_Matrix_opaque *P = f.t1 ? A->tr : A;
    const bool dims_ok = (f.t1 ? A->ncols : A->nrows) == u->n && (f.t1 ? A->nrows : A->ncols) == w->n && (!mask || mask->n == w->n);
    if (dims_ok && (!accum || accum->type == w->type->code) && !(!mask && f.comp) && w->n > 0 && want_push(u, P) &&
        push_core(w, mask, accum, semiring, P, u, /*flip=*/false, f)) {
    } else {
        GB_Matrix_opaque *S = f.t1 ? A : matrix_transpose_cached(A);
        mxv_core(w, mask, accum, semiring, S, u, /*flip=*/true, f);
    }
    GRB_CATCH(errp(w))
}
