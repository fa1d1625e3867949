// grb_internal.hpp -- object model, error plumbing and device-memory helpers of libgrb_mi355x.so.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   Matrix  : CSR, int64 row pointers, int32 column indices sorted within a row, values of the
//             matrix type (ONE value when iso); an optional cached transpose (same layout) and a
//             cached merge-path tile table for the pull SpMV.
//   Vector  : dense-with-presence ("bitmap"): n values + bit-packed presence (64-bit words; bit i of
//             word i>>6, which is also bit i&31 of 32-bit word i>>5).  nvals is cached, -1 = unknown.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>

#include "grb_mi355x.h"

namespace grb {

enum TypeCode : int {
    TC_BOOL = 0, TC_INT8, TC_INT16, TC_INT32, TC_INT64, TC_UINT8, TC_UINT16, TC_UINT32, TC_UINT64, TC_FP32, TC_FP64,
    TC_COUNT
};

// binary operators and monoids share one code space
enum OpCode : int {
    OP_NONE = -1,
    OP_FIRST = 0, OP_SECOND, OP_PAIR, OP_PLUS, OP_MINUS, OP_RMINUS, OP_TIMES, OP_MIN, OP_MAX,
    OP_LOR, OP_LAND, OP_LXOR, OP_LXNOR, OP_ANY, OP_COUNT,
    // comparisons: T x T -> BOOL; accepted by the element-wise vector operations only (grb_vecops.hip)
    OP_EQ = 32, OP_NE, OP_GT, OP_LT, OP_GE, OP_LE,
    // builtin operators that exist as handles (import-time surface, grb_surface.hip) but that no kernel implements: every entry
    // point rejects them with GrB_NOT_IMPLEMENTED (canonical_op)
    OP_UNSUPPORTED = 1000
};
inline bool op_is_comparison(int op) { return op >= OP_EQ && op <= OP_LE; }

constexpr uint64_t MAGIC_VECTOR = 0x4752425645435452ULL;  // "GRBVECTR"
constexpr uint64_t MAGIC_MATRIX = 0x4752424d41545258ULL;  // "GRBMATRX"
constexpr uint64_t MAGIC_FREED = 0xdeadbeefdeadbeefULL;

struct Error {
    GrB_Info info;
    std::string msg;
};

[[noreturn]] inline void fail(GrB_Info info, const std::string &msg) { throw Error{info, msg}; }

#define GRB_HIP(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            ::grb::fail(_e == hipErrorOutOfMemory ? GrB_OUT_OF_MEMORY : GrB_PANIC,                          \
                        std::string("HIP error '") + hipGetErrorString(_e) + "' at " __FILE__ ":" +        \
                            std::to_string(__LINE__) + " in " #expr);                                       \
        }                                                                                                   \
    } while (0)

struct Context {
    bool initialized = false;
    bool blocking = false;
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;  // null stream: ordered with torch's default stream
    void (*host_free)(void *) = nullptr;  // GxB_init: the deallocator of host arrays whose ownership an import / pack takes (nullptr: free)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // round 6: the kernels of one pull product that do not depend on each other run on two streams (mxv_overlap): the cold tiles gather
    // (TA-bound) on `aux_stream` while the hot strips stream (HBM-bound) on `stream`; ev_fork / ev_join order them with the rest of the call
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int mxv_overlap = 0;             // 1: k_mxv_ctile on the auxiliary stream next to k_mxv_hstrip; 0: one stream.  Measured (profiles/r06/overlap.txt): 0.477 vs 0.479 ms -- the two kernels stretch to the same total, the call is bound by the lines it moves; off
    int strip_wgs = 0;               // persistent workgroups of the merged k_mxv_hstrip launch (0 = one per CU); read when the layouts are built
    int64_t vec_pad_min_bytes = 1 << 20;  // vectors with at least this many bytes of values are allocated with the front pad
    int alloc_cache = 1;    // 1: freed device blocks are kept per size class and reused without calling the HIP allocator
    int drop_hot_cols = 1;     // release the re-coded copy of the whole column array once the long / short split is built from it
    int mxm_heavy_kernel = 1;  // SpGEMM rows beyond the LDS hash: 1 = a wavefront per row over column windows (k_spgemm_wave), 0 = the
                            // 1024-thread kernels of round 1 (k_spgemm_sym_lds / k_spgemm_win)
    int mxm_unit_small = 512, mxm_unit_mid = 1024, mxm_unit_dense = 4096;  // entry counts of a unit up to which one wavefront / four wavefronts with compact
                                                      // accumulators take it; denser units get an accumulator per column
    int64_t mxm_unit_min_flops = 1024;
    int64_t mxm_unit_min_per_window = 16;  // ... and than this many per column window
    int64_t mxm_masked_units_min_flops = 64ll << 20;  // mask-driven products below this many multiplies keep the row kernels
    int64_t mxm_bitmap_pool_cap = INT32_MAX;  // ... and at most this many bitmaps (tests: a pool that runs out)
    int64_t mxm_window_groups = 0;  // SpGEMM units walk groups of this many 16 Ki-column windows (1 / 2 / 4 / 8); 0 = from the width of B: at most 64 groups per row
    int mxm_xcd_map = 1;  // SpGEMM unit kernels: 1 = every XCD takes a contiguous eighth of the unit order (the windows of a row share one L2), 0 = round-robin
    int mxm_checksum_pass = 0;  // GrX_mxm_streamed: 0 = the checksum is folded into the numeric kernels' stores, 1 = a pass of its own over the product (round 4)
    int64_t mxm_sym_windows = 8;  // consecutive windows of a row a symbolic unit walks (rows of up to 128 entries of A; 1: one window per unit)
    int mxm_bitmap_min_cnt = 512;  // units with more entries than this keep their bitmap
    int64_t mxm_bitmap_pool_mb = 16384;  // bitmaps of the denser units kept from the symbolic for the numeric pass: at most this much  // rows with more products than this (and than 32 per window) are walked as units
    int mat_write_kernel = 1;  // the write rule of matrix results: 1 a wavefront per row / column piece, 0 a thread per row (rounds 1-2)
    int mxm_mask_mode = 1;  // mask-driven SpGEMM for non-complemented masks: 0 never, 1 when the full product costs more, 2 always;
                            // complemented masks are fused into the product unless 0
    int long_sub = 0;       // sub-ranges per class of the cold columns of the long rows (items of a class are walked sub-range by
                            // sub-range); 0 = sized from the operand image (~2 MiB per sub-range)
    int long_sub_min_len = 0;  // ... for rows with at least this many entries (0 = 512 per sub-range)
    int long_kernel = 5;    // long rows: 5 = by matrix type and size (items for BOOL; hot / cold strips from lean_min_nnz entries, mixed class
                            // strips below), 4 = hot strips + cold tiles (k_mxv_hstrip / k_mxv_ctile, generic: k_mxv_strip), 3 = by type with
                            // the mixed class strips, 2 = mixed class strips (k_mxv_strip), 1 = class-partitioned items (k_mxv_long_grp),
                            // 0 = chunk kernel (k_mxv_long)
    int short_kernel = 6;   // short rows of a split matrix: 6 = by size (tagged row groups from lean_min_nnz entries, row groups below), 5 = tagged
                            // row groups (k_mxv_rows_tag), 1 = row-group kernel (k_mxv_rows), 0 = merge-path kernel.  (2 / 3 / 4 -- sliced
                            // ELLPACK, persistent row groups with an LDS head, a lane per row -- were measured slower in rounds 1-2 and are
                            // gone from the tree: DESIGN.md section 4.1.3 keeps their numbers)
    int64_t lean_min_nnz = 48ll << 20;  // the round-3 layouts pay from about this many entries (measured: one rank's block of an 8-way
                            // scale-24 run -- 33 M entries -- 0.130 ms on the round-2 layouts against 0.144, scale 22 -- 67 M -- 0.203 against 0.190)
    GrX_Stats stats{};
    int debug_flags = 0;    // GRB_DEBUG: kernel ablation switches (benchmark diagnostics only)
    int tune_pull_ipt = 0;  // GRB_PULL_IPT: merge items per thread of the pull SpMV (0 = default)
    int64_t hot_min_cols = 1 << 20;  // matrices at least this wide get a hot-column table (pull SpMV)
    int64_t hot_k = 0;               // table entries (0 = ~2 MiB of x values)
    int64_t split_min_nnz = 1 << 22;  // matrices with at least this many entries are analysed for the long/short row split
    int split_min_len = 0;            // a row is "long" from this many entries (0 = 64 for the class strips, 256 for the item kernel)
    int long_classes = 16;            // column classes of the class strips (8, 16, 32, 64): distinct LDS heads across the chip
    int lazy_layout = 1;             // 1: the SpMV layouts of a large matrix are built at its SECOND pull product, not its first
    int64_t lazy_min_nnz = 1 << 22;  // ... for matrices with at least this many entries (smaller ones build at once)
    int push_mode = 1;               // 0 never push, 1 push when u has < n/64 entries, 2 always push (tests)
    int rows_tile = 1;               // 1: the short rows of an ordered matrix as sorted row tiles (k_mxv_rtile) where the call allows it (full operand, specialised semiring)
    int64_t stream_nt_min_nnz = 48ll << 20;  // matrices with at least this many entries stream their layouts non-temporal (k_mxv_rtile / _bool / k_mxv_ctile)
    int bool_probe = 8;  // entries of a long row of a BOOL matrix tested by k_long_init before the item kernels (terminal monoids: LOR, ANY); 0 = off
    int rtile_rows = 8192;           // ... rows per tile (8192 or 16384; 8-byte accumulators: half)
    int64_t rtile_entries = 49152;   // ... and about this many entries (round 6, tiles handed out heaviest first: 32 Ki / 40 Ki / 48 Ki / 64 Ki / 96 Ki / 128 Ki = 0.469 / 0.460 / 0.464 / 0.465 / 0.471 / 0.483 ms masked, 0.589 / - / - / 0.573 / - / 0.603 unmasked; profiles/r06/option_sweeps.txt)
    int rows_head = 1;               // 1: the short rows of an ordered matrix run with the hottest columns of the operand in LDS (k_mxv_rows_tag<.., HEAD>)
    int64_t rows_head_min_groups = 16384;  // ... from this many row groups of 64 (below, filling 512 heads costs more than they save)
    int push_small = 1;              // 1: a pushed frontier of at most 64 work items runs its three passes in one workgroup (k_push_small)
    int fill_absent = 1;             // 1: min_plus / max_plus over floating point with a sparse operand run the full-operand kernels on an image with the
                                     // absorbing value under the absent entries (ordered layouts, finite values; exact: section 4.1.10)
    int hub_min_len = 1024;          // rows of an ordered matrix from this many entries are dealt to 64 column classes (0: no hub level)
    int value_dict = 1;              // 1: hot-strip records carry 1-byte value codes when the matrix has at most 256 distinct values
    int order_mode = 1;              // 1: large square matrices get popularity-ordered pull layouts and keep their operands in that order
                                     // (grb_mxv_order.inc); 0: never
    int64_t order_min_nnz = 24ll << 20;  // ... from this many entries.  Round 5 (profiles/r05/small_scales.txt): scale 21 (33.5 M entries) 0.130 -> 0.087 ms unmasked,
                                         // 0.102 -> 0.080 masked on the ordered layouts; scale 20 (16.8 M) 0.067 -> 0.072: the crossover lies between.  The twin
                                         // always takes the lean layouts (hot strips + cold tiles + row tiles), whatever lean_min_nnz says
    int64_t reorder_count = 0;       // vectors converted between vertex orders so far (cumulative)
    void *host_stage = nullptr;      // 8 MiB of page-locked host memory: the library's own tables travel through it in both directions (grb_context.hip)
    bool host_stage_failed = false;
    void *host_pinned = nullptr;     // 4 KiB of page-locked host memory: small device-to-host reads land here (no staging copy in the runtime)
    unsigned long long *push_counters = nullptr;  // the thin push path's counters (grb_mxv_push.inc): two sets of four words, used in turn --
    int push_parity = 0;                          // a call's frontier kernel zeroes the set of the next call
    int lazy_tagged = 1;             // (round 6) 1: a matrix whose short rows have sorted row tiles builds the tagged row groups' ENTRIES (the layout of the calls the
                                     // tiles do not take) only when such a call arrives -- from the tiles; their offsets and "row has an entry" words are always built.
                                     // Scale 24: 0.32 GB less cached layouts and 2.3 ms less layout build when every call takes the tiles
    int ctile_pack = 0;              // (round 6, MEASURED AND OFF) cold tiles of an ordered matrix with every column range below 2^19 codes: 1 = an entry's column (offset in
                                     // its range) and row slot in ONE word (slot << 19 | column; tiles of 8192 slots, up to 128 ranges): 8 instead of 10 bytes per
                                     // entry; 2 = ... and one-byte dictionary codes for its value (5 bytes).  Headline 0.461-0.471 ms with three streams, 0.471-0.479
                                     // packed, 0.468-0.480 packed + codes: 4204 tiles instead of 2333 cost what 46 / 115 MB of stream save (profiles/r06/ctile_pack.txt)
    int strip_slot16 = 1;            // (round 6) 1: the strips keep a lane's accumulator slot as a 16-bit offset from its chunk's smallest slot (+ one base per chunk)
                                     // where every chunk's slots span less than 65535 rows: 2 instead of 4 bytes per lane record of the slot stream
    int rtile_pack = 1;              // (round 6) 1: the sorted row tiles of a dictionary-coded matrix with at most 2^24 columns keep column code and value
                                     // code in ONE 32-bit word per entry (code << 24 | column): 6 instead of 7 bytes per entry, one stream less per block
    int cold_in_rows = 0;            // (round 6, MEASURED AND OFF) > 0: long rows of an ordered matrix with fewer entries than this (and than hub_min_len) keep
                                     // only their LDS-resident (hot) entries in the strips; their COLD entries join the short rows' sorted row tiles (and
                                     // tagged row groups), whose kernel merges the strips' accumulator into the row's result.  Headline, scale 24: 0.458 ->
                                     // 0.487 ms with 1024 (k_mxv_ctile 124 -> 64 us, k_mxv_rtile 179 -> 270 us), unmasked 0.573 -> 0.665: an unpinned cold
                                     // gather moves a 128-byte line over the fabric EVERY time (0.7 GB more for the 5.5 M admitted gathers that moved), the
                                     // cold tiles' XCD-pinned column ranges fetch a line once per XCD (profiles/r06/cold_in_rows.txt).  0 = all cold
                                     // entries of the long rows as cold tiles (rounds 3-5)
};
Context &ctx();
void require_init();

// stream-ordered device memory
void *dev_alloc(size_t bytes);
void *dev_alloc_zero(size_t bytes);
void dev_free(void *p);
void dev_cache_release();  // return every cached block to the HIP allocator
void h2d(void *dst, const void *src, size_t bytes);
void d2h(void *dst, const void *src, size_t bytes);  // synchronous w.r.t. the host on return
void d2d(void *dst, const void *src, size_t bytes);
void sync_stream();

template <typename T>
struct DevBuf {  // RAII temporary
    T *p = nullptr;
    explicit DevBuf(size_t n, bool zero = false)
    {
        p = static_cast<T *>(zero ? dev_alloc_zero(sizeof(T) * (n ? n : 1)) : dev_alloc(sizeof(T) * (n ? n : 1)));
    }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { dev_free(p); }
    T *release()
    {
        T *q = p;
        p = nullptr;
        return q;
    }
};

// one per translation unit with kernels: touches a kernel so that the HIP runtime loads the unit's code object (GrB_init)
void preload_mxv();
void preload_stage();
void preload_mxm();
void preload_vecops();
void preload_object();
void preload_prim();

template <typename T>
struct DevPtr {  // owner of a device block that a callee allocated (freed unless released)
    T *p = nullptr;
    DevPtr() = default;
    explicit DevPtr(T *q) : p(q) {}
    DevPtr(const DevPtr &) = delete;
    DevPtr &operator=(const DevPtr &) = delete;
    ~DevPtr() { dev_free(p); }
    T *release()
    {
        T *q = p;
        p = nullptr;
        return q;
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t bits_words64(uint64_t n) { return (size_t)((n + 63) / 64); }

}  // namespace grb

// ---- opaque object definitions (global namespace: they are the C-ABI's incomplete types) ----------
struct GB_Type_opaque {
    int code;
    size_t size;
    const char *name;
};
struct GB_BinaryOp_opaque {
    int op;
    int type;
    const char *name;
};
struct GB_Monoid_opaque {
    int op;
    int type;
    const char *name;
};
struct GB_Semiring_opaque {
    int monoid;
    int mult;
    int type;
    const char *name;
};
struct GB_Descriptor_opaque {
    bool replace, comp, structure, t0, t1;
    bool builtin;
};

// A vertex order: the permutation a large square matrix's pull layouts are built in -- vertices by falling
// column count, so that a 128-byte line of an operand holds 32 columns of about the same popularity and the hot columns are the
// first K positions -- shared, by reference count, with the vectors that are kept in that order between calls.  Position p of an
// ordered vector holds the element with natural index d_inv[p]; d_rank is the inverse map.
struct GB_Perm {
    int64_t refs = 1;
    uint64_t n = 0;
    int32_t *d_rank = nullptr;  // natural index -> position
    int32_t *d_inv = nullptr;   // position -> natural index
    int64_t n_live_rows = 0;    // positions >= this hold vertices whose ROW of the matrix the order was built from is empty
    int64_t n_live_cols = 0;    // positions >= this hold vertices that (by the sampled histogram) no entry of that matrix refers to
};

constexpr size_t VEC_VAL_PAD = (size_t)8 << 20;    // the default hot-column table is 2 MiB of values ...
constexpr size_t VEC_BITS_PAD = (size_t)256 << 10;  // ... and at most 2 Mi presence bits (1-byte types)

struct GB_Vector_opaque {
    uint64_t magic;
    GrB_Type type;
    uint64_t n;
    void *d_val;       // n values (allocated lazily), nullptr while the vector has never been written
    uint64_t *d_bits;  // presence, ceil(n/64) words, bits >= n are always 0
    bool padded;       // the allocations start VEC_VAL_PAD / VEC_BITS_PAD bytes before d_val / d_bits: the pull SpMV
                       // writes its hot-column table there, so that [table | values] is one image without a copy
    int64_t nvals;     // -1 = unknown (counted on demand)
    GB_Perm *order = nullptr;  // nullptr: natural order; else the elements are stored in this vertex order (a reference is held).
                               // Every entry point takes vectors in natural order (check_vector converts) except the ones that are
                               // order-aware: mxv / vxm on the matrix the order belongs to, the element-wise operations, reduce, dup,
                               // element access.  An ordered vector without entries is converted for free.
    bool exported = false;     // its device pointers were handed out (GrX_Vector_export_dense_device): conversions keep them (copy back)
    bool pinned = false;       // never leave this vector in another than the natural order (its HBM image is aliased outside the library)
    std::string err;
};

struct GB_Matrix_opaque {
    uint64_t magic;
    GrB_Type type;
    uint64_t nrows, ncols;
    int64_t nvals;
    int64_t *d_ptr;  // nrows+1 (nullptr while empty)
    int32_t *d_col;
    void *d_val;  // nvals values, or 1 value when iso
    bool iso;
    bool owns;             // false for adopted (GrX import, copy=0) buffers
    GB_Matrix_opaque *tr;  // cached transpose (owned), or nullptr
    // merge-path tile table for the pull SpMV (grb_mxv.hip), built on first use
    int64_t *d_tile_row;
    int64_t n_tiles;
    int tile_items;
    // hot-column table for the pull SpMV (grb_mxv.hip): the K most referenced columns are renumbered
    // 0..K-1 (their x entries are gathered into a small L2-resident table per call), all others K+col
    int32_t *d_col_hot;   // nvals re-coded column indices, or nullptr
    int32_t *d_hot_cols;  // K original column indices, hottest first
    int64_t hot_k;
    int hot_state;        // 0 = not analysed, 1 = enabled, -1 = not worth it
    bool hot_cols_dropped = false;  // d_col_hot was released after the split was built from it (ensure_split)
    // long/short row split for the pull SpMV (grb_mxv.hip): rows with >= split_min_len entries are processed by a
    // lean wavefront-per-chunk kernel straight from this matrix's arrays; the remaining rows live in `short_part`
    // (same shape, long rows empty) and go through the merge-path kernel
    GB_Matrix_opaque *short_part;
    uint64_t *d_long_bits;    // bit r: row r is long
    int32_t *d_long_rows;     // n_long row indices
    int32_t *d_chunk_slot;    // per chunk: index into d_long_rows
    int64_t *d_chunk_start;   // per chunk: first entry
    int32_t *d_chunk_len;     // per chunk: entries (<= PULL_CHUNK)
    int32_t *d_long_prefix;   // per 64-row group: number of long rows before it
    // class-partitioned copy of the long rows (k_mxv_long_grp): items = at most LONG_ITEM entries of one (column class,
    // long row), stored contiguously in d_lcol / d_lval from d_it_start[i] (multiple of 4), sorted by class, then falling length
    int32_t *d_lcol;
    void *d_lval;             // nullptr for iso matrices
    int64_t *d_it_start;
    int32_t *d_it_len;
    int32_t *d_it_slot;
    int64_t *d_item_begin;    // device copy of item_begin
    int64_t item_begin[9];    // items of class c are [item_begin[c], item_begin[c+1])
    int64_t n_items;
    int64_t long_nnz;
    int cls_lds_lim;          // codes below it are stored pre-translated to LDS slots in d_lcol
    // ... or as flat class strips (k_mxv_strip, long_kernel = 2): d_lcol / d_lval hold the entries of class c in chunks
    // [strip_cb[c], strip_cb[c+1]) of 512 entries, sorted by (sub-range, row); segments start at multiples of 8 entries
    unsigned long long *d_sstart = nullptr;  // per chunk: bit l = a segment starts at lane l's 8 entries
    int32_t *d_sslot = nullptr;        // per lane (8 entries): accumulator slot (index into d_long_rows), -1 = padding
    int64_t strip_cb[161] = {0};        // (+ the hub level's classes behind the first strip_ncls, see hub_ncls)
    int strip_ncls = 8;
    // ... a matrix in its popularity order deals the entries of its HUB rows (>= hub_min_len entries) to hub_ncls = 64 classes instead of
    // 16: four times the LDS-resident codes (2.5 Mi: 95 % of the references of an R-MAT graph instead of 81 %) at segments that are still
    // long enough for the 8-entry lane records.  Their strips are the classes [strip_ncls, strip_ncls + hub_ncls) of strip_cb, run by a
    // second launch of the strip kernel
    int hub_ncls = 0;                  // 0: no hub level
    int hub_lds_lim = 0;               // codes below it are LDS-resident in the hub level's classes
    int32_t *d_wg_tab = nullptr;       // workgroup table of the one-launch form of k_mxv_hstrip over both levels (PullArgs::wg_tab), wg_tab_g workgroups
    int64_t *d_strip_cb = nullptr;     // device copy of strip_cb
    int wg_tab_g = 0;
    int64_t strip_nseg = 0;
    // ... or as HOT / COLD strips (long_kernel = 4, split kind 3; grb_mxv_strip.inc): the entries whose column code is LDS-resident in
    // its class live in d_hrec -- per lane of 8 entries one record [8 LDS slots as u16 (u32 for BOOL) | 8 values] -- in chunks
    // [strip_cb[c], strip_cb[c+1]) of the hot classes; all other entries (gathered from the operand image) in d_lcol / d_lval as
    // strips of `strip_cold_ncls` contiguous column ranges, chunks [cold_cb[c], cold_cb[c+1]) in the same chunk numbering
    // (d_lcol[0] is the first entry of chunk cold_cb[0]); d_sstart / d_sslot cover both
    char *d_hrec = nullptr;
    int hrec_bytes = 0;                // bytes of one lane record
    // ... with the values DICTIONARY-CODED when the matrix holds at most 256 distinct ones (4-byte types; grb_mxv_vdict.inc): a lane
    // record is [8 LDS slots as u16 | 8 value codes as u8] = 24 bytes, the kernels look the values up (LDS / L1) -- exact, whatever the
    // values are, and half the bytes of the stream the hot strips are bound by
    void *d_vdict = nullptr;           // 256 values of the matrix type (unused codes: 0)
    int vdict_n = 0;                   // distinct values found (0: no dictionary)
    double vals_absmax = 0.0;          // ... and the largest magnitude among them (from the dictionary)
    bool vals_finite = false;          // every stored value is finite (known from the dictionary's scan): lets a sparse operand of a min_plus /
                                       // max_plus product be run as a full one with +-inf under its absent entries (mxv_core)
    unsigned long long *d_vd_table = nullptr;  // the hash table value -> slot the codes were assigned from (build time: placement kernels)
    unsigned char *d_vd_codes = nullptr;       // code of every slot
    // (the cold tiles and the tagged row groups of such a matrix carry the one-byte codes too: d_ct_val / d_tg_val are then byte arrays)
    // ... the cold entries as tagged tiles (k_mxv_ctile, grb_mxv_ctile.inc): sorted by (column range, long row), tile t = (range
    // t / ct_nsb, rows [8192 (t % ct_nsb), ...)) holds entries [tiles[t].u0 * 4, ... + 4 n_units): column code, value, 16-bit row in the tile
    int32_t *d_ct_col = nullptr;
    void *d_ct_val = nullptr;          // nullptr for iso matrices
    uint16_t *d_ct_loc = nullptr;
    void *d_ct_tiles = nullptr;        // CTile[ct_ncr * ct_nsb]
    int ct_nsb = 0, ct_ncr = 0;
    int64_t ct_units = 0;
    int split_kind = 0;                // value of the long_kernel option the split was built for
    int pull_calls = 0;                // pull products run on this matrix since its layouts were last dropped
    // the short rows as tagged row groups (k_mxv_rows_tag; short_kernel = 5): per group of 64 rows its entries contiguous, padded to a
    // multiple of 4, one byte per entry naming its row inside the group (64 = padding)
    int32_t *d_tg_off = nullptr;       // per group (+1): first entry / 4
    int32_t *d_tg_col = nullptr;
    void *d_tg_val = nullptr;          // nullptr for iso matrices
    unsigned char *d_tg_tag = nullptr;
    uint64_t *d_tg_nonempty = nullptr; // per group: bit l = short row 64 g + l has an entry
    int64_t tg_units = 0;
    int tg_state = 0;  // 0 nothing, 2 offsets + non-empty words (ensure_tagged_index), 1 + the entries (ensure_tagged)
    bool short_tagged_only = false;    // the short part keeps its row pointers only: its entries live in the tagged row groups
    // ... and, round 5, an ordered twin's short rows a second time as SORTED ROW TILES (k_mxv_rtile, grb_mxv_rtile.inc): tiles of up to
    // rt_rows4 rows (8-byte accumulators: half) and ~rtile_entries entries, the entries of a tile sorted by column code in lane-transposed
    // blocks of 256: column code, 16-bit row inside the tile, value (one-byte code with a dictionary)
    int32_t *d_rt_col = nullptr;
    uint16_t *d_rt_tag = nullptr;
    void *d_rt_val = nullptr;
    void *d_rt_tiles = nullptr;        // RTile[rt_ntiles]
    int32_t *d_rt_order = nullptr;     // tile numbers, heaviest first (the order they are handed out in)
    unsigned int *d_rt_counter = nullptr;  // the hand-out counter of a call (zeroed by k_long_init)
    // bottom-up probe (round 5, BOOL matrices with class items): the first probe_k column codes of every long row, k-major
    // (d_probe[k * n_long + i]; -1 = the row has no k-th entry) -- k_long_init tests them before any item kernel runs
    int32_t *d_probe = nullptr;
    int probe_k = 0;
    int64_t rt_units = 0;
    int rt_ntiles = 0;
    int rt_rows4 = 0;                  // the tile height the layout was built for (rows with 4-byte accumulators)
    int rt_state = 0;                  // 0 = not built, 1 = built, -1 = not possible (sizes)
    // ---- vertex order (round 4; grb_mxv_order.inc): a large square matrix is laid out a second time as ord = P A P' for a permutation P
    //      by falling column count (perm); the pull kernels run on `ord` with operands kept in that order.  `ord` is a matrix object of its
    //      own whose column indices ARE the hot codes (hot_identity): the first hot_k positions are the hot table, no image is built
    bool ranked = false;               // (GrX_Matrix_hint_ranked, round 5) the caller's labels ARE popularity ranks: column j is referred to about as often as
                                       // or more often than column j + 1, heavy rows first.  The ordered layouts are then built in the caller's own order --
                                       // no permutation, vectors stay natural -- for square and non-square matrices alike (the row blocks of a sharded run
                                       // whose graph was relabelled once, up front).  A performance hint: no result depends on it.
    GB_Perm *perm = nullptr;           // the order of this matrix's vertex space (shared with its transpose and with vectors)
    bool col_order_only = false;       // (GrX_Matrix_shard_setup, round 6) `perm` orders the COLUMN space only: the matrix is a row block of a sharded
                                       // graph (m rows over n columns), its twin keeps the rows as they are and renames the columns; only the
                                       // operand of a product carries the order, outputs and masks (m elements) stay natural
    GB_Matrix_opaque *ord = nullptr;   // the matrix in that order (owned): layouts only -- its CSR arrays are released once they are built
    GB_Matrix_opaque *tr_of = nullptr; // this matrix is the cached transpose of tr_of (not owned)
    int ord_state = 0;                 // 0 = not analysed, 1 = `ord` is built, -1 = not worth it / not possible
    uint64_t ord_sig = 0;              // the layout options `ord` was built under
    bool hot_identity = false;         // (an `ord` twin) d_col_hot aliases d_col: codes are positions; codes >= hot_k are not offset by hot_k
    int64_t ord_live_rows = 0;         // (an `ord` twin) rows at and behind this position are empty
    int32_t *d_cold_bounds = nullptr;  // (an `ord` twin) first code of every column range of the cold tiles (ct_ncr + 1 values)
    int32_t *d_ct_order = nullptr;     // (cold tiles) tile numbers in the order the XCDs walk them: XCD x takes d_ct_order[ct_xoff[x] .. ct_xoff[x + 1])
    int64_t ct_xoff[9] = {0};
    int64_t ct_ntiles = 0;
    int64_t n_long, n_chunks;
    int split_state;          // 0 = not analysed, 1 = enabled, -1 = not worth it
    bool split_hot;           // short_part's columns are hot-coded
    std::string err;
    int ct_mode = 0;                   // (round 6, Context::ctile_pack) 0: d_ct_col / d_ct_val / d_ct_loc as three streams; 1: d_ct_col holds slot << 19 | (column - base of
                                       // the tile's column range), no d_ct_loc, the tiles' range bases behind the CTile array; 2: ... and d_ct_val holds one-byte codes
    uint16_t *d_sslot16 = nullptr;     // (round 6, Context::strip_slot16) per lane: slot - d_sslot_base[chunk], 0xffff = padding; then d_sslot is released
    int32_t *d_sslot_base = nullptr;   // ... per chunk of 64 lanes: its smallest slot
    int64_t tails_max_len = 0;         // > 0: long rows with fewer entries than this have their cold entries in `short_part` (Context::cold_in_rows): the
                                       // short-row kernels add them up and MERGE the long-row accumulator into the row's result
};

namespace grb {

GrB_Type type_of_code(int code);

void vector_set_order(GB_Vector_opaque *v, GB_Perm *order);     // converts v (in place: its device pointers stay) into `order` (nullptr = natural)
// validity only: for the entry points that work in whatever vertex order the vector is stored in
inline void check_vector_any(const GB_Vector_opaque *v, const char *what)
{
    if (!v) fail(GrB_NULL_POINTER, std::string(what) + " is NULL");
    if (v->magic != MAGIC_VECTOR)
        fail(v->magic == MAGIC_FREED ? GrB_UNINITIALIZED_OBJECT : GrB_INVALID_OBJECT, std::string(what) + " is not a valid GrB_Vector");
}
// ... and the default: the vector is brought back to the natural order first
inline void check_vector(const GB_Vector_opaque *v, const char *what)
{
    check_vector_any(v, what);
    if (v->order) vector_set_order(const_cast<GB_Vector_opaque *>(v), nullptr);
}
inline void check_matrix(const GB_Matrix_opaque *A, const char *what)
{
    if (!A) fail(GrB_NULL_POINTER, std::string(what) + " is NULL");
    if (A->magic != MAGIC_MATRIX)
        fail(A->magic == MAGIC_FREED ? GrB_UNINITIALIZED_OBJECT : GrB_INVALID_OBJECT, std::string(what) + " is not a valid GrB_Matrix");
}

inline std::string *errp(GB_Vector_opaque *v) { return (v && v->magic == MAGIC_VECTOR) ? &v->err : nullptr; }
inline std::string *errp(GB_Matrix_opaque *A) { return (A && A->magic == MAGIC_MATRIX) ? &A->err : nullptr; }

// ---- object services implemented in grb_object.hip --------------------------------------------------
void perm_retain(GB_Perm *p);
void perm_release(GB_Perm *p);                                  // frees the maps with the last reference
// one order for the operands of an element-wise operation: the first order found on a vector that holds entries (natural if one of them
// is pinned); every vector is converted to it (vectors without entries for free)
GB_Perm *vectors_common_order(GB_Vector_opaque *const *vs, int count);
uint64_t vector_position(GB_Vector_opaque *v, uint64_t i);      // where element i of v is stored (i itself in natural order)
void vector_ensure_storage(GB_Vector_opaque *v);               // allocate zeroed values+bits if absent
void vector_release_storage(GB_Vector_opaque *v);              // free buffers, nvals = 0
void vector_alloc_pair(const GB_Vector_opaque *v, bool padded, bool zero_val, void **val, uint64_t **bits);  // presence zeroed
void vector_free_pair(bool padded, void *val, uint64_t *bits);
int64_t vector_nvals(GB_Vector_opaque *v);                     // counts if unknown
int64_t vector_index_list(GB_Vector_opaque *v, uint64_t **d_idx);  // ascending indices of the entries (fresh device array)
GB_Vector_opaque *vector_new(GrB_Type type, uint64_t n);
void vector_free(GB_Vector_opaque *v);
void matrix_release_storage(GB_Matrix_opaque *A);              // frees CSR + caches, nvals = 0
GB_Matrix_opaque *matrix_new(GrB_Type type, uint64_t nrows, uint64_t ncols);
void matrix_free(GB_Matrix_opaque *A);
void matrix_invalidate_caches(GB_Matrix_opaque *A);
const int64_t *matrix_rowptr(GB_Matrix_opaque *A);             // allocates a zero row-pointer array for empty matrices
GB_Matrix_opaque *matrix_transpose_cached(GB_Matrix_opaque *A);  // builds A->tr on first use
// values cast: dst[i] = (dst_type) src[i]   (GraphBLAS typecast rules)
void cast_array(int dst_type, void *dst, int src_type, const void *src, int64_t n);
// returns a matrix of `type` sharing structure (fresh values); caller frees with matrix_free.  nullptr if A already has that type.
GB_Matrix_opaque *matrix_cast_copy(GB_Matrix_opaque *A, int type);
GB_Vector_opaque *vector_cast_copy(GB_Vector_opaque *v, int type);
// out_bits = present(v) & (structure ? 1 : value != 0)
void pack_bool_pv(const uint64_t *present, const bool *val, int64_t n, uint32_t *out);  // 2 bits per element: (present, value)
void vector_mask_bits(GB_Vector_opaque *m, bool structure, uint64_t *out_bits);
void vector_write_rule(GB_Vector_opaque *w, const void *t_val, const uint64_t *t_bits, const uint64_t *m_bits, bool comp, int accum,
                       bool replace);  // w<m, replace> = accum(w, t), in place, t of w's type
void pack_bool_values(const uint64_t *present, const bool *val, int64_t n, uint64_t *out);

// C<Mask, replace> = accum(C, T), T in C's type with sorted rows (grb_mxm.hip); takes T's storage when nothing masks or accumulates
void matrix_apply_write_rule(GB_Matrix_opaque *C, GB_Matrix_opaque *Mask, const GB_BinaryOp_opaque *accum, GB_Matrix_opaque *T,
                             bool replace, bool comp, bool structure);

// ---- primitives implemented in grb_prim.hip (rocPRIM-backed) ---------------------------------------
void prim_sort_pairs_u64_u32(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                             int64_t n, int end_bit);
// (stable; orders by key bits [begin_bit, end_bit) only)
void prim_sort_pairs_u64_u32_bits(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                                  int64_t n, int begin_bit, int end_bit);
void prim_exclusive_sum_i64(const int64_t *in, int64_t *out, int64_t n);  // in == out allowed

}  // namespace grb

// ---- C boundary helpers ----------------------------------------------------------------------------
#define GRB_TRY try {
#define GRB_CATCH(errstr_ptr)                                    \
    }                                                            \
    catch (const ::grb::Error &e)                                \
    {                                                            \
        std::string *_s = (errstr_ptr);                          \
        if (_s) *_s = e.msg;                                     \
        return e.info;                                           \
    }                                                            \
    catch (const std::bad_alloc &)                               \
    {                                                            \
        return GrB_OUT_OF_MEMORY;                                \
    }                                                            \
    catch (...)                                                  \
    {                                                            \
        return GrB_PANIC;                                        \
    }                                                            \
    return GrB_SUCCESS;
