// grb_context.hip -- library context: device selection, stream, stream-ordered memory, timing hooks.
// Counterpart of the reference's `initialize(blocking=..., memory_manager="numpy")` call at
// graphblas/__init__.py:170-173 (there: SuiteSparse GrB_init on the host; here: a gfx950 device is
// mandatory -- there is no CPU fallback).
#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "grb_internal.hpp"

namespace grb {

// "debug_flags": the shipped library only knows the switches that choose between two correct paths on the host side
// (128 no long/short row split, 256 no LDS bitmap in the symbolic SpGEMM pass, 2048 no (presence, value) packing, 65536 no
// row-length path); the kernel ablation switches -- some of which make results wrong on purpose -- are compiled in by
// -DGRB_ABLATE only (make ablate; the emulator build of the CPU test tier).
#ifdef GRB_ABLATE
static constexpr int DEBUG_FLAGS_MASK = ~0;
#else
static constexpr int DEBUG_FLAGS_MASK = 128 | 256 | 2048 | 65536;
#endif


Context &ctx()
{
    static Context c;
    return c;
}

void require_init()
{
    if (!ctx().initialized) fail(GrB_PANIC, "GrB_init has not been called (or no HIP device is available)");
}

// Device memory: hipMallocAsync / hipFreeAsync on the library's stream, behind a size-class cache.  Every kernel of the
// library runs in order on that one stream, so a block freed by one call can be handed to the next allocation of its
// size class without touching the HIP allocator (whose stream-ordered calls were measured to leave the GPU idle for
// tens of microseconds between two GrB calls).  The cache is emptied when an allocation fails, when the stream changes
// and at GrB_finalize.
namespace {
struct BlockCache {
    std::mutex mu;
    std::unordered_map<size_t, std::vector<void *>> free_blocks;  // size class -> blocks
    std::unordered_map<void *, size_t> size_of;                   // live + cached blocks -> size class
    size_t cached_bytes = 0;
};
BlockCache &cache()
{
    static BlockCache c;
    return c;
}
// classes: multiples of 512 B up to 64 KiB, then eight steps per power of two (at most 12.5 % over-allocation)
size_t size_class(size_t bytes)
{
    if (bytes <= 65536) return (bytes + 511) & ~(size_t)511;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
}
}  // namespace

void dev_cache_release()
{
    BlockCache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto &kv : c.free_blocks)
        for (void *p : kv.second) {
            c.size_of.erase(p);
            (void)hipFreeAsync(p, ctx().stream);
        }
    c.free_blocks.clear();
    c.cached_bytes = 0;
}

void *dev_alloc(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    BlockCache &c = cache();
    const size_t cls = size_class(bytes);
    if (ctx().alloc_cache) {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.free_blocks.find(cls);
        if (it != c.free_blocks.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            c.cached_bytes -= cls;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMallocAsync(&p, cls, ctx().stream);
    if ((e != hipSuccess || !p) && c.cached_bytes) {
        (void)hipGetLastError();
        dev_cache_release();
        (void)hipStreamSynchronize(ctx().stream);
        p = nullptr;
        e = hipMallocAsync(&p, cls, ctx().stream);
    }
    if (e != hipSuccess || !p) {
        (void)hipGetLastError();
        fail(GrB_OUT_OF_MEMORY, "device allocation of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    }
    if (ctx().alloc_cache) {
        std::lock_guard<std::mutex> lock(c.mu);
        c.size_of[p] = cls;
    }
    return p;
}

void *dev_alloc_zero(size_t bytes)
{
    void *p = dev_alloc(bytes);
    GRB_HIP(hipMemsetAsync(p, 0, bytes ? bytes : 16, ctx().stream));
    return p;
}

void dev_free(void *p)
{
    if (!p) return;
    BlockCache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.size_of.find(p);
        if (it != c.size_of.end()) {
            if (ctx().alloc_cache && ctx().initialized) {
                c.free_blocks[it->second].push_back(p);
                c.cached_bytes += it->second;
                return;
            }
            c.size_of.erase(it);
        }
    }
    (void)hipFreeAsync(p, ctx().stream);
}

// Host <-> device copies of the library's own small tables (tile descriptors, class bounds, counts): through page-locked memory of the
// context -- 4 KiB for the scalars, STAGE_BYTES for the tables -- instead of the runtime's staging of pageable memory.  Measured on the
// layout-building call of the headline matrix (profiles/r05/layout_build_timeline.txt): host time inside its 26 hipMemcpyAsync calls
// 13.0 -> 6.6 ms; the call's wall time does not move (44 ms: 39 ms of kernels, the waits reappear in the synchronisations).
// Larger copies (the caller's tuples at ingress / egress) keep the runtime's path.
static constexpr size_t STAGE_BYTES = 8u << 20;
static void *stage_block(size_t bytes)
{
    Context &c = ctx();
    if (bytes <= 4096) {
        if (!c.host_pinned && hipHostMalloc(&c.host_pinned, 4096, 0) != hipSuccess) {
            (void)hipGetLastError();
            c.host_pinned = nullptr;
        }
        if (c.host_pinned) return c.host_pinned;
    }
    if (bytes <= STAGE_BYTES) {
        if (!c.host_stage && !c.host_stage_failed && hipHostMalloc(&c.host_stage, STAGE_BYTES, 0) != hipSuccess) {
            (void)hipGetLastError();
            c.host_stage = nullptr;
            c.host_stage_failed = true;
        }
        return c.host_stage;
    }
    return nullptr;
}
void preload_stage() { (void)stage_block(4096); (void)stage_block(STAGE_BYTES); }
// (GRB_TRACE_COPIES=1: every host <-> device copy of the library that holds the host for more than 0.5 ms, on stderr)
static bool trace_copies()
{
    static const bool on = [] { const char *e = getenv("GRB_TRACE_COPIES"); return e && atoi(e) != 0; }();
    return on;
}
struct CopyTrace {
    const char *what;
    size_t bytes;
    bool staged;
    std::chrono::steady_clock::time_point t0;
    CopyTrace(const char *w, size_t b, bool s) : what(w), bytes(b), staged(s), t0(std::chrono::steady_clock::now()) {}
    ~CopyTrace()
    {
        if (!trace_copies()) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 0.5) fprintf(stderr, "[grb copy] %s %zu bytes %s: %.2f ms\n", what, bytes, staged ? "staged" : "direct", ms);
    }
};
void h2d(void *dst, const void *src, size_t bytes)
{
    if (!bytes) return;
    Context &c = ctx();
    if (void *st = stage_block(bytes)) {
        CopyTrace tr("h2d", bytes, true);
        memcpy(st, src, bytes);
        GRB_HIP(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, c.stream));
        GRB_HIP(hipStreamSynchronize(c.stream));  // (the block is free for the next copy)
        return;
    }
    CopyTrace tr("h2d", bytes, false);
    GRB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c.stream));
    GRB_HIP(hipStreamSynchronize(c.stream));  // the caller's host buffer is only borrowed for the call
}
void d2h(void *dst, const void *src, size_t bytes)
{
    // (entry counts, reduced scalars, the push path's counters, tile tables: a copy into pageable memory is staged by the runtime -- the
    //  page-locked blocks take them directly)
    Context &c = ctx();
    if (bytes) {
        if (void *st = stage_block(bytes)) {
            CopyTrace tr("d2h", bytes, true);
            GRB_HIP(hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, c.stream));
            GRB_HIP(hipStreamSynchronize(c.stream));
            memcpy(dst, st, bytes);
            return;
        }
        CopyTrace tr("d2h", bytes, false);
        GRB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
        GRB_HIP(hipStreamSynchronize(c.stream));
        return;
    }
    GRB_HIP(hipStreamSynchronize(c.stream));
}
void d2d(void *dst, const void *src, size_t bytes)
{
    if (bytes) GRB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream));
}
void sync_stream() { GRB_HIP(hipStreamSynchronize(ctx().stream)); }

}  // namespace grb

using namespace grb;

extern "C" GrB_Info GrB_init(GrB_Mode mode)
{
    Context &c = ctx();
    if (c.initialized) return GrB_INVALID_VALUE;
    if (mode != GrB_BLOCKING && mode != GrB_NONBLOCKING) return GrB_INVALID_VALUE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return GrB_PANIC;  // fail loudly: the product has no CPU path
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GrB_PANIC;
    c.device = dev;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c.num_cus = cus;
    }
    c.blocking = (mode == GrB_BLOCKING);
    c.stream = nullptr;
    // keep freed blocks in the pool: BFS/SSSP loops allocate and free the same temporaries every call
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        uint64_t thresh = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thresh);
    }
    (void)hipGetLastError();
    if (hipEventCreate(&c.ev0) != hipSuccess || hipEventCreate(&c.ev1) != hipSuccess) return GrB_PANIC;
    if (hipStreamCreateWithFlags(&c.aux_stream, hipStreamNonBlocking) != hipSuccess) c.aux_stream = nullptr;  // (no second stream: the kernels run one after another)
    if (hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming) != hipSuccess) return GrB_PANIC;
    // The HIP runtime loads a translation unit's code object at the first launch of one of its kernels: several ms for the
    // SpMV unit with its template instantiations -- paid by whichever product came first (it showed up as 8 of the 12.7 ms of
    // the first GrB_mxv of the scale-24 bench).  Load them here, once per process (GRB_PRELOAD=0: on demand, as before).
    if (const char *e = getenv("GRB_PRELOAD"); !e || atoi(e) != 0) {
        preload_mxv();
        preload_mxm();
        preload_vecops();
        preload_object();
        preload_prim();
        preload_stage();  // (the page-locked blocks of h2d / d2h: not inside the first matrix's layout build)
    }
    if (const char *e = getenv("GRB_DEBUG_FLAGS")) c.debug_flags = atoi(e) & DEBUG_FLAGS_MASK;
    // tuning knobs from the environment go through the same validation as GrX_option_set (an invalid value is ignored)
    static const struct { const char *env, *opt; } knobs[] = {
        {"GRB_PULL_IPT", "pull_ipt"}, {"GRB_MXV_OVERLAP", "mxv_overlap"}, {"GRB_STRIP_WGS", "strip_wgs"}, {"GRB_HOT_MIN_COLS", "hot_min_cols"}, {"GRB_HOT_K", "hot_k"}, {"GRB_PUSH_MODE", "push_mode"},
        {"GRB_ALLOC_CACHE", "alloc_cache"}, {"GRB_SHORT_KERNEL", "short_kernel"},
        {"GRB_LAZY_LAYOUT", "lazy_layout"}, {"GRB_MXM_HEAVY_KERNEL", "mxm_heavy_kernel"}, {"GRB_DROP_HOT_COLS", "drop_hot_cols"},
        {"GRB_MXM_UNIT_MIN_FLOPS", "mxm_unit_min_flops"}, {"GRB_MXM_UNIT_MIN_PER_WINDOW", "mxm_unit_min_per_window"},
        {"GRB_MXM_UNIT_SMALL", "mxm_unit_small"}, {"GRB_MXM_UNIT_DENSE", "mxm_unit_dense"}, {"GRB_MXM_UNIT_MID", "mxm_unit_mid"},
        {"GRB_MXM_BITMAP_POOL_MB", "mxm_bitmap_pool_mb"}, {"GRB_MXM_BITMAP_MIN_CNT", "mxm_bitmap_min_cnt"},
        {"GRB_LONG_KERNEL", "long_kernel"}, {"GRB_LONG_CLASSES", "long_classes"}, {"GRB_SPLIT_MIN_LEN", "split_min_len"},
        {"GRB_LONG_SUB", "long_sub"}, {"GRB_LONG_SUB_MIN_LEN", "long_sub_min_len"}, {"GRB_LEAN_MIN_NNZ", "lean_min_nnz"},
        {"GRB_MXM_MASK_MODE", "mxm_mask_mode"}, {"GRB_MAT_WRITE_KERNEL", "mat_write_kernel"}, {"GRB_MXM_SYM_WINDOWS", "mxm_sym_windows"}, {"GRB_MXM_WINDOW_GROUPS", "mxm_window_groups"}, {"GRB_MXM_CHECKSUM_PASS", "mxm_checksum_pass"}, {"GRB_MXM_XCD_MAP", "mxm_xcd_map"},
        {"GRB_ORDER_MODE", "order_mode"}, {"GRB_ORDER_MIN_NNZ", "order_min_nnz"}, {"GRB_VALUE_DICT", "value_dict"}, {"GRB_HUB_MIN_LEN", "hub_min_len"}, {"GRB_FILL_ABSENT", "fill_absent"},
        {"GRB_PUSH_SMALL", "push_small"}, {"GRB_ROWS_HEAD", "rows_head"}, {"GRB_ROWS_TILE", "rows_tile"}, {"GRB_COLD_IN_ROWS", "cold_in_rows"}, {"GRB_RTILE_PACK", "rtile_pack"}, {"GRB_STRIP_SLOT16", "strip_slot16"}, {"GRB_CTILE_PACK", "ctile_pack"}, {"GRB_LAZY_TAGGED", "lazy_tagged"}, {"GRB_BOOL_PROBE", "bool_probe"}, {"GRB_STREAM_NT_MIN_NNZ", "stream_nt_min_nnz"}, {"GRB_RTILE_ROWS", "rtile_rows"}, {"GRB_RTILE_ENTRIES", "rtile_entries"}, {"GRB_ROWS_HEAD_MIN_GROUPS", "rows_head_min_groups"},
    };
    c.initialized = true;  // (alloc_cache = 0 releases the block cache: only once the context is complete)
    for (const auto &k : knobs)
        if (const char *e = getenv(k.env))
            if (GrX_option_set(k.opt, atoll(e)) != GrB_SUCCESS)
                fprintf(stderr, "libgrb_mi355x: %s=%s rejected (not a valid value of option \"%s\"): the default stays\n", k.env, e, k.opt);
    return GrB_SUCCESS;
}

// SuiteSparse's initialiser with the caller's allocator (python-graphblas: graphblas/__init__.py:170-173 via
// suitesparse_graphblas.initialize(memory_manager="numpy")): this library allocates nothing on the host for the caller, but the GxB
// import / pack entries take OWNERSHIP of host arrays and release them with `user_free`.
extern "C" GrB_Info GxB_init(GrB_Mode mode, void *(*user_malloc)(size_t), void *(*user_calloc)(size_t, size_t),
                             void *(*user_realloc)(void *, size_t), void (*user_free)(void *))
{
    (void)user_malloc;
    (void)user_calloc;
    (void)user_realloc;
    const GrB_Info info = GrB_init(mode);
    if (info == GrB_SUCCESS) ctx().host_free = user_free;
    return info;
}

extern "C" GrB_Info GrB_finalize(void)
{
    Context &c = ctx();
    if (!c.initialized) return GrB_SUCCESS;
    if (c.push_counters) dev_free(c.push_counters);
    c.push_counters = nullptr;
    dev_cache_release();
    (void)hipStreamSynchronize(c.stream);
    if (c.host_pinned) (void)hipHostFree(c.host_pinned);
    c.host_pinned = nullptr;
    if (c.host_stage) (void)hipHostFree(c.host_stage);
    c.host_stage = nullptr;
    c.host_stage_failed = false;
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    c.ev0 = c.ev1 = nullptr;
    if (c.ev_fork) (void)hipEventDestroy(c.ev_fork);
    if (c.ev_join) (void)hipEventDestroy(c.ev_join);
    c.ev_fork = c.ev_join = nullptr;
    if (c.aux_stream) { (void)hipStreamSynchronize(c.aux_stream); (void)hipStreamDestroy(c.aux_stream); }
    c.aux_stream = nullptr;
    c.initialized = false;
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrB_getVersion(unsigned int *version, unsigned int *subversion)
{
    if (!version || !subversion) return GrB_NULL_POINTER;
    *version = GRB_VERSION;
    *subversion = GRB_SUBVERSION;
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrX_set_stream(void *hip_stream)
{
    GRB_TRY
    require_init();
    dev_cache_release();  // cached blocks were last used on the old stream
    sync_stream();
    ctx().stream = static_cast<hipStream_t>(hip_stream);
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_get_stream(void **hip_stream)
{
    GRB_TRY
    require_init();
    if (!hip_stream) fail(GrB_NULL_POINTER, "hip_stream is NULL");
    *hip_stream = (void *)ctx().stream;
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_synchronize(void)
{
    GRB_TRY
    require_init();
    sync_stream();
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_trim_memory(void)
{
    GRB_TRY
    require_init();
    sync_stream();
    dev_cache_release();
    sync_stream();
    int dev = 0;
    hipMemPool_t pool;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
    (void)hipGetLastError();
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_timer_start(void)
{
    GRB_TRY
    require_init();
    GRB_HIP(hipEventRecord(ctx().ev0, ctx().stream));
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_timer_stop(float *elapsed_ms)
{
    GRB_TRY
    require_init();
    if (!elapsed_ms) fail(GrB_NULL_POINTER, "elapsed_ms is NULL");
    GRB_HIP(hipEventRecord(ctx().ev1, ctx().stream));
    GRB_HIP(hipEventSynchronize(ctx().ev1));
    GRB_HIP(hipEventElapsedTime(elapsed_ms, ctx().ev0, ctx().ev1));
    GRB_CATCH(nullptr)
}

extern "C" GrB_Info GrX_last_stats(GrX_Stats *stats)
{
    if (!stats) return GrB_NULL_POINTER;
    *stats = ctx().stats;
    return GrB_SUCCESS;
}

extern "C" GrB_Info GrX_option_set(const char *name, int64_t value)
{
    if (!name) return GrB_NULL_POINTER;
    Context &c = ctx();
    const std::string n(name);
    if (n == "debug_flags") {
        if ((int)value & ~DEBUG_FLAGS_MASK) return GrB_INVALID_VALUE;  // kernel ablation switches exist in -DGRB_ABLATE builds only
        c.debug_flags = (int)value;
    }
    else if (n == "pull_ipt") c.tune_pull_ipt = (int)value;
    else if (n == "mxv_overlap") c.mxv_overlap = value ? 1 : 0;
    else if (n == "strip_wgs") c.strip_wgs = (int)std::max<int64_t>(0, std::min<int64_t>(value, 4096));
    else if (n == "hot_min_cols") c.hot_min_cols = value;
    else if (n == "hot_k") c.hot_k = value;
    else if (n == "push_mode") c.push_mode = (int)value;
    else if (n == "split_min_nnz") c.split_min_nnz = value;
    else if (n == "split_min_len") c.split_min_len = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (n == "short_kernel") {
        // (2 = sliced ELLPACK, 3 = persistent row groups with an LDS head, 4 = a lane per row: measured slower in rounds 1-2, removed)
        if (value < 0 || value > 6 || value == 2 || value == 3 || value == 4) return GrB_INVALID_VALUE;
        c.short_kernel = (int)value;
    }
    else if (n == "lazy_layout") c.lazy_layout = (int)value;
    else if (n == "lazy_min_nnz") c.lazy_min_nnz = value;
    else if (n == "lean_min_nnz") c.lean_min_nnz = value < 0 ? 0 : value;
    else if (n == "long_kernel") c.long_kernel = (int)value;
    else if (n == "long_classes") c.long_classes = (value == 16 || value == 32 || value == 64) ? (int)value : 8;
    else if (n == "long_sub") c.long_sub = (int)std::max<int64_t>(0, std::min<int64_t>(value, 16));
    else if (n == "long_sub_min_len") c.long_sub_min_len = (int)value;
    else if (n == "mxm_mask_mode") c.mxm_mask_mode = (int)value;
    else if (n == "mat_write_kernel") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.mat_write_kernel = (int)value;
    }
    else if (n == "mxm_heavy_kernel") c.mxm_heavy_kernel = (int)value;
    else if (n == "drop_hot_cols") c.drop_hot_cols = (int)value;
    else if (n == "mxm_unit_min_flops") c.mxm_unit_min_flops = value;
    else if (n == "mxm_unit_min_per_window") c.mxm_unit_min_per_window = value;
    else if (n == "mxm_masked_units_min_flops") c.mxm_masked_units_min_flops = value;
    else if (n == "mxm_unit_small" || n == "mxm_unit_dense" || n == "mxm_unit_mid") {  // class limits of the SpGEMM units: entry counts
        if (value < 1 || value > (1 << 30)) return GrB_INVALID_VALUE;
        (n == "mxm_unit_small" ? c.mxm_unit_small : n == "mxm_unit_dense" ? c.mxm_unit_dense : c.mxm_unit_mid) = (int)value;
    }
    else if (n == "mxm_sym_windows") {
        if (value < 1 || value > 64) return GrB_INVALID_VALUE;
        c.mxm_sym_windows = value;
    }
    else if (n == "mxm_window_groups") {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return GrB_INVALID_VALUE;
        c.mxm_window_groups = value;
    }
    else if (n == "mxm_xcd_map") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.mxm_xcd_map = (int)value;
    }
    else if (n == "mxm_checksum_pass") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.mxm_checksum_pass = (int)value;
    }
    else if (n == "mxm_bitmap_pool_mb") c.mxm_bitmap_pool_mb = value;
    else if (n == "mxm_bitmap_min_cnt") c.mxm_bitmap_min_cnt = (int)value;
    else if (n == "mxm_bitmap_pool_cap") c.mxm_bitmap_pool_cap = value;
    else if (n == "vec_pad_min_bytes") c.vec_pad_min_bytes = value;
    else if (n == "hub_min_len") c.hub_min_len = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (n == "fill_absent") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.fill_absent = (int)value;
    }
    else if (n == "rows_tile") {
        if (value != 0 && value != 1 && value != 2) return GrB_INVALID_VALUE;
        c.rows_tile = (int)value;
    }
    else if (n == "lazy_tagged") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.lazy_tagged = (int)value;
    }
    else if (n == "ctile_pack") {
        if (value < 0 || value > 2) return GrB_INVALID_VALUE;
        c.ctile_pack = (int)value;
    }
    else if (n == "strip_slot16") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.strip_slot16 = (int)value;
    }
    else if (n == "rtile_pack") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.rtile_pack = (int)value;
    }
    else if (n == "cold_in_rows") c.cold_in_rows = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (n == "stream_nt_min_nnz") c.stream_nt_min_nnz = value < 0 ? 0 : value;
    else if (n == "bool_probe") {
        if (value < 0 || value > 16) return GrB_INVALID_VALUE;
        c.bool_probe = (int)value;
    }
    else if (n == "rtile_rows") {
        if (value != 8192 && value != 16384) return GrB_INVALID_VALUE;
        c.rtile_rows = (int)value;
    }
    else if (n == "rtile_entries") {
        if (value < 256 || value > (1 << 24)) return GrB_INVALID_VALUE;
        c.rtile_entries = value;
    }
    else if (n == "rows_head") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.rows_head = (int)value;
    }
    else if (n == "rows_head_min_groups") c.rows_head_min_groups = value < 0 ? 0 : value;
    else if (n == "push_small") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.push_small = (int)value;
    }
    else if (n == "value_dict") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.value_dict = (int)value;
    }
    else if (n == "order_mode") {
        if (value != 0 && value != 1) return GrB_INVALID_VALUE;
        c.order_mode = (int)value;
    }
    else if (n == "order_min_nnz") c.order_min_nnz = value < 0 ? 0 : value;
    else if (n == "alloc_cache") {
        if (!value) dev_cache_release();
        c.alloc_cache = (int)value;
    }
    else return GrB_INVALID_VALUE;
    return GrB_SUCCESS;
}

extern "C" const char *GrX_version_string(void) { return "grb-mi355x 0.1 (gfx950, GraphBLAS C API 2.0 subset)"; }
