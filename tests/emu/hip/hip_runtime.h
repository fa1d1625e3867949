// TEST INFRASTRUCTURE ONLY -- a wave64 SIMT emulator standing in for <hip/hip_runtime.h>.
//
// tests/emu/build_emu.sh compiles the UNMODIFIED kernel sources of python-graphblas_amd/csrc with
// g++ against this header, producing tests/emu/libgrb_emu.so.  Every GPU thread becomes a fiber
// (ucontext); __syncthreads / __shfl_down / __ballot are rendezvous points across the fibers of a
// workgroup / 64-lane wavefront, so the kernels' control flow, LDS use, wave intrinsics and
// merge-path logic execute exactly as written -- only slowly and on the CPU.  It exists so that the
// CPU-only test tier (-m "not gpu") can check kernel LOGIC against the oracle before a GPU run.
// The product never loads it: python-graphblas_amd/_lib.py only opens libgrb_mi355x.so and GrB_init
// fails without a HIP device.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct ulonglong2 { unsigned long long x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

// ---- runtime API subset --------------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorOutOfMemory = 2;
typedef struct emu_stream_t *hipStream_t;
typedef struct emu_event_t *hipEvent_t;
typedef struct emu_pool_t *hipMemPool_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipMemPoolAttr { hipMemPoolAttrReleaseThreshold };

inline const char *hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n);
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }  // few CUs: persistent loops iterate
inline hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t) { memset(p, v, bytes); return hipSuccess; }
struct hipFuncAttributes { int numRegs; };
inline hipError_t hipFuncGetAttributes(hipFuncAttributes *, const void *) { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)1 << 30; *total_b = (size_t)4 << 30; return hipSuccess; }  // (the emulator: a small pool, so that its exhaustion is exercised too)
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t bytes, hipMemcpyKind, hipStream_t) { memmove(d, s, bytes); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t *p, int) { *p = nullptr; return hipSuccess; }
inline hipError_t hipMemPoolSetAttribute(hipMemPool_t, hipMemPoolAttr, void *) { return hipSuccess; }
inline hipError_t hipMemPoolTrimTo(hipMemPool_t, size_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
// (round 6: a second stream -- the emulator runs every launch to completion at once, so streams and the events between them are no-ops)
constexpr unsigned hipStreamNonBlocking = 1u, hipEventDisableTiming = 2u;
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t);
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

// ---- SIMT execution ----------------------------------------------------------------------------------------
namespace emu {
void run_grid(dim3 grid, dim3 block, const std::function<void()> &thread_body);
void block_barrier();
// all-lanes exchange inside the calling fiber's 64-lane wavefront; returns the payload lane `src`
// published (own payload if `src` is out of range or that lane has exited); *ballot = OR of 1<<lane
// over participating lanes whose `pred` is set
uint64_t wave_exchange(uint64_t payload, int src, bool pred, uint64_t *ballot);
int lane_id();
}  // namespace emu

template <typename... KArgs, typename... Args>
inline void emu_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args &&...args)
{
    std::function<void()> body = [&]() { kernel(args...); };
    emu::run_grid(grid, block, body);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { emu::block_barrier(); }
inline void __threadfence() {}

template <typename T>
inline T __shfl(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    uint64_t r = emu::wave_exchange(bits, src, false, nullptr);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T> inline T __shfl_down(T v, unsigned delta) { return __shfl(v, emu::lane_id() + (int)delta); }
template <typename T> inline T __shfl_up(T v, unsigned delta) { return __shfl(v, emu::lane_id() - (int)delta); }
template <typename T> inline T __shfl_xor(T v, int mask) { return __shfl(v, emu::lane_id() ^ mask); }
inline unsigned long long __ballot(int pred)
{
    uint64_t b = 0;
    emu::wave_exchange(0, -1, pred != 0, &b);
    return b;
}
inline int __any(int pred) { return __ballot(pred) != 0; }

inline long long clock64() { return 0; }
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_fence(int, const char *) {}
inline void __builtin_amdgcn_wave_barrier() { emu::wave_exchange(0, -1, false, nullptr); }  // fibers of the wavefront rendezvous
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // callers pass wave-uniform values
inline int __builtin_amdgcn_readlane(int v, int src_lane)  // (every lane of the wavefront calls, with the same lane number)
{
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(int));
    const uint64_t r = emu::wave_exchange(bits, src_lane, false, nullptr);
    int got;
    memcpy(&got, &r, sizeof(int));
    return got;
}
// DPP data movement inside a wavefront, the controls the kernels use: row_shr:n (0x110 + n: from lane - n of the same row of 16
// lanes), wave_shr:1 (0x138: from lane - 1, across the rows), row_bcast:15 (0x142: lane 15 of the previous row) and row_bcast:31
// (0x143: lane 31, rows 2 and 3); a lane whose row is
// not in row_mask, or whose source does not exist, keeps `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int, bool)
{
    const int l = emu::lane_id(), row = l >> 4;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) {
        const int n = ctrl - 0x110;
        if ((l & 15) >= n) from = l - n;
    } else if (ctrl == 0x138) {  // wave_shr:1
        if (l >= 1) from = l - 1;
    } else if (ctrl == 0x142) {
        if (row >= 1) from = row * 16 - 1;
    } else if (ctrl == 0x143) {
        if (row >= 2) from = 31;
    }
    uint64_t bits = 0;
    memcpy(&bits, &src, sizeof(int));
    const uint64_t r = emu::wave_exchange(bits, from >= 0 ? from : l, false, nullptr);  // (every lane takes part in the exchange)
    int got;
    memcpy(&got, &r, sizeof(int));
    return (from >= 0 && ((row_mask >> row) & 1)) ? got : old;
}
// raw buffer loads: out-of-range offsets return 0 (per dword), like the hardware range check with stride 0
struct __amdgpu_buffer_rsrc_t { const char *base; unsigned n; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short, int n, int) { return {(const char *)p, (unsigned)n}; }
template <typename T> inline T emu_buf_load(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    T v{};
    if ((unsigned long long)off + sizeof(T) <= r.n) memcpy(&v, r.base + off, sizeof(T));
    return v;
}
typedef unsigned emu_u32x4 __attribute__((vector_size(16)));
typedef unsigned emu_u32x2 __attribute__((vector_size(8)));
inline unsigned char __builtin_amdgcn_raw_buffer_load_b8(__amdgpu_buffer_rsrc_t r, unsigned off, int, int) { return emu_buf_load<unsigned char>(r, off); }
inline unsigned short __builtin_amdgcn_raw_buffer_load_b16(__amdgpu_buffer_rsrc_t r, unsigned off, int, int) { return emu_buf_load<unsigned short>(r, off); }
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned off, int, int) { return emu_buf_load<unsigned>(r, off); }
inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned off, int, int)
{
    emu_u32x2 v = {emu_buf_load<unsigned>(r, off), emu_buf_load<unsigned>(r, off + 4)};
    return v;
}
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned off, int, int)
{
    emu_u32x4 v = {emu_buf_load<unsigned>(r, off), emu_buf_load<unsigned>(r, off + 4), emu_buf_load<unsigned>(r, off + 8), emu_buf_load<unsigned>(r, off + 12)};
    return v;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

// fibers interleave only at rendezvous points, so plain read-modify-write is atomic here
template <typename T> inline T emu_rmw_add(T *p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return emu_rmw_add(p, v); }
inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return emu_rmw_add(p, v); }
inline int atomicAdd(int *p, int v) { return emu_rmw_add(p, v); }
inline float atomicAdd(float *p, float v) { return emu_rmw_add(p, v); }
inline double atomicAdd(double *p, double v) { return emu_rmw_add(p, v); }
inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { auto o = *p; *p = o | v; return o; }
inline unsigned int atomicOr(unsigned int *p, unsigned int v) { auto o = *p; *p = o | v; return o; }
inline unsigned long long atomicAnd(unsigned long long *p, unsigned long long v) { auto o = *p; *p = o & v; return o; }
inline unsigned int atomicAnd(unsigned int *p, unsigned int v) { auto o = *p; *p = o & v; return o; }
inline unsigned int atomicCAS(unsigned int *p, unsigned int cmp, unsigned int val) { auto o = *p; if (o == cmp) *p = val; return o; }
inline int atomicCAS(int *p, int cmp, int val) { auto o = *p; if (o == cmp) *p = val; return o; }
inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long val) { auto o = *p; if (o == cmp) *p = val; return o; }
inline unsigned int atomicExch(unsigned int *p, unsigned int v) { auto o = *p; *p = v; return o; }
inline int atomicMax(int *p, int v) { auto o = *p; if (v > o) *p = v; return o; }
inline unsigned int atomicMax(unsigned int *p, unsigned int v) { auto o = *p; if (v > o) *p = v; return o; }
inline int atomicMin(int *p, int v) { auto o = *p; if (v < o) *p = v; return o; }
inline unsigned int atomicMin(unsigned int *p, unsigned int v) { auto o = *p; if (v < o) *p = v; return o; }
inline long long atomicMin(long long *p, long long v) { auto o = *p; if (v < o) *p = v; return o; }
inline long long atomicMax(long long *p, long long v) { auto o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { auto o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { auto o = *p; if (v > o) *p = v; return o; }
inline float atomicMin(float *p, float v) { auto o = *p; if (v < o || o != o) *p = v; return o; }
inline float atomicMax(float *p, float v) { auto o = *p; if (v > o || o != o) *p = v; return o; }
inline double atomicMin(double *p, double v) { auto o = *p; if (v < o || o != o) *p = v; return o; }
inline double atomicMax(double *p, double v) { auto o = *p; if (v > o || o != o) *p = v; return o; }
inline unsigned int atomicXor(unsigned int *p, unsigned int v) { auto o = *p; *p = o ^ v; return o; }
inline unsigned long long atomicXor(unsigned long long *p, unsigned long long v) { auto o = *p; *p = o ^ v; return o; }
