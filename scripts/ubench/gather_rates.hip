// Micro-benchmarks that price the design choices of the pull SpMV on MI355X (scripts/ubench/README in scripts/README.md):
//   gather   -- random 4-byte / 2-byte gathers from a table of S bytes (TA path: L2 / MALL / HBM by size), G independent
//               gathers in flight per lane, `share` adjacent lanes reading the same 128-byte line
//   lds      -- random ds_read_b32 gathers from a 128 KiB LDS image, 1024-thread workgroups
//   dispatch -- empty-ish kernel with many small workgroups (workgroup dispatch rate)
// Prints one JSON line per measurement.  Build: hipcc -O3 --offload-arch=gfx950 gather_rates.hip -o gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// idx stream is precomputed (coalesced 16-byte loads, like the entry stream of the SpMV): 4 indices per lane per step
template <typename T, int AUX>
__global__ __launch_bounds__(256) void k_gather(const uint4 *idx, int64_t n4, const T *table, unsigned tbytes, float *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(table), 0, (int)tbytes, 0x00020000);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 c = idx[i];
        T v0, v1, v2, v3;
        if constexpr (sizeof(T) == 4) {
            v0 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, c.x * 4u, 0, AUX));
            v1 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, c.y * 4u, 0, AUX));
            v2 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, c.z * 4u, 0, AUX));
            v3 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, c.w * 4u, 0, AUX));
        } else {
            v0 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, c.x * 2u, 0, AUX));
            v1 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, c.y * 2u, 0, AUX));
            v2 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, c.z * 2u, 0, AUX));
            v3 = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, c.w * 2u, 0, AUX));
        }
        acc += (float)v0 + (float)v1 + (float)v2 + (float)v3;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// indices: `share` adjacent lanes fall into one 128-byte line of the table (entries of `esz` bytes)
__global__ void k_make_idx(uint32_t *idx, int64_t n, uint32_t entries, int share, int esz, uint32_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // element i = 4 * lane_slot + j: the j-th gather of a lane; lanes share lines per j
    const int64_t lane_slot = i >> 2;
    const int j = (int)(i & 3);
    const int64_t grp = lane_slot / share;
    uint32_t s = (uint32_t)(grp * 2654435761u) ^ (seed + 0x9e3779b9u * (uint32_t)j) ^ (uint32_t)(grp >> 20);
    rng(s); rng(s);
    const uint32_t per_line = 128u / (uint32_t)esz;
    const uint32_t lines = entries / per_line;
    const uint32_t line = rng(s) % lines;
    uint32_t s2 = (uint32_t)i * 747796405u + 2891336453u;
    rng(s2);
    idx[i] = line * per_line + (rng(s2) % per_line);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_lds(const uint4 *idx, int64_t n4, const uint32_t *table, float *out)
{
    __shared__ uint32_t s_x[32768];
    for (int k = threadIdx.x; k < 32768; k += WAVES * 64) s_x[k] = table[k];
    __syncthreads();
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 c = idx[i];
        acc += __builtin_bit_cast(float, s_x[c.x & 32767]) + __builtin_bit_cast(float, s_x[c.y & 32767]) +
               __builtin_bit_cast(float, s_x[c.z & 32767]) + __builtin_bit_cast(float, s_x[c.w & 32767]);
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(128) void k_small(const uint64_t *in, uint64_t *outp, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) outp[i] = in[i] + 1;
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int64_t n = 1ll << 27;  // 134 M gathers per launch (the masked bench call does 132 M)
    uint32_t *idx; float *out; char *table;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&table, 256ll << 20));
    CK(hipMemset(table, 0, 256ll << 20));
    const int blocks = 256 * 8;
    const size_t sizes[] = {128u << 10, 1u << 20, 2u << 20, 4u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20};
    for (int esz : {4, 2}) {
        for (size_t sb : sizes) {
            for (int share : {1, 2, 4, 16}) {
                if (share > 1 && sb != (2u << 20) && sb != (64u << 20)) continue;
                const uint32_t entries = (uint32_t)(sb / esz);
                hipLaunchKernelGGL(k_make_idx, dim3((unsigned)(n / 256)), dim3(256), 0, 0, idx, n, entries, share, esz, 17u);
                float ms;
                if (esz == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 0>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, (unsigned)sb, out); }, 3);
                else ms = time_ms([&] { hipLaunchKernelGGL((k_gather<uint16_t, 0>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const uint16_t *)table, (unsigned)sb, out); }, 3);
                printf("{\"bench\": \"gather\", \"elem_bytes\": %d, \"table_bytes\": %zu, \"lanes_per_line\": %d, \"ms\": %.4f, \"us_per_Mgather\": %.3f}\n",
                       esz, sb, share, ms, ms * 1e3 / (n / 1e6));
                fflush(stdout);
            }
        }
    }
    // cache-policy bits on a table that misses L2 (64 MiB) and one that hits (2 MiB)
    for (size_t sb : {size_t(2u << 20), size_t(64u << 20)}) {
        hipLaunchKernelGGL(k_make_idx, dim3((unsigned)(n / 256)), dim3(256), 0, 0, idx, n, (uint32_t)(sb / 4), 1, 4, 17u);
        float m1 = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 1>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, (unsigned)sb, out); }, 3);
        float m2 = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 2>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, (unsigned)sb, out); }, 3);
        float m16 = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 16>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, (unsigned)sb, out); }, 3);
        float m17 = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 17>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, (unsigned)sb, out); }, 3);
        printf("{\"bench\": \"gather_aux\", \"table_bytes\": %zu, \"ms_sc0\": %.4f, \"ms_nt\": %.4f, \"ms_sc1\": %.4f, \"ms_sc0sc1\": %.4f}\n", sb, m1, m2, m16, m17);
    }
    // LDS gathers: one workgroup per CU (16 waves) and two (2 x 8 waves would need 2 x 128 KiB: so 16 waves only) + 4-wave groups
    hipLaunchKernelGGL(k_make_idx, dim3((unsigned)(n / 256)), dim3(256), 0, 0, idx, n, 32768u, 1, 4, 17u);
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_lds<16>), dim3(256), dim3(1024), 0, 0, (const uint4 *)idx, n / 4, (const uint32_t *)table, out); }, 3);
        printf("{\"bench\": \"lds_gather\", \"waves_per_cu\": 16, \"ms\": %.4f, \"us_per_Mgather\": %.3f}\n", ms, ms * 1e3 / (n / 1e6));
        ms = time_ms([&] { hipLaunchKernelGGL((k_lds<8>), dim3(256), dim3(512), 0, 0, (const uint4 *)idx, n / 4, (const uint32_t *)table, out); }, 3);
        printf("{\"bench\": \"lds_gather\", \"waves_per_cu\": 8, \"ms\": %.4f, \"us_per_Mgather\": %.3f}\n", ms, ms * 1e3 / (n / 1e6));
    }
    // stream only (the idx loads alone): the floor of the gather kernels
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_gather<float, 0>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)idx, n / 4, (const float *)table, 0u, out); }, 3);
        printf("{\"bench\": \"stream_only\", \"bytes\": %lld, \"ms\": %.4f, \"GBps\": %.1f}\n", (long long)n * 4, ms, n * 4 / ms / 1e6);
    }
    // workgroup dispatch rate: 16.7 M threads in 128-thread workgroups, 8 bytes in, 8 bytes out per thread
    {
        const int64_t m = 1ll << 24;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_small, dim3((unsigned)(m / 128)), dim3(128), 0, 0, (const uint64_t *)table, (uint64_t *)(table + (128ll << 20)), m); }, 5);
        printf("{\"bench\": \"dispatch_128\", \"workgroups\": %lld, \"ms\": %.4f, \"wg_per_us\": %.1f, \"GBps\": %.1f}\n", (long long)(m / 128), ms, (m / 128) / (ms * 1e3), m * 16 / ms / 1e6);
    }
    return 0;
}
