// Micro-benchmark (round 6): what does a SKIPPED piece of a stream cost on MI355X?  A 4 GiB buffer is read as pieces of G bytes
// (G = 32 ... 4096); every piece is either read (16-byte loads, lanes contiguous inside the piece) or skipped according to a
// pseudo-random keep bit with probability p.  If the memory system moves whole 128-byte lines, G = 64 with p = 0.5 costs what the
// full stream costs; if it moves 64-byte sectors it costs half.  Prints one JSON line per (G, p, aux).
// Build: hipcc -O3 --offload-arch=gfx950 skip_rates.hip -o skip_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one thread = one 16-byte load per step; piece index = (byte offset / G); keep(piece) = hash(piece) < thr
template <int AUX>
__global__ __launch_bounds__(256) void k_skip(const char *buf, int64_t n16, unsigned gshift, uint32_t thr, float *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, (int)0xfffffff0u, 0x00020000);
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += 4 * stride) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t j = i + k * stride;
            const uint32_t off = (uint32_t)(j << 4);
            const uint32_t piece = off >> gshift;
            const bool keep = j < n16 && hash32(piece) <= thr;
            // predication through the address: an offset beyond the descriptor moves nothing
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, keep ? off : 0xfffffff8u, 0, AUX);
            acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[3]);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const int64_t bytes = (int64_t)0xfffffff0u & ~4095ll;  // just under 4 GiB: one buffer descriptor
    char *buf;
    float *out;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int64_t n16 = bytes / 16;
    const double ps[] = {1.0, 0.5, 0.25};
    for (int aux = 0; aux < 2; aux++)
        for (unsigned gs = 5; gs <= 12; gs++)
            for (double p : ps) {
                const uint32_t thr = p >= 1.0 ? 0xffffffffu : (uint32_t)(p * 4294967296.0);
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0));
                    if (aux == 0) k_skip<0><<<256 * 16, 256>>>(buf, n16, gs, thr, out);
                    else k_skip<2><<<256 * 16, 256>>>(buf, n16, gs, thr, out);  // 2 = nt (slc)
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("{\"piece_bytes\": %u, \"keep\": %.2f, \"nt\": %d, \"ms\": %.4f, \"kept_GBps\": %.1f, \"span_GBps\": %.1f}\n", 1u << gs, p,
                       aux, best, bytes * p / best * 1e-6, bytes / best * 1e-6);
                fflush(stdout);
            }
    return 0;
}
