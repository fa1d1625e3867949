// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/hip/hip_runtime.h (see that header).
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <chrono>
#include <cstdio>
#include <vector>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;  // the emulated device
    return hipSuccess;
}

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    *(double *)e = now_ms();
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = (float)(*(double *)b - *(double *)a);
    return hipSuccess;
}

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

enum WaitKind { W_NONE, W_BLOCK, W_WAVE };

struct WaveState {
    int active = 0, arrived = 0;
    unsigned gen = 0;
    uint64_t payload[2][WAVE];
    uint64_t pred[2];
    uint64_t valid[2];
};

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    unsigned tid = 0;
    bool done = true;
    WaitKind wait = W_NONE;
    unsigned wait_gen = 0;
    unsigned wave_ops = 0;
};

static std::vector<Fiber> g_fibers;
static std::vector<WaveState> g_waves;
static ucontext_t g_sched;
static Fiber *g_cur = nullptr;
static const std::function<void()> *g_body = nullptr;
static int g_block_active = 0, g_block_arrived = 0;
static unsigned g_block_gen = 0;

int lane_id() { return (int)(g_cur->tid % WAVE); }

static void yield_to_scheduler() { swapcontext(&g_cur->ctx, &g_sched); }

static void release_block_if_complete()
{
    if (g_block_arrived > 0 && g_block_arrived == g_block_active) {
        g_block_arrived = 0;
        g_block_gen++;
    }
}
static void release_wave_if_complete(WaveState &w)
{
    if (w.arrived > 0 && w.arrived == w.active) {
        w.arrived = 0;
        w.gen++;
    }
}

void block_barrier()
{
    Fiber *f = g_cur;
    const unsigned g = g_block_gen;
    g_block_arrived++;
    release_block_if_complete();
    if (g_block_gen == g) {
        f->wait = W_BLOCK;
        f->wait_gen = g;
        yield_to_scheduler();
        f->wait = W_NONE;
    }
}

uint64_t wave_exchange(uint64_t payload, int src, bool pred, uint64_t *ballot)
{
    Fiber *f = g_cur;
    WaveState &w = g_waves[f->tid / WAVE];
    const int lane = (int)(f->tid % WAVE);
    const int par = (int)(f->wave_ops++ & 1u);
    if (w.arrived == 0) {  // first arrival of this operation resets its buffers
        w.pred[par] = 0;
        w.valid[par] = 0;
    }
    w.payload[par][lane] = payload;
    w.valid[par] |= 1ull << lane;
    if (pred) w.pred[par] |= 1ull << lane;
    const unsigned g = w.gen;
    w.arrived++;
    release_wave_if_complete(w);
    if (w.gen == g) {
        f->wait = W_WAVE;
        f->wait_gen = g;
        yield_to_scheduler();
        f->wait = W_NONE;
    }
    if (ballot) *ballot = w.pred[par];
    if (src >= 0 && src < WAVE && ((w.valid[par] >> src) & 1ull)) return w.payload[par][src];
    return payload;
}

static void fiber_entry()
{
    (*g_body)();
    Fiber *f = g_cur;
    f->done = true;
    g_block_active--;
    WaveState &w = g_waves[f->tid / WAVE];
    w.active--;
    release_block_if_complete();
    release_wave_if_complete(w);
    swapcontext(&f->ctx, &g_sched);
}

static bool runnable(const Fiber &f)
{
    if (f.done) return false;
    if (f.wait == W_BLOCK) return g_block_gen != f.wait_gen;
    if (f.wait == W_WAVE) return g_waves[f.tid / WAVE].gen != f.wait_gen;
    return true;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body)
{
    const unsigned nthreads = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "emu: only 1-D launches are supported\n");
        abort();
    }
    if (grid.x == 0 || nthreads == 0 || nthreads > 1024) {
        // the hardware rejects these launches ("invalid configuration argument") and the error surfaces at a later call:
        // fail at the source instead
        fprintf(stderr, "emu: invalid launch configuration (grid %u, block %u)\n", grid.x, nthreads);
        abort();
    }
    if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    for (unsigned t = 0; t < nthreads; t++)
        if (!g_fibers[t].stack) g_fibers[t].stack = (char *)malloc(STACK_BYTES);
    const unsigned nwaves = (nthreads + WAVE - 1) / WAVE;
    if (g_waves.size() < nwaves) g_waves.resize(nwaves);
    blockDim = block;
    gridDim = grid;
    g_body = &body;
    for (unsigned b = 0; b < grid.x; b++) {
        g_block_active = (int)nthreads;
        g_block_arrived = 0;
        for (unsigned wv = 0; wv < nwaves; wv++) {
            g_waves[wv].active = (int)std::min<unsigned>(WAVE, nthreads - wv * WAVE);
            g_waves[wv].arrived = 0;
        }
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber &f = g_fibers[t];
            f.tid = t;
            f.done = false;
            f.wait = W_NONE;
            f.wave_ops = 0;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK_BYTES;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, fiber_entry, 0);
        }
        unsigned remaining = nthreads;
        while (remaining) {
            bool progress = false;
            for (unsigned t = 0; t < nthreads; t++) {
                Fiber &f = g_fibers[t];
                if (!runnable(f)) continue;
                g_cur = &f;
                threadIdx.x = t; threadIdx.y = 0; threadIdx.z = 0;
                blockIdx.x = b; blockIdx.y = 0; blockIdx.z = 0;
                swapcontext(&g_sched, &f.ctx);
                progress = true;
                if (f.done) remaining--;
            }
            if (!progress) {
                fprintf(stderr, "emu: deadlock in block %u -- a barrier or wave intrinsic was reached by only part of "
                                "a workgroup/wavefront (divergent __syncthreads/__shfl/__ballot)\n", b);
                abort();
            }
        }
    }
    g_cur = nullptr;
    g_body = nullptr;
}

}  // namespace emu
