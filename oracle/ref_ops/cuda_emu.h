// CPU execution harness for the reference's CUDA kernels (TEST INFRASTRUCTURE, see oracle/ref_ops/build.sh).
//
// The FlowNet2 native ops of the reference (models/flownet2_pytorch/networks/{correlation,resample2d,channelnorm}_package)
// exist only as CUDA kernels behind ATen launchers; there is no nvcc, no NVIDIA GPU and no legacy ATen API here, and the
// correlation kernel hard-codes 32-lane warps (THREADS_PER_BLOCK 32, __shfl_down_sync over 16..1).  This header gives the
// kernel BODIES -- extracted at build time, unmodified, from the files where they lie under /root/reference -- a host
// execution model: the CUDA built-ins they use (threadIdx / blockIdx / blockDim, __shfl_down_sync, __syncthreads,
// __syncwarp, atomicAdd, long4, min / max) with one fiber (ucontext) per CUDA thread of a block, resumed round-robin, so
// that shuffles and barriers have their lock-step semantics with warpSize = 32.  Kernels without synchronisation run as
// a plain loop over the thread index.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

using std::floor; using std::sqrt; using std::min; using std::max;      // the float overloads CUDA resolves to

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct long4 { long x, y, z, w; };
static inline long4 make_long4(long a, long b, long c, long d) { long4 r = {a, b, c, d}; return r; }

static dim3 threadIdx, blockIdx, blockDim, gridDim;
static const int warpSize = 32;

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__

// ---- fibers -----------------------------------------------------------------------------------------------------
namespace emu {
constexpr int MAX_THREADS = 1024;
constexpr size_t STACK = 64 * 1024;
static ucontext_t sched_ctx;
static std::vector<ucontext_t> ctx;
static std::vector<char> stacks;
static std::vector<char> done;
static int cur = 0, nthreads = 0;
static std::function<void()> body;
alignas(16) static char xchg[MAX_THREADS][16];

static void trampoline() {
    body();
    done[cur] = 1;
    swapcontext(&ctx[cur], &sched_ctx);
}
static inline void yield() { const int me = cur; swapcontext(&ctx[me], &sched_ctx); }

// one CUDA block: `n` threads running `fn` in lock step between synchronisation points
static void run_block(int n, const std::function<void()>& fn) {
    if (n > MAX_THREADS) { fprintf(stderr, "cuda_emu: block of %d threads\n", n); abort(); }
    nthreads = n; body = fn;
    ctx.resize(n); done.assign(n, 0); stacks.resize((size_t)n * STACK);
    for (int t = 0; t < n; ++t) {
        getcontext(&ctx[t]);
        ctx[t].uc_stack.ss_sp = stacks.data() + (size_t)t * STACK;
        ctx[t].uc_stack.ss_size = STACK;
        ctx[t].uc_link = &sched_ctx;
        makecontext(&ctx[t], trampoline, 0);
    }
    for (bool live = true; live;) {
        live = false;
        for (int t = 0; t < n; ++t) {
            if (done[t]) continue;
            cur = t; threadIdx = dim3(t, 0, 0);
            swapcontext(&sched_ctx, &ctx[t]);
            live = live || !done[t];
        }
    }
}
}  // namespace emu

static inline void __syncthreads() { emu::yield(); }
static inline void __syncwarp(unsigned = 0xffffffffu) { emu::yield(); }

// CUDA: lane l receives the value of lane l + offset of its 32-lane warp; lanes whose source is outside keep their own
template <typename T>
static inline T __shfl_down_sync(unsigned, T val, int offset) {
    static_assert(sizeof(T) <= 16, "shuffle payload");
    const int me = emu::cur, lane = me % warpSize;
    memcpy(emu::xchg[me], &val, sizeof(T));
    emu::yield();                                  // every thread of the block has published
    T r = val;
    if (lane + offset < warpSize && me + offset < emu::nthreads) memcpy(&r, emu::xchg[me + offset], sizeof(T));
    emu::yield();                                  // every thread has read before anyone publishes again
    return r;
}

template <typename T, typename U>
static inline T atomicAdd(T* p, U v) { const T o = *p; *p = o + (T)v; return o; }

// launch of a kernel WITHOUT synchronisation points: grid x block loop, one call per CUDA thread
template <typename F>
static void launch_flat(dim3 grid, dim3 block, F&& kernel_call) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned t = 0; t < block.x; ++t) { threadIdx = dim3(t, 0, 0); kernel_call(); }
    }
}
// launch of a kernel WITH shuffles / barriers: one fiber per thread of each block
template <typename F>
static void launch_lockstep(dim3 grid, dim3 block, F&& kernel_call) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        emu::run_block((int)block.x, kernel_call);
    }
}
