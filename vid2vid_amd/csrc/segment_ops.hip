// Instance-wise average pooling of Encoder.forward (models/networks.py:621-632): every pixel of an instance receives the
// mean of the encoder output over that instance, per sample and channel.  The reference loops over np.unique(inst) on the
// host and gathers / scatters with index tensors; here it is a segmented mean keyed by the (arbitrary, sparse) integer
// instance ids: ids -> slots of a small open-addressing table (atomicCAS), per-(slot, channel) sums accumulated through an
// LDS copy of the table (one block = one channel x a run of pixels, so the float atomics mostly stay in LDS), then one
// gather pass.  HBM-bound: feat is read twice, out written once.  First-frame path only (amortised over a sequence).
// The float atomics make the summation ORDER run-dependent (differences at the 1e-7 level); at most
// V2V_INSTANCE_SLOTS / 2 distinct ids per sample keep the probing short.
// Parity on the GPU: tests/test_gpu_golden.py::test_feature_encoding_first_frame_nets_vs_reference (Encoder.forward vs the reference).
#include "v2v_internal.h"

namespace v2v {

static const int SEG_SLOTS = 4096;                 // power of two
static const int SEG_EMPTY = (int)0x80000000;      // no instance map holds INT_MIN

struct SegArgs {
    const float* feat; const float* inst; float* out;
    int* keys; float* sums; int* slot;             // keys[SLOTS], sums[SLOTS][C + 1] (last = count), slot[HW]
    int C; long long HW;
};

__global__ __launch_bounds__(256) void seg_init_kernel(const SegArgs a) {
    const long long n = (long long)SEG_SLOTS * (a.C + 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        a.sums[i] = 0.f;
        if (i < SEG_SLOTS) a.keys[i] = SEG_EMPTY;
    }
}

__global__ __launch_bounds__(256) void seg_assign_kernel(const SegArgs a) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < a.HW; p += (long long)gridDim.x * 256) {
        const int id = (int)a.inst[p];
        unsigned h = ((unsigned)id * 2654435761u) & (SEG_SLOTS - 1);
        int s = -1;
        for (int probe = 0; probe < SEG_SLOTS; ++probe) {
            const int old = atomicCAS(&a.keys[h], SEG_EMPTY, id);
            if (old == SEG_EMPTY || old == id) { s = (int)h; break; }
            h = (h + 1) & (SEG_SLOTS - 1);
        }
        a.slot[p] = s;                              // -1: table full (more than SEG_SLOTS ids): the pixel keeps its own value
    }
}

// grid (pixel runs, C + 1): blockIdx.y < C accumulates channel y, blockIdx.y == C counts pixels
__global__ __launch_bounds__(256) void seg_accum_kernel(const SegArgs a, long long run) {
    __shared__ float acc[SEG_SLOTS];
    for (int i = threadIdx.x; i < SEG_SLOTS; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int c = blockIdx.y;
    const long long p0 = (long long)blockIdx.x * run;
    long long p1 = p0 + run; if (p1 > a.HW) p1 = a.HW;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int s = a.slot[p];
        if (s >= 0) atomicAdd(&acc[s], c < a.C ? a.feat[(long long)c * a.HW + p] : 1.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEG_SLOTS; i += 256)
        if (acc[i] != 0.f) atomicAdd(&a.sums[(long long)i * (a.C + 1) + c], acc[i]);
}

__global__ __launch_bounds__(256) void seg_gather_kernel(const SegArgs a) {
    const long long n = (long long)a.C * a.HW;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e / a.HW);
        const long long p = e - (long long)c * a.HW;
        const int s = a.slot[p];
        a.out[e] = s >= 0 ? a.sums[(long long)s * (a.C + 1) + c] / a.sums[(long long)s * (a.C + 1) + a.C] : a.feat[e];
    }
}

struct SegOp : Op {
    SegArgs a;
    int launch(hipStream_t s) override {
        auto blocks = [](long long n) { long long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); };
        hipLaunchKernelGGL(seg_init_kernel, dim3(blocks((long long)SEG_SLOTS * (a.C + 1))), dim3(256), 0, s, a);
        hipLaunchKernelGGL(seg_assign_kernel, dim3(blocks(a.HW)), dim3(256), 0, s, a);
        const long long run = 8192;
        hipLaunchKernelGGL(seg_accum_kernel, dim3((unsigned)ceil_div(a.HW, run), (unsigned)(a.C + 1)), dim3(256), 0, s, a, run);
        hipLaunchKernelGGL(seg_gather_kernel, dim3(blocks((long long)a.C * a.HW)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "instance_mean"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int64_t v2v_instance_mean_workspace(int32_t C, int64_t HW) {
    if (C < 1 || HW < 1) return V2V_EINVAL;
    return (int64_t)SEG_SLOTS * 4 + (int64_t)SEG_SLOTS * (C + 1) * 4 + HW * 4;
}

extern "C" int v2v_instance_mean_planar(const float* feat, const float* inst, float* out, void* workspace,
                                        int32_t C, int64_t HW, void* stream) {
    if (!feat || !inst || !out || !workspace || C < 1 || HW < 1 || ((uintptr_t)workspace & 3)) {
        set_error("instance_mean: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<SegOp>();
    char* ws = (char*)workspace;
    op->a = SegArgs{feat, inst, out, (int*)ws, (float*)(ws + (size_t)SEG_SLOTS * 4),
                    (int*)(ws + (size_t)SEG_SLOTS * 4 + (size_t)SEG_SLOTS * (C + 1) * 4), C, (long long)HW};
    return submit(std::move(op), stream);
}
