// Loss reductions and the fused optimizer step of the vid2vid training path (gfx950).
//
//   GANLoss (LSGAN = MSE against a constant, models/networks.py:731-774)
//   criterionFeat = nn.L1Loss between discriminator features (models/vid2vid_model_D.py:65,208-211)
//   MaskedL1Loss (models/networks.py:804-812): mean |a*m - b*m| with the mask broadcast over channels
//   torch.optim.Adam (models/vid2vid_model_G.py:84, vid2vid_model_D.py:86-91) as ONE launch over the
//   flat parameter / gradient / moment buffers of an optimizer.
//
// All are HBM-bound streaming kernels.  Reductions are deterministic: fixed per-block partials,
// then one block combines them in order (fp64) and writes the scaled scalar.
#include "v2v_internal.h"
#include <cstring>

namespace v2v {

enum { LOSS_MSE_CONST = 0, LOSS_L1 = 1 };

struct LossArgs {
    const void* a; const void* b; const float* mask;   // mask: planar [N][1][HW] (planar mode only) or NULL
    float target; int kind;
    long long P; int C, c_stride;                      // NHWC mode: P pixels, C real channels
    long long n, chw, hw;                              // planar mode: n elements, C*HW, HW
    int planar;
    float* partials; int nblk;
    float* out; double scale;                          // out[0] = scale * sum
    const float* gout; void* da;                       // backward: da = d(out)/d(a) * gout[0]
};

template <typename T>
__device__ __forceinline__ float loss_term(const LossArgs& a, long long idx, float& av, float& bv, float& mv, bool& valid) {
    valid = true; mv = 1.f; bv = a.target;
    if (a.planar) {
        av = reinterpret_cast<const float*>(a.a)[idx];
        if (a.kind == LOSS_L1) bv = reinterpret_cast<const float*>(a.b)[idx];
        if (a.mask) { const long long n = idx / a.chw; mv = a.mask[n * a.hw + (idx % a.hw)]; }
    } else {
        const int c = (int)(idx % a.c_stride);
        valid = c < a.C;
        av = load_act(reinterpret_cast<const T*>(a.a), idx);
        if (a.kind == LOSS_L1) bv = load_act(reinterpret_cast<const T*>(a.b), idx);
    }
    if (!valid) return 0.f;
    if (a.kind == LOSS_MSE_CONST) { const float d = av - bv; return d * d; }
    return fabsf(av * mv - bv * mv);
}

template <typename T>
__global__ __launch_bounds__(256) void loss_partial_kernel(const LossArgs a) {
    __shared__ float sh[256];
    const long long total = a.planar ? a.n : a.P * a.c_stride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        float av, bv, mv; bool valid;
        s += loss_term<T>(a, e, av, bv, mv, valid);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partials[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const LossArgs a) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nblk; i += 256) s += (double)a.partials[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[0] = (float)(sh[0] * a.scale);
}

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossArgs a) {
    const long long total = a.planar ? a.n : a.P * a.c_stride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float g = a.gout[0] * (float)a.scale;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        float av, bv, mv; bool valid;
        loss_term<T>(a, e, av, bv, mv, valid);
        float d = 0.f;
        if (valid) {
            if (a.kind == LOSS_MSE_CONST) d = 2.f * (av - bv) * g;
            else { const float df = av * mv - bv * mv; d = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * mv * g; }
        }
        if (a.planar) reinterpret_cast<float*>(a.da)[e] = d;
        else store_act(reinterpret_cast<T*>(a.da), e, d);
    }
}

static unsigned loss_grid(long long total) {
    long long b = ceil_div(total, 256 * 8);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (unsigned)b;
}

struct LossOp : Op {
    LossArgs a; int dtype; bool backward;
    int launch(hipStream_t s) override {
        const long long total = a.planar ? a.n : a.P * a.c_stride;
        if (backward) {
            long long b = ceil_div(total, 256); if (b > 4096) b = 4096; if (b < 1) b = 1;
            if (dtype == V2V_BF16 && !a.planar) hipLaunchKernelGGL(loss_bwd_kernel<bf16_t>, dim3((unsigned)b), dim3(256), 0, s, a);
            else                                 hipLaunchKernelGGL(loss_bwd_kernel<float>, dim3((unsigned)b), dim3(256), 0, s, a);
            return check_launch();
        }
        if (dtype == V2V_BF16 && !a.planar) hipLaunchKernelGGL(loss_partial_kernel<bf16_t>, dim3((unsigned)a.nblk), dim3(256), 0, s, a);
        else                                 hipLaunchKernelGGL(loss_partial_kernel<float>, dim3((unsigned)a.nblk), dim3(256), 0, s, a);
        int rc = check_launch(); if (rc) return rc;
        hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return backward ? "loss_backward" : "loss_forward"; }
};

// ---- Adam ------------------------------------------------------------------------------
struct AdamArgs { float* p; const float* g; float* m; float* v; long long n; float lr, b1, b2, eps, bc1, bc2_sqrt, wd, gscale; };

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float step_size = a.lr / a.bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float g = a.g[i] * a.gscale;
        const float p = a.p[i];
        if (a.wd != 0.f) g += a.wd * p;
        const float m = a.b1 * a.m[i] + (1.f - a.b1) * g;
        const float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
        a.m[i] = m; a.v[i] = v;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.p[i] = p - step_size * (m / denom);
    }
}

// Graph-replayable form (optim.FusedAdam(capturable)): the 1-based step count and the learning rate live in DEVICE memory, so a
// launch captured once into a hipGraph keeps advancing the bias corrections on every replay (a by-value `step` would be frozen
// at its capture-time value).  state = {int step; float lr}: one thread bumps `step` in a launch of its own just before.
struct AdamDevState { int step; float lr; };
struct AdamDevArgs { float* p; const float* g; float* m; float* v; long long n; float b1, b2, eps, wd, gscale; AdamDevState* st; };

__global__ void adam_tick_kernel(AdamDevState* st) { st->step += 1; }

__global__ __launch_bounds__(256) void adam_step_dev_kernel(const AdamDevArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int step = a.st->step;
    const double bc1 = 1.0 - pow((double)a.b1, (double)step);        // the host form's arithmetic (v2v_adam_step), once per thread
    const double bc2 = 1.0 - pow((double)a.b2, (double)step);
    const float step_size = a.st->lr / (float)bc1;
    const float bc2_sqrt = (float)sqrt(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float g = a.g[i] * a.gscale;
        const float p = a.p[i];
        if (a.wd != 0.f) g += a.wd * p;
        const float m = a.b1 * a.m[i] + (1.f - a.b1) * g;
        const float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
        a.m[i] = m; a.v[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        a.p[i] = p - step_size * (m / denom);
    }
}

struct AdamDevOp : Op {
    AdamDevArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, a.st);
        long long b = ceil_div(a.n, 256 * 4); if (b > 8192) b = 8192; if (b < 1) b = 1;
        hipLaunchKernelGGL(adam_step_dev_kernel, dim3((unsigned)b), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "adam_step_dev"; }
};

struct AdamOp : Op {
    AdamArgs a;
    int launch(hipStream_t s) override {
        long long b = ceil_div(a.n, 256 * 4); if (b > 8192) b = 8192; if (b < 1) b = 1;
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)b), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "adam_step"; }
};

struct MemsetOp : Op {
    void* p; size_t bytes;
    int launch(hipStream_t s) override {
        hipError_t e = hipMemsetAsync(p, 0, bytes, s);
        if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    const char* name() const override { return "memset_zero"; }
};

}  // namespace v2v

using namespace v2v;

static int loss_common(LossArgs& a, int kind, const void* x, const void* b, const float* mask, float target,
                       int64_t P, int32_t C, int32_t c_stride, int64_t N, int64_t CHW, int64_t HW, int planar) {
    memset(&a, 0, sizeof(a));
    if (!x || (kind == LOSS_L1 && !b) || (kind != LOSS_L1 && kind != LOSS_MSE_CONST)) { set_error("loss: bad argument"); return V2V_EINVAL; }
    a.a = x; a.b = b; a.mask = mask; a.target = target; a.kind = kind; a.planar = planar;
    if (planar) {
        if (N <= 0 || CHW <= 0 || HW <= 0 || CHW % HW != 0) { set_error("loss: planar shape"); return V2V_EINVAL; }
        a.n = N * CHW; a.chw = CHW; a.hw = HW;
        a.scale = 1.0 / (double)a.n;
    } else {
        if (P <= 0 || C <= 0 || C > c_stride || mask) { set_error("loss: nhwc shape"); return V2V_EINVAL; }
        a.P = P; a.C = C; a.c_stride = c_stride;
        a.scale = 1.0 / ((double)P * (double)C);
    }
    return 0;
}

extern "C" int v2v_loss_workspace_floats(void) { return 1024; }

// mean loss over the real elements, times `weight`.  NHWC mode (planar = 0): a, b activations
// [P][c_stride] of `dtype`; planar mode: fp32 [N][C][HW] with optional mask [N][1][HW].
extern "C" int v2v_loss_forward(int32_t kind, const void* a, const void* b, const float* mask, float target, float weight,
                                int64_t P, int32_t C, int32_t c_stride, int64_t N, int64_t CHW, int64_t HW, int32_t planar,
                                float* workspace, float* out, int32_t dtype, void* stream) {
    auto op = std::make_unique<LossOp>();
    int rc = loss_common(op->a, kind, a, b, mask, target, P, C, c_stride, N, CHW, HW, planar);
    if (rc) return rc;
    if (!workspace || !out) { set_error("loss: null output"); return V2V_EINVAL; }
    op->a.scale *= (double)weight;
    op->a.partials = workspace; op->a.out = out;
    op->a.nblk = (int)loss_grid(planar ? op->a.n : P * c_stride);
    op->dtype = dtype; op->backward = false;
    return submit(std::move(op), stream);
}

extern "C" int v2v_loss_backward(int32_t kind, const void* a, const void* b, const float* mask, float target, float weight,
                                 int64_t P, int32_t C, int32_t c_stride, int64_t N, int64_t CHW, int64_t HW, int32_t planar,
                                 const float* grad_out, void* da, int32_t dtype, void* stream) {
    auto op = std::make_unique<LossOp>();
    int rc = loss_common(op->a, kind, a, b, mask, target, P, C, c_stride, N, CHW, HW, planar);
    if (rc) return rc;
    if (!grad_out || !da) { set_error("loss: null gradient"); return V2V_EINVAL; }
    op->a.scale *= (double)weight;
    op->a.gout = grad_out; op->a.da = da;
    op->dtype = dtype; op->backward = true;
    return submit(std::move(op), stream);
}

// torch.optim.Adam semantics (no amsgrad): step = 1-based step count after this update
extern "C" int v2v_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                             int32_t step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) { set_error("adam: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<AdamOp>();
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    op->a = AdamArgs{param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), weight_decay, grad_scale};
    return submit(std::move(op), stream);
}

extern "C" int v2v_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                 void* state, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || !state) { set_error("adam (device state): bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<AdamDevOp>();
    op->a = AdamDevArgs{param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, grad_scale, reinterpret_cast<AdamDevState*>(state)};
    return submit(std::move(op), stream);
}

extern "C" int v2v_memset_zero(void* p, int64_t bytes, void* stream) {
    if (!p || bytes < 0) { set_error("memset: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<MemsetOp>();
    op->p = p; op->bytes = (size_t)bytes;
    return submit(std::move(op), stream);
}
