// Training-mode BatchNorm2d / InstanceNorm2d for the vid2vid generators and discriminators
// (reference: get_norm_layer, models/networks.py:23-30; .eval() is never called, so batch
// statistics are used even at inference -- SURVEY.md section 0).
//
// The conv epilogue leaves one row of per-channel (sum, sum^2) partials per M tile;
// bn_finalize reduces the rows in a fixed order (deterministic, fp64 combine) into
// scale/shift, and bn_apply streams the fp32 conv output once:
//     y = act(raw*scale + shift) [+ add0] [+ add1]        (HBM-bound, 16-byte vectors)
// which also performs the ResnetBlock residual add (models/networks.py:591-593) and the
// tower sums (models/networks.py:204, :298-299) without extra passes.
#include "v2v_internal.h"
#include <cstring>

namespace v2v {

struct BnFinArgs {
    const float* partials; int rows; int C; double inv_count; double unbias;
    const float* gamma; const float* beta; float eps;
    float* scale_shift; float* running_mean; float* running_var; float momentum;
};

// Stage 1 for layers with many statistics rows (large M: the fine scales of the 2048x1024 configs leave 4096-16384
// rows; one workgroup reducing them serially, whether the conv kernel's last arriver or a 1-block finalize, measured
// 0.6-1.0 ms, profiles/r01_v15_finalize_tail.txt): grid (C/64, G) blocks each reduce rows [g*rows/G, (g+1)*rows/G)
// of 64 channels in a fixed order into fp64 (sum, sum^2) -> ws[g][C][2].  Deterministic (fixed tree).
struct BnPartArgs {
    const float* partials; int rows; int C; int groups; double* ws;
    // stage 2 inside the same launch (round 4): the LAST group to finish a 64-channel slab (ticket word per slab, self re-arming)
    // runs bn_finalize_kernel<double>'s arithmetic on the group rows -- bit for bit the two-launch result, one launch fewer per
    // large layer (the 2048x1024 frame has ~50 of them, each a dependent ~6 us launch on the frame's critical path)
    int* ticket; BnFinArgs fin;
};

__global__ __launch_bounds__(256) void bn_partial_reduce_kernel(const BnPartArgs a) {
    __shared__ double sh[4][64][2];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int g = blockIdx.y;
    const int r0 = (int)(((long long)a.rows * g) / a.groups), r1 = (int)(((long long)a.rows * (g + 1)) / a.groups);
    double s1 = 0.0, s2 = 0.0;
    if (c < a.C) {
#pragma unroll 8                                         // 8 row loads in flight, added in the same order (one per trip measured 13 us per MB)
        for (int r = r0 + ph; r < r1; r += 4) {
            const float2 v = *reinterpret_cast<const float2*>(a.partials + ((long long)r * a.C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    }
    sh[ph][cx][0] = s1;
    sh[ph][cx][1] = s2;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        const double t1 = ((sh[0][cx][0] + sh[1][cx][0]) + sh[2][cx][0]) + sh[3][cx][0];
        const double t2 = ((sh[0][cx][1] + sh[1][cx][1]) + sh[2][cx][1]) + sh[3][cx][1];
        double* dst = a.ws + ((long long)g * a.C + c) * 2;
        if (a.ticket != nullptr) {       // write-through agent-scope stores: the finalizing workgroup may sit on another XCD (L2s are not coherent)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, (unsigned long long)__double_as_longlong(t2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { dst[0] = t1; dst[1] = t2; }
    }
    if (a.ticket == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last_flag;
    if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(a.ticket + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == a.groups - 1 ? 1 : 0;
        if (last) __hip_atomic_store(a.ticket + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the group rows were written through by their producers
    // ---- bn_finalize_kernel<double> on the group rows, same phases, same order ----
    const BnFinArgs& f = a.fin;
    s1 = 0.0; s2 = 0.0;
    if (c < a.C) {
#pragma unroll 8
        for (int r = ph; r < a.groups; r += 4) {
            s1 += a.ws[((long long)r * a.C + c) * 2 + 0];
            s2 += a.ws[((long long)r * a.C + c) * 2 + 1];
        }
    }
    sh[ph][cx][0] = s1;
    sh[ph][cx][1] = s2;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        s1 = ((sh[0][cx][0] + sh[1][cx][0]) + sh[2][cx][0]) + sh[3][cx][0];
        s2 = ((sh[0][cx][1] + sh[1][cx][1]) + sh[2][cx][1]) + sh[3][cx][1];
        const double mean = s1 * f.inv_count;
        double var = s2 * f.inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)f.eps);
        const double gm = f.gamma ? (double)f.gamma[c] : 1.0;
        const double bt = f.beta ? (double)f.beta[c] : 0.0;
        const double sc = gm * invstd;
        f.scale_shift[c] = (float)sc;
        f.scale_shift[a.C + c] = (float)(bt - mean * sc);
        f.scale_shift[2 * a.C + c] = (float)mean;
        f.scale_shift[3 * a.C + c] = (float)invstd;
        if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
        if (f.running_var)  f.running_var[c]  = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)(var * f.unbias);
    }
}

// block = 256 threads = 64 channels x 4 row-phases; rows are fp32 pairs (conv epilogue) or fp64 pairs (stage 1)
template <typename R>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnFinArgs a) {
    __shared__ double sh[4][64][2];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const R* rows_ = reinterpret_cast<const R*>(a.partials);
    double s1 = 0.0, s2 = 0.0;
    if (c < a.C) {
#pragma unroll 8
        for (int r = ph; r < a.rows; r += 4) {
            s1 += (double)rows_[((long long)r * a.C + c) * 2 + 0];
            s2 += (double)rows_[((long long)r * a.C + c) * 2 + 1];
        }
    }
    sh[ph][cx][0] = s1;
    sh[ph][cx][1] = s2;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        s1 = ((sh[0][cx][0] + sh[1][cx][0]) + sh[2][cx][0]) + sh[3][cx][0];
        s2 = ((sh[0][cx][1] + sh[1][cx][1]) + sh[2][cx][1]) + sh[3][cx][1];
        const double mean = s1 * a.inv_count;
        double var = s2 * a.inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)a.eps);
        const double g = a.gamma ? (double)a.gamma[c] : 1.0;
        const double b = a.beta ? (double)a.beta[c] : 0.0;
        const double sc = g * invstd;
        a.scale_shift[c] = (float)sc;
        a.scale_shift[a.C + c] = (float)(b - mean * sc);
        a.scale_shift[2 * a.C + c] = (float)mean;       // rows 2,3: saved for the backward pass
        a.scale_shift[3 * a.C + c] = (float)invstd;
        if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        if (a.running_var)  a.running_var[c]  = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * a.unbias);
    }
}

struct BnFinOp : Op {
    BnFinArgs a;
    bool rows_f64 = false;
    int launch(hipStream_t s) override {
        if (rows_f64) hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((unsigned)ceil_div(a.C, 64)), dim3(256), 0, s, a);
        else          hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((unsigned)ceil_div(a.C, 64)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "bn_finalize"; }
};

struct BnPartOp : Op {
    BnPartArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((unsigned)ceil_div(a.C, 64), (unsigned)a.groups), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "bn_partial_reduce"; }
};

struct BnApplyArgs {
    const float* raw; int c_stride_raw; const float* scale_shift;
    const void* add0; const void* add1; void* y;
    long long P; int C; int c_stride; int act; float act_param;
    // second member of a paired launch (v2v_bn_apply_pair, gridDim.y == 2): same geometry, its own tensors
    const float* raw1; const float* scale_shift1; const void* add0_1; const void* add1_1; void* y1;
    // fp32 only (v2v_bn_apply_x3): the result also as the bf16x3 operand [hi | lo | hi] of the consumer convolution, channel stride 3 C
    unsigned short* x3 = nullptr; unsigned short* x3_1 = nullptr;
    int raw_bf16 = 0;       // v2v_bn_apply_raw: `raw` holds bf16 (V2V_OUT_RAW_ACT_NHWC of a bf16 convolution), stride in bf16 elements
};

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a_in) {
    BnApplyArgs a = a_in;
    if (blockIdx.y != 0) { a.raw = a_in.raw1; a.scale_shift = a_in.scale_shift1; a.add0 = a_in.add0_1; a.add1 = a_in.add1_1; a.y = a_in.y1; a.x3 = a_in.x3_1; }
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = a.c_stride / VEC;                       // vectors per pixel row
    const long long nvec = a.P * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const T* add0 = reinterpret_cast<const T*>(a.add0);
    const T* add1 = reinterpret_cast<const T*>(a.add1);
    T* y = reinterpret_cast<T*>(a.y);
    if constexpr (VEC == 8) {
        // DENSE FAST PATH (round 5): bf16 activations, fp32 raw, C % 8 == 0 and no channel padding anywhere (c_stride == c_stride_raw
        // == C), one of NONE / RELU / LEAKY.  Then vector v IS elements [8v, 8v + 8) of every operand -- no 64-bit division per
        // vector (the general loop's `v / vpr`), the thread's channel group is fixed when the grid stride is a multiple of the vectors
        // per pixel, so scale / shift live in registers, the residuals arrive as one 16-byte load each instead of eight 2-byte loads,
        // and the activation dispatch is hoisted.  bn_apply was 2.2 ms of the 10.5 ms 2048x1024 frame at 48 % of the measured copy
        // rate (profiles/r04_f3_per_layer_roofline_hires.txt) -- an issue-bound streaming kernel.  Same fp32 expression per element
        // (fma(raw, scale, shift) -> activation -> + add0 -> + add1 -> RNE to bf16): bit-identical results.
        const bool simple = a.act == V2V_ACT_NONE || a.act == V2V_ACT_RELU || a.act == V2V_ACT_LEAKY;
        if (a.x3 == nullptr && (a.C & 7) == 0 && a.c_stride == a.C && a.c_stride_raw == a.C && simple &&
            nvec < (1ll << 28) && (stride % vpr) == 0) {
            const unsigned nv = (unsigned)nvec, st = (unsigned)stride;
            unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
            const int c0 = (int)(v % (unsigned)vpr) * 8;
            float sc[8], sh[8];
            {
                const float4 s0 = *reinterpret_cast<const float4*>(a.scale_shift + c0), s1 = *reinterpret_cast<const float4*>(a.scale_shift + c0 + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(a.scale_shift + a.C + c0), h1 = *reinterpret_cast<const float4*>(a.scale_shift + a.C + c0 + 4);
                sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
                sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
            }
            const bool is_none = a.act == V2V_ACT_NONE, is_relu = a.act == V2V_ACT_RELU;
            const float slope = a.act_param;
            const uint4* const r0 = reinterpret_cast<const uint4*>(add0);
            const uint4* const r1 = reinterpret_cast<const uint4*>(add1);
            const bool rawb = a.raw_bf16 != 0;               // round 6: raw stored as bf16 (persistent single-chunk tiles): one 16-byte load
            for (; v < nv; v += st) {
                float r[8];
                if (rawb) {
                    const uint4 rq = reinterpret_cast<const uint4*>(a.raw)[v];
                    r[0] = __uint_as_float(rq.x << 16); r[1] = __uint_as_float(rq.x & 0xffff0000u); r[2] = __uint_as_float(rq.y << 16); r[3] = __uint_as_float(rq.y & 0xffff0000u);
                    r[4] = __uint_as_float(rq.z << 16); r[5] = __uint_as_float(rq.z & 0xffff0000u); r[6] = __uint_as_float(rq.w << 16); r[7] = __uint_as_float(rq.w & 0xffff0000u);
                } else {
                    const float4 ra = *reinterpret_cast<const float4*>(a.raw + (size_t)v * 8), rb = *reinterpret_cast<const float4*>(a.raw + (size_t)v * 8 + 4);
                    r[0] = ra.x; r[1] = ra.y; r[2] = ra.z; r[3] = ra.w; r[4] = rb.x; r[5] = rb.y; r[6] = rb.z; r[7] = rb.w;
                }
                uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0;
                if (r0) q0 = r0[v];
                if (r1) q1 = r1[v];
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float t = r[q] * sc[q] + sh[q];
                    const float neg = is_relu ? 0.f : t * slope;
                    o[q] = (is_none || t > 0.f) ? t : neg;
                }
                if (r0) {
                    o[0] += __uint_as_float(q0.x << 16); o[1] += __uint_as_float(q0.x & 0xffff0000u); o[2] += __uint_as_float(q0.y << 16); o[3] += __uint_as_float(q0.y & 0xffff0000u);
                    o[4] += __uint_as_float(q0.z << 16); o[5] += __uint_as_float(q0.z & 0xffff0000u); o[6] += __uint_as_float(q0.w << 16); o[7] += __uint_as_float(q0.w & 0xffff0000u);
                }
                if (r1) {
                    o[0] += __uint_as_float(q1.x << 16); o[1] += __uint_as_float(q1.x & 0xffff0000u); o[2] += __uint_as_float(q1.y << 16); o[3] += __uint_as_float(q1.y & 0xffff0000u);
                    o[4] += __uint_as_float(q1.z << 16); o[5] += __uint_as_float(q1.z & 0xffff0000u); o[6] += __uint_as_float(q1.w << 16); o[7] += __uint_as_float(q1.w & 0xffff0000u);
                }
                uint4 pk;
                pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); pk.z = pack_bf16x2(o[4], o[5]); pk.w = pack_bf16x2(o[6], o[7]);
                reinterpret_cast<uint4*>(y)[v] = pk;
            }
            return;
        }
    }
    const bool small = nvec < (1ll << 31);                   // 32-bit division (a 64-bit one is ~100 instructions per vector)
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long pix = small ? (long long)((unsigned)v / (unsigned)vpr) : v / vpr;
        const int c0 = (int)(v - pix * vpr) * VEC;
        float o[VEC];
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const int c = c0 + q * 4;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f), sc = r, sh = r;
            if (c < a.C) {   // c_stride_raw is a multiple of 4 and >= C: the 4-wide raw load stays inside the pixel's row
                if (a.raw_bf16) {
                    const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.raw) + pix * a.c_stride_raw + c);
                    r = make_float4(__uint_as_float(rb.x << 16), __uint_as_float(rb.x & 0xffff0000u),
                                    __uint_as_float(rb.y << 16), __uint_as_float(rb.y & 0xffff0000u));
                } else
                r = *reinterpret_cast<const float4*>(a.raw + pix * a.c_stride_raw + c);
                if ((a.C & 3) == 0) {
                    sc = *reinterpret_cast<const float4*>(a.scale_shift + c);
                    sh = *reinterpret_cast<const float4*>(a.scale_shift + a.C + c);
                } else {     // C % 4 != 0 (2-channel test towers, 1027-channel --label_feat trunks): [2][C] rows are unaligned
                    sc.x = a.scale_shift[c]; sh.x = a.scale_shift[a.C + c];
                    if (c + 1 < a.C) { sc.y = a.scale_shift[c + 1]; sh.y = a.scale_shift[a.C + c + 1]; }
                    if (c + 2 < a.C) { sc.z = a.scale_shift[c + 2]; sh.z = a.scale_shift[a.C + c + 2]; }
                    if (c + 3 < a.C) { sc.w = a.scale_shift[c + 3]; sh.w = a.scale_shift[a.C + c + 3]; }
                }
            }
            o[q * 4 + 0] = apply_act(r.x * sc.x + sh.x, a.act, a.act_param);
            o[q * 4 + 1] = apply_act(r.y * sc.y + sh.y, a.act, a.act_param);
            o[q * 4 + 2] = apply_act(r.z * sc.z + sh.z, a.act, a.act_param);
            o[q * 4 + 3] = apply_act(r.w * sc.w + sh.w, a.act, a.act_param);
        }
        const long long e = pix * a.c_stride + c0;
        if (add0) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) o[q] += load_act(add0, e + q);
        }
        if (add1) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) o[q] += load_act(add1, e + q);
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            if (c0 + q >= a.C) o[q] = 0.f;
        }
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
            if (a.x3) {                                      // the arithmetic of split_x3_kernel (csrc/pointwise.hip), no second pass
                unsigned short hi[4], lo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hi[q] = f32_to_bf16_bits(o[q]);
                    lo[q] = f32_to_bf16_bits(o[q] - bf16_bits_to_f32(hi[q]));
                }
                const uint2 vh = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
                const uint2 vl = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                unsigned short* o3 = a.x3 + pix * 3ll * a.C + c0;
                *reinterpret_cast<uint2*>(o3) = vh;
                *reinterpret_cast<uint2*>(o3 + a.C) = vl;
                *reinterpret_cast<uint2*>(o3 + 2 * a.C) = vh;
            }
        } else {
            uint4 pk;
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
            pk.z = pack_bf16x2(o[4], o[5]);
            pk.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4*>(y + e) = pk;
        }
    }
}

struct BnApplyOp : Op {
    BnApplyArgs a; int dtype; int members = 1;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        long long nvec = a.P * (a.c_stride / vec);
        long long blocks = ceil_div(nvec, 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        const dim3 grid((unsigned)blocks, (unsigned)members);
        if (dtype == V2V_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "bn_apply"; }
};


// ---------------------------------------------------------------------------------------
// backward of  y = act(norm(raw)) (+ residuals)   (autograd of nn.BatchNorm2d / InstanceNorm2d
// in training mode + ReLU / LeakyReLU in the reference)
//   g      = dY * act'(raw*scale + shift)
//   xhat   = (raw - mean) * invstd
//   dbeta  = sum g,   dgamma = sum g*xhat
//   dRaw   = scale * (g - dbeta/M - xhat * dgamma/M)
// Three launches: per-block partial sums (deterministic rows), a tiny finalize, one streaming pass.
// The same reduce/finalize pair with mode 1 gives plain per-channel sums (bias gradients).
// ---------------------------------------------------------------------------------------
struct BnBwdRedArgs {
    const void* dy; const float* raw; const float* stats;   // stats: [4][C] scale, shift, mean, invstd
    float* partials;                                        // [nblk][C][2]
    long long P; int C, c_stride, c_stride_raw, act; float act_param; int mode;   // mode 1: sum of dy only
    long long ppb;                                          // pixels per block
    int vec;                                                // strides / base pointers allow the 4-channel vector loads
    // finalize by the last-arriving workgroup of a 64-channel slab (replaces the bn_bwd_finalize launch: 434 launches of
    // ~10 us per training step, profiles/r02_a14_train_kernel_stats.txt); ticket == NULL: partial rows only
    int* ticket; double inv_count; float* dgamma; float* dbeta; float* coef; int accumulate;
};

__device__ __forceinline__ float act_grad_pre(float pre, int act, float param) {
    switch (act) {
        case V2V_ACT_RELU:  return pre > 0.f ? 1.f : 0.f;
        case V2V_ACT_LEAKY: return pre > 0.f ? 1.f : param;
        default:            return 1.f;
    }
}

__device__ __forceinline__ void load4(const float* p, float v[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// grid (pixel blocks, 64-channel slabs): a 2048-pixel x 1024-channel layer is 32 x 16 = 512 workgroups instead of the
// 32 of a pixel-only split (which left 7/8 of the chip idle and cost 80 us per launch, profiles/r01_v17_train_*).
// Thread (tx, ty) of the 16 x 16 block owns 4 consecutive channels (one 8- or 16-byte load per operand) and every 16th
// pixel of the block; the 16 pixel-phases are added in a fixed order through LDS -> deterministic partial rows.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdRedArgs a) {
    __shared__ float sh[16][64][2];
    const T* dy = reinterpret_cast<const T*>(a.dy);
    const long long p0 = (long long)blockIdx.x * a.ppb;
    long long p1 = p0 + a.ppb; if (p1 > a.P) p1 = a.P;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64 + tx * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < a.C) {                       // channel strides are multiples of 4 and >= C: the 4-wide loads stay in the row
        float sc[4], sf[4], mean[4], inv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q < a.C ? c0 + q : a.C - 1;
            sc[q] = 1.f; sf[q] = 0.f; mean[q] = 0.f; inv[q] = 1.f;
            if (a.mode == 0) { sc[q] = a.stats[c]; sf[q] = a.stats[a.C + c]; mean[q] = a.stats[2 * a.C + c]; inv[q] = a.stats[3 * a.C + c]; }
        }
        for (long long p = p0 + ty; p < p1; p += 16) {
            float g[4];
            if (a.vec) load4(dy + p * a.c_stride + c0, g);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = c0 + q < a.C ? load_act(dy, p * a.c_stride + c0 + q) : 0.f;
            }
            if (a.mode == 0) {
                float r[4];
                if (a.vec) load4(a.raw + p * a.c_stride_raw + c0, r);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = c0 + q < a.C ? a.raw[p * a.c_stride_raw + c0 + q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g[q] *= act_grad_pre(r[q] * sc[q] + sf[q], a.act, a.act_param);
                    s2[q] += g[q] * ((r[q] - mean[q]) * inv[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s1[q] += g[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { sh[ty][tx * 4 + q][0] = s1[q]; sh[ty][tx * 4 + q][1] = s2[q]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.y * 64 + threadIdx.x;
        if (c < a.C) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { t1 += sh[q][threadIdx.x][0]; t2 += sh[q][threadIdx.x][1]; }
            float* dst = a.partials + ((long long)blockIdx.x * a.C + c) * 2;
            if (a.ticket != nullptr) {       // 8-byte write-through agent-scope store, read back below with agent-scope loads
                const unsigned long long bits = (unsigned long long)__float_as_uint(t1) | ((unsigned long long)__float_as_uint(t2) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { dst[0] = t1; dst[1] = t2; }
        }
    }
    if (a.ticket == nullptr) return;
    // ---- the last workgroup of this channel slab sums the rows in a fixed order (same arithmetic as bn_bwd_finalize_kernel:
    //      4 row phases in double, then ((p0 + p1) + p2) + p3) -> deterministic, independent of which workgroup is last ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last_flag;
    if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(a.ticket + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == (int)gridDim.x - 1 ? 1 : 0;
        if (last) __hip_atomic_store(a.ticket + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    double* shd = reinterpret_cast<double*>(&sh[0][0][0]);       // [4][64][2] doubles = 4 KiB of the 8 KiB
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    // the rows were written through to memory by their producers; one agent-scope acquire (invalidate) lets this workgroup
    // read them with ordinary, pipelined loads (128 serial agent-scope atomic loads per thread cost 13 us per launch)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double d1 = 0.0, d2 = 0.0;
    if (c < a.C) {
        const float2* rows2 = reinterpret_cast<const float2*>(a.partials);
#pragma unroll 8
        for (int r = ph; r < (int)gridDim.x; r += 4) {
            const float2 v = rows2[(long long)r * a.C + c];
            d1 += (double)v.x;
            d2 += (double)v.y;
        }
    }
    shd[(ph * 64 + cx) * 2] = d1; shd[(ph * 64 + cx) * 2 + 1] = d2;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        d1 = ((shd[(0 * 64 + cx) * 2] + shd[(1 * 64 + cx) * 2]) + shd[(2 * 64 + cx) * 2]) + shd[(3 * 64 + cx) * 2];
        d2 = ((shd[(0 * 64 + cx) * 2 + 1] + shd[(1 * 64 + cx) * 2 + 1]) + shd[(2 * 64 + cx) * 2 + 1]) + shd[(3 * 64 + cx) * 2 + 1];
        if (a.dbeta)  a.dbeta[c]  = (a.accumulate ? a.dbeta[c] : 0.f) + (float)d1;
        if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + (float)d2;
        if (a.coef) { a.coef[c] = (float)(d1 * a.inv_count); a.coef[a.C + c] = (float)(d2 * a.inv_count); }
    }
}

struct BnBwdFinArgs {
    const float* partials; int rows; int C; double inv_count;
    float* dgamma; float* dbeta; float* coef;   // coef: [2][C] = (sum g / M, sum g*xhat / M) or NULL
    int accumulate;
};

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BnBwdFinArgs a) {
    __shared__ double sh[4][64][2];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < a.C) {
        for (int r = ph; r < a.rows; r += 4) {
            const float2 v = *reinterpret_cast<const float2*>(a.partials + ((long long)r * a.C + c) * 2);
            s1 += (double)v.x; s2 += (double)v.y;
        }
    }
    sh[ph][cx][0] = s1; sh[ph][cx][1] = s2;
    __syncthreads();
    if (ph == 0 && c < a.C) {
        s1 = ((sh[0][cx][0] + sh[1][cx][0]) + sh[2][cx][0]) + sh[3][cx][0];
        s2 = ((sh[0][cx][1] + sh[1][cx][1]) + sh[2][cx][1]) + sh[3][cx][1];
        if (a.dbeta)  a.dbeta[c]  = (a.accumulate ? a.dbeta[c] : 0.f) + (float)s1;
        if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + (float)s2;
        if (a.coef) { a.coef[c] = (float)(s1 * a.inv_count); a.coef[a.C + c] = (float)(s2 * a.inv_count); }
    }
}

struct BnBwdApplyArgs {
    const void* dy; const float* raw; const float* stats; const float* coef; void* draw;
    long long P; int C, c_stride, c_stride_raw, c_stride_out, act; float act_param;
    int vec;
};

__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pack_bf16x2(v[0], v[1]);
    t.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}

// grid (64-pixel blocks, 64-channel slabs), thread (tx, ty) of 16 x 16: 4 fixed channels (statistics and reduction
// coefficients live in registers for the whole pixel walk), every 16th pixel; one vector load per operand and one
// vector store per pixel.  Writes dRaw in the activation dtype, pad channels zero.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApplyArgs a) {
    const T* dy = reinterpret_cast<const T*>(a.dy);
    T* out = reinterpret_cast<T*>(a.draw);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64 + tx * 4;
    if (c0 >= a.c_stride_out) return;
    float sc[4], sf[4], mean[4], inv[4], k1[4], k2[4];
    bool okc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        okc[q] = c0 + q < a.C;
        const int c = okc[q] ? c0 + q : a.C - 1;
        sc[q] = a.stats[c]; sf[q] = a.stats[a.C + c]; mean[q] = a.stats[2 * a.C + c]; inv[q] = a.stats[3 * a.C + c];
        k1[q] = a.coef[c]; k2[q] = a.coef[a.C + c];
    }
    const long long p0 = (long long)blockIdx.x * 64;
    long long p1 = p0 + 64; if (p1 > a.P) p1 = a.P;
    for (long long p = p0 + ty; p < p1; p += 16) {
        float g[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        if (c0 < a.C) {
            if (a.vec) { load4(dy + p * a.c_stride + c0, g); load4(a.raw + p * a.c_stride_raw + c0, r); }
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (okc[q]) { g[q] = load_act(dy, p * a.c_stride + c0 + q); r[q] = a.raw[p * a.c_stride_raw + c0 + q]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gg = g[q] * act_grad_pre(r[q] * sc[q] + sf[q], a.act, a.act_param);
            o[q] = okc[q] ? sc[q] * (gg - k1[q] - (r[q] - mean[q]) * inv[q] * k2[q]) : 0.f;
        }
        store4(out + p * a.c_stride_out + c0, o);
    }
}

// (Round 6 tried the whole norm backward of a small layer in ONE launch -- operands kept in registers across the in-launch
// finalize, the other workgroups of a 64-channel slab spinning on its flag: correct to a few ulp, but 27-52 us per launch for the
// 2048-pixel x 1024-channel layers against 20 + 8.5 us for the reduce / apply pair below, because the ticket -> finalize -> flag
// -> coefficient chain is five dependent agent-scope round trips and the spinning workgroups hold the CU slots the rest of the
// grid waits for while weight-gradient kernels share the chip (profiles/r06_v9_train_kernel_stats_bn_fused.txt, r06_v8_trainab.txt:
// 54.1 vs 54.3 ms per chunk).  Removed.)
struct BnBwdOp : Op {
    BnBwdRedArgs r; BnBwdFinArgs f; BnBwdApplyArgs ap; int dtype, nblk; bool do_apply;
    int launch(hipStream_t s) override {
        const dim3 rgrid((unsigned)nblk, (unsigned)ceil_div(r.C, 64));
        if (dtype == V2V_BF16) hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, s, r);
        else                   hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, rgrid, dim3(256), 0, s, r);
        int rc = check_launch(); if (rc) return rc;
        if (r.ticket == nullptr) {
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(f.C, 64)), dim3(256), 0, s, f);
            rc = check_launch(); if (rc) return rc;
        }
        if (do_apply) {
            const dim3 agrid((unsigned)ceil_div(ap.P, 64), (unsigned)ceil_div(ap.c_stride_out, 64));
            if (dtype == V2V_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, agrid, dim3(256), 0, s, ap);
            else                   hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, agrid, dim3(256), 0, s, ap);
            rc = check_launch();
        }
        return rc;
    }
    const char* name() const override { return do_apply ? "bn_backward" : "channel_sum"; }
};

// ---------------------------------------------------------------------------------------
// backward of a conv epilogue activation (no norm):  g = dY * act'(y) * out_scale, with dY / y
// either NHWC activations or planar fp32 NCHW (API-facing heads); g is NHWC, pad channels zero
// ---------------------------------------------------------------------------------------
struct ActBwdArgs {
    const void* dy; const void* y; void* g;
    long long NP; long long hw; int C, c_stride_in, c_stride_out, act, nchw; float act_param, out_scale;
};

__device__ __forceinline__ float act_grad_out(float yv, int act, float param, float out_scale) {
    switch (act) {
        case V2V_ACT_RELU:    return yv > 0.f ? out_scale : 0.f;
        case V2V_ACT_LEAKY:   return (yv > 0.f ? 1.f : param) * out_scale;
        case V2V_ACT_TANH:    { const float t = yv / out_scale; return (1.f - t * t) * out_scale; }
        case V2V_ACT_SIGMOID: { const float t = yv / out_scale; return t * (1.f - t) * out_scale; }
        default:              return out_scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const ActBwdArgs a) {
    T* g = reinterpret_cast<T*>(a.g);
    const long long total = a.NP * a.c_stride_out;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long pix = e / a.c_stride_out;       // n*hw + p
        const int c = (int)(e - pix * a.c_stride_out);
        float o = 0.f;
        if (c < a.C) {
            float d, yv;
            if (a.nchw) {
                const long long n = pix / a.hw, p = pix - n * a.hw;
                const long long idx = (n * a.C + c) * a.hw + p;
                d = reinterpret_cast<const float*>(a.dy)[idx];
                yv = a.act == V2V_ACT_NONE ? 0.f : reinterpret_cast<const float*>(a.y)[idx];
            } else {
                d = load_act(reinterpret_cast<const T*>(a.dy), pix * a.c_stride_in + c);
                yv = a.act == V2V_ACT_NONE ? 0.f : load_act(reinterpret_cast<const T*>(a.y), pix * a.c_stride_in + c);
            }
            o = d * act_grad_out(yv, a.act, a.act_param, a.out_scale);
        }
        store_act(g, e, o);
    }
}

struct ActBwdOp : Op {
    ActBwdArgs a; int dtype;
    int launch(hipStream_t s) override {
        long long blocks = ceil_div(a.NP * a.c_stride_out, 256);
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        if (dtype == V2V_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(act_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "act_backward"; }
};

// Self-re-arming ticket words for the fused finalize: one device buffer for the process, handed out round-robin (an op
// keeps its 64 words for life; two ops can only share words when 1024 ops apart, far beyond anything in flight together).
static int* take_tickets(int n) {
    constexpr int POOL = 1 << 16;
    static int* pool = nullptr;
    static bool failed = false;
    static unsigned next = 0;
    if (failed || n > 64) return nullptr;
    if (pool == nullptr) {
        if (hipMalloc(reinterpret_cast<void**>(&pool), POOL * sizeof(int)) != hipSuccess || hipMemset(pool, 0, POOL * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError(); pool = nullptr; failed = true; return nullptr;      // no device (dry run): separate finalize launch
        }
    }
    const unsigned at = __atomic_fetch_add(&next, 64u, __ATOMIC_RELAXED) % POOL;
    return pool + at;
}

static void fuse_finalize(BnBwdOp* op) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("V2V_BN_BWD_FUSED"); on = (e && e[0] == '0') ? 0 : 1; }
    op->r.ticket = nullptr;
    if (!on || v2v_get_dry_run()) return;
    int* t = take_tickets((int)ceil_div(op->f.C, 64));
    if (t == nullptr) return;
    op->r.ticket = t; op->r.inv_count = op->f.inv_count; op->r.dgamma = op->f.dgamma; op->r.dbeta = op->f.dbeta;
    op->r.coef = op->f.coef; op->r.accumulate = op->f.accumulate;
}

static int bwd_blocks(long long P, long long* ppb) {
    long long nblk = ceil_div(P, 64);
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    *ppb = ceil_div(P, nblk);
    return (int)ceil_div(P, *ppb);
}

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_bn_finalize_groups(int32_t rows) {
    return rows > 512 ? (rows >= 64 * 128 ? 64 : (rows + 127) / 128) : 0;
}

extern "C" int v2v_bn_finalize(const float* partials, int32_t rows, int32_t C, int64_t count,
                               const float* gamma, const float* beta, float eps,
                               float* scale_shift, float* running_mean, float* running_var, float momentum,
                               double* workspace, void* stream) {
    if (!partials || !scale_shift || rows <= 0 || C <= 0 || count <= 0) { set_error("bn_finalize: bad argument"); return V2V_EINVAL; }
    const int groups = workspace ? v2v_bn_finalize_groups(rows) : 0;
    auto op = std::make_unique<BnFinOp>();
    BnFinArgs& a = op->a;
    a.partials = groups > 0 ? reinterpret_cast<const float*>(workspace) : partials;
    a.rows = groups > 0 ? groups : rows; a.C = C;
    op->rows_f64 = groups > 0;
    a.inv_count = 1.0 / (double)count;
    a.unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.scale_shift = scale_shift; a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum;
    if (groups > 0) {
        auto p1 = std::make_unique<BnPartOp>();
        p1->a.partials = partials; p1->a.rows = rows; p1->a.C = C; p1->a.groups = groups; p1->a.ws = workspace;
        p1->a.ticket = nullptr; p1->a.fin = a;
        // stage 2 by the last group of every channel slab, inside the stage-1 launch (V2V_BN_FIN_FUSED=0: the two launches)
        static const int fused = [] { const char* e = getenv("V2V_BN_FIN_FUSED"); return (e && e[0] == '0') ? 0 : 1; }();
        if (fused && !v2v_get_dry_run()) p1->a.ticket = take_tickets((int)ceil_div(C, 64));
        const bool done = p1->a.ticket != nullptr;
        int rc = submit(std::move(p1), stream);
        if (rc != 0 || done) return rc;
    }
    return submit(std::move(op), stream);
}

extern "C" int v2v_bn_apply(const float* raw, int32_t c_stride_raw, const float* scale_shift,
                            const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                            int32_t act, float act_param, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!raw || !scale_shift || !y || P <= 0) { set_error("bn_apply: bad argument"); return V2V_EINVAL; }
    if (c_stride % vec != 0 || c_stride_raw % 4 != 0 || C < 1 || C > c_stride || C > c_stride_raw) {
        set_error("bn_apply: channel strides must be multiples of the vector width (C=%d stride=%d raw=%d)", C, c_stride, c_stride_raw);
        return V2V_EINVAL;
    }
    auto op = std::make_unique<BnApplyOp>();
    BnApplyArgs& a = op->a;
    a.raw = raw; a.c_stride_raw = c_stride_raw; a.scale_shift = scale_shift;
    a.add0 = add0; a.add1 = add1; a.y = y; a.P = P; a.C = C; a.c_stride = c_stride;
    a.act = act; a.act_param = act_param;
    a.raw1 = nullptr; a.scale_shift1 = nullptr; a.add0_1 = nullptr; a.add1_1 = nullptr; a.y1 = nullptr;
    op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_bn_apply_raw(const void* raw, int32_t raw_dtype, int32_t c_stride_raw, const float* scale_shift,
                                const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                                int32_t act, float act_param, int32_t dtype, void* stream) {
    if (raw_dtype == V2V_F32)
        return v2v_bn_apply(reinterpret_cast<const float*>(raw), c_stride_raw, scale_shift, add0, add1, y, P, C, c_stride, act, act_param, dtype, stream);
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (raw_dtype != V2V_BF16 || !raw || !scale_shift || !y || P <= 0) { set_error("bn_apply_raw: bad argument"); return V2V_EINVAL; }
    if (c_stride % vec != 0 || c_stride_raw % 8 != 0 || C < 1 || C > c_stride || C > c_stride_raw || (((uintptr_t)raw) & 15)) {
        set_error("bn_apply_raw: bf16 raw needs a 16-byte aligned tensor and channel strides that are multiples of 8 (C=%d stride=%d raw=%d)", C, c_stride, c_stride_raw);
        return V2V_EINVAL;
    }
    auto op = std::make_unique<BnApplyOp>();
    BnApplyArgs& a = op->a;
    a.raw = reinterpret_cast<const float*>(raw); a.c_stride_raw = c_stride_raw; a.scale_shift = scale_shift;
    a.add0 = add0; a.add1 = add1; a.y = y; a.P = P; a.C = C; a.c_stride = c_stride;
    a.act = act; a.act_param = act_param;
    a.raw1 = nullptr; a.scale_shift1 = nullptr; a.add0_1 = nullptr; a.add1_1 = nullptr; a.y1 = nullptr;
    a.raw_bf16 = 1;
    op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_bn_apply_pair(const float* raw_a, const float* scale_shift_a, const void* add0_a, const void* add1_a, void* y_a,
                                 const float* raw_b, const float* scale_shift_b, const void* add0_b, const void* add1_b, void* y_b,
                                 int32_t c_stride_raw, int64_t P, int32_t C, int32_t c_stride,
                                 int32_t act, float act_param, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!raw_a || !scale_shift_a || !y_a || !raw_b || !scale_shift_b || !y_b || P <= 0 || y_a == y_b) { set_error("bn_apply_pair: bad argument"); return V2V_EINVAL; }
    if ((add0_a != nullptr) != (add0_b != nullptr) || (add1_a != nullptr) != (add1_b != nullptr)) {
        set_error("bn_apply_pair: both members need the same set of residual operands"); return V2V_EINVAL;
    }
    if (c_stride % vec != 0 || c_stride_raw % 4 != 0 || C < 1 || C > c_stride || C > c_stride_raw) {
        set_error("bn_apply_pair: channel strides must be multiples of the vector width (C=%d stride=%d raw=%d)", C, c_stride, c_stride_raw);
        return V2V_EINVAL;
    }
    auto op = std::make_unique<BnApplyOp>();
    BnApplyArgs& a = op->a;
    a.raw = raw_a; a.c_stride_raw = c_stride_raw; a.scale_shift = scale_shift_a;
    a.add0 = add0_a; a.add1 = add1_a; a.y = y_a; a.P = P; a.C = C; a.c_stride = c_stride;
    a.act = act; a.act_param = act_param;
    a.raw1 = raw_b; a.scale_shift1 = scale_shift_b; a.add0_1 = add0_b; a.add1_1 = add1_b; a.y1 = y_b;
    op->dtype = dtype; op->members = 2;
    return submit(std::move(op), stream);
}

// fp32 bn_apply (one or two members) that ALSO writes the bf16x3 operand of the result (include/v2v_hip.h, v2v_split_x3): the consumer
// convolution of the fp32 engine's x3 mode reads it directly, the separate split pass disappears.
extern "C" int v2v_bn_apply_x3(const float* raw_a, const float* scale_shift_a, const void* add0_a, const void* add1_a, void* y_a, void* x3_a,
                               const float* raw_b, const float* scale_shift_b, const void* add0_b, const void* add1_b, void* y_b, void* x3_b,
                               int32_t c_stride_raw, int64_t P, int32_t C, int32_t act, float act_param, void* stream) {
    const bool two = raw_b != nullptr;
    if (!raw_a || !scale_shift_a || !y_a || !x3_a || P <= 0 || C < 4 || C % 4 != 0 || c_stride_raw % 4 != 0 || C > c_stride_raw ||
        (two && (!scale_shift_b || !y_b || !x3_b || y_a == y_b || x3_a == x3_b || (add0_a != nullptr) != (add0_b != nullptr) ||
                 (add1_a != nullptr) != (add1_b != nullptr)))) {
        set_error("bn_apply_x3: bad argument (fp32, dense channel stride C %% 4 == 0)"); return V2V_EINVAL;
    }
    auto op = std::make_unique<BnApplyOp>();
    BnApplyArgs& a = op->a;
    a.raw = raw_a; a.c_stride_raw = c_stride_raw; a.scale_shift = scale_shift_a;
    a.add0 = add0_a; a.add1 = add1_a; a.y = y_a; a.P = P; a.C = C; a.c_stride = C;
    a.act = act; a.act_param = act_param; a.x3 = reinterpret_cast<unsigned short*>(x3_a);
    a.raw1 = raw_b; a.scale_shift1 = scale_shift_b; a.add0_1 = add0_b; a.add1_1 = add1_b; a.y1 = y_b;
    a.x3_1 = reinterpret_cast<unsigned short*>(x3_b);
    op->dtype = V2V_F32; op->members = two ? 2 : 1;
    return submit(std::move(op), stream);
}

extern "C" int v2v_bn_backward_rows(int64_t P) {
    long long ppb; return bwd_blocks(P, &ppb);
}

extern "C" int v2v_bn_backward(const void* dy, const float* raw, int32_t c_stride_raw, const float* stats,
                               void* draw, int32_t c_stride_out, float* dgamma, float* dbeta, int32_t accumulate,
                               float* workspace, int64_t P, int32_t C, int32_t c_stride,
                               int32_t act, float act_param, int32_t dtype, void* stream) {
    if (!dy || !raw || !stats || !draw || !workspace || P <= 0 || C <= 0) { set_error("bn_backward: bad argument"); return V2V_EINVAL; }
    if (c_stride_out % 4 != 0 || C > c_stride || C > c_stride_raw || C > c_stride_out) { set_error("bn_backward: strides"); return V2V_EINVAL; }
    if ((uintptr_t)draw & 15) { set_error("bn_backward: draw must be 16-byte aligned"); return V2V_EINVAL; }
    if (act != V2V_ACT_NONE && act != V2V_ACT_RELU && act != V2V_ACT_LEAKY) { set_error("bn_backward: activation"); return V2V_EINVAL; }
    auto op = std::make_unique<BnBwdOp>();
    long long ppb; int nblk = bwd_blocks(P, &ppb);     // (the workspace bound the caller sized)
    const int slabs = (int)ceil_div(C, 64);
    if ((long long)nblk * slabs > 2048 && nblk > 64) {  // enough workgroups already: fewer, longer pixel blocks = fewer partial rows
        long long want = ceil_div(2048, slabs); if (want < 64) want = 64;
        if (want < nblk) { ppb = ceil_div(P, want); nblk = (int)ceil_div(P, ppb); }
    }
    op->dtype = dtype; op->nblk = nblk; op->do_apply = true;
    float* partials = workspace;                      // [nblk][C][2]
    float* coef = workspace + (long long)nblk * C * 2; // [2][C]
    const int vec = (c_stride % 4 == 0 && c_stride_raw % 4 == 0 && (((uintptr_t)dy | (uintptr_t)raw) & 15) == 0) ? 1 : 0;
    op->r = BnBwdRedArgs{dy, raw, stats, partials, P, C, c_stride, c_stride_raw, act, act_param, 0, ppb, vec};
    op->f = BnBwdFinArgs{partials, nblk, C, 1.0 / (double)P, dgamma, dbeta, coef, accumulate};
    op->ap = BnBwdApplyArgs{dy, raw, stats, coef, draw, P, C, c_stride, c_stride_raw, c_stride_out, act, act_param, vec};
    fuse_finalize(op.get());
    return submit(std::move(op), stream);
}

extern "C" int v2v_channel_sum(const void* x, float* out, int32_t accumulate, float* workspace,
                               int64_t P, int32_t C, int32_t c_stride, int32_t dtype, void* stream) {
    if (!x || !out || !workspace || P <= 0 || C <= 0 || C > c_stride) { set_error("channel_sum: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<BnBwdOp>();
    long long ppb; const int nblk = bwd_blocks(P, &ppb);
    op->dtype = dtype; op->nblk = nblk; op->do_apply = false;
    const int vec = (c_stride % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    op->r = BnBwdRedArgs{x, nullptr, nullptr, workspace, P, C, c_stride, 0, V2V_ACT_NONE, 0.f, 1, ppb, vec};
    op->f = BnBwdFinArgs{workspace, nblk, C, 1.0, nullptr, out, nullptr, accumulate};
    memset(&op->ap, 0, sizeof(op->ap));
    fuse_finalize(op.get());
    return submit(std::move(op), stream);
}

extern "C" int v2v_act_backward(const void* dy, const void* y, void* g, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t c_stride_in, int32_t c_stride_out, int32_t nchw, int32_t act, float act_param,
                                float out_scale, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!dy || !g || (act != V2V_ACT_NONE && !y) || c_stride_out % vec != 0 || C > c_stride_out || (!nchw && C > c_stride_in)) {
        set_error("act_backward: bad argument"); return V2V_EINVAL;
    }
    if ((act == V2V_ACT_TANH || act == V2V_ACT_SIGMOID) && out_scale == 0.f) { set_error("act_backward: out_scale 0"); return V2V_EINVAL; }
    auto op = std::make_unique<ActBwdOp>();
    op->a = ActBwdArgs{dy, y, g, (long long)N * H * W, (long long)H * W, C, c_stride_in, c_stride_out, act, nchw, act_param, out_scale};
    op->dtype = dtype;
    return submit(std::move(op), stream);
}
