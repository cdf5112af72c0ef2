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

namespace v2v {

struct BnFinArgs {
    const float* partials; int rows; int C; double inv_count; double unbias;
    const float* gamma; const float* beta; float eps;
    float* scale_shift; float* running_mean; float* running_var; float momentum;
};

// block = 256 threads = 64 channels x 4 row-phases
__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnFinArgs a) {
    __shared__ double sh[4][64][2];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < a.C) {
        for (int r = ph; r < a.rows; r += 4) {
            const float2 v = *reinterpret_cast<const float2*>(a.partials + ((long long)r * a.C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
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
        if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        if (a.running_var)  a.running_var[c]  = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * a.unbias);
    }
}

struct BnFinOp : Op {
    BnFinArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(a.C, 64)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "bn_finalize"; }
};

struct BnApplyArgs {
    const float* raw; int c_stride_raw; const float* scale_shift;
    const void* add0; const void* add1; void* y;
    long long P; int C; int c_stride; int act; float act_param;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = a.c_stride / VEC;                       // vectors per pixel row
    const long long nvec = a.P * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const T* add0 = reinterpret_cast<const T*>(a.add0);
    const T* add1 = reinterpret_cast<const T*>(a.add1);
    T* y = reinterpret_cast<T*>(a.y);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long pix = v / vpr;
        const int c0 = (int)(v - pix * vpr) * VEC;
        float o[VEC];
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const int c = c0 + q * 4;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f), sc = r, sh = r;
            if (c < a.C) {   // C and c_stride_raw are multiples of 4 by construction of the callers
                r = *reinterpret_cast<const float4*>(a.raw + pix * a.c_stride_raw + c);
                sc = *reinterpret_cast<const float4*>(a.scale_shift + c);
                sh = *reinterpret_cast<const float4*>(a.scale_shift + a.C + c);
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
        } else {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16_bits(o[0]) | ((unsigned)f32_to_bf16_bits(o[1]) << 16);
            pk.y = (unsigned)f32_to_bf16_bits(o[2]) | ((unsigned)f32_to_bf16_bits(o[3]) << 16);
            pk.z = (unsigned)f32_to_bf16_bits(o[4]) | ((unsigned)f32_to_bf16_bits(o[5]) << 16);
            pk.w = (unsigned)f32_to_bf16_bits(o[6]) | ((unsigned)f32_to_bf16_bits(o[7]) << 16);
            *reinterpret_cast<uint4*>(y + e) = pk;
        }
    }
}

struct BnApplyOp : Op {
    BnApplyArgs a; int dtype;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        long long nvec = a.P * (a.c_stride / vec);
        long long blocks = ceil_div(nvec, 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        if (dtype == V2V_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "bn_apply"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_bn_finalize(const float* partials, int32_t rows, int32_t C, int64_t count,
                               const float* gamma, const float* beta, float eps,
                               float* scale_shift, float* running_mean, float* running_var, float momentum,
                               void* stream) {
    if (!partials || !scale_shift || rows <= 0 || C <= 0 || count <= 0) { set_error("bn_finalize: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<BnFinOp>();
    BnFinArgs& a = op->a;
    a.partials = partials; a.rows = rows; a.C = C;
    a.inv_count = 1.0 / (double)count;
    a.unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.scale_shift = scale_shift; a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum;
    return submit(std::move(op), stream);
}

extern "C" int v2v_bn_apply(const float* raw, int32_t c_stride_raw, const float* scale_shift,
                            const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                            int32_t act, float act_param, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!raw || !scale_shift || !y || P <= 0) { set_error("bn_apply: bad argument"); return V2V_EINVAL; }
    if (c_stride % vec != 0 || C % 4 != 0 || c_stride_raw % 4 != 0 || C > c_stride || C > c_stride_raw) {
        set_error("bn_apply: channel counts must be multiples of 4 (C=%d stride=%d raw=%d)", C, c_stride, c_stride_raw);
        return V2V_EINVAL;
    }
    auto op = std::make_unique<BnApplyOp>();
    BnApplyArgs& a = op->a;
    a.raw = raw; a.c_stride_raw = c_stride_raw; a.scale_shift = scale_shift;
    a.add0 = add0; a.add1 = add1; a.y = y; a.P = P; a.C = C; a.c_stride = c_stride;
    a.act = act; a.act_param = act_param;
    op->dtype = dtype;
    return submit(std::move(op), stream);
}
