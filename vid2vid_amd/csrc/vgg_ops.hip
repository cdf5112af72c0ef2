// Pooling ops of the VGG19 perceptual loss (VGGLoss / Vgg19, models/networks.py:776-791, 840-870) and the planar
// one-hot writer behind `real_A_last` (models/vid2vid_model_G.py:209).  gfx950 only.  All of them are one read + one
// write of the tensor: HBM-bound, 16-byte lanes over the NHWC channel vector.
#include "v2v_internal.h"

namespace v2v {

static inline unsigned grid_for(long long n, int threads = 256, long long cap = 8192) {
    long long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

struct MaxPoolArgs {
    const void* x; const void* dy; void* y;      // forward: x -> y; backward: (dy, x) -> y (= dx)
    int N, H, W, OH, OW, c_stride;
};

// MaxPool2d(kernel 2, stride 2) of torchvision's VGG19 `features` (indices 4, 9, 18, 27): floor output size,
// window scan order (0,0) (0,1) (1,0) (1,1), a later element wins only if strictly greater (ATen max_pool2d:
// `val > maxval || isnan(val)`), so ties -- frequent behind a ReLU -- keep the FIRST element.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_nhwc_kernel(const MaxPoolArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const T* x = reinterpret_cast<const T*>(a.x);
    T* y = reinterpret_cast<T*>(a.y);
    const int vpr = a.c_stride / VEC;
    const long long total = (long long)a.N * a.OH * a.OW * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int v = (int)(e % vpr);
        long long t = e / vpr;
        const int ow = (int)(t % a.OW); t /= a.OW;
        const int oh = (int)(t % a.OH);
        const long long n = t / a.OH;
        float m[VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long base = (((n * a.H + 2 * oh + (k >> 1)) * a.W) + 2 * ow + (k & 1)) * a.c_stride + (long long)v * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float val = load_act(x, base + q);
                if (k == 0 || val > m[q] || val != val) m[q] = val;
            }
        }
        const long long ob = (((n * a.OH + oh) * a.OW) + ow) * a.c_stride + (long long)v * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q) store_act(y, ob + q, m[q]);
    }
}

// dX of the above: one thread per INPUT pixel vector; re-derives the window's argmax from x (first maximum) and
// routes dY to it; rows / columns outside every window (odd H or W) get zero.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_nhwc_bwd_kernel(const MaxPoolArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    T* dx = reinterpret_cast<T*>(a.y);
    const int vpr = a.c_stride / VEC;
    const long long total = (long long)a.N * a.H * a.W * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int v = (int)(e % vpr);
        long long t = e / vpr;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H);
        const long long n = t / a.H;
        const int oh = h >> 1, ow = w >> 1;
        const long long ib = (((n * a.H + h) * a.W) + w) * a.c_stride + (long long)v * VEC;
        if (oh >= a.OH || ow >= a.OW) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) store_act(dx, ib + q, 0.f);
            continue;
        }
        const int me = ((h & 1) << 1) | (w & 1);
        float m[VEC]; int arg[VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long base = (((n * a.H + 2 * oh + (k >> 1)) * a.W) + 2 * ow + (k & 1)) * a.c_stride + (long long)v * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float val = load_act(x, base + q);
                if (k == 0 || val > m[q] || val != val) { m[q] = val; arg[q] = k; }
            }
        }
        const long long ob = (((n * a.OH + oh) * a.OW) + ow) * a.c_stride + (long long)v * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q) store_act(dx, ib + q, arg[q] == me ? load_act(dy, ob + q) : 0.f);
    }
}

struct MaxPoolOp : Op {
    MaxPoolArgs a; int dtype; bool bwd;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        const long long n = (long long)a.N * (bwd ? (long long)a.H * a.W : (long long)a.OH * a.OW) * (a.c_stride / vec);
        if (!bwd) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL(maxpool2_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(maxpool2_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        } else {
            if (dtype == V2V_BF16) hipLaunchKernelGGL(maxpool2_nhwc_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(maxpool2_nhwc_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        }
        return check_launch();
    }
    const char* name() const override { return bwd ? "maxpool2_nhwc_backward" : "maxpool2_nhwc"; }
};

// AvgPool2d(2, stride 2, count_include_pad=False) without padding (VGGLoss.downsample, models/networks.py:782,785-786)
// on planar fp32 [planes][H][W]; floor output size.
struct AvgPool2Args { const float* x; float* y; long long planes; int H, W, OH, OW; };

__global__ __launch_bounds__(256) void avgpool2_planar_kernel(const AvgPool2Args a) {
    const long long total = a.planes * a.OH * a.OW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ow = (int)(e % a.OW);
        long long t = e / a.OW;
        const int oh = (int)(t % a.OH);
        const long long p = t / a.OH;
        const float* r0 = a.x + (p * a.H + 2 * oh) * a.W + 2 * ow;
        const float* r1 = r0 + a.W;
        a.y[e] = (r0[0] + r0[1] + r1[0] + r1[1]) * 0.25f;
    }
}

__global__ __launch_bounds__(256) void avgpool2_planar_bwd_kernel(const AvgPool2Args a) {   // x = dY, y = dX
    const long long total = a.planes * a.H * a.W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int w = (int)(e % a.W);
        long long t = e / a.W;
        const int h = (int)(t % a.H);
        const long long p = t / a.H;
        const int oh = h >> 1, ow = w >> 1;
        a.y[e] = (oh < a.OH && ow < a.OW) ? a.x[(p * a.OH + oh) * a.OW + ow] * 0.25f : 0.f;
    }
}

struct AvgPool2Op : Op {
    AvgPool2Args a; bool bwd;
    int launch(hipStream_t s) override {
        if (!bwd) hipLaunchKernelGGL(avgpool2_planar_kernel, dim3(grid_for(a.planes * a.OH * a.OW)), dim3(256), 0, s, a);
        else      hipLaunchKernelGGL(avgpool2_planar_bwd_kernel, dim3(grid_for(a.planes * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return bwd ? "avgpool2_planar_backward" : "avgpool2_planar"; }
};

// Planar fp32 one-hot of ONE label frame (+ instance-edge plane): out[c][h][w] = (label[h][w] == c), c < label_nc;
// out[label_nc][h][w] = edge(inst) with get_edges' 4-neighbour rule (models/base_model.py:146-152).  This is
// `real_A[0][0, -1]` of Vid2VidModelG.inference (models/vid2vid_model_G.py:209) produced straight from the label map:
// 4 bytes read per pixel, (label_nc + 1) * 4 written, fully coalesced along w (the NHWC -> NCHW unpack it replaces
// moved the same bytes with 72-byte-strided reads).
struct OneHotArgs { const void* labels; const void* inst; float* out; int H, W, label_nc; };

// LT / IT: float (the reference's float-encoded integers) or uint8 / int32 (v2v_onehot_planar_u8)
template <typename LT, typename IT>
__global__ __launch_bounds__(256) void onehot_planar_kernel(const OneHotArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const LT* labels = reinterpret_cast<const LT*>(a.labels);
    const IT* inst = reinterpret_cast<const IT*>(a.inst);
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += stride) {
        const int lab = (int)labels[p];
        for (int c = 0; c < a.label_nc; ++c) a.out[(long long)c * hw + p] = (c == lab) ? 1.f : 0.f;
        if (inst) {
            const int y = (int)(p / a.W), x = (int)(p - (long long)y * a.W);
            const IT v = inst[p];
            bool e = false;
            if (x > 0)       e |= inst[p - 1] != v;
            if (x + 1 < a.W) e |= inst[p + 1] != v;
            if (y > 0)       e |= inst[p - a.W] != v;
            if (y + 1 < a.H) e |= inst[p + a.W] != v;
            a.out[(long long)a.label_nc * hw + p] = e ? 1.f : 0.f;
        }
    }
}

struct OneHotOp : Op {
    OneHotArgs a; int in_u8 = 0;
    int launch(hipStream_t s) override {
        const dim3 g(grid_for((long long)a.H * a.W)), b(256);
        if (in_u8) hipLaunchKernelGGL((onehot_planar_kernel<unsigned char, int>), g, b, 0, s, a);
        else       hipLaunchKernelGGL((onehot_planar_kernel<float, float>), g, b, 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "onehot_planar"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_maxpool2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                                 int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!x || !y || N <= 0 || H < 2 || W < 2 || c_stride <= 0 || c_stride % vec) { set_error("maxpool2: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<MaxPoolOp>();
    op->a = MaxPoolArgs{x, nullptr, y, N, H, W, H / 2, W / 2, c_stride}; op->dtype = dtype; op->bwd = false;
    return submit(std::move(op), stream);
}

extern "C" int v2v_maxpool2_nhwc_backward(const void* dy, const void* x, void* dx, int32_t N, int32_t H, int32_t W,
                                          int32_t c_stride, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!dy || !x || !dx || N <= 0 || H < 2 || W < 2 || c_stride <= 0 || c_stride % vec) { set_error("maxpool2_backward: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<MaxPoolOp>();
    op->a = MaxPoolArgs{x, dy, dx, N, H, W, H / 2, W / 2, c_stride}; op->dtype = dtype; op->bwd = true;
    return submit(std::move(op), stream);
}

extern "C" int v2v_avgpool2_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
    if (!x || !y || planes <= 0 || H < 2 || W < 2) { set_error("avgpool2: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<AvgPool2Op>();
    op->a = AvgPool2Args{x, y, planes, H, W, H / 2, W / 2}; op->bwd = false;
    return submit(std::move(op), stream);
}

extern "C" int v2v_avgpool2_planar_backward(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, void* stream) {
    if (!dy || !dx || planes <= 0 || H < 2 || W < 2) { set_error("avgpool2_backward: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<AvgPool2Op>();
    op->a = AvgPool2Args{dy, dx, planes, H, W, H / 2, W / 2}; op->bwd = true;
    return submit(std::move(op), stream);
}

extern "C" int v2v_onehot_planar(const float* labels, const float* inst, float* out, int32_t H, int32_t W,
                                 int32_t label_nc, void* stream) {
    if (!labels || !out || H <= 0 || W <= 0 || label_nc <= 0) { set_error("onehot_planar: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<OneHotOp>();
    op->a = OneHotArgs{labels, inst, out, H, W, label_nc};
    return submit(std::move(op), stream);
}

extern "C" int v2v_onehot_planar_u8(const uint8_t* labels, const int32_t* inst, float* out, int32_t H, int32_t W,
                                    int32_t label_nc, void* stream) {
    if (!labels || !out || H <= 0 || W <= 0 || label_nc <= 0 || label_nc > 256) { set_error("onehot_planar_u8: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<OneHotOp>();
    op->a = OneHotArgs{labels, inst, out, H, W, label_nc}; op->in_u8 = 1;
    return submit(std::move(op), stream);
}
