// FlowNet2's three native ops, forward only (vid2vid runs FlowNet2 frozen under no_grad:
// models/flownet.py:18-26).  Written from the maths of the reference kernels, wave64-native:
//   correlation   correlation_cuda_kernel.cu:73-147  (+ output-size rule correlation_cuda.cc:25-38)
//   resample2d    resample2d_kernel.cu:15-64
//   channelnorm   channelnorm_kernel.cu:18-60
// The reference first copies both inputs into zero-padded NHWC scratch (channels_first,
// correlation_cuda_kernel.cu:46-70) and then reduces 32 channels per warp with shuffles;
// here padding is a bounds test and every lane owns whole output pixels (coalesced along x),
// so there is no scratch tensor and no cross-lane reduction at all.
#include "v2v_internal.h"

namespace v2v {

struct CorrArgs {
    const float* in1; const float* in2; float* out;
    int N, C, H, W, OH, OW, D;            // D = displacement_size
    int pad, ksize, krad, max_disp, s1, s2, drad;
};

// grid: (ceil(OW/32), OH * D, N); block (32, 8): thread (tx, ty) owns output x and the
// displacements ti = ty, ty+8, ... of row tj = blockIdx.y % D.
__global__ __launch_bounds__(256) void correlation_kernel(const CorrArgs a) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ox = blockIdx.x * 32 + tx;
    const int oy = blockIdx.y / a.D;
    const int tjn = blockIdx.y - oy * a.D;          // 0..D-1
    const int n = blockIdx.z;
    if (ox >= a.OW) return;
    const long long hw = (long long)a.H * a.W;
    const float* f1 = a.in1 + (long long)n * a.C * hw;
    const float* f2 = a.in2 + (long long)n * a.C * hw;
    // centre positions in the UNPADDED frame (reference works in the padded one)
    const int y1 = oy * a.s1 + a.max_disp - a.pad;
    const int x1 = ox * a.s1 + a.max_disp - a.pad;
    const int y2 = y1 + (tjn - a.drad) * a.s2;
    constexpr int MAXT = 8;                         // displacement columns per thread (D <= 64)
    float acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) acc[q] = 0.f;
    for (int j = -a.krad; j <= a.krad; ++j) {
        const int ya = y1 + j, yb = y2 + j;
        const bool yok = (unsigned)ya < (unsigned)a.H && (unsigned)yb < (unsigned)a.H;
        if (!yok) continue;                         // a zero-padded operand kills the product
        for (int i = -a.krad; i <= a.krad; ++i) {
            const int xa = x1 + i;
            if ((unsigned)xa >= (unsigned)a.W) continue;
            for (int c = 0; c < a.C; ++c) {
                const float v1 = f1[c * hw + (long long)ya * a.W + xa];
                const float* r2 = f2 + c * hw + (long long)yb * a.W;
#pragma unroll
                for (int q = 0; q < MAXT; ++q) {
                    const int tin = ty + q * 8;
                    if (tin < a.D) {
                        const int xb = xa + (tin - a.drad) * a.s2;
                        if ((unsigned)xb < (unsigned)a.W) acc[q] += v1 * r2[xb];
                    }
                }
            }
        }
    }
    const float nelems = (float)(a.ksize * a.ksize * a.C);
    const long long ohw = (long long)a.OH * a.OW;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const int tin = ty + q * 8;
        if (tin < a.D) {
            const int tc = tjn * a.D + tin;
            a.out[((long long)n * a.D * a.D + tc) * ohw + (long long)oy * a.OW + ox] = acc[q] / nelems;
        }
    }
}

struct CorrOp : Op {
    CorrArgs a;
    int launch(hipStream_t s) override {
        dim3 grid((unsigned)ceil_div(a.OW, 32), (unsigned)(a.OH * a.D), (unsigned)a.N);
        hipLaunchKernelGGL(correlation_kernel, grid, dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "correlation"; }
};

struct Resample2dArgs { const float* img; const float* flow; float* out; int N, C, H, W, OH, OW, ksize; };

__global__ __launch_bounds__(256) void resample2d_kernel(const Resample2dArgs a) {
    const long long ohw = (long long)a.OH * a.OW;
    const long long total = (long long)a.N * ohw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / ohw, pix = e - b * ohw;
        const int y = (int)(pix / a.OW), x = (int)(pix - (long long)y * a.OW);
        const float dx = a.flow[(b * 2 + 0) * ohw + pix], dy = a.flow[(b * 2 + 1) * ohw + pix];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        // clamps use the OUTPUT extent, exactly as resample2d_kernel.cu:46-49
        const int xL = max(min((int)floorf(xf), a.OW - 1), 0);
        const int xR = max(min((int)floorf(xf) + 1, a.OW - 1), 0);
        const int yT = max(min((int)floorf(yf), a.OH - 1), 0);
        const int yB = max(min((int)floorf(yf) + 1, a.OH - 1), 0);
        for (int c = 0; c < a.C; ++c) {
            const float* ip = a.img + (b * a.C + c) * (long long)a.H * a.W;
            float val = 0.f;
            for (int fy = 0; fy < a.ksize; ++fy)
                for (int fx = 0; fx < a.ksize; ++fx) {
                    val += (1.f - alpha) * (1.f - beta) * ip[(long long)(yT + fy) * a.W + xL + fx];
                    val += alpha * (1.f - beta) * ip[(long long)(yT + fy) * a.W + xR + fx];
                    val += (1.f - alpha) * beta * ip[(long long)(yB + fy) * a.W + xL + fx];
                    val += alpha * beta * ip[(long long)(yB + fy) * a.W + xR + fx];
                }
            a.out[(b * a.C + c) * ohw + pix] = val;
        }
    }
}

struct Resample2dOp : Op {
    Resample2dArgs a;
    int launch(hipStream_t s) override {
        long long blocks = ceil_div((long long)a.N * a.OH * a.OW, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(resample2d_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "resample2d"; }
};

struct ChannelNormArgs { const float* x; float* out; int N, C; long long hw; };

__global__ __launch_bounds__(256) void channelnorm_kernel(const ChannelNormArgs a) {
    const long long total = (long long)a.N * a.hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / a.hw, pix = e - b * a.hw;
        float r = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float v = a.x[(b * a.C + c) * a.hw + pix];
            r += v * v;
        }
        a.out[e] = sqrtf(r);
    }
}

struct ChannelNormOp : Op {
    ChannelNormArgs a;
    int launch(hipStream_t s) override {
        long long blocks = ceil_div((long long)a.N * a.hw, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(channelnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "channelnorm"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_correlation_out_size(int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                        int32_t max_displacement, int32_t stride1, int32_t stride2,
                                        int32_t* out_c, int32_t* out_h, int32_t* out_w) {
    // correlation_cuda.cc:25-38
    const int krad = (kernel_size - 1) / 2, border = krad + max_displacement;
    const int ph = H + 2 * pad_size, pw = W + 2 * pad_size;
    const int d = (max_displacement / stride2) * 2 + 1;
    if (out_c) *out_c = d * d;
    if (out_h) *out_h = (int)ceil_div(ph - 2 * border, stride1);
    if (out_w) *out_w = (int)ceil_div(pw - 2 * border, stride1);
    return 0;
}

extern "C" int v2v_correlation_forward(const float* in1, const float* in2, float* out,
                                       int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                       int32_t stride1, int32_t stride2, int32_t corr_type_multiply, void* stream) {
    if (!in1 || !in2 || !out || stride1 < 1 || stride2 < 1 || kernel_size < 1 || (kernel_size & 1) == 0) {
        set_error("correlation: bad argument"); return V2V_EINVAL;
    }
    if (corr_type_multiply != 1) { set_error("correlation: only corr_type_multiply=1 exists in the reference"); return V2V_EINVAL; }
    CorrArgs a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.N = N; a.C = C; a.H = H; a.W = W;
    a.pad = pad_size; a.ksize = kernel_size; a.krad = (kernel_size - 1) / 2; a.max_disp = max_displacement;
    a.s1 = stride1; a.s2 = stride2; a.drad = max_displacement / stride2; a.D = 2 * a.drad + 1;
    int oc;
    v2v_correlation_out_size(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &a.OH, &a.OW);
    if (a.D > 64 || a.OH <= 0 || a.OW <= 0) { set_error("correlation: unsupported geometry"); return V2V_EINVAL; }
    auto op = std::make_unique<CorrOp>();
    op->a = a;
    return submit(std::move(op), stream);
}

extern "C" int v2v_resample2d_forward(const float* img, const float* flow, float* out,
                                      int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                      int32_t kernel_size, void* stream) {
    if (!img || !flow || !out || kernel_size < 1) { set_error("resample2d: bad argument"); return V2V_EINVAL; }
    if (OH + kernel_size - 1 > H || OW + kernel_size - 1 > W) { set_error("resample2d: image smaller than flow"); return V2V_EINVAL; }
    auto op = std::make_unique<Resample2dOp>();
    op->a = Resample2dArgs{img, flow, out, N, C, H, W, OH, OW, kernel_size};
    return submit(std::move(op), stream);
}

extern "C" int v2v_channelnorm_forward(const float* x, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t norm_deg, void* stream) {
    if (!x || !out) { set_error("channelnorm: null"); return V2V_EINVAL; }
    if (norm_deg != 2) { set_error("channelnorm: the reference kernel implements norm_deg=2 only"); return V2V_EINVAL; }
    auto op = std::make_unique<ChannelNormOp>();
    op->a = ChannelNormArgs{x, out, N, C, (long long)H * W};
    return submit(std::move(op), stream);
}
