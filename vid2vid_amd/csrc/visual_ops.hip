// On-GPU visualisation conversions: the reference converts every saved frame on the host with numpy
// (util/util.py:48-89: tensor2im, tensor2label + Colorize :197-212) after a full-size D2H copy of fp32 planes; at
// hundreds of frames per second that path, not the generator, would bound test.py.  Here the conversion runs on the
// device and only the uint8 HWC image (1/4 .. 1/48 of the bytes) crosses PCIe.  Integer-exact w.r.t. the reference's
// numpy arithmetic (float32 ops in the same order, truncating cast).
#include "v2v_internal.h"

namespace v2v {

static inline unsigned grid_for(long long n, int threads = 256, long long cap = 4096) {
    long long b = (n + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

struct Im2Args { const float* x; unsigned char* out; int C, H, W, normalize; };

// out[h][w][c] = uint8(clip(normalize ? (x + 1) / 2 * 255 : x * 255, 0, 255))      (util/util.py:62-69)
__global__ __launch_bounds__(256) void tensor2im_kernel(const Im2Args a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = hw * a.C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long pix = e / a.C;
        const int c = (int)(e - pix * a.C);
        const float x = a.x[(long long)c * hw + pix];
        float v = a.normalize ? ((x + 1.f) / 2.f) * 255.f : x * 255.f;
        v = fminf(fmaxf(v, 0.f), 255.f);           // np.clip; NaN -> 0 here, undefined in numpy's cast
        a.out[e] = (unsigned char)(int)v;          // astype(uint8): truncation
    }
}

struct Lab2Args { const float* x; unsigned char* out; const unsigned char* cmap; int C, H, W, n; };

// label = C > 1 ? argmax_c x[c] (first maximum, torch.max) : the stored value; out[h][w][:] = cmap[label] (0 beyond n)
// (util/util.py:73-87, Colorize :197-212)
__global__ __launch_bounds__(256) void tensor2label_kernel(const Lab2Args a) {
    const long long hw = (long long)a.H * a.W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += stride) {
        int lab = -1;
        if (a.C > 1) {
            float best = a.x[p];
            lab = 0;
            for (int c = 1; c < a.C; ++c) {
                const float v = a.x[(long long)c * hw + p];
                if (v > best) { best = v; lab = c; }
            }
        } else {
            const float v = a.x[p];
            const int iv = (int)v;
            lab = ((float)iv == v) ? iv : -1;      // Colorize compares `label == gray_image` in float
        }
        unsigned char r = 0, g = 0, b = 0;
        if (lab >= 0 && lab < a.n) { r = a.cmap[lab * 3]; g = a.cmap[lab * 3 + 1]; b = a.cmap[lab * 3 + 2]; }
        a.out[p * 3 + 0] = r; a.out[p * 3 + 1] = g; a.out[p * 3 + 2] = b;
    }
}

// tensor2flow (util/util.py:89-107): flow (2, H, W) -> HSV-coded RGB, the usual optical-flow picture.  The reference calls
// OpenCV (cv2.cartToPolar, cv2.normalize NORM_MINMAX, cv2.cvtColor HSV2RGB on uint8; a dependency that is not in the tree):
//   hsv[..., 0] = uint8(ang * 180 / pi / 2)         ang = atan2(fy, fx) in [0, 2 pi)
//   hsv[..., 1] = 255
//   hsv[..., 2] = uint8((mag - min) * 255 / (max - min))        (cv2.normalize; a constant image maps to 0)
//   rgb = HSV2RGB_8u: h' = H * 6 / 180, sector = floor(h'), f = h' - sector, (v, p, q, t) = (V, V(1-S), V(1-Sf), V(1-S(1-f))) with
//         S, V scaled by 1/255, channel table {v,t,p},{q,v,p},{p,v,t},{p,q,v},{t,p,v},{v,p,q}, result * 255 rounded to nearest.
// Two launches: the magnitude range (non-negative floats order like their bit patterns: integer atomics), then the colours.
struct FlowImArgs { const float* flow; unsigned char* out; unsigned* range; int H, W; };

__global__ __launch_bounds__(256) void flow_range_kernel(const FlowImArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned lo = 0x7f800000u, hi = 0u;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += stride) {
        const float fx = a.flow[p], fy = a.flow[hw + p];
        const unsigned m = __float_as_uint(sqrtf(fx * fx + fy * fy));
        lo = m < lo ? m : lo; hi = m > hi ? m : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(a.range, lo); atomicMax(a.range + 1, hi); }
}

__global__ __launch_bounds__(256) void flow_colour_kernel(const FlowImArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float mn = __uint_as_float(a.range[0]), mx = __uint_as_float(a.range[1]);
    const float scale = (mx - mn) > 1.1920929e-07f ? 255.f / (mx - mn) : 0.f;          // cv::normalize: scale = 0 below DBL_EPSILON
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += stride) {
        const float fx = a.flow[p], fy = a.flow[hw + p];
        float ang = atan2f(fy, fx);
        if (ang < 0.f) ang += 6.283185307179586f;
        const float mag = sqrtf(fx * fx + fy * fy);
        const int H8 = (int)(ang * 180.f / 3.14159265358979323846f / 2.f) & 255;     // numpy's float -> uint8 store: truncation
        const int V8 = (int)((mag - mn) * scale) & 255;
        const float v = (float)V8 * (1.f / 255.f);                                    // S = 255 -> s = 1
        float h = (float)H8 * (6.f / 180.f);
        int sector = (int)floorf(h);
        h -= (float)sector;
        sector = ((sector % 6) + 6) % 6;
        const float tab[4] = {v, 0.f, v * (1.f - h), v * h};                          // v, p = v(1-s), q = v(1-s f), t = v(1-s(1-f))
        const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};      // (b, g, r) indices
        const float b = tab[sd[sector][0]], g = tab[sd[sector][1]], r = tab[sd[sector][2]];
        a.out[p * 3 + 0] = (unsigned char)__float2int_rn(fminf(fmaxf(r * 255.f, 0.f), 255.f));
        a.out[p * 3 + 1] = (unsigned char)__float2int_rn(fminf(fmaxf(g * 255.f, 0.f), 255.f));
        a.out[p * 3 + 2] = (unsigned char)__float2int_rn(fminf(fmaxf(b * 255.f, 0.f), 255.f));
    }
}

struct FlowImOp : Op {
    FlowImArgs a;
    int launch(hipStream_t s) override {
        const unsigned init[2] = {0x7f800000u, 0u};
        if (hipMemcpyAsync(a.range, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return check_launch();
        const dim3 g(grid_for((long long)a.H * a.W, 256, 1024)), b(256);
        hipLaunchKernelGGL(flow_range_kernel, g, b, 0, s, a);
        hipLaunchKernelGGL(flow_colour_kernel, g, b, 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "tensor2flow"; }
};

struct Im2Op : Op {
    Im2Args a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(tensor2im_kernel, dim3(grid_for((long long)a.H * a.W * a.C)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "tensor2im"; }
};
struct Lab2Op : Op {
    Lab2Args a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(tensor2label_kernel, dim3(grid_for((long long)a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "tensor2label"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_tensor2im(const float* x, uint8_t* out, int32_t C, int32_t H, int32_t W, int32_t normalize, void* stream) {
    if (!x || !out || C < 1 || C > 3 || H < 1 || W < 1) { set_error("tensor2im: bad argument (1..3 planes)"); return V2V_EINVAL; }
    auto op = std::make_unique<Im2Op>();
    op->a = Im2Args{x, out, C, H, W, normalize};
    return submit(std::move(op), stream);
}

extern "C" int v2v_tensor2label(const float* x, uint8_t* out, const uint8_t* cmap, int32_t n_label, int32_t C, int32_t H, int32_t W,
                                void* stream) {
    if (!x || !out || !cmap || C < 1 || H < 1 || W < 1 || n_label < 1) { set_error("tensor2label: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<Lab2Op>();
    op->a = Lab2Args{x, out, cmap, C, H, W, n_label};
    return submit(std::move(op), stream);
}

extern "C" int v2v_tensor2flow(const float* flow, uint8_t* out, uint32_t* range_ws, int32_t H, int32_t W, void* stream) {
    if (!flow || !out || !range_ws || H < 1 || W < 1) { set_error("tensor2flow: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<FlowImOp>();
    op->a = FlowImArgs{flow, out, range_ws, H, W};
    return submit(std::move(op), stream);
}
