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
