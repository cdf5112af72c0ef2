// HBM-bound helpers of the vid2vid hot path: input encoding, layout changes, pyramids and
// the warp-and-blend tail of the composite generators.  All are one-pass streaming kernels
// with coalesced accesses; none of them materialises the reference's intermediates
// (one-hot NCHW tensor, normalised grid, expanded masks).
#include "v2v_internal.h"

namespace v2v {

static inline unsigned grid_for(long long n, int threads = 256, long long cap = 4096) {
    long long b = ceil_div(n, threads);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// one 16-byte store of VEC consecutive channels (8 bf16 / 4 fp32) -- the layout kernels below used to leave through VEC separate
// 2- / 4-byte stores (pack_concat 0.93 ms, pack_nchw_to_nhwc 1.6 ms, unpack 1.1 ms per launch at 2048x1024: 8.5 % of a training
// chunk, profiles/r06_v14_train_kernel_stats.txt, for tensors that take a tenth of that at the HBM rate)
__device__ __forceinline__ void store_vec(bf16_t* y, long long e, const float (&v)[8]) {
    uint4 pk;
    pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(y + e) = pk;
}
__device__ __forceinline__ void store_vec(float* y, long long e, const float (&v)[4]) {
    *reinterpret_cast<float4*>(y + e) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void load_vec(const bf16_t* y, long long e, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(y + e);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u); v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
__device__ __forceinline__ void load_vec(const float* y, long long e, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(y + e);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// ---------------------------------------------------------------------------------------
// encode_input + get_edges + compute_mask
// ---------------------------------------------------------------------------------------
struct EncodeArgs {
    const void* labels; const void* inst; void* out; float* mask;
    int T, H, W, label_nc, c_stride; const int* fg; int n_fg;
};

// One thread per (pixel, 16-byte vector of output channels).  A vector usually spans at most two label frames (36 channels
// per frame, 4 or 8 channels per vector): their labels are loaded ONCE per thread, the instance-edge test runs only for a
// vector that holds an edge channel, and the vector leaves as ONE 16-byte store (round 1 issued 8 two-byte stores per
// thread and re-read the label per channel: 45 us for the 33 MB output at 512x256, 7x off the HBM rate).
// LT / IT: element types of the label / instance maps -- float (the reference's loader hands integers encoded as
// floats, data/temporal_dataset.py:60-70) or uint8 / int32 (4x less host-to-device traffic for labels).
template <typename T, typename LT, typename IT>
__global__ __launch_bounds__(256) void encode_labels_kernel(const EncodeArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = a.c_stride / VEC;
    const long long hw = (long long)a.H * a.W;
    const long long nvec = hw * vpr;
    const int per_frame = a.label_nc + (a.inst ? 1 : 0);
    const long long stride = (long long)gridDim.x * blockDim.x;
    const LT* labels = reinterpret_cast<const LT*>(a.labels);
    const IT* inst = reinterpret_cast<const IT*>(a.inst);
    T* out = reinterpret_cast<T*>(a.out);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long pix = v / vpr;
        const int c0 = (int)(v - pix * vpr) * VEC;
        const int t0 = c0 / per_frame, t1 = (c0 + VEC - 1) / per_frame;
        const int lab0 = t0 < a.T ? (int)labels[t0 * hw + pix] : -1;
        const int lab1 = (t1 != t0 && t1 < a.T) ? (int)labels[t1 * hw + pix] : -1;
        float o[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int ch = c0 + q;
            // a vector spans two frames at most when per_frame >= VEC - 1 (every label2city / face recipe); with fewer
            // channels per frame (label_nc <= 6 in bf16) it can span three or more: those middle frames' labels are
            // loaded on demand (they used to be attributed to t1 and silently encoded as 0)
            const int t = ch >= (t0 + 1) * per_frame ? (ch >= t1 * per_frame ? t1 : ch / per_frame) : t0;
            const int c = ch - t * per_frame;
            float val = 0.f;
            if (t < a.T) {
                if (c < a.label_nc) {
                    const int lab = t == t0 ? lab0 : (t == t1 ? lab1 : (int)labels[t * hw + pix]);
                    val = (lab == c) ? 1.f : 0.f;
                } else {
                    // instance-boundary edge: 4-neighbour inequality (models/base_model.py:146-152)
                    const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
                    const IT* ip = inst + t * hw;
                    const IT ctr = ip[pix];
                    bool e = false;
                    if (x > 0)       e = e || (ip[pix - 1] != ctr);
                    if (x < a.W - 1) e = e || (ip[pix + 1] != ctr);
                    if (y > 0)       e = e || (ip[pix - a.W] != ctr);
                    if (y < a.H - 1) e = e || (ip[pix + a.W] != ctr);
                    val = e ? 1.f : 0.f;
                }
            }
            o[q] = val;
        }
        const long long e0 = pix * a.c_stride + c0;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(out + e0) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            uint4 pk;                                  // 0.0 / 1.0 are exact in bf16
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
            pk.z = pack_bf16x2(o[4], o[5]);
            pk.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4*>(out + e0) = pk;
        }
        if (a.mask && c0 == 0) {
            // compute_mask (models/vid2vid_model_G.py:322-330) on the last frame
            const int lab = (int)labels[(long long)(a.T - 1) * hw + pix];
            float m = 0.f;
            for (int i = 0; i < a.n_fg; ++i) m += (a.fg[i] == lab) ? 1.f : 0.f;
            a.mask[pix] = fminf(fmaxf(m, 0.f), 1.f);
        }
    }
}

// encode_input + get_edges followed by ONE level of build_pyr (AvgPool2d(3, 2, 1, count_include_pad=False), base_model.py:122-134),
// straight from the label / instance maps (round 3): out[oy][ox][t * per + c] = (number of in-bounds window pixels whose label is c) /
// (number of in-bounds window pixels), the edge channel likewise with the 4-neighbour instance-boundary test -- bit for bit what
// avgpool_nhwc_kernel computes on the materialised encoding (window sums of 0 / 1 are exact integers, the same division), without
// the full-resolution one-hot tensor: at 2048 x 1024 that tensor is 537 MB written by encode_labels (0.45 ms) and read back by the
// pooling kernel (0.73 ms), and nothing else needs it once the finest scale's stems read the label maps themselves
// (csrc/onehot_stem.hip).  The full-resolution foreground mask (compute_mask) is written by the threads of channel vector 0.
template <typename T, typename LT, typename IT>
__global__ __launch_bounds__(256) void encode_labels_pooled_kernel(const EncodeArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int OH = (a.H - 1) / 2 + 1, OW = (a.W - 1) / 2 + 1;
    const unsigned vpr = (unsigned)(a.c_stride / VEC);
    const unsigned hw = (unsigned)a.H * (unsigned)a.W;                // < 2^31 (host check): 32-bit index arithmetic throughout
    const unsigned nvec = (unsigned)OH * (unsigned)OW * vpr;
    const int per_frame = a.label_nc + (a.inst ? 1 : 0);
    const unsigned stride = gridDim.x * blockDim.x;
    const LT* labels = reinterpret_cast<const LT*>(a.labels);
    const IT* inst = reinterpret_cast<const IT*>(a.inst);
    T* out = reinterpret_cast<T*>(a.out);
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const unsigned opix = v / vpr;
        const int c0 = (int)(v - opix * vpr) * VEC;
        const int oy = (int)(opix / (unsigned)OW), ox = (int)(opix - (unsigned)oy * (unsigned)OW);
        // frame / channel of each of the vector's channels: one division, then carries (a vector can span several frames when
        // a frame has fewer channels than the vector)
        int tq[VEC], cq[VEC];
        {
            int t = c0 / per_frame, c = c0 - t * per_frame;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                tq[q] = t; cq[q] = c;
                if (++c == per_frame) { c = 0; ++t; }
            }
        }
        const int t_first = tq[0], t_last = min(tq[VEC - 1], a.T - 1);
        // window counts as 8-bit fields of one 64-bit word (a window holds <= 9 pixels): per window pixel and frame ONE range test
        // and one shifted add instead of VEC compare-and-adds.  For frame t the vector's label channels are a contiguous run:
        // channel c sits at field qs[t] + (c - cs[t]) while cs[t] <= c < ce[t]; the frame's edge channel (if in the vector) at qe[t].
        unsigned long long acc = 0ull;
        int cnt = 0;
        const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 1, a.H - 1), x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 1, a.W - 1);
        for (int t = t_first; t <= t_last; ++t) {
            int qs = -1, cs = 0, ce = 0, qe = -1;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                if (tq[q] != t) continue;
                if (cq[q] < a.label_nc) { if (qs < 0) { qs = q; cs = cq[q]; } ce = cq[q] + 1; }
                else qe = q;
            }
            const LT* lp = labels + (unsigned)t * hw;
            const IT* ip = inst + (unsigned)t * hw;
            cnt = 0;
            for (int y = y0; y <= y1; ++y) {
                for (int x = x0; x <= x1; ++x) {
                    ++cnt;
                    const unsigned pix = (unsigned)y * (unsigned)a.W + (unsigned)x;
                    if (qs >= 0) {
                        const int lab = (int)lp[pix];
                        if (lab >= cs && lab < ce) acc += 1ull << (8 * (qs + lab - cs));
                    }
                    if (qe >= 0) {
                        const IT ctr = ip[pix];
                        bool e = false;
                        if (x > 0)       e = e || (ip[pix - 1] != ctr);
                        if (x < a.W - 1) e = e || (ip[pix + 1] != ctr);
                        if (y > 0)       e = e || (ip[pix - a.W] != ctr);
                        if (y < a.H - 1) e = e || (ip[pix + a.W] != ctr);
                        if (e) acc += 1ull << (8 * qe);
                    }
                }
            }
        }
        if (cnt == 0) cnt = (y1 - y0 + 1) * (x1 - x0 + 1);           // padding-only vectors: no frame visited
        float s[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) s[q] = (float)((acc >> (8 * q)) & 0xffull);
        const float fc = (float)cnt;
        T* const o = out + (size_t)opix * a.c_stride + c0;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(s[0] / fc, s[1] / fc, s[2] / fc, s[3] / fc);
        } else {
            uint4 pk;
            pk.x = pack_bf16x2(s[0] / fc, s[1] / fc);
            pk.y = pack_bf16x2(s[2] / fc, s[3] / fc);
            pk.z = pack_bf16x2(s[4] / fc, s[5] / fc);
            pk.w = pack_bf16x2(s[6] / fc, s[7] / fc);
            *reinterpret_cast<uint4*>(o) = pk;
        }
        if (a.mask && c0 == 0) {
            // compute_mask (models/vid2vid_model_G.py:322-330) on the last frame, FULL resolution: the 2 x 2 pixels this output pixel owns
            for (int yy = 2 * oy; yy < min(2 * oy + 2, a.H); ++yy)
                for (int xx = 2 * ox; xx < min(2 * ox + 2, a.W); ++xx) {
                    const int lab = (int)labels[(unsigned)(a.T - 1) * hw + (unsigned)yy * (unsigned)a.W + (unsigned)xx];
                    float m = 0.f;
                    for (int i = 0; i < a.n_fg; ++i) m += (a.fg[i] == lab) ? 1.f : 0.f;
                    a.mask[(unsigned)yy * (unsigned)a.W + (unsigned)xx] = fminf(fmaxf(m, 0.f), 1.f);
                }
        }
    }
}

// The same from the 1-byte label | edge << 7 codes (v2v_label_codes; 127 = no label plane) that the frame plan computes first for the
// gather-sum stems (round 5).  The kernel above walks its window with one dependent load after another -- a label load, five
// instance-map loads for an edge channel, a data-dependent add, the next pixel: ~30 us of serialized latency per thread, 394 us for
// the 1024x512 level of the 2048x1024 frame (HBM bound of its 117 MB output: 20 us).  Here the nine codes of a frame's window are nine
// INDEPENDENT byte loads from clamped coordinates (the edge tests were done once per pixel by v2v_label_codes), then arithmetic only.
// Same counts, same division: bit-identical output.  a.labels = the codes; a.inst != NULL only says "the frames have an edge channel".
template <typename T>
__global__ __launch_bounds__(256) void encode_codes_pooled_kernel(const EncodeArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int OH = (a.H - 1) / 2 + 1, OW = (a.W - 1) / 2 + 1;
    const unsigned vpr = (unsigned)(a.c_stride / VEC);
    const unsigned hw = (unsigned)a.H * (unsigned)a.W;
    const unsigned nvec = (unsigned)OH * (unsigned)OW * vpr;
    const int per_frame = a.label_nc + (a.inst ? 1 : 0);
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned char* const codes = reinterpret_cast<const unsigned char*>(a.labels);
    T* out = reinterpret_cast<T*>(a.out);
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const unsigned opix = v / vpr;
        const int c0 = (int)(v - opix * vpr) * VEC;
        const int oy = (int)(opix / (unsigned)OW), ox = (int)(opix - (unsigned)oy * (unsigned)OW);
        const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 1, a.H - 1), x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 1, a.W - 1);
        // the 3 x 3 window as clamped coordinates + validity: rows 2 oy - 1 .. 2 oy + 1 (the first may be outside, the last two may be)
        unsigned rowo[3], colo[3];
        bool rv[3], cv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int y = 2 * oy - 1 + j, x = 2 * ox - 1 + j;
            rv[j] = y >= 0 && y < a.H;
            cv[j] = x >= 0 && x < a.W;
            rowo[j] = (unsigned)min(max(y, 0), a.H - 1) * (unsigned)a.W;
            colo[j] = (unsigned)min(max(x, 0), a.W - 1);
        }
        const int t_first = c0 / per_frame, t_last = min((c0 + VEC - 1) / per_frame, a.T - 1);
        unsigned long long acc = 0ull;
        for (int t = t_first; t <= t_last; ++t) {
            const unsigned char* const cp = codes + (unsigned)t * hw;
            unsigned code[9];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) code[j * 3 + i] = cp[rowo[j] + colo[i]];
            const int base = t * per_frame - c0;                  // field of the frame's channel 0 inside this vector
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const unsigned cd = code[j * 3 + i];
                    const bool ok = rv[j] && cv[i];
                    const int lab = (int)(cd & 127u);
                    const int fl = base + lab;
                    if (ok && lab < a.label_nc && fl >= 0 && fl < VEC) acc += 1ull << (8 * fl);
                    const int fe = base + a.label_nc;
                    if (ok && a.inst != nullptr && (cd & 128u) && fe >= 0 && fe < VEC) acc += 1ull << (8 * fe);
                }
        }
        const int cnt = (y1 - y0 + 1) * (x1 - x0 + 1);
        float sv[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) sv[q] = (float)((acc >> (8 * q)) & 0xffull);
        const float fc = (float)cnt;
        T* const o = out + (size_t)opix * a.c_stride + c0;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(sv[0] / fc, sv[1] / fc, sv[2] / fc, sv[3] / fc);
        } else {
            uint4 pk;
            pk.x = pack_bf16x2(sv[0] / fc, sv[1] / fc);
            pk.y = pack_bf16x2(sv[2] / fc, sv[3] / fc);
            pk.z = pack_bf16x2(sv[4] / fc, sv[5] / fc);
            pk.w = pack_bf16x2(sv[6] / fc, sv[7] / fc);
            *reinterpret_cast<uint4*>(o) = pk;
        }
        if (a.mask && c0 == 0) {
            // compute_mask (models/vid2vid_model_G.py:322-330) on the last frame, FULL resolution: the 2 x 2 pixels this output pixel owns
            for (int yy = 2 * oy; yy < min(2 * oy + 2, a.H); ++yy)
                for (int xx = 2 * ox; xx < min(2 * ox + 2, a.W); ++xx) {
                    const int lab = (int)(codes[(unsigned)(a.T - 1) * hw + (unsigned)yy * (unsigned)a.W + (unsigned)xx] & 127u);
                    float m = 0.f;
                    for (int i = 0; i < a.n_fg; ++i) m += (a.fg[i] == lab) ? 1.f : 0.f;
                    a.mask[(unsigned)yy * (unsigned)a.W + (unsigned)xx] = fminf(fmaxf(m, 0.f), 1.f);
                }
        }
    }
}

struct EncodePooledOp : Op {
    EncodeArgs a; int dtype; int in_u8 = 0;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        const long long nvec = (long long)((a.H - 1) / 2 + 1) * ((a.W - 1) / 2 + 1) * (a.c_stride / vec);
        const dim3 g(grid_for(nvec, 256, 16384)), b(256);
        if (in_u8 == 2) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL((encode_codes_pooled_kernel<bf16_t>), g, b, 0, s, a);
            else                   hipLaunchKernelGGL((encode_codes_pooled_kernel<float>), g, b, 0, s, a);
        } else if (in_u8) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL((encode_labels_pooled_kernel<bf16_t, unsigned char, int>), g, b, 0, s, a);
            else                   hipLaunchKernelGGL((encode_labels_pooled_kernel<float, unsigned char, int>), g, b, 0, s, a);
        } else {
            if (dtype == V2V_BF16) hipLaunchKernelGGL((encode_labels_pooled_kernel<bf16_t, float, float>), g, b, 0, s, a);
            else                   hipLaunchKernelGGL((encode_labels_pooled_kernel<float, float, float>), g, b, 0, s, a);
        }
        return check_launch();
    }
    const char* name() const override { return "encode_labels_pooled"; }
};

struct EncodeOp : Op {
    EncodeArgs a; int dtype; int in_u8 = 0;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        const long long nvec = (long long)a.H * a.W * (a.c_stride / vec);
        const dim3 g(grid_for(nvec)), b(256);
        if (in_u8) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL((encode_labels_kernel<bf16_t, unsigned char, int>), g, b, 0, s, a);
            else                   hipLaunchKernelGGL((encode_labels_kernel<float, unsigned char, int>), g, b, 0, s, a);
        } else {
            if (dtype == V2V_BF16) hipLaunchKernelGGL((encode_labels_kernel<bf16_t, float, float>), g, b, 0, s, a);
            else                   hipLaunchKernelGGL((encode_labels_kernel<float, float, float>), g, b, 0, s, a);
        }
        return check_launch();
    }
    const char* name() const override { return "encode_labels"; }
};

// ---------------------------------------------------------------------------------------
// layout changes
// ---------------------------------------------------------------------------------------
struct PackArgs2 { const float* x; void* y; int N, C, H, W, c_stride; };

template <typename T>
__global__ __launch_bounds__(256) void pack_nchw_to_nhwc_kernel(const PackArgs2 a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = a.c_stride / VEC;
    const long long hw = (long long)a.H * a.W;
    const long long nvec = (long long)a.N * hw * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    T* y = reinterpret_cast<T*>(a.y);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        // consecutive threads -> consecutive pixels of one channel vector (coalesced reads)
        const long long pixg = v % ((long long)a.N * hw);
        const int cv = (int)(v / ((long long)a.N * hw));
        const long long n = pixg / hw, pix = pixg - n * hw;
        float vals[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int c = cv * VEC + q;
            vals[q] = c < a.C ? a.x[(n * a.C + c) * hw + pix] : 0.f;
        }
        store_vec(y, pixg * a.c_stride + cv * VEC, vals);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_nhwc_to_nchw_kernel(const PackArgs2 a) {
    // one 16-byte channel vector of one pixel per thread, consecutive threads on consecutive pixels: every plane is written in
    // coalesced runs (the first version read one 2-byte value per thread at a stride of a whole pixel)
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = (a.C + VEC - 1) / VEC;
    const long long hw = (long long)a.H * a.W;
    const long long npix = (long long)a.N * hw;
    const long long nvec = npix * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const T* y = reinterpret_cast<const T*>(a.y);
    float* x = const_cast<float*>(a.x);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long pixg = v % npix;
        const int cv = (int)(v / npix);
        const long long n = pixg / hw, pix = pixg - n * hw;
        float vals[VEC];
        load_vec(y, pixg * a.c_stride + cv * VEC, vals);          // c_stride is a multiple of VEC: the vector stays inside the pixel
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int c = cv * VEC + q;
            if (c < a.C) x[(n * a.C + c) * hw + pix] = vals[q];
        }
    }
}

struct LayoutOp : Op {
    PackArgs2 a; int dtype; bool pack;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        const long long n = (long long)a.N * a.H * a.W * (pack ? (a.c_stride / vec) : ((a.C + vec - 1) / vec));
        if (pack) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL(pack_nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(pack_nchw_to_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        } else {
            if (dtype == V2V_BF16) hipLaunchKernelGGL(unpack_nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(unpack_nhwc_to_nchw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        }
        return check_launch();
    }
    const char* name() const override { return pack ? "pack_nchw_to_nhwc" : "unpack_nhwc_to_nchw"; }
};

// ---------------------------------------------------------------------------------------
// AvgPool2d(3, stride 2, padding 1, count_include_pad=False)
// ---------------------------------------------------------------------------------------
struct PoolArgs { const void* x; void* y; long long planes; int N, H, W, OH, OW, c_stride; };

__global__ __launch_bounds__(256) void avgpool_planar_kernel(const PoolArgs a) {
    const float* x = reinterpret_cast<const float*>(a.x);
    float* y = reinterpret_cast<float*>(a.y);
    const long long total = a.planes * a.OH * a.OW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ox = (int)(e % a.OW);
        const long long t = e / a.OW;
        const int oy = (int)(t % a.OH);
        const long long pl = t / a.OH;
        const float* xp = x + pl * (long long)a.H * a.W;
        float s = 0.f; int cnt = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= a.H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= a.W) continue;
                s += xp[(long long)iy * a.W + ix];
                ++cnt;
            }
        }
        y[e] = s / (float)cnt;
    }
}

// One thread per (output pixel, 16-byte channel vector): 9 vector loads, one vector store (round 3: the element-per-thread form
// above it replaced needed 0.73 ms for the 1024 x 2048 x 128 label encoding, 0.9 TB/s).  Same arithmetic per element: window
// values summed in row-major order in fp32, divided by the number of in-bounds taps.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const PoolArgs a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const T* x = reinterpret_cast<const T*>(a.x);
    T* y = reinterpret_cast<T*>(a.y);
    const int vpr = a.c_stride / VEC;
    const long long total = (long long)a.N * a.OH * a.OW * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % vpr) * VEC;
        long long t = e / vpr;
        const int ox = (int)(t % a.OW); t /= a.OW;
        const int oy = (int)(t % a.OH);
        const long long n = t / a.OH;
        float s[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) s[q] = 0.f;
        int cnt = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= a.H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= a.W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + ((n * a.H + iy) * a.W + ix) * a.c_stride + c);
                if constexpr (VEC == 4) {
                    s[0] += __uint_as_float(v.x); s[1] += __uint_as_float(v.y); s[2] += __uint_as_float(v.z); s[3] += __uint_as_float(v.w);
                } else {
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s[2 * k] += __uint_as_float(w[k] << 16); s[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
                }
                ++cnt;
            }
        }
        const float fc = (float)cnt;
        T* o = y + (((n * a.OH + oy) * a.OW + ox) * (long long)a.c_stride + c);
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(s[0] / fc, s[1] / fc, s[2] / fc, s[3] / fc);
        } else {
            uint4 pk;
            pk.x = pack_bf16x2(s[0] / fc, s[1] / fc);
            pk.y = pack_bf16x2(s[2] / fc, s[3] / fc);
            pk.z = pack_bf16x2(s[4] / fc, s[5] / fc);
            pk.w = pack_bf16x2(s[6] / fc, s[7] / fc);
            *reinterpret_cast<uint4*>(o) = pk;
        }
    }
}

struct PoolOp : Op {
    PoolArgs a; int dtype; bool planar;
    int launch(hipStream_t s) override {
        if (planar) {
            hipLaunchKernelGGL(avgpool_planar_kernel, dim3(grid_for(a.planes * a.OH * a.OW)), dim3(256), 0, s, a);
        } else {
            const long long n = (long long)a.N * a.OH * a.OW * (a.c_stride / (dtype == V2V_BF16 ? 8 : 4));
            if (dtype == V2V_BF16) hipLaunchKernelGGL(avgpool_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(avgpool_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        }
        return check_launch();
    }
    const char* name() const override { return planar ? "avgpool3s2_planar" : "avgpool3s2_nhwc"; }
};

// ---------------------------------------------------------------------------------------
// grid_sample(bilinear, border) driven by a pixel-unit flow, and the composite blend
// ---------------------------------------------------------------------------------------
// Follows BaseNetwork.resample (models/networks.py:108-115): grid = linspace(-1,1) + flow /
// ((W-1)/2, (H-1)/2), then F.grid_sample.  gx/gy hold torch.linspace(-1,1,W/H) so the base
// grid is bit-identical to the reference's get_grid (models/networks.py:79-93).
__device__ __forceinline__ float unnormalize(float coord, int size, int align_corners) {
    if (align_corners) return ((coord + 1.f) / 2.f) * (float)(size - 1);
    return ((coord + 1.f) * (float)size - 1.f) / 2.f;
}

__device__ __forceinline__ void bilinear_setup(float fx, float fy, float gxv, float gyv, int H, int W, int ac,
                                               int& x0, int& y0, float& wx, float& wy) {
    const float nx = gxv + fx / (((float)W - 1.0f) / 2.0f);
    const float ny = gyv + fy / (((float)H - 1.0f) / 2.0f);
    float ix = unnormalize(nx, W, ac), iy = unnormalize(ny, H, ac);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));     // padding_mode='border'
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    x0 = (int)fx0; y0 = (int)fy0;
    wx = ix - fx0; wy = iy - fy0;
}

__device__ __forceinline__ float bilinear_fetch(const float* img, int H, int W, int x0, int y0, float wx, float wy) {
    // corners outside the image contribute 0 (their weight is 0 after the border clip)
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool xin = x1 < W, yin = y1 < H;
    const float nw = img[(long long)y0 * W + x0];
    const float ne = xin ? img[(long long)y0 * W + x1] : 0.f;
    const float sw = yin ? img[(long long)y1 * W + x0] : 0.f;
    const float se = (xin && yin) ? img[(long long)y1 * W + x1] : 0.f;
    return nw * ((1.f - wx) * (1.f - wy)) + ne * (wx * (1.f - wy)) + sw * ((1.f - wx) * wy) + se * (wx * wy);
}

struct WarpArgs {
    float* img_raw; const float* flow; const float* weight; const float* prev; const float* fg; const float* mask;
    float* img_final; float* img_warp; const float* gx; const float* gy;
    int N, C, H, W, align_corners;
};

__global__ __launch_bounds__(256) void warp_blend_kernel(const WarpArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.N * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long n = e / hw, pix = e - n * hw;
        const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
        int x0 = 0, y0 = 0; float wx = 0.f, wy = 0.f;
        const bool do_warp = a.flow != nullptr;
        float wgt = 1.f;
        if (do_warp) {
            const float fx = a.flow[(n * 2 + 0) * hw + pix], fy = a.flow[(n * 2 + 1) * hw + pix];
            bilinear_setup(fx, fy, a.gx[x], a.gy[y], a.H, a.W, a.align_corners, x0, y0, wx, wy);
            wgt = a.weight[n * hw + pix];
        }
        const float m = a.fg ? a.mask[n * hw + pix] : 0.f;
        for (int c = 0; c < a.C; ++c) {
            const long long o = (n * a.C + c) * hw + pix;
            float raw = a.img_raw[o];
            float fin = raw;
            if (do_warp) {
                const float wv = bilinear_fetch(a.prev + (n * a.C + c) * hw, a.H, a.W, x0, y0, wx, wy);
                if (a.img_warp) a.img_warp[o] = wv;
                fin = raw * wgt + wv * (1.f - wgt);
            }
            if (a.fg) {
                const float f = a.fg[o];
                fin = f * m + fin * (1.f - m);
                raw = f * m + raw * (1.f - m);
                a.img_raw[o] = raw;
            }
            a.img_final[o] = fin;
        }
    }
}

struct WarpOp : Op {
    WarpArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(warp_blend_kernel, dim3(grid_for((long long)a.N * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "warp_blend"; }
};

struct ResampleArgs { const float* img; const float* flow; float* out; const float* gx; const float* gy; int N, C, H, W, align_corners; };

__global__ __launch_bounds__(256) void resample_flow_kernel(const ResampleArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.N * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long n = e / hw, pix = e - n * hw;
        const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
        int x0, y0; float wx, wy;
        bilinear_setup(a.flow[(n * 2 + 0) * hw + pix], a.flow[(n * 2 + 1) * hw + pix], a.gx[x], a.gy[y],
                       a.H, a.W, a.align_corners, x0, y0, wx, wy);
        for (int c = 0; c < a.C; ++c)
            a.out[(n * a.C + c) * hw + pix] = bilinear_fetch(a.img + (n * a.C + c) * hw, a.H, a.W, x0, y0, wx, wy);
    }
}

struct ResampleOp : Op {
    ResampleArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(resample_flow_kernel, dim3(grid_for((long long)a.N * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "resample_flow"; }
};

// ---------------------------------------------------------------------------------------
// y = a + b on NHWC activations (coarse-to-fine feature sums, models/networks.py:299,305,319)
// ---------------------------------------------------------------------------------------
struct AddArgs { const void* a; const void* b; void* y; long long nvec; };

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const AddArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const T* a = reinterpret_cast<const T*>(p.a);
    const T* b = reinterpret_cast<const T*>(p.b);
    T* y = reinterpret_cast<T*>(p.y);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < p.nvec; v += stride) {
        const uint4 ua = *reinterpret_cast<const uint4*>(a + v * VEC);
        const uint4 ub = *reinterpret_cast<const uint4*>(b + v * VEC);
        uint4 uo;
        if constexpr (VEC == 4) {
            uo.x = __float_as_uint(__uint_as_float(ua.x) + __uint_as_float(ub.x));
            uo.y = __float_as_uint(__uint_as_float(ua.y) + __uint_as_float(ub.y));
            uo.z = __float_as_uint(__uint_as_float(ua.z) + __uint_as_float(ub.z));
            uo.w = __float_as_uint(__uint_as_float(ua.w) + __uint_as_float(ub.w));
        } else {
            auto add2 = [](unsigned x, unsigned z) {
                const float lo = bf16_bits_to_f32((unsigned short)(x & 0xffffu)) + bf16_bits_to_f32((unsigned short)(z & 0xffffu));
                const float hi = bf16_bits_to_f32((unsigned short)(x >> 16)) + bf16_bits_to_f32((unsigned short)(z >> 16));
                return pack_bf16x2(lo, hi);
            };
            uo.x = add2(ua.x, ub.x); uo.y = add2(ua.y, ub.y); uo.z = add2(ua.z, ub.z); uo.w = add2(ua.w, ub.w);
        }
        *reinterpret_cast<uint4*>(y + v * VEC) = uo;
    }
}

struct AddOp : Op {
    AddArgs a; int dtype;
    int launch(hipStream_t s) override {
        if (dtype == V2V_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(a.nvec)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(a.nvec)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "add_nhwc"; }
};

// ---------------------------------------------------------------------------------------
// compute_mask (models/vid2vid_model_G.py:322-330) on an NHWC (possibly pooled) label tensor
// ---------------------------------------------------------------------------------------
struct FgMaskArgs { const void* x; float* mask; long long P; int c_stride, base_ch; const int* fg; int n_fg; };

template <typename T>
__global__ __launch_bounds__(256) void fg_mask_kernel(const FgMaskArgs a) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < a.P; p += stride) {
        float m = 0.f;
        for (int i = 0; i < a.n_fg; ++i) m += load_act(x, p * a.c_stride + a.base_ch + a.fg[i]);
        a.mask[p] = fminf(fmaxf(m, 0.f), 1.f);
    }
}

struct FgMaskOp : Op {
    FgMaskArgs a; int dtype;
    int launch(hipStream_t s) override {
        if (dtype == V2V_BF16) hipLaunchKernelGGL(fg_mask_kernel<bf16_t>, dim3(grid_for(a.P)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(fg_mask_kernel<float>, dim3(grid_for(a.P)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "fg_mask_nhwc"; }
};


// ---------------------------------------------------------------------------------------
// training-path helpers: channel concat while packing, channel-offset unpack (its backward),
// reflection-pad fold (backward of nn.ReflectionPad2d), AvgPool backward, warp/blend backward
// ---------------------------------------------------------------------------------------
struct Pack2Args { const float* x0; const float* x1; void* y; int N, C0, C1, H, W, c_stride; float scale1; };

// cat([x0, x1], dim=1) (vid2vid_model_D.py:169-170,185-187) written straight to NHWC
template <typename T>
__global__ __launch_bounds__(256) void pack_concat_kernel(const Pack2Args a) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = a.c_stride / VEC;
    const long long hw = (long long)a.H * a.W;
    const long long npix = (long long)a.N * hw;
    const long long nvec = npix * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    T* y = reinterpret_cast<T*>(a.y);
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long pixg = v % npix;
        const int cv = (int)(v / npix);
        const long long n = pixg / hw, pix = pixg - n * hw;
        float vals[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int c = cv * VEC + q;
            float val = 0.f;
            if (c < a.C0) val = a.x0[(n * a.C0 + c) * hw + pix];
            else if (c < a.C0 + a.C1) val = a.x1[(n * a.C1 + (c - a.C0)) * hw + pix] * a.scale1;
            vals[q] = val;
        }
        store_vec(y, pixg * a.c_stride + cv * VEC, vals);
    }
}

struct Pack2Op : Op {
    Pack2Args a; int dtype;
    int launch(hipStream_t s) override {
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        const long long n = (long long)a.N * a.H * a.W * (a.c_stride / vec);
        if (dtype == V2V_BF16) hipLaunchKernelGGL(pack_concat_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(pack_concat_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "pack_concat_nhwc"; }
};

struct UnpackAtArgs { const void* y; float* x; int N, C, H, W, c_stride, c_off; };

template <typename T>
__global__ __launch_bounds__(256) void unpack_at_kernel(const UnpackAtArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.N * a.C * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const T* y = reinterpret_cast<const T*>(a.y);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long pix = e % hw;
        const long long nc = e / hw;
        const long long n = nc / a.C;
        const int c = (int)(nc - n * a.C);
        a.x[e] = load_act(y, (n * hw + pix) * a.c_stride + a.c_off + c);
    }
}

struct UnpackAtOp : Op {
    UnpackAtArgs a; int dtype;
    int launch(hipStream_t s) override {
        const long long n = (long long)a.N * a.C * a.H * a.W;
        if (dtype == V2V_BF16) hipLaunchKernelGGL(unpack_at_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(unpack_at_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "unpack_channels_nchw"; }
};

// dX[h][w] = sum of the padded-gradient entries that mirror onto (h, w)
struct FoldArgs { const void* xp; void* x; int N, H, W, pad, c_stride; };

__device__ __forceinline__ int fold_sources(int h, int H, int p, int* src) {
    int n = 0;
    src[n++] = h + p;
    if (h >= 1 && h <= p) src[n++] = p - h;
    if (h >= H - 1 - p && h <= H - 2) src[n++] = 2 * (H - 1) - h + p;
    return n;
}

template <typename T>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const FoldArgs a) {
    const T* xp = reinterpret_cast<const T*>(a.xp);
    T* x = reinterpret_cast<T*>(a.x);
    const int HP = a.H + 2 * a.pad, WP = a.W + 2 * a.pad;
    const long long total = (long long)a.N * a.H * a.W * a.c_stride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % a.c_stride);
        long long t = e / a.c_stride;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H);
        const long long n = t / a.H;
        int hs[3], ws[3];
        const int nh = fold_sources(h, a.H, a.pad, hs), nw = fold_sources(w, a.W, a.pad, ws);
        float s = 0.f;
        for (int i = 0; i < nh; ++i)
            for (int j = 0; j < nw; ++j)
                s += load_act(xp, ((n * HP + hs[i]) * WP + ws[j]) * a.c_stride + c);
        store_act(x, e, s);
    }
}

// The same fold, one 16-byte channel vector per thread (round 6): the scalar kernel above moves 2 bytes per lane behind three
// 64-bit divisions per element -- 13-23 us per launch, 116 launches per 512x256 training chunk on the serial chain of the backward
// pass (2.7 ms, profiles/r06_v7_train_kernel_stats.txt) for tensors that take 2 us at the HBM rate.  Same sources in the same
// order, summed in fp32 and rounded once: bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void reflect_fold_vec_kernel(const FoldArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const T* xp = reinterpret_cast<const T*>(a.xp);
    T* x = reinterpret_cast<T*>(a.x);
    const int HP = a.H + 2 * a.pad, WP = a.W + 2 * a.pad;
    const int vpp = a.c_stride / VEC;                                  // vectors per pixel
    const long long total = (long long)a.N * a.H * a.W * vpp;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long pix = e / vpp;
        const int cv = (int)(e - pix * vpp) * VEC;
        const int w = (int)(pix % a.W);
        const long long t = pix / a.W;
        const int h = (int)(t % a.H);
        const long long n = t / a.H;
        int hs[3], ws[3];
        const int nh = fold_sources(h, a.H, a.pad, hs), nw = fold_sources(w, a.W, a.pad, ws);
        float sum[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) sum[q] = 0.f;
        for (int i = 0; i < nh; ++i)
            for (int j = 0; j < nw; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(xp + ((n * HP + hs[i]) * WP + ws[j]) * a.c_stride + cv);
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (sizeof(T) == 2) { sum[2 * q] += __uint_as_float(u[q] << 16); sum[2 * q + 1] += __uint_as_float(u[q] & 0xffff0000u); }
                    else sum[q] += __uint_as_float(u[q]);
                }
            }
        uint4 o;
        if (sizeof(T) == 2) {
            o.x = pack_bf16x2(sum[0], sum[1]); o.y = pack_bf16x2(sum[2], sum[3]);
            o.z = pack_bf16x2(sum[4 % VEC], sum[5 % VEC]); o.w = pack_bf16x2(sum[6 % VEC], sum[7 % VEC]);
        } else {
            o.x = __float_as_uint(sum[0]); o.y = __float_as_uint(sum[1]); o.z = __float_as_uint(sum[2]); o.w = __float_as_uint(sum[3]);
        }
        *reinterpret_cast<uint4*>(x + pix * a.c_stride + cv) = o;
    }
}

struct FoldOp : Op {
    FoldArgs a; int dtype;
    int launch(hipStream_t s) override {
        const long long n = (long long)a.N * a.H * a.W * a.c_stride;
        const int vec = dtype == V2V_BF16 ? 8 : 4;
        if (a.c_stride % vec == 0 && (((uintptr_t)a.xp | (uintptr_t)a.x) & 15) == 0) {
            if (dtype == V2V_BF16) hipLaunchKernelGGL(reflect_fold_vec_kernel<bf16_t>, dim3(grid_for(n / vec)), dim3(256), 0, s, a);
            else                   hipLaunchKernelGGL(reflect_fold_vec_kernel<float>, dim3(grid_for(n / vec)), dim3(256), 0, s, a);
            return check_launch();
        }
        if (dtype == V2V_BF16) hipLaunchKernelGGL(reflect_fold_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "reflect_pad_fold"; }
};

__device__ __forceinline__ int pool_cnt(int o, int size) {   // valid taps of window o along one axis
    int c = 0;
    for (int d = -1; d <= 1; ++d) { const int i = 2 * o + d; c += (i >= 0 && i < size) ? 1 : 0; }
    return c;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_nhwc_bwd_kernel(const PoolArgs a) {
    // a.x = dY [N][OH][OW][cs], a.y = dX [N][H][W][cs]
    const T* dy = reinterpret_cast<const T*>(a.x);
    T* dx = reinterpret_cast<T*>(a.y);
    const long long total = (long long)a.N * a.H * a.W * a.c_stride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % a.c_stride);
        long long t = e / a.c_stride;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H);
        const long long n = t / a.H;
        float s = 0.f;
        const int oh0 = h / 2, oh1 = (h + 1) / 2, ow0 = w / 2, ow1 = (w + 1) / 2;   // windows containing h / w
        for (int oh = oh0; oh <= oh1; ++oh) {
            if (oh >= a.OH) continue;
            const int ch = pool_cnt(oh, a.H);
            for (int ow = ow0; ow <= ow1; ++ow) {
                if (ow >= a.OW) continue;
                s += load_act(dy, ((n * a.OH + oh) * a.OW + ow) * a.c_stride + c) / (float)(ch * pool_cnt(ow, a.W));
            }
        }
        store_act(dx, e, s);
    }
}

struct PoolBwdOp : Op {
    PoolArgs a; int dtype;
    int launch(hipStream_t s) override {
        const long long n = (long long)a.N * a.H * a.W * a.c_stride;
        if (dtype == V2V_BF16) hipLaunchKernelGGL(avgpool_nhwc_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(avgpool_nhwc_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "avgpool3s2_nhwc_backward"; }
};

// d(ix)/d(flow_x) through get_grid + unnormalize, and the border clip's gradient mask
// (ATen grid_sampler: clip_coordinates_set_grad -- borders count as out of bounds)
__device__ __forceinline__ void bilinear_setup_grad(float fx, float fy, float gxv, float gyv, int H, int W, int ac,
                                                    int& x0, int& y0, float& wx, float& wy, float& gmx, float& gmy) {
    const float nx = gxv + fx / (((float)W - 1.0f) / 2.0f);
    const float ny = gyv + fy / (((float)H - 1.0f) / 2.0f);
    float ix = unnormalize(nx, W, ac), iy = unnormalize(ny, H, ac);
    const float sx = (ac ? ((float)(W - 1) / 2.f) : ((float)W / 2.f)) / (((float)W - 1.0f) / 2.0f);
    const float sy = (ac ? ((float)(H - 1) / 2.f) : ((float)H / 2.f)) / (((float)H - 1.0f) / 2.0f);
    gmx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : sx;
    gmy = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : sy;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    x0 = (int)fx0; y0 = (int)fy0;
    wx = ix - fx0; wy = iy - fy0;
}

struct WarpBwdArgs {
    const float* d_final; const float* d_rawout;           // incoming gradients (d_rawout may be NULL)
    const float* raw; const float* flow; const float* weight; const float* prev; const float* fg; const float* mask;
    const float* gx; const float* gy;
    float* d_raw; float* d_flow; float* d_weight; float* d_prev; float* d_fg;   // d_prev: pre-zeroed, atomically accumulated
    int N, C, H, W, align_corners;
};

__global__ __launch_bounds__(256) void warp_blend_bwd_kernel(const WarpBwdArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.N * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long n = e / hw, pix = e - n * hw;
        const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
        const bool do_warp = a.flow != nullptr;
        int x0 = 0, y0 = 0; float wx = 0.f, wy = 0.f, gmx = 0.f, gmy = 0.f, wgt = 1.f;
        if (do_warp) {
            bilinear_setup_grad(a.flow[(n * 2 + 0) * hw + pix], a.flow[(n * 2 + 1) * hw + pix], a.gx[x], a.gy[y],
                                a.H, a.W, a.align_corners, x0, y0, wx, wy, gmx, gmy);
            wgt = a.weight[n * hw + pix];
        }
        const float m = a.fg ? a.mask[n * hw + pix] : 0.f;
        const int x1 = x0 + 1, y1 = y0 + 1;
        const bool xin = x1 < a.W, yin = y1 < a.H;
        float dwsum = 0.f, dix = 0.f, diy = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const long long o = (n * a.C + c) * hw + pix;
            const float df = a.d_final[o];
            const float dr = a.d_rawout ? a.d_rawout[o] : 0.f;
            if (a.d_fg) a.d_fg[o] = m * (df + dr);
            const float dpre = (1.f - m) * df;               // gradient of raw*w + warp*(1-w)
            float draw = (1.f - m) * dr;
            if (do_warp) {
                const float* img = a.prev + (n * a.C + c) * hw;
                const float nw = img[(long long)y0 * a.W + x0];
                const float ne = xin ? img[(long long)y0 * a.W + x1] : 0.f;
                const float sw = yin ? img[(long long)y1 * a.W + x0] : 0.f;
                const float se = (xin && yin) ? img[(long long)y1 * a.W + x1] : 0.f;
                const float wv = nw * ((1.f - wx) * (1.f - wy)) + ne * (wx * (1.f - wy)) + sw * ((1.f - wx) * wy) + se * (wx * wy);
                const float rawv = a.raw[o];
                dwsum += dpre * (rawv - wv);
                const float dwarp = dpre * (1.f - wgt);
                dix += dwarp * ((ne - nw) * (1.f - wy) + (se - sw) * wy);
                diy += dwarp * ((sw - nw) * (1.f - wx) + (se - ne) * wx);
                if (a.d_prev) {
                    float* dp = a.d_prev + (n * a.C + c) * hw;
                    atomicAdd(dp + (long long)y0 * a.W + x0, dwarp * (1.f - wx) * (1.f - wy));
                    if (xin) atomicAdd(dp + (long long)y0 * a.W + x1, dwarp * wx * (1.f - wy));
                    if (yin) atomicAdd(dp + (long long)y1 * a.W + x0, dwarp * (1.f - wx) * wy);
                    if (xin && yin) atomicAdd(dp + (long long)y1 * a.W + x1, dwarp * wx * wy);
                }
                draw += dpre * wgt;
            } else {
                draw += dpre;
            }
            a.d_raw[o] = draw;
        }
        if (do_warp) {
            a.d_weight[n * hw + pix] = dwsum;
            a.d_flow[(n * 2 + 0) * hw + pix] = dix * gmx;
            a.d_flow[(n * 2 + 1) * hw + pix] = diy * gmy;
        }
    }
}

struct WarpBwdOp : Op {
    WarpBwdArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(warp_blend_bwd_kernel, dim3(grid_for((long long)a.N * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "warp_blend_backward"; }
};

struct ResampleBwdArgs { const float* d_out; const float* img; const float* flow; const float* gx; const float* gy;
                         float* d_img; float* d_flow; int N, C, H, W, align_corners; };

__global__ __launch_bounds__(256) void resample_flow_bwd_kernel(const ResampleBwdArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.N * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long n = e / hw, pix = e - n * hw;
        const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
        int x0, y0; float wx, wy, gmx, gmy;
        bilinear_setup_grad(a.flow[(n * 2 + 0) * hw + pix], a.flow[(n * 2 + 1) * hw + pix], a.gx[x], a.gy[y],
                            a.H, a.W, a.align_corners, x0, y0, wx, wy, gmx, gmy);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const bool xin = x1 < a.W, yin = y1 < a.H;
        float dix = 0.f, diy = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float* img = a.img + (n * a.C + c) * hw;
            const float g = a.d_out[(n * a.C + c) * hw + pix];
            const float nw = img[(long long)y0 * a.W + x0];
            const float ne = xin ? img[(long long)y0 * a.W + x1] : 0.f;
            const float sw = yin ? img[(long long)y1 * a.W + x0] : 0.f;
            const float se = (xin && yin) ? img[(long long)y1 * a.W + x1] : 0.f;
            dix += g * ((ne - nw) * (1.f - wy) + (se - sw) * wy);
            diy += g * ((sw - nw) * (1.f - wx) + (se - ne) * wx);
            if (a.d_img) {
                float* dp = a.d_img + (n * a.C + c) * hw;
                atomicAdd(dp + (long long)y0 * a.W + x0, g * (1.f - wx) * (1.f - wy));
                if (xin) atomicAdd(dp + (long long)y0 * a.W + x1, g * wx * (1.f - wy));
                if (yin) atomicAdd(dp + (long long)y1 * a.W + x0, g * (1.f - wx) * wy);
                if (xin && yin) atomicAdd(dp + (long long)y1 * a.W + x1, g * wx * wy);
            }
        }
        if (a.d_flow) {
            a.d_flow[(n * 2 + 0) * hw + pix] = dix * gmx;
            a.d_flow[(n * 2 + 1) * hw + pix] = diy * gmy;
        }
    }
}

struct ResampleBwdOp : Op {
    ResampleBwdArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(resample_flow_bwd_kernel, dim3(grid_for((long long)a.N * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "resample_flow_backward"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_encode_labels(const float* labels, const float* inst, void* out, float* mask,
                                 int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                                 const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    const int need = T * (label_nc + (inst ? 1 : 0));
    if (!labels || !out || c_stride % vec != 0 || need > c_stride || (mask && n_fg > 0 && !fg_labels_dev)) {
        set_error("encode_labels: bad argument"); return V2V_EINVAL;
    }
    if (((uintptr_t)out & 15) != 0) { set_error("encode_labels: output must be 16-byte aligned"); return V2V_EINVAL; }
    auto op = std::make_unique<EncodeOp>();
    op->a = EncodeArgs{labels, inst, out, mask, T, H, W, label_nc, c_stride, fg_labels_dev, n_fg};
    op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_encode_labels_u8(const uint8_t* labels, const int32_t* inst, void* out, float* mask,
                                    int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                                    const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    const int need = T * (label_nc + (inst ? 1 : 0));
    if (!labels || !out || c_stride % vec != 0 || need > c_stride || label_nc > 256 || (mask && n_fg > 0 && !fg_labels_dev) ||
        ((uintptr_t)out & 15) != 0 || ((uintptr_t)inst & 3) != 0) {
        set_error("encode_labels_u8: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<EncodeOp>();
    op->a = EncodeArgs{labels, inst, out, mask, T, H, W, label_nc, c_stride, fg_labels_dev, n_fg};
    op->dtype = dtype; op->in_u8 = 1;
    return submit(std::move(op), stream);
}

extern "C" int v2v_encode_labels_pooled(const void* labels, const void* inst, void* out, float* mask,
                                        int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                                        const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, int32_t maps_u8, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    const int need = T * (label_nc + (inst ? 1 : 0));
    if (!labels || !out || T < 1 || H < 1 || W < 1 || label_nc < 1 || c_stride % vec != 0 || need > c_stride || (maps_u8 && label_nc > 256) ||
        maps_u8 < 0 || maps_u8 > 2 || (maps_u8 == 2 && label_nc > 126) ||
        (mask && n_fg > 0 && !fg_labels_dev) || ((uintptr_t)out & 15) != 0 || (maps_u8 != 2 && ((uintptr_t)inst & 3) != 0) ||
        (long long)T * H * W >= (1ll << 31) || (long long)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (c_stride / vec) >= (1ll << 31)) {
        set_error("encode_labels_pooled: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<EncodePooledOp>();
    op->a = EncodeArgs{labels, inst, out, mask, T, H, W, label_nc, c_stride, fg_labels_dev, n_fg};
    op->dtype = dtype; op->in_u8 = maps_u8;
    return submit(std::move(op), stream);
}

// ---------------------------------------------------------------------------------------
// bf16x3: an fp32 activation as three bf16 channel groups [hi | lo | hi], hi = bf16(x), lo = bf16(x - hi)   (round 3)
// ---------------------------------------------------------------------------------------
// A convolution of this tensor with the weights laid out [hi(W) | hi(W) | lo(W)] along the input channels is
//     hi(x) hi(W) + lo(x) hi(W) + hi(x) lo(W)  =  x W  up to the dropped lo(x) lo(W) term (2^-18 relative) and the 2^-17 residues
// -- near-fp32 products on the bf16 matrix pipe (16x the fp32-input MFMA rate at 3x the K extent) with NO change to any
// convolution kernel: the K extent simply triples.  fp32 accumulation as everywhere.  Used by the fp32 engine's "x3" mode for
// the 3x3 ResnetBlock convolutions (74 % of the frame's FLOP); the statistics, the norm and the activations stay fp32.
struct SplitX3Args { const float* x; unsigned short* y; long long P; int C, cs_in, cs_out; };

__global__ __launch_bounds__(256) void split_x3_kernel(const SplitX3Args a) {
    const int vpr = a.C >> 2;                                   // float4 per pixel
    const long long total = a.P * vpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long p = e / vpr;
        const int c = (int)(e - p * vpr) * 4;
        const float4 v = *reinterpret_cast<const float4*>(a.x + p * a.cs_in + c);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hi[q] = f32_to_bf16_bits(f[q]);
            lo[q] = f32_to_bf16_bits(f[q] - bf16_bits_to_f32(hi[q]));       // exact difference, then rounded: |residue| <= 2^-17 |x|
        }
        const uint2 vh = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
        const uint2 vl = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
        unsigned short* o = a.y + p * a.cs_out + c;
        *reinterpret_cast<uint2*>(o) = vh;
        *reinterpret_cast<uint2*>(o + a.C) = vl;
        *reinterpret_cast<uint2*>(o + 2 * a.C) = vh;
    }
}

struct SplitX3Op : Op {
    SplitX3Args a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(split_x3_kernel, dim3(grid_for(a.P * (a.C >> 2))), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "split_x3"; }
};

extern "C" int v2v_split_x3(const float* x, void* y, int64_t pixels, int32_t C, int32_t cs_in, int32_t cs_out, void* stream) {
    if (!x || !y || pixels < 1 || C < 4 || C % 4 != 0 || cs_in < C || cs_in % 4 != 0 || cs_out < 3 * C || cs_out % 4 != 0) {
        set_error("split_x3: C must be a multiple of 4, cs_in >= C, cs_out >= 3 C"); return V2V_EINVAL;
    }
    auto op = std::make_unique<SplitX3Op>();
    op->a = SplitX3Args{x, reinterpret_cast<unsigned short*>(y), pixels, C, cs_in, cs_out};
    return submit(std::move(op), stream);
}

extern "C" int v2v_pack_nchw_to_nhwc(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                     int32_t c_stride, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!x || !y || c_stride % vec != 0 || C > c_stride) { set_error("pack: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<LayoutOp>();
    op->a = PackArgs2{x, y, N, C, H, W, c_stride}; op->dtype = dtype; op->pack = true;
    return submit(std::move(op), stream);
}

extern "C" int v2v_unpack_nhwc_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t c_stride, int32_t dtype, void* stream) {
    if (!x || !y || C > c_stride) { set_error("unpack: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<LayoutOp>();
    op->a = PackArgs2{y, const_cast<void*>(x), N, C, H, W, c_stride}; op->dtype = dtype; op->pack = false;
    return submit(std::move(op), stream);
}

extern "C" int v2v_avgpool3s2_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
    if (!x || !y) { set_error("avgpool: null"); return V2V_EINVAL; }
    auto op = std::make_unique<PoolOp>();
    op->a = PoolArgs{x, y, planes, 1, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 0}; op->planar = true; op->dtype = V2V_F32;
    return submit(std::move(op), stream);
}

extern "C" int v2v_avgpool3s2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                                   int32_t dtype, void* stream) {
    if (!x || !y || c_stride % (dtype == V2V_BF16 ? 8 : 4) != 0) { set_error("avgpool: null pointer or a channel stride that is not whole 16-byte vectors"); return V2V_EINVAL; }
    auto op = std::make_unique<PoolOp>();
    op->a = PoolArgs{x, y, 0, N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, c_stride}; op->planar = false; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_warp_blend(float* img_raw, const float* flow, const float* weight, const float* prev,
                              const float* fg, const float* mask, float* img_final, float* img_warp,
                              const float* gx, const float* gy,
                              int32_t N, int32_t C, int32_t H, int32_t W, int32_t align_corners, void* stream) {
    if (!img_raw || !img_final || (flow && (!weight || !prev || !gx || !gy)) || (fg && !mask)) {
        set_error("warp_blend: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<WarpOp>();
    op->a = WarpArgs{img_raw, flow, weight, prev, fg, mask, img_final, img_warp, gx, gy, N, C, H, W, align_corners};
    return submit(std::move(op), stream);
}

extern "C" int v2v_resample_flow(const float* img, const float* flow, float* out, const float* gx, const float* gy,
                                 int32_t N, int32_t C, int32_t H, int32_t W, int32_t align_corners, void* stream) {
    if (!img || !flow || !out || !gx || !gy) { set_error("resample_flow: null"); return V2V_EINVAL; }
    auto op = std::make_unique<ResampleOp>();
    op->a = ResampleArgs{img, flow, out, gx, gy, N, C, H, W, align_corners};
    return submit(std::move(op), stream);
}

extern "C" int v2v_add_nhwc(const void* a, const void* b, void* y, int64_t n_elems, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!a || !b || !y || n_elems % vec != 0) { set_error("add: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<AddOp>();
    op->a = AddArgs{a, b, y, n_elems / vec}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_fg_mask_nhwc(const void* x, float* mask, int64_t P, int32_t c_stride, int32_t base_ch,
                                const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream) {
    if (!x || !mask || !fg_labels_dev || n_fg <= 0) { set_error("fg_mask: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<FgMaskOp>();
    op->a = FgMaskArgs{x, mask, P, c_stride, base_ch, fg_labels_dev, n_fg}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_pack_concat_nhwc(const float* x0, int32_t C0, const float* x1, int32_t C1, float scale1, void* y,
                                    int32_t N, int32_t H, int32_t W, int32_t c_stride, int32_t dtype, void* stream) {
    const int vec = dtype == V2V_BF16 ? 8 : 4;
    if (!x0 || !y || (C1 > 0 && !x1) || c_stride % vec != 0 || C0 + C1 > c_stride) { set_error("pack_concat: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<Pack2Op>();
    op->a = Pack2Args{x0, x1, y, N, C0, C1 > 0 ? C1 : 0, H, W, c_stride, scale1}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_unpack_channels_nchw(const void* y, float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                                        int32_t c_stride, int32_t c_offset, int32_t dtype, void* stream) {
    if (!x || !y || c_offset < 0 || c_offset + C > c_stride) { set_error("unpack_channels: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<UnpackAtOp>();
    op->a = UnpackAtArgs{y, x, N, C, H, W, c_stride, c_offset}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_reflect_pad_fold(const void* xp, void* x, int32_t N, int32_t H, int32_t W, int32_t pad,
                                    int32_t c_stride, int32_t dtype, void* stream) {
    if (!xp || !x || pad < 0 || pad >= H || pad >= W) { set_error("reflect_pad_fold: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<FoldOp>();
    op->a = FoldArgs{xp, x, N, H, W, pad, c_stride}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_avgpool3s2_nhwc_backward(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                                            int32_t dtype, void* stream) {
    if (!dy || !dx) { set_error("avgpool_backward: null"); return V2V_EINVAL; }
    auto op = std::make_unique<PoolBwdOp>();
    op->a = PoolArgs{dy, dx, 0, N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, c_stride}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_warp_blend_backward(const float* d_final, const float* d_rawout, const float* raw, const float* flow,
                                       const float* weight, const float* prev, const float* fg, const float* mask,
                                       const float* gx, const float* gy, float* d_raw, float* d_flow, float* d_weight,
                                       float* d_prev, float* d_fg, int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t align_corners, void* stream) {
    if (!d_final || !d_raw || (flow && (!raw || !weight || !prev || !gx || !gy || !d_flow || !d_weight)) || (fg && (!mask || !d_fg))) {
        set_error("warp_blend_backward: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<WarpBwdOp>();
    op->a = WarpBwdArgs{d_final, d_rawout, raw, flow, weight, prev, fg, mask, gx, gy, d_raw, d_flow, d_weight, d_prev,
                        fg ? d_fg : nullptr, N, C, H, W, align_corners};
    return submit(std::move(op), stream);
}

extern "C" int v2v_resample_flow_backward(const float* d_out, const float* img, const float* flow, const float* gx,
                                          const float* gy, float* d_img, float* d_flow, int32_t N, int32_t C,
                                          int32_t H, int32_t W, int32_t align_corners, void* stream) {
    if (!d_out || !img || !flow || !gx || !gy || (!d_img && !d_flow)) { set_error("resample_flow_backward: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<ResampleBwdOp>();
    op->a = ResampleBwdArgs{d_out, img, flow, gx, gy, d_img, d_flow, N, C, H, W, align_corners};
    return submit(std::move(op), stream);
}
