// 7x7 / stride 1 / reflection-pad-3 stem convolution over ONE-HOT label input as a weight GATHER-SUM (gfx950).
//
// encode_input (models/vid2vid_model_G.py:86-112) turns T = n_frames_G label maps into T x (label_nc one-hot planes + one
// instance-edge plane); the label tower and the foreground tower of CompositeGenerator start with
// ReflectionPad2d(3) + Conv2d(T*(label_nc+1), ngf, 7) on that tensor (models/networks.py:128-133, 153-156).  Dense, that is
// 108 x 49 MACs per output value -- 178 + 89 GFLOP of the 512x256 frame's 2115, and 337 us at the head of the frame's
// critical path (profiles/r02_a8_frame_timeline_serialized.txt).  But per pixel and tap exactly ONE label channel per
// frame is 1, so
//
//     out[p][co] = bias[co] + sum_{tap} sum_{t < T} ( W[co][t*per + label_t(p + tap)][tap]
//                                                     + edge_t(p + tap) * W[co][t*per + label_nc][tap] )
//
// is T gathered weight rows (plus a rare edge row) per tap: 36x fewer operations, exact in fp32.  The one-hot
// tensor is never read -- for n_scales_spatial = 1 it is not even materialised.
//
// Workgroup = 256 threads = TH x TW = 8 x 32 output pixels, thread = pixel, COUT fp32 accumulators in registers.
// Per tap the [T*per][COUT] weight slab goes to LDS (register-staged so that rows can be PADDED by 16 bytes: lanes that
// read different rows then hit different banks; lanes that read the same row -- the common case, labels are piecewise
// constant -- broadcast), double buffered; labels and edges of the tile + 3-pixel halo (reflection applied) sit in LDS as
// bytes.  Epilogue: raw fp32 NHWC output + the per-tile (sum, sum^2) statistics row of the training-mode norm that
// follows, reduced over the 64 lanes by recursive halving (deterministic, no atomics).
#include "v2v_internal.h"

namespace v2v {

struct OneHotConvArgs {
    const void* labels; const void* inst;     // [T][H][W] float or uint8 / int32
    const void* table;                        // [49][T*per][COUT] activation dtype
    const float* bias;                        // [cout] or NULL
    float* out; float* stats;                 // raw fp32 NHWC [H][W][cout_stride]; [tiles][cout][2] or NULL
    int T, H, W, label_nc, per, cout, cout_stride, tiles_w, in_u8;
};

constexpr int OS_TH = 8, OS_TW = 32, OS_PH = OS_TH + 6, OS_PW = OS_TW + 6;

template <typename T, int COUT>
__global__ __launch_bounds__(256) void onehot_conv7x7_kernel(const OneHotConvArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int ROWB = COUT * ES + 16;                   // padded row: 4 banks of skew per row
    constexpr int VPR = COUT * ES / 16;                    // 16-byte vectors per row
    constexpr int EPV = 16 / ES;                           // elements per vector
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = a.T * a.per;
    const int slab = rows * ROWB;
    char* const tb0 = smem;
    char* const tb1 = smem + slab;
    unsigned char* const lab_s = reinterpret_cast<unsigned char*>(smem + 2 * slab);          // [T][PH][PW]
    unsigned char* const edg_s = lab_s + a.T * OS_PH * OS_PW;                                 // [T][PH][PW]

    const int tid = threadIdx.x;
    const int ty = blockIdx.x / a.tiles_w, tx = blockIdx.x - ty * a.tiles_w;
    const int oh0 = ty * OS_TH, ow0 = tx * OS_TW;
    const int py = tid >> 5, px = tid & 31;
    const int H = a.H, W = a.W;
    const long long hw = (long long)H * W;

    // ---- labels / edges of the tile + halo, reflection applied (ReflectionPad2d(3) of the encoded tensor) ----
    for (int e = tid; e < a.T * OS_PH * OS_PW; e += 256) {
        const int t = e / (OS_PH * OS_PW);
        const int rem = e - t * OS_PH * OS_PW;
        const int qy = rem / OS_PW, qx = rem - qy * OS_PW;
        int y = oh0 + qy - 3, x = ow0 + qx - 3;
        y = y < 0 ? -y : y;  y = y >= H ? 2 * H - 2 - y : y;
        x = x < 0 ? -x : x;  x = x >= W ? 2 * W - 2 - x : x;
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);              // tile overhang beyond the mirror: any valid pixel, output is masked
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const long long p = (long long)t * hw + (long long)y * W + x;
        int lab;
        bool edge = false;
        if (a.in_u8) {
            lab = reinterpret_cast<const unsigned char*>(a.labels)[p];
            if (a.inst) {
                const int* ip = reinterpret_cast<const int*>(a.inst) + (long long)t * hw;
                const long long q = (long long)y * W + x;
                const int ctr = ip[q];
                if (x > 0)     edge |= ip[q - 1] != ctr;
                if (x < W - 1) edge |= ip[q + 1] != ctr;
                if (y > 0)     edge |= ip[q - W] != ctr;
                if (y < H - 1) edge |= ip[q + W] != ctr;
            }
        } else {
            lab = (int)reinterpret_cast<const float*>(a.labels)[p];
            if (a.inst) {
                const float* ip = reinterpret_cast<const float*>(a.inst) + (long long)t * hw;
                const long long q = (long long)y * W + x;
                const float ctr = ip[q];
                if (x > 0)     edge |= ip[q - 1] != ctr;
                if (x < W - 1) edge |= ip[q + 1] != ctr;
                if (y > 0)     edge |= ip[q - W] != ctr;
                if (y < H - 1) edge |= ip[q + W] != ctr;
            }
        }
        lab_s[e] = (unsigned char)((unsigned)lab < (unsigned)a.label_nc ? lab : 255);       // 255: no plane is hot
        edg_s[e] = edge ? 1 : 0;
    }

    // ---- weight slab staging: [rows][COUT] of one tap -> padded LDS rows ----
    const int nvec = rows * VPR;
    auto stage = [&](int tap, char* dst) {
        const char* src = reinterpret_cast<const char*>(a.table) + (long long)tap * rows * COUT * ES;
        for (int v = tid; v < nvec; v += 256) {
            const int r = v / VPR, j = v - r * VPR;
            const uint4 val = *reinterpret_cast<const uint4*>(src + (long long)v * 16);
            *reinterpret_cast<uint4*>(dst + r * ROWB + j * 16) = val;
        }
    };
    stage(0, tb0);

    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;

    auto add_row = [&](const char* rowp) {
#pragma unroll
        for (int j = 0; j < VPR; ++j) {
            const uint4 v = *reinterpret_cast<const uint4*>(rowp + j * 16);
            if constexpr (ES == 4) {
                acc[j * 4 + 0] += __uint_as_float(v.x); acc[j * 4 + 1] += __uint_as_float(v.y);
                acc[j * 4 + 2] += __uint_as_float(v.z); acc[j * 4 + 3] += __uint_as_float(v.w);
            } else {
                acc[j * 8 + 0] += __uint_as_float(v.x << 16); acc[j * 8 + 1] += __uint_as_float(v.x & 0xffff0000u);
                acc[j * 8 + 2] += __uint_as_float(v.y << 16); acc[j * 8 + 3] += __uint_as_float(v.y & 0xffff0000u);
                acc[j * 8 + 4] += __uint_as_float(v.z << 16); acc[j * 8 + 5] += __uint_as_float(v.z & 0xffff0000u);
                acc[j * 8 + 6] += __uint_as_float(v.w << 16); acc[j * 8 + 7] += __uint_as_float(v.w & 0xffff0000u);
            }
        }
    };

    for (int tap = 0; tap < 49; ++tap) {
        __syncthreads();                                   // slab `tap` staged (and labels, first round); buffer of tap-1 free
        char* const cur = (tap & 1) ? tb1 : tb0;
        if (tap + 1 < 49) stage(tap + 1, (tap & 1) ? tb0 : tb1);
        const int dy = tap / 7, dx = tap - dy * 7;
        const int q = (py + dy) * OS_PW + (px + dx);
        for (int t = 0; t < a.T; ++t) {
            const int lab = lab_s[t * OS_PH * OS_PW + q];
            if (lab != 255) add_row(cur + (t * a.per + lab) * ROWB);
            if (a.inst && edg_s[t * OS_PH * OS_PW + q]) add_row(cur + (t * a.per + a.label_nc) * ROWB);
        }
    }

    // ---- epilogue: bias, raw fp32 NHWC store, per-tile statistics ----
    const int oh = oh0 + py, ow = ow0 + px;
    const bool valid = oh < H && ow < W;
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] += c < a.cout ? a.bias[c] : 0.f;
    }
    if (valid) {
        float* op = a.out + ((long long)oh * W + ow) * a.cout_stride;
        if (a.cout == COUT && (a.cout_stride & 3) == 0) {
#pragma unroll
            for (int j = 0; j < COUT / 4; ++j)
                *reinterpret_cast<float4*>(op + j * 4) = make_float4(acc[j * 4], acc[j * 4 + 1], acc[j * 4 + 2], acc[j * 4 + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < COUT; ++c)
                if (c < a.cout) op[c] = acc[c];
        }
    }
    if (a.stats == nullptr) return;
    // recursive halving over the 64 lanes: after step m each lane keeps half of its channel range, summed with its partner's
    // copy of that half; after 6 steps lane l owns COUT / 64 channels, summed over the wave in a fixed tree order
    float s1[COUT / 2], s2[COUT / 2];
    const int lane = tid & 63;
    {
        const bool up = (lane & 32) != 0;
#pragma unroll
        for (int c = 0; c < COUT / 2; ++c) {
            const float lo = valid ? acc[c] : 0.f, hi = valid ? acc[c + COUT / 2] : 0.f;
            const float keep = up ? hi : lo, send = up ? lo : hi;
            const float got = __shfl_xor(send, 32);
            const float got2 = __shfl_xor(send * send, 32);
            s1[c] = keep + got;
            s2[c] = keep * keep + got2;
        }
    }
    int width = COUT / 2;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        if (width >= 2) {
            const bool up = (lane & m) != 0;
            const int half = width / 2;
#pragma unroll
            for (int c = 0; c < COUT / 4; ++c) {
                if (c < half) {
                    const float k1 = up ? s1[c + half] : s1[c], k2 = up ? s2[c + half] : s2[c];
                    const float t1 = up ? s1[c] : s1[c + half], t2 = up ? s2[c] : s2[c + half];
                    s1[c] = k1 + __shfl_xor(t1, m);
                    s2[c] = k2 + __shfl_xor(t2, m);
                }
            }
            width = half;
        } else {                                           // fewer channels than lanes left: plain butterfly on the one value
            s1[0] += __shfl_xor(s1[0], m);
            s2[0] += __shfl_xor(s2[0], m);
        }
    }
    // channel owned by this lane's surviving slot k (k < width): the halving picked the upper half at every step whose
    // lane bit was set, halves being COUT/2, COUT/4, ...
    __syncthreads();                                       // the table buffers are free: per-wave partials live there
    float* red = reinterpret_cast<float*>(smem);           // [4 waves][COUT][2]
    {
        int base = 0, span = COUT;
        for (int m = 32; m >= 1 && span > 1; m >>= 1) {
            span >>= 1;
            if (lane & m) base += span;
        }
        // with COUT >= 64 every lane ends with span = COUT / 64 >= 1 channels [base, base + span)
        const int wv = tid >> 6;
        constexpr int LEFT = COUT / 64 > 0 ? COUT / 64 : 1;
        const bool owner = COUT >= 64 || (lane & ((64 / COUT) - 1)) == 0;     // COUT < 64: duplicates after the butterfly tail
#pragma unroll
        for (int k = 0; k < LEFT; ++k)
            if (owner) { red[((wv * COUT) + base + k) * 2] = s1[k]; red[((wv * COUT) + base + k) * 2 + 1] = s2[k]; }
    }
    __syncthreads();
    if (tid < COUT && tid < a.cout) {
        const float t1 = ((red[(0 * COUT + tid) * 2] + red[(1 * COUT + tid) * 2]) + red[(2 * COUT + tid) * 2]) + red[(3 * COUT + tid) * 2];
        const float t2 = ((red[(0 * COUT + tid) * 2 + 1] + red[(1 * COUT + tid) * 2 + 1]) + red[(2 * COUT + tid) * 2 + 1]) + red[(3 * COUT + tid) * 2 + 1];
        float* dst = a.stats + ((long long)blockIdx.x * a.cout + tid) * 2;
        dst[0] = t1;
        dst[1] = t2;
    }
}

struct OneHotConvOp : Op {
    OneHotConvArgs a; int dtype, coutp, tiles;
    template <typename T, int COUT> int go(hipStream_t s) {
        const size_t lds = (size_t)2 * a.T * a.per * (COUT * sizeof(T) + 16) + (size_t)2 * a.T * OS_PH * OS_PW;
        auto kern = onehot_conv7x7_kernel<T, COUT>;
        static bool attr_done = false;
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096);
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), (lds + 15) / 16 * 16, s, a);
        return check_launch();
    }
    int launch(hipStream_t s) override {
        if (dtype == V2V_BF16) return coutp == 128 ? go<bf16_t, 128>(s) : go<bf16_t, 64>(s);
        return coutp == 128 ? go<float, 128>(s) : go<float, 64>(s);
    }
    const char* name() const override { return "onehot_conv7x7"; }
};

// weights [cout][cin][7][7] fp32 -> table [49][cin][coutp] of the activation dtype (pad columns zero)
struct OneHotPackArgs { const float* w; void* tab; int cin, cout, coutp, dtype; };

__global__ __launch_bounds__(256) void onehot_pack_kernel(const OneHotPackArgs a) {
    const long long total = 49ll * a.cin * a.coutp;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int co = (int)(e % a.coutp);
        const long long r = e / a.coutp;
        const int ci = (int)(r % a.cin), tap = (int)(r / a.cin);
        const float v = co < a.cout ? a.w[((long long)co * a.cin + ci) * 49 + tap] : 0.f;
        if (a.dtype == V2V_BF16) reinterpret_cast<unsigned short*>(a.tab)[e] = f32_to_bf16_bits(v);
        else                     reinterpret_cast<float*>(a.tab)[e] = v;
    }
}

struct OneHotPackOp : Op {
    OneHotPackArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(onehot_pack_kernel, dim3(1024), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "onehot_pack_weights"; }
};

static int coutp_of(int cout) { return cout <= 64 ? 64 : 128; }

}  // namespace v2v

using namespace v2v;

extern "C" int64_t v2v_onehot_conv_table_elems(int32_t cin, int32_t cout) {
    if (cin < 1 || cout < 1 || cout > 128) return V2V_EINVAL;
    return 49ll * cin * coutp_of(cout);
}

extern "C" int v2v_onehot_conv_pack_weights(const float* w, void* table, int32_t cin, int32_t cout, int32_t dtype, void* stream) {
    if (!w || !table || cin < 1 || cout < 1 || cout > 128 || (dtype != V2V_F32 && dtype != V2V_BF16)) {
        set_error("onehot_conv_pack_weights: bad argument (cout <= 128)"); return V2V_EINVAL;
    }
    auto op = std::make_unique<OneHotPackOp>();
    op->a = OneHotPackArgs{w, table, cin, cout, coutp_of(cout), dtype};
    return submit(std::move(op), stream);
}

extern "C" int v2v_onehot_conv_stats_rows(int32_t H, int32_t W) {
    if (H < 1 || W < 1) return V2V_EINVAL;
    return (int)(ceil_div(H, OS_TH) * ceil_div(W, OS_TW));
}

extern "C" int v2v_onehot_conv7x7(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                                  float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                                  int32_t cout, int32_t cout_stride, int32_t dtype, void* stream) {
    if (!labels || !table || !out || T < 1 || H < 4 || W < 4 || label_nc < 1 || label_nc > 254 || cout < 1 || cout > 128 ||
        cout_stride < cout || (dtype != V2V_F32 && dtype != V2V_BF16)) {
        set_error("onehot_conv7x7: bad argument (cout <= 128, label_nc <= 254, image at least 4x4 for the 3-pixel mirror)"); return V2V_EINVAL;
    }
    if (((uintptr_t)table | (uintptr_t)out) & 15) { set_error("onehot_conv7x7: table / output must be 16-byte aligned"); return V2V_EINVAL; }
    const int per = label_nc + (inst ? 1 : 0);
    const int coutp = coutp_of(cout);
    const size_t lds = (size_t)2 * T * per * (coutp * (dtype == V2V_BF16 ? 2 : 4) + 16) + (size_t)2 * T * OS_PH * OS_PW;
    if (lds > 156 * 1024) { set_error("onehot_conv7x7: T * (label_nc + 1) = %d weight rows do not fit the LDS", T * per); return V2V_EINVAL; }
    auto op = std::make_unique<OneHotConvOp>();
    op->a = OneHotConvArgs{labels, inst, table, bias, out, stats, T, H, W, label_nc, per, cout, cout_stride,
                           (int)ceil_div(W, OS_TW), in_u8};
    op->dtype = dtype; op->coutp = coutp; op->tiles = (int)(ceil_div(H, OS_TH) * ceil_div(W, OS_TW));
    return submit(std::move(op), stream);
}
