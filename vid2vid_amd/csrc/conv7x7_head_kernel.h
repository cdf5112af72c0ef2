// 7x7 stride-1 convolution with <= 32 output channels (gfx950), NHWC input; planar fp32 NCHW output with activation
// (generator heads) or raw fp32 NHWC + per-tile statistics (fine-scale stems in front of a norm layer).
//
// The generator heads (models/networks.py:180-183 model_final_img / _flow / _w, :279 and the fg tower :151) are
// Conv2d(C -> 3 | 2 | 1, k = 7) behind ReflectionPad2d(3).  As an implicit GEMM they waste >= 95 % of every MFMA
// (N = 3 padded to 32 / 64) and re-fetch the activation tile 49 times; measured 118-187 us per head at 512x256 for
// 2.5-4.9 GFLOP (profiles/r01_v9_*).  This kernel
//   * brings the (TH+6) x (TW+6) pixel patch of a TH x TW tile into LDS once per 128-byte channel chunk (LDS-DMA, same
//     row swizzle as conv3x3_patch_kernel.h) and walks the 49 taps over it: activations are read from HBM/L2 once;
//   * uses the 16-wide MFMA (v_mfma_f32_16x16x32_bf16 / exact-fp32 v_mfma_f32_16x16x4_f32): 16 pixels x 16 output
//     channels per instruction, so the padding waste is 16/cout instead of 64/cout;
//   * takes the B fragments (16 channel rows x 64 B, wave-identical) straight from the tap-major packed weights of
//     v2v_conv_pack_weights (korder 0; rows >= cout are zero) with one 16-byte load per lane per (tap, K half), one
//     kernel row (14 fragments) in flight, each reused by the wave's 4 pixel groups.
// (A first version that gave every lane one pixel and multiplied on the VALU -- v_dot2c_f32_bf16, then v_pk_fma_f32
//  with weights through the scalar cache -- measured 172 / >187 us: 512 B of weights per tap do not fit the SGPR file
//  and the scalar loads serialise.)
#pragma once
#include "conv_igemm_kernel.h"

namespace v2v {

// 16x16 MFMA on 16 bytes per lane of A (16 pixels x 64 B) and B (16 output channels x 64 B):
//   bf16: v_mfma_f32_16x16x32_bf16, lane l holds A[l&15][8*(l>>4)..+8];
//   fp32: four v_mfma_f32_16x16x4_f32 (exact fp32), lane l holds 4 consecutive k; MFMA j multiplies k in {j,4+j,8+j,12+j}.
// C/D: col = lane&15 (output channel), row = (lane>>4)*4 + reg (pixel).
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    typedef bf16x8 Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    typedef f32x4 Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    }
};

// Shared tail of the 7x7 kernels of this file: planar fp32 (+ activation, per-channel for merged heads) or raw fp32 NHWC + per-tile
// statistics.  Accumulator layout of the 16-wide MFMA: acc[g][n][r] = pixel (g >> 1 -> tile row 2 wid + that, (g & 1) * 16 + kg * 4 + r)
// x output channel n * 16 + lp.
template <typename T, int NT>
__device__ __forceinline__ void head_epilogue(const ConvKArgs& p, f32x4 (&acc)[4][NT], char* smem, const int tid, const int wid, const int lp,
                                              const int kg, const int mt, const int n_img, const int oh0, const int ow0) {
    const int H = p.OH, W = p.OW;                                     // the OUTPUT grid (= the input's but for conv7x7_c8_kernel with zero padding > 3)
    // activation-typed NHWC without activation (tile 61 as a backward-data operator, round 6): fp32 engines write exactly the raw layout
    const bool act_nhwc = p.out_mode == V2V_OUT_ACT_NHWC;
    const bool raw_mode = p.out_mode == V2V_OUT_RAW_F32_NHWC || (act_nhwc && sizeof(T) == 4);
    const bool act_bf16 = act_nhwc && sizeof(T) == 2;
    const long long hw = (long long)H * W;
    if (act_bf16) {
        // bf16 NHWC rows through the same wave-private [32][36] fp32 block: a lane packs 8 channels into one 16-byte store
        // (host: cout and cout_stride are whole 8-channel vectors, no bias / activation / statistics)
        const int lane = tid & 63;
        __syncthreads();
        float* const tw = reinterpret_cast<float*>(smem) + wid * (32 * 36);
        unsigned short* const outp = reinterpret_cast<unsigned short*>(p.out);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oh = oh0 + 2 * wid + h;
#pragma unroll
            for (int n2 = 0; n2 < (NT + 1) / 2; ++n2) {
                const int nn_cnt = NT - 2 * n2 >= 2 ? 2 : 1;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) {
                        if (nn < nn_cnt) {
                            const int n = 2 * n2 + nn;
#pragma unroll
                            for (int r = 0; r < 4; ++r) tw[(gi * 16 + kg * 4 + r) * 36 + nn * 16 + lp] = acc[2 * h + gi][n < NT ? n : 0][r];
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int lpp = nn_cnt * 2;                           // lanes per pixel (8 channels = 16 bytes each)
                const int ppp = 64 / lpp;
                const int pl0 = lane / lpp, c8 = (lane % lpp) * 8;
                for (int pl = pl0; pl < 32; pl += ppp) {
                    const int ow = ow0 + pl;
                    const int co = n2 * 32 + c8;
                    if (oh < H && ow < W && co < p.cout) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(tw + pl * 36 + c8);
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(tw + pl * 36 + c8 + 4);
                        uint4 pk;
                        pk.x = pack_bf16x2(a4[0], a4[1]); pk.y = pack_bf16x2(a4[2], a4[3]);
                        pk.z = pack_bf16x2(b4[0], b4[1]); pk.w = pack_bf16x2(b4[2], b4[3]);
                        *reinterpret_cast<uint4*>(outp + (((long long)n_img * H + oh) * W + ow) * p.cout_stride + co) = pk;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        return;
    }
    // Round 5: raw fp32 NHWC output leaves with 16-BYTE stores through a wave-private LDS block (one tile row of 32 pixels x up to 32
    // channels at a time: [32][36] fp32 = 4.5 KiB per wave), as in every other convolution kernel of the library.  The element-wise path
    // below writes 4 bytes per lane, 64-byte runs at a 128-byte pitch: 268 MB of the 6 -> 32 stem at 2048x1024 as 67 M four-byte stores
    // (173 us for 47 us of HBM time).  Same values, the statistics are summed from the registers in the same order: bit-identical.
    // (wave-uniform condition; whole float4 groups are valid or invalid together when cout % 4 == 0)
    const bool fast_raw = raw_mode && (p.cout & 3) == 0 && (p.cout_stride & 3) == 0 && (((unsigned long long)p.out) & 15ull) == 0 && !(p.ablate & 4);
    if (fast_raw) {
        const int lane = tid & 63;
        __syncthreads();                                              // the patch is dead: LDS becomes scratch
        float* const tw = reinterpret_cast<float*>(smem) + wid * (32 * 36);
        float* const outp = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oh = oh0 + 2 * wid + h;
#pragma unroll
            for (int n2 = 0; n2 < (NT + 1) / 2; ++n2) {
                const int nn_cnt = NT - 2 * n2 >= 2 ? 2 : 1;          // n-tiles (16 channels each) in this round
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) {
                        if (nn < nn_cnt) {
                            const int n = 2 * n2 + nn;
                            const int co = n * 16 + lp;
                            const float bv = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) tw[(gi * 16 + kg * 4 + r) * 36 + nn * 16 + lp] = acc[2 * h + gi][n < NT ? n : 0][r] + bv;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int lpp = nn_cnt * 4;                           // lanes per pixel (16 bytes each)
                const int ppp = 64 / lpp;                             // pixels per pass
                const int pl0 = lane / lpp, c4 = (lane % lpp) * 4;
                for (int pl = pl0; pl < 32; pl += ppp) {
                    const int ow = ow0 + pl;
                    const int co = n2 * 32 + c4;
                    if (oh < H && ow < W && co < p.cout) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(tw + pl * 36 + c4);
                        *reinterpret_cast<f32x4*>(outp + (((long long)n_img * H + oh) * W + ow) * p.cout_stride + co) = v4;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    float s1[NT], s2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        s1[n] = 0.f; s2[n] = 0.f;
        const int co = n * 16 + lp;
        const bool cvalid = co < p.cout;
        const float bv = (p.bias && cvalid) ? p.bias[co] : 0.f;
        // two heads merged into one launch (model_final_flow + model_final_w): the second head's channels have their own epilogue
        const bool second = p.act_split > 0 && co >= p.act_split;
        const int act = second ? p.act_b : p.act;
        const float act_param = second ? p.act_param_b : p.act_param, out_scale = second ? p.out_scale_b : p.out_scale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int oh = oh0 + 2 * wid + (g >> 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ow = ow0 + (g & 1) * 16 + kg * 4 + r;
                if (cvalid && oh < H && ow < W && !(p.ablate & 4)) {
                    float v = acc[g][n][r] + bv;
                    if (raw_mode) {
                        s1[n] += v;
                        s2[n] += v * v;
                        if (!fast_raw) reinterpret_cast<float*>(p.out)[(((long long)n_img * H + oh) * W + ow) * p.cout_stride + co] = v;
                    } else {
                        v = apply_act(v, act, act_param) * out_scale;
                        reinterpret_cast<float*>(p.out)[((long long)n_img * p.cout + co) * hw + (long long)oh * W + ow] = v;
                    }
                }
            }
        }
    }
    if (p.stats != nullptr) {
        // lanes lp, lp+16, lp+32, lp+48 hold the same channel: fold the 4 k-groups, then the 4 waves through LDS
        __syncthreads();                                              // the patch is dead: LDS becomes scratch
        float* red = reinterpret_cast<float*>(smem);                  // [4 waves][16*NT][2]
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float a1 = s1[n], a2 = s2[n];
            a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
            a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
            if (kg == 0) {
                red[(wid * 16 * NT + n * 16 + lp) * 2 + 0] = a1;
                red[(wid * 16 * NT + n * 16 + lp) * 2 + 1] = a2;
            }
        }
        __syncthreads();
        if (tid < 16 * NT && tid < p.cout) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += red[(w * 16 * NT + tid) * 2 + 0]; a2 += red[(w * 16 * NT + tid) * 2 + 1]; }
            float* dst = p.stats + ((long long)mt * p.cout + tid) * 2;
            dst[0] = a1;
            dst[1] = a2;
        }
    }
}

// HC = 1 (round 3): 64-byte patch rows -- channel strides that are whole HALF chunks (the 32- and 16-channel towers of the 1024x512 and
// 2048x1024 scales, which used to be widened to 64 channels in front of their heads: twice / four times the patch bytes, LDS reads and
// MFMAs for zeros).  One 16 x 16 x 32 MFMA step per tap, 16 patch rows per 1-KiB LDS-DMA piece, 36 KiB of LDS (four workgroups per CU).
// The 16-byte slot of row q is XOR-ed with (q >> 2) & 3: 16 consecutive rows x one slot index hit 16 distinct 16-byte bank groups.
template <typename T, int NT, int HC = 0>
__global__ __launch_bounds__(256) void conv7x7_head_kernel(const ConvKArgs p, const T* __restrict__ w_ro) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BKE = ElemTraits<T>::BKE;
    constexpr int TH = 8, TW = 32, HALO = 3, KS = 7;
    constexpr int PW = TW + 2 * HALO, PR = (TH + 2 * HALO) * PW;     // 38 x 14 = 532 patch rows
    constexpr int NW = 4;
    constexpr int ROWB = HC ? 64 : 128;                               // bytes per patch row = per channel chunk of a pixel
    constexpr int SLOTS = ROWB / 16, RPP = 1024 / ROWB, NH = ROWB / 64;
    constexpr int NG = (PR + RPP - 1) / RPP, GP = (NG + NW - 1) / NW; // 128-byte rows: 67 pieces, 17 per wave; 64-byte rows: 34 / 9
    constexpr int CKE = ROWB / (int)sizeof(T);                        // elements per row
    typedef typename Mma16<T>::Frag Frag;
#define V2V_HEAD_SWZ(q) (HC ? (((q) >> 2) & 3) : (((q) >> 1) & 7))

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int tpi = p.tiles_h * p.tiles_w;
    const int n_img = mt / tpi;
    const int trem = mt - n_img * tpi;
    const int th = trem / p.tiles_w;
    const int oh0 = th * TH, ow0 = (trem - th * p.tiles_w) * TW;
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc = cs / CKE;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;

    // patch loader geometry: piece g = k*NW + wid covers patch rows 8g..8g+7 (see conv3x3_patch_kernel.h)
    unsigned pp[GP];
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * RPP + lane / SLOTS;
        const int ls = (lane % SLOTS) ^ V2V_HEAD_SWZ(q);
        const int pr = q / PW, pc = q - pr * PW;
        int ih = oh0 + pr - HALO, iw = ow0 + pc - HALO;
        bool ok = q < PR;
        int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
        int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
        ih = reflect ? rh : ih;
        iw = reflect ? rw : iw;
        ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        pp[k] = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
        pok |= (ok ? 1u : 0u) << k;
    }

    // wave `wid` owns tile rows 2*wid, 2*wid+1 = 4 groups of 16 consecutive pixels; lane l: pixel l&15, k-group l>>4
    const int lp = lane & 15, kg = lane >> 4;
    int qg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) qg[g] = (2 * wid + (g >> 1)) * PW + (g & 1) * 16 + lp;
    // weights: rows (output channels) lp, lp+16 of the tap-major packed matrix; rows >= cout are zero
    const T* const wlane = w_ro + p.woff[0] + (long long)lp * p.wrow[0] + kg * VEC;
    const long long wnt = 16ll * p.wrow[0];
    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[g][n][0] = 0.f; acc[g][n][1] = 0.f; acc[g][n][2] = 0.f; acc[g][n][3] = 0.f; }

    for (int cc = 0; cc < ncc; ++cc) {
        if (cc > 0) __syncthreads();                                  // every wave is done with the previous chunk's patch
#pragma unroll
        for (int k = 0; k < GP; ++k) {
            const char* src = ((pok >> k) & 1u) ? p.in + pp[k] + cc * ROWB : p.zero_page;
            glds16(src, smem + (k * NW + wid) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const T* const wcc = wlane + cc * CKE;
        for (int dy = 0; dy < KS; ++dy) {
            Frag bf[KS][NH][NT];                                      // one kernel row of weight fragments in flight
#pragma unroll
            for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bf[dx][h][n] = *reinterpret_cast<const Frag*>(wcc + n * wnt + (long long)(dy * KS + dx) * cs + h * 4 * VEC);
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q = qg[g] + dy * PW + dx;
                    const char* const arow = smem + q * ROWB;
                    const int ax = V2V_HEAD_SWZ(q);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const Frag a = *reinterpret_cast<const Frag*>(arow + (((h * 4 + kg) ^ ax) << 4));
#pragma unroll
                        for (int n = 0; n < NT; ++n) Mma16<T>::run(a, bf[dx][h][n], acc[g][n]);
                    }
                }
            }
        }
    }

    head_epilogue<T, NT>(p, acc, smem, tid, wid, lp, kg, mt, n_img, oh0, ow0);
}

#undef V2V_HEAD_SWZ

// ---- 7x7 stems over 16-byte pixels (tile id 61; round 3) ----
// The previous-frame stems (Conv2d(6, ngf_s, 7) behind ReflectionPad2d(3): models/networks.py:134-135,262) read 6 channels = ONE
// 16-byte vector per pixel (bf16, stride 8).  As an implicit GEMM over 128-byte K chunks every tap fetches a chunk that is 7/8
// padding: 446 us for 6 -> 32 at 2048x1024 against 47 us of HBM time for its 33 MB in / 268 MB out (profiles/r03_c4_per_layer_
// roofline_hires.txt).  Here K = tap * 8 + c is walked in steps of FOUR TAPS: one 16 x 16 x 32 MFMA step takes, per lane, the
// 16 bytes of pixel (p + tap offset) from a 8.5-KB LDS halo patch (A) and the 16 bytes W[cout][tap][8] from the tap-major packed
// weights (B; 13 steps cover the 49 taps, the padding taps hit zero weights).  The pixel tile, the wave / accumulator layout and
// the epilogue (raw fp32 NHWC + per-tile statistics, or planar fp32 + activation) are those of conv7x7_head_kernel.
template <typename T, int NT, bool WLDS = false>
__global__ __launch_bounds__(256) void conv7x7_c8_kernel(const ConvKArgs p, const T* __restrict__ w_ro) {
    constexpr int VEC = ElemTraits<T>::VEC;                           // 8 bf16 / 4 fp32 = the whole channel stride
    constexpr int TH = 8, TW = 32, HALO = 3, KS = 7;
    constexpr int PW = TW + 2 * HALO, PR = (TH + 2 * HALO) * PW;     // 38 x 14 = 532 patch pixels, 16 bytes each
    constexpr int NW = 4, NG = (PR + 63) / 64, GP = (NG + NW - 1) / NW;  // 9 LDS-DMA pieces of 64 pixels, 3 per wave
    constexpr int STEPS = (KS * KS + 3) / 4;                          // 13 steps of 4 taps
    typedef typename Mma16<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int tpi = p.tiles_h * p.tiles_w;
    const int n_img = mt / tpi;
    const int trem = mt - n_img * tpi;
    const int th = trem / p.tiles_w;
    const int oh0 = th * TH, ow0 = (trem - th * p.tiles_w) * TW;
    const int H = p.H, W = p.W;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;

    // the patch: lane l of piece g brings pixel q = 64 g + l
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int g = k * NW + wid;
        if (g < NG) {
            const int q = g * 64 + lane;
            const int pr = q / PW, pc = q - pr * PW;
            int ih = oh0 + pr + p.dh0[0], iw = ow0 + pc + p.dw0[0];  // first tap's offset = -pad (3; zero padding: up to 6, the "full" convolution)
            int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
            int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
            ih = reflect ? rh : ih;
            iw = reflect ? rw : iw;
            const bool ok = q < PR && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            const char* src = ok ? p.in + ((long long)(n_img * H + ih) * W + iw) * 16 : p.zero_page;
            glds16(src, smem + g * 1024);
        }
    }
    const int lp = lane & 15, kg = lane >> 4;
    // lane's tap of step s: t = 4 s + kg (taps >= 49 read a valid patch pixel against zero weights)
    int qg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) qg[g] = (2 * wid + (g >> 1)) * PW + (g & 1) * 16 + lp;
    const T* const wlane = w_ro + p.woff[0] + (long long)lp * p.wrow[0] + kg * VEC;
    const long long wnt = 16ll * p.wrow[0];
    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[g][n][0] = 0.f; acc[g][n][1] = 0.f; acc[g][n][2] = 0.f; acc[g][n][3] = 0.f; }
    if constexpr (WLDS) {
        // Round 5: ALL weight fragments of the layer (13 steps x NT, 1 KiB each: a fragment IS 64 lanes x 16 bytes) go to LDS once, by
        // LDS-DMA next to the patch, lane-contiguous -- each step then takes its B fragments with one conflict-free ds_read_b128 per N
        // tile instead of a dependent L2 round trip per step (13 of them in series were most of a tile's life).  Same fragments, same
        // MFMA order: bit-identical.
        char* const wl = smem + NG * 1024;
        for (int f = wid; f < STEPS * NT; f += NW) {                  // fragment f = step * NT + n
            const int sstep = f / NT, n = f - sstep * NT;
            glds16(reinterpret_cast<const char*>(wlane + n * wnt + sstep * 4 * VEC), wl + f * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            Frag bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const Frag*>(wl + (s * NT + n) * 1024 + lane * 16);
            int t = 4 * s + kg;
            t = t < KS * KS ? t : KS * KS - 1;
            const int toff = (t / KS) * PW + (t % KS);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const Frag a = *reinterpret_cast<const Frag*>(smem + (qg[g] + toff) * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) Mma16<T>::run(a, bf[n], acc[g][n]);
            }
        }
    } else {
    Frag bf[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const Frag*>(wlane + n * wnt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        Frag nb[NT];
        if (s + 1 < STEPS) {                                          // next step's weight fragments under this step's MFMAs
#pragma unroll
            for (int n = 0; n < NT; ++n) nb[n] = *reinterpret_cast<const Frag*>(wlane + n * wnt + (s + 1) * 4 * VEC);
        }
        int t = 4 * s + kg;
        t = t < KS * KS ? t : KS * KS - 1;
        const int toff = (t / KS) * PW + (t % KS);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const Frag a = *reinterpret_cast<const Frag*>(smem + (qg[g] + toff) * 16);
#pragma unroll
            for (int n = 0; n < NT; ++n) Mma16<T>::run(a, bf[n], acc[g][n]);
        }
        if (s + 1 < STEPS) {
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = nb[n];
        }
    }
    }
    head_epilogue<T, NT>(p, acc, smem, tid, wid, lp, kg, mt, n_img, oh0, ow0);
}

template <typename T>
static inline int launch_c8_typed(const ConvKArgs& k, hipStream_t s) {
    constexpr int PR = (8 + 6) * (32 + 6);
    const size_t lds = 4 * 32 * 36 * sizeof(float);                   // 18 KiB: the epilogue's four [32][36] fp32 transposition blocks (the patch needs 9 KiB, the statistics scratch <= 4 KiB)
    static_assert(((PR + 63) / 64) * 1024 <= 4 * 32 * 36 * sizeof(float), "patch fits");
    const dim3 g((unsigned)k.m_tiles), b(256);
    const T* w = reinterpret_cast<const T*>(k.w);
    if constexpr (sizeof(T) == 2) {
        // bf16: the weight fragments in LDS (9 KiB patch + 13 NT KiB; cout <= 64: at most 61 KiB, two workgroups per CU).  V2V_C8_WLDS=0: the
        // fragments from L2 step by step (A/B)
        static const bool wlds = [] { const char* e = getenv("V2V_C8_WLDS"); return !(e && e[0] == '0'); }();
        if (wlds && k.cout <= 64) {
            const int nt = k.cout <= 16 ? 1 : k.cout <= 32 ? 2 : 4;
            const size_t need = (size_t)((PR + 63) / 64) * 1024 + (size_t)13 * nt * 1024;
            const size_t l2 = need > lds ? need : lds;
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv7x7_c8_kernel<T, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                attr_done = true;
            }
            if (nt == 1)      hipLaunchKernelGGL((conv7x7_c8_kernel<T, 1, true>), g, b, l2, s, k, w);
            else if (nt == 2) hipLaunchKernelGGL((conv7x7_c8_kernel<T, 2, true>), g, b, l2, s, k, w);
            else              hipLaunchKernelGGL((conv7x7_c8_kernel<T, 4, true>), g, b, l2, s, k, w);
            return check_launch();
        }
    }
    if (k.cout <= 16)      hipLaunchKernelGGL((conv7x7_c8_kernel<T, 1>), g, b, lds, s, k, w);
    else if (k.cout <= 32) hipLaunchKernelGGL((conv7x7_c8_kernel<T, 2>), g, b, lds, s, k, w);
    else if (k.cout <= 64) hipLaunchKernelGGL((conv7x7_c8_kernel<T, 4>), g, b, lds, s, k, w);
    else                   hipLaunchKernelGGL((conv7x7_c8_kernel<T, 8>), g, b, lds, s, k, w);
    return check_launch();
}

// ---- generator heads as a ROW GEMM + shifted sum (tile id 62, bf16; round 4) ----
// conv7x7_head_kernel spends one 16 x 16 x 32 MFMA per (tap, 16 pixels) with 3 of its 16 output columns alive: the 32 -> 3 heads of the
// 2048x1024 scale took 250 us each for 134 MB of input (profiles/r03_e4_per_layer_roofline_hires.txt: 10 % of their HBM bound), the
// 128 -> 3 heads at 512x256 63 us.  Here the kernel COLUMN joins the output channel in the GEMM's N index:
//     V[oh][iw][(dx, co)] = sum_{dy, c} x[oh + dy - 3][iw][c] * w[co][dy][dx][c]        (M = patch pixels, K = 7 rows x C, N = 7 x 4 -> 32)
//     out[oh][ow][co]     = sum_dx V[oh][ow + dx][(dx, co)]                              (iw in patch columns: ow + dx)
// i.e. 7 C instead of 49 C of K and all 32 columns of the 32-wide MFMA in use: 3.5x fewer matrix cycles (2.9x with the 38 / 32 halo
// columns).  With the patch rows flattened, m = oh * PW + iw, the A rows of V-tile row m at kernel row dy are patch rows m + dy * PW:
// contiguous, so an M tile is any 32 consecutive m (it may straddle image rows).  A 10 x 32 pixel tile is 380 V rows = 12 M tiles, three
// per wave.  The B rows (dy, n = dx * 4 + co) of a 32-channel chunk are brought from the tap-major packed matrix (korder 0: row co,
// K = (dy * 7 + dx) * cs + c; rows >= cout are zero; n >= 28 reads the zero page) into LDS next to the patch by the same LDS-DMA
// pieces, so no weight load sits in the MFMA loop.  V goes through LDS (fp32, row stride 33 words: the 7 shifted reads of 32
// consecutive ow hit 32 different banks) and leaves as coalesced planar fp32 rows with bias + activation (per channel for the merged
// flow + weight head).  53 KB of LDS: three workgroups per CU overlap one another's patch loads.
template <typename T>          // bf16 storage only (the fp32 / x3 paths keep conv7x7_head_kernel's exact-fp32 MFMA)
__global__ __launch_bounds__(256) void conv7x7_rowsum_kernel(const ConvKArgs p, const T* __restrict__ w_ro) {
    static_assert(sizeof(T) == 2, "bf16");
    constexpr int TH = 10, TW = 32, HALO = 3, KS = 7;
    constexpr int PW = TW + 2 * HALO, PR = (TH + 2 * HALO) * PW;     // 38 x 16 = 608 patch rows of 64 bytes (32 channels)
    constexpr int NW = 4, ROWB = 64, SLOTS = 4, RPP = 16;
    constexpr int NGP = PR / RPP, GPP = (NGP + NW - 1) / NW;          // 38 patch pieces, <= 10 per wave
    constexpr int NB = KS * 32, NGB = NB / RPP, GPB = (NGB + NW - 1) / NW;   // 224 weight rows = 14 pieces, <= 4 per wave
    constexpr int B_OFF = NGP * 1024;                                 // 38912
    constexpr int MT = 3;                                             // M tiles per wave (12 x 32 = 384 >= 380 V rows)
    constexpr int VS = 33;                                            // V row stride in words
    static_assert(PR % RPP == 0 && NB % RPP == 0 && 384 * VS * 4 <= B_OFF + NGB * 1024, "layout");
#define V2V_RS_SWZ(q) (((q) >> 2) & 3)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x;
    const int tpi = p.tiles_h * p.tiles_w;
    const int n_img = mt / tpi;
    const int trem = mt - n_img * tpi;
    const int th = trem / p.tiles_w;
    const int oh0 = th * TH, ow0 = (trem - th * p.tiles_w) * TW;
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc = cs / 32;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;

    // ---- loader geometry: patch piece g = k * NW + wid covers patch rows 16 g .. 16 g + 15, lane l row l / 4, 16-byte slot l % 4
    unsigned pp[GPP];
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GPP; ++k) {
        const int q = (k * NW + wid) * RPP + lane / SLOTS;
        const int ls = (lane % SLOTS) ^ V2V_RS_SWZ(q);
        const int pr = q / PW, pc = q - pr * PW;
        int ih = oh0 + pr - HALO, iw = ow0 + pc - HALO;
        bool ok = q < PR;
        int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
        int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
        ih = reflect ? rh : ih;
        iw = reflect ? rw : iw;
        ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        pp[k] = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * 8) * 2ll);
        pok |= (ok ? 1u : 0u) << k;
    }
    // weight piece g covers B rows 16 g .. 16 g + 15: row rb = dy * 32 + n, n = dx * 4 + co
    long long wp[GPB];
    unsigned wok = 0;
#pragma unroll
    for (int k = 0; k < GPB; ++k) {
        const int rb = (k * NW + wid) * RPP + lane / SLOTS;
        const int ls = (lane % SLOTS) ^ V2V_RS_SWZ(rb);
        const int dy = rb >> 5, n = rb & 31, dx = n >> 2, co = n & 3;
        const bool ok = rb < NB && n < 4 * KS;
        wp[k] = ok ? (p.woff[0] + (long long)co * p.wrow[0] + (long long)(dy * KS + dx) * cs + ls * 8) * 2ll : 0ll;
        wok |= (ok ? 1u : 0u) << k;
    }

    // ---- fragments: lane l = (row l & 31, k half l >> 5) of the 32 x 32 x 16 MFMA, both operands
    const int lr = lane & 31, hi = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int cc = 0; cc < ncc; ++cc) {
        if (cc > 0) __syncthreads();                                  // every wave is done with the previous chunk
#pragma unroll
        for (int k = 0; k < GPP; ++k) {
            if (k * NW + wid < NGP) {
                const char* src = ((pok >> k) & 1u) ? p.in + pp[k] + cc * ROWB : p.zero_page;
                glds16(src, smem + (k * NW + wid) * 1024);
            }
        }
#pragma unroll
        for (int k = 0; k < GPB; ++k) {
            if (k * NW + wid < NGB) {
                const char* src = ((wok >> k) & 1u) ? reinterpret_cast<const char*>(w_ro) + wp[k] + cc * ROWB : p.zero_page;
                glds16(src, smem + B_OFF + (k * NW + wid) * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rb = dy * 32 + lr;
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + B_OFF + rb * ROWB + (((2 * s + hi) ^ V2V_RS_SWZ(rb)) << 4));
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int q = (wid * MT + t) * 32 + lr + dy * PW;     // rows >= PR (V rows >= 380 only) read the weight area: never used
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + q * ROWB + (((2 * s + hi) ^ V2V_RS_SWZ(q)) << 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- V -> LDS (C layout: column n = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), then the shifted sum
    __syncthreads();                                                  // patch and weights are dead
    float* const Vs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (wid * MT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            Vs[m * VS + lr] = acc[t][r];
        }
    __syncthreads();
    const long long hw = (long long)H * W;
    const int nout = p.cout * TH * TW;
    for (int idx = tid; idx < nout; idx += 256) {
        const int co = idx / (TH * TW);
        const int px = idx - co * (TH * TW);
        const int ohl = px >> 5, owl = px & 31;
        const int oh = oh0 + ohl, ow = ow0 + owl;
        const float* v0 = Vs + (ohl * PW + owl) * VS + co;
        float v = 0.f;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) v += v0[dx * (VS + 4)];
        if (oh < H && ow < W) {
            const bool second = p.act_split > 0 && co >= p.act_split;
            const int act = second ? p.act_b : p.act;
            const float act_param = second ? p.act_param_b : p.act_param, out_scale = second ? p.out_scale_b : p.out_scale;
            v += p.bias ? p.bias[co] : 0.f;
            reinterpret_cast<float*>(p.out)[((long long)n_img * p.cout + co) * hw + (long long)oh * W + ow] = apply_act(v, act, act_param) * out_scale;
        }
    }
#undef V2V_RS_SWZ
}

static inline int launch_rowsum_bf16_impl(const ConvKArgs& k, hipStream_t s) {
    const size_t lds = (size_t)(38 + 14) * 1024;                      // 52 KiB: three workgroups per CU
    hipLaunchKernelGGL(conv7x7_rowsum_kernel<bf16_t>, dim3((unsigned)k.m_tiles), dim3(256), lds, s, k, reinterpret_cast<const bf16_t*>(k.w));
    return check_launch();
}

template <typename T>
static inline int launch_head_typed(const ConvKArgs& k, hipStream_t s) {
    constexpr int PR = (8 + 6) * (32 + 6);
    constexpr int GP = ((PR + 7) / 8 + 3) / 4;
    const size_t lds = (size_t)GP * 4 * 1024;                         // 68 KiB: two workgroups per CU
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv7x7_head_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv7x7_head_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    // V2V_HEAD_HC=1: 64-byte patch rows for every channel stride (a 128-byte chunk = two half chunks: twice the patch loads and
    // barriers, half the LDS -> four workgroups per CU instead of two); default: only where the stride requires it
    static const int force_hc = [] { const char* e = getenv("V2V_HEAD_HC"); return (e && e[0] == '1') ? 1 : 0; }();
    if (force_hc || (k.cin_stride * (int)sizeof(T)) % 128 != 0) {     // whole half chunks only: 64-byte patch rows
        constexpr int GPH = ((PR + 15) / 16 + 3) / 4;
        const size_t ldsh = (size_t)GPH * 4 * 1024;                   // 36 KiB
        if (k.cout <= 16)
            hipLaunchKernelGGL((conv7x7_head_kernel<T, 1, 1>), dim3((unsigned)k.m_tiles), dim3(256), ldsh, s, k, reinterpret_cast<const T*>(k.w));
        else
            hipLaunchKernelGGL((conv7x7_head_kernel<T, 2, 1>), dim3((unsigned)k.m_tiles), dim3(256), ldsh, s, k, reinterpret_cast<const T*>(k.w));
        return check_launch();
    }
    if (k.cout <= 16)
        hipLaunchKernelGGL((conv7x7_head_kernel<T, 1>), dim3((unsigned)k.m_tiles), dim3(256), lds, s, k, reinterpret_cast<const T*>(k.w));
    else
        hipLaunchKernelGGL((conv7x7_head_kernel<T, 2>), dim3((unsigned)k.m_tiles), dim3(256), lds, s, k, reinterpret_cast<const T*>(k.w));
    return check_launch();
}

}  // namespace v2v
