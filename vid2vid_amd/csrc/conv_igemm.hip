// Implicit-GEMM convolution / transposed convolution for gfx950 (MI355X), NHWC activations.
//
//   D[pixel][cout] = sum_{tap, cin} A[pixel @ tap][cin] * Wp[cout][tap][cin]
//
// One kernel template serves every conv on the vid2vid hot path (reference call sites:
// models/networks.py:132-183 CompositeGenerator, :247-279 CompositeLocalGenerator, :335-352
// GlobalGenerator, :571-587 ResnetBlock, :687-706 NLayerDiscriminator and
// models/flownet2_pytorch/networks/submodules.py:7-38):
//   * Conv2d k x k, stride 1/2, zero or reflection padding folded into the tile loader
//     (no padded copy is ever materialised),
//   * ConvTranspose2d(stride 2) as 4 output-parity classes (blockIdx.y) with 1/2/2/4 (3x3)
//     or 4x(2x2) (4x4) taps each -- no zero insertion,
//   * exact-fp32 path on v_mfma_f32_32x32x2_f32 (parity gate) and bf16 path on
//     v_mfma_f32_32x32x16_bf16 (throughput), same LDS image / fragment addressing in bytes,
//   * epilogue: bias, per-channel (sum, sum^2) partials for training-mode BatchNorm
//     (deterministic: one row of partials per M tile), or activation + store.
//
// Structure (1 workgroup = 4 waves owns a BM x BN tile, K advances in 128-byte chunks):
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction): no VGPR round trip and no ds_write pass, which alone would eat >100% of
//     the LDS cycles the MFMAs leave (ds_write_b128 = 13 cycles / wave-instruction);
//   * NS-stage LDS ring with NS-1 tiles in flight, retired with a COUNTED s_waitcnt vmcnt(N)
//     and ONE raw s_barrier per K chunk -- at batch 1 the dominant layer (1024->1024 3x3 at
//     32x64 pixels) yields exactly one workgroup per CU, so there is no second workgroup to
//     hide the ~1-2 us weight-stream latency and the prefetch distance has to do it;
//   * LDS rows are 128 B; the 16-byte slot index is XOR-swizzled with (row>>1)&7.  LDS-DMA
//     writes lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
//     fragment read (cdna guide rule 21); ds_read_b128 of the 32x32 MFMA operand layout is
//     bank-conflict free under it;
//   * padding / K-tail / ragged-M lanes fetch from a 16-byte zero page instead of branching;
//   * layers whose channel stride is a multiple of the K chunk (all wide layers) use a
//     uniform tap walk: row pointers are recomputed only when the tap changes;
//   * blockIdx.x is remapped so that the 32 workgroups resident on one XCD share weight
//     (N) tiles in that XCD's private L2.
#include "v2v_internal.h"
#include <cstdarg>
#include <cstring>

namespace v2v {

struct ConvKArgs {
    const char* in;
    const char* w;
    const char* zero_page;
    const float* bias;
    char* out;
    float* stats;
    int N, H, W, cin_stride;
    int cout, cout_stride, cout_p;
    int OH, OW;
    int sm;              // input step per class-grid step (conv: stride, convT: 1)
    int os;              // output step per class-grid step (conv: 1, convT: 2)
    int pad_mode;
    int Mc[4], OHc[4], OWc[4];   // rows per class and class grid (transposed: output pixels of parity class)
    int m_tiles, n_tiles;
    int out_mode, act;
    float act_param, out_scale;
    // per class (conv: class 0 only)
    int nkh[4], nkw[4];
    int dh0[4], dw0[4];  // first tap's input offset
    int dstep;           // +1 conv, -1 convT
    int ktot[4], kpad[4];
    long long woff[4];   // element offset of the class matrix inside w
    // in-kernel norm finalize (last-arriving workgroup of an N tile), optional
    int* fin_counter; const float* fin_gamma; const float* fin_beta; float* fin_out;
    float* fin_rmean; float* fin_rvar;
    float fin_eps, fin_momentum; double fin_inv_count, fin_unbias;
};

__device__ __forceinline__ int xcd_remap(int bid, int ntot) {
    const int q = ntot >> 3, r = ntot & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 16 bytes per lane, global -> LDS, asynchronous (counted by vmcnt).  `lds` must be
// wave-uniform: the hardware writes lane l at lds + 16*l.
__device__ __forceinline__ void glds16(const char* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    // lane (i = l&31, half h = l>>5) holds k = 4h..4h+3 of an 8-deep step: MFMA j multiplies
    // k in {j, 4+j}; A and B use the same convention so every k is covered exactly once.
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};

template <typename T, int BM, int BN, int WGM, int WGN, int NS>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_igemm_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BKE = ElemTraits<T>::BKE;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NW = WGM * WGN;             // waves per workgroup (4 or 8)
    constexpr int LR = NW * 8;                // LDS rows written per loader round (one 1 KiB DMA per wave)
    constexpr int RA = BM / LR, RB = BN / LR;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int D = NS - 1;                 // tiles in flight
    constexpr int LPT = RA + RB;              // LDS-DMA instructions per tile per wave
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(BM % LR == 0 && BN % LR == 0 && RA >= 1 && RB >= 1, "loader rounds");
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    static_assert(NS >= 2 && LPT * (D - 1) <= 63, "vmcnt range");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int cls = blockIdx.y;

    const int lin = xcd_remap(blockIdx.x, p.m_tiles * p.n_tiles);
    const int nt = lin / p.m_tiles;
    const int mt = lin - nt * p.m_tiles;

    const int nkh = p.nkh[cls], nkw = p.nkw[cls];
    const int dh0 = p.dh0[cls], dw0 = p.dw0[cls], dstep = p.dstep;
    const int kpad = p.kpad[cls];
    const int nk = kpad / BKE;
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const char* const zp = p.zero_page;

    // ---------------- loader geometry ----------------
    // wave `wid`, lane l writes LDS row (wid*8 + l>>3) + 32*i, physical 16-byte slot l&7, which
    // must hold the LOGICAL slot (l&7) ^ swz(row): that is the slot this lane fetches.
    const int lrow = wid * 8 + (lane >> 3);          // 0..LR-1
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
    const int koff = lslot * VEC;                    // element offset inside the 128-byte chunk
    char* const lds_wave = smem + wid * 8 * 128;     // wave-uniform part of the destination

    int pixbase[RA], ohs[RA], ows[RA];
    unsigned rowvalid = 0;
    {
        const int owc = p.OWc[cls];
        const int hw = p.OHc[cls] * owc;
        const int mcls = p.Mc[cls];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = mt * BM + lrow + LR * i;
            const bool ok = m < mcls;
            const int mm = ok ? m : 0;
            const int n = mm / hw;
            const int rem = mm - n * hw;
            const int oi = rem / owc;
            const int oj = rem - oi * owc;
            pixbase[i] = n * H * W;
            ohs[i] = oi * p.sm;
            ows[i] = oj * p.sm;
            rowvalid |= (ok ? 1u : 0u) << i;
        }
    }
    const char* wp[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const long long r = (long long)nt * BN + lrow + LR * i;
        wp[i] = p.w + ((long long)p.woff[cls] + r * kpad + koff) * (long long)sizeof(T);
    }

    // resolves one tap for one row: source pointer of this lane's slot (channel 0 of the chunk)
    auto tap_ptr = [&](int i, int dh, int dw, bool& ok) -> const char* {
        int ih = ohs[i] + dh, iw = ows[i] + dw;
        // branch-free: reflection is a select on a uniform flag, validity a bitwise AND
        int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
        int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
        ih = reflect ? rh : ih;
        iw = reflect ? rw : iw;
        ok = (bool)((int)ok & (int)((unsigned)ih < (unsigned)H) & (int)((unsigned)iw < (unsigned)W));
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        return p.in + ((long long)(pixbase[i] + ih * W + iw) * cs) * (long long)sizeof(T);
    };

    // ---- fast path (cs % BKE == 0): the tap is uniform over the workgroup ----
    const bool fastk = (cs % BKE) == 0;
    const char* ap[RA];
    unsigned aok = 0;
    int tap_h = 0, tap_w = 0, kcb = 0;               // uniform
    const int row_bytes = cs * (int)sizeof(T);
    auto set_tap = [&]() {
        const int dh = dh0 + tap_h * dstep, dw = dw0 + tap_w * dstep;
        const bool tvalid = tap_h < nkh;
        aok = 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool ok = tvalid && ((rowvalid >> i) & 1u);
            ap[i] = tap_ptr(i, dh, dw, ok) + koff * (int)sizeof(T);
            aok |= (ok ? 1u : 0u) << i;
        }
    };
    // ---- general path: per-lane (tap, channel) walk ----
    int kc, kth, ktw;
    {
        const int t = koff / cs;
        kc = koff - t * cs;
        kth = t / nkw;
        ktw = t - kth * nkw;
    }
    const int nwrap = (BKE + cs - 1) / cs;           // tap wraps per chunk (1 when cs >= BKE)
    if (fastk) set_tap();

    int issued = 0;                                  // tiles issued so far (uniform)
    auto issue = [&]() {
        char* sbase = lds_wave + (issued % NS) * STAGE;
        if (fastk) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const char* src = ((aok >> i) & 1u) ? ap[i] + kcb : zp;
                glds16(src, sbase + i * LR * 128);
            }
        } else {
            const bool kvalid = kth < nkh;
            const int dh = dh0 + kth * dstep, dw = dw0 + ktw * dstep;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                bool ok = kvalid && ((rowvalid >> i) & 1u);
                const char* src = tap_ptr(i, dh, dw, ok) + kc * (int)sizeof(T);
                src = ok ? src : zp;
                glds16(src, sbase + i * LR * 128);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            glds16(wp[i] + (long long)issued * 128, sbase + BM * 128 + i * LR * 128);
        // advance the K walk by one chunk
        if (fastk) {
            kcb += 128;
            if (kcb == row_bytes) {
                kcb = 0;
                if (++tap_w == nkw) { tap_w = 0; ++tap_h; }
                set_tap();
            }
        } else {
            kc += BKE;
            for (int it = 0; it < nwrap; ++it) {
                const bool wr = kc >= cs;
                kc -= wr ? cs : 0;
                ktw += wr ? 1 : 0;
                const bool w2 = ktw == nkw;
                ktw = w2 ? 0 : ktw;
                kth += w2 ? 1 : 0;
            }
        }
        ++issued;
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    const int fx = (lr >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = ((s * 2 + hi) ^ fx) << 4;
    const int a_row_off = (wm * WM + lr) * 128;
    const int b_row_off = BM * 128 + (wn * WN + lr) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---------------- main loop ----------------
    for (int t = 0; t < D && t < nk; ++t) issue();
    for (int ks = 0; ks < nk; ++ks) {
        // tile ks must have landed; in steady state tiles ks+1 .. ks+D-1 stay in flight
        if (ks + D <= nk) wait_vmcnt<LPT * (D - 1)>();
        else              wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();     // every wave's part of tile ks is in LDS, and every wave
                                          // is done reading stage (ks-1)%NS, which is refilled now
        if (ks + D < nk) issue();
        const char* sb = smem + (ks % NS) * STAGE;
        Frag fa[4][TM], fb[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[s][i] = *reinterpret_cast<const Frag*>(sb + a_row_off + i * 32 * 128 + foff[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[s][j] = *reinterpret_cast<const Frag*>(sb + b_row_off + j * 32 * 128 + foff[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
    }
    __syncthreads();                      // LDS ring is free: reused for the statistics reduction

    // ---------------- epilogue ----------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    float* red = reinterpret_cast<float*>(smem);   // [WGM][BN][2]
    const bool want_stats = p.stats != nullptr;
    const int a_par = cls >> 1, b_par = cls & 1;
    const int owc_e = p.OWc[cls];
    const int hwc = p.OHc[cls] * owc_e;
    const int mcls_e = p.Mc[cls];
    const long long ohow = (long long)p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ncol = nt * BN + wn * WN + j * 32 + lr;
        const bool nvalid = ncol < p.cout;
        const float bv = (p.bias != nullptr && nvalid) ? p.bias[ncol] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int m = mt * BM + row;
                if (m < mcls_e && nvalid) {
                    float v = acc[i][j][r] + bv;
                    long long opix;   // output pixel index in [N][OH][OW]
                    if (p.os == 1) {
                        opix = m;
                    } else {
                        const int n = m / hwc;
                        const int rem = m - n * hwc;
                        const int oi = rem / owc_e;
                        const int oj = rem - oi * owc_e;
                        opix = ((long long)n * p.OH + (oi * 2 + a_par)) * p.OW + (oj * 2 + b_par);
                    }
                    if (p.out_mode == V2V_OUT_RAW_F32_NHWC) {
                        s1 += v;
                        s2 += v * v;
                        reinterpret_cast<float*>(p.out)[opix * p.cout_stride + ncol] = v;
                    } else {
                        v = apply_act(v, p.act, p.act_param) * p.out_scale;
                        if (p.out_mode == V2V_OUT_ACT_NHWC) {
                            store_act(reinterpret_cast<T*>(p.out), opix * p.cout_stride + ncol, v);
                        } else {
                            const long long n = opix / ohow;
                            const long long pix = opix - n * ohow;
                            reinterpret_cast<float*>(p.out)[(n * p.cout + ncol) * ohow + pix] = v;
                        }
                    }
                }
            }
        }
        if (want_stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (hi == 0) {
                const int c = wn * WN + j * 32 + lr;
                red[(wm * BN + c) * 2 + 0] = s1;
                red[(wm * BN + c) * 2 + 1] = s2;
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < BN) {
            const int ncol = nt * BN + tid;
            if (ncol < p.cout) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < WGM; ++q) {
                    s1 += red[(q * BN + tid) * 2 + 0];
                    s2 += red[(q * BN + tid) * 2 + 1];
                }
                float* dst = p.stats + ((long long)(cls * p.m_tiles + mt) * p.cout + ncol) * 2;
                if (p.fin_counter != nullptr) {
                    // 8-byte agent-scope (write-through, sc1) store: the (sum, sum^2) granule is what the last
                    // workgroup reads back with agent-scope loads -- no L2 write-back fence is needed
                    const unsigned long long bits = (unsigned long long)__float_as_uint(s1) |
                                                    ((unsigned long long)__float_as_uint(s2) << 32);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
        if (p.fin_counter != nullptr) {
            // ---- training-mode norm finalize by the LAST workgroup to finish this N tile ----
            // (get_norm_layer, models/networks.py:23-30: batch statistics -> scale/shift; replaces a separate
            // bn_finalize launch per layer).  Hand-off (cdna guide G16, "sc1 payload -> vmcnt(0) -> flag" form):
            // the partial rows are 8-byte write-through agent-scope stores, every wave drains them, workgroup
            // barrier, then ONE relaxed agent-scope ticket; the last arriver reads all rows back with agent-scope
            // 8-byte loads in a fixed order (deterministic, independent of which workgroup happens to be last).
            // No release/acquire fence: a release would write back the XCD L2's dirty conv output (+30 us measured).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem + 16384);
            const int total = (int)gridDim.y * p.m_tiles;
            if (tid == 0) {
                const int tk = __hip_atomic_fetch_add(p.fin_counter + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = tk == total - 1 ? 1 : 0;
                if (last) __hip_atomic_store(p.fin_counter + nt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
                *flag = last;
            }
            __syncthreads();
            if (*flag) {
                constexpr int NT = NW * 64, PH = NT / BN;
                double* acc2 = reinterpret_cast<double*>(smem);      // [PH][BN][2], <= 8 KiB
                const int c = tid % BN, ph = tid / BN;
                const int ncol = nt * BN + c;
                double s1 = 0.0, s2 = 0.0;
                if (ncol < p.cout) {
                    for (int r = ph; r < total; r += PH) {
                        const unsigned long long bits = __hip_atomic_load(
                            reinterpret_cast<const unsigned long long*>(p.stats + ((long long)r * p.cout + ncol) * 2),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s1 += (double)__uint_as_float((unsigned)(bits & 0xffffffffull));
                        s2 += (double)__uint_as_float((unsigned)(bits >> 32));
                    }
                }
                acc2[(ph * BN + c) * 2 + 0] = s1;
                acc2[(ph * BN + c) * 2 + 1] = s2;
                __syncthreads();
                if (ph == 0 && ncol < p.cout) {
                    s1 = 0.0; s2 = 0.0;
#pragma unroll
                    for (int q = 0; q < PH; ++q) { s1 += acc2[(q * BN + c) * 2 + 0]; s2 += acc2[(q * BN + c) * 2 + 1]; }
                    const double mean = s1 * p.fin_inv_count;
                    double var = s2 * p.fin_inv_count - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const double invstd = 1.0 / sqrt(var + (double)p.fin_eps);
                    const double g = p.fin_gamma ? (double)p.fin_gamma[ncol] : 1.0;
                    const double b = p.fin_beta ? (double)p.fin_beta[ncol] : 0.0;
                    const double sc = g * invstd;
                    p.fin_out[ncol] = (float)sc;
                    p.fin_out[p.cout + ncol] = (float)(b - mean * sc);
                    p.fin_out[2 * p.cout + ncol] = (float)mean;
                    p.fin_out[3 * p.cout + ncol] = (float)invstd;
                    if (p.fin_rmean) p.fin_rmean[ncol] = (1.f - p.fin_momentum) * p.fin_rmean[ncol] + p.fin_momentum * (float)mean;
                    if (p.fin_rvar)  p.fin_rvar[ncol]  = (1.f - p.fin_momentum) * p.fin_rvar[ncol] + p.fin_momentum * (float)(var * p.fin_unbias);
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------
struct TileCfg { int id, BM, BN; };
static const TileCfg kCfgs[] = {
    {1, 128, 128}, {2, 128, 64}, {3, 64, 64}, {4, 128, 32}, {5, 64, 128}, {6, 256, 64},
    {7, 128, 64}, {8, 128, 128},      // deeper LDS-DMA rings of 2 / 1
    {9, 64, 64}, {10, 64, 64}, {11, 128, 64}, {12, 64, 128},   // occupancy / depth variants of 3, 2, 5
    {13, 128, 64}, {14, 128, 128}, {15, 128, 128}, {16, 256, 64}, {17, 64, 128},   // 8-wave workgroups
};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

static const TileCfg* find_cfg(int id) {
    for (int i = 0; i < kNumCfgs; ++i)
        if (kCfgs[i].id == id) return &kCfgs[i];
    return nullptr;
}

static int bke_of(int dtype) { return dtype == V2V_BF16 ? 64 : 32; }

// taps of transposed-conv output class `par` (output index = stride*i + par) along one axis:
// kernel taps k = k0 + stride*t, t = 0..nk-1, read input index i + d0 - t.
static void convt_axis(int K, int pad, int stride, int par, int* k0, int* nk, int* d0) {
    if (stride == 1) { *k0 = 0; *nk = K; *d0 = pad; return; }
    *k0 = (par + pad) & 1;
    *nk = (*k0 < K) ? (K - *k0 + 1) / 2 : 0;
    *d0 = (par + pad - *k0) / 2;   // (par+pad-k0) is even
}

struct ConvGeom {
    int ncls;
    int nkh[4], nkw[4], kh0[4], kw0[4], dh0[4], dw0[4], ktot[4], kpad[4];
    long long woff[4];
    long long total;
    int cout_p;
};

static int conv_geom(int cin_stride, int cout, int KH, int KW, int transposed, int stride, int pad, int dtype, ConvGeom* g) {
    const int bke = bke_of(dtype);
    memset(g, 0, sizeof(*g));
    g->cout_p = (int)round_up(cout, 128);
    long long off = 0;
    if (!transposed) {
        g->ncls = 1;
        g->nkh[0] = KH; g->nkw[0] = KW; g->kh0[0] = 0; g->kw0[0] = 0;
        g->dh0[0] = -pad; g->dw0[0] = -pad;
        g->ktot[0] = KH * KW * cin_stride;
        g->kpad[0] = (int)round_up(g->ktot[0], bke);
        g->woff[0] = 0;
        off = (long long)g->cout_p * g->kpad[0];
    } else {
        g->ncls = stride == 1 ? 1 : 4;
        for (int c = 0; c < g->ncls; ++c) {
            const int a = c >> 1, b = c & 1;
            convt_axis(KH, pad, stride, a, &g->kh0[c], &g->nkh[c], &g->dh0[c]);
            convt_axis(KW, pad, stride, b, &g->kw0[c], &g->nkw[c], &g->dw0[c]);
            g->ktot[c] = g->nkh[c] * g->nkw[c] * cin_stride;
            g->kpad[c] = (int)round_up(g->ktot[c] > 0 ? g->ktot[c] : 1, bke);
            g->woff[c] = off;
            off += (long long)g->cout_p * g->kpad[c];
        }
    }
    g->total = off;
    return 0;
}

// ---- weight packing kernel: PyTorch layout fp32 -> packed class matrices ----------------
struct PackArgs {
    const float* w; void* dst;
    int cin, cin_stride, cout, cout_p, KH, KW, transposed, kstep;
    int ncls;
    int nkh[4], nkw[4], kh0[4], kw0[4], kpad[4];
    long long woff[4];
    long long total;
    int dtype;
};

__global__ void pack_weights_kernel(const PackArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.total; e += stride) {
        int cls = 0;
#pragma unroll
        for (int c = 1; c < 4; ++c)
            if (c < a.ncls && e >= a.woff[c]) cls = c;
        const long long le = e - a.woff[cls];
        const int kp = a.kpad[cls];
        const int co = (int)(le / kp);
        const int k = (int)(le - (long long)co * kp);
        const int t = k / a.cin_stride;
        const int c = k - t * a.cin_stride;
        float v = 0.f;
        const int ntaps = a.nkh[cls] * a.nkw[cls];
        if (co < a.cout && t < ntaps && c < a.cin) {
            const int th = t / a.nkw[cls], tw = t - th * a.nkw[cls];
            int kh, kw;
            if (a.transposed) { kh = a.kh0[cls] + a.kstep * th; kw = a.kw0[cls] + a.kstep * tw; }
            else              { kh = th; kw = tw; }
            long long src;
            if (a.transposed) src = (((long long)c * a.cout + co) * a.KH + kh) * a.KW + kw;   // [cin][cout][kh][kw]
            else              src = (((long long)co * a.cin + c) * a.KH + kh) * a.KW + kw;    // [cout][cin][kh][kw]
            v = a.w[src];
        }
        if (a.dtype == V2V_BF16) reinterpret_cast<unsigned short*>(a.dst)[e] = f32_to_bf16_bits(v);
        else                     reinterpret_cast<float*>(a.dst)[e] = v;
    }
}

struct PackOp : Op {
    PackArgs a;
    int launch(hipStream_t s) override {
        const int threads = 256;
        long long blocks = ceil_div(a.total, threads);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "pack_weights"; }
};

// ---- conv launch ------------------------------------------------------------------------
template <typename T, int BM, int BN, int WGM, int WGN, int NS>
static int launch_cfg(const ConvKArgs& k, int ncls, hipStream_t s) {
    const size_t lds = (size_t)NS * (BM + BN) * 128;
    auto kern = conv_igemm_kernel<T, BM, BN, WGM, WGN, NS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles), (unsigned)ncls);
    hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, k);
    return check_launch();
}

template <typename T>
static int launch_typed(int cfg, const ConvKArgs& k, int ncls, hipStream_t s) {
    switch (cfg) {
        case 1: return launch_cfg<T, 128, 128, 2, 2, 3>(k, ncls, s);   //  96 KiB LDS
        case 2: return launch_cfg<T, 128, 64, 2, 2, 4>(k, ncls, s);    //  96 KiB
        case 3: return launch_cfg<T, 64, 64, 2, 2, 4>(k, ncls, s);     //  64 KiB (2 workgroups / CU)
        case 4: return launch_cfg<T, 128, 32, 4, 1, 4>(k, ncls, s);    //  80 KiB (2 / CU)
        case 5: return launch_cfg<T, 64, 128, 2, 2, 4>(k, ncls, s);    //  96 KiB
        case 6: return launch_cfg<T, 256, 64, 4, 1, 3>(k, ncls, s);    // 120 KiB
        case 7: return launch_cfg<T, 128, 64, 2, 2, 6>(k, ncls, s);    // 144 KiB
        case 8: return launch_cfg<T, 128, 128, 2, 2, 4>(k, ncls, s);   // 128 KiB
        case 9: return launch_cfg<T, 64, 64, 2, 2, 3>(k, ncls, s);     //  48 KiB (3 workgroups / CU)
        case 10: return launch_cfg<T, 64, 64, 2, 2, 2>(k, ncls, s);    //  32 KiB (5 / CU)
        case 11: return launch_cfg<T, 128, 64, 2, 2, 2>(k, ncls, s);   //  48 KiB (3 / CU)
        case 12: return launch_cfg<T, 64, 128, 2, 2, 3>(k, ncls, s);   //  72 KiB (2 / CU)
        case 13: return launch_cfg<T, 128, 64, 4, 2, 3>(k, ncls, s);   //  72 KiB, 8 waves (2 / CU)
        case 14: return launch_cfg<T, 128, 128, 4, 2, 2>(k, ncls, s);  //  64 KiB, 8 waves (2 / CU)
        case 15: return launch_cfg<T, 128, 128, 2, 4, 3>(k, ncls, s);  //  96 KiB, 8 waves
        case 16: return launch_cfg<T, 256, 64, 4, 2, 2>(k, ncls, s);   //  80 KiB, 8 waves
        case 17: return launch_cfg<T, 64, 128, 2, 4, 3>(k, ncls, s);   //  72 KiB, 8 waves (2 / CU)
    }
    set_error("conv: unknown tile config %d", cfg);
    return V2V_EINVAL;
}

static int choose_cfg(long long Mc, int cout, int ncls) {
    if (cout <= 32) return 4;
    auto tiles = [&](int id) {
        const TileCfg* c = find_cfg(id);
        return ceil_div(Mc, c->BM) * ceil_div(cout, c->BN) * ncls;
    };
    if (cout <= 64) return tiles(2) >= 512 ? 2 : 3;
    if (tiles(1) >= 448) return 1;
    if (tiles(2) >= 224) return 2;
    return 3;
}

struct ConvOp : Op {
    ConvKArgs k;
    int ncls, cfg, dtype;
    int launch(hipStream_t s) override {
        return dtype == V2V_BF16 ? launch_typed<bf16_t>(cfg, k, ncls, s) : launch_typed<float>(cfg, k, ncls, s);
    }
    const char* name() const override { return "conv_igemm"; }
};

static int build_conv(const v2v_conv_desc* d, ConvOp* op) {
    if (!d || !d->in || !d->w || !d->out || !d->zero_page) { set_error("conv: null pointer"); return V2V_EINVAL; }
    if (d->dtype != V2V_F32 && d->dtype != V2V_BF16) { set_error("conv: bad dtype"); return V2V_EINVAL; }
    const int vec = d->dtype == V2V_BF16 ? 8 : 4;
    if (d->cin_stride % vec != 0 || d->cin > d->cin_stride) { set_error("conv: cin_stride %d not a multiple of %d", d->cin_stride, vec); return V2V_EINVAL; }
    if (d->stride != 1 && d->stride != 2) { set_error("conv: stride %d (1 or 2 supported)", d->stride); return V2V_EINVAL; }
    if (d->transposed) {
        // nn.ConvTranspose2d(stride s, padding p, output_padding op < s): OH = (H-1)*s - 2p + KH + op
        const int oh0 = (d->H - 1) * d->stride - 2 * d->pad + d->KH, ow0 = (d->W - 1) * d->stride - 2 * d->pad + d->KW;
        if (d->pad_mode != V2V_PAD_ZERO || d->OH < 1 || d->OW < 1 || d->OH > oh0 + d->stride - 1 || d->OW > ow0 + d->stride - 1) {
            set_error("conv: transposed conv output (%d,%d) inconsistent with input (%d,%d) k=%d s=%d p=%d",
                      d->OH, d->OW, d->H, d->W, d->KH, d->stride, d->pad);
            return V2V_EINVAL;
        }
    } else {
        const int oh = (d->H + 2 * d->pad - d->KH) / d->stride + 1, ow = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        if (oh != d->OH || ow != d->OW) { set_error("conv: OH/OW mismatch (%d,%d) vs (%d,%d)", d->OH, d->OW, oh, ow); return V2V_EINVAL; }
        if (d->pad_mode == V2V_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W)) { set_error("conv: reflect pad >= size"); return V2V_EINVAL; }
    }
    if (d->out_mode != V2V_OUT_F32_NCHW && d->cout > d->cout_stride) { set_error("conv: cout_stride"); return V2V_EINVAL; }
    if (((uintptr_t)d->in | (uintptr_t)d->w | (uintptr_t)d->zero_page) & 15) { set_error("conv: operands must be 16-byte aligned"); return V2V_EINVAL; }
    ConvGeom g;
    conv_geom(d->cin_stride, d->cout, d->KH, d->KW, d->transposed, d->stride, d->pad, d->dtype, &g);
    ConvKArgs& k = op->k;
    memset(&k, 0, sizeof(k));
    k.in = (const char*)d->in; k.w = (const char*)d->w; k.zero_page = (const char*)d->zero_page;
    k.bias = d->bias; k.out = (char*)d->out; k.stats = d->stats;
    k.N = d->N; k.H = d->H; k.W = d->W; k.cin_stride = d->cin_stride;
    k.cout = d->cout; k.cout_stride = d->cout_stride; k.cout_p = g.cout_p;
    k.OH = d->OH; k.OW = d->OW;
    k.pad_mode = d->pad_mode;
    if ((long long)d->N * d->H * d->W >= (1ll << 31) || (long long)d->N * d->OH * d->OW >= (1ll << 31)) { set_error("conv: too many pixels"); return V2V_EINVAL; }
    if (d->transposed) {
        // class (a,b) owns output pixels (os*i + a, os*j + b); its grid is ceil((OH-a)/os) x ceil((OW-b)/os)
        k.sm = 1; k.os = d->stride; k.dstep = -1;
        for (int c = 0; c < g.ncls; ++c) {
            const int a = c >> 1, b = c & 1;
            k.OHc[c] = (d->OH - a + k.os - 1) / k.os;
            k.OWc[c] = (d->OW - b + k.os - 1) / k.os;
            k.Mc[c] = d->N * k.OHc[c] * k.OWc[c];
        }
    } else {
        k.sm = d->stride; k.os = 1; k.dstep = 1;
        k.OHc[0] = d->OH; k.OWc[0] = d->OW; k.Mc[0] = d->N * d->OH * d->OW;
    }
    const long long Mc = k.Mc[0];   // class 0 is the largest
    for (int c = 0; c < 4; ++c) {
        k.nkh[c] = g.nkh[c]; k.nkw[c] = g.nkw[c]; k.dh0[c] = g.dh0[c]; k.dw0[c] = g.dw0[c];
        k.ktot[c] = g.ktot[c]; k.kpad[c] = g.kpad[c]; k.woff[c] = g.woff[c];
    }
    k.out_mode = d->out_mode; k.act = d->act; k.act_param = d->act_param; k.out_scale = d->out_scale;
    if (d->fin_counter) {
        if (!d->stats || !d->fin_scale_shift || d->fin_count <= 0 || d->out_mode != V2V_OUT_RAW_F32_NHWC) {
            set_error("conv: in-kernel norm finalize needs stats, fin_scale_shift, fin_count and RAW output"); return V2V_EINVAL;
        }
        k.fin_counter = d->fin_counter; k.fin_gamma = d->fin_gamma; k.fin_beta = d->fin_beta; k.fin_out = d->fin_scale_shift;
        k.fin_rmean = d->fin_running_mean; k.fin_rvar = d->fin_running_var;
        k.fin_eps = d->fin_eps; k.fin_momentum = d->fin_momentum;
        k.fin_inv_count = 1.0 / (double)d->fin_count;
        k.fin_unbias = d->fin_count > 1 ? (double)d->fin_count / (double)(d->fin_count - 1) : 1.0;
    }
    op->ncls = g.ncls;
    op->dtype = d->dtype;
    op->cfg = d->tile ? d->tile : choose_cfg(Mc, d->cout, g.ncls);
    const TileCfg* c = find_cfg(op->cfg);
    if (!c) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
    k.m_tiles = (int)ceil_div(Mc, c->BM);
    k.n_tiles = (int)ceil_div(d->cout, c->BN);
    return 0;
}

}  // namespace v2v

using namespace v2v;

extern "C" int64_t v2v_conv_packed_elems(int32_t cin, int32_t cin_stride, int32_t cout, int32_t KH, int32_t KW,
                                         int32_t transposed, int32_t stride, int32_t pad, int32_t dtype) {
    (void)cin;
    ConvGeom g;
    conv_geom(cin_stride, cout, KH, KW, transposed, stride, pad, dtype, &g);
    return g.total;
}

extern "C" int v2v_conv_pack_weights(const float* w, void* dst, int32_t cin, int32_t cin_stride, int32_t cout,
                                     int32_t KH, int32_t KW, int32_t transposed, int32_t stride, int32_t pad,
                                     int32_t dtype, void* stream) {
    if (!w || !dst) { set_error("pack: null pointer"); return V2V_EINVAL; }
    ConvGeom g;
    conv_geom(cin_stride, cout, KH, KW, transposed, stride, pad, dtype, &g);
    auto op = std::make_unique<PackOp>();
    PackArgs& a = op->a;
    a.w = w; a.dst = dst; a.cin = cin; a.cin_stride = cin_stride; a.cout = cout; a.cout_p = g.cout_p;
    a.KH = KH; a.KW = KW; a.transposed = transposed; a.kstep = stride; a.ncls = g.ncls; a.total = g.total; a.dtype = dtype;
    for (int c = 0; c < 4; ++c) {
        a.nkh[c] = g.nkh[c]; a.nkw[c] = g.nkw[c]; a.kh0[c] = g.kh0[c]; a.kw0[c] = g.kw0[c];
        a.kpad[c] = g.kpad[c]; a.woff[c] = g.woff[c];
    }
    return submit(std::move(op), stream);
}

extern "C" int v2v_conv_stats_rows(const v2v_conv_desc* d) {
    ConvOp op;
    if (build_conv(d, &op) != 0) return V2V_EINVAL;
    return op.ncls * op.k.m_tiles;
}

extern "C" int v2v_conv_tile_config(const v2v_conv_desc* d) {
    ConvOp op;
    if (build_conv(d, &op) != 0) return V2V_EINVAL;
    return op.cfg;
}

extern "C" int v2v_conv2d(const v2v_conv_desc* d, void* stream) {
    auto op = std::make_unique<ConvOp>();
    int rc = build_conv(d, op.get());
    if (rc != 0) return rc;
    return submit(std::move(op), stream);
}
