// ConvTranspose2d(3x3, stride 2, padding 1[, output_padding 1]) with an LDS-resident input patch: ALL FOUR output-parity classes of a
// tile of input positions in one workgroup, on the single-phase pipeline of conv3x3_pp3_kernel.h (one barrier per step, two fragment
// register sets, LDS-DMA between the MFMAs, counted vmcnt -- pp3::np_at / pp3::pending_at unchanged: a channel chunk is still 9 steps).
// Replaces, for the up-sampling stages of the generators (reference models/networks.py:170-176,254-260: the img / flow branches and
// the foreground tower), the generic implicit-GEMM launch of 4 class grids, which fetches the activation tile once per tap and class
// (9 LDS-DMA tiles per channel chunk; profiles/r04_a3_kernel_phases.txt: main loops of 10-22 us with 2x spreads between the classes).
//
// out[2a + py][2b + px] = sum over the taps of class (py, px) of in[a + dy][b + dx] . W[ky][kx]:
//     py == 0: ky = 1 (dy 0);  py == 1: ky = 0 (dy 1), ky = 2 (dy 0);  the same along x       (oy = 2 iy - 1 + ky)
// so the 1 + 2 + 2 + 4 = 9 (class, tap) pairs of a channel chunk all read the SAME (TH+1) x (TW+1) patch of input pixels at offsets
// (dy, dx) in {0,1}^2 -- exactly the 9 tap steps of the stride-1 kernel with another offset table, a weight slice per (ky, kx) (the
// full-tap korder-1 packing of the transposed layer, v2v_conv_pack_weights korder 2) and ONE OF FOUR accumulator sets per step.
// Epilogue: the shared conv_epilogue once per class (its own statistics rows cls * m_tiles + mt, as the generic kernel's class grids).
#pragma once
#include "conv3x3_pp3_kernel.h"

namespace v2v {

namespace t2k {
// step -> class (py * 2 + px), kernel tap ky * 3 + kx, input offset (dy, dx)
constexpr int CLS[9] = {3, 3, 3, 3, 1, 1, 2, 2, 0};
constexpr int KK[9]  = {0, 2, 6, 8, 3, 5, 1, 7, 4};
constexpr int DY[9]  = {1, 1, 0, 0, 0, 0, 1, 0, 0};
constexpr int DX[9]  = {1, 0, 1, 0, 1, 0, 0, 0, 0};
}  // namespace t2k

template <typename T, int TH, int TW, int BN, int D>
__global__ __launch_bounds__(512) void conv3x3_t2_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NW = 8, WGM = 4, WGN = 2, SS = 4;
    constexpr int BM = TH * TW;                               // input positions (a, b) per tile; 4 BM output pixels
    constexpr int PW = TW + 1, PR = (TH + 1) * PW;
    constexpr int NG = (PR + 7) / 8;
    constexpr int GP = (NG + NW - 1) / NW;
    constexpr int PATCH = GP * NW * 1024;
    constexpr int BST = BN * 128;
    constexpr int LB = BN / 8 / NW;
    constexpr int NPT = 9 - D;
    constexpr int PPT = (GP + NPT - 1) / NPT;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NMMA = SS * TM * TN, NRD = SS * (TM + TN);
    static_assert(TW % 32 == 0 && (TW & (TW - 1)) == 0, "a 32-row fragment must lie inside one tile row");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1 && LB >= 1, "wave tile / weight loader rounds");
    static_assert(D >= 3 && D <= 5, "weight slices in flight");
    static_assert(2 * PATCH + D * BST <= 160 * 1024 && 2 * PATCH >= 40960, "LDS (the epilogue's scratch lives in the patch buffers)");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bring = smem + 2 * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;

    const int tiles = p.m_tiles * p.n_tiles;
    const int lin = xcd_remap(blockIdx.x, tiles);
    const int nt = lin / p.m_tiles;
    const int mt = lin - nt * p.m_tiles;
    const int tpi = p.tiles_h * p.tiles_w;
    const int n_img = mt / tpi;
    const int trem = mt - n_img * tpi;
    const int th = trem / p.tiles_w;
    const int a0 = th * TH, b0 = (trem - th * p.tiles_w) * TW;

    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc = cs * (int)sizeof(T) / 128;
    const int nsteps = ncc * 9;
    const char* const zp = p.zero_page;

    // ---------------- patch loader geometry: input pixels (a0 .. a0 + TH, b0 .. b0 + TW), zero beyond the image ----------------
    unsigned pp[GP];
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        int ih = a0 + pr, iw = b0 + pc;
        const bool ok = q < PR && ih < H && iw < W;
        ih = ih >= H ? H - 1 : ih;
        iw = iw >= W ? W - 1 : iw;
        pp[k] = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
        pok |= (ok ? 1u : 0u) << k;
    }
    auto issue_patch = [&](int k, int cc_local, char* buf) {
        const int cg = cc_local < ncc ? cc_local : ncc - 1;
        const char* src = ((pok >> k) & 1u) ? p.in + pp[k] + cg * 128 : zp;
        glds16(src, buf + (k * NW + wid) * 1024);
    };

    // ---------------- weight loader geometry (full-tap korder-1 packing: slice (chunk, ky * 3 + kx)) ----------------
    const int lrow = wid * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
    const char* wp[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        long long r = (long long)nt * BN + lrow + NW * 8 * i;
        r = r < p.cout_p ? r : p.cout_p - 1;
        wp[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T);
    }
    auto issue_w = [&](int i, int chunk, int kk, int stage) {
        const int cg = chunk < ncc ? chunk : ncc - 1;
        glds16(wp[i] + (long long)(cg * 9 + kk) * 128, bring + stage * BST + wid * 1024 + i * NW * 1024);
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    int qb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m0 = wm * WM + i * 32;
        qb[i] = (m0 / TW) * PW + (m0 % TW) + lr;
    }
    int foff[SS];
#pragma unroll
    for (int s = 0; s < SS; ++s) foff[s] = ((s * 2 + hi) ^ ((lr >> 1) & 7)) << 4;
    const int b_row_off = (wn * WN + lr) * 128;

    f32x16 acc[4][TM][TN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
    Frag fa[2][SS][TM], fb[2][SS][TN];

    auto read_frag = [&](auto qc, auto parc, const char* (&arow)[TM], int (&ax)[TM], const char* pb) {
        constexpr int q = decltype(qc)::value;
        constexpr int PARN = decltype(parc)::value;
        if constexpr (q < SS * TM) {
            constexpr int i = q / SS, s = q % SS;
            fa[PARN][s][i] = *reinterpret_cast<const Frag*>(arow[i] + (((s * 2 + hi) ^ ax[i]) << 4));
        } else {
            constexpr int s = (q - SS * TM) / TN, j = (q - SS * TM) % TN;
            fb[PARN][s][j] = *reinterpret_cast<const Frag*>(pb + j * 32 * 128 + foff[s]);
        }
    };

    // ---------------- prologue: patch 0 and weight slices 0 .. D-1; step 0's fragments into set 0 ----------------
#pragma unroll
    for (int k = 0; k < GP; ++k) issue_patch(k, 0, smem);
#pragma unroll
    for (int t = 0; t < D; ++t)
#pragma unroll
        for (int i = 0; i < LB; ++i) issue_w(i, 0, t2k::KK[t], t);
    wait_vmcnt<(D - 1) * LB>();
    __builtin_amdgcn_s_barrier();
    {
        const char* arow[TM]; int ax[TM];
        constexpr int tq0 = t2k::DY[0] * PW + t2k::DX[0];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int q = qb[i] + tq0; arow[i] = smem + q * 128; ax[i] = (q >> 1) & 7; }
        const char* const pb = bring + b_row_off;
        static_for<NRD>([&](auto qc) { read_frag(qc, std::integral_constant<int, 0>{}, arow, ax, pb); });
    }
    wait_vmcnt<(D - 2) * LB>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int stage = 0, cc = 0;
    const char* pa = smem;
    char* pn = smem + PATCH;

    auto iteration = [&](auto tc, auto pc) {
        constexpr int TAP = decltype(tc)::value;
        constexpr int PAR = decltype(pc)::value;
        constexpr int NT = (TAP + 1) % 9;
        constexpr int tq = t2k::DY[NT] * PW + t2k::DX[NT];
        constexpr int WT = (TAP + D) % 9, WC = (TAP + D) / 9;
        constexpr int k0 = pp3::cmin(TAP * PPT, GP);
        constexpr int npz = pp3::np_at(TAP, GP, NPT);
        constexpr int NDMA = LB + npz;
        constexpr int RSLOTS = (NMMA * 5) / 8 > 0 ? (NMMA * 5) / 8 : 1;
        constexpr int RPS = (NRD + RSLOTS - 1) / RSLOTS;
        constexpr int RUSED = (NRD + RPS - 1) / RPS;
        constexpr int CL = t2k::CLS[TAP];

        __builtin_amdgcn_s_barrier();
        const char* arow[TM]; int ax[TM];
        {
            const char* const pbuf = TAP == 8 ? pn : pa;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int qv = qb[i];
                asm volatile("" : "+v"(qv));
                const int q = qv + tq;
                arow[i] = pbuf + q * 128;
                ax[i] = (q >> 1) & 7;
            }
        }
        const int nstage = stage + 1 == D ? 0 : stage + 1;
        const char* const pb = bring + nstage * BST + b_row_off;
        auto dma = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < LB) issue_w(d, cc + WC, t2k::KK[WT], stage);
            else                  issue_patch(k0 + d - LB, cc + 1, pn);
        };
        auto reads_of_slot = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            static_for<RPS>([&](auto rc) {
                constexpr int q = m * RPS + decltype(rc)::value;
                if constexpr (q < NRD) read_frag(std::integral_constant<int, q>{}, std::integral_constant<int, 1 - PAR>{}, arow, ax, pb);
            });
        };
        static_for<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int s = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
            Mma<T>::run(fa[PAR][s][i], fb[PAR][s][j], acc[CL][i][j]);
            if constexpr (m < RUSED) {
                __builtin_amdgcn_sched_barrier(0);
                reads_of_slot(mc);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (m - RUSED < NDMA && m < NMMA - 1) {
                __builtin_amdgcn_sched_barrier(0);
                dma(std::integral_constant<int, m - RUSED>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        constexpr int DMA_IN_SLOTS = pp3::cmin(NDMA, NMMA - 1 - RUSED > 0 ? NMMA - 1 - RUSED : 0);
        static_for<NDMA - DMA_IN_SLOTS>([&](auto dc) { dma(std::integral_constant<int, DMA_IN_SLOTS + decltype(dc)::value>{}); });
        wait_vmcnt<pp3::pending_at(TAP, GP, LB, D)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stage = nstage;
        if constexpr (TAP == 8) {
            ++cc;
            const char* t = pa; pa = pn; pn = const_cast<char*>(t);
        }
    };
    auto chunk = [&](auto par0c) {
        constexpr int P0 = decltype(par0c)::value;
        iteration(std::integral_constant<int, 0>{}, std::integral_constant<int, P0>{});
        iteration(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 - P0>{});
        iteration(std::integral_constant<int, 2>{}, std::integral_constant<int, P0>{});
        iteration(std::integral_constant<int, 3>{}, std::integral_constant<int, 1 - P0>{});
        iteration(std::integral_constant<int, 4>{}, std::integral_constant<int, P0>{});
        iteration(std::integral_constant<int, 5>{}, std::integral_constant<int, 1 - P0>{});
        iteration(std::integral_constant<int, 6>{}, std::integral_constant<int, P0>{});
        iteration(std::integral_constant<int, 7>{}, std::integral_constant<int, 1 - P0>{});
        iteration(std::integral_constant<int, 8>{}, std::integral_constant<int, P0>{});
    };
    int c = 0;
    for (; c + 1 < ncc; c += 2) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
    }
    if (c < ncc) chunk(std::integral_constant<int, 0>{});
    (void)nsteps;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---------------- epilogue: one pass of the shared epilogue per output-parity class ----------------
    const int OH = p.OH, OW = p.OW;
    static_for<4>([&](auto cc_) {
        constexpr int CL = decltype(cc_)::value;
        auto pix_of = [&](int row) -> int {
            const int oy = 2 * (a0 + row / TW) + (CL >> 1), ox = 2 * (b0 + (row & (TW - 1))) + (CL & 1);
            if (oy >= OH || ox >= OW) return -1;
            return (n_img * OH + oy) * OW + ox;
        };
        conv_epilogue<T, BM, BN, WGM, WGN, false>(p, acc[CL], smem, tid, wm, wn, false, CL, tiles, lin, 0, 1, nt, CL * p.m_tiles + mt, pix_of);
        __syncthreads();                                      // the statistics / transposition scratch is reused by the next class
    });
}

template <typename T, int TH, int TW, int BN, int D>
static int launch_t2_cfg(const ConvKArgs& k, hipStream_t s) {
    constexpr int NW = 8;
    constexpr int GP = (((TH + 1) * (TW + 1) + 7) / 8 + NW - 1) / NW;
    const size_t lds = (size_t)2 * GP * NW * 1024 + (size_t)D * BN * 128;
    auto kern = conv3x3_t2_kernel<T, TH, TW, BN, D>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles), 1u, 1u);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, k);
    return check_launch();
}

// transposed stride-2 patch tile configurations (ids 110..113): (TH, TW) = tile of INPUT positions
static const PatchCfg kT2Cfgs[] = {{110, 4, 32, 64}, {111, 4, 32, 128}, {112, 8, 32, 64}, {113, 4, 32, 64},
                                   {114, 8, 32, 32}};    // 114: geometry only -- the persistent kernel of conv3x3_one_kernel.h (launch_one_typed)
static inline const PatchCfg* find_t2_cfg(int id) {
    for (const PatchCfg& c : kT2Cfgs)
        if (c.id == id) return &c;
    return nullptr;
}

template <typename T>
static inline int launch_t2_typed(int cfg, const ConvKArgs& k, hipStream_t s) {
    switch (cfg) {
        case 110: return launch_t2_cfg<T, 4, 32, 64, 4>(k, s);     // 128 positions x  64 channels x 4 classes, 80 KiB (two workgroups per CU)
        case 111: return launch_t2_cfg<T, 4, 32, 128, 4>(k, s);    // 128 positions x 128 channels, 112 KiB
        case 112: return launch_t2_cfg<T, 8, 32, 64, 4>(k, s);     // 256 positions x  64 channels, wave tile 64 x 32 per class, 112 KiB
        case 113: return launch_t2_cfg<T, 4, 32, 64, 3>(k, s);     // as 110, 3 slices in flight, 72 KiB
    }
    set_error("conv: unknown transposed stride-2 patch tile config %d", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
