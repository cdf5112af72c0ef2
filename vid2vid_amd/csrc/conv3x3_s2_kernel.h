// 3x3 / stride 2 / pad 1 (zero) convolution with an LDS-resident input patch -- the single-phase pipeline of conv3x3_pp3_kernel.h
// (one barrier per tap step, two fragment register sets, LDS-DMA between the MFMAs, counted vmcnt) generalised to a TAP TABLE.
// Replaces, for the stride-2 stages of the generators (reference models/networks.py:136,147,156,176,248), the generic implicit-GEMM
// tiles, which fetch the 128 x 128-byte activation tile once PER TAP (9 LDS-DMA tiles per channel chunk, the part of their main
// loop that is latency-bound: profiles/r04_a3_kernel_phases.txt, 19-28 us of a 38 us launch).
//
// Geometry.  Output tile TH x TW = 4 x 32 pixels.  Its (2 TH + 1) x (2 TW + 1) input pixels are held as the four PARITY PLANES of
// the input -- plane (ph, pw) = input rows of parity ph (relative to 2 oh0 - ph ...) -- because inside a plane the pixels that
// consecutive output columns need ARE consecutive: tap (kh, kw) reads plane (kh != 1, kw != 1) at plane pixel
// (r + (kh == 2), c + (kw == 2)) for output pixel (r, c), so an A fragment is 32 consecutive LDS pixels exactly as in the stride-1
// kernel and the (q >> 1) & 7 slot swizzle stays conflict-free.  Plane pixel (y, x) of plane (ph, pw) is input pixel
// (2 (oh0 + y) - ph, 2 (ow0 + x) - pw); pixels outside the image come from the zero page.
//     plane A (1,1): (TH+1) x (TW+1), taps (0,0) (0,2) (2,0) (2,2)      plane C (0,1):  TH    x (TW+1), taps (1,0) (1,2)
//     plane B (1,0): (TH+1) x  TW   , taps (0,1) (2,1)                  plane D (0,0):  TH    x  TW   , tap  (1,1)
// 585 pixels = 74 groups of 8 pixels (1 KB = one wave-level LDS-DMA instruction), 80 with the padding of the last round: 80 KB.
//
// ONE patch buffer, refilled plane by plane.  Double-buffering 80 KB does not fit beside the weight ring, and it is not needed: the
// K steps of a chunk run plane by plane (tap order A A A A B B C C D, the weight slices of the korder-1 packing are simply fetched
// in that order), so a plane's buffer is dead as soon as its last tap has been read and is refilled for the NEXT chunk while the
// other planes' taps execute.  The 74 groups are issued in refill order A B C D as 10 rounds of 8 (one group per wave and round):
//     iteration (chunk c, tap t):   t = 3..8 -> round t-3 of chunk c+1;   t = 0, 1 -> round 6, 7 of chunk c;   t = 2 -> rounds 8, 9 of chunk c
// Hazards, with "a piece issued in iteration i has landed at the end of iteration i + D - 2" (the counted wait below) and the reads of
// step s issued in iteration s - 1 (so a piece must be issued by iteration s - D), `it` = issue iteration relative to the chunk's tap 0:
//     plane A (taps 0-3): free from it = -6, needed by it = -D      rounds 0, 1, 2 at it = -6, -5, -4          (D <= 4)
//     plane B (taps 4,5): free from it = -4, needed by it = 4 - D   rounds 2 .. 5  at it = -4 .. -1
//     plane C (taps 6,7): free from it = -2, needed by it = 6 - D   rounds 5, 6, 7 at it = -1, 0, 1
//     plane D (tap 8)   : free from it = -1, needed by it = 8 - D   rounds 7, 8, 9 at it = 1, 2, 2
// (a plane is free once the reads of its last tap were drained, i.e. behind the barrier of that tap's own iteration).
// Weight ring: slice j + D refills the stage of slice j in iteration j, as in the stride-1 kernel.
#pragma once
#include "conv3x3_pp3_kernel.h"

namespace v2v {

namespace s2k {
// tap order -> kh * 3 + kw of the korder-1 weight packing, and each tap's plane / offsets
constexpr int PERM[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
constexpr int PLANE[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
constexpr int DY[9] = {0, 0, 1, 1, 0, 1, 0, 0, 0};            // kh == 2
constexpr int DX[9] = {0, 1, 0, 1, 0, 0, 0, 1, 0};            // kw == 2
constexpr int NPZ[9] = {1, 1, 2, 1, 1, 1, 1, 1, 1};           // patch pieces per wave issued in the iteration of tap t
constexpr int ROUND0[9] = {6, 7, 8, 0, 1, 2, 3, 4, 5};        // first round issued at tap t (tap 2 also issues round 9)
constexpr int pending_at(int tap, int LB, int D) {
    int x = 0;
    for (int u = 0; u <= D - 3; ++u) x += LB + NPZ[(tap - u + 9) % 9];
    return x;
}
}  // namespace s2k

template <typename T, int TH, int TW, int BN, int D>
__global__ __launch_bounds__(512) void conv3x3_s2_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NW = 8, WGM = 4, WGN = 2, SS = 4;
    constexpr int BM = TH * TW;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NMMA = SS * TM * TN, NRD = SS * (TM + TN);
    // planes: rows, columns, groups of 8 pixels, first pixel
    constexpr int PR_[4] = {TH + 1, TH + 1, TH, TH};
    constexpr int PC_[4] = {TW + 1, TW, TW + 1, TW};
    constexpr int G0 = (PR_[0] * PC_[0] + 7) / 8, G1 = (PR_[1] * PC_[1] + 7) / 8, G2 = (PR_[2] * PC_[2] + 7) / 8, G3 = (PR_[3] * PC_[3] + 7) / 8;
    constexpr int PB_[4] = {0, G0 * 8, (G0 + G1) * 8, (G0 + G1 + G2) * 8};
    constexpr int NG = G0 + G1 + G2 + G3;
    constexpr int NR = (NG + NW - 1) / NW;
    constexpr int PATCH = NR * NW * 1024;
    constexpr int BST = BN * 128;
    constexpr int LB = BN / 8 / NW;
    static_assert(TH == 4 && TW == 32 && NR == 10, "the round -> tap schedule above is that of the 4 x 32 output tile");
    // the window table of the header comment, checked: round r of plane X issued at `it` needs free(X) <= it <= first(X) - D
    static_assert(G0 <= 3 * NW && G0 > 2 * NW, "plane A ends inside round 2 (issued at it = -4 >= -6, <= -D)");
    static_assert(G0 + G1 > 5 * NW && G0 + G1 <= 6 * NW, "plane B ends inside round 5 (it = -1 <= 4 - D)");
    static_assert(G0 + G1 + G2 > 7 * NW && G0 + G1 + G2 <= 8 * NW, "plane C ends inside round 7 (it = 1 <= 6 - D), plane D starts there (it = 1 >= -1)");
    static_assert(D == 3 || D == 4, "weight slices in flight");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1 && LB >= 1, "wave tile / weight loader rounds");
    static_assert(PATCH + D * BST <= 160 * 1024 && PATCH >= 40960, "LDS (the epilogue's scratch lives in the patch buffer)");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bring = smem + PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int cls = 0;

    const int tiles = p.m_tiles * p.n_tiles;
    const int lin = xcd_remap(blockIdx.x, tiles);
    const int nt = lin / p.m_tiles;
    const int mt = lin - nt * p.m_tiles;
    const int tpi = p.tiles_h * p.tiles_w;
    const int n_img = mt / tpi;
    const int trem = mt - n_img * tpi;
    const int th = trem / p.tiles_w;
    const int oh0 = th * TH, ow0 = (trem - th * p.tiles_w) * TW;

    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc = cs * (int)sizeof(T) / 128;
    const int nsteps = ncc * 9;
    const char* const zp = p.zero_page;

    // ---------------- patch loader geometry: round r, this wave's group r * NW + wid ----------------
    unsigned pp[NR];
    unsigned pok = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int q = (r * NW + wid) * 8 + (lane >> 3);          // LDS pixel (refill order A B C D, each plane row-major)
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pl = q >= PB_[3] ? 3 : q >= PB_[2] ? 2 : q >= PB_[1] ? 1 : 0;
        const int pbase = pl == 3 ? PB_[3] : pl == 2 ? PB_[2] : pl == 1 ? PB_[1] : 0;
        const int pcols = (pl == 0 || pl == 2) ? TW + 1 : TW;
        const int prows = pl < 2 ? TH + 1 : TH;
        const int ph = pl < 2 ? 1 : 0, pw = (pl == 0 || pl == 2) ? 1 : 0;
        const int loc = q - pbase;
        const int y = loc / pcols, x = loc - y * pcols;
        int ih = 2 * (oh0 + y) - ph, iw = 2 * (ow0 + x) - pw;
        const bool ok = loc < prows * pcols && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        pp[r] = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
        pok |= (ok ? 1u : 0u) << r;
    }
    auto issue_round = [&](int r, int chunk) {
        const int cg = chunk < ncc ? chunk : ncc - 1;              // tail: a harmless reload keeps the DMA counts uniform
        const char* src = ((pok >> r) & 1u) ? p.in + pp[r] + cg * 128 : zp;
        glds16(src, smem + (r * NW + wid) * 1024);
    };

    // ---------------- weight loader geometry (korder 1: slice (chunk, tap) = 128 bytes of every row) ----------------
    const int lrow = wid * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
    const char* wp[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        long long r = (long long)nt * BN + lrow + NW * 8 * i;
        r = r < p.cout_p ? r : p.cout_p - 1;
        wp[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T);
    }
    // slice of tap index `tap` (this kernel's order) of chunk `chunk` into ring stage `stage`
    auto issue_w = [&](int i, int chunk, int tap_kk, int stage) {
        const int cg = chunk < ncc ? chunk : ncc - 1;              // tail duplicate into a free stage
        glds16(wp[i] + (long long)(cg * 9 + tap_kk) * 128, bring + stage * BST + wid * 1024 + i * NW * 1024);
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    int qr[TM], qc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m0 = wm * WM + i * 32;
        qr[i] = m0 / TW;
        qc[i] = (m0 % TW) + lr;
    }
    int foff[SS];
#pragma unroll
    for (int s = 0; s < SS; ++s) foff[s] = ((s * 2 + hi) ^ ((lr >> 1) & 7)) << 4;
    const int b_row_off = (wn * WN + lr) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Frag fa[2][SS][TM], fb[2][SS][TN];

    auto read_frag = [&](auto qc_, auto parc, const char* (&arow)[TM], int (&ax)[TM], const char* pb) {
        constexpr int q = decltype(qc_)::value;
        constexpr int PARN = decltype(parc)::value;
        if constexpr (q < SS * TM) {
            constexpr int i = q / SS, s = q % SS;
            fa[PARN][s][i] = *reinterpret_cast<const Frag*>(arow[i] + (((s * 2 + hi) ^ ax[i]) << 4));
        } else {
            constexpr int s = (q - SS * TM) / TN, j = (q - SS * TM) % TN;
            fb[PARN][s][j] = *reinterpret_cast<const Frag*>(pb + j * 32 * 128 + foff[s]);
        }
    };
    // A-fragment rows of tap index `tap`
    auto a_rows = [&](auto tc, const char* (&arow)[TM], int (&ax)[TM]) {
        constexpr int tap = decltype(tc)::value;
        constexpr int pl = s2k::PLANE[tap];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int rv = qr[i];
            asm volatile("" : "+v"(rv));                     // opaque: no hoisting of 9 x TM address sets out of the chunk loop
            const int q = PB_[pl] + (rv + s2k::DY[tap]) * PC_[pl] + qc[i] + s2k::DX[tap];
            arow[i] = smem + q * 128;
            ax[i] = (q >> 1) & 7;
        }
    };

    // ---------------- prologue ----------------
    // rounds 0 .. 4 of chunk 0 and weight slices 0 .. D-2, then -- as the "iteration -1" of the steady state -- slice D-1 and round 5
#pragma unroll
    for (int r = 0; r < 5; ++r) issue_round(r, 0);
#pragma unroll
    for (int t = 0; t < D - 1; ++t)
#pragma unroll
        for (int i = 0; i < LB; ++i) issue_w(i, 0, s2k::PERM[t], t);
#pragma unroll
    for (int i = 0; i < LB; ++i) issue_w(i, 0, s2k::PERM[D - 1], D - 1);
    issue_round(5, 0);
    wait_vmcnt<LB + 1>();                                    // everything before "iteration -1" has landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    {
        const char* arow[TM]; int ax[TM];
        a_rows(std::integral_constant<int, 0>{}, arow, ax);
        const char* const pb = bring + b_row_off;
        static_for<NRD>([&](auto qc_) { read_frag(qc_, std::integral_constant<int, 0>{}, arow, ax, pb); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int stage = 0, cc = 0;                                    // weight stage of the step being multiplied, its chunk

    auto iteration = [&](auto tc, auto pc) {
        constexpr int TAP = decltype(tc)::value;
        constexpr int PAR = decltype(pc)::value;
        constexpr int NT = (TAP + 1) % 9;                    // tap of the step whose fragments are read now
        constexpr int WT = (TAP + D) % 9, WC = (TAP + D) / 9; // tap / chunk offset of the weight slice issued now (step + D)
        constexpr int NPZ = s2k::NPZ[TAP];
        constexpr int NDMA = LB + NPZ;
        constexpr int RSLOTS = (NMMA * 5) / 8 > 0 ? (NMMA * 5) / 8 : 1;
        constexpr int RPS = (NRD + RSLOTS - 1) / RSLOTS;
        constexpr int RUSED = (NRD + RPS - 1) / RPS;

        __builtin_amdgcn_s_barrier();                        // B_step
        const char* arow[TM]; int ax[TM];
        a_rows(std::integral_constant<int, NT>{}, arow, ax);
        const int nstage = stage + 1 == D ? 0 : stage + 1;
        const char* const pb = bring + nstage * BST + b_row_off;
        auto dma = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < LB) issue_w(d, cc + WC, s2k::PERM[WT], stage);               // slice step + D refills the stage of slice `step`
            else                  issue_round(s2k::ROUND0[TAP] + (d - LB), TAP >= 3 ? cc + 1 : cc);
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
            Mma<T>::run(fa[PAR][s][i], fb[PAR][s][j], acc[i][j]);
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
        wait_vmcnt<s2k::pending_at(TAP, LB, D)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stage = nstage;
        if constexpr (TAP == 8) ++cc;
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

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tail duplicates must land before the LDS is reused
    __syncthreads();

    const int OH = p.OH, OW = p.OW;
    auto pix_of = [&](int row) -> int {
        const int oh = oh0 + row / TW, ow = ow0 + (row & (TW - 1));
        if (oh >= OH || ow >= OW) return -1;
        return (n_img * OH + oh) * OW + ow;
    };
    conv_epilogue<T, BM, BN, WGM, WGN, false>(p, acc, smem, tid, wm, wn, false, cls, tiles, lin, 0, 1, nt, mt, pix_of);
}

template <typename T, int TH, int TW, int BN, int D>
static int launch_s2_cfg(const ConvKArgs& k, hipStream_t s) {
    constexpr int NW = 8;
    constexpr int NG = ((TH + 1) * (TW + 1) + 7) / 8 + ((TH + 1) * TW + 7) / 8 + (TH * (TW + 1) + 7) / 8 + (TH * TW + 7) / 8;
    const size_t lds = (size_t)((NG + NW - 1) / NW) * NW * 1024 + (size_t)D * BN * 128;
    auto kern = conv3x3_s2_kernel<T, TH, TW, BN, D>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles), 1u, 1u);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, k);
    return check_launch();
}

// stride-2 patch tile configurations (ids 100..103)
static const PatchCfg kS2Cfgs[] = {{100, 4, 32, 64}, {101, 4, 32, 128}, {102, 4, 32, 64}, {103, 4, 32, 128}};
static inline const PatchCfg* find_s2_cfg(int id) {
    for (const PatchCfg& c : kS2Cfgs)
        if (c.id == id) return &c;
    return nullptr;
}

template <typename T>
static inline int launch_s2_typed(int cfg, const ConvKArgs& k, hipStream_t s) {
    switch (cfg) {
        case 100: return launch_s2_cfg<T, 4, 32, 64, 4>(k, s);     // 128 px x  64, 4 slices in flight, 112 KiB
        case 101: return launch_s2_cfg<T, 4, 32, 128, 4>(k, s);    // 128 px x 128, wave tile 32 x 64, 144 KiB
        case 102: return launch_s2_cfg<T, 4, 32, 64, 3>(k, s);     // as 100, 3 slices, 104 KiB
        case 103: return launch_s2_cfg<T, 4, 32, 128, 3>(k, s);    // as 101, 3 slices, 128 KiB
    }
    set_error("conv: unknown stride-2 patch tile config %d", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
