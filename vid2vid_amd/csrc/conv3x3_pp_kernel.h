// 3x3 stride-1 convolution, LDS-resident input patch, PING-PONG wave groups (gfx950).
//
// Same data movement as conv3x3_patch_kernel.h (channel-chunk outer / tap inner, double-buffered patch, weight ring),
// different schedule.  Measured on the patch kernel (profiles/r01_v7_ablate_patch.txt): one tap step of a 128x64
// tile costs ~790 cycles against 256 cycles of MFMA work, because every wave runs the step's phases back to back --
// ~100 cycles of issue time per LDS-DMA piece, the ds_read latency, then the MFMAs -- and with one wave per SIMD
// nothing fills the matrix pipe during the first two.  Here a workgroup is 8 waves = two GROUPS of four (one wave of
// each group per SIMD) that run half a step apart:
//
//      phase 2j   :  group A  LOAD(j)   = ds_read step j's fragments, issue its DMA share, vmcnt/lgkmcnt   | group B  MMA(j-1)
//      phase 2j+1 :  group A  MMA(j)    = 16 MFMAs from registers (s_setprio 1)                            | group B  LOAD(j)
//
// with ONE workgroup barrier between phases, so each SIMD always has one wave in a pure-MFMA phase beside one wave in
// a memory phase (cdna guide 5.5, T3+T4/T5: the 8-phase template's role split).  Tile = TH x TW pixels x BN channels,
// waves 4(M) x 2(N): group A owns the upper half of the pixel rows, B the lower; wave tile (TH*TW/4) x (BN/2).
//
// Hazards (phases are totally ordered by the barriers; p(X) = phase of X):
//   RAW  tile j is read in LOAD_A(j) [2j] and LOAD_B(j) [2j+1].  Every wave waits for ITS share of tile j+1 (weights;
//        and, being older in its in-order vmcnt queue, every patch piece issued up to tap 7) at the END of its
//        LOAD(j), i.e. by phase 2j+1 at the latest; the barrier after that phase publishes it before LOAD_A(j+1).
//        The counted wait leaves the younger DMAs in flight: (DB-1) weight shares and the patch pieces of the last
//        DB steps; taps >= 9-DB carry no patch pieces, so the wait of tap 8 retires the whole next patch.
//   WAR  LOAD(j) refills weight stage (j+DB)%NSB = stage of tile j-1, last read in LOAD_B(j-1) [2j-1], whose ds_reads are
//        drained (lgkmcnt(0)) before that phase's barrier.  The patch buffer of chunk c+1 is refilled from LOAD_A(9c)
//        on; its previous contents (chunk c-1) were last read in LOAD_B(9c-1), one phase earlier.
#pragma once
#include "conv3x3_patch_kernel.h"

namespace v2v {

template <typename T, int TH, int TW, int BN, int NSB>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BM = TH * TW;
    constexpr int PW = TW + 2, PR = (TH + 2) * PW;
    constexpr int NW = 8, WGM = 4, WGN = 2;
    constexpr int NG = (PR + 7) / 8;
    constexpr int GP = (NG + NW - 1) / NW;                    // patch pieces per wave per chunk
    constexpr int PATCH = GP * NW * 1024;
    constexpr int BST = BN * 128;
    constexpr int LB = BN / 8 / NW;                           // weight pieces per wave per slice
    constexpr int DB = NSB - 1;                               // weight slices in flight (2 or 3)
    constexpr int NPT = 9 - DB;                               // taps 0..NPT-1 may carry next-chunk patch pieces: the last
                                                              // one must be forced by the wait of tap 8
    constexpr int PPT = (GP + NPT - 1) / NPT;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TW % 32 == 0 && (TW & (TW - 1)) == 0, "a 32-row fragment must lie inside one tile row");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile");
    static_assert(BN % (8 * NW) == 0 && LB >= 1, "weight loader rounds");
    static_assert(NSB == 3 || NSB == 4, "weight ring depth");
    static_assert(2 * PATCH + NSB * BST <= 160 * 1024, "LDS");
    static_assert(2 * PATCH >= 32768, "epilogue scratch lives in the patch buffers");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bring = smem + 2 * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const bool grpA = wid < 4;
    const int cls = 0;

    const int tiles = p.m_tiles * p.n_tiles;
    const int S = p.splitk;
    const int lin_all = xcd_remap(blockIdx.x, tiles * S);
    int lin, slice, nt, mt, n_img, th, twi;
    patch_tile_index(p, lin_all, lin, slice, nt, mt, n_img, th, twi);
    const int oh0 = th * TH, ow0 = twi * TW;

    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc_all = cs * (int)sizeof(T) / 128;
    int ccb, ncc;
    patch_chunk_range(ncc_all, slice, S, ccb, ncc);
    const int nsteps = ncc * 9;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const char* const zp = p.zero_page;

    // ---------------- patch loader geometry (as conv3x3_patch_kernel, 8 loading waves) ----------------
    unsigned pp[GP];
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        int ih = oh0 + pr - 1, iw = ow0 + pc - 1;
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
    auto issue_patch = [&](int k, int cc_local, char* buf) {
        const int cg = cc_local < ncc ? cc_local : ncc - 1;    // tail: a harmless reload keeps the DMA counts uniform
        const char* src = (((pok >> k) & 1u) && !(p.ablate & 1)) ? p.in + pp[k] + (ccb + cg) * 128 : zp;
        glds16(src, buf + (k * NW + wid) * 1024);
    };

    // ---------------- weight loader geometry ----------------
    const int lrow = wid * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);         // (64*i >> 1) & 7 == 0
    const char* wp[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        long long r = (long long)nt * BN + lrow + NW * 8 * i;
        r = r < p.cout_p ? r : p.cout_p - 1;
        wp[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T) + (long long)ccb * 9 * 128;
    }
    auto issue_w = [&](int step, int stage) {
        const int sg = step < nsteps ? step : nsteps - 1;      // tail duplicate into a free stage
        char* dst = bring + stage * BST + wid * 1024;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            glds16(wp[i] + ((p.ablate & 2) ? 0ll : (long long)sg * 128), dst + i * NW * 1024);
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    int qb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m0 = wm * WM + i * 32;
        qb[i] = (m0 / TW) * PW + (m0 % TW) + lr;
    }
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = ((s * 2 + hi) ^ ((lr >> 1) & 7)) << 4;
    const int b_row_off = (wn * WN + lr) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Frag fa[4][TM], fb[4][TN];                               // one step's fragments: written in LOAD, consumed in MMA

    // ---------------- prologue: patch 0 and weight slices 0, 1; slice 0 (and the patch) must land ----------------
#pragma unroll
    for (int k = 0; k < GP; ++k) issue_patch(k, 0, smem);
#pragma unroll
    for (int t = 0; t < DB; ++t) issue_w(t, t);
    wait_vmcnt<(DB - 1) * LB>();
    __builtin_amdgcn_s_barrier();

    int step = 0, stage = 0, wstage = DB;
    int cc = 0;
    const char* pa = smem;
    char* pnext = smem + PATCH;

    // LOAD(step): fragments of `step` -> registers, this wave's DMA share of step+2, then its share of step+1 landed
    auto load_phase = [&](auto tc) {
        constexpr int tap = decltype(tc)::value;
        if (!(p.ablate & 16)) {
            const char* const pb = bring + stage * BST + b_row_off;
            constexpr int tq = (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int qv = qb[i];
                asm volatile("" : "+v"(qv));       // opaque: keeps the 9 x TM x 4 fragment addresses from being hoisted
                const int q = qv + tq;              // out of the chunk loop as loop invariants (register pressure -> spills)
                const int abase = q * 128, ax = (q >> 1) & 7;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    fa[s][i] = *reinterpret_cast<const Frag*>(pa + abase + (((s * 2 + hi) ^ ax) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[s][j] = *reinterpret_cast<const Frag*>(pb + j * 32 * 128 + foff[s]);
        }
        // DMA order inside a step: weights first, then the next chunk's patch pieces.  The wait must retire weight
        // slice step+1 (issued DB-1 steps ago); everything younger may stay in flight -- the patch pieces of the last
        // DB-1 steps included, so a piece is only forced DB steps after it was issued (the in-order vmcnt would
        // otherwise expose its full latency one step later although nobody reads it before the next chunk).
        issue_w(step + DB, wstage);
        constexpr int k0 = patch::cmin(tap * PPT, GP), k1 = patch::cmin((tap + 1) * PPT, GP);
        if constexpr (tap < NPT) {
#pragma unroll
            for (int k = k0; k < k1; ++k) issue_patch(k, cc + 1, pnext);
        }
        constexpr int np0 = tap < NPT ? k1 - k0 : 0;
        constexpr int tm1 = (tap + 8) % 9, tm2 = (tap + 7) % 9;
        constexpr int np1 = tm1 < NPT ? patch::cmin((tm1 + 1) * PPT, GP) - patch::cmin(tm1 * PPT, GP) : 0;
        constexpr int np2 = tm2 < NPT ? patch::cmin((tm2 + 1) * PPT, GP) - patch::cmin(tm2 * PPT, GP) : 0;
        wait_vmcnt<(DB - 1) * LB + np0 + np1 + (DB == 3 ? np2 : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: the stage may be refilled
        ++step;
        stage = stage + 1 == NSB ? 0 : stage + 1;
        wstage = wstage + 1 == NSB ? 0 : wstage + 1;
        if constexpr (tap == 8) {
            ++cc;
            pa = smem + (cc & 1) * PATCH;
            pnext = smem + ((cc + 1) & 1) * PATCH;
        }
    };
    auto mma_phase = [&]() {
        if (p.ablate & 16) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    if (grpA) {
        for (int c = 0; c < ncc; ++c) {
            auto stepA = [&](auto tc) {
                load_phase(tc);
                __builtin_amdgcn_s_barrier();
                mma_phase();
                __builtin_amdgcn_s_barrier();
            };
            stepA(std::integral_constant<int, 0>{}); stepA(std::integral_constant<int, 1>{}); stepA(std::integral_constant<int, 2>{});
            stepA(std::integral_constant<int, 3>{}); stepA(std::integral_constant<int, 4>{}); stepA(std::integral_constant<int, 5>{});
            stepA(std::integral_constant<int, 6>{}); stepA(std::integral_constant<int, 7>{}); stepA(std::integral_constant<int, 8>{});
        }
    } else {
        bool first = true;
        for (int c = 0; c < ncc; ++c) {
            auto stepB = [&](auto tc) {
                if (!first) mma_phase();                      // MMA(step-1) beside group A's LOAD(step)
                first = false;
                __builtin_amdgcn_s_barrier();
                load_phase(tc);
                __builtin_amdgcn_s_barrier();
            };
            stepB(std::integral_constant<int, 0>{}); stepB(std::integral_constant<int, 1>{}); stepB(std::integral_constant<int, 2>{});
            stepB(std::integral_constant<int, 3>{}); stepB(std::integral_constant<int, 4>{}); stepB(std::integral_constant<int, 5>{});
            stepB(std::integral_constant<int, 6>{}); stepB(std::integral_constant<int, 7>{}); stepB(std::integral_constant<int, 8>{});
        }
        mma_phase();                                          // MMA(nsteps-1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tail duplicates must land before the LDS is reused
    __syncthreads();

    conv_epilogue<T, BM, BN, WGM, WGN>(p, acc, smem, tid, wm, wn, false, cls, tiles, lin, slice, S, nt, mt,
        [&](int row) -> int {                  // TW is a power of two; N*OH*OW < 2^31 (host check)
            const int oh = oh0 + row / TW, ow = ow0 + (row & (TW - 1));
            if (oh >= H || ow >= W) return -1;
            return (n_img * H + oh) * W + ow;
        }, oh0 + TH <= H && ow0 + TW <= W);
}

template <typename T, int TH, int TW, int BN, int NSB>
static int launch_pp_cfg(const ConvKArgs& k, hipStream_t s) {
    constexpr int GP = (((TH + 2) * (TW + 2) + 7) / 8 + 7) / 8;
    const size_t lds = (size_t)2 * GP * 8 * 1024 + (size_t)NSB * BN * 128;
    auto kern = conv3x3_pp_kernel<T, TH, TW, BN, NSB>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles * k.splitk), 1u);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, k);
    return check_launch();
}

// ping-pong tile configurations (ids 50..55)
static const PatchCfg kPpCfgs[] = {
    {50, 4, 64, 128}, {51, 4, 64, 64}, {52, 2, 64, 128}, {53, 8, 32, 128}, {54, 8, 32, 64}, {55, 4, 32, 128},
    {56, 8, 32, 64}, {57, 4, 64, 64},
};
static inline const PatchCfg* find_pp_cfg(int id) {
    for (const PatchCfg& c : kPpCfgs)
        if (c.id == id) return &c;
    return nullptr;
}

template <typename T>
static inline int launch_pp_typed(int cfg, const ConvKArgs& k, hipStream_t s) {
    switch (cfg) {
        case 50: return launch_pp_cfg<T, 4, 64, 128, 3>(k, s);   // 256 px x 128, wave tile 64x64, 160 KiB
        case 51: return launch_pp_cfg<T, 4, 64, 64, 4>(k, s);    // 256 px x  64, wave tile 64x32, 4-deep weight ring
        case 52: return launch_pp_cfg<T, 2, 64, 128, 4>(k, s);   // 128 px x 128, wave tile 32x64
        case 53: return launch_pp_cfg<T, 8, 32, 128, 4>(k, s);   // 256 px x 128 for 32-wide tiles
        case 54: return launch_pp_cfg<T, 8, 32, 64, 4>(k, s);
        case 55: return launch_pp_cfg<T, 4, 32, 128, 4>(k, s);
        case 56: return launch_pp_cfg<T, 8, 32, 64, 3>(k, s);    // as 54 with the 3-deep ring
        case 57: return launch_pp_cfg<T, 4, 64, 64, 3>(k, s);    // as 51 with the 3-deep ring
    }
    set_error("conv: unknown ping-pong tile config %d", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
