// 3x3 stride-1 convolution with an LDS-RESIDENT INPUT PATCH (gfx950), NHWC activations.
//
// Why a second kernel (measured, profiles/r01_v4_ablate_*): the implicit-GEMM kernel (conv_igemm_kernel.h) re-fetches
// its BM x 128 B activation tile for every one of the 9 taps, and at batch 1 it is bound by the CU's vector-memory
// path (LDS-DMA runs at <= 46 B/clk/CU even from L1-hot lines; 24 KiB per 258 MFMA-cycles of work for a 128x64
// tile), not by HBM, L2 misses or the matrix pipe.  The ResnetBlock convolutions (networks.py:571-587; 66 % + 8 %
// of the frame's FLOPs) are 3x3 / stride 1 / pad 1, so here
//   * the K loop runs channel-chunk OUTER, tap INNER (weights are packed in that order, v2v_conv_pack_weights
//     korder = 1): for one 128-byte channel chunk the (TH+2) x (TW+2) pixel patch of the TH x TW output tile is
//     brought into LDS ONCE (double buffered: chunk c+1 streams in during the first 5 tap steps of chunk c) and the
//     9 taps read it through shifted fragment addresses -- activation traffic per tap step drops ~8x and only the
//     BN x 128 B weight slice is fetched per step;
//   * the weight ring is NSB deep (NSB-1 slices in flight); one counted s_waitcnt vmcnt + ONE raw s_barrier per
//     tap step.  The 9 tap steps are unrolled, so every vmcnt immediate is a compile-time constant; the tail issues
//     harmless duplicate DMAs instead of changing the counts;
//   * patch rows are 128 B with the same XOR swizzle ((row>>1)&7 on the 16-byte slot) as the GEMM kernel: the
//     32 consecutive patch rows a 32x32 MFMA fragment reads are conflict-free for ANY start row (each
//     ds_read_b128 lane group {b..b+3, b+12..b+15, b+20..b+27} holds 8 even and 8 odd rows whose (row>>1)&7 are
//     all distinct), which is what lets a tap be a plain row offset;
//   * zero / reflection padding live in the patch loader (zero page / mirrored source address).
// Epilogue (split-K over channel chunks, bias, statistics, in-kernel norm finalize, activation) is shared:
// conv_epilogue() in conv_igemm_kernel.h.
#pragma once
#include "conv_igemm_kernel.h"
#include <type_traits>

namespace v2v {

namespace patch {
constexpr int NPT = 5;   // tap steps 0..NPT-1 of a channel chunk carry the NEXT chunk's patch pieces
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
// patch pieces (1 KiB LDS-DMA each) a wave issues at tap step `t` (GP pieces per wave per patch)
constexpr int np_at(int t, int GP) {
    const int ppt = (GP + NPT - 1) / NPT;
    return t < NPT ? cmax(0, cmin(ppt, GP - t * ppt)) : 0;
}
// LDS-DMAs issued after the weight slice of step j was issued (at step j-DB) and before step j's wait: the steps
// j-DB+1 .. j-1, each [patch pieces][LB weight pieces]
constexpr int pending_at(int tap, int GP, int LB, int DB) {
    int x = 0;
    for (int u = 1; u < DB; ++u) x += np_at((tap - u + 18) % 9, GP) + LB;
    return x;
}
}  // namespace patch

template <typename T, int TH, int TW, int BN, int WGM, int WGN, int NSB, int NL, int NPF>
__global__ __launch_bounds__((WGM * WGN + NL + NPF) * 64) void conv3x3_patch_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BM = TH * TW;
    constexpr int PW = TW + 2, PR = (TH + 2) * PW;            // patch width / rows (one row = one pixel x 128 B)
    constexpr int NWC = WGM * WGN;                            // compute (MFMA) waves
    // NL = 0: every wave also loads.  NL > 0: NL dedicated LOADER waves issue every LDS-DMA and own the vmcnt waits;
    // the compute waves only meet them at the per-step barrier.  (Measured: an LDS-DMA costs the issuing wave
    // ~60-180 cycles of issue time; in front of a wave's ds_read / MFMA work that is the step's critical path.)
    constexpr int NW = NL > 0 ? NL : NWC;                     // loading waves
    constexpr int NG = (PR + 7) / 8;                          // 8-row DMA pieces that hold real rows
    constexpr int GP = (NG + NW - 1) / NW;                    // pieces per loading wave per patch (dummy pieces pad the tail)
    constexpr int PATCH = GP * NW * 1024;                     // bytes per patch buffer
    constexpr int BST = BN * 128;                             // bytes per weight stage
    constexpr int LB = BN / 8 / NW;                           // weight pieces per wave per stage
    constexpr int DB = NSB - 1;                               // weight slices in flight
    constexpr int PPT = (GP + patch::NPT - 1) / patch::NPT;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TW % 32 == 0 && (TW & (TW - 1)) == 0, "a 32-row fragment must lie inside one tile row");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile");
    static_assert(BN % (8 * NW) == 0 && LB >= 1, "weight loader rounds");
    static_assert(DB >= 2 && DB <= 9 - patch::NPT, "patch pieces must be older than the weight slice that gates the next chunk");
    static_assert(2 * PATCH + NSB * BST + 256 <= 160 * 1024, "LDS");
    static_assert(2 * PATCH >= 32768, "epilogue scratch lives in the patch buffers");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bring = smem + 2 * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int cls = 0;
    const bool is_loader = NL == 0 ? wid < NWC : (wid >= NWC && wid < NWC + NL);
    const bool is_compute = wid < NWC;
    const int li = NL == 0 ? wid : (wid >= NWC ? wid - NWC : 0);     // loader index 0..NW-1
    // NPF = 1: one more wave that only touches the weight lines p.pf_dist steps ahead (4-byte LDS-DMA into a dummy
    // line; it never waits), so that the loaders' weight fetches hit L2: at batch 1 every weight line is a cold HBM
    // miss (~2.4k cycles under load) and the ring holds only NSB-1 slices
    const bool is_pf = NPF > 0 && wid >= NWC + NL;

    const int tiles = p.m_tiles * p.n_tiles;
    const int S = p.splitk;
    const int lin_all = xcd_remap(blockIdx.x, tiles * S);
    int lin, slice, nt, mt, n_img, th, twi;
    patch_tile_index(p, lin_all, lin, slice, nt, mt, n_img, th, twi);
    const int oh0 = th * TH, ow0 = twi * TW;

    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc_all = cs * (int)sizeof(T) / 128;            // channel chunks (host guarantees cs*sizeof % 128 == 0)
    int ccb, ncc;
    patch_chunk_range(ncc_all, slice, S, ccb, ncc);
    const int nsteps = ncc * 9;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const char* const zp = p.zero_page;

    // ---------------- patch loader geometry ----------------
    // piece g = k*NW + wid covers patch rows 8g..8g+7; lane l writes row 8g + (l>>3), physical 16-byte slot l&7,
    // which must hold the LOGICAL slot (l&7) ^ ((row>>1)&7): that is the slot this lane fetches.
    unsigned pp[GP];            // byte offset of this lane's slot from p.in (host guarantees the input is < 4 GiB)
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + li) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        int ih = oh0 + pr - 1, iw = ow0 + pc - 1;
        bool ok = q < PR;
        int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
        int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
        ih = reflect ? rh : ih;
        iw = reflect ? rw : iw;
        ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;   // zero padding, tile overhang
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        pp[k] = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
        pok |= (ok ? 1u : 0u) << k;
    }
    auto issue_patch = [&](int k, int cc_local, char* buf) {
        int cg = cc_local < ncc ? cc_local : ncc - 1;          // tail: a harmless reload keeps the DMA counts uniform
        const char* src = (((pok >> k) & 1u) && !(p.ablate & 1)) ? p.in + pp[k] + (ccb + cg) * 128 : zp;
        glds16(src, buf + (k * NW + li) * 1024);
    };

    // ---------------- weight loader geometry (as conv_igemm_kernel) ----------------
    const int lrow = li * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);     // (NW*8*i >> 1) & 7 == 0 for NW = 2, 4, 8
    const char* wp[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        long long r = (long long)nt * BN + lrow + NW * 8 * i;
        r = r < p.cout_p ? r : p.cout_p - 1;
        wp[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T) + (long long)ccb * 9 * 128;
    }
    auto issue_w = [&](int step, int stage) {
        const int sg = step < nsteps ? step : nsteps - 1;      // tail duplicate (its stage is free, see below)
        char* dst = bring + stage * BST + li * 1024;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            glds16(wp[i] + ((p.ablate & 2) ? 0ll : (long long)sg * 128), dst + i * NW * 1024);
    };

    // prefetch wave: lane l -> weight row nt*BN + l (+64 ...)
    constexpr int PFR = (BN + 63) / 64;
    const char* wpf[PFR];
#pragma unroll
    for (int i = 0; i < PFR; ++i) {
        long long r = (long long)nt * BN + lane + 64 * i;
        r = r < p.cout_p ? r : p.cout_p - 1;
        wpf[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0]) * (long long)sizeof(T) + (long long)ccb * 9 * 128;
    }
    const bool pf_on = is_pf && (mt & p.pf_mask) == 0 && p.pf_dist > 0;
    auto touch_w = [&](int step) {
        const int sg = step < nsteps ? step : nsteps - 1;
#pragma unroll
        for (int i = 0; i < PFR; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wpf[i] + (long long)sg * 128),
                                             (__attribute__((address_space(3))) void*)(bring + NSB * BST), 4, 0, 0);
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    int qb[TM];                                               // patch row of this lane's A row at tap (0,0)
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

    // ---------------- prologue: patch 0, weight slices 0..DB-1 ----------------
    if (is_loader) {
#pragma unroll
        for (int k = 0; k < GP; ++k) issue_patch(k, 0, smem);
#pragma unroll
        for (int t = 0; t < DB; ++t) issue_w(t, t);
    }
    if (pf_on)
        for (int t = DB; t < DB + p.pf_dist && t < nsteps; ++t) touch_w(t);

    // ---------------- main loop: channel chunk outer, 9 unrolled tap steps inner ----------------
    int step = 0, stage = 0, wstage = DB;                    // wstage = (step + DB) % NSB
    for (int cc = 0; cc < ncc; ++cc) {
        const char* const pa = smem + (cc & 1) * PATCH;
        char* const pnext = smem + ((cc + 1) & 1) * PATCH;
        auto tap_step = [&](auto tc) {
            constexpr int tap = decltype(tc)::value;
            // weight slice `step` (and every older DMA: the whole patch of this chunk) must have landed
            if (is_loader) wait_vmcnt<patch::pending_at(tap, GP, LB, DB)>();
            __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is done reading stage (step-1)%NSB
                                            // and, at tap 0, the patch buffer of chunk cc-1: both are refilled now
            if (is_loader) {
                if constexpr (tap < patch::NPT) {
                    constexpr int k0 = tap * PPT, k1 = patch::cmin((tap + 1) * PPT, GP);
#pragma unroll
                    for (int k = k0; k < k1; ++k) issue_patch(k, cc + 1, pnext);
                }
                issue_w(step + DB, wstage);
            }
            if (is_compute && !(p.ablate & 16)) {
                const char* const pb = bring + stage * BST + b_row_off;
                constexpr int tq = (tap / 3) * PW + (tap % 3);
                int abase[TM], ax[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int q = qb[i] + tq;
                    abase[i] = q * 128;
                    ax[i] = (q >> 1) & 7;
                }
                // every fragment of the step is requested up front (counted lgkmcnt ladders then let the first MFMAs
                // start while the rest stream in); per-k-substep reads with lgkmcnt(0) in between expose the LDS
                // latency four times per step when a SIMD hosts a single wave
                Frag fa[4][TM], fb[4][TN];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[s][i] = *reinterpret_cast<const Frag*>(pa + abase[i] + (((s * 2 + hi) ^ ax[i]) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[s][j] = *reinterpret_cast<const Frag*>(pb + j * 32 * 128 + foff[s]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
            }
            if (pf_on) touch_w(step + DB + p.pf_dist);
            ++step;
            stage = stage + 1 == NSB ? 0 : stage + 1;
            wstage = wstage + 1 == NSB ? 0 : wstage + 1;
        };
        tap_step(std::integral_constant<int, 0>{}); tap_step(std::integral_constant<int, 1>{});
        tap_step(std::integral_constant<int, 2>{}); tap_step(std::integral_constant<int, 3>{});
        tap_step(std::integral_constant<int, 4>{}); tap_step(std::integral_constant<int, 5>{});
        tap_step(std::integral_constant<int, 6>{}); tap_step(std::integral_constant<int, 7>{});
        tap_step(std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tail duplicates: nothing may land after the LDS is reused
    __syncthreads();

    conv_epilogue<T, BM, BN, WGM, WGN>(p, acc, smem, tid, wm, wn, wid >= NWC, cls, tiles, lin, slice, S, nt, mt,
        [&](int row) -> int {                  // TW is a power of two; N*OH*OW < 2^31 (host check)
            const int oh = oh0 + row / TW, ow = ow0 + (row & (TW - 1));
            if (oh >= H || ow >= W) return -1;
            return (n_img * H + oh) * W + ow;
        }, oh0 + TH <= H && ow0 + TW <= W);
}

template <typename T, int TH, int TW, int BN, int WGM, int WGN, int NSB, int NL, int NPF>
static int launch_patch_cfg(const ConvKArgs& k, hipStream_t s) {
    constexpr int NW = NL > 0 ? NL : WGM * WGN;
    constexpr int GP = (((TH + 2) * (TW + 2) + 7) / 8 + NW - 1) / NW;
    const size_t lds = (size_t)2 * GP * NW * 1024 + (size_t)NSB * BN * 128 + 256;
    auto kern = conv3x3_patch_kernel<T, TH, TW, BN, WGM, WGN, NSB, NL, NPF>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles * k.splitk), 1u);
    hipLaunchKernelGGL(kern, grid, dim3((WGM * WGN + NL + NPF) * 64), lds, s, k);
    return check_launch();
}

// Patch-kernel tile configurations (ids >= 32; TH x TW output pixels x BN output channels)
struct PatchCfg { int id, TH, TW, BN; };
static const PatchCfg kPatchCfgs[] = {
    {32, 2, 64, 64}, {33, 4, 64, 64}, {34, 2, 64, 128}, {35, 4, 32, 64}, {36, 8, 32, 64}, {37, 4, 32, 128},
    // the same tiles with 2 dedicated loader waves
    {40, 2, 64, 64}, {41, 4, 64, 64}, {42, 2, 64, 128}, {43, 4, 32, 64}, {44, 8, 32, 64}, {45, 4, 32, 128},
    // 8 compute waves (2 per SIMD) + 2 loader waves (46, 47); 256 x 128 tile (48)
    {46, 4, 64, 64}, {47, 2, 64, 128}, {48, 4, 64, 128},
};
static inline const PatchCfg* find_patch_cfg(int id) {
    for (const PatchCfg& c : kPatchCfgs)
        if (c.id == id) return &c;
    return nullptr;
}

template <typename T>
static inline int launch_patch_typed(int cfg, const ConvKArgs& k, hipStream_t s) {
    switch (cfg) {
        case 32: return k.pf_dist > 0 ? launch_patch_cfg<T, 2, 64, 64, 2, 2, 5, 0, 1>(k, s) : launch_patch_cfg<T, 2, 64, 64, 2, 2, 5, 0, 0>(k, s);    // 128 px x  64, wave tile 64x32, 114 KiB
        case 33: return k.pf_dist > 0 ? launch_patch_cfg<T, 4, 64, 64, 4, 1, 5, 0, 1>(k, s) : launch_patch_cfg<T, 4, 64, 64, 4, 1, 5, 0, 0>(k, s);    // 256 px x  64, wave tile 64x64, 144 KiB
        case 34: return k.pf_dist > 0 ? launch_patch_cfg<T, 2, 64, 128, 2, 2, 5, 0, 1>(k, s) : launch_patch_cfg<T, 2, 64, 128, 2, 2, 5, 0, 0>(k, s);   // 128 px x 128, wave tile 64x64, 152 KiB
        case 35: return k.pf_dist > 0 ? launch_patch_cfg<T, 4, 32, 64, 2, 2, 5, 0, 1>(k, s) : launch_patch_cfg<T, 4, 32, 64, 2, 2, 5, 0, 0>(k, s);    // 128 px x  64 for W % 64 != 0
        case 36: return k.pf_dist > 0 ? launch_patch_cfg<T, 8, 32, 64, 4, 1, 5, 0, 1>(k, s) : launch_patch_cfg<T, 8, 32, 64, 4, 1, 5, 0, 0>(k, s);    // 256 px x  64
        case 37: return k.pf_dist > 0 ? launch_patch_cfg<T, 4, 32, 128, 2, 2, 5, 0, 1>(k, s) : launch_patch_cfg<T, 4, 32, 128, 2, 2, 5, 0, 0>(k, s);   // 128 px x 128
        case 40: return launch_patch_cfg<T, 2, 64, 64, 2, 2, 5, 2, 0>(k, s);
        case 41: return launch_patch_cfg<T, 4, 64, 64, 4, 1, 5, 2, 0>(k, s);
        case 42: return launch_patch_cfg<T, 2, 64, 128, 2, 2, 5, 2, 0>(k, s);
        case 43: return launch_patch_cfg<T, 4, 32, 64, 2, 2, 5, 2, 0>(k, s);
        case 44: return launch_patch_cfg<T, 8, 32, 64, 4, 1, 5, 2, 0>(k, s);
        case 45: return launch_patch_cfg<T, 4, 32, 128, 2, 2, 5, 2, 0>(k, s);
        case 46: return launch_patch_cfg<T, 4, 64, 64, 4, 2, 5, 2, 0>(k, s);    // 256 px x  64, 8 compute waves (64x32)
        case 47: return launch_patch_cfg<T, 2, 64, 128, 2, 4, 5, 2, 0>(k, s);   // 128 px x 128, 8 compute waves (64x32)
        case 48: return launch_patch_cfg<T, 4, 64, 128, 4, 1, 3, 2, 0>(k, s);   // 256 px x 128, 4 compute waves (64x128), 3-deep ring
    }
    set_error("conv: unknown patch tile config %d", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
