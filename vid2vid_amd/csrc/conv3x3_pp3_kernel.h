// 3x3 stride-1 convolution, LDS-resident input patch -- SINGLE-PHASE software-pipelined schedule (gfx950).
//
// Why a third schedule.  scripts/pp2_ablate.py on the instrumented ping-pong kernel (profiles/r02_a4_pp2_ablate.txt, the
// 1024 -> 1024 layer, 256 px x 64 tile, 144 tap steps) decomposes its 106 us as: the 16 MFMAs of a step need 213 ns of
// matrix pipe per SIMD; the two barriers of a step cost ~146 ns; a LOAD phase (12 ds_read_b128 per wave: address VALU,
// 192 LDS cycles, the read latency, all of it between two barriers) costs ~190 ns and there are two per step --
// 600 ns per step, 36 % of the time in the matrix pipe.  Moving the LDS-DMA issue into the MMA phase (conv3x3_pp2_kernel)
// changed nothing: the DMA was never the problem, the exposed start-up / drain latency of the fragment reads is.
//
// Here every wave keeps TWO register sets of fragments.  Iteration j multiplies step j from set `cur` while it issues the
// ds_reads of step j+1 into set `nxt` and its share of the LDS-DMA of step j+D between those MFMAs; ONE barrier per step:
//
//      B_j | for m in MFMAs(j): mfma(cur) ; a few ds_read(j+1 -> nxt) / one DMA piece | vmcnt, lgkmcnt(0) | swap
//
// so the read latency hides behind the matrix pipe instead of behind a barrier.  Tile = TH x TW pixels x BN channels,
// 8 waves as 4(M) x 2(N), two waves per SIMD (both in the same code, the pipe alternates between them).
//
// Hazards (checked by scripts/pp_sched_sim.py pp3 for every (GP, LB, D) in use; phase j = between B_j and B_j+1):
//   RAW  the reads of step j+1 are issued after B_j; every wave retired ITS share of weight slice j+1 (and, at tap 7,
//        of the whole next patch) with the counted wait at the end of iteration j-1, before arriving at B_j.  At the end
//        of iteration j slice j+2 (issued in iteration j+2-D) must have landed; younger and allowed in flight: slices
//        j+3 .. j+D and the patch pieces of iterations j+2-D .. j  ->  vmcnt((D-2)*LB + sum_{u=0..D-2} np(tap-u)); taps
//        >= 9-D carry no patch pieces, so the wait of tap 7 retires the next chunk's patch before its first read (tap 8).
//   WAR  slice j+D refills the stage of slice j (ring of D stages), whose reads were drained (lgkmcnt(0)) before B_j;
//        the patch buffer of chunk c+1 held chunk c-1, last read in iteration 9c-2, and is refilled from iteration 9c on.
#pragma once
#include "conv3x3_pp2_kernel.h"

namespace v2v {

namespace pp3 {
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int np_at(int tap, int GP, int NPT) {
    const int ppt = (GP + NPT - 1) / NPT;
    return tap < NPT ? cmin((tap + 1) * ppt, GP) - cmin(tap * ppt, GP) : 0;
}
constexpr int pending_at(int tap, int GP, int LB, int D, int NT = 9) {
    int x = (D - 2) * LB;
    for (int u = 0; u < D - 1; ++u) x += np_at((tap - u + 2 * NT) % NT, GP, NT - D);
    return x;
}
// One barrier per BP steps (BP = 2, 3): pieces that may stay in flight behind the wait at the end of the LAST iteration i of a barrier
// period -- slices i+BP+2 .. i+D-BP+1 and the patch pieces of iterations i+2BP-D .. i (slice i+BP+1, the youngest that must have
// landed, was issued first in iteration i+2BP-D).  BP = 1 gives pending_at.
constexpr int pending_bp(int tap, int GP, int LB, int D, int NT, int BP, int NPT) {
    int x = (D - 2 * BP) * LB;
    for (int u = 0; u <= D - 2 * BP; ++u) x += np_at((tap - u + 2 * NT) % NT, GP, NPT);
    return x;
}
}  // namespace pp3

// taps 0 .. NT-1 of one channel chunk on alternating fragment register sets (first set P0)
template <int P0, typename F, int... I>
__device__ __forceinline__ void pp3_run_chunk(F& it, std::integer_sequence<int, I...>) {
    (it(std::integral_constant<int, I>{}, std::integral_constant<int, (P0 + I) & 1>{}), ...);
}

// ABL = 1: ablation instance (scripts/pp2_ablate.py): 1 / 2 hot operands, 4 no stores, 32 no fragment ds_reads,
// 64 no MFMAs, 128 no LDS-DMA in the main loop, 256 no vmcnt wait in the main loop, 512 return at once, 1024 no main loop.  Results are wrong when set.
// WGM x WGN waves: 4 x 2 (512 threads, wave tile BM/4 x BN/2) or 2 x 2 (256 threads, one wave per SIMD, wave tile BM/2 x BN/2).
// The 4-wave form with 128 px x 128 channels has 64 x 64 wave tiles: 16 MFMAs per 16 fragment reads instead of 8 per 12 --
// 64 KB of LDS reads per step and CU instead of 96 KB, below the ~200 B/clk the LDS delivers beside the 512 MFMA cycles.
// KS_ = 2 ("K pairs"): the 8 waves are WGM x WGN wave tiles x 2 K halves -- wave (tile, half h) multiplies K sub-steps 2h, 2h+1 of every
// step on a wave tile twice as wide (4 x 1 wave tiles of 64 x 64 in the 256 px x 64 workgroup tile instead of 4 x 2 of 64 x 32): 8 MFMAs
// per step as before, but 8 fragment reads instead of 12 (the B fragments of a 64-wide wave tile are shared by its two row tiles), i.e.
// 64 KB of LDS reads per step and CU instead of 96 KB -- the main loop of the 4 x 2 form is co-limited by the LDS read rate
// (12 ds_read_b128 x 8 waves = 96 KB per step against 512 MFMA cycles; profiles/r02_a5_pp3_ablate.txt: removing the reads saves as much
// as removing the MFMAs).  After the loop the two halves of a pair exchange one half of their accumulators through LDS (64 KB, once per
// launch): each wave ends up with the complete sums of a 64 x 32 tile, exactly the 4 x 2 layout the shared epilogue expects.
// KS_ = 4 ("K quads"): 2 x 1 wave tiles of 128 x 64, four K quarters -- one sub-step per wave and step, 8 MFMAs per 6 fragment reads
// (48 KB of LDS reads per step and CU), 128 accumulator registers per lane, a three-round reduce-scatter at the end.
// ONE = true (tiles 94 / 95): layers with a SINGLE 128-byte channel chunk (64 bf16 input channels: the ResnetBlocks of the fine
// scales).  Such a launch is thousands of tiles of 9 tap steps (~3 us) between a prologue and an epilogue that cost more than the
// steps; the second patch buffer is never used, so it is dropped: PATCH + ring = 72 / 80 KiB.  Bit-identical to tiles 80 / 83 and
// 0-5 % faster (profiles/r04_d6_*, r04_d8_*).  NOTE: the intent was two workgroups per CU, but LDS is not what decides that here --
// the kernel descriptor says 167 VGPRs (llvm-readelf --notes; rocprofv3's kernel-trace column prints half of that, 84), i.e. three
// waves per SIMD, so an 8-wave workgroup stays alone on its CU whatever its LDS footprint.  Whether a second resident workgroup
// would hide a tile's ~7 us of prologue / epilogue (mostly instruction issue: profiles/r04_d7_*) is therefore still untested for
// this kernel; on the ping-pong kernel a 128-register cap did give two workgroups per CU, spilled 39 registers and was slower
// (profiles/r04_d4_single_chunk_two_wg_per_cu.txt).
// KK = 7 (tiles 120 / 121, STAGED FOR ROUND 5 -- built, not yet run on a GPU): the same schedule for a KK x KK window, pad KK / 2 --
// the dense 7x7 stems on the pooled label encodings (108 -> 64 / 32 at 1024x512, 108 -> 128 / 64 at 512x256 and edge2face's
// 45 -> 128), which the 2048x1024 per-layer table puts first (1.47 ms on generic tiles that re-fetch their activations for each
// of the 49 taps: profiles/r04_f3_per_layer_roofline_hires.txt).  NT = KK^2 tap steps per channel chunk over a (TH + KK - 1) x
// (TW + KK - 1) patch; every pipeline constant that said 9 says NT, the tap offsets come from (tap / KK, tap % KK).
// FLAGS (round 5, experiments on the dominant 1024 -> 1024 pair; tiles 97-99):
//   bits 0-1 = BP - 1, the steps per barrier (1, 2 or 3).  BP = 2 ("B2"): ONE BARRIER PER TWO STEPS.  The barrier stays in front of the iterations that multiply from register set 0 (even
//     iterations; the parity is the register-set parity, which is the iteration parity because NT is odd and chunks alternate).
//     Between B_j and B_j+2 the waves may drift by up to two steps, so
//       RAW  both slices read in the pair (j+1 in iteration j, j+2 in iteration j+1) are retired by every wave BEFORE B_j: the counted
//            vmcnt wait sits at the end of the ODD iterations only and retires two slices;
//       WAR  an iteration may only refill a stage whose reads ended before the last barrier: iteration i issues slice i+D-1 into the
//            stage of slice i-1 (read in iteration i-2, in front of the barrier for either parity) -- one stage "older" than the
//            single-step schedule, i.e. D stages hold D-1 slices in flight (the prologue issues D-1 slices); the fragment reads are
//            drained (lgkmcnt(0)) at the end of the odd iterations, in front of the barrier that releases their stage;
//       patch: chunk c+1 streams into the buffer of chunk c-1 (last read in iteration NT*c-2) from iteration NT*c on; a barrier lies
//            between them for either parity of NT*c.  Its pieces (taps 0..NPT-1) are retired by the odd wait at most D-4+1
//            iterations later, in front of the barrier before iteration NT*c+NT-1, the first that reads the new patch.
//     Checked by scripts/pp_sched_sim.py pp3b2 (two drifting wave groups).
//     BP = 3: the same with three steps per barrier (iteration i issues slice i+D-2 into the stage of slice i-2, three slices retired
//     per barrier; NT = 9 makes iteration % 3 == tap % 3, so the barriers sit in front of taps 0, 3, 6 of every chunk).
//   bit 2: static priority -- the second-dispatched half of the waves runs at s_setprio 1 for the whole main loop
//     (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
template <typename T, int TH, int TW, int BN, int D, int ABL = 0, int WGM_ = 4, int WGN_ = 2, int KS_ = 1, bool ONE = false, int KK = 3, int FLAGS = 0>
__device__ __forceinline__ void conv3x3_pp3_body(const ConvKArgs& p_in) {
    constexpr int BP = 1 + (FLAGS & 3);                       // steps per barrier
    constexpr bool B2 = BP > 1;                               // (the name of the first variant: one barrier per two steps)
    static_assert(BP <= 3, "barrier period: a barrier must lie between the last read of a patch buffer (iteration NT*c-2) and its refill (NT*c)");
    static_assert(!B2 || (!ONE && KK == 3 && D >= 2 * BP + 1), "BP steps per barrier: D stages hold D-BP+1 slices in flight, BP of them retired per barrier");
    // grouped launch: which member and which tile this workgroup works on (the members have identical geometry, so the
    // tile count is known before the member is)
    int member = (int)blockIdx.z, lin_all;
    {
        const int ntot = p_in.m_tiles * p_in.n_tiles * p_in.splitk;
        if (p_in.grp_xcd && gridDim.z == 2 && (ntot & 7) == 0) lin_all = grouped_xcd_map(blockIdx.x, blockIdx.z, ntot, member);
        else                                                  lin_all = xcd_remap(blockIdx.x, ntot);
    }
    const ConvKArgs p = select_group(p_in, member);
    const int ab = ABL ? p.ablate : 0;
    if (ab & 512) return;                                     // ablation: the launch itself
    V2V_STAMP(p, 0);
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BM = TH * TW;
    constexpr int NT = KK * KK, PADK = KK / 2;                // tap steps per channel chunk; padding of the window
    constexpr int PW = TW + KK - 1, PR = (TH + KK - 1) * PW;
    constexpr int WGM = WGM_, WGN = WGN_, KS = KS_, NW = WGM * WGN * KS;
    constexpr int SS = 4 / KS;                                // K sub-steps (16 elements each) of a step that one wave multiplies
    constexpr int NG = (PR + 7) / 8;
    constexpr int GP = (NG + NW - 1) / NW;                    // patch pieces per wave per chunk
    constexpr int PATCH = GP * NW * 1024;
    constexpr int BST = BN * 128;
    constexpr int LB = BN / 8 / NW;                           // weight pieces per wave per slice
    constexpr int NSB = D;                                    // weight ring: slice j+D refills the stage of slice j
    constexpr int NPT = (FLAGS & 3) == 2 ? 11 - D : NT - D;   // taps 0..NPT-1 carry next-chunk patch pieces (BP = 3: retired in front of B_{NT*c+6})
    constexpr int PPT = (GP + NPT - 1) / NPT;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NMMA = SS * TM * TN;                        // MFMAs of one step (per wave)
    constexpr int NRD = SS * (TM + TN);                       // fragment reads of one step (per wave)
    static_assert(KS == 1 || (KS == 2 && TN == 2 && WGN == 1) || (KS == 4 && TM == 4 && WGN == 1),
                  "K pairs: 64-wide wave tiles that split into two 32-wide epilogue tiles; K quads: 128-row wave tiles that split into four 32-row ones");
    static_assert(TW % 32 == 0 && (TW & (TW - 1)) == 0, "a 32-row fragment must lie inside one tile row");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile");
    static_assert(BN % (8 * NW) == 0 && LB >= 1, "weight loader rounds");
    static_assert(D >= 3 && D <= 8, "weight slices in flight");
    static_assert((D - 2) * LB + (D - 1) * ((GP + NT - D - 1) / (NT - D)) <= 63, "vmcnt immediate range");
    constexpr int NPB = ONE ? 1 : 2;                          // patch buffers
    static_assert(!ONE || KS == 1, "single-chunk tiles: no accumulator exchange (it would need 64 KiB of scratch)");
    static_assert(NPB * PATCH + NSB * BST <= 160 * 1024 / (ONE ? 2 : 1), "LDS");
    static_assert(NPB * PATCH >= 20480 + (NW > 4 ? 20480 : 0), "epilogue scratch (statistics rows + a 4 KiB transposition block per wave; the finalize flag at 16 KiB) lives in the patch buffers");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bring = smem + NPB * PATCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wid % KS;                                  // K half of this wave (0 when KS == 1)
    const int wm = (wid / KS) / WGN, wn = (wid / KS) % WGN;
    const int cls = 0;

    const int tiles = p.m_tiles * p.n_tiles;
    const int S = p.splitk;
    int lin, slice, nt, mt, n_img, th, twi;
    patch_tile_index(p, lin_all, lin, slice, nt, mt, n_img, th, twi);
    const int oh0 = th * TH, ow0 = twi * TW;
    // fused norm: the launch tag of this channel tile's statistics granules (conv_epilogue), fetched here so that its round trip is
    // over long before the epilogue needs it
    unsigned fin_epoch = 0u;
    if constexpr (ABL == 0) {
        if (p.out_mode == V2V_OUT_NORM_ACT_NHWC)
            fin_epoch = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.fin_counter + V2V_FIN_TAG_WORD + nt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const int H = p.H, W = p.W, cs = p.cin_stride;
    const int ncc_all = cs * (int)sizeof(T) / 128;
    int ccb, ncc;
    patch_chunk_range(ncc_all, slice, S, ccb, ncc);
    const int nsteps = ncc * NT;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const char* const zp = p.zero_page;

    // ---------------- patch loader geometry (as conv3x3_pp_kernel) ----------------
    unsigned pp[GP];
    unsigned pok = 0;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        // first tap's input offset = -pad: PADK for the layer's own operator; 3x3 backward-data of a reflect-padded layer is the same
        // kernel with pad 2 over an output grid two pixels larger (round 6: v2v_conv_desc.pad 1 or 2 for the single-phase tiles)
        int ih = oh0 + pr + p.dh0[0], iw = ow0 + pc + p.dw0[0];
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
    auto issue_patch = [&](int k, int cc_local, char* buf) __attribute__((always_inline)) {
        const int cg = cc_local < ncc ? cc_local : ncc - 1;    // tail: a harmless reload keeps the DMA counts uniform
        const char* src = (((pok >> k) & 1u) && !(ab & 1)) ? p.in + pp[k] + (ccb + cg) * 128 : zp;
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
        wp[i] = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T) + (long long)ccb * NT * 128;
    }
    auto issue_w_piece = [&](int i, int step, int stage) __attribute__((always_inline)) {
        const int sg = step < nsteps ? step : nsteps - 1;      // tail duplicate into a free stage
        glds16(wp[i] + ((ab & 2) ? 0ll : (long long)sg * 128), bring + stage * BST + wid * 1024 + i * NW * 1024);
    };

    // ---------------- fragment addressing ----------------
    const int lr = lane & 31, hi = lane >> 5;
    int qb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m0 = wm * WM + i * 32;
        qb[i] = (m0 / TW) * PW + (m0 % TW) + lr;
    }
    int foff[SS];                                             // B fragments: 16-byte slot of sub-step wk * SS + s, swizzled by the weight row
#pragma unroll
    for (int s = 0; s < SS; ++s) foff[s] = (((wk * SS + s) * 2 + hi) ^ ((lr >> 1) & 7)) << 4;
    const int b_row_off = (wn * WN + lr) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Frag fa[2][SS][TM], fb[2][SS][TN];                       // two register sets: multiply one, fill the other

    // fragment read q of a step (0 .. NRD-1: the A reads i-major, then the B reads) into set `PARN`
    // aaddr[i]: byte address of patch row (qb[i] + tap offset) with its swizzle term ax[i]; pb: weight stage + row
    auto read_frag = [&](auto qc, auto parc, const char* (&arow)[TM], int (&ax)[TM], const char* pb) {
        constexpr int q = decltype(qc)::value;
        constexpr int PARN = decltype(parc)::value;
        if constexpr (q < SS * TM) {
            constexpr int i = q / SS, s = q % SS;
            fa[PARN][s][i] = *reinterpret_cast<const Frag*>(arow[i] + ((((wk * SS + s) * 2 + hi) ^ ax[i]) << 4));
        } else {
            constexpr int s = (q - SS * TM) / TN, j = (q - SS * TM) % TN;
            fb[PARN][s][j] = *reinterpret_cast<const Frag*>(pb + j * 32 * 128 + foff[s]);
        }
    };

    // ---------------- prologue: patch 0 and weight slices 0 .. D-1; step 0's fragments into set 0 ----------------
#pragma unroll
    for (int k = 0; k < GP; ++k) issue_patch(k, 0, smem);
    constexpr int DP = D - (BP - 1);                          // slices the prologue issues (BP > 1: iteration 0 issues slice D-BP+1 itself)
#pragma unroll
    for (int t = 0; t < DP; ++t)
#pragma unroll
        for (int i = 0; i < LB; ++i) issue_w_piece(i, t, t);
    if constexpr ((FLAGS & 4) != 0) {
        if (wid >= NW / 2) __builtin_amdgcn_s_setprio(1);
    }
    V2V_STAMP(p, 1);
    wait_vmcnt<(DP - 1) * LB>();                             // the patch and slice 0 have landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    V2V_STAMP(p, 2);
    {
        const char* arow[TM]; int ax[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { arow[i] = smem + qb[i] * 128; ax[i] = (qb[i] >> 1) & 7; }
        const char* const pb = bring + b_row_off;
        if (!(ab & 32))
            static_for<NRD>([&](auto qc) { read_frag(qc, std::integral_constant<int, 0>{}, arow, ax, pb); });
    }
    wait_vmcnt<(DP - 1 - BP) * LB>();                        // slices 1 .. BP: published by B_0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int step = 0, stage = 0, cc = 0;                          // step being multiplied, its weight stage, its chunk
    const char* pa = smem;                                    // patch buffer of chunk cc
    char* pn = smem + (ONE ? 0 : PATCH);                      // the other one (chunk cc+1 streams in; ONE: there is none)

    // iteration: multiply step `step` (tap TAP) from set PAR, fill set 1-PAR with step+1, issue the DMA of step+D
    auto iteration = [&](auto tc, auto pc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tc)::value;
        constexpr int PAR = decltype(pc)::value;
        constexpr int NTAP = (TAP + 1) % NT;                 // tap of the step whose fragments are read now
        constexpr int tq = (NTAP / KK) * PW + (NTAP % KK);
        constexpr int k0 = pp3::cmin(TAP * PPT, GP);
        constexpr int npz = ONE ? 0 : pp3::np_at(TAP, GP, NPT);   // ONE: no next chunk (host check), no next patch
        constexpr int NDMA = LB + npz;
        // issue slots: after MFMA m (m = 0 .. NMMA-2).  The reads go first (their latency then hides behind the remaining
        // MFMAs), RPS per slot; the DMA pieces follow, one per slot
        constexpr int RSLOTS = (NMMA * 5) / 8 > 0 ? (NMMA * 5) / 8 : 1;
        constexpr int RPS = (NRD + RSLOTS - 1) / RSLOTS;
        constexpr int RUSED = (NRD + RPS - 1) / RPS;          // slots that actually carry reads

        constexpr int PH = BP == 1 ? 0 : BP == 2 ? PAR : TAP % 3;      // position inside the barrier period (NT = 9: iteration % 3 == tap % 3)
        if constexpr (PH == 0)
            __builtin_amdgcn_s_barrier();                    // B_step: slice step+1 (and at tap 8 the next patch) is visible,
                                                             // the stage of slice `step` and its fragment reads are retired
                                                             // (B2: only in front of the even iterations, for two steps at once)
        const char* arow[TM]; int ax[TM];
        {
            const char* const pbuf = TAP == NT - 1 ? pn : pa;     // step+1 belongs to the next chunk at the last tap
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int qv = qb[i];
                asm volatile("" : "+v"(qv));               // opaque: no hoisting of 9 x TM address sets out of the chunk loop
                const int q = qv + tq;
                arow[i] = pbuf + q * 128;
                ax[i] = (q >> 1) & 7;
            }
        }
        const int nstage = stage + 1 == NSB ? 0 : stage + 1;
        const char* const pb = bring + nstage * BST + b_row_off;
        auto dma = [&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            if (ab & 128) return;
            if constexpr (d < LB) {
                if constexpr (B2) issue_w_piece(d, step + D - (BP - 1), stage >= BP - 1 ? stage - (BP - 1) : stage - (BP - 1) + NSB);   // slice step+D-BP+1 refills the stage of slice step-BP+1
                else              issue_w_piece(d, step + D, stage);                                   // slice step+D refills the stage of slice `step`
            } else                issue_patch(k0 + d - LB, cc + 1, pn);
        };
        auto reads_of_slot = [&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            if (ab & 32) return;
            static_for<RPS>([&](auto rc) {
                constexpr int q = m * RPS + decltype(rc)::value;
                if constexpr (q < NRD) read_frag(std::integral_constant<int, q>{}, std::integral_constant<int, 1 - PAR>{}, arow, ax, pb);
            });
        };
        static_for<NMMA>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int s = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
            if (!(ab & 64)) Mma<T>::run(fa[PAR][s][i], fb[PAR][s][j], acc[i][j]);
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
        // whatever found no slot (short MFMA sequences with many pieces)
        constexpr int DMA_IN_SLOTS = pp3::cmin(NDMA, NMMA - 1 - RUSED > 0 ? NMMA - 1 - RUSED : 0);
        static_for<NDMA - DMA_IN_SLOTS>([&](auto dc) { dma(std::integral_constant<int, DMA_IN_SLOTS + decltype(dc)::value>{}); });
        if constexpr (B2) {
            if constexpr (PH == BP - 1) {                    // in front of the barrier: slices step+2 .. step+BP+1 retired, all fragment reads drained
                wait_vmcnt<pp3::pending_bp(TAP, GP, LB, D, NT, BP, NPT)>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            if (!(ab & 256)) wait_vmcnt<(ONE ? (D - 2) * LB : pp3::pending_at(TAP, GP, LB, D, NT))>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // set 1-PAR is complete; the reads of slice step+1 are retired
        }
        ++step;
        stage = nstage;
        if constexpr (TAP == NT - 1 && !ONE) {
            ++cc;
            const char* t = pa; pa = pn; pn = const_cast<char*>(t);
        }
    };
    static_assert(NT % 2 == 1, "an odd number of taps per chunk: the register-set parity flips from chunk to chunk");
    auto chunk = [&](auto par0c) __attribute__((always_inline)) {
        constexpr int P0 = decltype(par0c)::value;
        if constexpr (NT == 9) {                             // the 3x3 kernels exactly as they shipped in round 4
            iteration(std::integral_constant<int, 0>{}, std::integral_constant<int, P0>{});
            iteration(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 - P0>{});
            iteration(std::integral_constant<int, 2>{}, std::integral_constant<int, P0>{});
            iteration(std::integral_constant<int, 3>{}, std::integral_constant<int, 1 - P0>{});
            iteration(std::integral_constant<int, 4>{}, std::integral_constant<int, P0>{});
            iteration(std::integral_constant<int, 5>{}, std::integral_constant<int, 1 - P0>{});
            iteration(std::integral_constant<int, 6>{}, std::integral_constant<int, P0>{});
            iteration(std::integral_constant<int, 7>{}, std::integral_constant<int, 1 - P0>{});
            iteration(std::integral_constant<int, 8>{}, std::integral_constant<int, P0>{});
        } else {
            pp3_run_chunk<P0>(iteration, std::make_integer_sequence<int, NT>{});     // taps 0 .. NT-1, register sets P0, 1 - P0, P0, ...
        }
    };
    // NT (odd) taps per chunk: the register-set parity flips from chunk to chunk
    int c = (ab & 1024) ? ncc : 0;                            // ablation 1024: prologue + epilogue only
    for (; c + 1 < ncc; c += 2) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
    }
    if (c < ncc) chunk(std::integral_constant<int, 0>{});

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tail duplicates must land before the LDS is reused
    if constexpr ((FLAGS & 4) != 0) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    V2V_STAMP(p, 3);

    // OUTPUT grid (= the input grid for the layers' own pad-1 operators; two pixels larger for the pad-2 backward-data form, round 6)
    const int OHo = p.OH, OWo = p.OW;
    const bool tile_full = oh0 + TH <= OHo && ow0 + TW <= OWo;    // every row of the tile is a pixel of the layer (uniform: conv_epilogue's fast paths)
    auto pix_of = [&](int row) __attribute__((always_inline)) -> int {        // TW is a power of two; N*OH*OW < 2^31 (host check)
        const int oh = oh0 + row / TW, ow = ow0 + (row & (TW - 1));
        if (oh >= OHo || ow >= OWo) return -1;
        return (n_img * OHo + oh) * OWo + ow;
    };
    if constexpr (KS == 1) {
        conv_epilogue<T, BM, BN, WGM, WGN, ABL == 0>(p, acc, smem, tid, wm, wn, false, cls, tiles, lin, slice, S, nt, mt, pix_of, tile_full, fin_epoch);
    } else if constexpr (KS == 4) {
        // K quads: the four waves of a 128-row wave tile each hold one K quarter of all of it.  Reduce-scatter in three rounds: in
        // round d wave h hands row tile (h + d) % 4 to wave (h + d) % 4 of its quad and collects its own row tile h from wave
        // (h - d) % 4 -- 64 KB per round through the patch buffers ([wave][j][reg][lane], lane-linear).  The sum order
        // own + (h-1) + (h-2) + (h-3) is fixed, so the result does not depend on timing.  Then the 8 x 1 layout of 32 x BN tiles.
        static_assert(2 * PATCH >= NW * TN * 16 * 64 * 4, "K quads: the accumulator exchange lives in the patch buffers");
        float* const xch = reinterpret_cast<float*>(smem);
        f32x16 acc4[1][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc4[0][j][r] = wk == 0 ? acc[0][j][r] : wk == 1 ? acc[1][j][r] : wk == 2 ? acc[2][j][r] : acc[3][j][r];
#pragma unroll
        for (int d = 1; d < 4; ++d) {
            const int to = (wk + d) & 3;                       // row tile (and quad member) that receives this round
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float give = to == 0 ? acc[0][j][r] : to == 1 ? acc[1][j][r] : to == 2 ? acc[2][j][r] : acc[3][j][r];
                    xch[((wid * TN + j) * 16 + r) * 64 + lane] = give;
                }
            __syncthreads();
            const int from = (wid & ~3) | ((wk - d) & 3);     // the quad member that wrote row tile wk this round
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc4[0][j][r] += xch[((from * TN + j) * 16 + r) * 64 + lane];
            __syncthreads();
        }
        conv_epilogue<T, BM, BN, WGM * 4, WGN, ABL == 0>(p, acc4, smem, tid, wm * 4 + wk, wn, false, cls, tiles, lin, slice, S, nt, mt, pix_of, tile_full, fin_epoch);
    } else {
        // K pairs: wave (tile, half h) keeps column tile j = h of its 64-wide wave tile and hands column tile 1 - h to its partner
        // (wave id ^ 1), which holds the other half of the K sum for it.  Same lane <-> element map on both sides (same MFMA
        // layout), so the exchange is lane-linear: [wave][i][reg][lane] floats, conflict-free 4-byte accesses.
        static_assert(2 * PATCH >= NW * TM * 16 * 64 * 4, "K pairs: the accumulator exchange lives in the patch buffers");
        float* const xch = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float give = wk == 0 ? acc[i][1][r] : acc[i][0][r];
                xch[((wid * TM + i) * 16 + r) * 64 + lane] = give;
            }
        __syncthreads();
        f32x16 acc2[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float own = wk == 0 ? acc[i][0][r] : acc[i][1][r];
                acc2[i][0][r] = own + xch[(((wid ^ 1) * TM + i) * 16 + r) * 64 + lane];
            }
        __syncthreads();                                      // the exchange area becomes the epilogue's scratch
        // 4 x 2 layout of 64 x 32 tiles: this wave's tile is (wm, 2 * wn + wk)
        conv_epilogue<T, BM, BN, WGM, 2 * WGN, ABL == 0>(p, acc2, smem, tid, wm, 2 * wn + wk, false, cls, tiles, lin, slice, S, nt, mt, pix_of, tile_full, fin_epoch);
    }
}

template <typename T, int TH, int TW, int BN, int D, int ABL = 0, int WGM_ = 4, int WGN_ = 2, int KS_ = 1, bool ONE = false, int FLAGS = 0>
__global__ __launch_bounds__(WGM_ * WGN_ * KS_ * 64) void conv3x3_pp3_kernel(const ConvKArgs p_in) {
    conv3x3_pp3_body<T, TH, TW, BN, D, ABL, WGM_, WGN_, KS_, ONE, 3, FLAGS>(p_in);
}

// 7x7 window: its own entry point, pinned to two waves per SIMD.  The lambdas of the body are always_inline: with 98 unrolled tap
// steps the inliner otherwise leaves some of them as calls, their by-reference captures pin the argument block (a 700-byte struct)
// and the pipeline state to scratch memory -- 85 registers + 1.8 KB of scratch per lane for the 128-channel tile, and a back end that
// cannot honour the epilogue's SGPR pins ("illegal VGPR to SGPR copy") -- -Rpass-analysis=kernel-resource-usage before / after
template <typename T, int TH, int TW, int BN, int D, int WGM_ = 4, int WGN_ = 2>
__global__ __launch_bounds__(WGM_ * WGN_ * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv7x7_pp3_kernel(const ConvKArgs p_in) {
    conv3x3_pp3_body<T, TH, TW, BN, D, 0, WGM_, WGN_, 1, false, 7>(p_in);
}

template <typename T, int TH, int TW, int BN, int D, int ABL = 0, int WGM = 4, int WGN = 2, int KS = 1, bool ONE = false, int KK = 3, int FLAGS = 0>
static int launch_pp3_cfg(const ConvKArgs& k, int groups, hipStream_t s) {
    constexpr int NW = WGM * WGN * KS;
    constexpr int GP = (((TH + KK - 1) * (TW + KK - 1) + 7) / 8 + NW - 1) / NW;
    const size_t lds = (size_t)(ONE ? 1 : 2) * GP * NW * 1024 + (size_t)D * BN * 128;
    void (*kern)(const ConvKArgs);
    if constexpr (KK == 7) kern = conv7x7_pp3_kernel<T, TH, TW, BN, D, WGM, WGN>;
    else                   kern = conv3x3_pp3_kernel<T, TH, TW, BN, D, ABL, WGM, WGN, KS, ONE, FLAGS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles * k.splitk), 1u, (unsigned)groups);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, k);
    return check_launch();
}

// single-phase tile configurations (ids 80..93)
static const PatchCfg kPp3Cfgs[] = {
    {80, 8, 32, 64}, {81, 8, 32, 128}, {82, 8, 32, 64}, {83, 4, 64, 64}, {84, 4, 32, 128}, {85, 4, 64, 128},
    {86, 4, 32, 128}, {87, 2, 64, 128},     // 4 waves (2 x 2), 64 x 64 wave tiles
    {88, 8, 32, 128}, {89, 8, 32, 64},     // ablation instances of 81 / 80
    {90, 8, 32, 64}, {91, 4, 64, 64},      // K pairs: 4 x 1 wave tiles of 64 x 64, two K halves (see the kernel comment)
    {92, 8, 32, 64}, {93, 4, 64, 64},      // K quads: 2 x 1 wave tiles of 128 x 64, four K quarters: 6 reads per 8 MFMAs
    {94, 8, 32, 64}, {95, 4, 64, 64},      // single-chunk layers: one patch buffer, 72 / 80 KiB
    // (round 6: the round-5 experiment tiles 97-99 / 130-132 -- one barrier per two / three steps, deeper weight rings, static wave
    //  priority on tile 90's geometry, all bit-identical to tile 90 and none faster, DESIGN 3.1 -- are no longer instantiated; the
    //  FLAGS parameter of the body that built them stays)
    {140, 8, 32, 64}, {141, 8, 32, 64}, {143, 8, 32, 64},    // conv3x3_one_kernel.h: persistent, weights-resident single-chunk tile (geometry only; launched by launch_one_typed)
    {96, 4, 32, 64},                       // single-chunk layers, FOUR waves (2 x 2, 64 x 32 wave tiles), 28 + 24 = 52 KiB, 167 + 32 registers: the
                                           // tile that really puts two workgroups on a CU (staged for round 5; 94 / 95 never did: DESIGN 3.6 item 15)
    {120, 4, 32, 64}, {121, 4, 32, 128},   // 7x7 window (staged for round 5): 10 x 38 pixel patch, 49 tap steps per channel chunk
};
static inline const PatchCfg* find_pp3_cfg(int id) {
    for (const PatchCfg& c : kPp3Cfgs)
        if (c.id == id) return &c;
    return nullptr;
}

template <typename T>
static inline int launch_pp3_typed(int cfg, const ConvKArgs& k, int groups, hipStream_t s) {
    switch (cfg) {
        case 80: return launch_pp3_cfg<T, 8, 32, 64, 4>(k, groups, s);    // 256 px x  64, wave tile 64x32, 4 slices in flight, 128 KiB
        case 81: return launch_pp3_cfg<T, 8, 32, 128, 4>(k, groups, s);   // 256 px x 128, wave tile 64x64, 160 KiB
        case 82: return launch_pp3_cfg<T, 8, 32, 64, 5>(k, groups, s);    // as 80, 5 slices in flight, 136 KiB
        case 83: return launch_pp3_cfg<T, 4, 64, 64, 4>(k, groups, s);    // 256 px x  64 for 64-wide tiles, 144 KiB
        case 84: return launch_pp3_cfg<T, 4, 32, 128, 4>(k, groups, s);   // 128 px x 128, wave tile 32x64
        case 85: return launch_pp3_cfg<T, 4, 64, 128, 3>(k, groups, s);   // 256 px x 128 for 64-wide tiles, 3 slices, 160 KiB
        case 86: return launch_pp3_cfg<T, 4, 32, 128, 4, 0, 2, 2>(k, groups, s);   // 128 px x 128, FOUR waves, wave tile 64x64, 120 KiB
        case 87: return launch_pp3_cfg<T, 2, 64, 128, 4, 0, 2, 2>(k, groups, s);   // same for 64-wide tile rows
        case 90: return launch_pp3_cfg<T, 8, 32, 64, 5, 0, 4, 1, 2>(k, groups, s);   // as 82 (256 px x 64, 5 slices), K pairs: 8 reads per 8 MFMAs
        case 91: return launch_pp3_cfg<T, 4, 64, 64, 4, 0, 4, 1, 2>(k, groups, s);   // as 83 for 64-wide tile rows
        case 92: return launch_pp3_cfg<T, 8, 32, 64, 5, 0, 2, 1, 4>(k, groups, s);   // as 82, K quads
        case 93: return launch_pp3_cfg<T, 4, 64, 64, 4, 0, 2, 1, 4>(k, groups, s);   // as 83, K quads
        case 120: case 121:                // 7x7 window: bf16 only for now (the fp32 instantiations double an 8-minute translation unit)
            if constexpr (std::is_same<T, bf16_t>::value) {
                if (cfg == 120) return launch_pp3_cfg<T, 4, 32, 64, 4, 0, 4, 2, 1, false, 7>(k, 1, s);    // 128 px x  64, 2 x 48 + 32 = 128 KiB, 109 registers
                return launch_pp3_cfg<T, 4, 32, 128, 3, 0, 4, 2, 1, false, 7>(k, 1, s);                   // 128 px x 128, 2 x 48 + 48 = 144 KiB, 163 registers
            }
            break;
        case 96:                           // single-chunk, four waves: two co-resident workgroups per CU by registers (2 waves / SIMD) and LDS (52 KiB)
            if constexpr (std::is_same<T, bf16_t>::value) return launch_pp3_cfg<T, 4, 32, 64, 3, 0, 2, 2, 1, true>(k, 1, s);
            break;
        case 94: case 95:                  // single-chunk tiles: bf16 only (64 input channels = one 128-byte chunk), single launches only
            if constexpr (std::is_same<T, bf16_t>::value) {
                if (cfg == 94) return launch_pp3_cfg<T, 8, 32, 64, 3, 0, 4, 2, 1, true>(k, 1, s);   // as 80 with 3 slices: 48 + 24 KiB
                return launch_pp3_cfg<T, 4, 64, 64, 3, 0, 4, 2, 1, true>(k, 1, s);                  // as 83 with 3 slices: 56 + 24 KiB
            }
            break;
        case 88: return launch_pp3_cfg<T, 8, 32, 128, 4, 1>(k, groups, s);
        case 89: return launch_pp3_cfg<T, 8, 32, 64, 4, 1>(k, groups, s);
    }
    set_error("conv: unknown single-phase tile config %d", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
