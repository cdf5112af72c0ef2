// 3x3 / stride 1 / pad 1 convolution of layers with ONE 128-byte channel chunk (64 bf16 input channels) and at most 64 output
// channels -- the ResnetBlocks of the fine scales (models/networks.py:554-593 at ngf_s = 64: 12 layers of 64 -> 64 at 1024x512 in
// the 2048x1024 frame, the foreground tower's 64 -> 64 at 512x256) -- as a PERSISTENT, WEIGHTS-RESIDENT kernel (gfx950).
//
// Why (profiles/r04_d7_fine_scale_conv_ablate.txt).  Such a layer is thousands of tiles of 9 tap steps (~3 us of
// matrix work) each; on the single-phase tiles 94 / 95 every tile is its own workgroup: launch, index arithmetic, 48 KB of patch AND
// the whole 72 KB weight matrix through LDS-DMA, a ring of barriers, the epilogue, exit -- 9.7 of a tile's 14.4 us are outside the
// main loop, the workgroup is alone on its CU (167 VGPRs x 8 waves), so every latency in that chain is exposed.
//
// Here ONE workgroup per CU walks its tiles:
//   * the 9 x [64 rows][128 B] weight slices are loaded ONCE (72 KB of LDS, the ring-stage layout of conv3x3_pp3_kernel) -- 60 % of
//     the bytes a tile used to pull through the LDS-DMA path are gone, and with them every weight-ring hazard: the main loop has NO
//     barrier, NO vmcnt wait and NO DMA, only ds_reads (two fragment register sets, step j+1 read under the MFMAs of step j) and MFMAs;
//   * one 48 KB patch buffer.  Per tile: [patch landed] barrier | 9 steps | barrier [patch free] | issue the NEXT tile's patch |
//     epilogue of this tile (its own 34 KB of LDS scratch) -- the next patch travels while the epilogue stores, so its latency is
//     hidden without a second buffer (72 + 48 + 34 = 154 KB of LDS);
//   * ONE output mode per instantiation (raw fp32 NHWC + a statistics row, full tiles): the shared conv_epilogue dispatches on the
//     output mode, the activation, ragged rows and channels, split-K and the in-kernel finalize at run time -- ~900 vector
//     instructions per wave and a kernel at the SGPR ceiling; here the tail is the ~150 instructions this mode needs;
//   * interior tiles (no patch pixel outside the image) take their source offsets from a per-kernel table: one add per piece
//     instead of the reflect / clamp arithmetic;
//   * tile -> workgroup mapping: workgroup b takes virtual block ids b, b + G, b + 2G, ... (G = grid size, a multiple of 8), each
//     through xcd_remap: every XCD keeps a contiguous range of tiles (halo rows shared through its L2) as with one tile per workgroup.
// Same MFMA order per accumulator as tiles 80 / 94 (taps 0..8, K sub-steps 0..3), the same epilogue arithmetic in the same order as
// conv_epilogue's fast path: bit-identical results (raw output and statistics rows).
#pragma once
#include "conv3x3_pp3_kernel.h"
#include "conv3x3_t2_kernel.h"

namespace v2v {

// Start-up stagger (experiment, V2V_ONE_STAGGER=<units>[,<groups>]; units of ~0.5 us, 0 = off): persistent workgroups start together,
// do identical work and therefore stay in lockstep -- every CU loads, then every CU computes, then every CU stores, and the HBM and the
// matrix pipes are never busy at the same time.  Group g = (blockIdx.x >> 3) % groups waits g x units once, after its first loads are
// in flight (bits 16.. of ConvKArgs::ablate carry units | groups << 8; the host packs them in launch_one_typed).
__device__ __forceinline__ void one_stagger(const ConvKArgs& p) {
    const int units = (p.ablate >> 16) & 0xff, groups = (p.ablate >> 24) & 0x7f;
    if (units == 0 || groups < 2) return;
    const int g = ((int)blockIdx.x >> 3) % groups;
    for (int i = 0; i < g * units; ++i) __builtin_amdgcn_s_sleep(16);
}

// Statistics of a persistent workgroup (round 5, third step): ONE (sum, sum^2) row per WORKGROUP, not per tile.  Every lane keeps its
// channel's two running sums in registers across all the tiles its workgroup walks; after the last tile the 2 x WGM wave partials are
// combined through `red` and row blockIdx.x of p.stats is written: a 2048-tile layer leaves 256 rows (v2v_conv_stats_rows reports the
// grid size), so the layer's finalize needs no bn_partial_reduce launch -- and, with p.fin_counter, no launch at all: the LAST
// workgroup to publish its row reduces the <= 256 rows and writes the scale / shift record (the hand-off and the arithmetic of
// conv_epilogue's single-level in-kernel finalize, conv_igemm_kernel.h: 8-byte agent-scope row stores, vmcnt(0), barrier, one relaxed
// ticket; rows summed in fp64 in a fixed order, whichever workgroup happens to be last).  The per-tile statistics exchange (LDS
// partials + a workgroup barrier + 64 global stores per tile) is gone from the tile loop.
// `scratch`: >= 8 KiB of LDS nobody reads any more (a patch buffer: every wave has passed its last step barrier and drained its DMA).
template <int BN, int WGM, int NW>
__device__ __forceinline__ void one_stats_finish(const ConvKArgs& p, float s1, float s2, float* const red, char* const scratch,
                                                 const int wm, const int ccol, const int hi, const int tid) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (hi == 0) {
        red[(wm * BN + ccol) * 2 + 0] = s1;
        red[(wm * BN + ccol) * 2 + 1] = s2;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's last stores and (tail) dummy patch pieces are done
    __builtin_amdgcn_s_barrier();
    const bool fin = p.fin_counter != nullptr;
    const bool pairx = p.pair_x != 0;
    const int cs = p.pair_x == 1 ? 32 : p.pair_x == 2 ? 16 : p.cout;   // statistics columns (paired-x views: accumulator columns c, c + cs, .. are one channel)
    if (tid < BN && tid < cs) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < WGM; ++q) { t1 += red[(q * BN + tid) * 2 + 0]; t2 += red[(q * BN + tid) * 2 + 1]; }
        if (pairx) {
            for (int c2 = tid + cs; c2 < BN; c2 += cs) {
#pragma unroll
                for (int q = 0; q < WGM; ++q) { t1 += red[(q * BN + c2) * 2 + 0]; t2 += red[(q * BN + c2) * 2 + 1]; }
            }
        }
        float* const dst = p.stats + ((long long)blockIdx.x * cs + tid) * 2;
        if (fin) {
            const unsigned long long bits = (unsigned long long)__float_as_uint(t1) | ((unsigned long long)__float_as_uint(t2) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            dst[0] = t1;
            dst[1] = t2;
        }
    }
    if (!fin) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(scratch + 8192);
    const int total = (int)gridDim.x;
    if (tid == 0) {
        const int tk = __hip_atomic_fetch_add(p.fin_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == total - 1 ? 1 : 0;
        if (last) __hip_atomic_store(p.fin_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    constexpr int PH = NW * 64 / BN;
    double* const acc2 = reinterpret_cast<double*>(scratch);      // [PH][BN][2] fp64 = 8 KiB
    const int c = tid % BN, ph = tid / BN;
    double d1 = 0.0, d2 = 0.0;
    if (c < cs) sum_stat_rows<8>(p.stats, (long long)c * 2, (long long)cs * 2, ph, PH, total, d1, d2);
    acc2[(ph * BN + c) * 2 + 0] = d1;
    acc2[(ph * BN + c) * 2 + 1] = d2;
    __syncthreads();
    if (ph == 0 && c < cs) {
        d1 = 0.0; d2 = 0.0;
#pragma unroll
        for (int q = 0; q < PH; ++q) { d1 += acc2[(q * BN + c) * 2 + 0]; d2 += acc2[(q * BN + c) * 2 + 1]; }
        const double mean = d1 * p.fin_inv_count;
        double var = d2 * p.fin_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)p.fin_eps);
        const double g = p.fin_gamma ? (double)p.fin_gamma[c] : 1.0;
        const double b = p.fin_beta ? (double)p.fin_beta[c] : 0.0;
        const double sc = g * invstd;
        p.fin_out[c] = (float)sc;
        p.fin_out[cs + c] = (float)(b - mean * sc);
        p.fin_out[2 * cs + c] = (float)mean;
        p.fin_out[3 * cs + c] = (float)invstd;
        if (p.fin_rmean) p.fin_rmean[c] = (1.f - p.fin_momentum) * p.fin_rmean[c] + p.fin_momentum * (float)mean;
        if (p.fin_rvar)  p.fin_rvar[c]  = (1.f - p.fin_momentum) * p.fin_rvar[c] + p.fin_momentum * (float)(var * p.fin_unbias);
    }
}

// Four consecutive channels of one output pixel, element index e of the raw tensor: fp32 (16 bytes) or -- round 6,
// V2V_OUT_RAW_ACT_NHWC -- rounded to bf16 (8 bytes).  The raw tensor of these HBM-bound layers is written once and read once by
// bn_apply: as bf16 the pair moves 4 instead of 8 bytes per element (the statistics are taken from the fp32 accumulators either way).
__device__ __forceinline__ void one_store4(float* out, unsigned e, const f32x4 v4, bool raw_bf16) {
    if (raw_bf16) {
        uint2 pk;
        pk.x = pack_bf16x2(v4[0], v4[1]);
        pk.y = pack_bf16x2(v4[2], v4[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + e) = pk;
    } else *reinterpret_cast<f32x4*>(out + e) = v4;
}

template <typename T, int TH, int TW, int BN>
__global__ __launch_bounds__(512) void conv3x3_one_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BM = TH * TW;
    constexpr int PW = TW + 2, PR = (TH + 2) * PW;
    constexpr int WGM = 4, WGN = 2, NW = 8;
    constexpr int SS = 4;                                     // K sub-steps (16 elements each) of a step
    constexpr int NG = (PR + 7) / 8;
    constexpr int GP = (NG + NW - 1) / NW;                    // patch pieces per wave
    constexpr int PATCH = GP * NW * 1024;
    constexpr int BST = BN * 128;                             // one weight slice: BN rows of one 128-byte chunk
    constexpr int LB = BN / 8 / NW;                           // weight pieces per wave per slice
    constexpr int WBYTES = 9 * BST;
    constexpr int RED = WGM * BN * 2 * 4;                     // per-row-of-waves (sum, sum^2) partials
    constexpr int EPI = RED + NW * 4096;                      // + one 4 KiB transposition block per wave
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NMMA = SS * TM * TN;
    static_assert(sizeof(T) == 2, "single-chunk persistent tile: bf16 (64 input channels = one 128-byte chunk)");
    static_assert(TW == 32 && WM % 32 == 0 && WN == 32 && TN == 1 && LB >= 1, "tile geometry (one 32-pixel tile row per row fragment)");
    static_assert(WBYTES + PATCH + EPI <= 160 * 1024, "LDS");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wres = smem;                                  // resident weights: [tap][BN rows][128 B]
    char* const patch = smem + WBYTES;
    float* const red = reinterpret_cast<float*>(smem + WBYTES + PATCH);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    float* const tw = reinterpret_cast<float*>(smem + WBYTES + PATCH + RED) + wid * 1024;
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const bool pairx = p.pair_x != 0;
    const char* const zp = p.zero_page;
    const int ntot = p.m_tiles;                               // n_tiles == 1 (host check): the resident weights serve every tile
    const int G = (int)gridDim.x;

    // ---------------- weights: once ----------------
    {
        const int lrow = wid * 8 + (lane >> 3);
        const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            long long r = (long long)lrow + NW * 8 * i;
            r = r < p.cout_p ? r : p.cout_p - 1;
            const char* const wp = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) glds16(wp + tap * 128, wres + tap * BST + wid * 1024 + i * NW * 1024);
        }
    }

    // ---------------- patch loader ----------------
    // piece k of this wave = 8 patch pixels x 128 B; pixel q of the (TH+2) x (TW+2) patch sits at row pr, column pc.  For an INTERIOR
    // tile (no pixel of the patch outside the image) the source offset is tile base + a tile-independent term: one add per piece.
    int rel[GP];                                              // ((pr - 1) * W + (pc - 1)) * cs * 2 + swizzled 16-byte slot, or < 0: no such pixel
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        rel[k] = q < PR ? ((pr - 1) * W + (pc - 1)) * cs * (int)sizeof(T) + ls * 16 : -1;
    }
    auto issue_patch_of = [&](const int n_img, const int oh0, const int ow0) __attribute__((always_inline)) {
        const bool interior = oh0 >= 1 && ow0 >= 1 && oh0 + TH + 1 <= H && ow0 + TW + 1 <= W;     // wave-uniform
        if (interior) {
            const char* const base = p.in + ((long long)(n_img * H + oh0) * W + ow0) * cs * (long long)sizeof(T);
#pragma unroll
            for (int k = 0; k < GP; ++k) glds16(rel[k] != -1 ? base + rel[k] : zp, patch + (k * NW + wid) * 1024);
        } else {
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
                iw = (reflect && !pairx) ? rw : iw;                     // paired-x: pixel -1 = pixel 1 lives in paired pixel 0 (clamp below)
                ok = ok && (unsigned)ih < (unsigned)H && ((unsigned)iw < (unsigned)W || (reflect && pairx));
                ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
                const unsigned off = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
                glds16(ok ? p.in + off : zp, patch + (k * NW + wid) * 1024);
            }
        }
    };

    // ---------------- fragment addressing (tile independent) ----------------
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
    const char* const wrow = wres + (wn * WN + lr) * 128;

    Frag fa[2][SS][TM], fb[2][SS][TN];
    auto read_step = [&](auto tapc, auto parc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int PARN = decltype(parc)::value;
        constexpr int tq = (TAP / 3) * PW + (TAP % 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int qv = qb[i];
            asm volatile("" : "+v"(qv));                   // opaque: no hoisting of 9 x TM address sets out of the tile loop
            const int q = qv + tq;
            const char* const arow = patch + q * 128;
            const int ax = (q >> 1) & 7;
#pragma unroll
            for (int s = 0; s < SS; ++s) fa[PARN][s][i] = *reinterpret_cast<const Frag*>(arow + (((s * 2 + hi) ^ ax) << 4));
        }
#pragma unroll
        for (int s = 0; s < SS; ++s) fb[PARN][s][0] = *reinterpret_cast<const Frag*>(wrow + TAP * BST + foff[s]);
    };

    // ---------------- epilogue operands (tile independent) ----------------
    // ONE output mode: raw fp32 NHWC + one (sum, sum^2) statistics row per tile; full tiles only (host check), so no element carries a
    // predicate.  The arithmetic and its order are conv_epilogue's fast path: bit-identical to tiles 80 / 94.
    float* const out = reinterpret_cast<float*>(p.out);
    const unsigned cs_out = (unsigned)p.cout_stride;
    const bool raw_bf16 = p.out_mode == V2V_OUT_RAW_ACT_NHWC;   // (wave-uniform) raw output rounded to bf16: half the bytes
    const int ccol = wn * WN + lr;                            // this lane's output channel in the accumulator layout
    const float bv = (p.bias != nullptr && ccol < p.cout) ? p.bias[pairx ? (ccol & 31) : ccol] : 0.f;
    const int vcol = wn * WN + 4 * (lane & 7);                // first channel of the 16-byte vectors this lane stores
    const bool vfull = vcol + 4 <= p.cout;                    // cout % 4 == 0 and 16-byte aligned rows (host check)
    const bool want_stats = p.stats != nullptr;

    // ---------------- first tile ----------------
    int vb = (int)blockIdx.x;                                 // virtual block id of the current tile
    int lin, slice, nt, mt, n_img, th, twi;
    if (vb < ntot) {
        patch_tile_index(p, xcd_remap(vb, ntot), lin, slice, nt, mt, n_img, th, twi);
        issue_patch_of(n_img, th * TH, twi * TW);
    }
    one_stagger(p);

    float s1 = 0.f, s2 = 0.f;                                 // this lane's channel, over every tile of this workgroup (one_stats_finish)
    while (vb < ntot) {
        const int oh0 = th * TH, ow0 = twi * TW;
        const int c_img = n_img;
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the patch (first tile: and of the weights) have landed
        __builtin_amdgcn_s_barrier();                        // ... everyone's: the patch is complete; the previous tile's `red` rows are retired
        read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        // 9 steps, no barrier: step t multiplies set t & 1 while step t+1 is read into the other set
        static_for<9>([&](auto tc) {
            constexpr int TAP = decltype(tc)::value;
            constexpr int PAR = TAP & 1;
            static_for<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int s = m / TM, i = m % TM;
                Mma<T>::run(fa[PAR][s][i], fb[PAR][s][0], acc[i]);
                if constexpr (m == 0 && TAP < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_step(std::integral_constant<int, TAP + 1>{}, std::integral_constant<int, 1 - PAR>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave has finished reading the patch

        // the next tile's patch streams in under this tile's epilogue
        vb += G;
        if (vb < ntot) {
            patch_tile_index(p, xcd_remap(vb, ntot), lin, slice, nt, mt, n_img, th, twi);
            issue_patch_of(n_img, th * TH, twi * TW);
        }

        // ---- epilogue: statistics + 16-byte stores through the wave's transposition block ----
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][r] + bv;
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
                tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // rows wm * WM + i * 32 .. + 31 of the tile = tile row wm * (WM / TW) + i (TW == 32), columns 0 .. 31
            const unsigned e0 = ((unsigned)((c_img * H + oh0 + wm * (WM / TW) + i) * W + ow0)) * cs_out + (unsigned)vcol;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(tw + ((lane >> 3) + 8 * k) * 32 + 4 * (lane & 7));
                if (vfull) one_store4(out, e0 + (unsigned)((lane >> 3) + 8 * k) * cs_out, v4, raw_bf16);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the block is in registers before the next one overwrites it
        }
    }
    // (no workgroup barrier between a tile's epilogue and the next tile's steps is needed: the transposition blocks are wave private,
    //  the next patch was issued into a buffer every wave had finished reading, and the statistics stay in registers.  NOTE for any
    //  barrier added to this loop: `s_waitcnt lgkmcnt(0)` + `s_barrier`, NOT __syncthreads() -- its fence is a vmcnt(0), it would drain
    //  the next patch's LDS-DMA and this tile's stores every tile)
    if (want_stats) one_stats_finish<BN, WGM, NW>(p, s1, s2, red, patch, wm, ccol, hi, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Tile 141: the same kernel with TWO patch buffers (round 5, second step).  Tile 140 hides the next patch only behind the epilogue
// (~1.5 us); measured 72 us on the 64 -> 64 layer at 1024x512 (tile 94: 89.5, HBM bound 32): per tile ~3 us of patch latency + 2.5 us
// of steps + the epilogue are still in series.  Here the patch of tile i+1 travels during the WHOLE of tile i (steps + epilogue):
//   LDS = 72 KB weights + 2 x 43 KB patches (exactly the 43 groups of 8 pixels a (8+2) x (32+2) patch has, not rounded up to the
//   waves) + 2 KB statistics partials = 160 KB: all of it -- so the epilogue's 8 x 4 KB transposition blocks live in the patch buffer
//   the steps have just finished with; the patch of tile i+2 is issued into that buffer after the epilogue.
//   vmcnt: at the top of tile i+1 only the pieces of tile i+2 (issued last) may stay in flight -- `vmcnt(n)` with n = this wave's
//   pieces per patch; the stores of tile i's epilogue are older and retire with the wait (no store count enters the immediate, so a
//   channel tile whose stores are branched around cannot break it).  A workgroup without a tile i+2 issues the same number of
//   dummy pieces (zero page) so that the count holds in the tail.
// ASYNC 2 (tile 143): tile 141's order and 16-byte stores, but the counted wait at the top of tile i+1 is `vmcnt(n + 8)`: this tile's patch is
// OLDER than the 8 stores of tile i's epilogue and the patch of tile i+2 behind them, so both may stay in flight -- a wave no longer
// waits for its own stores right after issuing them (tile 141's `vmcnt(n)` did: 39 % of its wave cycles were parked,
// profiles/r05_v7_onepmc.txt).  The stores are unconditional (the immediate counts them): exactly 64 output channels (host check).
// (ASYNC 1, tile 142 -- stores straight from the accumulators, issued behind the patch of tile i+2 -- measured slower, 75.8 us on the
// 2048-tile layer: 4-byte stores lose to the LDS-transposed 16-byte ones; removed in round 6)
template <typename T, int TH, int TW, int BN, int ASYNC = 0>
__global__ __launch_bounds__(512) void conv3x3_one_db_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int BM = TH * TW;
    constexpr int PW = TW + 2, PR = (TH + 2) * PW;
    constexpr int WGM = 4, WGN = 2, NW = 8;
    constexpr int SS = 4;
    constexpr int NG = (PR + 7) / 8;                          // groups of 8 patch pixels (1 KB each)
    constexpr int GP = (NG + NW - 1) / NW;                    // pieces of the fullest waves
    constexpr int NFULL = NG - (GP - 1) * NW;                 // waves 0 .. NFULL-1 carry GP pieces, the others GP - 1
    constexpr int PATCH = NG * 1024;
    constexpr int BST = BN * 128;
    constexpr int LB = BN / 8 / NW;
    constexpr int WBYTES = 9 * BST;
    constexpr int RED = WGM * BN * 2 * 4;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NMMA = SS * TM * TN;
    static_assert(sizeof(T) == 2 && TW == 32 && WM % 32 == 0 && WN == 32 && TN == 1 && LB >= 1, "tile geometry");
    static_assert(WBYTES + 2 * PATCH + RED <= 160 * 1024, "LDS");
    static_assert(PATCH >= NW * 4096, "the epilogue's transposition blocks live in the patch buffer the steps are done with");
    static_assert(NFULL >= 1 && NFULL <= NW && GP >= 2, "piece split");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wres = smem;
    char* const pbase = smem + WBYTES;                        // patch buffers at pbase, pbase + PATCH
    float* const red = reinterpret_cast<float*>(smem + WBYTES + 2 * PATCH);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const bool reflect = p.pad_mode == V2V_PAD_REFLECT;
    const bool pairx = p.pair_x != 0;
    const char* const zp = p.zero_page;
    const int ntot = p.m_tiles;
    const int G = (int)gridDim.x;
    const bool full_wave = wid < NFULL;                       // wave-uniform: this wave carries GP pieces per patch

    {   // weights: once
        const int lrow = wid * 8 + (lane >> 3);
        const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            long long r = (long long)lrow + NW * 8 * i;
            r = r < p.cout_p ? r : p.cout_p - 1;
            const char* const wp = p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) glds16(wp + tap * 128, wres + tap * BST + wid * 1024 + i * NW * 1024);
        }
    }

    int rel[GP];
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        rel[k] = q < PR ? ((pr - 1) * W + (pc - 1)) * cs * (int)sizeof(T) + ls * 16 : -1;
    }
    // `vbn` >= ntot: no such tile -- the same number of pieces from the zero page (the vmcnt immediates count them)
    auto issue_patch_vb = [&](const int vbn, char* const buf) __attribute__((always_inline)) {
        int l_, s_, nt_, mt_, n_img = 0, th_ = 0, tw_ = 0;
        const bool real = vbn < ntot;
        if (real) patch_tile_index(p, xcd_remap(vbn, ntot), l_, s_, nt_, mt_, n_img, th_, tw_);
        const int oh0 = th_ * TH, ow0 = tw_ * TW;
        const bool interior = real && oh0 >= 1 && ow0 >= 1 && oh0 + TH + 1 <= H && ow0 + TW + 1 <= W;     // wave-uniform
        if (!real) {
#pragma unroll
            for (int k = 0; k < GP; ++k)
                if (k < GP - 1 || full_wave) glds16(zp, buf + (k * NW + wid) * 1024);
        } else if (interior) {
            const char* const base = p.in + ((long long)(n_img * H + oh0) * W + ow0) * cs * (long long)sizeof(T);
#pragma unroll
            for (int k = 0; k < GP; ++k)
                if (k < GP - 1 || full_wave) glds16(rel[k] != -1 ? base + rel[k] : zp, buf + (k * NW + wid) * 1024);
        } else {
#pragma unroll
            for (int k = 0; k < GP; ++k) {
                if (!(k < GP - 1 || full_wave)) continue;
                const int q = (k * NW + wid) * 8 + (lane >> 3);
                const int ls = (lane & 7) ^ ((q >> 1) & 7);
                const int pr = q / PW, pc = q - pr * PW;
                int ih = oh0 + pr - 1, iw = ow0 + pc - 1;
                bool ok = q < PR;
                int rh = ih < 0 ? -ih : ih;  rh = rh >= H ? 2 * H - 2 - rh : rh;
                int rw = iw < 0 ? -iw : iw;  rw = rw >= W ? 2 * W - 2 - rw : rw;
                ih = reflect ? rh : ih;
                iw = (reflect && !pairx) ? rw : iw;                     // paired-x: pixel -1 = pixel 1 lives in paired pixel 0 (clamp below)
                ok = ok && (unsigned)ih < (unsigned)H && ((unsigned)iw < (unsigned)W || (reflect && pairx));
                ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
                const unsigned off = (unsigned)(((long long)((n_img * H + ih) * W + iw) * cs + ls * VEC) * (long long)sizeof(T));
                glds16(ok ? p.in + off : zp, buf + (k * NW + wid) * 1024);
            }
        }
    };

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
    const char* const wrow = wres + (wn * WN + lr) * 128;

    Frag fa[2][SS][TM], fb[2][SS][TN];
    auto read_step = [&](auto tapc, auto parc, const char* const patch) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int PARN = decltype(parc)::value;
        constexpr int tq = (TAP / 3) * PW + (TAP % 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int qv = qb[i];
            asm volatile("" : "+v"(qv));
            const int q = qv + tq;
            const char* const arow = patch + q * 128;
            const int ax = (q >> 1) & 7;
#pragma unroll
            for (int s = 0; s < SS; ++s) fa[PARN][s][i] = *reinterpret_cast<const Frag*>(arow + (((s * 2 + hi) ^ ax) << 4));
        }
#pragma unroll
        for (int s = 0; s < SS; ++s) fb[PARN][s][0] = *reinterpret_cast<const Frag*>(wrow + TAP * BST + foff[s]);
    };

    float* const out = reinterpret_cast<float*>(p.out);
    const unsigned cs_out = (unsigned)p.cout_stride;
    const bool raw_bf16 = p.out_mode == V2V_OUT_RAW_ACT_NHWC;   // (wave-uniform) raw output rounded to bf16: half the bytes
    const int ccol = wn * WN + lr;
    const float bv = (p.bias != nullptr && ccol < p.cout) ? p.bias[pairx ? (ccol & 31) : ccol] : 0.f;
    const int vcol = wn * WN + 4 * (lane & 7);
    const bool vfull = vcol + 4 <= p.cout;
    const bool want_stats = p.stats != nullptr;

    int vb = (int)blockIdx.x;
    if (vb >= ntot) return;                                   // (the host launches min(tiles, CUs) workgroups)
    issue_patch_vb(vb, pbase);
    issue_patch_vb(vb + G, pbase + PATCH);
    one_stagger(p);
    int cur = 0;
    bool first = true;
    float s1 = 0.f, s2 = 0.f;                                 // this lane's channel, over every tile of this workgroup (one_stats_finish)
    while (vb < ntot) {
        int lin, slice, nt, mt, n_img, th, twi;
        patch_tile_index(p, xcd_remap(vb, ntot), lin, slice, nt, mt, n_img, th, twi);
        const int oh0 = th * TH, ow0 = twi * TW;
        char* const patch = pbase + cur * PATCH;
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        // only the patch issued LAST (tile i+1, or its dummy) may still be in flight: this tile's patch, the weights and the stores of
        // the previous epilogue are older
        constexpr int NST = 8;                               // stores per wave and tile: 16-byte (or 8-byte bf16) through the LDS block
        if (ASYNC != 0 && !first) {                          // ... and, ASYNC 2 (tile 143), the stores of the previous tile's epilogue, issued
            // in front of that patch -- YOUNGER than this tile's patch, so they may stay in flight)
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP + NST) : "memory");
            else           asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP - 1 + NST) : "memory");
        } else {
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP) : "memory");
            else           asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP - 1) : "memory");
        }
        first = false;
        __builtin_amdgcn_s_barrier();
        read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, patch);
        static_for<9>([&](auto tc) {
            constexpr int TAP = decltype(tc)::value;
            constexpr int PAR = TAP & 1;
            static_for<NMMA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int s = m / TM, i = m % TM;
                Mma<T>::run(fa[PAR][s][i], fb[PAR][s][0], acc[i]);
                if constexpr (m == 0 && TAP < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_step(std::integral_constant<int, TAP + 1>{}, std::integral_constant<int, 1 - PAR>{}, patch);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave has finished reading this patch: the buffer becomes epilogue scratch

        {
        float* const tw = reinterpret_cast<float*>(patch) + wid * 1024;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][r] + bv;
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
                tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned e0 = ((unsigned)((n_img * H + oh0 + wm * (WM / TW) + i) * W + ow0)) * cs_out + (unsigned)vcol;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(tw + ((lane >> 3) + 8 * k) * 32 + 4 * (lane & 7));
                if (ASYNC == 2 || vfull) one_store4(out, e0 + (unsigned)((lane >> 3) + 8 * k) * cs_out, v4, raw_bf16);   // ASYNC 2: exactly 64 channels (host check): unconditional, the vmcnt immediate counts them
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave's transposition block is retired
        __builtin_amdgcn_s_barrier();                        // (NOT __syncthreads(): its fence is a vmcnt(0) -- prefetch and stores must stay in flight)
        // the patch of tile i+2 into the buffer this tile is done with
        issue_patch_vb(vb + 2 * G, patch);
        vb += G;
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (want_stats) one_stats_finish<BN, WGM, NW>(p, s1, s2, red, pbase, wm, ccol, hi, tid);
}

// Tile 114: ConvTranspose2d(3x3, stride 2, padding 1, output_padding 1) of a single-chunk layer (64 bf16 input channels) with at most 32
// output channels -- the last up-sampling stage of the finest generators (models/networks.py:254-260 at ngf_s = 32: 64 -> 32 at
// 1024x512 -> 2048x1024, twice per frame; the foreground tower's 64 -> 32 one scale below) -- as a PERSISTENT, WEIGHTS-RESIDENT kernel:
// conv3x3_t2_kernel.h's arithmetic (all four output-parity classes of a tile of INPUT positions per workgroup, t2k's step table:
// same products, same order per accumulator -- bit-identical raw output) on conv3x3_one_db_kernel's skeleton.
//   On tile 112 such a layer is 2048 workgroups of 26 us each (210 us; HBM bound 53): half of every workgroup's MFMAs multiply the 32
//   padding columns of its 64-wide channel tile, 36 KB of weights and the prologue / four passes of the shared epilogue (one per class,
//   a workgroup barrier each) are paid per tile.  Here: 9 x [32 rows][128 B] weight slices resident (36 KB), two exact-size patch
//   buffers ((TH+1) x (TW+1) input pixels = 38 KB each; the patch of tile i+1 travels during the whole of tile i), 8 waves x one tile
//   row of 32 positions x 32 channels x 4 classes (4 accumulators per wave, no padding columns), 9 barrier-free steps, a lean epilogue
//   per class through the wave's 4 KB transposition block (inside the patch buffer the steps are done with): 16-byte stores, each
//   8-lane group one 128-byte output pixel; statistics in registers across classes and tiles, one row per workgroup
//   (one_stats_finish).  Full tiles only (H % TH == 0, W % TW == 0, OH = 2H, OW = 2W: host check).
template <typename T, int TH, int TW, int BN>
__global__ __launch_bounds__(512) void conv3x3_t2_one_kernel(const ConvKArgs p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int NW = 8, SS = 4;
    constexpr int PW = TW + 1, PR = (TH + 1) * PW;
    constexpr int NG = (PR + 7) / 8;
    constexpr int GP = (NG + NW - 1) / NW;
    constexpr int NFULL = NG - (GP - 1) * NW;
    constexpr int PATCH = NG * 1024;
    constexpr int BST = BN * 128;
    constexpr int WBYTES = 9 * BST;
    constexpr int WPIECES = 9 * BN / 8;                       // 1 KiB pieces (8 rows of one tap) of the resident weights
    constexpr int WPW = (WPIECES + NW - 1) / NW;
    constexpr int RED = NW * BN * 2 * 4;
    static_assert(sizeof(T) == 2 && TW == 32 && TH == NW && BN == 32, "one tile row of 32 positions x 32 channels x 4 classes per wave");
    static_assert(WBYTES + 2 * PATCH + RED <= 160 * 1024 && PATCH >= NW * 4096 && NFULL >= 1 && NFULL <= NW && GP >= 2, "LDS / piece split");
    typedef typename Mma<T>::Frag Frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wres = smem;
    char* const pbase = smem + WBYTES;
    float* const red = reinterpret_cast<float*>(smem + WBYTES + 2 * PATCH);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, cs = p.cin_stride;
    const char* const zp = p.zero_page;
    const int ntot = p.m_tiles;
    const int G = (int)gridDim.x;
    const bool full_wave = wid < NFULL;

#pragma unroll
    for (int i = 0; i < WPW; ++i) {                           // weights: once (full-tap chunk-major rows: tap ky * 3 + kx at byte 128 (ky * 3 + kx))
        const int j = wid + NW * i;
        if (j < WPIECES) {
            const int tap = j / (BN / 8);
            long long r = (j % (BN / 8)) * 8 + (lane >> 3);
            const int lslot = (lane & 7) ^ (((int)r >> 1) & 7);
            r = r < p.cout_p ? r : p.cout_p - 1;
            glds16(p.w + ((long long)p.woff[0] + r * p.wrow[0] + lslot * VEC) * (long long)sizeof(T) + tap * 128, wres + j * 1024);
        }
    }

    int rel[GP];                                              // (pr * W + pc) * cs * 2 + swizzled slot of this lane's patch pixel, or -1
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const int q = (k * NW + wid) * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((q >> 1) & 7);
        const int pr = q / PW, pc = q - pr * PW;
        rel[k] = q < PR ? (pr * W + pc) * cs * (int)sizeof(T) + ls * 16 : -1;
    }
    auto issue_patch_vb = [&](const int vbn, char* const buf) __attribute__((always_inline)) {
        int l_, s_, nt_, mt_, n_img = 0, th_ = 0, tw_ = 0;
        const bool real = vbn < ntot;
        if (real) patch_tile_index(p, xcd_remap(vbn, ntot), l_, s_, nt_, mt_, n_img, th_, tw_);
        const int a0 = th_ * TH, b0 = tw_ * TW;
        const bool interior = real && a0 + TH < H && b0 + TW < W;                  // the patch's extra row and column are inside the image
        const char* const base = p.in + ((long long)(n_img * H + a0) * W + b0) * cs * (long long)sizeof(T);
#pragma unroll
        for (int k = 0; k < GP; ++k) {
            if (!(k < GP - 1 || full_wave)) continue;
            bool ok = real && rel[k] != -1;
            if (!interior && ok) {
                const int q = (k * NW + wid) * 8 + (lane >> 3);
                const int pr = q / PW, pc = q - pr * PW;
                ok = a0 + pr < H && b0 + pc < W;                                  // zero beyond the image (the transposed layer's padding)
            }
            glds16(ok ? base + rel[k] : zp, buf + (k * NW + wid) * 1024);
        }
    };

    const int lr = lane & 31, hi = lane >> 5;
    const int qb = wid * PW + lr;                             // this lane's input position in the patch (tile row = wave)
    int foff[SS];
#pragma unroll
    for (int s = 0; s < SS; ++s) foff[s] = ((s * 2 + hi) ^ ((lr >> 1) & 7)) << 4;
    const char* const wrow = wres + lr * 128;

    Frag fa[2][SS], fb[2][SS];
    auto read_step = [&](auto tapc, auto parc, const char* const patch) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int PARN = decltype(parc)::value;
        constexpr int tq = t2k::DY[TAP] * PW + t2k::DX[TAP];
        int qv = qb;
        asm volatile("" : "+v"(qv));
        const int q = qv + tq;
        const char* const arow = patch + q * 128;
        const int ax = (q >> 1) & 7;
#pragma unroll
        for (int s = 0; s < SS; ++s) fa[PARN][s] = *reinterpret_cast<const Frag*>(arow + (((s * 2 + hi) ^ ax) << 4));
#pragma unroll
        for (int s = 0; s < SS; ++s) fb[PARN][s] = *reinterpret_cast<const Frag*>(wrow + t2k::KK[TAP] * BST + foff[s]);
    };

    float* const out = reinterpret_cast<float*>(p.out);
    const unsigned cs_out = (unsigned)p.cout_stride;
    const bool raw_bf16 = p.out_mode == V2V_OUT_RAW_ACT_NHWC;   // (wave-uniform) raw output rounded to bf16: half the bytes
    const int OH = p.OH, OW = p.OW;
    const float bv = (p.bias != nullptr && lr < p.cout) ? p.bias[p.pair_x ? (lr & 15) : lr] : 0.f;      // (paired-x view: 32 columns = 2 pixels x 16 channels)
    const int vcol = 4 * (lane & 7);
    const bool vfull = vcol + 4 <= p.cout;
    const bool want_stats = p.stats != nullptr;

    int vb = (int)blockIdx.x;
    if (vb >= ntot) return;                                   // (the host launches min(tiles, CUs) workgroups)
    issue_patch_vb(vb, pbase);
    issue_patch_vb(vb + G, pbase + PATCH);
    int cur = 0;
    float s1 = 0.f, s2 = 0.f;                                 // this lane's channel (lr), over the 4 classes of every tile of this workgroup
    while (vb < ntot) {
        int lin, slice, nt, mt, n_img, th, twi;
        patch_tile_index(p, xcd_remap(vb, ntot), lin, slice, nt, mt, n_img, th, twi);
        const int a0 = th * TH, b0 = twi * TW;
        char* const patch = pbase + cur * PATCH;
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

        // only the patch issued LAST (tile i+1, or its dummy) may still be in flight: this tile's patch, the weights and the stores of the
        // previous epilogue are older
        if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP) : "memory");
        else           asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP - 1) : "memory");
        __builtin_amdgcn_s_barrier();
        read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, patch);
        static_for<9>([&](auto tc) {
            constexpr int TAP = decltype(tc)::value;
            constexpr int PAR = TAP & 1;
            constexpr int CL = t2k::CLS[TAP];
            static_for<SS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                Mma<T>::run(fa[PAR][s], fb[PAR][s], acc[CL]);
                if constexpr (s == 0 && TAP < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_step(std::integral_constant<int, (TAP < 8 ? TAP + 1 : 8)>{}, std::integral_constant<int, 1 - PAR>{}, patch);    // (the discarded branch of TAP 8 is still instantiated: keep the table index in range)
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave has finished reading this patch: the buffer becomes epilogue scratch

        float* const tw = reinterpret_cast<float*>(patch) + wid * 1024;
        static_for<4>([&](auto cc_) {
            constexpr int CL = decltype(cc_)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[CL][r] + bv;
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
                tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // block row m = input position (a0 + wid, b0 + m) -> output pixel (2 (a0 + wid) + py, 2 (b0 + m) + px)
            const unsigned e0 = ((unsigned)((n_img * OH + 2 * (a0 + wid) + (CL >> 1)) * OW + 2 * b0 + (CL & 1))) * cs_out + (unsigned)vcol;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(tw + ((lane >> 3) + 8 * k) * 32 + 4 * (lane & 7));
                if (vfull) one_store4(out, e0 + (unsigned)(2 * ((lane >> 3) + 8 * k)) * cs_out, v4, raw_bf16);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (NOT __syncthreads(): its fence is a vmcnt(0) -- prefetch and stores must stay in flight)
        issue_patch_vb(vb + 2 * G, patch);
        vb += G;
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (want_stats) one_stats_finish<BN, NW, NW>(p, s1, s2, red, pbase, wid, lr, hi, tid);
}

// Grid of the persistent tiles = statistics rows they leave (v2v_conv_stats_rows): one workgroup per CU (the LDS footprint allows no
// second one), never more workgroups than tiles.
static inline int one_grid_size(int ntot, int cus) {
    if (cus < 8) cus = 256;
    return ntot < cus ? ntot : cus;
}

template <typename T>
static inline int launch_one_typed(int cfg, const ConvKArgs& k_in, int cus, hipStream_t s) {
    static const int stag = [] {                              // V2V_ONE_STAGGER=<units>[,<groups>] (experiment; default off)
        const char* e = getenv("V2V_ONE_STAGGER");
        if (!e) return 0;
        int u = 0, g = 2;
        sscanf(e, "%d,%d", &u, &g);
        return (u & 0xff) | ((g & 0x7f) << 8);
    }();
    ConvKArgs k = k_in;
    k.ablate = (k.ablate & 0xffff) | (stag << 16);
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (cfg == 140) {
            constexpr int TH = 8, TW = 32, BN = 64, NW = 8;
            constexpr int GP = (((TH + 2) * (TW + 2) + 7) / 8 + NW - 1) / NW;
            const size_t lds = (size_t)9 * BN * 128 + (size_t)GP * NW * 1024 + (size_t)(4 * BN * 2 * 4 + NW * 4096);
            void (*kern)(const ConvKArgs) = conv3x3_one_kernel<T, TH, TW, BN>;
            static bool attr_done = false;
            if (!attr_done) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_done = true;
            }
            const int g = one_grid_size(k.m_tiles * k.n_tiles, cus);
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), lds, s, k);
            return check_launch();
        }
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (cfg == 141 || cfg == 143) {
            constexpr int TH = 8, TW = 32, BN = 64, NW = 8;
            constexpr int NG = ((TH + 2) * (TW + 2) + 7) / 8;
            const size_t lds = (size_t)9 * BN * 128 + (size_t)2 * NG * 1024 + (size_t)(4 * BN * 2 * 4);
            void (*kern)(const ConvKArgs) = cfg == 143 ? conv3x3_one_db_kernel<T, TH, TW, BN, 2> : conv3x3_one_db_kernel<T, TH, TW, BN, 0>;
            static bool attr_done[3] = {false, false, false};
            if (!attr_done[cfg - 141]) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_done[cfg - 141] = true;
            }
            const int g = one_grid_size(k.m_tiles * k.n_tiles, cus);
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), lds, s, k);
            return check_launch();
        }
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (cfg == 114) {
            constexpr int TH = 8, TW = 32, BN = 32, NW = 8;
            constexpr int NG = ((TH + 1) * (TW + 1) + 7) / 8;
            const size_t lds = (size_t)9 * BN * 128 + (size_t)2 * NG * 1024 + (size_t)(NW * BN * 2 * 4);
            void (*kern)(const ConvKArgs) = conv3x3_t2_one_kernel<T, TH, TW, BN>;
            static bool attr_done = false;
            if (!attr_done) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_done = true;
            }
            const int g = one_grid_size(k.m_tiles * k.n_tiles, cus);
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), lds, s, k);
            return check_launch();
        }
    }
    set_error("conv: unknown persistent single-chunk tile config %d (bf16 only)", cfg);
    return V2V_EINVAL;
}

}  // namespace v2v
