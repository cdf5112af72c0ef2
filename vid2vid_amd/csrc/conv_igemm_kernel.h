// Implicit-GEMM convolution kernel template (see conv_igemm.hip for the design notes); included by the
// per-dtype instantiation units conv_igemm_bf16.hip / conv_igemm_f32.hip and by the host side.
#pragma once
#include "v2v_internal.h"

namespace v2v {

// Tensors of the SECOND member of a grouped launch (v2v_conv2d_pair; gridDim.z == 2): same geometry as the first,
// its own operands, statistics / finalize outputs, tickets and split-K scratch.
struct ConvGroupPtrs {
    const char* in; const char* w; const float* bias; char* out; float* stats;
    int* fin_counter; const float* fin_gamma; const float* fin_beta; float* fin_out; float* fin_rmean; float* fin_rvar;
    float* slabs; int* sk_counter;
    const char* res0; const char* res1;
};

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
    int ktot[4], kpad[4], wrow[4];   // K extent (padded to the chunk) and row stride of the packed class matrix
    long long woff[4];   // element offset of the class matrix inside w
    // in-kernel norm finalize (last-arriving workgroup of an N tile), optional
    int* fin_counter; const float* fin_gamma; const float* fin_beta; float* fin_out;
    float* fin_rmean; float* fin_rvar;
    float fin_eps, fin_momentum; double fin_inv_count, fin_unbias;
    // split-K: `splitk` workgroups share one output tile (K chunks [nk*s/S, nk*(s+1)/S)); each publishes its
    // fp32 partial tile to `slabs`, the LAST one to arrive (ticket in sk_counter) sums them in slice order
    int splitk; float* slabs; int* sk_counter;
    // weight-stream prefetch: an extra (helper) wave touches the weight lines `pf_dist` K chunks ahead so that the
    // LDS-DMA of the real loaders hits L2 instead of waiting for HBM; workgroups with (mt & pf_mask) != 0 skip it
    int pf_dist, pf_mask;
    int tiles_h, tiles_w; // conv3x3_patch_kernel: output tiles per image (m_tiles = N * tiles_h * tiles_w)
    int ablate;          // profiling ablations (v2v_conv_desc.ablate); results are WRONG when non-zero
    int cls_rev;         // dispatch the parity classes of a transposed convolution in descending order of their tap count
    int act_split, act_b; float act_param_b, out_scale_b;   // conv7x7_head_kernel: channels >= act_split (> 0) use this activation / scale
    const char* res0; const char* res1;   // V2V_OUT_NORM_ACT_NHWC: residuals added after the activation (or NULL)
    // exact division of 0 <= n < 2^31 by the class grid sizes: q = (umulhi(M, n) + n) >> l (v2v_fastdiv_magic), [class][0: OHc*OWc, 1: OWc]
    unsigned div_m[4][2]; int div_l[4][2];
    int* status;         // host-mapped status word of the library (v2v_device_status) or NULL: bit 0 = a fused-norm barrier gave up
    ConvGroupPtrs g1;    // grouped launch (conv3x3_pp2_kernel): operands of block z == 1
    int fin_groups; double* fin_ws;   // two-level in-kernel finalize (rows > 512): row groups and their fp64 (sum, sum^2) rows [groups][cout][2], or 0 / NULL
    int fin_rows;        // statistics rows (= finalize tickets) per channel tile when it is not gridDim.y * m_tiles (conv3x3_t2_kernel: 4 m_tiles), else 0
    unsigned long long* dbg;   // v2v_conv_debug_clocks: [workgroup][8] constant-rate (100 MHz) wall-clock stamps of the kernel's phases, or NULL
    int grp_xcd;         // grouped launch: member 0 on XCDs 0-3, member 1 on XCDs 4-7 (grouped_xcd_map) instead of both members on every XCD
    int ep_slow;         // V2V_EPILOGUE_FAST=0: full tiles take the predicated epilogue paths too (A/B switch of conv_epilogue's fast paths)
    // patch kernels (tiles >= 32): exact division of the tile index by m_tiles, tiles_h * tiles_w and tiles_w as q = (umulhi(M, n) + n) >> l
    // (v2v_fastdiv_magic) -- three scalar instructions each instead of the ~30 of a run-time division, which every wave of every
    // workgroup executed four times before its first load (and two 64-bit divisions for the split-K chunk range, now behind S > 1)
    unsigned idx_m[3]; int idx_l[3];
    // persistent single-chunk tiles on a layer with 64-byte pixels (v2v_conv_desc.w_korder 3): this launch is the PAIRED-X view of a
    // <= 32 -> 32 channel layer (W, OW = half the layer's, 64 -> 64 channels, structured weights): horizontal reflection is a clamp,
    // bias index and statistics column = channel & 31 (the two halves of a paired pixel are the same 32 channels)
    int pair_x;
};

// phase stamp k of this workgroup (thread 0): 0 entry, 1 prologue set up (first loads issued), 2 first tile landed, 3 main loop done,
// 4 outputs stored, 5 statistics row published, 6 exit.  COMPILED OUT of the product build (V2V_STAMP_MASK = 0).  The instrumented
// library is a PROFILING build only:  touch conv_igemm_kernel.h && make -C vid2vid_amd/csrc EXTRA=-DV2V_STAMP_MASK=0x7f, run
// scripts/kernel_phases.py, rebuild without EXTRA.
//   History: rounds 4-5 saw this build fail fp32 golden tests -- deterministically for one tile selection (the 32 -> 32 3x3 layer at
//   8x16 pixels on tile 10 x split-K 2: profiles/r05_v3_stamp_bisect.txt), never on the product build.  Round 6 root-caused it
//   (profiles/r06_v58_stamp_rootcause.txt): not a race and not undefined behaviour of this source -- stamp 3 alone moves the epilogue's
//   `v_accvgpr_read_b32 v17, a15` to the top of the block the loop's exit branch reaches, 2-7 wait states behind the last 16-pass
//   v_mfma_f32_32x32x2_f32 where the hardware needs 18 and has no interlock; the compiler's hazard search prices only the long way
//   round the loop (see mfma_results_ready() below, which now separates the K loop from the epilogue, and
//   scripts/mfma_hazard_check.py, which every build of the library has to pass).
#ifndef V2V_STAMP_MASK
#define V2V_STAMP_MASK 0
#endif
#define V2V_STAMP(p_, k_)                                                                                                   \
    do {                                                                                                                   \
        if (((V2V_STAMP_MASK >> (k_)) & 1) && (p_).dbg != nullptr && threadIdx.x == 0)                                                                       \
            (p_).dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k_)] = wall_clock64(); \
    } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int ntot) {
    const int q = ntot >> 3, r = ntot & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Grouped launch (gridDim.z == 2, ntot = tiles * splitk workgroups per member, ntot % 8 == 0) with the members on DISJOINT XCD
// halves.  The dispatcher deals workgroups to the 8 XCDs round-robin in launch order (x fastest, then z), i.e. XCD = blockIdx.x & 7
// for either z.  xcd_remap gives every XCD ntot/8 consecutive tiles of BOTH members: with the 1024 -> 1024 pair (8 pixel tiles x 16
// channel tiles per member) each XCD's L2 pulls the whole input of both members (2 x 4.2 MB) and 2 x 2 channel tiles of weights
// (2 x 2.4 MB) = 13.1 MB, 105 MB over the fabric for 44 MB of distinct operands (profiles/r03_d1_traffic.json: 114 MB fetched with
// the residuals).  Here XCD x works on member x >> 2 only: ntot/4 consecutive tiles = 4 channel tiles x all 8 pixel tiles, i.e. one
// input (4.2 MB) + 4.7 MB of weights = 8.9 MB per XCD, 71 MB in total.  Returns the tile index, sets the member.
__device__ __forceinline__ int grouped_xcd_map(int bx, int bz, int ntot, int& member) {
    const int xcd = bx & 7;
    const int idx = (bx >> 3) + bz * (ntot >> 3);          // 0 .. ntot/4 - 1: this XCD's workgroups of both z slices
    member = xcd >> 2;
    return (xcd & 3) * (ntot >> 2) + idx;
}

__device__ __forceinline__ int fast_div(int n, unsigned M, int l) { return (int)((__umulhi(M, (unsigned)n) + (unsigned)n) >> l); }

// tile index of a patch-kernel workgroup -> (K slice, channel tile, pixel tile, image, tile row, tile column); see ConvKArgs.idx_m
__device__ __forceinline__ void patch_tile_index(const ConvKArgs& p, const int lin_all, int& lin, int& slice, int& nt, int& mt,
                                                 int& n_img, int& th, int& tw) {
    const int S = p.splitk;
    lin = lin_all; slice = 0;
    if (S > 1) { lin = lin_all / S; slice = lin_all - lin * S; }
    nt = fast_div(lin, p.idx_m[0], p.idx_l[0]);
    mt = lin - nt * p.m_tiles;
    n_img = fast_div(mt, p.idx_m[1], p.idx_l[1]);
    const int trem = mt - n_img * (p.tiles_h * p.tiles_w);
    th = fast_div(trem, p.idx_m[2], p.idx_l[2]);
    tw = trem - th * p.tiles_w;
}
// channel chunks [ccb, ccb + ncc) of K slice `slice` of S (whole 128-byte chunks; S == 1: all of them, no division)
__device__ __forceinline__ void patch_chunk_range(const int ncc_all, const int slice, const int S, int& ccb, int& ncc) {
    ccb = 0; ncc = ncc_all;
    if (S > 1) {
        ccb = (int)(((long long)ncc_all * slice) / S);
        ncc = (int)(((long long)ncc_all * (slice + 1)) / S) - ccb;
    }
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

// gfx950 has no interlock between an MFMA and a later NON-MFMA access of its result registers: software keeps passes + 2 (fp32
// input) / passes + 3 wait states between them, and the compiler's hazard recogniser can miss the edge "last MFMA of the K loop ->
// first accumulator read of the epilogue" (its backwards search marks the loop's MFMA block as visited when it first reaches it the
// long way round, through the loop header: GCNHazardRecognizer::getWaitStatesSince).  Round 6 found exactly that in the
// V2V_STAMP_MASK build of conv_igemm_kernel<float,64,64,2,2,2,false>: `v_accvgpr_read_b32 v17, a15` two wait states behind a 16-pass
// v_mfma_f32_32x32x2_f32 that needs 18 -- the "stamp build miscompute" of round 5 (scripts/stamp_probe.py,
// scripts/mfma_hazard_check.py = the static check every build of the library has to pass, tests/test_cpu_boundary.py).
// Called between a K loop and the first use of its accumulators: 19 wait states (~40 ns once per workgroup), and nothing is
// scheduled across it.
__device__ __forceinline__ void mfma_results_ready() {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
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

// Shared tail of the conv kernels: split-K hand-off, bias / activation / store, per-tile statistics and the in-kernel
// norm finalize.  `pix_of(row)` maps a tile row (0..BM-1) to the output pixel index in [N][OH][OW], or < 0 when the
// row lies outside the layer.  Every wave of the workgroup (the prefetch helper included) must call it: it contains
// workgroup barriers.
template <typename T, int BM, int BN, int WGM, int WGN, bool FUSED_NORM = false, typename PixOf = void>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& p, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], char* smem,
                                              const int tid, const int wm, const int wn, const bool helper,
                                              const int cls, const int tiles, const int lin, const int slice, const int S,
                                              const int nt, const int stat_row, PixOf pix_of, const bool tile_full = false,
                                              const unsigned fin_epoch = 0u) {
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NW = WGM * WGN;
    const int lane = tid & 63;
    const int lr = lane & 31, hi = lane >> 5;
    // ---------------- split-K hand-off ----------------
    // (cdna guide 5 "in-launch split-K reduction", write-through form): every slice stores its fp32 partial
    // tile with 16-byte sc1 stores, every storing wave drains them, workgroup barrier, ONE relaxed agent-scope
    // ticket.  The slice that draws S-1 re-arms the ticket, reads all S slabs back with sc1 loads and sums them
    // in SLICE order, so the result does not depend on which slice happened to be last.  No spin anywhere:
    // nothing can hang.
    if (S > 1) {
        constexpr int NT = NW * 64;
        constexpr unsigned SLAB = (unsigned)BM * BN * 4u;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const long long tile_id = (long long)cls * tiles + lin;
        char* const sbase = reinterpret_cast<char*>(p.slabs) + tile_id * (long long)S * SLAB;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(sbase, 0, (int)(S * SLAB), 0x00020000);
        if (!helper) {
            const unsigned my = (unsigned)slice * SLAB + (unsigned)tid * 16u;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u32x4 v;
                        v[0] = __float_as_uint(acc[i][j][4 * q + 0]); v[1] = __float_as_uint(acc[i][j][4 * q + 1]);
                        v[2] = __float_as_uint(acc[i][j][4 * q + 2]); v[3] = __float_as_uint(acc[i][j][4 * q + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, my + (unsigned)(((i * TN + j) * 4 + q) * NT * 16), 0, 16);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + 16384);
        if (tid == 0) {
            int* cnt = p.sk_counter + tile_id;
            const int tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = tk == S - 1 ? 1 : 0;
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        if (!helper) {
            // S == 2: a + b is commutative, so the reducer adds the OTHER slice's slab to its own registers;
            // S > 2: every slab, the reducer's own included, is read back and summed in slice order 0..S-1
            if (S > 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            for (int sl = 0; sl < S; ++sl) {
                if (S == 2 && sl == slice) continue;
                const unsigned off = (unsigned)sl * SLAB + (unsigned)tid * 16u;
                u32x4 v[TM][TN][4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[i][j][q] = __builtin_amdgcn_raw_buffer_load_b128(
                                rsrc, off + (unsigned)(((i * TN + j) * 4 + q) * NT * 16), 0, 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[i][j][4 * q + 0] += __uint_as_float(v[i][j][q][0]); acc[i][j][4 * q + 1] += __uint_as_float(v[i][j][q][1]);
                            acc[i][j][4 * q + 2] += __uint_as_float(v[i][j][q][2]); acc[i][j][4 * q + 3] += __uint_as_float(v[i][j][q][3]);
                        }
            }
        }
        __syncthreads();                  // the flag word is reused by the norm-finalize hand-off below
    }

    // ---------------- epilogue ----------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    //
    // Round 2: the first version of this block evaluated pix_of(), three output-mode branches, the activation switch and
    // 64-bit index arithmetic PER ELEMENT (~350 instructions x 32-64 elements per lane, 60 KB of straight-line code for
    // the 256 px x 64 tile) -- a fixed ~35 us per launch, a third of the 1024 -> 1024 layer's 93 us
    // (profiles/r02_a3_pp2_ablate.txt: the kernel with an empty main loop took 43 us whatever the K extent).  Now the
    // row -> pixel map is evaluated once per row into registers, the mode / activation dispatch is hoisted out of the
    // element loops, and element offsets are 32-bit (the host rejects outputs of 2^31 elements or more).  Same
    // arithmetic per element, bitwise identical results.
    float* red = reinterpret_cast<float*>(smem);   // [WGM][BN][2]
    const bool want_stats = p.stats != nullptr;
    // FULL tiles (round 4): the caller knows (one scalar test) that every row of the tile is a pixel of the layer; with every
    // channel of the tile inside the layer as well, no element needs a predicate -- the 32 per-lane row -> pixel evaluations
    // below and the compare / select pair of every statistics update go away (~450 of the ~900 vector instructions a wave
    // spends behind the main loop: profiles/r04_d7_fine_scale_conv_ablate.txt).  Wave-uniform branch, the same
    // additions in the same order: bit-identical results; ragged tiles, ragged channel tiles and the helper wave take the old path.
    const bool fast = tile_full && !p.ep_slow && !helper && !(p.ablate & 4) && (nt + 1) * BN <= p.cout &&
                      (p.out_mode == V2V_OUT_RAW_F32_NHWC || (FUSED_NORM && p.out_mode == V2V_OUT_NORM_ACT_NHWC));   // the two paths with a fast form
    int opx[TM][16];                               // output pixel index of this lane's rows, < 0: outside the layer
    if (!fast) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = pix_of(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                opx[i][r] = (helper || (p.ablate & 4)) ? -1 : o;
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) opx[i][r] = 0;         // never read on the fast paths; any valid value for the others
    }
    const unsigned cs_out = (unsigned)p.cout_stride;
    if constexpr (FUSED_NORM) {
        if (p.out_mode == V2V_OUT_NORM_ACT_NHWC) {
            // ---- conv + training-mode norm + activation (+ residuals) in one launch (include/v2v_hip.h, "fused norm") ----
            // 1. this tile's (sum, sum^2) per output channel -> its statistics row (agent-scope store)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol = nt * BN + wn * WN + j * 32 + lr;
                const bool nvalid = ncol < p.cout;
                const float bv = (p.bias != nullptr && nvalid) ? p.bias[ncol] : 0.f;
                float s1 = 0.f, s2 = 0.f;
                if (fast) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][j][r] + bv;
                            s1 += v;
                            s2 = __builtin_fmaf(v, v, s2);
                        }
                } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (opx[i][r] >= 0 && nvalid) {
                            const float v = acc[i][j][r] + bv;
                            s1 += v;
                            s2 = __builtin_fmaf(v, v, s2);          // explicit: the raw and the fused-norm paths must round alike
                        }
                }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (hi == 0) {
                    const int c = wn * WN + j * 32 + lr;
                    red[(wm * BN + c) * 2 + 0] = s1;
                    red[(wm * BN + c) * 2 + 1] = s2;
                }
            }
            // Hand-off (round 6; MI355X_MICROARCH "hand-off granule", profiles/r06_v66_fused_tail_phases.txt): a row is published as
            // self-validating 8-byte granules {value, launch tag} -- one write-through store each, no acknowledgement wait, no arrival
            // ticket -- and every workgroup of the channel tile polls the granules it needs (step 4).  The tag is 1 + the number of
            // fused launches that have completed on this channel tile's words of fin_counter (fin_epoch, read at kernel entry; the last
            // workgroup to leave increments it), so granules of earlier launches in the (zero-initialised, fused-launches-only) row
            // buffer never match.  The ticket form it replaces cost three dependent agent-scope round trips (store acknowledgement,
            // returning atomic, row loads) behind the slowest K loop; this one costs one.
            const int total = p.fin_rows > 0 ? p.fin_rows : (int)gridDim.y * p.m_tiles;
            const unsigned tag = fin_epoch + 1u;
            int* const flag = reinterpret_cast<int*>(smem + 16384);
            if (tid == 0) *flag = 1;
            __syncthreads();
            unsigned long long* const gran = reinterpret_cast<unsigned long long*>(p.stats);      // [row][cout][2] granules
            // ablate 2048 (test of the give-up path): the first workgroup of the channel tile never publishes
            if (tid < BN && !((p.ablate & 2048) && stat_row == 0)) {
                const int ncol = nt * BN + tid;
                if (ncol < p.cout) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int q = 0; q < WGM; ++q) { s1 += red[(q * BN + tid) * 2 + 0]; s2 += red[(q * BN + tid) * 2 + 1]; }
                    unsigned long long* const g = gran + ((long long)stat_row * p.cout + ncol) * 2;
                    __hip_atomic_store(g, (unsigned long long)__float_as_uint(s1) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(g + 1, (unsigned long long)__float_as_uint(s2) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            V2V_STAMP(p, 4);              // fused norm: 4 row published, 5 every row of the channel tile read, 6 scale / shift in LDS,
                                          // 7 normalised tile stored (scripts/fused_tail_phases.py)
            // 3. while the other workgroups of this channel tile arrive: park the pre-norm values (fp32) in LDS in pixel-major
            //    order and fetch this thread's share of the residuals, so that the work behind the barrier is one short,
            //    vectorised pass (the first version normalised in the MFMA layout with 2-byte stores: +8 us per launch)
            constexpr int NTH = NW * 64, PH = NTH / BN;
            constexpr int VEC = ElemTraits<T>::VEC;
            constexpr int CPR = BN / VEC;                            // 16-byte output chunks per pixel row of the tile
            constexpr int NV = BM * CPR / NTH;                       // chunks per thread
            static_assert(BM * CPR % NTH == 0 && NV >= 1, "fused norm: tile / thread split");
            static_assert(20480 + BM * BN * 4 <= 160 * 1024, "fused norm: staging tile");
            float* const stg = reinterpret_cast<float*>(smem + 20480);     // [BM][BN] fp32, behind acc2 / ssl / flag
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = wn * WN + j * 32 + lr;
                const int ncol = nt * BN + c;
                const float bv = (p.bias != nullptr && ncol < p.cout) ? p.bias[ncol] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BN + c] = acc[i][j][r] + bv;
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 rv0[NV], rv1[NV];
            int vpix[NV];
            const T* const res0 = reinterpret_cast<const T*>(p.res0);
            const T* const res1 = reinterpret_cast<const T*>(p.res1);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int v = tid + q * NTH;
                const int row = v / CPR, ch = (v - row * CPR) * VEC;
                const int o = pix_of(row);
                vpix[q] = (o >= 0 && nt * BN + ch < p.cout) ? o : -1;        // cout % VEC == 0 (host check): whole chunks
                rv0[q] = u32x4{0u, 0u, 0u, 0u}; rv1[q] = rv0[q];
                if (vpix[q] >= 0) {
                    const unsigned e = (unsigned)o * cs_out + (unsigned)(nt * BN + ch);
                    if (res0) rv0[q] = *reinterpret_cast<const u32x4*>(res0 + e);
                    if (res1) rv1[q] = *reinterpret_cast<const u32x4*>(res1 + e);
                }
            }
            // (the statistics partials of step 1 and the double-precision partials of step 5 share LDS: every wave is past step 1's reads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // 4. + 5. every row of the channel tile, polled granule by granule (all workgroups resident: host check), then scale / shift
            //    in a fixed order: the arithmetic of the in-kernel finalize below, in every workgroup.  A row that does not appear
            //    within ~1 s (a co-tenant holding compute units) makes the launch give up: NaN outputs + the status flag.
            double* acc2 = reinterpret_cast<double*>(smem);          // [PH][BN][2], <= 8 KiB
            float* ssl = reinterpret_cast<float*>(smem + 12288);     // [2][BN] scale, shift
            {
                const int c = tid % BN, ph = tid / BN;
                const int ncol = nt * BN + c;
                double s1 = 0.0, s2 = 0.0;
                if (ncol < p.cout && ph < PH) {
                    for (int r = ph; r < total; r += PH) {
                        const unsigned long long* const g = gran + ((long long)r * p.cout + ncol) * 2;
                        unsigned long long b1 = 0ull, b2 = 0ull;
                        bool got = false;
                        for (int it = 0; it < (1 << 19); ++it) {
                            b1 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b2 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            got = (unsigned)(b1 >> 32) == tag && (unsigned)(b2 >> 32) == tag;
                            if (got) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (!got) *flag = 0;
                        s1 += (double)__uint_as_float((unsigned)(b1 & 0xffffffffull));
                        s2 += (double)__uint_as_float((unsigned)(b2 & 0xffffffffull));
                    }
                }
                if (ph < PH) { acc2[(ph * BN + c) * 2 + 0] = s1; acc2[(ph * BN + c) * 2 + 1] = s2; }
                __syncthreads();
                V2V_STAMP(p, 5);
                const bool barrier_ok = *flag != 0;
                if (!barrier_ok && tid == 0 && p.status) __hip_atomic_fetch_or(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ph == 0) {
                    float fsc = 0.f, fsh = 0.f;
                    if (ncol < p.cout) {
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
                        fsc = (float)sc; fsh = (float)(b - mean * sc);
                        if (!barrier_ok) fsc = fsh = __uint_as_float(0x7fc00000u);      // poison: the caller sees NaN, not stale data
                        if (stat_row == 0) {             // one writer per channel tile: the [4][cout] record and the running statistics
                            p.fin_out[ncol] = fsc;
                            p.fin_out[p.cout + ncol] = fsh;
                            p.fin_out[2 * p.cout + ncol] = (float)mean;
                            p.fin_out[3 * p.cout + ncol] = (float)invstd;
                            if (p.fin_rmean) p.fin_rmean[ncol] = (1.f - p.fin_momentum) * p.fin_rmean[ncol] + p.fin_momentum * (float)mean;
                            if (p.fin_rvar)  p.fin_rvar[ncol]  = (1.f - p.fin_momentum) * p.fin_rvar[ncol] + p.fin_momentum * (float)(var * p.fin_unbias);
                        }
                    }
                    ssl[c] = fsc; ssl[BN + c] = fsh;
                }
            }
            __syncthreads();
            V2V_STAMP(p, 6);
            // 6. normalise, activate, add the residuals, 16-byte stores (bn_apply_kernel's arithmetic, element for element)
            T* const out = reinterpret_cast<T*>(p.out);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (vpix[q] < 0) continue;
                const int v = tid + q * NTH;
                const int row = v / CPR, ch = (v - row * CPR) * VEC;
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    o[e] = apply_act(stg[row * BN + ch + e] * ssl[ch + e] + ssl[BN + ch + e], p.act, p.act_param);
                const unsigned eo = (unsigned)vpix[q] * cs_out + (unsigned)(nt * BN + ch);
                if constexpr (VEC == 4) {
                    if (res0) { o[0] += __uint_as_float(rv0[q][0]); o[1] += __uint_as_float(rv0[q][1]); o[2] += __uint_as_float(rv0[q][2]); o[3] += __uint_as_float(rv0[q][3]); }
                    if (res1) { o[0] += __uint_as_float(rv1[q][0]); o[1] += __uint_as_float(rv1[q][1]); o[2] += __uint_as_float(rv1[q][2]); o[3] += __uint_as_float(rv1[q][3]); }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + eo) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    if (res0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[2 * e] += __uint_as_float(rv0[q][e] << 16); o[2 * e + 1] += __uint_as_float(rv0[q][e] & 0xffff0000u); }
                    }
                    if (res1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[2 * e] += __uint_as_float(rv1[q][e] << 16); o[2 * e + 1] += __uint_as_float(rv1[q][e] & 0xffff0000u); }
                    }
                    u32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out) + eo) = pk;
                }
            }
            // everyone has read the rows (the barrier behind step 5): the last to leave re-arms the departure ticket and advances the
            // launch tag for the next launch / graph replay.  Behind the stores (round 6): the returning atomic is a ~2 us round trip that thread 0's wave used to sit out IN FRONT
            // of its share of step 6 (profiles/r06_v66_fused_tail_phases.txt); here it travels beside the stores' own completion
            if (tid == 0) {
                int* const depart = p.fin_counter + 128 + nt;
                const int tk = __hip_atomic_fetch_add(depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk == total - 1) {
                    __hip_atomic_store(depart, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(p.fin_counter + V2V_FIN_TAG_WORD + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if ((V2V_STAMP_MASK >> 7) & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // profiling build only: the stamp behind the stores' completion
            V2V_STAMP(p, 7);
            return;
        }
    }
    // ---- 16-byte stores through a wave-private LDS transposition (round 4) ----
    // The MFMA C layout gives a lane ONE channel of 16 rows: stored as it lies that is one 4-byte (fp32) or 2-byte (bf16) store
    // per element -- 32 store instructions per 32x32 block, and the stride-2 / transposed / fine-scale layers spent ~10 us of
    // every launch issuing them (profiles/r02_a67_s2_ablate_trace.txt; guide T21: an epilogue's store tail is ISSUE-bound).
    // Each wave now parks a 32x32 block (after bias / activation, fp32) in its own 4 KB of LDS -- lane (col, half) writes
    // row-major, conflict-free -- and reads it back 16 bytes of consecutive channels per lane (fp32: linear in the lane id,
    // conflict-free; bf16: 8 floats per lane), i.e. 4 (fp32) / 2 (bf16) vector stores per block instead of 32.  Wave-private: no
    // workgroup barrier, only the in-order LDS queue.  Same values, same statistics arithmetic as before (the statistics still
    // come from the registers in the old order); a vector is used only when it lies wholly inside the layer's own channels and
    // the output is 16-byte aligned -- otherwise (partial channel vectors, outputs written at an odd channel offset of a wider
    // tensor) the scalar stores remain.
    constexpr int RED_BYTES = ((WGM * BN * 8 + 1023) / 1024) * 1024;
    float* const tw = reinterpret_cast<float*>(smem + RED_BYTES) + (wm * WGN + wn) * 1024;
    const bool vec_ok = !helper && ((reinterpret_cast<unsigned long long>(p.out) & 15ull) == 0ull);
    if (p.out_mode == V2V_OUT_RAW_F32_NHWC) {
        float* const out = reinterpret_cast<float*>(p.out);
        const bool vec4 = vec_ok && (cs_out & 3u) == 0u;
        int spx[TM][4];                                // output pixel of the rows this lane STORES: (lane >> 3) + 8 k
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = pix_of(wm * WM + i * 32 + (lane >> 3) + 8 * k);
                spx[i][k] = (helper || (p.ablate & 4)) ? -1 : o;
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = nt * BN + wn * WN + j * 32 + lr;
            const bool nvalid = ncol < p.cout && !helper;
            const float bv = (p.bias != nullptr && nvalid) ? p.bias[ncol] : 0.f;
            const int vcol = nt * BN + wn * WN + j * 32 + 4 * (lane & 7);     // first channel of this lane's vectors
            const bool vfull = vec4 && vcol + 4 <= p.cout;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (fast) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r] + bv;
                        s1 += v;
                        s2 = __builtin_fmaf(v, v, s2);
                        tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v;
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] + bv;
                    if (opx[i][r] >= 0 && nvalid) {
                        s1 += v;
                        s2 = __builtin_fmaf(v, v, s2);          // explicit: the raw and the fused-norm paths must round alike
                    }
                    tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v;
                }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(tw + ((lane >> 3) + 8 * k) * 32 + 4 * (lane & 7));
                    if (spx[i][k] < 0) continue;
                    float* const dst = out + (unsigned)spx[i][k] * cs_out + (unsigned)vcol;
                    if (vfull) *reinterpret_cast<f32x4*>(dst) = v4;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (vcol + e < p.cout && !helper) dst[e] = v4[e];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the block is in registers before the next one overwrites it
            }
            if (want_stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (hi == 0 && !helper) {
                    const int c = wn * WN + j * 32 + lr;
                    red[(wm * BN + c) * 2 + 0] = s1;
                    red[(wm * BN + c) * 2 + 1] = s2;
                }
            }
        }
    } else if (p.out_mode == V2V_OUT_ACT_NHWC || p.out_mode == V2V_OUT_RAW_ACT_NHWC) {
        T* const out = reinterpret_cast<T*>(p.out);
        // V2V_OUT_RAW_ACT_NHWC: the pre-norm value (bias added, no activation) stored in T; the statistics below come from the fp32
        // accumulators in the order of the RAW_F32 branch, so the (sum, sum^2) rows are bit for bit those of mode 0
        const bool raw_t = p.out_mode == V2V_OUT_RAW_ACT_NHWC;
        // uniform epilogue operands pinned in SGPRs: the unrolled store loops otherwise re-fetch them from the kernel-argument
        // segment per element (profiles/r02_isa_sload_report.txt: 32-128 s_load_dword + s_waitcnt per launch)
        // (readfirstlane: a no-op for values that already live in scalar registers -- every shipped instantiation; the 7x7-window
        //  instantiation of the single-phase kernel is large enough that the argument block reaches here through vector registers)
        int e_act = __builtin_amdgcn_readfirstlane(raw_t ? V2V_ACT_NONE : p.act);
        float e_act_param = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(p.act_param)));
        int e_scale_bits = __builtin_amdgcn_readfirstlane(raw_t ? 0x3f800000 : __float_as_int(p.out_scale));     // (an integer select: a float one would be a VALU op)
        asm volatile("" : "+s"(e_act), "+s"(e_act_param), "+s"(e_scale_bits));
        const float e_out_scale = __int_as_float(e_scale_bits);
        // NONE / RELU / LEAKY (every norm-less hidden layer) as one select with hoisted conditions: bit for bit apply_act()
        const bool simple = e_act == V2V_ACT_NONE || e_act == V2V_ACT_RELU || e_act == V2V_ACT_LEAKY;
        const bool is_none = e_act == V2V_ACT_NONE, is_relu = e_act == V2V_ACT_RELU;
        constexpr int VEC = ElemTraits<T>::VEC;               // channels per 16-byte vector
        constexpr int LPR = 32 / VEC;                        // lanes per 32-channel row: 8 (fp32) / 4 (bf16)
        constexpr int RPP = 64 / LPR;                        // rows per pass of the wave: 8 / 16
        constexpr int NPASS = 32 / RPP;                      // 4 / 2
        const bool vecT = vec_ok && (cs_out % (unsigned)VEC) == 0u;
        int spx[TM][NPASS];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int o = pix_of(wm * WM + i * 32 + lane / LPR + RPP * k);
                spx[i][k] = (helper || (p.ablate & 4)) ? -1 : o;
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = nt * BN + wn * WN + j * 32 + lr;
            const bool nvalid = ncol < p.cout && !helper;
            const float bv = (p.bias != nullptr && nvalid) ? p.bias[ncol] : 0.f;
            const int vcol = nt * BN + wn * WN + j * 32 + VEC * (lane % LPR);
            const bool vfull = vecT && vcol + VEC <= p.cout;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (raw_t && want_stats && opx[i][r] >= 0 && nvalid) {
                        s1 += v;
                        s2 = __builtin_fmaf(v, v, s2);
                    }
                    if (simple) {
                        const float neg = is_relu ? 0.f : v * e_act_param;
                        v = (is_none || v > 0.f) ? v : neg;
                    } else {
                        v = apply_act(v, e_act, e_act_param);
                    }
                    tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lr] = v * e_out_scale;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NPASS; ++k) {
                    const float* const src = tw + (lane / LPR + RPP * k) * 32 + VEC * (lane % LPR);
                    float o[VEC];
#pragma unroll
                    for (int q = 0; q < VEC / 4; ++q) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(src + 4 * q);
                        o[4 * q + 0] = v4[0]; o[4 * q + 1] = v4[1]; o[4 * q + 2] = v4[2]; o[4 * q + 3] = v4[3];
                    }
                    if (spx[i][k] < 0) continue;
                    const unsigned eo = (unsigned)spx[i][k] * cs_out + (unsigned)vcol;
                    if (vfull) {
                        if constexpr (VEC == 4) {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + eo) = f32x4{o[0], o[1], o[2], o[3]};
                        } else {
                            typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
                            u32x4s pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
                            *reinterpret_cast<u32x4s*>(reinterpret_cast<unsigned short*>(out) + eo) = pk;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (vcol + e < p.cout && !helper) store_act(out, eo + e, o[e]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (raw_t && want_stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (hi == 0 && !helper) {
                    const int c = wn * WN + j * 32 + lr;
                    red[(wm * BN + c) * 2 + 0] = s1;
                    red[(wm * BN + c) * 2 + 1] = s2;
                }
            }
        }
    } else {                                       // planar fp32 NCHW (API-facing heads)
        float* const out = reinterpret_cast<float*>(p.out);
        int e_act = __builtin_amdgcn_readfirstlane(p.act);
        float e_act_param = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(p.act_param)));
        float e_out_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(p.out_scale)));
        asm volatile("" : "+s"(e_act), "+s"(e_act_param), "+s"(e_out_scale));
        const unsigned ohow = (unsigned)(p.OH * p.OW);
        if (p.N > 1) {                             // pixel index -> n * cout * OH*OW + pixel-in-image
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (opx[i][r] >= 0) {
                        const unsigned n = (unsigned)opx[i][r] / ohow;
                        opx[i][r] = (int)(n * (unsigned)p.cout * ohow + ((unsigned)opx[i][r] - n * ohow));
                    }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = nt * BN + wn * WN + j * 32 + lr;
            const bool nvalid = ncol < p.cout && !helper;
            const float bv = (p.bias != nullptr && nvalid) ? p.bias[ncol] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (opx[i][r] >= 0 && nvalid)
                        out[(unsigned)opx[i][r] + (unsigned)ncol * ohow] = apply_act(acc[i][j][r] + bv, e_act, e_act_param) * e_out_scale;
        }
    }
    V2V_STAMP(p, 4);
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
                float* dst = p.stats + ((long long)stat_row * p.cout + ncol) * 2;
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
            V2V_STAMP(p, 5);
            int* flag = reinterpret_cast<int*>(smem + 16384);
            const int total = p.fin_rows > 0 ? p.fin_rows : (int)gridDim.y * p.m_tiles;
            constexpr int NT = NW * 64;
            if (p.fin_groups > 0) {
                // ---- two levels (round 4): layers with more than 512 rows (the fine scales of 2048x1024: 2048-16384 rows) ----
                // Level 1: the rows are cut into G groups exactly as v2v_bn_finalize cuts them (group g = rows [total g / G,
                // total (g+1) / G)); the LAST workgroup of a group to publish its row reduces the group with the arithmetic of
                // bn_partial_reduce_kernel (4 row phases per channel in fp64, combined ((p0+p1)+p2)+p3) into an fp64 row of fin_ws.
                // Level 2: the last group to finish reduces the G group rows with the arithmetic of bn_finalize_kernel<double> and
                // writes the scale / shift record.  Same bits as the two launches it replaces, whichever workgroup happens to be last.
                const int G = p.fin_groups;
                int g = (int)(((long long)stat_row * G) / total);
                while ((int)(((long long)total * (g + 1)) / G) <= stat_row) ++g;
                while ((int)(((long long)total * g) / G) > stat_row) --g;
                const int r0 = (int)(((long long)total * g) / G), r1 = (int)(((long long)total * (g + 1)) / G);
                if (tid == 0) {
                    int* cnt = p.fin_counter + 256 + g * p.n_tiles + nt;
                    const int tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = tk == (r1 - r0) - 1 ? 1 : 0;
                    if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *flag = last;
                }
                __syncthreads();
                if (!*flag) { V2V_STAMP(p, 6); return; }
                __syncthreads();
                double* const sh = reinterpret_cast<double*>(smem);          // [4][BN][2] fp64, <= 8 KiB
                constexpr int CP = NT / 4 < BN ? NT / 4 : BN;                // channels per pass (4 row phases each)
                for (int cb = 0; cb < BN; cb += CP) {
                    const int cl = cb + tid % CP, ph = tid / CP;
                    const int ncol = nt * BN + cl;
                    if (tid < 4 * CP) {
                        double s1 = 0.0, s2 = 0.0;
                        if (ncol < p.cout) {
                            const unsigned long long* const base = reinterpret_cast<const unsigned long long*>(p.stats) + ncol;
#pragma unroll 8
                            for (int r = r0 + ph; r < r1; r += 4) {
                                const unsigned long long b = __hip_atomic_load(base + (long long)r * p.cout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                s1 += (double)__uint_as_float((unsigned)(b & 0xffffffffull));
                                s2 += (double)__uint_as_float((unsigned)(b >> 32));
                            }
                        }
                        sh[(ph * BN + cl) * 2 + 0] = s1;
                        sh[(ph * BN + cl) * 2 + 1] = s2;
                    }
                    __syncthreads();
                    if (tid < CP && ncol < p.cout) {
                        const double t1 = ((sh[(0 * BN + cl) * 2] + sh[(1 * BN + cl) * 2]) + sh[(2 * BN + cl) * 2]) + sh[(3 * BN + cl) * 2];
                        const double t2 = ((sh[(0 * BN + cl) * 2 + 1] + sh[(1 * BN + cl) * 2 + 1]) + sh[(2 * BN + cl) * 2 + 1]) + sh[(3 * BN + cl) * 2 + 1];
                        unsigned long long* const dst = reinterpret_cast<unsigned long long*>(p.fin_ws + ((long long)g * p.cout + ncol) * 2);
                        __hip_atomic_store(dst, (unsigned long long)__double_as_longlong(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, (unsigned long long)__double_as_longlong(t2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    const int tk = __hip_atomic_fetch_add(p.fin_counter + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = tk == G - 1 ? 1 : 0;
                    if (last) __hip_atomic_store(p.fin_counter + nt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *flag = last;
                }
                __syncthreads();
                if (!*flag) { V2V_STAMP(p, 6); return; }
                __syncthreads();
                for (int cb = 0; cb < BN; cb += CP) {
                    const int cl = cb + tid % CP, ph = tid / CP;
                    const int ncol = nt * BN + cl;
                    if (tid < 4 * CP) {
                        double s1 = 0.0, s2 = 0.0;
                        if (ncol < p.cout) {
                            const unsigned long long* const base = reinterpret_cast<const unsigned long long*>(p.fin_ws) + (long long)ncol * 2;
#pragma unroll 8
                            for (int r = ph; r < G; r += 4) {
                                const unsigned long long b1 = __hip_atomic_load(base + (long long)r * p.cout * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned long long b2 = __hip_atomic_load(base + (long long)r * p.cout * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                s1 += __longlong_as_double((long long)b1);
                                s2 += __longlong_as_double((long long)b2);
                            }
                        }
                        sh[(ph * BN + cl) * 2 + 0] = s1;
                        sh[(ph * BN + cl) * 2 + 1] = s2;
                    }
                    __syncthreads();
                    if (tid < CP && ncol < p.cout) {
                        const double s1 = ((sh[(0 * BN + cl) * 2] + sh[(1 * BN + cl) * 2]) + sh[(2 * BN + cl) * 2]) + sh[(3 * BN + cl) * 2];
                        const double s2 = ((sh[(0 * BN + cl) * 2 + 1] + sh[(1 * BN + cl) * 2 + 1]) + sh[(2 * BN + cl) * 2 + 1]) + sh[(3 * BN + cl) * 2 + 1];
                        const double mean = s1 * p.fin_inv_count;
                        double var = s2 * p.fin_inv_count - mean * mean;
                        if (var < 0.0) var = 0.0;
                        const double invstd = 1.0 / sqrt(var + (double)p.fin_eps);
                        const double gm = p.fin_gamma ? (double)p.fin_gamma[ncol] : 1.0;
                        const double bt = p.fin_beta ? (double)p.fin_beta[ncol] : 0.0;
                        const double sc = gm * invstd;
                        p.fin_out[ncol] = (float)sc;
                        p.fin_out[p.cout + ncol] = (float)(bt - mean * sc);
                        p.fin_out[2 * p.cout + ncol] = (float)mean;
                        p.fin_out[3 * p.cout + ncol] = (float)invstd;
                        if (p.fin_rmean) p.fin_rmean[ncol] = (1.f - p.fin_momentum) * p.fin_rmean[ncol] + p.fin_momentum * (float)mean;
                        if (p.fin_rvar)  p.fin_rvar[ncol]  = (1.f - p.fin_momentum) * p.fin_rvar[ncol] + p.fin_momentum * (float)(var * p.fin_unbias);
                    }
                    __syncthreads();
                }
                V2V_STAMP(p, 6);
                return;
            }
            if (tid == 0) {
                const int tk = __hip_atomic_fetch_add(p.fin_counter + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = tk == total - 1 ? 1 : 0;
                if (last) __hip_atomic_store(p.fin_counter + nt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
                *flag = last;
            }
            __syncthreads();
            if (*flag) {
                constexpr int PH = NT / BN;
                double* acc2 = reinterpret_cast<double*>(smem);      // [PH][BN][2], <= 8 KiB
                const int c = tid % BN, ph = tid / BN;
                const int ncol = nt * BN + c;
                double s1 = 0.0, s2 = 0.0;
                if (ncol < p.cout && ph < PH) sum_stat_rows<16>(p.stats, (long long)ncol * 2, (long long)p.cout * 2, ph, PH, total, s1, s2);
                if (ph < PH) {
                    acc2[(ph * BN + c) * 2 + 0] = s1;
                    acc2[(ph * BN + c) * 2 + 1] = s2;
                }
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
    V2V_STAMP(p, 6);
}

template <typename T, int BM, int BN, int WGM, int WGN, int NS, bool HELPER>
__global__ __launch_bounds__((WGM * WGN + (HELPER ? 1 : 0)) * 64) void conv_igemm_kernel(const ConvKArgs p) {
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
    if (p.ablate & 512) return;               // ablation: the launch itself (dispatch of the grid with this LDS / wave footprint)
    V2V_STAMP(p, 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = HELPER && wid >= NW;   // prefetch wave (HELPER instances are launched with NW+1 waves)
    const int wm = wid / WGN, wn = wid % WGN;
    // heaviest parity class first (ConvTranspose2d 3x3 / stride 2: classes of 1, 2, 2, 4 taps; workgroups are dispatched in
    // blockIdx order, so the long ones must not start last)
    const int cls = p.cls_rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;

    const int tiles = p.m_tiles * p.n_tiles;
    const int S = p.splitk;
    const int lin_all = xcd_remap(blockIdx.x, tiles * S);   // a tile's K slices are neighbours on one XCD
    const int lin = lin_all / S;
    const int slice = lin_all - lin * S;
    const int nt = lin / p.m_tiles;
    const int mt = lin - nt * p.m_tiles;

    const int nkh = p.nkh[cls], nkw = p.nkw[cls];
    const int dh0 = p.dh0[cls], dw0 = p.dw0[cls], dstep = p.dstep;
    const int kpad = p.kpad[cls];
    const int nk_all = kpad / BKE;
    const int kb = (int)(((long long)nk_all * slice) / S);            // this slice's first K chunk
    const int nk = (p.ablate & 1024) ? 0 : (int)(((long long)nk_all * (slice + 1)) / S) - kb; // and its chunk count (ablation 1024: no main loop)
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
        long long r = (long long)nt * BN + lrow + LR * i;
        r = r < p.cout_p ? r : p.cout_p - 1;         // BN = 256 tiles can overhang the packed rows (cout_p = 128-multiple)
        wp[i] = p.w + ((long long)p.woff[cls] + r * p.wrow[cls] + koff) * (long long)sizeof(T) + (long long)kb * 128;
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
    if (kb > 0) {                                    // split-K: start the uniform walk at chunk kb
        const int e0 = kb * BKE, tap0 = e0 / cs;
        kcb = (e0 - tap0 * cs) * (int)sizeof(T);
        tap_h = tap0 / nkw;
        tap_w = tap0 - tap_h * nkw;
    }
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
        const int k0 = koff + kb * BKE;
        const int t = k0 / cs;
        kc = k0 - t * cs;
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
                const char* src = (((aok >> i) & 1u) && !(p.ablate & 1)) ? ap[i] + kcb : zp;
                glds16(src, sbase + i * LR * 128);
            }
        } else {
            const bool kvalid = kth < nkh;
            const int dh = dh0 + kth * dstep, dw = dw0 + ktw * dstep;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                bool ok = kvalid && ((rowvalid >> i) & 1u);
                const char* src = tap_ptr(i, dh, dw, ok) + kc * (int)sizeof(T);
                src = (ok && !(p.ablate & 1)) ? src : zp;
                glds16(src, sbase + i * LR * 128);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            glds16(wp[i] + ((p.ablate & 2) ? 0ll : (long long)issued * 128), sbase + BM * 128 + i * LR * 128);
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
    if (helper) {
        // Prefetch wave: lane l touches weight row (nt*BN + l [+64 ...]) of chunk c with a 4-byte LDS-DMA into
        // a dummy LDS line (no VGPR destination, nothing ever waits for it inside the loop); it only keeps
        // step with the workgroup through the per-chunk barrier.
        constexpr int PR = (BN + 63) / 64;
        const bool pf_on = (mt & p.pf_mask) == 0;
        char* const sink = smem + NS * STAGE;
        const char* prow[PR];
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            long long r = (long long)nt * BN + lane + 64 * i;
            r = r < p.cout_p ? r : p.cout_p - 1;
            prow[i] = p.w + ((long long)p.woff[cls] + r * p.wrow[cls]) * (long long)sizeof(T) + (long long)kb * 128;
        }
        const int P = p.pf_dist;
        if (pf_on)
            for (int c = 0; c < P && c < nk; ++c)
#pragma unroll
                for (int i = 0; i < PR; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prow[i] + (long long)c * 128),
                                                     (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
        for (int ks = 0; ks < nk; ++ks) {
            __builtin_amdgcn_s_barrier();
            if (pf_on && ks + P < nk)
#pragma unroll
                for (int i = 0; i < PR; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prow[i] + (long long)(ks + P) * 128),
                                                     (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
        }
    } else {
        for (int t = 0; t < D && t < nk; ++t) issue();
        V2V_STAMP(p, 1);
        for (int ks = 0; ks < nk; ++ks) {
            // tile ks must have landed; in steady state tiles ks+1 .. ks+D-1 stay in flight
            if (ks + D <= nk) wait_vmcnt<LPT * (D - 1)>();
            else              wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();     // every wave's part of tile ks is in LDS, and every wave
                                              // is done reading stage (ks-1)%NS, which is refilled now
            if (ks == 0) V2V_STAMP(p, 2);
            if (ks + D < nk) issue();
            const char* sb = smem + (ks % NS) * STAGE;
            if (p.ablate & 16) continue;      // loader-only ablation
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Frag fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const Frag*>(sb + a_row_off + i * 32 * 128 + foff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const Frag*>(sb + b_row_off + j * 32 * 128 + foff[s]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
            }
        }
    }
    mfma_results_ready();
    __syncthreads();                      // LDS ring is free: reused for the statistics reduction
    V2V_STAMP(p, 3);

    // row -> output pixel.  The uniform operands are pinned in SGPRs: left to itself the compiler re-fetches them from the kernel
    // argument segment for EVERY row of the unrolled epilogue (s_load + s_waitcnt lgkmcnt(0), 1-4 per row, 133 in a 128 x 128 tile)
    // and rebuilds the division reciprocals each time -- 7-13 us of every launch with neither main loop nor stores
    // (profiles/r02_a72_fixed_cost_trace_nofin.txt)
    int e_os = p.os, e_mcls = p.Mc[cls], e_owc = p.OWc[cls], e_hwc = p.OHc[cls] * p.OWc[cls], e_oh = p.OH, e_ow = p.OW;
    int e_m0 = mt * BM;
    unsigned e_mh = p.div_m[cls][0], e_mw = p.div_m[cls][1]; int e_lh = p.div_l[cls][0], e_lw = p.div_l[cls][1];
    asm volatile("" : "+s"(e_os), "+s"(e_mcls), "+s"(e_owc), "+s"(e_hwc), "+s"(e_oh), "+s"(e_ow), "+s"(e_m0));
    asm volatile("" : "+s"(e_mh), "+s"(e_mw), "+s"(e_lh), "+s"(e_lw));
    conv_epilogue<T, BM, BN, WGM, WGN>(p, acc, smem, tid, wm, wn, helper, cls, tiles, lin, slice, S, nt, cls * p.m_tiles + mt,
        [&](int row) -> int {                  // N*OH*OW < 2^31 (host check)
            const int m = e_m0 + row;
            if (m >= e_mcls) return -1;
            if (e_os == 1) return m;
            const int n = (int)((__umulhi(e_mh, (unsigned)m) + (unsigned)m) >> e_lh);       // m / e_hwc, exact for 0 <= m < 2^31
            const int rem = m - n * e_hwc;
            const int oi = (int)((__umulhi(e_mw, (unsigned)rem) + (unsigned)rem) >> e_lw);  // rem / e_owc
            const int oj = rem - oi * e_owc;
            return (n * e_oh + (oi * 2 + (cls >> 1))) * e_ow + (oj * 2 + (cls & 1));
        });
}

// ---- conv launch ------------------------------------------------------------------------
template <typename T, int BM, int BN, int WGM, int WGN, int NS, bool HELPER>
static int launch_cfg(const ConvKArgs& k, int ncls, hipStream_t s) {
    const size_t lds = (size_t)NS * (BM + BN) * 128 + (HELPER ? 256 : 0);    // + the prefetch wave's dummy LDS line (round 4: only where that wave exists -- tile 10 is then exactly 32 KiB and fits beside a 128 KiB single-phase workgroup on one CU)
    auto kern = conv_igemm_kernel<T, BM, BN, WGM, WGN, NS, HELPER>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(k.m_tiles * k.n_tiles * k.splitk), (unsigned)ncls);
    hipLaunchKernelGGL(kern, grid, dim3((WGM * WGN + (HELPER ? 1 : 0)) * 64), lds, s, k);
    return check_launch();
}

template <typename T>
static inline int launch_typed(int cfg, const ConvKArgs& k, int ncls, hipStream_t s) {
    switch (cfg) {
        case 1: return launch_cfg<T, 128, 128, 2, 2, 3, false>(k, ncls, s);   //  96 KiB LDS
        case 2: return k.pf_dist > 0 ? launch_cfg<T, 128, 64, 2, 2, 4, true>(k, ncls, s) : launch_cfg<T, 128, 64, 2, 2, 4, false>(k, ncls, s);    //  96 KiB
        case 3: return k.pf_dist > 0 ? launch_cfg<T, 64, 64, 2, 2, 4, true>(k, ncls, s) : launch_cfg<T, 64, 64, 2, 2, 4, false>(k, ncls, s);     //  64 KiB (2 workgroups / CU)
        case 4: return launch_cfg<T, 128, 32, 4, 1, 4, false>(k, ncls, s);    //  80 KiB (2 / CU)
        case 5: return k.pf_dist > 0 ? launch_cfg<T, 64, 128, 2, 2, 4, true>(k, ncls, s) : launch_cfg<T, 64, 128, 2, 2, 4, false>(k, ncls, s);    //  96 KiB
        case 6: return launch_cfg<T, 256, 64, 4, 1, 3, false>(k, ncls, s);    // 120 KiB
        case 7: return k.pf_dist > 0 ? launch_cfg<T, 128, 64, 2, 2, 6, true>(k, ncls, s) : launch_cfg<T, 128, 64, 2, 2, 6, false>(k, ncls, s);    // 144 KiB
        case 8: return launch_cfg<T, 128, 128, 2, 2, 4, false>(k, ncls, s);   // 128 KiB
        case 9: return k.pf_dist > 0 ? launch_cfg<T, 64, 64, 2, 2, 3, true>(k, ncls, s) : launch_cfg<T, 64, 64, 2, 2, 3, false>(k, ncls, s);     //  48 KiB (3 workgroups / CU)
        case 10: return launch_cfg<T, 64, 64, 2, 2, 2, false>(k, ncls, s);    //  32 KiB (5 / CU)
        case 11: return k.pf_dist > 0 ? launch_cfg<T, 128, 64, 2, 2, 2, true>(k, ncls, s) : launch_cfg<T, 128, 64, 2, 2, 2, false>(k, ncls, s);   //  48 KiB (3 / CU)
        case 12: return k.pf_dist > 0 ? launch_cfg<T, 64, 128, 2, 2, 3, true>(k, ncls, s) : launch_cfg<T, 64, 128, 2, 2, 3, false>(k, ncls, s);   //  72 KiB (2 / CU)
        case 13: return k.pf_dist > 0 ? launch_cfg<T, 128, 64, 4, 2, 3, true>(k, ncls, s) : launch_cfg<T, 128, 64, 4, 2, 3, false>(k, ncls, s);   //  72 KiB, 8 waves (2 / CU)
        case 14: return launch_cfg<T, 128, 128, 4, 2, 2, false>(k, ncls, s);  //  64 KiB, 8 waves (2 / CU)
        case 15: return launch_cfg<T, 128, 128, 2, 4, 3, false>(k, ncls, s);  //  96 KiB, 8 waves
        case 16: return launch_cfg<T, 256, 64, 4, 2, 2, false>(k, ncls, s);   //  80 KiB, 8 waves
        case 17: return k.pf_dist > 0 ? launch_cfg<T, 64, 128, 2, 4, 3, true>(k, ncls, s) : launch_cfg<T, 64, 128, 2, 4, 3, false>(k, ncls, s);   //  72 KiB, 8 waves (2 / CU)
        case 18: return launch_cfg<T, 256, 128, 4, 2, 3, false>(k, ncls, s);  // 144 KiB, 8 waves, wave tile 64x64
        case 19: return launch_cfg<T, 256, 128, 2, 2, 3, false>(k, ncls, s);  // 144 KiB, 4 waves, wave tile 128x64
        case 20: return launch_cfg<T, 128, 256, 2, 4, 3, false>(k, ncls, s);  // 144 KiB, 8 waves, wave tile 64x64
        case 21: return launch_cfg<T, 128, 128, 2, 2, 2, false>(k, ncls, s);  //  64 KiB, 4 waves, wave tile 64x64 (2 / CU)
        case 22: return launch_cfg<T, 256, 128, 4, 2, 2, false>(k, ncls, s);  //  96 KiB, 8 waves, wave tile 64x64
        case 23: return launch_cfg<T, 128, 256, 2, 2, 3, false>(k, ncls, s);  // 144 KiB, 4 waves, wave tile 64x128
    }
    set_error("conv: unknown tile config %d", cfg);
    return V2V_EINVAL;
}


static inline bool cfg_has_helper_impl(int cfg) {
    switch (cfg) { case 2: case 3: case 5: case 7: case 9: case 11: case 12: case 13: case 17: return true; }
    return false;
}

}  // namespace v2v
