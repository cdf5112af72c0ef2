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
//     (N) tiles in that XCD's private L2;
//   * weight stream: at batch 1 every weight line is an HBM miss that all M tiles of an N column
//     wait for together, and the in-order vmcnt makes the (L2-hitting) activation rows wait behind
//     it.  An optional HELPER wave per workgroup (no MFMA work, never waits) touches the weight
//     lines `pf_dist` chunks ahead with 4-byte LDS-DMA reads into a dummy LDS line, so the real
//     loaders' weight fetches are L2 hits;
//   * split-K: small-M layers (M = 2048 pixels at 512x256) cannot fill 256 CUs with tiles large
//     enough to be LDS-read efficient (wave tile >= 64x64); `splitk` workgroups share a tile,
//     publish fp32 partial tiles write-through (sc1) and the last arriver reduces them in slice
//     order (deterministic) and runs the normal epilogue.
#include "conv3x3_pp_kernel.h"
#include "conv3x3_pp2_kernel.h"
#include "conv3x3_pp3_kernel.h"
#include "conv3x3_s2_kernel.h"
#include "conv3x3_t2_kernel.h"
#include "conv7x7_head_kernel.h"
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <algorithm>

namespace v2v {

// -------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------
struct TileCfg { int id, BM, BN; };
static const TileCfg kCfgs[] = {
    {1, 128, 128}, {2, 128, 64}, {3, 64, 64}, {4, 128, 32}, {5, 64, 128}, {6, 256, 64},
    {7, 128, 64}, {8, 128, 128},      // deeper LDS-DMA rings of 2 / 1
    {9, 64, 64}, {10, 64, 64}, {11, 128, 64}, {12, 64, 128},   // occupancy / depth variants of 3, 2, 5
    {13, 128, 64}, {14, 128, 128}, {15, 128, 128}, {16, 256, 64}, {17, 64, 128},   // 8-wave workgroups
    // wave tiles >= 64x64 (LDS-read efficient); meant to be combined with split-K on small-M layers
    {18, 256, 128}, {19, 256, 128}, {20, 128, 256}, {21, 128, 128}, {22, 256, 128}, {23, 128, 256},
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
    int wrow[4];         // row stride of the class matrix, elements (>= kpad)
    long long woff[4];
    long long total;
    int cout_p;
};

// Row stride of a packed weight matrix.  Rows whose byte length is an even number of 128-byte lines put the same
// K chunk of every row of a tile at the same line index modulo a power of two: every workgroup of an XCD then
// requests a chunk's 64-256 weight lines from the same few L2 channels at the same time.  One extra (zero) line
// makes the line stride odd, which spreads consecutive rows over all channels.  Measured (profiles/r01_v4_ablate_*):
// no effect on gfx950 (the L2 channel hash already spreads 2 KiB / 18 KiB strides), so it is OFF unless V2V_WPAD=1.
static int weight_row_stride(int kpad, int dtype) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("V2V_WPAD"); on = (e && e[0] == '1') ? 1 : 0; }
    const int bke = dtype == V2V_BF16 ? 64 : 32;
    const int lines = kpad / bke;
    return (on && lines > 1 && (lines & 1) == 0) ? kpad + bke : kpad;
}

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
        g->wrow[0] = weight_row_stride(g->kpad[0], dtype);
        g->woff[0] = 0;
        off = (long long)g->cout_p * g->wrow[0];
    } else {
        g->ncls = stride == 1 ? 1 : 4;
        for (int c = 0; c < g->ncls; ++c) {
            const int a = c >> 1, b = c & 1;
            convt_axis(KH, pad, stride, a, &g->kh0[c], &g->nkh[c], &g->dh0[c]);
            convt_axis(KW, pad, stride, b, &g->kw0[c], &g->nkw[c], &g->dw0[c]);
            g->ktot[c] = g->nkh[c] * g->nkw[c] * cin_stride;
            g->kpad[c] = (int)round_up(g->ktot[c] > 0 ? g->ktot[c] : 1, bke);
            g->wrow[c] = weight_row_stride(g->kpad[c], dtype);
            g->woff[c] = off;
            off += (long long)g->cout_p * g->wrow[c];
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
    int nkh[4], nkw[4], kh0[4], kw0[4], kpad[4], wrow[4];
    long long woff[4];
    long long total;
    int dtype;
    int korder, bke;     // korder 1: k = (chunk*ntaps + tap)*bke + c_in_chunk  (channel-chunk outer, tap inner)
    int src_cl;          // the fp32 source is channels-last: [dim0][KH][KW][dim1] (optim.FlatBuffers master weights)
};

__global__ void pack_weights_kernel(const PackArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.total; e += stride) {
        int cls = 0;
#pragma unroll
        for (int c = 1; c < 4; ++c)
            if (c < a.ncls && e >= a.woff[c]) cls = c;
        const long long le = e - a.woff[cls];
        const int kp = a.wrow[cls];
        const int co = (int)(le / kp);
        const int k = (int)(le - (long long)co * kp);      // k >= kpad: the stride-padding line, zero
        const int ntaps = a.nkh[cls] * a.nkw[cls];
        int t, c;
        if (a.korder == 1) {
            const int chunk = k / (ntaps * a.bke), rem = k - chunk * ntaps * a.bke;
            t = rem / a.bke;
            c = chunk * a.bke + (rem - t * a.bke);
            if (k >= a.kpad[cls]) t = ntaps;             // stride-padding line
        } else {
            t = k / a.cin_stride;
            c = k - t * a.cin_stride;
        }
        float v = 0.f;
        if (co < a.cout && t < ntaps && c < a.cin) {
            const int th = t / a.nkw[cls], tw = t - th * a.nkw[cls];
            int kh, kw;
            if (a.transposed) { kh = a.kh0[cls] + a.kstep * th; kw = a.kw0[cls] + a.kstep * tw; }
            else              { kh = th; kw = tw; }
            long long src;
            if (a.src_cl) src = a.transposed ? (((long long)c * a.KH + kh) * a.KW + kw) * a.cout + co       // [cin][kh][kw][cout]
                                             : (((long long)co * a.KH + kh) * a.KW + kw) * a.cin + c;      // [cout][kh][kw][cin]
            else if (a.transposed) src = (((long long)c * a.cout + co) * a.KH + kh) * a.KW + kw;   // [cin][cout][kh][kw]
            else                   src = (((long long)co * a.cin + c) * a.KH + kh) * a.KW + kw;    // [cout][cin][kh][kw]
            v = a.w[src];
        }
        if (a.dtype == V2V_BF16) reinterpret_cast<unsigned short*>(a.dst)[e] = f32_to_bf16_bits(v);
        else                     reinterpret_cast<float*>(a.dst)[e] = v;
    }
}

// Tiled variant for kernels of up to 16 taps (1x1, 3x3, 4x4: everything but the 7x7 stems / heads).  The kernel above
// walks the DESTINATION and gathers 4 bytes every KH*KW*4 bytes per lane with two 64-bit divisions per element: 28 us per
// call, 353 calls = 7.3 ms = 10 % of a training step (profiles/r02_a14_train_kernel_stats.txt), since every optimizer
// step re-packs every layer for its forward and backward-data operators.  Here a workgroup owns 8 rows x 64 channels
// x all taps: the source runs are contiguous ([c0..c0+63][KH][KW] of a row for a Conv2d weight, [co0..co0+7][KH][KW] of
// a channel for a ConvTranspose2d-layout read), go through LDS, and leave as 64-element contiguous runs per (row, tap).
constexpr int PK_MAXT = 16;

// PK_TCO x PK_TC: 8 rows x 64 channels (sources whose channel / tap run is the contiguous one), or 32 x 32 for the transposed
// read of a channels-last source, whose contiguous run is along the ROWS
template <int PK_TCO, int PK_TC>
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const PackArgs a) {
    extern __shared__ float sh[];                            // [PK_TCO * KHW][PK_TC + KHW]
    const int c0 = blockIdx.x * PK_TC, co0 = blockIdx.y * PK_TCO;
    const int KHW = a.KH * a.KW;
    const int LS = PK_TC + KHW;                              // LDS row stride: bank = tap * KHW + channel, distinct over a wave's run
    const int tid = threadIdx.x;
    // ---- read: contiguous source runs (16-byte loads) -> sh[(row * KHW + full tap) * LS + channel]
    if (a.src_cl && !a.transposed) {
        // channels-last master weight [co][tap][c]: per (row, tap) the tile's 64 channels are one contiguous run
        const int nc = min(PK_TC, a.cin - c0);
        for (int i = tid; i < PK_TCO * KHW * PK_TC; i += 256) {
            const int cl = i % PK_TC, rf = i / PK_TC;
            const int r = rf / KHW, f = rf - r * KHW;
            const int co = co0 + r;
            if (co < a.cout && cl < nc) sh[(r * KHW + f) * LS + cl] = a.w[((long long)co * KHW + f) * a.cin + c0 + cl];
        }
    } else if (a.src_cl) {
        // transposed read of a channels-last weight [c][tap][co]: per (channel, tap) the tile's rows are one contiguous run
        const int nr = min(PK_TCO, a.cout - co0);
        for (int i = tid; i < PK_TC * KHW * PK_TCO; i += 256) {
            const int r = i % PK_TCO, cf = i / PK_TCO;
            const int cl = cf / KHW, f = cf - cl * KHW;
            const int c = c0 + cl;
            if (c < a.cin && r < nr) sh[(r * KHW + f) * LS + cl] = a.w[((long long)c * KHW + f) * a.cout + co0 + r];
        }
    } else if (!a.transposed) {
        // All of a thread's 16-byte loads are issued BEFORE the first LDS store (round 4): the row-by-row loop of the first
        // version waited for one global load per row -- 8 serialised memory latencies per workgroup, 43 us for a
        // 1024 x 1024 x 3 x 3 layer that moves 56 MB (profiles/r04_a6_train_kernel_stats.txt: 6.6 % of the training step).
        const int nc = min(PK_TC, a.cin - c0);
        const int run = nc > 0 ? nc * KHW : 0;
        const int nrow = min(PK_TCO, a.cout - co0);
        const int Q = (run + 3) >> 2;                          // 16-byte pieces per row
        const int totq = nrow > 0 ? nrow * Q : 0;
        constexpr int MAXQ = PK_TCO * PK_MAXT * PK_TC / 4 / 256;
        const float* const src0 = a.w + ((long long)co0 * a.cin + c0) * KHW;
        const long long rstride = (long long)a.cin * KHW;
        const bool al = ((((unsigned long long)src0) | ((unsigned long long)rstride * 4ull)) & 15) == 0;
        float4 v[MAXQ];
#pragma unroll
        for (int u = 0; u < MAXQ; ++u) {
            const int j = tid + u * 256;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < totq) {
                const int r = j / Q, i4 = (j - r * Q) * 4;
                const float* src = src0 + r * rstride + i4;
                if (al && i4 + 4 <= run) v[u] = *reinterpret_cast<const float4*>(src);
                else {
                    v[u].x = src[0];
                    if (i4 + 1 < run) v[u].y = src[1];
                    if (i4 + 2 < run) v[u].z = src[2];
                    if (i4 + 3 < run) v[u].w = src[3];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MAXQ; ++u) {
            const int j = tid + u * 256;
            if (j < totq) {
                const int r = j / Q, i4 = (j - r * Q) * 4;
                int cl = i4 / KHW, f = i4 - cl * KHW;
                const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i4 + q < run) sh[(r * KHW + f) * LS + cl] = vv[q];
                    if (++f == KHW) { f = 0; ++cl; }
                }
            }
        }
    } else {
        // transposed read ([c][co][kh][kw] source): per channel the tile's rows x taps are one contiguous run of nr * KHW floats;
        // 8 independent loads per thread in flight per batch (the first version: one load per thread and pass, 8-32 passes)
        const int nr = min(PK_TCO, a.cout - co0);
        const int run = nr > 0 ? nr * KHW : 0;
        const int ncl = min(PK_TC, a.cin - c0);
        const int tot = ncl > 0 ? ncl * run : 0;
        for (int base = 0; base < tot; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = base + u * 256 + tid;
                v[u] = 0.f;
                if (j < tot) {
                    const int cl = j / run, i = j - cl * run;
                    v[u] = a.w[((long long)(c0 + cl) * a.cout + co0) * KHW + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = base + u * 256 + tid;
                if (j < tot) {
                    const int cl = j / run, i = j - cl * run;
                    const int r = i / KHW, f = i - r * KHW;
                    sh[(r * KHW + f) * LS + cl] = v[u];
                }
            }
        }
    }
    __syncthreads();
    // ---- write: per class, 8-channel vectors (one 16-byte store for bf16, two for fp32) per (row, tap)
    for (int cls = 0; cls < a.ncls; ++cls) {
        const int nkw = a.nkw[cls], nt = a.nkh[cls] * nkw;
        const int wrow = a.wrow[cls];
        const int n = PK_TCO * nt * (PK_TC / 8);
        for (int j = tid; j < n; j += 256) {
            const int c8 = (j % (PK_TC / 8)) * 8;
            const int rt = j / (PK_TC / 8);
            const int r = rt / nt, t = rt - r * nt;
            const int co = co0 + r, c = c0 + c8;
            if (co >= a.cout_p || c >= a.cin_stride) continue;
            const int th = t / nkw, tw = t - th * nkw;
            const int kh = a.transposed ? a.kh0[cls] + a.kstep * th : th;
            const int kw = a.transposed ? a.kw0[cls] + a.kstep * tw : tw;
            const float* sp = sh + (r * KHW + kh * a.KW + kw) * LS + c8;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (co < a.cout && c + q < a.cin) ? sp[q] : 0.f;
            // cin_stride is a multiple of 8 (bf16) / 4 (fp32) and bke of 64 / 32: the 8 (4 + 4) channels stay inside one chunk
            const int k = a.korder == 1 ? ((c / a.bke) * nt + t) * a.bke + (c % a.bke) : t * a.cin_stride + c;
            const long long e = a.woff[cls] + (long long)co * wrow + k;
            if (a.dtype == V2V_BF16) {
                uint4 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                pk.z = pack_bf16x2(v[4], v[5]);
                pk.w = pack_bf16x2(v[6], v[7]);
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.dst) + e) = pk;
            } else {
                float* d = reinterpret_cast<float*>(a.dst) + e;
                *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                if (c + 4 < a.cin_stride) *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        if (blockIdx.x == 0) {                               // the zero lines behind a row's taps (K padded to the tile depth)
            const int ktot = nt * a.cin_stride, npad = wrow - ktot;
            for (int j = tid; j < PK_TCO * npad; j += 256) {
                const int r = j / npad, k = ktot + (j - r * npad);
                if (co0 + r >= a.cout_p) continue;
                const long long e = a.woff[cls] + (long long)(co0 + r) * wrow + k;
                if (a.dtype == V2V_BF16) reinterpret_cast<unsigned short*>(a.dst)[e] = 0;
                else                     reinterpret_cast<float*>(a.dst)[e] = 0.f;
            }
        }
    }
}

struct PackOp : Op {
    PackArgs a;
    int launch(hipStream_t s) override {
        if (a.KH * a.KW <= PK_MAXT) {
            const int khw = a.KH * a.KW;
            if (a.src_cl && a.transposed) {
                const size_t lds = (size_t)32 * khw * (32 + khw) * sizeof(float);
                auto kern = pack_weights_tiled_kernel<32, 32>;
                static bool attr_done = false;
                if (!attr_done) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr_done = true; }
                const dim3 grid((unsigned)ceil_div(a.cin_stride, 32), (unsigned)ceil_div(a.cout_p, 32));
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
            } else {
                const size_t lds = (size_t)8 * khw * (64 + khw) * sizeof(float);
                const dim3 grid((unsigned)ceil_div(a.cin_stride, 64), (unsigned)ceil_div(a.cout_p, 8));
                hipLaunchKernelGGL((pack_weights_tiled_kernel<8, 64>), grid, dim3(256), lds, s, a);
            }
            return check_launch();
        }
        const int threads = 256;
        long long blocks = ceil_div(a.total, threads);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "pack_weights"; }
};

// per-dtype launchers live in conv_igemm_bf16.hip / conv_igemm_f32.hip (one translation unit per dtype
// keeps the build parallel)
// compute units of the current device (no device -- dry run on a CPU host: the MI355X's 256 CUs are assumed)
static int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else { (void)hipGetLastError(); cus = 256; }
    }
    return cus;
}
int launch_conv_bf16(int cfg, const ConvKArgs& k, int ncls, hipStream_t s);
int launch_conv_f32(int cfg, const ConvKArgs& k, int ncls, hipStream_t s);
bool conv_cfg_has_helper(int cfg);
int launch_patch_bf16(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_patch_f32(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_pp_bf16(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_pp_f32(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_pp2_bf16(int cfg, const ConvKArgs& k, int groups, hipStream_t s);
int launch_pp2_f32(int cfg, const ConvKArgs& k, int groups, hipStream_t s);
int launch_pp3_bf16(int cfg, const ConvKArgs& k, int groups, hipStream_t s);
int launch_one_bf16(int cfg, const ConvKArgs& k, int cus, hipStream_t s);     // persistent single-chunk tiles (conv3x3_one_kernel.h)
int one_grid(int ntot, int cus);                                              // ... their grid = the statistics rows they leave
int launch_pp3_f32(int cfg, const ConvKArgs& k, int groups, hipStream_t s);
int launch_s2_bf16(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_s2_f32(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_t2_bf16(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_t2_f32(int cfg, const ConvKArgs& k, hipStream_t s);
int launch_head_bf16(const ConvKArgs& k, hipStream_t s);
int launch_head_f32(const ConvKArgs& k, hipStream_t s);
int launch_c8_bf16(const ConvKArgs& k, hipStream_t s);
int launch_rowsum_bf16(const ConvKArgs& k, hipStream_t s);
int launch_c8_f32(const ConvKArgs& k, hipStream_t s);

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

static unsigned long long* g_conv_dbg_clocks = nullptr;    // v2v_conv_debug_clocks

struct ConvOp : Op {
    ConvKArgs k;
    int ncls, cfg, dtype;
    long long slab_bytes; int sk_tickets;
    int groups = 1;      // 2: grouped launch (v2v_conv2d_pair), second member's tensors in k.g1
    int launch(hipStream_t s) override {
        if (cfg >= 140 || cfg == 114) return launch_one_bf16(cfg, k, device_cus(), s);     // persistent, weights-resident single-chunk tiles (bf16: host check); 114: the transposed stride-2 one
        if (cfg >= 120) return dtype == V2V_BF16 ? launch_pp3_bf16(cfg, k, 1, s) : launch_pp3_f32(cfg, k, 1, s);    // 7x7 window on the single-phase kernel
        if (cfg >= 110) return dtype == V2V_BF16 ? launch_t2_bf16(cfg, k, s) : launch_t2_f32(cfg, k, s);
        if (cfg >= 100) return dtype == V2V_BF16 ? launch_s2_bf16(cfg, k, s) : launch_s2_f32(cfg, k, s);
        if (cfg >= 80) return dtype == V2V_BF16 ? launch_pp3_bf16(cfg, k, groups, s) : launch_pp3_f32(cfg, k, groups, s);
        if (cfg >= 70) return dtype == V2V_BF16 ? launch_pp2_bf16(cfg, k, groups, s) : launch_pp2_f32(cfg, k, groups, s);
        if (cfg == 62) return launch_rowsum_bf16(k, s);
        if (cfg == 61) return dtype == V2V_BF16 ? launch_c8_bf16(k, s) : launch_c8_f32(k, s);
        if (cfg == 60) return dtype == V2V_BF16 ? launch_head_bf16(k, s) : launch_head_f32(k, s);
        if (cfg >= 50) return dtype == V2V_BF16 ? launch_pp_bf16(cfg, k, s) : launch_pp_f32(cfg, k, s);
        if (cfg >= 32) return dtype == V2V_BF16 ? launch_patch_bf16(cfg, k, s) : launch_patch_f32(cfg, k, s);
        return dtype == V2V_BF16 ? launch_conv_bf16(cfg, k, ncls, s) : launch_conv_f32(cfg, k, ncls, s);
    }
    const char* name() const override { return "conv_igemm"; }
};

// One host-mapped (pinned, device-visible) status word per process: kernels OR bits into it at system scope, the host reads it
// without any synchronisation (v2v_device_status).  Bit 0: a fused-norm spin barrier gave up (its outputs are NaN).  Allocated on
// the first fused-norm launch; no device or a failed allocation -> NULL, the kernels then only poison their outputs.
static int* status_word() {
    static int* dev = nullptr;
    static bool tried = false;
    if (!tried) {
        if (v2v_get_dry_run()) return nullptr;                       // CPU host: nothing is launched, nothing to report
        tried = true;
        int* host = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *host = 0;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || d != (void*)host) { (void)hipGetLastError(); (void)hipHostFree(host); return nullptr; }
        dev = host;
    }
    return dev;
}

static int build_conv(const v2v_conv_desc* d_in, ConvOp* op, bool launching = true) {
    const v2v_conv_desc* d = d_in;
    if (!d || !d->in || !d->w || !d->out || !d->zero_page) { set_error("conv: null pointer"); return V2V_EINVAL; }
    // Persistent single-chunk tiles (140 - 143) on a layer with 64-byte pixels: the PAIRED-X view (include/v2v_hip.h, w_korder 3).  The
    // NHWC tensors [H][W][32] are [H][W/2][64]; the caller packed the 64 -> 64 matrix of that view (engine.PairedXConv) chunk-major.
    v2v_conv_desc dd;
    bool pair_x = false;
    if (d->tile >= 140 && d->tile <= 143 && d->cin_stride == 32) {
        if (d->dtype != V2V_BF16 || d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->cin > 32 || d->cout != 32 ||
            d->cout_stride != 32 || (d->out_mode != V2V_OUT_RAW_F32_NHWC && d->out_mode != V2V_OUT_RAW_ACT_NHWC) || d->w_korder != 3 || (d->W & 1) || d->OW != d->W || d->OH != d->H) {
            set_error("conv: tile configs 140 - 143 on 64-byte pixels (paired-x view) need a bf16 3x3/s1/p1 Conv2d, <= 32 -> exactly 32 channels, "
                      "dense raw NHWC output (fp32 or bf16), an even width and paired-x weights (w_korder 3)");
            return V2V_EINVAL;
        }
        dd = *d;
        dd.W = d->W / 2; dd.OW = d->OW / 2;
        dd.cin = 64; dd.cin_stride = 64; dd.cout = 64; dd.cout_stride = 64; dd.w_korder = 1;
        d = &dd;
        pair_x = true;
    } else if (d->tile == 114 && d->cin_stride == 32) {
        // ... and the transposed counterpart: ConvTranspose2d <= 32 -> 16 over [H][W][32] as 64 -> 32 over [H][W/2][64], output [2H][2W][16] = [2H][W][32]
        if (d->dtype != V2V_BF16 || !d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->cin > 32 || d->cout != 16 ||
            d->cout_stride != 16 || (d->out_mode != V2V_OUT_RAW_F32_NHWC && d->out_mode != V2V_OUT_RAW_ACT_NHWC) || d->w_korder != 3 || (d->W & 1) || d->OW != 2 * d->W || d->OH != 2 * d->H) {
            set_error("conv: tile config 114 on 64-byte pixels (paired-x view) needs a bf16 ConvTranspose2d(3x3, s2, p1, op1), <= 32 -> exactly 16 channels, "
                      "dense raw NHWC output (fp32 or bf16), an even width and paired-x weights (w_korder 3)");
            return V2V_EINVAL;
        }
        dd = *d;
        dd.W = d->W / 2; dd.OW = d->OW / 2;
        dd.cin = 64; dd.cin_stride = 64; dd.cout = 32; dd.cout_stride = 32; dd.w_korder = 2;
        d = &dd;
        pair_x = true;
    } else if (d->w_korder == 3) {
        set_error("conv: paired-x weights (w_korder 3) are read by tile configs 140 - 143 / 114 on layers with a 32-channel stride only"); return V2V_EINVAL;
    }
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
    k.pair_x = pair_x ? (d->transposed ? 2 : 1) : 0;            // 1: statistics columns / bias index = channel & 31, 2 (transposed view): & 15
    k.in = (const char*)d->in; k.w = (const char*)d->w; k.zero_page = (const char*)d->zero_page;
    k.bias = d->bias; k.out = (char*)d->out; k.stats = d->stats;
    k.N = d->N; k.H = d->H; k.W = d->W; k.cin_stride = d->cin_stride;
    k.cout = d->cout; k.cout_stride = d->cout_stride; k.cout_p = g.cout_p;
    k.OH = d->OH; k.OW = d->OW;
    k.pad_mode = d->pad_mode;
    if ((long long)d->N * d->H * d->W >= (1ll << 31) || (long long)d->N * d->OH * d->OW >= (1ll << 31)) { set_error("conv: too many pixels"); return V2V_EINVAL; }
    // the epilogue addresses the output with 32-bit element offsets
    if ((long long)d->N * d->OH * d->OW * (d->out_mode == V2V_OUT_F32_NCHW ? d->cout : d->cout_stride) >= (1ll << 31)) {
        set_error("conv: output of 2^31 elements or more"); return V2V_EINVAL;
    }
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
    for (int c = 0; c < 4; ++c) {                              // reciprocals of the class grid sizes (epilogue row -> pixel map)
        uint32_t m; int32_t l;
        v2v_fastdiv_magic((uint32_t)(k.OHc[c] * k.OWc[c] > 0 ? k.OHc[c] * k.OWc[c] : 1), &m, &l); k.div_m[c][0] = m; k.div_l[c][0] = l;
        v2v_fastdiv_magic((uint32_t)(k.OWc[c] > 0 ? k.OWc[c] : 1), &m, &l);                       k.div_m[c][1] = m; k.div_l[c][1] = l;
    }
    const long long Mc = k.Mc[0];   // class 0 is the largest
    for (int c = 0; c < 4; ++c) {
        k.nkh[c] = g.nkh[c]; k.nkw[c] = g.nkw[c]; k.dh0[c] = g.dh0[c]; k.dw0[c] = g.dw0[c];
        k.ktot[c] = g.ktot[c]; k.kpad[c] = g.kpad[c]; k.wrow[c] = g.wrow[c]; k.woff[c] = g.woff[c];
    }
    k.out_mode = d->out_mode; k.act = d->act; k.act_param = d->act_param; k.out_scale = d->out_scale;
    if (d->act_split != 0) {
        if (d->act_split < 0 || d->act_split >= d->cout || (d->tile != 60 && d->tile != 62) || d->out_mode != V2V_OUT_F32_NCHW) {
            set_error("conv: act_split (merged heads) needs tile 60 / 62, planar fp32 output and 0 < act_split < cout"); return V2V_EINVAL;
        }
        k.act_split = d->act_split; k.act_b = d->act_b; k.act_param_b = d->act_param_b; k.out_scale_b = d->out_scale_b;
    }
    if (d->out_mode == V2V_OUT_NORM_ACT_NHWC) {
        if ((launching && (!d->fin_counter || !d->stats || !d->fin_scale_shift || d->fin_count <= 0)) || d->splitk > 1 || d->transposed ||
            d->cout != d->cout_stride || !((d->tile >= 80 && d->tile < 88) || (d->tile >= 90 && d->tile <= 93)) || d->cout > 128 * 64) {
            set_error("conv: fused norm needs tile 80..87 / 90..93, splitk <= 1, cout == cout_stride, stats (tagged granules: rows * cout * 4 zero-initialised floats), fin_counter (V2V_FIN_TAG_WORD + 128 ints), fin_scale_shift, fin_count");
            return V2V_EINVAL;
        }
        k.res0 = (const char*)d->res0; k.res1 = (const char*)d->res1;
        k.status = launching ? status_word() : nullptr;
    }
    if (d->fin_counter) {
        if (!d->stats || !d->fin_scale_shift || d->fin_count <= 0 ||
            (d->out_mode != V2V_OUT_RAW_F32_NHWC && d->out_mode != V2V_OUT_NORM_ACT_NHWC && d->out_mode != V2V_OUT_RAW_ACT_NHWC)) {
            set_error("conv: in-kernel norm finalize needs stats, fin_scale_shift, fin_count and RAW output"); return V2V_EINVAL;
        }
        k.fin_counter = d->fin_counter; k.fin_gamma = d->fin_gamma; k.fin_beta = d->fin_beta; k.fin_out = d->fin_scale_shift;
        k.fin_rmean = d->fin_running_mean; k.fin_rvar = d->fin_running_var;
        k.fin_eps = d->fin_eps; k.fin_momentum = d->fin_momentum;
        k.fin_inv_count = 1.0 / (double)d->fin_count;
        k.fin_unbias = d->fin_count > 1 ? (double)d->fin_count / (double)(d->fin_count - 1) : 1.0;
    }
    if (d->out_mode == V2V_OUT_RAW_ACT_NHWC && d->dtype == V2V_F32) k.out_mode = V2V_OUT_RAW_F32_NHWC;     // fp32 storage: the same tensor
    if (d->out_mode < 0 || d->out_mode > V2V_OUT_RAW_ACT_NHWC) { set_error("conv: bad out_mode %d", d->out_mode); return V2V_EINVAL; }
    if (d->stats && d->out_mode == V2V_OUT_RAW_ACT_NHWC && d->cout_stride % (d->dtype == V2V_BF16 ? 8 : 4) != 0) {
        set_error("conv: RAW_ACT output needs a channel stride of whole 16-byte vectors"); return V2V_EINVAL;
    }
    op->ncls = g.ncls;
    op->dtype = d->dtype;
    op->cfg = d->tile ? d->tile : choose_cfg(Mc, d->cout, g.ncls);
    if (d->out_mode == V2V_OUT_RAW_ACT_NHWC && (op->cfg == 60 || op->cfg == 61)) {
        set_error("conv: tile config %d writes fp32 raw only (V2V_OUT_RAW_F32_NHWC)", op->cfg); return V2V_EINVAL;
    }
    int tile_bm, tile_bn;
    if (op->cfg == 61) {
        // conv7x7_c8_kernel: 7x7 / stride 1 / pad 3 Conv2d over pixels of exactly 16 bytes (8 bf16 / 4 fp32 channels), <= 128 output channels
        // Round 6: zero padding of 3 ... 6 (the output grid is then (H + 2 pad - 6) x (W + 2 pad - 6): pad 6 is the "full" convolution the
        // backward-data of a head behind ReflectionPad2d(3) is), and activation-typed NHWC output without activation for that use
        const int vec = d->dtype == V2V_BF16 ? 8 : 4;
        const bool act_out = d->out_mode == V2V_OUT_ACT_NHWC;
        if (d->transposed || d->KH != 7 || d->KW != 7 || d->stride != 1 || d->pad < 3 || d->pad > 6 || (d->pad != 3 && d->pad_mode != V2V_PAD_ZERO) ||
            d->OH != d->H + 2 * d->pad - 6 || d->OW != d->W + 2 * d->pad - 6 || d->cout > 128 ||
            d->cin_stride * (d->dtype == V2V_BF16 ? 2 : 4) != 16 || d->w_korder != 0 || d->splitk > 1 || d->fin_counter || d->act_split != 0 ||
            !(d->out_mode == V2V_OUT_F32_NCHW || d->out_mode == V2V_OUT_RAW_F32_NHWC || act_out) ||
            (act_out && (d->act != V2V_ACT_NONE || d->out_scale != 1.f || d->cout % vec != 0 || d->cout_stride % vec != 0 || ((unsigned long long)d->out & 15ull))) ||
            (d->stats && d->out_mode != V2V_OUT_RAW_F32_NHWC) ||
            (long long)d->N * d->H * d->W * 16 >= (1ll << 32)) {
            set_error("conv: tile config 61 (7x7 over 16-byte pixels) needs a 7x7/s1 Conv2d with pad 3 (or zero padding of 3 ... 6), a channel stride of "
                      "16 bytes, cout <= 128, planar fp32 / raw NHWC output without in-kernel norm finalize, or plain activation-typed NHWC output "
                      "(no activation, whole 16-byte vectors)"); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, 8);
        k.tiles_w = (int)ceil_div(d->OW, 32);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = 1;
        tile_bm = 256; tile_bn = 4;
    } else if (op->cfg == 62) {
        // conv7x7_rowsum_kernel: the generator heads (7x7 / stride 1 / pad 3 Conv2d, <= 4 output channels, planar fp32 + activation), bf16
        if (d->transposed || d->KH != 7 || d->KW != 7 || d->stride != 1 || d->pad != 3 || d->cout > 4 || d->dtype != V2V_BF16 ||
            d->cin_stride % 32 != 0 || d->w_korder != 0 || d->splitk > 1 || d->fin_counter || d->stats || d->out_mode != V2V_OUT_F32_NCHW ||
            (long long)d->N * d->H * d->W * d->cin_stride * 2 >= (1ll << 32)) {
            set_error("conv: tile config 62 (7x7 heads as row GEMM + shifted sum) needs a bf16 7x7/s1/p3 Conv2d with cout <= 4, cin_stride %% 32 == 0, "
                      "planar fp32 output, no statistics"); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, 10);
        k.tiles_w = (int)ceil_div(d->OW, 32);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = 1;
        tile_bm = 320; tile_bn = 4;
    } else if (op->cfg == 60) {
        // conv7x7_head_kernel: 7x7 / stride 1 / pad 3 Conv2d with <= 16 output channels written planar fp32
        if (d->transposed || d->KH != 7 || d->KW != 7 || d->stride != 1 || d->pad != 3 || d->cout > 32 ||
            d->cin_stride % (bke_of(d->dtype) / 2) != 0 || d->w_korder != 0 || d->splitk > 1 || d->fin_counter ||
            !(d->out_mode == V2V_OUT_F32_NCHW || d->out_mode == V2V_OUT_RAW_F32_NHWC) ||
            (d->stats && d->out_mode != V2V_OUT_RAW_F32_NHWC) ||
            (long long)d->N * d->H * d->W * d->cin_stride * (d->dtype == V2V_BF16 ? 2 : 4) >= (1ll << 32)) {
            set_error("conv: tile config 60 (7x7, cout <= 32) needs a 7x7/s1/p3 Conv2d, cin_stride %% %d == 0 (whole 64-byte half chunks), planar fp32 or raw "
                      "NHWC output without in-kernel norm finalize", bke_of(d->dtype) / 2); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, 8);
        k.tiles_w = (int)ceil_div(d->OW, 32);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = 1;
        tile_bm = 256; tile_bn = 4;
    } else if (op->cfg >= 120 && op->cfg < 130) {
        // conv3x3_pp3_body with a 7x7 window (tile 120; staged for round 5): dense 7x7 / stride 1 / pad 3 Conv2d whose channel
        // stride is a whole number of 128-byte chunks, weights channel-chunk major (korder 1), bf16
        const PatchCfg* pc = find_pp3_cfg(op->cfg);
        if (!pc) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
        if (d->transposed || d->KH != 7 || d->KW != 7 || d->stride != 1 || d->pad != 3 || d->dtype != V2V_BF16 ||
            d->cin_stride % bke_of(d->dtype) != 0 || d->w_korder != 1 || d->out_mode == V2V_OUT_NORM_ACT_NHWC ||
            (d->pad_mode == V2V_PAD_REFLECT && (d->H <= 3 || d->W <= 3)) ||
            (long long)d->N * d->H * d->W * d->cin_stride * 2 >= (1ll << 32)) {
            set_error("conv: tile config %d needs a bf16 7x7/s1/p3 Conv2d, cin_stride %% %d == 0, korder-1 weights, no fused norm",
                      op->cfg, bke_of(d->dtype)); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, pc->TH);
        k.tiles_w = (int)ceil_div(d->OW, pc->TW);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = (int)ceil_div(d->cout, pc->BN);
        tile_bm = pc->TH * pc->TW; tile_bn = pc->BN;
    } else if (op->cfg >= 110 && op->cfg < 120) {
        // conv3x3_t2_kernel: ConvTranspose2d(3x3, stride 2, padding 1), all four output-parity classes per workgroup, full-tap (korder 2) weights
        const PatchCfg* pc = find_t2_cfg(op->cfg);
        if (!pc) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
        if (!d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 ||
            d->cin_stride % bke_of(d->dtype) != 0 || d->w_korder != 2 || d->splitk > 1 || d->out_mode == V2V_OUT_NORM_ACT_NHWC ||
            (long long)d->N * d->H * d->W * d->cin_stride * (d->dtype == V2V_BF16 ? 2 : 4) >= (1ll << 32)) {
            set_error("conv: transposed stride-2 patch tile config %d needs a ConvTranspose2d(3x3, s2, p1), cin_stride %% %d == 0, korder-2 weights, no split-K",
                      op->cfg, bke_of(d->dtype)); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div((d->OH + 1) / 2, pc->TH);    // tiles of input positions (a, b): output pixels (2a + py, 2b + px)
        k.tiles_w = (int)ceil_div((d->OW + 1) / 2, pc->TW);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = (int)ceil_div(d->cout, pc->BN);
        if (op->cfg == 114 && (d->dtype != V2V_BF16 || d->cin_stride != 64 || d->cout > pc->BN || (d->cout & 3) || (d->cout_stride & 3) || ((unsigned long long)d->out & 15ull) ||
                               (d->out_mode != V2V_OUT_RAW_F32_NHWC && d->out_mode != V2V_OUT_RAW_ACT_NHWC) || d->fin_workspace != nullptr || d->H % pc->TH != 0 || d->W % pc->TW != 0 ||
                               d->OH != 2 * d->H || d->OW != 2 * d->W)) {
            // conv3x3_one_kernel.h, conv3x3_t2_one_kernel: ONE output mode (raw fp32 NHWC + one statistics row per workgroup), full tiles only
            set_error("conv: tile config 114 (persistent transposed stride-2 tile) needs bf16, cin_stride 64, cout <= %d and %% 4 == 0, raw fp32 NHWC output (16-byte aligned "
                      "rows), no two-level finalize workspace, H %% %d == 0, W %% %d == 0, OH = 2 H, OW = 2 W", pc->BN, pc->TH, pc->TW);
            return V2V_EINVAL;
        }
        k.woff[0] = 0; k.wrow[0] = 9 * d->cin_stride;          // the single full-tap matrix
        k.fin_rows = op->cfg == 114 ? 0 : 4 * k.m_tiles;       // every workgroup publishes one statistics row per class (114: one per workgroup, all classes)
        tile_bm = pc->TH * pc->TW; tile_bn = pc->BN;
    } else if (op->cfg >= 100 && op->cfg < 110) {
        // conv3x3_s2_kernel: 3x3 / stride 2 / pad 1 (zero) Conv2d, channel stride a multiple of the 128-byte chunk, korder-1 weights
        const PatchCfg* pc = find_s2_cfg(op->cfg);
        if (!pc) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
        if (d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->pad_mode != V2V_PAD_ZERO ||
            d->cin_stride % bke_of(d->dtype) != 0 || d->w_korder != 1 || d->splitk > 1 || d->out_mode == V2V_OUT_NORM_ACT_NHWC ||
            (long long)d->N * d->H * d->W * d->cin_stride * (d->dtype == V2V_BF16 ? 2 : 4) >= (1ll << 32)) {
            set_error("conv: stride-2 patch tile config %d needs a 3x3/s2/p1 zero-padded Conv2d, cin_stride %% %d == 0, korder-1 weights, no split-K",
                      op->cfg, bke_of(d->dtype)); return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, pc->TH);
        k.tiles_w = (int)ceil_div(d->OW, pc->TW);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = (int)ceil_div(d->cout, pc->BN);
        tile_bm = pc->TH * pc->TW; tile_bn = pc->BN;
    } else if (op->cfg >= 32) {
        // conv3x3_patch_kernel: 3x3 / stride 1 / pad 1 Conv2d, channel stride a multiple of the 128-byte chunk,
        // weights packed channel-chunk outer (korder 1)
        const PatchCfg* pc = op->cfg >= 80 ? find_pp3_cfg(op->cfg) : op->cfg >= 70 ? find_pp2_cfg(op->cfg) : op->cfg >= 50 ? find_pp_cfg(op->cfg) : find_patch_cfg(op->cfg);
        if (!pc) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
        const bool pad2_ok = d->pad == 2 && op->cfg >= 80 && op->cfg <= 93 && d->pad_mode == V2V_PAD_ZERO && d->out_mode != V2V_OUT_NORM_ACT_NHWC;   // single-phase tiles: "full" 3x3 convolution (backward-data behind a ReflectionPad2d)
        if (d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 1 || (d->pad != 1 && !pad2_ok) ||
            d->cin_stride % bke_of(d->dtype) != 0 || d->w_korder != 1 ||
            (long long)d->N * d->H * d->W * d->cin_stride * (d->dtype == V2V_BF16 ? 2 : 4) >= (1ll << 32)) {
            set_error("conv: patch tile config %d needs a 3x3/s1/p1 Conv2d, cin_stride %% %d == 0 and korder-1 weights",
                      op->cfg, bke_of(d->dtype)); return V2V_EINVAL;
        }
        if ((op->cfg >= 140 && op->cfg <= 143) && ((op->cfg >= 142 && d->cout != pc->BN) || d->cout > pc->BN || (d->cout & 3) || (d->cout_stride & 3) || ((unsigned long long)d->out & 15ull) ||
                               (d->out_mode != V2V_OUT_RAW_F32_NHWC && d->out_mode != V2V_OUT_RAW_ACT_NHWC) || d->fin_workspace != nullptr || d->OH % pc->TH != 0 || d->OW % pc->TW != 0)) {
            // conv3x3_one_kernel.h: ONE output mode (raw fp32 NHWC + one statistics row per workgroup, single-level in-kernel finalize), full tiles only
            set_error("conv: tile configs 140 - 143 (persistent, weights resident) need cout <= %d (142 / 143: exactly) and %% 4 == 0, raw fp32 NHWC output (16-byte aligned rows), "
                      "no two-level finalize workspace, OH %% %d == 0 and OW %% %d == 0", pc->BN, pc->TH, pc->TW);
            return V2V_EINVAL;
        }
        if ((op->cfg == 94 || op->cfg == 95 || op->cfg == 96 || (op->cfg >= 140 && op->cfg <= 143)) && (d->dtype != V2V_BF16 || d->cin_stride != bke_of(d->dtype) || d->splitk > 1 ||
                                                 d->out_mode == V2V_OUT_NORM_ACT_NHWC)) {
            set_error("conv: tile config %d is a single-chunk tile: bf16, cin_stride exactly %d, no split-K, no fused norm", op->cfg, bke_of(d->dtype));
            return V2V_EINVAL;
        }
        k.tiles_h = (int)ceil_div(d->OH, pc->TH);
        k.tiles_w = (int)ceil_div(d->OW, pc->TW);
        k.m_tiles = d->N * k.tiles_h * k.tiles_w;
        k.n_tiles = (int)ceil_div(d->cout, pc->BN);
        tile_bm = pc->TH * pc->TW; tile_bn = pc->BN;
    } else {
        if (d->w_korder != 0) { set_error("conv: tile config %d reads tap-major (korder 0) weights", op->cfg); return V2V_EINVAL; }
        const TileCfg* c = find_cfg(op->cfg);
        if (!c) { set_error("conv: unknown tile config %d", op->cfg); return V2V_EINVAL; }
        k.m_tiles = (int)ceil_div(Mc, c->BM);
        k.n_tiles = (int)ceil_div(d->cout, c->BN);
        tile_bm = c->BM; tile_bn = c->BN;
    }
    if (op->cfg >= 32 && k.tiles_h > 0 && k.tiles_w > 0 && k.m_tiles > 0) {   // patch kernels: tile-index divisions by multiplication (ConvKArgs.idx_m)
        const uint32_t dv[3] = {(uint32_t)k.m_tiles, (uint32_t)(k.tiles_h * k.tiles_w), (uint32_t)k.tiles_w};
        for (int i = 0; i < 3; ++i) {
            uint32_t m = 0; int32_t l = 0;
            if (v2v_fastdiv_magic(dv[i], &m, &l) != 0) return V2V_EINVAL;
            k.idx_m[i] = m; k.idx_l[i] = l;
        }
    }
    // ---- split-K / weight prefetch ----
    k.splitk = d->splitk > 1 ? d->splitk : 1;
    int nk_min = k.kpad[0] / bke_of(d->dtype);
    for (int cc = 1; cc < g.ncls; ++cc) nk_min = std::min(nk_min, k.kpad[cc] / bke_of(d->dtype));
    if (op->cfg >= 32) nk_min = 2 * (d->cin_stride / bke_of(d->dtype));     // slices are whole channel chunks (>= 1 each)
    if (k.splitk > 1 && (k.splitk > 16 || nk_min / k.splitk < 2)) {
        set_error("conv: splitk %d needs >= 2 K chunks per slice (layer has %d)", k.splitk, nk_min); return V2V_EINVAL;
    }
    op->sk_tickets = g.ncls * k.m_tiles * k.n_tiles;
    op->slab_bytes = k.splitk > 1 ? (long long)op->sk_tickets * k.splitk * tile_bm * tile_bn * 4 : 0;
    if (k.splitk > 1) {
        if (launching && (!d->slabs || !d->sk_counter)) { set_error("conv: splitk needs slabs and sk_counter"); return V2V_EINVAL; }
        k.slabs = (float*)d->slabs; k.sk_counter = d->sk_counter;
    }
    if (d->fin_counter && d->fin_workspace && d->out_mode != V2V_OUT_NORM_ACT_NHWC) {
        // two-level in-kernel finalize for layers with more than 512 statistics rows (the arithmetic of v2v_bn_finalize's two stages)
        const int rows = k.fin_rows > 0 ? k.fin_rows : g.ncls * k.m_tiles;
        const int groups = rows > 512 ? (rows >= 64 * 128 ? 64 : (rows + 127) / 128) : 0;      // = v2v_bn_finalize_groups(rows)
        if (groups > 0) {
            if (k.n_tiles > 128) { set_error("conv: two-level finalize supports up to 128 channel tiles"); return V2V_EINVAL; }
            k.fin_groups = groups; k.fin_ws = d->fin_workspace;
        }
    }
    k.ablate = d->ablate;
    k.dbg = g_conv_dbg_clocks;
    {
        static const int rev = [] { const char* e = getenv("V2V_CLS_ORDER"); return (e && e[0] == '0') ? 0 : 1; }();
        k.cls_rev = rev;
    }
    k.pf_dist = (d->prefetch > 0 && (conv_cfg_has_helper(op->cfg) || (op->cfg >= 32 && op->cfg <= 37))) ? d->prefetch : 0;
    {
        static const int ep_fast = [] { const char* e = getenv("V2V_EPILOGUE_FAST"); return (e && e[0] == '0') ? 0 : 1; }();
        k.ep_slow = ep_fast ? 0 : 1;
    }
    k.pf_mask = (k.pf_dist > 0 && k.m_tiles >= 8) ? 3 : 0;      // 1 prefetching workgroup per 4 M tiles of an N column
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
                                     int32_t dtype, int32_t korder, void* stream) {
    if (!w || !dst) { set_error("pack: null pointer"); return V2V_EINVAL; }
    const int src_cl = (korder >> 8) & 1;                      // + 256: the source tensor is channels-last (see the header)
    korder &= 255;
    if (korder == 2) {
        // full-tap packing of a ConvTranspose2d(3x3, stride 2) for conv3x3_t2_kernel: ONE matrix [cout_p][9 * cin_stride], channel-chunk
        // outer, kernel tap ky * 3 + kx inner -- the korder-1 layout of a Conv2d, read from the [cin][cout][kh][kw] parameter.  Same
        // element count as the four class matrices of korder 0 (1 + 2 + 2 + 4 = 9 taps).
        if (!transposed || stride != 2 || KH != 3 || KW != 3 || cin_stride % bke_of(dtype) != 0) {
            set_error("pack: korder 2 needs a ConvTranspose2d(3x3, stride 2) whose channel stride is a multiple of the 128-byte chunk"); return V2V_EINVAL;
        }
    } else if (korder == 4) {
        // backward-data operator of a 3x3 / stride 1 Conv2d AS A CONVOLUTION (round 6): the parameter read with its roles swapped
        // (transposed layout: dim 0 = this operator's input channels) and its taps FLIPPED, packed channel-chunk-major like a
        // Conv2d's korder 1 -- what the 3x3 patch kernels read; the caller runs them with pad = 2 - (the layer's pad)
        if (!transposed || stride != 1 || KH != 3 || KW != 3 || cin_stride % bke_of(dtype) != 0 || src_cl) {
            set_error("pack: korder 4 needs the role-swapped (transposed = 1) read of a 3x3 / stride 1 Conv2d weight, channel stride a multiple of the 128-byte chunk"); return V2V_EINVAL;
        }
    } else if (korder == 5) {
        // the same operator for a square stride-1 Conv2d of any size, packed TAP-major like a Conv2d's korder 0: what conv7x7_c8_kernel
        // (tile 61) reads for the backward-data of the 7x7 heads (3 gradient channels = one 16-byte pixel); run with pad = 6 - (the layer's pad)
        if (!transposed || stride != 1 || KH != KW || !(KH & 1) || src_cl) {
            set_error("pack: korder 5 needs the role-swapped (transposed = 1) read of a square, odd, stride-1 Conv2d weight"); return V2V_EINVAL;
        }
    } else if (korder != 0 && (korder != 1 || transposed || cin_stride % bke_of(dtype) != 0)) {
        set_error("pack: korder 1 needs a Conv2d whose channel stride is a multiple of the 128-byte chunk"); return V2V_EINVAL;
    }
    ConvGeom g;
    conv_geom(cin_stride, cout, KH, KW, (korder == 4 || korder == 5) ? 0 : transposed, stride, pad, dtype, &g);
    int kh0_flip = -1;
    if (korder == 4) { kh0_flip = KH - 1; korder = 1; }
    if (korder == 5) { kh0_flip = KH - 1; korder = 0; }
    if (korder == 2) {
        g.ncls = 1; g.nkh[0] = 3; g.nkw[0] = 3; g.kh0[0] = 0; g.kw0[0] = 0;
        g.ktot[0] = g.kpad[0] = g.wrow[0] = 9 * cin_stride; g.woff[0] = 0;
        g.total = (long long)g.cout_p * g.wrow[0];
        stride = 1;                                            // kstep: consecutive taps
        korder = 1;                                            // destination order
    }
    auto op = std::make_unique<PackOp>();
    PackArgs& a = op->a;
    a.w = w; a.dst = dst; a.cin = cin; a.cin_stride = cin_stride; a.cout = cout; a.cout_p = g.cout_p;
    a.KH = KH; a.KW = KW; a.transposed = transposed; a.kstep = stride; a.ncls = g.ncls; a.total = g.total; a.dtype = dtype;
    a.korder = korder; a.bke = bke_of(dtype); a.src_cl = src_cl;
    for (int c = 0; c < 4; ++c) {
        a.nkh[c] = g.nkh[c]; a.nkw[c] = g.nkw[c]; a.kh0[c] = g.kh0[c]; a.kw0[c] = g.kw0[c];
        a.kpad[c] = g.kpad[c]; a.wrow[c] = g.wrow[c]; a.woff[c] = g.woff[c];
    }
    if (kh0_flip >= 0) { a.kh0[0] = kh0_flip; a.kw0[0] = KW - 1; a.kstep = -1; }      // tap t of the matrix = kernel tap (2 - t): flipped
    return submit(std::move(op), stream);
}

extern "C" int v2v_conv_debug_clocks(void* device_buffer) {
    g_conv_dbg_clocks = reinterpret_cast<unsigned long long*>(device_buffer);
    return 0;
}

extern "C" int v2v_conv_stats_rows(const v2v_conv_desc* d) {
    ConvOp op;
    if (build_conv(d, &op, false) != 0) return V2V_EINVAL;
    if (op.cfg >= 140 || op.cfg == 114) return one_grid(op.k.m_tiles * op.k.n_tiles, device_cus());     // persistent tiles: one row per workgroup
    return op.ncls * op.k.m_tiles;
}

extern "C" int64_t v2v_conv_splitk_workspace(const v2v_conv_desc* d, int32_t* tickets) {
    ConvOp op;
    if (build_conv(d, &op, false) != 0) return V2V_EINVAL;
    if (tickets) *tickets = op.sk_tickets;
    return op.slab_bytes;
}

extern "C" int v2v_conv_tile_config(const v2v_conv_desc* d) {
    ConvOp op;
    if (build_conv(d, &op, false) != 0) return V2V_EINVAL;
    return op.cfg;
}

// Fused norm: the spin barrier needs every workgroup of the launch on the chip at once (one workgroup per CU at these
// LDS sizes).  No device (dry run on a CPU host): the MI355X's 256 CUs are assumed.

// Division of 0 <= n < 2^31 by a constant 1 <= d < 2^31 as q = (umulhi(M, n) + n) >> l  (Granlund & Montgomery, "Division by invariant
// integers using multiplication", the round-up form): l = ceil(log2 d), M = floor(2^32 (2^l - d) / d) + 1 < 2^32; umulhi(M, n) < n, so
// the sum stays below 2^32.  Used by the conv epilogue for the per-row divisions by the (uniform) class grid sizes.
extern "C" int v2v_fastdiv_magic(uint32_t d, uint32_t* m_out, int32_t* l_out) {
    if (d == 0 || d >= (1u << 31) || !m_out || !l_out) { set_error("fastdiv: divisor out of range"); return V2V_EINVAL; }
    int l = 0;
    while ((1ull << l) < d) ++l;
    *m_out = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    *l_out = l;
    return 0;
}

extern "C" int v2v_device_status(int32_t clear) {
    int* w = status_word();
    if (w == nullptr) return 0;
    const int v = __atomic_load_n(w, __ATOMIC_RELAXED);
    if (clear && v) __atomic_fetch_and(w, 0, __ATOMIC_RELAXED);
    return v;
}

static int fused_norm_resident(const ConvOp* op) {
    const long long wgs = (long long)op->k.m_tiles * op->k.n_tiles * op->groups;
    if (wgs > device_cus()) {
        set_error("conv: fused norm needs all %lld workgroups resident at once, the device has %d compute units", wgs, device_cus());
        return V2V_EINVAL;
    }
    return 0;
}

extern "C" int v2v_conv_fused_norm_max_workgroups(void) { return device_cus(); }

extern "C" int v2v_conv2d_pair(const v2v_conv_desc* a, const v2v_conv_desc* b, void* stream) {
    auto op = std::make_unique<ConvOp>();
    ConvOp ob;
    int rc = build_conv(a, op.get());
    if (rc == 0) rc = build_conv(b, &ob);
    if (rc != 0) return rc;
    if (op->cfg < 70 || op->cfg >= 94 || ob.cfg != op->cfg) {
        set_error("conv pair: both members need the same grouped-launch tile config (70..93), got %d / %d", op->cfg, ob.cfg);
        return V2V_EINVAL;
    }
    const bool same =
        a->N == b->N && a->H == b->H && a->W == b->W && a->cin == b->cin && a->cin_stride == b->cin_stride &&
        a->cout == b->cout && a->cout_stride == b->cout_stride && a->KH == b->KH && a->KW == b->KW && a->stride == b->stride &&
        a->pad == b->pad && a->pad_mode == b->pad_mode && a->transposed == b->transposed && a->OH == b->OH && a->OW == b->OW &&
        a->dtype == b->dtype && a->out_mode == b->out_mode && a->act == b->act && a->act_param == b->act_param &&
        a->out_scale == b->out_scale && a->splitk == b->splitk && a->w_korder == b->w_korder && a->ablate == b->ablate &&
        (a->bias != nullptr) == (b->bias != nullptr) && (a->stats != nullptr) == (b->stats != nullptr) &&
        (a->fin_counter != nullptr) == (b->fin_counter != nullptr) && a->fin_eps == b->fin_eps &&
        a->fin_momentum == b->fin_momentum && a->fin_count == b->fin_count &&
        (a->fin_gamma != nullptr) == (b->fin_gamma != nullptr) && (a->fin_beta != nullptr) == (b->fin_beta != nullptr) &&
        (a->fin_running_mean != nullptr) == (b->fin_running_mean != nullptr) &&
        (a->fin_running_var != nullptr) == (b->fin_running_var != nullptr);
    if (!same) { set_error("conv pair: the two members must have identical geometry, modes and optional-operand sets"); return V2V_EINVAL; }
    if (a->out == b->out || (a->stats && a->stats == b->stats) || (a->fin_counter && a->fin_counter == b->fin_counter) ||
        (a->fin_scale_shift && a->fin_scale_shift == b->fin_scale_shift) || (a->splitk > 1 && (a->slabs == b->slabs || a->sk_counter == b->sk_counter))) {
        set_error("conv pair: the members must not share outputs, statistics, tickets or split-K scratch"); return V2V_EINVAL;
    }
    ConvGroupPtrs& g = op->k.g1;
    g.in = ob.k.in; g.w = ob.k.w; g.bias = ob.k.bias; g.out = ob.k.out; g.stats = ob.k.stats;
    g.fin_counter = ob.k.fin_counter; g.fin_gamma = ob.k.fin_gamma; g.fin_beta = ob.k.fin_beta; g.fin_out = ob.k.fin_out;
    g.fin_rmean = ob.k.fin_rmean; g.fin_rvar = ob.k.fin_rvar; g.slabs = ob.k.slabs; g.sk_counter = ob.k.sk_counter;
    g.res0 = ob.k.res0; g.res1 = ob.k.res1;
    op->groups = 2;
    {   // members on disjoint XCD halves (grouped_xcd_map; tiles 80-93 only -- conv3x3_pp2_kernel keeps the shared mapping); V2V_GROUP_XCD=0: off
        static const int on = [] { const char* e = getenv("V2V_GROUP_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
        op->k.grp_xcd = on;
    }
    if (a->out_mode == V2V_OUT_NORM_ACT_NHWC) {
        if ((a->res0 != nullptr) != (b->res0 != nullptr) || (a->res1 != nullptr) != (b->res1 != nullptr)) {
            set_error("conv pair: the members must have the same residual operands"); return V2V_EINVAL;
        }
        rc = fused_norm_resident(op.get());
        if (rc != 0) return rc;
    }
    return submit(std::move(op), stream);
}

extern "C" int v2v_conv2d(const v2v_conv_desc* d, void* stream) {
    auto op = std::make_unique<ConvOp>();
    int rc = build_conv(d, op.get());
    if (rc != 0) return rc;
    if (d->out_mode == V2V_OUT_NORM_ACT_NHWC) {
        rc = fused_norm_resident(op.get());
        if (rc != 0) return rc;
    }
    return submit(std::move(op), stream);
}
