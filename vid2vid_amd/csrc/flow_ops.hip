// FlowNet2's three native ops, forward only (vid2vid runs FlowNet2 frozen under no_grad:
// models/flownet.py:18-26).  Written from the maths of the reference kernels, wave64-native:
//   correlation   correlation_cuda_kernel.cu:73-147  (+ output-size rule correlation_cuda.cc:25-38)
//   resample2d    resample2d_kernel.cu:15-64
//   channelnorm   channelnorm_kernel.cu:18-60
// The reference first copies both inputs into zero-padded NHWC scratch (channels_first,
// correlation_cuda_kernel.cu:46-70) and then reduces 32 channels per warp with shuffles;
// here padding is a bounds test and every lane owns whole output pixels (coalesced along x),
// so there is no scratch tensor and no cross-lane reduction at all.
#include "v2v_internal.h"
#include <cstdlib>

namespace v2v {

struct CorrArgs {
    const float* in1; const float* in2; float* out;
    int N, C, H, W, OH, OW, D;            // D = displacement_size
    int pad, ksize, krad, max_disp, s1, s2, drad;
};

// grid: (ceil(OW/32), OH * D, N); block (32, 8): thread (tx, ty) owns output x and the
// displacements ti = ty, ty+8, ... of row tj = blockIdx.y % D.
__global__ __launch_bounds__(256) void correlation_kernel(const CorrArgs a) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ox = blockIdx.x * 32 + tx;
    const int oy = blockIdx.y / a.D;
    const int tjn = blockIdx.y - oy * a.D;          // 0..D-1
    const int n = blockIdx.z;
    if (ox >= a.OW) return;
    const long long hw = (long long)a.H * a.W;
    const float* f1 = a.in1 + (long long)n * a.C * hw;
    const float* f2 = a.in2 + (long long)n * a.C * hw;
    // centre positions in the UNPADDED frame (reference works in the padded one)
    const int y1 = oy * a.s1 + a.max_disp - a.pad;
    const int x1 = ox * a.s1 + a.max_disp - a.pad;
    const int y2 = y1 + (tjn - a.drad) * a.s2;
    constexpr int MAXT = 8;                         // displacement columns per thread (D <= 64)
    float acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) acc[q] = 0.f;
    for (int j = -a.krad; j <= a.krad; ++j) {
        const int ya = y1 + j, yb = y2 + j;
        const bool yok = (unsigned)ya < (unsigned)a.H && (unsigned)yb < (unsigned)a.H;
        if (!yok) continue;                         // a zero-padded operand kills the product
        for (int i = -a.krad; i <= a.krad; ++i) {
            const int xa = x1 + i;
            if ((unsigned)xa >= (unsigned)a.W) continue;
            for (int c = 0; c < a.C; ++c) {
                const float v1 = f1[c * hw + (long long)ya * a.W + xa];
                const float* r2 = f2 + c * hw + (long long)yb * a.W;
#pragma unroll
                for (int q = 0; q < MAXT; ++q) {
                    const int tin = ty + q * 8;
                    if (tin < a.D) {
                        const int xb = xa + (tin - a.drad) * a.s2;
                        if ((unsigned)xb < (unsigned)a.W) acc[q] += v1 * r2[xb];
                    }
                }
            }
        }
    }
    const float nelems = (float)(a.ksize * a.ksize * a.C);
    const long long ohw = (long long)a.OH * a.OW;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const int tin = ty + q * 8;
        if (tin < a.D) {
            const int tc = tjn * a.D + tin;
            a.out[((long long)n * a.D * a.D + tc) * ohw + (long long)oy * a.OW + ox] = acc[q] / nelems;
        }
    }
}

// ---- LDS-staged correlation for FlowNetC's geometry class (kernel_size 1, stride1 1, stride2 2, pad == max_disp) ----
// out[n][tj*D + ti][y][x] = (1/C) sum_c f1[n][c][y][x] * f2[n][c][y + 2(tj - drad)][x + 2(ti - drad)]     (zero outside)
// The kernel above gives every thread whole output pixels and reads both operands from global memory for each of the
// D*D = 441 displacements: 287 us for the 512x256 frame pair (0.46 GFLOP; profiles/r01_v20_train_kernel_stats.txt).
// Here a workgroup owns TWO output rows of the same parity (oy0, oy0 + 2: their 21 + 21 operand rows y + 2(tj - drad)
// overlap in 20) x 32 columns, and walks the channels in chunks of 8:
//   * stage: the 2 x 32 f1 values and the (D + 1) rows x (32 + 4 drad) columns of f2 for 8 channels go to LDS with
//     coalesced loads; padding is a zero written at staging time; the f2 columns are stored split by PARITY
//     ([even columns | odd columns]), so the D taps x + 2(ti - drad) of one pixel are D CONSECUTIVE floats;
//   * compute: thread (row r, tj, strip) owns 4 same-parity pixels (x, x+2, x+4, x+6) x all D values of ti = 84
//     accumulators; per channel it reads ONE 16-byte-aligned window of 24 floats (6 ds_read_b128) that serves all
//     4 x 21 products -- 3.5 FMAs per LDS float instead of 1 global load per FMA.
// Round 3: the displacement rows are split into groups of CL_TJ = 7 over blockIdx.y (3x the workgroups, 128 threads, 18 KB of
// LDS each, the same staged bytes in total: a group stages its own 7 + 1 f2 rows) -- this planar kernel stays the C-ABI form of
// the op (v2v_correlation_forward); FlowNet2's plan uses correlation_mma_kernel below.
constexpr int CL_CC = 8, CL_TX = 32, CL_ROWS = 2, CL_DMAX = 21, CL_HALF = 36, CL_TJ = 7, CL_NR = CL_TJ + CL_ROWS - 1, CL_THREADS = 128;

__global__ __launch_bounds__(CL_THREADS) void correlation_lds_kernel(const CorrArgs a) {
    __shared__ __attribute__((aligned(16))) float f2s[CL_CC][CL_NR][2 * CL_HALF];
    __shared__ float f1s[CL_CC][CL_ROWS][CL_TX];
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * CL_TX;
    const int ngroups = (a.D + CL_TJ - 1) / CL_TJ;
    const int by = blockIdx.y / ngroups, tj0 = (blockIdx.y - by * ngroups) * CL_TJ;
    const int oy0 = (by >> 1) * 4 + (by & 1);                          // rows oy0 and oy0 + 2
    const int D = a.D, drad = a.drad;
    const int ncols = CL_TX + 4 * drad, nrows = CL_NR;
    const long long hw = (long long)a.H * a.W;
    const float* f1 = a.in1 + (long long)n * a.C * hw;
    const float* f2 = a.in2 + (long long)n * a.C * hw;
    // compute role
    const bool worker = tid < CL_ROWS * CL_TJ * 8;
    const int r = tid / (CL_TJ * 8), u = tid - r * (CL_TJ * 8);
    const int tjl = u >> 3, strip = u & 7;
    const int tj = tj0 + tjl;
    const int par = strip & 1, s4 = strip >> 1;
    const bool active = worker && tj < D;
    float acc[4][CL_DMAX];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < CL_DMAX; ++t) acc[k][t] = 0.f;

    for (int c0 = 0; c0 < a.C; c0 += CL_CC) {
        __syncthreads();                                             // previous chunk fully consumed
        for (int e = tid; e < CL_CC * nrows * ncols; e += CL_THREADS) {
            const int c = e / (nrows * ncols);
            const int rem = e - c * nrows * ncols;
            const int q = rem / ncols, xx = rem - q * ncols;
            const int y = oy0 - 2 * drad + 2 * (tj0 + q), x = ox0 - 2 * drad + xx;
            float v = 0.f;
            if (c0 + c < a.C && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) v = f2[(c0 + c) * hw + (long long)y * a.W + x];
            f2s[c][q][(xx & 1) * CL_HALF + (xx >> 1)] = v;
        }
        for (int e = tid; e < CL_CC * CL_ROWS * CL_TX; e += CL_THREADS) {
            const int c = e / (CL_ROWS * CL_TX);
            const int rem = e - c * CL_ROWS * CL_TX;
            const int rr = rem / CL_TX, px = rem - rr * CL_TX;
            const int y = oy0 + 2 * rr, x = ox0 + px;
            float v = 0.f;
            if (c0 + c < a.C && y < a.H && x < a.W) v = f1[(c0 + c) * hw + (long long)y * a.W + x];
            f1s[c][rr][px] = v;
        }
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int c = 0; c < CL_CC; ++c) {
                // pixels px_k = par + 8*s4 + 2k; their taps live at parity `par`, index 4*s4 + k + t  (t = 0 .. D-1)
                const float4* wv = reinterpret_cast<const float4*>(&f2s[c][r + tjl][par * CL_HALF + 4 * s4]);
                float w[24];
#pragma unroll
                for (int v = 0; v < 6; ++v) { const float4 t4 = wv[v]; w[4 * v] = t4.x; w[4 * v + 1] = t4.y; w[4 * v + 2] = t4.z; w[4 * v + 3] = t4.w; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v1 = f1s[c][r][par + 8 * s4 + 2 * k];
#pragma unroll
                    for (int t = 0; t < CL_DMAX; ++t) acc[k][t] += v1 * w[k + t];
                }
            }
        }
    }
    if (!active) return;
    const int oy = oy0 + 2 * r;
    if (oy >= a.OH) return;
    const float inv = 1.f / (float)a.C;                               // kernel_size 1: nelems = C
    const long long ohw = (long long)a.OH * a.OW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = ox0 + par + 8 * s4 + 2 * k;
        if (ox >= a.OW) continue;
#pragma unroll
        for (int t = 0; t < CL_DMAX; ++t)
            if (t < D) a.out[((long long)n * D * D + tj * D + t) * ohw + (long long)oy * a.OW + ox] = acc[k][t] * inv;
    }
}

// ---- correlation on the matrix pipe, NHWC in / NHWC out (FlowNetC's geometry class: kernel_size 1, stride1 1, stride2 2,
// pad == max_disp; round 3) ----
// The LDS kernel above needs 401 us for the 512x256 frame pair (N = 2, C = 256, 32 x 64 -> 441 x 32 x 64): 32-64 workgroups on
// 256 CUs, each walking 32 channel chunks behind two barriers, in front of an unpack (NHWC -> planar fp32) and behind a pack
// (planar -> NHWC + LeakyReLU) launch.  The work is a banded matrix product: for one output row y, one displacement row tj and
// 32 consecutive pixels x0 .. x0+31 the outputs are
//     out[x][ti] = sum_c f1[y][x][c] * f2[y + 2(tj - drad)][x + 2(ti - drad)][c]
// i.e. the (m, n = m + 2 ti) entries of the 32 x 96 product of the f1 row segment [32 px][C] with the f2 row segment
// [x0 - 2 drad .. x0 - 2 drad + 95][C]: three 32 x 32 MFMA tiles per K step with K = channels, which are CONTIGUOUS in the NHWC
// activations -- every fragment is one 16-byte load per lane straight from global memory / L2 (both feature maps together are
// 2-4 MB), no LDS staging, no barrier.  22-44 % of the products are wanted (2 drad + 1 = 21 of 96 columns per pixel, one
// parity), which the matrix pipe's rate pays for many times over in bf16 and about evens out in fp32
// (v_mfma_f32_32x32x2_f32, exact products).  One wave = (n, y, x tile, tj): N * OH * ceil(OW / 32) * D waves (2688 for the
// frame pair above) instead of 32-64 workgroups.  The epilogue gathers the wanted diagonal band through a 32 x D LDS tile, applies
// 1 / C and the LeakyReLU that follows the correlation in FlowNetC (FlowNetC.py:88-89) and writes D consecutive channels per
// pixel into the NHWC concat buffer of conv3_1 at its channel offset -- the planar 441-channel tensor never exists.
struct CorrMmaArgs {
    const void* f1; const void* f2; void* out;
    int N, C, H, W, cs_in, cs_out, c_off, D, drad;
    float inv_c, leaky;
};

template <typename T> struct CorrFrag;
template <> struct CorrFrag<bf16_t> {
    typedef bf16x8 V;
    static constexpr int KS = 16;                    // channels per MFMA step
    __device__ static __forceinline__ V zero() { V z; for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f; return z; }
    __device__ static __forceinline__ void mma(const V& a, const V& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct CorrFrag<float> {
    typedef f32x4 V;
    static constexpr int KS = 8;                     // lane half h holds channels 4h .. 4h+3 of the step; MFMA j contracts {j, 4 + j}
    __device__ static __forceinline__ V zero() { V z = {0.f, 0.f, 0.f, 0.f}; return z; }
    __device__ static __forceinline__ void mma(const V& a, const V& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};

constexpr int CM_DMAX = 21;

// grid (ceil(OW / 32), OH, N * ceil(D / 4)), block 256: wave w of the block takes tj = 4 * (blockIdx.z % groups) + w
template <typename T>
__global__ __launch_bounds__(256) void correlation_mma_kernel(const CorrMmaArgs a) {
    typedef CorrFrag<T> F;
    typedef typename F::V V;
    __shared__ float tile[4][32 * CM_DMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (a.D + 3) >> 2;
    const int n = blockIdx.z / groups, tj = (blockIdx.z - n * groups) * 4 + wave;
    if (tj >= a.D) return;
    const int y = blockIdx.y, x0 = blockIdx.x * 32;
    const int lr = lane & 31, hi = lane >> 5;
    const int y2 = y + 2 * (tj - a.drad);
    const T* f1 = reinterpret_cast<const T*>(a.f1) + (long long)n * a.H * a.W * a.cs_in;
    const T* f2 = reinterpret_cast<const T*>(a.f2) + (long long)n * a.H * a.W * a.cs_in;
    float* const my = tile[wave];
    const bool row_ok = (unsigned)y2 < (unsigned)a.H;
    if (row_ok) {
        f32x16 acc[3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        constexpr int KE = F::KS / 2;                       // elements per lane and step (16 bytes)
        const int xa = x0 + lr;
        const bool a_ok = xa < a.W;
        const T* pa = f1 + ((long long)y * a.W + (a_ok ? xa : 0)) * a.cs_in + hi * KE;
        const T* pb[3]; bool b_ok[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int xb = x0 - 2 * a.drad + 32 * b + lr;
            b_ok[b] = (unsigned)xb < (unsigned)a.W;
            pb[b] = f2 + ((long long)y2 * a.W + (b_ok[b] ? xb : 0)) * a.cs_in + hi * KE;
        }
        const int steps = a.C / F::KS;
        V fa = a_ok ? *reinterpret_cast<const V*>(pa) : F::zero(), fb[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) fb[b] = b_ok[b] ? *reinterpret_cast<const V*>(pb[b]) : F::zero();
        for (int ks = 0; ks < steps; ++ks) {
            V na = F::zero(), nb[3] = {F::zero(), F::zero(), F::zero()};
            if (ks + 1 < steps) {                            // next step's fragments are in flight under this step's MFMAs
                const int o = (ks + 1) * F::KS;
                if (a_ok) na = *reinterpret_cast<const V*>(pa + o);
#pragma unroll
                for (int b = 0; b < 3; ++b) if (b_ok[b]) nb[b] = *reinterpret_cast<const V*>(pb[b] + o);
            }
#pragma unroll
            for (int b = 0; b < 3; ++b) F::mma(fa, fb[b], acc[b]);
            fa = na;
#pragma unroll
            for (int b = 0; b < 3; ++b) fb[b] = nb[b];
        }
        // the wanted band: output (m, ti) is the product entry (m, n') with n' = m + 2 ti (n' counts f2 columns from x0 - 2 drad)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = (i >> 2) * 8 + hi * 4 + (i & 3);
                const int d = 32 * b + lr - m;
                if (d >= 0 && !(d & 1) && (d >> 1) < a.D) my[m * CM_DMAX + (d >> 1)] = acc[b][i];
            }
    }
    // (wave-private LDS tile: the wave's own writes are visible to its own reads after the LDS counter drains)
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    T* out = reinterpret_cast<T*>(a.out);
    const int total = 32 * a.D;
    for (int e = lane; e < total; e += 64) {
        const int m = e / a.D, ti = e - m * a.D;
        const int x = x0 + m;
        if (x >= a.W) continue;
        float v = row_ok ? my[m * CM_DMAX + ti] * a.inv_c : 0.f;
        v = v > 0.f ? v : v * a.leaky;
        store_act(out, (((long long)n * a.H + y) * a.W + x) * a.cs_out + a.c_off + tj * a.D + ti, v);
    }
}

struct CorrMmaOp : Op {
    CorrMmaArgs a; int dtype;
    int launch(hipStream_t s) override {
        dim3 grid((unsigned)ceil_div(a.W, 32), (unsigned)a.H, (unsigned)(a.N * ((a.D + 3) / 4)));
        if (dtype == V2V_BF16) hipLaunchKernelGGL(correlation_mma_kernel<bf16_t>, grid, dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(correlation_mma_kernel<float>, grid, dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "correlation"; }
};

struct CorrLdsOp : Op {
    CorrArgs a;
    int launch(hipStream_t s) override {
        dim3 grid((unsigned)ceil_div(a.OW, CL_TX), (unsigned)(ceil_div(a.OH, 4) * 2 * ceil_div(a.D, CL_TJ)), (unsigned)a.N);
        hipLaunchKernelGGL(correlation_lds_kernel, grid, dim3(CL_THREADS), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "correlation"; }
};

struct CorrOp : Op {
    CorrArgs a;
    int launch(hipStream_t s) override {
        dim3 grid((unsigned)ceil_div(a.OW, 32), (unsigned)(a.OH * a.D), (unsigned)a.N);
        hipLaunchKernelGGL(correlation_kernel, grid, dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "correlation"; }
};

struct Resample2dArgs { const float* img; const float* flow; float* out; int N, C, H, W, OH, OW, ksize; };

__global__ __launch_bounds__(256) void resample2d_kernel(const Resample2dArgs a) {
    const long long ohw = (long long)a.OH * a.OW;
    const long long total = (long long)a.N * ohw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / ohw, pix = e - b * ohw;
        const int y = (int)(pix / a.OW), x = (int)(pix - (long long)y * a.OW);
        const float dx = a.flow[(b * 2 + 0) * ohw + pix], dy = a.flow[(b * 2 + 1) * ohw + pix];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        // clamps use the OUTPUT extent, exactly as resample2d_kernel.cu:46-49
        const int xL = max(min((int)floorf(xf), a.OW - 1), 0);
        const int xR = max(min((int)floorf(xf) + 1, a.OW - 1), 0);
        const int yT = max(min((int)floorf(yf), a.OH - 1), 0);
        const int yB = max(min((int)floorf(yf) + 1, a.OH - 1), 0);
        for (int c = 0; c < a.C; ++c) {
            const float* ip = a.img + (b * a.C + c) * (long long)a.H * a.W;
            float val = 0.f;
            for (int fy = 0; fy < a.ksize; ++fy)
                for (int fx = 0; fx < a.ksize; ++fx) {
                    val += (1.f - alpha) * (1.f - beta) * ip[(long long)(yT + fy) * a.W + xL + fx];
                    val += alpha * (1.f - beta) * ip[(long long)(yT + fy) * a.W + xR + fx];
                    val += (1.f - alpha) * beta * ip[(long long)(yB + fy) * a.W + xL + fx];
                    val += alpha * beta * ip[(long long)(yB + fy) * a.W + xR + fx];
                }
            a.out[(b * a.C + c) * ohw + pix] = val;
        }
    }
}

struct Resample2dOp : Op {
    Resample2dArgs a;
    int launch(hipStream_t s) override {
        long long blocks = ceil_div((long long)a.N * a.OH * a.OW, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(resample2d_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "resample2d"; }
};

struct ChannelNormArgs { const float* x; float* out; int N, C; long long hw; };

__global__ __launch_bounds__(256) void channelnorm_kernel(const ChannelNormArgs a) {
    const long long total = (long long)a.N * a.hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / a.hw, pix = e - b * a.hw;
        float r = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float v = a.x[(b * a.C + c) * a.hw + pix];
            r += v * v;
        }
        a.out[e] = sqrtf(r);
    }
}

struct ChannelNormOp : Op {
    ChannelNormArgs a;
    int launch(hipStream_t s) override {
        long long blocks = ceil_div((long long)a.N * a.hw, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(channelnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "channelnorm"; }
};


// ---------------------------------------------------------------------------------------
// FlowNet2 glue (models/flownet2_pytorch/models.py:96-161, models/flownet.py:43-59): the
// reference strings these together from ~40 ATen calls per forward (mean, sub, div, cat, Upsample,
// slicing); here each is one streaming kernel writing straight into the consumer's layout.
// ---------------------------------------------------------------------------------------
static inline unsigned fgrid(long long n, long long cap = 4096) {
    long long b = ceil_div(n, 256);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// x = (inputs - rgb_mean) / rgb_max with rgb_mean over both frames and all pixels per (b, c)
// (models.py:97-102).  im1, im2, x1, x2: [B][3][HW]  (x = cat(x1, x2) is only ever used through its halves).
struct NormArgs { const float* im1; const float* im2; float* x1; float* x2; float* partials; int B; long long hw; float inv_rgb_max; int chunks; };

__global__ __launch_bounds__(256) void flownet_mean_partial_kernel(const NormArgs a) {
    __shared__ float sh[256];
    const int plane = blockIdx.y;                 // b*3 + c
    const long long per = (2 * a.hw + a.chunks - 1) / a.chunks;
    const long long e0 = (long long)blockIdx.x * per;
    long long e1 = e0 + per; if (e1 > 2 * a.hw) e1 = 2 * a.hw;
    float s = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256)
        s += e < a.hw ? a.im1[(long long)plane * a.hw + e] : a.im2[(long long)plane * a.hw + (e - a.hw)];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) a.partials[plane * a.chunks + blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void flownet_normalize_kernel(const NormArgs a) {
    __shared__ float sh[256];
    const int oplane = blockIdx.y;                // b*6 + j
    const int b = oplane / 6, j = oplane - b * 6, c = j % 3;
    float s = 0.f;
    for (int i = threadIdx.x; i < a.chunks; i += 256) s += a.partials[(b * 3 + c) * a.chunks + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    const float mean = sh[0] / (float)(2 * a.hw);
    const float* src = (j < 3 ? a.im1 : a.im2) + (long long)(b * 3 + c) * a.hw;
    float* dst = (j < 3 ? a.x1 : a.x2) + (long long)(b * 3 + c) * a.hw;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.hw; e += (long long)gridDim.x * 256)
        dst[e] = (src[e] - mean) * a.inv_rgb_max;
}

struct NormOp : Op {
    NormArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(flownet_mean_partial_kernel, dim3((unsigned)a.chunks, (unsigned)(a.B * 3)), dim3(256), 0, s, a);
        int rc = check_launch(); if (rc) return rc;
        hipLaunchKernelGGL(flownet_normalize_kernel, dim3(fgrid(a.hw, 256), (unsigned)(a.B * 6)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "flownet_normalize"; }
};

// warped = Resample2d(img1, flow) (kernel_size 1); out_norm = ChannelNorm(img0 - warped)  [mode 0]
// or (sum_c (img0 - warped)^2 < thr) as 0/1 (FlowNet.compute_flow_and_conf, flownet.py:55)  [mode 1]
struct WarpDiffArgs { const float* img0; const float* img1; const float* flow; float* warped; float* out_norm;
                      int B, C, H, W; long long bs0, bs1; int mode; float thr; };

__global__ __launch_bounds__(256) void warp_diff_norm_kernel(const WarpDiffArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.B * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / hw, pix = e - b * hw;
        const int y = (int)(pix / a.W), x = (int)(pix - (long long)y * a.W);
        const float dx = a.flow[(b * 2 + 0) * hw + pix], dy = a.flow[(b * 2 + 1) * hw + pix];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
        const int xL = max(min((int)floorf(xf), a.W - 1), 0), xR = max(min((int)floorf(xf) + 1, a.W - 1), 0);
        const int yT = max(min((int)floorf(yf), a.H - 1), 0), yB = max(min((int)floorf(yf) + 1, a.H - 1), 0);
        float r = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float* ip = a.img1 + b * a.bs1 + c * hw;
            float val = 0.f;                       // same operation order as resample2d_kernel
            val += (1.f - alpha) * (1.f - beta) * ip[(long long)yT * a.W + xL];
            val += alpha * (1.f - beta) * ip[(long long)yT * a.W + xR];
            val += (1.f - alpha) * beta * ip[(long long)yB * a.W + xL];
            val += alpha * beta * ip[(long long)yB * a.W + xR];
            if (a.warped) a.warped[(b * a.C + c) * hw + pix] = val;
            const float d = a.img0[b * a.bs0 + c * hw + pix] - val;
            r += d * d;
        }
        if (a.out_norm) a.out_norm[e] = a.mode == 0 ? sqrtf(r) : (r < a.thr ? 1.f : 0.f);
    }
}

struct WarpDiffOp : Op {
    WarpDiffArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(warp_diff_norm_kernel, dim3(fgrid((long long)a.B * a.H * a.W)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "warp_diff_norm"; }
};

// nn.Upsample(scale_factor / size, mode = 'bilinear' (align_corners False) | 'nearest') on planar fp32,
// times out_scale (models.py:49-61,105,117; flownet.py:49-50,56-58)
struct ResizeArgs { const float* x; float* y; long long planes; int H, W, OH, OW, bilinear; float out_scale; };

__global__ __launch_bounds__(256) void resize_planar_kernel(const ResizeArgs a) {
    const long long total = a.planes * a.OH * a.OW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float sh = (float)a.H / (float)a.OH, sw = (float)a.W / (float)a.OW;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ox = (int)(e % a.OW);
        const long long t = e / a.OW;
        const int oy = (int)(t % a.OH);
        const long long pl = t / a.OH;
        const float* xp = a.x + pl * (long long)a.H * a.W;
        float v;
        if (a.bilinear) {
            float fy = sh * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
            float fx = sw * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
            const int y0 = min((int)fy, a.H - 1), x0 = min((int)fx, a.W - 1);
            const int y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            v = (1.f - ly) * ((1.f - lx) * xp[(long long)y0 * a.W + x0] + lx * xp[(long long)y0 * a.W + x1]) +
                ly * ((1.f - lx) * xp[(long long)y1 * a.W + x0] + lx * xp[(long long)y1 * a.W + x1]);
        } else {
            const int y0 = min((int)floorf((float)oy * sh), a.H - 1), x0 = min((int)floorf((float)ox * sw), a.W - 1);
            v = xp[(long long)y0 * a.W + x0];
        }
        a.y[e] = v * a.out_scale;
    }
}

struct ResizeOp : Op {
    ResizeArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(resize_planar_kernel, dim3(fgrid(a.planes * a.OH * a.OW)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "resize_planar"; }
};

// planar fp32 [N][C][HW] -> channels [c_off, c_off+C) of an NHWC activation buffer, y = leaky(x*scale)
struct PackAtArgs { const float* x; void* y; int N, C; long long hw; int c_stride, c_off; float scale, slope; };

template <typename T>
__global__ __launch_bounds__(256) void pack_at_kernel(const PackAtArgs a) {
    T* y = reinterpret_cast<T*>(a.y);
    const long long total = (long long)a.N * a.C * a.hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long pix = e % a.hw;
        const long long nc = e / a.hw;
        const long long n = nc / a.C;
        const int c = (int)(nc - n * a.C);
        float v = a.x[e] * a.scale;
        v = v > 0.f ? v : v * a.slope;
        store_act(y, (n * a.hw + pix) * a.c_stride + a.c_off + c, v);
    }
}

struct PackAtOp : Op {
    PackAtArgs a; int dtype;
    int launch(hipStream_t s) override {
        const long long n = (long long)a.N * a.C * a.hw;
        if (dtype == V2V_BF16) hipLaunchKernelGGL(pack_at_kernel<bf16_t>, dim3(fgrid(n)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(pack_at_kernel<float>, dim3(fgrid(n)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "pack_channels_nhwc"; }
};

// torch.cat along channels of NHWC activations: dst[p][dst_off + c] = src[p][src_off + c]
struct CopyChArgs { const void* src; void* dst; long long P; int C, src_stride, src_off, dst_stride, dst_off; };

template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const CopyChArgs a) {
    const T* src = reinterpret_cast<const T*>(a.src);
    T* dst = reinterpret_cast<T*>(a.dst);
    const long long total = a.P * a.C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long p = e / a.C;
        const int c = (int)(e - p * a.C);
        dst[p * a.dst_stride + a.dst_off + c] = src[p * a.src_stride + a.src_off + c];
    }
}

struct CopyChOp : Op {
    CopyChArgs a; int dtype;
    int launch(hipStream_t s) override {
        if (dtype == V2V_BF16) hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(fgrid(a.P * a.C)), dim3(256), 0, s, a);
        else                   hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(fgrid(a.P * a.C)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "concat_channels_nhwc"; }
};

}  // namespace v2v

using namespace v2v;

extern "C" int v2v_correlation_out_size(int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                        int32_t max_displacement, int32_t stride1, int32_t stride2,
                                        int32_t* out_c, int32_t* out_h, int32_t* out_w) {
    // correlation_cuda.cc:25-38
    const int krad = (kernel_size - 1) / 2, border = krad + max_displacement;
    const int ph = H + 2 * pad_size, pw = W + 2 * pad_size;
    const int d = (max_displacement / stride2) * 2 + 1;
    if (out_c) *out_c = d * d;
    if (out_h) *out_h = (int)ceil_div(ph - 2 * border, stride1);
    if (out_w) *out_w = (int)ceil_div(pw - 2 * border, stride1);
    return 0;
}

extern "C" int v2v_correlation_forward(const float* in1, const float* in2, float* out,
                                       int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                       int32_t stride1, int32_t stride2, int32_t corr_type_multiply, void* stream) {
    if (!in1 || !in2 || !out || stride1 < 1 || stride2 < 1 || kernel_size < 1 || (kernel_size & 1) == 0) {
        set_error("correlation: bad argument"); return V2V_EINVAL;
    }
    if (corr_type_multiply != 1) { set_error("correlation: only corr_type_multiply=1 exists in the reference"); return V2V_EINVAL; }
    CorrArgs a;
    a.in1 = in1; a.in2 = in2; a.out = out; a.N = N; a.C = C; a.H = H; a.W = W;
    a.pad = pad_size; a.ksize = kernel_size; a.krad = (kernel_size - 1) / 2; a.max_disp = max_displacement;
    a.s1 = stride1; a.s2 = stride2; a.drad = max_displacement / stride2; a.D = 2 * a.drad + 1;
    int oc;
    v2v_correlation_out_size(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &a.OH, &a.OW);
    if (a.D > 64 || a.OH <= 0 || a.OW <= 0) { set_error("correlation: unsupported geometry"); return V2V_EINVAL; }
    static const bool no_lds = getenv("V2V_CORR_LDS") && getenv("V2V_CORR_LDS")[0] == '0';
    if (!no_lds && kernel_size == 1 && stride1 == 1 && stride2 == 2 && pad_size == max_displacement && (max_displacement & 1) == 0 &&
        a.drad <= 10 && a.OH == H && a.OW == W) {                     // FlowNetC's geometry class (FlowNetC.py:31): LDS-staged kernel
        auto op2 = std::make_unique<CorrLdsOp>();
        op2->a = a;
        return submit(std::move(op2), stream);
    }
    auto op = std::make_unique<CorrOp>();
    op->a = a;
    return submit(std::move(op), stream);
}

extern "C" int v2v_correlation_nhwc(const void* f1, const void* f2, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                    int32_t cs_in, int32_t cs_out, int32_t c_off, int32_t max_displacement, int32_t stride2,
                                    float leaky_slope, int32_t dtype, void* stream) {
    if (!f1 || !f2 || !out || N < 1 || H < 1 || W < 1 || (dtype != V2V_F32 && dtype != V2V_BF16)) { set_error("correlation_nhwc: bad argument"); return V2V_EINVAL; }
    const int ks = dtype == V2V_BF16 ? 16 : 8, vec = dtype == V2V_BF16 ? 8 : 4;
    const int drad = stride2 > 0 ? max_displacement / stride2 : -1;
    if (stride2 != 2 || (max_displacement & 1) || drad < 1 || 2 * drad + 1 > CM_DMAX || C < ks || C % ks != 0 || cs_in % vec != 0 || cs_in < C ||
        c_off < 0 || cs_out < c_off + (2 * drad + 1) * (2 * drad + 1)) {
        set_error("correlation_nhwc: FlowNetC's geometry class only (kernel 1, stride1 1, stride2 2, pad = max_displacement even, <= 21 x 21 displacements), "
                  "channels a multiple of %d, got C=%d cs_in=%d max_disp=%d stride2=%d", ks, C, cs_in, max_displacement, stride2);
        return V2V_EINVAL;
    }
    auto op = std::make_unique<CorrMmaOp>();
    op->a = CorrMmaArgs{f1, f2, out, N, C, H, W, cs_in, cs_out, c_off, 2 * drad + 1, drad, 1.f / (float)C, leaky_slope};
    op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_resample2d_forward(const float* img, const float* flow, float* out,
                                      int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                      int32_t kernel_size, void* stream) {
    if (!img || !flow || !out || kernel_size < 1) { set_error("resample2d: bad argument"); return V2V_EINVAL; }
    if (OH + kernel_size - 1 > H || OW + kernel_size - 1 > W) { set_error("resample2d: image smaller than flow"); return V2V_EINVAL; }
    auto op = std::make_unique<Resample2dOp>();
    op->a = Resample2dArgs{img, flow, out, N, C, H, W, OH, OW, kernel_size};
    return submit(std::move(op), stream);
}

extern "C" int v2v_channelnorm_forward(const float* x, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                       int32_t norm_deg, void* stream) {
    if (!x || !out) { set_error("channelnorm: null"); return V2V_EINVAL; }
    if (norm_deg != 2) { set_error("channelnorm: the reference kernel implements norm_deg=2 only"); return V2V_EINVAL; }
    auto op = std::make_unique<ChannelNormOp>();
    op->a = ChannelNormArgs{x, out, N, C, (long long)H * W};
    return submit(std::move(op), stream);
}

extern "C" int v2v_flownet_normalize(const float* im1, const float* im2, float* x1, float* x2, float* workspace,
                                     int32_t B, int32_t H, int32_t W, float rgb_max, void* stream) {
    if (!im1 || !im2 || !x1 || !x2 || !workspace || B < 1 || rgb_max == 0.f) { set_error("flownet_normalize: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<NormOp>();
    op->a = NormArgs{im1, im2, x1, x2, workspace, B, (long long)H * W, 1.f / rgb_max, 64};
    return submit(std::move(op), stream);
}

extern "C" int v2v_warp_diff_norm(const float* img0, int64_t batch_stride0, const float* img1, int64_t batch_stride1,
                                  const float* flow, float* warped, float* out_norm, int32_t B, int32_t C, int32_t H,
                                  int32_t W, int32_t mode, float threshold, void* stream) {
    if (!img0 || !img1 || !flow || (!warped && !out_norm) || (mode != 0 && mode != 1)) { set_error("warp_diff_norm: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<WarpDiffOp>();
    op->a = WarpDiffArgs{img0, img1, flow, warped, out_norm, B, C, H, W, batch_stride0, batch_stride1, mode, threshold};
    return submit(std::move(op), stream);
}

extern "C" int v2v_resize_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                 int32_t bilinear, float out_scale, void* stream) {
    if (!x || !y || planes < 1 || H < 1 || W < 1 || OH < 1 || OW < 1) { set_error("resize_planar: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<ResizeOp>();
    op->a = ResizeArgs{x, y, planes, H, W, OH, OW, bilinear, out_scale};
    return submit(std::move(op), stream);
}

extern "C" int v2v_pack_channels_nhwc(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                      int32_t c_stride, int32_t c_offset, float scale, float leaky_slope,
                                      int32_t dtype, void* stream) {
    if (!x || !y || c_offset < 0 || c_offset + C > c_stride) { set_error("pack_channels: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<PackAtOp>();
    op->a = PackAtArgs{x, y, N, C, (long long)H * W, c_stride, c_offset, scale, leaky_slope}; op->dtype = dtype;
    return submit(std::move(op), stream);
}

extern "C" int v2v_concat_channels_nhwc(const void* src, int32_t src_stride, int32_t src_offset, void* dst,
                                        int32_t dst_stride, int32_t dst_offset, int32_t C, int64_t P,
                                        int32_t dtype, void* stream) {
    if (!src || !dst || C < 1 || P < 1 || src_offset + C > src_stride || dst_offset + C > dst_stride) {
        set_error("concat_channels: bad argument"); return V2V_EINVAL;
    }
    auto op = std::make_unique<CopyChOp>();
    op->a = CopyChArgs{src, dst, P, C, src_stride, src_offset, dst_stride, dst_offset}; op->dtype = dtype;
    return submit(std::move(op), stream);
}
