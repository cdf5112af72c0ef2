// 7x7 / stride 1 / reflection-pad-3 stem convolution over ONE-HOT label input as a weight GATHER-SUM (gfx950).
//
// encode_input (models/vid2vid_model_G.py:86-112) turns T = n_frames_G label maps into T x (label_nc one-hot planes + one
// instance-edge plane); the label tower and the foreground tower of CompositeGenerator start with
// ReflectionPad2d(3) + Conv2d(T*(label_nc+1), ngf, 7) on that tensor (models/networks.py:128-133, 153-156).  Dense, that is
// 108 x 49 MACs per output value -- 178 + 89 GFLOP of the 512x256 frame's 2115, and 337 us at the head of the frame's
// critical path (profiles/r02_a8_frame_timeline_serialized.txt).  But per pixel and tap exactly ONE label channel per
// frame is 1, so
//
//     out[p][co] = bias[co] + sum_{tap} sum_{t < T} ( W[co][t*per + label_t(p + tap)][tap]
//                                                     + edge_t(p + tap) * W[co][t*per + label_nc][tap] )
//
// is T gathered weight rows (plus a rare edge row) per tap: 36x fewer operations, exact in fp32.  The one-hot
// tensor is never read -- for n_scales_spatial = 1 it is not even materialised.
//
// Workgroup = 256 threads = 8 x 32 output pixels x one SLICE of CS = 32 or 64 output channels (gridDim.y slices),
// thread = pixel, CS fp32 accumulators in registers.  The packed table is [49 taps][slices][blob]: a blob is the slice's
// [T*per + 1][CS] weight rows (the extra row is zero: out-of-range labels and "no edge" point there, the inner loop has
// no branches), each row PADDED by 16 bytes in global memory already, so that one tap's blob goes to LDS with plain
// LDS-DMA (global_load_lds_dwordx4, no registers), double buffered, one barrier per tap, and lanes that read different
// rows hit different banks (lanes that read the same row -- labels are piecewise constant -- broadcast).  The tile's
// labels / edges (+ 3-pixel halo, reflection applied) sit in LDS as ready-made 16-bit row offsets.
// bf16 rows are accumulated with v_dot2c_f32_bf16 against the constant pairs (1, 0) / (0, 1): unpack + fp32 add in one
// VALU instruction (scripts/ubench/valu_rate.hip: 4.4 cycles vs 4.1 + 2.5 for shift/and + v_add_f32; v_pk_add_f32 is
// slower than two v_add_f32 on gfx950).  The selectors live in SGPRs: as INLINE constants the assembler's "1.0" is not
// what the bf16 operand decodes (measured: wrong sums).
// Epilogue: raw fp32 NHWC output (coalesced through LDS) + the per-tile (sum, sum^2) statistics row of the
// training-mode norm that follows (fixed summation order, no atomics).
#include "v2v_internal.h"

namespace v2v {

struct OneHotConvArgs {
    const void* labels; const void* inst;     // [T][H][W] float or uint8 / int32
    const char* table;                        // [49][slices][blob bytes]
    const float* bias;                        // [cout] or NULL
    float* out; float* stats;                 // raw fp32 NHWC [H][W][cout_stride]; [tiles][cout][2] or NULL
    int T, H, W, label_nc, per, cout, cout_stride, tiles_w, in_u8, blob;
    unsigned sel_lo, sel_hi;                  // bf16 pairs (1, 0) and (0, 1)
    // bf16 only: the instance-edge planes as a small dense GEMM on the matrix pipe (see edge_mfma below)
    const char* etab;                         // [slices][CS / 32][ksteps][64 lanes][8] bf16 B fragments, or NULL: edge rows in the tap loop
    int ksteps, xoff;                         // ceil(49 T / 16); LDS byte offset of the edge bits / k -> offset table / flag
    // training-mode norm finalize by the last workgroup of a channel slice (as v2v_conv2d's fin_*), optional
    int* fin_counter; const float* fin_gamma; const float* fin_beta; float* fin_out; float* fin_rmean; float* fin_rvar;
    float fin_eps, fin_momentum; double fin_inv_count, fin_unbias;
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float dot_sel(unsigned v, unsigned sel, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, v), __builtin_bit_cast(bf16x2_t, sel), acc, false);
}
// LDS-DMA as inline assembly, not __builtin_amdgcn_global_load_lds: the waitcnt pass cannot tell the two table buffers
// apart (run-time offsets) and puts s_waitcnt vmcnt(0) in front of every row read while the next tap's DMA is in flight,
// which serialises copy and compute.  The kernel orders DMA and reads itself (vmcnt(0) + barrier once per tap).
__device__ __forceinline__ void os_glds16(const char* g, unsigned lds_addr) {      // lds_addr: wave-uniform LDS byte address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

constexpr int OS_TH = 8, OS_TW = 32, OS_PH = OS_TH + 6, OS_PW = OS_TW + 6, OS_PHW = OS_PH * OS_PW;
constexpr size_t OS_EPI_LDS = (256 * 33 + 8 * 32 * 2 + 4) * sizeof(float);    // epilogue tile + partial sums + finalize flag

__host__ __device__ constexpr int os_rowb(int cs, int es) { return cs * es + 16; }
__host__ __device__ inline int os_blob(int rows, int cs, int es) { return ((rows + 1) * os_rowb(cs, es) + 1023) / 1024 * 1024; }
// LDS: [2 table blobs][row offsets][edge-row offsets] reused by the epilogue tile, then at `xoff` what must survive into the
// epilogue: edge bits, k -> offset table, flag
inline int os_ksteps(int T) { return (49 * T + 15) / 16; }
inline size_t os_xoff(int T, int blob) {
    size_t m = (size_t)2 * blob + (size_t)2 * T * OS_PHW * sizeof(unsigned short);
    if (m < OS_EPI_LDS) m = OS_EPI_LDS;
    return (m + 15) / 16 * 16;
}
inline size_t os_xbytes(int T, int ksteps) { return (size_t)((T * OS_PHW + 15) & ~15) + (size_t)ksteps * 32 + 16; }
inline size_t os_etab_bytes(int slices, int cs, int ksteps) { return (size_t)slices * ((cs + 31) / 32) * ksteps * 64 * 16; }

// NT: frames per input (n_frames_G), 0 = run-time loop
template <typename T, int CS, int NT>
__global__ __launch_bounds__(256, CS == 64 ? 4 : 6) void onehot_conv7x7_kernel(const OneHotConvArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int ROWB = os_rowb(CS, ES);
    constexpr int VPR = CS * ES / 16;                      // 16-byte vectors per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nT = NT ? NT : a.T;
    const int rows = nT * a.per;
    const unsigned zero_off = (unsigned)rows * ROWB;       // the appended all-zero row
    char* const tb0 = smem;
    char* const tb1 = smem + a.blob;
    unsigned short* const off_s = reinterpret_cast<unsigned short*>(smem + 2 * a.blob);     // [T][PH][PW] row offsets
    unsigned short* const eoff_s = off_s + nT * OS_PHW;                                      // edge-row offset or zero_off
    const bool use_emfma = ES == 2 && a.etab != nullptr && a.inst != nullptr;
    unsigned char* const ebit_s = reinterpret_cast<unsigned char*>(smem + a.xoff);           // [T][PH][PW] 0 / 1 (kept through the epilogue)
    unsigned short* const koff_s = reinterpret_cast<unsigned short*>(smem + a.xoff + ((nT * OS_PHW + 15) & ~15));
    int* const eflag = reinterpret_cast<int*>(koff_s + a.ksteps * 16);
    bool any_e = false;

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int c0 = blockIdx.y * CS;                        // this workgroup's channel slice [c0, c0 + CS)
    const int ty = blockIdx.x / a.tiles_w, tx = blockIdx.x - ty * a.tiles_w;
    const int oh0 = ty * OS_TH, ow0 = tx * OS_TW;
    const int py = tid >> 5, px = tid & 31;
    const int H = a.H, W = a.W;
    const long long hw = (long long)H * W;
    const int npieces = a.blob >> 10;
    const char* const tslice = a.table + (long long)blockIdx.y * a.blob;
    const long long tap_stride = (long long)gridDim.y * a.blob;

    auto issue = [&](int tap, char* dst) {                 // wave w: pieces w, w + 4, ...
        const char* src = tslice + tap * tap_stride + lane * 16;
        const unsigned d = lds_addr_of(dst);
        for (int p = wid; p < npieces; p += 4) os_glds16(src + p * 1024, d + p * 1024);
    };
    issue(0, tb0);
    if (use_emfma && tid == 0) *eflag = 0;
    __syncthreads();

    // ---- row offsets of the tile + halo, reflection applied (ReflectionPad2d(3) of the encoded tensor) ----
    for (int e = tid; e < nT * OS_PHW; e += 256) {
        const int t = e / OS_PHW;
        const int rem = e - t * OS_PHW;
        const int qy = rem / OS_PW, qx = rem - qy * OS_PW;
        int y = oh0 + qy - 3, x = ow0 + qx - 3;
        y = y < 0 ? -y : y;  y = y >= H ? 2 * H - 2 - y : y;
        x = x < 0 ? -x : x;  x = x >= W ? 2 * W - 2 - x : x;
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);              // tile overhang beyond the mirror: any valid pixel, output is masked
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const long long q = (long long)y * W + x;
        const long long p = (long long)t * hw + q;
        int lab;
        bool edge = false;
        if (a.in_u8 == 2) {                                 // v2v_label_codes: label | edge << 7, 127 = no label plane
            const int code = reinterpret_cast<const unsigned char*>(a.labels)[p];
            lab = (code & 127) == 127 ? -1 : (code & 127);
            edge = (code & 128) != 0;
        } else if (a.in_u8) {
            lab = reinterpret_cast<const unsigned char*>(a.labels)[p];
            if (a.inst) {
                const int* ip = reinterpret_cast<const int*>(a.inst) + (long long)t * hw;
                const int ctr = ip[q];
                if (x > 0)     edge |= ip[q - 1] != ctr;
                if (x < W - 1) edge |= ip[q + 1] != ctr;
                if (y > 0)     edge |= ip[q - W] != ctr;
                if (y < H - 1) edge |= ip[q + W] != ctr;
            }
        } else {
            lab = (int)reinterpret_cast<const float*>(a.labels)[p];
            if (a.inst) {
                const float* ip = reinterpret_cast<const float*>(a.inst) + (long long)t * hw;
                const float ctr = ip[q];
                if (x > 0)     edge |= ip[q - 1] != ctr;
                if (x < W - 1) edge |= ip[q + 1] != ctr;
                if (y > 0)     edge |= ip[q - W] != ctr;
                if (y < H - 1) edge |= ip[q + W] != ctr;
            }
        }
        off_s[e] = (unsigned short)((unsigned)lab < (unsigned)a.label_nc ? (unsigned)(t * a.per + lab) * ROWB : zero_off);
        eoff_s[e] = (unsigned short)(edge ? (unsigned)(t * a.per + a.label_nc) * ROWB : zero_off);
        if (use_emfma) { ebit_s[e] = edge ? 1 : 0; any_e |= edge; }
    }
    if (use_emfma) {
        for (int k = tid; k < a.ksteps * 16; k += 256) {    // k = t * 49 + tap -> offset of edge bit (t, dy, dx) relative to a pixel's q0
            const int t = k / 49, tap = k - t * 49;
            koff_s[k] = (unsigned short)(t < nT ? t * OS_PHW + (tap / 7) * OS_PW + (tap % 7) : 0xffff);
        }
        if (__builtin_amdgcn_ballot_w64(any_e) != 0 && lane == 0) *eflag = 1;     // benign race: every writer stores 1
    }

    float acc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[c] = 0.f;
    const unsigned sel_lo = a.sel_lo, sel_hi = a.sel_hi;

    auto add_row = [&](const char* rowp) {
#pragma unroll
        for (int j = 0; j < VPR; ++j) {
            const uint4 v = *reinterpret_cast<const uint4*>(rowp + j * 16);
            if constexpr (ES == 4) {
                acc[j * 4 + 0] += __uint_as_float(v.x); acc[j * 4 + 1] += __uint_as_float(v.y);
                acc[j * 4 + 2] += __uint_as_float(v.z); acc[j * 4 + 3] += __uint_as_float(v.w);
            } else {
                acc[j * 8 + 0] = dot_sel(v.x, sel_lo, acc[j * 8 + 0]); acc[j * 8 + 1] = dot_sel(v.x, sel_hi, acc[j * 8 + 1]);
                acc[j * 8 + 2] = dot_sel(v.y, sel_lo, acc[j * 8 + 2]); acc[j * 8 + 3] = dot_sel(v.y, sel_hi, acc[j * 8 + 3]);
                acc[j * 8 + 4] = dot_sel(v.z, sel_lo, acc[j * 8 + 4]); acc[j * 8 + 5] = dot_sel(v.z, sel_hi, acc[j * 8 + 5]);
                acc[j * 8 + 6] = dot_sel(v.w, sel_lo, acc[j * 8 + 6]); acc[j * 8 + 7] = dot_sel(v.w, sel_hi, acc[j * 8 + 7]);
            }
        }
    };

    const int q0 = py * OS_PW + px;
    for (int tap = 0; tap < 49; ++tap) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab `tap` (issued a whole tap ago)
        __syncthreads();                                   // slab `tap` complete (+ offsets, first round); buffer of tap-1 free
        char* const cur = (tap & 1) ? tb1 : tb0;
        if (tap + 1 < 49) issue(tap + 1, (tap & 1) ? tb0 : tb1);
        const int dy = tap / 7, dx = tap - dy * 7;
        const int q = q0 + dy * OS_PW + dx;
        if constexpr (NT > 0) {
            unsigned o[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = off_s[t * OS_PHW + q];
#pragma unroll
            for (int t = 0; t < NT; ++t) add_row(cur + o[t]);
            if (a.inst && !use_emfma) {
                unsigned e[NT];
                bool any = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) { e[t] = eoff_s[t * OS_PHW + q]; any |= e[t] != zero_off; }
                if (__builtin_amdgcn_ballot_w64(any) != 0) {      // wave-uniform: most waves see no edge pixel at this tap
#pragma unroll
                    for (int t = 0; t < NT; ++t) add_row(cur + e[t]);
                }
            }
        } else {
            for (int t = 0; t < nT; ++t) {
                add_row(cur + off_s[t * OS_PHW + q]);
                if (a.inst && !use_emfma) {
                    const unsigned e = eoff_s[t * OS_PHW + q];
                    if (__builtin_amdgcn_ballot_w64(e != zero_off) != 0) add_row(cur + e);
                }
            }
        }
    }

    // ---- epilogue: bias, then 32-channel chunks through LDS: coalesced raw fp32 NHWC store + per-tile statistics ----
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < CS; ++c) acc[c] += c0 + c < a.cout ? a.bias[c0 + c] : 0.f;
    }
    __syncthreads();                                       // the table buffers are free
    float* const tile = reinterpret_cast<float*>(smem);    // [256 pixels][33]: stride 33 keeps the per-pixel writes conflict-free
    float* const red = tile + 256 * 33;                    // [8 parts][32][2]
    const bool valid = oh0 + py < H && ow0 + px < W;
    const bool ragged = oh0 + OS_TH > H || ow0 + OS_TW > W;
    const bool do_emfma = use_emfma && *eflag != 0;        // no edge pixel in the tile + halo: nothing to add
    constexpr int NCH = (CS + 31) / 32, CW = CS < 32 ? CS : 32;       // 32-channel chunks of the slice; real channels per chunk (16-channel slices: 16)
    // (round 5 tried the tile in two halves of 128 pixels -- 26.8 KB of LDS per workgroup, 6 instead of 4 workgroups per CU: 6-13 % SLOWER on
    //  the 512x256 stems, profiles/r05_v16_ohab.txt: occupancy is not what this kernel lacks; and a third table buffer -- the blob of tap
    //  t + 2 in flight, counted vmcnt, raw s_barrier: 2-4 % slower, profiles/r05_v18_ohab.txt: nor is the blob's latency)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
        for (int i = 0; i < CW; ++i) tile[tid * 33 + i] = acc[k * 32 + i];
        if constexpr (ES == 2) {
            if (do_emfma) {
                // edge planes of this 32-channel chunk: E[pixel][c] = sum_k e[pixel][k] W_edge[k][c], k = (frame, tap), as
                // v_mfma_f32_32x32x16_bf16: A = 0 / 1 edge bits of the wave's 2 x 32 pixels gathered from LDS, B = the packed
                // edge-row fragments (global, L2-resident), C added to the pixel-major tile (one owner lane per element)
                f32x16_t cm[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cm[m][r] = 0.f;
                const int kg = lane >> 5;
                const char* bsrc = a.etab + (((long long)blockIdx.y * NCH + k) * a.ksteps * 64 + lane) * 16;
                const int qm0 = (2 * wid) * OS_PW + (lane & 31);
                for (int ks = 0; ks < a.ksteps; ++ks) {
                    const bf16x8_t bfrag = *reinterpret_cast<const bf16x8_t*>(bsrc + (long long)ks * 64 * 16);
                    const uint4 ko = *reinterpret_cast<const uint4*>(koff_s + ks * 16 + kg * 8);
                    const unsigned kk[8] = {ko.x & 0xffffu, ko.x >> 16, ko.y & 0xffffu, ko.y >> 16, ko.z & 0xffffu, ko.z >> 16, ko.w & 0xffffu, ko.w >> 16};
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int qm = qm0 + m * OS_PW;
                        unsigned w4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned lo = kk[2 * j] != 0xffffu && ebit_s[kk[2 * j] + qm] ? 0x3f80u : 0u;
                            const unsigned hi = kk[2 * j + 1] != 0xffffu && ebit_s[kk[2 * j + 1] + qm] ? 0x3f800000u : 0u;
                            w4[j] = lo | hi;
                        }
                        const uint4 au = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        cm[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, au), bfrag, cm[m], 0, 0, 0);
                    }
                }
                __syncthreads();                           // every thread's own accumulators are in the tile
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int prow = 8 * (r >> 2) + 4 * kg + (r & 3);          // C layout of the 32x32 MFMA: row, col = lane & 31
                        if ((lane & 31) < CW) tile[(wid * 64 + m * 32 + prow) * 33 + (lane & 31)] += cm[m][r];
                    }
            }
        }
        __syncthreads();
        if (!valid) {                                      // pixels beyond the image: out of the statistics
#pragma unroll
            for (int i = 0; i < CW; ++i) tile[tid * 33 + i] = 0.f;
        }
        if (ragged) __syncthreads();                       // (uniform: the tile overhangs the image)
#pragma unroll
        for (int it = 0; it < 8; ++it) {                   // 8 lanes x float4 = the 128 bytes of one pixel's chunk
            const int idx = it * 256 + tid;
            const int p = idx >> 3, c4 = (idx & 7) * 4;
            const int oh = oh0 + (p >> 5), ow = ow0 + (p & 31);
            const int c = c0 + k * 32 + c4;
            if (oh < H && ow < W && c4 < CW && c < a.cout) {
                float* op = a.out + ((long long)oh * W + ow) * a.cout_stride + c;
                const float* tp = tile + p * 33 + c4;
                if (c + 4 <= a.cout && (a.cout_stride & 3) == 0) *reinterpret_cast<float4*>(op) = make_float4(tp[0], tp[1], tp[2], tp[3]);
                else for (int e = 0; e < 4 && c + e < a.cout; ++e) op[e] = tp[e];
            }
        }
        if (a.stats) {                                     // thread (channel tid & 31, part tid >> 5): 32 pixels in order
            const int c = tid & 31, part = tid >> 5;
            float s1 = 0.f, s2 = 0.f;
            if (c < CW) {
#pragma unroll 8
                for (int i = 0; i < 32; ++i) { const float v = tile[(part * 32 + i) * 33 + c]; s1 += v; s2 += v * v; }
            }
            red[(part * 32 + c) * 2] = s1;
            red[(part * 32 + c) * 2 + 1] = s2;
        }
        __syncthreads();
        if (a.stats && tid < CW && c0 + k * 32 + tid < a.cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int part = 0; part < 8; ++part) { t1 += red[(part * 32 + tid) * 2]; t2 += red[(part * 32 + tid) * 2 + 1]; }
            float* dst = a.stats + ((long long)blockIdx.x * a.cout + c0 + k * 32 + tid) * 2;
            if (a.fin_counter != nullptr) {          // read back by the finalizing workgroup: agent-scope write-through store
                const unsigned long long bits = (unsigned long long)__float_as_uint(t1) | ((unsigned long long)__float_as_uint(t2) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                dst[0] = t1;
                dst[1] = t2;
            }
        }
    }
    if (a.stats == nullptr || a.fin_counter == nullptr) return;
    // ---- norm finalize by the LAST workgroup of this channel slice (get_norm_layer, models/networks.py:23-30): replaces a
    //      512-row single-workgroup bn_finalize launch (34 us on the frame's critical path, profiles/r02_a22_frame_timeline*) ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(red + 8 * 32 * 2);
    if (tid == 0) {
        const int tk = __hip_atomic_fetch_add(a.fin_counter + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == (int)gridDim.x - 1 ? 1 : 0;
        if (last) __hip_atomic_store(a.fin_counter + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    constexpr int PH = 256 / CS;
    double* const acc2 = reinterpret_cast<double*>(smem);        // [PH][CS][2] <= 8 KiB
    {
        const int c = tid % CS, ph = tid / CS;
        const int ncol = c0 + c;
        double s1 = 0.0, s2 = 0.0;
        if (ncol < a.cout) sum_stat_rows<16>(a.stats, (long long)ncol * 2, (long long)a.cout * 2, ph, PH, (int)gridDim.x, s1, s2);
        acc2[(ph * CS + c) * 2 + 0] = s1;
        acc2[(ph * CS + c) * 2 + 1] = s2;
        __syncthreads();
        if (ph == 0 && ncol < a.cout) {
            s1 = 0.0; s2 = 0.0;
#pragma unroll
            for (int q = 0; q < PH; ++q) { s1 += acc2[(q * CS + c) * 2 + 0]; s2 += acc2[(q * CS + c) * 2 + 1]; }
            const double mean = s1 * a.fin_inv_count;
            double var = s2 * a.fin_inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / sqrt(var + (double)a.fin_eps);
            const double g = a.fin_gamma ? (double)a.fin_gamma[ncol] : 1.0;
            const double b = a.fin_beta ? (double)a.fin_beta[ncol] : 0.0;
            const double sc = g * invstd;
            a.fin_out[ncol] = (float)sc;
            a.fin_out[a.cout + ncol] = (float)(b - mean * sc);
            a.fin_out[2 * a.cout + ncol] = (float)mean;
            a.fin_out[3 * a.cout + ncol] = (float)invstd;
            if (a.fin_rmean) a.fin_rmean[ncol] = (1.f - a.fin_momentum) * a.fin_rmean[ncol] + a.fin_momentum * (float)mean;
            if (a.fin_rvar)  a.fin_rvar[ncol]  = (1.f - a.fin_momentum) * a.fin_rvar[ncol] + a.fin_momentum * (float)(var * a.fin_unbias);
        }
    }
}

struct OneHotConvOp : Op {
    OneHotConvArgs a; int dtype, cs, slices, tiles;
    template <typename T, int CS, int NT> int go(hipStream_t s) {
        const size_t lds = (size_t)a.xoff + os_xbytes(a.T, a.ksteps);
        auto kern = onehot_conv7x7_kernel<T, CS, NT>;
        static bool attr_done = false;
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)slices), dim3(256), (lds + 15) / 16 * 16, s, a);
        return check_launch();
    }
    template <typename T, int CS> int go_t(hipStream_t s) { return a.T == 3 ? go<T, CS, 3>(s) : go<T, CS, 0>(s); }
    int launch(hipStream_t s) override {
        if (dtype == V2V_BF16) return cs == 64 ? go_t<bf16_t, 64>(s) : cs == 16 ? go_t<bf16_t, 16>(s) : go_t<bf16_t, 32>(s);
        return cs == 64 ? go_t<float, 64>(s) : cs == 16 ? go_t<float, 16>(s) : go_t<float, 32>(s);
    }
    const char* name() const override { return "onehot_conv7x7"; }
};

// weights [cout][cin][7][7] fp32 -> table [49][slices][blob]; blob rows [cin + 1][cs] + 16 bytes of pad per row, zero-filled
struct OneHotPackArgs { const float* w; char* tab; int cin, cout, cs, slices, blob, dtype; };

__global__ __launch_bounds__(256) void onehot_pack_kernel(const OneHotPackArgs a) {
    const int es = a.dtype == V2V_BF16 ? 2 : 4;
    const int per_blob = a.blob / es;                      // elements, pads included
    const int rowe = a.cs + 16 / es;
    const long long total = 49ll * a.slices * per_blob;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int within = (int)(e % per_blob);
        const long long b = e / per_blob;
        const int sl = (int)(b % a.slices), tap = (int)(b / a.slices);
        const int r = within / rowe, c = within - r * rowe;
        const int co = sl * a.cs + c;
        const float v = (r < a.cin && c < a.cs && co < a.cout) ? a.w[((long long)co * a.cin + r) * 49 + tap] : 0.f;
        if (a.dtype == V2V_BF16) reinterpret_cast<unsigned short*>(a.tab)[e] = f32_to_bf16_bits(v);
        else                     reinterpret_cast<float*>(a.tab)[e] = v;
    }
}

// edge-row B fragments for the bf16 matrix pipe: etab[slice][n-tile][kstep][lane][j] = W[co][t * per + label_nc][tap] with
// co = slice * cs + ntile * 32 + (lane & 31), k = kstep * 16 + (lane >> 5) * 8 + j = t * 49 + tap (zero beyond 49 T)
struct OneHotEdgePackArgs { const float* w; unsigned short* etab; int cin, cout, cs, slices, ksteps, T, per, label_nc; };

__global__ __launch_bounds__(256) void onehot_edge_pack_kernel(const OneHotEdgePackArgs a) {
    const int nch = (a.cs + 31) / 32;                          // 32-column MFMA tiles per slice (16-channel slices: the upper 16 columns are zero)
    const long long total = (long long)a.slices * nch * a.ksteps * 64 * 8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        long long r = e >> 9;
        const int ks = (int)(r % a.ksteps); r /= a.ksteps;
        const int ntile = (int)(r % nch), sl = (int)(r / nch);
        const int cin_slice = ntile * 32 + (lane & 31);
        const int co = sl * a.cs + cin_slice;
        const int k = ks * 16 + (lane >> 5) * 8 + j;
        const int t = k / 49, tap = k - t * 49;
        float v = 0.f;
        if (t < a.T && cin_slice < a.cs && co < a.cout) v = a.w[((long long)co * a.cin + t * a.per + a.label_nc) * 49 + tap];
        a.etab[e] = f32_to_bf16_bits(v);
    }
}

struct OneHotPackOp : Op {
    OneHotPackArgs a; OneHotEdgePackArgs e; bool edges;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(onehot_pack_kernel, dim3(1024), dim3(256), 0, s, a);
        int rc = check_launch();
        if (rc != 0 || !edges) return rc;
        hipLaunchKernelGGL(onehot_edge_pack_kernel, dim3(256), dim3(256), 0, s, e);
        return check_launch();
    }
    const char* name() const override { return "onehot_pack_weights"; }
};

// label | edge << 7 per (frame, pixel): the stems' staging then is one byte per halo entry instead of a label load and five
// instance-map loads in every one of the 2048 workgroups (512 tiles x 4 slices)
struct LabelCodeArgs { const void* labels; const void* inst; unsigned char* codes; int T, H, W, label_nc, in_u8; };

__global__ __launch_bounds__(256) void label_codes_kernel(const LabelCodeArgs a) {
    const long long hw = (long long)a.H * a.W, total = hw * a.T;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long t = e / hw, q = e - t * hw;
    const int y = (int)(q / a.W), x = (int)(q - (long long)y * a.W);
    int lab;
    bool edge = false;
    if (a.in_u8) {
        lab = reinterpret_cast<const unsigned char*>(a.labels)[e];
        if (a.inst) {
            const int* ip = reinterpret_cast<const int*>(a.inst) + t * hw;
            const int ctr = ip[q];
            if (x > 0)       edge |= ip[q - 1] != ctr;
            if (x < a.W - 1) edge |= ip[q + 1] != ctr;
            if (y > 0)       edge |= ip[q - a.W] != ctr;
            if (y < a.H - 1) edge |= ip[q + a.W] != ctr;
        }
    } else {
        lab = (int)reinterpret_cast<const float*>(a.labels)[e];
        if (a.inst) {
            const float* ip = reinterpret_cast<const float*>(a.inst) + t * hw;
            const float ctr = ip[q];
            if (x > 0)       edge |= ip[q - 1] != ctr;
            if (x < a.W - 1) edge |= ip[q + 1] != ctr;
            if (y > 0)       edge |= ip[q - a.W] != ctr;
            if (y < a.H - 1) edge |= ip[q + a.W] != ctr;
        }
    }
    a.codes[e] = (unsigned char)(((unsigned)lab < (unsigned)a.label_nc ? lab : 127) | (edge ? 128 : 0));
}

struct LabelCodeOp : Op {
    LabelCodeArgs a;
    int launch(hipStream_t s) override {
        hipLaunchKernelGGL(label_codes_kernel, dim3((unsigned)ceil_div((long long)a.T * a.H * a.W, 256)), dim3(256), 0, s, a);
        return check_launch();
    }
    const char* name() const override { return "label_codes"; }
};

// slice width: 0 = default = 32 channels (measured: 153 us vs 181 us for the 108 -> 128 stem with 64-channel slices,
// profiles/r02_a23_stem_bench.txt: 68 VGPRs -> 6 waves per SIMD hide the LDS latency of the row gathers), or 32 / 64 as given
// Round 5: layers with <= 16 output channels (the finest foreground tower's 108 -> 16 stem at 2048x1024) take 16-channel slices by
// default: the kernel's time is the row gathers x slice width, a 32-wide slice spent half of them on padding columns (590 us, the
// same as the 108 -> 32 stem beside it).
static int slice_of(int cout, int dtype, int slice) {
    (void)dtype;
    return slice == 64 ? 64 : (slice == 0 && cout <= 16) ? 16 : 32;
}
static bool onehot_args_ok(int cin, int cout, int dtype, int slice) {
    return cin >= 1 && cout >= 1 && cout <= 128 && (dtype == V2V_F32 || dtype == V2V_BF16) && (slice == 0 || slice == 32 || slice == 64);
}

}  // namespace v2v

using namespace v2v;

// the edge planes ride the matrix pipe when the layer has them (cin == T * (label_nc + 1)) and the table is bf16
static bool has_edge_tab(int cin, int dtype, int T, int label_nc) { return dtype == V2V_BF16 && T >= 1 && cin == T * (label_nc + 1); }

extern "C" int64_t v2v_onehot_conv_table_bytes(int32_t cin, int32_t cout, int32_t dtype, int32_t slice, int32_t T, int32_t label_nc) {
    if (!onehot_args_ok(cin, cout, dtype, slice) || T < 1 || label_nc < 1 || (cin != T * label_nc && cin != T * (label_nc + 1))) return V2V_EINVAL;
    const int cs = slice_of(cout, dtype, slice), es = dtype == V2V_BF16 ? 2 : 4;
    const int blob = os_blob(cin, cs, es);
    if ((cin + 1) * os_rowb(cs, es) > 65535) return V2V_EINVAL;        // 16-bit row offsets
    const int slices = (int)ceil_div(cout, cs);
    return 49ll * slices * blob + (has_edge_tab(cin, dtype, T, label_nc) ? (int64_t)os_etab_bytes(slices, cs, os_ksteps(T)) : 0);
}

extern "C" int v2v_onehot_conv_pack_weights(const float* w, void* table, int32_t cin, int32_t cout, int32_t dtype, int32_t slice,
                                            int32_t T, int32_t label_nc, void* stream) {
    if (!w || !table || !onehot_args_ok(cin, cout, dtype, slice) || T < 1 || label_nc < 1 || (cin != T * label_nc && cin != T * (label_nc + 1))) {
        set_error("onehot_conv_pack_weights: bad argument (cout <= 128, slice 0 / 32 / 64, cin = T * label_nc or T * (label_nc + 1))"); return V2V_EINVAL;
    }
    const int cs = slice_of(cout, dtype, slice), es = dtype == V2V_BF16 ? 2 : 4;
    const int slices = (int)ceil_div(cout, cs), blob = os_blob(cin, cs, es);
    auto op = std::make_unique<OneHotPackOp>();
    op->a = OneHotPackArgs{w, reinterpret_cast<char*>(table), cin, cout, cs, slices, blob, dtype};
    op->edges = has_edge_tab(cin, dtype, T, label_nc);
    op->e = OneHotEdgePackArgs{w, reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(table) + 49ll * slices * blob),
                               cin, cout, cs, slices, os_ksteps(T), T, label_nc + 1, label_nc};
    return submit(std::move(op), stream);
}

extern "C" int v2v_label_codes(const void* labels, const void* inst, int32_t in_u8, uint8_t* codes, int32_t T, int32_t H, int32_t W,
                               int32_t label_nc, void* stream) {
    if (!labels || !codes || T < 1 || H < 1 || W < 1 || label_nc < 1 || label_nc > 126 || (in_u8 != 0 && in_u8 != 1)) {
        set_error("label_codes: bad argument (label_nc <= 126)"); return V2V_EINVAL;
    }
    auto op = std::make_unique<LabelCodeOp>();
    op->a = LabelCodeArgs{labels, inst, codes, T, H, W, label_nc, in_u8};
    return submit(std::move(op), stream);
}

extern "C" int v2v_onehot_conv_stats_rows(int32_t H, int32_t W) {
    if (H < 1 || W < 1) return V2V_EINVAL;
    return (int)(ceil_div(H, OS_TH) * ceil_div(W, OS_TW));
}

static int onehot_conv_submit(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                              float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                              int32_t cout, int32_t cout_stride, int32_t dtype, int32_t slice, const v2v_onehot_norm* fin, void* stream) {
    if (!labels || !table || !out || T < 1 || H < 4 || W < 4 || label_nc < 1 || cout_stride < cout || !onehot_args_ok(1, cout, dtype, slice) ||
        in_u8 < 0 || in_u8 > 2 || (in_u8 == 2 && label_nc > 126)) {
        set_error("onehot_conv7x7: bad argument (cout <= 128, slice 0 / 32 / 64, image at least 4x4 for the 3-pixel mirror)"); return V2V_EINVAL;
    }
    if (((uintptr_t)table | (uintptr_t)out) & 15) { set_error("onehot_conv7x7: table / output must be 16-byte aligned"); return V2V_EINVAL; }
    const int per = label_nc + (inst ? 1 : 0);
    const int cs = slice_of(cout, dtype, slice), es = dtype == V2V_BF16 ? 2 : 4;
    const int rows = T * per;
    const int blob = os_blob(rows, cs, es);
    const int ksteps = os_ksteps(T);
    const size_t lds = os_xoff(T, blob) + os_xbytes(T, ksteps);
    if ((rows + 1) * os_rowb(cs, es) > 65535 || lds > 156 * 1024 || (size_t)T * OS_PHW + 49 * OS_PW > 65535) {
        set_error("onehot_conv7x7: T * (label_nc + 1) = %d weight rows do not fit the LDS", rows); return V2V_EINVAL;
    }
    auto op = std::make_unique<OneHotConvOp>();
    op->a = OneHotConvArgs{labels, inst, reinterpret_cast<const char*>(table), bias, out, stats, T, H, W, label_nc, per, cout, cout_stride,
                           (int)ceil_div(W, OS_TW), in_u8, blob, 0x00003f80u, 0x3f800000u,
                           (inst && has_edge_tab(rows, dtype, T, label_nc))
                               ? reinterpret_cast<const char*>(table) + 49ll * ceil_div(cout, cs) * blob : nullptr,
                           ksteps, (int)os_xoff(T, blob),
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.0, 0.0};
    if (fin != nullptr) {
        if (!stats || !fin->counter || !fin->scale_shift || fin->count <= 0) {
            set_error("onehot_conv7x7: in-kernel norm finalize needs stats, counter (>= 4 zero ints), scale_shift and count"); return V2V_EINVAL;
        }
        OneHotConvArgs& a = op->a;
        a.fin_counter = fin->counter; a.fin_gamma = fin->gamma; a.fin_beta = fin->beta; a.fin_out = fin->scale_shift;
        a.fin_rmean = fin->running_mean; a.fin_rvar = fin->running_var; a.fin_eps = fin->eps; a.fin_momentum = fin->momentum;
        a.fin_inv_count = 1.0 / (double)fin->count;
        a.fin_unbias = fin->count > 1 ? (double)fin->count / (double)(fin->count - 1) : 1.0;
    }
    op->dtype = dtype; op->cs = cs; op->slices = (int)ceil_div(cout, cs);
    op->tiles = (int)(ceil_div(H, OS_TH) * ceil_div(W, OS_TW));
    return submit(std::move(op), stream);
}

extern "C" int v2v_onehot_conv7x7(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                                  float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                                  int32_t cout, int32_t cout_stride, int32_t dtype, int32_t slice, void* stream) {
    return onehot_conv_submit(labels, inst, in_u8, table, bias, out, stats, T, H, W, label_nc, cout, cout_stride, dtype, slice, nullptr, stream);
}

extern "C" int v2v_onehot_conv7x7_norm(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                                       float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                                       int32_t cout, int32_t cout_stride, int32_t dtype, int32_t slice, const v2v_onehot_norm* fin,
                                       void* stream) {
    if (!fin) { set_error("onehot_conv7x7_norm: fin is NULL"); return V2V_EINVAL; }
    return onehot_conv_submit(labels, inst, in_u8, table, bias, out, stats, T, H, W, label_nc, cout, cout_stride, dtype, slice, fin, stream);
}
