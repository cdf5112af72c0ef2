// Instruction issue rates on gfx950 (cycles per wave64 instruction per SIMD), measured: the one-hot stem kernel's choice of
// unpack + accumulate instruction hangs on these.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    float b0 = 0, b1 = 1, b2 = 2, b3 = 3, b4 = 4, b5 = 5, b6 = 6, b7 = 7;
    unsigned s = 0x3f803f80u + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {          // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                               "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        } else if constexpr (MODE == 1) {   // v_dot2c_f32_bf16
            REP16(asm volatile("v_dot2c_f32_bf16 %0, 1.0, %8\n v_dot2c_f32_bf16 %1, 1.0, %8\n v_dot2c_f32_bf16 %2, 1.0, %8\n v_dot2c_f32_bf16 %3, 1.0, %8\n"
                               "v_dot2c_f32_bf16 %4, 1.0, %8\n v_dot2c_f32_bf16 %5, 1.0, %8\n v_dot2c_f32_bf16 %6, 1.0, %8\n v_dot2c_f32_bf16 %7, 1.0, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        } else if constexpr (MODE == 2) {   // v_pk_add_f32 (two fp32 adds per lane per instruction)
            REP16(asm volatile("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                               "v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                               : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
        } else if constexpr (MODE == 3) {   // v_fma_mix_f32: fp16 half (op_sel) * 1.0 + fp32
            REP16(asm volatile("v_fma_mix_f32 %0, %8, 1.0, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %8, 1.0, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %8, 1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, 1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %8, 1.0, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, 1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        } else if constexpr (MODE == 4) {   // unpack: v_lshlrev_b32 / v_and_b32
            REP16(asm volatile("v_lshlrev_b32 %0, 16, %8\n v_and_b32 %1, 0xffff0000, %8\n v_lshlrev_b32 %2, 16, %8\n v_and_b32 %3, 0xffff0000, %8\n"
                               "v_lshlrev_b32 %4, 16, %8\n v_and_b32 %5, 0xffff0000, %8\n v_lshlrev_b32 %6, 16, %8\n v_and_b32 %7, 0xffff0000, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        } else if constexpr (MODE == 5) {   // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n"
                               "v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n"
                               : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
        } else if constexpr (MODE == 6) {   // v_dot2_f32_f16 (VOP3P fdot2)
            REP16(asm volatile("v_dot2_f32_f16 %0, %8, %8, %0\n v_dot2_f32_f16 %1, %8, %8, %1\n v_dot2_f32_f16 %2, %8, %8, %2\n v_dot2_f32_f16 %3, %8, %8, %3\n"
                               "v_dot2_f32_f16 %4, %8, %8, %4\n v_dot2_f32_f16 %5, %8, %8, %5\n v_dot2_f32_f16 %6, %8, %8, %6\n v_dot2_f32_f16 %7, %8, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

// LDS read throughput: MODE 0 all lanes one address (broadcast), 1 lane-distinct padded rows, 2 lane-linear 16 B
template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr = MODE == 0 ? 0u : MODE == 1 ? (unsigned)(lane % 36) * 272u : (unsigned)lane * 16u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
        float4 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                     "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr));
        acc.x += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

template <typename K>
static double run(K kern, int wgs, int iters, size_t lds, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, %.2f GHz (clockRate)\n", p.name, cus, ghz);
    const int iters = 2000;
    const char* names[] = {"v_add_f32", "v_dot2c_f32_bf16", "v_pk_add_f32", "v_fma_mix_f32 (f16 half + f32)", "v_lshlrev_b32 / v_and_b32", "v_pk_fma_f32", "v_dot2_f32_f16"};
    for (int wps = 1; wps <= 2; ++wps) {          // waves per SIMD
        const int wgs = cus * wps;                // 256 threads = 4 waves = 1 per SIMD
        double ms[7] = {run(rate_kernel<0>, wgs, iters, 0, out), run(rate_kernel<1>, wgs, iters, 0, out), run(rate_kernel<2>, wgs, iters, 0, out),
                        run(rate_kernel<3>, wgs, iters, 0, out), run(rate_kernel<4>, wgs, iters, 0, out), run(rate_kernel<5>, wgs, iters, 0, out),
                        run(rate_kernel<6>, wgs, iters, 0, out)};
        for (int m = 0; m < 7; ++m)
            printf("%d wave(s)/SIMD  %-32s %8.3f ms  -> %.2f cycles per wave64 instruction per SIMD\n", wps, names[m], ms[m],
                   ms[m] * 1e-3 * ghz * 1e9 / ((double)iters * 128 * wps));
    }
    const char* ln[] = {"ds_read_b128, all lanes one address", "ds_read_b128, lane-distinct padded rows (272 B stride)", "ds_read_b128, lane-linear"};
    for (int wps = 1; wps <= 2; ++wps) {
        const int wgs = cus * wps;
        double ms[3] = {run(lds_kernel<0>, wgs, iters, 65536, out), run(lds_kernel<1>, wgs, iters, 65536, out), run(lds_kernel<2>, wgs, iters, 65536, out)};
        for (int m = 0; m < 3; ++m) {
            const double cyc = ms[m] * 1e-3 * ghz * 1e9 / ((double)iters * 8 * wps * 4);      // per wave-instruction per CU
            printf("%d wave(s)/SIMD  %-55s %8.3f ms  -> %.1f cycles per wave instruction per CU = %.0f B/clk/CU\n", wps, ln[m], ms[m], cyc, 1024.0 / cyc);
        }
    }
    return 0;
}
