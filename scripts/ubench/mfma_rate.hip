// What the whole chip shares when every CU runs the matrix pipe (VERDICT r4 item 3 iii): a pure-register
// v_mfma_f32_32x32x16_bf16 loop -- no memory, no LDS, no barrier -- on G workgroups of 4 or 8 waves, G = 64 / 128 / 256 / 512.
// Reports achieved TFLOP/s against the 2500 TFLOP/s of MI355X_MICROARCH.md and the shader clock seen by the waves
// (s_memtime ticks / s_memrealtime ticks at 100 MHz), i.e. how far the clock (power) budget lets a full-chip MFMA stream go.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.hip -o scripts/mfma_rate.bin && scripts/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_kernel(float* out, unsigned long long* clk, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    const unsigned long long r0 = wall_clock64();                    // s_memrealtime: 100 MHz
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1024 * 512 * sizeof(float));
    hipMalloc(&clk, 2 * 1024 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    printf("# pure-register v_mfma_f32_32x32x16_bf16 stream, %d x 16 MFMAs per wave; peak 2500 TFLOP/s = 256 CUs x 4 SIMDs x 1 MFMA / 32 cycles at 2.4 GHz\n", iters);
    printf("%-10s %-6s %10s %10s %8s %12s\n", "workgroups", "waves", "us", "TFLOP/s", "of 2500", "shader MHz");
    for (int waves : {4, 8})
        for (int g : {64, 128, 192, 256, 512}) {
            std::vector<float> ts;
            std::vector<unsigned long long> h(2 * g);
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_kernel<4>, dim3(g), dim3(waves * 64), 0, 0, out, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
            }
            hipMemcpy(h.data(), clk, 2 * g * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::sort(ts.begin(), ts.end());
            const double us = ts[2] * 1e3;
            const double flop = (double)g * waves * iters * 16.0 * 2.0 * 32 * 32 * 16;
            double mhz = 0; for (int i = 0; i < g; ++i) mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; mhz /= g;
            printf("%-10d %-6d %10.1f %10.1f %8.3f %12.0f\n", g, waves, us, flop / us / 1e6, flop / us / 1e6 / 2500.0, mhz);
        }
    return 0;
}
