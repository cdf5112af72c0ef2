// Internal helpers shared by the HIP translation units of libv2v_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <memory>
#include "../../include/v2v_hip.h"

namespace v2v {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8)))  __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4)))  float  f32x4;
typedef __attribute__((ext_vector_type(16))) float  f32x16;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
// Round 4: the conversion is ONE gfx950 instruction (v_cvt_pk_bf16_f32: two fp32 -> packed bf16, round-to-nearest-even) instead of
// the ~7 integer operations of the software rounding it replaces -- every bf16 epilogue (conv outputs, the fused norm, bn_apply, the
// pointwise kernels) converts 8 values per 16-byte store.  Same results for every finite input and infinities (checked bit for bit
// against torch's conversion on the device: tests/test_gpu_kernels.py::test_bf16_hardware_conversion_is_rne); a NaN stays a NaN.
typedef __attribute__((ext_vector_type(2))) float v2v_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 v2v_bf16x2;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const v2v_f32x2 v = {lo, hi};
    const v2v_bf16x2 r = __builtin_convertvector(v, v2v_bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int VEC = 4;    // elements per 16-byte vector
    static constexpr int BKE = 32;   // elements per 128-byte K chunk
    static constexpr int DTYPE = V2V_F32;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int BKE = 64;
    static constexpr int DTYPE = V2V_BF16;
};

__device__ __forceinline__ float load_act(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float load_act(const bf16_t* p, int64_t i) {
    return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p)[i]);
}
__device__ __forceinline__ void store_act(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void store_act(bf16_t* p, int64_t i, float v) {
    reinterpret_cast<unsigned short*>(p)[i] = f32_to_bf16_bits(v);
}

__device__ __forceinline__ float apply_act(float v, int act, float param) {
    switch (act) {
        case V2V_ACT_RELU:    return v > 0.f ? v : 0.f;
        case V2V_ACT_LEAKY:   return v > 0.f ? v : v * param;
        case V2V_ACT_TANH:    return tanhf(v);
        case V2V_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default:              return v;
    }
}

// In-kernel norm finalize, shared by the conv and gather-sum stem kernels: thread (channel, phase ph of PH) adds the (sum, sum^2)
// rows ph, ph + PH, ph + 2 PH, ... (< total) of its channel IN THAT ORDER, in double.  The rows were published by other workgroups
// with 8-byte agent-scope (write-through) stores and are read with agent-scope loads, U of them in flight at a time: one load per
// loop trip (the first version) serialised 32-64 memory round trips in the last workgroup of a 128-512 row layer -- 3-19 us of
// tail per launch (profiles/r02_a68_s2_fin_ab.txt).  The summation order, hence every bit of the result, is unchanged.
template <int U>
__device__ __forceinline__ void sum_stat_rows(const float* stats, long long col2, long long row_stride, int ph, int PH, int total,
                                              double& s1, double& s2) {
    const unsigned long long* const base = reinterpret_cast<const unsigned long long*>(stats + col2);   // 8-byte granules
    const long long rs = row_stride / 2;
    int r = ph;
    for (; r + (U - 1) * PH < total; r += U * PH) {
        unsigned long long b[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            b[u] = __hip_atomic_load(base + (long long)(r + u * PH) * rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s1 += (double)__uint_as_float((unsigned)(b[u] & 0xffffffffull));
            s2 += (double)__uint_as_float((unsigned)(b[u] >> 32));
        }
    }
    for (; r < total; r += PH) {
        const unsigned long long b = __hip_atomic_load(base + (long long)r * rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 += (double)__uint_as_float((unsigned)(b & 0xffffffffull));
        s2 += (double)__uint_as_float((unsigned)(b >> 32));
    }
}

// ---- host side: op recording ---------------------------------------------------------
struct Op {
    virtual ~Op() {}
    virtual int launch(hipStream_t s) = 0;
    virtual const char* name() const = 0;
    std::string label;
    int lane = 0;          // plan lane (hipGraph branch) the op was recorded on; see plan.hip
};

// Either launches `op` now on `stream` or, when a plan is recording on this thread,
// appends it to the plan (plan.cpp).
int submit(std::unique_ptr<Op> op, void* stream);
void set_error(const char* fmt, ...);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("HIP launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace v2v
