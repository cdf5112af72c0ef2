// exact-fp32 instantiations of the implicit-GEMM conv template (v_mfma_f32_32x32x2_f32 path: the parity gate).
#include "conv3x3_pp_kernel.h"
#include "conv3x3_pp2_kernel.h"
#include "conv3x3_pp3_kernel.h"
#include "conv3x3_s2_kernel.h"
#include "conv3x3_t2_kernel.h"
#include "conv7x7_head_kernel.h"
namespace v2v {
int launch_conv_f32(int cfg, const ConvKArgs& k, int ncls, hipStream_t s) { return launch_typed<float>(cfg, k, ncls, s); }
int launch_patch_f32(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_patch_typed<float>(cfg, k, s); }
int launch_pp_f32(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_pp_typed<float>(cfg, k, s); }
int launch_pp2_f32(int cfg, const ConvKArgs& k, int groups, hipStream_t s) { return launch_pp2_typed<float>(cfg, k, groups, s); }
int launch_pp3_f32(int cfg, const ConvKArgs& k, int groups, hipStream_t s) { return launch_pp3_typed<float>(cfg, k, groups, s); }
int launch_s2_f32(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_s2_typed<float>(cfg, k, s); }
int launch_t2_f32(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_t2_typed<float>(cfg, k, s); }
int launch_head_f32(const ConvKArgs& k, hipStream_t s) { return launch_head_typed<float>(k, s); }
int launch_c8_f32(const ConvKArgs& k, hipStream_t s) { return launch_c8_typed<float>(k, s); }
}
