// bf16 instantiations of the implicit-GEMM conv template (v_mfma_f32_32x32x16_bf16 path).
#include "conv3x3_pp_kernel.h"
#include "conv3x3_pp2_kernel.h"
#include "conv3x3_pp3_kernel.h"
#include "conv3x3_one_kernel.h"
#include "conv3x3_s2_kernel.h"
#include "conv3x3_t2_kernel.h"
#include "conv7x7_head_kernel.h"
namespace v2v {
int launch_conv_bf16(int cfg, const ConvKArgs& k, int ncls, hipStream_t s) { return launch_typed<bf16_t>(cfg, k, ncls, s); }
bool conv_cfg_has_helper(int cfg) { return cfg_has_helper_impl(cfg); }
int launch_patch_bf16(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_patch_typed<bf16_t>(cfg, k, s); }
int launch_pp_bf16(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_pp_typed<bf16_t>(cfg, k, s); }
int launch_pp2_bf16(int cfg, const ConvKArgs& k, int groups, hipStream_t s) { return launch_pp2_typed<bf16_t>(cfg, k, groups, s); }
int launch_pp3_bf16(int cfg, const ConvKArgs& k, int groups, hipStream_t s) { return launch_pp3_typed<bf16_t>(cfg, k, groups, s); }
int launch_one_bf16(int cfg, const ConvKArgs& k, int cus, hipStream_t s) { return launch_one_typed<bf16_t>(cfg, k, cus, s); }
int one_grid(int ntot, int cus) { return one_grid_size(ntot, cus); }
int launch_s2_bf16(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_s2_typed<bf16_t>(cfg, k, s); }
int launch_t2_bf16(int cfg, const ConvKArgs& k, hipStream_t s) { return launch_t2_typed<bf16_t>(cfg, k, s); }
int launch_head_bf16(const ConvKArgs& k, hipStream_t s) { return launch_head_typed<bf16_t>(k, s); }
int launch_c8_bf16(const ConvKArgs& k, hipStream_t s) { return launch_c8_typed<bf16_t>(k, s); }
int launch_rowsum_bf16(const ConvKArgs& k, hipStream_t s) { return launch_rowsum_bf16_impl(k, s); }
}
