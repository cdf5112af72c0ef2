// exact-fp32 instantiations of the implicit-GEMM conv template (v_mfma_f32_32x32x2_f32 path: the parity gate).
#include "conv_igemm_kernel.h"
namespace v2v {
int launch_conv_f32(int cfg, const ConvKArgs& k, int ncls, hipStream_t s) { return launch_typed<float>(cfg, k, ncls, s); }
}
