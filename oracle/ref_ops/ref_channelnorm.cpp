// Reference ChannelNorm forward / backward executed on the host: kernel bodies = channelnorm_kernel.cu:18-96 with the file's
// own macros (:7-14), extracted by build.sh into _ref/gen_channelnorm.inc.  Launch geometry: channelnorm_kernel_forward /
// _backward (:98-177): (n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS blocks of CUDA_NUM_THREADS threads.
#include "cuda_emu.h"
#include "gen_channelnorm.inc"

static long4 sz(int a, int b, int c, int d) { return make_long4(a, b, c, d); }
static long4 st(int a, int b, int c, int d) { (void)a; return make_long4((long)b * c * d, (long)c * d, d, 1); }

// x [N][C][H][W] -> out [N][1][H][W]
extern "C" int ref_channelnorm_forward(const float* x, float* out, int N, int C, int H, int W, int norm_deg) {
    const int n = N * H * W;                                                  // output.numel()  (:109)
    launch_flat(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_channelnorm_update_output<float>(n, x, sz(N, C, H, W), st(N, C, H, W), out, sz(N, 1, H, W), st(N, 1, H, W), norm_deg); });
    return 0;
}

extern "C" int ref_channelnorm_backward(const float* x, const float* out, const float* grad_out, float* grad_in, int N, int C, int H, int W, int norm_deg) {
    const int n = N * C * H * W;                                              // gradInput1.numel()  (:152)
    launch_flat(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_channelnorm_backward_input1<float>(n, x, sz(N, C, H, W), st(N, C, H, W), out, sz(N, 1, H, W), st(N, 1, H, W),
                                                  grad_out, sz(N, 1, H, W), st(N, 1, H, W), grad_in, sz(N, C, H, W), st(N, C, H, W), norm_deg); });
    return 0;
}
