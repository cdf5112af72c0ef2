// Reference Resample2d forward / backward executed on the host: kernel bodies = resample2d_kernel.cu:15-190
// (kernel_resample2d_update_output, _backward_input1, _backward_input2) with the file's own macros (:5-13), extracted by
// build.sh into _ref/gen_resample2d.inc.  Launch geometry: resample2d_kernel_forward / _backward (:192-310):
// (n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS blocks of CUDA_NUM_THREADS threads, n = numel of the written tensor.
#include "cuda_emu.h"
#include "gen_resample2d.inc"

static long4 sz(int a, int b, int c, int d) { return make_long4(a, b, c, d); }
static long4 st(int a, int b, int c, int d) { (void)a; return make_long4((long)b * c * d, (long)c * d, d, 1); }   // contiguous NCHW

// img [N][C][H][W], flow [N][2][OH][OW] -> out [N][C][OH][OW]
extern "C" int ref_resample2d_forward(const float* img, const float* flow, float* out, int N, int C, int H, int W, int OH, int OW, int kernel_size) {
    const int n = N * C * OH * OW;
    launch_flat(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_update_output<float>(n, img, sz(N, C, H, W), st(N, C, H, W), flow, sz(N, 2, OH, OW), st(N, 2, OH, OW),
                                               out, sz(N, C, OH, OW), st(N, C, OH, OW), kernel_size); });
    return 0;
}

// gradients w.r.t. img (zero-filled, then atomically accumulated: resample2d.py:31-32 + kernel) and w.r.t. flow
extern "C" int ref_resample2d_backward(const float* img, const float* flow, const float* grad_out, float* grad_img, float* grad_flow,
                                       int N, int C, int H, int W, int OH, int OW, int kernel_size) {
    std::fill(grad_img, grad_img + (size_t)N * C * H * W, 0.f);
    std::fill(grad_flow, grad_flow + (size_t)N * 2 * OH * OW, 0.f);
    int n = N * C * OH * OW;                                                  // gradOutput.numel()  (:251)
    launch_flat(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_backward_input1<float>(n, img, sz(N, C, H, W), st(N, C, H, W), flow, sz(N, 2, OH, OW), st(N, 2, OH, OW),
                                                 grad_out, sz(N, C, OH, OW), st(N, C, OH, OW), grad_img, sz(N, C, H, W), st(N, C, H, W), kernel_size); });
    n = N * 2 * OH * OW;                                                      // gradInput2.numel()  (:282)
    launch_flat(dim3((n + CUDA_NUM_THREADS - 1) / CUDA_NUM_THREADS), dim3(CUDA_NUM_THREADS), [&] {
        kernel_resample2d_backward_input2<float>(n, img, sz(N, C, H, W), st(N, C, H, W), flow, sz(N, 2, OH, OW), st(N, 2, OH, OW),
                                                 grad_out, sz(N, C, OH, OW), st(N, C, OH, OW), grad_flow, sz(N, 2, OH, OW), st(N, 2, OH, OW), kernel_size); });
    return 0;
}
