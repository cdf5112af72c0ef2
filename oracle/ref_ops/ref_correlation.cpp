// Reference correlation forward executed on the host: kernel bodies = correlation_cuda_kernel.cu:16-147 (warpReduceSum,
// blockReduceSum, channels_first, correlation_forward), extracted by build.sh into _ref/gen_correlation.inc.
// The launch geometry below restates correlation_forward_cuda_kernel (correlation_cuda_kernel.cu:372-410) and the output
// size / padded-buffer rule of correlation_cuda.cc:25-42.
#include "cuda_emu.h"
#include "gen_correlation.inc"

extern "C" int ref_correlation_out_size(int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                        int* oc, int* oh, int* ow) {
    const int kernel_radius = (kernel_size - 1) / 2, border_radius = kernel_radius + max_displacement;   // correlation_cuda.cc:23-24
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;                                               // :26-27
    *oc = ((max_displacement / stride2) * 2 + 1) * ((max_displacement / stride2) * 2 + 1);                // :29
    *oh = (int)std::ceil((float)(pH - 2 * border_radius) / (float)stride1);                               // :31
    *ow = (int)std::ceil((float)(pW - 2 * border_radius) / (float)stride1);                               // :32
    return 0;
}

// in1, in2: [N][C][H][W] fp32 contiguous; out: [N][oc][oh][ow] (ref_correlation_out_size)
extern "C" int ref_correlation_forward(const float* in1, const float* in2, float* out, int N, int C, int H, int W,
                                       int pad_size, int kernel_size, int max_displacement, int stride1, int stride2) {
    int oc, oh, ow;
    ref_correlation_out_size(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    // GUARD BAND.  The reference kernel offsets its window centre by max_displacement, not by the border radius
    // (correlation_cuda_kernel.cu:90-91: y1 = blockIdx.y * stride1 + max_displacement), so for every kernel_size > 1 the first
    // output row / column reads rows and columns -kernel_rad .. -1 of rInput2 (:111-123, y2 + j = blockIdx.y * stride1 - kernel_rad
    // at tj = -displacement_rad): for n = 0 that is memory IN FRONT of the padded buffer, whatever pad_size is.  On a GPU the
    // result then depends on what the allocator left there; here it would depend on the heap (found by the round-4 judge with
    // MALLOC_PERTURB_).  The checker therefore allocates kernel_rad rows + kernel_rad pixels of ZEROS in front of and behind both
    // padded buffers: the out-of-bounds operand is a defined 0 (what a zero-filled neighbouring allocation would give), and
    // results are deterministic.  Reads that stay inside the buffer but wrap to the previous row / image are the reference's
    // own arithmetic and are left alone.
    const int kernel_rad = (kernel_size - 1) / 2;
    const size_t guard = (size_t)kernel_rad * pW * C + (size_t)kernel_rad * C + 64;
    const size_t body = (size_t)N * pH * pW * C;
    std::vector<float> r1(body + 2 * guard, 0.f), r2(body + 2 * guard, 0.f);                   // rInput.resize_ + fill_(0), correlation_cuda.cc:34-39
    std::fill(out, out + (size_t)N * oc * oh * ow, 0.f);                                       // output.fill_(0), :40
    float* p1 = r1.data() + guard; float* p2 = r2.data() + guard;
    // channels_first<<<(N, H, W), THREADS_PER_BLOCK>>>  (correlation_cuda_kernel.cu:382-397): no synchronisation inside
    launch_flat(dim3(N, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<float>(in1, p1, C, H, W, pad_size); });
    launch_flat(dim3(N, H, W), dim3(THREADS_PER_BLOCK), [&] { channels_first<float>(in2, p2, C, H, W, pad_size); });
    // correlation_forward<<<(N, oh, ow), THREADS_PER_BLOCK>>>  (:399-413): 32 threads in lock step (warp shuffles)
    launch_lockstep(dim3(N, oh, ow), dim3(THREADS_PER_BLOCK),
                    [&] { correlation_forward<float>(out, oc, oh, ow, p1, C, H, W, p2, pad_size, kernel_size, max_displacement, stride1, stride2); });
    return 1;                                                                                  // success code of the reference (:426)
}
