"""Second, independent pin for FlowNet2's three CUDA-only ops (TEST INFRASTRUCTURE, never imported by the product).

`oracle/vid2vid_oracle.py` restates Correlation / Resample2d / ChannelNorm as vectorised torch expressions.  The reference
cannot be compiled here (no nvcc, legacy ATen API) and ships no vectors, so a shared misreading of the maths by one author
would pass every parity test (missing #7).  This file is a deliberately different formulation: a scalar-loop
transliteration of the `.cu` files' FLAT INDEX ARITHMETIC -- the padded NHWC scratch copies `rInput1/2` that
`channels_first` builds, the `indx1 / indx2 / tindx` expressions, the per-lane strided channel walk and the shuffle-down
reduction tree of the 32-thread block -- in numpy float32, one output element at a time.  Each function cites the lines it
follows.  tests/test_cpu_oracle.py cross-checks the two formulations on FlowNetC's geometry (pad 20, displacement 20,
stride2 2) and on flows that leave the image; the HIP kernels are then checked against both.

Parity status: still "unpinned" in the strict sense (no execution of the reference's CUDA code exists anywhere), but no
longer a single reading.
"""
import math

import numpy as np

F32 = np.float32
THREADS_PER_BLOCK = 32          # correlation_cuda_kernel.cu:10


def channels_first(inp, pad_size):
    """correlation_cuda_kernel.cu:46-70: NCHW -> zero-padded NHWC scratch, flat indices as written there."""
    n_, channels, height, width = inp.shape
    flat = np.ascontiguousarray(inp, dtype=F32).reshape(-1)
    dimcyx, dimyx = channels * height * width, height * width
    p_dimx, p_dimy = width + 2 * pad_size, height + 2 * pad_size
    p_dimyxc, p_dimxc = channels * p_dimy * p_dimx, p_dimx * channels
    rinput = np.zeros(n_ * p_dimyxc, dtype=F32)                     # rInput.fill_(0), correlation_cuda.cc:40-41
    for n in range(n_):                                             # grid (N, H, W), block 32 striding the channels
        for y in range(height):
            for x in range(width):
                for ch_off in range(THREADS_PER_BLOCK):
                    for c in range(ch_off, channels, THREADS_PER_BLOCK):
                        rinput[n * p_dimyxc + (y + pad_size) * p_dimxc + (x + pad_size) * channels + c] = \
                            flat[n * dimcyx + c * dimyx + y * width + x]
    return rinput


def _warp_reduce_sum(vals):
    """correlation_cuda_kernel.cu:17-21: val += __shfl_down_sync(full, val, offset) for offset = 16, 8, 4, 2, 1; lanes
    whose source lies beyond the warp keep their own value (shfl_down semantics).  Returns lane 0's result."""
    v = np.array(vals, dtype=F32)
    offset = 16
    while offset > 0:
        src = np.arange(32) + offset
        got = np.where(src < 32, v[np.minimum(src, 31)], v)
        v = (v + got).astype(F32)
        offset //= 2
    return v[0]


def correlation_forward(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda.cc:25-38 (output size) + correlation_cuda_kernel.cu:73-147 (one block per output pixel, thread c
    walks channels c, c+32, ...; fp32 accumulate; warp reduction; thread 0 writes acc0 / nelems)."""
    batch, n_in_ch, in_h, in_w = in1.shape
    kernel_radius = (kernel_size - 1) // 2
    border_radius = kernel_radius + max_displacement
    p_h, p_w = in_h + 2 * pad_size, in_w + 2 * pad_size
    d = (max_displacement // stride2) * 2 + 1
    n_out_ch = d * d
    out_h = int(math.ceil(float(p_h - 2 * border_radius) / float(stride1)))
    out_w = int(math.ceil(float(p_w - 2 * border_radius) / float(stride1)))
    r1, r2 = channels_first(in1, pad_size), channels_first(in2, pad_size)
    output = np.zeros(batch * n_out_ch * out_h * out_w, dtype=F32)
    kernel_rad = (kernel_size - 1) // 2
    displacement_rad = max_displacement // stride2
    displacement_size = 2 * displacement_rad + 1
    pdimyxc, pdimxc, pdimc = p_h * p_w * n_in_ch, p_w * n_in_ch, n_in_ch
    tdimcyx, tdimyx, tdimx = n_out_ch * out_h * out_w, out_h * out_w, out_w
    nelems = kernel_size * kernel_size * pdimc
    for n in range(batch):
        for by in range(out_h):
            for bz in range(out_w):
                y1 = by * stride1 + max_displacement
                x1 = bz * stride1 + max_displacement
                for tj in range(-displacement_rad, displacement_rad + 1):
                    for ti in range(-displacement_rad, displacement_rad + 1):
                        x2, y2 = x1 + ti * stride2, y1 + tj * stride2
                        lanes = np.zeros(32, dtype=F32)
                        for j in range(-kernel_rad, kernel_rad + 1):
                            for i in range(-kernel_rad, kernel_rad + 1):
                                indx1 = n * pdimyxc + (y1 + j) * pdimxc + (x1 + i) * pdimc
                                indx2 = n * pdimyxc + (y2 + j) * pdimxc + (x2 + i) * pdimc
                                for c in range(32):                  # thread c: ch = c, c + 32, ...
                                    for ch in range(c, pdimc, 32):
                                        lanes[c] = F32(lanes[c] + F32(r1[indx1 + ch] * r2[indx2 + ch]))
                        acc0 = _warp_reduce_sum(lanes)
                        tc = (tj + displacement_rad) * displacement_size + (ti + displacement_rad)
                        output[n * tdimcyx + tc * tdimyx + by * tdimx + bz] = F32(acc0 / F32(nelems))
    return output.reshape(batch, n_out_ch, out_h, out_w)


def resample2d_forward(img, flow, kernel_size=1):
    """resample2d_kernel.cu:15-64, one output element per "thread": flat index -> (b, c, y, x), flow read at (b, {0,1}, y, x),
    alpha / beta from the UNCLAMPED position, four clamped taps, fp32 accumulation in the order written."""
    b_, c_, _, _ = img.shape
    _, _, dim_h, dim_w = flow.shape
    img = np.ascontiguousarray(img, dtype=F32)
    flow = np.ascontiguousarray(flow, dtype=F32)
    out = np.zeros((b_, c_, dim_h, dim_w), dtype=F32)
    dim_chw, dim_hw = c_ * dim_h * dim_w, dim_h * dim_w
    for index in range(b_ * dim_chw):
        b = (index // dim_chw) % b_
        c = (index // dim_hw) % c_
        y = (index // dim_w) % dim_h
        x = index % dim_w
        dx, dy = flow[b, 0, y, x], flow[b, 1, y, x]
        xf, yf = F32(F32(x) + dx), F32(F32(y) + dy)
        alpha, beta = F32(xf - np.floor(xf)), F32(yf - np.floor(yf))
        xL = max(min(int(np.floor(xf)), dim_w - 1), 0)
        xR = max(min(int(np.floor(xf) + 1), dim_w - 1), 0)
        yT = max(min(int(np.floor(yf)), dim_h - 1), 0)
        yB = max(min(int(np.floor(yf) + 1), dim_h - 1), 0)
        val = F32(0.0)
        for fy in range(kernel_size):
            for fx in range(kernel_size):
                # (1. - alpha) is double arithmetic in the .cu (literal 1.), the product is cast to float before the +=
                val = F32(val + F32((1.0 - float(alpha)) * (1.0 - float(beta)) * float(img[b, c, yT + fy, xL + fx])))
                val = F32(val + F32(float(alpha) * (1.0 - float(beta)) * float(img[b, c, yT + fy, xR + fx])))
                val = F32(val + F32((1.0 - float(alpha)) * float(beta) * float(img[b, c, yB + fy, xL + fx])))
                val = F32(val + F32(float(alpha) * float(beta) * float(img[b, c, yB + fy, xR + fx])))
        out[b, c, y, x] = val
    return out


def channelnorm_forward(x):
    """channelnorm_kernel.cu:18-60: result += float(val * val) over the channels in order, sqrt, one thread per pixel."""
    b_, c_, h_, w_ = x.shape
    flat = np.ascontiguousarray(x, dtype=F32).reshape(-1)
    out = np.zeros((b_, 1, h_, w_), dtype=F32)
    chw, hw = c_ * h_ * w_, h_ * w_
    for index in range(b_ * hw):
        b = (index // hw) % b_
        y = (index // w_) % h_
        xx = index % w_
        result = F32(0.0)
        for c in range(c_):
            val = flat[b * chw + c * hw + y * w_ + xx]
            result = F32(result + F32(val * val))
        out[b, 0, y, xx] = F32(np.sqrt(result))
    return out
