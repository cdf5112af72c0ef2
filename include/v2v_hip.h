/*
 * v2v_hip.h -- C ABI of libv2v_hip.so, the MI355X (gfx950) native backend of the
 * vid2vid hot path.
 *
 * Conventions (all entry points):
 *   - plain C: raw DEVICE pointers, ints, floats and a hipStream_t passed as void*.
 *     No torch / ATen types cross this boundary.
 *   - the callee never allocates, never synchronises and keeps no global state; every
 *     launch goes to the stream the caller passes (the reference launches on
 *     at::cuda::getCurrentCUDAStream(): correlation_cuda.cc:62, resample2d_kernel.cu:212).
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch or
 *     V2V_EINVAL (-1) for an argument the kernel family does not support.  The reference
 *     returns 1/0 and raises through AT_ERROR (correlation_cuda.cc:81-83); the Python
 *     binding (vid2vid_amd/lib.py) turns a non-zero code into RuntimeError the same way.
 *   - activations are NHWC ("channels last") with a channel stride that is a multiple of
 *     16 bytes; dtype code V2V_F32 = exact fp32 MFMA path (parity), V2V_BF16 = bf16
 *     storage + fp32 accumulate (throughput).
 *
 * What each group replaces in the reference (paths relative to the reference root):
 *   v2v_conv2d / v2v_bn_* / v2v_pool / v2v_warp_blend / v2v_encode_labels
 *       the ATen/cuDNN ops reached through torch.nn in models/networks.py:117-325
 *       (CompositeGenerator / CompositeLocalGenerator), :327-419 (GlobalGenerator /
 *       LocalEnhancer), :554-593 (ResnetBlock), :634-725 (discriminators) and
 *       models/vid2vid_model_G.py:86-112 (encode_input), models/base_model.py:122-152.
 *   v2v_correlation_forward   correlation_cuda.forward  (correlation_cuda.cc:10-87)
 *   v2v_resample2d_forward    resample2d_cuda.forward   (resample2d_cuda.cc:6-13)
 *   v2v_channelnorm_forward   channelnorm_cuda.forward  (channelnorm_cuda.cc:6-14)
 */
#ifndef V2V_HIP_H
#define V2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2V_EINVAL (-1)

/* dtype codes */
#define V2V_F32  0
#define V2V_BF16 1

/* padding modes */
#define V2V_PAD_ZERO    0
#define V2V_PAD_REFLECT 1

/* epilogue activations: y = act(x + bias) * out_scale */
#define V2V_ACT_NONE    0
#define V2V_ACT_RELU    1
#define V2V_ACT_LEAKY   2   /* slope in act_param */
#define V2V_ACT_TANH    3
#define V2V_ACT_SIGMOID 4

/* conv output modes */
#define V2V_OUT_RAW_F32_NHWC 0  /* fp32 NHWC, pre-norm; optional per-tile statistics */
#define V2V_OUT_ACT_NHWC     1  /* activation dtype NHWC, after bias/act/scale        */
#define V2V_OUT_F32_NCHW     2  /* fp32 planar NCHW (API-facing heads)                */
#define V2V_OUT_NORM_ACT_NHWC 3 /* activation dtype NHWC after the training-mode norm (batch statistics of THIS launch),
                                 * activation and residual adds -- see "fused norm" below                            */
#define V2V_OUT_RAW_ACT_NHWC 4  /* pre-norm like mode 0, but STORED in the activation dtype (bf16 storage: half the bytes of the
                                 * raw tensor the following v2v_bn_apply_raw reads back); the per-tile statistics and the in-kernel
                                 * finalize still come from the fp32 accumulators, bit for bit those of mode 0.  Generic, patch
                                 * and single-phase tiles, and (round 6) the persistent single-chunk tiles 140 / 141 / 143 / 114,
                                 * for which it is the inference path's default; not tiles 60 / 61.  With V2V_F32 storage identical
                                 * to mode 0.                                                                                      */

/* Descriptor of one convolution / transposed convolution launch.  POD, passed by pointer,
 * copied by the callee before it returns. */
typedef struct v2v_conv_desc {
    const void*  in;        /* [N][H][W][cin_stride]  activation dtype                       */
    const void*  w;         /* packed weights, see v2v_conv_packed_elems()                   */
    const float* bias;      /* [cout] fp32 or NULL                                           */
    void*        out;       /* see out_mode                                                  */
    float*       stats;     /* NULL or [n_classes*m_tiles][cout][2] fp32 (sum, sum of sq.)   */
    const void*  zero_page; /* >= 16 zero bytes, 16-B aligned: source of padded / ragged lanes */
    int32_t N, H, W;        /* input batch / height / width                                  */
    int32_t cin;            /* real input channels (weights beyond are zero)                 */
    int32_t cin_stride;     /* channel stride of `in`, elements; multiple of 16 B            */
    int32_t cout;           /* real output channels                                          */
    int32_t cout_stride;    /* channel stride of `out` for NHWC output modes                 */
    int32_t KH, KW;         /* kernel size                                                   */
    int32_t stride;         /* 1 or 2                                                        */
    int32_t pad;            /* symmetric padding                                             */
    int32_t pad_mode;       /* V2V_PAD_*  (transposed: zero only)                            */
    int32_t transposed;     /* 0 = Conv2d, 1 = ConvTranspose2d (or Conv2d backward-data), see below */
    int32_t OH, OW;         /* output height / width                                         */
    int32_t dtype;          /* V2V_F32 / V2V_BF16 : dtype of in, w (and out in ACT mode)     */
    int32_t out_mode;       /* V2V_OUT_*                                                     */
    int32_t act;            /* V2V_ACT_* (ignored for RAW output)                            */
    float   act_param;      /* leaky slope                                                   */
    float   out_scale;      /* multiplies the activated value (flow heads: 20 * 2^scale)     */
    int32_t tile;           /* 0 = auto, else force tile config id (testing/tuning)          */
    int32_t* fin_counter;   /* NULL, or >= 64 zero-initialised tickets: finalize the training-mode norm in-kernel */
    const float* fin_gamma; /* [cout] or NULL (affine = False)                               */
    const float* fin_beta;  /* [cout] or NULL                                                */
    float*  fin_scale_shift;  /* [4][cout] fp32 out: scale, shift, mean, invstd (as v2v_bn_finalize) */
    float*  fin_running_mean; /* [cout] or NULL: updated with fin_momentum                     */
    float*  fin_running_var;  /* [cout] or NULL (unbiased variance)                            */
    float   fin_eps;
    float   fin_momentum;
    int64_t fin_count;      /* N*OH*OW                                                       */
    int32_t splitk;         /* 0/1 = off; S > 1: S workgroups share a tile along K (see below)        */
    int32_t prefetch;       /* 0 = off; P > 0: weight-prefetch helper wave, P K-chunks ahead (see below) */
    void*   slabs;          /* splitk > 1: v2v_conv_splitk_workspace() bytes of scratch              */
    int32_t* sk_counter;    /* splitk > 1: `tickets` ints, zero before the first launch (re-armed in-kernel) */
    int32_t w_korder;       /* K order `w` was packed in (v2v_conv_pack_weights): 0 tap-major, 1 channel-chunk-major, 2 full-tap chunk-major (transposed), 3 paired-x (below); a korder-4 packing is read as 1, a korder-5 packing as 0 */
    int32_t ablate;         /* profiling only, results are WRONG when non-zero: 1 = activation tiles from the zero page, 2 = weight tiles from one hot line, 4 = no output stores, 16 = loaders only (no LDS reads / MFMA), 512 = return at once (launch floor), 1024 = no main loop (prologue + epilogue), 2048 = one workgroup per channel tile stays away from the fused-norm barrier (exercises its give-up path: NaN outputs + v2v_device_status bit 0) */
    const void* res0;       /* V2V_OUT_NORM_ACT_NHWC: NULL or a residual [N][OH][OW][cout_stride] (activation dtype) added after the activation */
    const void* res1;       /* second residual, NULL or as res0                                                   */
    int32_t act_split;      /* 0, or first output channel of a SECOND head merged into this launch (see "merged heads")  */
    int32_t act_b;          /* activation of channels >= act_split                                                */
    float   act_param_b;
    float   out_scale_b;
    double* fin_workspace;  /* round 4, with fin_counter: NULL, or v2v_bn_finalize_groups(rows) * cout * 2 doubles -> layers with more than 512 statistics rows finalize IN the launch in two levels (see "two-level finalize" below) */
} v2v_conv_desc;

/* Two-level finalize (fin_counter + fin_workspace, more than 512 statistics rows): the last workgroup of each row group reduces the
 * group (the arithmetic of bn_partial_reduce), the last group finalizes (that of bn_finalize on the group rows): bit for bit the
 * v2v_bn_finalize result, two launches fewer.  fin_counter must then hold 256 + groups * ceil(cout / 64) ints (zero before the first
 * launch, re-armed in-kernel).  fin_workspace == NULL: such layers are not finalized in-kernel (call v2v_bn_finalize).  Measured
 * slower than the two separate launches for thousand-tile layers: the Python engine leaves it off (V2V_FIN2=1 enables). */
/* Merged heads (tile 60, V2V_OUT_F32_NCHW, act_split > 0): model_final_flow (2 channels, no activation, x 20) and
 * model_final_w (1 channel, sigmoid) read the same tensor (models/networks.py:181-183, 224-226); with their weights
 * concatenated along cout they are one launch whose channels >= act_split use act_b / act_param_b / out_scale_b. */

/* first of the 128 per-channel-tile launch-tag words of a fused-norm launch inside fin_counter: behind the arrive / depart tickets [0, 256),
 * the two-level finalize's row-group tickets [256, 256 + 64 * 128) and the one-hot stems' slice tickets (256 words) */
#define V2V_FIN_TAG_WORD (256 + 64 * 128 + 256)

/* Fused norm (out_mode V2V_OUT_NORM_ACT_NHWC): conv + BatchNorm2d / InstanceNorm2d in training mode + activation
 * (+ residuals) in ONE launch -- what [pad, conv, norm, relu] and the tail of a ResnetBlock are in the reference
 * (models/networks.py:571-593).  The statistics need every output pixel, so the workgroups that share an output-channel
 * tile meet between the main loop and the store: each publishes its tile's (sum, sum^2) row, polls for the others',
 * derives scale / shift from all rows in a fixed order (same arithmetic as the in-kernel finalize), and
 * applies it to the accumulators it still holds in registers.  No fp32 raw tensor, no bn_apply launch.
 * Round 6: the rows are self-validating 8-byte granules {fp32 value, launch tag} (two per channel), so `stats` of a fused
 * launch is rows * cout * 4 floats, ZERO-INITIALISED by the caller and written by fused launches ONLY (a buffer it shares with the
 * untagged rows of other launches could hold a matching tag by accident); the launch tag is 1 + the word
 * fin_counter[V2V_FIN_TAG_WORD + channel tile], which the last workgroup to leave increments.
 * Requirements, checked by the library: tile ids 80..93, splitk <= 1, cout == cout_stride, `stats`, `fin_counter`
 * (>= V2V_FIN_TAG_WORD + 128 zero ints; the library re-arms / advances them) and `fin_scale_shift` given, and ALL workgroups of the launch co-resident:
 * m_tiles * n_tiles (* 2 for v2v_conv2d_pair) <= the number of compute units.  Two such launches must never run
 * concurrently on one device (each could hold half of the CUs and wait for the rest): issue them from one stream / one
 * plan lane only, and keep the device to ONE process while they run (a co-tenant holding compute units can keep part of the
 * workgroups off the chip).  A barrier that does not complete within ~1 s gives up: the outputs of that launch are NaN AND bit 0
 * of the library's host-visible status word is set -- poll it with v2v_device_status() (no synchronisation needed). */

/* splitk = S > 1: the S workgroups of a tile each reduce 1/S of the K chunks and publish an fp32 partial tile; the
 * last to arrive sums the S partial tiles in slice order (deterministic, independent of arrival order) and runs the
 * normal epilogue (bias, statistics, norm finalize, activation).  For small-M layers that cannot fill 256 CUs with
 * LDS-read-efficient tiles.
 * prefetch = P > 0: tile configurations with a helper instance launch one extra wave per workgroup that touches the
 * weight lines P K-chunks (128 B per weight row each) ahead of the LDS-DMA loaders, so the batch-1 weight stream
 * (every line a cold HBM miss) is an L2 hit when the loaders fetch it.  Bitwise neutral. */

/* transposed = 1: out[s*i - pad + k] += in[i]*w[k] for any OH <= (H-1)*s - 2*pad + KH + (s-1)
 * (output_padding < s).  The same operator is the backward-data pass of a Conv2d (same weight tensor,
 * roles of cin / cout swapped): see v2v_conv2d. */

/* Packed weight layout.  One matrix per class (Conv2d: 1 class; ConvTranspose2d stride 1: 1 class,
 * stride 2: 4 output-parity classes (a,b)), each [round_up(cout,128)][kpad] with
 * k = tap * cin_stride + c (taps row-major over the class's (kh,kw) list) and kpad = k
 * rounded up to 128 bytes; zero filled.  v2v_conv_packed_elems gives the total element count;
 * v2v_conv_pack_weights is the device kernel that fills it from PyTorch-layout fp32 weights
 * [cout][cin][KH][KW] (Conv2d) or [cin][cout][KH][KW] (ConvTranspose2d), converting to `dtype`. */
int64_t v2v_conv_packed_elems(int32_t cin, int32_t cin_stride, int32_t cout, int32_t KH, int32_t KW,
                              int32_t transposed, int32_t stride, int32_t pad, int32_t dtype);
int     v2v_conv_pack_weights(const float* w, void* dst, int32_t cin, int32_t cin_stride, int32_t cout,
                              int32_t KH, int32_t KW, int32_t transposed, int32_t stride, int32_t pad,
                              int32_t dtype, int32_t korder, void* stream);
/* korder + 256: the fp32 source tensor is channels-last -- [cout][KH][KW][cin] for a Conv2d weight, [cin][KH][KW][cout] for a
 * ConvTranspose2d weight (the fused optimizer's flat buffers hold 4-D weights that way; same packed result).
 * korder 0 (tap-major, above) is what the implicit-GEMM tile configurations (ids 1..23) read.  korder 1 is
 * k = (chunk * KH*KW + tap) * E + c_in_chunk with E = elements per 128 bytes and channel c = chunk*E + c_in_chunk
 * (channel-chunk outer, tap inner; Conv2d with cin_stride % E == 0 only; same element count): the layout of the
 * LDS-resident-patch 3x3 kernel (tile ids 32..37, csrc/conv3x3_patch_kernel.h), which fetches the input patch of
 * a channel chunk once and walks the 9 taps over it.
 * v2v_conv_desc.w_korder 3 (no packing mode of its own): the PAIRED-X view of a 3x3 / stride 1 / pad 1 Conv2d with <= 32 input and
 * exactly 32 output channels (models/networks.py:554-593 at ngf_s = 32: 64-byte bf16 pixels), read by tile ids 140..143 when
 * cin_stride == 32.  The NHWC tensors [H][W][32] ARE [H][W/2][64] (paired pixel X = pixels 2X, 2X+1); `w` is the korder-1 packing of
 * the 64 -> 64 weight W'[a*32+co][b*32+ci][ky][kX] = W[co][ci][ky][2kX+b-a-1] (zero outside 0..2) that the caller assembled; the
 * descriptor keeps the layer's own geometry (cin <= 32, cin_stride = cout = cout_stride = 32, W even, V2V_OUT_RAW_F32_NHWC), `bias`
 * its 32 values, the statistics rows / finalize record its 32 channels.  Horizontal reflection becomes a clamp inside the kernel.
 * Transposed counterpart (tile id 114 when cin_stride == 32): ConvTranspose2d(3x3, s2, p1, op1) with <= 32 input and exactly 16 output
 * channels (models/networks.py:254-260 at ngf_s = 16) as the 64 -> 32 transposed layer over [H][W/2][64] -> [2H][W][32] whose weight
 * W3[b*32+ci][e*16+co][ky][kx'] = W[ci][co][ky][kx] (kx = e+1-2b, 3+e-2b, e-1-2b for kx' = 1, 2, 0; zero outside 0..2) the caller packed
 * with korder 2; descriptor: the layer's own geometry (cin_stride 32, cout = cout_stride = 16, OW = 2 W), 16 bias values, 16 statistics columns.
 * korder 4 (round 6): the BACKWARD-DATA operator of a 3x3 / stride 1 Conv2d as a convolution -- `w` is the layer's own [cout][cin][3][3]
 * parameter, passed with transposed = 1 (role-swapped read: `cin` = the layer's cout = channels of dY, `cout` = the layer's cin = channels
 * of dX), packed channel-chunk-major like korder 1 with the taps FLIPPED (matrix tap t = kernel tap 8 - t).  Run it with
 * v2v_conv_desc { transposed 0, KH = KW = 3, stride 1, pad = 2 - p, V2V_PAD_ZERO, w_korder 1 } on tile ids 80..93: p = 1 gives dX on the layer's
 * grid, p = 0 (the layer sat behind a ReflectionPad2d(1)) the gradient of the PADDED input, (H+2) x (W+2), which v2v_reflect_pad_fold folds
 * (the autograd of models/networks.py:571-587).  Tile ids 80..93 accept pad 1 or 2; every other 3x3 patch tile pad 1 only.
 * korder 5 (round 6): the same operator for a square, odd, stride-1 Conv2d of any size, packed TAP-major like korder 0 with the taps flipped
 * -- what tile id 61 (7x7 over 16-byte pixels) reads for the backward-data of the 7x7 heads (models/networks.py:178-183: dY has 3 | 2 | 1
 * channels = one 16-byte vector per pixel).  Run it with v2v_conv_desc { transposed 0, KH = KW = 7, stride 1, pad = 6 - p, V2V_PAD_ZERO,
 * w_korder 0, tile 61, out_mode V2V_OUT_ACT_NHWC }: tile 61 accepts zero padding of 3 ... 6 (OH = H + 2 pad - 6) and, besides planar fp32
 * and raw fp32 NHWC, plain activation-typed NHWC output (no activation, cout and cout_stride whole 16-byte vectors, no statistics). */

/* Number of statistics rows the launch described by `d` writes: n_classes * m_tiles -- except for the persistent tile ids 140..143 and
 * 114, which keep their sums in registers across the tiles (114: and the four parity classes) a workgroup walks and leave ONE row per
 * workgroup (min(tiles, compute units)); with fin_counter they also write the scale / shift record themselves (last workgroup), at any
 * layer size. */
int     v2v_conv_stats_rows(const v2v_conv_desc* d);
/* Bytes of `slabs` scratch (and number of `sk_counter` ints via *tickets) the launch needs; 0 when splitk <= 1. */
int64_t v2v_conv_splitk_workspace(const v2v_conv_desc* d, int32_t* tickets);
/* Largest m_tiles * n_tiles (* 2 for a pair) a V2V_OUT_NORM_ACT_NHWC launch may have on this device (= its CU count). */
int     v2v_conv_fused_norm_max_workgroups(void);
/* Constants of the exact division q = (umulhi(M, n) + n) >> l of 0 <= n < 2^31 by 1 <= d < 2^31 (conv epilogue: row -> pixel map of the transposed layers; no reference counterpart, host-side helper exported for its test). */
int     v2v_fastdiv_magic(uint32_t d, uint32_t* m_out, int32_t* l_out);
/* Tile configuration id the launch would use (after auto selection). */
int     v2v_conv_tile_config(const v2v_conv_desc* d);
/* Launch.  nn.Conv2d / nn.ConvTranspose2d forward (models/networks.py:132-183 etc.), and -- with the
 * SAME parameter tensor packed in the opposite role -- their backward-data (autograd of F.conv2d /
 * F.conv_transpose2d in the reference):
 *   dX of Conv2d(w[cout][cin], stride s, zero pad p)   = transposed conv of dY with `w` read as a
 *       ConvTranspose2d weight [cin'=cout][cout'=cin], stride s, pad p, OH = H of the forward input;
 *   dX of Conv2d behind ReflectionPad2d(p)             = the same with pad 0 and OH = H + 2p, followed by
 *       v2v_reflect_pad_fold;
 *   dX of ConvTranspose2d(w[cin][cout], stride 2, p)   = Conv2d of dY with `w` read as a Conv2d weight
 *       [cout'=cin][cin'=cout], stride 2, pad p. */
int     v2v_conv2d(const v2v_conv_desc* d, void* stream);
/* Grouped launch: two convolutions of IDENTICAL geometry, modes and tile configuration (tile ids 70..89: the
 * second-schedule ping-pong and the single-phase 3x3 kernels) as ONE launch -- block z picks its member's tensors.  The twin chains of
 * CompositeGenerator (label / image towers, image / flow branches: models/networks.py:203-232) each fill only half
 * of the 256 CUs at batch 1; paired they fill the chip without split-K.  Each member keeps its own output, statistics,
 * finalize tickets and split-K scratch; results are bitwise those of two v2v_conv2d calls. */
int     v2v_conv2d_pair(const v2v_conv_desc* a, const v2v_conv_desc* b, void* stream);

/* Weight gradient:  G[r][c][kh][kw] (+)= sum_{n,oi,oj} P[n][oi][oj][r] * Q[n][oi*s+kh-pad][oj*s+kw-pad][c]
 *   Conv2d:           P = dY (rows = cout), Q = X  (cols = cin)  -> dW[cout][cin][KH][KW]
 *   ConvTranspose2d:  P = X  (rows = cin),  Q = dY (cols = cout) -> dW[cin][cout][KH][KW], s = 2
 * P, Q are NHWC activations of `dtype`; accumulation is exact fp32 (v_mfma_f32_32x32x2_f32). */
typedef struct v2v_wgrad_desc {
    const void*  p;          /* [N][OH][OW][p_stride]                                         */
    const void*  q;          /* [N][QH][QW][q_stride]                                         */
    float*       grad;       /* [rows][cols][KH][KW] fp32                                     */
    float*       workspace;  /* v2v_conv_wgrad_workspace(d) bytes                             */
    const void*  zero_page;  /* >= 16 zero bytes, 16-B aligned                                */
    int32_t N, OH, OW, QH, QW;
    int32_t rows, cols;      /* real channel counts of P / Q                                  */
    int32_t p_stride, q_stride;
    int32_t KH, KW, stride, pad, pad_mode;
    int32_t dtype;
    int32_t accumulate;      /* bit 0: add into grad (optimizer .grad buffer) instead of overwriting; bit 1 (+2): grad is
                              * CHANNELS-LAST, [rows][KH][KW][cols] (optim.FlatBuffers keeps 4-D weights that way): the
                              * column order the kernel computes in, so no transpose pass -- and with one K split and
                              * q_stride == cols the tiles are written straight into grad                                 */
} v2v_wgrad_desc;
int64_t v2v_conv_wgrad_workspace(const v2v_wgrad_desc* d);
int     v2v_conv_wgrad(const v2v_wgrad_desc* d, void* stream);

/* In-kernel alternative (v2v_conv_desc.fin_*): the last workgroup to finish an N tile of v2v_conv2d reduces
 * the partial rows and writes scale/shift itself (agent-scope release/acquire hand-off), which removes one
 * launch per norm layer.  Same arithmetic and summation order as v2v_bn_finalize. */

/* Training-mode BatchNorm2d / InstanceNorm2d(batch 1) statistics -> per-channel scale/shift
 * (get_norm_layer, models/networks.py:23-30).  partials: [rows][C][2]; count = N*OH*OW.
 * gamma/beta may be NULL (affine=False).  scale_shift: [4][C] fp32 = scale, shift, mean, invstd (the last
 * two rows are what the backward pass needs).  If running_mean/var
 * are non-NULL they are updated with `momentum` (unbiased variance), as nn.BatchNorm2d does. */
int v2v_bn_finalize(const float* partials, int32_t rows, int32_t C, int64_t count,
                    const float* gamma, const float* beta, float eps,
                    float* scale_shift, float* running_mean, float* running_var, float momentum,
                    double* workspace, void* stream);
/* workspace: NULL, or v2v_bn_finalize_groups(rows) * C * 2 doubles.  With a workspace, layers that leave more than 512
 * statistics rows (large M) are reduced by a parallel two-stage tree (groups x C/64 workgroups, then one per 64
 * channels) instead of one workgroup walking every row; the result is deterministic either way.  Round 4: the second stage
 * runs inside the first stage's launch (the last group of a 64-channel slab finalizes; ticket words from the library's own
 * device pool, self re-arming) -- same bits, one launch fewer; V2V_BN_FIN_FUSED=0 keeps the two launches.  Returns 0 groups
 * for rows <= 512 (single stage, the summation order the in-kernel finalize of v2v_conv2d reproduces). */
int v2v_bn_finalize_groups(int32_t rows);

/* y = act(raw * scale[c] + shift[c]) (+ add0) (+ add1);  raw fp32 NHWC [P][c_stride_raw],
 * y / add0 / add1 activation dtype NHWC [P][c_stride].  Channels >= C of y are written 0.
 * add_after_act: residual is added after the activation (ResnetBlock: act none). */
int v2v_bn_apply(const float* raw, int32_t c_stride_raw, const float* scale_shift,
                 const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                 int32_t act, float act_param, int32_t dtype, void* stream);
/* v2v_bn_apply whose raw operand is stored in `raw_dtype` (V2V_F32: exactly v2v_bn_apply; V2V_BF16: the V2V_OUT_RAW_ACT_NHWC
 * output of a bf16 convolution, [P][c_stride_raw] bf16 with c_stride_raw % 8 == 0).  Same arithmetic on the widened values. */
int v2v_bn_apply_raw(const void* raw, int32_t raw_dtype, int32_t c_stride_raw, const float* scale_shift,
                     const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                     int32_t act, float act_param, int32_t dtype, void* stream);
/* Two bn_apply passes of identical geometry as ONE launch (block y picks its member): the norm + activation
 * (+ residual) passes behind a v2v_conv2d_pair.  Bitwise the result of two v2v_bn_apply calls. */
int v2v_bn_apply_pair(const float* raw_a, const float* scale_shift_a, const void* add0_a, const void* add1_a, void* y_a,
                      const float* raw_b, const float* scale_shift_b, const void* add0_b, const void* add1_b, void* y_b,
                      int32_t c_stride_raw, int64_t P, int32_t C, int32_t c_stride,
                      int32_t act, float act_param, int32_t dtype, void* stream);

/* Backward of bn_finalize + bn_apply (autograd of training-mode BatchNorm2d/InstanceNorm2d + ReLU /
 * LeakyReLU in the reference):  g = dY*act'(raw*scale+shift);  dbeta (+)= sum g;  dgamma (+)= sum g*xhat;
 * dRaw = scale*(g - mean(g) - xhat*mean(g*xhat)), written NHWC in `dtype` with channel stride
 * c_stride_out (pad channels zero).  dy: NHWC `dtype` [P][c_stride]; raw: fp32 [P][c_stride_raw];
 * stats: the [4][C] array of v2v_bn_finalize.  dgamma/dbeta may be NULL.
 * workspace: (v2v_bn_backward_rows(P)*2*C + 2*C) floats. */
int v2v_bn_backward_rows(int64_t P);
int v2v_bn_backward(const void* dy, const float* raw, int32_t c_stride_raw, const float* stats,
                    void* draw, int32_t c_stride_out, float* dgamma, float* dbeta, int32_t accumulate,
                    float* workspace, int64_t P, int32_t C, int32_t c_stride,
                    int32_t act, float act_param, int32_t dtype, void* stream);
/* out[c] (+)= sum over pixels of x[p][c]  (bias gradients).  workspace: v2v_bn_backward_rows(P)*2*C floats */
int v2v_channel_sum(const void* x, float* out, int32_t accumulate, float* workspace,
                    int64_t P, int32_t C, int32_t c_stride, int32_t dtype, void* stream);
/* Backward of a conv epilogue activation without norm: g = dY * act'(y) * out_scale.  dy / y are NHWC
 * `dtype` [P][c_stride_in] (nchw = 0) or planar fp32 [N][C][H][W] (nchw = 1, API-facing heads);
 * g is NHWC `dtype` [P][c_stride_out], pad channels zero. */
int v2v_act_backward(const void* dy, const void* y, void* g, int32_t N, int32_t H, int32_t W, int32_t C,
                     int32_t c_stride_in, int32_t c_stride_out, int32_t nchw, int32_t act, float act_param,
                     float out_scale, int32_t dtype, void* stream);

/* AvgPool2d(3, stride 2, pad 1, count_include_pad=False) on planar fp32 [planes][H][W]
 * (build_pyr, models/base_model.py:122-134; MultiscaleDiscriminator.downsample :652). */
int v2v_avgpool3s2_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
/* same on NHWC activations */
int v2v_avgpool3s2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                        int32_t dtype, void* stream);

int v2v_avgpool3s2_nhwc_backward(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                                 int32_t dtype, void* stream);   /* H, W: the pooled layer's INPUT size */

/* MaxPool2d(2, stride 2) on NHWC activations: the four pooling stages of torchvision's VGG19 `features` (indices 4, 9,
 * 18, 27) inside Vgg19 (models/networks.py:840-870).  Floor output size; ties keep the first window element
 * (ATen max_pool2d), which decides where the backward pass routes dY.  c_stride % (16 bytes) == 0. */
int v2v_maxpool2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                      int32_t dtype, void* stream);
int v2v_maxpool2_nhwc_backward(const void* dy, const void* x, void* dx, int32_t N, int32_t H, int32_t W,
                               int32_t c_stride, int32_t dtype, void* stream);   /* H, W: the pooled layer's INPUT size */
/* AvgPool2d(2, stride 2, count_include_pad=False), no padding, planar fp32 [planes][H][W]
 * (VGGLoss.downsample, models/networks.py:782,785-786: applied while the width exceeds 1024). */
int v2v_avgpool2_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
int v2v_avgpool2_planar_backward(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, void* stream);
/* Planar fp32 one-hot (+ instance edge plane when inst != NULL) of ONE label frame:
 * out[(label_nc + (inst != NULL))][H][W] = real_A[0][0, -1] of Vid2VidModelG.inference (models/vid2vid_model_G.py:209;
 * encode_input :86-112, get_edges models/base_model.py:146-152). */
int v2v_onehot_planar(const float* labels, const float* inst, float* out, int32_t H, int32_t W,
                      int32_t label_nc, void* stream);
int v2v_onehot_planar_u8(const uint8_t* labels, const int32_t* inst, float* out, int32_t H, int32_t W,
                         int32_t label_nc, void* stream);

/* Instance-wise average pooling of Encoder.forward (models/networks.py:621-632) for ONE sample: out[c][p] = mean of
 * feat[c][q] over the pixels q with inst[q] == inst[p].  feat / out: planar fp32 [C][HW]; inst: fp32 [HW] holding integer ids
 * (any values).  workspace: v2v_instance_mean_workspace(C, HW) bytes, 4-byte aligned, owned by the caller.  Sums use float
 * atomics: the summation order (1e-7-level rounding) is run-dependent.  Up to 2048 distinct ids per sample probe quickly;
 * beyond 4096 the excess pixels keep their own value. */
int64_t v2v_instance_mean_workspace(int32_t C, int64_t HW);
int v2v_instance_mean_planar(const float* feat, const float* inst, float* out, void* workspace,
                             int32_t C, int64_t HW, void* stream);

/* encode_input (models/vid2vid_model_G.py:86-112) + get_edges (models/base_model.py:146-152)
 * + compute_mask (:322-330) for `T` frames, written straight to the NHWC stem input:
 *   out[h][w][t*(label_nc+use_inst) + c] = (label[t][h][w] == c),  edge in channel label_nc.
 * labels / inst: fp32 planar [T][H][W] holding integers (Appendix C.10).  mask: [H][W] fp32 =
 * clamp(sum_{l in fg_labels} onehot_l of frame T-1, 0, 1) or NULL. */
int v2v_encode_labels(const float* labels, const float* inst, void* out, float* mask,
                      int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                      const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream);
/* The same with the label map as uint8 and the instance map as int32 (what a loader holds before the
 * reference multiplies by 255 and casts to float, data/temporal_dataset.py:60-70): 4x less host-to-device traffic. */
int v2v_encode_labels_u8(const uint8_t* labels, const int32_t* inst, void* out, float* mask,
                         int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                         const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream);
/* encode_input + get_edges + ONE level of build_pyr (AvgPool2d(3, 2, 1, count_include_pad=False), base_model.py:122-134) straight
 * from the label / instance maps (fp32-encoded integers, or uint8 / int32 with maps_u8): out is NHWC [(H-1)/2+1][(W-1)/2+1][c_stride],
 * bit-identical to v2v_avgpool3s2_nhwc of the v2v_encode_labels output, which is never materialised; mask (optional) is the
 * FULL-resolution foreground mask [H][W] as v2v_encode_labels writes it.
 * maps_u8 == 2: `labels` are the [T][H][W] 1-byte codes of v2v_label_codes (label | edge << 7; label_nc <= 126); `inst` is not read,
 * non-NULL says that the frames carry an edge channel.  Same output bit for bit, nine independent byte loads per frame and thread. */
int v2v_encode_labels_pooled(const void* labels, const void* inst, void* out, float* mask,
                             int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                             const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, int32_t maps_u8, void* stream);

/* compute_mask on an NHWC (possibly AvgPool'ed) label tensor: mask[p] = clamp(sum_i x[p][base_ch+fg[i]],0,1) */
int v2v_fg_mask_nhwc(const void* x, float* mask, int64_t P, int32_t c_stride, int32_t base_ch,
                     const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream);

/* planar fp32 NCHW [C][H][W] -> NHWC activation dtype [H][W][c_stride] (zero channel padding) */
/* bf16x3 operand of the fp32 engine's "x3" mode: x fp32 NHWC [pixels][cs_in] (C real channels, C % 4 == 0) -> bf16 NHWC
 * [pixels][cs_out], channels [hi | lo | hi] with hi = bf16(x), lo = bf16(x - hi).  Convolved with weights laid out [hi(W) | hi(W) |
 * lo(W)] along the input channels (any conv kernel of this library, dtype V2V_BF16, cin = 3 C) this gives x * W to ~2^-17 relative. */
int v2v_split_x3(const float* x, void* y, int64_t pixels, int32_t C, int32_t cs_in, int32_t cs_out, void* stream);
/* v2v_bn_apply / v2v_bn_apply_pair in fp32 (raw_b == NULL: one member) that also write the result as the bf16x3 operand
 * [pixels][3 C] of the consumer convolution: no separate v2v_split_x3 pass.  Dense channel stride (c_stride == C), C % 4 == 0. */
int v2v_bn_apply_x3(const float* raw_a, const float* scale_shift_a, const void* add0_a, const void* add1_a, void* y_a, void* x3_a,
                    const float* raw_b, const float* scale_shift_b, const void* add0_b, const void* add1_b, void* y_b, void* x3_b,
                    int32_t c_stride_raw, int64_t P, int32_t C, int32_t act, float act_param, void* stream);
int v2v_pack_nchw_to_nhwc(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                          int32_t c_stride, int32_t dtype, void* stream);
/* NHWC activation dtype -> planar fp32 NCHW */
int v2v_unpack_nhwc_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t c_stride, int32_t dtype, void* stream);

/* torch.cat([x0, x1], dim=1) of planar fp32 tensors written straight to NHWC (discriminator inputs,
 * models/vid2vid_model_D.py:169-170,185-187); x1 may be NULL (C1 = 0) and is multiplied by scale1
 * (the temporal discriminator sees flow_ref/20, vid2vid_model_D.py:107-108). */
int v2v_pack_concat_nhwc(const float* x0, int32_t C0, const float* x1, int32_t C1, float scale1, void* y,
                         int32_t N, int32_t H, int32_t W, int32_t c_stride, int32_t dtype, void* stream);
/* channels [c_offset, c_offset+C) of an NHWC tensor -> planar fp32 NCHW (backward of the above) */
int v2v_unpack_channels_nchw(const void* y, float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                             int32_t c_stride, int32_t c_offset, int32_t dtype, void* stream);
/* backward of nn.ReflectionPad2d(pad): xp [N][H+2p][W+2p][cs] -> x [N][H][W][cs] */
int v2v_reflect_pad_fold(const void* xp, void* x, int32_t N, int32_t H, int32_t W, int32_t pad,
                         int32_t c_stride, int32_t dtype, void* stream);

/* y = a + b on NHWC activations, n_elems = N*H*W*c_stride (models/networks.py:299,305,319) */
int v2v_add_nhwc(const void* a, const void* b, void* y, int64_t n_elems, int32_t dtype, void* stream);

/* Composite tail (models/networks.py:216-230): all tensors planar fp32 NCHW.
 *   warp   = grid_sample(prev[N][C][H][W], grid + flow/((W-1)/2,(H-1)/2), bilinear, border,
 *                        align_corners as given)               (BaseNetwork.resample :108-115)
 *   final  = raw*w + warp*(1-w)            (skipped when flow == NULL: final = raw)
 *   if fg != NULL: final = fg*m + final*(1-m); raw = fg*m + raw*(1-m)   (img_raw updated in place)
 * gx[W], gy[H]: the base grid torch.linspace(-1,1,W/H) of get_grid (:79-93).  img_warp may be NULL. */
int v2v_warp_blend(float* img_raw, const float* flow, const float* weight, const float* prev,
                   const float* fg, const float* mask, float* img_final, float* img_warp,
                   const float* gx, const float* gy,
                   int32_t N, int32_t C, int32_t H, int32_t W, int32_t align_corners, void* stream);

/* Backward of v2v_warp_blend.  raw = the PRE-blend img_raw; d_rawout = gradient w.r.t. the blended
 * img_raw output (NULL if unused); d_prev (optional) must be pre-zeroed, it is accumulated atomically
 * like ATen's grid_sampler backward.  Gradient through the border clip follows ATen
 * (clip_coordinates_set_grad: a clipped coordinate passes no gradient to the flow). */
int v2v_warp_blend_backward(const float* d_final, const float* d_rawout, const float* raw, const float* flow,
                            const float* weight, const float* prev, const float* fg, const float* mask,
                            const float* gx, const float* gy, float* d_raw, float* d_flow, float* d_weight,
                            float* d_prev, float* d_fg, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t align_corners, void* stream);
int v2v_resample_flow_backward(const float* d_out, const float* img, const float* flow, const float* gx,
                               const float* gy, float* d_img, float* d_flow, int32_t N, int32_t C,
                               int32_t H, int32_t W, int32_t align_corners, void* stream);

/* F.grid_sample(bilinear, border) driven by a pixel-unit flow, planar fp32 (resample()). */
int v2v_resample_flow(const float* img, const float* flow, float* out, const float* gx, const float* gy,
                      int32_t N, int32_t C, int32_t H, int32_t W, int32_t align_corners, void* stream);

/* ---- FlowNet2 native ops: same argument meaning as the reference pybind11 modules ---- */
/* correlation_cuda.forward (correlation_cuda.cc:10-87, kernel correlation_cuda_kernel.cu:73-147).
 * in1/in2: planar NCHW fp32 [N][C][H][W]; out: [N][(2*(max_disp/stride2)+1)^2][OH][OW],
 * OH = ceil((H + 2*pad - 2*(max_disp + (k-1)/2)) / stride1).  rbot scratch is not needed. */
int v2v_correlation_out_size(int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                             int32_t max_displacement, int32_t stride1, int32_t stride2,
                             int32_t* out_c, int32_t* out_h, int32_t* out_w);
int v2v_correlation_forward(const float* in1, const float* in2, float* out,
                            int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                            int32_t stride1, int32_t stride2, int32_t corr_type_multiply, void* stream);
/* The same correlation for FlowNetC's geometry class (kernel_size 1, stride1 1, stride2 2, pad = max_displacement: FlowNetC.py:31)
 * on the matrix pipe, between two NHWC activation tensors [N][H][W][cs_in] (C real channels, dtype = activation dtype), fused with
 * what surrounds it in FlowNetC.forward (FlowNetC.py:86-93): the result / C passes LeakyReLU(leaky_slope) and lands as
 * (2 max_disp / stride2 + 1)^2 consecutive channels at channel c_off of the NHWC tensor `out` [N][H][W][cs_out] -- the concat buffer
 * of conv3_1.  C must be a multiple of 16 (bf16) / 8 (fp32). */
int v2v_correlation_nhwc(const void* f1, const void* f2, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                         int32_t cs_in, int32_t cs_out, int32_t c_off, int32_t max_displacement, int32_t stride2,
                         float leaky_slope, int32_t dtype, void* stream);
/* resample2d_cuda.forward (resample2d_kernel.cu:15-64): out[b,c,y,x] = bilinear(img, x+fx, y+fy) */
int v2v_resample2d_forward(const float* img, const float* flow, float* out,
                           int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                           int32_t kernel_size, void* stream);
/* channelnorm_cuda.forward (channelnorm_kernel.cu:18-60): out[b,0,y,x] = sqrt(sum_c x^2) */
int v2v_channelnorm_forward(const float* x, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t norm_deg, void* stream);

/* ---- FlowNet2 glue (models/flownet2_pytorch/models.py:96-161, models/flownet.py:43-59) ---- */
/* x1, x2 [B][3][HW] = (im1, im2 - rgb_mean) / rgb_max, rgb_mean per (b, c) over both frames (models.py:97-102).
 * workspace: B*3*64 floats */
int v2v_flownet_normalize(const float* im1, const float* im2, float* x1, float* x2, float* workspace,
                          int32_t B, int32_t H, int32_t W, float rgb_max, void* stream);
/* warped = Resample2d(img1, flow); out_norm = ChannelNorm(img0 - warped) (mode 0) or
 * (sum_c (img0-warped)^2 < threshold) as 0/1 (mode 1, flownet.py:55).  img0 / img1: planar [B][>=C][HW] with the
 * given batch strides (channel slices of the 6-channel x); warped / out_norm may be NULL. */
int v2v_warp_diff_norm(const float* img0, int64_t batch_stride0, const float* img1, int64_t batch_stride1,
                       const float* flow, float* warped, float* out_norm, int32_t B, int32_t C, int32_t H,
                       int32_t W, int32_t mode, float threshold, void* stream);
/* nn.Upsample to (OH, OW): bilinear (align_corners False) or nearest, result * out_scale */
int v2v_resize_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                      int32_t bilinear, float out_scale, void* stream);
/* planar fp32 [N][C][HW] -> channels [c_offset, c_offset+C) of an NHWC buffer, y = leaky_relu(x*scale, slope) */
int v2v_pack_channels_nhwc(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                           int32_t c_stride, int32_t c_offset, float scale, float leaky_slope,
                           int32_t dtype, void* stream);
/* torch.cat along channels on NHWC activations: dst[p][dst_offset + c] = src[p][src_offset + c], c < C */
int v2v_concat_channels_nhwc(const void* src, int32_t src_stride, int32_t src_offset, void* dst,
                             int32_t dst_stride, int32_t dst_offset, int32_t C, int64_t P,
                             int32_t dtype, void* stream);

/* ---- losses and optimizer (models/networks.py:731-812, models/vid2vid_model_D.py:199-213; torch.optim.Adam) ----
 * kind: 0 = mean((a - target)^2) (LSGAN GANLoss), 1 = mean(|a*m - b*m|) (L1Loss / MaskedL1Loss); result * weight.
 * NHWC mode (planar = 0): a, b are [P][c_stride] activations of `dtype`, mean over P*C real elements;
 * planar mode: fp32 [N][CHW/HW][HW], optional mask [N][1][HW] broadcast over channels.
 * workspace: v2v_loss_workspace_floats() floats.  backward: da = d(out)/da * grad_out[0] (device scalar). */
#define V2V_LOSS_MSE_CONST 0
#define V2V_LOSS_L1        1
int v2v_loss_workspace_floats(void);
int v2v_loss_forward(int32_t kind, const void* a, const void* b, const float* mask, float target, float weight,
                     int64_t P, int32_t C, int32_t c_stride, int64_t N, int64_t CHW, int64_t HW, int32_t planar,
                     float* workspace, float* out, int32_t dtype, void* stream);
int v2v_loss_backward(int32_t kind, const void* a, const void* b, const float* mask, float target, float weight,
                      int64_t P, int32_t C, int32_t c_stride, int64_t N, int64_t CHW, int64_t HW, int32_t planar,
                      const float* grad_out, void* da, int32_t dtype, void* stream);
/* one fused Adam update over flat fp32 buffers; `step` is the 1-based count of this update; the gradient is
 * read as grad * grad_scale (1/world after a sum all-reduce, so averaging costs no extra pass) */
int v2v_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  int32_t step, void* stream);
/* The same update with its step count and learning rate in DEVICE memory: `state` = {int32 step; float lr} (8 bytes, owned
 * by the caller, step = number of updates done so far).  The launch increments `step` first, then applies update number
 * `step` -- so a launch captured into a hipGraph (stream capture of a whole training chunk) advances the bias corrections on
 * every replay, and the host changes the learning rate by writing state->lr (torch.optim.Adam(capturable=True) is the
 * reference-side analogue; the reference's optimizers are models/vid2vid_model_G.py:84, vid2vid_model_D.py:86-91). */
int v2v_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      void* state, void* stream);
int v2v_memset_zero(void* p, int64_t bytes, void* stream);

/* ---- plan executor: a recorded launch sequence replayed as one call / one hipGraph ---- */
typedef struct v2v_plan v2v_plan;
v2v_plan* v2v_plan_create(void);
void      v2v_plan_destroy(v2v_plan* p);
int       v2v_plan_begin_record(v2v_plan* p);   /* subsequent v2v_* launches are recorded, not run */
int       v2v_plan_end_record(v2v_plan* p);
int       v2v_plan_num_ops(const v2v_plan* p);
int       v2v_plan_run(v2v_plan* p, void* stream);              /* eager replay                   */
int       v2v_plan_instantiate_graph(v2v_plan* p, void* stream);/* capture into a hipGraphExec    */
int       v2v_plan_launch_graph(v2v_plan* p, void* stream);
/* The launch program v2v_plan_instantiate_graph builds for a plan with lanes (no device needed): steps[3*i..] = {0, segment,
 * lane} launch a linear graph of consecutive ops of one lane, or {1, waiter, signal} an event edge; seg_of_op[i] = segment of
 * recorded op i (-1 for lane_wait ops; may be NULL).  Returns the number of steps. */
int       v2v_plan_segment_program(const v2v_plan* p, int32_t* steps, int32_t max_steps, int32_t* seg_of_op, int32_t n_ops);
/* per-op HIP-event timing of one eager replay: ms[i] for op i, names via v2v_plan_op_name */
int       v2v_plan_profile(v2v_plan* p, void* stream, float* ms, int32_t n);
/* concurrent timeline of one eager replay with every lane on its own stream: start / end of op i in ms since the replay began,
 * and the lane it ran on (lanes may be NULL); the schedule the lanes produce, which a serialising kernel trace cannot show.
 * Time comes from a one-thread kernel storing the device wall clock before / after every op (timing events of different streams
 * gave inconsistent intervals), which costs a few us per op. */
int       v2v_plan_timeline(v2v_plan* p, void* stream, float* t0_ms, float* t1_ms, int32_t* lanes, int32_t n);
/* the same measurement with the ops and the stamps captured into a hipGraph (the schedule of the real graph replay) */
int       v2v_plan_timeline_graph(v2v_plan* p, void* stream, float* t0_ms, float* t1_ms, int32_t* lanes, int32_t n);
const char* v2v_plan_op_name(const v2v_plan* p, int32_t i);
int       v2v_plan_op_lane(const v2v_plan* p, int32_t i);       /* lane of op i; for a lane_wait op: waiter | signal << 8 */
int       v2v_plan_set_label(v2v_plan* p, const char* label);   /* labels the last recorded op */
const char* v2v_plan_op_label(const v2v_plan* p, int32_t i);
/* Lanes = parallel branches of a recorded plan (0..7, thread-local; default 0).  Ops recorded after v2v_plan_set_lane(k)
 * belong to lane k; v2v_plan_lane_wait(w, s) records the only kind of cross-lane edge: lane w continues after everything
 * recorded so far on lane s (a lane's first use must be preceded by such a wait = the fork).  v2v_plan_instantiate_graph
 * captures each lane on its own stream (parallel hipGraph paths, all joined into lane 0 at the end); eager replays run in
 * recording order.  Outside a recording both calls are no-ops on the data path. */
int       v2v_plan_set_lane(int32_t lane);
int       v2v_plan_lane_wait(int32_t waiter, int32_t signal);

/* Visualisation conversions on the device (util/util.py:48-89 of the reference does them in numpy after a D2H copy of
 * the fp32 planes): x is planar fp32 [C][H][W]; out is uint8 [H][W][C] (tensor2im, C <= 3; normalize: (x+1)/2*255, else
 * x*255, clipped, truncated) or uint8 [H][W][3] (tensor2label: argmax over C > 1 planes or the stored id for C == 1,
 * coloured through cmap[n_label][3]).  Integer-exact against the reference's numpy arithmetic. */
int v2v_tensor2im(const float* x, uint8_t* out, int32_t C, int32_t H, int32_t W, int32_t normalize, void* stream);
/* tensor2flow (util/util.py:89-107): flow planar fp32 [2][H][W] -> uint8 [H][W][3], hue = direction / 2 degrees, value = magnitude
 * min-max normalised over the image, full saturation, HSV -> RGB as OpenCV's 8-bit conversion.  range_ws: 2 uint32 of scratch. */
int v2v_tensor2flow(const float* flow, uint8_t* out, uint32_t* range_ws, int32_t H, int32_t W, void* stream);
int v2v_tensor2label(const float* x, uint8_t* out, const uint8_t* cmap, int32_t n_label, int32_t C, int32_t H, int32_t W,
                     void* stream);

/* 7x7 stem convolution over one-hot label input as a weight gather-sum.  Replaces, for label-map input,
 * encode_input (models/vid2vid_model_G.py:86-112) + ReflectionPad2d(3) + Conv2d(T*(label_nc+1), cout, 7) at the head of
 * the label and foreground towers (models/networks.py:128-133, 153-156): per output pixel and tap exactly one label
 * plane per frame is 1, so the convolution is T gathered weight rows (+ the edge row where the instance map has an
 * edge) per tap -- 1/36 of the dense operations, same sums.  Output: raw fp32 NHWC [H][W][cout_stride] (bias added)
 * + optional per-tile statistics rows [v2v_onehot_conv_stats_rows(H,W)][cout][2] for v2v_bn_finalize, exactly like
 * v2v_conv2d with V2V_OUT_RAW_F32_NHWC.  labels / inst: [T][H][W] fp32-encoded integers (in_u8 = 0) or uint8 / int32
 * (in_u8 = 1); inst may be NULL (no edge plane; the layer then has T*label_nc input channels).  Labels outside
 * [0, label_nc) select no plane.
 * table: v2v_onehot_conv_table_bytes(cin, cout, dtype, slice, T, label_nc) bytes, filled by v2v_onehot_conv_pack_weights
 * from the layer's fp32 [cout][cin][7][7] weight (cin = T * (label_nc + (inst ? 1 : 0))).  cout <= 128.  With a bf16 table the
 * instance-edge planes are not gathered row by row (an edge pixel anywhere in a wave made all 64 lanes add the row): they
 * are a [pixels][49 T] x [49 T][cout] product of 0 / 1 edge bits with the edge rows on the matrix pipe, added in the epilogue.
 * slice: output channels per workgroup, 32 or 64, 0 = default; the same value must be given to all three calls.
 * in_u8 = 2: `labels` is the [T][H][W] byte map of v2v_label_codes (label | edge << 7, 127 = no plane) and `inst` only says
 * whether the layer has edge planes (any non-NULL pointer): the per-workgroup staging then reads one byte per halo entry. */
int     v2v_label_codes(const void* labels, const void* inst, int32_t in_u8, uint8_t* codes, int32_t T, int32_t H, int32_t W,
                        int32_t label_nc, void* stream);
/* (declarations of the stem entry points follow) */
int64_t v2v_onehot_conv_table_bytes(int32_t cin, int32_t cout, int32_t dtype, int32_t slice, int32_t T, int32_t label_nc);
int     v2v_onehot_conv_pack_weights(const float* w, void* table, int32_t cin, int32_t cout, int32_t dtype, int32_t slice,
                                     int32_t T, int32_t label_nc, void* stream);
int     v2v_onehot_conv_stats_rows(int32_t H, int32_t W);
int     v2v_onehot_conv7x7(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                           float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                           int32_t cout, int32_t cout_stride, int32_t dtype, int32_t slice, void* stream);
/* Same launch + the training-mode norm statistics finalized by the last workgroup of each channel slice (what
 * v2v_conv_desc.fin_* does for v2v_conv2d): scale_shift receives [4][cout] = scale, shift, mean, invstd. */
typedef struct v2v_onehot_norm {
    int32_t* counter;          /* >= 4 zero-initialised ints, re-armed in-kernel                  */
    const float* gamma;        /* [cout] or NULL                                                  */
    const float* beta;         /* [cout] or NULL                                                  */
    float* scale_shift;        /* [4][cout] out                                                   */
    float* running_mean;       /* [cout] or NULL                                                  */
    float* running_var;        /* [cout] or NULL                                                  */
    float eps, momentum;
    int64_t count;             /* H*W                                                             */
} v2v_onehot_norm;
int     v2v_onehot_conv7x7_norm(const void* labels, const void* inst, int32_t in_u8, const void* table, const float* bias,
                                float* out, float* stats, int32_t T, int32_t H, int32_t W, int32_t label_nc,
                                int32_t cout, int32_t cout_stride, int32_t dtype, int32_t slice, const v2v_onehot_norm* fin,
                                void* stream);

/* recordable device-to-device copy (rolling fake_B_prev window, vid2vid_model_G.py:228) */
int v2v_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream);

/* Dry run (process-wide switch, returns the previous setting): every entry point still validates its arguments and
 * plans still record, but NOTHING is launched -- v2v_* launch calls outside a recording and v2v_plan_run return 0
 * immediately.  For CPU hosts without a GPU: the test-suite drives the reference's own train.py / test.py control flow
 * through the model API this way.  Not a compute path: outputs are never written. */
int         v2v_set_dry_run(int32_t on);
int         v2v_get_dry_run(void);

/* library / device info */
int         v2v_version(void);
const char* v2v_last_error(void);
/* Asynchronous device-side failures (kernels cannot return an error): a host-mapped status word the kernels OR bits into.
 * bit 0: a fused-norm spin barrier timed out (see "Fused norm"), the outputs of that launch are NaN.  Reading it needs no
 * stream synchronisation; the value reflects the launches that have executed so far.  clear != 0 resets it. */
int         v2v_device_status(int32_t clear);
int         v2v_device_info(int32_t* cus, int32_t* lds_per_cu, int64_t* hbm_bytes, char* arch, int32_t arch_len);

#ifdef __cplusplus
}
#endif
#endif /* V2V_HIP_H */
