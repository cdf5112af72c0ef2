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
    int32_t stride;         /* 1 or 2 (transposed: always 2)                                 */
    int32_t pad;            /* symmetric padding                                             */
    int32_t pad_mode;       /* V2V_PAD_*  (transposed: zero only)                            */
    int32_t transposed;     /* 0 = Conv2d, 1 = ConvTranspose2d(stride 2, OH = 2H)            */
    int32_t OH, OW;         /* output height / width                                         */
    int32_t dtype;          /* V2V_F32 / V2V_BF16 : dtype of in, w (and out in ACT mode)     */
    int32_t out_mode;       /* V2V_OUT_*                                                     */
    int32_t act;            /* V2V_ACT_* (ignored for RAW output)                            */
    float   act_param;      /* leaky slope                                                   */
    float   out_scale;      /* multiplies the activated value (flow heads: 20 * 2^scale)     */
    int32_t tile;           /* 0 = auto, else force tile config id (testing/tuning)          */
} v2v_conv_desc;

/* Packed weight layout.  One matrix per class (Conv2d: 1 class; ConvTranspose2d stride 2:
 * 4 output-parity classes (a,b)), each [round_up(cout,128)][kpad] with
 * k = tap * cin_stride + c (taps row-major over the class's (kh,kw) list) and kpad = k
 * rounded up to 128 bytes; zero filled.  v2v_conv_packed_elems gives the total element count;
 * v2v_conv_pack_weights is the device kernel that fills it from PyTorch-layout fp32 weights
 * [cout][cin][KH][KW] (Conv2d) or [cin][cout][KH][KW] (ConvTranspose2d), converting to `dtype`. */
int64_t v2v_conv_packed_elems(int32_t cin, int32_t cin_stride, int32_t cout, int32_t KH, int32_t KW,
                              int32_t transposed, int32_t pad, int32_t dtype);
int     v2v_conv_pack_weights(const float* w, void* dst, int32_t cin, int32_t cin_stride, int32_t cout,
                              int32_t KH, int32_t KW, int32_t transposed, int32_t pad, int32_t dtype,
                              void* stream);

/* Number of statistics rows (n_classes * m_tiles) the launch described by `d` writes. */
int     v2v_conv_stats_rows(const v2v_conv_desc* d);
/* Tile configuration id the launch would use (after auto selection). */
int     v2v_conv_tile_config(const v2v_conv_desc* d);
/* Launch.  nn.Conv2d / nn.ConvTranspose2d forward (models/networks.py:132-183 etc.). */
int     v2v_conv2d(const v2v_conv_desc* d, void* stream);

/* Training-mode BatchNorm2d / InstanceNorm2d(batch 1) statistics -> per-channel scale/shift
 * (get_norm_layer, models/networks.py:23-30).  partials: [rows][C][2]; count = N*OH*OW.
 * gamma/beta may be NULL (affine=False).  scale_shift: [2][C] fp32.  If running_mean/var
 * are non-NULL they are updated with `momentum` (unbiased variance), as nn.BatchNorm2d does. */
int v2v_bn_finalize(const float* partials, int32_t rows, int32_t C, int64_t count,
                    const float* gamma, const float* beta, float eps,
                    float* scale_shift, float* running_mean, float* running_var, float momentum,
                    void* stream);

/* y = act(raw * scale[c] + shift[c]) (+ add0) (+ add1);  raw fp32 NHWC [P][c_stride_raw],
 * y / add0 / add1 activation dtype NHWC [P][c_stride].  Channels >= C of y are written 0.
 * add_after_act: residual is added after the activation (ResnetBlock: act none). */
int v2v_bn_apply(const float* raw, int32_t c_stride_raw, const float* scale_shift,
                 const void* add0, const void* add1, void* y, int64_t P, int32_t C, int32_t c_stride,
                 int32_t act, float act_param, int32_t dtype, void* stream);

/* AvgPool2d(3, stride 2, pad 1, count_include_pad=False) on planar fp32 [planes][H][W]
 * (build_pyr, models/base_model.py:122-134; MultiscaleDiscriminator.downsample :652). */
int v2v_avgpool3s2_planar(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
/* same on NHWC activations */
int v2v_avgpool3s2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t c_stride,
                        int32_t dtype, void* stream);

/* encode_input (models/vid2vid_model_G.py:86-112) + get_edges (models/base_model.py:146-152)
 * + compute_mask (:322-330) for `T` frames, written straight to the NHWC stem input:
 *   out[h][w][t*(label_nc+use_inst) + c] = (label[t][h][w] == c),  edge in channel label_nc.
 * labels / inst: fp32 planar [T][H][W] holding integers (Appendix C.10).  mask: [H][W] fp32 =
 * clamp(sum_{l in fg_labels} onehot_l of frame T-1, 0, 1) or NULL. */
int v2v_encode_labels(const float* labels, const float* inst, void* out, float* mask,
                      int32_t T, int32_t H, int32_t W, int32_t label_nc, int32_t c_stride,
                      const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream);

/* compute_mask on an NHWC (possibly AvgPool'ed) label tensor: mask[p] = clamp(sum_i x[p][base_ch+fg[i]],0,1) */
int v2v_fg_mask_nhwc(const void* x, float* mask, int64_t P, int32_t c_stride, int32_t base_ch,
                     const int32_t* fg_labels_dev, int32_t n_fg, int32_t dtype, void* stream);

/* planar fp32 NCHW [C][H][W] -> NHWC activation dtype [H][W][c_stride] (zero channel padding) */
int v2v_pack_nchw_to_nhwc(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                          int32_t c_stride, int32_t dtype, void* stream);
/* NHWC activation dtype -> planar fp32 NCHW */
int v2v_unpack_nhwc_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
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
/* resample2d_cuda.forward (resample2d_kernel.cu:15-64): out[b,c,y,x] = bilinear(img, x+fx, y+fy) */
int v2v_resample2d_forward(const float* img, const float* flow, float* out,
                           int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                           int32_t kernel_size, void* stream);
/* channelnorm_cuda.forward (channelnorm_kernel.cu:18-60): out[b,0,y,x] = sqrt(sum_c x^2) */
int v2v_channelnorm_forward(const float* x, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t norm_deg, void* stream);

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
/* per-op HIP-event timing of one eager replay: ms[i] for op i, names via v2v_plan_op_name */
int       v2v_plan_profile(v2v_plan* p, void* stream, float* ms, int32_t n);
const char* v2v_plan_op_name(const v2v_plan* p, int32_t i);
int       v2v_plan_set_label(v2v_plan* p, const char* label);   /* labels the last recorded op */
const char* v2v_plan_op_label(const v2v_plan* p, int32_t i);

/* recordable device-to-device copy (rolling fake_B_prev window, vid2vid_model_G.py:228) */
int v2v_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream);

/* library / device info */
int         v2v_version(void);
const char* v2v_last_error(void);
int         v2v_device_info(int32_t* cus, int32_t* lds_per_cu, int64_t* hbm_bytes, char* arch, int32_t arch_len);

#ifdef __cplusplus
}
#endif
#endif /* V2V_HIP_H */
