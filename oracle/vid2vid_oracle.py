"""CPU ORACLE for the vid2vid hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under vid2vid_amd/ does (tests/test_cpu_boundary.py enforces it).

A functional fp32 restatement (plain torch CPU ops) of the reference's algorithm, driven by
state_dicts that use the reference's parameter names.  Every function cites the reference
lines it follows (paths relative to the reference root).

Pinning (tests/test_cpu_oracle.py):
  * generators / discriminators / inference() are pinned against tests/golden/*.npz, which
    tests/golden/make_golden.py produced by executing the reference's own Python on CPU;
  * correlation / resample2d / channelnorm are CUDA-only in the reference (no nvcc, legacy ATen
    API, warp-32 code) and the reference ships no test vectors for them: their restatements
    below follow the .cu files line by line and are checked against hand-computed cases only
    -> "parity unpinned by the reference" for these three.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# layer primitives
# --------------------------------------------------------------------------------------
def _conv(sd, key, x, stride=1, padding=0):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def _convT(sd, key, x):
    # nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1)   models/networks.py:147,176
    return F.conv_transpose2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=2, padding=1, output_padding=1)


def _norm(sd, key, x, norm):
    """get_norm_layer (models/networks.py:23-30).  Modules are never put in eval(), so
    'batch' = batch statistics + affine, 'instance' = per-sample statistics, no affine."""
    if norm == "batch":
        return F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.1, 1e-5)
    if norm == "instance":
        return F.instance_norm(x, eps=1e-5)
    raise ValueError(norm)


def _reflect(x, p):
    return F.pad(x, (p, p, p, p), mode="reflect")


class _Walker:
    """Index bookkeeping for the reference's nn.Sequential lists: prefix.<idx>.<param>"""

    def __init__(self, sd, prefix, norm):
        self.sd, self.p, self.i, self.norm = sd, prefix, 0, norm

    def key(self, off=0):
        return "%s.%d" % (self.p, self.i + off)

    def stem7(self, x, act=True):
        # [ReflectionPad2d(3), Conv2d(k7), norm, ReLU]        models/networks.py:153
        x = _conv(self.sd, self.key(1), _reflect(x, 3))
        x = F.relu(_norm(self.sd, self.key(2), x, self.norm))
        self.i += 4
        return x

    def down3(self, x):
        # [Conv2d(k3, s2, p1), norm, ReLU]                     models/networks.py:156-157
        x = F.relu(_norm(self.sd, self.key(1), _conv(self.sd, self.key(0), x, 2, 1), self.norm))
        self.i += 3
        return x

    def up3(self, x):
        # [ConvTranspose2d(k3, s2, p1, op1), norm, ReLU]       models/networks.py:176-177
        x = F.relu(_norm(self.sd, self.key(1), _convT(self.sd, self.key(0), x), self.norm))
        self.i += 3
        return x

    def resblock(self, x):
        # ResnetBlock with reflect padding                     models/networks.py:554-593
        k = "%s.%d.conv_block" % (self.p, self.i)
        h = F.relu(_norm(self.sd, k + ".2", _conv(self.sd, k + ".1", _reflect(x, 1)), self.norm))
        h = _norm(self.sd, k + ".6", _conv(self.sd, k + ".5", _reflect(h, 1)), self.norm)
        self.i += 1
        return x + h

    def head7(self, x, act=None):
        # [ReflectionPad2d(3), Conv2d(k7)] (+ Tanh / Sigmoid)  models/networks.py:178,182-183
        x = _conv(self.sd, self.key(1), _reflect(x, 3))
        self.i += 2 + (1 if act else 0)
        if act == "tanh":
            return torch.tanh(x)
        if act == "sigmoid":
            return torch.sigmoid(x)
        return x


# --------------------------------------------------------------------------------------
# first-frame generators with instance-wise feature encoding (models/networks.py:421-632)
# Oracle only so far: the HIP lowering of these three is the next row to build.
# --------------------------------------------------------------------------------------
def global_with_z(sd, x, z, n_downsample_G, n_blocks, norm="instance"):
    """Global_with_z.forward (models/networks.py:460-467); layers :430-458."""
    zd = z
    for _ in range(n_downsample_G):
        zd = avgpool3s2(zd)
    w = _Walker(sd, "model_downsample", norm)
    h = w.stem7(torch.cat([x, z], 1))
    for _ in range(n_downsample_G):
        h = w.down3(h)
    w = _Walker(sd, "model_resnet", norm)
    h = torch.cat([h, zd], 1)
    for _ in range(n_blocks):
        h = w.resblock(h)
    w = _Walker(sd, "model_upsample", norm)
    h = torch.cat([h, zd], 1)
    for _ in range(n_downsample_G):
        h = w.up3(h)
    return _Walker(sd, "model_upsample_conv", norm).head7(torch.cat([h, z], 1), "tanh")


def local_with_z(sd, x, z, n_downsample_global, n_blocks_global, n_local_enhancers, n_blocks_local, norm="instance"):
    """Local_with_z.forward (models/networks.py:512-552); layers :476-510."""
    pyr = [x]
    for _ in range(n_local_enhancers):
        pyr.append(avgpool3s2(pyr[-1]))
    z_local = z
    for _ in range(n_local_enhancers):
        z_local = avgpool3s2(z_local)
    z_global = z_local
    for _ in range(n_downsample_global):
        z_global = avgpool3s2(z_global)
    w = _Walker(sd, "model_downsample", norm)
    h = w.stem7(torch.cat([pyr[-1], z_local], 1))
    for _ in range(n_downsample_global):
        h = w.down3(h)
    w = _Walker(sd, "model_resnet", norm)
    h = torch.cat([h, z_global], 1)
    for _ in range(n_blocks_global):
        h = w.resblock(h)
    w = _Walker(sd, "model_upsample", norm)
    h = torch.cat([h, z_global], 1)
    for _ in range(n_downsample_global):
        h = w.up3(h)
    out_prev = h
    for n in range(1, n_local_enhancers + 1):
        inp = pyr[n_local_enhancers - n]
        if n == n_local_enhancers:
            inp = torch.cat([inp, z], 1)
        w = _Walker(sd, "model%d_1" % n, norm)
        comb = w.down3(w.stem7(inp)) + out_prev
        if n == 1:
            comb = torch.cat([comb, z_local], 1)
        w = _Walker(sd, "model%d_2" % n, norm)
        for _ in range(n_blocks_local):
            comb = w.resblock(comb)
        out_prev = w.up3(comb)
    return _Walker(sd, "model_final", norm).head7(torch.cat([out_prev, z], 1), "tanh")


def instance_mean(feat, inst):
    """Instance-wise average pooling of Encoder.forward (models/networks.py:621-632): every pixel of an instance gets the
    mean of its instance, per sample and channel.  inst: (B, 1, H, W) holding integer ids."""
    out = feat.clone()
    B, C = feat.shape[:2]
    for b in range(B):
        ids = inst[b, 0].long()
        for i in torch.unique(ids):
            m = ids == i
            for j in range(C):
                out[b, j][m] = feat[b, j][m].mean()
    return out


def encoder(sd, x, inst, n_downsampling=4, norm="instance"):
    """Encoder.forward (models/networks.py:617-632); layers :600-615."""
    w = _Walker(sd, "model", norm)
    h = w.stem7(x)
    for _ in range(n_downsampling):
        h = w.down3(h)
    for _ in range(n_downsampling):
        h = w.up3(h)
    return instance_mean(w.head7(h, "tanh"), inst)


# --------------------------------------------------------------------------------------
# warping (models/networks.py:79-115)
# --------------------------------------------------------------------------------------
def get_grid(b, rows, cols):
    hor = torch.linspace(-1.0, 1.0, cols).view(1, 1, 1, cols).expand(b, 1, rows, cols)
    ver = torch.linspace(-1.0, 1.0, rows).view(1, 1, rows, 1).expand(b, 1, rows, cols)
    return torch.cat([hor, ver], 1)


def resample(image, flow, align_corners=False):
    """BaseNetwork.resample: grid = linspace + flow / ((w-1)/2, (h-1)/2); F.grid_sample(bilinear, border).
    align_corners=False is what the reference executes under torch >= 1.3."""
    b, c, h, w = image.shape
    grid = get_grid(b, h, w)
    flow = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
    final_grid = (grid + flow).permute(0, 2, 3, 1)
    return F.grid_sample(image, final_grid, mode="bilinear", padding_mode="border", align_corners=align_corners)


# --------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------
def composite_generator(sd, x, img_prev, mask, n_downsampling, n_blocks, use_fg, no_flow=False,
                        use_raw_only=False, norm="batch", align_corners=False):
    """CompositeGenerator.forward (models/networks.py:203-232); layer layout :128-201."""
    def tower(prefix, inp):
        w = _Walker(sd, prefix, norm)
        h = w.stem7(inp)
        for _ in range(n_downsampling):
            h = w.down3(h)
        for _ in range(n_blocks - n_blocks // 2):
            h = w.resblock(h)
        return h

    def res_up(res_prefix, up_prefix, h):
        w = _Walker(sd, res_prefix, norm)
        for _ in range(n_blocks // 2):
            h = w.resblock(h)
        w = _Walker(sd, up_prefix, norm)
        for _ in range(n_downsampling):
            h = w.up3(h)
        return h

    downsample = tower("model_down_seg", x) + tower("model_down_img", img_prev)
    img_feat = res_up("model_res_img", "model_up_img", downsample)
    img_raw = _Walker(sd, "model_final_img", norm).head7(img_feat, "tanh")
    flow = weight = flow_feat = None
    if not no_flow:
        flow_feat = res_up("model_res_flow", "model_up_flow", downsample)
        flow = _Walker(sd, "model_final_flow", norm).head7(flow_feat) * 20
        weight = _Walker(sd, "model_final_w", norm).head7(flow_feat, "sigmoid")
    if use_raw_only or no_flow:
        img_final = img_raw
    else:
        img_warp = resample(img_prev[:, -3:], flow, align_corners)
        img_final = img_raw * weight + img_warp * (1 - weight)
    img_fg_feat = None
    if use_fg:
        w = _Walker(sd, "indv_down", norm)
        h = w.stem7(x)
        for _ in range(n_downsampling):
            h = w.down3(h)
        w = _Walker(sd, "indv_res", norm)
        for _ in range(n_blocks):
            h = w.resblock(h)
        w = _Walker(sd, "indv_up", norm)
        for _ in range(n_downsampling):
            h = w.up3(h)
        img_fg_feat = h
        img_fg = _Walker(sd, "indv_final", norm).head7(img_fg_feat, "tanh")
        m = mask.expand_as(img_raw)
        img_final = img_fg * m + img_final * (1 - m)
        img_raw = img_fg * m + img_raw * (1 - m)
    return img_final, flow, weight, img_raw, img_feat, flow_feat, img_fg_feat


def composite_local_generator(sd, x, img_prev, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse,
                              n_blocks_local, scale, use_fg, no_flow=False, use_raw_only=False, norm="batch",
                              align_corners=False):
    """CompositeLocalGenerator.forward (models/networks.py:296-325); layer layout :244-294."""
    def stem_down(prefix, inp):
        w = _Walker(sd, prefix, norm)
        return w.down3(w.stem7(inp))

    def up(prefix, h):
        w = _Walker(sd, prefix, norm)
        for _ in range(n_blocks_local):
            h = w.resblock(h)
        return w.up3(h)

    flow_multiplier = 20 * (2 ** scale)
    down_img = stem_down("model_down_seg", x) + stem_down("model_down_img", img_prev)
    img_feat = up("model_up_img", down_img + img_feat_coarse)
    img_raw = _Walker(sd, "model_final_img", norm).head7(img_feat, "tanh")
    flow = weight = flow_feat = None
    if not no_flow:
        flow_feat = up("model_up_flow", down_img + flow_feat_coarse)
        flow = _Walker(sd, "model_final_flow", norm).head7(flow_feat) * flow_multiplier
        weight = _Walker(sd, "model_final_w", norm).head7(flow_feat, "sigmoid")
    if use_raw_only or no_flow:
        img_final = img_raw
    else:
        img_warp = resample(img_prev[:, -3:], flow, align_corners)
        img_final = img_raw * weight + img_warp * (1 - weight)
    img_fg_feat = None
    if use_fg:
        img_fg_feat = up("indv_up", stem_down("indv_down", x) + img_fg_feat_coarse)
        img_fg = _Walker(sd, "indv_final", norm).head7(img_fg_feat, "tanh")
        m = mask.expand_as(img_raw)
        img_final = img_fg * m + img_final * (1 - m)
        img_raw = img_fg * m + img_raw * (1 - m)
    return img_final, flow, weight, img_raw, img_feat, flow_feat, img_fg_feat


def global_generator(sd, x, n_downsampling, n_blocks, norm="instance", prefix="model", with_head=True):
    """GlobalGenerator.forward (models/networks.py:335-359)."""
    w = _Walker(sd, prefix, norm)
    h = w.stem7(x)
    for _ in range(n_downsampling):
        h = w.down3(h)
    for _ in range(n_blocks):
        h = w.resblock(h)
    for _ in range(n_downsampling):
        h = w.up3(h)
    return w.head7(h, "tanh") if with_head else h


def avgpool3s2(x):
    # nn.AvgPool2d(3, stride=2, padding=[1,1], count_include_pad=False)   models/networks.py:400,652
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


def local_enhancer(sd, x, n_downsample_global, n_blocks_global, n_local_enhancers, n_blocks_local, norm="instance"):
    """LocalEnhancer.forward (models/networks.py:402-419)."""
    pyr = [x]
    for _ in range(n_local_enhancers):
        pyr.append(avgpool3s2(pyr[-1]))
    out = global_generator(sd, pyr[-1], n_downsample_global, n_blocks_global, norm, "model", with_head=False)
    for n in range(1, n_local_enhancers + 1):
        w = _Walker(sd, "model%d_1" % n, norm)
        h = w.down3(w.stem7(pyr[n_local_enhancers - n])) + out
        w = _Walker(sd, "model%d_2" % n, norm)
        for _ in range(n_blocks_local):
            h = w.resblock(h)
        h = w.up3(h)
        out = w.head7(h, "tanh") if n == n_local_enhancers else h
    return out


# --------------------------------------------------------------------------------------
# discriminators (models/networks.py:634-725)
# --------------------------------------------------------------------------------------
def nlayer_discriminator_feats(sd, prefix_fmt, x, n_layers, norm="batch"):
    """One PatchGAN with intermediate features; prefix_fmt % j names group j ('scale0_layer%d')."""
    feats = []
    h = F.leaky_relu(_conv(sd, (prefix_fmt % 0) + ".0", x, 2, 2), 0.2)
    feats.append(h)
    for j in range(1, n_layers):
        k = prefix_fmt % j
        h = F.leaky_relu(_norm(sd, k + ".1", _conv(sd, k + ".0", h, 2, 2), norm), 0.2)
        feats.append(h)
    k = prefix_fmt % n_layers
    h = F.leaky_relu(_norm(sd, k + ".1", _conv(sd, k + ".0", h, 1, 2), norm), 0.2)
    feats.append(h)
    h = _conv(sd, (prefix_fmt % (n_layers + 1)) + ".0", h, 1, 2)
    feats.append(h)
    return feats


def multiscale_discriminator(sd, x, n_layers, num_D, norm="batch"):
    """MultiscaleDiscriminator.forward with getIntermFeat=True (models/networks.py:663-675)."""
    result = []
    cur = x
    for i in range(num_D):
        result.append(nlayer_discriminator_feats(sd, "scale%d_layer%%d" % (num_D - 1 - i), cur, n_layers, norm))
        if i != num_D - 1:
            cur = avgpool3s2(cur)
    return result


# --------------------------------------------------------------------------------------
# model-level input encoding and inference (models/vid2vid_model_G.py, models/base_model.py)
# --------------------------------------------------------------------------------------
def get_edges(t):
    """models/base_model.py:146-152 -- 4-neighbour instance boundaries."""
    edge = torch.zeros_like(t, dtype=torch.bool)
    edge[..., :, 1:] |= t[..., :, 1:] != t[..., :, :-1]
    edge[..., :, :-1] |= t[..., :, 1:] != t[..., :, :-1]
    edge[..., 1:, :] |= t[..., 1:, :] != t[..., :-1, :]
    edge[..., :-1, :] |= t[..., 1:, :] != t[..., :-1, :]
    return edge.float()


def encode_input(input_map, inst_map, label_nc):
    """models/vid2vid_model_G.py:86-112.  input_map (B,T,1,H,W) floats holding ints -> (B,T,label_nc(+1),H,W)."""
    if label_nc != 0:
        b, t, _, h, w = input_map.shape
        onehot = torch.zeros(b, t, label_nc, h, w)
        input_map = onehot.scatter_(2, input_map.long(), 1.0)
    if inst_map is not None:
        input_map = torch.cat([input_map, get_edges(inst_map)], dim=2)
    return input_map


def build_pyr(t, n_scales):
    """models/base_model.py:122-134 on (B,T,C,H,W)."""
    pyr = [t]
    for _ in range(1, n_scales):
        b, tt, c, h, w = pyr[-1].shape
        pyr.append(avgpool3s2(pyr[-1].reshape(-1, 1, h, w)).reshape(b, tt, c, h // 2, w // 2))
    return pyr


def compute_mask(real_As, ts, fg_labels):
    """models/vid2vid_model_G.py:322-330."""
    m = real_As[:, ts:ts + 1, fg_labels[0]].clone()
    for l in fg_labels[1:]:
        m = m + real_As[:, ts:ts + 1, l]
    return torch.clamp(m, 0, 1)


class InferenceOracle:
    """Vid2VidModelG.inference (models/vid2vid_model_G.py:198-229) with use_real_img / no_first_img
    first frames (:231-251), for label2city-style integer label input or raw multi-channel input."""

    def __init__(self, sds, label_nc, use_instance, fg, fg_labels, n_downsample_G, n_blocks, n_blocks_local,
                 tG=3, no_first_img=False, align_corners=False):
        self.sds, self.S = sds, len(sds)
        self.label_nc, self.use_instance, self.fg, self.fg_labels = label_nc, use_instance, fg, list(fg_labels)
        self.n_down, self.n_blocks, self.n_blocks_local, self.tG = n_downsample_G, n_blocks, n_blocks_local, tG
        self.no_first_img, self.align_corners = no_first_img, align_corners
        self.fake_B_prev = None
        self.last = {}

    def step(self, input_A, input_B, inst_A):
        tG, S = self.tG, self.S
        with torch.no_grad():
            real_A = encode_input(input_A, inst_A if self.use_instance else None, self.label_nc)
            first = self.fake_B_prev is None
            if first:
                if self.no_first_img:
                    b, _, _, h, w = real_A.shape
                    fb = torch.zeros(b, tG - 1, 3, h, w)
                else:
                    fb = input_B[:, :tG - 1]
                self.fake_B_prev = [p[0] for p in build_pyr(fb, S)]
            pyr = build_pyr(real_A, S)
            feat = flow_feat = fg_feat = None
            fake_B = None
            for s in range(S):
                si = S - 1 - s
                rA = pyr[si]
                _, _, _, h, w = rA.shape
                x = rA[0, :tG].reshape(1, -1, h, w)
                prev = self.fake_B_prev[si].reshape(1, -1, h, w)
                mask = compute_mask(rA, tG - 1, self.fg_labels)[0].unsqueeze(0) if self.fg else None
                if mask is not None:
                    mask = mask.reshape(1, 1, h, w)
                raw_only = self.no_first_img and first
                if s == 0:
                    out = composite_generator(self.sds[0], x, prev, mask, self.n_down, self.n_blocks, self.fg,
                                              use_raw_only=raw_only, align_corners=self.align_corners)
                else:
                    out = composite_local_generator(self.sds[s], x, prev, mask, feat, flow_feat, fg_feat,
                                                    self.n_blocks_local, s, self.fg, use_raw_only=raw_only,
                                                    align_corners=self.align_corners)
                fake_B, flow_s, weight_s, raw_s, feat, flow_feat, fg_feat = out
                # per-scale heads of the newest frame (si = 0 finest), for the full-width parity checks
                self.last["flow%d" % si], self.last["weight%d" % si], self.last["raw%d" % si] = flow_s, weight_s, raw_s
                self.fake_B_prev[si] = torch.cat([self.fake_B_prev[si][1:], fake_B])
            return fake_B, pyr[0][0, -1]


# --------------------------------------------------------------------------------------
# FlowNet2 native ops, restated from the CUDA sources
# --------------------------------------------------------------------------------------
def correlation(in1, in2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda_kernel.cu:73-147 (forward) with the output-size rule of
    correlation_cuda.cc:25-38.  in1, in2: (N,C,H,W) -> (N, D*D, OH, OW)."""
    n, c, h, w = in1.shape
    krad = (kernel_size - 1) // 2
    border = krad + max_displacement
    ph, pw = h + 2 * pad_size, w + 2 * pad_size
    oh = int(math.ceil(float(ph - 2 * border) / float(stride1)))
    ow = int(math.ceil(float(pw - 2 * border) / float(stride1)))
    drad = max_displacement // stride2
    d = 2 * drad + 1
    p1 = F.pad(in1, (pad_size,) * 4)      # channels_first(): zero-padded copies (:46-70)
    p2 = F.pad(in2, (pad_size,) * 4)
    out = torch.zeros(n, d * d, oh, ow)
    nelems = kernel_size * kernel_size * c
    ys = torch.arange(oh) * stride1 + max_displacement
    xs = torch.arange(ow) * stride1 + max_displacement
    for tj in range(-drad, drad + 1):
        for ti in range(-drad, drad + 1):
            acc = torch.zeros(n, oh, ow)
            for j in range(-krad, krad + 1):
                for i in range(-krad, krad + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    y2, x2 = ys + tj * stride2 + j, xs + ti * stride2 + i
                    ok = ((y2 >= 0) & (y2 < ph))[:, None] & ((x2 >= 0) & (x2 < pw))[None, :]
                    b = p2[:, :, y2.clamp(0, ph - 1)[:, None], x2.clamp(0, pw - 1)[None, :]] * ok
                    acc += (a * b).sum(1)
            out[:, (tj + drad) * d + (ti + drad)] = acc / nelems
    return out


def resample2d(img, flow, kernel_size=1):
    """resample2d_kernel.cu:15-64: out[b,c,y,x] = bilinear(img[b,c], x+fx, y+fy); indices clamped to the
    output extent, weights from the unclamped fractional part."""
    b, c, _, _ = img.shape
    _, _, oh, ow = flow.shape
    ys, xs = torch.meshgrid(torch.arange(oh, dtype=torch.float32), torch.arange(ow, dtype=torch.float32),
                            indexing="ij")
    xf, yf = xs[None] + flow[:, 0], ys[None] + flow[:, 1]
    alpha, beta = xf - torch.floor(xf), yf - torch.floor(yf)
    xL = torch.floor(xf).long().clamp(0, ow - 1); xR = (torch.floor(xf).long() + 1).clamp(0, ow - 1)
    yT = torch.floor(yf).long().clamp(0, oh - 1); yB = (torch.floor(yf).long() + 1).clamp(0, oh - 1)
    out = torch.zeros(b, c, oh, ow)
    bi = torch.arange(b)[:, None, None]
    for ch in range(c):
        im = img[:, ch]
        val = torch.zeros(b, oh, ow)
        for fy in range(kernel_size):
            for fx in range(kernel_size):
                val = val + (1. - alpha) * (1. - beta) * im[bi, yT + fy, xL + fx]
                val = val + alpha * (1. - beta) * im[bi, yT + fy, xR + fx]
                val = val + (1. - alpha) * beta * im[bi, yB + fy, xL + fx]
                val = val + alpha * beta * im[bi, yB + fy, xR + fx]
        out[:, ch] = val
    return out


def channelnorm(x):
    """channelnorm_kernel.cu:18-60: out[b,0,y,x] = sqrt(sum_c x[b,c,y,x]^2)."""
    return (x * x).sum(1, keepdim=True).sqrt()


# --------------------------------------------------------------------------------------
# FlowNet2 (models/flownet2_pytorch/models.py:96-161 and networks/*.py), batchNorm=False
# --------------------------------------------------------------------------------------
def _fconv(sd, key, x, stride=1):
    """submodules.conv (:7-19): Conv2d(pad (k-1)//2) + LeakyReLU(0.1); parameters at <key>.0.*"""
    w = sd[key + ".0.weight"]
    return F.leaky_relu(F.conv2d(x, w, sd[key + ".0.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2), 0.1)


def _fdeconv(sd, key, x):
    """submodules.deconv (:36-40): ConvTranspose2d(4, 2, 1) + LeakyReLU(0.1)"""
    return F.leaky_relu(F.conv_transpose2d(x, sd[key + ".0.weight"], sd[key + ".0.bias"], stride=2, padding=1), 0.1)


def _fpredict(sd, key, x):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], padding=1)          # submodules.predict_flow (:33-34)


def _fup(sd, key, x):
    return F.conv_transpose2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=2, padding=1)


def _fdecode(sd, p, c6, skips, with_inter):
    """The predict_flow / upsampled_flow / deconv / cat ladder shared by FlowNetC (:111-129), FlowNetS (:69-87)
    and FlowNetSD (:79-100, with inter_conv)."""
    feat = c6
    flow = _fpredict(sd, p + "predict_flow6", c6)
    for lvl, skip in zip((5, 4, 3, 2), skips):
        flow_up = _fup(sd, p + "upsampled_flow%d_to_%d" % (lvl + 1, lvl), flow)
        feat = torch.cat((skip, _fdeconv(sd, p + "deconv%d" % lvl, feat), flow_up), 1)
        head_in = feat
        if with_inter:
            k = p + "inter_conv%d.0" % lvl
            head_in = F.conv2d(feat, sd[k + ".weight"], sd[k + ".bias"], padding=1)
        flow = _fpredict(sd, p + "predict_flow%d" % lvl, head_in)
    return flow


def flownet_c(sd, x, p="flownetc."):
    """FlowNetC.forward (networks/FlowNetC.py:71-131)"""
    def stream(im):
        c1 = _fconv(sd, p + "conv1", im, 2)
        c2 = _fconv(sd, p + "conv2", c1, 2)
        return c2, _fconv(sd, p + "conv3", c2, 2)
    c2a, c3a = stream(x[:, 0:3])
    _, c3b = stream(x[:, 3:])
    corr = F.leaky_relu(correlation(c3a, c3b, 20, 1, 20, 1, 2), 0.1)
    c3_1 = _fconv(sd, p + "conv3_1", torch.cat((_fconv(sd, p + "conv_redir", c3a), corr), 1))
    c4 = _fconv(sd, p + "conv4_1", _fconv(sd, p + "conv4", c3_1, 2))
    c5 = _fconv(sd, p + "conv5_1", _fconv(sd, p + "conv5", c4, 2))
    c6 = _fconv(sd, p + "conv6_1", _fconv(sd, p + "conv6", c5, 2))
    return _fdecode(sd, p, c6, (c5, c4, c3_1, c2a), False)


def flownet_s(sd, x, p):
    """FlowNetS.forward (networks/FlowNetS.py:60-93)"""
    c2 = _fconv(sd, p + "conv2", _fconv(sd, p + "conv1", x, 2), 2)
    c3 = _fconv(sd, p + "conv3_1", _fconv(sd, p + "conv3", c2, 2))
    c4 = _fconv(sd, p + "conv4_1", _fconv(sd, p + "conv4", c3, 2))
    c5 = _fconv(sd, p + "conv5_1", _fconv(sd, p + "conv5", c4, 2))
    c6 = _fconv(sd, p + "conv6_1", _fconv(sd, p + "conv6", c5, 2))
    return _fdecode(sd, p, c6, (c5, c4, c3, c2), False)


def flownet_sd(sd, x, p="flownets_d."):
    """FlowNetSD.forward (networks/FlowNetSD.py:67-107)"""
    c0 = _fconv(sd, p + "conv0", x)
    c1 = _fconv(sd, p + "conv1_1", _fconv(sd, p + "conv1", c0, 2))
    c2 = _fconv(sd, p + "conv2_1", _fconv(sd, p + "conv2", c1, 2))
    c3 = _fconv(sd, p + "conv3_1", _fconv(sd, p + "conv3", c2, 2))
    c4 = _fconv(sd, p + "conv4_1", _fconv(sd, p + "conv4", c3, 2))
    c5 = _fconv(sd, p + "conv5_1", _fconv(sd, p + "conv5", c4, 2))
    c6 = _fconv(sd, p + "conv6_1", _fconv(sd, p + "conv6", c5, 2))
    return _fdecode(sd, p, c6, (c5, c4, c3, c2), True)


def flownet_fusion(sd, x, p="flownetfusion."):
    """FlowNetFusion.forward (networks/FlowNetFusion.py:48-69)"""
    def inter(key, t):
        return F.conv2d(t, sd[p + key + ".0.weight"], sd[p + key + ".0.bias"], padding=1)
    c0 = _fconv(sd, p + "conv0", x)
    c1 = _fconv(sd, p + "conv1_1", _fconv(sd, p + "conv1", c0, 2))
    c2 = _fconv(sd, p + "conv2_1", _fconv(sd, p + "conv2", c1, 2))
    flow2 = _fpredict(sd, p + "predict_flow2", c2)
    cat1 = torch.cat((c1, _fdeconv(sd, p + "deconv1", c2), _fup(sd, p + "upsampled_flow2_to_1", flow2)), 1)
    flow1 = _fpredict(sd, p + "predict_flow1", inter("inter_conv1", cat1))
    cat0 = torch.cat((c0, _fdeconv(sd, p + "deconv0", cat1), _fup(sd, p + "upsampled_flow1_to_0", flow1)), 1)
    return _fpredict(sd, p + "predict_flow0", inter("inter_conv0", cat0))


def flownet2(sd, im1, im2, div_flow=20.0, rgb_max=1.0):
    """FlowNet2.forward (models/flownet2_pytorch/models.py:96-161) on im1, im2: (B,3,H,W)."""
    inputs = torch.stack((im1, im2), dim=2)                                  # (B,3,2,H,W), flownet.py:53
    rgb_mean = inputs.reshape(inputs.shape[0], 3, -1).mean(-1).view(-1, 3, 1, 1, 1)
    xx = (inputs - rgb_mean) / rgb_max
    x = torch.cat((xx[:, :, 0], xx[:, :, 1]), 1)
    x1, x2 = x[:, :3], x[:, 3:]
    up_bil = lambda t: F.interpolate(t, scale_factor=4, mode="bilinear", align_corners=False)
    up_near = lambda t: F.interpolate(t, scale_factor=4, mode="nearest")
    flow = up_bil(flownet_c(sd, x) * div_flow)
    for p, up in (("flownets_1.", up_bil), ("flownets_2.", up_near)):
        warped = resample2d(x2.contiguous(), flow)
        nrm = channelnorm(x1 - warped)
        flow = up(flownet_s(sd, torch.cat((x, warped, flow / div_flow, nrm), 1), p) * div_flow)
    flow_s2 = flow
    norm_s2 = channelnorm(flow_s2)
    diff_s2 = channelnorm(x1 - resample2d(x2.contiguous(), flow_s2))
    flow_sd = up_near(flownet_sd(sd, x) / div_flow)
    norm_sd = channelnorm(flow_sd)
    diff_sd = channelnorm(x1 - resample2d(x2.contiguous(), flow_sd))
    cat3 = torch.cat((x1, flow_sd, flow_s2, norm_sd, norm_s2, diff_sd, diff_s2), 1)
    return flownet_fusion(sd, cat3)


def flow_and_conf(sd, im1, im2):
    """FlowNet.compute_flow_and_conf (models/flownet.py:43-59) incl. the resize to multiples of 64."""
    old_h, old_w = im1.shape[2], im1.shape[3]
    new_h, new_w = old_h // 64 * 64, old_w // 64 * 64
    if old_h != new_h:
        im1 = F.interpolate(im1, size=(new_h, new_w), mode="bilinear", align_corners=False)
        im2 = F.interpolate(im2, size=(new_h, new_w), mode="bilinear", align_corners=False)
    flow = flownet2(sd, im1, im2)
    d = im1 - resample2d(im2.contiguous(), flow)
    conf = ((d * d).sum(1, keepdim=True) < 0.02).float()
    if old_h != new_h:
        flow = F.interpolate(flow, size=(old_h, old_w), mode="bilinear", align_corners=False) * old_h / new_h
        conf = F.interpolate(conf, size=(old_h, old_w), mode="bilinear", align_corners=False)
    return flow, conf


# --------------------------------------------------------------------------------------
# training step: Vid2VidModelG.forward, Vid2VidModelD.forward, losses (differentiable: plain torch ops,
# so torch.autograd of this restatement is the gradient oracle)
# --------------------------------------------------------------------------------------
def gan_loss(preds, target_is_real):
    """GANLoss.__call__ for multiscale outputs (models/networks.py:764-771): sum over scales of
    MSELoss(pred[-1], 1 or 0)."""
    loss = 0
    for feats in preds:
        p = feats[-1]
        loss = loss + F.mse_loss(p, torch.full_like(p, 1.0 if target_is_real else 0.0))
    return loss


def masked_l1(a, b, mask):
    """MaskedL1Loss (models/networks.py:804-812)"""
    m = mask.expand(-1, a.shape[1], -1, -1)
    return F.l1_loss(a * m, b * m)


def gan_and_fm_loss(pred_real, pred_fake, n_layers_D, num_D, lambda_feat):
    """Vid2VidModelD.GAN_and_FM_loss (models/vid2vid_model_D.py:199-213)"""
    loss_gan = gan_loss(pred_fake, True)
    loss_fm = torch.zeros(())
    fw, dw = 4.0 / (n_layers_D + 1), 1.0 / num_D
    for i in range(min(len(pred_fake), num_D)):
        for j in range(len(pred_fake[i]) - 1):
            loss_fm = loss_fm + dw * fw * F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat
    return loss_gan, loss_fm


def compute_loss_D(sdD, x_real, x_fake, n_layers_D, num_D, lambda_feat):
    """compute_loss_D / compute_loss_D_T (models/vid2vid_model_D.py:168-197) on already concatenated inputs:
    three discriminator passes (real, fake detached, fake)."""
    pred_real = multiscale_discriminator(sdD, x_real, n_layers_D, num_D)
    pred_fake = multiscale_discriminator(sdD, x_fake.detach(), n_layers_D, num_D)
    l_real, l_fake = gan_loss(pred_real, True), gan_loss(pred_fake, False)
    pred_fake = multiscale_discriminator(sdD, x_fake, n_layers_D, num_D)
    l_gan, l_fm = gan_and_fm_loss(pred_real, pred_fake, n_layers_D, num_D, lambda_feat)
    return l_real, l_fake, l_gan, l_fm


# torchvision.models.vgg19().features, configuration 'E' (Simonyan & Zisserman 2014; torchvision is an external
# dependency of the reference and absent here -- any version: the architecture has not changed): every conv is
# Conv2d(3x3, padding 1) + ReLU, 'M' = MaxPool2d(2, 2).  Module indices: 0,2 | 4=M 5,7 | 9=M 10,12,14,16 | 18=M
# 19,21,23,25 | 27=M 28.  Vgg19 (models/networks.py:840-870) cuts it into slices [0,2) [2,7) [7,12) [12,21) [21,30)
# whose parameters are named slice<k>.<index>.*.
_VGG_SLICES = [((0,), False), ((2, 5), True), ((7, 10), True), ((12, 14, 16, 19), True), ((21, 23, 25, 28), True)]
_VGG_POOL_BEFORE = {5: True, 10: True, 19: True, 28: True}


def vgg19_slices(sd, x):
    """Vgg19.forward (models/networks.py:862-870): [h_relu1 .. h_relu5]."""
    outs = []
    for k, (idxs, _) in enumerate(_VGG_SLICES):
        for idx in idxs:
            if _VGG_POOL_BEFORE.get(idx):
                x = F.max_pool2d(x, 2, 2)
            x = F.relu(_conv(sd, "slice%d.%d" % (k + 1, idx), x, 1, 1))
        outs.append(x)
    return outs


def vgg_loss(sd, x, y):
    """VGGLoss.forward (models/networks.py:784-791): both inputs AvgPool2d(2,2)-ed while wider than 1024,
    sum_i w_i * L1(vgg_i(x), vgg_i(y).detach())."""
    while x.size(3) > 1024:
        x, y = F.avg_pool2d(x, 2, 2, count_include_pad=False), F.avg_pool2d(y, 2, 2, count_include_pad=False)
    xs, ys = vgg19_slices(sd, x), vgg19_slices(sd, y)
    loss = 0
    for w, a, b in zip([1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0], xs, ys):
        loss = loss + w * F.l1_loss(a, b.detach())
    return loss


def model_D_image_losses(sdD, t, n_layers_D=3, num_D=2, lambda_feat=10.0, lambda_F=10.0, lambda_T=10.0, n_scales_spatial=1,
                         sd_vgg=None):
    """Vid2VidModelD.forward(scale_T=0, ...) (models/vid2vid_model_D.py:110-166); --no_vgg when sd_vgg is None.
    t: dict of (n, ch, H, W) tensors real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight,
    flow_ref, conf_ref.  Returns the dict keyed by loss_names."""
    out = {"G_VGG": torch.zeros(()), "W": torch.zeros(())}
    if sd_vgg is not None:                                       # :136, :143-144
        out["G_VGG"] = (vgg_loss(sd_vgg, t["fake_B"], t["real_B"]) + vgg_loss(sd_vgg, t["fake_B_raw"], t["real_B"])) * lambda_feat
    out["F_Flow"] = masked_l1(t["flow"], t["flow_ref"], t["conf_ref"]) * lambda_F / (2 ** (n_scales_spatial - 1))
    real_B_warp = resample(t["real_B_prev"], t["flow"])
    out["F_Warp"] = masked_l1(real_B_warp, t["real_B"], t["conf_ref"]) * lambda_T
    real_AB = torch.cat((t["real_A"], t["real_B"]), 1)
    l_real, l_fake, l_gan, l_fm = compute_loss_D(sdD, real_AB, torch.cat((t["real_A"], t["fake_B"]), 1),
                                                 n_layers_D, num_D, lambda_feat)
    warp_ref = resample(t["fake_B_prev"], t["flow_ref"]).detach()
    out["G_Warp"] = masked_l1(t["fake_B"], warp_ref, t["conf_ref"]) * lambda_T
    r2, f2, g2, m2 = compute_loss_D(sdD, real_AB, torch.cat((t["real_A"], t["fake_B_raw"]), 1), n_layers_D, num_D, lambda_feat)
    out.update({"D_real": l_real + r2, "D_fake": l_fake + f2, "G_GAN": l_gan + g2, "G_GAN_Feat": l_fm + m2})
    return out


def model_D_temporal_losses(sdDT, real_B, fake_B, flow_ref, tD=3, n_layers_D=3, num_D=2, lambda_feat=10.0):
    """Vid2VidModelD.forward(scale_T>0, ...) (:104-112) -> compute_loss_D_T (:181-197); inputs (n, tD, C, H, W)."""
    n, _, _, H, W = real_B.shape
    fr = (flow_ref / 20).reshape(n, -1, H, W)
    l_real, l_fake, l_gan, l_fm = compute_loss_D(sdDT, torch.cat((real_B.reshape(n, -1, H, W), fr), 1),
                                                 torch.cat((fake_B.reshape(n, -1, H, W), fr), 1), n_layers_D, num_D, lambda_feat)
    return {"G_T_GAN": l_gan, "G_T_GAN_Feat": l_fm, "D_T_real": l_real, "D_T_fake": l_fake, "G_T_Warp": torch.zeros(())}


def generate_frames_train(sds, real_A_all, real_B_all, fg, fg_labels, n_down, n_blocks, n_blocks_local, n_frames_load, tG=3,
                          return_pyramid=False):
    """Vid2VidModelG.forward / generate_frame_train (models/vid2vid_model_G.py:114-196) for batch 1, first chunk
    (real first frames), n_frames_bp = 1, finetune_all.  real_A_all (1,T,C,H,W) encoded labels, real_B_all (1,T,3,H,W).
    Returns fake_B, fake_B_raw, flow, weight: (1, n_frames_load, ., H, W); with return_pyramid additionally the
    fake_B_pyr of :139-196 -- per scale (finest first) the tG-1 given frames followed by the generated ones, detached."""
    S = len(sds)
    A_pyr = build_pyr(real_A_all, S)
    B_pyr = build_pyr(real_B_all[:, :tG - 1], S)
    fakes = [p for p in B_pyr]
    raws, flows, weights = [], [], []
    for t in range(n_frames_load):
        feat = flow_feat = fg_feat = None
        for s in range(S):
            si = S - 1 - s
            rA = A_pyr[si]
            _, _, _, h, w = rA.shape
            x = rA[:, t:t + tG].reshape(1, -1, h, w)
            prev = fakes[si][:, t:t + tG - 1].detach().reshape(1, -1, h, w)       # n_frames_bp = 1 (:167-168)
            mask = compute_mask(rA, t + tG - 1, fg_labels)[0].reshape(1, 1, h, w) if fg else None
            if s == 0:
                out = composite_generator(sds[0], x, prev, mask, n_down, n_blocks, fg)
            else:
                out = composite_local_generator(sds[s], x, prev, mask, feat, flow_feat, fg_feat, n_blocks_local, s, fg)
            fake_B, flow, weight, raw, feat, flow_feat, fg_feat = out
            fakes[si] = torch.cat([fakes[si], fake_B.unsqueeze(1)], 1)
            if s == S - 1:
                raws.append(raw.unsqueeze(1)); flows.append(flow.unsqueeze(1)); weights.append(weight.unsqueeze(1))
    res = fakes[0][:, tG - 1:], torch.cat(raws, 1), torch.cat(flows, 1), torch.cat(weights, 1)
    if return_pyramid:
        return res + ([f.detach() for f in fakes],)
    return res


# --------------------------------------------------------------------------------------
# visualisation: tensor2flow (util/util.py:89-107)
# --------------------------------------------------------------------------------------
def tensor2flow(output):
    """util/util.py:89-107 restated in numpy.  The reference calls OpenCV (opencv-python, unpinned in docker/Dockerfile and
    absent from /root/reference and from this image: PARITY UNPINNED for this helper), so its three calls are restated
    from OpenCV's published definitions:
      cv2.cartToPolar(x, y)            -> mag = sqrt(x^2 + y^2), ang = atan2(y, x) in [0, 2 pi)  (OpenCV's fastAtan2 is a
                                          polynomial with ~0.3 degree error; the exact angle is used here, so a hue level may
                                          differ by one from a run of the reference with a given OpenCV build)
      cv2.normalize(mag, None, 0, 255, NORM_MINMAX) -> (mag - min) * 255 / (max - min)   (0 for a constant image)
      cv2.cvtColor(hsv, COLOR_HSV2RGB) for uint8    -> H in half degrees, S = V = x / 255, the six-sector (v, p, q, t)
                                          table, * 255 rounded to nearest
    numpy stores of floats into the uint8 `hsv` array truncate.  Returns uint8 (H, W, 3)."""
    import numpy as np
    t = output
    if t.dim() == 5:
        t = t[0, -1]
    if t.dim() == 4:
        t = t[0]
    f = np.transpose(t.detach().cpu().float().numpy(), (1, 2, 0))
    fx, fy = f[..., 0], f[..., 1]
    mag = np.sqrt(fx * fx + fy * fy).astype(np.float32)
    ang = np.arctan2(fy, fx).astype(np.float32)
    ang = np.where(ang < 0, ang + np.float32(2 * np.pi), ang).astype(np.float32)
    H8 = (ang * np.float32(180.0) / np.float32(np.pi) / np.float32(2.0)).astype(np.int32) & 255
    mn, mx = mag.min(), mag.max()
    scale = np.float32(255.0) / (mx - mn) if (mx - mn) > 1.1920929e-07 else np.float32(0.0)
    V8 = ((mag - mn) * scale).astype(np.int32) & 255
    v = V8.astype(np.float32) * np.float32(1.0 / 255.0)
    h = H8.astype(np.float32) * np.float32(6.0 / 180.0)
    sector = np.floor(h).astype(np.int32)
    fr = h - sector.astype(np.float32)
    sector = sector % 6
    tab = np.stack([v, np.zeros_like(v), v * (np.float32(1.0) - fr), v * fr], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])          # (b, g, r)
    idx = sd[sector]
    b = np.take_along_axis(tab, idx[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, idx[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, idx[..., 2:3], -1)[..., 0]
    rgb = np.stack([r, g, b], -1) * np.float32(255.0)
    return np.rint(np.clip(rgb, 0, 255)).astype(np.uint8)
