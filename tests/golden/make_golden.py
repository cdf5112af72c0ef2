#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by executing the REFERENCE implementation
(/root/reference, read-only) on CPU.  Run in the build container only -- the GPU box has no
/root/reference; the tests read the committed .npz files.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Shims follow SURVEY.md Appendix B: stub modules for absent third-party imports, Tensor.cuda ->
identity, torch.cuda.FloatTensor/ByteTensor -> CPU types.  No reference file is modified and no
reference source is copied: the script only imports and calls it.

Each fixture stores: the (small) network's full state_dict, the seeded inputs and the reference
outputs.  Weights are stored rather than re-derived from a seed so that the fixtures do not
depend on the host's vectorised RNG paths.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("V2V_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    sys.path.insert(0, REF)
    for m in ["torchvision", "torchvision.models", "cv2", "dominate", "dominate.tags", "scipy.misc"]:
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.ByteTensor = torch.ByteTensor
    torch.Tensor.get_device = lambda self: 0


def opt_ns(**kw):
    d = dict(fp16=False, n_blocks=2, n_blocks_local=1, n_local_enhancers=1, fg=True, no_flow=False,
             feat_num=3)
    d.update(kw)
    return types.SimpleNamespace(**d)


def synth_labels(gen, T, H, W, n_labels, cell=8, fg_label=26):
    """Blocky label / instance maps, floats holding integers."""
    lo = torch.randint(0, n_labels, (T, H // cell, W // cell), generator=gen)
    lo[:, 0, :2] = fg_label % n_labels
    lab = lo.repeat_interleave(cell, 1).repeat_interleave(cell, 2)
    for t in range(T):
        lab[t] = torch.roll(lab[t], shifts=2 * t, dims=1)
    return lab.float()


def sd_to_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def golden_composite(networks):
    """CompositeGenerator.forward (models/networks.py:203-232), fg tower, n_downsampling=3."""
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(12)
    label_nc, tG, H, W, ngf = 35, 3, 32, 64, 8
    opt = opt_ns(n_blocks=2)
    net = networks.define_G(tG * (label_nc + 1), 3, (tG - 1) * 3, ngf, "composite", 3, "batch", 0, [], opt)
    x = torch.zeros(1, tG * (label_nc + 1), H, W)
    lab = synth_labels(gen, tG, H, W, label_nc)
    inst = synth_labels(gen, tG, H, W, 7)
    for t in range(tG):
        x[0, t * 36:(t * 36 + 35)].scatter_(0, lab[t].long().unsqueeze(0), 1.0)
        e = torch.zeros(H, W, dtype=torch.bool)
        e[:, 1:] |= inst[t][:, 1:] != inst[t][:, :-1]; e[:, :-1] |= inst[t][:, 1:] != inst[t][:, :-1]
        e[1:, :] |= inst[t][1:, :] != inst[t][:-1, :]; e[:-1, :] |= inst[t][1:, :] != inst[t][:-1, :]
        x[0, t * 36 + 35] = e.float()
    prev = torch.tanh(torch.nn.functional.interpolate(torch.randn(1, 6, H // 8, W // 8, generator=gen),
                                                      scale_factor=8, mode="bilinear", align_corners=False))
    mask = (lab[tG - 1] == 26).float().view(1, 1, H, W)
    # keep flows O(few px) so that the warp is a meaningful check
    with torch.no_grad():
        net.model_final_flow[1].weight.mul_(0.1)
        outs = net.forward(x, prev, mask, None, None, None, False)
        outs_raw = net.forward(x, prev, mask, None, None, None, True)
    names = ["img_final", "flow", "weight", "img_raw", "img_feat", "flow_feat", "img_fg_feat"]
    arrays = sd_to_np(net.state_dict(), "sd.")
    arrays.update({"in.labels": lab.numpy(), "in.inst": inst.numpy(), "in.x": x.numpy(), "in.prev": prev.numpy(),
                   "in.mask": mask.numpy()})
    arrays.update({"out." + n: o.numpy() for n, o in zip(names, outs)})
    arrays["out.img_final_rawonly"] = outs_raw[0].numpy()
    arrays["cfg"] = np.array([label_nc, tG, H, W, ngf, 3, 2, 1], dtype=np.int64)  # ..., n_down, n_blocks, fg
    save("composite_fg_32x64", **arrays)


def golden_composite_local(networks):
    """CompositeGenerator (scale 0) -> CompositeLocalGenerator (scale 1), no fg (models/networks.py:296-325)."""
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(22)
    in_nc, H, W, ngf = 12, 32, 64, 8
    opt = opt_ns(n_blocks=2, n_blocks_local=1, fg=False)
    g0 = networks.define_G(in_nc, 3, 6, ngf, "composite", 2, "batch", 0, [], opt)
    g1 = networks.define_G(in_nc, 3, 6, ngf // 2, "compositeLocal", 2, "batch", 1, [], opt)
    x1 = torch.rand(1, in_nc, H, W, generator=gen)
    p1 = torch.tanh(torch.randn(1, 6, H, W, generator=gen))
    pool = torch.nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
    x0, p0 = pool(x1), pool(p1)
    with torch.no_grad():
        g0.model_final_flow[1].weight.mul_(0.1)
        g1.model_final_flow[1].weight.mul_(0.1)
        o0 = g0.forward(x0, p0, None, None, None, None, False)
        o1 = g1.forward(x1, p1, None, o0[4], o0[5], o0[6], False)
    arrays = sd_to_np(g0.state_dict(), "sd0.")
    arrays.update(sd_to_np(g1.state_dict(), "sd1."))
    arrays.update({"in.x1": x1.numpy(), "in.p1": p1.numpy(), "in.x0": x0.numpy(), "in.p0": p0.numpy()})
    for i, n in enumerate(["img_final", "flow", "weight", "img_raw"]):
        arrays["out0." + n] = o0[i].numpy()
        arrays["out1." + n] = o1[i].numpy()
    arrays["cfg"] = np.array([in_nc, H, W, ngf, 2, 2, 1], dtype=np.int64)
    save("composite_local_32x64", **arrays)


def golden_discriminator(networks):
    """MultiscaleDiscriminator.forward with getIntermFeat (models/networks.py:663-675)."""
    torch.manual_seed(31)
    gen = torch.Generator().manual_seed(32)
    in_nc, H, W, ndf = 13, 64, 96, 8
    net = networks.define_D(in_nc, ndf, 3, "batch", 2, True, [])
    x = torch.randn(1, in_nc, H, W, generator=gen)
    with torch.no_grad():
        res = net.forward(x)
    arrays = sd_to_np(net.state_dict(), "sd.")
    arrays["in.x"] = x.numpy()
    for i, feats in enumerate(res):
        for j, f in enumerate(feats):
            arrays["out.%d.%d" % (i, j)] = f.numpy()
    arrays["cfg"] = np.array([in_nc, H, W, ndf, 3, 2], dtype=np.int64)
    save("multiscale_d_64x96", **arrays)


def golden_global(networks):
    """GlobalGenerator / LocalEnhancer with InstanceNorm (first-frame nets, models/networks.py:327-419)."""
    torch.manual_seed(41)
    gen = torch.Generator().manual_seed(42)
    opt = opt_ns(n_blocks=2, n_blocks_local=1)
    g = networks.define_G(11, 3, 0, 8, "global", 2, "instance", 0, [], opt)
    le = networks.define_G(11, 3, 0, 4, "local", 2, "instance", 0, [], opt)
    x = torch.rand(1, 11, 32, 64, generator=gen)
    with torch.no_grad():
        og = g.forward(x)
        ol = le.forward(x)
    arrays = sd_to_np(g.state_dict(), "sdg.")
    arrays.update(sd_to_np(le.state_dict(), "sdl."))
    arrays.update({"in.x": x.numpy(), "out.global": og.numpy(), "out.local": ol.numpy()})
    save("first_frame_nets_32x64", **arrays)


def golden_inference(cases=((1, "s1", 32, 64), (2, "s2", 32, 64))):
    """create_model(opt) -> Vid2VidModelG.inference over 3 generated frames, label2city flags
    (test.py:25-41, models/vid2vid_model_G.py:198-229), n_scales_spatial = 1 and 2 (and 3 at 64x128: the
    BASELINE configs[4] scale count, `python make_golden.py inference_s3`)."""
    from options.test_options import TestOptions
    from models import networks
    from models.models import create_model
    import tempfile
    for S, tag, H, W in cases:
        ck = tempfile.mkdtemp()
        argv = ["test.py", "--name", "g", "--label_nc", "35", "--loadSize", "64", "--use_instance", "--fg",
                "--use_real_img", "--gpu_ids", "-1", "--checkpoints_dir", ck, "--ngf", "8", "--n_blocks", "2",
                "--n_blocks_local", "1", "--n_scales_spatial", str(S), "--n_downsample_G", "2"]
        sys.argv = argv
        opt = TestOptions().parse(save=False)
        torch.manual_seed(50 + S)
        nets = [networks.define_G(108, 3, 6, opt.ngf, "composite", opt.n_downsample_G, opt.norm, 0, [], opt)]
        for s in range(1, S):
            nets.append(networks.define_G(108, 3, 6, opt.ngf // (2 ** s), "compositeLocal", opt.n_downsample_G,
                                          opt.norm, s, [], opt))
        with torch.no_grad():
            for n in nets:
                n.model_final_flow[1].weight.mul_(0.1)
        os.makedirs(os.path.join(ck, "g"), exist_ok=True)
        for s, n in enumerate(nets):
            torch.save(n.state_dict(), os.path.join(ck, "g", "latest_net_G%d.pth" % s))
        model = create_model(opt)
        gen = torch.Generator().manual_seed(60 + S)
        T = 5
        lab = synth_labels(gen, T, H, W, 35)
        inst = synth_labels(gen, T, H, W, 9)
        Bfirst = torch.tanh(torch.randn(1, 2, 3, H, W, generator=gen))
        outs = []
        model.fake_B_prev = None
        for t in range(T - 2):
            A = lab[t:t + 3].view(1, 3, 1, H, W)
            I = inst[t:t + 3].view(1, 3, 1, H, W)
            fake, real_A = model.inference(A, Bfirst if t == 0 else None, I)
            outs.append(fake.clone())
            if t == 0:
                last_label = real_A.clone()
        arrays = {}
        for s, n in enumerate(nets):
            arrays.update(sd_to_np(n.state_dict(), "sd%d." % s))
        arrays.update({"in.labels": lab.numpy(), "in.inst": inst.numpy(), "in.B": Bfirst.numpy(),
                       "out.fake": torch.cat(outs).numpy(), "out.real_A_last": last_label.numpy()})
        save("inference_label2city_%s_%dx%d" % (tag, H, W), **arrays)


def golden_edge2face():
    """BASELINE config C4 geometry (edge2face: label_nc = 0, input_nc = 15, no instance map, no fg tower;
    scripts/face/test_512.sh): create_model(opt) -> inference over 3 generated frames, real first frames given."""
    from options.test_options import TestOptions
    from models import networks
    from models.models import create_model
    import tempfile
    ck = tempfile.mkdtemp()
    sys.argv = ["test.py", "--name", "g", "--label_nc", "0", "--input_nc", "15", "--loadSize", "64",
                "--use_real_img", "--gpu_ids", "-1", "--checkpoints_dir", ck, "--ngf", "8", "--n_blocks", "2",
                "--n_scales_spatial", "1", "--n_downsample_G", "2"]
    opt = TestOptions().parse(save=False)
    torch.manual_seed(77)
    net = networks.define_G(45, 3, 6, opt.ngf, "composite", opt.n_downsample_G, opt.norm, 0, [], opt)
    with torch.no_grad():
        net.model_final_flow[1].weight.mul_(0.1)
    os.makedirs(os.path.join(ck, "g"), exist_ok=True)
    torch.save(net.state_dict(), os.path.join(ck, "g", "latest_net_G0.pth"))
    model = create_model(opt)
    gen = torch.Generator().manual_seed(78)
    H, W, T = 32, 32, 5
    A = torch.rand(1, T, 15, H, W, generator=gen)
    A[:, :, 0] = (A[:, :, 0] < 0.05).float()                  # sparse binary edge channel + smooth maps
    Bfirst = torch.tanh(torch.randn(1, 2, 3, H, W, generator=gen))
    outs = []
    model.fake_B_prev = None
    for t in range(T - 2):
        fake, real_A = model.inference(A[:, t:t + 3], Bfirst if t == 0 else None, None)
        outs.append(fake.clone())
        if t == 0:
            last = real_A.clone()
    arrays = sd_to_np(net.state_dict(), "sd0.")
    arrays.update({"in.A": A.numpy(), "in.B": Bfirst.numpy(), "out.fake": torch.cat(outs).numpy(),
                   "out.real_A_last": last.numpy()})
    save("inference_edge2face_s1_32x32", **arrays)


def golden_training():
    """One training chunk of the reference itself: Vid2VidModelG.forward (models/vid2vid_model_G.py:114-196),
    Vid2VidModelD.forward for the image and the temporal discriminators (models/vid2vid_model_D.py:93-213),
    get_losses (:249-264) and the three backward passes of train.py:86-93 (no optimizer step), label2city
    flags, n_scales_spatial = 2, 3 frames.  flow_ref / conf_ref are seeded tensors (FlowNet2 is CUDA-only)."""
    import tempfile
    from options.train_options import TrainOptions
    ck = tempfile.mkdtemp()
    sys.argv = ["train.py", "--name", "g", "--label_nc", "35", "--loadSize", "64", "--use_instance", "--fg",
                "--gpu_ids", "-1", "--checkpoints_dir", ck, "--ngf", "8", "--n_blocks", "2", "--n_blocks_local", "1",
                "--n_scales_spatial", "2", "--n_downsample_G", "2", "--no_vgg", "--num_D", "2", "--ndf", "8",
                "--n_frames_total", "6", "--max_frames_per_gpu", "3", "--n_scales_temporal", "1",
                "--niter_fix_global", "0"]
    opt = TrainOptions().parse(save=False)
    opt.gpu_ids = [-1]
    opt.n_gpus_gen = 1
    from models.vid2vid_model_G import Vid2VidModelG
    from models.vid2vid_model_D import Vid2VidModelD
    torch.manual_seed(70)
    G = Vid2VidModelG(); G.initialize(opt)
    D = Vid2VidModelD(); D.initialize(opt)
    with torch.no_grad():
        G.netG0.model_final_flow[1].weight.mul_(0.1)
        G.netG1.model_final_flow[1].weight.mul_(0.1)
    gen = torch.Generator().manual_seed(71)
    H, W, tG, tD = 32, 64, 3, 3
    nfl = G.n_frames_load
    t_len = nfl + tG - 1
    lab = synth_labels(gen, t_len, H, W, 35).view(1, t_len, 1, H, W)
    inst = synth_labels(gen, t_len, H, W, 9).view(1, t_len, 1, H, W)
    Bimg = torch.tanh(torch.nn.functional.interpolate(torch.randn(t_len, 3, H // 4, W // 4, generator=gen), scale_factor=4,
                                                      mode="bilinear", align_corners=False)).view(1, t_len, 3, H, W)
    flow_ref = torch.randn(1, nfl, 2, H, W, generator=gen) * 2.0
    conf_ref = (torch.rand(1, nfl, 1, H, W, generator=gen) > 0.3).float()

    def reshape(ts):
        return [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]

    fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = G(lab, Bimg, inst, None)
    real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
    fake_B_prev = G.compute_fake_B_prev(real_B_prev, None, fake_B)
    losses = D(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
    losses = [torch.mean(x) if x is not None else 0 for x in losses]
    loss_dict = dict(zip(D.loss_names, losses))
    frames_all = (None, None, None, None)
    frames_all, frames_skipped = D.get_all_skipped_frames(frames_all, real_B, fake_B, flow_ref, conf_ref, 1, tD, nfl, 0, None)
    loss_dict_T = []
    if frames_skipped[0][0] is not None:
        lt = D(1, [f[0] for f in frames_skipped])
        lt = [torch.mean(x) if not isinstance(x, int) else x for x in lt]
        loss_dict_T.append(dict(zip(D.loss_names_T, lt)))
    loss_G, loss_D, loss_D_T, t_act = D.get_losses(loss_dict, loss_dict_T, 1)
    assert t_act == 1

    nets = {"G0": G.netG0, "G1": G.netG1, "D": D.netD, "DT0": D.netD_T0}

    def zero():
        for n in nets.values():
            for p in n.parameters():
                p.grad = None

    arrays = {}
    for k, n in nets.items():
        arrays.update(sd_to_np(n.state_dict(), "sd%s." % k))

    def grab(tag, keys):
        for k in keys:
            for name, p in nets[k].named_parameters():
                if p.grad is not None:
                    arrays["grad%s.%s.%s" % (tag, k, name)] = p.grad.detach().numpy().copy()

    zero(); loss_G.backward(retain_graph=True); grab("G", ["G0", "G1"])
    zero(); loss_D.backward(retain_graph=True); grab("D", ["D"])
    zero(); loss_D_T[0].backward(); grab("DT", ["DT0"])
    arrays.update({"in.labels": lab.numpy(), "in.inst": inst.numpy(), "in.B": Bimg.numpy(),
                   "in.flow_ref": flow_ref.numpy(), "in.conf_ref": conf_ref.numpy(),
                   "out.fake_B": fake_B.detach().numpy(), "out.fake_B_raw": fake_B_raw.detach().numpy(),
                   "out.flow": flow.detach().numpy(), "out.weight": weight.detach().numpy(),
                   "out.real_A": real_A.detach().numpy()})
    for k, v in loss_dict.items():
        arrays["loss." + k] = np.array(float(v))
    for k, v in loss_dict_T[0].items():
        arrays["loss." + k] = np.array(float(v))
    arrays["loss.total_G"] = np.array(float(loss_G)); arrays["loss.total_D"] = np.array(float(loss_D))
    arrays["loss.total_D_T0"] = np.array(float(loss_D_T[0]))
    for i, t in enumerate(frames_skipped):
        arrays["skipped.%d" % i] = t[0].detach().numpy()
    save("training_label2city_s2_32x64", **arrays)
    print({k: float(v) for k, v in arrays.items() if k.startswith("loss.")})


def golden_flownet2():
    """FlowNet2.forward (models/flownet2_pytorch/models.py:96-161) + FlowNet.compute_flow_and_conf
    (models/flownet.py:43-59) executed from the reference's own Python on CPU.  The three CUDA-only
    extensions cannot be built here (no nvcc; legacy ATen API), so the modules `resample2d_cuda`,
    `channelnorm_cuda` and the `Correlation` layer are backed by the CPU restatements in
    oracle/vid2vid_oracle.py (which follow the .cu files): this fixture pins the network COMPOSITION
    (117 convs, upsampling, warping chain), not those three kernels ("parity unpinned").
    Weights: tests/util.seeded_flownet2_weights (162.5 M values, regenerated from a seed by the tests)."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))           # tests/
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))        # repo root
    from util import seeded_flownet2_weights
    from oracle import vid2vid_oracle as O

    r2d = types.ModuleType("resample2d_cuda")
    def r2d_forward(img, flow, out, ksize):
        out.copy_(O.resample2d(img, flow, ksize))
    r2d.forward = r2d_forward
    cn = types.ModuleType("channelnorm_cuda")
    def cn_forward(x, out, norm_deg):
        assert norm_deg == 2
        out.copy_(O.channelnorm(x))
    cn.forward = cn_forward
    sys.modules["resample2d_cuda"] = r2d
    sys.modules["channelnorm_cuda"] = cn
    sys.modules["correlation_cuda"] = types.ModuleType("correlation_cuda")
    from models.flownet2_pytorch import models as f2
    from models.flownet2_pytorch.networks.correlation_package import correlation as corr_mod

    def corr_forward(self, a, b):        # legacy (non-static) autograd Function cannot run on torch >= 1.x
        return O.correlation(a, b, self.pad_size, self.kernel_size, self.max_displacement, self.stride1, self.stride2)
    corr_mod.Correlation.forward = corr_forward

    net = f2.FlowNet2(fp16=False)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(seeded_flownet2_weights(shapes))
    net.eval()
    gen = torch.Generator().manual_seed(90)
    B, H, W = 2, 64, 128
    base = torch.nn.functional.interpolate(torch.randn(B, 3, 8, 16, generator=gen), size=(H, W), mode="bicubic",
                                           align_corners=False)
    im1 = torch.tanh(base)
    im2 = torch.tanh(torch.roll(base, shifts=(1, 2), dims=(2, 3)) + 0.05 * torch.randn(B, 3, H, W, generator=gen))
    with torch.no_grad():
        data = torch.cat([im1.unsqueeze(2), im2.unsqueeze(2)], dim=2)
        flow = net(data)
        warped = O.resample2d(im2, flow)
        ssd = ((im1 - warped) ** 2).sum(1, keepdim=True)
        conf = (ssd < 0.02).float()
    arrays = {"in.im1": im1.numpy(), "in.im2": im2.numpy(), "out.flow": flow.numpy(), "out.conf": conf.numpy(),
              "out.ssd": ssd.numpy()}
    arrays["keys"] = np.array(sorted(shapes.keys()))
    arrays["shapes"] = np.array([",".join(str(d) for d in shapes[k]) for k in sorted(shapes.keys())])
    save("flownet2_64x128", **arrays)
    print("flow rms %.3f  max %.3f  conf mean %.3f" % (flow.pow(2).mean().sqrt(), flow.abs().max(), conf.mean()))


def main():
    install_shims()
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only == "training":
        return golden_training()
    if only == "flownet2":
        return golden_flownet2()
    if only == "edge2face":
        return golden_edge2face()
    if only == "inference_s3":
        return golden_inference(((3, "s3", 64, 128),))
    from models import networks
    golden_composite(networks)
    golden_composite_local(networks)
    golden_discriminator(networks)
    golden_global(networks)
    golden_inference()
    golden_inference(((3, "s3", 64, 128),))
    golden_edge2face()
    golden_training()
    golden_flownet2()


if __name__ == "__main__":
    main()
