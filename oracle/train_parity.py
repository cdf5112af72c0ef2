"""Training-chunk parity harness: ONE chunk of train.py's inner loop on both sides, compared.

TEST INFRASTRUCTURE (like everything under oracle/): imported by tests/ and by bench.py's parity legs only, never by
the product package.  The CPU side is the oracle restatement (oracle/vid2vid_oracle.py -- pinned to the reference's own
Vid2VidModelG / Vid2VidModelD outputs, losses and autograd gradients by tests/golden/training_label2city_s2_32x64.npz,
tests/test_cpu_oracle.py::test_training_oracle_vs_reference) driven by plain torch CPU autograd; the other side is the
product's Vid2VidModelG.forward / Vid2VidModelD.forward / backward on the MI355X, handed in as objects.

What one chunk is (reference train.py:50-93, first chunk of a sequence):
    fake_B, fake_B_raw, flow, weight, real_A, real_Bp, _ = modelG(input_A, input_B, inst_A, None)
    losses   = modelD(0, [real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref])
    losses_T = modelD(1, skipped frames of temporal scale 0)          (active when the chunk holds >= tD frames)
    loss_G, loss_D, loss_D_T = get_losses(...)                        (models/vid2vid_model_D.py:249-264)
    three backward passes: loss_G -> G parameters, loss_D -> D parameters, loss_D_T[0] -> D_T0 parameters
flow_ref / conf_ref are INPUTS here (the same tensors on both sides; FlowNet2 has its own parity tests).
"""
import time

import torch

from . import vid2vid_oracle as O


def _trainable(sd, dtype=torch.float32):
    out = {}
    for k, v in sd.items():
        v = v.detach().to(dtype).cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


def _flat(named):
    return torch.cat([t.reshape(-1).double() for _, t in named]) if named else torch.zeros(0, dtype=torch.float64)


def oracle_chunk(sds_G, sd_D, sd_DT, lab, inst, B, flow_ref, conf_ref, *, label_nc=35, fg=True, fg_labels=(26,),
                 n_down=3, n_blocks=9, n_blocks_local=3, n_frames_load=3, tG=3, tD=3, n_layers_D=3, num_D=2,
                 lambda_feat=10.0, lambda_F=10.0, lambda_T=10.0, sd_vgg=None, param_names=None, dtype=torch.float32):
    """CPU side.  sds_G: state_dicts of netG0..netG{S-1}; lab / inst (1,T,1,H,W), B (1,T,3,H,W) with
    T = n_frames_load + tG - 1; flow_ref (1,n_frames_load,2,H,W), conf_ref (1,n_frames_load,1,H,W).
    param_names: {"G": [[names of netG0], ...], "D": [...], "DT": [...]} -- the parameter order the gradients are
    flattened in (the product's named_parameters() order)."""
    t0 = time.perf_counter()
    S = len(sds_G)
    # dtype = torch.float64 (with torch.set_default_dtype(torch.float64) around the call): the exact-arithmetic stand-in that
    # scripts/oracle_noise_floor.py measures the fp32 oracle itself against
    sds_G = [_trainable(sd, dtype) for sd in sds_G]
    sd_D = _trainable(sd_D, dtype)
    sd_DT = None if sd_DT is None else _trainable(sd_DT, dtype)
    real_A = O.encode_input(lab, inst, label_nc)
    fake_B, fake_B_raw, flow, weight, fake_pyr = O.generate_frames_train(sds_G, real_A, B, fg, list(fg_labels), n_down, n_blocks,
                                                                         n_blocks_local, n_frames_load, tG, return_pyramid=True)
    real_Bp = B[:, tG - 2:]
    real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
    fake_B_prev = torch.cat([real_B_prev[:, 0:1], fake_B[:, :-1].detach()], 1)          # compute_fake_B_prev (:332-336)
    r4 = lambda t: t.reshape(-1, t.shape[2], t.shape[3], t.shape[4])
    t = dict(real_B=r4(real_B), fake_B=r4(fake_B), fake_B_raw=r4(fake_B_raw), real_A=r4(real_A[:, tG - 1:]),
             real_B_prev=r4(real_B_prev), fake_B_prev=r4(fake_B_prev), flow=r4(flow), weight=r4(weight),
             flow_ref=r4(flow_ref), conf_ref=r4(conf_ref))
    losses = O.model_D_image_losses(sd_D, t, n_layers_D, num_D, lambda_feat, lambda_F, lambda_T, S, sd_vgg)
    loss_G = (losses["G_GAN"] + losses["G_GAN_Feat"] + losses["G_VGG"] + losses["G_Warp"] + losses["F_Flow"]
              + losses["F_Warp"] + losses["W"])
    loss_D = (losses["D_fake"] + losses["D_real"]) * 0.5
    loss_DT = None
    lt = {}
    if sd_DT is not None and n_frames_load >= tD:
        # temporal scale 0 of the first chunk: the groups of tD consecutive frames ending at each of the newest frames
        # (get_skipped_frames, models/vid2vid_model_D.py:274-290); the flow of a group = the tD-1 flows between its frames
        ng = n_frames_load - tD + 1
        rb = torch.cat([real_B[:, i:i + tD] for i in range(ng)], 0)
        fb = torch.cat([fake_B[:, i:i + tD] for i in range(ng)], 0)
        fr = torch.cat([flow_ref[:, i + 1:i + tD] for i in range(ng)], 0)
        lt = O.model_D_temporal_losses(sd_DT, rb, fb, fr, tD, n_layers_D, num_D, lambda_feat)
        loss_G = loss_G + lt["G_T_GAN"] + lt["G_T_GAN_Feat"] + lt["G_T_Warp"]
        loss_DT = (lt["D_T_fake"] + lt["D_T_real"]) * 0.5

    def grads(loss, sd, names):
        ps = [sd[n] for n in names]
        gs = torch.autograd.grad(loss, ps, retain_graph=True, allow_unused=True)
        return [(n, torch.zeros_like(p) if g is None else g.detach()) for n, p, g in zip(names, ps, gs)]

    if param_names is None:
        param_names = {"G": [[k for k, v in sd.items() if v.requires_grad] for sd in sds_G],
                       "D": [k for k, v in sd_D.items() if v.requires_grad],
                       "DT": [] if sd_DT is None else [k for k, v in sd_DT.items() if v.requires_grad]}
    gG = []
    for si in range(S):
        gG += [("G%d.%s" % (si, n), g) for n, g in grads(loss_G, sds_G[si], param_names["G"][si])]
    gD = grads(loss_D, sd_D, param_names["D"])
    gDT = grads(loss_DT, sd_DT, param_names["DT"]) if loss_DT is not None else []
    out = dict(
        outs=dict(fake_B=fake_B.detach(), fake_B_raw=fake_B_raw.detach(), flow=flow.detach(), weight=weight.detach()),
        losses={k: float(v.detach()) for k, v in list(losses.items()) + list(lt.items())},
        totals=dict(G=float(loss_G.detach()), D=float(loss_D.detach()), DT=None if loss_DT is None else float(loss_DT.detach())),
        grads=dict(G=_flat(gG), D=_flat(gD), DT=_flat(gDT)),
        # (name, numel) of every tensor of the flattened gradients, in order: compare() attributes the L2 distance to tensors
        grad_names=dict(G=[(n, g.numel()) for n, g in gG], D=[(n, g.numel()) for n, g in gD], DT=[(n, g.numel()) for n, g in gDT]),
        # fake_B_pyr of the chunk (finest scale first; the tG-1 given frames, then the generated ones): what a teacher-forced
        # product run starts every frame t > 0 from (hip_chunk(teacher=...))
        fake_pyr=fake_pyr,
        seconds=time.perf_counter() - t0)
    return out


def hip_chunk(G, D, lab, inst, B, flow_ref, conf_ref, teacher=None):
    """The product side: the same chunk through Vid2VidModelG.forward / Vid2VidModelD.forward (train.py:50-82) and the three
    backward passes (train.py:86-93).  G, D: initialised Vid2VidModelG / Vid2VidModelD on the GPU; tensors on the GPU.

    teacher = the oracle's `fake_pyr`: TEACHER-FORCED run.  north_star's bar is "the same inputs" on both sides; in a
    chunk of several frames the inputs of frame t > 0 include the previous fake frames, so every frame t > 0 starts from the
    REFERENCE's previous frames at every scale instead of the product's own (which differ by ~1e-4 and would propagate).
    With n_frames_bp = 1 those frames are detached on both sides (models/vid2vid_model_G.py:167-168), so the autograd graph
    is the chunk's own: the frames are generated by one Vid2VidModelG.forward call each (`frame_range`, the entry point the
    generator ranks of roles.py use), concatenated, and everything behind G -- the discriminators' batch over all frames,
    the losses, the three backward passes -- is the single chunk of train.py.  teacher=None: free-running chunk."""
    opt = G.opt
    tD = opt.n_frames_D
    tG = opt.n_frames_G

    def reshape(ts):
        return [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]

    if teacher is None or G.n_frames_load == 1:
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp, _ = G(lab, B, inst, None)
    else:
        dev = B.device
        parts = []
        for t in range(G.n_frames_load):
            prev = None if t == 0 else [p[:, t:t + tG - 1].to(dev, torch.float32).contiguous() for p in teacher]
            o = G(lab, B, inst, prev, frame_range=(t, t + 1), first_chunk=True)
            parts.append(o[:4])
            real_A, real_Bp = o[4], o[5]
        fake_B, fake_B_raw, flow, weight = [torch.cat([p[i] for p in parts], 1) for i in range(4)]
    real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
    fake_B_prev = G.compute_fake_B_prev(real_B_prev, None, fake_B)
    losses = D(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
    loss_dict = dict(zip(D.loss_names, [torch.mean(x) for x in losses]))
    t_scales = opt.n_scales_temporal
    _, skipped = D.get_all_skipped_frames((None,) * 4, real_B, fake_B, flow_ref, conf_ref, t_scales, tD,
                                          G.n_frames_load, 0, None)
    loss_dict_T = []
    for s in range(t_scales):
        if skipped[0] is not None and skipped[0][s] is not None:
            lt = D(s + 1, [f[s] for f in skipped])
            loss_dict_T.append(dict(zip(D.loss_names_T, [torch.mean(x) for x in lt])))
    loss_G, loss_D, loss_D_T, t_act = D.get_losses(loss_dict, loss_dict_T, t_scales)
    opts_T = [getattr(D, "optimizer_D_T%d" % s) for s in range(t_scales)]

    def zero_all():
        G.optimizer_G.zero_grad(); D.optimizer_D.zero_grad()
        for o in opts_T:
            o.zero_grad()

    def flat_of(nets):
        named = []
        for tag, net in nets:
            for n, p in net.named_parameters():
                if p.requires_grad:
                    named.append((tag + n, torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()))
        return torch.cat([t.reshape(-1).double().cpu() for _, t in named])

    S = opt.n_scales_spatial
    netsG = [("G%d." % si, getattr(G, "netG%d" % si)) for si in range(S)]
    zero_all()
    loss_G.backward(retain_graph=True)
    gG = flat_of(netsG)
    zero_all()
    loss_D.backward(retain_graph=True)
    gD = flat_of([("", D.netD)])
    gDT = torch.zeros(0, dtype=torch.float64)
    if t_act > 0:
        zero_all()
        loss_D_T[0].backward()
        gDT = flat_of([("", D.netD_T0)])
    zero_all()
    losses_f = {k: float(v.detach()) for k, v in loss_dict.items()}
    if loss_dict_T:
        losses_f.update({k: float(v.detach()) for k, v in loss_dict_T[0].items()})
    return dict(
        outs=dict(fake_B=fake_B.detach().float().cpu(), fake_B_raw=fake_B_raw.detach().float().cpu(),
                  flow=flow.detach().float().cpu(), weight=weight.detach().float().cpu()),
        losses=losses_f,
        totals=dict(G=float(loss_G.detach()), D=float(loss_D.detach()), DT=float(loss_D_T[0].detach()) if t_act > 0 else None),
        grads=dict(G=gG, D=gD, DT=gDT))


def param_names_of(G, D):
    S = G.opt.n_scales_spatial
    names = {"G": [[n for n, p in getattr(G, "netG%d" % si).named_parameters() if p.requires_grad] for si in range(S)],
             "D": [n for n, p in D.netD.named_parameters() if p.requires_grad],
             "DT": [n for n, p in D.netD_T0.named_parameters() if p.requires_grad] if hasattr(D, "netD_T0") else []}
    return names


def attribute_grad_error(g, r, names, top=12):
    """Where the relative L2 distance of a flattened gradient comes from: per parameter tensor
    d_i = |g_i - r_i|^2, reported as its share of sum d, with the tensor's own relative error |g_i - r_i| / |r_i| and the
    ratio of the norms; grouped by parameter kind (conv weight / conv bias / norm weight / norm bias) and by the sub-module the
    tensor belongs to.  `share` sums to 1 over all tensors."""
    rows, o = [], 0
    for n, k in names:
        gi, ri = g[o:o + k], r[o:o + k]
        o += k
        d2, r2, g2 = float((gi - ri).pow(2).sum()), float(ri.pow(2).sum()), float(gi.pow(2).sum())
        rows.append((n, k, d2, r2, g2))
    assert o == g.numel() == r.numel()
    tot = sum(x[2] for x in rows) or 1e-300
    rtot = sum(x[3] for x in rows) or 1e-300

    def kind(n, k):
        leaf = n.rsplit(".", 1)[-1]
        # the smallest conv weight of these networks (16 -> 3, 7x7 aside: 3 x 16 x 49) is larger than the widest norm (1024)
        return ("norm " if leaf == "weight" and k <= 1024 else "conv / norm " if leaf == "bias" else "conv ") + leaf

    def group(n):
        parts = n.split(".")
        return ".".join(parts[:2]) if len(parts) > 2 else parts[0]      # "G0.model_down_seg", "scale0_layer0", ...
    by_group, by_kind = {}, {}
    for n, k, d2, r2, g2 in rows:
        a = by_group.setdefault(group(n), [0.0, 0.0]); a[0] += d2; a[1] += r2
        b = by_kind.setdefault(kind(n, k), [0.0, 0.0]); b[0] += d2; b[1] += r2
    fmt = lambda v: float("%.3e" % v)
    return {
        "l2_rel_err": fmt((tot / rtot) ** 0.5),
        "top_tensors": [{"name": n, "numel": k, "share": fmt(d2 / tot), "rel_err": fmt((d2 / max(r2, 1e-300)) ** 0.5),
                         "norm_ratio": fmt((g2 / max(r2, 1e-300)) ** 0.5), "ref_norm_share": fmt((r2 / rtot) ** 0.5)}
                        for n, k, d2, r2, g2 in sorted(rows, key=lambda x: -x[2])[:top]],
        "by_module": {m: {"share": fmt(v[0] / tot), "rel_err": fmt((v[0] / max(v[1], 1e-300)) ** 0.5)}
                      for m, v in sorted(by_group.items(), key=lambda kv: -kv[1][0])[:top]},
        "by_kind": {m: {"share": fmt(v[0] / tot), "rel_err": fmt((v[0] / max(v[1], 1e-300)) ** 0.5)} for m, v in sorted(by_kind.items(), key=lambda kv: -kv[1][0])},
        "tensors": len(rows),
    }


def compare(got, ref):
    """Errors of the product's chunk against the oracle's.  Per-pixel measure of the forward tensors as everywhere
    (|got-ref| / (|ref| + rms(ref)): maximum and mean); losses: |got-ref| / max(|ref|, 1e-3); gradients per optimizer:
    relative error of the norm and relative L2 distance of the whole flattened gradient."""
    out = {"forward": {}, "losses": {}, "grads": {}}
    for k, r in ref["outs"].items():
        g = got["outs"][k]
        rms = r.pow(2).mean().sqrt().item() + 1e-12
        e = (g - r).abs() / (r.abs() + rms)
        # frame 0 of the chunk has identical inputs on both sides (the given real frames); later frames are fed by each
        # side's own previous outputs, so their error includes the propagated difference
        out["forward"][k] = {"max_rel": float("%.3e" % e.max().item()), "mean_rel": float("%.3e" % e.mean().item()),
                             "max_rel_frame0": float("%.3e" % e[:, 0].max().item()),
                             "finite": bool(torch.isfinite(g).all().item())}
    for k, r in ref["losses"].items():
        if k in got["losses"]:
            out["losses"][k] = float("%.3e" % (abs(got["losses"][k] - r) / max(abs(r), 1e-3)))
    for k in ("G", "D", "DT"):
        r = ref["totals"][k]
        if r is not None and got["totals"][k] is not None:
            out["losses"]["total_" + k] = float("%.3e" % (abs(got["totals"][k] - r) / max(abs(r), 1e-3)))
    for k in ("G", "D", "DT"):
        r, g = ref["grads"][k], got["grads"][k]
        if r.numel() == 0 or g.numel() == 0:
            continue
        assert r.numel() == g.numel(), "gradient of %s: %d values vs %d" % (k, g.numel(), r.numel())
        rn, gn = r.norm().item(), g.norm().item()
        out["grads"][k] = {"norm_ref": float("%.6e" % rn), "norm_got": float("%.6e" % gn),
                           "norm_rel_err": float("%.3e" % (abs(gn - rn) / max(rn, 1e-30))),
                           "l2_rel_err": float("%.3e" % ((g - r).norm().item() / max(rn, 1e-30))),
                           "finite": bool(torch.isfinite(g).all().item()), "numel": int(r.numel())}
    if "grad_names" in ref:
        out["grad_error_by_tensor"] = {k: attribute_grad_error(got["grads"][k], ref["grads"][k], ref["grad_names"][k])
                                       for k in ("G", "D", "DT") if ref["grads"][k].numel() and got["grads"][k].numel()}
    out["max_forward"] = max(v["max_rel"] for v in out["forward"].values())
    out["max_forward_frame0"] = max(v["max_rel_frame0"] for v in out["forward"].values())
    out["max_loss"] = max(out["losses"].values())
    out["max_grad_norm"] = max(v["norm_rel_err"] for v in out["grads"].values())
    out["max_grad_l2"] = max(v["l2_rel_err"] for v in out["grads"].values())
    return out
