"""Generator / discriminator rank roles (vid2vid_amd/roles.py, the reference's `n_gpus_gen` split as one process per GPU)
on CPU with gloo: a sequence group of 2 generator ranks + 2 (or 1) discriminator ranks runs the reference's train.py
inner loop (train.py:50-93) for two chunks and must reproduce a SINGLE process doing the same chunks -- every loss value,
the complete gradients of G (summed over the generator ranks by the all-reduce), D and D_T, and the updated parameters.

The models here are small torch stand-ins with the model API's surface (the real networks only compute on the MI355X):
what is under test is the role runtime -- frame ranges, the fake-frame hand-over between generator ranks and between
chunks, the autograd bridges (_GradSink / _Remote), loss-value forwarding, per-role optimizers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

H, W, TG, TD = 8, 12, 3, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _Opt:
    def __init__(self, n_gen, group, k):
        self.n_frames_G, self.n_frames_D, self.output_nc, self.no_flow = TG, TD, 3, False
        self.n_gpus_gen, self.gpu_ids, self.batchSize, self.isTrain = n_gen, list(range(group)), 1, True
        self.n_scales_temporal, self.lr, self.max_frames_per_gpu, self.sparse_D = 1, 0.05, k, False


class FlatOpt:
    """SGD over optim.FlatBuffers with the FusedAdam hooks roles.py uses (flat, grad_sync); the all-reduced gradient stays
    readable in flat.flat_grad after step()."""

    def __init__(self, params, lr):
        from vid2vid_amd.optim import FlatBuffers
        self.flat, self.lr, self.grad_sync = FlatBuffers(list(params)), lr, None
        self.param_groups = [{"lr": lr}]

    def zero_grad(self):
        self.flat.rebind_grads(); self.flat.flat_grad.zero_()

    def step(self):
        scale = self.grad_sync.all_reduce(self.flat.flat_grad) if self.grad_sync is not None else 1.0
        self.flat.flat_grad.mul_(scale)
        with torch.no_grad():
            self.flat.flat_param.add_(self.flat.flat_grad, alpha=-self.lr)


class TinyG(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt, self.n_scales = opt, 1
        self.n_gpus = opt.n_gpus_gen
        self.n_frames_per_gpu = opt.max_frames_per_gpu
        self.n_frames_load = self.n_gpus * self.n_frames_per_gpu
        self.net = nn.Sequential(nn.Conv2d(TG + (TG - 1) * 3, 8, 3, padding=1), nn.Tanh(), nn.Conv2d(8, 3 + 3 + 2 + 1, 3, padding=1))
        self.optimizer_G = FlatOpt(self.net.parameters(), opt.lr)
        self.saved = []

    def encode_input(self, A, B, inst):
        return A.float(), B.float(), None

    def forward(self, A, B, inst, prev, frame_range=None, first_chunk=None):
        real_A, real_B, _ = self.encode_input(A, B, inst)
        pyr = [real_B[:, :TG - 1]] if prev is None else [p for p in prev]
        t0, t1 = (0, self.n_frames_load) if frame_range is None else frame_range
        fakes, raws, flows, weights = [], [], [], []
        hist = pyr[0]
        for t in range(t0, t1):
            lt = t - t0
            x = torch.cat([real_A[:, t:t + TG].reshape(1, -1, H, W), hist[:, lt:lt + TG - 1].detach().reshape(1, -1, H, W)], 1)
            y = self.net(x)
            raw, w = torch.tanh(y[:, 3:6]), torch.sigmoid(y[:, 8:9])
            fake = torch.tanh(y[:, :3]) * w + raw * (1 - w)
            hist = torch.cat([hist, fake.unsqueeze(1)], 1)
            fakes.append(fake.unsqueeze(1)); raws.append(raw.unsqueeze(1)); flows.append(y[:, 6:8].unsqueeze(1)); weights.append(w.unsqueeze(1))
        cat = lambda xs: torch.cat(xs, 1)
        return cat(fakes), cat(raws), cat(flows), cat(weights), real_A[:, TG - 1:], real_B[:, TG - 2:], [hist[:, -(TG - 1):].detach()]

    def compute_fake_B_prev(self, real_B_prev, fake_B_last, fake_B):
        prev = real_B_prev[:, 0:1] if fake_B_last is None else fake_B_last[0][:, -1:]
        return torch.cat([prev, fake_B[:, :-1].detach()], 1) if fake_B.size(1) > 1 else prev

    def save(self, label):
        self.saved.append(label)


class TinyD(nn.Module):
    loss_names = ["G_VGG", "G_GAN", "G_GAN_Feat", "D_real", "D_fake", "G_Warp", "F_Flow", "F_Warp", "W"]
    loss_names_T = ["G_T_GAN", "G_T_GAN_Feat", "D_T_real", "D_T_fake", "G_T_Warp"]

    def __init__(self, opt):
        super().__init__()
        self.opt, self.gpu_ids = opt, [0]
        self.netD = nn.Sequential(nn.Conv2d(1 + 3, 6, 3, padding=1), nn.BatchNorm2d(6), nn.LeakyReLU(0.2), nn.Conv2d(6, 1, 3, padding=1))
        self.netD_T0 = nn.Sequential(nn.Conv2d(3 * TD + 2 * (TD - 1), 6, 3, padding=1), nn.BatchNorm2d(6), nn.LeakyReLU(0.2), nn.Conv2d(6, 1, 3, padding=1))
        self.optimizer_D = FlatOpt(self.netD.parameters(), opt.lr)
        self.optimizer_D_T0 = FlatOpt(self.netD_T0.parameters(), opt.lr)
        self.saved = []

    @staticmethod
    def _gan(net, real, fake):
        pr, pf_d, pf = net(real), net(fake.detach()), net(fake)
        return F.mse_loss(pr, torch.ones_like(pr)), F.mse_loss(pf_d, torch.zeros_like(pf_d)), F.mse_loss(pf, torch.ones_like(pf)), F.l1_loss(pf, pr.detach())

    def forward(self, scale_T, ts):
        z = torch.zeros(())
        if scale_T > 0:
            real_B, fake_B, flow_ref, conf_ref = ts
            n = real_B.shape[0]
            fr = (flow_ref / 20).reshape(n, -1, H, W)
            dr, df, g, fm = self._gan(self.netD_T0, torch.cat([real_B.reshape(n, -1, H, W), fr], 1), torch.cat([fake_B.reshape(n, -1, H, W), fr], 1))
            return [l.view(1, 1) for l in (g, fm, dr, df, z)]
        real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref = ts
        dr, df, g, fm = self._gan(self.netD, torch.cat([real_A, real_B], 1), torch.cat([real_A, fake_B], 1))
        dr2, df2, g2, fm2 = self._gan(self.netD, torch.cat([real_A, real_B], 1), torch.cat([real_A, fake_B_raw], 1))
        f_flow = F.l1_loss(flow * conf_ref, flow_ref * conf_ref)
        g_warp = F.l1_loss(fake_B * conf_ref, (fake_B_prev * 0.5).detach() * conf_ref)
        f_warp = F.l1_loss((real_B_prev + 0.1 * flow.mean(1, keepdim=True)) * conf_ref, real_B * conf_ref)
        w_loss = weight.mean() * 0.01
        return [l.view(1, 1) for l in (z, g + g2, fm + fm2, dr + dr2, df + df2, g_warp, f_flow, f_warp, w_loss)]

    def get_all_skipped_frames(self, frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_frames_load, i, flowNet):
        from vid2vid_amd.models.vid2vid_model_D import get_skipped_frames, get_skipped_flows
        rB_all, fB_all, fl_all, cf_all = frames_all
        rB_all, rB_sk = get_skipped_frames(rB_all, real_B, t_scales, tD)
        fB_all, fB_sk = get_skipped_frames(fB_all, fake_B, t_scales, tD)
        fl_all, cf_all, fl_sk, cf_sk = get_skipped_flows(flowNet, fl_all, cf_all, rB_sk, flow_ref, conf_ref, t_scales, tD)
        return (rB_all, fB_all, fl_all, cf_all), (rB_sk, fB_sk, fl_sk, cf_sk)

    def get_losses(self, loss_dict, loss_dict_T, t_scales):
        from vid2vid_amd.models.vid2vid_model_D import Vid2VidModelD
        return Vid2VidModelD.get_losses(self, loss_dict, loss_dict_T, t_scales)

    def save_network(self, net, name, label, gpu_ids):
        self.saved.append((name, label))

    def save(self, label):
        self.save_network(self.netD, "D", label, None); self.save_network(self.netD_T0, "D_T0", label, None)


class TinyFlow(nn.Module):
    def forward(self, a, b):
        bsz, n = a.shape[:2]
        flow = torch.stack([(a - b).mean(2), (a + b).mean(2) * 0.5], 2) * 3.0
        conf = ((a - b).abs().mean(2, keepdim=True) < 0.6).float()
        return flow, conf


def _build(n_gen, group, k, seed=5):
    torch.manual_seed(seed)
    opt = _Opt(n_gen, group, k)
    return opt, TinyG(opt), TinyD(opt), TinyFlow()


def _sequence(n_frames, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randint(0, 4, (1, n_frames, 1, H, W), generator=g).float()
    B = torch.tanh(torch.randn(1, n_frames, 3, H, W, generator=g))
    return A, B


def _train(modelG, modelD, flowNet, opt, A, B, n_chunks, record):
    """train.py:47-93 for one sequence (models already wrapped): returns nothing, fills `record`."""
    mG, mD = modelG.module, modelD.module
    n_load, t_scales, tD = mG.n_frames_load, opt.n_scales_temporal, TD
    optimizer_G, optimizer_D, optimizer_D_T = mG.optimizer_G, mD.optimizer_D, [mD.optimizer_D_T0]
    fake_B_prev_last, frames_all = None, (None, None, None, None)
    reshape = lambda ts: [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]
    for c in range(n_chunks):
        i = c * n_load
        sl = slice(i, i + n_load + TG - 1)
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = modelG(A[:, sl], B[:, sl], None, fake_B_prev_last)
        real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
        flow_ref, conf_ref = flowNet(real_B, real_B_prev)
        fake_B_prev = mG.compute_fake_B_prev(real_B_prev, fake_B_prev_last, fake_B)
        fake_B_prev_last = fake_B_last
        losses = modelD(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
        loss_dict = dict(zip(mD.loss_names, [torch.mean(x) for x in losses]))
        frames_all, skipped = mD.get_all_skipped_frames(frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_load, i, flowNet)
        loss_dict_T = []
        for s in range(t_scales):
            if skipped[0][s] is not None:
                lt = modelD(s + 1, [f[s] for f in skipped])
                loss_dict_T.append(dict(zip(mD.loss_names_T, [torch.mean(x) for x in lt])))
        loss_G, loss_D, loss_D_T, t_act = mD.get_losses(loss_dict, loss_dict_T, t_scales)
        rec = {"losses": {k: float(v.detach()) for k, v in list(loss_dict.items()) + [kv for d in loss_dict_T for kv in d.items()]}}
        for name, loss, opt_ in [("G", loss_G, optimizer_G), ("D", loss_D, optimizer_D)] + [("DT", loss_D_T[s], optimizer_D_T[s]) for s in range(t_act)]:
            opt_.zero_grad(); loss.backward(); opt_.step()
            if hasattr(opt_, "flat"):
                rec["grad_" + name] = opt_.flat.flat_grad.clone()
        record.append(rec)


class _Plain(nn.Module):
    def __init__(self, m):
        super().__init__(); self.module = m

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _baseline(n_gen, k, n_chunks, seq_seed):
    opt, G, D, Fn = _build(n_gen, n_gen, k)                   # one process generates all n_gen * k frames of a chunk
    A, B = _sequence(n_chunks * n_gen * k + TG - 1, seq_seed)
    rec = []
    _train(_Plain(G), _Plain(D), _Plain(Fn), opt, A, B, n_chunks, rec)
    params = {"G": G.optimizer_G.flat.flat_param.clone(), "D": D.optimizer_D.flat.flat_param.clone(), "DT": D.optimizer_D_T0.flat.flat_param.clone()}
    return rec, params


def _role_worker(rank, world, port, n_gen, group, k, n_chunks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from vid2vid_amd import roles, parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opt, G, D, Fn = _build(n_gen, group, k)
        L = roles.layout_from_opt(opt)
        assert L is not None and L.n_gen == n_gen and L.n_disc == group - n_gen and L.n_groups == world // group
        opt.role_group_size = group
        modelG, modelD, flowNet = roles.wrap_roles(opt, G, D, Fn, L)
        assert isinstance(G.optimizer_G, roles.NullOptimizer) == (L.role != "G")
        assert isinstance(D.optimizer_D, roles.NullOptimizer) == (not L.owns_D)
        assert isinstance(D.optimizer_D_T0, roles.NullOptimizer) == (not L.owns_DT)
        A, B = _sequence(n_chunks * n_gen * k + TG - 1, 100 + L.seq)              # every rank of a sequence group loads the same sequence
        rec = []
        _train(modelG, modelD, flowNet, opt, A, B, n_chunks, rec)
        G.save("latest"); D.save("latest")
        out = {"role": L.role, "g": L.g, "d": L.d, "seq": L.seq, "owns_D": L.owns_D, "owns_DT": L.owns_DT, "rec": rec,
               "saved_G": G.saved, "saved_D": D.saved}
        if L.owns_G:
            out["param_G"] = G.optimizer_G.flat.flat_param.clone()
        if L.owns_D:
            out["param_D"] = D.optimizer_D.flat.flat_param.clone()
        if L.owns_DT:
            out["param_DT"] = D.optimizer_D_T0.flat.flat_param.clone()
        dist.barrier()
        q.put((rank, _plain(out)))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        parallel._ACTIVE_SYNCS.clear()
        dist.destroy_process_group()


def _plain(o):
    """Tensors -> numpy before a result crosses the process boundary: a tensor in a multiprocessing queue travels as a file
    descriptor served by the SENDER, and a worker that exits right after q.put() takes that server with it (FileNotFoundError
    in the parent's q.get(), seen once the workers got fast enough)."""
    if isinstance(o, torch.Tensor):
        return ("__t__", o.detach().cpu().numpy())
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(v) for v in o)
    return o


def _tensors(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == "__t__":
        return torch.from_numpy(o[1])
    if isinstance(o, dict):
        return {k: _tensors(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_tensors(v) for v in o)
    return o


def _run_roles(world, n_gen, group, k, n_chunks):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_role_worker, args=(r, world, port, n_gen, group, k, n_chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out = q.get(timeout=300)
        res[r] = _tensors(out)
    for p in procs:
        p.join(timeout=60)
    for r, out in res.items():
        assert "error" not in out, "rank %d:\n%s" % (r, out.get("error"))
    return res


def _close(a, b, what, tol=2e-6):
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
    assert err <= tol, "%s: relative difference %.3e" % (what, err)


@pytest.mark.parametrize("group,n_gen", [(4, 2), (3, 2)])
def test_role_split_equals_single_process(group, n_gen):
    """2 generator ranks (2 frames each per chunk) + 2 discriminator ranks (image D | FlowNet + temporal D), and + 1
    discriminator rank doing both: two chunks of one sequence against a single process generating the 4 frames itself."""
    k, n_chunks = 2, 2
    ref, ref_params = _baseline(n_gen, k, n_chunks, 100)
    res = _run_roles(group, n_gen, group, k, n_chunks)
    for r, out in res.items():
        for c in range(n_chunks):
            got, want = out["rec"][c], ref[c]
            assert set(got["losses"]) == set(want["losses"]), (r, c)
            for name, v in want["losses"].items():
                owner_is_T = "_T_" in name
                # every generator rank sees every loss value; a discriminator rank sees the losses it computes (zeros for the other's)
                sees = out["role"] == "G" or (out["owns_DT"] if owner_is_T else out["owns_D"])
                if sees:
                    assert abs(got["losses"][name] - v) <= 2e-6 * max(abs(v), 1e-3), (r, c, name, got["losses"][name], v)
            if out["role"] == "G":
                _close(got["grad_G"], want["grad_G"], "rank %d chunk %d: all-reduced G gradient" % (r, c))
                assert "grad_D" not in got and "grad_DT" not in got
            if out["owns_D"]:
                _close(got["grad_D"], want["grad_D"], "rank %d chunk %d: D gradient" % (r, c))
            if out["owns_DT"] and "grad_DT" in want:
                _close(got["grad_DT"], want["grad_DT"], "rank %d chunk %d: D_T gradient" % (r, c))
        if out["role"] == "G":
            _close(out["param_G"], ref_params["G"], "rank %d: G parameters after %d chunks" % (r, n_chunks))
        if out["owns_D"]:
            _close(out["param_D"], ref_params["D"], "rank %d: D parameters" % r)
        if out["owns_DT"]:
            _close(out["param_DT"], ref_params["DT"], "rank %d: D_T parameters" % r)
        # checkpoints: G by generator rank 0, D by the image rank, D_T by the temporal rank -- once each
        assert out["saved_G"] == (["latest"] if (out["role"] == "G" and out["g"] == 0) else [])
        want_saved = ([("D", "latest")] if out["owns_D"] else []) + ([("D_T0", "latest")] if out["owns_DT"] else [])
        assert out["saved_D"] == want_saved, (r, out["saved_D"])
    assert "grad_DT" in ref[1], "the second chunk must exercise the temporal discriminator"


def test_two_sequence_groups_average_their_gradients():
    """world 4 = 2 sequence groups of (1 generator + 1 discriminator rank) on DIFFERENT sequences: the per-role all-reduce
    gives the mean over the sequence groups (what DataParallel over sequences gives the reference)."""
    k, n_chunks = 3, 1
    refs = [_baseline(1, k, n_chunks, 100 + s)[0] for s in range(2)]
    res = _run_roles(4, 1, 2, k, n_chunks)
    mean = lambda key: (refs[0][0][key] + refs[1][0][key]) / 2
    for r, out in res.items():
        got = out["rec"][0]
        if out["role"] == "G":
            _close(got["grad_G"], mean("grad_G"), "rank %d: G gradient averaged over 2 sequences" % r)
        else:
            _close(got["grad_D"], mean("grad_D"), "rank %d: D gradient averaged over 2 sequences" % r)
            _close(got["grad_DT"], mean("grad_DT"), "rank %d: D_T gradient averaged over 2 sequences" % r)
        assert out["saved_G"] == (["latest"] if (out["role"] == "G" and out["seq"] == 0) else [])
