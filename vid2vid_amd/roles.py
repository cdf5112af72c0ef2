"""Generator / discriminator rank roles: the MI355X-native form of the reference's `n_gpus_gen` split.

What the reference does (models/models.py:10-23, models/vid2vid_model_G.py:126-133,152-166,189-194,
models/vid2vid_model_D.py:19-22; `--gpu_ids 0,..,7 --n_gpus_gen 6`, batchSize 1): ONE process; G is replicated onto
`n_gpus_gen` GPUs with `torch.nn.parallel.replicate` on every forward call, GPU 1 + t // n_frames_per_gpu generates frame
t of the chunk (frame t needs the fake frames t-1, t-2 of every scale: the GPUs run one after the other, tensors hop
with `.cuda(gpu_id)`), all results are copied back to GPU 0, and modelD / flowNet are `nn.DataParallel` over GPU 0 and the
remaining GPUs.  A chunk is `n_gpus_gen * n_frames_per_gpu` frames of ONE sequence.

Here: one process per GPU, persistent parameters, RCCL point-to-point over xGMI (`dist.send / recv`; gloo in the CPU
tests).  A *sequence group* of `group_size = len(opt.gpu_ids)` consecutive ranks works on one sequence:

    G-rank g (g < n_gpus_gen)   owns G, generates frames [g k, (g+1) k) of the chunk (k = n_frames_per_gpu): receives the
                                last tG-1 fake frames of every scale from G-rank g-1 (detached: n_frames_bp = 1,
                                vid2vid_model_G.py:167-168), sends its own tail to g+1, sends its frames' fake_B / fake_B_raw /
                                flow / weight to the D-ranks.  The last G-rank hands the chunk's tail back to G-rank 0 for the
                                next chunk (train.py:59-61).
    D-rank 0                    owns the image discriminator netD (+ VGG): Vid2VidModelD.forward(0, ...) on ALL frames of the
                                chunk -- the same batch, the same batch-norm statistics as a single process.
    D-rank 1 (if present)       owns FlowNet2 and the temporal discriminators netD_T*: flowNet(...) (sent to D-rank 0) and
                                Vid2VidModelD.forward(s + 1, ...).  With one D-rank, it does both.

Backward: `loss_G.backward()` on a D-rank ends in `_Remote.backward`, which sends d loss / d (fake_B, fake_B_raw, flow,
weight) of each G-rank's frames back to it; on a G-rank it ends in `_GradSink.backward`, which receives those tensors
(from every D-rank, summed) and continues into the rank's own generator graph.  Frames of a chunk do not exchange
gradients (the previous frames are detached), so the G-ranks' backward passes run concurrently.  `optimizer_G.step()`
all-reduces the flat G gradient over the G-ranks (SUM over the frames of the chunk, mean over sequence groups);
netD / netD_T* live on one rank per sequence group and are all-reduced over the sequence groups only.

train.py runs UNCHANGED on every rank: the three wrappers below stand where models.RankModel stands and make
`modelG(...)`, `flowNet(...)`, `modelD(...)`, `loss.backward()` and `optimizer.step()` do the role's share.  Losses that
a rank does not compute arrive as values (for logging) attached to a zero-gradient graph, so every `backward()` call of
train.py:86-93 is legal on every rank.

Every point-to-point transfer below happens in a fixed program order on both sides (sends are `isend`, completed before
the buffers are reused), so there is no tag matching and no deadlock by construction.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class RoleLayout:
    """Who is who.  rank -> (sequence group, role, index); process groups for the per-role gradient all-reduces."""

    def __init__(self, rank, world, group_size, n_gen):
        if world % group_size != 0:
            raise ValueError("world size %d is not a multiple of the sequence-group size %d" % (world, group_size))
        n_disc = group_size - n_gen
        if n_gen < 1 or n_disc not in (1, 2):
            raise ValueError("role split needs 1 or 2 discriminator ranks per sequence group "
                             "(len(gpu_ids) - n_gpus_gen), got %d generator + %d discriminator ranks" % (n_gen, n_disc))
        self.rank, self.world, self.group_size, self.n_gen, self.n_disc = rank, world, group_size, n_gen, n_disc
        self.n_groups = world // group_size
        self.seq, self.idx = rank // group_size, rank % group_size
        self.role = "G" if self.idx < n_gen else "D"
        self.g = self.idx if self.role == "G" else -1
        self.d = self.idx - n_gen if self.role == "D" else -1
        base = self.seq * group_size
        self.g_ranks = [base + i for i in range(n_gen)]
        self.d_ranks = [base + n_gen + j for j in range(n_disc)]
        self.d_image = self.d_ranks[0]             # owner of netD (+ VGG)
        self.d_temporal = self.d_ranks[-1]         # owner of FlowNet2 + netD_T*
        # process groups (created by every rank, in the same order): all G-ranks of all sequence groups; per D index
        # the ranks holding that index in every sequence group
        self.pg_G = self.pg_D = None
        if dist.is_initialized() and world > 1:
            all_g = [s * group_size + i for s in range(self.n_groups) for i in range(n_gen)]
            pg = dist.new_group(all_g)
            if self.role == "G":
                self.pg_G = pg
            for j in range(n_disc):
                ranks = [s * group_size + n_gen + j for s in range(self.n_groups)]
                pg = dist.new_group(ranks)
                if self.role == "D" and self.d == j:
                    self.pg_D = pg

    @property
    def owns_G(self): return self.role == "G"
    @property
    def owns_D(self): return self.rank == self.d_image
    @property
    def owns_DT(self): return self.rank == self.d_temporal
    @property
    def saves(self): return self.seq == 0            # checkpoint writer: the owning rank of sequence group 0


def layout_from_opt(opt, rank=None, world=None):
    """The reference's flags decide: `--gpu_ids` lists the GPUs of ONE sequence group, `--n_gpus_gen` how many of them
    generate.  No split (plain data parallelism over sequences, models.RankModel) when n_gpus_gen covers the group."""
    if not dist.is_initialized():
        return None
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    group_size = int(getattr(opt, "role_group_size", 0) or len(opt.gpu_ids))
    n_gen = opt.n_gpus_gen
    if group_size <= 1 or n_gen < 0 or n_gen >= group_size or opt.batchSize != 1 or not opt.isTrain:
        return None
    if int(getattr(opt, "max_frames_backpropagate", 1)) > 1:
        # the reference lets gradients cross the generator GPUs when n_frames_bp > 1 (`.cuda(gpu_id)` keeps the autograd edge,
        # models/vid2vid_model_G.py:152-168); here the previous frames arrive detached from the generator rank before, so a
        # larger n_frames_bp would silently train something else
        raise ValueError("the generator / discriminator rank roles support --max_frames_backpropagate 1 only (previous frames cross "
                         "generator ranks detached); got %d" % opt.max_frames_backpropagate)
    return RoleLayout(rank, world, group_size, n_gen)


# ---------------------------------------------------------------------------------------------------------------
# point-to-point plumbing
# ---------------------------------------------------------------------------------------------------------------
class _Sends:
    """Outstanding isends of one phase: the tensors stay referenced until wait()."""

    def __init__(self):
        self.work = []

    def send(self, t, dst):
        t = t.detach().contiguous()
        if _staged(t):
            t = t.cpu()                              # gloo: CPU transport (see parallel.host_staged); RCCL sends device memory
        self.work.append((dist.isend(t, dst), t))

    def wait(self):
        for w, _ in self.work:
            w.wait()
        self.work = []


def _staged(t_or_device):
    dev = t_or_device.device if torch.is_tensor(t_or_device) else t_or_device
    return bool(dev is not None and torch.device(dev).type == "cuda" and dist.get_backend() == "gloo")


def _recv(shape, src, device, dtype=torch.float32):
    if _staged(device):
        t = torch.empty(shape, dtype=dtype)
        dist.recv(t, src)
        return t.to(device)
    t = torch.empty(shape, dtype=dtype, device=device)
    dist.recv(t, src)
    return t


class _GradSink(torch.autograd.Function):
    """G-rank: own frames' outputs -> chunk-sized tensors (zeros in the other ranks' frames); backward RECEIVES the
    gradients of the own frames from every D-rank, sums them and hands them to the generator graph."""

    @staticmethod
    def forward(ctx, layout, t0, t1, n_load, *outs):
        ctx.layout, ctx.t0, ctx.t1 = layout, t0, t1
        ctx.shapes = [tuple(o.shape) for o in outs]
        full = []
        for o in outs:
            f = o.new_zeros((o.shape[0], n_load) + tuple(o.shape[2:]))
            f[:, t0:t1] = o
            full.append(f)
        return tuple(full)

    @staticmethod
    def backward(ctx, *local):
        grads = None
        dev = local[0].device if local[0] is not None else None
        for d in ctx.layout.d_ranks:
            got = [_recv(s, d, dev) for s in ctx.shapes]
            grads = got if grads is None else [a + b for a, b in zip(grads, got)]
        return (None, None, None, None) + tuple(grads)


class _Remote(torch.autograd.Function):
    """D-rank: the tensors received from the G-ranks enter the graph here; backward SENDS each G-rank the gradient of
    its frames."""

    @staticmethod
    def forward(ctx, layout, k, anchor, *full):
        ctx.layout, ctx.k = layout, k
        ctx.shapes = [tuple(f.shape) for f in full]
        return tuple(f.clone() for f in full)

    @staticmethod
    def backward(ctx, *grads):
        sends = _Sends()
        k = ctx.k
        for gi, g_rank in enumerate(ctx.layout.g_ranks):
            for gr, shp in zip(grads, ctx.shapes):
                if gr is None:
                    raise RuntimeError("role split: a generator output received no gradient on a discriminator rank")
                sends.send(gr[:, gi * k:(gi + 1) * k], g_rank)
        sends.wait()
        return (None, None, None) + (None,) * len(grads)


class NullOptimizer:
    """Stands in for the optimizers of networks this rank does not own (train.py steps all of them on every rank)."""

    def __init__(self, lr=0.0):
        self.param_groups = [{"lr": lr, "params": []}]
        self.grad_sync = None

    def zero_grad(self, set_to_none=False): pass
    def step(self, closure=None): return None
    def state_dict(self): return {}
    def load_state_dict(self, state): pass

    def rebuild(self, params, lr=None, betas=None):
        """FusedAdam.rebuild's signature (BaseModel.update_fixed_params runs on EVERY rank, also on the ranks that do not own
        the generator): nothing to re-home here, only the learning rate is kept for update_learning_rate's print-out."""
        if lr is not None:
            self.param_groups[0]["lr"] = lr


# ---------------------------------------------------------------------------------------------------------------
# the three wrappers (drop-in for models.RankModel)
# ---------------------------------------------------------------------------------------------------------------
class _RoleState:
    """Per-chunk state shared by the three wrappers of a rank."""

    def __init__(self, layout):
        self.layout = layout
        self.flow_calls = 0
        self.anchor = None           # D-rank: tensors whose zero-weighted sum keeps a loss attached to _Remote
        self.sink = None             # G-rank: ditto for _GradSink
        self.fake_tail = None        # D-rank: last tG-1 fake frames (finest scale) of the previous chunk


def _broadcast_state(nets, group, src):
    """Every tensor of the networks' state_dicts (parameters outside the optimizer's flat buffer -- the coarse scales while
    `niter_fix_global` holds them fixed -- and the norm layers' running statistics) from `src` to the ranks of `group`:
    replicas of one network must start identical whatever each process's RNG did."""
    if group is None:
        return
    with torch.no_grad():
        for net in nets:
            for t in net.state_dict().values():
                if not torch.is_tensor(t) or t.numel() == 0:
                    continue
                buf = t.detach().contiguous()
                if _staged(buf):
                    buf = buf.cpu()
                dist.broadcast(buf, src=src, group=group)
                if buf.data_ptr() != t.data_ptr():
                    t.copy_(buf)


class RoleModelG(nn.Module):
    def __init__(self, opt, model, layout, state):
        super().__init__()
        self.opt, self.module, self.layout, self.state = opt, model, layout, state
        self._dummy = nn.Parameter(torch.zeros(1, device=getattr(model, "device", None)))

    def forward(self, input_A, input_B, inst_A, fake_B_prev_last):
        from . import parallel
        parallel.wait_pending()
        L, m, st = self.layout, self.module, self.state
        st.flow_calls = 0
        tG, k, S = self.opt.n_frames_G, m.n_frames_per_gpu, m.n_scales
        n_load = m.n_frames_load
        dev = self._dummy.device
        _, _, _, H, W = input_B.shape
        has_flow = not getattr(self.opt, "no_flow", False)
        shapes = [(1, k, self.opt.output_nc, H, W), (1, k, self.opt.output_nc, H, W)] + ([(1, k, 2, H, W), (1, k, 1, H, W)] if has_flow else [])
        pyr_shapes = [(1, tG - 1, self.opt.output_nc, H >> s, W >> s) for s in range(S)]
        if getattr(m, "n_frames_bp", 1) > 1:
            raise RuntimeError("role split: n_frames_bp = %d > 1 (update_training_batch with --max_frames_backpropagate > 1) would need "
                               "gradients to cross the generator ranks; not supported" % m.n_frames_bp)
        if L.role == "G":
            g = L.g
            if g > 0:                                   # the frames just before mine were generated by G-rank g-1
                prev = [_recv(s, L.g_ranks[g - 1], dev) for s in pyr_shapes]
            else:
                prev = fake_B_prev_last
            outs = m(input_A, input_B, inst_A, prev, frame_range=(g * k, (g + 1) * k), first_chunk=fake_B_prev_last is None)
            fake_B, fake_B_raw, flow, weight, real_A, real_Bp, tail = outs
            sends = _Sends()
            if L.n_gen > 1:
                nxt = L.g_ranks[g + 1] if g + 1 < L.n_gen else L.g_ranks[0]
                for t in tail:
                    sends.send(t, nxt)
            mine = [fake_B, fake_B_raw] + ([flow, weight] if has_flow else [])
            for d in L.d_ranks:
                for t in mine:
                    sends.send(t, d)
            if g == 0 and L.n_gen > 1:                  # the chunk's tail, for the next chunk (train.py:59-61)
                tail = [_recv(s, L.g_ranks[-1], dev) for s in pyr_shapes]
            sends.wait()
            full = _GradSink.apply(L, g * k, (g + 1) * k, n_load, *mine)
            st.sink = full[0]
            flow_f, weight_f = (full[2], full[3]) if has_flow else (None, None)
            return full[0], full[1], flow_f, weight_f, real_A, real_Bp, tail
        # ---- discriminator rank: no generator work; the encoded labels / real frames it needs are inputs ----
        with torch.no_grad():
            real_A_all, real_B_all, _ = m.encode_input(input_A, input_B, inst_A)
        recvd = [[_recv(s, g_rank, dev) for s in shapes] for g_rank in L.g_ranks]
        full = [torch.cat([r[i] for r in recvd], dim=1) for i in range(len(shapes))]
        full = _Remote.apply(L, k, self._dummy, *full)
        st.anchor = full
        hist = full[0].detach() if st.fake_tail is None or fake_B_prev_last is None else torch.cat([st.fake_tail, full[0].detach()], 1)
        st.fake_tail = hist[:, -(tG - 1):]
        tail = [st.fake_tail]
        flow_f, weight_f = (full[2], full[3]) if has_flow else (None, None)
        return full[0], full[1], flow_f, weight_f, real_A_all[:, tG - 1:], real_B_all[:, tG - 2:], tail


class RoleFlowNet(nn.Module):
    def __init__(self, opt, model, layout, state):
        super().__init__()
        self.opt, self.module, self.layout, self.state = opt, model, layout, state

    def forward(self, input_A, input_B):
        L, st = self.layout, self.state
        primary = st.flow_calls == 0                    # train.py:57, the chunk's own frames; later calls: get_skipped_flows
        st.flow_calls += 1
        b, n, _, h, w = input_A.shape
        dev = input_A.device
        zeros = lambda: (torch.zeros(b, n, 2, h, w, device=dev), torch.zeros(b, n, 1, h, w, device=dev))
        if L.role == "G":
            return zeros()
        if L.owns_DT:
            flow, conf = self.module(input_A, input_B)
            if primary and L.d_image != L.rank:
                s = _Sends()
                s.send(flow, L.d_image); s.send(conf, L.d_image)
                s.wait()
            return flow, conf
        if primary:                                     # image-discriminator rank: the flow losses need the reference flow
            return _recv((b, n, 2, h, w), L.d_temporal, dev), _recv((b, n, 1, h, w), L.d_temporal, dev)
        return zeros()


class RoleModelD(nn.Module):
    def __init__(self, opt, model, layout, state):
        super().__init__()
        self.opt, self.module, self.layout, self.state = opt, model, layout, state
        self._dummy = nn.Parameter(torch.zeros(1, device=getattr(model, "device", None)))

    def forward(self, scale_T, tensors_list):
        from . import parallel
        parallel.wait_pending()
        L, st, m = self.layout, self.state, self.module
        names = m.loss_names if scale_T == 0 else m.loss_names_T
        owner = L.d_image if scale_T == 0 else L.d_temporal
        dev = self._dummy.device
        fake = tensors_list[1]                          # fake_B (scale 0) / skipped fake frames (scale > 0): the graph's root on this rank

        def attach(values, graph_root):
            # loss values on a zero-gradient graph: G-type losses hang on the generator tensors (their backward() must reach
            # _GradSink / _Remote), D-type losses on a private dummy parameter
            zg = graph_root.sum() * 0.0 if graph_root is not None and graph_root.requires_grad else self._dummy.sum() * 0.0
            zd = self._dummy.sum() * 0.0
            return [(v + (zd if n.startswith("D_") else zg)).view(1, 1) for n, v in zip(names, values)]

        if L.rank == owner:
            losses = m(scale_T, tensors_list)
            vals = torch.stack([l.detach().reshape(()) for l in losses])
            s = _Sends()
            for g_rank in L.g_ranks:
                s.send(vals, g_rank)
            s.wait()
            if st.anchor is not None:                   # keep loss_G attached to EVERY generator output (zero weight): _Remote.backward
                z = sum(a.sum() for a in st.anchor) * 0.0   # then always runs and always has a gradient for each of them
                losses = [l + z if not n.startswith("D_") else l for n, l in zip(names, losses)]
            return losses
        if L.role == "G":
            vals = _recv((len(names),), owner, dev)
            return attach([vals[i] for i in range(len(names))], st.sink if st.sink is not None else fake)
        # the other discriminator rank: zeros, still attached to the generator tensors so that its _Remote.backward runs
        root = None
        if st.anchor is not None:
            root = torch.cat([a.reshape(-1)[:1] for a in st.anchor])
        return attach([torch.zeros((), device=dev) for _ in names], root)


def wrap_roles(opt, modelG, modelD, flowNet, layout):
    """models.wrap_model in role mode: the three wrappers + per-role optimizers / checkpoint writers."""
    from . import parallel
    st = _RoleState(layout)
    L = layout
    lr = opt.lr
    # optimizers of networks this rank does not own become no-ops; the owners all-reduce over their role's process group
    if L.owns_G:
        gs = parallel.GradSync(L.pg_G, scale=1.0 / L.n_groups)       # SUM over the chunk's frames, mean over sequence groups
        modelG.optimizer_G.grad_sync = gs
        if L.pg_G is not None:
            gs.broadcast(modelG.optimizer_G.flat.flat_param, src=0)
            # ... and everything the flat buffer does not hold: the coarse scales while niter_fix_global keeps them out of the
            # optimizer, and every norm layer's running statistics
            _broadcast_state([getattr(modelG, "netG%d" % s_) for s_ in range(getattr(modelG, "n_scales", 1)) if hasattr(modelG, "netG%d" % s_)],
                             L.pg_G, 0)
        parallel._ACTIVE_SYNCS.append(gs)
    else:
        modelG.optimizer_G = NullOptimizer(lr)
    t_scales = opt.n_scales_temporal
    gsd = parallel.GradSync(L.pg_D, scale=1.0 / L.n_groups) if L.role == "D" else None
    if gsd is not None:
        parallel._ACTIVE_SYNCS.append(gsd)
    if L.owns_D:
        modelD.optimizer_D.grad_sync = gsd
        if L.pg_D is not None:
            gsd.broadcast(modelD.optimizer_D.flat.flat_param, src=L.n_gen + 0)
            if hasattr(modelD, "netD"):
                _broadcast_state([modelD.netD], L.pg_D, L.n_gen + 0)
    else:
        modelD.optimizer_D = NullOptimizer(lr)
    for s in range(t_scales):
        name = "optimizer_D_T%d" % s
        if L.owns_DT:
            getattr(modelD, name).grad_sync = gsd
            if L.pg_D is not None:
                gsd.broadcast(getattr(modelD, name).flat.flat_param, src=L.n_gen + L.n_disc - 1)
                if hasattr(modelD, "netD_T%d" % s):
                    _broadcast_state([getattr(modelD, "netD_T%d" % s)], L.pg_D, L.n_gen + L.n_disc - 1)
        else:
            setattr(modelD, name, NullOptimizer(lr))
    # checkpoints: each network is written by its owner in sequence group 0 (the other replicas are never updated)
    save_G, save_net = modelG.save, modelD.save_network

    def save_g(label):
        if L.owns_G and L.g == 0 and L.saves:
            save_G(label)

    def save_d(label):
        if L.owns_D and L.saves:
            save_net(modelD.netD, "D", label, modelD.gpu_ids)
        if L.owns_DT and L.saves:
            for s in range(t_scales):
                save_net(getattr(modelD, "netD_T" + str(s)), "D_T" + str(s), label, modelD.gpu_ids)
    modelG.save, modelD.save = save_g, save_d
    return RoleModelG(opt, modelG, L, st), RoleModelD(opt, modelD, L, st), RoleFlowNet(opt, flowNet, L, st)
