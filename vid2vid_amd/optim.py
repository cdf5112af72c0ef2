"""Fused Adam over flat parameter / gradient / moment buffers.

Replaces the reference's `torch.optim.Adam` instances (models/vid2vid_model_G.py:84,
models/vid2vid_model_D.py:86-91; stepped 2+T times per chunk by train.py:130-138).  All parameters of
an optimizer are re-homed as views into ONE flat fp32 buffer, their `.grad`s as views into a second
one; the HIP backward kernels accumulate straight into those views (autograd.py), so that

    zero_grad()  = one hipMemsetAsync            (v2v_memset_zero)
    all-reduce   = RCCL on the flat gradient     (parallel.GradSync, bucketed views of the same buffer)
    step()       = one kernel launch             (v2v_adam_step)

instead of ~400 tensors x (7 elementwise passes + launches).  Same update rule and defaults as
torch.optim.Adam (no amsgrad); `param_groups[0]['lr']` is honoured so the reference's
update_learning_rate (models/base_model.py:154-160) keeps working.
"""
import ctypes as C
import os

import torch


class FlatBuffers:
    """Re-homes parameters (and their gradients) into contiguous flat buffers.  Pure tensor-view
    bookkeeping: works on any device (the gloo CPU tests use it as is)."""

    def __init__(self, params, align=64):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatBuffers: no trainable parameters")
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("parameters must be fp32 (bf16 is a storage format of activations / packed weights)")
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        self.numel = total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = offs
        # update counter of THIS buffer: the fused optimizer writes flat_param from a HIP kernel (invisible to torch's
        # tensor version counters), so packed weight copies (engine.PackedConv) watch this box through the parameter
        self.epoch = [0]
        # gradients written NOW will be used by a step: False only between step() and the next zero_grad() of a FusedAdam with
        # `discard_stale_grads` (the backward kernels then leave this buffer's gradients out: autograd.ConvFn.backward)
        self.live = True
        self.ready = None                            # parallel.BucketReady once a GradSync is attached (sync_optimizers)
        # V2V_WEIGHTS_CL=1: 4-D (convolution) weights live CHANNELS-LAST in the flat buffers: physically [d0][KH][KW][d1],
        # logically still [d0][d1][KH][KW] (a permuted view, as torch.channels_last tensors are).  That is the column order the
        # weight-gradient kernel computes in -- an unsplit launch writes its tiles straight into .grad and a split one needs no
        # transpose in its reduce pass -- and the order the forward re-pack reads in contiguous runs.  Measured on the 512x256
        # training chunk it is a wash (+0.9 %: the unsplit weight-gradient launch and the transposed re-pack of the backward-data
        # operator give back what the transpose-free reduce saves; profiles/r02_a64_weights_cl_ab.txt), so it is OFF by default;
        # both layouts are covered by tests/test_gpu_train_ops.py.
        self.channels_last = os.environ.get("V2V_WEIGHTS_CL", "0") == "1"
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self._view(self.flat_param, p, o)
                view.copy_(p.data)
                p.data = view
                p.grad = self._view(self.flat_grad, p, o)
                p._v2v_epoch = self.epoch
                p._v2v_flat = self

    def _view(self, flat, p, o):
        seg = flat[o:o + p.numel()]
        if self.channels_last and p.dim() == 4:
            d0, d1, kh, kw = p.shape
            return seg.view(d0, kh, kw, d1).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def rebind_grads(self):
        """Re-attach `.grad` views (e.g. after a foreign zero_grad(set_to_none=True))."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self._view(self.flat_grad, p, o)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=0.0, grad_sync=None):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdam keeps one flat buffer: pass a single parameter list")
        self.flat = FlatBuffers(self.param_groups[0]["params"])
        self.flat.owner_key = id(self)               # parallel.GradSync files in-flight collectives of this buffer under the optimizer
        self.exp_avg = torch.zeros_like(self.flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_param)
        self.step_count = 0
        self.grad_sync = grad_sync          # parallel.GradSync or None (single process)
        # capturable (graphed.ChunkGraphs): step count and learning rate live in device memory ({int32 step; float lr}) and the
        # update runs as v2v_adam_step_dev, so a step() captured into a hipGraph advances its bias corrections on every replay
        self.capturable = False
        self._dev_state = None
        self._dev_lr = None
        # train.py's protocol is zero_grad() -> backward() -> step() per optimizer (train.py:130-138), and loss_G.backward() also
        # runs through the discriminators: their weight gradients of THAT pass are wiped by optimizer_D.zero_grad() before anything
        # reads them (in the reference too -- torch computes and discards them).  With this flag (models.create_optimizer sets it
        # on the three training optimizers) gradients of this optimizer's parameters are produced only between its zero_grad() and
        # its step() -- and before its first step, when .grad is still what torch would hold.  Off: torch's semantics exactly
        # (gradients accumulate whenever a backward pass reaches the parameter).
        self.discard_stale_grads = False

    def rebuild(self, params, lr=None, betas=None):
        """Re-home a (larger) parameter list in fresh flat buffers IN PLACE: same optimizer object, same grad_sync,
        zeroed moments and step count (what constructing a new Adam would give).  Parameters that lived in the old
        flat buffer are copied over by FlatBuffers (their .data views move to the new one), so the old buffer is simply
        dropped -- nothing keeps stepping memory that no parameter views (update_fixed_params)."""
        params = [p for p in params]
        grp = self.param_groups[0]
        grp["params"] = params
        if lr is not None:
            grp["lr"] = lr
        if betas is not None:
            grp["betas"] = tuple(betas)
        had_ready = self.flat.ready is not None
        self.flat = FlatBuffers(params)
        self.flat.owner_key = id(self)
        if had_ready and self.grad_sync is not None:
            from .parallel import BucketReady
            self.flat.ready = BucketReady(self.flat, self.grad_sync)
        self.exp_avg = torch.zeros_like(self.flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_param)
        self.step_count = 0
        if self.capturable:
            self.make_capturable()
        self.state.clear()

    def make_capturable(self):
        """Move the step count / learning rate to device memory (idempotent).  From here on `step_count` of this object is the
        count of step() CALLS the host has seen; the authoritative count is on the device (`device_step()`), because graph
        replays run updates the host never sees."""
        if self._dev_state is None:
            self._dev_state = torch.zeros(2, dtype=torch.int32, device=self.flat.flat_param.device)
        lr = float(self.param_groups[0]["lr"])
        host = torch.zeros(2, dtype=torch.int32)
        host[0] = int(self.step_count)               # word 0: updates done so far (int32)
        host[1:2].view(torch.float32)[0] = lr        # word 1: learning rate (float bits)
        self._dev_state.copy_(host)
        self._dev_lr = lr
        self.capturable = True

    def sync_hyper(self):
        """Capturable mode: push a learning rate changed on the host (update_learning_rate, models/base_model.py:154-160) to
        the device word the captured update reads.  Cheap when nothing changed; call before replaying a captured step."""
        if self.capturable:
            lr = float(self.param_groups[0]["lr"])
            if lr != self._dev_lr:
                self._dev_state[1:2].view(torch.float32).fill_(lr)
                self._dev_lr = lr

    def device_step(self):
        return int(self._dev_state[0].item()) if self.capturable else self.step_count

    # ---- checkpointing (an extension: the reference never saves optimizer state) ----
    def state_dict(self):
        """Flat moments + step count + hyper-parameters.  The layout is the parameter order of this optimizer
        (FlatBuffers.offsets), so a checkpoint is valid for the same network definition."""
        grp = self.param_groups[0]
        return {"step": self.device_step(), "exp_avg": self.exp_avg.detach().cpu().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(), "numel": self.flat.numel,
                "hyper": {k: grp[k] for k in ("lr", "betas", "eps", "weight_decay")}}

    def load_state_dict(self, state):
        if int(state["numel"]) != self.flat.numel:
            raise ValueError("FusedAdam checkpoint holds %d values, this optimizer %d (different network definition)"
                             % (int(state["numel"]), self.flat.numel))
        self.step_count = int(state["step"])
        with torch.no_grad():
            self.exp_avg.copy_(state["exp_avg"])
            self.exp_avg_sq.copy_(state["exp_avg_sq"])
        for k, v in state.get("hyper", {}).items():
            self.param_groups[0][k] = tuple(v) if k == "betas" else v
        if self.capturable:
            self.make_capturable()                   # re-seed the device words from the restored count / learning rate

    def zero_grad(self, set_to_none=False):
        from .lib import lib, check
        if self.grad_sync is not None:
            self.grad_sync.wait_pending(owner=self)  # only THIS optimizer's overlapped step still reads these gradients
        self.flat.rebind_grads()
        g = self.flat.flat_grad
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if g.is_cuda else None
        if not g.is_cuda and not lib.v2v_get_dry_run():
            raise RuntimeError("FusedAdam runs on the MI355X only")
        check(lib.v2v_memset_zero(C.c_void_p(g.data_ptr()), g.numel() * 4, stream), "memset_zero")
        self.flat.live = True
        if self.flat.ready is not None:
            self.flat.ready.arm()                    # the backward pass that follows writes the gradients of THIS step

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by vid2vid")
        from .lib import lib, check
        f = self.flat
        if not f.flat_param.is_cuda and not lib.v2v_get_dry_run():
            raise RuntimeError("FusedAdam runs on the MI355X only")
        f.rebind_grads()
        grp = self.param_groups[0]
        self.step_count += 1
        b1, b2 = grp["betas"]
        step_no = self.step_count

        def adam(gscale, stream):
            sptr = None
            if f.flat_param.is_cuda:
                sptr = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
            if self.capturable:
                self.sync_hyper()
                check(lib.v2v_adam_step_dev(C.c_void_p(f.flat_param.data_ptr()), C.c_void_p(f.flat_grad.data_ptr()),
                                            C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                            f.numel, float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                                            float(gscale), C.c_void_p(self._dev_state.data_ptr()), sptr), "adam_step_dev")
                return
            check(lib.v2v_adam_step(C.c_void_p(f.flat_param.data_ptr()), C.c_void_p(f.flat_grad.data_ptr()),
                                    C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                    f.numel, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                                    float(grp["weight_decay"]), float(gscale), step_no, sptr), "adam_step")

        gs = self.grad_sync
        if gs is not None and hasattr(gs, "mark_step"):
            gs.mark_step(f.flat_param.device)
        rest = f.ready.finish() if (f.ready is not None and f.ready.armed) else None      # buckets not sent from inside the pass
        overlapped = False
        if gs is not None and f.flat_param.is_cuda and (gs.world > 1 or gs.force_collective):
            overlapped = True                                # Adam runs on GradSync's side stream: packings follow lazily, behind wait_pending
            gs.run_overlapped(f.flat_grad, adam, owner=self, rest=rest)     # RCCL all-reduce + Adam on the side stream, behind this backward pass
        elif gs is not None and rest is not None:
            for b in rest:
                gs.late_buckets += 1
                gs.reduce_bucket(b)
            adam(gs.grad_scale(), None)
        else:
            adam(gs.all_reduce(f.flat_grad) if gs is not None else 1.0, None)
        f.epoch[0] += 1                                      # packed copies of THESE parameters are now stale
        if self.discard_stale_grads:
            f.live = False                                   # until the next zero_grad(): gradients written in between would be wiped unread
        if f.flat_param.is_cuda and not overlapped:
            from .engine import repack_after_step
            repack_after_step(f)                             # ... and are re-packed beside the next backward pass (side stream)
        return None
