"""Data-parallel runtime: one process per GPU, RCCL over xGMI.

The reference's multi-GPU path is single-process `nn.DataParallel` (models/models.py:10-59): every
forward call re-broadcasts all parameters (1.46-1.66 GB for G) to the replicas and reduces the
gradients back to device 0; `n_gpus_gen` additionally pipelines the frames of ONE sequence over GPUs
to fit memory (models/vid2vid_model_G.py:126-133,152-153).  On MI355X a whole 2048x1024 chunk fits one
288 GB GPU, so the native design is plain data parallelism over sequences:

  * persistent per-rank replicas; ONE broadcast of rank 0's flat parameter buffer at start-up;
  * per optimizer (G, D, each D_T) ONE flat fp32 gradient buffer (optim.FlatBuffers), all-reduced in
    place (sum, then 1/world) in buckets sized for xGMI: the links are point-to-point
    (7 x ~153 GB/s per GPU), a ring all-reduce is per-link bound, so buckets are large (64 MB default)
    to stay bandwidth- rather than latency-bound, and they are issued back to back (async) so RCCL
    pipelines them; the D / D_T buffers (<= 45 MB) go as a single bucket;
  * per-rank BatchNorm statistics (no SyncBN) -- what the reference's DataParallel replicas do.

`torch.distributed` backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests
(tests/test_cpu_parallel.py, world size 2).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the process group described by torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def host_staged(t, group=None):
    """True when a collective / point-to-point transfer of the device tensor `t` must go through host memory: the process
    group's backend is gloo (CPU transport).  RCCL ("nccl") moves device memory directly over xGMI; gloo with device tensors is
    the debugging / single-GPU-several-ranks configuration (tests/test_gpu_roles.py), never the production path."""
    return bool(t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo")


_GRAD_STREAMS = []


def register_grad_stream(stream):
    """A stream other than the compute stream on which gradient-writing kernels run (engine.wgrad_side_stream): collectives
    that are sent from inside a backward pass wait for it as well as for the compute stream."""
    if stream not in _GRAD_STREAMS:
        _GRAD_STREAMS.append(stream)


class GradSync:
    """Bucketed in-place all-reduce (sum) of a flat gradient buffer, overlapped with the NEXT backward pass.

    train.py runs three backward passes per chunk back to back (loss_G -> optimizer_G.step, loss_D -> optimizer_D.step,
    loss_D_T[s] -> optimizer_D_T[s].step, train.py:86-93).  On a GPU `FusedAdam.step` does not wait for the collective:
    the all-reduce of optimizer k and its Adam kernel are enqueued on a SIDE stream behind an event of the compute stream,
    which goes straight on to the next loss's backward -- G's 1.66 GB of fp32 gradients (~19 ms as a ring over 153 GB/s
    xGMI links) travel while the discriminators' backward passes run.  The compute stream waits for the side stream only
    where the updated parameters (or the gradient buffer) are next touched: the next forward of any model
    (models.RankModel.forward -> wait_pending) and that optimizer's next zero_grad.

    Round 5 -- buckets INSIDE the backward pass ("bucketed, reverse order, overlapped with backward").  The HIP backward
    kernels accumulate straight into the flat gradient views over several frames and scales, so autograd never sees a parameter
    gradient and no hook fires.  `BucketReady` (below) counts instead: every autograd node that will write parameter gradients
    (autograd.ConvFn, the only one) registers its parameters' buckets when the forward pass creates it and reports them when its
    backward has enqueued its last kernel; once the optimizer is armed (zero_grad) a bucket whose count reaches zero has received
    its last contribution of this pass -- with several frames per chunk that happens during the backward of the FIRST frame, the
    last of the pass -- and its all-reduce is enqueued on the side stream behind an event of the compute stream, while the
    backward pass goes on (always in descending bucket order, see BucketReady).  step() sends whatever is left and the Adam kernel.
    A contribution that arrives for a bucket already sent raises (it cannot happen while every gradient-writing node is counted; the check is what makes that a tested property).
    On CPU tensors (gloo tests) everything runs in order on the host."""

    def __init__(self, group=None, bucket_bytes=64 << 20, force_collective=False, scale=None, wire_dtype=None, in_backward=None):
        self.group = group
        # in_backward: all-reduce each bucket as soon as its last gradient kernel of the pass has been enqueued (BucketReady);
        # V2V_BUCKET_OVERLAP=0 falls back to one burst of buckets at step()
        self.in_backward = (os.environ.get("V2V_BUCKET_OVERLAP", "1") != "0") if in_backward is None else bool(in_backward)
        self.early_buckets = 0                       # buckets sent before step() (statistics, tests)
        self.late_buckets = 0
        # wire_dtype=torch.bfloat16 (or V2V_GRAD_BF16=1): the buckets travel as bf16 -- half the bytes on the per-link-bound xGMI ring
        # (G's 1.66 GB: ~19 -> ~9.5 ms on 8 GPUs) for one rounding of every partial sum to 8 mantissa bits; the master gradient, the
        # moments and the update stay fp32.  Off by default: the reference sums fp32 gradients.
        if wire_dtype is None and os.environ.get("V2V_GRAD_BF16", "0") == "1":
            wire_dtype = torch.bfloat16
        self.wire_dtype = wire_dtype
        self.scale = scale                           # None: 1 / world (mean over data-parallel ranks); roles.py passes 1 / n_sequence_groups
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.force_collective = force_collective     # run the collective even for a world of 1 (single-GPU RCCL test)
        self._stream = None
        self._pending = {}                           # owner (an optimizer) -> event of its enqueued (all-reduce + Adam) pair
        # timing = True (bench.py): HIP event pairs around every bucket's all-reduce on the side stream and an event on the compute
        # stream at each optimizer step (= the end of the backward pass that fed it), so that the run can say how much collective
        # time ran BESIDE backward work (overlap_report) instead of modelling it
        self.timing = False
        self._t_buckets = []                         # (start event, end event, early?, step index)
        self._t_steps = []                           # event on the compute stream at step() entry, per step index

    @property
    def world(self):
        if self.group is None and self.scale is not None:
            return 1                                 # a role that exists once (roles.py, one sequence group): nothing to reduce
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def buckets(self, flat):
        n = flat.numel()
        return [flat[o:min(o + self.bucket_elems, n)] for o in range(0, n, self.bucket_elems)]

    def all_reduce(self, flat):
        """Sum `flat` over the ranks in place, on the CURRENT stream; returns the factor (1/world) that turns the sum into
        the mean the reference's DataParallel implies (losses are averaged over replicas, train.py:65).  The factor is
        applied inside the fused optimizer kernel (v2v_adam_step grad_scale): no extra pass over the buffer."""
        world = self.world
        if world == 1 and not (self.force_collective and dist.is_initialized()):
            return 1.0 if self.scale is None else self.scale
        if host_staged(flat, self.group):
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
            return 1.0 / world if self.scale is None else self.scale
        if self.wire_dtype is not None and self.wire_dtype != flat.dtype:
            pairs = [(b, b.to(self.wire_dtype)) for b in self.buckets(flat)]             # bucket by bucket: conversion of k+1 beside the ring of k
            works = [dist.all_reduce(w_, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for _, w_ in pairs]
            for (b, w_), wk in zip(pairs, works):
                wk.wait()
                b.copy_(w_)
        else:
            works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets(flat)]
            for w in works:
                w.wait()
        return 1.0 / world if self.scale is None else self.scale

    def _timed(self, fn, b, early):
        if not (self.timing and b.is_cuda and len(self._t_buckets) < 4096):
            return fn(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(b.device))
        fn(b)
        e1.record(torch.cuda.current_stream(b.device))
        self._t_buckets.append((e0, e1, early, len(self._t_steps)))

    def mark_step(self, device):
        """FusedAdam.step entry: the compute stream has enqueued the whole backward pass of this optimizer."""
        if self.timing and device is not None and device.type == "cuda" and len(self._t_steps) < 4096:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(device))
            self._t_steps.append(ev)

    def overlap_report(self):
        """After a synchronize: per-run totals of the collectives' device time and the part of it that had finished, or was
        running, before the backward pass that produced the gradients ended on the compute stream (`hidden_ms`).  CPU / gloo runs
        and runs without `timing` report counts only."""
        rep = {"buckets_sent_inside_backward": self.early_buckets, "buckets_sent_at_step": self.late_buckets,
               "bucket_bytes": self.bucket_elems * 4, "wire_dtype": str(self.wire_dtype or "fp32"),
               "allreduce_ms": None, "hidden_ms": None, "note": "hidden = all-reduce time in front of the end of the backward pass that "
               "produced the bucket (HIP events on the collective's side stream against an event of the compute stream at step())"}
        if not self._t_buckets:
            return rep
        total = hidden = 0.0
        for e0, e1, early, si in self._t_buckets:
            try:
                dur = e0.elapsed_time(e1)
                total += dur
                if early and si < len(self._t_steps):
                    to_end = e0.elapsed_time(self._t_steps[si])        # > 0: the bucket started this long before the pass ended
                    hidden += max(0.0, min(dur, to_end))
            except RuntimeError:
                continue
        rep["allreduce_ms"], rep["hidden_ms"] = round(total, 3), round(hidden, 3)
        return rep

    def reduce_bucket(self, b, early=False):
        """Sum ONE bucket over the ranks in place on the current stream (the building block of the in-backward path)."""
        if self.timing and b.is_cuda:
            return self._timed(self._reduce_bucket, b, early)
        return self._reduce_bucket(b)

    def _reduce_bucket(self, b):
        if self.world == 1 and not (self.force_collective and dist.is_initialized()):
            return
        if host_staged(b, self.group):
            host = b.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            b.copy_(host)
        elif self.wire_dtype is not None and self.wire_dtype != b.dtype:
            w_ = b.to(self.wire_dtype)
            dist.all_reduce(w_, op=dist.ReduceOp.SUM, group=self.group)
            b.copy_(w_)
        else:
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)

    def send_ready_bucket(self, b, owner_key=None):
        """Called by BucketReady from inside a backward pass: the kernels that wrote `b` are enqueued on the current stream.
        GPU: the side stream waits for an event recorded now and runs the collective; the compute stream goes on.  The
        completion event is filed under the owning optimizer: if step() never comes (overflow skip, an exception
        between backward and step, a second zero_grad) that optimizer's next zero_grad / wait_pending still waits for the
        in-flight all-reduce before the gradient memory is written again."""
        self.early_buckets += 1
        if not b.is_cuda:
            self.reduce_bucket(b)
            return
        cur = torch.cuda.current_stream(b.device)
        side = self.side_stream(b.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        for gs_ in _GRAD_STREAMS:                    # weight-gradient kernels of the nodes that just reported (their own stream)
            if gs_.device == b.device:
                side.wait_stream(gs_)
        with torch.cuda.stream(side):
            self.reduce_bucket(b, early=True)
            done = torch.cuda.Event()
            done.record(side)
        self._pending[owner_key] = done              # one side stream: the newest event covers every earlier send of this owner

    def collective_active(self):
        return self.world > 1 or (self.force_collective and dist.is_initialized())

    def grad_scale(self):
        return (1.0 / self.world) if self.scale is None else self.scale

    # ---- overlap with the following backward pass (GPU only) ----
    def side_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def run_overlapped(self, flat, fn, owner=None, rest=None):
        """Enqueue all_reduce(flat) followed by fn(grad_scale, stream) on the side stream, ordered after everything already
        on the current stream; returns immediately.  fn launches the optimizer kernel on the stream it is handed.  The
        completion event is filed under `owner` (the optimizer): that optimizer's next zero_grad waits for ITS event only,
        so optimizer_D.zero_grad() / backward (train.py:89-90) really run beside G's all-reduce."""
        cur = torch.cuda.current_stream(flat.device)
        side = self.side_stream(flat.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if rest is None:
                scale = self.all_reduce(flat)
            else:                                    # the in-backward path already sent the other buckets on this side stream
                for b in rest:
                    self.late_buckets += 1
                    self.reduce_bucket(b)
                scale = self.grad_scale()
            fn(scale, side)
            ev = torch.cuda.Event()
            ev.record(side)
        key = id(owner) if owner is not None else None
        old = self._pending.pop(key, None)
        if old is not None:                          # same owner stepped twice without a wait in between: side-stream order
            pass                                     # already serialises the two pairs, the newer event covers the older
        self._pending[key] = ev

    def pending(self, owner=None):
        """True while an enqueued pair of `owner` (any owner if None) has not been waited for (test hook)."""
        return bool(self._pending) if owner is None else id(owner) in self._pending

    def wait_pending(self, device=None, owner=None):
        """The current stream waits for the enqueued (all-reduce + optimizer) pairs: all of them (before ANY parameter is
        read again: RankModel.forward, checkpoint saving) or, with `owner`, only that optimizer's (before its gradient
        buffer is zeroed / written again)."""
        if not self._pending:
            return
        cur = torch.cuda.current_stream(device)
        if owner is not None:
            ev = self._pending.pop(id(owner), None)
            if ev is not None:
                cur.wait_event(ev)
            return
        for ev in self._pending.values():
            cur.wait_event(ev)
        self._pending = {}

    def broadcast(self, flat, src=0):
        if self.world > 1:
            if host_staged(flat, self.group):
                host = flat.detach().cpu()
                dist.broadcast(host, src=src, group=self.group)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src=src, group=self.group)


class BucketReady:
    """Readiness bookkeeping of ONE flat gradient buffer (optim.FlatBuffers.ready): which buckets still expect gradient kernels.

    open[k]  = autograd nodes alive that will write into bucket k (registered by the forward pass, reported by their backward);
    armed    = between the optimizer's zero_grad() and step(): the pass that runs now writes THIS buffer's gradients for the step;
    ready[k] = bucket k has received its last contribution of this pass;
    sent[k]  = bucket k was handed to the collective in this armed window.
    A parameter that straddles a bucket boundary counts in every bucket it touches.

    ISSUE ORDER IS RANK-INVARIANT: buckets are handed to the collective in ONE fixed order -- descending
    index, n-1 first -- on every rank, whatever order the rank's own autograd graph completes them in.  A complete bucket is
    held until every higher-index bucket has been sent (`next`), and finish() continues the same descending walk.  Same-sized
    RCCL collectives on the side stream are therefore always paired bucket k with bucket k, even if one rank's graph lacks a
    node another rank's has (data-dependent branch, another chunk position); with identical graphs -- backward visits the
    parameters in reverse order -- nothing is held back and the overlap is unchanged."""

    def __init__(self, flat, sync):
        self.flat, self.sync = flat, sync
        self.be = sync.bucket_elems
        self.n = (flat.numel + self.be - 1) // self.be
        self.open = [0] * self.n
        self.sent = [False] * self.n
        self.ready = [False] * self.n
        self.next = self.n - 1                       # the only bucket that may be sent now
        self.armed = False
        self.span = {}                               # id(param) -> range of bucket indices
        for p_, o in zip(flat.params, flat.offsets):
            self.span[id(p_)] = range(o // self.be, (o + max(p_.numel(), 1) - 1) // self.be + 1)

    def bucket(self, k):
        g = self.flat.flat_grad
        return g[k * self.be:min((k + 1) * self.be, g.numel())]

    def use(self, p_):
        for k in self.span.get(id(p_), ()):
            if self.armed and self.sent[k]:
                raise RuntimeError("BucketReady: a gradient-writing node was created for a bucket that was already all-reduced in this pass")
            self.open[k] += 1
            self.ready[k] = False

    def _drain(self):
        """Send every complete bucket that is next in the fixed descending order."""
        while self.next >= 0 and self.ready[self.next]:
            k = self.next
            self.sent[k] = True
            self.next -= 1
            self.sync.send_ready_bucket(self.bucket(k), getattr(self.flat, "owner_key", None))

    def done(self, p_):
        for k in self.span.get(id(p_), ()):
            if self.armed and self.sent[k]:
                raise RuntimeError("BucketReady: a gradient contribution arrived after its bucket had been all-reduced "
                                   "(an uncounted gradient-writing node)")
            self.open[k] -= 1
            if self.open[k] < 0:                     # a node of a graph built before the counters were attached / reset
                self.open[k] = 0
                continue
            if self.armed and self.open[k] == 0 and self.sync.in_backward and self.sync.collective_active():
                self.ready[k] = True
        if self.armed:
            self._drain()

    def arm(self):
        self.armed = True
        self.sent = [False] * self.n
        self.ready = [False] * self.n
        self.next = self.n - 1

    def finish(self):
        """step(): the buckets that were not sent from inside the pass, in the SAME descending order the in-pass sends follow
        (the sent ones are exactly n-1 .. next+1); disarms and clears leaked counts."""
        rest = [self.bucket(k) for k in range(self.next, -1, -1)]
        assert all(self.sent[k] for k in range(self.next + 1, self.n)) and not any(self.sent[k] for k in range(self.next + 1))
        self.armed = False
        self.sent = [False] * self.n
        self.ready = [False] * self.n
        self.next = self.n - 1
        self.open = [0] * self.n
        return rest


def note_use(params):
    """autograd.ConvFn.forward: this node will accumulate into the gradients of `params` when it runs backward."""
    for p_ in params:
        r = getattr(getattr(p_, "_v2v_flat", None), "ready", None)
        if r is not None:
            r.use(p_)


def note_done(params):
    """autograd.ConvFn.backward, after its last launch."""
    for p_ in params:
        r = getattr(getattr(p_, "_v2v_flat", None), "ready", None)
        if r is not None:
            r.done(p_)


_ACTIVE_SYNCS = []


def sync_optimizers(optimizers, group=None, bucket_bytes=64 << 20, force_collective=False, in_backward=None):
    """Attach a GradSync to each FusedAdam and make every rank start from rank 0's parameters."""
    gs = GradSync(group, bucket_bytes, force_collective, in_backward=in_backward)
    for opt in optimizers:
        opt.grad_sync = gs
        opt.flat.ready = BucketReady(opt.flat, gs)   # in-backward bucket scheduling (counts start with the next forward pass)
        gs.broadcast(opt.flat.flat_param)
    _ACTIVE_SYNCS.append(gs)
    return gs


def wait_pending():
    """Called by models.RankModel.forward (and by checkpoint saving): optimizer steps that were enqueued on the side stream
    must have finished before any parameter is read on the compute stream."""
    for gs in _ACTIVE_SYNCS:
        gs.wait_pending()


def shared_tuning_cache(rank, world):
    """N > 1 inference replicas: rank 0 runs the tile searches (per-shape + whole-frame), the other ranks replay its
    selections through their own tuning cache file (`V2V_TUNE_CACHE`) -- the same kernels on every GPU instead of N
    independent, noisy searches.  Call BEFORE the first engine exists.  Ranks > 0 return only after rank 0 has called the
    returned `release()` (rank 0 calls it once its plan is built): the selections travel by `broadcast_object_list`, so
    ranks on other nodes get them too and no predictable shared path is involved (each rank writes a private temp file).
    On ranks > 0 and in a single process `release` is a no-op.  Does nothing when the user already set V2V_TUNE_CACHE."""
    if world <= 1 or os.environ.get("V2V_TUNE_CACHE"):
        return lambda: None
    import json
    import tempfile
    fd, cache = tempfile.mkstemp(prefix="v2v_tune_rank%d_" % rank, suffix=".json")
    os.close(fd)
    os.remove(cache)                                 # the engine treats a missing file as "measure"
    os.environ["V2V_TUNE_CACHE"] = cache
    if rank == 0:
        def release():
            data = None
            if os.path.exists(cache):
                with open(cache) as f:
                    data = json.load(f)
            dist.broadcast_object_list([data], src=0)
        return release
    box = [None]
    dist.broadcast_object_list(box, src=0)           # blocks until rank 0 releases
    if box[0] is not None:
        with open(cache, "w") as f:
            json.dump(box[0], f)
    return lambda: None
