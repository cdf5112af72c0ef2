"""In-backward gradient buckets (parallel.BucketReady, SURVEY 8e "bucketed, reverse order, overlapped with backward"):
world-2 gloo on CPU.  The HIP backward kernels write parameter gradients straight into flat-buffer views, so autograd never
sees them; readiness is counted per bucket by the gradient-writing nodes themselves.  Checked here:
  * the gradients after step() equal the end-of-pass all-reduce bit for bit (two frames sharing the parameters: a bucket is
    complete only after the LAST frame's backward reached it), with most buckets sent from inside the pass;
  * buckets leave in reverse parameter order, the same order on every rank;
  * a gradient-writing node that was not counted is caught (RuntimeError), not silently dropped from the sum;
  * the real autograd.ConvFn reports its parameters (dry-run backend: launches argument-checked, nothing executed).
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _DepositFn(torch.autograd.Function):
    """Stand-in for autograd.ConvFn: the parameter gradient is deposited in place by the node's backward (as the HIP weight-
    gradient kernel does) and None is returned for it; use / done are reported exactly as ConvFn reports them."""

    @staticmethod
    def forward(ctx, x, w, tag, counted):
        from vid2vid_amd.parallel import note_use
        ctx.w, ctx.tag, ctx.counted = w, tag, counted
        if counted:
            note_use([w])
        return x + 0.0

    @staticmethod
    def backward(ctx, dy):
        from vid2vid_amd.parallel import note_done
        w = ctx.w
        w.grad.add_(torch.full_like(w.grad, ctx.tag))                 # "kernel" accumulating into the flat view
        if ctx.counted:
            note_done([w])
        return dy, None, None, None


def _chunk(ws, rank, frames, uncounted=None):
    """`frames` frames through all layers; frame t feeds frame t+1 (as fake_B_prev does): backward visits the last frame first."""
    x = torch.zeros(1, requires_grad=True)
    for t in range(frames):
        for i, w in enumerate(ws):
            x = _DepositFn.apply(x, w, float((rank + 1) * (i + 1) * (t + 1)), uncounted != (t, i))
    return x.sum()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from vid2vid_amd import networks as N, parallel
    from vid2vid_amd.optim import FusedAdam
    N.set_record_only(True)                                           # dry-run backend: memset / Adam launches are argument-checked only
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [40, 7, 130, 64, 3, 90, 33]                           # bucket = 64 values: parameters straddle buckets, buckets hold several
        results = {}
        for mode in (True, False):
            torch.manual_seed(5)
            ws = [nn.Parameter(torch.randn(n)) for n in sizes]
            opt = FusedAdam(ws)
            gs = parallel.sync_optimizers([opt], bucket_bytes=64 * 4, in_backward=mode)
            assert opt.flat.ready is not None and opt.flat.ready.n == (opt.flat.numel + 63) // 64 >= 6
            order = []
            send = gs.send_ready_bucket

            def spy(b, key=None, _send=send, _flat=opt.flat.flat_grad):
                order.append((b.data_ptr() - _flat.data_ptr()) // 4 // 64)
                _send(b, key)
            gs.send_ready_bucket = spy
            loss = _chunk(ws, rank, frames=2)
            assert sum(opt.flat.ready.open) >= 2 * len(ws)            # every node registered its buckets
            opt.zero_grad()
            opt.flat.flat_grad.zero_()                                # (dry run: the memset launch did not execute)
            loss.backward()
            n_b = opt.flat.ready.n
            if mode:
                # reverse PARAMETER order (a parameter that spans several buckets completes them together, lowest first):
                # the last layer's bucket leaves first, the first layer's last
                # ... and since round 6 in ONE fixed order, descending bucket index, whatever the graph
                assert order == list(range(n_b - 1, -1, -1)) and gs.early_buckets == n_b, (order, n_b)
                spans = [opt.flat.ready.span[id(w)] for w in ws]     # per parameter: its buckets leave no later than those of the parameter in front of it
                last_sent = [max(order.index(k) for k in sp) for sp in spans]
                assert all(last_sent[i] >= last_sent[i + 1] for i in range(len(ws) - 1)), (order, last_sent)
            else:
                assert order == [] and gs.early_buckets == 0
            opt.step()
            assert gs.early_buckets + gs.late_buckets == n_b and not opt.flat.ready.armed and sum(opt.flat.ready.open) == 0
            results[mode] = opt.flat.flat_grad.clone()
            for i, w in enumerate(ws):                                # sum over ranks and frames of what the nodes deposited
                expect = sum((rk + 1) * (i + 1) * (t + 1) for rk in range(world) for t in range(2))
                assert torch.equal(w.grad, torch.full_like(w.grad, float(expect))), (mode, i)
            all_orders = [None] * world
            dist.all_gather_object(all_orders, order)
            assert all(o == all_orders[0] for o in all_orders)        # collectives issued in the same order on every rank
            parallel._ACTIVE_SYNCS.clear()
        assert torch.equal(results[True], results[False])             # bit-identical to the end-of-pass all-reduce

        # ---- rank-divergent graphs: rank 1 holds a counted node that its backward never reaches (a branch
        # the loss does not depend on), so ITS bucket of layer 3 never completes inside the pass while rank 0 completes all of
        # them.  The issue order must still be the same on both ranks -- descending, the held bucket and everything below it
        # at step() -- and the sums right; before the fix rank 0 sent all buckets early and rank 1 paired other buckets with them.
        torch.manual_seed(5)
        ws = [nn.Parameter(torch.randn(n)) for n in sizes]
        opt = FusedAdam(ws)
        gs = parallel.sync_optimizers([opt], bucket_bytes=64 * 4, in_backward=True)
        issued = []
        reduce_ = gs.reduce_bucket

        def spy_reduce(b, _r=reduce_, _flat=opt.flat.flat_grad):
            issued.append((b.data_ptr() - _flat.data_ptr()) // 4 // 64)
            _r(b)
        gs.reduce_bucket = spy_reduce
        loss = _chunk(ws, rank, frames=2)
        if rank == 1:
            _DepositFn.apply(torch.zeros(1, requires_grad=True), ws[3], 99.0, True)      # registered, never walked
        opt.zero_grad()
        opt.flat.flat_grad.zero_()
        loss.backward()
        n_b = opt.flat.ready.n
        held = min(opt.flat.ready.span[id(ws[3])])
        if rank == 0:
            assert gs.early_buckets == n_b
        else:
            assert gs.early_buckets == n_b - 1 - max(opt.flat.ready.span[id(ws[3])]) < n_b, (gs.early_buckets, held)
        opt.step()
        assert issued == list(range(n_b - 1, -1, -1)), issued            # every rank: the same order, whatever was early or late
        for i, w in enumerate(ws):
            expect = sum((rk + 1) * (i + 1) * (t + 1) for rk in range(world) for t in range(2))
            assert torch.equal(w.grad, torch.full_like(w.grad, float(expect))), ("divergent", i)
        parallel._ACTIVE_SYNCS.clear()

        # an uncounted gradient-writing node: its bucket leaves early, the late contribution must raise
        ws = [nn.Parameter(torch.randn(n)) for n in sizes]
        opt = FusedAdam(ws)
        gs = parallel.sync_optimizers([opt], bucket_bytes=64 * 4, in_backward=True)
        loss = _chunk(ws, rank, frames=2, uncounted=(0, len(ws) - 1))       # frame 0 (backward: last) of the last layer is not counted
        opt.zero_grad()
        raised = False
        try:
            loss.backward()
            # the uncounted node itself reports nothing; the next counted node of that bucket does not exist either, so the
            # check that fires is the one on buckets shared with a counted parameter -- force it through `done`
            opt.flat.ready.done(ws[-1])
        except RuntimeError as ex:
            raised = "already" in str(ex) or "after its bucket" in str(ex)
        assert raised
        opt.step()                                                    # the runtime recovers: counters cleared
        parallel._ACTIVE_SYNCS.clear()

        # ---- the real ConvFn (dry-run backend): a two-layer stack run for two "frames" ----
        from vid2vid_amd.engine import Engine
        from vid2vid_amd import autograd as AG, lib as L
        eng = Engine(torch.device("cpu"), L.F32, record_only=True)
        convs = [nn.Conv2d(8, 8, 3, padding=1), nn.Conv2d(8, 8, 3, padding=1)]
        norms = [nn.BatchNorm2d(8), nn.BatchNorm2d(8)]
        params = [p for m in convs + norms for p in m.parameters()]
        opt = FusedAdam(params)
        gs = parallel.sync_optimizers([opt], bucket_bytes=256 * 4, in_backward=True)
        x = eng.pack(torch.randn(1, 8, 16, 16))
        h = x
        for t in range(2):
            for c, n_ in zip(convs, norms):
                h = AG.conv_group(eng, h, c, L.PAD_REFLECT, None, n_, L.ACT_RELU, 0.0, None, None, False, 1.0, "test.conv")
        assert sum(opt.flat.ready.open) > 0
        opt.zero_grad()
        h.t.float().sum().backward()
        assert sum(opt.flat.ready.open) == 0 and gs.early_buckets == opt.flat.ready.n      # every bucket left from inside the pass
        opt.step()
        assert gs.late_buckets == 0
        dist.barrier()
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_in_backward_buckets_equal_the_end_of_pass_all_reduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(0, "ok"), (1, "ok")]
