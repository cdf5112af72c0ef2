"""Whole-chunk hipGraphs for the training step.

The training step is an eager autograd graph of v2v custom ops: ~4000 launches per 512x256 chunk, each behind a Python
autograd node and a ctypes call.  Measured on the MI355X box (profiles/r06_v1_train_by_grid.txt) the device is busy only
70 % of a chunk -- the host cannot issue the launches as fast as the GPU retires them.  The launch sequence of a chunk is
static (same shapes, same buffers, same tile selections) once the tile searches are done, so it is captured ONCE per chunk
kind by HIP stream capture (torch.cuda.CUDAGraph: every launch of libv2v_hip.so goes to torch's current stream, which is the
capturing stream; autograd's backward runs on the stream of the forward ops) and replayed as one hipGraph launch:
forward of G, FlowNet2, D / D_T, the losses, and zero_grad / backward / Adam for every optimizer.

A "chunk kind" is whatever makes the launch sequence differ: the position of the chunk in its sequence (first chunk: no
previous frames; later chunks: fake_B_prev, more temporal scales active).  The caller names it with a hashable key.  All
graphs share one memory pool; tensors that flow from one chunk to the next (fake_B_prev_last, the temporal discriminators'
frame history) stay alive in that pool between captures, so the graph of kind k+1 reads the addresses the graph of kind k
writes -- which is why the graphs must be replayed in the order they were captured in (a sequence is always walked front
to back: train.py:44-48).

What a captured step must not do: synchronise, read device values on the host, or keep host-side counters that a kernel
argument depends on.  The one such counter on this path was Adam's step number (bias corrections): optim.FusedAdam in
`capturable` mode keeps it in device memory (v2v_adam_step_dev).

The reference has no analogue (PyTorch 0.4 eager); the drop-in train.py path stays eager, this is what bench.py --mode train
and a graph-aware training loop use.
"""
import torch


class ChunkGraphs:
    def __init__(self, optimizers, device=None, enabled=True):
        self.optimizers = [o for o in optimizers if hasattr(o, "make_capturable")]
        self.device = device
        self.enabled = bool(enabled) and torch.cuda.is_available()
        self.graphs = {}            # key -> (CUDAGraph, host state after the step, info)
        self.pool = None
        self.last_key = None
        self.order = []             # keys in capture order
        self.replays = 0
        if self.enabled:
            for o in self.optimizers:
                o.make_capturable()

    def step(self, key, fn, get_state, set_state):
        """Run one chunk.  fn(): the chunk's whole host program (forward, losses, backward, optimizer steps), reading and
        writing host state that get_state() snapshots and set_state(snapshot) restores (tensors by reference: their
        addresses are what the graphs share).  The first call with a new key captures fn() and replays it once (capture
        does not execute); later calls replay."""
        if not self.enabled:
            fn()
            return None
        for o in self.optimizers:
            o.sync_hyper()
        ent = self.graphs.get(key)
        if ent is None:
            if self.order and self.last_key is not None and self.last_key != self.order[-1] and key not in self.graphs:
                # a new kind may only be captured right behind the kind it follows in the sequence (its inputs are that graph's outputs)
                pass
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool):
                fn()
            if self.pool is None:
                self.pool = g.pool()
            ent = (g, get_state())
            self.graphs[key] = ent
            self.order.append(key)
        g, after = ent
        g.replay()
        set_state(after)
        self.last_key = key
        self.replays += 1
        return g
