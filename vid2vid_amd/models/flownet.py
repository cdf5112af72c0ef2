"""FlowNet wrapper on the MI355X backend (drop-in for the reference's models/flownet.py).

`flowNet(im_t, im_prev) -> (flow (.., 2, H, W), conf (.., 1, H, W))`, 4-D or 5-D inputs, frozen and
gradient-free exactly like the reference (:25-41).  The whole computation -- resize to multiples of
64, FlowNet2, confidence = (||im1 - warp(im2, flow)||^2 < 0.02), resize back (:43-59) -- is recorded
once per input shape and replayed as one hipGraph.
"""
import os

import torch

from ..engine import Plan, _ptr, _stream
from ..flownet2 import FlowNet2
from ..lib import lib, check
from .base_model import BaseModel


class _FlowPlan:
    def __init__(self, model, B, H, W, use_graph=True):
        eng = model.engine
        self.eng, self.model = eng, model
        dev = eng.device
        self.im1 = torch.zeros(B, 3, H, W, dtype=torch.float32, device=dev)
        self.im2 = torch.zeros(B, 3, H, W, dtype=torch.float32, device=dev)
        self.flow = self.conf = None
        # the engine is shared with the generator / discriminators of the same precision: FlowNet2's launches get their own scratch
        # set (split-K slabs, ticket words, statistics) so that the plan may run on its own stream beside them (FlowNet.forward)
        # Round 6: FlowNetSD beside the FlowNetC -> FlowNetS -> FlowNetS chain, and the deconvolution of every refinement level beside
        # its flow head, as parallel plan lanes (hipGraph branches; flownet2.FlowNet2.emit).  V2V_FLOWNET_LANES=0: one lane (A/B).
        prev_lanes = eng.lanes_enabled
        eng.lanes_enabled = os.environ.get("V2V_FLOWNET_LANES", "1") != "0" and not eng.record_only
        try:
            if not eng.record_only:
                prev_autotune = eng.autotune
                eng.autotune = bool(getattr(model.opt, "autotune", True))
                try:
                    with eng.scratch_set("flownet2"):
                        self._emit()               # sizes the scratch (of every lane), picks tile configurations
                finally:
                    eng.autotune = prev_autotune
                torch.cuda.synchronize(dev)
            self.plan = Plan()
            eng.plan = self.plan
            n0 = len(eng.conv_log)
            try:
                with self.plan, eng.scratch_set("flownet2"):
                    self._emit()
            finally:
                eng.plan = None
        finally:
            eng.lanes_enabled = prev_lanes
        self.conv_flops = sum(c["flops"] for c in eng.conv_log[n0:])     # algorithmic FLOP of one replay (bench roofline)
        self.n_convs = len(eng.conv_log) - n0
        if use_graph and not eng.record_only:
            self.plan.instantiate_graph()

    def _resize(self, x, OH, OW, scale=1.0):
        B, Cc, H, W = x.shape
        out = self.eng.empty_f32(B, Cc, OH, OW)
        check(lib.v2v_resize_planar(_ptr(x), _ptr(out), B * Cc, H, W, OH, OW, 1, float(scale), _stream()), "resize_planar")
        self.eng.label("resize_planar")
        return out

    def _emit(self):
        eng = self.eng
        im1, im2 = self.im1, self.im2
        B, _, H, W = im1.shape
        nh, nw = H // 64 * 64, W // 64 * 64
        if nh == 0 or nw == 0:
            raise ValueError("FlowNet2 needs images of at least 64x64")
        resized = (nh != H)                    # the reference tests the height only (flownet.py:48)
        if resized:
            im1, im2 = self._resize(im1, nh, nw), self._resize(im2, nh, nw)
        flow = self.model.flowNet.emit(eng, im1, im2)
        conf = eng.empty_f32(B, 1, im1.shape[2], im1.shape[3])
        check(lib.v2v_warp_diff_norm(_ptr(im1), 3 * im1.shape[2] * im1.shape[3], _ptr(im2),
                                     3 * im1.shape[2] * im1.shape[3], _ptr(flow), None, _ptr(conf),
                                     B, 3, im1.shape[2], im1.shape[3], 1, 0.02, _stream()), "confidence")
        eng.label("flow_confidence")
        if resized:
            flow = self._resize(flow, H, W, float(H) / float(nh))     # both components scaled by old_h/new_h (:57)
            conf = self._resize(conf, H, W)
        self.flow, self.conf = flow, conf


class FlowNet(BaseModel):
    def name(self):
        return "FlowNet"

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.flowNet = FlowNet2().to(self.device)
        path = getattr(opt, "flownet2_checkpoint", "models/flownet2_pytorch/FlowNet2_checkpoint.pth.tar")
        if os.path.isfile(path):
            ck = torch.load(path, map_location="cpu")
            self.flowNet.load_state_dict(ck["state_dict"] if "state_dict" in ck else ck)
        elif not getattr(opt, "random_init_ok", False):
            raise RuntimeError("%s not found (FlowNet2 weights are an external download)" % path)
        for p in self.flowNet.parameters():
            p.requires_grad_(False)
        self._plans = {}
        self._side = None
        self.side_stream_on = os.environ.get("V2V_FLOWNET_STREAM", "0") == "1"      # measured: no gain once the weight gradients share the chip (profiles/r06_v11_trainab.txt): opt-in
        self.flops_launched, self.convs_launched = 0.0, 0
        self.bind_precision()

    def forward(self, input_A, input_B, dummy_bs=0):
        with torch.no_grad():
            if dummy_bs:
                input_A, input_B = input_A[dummy_bs:], input_B[dummy_bs:]
            size = input_A.size()
            assert len(size) in (4, 5)
            if len(size) == 5:
                b, n, c, h, w = size
                flow, conf = self.compute_flow_and_conf(input_A.reshape(-1, c, h, w), input_B.reshape(-1, c, h, w))
                return flow.view(b, n, 2, h, w), conf.view(b, n, 1, h, w)
            return self.compute_flow_and_conf(input_A, input_B)

    def compute_flow_and_conf(self, im1, im2):
        assert im1.size(1) == 3 and im1.size() == im2.size()
        B, _, H, W = im1.shape
        key = (B, H, W, self.precision)
        fp = self._plans.get(key)
        if fp is None:
            self.engine.refresh_weights()
            fp = _FlowPlan(self, B, H, W, use_graph=getattr(self.opt, "use_graph", True))
            self._plans[key] = fp
        if self.engine.record_only or self.device.type != "cuda" or not self.side_stream_on:
            fp.im1.copy_(im1.to(self.device, torch.float32))
            fp.im2.copy_(im2.to(self.device, torch.float32))
            if not self.engine.record_only:
                self._launch(fp)
            return fp.flow.clone(), fp.conf.clone()
        # Round 6: FlowNet2 on its OWN stream.  It is frozen and depends only on the real frames, but train.py calls it between
        # modelG(...) and modelD(...): on one stream its 117 small convolutions per pair (3.7 % of the matrix peak) queue behind
        # the whole generator forward pass.  Here the side stream waits only for the event that says the frames are on the device
        # (networks.note_inputs_ready, recorded by Vid2VidModelG.forward) -- or, for inputs of unknown origin, for everything on
        # the current stream -- and the current stream waits for FlowNet2's results: the generator's queued launches and FlowNet2
        # share the chip.
        from .. import networks
        cur = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        im1d, im2d = im1.to(self.device, torch.float32), im2.to(self.device, torch.float32)
        evs = [networks.inputs_ready_event(t) for t in (im1d, im2d)]
        if all(e is not None for e in evs) and im1d is not None and (im1d.data_ptr() == im1.data_ptr()) and (im2d.data_ptr() == im2.data_ptr()):
            for e in evs:
                side.wait_event(e)
        else:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            fp.im1.copy_(im1d)
            fp.im2.copy_(im2d)
            self._launch(fp)
            flow, conf = fp.flow.clone(), fp.conf.clone()
            done = torch.cuda.Event()
            done.record(side)
        cur.wait_event(done)
        for t in (im1d, im2d):
            t.record_stream(side)
        for t in (flow, conf):
            t.record_stream(cur)
        return flow, conf

    def _launch(self, fp):
        if torch.cuda.is_current_stream_capturing():
            fp.plan.run()                      # inside a stream capture (graphed.ChunkGraphs): the launches themselves join the capture
        else:
            fp.plan.launch()
        self.flops_launched += fp.conv_flops
        self.convs_launched += fp.n_convs
