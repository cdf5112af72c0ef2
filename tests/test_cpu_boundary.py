"""CPU-side checks of the drop-in boundary: C ABI surface, checkpoint-key compatibility, the
lowering (plan recording, no execution) and the 'oracle is test infrastructure' rule."""
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT
from util import sd_from_npz


def test_library_exports_every_declared_symbol():
    from vid2vid_amd import lib
    header = open(os.path.join(ROOT, "include", "v2v_hip.h")).read()
    declared = set(re.findall(r"\b(v2v_[a-z0-9_]+)\s*\(", header)) - {"v2v_conv_desc", "v2v_plan"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib.lib, name), "libv2v_hip.so does not export %s" % name
    assert declared <= set(lib.exported_symbols()), sorted(declared - set(lib.exported_symbols()))
    assert lib.lib.v2v_version() >= 100


def test_conv_desc_layout_matches_header():
    """ctypes mirror of struct v2v_conv_desc: field order and count follow the header."""
    from vid2vid_amd.lib import ConvDesc
    header = open(os.path.join(ROOT, "include", "v2v_hip.h")).read()
    body = header[header.index("typedef struct v2v_conv_desc {"):header.index("} v2v_conv_desc;")]
    names = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip().rstrip(";")
        if not line:
            continue
        decl = line.split(None, 1)[1] if not line.startswith("const") else line.split("*", 1)[1]
        names += [n.strip().lstrip("*").strip() for n in decl.replace("*", "").split(",")]
    mine = [n.rstrip("_") for n, _ in ConvDesc._fields_]
    assert mine == names


def test_packed_weight_size_rule():
    from vid2vid_amd import lib
    L = lib.lib
    # Conv2d 3x3 1024->1024 bf16: [1024][9*1024]
    assert L.v2v_conv_packed_elems(1024, 1024, 1024, 3, 3, 0, 1, 1, lib.BF16) == 1024 * 9216
    # ConvTranspose2d 3x3 s2 p1: parity classes have 1,2,2,4 taps
    assert L.v2v_conv_packed_elems(64, 64, 128, 3, 3, 1, 2, 1, lib.F32) == 128 * 64 * (1 + 2 + 2 + 4)
    # K padded to 128 bytes, cout padded to 128 rows
    assert L.v2v_conv_packed_elems(6, 8, 3, 7, 7, 0, 1, 0, lib.BF16) == 128 * 448


def test_correlation_output_size_rule():
    from vid2vid_amd import lib
    import ctypes as C
    c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
    lib.lib.v2v_correlation_out_size(32, 64, 20, 1, 20, 1, 2, C.byref(c), C.byref(h), C.byref(w))
    assert (c.value, h.value, w.value) == (441, 32, 64)          # FlowNetC.py:31
    lib.lib.v2v_correlation_out_size(17, 23, 4, 3, 4, 2, 1, C.byref(c), C.byref(h), C.byref(w))
    assert (c.value, h.value, w.value) == (81, 8, 11)


def test_invalid_arguments_fail_loudly():
    from vid2vid_amd import lib
    with pytest.raises(RuntimeError):
        lib.check(lib.lib.v2v_channelnorm_forward(None, None, 1, 1, 1, 1, 2, None), "channelnorm")
    d = lib.ConvDesc()
    assert lib.lib.v2v_conv2d(d, None) != 0
    assert b"null" in lib.lib.v2v_last_error()


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+\.*oracle\b|oracle\.vid2vid_oracle|/oracle/", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vid2vid_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), "%s references the oracle" % f


def _opt(**kw):
    d = dict(fp16=False, n_blocks=2, n_blocks_local=1, n_local_enhancers=1, fg=True, no_flow=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_state_dict_keys_match_reference_checkpoints(golden):
    """Strict load of reference-produced state_dicts = same parameter names and shapes."""
    from vid2vid_amd import networks as N
    g = golden("composite_fg_32x64")
    net = N.define_G(108, 3, 6, 8, "composite", 3, "batch", 0, [], _opt())
    net.load_state_dict(sd_from_npz(g, "sd."), strict=True)
    g = golden("composite_local_32x64")
    N.define_G(12, 3, 6, 8, "composite", 2, "batch", 0, [], _opt(fg=False)).load_state_dict(sd_from_npz(g, "sd0."), strict=True)
    N.define_G(12, 3, 6, 4, "compositeLocal", 2, "batch", 1, [], _opt(fg=False)).load_state_dict(sd_from_npz(g, "sd1."), strict=True)
    g = golden("multiscale_d_64x96")
    N.define_D(13, 8, 3, "batch", 2, True, []).load_state_dict(sd_from_npz(g, "sd."), strict=True)
    g = golden("first_frame_nets_32x64")
    N.define_G(11, 3, 0, 8, "global", 2, "instance", 0, [], _opt()).load_state_dict(sd_from_npz(g, "sdg."), strict=True)
    N.define_G(11, 3, 0, 4, "local", 2, "instance", 0, [], _opt()).load_state_dict(sd_from_npz(g, "sdl."), strict=True)


def test_seeded_init_equals_reference_init(golden):
    """Same construction order + same initialiser => identical weights under the same seed
    (the golden state_dict was drawn by the reference under manual_seed(11))."""
    from vid2vid_amd import networks as N
    g = golden("composite_fg_32x64")
    torch.manual_seed(11)
    net = N.define_G(108, 3, 6, 8, "composite", 3, "batch", 0, [], _opt())
    ref = sd_from_npz(g, "sd.")
    for k, v in net.state_dict().items():
        if "running_" in k or "num_batches" in k:
            continue                         # buffers were updated by the reference's forward passes
        if k == "model_final_flow.1.weight":
            v = v * 0.1                      # make_golden.py scales the flow head after init
        assert torch.allclose(v.float(), ref[k].float(), rtol=0, atol=1e-7), k


def test_inference_lowering_census_512x256():
    """Record (not run) the per-frame plan of BASELINE config C2 and compare the conv census with
    SURVEY.md Appendix A.1: 79 convolutions, 2115.0 GFLOP, 411.3 M parameters."""
    from vid2vid_amd import networks as N
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True,
                       precision="bf16", gpu_ids=[])
        m = create_model(opt)
        assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 411.3) < 0.1
        H, W = 256, 512
        A = torch.randint(0, 35, (1, 3, 1, H, W)).float()
        inst = torch.randint(0, 20, (1, 3, 1, H, W)).float()
        fake, lab = m.inference(A, torch.zeros(1, 2, 3, H, W), inst)
        assert fake.shape == (1, 3, H, W) and lab.shape == (36, H, W)
        fp = m._active_plan
        assert sum(c.get("convs", 1) for c in fp.conv_log) == 79        # (model_final_flow + model_final_w are one launch)
        assert abs(sum(c["flops"] for c in fp.conv_log) / 1e9 - 2115.0) < 0.5
        assert 150 < fp.plan.num_ops <= 204          # 200 ops (launches + lane edges) after the fused norm / in-kernel stem finalize; was 236
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_flownet2_checkpoint_keys_match_reference(golden):
    """vid2vid_amd.flownet2.FlowNet2 exposes exactly the reference's state_dict keys / shapes, so
    FlowNet2_checkpoint.pth.tar loads by name (models/flownet.py:19-20)."""
    from vid2vid_amd.flownet2 import FlowNet2
    g = golden("flownet2_64x128")
    ref = {k: tuple(int(d) for d in s.split(",")) for k, s in zip(g["keys"], g["shapes"])}
    with torch.device("meta"):
        net = FlowNet2()
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == ref
    assert sum(int(np.prod(s)) for s in mine.values()) == 162518834


def test_inference_lowering_census_edge2face_512():
    """Record (not run) the per-frame plan of BASELINE config C4 (edge2face 512x512, input_nc=15, no fg tower) and
    compare with SURVEY.md section 8(a7): 3434.4 GFLOP / frame, 365 M parameters."""
    from vid2vid_amd import networks as N
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(label_nc=0, input_nc=15, use_instance=False, fg=False, use_real_img=True, random_init_ok=True,
                       dataroot="datasets/face/", precision="bf16", gpu_ids=[])
        m = create_model(opt)
        assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 365.0) < 1.0
        H = W = 512
        fake, last = m.inference(torch.rand(1, 3, 15, H, W), torch.zeros(1, 2, 3, H, W), None)
        assert fake.shape == (1, 3, H, W) and last.shape == (15, H, W)
        assert abs(sum(c["flops"] for c in m._active_plan.conv_log) / 1e9 - 3434.4) < 1.0
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def _face_opt(**kw):
    import types
    d = dict(fp16=False, n_blocks=2, n_blocks_local=1, n_local_enhancers=1, fg=True, no_flow=False, feat_num=4)
    d.update(kw)
    return types.SimpleNamespace(**d)


FACE_NETS = (("sdG.", 501, (5, 3, 0, 8, "global_with_features", 2, "instance", 0, [])),
             ("sdL.", 502, (5, 3, 0, 4, "local_with_features", 2, "instance", 0, [])),
             ("sdE.", 503, (3, 4, 0, 4, "encoder", 2, "instance", 0, [])))


def test_feature_encoding_nets_checkpoint_keys_and_seeded_init(golden):
    """Global_with_z / Local_with_z / Encoder: state_dict keys == the reference's (checkpoint format)
    and a seeded construction draws the reference's initial weights (same module creation order)."""
    from vid2vid_amd import networks as N
    g = golden("face_first_frame_nets_32x32")
    for tag, seed, args in FACE_NETS:
        torch.manual_seed(seed)
        net = N.define_G(*args, _face_opt())
        ref = sd_from_npz(g, tag)
        sd = net.state_dict()
        assert set(sd.keys()) == set(ref.keys()), tag
        for k, v in sd.items():
            if "running_" in k or "num_batches" in k:
                continue                     # buffers were updated by the reference's forward pass
            assert torch.allclose(v.float(), ref[k].float(), rtol=0, atol=1e-7), tag + k


def test_feature_encoding_nets_lowering_records(golden):
    """Record (not run) the lowering of Global_with_z / Local_with_z: every launch passes the library's argument checks
    and the head produces the planar (1, 3, H, W) image.  (Numerical parity of this lowering: GPU test, next round.)"""
    from vid2vid_amd import networks as N
    from vid2vid_amd.engine import Plan
    if torch.cuda.is_available():
        pytest.skip("dry-run lowering is a CPU-host check")
    g = golden("face_first_frame_nets_32x32")
    N.set_record_only(True)
    try:
        eng = N.get_engine("cpu")
        x, z = torch.from_numpy(g["in.x"]), torch.from_numpy(g["in.z"])
        for tag, seed, args in FACE_NETS[:2]:
            net = N.define_G(*args, _face_opt())
            net.load_state_dict(sd_from_npz(g, tag))
            plan = Plan()
            eng.plan = plan
            try:
                with plan:
                    out = net.emit(eng, eng.pack(x), eng.pack(z))
            finally:
                eng.plan = None
            assert tuple(out.shape) == (1, 3, 32, 32)
            assert plan.num_ops > 20
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_plan_lanes_record_fork_and_join_edges():
    """Frame plans record the generator's independent towers / branches on lanes (parallel hipGraph paths): the recorded
    op list of the 512x256 label2city frame carries the fork / join edges (lane_wait) and the same 79 convolutions, and a
    linear plan (opt.lanes = 0) carries none."""
    from vid2vid_amd import networks as N
    from vid2vid_amd.lib import lib
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    counts = {}
    for lanes in (1, 0):
        N.set_record_only(True)
        try:
            opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True,
                           precision="bf16", gpu_ids=[], ngf=16, n_blocks=2)
            opt.lanes = lanes
            opt.twin = 0                    # the paired launches of the twin chains have their own test below
            m = create_model(opt)
            H, W = 64, 128
            A = torch.randint(0, 35, (1, 3, 1, H, W)).float()
            m.inference(A, torch.zeros(1, 2, 3, H, W), torch.randint(0, 9, (1, 3, 1, H, W)).float())
            plan = m._active_plan.plan
            names = [lib.v2v_plan_op_name(plan.h, i).decode() for i in range(plan.num_ops)]
            counts[lanes] = (names.count("lane_wait"), names.count("conv_igemm"), names.count("add_nhwc"))
        finally:
            N.set_record_only(False)
            N._ENGINES.clear()
    assert counts[1][0] == 6 and counts[0][0] == 0          # fork seg, fork fg, join seg, fork flow, join flow, join fg
    assert counts[1][1] == counts[0][1]                     # same convolutions either way
    assert counts[1][2] == counts[0][2] + 1                 # the tower sum is its own launch when the towers run in parallel
    assert lib.v2v_plan_set_lane(9) != 0 and lib.v2v_plan_lane_wait(0, 8) != 0 and lib.v2v_plan_set_lane(0) == 0


def test_twin_chains_record_paired_launches():
    """With opt.twin (the default) the ResnetBlock chains of the label / image towers and of the image / flow branches are
    recorded as v2v_conv2d_pair launches with the norm fused in (V2V_OUT_NORM_ACT_NHWC): the 512x256 label2city frame keeps its census of 79
    convolutions / 2115 GFLOP while 36 of them (the 1024 -> 1024 layers) travel as 18 launches, and the library's argument
    checks accept every descriptor pair."""
    from vid2vid_amd import networks as N
    from vid2vid_amd.lib import lib
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    N.set_record_only(True)
    try:
        ops = {}
        for twin in (1, 0):
            opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True,
                           precision="bf16", gpu_ids=[])
            opt.twin = twin
            m = create_model(opt)
            H, W = 256, 512
            A = torch.randint(0, 35, (1, 3, 1, H, W)).float()
            m.inference(A, torch.zeros(1, 2, 3, H, W), torch.randint(0, 20, (1, 3, 1, H, W)).float())
            fp = m._active_plan
            assert sum(c.get("convs", 1) for c in fp.conv_log) == 79 and abs(sum(c["flops"] for c in fp.conv_log) / 1e9 - 2115.0) < 0.5
            names = [lib.v2v_plan_op_name(fp.plan.h, i).decode() for i in range(fp.plan.num_ops)]
            ops[twin] = (names.count("conv_igemm"), names.count("bn_apply"), sum(1 for c in fp.conv_log if c.get("pair")),
                         sum(1 for c in fp.conv_log if c.get("fused_norm")))
            N._ENGINES.clear()
        assert ops[1][2] == 36 and ops[0][2] == 0
        # 36 convolutions in 18 launches; their norm / ReLU / residual passes run inside those launches (fused norm: the
        # 256 workgroups of a pair are all resident on the 256 CUs), so 36 bn_apply launches disappear
        assert ops[1][3] == 36 and ops[0][3] == 0
        assert ops[1][0] == ops[0][0] - 18 and ops[1][1] == ops[0][1] - 36
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_segment_program_of_the_frame_plan_respects_every_edge():
    """The plan executor replays a plan with lanes as one linear hipGraph per lane segment on per-lane streams with event edges
    (csrc/plan.hip, plan_segment_program).  For the recorded 512x256 frame (three lanes) the launch program must keep every lane's
    ops in recording order, put an op into exactly one segment of its own lane, and place every edge behind all earlier ops of
    the signalling lane and in front of all later ops of the waiting lane -- checked without a device."""
    import ctypes as C
    from vid2vid_amd import networks as N
    from vid2vid_amd.lib import lib
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, precision="bf16", gpu_ids=[])
        m = create_model(opt)
        H, W = 256, 512
        m.inference(torch.randint(0, 35, (1, 3, 1, H, W)).float(), torch.zeros(1, 2, 3, H, W), torch.randint(0, 20, (1, 3, 1, H, W)).float())
        plan = m._active_plan.plan
        n = plan.num_ops
        names = [lib.v2v_plan_op_name(plan.h, i).decode() for i in range(n)]
        lanes = [lib.v2v_plan_op_lane(plan.h, i) for i in range(n)]
        steps = (C.c_int32 * (3 * 4 * n))()
        seg_of = (C.c_int32 * n)()
        ns = lib.v2v_plan_segment_program(plan.h, steps, 4 * n, seg_of, n)
        assert ns > 0
        prog = [(steps[3 * i], steps[3 * i + 1], steps[3 * i + 2]) for i in range(ns)]
        waits = [i for i in range(n) if names[i] == "lane_wait"]
        used = sorted({l for i, l in enumerate(lanes) if names[i] != "lane_wait"})
        assert used == [0, 1, 2]
        # every op in exactly one segment of its own lane; lane_wait ops in none
        seg_lane, seg_pos = {}, {}
        for pos, (kind, a, b) in enumerate(prog):
            if kind == 0:
                assert a not in seg_lane
                seg_lane[a], seg_pos[a] = b, pos
        for i in range(n):
            if names[i] == "lane_wait":
                assert seg_of[i] == -1
            else:
                assert seg_lane[seg_of[i]] == lanes[i]
        # per lane: segments are launched in the recording order of their ops
        for l in used:
            ops = [i for i in range(n) if names[i] != "lane_wait" and lanes[i] == l]
            pos = [seg_pos[seg_of[i]] for i in ops]
            assert pos == sorted(pos)
        # edges: the recorded ones in order, then one join of every forked lane into lane 0
        edges = [(pos, a, b) for pos, (kind, a, b) in enumerate(prog) if kind == 1]
        assert len(edges) == len(waits) + len(used) - 1
        assert sorted((a, b) for _, a, b in edges[len(waits):]) == [(0, l) for l in used if l != 0]
        for (pos, a, b), i in zip(edges, waits):
            waiter, signal = lanes[i] & 0xff, lanes[i] >> 8
            assert (a, b) == (waiter, signal)
            for j in range(n):
                if names[j] == "lane_wait":
                    continue
                if lanes[j] == signal and j < i:
                    assert seg_pos[seg_of[j]] < pos          # recorded (hence launched) before the event is recorded
                if lanes[j] == waiter and j > i:
                    assert seg_pos[seg_of[j]] > pos          # launched behind the wait
        assert sum(1 for k, _, _ in prog if k == 0) <= 12       # 9 segment graphs for this frame: a handful of launches, not 120
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_three_scale_narrow_towers_lowering_records(precision):
    """n_scales_spatial = 3 at the widths of tests/golden/inference_label2city_s3_64x128.npz (ngf 8 -> 4 -> 2: towers with
    2, 4 and 6 channels, not multiples of the 4-wide vector loads): every launch of the frame passes the library's argument
    checks (round 1 never recorded this configuration and its norm kernel rejected C = 2 on the GPU)."""
    from vid2vid_amd import networks as N
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    if torch.cuda.is_available():
        pytest.skip("dry-run census is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, ngf=8, n_blocks=2,
                       n_blocks_local=1, n_scales_spatial=3, n_downsample_G=2, loadSize=128, precision=precision, gpu_ids=[])
        m = create_model(opt)
        H, W = 64, 128
        A = torch.randint(0, 35, (1, 3, 1, H, W)).float()
        fake, lab = m.inference(A, torch.zeros(1, 2, 3, H, W), torch.randint(0, 9, (1, 3, 1, H, W)).float())
        assert fake.shape == (1, 3, H, W) and lab.shape == (36, H, W)
        assert m._active_plan.plan.num_ops > 200
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_conv7x7_window_tile_is_validated_by_the_library():
    """Tile 120 (the single-phase patch kernel with a 7x7 window; STAGED for round 5, never run on a GPU yet): in dry-run mode the
    C ABI accepts it for a dense bf16 7x7 / stride 1 / pad 3 Conv2d whose channel stride is a whole number of 128-byte chunks with
    channel-chunk-major weights -- and rejects it for a 3x3 layer, for tap-major weights and for a ragged channel stride."""
    import ctypes as C
    from vid2vid_amd import networks as N
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib, ConvDesc
    if torch.cuda.is_available():
        pytest.skip("dry-run validation is a CPU-host check")
    N.set_record_only(True)
    try:
        buf = torch.zeros(1 << 20)
        def desc(K, cin_stride, korder, tile):
            d = ConvDesc()
            d.in_, d.w, d.out, d.zero_page = buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr()
            d.bias, d.stats = None, None
            d.N, d.H, d.W, d.OH, d.OW = 1, 64, 128, 64, 128
            d.cin, d.cin_stride, d.cout, d.cout_stride = min(cin_stride, 108), cin_stride, 64, 64
            d.KH = d.KW = K
            d.stride, d.pad, d.pad_mode, d.transposed = 1, K // 2, L.PAD_REFLECT, 0
            d.dtype, d.out_mode, d.act, d.act_param, d.out_scale = L.BF16, L.OUT_RAW_F32_NHWC, L.ACT_NONE, 0.0, 1.0
            d.tile, d.splitk, d.prefetch, d.w_korder = tile, 1, 0, korder
            return d
        assert lib.v2v_conv2d(C.byref(desc(7, 128, 1, 120)), None) == 0, lib.v2v_last_error()
        assert lib.v2v_conv_stats_rows(C.byref(desc(7, 128, 1, 120))) == 16 * 4        # 4 x 32 pixel tiles of the 64 x 128 image
        d121 = desc(7, 128, 1, 121); d121.cout = d121.cout_stride = 128
        assert lib.v2v_conv2d(C.byref(d121), None) == 0, lib.v2v_last_error()          # the 128-channel tile
        assert lib.v2v_conv2d(C.byref(desc(3, 128, 1, 120)), None) != 0                # a 3x3 layer
        assert lib.v2v_conv2d(C.byref(desc(7, 128, 0, 120)), None) != 0                # tap-major weights
        assert lib.v2v_conv2d(C.byref(desc(7, 112, 1, 120)), None) != 0                # 112 channels: not whole 128-byte chunks
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_round5_experiment_tiles_are_reachable_through_the_engine():
    """The experiment tiles still in the library (round 6: 143; the round-5 tiles 97-99 / 130-132 / 142 were removed) and the persistent
    tiles 140 / 141: the engine must hand them channel-chunk-major weights (tile_korder) and the library must accept them in dry-run
    mode -- the path the GPU parity tests take.  (Round 5: 97-99 were first missing from engine.is_patch_tile and every launch failed
    on the GPU box.)"""
    import torch.nn as nn
    from vid2vid_amd import networks as N
    from vid2vid_amd import lib as L
    from vid2vid_amd.engine import Engine, EXP_TILES, PATCH_CFGS, is_patch_tile, tile_korder
    if torch.cuda.is_available():
        pytest.skip("dry-run validation is a CPU-host check")
    N.set_record_only(True)
    try:
        eng = Engine(torch.device("cpu"), L.BF16, record_only=True)
        cin = cout = 128
        convs = [nn.Conv2d(cin, cout, 3) for _ in range(2)]
        norms = [nn.BatchNorm2d(cout) for _ in range(2)]
        xs = [eng.pack(torch.randn(1, cin, 32, 64)) for _ in range(2)]
        res = [eng.pack(torch.randn(1, cout, 32, 64)) for _ in range(2)]
        conv64 = nn.Conv2d(64, 64, 3)
        x64 = eng.pack(torch.randn(1, 64, 32, 64))
        for t in EXP_TILES + (140, 141):
            assert t in PATCH_CFGS and is_patch_tile(t) and tile_korder(t) == 1, t
            if t in (140, 141, 143):                                       # persistent single-chunk tile: 64 input channels, <= 64 output channels, single launches
                eng.tile_override[(64, 64, 3, 1, 0)] = (t, 1, 0)
                _, rows, _ = eng.conv(x64, conv64, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
                assert eng.conv_log[-1]["tile"] == t and rows == 8          # 8 tiles of 8 x 32 pixels, one statistics row per WORKGROUP
                big = eng.pack(torch.randn(1, 64, 256, 512))                # 2048 tiles on (dry run: assumed) 256 CUs: 256 rows, and the
                ss = torch.zeros(4 * 64)                                    # finalize stays in the launch although the layer is large
                _, rows, _ = eng.conv(big, conv64, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(nn.BatchNorm2d(64), ss))
                assert rows == 256 and eng.last_finalized
                eng.tile_override[(cin, cout, 3, 1, 0)] = (t, 1, 0)
                with pytest.raises(RuntimeError):         # 128 channels: refused by the library
                    eng.conv(xs[0], convs[0], L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
                continue
            eng.tile_override[(cin, cout, 3, 1, 0)] = (t, 1, 0)
            eng.conv(xs[0], convs[0], L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
            assert eng.conv_log[-1]["tile"] == t
            eng.pair_override = (t, 1)
            assert eng.fused_norm_fits((t, 1, 0), 1, 32, 64, cout)
            ssa = eng.scratch("scale_shift", 4 * cout)
            with eng.scratch_set(1):
                ssb = eng.scratch("scale_shift", 4 * cout)
            ya, yb = eng.empty_act(1, 32, 64, cout), eng.empty_act(1, 32, 64, cout)
            eng.conv_pair(xs[0], convs[0], xs[1], convs[1], L.PAD_REFLECT, 1, ((norms[0], ssa), (norms[1], ssb)), ("a", "b"),
                          fuse=(L.ACT_NONE, 0.0, (res[0], None), (res[1], None), ya, yb))
            assert eng.conv_log[-1]["tile"] == t and eng.conv_log[-1]["fused_norm"]
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_paired_x_view_of_32_channel_layers():
    """engine.PairedXConv: a <= 32 -> 32 channel 3x3 layer as a 64 -> 64 layer over PAIRS of horizontally adjacent pixels (the NHWC
    tensor [H][W][32] is [H][W/2][64]) -- what puts the 64-byte-pixel ResnetBlocks (models/networks.py:554-593 at ngf_s = 32) on the
    persistent single-chunk kernels.  (1) The assembled weight reproduces the layer in torch, with reflection (vertical: reflection;
    horizontal: a clamp in the paired domain) and zero padding, 32 and fewer input channels.  (2) Dry run: the engine hands tiles
    140 / 141 the paired-x packing (w_korder 3), the library accepts the launch, reports one statistics row per workgroup and keeps the
    finalize in the launch; shapes the view does not cover are refused."""
    import torch.nn as nn
    import torch.nn.functional as F
    from vid2vid_amd import networks as N
    from vid2vid_amd import lib as L
    from vid2vid_amd.engine import Engine, PairedXConv
    torch.manual_seed(0)
    for cin in (32, 27):
        conv = nn.Conv2d(cin, 32, 3).requires_grad_(False)
        x = torch.randn(2, cin, 12, 16)
        px = PairedXConv(conv)
        xp = F.pad(x, (0, 0, 0, 0, 0, 32 - cin))
        n, c, H, W = xp.shape
        xpair = xp.permute(0, 2, 3, 1).reshape(n, H, W // 2, 64).permute(0, 3, 1, 2)
        unpair = lambda y: y.permute(0, 2, 3, 1).reshape(n, H, W, 32).permute(0, 3, 1, 2)
        t = F.pad(F.pad(xpair, (0, 0, 1, 1), mode="reflect"), (1, 1, 0, 0), mode="replicate")
        ref = F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), conv.weight, conv.bias)
        assert float((unpair(F.conv2d(t, px.weight, px.bias)) - ref).abs().max()) < 1e-5
        refz = F.conv2d(x, conv.weight, conv.bias, padding=1)
        assert float((unpair(F.conv2d(xpair, px.weight, px.bias, padding=1)) - refz).abs().max()) < 1e-5
    with pytest.raises(ValueError):
        PairedXConv(nn.Conv2d(32, 16, 3))
    # transposed counterpart: ConvTranspose2d <= 32 -> 16 as the 64 -> 32 transposed layer over paired pixels
    from vid2vid_amd.engine import PairedXConvT
    for cin in (32, 25):
        up = nn.ConvTranspose2d(cin, 16, 3, stride=2, padding=1, output_padding=1).requires_grad_(False)
        x = torch.randn(2, cin, 6, 8)
        px = PairedXConvT(up)
        xp = F.pad(x, (0, 0, 0, 0, 0, 32 - cin))
        n, c, H, W = xp.shape
        xpair = xp.permute(0, 2, 3, 1).reshape(n, H, W // 2, 64).permute(0, 3, 1, 2)
        y = F.conv_transpose2d(xpair, px.weight, px.bias, stride=2, padding=1, output_padding=1)       # [n][32][2H][W]
        yo = y.permute(0, 2, 3, 1).reshape(n, 2 * H, 2 * W, 16).permute(0, 3, 1, 2)
        assert float((yo - up(x)).abs().max()) < 1e-5
    with pytest.raises(ValueError):
        PairedXConvT(nn.ConvTranspose2d(32, 32, 3, stride=2, padding=1, output_padding=1))
    if torch.cuda.is_available():
        return                                       # the dry-run half is a CPU-host check
    N.set_record_only(True)
    try:
        eng = Engine(torch.device("cpu"), L.BF16, record_only=True)
        conv = nn.Conv2d(32, 32, 3)
        x = eng.pack(torch.randn(1, 32, 64, 1024))
        assert x.Cs == 32
        for t in (140, 141):
            eng.tile_override[(32, 32, 3, 1, 0)] = (t, 1, 0)
            ss = torch.zeros(4 * 32)
            _, rows, _ = eng.conv(x, conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(nn.BatchNorm2d(32), ss))
            assert eng.conv_log[-1]["tile"] == t and rows == 128 and eng.last_finalized      # 8 x 16 tiles of 8 x 32 PAIRED pixels
            with pytest.raises(RuntimeError):        # odd paired width (1000 / 2 = 500 is not a multiple of 32)
                eng.conv(eng.pack(torch.randn(1, 32, 64, 1000)), conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
        # tile 114: the persistent transposed stride-2 tile (64 -> <= 32 channels): full-tap packing (korder 2), one row per workgroup
        up = nn.ConvTranspose2d(64, 32, 3, stride=2, padding=1, output_padding=1)
        eng.tile_override[(64, 32, 3, 2, 1)] = (114, 1, 0)
        ss = torch.zeros(4 * 32)
        _, rows, (n_, OH, OW) = eng.conv(eng.pack(torch.randn(1, 64, 512, 1024)), up, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=True,
                                         fin=(nn.BatchNorm2d(32), ss))
        assert eng.conv_log[-1]["tile"] == 114 and rows == 256 and (OH, OW) == (1024, 2048) and eng.last_finalized
        up16 = nn.ConvTranspose2d(32, 16, 3, stride=2, padding=1, output_padding=1)      # 64-byte pixels: the paired-x view of tile 114
        eng.tile_override[(32, 16, 3, 2, 1)] = (114, 1, 0)
        ss = torch.zeros(4 * 16)
        _, rows, (n_, OH, OW) = eng.conv(eng.pack(torch.randn(1, 32, 64, 1024)), up16, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=True,
                                         fin=(nn.BatchNorm2d(16), ss))
        assert eng.conv_log[-1]["tile"] == 114 and rows == 128 and (OH, OW) == (128, 2048) and eng.last_finalized
        with pytest.raises(RuntimeError):            # 64 output channels: refused
            eng.tile_override[(64, 64, 3, 2, 1)] = (114, 1, 0)
            eng.conv(eng.pack(torch.randn(1, 64, 16, 64)), nn.ConvTranspose2d(64, 64, 3, stride=2, padding=1, output_padding=1), L.PAD_ZERO, None,
                     L.OUT_RAW_F32_NHWC, want_stats=True)
    finally:
        N.set_record_only(False)


def test_round5_label_paths_dry_run():
    """Round 5, host side of two label-path changes (CPU dry run; the GPU parity is in tests/test_gpu_kernels.py):
    (1) gather-sum stems with <= 16 output channels take 16-channel slices by default -- the table the library asks for has that
    geometry ([49 taps][1 slice][(cin + 1) rows x (16 x 2 + 16) bytes, rounded to 1 KiB] + the edge-row fragments of one 32-column MFMA
    tile), wider layers keep 32-channel slices, an explicit slice width is honoured;
    (2) the pooled label encoding is computed from the 1-byte label | edge codes when the frame plan has them (maps_u8 = 2), from the
    maps otherwise; label_nc > 126 cannot be coded and is refused on that path."""
    import ctypes as C
    from vid2vid_amd import networks as N
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib
    from vid2vid_amd.engine import Engine, LabelSource
    T, nc = 3, 35
    cin = T * (nc + 1)
    blob16 = ((cin + 1) * (16 * 2 + 16) + 1023) // 1024 * 1024
    blob32 = ((cin + 1) * (32 * 2 + 16) + 1023) // 1024 * 1024
    etab = lambda slices, ntiles: slices * ntiles * ((49 * T + 15) // 16) * 64 * 16
    assert lib.v2v_onehot_conv_table_bytes(cin, 16, L.BF16, 0, T, nc) == 49 * blob16 + etab(1, 1)
    assert lib.v2v_onehot_conv_table_bytes(cin, 12, L.BF16, 0, T, nc) == 49 * blob16 + etab(1, 1)
    assert lib.v2v_onehot_conv_table_bytes(cin, 16, L.BF16, 32, T, nc) == 49 * blob32 + etab(1, 1)      # explicit 32-wide slices
    assert lib.v2v_onehot_conv_table_bytes(cin, 32, L.BF16, 0, T, nc) == 49 * blob32 + etab(1, 1)
    assert lib.v2v_onehot_conv_table_bytes(cin, 128, L.BF16, 0, T, nc) == 49 * 4 * blob32 + etab(4, 1)
    if torch.cuda.is_available():
        return
    N.set_record_only(True)
    try:
        eng = Engine(torch.device("cpu"), L.BF16, record_only=True)
        H, W = 64, 96
        lab = torch.randint(0, nc, (T, H, W)).to(torch.uint8)
        inst = torch.randint(0, 5, (T, H, W)).to(torch.int32)
        src = LabelSource(lab, inst, T, nc)
        assert eng.label_codes(src, H, W) is not None and src.codes is not None
        x0, pooled, mask = eng.encode_labels_pooled(lab, inst, T, H, W, nc, [26], True, chunk_stride=True, source=src)     # from the codes
        assert pooled.t.shape == (1, H // 2, W // 2, 128) and mask.shape == (1, 1, H, W) and x0.onehot is src
        x0, pooled, _ = eng.encode_labels_pooled(lab, inst, T, H, W, nc, [26], True, chunk_stride=True)                     # from the maps
        assert pooled.t.shape == (1, H // 2, W // 2, 128)
        out = torch.empty(1, H // 2, W // 2, 1024, dtype=torch.bfloat16)
        rc = lib.v2v_encode_labels_pooled(C.c_void_p(src.codes.data_ptr()), None, C.c_void_p(out.data_ptr()), None, T, H, W, 200, 1024, None, 0, L.BF16, 2, None)
        assert rc != 0                                  # 200 labels do not fit the 7-bit code
    finally:
        N.set_record_only(False)


def _load_hazard_check():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_hazard_check", os.path.join(ROOT, "scripts", "mfma_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mfma_hazard_check_flags_the_round5_miscompute_pattern(tmp_path, capsys):
    """The checker on a hand-written disassembly of the pattern the V2V_STAMP_MASK build of conv_igemm_kernel<float,64,64,2,2,2,false>
    contained (round 5's "stamp build miscompute", root-caused in round 6): the loop's last 16-pass fp32 MFMA, the exit branch, and a
    `v_accvgpr_read_b32` of the last accumulator register a few wait states later -- once too early, once behind enough s_nop."""
    mod = _load_hazard_check()

    def dis(nops):
        lines = ["0000000000001000 <k>:",
                 "\tv_mfma_f32_32x32x2_f32 a[0:15], v35, v39, a[0:15]           // 000000001000: D3C40000 04024F23",
                 "\ts_cmp_eq_u32 s85, s5                                       // 000000001008: BF060555",
                 "\ts_cbranch_scc1 2                                           // 00000000100C: BF850002",
                 "\ts_mov_b32 s84, s85                                         // 000000001010: BED40055",
                 "\ts_branch 65531                                             // 000000001014: BF82FFFB",
                 "\ts_load_dwordx2 s[6:7], s[0:1], 0x2b0                       // 000000001018: C0060180 000002B0"]
        a = 0x1020
        for n in nops:
            lines.append("\ts_nop %d                                                    // %012X: BF80%04X" % (n, a, n))
            a += 4
        lines.append("\tv_accvgpr_read_b32 v17, a15                                // %012X: D3D84011 1800010F" % a)
        lines.append("\ts_endpgm                                                   // %012X: BF810000" % (a + 8))
        p = tmp_path / ("k_%d.dis" % len(nops))
        p.write_text("\n".join(lines) + "\n")
        return str(p)

    assert mod.check(dis([3])) == 1                      # 7 of 18 wait states: what the stamp build did
    assert "VIOLATION 7 of 18" in capsys.readouterr().out
    assert mod.check(dis([7, 6])) == 0                   # 18 wait states: fine
    assert mod.passes_of("v_mfma_f32_32x32x16_bf16") == (8, 11) and mod.passes_of("v_mfma_f32_16x16x32_bf16") == (4, 7)


def test_shipped_library_passes_the_mfma_hazard_check():
    """Every MFMA of the in-tree libv2v_hip.so keeps the result latency towards every non-MFMA access of its accumulators on every
    control-flow path (the compiler's own hazard search can miss the loop-exit edge; `__graft_entry__.build()` runs the same check)."""
    so = os.path.join(ROOT, "vid2vid_amd", "libv2v_hip.so")
    if not os.path.exists(so) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no built library / no llvm-objdump here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "mfma_hazard_check.py"), "--lib", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "MFMAs" in r.stdout


def test_fused_norm_tag_words_match_the_header():
    """include/v2v_hip.h V2V_FIN_TAG_WORD (the fused-norm launches' per-channel-tile launch tags inside fin_counter) is what the engine
    allocates for, and a fused launch's statistics buffer is the zero-initialised, tagged one (never the shared untagged `stats`)."""
    hdr = open(os.path.join(ROOT, "include", "v2v_hip.h")).read()
    m = re.search(r"#define\s+V2V_FIN_TAG_WORD\s+\(([^)]+)\)", hdr)
    assert m, "V2V_FIN_TAG_WORD missing from the header"
    from vid2vid_amd import engine as E
    assert eval(m.group(1), {"__builtins__": {}}) == E.FIN_TAG_OFFSET == E.FIN_ONEHOT_OFFSET + 256
    assert E.FIN_COUNTER_WORDS == E.FIN_TAG_OFFSET + 128
    eng = E.Engine("cpu", record_only=True)
    a = eng.scratch("stats_tagged", 64, zero=True)
    assert int(a.abs().sum()) == 0
    a.fill_(7.0)
    assert eng.scratch("stats_tagged", 32, zero=True) is a                 # no re-allocation, no re-zeroing while it fits
    b = eng.scratch("stats_tagged", 128, zero=True)
    assert b is not a and int(b.abs().sum()) == 0                          # a grown buffer starts as zeros: stale tags cannot survive
    src = open(os.path.join(ROOT, "vid2vid_amd", "engine.py")).read()
    assert 'self.scratch("stats_tagged", rows * pc.cout * 4, zero=True)' in src
