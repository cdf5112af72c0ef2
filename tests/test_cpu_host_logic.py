"""Host-side logic of the training path against vectors produced by the REFERENCE's own functions
(tests/golden/make_golden_skipped.py): the rolling frame history and the per-temporal-scale frame groups of
models/vid2vid_model_D.py:274-328.  CPU only; no kernels involved (pure indexing)."""
import json
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skipped_frames.json")


def _flownet_stub(a, b):
    return a - b / 2, a + b / 4


def _rows(t):
    return None if t is None else [list(map(float, row)) for row in t.reshape(t.shape[0], -1).tolist()]


def test_skipped_frames_dense_and_sparse_match_reference():
    from vid2vid_amd.models import vid2vid_model_D as M
    cases = json.load(open(GOLDEN))["cases"]
    assert len(cases) >= 6
    for rec in cases:
        t_scales, tD, nfl = rec["t_scales"], rec["tD"], rec["n_frames_load"]
        real_all = flow_all = conf_all = None
        for c, want in enumerate(rec["dense"]):
            fr = torch.arange(c * nfl, (c + 1) * nfl, dtype=torch.float32).view(1, nfl, 1, 1, 1)
            real_all, real_sk = M.get_skipped_frames(real_all, fr, t_scales, tD)
            flow_all, conf_all, flow_sk, conf_sk = M.get_skipped_flows(_flownet_stub, flow_all, conf_all, real_sk, fr * 10,
                                                                      fr * 100, t_scales, tD)
            ctx = "dense t_scales=%d tD=%d nfl=%d chunk %d" % (t_scales, tD, nfl, c)
            assert _rows(real_all) == want["all"], ctx
            assert [_rows(t) for t in real_sk] == want["sk"], ctx
            assert _rows(flow_all) == want["flow_all"], ctx
            assert [_rows(t) for t in flow_sk] == want["flow_sk"], ctx
            assert [_rows(t) for t in conf_sk] == want["conf_sk"], ctx
        b_all, f_all = [None] * t_scales, [None] * t_scales
        for c, want in enumerate(rec["sparse"]):
            i = c * nfl
            fr = torch.arange(i, i + nfl, dtype=torch.float32).view(1, nfl, 1, 1, 1)
            b_all, b_sk = M.get_skipped_frames_sparse(b_all, fr, t_scales, tD, nfl, i)
            f_all, f_sk = M.get_skipped_frames_sparse(f_all, fr * 10, t_scales, tD, nfl, i, is_flow=True)
            ctx = "sparse t_scales=%d tD=%d nfl=%d chunk %d" % (t_scales, tD, nfl, c)
            assert [_rows(t) for t in b_all] == want["all"], ctx
            assert [_rows(t) for t in b_sk] == want["sk"], ctx
            assert [_rows(t) for t in f_sk] == want["flow_sk"], ctx


class _Rec:
    """Recording stand-in for modelG / modelD / data_loader / visualizer (same as tests/golden/make_golden_schedule.py)."""

    def __init__(self, log, name):
        self.log, self.name = log, name
        self.module = self
        self.dataset = self

    def __getattr__(self, attr):
        def call(*a, **k):
            self.log.append([self.name, attr] + [x if isinstance(x, (int, str, float)) else str(x) for x in a])
        return call

    def __len__(self):
        return 37


def _norm(log):
    """numpy integers of the reference's np.loadtxt were recorded as strings: compare by value"""
    return [[int(x) if isinstance(x, str) and x.lstrip("-").isdigit() else x for x in row] for row in log]


def test_resume_and_schedule_helpers_match_reference_traces(tmp_path):
    """init_params / update_models / save_models (models/models.py:104-163) over 24 option sets: same return tuple, same
    calls on the models / data loader in the same order, same iter.txt contents as the reference's own functions."""
    import types
    import numpy as np
    from vid2vid_amd.models import schedule as S
    traces = json.load(open(os.path.join(os.path.dirname(GOLDEN), "schedule_traces.json")))
    assert len(traces) == 24
    for n, rec in enumerate(traces):
        sc = rec["scenario"]
        ck = str(tmp_path / ("c%d" % n))
        os.makedirs(os.path.join(ck, "x"))
        opt = types.SimpleNamespace(checkpoints_dir=ck, name="x", **{k: v for k, v in sc.items() if k != "iter"})
        if sc["continue_train"]:
            np.savetxt(os.path.join(ck, "x", "iter.txt"), sc["iter"], delimiter=",", fmt="%d")
        log = []
        G, D, loader, vis = _Rec(log, "G"), _Rec(log, "D"), _Rec(log, "loader"), _Rec(log, "vis")
        ret = S.init_params(opt, G, D, loader)
        assert [int(v) for v in ret[:-1]] == rec["init_ret"], sc
        assert _norm(log) == _norm(rec["init_log"]), sc
        for ep in rec["epochs"]:
            del log[:]
            S.update_models(opt, ep["epoch"], G, D, loader)
            assert _norm(log) == _norm(ep["update_log"]), (sc, ep["epoch"])
            del log[:]
            S.save_models(opt, ep["epoch"], 3, 2000, vis, ret[-1], G, D, end_of_epoch=False)
            S.save_models(opt, ep["epoch"], 3, 2001, vis, ret[-1], G, D, end_of_epoch=True)
            assert _norm([l for l in log if l[0] != "vis"]) == _norm(ep["save_log"]), (sc, ep["epoch"])
            itxt = open(ret[-1]).read().split() if os.path.exists(ret[-1]) else None
            assert itxt == ep["iter_txt"], (sc, ep["epoch"])


def test_basemodel_schedule_arithmetic_matches_reference():
    """update_training_batch / update_learning_rate (models/base_model.py:154-181) against the reference's own methods
    run on plain namespaces (tests/golden/make_golden_basemodel.py)."""
    import types
    from vid2vid_amd.models.base_model import BaseModel as B
    ref = json.load(open(os.path.join(os.path.dirname(GOLDEN), "basemodel_schedule.json")))
    for rec in ref["batch"]:
        s = types.SimpleNamespace(n_gpus=rec["n_gpus"], n_frames_per_gpu=1, n_frames_load=rec["n_gpus"], n_frames_bp=1,
                                  opt=types.SimpleNamespace(max_frames_per_gpu=rec["max_frames_per_gpu"],
                                                            max_frames_backpropagate=rec["max_frames_backpropagate"]))
        trace = []
        for ratio in range(0, 5):
            B.update_training_batch(s, ratio)
            trace.append([s.n_frames_bp, s.n_frames_per_gpu, s.n_frames_load])
        assert trace == rec["trace"], rec
    for rec in ref["lr"]:
        group = {"lr": rec["lr"]}
        s = types.SimpleNamespace(opt=types.SimpleNamespace(lr=rec["lr"], niter=rec["niter"], niter_decay=rec["niter_decay"]),
                                  old_lr=rec["lr"], optimizer_G=types.SimpleNamespace(param_groups=[group]))
        for epoch, lr, old in rec["vals"]:
            B.update_learning_rate(s, epoch, "G")
            assert abs(group["lr"] - lr) <= 1e-15 and abs(s.old_lr - old) <= 1e-15


def test_basemodel_tensor_helpers_match_reference():
    """get_edges (models/base_model.py:146-152), compute_mask / compute_fake_B_prev (models/vid2vid_model_G.py:322-336)."""
    import types
    import numpy as np
    from vid2vid_amd.models.base_model import BaseModel as B
    from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG as G
    g = dict(np.load(os.path.join(os.path.dirname(GOLDEN), "basemodel_helpers.npz")))
    t = lambda k: torch.from_numpy(g[k])
    assert torch.equal(B.get_edges(types.SimpleNamespace(), t("inst")), t("edges"))
    ns = lambda labels: types.SimpleNamespace(opt=types.SimpleNamespace(fg_labels=labels))
    assert torch.equal(G.compute_mask(ns([2]), t("real_As"), 1), t("m1"))
    assert torch.equal(G.compute_mask(ns([0, 3, 5]), t("real_As"), 1, 3), t("m2"))
    assert torch.equal(G.compute_fake_B_prev(None, t("rb_prev"), None, t("fake")), t("p1"))
    assert torch.equal(G.compute_fake_B_prev(None, t("rb_prev"), [t("last")], t("fake")), t("p2"))
    assert torch.equal(G.compute_fake_B_prev(None, t("rb_prev"), [t("last")], t("fake")[:, :1]), t("p3"))


def test_face_feature_lookup_matches_reference():
    """nearest_face_features == the nearest-neighbour half of the reference's get_face_features
    (models/vid2vid_model_G.py:296-320; vectors from tests/golden/make_golden_facefeat.py)."""
    import numpy as np
    from vid2vid_amd.models.vid2vid_model_G import nearest_face_features
    g = dict(np.load(os.path.join(os.path.dirname(GOLDEN), "face_feature_lookup.npz")))
    feat_num = int(g["feat_num"])
    features = {k: g["features.%d" % k] for k in range(7)}
    for i in range(3):
        out = nearest_face_features(torch.from_numpy(g["case%d.feat_map" % i]), torch.from_numpy(g["case%d.inst" % i]),
                                    features, feat_num)
        assert torch.allclose(out, torch.from_numpy(g["case%d.out" % i]), rtol=0, atol=1e-6), i


def test_label_colormaps_match_reference(golden):
    """vid2vid_amd.visual.labelcolormap == the reference's util.labelcolormap (util/util.py:156-181) for the two Cityscapes
    tables and the generated bit-interleaved map (fixture made by tests/golden/make_golden_visual.py from the reference)."""
    import numpy as np
    from vid2vid_amd import visual
    g = golden("visual_util")
    for n in (35, 20, 12):
        assert np.array_equal(visual.labelcolormap(n), g["lab%d.cmap" % n]), n


def test_fastdiv_magic_divides_exactly():
    """Conv epilogue row -> pixel map: q = (umulhi(M, n) + n) >> l with the library's (M, l) equals n // d for every 0 <= n < 2^31."""
    import ctypes as C
    import random
    from vid2vid_amd import lib as L
    rnd = random.Random(7)
    ds = list(range(1, 600)) + [1 << k for k in range(31)] + [(1 << k) - 1 for k in range(2, 32)] + [(1 << k) + 1 for k in range(1, 31)] \
        + [rnd.randrange(1, 1 << 31) for _ in range(500)] + [32 * 64, 64 * 128, 128 * 256, 256 * 512, 511 * 1023, 2048 * 1024 // 4]
    for d in ds:
        if not 1 <= d < (1 << 31):
            continue
        m, l = C.c_uint32(0), C.c_int32(0)
        assert L.lib.v2v_fastdiv_magic(d, C.byref(m), C.byref(l)) == 0
        M, sh = m.value, l.value
        for n in [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - 2] + [rnd.randrange(0, 1 << 31) for _ in range(40)]:
            if 0 <= n < (1 << 31):
                t = (M * n) >> 32
                assert t + n < (1 << 32)
                assert (t + n) >> sh == n // d, (n, d, M, sh)
    m, l = C.c_uint32(0), C.c_int32(0)
    assert L.lib.v2v_fastdiv_magic(0, C.byref(m), C.byref(l)) != 0


def test_grouped_xcd_map_is_a_bijection_with_one_member_per_xcd():
    """csrc/conv_igemm_kernel.h grouped_xcd_map (paired launches, round 4): restated on the host -- every (member, tile) pair is
    worked on by exactly one workgroup, XCD x (= blockIdx.x & 7, whichever z) works on member x >> 2 only, and an XCD's tiles are
    one contiguous run of ntot / 4 tiles (4 channel tiles x all pixel tiles for the 1024 -> 1024 pair: one input + a quarter of
    the weights per XCD instead of both inputs + an eighth of each member's weights)."""
    def grouped_xcd_map(bx, bz, ntot):
        xcd = bx & 7
        idx = (bx >> 3) + bz * (ntot >> 3)
        return xcd >> 2, (xcd & 3) * (ntot >> 2) + idx
    for ntot in (8, 16, 128, 256, 1024):
        seen = {}
        per_xcd = {}
        for bz in (0, 1):
            for bx in range(ntot):
                member, tile = grouped_xcd_map(bx, bz, ntot)
                assert 0 <= tile < ntot and member in (0, 1)
                assert (member, tile) not in seen, (ntot, bx, bz)
                seen[(member, tile)] = (bx, bz)
                per_xcd.setdefault(bx & 7, []).append((member, tile))
        assert len(seen) == 2 * ntot
        for xcd, lst in per_xcd.items():
            assert {m for m, _ in lst} == {xcd >> 2}
            tiles = sorted(t for _, t in lst)
            assert tiles == list(range(tiles[0], tiles[0] + ntot // 4)) and tiles[0] == (xcd & 3) * (ntot // 4)
    # the 1024 -> 1024 pair: 8 pixel tiles x 16 channel tiles per member, channel tile = tile // 8
    chan_tiles = {}
    for bz in (0, 1):
        for bx in range(128):
            member, tile = grouped_xcd_map(bx, bz, 128)
            chan_tiles.setdefault(bx & 7, set()).add((member, tile // 8))
    assert all(len(v) == 4 for v in chan_tiles.values())          # 4 channel tiles of ONE member per XCD (was 2 + 2)


def test_patch_tile_index_by_multiplication_equals_the_divisions():
    """csrc/conv_igemm_kernel.h patch_tile_index: tile index -> (channel tile, pixel tile, image, tile row, tile column) with the
    library's magic numbers (ConvKArgs.idx_m / idx_l) instead of run-time divisions -- same values for every tile of the BASELINE
    geometries and a few ragged ones."""
    import ctypes as C
    from vid2vid_amd import lib as L

    def magic(d):
        m, l = C.c_uint32(0), C.c_int32(0)
        assert L.lib.v2v_fastdiv_magic(d, C.byref(m), C.byref(l)) == 0
        return m.value, l.value

    def fdiv(n, ml):
        return (((ml[0] * n) >> 32) + n) >> ml[1]

    for N, OH, OW, TH, TW, cout, BN in [(1, 32, 64, 8, 32, 1024, 64), (1, 512, 1024, 8, 32, 64, 64), (1, 512, 1024, 4, 64, 64, 64),
                                        (2, 37, 70, 4, 32, 200, 128), (3, 9, 33, 2, 64, 40, 64), (1, 1024, 2048, 8, 32, 32, 64)]:
        tiles_h, tiles_w = -(-OH // TH), -(-OW // TW)
        m_tiles, n_tiles = N * tiles_h * tiles_w, -(-cout // BN)
        mm, mt_, mw = magic(m_tiles), magic(tiles_h * tiles_w), magic(tiles_w)
        step = max(1, (m_tiles * n_tiles) // 5000)
        for lin in list(range(0, m_tiles * n_tiles, step)) + [m_tiles * n_tiles - 1]:
            nt = fdiv(lin, mm); mt = lin - nt * m_tiles
            n_img = fdiv(mt, mt_); trem = mt - n_img * tiles_h * tiles_w
            th = fdiv(trem, mw); tw = trem - th * tiles_w
            assert (nt, mt) == divmod(lin, m_tiles)
            assert (n_img, trem) == divmod(mt, tiles_h * tiles_w)
            assert (th, tw) == divmod(trem, tiles_w)
            assert 0 <= nt < n_tiles and 0 <= n_img < N and 0 <= th < tiles_h and 0 <= tw < tiles_w
