"""Drop-in proof: the reference's OWN train.py and test.py, unmodified, driven through the three-line
`models/models.py` shim of INTEGRATION.md section A on this GPU-less host (tests/dropin/run_reference.py; the backend in
dry-run mode: every launch argument-checked, autograd graphs built and walked, optimizers stepped, nothing executed).
Needs /root/reference (the build container); nothing here runs on the GPU box."""
import json
import os
import subprocess
import sys

import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or torch.cuda.is_available(),
                                reason="needs the reference checkout and a GPU-less host (dry-run backend)")


def _run(mode, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin", "run_reference.py"), mode, str(tmp_path)],
                       capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_REPORT ")][-1]
    return json.loads(line[len("DROPIN_REPORT "):]), r.stdout


def test_reference_test_py_runs_through_the_shim(tmp_path):
    """test.py:25-54: create_model(opt) -> per-frame model.inference(A, B, inst) with the hidden state reset on change_seq,
    outputs consumed by the reference's util.tensor2label / tensor2im / Visualizer.save_images."""
    rep, out = _run("test", tmp_path)
    assert rep["calls"]["Vid2VidModelG.inference"] == 6                       # --how_many 6
    assert rep["shapes"]["inference"] == [[1, 3, 64, 128], [36, 64, 128]]       # (fake_B, real_A[0][0, -1])
    assert any(f.startswith("fake_B_") for f in rep["saved"]) and any(f.startswith("real_A_") for f in rep["saved"])
    assert out.count("process image...") == 6


def test_reference_train_py_runs_through_the_shim(tmp_path):
    """train.py:28-138 for two epochs of two sequences: create_model / create_optimizer / init_params, the chunk loop
    (modelG -> flowNet -> compute_fake_B_prev -> modelD(0, ...) -> get_all_skipped_frames -> modelD(s+1, ...) ->
    get_losses -> three loss_backward calls), Visualizer.print_current_errors on loss_names / loss_names_T,
    save_models and update_models -- including update_training_batch (niter_step 1) and update_fixed_params
    (niter_fix_global 1), after which the optimizer train.py captured at start-up keeps stepping."""
    rep, out = _run("train", tmp_path)
    c = rep["calls"]
    assert c["Vid2VidModelG.forward"] >= 8 and c["FlowNet.forward"] >= c["Vid2VidModelG.forward"]
    assert c["Vid2VidModelD.forward"] > c["Vid2VidModelG.forward"]             # image D + active temporal scales
    assert c["optimizer.step"] >= 2 * c["Vid2VidModelG.forward"]               # G, D (+ D_T per active scale) per chunk
    g = rep["shapes"]["Vid2VidModelG"]                                          # the reference's 7-tuple (vid2vid_model_G.py:137)
    assert g[0] == [1, 2, 3, 64, 128] and g[1] == g[0] and g[2] == [1, 2, 2, 64, 128] and g[3] == [1, 2, 1, 64, 128]
    assert g[4] == [1, 2, 36, 64, 128] and g[5] == [1, 3, 3, 64, 128] and g[6] == [[1, 2, 3, 64, 128], [1, 2, 3, 32, 64]]
    assert rep["shapes"]["FlowNet"] == [[1, 2, 2, 64, 128], [1, 2, 1, 64, 128]]
    assert rep["shapes"]["Vid2VidModelD"] == [[1, 1]] * 9                       # one (1,1) tensor per loss name
    for f in ("latest_net_G0.pth", "latest_net_G1.pth", "latest_net_D.pth", "latest_net_D_T0.pth", "latest_net_D_T1.pth",
              "2_net_G0.pth", "iter.txt", "loss_log.txt", "opt.txt"):
        assert f in rep["checkpoints"], f
    assert "Now finetuning all scales" in out and "Updating training sequence length" in out
    for name in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake", "G_Warp", "F_Flow", "F_Warp", "G_T_GAN0", "D_T_real0"):
        assert name + ":" in out, name
