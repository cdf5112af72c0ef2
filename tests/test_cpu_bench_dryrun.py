"""The commands the driver launches for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N [--mode train]` -- executed here with N = 2 on the CPU host: `--dry-run` swaps RCCL for gloo and puts
the library in dry-run mode (every launch argument-checked, nothing executed), everything else is the code path of the real run: process
group, one sequence per rank, the start-up broadcast and the bucketed gradient all-reduce of the training mode, the barrier +
max-over-ranks timer, ONE JSON line from rank 0 (the N > 1 command had never been executed anywhere)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "V2V_TUNE_CACHE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dry-run", "--ngf", "16", "--width", "128", "--height", "64"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    # exactly one JSON line, from rank 0, and it is the LAST line of stdout (gloo's own "[Gloo] Rank r is connected ..." notices
    # precede it; RCCL prints none) -- the driver parses the last line
    assert len(js) == 1 and lines[-1] == js[0], r.stdout[-2000:]
    return json.loads(js[0])


def _contract(j):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in j, k
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True
    c = j["config"]
    assert c["world_size"] == 2 and c["sequences"] == 2 and c["backend"] == "gloo" and c["dry_run"] is True
    assert isinstance(c["collective"], str) and c["collective"]
    assert j["data"].startswith("dry-run") and j["value"] > 0


def test_inference_replicas_two_ranks_dry_run():
    j = _run([])
    _contract(j)
    assert "frames/sec" in j["metric"] and "replicas only" in j["config"]["parallelism"]
    # whole-job value: the frames of BOTH ranks over the max-over-ranks window
    assert abs(j["value"] - 2 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3


def test_training_data_parallel_two_ranks_dry_run():
    j = _run(["--mode", "train"])
    _contract(j)
    c = j["config"]
    assert "all-reduce" in c["collective"] and c["parallelism"].startswith("dp2")
    # three optimizers (G, D, D_T0) stepped per chunk: their buckets went through the process group, G's from inside its backward pass
    assert c["grad_sync_buckets_inside_backward"] + c["grad_sync_buckets_at_step"] > 0
    assert c["frames_per_step"] == 2 and abs(j["value"] - 2 * 2 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3
