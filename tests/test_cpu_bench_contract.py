"""The bench.py JSON line (driver contract + the `roofline` / `cpu_baseline` objects): checked on the newest committed default run
under profiles/ -- the same consistency checks the judge applies to the driver's own BENCH record."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest(pattern):
    natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=natural)
    return files[-1] if files else None


@pytest.fixture(scope="module")
def line():
    """The FULL record of the newest committed default run (bench.py writes it to bench_full.json; the stdout line is the
    compact form checked by test_stdout_line_is_driver_readable)."""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default*.json")) if not f.endswith("_line.json")]
    natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]
    assert files, "no committed default bench record under profiles/"
    return json.load(open(sorted(files, key=natural)[-1]))


def test_driver_contract_keys(line):
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict)):
        assert isinstance(line[k], typ), k
    assert "vs_baseline" in line and line["vs_baseline"] is None         # BASELINE.md publishes no number for this metric
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "frames/sec" in line["metric"] and "frames/sec" in base["metric"]
    assert "512x256" in line["metric"] and "label2city 512x256" in line["config"]["workload"]     # BASELINE configs[1]
    # whole-job value = frames of all ranks / max-over-ranks time
    assert abs(line["value"] - line["n_gpus"] * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3


def test_roofline_object_is_self_consistent(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # achieved = algorithmic FLOP per launch / average launch duration of the kernel IN the frame graph
    assert abs(r["achieved"] - r["flop_per_launch"] / r["avg_launch_us"] / 1e6) / r["achieved"] < 1e-3
    # `frac` is the in-graph figure: the committed rocprofv3 kernel-trace average of the bench command for the
    # selected configuration when one exists, else this run's timeline stamps; the eager (alone-on-chip) figure is the companion
    g, e = r.get("in_graph_rocprof"), r["eager"]
    assert e["avg_launch_us"] > 0 and 0 < e["frac"] < 1
    if g is not None:
        # (a line measured before the matching rocprof file was committed carries the timeline figure instead: an upper bound)
        assert abs(g["avg_launch_us"] - r["avg_launch_us"]) / r["avg_launch_us"] < 1e-6 or \
            (r["measured"].startswith("in the frame graph") and r["avg_launch_us"] >= g["avg_launch_us"])
        assert os.path.exists(os.path.join(ROOT, g["source"].split(" ")[0]))
        assert abs(g["avg_launch_us"] - e["avg_launch_us"]) / e["avg_launch_us"] < 0.15      # in the graph vs alone: same kernel
    live = r.get("in_graph_live")
    if live is not None:                                     # stamps are an upper bound of the rocprof duration
        assert live["avg_launch_us"] >= 0.95 * r["avg_launch_us"]
    # measured HBM traffic (PMC) is per launch like `achieved`, and not below the algorithmic bytes
    assert r["traffic"] is None or r["traffic"] >= r["traffic_detail"]["algorithmic_bytes_per_launch"]
    # the dominant kernel's launches fit into the frame: serial sum <= lanes x frame time; whole-frame work <= peak
    lanes = line["config"]["graph_lanes"]
    assert r["launches_per_frame"] * r["avg_launch_us"] * 1e-3 <= lanes * line["ms_per_step"]
    assert 0 < r["frame_in_graph"]["frac"] <= r["frac"] + 0.05


def test_cpu_baseline_and_parity_objects(line):
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == line["unit"]
    assert isinstance(c["sample"], str) and "frames" in c["sample"]
    p = line["parity"]
    assert p, "parity object missing from the bench line"
    fp32_errs = [v for k, v in _walk(p["fp32"]) if k.endswith("max_rel")]
    assert fp32_errs and max(fp32_errs) <= p["tolerance_fp32"] == 1e-3    # north_star: 1e-3 relative, fp32, every head
    assert p["fp32_ok"] is True and p["bf16_max_rel"] > 0                  # bf16 error is a committed number, not a print
    assert line["fp32"]["value"] > 0 and line["host_fed"]["value"] > 0 and line["train"]["value"] > 0


def _walk(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _walk(v, prefix + k + ".")
        elif isinstance(v, (int, float)) and not isinstance(v, bool):
            yield prefix + k, float(v)


def test_round3_companion_objects(line):
    """The default line carries BOTH resolutions of BASELINE's metric, the parity-carrying precisions beside the bf16
    throughput mode, a stable timer, and the training chunk's parity against the oracle."""
    t = line["timing"]
    w = sorted(t["windows_ms_per_step"])
    assert len(w) >= 5 and abs(w[len(w) // 2] - line["ms_per_step"]) < 1e-3 and (w[-1] - w[0]) / w[0] < 0.05
    h = line["hires"]
    assert "2048x1024" in h["metric"] and h["value"] > 0 and abs(h["value"] - 1e3 / h["ms_per_step"]) / h["value"] < 1e-3
    assert h["parity"]["fp32_ok"] is True and h["parity"]["fp32_max_rel"] <= 1e-3 and h["cpu_baseline"]["value"] > 0
    hr = h["roofline"]                                       # the configuration holding the most TIME of the 2048x1024 frame
    assert hr["bound"] in ("hbm", "mfma") and 0 < hr["frac"] < 1 and abs(hr["frac"] - hr["achieved"] / hr["peak"]) < 1e-3
    assert hr["ms_per_frame"] >= hr["flop_heaviest"]["ms_per_frame"] and 0 < hr["frame_vs_per_layer_bounds"]["frac"] < 1
    x = line["x3"]
    assert x["ok_1e-3"] is True and x["max_rel"] <= 1e-3 and x["value"] > line["fp32"]["value"] * 1.5      # the point of the mode
    assert x["x3_flop_share"] > 0.7
    p = line["train"]["parity"]
    assert p["fp32_ok"] is True and p["fp32"]["max_forward"] <= 1e-3 and p["fp32"]["max_loss"] <= 1e-3        # EVERY frame (teacher-forced)
    assert set(p["fp32"]["grads"]) == {"G", "D", "DT"} and all(g["finite"] for g in p["fp32"]["grads"].values())
    assert p["fp32"]["max_grad_norm"] <= p["tolerance_fp32"]["grad_norm"] == 2.5e-3 and p["fp32"]["max_grad_l2"] <= 5e-3 and p["bf16"]["max_grad_l2"] < 0.5
    assert p["free_running_fp32"]["max_forward"] < 1e-2
    assert line["train"]["flownet2"]["pairs_per_s"] > 0


def test_round4_objects(line):
    """the parity-carrying throughput as flat top-level keys, BASELINE configs[0] literally (256x128
    2-frame clip, GPU vs the CPU oracle timed on the same clip), and configs[4]'s geometry as a TRAINING chunk on one GPU
    with its fp32 parity against the oracle, frames trained/s and peak memory."""
    assert line["parity_ok"] is True and line["parity_max_rel"] <= 1e-3 and line["parity_value"] == line["x3"]["value"]
    assert line["hires_value"] == line["hires"]["value"] and line["train_hires_value"] == line["train_hires"]["value"]
    # the flat keys sit ahead of the nested objects (the driver's record keeps top-level scalars)
    keys = list(line.keys())
    assert keys.index("parity_value") < keys.index("config") and keys.index("train_hires_value") < keys.index("roofline")
    c1 = line["c1"]
    assert "256x128" in c1["workload"] and "2-frame clip" in c1["workload"] and c1["value"] > 0
    assert c1["parity"]["fp32_ok"] is True and c1["parity"]["x3_ok"] is True and c1["parity"]["frames"] == 2
    assert c1["cpu_baseline"]["value"] > 0 and c1["cpu_baseline"]["cores"] >= 1 and c1["value"] > 100 * c1["cpu_baseline"]["value"]
    th = line["train_hires"]
    assert "2048x1024" in th["metric"] and "n_scales_spatial=3 num_D=4" in th["workload"] and th["value"] > 0
    assert 0 < th["peak_memory_gb"] < 288
    p = th["parity"]
    if p is None:          # round 5: the 2048x1024 chunk's CPU-oracle parity is opt-in in bench.py (--train-hires-parity); it is the
        return             # -m gpu test test_full_width_training_chunk_2048x1024_s3_vs_oracle
    assert p["fp32_ok"] is True and p["fp32"]["max_forward"] <= 1e-3 and p["fp32"]["max_loss"] <= 1e-3 and p["fp32"]["max_grad_norm"] <= 1e-3
    assert set(p["fp32"]["grads"]) == {"G", "D"} and p["fp32"]["grads"]["G"]["numel"] > 4e8          # all three scales' parameters


def test_stdout_line_is_driver_readable(line):
    """BENCH_r04.parsed was null -- the one stdout line had grown to 21.6 KB and the driver keeps a 10 KB tail.  The
    stdout line is now the compact form: well under 8 KB, strict JSON (no NaN / Infinity), and the last line of a 10 KB tail."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    compact = bench.compact_line(line)
    text = json.dumps(compact, allow_nan=False)
    assert len(text) < 8192 and len(text) <= bench.LINE_BUDGET and "\n" not in text
    strict = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in strict, k
    assert strict["value"] == line["value"] and strict["ms_per_step"] == line["ms_per_step"]
    r = strict["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = strict["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("reference", "port") and isinstance(c["sample"], str)
    # nothing nested deeper than one object: the driver's record keeps top-level scalars and these few objects
    for k, v in strict.items():
        if isinstance(v, dict):
            assert all(not isinstance(x, dict) for x in v.values()), k
    # a 10 KB stdout tail whose head is cut anywhere still ends in the complete line
    noise = "x" * 20000 + "\n"
    tail = (noise + text + "\n")[-10000:]
    assert json.loads(tail.strip().splitlines()[-1])["value"] == line["value"]
    # non-finite floats never reach the line
    bad = dict(line, value=float("nan"))
    assert json.loads(json.dumps(bench.compact_line(bench._finite(bad)), allow_nan=False))["value"] is None


def test_emit_record_writes_one_line_and_the_full_record(line, tmp_path, monkeypatch):
    import io
    import sys
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf, err = io.StringIO(), io.StringIO()
    monkeypatch.setattr(sys, "stderr", err)
    bench.emit_record(line, buf)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_BUDGET
    assert json.loads(lines[0])["full_record"] == bench.FULL_RECORD
    full = json.load(open(tmp_path / bench.FULL_RECORD))
    assert full["value"] == line["value"] and "train" in full and len(err.getvalue()) < 200
