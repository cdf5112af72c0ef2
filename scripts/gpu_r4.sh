#!/bin/bash
# Round-4 GPU visits.  scripts/gpu_r4.sh <tag> [parts...]   (every part writes under gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4}; shift
WHAT=${*:-parity}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
if has parity; then     # teacher-forced training parity at 1e-3, configs[4] as a training chunk, configs[0] literally, roles with real networks
  timeout 1700 python -m pytest tests -m gpu -q -rf --tb=short --timeout 1500 -s --durations=8 \
      -k "full_width_training or 256x128_two_frame or role_split_with_real" > gpurun_out/${TAG}_parity.log 2>&1; echo "parity rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|training chunk|forward|losses|grads|free-running|max \|d\||C1 256|^E  " gpurun_out/${TAG}_parity.log | cut -c1-1500 | tail -80
  lap parity
fi
if has benchdefault; then    # the driver's command, timed
  TB=$(date +%s)
  timeout 1700 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_bench_default.json"))
print({k: v for k, v in j.items() if not isinstance(v, (dict, list))})
print("windows", j["timing"]["windows_ms_per_step"])
r = j["roofline"]
print("roofline", r["kernel"], r["frac"], r["avg_launch_us"], "eager", r.get("eager"), "rocprof", r.get("in_graph_rocprof"))
print("per_kernel_ms", r["per_kernel_ms"])
for k in ("c1", "hires", "train", "train_hires"):
    print(k, json.dumps(j.get(k))[:2500])
PY
  tail -5 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap benchdefault
fi
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short --timeout 1500 --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -30
  lap tests
fi
