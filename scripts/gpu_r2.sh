#!/bin/bash
# Round-2 GPU visits.  scripts/gpu_r2.sh <tag> [parts...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r2}; shift
WHAT=${*:-newtests}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
if has newtests; then
  timeout 1500 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 900 -s --durations=8 \
      -k "full_size or three_scales or feature_encoding" > gpurun_out/${TAG}_newtests.log 2>&1; echo "newtests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|rel err|Error|error" gpurun_out/${TAG}_newtests.log | cut -c1-400 | tail -40
  lap newtests
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 --durations=10 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -30
  tail -40 gpurun_out/${TAG}_pytest_gpu.log > gpurun_out/${TAG}_pytest_gpu_tail.txt
  lap tests
fi
if has smoke; then
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
  tail -1 gpurun_out/${TAG}_smoke.log
  lap smoke
fi
if has kernels; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "patch_kernel or conv2d_pair or norm" > gpurun_out/${TAG}_kernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/${TAG}_kernels.log | cut -c1-300 | tail -30
  lap kernels
fi
if has pp2bench; then
  timeout 600 python scripts/pp2_bench.py bf16 > gpurun_out/${TAG}_pp2_bench.txt 2>&1; echo "pp2 bench rc=$?"
  cat gpurun_out/${TAG}_pp2_bench.txt | cut -c1-200
  lap pp2bench
fi
if has golden; then
  timeout 1200 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -30
  lap golden
fi
if has twinbench; then
  for tw in 1 0; do
    V2V_TWIN=$tw timeout 500 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_twin$tw.json 2> gpurun_out/${TAG}_bench_twin$tw.err; echo "bench twin=$tw rc=$?"
    cut -c1-220 gpurun_out/${TAG}_bench_twin$tw.json; grep -E "frame tune|Error|error" gpurun_out/${TAG}_bench_twin$tw.err | tail -3
  done
  lap twinbench
fi
if has ablate; then
  timeout 300 python scripts/pp2_ablate.py > gpurun_out/${TAG}_pp2_ablate.txt 2>&1; echo "pp2 ablate rc=$?"
  cat gpurun_out/${TAG}_pp2_ablate.txt | cut -c1-160
  lap ablate
fi
if has pairtest; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "conv2d_pair" > gpurun_out/${TAG}_pairtest.log 2>&1; echo "pair test rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_pairtest.log | cut -c1-300 | tail -20
  lap pairtest
fi
if has benchnew; then
  rm -f gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 900 python bench.py --steps 30 --warmup 5 --retune --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
  cut -c1-3000 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
  lap benchnew
fi
if has allkernels; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short --timeout 300 > gpurun_out/${TAG}_allkernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/${TAG}_allkernels.log | cut -c1-300 | tail -30
  lap allkernels
fi
