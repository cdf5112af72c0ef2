#!/bin/bash
# GPU visit: parity suite, conv sweep (tile x split-K x prefetch), bench with autotune.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-v3}; shift
WHAT=${*:-tests sweep bench}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error|error:" gpurun_out/${TAG}_pytest_gpu.log | tail -30
fi
if has newtests; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "splitk or prefetch or every_tile or finalize" > gpurun_out/${TAG}_pytest_new.log 2>&1; echo "pytest-new rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error|error:" gpurun_out/${TAG}_pytest_new.log | tail -30
fi
if has sweep; then
  timeout 500 python scripts/conv_sweep2.py bf16 > gpurun_out/${TAG}_conv_sweep2_bf16.txt 2> gpurun_out/${TAG}_conv_sweep2.err; echo "sweep rc=$?"
  cut -c1-1500 gpurun_out/${TAG}_conv_sweep2_bf16.txt
fi
if has bench; then
  timeout 600 python bench.py --steps 30 --warmup 5 --dump-ops gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; echo "bench rc=$?"
  cat gpurun_out/${TAG}_bench_bf16.json; tail -3 gpurun_out/${TAG}_bench_bf16.err
fi
