#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -30
timeout 600 python bench.py --steps 30 --warmup 5 --dump-ops gpurun_out/ops_bf16.json > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench rc=$?"
cat gpurun_out/bench_bf16.json
