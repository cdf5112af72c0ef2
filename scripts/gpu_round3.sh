#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 30 --warmup 5 --dump-ops gpurun_out/ops_bf16.json > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench rc=$?"
cat gpurun_out/bench_bf16.json
timeout 900 python scripts/conv_sweep.py bf16 > gpurun_out/conv_sweep_bf16_cold.txt 2> gpurun_out/conv_sweep.err; echo "sweep rc=$?"
cat gpurun_out/conv_sweep_bf16_cold.txt
