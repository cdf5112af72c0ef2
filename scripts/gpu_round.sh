#!/bin/bash
# one GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 --dump-ops gpurun_out/ops_bf16.json > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench rc=$?"
cat gpurun_out/bench_bf16.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err; echo "rocprof rc=$?"
ls -R $R/gpurun_out/prof | head -30
