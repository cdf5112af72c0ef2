#!/bin/bash
# GPU-box visit: full GPU parity suite without -x (every failure is reported), log under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${1:-1500} python -m pytest tests -m gpu -q -rf --tb=short ${2:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -80
