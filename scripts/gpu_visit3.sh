#!/bin/bash
# Full GPU visit: parity suite, smoke, bench (512x256 + 2048x1024 S=3), rocprofv3 kernel stats, PMC traffic passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-v}; shift
WHAT=${*:-tests smoke bench prof pmc big}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1200 python -m pytest tests -m gpu -q -rf --tb=short > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -20
fi
if has smoke; then
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
  tail -1 gpurun_out/${TAG}_smoke.log
fi
if has bench; then
  rm -f gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 600 python bench.py --steps 30 --warmup 5 --dump-ops gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; echo "bench rc=$?"
  cut -c1-1200 gpurun_out/${TAG}_bench_bf16.json
fi
if has big; then
  timeout 900 python bench.py --steps 10 --warmup 3 --width 2048 --height 1024 --scales 3 --no-cpu-baseline --dump-ops gpurun_out/${TAG}_ops_2048_bf16.json > gpurun_out/${TAG}_bench_2048_bf16.json 2> gpurun_out/${TAG}_bench_2048_bf16.err; echo "bench2048 rc=$?"
  cut -c1-900 gpurun_out/${TAG}_bench_2048_bf16.json; tail -2 gpurun_out/${TAG}_bench_2048_bf16.err
fi
if has prof; then
  cd /tmp
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.json 2> $R/gpurun_out/${TAG}_bench_prof.err; echo "rocprof rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) "# round 1, visit $TAG: V2V_TUNE_CACHE=<the preceding bench run's selections> rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline (bf16, 512x256)" > $R/gpurun_out/${TAG}_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_bench_prof.err
  head -14 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
  cd $R
fi
if has pmc; then
  cd /tmp
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 400 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-frames 1 > $R/gpurun_out/${TAG}_pmc_$tag.json 2> $R/gpurun_out/${TAG}_pmc_$tag.err; echo "pmc $tag rc=$?"
    python $R/scripts/pmc_summary.py $(find /tmp/pmc_$tag -name "*.db" | head -1) "# rocprofv3 --kernel-trace --pmc $pass -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-frames 1" > $R/gpurun_out/${TAG}_pmc_$tag.txt 2>> $R/gpurun_out/${TAG}_pmc_$tag.err
  done
  grep -h "conv3x3\|conv7x7" $R/gpurun_out/${TAG}_pmc_*.txt | cut -c1-60,97- | head -40
  cd $R
fi
if has traffic; then
  CFG=$(python - <<PY
import json
ops = json.load(open("$R/gpurun_out/${TAG}_ops_bf16.json"))
c = [o["tile"] for o in ops if o["op"] == "conv_igemm" and o["label"].endswith("res_img.0.c1")]
print(",".join(str(v) for v in c[0]))
PY
)
  echo "dominant res1024 config: $CFG"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tr_f -o pmc -- python $R/scripts/conv_layer_run.py --cfg $CFG > $R/gpurun_out/${TAG}_traffic_fetch.log 2>&1; echo "traffic fetch rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tr_w -o pmc -- python $R/scripts/conv_layer_run.py --cfg $CFG > $R/gpurun_out/${TAG}_traffic_write.log 2>&1; echo "traffic write rc=$?"
  python $R/scripts/pmc_traffic.py $(find /tmp/tr_f -name "*.db" | head -1) $(find /tmp/tr_w -name "*.db" | head -1) $CFG $R/gpurun_out/${TAG}_traffic.json
  cd $R
fi
