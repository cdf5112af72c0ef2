#!/bin/bash
# Short GPU visit for kernel iteration: probe, micro-benchmarks, the training-op parity tests, a short training bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
TAG=${1:-q}; shift
WHAT=${*:-probe wgrad tests train}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has probe; then timeout 60 scripts/probe_tr16.bin > gpurun_out/${TAG}_probe_tr16.txt 2>&1; head -20 gpurun_out/${TAG}_probe_tr16.txt; fi
if has wgrad; then
  timeout 300 python scripts/wgrad_bench.py > gpurun_out/${TAG}_wgrad.txt 2>&1; echo "wgrad rc=$?"; tail -12 gpurun_out/${TAG}_wgrad.txt
  V2V_WGRAD_BF16=legacy timeout 300 python scripts/wgrad_bench.py > gpurun_out/${TAG}_wgrad_legacy.txt 2>&1; echo "wgrad(legacy) rc=$?"; tail -10 gpurun_out/${TAG}_wgrad_legacy.txt | cut -c1-110
fi
if has wgradcfg; then
  for c in ${WGCFGS:-0 8 9}; do
    echo "== V2V_WGRAD_CFG=$c"; V2V_WGRAD_CFG=$c timeout 200 python scripts/wgrad_bench.py 2>&1 | grep -E "^wgrad|worst|Error|error" | head -9 | cut -c1-150
  done > gpurun_out/${TAG}_wgrad_cfgs.txt 2>&1; cat gpurun_out/${TAG}_wgrad_cfgs.txt
fi
if has lanes; then
  V2V_LANES=1 timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 180 -k "inference or graph or full_size" > gpurun_out/${TAG}_pytest_lanes.log 2>&1; echo "pytest(lanes) rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_lanes.log | tail -8
  LT=$R/gpurun_out/${TAG}_tune_lanes.json; rm -f $LT
  V2V_LANES=0 V2V_TUNE_CACHE=$LT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_lanes0.json 2> gpurun_out/${TAG}_bench_lanes0.err; echo "bench lanes=0 rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_lanes0.json
  V2V_LANES=1 V2V_TUNE_CACHE=$LT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_lanes1.json 2> gpurun_out/${TAG}_bench_lanes1.err; echo "bench lanes=1 rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_lanes1.json; tail -2 gpurun_out/${TAG}_bench_lanes1.err
  V2V_LANES=0 V2V_TUNE_CACHE=$LT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_lanes0b.json 2> gpurun_out/${TAG}_bench_lanes0b.err; echo "bench lanes=0 (again) rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_lanes0b.json
  V2V_LANES=1 V2V_TUNE_CACHE=$LT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_lanes1b.json 2> gpurun_out/${TAG}_bench_lanes1b.err; echo "bench lanes=1 (again) rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_lanes1b.json
fi
if has big; then
  BT=$R/gpurun_out/${TAG}_tune_2048.json; rm -f $BT
  V2V_LANES=0 V2V_TUNE_CACHE=$BT timeout 600 python bench.py --steps 10 --warmup 3 --width 2048 --height 1024 --scales 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_2048_lanes0.json 2> gpurun_out/${TAG}_bench_2048_lanes0.err; echo "bench2048 lanes=0 rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_2048_lanes0.json
  V2V_LANES=1 V2V_TUNE_CACHE=$BT timeout 600 python bench.py --steps 10 --warmup 3 --width 2048 --height 1024 --scales 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_2048_bf16.json 2> gpurun_out/${TAG}_bench_2048_bf16.err; echo "bench2048 lanes=1 rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_2048_bf16.json; tail -2 gpurun_out/${TAG}_bench_2048_bf16.err
fi
if has override; then
  OT=$R/gpurun_out/${TAG}_tune_ov.json; rm -f $OT
  V2V_TUNE_CACHE=$OT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_ov_base.json 2>/dev/null; echo "base: $(cut -c60-130 gpurun_out/${TAG}_bench_ov_base.json)"
  for ov in "1024,1024,3,1,0:55,1,0" "1024,1024,3,1,0:57,1,0" "1024,1024,3,1,0:53,1,0" "1024,1024,3,1,0:50,1,0" "1024,1024,3,1,0:55,1,0;512,512,3,1,0:55,1,0" "512,512,3,1,0:55,1,0" "512,512,3,1,0:57,1,0"; do
    V2V_TILE_OVERRIDE="$ov" V2V_TUNE_CACHE=$OT timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_ov.json 2>/dev/null; echo "$ov: $(cut -c60-130 gpurun_out/${TAG}_bench_ov.json)"
  done
fi
if has ftune; then
  FT=$R/gpurun_out/${TAG}_tune_ft.json; rm -f $FT
  V2V_TUNE_CACHE=$FT timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_ft.json 2> gpurun_out/${TAG}_bench_ft.err; echo "bench frame-tune rc=$?"; cut -c1-700 gpurun_out/${TAG}_bench_ft.json; grep "frame tune" gpurun_out/${TAG}_bench_ft.err
  V2V_TUNE_CACHE=$FT timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_ft2.json 2> gpurun_out/${TAG}_bench_ft2.err; echo "bench replay rc=$?"; cut -c60-130 gpurun_out/${TAG}_bench_ft2.json
  V2V_FRAME_TUNE=0 timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_noft.json 2> gpurun_out/${TAG}_bench_noft.err; echo "bench no frame-tune rc=$?"; cut -c60-130 gpurun_out/${TAG}_bench_noft.json
fi
if has pmcwgrad; then
  # where do the wgrad kernel's cycles go (one counter group per pass; rocprofv3 refuses mixed trace domains)
  cd /tmp; export TMPDIR=/tmp
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
    tag=$(echo $pass | cut -d' ' -f1)
    V2V_WGRAD_CFG=${WGCFG:-0} timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmcw_$tag -o pmc -- python $R/scripts/wgrad_bench.py > $R/gpurun_out/${TAG}_pmcw_$tag.log 2>&1; echo "pmc $tag rc=$?"
    python $R/scripts/pmc_summary.py $(find /tmp/pmcw_$tag -name "*.db" | head -1) "# V2V_WGRAD_CFG=${WGCFG:-0} rocprofv3 --kernel-trace --pmc $pass -- python scripts/wgrad_bench.py" > $R/gpurun_out/${TAG}_pmcw_$tag.txt 2>> $R/gpurun_out/${TAG}_pmcw_$tag.log
    grep -h "wgrad" $R/gpurun_out/${TAG}_pmcw_$tag.txt | cut -c1-60,97- | head -6
  done
  cd $R
fi
if has tests; then
  timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 180 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | tail -20
fi
if has alltests; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 180 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | tail -20
fi
if has train; then
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_train.json timeout 600 python bench.py --mode train --steps 12 --warmup 3 --with-vgg > gpurun_out/${TAG}_train_512_vgg_bf16.json 2> gpurun_out/${TAG}_train_512_vgg_bf16.err; echo "train rc=$?"
  cut -c1-250 gpurun_out/${TAG}_train_512_vgg_bf16.json; tail -2 gpurun_out/${TAG}_train_512_vgg_bf16.err
fi
if has trainprof; then
  cd /tmp; export TMPDIR=/tmp
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_train.json timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/proft_$TAG -o bench -- python $R/bench.py --mode train --steps 6 --warmup 3 --with-vgg > $R/gpurun_out/${TAG}_train_prof.json 2> $R/gpurun_out/${TAG}_train_prof.err; echo "rocprof train rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/proft_$TAG -name "*.db" | head -1) "# visit $TAG: V2V_TUNE_CACHE=<selections of the preceding run> rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 3 --with-vgg (bf16, 512x256, 2 frames per chunk)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train_prof.err
  head -22 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-180
  cd $R
fi
