#!/bin/bash
# Round-6 GPU visits.  scripts/gpu_r6.sh <tag> [parts...]   (every part writes under gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r6}; shift
WHAT=${*:-alltests}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
LEAN="--no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3"
if has fn2tests; then    # FlowNet2 at BASELINE sizes, and its real output inside the 512x256 training-chunk parity
  timeout 1500 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short -s -k "flownet2_at_baseline_size or full_width_training_chunk_512x256" -p no:cacheprovider > gpurun_out/${TAG}_fn2tests.log 2>&1; echo "fn2tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|flownet2 fp32 at|^E  |forward|losses|grads  " gpurun_out/${TAG}_fn2tests.log | cut -c1-400 | tail -40
  lap fn2tests
fi
if has trainprof; then   # steady-state rocprofv3 table of the training step (a first run fills the tile cache, the profiled second run replays it)
  GEO=${GEO:-}
  rm -f /tmp/tune_train.json
  V2V_TUNE_CACHE=/tmp/tune_train.json timeout 900 python bench.py --mode train --steps 4 --warmup 1 --no-train-parity $GEO > gpurun_out/${TAG}_train_first.json 2> gpurun_out/${TAG}_train_first.err; echo "train(first) rc=$?"
  cd /tmp
  V2V_TUNE_CACHE=/tmp/tune_train.json timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr_$TAG -o tr -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-train-parity $GEO > $R/gpurun_out/${TAG}_train.json 2> $R/gpurun_out/${TAG}_train.err; echo "train(profiled) rc=$?"
  DB=$(find /tmp/prof_tr_$TAG -name "*.db" | head -1)
  python $R/scripts/rocprof_summary.py $DB "# round 6, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 2 --no-train-parity $GEO (bf16, VGG on; tile selections replayed from a cache filled by a previous run)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train.err
  python $R/scripts/rocprof_by_grid.py $DB > $R/gpurun_out/${TAG}_train_by_grid.txt 2>> $R/gpurun_out/${TAG}_train.err
  python $R/scripts/rocprof_gaps.py $DB > $R/gpurun_out/${TAG}_train_gaps.txt 2>> $R/gpurun_out/${TAG}_train.err
  head -50 $R/gpurun_out/${TAG}_train_gaps.txt | cut -c1-160
  head -45 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-200
  head -60 $R/gpurun_out/${TAG}_train_by_grid.txt | cut -c1-200
  python -c "
import json
for f in ('first', ''):
    j = json.load(open('$R/gpurun_out/${TAG}_train' + ('_first' if f else '') + '.json')); print('train', f or 'profiled', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config'].get('autotune_s'), j['roofline'].get('conv_launches_per_step'))"
  cd $R
  lap trainprof
fi
if has traingraph; then   # whole-chunk hipGraphs (vid2vid_amd/graphed.py): eager vs graph on this box, then the capturable-Adam test
  for flag in "" "--train-graph"; do
    timeout 900 python bench.py --mode train --steps 12 --warmup 3 --no-train-parity $flag $GEO 2>gpurun_out/${TAG}_traingraph${flag}.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train [$flag]:', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', j['roofline'].get('frac'), j['config'].get('train_graph'), 'loss_G', j['config'].get('loss_G'), j['config'].get('loss_D'), 'peak GB', j['config'].get('peak_memory_gb'))"
    tail -4 gpurun_out/${TAG}_traingraph${flag}.err | cut -c1-300
  done 2>&1 | tee gpurun_out/${TAG}_traingraph.txt
  timeout 300 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short -k "adam" -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tail -12
  lap traingraph
fi
if has rawbf16; then     # persistent single-chunk tiles with bf16 raw output: kernel tests, then both resolutions with the switch off / on, same box
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "persistent or paired_x or conv2d_every_tile or fused_norm_pair or pair_equals or bn_apply" -p no:cacheprovider > gpurun_out/${TAG}_rawbf16_tests.log 2>&1; echo "rawbf16 tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_rawbf16_tests.log | cut -c1-300 | tail -20
  for rep in 1 2; do
    for v in 0 1; do
      V2V_RAW_BF16_ONE=$v timeout 900 python bench.py $LEAN 2>gpurun_out/${TAG}_rawbf16_$v.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2V_RAW_BF16_ONE=$v run $rep: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), 'frames/s')"
      python - <<PY
import json
j = json.load(open("bench_full.json"))
h = j.get("hires", {})
pk = h.get("roofline", {}).get("per_kernel_ms", {})
print("   hires per-kernel ms:", {k: v for k, v in list(pk.items())[:6]}, "| hires bf16 parity:", (h.get("parity") or {}).get("bf16"))
PY
    done
  done 2>&1 | tee gpurun_out/${TAG}_rawbf16.txt
  lap rawbf16
fi
if has wgradbench; then
  timeout 300 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short -s -k "nine_tap" -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  |nine-tap" | cut -c1-300 | tail -20
  timeout 300 python scripts/wgrad_bench.py 2>&1 | tee gpurun_out/${TAG}_wgrad_bench.txt | cut -c1-250
  lap wgradbench
fi
if has trainops; then
  timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short -p no:cacheprovider > gpurun_out/${TAG}_trainops.log 2>&1; echo "trainops rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_trainops.log | cut -c1-300 | tail -30
  lap trainops
fi
if has wgradpmc; then   # where do the nine-tap weight-gradient kernel's wave cycles go?  (one counter group per pass; 1024 -> 1024 at 64x32)
  cd /tmp
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/wg3pmc_$tag -o pmc -- python $R/scripts/wgrad3_run.py 1024 1024 32 64 12 1 > $R/gpurun_out/${TAG}_wgradpmc_$tag.log 2>&1; echo "wgradpmc $tag rc=$?"
    python $R/scripts/pmc_summary.py $(find /tmp/wg3pmc_$tag -name "*.db" | head -1) "# rocprofv3 --kernel-trace --pmc $pass -- python scripts/wgrad3_run.py 1024 1024 32 64 12 1" 2>>$R/gpurun_out/${TAG}_wgradpmc_$tag.log | grep -E "^#|wgrad3x3" | cut -c1-30,96-200
  done 2>&1 | tee $R/gpurun_out/${TAG}_wgradpmc.txt
  grep "wgrad 3x3" $R/gpurun_out/${TAG}_wgradpmc_SQ_WAVE_CYCLES.log
  cd $R
  lap wgradpmc
fi
if has wgradtraffic; then   # HBM bytes per launch of the nine-tap weight-gradient kernel (1024 -> 1024 at 64x32, accumulate): FETCH_SIZE / WRITE_SIZE in separate passes
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/wg3tr_$c -o pmc -- python $R/scripts/wgrad3_run.py 1024 1024 32 64 12 1 > $R/gpurun_out/${TAG}_wgradtraffic_$c.log 2>&1; echo "wgradtraffic $c rc=$?"
  done
  python - <<PY
import glob, json, sqlite3
def avg(pat, counter):
    db = glob.glob("/tmp/wg3tr_%s/**/*.db" % pat, recursive=True)[0]
    c = sqlite3.connect(db)
    r = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like '%wgrad3x3%' group by kernel_name", (counter,)).fetchall()
    return r[0]
f, w = avg("FETCH_SIZE", "FETCH_SIZE"), avg("WRITE_SIZE", "WRITE_SIZE")
res = {"kernel": f[0], "layer": [1024, 1024, 32, 64], "dispatches": f[1], "fetch_kib_per_launch_reported": f[2], "write_kib_per_launch_reported": w[2],
       "hbm_bytes_per_launch": (2.0 * f[2] + w[2]) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over scripts/wgrad3_run.py 1024 1024 32 64 12 1 (cold operands, accumulate = 1); FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE as reported; units KiB"}
json.dump(res, open("$R/gpurun_out/${TAG}_wgrad_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
  cd $R
  lap wgradtraffic
fi
if has trainab; then     # one-stream vs side-stream weight gradients, GEMM-view vs nine-tap kernel, alternating on this box
  for rep in 1 2; do
    for cfg in ${ABCFGS:-V2V_BWD_PATCH=0 V2V_BWD_PATCH=1}; do
      cfg=$(echo $cfg | tr ',' ' ')
      env $cfg timeout 600 python bench.py --mode train --steps 9 --warmup 3 --no-train-parity $GEO 2>gpurun_out/${TAG}_trainab.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train [$cfg] run $rep:', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', 'host issue', j['config'].get('host_issue_ms_per_step'), 'loss_G', j['config'].get('loss_G'), j['config'].get('loss_D'))"
    done
  done 2>&1 | tee gpurun_out/${TAG}_trainab.txt
  tail -3 gpurun_out/${TAG}_trainab.err | cut -c1-300
  lap trainab
fi
if has train; then       # the three training geometries of the default line, timed only
  for geo in "" "--width 1024 --height 512 --scales 2 --num-D 3" "--width 2048 --height 1024 --scales 3 --num-D 4 --frames-per-gpu 1"; do
    timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-train-parity $geo 2>gpurun_out/${TAG}_train_t.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $geo:', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', j['roofline'].get('frac'))"
  done 2>&1 | tee gpurun_out/${TAG}_train_values.txt
  lap train
fi
if has benchdefault; then    # the driver's command, timed; the stdout line must be the LAST line of a 10 KB tail and parse
  TB=$(date +%s)
  timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  cp bench_full.json gpurun_out/${TAG}_bench_default_full.json 2>/dev/null
  python - <<PY
import json
t = open("gpurun_out/${TAG}_bench_default_line.json").read()
print("stdout bytes", len(t), "lines", t.count("\n"))
j = json.loads(t[-10000:].strip().splitlines()[-1])
print(json.dumps(j)[:6000])
PY
  tail -3 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap benchdefault
fi
if has alltests; then
  timeout 1700 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -20
  lap alltests
fi
if has quicktests; then   # everything but the full-size CPU-oracle cases
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 300 -k "not full_size and not full_width" > gpurun_out/${TAG}_pytest_quick.log 2>&1; echo "pytest(quick) rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_quick.log | tail -20
  lap quicktests
fi
