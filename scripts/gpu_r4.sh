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
print("per_kernel_ms", r.get("per_kernel_ms"))
for k in ("c1", "hires", "train", "train_hires"):
    print(k, json.dumps(j.get(k))[:2500])
PY
  tail -5 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap benchdefault
fi
if has fillcache; then   # configs[2] / configs[3] legs of the default line: their shapes enter the tile cache (copied to profiles/tune_cache.json afterwards)
  cp profiles/tune_cache.json gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 1200 python bench.py --no-hires --no-train-line --no-train-hires --no-c1 > gpurun_out/${TAG}_bench_c3c4.json 2> gpurun_out/${TAG}_bench_c3c4.err; echo "fillcache rc=$?"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_bench_c3c4.json"))
print({k: v for k, v in j.items() if not isinstance(v, (dict, list))})
for k in ("c4", "train_c3"):
    print(k, json.dumps(j.get(k))[:3000])
PY
  tail -5 gpurun_out/${TAG}_bench_c3c4.err | cut -c1-300
  lap fillcache
fi
if has c3test; then     # the 3-frame C3 training chunk (teacher-forced) + the role split with real networks
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 1400 -s --durations=5 \
      -k "training_chunk_1024x512 or role_split_with_real" > gpurun_out/${TAG}_c3test.log 2>&1; echo "c3test rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|training chunk|forward|grads|free-running|^E  " gpurun_out/${TAG}_c3test.log | cut -c1-1200 | tail -30
  lap c3test
fi
if has heads; then      # conv7x7_rowsum_kernel (tile 62): parity, the kernel beside tile 60 on the head shapes, then both resolutions with / without it
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "rowsum or conv7x7_head_kernel or merged_heads" > gpurun_out/${TAG}_heads_tests.log 2>&1; echo "heads tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_heads_tests.log | cut -c1-300 | tail -20
  timeout 600 python scripts/head_bench.py 2>&1 | tee gpurun_out/${TAG}_head_bench.txt | cut -c1-250
  for hr in 1 0 1 0; do
    V2V_HEAD_ROWSUM=$hr timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>gpurun_out/${TAG}_heads_$hr.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']
print('V2V_HEAD_ROWSUM=$hr: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms conv', j['roofline']['per_kernel_ms'].get('conv_igemm'), '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms conv', h['roofline']['per_kernel_ms'].get('conv_igemm'), h['roofline']['slowest_configs_ms'])"
  done | tee gpurun_out/${TAG}_heads_ab.txt
  tail -3 gpurun_out/${TAG}_heads_1.err | cut -c1-300
  lap heads
fi
if has epi; then        # the vectorised conv epilogue (every RAW / ACT output of every tile): the whole suite minus the CPU-oracle-heavy tests
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 -k "not full_width_training and not full_size" > gpurun_out/${TAG}_epi_tests.log 2>&1; echo "epi tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_epi_tests.log | cut -c1-400 | tail -30
  lap epi
fi
if has benchq; then     # quick A/B figure: both resolutions, no CPU legs, no training line
  timeout 900 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_benchq.json 2> gpurun_out/${TAG}_benchq.err; echo "benchq rc=$?"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_benchq.json"))
print("512x256", j["value"], "fps", j["ms_per_step"], "ms", j["timing"]["windows_ms_per_step"])
print(" per_kernel_ms", j["roofline"]["per_kernel_ms"], "eager sum", j["roofline"]["frame_ms_eager_events"])
h = j["hires"]; print("2048x1024", h["value"], "fps", h["ms_per_step"], "ms; eager sum", h["roofline"]["frame_ms_eager_events"], h["roofline"]["slowest_configs_ms"])
print(" per_kernel_ms", h["roofline"]["per_kernel_ms"])
print(" hires roofline", h["roofline"]["kernel"], h["roofline"]["bound"], h["roofline"]["frac"], h["roofline"]["frame_vs_per_layer_bounds"])
PY
  tail -3 gpurun_out/${TAG}_benchq.err | cut -c1-300
  lap benchq
fi
if has phases; then     # per-workgroup phase stamps of the stride-2 / transposed / ResnetBlock launches
  timeout 600 python scripts/kernel_phases.py > gpurun_out/${TAG}_kernel_phases.txt 2> gpurun_out/${TAG}_kernel_phases.err; echo "phases rc=$?"
  cut -c1-330 gpurun_out/${TAG}_kernel_phases.txt; tail -3 gpurun_out/${TAG}_kernel_phases.err
  lap phases
fi
if has benchpar; then   # 512x256 with the CPU-oracle parity legs only (bf16 error of the benchmarked path), no companions
  timeout 900 python bench.py --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 --no-hires > gpurun_out/${TAG}_benchpar.json 2> gpurun_out/${TAG}_benchpar.err; echo "benchpar rc=$?"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_benchpar.json"))
print("512x256", j["value"], "fps", j["ms_per_step"], "ms")
p = j["parity"]; print(" fp32", p["fp32_max_rel"], "bf16", p["bf16"], "drift", p["free_running_drift"])
print(" x3", j["x3"]["value"], j["x3"]["max_rel"], "fp32", j["fp32"]["value"])
PY
  lap benchpar
fi
if has sqpmc; then      # SQ counters of the dominant tile (paired 1024->1024 3x3 + fused norm, tile 91): MFMA busy cycles, LDS bank conflicts, wave / wait cycles (separate passes)
  cd /tmp
  NEEDLE=${NEEDLE:-conv3x3_pp3_kernel}
  P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
  P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS"
  P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
  : > $R/gpurun_out/${TAG}_sq_counters.txt
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/sq_${TAG}_$i -o pmc -- python $R/scripts/conv_layer_run.py --pair --fused --cfg ${CFG:-91,1},0 --reps 20 > /tmp/sq_${TAG}_$i.log 2>&1; echo "sq pass $i rc=$?"
    python $R/scripts/pmc_all.py $(find /tmp/sq_${TAG}_$i -name "*.db" | head -1) $NEEDLE "pass $i: rocprofv3 --kernel-trace --pmc $P -- python scripts/conv_layer_run.py --pair --fused --cfg ${CFG:-91,1},0 --reps 20 (paired 1024->1024 3x3 @64x32 + fused norm, bf16, cold cache)" >> $R/gpurun_out/${TAG}_sq_counters.txt 2>&1
  done
  cat $R/gpurun_out/${TAG}_sq_counters.txt | cut -c1-200
  cd $R
  lap sqpmc
fi
if has trainprof; then  # rocprofv3 table of the training step (bf16, 512x256, VGG on)
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr_$TAG -o tr -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-train-parity > $R/gpurun_out/${TAG}_train.json 2> $R/gpurun_out/${TAG}_train.err; echo "train rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_tr_$TAG -name "*.db" | head -1) "# round 4, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 2 --no-train-parity (bf16, 512x256, VGG on; autotune launches of the first chunks included)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train.err
  head -40 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-220
  python -c "
import json; j = json.load(open('$R/gpurun_out/${TAG}_train.json')); print('train', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['gflop_per_step_by_kind']); print(j['flownet2'])"
  cd $R
  lap trainprof
fi
if has s2test; then     # stride-2 patch kernel: parity, then the kernel beside the generic tiles on the three main-tower shapes
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "stride2_patch or transpose_stride2_patch" > gpurun_out/${TAG}_s2_tests.log 2>&1; echo "s2 tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_s2_tests.log | cut -c1-300 | tail -20
  lap s2test
fi
if has s2bench; then
  timeout 600 python scripts/s2_bench.py 2>&1 | tee gpurun_out/${TAG}_s2_bench.txt | cut -c1-250
  lap s2bench
fi
if has retune; then     # new tile selections (stride-2 / transposed patch tiles eligible) for 512x256 AND the 2048x1024 companion, one cache; then a replay
  rm -f gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 1700 python bench.py --retune --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_bench_retune.json 2> gpurun_out/${TAG}_bench_retune.err; echo "retune rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_retune.json'))
print('retune value', j['value'], j['ms_per_step'], j['timing']['windows_ms_per_step'], 'eager sum', j['roofline']['frame_ms_eager_events']); print(j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['avg_launch_us'])
print('frame tune', j['config'].get('frame_tune'))
print('hires', j['hires']['value'], j['hires']['ms_per_step'], j['hires']['plan_build_s'], j['hires']['roofline']['slowest_configs_ms'])"
  tail -3 gpurun_out/${TAG}_bench_retune.err | cut -c1-300
  cp gpurun_out/${TAG}_tune.json /tmp/tune_new.json
  for i in 1 2; do
  V2V_TUNE_CACHE=/tmp/tune_new.json timeout 900 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']
print('replay(new cache): 512x256', j['value'], 'fps', j['ms_per_step'], 'ms eager', j['roofline']['frame_ms_eager_events'], '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms eager', h['roofline']['frame_ms_eager_events'])"
  timeout 900 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']
print('replay(committed cache): 512x256', j['value'], 'fps', j['ms_per_step'], 'ms eager', j['roofline']['frame_ms_eager_events'], '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms eager', h['roofline']['frame_ms_eager_events'])"
  done
  lap retune
fi
if has x3tune; then     # tile selections of the x3 plan (bf16 sub-engine, K tripled) measured on top of the committed cache, whole-frame search included
  cp profiles/tune_cache.json /tmp/tune_x3.json
  V2V_TUNE_CACHE=/tmp/tune_x3.json timeout 900 python bench.py --precision x3 --no-cpu-baseline --no-hires --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 > gpurun_out/${TAG}_bench_x3_tune.json 2> gpurun_out/${TAG}_bench_x3_tune.err; echo "x3 tune rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_x3_tune.json')); print('x3 (tuning run)', j['value'], 'fps', j['ms_per_step'], 'ms', j['config'].get('frame_tune'))"
  cp /tmp/tune_x3.json gpurun_out/${TAG}_tune_x3.json
  for i in 1 2; do
    V2V_TUNE_CACHE=/tmp/tune_x3.json timeout 900 python bench.py --precision x3 --no-cpu-baseline --no-hires --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('x3 replay(new cache)', j['value'], 'fps', j['ms_per_step'], 'ms launches', j['config']['launches_per_frame'])"
  done
  timeout 900 python bench.py --precision x3 --no-cpu-baseline --no-hires --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('x3 replay(committed cache + in-run search of the missing shapes)', j['value'], 'fps', j['ms_per_step'], 'ms')"
  lap x3tune
fi
if has bwdpatch; then   # backward-data on the patch kernels: parity, then the training step with / without them (tile search of the first sequence each time)
  timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short -k "backward_data_on_the_patch or conv2d_backward or conv_transpose2d_backward" > gpurun_out/${TAG}_bwd_tests.log 2>&1; echo "bwd tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_bwd_tests.log | cut -c1-300 | tail -20
  for pt in 1 0 1 0; do
    V2V_S2_PATCH=$pt V2V_T2_PATCH=$pt timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-train-parity 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('patch kernels in the tile search = $pt: train', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', j['roofline']['frac'], 'autotune', j['config']['autotune_s'], 's')"
  done | tee gpurun_out/${TAG}_train_patch_ab.txt
  lap bwdpatch
fi
if has fin2; then       # two-level in-kernel finalize: parity, then both resolutions with / without it on one box
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "two_level_in_kernel" > gpurun_out/${TAG}_fin2_tests.log 2>&1; echo "fin2 tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_fin2_tests.log | cut -c1-300 | tail -20
  for f2 in 1 0 1 0; do
    V2V_FIN2=$f2 timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>gpurun_out/${TAG}_fin2_$f2.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']; k = h['roofline']['per_kernel_ms']
print('V2V_FIN2=$f2: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms launches', j['config']['launches_per_frame'], '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms launches', h['launches_per_frame'], 'bn_finalize', k.get('bn_finalize'), 'bn_partial_reduce', k.get('bn_partial_reduce'), 'conv', k.get('conv_igemm'))"
  done | tee gpurun_out/${TAG}_fin2_ab.txt
  lap fin2
fi
if has rawab; then      # bf16 raw tensors on / off on ONE box: both resolutions + the bf16 error of the 512x256 frame
  for rb in 1 0 1 0; do
    V2V_RAW_BF16=$rb timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']
print('V2V_RAW_BF16=$rb: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms bn_apply', j['roofline']['per_kernel_ms'].get('bn_apply'), 'conv', j['roofline']['per_kernel_ms'].get('conv_igemm'), '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms bn_apply', h['roofline']['per_kernel_ms'].get('bn_apply'), 'conv', h['roofline']['per_kernel_ms'].get('conv_igemm'))"
  done | tee gpurun_out/${TAG}_raw_bf16_ab.txt
  lap rawab
fi
if has trainprof2; then # steady-state table of the training step: a first run fills the tile cache, the profiled second run replays it (no autotune launches in the trace)
  rm -f /tmp/tune_train.json
  V2V_TUNE_CACHE=/tmp/tune_train.json timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-train-parity > gpurun_out/${TAG}_train_first.json 2> gpurun_out/${TAG}_train_first.err; echo "train(first) rc=$?"
  cd /tmp
  V2V_TUNE_CACHE=/tmp/tune_train.json timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr2_$TAG -o tr -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-train-parity > $R/gpurun_out/${TAG}_train.json 2> $R/gpurun_out/${TAG}_train.err; echo "train(profiled) rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_tr2_$TAG -name "*.db" | head -1) "# round 4, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 2 --no-train-parity (bf16, 512x256, VGG on, 2 frames per chunk; tile selections replayed from a cache filled by a previous run; 3 tuning-sequence chunks + 2 warm-up + 6 timed chunks in the trace)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train.err
  head -60 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-200
  python -c "
import json
for f in ('first', ''):
    j = json.load(open('$R/gpurun_out/${TAG}_train' + ('_first' if f else '') + '.json')); print('train', f or 'profiled', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['autotune_s'], j['roofline']['conv_launches_per_step'])"
  cd $R
  lap trainprof2
fi
if has roles; then
  timeout 600 python -m pytest tests/test_gpu_roles.py -m gpu -q -rf --tb=short -s > gpurun_out/${TAG}_roles.log 2>&1; echo "roles rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|max \|d\||^E  " gpurun_out/${TAG}_roles.log | cut -c1-300 | tail -40
  lap roles
fi
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short --timeout 1500 --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -30
  lap tests
fi
if has dbg1; then
  for envs in ${DBGENVS:-"X=1" "V2V_FUSED_NORM=0" "V2V_TWIN=0" "V2V_LANES=0" "V2V_RAW_BF16=0"}; do
    echo "== $envs"
    env $envs timeout 300 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=line -k "inference_api_vs_reference or flownet2_vs_reference" 2>&1 | grep -E "passed|failed|Error|error" | cut -c1-250
  done
  lap dbg1
fi
if has kernarg; then    # where kernel arguments live: HIP_FORCE_DEV_KERNARG (device memory) on / off, same box
  for ka in 1 0 1 0; do
    HIP_FORCE_DEV_KERNARG=$ka timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']
print('HIP_FORCE_DEV_KERNARG=$ka: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms eager sum', j['roofline']['frame_ms_eager_events'], '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms')"
  done | tee gpurun_out/${TAG}_kernarg_ab.txt
  lap kernarg
fi
if has coresident; then  # can the foreground tower run UNDER the paired launches?  pairs on a 128 KiB tile, the tower's 512->512 convolutions on the 32 KiB generic tile
  run() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 --no-hires 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); r = j['roofline']
print('$*:', j['value'], 'fps', j['ms_per_step'], 'ms eager sum', r['frame_ms_eager_events'], '| dominant', r['kernel'][:60], 'in-graph live', (r.get('in_graph_live') or {}).get('avg_launch_us'), 'eager', r['eager']['avg_launch_us'])"; }
  for rep in 1 2; do
    run X=base
    run V2V_PAIR_TILE=80,1
    run V2V_PAIR_TILE=94,1
    run V2V_PAIR_TILE=80,1 "V2V_TILE_OVERRIDE=512,512,3,1,0:10,1,0"
    run V2V_PAIR_TILE=94,1 "V2V_TILE_OVERRIDE=512,512,3,1,0:10,1,0"
    run "V2V_TILE_OVERRIDE=512,512,3,1,0:10,1,0"
  done | tee gpurun_out/${TAG}_coresident_ab.txt
  lap coresident
fi
if has xcdgrp; then     # paired launches with the members on disjoint XCD halves (grouped_xcd_map) + the row-GEMM heads (tile 62): parity, A/B on one box, fabric traffic
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "conv2d_pair or fused_norm_pair or rowsum or conv7x7_head_kernel or merged_heads" > gpurun_out/${TAG}_xcdgrp_tests.log 2>&1; echo "xcdgrp tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_xcdgrp_tests.log | cut -c1-300 | tail -20
  lap xcdtests
  timeout 300 python scripts/head_bench.py 2>&1 | tee gpurun_out/${TAG}_head_bench.txt | cut -c1-250
  lap headbench
  run() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>gpurun_out/${TAG}_xcdgrp.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); r = j['roofline']; h = j['hires']
print('$*: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms | dominant in-graph live', (r.get('in_graph_live') or {}).get('avg_launch_us'), 'eager', r['eager']['avg_launch_us'], 'conv', r['per_kernel_ms'].get('conv_igemm'), '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms conv', h['roofline']['per_kernel_ms'].get('conv_igemm'), h['roofline']['slowest_configs_ms'])"; }
  for rep in 1 2; do
    run V2V_GROUP_XCD=0 V2V_HEAD_ROWSUM=0
    run V2V_GROUP_XCD=1 V2V_HEAD_ROWSUM=0
    run V2V_GROUP_XCD=1 V2V_HEAD_ROWSUM=1
  done | tee gpurun_out/${TAG}_xcdgrp_ab.txt
  tail -3 gpurun_out/${TAG}_xcdgrp.err | cut -c1-300
  lap xcdab
  cd /tmp
  for g in 1 0; do
    V2V_GROUP_XCD=$g timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/trf_$g -o pmc -- python $R/scripts/conv_layer_run.py --pair --fused --cfg 91,1,0 > $R/gpurun_out/${TAG}_traffic_fetch_$g.log 2>&1; echo "traffic fetch (grp_xcd=$g) rc=$?"
    V2V_GROUP_XCD=$g timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/trw_$g -o pmc -- python $R/scripts/conv_layer_run.py --pair --fused --cfg 91,1,0 > $R/gpurun_out/${TAG}_traffic_write_$g.log 2>&1; echo "traffic write (grp_xcd=$g) rc=$?"
    python $R/scripts/pmc_traffic.py $(find /tmp/trf_$g -name "*.db" | head -1) $(find /tmp/trw_$g -name "*.db" | head -1) 91,1,2 $R/gpurun_out/${TAG}_traffic_grpxcd$g.json | cut -c1-500
  done
  cd $R
  lap xcdtraffic
fi
if has traintests; then  # the kernels behind the training step that changed this visit: weight re-pack (every conv test packs), bn backward reduce / apply
  timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short > gpurun_out/${TAG}_traintests.log 2>&1; echo "traintests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_traintests.log | cut -c1-300 | tail -20
  lap traintests
fi
if has trainab; then    # training step A/B by environment (one box): weight-gradient K split target, channels-last master weights
  rm -f /tmp/tune_train.json
  V2V_TUNE_CACHE=/tmp/tune_train.json timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-train-parity > /dev/null 2>&1
  runt() { env "$@" V2V_TUNE_CACHE=/tmp/tune_train.json timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-train-parity 2>gpurun_out/${TAG}_trainab.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('$*: train', j['value'], 'frames/s', j['ms_per_step'], 'ms per chunk')"; }
  for rep in 1 2; do
    for envs in ${TRAINAB:-"X=base" "V2V_WGRAD_WGS=512" "V2V_WEIGHTS_CL=1" "V2V_WEIGHTS_CL=1,V2V_WGRAD_WGS=512"}; do
      runt $(echo $envs | tr ',' ' ')
    done
  done | tee gpurun_out/${TAG}_trainab.txt
  tail -3 gpurun_out/${TAG}_trainab.err | cut -c1-300
  lap trainab
fi
if has onechunk; then   # single-chunk ping-pong tiles 58 / 59: parity, then beside the other tiles on the fine-scale shapes
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "conv3x3_patch_kernel" > gpurun_out/${TAG}_one_tests.log 2>&1; echo "onechunk tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_one_tests.log | cut -c1-300 | tail -20
  timeout 300 python scripts/one_bench.py 2>&1 | tee gpurun_out/${TAG}_one_bench.txt | cut -c1-400
  lap onechunk
fi
if has finfused; then   # second finalize stage inside bn_partial_reduce (one launch fewer per large layer): parity, then 2048x1024 with / without on one box
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "bn_finalize or in_kernel_finalize or two_level" > gpurun_out/${TAG}_finfused_tests.log 2>&1; echo "finfused tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_finfused_tests.log | cut -c1-300 | tail -20
  for f in 1 0 1 0; do
    V2V_BN_FIN_FUSED=$f timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>gpurun_out/${TAG}_finfused.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); h = j['hires']; pk = h['roofline']['per_kernel_ms']
print('V2V_BN_FIN_FUSED=$f: 512x256', j['value'], 'fps | 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms | bn_partial_reduce', pk.get('bn_partial_reduce'), 'bn_finalize', pk.get('bn_finalize'), 'parity', (h.get('parity') or {}).get('fp32_max_rel'))"
  done | tee gpurun_out/${TAG}_finfused_ab.txt
  tail -3 gpurun_out/${TAG}_finfused.err | cut -c1-300
  lap finfused
fi
if has stagger; then    # single-chunk tiles: start-up stagger of the second workgroup slot (V2V_ONE_STAGGER x ~4096 cycles)
  for st in 0 1 2 3 4 6; do
    echo "V2V_ONE_STAGGER=$st"; V2V_ONE_STAGGER=$st ONE_TILES=80,94,95 timeout 200 python scripts/one_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done | tee gpurun_out/${TAG}_stagger.txt
  lap stagger
fi
if has final; then      # evidence for the committed line: in-graph duration (rocprofv3 kernel trace of the bench command) + PMC traffic of the dominant paired tile, copied where bench.py looks, then the driver's command
  DOM=${DOM:-90,1,2} WGS=256 NEEDLE=${NEEDLE:-conv3x3_pp3_kernelIDF16bLi8ELi32ELi64ELi5ELi0ELi4ELi1ELi2} bash scripts/gpu_r2.sh ${TAG} prof2
  cp gpurun_out/${TAG}_in_graph.json profiles/${TAG}_in_graph.json; cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json
  TB=$(date +%s)
  timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_bench_default.json"))
print({k: v for k, v in j.items() if not isinstance(v, (dict, list))})
r = j["roofline"]
print("roofline", r["kernel"], r["frac"], r["avg_launch_us"], "traffic", r["traffic"], "eager", r.get("eager"), "rocprof", r.get("in_graph_rocprof"))
print("per_kernel_ms", r.get("per_kernel_ms"))
for k in ("x3", "fp32", "c1", "hires", "train", "train_hires", "c4", "train_c3", "cpu_baseline"):
    print(k, json.dumps(j.get(k))[:1200])
PY
  tail -3 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap final
fi
if has fastep; then     # full-tile fast paths of the shared conv epilogue: bit-exactness tests, then the frame at both resolutions (compare with the previous visit's numbers of the same box class)
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "conv3x3_patch_kernel or conv2d_pair or fused_norm_pair or in_kernel_norm_finalize or two_level" > gpurun_out/${TAG}_fastep_tests.log 2>&1; echo "fastep tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_fastep_tests.log | cut -c1-300 | tail -20
  timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=short -k "inference_api_vs_reference or composite_generator" 2>&1 | tail -3
  for f in 1 0 1 0; do
  V2V_EPILOGUE_FAST=$f timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 2>gpurun_out/${TAG}_fastep.err | python -c "
import sys, json; j = json.loads(sys.stdin.read()); r = j['roofline']; h = j['hires']
print('V2V_EPILOGUE_FAST=$f: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms | dominant in-graph live', (r.get('in_graph_live') or {}).get('avg_launch_us'), 'eager', r['eager']['avg_launch_us'], 'conv', r['per_kernel_ms'].get('conv_igemm'), '| 2048x1024', h['value'], 'fps', h['ms_per_step'], 'ms conv', h['roofline']['per_kernel_ms'].get('conv_igemm'))"
  done | tee gpurun_out/${TAG}_fastep_ab.txt
  timeout 200 python scripts/one_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee gpurun_out/${TAG}_fastep_one_bench.txt
  lap fastep
fi
if has evidence2; then  # end-of-round evidence: per-op dumps of both resolutions (per-layer roofline tables) and the steady-state training table
  timeout 600 python bench.py --no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3 --dump-ops gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_benchq.json 2> gpurun_out/${TAG}_benchq.err; echo "benchq rc=$?"
  python scripts/per_layer_roofline.py gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_per_layer_roofline.txt 2>&1; tail -3 gpurun_out/${TAG}_per_layer_roofline.txt | cut -c1-300
  python scripts/per_layer_roofline_hires.py gpurun_out/${TAG}_ops_bf16.json.hires.json > gpurun_out/${TAG}_per_layer_roofline_hires.txt 2>&1; tail -3 gpurun_out/${TAG}_per_layer_roofline_hires.txt | cut -c1-400
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_benchq.json')); print('512x256', j['value'], j['ms_per_step'], 'eager sum', j['roofline']['frame_ms_eager_events'], '| 2048x1024', j['hires']['value'], j['hires']['ms_per_step'], 'eager sum', j['hires']['roofline']['frame_ms_eager_events'])"
  lap ops
fi
