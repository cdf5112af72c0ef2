#!/bin/bash
# Round-3 GPU visits (branch r3-prep: first validation of what was staged at the end of round 2).  scripts/gpu_r3.sh <tag> [parts...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3}; shift
WHAT=${*:-kpairs}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
if has kpairs; then      # tiles 90 / 91 (K pairs): parity of the single, paired and fused-norm forms, then the kernel time beside tile 82
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "conv3x3_patch_kernel or conv2d_pair or fused_norm_pair" > gpurun_out/${TAG}_kpairs_tests.log 2>&1; echo "kpairs tests rc=$?"
  tail -5 gpurun_out/${TAG}_kpairs_tests.log | cut -c1-300
  lap kpairs_tests
  cd /tmp
  for cfg in 82,1 90,1 92,1; do
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kp_${TAG}_${cfg/,/_} -o kp -- python $R/scripts/conv_layer_run.py --pair --fused --cfg $cfg,0 --reps 40 > /dev/null 2>&1
    python $R/scripts/rocprof_summary.py $(find /tmp/kp_${TAG}_${cfg/,/_} -name "*.db" | head -1) "# paired fused 1024->1024 3x3 @64x32, tile $cfg, cold cache" | grep -E "conv3x3_pp3|^#" | cut -c1-200
  done | tee $R/gpurun_out/${TAG}_kpairs_kernel.txt
  cd $R
  lap kpairs_kernel
fi
if has epilogue; then    # SGPR pins + reciprocal division: parity of every generic tile / output mode, then the fixed cost again
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -m gpu -q -x --tb=short -k "conv2d_all_output_modes or conv2d_every_tile_config or conv_transpose or convtranspose or backward" > gpurun_out/${TAG}_epilogue_tests.log 2>&1; echo "epilogue tests rc=$?"
  tail -5 gpurun_out/${TAG}_epilogue_tests.log | cut -c1-300
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_fc_$TAG -o fc -- python $R/scripts/fixed_cost.py > /dev/null 2>&1
  python $R/scripts/fixed_cost_trace.py $(find /tmp/prof_fc_$TAG -name "*.db" | head -1) | tee $R/gpurun_out/${TAG}_fixed_cost_trace.txt
  cd $R
  lap epilogue
fi
if has benchab; then     # the frame with the committed cache (pairs on tile 82) against pairs forced to tile 90, one box
  for pt in "" "90,1" "92,1"; do
    V2V_PAIR_TILE=$pt timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read()); print('pair tile ${pt:-cache}:', j['value'], 'frames/s', j['ms_per_step'], 'ms', j['roofline']['kernel'][:70], j['roofline']['avg_launch_us'], 'us')"
  done | tee gpurun_out/${TAG}_kpairs_bench_ab.txt
  lap benchab
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 --durations=10 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -30
  lap tests
fi
if has r3new; then      # round-3 additions: ADVICE fixes, barrier give-up report, full-width training parity
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 -s --durations=8 \
      -k "few_classes or follow_a_fused_optimizer or barrier_timeout or two_optimizers or full_width_training or fused_norm_pair" > gpurun_out/${TAG}_r3new.log 2>&1; echo "r3new rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|training chunk|forward|losses|grads|^E  " gpurun_out/${TAG}_r3new.log | cut -c1-1200 | tail -60
  lap r3new
fi
if has benchdefault; then    # the driver's command, timed
  TB=$(date +%s)
  timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_bench_default.json"))
print("value", j["value"], "ms", j["ms_per_step"], "windows", j["timing"]["windows_ms_per_step"])
print("parity fp32", j["parity"]["fp32_max_rel"], "bf16", j["parity"]["bf16_max_rel"], j["parity"]["bf16_mean_rel"])
print("roofline", j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"])
print("hires", json.dumps(j.get("hires"))[:1500])
print("train", json.dumps(j.get("train"))[:3000])
PY
  tail -5 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap benchdefault
fi
if has retune; then     # new tile selections (K-pair tiles eligible) for 512x256 AND the 2048x1024 / 3-scale companion, one cache
  rm -f gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 1500 python bench.py --retune --no-cpu-baseline --no-train-line --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_bench_retune.json 2> gpurun_out/${TAG}_bench_retune.err; echo "retune rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_retune.json'))
print('value', j['value'], j['ms_per_step'], j['timing']['windows_ms_per_step']); print(j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['avg_launch_us'])
print('hires', j['hires']['value'], j['hires']['ms_per_step'], j['hires']['plan_build_s'], j['hires']['roofline']['slowest_configs_ms'])"
  tail -3 gpurun_out/${TAG}_bench_retune.err | cut -c1-300
  lap retune
  cp gpurun_out/${TAG}_tune.json /tmp/tune_new.json
  V2V_TUNE_CACHE=/tmp/tune_new.json timeout 900 python bench.py --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_replay.json 2> gpurun_out/${TAG}_bench_replay.err; echo "replay rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_replay.json'))
print('replay value', j['value'], j['ms_per_step'], j['timing']['windows_ms_per_step']); print(j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['avg_launch_us'])
print('hires', j['hires']['value'], j['hires']['ms_per_step'], j['hires']['plan_build_s'])"
  lap replay
fi
if has corr; then       # matrix-pipe correlation: parity, then the kernel and FlowNet2 under rocprofv3, then the train line
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -k "correlation or flownet2 or tensor2flow or grad_scale_is_the_mean" > gpurun_out/${TAG}_corr_tests.log 2>&1; echo "corr tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_corr_tests.log | cut -c1-400 | tail -20
  lap corr_tests
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr_$TAG -o tr -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-train-parity > $R/gpurun_out/${TAG}_train.json 2> $R/gpurun_out/${TAG}_train.err; echo "train rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_tr_$TAG -name "*.db" | head -1) "# round 3, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 2 --no-train-parity (bf16, 512x256, VGG on; autotune launches of the first chunks included)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train.err
  head -30 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-220
  python -c "
import json; j = json.load(open('$R/gpurun_out/${TAG}_train.json')); print('train', j['value'], j['ms_per_step'], j['roofline']['frac']); print(j['flownet2'])"
  cd $R
  lap corr_train
fi
if has hiresops; then    # per-op tables of both resolutions with a given tune cache (TUNE=path inside the repo)
  cp ${TUNE:-profiles/tune_cache.json} /tmp/tune_ops.json
  V2V_TUNE_CACHE=/tmp/tune_ops.json timeout 900 python bench.py --no-cpu-baseline --no-train-line --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_bench_ops.json 2> gpurun_out/${TAG}_bench_ops.err; echo "ops rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_ops.json'))
print('value', j['value'], j['ms_per_step']); print('hires', j['hires']['value'], j['hires']['ms_per_step'], j['hires']['plan_build_s'])"
  cp /tmp/tune_ops.json gpurun_out/${TAG}_tune_after.json
  lap hiresops
fi
if has x3; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -s -k "x3_conv_groups or full_size_512x256" > gpurun_out/${TAG}_x3_tests.log 2>&1; echo "x3 tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  |rel err" gpurun_out/${TAG}_x3_tests.log | cut -c1-400 | tail -20
  lap x3_tests
  cp ${TUNE:-profiles/tune_cache.json} /tmp/tune_x3.json
  V2V_TUNE_CACHE=/tmp/tune_x3.json timeout 900 python bench.py --no-hires --no-train-line > gpurun_out/${TAG}_bench_x3.json 2> gpurun_out/${TAG}_bench_x3.err; echo "bench rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_x3.json'))
print('value', j['value'], j['ms_per_step']); print('fp32', j['fp32']); print('x3', json.dumps(j['x3'])[:1200])"
  tail -3 gpurun_out/${TAG}_bench_x3.err | cut -c1-300
  cp /tmp/tune_x3.json gpurun_out/${TAG}_tune_after.json
  lap x3_bench
fi
if has hirestune; then   # the isolated tile search is noisy at 2048x1024: search it NRUN times on top of the base cache (only NEW keys are measured), keep every result
  for i in 1 2 3; do
    cp ${TUNE:-profiles/tune_cache.json} /tmp/tune_h$i.json
    V2V_TUNE_CACHE=/tmp/tune_h$i.json timeout 600 python bench.py --width 2048 --height 1024 --scales 3 --no-cpu-baseline --no-train-line --steps 20 > gpurun_out/${TAG}_hires_$i.json 2> gpurun_out/${TAG}_hires_$i.err; echo "hires run $i rc=$?"
    python -c "
import json; j = json.load(open('gpurun_out/${TAG}_hires_$i.json')); print('run $i:', j['value'], 'frames/s', j['ms_per_step'], 'ms', j['config'].get('frame_tune'))"
    cp /tmp/tune_h$i.json gpurun_out/${TAG}_tune_h$i.json
  done
  lap hirestune
fi
if has pooltest; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -k "pooled or encode_labels or three_scales or full_size_2048 or avgpool or pyr" > gpurun_out/${TAG}_pool_tests.log 2>&1; echo "pool tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_pool_tests.log | cut -c1-400 | tail -20
  lap pooltest
fi
if has headhc; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -k "conv7x7 or merged_heads or three_scales or onehot_stem" > gpurun_out/${TAG}_head_tests.log 2>&1; echo "head tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_head_tests.log | cut -c1-400 | tail -20
  lap headtests
  python - <<'PY'
import sqlite3, glob
for db in glob.glob('/tmp/prof_*/**/*.db', recursive=True)[:1]:
    c = sqlite3.connect(db); print(db, [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()])
PY
fi
if has lazytest; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -k "conv7x7_head or three_scales or full_size_2048 or training_chunk_vs_reference or inference" > gpurun_out/${TAG}_lazy_tests.log 2>&1; echo "lazy tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_lazy_tests.log | cut -c1-400 | tail -20
  lap lazytest
fi
if has pooledonly; then
  timeout 600 python -m pytest tests -m gpu -q --tb=short -k "encode_labels_pooled" 2>&1 | tail -3
  lap pooledonly
fi
if has corrpmc; then     # the correlation kernel: duration (kernel trace) and fabric traffic (separate PMC passes), bf16 and fp32
  cd /tmp
  for pr in bf16 fp32; do
    timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/cr_$pr -o cr -- python $R/scripts/corr_run.py --precision $pr > /tmp/cr_$pr.log 2>&1
    python $R/scripts/rocprof_summary.py $(find /tmp/cr_$pr -name "*.db" | head -1) "# v2v_correlation_nhwc, $pr, N=2 C=256 32x64 -> 441 channels, cold cache" | grep -E "correlation|^#"
    tail -1 /tmp/cr_$pr.log
  done > $R/gpurun_out/${TAG}_correlation_profile.txt 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/cp_$ctr -o cp -- python $R/scripts/corr_run.py > /dev/null 2>&1
    python $R/scripts/pmc_kernel.py $(find /tmp/cp_$ctr -name "*.db" | head -1) $ctr correlation_mma
  done >> $R/gpurun_out/${TAG}_correlation_profile.txt 2>&1
  cat $R/gpurun_out/${TAG}_correlation_profile.txt | cut -c1-220
  cd $R
  lap corrpmc
fi
if has final; then       # evidence for the committed line: in-graph duration + PMC traffic of the dominant tile, copied where bench.py looks, then the driver's command
  DOM=${DOM:-91,1,2} WGS=256 NEEDLE=${NEEDLE:-conv3x3_pp3_kernelIDF16bLi4ELi64ELi64ELi4ELi0ELi4ELi1ELi2} bash scripts/gpu_r2.sh ${TAG} prof2
  cp gpurun_out/${TAG}_in_graph.json profiles/${TAG}_in_graph.json; cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json
  TB=$(date +%s)
  timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_default.json'))
print('value', j['value'], j['ms_per_step'], j['timing']['windows_ms_per_step']); r = j['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us'], r['in_graph'], r['traffic'])
print('x3', j['x3']['value'], j['x3']['max_rel'], 'fp32', j['fp32']['value'], 'hires', j['hires']['value'], j['hires']['parity']['fp32_max_rel'], 'train', j['train']['value'], j['train']['parity']['fp32_ok'])"
  lap final
fi
if has headab; then      # 7x7 heads: 128-byte patch rows (two workgroups per CU) against 64-byte rows for every stride (four per CU)
  for hc in 0 1; do
    echo "V2V_HEAD_HC=$hc"; V2V_HEAD_HC=$hc timeout 200 python scripts/head_bench.py 2>/dev/null
  done | tee gpurun_out/${TAG}_head_hc_ab.txt
  V2V_HEAD_HC=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv7x7 or merged_heads" 2>&1 | tail -2
  for hc in 0 1; do
    cp profiles/tune_cache.json /tmp/tune_hc$hc.json
    V2V_HEAD_HC=$hc V2V_TUNE_CACHE=/tmp/tune_hc$hc.json timeout 600 python bench.py --no-cpu-baseline --no-train-line 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('V2V_HEAD_HC=$hc: 512x256', j['value'], 'fps', j['ms_per_step'], 'ms;  2048x1024', j['hires']['value'], 'fps', j['hires']['ms_per_step'], 'ms')"
  done | tee -a gpurun_out/${TAG}_head_hc_ab.txt
  lap headab
fi
if has smoke; then
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
  tail -1 gpurun_out/${TAG}_smoke.log
  lap smoke
fi
if has configs; then     # the other BASELINE configurations on one GPU: configs[2] geometry (1024x512 train, 2 scales, num_D=3) and configs[3] (edge2face 512x512)
  cp profiles/tune_cache.json /tmp/tune_cfg.json
  V2V_TUNE_CACHE=/tmp/tune_cfg.json timeout 900 python bench.py --mode train --width 1024 --height 512 --scales 2 --num-D 3 --frames-per-gpu 1 --steps 6 --warmup 2 > gpurun_out/${TAG}_train_1024_s2.json 2> gpurun_out/${TAG}_train_1024_s2.err; echo "train 1024 rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_train_1024_s2.json')); print('train 1024x512 S=2 num_D=3:', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', j['roofline']['frac']); p = j['parity']; print('parity fp32_ok', p.get('fp32_ok'), p['fp32']['max_forward'], p['fp32']['max_loss'], p['fp32']['max_grad_norm'], 'oracle', p['oracle_seconds'], 's')"
  lap train1024
  V2V_TUNE_CACHE=/tmp/tune_cfg.json timeout 600 python bench.py --dataset edge2face --width 512 --height 512 --cpu-frames 1 > gpurun_out/${TAG}_bench_edge2face.json 2> gpurun_out/${TAG}_bench_edge2face.err; echo "edge2face rc=$?"
  python -c "
import json; j = json.load(open('gpurun_out/${TAG}_bench_edge2face.json')); print('edge2face 512x512:', j['value'], 'frames/s', j['ms_per_step'], 'ms', j['roofline']['kernel'][:60], j['roofline']['frac']); print('parity fp32', j['parity']['fp32_max_rel'], 'bf16', j['parity']['bf16_max_rel'], 'fp32 line', j['fp32'])"
  lap edge2face
fi
if has c8; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "conv7x7" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-300 | tail -12
  lap c8test
fi
if has subset; then
  timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=short -k "three_scales or inference or composite or first_frame or local" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-300 | tail -8
  lap subset
fi
