#!/bin/bash
# Round-2 GPU visits.  scripts/gpu_r2.sh <tag> [parts...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r2}; shift
WHAT=${*:-newtests}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
if has newtests; then
  timeout 1500 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 900 -s --durations=8 \
      -k "full_size or three_scales or feature_encoding" > gpurun_out/${TAG}_newtests.log 2>&1; echo "newtests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|rel err|Error|error" gpurun_out/${TAG}_newtests.log | cut -c1-400 | tail -40
  lap newtests
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 --durations=10 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300 | tail -30
  tail -40 gpurun_out/${TAG}_pytest_gpu.log > gpurun_out/${TAG}_pytest_gpu_tail.txt
  lap tests
fi
if has smoke; then
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
  tail -1 gpurun_out/${TAG}_smoke.log
  lap smoke
fi
if has kernels; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "patch_kernel or conv2d_pair or norm" > gpurun_out/${TAG}_kernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/${TAG}_kernels.log | cut -c1-300 | tail -30
  lap kernels
fi
if has pp2bench; then
  timeout 600 python scripts/pp2_bench.py bf16 > gpurun_out/${TAG}_pp2_bench.txt 2>&1; echo "pp2 bench rc=$?"
  cat gpurun_out/${TAG}_pp2_bench.txt | cut -c1-200
  lap pp2bench
fi
if has golden; then
  timeout 1200 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -30
  lap golden
fi
if has twinbench; then
  for tw in 1 0; do
    V2V_TWIN=$tw timeout 500 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_twin$tw.json 2> gpurun_out/${TAG}_bench_twin$tw.err; echo "bench twin=$tw rc=$?"
    cut -c1-220 gpurun_out/${TAG}_bench_twin$tw.json; grep -E "frame tune|Error|error" gpurun_out/${TAG}_bench_twin$tw.err | tail -3
  done
  lap twinbench
fi
if has ablate; then
  timeout 300 python scripts/pp2_ablate.py > gpurun_out/${TAG}_pp2_ablate.txt 2>&1; echo "pp2 ablate rc=$?"
  cat gpurun_out/${TAG}_pp2_ablate.txt | cut -c1-160
  lap ablate
fi
if has pairtest; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "conv2d_pair" > gpurun_out/${TAG}_pairtest.log 2>&1; echo "pair test rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_pairtest.log | cut -c1-300 | tail -20
  lap pairtest
fi
if has benchnew; then
  rm -f gpurun_out/${TAG}_tune.json
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune.json timeout 900 python bench.py --steps 30 --warmup 5 --retune --dump-ops gpurun_out/${TAG}_ops.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
  cut -c1-3000 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
  lap benchnew
fi
if has allkernels; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short --timeout 300 > gpurun_out/${TAG}_allkernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/${TAG}_allkernels.log | cut -c1-300 | tail -30
  lap allkernels
fi
if has prof2; then
  # tile selections of profiles/tune_cache.json are replayed by default: the profile describes the benchmarked kernels
  DOM=${DOM:-82,1,2}; NEEDLE=${NEEDLE:-conv3x3_pp3_kernelIDF16bLi8ELi32ELi64ELi5}
  cd /tmp
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-line ${PROF_FLAGS:---no-train-hires --no-c1 --no-c4 --no-train-c3 --no-hires} > $R/gpurun_out/${TAG}_bench_prof.json 2> $R/gpurun_out/${TAG}_bench_prof.err; echo "rocprof rc=$?"
  DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
  python $R/scripts/rocprof_summary.py $DB "# round 2, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline (bf16, 512x256; tile selections replayed from profiles/tune_cache.json, no autotune launches in this trace; kernels run inside the 3-lane frame graph)" > $R/gpurun_out/${TAG}_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_bench_prof.err
  head -14 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
  python $R/scripts/in_graph_json.py $DB $NEEDLE $DOM $R/gpurun_out/${TAG}_in_graph.json ${WGS:-0}
  cut -c1-300 $R/gpurun_out/${TAG}_bench_prof.json
  CFG2=$(echo $DOM | cut -d, -f1,2)
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tr_f -o pmc -- python $R/scripts/conv_layer_run.py --pair --fused --cfg $CFG2,0 > $R/gpurun_out/${TAG}_traffic_fetch.log 2>&1; echo "traffic fetch rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tr_w -o pmc -- python $R/scripts/conv_layer_run.py --pair --fused --cfg $CFG2,0 > $R/gpurun_out/${TAG}_traffic_write.log 2>&1; echo "traffic write rc=$?"
  python $R/scripts/pmc_traffic.py $(find /tmp/tr_f -name "*.db" | head -1) $(find /tmp/tr_w -name "*.db" | head -1) $DOM $R/gpurun_out/${TAG}_traffic.json | cut -c1-400
  cd $R
  lap prof2
fi
if has benchfinal; then
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err; echo "bench(final, committed tune cache) rc=$?"
  cut -c1-400 gpurun_out/${TAG}_bench_final.json; tail -3 gpurun_out/${TAG}_bench_final.err
  lap benchfinal
fi
if has newtests2; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 600 -k "visualisation or uint8 or rccl or grad_scale or fused_adam" > gpurun_out/${TAG}_newtests2.log 2>&1; echo "newtests2 rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_newtests2.log | cut -c1-300 | tail -20
  lap newtests2
fi
if has timeline; then
  cd /tmp
  timeout 500 rocprofv3 --kernel-trace -d /tmp/tl_$TAG -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-train-line > $R/gpurun_out/${TAG}_bench_tl.json 2> $R/gpurun_out/${TAG}_bench_tl.err; echo "rocprof(timeline) rc=$?"
  python $R/scripts/frame_timeline.py $(find /tmp/tl_$TAG -name "*.db" | head -1) > $R/gpurun_out/${TAG}_frame_timeline.txt 2>&1
  head -4 $R/gpurun_out/${TAG}_frame_timeline.txt
  cd $R
  lap timeline
fi
if has stem; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "onehot_stem or flownet2_native" > gpurun_out/${TAG}_stemtest.log 2>&1; echo "stem/native tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_stemtest.log | cut -c1-300 | tail -20
  timeout 300 python scripts/stem_bench.py bf16 > gpurun_out/${TAG}_stem_bench.txt 2>&1; echo "stem bench rc=$?"
  cat gpurun_out/${TAG}_stem_bench.txt | cut -c1-200
  lap stem
fi
if has ubench; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ubench/valu_rate.hip -o /tmp/valu_rate && timeout 120 /tmp/valu_rate > gpurun_out/${TAG}_valu_rate.txt 2>&1; echo "ubench rc=$?"
  cat gpurun_out/${TAG}_valu_rate.txt
  lap ubench
fi
if has stemtest; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "onehot_stem or flownet2_native" > gpurun_out/${TAG}_stemtest.log 2>&1; echo "stem/native tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_stemtest.log | cut -c1-300 | tail -20
  lap stemtest
fi
if has stemab; then
  for oh in 1 0 1 0; do
    V2V_ONEHOT_STEM=$oh timeout 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_oh$oh.json 2> gpurun_out/${TAG}_bench_oh$oh.err; echo "bench onehot=$oh rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_oh$oh.json
  done
  lap stemab
fi
if has trainprof; then
  timeout 900 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.json 2> gpurun_out/${TAG}_train.err; echo "train bench rc=$?"
  cut -c1-600 gpurun_out/${TAG}_train.json; tail -3 gpurun_out/${TAG}_train.err
  cd /tmp
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_train_tune.json timeout 900 python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1   # fills the tile cache: no autotune launches in the trace
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_train_tune.json timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/trp_$TAG -o train -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_train_prof.json 2> $R/gpurun_out/${TAG}_train_prof.err; echo "rocprof(train) rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/trp_$TAG -name "*.db" | head -1) "# round 2, visit $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 2 (bf16 512x256, VGG on)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train_prof.err
  head -45 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-220
  cd $R
  lap trainprof
fi
if has trainab; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short --timeout 300 -x > gpurun_out/${TAG}_allkernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/${TAG}_allkernels.log | cut -c1-300 | tail -10
  timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.json 2> gpurun_out/${TAG}_train.err; echo "train bench rc=$?"
  cut -c1-220 gpurun_out/${TAG}_train.json
  V2V_WGRAD_WGS=512 timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train_wg512.json 2> gpurun_out/${TAG}_train_wg512.err; echo "train bench (wgrad 512 WGs) rc=$?"
  cut -c1-220 gpurun_out/${TAG}_train_wg512.json
  V2V_BN_BWD_FUSED=0 timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train_nofuse.json 2> gpurun_out/${TAG}_train_nofuse.err; echo "train bench (separate bn_bwd_finalize) rc=$?"
  cut -c1-220 gpurun_out/${TAG}_train_nofuse.json
  lap trainab
fi
if has benchdefault; then
  timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench (defaults) rc=$?"
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_bench_default.json"))
print("value", j["value"], "ms", j["ms_per_step"], "fp32", j.get("fp32", {}).get("value"), "host_fed", j.get("host_fed"), "\ntrain", j.get("train"), "\ncpu", j.get("cpu_baseline"))
PY
  tail -3 gpurun_out/${TAG}_bench_default.err
  timeout 600 python bench.py --mode train --steps 8 --warmup 2 > gpurun_out/${TAG}_train.json 2> gpurun_out/${TAG}_train.err; echo "train bench rc=$?"
  python -c "import json; j=json.load(open('gpurun_out/${TAG}_train.json')); print(j['value'], j['ms_per_step'], j['flownet2'])"
  lap benchdefault
fi
if has fusedab; then
  timeout 120 python scripts/dbg_fused.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -x -k "fused_norm or conv2d_pair" > gpurun_out/${TAG}_fusedtest.log 2>&1; echo "fused norm tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_fusedtest.log | cut -c1-300 | tail -12
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -8
  for fu in 1 0 1 0; do
    V2V_FUSED_NORM=$fu timeout 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_fu$fu.json 2> gpurun_out/${TAG}_bench_fu$fu.err; echo "bench fused_norm=$fu rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_fu$fu.json
  done
  lap fusedab
fi
if has fusedtest; then
  timeout 120 python scripts/dbg_fused.py 2>&1 | grep -v amdgpu.ids | grep "diff" | cut -c1-300
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short --timeout 300 > gpurun_out/${TAG}_allkernels.log 2>&1; echo "kernel tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_allkernels.log | cut -c1-300 | tail -12
  lap fusedtest
fi
if has overhead; then
  timeout 300 python scripts/frame_overhead.py 2>/dev/null | tee gpurun_out/${TAG}_frame_overhead.txt
  lap overhead
fi
if has codesab; then
  for v in 1 0 3 1 0 3; do
    V2V_LABEL_CODES=$v timeout 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_codes$v.json 2> gpurun_out/${TAG}_bench_codes$v.err; echo "bench label_codes=$v rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_codes$v.json
  done
  lap codesab
fi
if has otherconfigs; then
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_2048.json timeout 900 python bench.py --width 2048 --height 1024 --scales 3 --steps 20 --warmup 4 --no-cpu-baseline --no-train-line --retune > gpurun_out/${TAG}_bench_2048_s3.json 2> gpurun_out/${TAG}_bench_2048_s3.err; echo "bench 2048x1024 S=3 rc=$?"
  cut -c1-700 gpurun_out/${TAG}_bench_2048_s3.json; tail -2 gpurun_out/${TAG}_bench_2048_s3.err
  V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_face.json timeout 900 python bench.py --dataset edge2face --width 512 --height 512 --steps 30 --warmup 5 --no-cpu-baseline --no-train-line --retune > gpurun_out/${TAG}_bench_face.json 2> gpurun_out/${TAG}_bench_face.err; echo "bench edge2face 512x512 rc=$?"
  cut -c1-700 gpurun_out/${TAG}_bench_face.json; tail -2 gpurun_out/${TAG}_bench_face.err
  lap otherconfigs
fi
if has clsab; then
  for v in 1 0; do
    V2V_MERGE_HEADS=0 V2V_CLS_ORDER=$v V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_cls$v.json timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line --retune > gpurun_out/${TAG}_bench_cls$v.json 2> gpurun_out/${TAG}_bench_cls$v.err; echo "bench cls_order=$v rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_cls$v.json; grep "frame tune" gpurun_out/${TAG}_bench_cls$v.err | tail -1
  done
  lap clsab
fi
if has headsab; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -x -k "merged_heads or head_kernel or conv_transpose or convtranspose" > gpurun_out/${TAG}_headstest.log 2>&1; echo "heads tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_headstest.log | cut -c1-300 | tail -8
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -8
  for v in 1 0 1 0; do
    V2V_MERGE_HEADS=$v timeout 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_mh$v.json 2> gpurun_out/${TAG}_bench_mh$v.err; echo "bench merge_heads=$v rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_mh$v.json
  done
  lap headsab
fi
if has w4; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -x -k "patch_kernel or conv2d_pair or fused_norm" > gpurun_out/${TAG}_w4test.log 2>&1; echo "4-wave tile tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_w4test.log | cut -c1-300 | tail -8
  timeout 600 python scripts/pp2_bench.py bf16 > gpurun_out/${TAG}_pp3_bench.txt 2>&1; echo "pp3 bench rc=$?"
  grep -v amdgpu gpurun_out/${TAG}_pp3_bench.txt | head -60 | cut -c1-120
  lap w4
fi
if has deferab; then
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -8
  for v in 1 0 1 0; do
    V2V_DEFER_ENCODE=$v timeout 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_de$v.json 2> gpurun_out/${TAG}_bench_de$v.err; echo "bench defer_encode=$v rc=$?"
    cut -c1-200 gpurun_out/${TAG}_bench_de$v.json
  done
  lap deferab
fi
if has lanetl; then
  timeout 300 python scripts/lane_timeline.py > gpurun_out/${TAG}_lane_timeline.txt 2> gpurun_out/${TAG}_lane_timeline.err; echo "lane timeline rc=$?"
  head -3 gpurun_out/${TAG}_lane_timeline.txt; tail -2 gpurun_out/${TAG}_lane_timeline.err
  lap lanetl
fi
if has graphq; then
  for q in default 1 2 3 4 8; do
    if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
    timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_q$q.json 2> gpurun_out/${TAG}_bench_q$q.err; echo "bench DEBUG_HIP_FORCE_GRAPH_QUEUES=$q rc=$?"
    cut -c1-160 gpurun_out/${TAG}_bench_q$q.json
    timeout 300 python scripts/lane_timeline.py > gpurun_out/${TAG}_lane_timeline_q$q.txt 2>/dev/null; head -2 gpurun_out/${TAG}_lane_timeline_q$q.txt | cut -c1-200; grep "down_img.4 *$" gpurun_out/${TAG}_lane_timeline_q$q.txt | head -1
  done
  unset DEBUG_HIP_FORCE_GRAPH_QUEUES
  lap graphq
fi
if has imglane; then
  for v in 1 0 1 0; do
    V2V_IMG_LANE=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_il$v.json 2> gpurun_out/${TAG}_bench_il$v.err; echo "bench img_lane=$v rc=$?"
    cut -c1-160 gpurun_out/${TAG}_bench_il$v.json
  done
  V2V_IMG_LANE=1 timeout 300 python scripts/lane_timeline.py > gpurun_out/${TAG}_lane_timeline.txt 2>/dev/null; head -2 gpurun_out/${TAG}_lane_timeline.txt | cut -c1-200; grep "down_img.4 *$\|down_seg.13.c1" gpurun_out/${TAG}_lane_timeline.txt | head -2
  lap imglane
fi
if has hwq; then
  for q in default 2 8 16; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    for il in 1 0; do
      V2V_IMG_LANE=$il timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_hwq${q}_il$il.json 2> gpurun_out/${TAG}_bench_hwq${q}_il$il.err; echo "bench GPU_MAX_HW_QUEUES=$q img_lane=$il rc=$?"
      cut -c1-160 gpurun_out/${TAG}_bench_hwq${q}_il$il.json
    done
    V2V_IMG_LANE=1 timeout 300 python scripts/lane_timeline.py > gpurun_out/${TAG}_lane_timeline_hwq$q.txt 2>/dev/null; head -1 gpurun_out/${TAG}_lane_timeline_hwq$q.txt | cut -c1-200; grep "down_img.4 *$\|down_seg.13.c1" gpurun_out/${TAG}_lane_timeline_hwq$q.txt | head -2
  done
  unset GPU_MAX_HW_QUEUES
  lap hwq
fi
if has graphlog; then
  AMD_LOG_LEVEL=4 timeout 300 python scripts/lane_timeline.py 2>&1 >/dev/null | grep -i "max_streams\|parallel streams\|max streams" | sort | uniq -c | head -10
  lap graphlog
fi
if has segab; then
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "not full_size" > gpurun_out/${TAG}_golden.log 2>&1; echo "golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_golden.log | cut -c1-300 | tail -8
  for v in segments single segments single; do
    V2V_GRAPH_MODE=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_gm$v.json 2> gpurun_out/${TAG}_bench_gm$v.err; echo "bench graph_mode=$v rc=$?"
    cut -c1-160 gpurun_out/${TAG}_bench_gm$v.json
  done
  timeout 300 python scripts/lane_timeline.py > gpurun_out/${TAG}_lane_timeline.txt 2>/dev/null; head -2 gpurun_out/${TAG}_lane_timeline.txt | cut -c1-200; grep "down_img.4 *$\|down_seg.13.c1\|warp_blend" gpurun_out/${TAG}_lane_timeline.txt | head -3
  lap segab
fi
if has hostfed; then
  for v in segments single segments single; do
    V2V_GRAPH_MODE=$v timeout 300 python bench.py --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_hf$v.json 2> gpurun_out/${TAG}_bench_hf$v.err; echo "bench graph_mode=$v rc=$?"
    python -c "import json; j=json.load(open('gpurun_out/${TAG}_bench_hf$v.json')); print(j['value'], j['host_fed']['value'])"
  done
  lap hostfed
fi
if has overhead2; then
  for v in segments single; do
    echo "== V2V_GRAPH_MODE=$v"; V2V_GRAPH_MODE=$v timeout 300 python scripts/frame_overhead.py 2>/dev/null
  done | tee gpurun_out/${TAG}_frame_overhead.txt
  lap overhead2
fi
if has fgfork; then
  for v in late early late early; do
    V2V_FG_FORK=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-line > gpurun_out/${TAG}_bench_fg$v.json 2> gpurun_out/${TAG}_bench_fg$v.err; echo "bench fg_fork=$v rc=$?"
    cut -c1-160 gpurun_out/${TAG}_bench_fg$v.json
  done
  lap fgfork
fi
if has heads2; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -x -k "head or merged" > gpurun_out/${TAG}_headstest.log 2>&1; echo "heads tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_headstest.log | cut -c1-300 | tail -6
  timeout 200 python scripts/head_bench.py 2>/dev/null | tee gpurun_out/${TAG}_head_bench.txt
  lap heads2
fi
if has cltest; then
  timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -rf --tb=short --timeout 300 -x -k "conv2d_backward or conv_transpose2d_backward or norm_act_residual or fused_adam" > gpurun_out/${TAG}_cltest.log 2>&1; echo "channels-last tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_cltest.log | cut -c1-300 | tail -8
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 600 -k "training or model_D" > gpurun_out/${TAG}_traingolden.log 2>&1; echo "training golden rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E " gpurun_out/${TAG}_traingolden.log | cut -c1-300 | tail -6
  for v in 1 0; do
    V2V_WEIGHTS_CL=$v timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train_cl$v.json 2> gpurun_out/${TAG}_train_cl$v.err; echo "train bench weights_cl=$v rc=$?"
    cut -c1-200 gpurun_out/${TAG}_train_cl$v.json
  done
  lap cltest
fi
if has benchdef1; then
  timeout 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench (defaults) rc=$?"
  cut -c1-600 gpurun_out/${TAG}_bench_default.json; tail -3 gpurun_out/${TAG}_bench_default.err
  lap benchdef1
fi
if has s2abl; then
  timeout 200 python scripts/s2_ablate.py > gpurun_out/${TAG}_s2_ablate.txt 2> gpurun_out/${TAG}_s2_ablate.err; echo "s2 ablate rc=$?"
  cat gpurun_out/${TAG}_s2_ablate.txt; tail -3 gpurun_out/${TAG}_s2_ablate.err
  lap s2abl
fi
if has s2trace; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s2_$TAG -o s2 -- python $R/scripts/s2_ablate.py > $R/gpurun_out/${TAG}_s2_ablate_host.txt 2> $R/gpurun_out/${TAG}_s2_trace.err; echo "rocprof rc=$?"
  python $R/scripts/s2_ablate_trace.py $(find /tmp/prof_s2_$TAG -name "*.db" | head -1) > $R/gpurun_out/${TAG}_s2_ablate_trace.txt 2>> $R/gpurun_out/${TAG}_s2_trace.err
  cat $R/gpurun_out/${TAG}_s2_ablate_trace.txt; tail -3 $R/gpurun_out/${TAG}_s2_trace.err
  cd $R
  lap s2trace
fi
if has s2fin; then
  for v in "1 1" "0 1" "0 0"; do set -- $v
    FIN=$1 STATS=$2 timeout 100 python scripts/s2_ablate.py 2>/dev/null | grep -v cold | cut -c1-150
  done > gpurun_out/${TAG}_s2_fin_ab.txt; cat gpurun_out/${TAG}_s2_fin_ab.txt
  lap s2fin
fi
if has fintest; then
  timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 300 \
      -k "finalize or onehot or fused_norm or batchnorm or running_stats or conv_transpose or inference_api or composite_generator or graph_replay" > gpurun_out/${TAG}_fintest.log 2>&1; echo "fin tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_fintest.log | cut -c1-300 | tail -20
  lap fintest
  FIN=1 STATS=1 timeout 100 python scripts/s2_ablate.py 2>/dev/null | grep -v cold | cut -c1-150 > gpurun_out/${TAG}_s2_fin_after.txt; cat gpurun_out/${TAG}_s2_fin_after.txt
  lap s2after
fi
if has fixedcost; then
  timeout 150 python scripts/fixed_cost.py > gpurun_out/${TAG}_fixed_cost.txt 2> gpurun_out/${TAG}_fixed_cost.err; echo "fixed cost rc=$?"
  cat gpurun_out/${TAG}_fixed_cost.txt; tail -3 gpurun_out/${TAG}_fixed_cost.err
  lap fixedcost
fi
if has fixedtrace; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_fc_$TAG -o fc -- python $R/scripts/fixed_cost.py > $R/gpurun_out/${TAG}_fixed_cost_host.txt 2> $R/gpurun_out/${TAG}_fc_trace.err; echo "rocprof rc=$?"
  python $R/scripts/fixed_cost_trace.py $(find /tmp/prof_fc_$TAG -name "*.db" | head -1) > $R/gpurun_out/${TAG}_fixed_cost_trace.txt 2>> $R/gpurun_out/${TAG}_fc_trace.err
  cat $R/gpurun_out/${TAG}_fixed_cost_trace.txt
  cd $R
  lap fixedtrace
fi
if has fixedtrace3; then
  cd /tmp
  for v in "0 0" "0 1"; do set -- $v
    FIN=$1 STATS=$2 timeout 100 rocprofv3 --kernel-trace -d /tmp/prof_fc_${TAG}_$1$2 -o fc -- python $R/scripts/fixed_cost.py > /dev/null 2>> $R/gpurun_out/${TAG}_fc_trace.err
    echo "FIN=$1 STATS=$2" >> $R/gpurun_out/${TAG}_fixed_cost_trace3.txt
    python $R/scripts/fixed_cost_trace.py $(find /tmp/prof_fc_${TAG}_$1$2 -name "*.db" | head -1) | grep -v "^#" >> $R/gpurun_out/${TAG}_fixed_cost_trace3.txt 2>> $R/gpurun_out/${TAG}_fc_trace.err
  done
  cat $R/gpurun_out/${TAG}_fixed_cost_trace3.txt
  cd $R
  lap fixedtrace3
fi
if has quickcheck; then
  timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 300 -x \
      -k "finalize or onehot or fused_norm or conv_transpose or inference_api or graph_replay or conv2d_pair" > gpurun_out/${TAG}_quick.log 2>&1; echo "quick tests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_quick.log | cut -c1-300 | tail -8
  lap quickcheck
fi
if has pixcheck; then
  timeout 60 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider \
      -k "conv_transpose2d or convtranspose_splitk or (conv2d_all_output_modes and bf16)" > gpurun_out/${TAG}_pixcheck.log 2>&1; echo "pix tests rc=$?"
  tail -3 gpurun_out/${TAG}_pixcheck.log | cut -c1-200
  lap pixcheck
fi
