#!/bin/bash
# GPU visit (round 1, v17+): parity suite, smoke, inference bench (tune -> traffic PMC passes -> official line with
# traffic), rocprofv3 kernel stats of the tuned inference bench, training bench (+VGG, + C3 geometry) with its own
# rocprofv3 kernel stats, edge2face 512x512 and 2048x1024 S=3 inference lines.
#   scripts/gpu_visit4.sh <tag> [parts...]     parts: tests smoke bench traffic prof train trainprof train1024 face big
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-v}; shift
WHAT=${*:-tests smoke bench traffic prof profeager train trainprof train1024 face big}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
TUNE=$R/gpurun_out/${TAG}_tune.json
if has tests; then
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 180 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -20
  tail -30 gpurun_out/${TAG}_pytest_gpu.log > gpurun_out/${TAG}_pytest_gpu_tail.txt
  lap tests
fi
if has smoke; then
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
  tail -1 gpurun_out/${TAG}_smoke.log
  lap smoke
fi
if has bench; then
  rm -f $TUNE
  V2V_TUNE_CACHE=$TUNE timeout 500 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dump-ops gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_bench_tune.json 2> gpurun_out/${TAG}_bench_tune.err; echo "bench(tune) rc=$?"
  cut -c1-300 gpurun_out/${TAG}_bench_tune.json; tail -2 gpurun_out/${TAG}_bench_tune.err
  lap bench-tune
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
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tr_f -o pmc -- python $R/scripts/conv_layer_run.py --cfg $CFG > $R/gpurun_out/${TAG}_traffic_fetch.log 2>&1; echo "traffic fetch rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tr_w -o pmc -- python $R/scripts/conv_layer_run.py --cfg $CFG > $R/gpurun_out/${TAG}_traffic_write.log 2>&1; echo "traffic write rc=$?"
  python $R/scripts/pmc_traffic.py $(find /tmp/tr_f -name "*.db" | head -1) $(find /tmp/tr_w -name "*.db" | head -1) $CFG $R/gpurun_out/${TAG}_traffic.json | cut -c1-300
  cp $R/gpurun_out/${TAG}_traffic.json $R/profiles/r01_${TAG}_traffic.json 2>/dev/null
  cd $R
  lap traffic
fi
if has bench; then
  # the official line: replays the tile selections of the tuning run, carries roofline.traffic and the CPU baseline
  V2V_TUNE_CACHE=$TUNE timeout 500 python bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; echo "bench rc=$?"
  cut -c1-1500 gpurun_out/${TAG}_bench_bf16.json; tail -2 gpurun_out/${TAG}_bench_bf16.err
  lap bench
fi
if has prof; then
  cd /tmp
  V2V_TUNE_CACHE=$TUNE timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.json 2> $R/gpurun_out/${TAG}_bench_prof.err; echo "rocprof rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) "# round 1, visit $TAG: V2V_TUNE_CACHE=<selections of the preceding bench run> rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline (bf16, 512x256; no autotune launches in this trace)" > $R/gpurun_out/${TAG}_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_bench_prof.err
  head -12 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-180
  cd $R
  lap prof
fi
if has profeager; then
  # the same selections replayed WITHOUT the graph (one stream, kernels alone on the chip): per-kernel durations that
  # the HIP-event figures of roofline.achieved must agree with
  cd /tmp
  V2V_TUNE_CACHE=$TUNE timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/profe_$TAG -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph > $R/gpurun_out/${TAG}_bench_prof_eager.json 2> $R/gpurun_out/${TAG}_bench_prof_eager.err; echo "rocprof(eager) rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/profe_$TAG -name "*.db" | head -1) "# round 1, visit $TAG: V2V_TUNE_CACHE=<selections of the preceding bench run> rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph (bf16, 512x256; eager single-stream replay: every kernel alone on the chip)" > $R/gpurun_out/${TAG}_kernel_stats_eager.txt 2>> $R/gpurun_out/${TAG}_bench_prof_eager.err
  head -8 $R/gpurun_out/${TAG}_kernel_stats_eager.txt | cut -c1-180
  cut -c1-200 $R/gpurun_out/${TAG}_bench_prof_eager.json
  cd $R
  lap profeager
fi
TTUNE=$R/gpurun_out/${TAG}_tune_train.json
if has train; then
  timeout 300 python bench.py --mode train --width 128 --height 64 --steps 3 --warmup 0 --no-autotune > gpurun_out/${TAG}_train_tiny.json 2> gpurun_out/${TAG}_train_tiny.err; echo "train tiny rc=$?"
  cut -c1-400 gpurun_out/${TAG}_train_tiny.json; tail -3 gpurun_out/${TAG}_train_tiny.err
  lap train-tiny
  rm -f $TTUNE
  V2V_TUNE_CACHE=$TTUNE timeout 600 python bench.py --mode train --steps 12 --warmup 3 --with-vgg > gpurun_out/${TAG}_train_512_vgg_bf16.json 2> gpurun_out/${TAG}_train_512_vgg_bf16.err; echo "train 512 (+vgg) rc=$?"
  cut -c1-1600 gpurun_out/${TAG}_train_512_vgg_bf16.json; tail -3 gpurun_out/${TAG}_train_512_vgg_bf16.err
  lap train-512-vgg
fi
if has trainnovgg; then
  V2V_TUNE_CACHE=$TTUNE timeout 400 python bench.py --mode train --steps 12 --warmup 3 > gpurun_out/${TAG}_train_512_novgg_bf16.json 2> gpurun_out/${TAG}_train_512_novgg_bf16.err; echo "train 512 (--no_vgg) rc=$?"
  cut -c1-600 gpurun_out/${TAG}_train_512_novgg_bf16.json; tail -3 gpurun_out/${TAG}_train_512_novgg_bf16.err
  lap train-512
fi
if has trainprof; then
  cd /tmp
  V2V_TUNE_CACHE=$TTUNE timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/proft_$TAG -o bench -- python $R/bench.py --mode train --steps 6 --warmup 3 --with-vgg > $R/gpurun_out/${TAG}_train_prof.json 2> $R/gpurun_out/${TAG}_train_prof.err; echo "rocprof train rc=$?"
  python $R/scripts/rocprof_summary.py $(find /tmp/proft_$TAG -name "*.db" | head -1) "# round 1, visit $TAG: V2V_TUNE_CACHE=<selections of the preceding run> rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 6 --warmup 3 --with-vgg (bf16, 512x256, 2 frames per chunk)" > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>> $R/gpurun_out/${TAG}_train_prof.err
  head -16 $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-180
  cd $R
  lap trainprof
fi
if has train1024; then
  timeout 600 python bench.py --mode train --width 1024 --height 512 --scales 2 --num-D 3 --frames-total 4 --frames-per-gpu 1 --steps 8 --warmup 2 --with-vgg > gpurun_out/${TAG}_train_1024_s2_bf16.json 2> gpurun_out/${TAG}_train_1024_s2_bf16.err; echo "train 1024 rc=$?"
  cut -c1-1200 gpurun_out/${TAG}_train_1024_s2_bf16.json; tail -3 gpurun_out/${TAG}_train_1024_s2_bf16.err
  lap train-1024
fi
if has face; then
  timeout 500 python bench.py --dataset edge2face --width 512 --height 512 --steps 20 --warmup 3 --cpu-frames 1 > gpurun_out/${TAG}_bench_edge2face_512_bf16.json 2> gpurun_out/${TAG}_bench_edge2face_512_bf16.err; echo "edge2face rc=$?"
  cut -c1-700 gpurun_out/${TAG}_bench_edge2face_512_bf16.json; tail -2 gpurun_out/${TAG}_bench_edge2face_512_bf16.err
  lap face
fi
if has big; then
  timeout 700 python bench.py --steps 10 --warmup 3 --width 2048 --height 1024 --scales 3 --no-cpu-baseline --dump-ops gpurun_out/${TAG}_ops_2048_bf16.json > gpurun_out/${TAG}_bench_2048_bf16.json 2> gpurun_out/${TAG}_bench_2048_bf16.err; echo "bench2048 rc=$?"
  cut -c1-900 gpurun_out/${TAG}_bench_2048_bf16.json; tail -2 gpurun_out/${TAG}_bench_2048_bf16.err
  lap big
fi
