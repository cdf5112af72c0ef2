#!/bin/bash
# Round-5 GPU visits.  scripts/gpu_r5.sh <tag> [parts...]   (every part writes under gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5}; shift
WHAT=${*:-alltests}
has() { [[ " $WHAT " == *" $1 "* ]]; }
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
LEAN="--no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3"
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
if has ab; then    # two BUILDS of the library alternating on this box: round-4 main (libv2v_hip_r4main.so) vs the tree's build
  for i in 1 2 3; do
    for libn in libv2v_hip_r4main.so libv2v_hip.so; do
      V2V_LIB_PATH=$R/vid2vid_amd/$libn timeout 300 python bench.py $LEAN --no-hires 2>gpurun_out/${TAG}_ab.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$libn run $i: 512x256', j['value'], 'frames/s', j['ms_per_step'], 'ms; dominant', j['roofline']['kernel'][:40], j['roofline'].get('eager_us'), 'us eager')"
    done
  done 2>&1 | tee gpurun_out/${TAG}_ab.txt
  lap ab
fi
if has hires; then   # both resolutions with the 7x7-window tiles offered to the search: the cache copy forgets the dense 7x7 stems (raw output, whole-chunk
                     # channel stride), everything else replays profiles/tune_cache.json; compare with the default line of this visit
  python - <<PY
import json
d = json.load(open("profiles/tune_cache.json"))
for dt, v in d.items():
    for k in [k for k in v if k.split(",")[2] == "7" and k.split(",")[8] == "0" and int(k.split(",")[9]) % 64 == 0]:
        del v[k]
json.dump(d, open("gpurun_out/${TAG}_tune_s7.json", "w"))
PY
  V2V_S7_PATCH=1 V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_s7.json timeout 900 python bench.py $LEAN 2>gpurun_out/${TAG}_hires_s7.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2V_S7_PATCH=1', {k: v for k, v in j.items() if not isinstance(v, (dict, list))})"
  cp bench_full.json gpurun_out/${TAG}_hires_s7_full.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_tune_s7.json"))
for dt, v in d.items():
    print(dt, {k: v[k] for k in v if k.split(",")[2] == "7" and k.split(",")[8] == "0" and int(k.split(",")[9]) % 64 == 0})
PY
  lap hires
fi
if has stem7; then
  timeout 300 python scripts/stem7_bench.py 2>&1 | tee gpurun_out/${TAG}_stem7_bench.txt | cut -c1-260
  timeout 300 python scripts/one_bench.py 2>&1 | tee gpurun_out/${TAG}_one_bench.txt | cut -c1-300
  lap stem7
fi
if has stamp; then   # the V2V_STAMP_MASK=0x7f build -- which DETERMINISTIC kernel tests fail with it?
  export V2V_LIB_PATH=$R/vid2vid_amd/libv2v_hip_stamp.so
  timeout 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=line --timeout 120 -k "fp32 or not bf16" -p no:cacheprovider > gpurun_out/${TAG}_stamp_kernels.log 2>&1; echo "stamp kernels rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_stamp_kernels.log | cut -c1-260 | tail -40
  for i in 1 2 3; do
    timeout 120 python -m pytest tests/test_gpu_golden.py -m gpu -q -rf --tb=line -k "inference_api_vs_reference or flownet2_vs_reference" -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|Error" | cut -c1-260
  done 2>&1 | tee gpurun_out/${TAG}_stamp_golden.txt
  unset V2V_LIB_PATH
  lap stamp
fi
if has alltests; then
  timeout 1700 python -m pytest tests -m gpu -q -rf --tb=short --timeout 900 --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -20
  lap alltests
fi
if has quicktests; then   # everything but the full-size CPU-oracle cases (those are ~11 of the suite's 13 minutes)
  timeout 900 python -m pytest tests -m gpu -q -rf --tb=short --timeout 300 -k "not full_size and not full_width" > gpurun_out/${TAG}_pytest_quick.log 2>&1; echo "pytest(quick) rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_quick.log | tail -20
  lap quicktests
fi
if has dom; then     # the dominant pair launch -- ring depth, one barrier per two steps, static priority, the two-sequence geometry
  timeout 300 python scripts/dom_bench.py 2>&1 | tee gpurun_out/${TAG}_dom_bench.txt | cut -c1-200
  timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "fused_norm_pair or pair_equals or conv3x3_patch_kernel" -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tail -12
  lap dom
fi
if has frameexp; then   # both resolutions with the round-5 experiment tiles offered to the tile searches (V2V_EXP_TILES=1): the cache copy forgets the dominant
                        # pair and the single-chunk 64 -> 64 layers (and, with S7=1, the dense 7x7 stems); baseline first, alternating twice
  python - <<PY
import json
d = json.load(open("profiles/tune_cache.json"))
drop7 = "${S7:-1}" == "1"
for dt, v in d.items():
    for k in list(v):
        f = k.split(",")
        if k.startswith("-2,1024,1024,1,32,64") or (f[0] == "64" and f[1] == "64" and f[2] == "3" and f[3] == "1" and f[5] == "1") or \
           (drop7 and f[2] == "7" and f[8] == "0" and int(f[9]) % 64 == 0):
            del v[k]
json.dump(d, open("gpurun_out/${TAG}_tune_exp.json", "w"))
PY
  for i in 1 2; do
    timeout 600 python bench.py $LEAN 2>gpurun_out/${TAG}_frame_base.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('baseline run $i:', j['value'], 'frames/s |', j.get('hires_value'), 'frames/s @2048x1024 | dominant eager', j['roofline'].get('eager_us'), 'us live', j['roofline'].get('in_graph_live_us'))"
    V2V_EXP_TILES=1 V2V_S7_PATCH=${S7:-1} V2V_TUNE_CACHE=$R/gpurun_out/${TAG}_tune_exp.json timeout 900 python bench.py $LEAN 2>gpurun_out/${TAG}_frame_exp.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('experiment tiles run $i:', j['value'], 'frames/s |', j.get('hires_value'), 'frames/s @2048x1024 | dominant', j['roofline']['kernel'][:70], 'eager', j['roofline'].get('eager_us'), 'us live', j['roofline'].get('in_graph_live_us'))"
    cp bench_full.json gpurun_out/${TAG}_frame_exp_full.json
  done 2>&1 | tee gpurun_out/${TAG}_frameexp.txt
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_tune_exp.json"))
b = json.load(open("profiles/tune_cache.json"))
for dt, v in d.items():
    print(dt, {k: (b.get(dt, {}).get(k), v[k]) for k in v if b.get(dt, {}).get(k) != v[k]})
PY
  lap frameexp
fi
if has mfma; then    # what the whole chip shares when every CU runs the matrix pipe: pure-register MFMA stream on 64 .. 512 workgroups
  timeout 120 scripts/mfma_rate.bin 2>&1 | tee gpurun_out/${TAG}_mfma_rate.txt
  lap mfma
fi
if has tcc; then     # L2 hit / miss / fabric read requests of the 1024 -> 1024 kernel on 128 workgroups (N = 1) and 256 (N = 2)
  cd /tmp
  for b in 1 2; do
    for pass in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
      tag=$(echo $pass | cut -d' ' -f1)
      timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/tcc_${b}_$tag -o pmc -- python $R/scripts/conv_layer_run.py --cfg 90,1,0 --batch $b --reps 12 > $R/gpurun_out/${TAG}_tcc_${b}_$tag.log 2>&1; echo "tcc batch $b $tag rc=$?"
      python $R/scripts/pmc_summary.py $(find /tmp/tcc_${b}_$tag -name "*.db" | head -1) "# batch $b (tile 90, $((128 * b)) workgroups): rocprofv3 --kernel-trace --pmc $pass -- python scripts/conv_layer_run.py --cfg 90,1,0 --batch $b" 2>>$R/gpurun_out/${TAG}_tcc_${b}_$tag.log | grep -E "^#|conv3x3" | cut -c1-20,60-200
    done
  done 2>&1 | tee $R/gpurun_out/${TAG}_tcc.txt
  cd $R
  lap tcc
fi
if has stampdiag; then   # the stamp build's flaky golden tests -- WHICH tile selections do the failing runs replay?
  for i in 1 2 3 4 5 6 7 8; do
    rm -f /tmp/stamp_tune_$i.json
    V2V_LIB_PATH=$R/vid2vid_amd/libv2v_hip_stamp.so V2V_TUNE_CACHE=/tmp/stamp_tune_$i.json timeout 120 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=line -k "inference_api_vs_reference or flownet2_vs_reference" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -1 > /tmp/stamp_res_$i.txt
    echo "stamp run $i: $(cat /tmp/stamp_res_$i.txt)"; cp /tmp/stamp_tune_$i.json gpurun_out/${TAG}_stamp_tune_$i.json 2>/dev/null
  done 2>&1 | tee gpurun_out/${TAG}_stampdiag.txt
  # replay every run's selections with the PRODUCT build: does a selection that failed on the stamp build fail here too?
  for i in 1 2 3 4 5 6 7 8; do
    cp /tmp/stamp_tune_$i.json /tmp/replay_$i.json 2>/dev/null || continue
    echo "product build replaying the selections of stamp run $i ($(cat /tmp/stamp_res_$i.txt)): $(V2V_TUNE_CACHE=/tmp/replay_$i.json timeout 120 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=line -k 'inference_api_vs_reference or flownet2_vs_reference' -p no:cacheprovider 2>&1 | grep -E 'passed|failed' | tail -1)"
    echo "stamp build replaying its own selections of run $i: $(V2V_LIB_PATH=$R/vid2vid_amd/libv2v_hip_stamp.so V2V_TUNE_CACHE=/tmp/replay_$i.json timeout 120 python -m pytest tests/test_gpu_golden.py -m gpu -q --tb=line -k 'inference_api_vs_reference or flownet2_vs_reference' -p no:cacheprovider 2>&1 | grep -E 'passed|failed' | tail -1)"
  done 2>&1 | tee -a gpurun_out/${TAG}_stampdiag.txt
  lap stampdiag
fi
if has newtests; then   # the tests that failed in visit 1 only because tiles 97-99 were not reachable + the new tiles
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "conv3x3_patch_kernel or pair_equals or fused_norm_pair or persistent_single_chunk or conv7x7_window" -p no:cacheprovider > gpurun_out/${TAG}_newtests.log 2>&1; echo "newtests rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_newtests.log | cut -c1-300 | tail -25
  lap newtests
fi
if has onebench; then
  timeout 300 python scripts/one_bench.py 2>&1 | tee gpurun_out/${TAG}_one_bench.txt | cut -c1-400
  lap onebench
fi
if has stampbisect; then   # which (shape -> tile, split-K) entry makes the stamp build fail?  (scripts/stamp_bisect.py)
  timeout 600 python scripts/stamp_bisect.py profiles/r05_v2_stamp_tune_fail.json profiles/r05_v2_stamp_tune_pass.json 2>&1 | tee gpurun_out/${TAG}_stamp_bisect.txt | cut -c1-300
  lap stampbisect
fi
if has trainsplit; then   # is model_final_flow's gradient error (4x the fp32 oracle's own) the length of the fp32 accumulation chains of the
                          # split-K weight-gradient kernel?  Same chunk, 8x more K splits (V2V_WGRAD_WGS): per-tensor table of both runs
  for wgs in 1024 8192; do
    V2V_WGRAD_WGS=$wgs timeout 600 python bench.py --mode train --steps 2 --warmup 1 --no-autotune > gpurun_out/${TAG}_trainsplit_$wgs.json 2> gpurun_out/${TAG}_trainsplit_$wgs.err; echo "train (V2V_WGRAD_WGS=$wgs) rc=$?"
    cp bench_full.json gpurun_out/${TAG}_trainsplit_${wgs}_full.json
    python - <<PY
import json
f = json.load(open("gpurun_out/${TAG}_trainsplit_${wgs}_full.json"))
p = f["parity"]["fp32"]
print("V2V_WGRAD_WGS=$wgs: grads", {k: (v["norm_rel_err"], v["l2_rel_err"]) for k, v in p["grads"].items()})
g = p["grad_error_by_tensor"]["G"]
print("   G by module", {m: v for m, v in list(g["by_module"].items())[:4]})
print("   top", [(t["name"], t["share"], t["rel_err"]) for t in g["top_tensors"][:3]])
PY
  done 2>&1 | tee gpurun_out/${TAG}_trainsplit.txt
  lap trainsplit
fi
if has final; then      # evidence for the committed line: in-graph duration (rocprofv3 kernel trace of the bench command) + PMC traffic of the dominant paired
                        # tile, copied where bench.py looks, then the driver's command
  DOM=${DOM:-90,1,2} WGS=256 NEEDLE=${NEEDLE:-conv3x3_pp3_kernelIDF16bLi8ELi32ELi64ELi5ELi0ELi4ELi1ELi2} bash scripts/gpu_r2.sh ${TAG} prof2
  cp gpurun_out/${TAG}_in_graph.json profiles/${TAG}_in_graph.json; cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json
  TB=$(date +%s)
  timeout 1500 python bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - TB )) s"
  cp bench_full.json gpurun_out/${TAG}_bench_default_full.json 2>/dev/null
  python - <<PY
import json
t = open("gpurun_out/${TAG}_bench_default_line.json").read()
print("stdout bytes", len(t), "lines", t.count("\n"))
j = json.loads(t[-10000:].strip().splitlines()[-1])
print({k: v for k, v in j.items() if not isinstance(v, (dict, list))})
print("roofline", j["roofline"]); print("cpu_baseline", j["cpu_baseline"]); print("leg_seconds", j.get("leg_seconds"))
f = json.load(open("gpurun_out/${TAG}_bench_default_full.json"))
print("hires per_kernel_ms", f["hires"]["roofline"]["per_kernel_ms"]); print("512x256 per_kernel_ms", f["roofline"]["per_kernel_ms"])
PY
  tail -3 gpurun_out/${TAG}_bench_default.err | cut -c1-300
  lap final
fi
if has ab2; then    # round-4 main build + round-4 tile cache vs the tree's build + the tree's cache, alternating on this box, both resolutions
  for i in 1 2 3; do
    V2V_LIB_PATH=$R/vid2vid_amd/libv2v_hip_r4main.so V2V_TUNE_CACHE=$R/scripts/ab/tune_cache_r4.json V2V_S7_PATCH=0 timeout 400 python bench.py $LEAN 2>gpurun_out/${TAG}_ab2.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-4 build run $i: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), '| dominant eager', j['roofline'].get('eager_us'), 'live', j['roofline'].get('in_graph_live_us'))"
    timeout 400 python bench.py $LEAN 2>gpurun_out/${TAG}_ab2.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree build    run $i: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), '| dominant eager', j['roofline'].get('eager_us'), 'live', j['roofline'].get('in_graph_live_us'))"
  done 2>&1 | tee gpurun_out/${TAG}_ab2.txt
  lap ab2
fi
if has bntest; then
  timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "bn_apply or batchnorm or splitk_on_a_tiny" -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^E  " | cut -c1-300 | tail -12
  lap bntest
fi
if has ops; then        # per-op dumps of both resolutions -> per-layer roofline tables (every conv alone on the chip)
  timeout 600 python bench.py $LEAN --dump-ops gpurun_out/${TAG}_ops_bf16.json > gpurun_out/${TAG}_benchq.json 2> gpurun_out/${TAG}_benchq.err; echo "benchq rc=$?"
  # (the 512x256 table joins the dump with a record-only census: run  python scripts/per_layer_roofline.py gpurun_out/${TAG}_ops_bf16.json  on a CPU host afterwards)
  python scripts/per_layer_roofline_hires.py gpurun_out/${TAG}_ops_bf16.json.hires.json > gpurun_out/${TAG}_per_layer_roofline_hires.txt 2>&1; tail -3 gpurun_out/${TAG}_per_layer_roofline_hires.txt | cut -c1-600
  lap ops
fi
if has onetest; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short --timeout 300 -k "persistent_single_chunk or paired_x or stride2_persistent or transposed_persistent" -p no:cacheprovider > gpurun_out/${TAG}_onetest.log 2>&1; echo "onetest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_onetest.log | cut -c1-300 | tail -25
  lap onetest
fi
if has frames; then      # both frame rates, twice
  for i in 1 2; do
    timeout 400 python bench.py $LEAN 2>gpurun_out/${TAG}_frames.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), 'frames/s | dominant eager', j['roofline'].get('eager_us'), 'live', j['roofline'].get('in_graph_live_us'))"
  done 2>&1 | tee gpurun_out/${TAG}_frames.txt
  python - <<PY
import json
f = json.load(open("bench_full.json"))
print("hires per_kernel_ms", f["hires"]["roofline"]["per_kernel_ms"])
PY
  lap frames
fi
if has ohtest; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 300 -k "onehot or encode_labels or inference_api_vs_reference or composite_generator or three_scales or uint8_labels" -p no:cacheprovider > gpurun_out/${TAG}_ohtest.log 2>&1; echo "ohtest rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_ohtest.log | cut -c1-300 | tail -25
  lap ohtest
fi
if has ohab; then       # one-hot stems: the previous epilogue (one pass, 37.8 KB of LDS, 32-wide slices only) vs the tree's, alternating on this box
  for i in 1 2 3; do
    V2V_LIB_PATH=$R/vid2vid_amd/libv2v_hip_oldoh.so timeout 120 python scripts/onehot_ab.py 2>&1 | grep -v amdgpu.ids
    timeout 120 python scripts/onehot_ab.py 2>&1 | grep -v amdgpu.ids
  done | tee gpurun_out/${TAG}_ohab.txt
  lap ohab
fi
if has c8; then         # head_epilogue's raw path with 16-byte stores: tiles 60 / 61 (tests + micro-benchmark)
  timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -rf --tb=short --timeout 300 -k "conv7x7 or head or c8 or raw_stats or first_frame or composite_local or three_scales" -p no:cacheprovider > gpurun_out/${TAG}_c8test.log 2>&1; echo "c8test rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_c8test.log | cut -c1-300 | tail -25
  timeout 200 python scripts/c8_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_c8_bench.txt | cut -c1-300
  lap c8
fi
if has t2bench; then
  T2_ONLY=1 timeout 300 python scripts/s2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_t2_bench.txt | cut -c1-300
  lap t2bench
fi
if has onefin; then     # persistent tiles: one statistics row per workgroup (no bn_partial_reduce) + finalize in the launch (V2V_ONE_FIN=0: separate bn_finalize), alternating
  for i in 1 2; do
    for f in 0 1; do
      V2V_ONE_FIN=$f timeout 400 python bench.py $LEAN 2>gpurun_out/${TAG}_onefin.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2V_ONE_FIN=$f run $i: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), 'frames/s')"
    done
  done 2>&1 | tee gpurun_out/${TAG}_onefin.txt
  python - <<PY
import json
f = json.load(open("bench_full.json"))
print("hires per_kernel_ms", f["hires"]["roofline"]["per_kernel_ms"])
PY
  lap onefin
fi
if has stagger; then   # persistent single-chunk kernels: start-up stagger of workgroup groups (V2V_ONE_STAGGER=<units of ~0.5 us>,<groups>) -- do lockstep phases
                       # (every CU loads, then computes, then stores) explain the 70 us of tile 140 against a 32 us HBM bound?
  for st in 0 "4,2" "8,2" "12,2" "16,2" "4,4" "8,4" "3,8" "6,8"; do
    echo "V2V_ONE_STAGGER=$st"; V2V_ONE_STAGGER=$st ONE_TILES=94,140,141,142,143 timeout 120 python scripts/one_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done | tee gpurun_out/${TAG}_stagger.txt
  lap stagger
fi
if has onepmc; then    # where do the persistent single-chunk kernels' wave cycles go?  (one counter group per pass)
  cd /tmp
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    ONE_TILES=94,140,141 timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/onepmc_$tag -o pmc -- python $R/scripts/one_bench.py > $R/gpurun_out/${TAG}_onepmc_$tag.log 2>&1; echo "onepmc $tag rc=$?"
    python $R/scripts/pmc_summary.py $(find /tmp/onepmc_$tag -name "*.db" | head -1) "# ONE_TILES=94,140,141 rocprofv3 --kernel-trace --pmc $pass -- python scripts/one_bench.py (four layer shapes per tile: per-dispatch averages mix them; the 1024x512 layer dominates)" 2>>$R/gpurun_out/${TAG}_onepmc_$tag.log | grep -E "^#|conv3x3_one|ELb1EEEvNS_9ConvKArgsE |Li3ELi0ELi4ELi2ELi1ELb1E" | cut -c1-40,70-200
  done 2>&1 | tee $R/gpurun_out/${TAG}_onepmc.txt
  cd $R
  lap onepmc
fi
