cd $GRAFT_REPO_ROOT
LEAN="--no-cpu-baseline --no-train-line --no-train-hires --no-c1 --no-c4 --no-train-c3"
for rep in 1 2; do
for v in 0 1; do
V2V_RAW_BF16=$v timeout 900 python bench.py $LEAN 2>gpurun_out/rawall_$v.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2V_RAW_BF16=$v run $rep: 512x256', j['value'], 'frames/s | 2048x1024', j.get('hires_value'), 'frames/s')"
python - <<PY
import json
j = json.load(open("bench_full.json"))
h = j.get("hires", {})
pk = h.get("roofline", {}).get("per_kernel_ms", {})
print("   hires per-kernel ms:", {k: v for k, v in list(pk.items())[:5]})
PY
done
done
