cd $GRAFT_REPO_ROOT
P='import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "value", j["value"], "train", j.get("train_value"), "hires", j.get("hires_value"), "c1", j.get("c1_value"), "c4", j.get("c4_value"), "leg_s", j.get("leg_seconds"))'
C="--no-cpu-baseline --no-train-hires --no-train-c3 --no-train-parity"
timeout 900 python bench.py $C --no-c1 --no-c4 2>/dev/null | python -c "$P" hires_only
timeout 900 python bench.py $C --no-hires --no-c4 2>/dev/null | python -c "$P" c1_only
timeout 900 python bench.py $C --no-hires --no-c1 2>/dev/null | python -c "$P" c4_only
