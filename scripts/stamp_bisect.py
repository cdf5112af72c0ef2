#!/usr/bin/env python3
"""which tile selection makes the V2V_STAMP_MASK build fail a golden test?

Round 5, visit 2 established: the stamp build's failures are NOT a run-time race -- a run's tile selections (V2V_TUNE_CACHE) replayed
on the stamp build reproduce its verdict every time, and the same selections pass on the product build.  The selections differ from
run to run because the per-shape search is timing based.  This script takes a failing and a passing selection file and
delta-debugs the difference down to the smallest set of (layer shape -> tile, split-K) entries that still fails on the stamp build.

    python scripts/stamp_bisect.py <failing_cache.json> <passing_cache.json> [<lib.so>]      (on the GPU box)
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fail_p, pass_p = sys.argv[1], sys.argv[2]
LIB = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "vid2vid_amd", "libv2v_hip_stamp.so")
F, P = json.load(open(fail_p)), json.load(open(pass_p))
SEL = "inference_api_vs_reference or flownet2_vs_reference"
runs = 0


def run(cache, select=SEL, lib=LIB):
    """-> (failed test ids, raw tail)"""
    global runs
    runs += 1
    fd, path = tempfile.mkstemp(suffix=".json")
    os.close(fd)
    json.dump(cache, open(path, "w"))
    env = dict(os.environ, V2V_TUNE_CACHE=path)
    if lib:
        env["V2V_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_golden.py", "-m", "gpu", "-q", "-rf", "--tb=line", "-k", select,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    os.remove(path)
    failed = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("FAILED ")]
    msgs = [l for l in r.stdout.splitlines() if "relative error" in l][:2]
    return failed, msgs


def hybrid(keys):
    h = json.loads(json.dumps(P))
    for dt, k in keys:
        h.setdefault(dt, {})[k] = F[dt][k]
    return h


failed, msgs = run(F)
print("failing selection on %s: %s %s" % (os.path.basename(LIB), failed, msgs))
if not failed:
    sys.exit("the 'failing' selection passes here")
pf, _ = run(P)
print("passing selection: %s" % (pf or "passes"))
test = failed[0].split("::")[-1]
sel = test.split("[")[0] if "[" not in test else test.replace("[", " and ").replace("]", "")
diff = [(dt, k) for dt in F for k in F[dt] if P.get(dt, {}).get(k) != F[dt][k]]
print("%d entries differ; delta-debugging on test %s" % (len(diff), test))
fails = lambda keys: bool(run(hybrid(keys), select=test.split("[")[0])[0])
S, n = diff, 2
while len(S) > 1 and runs < 60:
    chunk = max(1, len(S) // n)
    parts = [S[i:i + chunk] for i in range(0, len(S), chunk)]
    for part in parts:
        if fails(part):
            S, n = part, 2
            break
    else:
        for part in parts:
            rest = [k for k in S if k not in part]
            if rest and fails(rest):
                S, n = rest, max(n - 1, 2)
                break
        else:
            if n >= len(S):
                break
            n = min(len(S), 2 * n)
print("minimal failing set after %d runs (%d entries):" % (runs, len(S)))
for dt, k in S:
    print("   dtype %s  shape key %s :  failing selection %s   passing selection %s" % (dt, k, F[dt][k], P.get(dt, {}).get(k)))
h = hybrid(S)
print("replay on the stamp build:", run(h, select=test.split("[")[0]))
print("replay on the PRODUCT build:", run(h, select=test.split("[")[0], lib=""))
