#!/usr/bin/env python3
"""Static check of the gfx950 ISA of every kernel in csrc/: scalar (kernel-argument) loads that the compiler re-issues inside unrolled
code.  A uniform operand used in an unrolled epilogue is often NOT kept in an SGPR but re-fetched per element (`s_load_dword` +
`s_waitcnt lgkmcnt(0)`, ~200 cycles each) -- invisible in the source, 5-10 us per launch in the conv epilogues.
Per kernel: instruction lines, scalar loads, and the most-repeated (base, offset) pair.  No GPU needed.
    python scripts/isa_sload_report.py [file.hip ...] > profiles/rNN_isa_sload_report.txt"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vid2vid_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
print("# kernels with a scalar load repeated >= 8 times (same base register pair and offset): candidates for pinning the operand in an SGPR")
print("%-110s %7s %7s %7s  %s" % ("kernel", "lines", "s_load", "repeat", "most repeated operand"))
for f in files:
    asm = os.path.join("/tmp", "isa_" + os.path.basename(f) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", "-o", asm, os.path.join(CSRC, f)], cwd=CSRC, stderr=subprocess.DEVNULL, check=True)
    cur, stats = None, {}
    for ln in open(asm):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            stats[cur] = [0, collections.Counter()]
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if ln.startswith("\t") and not ln.startswith("\t."):
            stats[cur][0] += 1
        m = re.search(r"s_load_dword\w*\s+\S+,\s*(s\[\d+:\d+\]),\s*(\S+)", ln)
        if m:
            stats[cur][1][(m.group(1), m.group(2))] += 1
    for k, (n, c) in sorted(stats.items(), key=lambda kv: -max(kv[1][1].values(), default=0)):
        if not c or max(c.values()) < 8:
            continue
        (base, off), rep = c.most_common(1)[0]
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print("%-110s %7d %7d %7d  %s + %s" % (name[:110], n, sum(c.values()), rep, base, off))
    os.remove(asm)
