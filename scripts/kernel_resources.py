#!/usr/bin/env python3
"""Compile one HIP translation unit with -Rpass-analysis=kernel-resource-usage and print VGPR / SGPR / scratch /
occupancy per kernel (the no-GPU check that a tile configuration does not spill).
    python scripts/kernel_resources.py vid2vid_amd/csrc/conv_igemm_bf16.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + inc,
                    "-c", src] + sys.argv[3:] + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kr.o"], capture_output=True, text=True)
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split("\n")[0].strip()
    if flt not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    m = re.search(r"conv_igemm_kernelI(\w+?)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
    tag = "%s %sx%s w%sx%s ns%s h%s" % m.groups() if m else name[:60]
    print("%-40s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s" % (tag, g("VGPRs"), g("AGPRs"), g("SGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
