#!/usr/bin/env python3
"""Static check of a gfx950 code object for the hazard behind the "stamp build miscompute" (DESIGN 4).

gfx950 has no interlock between an MFMA and a later NON-MFMA instruction that reads (or overwrites) the MFMA's result registers:
software must keep `passes + 2` (fp32-input MFMA) / `passes + 3` (bf16 / f16 / f8 ... MFMA) wait states between the two
(CDNA3 ISA 4.5, "manually inserted wait states"; LLVM: GCNHazardRecognizer::checkMAIVALUHazards).  The compiler inserts the
`s_nop`s, but its backwards search over the control-flow graph marks a block as visited on the FIRST path that reaches it
(GCNHazardRecognizer::getWaitStatesSince): when the block that ends with the last MFMA of a K loop is first reached the long way
round (through the loop header), the direct edge loop-exit -> epilogue is never priced, and an `v_accvgpr_read` of the last
accumulator register can be issued 7 wait states behind a 16-pass MFMA that needs 18.  That is what the V2V_STAMP_MASK build of
conv_igemm_kernel<float,64,64,2,2,2,false> does (register 15 of every lane is read before the last v_mfma_f32_32x32x2_f32 has
written it: pixels (odd row, column 11 / 15) of the 8 x 16 layer wrong in every channel).

This script walks every path FORWARD from every MFMA of a disassembled code object and reports each non-MFMA access to the
result registers that comes too early.

    llvm-objdump --offloading libv2v_hip.so ; llvm-objdump -d <bundle> > x.dis ; python scripts/mfma_hazard_check.py x.dis
    python scripts/mfma_hazard_check.py --lib vid2vid_amd/libv2v_hip.so          (does the extraction itself, into a temp dir)
exit code 1 when a violation was found.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
INS = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):((?:\s[0-9A-Fa-f]{8})+)")
FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))")


def passes_of(mn):
    """-> (passes, wait states a non-MFMA access of the result has to keep)"""
    f32_in = mn.endswith("_f32") and ("x1_" in mn or "x2_" in mn or "x4_" in mn) and "xf32" not in mn and "f8" not in mn
    if "_f64_" in mn:
        return 16, 19
    if "32x32" in mn:
        p = 16 if (f32_in or mn.endswith("x1_2b_f32") or "x4_" in mn) else 8
    elif "16x16" in mn:
        p = 8 if f32_in else 4
    else:
        p = 4 if not f32_in else 2
    if "f8f6f4" in mn or "smfmac" in mn:
        p = max(p, 8 if "32x32" in mn else 4) * (2 if "f8f6f4" in mn else 1)
    return p, p + (2 if f32_in else 3)


def regs_of(ops):
    out = []
    for m in REG.finditer(ops):
        if m.group(4) is not None:
            lo = hi = int(m.group(4))
        else:
            lo, hi = int(m.group(2)), int(m.group(3))
        out.append((m.group(1), lo, hi))
    return out


QUIET = False


def check(path, verbose=False):
    funcs, cur = [], None
    for line in open(path, errors="replace"):
        m = FUNC.match(line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        m = INS.match(line)
        if m and cur is not None:
            addr = int(m.group(3), 16)
            size = 4 * len(m.group(4).split())
            cur[1].append((addr, size, m.group(1), m.group(2)))
    n_mfma = n_viol = 0
    worst = {}
    for name, ins in funcs:
        at = {a: i for i, (a, _, _, _) in enumerate(ins)}
        for i, (addr, size, mn, ops) in enumerate(ins):
            if not (mn.startswith("v_mfma") or mn.startswith("v_smfmac")):
                continue
            n_mfma += 1
            dst = regs_of(ops.split(",")[0])
            if not dst:
                continue
            bank, lo, hi = dst[0]
            p, need = passes_of(mn)
            best = {}
            stack = [(i + 1, 0)]
            while stack:
                j, ws = stack.pop()
                while j < len(ins) and ws < need:
                    if best.get(j, 1 << 30) <= ws:
                        break
                    best[j] = ws
                    a2, s2, mn2, ops2 = ins[j]
                    touched = [r for r in regs_of(ops2) if r[0] == bank and r[1] <= hi and r[2] >= lo]
                    if mn2.startswith("v_mfma") or mn2.startswith("v_smfmac"):
                        if touched:
                            break                       # the next link of the accumulation chain: the hardware's own dependency rules
                    elif touched:
                        n_viol += 1
                        key = (name, mn, mn2)
                        if key not in worst or ws < worst[key][0]:
                            worst[key] = (ws, need, addr, a2, ops2)
                        break
                    if mn2 == "s_endpgm" or mn2.startswith("s_setpc") or mn2.startswith("s_swappc"):
                        break
                    step = 1
                    if mn2.startswith("v_mfma") or mn2.startswith("v_smfmac"):
                        step = passes_of(mn2)[0]        # an independent MFMA holds the (in-order) matrix pipe for its passes
                    if mn2 == "s_nop":
                        step = int(ops2.split()[0], 0) + 1
                    if mn2 == "s_branch" or mn2.startswith("s_cbranch"):
                        off = int(ops2.split()[0], 0)
                        off = off - 65536 if off >= 32768 else off
                        tgt = a2 + 4 + 4 * off
                        if tgt in at:
                            stack.append((at[tgt], ws + step))
                        if mn2 == "s_branch":
                            break
                    ws += step
                    j += 1
    if not QUIET or n_viol:
        print("%s: %d functions, %d MFMAs, %d early accesses" % (os.path.basename(path), len(funcs), n_mfma, n_viol))
    for (name, mn, mn2), (ws, need, a1, a2, ops2) in sorted(worst.items(), key=lambda kv: kv[1][0]):
        print("  VIOLATION %d of %d wait states: %s @%x -> %s %s @%x in %s" % (ws, need, mn, a1, mn2, ops2[:50], a2, name[:150]))
    return n_viol


def main():
    global QUIET
    args = [a for a in sys.argv[1:] if a != "--quiet"]
    QUIET = "--quiet" in sys.argv[1:]
    total = 0
    if args and args[0] == "--lib":
        lib = os.path.abspath(args[1])
        with tempfile.TemporaryDirectory() as td:
            cp = os.path.join(td, os.path.basename(lib))
            os.symlink(lib, cp)
            subprocess.run([OBJDUMP, "--offloading", cp], cwd=td, check=True, capture_output=True)
            for b in sorted(glob.glob(cp + ".*gfx950")):
                dis = b + ".dis"
                with open(dis, "w") as f:
                    subprocess.run([OBJDUMP, "-d", b], stdout=f, check=True)
                total += check(dis)
                os.remove(dis)
    else:
        for p in args:
            total += check(p)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
