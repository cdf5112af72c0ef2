"""Host stubs and device code of every translation unit come from the SAME sources.

hipcc compiles a .hip file twice (device pass, host pass) and reads the headers in each pass.  Round 5: a header was edited while
`make` was running; one object ended up with the new kernel names in its host stubs and the OLD kernels in its embedded gfx950
code object, `make` considered it up to date (it was newer than the header), and every launch of the new kernel died on the GPU box
with "Cannot find Symbol".  This test unbundles the gfx950 code object of every in-tree object file and checks that each
`__device_stub__<kernel>` the host side registers has its `<kernel>` function in the device code."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJS = sorted(glob.glob(os.path.join(ROOT, "vid2vid_amd", "csrc", "*.o")))


@pytest.mark.skipif(not OBJS or not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")),
                    reason="needs the in-tree object files and the ROCm LLVM tools")
@pytest.mark.parametrize("obj", OBJS, ids=[os.path.basename(o) for o in OBJS])
def test_host_stubs_have_their_device_kernels(obj, tmp_path):
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], check=True, capture_output=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co, "--unbundle"], check=True, capture_output=True)
    nm = shutil.which("nm") or os.path.join(LLVM, "llvm-nm")
    host = subprocess.run([nm, obj], check=True, capture_output=True, text=True).stdout
    # _ZN3v2v<len>__device_stub__<name>...  ->  _ZN3v2v<len - 15><name>...   (15 = len("__device_stub__"))
    stubs = set()
    for m in re.finditer(r"\b(_ZN3v2v)(\d+)__device_stub__(\S+)", host):
        n = int(m.group(2)) - 15
        stubs.add("%s%d%s" % (m.group(1), n, m.group(3)))
    for m in re.finditer(r"\b_Z(\d+)__device_stub__(\S+)", host):            # kernels outside the namespace
        stubs.add("_Z%d%s" % (int(m.group(1)) - 15, m.group(2)))
    dev = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], check=True, capture_output=True, text=True).stdout
    funcs = {l.split()[-1] for l in dev.splitlines() if " FUNC " in l}
    assert stubs, "no kernels found in %s" % obj
    missing = sorted(s for s in stubs if s not in funcs)
    assert not missing, "%s: host stubs without device code (stale object -- rebuild it): %s" % (os.path.basename(obj), missing[:5])
