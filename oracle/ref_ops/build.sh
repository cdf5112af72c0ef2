#!/bin/bash
# Builds oracle/_ref/libref_ops.so: the reference's OWN FlowNet2 native-op kernels, executed on host cores.
#
#   * sources stay where they lie under /root/reference (read-only); the kernel BODIES are cut out of the .cu files by line
#     range at build time into oracle/_ref/gen_*.inc (a build output: oracle/_ref/ is git-ignored, nothing of the reference
#     is committed), each range guarded by an anchor check so that a changed reference fails the build instead of
#     silently compiling something else;
#   * oracle/ref_ops/cuda_emu.h supplies the CUDA execution model on the CPU (fibers per thread, 32-lane shuffles);
#   * oracle/ref_ops/ref_*.cpp restate only the host-side launch geometry (cited there).
# TEST INFRASTRUCTURE: loaded by oracle/ref_ops.py for tests/ only.  The .so travels to the GPU box with the snapshot;
# /root/reference does not exist there, so this script is a no-op when the reference is absent and a library exists.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
REF=${V2V_REFERENCE:-/root/reference}/models/flownet2_pytorch/networks
mkdir -p "$OUT"
if [ ! -d "$REF" ]; then
  if [ -f "$OUT/libref_ops.so" ]; then echo "ref_ops: reference tree absent, keeping the prebuilt $OUT/libref_ops.so"; exit 0; fi
  echo "ref_ops: $REF not found and no prebuilt library" >&2; exit 1
fi
anchor() {   # anchor <file> <line> <expected substring>
  sed -n "${2}p" "$1" | grep -qF -- "$3" || { echo "ref_ops: $1:$2 does not contain '$3' (reference changed?)" >&2; exit 1; }
}
cut_lines() { sed -n "${2},${3}p" "$1"; }

F=$REF/correlation_package/correlation_cuda_kernel.cu
anchor $F 6 "#define THREADS_PER_BLOCK 32"; anchor $F 17 "warpReduceSum"; anchor $F 47 "channels_first"
anchor $F 74 "correlation_forward"; anchor $F 147 "}"; anchor $F 151 "correlation_backward_input1"
{ cut_lines $F 5 7; cut_lines $F 16 147; } > "$OUT/gen_correlation.inc"

F=$REF/resample2d_package/resample2d_kernel.cu
anchor $F 5 "#define CUDA_NUM_THREADS 512"; anchor $F 16 "kernel_resample2d_update_output"; anchor $F 68 "kernel_resample2d_backward_input1"
anchor $F 120 "kernel_resample2d_backward_input2"; anchor $F 190 "}"; anchor $F 192 "void resample2d_kernel_forward"
{ cut_lines $F 5 13; cut_lines $F 15 190; } > "$OUT/gen_resample2d.inc"

F=$REF/channelnorm_package/channelnorm_kernel.cu
anchor $F 7 "#define CUDA_NUM_THREADS 512"; anchor $F 19 "kernel_channelnorm_update_output"; anchor $F 64 "kernel_channelnorm_backward_input1"
anchor $F 96 "}"; anchor $F 98 "void channelnorm_kernel_forward"
{ cut_lines $F 7 14; cut_lines $F 18 96; } > "$OUT/gen_channelnorm.inc"

# -ffp-contract=off: no fused multiply-add on any host, so the library computes the same bits wherever it was built
CXX=${CXX:-g++}
$CXX -O2 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-variable -Wno-unused-but-set-variable \
    -I"$HERE" -I"$OUT" "$HERE/ref_correlation.cpp" "$HERE/ref_resample2d.cpp" "$HERE/ref_channelnorm.cpp" -o "$OUT/libref_ops.so"
echo "ref_ops: built $OUT/libref_ops.so from $REF"
