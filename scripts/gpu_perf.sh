#!/bin/bash
# GPU-box visit: conv sweep + PMC counter passes over bench.py (separate passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_sweep.py bf16 > gpurun_out/conv_sweep_bf16.txt 2> gpurun_out/conv_sweep.err; echo "sweep rc=$?"
cat gpurun_out/conv_sweep_bf16.txt
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-frames 1 > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err; echo "pmc $tag rc=$?"
  python $R/scripts/pmc_summary.py $(find /tmp/pmc_$tag -name "*.db" | head -1) "# rocprofv3 --kernel-trace --pmc $pass -- python bench.py --steps 4 --warmup 2" > $R/gpurun_out/pmc_$tag.txt 2>> $R/gpurun_out/pmc_$tag.err
done
grep -h "conv_igemm" $R/gpurun_out/pmc_*.txt | cut -c1-60,97- | head -60
