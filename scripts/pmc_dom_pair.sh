R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/dom_$tag -o pmc -- python $R/scripts/conv_layer_run.py --cfg 90,1,0 --pair --fused --reps 12 > $R/gpurun_out/r06_v74_pmc_$tag.log 2>&1; echo "pass $tag rc=$?"
  python $R/scripts/pmc_summary.py $(find /tmp/dom_$tag -name "*.db" | head -1) "# fused pair (tile 90, 256 workgroups, final tree: tagged-granule hand-off): rocprofv3 --kernel-trace --pmc $pass -- python scripts/conv_layer_run.py --cfg 90,1,0 --pair --fused --reps 12" 2>>$R/gpurun_out/r06_v74_pmc_$tag.log | grep -E "^#|conv3x3" | cut -c1-20,60-260
done
