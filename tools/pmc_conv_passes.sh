cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in 2 3; do
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  GDL_VARIANT=$V timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_v${V}_$i -- python $R/tools/pmc_conv.py fwd > /dev/null 2>&1
done
done
ls $R/gpurun_out/ | grep pmc_v
