#!/bin/bash
# round-2 diagnostics: (1) block timeline / cycle probe of the short-K GEMMs, (2) HBM-side PMC passes (FETCH_SIZE,
# WRITE_SIZE, L2 hit/miss, MFMA busy) of the dominant 3x3 kernel and of the ViT qkv GEMM, 3 launches each.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_diag
mkdir -p $O
timeout 300 python $R/tools/probe_gemm_timeline.py 32 > $O/timeline.txt 2>&1
for W in fwd qkv; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$i -- python $R/tools/pmc_conv.py $W 32 > $O/pmc_${W}_$i.log 2>&1
  done
done
python $R/tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
# keep the merged output small
find $O -name "*.csv" -size +2M -delete
ls -la $O
