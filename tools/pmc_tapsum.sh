#!/bin/bash
# PMC counters of the gather-sum kernels (neck x4 level, batch 32): separate passes, --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_tapsum}; mkdir -p $O
for mode in ${2:-2 1}; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/m${mode}_p$i -- python $R/tools/run_tapsum_once.py $mode > $O/m${mode}_p$i.log 2>&1
  done
  echo "== mode $mode"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$O/m${mode}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "fwd_sum" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"  {c:28s} {sum(v)/len(v):18.1f}  n={len(v)}")
PY
done
find $O -name "*.csv" -size +2M -delete
