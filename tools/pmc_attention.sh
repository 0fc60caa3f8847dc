#!/bin/bash
# PMC counters of the fused attention forward kernels (DOFA-base shape): separate passes, --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_attention}; mkdir -p $O
for ver in 3; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/v${ver}_p$i -- python $R/tools/run_attention_once.py $ver > $O/v${ver}_p$i.log 2>&1
  done
  echo "== flash forward, kernel version $ver (3 = round 3: 32-query waves, deferred maximum; 4 = experiment: reference on the matrix pipe, no per-tile maximum)"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$O/v${ver}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "flash_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"  {c:28s} {sum(v)/len(v):18.1f}  n={len(v)}")
PY
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" -size +2M -delete
