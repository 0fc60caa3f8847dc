#!/bin/bash
# round 6: where the waves of the three-source rolling gather-sum spend their cycles (rocprofv3 --pmc, kernel trace only; two passes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tapsum_stalls
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $OUT/p1 -- python $GRAFT_REPO_ROOT/tools/debug/r06_roll3_parts.py > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM -d $OUT/p2 -- python $GRAFT_REPO_ROOT/tools/debug/r06_roll3_parts.py > $OUT/log2.txt 2>&1
python - <<'P'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_tapsum_stalls"
for p in ("p1", "p2"):
    rows = collections.defaultdict(dict)
    for path in glob.glob(f"{out}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if "roll3" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    # 13 launches per mode (3 warm-ups + 10), modes in the order of the script
    names = ["as is", "no matrix-core work", "no row fetches", "no stores", "no fetches, no compute", "no compute, no stores", "no fetches, no stores", "barriers and tile only", "as is"]
    for m, name in enumerate(names):
        sel = ids[13 * m + 3:13 * m + 13]
        if not sel: continue
        keys = sorted(rows[sel[0]])
        print(f"{name:26s}", {k: f"{sum(rows[i][k] for i in sel) / len(sel):.3e}" for k in keys})
P
