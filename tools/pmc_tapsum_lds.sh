#!/bin/bash
# round 6: LDS bank conflicts of the forward gather-sum kernels (rocprofv3 --pmc, kernel trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tapsum_lds
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/p1 -- python $GRAFT_REPO_ROOT/tools/bench_tapsum_roll3.py 64 > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/p2 -- python $GRAFT_REPO_ROOT/tools/bench_tapsum_roll.py 64 > $OUT/log2.txt 2>&1
python - <<'P'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_tapsum_lds"
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{out}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if "fwd_sum" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        print(k)
        print("   launches", max(len(v) for v in cs.values()), {c: f"{v:.3e}" for c, v in m.items()},
              "conflict share of LDS cycles %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
P
