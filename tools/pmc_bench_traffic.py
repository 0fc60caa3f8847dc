#!/usr/bin/env python
"""Aggregate the rocprofv3 --pmc passes of tools/pmc_bench_traffic.sh (one bench.py training step at the bench default per-GPU batch, 64 since round 6):
per kernel, launches and mean per-launch FETCH_SIZE / WRITE_SIZE / L2 hit rate / MFMA-busy share, and the JSON that
bench.py reads for `roofline.traffic` (profiles/pmc_dominant_kernel_traffic.json).

Corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 64 B
per 128-B request of a wide coalesced read (global_load_dwordx4 and buffer_load ... lds alike) -> doubled.  WRITE_SIZE is
taken as reported (uncalibrated per the guide)."""
import collections
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))


from gemm_source_hash import gemm_source_hash  # noqa: E402

root = sys.argv[1]
# steps the traced command ran (tools/pmc_bench_traffic.sh: --warmup 1 --steps 1 = 2 training steps)
STEPS_TRACED = int(sys.argv[2]) if len(sys.argv) > 2 else 2
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{root}/pass_*/*/*counter_collection.csv"):
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    fetch = 2.0 * mean.get("FETCH_SIZE", 0.0) * 1024
    write = mean.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = mean.get("TCC_HIT_sum", 0.0), mean.get("TCC_MISS_sum", 0.0)
    busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mean.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((n * (fetch + write), k, n, fetch, write, hit / max(hit + miss, 1.0), busy / max(gui / 8 * 1024, 1.0)))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s} {'L2 hit':>7s} {'MFMA busy':>9s}")
for _, k, n, fetch, write, hr, mf in rows[:40]:
    print(f"{k[:70]:70s} {n:8d} {fetch / 1e6:16.1f} {write / 1e6:16.1f} {hr:7.3f} {mf:9.3f}")
# every implicit-GEMM kernel class of the step, keyed by the names bench.py's KernelTimer uses: the bench line takes the
# entry of whichever class dominates its step
NAMES = (("conv3x3_sf_kernel<bf16_tag>", "conv3x3_sf_kernel<bf16>"),
         ("conv_gemm_dual_kernel", "conv_gemm_dual_kernel"),
         ("conv_gemm_w4_kernel", "conv_gemm_w4_kernel"),
         ("conv_gemm_persist_kernel", "conv_gemm_persist_kernel"),
         ("conv_gemm_w4p_kernel", "conv_gemm_w4p_kernel"),
         ("conv_gemm_kernel<bf16_tag, 2, 4, 4, 2, false, true", "conv_gemm_kernel<bf16,2,4,4,2,pingpong>"),
         ("conv_gemm_kernel<bf16_tag, 2, 2, 2, 2", "conv_gemm_kernel<bf16,2,2,2,2>"),
         ("conv_gemm_kernel<bf16_tag, 2, 2, 1, 1", "conv_gemm_kernel<bf16,2,2,1,1>"),
         ("conv_gemm_kernel<bf16_tag, 4, 1, 2, 2", "conv_gemm_kernel<bf16,4,1,2,2>"),
         ("conv3x3_narrow_kernel", "conv3x3_narrow_kernel"))
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over one bench.py training step at "
        "per-GPU batch of tools/pmc_bench_traffic.sh (64 since round 6; 32 before), mean over the step's launches of this kernel class; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B "
        "request), KiB -> bytes; tools/pmc_bench_traffic.sh")
entries = []
for prof_name, bench_name in NAMES:
    sel = [r for r in rows if prof_name in r[1]]
    if not sel:
        continue
    n = sum(r[2] for r in sel)
    fetch = sum(r[3] * r[2] for r in sel) / n
    write = sum(r[4] * r[2] for r in sel) / n
    hr = sum(r[5] * r[2] for r in sel) / n
    mf = sum(r[6] * r[2] for r in sel) / n
    entries.append({"kernel_substring": bench_name, "launches_per_train_step": n // STEPS_TRACED, "launches_traced": n, "hbm_bytes_per_launch": round(fetch + write),
                    "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write), "l2_hit_rate": round(hr, 4),
                    "mfma_busy_share": round(mf, 4)})
with open(f"{root}/pmc_dominant_kernel_traffic.json", "w") as f:
    json.dump({"note": NOTE, "gemm_source_hash": gemm_source_hash(), "kernels": entries}, f, indent=1)
print(json.dumps({"kernels": entries}))
