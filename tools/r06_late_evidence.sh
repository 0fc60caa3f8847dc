#!/bin/bash
# round 6, late items: the evidence files of DESIGN.md 4.2 (rolling gather-sum, XCD-major weight gradients, batch-4 tile threshold) from ONE box
O=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $O
python tools/bench_tapsum_roll.py 64 > $O/bench_tapsum_rolling_window_neck_x4_x2_b64.txt 2>&1
python tools/bench_tapsum_roll.py 32 >> $O/bench_tapsum_rolling_window_neck_x4_x2_b64.txt 2>&1
python tools/bench_tapsum_roll3.py 64 > $O/bench_tapsum_rolling_window_fpn_bottleneck_b64.txt 2>&1
python tools/debug/r06_roll3_parts.py >> $O/bench_tapsum_rolling_window_fpn_bottleneck_b64.txt 2>&1
python tools/bench_small_m.py 4 > $O/bench_small_m_batch4_64_vs_128_tiles.txt 2>&1
tools/pmc_tapsum_lds.sh > $O/pmc_tapsum_lds_bank_conflicts.txt 2>&1
tools/ab_env.sh GDL_TAPSUM_ROLL 0 1 2 > $O/same_box_ab_rolling_gather_sum.txt 2>&1
tools/ab_env.sh GDL_WGRAD_XCD_GROUP 8 1 2 > $O/same_box_ab_wgrad_xcd_major.txt 2>&1
grep -v "amdgpu.ids" $O/*.txt | tail -60
