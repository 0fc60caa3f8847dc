#!/bin/bash
# Same-box A/B of the one-launch rebuild of the derived conv operands (GDL_REPACK_FUSION) and of the four-stage 64^2 tile
# (GDL_CONV_STAGE4) at the reference's per-GPU batch 4 (hipGraph replay and eager).   RUN=r06q tools/ab_repack.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; RUN=${RUN:-r06q}; O=$R/gpurun_out/$RUN; mkdir -p $O; cd $R
for rep in 1 2; do
  for cfg in "GDL_REPACK_FUSION=0 GDL_CONV_STAGE4=0" "GDL_REPACK_FUSION=1 GDL_CONV_STAGE4=0" "GDL_REPACK_FUSION=1 GDL_CONV_STAGE4=1"; do
    echo "== $cfg: $(env $cfg python tools/ab_repack_graph.py ${B:-4} 200 2>/dev/null | tail -1)"
  done
done | tee $O/same_box_ab_repack_stage4_graph_b${B:-4}.txt
