#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03v}
mkdir -p $O
cd $R
for e in 0 1; do
  echo "== GDL_CONV_EPILOGUE=$e" >> $O/bisect.txt
  GDL_CONV_EPILOGUE=$e timeout 600 python -m pytest tests/test_hip_unetpp.py -m gpu -x -q -s -k "512_bf16" 2>&1 | grep -v "^$" | tail -6 | cut -c1-300 >> $O/bisect.txt
done
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "bwd_gather" 2>&1 | tail -8 | cut -c1-400 > $O/gather.txt
cat $O/bisect.txt $O/gather.txt
