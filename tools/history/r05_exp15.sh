#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
bash tools/r04_ab.sh r05p/ab "GDL_LOWRES_DICE=0" "GDL_LOWRES_DICE=1"
timeout 900 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "dofa or graph or ddp or trainer or tiny or base_512_train" > $O/pytest.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -6
