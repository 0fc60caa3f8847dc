#!/bin/bash
# wide (C = 512 / 768 / 1024) and Dropout2d-scaled classifier-head kernels: parity, SegFormer-B2 same-box A/B, SegFormer / DOFA task tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z2; mkdir -p $O; cd $R
timeout 150 python -m pytest tests/test_hip_ops.py -q -x -k "head_mfma or head_and_logit" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6
for v in 0 1; do
  GDL_HEAD_MFMA=$v timeout 100 python bench.py --model segformer --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/segformer_$v.json
  python -c "
import json; d=json.loads(open('$O/segformer_$v.json').read()); print('GDL_HEAD_MFMA=$v SegFormer-B2: train %.1f tiles/s (%.2f ms), inference %.1f (%.2f ms)' % (d['value'], d['ms_per_step'], d['inference_tiles_per_s'], d['inference_ms_per_step']))"
done | tee $O/segformer_ab.txt
timeout 120 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "segformer" > $O/pytest_seg.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_seg.txt | tail -4
