#!/bin/bash
# sanity of the multi-GPU launch path as far as one GPU allows: torchrun with one rank, and --force-ddp at batch 32 (eager DDP over a
# one-rank RCCL group with SyncBatchNorm conversion), plus the tightened production-shape tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-input-stage 2>$O/torchrun.err | tail -1 > $O/torchrun.json
python -c "
import json; d=json.loads(open('$O/torchrun.json').read()); print('torchrun 1 rank:', d['value'], d.get('inference_tiles_per_s'), d['n_gpus'])" | tee -a $O/summary.txt
timeout 600 python bench.py --force-ddp --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/force_ddp.err | tail -1 > $O/force_ddp.json
python -c "
import json; d=json.loads(open('$O/force_ddp.json').read()); print('force-ddp b32:', d['value'], d.get('ddp'))" | tee -a $O/summary.txt
cp $R/gpurun_out/bench_details.json $O/force_ddp_details.json
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "plain_conv_modules" 2>&1 | tail -2 | tee -a $O/summary.txt
