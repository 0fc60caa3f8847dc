#!/bin/bash
# final code: the multi-GPU launch path as far as one GPU allows -- torchrun with one rank at batch 32 (eager DDP + SyncBN over a one-rank
# RCCL group: what the driver's scaling run launches per rank), and --force-ddp at the reference's per-GPU batch 4 (eager vs captured)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05w3; mkdir -p $O; cd $R
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/torchrun1.err | tail -1 > $O/torchrun1.json
python -c "
import json; d=json.loads(open('$O/torchrun1.json').read()); print('torchrun 1 rank b32: train', d['value'], 'inference', d.get('inference_tiles_per_s'), 'ddp', json.dumps(d.get('ddp'))[:400])" | tee $O/summary.txt
timeout 300 python bench.py --force-ddp --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/bench_ddp_b4.err | tail -1 > $O/bench_ddp_b4.json
python -c "
import json; d=json.loads(open('$O/bench_ddp_b4.json').read()); dd=d.get('ddp') or {}; print('force-ddp b4: eager', d['value'], 'tiles/s', d['ms_per_step'], 'ms; graphed', dd.get('graphed'))" | tee -a $O/summary.txt
timeout 200 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>/dev/null | tail -1 > $O/bench_b4.json
python -c "
import json; d=json.loads(open('$O/bench_b4.json').read()); print('single process b4 (eager line):', d['value'], 'by_batch', d.get('by_batch'))" | tee -a $O/summary.txt
