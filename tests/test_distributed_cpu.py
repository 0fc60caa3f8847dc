"""world_size-2 gloo tests (CPU) for the N>1 host logic: SyncBatchNorm statistic exchange,
BN-backward sum exchange, and bench.py's max-over-ranks timing reduction."""

import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdlhip import nn as gnn
        g = torch.Generator().manual_seed(5)
        # RAGGED split (a short last batch on one rank): rank 0 holds 6 pixels, rank 1 holds 10
        sizes = [6, 10][:world]
        full = torch.randn(sum(sizes), 8, generator=g) * 2 + 0.7     # [pixels over all ranks, C]
        lo = sum(sizes[:rank])
        mine = full[lo:lo + sizes[rank]]
        mean, var = mine.mean(0), mine.var(0, unbiased=False)
        gm, gv, total = gnn.sync_batch_stats(mean, var, count=mine.shape[0])
        ok = torch.allclose(gm, full.mean(0), atol=1e-6) and torch.allclose(gv, full.var(0, unbiased=False), atol=1e-5)
        ok = ok and isinstance(total, torch.Tensor) and float(total) == float(full.shape[0])
        rm, rv = torch.zeros(8), torch.ones(8)
        gnn.update_running_stats(rm, rv, gm, gv, 0.1, total)         # count as a tensor: no host read-back
        ref = torch.nn.BatchNorm1d(8)
        ref.train()
        ref(full)
        ok = ok and torch.allclose(rm, ref.running_mean, atol=1e-6) and torch.allclose(rv, ref.running_var, atol=1e-5)
        rm2, rv2 = torch.zeros(8), torch.ones(8)
        gnn.update_running_stats(rm2, rv2, gm, gv, 0.1, full.shape[0])
        ok = ok and torch.allclose(rm2, rm) and torch.allclose(rv2, rv)
        # backward: the dx kernel divides the exchanged sums by the LOCAL count; scaled by p_local / P_global that is
        # the global mean of dy (what SyncBatchNorm's backward uses)
        share = mine.shape[0] / total
        s1, _ = gnn.sync_sum_pair(mine.sum(0), (mine * mine).sum(0))
        ok = ok and torch.allclose(s1 * share / mine.shape[0], full.mean(0), atol=1e-6)
        a, b = gnn.sync_sum_pair(mine.sum(0), (mine * mine).sum(0))
        ok = ok and torch.allclose(a, full.sum(0), atol=1e-5) and torch.allclose(b, (full * full).sum(0), atol=1e-4)
        # bench.py timing rule: the job time is the MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_syncbn_exchange_world2():
    world = 2
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _agree_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdlhip.trainer import MiniTrainer
        tr = MiniTrainer(max_epochs=1, accelerator="cpu", enable_checkpointing=False, logger=False)
        tr._ddp_active = True
        out = [tr._agree("t", "ok"),                                   # unanimous
               tr._agree("t", "no" if rank == 1 else "ok"),            # one rank's veto reaches everybody
               tr._agree("t", "fatal" if rank == 0 else "ok")]         # ... and so does a dead capture
        if rank == 0:                                                   # rank 1 never answers the fourth question: a timeout is "fatal"
            out.append(tr._agree("alone", "ok", timeout_s=2.0))
        # the protocol used no collective: the process group is still in step
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        out.append(t.item())
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_capture_verdicts_are_agreed_through_the_store_not_a_collective():
    """MiniTrainer._agree (advisor, round 5): the ranks settle "do we capture the DDP step" / "did the capture work" through the
    process group's key-value store, so a rank that failed before the warm-up collectives, or whose capture died, cannot pair a
    flag all-reduce with a peer's bucket all-reduce; a rank that never answers counts as a dead capture after the timeout."""
    world, port = 2, 29500 + (os.getpid() + 7) % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] == ["ok", "no", "fatal", "fatal", 1.0]
    assert ret[1] == ["ok", "no", "fatal", 1.0]


def test_bench_gpus2_starts_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` started as ONE process (the way the driver starts --gpus 1) re-executes itself under
    torch.distributed.run with two ranks, and rank 0 prints exactly one JSON line with ddp.ranks == 2 (the launch path on CPU:
    GDL_BENCH_DRY_RUN=1 swaps the HIP step for a toy step and RCCL for gloo -- reference: configs/dofa_config_RGB.yaml:3-13)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, GDL_BENCH_DRY_RUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ddp"]["ranks"] == 2 and line["steps"] == 3 and line["dry_run"] is True
    assert "starting the ranks myself" in proc.stderr
