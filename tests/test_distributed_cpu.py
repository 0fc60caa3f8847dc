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
