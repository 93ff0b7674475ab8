"""N>1 path on CPU: world_size-2 gloo run of the flat-buffer gradient all-reduce and the sample sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gcl_amd import ddp
    r, w, _ = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4))
    d = ddp.FlatDDP(overlap=overlap).attach(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.8, weight_decay=1e-4)
    torch.manual_seed(7)
    X, Y = torch.randn(8, 6, 8), torch.randn(8, 6, 4)   # 8 "samples"
    mine = ddp.shard_indices(8, rank, world)
    for step in range(3):
        opt.zero_grad(set_to_none=False)
        loss = sum(((model(X[i]) - Y[i]) ** 2).mean() for i in mine) / len(mine)
        loss.backward()
        d.all_reduce_gradients()
        opt.step()
    out[rank] = (d.flat_param.clone(), d.flat_grad.clone(), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_ddp_two_ranks_gloo():
    mgr = mp.Manager()
    out, out_plain = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)               # bucketed, overlapped
    mp.spawn(_worker, args=(2, _free_port(), out_plain, False), nprocs=2, join=True)  # one collective per step
    (p0, g0, m0), (p1, g1, m1) = out[0], out[1]
    assert torch.equal(p0, out_plain[0][0]) and torch.equal(g0, out_plain[0][1]), "bucketing changes nothing"
    assert sorted(m0 + m1) == list(range(8)) and not set(m0) & set(m1)
    assert torch.equal(p0, p1), "ranks hold identical parameters after averaged-gradient steps"
    assert torch.equal(g0, g1) and g0.abs().sum() > 0
    # single-process reference: same init as rank 0, per-rank BN statistics, averaged gradients
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4))
    import copy
    reps = [ref, copy.deepcopy(ref)]
    opts = [torch.optim.SGD(r.parameters(), lr=0.1, momentum=0.8, weight_decay=1e-4) for r in reps]
    torch.manual_seed(7)
    X, Y = torch.randn(8, 6, 8), torch.randn(8, 6, 4)
    for step in range(3):
        for r, o, mine in zip(reps, opts, (m0, m1)):
            o.zero_grad()
            (sum(((r(X[i]) - Y[i]) ** 2).mean() for i in mine) / len(mine)).backward()
        for pa, pb in zip(reps[0].parameters(), reps[1].parameters()):
            avg = (pa.grad + pb.grad) / 2
            pa.grad.copy_(avg)
            pb.grad.copy_(avg)
        for o in opts:
            o.step()
    flat_ref = torch.cat([p.detach().reshape(-1) for p in reps[0].parameters()])
    assert torch.allclose(p0, flat_ref, rtol=1e-5, atol=1e-6)


def test_size_aware_epoch_plan_balances_ranks():
    """per-rank voxel counts differ by +-30 % under a plain sampler; the greedy per-global-batch assignment
    keeps the heaviest rank within a few percent of the mean, uses every sample once and is identical on all ranks."""
    import numpy as np
    from gcl_amd import ddp
    rng = np.random.RandomState(0)
    sizes = (rng.uniform(0.7, 1.3, 8 * 4 * 25) * 19000).astype(int).tolist()       # 25 global batches of 8 x 4 samples
    plan = ddp.epoch_plan(sizes, world=8, batch_size=4, seed=3)
    plain = ddp.epoch_plan(sizes, world=8, batch_size=4, seed=3, balance=False)
    assert plan == ddp.epoch_plan(sizes, world=8, batch_size=4, seed=3), "deterministic: every rank derives the same plan"
    flat = [i for r in plan for b in r for i in b]
    assert sorted(flat) == list(range(len(sizes))) and all(len(b) == 4 for r in plan for b in r)

    def worst(p):
        out = []
        for step in range(len(p[0])):
            loads = [sum(sizes[i] for i in p[r][step]) for r in range(8)]
            out.append(max(loads) / (sum(loads) / 8))
        return float(np.mean(out))

    assert worst(plan) < 1.03 < worst(plain), (worst(plan), worst(plain))
    import pytest
    with pytest.raises(ValueError):
        ddp.balance_global_batch([1, 2, 3], 2)


def _worker_trainer_shaped(rank, world, port, out):
    """Flat buffer of the real model (8.75 M parameters, registration order of ResUNetBN2C), 4 ranks, a different number
    of rows per rank, gradient accumulation over two micro-steps (buckets start in the last backward pass only)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from gcl_amd import ddp
    from gcl_amd.model import load_model
    ddp.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)
    model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    d = ddp.FlatDDP(overlap=True, n_buckets=2).attach(model)
    params = list(model.parameters())
    assert d.flat_param.numel() == 8753408 and len(d._bounds) == 2
    first = d.flat_param[:1000].clone()
    n_rows = [3, 5, 2, 7][rank]                    # unequal shard sizes: the "loss" is a sum over this rank's rows
    d.flat_grad.zero_()
    for micro in range(2):
        d.set_last_microstep(micro == 1)
        loss = sum((p * float(rank + 1 + micro)).sum() for p in params) * n_rows
        loss.backward()
        if micro == 0:
            assert not d._works, "no collective may start before the last micro-step"
    d.all_reduce_gradients()
    out[rank] = (first, d.flat_grad[:5].clone(), d.flat_grad[-5:].clone(), float(d.flat_grad.min()), float(d.flat_grad.max()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_ddp_four_ranks_trainer_shaped_unequal_shards():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_trainer_shaped, args=(4, _free_port(), out), nprocs=4, join=True)
    # gradient of rank r = sum over micro-steps of (r + 1 + micro) * n_rows[r]; averaged over the 4 ranks
    want = sum(((r + 1) + (r + 2)) * n for r, n in enumerate([3, 5, 2, 7])) / 4.0
    for r in range(4):
        first, g0, g1, gmin, gmax = out[r]
        assert torch.equal(first, out[0][0]), "parameters were broadcast from rank 0"
        assert abs(gmin - want) < 1e-4 and abs(gmax - want) < 1e-4 and torch.allclose(g0, torch.full((5,), want))
