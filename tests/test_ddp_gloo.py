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
