"""Two-rank data-parallel self test on ONE device (tests/test_gpu_plan.py starts it through torch.distributed.run with
gloo and both ranks on cuda:0; never launched by the benchmark driver).

Each rank trains 3 steps on its own small batches with FlatDDP + the whole-network plan, records which buckets were
started between the plan's backward segments and where, and compares its parameters with a single-process run that
accumulates both ranks' batches (iter_size = 2: every loss term halved = the gradient average of the two ranks)."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from gcl_amd import ddp, synthetic
    from gcl_amd.MinkowskiEngine import native
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    rank, world, _ = ddp.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    steps = 3
    keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
    # batches[r][s]: every rank generates both ranks' batches (the reference run needs them)
    batches = [[synthetic.collate_train([synthetic.make_train_sample(500 + 10 * r + s, num_neighborhood=2, n_boxes=10)])
                for s in range(steps)] for r in range(world)]
    rng = np.random.RandomState(0)
    draws = [[None] * steps for _ in range(world)]
    for r in range(world):
        for s in range(steps):
            b = batches[r][s]
            N, G = len(b["sinput_C"]), len(b["group"])
            draws[r][s] = (rng.choice(G, min(G, 64), replace=False), rng.choice(N, 128, replace=False),
                           rng.choice(N, 128, replace=False))

    def on_device(b):
        return {k: (v.to(dev) if isinstance(v, torch.Tensor) and k in keys else v) for k, v in b.items()}

    cfg = dict(batch_size=1, num_pos_per_batch=64, num_hn_samples_per_batch=128, lr=0.05)
    torch.manual_seed(0)
    tr = FinestContrastiveLossTrainer(make_config(**cfg), device=dev, ddp=ddp.FlatDDP())
    init = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    orders, first_cut = [], []
    with torch.cuda.device(dev):
        for s in range(steps):
            b = tr._prefetch_maps(on_device(batches[rank][s]))        # native maps on the side stream, like train_steps
            print(f'rank {rank} step {s} plan {type(tr.model._plan).__name__}', flush=True)
            loss, _, _ = tr.train_step(b, draws[rank][s])
            orders.append(list(getattr(tr.ddp, "last_launch_order", [])))
            plan = tr.model._plan
            if isinstance(plan, native.NetworkPlan):
                plan.bucket_of_param, plan.on_bucket = tr._plan_buckets(), (lambda _b: None)
                segs = plan._segments()
                plan.bucket_of_param = plan.on_bucket = None
                first_cut.append(segs[0][0])
        torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()

    # single-process reference: both ranks' batches of a step accumulated with iter_size = 2
    torch.manual_seed(0)
    ref = FinestContrastiveLossTrainer(make_config(iter_size=2, **cfg), device=dev)
    ref.model.load_state_dict(init)
    with torch.cuda.device(dev):
        for s in range(steps):
            micro = [ref._prefetch_maps(on_device(batches[r][s])) for r in range(world)]
            ref.train_step(micro, [draws[r][s] for r in range(world)])
        torch.cuda.synchronize()
    flat_ref = torch.cat([p.detach().reshape(-1) for p in ref.model.parameters()]).cpu()

    rec = {"rank": rank, "param_sha": hashlib.sha256(flat.numpy().tobytes()).hexdigest(),
           "finite": bool(torch.isfinite(flat).all()) and bool(torch.isfinite(loss).item()),
           "plan_used": isinstance(tr.model._plan, native.NetworkPlan) and isinstance(ref.model._plan, native.NetworkPlan),
           "n_buckets": len(tr.ddp._bounds), "launch_order": orders[-1], "launch_orders": orders,
           "first_bucket_before_record": first_cut[-1] if first_cut else -1,
           "max_abs_diff_vs_averaged_single_process": float((flat - flat_ref).abs().max()),
           "max_abs_param": float(flat_ref.abs().max())}
    out = os.environ.get("GCL_DDP_SELFTEST_DIR", ".")
    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist
    dist.barrier()


if __name__ == "__main__":
    main()
