"""Headline benchmark: active voxels / second through ResUNetBN2C fwd + GCL loss + bwd + SGD step
(BASELINE.json metric; workload = configs[2], the GCL training step at KITTI 0.3 m, bs = 4 x 7 clouds).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL gradient all-reduce over xGMI)

One "step" = one pass of the hot path over one synthetic batch that is ALREADY RESIDENT IN HBM: SparseTensor /
coordinate-map build, kernel maps, 23 sparse convs + 21 BNs forward, finest-contrastive loss, backward, (gradient
all-reduce,) SGD step.  Rank 0 prints ONE JSON line; `roofline` prices the dominant HIP kernel against HBM peak with
per-launch times measured by events on the launch stream inside the timed region; `cpu_baseline` times the CPU oracle
(a torch-CPU restatement, kind "port") on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F32_MATRIX_PEAK_TFS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--group-mode", default="fixed16", choices=["fixed16", "radius"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host and over-subscribes a quota-limited container by an order of magnitude)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline_worker():
    """The CPU oracle (oracle/: gather -> GEMM -> scatter per offset, torch-CPU fp32, all usable host cores) on a
    bounded sample: ONE sample of 1 + 2 clouds (same generator, same model, same loss), forward + loss + backward."""
    from gcl_amd import synthetic
    from oracle import loss_oracle, me_oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    batch = synthetic.collate_train([synthetic.make_train_sample(900, num_neighborhood=2, group_mode="radius")])
    C, F = batch["sinput_C"].numpy(), batch["sinput_F"].float()
    st = me_oracle.random_state(0, dtype=torch.float32)
    n_done, t_total = 0, 0.0
    np.random.seed(0)
    while t_total < 20.0 and n_done < 3:
        leaves = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
        t0 = time.perf_counter()
        mgr = me_oracle.CoordinateManager(C)          # coordinate + kernel maps are part of the step
        out = me_oracle.resunet_forward(leaves, C, F, 5, True, True, 0.05, mgr=mgr)
        pos, fin, neg = loss_oracle.finest_contrastive_loss(out, batch["group"].numpy(), batch["index"].numpy(),
                                                            batch["index_hash"], batch["finest_flag"].numpy(),
                                                            max_pos_cluster=256, max_hn_samples=256)
        (pos + fin + neg).backward()
        t_total += time.perf_counter() - t0
        n_done += 1
    return {"value": round(len(C) * n_done / t_total, 1), "unit": "active voxels/s", "cores": cores, "kind": "port",
            "sample": f"{n_done} step(s) of 1 sample x 3 clouds ({len(C)} voxels), fwd+loss+bwd, torch-CPU fp32 "
                      f"restatement of ME's gather-GEMM-scatter (oracle/me_oracle.py), {t_total:.1f} s"}


def cpu_baseline(timeout_s=240):
    """Runs the worker in a child process (never touches the GPU) under a hard wall-clock limit."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # report, never hang the benchmark line
        return {"value": None, "unit": "active voxels/s", "cores": usable_cores(), "kind": "port",
                "sample": f"CPU baseline did not finish within {timeout_s} s ({type(e).__name__})"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker()), flush=True)
        return
    from gcl_amd import ddp, synthetic
    # test hooks for a 1-GPU box (never set by the driver): run the N>1 code path with every rank on cuda:0 and gloo
    if os.environ.get("GCL_BENCH_SINGLE_DEVICE") == "1":
        rank, world, local = ddp.init_from_env(backend="gloo")
        local = 0
    else:
        rank, world, local = ddp.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    import torch.distributed as dist
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    from gcl_amd.MinkowskiEngine import ops

    # ---- synthetic batch of this rank (weak scaling: every rank gets its own bs=4 batch), moved to HBM --------
    torch.manual_seed(0)
    np.random.seed(rank)
    batch = synthetic.make_train_batch(100 + rank, batch_size=args.batch_size, group_mode=args.group_mode)
    n_vox = len(batch["sinput_C"])
    log(f"rank {rank}: batch ready, {n_vox} voxels, {len(batch['group'])} groups")
    dbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k != "index_hash"}
    cfg = make_config(batch_size=args.batch_size)
    trainer = FinestContrastiveLossTrainer(cfg, device=dev, ddp=ddp.FlatDDP() if world > 1 else None)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import itertools
    for loss, _, _ in trainer.train_steps(itertools.repeat(dbatch, args.warmup)):
        pass
    sync()
    log("warmup done")
    t0 = time.perf_counter()
    # the trainer's epoch loop: all K x 3 np.random draws happen inside the timed region (helper thread, one step
    # ahead).  Per-launch events (roofline) are recorded during the LAST timed step only: recording ~300 events per
    # step costs ~3 ms of host time, which would otherwise perturb every step of the timed region
    steps = trainer.train_steps(itertools.repeat(dbatch, args.steps))
    for it in range(args.steps):
        if it == args.steps - 1 and not args.no_kernel_events:
            ops.PROFILE = []
        loss, parts, _ = next(steps)
    steps.close()
    t_enqueue = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step (host finished enqueuing after "
        f"{t_enqueue / args.steps * 1e3:.2f} ms/step)")
    prof, ops.PROFILE = ops.PROFILE, None
    stats = torch.tensor([dt, float(n_vox)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        vox = stats[1:].clone()
        dist.all_reduce(vox, op=dist.ReduceOp.SUM)
        dt, total_vox = tmax.item(), vox.item()
    else:
        total_vox = float(n_vox)
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the benchmark")

    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    # ---- roofline of the dominant kernel (per-launch event timings from the timed region) -----------------------
    roofline = None
    if prof:
        agg = {}
        log("per-launch timings of the last timed step (events on the launch stream):")
        for name, e0, e1, pairs, cin, cout in prof:
            ms = e0.elapsed_time(e1)
            log(f"  {name:36s} pairs={pairs:9d} {cin:3d}->{cout:3d}  {ms * 1e3:7.1f} us  "
                f"{2.0 * pairs * cin * cout / (ms * 1e-3) / 1e12:6.1f} TFLOP/s  "
                f"{(pairs * (cin + cout) * 4.0 + pairs * 8.0) / (ms * 1e-3) / 1e9:7.0f} GB/s")
            a = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
            a[0] += ms
            a[1] += pairs * (cin + cout) * 4.0 + pairs * 8.0          # algorithmic bytes of a launch
            a[2] += 2.0 * pairs * cin * cout
            a[3] += 1
        dom = max(agg, key=lambda k: agg[k][0])
        ms, by, fl, cnt = agg[dom]
        gbs, tfs = by / (ms * 1e-3) / 1e9, fl / (ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get(dom.replace(" ", ""), {}).get("hbm_bytes_per_launch")
        # which roof bounds the kernel: algorithmic intensity of its launches vs the f32 ridge (157.3 TF / 8 TB/s)
        mfma_bound = fl / by > F32_MATRIX_PEAK_TFS * 1e12 / (HBM_PEAK_GBS * 1e9)
        roofline = {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom,
                    "achieved": round(tfs if mfma_bound else gbs, 2),
                    "peak": F32_MATRIX_PEAK_TFS if mfma_bound else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": round(tfs / F32_MATRIX_PEAK_TFS if mfma_bound else gbs / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "launches": cnt, "avg_launch_us": round(ms * 1e3 / cnt, 2),
                    "algorithmic_GBps": round(gbs, 1), "algorithmic_fp32_TFLOPs": round(tfs, 2),
                    "avg_algorithmic_MB_per_launch": round(by / cnt / 1e6, 3),
                    "note": "algorithmic = exact sparse work of the layer (pairs*(Cin+Cout)*4+8 B, 2*pairs*Cin*Cout flop); "
                            "peak = dense f32 matrix peak (results are fp32-equivalent: fp16x3 / bf16x6 split MFMA) or HBM3E spec",
                    "kernel_ms_last_step": {k: round(v[0], 3) for k, v in agg.items()}}
    out = {
        "metric": "active voxels/sec thru ResUNetBN2C fwd+bwd+GCL loss, KITTI 0.3m",
        "value": round(total_vox * args.steps / dt, 1), "unit": "active voxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if ops.PRECISION != "bf16x3" else "f32 via bf16x3 split (reduced: ~1.5e-5)",
        "arithmetic": ops.PRECISION, "data": "synthetic",
        "config": {"workload": "configs[2]: GCL training step (finest_contrastive_loss), ResUNetBN2C-32 conv1 k=5, "
                               f"KITTI-shaped ray-cast clouds @0.3 m, bs={args.batch_size} x 7 clouds per GPU, "
                               f"positive groups '{args.group_mode}'",
                   "voxels_per_step_rank0": n_vox, "global_batch": args.batch_size * world,
                   "parallelism": f"dp{world}", "loss": float(loss.item())},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
