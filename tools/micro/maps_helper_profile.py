"""Where the map-prefetch helper's host time goes: _prefetch_maps alone on an idle GPU, cProfile'd (GPU box)."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    batches = bench.make_batches(0, 2, 4, "fixed16")
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
    batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items() if k in keys} for b in batches]
    trainer = FinestContrastiveLossTrainer(make_config(batch_size=4), device=dev)
    from gcl_amd.lib.colocation_trainer import prefetch_to_device
    devb = list(prefetch_to_device(batches, dev))
    torch.cuda.synchronize()
    for b in devb:
        trainer._prefetch_maps(b)
    torch.cuda.synchronize()
    n = 20
    w0, c0 = time.perf_counter(), time.thread_time()
    for i in range(n):
        trainer._prefetch_maps(devb[i % 2])
    w1, c1 = time.perf_counter(), time.thread_time()
    torch.cuda.synchronize()
    print(f"idle GPU: {(w1-w0)/n*1e3:.2f} ms wall, {(c1-c0)/n*1e3:.2f} ms CPU per call")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        trainer._prefetch_maps(devb[i % 2])
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(25)
    st.sort_stats("cumtime").print_stats(30)


main()
