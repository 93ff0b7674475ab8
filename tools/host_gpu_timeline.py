"""DIAGNOSTIC (GPU box): who waits for whom at a step boundary.  Runs a short training loop with GCL_TRACE_HELPERS=2 and
prints, per step, on ONE clock (host perf_counter; GPU events mapped through an anchor event): when the enqueuing thread
started waiting for its helpers, when they were ready, when the step was fully enqueued, when the GPU finished it; and the
map helper's start / end per batch."""
import os, sys
os.environ["GCL_TRACE_HELPERS"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcl_amd import synthetic
from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config, prefetch_to_device

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
host = []
for i in range(4):
    b = synthetic.make_train_batch(100 + i, batch_size=4, group_mode="fixed16")
    host.append({k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items() if k in keys})
trainer = FinestContrastiveLossTrainer(make_config(batch_size=4), device=dev)
for _ in trainer.train_steps(prefetch_to_device((host[i % 4] for i in range(8)), dev, keys)):
    pass                     # warm-up epoch: plan recorded, pools grown
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
for _ in trainer.train_steps(prefetch_to_device((host[i % 4] for i in range(n_steps)), dev, keys)):
    pass
torch.cuda.synchronize()
tl = trainer._timeline
anchor = next(t for t in tl if t[0] == "anchor")
t_a, e_a = anchor[2], anchor[3]
ms = lambda t: (t - t_a) * 1e3
steps = [t for t in tl if t[0] == "step"]
maps = [t for t in tl if t[0] == "maps"]
print("step:  wait from .. helpers ready .. enqueued ..  GPU done   | per step: waited, enqueue took, host lead over the GPU")
prev_gpu = None
for i, (_, _, w0, w1, w2, ev) in enumerate(steps):
    g = e_a.elapsed_time(ev)
    print(f"{i:3d}: {ms(w0):9.2f} {ms(w1):9.2f} {ms(w2):9.2f} {g:9.2f}   | {ms(w1) - ms(w0):6.2f} {ms(w2) - ms(w1):6.2f} {g - ms(w2):6.2f}"
          + (f"   GPU step {g - prev_gpu:6.2f}" if prev_gpu is not None else ""))
    prev_gpu = g
print("map helper calls (start .. end, duration):")
for i, (_, _, t0, t1) in enumerate(maps):
    print(f"{i:3d}: {ms(t0):9.2f} {ms(t1):9.2f}  {ms(t1) - ms(t0):6.2f}")
