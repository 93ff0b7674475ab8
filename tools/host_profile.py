"""DIAGNOSTIC: cProfile of the host side of the training step (the step is host-bound once the GPU work is < ~28 ms)."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gcl_amd import synthetic
from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config

from gcl_amd.lib.colocation_trainer import prefetch_to_device
keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
batches = [synthetic.make_train_batch(100 + 1000 * j, batch_size=4, group_mode="fixed16") for j in range(2)]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
host = [{k: v.pin_memory() for k, v in b.items() if k in keys} for b in batches]
tr = FinestContrastiveLossTrainer(make_config(), device=dev)
for _ in tr.train_steps(prefetch_to_device([host[i % 2] for i in range(4)], dev, keys)):
    pass
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in tr.train_steps(prefetch_to_device([host[i % 2] for i in range(10)], dev, keys)):
    pass
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
