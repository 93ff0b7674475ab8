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

batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = torch.device("cuda:0")
dbatch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k != "index_hash"}
tr = FinestContrastiveLossTrainer(make_config(), device=dev)
for _ in range(3):
    tr.train_step(dbatch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
import itertools
for _ in tr.train_steps(itertools.repeat(dbatch, 10)):
    pass
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
