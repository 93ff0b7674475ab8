"""DIAGNOSTIC: loss / parameter norms over many steps of the benchmark configuration (does the run stay finite?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gcl_amd import synthetic
from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config, prefetch_to_device
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
torch.manual_seed(0); np.random.seed(0)
nb = int(os.environ.get("NB", "2"))
batches = [synthetic.make_train_batch(100 + 1000 * j, batch_size=4, group_mode="fixed16") for j in range(nb)]
keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
host = [{k: v.pin_memory() for k, v in b.items() if k in keys} for b in batches]
tr = FinestContrastiveLossTrainer(make_config(), device=dev)
n = int(os.environ.get("STEPS", "90"))
for i, (loss, parts, _) in enumerate(tr.train_steps(prefetch_to_device([host[j % nb] for j in range(n)], dev, keys))):
    if i % 5 == 0 or not torch.isfinite(loss):
        pmax = max(float(p.detach().abs().max()) for p in tr.model.parameters())
        gmax = max(float(p.grad.abs().max()) for p in tr.model.parameters() if p.grad is not None)
        print(i, float(loss), [float(x) for x in parts], "max|p|", pmax, "max|g|", gmax, flush=True)
    if not torch.isfinite(loss):
        for name, p in tr.model.named_parameters():
            if not torch.isfinite(p).all():
                print("non-finite parameter:", name)
        break
