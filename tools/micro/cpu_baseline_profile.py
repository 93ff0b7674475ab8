import sys, time, cProfile, pstats
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcl_amd import synthetic
from oracle import loss_oracle, me_cpu, me_oracle
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(NT); me_cpu.lib().me_set_num_threads(NT)
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = synthetic.make_train_batch(100, batch_size=BS, group_mode="fixed16")
C, F = batch["sinput_C"].numpy(), batch["sinput_F"].float()
st = me_oracle.random_state(0, dtype=torch.float32)
def step():
    leaves = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    t0=time.perf_counter()
    mgr = me_cpu.CoordinateManager(C)
    t1=time.perf_counter()
    out = me_oracle.resunet_forward(leaves, C, F, 5, True, True, 0.05, mgr=mgr)
    t2=time.perf_counter()
    pos, fin, neg = loss_oracle.finest_contrastive_loss(out, batch["group"].numpy(), batch["index"].numpy(), batch["index_hash"], batch["finest_flag"].numpy(), max_pos_cluster=256 * BS, max_hn_samples=256 * BS)
    t3=time.perf_counter()
    (pos+fin+neg).backward()
    t4=time.perf_counter()
    print(f"N={len(C)} mgr {t1-t0:.2f} fwd {t2-t1:.2f} loss {t3-t2:.2f} bwd {t4-t3:.2f} total {t4-t0:.2f}")
np.random.seed(0)
step(); step()
if len(sys.argv) > 3:
    pr=cProfile.Profile(); pr.enable(); step(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
else:
    step(); step()
