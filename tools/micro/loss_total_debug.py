import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcl_amd.lib.colocation_trainer import finest_contrastive_loss
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
DEV = "cuda:0"
for path in sorted(glob.glob(os.path.join(G, "finest_loss_*.npz"))):
    z = np.load(path)
    sw = {k: bool(z[k]) for k in ("square_loss", "block_finest_gradient", "use_pair_group_positive_loss", "finest_term", "use_hard_negative") if k in z.files}
    if not sw.get("use_hard_negative", True) or np.isnan(float(z["neg"])):
        continue
    draws = (z["pos_sel"], z["sel_hn1"], z["sel_hn2"], z["pair_pos"] if "pair_pos" in z.files else None)
    kw = dict(max_pos_cluster=int(z["max_pos_cluster"]), max_hn_samples=int(z["max_hn_samples"]), **sw)
    args = lambda F: (F, torch.from_numpy(z["group"]), torch.from_numpy(z["index"]), z["index_hash"], torch.from_numpy(z["finest_flag"]))
    print(os.path.basename(path), sw)
    for w in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (0.7, 1.3, 0.9)):
        F2 = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
        tot = finest_contrastive_loss(*args(F2), draws=draws, total_weights=w, **kw)[0]
        tot.backward()
        F3 = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
        p, f, n = finest_contrastive_loss(*args(F3), draws=draws, **kw)
        (w[0] * p + w[1] * f + w[2] * n).backward()
        d = (F2.grad - F3.grad).norm().item() / max(F3.grad.norm().item(), 1e-30)
        F4 = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
        p, f, n = finest_contrastive_loss(*args(F4), draws=draws, **kw)
        (w[0] * p + w[1] * f + w[2] * n).backward()
        d2 = (F4.grad - F3.grad).norm().item() / max(F3.grad.norm().item(), 1e-30)
        print("   w", w, "fused vs unfused rel", d, " unfused vs unfused rel", d2, " |g|", F3.grad.norm().item())
