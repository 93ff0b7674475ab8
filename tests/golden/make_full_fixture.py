"""Generates tests/golden/full_bs4_sample.npz: the fp64 CPU oracle on the FULL-SIZE benchmark batch (BASELINE
configs[2]: seed 100, bs 4 x 7 clouds, 530 321 voxels, groups fixed16) -- 4096 sampled feature rows, per-column sums
of the features, and the loss triple for fixed draws.  Run in the build container (minutes of CPU, ~20 GB):

    python tests/golden/make_full_fixture.py              (forward fixture)
    python tests/golden/make_full_fixture.py --backward   (full_bs4_backward.npz: one oracle training step's gradients)

Inputs are re-created on the GPU box from the same seeds (gcl_amd.synthetic is deterministic numpy; the parameters are
oracle.me_oracle.random_state(0)); only the expected outputs travel in the fixture.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gcl_amd import synthetic                       # noqa: E402
from oracle import loss_oracle, me_oracle           # noqa: E402

SEED, BS, MODE = 100, 4, "fixed16"


def fixed_draws(n_groups, n_rows, max_pos=1024, max_hn=1024):
    rng = np.random.RandomState(12345)
    return (rng.choice(n_groups, min(n_groups, max_pos), replace=False),
            rng.choice(n_rows, min(n_rows, max_hn), replace=False),
            rng.choice(n_rows, min(n_rows, max_hn), replace=False))


def backward_fixture():
    """tests/golden/full_bs4_backward.npz: ONE fp64 oracle training step's gradients on the same batch / parameters /
    draws (reference step: lib/colocation_trainer.py:875-887, loss = pos + finest + neg, weights 1): per-parameter norm,
    256 sampled entries per parameter tensor (all of it when smaller), column sums of dL/dF_out.  ~30 GB, tens of minutes."""
    torch.set_num_threads(8)
    t0 = time.time()
    batch = synthetic.make_train_batch(SEED, batch_size=BS, group_mode=MODE)
    C, F = batch["sinput_C"].numpy(), batch["sinput_F"].double()
    st = me_oracle.random_state(0, dtype=torch.float64)
    leaves = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    out = me_oracle.resunet_forward(leaves, C, F, 5, True, True, 0.05)
    out.retain_grad()
    print(f"oracle forward (autograd) done ({time.time() - t0:.0f} s)", flush=True)
    draws = fixed_draws(len(batch["group"]), len(C))
    pos, fin, neg = loss_oracle.finest_contrastive_loss(out, batch["group"].numpy(), batch["index"].numpy(),
                                                        batch["index_hash"], batch["finest_flag"].numpy(), draws=draws,
                                                        max_pos_cluster=1024, max_hn_samples=1024)
    (pos + fin + neg).backward()
    print(f"oracle backward done ({time.time() - t0:.0f} s)", flush=True)
    rec = {"loss": np.array([pos.item(), fin.item(), neg.item()]), "dF_col_sum": out.grad.sum(0).numpy(),
           "dF_col_abs_sum": out.grad.abs().sum(0).numpy(), "n_voxels": len(C)}
    rng = np.random.RandomState(11)
    names = [k for k, v in leaves.items() if v.requires_grad]
    rec["names"] = np.array(names)
    for j, k in enumerate(names):
        g = leaves[k].grad.reshape(-1).numpy()
        idx = np.sort(rng.choice(g.size, min(g.size, 256), replace=False))
        rec[f"norm_{j}"] = np.float64(np.linalg.norm(g))
        rec[f"idx_{j}"] = idx.astype(np.int64)
        rec[f"val_{j}"] = g[idx].astype(np.float64)
    path = os.path.join(ROOT, "tests", "golden", "full_bs4_backward.npz")
    np.savez_compressed(path, **rec)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB) in {time.time() - t0:.0f} s", flush=True)


def main():
    if "--backward" in sys.argv:
        return backward_fixture()
    torch.set_num_threads(8)
    t0 = time.time()
    batch = synthetic.make_train_batch(SEED, batch_size=BS, group_mode=MODE)
    C, F = batch["sinput_C"].numpy(), batch["sinput_F"].double()
    print(f"batch: {len(C)} voxels, {len(batch['group'])} groups ({time.time() - t0:.0f} s)", flush=True)
    st = me_oracle.random_state(0, dtype=torch.float64)
    with torch.no_grad():
        out = me_oracle.resunet_forward(st, C, F, 5, True, True, 0.05)
    print(f"oracle forward done ({time.time() - t0:.0f} s)", flush=True)
    draws = fixed_draws(len(batch["group"]), len(C))
    pos, fin, neg = loss_oracle.finest_contrastive_loss(out, batch["group"].numpy(), batch["index"].numpy(),
                                                        batch["index_hash"], batch["finest_flag"].numpy(), draws=draws,
                                                        max_pos_cluster=1024, max_hn_samples=1024)
    rows = np.sort(np.random.RandomState(7).choice(len(C), 4096, replace=False))
    path = os.path.join(ROOT, "tests", "golden", "full_bs4_sample.npz")
    np.savez_compressed(path, n_voxels=len(C), n_groups=len(batch["group"]), rows=rows.astype(np.int64),
                        feats=out[rows].numpy().astype(np.float64), col_sum=out.sum(0).numpy(),
                        col_abs_sum=out.abs().sum(0).numpy(), coord_checksum=np.int64(C.astype(np.int64).sum()),
                        loss=np.array([pos.item(), fin.item(), neg.item()]),
                        running_mean_norm1=st["norm1.bn.running_mean"].numpy(),
                        running_var_block4=st["block4.norm2.bn.running_var"].numpy())
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB) in {time.time() - t0:.0f} s; loss {pos.item():.6f} "
          f"{fin.item():.6f} {neg.item():.6f}")


if __name__ == "__main__":
    main()
