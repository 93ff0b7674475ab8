"""Pins oracle/loss_oracle.py against golden vectors captured from the reference's own code
(tests/golden/make_golden.py imported lib/metrics.py, lib/eval.py, util/misc.py, lib/colocation_trainer.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as L

G = os.path.join(os.path.dirname(__file__), "golden")


def test_pdist_golden():
    z = np.load(os.path.join(G, "pdist.npz"))
    A, B = torch.from_numpy(z["A"]), torch.from_numpy(z["B"])
    assert np.array_equal(L.pdist(A[:96], B[:96], "L2").numpy(), z["L2_sub"])
    assert np.array_equal(L.pdist(A[:96], B[:96], "SquareL2").numpy(), z["Sq_sub"])
    d, i = L.pdist(A, B, "L2").min(1)
    assert np.array_equal(d.numpy(), z["L2_rowmin"]) and np.array_equal(i.numpy(), z["L2_rowarg"])


@pytest.mark.parametrize("nn_max_n", [500, 2000])
def test_find_nn_golden(nn_max_n):
    z = np.load(os.path.join(G, "find_nn.npz"))
    F0, F1 = torch.from_numpy(z["F0"]), torch.from_numpy(z["F1"])
    idx, dist = L.find_nn(F0, F1, nn_max_n=nn_max_n, return_distance=True)
    assert np.array_equal(idx.numpy(), z[f"idx_{nn_max_n}"])
    assert np.array_equal(dist.numpy()[:, 0], z[f"dist_{nn_max_n}"])
    # chunking never changes the answer (the reference's three settings agree with each other)
    assert np.array_equal(z["idx_-1"], z["idx_500"]) and np.array_equal(z["idx_-1"], z["idx_2000"])


def test_hash_golden():
    z = np.load(os.path.join(G, "hash.npz"))
    M = int(z["M"])
    split = np.split(z["index"], np.cumsum(z["group"])[:-1])
    assert np.array_equal(L.exhaustive_hash(split, M), z["exhaustive"])
    assert np.array_equal(L.neg_hash(z["i1"], z["i2"], M), z["neg"])
    assert np.array_equal(L.positional_hash(z["arr"], 97), z["hash_arr"])
    assert np.array_equal(L.positional_hash([z["arr"][:, 0], z["arr"][:, 1]], 97), z["hash_list"])
    # the symmetric key is collision-free: key == min(i,j)*M + max(i,j)
    lo, hi = np.minimum(z["i1"], z["i2"]), np.maximum(z["i1"], z["i2"])
    assert np.array_equal(z["neg"], lo * M + hi)


def _loss_case(z):
    """(draws, switches) recorded in a finest_loss_*.npz fixture (tests/golden/make_golden.py)."""
    sw = {k: bool(z[k]) for k in ("square_loss", "block_finest_gradient", "use_pair_group_positive_loss",
                                  "finest_term", "use_hard_negative") if k in z.files}
    draws = (z["pos_sel"], z["sel_hn1"], z["sel_hn2"], z["pair_pos"] if "pair_pos" in z.files else None)
    if "random_cols" in z.files:          # use_hard_negative == False: the drawn columns (:514)
        draws = draws + (z["random_cols"],)
    return draws, sw


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "finest_loss_*.npz"))))
def test_finest_contrastive_loss_golden(path):
    z = np.load(path)
    draws, sw = _loss_case(z)
    F = torch.from_numpy(z["F_out"]).requires_grad_(True)
    # (a) replaying the recorded draws
    pos, fin, neg = L.finest_contrastive_loss(
        F, z["group"], z["index"], z["index_hash"], z["finest_flag"],
        max_pos_cluster=int(z["max_pos_cluster"]), max_hn_samples=int(z["max_hn_samples"]),
        draws=draws, **sw)
    assert abs(pos.item() - float(z["pos"])) <= 1e-6 * max(1, abs(float(z["pos"])))
    assert abs(fin.item() - float(z["finest"])) <= 1e-6 * max(1, abs(float(z["finest"])))
    if np.isnan(float(z["neg"])):
        assert np.isnan(neg.item())            # every hardest negative was a self match -> mean of empty
        (pos + fin).backward()
    else:
        assert abs(neg.item() - float(z["neg"])) <= 1e-6
        (pos + fin + neg).backward()
        assert np.allclose(F.grad.numpy(), z["grad"], rtol=1e-5, atol=1e-7)
    # (b) drawing from np.random in the reference's order gives the same selections
    np.random.seed(int(z["np_seed"]))
    p2, f2, n2 = L.finest_contrastive_loss(
        F.detach(), z["group"], z["index"], z["index_hash"], z["finest_flag"],
        max_pos_cluster=int(z["max_pos_cluster"]), max_hn_samples=int(z["max_hn_samples"]), **sw)
    assert abs(p2.item() - float(z["pos"])) <= 1e-6 and abs(f2.item() - float(z["finest"])) <= 1e-6
    if not np.isnan(float(z["neg"])):
        assert abs(n2.item() - float(z["neg"])) <= 1e-6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "circle_loss_*.npz"))))
def test_location_circle_loss_golden(path):
    """oracle/loss_oracle.location_circle_loss vs the reference's own output (all switches), incl. the RNG stream."""
    z = np.load(path)
    sw = {k: bool(z[k]) for k in ("square_loss", "block_finest_gradient", "use_pair_group_positive_loss")}
    F = torch.from_numpy(z["F_out"]).requires_grad_(True)
    draws = (z["pos_sel"], z["pair_pos"] if "pair_pos" in z.files else None)
    args = (z["group"], z["index"], z["finest_flag"], z["points"], z["batch_lengths"])
    pos, fin, neg = L.location_circle_loss(F, *args, max_pos_cluster=int(z["max_pos_cluster"]), draws=draws, **sw)
    for got, key in ((pos, "pos"), (fin, "finest"), (neg, "neg")):
        assert abs(got.item() - float(z[key])) <= 2e-6 * max(1, abs(float(z[key])))
    (pos + fin + neg).backward()
    assert np.allclose(F.grad.numpy(), z["grad"], rtol=1e-4, atol=1e-6)
    np.random.seed(int(z["np_seed"]))
    p2, f2, n2 = L.location_circle_loss(F.detach(), *args, max_pos_cluster=int(z["max_pos_cluster"]), **sw)
    assert abs(p2.item() - float(z["pos"])) <= 2e-6 * max(1, abs(float(z["pos"])))
