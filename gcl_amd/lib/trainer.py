"""FCGF baseline loss on the MI355X kernels (interface of lib/trainer.py:408-462).

``contrastive_hardest_negative_loss``: positive pairs pulled together, and for each positive the hardest negative among
a random subset of the other cloud pushed apart.  The two [num_pos, num_hn] distance matrices are never formed: the
row minimum / arg-minimum come from ``gcl_nn_rowmin`` (one launch each); the selected distances are then re-evaluated
with differentiable torch ops on [num_pos, C] slices, and the positional-hash membership test
(``np.isin(_hash(...), pos_keys)``, :447-456) runs on the device.
"""
import numpy as np
import torch
import torch.nn.functional as F_

from gcl_amd.lib.metrics import pdist_min


def draw_hardest_selections(N0, N1, n_pos_pairs, num_pos, num_hn_samples):
    """The reference's three np.random draws in its order (lib/trainer.py:424-431)."""
    sel0 = np.random.choice(N0, min(N0, num_hn_samples), replace=False)
    sel1 = np.random.choice(N1, min(N1, num_hn_samples), replace=False)
    pos_sel = np.random.choice(n_pos_pairs, num_pos, replace=False) if n_pos_pairs > num_pos else None
    return sel0, sel1, pos_sel


def contrastive_hardest_negative_loss(F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048, pos_thresh=0.1,
                                      neg_thresh=1.4, draws=None):
    """Returns (pos_loss, neg_loss) like HardestContrastiveLossTrainer.contrastive_hardest_negative_loss
    (``self.pos_thresh`` / ``self.neg_thresh`` become arguments).  F0 [N0, C], F1 [N1, C] on the GPU; positive_pairs
    int [P, 2].  ``draws = (sel0, sel1, pos_sel or None)`` replays recorded selections."""
    dev = F0.device
    N0, N1 = len(F0), len(F1)
    pairs = torch.as_tensor(np.asarray(positive_pairs) if not torch.is_tensor(positive_pairs) else positive_pairs)
    pairs = pairs.to(torch.int64)
    if draws is None:
        draws = draw_hardest_selections(N0, N1, len(pairs), num_pos, num_hn_samples)
    sel0, sel1, pos_sel = draws
    hash_seed = max(N0, N1)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev, non_blocking=True)
    sel0, sel1 = to_dev(sel0), to_dev(sel1)
    pairs = pairs.to(dev, non_blocking=True)
    sample = pairs if pos_sel is None else pairs[to_dev(pos_sel)]
    ind0, ind1 = sample[:, 0].contiguous(), sample[:, 1].contiguous()
    # hardest negative of every positive among the sampled rows of the other cloud (:441-445)
    _, a01 = pdist_min(F0, F1, "L2", rows_a=ind0, rows_b=sel1)
    _, a10 = pdist_min(F1, F0, "L2", rows_a=ind1, rows_b=sel0)
    n01, n10 = sel1[a01.long()], sel0[a10.long()]
    posF0, posF1 = F0[ind0], F1[ind1]
    D01min = torch.sqrt((posF0 - F1[n01]).pow(2).sum(1) + 1e-7)
    D10min = torch.sqrt((posF1 - F0[n10]).pow(2).sum(1) + 1e-7)
    # positional hash i0 + i1 * hash_seed (util/misc.py:43-55) and the membership test (:447-456)
    pos_keys = pairs[:, 0] + pairs[:, 1] * hash_seed
    mask0 = torch.logical_not(torch.isin(ind0 + n01 * hash_seed, pos_keys))
    mask1 = torch.logical_not(torch.isin(n10 + ind1 * hash_seed, pos_keys))
    pos_loss = F_.relu((posF0 - posF1).pow(2).sum(1) - pos_thresh)
    neg_loss0 = F_.relu(neg_thresh - D01min[mask0]).pow(2)
    neg_loss1 = F_.relu(neg_thresh - D10min[mask1]).pow(2)
    return pos_loss.mean(), (neg_loss0.mean() + neg_loss1.mean()) / 2
