"""CPU ORACLE (test infrastructure, NOT product code) -- loss / kNN / hash half of the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

A restatement, in torch-CPU / numpy, of the reference's own Python for this half.  Parity is PINNED:
tests/test_oracle_loss.py checks every function here against tests/golden/*.npz, which were produced by
importing the reference itself (tests/golden/make_golden.py).

* ``pdist``                     lib/metrics.py:22-29
* ``find_nn``                   lib/eval.py:18-48         (chunked brute-force 1-NN, SquareL2)
* ``neg_hash / exhaustive_hash / positional_hash``   util/misc.py:29-55
* ``finest_contrastive_loss``   lib/colocation_trainer.py:430-535 with the script's flags
  (square_loss=True, block_finest_gradient=False, use_pair_group_positive_loss=False,
  use_hard_negative=True; scripts/train_gcl_kitti.sh:96-105)
"""
import numpy as np
import torch


def pdist(A, B, dist_type="L2"):
    """lib/metrics.py:22-29 -- broadcast-subtract pairwise distance (materialises [M, M', C])."""
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    if dist_type == "L2":
        return torch.sqrt(D2 + 1e-7)
    if dist_type == "SquareL2":
        return D2
    raise NotImplementedError("Not implemented")


def find_nn(F0, F1, nn_max_n=-1, return_distance=False, dist_type="SquareL2"):
    """lib/eval.py:18-48 -- 1-NN of every F0 row in F1, F0 processed in chunks of nn_max_n rows."""
    if nn_max_n > 1:
        N = len(F0)
        n_chunks = int(np.ceil(N / nn_max_n))
        dists, inds = [], []
        for i in range(n_chunks):
            d = pdist(F0[i * nn_max_n:(i + 1) * nn_max_n], F1, dist_type)
            md, ind = d.min(dim=1)
            dists.append(md.detach().unsqueeze(1))
            inds.append(ind)
        dists, inds = torch.cat(dists), torch.cat(inds)
        assert len(inds) == N
    else:
        d = pdist(F0, F1, dist_type)
        md, inds = d.min(dim=1)
        dists = md.detach().unsqueeze(1)
    return (inds, dists) if return_distance else inds


def neg_hash(i1, i2, M):
    """util/misc.py:39-40 -- symmetric pair key."""
    i1, i2 = np.asarray(i1, dtype=np.int64), np.asarray(i2, dtype=np.int64)
    return np.minimum(i1 * M + i2, i1 + i2 * M)


def exhaustive_hash(index_split, M):
    """util/misc.py:29-36 -- keys of all in-group pairs, group by group."""
    res = []
    for idx in index_split:
        idx = np.asarray(idx, dtype=np.int64)
        for i in range(len(idx) - 1):
            res.append(np.minimum(idx[i] + idx[i + 1:] * M, idx[i] * M + idx[i + 1:]))
    return np.concatenate(res, axis=0)


def positional_hash(arr, M):
    """util/misc.py:43-55 -- sum_d arr[:, d] * M**d (ndarray [N, D] or list of D arrays)."""
    if isinstance(arr, np.ndarray):
        cols = [arr[:, d] for d in range(arr.shape[1])]
    else:
        cols = list(arr)
    h = np.zeros(len(cols[0]), dtype=np.int64)
    for d, c in enumerate(cols):
        h += np.asarray(c, dtype=np.int64) * M ** d
    return h


def finest_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster=256,
                            max_hn_samples=2048, pos_thresh=0.1, neg_thresh=1.4, finest_thresh=0.2,
                            draws=None, square_loss=True, block_finest_gradient=False,
                            use_pair_group_positive_loss=False, finest_term=True, use_hard_negative=True):
    """lib/colocation_trainer.py:430-535 with its config switches (:466-488, and use_hard_negative == False as the
    reference's code evaluates it: the [M, 1] index tensor of :514-515 broadcasts; draws[4] = the drawn columns); ``finest_term=False`` with ``square_loss=False`` is location_contrastive_loss
    (:734-809).  ``draws`` = (pos_sel, sel_hn1, sel_hn2[, pair_pos]) replays recorded RNG draws; when None they are
    drawn from ``np.random`` in the reference's order (:457, :467, :506-507)."""
    N_out = len(F_out)
    group = [int(g) for g in np.asarray(group)]
    index = torch.as_tensor(np.asarray(index), dtype=torch.long)
    finest_flag = torch.as_tensor(np.asarray(finest_flag), dtype=torch.bool)
    n_groups = len(group)
    index_split = torch.split(index, tuple(group))
    finest_split = torch.split(finest_flag, tuple(group))
    pair_pos = None
    if draws is not None:
        pos_sel, sel_hn1, sel_hn2 = (np.asarray(d) for d in draws[:3])
        pair_pos = draws[3] if len(draws) > 3 else None
    else:
        if n_groups > max_pos_cluster:
            pos_sel = np.random.choice(n_groups, max_pos_cluster, replace=False)
        else:
            pos_sel = np.arange(n_groups)
        sel_hn1 = sel_hn2 = None

    def dist(d2):
        return d2 if square_loss else torch.sqrt(d2 + 1e-7)

    pos_loss, finest_loss = 0, torch.zeros(())
    for s, i in enumerate(pos_sel):
        fs = F_out[index_split[i]]
        fl = finest_split[i]
        mean = torch.mean(fs, dim=0)
        if use_pair_group_positive_loss:                                                              # :466-470
            a, b = (np.random.choice(len(fs), 2, replace=False) if pair_pos is None else pair_pos[s])
            pos_loss = pos_loss + torch.relu(dist((fs[a] - fs[b]).pow(2).sum(-1)) - pos_thresh)
        else:                                                                                         # :472-476
            pos_loss = pos_loss + torch.relu(torch.mean(dist((mean - fs).pow(2).sum(-1))) - pos_thresh)
        if not finest_term:
            continue
        if block_finest_gradient:                                                                     # :478-481
            blocked = fs[torch.bitwise_not(fl)]
            finest_loss = finest_loss + torch.relu(torch.sqrt(
                (torch.mean(blocked, dim=0) - fs[fl][0].detach()).pow(2).sum() + 1e-7) - finest_thresh)
        else:                                                                                         # :483-488
            finest_loss = finest_loss + torch.relu(dist((mean - fs[fl][0]).pow(2).sum()) - finest_thresh)
    pos_loss, finest_loss = pos_loss / len(pos_sel), finest_loss / len(pos_sel)                      # :500
    if sel_hn1 is None:
        sel_hn1 = np.random.choice(N_out, min(N_out, max_hn_samples), replace=False)
        sel_hn2 = np.random.choice(N_out, min(N_out, max_hn_samples), replace=False)
    D = pdist(F_out[sel_hn1], F_out[sel_hn2], "L2")                                                    # :510
    if use_hard_negative:
        Dmin, Dind = D.min(1)                                                                          # :512
        Dind = Dind.numpy()
    else:                                                                                              # :514-515
        cols = draws[4] if (draws is not None and len(draws) > 4) else \
            np.array([np.random.choice(D.shape[1], 1)[0] for _ in range(D.shape[0])])
        Dind = np.asarray(cols, dtype=np.int64).reshape(-1, 1)            # [M, 1]: everything below broadcasts to [M, M]
        Dmin = D[torch.arange(D.shape[0]), torch.from_numpy(Dind)]
    closest = sel_hn2[Dind]
    mask_self = sel_hn1 != closest                                                                     # :521
    neg_keys = neg_hash(sel_hn1, closest, N_out)                                                       # :526
    mask = np.logical_not(np.isin(neg_keys, index_hash, assume_unique=False))                          # :529
    keep = torch.from_numpy(mask & mask_self)
    neg = torch.relu(neg_thresh - Dmin[keep]).pow(2)                                                   # :530
    return pos_loss, finest_loss, neg.mean()


def square_distance(src, dst, normalised=False):
    """util/misc.py:7-26 for [N, C] / [M, C] inputs."""
    dist = -2 * src @ dst.T
    if normalised:
        dist = dist + 2
    else:
        dist = dist + (src ** 2).sum(-1)[:, None] + (dst ** 2).sum(-1)[None, :]
    return torch.clamp(dist, min=1e-12)


def location_circle_loss(F_out, group, index, finest_flag, points, batch_lengths, max_pos_cluster=256, pos_thresh=0.1,
                         neg_thresh=1.4, finest_thresh=0.2, safe_radius=0.75, log_scale=16, draws=None,
                         square_loss=True, block_finest_gradient=True, use_pair_group_positive_loss=False):
    """lib/colocation_trainer.py:538-681.  ``draws`` = (pos_sel (sorted), pair_pos or None)."""
    import torch.nn.functional as Fn
    group = [int(g) for g in np.asarray(group)]
    index = torch.as_tensor(np.asarray(index), dtype=torch.long)
    finest_flag = torch.as_tensor(np.asarray(finest_flag), dtype=torch.bool)
    points = torch.as_tensor(np.asarray(points))
    index_split = torch.split(index, tuple(group))
    finest_split = torch.split(finest_flag, tuple(group))
    n_groups = len(group)
    pair_pos = None
    if draws is not None:
        pos_sel = np.asarray(draws[0])
        pair_pos = draws[1] if len(draws) > 1 else None
    elif n_groups > max_pos_cluster:
        pos_sel = np.sort(np.random.choice(n_groups, max_pos_cluster, replace=False))            # :561-562
    else:
        pos_sel = np.arange(n_groups)
    M = len(pos_sel)
    coords_sel = torch.zeros((M, 3))
    feats_rows = []
    acc = np.cumsum(np.asarray(batch_lengths, dtype=np.float64))                                  # :574-579
    item = np.zeros(M, dtype=np.int64)

    def dist(d2):
        return d2 if square_loss else torch.sqrt(d2 + 1e-7)

    def circle(v):                                                                                # :613-618
        return Fn.softplus(torch.logsumexp(log_scale * v * torch.clamp(v, min=0).detach(), dim=-1)) / log_scale

    pos_loss, finest_loss = 0, 0
    for s, i in enumerate(pos_sel):
        idx, fl = index_split[i], finest_split[i]
        coords_sel[s] = points[idx[0]].float()                                                    # :583
        fs = F_out[idx]
        mean = torch.mean(fs, dim=0)
        feats_rows.append(mean)                                                                   # :585
        item[s] = int(np.sum(int(idx[0]) > acc))                                                  # :589-591
        if use_pair_group_positive_loss:                                                          # :597-605
            a, b = (np.random.choice(len(fs), 2, replace=False) if pair_pos is None else pair_pos[s])
            pos_loss = pos_loss + Fn.softplus(dist((fs[a] - fs[b]).pow(2).sum(-1)) - pos_thresh)
        else:                                                                                     # :607-618
            pos_loss = pos_loss + circle(dist((mean - fs).pow(2).sum(-1)) - pos_thresh / 2)
        if block_finest_gradient:                                                                 # :621-628
            fd = dist((fs[~fl] - fs[fl][0].detach()).pow(2).sum(-1)) - finest_thresh
        else:                                                                                     # :629-635
            fd = dist((fs - fs[fl][0]).pow(2).sum(-1)) - finest_thresh
        finest_loss = finest_loss + circle(fd)
    pos_loss, finest_loss = pos_loss / M, finest_loss / M
    feats_sel = torch.stack(feats_rows)
    batch_mask = torch.from_numpy(item[:, None] == item[None, :])                                 # :645-650 (sorted selection)
    coords_dist = torch.sqrt(square_distance(coords_sel, coords_sel))
    feats_dist = torch.sqrt(square_distance(feats_sel, feats_sel, normalised=True))
    neg_mask = (coords_dist > safe_radius) & batch_mask                                           # :665
    sel = neg_mask.sum(-1) > 0
    neg_weight = feats_dist + 1e5 * (~neg_mask).float()
    neg_weight = torch.clamp(neg_thresh - neg_weight, min=0).detach()
    lse = torch.logsumexp(log_scale * (neg_thresh - feats_dist) * neg_weight, dim=-1)
    return pos_loss, finest_loss, (Fn.softplus(lse) / log_scale)[sel].mean()
