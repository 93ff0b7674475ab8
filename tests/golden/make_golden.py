"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (this container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; writes small .npz fixtures)

The reference never travels to the GPU box: only the arrays written here do (inputs, the exact RNG
draws, outputs, gradients).  What is captured, and from where:

* ``lib.metrics.pdist``                                    lib/metrics.py:22-29
* ``lib.eval.find_nn_gpu``                                 lib/eval.py:18-48
* ``util.misc._hash / _neg_hash / _exhaustive_hash``       util/misc.py:29-55
* ``FinestContrastiveLossTrainer.finest_contrastive_loss`` lib/colocation_trainer.py:430-535
  (instance built with ``__new__`` + the attributes the method reads; np.random seeded, and the
  three ``np.random.choice`` draws re-drawn in the same order to record them); every config switch of the method and
  ``location_contrastive_loss`` (:734-809)
* ``HardestContrastiveLossTrainer.contrastive_hardest_negative_loss``  lib/trainer.py:410-462  (FCGF baseline)
* ``Matcher.SC2_PCR``                                      scripts/SC2_PCR/SC2_PCR.py:304-381 (KITTI config)
* ``FinestContrastiveLossTrainer.location_circle_loss``    lib/colocation_trainer.py:538-681 (all switches)
* ``util.transform_estimation.est_quad_linear_robust``     util/transform_estimation.py:97-126, ``lib.metrics.corr_dist``
  (lib/metrics.py:13-19), ``evaluate_hit_ratio`` (lib/colocation_trainer.py:397-400): the validation step's host half

Third-party modules the reference imports but that are absent here (MinkowskiEngine, open3d,
tensorboardX, easydict) are replaced by EMPTY stub modules -- none of their code is on this path.
"""
import codecs
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    # '# -*- coding: future_fstrings -*-' headers: decode as utf-8
    codecs.register(lambda name: codecs.lookup("utf-8") if name in ("future_fstrings", "future-fstrings") else None)
    for name in ["MinkowskiEngine", "MinkowskiEngine.MinkowskiFunctional", "open3d", "tensorboardX", "easydict"]:
        m = types.ModuleType(name)
        sys.modules[name] = m
    me = sys.modules["MinkowskiEngine"]
    me.MinkowskiFunctional = sys.modules["MinkowskiEngine.MinkowskiFunctional"]
    me.MinkowskiNetwork = torch.nn.Module
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["easydict"].EasyDict = dict
    sys.path.insert(0, REF)
    from lib.metrics import pdist
    from lib.eval import find_nn_gpu
    from util.misc import _hash, _neg_hash, _exhaustive_hash
    from lib.colocation_trainer import FinestContrastiveLossTrainer
    global HardestTrainer
    sys.modules["model"] = types.ModuleType("model")          # lib/trainer.py:17 (the ME-based model package)
    sys.modules["model"].load_model = None
    from lib.trainer import HardestContrastiveLossTrainer as HardestTrainer
    return pdist, find_nn_gpu, _hash, _neg_hash, _exhaustive_hash, FinestContrastiveLossTrainer


def _unit_rows(g, n, c):
    f = torch.randn(n, c, generator=g)
    return f / f.norm(dim=1, keepdim=True)


def make_groups(rng, N, n_groups, sizes):
    """Random positive groups whose members are near-duplicates of a shared anchor (so losses are non-trivial)."""
    group = rng.choice(sizes, n_groups).astype(np.int32)
    index = np.concatenate([rng.choice(N, g, replace=False) for g in group]).astype(np.int64)
    finest = np.zeros(len(index), dtype=bool)
    p = 0
    for g in group:
        finest[p + rng.randint(0, g)] = True
        p += g
    return group, index, finest


def main():
    pdist, find_nn_gpu, _hash, _neg_hash, _exhaustive_hash, Trainer = _import_reference()
    global _TRAINER_AND_HASH
    _TRAINER_AND_HASH = (Trainer, _exhaustive_hash)
    torch.set_num_threads(4)
    only = os.environ.get("GCL_GOLDEN_ONLY")       # e.g. rand_s7: (re)generate ONE finest_loss case, leave the rest alone
    if only:
        return _finest_cases(Trainer, _exhaustive_hash, only)

    # ---- pdist ---------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(0)
    A, B = _unit_rows(g, 1024, 32), _unit_rows(g, 1024, 32)
    sub = slice(0, 96)
    np.savez_compressed(os.path.join(HERE, "pdist.npz"), A=A.numpy(), B=B.numpy(),
                        L2_sub=pdist(A[sub], B[sub], "L2").numpy(),
                        Sq_sub=pdist(A[sub], B[sub], "SquareL2").numpy(),
                        L2_rowmin=pdist(A, B, "L2").min(1)[0].numpy(),
                        L2_rowarg=pdist(A, B, "L2").min(1)[1].numpy())

    # ---- find_nn_gpu ---------------------------------------------------------------------------
    F0, F1 = _unit_rows(g, 5000, 32), _unit_rows(g, 5000, 32)
    F1[:700] = F0[100:800] + 0.01 * torch.randn(700, 32, generator=g)        # some true matches
    out = {"F0": F0.numpy(), "F1": F1.numpy()}
    for nn_max_n in (-1, 500, 2000):
        # (-1 materialises [5000,5000,32] = 3.2 GB; fine here)
        idx, dist = find_nn_gpu(F0, F1, nn_max_n=nn_max_n, return_distance=True)
        out[f"idx_{nn_max_n}"] = idx.numpy()
        out[f"dist_{nn_max_n}"] = dist.numpy()[:, 0]
    np.savez_compressed(os.path.join(HERE, "find_nn.npz"), **out)

    # ---- hashes --------------------------------------------------------------------------------
    rng = np.random.RandomState(0)
    M = 4000
    group, index, _ = make_groups(rng, M, 40, [2, 3, 5, 16, 35])
    split = torch.split(torch.from_numpy(index), tuple(group.tolist()))
    i1, i2 = rng.randint(0, M, 500), rng.randint(0, M, 500)
    arr = rng.randint(0, 50, (64, 3)).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "hash.npz"), M=M, group=group, index=index,
                        exhaustive=_exhaustive_hash(split, M), i1=i1, i2=i2, neg=_neg_hash(i1, i2, M),
                        arr=arr, hash_arr=_hash(arr, 97), hash_list=_hash([arr[:, 0], arr[:, 1]], 97))

    _finest_cases(Trainer, _exhaustive_hash, None)


def _finest_cases(Trainer, _exhaustive_hash, only):
    # ---- finest_contrastive_loss ---------------------------------------------------------------
    cases = {
        "var_s0": dict(seed=0, N=4000, n_groups=200, sizes=[2, 3, 5, 7, 16, 35], max_pos=64, max_hn=512),
        "g16_s1": dict(seed=1, N=6000, n_groups=300, sizes=[16], max_pos=128, max_hn=1024),
        "g16_s2": dict(seed=2, N=3000, n_groups=50, sizes=[16], max_pos=256, max_hn=4096),   # no subsampling
        "var_s3": dict(seed=3, N=5000, n_groups=400, sizes=[2, 4, 9, 16, 30], max_pos=1024, max_hn=1024),
    }
    # the other config switches of the same function (config.py:38-43 defaults differ from the training script's:
    # block_finest_gradient defaults to True) and location_contrastive_loss (finest_weight == 0)
    base = dict(N=3000, n_groups=150, sizes=[2, 3, 5, 8, 16, 35], max_pos=64, max_hn=384)
    cases.update({
        "sqrt_s4": dict(base, seed=4, square_loss=False),
        "block_s5": dict(base, seed=5, block_finest_gradient=True),
        "pair_s6": dict(base, seed=6, use_pair_group_positive_loss=True),
        # use_hard_negative=False: the reference indexes D_fs with an [M, 1] index tensor (:514-515), which broadcasts to
        # an [M, M] average over (row, drawn column) pairs; captured as the code evaluates it
        "rand_s7": dict(base, seed=7, use_hard_negative=False),
        "all_s8": dict(base, seed=8, square_loss=False, block_finest_gradient=True,
                       use_pair_group_positive_loss=True),
        "loc_s9": dict(base, seed=9, fn="location_contrastive_loss", square_loss=False),   # always the sqrt form
    })
    for name, c in cases.items():
        if only and name != only:
            continue
        rng = np.random.RandomState(c["seed"])
        gt = torch.Generator().manual_seed(c["seed"])
        N = c["N"]
        Fo = _unit_rows(gt, N, 32)
        group, index, finest = make_groups(rng, N, c["n_groups"], c["sizes"])
        # make group members similar (anchor + noise) and plant near-duplicate NON-group rows so that the
        # hardest negatives are below the 1.4 threshold and some of them hit the positive-pair mask
        p = 0
        for gsz in group:
            rows = index[p:p + gsz]
            Fo[rows] = Fo[rows[0]] + 0.25 * torch.randn(gsz, 32, generator=gt)
            p += gsz
        Fo = Fo / Fo.norm(dim=1, keepdim=True)
        Fo = Fo.clone().requires_grad_(True)
        split = torch.split(torch.from_numpy(index), tuple(group.tolist()))
        index_hash = _exhaustive_hash(split, N)

        tr = Trainer.__new__(Trainer)
        tr.device = torch.device("cpu")
        tr.pos_thresh, tr.neg_thresh, tr.finest_thresh = 0.1, 1.4, 0.2
        tr.square_loss = c.get("square_loss", True)
        tr.block_finest_gradient = c.get("block_finest_gradient", False)
        tr.use_hard_negative = c.get("use_hard_negative", True)
        tr.use_pair_group_positive_loss = c.get("use_pair_group_positive_loss", False)
        fn = getattr(tr, c.get("fn", "finest_contrastive_loss"))

        np.random.seed(c["seed"] + 100)
        pos, fin, neg = fn(
            Fo, torch.from_numpy(group), torch.from_numpy(index), index_hash, torch.from_numpy(finest),
            max_pos_cluster=c["max_pos"], max_hn_samples=c["max_hn"])
        loss = pos + fin + neg
        loss.backward()
        # re-draw in the same order to record the RNG draws (lib/colocation_trainer.py:457,467,506-507)
        np.random.seed(c["seed"] + 100)
        G = len(group)
        pos_sel = np.random.choice(G, c["max_pos"], replace=False) if G > c["max_pos"] else np.arange(G)
        extra = {}
        if tr.use_pair_group_positive_loss:
            extra["pair_pos"] = np.stack([np.random.choice(int(group[i]), 2, replace=False) for i in pos_sel])
        sel1 = np.random.choice(N, min(N, c["max_hn"]), replace=False)
        sel2 = np.random.choice(N, min(N, c["max_hn"]), replace=False)
        if not tr.use_hard_negative:
            extra["random_cols"] = np.array([np.random.choice(len(sel2), 1)[0] for _ in range(len(sel1))])
        np.savez_compressed(os.path.join(HERE, f"finest_loss_{name}.npz"),
                            F_out=Fo.detach().numpy(), group=group, index=index, finest_flag=finest,
                            index_hash=index_hash, np_seed=c["seed"] + 100,
                            max_pos_cluster=c["max_pos"], max_hn_samples=c["max_hn"],
                            pos_sel=pos_sel, sel_hn1=sel1, sel_hn2=sel2,
                            square_loss=tr.square_loss, block_finest_gradient=tr.block_finest_gradient,
                            use_pair_group_positive_loss=tr.use_pair_group_positive_loss,
                            use_hard_negative=tr.use_hard_negative,
                            finest_term=c.get("fn", "") != "location_contrastive_loss",
                            pos=pos.item(), finest=fin.item(), neg=neg.item(), grad=Fo.grad.numpy(), **extra)
        print(name, pos.item(), fin.item(), neg.item(), float(Fo.grad.abs().sum()))


def hardest_golden():
    """FCGF contrastive_hardest_negative_loss (lib/trainer.py:410-462)."""
    for seed, (N0, N1, P, num_pos, num_hn) in {0: (3000, 2800, 900, 256, 512), 1: (2000, 2500, 300, 1024, 4096)}.items():
        gt = torch.Generator().manual_seed(50 + seed)
        rng = np.random.RandomState(50 + seed)
        F0, F1 = _unit_rows(gt, N0, 32), _unit_rows(gt, N1, 32)
        pairs = np.stack([rng.choice(N0, P, replace=False), rng.choice(N1, P, replace=False)], 1).astype(np.int64)
        F1[pairs[:, 1]] = F0[pairs[:, 0]] + 0.3 * torch.randn(P, 32, generator=gt)
        F1[rng.choice(N1, 200)] = F0[rng.choice(N0, 200)] + 0.05 * torch.randn(200, 32, generator=gt)   # hard negatives
        F1 = F1 / F1.norm(dim=1, keepdim=True)
        F0, F1 = F0.clone().requires_grad_(True), F1.clone().requires_grad_(True)
        tr = HardestTrainer.__new__(HardestTrainer)
        tr.pos_thresh, tr.neg_thresh = 0.1, 1.4
        np.random.seed(seed + 200)
        pos, neg = tr.contrastive_hardest_negative_loss(F0, F1, torch.from_numpy(pairs), num_pos=num_pos,
                                                        num_hn_samples=num_hn)
        (pos + neg).backward()
        np.random.seed(seed + 200)
        sel0 = np.random.choice(N0, min(N0, num_hn), replace=False)
        sel1 = np.random.choice(N1, min(N1, num_hn), replace=False)
        pos_sel = np.random.choice(P, num_pos, replace=False) if P > num_pos else np.zeros(0, np.int64)
        np.savez_compressed(os.path.join(HERE, f"hardest_loss_s{seed}.npz"), F0=F0.detach().numpy(),
                            F1=F1.detach().numpy(), pairs=pairs, num_pos=num_pos, num_hn=num_hn, np_seed=seed + 200,
                            sel0=sel0, sel1=sel1, pos_sel=pos_sel, subsampled=P > num_pos, pos=pos.item(),
                            neg=neg.item(), grad0=F0.grad.numpy(), grad1=F1.grad.numpy())
        print("hardest", seed, pos.item(), neg.item())


def circle_golden(Trainer, _exhaustive_hash):
    """location_circle_loss (lib/colocation_trainer.py:538-681) with its switches."""
    cases = {"c0": dict(seed=20), "c1": dict(seed=21, square_loss=False), "c2": dict(seed=22, block_finest_gradient=False),
             "c3": dict(seed=23, use_pair_group_positive_loss=True)}
    for name, c in cases.items():
        rng = np.random.RandomState(c["seed"])
        gt = torch.Generator().manual_seed(c["seed"])
        N, n_groups, max_pos = 3000, 160, 96
        batch_lengths = [1100, 900, 1000]
        Fo = _unit_rows(gt, N, 32)
        # groups stay inside one sample and are ordered by sample, as the collate function produces them
        group, index, finest = [], [], []
        starts = np.concatenate([[0], np.cumsum(batch_lengths)])
        for s in range(3):
            for _ in range(n_groups // 3 + (1 if s == 0 else 0)):
                g = int(rng.choice([2, 3, 5, 8, 16, 35]))
                rows = starts[s] + rng.choice(batch_lengths[s], g, replace=False)
                fl = np.zeros(g, dtype=bool)
                fl[rng.randint(0, g)] = True
                group.append(g); index.append(rows); finest.append(fl)
        group = np.asarray(group, dtype=np.int32)
        index = np.concatenate(index).astype(np.int64)
        finest = np.concatenate(finest)
        p = 0
        for gsz in group:
            rows = index[p:p + gsz]
            Fo[rows] = Fo[rows[0]] + 0.25 * torch.randn(int(gsz), 32, generator=gt)
            p += gsz
        Fo = (Fo / Fo.norm(dim=1, keepdim=True)).clone().requires_grad_(True)
        points = torch.from_numpy(rng.randint(-40, 40, (N, 3)).astype(np.int32))
        tr = Trainer.__new__(Trainer)
        tr.device = torch.device("cpu")
        tr.pos_thresh, tr.neg_thresh, tr.finest_thresh, tr.safe_radius, tr.log_scale = 0.1, 1.4, 0.2, 0.75, 16
        tr.square_loss = c.get("square_loss", True)
        tr.block_finest_gradient = c.get("block_finest_gradient", True)
        tr.use_pair_group_positive_loss = c.get("use_pair_group_positive_loss", False)
        np.random.seed(c["seed"] + 100)
        pos, fin, neg = tr.location_circle_loss(Fo, torch.from_numpy(group), torch.from_numpy(index), None,
                                                torch.from_numpy(finest), max_pos_cluster=max_pos, points=points,
                                                batch_lengths=batch_lengths)
        (pos + fin + neg).backward()
        np.random.seed(c["seed"] + 100)
        G = len(group)
        pos_sel = np.sort(np.random.choice(G, max_pos, replace=False)) if G > max_pos else np.arange(G)
        extra = {}
        if tr.use_pair_group_positive_loss:
            extra["pair_pos"] = np.stack([np.random.choice(int(group[i]), 2, replace=False) for i in pos_sel])
        np.savez_compressed(os.path.join(HERE, f"circle_loss_{name}.npz"), F_out=Fo.detach().numpy(), group=group,
                            index=index, finest_flag=finest, points=points.numpy(),
                            batch_lengths=np.asarray(batch_lengths), max_pos_cluster=max_pos, np_seed=c["seed"] + 100,
                            pos_sel=pos_sel, square_loss=tr.square_loss, block_finest_gradient=tr.block_finest_gradient,
                            use_pair_group_positive_loss=tr.use_pair_group_positive_loss, pos=pos.item(),
                            finest=fin.item(), neg=neg.item(), grad=Fo.grad.numpy(), **extra)
        print("circle", name, pos.item(), fin.item(), neg.item(), float(Fo.grad.abs().sum()))


def sc2pcr_problem(seed, N, inlier_ratio, noise=0.03):
    """Synthetic putative correspondences: a planar-ish 80 m scene, ground-truth yaw + translation, uniform outliers."""
    rng = np.random.RandomState(seed)
    src = rng.uniform(-40, 40, (N, 3)).astype(np.float32)
    src[:, 2] *= 0.1
    ang = np.deg2rad(20.0 + seed)
    c, s = np.cos(ang), np.sin(ang)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    t = np.array([3.0, 1.0 - seed, 0.5])
    tgt = (src @ R.T + t + rng.normal(0, noise, (N, 3))).astype(np.float32)
    out = rng.rand(N) > inlier_ratio
    tgt[out] = rng.uniform(-40, 40, (int(out.sum()), 3)).astype(np.float32)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return src, tgt, T


def sc2pcr_golden():
    """Matcher.SC2_PCR (scripts/SC2_PCR/SC2_PCR.py:304-381) with scripts/SC2_PCR/config_json/config_KITTI.json."""
    from scripts.SC2_PCR.SC2_PCR import Matcher
    cfg = dict(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
               nms_radius=0.6, max_points=8000, k1=30, k2=20)
    m = Matcher(**cfg)
    for seed, N, ratio in [(0, 1500, 0.35), (1, 2500, 0.5), (2, 800, 0.15), (3, 1200, 0.08)]:
        src, tgt, T = sc2pcr_problem(seed, N, ratio)
        est = m.SC2_PCR(torch.from_numpy(src)[None], torch.from_numpy(tgt)[None])[0].numpy()
        np.savez_compressed(os.path.join(HERE, f"sc2pcr_s{seed}.npz"), src=src, tgt=tgt, T_true=T, T_ref=est,
                            **{k: v for k, v in cfg.items() if k not in ("use_mutual",)})
        print("sc2pcr", seed, N, ratio, float(np.abs(est - T).max()))


def validation_golden(Trainer):
    """The validation step's host arithmetic (lib/colocation_trainer.py:306-400): util.transform_estimation.
    est_quad_linear_robust (util/transform_estimation.py:97-126: 20 re-weighted Gauss-Newton steps on the small-angle
    linearisation), lib.metrics.corr_dist (lib/metrics.py:13-19) and the trainer's evaluate_hit_ratio (:397-400), on
    correspondence sets with outliers (the same generator as the SC2-PCR problems, smaller motions: the estimator is a
    local one)."""
    from util import transform_estimation as te
    from lib.metrics import corr_dist
    tr = Trainer.__new__(Trainer)
    for seed, N, ratio, ang, noise in [(0, 1200, 0.2, 0.05, 0.02), (1, 3000, 0.45, -0.12, 0.05), (2, 400, 0.0, 0.2, 0.0)]:
        rng = np.random.RandomState(100 + seed)
        src = rng.uniform(-20, 20, (N, 3)).astype(np.float32)
        c, s_ = np.cos(ang), np.sin(ang)
        R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(ang / 2), -np.sin(ang / 2)],
                                                                      [0, np.sin(ang / 2), np.cos(ang / 2)]])
        t = np.array([0.8, -0.3 * seed, 0.1])
        tgt = (src @ R.T + t + rng.normal(0, noise, (N, 3))).astype(np.float32)
        out = rng.rand(N) < ratio
        tgt[out] = (src[out] + rng.uniform(-6, 6, (int(out.sum()), 3))).astype(np.float32)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, t
        a, b, Tg = torch.from_numpy(src), torch.from_numpy(tgt), torch.from_numpy(T)
        est = te.est_quad_linear_robust(a, b)
        w = torch.from_numpy(rng.rand(N, 1).astype(np.float32))
        est_w = te.est_quad_linear_robust(a, b, w)
        loss = corr_dist(est, Tg, a, b, weight=None)
        hit = tr.evaluate_hit_ratio(a, b, Tg, thresh=0.3)
        np.savez_compressed(os.path.join(HERE, f"validation_s{seed}.npz"), src=src, tgt=tgt, T_gt=T, T_ref=est.numpy(),
                            weight=w.numpy(), T_ref_weighted=est_w.numpy(), corr_dist=float(loss), hit_ratio=float(hit),
                            hit_thresh=0.3)
        print("validation", seed, N, ratio, float((est - Tg).abs().max()), float(loss), hit)


if __name__ == "__main__":
    main()
    if os.environ.get("GCL_GOLDEN_ONLY"):
        sys.exit(0)
    hardest_golden()
    sc2pcr_golden()
    circle_golden(*_TRAINER_AND_HASH)
    validation_golden(_TRAINER_AND_HASH[0])
