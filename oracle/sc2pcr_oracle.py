"""CPU restatement of SC2-PCR as the reference ships it (TEST INFRASTRUCTURE: imported by tests/, never by the product).

Follows scripts/SC2_PCR/SC2_PCR.py (Matcher.SC2_PCR :304-381, pick_seeds :32-58, cal_seed_trans :60-165,
cal_leading_eigenvector :167-193, post_refinement :238-279) and scripts/SC2_PCR/common.py:7-45 (rigid_transform_3d),
batch size 1, float32 like the reference.  Where the reference leaves an order unspecified (torch.argsort /
torch.argmax among equal values) this restatement -- and the HIP build -- take the LOWEST index first; the power
iterations of the per-seed 20x20 matrices always run ``num_iterations`` steps (the reference stops all seeds together
once every vector passes torch.allclose, a change below 1e-5 relative).  Pinned against outputs of the reference
itself: tests/golden/sc2pcr_*.npz (tests/golden/make_golden.py), final transformation within 2e-3.
"""
import numpy as np
import torch


def _pdist3(p):
    return torch.norm(p[:, None, :] - p[None, :, :], dim=-1)


def _argsort_desc(v, dim=-1):
    """Descending, ties -> lowest index (stable sort of the negated values)."""
    return torch.sort(-v, dim=dim, stable=True)[1]


def leading_eigenvector(M, num_iterations, early_stop=True):
    """:167-185, M [b, n, n] -> [b, n]."""
    x = torch.ones_like(M[:, :, 0:1])
    last = x
    for _ in range(num_iterations):
        x = torch.bmm(M, x)
        x = x / (torch.norm(x, dim=1, keepdim=True) + 1e-6)
        if early_stop and torch.allclose(x, last):
            break
        last = x
    return x.squeeze(-1)


def rigid_transform_3d(A, B, w):
    """common.py:7-45 for [b, n, 3] clouds and weights [b, n]; returns [b, 4, 4]."""
    sw = w.sum(1, keepdim=True)[:, :, None] + 1e-6
    ca = (A * w[:, :, None]).sum(1, keepdim=True) / sw
    cb = (B * w[:, :, None]).sum(1, keepdim=True) / sw
    Am, Bm = A - ca, B - cb
    H = Am.permute(0, 2, 1) @ (w[:, :, None] * Bm)
    U, S, V = torch.svd(H.double())
    d = torch.det(V @ U.permute(0, 2, 1))
    D = torch.eye(3, dtype=torch.float64)[None].repeat(A.shape[0], 1, 1)
    D[:, 2, 2] = d
    R = (V @ D @ U.permute(0, 2, 1)).float()
    t = cb.permute(0, 2, 1) - R @ ca.permute(0, 2, 1)
    T = torch.eye(4)[None].repeat(A.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3:4] = t
    return T


def sc2_pcr(src, tgt, inlier_threshold=0.6, d_thre=0.1, num_iterations=20, ratio=0.2, nms_radius=0.6,
            max_points=8000, k1=30, k2=20, return_stages=False):
    """src, tgt float32 [N, 3] -> final 4x4 transformation (float32 tensor)."""
    src, tgt = torch.as_tensor(src, dtype=torch.float32), torch.as_tensor(tgt, dtype=torch.float32)
    if src.shape[0] > max_points:                                              # :320-323
        src, tgt = src[:max_points], tgt[:max_points]
    N = src.shape[0]
    sd, td = _pdist3(src), _pdist3(tgt)                                        # :329-331
    cross = torch.abs(sd - td)
    SC = torch.clamp(1.0 - cross ** 2 / d_thre ** 2, min=0)                    # :337
    hard = (cross < d_thre).float()
    conf = leading_eigenvector(SC[None], num_iterations)[0]                    # :345
    # pick_seeds :32-58
    rel = (conf[:, None] >= conf[None, :]) | (sd >= nms_radius)
    is_max = rel.min(-1)[0].float()
    n_seeds = int(N * ratio)
    seeds = _argsort_desc(conf * is_max)[:n_seeds]
    # second order measure :353-361
    tight = (cross < d_thre / 2).float()
    SC2 = (tight[seeds] @ tight) * hard[seeds]                                 # [S, N]
    # cal_seed_trans :60-165
    kk1, kk2 = (k1, k2) if k1 <= N else (4, 4)
    knn = _argsort_desc(SC2, dim=1)[:, :kk1]                                   # [S, k1]
    s_knn, t_knn = src[knn], tgt[knn]                                          # [S, k1, 3]
    cd = torch.abs(torch.norm(s_knn[:, :, None] - s_knn[:, None], dim=-1) -
                   torch.norm(t_knn[:, :, None] - t_knn[:, None], dim=-1))
    lh = (cd < d_thre).float()
    lsc2 = torch.matmul(lh[:, :1, :], lh)[:, 0]                                # [S, k1]
    fine = _argsort_desc(lsc2, dim=1)[:, :kk2]                                 # [S, k2]
    s_f = torch.gather(s_knn, 1, fine[:, :, None].expand(-1, -1, 3))
    t_f = torch.gather(t_knn, 1, fine[:, :, None].expand(-1, -1, 3))
    cd = torch.abs(torch.norm(s_f[:, :, None] - s_f[:, None], dim=-1) -
                   torch.norm(t_f[:, :, None] - t_f[:, None], dim=-1))
    lsc = torch.clamp(1 - cd ** 2 / d_thre ** 2, min=0)                        # :121-123
    ar = torch.arange(kk2)
    lsc[:, ar, ar] = 0
    wgt = leading_eigenvector(lsc, num_iterations, early_stop=False)
    wgt = wgt / (wgt.sum(-1, keepdim=True) + 1e-6)
    Ts = rigid_transform_3d(s_f, t_f, wgt)                                     # [S, 4, 4]
    pred = torch.einsum("snm,jm->sjn", Ts[:, :3, :3], src) + Ts[:, None, :3, 3]   # [S, N, 3]
    fitness = (torch.norm(pred - tgt[None], dim=-1) < inlier_threshold).float().sum(-1)
    best = int(_argsort_desc(fitness)[0])
    T = Ts[best:best + 1]
    T0 = T.clone()
    # post_refinement :238-279
    thr = 0.10 if inlier_threshold == 0.10 else 1.2
    prev = 0
    for _ in range(20):
        warped = src @ T[0, :3, :3].T + T[0, :3, 3]
        d = torch.norm(warped - tgt, dim=-1)
        inl = d < thr
        cnt = int(inl.sum())
        if abs(cnt - prev) < 1:
            break
        prev = cnt
        T = rigid_transform_3d(src[None, inl], tgt[None, inl], (1 / (1 + (d / thr) ** 2))[None, inl])
    if return_stages:
        return T[0], dict(conf=conf, seeds=seeds, SC2=SC2, knn=knn, fine=fine, seed_trans=Ts, fitness=fitness,
                          best=best, initial=T0[0])
    return T[0]
