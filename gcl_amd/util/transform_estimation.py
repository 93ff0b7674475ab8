"""Robust rigid alignment of matched points for the validation step (interface of util/transform_estimation.py:97-126,
called by lib/colocation_trainer.py:341).

Host arithmetic on a few thousand correspondences (the reference runs it on CPU tensors too): 20 re-weighted
Gauss-Newton steps on the small-angle linearisation  w (omega x p + t) = w (q - p),  each solved through its 6 x 6
normal equations, the step applied as Rz Ry Rx + t, weights par / (|p - q| + par) with par halved every five steps.
Written from the algorithm, not the reference's code: the normal matrix is accumulated from per-point 3 x 6 blocks in
float64 (the reference stacks a [3N, 6] float32 system and inverts A^T A); results agree to float32 rounding
(tests/golden/validation_*.npz, captured from the reference)."""
import torch


def _euler_zyx(a):
    """R = Rz(a[2]) Ry(a[1]) Rx(a[0])  (util/transform_estimation.py:5-45)."""
    cx, sx, cy, sy, cz, sz = torch.cos(a[0]), torch.sin(a[0]), torch.cos(a[1]), torch.sin(a[1]), torch.cos(a[2]), torch.sin(a[2])
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx]).reshape(3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy]).reshape(3, 3)
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one]).reshape(3, 3)
    return rz @ ry @ rx


def get_trans(x):
    """4 x 4 transform of the 6-vector (omega, t)."""
    x = x.reshape(-1)
    T = torch.eye(4, dtype=x.dtype)
    T[:3, :3] = _euler_zyx(x[:3])
    T[:3, 3] = x[3:]
    return T


def update_pcd(pts, trans):
    return pts @ trans[:3, :3].t() + trans[:3, 3]


def compute_weights(pts0, pts1, par):
    return par / (torch.norm(pts0 - pts1, dim=1, keepdim=True) + par)


def _gauss_newton_step(p, q, w):
    """Solution of the weighted linearised system: rows w_i [-[p_i]x | I] x = w_i (q_i - p_i)."""
    n = p.shape[0]
    A = torch.zeros((n, 3, 6), dtype=p.dtype)
    A[:, 0, 1], A[:, 0, 2] = p[:, 2], -p[:, 1]
    A[:, 1, 0], A[:, 1, 2] = -p[:, 2], p[:, 0]
    A[:, 2, 0], A[:, 2, 1] = p[:, 1], -p[:, 0]
    A[:, 0, 3] = A[:, 1, 4] = A[:, 2, 5] = 1.0
    w2 = (w.reshape(-1) ** 2)[:, None, None]
    M = torch.einsum("nij,nik->jk", A * w2, A)
    g = torch.einsum("nij,ni->j", A * w2, q - p)
    return torch.linalg.solve(M, g)


def est_quad_linear_robust(pts0, pts1, weight=None):
    """Transformation T (4 x 4, dtype of ``pts0``) with T pts0 ~ pts1, robust to outlier correspondences near the solution."""
    p0, q = pts0.detach().cpu().double(), pts1.detach().cpu().double()
    cur = p0
    trans = torch.eye(4, dtype=torch.float64)
    par = 1.0
    w = torch.ones((p0.shape[0], 1), dtype=torch.float64) if weight is None else weight.detach().cpu().double().reshape(-1, 1)
    for i in range(20):
        if i > 0 and i % 5 == 0:
            par /= 2.0
        step = get_trans(_gauss_newton_step(cur, q, w))
        cur = update_pcd(cur, step)
        w = compute_weights(cur, q, par)
        trans = step @ trans
    return trans.to(pts0.dtype)
