"""CPU ORACLE (test infrastructure, NOT product code): restatement of the validation step's host arithmetic --
util/transform_estimation.py:56-126 (est_quad_linear_robust and its helpers), lib/metrics.py:13-19 (corr_dist) and
lib/colocation_trainer.py:397-400 (evaluate_hit_ratio) -- in numpy float32, following the reference's formulation (the
stacked [3N, 6] system, explicit inverse of the normal matrix).  Pinned by tests/golden/validation_*.npz, which
tests/golden/make_golden.py captured from the reference's own functions.
Only tests/ may import this module."""
import numpy as np


def _rot(axis, a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(3, dtype=np.float32)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    m[i, i], m[j, j] = c, c
    if axis == 1:          # rot_y: +s above the diagonal  (util/transform_estimation.py:17-27)
        m[i, j], m[j, i] = s, -s
    else:                  # rot_x, rot_z                  (:5-15, :29-39)
        m[i, j], m[j, i] = -s, s
    return m


def get_trans(x):
    """:42-46: R = Rz(x2) Ry(x1) Rx(x0), t = x[3:]."""
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = _rot(2, x[2]) @ _rot(1, x[1]) @ _rot(0, x[0])
    T[:3, 3] = x[3:6]
    return T


def build_linear_system(p, q, w):
    """:56-81: three N-row blocks (x, y, z components), every row scaled by the point's weight."""
    n = len(p)
    A = np.zeros((3, n, 6), np.float32)
    A[0, :, 1], A[0, :, 2], A[0, :, 3] = p[:, 2], -p[:, 1], 1
    A[1, :, 0], A[1, :, 2], A[1, :, 4] = -p[:, 2], p[:, 0], 1
    A[2, :, 0], A[2, :, 1], A[2, :, 5] = p[:, 1], -p[:, 0], 1
    A = (A * w.reshape(1, n, 1)).reshape(3 * n, 6)
    b = ((q - p).T * w.reshape(1, n)).reshape(3 * n, 1)
    return A.astype(np.float32), b.astype(np.float32)


def est_quad_linear_robust(p0, q, weight=None):
    """:97-126."""
    p0, q = np.asarray(p0, np.float32), np.asarray(q, np.float32)
    cur, trans, par = p0, np.eye(4, dtype=np.float32), np.float32(1.0)
    w = np.ones((len(p0), 1), np.float32) if weight is None else np.asarray(weight, np.float32).reshape(-1, 1)
    for i in range(20):
        if i > 0 and i % 5 == 0:
            par = np.float32(par / 2)
        A, b = build_linear_system(cur, q, w)
        x = (np.linalg.inv(A.T @ A) @ A.T @ b).reshape(-1)                    # :84-86
        step = get_trans(x)
        cur = (cur @ step[:3, :3].T + step[:3, 3]).astype(np.float32)         # :49-53
        w = (par / (np.linalg.norm(cur - q, axis=1, keepdims=True) + par)).astype(np.float32)   # :89-90
        trans = step @ trans
    return trans


def corr_dist(est, gth, xyz0, max_dist=1.0):
    """lib/metrics.py:13-19 (weight=None)."""
    d = np.linalg.norm((xyz0 @ est[:3, :3].T + est[:3, 3]) - (xyz0 @ gth[:3, :3].T + gth[:3, 3]), axis=1)
    return float(np.minimum(d, max_dist).mean())


def hit_ratio(xyz0, xyz1, T, thresh):
    """lib/colocation_trainer.py:397-400."""
    d = np.sqrt((((xyz0 @ T[:3, :3].T + T[:3, 3]) - xyz1) ** 2).sum(1) + 1e-6)
    return float((d < thresh).mean())
