"""Pairwise feature distances and the validation metric (interface of lib/metrics.py:13-29).

``pdist`` keeps the reference's signature and returns the full [M, M'] matrix (torch ops, compatibility only);
the hot path never materialises it: ``pdist_min`` returns the row minimum and arg-minimum from one HIP kernel
(the only way the reference consumes pdist on the hot path: lib/colocation_trainer.py:510-512, lib/eval.py:25-29).
"""
import torch

from gcl_amd import _lib


def corr_dist(est, gth, xyz0, xyz1, weight=None, max_dist=1):
    """Mean (clamped) distance between the points moved by the estimated and by the true transformation
    (lib/metrics.py:13-19; the validation step's "loss", lib/colocation_trainer.py:343).  ``xyz1`` is unused, as there."""
    moved_est = xyz0 @ est[:3, :3].t() + est[:3, 3]
    moved_gth = xyz0 @ gth[:3, :3].t() + gth[:3, 3]
    dists = torch.clamp(torch.sqrt(((moved_est - moved_gth) ** 2).sum(1)), max=max_dist)
    if weight is not None:
        dists = weight * dists
    return dists.mean()


def pdist(A, B, dist_type="L2"):
    if dist_type not in ("L2", "SquareL2"):
        raise NotImplementedError("Not implemented")
    out = torch.empty((A.shape[0], B.shape[0]), dtype=A.dtype, device=A.device)
    step = max(1, (1 << 26) // max(1, B.shape[0] * A.shape[1]))        # bound the broadcast temp to 256 MB
    for i in range(0, A.shape[0], step):
        out[i:i + step] = torch.sum((A[i:i + step].unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(out + 1e-7) if dist_type == "L2" else out


def pdist_min(A, B, dist_type="L2", rows_a=None, rows_b=None):
    """Row-wise (min, argmin) of pdist(A[rows_a], B[rows_b]) without forming the matrix.  Ties -> lowest index."""
    lib = _lib.require_gpu()
    if dist_type not in ("L2", "SquareL2"):
        raise NotImplementedError("Not implemented")
    A, B = A.detach().contiguous(), B.detach().contiguous()
    ma = A.shape[0] if rows_a is None else rows_a.shape[0]
    mb = B.shape[0] if rows_b is None else rows_b.shape[0]
    dmin = torch.empty(ma, dtype=torch.float32, device=A.device)
    arg = torch.empty(ma, dtype=torch.int32, device=A.device)
    ns = lib.gcl_nn_rowmin_scratch_len(ma, mb)
    scratch = torch.empty(ns, dtype=torch.int32, device=A.device) if ns else None
    _lib.check(lib.gcl_nn_rowmin(_lib.ptr(A, torch.float32), _lib.ptr(rows_a, torch.int64), ma,
                                 _lib.ptr(B, torch.float32), _lib.ptr(rows_b, torch.int64), mb, A.shape[1],
                                 1 if dist_type == "L2" else 0, _lib.ptr(scratch), _lib.ptr(dmin), _lib.ptr(arg),
                                 _lib.stream()), "gcl_nn_rowmin")
    return dmin, arg
