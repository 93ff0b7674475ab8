"""Host-side (CPU, loader-worker) helpers of the MinkowskiEngine operator surface.

Mirrors the three `ME.utils.*` entry points the reference calls from its DataLoader
worker processes (they run on CPU in the reference as well, never on the GPU):

* ``sparse_quantize``      -- util/misc.py:118, lib/colocation_data_loader.py:379,388
* ``batched_coordinates``  -- util/misc.py:120
* ``sparse_collate``       -- lib/colocation_data_loader.py:446

Pure numpy/torch integer work, bit-exact by construction.
"""
import numpy as np
import torch

__all__ = ["sparse_quantize", "batched_coordinates", "sparse_collate"]


def _as_numpy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy(), True
    return np.asarray(x), False


def sparse_quantize(coordinates, features=None, return_index=False, quantization_size=None):
    """Floor to the integer voxel grid and keep ONE row per occupied voxel.

    Returns ``coords_int32[N,3]`` (and ``index[N]``): the kept rows are the FIRST
    occurrence of every voxel, listed in ascending row order, so
    ``coords == floor(coordinates)[index]`` and ``index`` is sorted.
    Floor is toward -inf (``np.floor``), not truncation.
    """
    arr, was_torch = _as_numpy(coordinates)
    if arr.ndim != 2:
        raise ValueError("coordinates must be a [P, D] matrix")
    if quantization_size is not None:
        arr = arr / quantization_size
    disc = np.floor(arr).astype(np.int64)
    if disc.size and (np.abs(disc).max() >= (1 << 20)):
        raise ValueError("coordinate out of the supported +-2^20 voxel range")
    # lexicographic key; unique() sorts, return_index gives the first occurrence
    key = np.zeros(len(disc), dtype=np.int64)
    for d in range(disc.shape[1]):
        key = key * (1 << 21) + (disc[:, d] + (1 << 20))
    _, first = np.unique(key, return_index=True)
    first.sort()
    out = disc[first].astype(np.int32)
    if was_torch:
        out_c = torch.from_numpy(out)
        idx = torch.from_numpy(first.astype(np.int64))
    else:
        out_c, idx = out, first.astype(np.int64)
    if features is not None:
        feats = features[idx] if was_torch else np.asarray(features)[first]
        if return_index:
            return out_c, feats, idx
        return out_c, feats
    if return_index:
        return out_c, idx
    return out_c


def batched_coordinates(coords, dtype=torch.int32):
    """``[coords_b [N_b, D]] -> int32 [sum N_b, 1+D]`` with the batch id in column 0."""
    outs = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(np.asarray(c) if not isinstance(c, torch.Tensor) else c)
        c = torch.floor(c).to(dtype) if c.is_floating_point() else c.to(dtype)
        bcol = torch.full((c.shape[0], 1), b, dtype=dtype)
        outs.append(torch.cat([bcol, c], dim=1))
    if not outs:
        return torch.zeros((0, 4), dtype=dtype)
    return torch.cat(outs, dim=0)


def sparse_collate(coords, feats, labels=None, dtype=torch.int32):
    """Concatenate per-cloud coordinates (batch id = position in the list) and features."""
    if len(coords) != len(feats):
        raise ValueError("coords and feats lists differ in length")
    bcoords = batched_coordinates(coords, dtype=dtype)
    f = [torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x for x in feats]
    bfeats = torch.cat(f, dim=0) if f else torch.zeros((0, 1))
    if labels is not None:
        l = [torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x for x in labels]
        return bcoords, bfeats, torch.cat(l, dim=0)
    return bcoords, bfeats
