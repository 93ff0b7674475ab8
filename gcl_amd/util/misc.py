"""Pair hashing + feature extraction (interfaces of util/misc.py:29-55 and :58-130)."""
import numpy as np
import torch

import gcl_amd.MinkowskiEngine as ME


def _neg_hash(inds1, inds2, M):
    """Symmetric pair key min(i*M + j, i + j*M) == min(i,j)*M + max(i,j) (util/misc.py:39-40)."""
    i1, i2 = np.asarray(inds1, dtype=np.int64), np.asarray(inds2, dtype=np.int64)
    return np.minimum(i1 * M + i2, i1 + i2 * M)


def _exhaustive_hash(index_split, M):
    """Keys of every in-group pair, group by group, first member major (util/misc.py:29-36); vectorised per group."""
    res = []
    for idx in index_split:
        idx = np.asarray(idx.cpu() if isinstance(idx, torch.Tensor) else idx, dtype=np.int64)
        g = len(idx)
        if g < 2:
            continue
        a, b = np.triu_indices(g, k=1)
        res.append(np.minimum(idx[a] + idx[b] * M, idx[a] * M + idx[b]))
    return np.concatenate(res, axis=0) if res else np.zeros(0, dtype=np.int64)


def _hash(arr, M):
    """Positional hash sum_d arr[:, d] * M**d (util/misc.py:43-55)."""
    cols = [arr[:, d] for d in range(arr.shape[1])] if isinstance(arr, np.ndarray) else list(arr)
    h = np.zeros(len(cols[0]), dtype=np.int64)
    for d, c in enumerate(cols):
        h += np.asarray(c, dtype=np.int64) * M ** d
    return h


def extract_features(model, xyz, rgb=None, normal=None, voxel_size=0.05, device=None, skip_check=False, is_eval=True):
    """numpy cloud -> voxelise -> SparseTensor -> model -> (xyz[inds], F)   (util/misc.py:58-130)."""
    if is_eval:
        model.eval()
    if not skip_check:
        assert xyz.shape[1] == 3
        if rgb is not None:
            assert len(rgb) == len(xyz) and rgb.shape[1] == 3
            if np.any(rgb > 1):
                raise ValueError("Invalid color. Color must range from [0, 1]")
        if normal is not None:
            assert len(normal) == len(xyz) and normal.shape[1] == 3
            if np.any(normal > 1):
                raise ValueError("Invalid normal. Normal must range from [-1, 1]")
    if device is None:
        device = torch.device("cuda:0")
    feats = []
    if rgb is not None:
        feats.append(rgb - 0.5)
    if normal is not None:
        feats.append(normal / 2)
    if rgb is None and normal is None:
        feats.append(np.ones((len(xyz), 1)))
    feats = np.hstack(feats)
    coords, inds = ME.utils.sparse_quantize(np.floor(xyz / voxel_size), return_index=True)
    coords = ME.utils.batched_coordinates([coords])
    stensor = ME.SparseTensor(torch.tensor(feats[inds], dtype=torch.float32), coordinates=coords, device=device)
    return xyz[inds], model(stensor).F
