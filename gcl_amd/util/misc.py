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


def _point_channels(n_points, rgb, normal, check):
    """Input feature columns of ``extract_features`` as (array, offset, scale) triples: colours shifted to
    [-0.5, 0.5], normals halved, or a single column of ones when the cloud carries neither (util/misc.py:58-130)."""
    out = []
    for name, arr, lim, off, scale in (("color", rgb, "[0, 1]", -0.5, 1.0), ("normal", normal, "[-1, 1]", 0.0, 0.5)):
        if arr is None:
            continue
        arr = np.asarray(arr)
        if check:
            if arr.shape != (n_points, 3):
                raise AssertionError(f"{name} must be [{n_points}, 3], got {arr.shape}")
            if np.any(arr > 1):
                raise ValueError(f"Invalid {name}. {name.capitalize()} must range from {lim}")
        out.append((arr, off, scale))
    return out


def extract_features(model, xyz, rgb=None, normal=None, voxel_size=0.05, device=None, skip_check=False, is_eval=True):
    """Features of one raw cloud: ``(xyz[kept], F)`` with one row per occupied voxel (interface of util/misc.py:58-130;
    called by demo.py:35-40).  The voxelisation runs on the GPU (``sparse_quantize_gpu`` -- bit-identical to
    ``ME.utils.sparse_quantize(np.floor(xyz / voxel_size), return_index=True)``: first point of every voxel, ascending)
    and the per-voxel input features are gathered there too, so only the raw points cross PCIe.  float32 clouds are
    divided on the device (correctly rounded fp32 quotient = numpy's); any other dtype keeps numpy's own quotient on
    the host and only the voxel hash runs on the device.  There is no CPU path: ``device`` defaults to cuda:0."""
    from gcl_amd.lib.colocation_data_gpu import sparse_quantize_gpu, unique_coords_gpu
    dev = torch.device("cuda:0" if device is None else device)
    if is_eval:
        model.eval()
    pts = np.asarray(xyz)
    if not skip_check and (pts.ndim != 2 or pts.shape[1] != 3):
        raise AssertionError(f"xyz must be [N, 3], got {pts.shape}")
    channels = _point_channels(len(pts), rgb, normal, check=not skip_check)
    with torch.cuda.device(dev):
        if pts.dtype == np.float32:
            coords, kept = sparse_quantize_gpu(torch.from_numpy(np.ascontiguousarray(pts)).to(dev), voxel_size)
        else:
            cells = np.zeros((len(pts), 4), dtype=np.int32)            # column 0 = batch id 0
            cells[:, 1:] = np.floor(pts / voxel_size)
            coords, kept = unique_coords_gpu(torch.from_numpy(cells).to(dev))
        if channels:
            cols = [(torch.from_numpy(np.ascontiguousarray(a)).to(dev)[kept].double() + off) * scale
                    for a, off, scale in channels]
            feats = torch.cat(cols, dim=1).float()
        else:
            feats = torch.ones((len(kept), 1), dtype=torch.float32, device=dev)
        out = model(ME.SparseTensor(feats, coordinates=coords))
    return xyz[kept.cpu().numpy()], out.F
