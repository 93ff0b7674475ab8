"""Loader work on the GPU: voxelisation and co-location group building.

Replaces, with identical results, the CPU side of ``ColocationKittiDataset.__getitem__``
(lib/colocation_data_loader.py:378-421) that dominates the reference's data time:

* ``sparse_quantize_gpu``      -- ``ME.utils.sparse_quantize(xyz / voxel_size, return_index=True)`` (:379, :388)
* ``colocation_groups_gpu``    -- ``get_matching_indices_colocation`` (util/pointcloud.py:69-132; one open3d KD-tree
  radius query per point per cloud in a Python loop) as two kernels over the voxel hash map of the sample
* ``build_sample_gpu`` / ``collate_gpu`` -- the per-sample tuple and the batch dict of ``collate_colocation_fn``
  (:424-475) with every tensor already resident on the device (``index_hash`` is not produced: the GPU loss does not
  need it, see gcl_amd/lib/colocation_trainer.py).
"""
import ctypes

import numpy as np
import torch

from gcl_amd import _lib


def _cap(n):
    cap = 64
    while cap < 2 * n:
        cap *= 2
    return cap


def unique_coords_gpu(raw, return_table=False):
    """First-occurrence unique rows of int32 [P,4] voxel coordinates on the GPU (gcl_unique_coords): (coords [N,4],
    index int64 [N] ascending) -- the ``return_index=True`` half of ``ME.utils.sparse_quantize``."""
    lib = _lib.require_gpu()
    raw = raw.contiguous()
    P = raw.shape[0]
    dev = raw.device
    cap = _cap(P)
    table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    scratch = torch.empty(lib.gcl_scan_scratch_len(P), dtype=torch.int32, device=dev)
    coords = torch.empty((P, 4), dtype=torch.int32, device=dev)
    index = torch.empty(P, dtype=torch.int64, device=dev)
    meta = torch.empty(8, dtype=torch.int32, device=dev)
    off = lambda t, e: ctypes.c_void_p(t.data_ptr() + e * t.element_size())
    _lib.check(lib.gcl_unique_coords(_lib.ptr(raw, torch.int32), P, _lib.ptr(table), cap, _lib.ptr(scratch),
                                     _lib.ptr(coords), _lib.ptr(index), off(meta, 0), off(meta, 4), _lib.stream()),
               "gcl_unique_coords")
    m = meta.tolist()
    if m[4]:
        raise ValueError(f"{m[4]} points outside the packable voxel range")
    n = m[0]
    if return_table:
        return coords[:n], index[:n], (table, cap)
    return coords[:n], index[:n]


def sparse_quantize_gpu(xyz, voxel_size, batch_id=0, return_table=False):
    """``xyz`` float32 [P,3] on the GPU -> (coords int32 [N,4] with ``batch_id`` in column 0, index int64 [N]):
    coords = floor(xyz / voxel_size), first occurrence per voxel, index ascending -- exactly
    ``ME.utils.sparse_quantize(xyz / voxel_size, return_index=True)`` + the batch column."""
    lib = _lib.require_gpu()
    xyz = xyz.contiguous()
    P = xyz.shape[0]
    raw = torch.empty((P, 4), dtype=torch.int32, device=xyz.device)
    _lib.check(lib.gcl_voxel_coords(_lib.ptr(xyz, torch.float32), P, float(voxel_size), int(batch_id), _lib.ptr(raw),
                                    _lib.stream()), "gcl_voxel_coords")
    return unique_coords_gpu(raw, return_table)


def colocation_groups_gpu(xyz_own, xyz_cf, coords, n_center, n_clouds, list_M, voxel_size, radius, K=5):
    """Groups of one sample.  ``xyz_own`` / ``xyz_cf`` float32 [Ntot,3] (voxel representatives, centre cloud first, in
    their own sensor frames / in the centre frame), ``coords`` int32 [Ntot,4] = floor(xyz_own / voxel) with the cloud
    id in column 0, ``list_M`` the neighbour -> centre transforms (4x4).  Returns device tensors
    (group int32 [G], index int64 [sum], finest_flag bool [sum]) with the reference's member order."""
    lib = _lib.require_gpu()
    dev = xyz_own.device
    ntot = xyz_own.shape[0]
    cap = _cap(ntot)
    table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    status = torch.empty(4, dtype=torch.int32, device=dev)
    coords = coords.contiguous()          # named, not temporaries inside the call: a temporary's block is reused at once
    xyz_own, xyz_cf = xyz_own.contiguous(), xyz_cf.contiguous()
    _lib.check(lib.gcl_coords_insert(_lib.ptr(coords, torch.int32), ntot, _lib.ptr(table), cap,
                                     _lib.ptr(status), _lib.stream()), "gcl_coords_insert")
    to_cloud = np.zeros((n_clouds, 12), dtype=np.float64)
    to_cloud[0] = np.eye(4)[:3].reshape(-1)
    for j, M in enumerate(list_M):
        to_cloud[j + 1] = np.linalg.inv(np.asarray(M, dtype=np.float64))[:3].reshape(-1)
    hits = torch.empty((n_center, n_clouds, K), dtype=torch.int32, device=dev)
    cnt = torch.empty((n_center, n_clouds), dtype=torch.int32, device=dev)
    rng = torch.empty((n_center, n_clouds), dtype=torch.float64, device=dev)
    _lib.check(lib.gcl_colocation_hits(_lib.ptr(xyz_own, torch.float32),
                                       _lib.ptr(xyz_cf, torch.float32), n_center, n_clouds,
                                       to_cloud.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _lib.ptr(table), cap,
                                       float(1.0 / voxel_size), float(radius), K, _lib.ptr(hits), _lib.ptr(cnt),
                                       _lib.ptr(rng), _lib.stream()), "gcl_colocation_hits")
    scratch = torch.empty(4 * n_center + n_center // 2048 + 64, dtype=torch.int32, device=dev)
    group = torch.empty(n_center, dtype=torch.int32, device=dev)
    index = torch.empty(n_center * n_clouds * K, dtype=torch.int64, device=dev)
    finest = torch.empty(n_center * n_clouds * K, dtype=torch.uint8, device=dev)
    totals = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.check(lib.gcl_colocation_emit(_lib.ptr(hits), _lib.ptr(cnt), _lib.ptr(rng), n_center, n_clouds, K,
                                       _lib.ptr(scratch), _lib.ptr(group), _lib.ptr(index), _lib.ptr(finest),
                                       _lib.ptr(totals), _lib.stream()), "gcl_colocation_emit")
    g, e = totals.tolist()
    st = status.tolist()
    if st[0] or st[1]:
        raise ValueError("coordinates of the sample are not unique / out of range")
    return group[:g], index[:e], finest[:e].bool()


def build_sample_gpu(xyz_list, list_M, voxel_size, radius, device, K=5):
    """One training sample from raw sensor-frame clouds (centre first): voxelise every cloud, transform the neighbours
    into the centre frame (host fp64 -> fp32, as the reference), build the groups.  Returns a dict of device tensors."""
    dev = torch.device(device)
    coords, xyz_own, xyz_cf = [], [], []
    for c, xyz in enumerate(xyz_list):
        x = torch.as_tensor(xyz, dtype=torch.float32, device=dev)
        cc, idx = sparse_quantize_gpu(x, voxel_size, batch_id=c)
        xo = x[idx]
        coords.append(cc)
        xyz_own.append(xo)
        if c == 0:
            xyz_cf.append(xo)
        else:
            M = np.asarray(list_M[c - 1], dtype=np.float64)
            xh = xo.cpu().numpy()
            xyz_cf.append(torch.from_numpy((xh @ M[:3, :3].T + M[:3, 3]).astype(np.float32)).to(dev))
    C, XO, XC = torch.cat(coords), torch.cat(xyz_own), torch.cat(xyz_cf)
    group, index, finest = colocation_groups_gpu(XO, XC, C, len(coords[0]), len(coords), list_M, voxel_size, radius, K)
    return {"coords": C, "xyz_own": XO, "group": group, "index": index, "finest_flag": finest,
            "n_rows": C.shape[0], "n_center": len(coords[0])}


def collate_gpu(samples):
    """``collate_colocation_fn`` (lib/colocation_data_loader.py:424-475) on device tensors: batch ids increment per
    cloud, ``index`` gets the sample's row offset, features are ones."""
    Cs, idx, grp, fin, lengths = [], [], [], [], []
    start, cloud0 = 0, 0
    for s in samples:
        C = s["coords"].clone()
        C[:, 0] += cloud0
        cloud0 = int(C[-1, 0].item()) + 1 if len(C) else cloud0
        Cs.append(C)
        idx.append(s["index"] + start)
        grp.append(s["group"])
        fin.append(s["finest_flag"])
        start += s["n_rows"]
        lengths.append(s["n_rows"])
    C = torch.cat(Cs)
    return {"sinput_C": C, "sinput_F": torch.ones((len(C), 1), dtype=torch.float32, device=C.device),
            "group": torch.cat(grp), "index": torch.cat(idx), "finest_flag": torch.cat(fin), "index_hash": None,
            "batch_lengths": lengths}
