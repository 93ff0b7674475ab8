"""Seeded synthetic KITTI-shaped inputs for the hot path (no dataset is reachable offline).

Emits exactly the dictionaries the reference's collate functions hand to the hot path:

* training batch  -- keys of ``collate_colocation_fn`` (lib/colocation_data_loader.py:462-475),
  built the way ``ColocationKittiDataset.__getitem__`` does (:315-421): one centre scan +
  ``num_neighborhood`` neighbour scans of the same scene from sensor poses 5..60 m away,
  random rotation (<=45 deg) + scale U(0.8, 1.2), voxelise, radius-search groups
  (``get_matching_indices_colocation``, util/pointcloud.py:69-132, K=5).
* eval pair       -- keys of ``collate_debug_pair_fn`` (lib/complement_data_loader.py:1323-1333).

The scene is an HDL-64-like ray cast (64 beams +2..-24.8 deg, 1800 azimuths, ground plane at
z=-1.73 m, random boxes, 2 cm range noise, 80 m max range) -- SURVEY.md section 8.
Everything here is CPU/numpy loader work (as it is in the reference) and is NOT timed by bench.py.
"""
import numpy as np
import torch

from gcl_amd.MinkowskiEngine import utils as me_utils

N_BEAMS = 64
N_AZIMUTH = 1800
SENSOR_HEIGHT = 1.73
MAX_RANGE = 80.0


def make_scene(seed, n_boxes=60, extent=70.0):
    """Axis-aligned boxes (cars / walls / poles) scattered on the ground plane, in WORLD frame."""
    rng = np.random.RandomState(seed)
    centers = np.stack([rng.uniform(-extent, extent + 60.0, n_boxes),
                        rng.uniform(-extent, extent, n_boxes)], axis=1)
    kind = rng.randint(0, 3, n_boxes)
    size = np.where(kind[:, None] == 0, rng.uniform([3.5, 1.6, 1.4], [5.0, 2.1, 2.0], (n_boxes, 3)),
                    np.where(kind[:, None] == 1, rng.uniform([6.0, 0.4, 2.5], [25.0, 1.0, 8.0], (n_boxes, 3)),
                             rng.uniform([0.3, 0.3, 3.0], [0.8, 0.8, 9.0], (n_boxes, 3))))
    swap = rng.rand(n_boxes) < 0.5
    size[swap, 0], size[swap, 1] = size[swap, 1].copy(), size[swap, 0].copy()
    lo = np.concatenate([centers - size[:, :2] / 2, np.full((n_boxes, 1), -SENSOR_HEIGHT)], axis=1)
    hi = np.concatenate([centers + size[:, :2] / 2, (-SENSOR_HEIGHT + size[:, 2])[:, None]], axis=1)
    # keep the road corridor around y=0 free so that every sensor pose along x sees something
    keep = ~((lo[:, 1] < 2.5) & (hi[:, 1] > -2.5))
    return {"lo": lo[keep], "hi": hi[keep]}


def raycast(scene, sensor_xyz, seed, yaw=0.0):
    """Cast the 64x1800 beam fan from ``sensor_xyz`` (world); return hits in the SENSOR frame [P,3] float32."""
    rng = np.random.RandomState(seed)
    elev = np.deg2rad(np.linspace(2.0, -24.8, N_BEAMS))
    azim = np.linspace(-np.pi, np.pi, N_AZIMUTH, endpoint=False) + yaw
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d = np.stack([ce * np.cos(azim)[None], ce * np.sin(azim)[None], se * np.ones_like(azim)[None]], axis=-1)
    d = d.reshape(-1, 3)
    o = np.asarray(sensor_xyz, dtype=np.float64)
    t = np.full(len(d), np.inf)
    # ground plane z = -SENSOR_HEIGHT (world)
    down = d[:, 2] < -1e-6
    tg = np.where(down, (-SENSOR_HEIGHT - o[2]) / np.where(down, d[:, 2], -1.0), np.inf)
    t = np.minimum(t, tg)
    lo, hi = scene["lo"], scene["hi"]
    inv = 1.0 / np.where(np.abs(d) < 1e-9, 1e-9, d)
    for b in range(len(lo)):
        t0 = (lo[b] - o) * inv
        t1 = (hi[b] - o) * inv
        tn = np.minimum(t0, t1).max(axis=1)
        tf = np.maximum(t0, t1).min(axis=1)
        hit = (tn <= tf) & (tf > 0) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)
    ok = np.isfinite(t) & (t < MAX_RANGE)
    t = t[ok] + rng.normal(0.0, 0.02, ok.sum())
    pts_sensor = d[ok] * t[:, None]
    if yaw != 0.0:
        c, s = np.cos(-yaw), np.sin(-yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        pts_sensor = pts_sensor @ R.T
    return pts_sensor.astype(np.float32)


def _random_rotation(rng, max_angle):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    return T


def _apply(T, xyz):
    return (xyz @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def colocation_groups(center_xyz, nghb_xyz, list_M, radius, K=5):
    """Group builder with the semantics of ``get_matching_indices_colocation`` (util/pointcloud.py:69-132).

    For every centre voxel: <=K centre-cloud radius hits (nearest first, i.e. itself first), then <=K hits
    per neighbour cloud (neighbour aligned into the centre frame by ``list_M[j]``); a group exists only if
    at least one neighbour cloud matched.  ``finest`` marks the member whose own sensor is closest
    (sequential strict-< update in the reference == first arg-min).
    Row ids: centre rows 0..Nc-1, neighbour j offset by Nc + sum_{i<j} N_i.  Vectorised over centre voxels.
    """
    from scipy.spatial import cKDTree

    Nc = len(center_xyz)
    offs = np.concatenate([[0, Nc], Nc + np.cumsum([len(x) for x in nghb_xyz])]).astype(np.int64)
    d, i = cKDTree(center_xyz).query(center_xyz, k=K, distance_upper_bound=radius)
    cols_idx, cols_ok = [i.astype(np.int64)], [np.isfinite(d)]
    first_range = [np.linalg.norm(center_xyz.astype(np.float64), axis=1)]   # centre voxel's own sensor range (fp64, as o3d)
    for j, xyz in enumerate(nghb_xyz):
        dj, ij = cKDTree(_apply(list_M[j], xyz)).query(center_xyz, k=K, distance_upper_bound=radius)
        ok = np.isfinite(dj)
        rng_j = np.linalg.norm(xyz.astype(np.float64), axis=1)
        first = np.where(ok[:, 0], rng_j[np.minimum(ij[:, 0], len(xyz) - 1)], np.inf)
        cols_idx.append(ij.astype(np.int64) + offs[j + 1])
        cols_ok.append(ok)
        first_range.append(first)
    idx = np.concatenate(cols_idx, axis=1)                       # [Nc, K * (1 + n_nghb)], reference member order
    ok = np.concatenate(cols_ok, axis=1)
    has_nghb = ok[:, K:].any(axis=1)
    best = np.argmin(np.stack(first_range, axis=1), axis=1)      # 0 = centre, j + 1 = neighbour j
    before = np.cumsum(ok, axis=1) - ok                          # members listed before each column
    fpos = np.where(best == 0, 0, before[np.arange(Nc), best * K])
    rank = before                                                # position of a valid column inside its group
    flag = ok & (rank == fpos[:, None])
    keep = ok & has_nghb[:, None]
    group = ok.sum(axis=1)[has_nghb]
    return group.tolist(), idx[keep].tolist(), flag[keep].tolist()


def fixed_size_groups(center_xyz, nghb_xyz, list_M, radius, size=16, K=5):
    """BASELINE.json's 'positive-group size 16' variant: keep radius groups with >= size members, cut to size
    (the finest member is moved inside the kept prefix so that exactly one flag stays set)."""
    group, index, finest = colocation_groups(center_xyz, nghb_xyz, list_M, radius, K=K)
    g2, i2, f2 = [], [], []
    p = 0
    for g in group:
        idx, fl = index[p:p + g], finest[p:p + g]
        p += g
        if g < size:
            continue
        fpos = fl.index(True)
        if fpos >= size:
            idx[size - 1], fl[size - 1] = idx[fpos], True
        g2.append(size)
        i2 += idx[:size]
        f2 += fl[:size]
    return g2, i2, f2


def exhaustive_hash(index, group, M):
    """Vectorised numpy equivalent of ``_exhaustive_hash`` (util/misc.py:29-36): the symmetric key
    min(i + j*M, i*M + j) of every in-group pair, group by group, i-major."""
    out = []
    p = 0
    index = np.asarray(index, dtype=np.int64)
    for g in group:
        idx = index[p:p + g]
        p += g
        for a in range(g - 1):
            out.append(np.minimum(idx[a] + idx[a + 1:] * M, idx[a] * M + idx[a + 1:]))
    if not out:
        return np.zeros(0, dtype=np.int64)
    return np.concatenate(out)


def make_raw_sample(seed, voxel_size=0.3, num_neighborhood=6, min_dist=5.0, max_dist=60.0, search_mult=1.5,
                    random_rotation=True, random_scale=True, n_boxes=60):
    """What ``ColocationKittiDataset.__getitem__`` holds BEFORE voxelisation (lib/colocation_data_loader.py:315-370): the
    centre scan and its neighbours after the random rotation / scale, ``list_M`` (neighbour -> centre) and the scaled
    matching radius -- the input of ``gcl_amd.lib.colocation_data_gpu.build_batch_gpu``.  ``"rng"`` is the sample's
    generator positioned where ``make_train_sample`` draws the feature jitter (after voxelisation, :414-415)."""
    rng = np.random.RandomState(seed + 7919)
    scene = make_scene(seed, n_boxes=n_boxes)
    center_pos = np.array([0.0, 0.0, 0.0])
    xyz = raycast(scene, center_pos, seed * 101 + 1)
    shifts = np.linspace(min_dist, max_dist, num_neighborhood)
    xyz_cmpl, list_M = [], []
    for j, s in enumerate(shifts):
        pos = np.array([s, rng.uniform(-0.5, 0.5), 0.0])
        xyz_cmpl.append(raycast(scene, pos, seed * 101 + 2 + j))
        M = np.eye(4)
        M[:3, 3] = pos - center_pos          # neighbour sensor frame -> centre sensor frame
        list_M.append(M)
    search = voxel_size * search_mult
    if random_rotation:
        T0 = _random_rotation(rng, np.pi / 4)
        xyz = _apply(T0, xyz)
        for j in range(len(xyz_cmpl)):
            Tc = _random_rotation(rng, np.pi / 4)
            xyz_cmpl[j] = _apply(Tc, xyz_cmpl[j])
            list_M[j] = T0 @ list_M[j] @ np.linalg.inv(Tc)
    if random_scale and rng.rand() < 0.95:
        scale = 0.8 + 0.4 * rng.rand()
        search *= scale
        xyz = (scale * xyz).astype(np.float32)
        for j in range(len(xyz_cmpl)):
            xyz_cmpl[j] = (scale * xyz_cmpl[j]).astype(np.float32)
            list_M[j][:3, 3] *= scale
    return {"xyz": [np.ascontiguousarray(xyz, dtype=np.float32)] +
                   [np.ascontiguousarray(x, dtype=np.float32) for x in xyz_cmpl],
            "list_M": list_M, "radius": float(search), "rng": rng}


def raw_sample_jitter(raw_samples):
    """The ``jitter`` callable of ``build_batch_gpu`` for ``make_raw_sample`` dicts: the centre cloud's feature jitter as
    ``make_train_sample`` draws it (lib/transforms.py:24-29: N(0, 0.01) with probability 0.95), from COPIES of the samples'
    generators (a raw sample can be fed any number of times)."""
    def jitter(si, n_center):
        rng = np.random.RandomState()
        rng.set_state(raw_samples[si]["rng"].get_state())
        if rng.rand() < 0.95:
            return rng.normal(0.0, 0.01, (n_center, 1)).astype(np.float32)
        return None
    return jitter


def make_train_sample(seed, voxel_size=0.3, num_neighborhood=6, min_dist=5.0, max_dist=60.0,
                      search_mult=1.5, random_rotation=True, random_scale=True, group_mode="radius",
                      n_boxes=60):
    """One ``__getitem__`` tuple (lib/colocation_data_loader.py:419-421) on the synthetic scene."""
    raw = make_raw_sample(seed, voxel_size, num_neighborhood, min_dist, max_dist, search_mult, random_rotation,
                          random_scale, n_boxes)
    rng, xyz, xyz_cmpl, list_M, search = raw["rng"], raw["xyz"][0], raw["xyz"][1:], raw["list_M"], raw["radius"]
    _, sel = me_utils.sparse_quantize(xyz / voxel_size, return_index=True)
    xyz_th = xyz[sel]
    xyz_cmpl_th = []
    for j in range(len(xyz_cmpl)):
        _, s = me_utils.sparse_quantize(xyz_cmpl[j] / voxel_size, return_index=True)
        xyz_cmpl_th.append(xyz_cmpl[j][s])
    if group_mode == "radius":
        group, index, finest = colocation_groups(xyz_th, xyz_cmpl_th, list_M, search)
    elif group_mode == "fixed16":
        group, index, finest = fixed_size_groups(xyz_th, xyz_cmpl_th, list_M, search, size=16)
    else:
        raise ValueError(group_mode)
    coords = [np.floor(xyz_th / voxel_size).astype(np.int32)] + \
             [np.floor(x / voxel_size).astype(np.int32) for x in xyz_cmpl_th]
    feats = [np.ones((len(c), 1), dtype=np.float32) for c in coords]
    # Jitter on the centre cloud only (lib/transforms.py:24-29): N(0, 0.01) with prob. 0.95
    if rng.rand() < 0.95:
        feats[0] = feats[0] + rng.normal(0.0, 0.01, feats[0].shape).astype(np.float32)
    return (xyz_th, xyz_cmpl_th, coords, feats, group, index, finest, list_M)


def sample_search_radius(seed, voxel_size=0.3, num_neighborhood=6, search_mult=1.5, random_rotation=True,
                         random_scale=True):
    """The (scaled) matching radius make_train_sample used for ``seed`` (it replays the generator's draws)."""
    rng = np.random.RandomState(seed + 7919)
    for _ in range(num_neighborhood):
        rng.uniform(-0.5, 0.5)
    if random_rotation:
        for _ in range(1 + num_neighborhood):
            _random_rotation(rng, np.pi / 4)
    search = voxel_size * search_mult
    if random_scale and rng.rand() < 0.95:
        search *= 0.8 + 0.4 * rng.rand()
    return search


def collate_train(samples):
    """``collate_colocation_fn`` (lib/colocation_data_loader.py:424-475) on ``make_train_sample`` tuples."""
    index_batch, group_batch, finest_batch, batch_lengths = [], [], [], []
    coords_all, feats_all = [], []
    start = 0
    for (_, _, coords, feats, group, index, finest, _) in samples:
        if len(group):
            index_batch.append(np.asarray(index, dtype=np.int64) + start)
        n = int(sum(len(c) for c in coords))
        start += n
        batch_lengths.append(n)
        coords_all += coords
        feats_all += feats
        group_batch += list(group)
        finest_batch += list(finest)
    C, F = me_utils.sparse_collate(coords_all, feats_all)
    index = np.concatenate(index_batch) if index_batch else np.zeros(0, dtype=np.int64)
    return {
        "sinput_C": C,
        "sinput_F": F.float(),
        "group": torch.tensor(group_batch, dtype=torch.int32),
        "index": torch.from_numpy(index).long(),
        "finest_flag": torch.tensor(finest_batch, dtype=torch.bool),
        "index_hash": exhaustive_hash(index, group_batch, len(C)),
        "batch_lengths": batch_lengths,
        "pcd_center": [s[0] for s in samples],
    }


def make_train_batch(seed, batch_size=4, voxel_size=0.3, group_mode="radius", **kw):
    """The hot path's training input: ``batch_size`` samples x (1 + num_neighborhood) clouds."""
    return collate_train([make_train_sample(seed * 1000 + b, voxel_size, group_mode=group_mode, **kw)
                          for b in range(batch_size)])


def make_eval_pair(seed, voxel_size=0.3, baseline=10.0, n_boxes=60):
    """``collate_debug_pair_fn``-shaped dict (lib/complement_data_loader.py:1323-1333) for one pair."""
    scene = make_scene(seed, n_boxes=n_boxes)
    out = {}
    xyzs = []
    for k, x in enumerate([0.0, baseline]):
        xyz = raycast(scene, np.array([x, 0.0, 0.0]), seed * 37 + k)
        _, sel = me_utils.sparse_quantize(xyz / voxel_size, return_index=True)
        xyz = xyz[sel]
        xyzs.append(xyz)
        coords = me_utils.batched_coordinates([np.floor(xyz / voxel_size).astype(np.int32)])
        out[f"pcd{k}"] = (torch.from_numpy(xyz),)
        out[f"sinput{k}_C"] = coords
        out[f"sinput{k}_F"] = torch.ones((len(coords), 1), dtype=torch.float32)
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = -baseline                     # maps cloud 0 (sensor-0 frame) into sensor-1 frame
    out["T_gt"] = torch.from_numpy(T)
    out["len_batch"] = [[len(xyzs[0]), len(xyzs[1])]]
    return out


def make_twin_eval_pair(seed, inlier_share=0.3, voxel_size=0.3, shift_voxels=(16, 8, 0), gap_voxels=16, n_boxes=60):
    """An eval pair (keys of ``make_eval_pair``) with a CONTROLLED share of true correspondences, for timing the eval loop on
    registrations that have something to find with an UNTRAINED network (bench.py ``secondary``, configs[4]).

    A random-init network gives unrelated views of a scene unrelated features: every putative correspondence is an outlier,
    SC2-PCR's compatibility matrix is empty and the registration times what it never does on real data.  Here the two clouds
    share a part exactly: the voxels of scan A on one side of a plane y = cut appear in cloud 0 and, moved by a multiple of 8
    voxels (every U-Net level stays aligned, so twins get bit-equal features wherever their receptive fields hold twins
    only), in cloud 1; the other side of the plane (``gap_voxels`` away) is filled with two UNRELATED scans, one per cloud.
    ``inlier_share`` sets the shared part's share of each cloud's voxels; the share of true correspondences among the
    loop's putative ones is lower (only a part of the twins survives the loop's 5000-row subsamples) and is what
    bench.py measures and prints.  Input features carry a per-voxel jitter (equal on twins) so that no two voxels tie.
    T_gt maps cloud 0 onto cloud 1 (scripts/test_kitti.py:155 applies it to xyz0)."""
    sh = np.asarray(shift_voxels, dtype=np.int64)
    assert (sh % 8 == 0).all(), "twins stay aligned on every level only for shifts that are multiples of 8 voxels"

    def scan(scene_seed, ray_seed):
        xyz = raycast(make_scene(scene_seed, n_boxes=n_boxes), np.zeros(3), ray_seed)
        _, sel = me_utils.sparse_quantize(xyz / voxel_size, return_index=True)
        xyz = xyz[sel]
        return xyz, np.floor(xyz / voxel_size).astype(np.int64)

    xa, ca = scan(seed, seed * 37)
    n_common = int(round(len(ca) * float(inlier_share)))
    order = np.argsort(ca[:, 1], kind="stable")
    cut = ca[order[max(0, n_common - 1)], 1] if n_common > 0 else ca[:, 1].min() - 1
    common = ca[:, 1] <= cut
    clouds = []
    for k in (0, 1):
        xb, cb = scan(seed * 1000 + 17 + k, seed * 37 + 5 + k)
        # the unrelated scan, moved so that its own densest part (around its sensor) sits beyond the gap
        far = np.argsort(cb[:, 1], kind="stable")
        n_other = max(0, len(ca) - int(common.sum()))
        # keep the n_other voxels of lowest y, then translate them to start at cut + gap
        keep = far[:n_other]
        dy = (cut + gap_voxels) - cb[keep, 1].min() if n_other else 0
        dy = int(-((-dy) // 8) * 8)               # a multiple of 8 voxels, rounding away from the shared part
        cbk, xbk = cb[keep].copy(), xb[keep].copy()
        # cloud 1's unrelated part also sits 3 voxels higher and 5 further along x: the ground plane, which every scan shares,
        # is not the T_gt image of cloud 0's (a handful of box voxels in 10^4 still coincide by chance)
        off = np.array([5 * k, dy, 3 * k], dtype=np.int64)
        cbk += off
        xbk += (off * voxel_size).astype(np.float32)
        move = sh if k == 1 else np.zeros(3, dtype=np.int64)
        c = np.concatenate([ca[common] + move, cbk + move])
        x = np.concatenate([xa[common] + (move * voxel_size).astype(np.float32),
                            xbk + (move * voxel_size).astype(np.float32)]).astype(np.float32)
        jit = np.concatenate([np.random.RandomState(seed + 1).normal(0, 0.05, int(common.sum())),      # equal on twins
                              np.random.RandomState(seed + 2 + k).normal(0, 0.05, len(cbk))])
        # the loader's row order carries no meaning to the path: shuffle, so that twins do not sit at equal row numbers
        perm = np.random.RandomState(seed * 7 + k).permutation(len(c))
        clouds.append((x[perm], c[perm].astype(np.int32), (1.0 + jit[perm]).astype(np.float32)))
    out = {}
    for k, (x, c, f) in enumerate(clouds):
        out[f"pcd{k}"] = (torch.from_numpy(x),)
        out[f"sinput{k}_C"] = me_utils.batched_coordinates([c])
        out[f"sinput{k}_F"] = torch.from_numpy(f)[:, None].contiguous()
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = (sh * voxel_size).astype(np.float32)
    out["T_gt"] = torch.from_numpy(T)
    out["len_batch"] = [[len(clouds[0][0]), len(clouds[1][0])]]
    out["shared_voxels"] = int(common.sum())
    return out


def make_box_cloud(seed, n_points=5000, cube=10.0, n_boxes=8):
    """configs[0] plumbing cloud: points uniform on the surfaces of random boxes in a ``cube`` metre cube."""
    rng = np.random.RandomState(seed)
    pts = []
    per = n_points // n_boxes
    for b in range(n_boxes):
        c = rng.uniform(1.5, cube - 1.5, 3)
        s = rng.uniform(0.8, 3.0, 3)
        n = per if b < n_boxes - 1 else n_points - per * (n_boxes - 1)
        u = rng.uniform(-0.5, 0.5, (n, 3)) * s
        face = rng.randint(0, 3, n)
        sign = rng.choice([-0.5, 0.5], n)
        u[np.arange(n), face] = sign * s[face]
        pts.append(c + u)
    return np.concatenate(pts).astype(np.float32)
