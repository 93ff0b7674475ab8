"""CPU ORACLE (test infrastructure, NOT product code) -- the co-location group builder of the loader.

Restates ``get_matching_indices_colocation`` (/root/reference/util/pointcloud.py:69-132) the way it is written there:
a loop over the centre cloud's points, one radius search per cloud, ``[:K]`` nearest hits, the strict-``<`` update of
the finest member, groups without any neighbour-cloud hit skipped.  The reference's searches go through
``open3d.geometry.KDTreeFlann.search_radius_vector_3d`` (nanoflann: squared distance < radius^2, results sorted by
distance).  open3d is not installable here and the reference holds no fixture for this function, so this restatement is
**parity unpinned** against open3d itself (tie order among equidistant points and the exact floating-point form of the
distance are the open points; here: float64 squared distances, ties -> lowest row).  Brute force, O(Nc * N): for the
small samples of the tests only.
"""
import numpy as np


class _Cloud:
    """Radius search by brute force inside the slab |x - qx| < r (rows pre-sorted by x; candidates are put back into
    ascending row order before the distance sort, so ties resolve to the lowest row)."""

    def __init__(self, pts64):
        self.all = pts64
        self.order = np.argsort(pts64[:, 0], kind="stable")
        self.xs = pts64[self.order, 0]

    def hits(self, query, r, K):
        lo, hi = np.searchsorted(self.xs, query[0] - r, "left"), np.searchsorted(self.xs, query[0] + r, "right")
        if hi <= lo:
            return []
        rows = np.sort(self.order[lo:hi])
        d2 = ((self.all[rows] - query) ** 2).sum(1)
        keep = np.nonzero(d2 < r * r)[0]
        if len(keep) == 0:
            return []
        keep = keep[np.argsort(d2[keep], kind="stable")]
        out = rows[keep]
        return [int(v) for v in (out[:K] if K is not None else out)]


def _make_cloud(pts64):
    return _Cloud(pts64)


def colocation_groups(center_xyz, nghb_xyz, list_trans, search_radius, K=5, nghb_cf=None):
    """(group, index, finest_flag) as the reference returns them (lists).  ``center_xyz`` [Nc, 3] and ``nghb_xyz[j]``
    [N_j, 3] are sensor-centred clouds; ``list_trans[j]`` maps neighbour j into the centre frame (:87-89).
    ``nghb_cf`` (optional): the neighbour clouds already in the centre frame -- the device builder receives them as
    float32 (rounded after the transform), the reference transforms in float64; passing the float32 points here
    compares the two builders on identical inputs."""
    center64 = np.asarray(center_xyz, dtype=np.float64)
    nghb_own = [np.asarray(x, dtype=np.float64) for x in nghb_xyz]
    if nghb_cf is None:
        # pcd.transform(T) on float64 points (open3d stores points as doubles); inputs are float32 voxel representatives
        nghb_cf = [x @ np.asarray(T, dtype=np.float64)[:3, :3].T + np.asarray(T, dtype=np.float64)[:3, 3]
                   for x, T in zip(nghb_own, list_trans)]
    else:
        nghb_cf = [np.asarray(x, dtype=np.float64) for x in nghb_cf]
    r = float(search_radius)
    center_tree = _make_cloud(center64)
    nghb_trees = [_make_cloud(cf) for cf in nghb_cf]
    group, index, finest = [], [], []
    for i, point in enumerate(center64):
        closest = np.linalg.norm(point)                                     # :95
        members = center_tree.hits(point, r, K)                             # :96-99
        n_own = len(members)
        finest_position = 0
        start = len(center64)
        for j, tree in enumerate(nghb_trees):                               # :106-118
            idx = tree.hits(point, r, K)
            if idx:
                dist = np.linalg.norm(nghb_own[j][idx[0]])                 # the hit's range from ITS OWN sensor (:111)
                if dist < closest:
                    closest = dist
                    finest_position = len(members)
                members += [int(v) + start for v in idx]
            start += len(nghb_own[j])
        if len(members) == n_own:                                           # :119-120
            continue
        group.append(len(members))
        index += [int(v) for v in members]
        flags = [False] * len(members)
        flags[finest_position] = True
        finest += flags
    return group, index, finest
