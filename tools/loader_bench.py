"""Loader-side throughput: one training sample (1 centre + 6 neighbour scans, ~115 k points
each) -> voxelise -> co-location groups.  CPU path = the numpy / cKDTree restatement used by the synthetic generator
(the reference runs an open3d KD-tree query per point in a Python loop, util/pointcloud.py:92-130, slower still);
GPU path = gcl_amd.lib.colocation_data_gpu.  Usage on the GPU box:  python tools/loader_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gcl_amd.MinkowskiEngine as ME
from gcl_amd import synthetic
from gcl_amd.lib.colocation_data_gpu import build_sample_gpu, collate_gpu

seed, voxel, nn = 7, 0.3, 6
scene = synthetic.make_scene(seed)
shifts = np.linspace(5.0, 60.0, nn)
clouds = [synthetic.raycast(scene, np.zeros(3), 1)]
Ms = []
for j, s in enumerate(shifts):
    pos = np.array([s, 0.1 * j, 0.0])
    clouds.append(synthetic.raycast(scene, pos, 2 + j))
    M = np.eye(4)
    M[:3, 3] = pos
    Ms.append(M)
pts = sum(len(c) for c in clouds)
radius = 1.5 * voxel

t0 = time.perf_counter()
th = []
for c in clouds:
    _, sel = ME.utils.sparse_quantize(c / voxel, return_index=True)
    th.append(c[sel])
t1 = time.perf_counter()
g, idx, fl = synthetic.colocation_groups(th[0], th[1:], Ms, radius)
t2 = time.perf_counter()
nvox = sum(len(x) for x in th)
print(f"sample: {pts} points -> {nvox} voxels, {len(g)} groups, {len(idx)} members")
print(f"CPU  voxelise {1e3 * (t1 - t0):7.1f} ms   groups {1e3 * (t2 - t1):7.1f} ms   -> {nvox / (t2 - t0) / 1e6:.3f} M voxels/s (1 core)")

dev = "cuda:0"
dclouds = [torch.from_numpy(c).to(dev) for c in clouds]
for _ in range(2):
    s = build_sample_gpu(dclouds, Ms, voxel, radius, dev)
torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    s = build_sample_gpu(dclouds, Ms, voxel, radius, dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
assert np.array_equal(s["group"].cpu().numpy(), np.asarray(g, dtype=np.int32))
assert np.array_equal(s["index"].cpu().numpy(), np.asarray(idx, dtype=np.int64))
print(f"GPU  voxelise + groups {1e3 * dt:7.2f} ms per sample (inputs resident)   -> {nvox / dt / 1e6:.2f} M voxels/s; "
      f"identical groups")
t0 = time.perf_counter()
for _ in range(reps):
    dcl = [torch.from_numpy(c).to(dev) for c in clouds]
    s = build_sample_gpu(dcl, Ms, voxel, radius, dev)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / reps
print(f"GPU  incl. H2D of the raw points ({pts * 12 / 1e6:.1f} MB): {1e3 * dt2:7.2f} ms per sample -> {nvox / dt2 / 1e6:.2f} M voxels/s")

# ---- round 6: the whole batch in one pass (build_batch_gpu): 4 samples x 7 scans, pinned staging + H2D included
from gcl_amd.lib.colocation_data_gpu import build_batch_gpu
raws = [synthetic.make_raw_sample(100 + b) for b in range(4)]
pts4 = sum(len(x) for r in raws for x in r["xyz"])
for _ in range(2):
    b = build_batch_gpu(raws, voxel, dev, jitter=synthetic.raw_sample_jitter(raws))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    b = build_batch_gpu(raws, voxel, dev, jitter=synthetic.raw_sample_jitter(raws))
torch.cuda.synchronize()
dt3 = (time.perf_counter() - t0) / reps
print(f"GPU  build_batch_gpu, 4 samples ({pts4} points, {pts4 * 12 / 1e6:.1f} MB H2D included) -> {len(b['sinput_C'])} voxels, "
      f"{len(b['group'])} groups: {1e3 * dt3:7.2f} ms per batch = {1e3 * dt3 / 4:.2f} ms per sample -> "
      f"{len(b['sinput_C']) / dt3 / 1e6:.2f} M voxels/s")
