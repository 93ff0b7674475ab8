"""rocprofv3 target: 10 x Matcher.SC2_PCR on 8000 synthetic correspondences (config_KITTI.json sizes), inlier share argv[1]
(default 0.3; the outliers' targets are uniform in the scene).
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_sc2 -- python3 tools/sc2pcr_profile.py 0.3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gcl_amd.scripts.SC2_PCR import Matcher

rng = np.random.RandomState(0)
n = 8000
src = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
src[:, 2] *= 0.1
ang = np.deg2rad(15.0)
R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
tgt = (src @ R.T + np.array([5.0, 1.0, 0.3]) + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
out = rng.rand(n) >= share
tgt[out] = rng.uniform(-40, 40, (int(out.sum()), 3)).astype(np.float32)
m = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
            nms_radius=0.6, max_points=8000, k1=30, k2=20)
s, t = torch.from_numpy(src).cuda()[None], torch.from_numpy(tgt).cuda()[None]
T = m.SC2_PCR(s, t)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    T = m.SC2_PCR(s, t)
e1.record()
torch.cuda.synchronize()
print(f"inlier share {share}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per registration (events, 10 back to back)")
print(T[0].cpu().numpy().round(4))
