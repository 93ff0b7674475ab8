"""configs[4] on pairs with a controlled share of true correspondences (synthetic.make_twin_eval_pair): pairs/s of the packaged
eval loop, success rate and the MEASURED inlier share of the putative correspondences, per voxel share.
  python3 tools/micro/twin_eval_probe.py [share ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from gcl_amd import synthetic
from gcl_amd.model import load_model
from gcl_amd.scripts.SC2_PCR import Matcher
from gcl_amd.scripts.test_kitti import eval_pairs

dev = torch.device("cuda:0")
torch.manual_seed(0)
np.random.seed(0)
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
model.eval()
matcher = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                  nms_radius=0.6, max_points=8000, k1=30, k2=20)
shares = [float(a) for a in sys.argv[1:]] or [0.0, 0.3, 0.5, 0.7]
for share in shares:
    t0 = time.perf_counter()
    if share > 0:
        pairs = [synthetic.make_twin_eval_pair(200 + s, share) for s in range(8)]
    else:
        pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
    gen = time.perf_counter() - t0
    r = eval_pairs(model, pairs, matcher, device=dev, batch_pairs=8, collect=True)
    meas = float(np.mean([np.mean(np.asarray(d) < 0.6) for d in r["dists_nn"]]))
    labels = float(matcher.last["labels"].mean())
    for B in (8, 1):
        eval_pairs(model, pairs, matcher, device=dev, batch_pairs=B)
        torch.cuda.synchronize()
        rates = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(4):
                r = eval_pairs(model, pairs, matcher, device=dev, batch_pairs=B)
            torch.cuda.synchronize()
            rates.append(32 / (time.perf_counter() - t0))
        print(f"voxel share {share}: batch_pairs={B}: median {np.median(rates):.1f} pairs/s (min {min(rates):.1f}, max {max(rates):.1f}), "
              f"success {r['success_rate']:.2f}, rte {r['rte_avg']:.3f} m, measured inlier share of find_corr's 5000 matches "
              f"{meas:.3f}, of the last registration's 8000 correspondences {labels:.3f} (pairs generated in {gen:.1f} s)", flush=True)
