"""Secondary measurements (BASELINE configs[1] and configs[4]); the headline line is bench.py.

configs[1]: KITTI-shaped pair, voxel 0.3 m, ResUNetBN2C-32 forward-only feature extraction (eval mode, batch 1 each,
            as scripts/test_kitti.py:141-152) -> active voxels/s.
configs[4]: LoKITTI-shaped eval: 2 x forward + find_corr (5000 x 5000 feature 1-NN, scripts/test_kitti.py:154) ->
            pairs/s and voxels/s (registration excluded: open3d RANSAC / SC2-PCR are out of scope).
Writes one JSON object to stdout.  Usage: python tools/bench_configs.py [--pairs 3] [--iters 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.lib.eval import find_corr
    from gcl_amd.model import load_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
    model.eval()
    pairs = [synthetic.make_eval_pair(s, baseline=20.0 + 10.0 * s) for s in range(args.pairs)]
    dpairs = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in p.items()} for p in pairs]

    def forward(p, k):
        return model(ME.SparseTensor(p[f"sinput{k}_F"], coordinates=p[f"sinput{k}_C"])).F

    with torch.no_grad():
        for p in dpairs:                                   # warm-up
            forward(p, 0), forward(p, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nvox = 0
        for _ in range(args.iters):
            for p in dpairs:
                F0, F1 = forward(p, 0), forward(p, 1)
                nvox += len(F0) + len(F1)
        torch.cuda.synchronize()
        t_fwd = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.iters):
            for p, dp in zip(pairs, dpairs):
                F0, F1 = forward(dp, 0), forward(dp, 1)
                xyz0, xyz1 = p["pcd0"][0].numpy(), p["pcd1"][0].numpy()
                find_corr(xyz0, xyz1, F0, F1, subsample_size=5000)          # includes the D2H of 5000 indices
        torch.cuda.synchronize()
        t_eval = time.perf_counter() - t0
    n_pairs = args.iters * len(pairs)
    print(json.dumps({
        "configs[1] forward-only": {"voxels_per_s": round(nvox / t_fwd, 1), "ms_per_cloud": round(t_fwd / (2 * n_pairs) * 1e3, 3),
                                    "avg_voxels_per_cloud": round(nvox / (2 * n_pairs), 1)},
        "configs[4] eval (2x fwd + 5000x5000 feature 1-NN, no registration)": {
            "pairs_per_s": round(n_pairs / t_eval, 2), "ms_per_pair": round(t_eval / n_pairs * 1e3, 3),
            "voxels_per_s": round(nvox / t_eval, 1)},
        "n_gpus": 1, "dtype": "f32", "data": "synthetic"}))


if __name__ == "__main__":
    main()
