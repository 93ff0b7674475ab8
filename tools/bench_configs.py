"""Secondary measurements (BASELINE configs[1] and configs[4]); the headline line is bench.py.

configs[1]: KITTI-shaped pair, voxel 0.3 m, ResUNetBN2C-32 forward-only feature extraction (eval mode, batch 1 each,
            as scripts/test_kitti.py:141-152) -> active voxels/s.
configs[4]: LoKITTI-shaped eval: 2 x forward + find_corr (5000 x 5000 feature 1-NN, scripts/test_kitti.py:154) ->
            pairs/s and voxels/s, without registration and with the SC2-PCR back-end (gcl_amd.scripts.SC2_PCR,
            config_KITTI.json: 5000 sampled voxels per cloud, 8000 nodes) -- the README's "7 FPS on an RTX 3090"
            configuration (README.md:193).  The SC2-PCR stage alone is also timed on the CPU oracle.
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
        # scripts/test_kitti.py:158-180 with use_RANSAC false: random 5000 voxels per cloud -> Matcher.estimator
        from gcl_amd.scripts.SC2_PCR import Matcher
        matcher = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20,
                          ratio=0.2, nms_radius=0.6, max_points=8000, k1=30, k2=20)

        def register(dp, F0, F1):
            x0, x1 = dp["xyz0_dev"], dp["xyz1_dev"]
            s0 = torch.from_numpy(np.random.choice(len(F0), min(5000, len(F0)), replace=False)).to(dev)
            s1 = torch.from_numpy(np.random.choice(len(F1), min(5000, len(F1)), replace=False)).to(dev)
            return matcher.estimator(x0[s0][None], x1[s1][None], F0[s0][None], F1[s1][None])[0][0]

        for dp in dpairs:
            dp["xyz0_dev"], dp["xyz1_dev"] = dp["pcd0"][0].to(dev), dp["pcd1"][0].to(dev)
            T = register(dp, forward(dp, 0), forward(dp, 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        errs = []
        for _ in range(args.iters):
            for p, dp in zip(pairs, dpairs):
                F0, F1 = forward(dp, 0), forward(dp, 1)
                find_corr(p["pcd0"][0].numpy(), p["pcd1"][0].numpy(), F0, F1, subsample_size=5000)
                T = register(dp, F0, F1).cpu()                               # the eval loop reads T on the host
        torch.cuda.synchronize()
        t_reg = time.perf_counter() - t0
        # the same eval step with both clouds in ONE forward pass (gcl_amd.lib.eval.forward_pair: bitwise equal features)
        from gcl_amd.lib.eval import forward_pair
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            for p, dp in zip(pairs, dpairs):
                F0, F1 = forward_pair(model, dp["sinput0_F"], dp["sinput0_C"], dp["sinput1_F"], dp["sinput1_C"])
                find_corr(p["pcd0"][0].numpy(), p["pcd1"][0].numpy(), F0, F1, subsample_size=5000)
                T = register(dp, F0, F1).cpu()
        torch.cuda.synchronize()
        t_reg_pair = time.perf_counter() - t0
        # the registration stage alone, device vs the CPU oracle on the same correspondences
        dp = dpairs[0]
        F0, F1 = forward(dp, 0), forward(dp, 1)
        np.random.seed(1)
        s0 = torch.from_numpy(np.random.choice(len(F0), min(5000, len(F0)), replace=False)).to(dev)
        s1 = torch.from_numpy(np.random.choice(len(F1), min(5000, len(F1)), replace=False)).to(dev)
        sc, tc = matcher.match_pair(dp["xyz0_dev"][s0][None], dp["xyz1_dev"][s1][None], F0[s0][None], F1[s1][None])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            Tg = matcher.SC2_PCR(sc, tc)
        torch.cuda.synchronize()
        t_sc2_gpu = (time.perf_counter() - t0) / 10
        sys.path.insert(0, ROOT)
        from oracle.sc2pcr_oracle import sc2_pcr
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        t0 = time.perf_counter()
        To = sc2_pcr(sc[0].cpu().numpy(), tc[0].cpu().numpy())
        t_sc2_cpu = time.perf_counter() - t0
        sc2_diff = float((Tg[0].cpu() - To).abs().max())
        # ---- round 2: the packaged eval loop (gcl_amd.scripts.test_kitti.eval_pairs) with B pairs per forward pass
        from gcl_amd.scripts.test_kitti import eval_pairs, forward_clouds
        big = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
        dbig = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in p_.items()} for p_ in big]
        batched = {}
        for B in (1, 2, 4, 8):
            clouds = [(dp_[f"sinput{k}_F"], dp_[f"sinput{k}_C"]) for dp_ in dbig[:B] for k in (0, 1)]
            forward_clouds(model, clouds)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nv = 0
            for _ in range(args.iters):
                for c0 in range(0, 8, B):
                    cl = [(dp_[f"sinput{k}_F"], dp_[f"sinput{k}_C"]) for dp_ in dbig[c0:c0 + B] for k in (0, 1)]
                    nv += sum(len(f) for f, _ in cl)
                    forward_clouds(model, cl)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            batched[f"B={B} pairs ({2 * B} clouds) per forward"] = {
                "voxels_per_s": round(nv / dtb, 1), "ms_per_cloud": round(dtb / (args.iters * 16) * 1e3, 3)}
        np.random.seed(0)
        eval_pairs(model, big, matcher, device=dev, batch_pairs=8)                   # warm-up
        loop = {}
        for B in (1, 8):
            np.random.seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, args.iters // 3)):
                r = eval_pairs(model, big, matcher, device=dev, batch_pairs=B)
            torch.cuda.synchronize()
            dtl = (time.perf_counter() - t0) / max(1, args.iters // 3)
            loop[f"batch_pairs={B}"] = {"pairs_per_s": round(len(big) / dtl, 2), "ms_per_pair": round(dtl / len(big) * 1e3, 3),
                                        "voxels_per_s": round(r["n_voxels"] / dtl, 1)}
    n_pairs = args.iters * len(pairs)
    print(json.dumps({
        "configs[1] forward-only, B pairs per forward pass (forward_clouds: bitwise-equal features)": batched,
        "configs[4] packaged eval loop eval_pairs (2 x fwd + find_corr + random_sample + SC2-PCR + RTE/RRE meters)": loop,
        "configs[4] eval incl. SC2-PCR registration (8000 nodes)": {
            "pairs_per_s": round(n_pairs / t_reg, 2), "ms_per_pair": round(t_reg / n_pairs * 1e3, 3)},
        "configs[4] eval incl. SC2-PCR, both clouds in one forward pass (forward_pair)": {
            "pairs_per_s": round(n_pairs / t_reg_pair, 2), "ms_per_pair": round(t_reg_pair / n_pairs * 1e3, 3)},
        "SC2-PCR stage alone (8000 correspondences)": {
            "gpu_ms": round(t_sc2_gpu * 1e3, 3), "cpu_oracle_ms": round(t_sc2_cpu * 1e3, 1),
            "cpu_threads": torch.get_num_threads(), "max_abs_diff_T": sc2_diff},
        "configs[1] forward-only": {"voxels_per_s": round(nvox / t_fwd, 1), "ms_per_cloud": round(t_fwd / (2 * n_pairs) * 1e3, 3),
                                    "avg_voxels_per_cloud": round(nvox / (2 * n_pairs), 1)},
        "configs[4] eval (2x fwd + 5000x5000 feature 1-NN, no registration)": {
            "pairs_per_s": round(n_pairs / t_eval, 2), "ms_per_pair": round(t_eval / n_pairs * 1e3, 3),
            "voxels_per_s": round(nvox / t_eval, 1)},
        "n_gpus": 1, "dtype": "f32", "data": "synthetic"}))


if __name__ == "__main__":
    main()
