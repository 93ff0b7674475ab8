"""Where a pair of the packaged eval loop (gcl_amd.scripts.test_kitti.eval_pairs, configs[4]) spends its time.
  python3 tools/micro/eval_tail_probe.py            wall time + cProfile of the enqueuing thread
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_eval -o r --output-format csv -- python3 tools/micro/eval_tail_probe.py noprof
"""
import cProfile
import os
import pstats
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
pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
matcher = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                  nms_radius=0.6, max_points=8000, k1=30, k2=20)
for B in (8, 1):
    eval_pairs(model, pairs, matcher, device=dev, batch_pairs=B)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r = eval_pairs(model, pairs, matcher, device=dev, batch_pairs=B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"batch_pairs={B}: {8 / dt:.1f} pairs/s, {dt / 8 * 1e3:.2f} ms per pair (enqueue: features {r['feat_enqueue_time'] / 8 * 1e3:.2f} ms, "
          f"registration {r['reg_enqueue_time'] / 8 * 1e3:.2f} ms per pair), success {r['success_rate']}")
if len(sys.argv) > 1 and sys.argv[1] == "noprof":
    sys.exit(0)
PB = int(sys.argv[2]) if len(sys.argv) > 2 else 8          # python3 eval_tail_probe.py prof 1: the profile at batch_pairs = 1
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eval_pairs(model, pairs, matcher, device=dev, batch_pairs=PB)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
if PB != 8:
    st.sort_stats("cumtime").print_stats(18)
