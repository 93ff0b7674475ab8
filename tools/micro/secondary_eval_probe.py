import sys, json, time, torch, numpy as np
sys.path.insert(0, '.')
from gcl_amd import synthetic
from gcl_amd.model import load_model
from gcl_amd.scripts.SC2_PCR import Matcher
from gcl_amd.scripts.test_kitti import eval_pairs
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev)
model.eval()
pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
matcher = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                  nms_radius=0.6, max_points=8000, k1=30, k2=20)
for seed in (None, 0, None):
    if seed is not None:
        np.random.seed(seed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eval_pairs(model, pairs, matcher, device=dev, batch_pairs=8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"seed {seed}: {8/dt:.1f} pairs/s feat {r['feat_enqueue_time']*1e3:.1f} ms reg {r['reg_enqueue_time']*1e3:.1f} ms total {dt*1e3:.1f} ms success {r['success_rate']}")
