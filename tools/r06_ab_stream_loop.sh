#!/bin/bash
# same-box A/B (tools/ab_prev.sh export <commit> first): forward_clouds_stream loop rate, old tree vs working tree
R=${GRAFT_REPO_ROOT:-$PWD}
cat > /tmp/stream_probe.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gcl_amd import synthetic
from gcl_amd.model import load_model
from gcl_amd.scripts.test_kitti import forward_clouds_stream
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(0)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev).eval()
pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
d = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in p.items()} for p in pairs]
nv = sum(len(x["sinput0_C"]) + len(x["sinput1_C"]) for x in d) * 4
with torch.no_grad(), torch.cuda.device(dev):
    def run():
        for _ in forward_clouds_stream(model, ([(x[f"sinput{k}_F"], x[f"sinput{k}_C"]) for k in (0, 1)] for x in d * 4), device=dev):
            pass
    run(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            run(); n += 1; torch.cuda.synchronize()
        print(f"  loop over pairs: {nv * n / (time.perf_counter() - t0) / 1e6:.1f} M voxels/s")
PY
for i in 1 2; do
  echo "== prev"; (cd $R/.ab_prev && python3 /tmp/stream_probe.py 2>&1 | grep "M voxels")
  echo "== new";  (cd $R && python3 /tmp/stream_probe.py 2>&1 | grep "M voxels")
done
