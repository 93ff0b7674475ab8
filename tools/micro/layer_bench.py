"""DIAGNOSTIC: forward time of the network's convolution shapes on the benchmark batch (one line per layer shape).
Usage on the GPU box:  python tools/micro/layer_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import synthetic  # noqa: E402

batch = synthetic.make_train_batch(100, batch_size=int(os.environ.get("LB_BATCH", "4")), group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
LAYERS = [(1, 32, 32, 1, False), (1, 64, 64, 1, False), (2, 64, 64, 1, False), (1, 32, 64, 2, False), (2, 64, 128, 2, False),
          (4, 128, 128, 1, False), (4, 128, 256, 2, False), (8, 256, 256, 1, False), (8, 256, 128, 2, True),
          (4, 256, 64, 2, True), (2, 128, 64, 2, True)]
mgr = ME.CoordinateManager(C)
tot = 0.0
for (t, cin, cout, stride, tr) in LAYERS:
    cls = ME.MinkowskiConvolutionTranspose if tr else ME.MinkowskiConvolution
    torch.manual_seed(0)
    conv = cls(cin, cout, kernel_size=3, stride=stride, dimension=3).to(dev)
    n = mgr.num_rows(t)
    x = ME.SparseTensor(torch.randn(n, cin, device=dev), coordinate_map_key=ME.CoordinateMapKey(t), coordinate_manager=mgr)
    with torch.no_grad():
        y = conv(x).F
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = conv(x).F
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tot += us
    pairs = mgr.get_kernel_map(t // 2 if tr else t, 3, stride).n_pairs
    print(f"t={t} {cin:3d}->{cout:3d} s{stride}{' tr' if tr else '   '} n={n:7d} pairs={pairs:8d}: {us:7.1f} us  "
          f"{2e-6 * pairs * cin * cout / us:6.1f} TF/s fp32-equivalent (x3 MFMA terms), "
          f"{(pairs * cin + len(y) * cout) * 4e-6 / us:5.2f} TB/s gathered+written  checksum {float(y.double().sum()):.6e}")
print(f"sum {tot:.1f} us")
