"""DIAGNOSTIC: forward time of the C <= 64 layers under the table orderings -- global mask sort, mask sort inside
windows of the loader order, mask sort inside windows of the spatial (cloud, Morton cell) order + contiguous tile range
per XCD.  Usage on the GPU box:  python tools/micro/spatial_order_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import synthetic  # noqa: E402
from gcl_amd.MinkowskiEngine import core  # noqa: E402

batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
LAYERS = [(1, 32, 32, 1, False), (1, 64, 64, 1, False), (2, 64, 64, 1, False), (1, 32, 64, 2, False), (2, 128, 64, 2, True),
          (4, 128, 128, 1, False)]
MODES = [("global", 0, 0, 0), ("loader-win4096", 4096, 0, 0), ("spatial-win4096", 0, 8, 4096), ("spatial-win2048", 0, 8, 2048)]
res = {}
for name, sw, smax, swin in MODES:
    core.SORT_WINDOW, core.SPATIAL_MAX_STRIDE, core.SPATIAL_WINDOW, core.SPATIAL_MIN_ROWS = sw, smax, max(swin, 2048), 0
    mgr = ME.CoordinateManager(C)
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
        res.setdefault((t, cin, cout, stride, tr), {})[name] = (e0.elapsed_time(e1) / 10 * 1e3, y)
for key, d in res.items():
    base = d["global"][1]
    line = " ".join(f"{m}={v[0]:7.1f}us{'' if torch.equal(v[1], base) else ' (DIFFERS!)'}" for m, v in d.items())
    print(f"t={key[0]} {key[1]:3d}->{key[2]:3d} s{key[3]}{' tr' if key[4] else ''}: {line}")
