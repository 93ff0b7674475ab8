"""DIAGNOSTIC (round 4): forward time of the network's convolution shapes on the benchmark batch with the global mask sort
and with region-sorted tables (GCL_SORT_REGIONS=8: every XCD runs the tiles of one eighth of the natural row order),
alternating in ONE process; checks that y is bitwise the same.  LB_MODES="0,8" LB_ROUNDS=3 LB_ONLY=<t> select.
Usage on the GPU box:  python tools/micro/region_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import _lib, synthetic  # noqa: E402

batch = synthetic.make_train_batch(100, batch_size=int(os.environ.get("LB_BATCH", "4")), group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
LAYERS = [(1, 32, 32, 1, False), (1, 64, 64, 1, False), (2, 64, 64, 1, False), (1, 32, 64, 2, False), (2, 64, 128, 2, False),
          (4, 128, 128, 1, False), (4, 128, 256, 2, False), (8, 256, 256, 1, False), (8, 256, 128, 2, True),
          (4, 256, 64, 2, True), (2, 128, 64, 2, True)]
only = os.environ.get("LB_ONLY")
if only:
    LAYERS = [l for l in LAYERS if str(l[0]) in only.split(",")]
# mode 0: global sort, interleaved tiles; 8: region sort, one contiguous tile range per XCD; 80: region-sorted TABLES run in
# the interleaved tile order (separates what the sort costs from what the XCD assignment costs)
modes = [int(m) for m in os.environ.get("LB_MODES", "0,8,80").split(",")]
rounds = int(os.environ.get("LB_ROUNDS", "3"))
reps = int(os.environ.get("LB_REPS", "10"))
lib = _lib.load()
mgrs = {}
for m in modes:
    _lib.check(lib.gcl_set_sort_regions(8 if m else 0, int(os.environ.get("LB_MIN_ROWS", "8192"))), "gcl_set_sort_regions")
    mgrs[m] = ME.CoordinateManager(C)
    for (t, cin, cout, stride, tr) in LAYERS:          # build the sorted tables under this mode
        km = mgrs[m].get_kernel_map(t // 2 if tr else t, 3, stride)
        km.sorted_table(transposed=tr)
torch.cuda.synchronize()
tot = {m: 0.0 for m in modes}
for (t, cin, cout, stride, tr) in LAYERS:
    cls = ME.MinkowskiConvolutionTranspose if tr else ME.MinkowskiConvolution
    torch.manual_seed(0)
    conv = cls(cin, cout, kernel_size=3, stride=stride, dimension=3).to(dev)
    n = mgrs[modes[0]].num_rows(t)
    F = torch.randn(n, cin, device=dev)
    best, ys = {m: [] for m in modes}, {}
    for r in range(rounds):
        for m in modes:
            _lib.check(lib.gcl_set_sort_regions(8 if m == 8 else 0, 0), "gcl_set_sort_regions")
            x = ME.SparseTensor(F, coordinate_map_key=ME.CoordinateMapKey(t), coordinate_manager=mgrs[m])
            with torch.no_grad():
                y = conv(x).F
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    y = conv(x).F
                e1.record()
                torch.cuda.synchronize()
            best[m].append(e0.elapsed_time(e1) / reps * 1e3)
            ys[m] = y
    same = all(torch.equal(ys[modes[0]], ys[m]) for m in modes)
    pairs = mgrs[modes[0]].get_kernel_map(t // 2 if tr else t, 3, stride).n_pairs
    line = f"t={t} {cin:3d}->{cout:3d} s{stride}{' tr' if tr else '   '} n={n:7d} pairs={pairs:8d}:"
    for m in modes:
        med = sorted(best[m])[len(best[m]) // 2]
        tot[m] += med
        line += f"  regions={m}: {med:7.1f} us (min {min(best[m]):7.1f})"
    print(line + f"  bitwise_equal={same}", flush=True)
print("sum " + "  ".join(f"regions={m}: {tot[m]:.1f} us" for m in modes))
