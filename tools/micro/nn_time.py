"""Time of gcl_nn_rowmin (lib/metrics.py::pdist_min) at the eval loop's shapes:  python tools/micro/nn_time.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcl_amd.lib.metrics import pdist_min
dev = torch.device("cuda:0")
SHAPES = [(5000, 5000, 32), (8000, 8000, 32), (5000, 35000, 32), (1024, 1024, 32), (5000, 5000, 16), (5000, 5000, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in sys.argv[1].split("x"))]
for ma, mb, c in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(0)
    A = torch.nn.functional.normalize(torch.randn(ma, c, generator=g), dim=1).to(dev)
    B = torch.nn.functional.normalize(torch.randn(mb, c, generator=g), dim=1).to(dev)
    for _ in range(3):
        pdist_min(A, B, "SquareL2")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        d, i = pdist_min(A, B, "SquareL2")
    e1.record(); torch.cuda.synchronize()
    ref = torch.cdist(A.double(), B.double()).argmin(1)
    print(f"{ma} x {mb} x {c}: {e0.elapsed_time(e1) * 50:.1f} us per call (alloc + interleave + search + merge); "
          f"index agreement with fp64 {(ref == i.long()).float().mean().item():.5f}")
