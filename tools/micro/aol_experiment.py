"""EXPERIMENT 71 (round 6): "norm1 applied on load by the consumer convolution", C <= 64.
The consumer gathers the PRE-BatchNorm rows and forms relu(x * s + t) where the fp32 pieces leave its LDS tile
(k_conv_fwd_dma<NB, false, false, AOL = true>, switched by gcl_debug_apply_on_load) -- against today's two launches: the apply
pass (read x, write y) + the convolution on y.  Per layer shape of the benchmark batch: parity of the two forms, event time of
10 launches each; run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
  python3 tools/micro/aol_experiment.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import _lib, synthetic  # noqa: E402

batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
lib = _lib.load()
mgr = ME.CoordinateManager(C)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


for (t, cin, cout) in [(1, 32, 32), (1, 64, 64), (2, 64, 64)]:
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3).to(dev)
    n = mgr.num_rows(t)
    x = torch.randn(n, cin, device=dev)
    s = (torch.rand(cin, device=dev) + 0.5).contiguous()
    sh = (0.2 * torch.randn(cin, device=dev)).contiguous()
    key = ME.CoordinateMapKey(t)
    with torch.no_grad():
        def apply_pass():
            return torch.relu(x * s + sh)           # one read + one write of [n, cin]: what k_bn_apply moves
        def conv_on(v):
            return conv(ME.SparseTensor(v, coordinate_map_key=key, coordinate_manager=mgr)).F
        us_apply, y = timed(apply_pass)
        us_conv, ref = timed(lambda: conv_on(y))
        _lib.check(lib.gcl_debug_apply_on_load(_lib.ptr(s), _lib.ptr(sh)), "aol on")
        try:
            us_aol, got = timed(lambda: conv_on(x))
        finally:
            _lib.check(lib.gcl_debug_apply_on_load(None, None), "aol off")
        err = float((got - ref).norm() / ref.norm())
    print(f"t={t} {cin}->{cout} n={n}: apply pass {us_apply:6.1f} us + conv {us_conv:6.1f} us = {us_apply + us_conv:6.1f} us;  "
          f"conv with apply-on-load {us_aol:6.1f} us  (delta {us_aol - us_conv:+6.1f} us vs the {us_apply:5.1f} us pass it replaces); "
          f"rel-L2 of the two results {err:.2e}")
