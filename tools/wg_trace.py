"""DIAGNOSTIC (GPU box): when do the workgroups of one k_conv_fwd_split launch start and end?  Stamped build
(-DGCL_STAMPS, as tools/stamp_conv.py); prints, per layer, the launch span, the throughput bound (sum of workgroup
durations / slots in use) and the longest workgroups -- i.e. whether the launch ends on its heaviest tiles."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gcl_amd", "csrc")
LIB = os.path.join("/tmp", "libgcl_hip_stamps.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGCL_STAMPS", "-o", LIB] +
               [os.path.join(CSRC, f) for f in ("coords.hip", "conv.hip", "norm.hip", "loss.hip", "data.hip", "sc2pcr.hip", "plan.hip")], check=True)
os.environ["GCL_LIB_PATH"] = LIB

import numpy as np  # noqa: E402
import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import _lib, synthetic  # noqa: E402

lib = _lib.load()
lib.gcl_debug_wgtrace.restype = ctypes.c_int
lib.gcl_debug_wgtrace.argtypes = [ctypes.c_void_p, ctypes.c_int]
batch = synthetic.make_train_batch(100, batch_size=int(os.environ.get("LB_BATCH", "4")), group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
mgr = ME.CoordinateManager(C)
N = 16384
for (t, cin, cout, stride, tr) in [(8, 256, 256, 1, False), (4, 128, 128, 1, False), (2, 64, 64, 1, False), (4, 128, 256, 2, False)]:
    cls = ME.MinkowskiConvolutionTranspose if tr else ME.MinkowskiConvolution
    conv = cls(cin, cout, kernel_size=3, stride=stride, dimension=3).to(dev)
    n = mgr.num_rows(t)
    x = ME.SparseTensor(torch.randn(n, cin, device=dev), coordinate_map_key=ME.CoordinateMapKey(t), coordinate_manager=mgr)
    with torch.no_grad():
        for _ in range(3):
            conv(x)
        torch.cuda.synchronize()
    buf = np.zeros(N * 4, np.uint64)
    lib.gcl_debug_wgtrace(buf.ctypes.data_as(ctypes.c_void_p), N)
    tr_ = buf.reshape(N, 4).astype(np.int64)
    last = tr_[:, 1].max()
    sel = tr_[(tr_[:, 1] > 0) & (tr_[:, 0] > last - 200000) & (tr_[:, 2] > 0)]      # this launch: within 2 ms of its end
    t0 = sel[:, 0].min()
    start, end, steps = (sel[:, 0] - t0) / 100.0, (sel[:, 1] - t0) / 100.0, sel[:, 2]      # microseconds
    dur = end - start
    span = end.max()
    order = np.argsort(-dur)
    print(f"t={t} {cin}->{cout} s{stride}: {len(sel)} working workgroups, launch span {span:.0f} us (stamped build), "
          f"sum of durations / 1024 slots = {dur.sum() / 1024:.0f} us, sum of steps {steps.sum()}, "
          f"mean us per step {dur.sum() / steps.sum():.2f}")
    print("   longest workgroups (start, end, steps, us/step): " +
          "  ".join(f"({start[i]:.0f}, {end[i]:.0f}, {steps[i]}, {dur[i] / steps[i]:.2f})" for i in order[:6]))
    late = np.argsort(-end)[:6]
    print("   last to finish            (start, end, steps, us/step): " +
          "  ".join(f"({start[i]:.0f}, {end[i]:.0f}, {steps[i]}, {dur[i] / steps[i]:.2f})" for i in late))
    ticks = sel[:, 3].astype(np.float64)
    ok = dur > 20
    print(f"   s_memtime ticks per microsecond of s_memrealtime over the workgroups' lifetimes: median "
          f"{np.median(ticks[ok] / dur[ok]):.1f} (p10 {np.percentile(ticks[ok] / dur[ok], 10):.1f}, p90 {np.percentile(ticks[ok] / dur[ok], 90):.1f})")
    q = np.percentile(end, [50, 90, 99])
    print(f"   workgroup end times p50 / p90 / p99 / max: {q[0]:.0f} / {q[1]:.0f} / {q[2]:.0f} / {span:.0f} us; "
          f"started after t = 10 us: {(start > 10).sum()}")
