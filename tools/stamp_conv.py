"""DIAGNOSTIC (not product, not test): where does a k_conv_fwd_split wave spend its cycles?

Builds libgcl_hip_stamps.so with -DGCL_STAMPS (s_memtime stamps around the two barriers of every step), runs the
forward convolutions of one synthetic batch layer by layer and prints the per-phase cycle shares.  The stamped build is
slower than the product build: read the SHARES, not the times.  Usage on the GPU box:  python tools/stamp_conv.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gcl_amd", "csrc")
LIB = os.path.join("/tmp", "libgcl_hip_stamps.so")
os.makedirs(os.path.dirname(LIB), exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGCL_STAMPS", "-o", LIB] +
               [os.path.join(CSRC, f) for f in ("coords.hip", "conv.hip", "norm.hip", "loss.hip", "data.hip", "sc2pcr.hip", "plan.hip")], check=True)
os.environ["GCL_LIB_PATH"] = LIB

import torch  # noqa: E402
import gcl_amd.MinkowskiEngine as ME  # noqa: E402
from gcl_amd import _lib, synthetic  # noqa: E402

lib = _lib.load()
lib.gcl_debug_stamps.restype = ctypes.c_int
lib.gcl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
mgr = ME.CoordinateManager(C)
names = ["lds_read+mfma", "gather_wait+lds_write", "load_issue", "barrier", "prologue", "-", "wave_steps", "mine_steps"]
for (t, cin, cout, stride, tr) in [(1, 32, 32, 1, False), (1, 64, 64, 1, False), (2, 64, 64, 1, False),
                                   (4, 128, 128, 1, False), (8, 256, 256, 1, False), (4, 128, 256, 2, False),
                                   (8, 256, 128, 2, True)]:
    cls = ME.MinkowskiConvolutionTranspose if tr else ME.MinkowskiConvolution
    conv = cls(cin, cout, kernel_size=3, stride=stride, dimension=3).to(dev)
    n = mgr.num_rows(t)
    x = ME.SparseTensor(torch.randn(n, cin, device=dev), coordinate_map_key=ME.CoordinateMapKey(t), coordinate_manager=mgr)
    with torch.no_grad():
        conv(x)                                           # builds the maps
        torch.cuda.synchronize()
        lib.gcl_debug_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv(x)
        e1.record()
        torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.gcl_debug_stamps(buf, 0)
    v = list(buf)
    tot = sum(v[:4])
    steps = max(v[6], 1)
    print(f"t={t} {cin}->{cout} s{stride}{' tr' if tr else ''} n={n}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us/launch (stamped build) | "
          + " ".join(f"{names[q]}={v[q] / tot * 100:.0f}%({v[q] / steps:.0f}cyc/step)" for q in range(4))
          + f" | mine {v[7] / steps * 100:.0f}% of {steps / 5:.0f} wave-steps/launch")
