"""DIAGNOSTIC: time of the first-layer kernels on the synthetic training batch (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gcl_amd.MinkowskiEngine as ME
from gcl_amd import synthetic
batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
conv = ME.MinkowskiConvolution(1, 32, kernel_size=5, stride=1, dimension=3).to(dev)
x = ME.SparseTensor(torch.ones(len(C), 1, device=dev), coordinates=C)
y = conv(x)
g = torch.randn_like(y.F)
for name, fn in (("fwd", lambda: conv(x)), ("fwd+bwd", lambda: conv(x).F.backward(g))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
