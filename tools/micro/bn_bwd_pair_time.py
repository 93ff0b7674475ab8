"""DIAGNOSTIC (GPU box): time of one BatchNorm backward (reduce pass, then apply pass over the SAME x / dy, as in a training
step) at the training batch's layer shapes.  Run with GCL_BN_REVERSE=0 / 1: the apply pass walking the tensors top-down
finds in the Infinity Cache what the reduce pass read last."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcl_amd import _lib
lib = _lib.load()
dev = "cuda:0"
p, s = _lib.ptr, _lib.stream()
for n, c in ((530321, 32), (530321, 64), (240825, 64), (100054, 128), (40574, 256)):
    R = 3                                      # a few tensor sets in rotation, like successive layers of a step
    xs = [torch.randn(n, c, device=dev) for _ in range(R)]
    dys = [torch.randn(n, c, device=dev) for _ in range(R)]
    dxs = [torch.empty(n, c, device=dev) for _ in range(R)]
    mean, rstd, w = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev)
    sg, sx = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    mask = torch.full((lib.gcl_bn_mask_len(n, c),), -1, dtype=torch.int64, device=dev)
    scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=dev)
    slot = torch.zeros(512, dtype=torch.int32, device=dev)

    def pair(i):
        lib.gcl_bn_bwd_reduce(p(xs[i]), p(dys[i]), None, p(mask), n, c, p(mean), p(rstd), 1, p(scratch), p(sg), p(sx), s)
        lib.gcl_bn_bwd_apply(p(xs[i]), p(dys[i]), None, p(mask), n, c, p(mean), p(rstd), p(w), p(sg), p(sx), 1, p(dxs[i]),
                             None, p(slot), s)
    for i in range(R):
        pair(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 30
    for r in range(reps):
        pair(r % R)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    mb = n * c * 4 / 1e6
    print(f"n = {n:7d} c = {c:3d}: reduce + apply {t:6.1f} us  ({5 * mb / t * 1e-3 * 1e3:.0f} GB/s of 5 tensor passes of {mb:.0f} MB)"
          f"  GCL_BN_REVERSE={os.environ.get('GCL_BN_REVERSE', '1')}")
