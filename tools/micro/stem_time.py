"""DIAGNOSTIC: time of the first-layer kernels on the synthetic training batch, table path against occupancy path
(GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gcl_amd.MinkowskiEngine as ME
from gcl_amd import synthetic, _lib
lib = _lib.load()
batch = synthetic.make_train_batch(100, batch_size=4, group_mode="fixed16")
dev = "cuda:0"
C = batch["sinput_C"].to(dev)
mgr = ME.CoordinateManager(C)
km = mgr.get_kernel_map(1, 5, 1)
K, n = km.nbr.shape[0], len(C)
words = (K + 31) // 32
x = torch.ones(n, 1, device=dev)
W = torch.randn(K, 1, 32, device=dev)
dy = torch.randn(n, 32, device=dev)
y = torch.empty(n, 32, device=dev)
dw = torch.empty(K, 1, 32, device=dev)
bits = torch.empty(n * words, dtype=torch.int32, device=dev)
cloud = torch.zeros(4096, dtype=torch.int32, device=dev)
rows = torch.zeros(n, dtype=torch.int32, device=dev)
xb = batch["sinput_F"].to(dev).float().contiguous()      # the batch's own features: centre clouds jittered, the others ones
Ci = C.int().contiguous()
scratch = torch.empty(lib.gcl_stem_bwd_weight_scratch_len(K, 1, 32, n), dtype=torch.float32, device=dev)
s = _lib.stream()
p = _lib.ptr


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 20 * 1e3:7.0f} us   (n = {n}, K = {K})")


timed("presence_bits", lambda: lib.gcl_presence_bits(p(km.nbr), K, n, p(bits), s))
for label, feats in (("all ones", x), ("batch features", xb)):
    timed("not_ones_rows " + label, lambda: lib.gcl_not_ones_rows(p(feats), 1, p(Ci), n, p(cloud), 4096, p(rows), s))
    print(f"  rows on the table path: {int(rows.sum().item())} of {n}")
    for name, pb, pf in (("table", None, None), ("occupancy rows", p(bits), p(rows))):
        timed(f"fwd {name} ({label})", lambda: lib.gcl_stem_fwd(p(feats), p(W), p(km.nbr), n, K, 1, 32, p(y), pb, pf, s))
        timed(f"bwd_weight {name} ({label})",
              lambda: lib.gcl_stem_bwd_weight(p(feats), p(dy), p(km.nbr), n, K, 1, 32, p(scratch), p(dw), pb, pf, s))
