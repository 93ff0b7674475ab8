"""DIAGNOSTIC (GPU box): achieved bytes/s of the BatchNorm passes at the training batch's layer shapes, against a plain
device-to-device copy of the same size (what this box's HBM gives a streaming kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcl_amd import _lib
lib = _lib.load()
dev = "cuda:0"
p, s = _lib.ptr, _lib.stream()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# rotate over several buffers so that the 256 MB Infinity Cache does not serve the reads
for n, c in ((530321, 32), (530321, 64), (240825, 64), (100054, 128), (40574, 256)):
    R = max(2, int(600e6 // (n * c * 4)) + 1)
    xs = [torch.randn(n, c, device=dev) for _ in range(R)]
    ys = [torch.empty(n, c, device=dev) for _ in range(R)]
    dys = [torch.randn(n, c, device=dev) for _ in range(R)]
    mean, rstd, w, b = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev), torch.zeros(c, device=dev)
    sg, sx = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    mask = torch.empty(lib.gcl_bn_mask_len(n, c), dtype=torch.int64, device=dev)
    scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=dev)
    slot = torch.zeros(64, dtype=torch.int32, device=dev)
    it = [0]

    def nxt():
        it[0] = (it[0] + 1) % R
        return it[0]
    mb = n * c * 4 / 1e6
    t_copy = timed(lambda: ys[nxt()].copy_(xs[it[0]]))
    t_app = timed(lambda: lib.gcl_bn_apply(p(xs[nxt()]), n, c, p(mean), p(rstd), p(w), p(b), None, 1, p(ys[it[0]]), p(mask), p(slot), s))
    t_app_r = timed(lambda: lib.gcl_bn_apply(p(xs[nxt()]), n, c, p(mean), p(rstd), p(w), p(b), p(dys[it[0]]), 1, p(ys[it[0]]), p(mask), p(slot), s))
    t_red = timed(lambda: lib.gcl_bn_bwd_reduce(p(xs[nxt()]), p(dys[it[0]]), None, p(mask), n, c, p(mean), p(rstd), 1, p(scratch), p(sg), p(sx), s))
    t_bwd = timed(lambda: lib.gcl_bn_bwd_apply(p(xs[nxt()]), p(dys[it[0]]), None, p(mask), n, c, p(mean), p(rstd), p(w), p(sg), p(sx), 1, p(ys[it[0]]), None, p(slot), s))
    print(f"[{n:6d} x {c:3d}] {mb:6.1f} MB  copy {t_copy:6.1f} us ({2 * mb / t_copy:5.2f} TB/s)  apply {t_app:6.1f} us ({2 * mb / t_app:5.2f})  "
          f"apply+res {t_app_r:6.1f} us ({3 * mb / t_app_r:5.2f})  bwd_reduce {t_red:6.1f} us ({2 * mb / t_red:5.2f})  bwd_apply {t_bwd:6.1f} us ({3 * mb / t_bwd:5.2f})")
