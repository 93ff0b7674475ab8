"""Debug: gcl_sc2_confidence_sparse vs gcl_sc2_confidence after 1 and 20 products (partial sums and confidences must be
bitwise equal), and the kept entries against a torch rebuild of the matrix.  (An earlier, two-pass version of the build was
debugged with this: profiles/r05_conv_experiments.txt 52.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcl_amd import _lib
lib = _lib.load()
dev = "cuda:0"
for n in (1500, 8000):
    rng = np.random.RandomState(n)
    src = rng.uniform(-40, 40, (n, 3)).astype(np.float32); src[:, 2] *= 0.1
    ang = np.deg2rad(9.0)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tgt = (src @ R.T + np.array([3.0, 1.0, 0.1]) + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
    out = rng.rand(n) > 0.4
    tgt[out] = rng.uniform(-40, 40, (int(out.sum()), 3)).astype(np.float32)
    s, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    per = (n + 7) // 8
    for iters in (1, 20):
        res = []
        for sparse in (False, True):
            conf = torch.ones(n, device=dev)
            partial = torch.zeros(lib.gcl_sc2_chunks() * n, device=dev)
            done = torch.zeros(1, dtype=torch.int32, device=dev)
            if sparse:
                scratch = torch.zeros(lib.gcl_sc2_confidence_scratch_bytes(n), dtype=torch.uint8, device=dev)
                _lib.check(lib.gcl_sc2_confidence_sparse(_lib.ptr(s), _lib.ptr(t), n, 0.1, iters, _lib.ptr(partial), _lib.ptr(conf),
                                                         _lib.ptr(done), _lib.ptr(scratch), _lib.stream()), "sparse")
            else:
                _lib.check(lib.gcl_sc2_confidence(_lib.ptr(s), _lib.ptr(t), n, 0.1, iters, _lib.ptr(partial), _lib.ptr(conf),
                                                  _lib.ptr(done), _lib.stream()), "dense")
            torch.cuda.synchronize()
            res.append((conf.clone(), partial.clone(), done.clone()))
        d = (res[0][0] - res[1][0]).abs().max().item()
        dp = (res[0][1] - res[1][1]).abs().max().item()
        print(f"n={n} iters={iters}: max|conf diff| {d:.3e}  max|partial diff| {dp:.3e}  done {res[0][2].item()} {res[1][2].item()}", flush=True)
        if iters == 1:
            count = scratch[: 8 * n * 4].view(torch.int32)
            ds = (s[:, None, :] - s[None, :, :]).pow(2).sum(-1).sqrt(); dt = (t[:, None, :] - t[None, :, :]).pow(2).sum(-1).sqrt()
            M = torch.clamp(1.0 - (ds - dt).abs() ** 2 / (0.1 * 0.1), min=0.0)
            ref_count = torch.stack([(M[:, c * per:min(n, (c + 1) * per)] != 0).sum(1) for c in range(8)]).reshape(-1)
            print(f"   total nnz {int(count.sum())} ({int(count.sum()) / n / n:.4f} of n^2); count differs from a torch rebuild in "
                  f"{(ref_count.int() != count).sum().item()} of {8 * n} segments (borderline entries round differently in torch)")
