"""Debug: gcl_sc2_confidence_sparse vs gcl_sc2_confidence after 1 and 20 products; scratch contents vs a torch rebuild.
(This is how the count / fill predicate mismatch of the first version was found: profiles/r05_conv_experiments.txt 52.)"""
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
    for iters in (1, 20):
        res = []
        for sparse in (False, True):
            conf = torch.ones(n, device=dev)
            partial = torch.zeros(lib.gcl_sc2_chunks() * n, device=dev)
            done = torch.zeros(1, dtype=torch.int32, device=dev)
            if sparse:
                nb = lib.gcl_sc2_confidence_scratch_bytes(n)
                scratch = torch.zeros(nb, dtype=torch.uint8, device=dev)
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
            ints = scratch[: (16 * n + 64) * 4].view(torch.int32)
            count, offset, ovf = ints[:8 * n], ints[8 * n:16 * n], ints[16 * n]
            total = int(count.sum())
            print(f"   overflow {int(ovf)} total nnz {total} ({total / n / n:.4f} of n^2) segments disjoint {bool(((torch.sort(offset.long())[0][1:] - (torch.sort(offset.long())[0] + count.long()[torch.sort(offset.long())[1]])[:-1]) >= 0).all())}")
            # torch rebuild of the counts (fp32 arithmetic as the kernel's)
            ds = torch.cdist(s.double(), s.double()).float(); dt = torch.cdist(t.double(), t.double()).float()
            ds = (s[:, None, :] - s[None, :, :]).pow(2).sum(-1).sqrt(); dt = (t[:, None, :] - t[None, :, :]).pow(2).sum(-1).sqrt()
            cd = (ds - dt).abs()
            M = torch.clamp(1.0 - cd * cd / (0.1 * 0.1), min=0.0)
            per = (n + 7) // 8
            ref_count = torch.stack([(M[:, c * per:min(n, (c + 1) * per)] != 0).sum(1) for c in range(8)]).reshape(-1)
            print(f"   count mismatches vs torch {(ref_count.int() != count).sum().item()} of {8 * n}; dense partial vs torch {((M @ torch.ones(n, device=dev)) - res[0][1].view(8, n).sum(0)).abs().max().item():.3e}")
            ent = scratch[(16 * n + 64) * 4:(16 * n + 64) * 4 + total * 8].view(torch.int32).view(-1, 2)
            print(f"   entry column range {int(ent[:, 0].min())} .. {int(ent[:, 0].max())}")
            vals = ent[:, 1].contiguous().view(torch.float32)
            # (a workgroup's range of the entry buffer is reserved atomically: walk the segments in offset order)
            perm = torch.sort(offset.long())[1]
            segid = torch.repeat_interleave(perm, count.long()[perm])
            part_from_entries = torch.zeros(8 * n, device=dev, dtype=torch.float64).index_add_(0, segid, vals.double())
            print(f"   sum of entries per segment vs dense partial {(part_from_entries - res[0][1].double()).abs().max().item():.3e}; vs sparse partial {(part_from_entries - res[1][1].double()).abs().max().item():.3e}")
            rows = segid % n
            mv = M[rows, ent[:, 0].long()]
            print(f"   entry values vs torch M {(mv - vals).abs().max().item():.3e}; zero-valued entries {(vals == 0).sum().item()}")
            bad = (res[0][1] - res[1][1]).abs() > 1e-3
            idx = bad.nonzero().flatten()[:8].tolist()
            print("   bad segments (chunk,row,count,offset,dense,sparse):", [(k // n, k % n, int(count[k]), int(offset[k]), round(float(res[0][1][k]), 4), round(float(res[1][1][k]), 4)) for k in idx], "n bad", int(bad.sum()))
