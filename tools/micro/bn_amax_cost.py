"""DIAGNOSTIC: does publishing max|y| from k_bn_apply / k_bn_bwd_apply cost time?  (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcl_amd import _lib
lib = _lib.load()
dev = "cuda:0"
for n, c in [(530321, 32), (530321, 64), (240825, 64), (100054, 128), (40574, 256)]:
    x = torch.randn(n, c, device=dev); res = torch.randn(n, c, device=dev); y = torch.empty_like(x)
    dy = torch.randn(n, c, device=dev); dx = torch.empty_like(x); dres = torch.empty_like(x)
    mean = torch.zeros(c, device=dev); rstd = torch.ones(c, device=dev); w = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    slot = torch.zeros(1, dtype=torch.int32, device=dev)
    out = []
    for with_amax in (False, True, False, True):
        for which in ("apply", "bwd_apply"):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            for rep in range(25):
                if rep == 5:
                    ev[0].record()
                s = _lib.ptr(slot) if with_amax else None
                if which == "apply":
                    lib.gcl_bn_apply(_lib.ptr(x), n, c, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), 1, _lib.ptr(y), s, _lib.stream())
                else:
                    lib.gcl_bn_bwd_apply(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), n, c, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(w), _lib.ptr(mean), _lib.ptr(mean), 1, _lib.ptr(dx), _lib.ptr(dres), s, _lib.stream())
            ev[1].record()
            torch.cuda.synchronize()
            out.append(f"{which}{'+amax' if with_amax else ''}={ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f}us")
    print(n, c, " ".join(out))
