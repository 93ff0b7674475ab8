"""SGD with momentum / weight decay as ONE HIP launch per step (``gcl_sgd_multi``) behind torch.optim's interface.

The reference trains with ``torch.optim.SGD(lr, momentum, weight_decay)`` + ``ExponentialLR``
(lib/colocation_trainer.py:73-79, stepped at :887).  This subclass keeps that interface -- param_groups (so the
scheduler works), ``state[p]['momentum_buffer']`` (so optimizer state dicts interchange with torch's) -- and replaces the
seven multi-tensor launches and ~0.6 ms of host time of torch's foreach step by one launch over a pointer table.
Same arithmetic: d = g + wd p; buf = d (first step) or momentum buf + d; p -= lr buf (dampening 0, no Nesterov).
"""
import numpy as np
import torch

from gcl_amd import _lib


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False)
        self._sizes = {}          # per parameter list (identity of its tensors): device int64 element counts

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedSGD.step takes no closure")
        lib = _lib.require_gpu()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if any(p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() for p in ps):
                raise ValueError("FusedSGD: contiguous fp32 parameters and gradients only")
            dev = ps[0].device
            bufs, first = [], None
            for p in ps:
                st = self.state[p]
                b = st.get("momentum_buffer")
                if (b is None) != (first if first is not None else (b is None)):
                    raise ValueError("FusedSGD: parameters of a group must share their first step")
                first = b is None
                if b is None:
                    b = st["momentum_buffer"] = torch.empty_like(p)
                bufs.append(b)
            ptrs = np.empty((len(ps), 3), dtype=np.int64)
            for i, (p, b) in enumerate(zip(ps, bufs)):
                ptrs[i, 0], ptrs[i, 1], ptrs[i, 2] = p.data_ptr(), p.grad.data_ptr(), b.data_ptr()
            # element counts of THIS list of tensors (another param group, or a step in which a different subset has
            # gradients, is another list: keyed on the tensors' identities, never on the list length)
            key = (dev,) + tuple(id(p) for p in ps)
            sizes = self._sizes.get(key)
            if sizes is None:
                sizes = self._sizes[key] = torch.tensor([p.numel() for p in ps], dtype=torch.int64).to(dev)
            # gradients are fresh tensors every step: new pointers.  A FRESH pinned staging tensor per step -- the copy is
            # asynchronous and the host runs up to a step ahead of the GPU, so a re-used buffer would be overwritten
            # before the previous step's copy has executed (torch's pinned allocator recycles a block only after that)
            pin = torch.empty((len(ps), 3), dtype=torch.int64, pin_memory=True)
            pin.numpy()[:] = ptrs
            table = pin.to(dev, non_blocking=True)
            _lib.check(lib.gcl_sgd_multi(_lib.ptr(table), _lib.ptr(sizes), len(ps), float(group["lr"]),
                                         float(group["momentum"]), float(group["weight_decay"]), int(first),
                                         _lib.stream()), "gcl_sgd_multi")
            # the kernel wrote the parameters through raw pointers: bump their version counters like an in-place torch
            # op would, so that everything keyed on ``p._version`` (packed kernels / max|W| of the eval path, the
            # BatchNorm eval affine cache) sees the update
            bump = getattr(torch.autograd.graph, "increment_version", None)
            if bump is not None:
                for p in ps:
                    bump(p)
            else:
                from gcl_amd.MinkowskiEngine import ops
                ops.invalidate_amax()
        return None
