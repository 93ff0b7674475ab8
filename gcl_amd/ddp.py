"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (backend "nccl") over xGMI.

The reference is single-GPU; this is the new capability BASELINE.json asks for.  The path shards
naturally: every rank takes its own samples, builds its own coordinate maps, mines negatives inside its own batch
(the reference's bs=4 semantics per rank) and keeps its own BatchNorm statistics (the reference has no SyncBN).
The only exchange is the gradient: all parameters are re-seated as views of ONE flat fp32 buffer (35 MB), all-reduced
in two large contiguous buckets (xGMI rings are per-link bound: few large collectives beat many small ones), followed by
a scale by 1/world_size.  Overlap: with the native plan (gcl_amd/MinkowskiEngine/native.py) the backward pass is enqueued
in two segments and ``bucket_ready`` starts the decoder bucket's all-reduce in between, so it runs under the encoder half
of the backward pass.  With the whole-network Tape (GCL_PLAN=0, or a model's first, recorded step) every gradient leaves
one autograd node at the END of the backward pass: the post-accumulate hooks then start both buckets back to back and
nothing overlaps; with GCL_TAPE=0 (one autograd node per layer) the hooks start the decoder bucket early again.
Parameters and BN buffers are broadcast from rank 0 once.
Works with any torch.distributed backend (gloo on CPU in tests/test_ddp_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > local:
        torch.cuda.set_device(local)     # every backend: the HIP kernels launch on the current device
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """Sample ids of this rank (strided, like DistributedSampler without shuffling)."""
    return list(range(rank, n_items, world))


def balance_global_batch(sizes, world):
    """Splits one global batch (``len(sizes)`` = world * per-rank batch size samples with voxel counts ``sizes``) into
    ``world`` groups of EQUAL sample count whose voxel sums are as even as a greedy assignment gets them (largest sample
    first, to the lightest rank that still has room).  Per-rank work is proportional to the voxel count and every step
    ends in an all-reduce, so the heaviest rank sets the step time (+-30 % per-rank spread with a plain
    DistributedSampler).  Deterministic (ties -> lowest rank), so every rank computes the same plan."""
    n = len(sizes)
    if n % world:
        raise ValueError(f"{n} samples do not split evenly over {world} ranks")
    per = n // world
    order = sorted(range(n), key=lambda i: (-int(sizes[i]), i))
    groups, load = [[] for _ in range(world)], [0] * world
    for i in order:
        r = min((r for r in range(world) if len(groups[r]) < per), key=lambda r: (load[r], r))
        groups[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(g) for g in groups]


def epoch_plan(sizes, world, batch_size, seed=0, balance=True):
    """Per-rank batches of one epoch: samples are shuffled (seeded: identical on every rank), cut into global batches of
    world * batch_size samples (a trailing partial one is dropped, like DistributedSampler(drop_last=True)), and every
    global batch is dealt to the ranks by ``balance_global_batch`` (or strided when ``balance`` is False).
    Returns plan[rank] = list of batches, each a list of ``batch_size`` sample ids."""
    import random
    ids = list(range(len(sizes)))
    random.Random(seed).shuffle(ids)
    gb = world * batch_size
    plan = [[] for _ in range(world)]
    for b0 in range(0, len(ids) - gb + 1, gb):
        chunk = ids[b0:b0 + gb]
        groups = balance_global_batch([sizes[i] for i in chunk], world) if balance else \
            [list(range(r, gb, world)) for r in range(world)]
        for r in range(world):
            plan[r].append([chunk[j] for j in groups[r]])
    return plan


class FlatDDP:
    """Flat-buffer gradient all-reduce for a module (no autograd hooks, no per-parameter collectives)."""

    def __init__(self, process_group=None, overlap=True, n_buckets=2):
        """``overlap``: split the flat gradient into ``n_buckets`` contiguous ranges and start the all-reduce of a
        range as soon as the backward pass has produced all of its gradients (later-registered layers first), so the
        collective of the decoder half runs under the encoder half of the backward pass.  Needs exactly one
        ``backward()`` per ``all_reduce_gradients()``; use ``overlap=False`` when accumulating several."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat_param = self.flat_grad = None
        self.params = []
        self.overlap, self.n_buckets = bool(overlap), max(1, int(n_buckets))
        self._bounds, self._bucket_of, self._count, self._pending, self._works = [], {}, [], [], {}
        self._sync = True
        self.launch_order = []         # buckets started early by bucket_ready (diagnostic / tests)

    def attach(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameter")
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)
                p.grad = self.flat_grad[off:off + n].view_as(p)
                off += n
        self.params = params
        # buckets = contiguous ranges of (almost) equal size, cut at parameter boundaries
        cuts, acc, nb = [0], 0, self.n_buckets
        for i, p in enumerate(params):
            acc += p.numel()
            if len(cuts) < nb and acc >= total * len(cuts) / nb and i + 1 < len(params):
                cuts.append(i + 1)
        cuts.append(len(params))
        offs = [0]
        for p in params:
            offs.append(offs[-1] + p.numel())
        self._bounds = [(offs[a], offs[b]) for a, b in zip(cuts[:-1], cuts[1:])]
        self._count = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
        for bi, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            for p in params[a:b]:
                self._bucket_of[id(p)] = bi
        self._pending, self._works = list(self._count), {}
        if self.overlap and self.world > 1:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)
        from .MinkowskiEngine import ops
        ops.invalidate_amax()                    # parameter storage moved (and is about to be broadcast into)
        if self.world > 1:
            dist.broadcast(self.flat_param, src=0, group=self.group)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=self.group)
        return self

    def _launch(self, b):
        lo, hi = self._bounds[b]
        self._works[b] = dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def set_last_microstep(self, last):
        """Gradient accumulation: call with False before every backward pass except the last one of an optimizer step
        (the hooks then leave the buckets alone) and with True before the last (buckets start as they complete)."""
        self._sync = bool(last)

    def bucket_ready(self, b):
        """Every gradient of bucket ``b`` has been written by work already enqueued on the current stream (the
        whole-network plan calls this between its backward segments): start the bucket's all-reduce now, under the rest
        of the backward pass.  No-op outside the last micro-step of an accumulated optimizer step."""
        if not self._sync or self.world <= 1 or b in self._works:
            return
        self.launch_order.append(b)
        self._launch(b)

    def _on_grad(self, p):
        if not self._sync:
            return
        b = self._bucket_of[id(p)]
        if b in self._works:
            raise RuntimeError("FlatDDP(overlap=True) saw a second backward pass before all_reduce_gradients(); "
                               "call set_last_microstep(False) before the earlier passes of an accumulated step "
                               f"(bucket {b}, pending {self._pending}, started {sorted(self._works)})")
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def all_reduce_gradients(self):
        """Average gradients over ranks: ONE collective on the flat buffer (optimizer.zero_grad(set_to_none=False)
        keeps p.grad seated in it)."""
        for p in self.params:        # autograd may have replaced .grad if zero_grad(set_to_none=True) was used
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_grad.data_ptr() + self.flat_grad.numel() * self.flat_grad.element_size():
                raise RuntimeError("parameter gradient left the flat buffer: call optimizer.zero_grad(set_to_none=False)")
        if self.world > 1:
            if self.overlap:
                for b in range(len(self._bounds)):       # ranges whose parameters got no gradient in this pass
                    if b not in self._works:
                        self._launch(b)
                for b in sorted(self._works):
                    self._works[b].wait()
                self._pending, self._works = list(self._count), {}
                self._sync = True
                self.last_launch_order, self.launch_order = self.launch_order, []
            else:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.flat_grad.div_(self.world)
