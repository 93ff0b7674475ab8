"""GCL training step: sparse U-Net forward, group-wise contrastive loss, backward, SGD
(interface of lib/colocation_trainer.py: AlignmentTrainer :29-174, FinestContrastiveLossTrainer :403-535,
_train_epoch :811-916).

What differs from the reference, by design:
* the loss never loops over groups in Python and never returns to the host: one wave per selected group for the
  positive / finest terms, one fused pdist+row-min kernel and an on-device positive-pair mask for the negatives
  (csrc/loss.hip).  The three ``np.random.choice`` draws are made on the host in the reference's order (:457,
  :506-507), so a seeded run selects the same groups / rows.
* ``index_hash`` is accepted for signature parity but not needed: a pair is masked iff both rows share a positive
  group, which is what membership of its (collision-free) symmetric key in ``index_hash`` means (:521-529).
* data-parallel training (absent from the reference) shards samples over ranks and all-reduces one flat gradient
  buffer (gcl_amd/ddp.py).
"""
import os
import types

import numpy as np
import torch
from torch.autograd.function import once_differentiable

import gcl_amd.MinkowskiEngine as ME
from gcl_amd import _lib
from gcl_amd.model import load_model

# defaults = config.py of the reference overridden by scripts/train_gcl_kitti.sh
DEFAULT_CONFIG = dict(
    model="ResUNetBN2C", model_n_out=32, conv1_kernel_size=5, normalize_feature=True, bn_momentum=0.05,
    optimizer="SGD", lr=0.1, momentum=0.8, weight_decay=1e-4, exp_gamma=0.99,
    batch_size=4, num_pos_per_batch=256, num_hn_samples_per_batch=256, iter_size=1,
    hit_ratio_thresh=0.3, val_max_iter=400, nn_max_n=500,
    pos_thresh=0.1, neg_thresh=1.4, finest_thresh=0.2, pos_weight=1.0, neg_weight=1.0, finest_weight=1.0,
    square_loss=True, block_finest_gradient=False, use_hard_negative=True, use_pair_group_positive_loss=False,
    use_group_circle_loss=False, safe_radius=0.75, voxel_size=0.3,
)


def make_config(**overrides):
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(overrides)
    return types.SimpleNamespace(**cfg)


class _GCLLossFn(torch.autograd.Function):
    """(pos[s], fin[s], neg) of finest_contrastive_loss for fixed selections; one gradient buffer for all terms."""

    @staticmethod
    def forward(ctx, F, index, goff, flag, sel, sel1, sel2, pos_thresh, fin_thresh, neg_thresh, flags=0,
                pairpos=None):
        lib = _lib.require_gpu()
        F = F.contiguous()
        n, c = F.shape
        dev = F.device
        n_sel, m = sel.shape[0], sel1.shape[0]
        pos = torch.empty(n_sel, dtype=torch.float32, device=dev)
        fin = torch.empty(n_sel, dtype=torch.float32, device=dev)
        st = _lib.stream()
        _lib.check(lib.gcl_group_loss_fwd(_lib.ptr(F, torch.float32), c, _lib.ptr(index, torch.int64),
                                          _lib.ptr(goff, torch.int64), _lib.ptr(flag, torch.uint8),
                                          _lib.ptr(sel, torch.int64), n_sel, pos_thresh, fin_thresh, int(flags),
                                          _lib.ptr(pairpos, torch.int32), _lib.ptr(pos), _lib.ptr(fin), st),
                   "gcl_group_loss_fwd")
        dmin = torch.empty(m, dtype=torch.float32, device=dev)
        arg = torch.empty(m, dtype=torch.int32, device=dev)     # hardest negative: fused pdist + row minimum (:510-512)
        ns = lib.gcl_nn_rowmin_scratch_len(m, m)
        nn_scratch = torch.empty(ns, dtype=torch.int32, device=dev) if ns else None
        _lib.check(lib.gcl_nn_rowmin(_lib.ptr(F), _lib.ptr(sel1, torch.int64), m, _lib.ptr(F),
                                     _lib.ptr(sel2, torch.int64), m, c, 1, _lib.ptr(nn_scratch), _lib.ptr(dmin),
                                     _lib.ptr(arg), st), "gcl_nn_rowmin")
        cap = 64
        while cap < 2 * m:
            cap *= 2
        table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        keep = torch.empty(m, dtype=torch.uint8, device=dev)
        _lib.check(lib.gcl_neg_mask(_lib.ptr(sel1), _lib.ptr(sel2), _lib.ptr(arg), m, _lib.ptr(index), _lib.ptr(goff),
                                    goff.shape[0] - 1, index.shape[0], _lib.ptr(table), cap, _lib.ptr(keep), st),
                   "gcl_neg_mask")
        out = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(lib.gcl_neg_loss_fwd(_lib.ptr(dmin), _lib.ptr(keep), m, neg_thresh, _lib.ptr(out), st),
                   "gcl_neg_loss_fwd")
        ctx.save_for_backward(F, index, goff, flag, sel, sel1, sel2, dmin, arg, keep, out, pairpos)
        ctx.th = (pos_thresh, fin_thresh, neg_thresh, int(flags))
        return pos, fin, out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gpos, gfin, gneg):
        return (_GCLLossFn.backward_terms(ctx, gpos, gfin, gneg),) + (None,) * 11

    @staticmethod
    def backward_terms(ctx, gpos, gfin, gneg):
        lib = _lib.load()
        F, index, goff, flag, sel, sel1, sel2, dmin, arg, keep, out, pairpos = ctx.saved_tensors
        pos_thresh, fin_thresh, neg_thresh, flags = ctx.th
        n, c = F.shape
        st = _lib.stream()
        dF = torch.zeros_like(F)
        # the contiguous copies are NAMED: a temporary handed to _lib.ptr dies with that call and the caching allocator
        # gave its block to the next temporary -- gpos and gfin (expanded tensors out of sum()'s backward) then shared one
        # address and the kernel read gfin for both, unnoticed while pos_weight == finest_weight (every test, the defaults)
        gpos_c, gfin_c = gpos.contiguous(), gfin.contiguous()
        _lib.check(lib.gcl_group_loss_bwd(_lib.ptr(F), c, _lib.ptr(index), _lib.ptr(goff), _lib.ptr(flag),
                                          _lib.ptr(sel), sel.shape[0], pos_thresh, fin_thresh, flags,
                                          _lib.ptr(pairpos), _lib.ptr(gpos_c), _lib.ptr(gfin_c), _lib.ptr(dF), st),
                   "gcl_group_loss_bwd")
        g = gneg.reshape(1).contiguous()
        _lib.check(lib.gcl_neg_loss_bwd(_lib.ptr(F), c, _lib.ptr(sel1), _lib.ptr(sel2), _lib.ptr(arg), _lib.ptr(dmin),
                                        _lib.ptr(keep), sel1.shape[0], neg_thresh, _lib.ptr(out), _lib.ptr(g),
                                        _lib.ptr(dF), st), "gcl_neg_loss_bwd")
        return dF


class _GCLTotalLossFn(torch.autograd.Function):
    """(total, pos_mean, fin_mean, neg) with total = w_pos * pos.sum() / n + w_fin * fin.sum() / n + w_neg * neg
    (lib/colocation_trainer.py:533-535, :865-868) as ONE autograd node: the two sums, two divisions and the weighted
    sum are one launch (gcl_loss_combine), their backward another (gcl_loss_seed) -- the same values, without the ~20
    one-element torch kernels (and their launch gaps) that sat between the network's forward and backward pass of every
    training step.  Only ``total`` carries a gradient; the three terms are returned for the meters."""

    @staticmethod
    def forward(ctx, F, index, goff, flag, sel, sel1, sel2, pos_thresh, fin_thresh, neg_thresh, flags, pairpos,
                w_pos, w_fin, w_neg):
        lib = _lib.require_gpu()
        pos, fin, neg = _GCLLossFn.forward(ctx, F, index, goff, flag, sel, sel1, sel2, pos_thresh, fin_thresh, neg_thresh,
                                           flags, pairpos)
        out = torch.empty(4, dtype=torch.float32, device=F.device)
        _lib.check(lib.gcl_loss_combine(_lib.ptr(pos), _lib.ptr(fin), pos.shape[0], _lib.ptr(neg), float(w_pos),
                                        float(w_fin), float(w_neg), _lib.ptr(out), _lib.stream()), "gcl_loss_combine")
        ctx.w = (float(w_pos), float(w_fin), float(w_neg))
        total, pm, fm, ng = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(pm, fm, ng)
        return total, pm, fm, ng

    @staticmethod
    @once_differentiable
    def backward(ctx, gtotal, _gp, _gf, _gn):
        lib = _lib.load()
        n_sel = ctx.saved_tensors[4].shape[0]
        dev = gtotal.device
        seeds = torch.empty(2 * n_sel + 1, dtype=torch.float32, device=dev)
        g = gtotal.reshape(1).contiguous().float()
        _lib.check(lib.gcl_loss_seed(_lib.ptr(g), ctx.w[0], ctx.w[1], ctx.w[2], n_sel, _lib.ptr(seeds),
                                     _lib.ptr(seeds[n_sel:]), _lib.ptr(seeds[2 * n_sel:]), _lib.stream()), "gcl_loss_seed")
        dF = _GCLLossFn.backward_terms(ctx, seeds[:n_sel], seeds[n_sel:2 * n_sel], seeds[2 * n_sel:])
        return (dF,) + (None,) * 14


LOSS_SQRT, LOSS_BLOCK, LOSS_PAIR, LOSS_NOFIN = 1, 2, 4, 8      # GCL_LOSS_* of include/gcl_amd.h


# Serialises every use of numpy's GLOBAL RandomState by the trainer's threads: legacy_choice holds it from get_state to
# set_state (the native shuffle in between runs without the interpreter lock), and train_steps holds it while it pulls
# the next batch from the caller's iterator (a num_workers=0 loader with np.random augmentation draws there).  Loader
# code running on OTHER threads of the caller must not use the global np.random while train_steps runs.
import threading
NP_RANDOM_LOCK = threading.RLock()


def legacy_choice(n, k):
    """``np.random.choice(n, k, replace=False)`` on numpy's global RandomState -- same values, same stream position
    afterwards -- computed by the native host routine gcl_host_legacy_choice outside the interpreter lock (numpy: 8 ms per
    call at n = 0.5 M, holding the lock, which stalls the thread that enqueues the GPU work).  Small n go to numpy."""
    if n < 4096 or k > n:
        with NP_RANDOM_LOCK:
            return np.random.choice(n, k, replace=False)
    import ctypes
    lib = _lib.load()
    with NP_RANDOM_LOCK:
        st = np.random.get_state()
        if st[0] != "MT19937":
            return np.random.choice(n, k, replace=False)
        key = np.array(st[1], dtype=np.uint32, copy=True)
        pos = ctypes.c_int32(int(st[2]))
        work, out = np.empty(n + n // 32 + 64, np.int64), np.empty(k, np.int64)
        _lib.check(lib.gcl_host_legacy_choice(ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos), n, k,
                                              ctypes.c_void_p(work.ctypes.data), ctypes.c_void_p(out.ctypes.data)),
                   "gcl_host_legacy_choice")
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def draw_selections(n_groups, n_out, max_pos_cluster, max_hn_samples, group_sizes=None, pair_positive=False,
                    random_negative=False):
    """The three host-side ``np.random.choice`` draws of finest_contrastive_loss, in the reference's order
    (lib/colocation_trainer.py:457, :506-507): selected positive groups, then the two negative row subsets.
    ``np.random.choice(n, k, replace=False)`` permutes all n rows (~4 ms at n = 0.5 M): the trainer therefore draws at
    the START of a step, while the GPU is still busy with the previous step's backward pass."""
    if n_groups > max_pos_cluster:
        pos_sel = legacy_choice(n_groups, max_pos_cluster)
    else:
        pos_sel = np.arange(n_groups)
    pair_pos = None
    if pair_positive:        # two members per selected group, drawn inside the group loop (:467)
        sizes = np.asarray(group_sizes)
        pair_pos = np.stack([np.random.choice(int(sizes[i]), 2, replace=False) for i in pos_sel]).astype(np.int32)
    sel_hn1 = legacy_choice(n_out, min(n_out, max_hn_samples))
    sel_hn2 = legacy_choice(n_out, min(n_out, max_hn_samples))
    if random_negative:       # use_hard_negative == False: one random column per row, drawn after the two subsets (:514)
        with NP_RANDOM_LOCK:
            cols = np.array([np.random.choice(len(sel_hn2), 1)[0] for _ in range(len(sel_hn1))], dtype=np.int64)
        return pos_sel, sel_hn1, sel_hn2, pair_pos, cols
    return pos_sel, sel_hn1, sel_hn2, pair_pos


def prepare_loss_inputs(group, index, finest_flag, device):
    """(goff, index, flag) on the device in the layouts the loss kernels read: exclusive group offsets int64 [G + 1],
    member rows int64, finest flags uint8.  ``train_steps`` computes them for the next batch on the side stream."""
    group = torch.as_tensor(group)
    goff = torch.zeros(int(group.shape[0]) + 1, dtype=torch.int64, device=device)       # no host sync when group is on the GPU
    goff[1:] = torch.cumsum(group.to(device, torch.int64, non_blocking=True), 0)
    index = torch.as_tensor(index).to(device, torch.int64, non_blocking=True).contiguous()
    flag = torch.as_tensor(finest_flag).to(device, non_blocking=True).to(torch.uint8).contiguous()
    return goff, index, flag


def finest_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster=256, max_hn_samples=2048,
                            points=None, batch_lengths=None, pos_thresh=0.1, neg_thresh=1.4, finest_thresh=0.2,
                            draws=None, square_loss=True, block_finest_gradient=False,
                            use_pair_group_positive_loss=False, use_hard_negative=True, finest_term=True,
                            prepared=None, total_weights=None):
    """(pos_loss, finest_loss, neg_loss) of lib/colocation_trainer.py:430-535 with its four config switches
    (defaults = scripts/train_gcl_kitti.sh:96-105).  ``finest_term=False`` gives ``location_contrastive_loss``
    (:734-809) when combined with ``square_loss=False``.

    ``draws=(pos_sel, sel_hn1, sel_hn2[, pair_pos[, random_cols]])`` replays recorded selections; otherwise they are
    drawn from ``np.random`` in the reference's order.  ``use_hard_negative=False`` (a debug switch no script sets)
    reproduces what the reference's code computes there: it indexes the distance matrix with an [M, 1] tensor
    (:514-515), which BROADCASTS -- every row j is paired with every drawn column c_i, the masks (:521-529) broadcast
    the same way, and the loss is the mean of relu(neg_thresh - D[j, c_i])^2 over the kept (i, j) cells
    (``random_negative_term``; pinned by a golden captured from the reference).  ``index_hash``, ``points`` and
    ``batch_lengths`` are unused (see module docstring).  ``total_weights=(w_pos, w_fin, w_neg)`` (the trainer's step, with
    hard negatives): returns ``(total, pos, fin, neg)`` from one autograd node instead -- same values, the scalar arithmetic
    in one launch each way; only ``total`` carries a gradient.
    """
    dev = F_out.device
    n_out = F_out.shape[0]
    group = torch.as_tensor(group)
    n_groups = int(group.shape[0])
    if draws is None:
        draws = draw_selections(n_groups, n_out, max_pos_cluster, max_hn_samples,
                                group.cpu().numpy() if use_pair_group_positive_loss else None,
                                use_pair_group_positive_loss, random_negative=not use_hard_negative)
    pos_sel, sel_hn1, sel_hn2 = draws[:3]
    pair_pos = draws[3] if len(draws) > 3 else None
    rand_cols = draws[4] if len(draws) > 4 else None
    if not use_hard_negative and rand_cols is None:
        raise ValueError("use_hard_negative=False needs the drawn columns (draws[4])")
    if use_pair_group_positive_loss and pair_pos is None:
        raise ValueError("use_pair_group_positive_loss needs the drawn member positions (draws[3])")
    if len(pos_sel) == 0:
        raise ZeroDivisionError("no positive group in the batch")
    flags = (0 if square_loss else LOSS_SQRT) | (LOSS_BLOCK if block_finest_gradient else 0) | \
            (LOSS_PAIR if use_pair_group_positive_loss else 0) | (0 if finest_term else LOSS_NOFIN)
    goff, index, flag = prepared if prepared is not None else prepare_loss_inputs(group, index, finest_flag, dev)
    to_dev32 = lambda a: None if a is None else \
        torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev, non_blocking=True)
    n_pos, n_hn = len(pos_sel), len(sel_hn1)                   # the three selections travel in ONE host -> device copy
    sel_host = torch.empty(n_pos + 2 * n_hn, dtype=torch.int64, pin_memory=True)     # pinned: the copy is asynchronous
    np.concatenate([np.asarray(pos_sel, dtype=np.int64), np.asarray(sel_hn1, dtype=np.int64),
                    np.asarray(sel_hn2, dtype=np.int64)], out=sel_host.numpy())
    sel_all = sel_host.to(dev, non_blocking=True)
    if total_weights is not None and use_hard_negative:
        # the trainer's step: (total, pos, fin, neg) from one autograd node (``_GCLTotalLossFn``); only total has a gradient
        return _GCLTotalLossFn.apply(F_out, index, goff, flag, sel_all[:n_pos], sel_all[n_pos:n_pos + n_hn],
                                     sel_all[n_pos + n_hn:], float(pos_thresh), float(finest_thresh), float(neg_thresh),
                                     flags, to_dev32(pair_pos) if use_pair_group_positive_loss else None,
                                     *[float(w) for w in total_weights])
    pos, fin, neg = _GCLLossFn.apply(F_out, index, goff, flag, sel_all[:n_pos], sel_all[n_pos:n_pos + n_hn],
                                     sel_all[n_pos + n_hn:], float(pos_thresh), float(finest_thresh), float(neg_thresh),
                                     flags, to_dev32(pair_pos) if use_pair_group_positive_loss else None)
    if not use_hard_negative:
        neg = random_negative_term(F_out, sel_all[n_pos:n_pos + n_hn], sel_all[n_pos + n_hn:], rand_cols, group, index,
                                   index_hash, float(neg_thresh))
    return pos.sum() / len(pos_sel), fin.sum() / len(pos_sel), neg


class _RandomNegFn(torch.autograd.Function):
    """mean over the kept (i, j) of relu(neg_thresh - sqrt(sum_c (F[rows_i[i]] - F[rows_j[j]])^2 + 1e-7))^2, evaluated in
    blocks of rows_i so that the [block, M, C] difference tensor -- never the [M, M, C] one (8.6 GB at M = 8192, C = 32,
    plus autograd's copies: ADVICE round 3) -- is what lives at a time; backward recomputes block by block."""
    BLOCK_BYTES = 256 << 20

    @staticmethod
    def _blocks(m_i, m_j, c):
        rows = max(1, min(m_i, _RandomNegFn.BLOCK_BYTES // max(1, m_j * c * 4)))
        return [(a, min(m_i, a + rows)) for a in range(0, m_i, rows)]

    @staticmethod
    def _block_terms(Fi, Fj, rows_i, rows_j, n_out, pos_keys, neg_thresh):
        D = torch.sqrt(((Fi.unsqueeze(1) - Fj.unsqueeze(0)) ** 2).sum(2) + 1e-7)       # lib/metrics.py:22-29, exact form
        a, b = rows_j.unsqueeze(0), rows_i.unsqueeze(1)
        keys = torch.minimum(a * n_out + b, a + b * n_out)                            # _neg_hash (util/misc.py:39-40)
        keep = (a != b) & ~torch.isin(keys, pos_keys)
        return torch.relu(neg_thresh - D[keep]).pow(2), keep

    @staticmethod
    def forward(ctx, F_out, rows_i, rows_j, pos_keys, neg_thresh):
        n_out, c = F_out.shape
        Fj = F_out[rows_j]
        tot = torch.zeros((), dtype=torch.float64, device=F_out.device)
        cnt = 0
        for a, b in _RandomNegFn._blocks(len(rows_i), len(rows_j), c):
            t, keep = _RandomNegFn._block_terms(F_out[rows_i[a:b]], Fj, rows_i[a:b], rows_j, n_out, pos_keys, neg_thresh)
            tot += t.double().sum()
            cnt += int(keep.sum())
        ctx.save_for_backward(F_out, rows_i, rows_j, pos_keys)
        ctx.neg_thresh, ctx.cnt = neg_thresh, cnt
        return (tot / max(cnt, 1)).float() if cnt else torch.full((), float("nan"), device=F_out.device)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        F_out, rows_i, rows_j, pos_keys = ctx.saved_tensors
        n_out, c = F_out.shape
        gF = torch.zeros_like(F_out)
        if not ctx.cnt:
            return gF, None, None, None, None
        scale = gout / ctx.cnt
        for a, b in _RandomNegFn._blocks(len(rows_i), len(rows_j), c):
            with torch.enable_grad():
                Fi = F_out[rows_i[a:b]].detach().requires_grad_(True)
                Fj = F_out[rows_j].detach().requires_grad_(True)
                t, _ = _RandomNegFn._block_terms(Fi, Fj, rows_i[a:b], rows_j, n_out, pos_keys, ctx.neg_thresh)
                gi, gj = torch.autograd.grad(t.sum(), (Fi, Fj))
            gF.index_add_(0, rows_i[a:b], gi * scale)
            gF.index_add_(0, rows_j, gj * scale)
        return gF, None, None, None, None


def random_negative_term(F_out, sel1, sel2, cols, group, index, index_hash, neg_thresh):
    """The negative term with ``use_hard_negative == False`` exactly as the reference's code evaluates it
    (lib/colocation_trainer.py:513-530): ``D_fs[torch.arange(M), D_fs_ind]`` with ``D_fs_ind`` of shape [M, 1]
    broadcasts to ``out[i, j] = D_fs[j, c_i]``, and ``sel_hn2[D_fs_ind]`` / ``_neg_hash`` / ``np.isin`` broadcast
    alike: the loss is the mean over ALL (i, j) of relu(neg_thresh - D_fs[j, c_i])^2 whose pair (sel1[j], sel2[c_i]) is
    neither the same row nor inside one positive group.  Distances in the exact (a - b)^2 form of lib/metrics.py:22-29
    (torch on the device, differentiable, evaluated in row blocks: _RandomNegFn); the positive-pair keys are ``index_hash``
    when the loader supplies it, else ``_exhaustive_hash(group, index)`` (util/misc.py:29-36).  A debug path, no HIP kernel."""
    from gcl_amd.util.misc import _exhaustive_hash
    dev = F_out.device
    n_out = F_out.shape[0]
    cols_d = torch.as_tensor(np.asarray(cols, dtype=np.int64)).to(dev)
    rows_j, rows_i = sel1, sel2[cols_d]                                          # sel1[j], sel2[c_i]
    if index_hash is None:
        split = torch.split(torch.as_tensor(index).cpu(), tuple(int(g) for g in torch.as_tensor(group).tolist()))
        index_hash = _exhaustive_hash(split, n_out)
    pos_keys = torch.as_tensor(np.asarray(index_hash, dtype=np.int64)).to(dev)
    return _RandomNegFn.apply(F_out, rows_i, rows_j, pos_keys, float(neg_thresh))


def location_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster=256, max_hn_samples=2048,
                              points=None, batch_lengths=None, pos_thresh=0.1, neg_thresh=1.4, draws=None,
                              use_pair_group_positive_loss=False, use_hard_negative=True):
    """lib/colocation_trainer.py:734-809 (the trainer's loss when finest_weight == 0): non-squared positive term, no
    finest term (returned as 0), the same negative term."""
    return finest_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster, max_hn_samples,
                                   points, batch_lengths, pos_thresh, neg_thresh, 0.0, draws, square_loss=False,
                                   block_finest_gradient=False,
                                   use_pair_group_positive_loss=use_pair_group_positive_loss,
                                   use_hard_negative=use_hard_negative, finest_term=False)


class _CircleGroupFn(torch.autograd.Function):
    """Per-group terms of location_circle_loss and the groups' mean features (gcl_circle_group_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, F, index, goff, flag, sel, pos_thresh, fin_thresh, log_scale, flags, pairpos):
        lib = _lib.require_gpu()
        F = F.contiguous()
        n, c = F.shape
        n_sel = sel.shape[0]
        dev = F.device
        pos = torch.empty(n_sel, dtype=torch.float32, device=dev)
        fin = torch.empty(n_sel, dtype=torch.float32, device=dev)
        mean = torch.empty((n_sel, c), dtype=torch.float32, device=dev)
        _lib.check(lib.gcl_circle_group_fwd(_lib.ptr(F, torch.float32), c, _lib.ptr(index, torch.int64),
                                            _lib.ptr(goff, torch.int64), _lib.ptr(flag, torch.uint8),
                                            _lib.ptr(sel, torch.int64), n_sel, pos_thresh, fin_thresh, log_scale,
                                            int(flags), _lib.ptr(pairpos, torch.int32), _lib.ptr(pos), _lib.ptr(fin),
                                            _lib.ptr(mean), _lib.stream()), "gcl_circle_group_fwd")
        ctx.save_for_backward(F, index, goff, flag, sel, pairpos)
        ctx.cfg = (pos_thresh, fin_thresh, log_scale, int(flags))
        return pos, fin, mean

    @staticmethod
    @once_differentiable
    def backward(ctx, gpos, gfin, gmean):
        lib = _lib.load()
        F, index, goff, flag, sel, pairpos = ctx.saved_tensors
        pos_thresh, fin_thresh, log_scale, flags = ctx.cfg
        dF = torch.zeros_like(F)
        gpos_c, gfin_c, gmean_c = gpos.contiguous(), gfin.contiguous(), gmean.contiguous()     # named: see _GCLLossFn
        _lib.check(lib.gcl_circle_group_bwd(_lib.ptr(F), F.shape[1], _lib.ptr(index), _lib.ptr(goff), _lib.ptr(flag),
                                            _lib.ptr(sel), sel.shape[0], pos_thresh, fin_thresh, log_scale, flags,
                                            _lib.ptr(pairpos), _lib.ptr(gpos_c), _lib.ptr(gfin_c), _lib.ptr(gmean_c),
                                            _lib.ptr(dF), _lib.stream()), "gcl_circle_group_bwd")
        return (dF,) + (None,) * 9


def draw_circle_selections(n_groups, max_pos_cluster, group_sizes=None, pair_positive=False):
    """np.random draws of location_circle_loss in the reference's order (:561-564 sorted selection, :599 pairs)."""
    if n_groups > max_pos_cluster:
        pos_sel = np.sort(np.random.choice(n_groups, max_pos_cluster, replace=False))
    else:
        pos_sel = np.arange(n_groups)
    pair_pos = None
    if pair_positive:
        sizes = np.asarray(group_sizes)
        pair_pos = np.stack([np.random.choice(int(sizes[i]), 2, replace=False) for i in pos_sel]).astype(np.int32)
    return pos_sel, pair_pos


def location_circle_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster=256, max_hn_samples=None,
                         points=None, batch_lengths=None, pos_thresh=0.1, neg_thresh=1.4, finest_thresh=0.2,
                         safe_radius=0.75, log_scale=16, draws=None, square_loss=True, block_finest_gradient=True,
                         use_pair_group_positive_loss=False):
    """(pos_loss, finest_loss, neg_loss) of lib/colocation_trainer.py:538-681.  ``points`` [N, 3] (the trainer passes
    the voxel coordinates ``sinput_C[:, 1:]``, :858) and ``batch_lengths`` (rows per sample) are required here.
    The per-group terms and the group means come from one HIP kernel pair; the negative term is a handful of torch
    operations on the [M, M] matrices of the M <= max_pos_cluster selected groups (:642-676)."""
    dev = F_out.device
    group_t = torch.as_tensor(group)
    n_groups = int(group_t.shape[0])
    if draws is None:
        draws = draw_circle_selections(n_groups, max_pos_cluster,
                                       group_t.cpu().numpy() if use_pair_group_positive_loss else None,
                                       use_pair_group_positive_loss)
    pos_sel, pair_pos = draws[0], (draws[1] if len(draws) > 1 else None)
    if len(pos_sel) == 0:
        raise ZeroDivisionError("no positive group in the batch")
    if use_pair_group_positive_loss and pair_pos is None:
        raise ValueError("use_pair_group_positive_loss needs the drawn member positions (draws[1])")
    flags = (0 if square_loss else LOSS_SQRT) | (LOSS_BLOCK if block_finest_gradient else 0) | \
            (LOSS_PAIR if use_pair_group_positive_loss else 0)
    goff = torch.zeros(n_groups + 1, dtype=torch.int64, device=dev)
    goff[1:] = torch.cumsum(group_t.to(dev, torch.int64), 0)
    index_d = torch.as_tensor(index).to(dev, torch.int64).contiguous()
    flag = torch.as_tensor(finest_flag).to(dev).to(torch.uint8).contiguous()
    sel = torch.from_numpy(np.ascontiguousarray(pos_sel, dtype=np.int64)).to(dev)
    pp = None if pair_pos is None else torch.from_numpy(np.ascontiguousarray(pair_pos, dtype=np.int32)).to(dev)
    pos, fin, feats_sel = _CircleGroupFn.apply(F_out, index_d, goff, flag, sel, float(pos_thresh), float(finest_thresh),
                                               float(log_scale), flags, pp)
    pos_loss, finest_loss = pos.sum() / len(pos_sel), fin.sum() / len(pos_sel)
    # anchors: first member of every selected group (:583) and the sample it belongs to (:589-591)
    pivot = index_d[goff[sel]]
    coords_sel = torch.as_tensor(points).to(dev)[pivot].float()
    acc = torch.cumsum(torch.as_tensor(np.asarray(batch_lengths, dtype=np.float64)), 0).to(dev)
    item = torch.sum(pivot[:, None].double() > acc[None, :], dim=1)          # np.sum(pivot > accumulated)
    batch_mask = item[:, None] == item[None, :]
    d2 = -2 * coords_sel @ coords_sel.T + (coords_sel ** 2).sum(-1)[:, None] + (coords_sel ** 2).sum(-1)[None, :]
    coords_dist = torch.sqrt(torch.clamp(d2, min=1e-12))                     # util/misc.py:7-26
    feats_dist = torch.sqrt(torch.clamp(2 - 2 * feats_sel @ feats_sel.T, min=1e-12))     # normalised=True (:657-659)
    neg_mask = (coords_dist > safe_radius) & batch_mask
    has_neg = neg_mask.sum(-1) > 0
    neg_weight = torch.clamp(neg_thresh - (feats_dist + 1e5 * (~neg_mask).float()), min=0).detach()
    lse = torch.logsumexp(log_scale * (neg_thresh - feats_dist) * neg_weight, dim=-1)
    neg_loss = (torch.nn.functional.softplus(lse) / log_scale)[has_neg].mean()
    return pos_loss, finest_loss, neg_loss


class FinestContrastiveLossTrainer:
    """The GCL trainer's hot loop body (``_train_epoch`` :811-916) without the dataset / logging / checkpoint shell."""

    def __init__(self, config=None, model=None, device=None, ddp=None):
        self.config = config or make_config()
        cfg = self.config
        self.device = torch.device(device if device is not None else "cuda:0")
        if model is None:
            Model = load_model(cfg.model)
            model = Model(1, cfg.model_n_out, bn_momentum=cfg.bn_momentum, normalize_feature=cfg.normalize_feature,
                          conv1_kernel_size=cfg.conv1_kernel_size, D=3)
        self.model = model.to(self.device)
        self.ddp = ddp
        if ddp is not None:
            ddp.attach(self.model)          # flat parameter/gradient buffers + initial broadcast
        # lib/colocation_trainer.py:73-77: SGD(lr, momentum, weight_decay), dampening left at its default
        if os.environ.get("GCL_FUSED_SGD", "1") == "1":          # same update, one launch (gcl_amd/lib/optim.py)
            from gcl_amd.lib.optim import FusedSGD
            self.optimizer = FusedSGD(self.model.parameters(), lr=cfg.lr, momentum=cfg.momentum,
                                      weight_decay=cfg.weight_decay)
        else:
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=cfg.lr, momentum=cfg.momentum,
                                             weight_decay=cfg.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, cfg.exp_gamma)
        self.pos_weight, self.neg_weight, self.finest_weight = cfg.pos_weight, cfg.neg_weight, cfg.finest_weight
        self._params = [p for p in self.model.parameters()]
        self.map_prefetch = os.environ.get("GCL_MAP_PREFETCH", "1") == "1"
        self._side = None

    # ---- whole-network plan (gcl_amd/MinkowskiEngine/native.py) ------------------------------------------------
    def _plan_for_step(self, micro):
        """The NetworkPlan every micro-batch of this step will run through, or None (Tape / per-layer autograd path)."""
        plan_for = getattr(self.model, "plan_for", None)
        if plan_for is None or not self.model.training:
            return None
        plan = None
        for b in micro:
            mgr = b.get("_coordinate_manager") if isinstance(b, dict) else None
            if mgr is None or mgr.native is None:
                return None
            probe = types.SimpleNamespace(coordinate_manager=mgr, coordinate_map_key=ME.CoordinateMapKey(1),
                                          F=b["sinput_F"])
            plan = plan_for(probe)
            if plan is None:
                return None
        return plan

    def _seat_gradients(self):
        """``p.grad`` of every parameter = its view of ONE flat buffer (FlatDDP's when data-parallel)."""
        if getattr(self, "_grad_views", None) is None:
            ps = self._params
            total = sum(p.numel() for p in ps)
            if self.ddp is not None:
                if [id(p) for p in self.ddp.params] != [id(p) for p in ps]:
                    raise RuntimeError("FlatDDP and the trainer disagree on the parameter list")
                self._flat_grad = self.ddp.flat_grad
            else:
                self._flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
            self._flat_grad2 = None
            self._grad_views, off = [], 0
            for p in ps:
                self._grad_views.append(self._flat_grad[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._grad_views2 = None
        if self._grad_views2 is None and int(getattr(self.config, "iter_size", 1)) > 1:
            self._flat_grad2 = torch.zeros_like(self._flat_grad)
            self._grad_views2, off = [], 0
            for p in self._params:
                self._grad_views2.append(self._flat_grad2[off:off + p.numel()].view_as(p))
                off += p.numel()
        for p, v in zip(self._params, self._grad_views):
            g = p.grad
            if g is None or g.data_ptr() != v.data_ptr():
                p.grad = v

    def _plan_buckets(self):
        if getattr(self, "_bucket_map", None) is None:
            self._bucket_map = {i: self.ddp._bucket_of[id(p)] for i, p in enumerate(self._params)}
        return self._bucket_map

    def location_loss(self, F_out, group, index, index_hash, finest_flag, max_pos_cluster, max_hn_samples,
                      points=None, batch_lengths=None, draws=None, prepared=None, total_weights=None):
        cfg = self.config
        if cfg.use_group_circle_loss:         # lib/colocation_trainer.py:423-424
            return location_circle_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster, max_hn_samples,
                                        points, batch_lengths, cfg.pos_thresh, cfg.neg_thresh, cfg.finest_thresh,
                                        cfg.safe_radius, 16, draws, cfg.square_loss, cfg.block_finest_gradient,
                                        cfg.use_pair_group_positive_loss)
        if cfg.finest_weight == 0:            # lib/colocation_trainer.py:425-428
            return location_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster,
                                             max_hn_samples, points, batch_lengths, cfg.pos_thresh, cfg.neg_thresh,
                                             draws, cfg.use_pair_group_positive_loss, cfg.use_hard_negative)
        return finest_contrastive_loss(F_out, group, index, index_hash, finest_flag, max_pos_cluster, max_hn_samples,
                                       points, batch_lengths, cfg.pos_thresh, cfg.neg_thresh, cfg.finest_thresh, draws,
                                       cfg.square_loss, cfg.block_finest_gradient, cfg.use_pair_group_positive_loss,
                                       cfg.use_hard_negative, prepared=prepared, total_weights=total_weights)

    def forward_loss(self, input_dict, draws=None, fused_total=False):
        """``fused_total``: the weighted total comes out of the loss node itself (one optimizer step on one batch; with
        gradient accumulation the terms are rescaled first, :875-877, and the sum stays in torch)."""
        cfg = self.config
        sinput = ME.SparseTensor(input_dict["sinput_F"].to(self.device, non_blocking=True),
                                 coordinates=input_dict["sinput_C"].to(self.device, non_blocking=True),
                                 coordinate_manager=input_dict.get("_coordinate_manager"))
        F_out = self.model(sinput).F
        fuse = (fused_total and not cfg.use_group_circle_loss and cfg.finest_weight != 0 and cfg.use_hard_negative
                and os.environ.get("GCL_FUSED_LOSS_TOTAL", "1") == "1")
        terms = self.location_loss(
            F_out, input_dict["group"], input_dict["index"], input_dict.get("index_hash"), input_dict["finest_flag"],
            max_pos_cluster=cfg.num_pos_per_batch * cfg.batch_size,
            max_hn_samples=cfg.num_hn_samples_per_batch * cfg.batch_size,
            points=input_dict["sinput_C"][:, 1:] if cfg.use_group_circle_loss else None,
            batch_lengths=input_dict.get("batch_lengths"), draws=draws, prepared=input_dict.get("_loss_inputs"),
            total_weights=(self.pos_weight, self.finest_weight, self.neg_weight) if fuse else None)
        if len(terms) == 4:
            return terms[0], tuple(terms[1:]), F_out
        pos, fin, neg = terms
        loss = self.pos_weight * pos + self.finest_weight * fin + self.neg_weight * neg
        return loss, (pos, fin, neg), F_out

    def _draw_for(self, input_dict):
        cfg = self.config
        if cfg.use_group_circle_loss:
            sizes = torch.as_tensor(input_dict["group"]).cpu().numpy() if cfg.use_pair_group_positive_loss else None
            return draw_circle_selections(len(input_dict["group"]), cfg.num_pos_per_batch * cfg.batch_size, sizes,
                                          cfg.use_pair_group_positive_loss)
        sizes = None
        if cfg.use_pair_group_positive_loss:
            sizes = torch.as_tensor(input_dict["group"]).cpu().numpy()
        return draw_selections(len(input_dict["group"]), len(input_dict["sinput_C"]),
                               cfg.num_pos_per_batch * cfg.batch_size, cfg.num_hn_samples_per_batch * cfg.batch_size,
                               sizes, cfg.use_pair_group_positive_loss, random_negative=not cfg.use_hard_negative)

    def _prefetch_maps(self, batch):
        """Loader-side work for a coming batch that is already on the device: its coordinate manager with every kernel
        map, sorted table and pair list of the network, built on a low-priority SIDE stream (CoordinateManager.prefetch)
        by a helper thread while the previous step is being enqueued.  The maps depend on the coordinates only; building
        them ahead removes the level-size read-back (the one host sync of a step) and ~150 small launches from the
        training stream, whose gaps they fill instead.  Returns the batch (a copy carrying the manager) to train on."""
        C = batch.get("sinput_C") if isinstance(batch, dict) else None
        if isinstance(batch, dict) and "_coordinate_manager" in batch:       # a stale manager of an earlier pass
            batch = {k: v for k, v in batch.items()
                     if k not in ("_coordinate_manager", "_maps_event", "_loss_inputs", "_map_arena")}
        specs = getattr(getattr(self, "model", None), "map_specs", None)
        if getattr(self, "map_prefetch", False) and specs is not None and isinstance(C, torch.Tensor) and C.is_cuda:
            with torch.cuda.device(self.device):
                # one side stream per helper THREAD (GCL_MAP_WORKERS > 1: two batches' maps in flight, each build's host reads
                # wait for its own stream only)
                import threading
                tl = self.__dict__.setdefault("_side_local", threading.local())
                side = getattr(tl, "stream", None)
                if side is None:
                    lo, hi = torch.cuda.Stream.priority_range()         # (lowest priority, highest priority)
                    # low: with the helpers two steps ahead nothing waits for the maps, and at high priority their kernels
                    # took the chip from the training stream (13.0 -> 12.7 ms per step; with the maps of a batch cached, a
                    # diagnostic, the step is 11.95 ms: building them still costs ~0.75 ms of a step)
                    prio = {"low": lo, "high": hi}.get(os.environ.get("GCL_SIDE_PRIORITY", "low"), 0)
                    side = tl.stream = torch.cuda.Stream(device=self.device, priority=prio)
                from gcl_amd.MinkowskiEngine import native
                nspecs = getattr(self.model, "native_map_specs", None)
                use_native = native.PLAN_ENABLED and nspecs is not None
                slot = None
                with torch.cuda.stream(side):
                    ev = batch.get("_h2d_event")
                    if ev is not None:
                        side.wait_event(ev)
                    if use_native:
                        # ONE native call (interpreter lock released, both host syncs inside) into a pooled arena
                        slot = self._take_map_arena()
                        if slot["free"] is not None:
                            side.wait_event(slot["free"])
                        mgr = ME.CoordinateManager.build_native(C, nspecs(), arena=slot["arena"])
                        slot["arena"] = mgr.native.arena
                    else:
                        mgr = ME.CoordinateManager(C).prefetch(specs())
                    prep = prepare_loss_inputs(batch["group"], batch["index"], batch["finest_flag"], self.device)
                    done = torch.cuda.Event()
                    done.record(side)
            batch = dict(batch)               # never mutate the caller's dict (it may be fed again)
            batch["_coordinate_manager"], batch["_maps_event"], batch["_loss_inputs"] = mgr, done, prep
            batch["_map_arena"] = slot
        return batch

    def _take_map_arena(self):
        """A free arena of the map pool (``{"arena", "free", "busy"}``): handed out by the map helper thread, released by
        ``release_batch`` on the training thread once the batch's backward pass is enqueued (the event recorded there
        is what the side stream waits for before the arena is overwritten).  Grows on demand; never recycles a busy one."""
        import threading
        if getattr(self, "_map_arenas", None) is None:
            self._map_arenas, self._map_lock = [], threading.Lock()
        with self._map_lock:
            slot = next((a for a in self._map_arenas if not a["busy"]), None)
            if slot is None:
                slot = {"arena": None, "free": None, "busy": False, "lock": self._map_lock}
                self._map_arenas.append(slot)
            slot["busy"] = True
        return slot

    def train_steps(self, batches):
        """The epoch loop (``_train_epoch`` :811-916): yields train_step(...) for every optimizer step, i.e. for every
        ``config.iter_size`` consecutive batches (:838, a trailing incomplete group is dropped like ``len // iter_size``).
        Two helper threads work ``GCL_PREFETCH_DEPTH`` (default 2) steps ahead of the step being enqueued (the map
        helper's latency is a full step when the GPU is saturated -- its side-stream kernels queue behind the training
        stream's -- so one step of lead made the enqueuing thread wait ~3 ms per step for it): one makes the ``np.random`` draws of the next
        step (two permutations of all N rows per batch: ~12 ms of native code outside the interpreter lock; every draw
        still after the previous batch's, so the random stream is consumed in the order of a serial loop), the other
        builds the next batches' coordinate maps on a side stream (``_prefetch_maps``)."""
        from concurrent.futures import ThreadPoolExecutor
        it = iter(batches)
        k = max(1, int(getattr(self.config, "iter_size", 1)))

        def take():
            grp = []
            with NP_RANDOM_LOCK:          # a loader that draws from the global np.random must not interleave with a draw
                for b in it:
                    grp.append(b)
                    if len(grp) == k:
                        return grp
            return None

        import time
        trace = os.environ.get("GCL_TRACE_HELPERS") == "1"          # diagnostic: wall / CPU time of the helper threads
        acc = self._helper_times = getattr(self, "_helper_times", {"draw": [0.0, 0.0, 0], "maps": [0.0, 0.0, 0]})

        def timed(name, fn, grp):
            if not trace:
                return fn(grp)
            w0, c0 = time.perf_counter(), time.thread_time()
            out = fn(grp)
            a = acc[name]
            a[0] += time.perf_counter() - w0
            a[1] += time.thread_time() - c0
            a[2] += 1
            return out

        # GCL_TRACE_HELPERS=2 (diagnostic): host time stamps of every helper call / wait / enqueue and a GPU event behind
        # every step, for tools/host_gpu_timeline.py (who waits for whom at a step boundary)
        tl = self._timeline = [] if os.environ.get("GCL_TRACE_HELPERS") == "2" else None
        if tl is not None:
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda.synchronize()
            tl.append(("anchor", 0, time.perf_counter(), e0))

        def stamped(name, fn, grp):
            if tl is None:
                return fn(grp)
            t0 = time.perf_counter()
            out = fn(grp)
            tl.append((name, 0, t0, time.perf_counter()))
            return out

        def draw(grp):
            return timed("draw", lambda g: stamped("draw", lambda g2: [self._draw_for(b) for b in g2], g), grp)

        def maps(grp):
            return timed("maps", lambda g: stamped("maps", lambda g2: [self._prefetch_maps(b) for b in g2], g), grp)

        n_map = max(1, int(os.environ.get("GCL_MAP_WORKERS", "1")))
        depth = max(1, int(os.environ.get("GCL_PREFETCH_DEPTH", "2")))      # steps the helpers work ahead
        depth = max(depth, n_map)
        from collections import deque
        # GCL_MAP_WORKERS (default 1; experiment 77, round 6): a map build waits twice for its low-priority stream and beside a
        # saturated GPU that latency is a whole step, so ONE helper keeps exactly pace with the steps, the enqueuing thread waits
        # for every build and the training stream idles ~ 0.45 ms at each step boundary (tools/r06_timeline.sh).  Two helpers
        # (two builds in flight on two streams) remove that wait and cost more than it: 11.80 / 11.83 -> 12.38 / 12.40 ms per
        # step (three: 12.21 - 12.46; depth 3 with one helper: 11.79) -- the single helper throttles the map stream to the rate
        # at which it disturbs the training stream least.
        with ThreadPoolExecutor(max_workers=1) as draw_pool, ThreadPoolExecutor(max_workers=n_map) as map_pool:
            pending = deque()

            def refill():
                while len(pending) < depth:
                    grp = take()
                    if grp is None:
                        return
                    pending.append((draw_pool.submit(draw, grp), map_pool.submit(maps, grp)))

            refill()
            while pending:
                fd, fm = pending.popleft()
                w0 = time.perf_counter()
                draws, cur = fd.result(), fm.result()
                if trace:       # how long the enqueuing thread stood waiting for its helpers
                    wt = acc.setdefault("wait", [0.0, 0.0, 0])
                    wt[0] += time.perf_counter() - w0
                    wt[2] += 1
                refill()
                if tl is None:
                    yield self.train_step(cur if k > 1 else cur[0], draws if k > 1 else draws[0])
                    continue
                w1 = time.perf_counter()
                out = self.train_step(cur if k > 1 else cur[0], draws if k > 1 else draws[0])
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                tl.append(("step", 0, w0, w1, time.perf_counter(), ev))
                yield out

    # ---- validation step (``_valid_epoch`` :306-379) -----------------------------------------------------------
    @staticmethod
    def apply_transform(pts, trans):
        """lib/colocation_trainer.py:234-239."""
        return pts @ trans[:3, :3].t() + trans[:3, 3]

    def find_corr(self, xyz0, xyz1, F0, F1, subsample_size=-1):
        """lib/colocation_trainer.py:381-395 (np.random.choice subsample of both clouds, then feature 1-NN)."""
        from gcl_amd.lib.eval import find_corr
        return find_corr(xyz0, xyz1, F0, F1, subsample_size=subsample_size, nn_max_n=getattr(self.config, "nn_max_n", 500))

    def evaluate_hit_ratio(self, xyz0, xyz1, T_gth, thresh=0.1):
        """lib/colocation_trainer.py:397-400."""
        moved = self.apply_transform(xyz0, T_gth)
        dist = torch.sqrt(((moved - xyz1) ** 2).sum(1) + 1e-6)
        return (dist < thresh).float().mean().item()

    def _valid_epoch(self, val_batches, val_max_iter=None):
        """``_valid_epoch`` (lib/colocation_trainer.py:306-379) over ``val_batches`` (pair dicts of
        ``collate_debug_pair_fn``: sinput{0,1}_C / _F, pcd0 / pcd1, T_gt): eval-mode features of both clouds (ONE
        forward pass per pair: bitwise equal to the reference's two, ``scripts.test_kitti.forward_clouds``), feature
        1-NN correspondences on 5000-row subsamples, ``est_quad_linear_robust``, and the five meters the reference
        returns.  The dataset shell (``reset_seed``, timers, logging) stays with the caller."""
        from gcl_amd.lib.metrics import corr_dist
        from gcl_amd.scripts.test_kitti import AverageMeter, forward_clouds
        from gcl_amd.util.transform_estimation import est_quad_linear_robust
        self.model.eval()
        limit = getattr(self.config, "val_max_iter", -1) if val_max_iter is None else val_max_iter
        hit_ratio_meter, feat_match_ratio, loss_meter, rte_meter, rre_meter = (AverageMeter() for _ in range(5))
        num_data = 0
        with torch.no_grad(), torch.cuda.device(self.device):
            for input_dict in val_batches:
                if 0 < limit <= num_data:
                    break
                dev = self.device
                F0, F1 = forward_clouds(self.model, [(input_dict["sinput0_F"].to(dev), input_dict["sinput0_C"].to(dev).int()),
                                                     (input_dict["sinput1_F"].to(dev), input_dict["sinput1_C"].to(dev).int())])
                xyz0, xyz1, T_gt = input_dict["pcd0"][0], input_dict["pcd1"][0], input_dict["T_gt"]
                T_gt = torch.as_tensor(T_gt).float().reshape(4, 4)
                xyz0_corr, xyz1_corr = self.find_corr(xyz0, xyz1, F0, F1, subsample_size=5000)
                T_est = est_quad_linear_robust(xyz0_corr, xyz1_corr)
                loss_meter.update(float(corr_dist(T_est, T_gt, xyz0, xyz1, weight=None)))
                rte_meter.update(float(np.linalg.norm((T_est[:3, 3] - T_gt[:3, 3]).numpy())))
                with np.errstate(invalid="ignore"):
                    rre = float(np.arccos((float(torch.trace(T_est[:3, :3].t() @ T_gt[:3, :3])) - 1) / 2))
                if not np.isnan(rre):
                    rre_meter.update(rre)
                hit_ratio = self.evaluate_hit_ratio(xyz0_corr, xyz1_corr, T_gt,
                                                    thresh=getattr(self.config, "hit_ratio_thresh", 0.3))
                hit_ratio_meter.update(hit_ratio)
                feat_match_ratio.update(float(hit_ratio > 0.05))
                num_data += 1
        return {"loss": loss_meter.avg, "rre": rre_meter.avg, "rte": rte_meter.avg,
                "feat_match_ratio": feat_match_ratio.avg, "hit_ratio": hit_ratio_meter.avg, "n": num_data}

    def train_step(self, input_dict, draws=None):
        """One optimizer step.  ``input_dict``: one batch, or a list of ``iter_size`` batches whose gradients are
        accumulated with every loss term divided by ``iter_size`` (lib/colocation_trainer.py:838-887); ``draws`` then is
        a list too.  Returns device scalars (summed over the micro-batches like ``batch_loss`` :880-883; no host sync)."""
        with torch.cuda.device(self.device):      # the HIP kernels launch on the current device's stream
            return self._train_step(input_dict, draws)

    def _train_step(self, input_dict, draws=None):
        self.model.train()
        micro = list(input_dict) if isinstance(input_dict, (list, tuple)) else [input_dict]
        n_micro = len(micro)
        if draws is None:     # host RNG first: overlaps with the GPU work still queued from the previous step
            mdraws = [self._draw_for(b) for b in micro]
        else:
            mdraws = list(draws) if isinstance(input_dict, (list, tuple)) else [draws]
        plan = self._plan_for_step(micro)
        if plan is not None:
            # whole-network plan: every gradient is WRITTEN into its seat in the flat buffer (no zero fill, no accumulate)
            self._seat_gradients()
        elif self.ddp is not None:
            self.ddp.flat_grad.zero_()                     # one memset; gradients stay seated in the flat buffer
        else:
            if os.environ.get("GCL_ZERO_NONE", "1") == "1":     # gradients are assigned, not accumulated into zeros
                for p in self._params:
                    p.grad = None
            else:
                self.optimizer.zero_grad(set_to_none=False)
        tot_loss, tot_parts, n_rows = None, None, 0
        for i, (b, d) in enumerate(zip(micro, mdraws)):
            if self.ddp is not None:
                self.ddp.set_last_microstep(i == n_micro - 1)     # bucket all-reduces start in the LAST backward only
            if plan is not None:
                # micro-batch 0 writes the seats; later ones write a second buffer that is added afterwards (:875-887)
                plan.grad_targets = self._grad_views if i == 0 else self._grad_views2
                overlap = self.ddp is not None and n_micro == 1 and self.ddp.overlap and self.ddp.world > 1
                plan.bucket_of_param = self._plan_buckets() if overlap else None
                plan.on_bucket = self.ddp.bucket_ready if overlap else None
            wait_for_batch(b)
            try:
                loss, parts, F_out = self.forward_loss(b, d, fused_total=(n_micro == 1))
                if n_micro > 1:
                    parts = tuple(p / n_micro for p in parts)         # :875-877
                    loss = self.pos_weight * parts[0] + self.finest_weight * parts[1] + self.neg_weight * parts[2]
                loss.backward()
            finally:
                if plan is not None:
                    plan.grad_targets = plan.bucket_of_param = plan.on_bucket = None
            if plan is not None and i > 0:
                self._flat_grad.add_(self._flat_grad2)
            release_batch(b)
            n_rows += F_out.shape[0]
            dl, dp = loss.detach(), tuple(p.detach() for p in parts)
            tot_loss = dl if tot_loss is None else tot_loss + dl
            tot_parts = dp if tot_parts is None else tuple(a + b_ for a, b_ in zip(tot_parts, dp))
        if self.ddp is not None:
            self.ddp.all_reduce_gradients()
        self.optimizer.step()
        return tot_loss, tot_parts, n_rows


def prefetch_to_device(batches, device, keys=("sinput_C", "sinput_F", "group", "index", "finest_flag"), ring=4):
    """What the reference's step does first -- ``input_dict[...].to(self.device)`` (lib/colocation_trainer.py:843-845) --
    as a loader-side prefetch: the tensors of batch i+1 are copied host -> device (pinned memory, non-blocking) on a
    COPY stream while the kernels of batch i run on the compute stream.  Yields dicts whose ``keys`` are device tensors
    (everything else passes through) plus ``"_h2d_event"``: ``train_step`` makes the compute stream wait for it when it
    first touches the batch (NOT when the batch is pulled: ``train_steps`` pulls one batch ahead of the step it
    enqueues, which is what gives the copy its head start).
    The device side is a pool of persistent staging slots (``ring`` to start with, grown on demand), not fresh
    allocations: a block that another stream has just used is not reusable by the caching allocator until its events
    have passed, so per-step allocations on the copy stream kept falling through to hipMalloc (~0.1 ms each on the
    enqueuing thread).  A slot is handed out again only after its consumer has called ``release_batch`` (the trainer
    does, once the batch's forward / backward is enqueued): that records an event on the compute stream which the copy
    stream waits for before overwriting the slot.  A slot that was handed out and not released is NEVER
    recycled -- the pool grows instead (``train_steps`` legitimately holds ``(GCL_PREFETCH_DEPTH + 1) * iter_size``
    unreleased batches; a consumer that never releases simply gets one slot per batch, i.e. per-batch allocations)."""
    dev = torch.device(device)
    copy_stream = torch.cuda.Stream(device=dev)
    slots = [{"bufs": {}, "busy": False, "free": None, "age": -1} for _ in range(max(2, int(ring)))]
    for i, b in enumerate(batches):
        out = dict(b)
        slot = next((s_ for s_ in slots if not s_["busy"]), None)
        if slot is None:          # every slot belongs to a batch that is still pending: grow, never overwrite
            slot = {"bufs": {}, "busy": False, "free": None, "age": -1}
            slots.append(slot)
        if slot["free"] is not None:
            copy_stream.wait_event(slot["free"])
        slot["busy"], slot["age"], slot["free"] = True, i, None
        with torch.cuda.stream(copy_stream):
            for k in keys:
                v = b.get(k)
                if isinstance(v, torch.Tensor) and not v.is_cuda:
                    if not v.is_pinned():
                        v = v.pin_memory()
                    n = v.numel()
                    buf = slot["bufs"].get(k)
                    if buf is None or buf.dtype != v.dtype or buf.numel() < n:
                        buf = slot["bufs"][k] = torch.empty(int(n * 1.25) + 16, dtype=v.dtype, device=dev)
                    dst = buf[:n].view(v.shape)
                    dst.copy_(v, non_blocking=True)
                    out[k] = dst
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        out["_h2d_event"], out["_h2d_slot"] = ev, (slot, i)
        yield out


def release_batch(batch):
    """The consumer is done ENQUEUING work that reads a prefetched batch: its staging slot may be overwritten once the
    current stream has passed this point (see ``prefetch_to_device``).  No-op for ordinary batches."""
    ar = batch.get("_map_arena") if isinstance(batch, dict) else None
    if ar is not None and ar["busy"]:      # the pooled arena of the batch's native maps (trainer._take_map_arena)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with ar["lock"]:
            ar["free"], ar["busy"] = ev, False
    st = batch.get("_h2d_slot") if isinstance(batch, dict) else None
    if st is None:
        return
    slot, age = st
    if slot["busy"] and slot["age"] == age:
        slot["free"] = torch.cuda.Event()
        slot["free"].record(torch.cuda.current_stream())
        slot["busy"] = False


def wait_for_batch(batch):
    """Orders the current stream behind the host -> device copy of a prefetched batch (no-op for ordinary batches)."""
    if not isinstance(batch, dict):
        return
    ev, mev = batch.get("_h2d_event"), batch.get("_maps_event")
    if ev is None and mev is None:
        return
    cur = torch.cuda.current_stream()
    if ev is not None:
        cur.wait_event(ev)
        for v in batch.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)          # allocated on the copy stream, consumed on the compute stream
    if mev is not None:                       # coordinate manager prefetched on the side stream
        cur.wait_event(mev)
        for t in batch["_coordinate_manager"].device_tensors() + list(batch.get("_loss_inputs") or ()):
            t.record_stream(cur)
