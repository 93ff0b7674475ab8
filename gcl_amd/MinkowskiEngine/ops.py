"""torch.autograd wrappers around the C-ABI kernels (include/gcl_amd.h).  GPU tensors only."""
import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

from .. import _lib

# Arithmetic of the MFMA convolution kernels (forward and input gradient):
#   "f32"    exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
#   "bf16x6" fp32 operands split into 3 bf16 planes, 6 MFMA terms, fp32 accumulate -- error vs the fp64 oracle equal
#            to native fp32 (tests/test_gpu_parity.py), 2.7x the MFMA rate
#   "bf16x3" 2 planes, 3 terms (~1.5e-5 relative), 5.3x the MFMA rate
#   "fp16x3" 2 fp16 planes of operands pre-scaled by a per-tensor power of two (max-abs), 3 terms: 24 significand
#            bits like fp32 (measured 0.9e-6 vs 1.15e-6 for exact f32), 5.3x the MFMA rate; costs one max-abs pass
#            per operand tensor  [default]
_PREC_CODES = {"f32": 0, "bf16x3": 2, "bf16x6": 3, "fp16x3": 4}
PRECISION = os.environ.get("GCL_CONV_PRECISION", "fp16x3")
if PRECISION not in _PREC_CODES:
    raise ValueError(f"GCL_CONV_PRECISION must be one of {sorted(_PREC_CODES)}")


def set_conv_precision(name):
    global PRECISION
    if name not in _PREC_CODES:
        raise ValueError(f"precision must be one of {sorted(_PREC_CODES)}")
    PRECISION = name


# bench.py sets this to a list to time individual launches with events on the launch stream:
# entries are (kernel name, start event, end event, pairs, cin, cout, n_in, n_out, K)
PROFILE = None


class _Timed:
    """Brackets one launch with events on torch's current stream (the stream the kernel is launched on)."""

    def __init__(self, name, pairs, cin, cout, n_in=0, n_out=0, K=1):
        self.rec = None
        if PROFILE is not None:
            self.rec = (name, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                        int(pairs), cin, cout, int(n_in), int(n_out), int(K))

    def __enter__(self):
        if self.rec is not None:
            self.rec[1].record()

    def __exit__(self, *exc):
        if self.rec is not None:
            self.rec[2].record()
            PROFILE.append(self.rec)


_LAST_BN_AMAX = None
_AMAX_POOL = {}      # device -> [zero-filled int32 [1024, 512], cursor]: slots are used once


AMAX_WORDS = 512        # GCL_AMAX_WORDS (include/gcl_amd.h): 16 entries on separate 128-byte lines


def amax_slot(device):
    """A zero-initialised amax slot (device int32[GCL_AMAX_WORDS], used once); one 2 MB fill per 1024 slots."""
    pool = _AMAX_POOL.get(device)
    if pool is None or pool[1] >= pool[0].shape[0]:
        pool = _AMAX_POOL[device] = [torch.zeros((1024, AMAX_WORDS), dtype=torch.int32, device=device), 0]
    out = pool[0][pool[1]]
    pool[1] += 1
    return out


def amax_value(slot):
    """The float held by an amax slot, as a device tensor (tests / diagnostics)."""
    return slot.view(-1, 32)[:, 0].max().view(1).view(torch.float32)


_AMAX_EPOCH = 0


def invalidate_amax():
    """Void every amax tag.  Needed only after writing a tagged tensor through an alias torch does not version
    together with it (e.g. updating FlatDDP.flat_param directly instead of stepping the parameters)."""
    global _AMAX_EPOCH
    _AMAX_EPOCH += 1


def tag_amax(t, slot):
    """Remember that ``slot`` holds gcl_amax of ``t`` as it is now (in-place writes bump _version and void the tag)."""
    t._gcl_amax = (slot, t._version, _AMAX_EPOCH)


def known_amax(t):
    tag = getattr(t, "_gcl_amax", None)
    return tag[0] if tag is not None and tag[1] == t._version and tag[2] == _AMAX_EPOCH else None


def tensor_amax(lib, t):
    """Device int32[1] holding the bit pattern of max|t| (only the fp16x3 mode needs it).  Producers that already
    stream the tensor (BatchNorm apply / backward apply) publish it themselves and tag the tensor; parameters of a
    WeightAmaxGroup are refreshed together in one launch."""
    out = known_amax(t)
    if out is not None:
        return out
    group = getattr(t, "_gcl_amax_group", None)
    if group is not None:
        return group.refresh(lib, t)
    out = amax_slot(t.device)
    _lib.check(lib.gcl_amax(_lib.ptr(t, torch.float32), t.numel(), _lib.ptr(out), 1, _lib.stream()), "gcl_amax")
    tag_amax(t, out)
    return out


PRESPLIT_MIN_C = int(os.environ.get("GCL_PRESPLIT_MIN_C", "128"))     # operands at least this wide are pre-split
BN_PLANES = os.environ.get("GCL_BN_PLANES", "1") != "0"                  # BatchNorm bound mode (csrc/plan.hip bound_mode)


def planes_of(lib, t, amax):
    """fp16x3 operand image of activation tensor ``t`` (gcl_split_planes), made once per tensor version and reused by
    every launch that consumes it (forward + weight gradient, or input gradient + weight gradient)."""
    tag = getattr(t, "_gcl_planes", None)
    if tag is not None and tag[1] == t._version and tag[2] == _AMAX_EPOCH and tag[3] is amax:
        return tag[0]
    n, c = t.shape
    planes = torch.empty((n, c), dtype=torch.int32, device=t.device)         # 4 bytes per element: hi | lo
    _lib.check(lib.gcl_split_planes(_lib.ptr(t, torch.float32), n, c, _lib.ptr(amax), _lib.ptr(planes),
                                    _lib.stream()), "gcl_split_planes")
    t._gcl_planes = (planes, t._version, _AMAX_EPOCH, amax)
    return planes


def _want_planes(c):
    return PRECISION == "fp16x3" and c >= PRESPLIT_MIN_C and c % 32 == 0


class WeightAmaxGroup:
    """All convolution kernels of a network, refreshed together once per optimizer step: gcl_amax of every tensor in
    ONE launch (gcl_amax_multi) and the MFMA-order weight packs of a direction in ONE launch each
    (gcl_pack_weights_multi) -- forward packs with the refresh, input-gradient packs at the first backward request."""

    def __init__(self, params):
        self.params = [p for p in params]
        for i, p in enumerate(self.params):
            p._gcl_amax_group = self
            p._gcl_group_index = i
        self.ptrs = None
        self.packs = {}          # direction ("fwd" / "bwd") -> (tag, [packed views])
        self.bwd_modes = {}      # param index -> 1 | 2, recorded by the convolutions during the forward pass

    def void_tags(self):
        for p in self.params:
            p._gcl_amax = None

    def _shape3(self, p):
        return tuple(p.shape) if p.dim() == 3 else (1,) + tuple(p.shape)

    def refresh(self, lib, want):
        ps = self.params
        dev = ps[0].device
        ptrs = [p.data_ptr() for p in ps]
        if self.ptrs != ptrs:
            if any((not p.is_contiguous()) or p.dtype != torch.float32 or p.device != dev for p in ps):
                raise ValueError("WeightAmaxGroup: parameters must be contiguous fp32 tensors on one device")
            self.ptrs = ptrs
            self.table = torch.tensor(ptrs + [p.numel() for p in ps], dtype=torch.int64).to(dev)
            self.out = torch.empty((len(ps), AMAX_WORDS), dtype=torch.int32, device=dev)
            self.desc = {}
        else:
            self.out = torch.empty_like(self.out)       # earlier slots may still be referenced by saved contexts
        n = len(ps)
        _lib.check(lib.gcl_amax_multi(_lib.ptr(self.table), ctypes_offset(self.table, n), n, _lib.ptr(self.out),
                                      _lib.stream()), "gcl_amax_multi")
        found = None
        for i, p in enumerate(ps):
            slot = self.out[i]
            tag_amax(p, slot)
            if p is want:
                found = slot
        self.packs = {}
        return found

    def _tag(self):
        return (tuple(p._version for p in self.params), _AMAX_EPOCH, PRECISION, self.out.data_ptr())

    def packed(self, lib, p, mode):
        """Packed weights of parameter ``p`` for ``mode`` (0 forward; 1 / 2 input gradient) from the group launch of
        this step, or None when the group cannot serve it (f32 arithmetic, mode not recorded, stale amax)."""
        prec = _PREC_CODES[PRECISION]
        if prec != 4 or known_amax(p) is None:        # the group launches serve the default arithmetic only
            return None
        key = "fwd" if mode == 0 else "bwd"
        i = p._gcl_group_index
        if key == "bwd" and self.bwd_modes.get(i) != mode:
            return None
        tag = self._tag()
        entry = self.packs.get(key)
        if entry is None or entry[0] != tag:
            idx = list(range(len(self.params))) if key == "fwd" else sorted(self.bwd_modes)
            modes = [0] * len(idx) if key == "fwd" else [self.bwd_modes[j] for j in idx]
            dkey = (key, tuple(idx), tuple(modes), prec)
            if dkey not in self.desc:
                rows, off, wg = [], 0, 0
                for j, m in zip(idx, modes):
                    K, ci, co = self._shape3(self.params[j])
                    rows.append([self.ptrs[j], K, ci, co, m, j, off, wg])
                    off += (lib.gcl_pack_weights_bytes(K, ci, co, prec) + 255) // 256 * 256
                    wg += (K * ci * co + 255) // 256
                offs = [r[6] for r in rows] + [off]
                self.desc[dkey] = (torch.tensor(rows, dtype=torch.int64).to(self.params[0].device), offs, wg)
            desc, offs, wgs = self.desc[dkey]
            buf = torch.empty(offs[-1], dtype=torch.uint8, device=self.params[0].device)
            _lib.check(lib.gcl_pack_weights_multi(_lib.ptr(desc), len(idx), wgs, prec, _lib.ptr(self.out),
                                                  _lib.ptr(buf), _lib.stream()), "gcl_pack_weights_multi")
            views = {j: buf[offs[q]:offs[q + 1]] for q, j in enumerate(idx)}
            entry = self.packs[key] = (tag, views)
        return entry[1].get(i)


def ctypes_offset(t, elem):
    return ctypes.c_void_p(t.data_ptr() + elem * t.element_size())


def _conv_launch(lib, x, Wk, mode, table, n_out, cin, cout, bias, pairs=0, want_stats=False, x_amax=None,
                 w_amax=None, wp=None, x_planes=None, generic=False, add=None):
    """One output-stationary convolution launch.  ``Wk`` [K, *, *] is packed for ``mode`` (0 forward, 1 transposed,
    2 transposed + mirrored offsets) in the current precision; ``table`` = (tbl, order, tile_mask) from
    KernelMap.sorted_table(), or None for a kernel_size-1 conv; (cin, cout) are the EFFECTIVE widths of the launch.
    ``want_stats``: also return the per-workgroup column sums [ceil(n_out/128), 2, cout] for a following BatchNorm.
    ``add`` [n_out, cout] (split-precision MFMA kernels only): added to the result in the epilogue (a gradient that
    already reached the same tensor through another path: saves the separate accumulation pass)."""
    prec = _PREC_CODES[PRECISION]
    K, wc_in, wc_out = Wk.shape
    if prec == 4 and not generic:
        x_amax = x_amax if x_amax is not None else tensor_amax(lib, x)
        w_amax = w_amax if w_amax is not None else tensor_amax(lib, Wk)
    if wp is None:      # not served by a WeightAmaxGroup launch: pack this tensor now
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, wc_in, wc_out, prec), dtype=torch.uint8, device=Wk.device)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(Wk, torch.float32), K, wc_in, wc_out, mode, prec, _lib.ptr(w_amax),
                                        _lib.ptr(wp), _lib.stream()), "gcl_pack_weights")
    tbl, order, tile_mask = table if table is not None else (None, None, None)
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    stats = None
    if want_stats and prec != 0:
        stats = torch.empty((4, cout, (n_out + 127) // 128), dtype=torch.float32, device=x.device)     # channel-major partials: sum, squares, min, max
    name = None
    if PROFILE is not None:
        nb = lib.gcl_conv_fwd_nb(n_out, cout, prec)
        pre = "true" if x_planes is not None else "false"
        name = "k_conv_generic" if generic else \
            (f"k_conv_fwd<{nb}>" if prec == 0 else
             f"k_conv_fwd_split<{nb},{prec},{pre},{'true' if add is not None else 'false'}>")
        if prec == 4 and not generic and os.environ.get("GCL_FWD_DMA", "1") != "0" and \
                (x_planes is not None or os.environ.get("GCL_FWD_DMA_ROWS", "1") != "0"):
            name = f"k_conv_fwd_dma<{nb},{pre},{'true' if add is not None else 'false'}>"
    with _Timed(name, pairs, cin, cout, x.shape[0], n_out, K):
        xin, is_planes = (x_planes, 1) if x_planes is not None else (x, 0)
        if add is not None:
            _lib.check(lib.gcl_conv_fwd_fused(_lib.ptr(xin), x.shape[0], is_planes, _lib.ptr(wp), prec, _lib.ptr(x_amax),
                                              _lib.ptr(w_amax), _lib.ptr(tbl), _lib.ptr(order), _lib.ptr(tile_mask), n_out,
                                              K, cin, cout, _lib.ptr(bias), None, _lib.ptr(add, torch.float32), 0, None,
                                              _lib.ptr(y), _lib.ptr(stats), getattr(tbl, "_gcl_flags", 0), _lib.stream()),
                       "gcl_conv_fwd_fused")
            return (y, stats) if want_stats else y
        _lib.check(lib.gcl_conv_fwd(_lib.ptr(xin), x.shape[0], is_planes, _lib.ptr(wp), prec, _lib.ptr(x_amax),
                                    _lib.ptr(w_amax), _lib.ptr(tbl), _lib.ptr(order), _lib.ptr(tile_mask), n_out, K,
                                    cin, cout, _lib.ptr(bias), _lib.ptr(y), _lib.ptr(stats),
                                    getattr(tbl, "_gcl_flags", 0), _lib.stream()),
                   "gcl_conv_fwd")
    return (y, stats) if want_stats else y


class _SparseConvFn(torch.autograd.Function):
    """y[v] = sum_k x[u(k, v)] W_k (+ bias) -- forward, input gradient and weight gradient all in HIP."""

    @staticmethod
    def forward(ctx, x, W, bias, kmap, n_out, transpose, mgr, want_stats):
        lib = _lib.require_gpu()
        stats = None
        x = x.contiguous()
        Wk = (W if W.dim() == 3 else W.unsqueeze(0)).contiguous()
        K, cin, cout = Wk.shape
        if x.shape[1] != cin:
            raise ValueError(f"feature width {x.shape[1]} != in_channels {cin}")
        # three kernel families behind the C ABI: the first-layer VALU kernels (Cin <= 4), the exact-fp32 VALU kernels
        # for generic shapes (any Cin / Cout, K <= 125) and the MFMA kernels (Cin, Cout multiples of 32, K <= 27)
        ctx.stem = cin <= 4 and cout % 32 == 0 and not transpose and kmap is not None and bias is None
        ctx.generic = generic = (not ctx.stem) and (cin % 32 != 0 or cout % 32 != 0 or K > 27)
        fp16x3 = _PREC_CODES[PRECISION] == 4 and not ctx.stem and not generic
        x_known = known_amax(x) if fp16x3 else None
        w_known = tensor_amax(lib, W) if (fp16x3 and W.is_contiguous()) else None
        ctx.x_amax = ctx.w_amax = None
        ctx.group = ctx.param = None
        if ctx.stem:
            y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
            _lib.check(lib.gcl_stem_fwd(_lib.ptr(x, torch.float32), _lib.ptr(Wk), _lib.ptr(kmap.nbr), n_out, K, cin,
                                        cout, _lib.ptr(y), None, None, _lib.stream()), "gcl_stem_fwd")
        else:
            tbl = None if kmap is None else kmap.sorted_table(transposed=transpose)
            b = bias.detach().contiguous().view(-1) if bias is not None else None
            # pair counts reach the host asynchronously; only the profiler needs them in the forward pass
            ctx.pairs = (kmap.n_pairs if kmap is not None else n_out) if PROFILE is not None else 0
            if fp16x3:
                ctx.x_amax = x_known if x_known is not None else tensor_amax(lib, x)
                ctx.w_amax = w_known if w_known is not None else tensor_amax(lib, Wk)
            group = getattr(W, "_gcl_amax_group", None) if (w_known is not None or generic) else None
            wp = None
            if group is not None:     # packed with all other kernels of the network; remember the backward layout
                group.bwd_modes[W._gcl_group_index] = 2 if (kmap is not None and not transpose and kmap.same_map) else 1
                wp = group.packed(lib, W, 0)
            ctx.group, ctx.param = group, (W if group is not None else None)
            xp = planes_of(lib, x, ctx.x_amax) if (fp16x3 and _want_planes(cin)) else None
            want_stats = want_stats and not generic
            y = _conv_launch(lib, x, Wk, 0, tbl, n_out, cin, cout, b, ctx.pairs, want_stats=want_stats,
                             x_amax=ctx.x_amax, w_amax=ctx.w_amax, wp=wp, x_planes=xp, generic=generic)
            if want_stats:
                y, stats = y
        ctx.save_for_backward(x, Wk)
        ctx.set_materialize_grads(False)       # no zero-filled gradient for the (non-differentiable) statistics output
        ctx.kmap, ctx.transpose, ctx.mgr, ctx.w_shape, ctx.has_bias = kmap, transpose, mgr, W.shape, bias is not None
        if stats is None:
            stats = y.new_empty(0)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dstats):
        lib = _lib.load()
        if dy is None:
            return (None,) * 8
        x, Wk = ctx.saved_tensors
        K, cin, cout = Wk.shape
        kmap, transpose, generic = ctx.kmap, ctx.transpose, ctx.generic
        prec = _PREC_CODES[PRECISION]
        fp16x3 = prec == 4 and not ctx.stem and not generic
        dy_amax = known_amax(dy) if fp16x3 else None
        dy = dy.contiguous()
        dx = dW = dbias = None
        if dy_amax is None and fp16x3:
            dy_amax = tensor_amax(lib, dy)
        x_amax = ctx.x_amax if (not fp16x3 or ctx.x_amax is not None) else tensor_amax(lib, x)
        w_amax = ctx.w_amax if (not fp16x3 or ctx.w_amax is not None) else tensor_amax(lib, Wk)
        if not ctx.stem and PROFILE is not None:
            ctx.pairs = kmap.n_pairs if kmap is not None else x.shape[0]
        if ctx.needs_input_grad[0]:
            if ctx.stem:
                raise NotImplementedError("input gradient of the Cin <= 4 first conv is not needed by the hot path")
            if kmap is None:
                mode, tbl = 1, None
            elif transpose:
                mode, tbl = 1, kmap.sorted_table(transposed=False)
            elif kmap.same_map:
                mode, tbl = 2, kmap.sorted_table(transposed=False)
            else:
                mode, tbl = 1, kmap.sorted_table(transposed=True)
            group = ctx.group
            wp = group.packed(lib, ctx.param, mode) if group is not None else None
            dyp = planes_of(lib, dy, dy_amax) if (fp16x3 and _want_planes(cout)) else None
            acc = getattr(ctx, "dx_accumulate", None)       # Tape: a gradient that already reached x through another path
            if acc is not None and (generic or prec == 0 or acc.shape != (x.shape[0], cin) or not acc.is_contiguous()
                                    or acc.dtype != torch.float32):
                acc = None
            ctx.dx_accumulated = acc is not None
            dx = _conv_launch(lib, dy, Wk, mode, tbl, x.shape[0], cout, cin, None, ctx.pairs, x_amax=dy_amax,
                              w_amax=w_amax, wp=wp, x_planes=dyp, generic=generic, add=acc)
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(Wk)
            if ctx.stem:
                n_out = dy.shape[0]
                scratch = torch.empty(lib.gcl_stem_bwd_weight_scratch_len(K, cin, cout, n_out), dtype=torch.float32,
                                      device=x.device)
                _lib.check(lib.gcl_stem_bwd_weight(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(kmap.nbr), n_out, K, cin, cout,
                                                   _lib.ptr(scratch), _lib.ptr(dW), None, None, _lib.stream()),
                           "gcl_stem_bwd_weight")
            elif kmap is None and fp16x3 and lib.gcl_conv_bwd_weight_rows_scratch_len(cin, cout, prec, x.shape[0]) > 0:
                # kernel_size 1: every row is its own pair -- both operands streamed once, no pair list (k_bwd_weight_rows)
                scratch = torch.empty(lib.gcl_conv_bwd_weight_rows_scratch_len(cin, cout, prec, x.shape[0]),
                                      dtype=torch.float32, device=x.device)
                with _Timed(f"k_bwd_weight_rows<{cin // 32},{cout // 32}>", ctx.pairs, cin, cout, x.shape[0], dy.shape[0], K):
                    _lib.check(lib.gcl_conv_bwd_weight_rows(_lib.ptr(x), _lib.ptr(dy), x.shape[0], cin, cout, prec,
                                                            _lib.ptr(x_amax), _lib.ptr(dy_amax), _lib.ptr(scratch),
                                                            _lib.ptr(dW), _lib.stream()), "gcl_conv_bwd_weight_rows")
            else:
                if kmap is None:
                    pa, pb, seg, seg_host = ctx.mgr.identity_pairs(x.shape[0])
                    sorted_side = 0
                else:
                    pin, pout, seg, seg_host = kmap.pairs()
                    pa, pb = (pout, pin) if transpose else (pin, pout)
                    sorted_side = 1 if transpose else 2       # the map's OUT rows ascend inside every offset segment
                n_sorted = (x.shape[0] if sorted_side == 1 else dy.shape[0]) if sorted_side else 0
                scratch = torch.empty(lib.gcl_conv_bwd_weight_scratch_len(K, cin, cout, seg[-1], n_sorted),
                                      dtype=torch.float32, device=x.device)
                tile = f"{64 if cin % 64 == 0 else 32},{64 if cout % 64 == 0 else 32}"
                use_pl = fp16x3 and _want_planes(cin) and _want_planes(cout)
                name = "k_conv_bwd_weight_generic" if (cin % 32 or cout % 32) else \
                    (f"k_conv_bwd_weight<{tile}>" if prec == 0 else
                     f"k_conv_bwd_weight_split<{tile},{prec},{'true' if use_pl else 'false'}>")
                if use_pl and cin % 128 == 0 and cout % 128 == 0 and os.environ.get("GCL_DW_WG128", "1") != "0":
                    name = "k_conv_bwd_weight_wg128"
                if prec == 4 and not fp16x3 and not (cin % 32 or cout % 32):     # K > 27 with MFMA-shaped channels
                    x_amax, dy_amax = tensor_amax(lib, x), tensor_amax(lib, dy)
                with _Timed(name, ctx.pairs, cin, cout, x.shape[0], dy.shape[0], K):
                    xa = planes_of(lib, x, x_amax) if use_pl else x
                    ya = planes_of(lib, dy, dy_amax) if use_pl else dy
                    _lib.check(lib.gcl_conv_bwd_weight(_lib.ptr(xa), x.shape[0], _lib.ptr(ya), dy.shape[0], int(use_pl),
                                                       sorted_side, _lib.ptr(pa), _lib.ptr(pb), seg_host,
                                                       K, cin, cout, prec, _lib.ptr(x_amax), _lib.ptr(dy_amax),
                                                       _lib.ptr(scratch), _lib.ptr(dW), _lib.stream()),
                               "gcl_conv_bwd_weight")
            dW = dW.view(ctx.w_shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = _col_sum(lib, dy)
        return dx, dW, dbias, None, None, None, None, None


def _col_sum(lib, dy):
    """``dy.sum(0, keepdim=True)`` by gcl_col_sum (ordered fp64 partials: deterministic, and the same launch the
    whole-network plan issues, so both paths stay bitwise equal); widths the kernel does not take use torch."""
    n, c = dy.shape
    if n == 0 or c < 4 or c % 4 or 256 % (c // 4) or dy.dtype != torch.float32:
        return dy.sum(0, keepdim=True)
    out = torch.empty((1, c), dtype=torch.float32, device=dy.device)
    scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=dy.device)
    _lib.check(lib.gcl_col_sum(_lib.ptr(dy, torch.float32), n, c, _lib.ptr(scratch), _lib.ptr(out), _lib.stream()),
               "gcl_col_sum")
    return out


class _TraceCtx:
    """What NetworkPlan.from_tape reads of a convolution entry (a stand-in for the autograd ctx of the training Tape)."""

    def __init__(self, kmap, transpose, W):
        Wk = W if W.dim() == 3 else W.unsqueeze(0)
        K, cin, cout = Wk.shape
        self.kmap, self.transpose = kmap, bool(transpose)
        self.stem = cin <= 4 and cout % 32 == 0 and not transpose and kmap is not None
        self.generic = (not self.stem) and (cin % 32 != 0 or cout % 32 != 0 or K > 27)


# Inference trace: the layer calls of ONE eval-mode forward pass (model.eval(), torch.no_grad()) recorded in the Tape's
# entry format, so that NetworkPlan.from_tape can turn them into the operator records of gcl_plan_forward_eval.  Nothing
# is replayed from it (there is no backward pass); ME.conv_bn adds the "convbn" entries itself.
_EVAL_TRACE = None
CONV_TALL = 4      # include/gcl_amd.h GCL_CONV_TALL: inference launches (conv_bn_eval here, the plan's eval records)
GROUP_LAUNCHES = True      # conv_bn_eval hands the offset-group launches their scratch (tests switch it off: the sixteen-wave kernel)


def sparse_conv(x, W, kmap, n_out, transpose, bias, mgr, want_stats=False):
    """Returns (y, stats): ``stats`` = per-tile column sums for a following BatchNorm (None unless requested and
    available in the current precision)."""
    if _EVAL_TRACE is not None and not torch.is_grad_enabled():
        y, stats = _SparseConvFn.apply(x, W, bias, kmap, n_out, transpose, mgr, want_stats)
        _EVAL_TRACE.add("conv", y, _TraceCtx(kmap, transpose, W), x, (W, bias))
        return y, (stats if stats.numel() else None)
    if _TAPE is not None:
        with torch.no_grad():
            c1 = _SubCtx()
            y, stats = _SparseConvFn.forward(c1, x, W, bias, kmap, n_out, transpose, mgr, want_stats)
        _TAPE.add("conv", y, c1, x, (W, bias))
        return y, (stats if stats.numel() else None)
    y, stats = _SparseConvFn.apply(x, W, bias, kmap, n_out, transpose, mgr, want_stats)
    return y, (stats if stats.numel() else None)


class _BatchNormFn(torch.autograd.Function):
    """y = BN(x) (+ residual) (relu): statistics pass + fused apply pass; backward = reduce + fused apply."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, residual, relu, tile_stats):
        lib = _lib.require_gpu()
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        bound, bound_slot, xrange = False, None, None
        if training:
            mr = torch.empty((2, c), dtype=torch.float32, device=dev)          # one allocation: mean | rstd
            mean, rstd = mr[0], mr[1]
            if tile_stats is not None:      # column sums (and ranges) already produced by the convolution epilogue
                nt = tile_stats.shape[2]
                # bound mode (csrc/plan.hip GCL_OP_CONVBN): an output at least PRESPLIT_MIN_C channels wide is consumed as a
                # plane image, whose scale the native plan fixes BEFORE the apply pass from a bound of max|y| -- the same
                # launch with the same arguments here, so that both paths hand their consumers the same slot value
                res_amax = known_amax(residual) if residual is not None else None
                bound = (BN_PLANES and PRECISION == "fp16x3" and c >= PRESPLIT_MIN_C and c % 32 == 0
                         and (residual is None or res_amax is not None))
                if bound:
                    bound_slot = amax_slot(dev)
                    xrange = torch.empty((2, c), dtype=torch.float32, device=dev)
                _lib.check(lib.gcl_bn_stats_from_tiles_range(
                    _lib.ptr(tile_stats, torch.float32), nt, n, c, float(eps), float(momentum), _lib.ptr(running_mean),
                    _lib.ptr(running_var), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(xrange), _lib.ptr(weight.detach()),
                    _lib.ptr(bias.detach()), int(relu), _lib.ptr(res_amax) if bound else None, None, _lib.ptr(bound_slot),
                    _lib.stream()), "gcl_bn_stats_from_tiles_range")
            else:
                scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=dev)
                _lib.check(lib.gcl_bn_stats(_lib.ptr(x, torch.float32), n, c, float(eps), float(momentum),
                                            _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(scratch),
                                            _lib.ptr(mean), _lib.ptr(rstd), _lib.stream()), "gcl_bn_stats")
        else:
            mean = running_mean.detach().contiguous()
            rstd = torch.rsqrt(running_var.detach() + eps).contiguous()
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        global _LAST_BN_AMAX
        # bound mode: the slot already holds the bound (the apply pass's own, smaller maximum changes nothing in it)
        _LAST_BN_AMAX = slot = bound_slot if bound else (amax_slot(dev) if PRECISION == "fp16x3" else None)
        # with relu the sign bits of y (1 bit / element) are kept for the backward pass instead of y itself
        mask = torch.empty(lib.gcl_bn_mask_len(n, c), dtype=torch.int64, device=dev) if relu else None
        _lib.check(lib.gcl_bn_apply(_lib.ptr(x), n, c, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(weight.detach()),
                                    _lib.ptr(bias.detach()), _lib.ptr(res), int(relu), _lib.ptr(y), _lib.ptr(mask),
                                    _lib.ptr(slot), _lib.stream()), "gcl_bn_apply")
        ctx.save_for_backward(x, mask, weight, mean, rstd)
        ctx.relu, ctx.training, ctx.has_res = bool(relu), bool(training), residual is not None
        ctx.xrange = xrange
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        x, mask, weight, mean, rstd = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        dy = dy.contiguous()
        sums = torch.empty((2, c), dtype=torch.float32, device=dev)           # one allocation: sum_g | sum_gx
        sum_g, sum_gx = sums[0], sums[1]
        scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=dev)
        slot = amax_slot(dev) if PRECISION == "fp16x3" else None
        # bound mode (as csrc/plan.hip): max|dx| bounded by the reduce launch; the apply pass publishes nothing larger
        xrange = getattr(ctx, "xrange", None)
        bound = xrange is not None and slot is not None and ctx.training and c >= PRESPLIT_MIN_C
        _lib.check(lib.gcl_bn_bwd_reduce_range(_lib.ptr(x), _lib.ptr(dy), 0, None, _lib.ptr(mask), n, c, _lib.ptr(mean),
                                               _lib.ptr(rstd), int(ctx.relu), _lib.ptr(scratch), _lib.ptr(sum_g),
                                               _lib.ptr(sum_gx), _lib.ptr(xrange), _lib.ptr(weight.detach()),
                                               _lib.ptr(slot) if bound else None, _lib.stream()), "gcl_bn_bwd_reduce_range")
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        if ctx.training:
            sg, sx = sum_g, sum_gx
        else:                       # running statistics are constants: no batch-statistics terms
            sg = sx = torch.zeros(c, dtype=torch.float32, device=dev)
        _lib.check(lib.gcl_bn_bwd_apply(_lib.ptr(x), _lib.ptr(dy), None, _lib.ptr(mask), n, c, _lib.ptr(mean), _lib.ptr(rstd),
                                        _lib.ptr(weight.detach()), _lib.ptr(sg), _lib.ptr(sx), int(ctx.relu),
                                        _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(slot), _lib.stream()),
                   "gcl_bn_bwd_apply")
        if slot is not None:
            tag_amax(dx, slot)
        return dx, sum_gx, sum_g, None, None, None, None, None, dres, None, None


def tape_guard(what, *tensors):
    """While a Tape is active only the tape-aware entry points may consume a tensor the tape produced: anything else
    (a stand-alone BatchNorm after a frozen / biased convolution, ``a + b`` on SparseTensors, ...) would put an
    ordinary autograd node behind a tensor that has no autograd history, and Tape.backward would silently drop the
    gradient of everything upstream.  Fail loudly instead; ``GCL_TAPE=0`` (or eval-mode submodules, which
    ResUNet2.forward detects itself) selects the per-layer autograd path."""
    if _TAPE is not None and any(t is not None and id(t) in _TAPE.made for t in tensors):
        raise RuntimeError(f"{what} consumed a tensor recorded by the whole-network Tape, which only knows conv_bn / "
                           "sparse_conv / relu / cat / l2_normalize_rows; run this model with GCL_TAPE=0 "
                           "(gcl_amd.MinkowskiEngine.ops.TAPE_ENABLED = False)")


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum, eps, residual=None, relu=False,
               tile_stats=None):
    tape_guard("MinkowskiBatchNorm", x, residual)
    y = _BatchNormFn.apply(x, weight, bias, running_mean, running_var, training, momentum, eps, residual, relu,
                           tile_stats)
    if _LAST_BN_AMAX is not None:
        tag_amax(y, _LAST_BN_AMAX)
    return y


class _SubCtx:
    """Stand-in for an autograd ctx when one Function runs the forward / backward of another as a sub-step."""
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


class _ConvBNFn(torch.autograd.Function):
    """``norm(conv(x))`` (+ residual)(+ ReLU) of a bias-free convolution and a BatchNorm as ONE autograd node: the same
    kernel launches as _SparseConvFn followed by _BatchNormFn (their forward / backward bodies are run as sub-steps),
    half the autograd nodes and wrapper objects per layer -- the training step is bound by the host on slower hosts."""

    @staticmethod
    def forward(ctx, x, W, bn_w, bn_b, residual, running_mean, running_var, kmap, n_out, transpose, mgr, momentum, eps,
                relu, want_stats):
        c1, c2 = _SubCtx(), _SubCtx()
        y, stats = _SparseConvFn.forward(c1, x, W, None, kmap, n_out, transpose, mgr, want_stats)
        z = _BatchNormFn.forward(c2, y, bn_w, bn_b, running_mean, running_var, True, momentum, eps, residual, relu,
                                 stats if stats.numel() else None)
        ctx.c1, ctx.c2 = c1, c2
        ctx.amax = _LAST_BN_AMAX
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        c1, c2 = ctx.c1, ctx.c2
        g = _BatchNormFn.backward(c2, dz.contiguous())
        dy, d_w, d_b, dres = g[0], g[1], g[2], g[8]
        c1.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], False)
        dx, dW = _SparseConvFn.backward(c1, dy, None)[:2]
        ctx.c1 = ctx.c2 = None
        return dx, dW, d_w, d_b, dres, None, None, None, None, None, None, None, None, None, None


class Tape:
    """The layer calls of ONE training forward pass of a network, recorded so that the whole network becomes a single
    autograd node (``with ops.tape() as t: ...; out = t.finish(out)``): while a tape is active, ``conv_bn_train``,
    ``sparse_conv``, ``relu``, ``cat`` and ``l2_normalize_rows`` run their forward bodies without autograd and append
    an entry; ``finish`` wraps the result in ``_TapeFn``, whose backward walks the entries in reverse with the same
    backward bodies autograd would have called, in the same order.  Same launches, same arithmetic -- what disappears is
    the per-node cost of ~50 autograd nodes per step (engine dispatch, the backward thread hand-over, wrapper objects),
    which matters because the step is bound by the host's launch rate.  Only for graphs made of exactly these ops
    (gcl_amd.model.ResUNet2 with BatchNorm); anything else keeps the ordinary autograd path."""

    def __init__(self):
        self.entries = []
        self.made = set()          # id() of tensors produced inside the tape (their producers want an input gradient)

    def add(self, kind, out, *payload):
        self.entries.append((kind, out) + payload)
        self.made.add(id(out))
        return out

    def params(self):
        seen, out = set(), []
        for e in self.entries:
            for t in e[-1]:
                if t is not None and id(t) not in seen:
                    seen.add(id(t))
                    out.append(t)
        return out

    def finish(self, out):
        global _TAPE
        _TAPE = None
        ps = self.params()
        return _TapeFn.apply(self, out, *ps)

    def backward(self, dout, out):
        grads, pgrads = {id(out): dout}, {}

        def give(t, g):
            if g is None or t is None or id(t) not in self.made:
                return
            k = id(t)
            grads[k] = g if k not in grads else grads[k] + g

        def pgive(p, g):
            if p is not None and g is not None:
                k = id(p)
                pgrads[k] = g if k not in pgrads else pgrads[k] + g

        for e in reversed(self.entries):
            kind, y = e[0], e[1]
            g = grads.pop(id(y), None)
            if g is None:
                continue
            if kind == "convbn":
                c1, c2, x, res, (W, bw, bb) = e[2], e[3], e[4], e[5], e[6]
                r = _BatchNormFn.backward(c2, g.contiguous())
                c1.needs_input_grad = (id(x) in self.made, True, False)
                c1.dx_accumulate = grads.get(id(x)) if FUSE_GRAD_ADD else None
                dx, dW = _SparseConvFn.backward(c1, r[0], None)[:2]
                if getattr(c1, "dx_accumulated", False):
                    grads[id(x)] = dx                    # the launch's epilogue added the gradient that was waiting
                else:
                    give(x, dx)
                give(res, r[8])
                pgive(W, dW)
                pgive(bw, r[1])
                pgive(bb, r[2])
            elif kind == "conv":
                c1, x, (W, b) = e[2], e[3], e[4]
                c1.needs_input_grad = (id(x) in self.made, True, b is not None)
                c1.dx_accumulate = grads.get(id(x)) if FUSE_GRAD_ADD else None
                dx, dW, db = _SparseConvFn.backward(c1, g, None)[:3]
                if getattr(c1, "dx_accumulated", False):
                    grads[id(x)] = dx
                else:
                    give(x, dx)
                pgive(W, dW)
                pgive(b, db)
            elif kind == "relu":
                give(e[2], torch.ops.aten.threshold_backward(g, y, 0))
            elif kind == "cat":
                off = 0
                for t in e[2]:
                    give(t, g[:, off:off + t.shape[1]])
                    off += t.shape[1]
            elif kind == "rownorm":
                give(e[3], _RowNormalizeFn.backward(e[2], g))
        self.entries = None
        return pgrads


_TAPE = None
TAPE_ENABLED = os.environ.get("GCL_TAPE", "1") == "1"
# Tape backward: a convolution's input gradient lands on a tensor that already holds a gradient from another path (the
# residual branch of a block, a skip connection) -> the waiting gradient is added in the launch's epilogue
FUSE_GRAD_ADD = os.environ.get("GCL_FUSE_GRAD_ADD", "1") == "1"


class tape:
    """Context manager: ``with ops.tape() as t`` activates a Tape for the layer calls inside (training only)."""

    def __enter__(self):
        global _TAPE
        _TAPE = Tape()
        return _TAPE

    def __exit__(self, *exc):
        global _TAPE
        _TAPE = None
        return False


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tp, out, *params):
        ctx.tp, ctx.out, ctx.n = tp, out, len(params)
        ctx.pids = [id(p) for p in params]
        return out.view_as(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        pg = ctx.tp.backward(dout, ctx.out)
        ctx.tp = ctx.out = None
        return (None, None) + tuple(pg.get(k) for k in ctx.pids)


def conv_bn_train(x, W, kmap, n_out, transpose, mgr, bn_w, bn_b, running_mean, running_var, momentum, eps, residual, relu,
                  want_stats, bn_module=None):
    if _TAPE is not None:
        with torch.no_grad():
            c1, c2 = _SubCtx(), _SubCtx()
            y, stats = _SparseConvFn.forward(c1, x, W, None, kmap, n_out, transpose, mgr, want_stats)
            z = _BatchNormFn.forward(c2, y, bn_w, bn_b, running_mean, running_var, True, momentum, eps, residual, relu,
                                     stats if stats.numel() else None)
        c2.bn_extra = (running_mean, running_var, momentum, eps, bn_module)     # what a NetworkPlan record needs
        _TAPE.add("convbn", z, c1, c2, x, residual, (W, bn_w, bn_b))
    else:
        z = _ConvBNFn.apply(x, W, bn_w, bn_b, residual, running_mean, running_var, kmap, n_out, transpose, mgr, momentum,
                            eps, relu, want_stats)
    if _LAST_BN_AMAX is not None:
        tag_amax(z, _LAST_BN_AMAX)
    return z


def relu(x):
    """ReLU of a feature matrix (tape-aware; MEF.relu)."""
    if _EVAL_TRACE is not None and id(x) in _EVAL_TRACE.made:
        return _EVAL_TRACE.add("relu", torch.relu(x), x, ())
    if _TAPE is not None and id(x) in _TAPE.made:
        with torch.no_grad():
            y = torch.relu(x)
        return _TAPE.add("relu", y, x, ())
    return torch.relu(x)


def cat_features(tensors):
    """Channel concatenation of feature matrices (tape-aware; ME.cat)."""
    if _EVAL_TRACE is not None and any(id(t) in _EVAL_TRACE.made for t in tensors):
        return _EVAL_TRACE.add("cat", torch.cat(tensors, dim=1), tuple(tensors), ())
    if _TAPE is not None and any(id(t) in _TAPE.made for t in tensors):
        with torch.no_grad():
            y = torch.cat(tensors, dim=1)
            tags = [known_amax(t) for t in tensors]
            if PRECISION == "fp16x3" and all(t is not None for t in tags):
                # max|cat| = max over the inputs' slots -- what the native plan's in-place cat holds in the shared slot
                # (a slot of a BatchNorm in bound mode carries its bound, not the measured maximum: csrc/plan.hip)
                slot = tags[0]
                for t in tags[1:]:
                    slot = torch.maximum(slot, t)
                tag_amax(y, slot)
        return _TAPE.add("cat", y, tuple(tensors), ())
    return torch.cat(tensors, dim=1)


class _RowNormalizeFn(torch.autograd.Function):
    """y = x / ||x||_2 per row (model/resunet.py:226-230) in one pass; backward in one pass."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.require_gpu()
        x = x.contiguous()
        n, c = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(n, dtype=torch.float32, device=x.device)
        _lib.check(lib.gcl_row_normalize_fwd(_lib.ptr(x, torch.float32), n, c, _lib.ptr(y), _lib.ptr(norm),
                                             _lib.stream()), "gcl_row_normalize_fwd")
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        y, norm = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        slot = amax_slot(y.device) if PRECISION == "fp16x3" else None       # the consumer is `final`'s input gradient
        _lib.check(lib.gcl_row_normalize_bwd(_lib.ptr(y), _lib.ptr(dy, torch.float32), _lib.ptr(norm), y.shape[0],
                                             y.shape[1], _lib.ptr(dx), _lib.ptr(slot), _lib.stream()),
                   "gcl_row_normalize_bwd")
        if slot is not None:
            tag_amax(dx, slot)
        return dx


def l2_normalize_rows(x):
    """``x / torch.norm(x, p=2, dim=1, keepdim=True)``; widths the kernel does not cover use that expression."""
    c = x.shape[1]
    if x.dim() == 2 and x.shape[0] > 0 and 4 <= c <= 256 and (c & (c - 1)) == 0 and x.dtype == torch.float32:
        if _EVAL_TRACE is not None and id(x) in _EVAL_TRACE.made:
            return _EVAL_TRACE.add("rownorm", _RowNormalizeFn.apply(x), None, x, ())
        if _TAPE is not None and id(x) in _TAPE.made:
            with torch.no_grad():
                cx = _SubCtx()
                y = _RowNormalizeFn.forward(cx, x)
            return _TAPE.add("rownorm", y, cx, x, ())
        return _RowNormalizeFn.apply(x)
    return x / torch.norm(x, p=2, dim=1, keepdim=True)


class _InstanceNormFn(torch.autograd.Function):
    """ME.MinkowskiInstanceNorm: per cloud (row segment) and channel, (x - mean) / sqrt(var + eps) with the biased
    variance, then the shared affine map; (+ residual)(relu) fused like BatchNorm.  Every segment runs the BatchNorm
    kernels on its row range."""

    @staticmethod
    def forward(ctx, x, weight, bias, segments, eps, residual, relu):
        lib = _lib.require_gpu()
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        ns = len(segments)
        mean = torch.empty((ns, c), dtype=torch.float32, device=dev)
        rstd = torch.empty((ns, c), dtype=torch.float32, device=dev)
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        w, b = weight.detach().contiguous().view(-1), bias.detach().contiguous().view(-1)
        global _LAST_BN_AMAX
        _LAST_BN_AMAX = slot = amax_slot(dev) if PRECISION == "fp16x3" else None
        masks = []
        st = _lib.stream()
        for s, (r0, rows) in enumerate(segments):
            xs = x[r0:r0 + rows]
            scratch = torch.empty(lib.gcl_bn_scratch_len(rows, c), dtype=torch.float64, device=dev)
            _lib.check(lib.gcl_bn_stats(_lib.ptr(xs, torch.float32), rows, c, float(eps), 0.0, None, None,
                                        _lib.ptr(scratch), _lib.ptr(mean[s]), _lib.ptr(rstd[s]), st), "gcl_bn_stats")
            mask = torch.empty(lib.gcl_bn_mask_len(rows, c), dtype=torch.int64, device=dev) if relu else None
            masks.append(mask)
            _lib.check(lib.gcl_bn_apply(_lib.ptr(xs), rows, c, _lib.ptr(mean[s]), _lib.ptr(rstd[s]), _lib.ptr(w),
                                        _lib.ptr(b), _lib.ptr(res[r0:r0 + rows]) if res is not None else None,
                                        int(relu), _lib.ptr(y[r0:r0 + rows]), _lib.ptr(mask), _lib.ptr(slot), st),
                       "gcl_bn_apply")
        ctx.save_for_backward(x, weight, mean, rstd, *[m for m in masks if m is not None])
        ctx.segments, ctx.relu, ctx.has_res = segments, bool(relu), residual is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, mean, rstd, *masks = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        dy = dy.contiguous()
        ns = len(ctx.segments)
        sum_g = torch.empty((ns, c), dtype=torch.float32, device=dev)
        sum_gx = torch.empty((ns, c), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        w = weight.detach().contiguous().view(-1)
        slot = amax_slot(dev) if PRECISION == "fp16x3" else None
        st = _lib.stream()
        for s, (r0, rows) in enumerate(ctx.segments):
            xs, gs = x[r0:r0 + rows], dy[r0:r0 + rows]
            mask = masks[s] if ctx.relu else None
            scratch = torch.empty(lib.gcl_bn_scratch_len(rows, c), dtype=torch.float64, device=dev)
            _lib.check(lib.gcl_bn_bwd_reduce(_lib.ptr(xs), _lib.ptr(gs), None, _lib.ptr(mask), rows, c,
                                             _lib.ptr(mean[s]), _lib.ptr(rstd[s]), int(ctx.relu), _lib.ptr(scratch),
                                             _lib.ptr(sum_g[s]), _lib.ptr(sum_gx[s]), st), "gcl_bn_bwd_reduce")
            _lib.check(lib.gcl_bn_bwd_apply(_lib.ptr(xs), _lib.ptr(gs), None, _lib.ptr(mask), rows, c,
                                            _lib.ptr(mean[s]), _lib.ptr(rstd[s]), _lib.ptr(w), _lib.ptr(sum_g[s]),
                                            _lib.ptr(sum_gx[s]), int(ctx.relu), _lib.ptr(dx[r0:r0 + rows]),
                                            _lib.ptr(dres[r0:r0 + rows]) if dres is not None else None,
                                            _lib.ptr(slot), st), "gcl_bn_bwd_apply")
        if slot is not None:
            tag_amax(dx, slot)
        return dx, sum_gx.sum(0).view_as(weight), sum_g.sum(0).view_as(weight), None, None, dres, None


def instance_norm(x, weight, bias, segments, eps=1e-8, residual=None, relu=False):
    tape_guard("MinkowskiInstanceNorm", x, residual)
    y = _InstanceNormFn.apply(x, weight, bias, segments, eps, residual, relu)
    if _LAST_BN_AMAX is not None:
        tag_amax(y, _LAST_BN_AMAX)
    return y


def conv_bn_eval(x, W, kmap, n_out, transpose, scale, shift, residual=None, relu=False):
    """Inference: convolution + BatchNorm (running statistics, as per-column ``scale`` / ``shift``) + residual add + ReLU
    in one launch (gcl_conv_fwd_fused).  No autograd graph: callers use it under torch.no_grad() only."""
    lib = _lib.require_gpu()
    if PRECISION != "fp16x3":
        raise RuntimeError("conv_bn_eval is built for the default fp16x3 arithmetic")
    x = x.contiguous()
    Wk = (W if W.dim() == 3 else W.unsqueeze(0)).contiguous()
    K, cin, cout = Wk.shape
    x_amax = tensor_amax(lib, x)
    w_amax = tensor_amax(lib, W if W.is_contiguous() else Wk)
    group = getattr(W, "_gcl_amax_group", None)
    wp = group.packed(lib, W, 0) if group is not None else None
    prec = 4
    if wp is None:
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, prec), dtype=torch.uint8, device=x.device)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(Wk, torch.float32), K, cin, cout, 0, prec, _lib.ptr(w_amax),
                                        _lib.ptr(wp), _lib.stream()), "gcl_pack_weights")
    tbl, order, tile_mask = kmap.sorted_table(transposed=transpose) if kmap is not None else (None, None, None)
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    slot = amax_slot(x.device)
    res = residual.contiguous() if residual is not None else None
    # scratch of the offset-group launches (the small deep layers of a pass; 0 floats: the shape / size takes another kernel)
    gs_len = lib.gcl_conv_fwd_groups_scratch_len(n_out, K, cin, cout) if (tbl is not None and GROUP_LAUNCHES) else 0
    gscratch = torch.empty(gs_len, dtype=torch.float32, device=x.device) if gs_len > 0 else None
    _lib.check(lib.gcl_conv_fwd_fused(_lib.ptr(x, torch.float32), x.shape[0], 0, _lib.ptr(wp), prec, _lib.ptr(x_amax),
                                      _lib.ptr(w_amax), _lib.ptr(tbl), _lib.ptr(order), _lib.ptr(tile_mask), n_out, K,
                                      cin, cout, _lib.ptr(shift, torch.float32), _lib.ptr(scale, torch.float32),
                                      _lib.ptr(res), int(relu), _lib.ptr(slot), _lib.ptr(y), _lib.ptr(gscratch),
                                      getattr(tbl, "_gcl_flags", 0) | CONV_TALL, _lib.stream()),
               "gcl_conv_fwd_fused")
    tag_amax(y, slot)
    return y
