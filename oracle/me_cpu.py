"""CPU ORACLE (test infrastructure, NOT product code): ctypes front end of oracle/me_cpu.c, the C / OpenMP restatement
of MinkowskiEngine's CPU sparse-convolution path (hash map -> per-offset kernel map -> gather / GEMM / scatter-add).

``CoordinateManager`` here has the interface of ``me_oracle.CoordinateManager``; its kernel maps carry a ``conv``
method, which ``me_oracle.sparse_conv`` dispatches to, so ``me_oracle.resunet_forward(..., mgr=me_cpu.CoordinateManager(C))``
runs the whole network with the convolutions (forward, input gradient, weight gradient) in C and everything else
(BatchNorm, ReLU, cat, loss) in torch-CPU fp32.  Pinned against me_oracle's torch path (tests/test_oracle_conv.py), which
is pinned against dense conv3d.  bench.py's ``cpu_baseline`` times exactly this on the GPU box's host cores.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "libme_cpu.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libme_cpu.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "me_cpu.c")):
            build()
        L = ctypes.CDLL(path)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        L.me_map_build.restype = i64
        L.me_map_build.argtypes = [vp, i64, vp, vp, i64]
        L.me_stride_coords.restype = i64
        L.me_stride_coords.argtypes = [vp, i64, i32, vp, vp, vp, i64]
        L.me_kernel_map.restype = None
        L.me_kernel_map.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, vp, vp]
        L.me_conv_apply.restype = None
        L.me_conv_apply.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, i64, vp]
        L.me_conv_grad_weight.restype = None
        L.me_conv_grad_weight.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, i64, vp]
        L.me_num_threads.restype = i32
        L.me_set_num_threads.argtypes = [i32]
        _LIB = L
    return _LIB


def _p(a):
    return ctypes.c_void_p(a.ctypes.data if isinstance(a, np.ndarray) else a.data_ptr())


def _cap(n):
    cap = 64
    while cap < 2 * n:
        cap *= 2
    return cap


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, km, n_out, transpose):
        L = lib()
        x = x.contiguous().float()
        Wk = (W if W.dim() == 3 else W.unsqueeze(0)).contiguous().float()
        K, cin, cout = Wk.shape
        src, dst = (km.out_rows, km.in_rows) if transpose else (km.in_rows, km.out_rows)
        y = torch.zeros((n_out, cout), dtype=torch.float32)
        L.me_conv_apply(_p(x), cin, _p(Wk), cout, K, _p(src), _p(dst), _p(km.counts), km.seg, _p(y))
        ctx.save_for_backward(x, Wk)
        ctx.km, ctx.transpose, ctx.w_shape = km, transpose, W.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        x, Wk = ctx.saved_tensors
        K, cin, cout = Wk.shape
        km = ctx.km
        src, dst = (km.out_rows, km.in_rows) if ctx.transpose else (km.in_rows, km.out_rows)
        dy = dy.contiguous().float()
        dx = dW = None
        if ctx.needs_input_grad[0]:
            Wt = Wk.transpose(1, 2).contiguous()
            dx = torch.zeros_like(x)
            L.me_conv_apply(_p(dy), cout, _p(Wt), cin, K, _p(dst), _p(src), _p(km.counts), km.seg, _p(dx))
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(Wk)
            L.me_conv_grad_weight(_p(x), cin, _p(dy), cout, K, _p(src), _p(dst), _p(km.counts), km.seg, _p(dW))
            dW = dW.view(ctx.w_shape)
        return dx, dW, None, None, None


class KernelMap:
    """Per-offset compacted (in, out) row lists: segment k = rows [k * seg, k * seg + counts[k])."""

    def __init__(self, in_rows, out_rows, counts, seg):
        self.in_rows, self.out_rows, self.counts, self.seg = in_rows, out_rows, counts, int(seg)

    def conv(self, x, W, n_out, transpose=False, bias=None):
        y = _ConvFn.apply(x, W, self, n_out, transpose)
        return y + bias.to(y.dtype) if bias is not None else y

    def triples(self):
        out = []
        for k, c in enumerate(self.counts):
            a = slice(k * self.seg, k * self.seg + int(c))
            out.append(np.stack([np.full(int(c), k, np.int64), self.in_rows[a].astype(np.int64),
                                 self.out_rows[a].astype(np.int64)], 1))
        return np.concatenate(out) if out else np.zeros((0, 3), np.int64)


class CoordinateManager:
    """Hash maps per tensor stride and kernel maps per (t_in, ks, stride), all built by the C code."""

    def __init__(self, C):
        C = np.ascontiguousarray(np.asarray(C, dtype=np.int32))
        n = len(C)
        cap = _cap(n)
        keys, vals = np.empty(cap, np.uint64), np.empty(cap, np.int32)
        got = lib().me_map_build(_p(C), n, _p(keys), _p(vals), cap)
        if got < 0:
            raise ValueError("coordinate outside the packable range")
        if got != n:
            raise ValueError("duplicate coordinates")
        self.maps = {1: (C, keys, vals, cap)}
        self.kmaps = {}

    def get_coords(self, t):
        if t not in self.maps:
            base_t = max(s for s in self.maps if s < t)
            Cb = self.get_coords(base_t)
            cap = _cap(len(Cb))
            keys, vals = np.empty(cap, np.uint64), np.empty(cap, np.int32)
            out = np.empty((len(Cb), 4), np.int32)
            m = lib().me_stride_coords(_p(Cb), len(Cb), t, _p(out), _p(keys), _p(vals), cap)
            self.maps[t] = (np.ascontiguousarray(out[:m]), keys, vals, cap)
        return self.maps[t][0]

    def get_kernel_map(self, t_in, ks, stride):
        key = (t_in, ks, stride)
        if key not in self.kmaps:
            self.get_coords(t_in)
            C_out = self.get_coords(t_in * stride)
            _, keys, vals, cap = self.maps[t_in]
            K, n_out = ks ** 3, len(C_out)
            in_rows, out_rows = np.empty(K * n_out, np.int32), np.empty(K * n_out, np.int32)
            counts = np.zeros(K, np.int64)
            lib().me_kernel_map(_p(C_out), n_out, _p(keys), _p(vals), cap, ks, t_in, _p(in_rows), _p(out_rows), _p(counts))
            self.kmaps[key] = KernelMap(in_rows, out_rows, counts, n_out)
        return self.kmaps[key]
