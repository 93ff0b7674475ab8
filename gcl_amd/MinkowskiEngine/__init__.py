"""MinkowskiEngine-compatible operator surface on top of libgcl_hip.so (MI355X only, no CPU fallback).

Exports exactly the symbols the reference's hot path uses:
``SparseTensor, MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm,
MinkowskiInstanceNorm (declared, out of scope), MinkowskiFunctional.relu, cat, MinkowskiNetwork,
utils.sparse_quantize / batched_coordinates / sparse_collate``
with the call signatures found at model/resunet.py:38-171, model/residual_block.py:23-53, model/common.py:4-10,
lib/colocation_trainer.py:843-845, scripts/test_kitti.py:143-147, util/misc.py:118-128.

``sys.modules['MinkowskiEngine'] = gcl_amd.MinkowskiEngine`` makes the reference's model files run on it unchanged
(INTEGRATION.md).  Semantics are those of oracle/me_oracle.py; every operator is a hand-written HIP kernel
reached through the C ABI in include/gcl_amd.h.
"""
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from . import utils
from . import MinkowskiFunctional  # noqa: F401  (import MinkowskiEngine.MinkowskiFunctional as MEF)
from .core import CoordinateManager, CoordinateMapKey, SparseTensor, cat
from . import ops
from .ops import batch_norm, invalidate_amax, set_conv_precision, sparse_conv

__all__ = ["SparseTensor", "CoordinateManager", "CoordinateMapKey", "MinkowskiConvolution",
           "MinkowskiConvolutionTranspose", "MinkowskiBatchNorm", "MinkowskiInstanceNorm", "MinkowskiNetwork",
           "MinkowskiFunctional", "cat", "utils", "set_conv_precision", "invalidate_amax"]


# tuning knob: let convolutions emit the column sums a following BatchNorm needs (saves its statistics pass)
FUSED_BN_STATS = os.environ.get("GCL_FUSED_BN_STATS", "1") == "1"
# training: convolution + BatchNorm of ME.conv_bn as one autograd node (same launches, less host work per layer)
FUSED_CONV_BN_NODE = os.environ.get("GCL_FUSED_CONV_BN_NODE", "1") == "1"


class MinkowskiNetwork(nn.Module):
    """``ME.MinkowskiNetwork.__init__(self, D)`` (model/resunet.py:31)."""

    def __init__(self, D):
        super().__init__()
        self.D = D
        self._amax_group = None
        self.register_forward_pre_hook(MinkowskiNetwork._ensure_amax_group)

    @staticmethod
    def _ensure_amax_group(self, _args):
        # fp16x3: max|W| of every convolution kernel of the network is refreshed by ONE launch per optimizer step
        if self._amax_group is None:
            ws = [m.kernel for m in self.modules() if isinstance(m, _ConvBase) and m.in_channels > 4]
            self._amax_group = ops.WeightAmaxGroup(ws) if ws else False
        elif self._amax_group and self.training:
            # training forwards re-measure max|W| unconditionally (the launch happens once per optimizer step anyway):
            # writes through ``p.data`` / raw aliases do not bump ``p._version`` and would leave a stale scale.  In eval
            # mode the tags are keyed on ``_version``; after such a write call ME.invalidate_amax()
            self._amax_group.void_tags()


class _ConvBase(nn.Module):
    TRANSPOSE = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, dimension=None):
        super().__init__()
        if dimension != 3:
            raise NotImplementedError("gcl_amd implements dimension=3 only")
        if kernel_generator is not None:
            raise NotImplementedError("custom kernel generators are not supported")
        if kernel_size not in (1, 3, 5):
            raise NotImplementedError(f"kernel_size {kernel_size} not supported (1, 3, 5)")
        if stride not in (1, 2) or dilation != 1:
            raise NotImplementedError("supported: stride 1 or 2, dilation 1")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.dimension = kernel_size, stride, dilation, dimension
        self.kernel_volume = kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME: uniform(-1/sqrt(n), 1/sqrt(n)) with n = (Cout if transposed else Cin) * kernel_volume
        n = (self.out_channels if self.TRANSPOSE else self.in_channels) * self.kernel_volume
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, x):
        mgr, t_out, kmap, n_out = self._maps(x)
        # in training mode the epilogue also emits per-tile column sums, which a following MinkowskiBatchNorm consumes
        F, stats = sparse_conv(x.F, self.kernel, kmap, n_out, self.TRANSPOSE, self.bias, mgr,
                               want_stats=FUSED_BN_STATS and self.training and self.bias is None
                               and self.in_channels > 4)
        out = SparseTensor(F, coordinate_map_key=CoordinateMapKey(t_out), coordinate_manager=mgr)
        out._bn_stats = stats
        return out

    def _maps(self, x):
        if not isinstance(x, SparseTensor):
            raise TypeError("input must be a SparseTensor")
        mgr = x.coordinate_manager
        t_in = x.coordinate_map_key.tensor_stride
        if self.TRANSPOSE:
            if t_in % self.stride:
                raise ValueError("transposed convolution below tensor stride 1")
            t_out = t_in // self.stride
        else:
            t_out = t_in * self.stride
        if self.kernel_volume == 1:
            if self.stride != 1:
                raise NotImplementedError("kernel_size 1 with stride > 1")
            kmap = None
            n_out = len(x)
        elif self.TRANSPOSE:
            kmap = mgr.get_kernel_map(t_out, self.kernel_size, self.stride)     # fine -> coarse map, used swapped
            n_out = mgr.num_rows(t_out)
        else:
            kmap = mgr.get_kernel_map(t_in, self.kernel_size, self.stride)
            n_out = mgr.num_rows(t_out)
        return mgr, t_out, kmap, n_out

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, dilation={self.dilation}")


class MinkowskiConvolution(_ConvBase):
    """Generalized sparse convolution (model/resunet.py:38-45 and the other forward convs)."""
    TRANSPOSE = False


class MinkowskiConvolutionTranspose(_ConvBase):
    """Stride-2 up-convolution onto the existing finer coordinate map (model/resunet.py:101-134)."""
    TRANSPOSE = True


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over the rows of ``x.F``; parameters live in the submodule ``bn`` (state_dict keys
    ``*.bn.weight`` ... as in ME).  ``forward(x, residual=None, relu=False)`` additionally exposes the fused
    residual-add / ReLU of BasicBlock (one kernel instead of three)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        if not (affine and track_running_stats):
            raise NotImplementedError("affine=True, track_running_stats=True only")
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=True, track_running_stats=True)
        self._pending_batches = 0                # training forwards not yet added to bn.num_batches_tracked (device)
        self._train_forwards = 0                 # monotonic: the running statistics change through raw pointers
        self.register_state_dict_pre_hook(MinkowskiBatchNorm._flush_batches)

    @staticmethod
    def _flush_batches(self, *_):
        if self._pending_batches:
            self.bn.num_batches_tracked += self._pending_batches
            self._pending_batches = 0

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        return super()._load_from_state_dict(*args, **kwargs)

    def eval_affine(self):
        """(scale, shift) of the eval-mode BatchNorm y = x * scale + shift, cached until a parameter or buffer changes."""
        bn = self.bn
        key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               bn.weight.data_ptr(), bn.running_mean.data_ptr(), self._train_forwards, float(bn.eps))
        cached = getattr(self, "_affine", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).float().contiguous()
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
            self._affine = cached = (key, scale, shift)
        return cached[1], cached[2]

    def forward(self, x, residual=None, relu=False):
        bn = self.bn
        res = residual.F if residual is not None else None
        if residual is not None and residual.coordinate_map_key != x.coordinate_map_key:
            raise ValueError("residual lives on a different coordinate map")
        F = batch_norm(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                       bn.momentum, bn.eps, res, relu, getattr(x, "_bn_stats", None))
        if self.training:
            self._pending_batches += 1           # flushed into bn.num_batches_tracked when the state is read
            self._train_forwards += 1            # voids the cached eval-mode (scale, shift)
        out = SparseTensor(F, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        out._nonneg = bool(relu)
        return out


def conv_bn(conv, norm, x, residual=None, relu=False):
    """``norm(conv(x))`` with the fused residual add / ReLU of MinkowskiBatchNorm.forward.  In inference (module in eval
    mode, autograd off, BatchNorm, default arithmetic) convolution + BatchNorm + residual + ReLU are ONE launch
    (gcl_conv_fwd_fused); otherwise the two modules run one after the other."""
    trace = ops._EVAL_TRACE
    if trace is not None and not torch.is_grad_enabled():
        # inference trace (ops._EVAL_TRACE): run the layer as usual with the trace suspended, then record it as ONE entry
        if not isinstance(norm, MinkowskiBatchNorm) or conv.training or norm.training or conv.bias is not None:
            raise ValueError("the inference plan covers bias-free convolutions followed by a BatchNorm in eval mode")
        ops._EVAL_TRACE = None
        try:
            out = conv_bn(conv, norm, x, residual, relu)
        finally:
            ops._EVAL_TRACE = trace
        _, _, kmap, _ = conv._maps(x)
        c2 = ops._SubCtx()
        c2.relu = bool(relu)
        c2.bn_extra = (norm.bn.running_mean, norm.bn.running_var, norm.bn.momentum, norm.bn.eps, norm)
        trace.add("convbn", out.F, ops._TraceCtx(kmap, conv.TRANSPOSE, conv.kernel), c2, x.F,
                  residual.F if residual is not None else None, (conv.kernel, norm.bn.weight, norm.bn.bias))
        return out
    fused = (not conv.training and not norm.training and not torch.is_grad_enabled()
             and isinstance(norm, MinkowskiBatchNorm) and conv.bias is None and conv.in_channels > 4
             and ops.PRECISION == "fp16x3")
    if (not fused and FUSED_CONV_BN_NODE and conv.training and norm.training and isinstance(norm, MinkowskiBatchNorm)
            and conv.bias is None and torch.is_grad_enabled()):
        # training: the same launches as norm(conv(x)) behind ONE autograd node (ops._ConvBNFn)
        mgr, t_out, kmap, n_out = conv._maps(x)
        if residual is not None and residual.coordinate_map_key.tensor_stride != t_out:
            raise ValueError("residual lives on a different coordinate map")
        bn = norm.bn
        F = ops.conv_bn_train(x.F, conv.kernel, kmap, n_out, conv.TRANSPOSE, mgr, bn.weight, bn.bias, bn.running_mean,
                              bn.running_var, bn.momentum, bn.eps, residual.F if residual is not None else None,
                              bool(relu), FUSED_BN_STATS and conv.in_channels > 4, bn_module=norm)
        norm._pending_batches += 1
        norm._train_forwards += 1
        out = SparseTensor(F, coordinate_map_key=CoordinateMapKey(t_out), coordinate_manager=mgr)
        out._nonneg = bool(relu)
        return out
    if not fused:
        return norm(conv(x), residual=residual, relu=relu)
    mgr, t_out, kmap, n_out = conv._maps(x)
    if residual is not None and residual.coordinate_map_key.tensor_stride != t_out:
        raise ValueError("residual lives on a different coordinate map")
    scale, shift = norm.eval_affine()
    F = ops.conv_bn_eval(x.F, conv.kernel, kmap, n_out, conv.TRANSPOSE, scale, shift,
                         residual.F if residual is not None else None, relu)
    out = SparseTensor(F, coordinate_map_key=CoordinateMapKey(t_out), coordinate_manager=mgr)
    out._nonneg = bool(relu)
    return out


class MinkowskiInstanceNorm(nn.Module):
    """``ME.MinkowskiInstanceNorm(num_features)`` (model/common.py:7-8 passes ``dimension=D`` too): every cloud of the
    batch is normalised on its own -- per channel (x - mean) / sqrt(var + 1e-8) with the biased variance over the
    cloud's voxels, no running statistics (train == eval) -- followed by a shared affine map with parameters
    ``weight`` / ``bias`` of shape [1, C] (ME 0.5).  ``forward(x, residual=None, relu=False)`` exposes the same fused
    residual-add / ReLU as MinkowskiBatchNorm."""

    EPS = 1e-8

    def __init__(self, num_features, dimension=-1):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x, residual=None, relu=False):
        if residual is not None and residual.coordinate_map_key != x.coordinate_map_key:
            raise ValueError("residual lives on a different coordinate map")
        seg = x.coordinate_manager.batch_segments(x.coordinate_map_key.tensor_stride)
        F = ops.instance_norm(x.F, self.weight, self.bias, seg, self.EPS,
                              residual.F if residual is not None else None, relu)
        out = SparseTensor(F, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        out._nonneg = bool(relu)
        return out


def _selfcheck():
    """Fail loudly at first use when there is no GPU / no library."""
    _lib.require_gpu()
