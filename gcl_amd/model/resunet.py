"""ResUNet2 family -- the sparse 4-level residual U-Net of GCL/FCGF on the MI355X operator surface.

Interface and parameter names follow model/resunet.py (constructor :24-30, forward :173-232, widths :235-266), so a
``state_dict`` is interchangeable with the reference's.  Layers are generated from the width tables instead of being
spelled out; BN + ReLU (+ residual) sites use the fused kernels of gcl_amd.MinkowskiEngine.MinkowskiBatchNorm.
The IN variants (:269-291: BatchNorm after the level convolutions, InstanceNorm inside the residual blocks) are built
too; the ``KERNEL_SIZES[0]`` "extra" branch (:48-57, :141-151) is not (unused by GCL's scripts).
"""
import os
import torch

import gcl_amd.MinkowskiEngine as ME
import gcl_amd.MinkowskiEngine.MinkowskiFunctional as MEF
from gcl_amd.model.common import get_norm
from gcl_amd.model.residual_block import get_block


# Structure version of every torch module of the process (see ResUNet2._module_list): bumped by torch's global registration
# hooks whenever a sub-module or parameter is (re)registered anywhere.  The hooks return None (they replace nothing).
_STRUCTURE_VERSION = [0]


def _bump_structure_version(*_args, **_kwargs):
    _STRUCTURE_VERSION[0] += 1


try:
    from torch.nn.modules import module as _torch_module
    _torch_module.register_module_module_registration_hook(_bump_structure_version)
    _torch_module.register_module_parameter_registration_hook(_bump_structure_version)
    _HOOKED = True
except Exception:          # an older torch: fall back to comparing identities
    _HOOKED = False


class ResUNet2(ME.MinkowskiNetwork):
    NORM_TYPE = None
    BLOCK_NORM_TYPE = "BN"
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]
    STRIDES = [1, 2, 2, 2]
    KERNEL_SIZES = [None, 3, 3, 3]
    DILATIONS = [1, 1, 1, 1]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3):
        super().__init__(D)
        if self.KERNEL_SIZES[0] is not None:
            raise NotImplementedError("the conv1_extra variants (ResUNetFatBNEXP_V2) are outside the GCL hot path")
        ch, tr = self.CHANNELS, self.TR_CHANNELS
        self.normalize_feature = normalize_feature

        def conv(cin, cout, ks, level, transpose=False):
            cls = ME.MinkowskiConvolutionTranspose if transpose else ME.MinkowskiConvolution
            return cls(in_channels=cin, out_channels=cout, kernel_size=ks, stride=self.STRIDES[level],
                       dilation=self.DILATIONS[level], bias=False, dimension=D)

        def norm(c):
            return get_norm(self.NORM_TYPE, c, bn_momentum=bn_momentum, D=D)

        def block(c):
            return get_block(self.BLOCK_NORM_TYPE, c, c, bn_momentum=bn_momentum, D=D)

        # encoder: conv{l}, norm{l}, block{l}
        enc_in = [None, in_channels, ch[1], ch[2], ch[3]]
        for l in (1, 2, 3, 4):
            ks = conv1_kernel_size if l == 1 else self.KERNEL_SIZES[l - 1]
            setattr(self, f"conv{l}", conv(enc_in[l], ch[l], ks, l - 1))
            setattr(self, f"norm{l}", norm(ch[l]))
            setattr(self, f"block{l}", block(ch[l]))
        # decoder: conv{l}_tr, norm{l}_tr, block{l}_tr ; input = previous decoder output (+ skip)
        dec_in = {4: ch[4], 3: ch[3] + tr[4], 2: ch[2] + tr[3]}
        for l in (4, 3, 2):
            setattr(self, f"conv{l}_tr", conv(dec_in[l], tr[l], self.KERNEL_SIZES[l - 1], l - 1, transpose=True))
            setattr(self, f"norm{l}_tr", norm(tr[l]))
            setattr(self, f"block{l}_tr", block(tr[l]))
        self.conv1_tr = ME.MinkowskiConvolution(in_channels=ch[1] + tr[2], out_channels=tr[1], kernel_size=1,
                                                stride=self.STRIDES[0], dilation=self.DILATIONS[0], bias=False,
                                                dimension=D)
        self.final = ME.MinkowskiConvolution(in_channels=tr[1], out_channels=out_channels, kernel_size=1, stride=1,
                                             dilation=1, bias=True, dimension=D)

    def map_specs(self):
        """(t_in, kernel_size, stride, tables, pairs) of every kernel map a training step of this network touches
        (CoordinateManager.prefetch): the first layer reads its map directly (no sorted table, no pair lists), the
        stride-1 levels need one table, the strided maps both (down-convolution forward / up-convolution input gradient
        read ``nbr``, the other two directions ``nbr_t``)."""
        specs = [(1, self.conv1.kernel_size, 1, () if self.conv1.in_channels <= 4 else (False,), self.conv1.in_channels > 4)]
        t = 1
        for l in (1, 2, 3, 4):
            if l > 1:
                specs.append((t // 2, self.KERNEL_SIZES[l - 1], 2, (False, True), True))
            specs.append((t, 3, 1, (False,), True))
            t *= 2
        return specs

    def native_map_specs(self, training=True):
        """``map_specs`` + the identity pair list of the two kernel_size-1 heads: what CoordinateManager.build_native
        builds in one call, in the order the whole-network plan indexes it.  ``training=False``: the same maps and
        tables without the weight gradient's pair lists (inference)."""
        merged = {}
        for t_in, ks, stride, tables, pairs in self.map_specs() + [(1, 1, 1, (), True)]:
            key = (t_in, ks, stride)          # conv1 with kernel 3 shares its map with block1
            old = merged.get(key, ((), False))
            merged[key] = (tuple(sorted(set(old[0]) | set(bool(t) for t in tables))), (old[1] or bool(pairs)) and training)
        out = [k + v for k, v in merged.items()]
        if self.conv1.in_channels == 1 and os.environ.get("GCL_STEM_OCC", "1") != "0":
            # the reference's loaders feed occupancy features, torch.ones((n, 1)): every cloud at inference
            # (lib/data_loaders.py test sets / scripts/test_kitti.py), the neighbour clouds of a sample in training -- only
            # the centre cloud carries lib/transforms.py:18 Jitter (lib/colocation_data_loader.py:401-415).  With the
            # presence words of the first layer's table its kernels add W[k] over the set bits for the rows of all-ones
            # clouds and walk the table for the others (per-row device flags, bitwise the same results).
            k1 = (1, self.conv1.kernel_size, 1)
            out = [(s + ("presence",)) if s[:3] == k1 else s for s in out]
        return out

    _plan = None          # native.NetworkPlan once recorded; False when the graph is outside what the plan covers

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_plan", None)          # a native handle: rebuilt from the next recorded step
        state.pop("_walk_cache", None)    # module walk of this process (ResUNet2._module_list)
        return state

    def _use_tape(self, x):
        ops = ME.ops
        return (ops.TAPE_ENABLED and self.training and torch.is_grad_enabled() and self.NORM_TYPE == "BN"
                and self.BLOCK_NORM_TYPE == "BN" and ME.FUSED_CONV_BN_NODE and not x.F.requires_grad
                and self._all_training())

    def plan_for(self, x):
        """The native.NetworkPlan ``forward(x)`` will run (None: it takes the Tape / per-layer path): a plan recorded from
        an earlier training step, the default arithmetic, and a coordinate manager built by build_native with this
        model's ``native_map_specs``."""
        from gcl_amd.MinkowskiEngine import native
        plan = self.__dict__.get("_plan")
        nm = x.coordinate_manager.native
        if (not native.PLAN_ENABLED or not isinstance(plan, native.NetworkPlan) or nm is None
                or nm.keys != plan.spec_keys or x.coordinate_map_key.tensor_stride != 1
                or ME.ops.PRECISION != "fp16x3" or not self._use_tape(x)
                or len(plan.params) != sum(1 for _ in self.parameters())
                # frozen parameters (fine-tuning) / parameters no record claims: the plan would write their gradients
                # and the optimizer would update them; the Tape leaves p.grad None for them
                or not getattr(plan, "covers_all_params", False) or not all(p.requires_grad for p in plan.params)):
            return None
        return plan

    def _module_list(self):
        """``list(self.modules())`` / the parameter count, walked once per model structure: the two guards of the inference
        path cost 0.17 ms per pass when they walk the module tree (a pass over one pair is host-bound).  Re-walked when the
        registered sub-modules or parameters of ANY module changed -- by count (add_module / register_parameter / ``del``) or
        by IDENTITY (``model.block1 = NewBlock()``, a norm-layer swap under the same name keeps every count).
        Identity changes are seen through torch's global registration hooks (every ``Module.__setattr__`` / ``add_module`` /
        ``register_parameter`` of the process bumps ``_STRUCTURE_VERSION``): one integer compare per pass; where a torch
        build lacks the hooks, the ids of every module's children and parameters are compared instead (~ 90 us per call)."""
        c = self.__dict__.get("_walk_cache")
        if c is not None and all(len(m._modules) == a and len(m._parameters) == b for m, a, b, _, _ in c[2]):
            if _HOOKED:
                if c[3] == _STRUCTURE_VERSION[0]:
                    return c[0]
            elif all(tuple(map(id, m._modules.values())) == mi and tuple(map(id, m._parameters.values())) == pi
                     for m, _, _, mi, pi in c[2]):
                return c[0]
        mods = list(self.modules())
        ids = [(m, len(m._modules), len(m._parameters), tuple(map(id, m._modules.values())),
                tuple(map(id, m._parameters.values()))) for m in mods]
        self.__dict__["_walk_cache"] = (mods, sum(1 for _ in self.parameters()), ids, _STRUCTURE_VERSION[0])
        return mods

    def _n_parameters(self, validated=False):
        """``validated``: the caller has just called ``_module_list`` (one check per pass, not two)."""
        if not validated:
            self._module_list()
        return self.__dict__["_walk_cache"][1]

    def _forward_eval(self, x):
        """Inference through the native plan (ONE gcl_maps_build + ONE gcl_plan_forward_eval call per pass), or None when
        the pass has to take the per-operator path: BatchNorm models in eval mode under torch.no_grad(), default
        arithmetic, a stride-1 input whose coordinate manager is native or still empty (then it is rebuilt natively).
        The first such pass is traced (ops._EVAL_TRACE) to derive the plan unless a training step already recorded one."""
        from gcl_amd.MinkowskiEngine import native
        ops = ME.ops
        if (self.training or torch.is_grad_enabled() or not native.PLAN_ENABLED or ops.PRECISION != "fp16x3"
                or self.NORM_TYPE != "BN" or self.BLOCK_NORM_TYPE != "BN" or x.coordinate_map_key.tensor_stride != 1
                or self.__dict__.get("_plan") is False or any(m.training for m in self._module_list())
                or ME.core.SORT_WINDOW or ME.core.SPATIAL_MAX_STRIDE):
            return None
        mgr = x.coordinate_manager
        plan = self.__dict__.get("_plan")
        # the maps-independent half of the pass's Python (native.NetworkPlan.prepare_eval) BEFORE the map build
        prep = plan.prepare_eval(x.F.device) if isinstance(plan, native.NetworkPlan) and x.F.is_cuda else None
        if mgr.native is None:
            if mgr._kmaps or len(mgr._maps) > 1:      # maps already built from Python: keep using them
                return None
            # with a plan in hand the deeper levels' maps are built on a side stream, beside the first layers (the plan waits)
            side = (native.eval_side_stream(x.F.device, x.F.shape[0])
                    if isinstance(self.__dict__.get("_plan"), native.NetworkPlan) else None)
            mgr = ME.CoordinateManager.build_native(mgr.get_coords(1), self.native_map_specs(training=False), side_stream=side)
            x = ME.SparseTensor(x.F, coordinate_map_key=x.coordinate_map_key, coordinate_manager=mgr)
        plan = self.__dict__.get("_plan")
        if isinstance(plan, native.NetworkPlan):
            if mgr.native.keys != plan.spec_keys or len(plan.params) != self._n_parameters(validated=True):
                mgr.native.wait_ready()
                return None
            try:
                F = plan.run_eval(x.F, mgr.native, prepared=prep)
            except Exception:
                mgr.native.wait_ready()       # the side stream may still be writing the arena this frame is about to drop
                raise
            return ME.SparseTensor(F, coordinate_map_key=ME.CoordinateMapKey(1 << plan.records[-1]["level_out"]),
                                   coordinate_manager=mgr)
        trace, ops._EVAL_TRACE = ops.Tape(), None
        ops._EVAL_TRACE = trace
        try:
            out = self._forward(x)
        finally:
            ops._EVAL_TRACE = None
        try:
            self._plan = native.NetworkPlan.from_tape(trace, self, x.F, 0, mgr.native.keys)
        except ValueError as e:
            self._plan, self._plan_error = False, str(e)
        return out

    def forward(self, x):
        ops = ME.ops
        if not self.training and not torch.is_grad_enabled():
            out = self._forward_eval(x)
            if out is not None:
                return out
        if not self._use_tape(x):
            return self._forward(x)
        plan = self.plan_for(x)
        if plan is not None:              # the whole pass is ONE native call (csrc/plan.hip)
            F = plan.run(x.F, x.coordinate_manager.native)
            return ME.SparseTensor(F, coordinate_map_key=ME.CoordinateMapKey(1 << plan.records[-1]["level_out"]),
                                   coordinate_manager=x.coordinate_manager)
        from gcl_amd.MinkowskiEngine import native
        nm = x.coordinate_manager.native
        with ops.tape() as tp:          # the whole network as ONE autograd node (ops.Tape)
            out = self._forward(x)
            if (native.PLAN_ENABLED and nm is not None and self.__dict__.get("_plan") is None
                    and x.coordinate_map_key.tensor_stride == 1):
                try:                    # this recorded pass becomes the plan of the following steps
                    self._plan = native.NetworkPlan.from_tape(tp, self, x.F, 0, nm.keys)
                except ValueError as e:
                    self._plan, self._plan_error = False, str(e)
            F = tp.finish(out.F)
        return ME.SparseTensor(F, coordinate_map_key=out.coordinate_map_key, coordinate_manager=out.coordinate_manager)

    def _all_training(self):
        """The Tape records ``conv_bn`` calls only; a submodule in eval mode (frozen-BN fine-tuning) makes ME.conv_bn
        fall back to ``norm(conv(x))``, which the Tape cannot follow -- such models take the per-layer autograd path."""
        return all(m.training for m in self.modules())

    def _forward(self, x):
        skips = {}
        out = x
        for l in (1, 2, 3, 4):
            out = ME.conv_bn(getattr(self, f"conv{l}"), getattr(self, f"norm{l}"), out)
            out = getattr(self, f"block{l}")(out)          # ends in a fused relu
            skips[l] = out
            out = MEF.relu(out)
        for l in (4, 3, 2):
            out = ME.conv_bn(getattr(self, f"conv{l}_tr"), getattr(self, f"norm{l}_tr"), out)
            out = MEF.relu(getattr(self, f"block{l}_tr")(out))
            out = ME.cat(out, skips[l - 1])
        out = self.final(MEF.relu(self.conv1_tr(out)))
        if self.normalize_feature:
            return ME.SparseTensor(ME.ops.l2_normalize_rows(out.F),        # = out.F / torch.norm(out.F, 2, 1, True)
                                   coordinate_map_key=out.coordinate_map_key,
                                   coordinate_manager=out.coordinate_manager)
        return out


def _variant(name, tr_channels, channels=None, doc=""):
    attrs = {"NORM_TYPE": "BN", "TR_CHANNELS": tr_channels, "__doc__": doc}
    if channels is not None:
        attrs["CHANNELS"] = channels
    return type(name, (ResUNet2,), attrs)


ResUNetBN2 = _variant("ResUNetBN2", [None, 32, 64, 64, 128], doc="model/resunet.py:235-236")
ResUNetBN2B = _variant("ResUNetBN2B", [None, 64, 64, 64, 64], doc="model/resunet.py:239-242")
ResUNetBN2C = _variant("ResUNetBN2C", [None, 64, 64, 64, 128], doc="model/resunet.py:245-248 (north-star model)")
ResUNetBN2D = _variant("ResUNetBN2D", [None, 64, 64, 128, 128], doc="model/resunet.py:251-254")
ResUNetBN2E = _variant("ResUNetBN2E", [None, 64, 128, 128, 128], [None, 128, 128, 128, 256],
                       doc="model/resunet.py:257-260")
ResUNetFatBN = _variant("ResUNetFatBN", [None, 128, 128, 128, 256], doc="model/resunet.py:263-266 (script default)")


def _in_variant(name, base, doc):
    return type(name, (base,), {"NORM_TYPE": "BN", "BLOCK_NORM_TYPE": "IN", "__doc__": doc})


ResUNetIN2 = _in_variant("ResUNetIN2", ResUNetBN2, "model/resunet.py:269-271")
ResUNetIN2B = _in_variant("ResUNetIN2B", ResUNetBN2B, "model/resunet.py:274-276")
ResUNetIN2C = _in_variant("ResUNetIN2C", ResUNetBN2C, "model/resunet.py:279-281")
ResUNetIN2D = _in_variant("ResUNetIN2D", ResUNetBN2D, "model/resunet.py:284-286")
ResUNetIN2E = _in_variant("ResUNetIN2E", ResUNetBN2E, "model/resunet.py:289-291")
