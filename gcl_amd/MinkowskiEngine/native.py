"""Host side of the native step runtime (include/gcl_amd.h, "Native step runtime"; csrc/plan.hip).

``NativeMaps``    everything a CoordinateManager builds for a network -- stride maps, kernel maps, mask-sorted tables,
                  pair lists -- by ONE ``gcl_maps_build`` call into one arena (the call releases the interpreter lock and
                  does its two host syncs itself: a loader-side thread runs it on a side stream).
``NetworkPlan``   a network's training pass as operator records, derived from ONE recorded ``ops.Tape`` of the model
                  (whatever model/resunet.py:173-232 calls on the ME surface); ``run`` = one ``gcl_plan_forward`` call,
                  the backward pass = one ``gcl_plan_backward`` call per gradient bucket.

Both issue exactly the launches of the per-operator path (same entry points, same arguments, same order), so losses
and parameters stay bitwise equal (tests/test_gpu_plan.py); what disappears is ~560 Python -> ctypes round trips per
training step.  There is no CPU path here either: everything raises without the library / a GPU.
"""
import ctypes
import os
import threading

import torch
from torch.autograd.function import once_differentiable

from .. import _lib

# the whole-network plan replaces the Tape after the first (recorded) training step of a model; GCL_PLAN=0 keeps the Tape
PLAN_ENABLED = os.environ.get("GCL_PLAN", "1") == "1"
# weight gradients of the plan's backward pass on a second stream ("0" off, "1" same priority [default], "low" lowest
# priority): measured 14.48 -> 13.72 ms per step (same box, alternating runs; the optimizer is their only consumer)
AUX_STREAM = {"0": "", "1": "1", "low": "low", "high": "high"}[os.environ.get("GCL_PLAN_AUX", "low")]      # "1": normal priority


def _addr(t):
    return t.data_ptr() if t is not None else 0


_SIDE_STREAMS = {}


def eval_side_stream(device, n_rows=None):
    """The stream a one-call inference pass builds its deeper levels' maps on, or None = one stream (GCL_EVAL_SPLIT_MAPS=0, or
    fewer than GCL_EVAL_SPLIT_MIN_ROWS = 100 000 rows: a pass over one pair of clouds is bound by the enqueuing thread right
    after the level sizes' read-back, and the split costs it sixteen more launches -- 26.2 -> 25.2 M voxels/s at 36 k rows,
    93.2 -> 97.0 at 279 k; profiles/r06_conv_experiments.txt, 86)."""
    if os.environ.get("GCL_EVAL_SPLIT_MAPS", "1") == "0":
        return None
    if n_rows is not None and n_rows < int(os.environ.get("GCL_EVAL_SPLIT_MIN_ROWS", "100000")):
        return None
    key = (torch.device(device).index, threading.get_ident())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class NativeMaps:
    """The maps of one batch.  ``specs``: (t_in, kernel_size, stride, tables, pairs) tuples as in
    CoordinateManager.prefetch, plus (t, 1, 1, (), True) for the identity pair list of a kernel_size-1 convolution.
    Built on torch's CURRENT stream; ``arena`` (uint8 device tensor) may be passed in to re-use memory.
    ``side_stream`` (inference passes over few rows): gcl_maps_build_split -- the maps the first layers do not use are enqueued
    THERE, ``self.ready`` is recorded behind them, and the native plan makes the current stream wait for it in front of the
    first record that needs one; any OTHER consumer of the maps must ``wait_ready()`` first."""

    def __init__(self, coordinates, specs, n_levels=4, arena=None, side_stream=None):
        lib = _lib.require_gpu()
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N, 4] = (batch, x, y, z)")
        C = coordinates.to(torch.int32).contiguous()
        n = C.shape[0]
        if n == 0:
            raise ValueError("empty SparseTensor")
        self.coords, self.device, self.n_levels = C, C.device, int(n_levels)
        self.keys = [(int(s[0]), int(s[1]), int(s[2])) for s in specs]
        if len(self.keys) > _lib.MAX_MAPS:
            raise ValueError(f"at most {_lib.MAX_MAPS} kernel maps per network")
        arr = (_lib.MapSpec * len(specs))()
        for a, s in zip(arr, specs):
            a.t_in, a.kernel_size, a.stride = int(s[0]), int(s[1]), int(s[2])
            a.tables = sum(1 << int(bool(tr)) for tr in set(bool(t) for t in s[3]))
            if len(s) > 5 and s[5] == "presence":      # presence words of nbr (first-layer occupancy path)
                a.tables |= 4
            a.pairs = int(bool(s[4]))
        need = lib.gcl_maps_arena_bytes(n, arr, len(specs), self.n_levels)
        if need < 0:
            raise ValueError("gcl_maps_arena_bytes rejected the map specification")
        if arena is None or arena.numel() < need:
            # 12 % of headroom: pooled arenas (the trainer's map slots, the eval loop's ring) meet batches of different sizes, and
            # an exact-size arena is re-allocated every time its slot meets a larger one -- a hipMalloc inside a training step
            # (bench.py's driver command: one 13.2 ms step among 11.5 ms ones, eleven steps in)
            arena = torch.empty(int(need * 1.125) + 4096, dtype=torch.uint8, device=C.device)
        self.arena = arena
        self.pinned = torch.empty(_lib.MAPS_PINNED_BYTES // 4, dtype=torch.int32, pin_memory=True)
        self.desc = _lib.MapsDesc()
        self.ready = None
        if side_stream is None:
            rc = lib.gcl_maps_build(_lib.ptr(C), n, arr, len(specs), self.n_levels, _lib.ptr(arena), arena.numel(),
                                    ctypes.c_void_p(self.pinned.data_ptr()), ctypes.byref(self.desc), _lib.stream())
        else:
            rc = lib.gcl_maps_build_split(_lib.ptr(C), n, arr, len(specs), self.n_levels, _lib.ptr(arena), arena.numel(),
                                          ctypes.c_void_p(self.pinned.data_ptr()), ctypes.byref(self.desc), _lib.stream(),
                                          ctypes.c_void_p(side_stream.cuda_stream))
            if rc == 0 and self.desc.late_mask:
                self.ready = torch.cuda.Event()
                self.ready.record(side_stream)
                self.desc.ready_event = ctypes.c_void_p(self.ready.cuda_event)
        if rc != 0:
            msg = lib.gcl_last_error().decode()
            if rc == -1:
                raise ValueError(msg)
            raise RuntimeError(f"libgcl_hip gcl_maps_build failed (rc={rc}): {msg}")

    def wait_ready(self):
        """The current stream waits for the maps a split build left on the side stream (no-op otherwise)."""
        if self.ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self.ready)

    def index_of(self, t_in, kernel_size, stride):
        return self.keys.index((int(t_in), int(kernel_size), int(stride)))

    def num_rows(self, level):
        return int(self.desc.n_rows[level])

    def view(self, addr, shape, dtype):
        """A tensor over arena memory at device address ``addr`` (None for NULL)."""
        if not addr:
            return None
        if addr == self.coords.data_ptr():
            return self.coords
        numel = 1
        for s in shape:
            numel *= int(s)
        nbytes = numel * torch.empty(0, dtype=dtype).element_size()
        off = addr - self.arena.data_ptr()
        if off < 0 or off + nbytes > self.arena.numel():
            raise RuntimeError("NativeMaps.view: address outside the arena")
        return self.arena[off:off + nbytes].view(dtype).view(*shape)


class _PlanRun:
    """One forward pass of a plan that is waiting for its backward pass (keeps the arena and the maps alive)."""

    __slots__ = ("plan", "arena", "maps", "y", "x", "grad_targets", "segments", "done")

    def __del__(self):          # the output was dropped without a backward pass: un-park the pass on the native side
        try:
            if not self.done and self.plan is not None and self.plan.handle and self.arena is not None:
                _lib.load().gcl_plan_release(self.plan.handle, ctypes.c_void_p(self.arena.data_ptr()))
        except Exception:
            pass


class _PlanFn(torch.autograd.Function):
    """Autograd node of a whole plan pass.  ``params`` are inputs so that autograd receives their gradients when the
    caller has not seated them (``NetworkPlan.grad_targets`` is None); with seated gradients the pass writes them in
    place and returns None for every parameter."""

    @staticmethod
    def forward(ctx, run, *params):
        ctx.run = run
        return run.y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        run, ctx.run = ctx.run, None
        plan = run.plan
        grads = plan._backward(run, dy.contiguous())
        if grads is None:
            return None, None
        return (None,) + tuple(grads)


class NetworkPlan:
    """Operator records of a network + the native handle.  Build with ``NetworkPlan.from_tape``."""

    def __init__(self, records, n_tensors, params, bn_buffers, bn_modules, weight_order, spec_keys, out_channels):
        lib = _lib.load()
        from . import ops
        self.params = list(params)
        self.bn_buffers = list(bn_buffers)           # [(running_mean, running_var)]
        self.bn_modules = list(bn_modules)           # MinkowskiBatchNorm modules (training-forward counters)
        self.spec_keys = list(spec_keys)
        self.records = records
        self.out_channels = int(out_channels)
        arr = (_lib.PlanOp * len(records))()
        for a, r in zip(arr, records):
            for k, v in r.items():
                setattr(a, k, v)
        worder = (ctypes.c_int32 * max(1, len(weight_order)))(*weight_order)
        self.handle = lib.gcl_plan_create(arr, len(records), n_tensors, len(self.params), worder, len(weight_order),
                                          ops.PRESPLIT_MIN_C)
        if not self.handle:
            raise ValueError(lib.gcl_last_error().decode())
        self.handle = ctypes.c_void_p(self.handle)
        self._state = None
        self._key = None
        self.grad_targets = None       # list of tensors (one per parameter) that RECEIVE the gradients, or None
        self.bucket_of_param = None    # parameter index -> bucket id (FlatDDP), with ``on_bucket`` called as they complete
        self.on_bucket = None
        self._param_ptrs = self._bn_ptrs = None
        self._ptr_key = None
        self.profile_next = False
        self.last_profile = None
        self._anchor = None
        self._aux = None
        self._eval_key = None

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.load().gcl_plan_destroy(h)
            except Exception:
                pass

    # ---- construction from a recorded Tape ---------------------------------------------------------------------
    @classmethod
    def from_tape(cls, tape, model, x_in, in_level, spec_keys):
        """``tape``: the ops.Tape of one training forward pass of ``model`` (before ``finish``); ``x_in``: the input
        feature tensor; ``spec_keys``: [(t_in, kernel_size, stride)] in the order of the NativeMaps the plan will run on.
        Raises ValueError for graphs the plan does not cover (the caller then stays on the Tape)."""
        from . import ops
        if ops.PRECISION != "fp16x3" or not ops.FUSE_GRAD_ADD:
            raise ValueError("the plan runs the default arithmetic (fp16x3, fused gradient adds) only")
        params = [p for p in model.parameters()]
        pid = {id(p): i for i, p in enumerate(params)}
        tid = {id(x_in): 0}
        level = {0: int(in_level)}
        chans = {0: int(x_in.shape[1])}
        keys = list(spec_keys)
        records, bn_buffers, bn_modules = [], [], []

        def tensor(t, create=False):
            k = id(t)
            if k not in tid:
                if not create:
                    raise ValueError("a record consumes a tensor the tape did not produce")
                tid[k] = len(tid)
            return tid[k]

        def lvl(t_stride):
            l = int(t_stride).bit_length() - 1
            if (1 << l) != int(t_stride):
                raise ValueError("tensor stride is not a power of two")
            return l

        def base():
            return dict(kind=0, x=-1, x2=-1, y=-1, level_in=0, level_out=0, cin=0, cout=0, map=-1, transpose=0, K=1, w=-1,
                        bias=-1, bn_w=-1, bn_b=-1, bn=-1, relu=0, momentum=0.0, eps=0.0)

        def conv_fields(r, c1, x, y, W, b):
            if c1.generic:
                raise ValueError("generic (non-MFMA) convolution shapes stay on the per-operator path")
            xi = tensor(x)
            K, cin, cout = (W.shape if W.dim() == 3 else (1,) + tuple(W.shape))
            r.update(x=xi, cin=int(cin), cout=int(cout), K=int(K), w=pid[id(W)], transpose=int(bool(c1.transpose)),
                     bias=pid[id(b)] if b is not None else -1, level_in=level[xi])
            km = c1.kmap
            if km is None:
                r["level_out"] = level[xi]
                r["map"] = keys.index((1 << level[xi], 1, 1))
            else:
                l_in, l_out = lvl(km._t_in), lvl(km._t_out)
                if c1.transpose:
                    l_in, l_out = l_out, l_in
                if l_in != level[xi]:
                    raise ValueError("kernel map and input tensor live on different levels")
                r["level_out"] = l_out
                ks = round(km.K ** (1.0 / 3.0))
                r["map"] = keys.index((int(km._t_in), int(ks), int(km._t_out // km._t_in)))
            yi = tensor(y, create=True)
            level[yi], chans[yi] = r["level_out"], int(cout)
            r["y"] = yi

        for e in tape.entries:
            kind, y = e[0], e[1]
            r = base()
            if kind == "convbn":
                c1, c2, x, res, (W, bw, bb) = e[2], e[3], e[4], e[5], e[6]
                r["kind"] = _lib.OP_CONVBN
                conv_fields(r, c1, x, y, W, None)
                rm, rv, momentum, eps, module = c2.bn_extra
                r.update(bn_w=pid[id(bw)], bn_b=pid[id(bb)], bn=len(bn_buffers), relu=int(bool(c2.relu)),
                         momentum=float(momentum), eps=float(eps), x2=tensor(res) if res is not None else -1)
                bn_buffers.append((rm, rv))
                bn_modules.append(module)
            elif kind == "conv":
                c1, x, (W, b) = e[2], e[3], e[4]
                r["kind"] = _lib.OP_CONV
                conv_fields(r, c1, x, y, W, b)
            elif kind == "relu":
                xi = tensor(e[2])
                yi = tensor(y, create=True)
                level[yi], chans[yi] = level[xi], chans[xi]
                r.update(kind=_lib.OP_RELU, x=xi, y=yi, level_in=level[xi], level_out=level[xi], cin=chans[xi],
                         cout=chans[xi])
            elif kind == "cat":
                if len(e[2]) != 2:
                    raise ValueError("ME.cat of exactly two tensors")
                a, b = tensor(e[2][0]), tensor(e[2][1])
                yi = tensor(y, create=True)
                level[yi], chans[yi] = level[a], chans[a] + chans[b]
                if level[a] != level[b]:
                    raise ValueError("ME.cat across levels")
                r.update(kind=_lib.OP_CAT, x=a, x2=b, y=yi, level_in=level[a], level_out=level[a], cin=chans[a],
                         cout=chans[a] + chans[b])
            elif kind == "rownorm":
                xi = tensor(e[3])
                yi = tensor(y, create=True)
                level[yi], chans[yi] = level[xi], chans[xi]
                r.update(kind=_lib.OP_ROWNORM, x=xi, y=yi, level_in=level[xi], level_out=level[xi], cin=chans[xi],
                         cout=chans[xi])
            else:
                raise ValueError(f"tape entry '{kind}' has no plan record")
            records.append(r)
        group = getattr(model, "_amax_group", None)
        worder = [pid[id(p)] for p in group.params] if group else []
        plan = cls(records, len(tid), params, bn_buffers, bn_modules, worder, keys, chans[records[-1]["y"]])
        # a TRAINING pass through the plan writes the gradient of every record's parameters through raw pointers and
        # the trainer seats / updates every entry of model.parameters(): that is only right when each parameter is
        # claimed by a record (an unclaimed one would keep a stale seat and still receive weight decay)
        claimed = {r[f] for r in records for f in ("w", "bias", "bn_w", "bn_b") if r[f] >= 0}
        plan.covers_all_params = claimed == set(range(len(params)))
        return plan

    # ---- running -------------------------------------------------------------------------------------------------
    def _pointers(self):
        key = tuple(p.data_ptr() for p in self.params)
        if key != self._ptr_key:
            if any(p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
                raise ValueError("plan parameters must be contiguous fp32 tensors")
            self._ptr_key = key
            self._param_ptrs = (ctypes.c_void_p * len(key))(*key)
            bn = []
            for rm, rv in self.bn_buffers:
                bn += [rm.data_ptr(), rv.data_ptr()]
            self._bn_ptrs = (ctypes.c_void_p * max(1, len(bn)))(*bn)
        return self._param_ptrs, self._bn_ptrs

    def run(self, x_feats, maps):
        """Forward pass on ``maps`` (NativeMaps built with this plan's spec order).  Returns the output feature tensor
        (a view of the pass's arena), attached to autograd when gradients are enabled."""
        lib = _lib.require_gpu()
        if maps.keys != self.spec_keys:
            raise ValueError("the maps were built for a different map specification than the plan")
        self._check_input(x_feats, maps)
        x = x_feats.contiguous()
        dev = x.device
        need = lib.gcl_plan_arena_bytes(self.handle, ctypes.byref(maps.desc))
        if need < 0:
            raise RuntimeError("gcl_plan_arena_bytes: " + lib.gcl_last_error().decode())
        arena = torch.empty(int(need), dtype=torch.uint8, device=dev)
        self._ensure_state(lib, dev)
        pp, bp = self._pointers()
        if AUX_STREAM and self._aux is None:
            # weight gradients run on a second (lower-priority) stream beside the input-gradient chain
            lo, _hi = torch.cuda.Stream.priority_range()
            pct = int(os.environ.get("GCL_AUX_CU_PCT", "0"))
            if 0 < pct < 100:      # measurement: the weight gradients on a share of the CUs
                h = ctypes.c_void_p()
                _lib.check(lib.gcl_stream_create_cu_share(pct, 1, ctypes.byref(h)), "gcl_stream_create_cu_share")
                self._aux = torch.cuda.ExternalStream(h.value, device=dev)
            else:
                self._aux = torch.cuda.Stream(device=dev, priority={"low": lo, "high": _hi}.get(AUX_STREAM, 0))
            lib.gcl_plan_set_aux_stream(self.handle, ctypes.c_void_p(self._aux.cuda_stream))
        if self.profile_next:
            lib.gcl_plan_profile(self.handle, 1)
        y_ptr = ctypes.c_void_p()
        _lib.check(lib.gcl_plan_forward(self.handle, ctypes.byref(maps.desc), _lib.ptr(x, torch.float32), pp, bp,
                                        _lib.ptr(self._state), _lib.ptr(arena), arena.numel(), ctypes.byref(y_ptr),
                                        _lib.stream()), "gcl_plan_forward")
        for m in self.bn_modules:          # what MinkowskiBatchNorm.forward keeps on the host per training forward
            m._pending_batches += 1
            m._train_forwards += 1
        n_out = int(maps.desc.n_rows[self.records[-1]["level_out"]])
        off = y_ptr.value - arena.data_ptr()
        y = arena[off:off + n_out * self.out_channels * 4].view(torch.float32).view(n_out, self.out_channels)
        if not torch.is_grad_enabled():       # no backward pass will come: un-park the pass
            lib.gcl_plan_release(self.handle, _lib.ptr(arena))
            return y
        run = _PlanRun()
        run.plan, run.arena, run.maps, run.y, run.x = self, arena, maps, y, x
        run.grad_targets, run.done = self.grad_targets, False
        run.segments = self._segments()
        if run.grad_targets is not None:
            # seated gradients are written in place: the parameters stay out of the autograd node (autograd would run their
            # accumulation hooks even for an undefined gradient); a private leaf makes the node differentiable
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            return _PlanFn.apply(run, self._anchor)
        return _PlanFn.apply(run, *self.params)

    def _check_input(self, x_feats, maps):
        r0 = self.records[0]
        if x_feats.dim() != 2 or x_feats.shape[1] != r0["cin"]:
            raise ValueError(f"the plan was recorded for {r0['cin']} input channels, got a tensor of shape {tuple(x_feats.shape)}")
        if x_feats.shape[0] != int(maps.desc.n_rows[r0["level_in"]]):
            raise ValueError("features and coordinate maps differ in length")

    def _ensure_state(self, lib, dev):
        if self._state is None or self._state.device != dev:     # tables + (inference) the persistent packed kernels
            self._state = torch.empty(int(lib.gcl_plan_eval_state_bytes(self.handle)), dtype=torch.uint8, device=dev)
            self._eval_key = None

    def prepare_eval(self, dev):
        """Everything an inference pass needs that does NOT depend on the maps (parameter pointers, the BatchNorm modules'
        eval-mode (scale, shift), the re-pack key): ~ 110 us of Python per pass.  ResUNet2._forward_eval calls it BEFORE the
        native map build, whose read-back the host waits for anyway -- behind the build the device's queue is empty, and
        every microsecond of Python between the build and gcl_plan_forward_eval is a microsecond of an idle GPU."""
        from . import ops
        lib = _lib.require_gpu()
        self._ensure_state(lib, dev)
        pp, _ = self._pointers()
        # BatchNorm in eval mode: (scale, shift) cached by the modules, (mean, rstd) for the un-fused first layer
        aff, ptrs = [], []
        for m in self.bn_modules:
            scale, shift = m.eval_affine()
            rstd = getattr(m, "_eval_rstd", None)
            if rstd is None or rstd[0] is not scale:
                with torch.no_grad():
                    rstd = m._eval_rstd = (scale, torch.rsqrt(m.bn.running_var.detach() + m.bn.eps).contiguous())
            aff.append((scale, shift, rstd[1]))
            ptrs += [scale.data_ptr(), shift.data_ptr(), m.bn.running_mean.data_ptr(), rstd[1].data_ptr()]
        bn = (ctypes.c_void_p * len(ptrs))(*ptrs)
        # the packed kernels persist in the state buffer: re-pack when a parameter (or the amax epoch) changed
        key = (tuple(p._version for p in self.params), tuple(p.data_ptr() for p in self.params), ops._AMAX_EPOCH)
        return pp, bn, aff, key, dev

    def run_eval(self, x_feats, maps, prepared=None):
        """Inference pass (model.eval(), torch.no_grad()): ONE gcl_plan_forward_eval call; every conv + BatchNorm
        (+ residual)(+ ReLU) of the network is one fused launch.  Bitwise equal to the per-operator eval path.
        ``prepared``: the result of ``prepare_eval`` made earlier in the same (no-grad, single-threaded) pass."""
        lib = _lib.require_gpu()
        if maps.keys != self.spec_keys:
            raise ValueError("the maps were built for a different map specification than the plan")
        self._check_input(x_feats, maps)
        x = x_feats.contiguous()
        dev = x.device
        if prepared is None or prepared[4] != dev:
            prepared = self.prepare_eval(dev)
        pp, bn, aff, key, _ = prepared
        need = lib.gcl_plan_eval_arena_bytes(self.handle, ctypes.byref(maps.desc))
        if need < 0:
            raise RuntimeError("gcl_plan_eval_arena_bytes: " + lib.gcl_last_error().decode())
        arena = torch.empty(int(need), dtype=torch.uint8, device=dev)
        repack = int(key != self._eval_key)
        y_ptr = ctypes.c_void_p()
        _lib.check(lib.gcl_plan_forward_eval(self.handle, ctypes.byref(maps.desc), _lib.ptr(x, torch.float32), pp, bn, repack,
                                             _lib.ptr(self._state), _lib.ptr(arena), arena.numel(), ctypes.byref(y_ptr),
                                             _lib.stream()), "gcl_plan_forward_eval")
        self._eval_key = key
        n_out = int(maps.desc.n_rows[self.records[-1]["level_out"]])
        off = y_ptr.value - arena.data_ptr()
        return arena[off:off + n_out * self.out_channels * 4].view(torch.float32).view(n_out, self.out_channels)

    def _segments(self):
        """[(first_op, last_op, [buckets complete after it])], highest records first.  A bucket is complete once the
        backward pass has gone through the LOWEST record that holds one of its parameters."""
        n = len(self.records)
        if self.bucket_of_param is None or self.on_bucket is None:
            return [(0, n, [])]
        lowest = {}
        for i, r in enumerate(self.records):
            for k in ("w", "bias", "bn_w", "bn_b"):
                p = r[k]
                if p >= 0 and p in self.bucket_of_param:
                    b = self.bucket_of_param[p]
                    lowest[b] = min(lowest.get(b, i), i)
        cuts = sorted(set(lowest.values()), reverse=True)
        segs, hi = [], n
        for c in cuts:
            segs.append((c, hi, sorted(b for b, i in lowest.items() if i == c)))
            hi = c
        if hi > 0:
            segs.append((0, hi, []))
        return segs

    def _backward(self, run, dy):
        lib = _lib.load()
        targets = run.grad_targets
        flat = None
        if targets is None:      # gradients go to autograd: fresh memory, never aliased with an existing p.grad
            sizes = [p.numel() for p in self.params]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dy.device)
            targets, off = [], 0
            for p, s in zip(self.params, sizes):
                targets.append(flat[off:off + s].view_as(p))
                off += s
        gp = (ctypes.c_void_p * len(targets))(*[t.data_ptr() for t in targets])
        st = _lib.stream()
        for first, last, buckets in run.segments:
            _lib.check(lib.gcl_plan_backward(self.handle, _lib.ptr(run.arena), _lib.ptr(dy, torch.float32), gp, first,
                                             last, st), "gcl_plan_backward")
            for b in buckets:
                self.on_bucket(b)
        if self.profile_next:
            self.profile_next = False
            self._profile_pending = True
            lib.gcl_plan_profile(self.handle, 0)
        run.done = True
        run.arena = run.maps = run.y = run.x = None
        return None if flat is None else targets

    def profile_records(self):
        """Per-launch records of the profiled pass (call after synchronising the stream): list of
        (kind, ms, pairs, cin, cout, n_in, n_out, K); kind 0 = forward / input gradient, 1 = input gradient with the fused
        gradient add, 2 = weight gradient over pair lists, 3 = weight gradient of a kernel_size-1 convolution (row stream)."""
        lib = _lib.load()
        buf = (ctypes.c_double * (8 * 1024))()
        n = lib.gcl_plan_profile_read(self.handle, buf, 1024)
        return [tuple(buf[8 * i + j] for j in range(8)) for i in range(max(0, n))]
