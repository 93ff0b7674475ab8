"""SparseTensor + coordinate manager (host side of the coordinate/kernel-map kernels in csrc/coords.hip)."""
import os
import weakref

import torch

from .. import _lib


# rows are mask-sorted inside windows of this many consecutive rows (0 = one global sort); tuning knob
SORT_WINDOW = int(os.environ.get("GCL_SORT_WINDOW", "0"))
# TUNING KNOB, default off (0).  Levels whose tensor stride is <= SPATIAL_MAX_STRIDE mask-sort their tables inside
# windows of SPATIAL_WINDOW rows of a SPATIAL pre-order (cloud, Morton cell; gcl_spatial_order) and run with one
# contiguous tile range per XCD, so that the rows in flight on an XCD are spatial neighbours and find each other's
# gathered input rows in L2.  Measured (profiles/r02_spatial_micro.txt): much better than windows of the loader order
# (64->64 @s1: 274 vs 339 us) but no better than the global mask sort (247 us) -- the extra (offset, slice) steps of
# the less uniform tiles cost what the L2 hits save: the kernels are bound by the latency of a step, not by bytes.
SPATIAL_MAX_STRIDE = int(os.environ.get("GCL_SPATIAL_MAX_STRIDE", "0"))
SPATIAL_WINDOW = int(os.environ.get("GCL_SPATIAL_WINDOW", "4096"))
SPATIAL_MIN_ROWS = int(os.environ.get("GCL_SPATIAL_MIN_ROWS", "32768"))     # single clouds stay on the global sort


def _pow2_cap(n):
    cap = 64
    while cap < 2 * n:
        cap *= 2
    return cap


class CoordinateMapKey:
    """Identifies a coordinate map by its tensor stride (one map per stride per manager)."""

    __slots__ = ("tensor_stride",)

    def __init__(self, tensor_stride=1):
        self.tensor_stride = int(tensor_stride)

    def get_tensor_stride(self):
        return [self.tensor_stride] * 3

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and other.tensor_stride == self.tensor_stride

    def __hash__(self):
        return hash(self.tensor_stride)

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.tensor_stride})"


class KernelMap:
    """Forward kernel map (in stride t_in -> out stride t_in * stride), k-major neighbour tables.

    nbr   [K, n_out] int32: input row of out row v at offset k, -1 if absent
    nbr_t [K, n_in]  int32: out row reached from input row u through offset k (None when in == out map:
                            there nbr_t[k] == nbr[K-1-k])
    pair lists (built lazily, only the weight gradient needs them): per offset compacted (in, out) rows,
    each offset segment padded with -1 to a multiple of GCL_PAIR_CHUNK.
    """

    def __init__(self, nbr, nbr_t, counts_dev, n_in, n_out, K, mgr=None, t_in=1, t_out=1):
        self.nbr, self.nbr_t = nbr, nbr_t
        self._mgr = weakref.ref(mgr) if mgr is not None else (lambda: None)   # no manager <-> map reference cycle
        self._t_in, self._t_out = t_in, t_out
        self.n_in, self.n_out, self.K = n_in, n_out, K
        self.same_map = nbr_t is None
        self._pairs = None
        self._sorted = {}
        # per-offset pair counts: asynchronous copy into pinned memory; the host only waits for it when the weight
        # gradient first needs the pair lists (long after the copy has completed) -- no sync in the forward pass
        self._counts_dev = counts_dev
        self._counts_host = torch.empty(K, dtype=torch.int32, pin_memory=True)
        self._counts_host.copy_(counts_dev, non_blocking=True)
        self._counts_event = torch.cuda.Event()
        self._counts_event.record()
        self._counts = None

    @property
    def counts(self):
        if self._counts is None:
            self._counts_event.synchronize()
            self._counts = self._counts_host.tolist()
        return self._counts

    @property
    def n_pairs(self):
        return int(sum(self.counts))

    def sorted_table(self, transposed=False):
        """(tbl_sorted, order, tile_mask) of ``nbr`` (or ``nbr_t``): rows re-ordered by neighbour-presence mask so
        that every 32-row wave tile of the convolution kernel visits few offsets (gcl_table_sort).  K <= 27 only."""
        key = "t" if transposed else "n"
        if key not in self._sorted:
            tbl = self.nbr_t if transposed else self.nbr
            if self.K > 27:
                self._sorted[key] = (tbl, None, None)
            else:
                lib = _lib.load()
                n = tbl.shape[1]
                dev = tbl.device
                scratch = torch.empty(lib.gcl_table_sort_scratch_len(n), dtype=torch.int32, device=dev)
                order = torch.empty(n, dtype=torch.int32, device=dev)
                tbl_sorted = torch.empty_like(tbl)
                tile_mask = torch.empty((n + 31) // 32, dtype=torch.int32, device=dev)
                t_rows = self._t_in if transposed else self._t_out          # the level whose rows this table lists
                mgr = self._mgr()
                spatial = mgr is not None and 0 < t_rows <= SPATIAL_MAX_STRIDE and n >= SPATIAL_MIN_ROWS
                pre = mgr.spatial_order(t_rows) if spatial else None
                _lib.check(lib.gcl_table_sort_pre(_lib.ptr(tbl), self.K, n, SPATIAL_WINDOW if spatial else SORT_WINDOW,
                                                  _lib.ptr(pre), _lib.ptr(scratch), _lib.ptr(order), _lib.ptr(tbl_sorted),
                                                  _lib.ptr(tile_mask), _lib.stream()), "gcl_table_sort")
                tbl_sorted._gcl_flags = 1 if spatial else 0                # GCL_CONV_XCD_RANGES
                self._sorted[key] = (tbl_sorted, order, tile_mask)
        return self._sorted[key]

    def pairs(self):
        if self._pairs is None:
            lib = _lib.load()
            ch = _lib.PAIR_CHUNK
            seg = [0]
            for c in self.counts:
                seg.append(seg[-1] + (c + ch - 1) // ch * ch)
            total = seg[-1]
            dev = self.nbr.device
            both = torch.empty(2 * max(total, 1), dtype=torch.int32, device=dev)    # adjacent: one fill for both
            pair_in, pair_out = both[:max(total, 1)], both[max(total, 1):]
            nb = (self.n_out + 1023) // 1024
            scratch = torch.empty(self.K * nb + self.K + 1, dtype=torch.int32, device=dev)
            seg_host = _lib.host_i64(seg)
            _lib.check(lib.gcl_kernel_map_pairs(_lib.ptr(self.nbr), self.K, self.n_out, seg_host, _lib.ptr(scratch),
                                                _lib.ptr(pair_in), _lib.ptr(pair_out), _lib.stream()),
                       "gcl_kernel_map_pairs")
            self._pairs = (pair_in, pair_out, seg, seg_host)
        return self._pairs


class CoordinateManager:
    """Owns the coordinate maps (one per tensor stride) and caches kernel maps per (t_in, kernel_size, stride),
    so the two convs of a residual block and matching encoder/decoder levels share one map."""

    native = None       # NativeMaps when the maps were built by ONE gcl_maps_build call (build_native)

    @classmethod
    def build_native(cls, coordinates, specs, n_levels=4, arena=None, side_stream=None):
        """A manager whose stride maps, kernel maps, mask-sorted tables and pair lists were all built by one native call
        (``specs`` as in ``prefetch`` + (t, 1, 1, (), True) for kernel_size-1 identity pairs; native.NativeMaps).  The
        whole-network plan reads the native descriptor directly; the Python accessors below (get_coords,
        get_kernel_map, identity_pairs) hand out tensor VIEWS of the same arena on demand, so the per-operator path
        runs on exactly the same maps."""
        from .native import NativeMaps
        nm = NativeMaps(coordinates, specs, n_levels, arena, side_stream=side_stream)
        self = cls.__new__(cls)
        self.native, self.device = nm, nm.device
        self._segments, self._maps, self._status = {}, {}, {}
        self._checked = set(1 << l for l in range(nm.n_levels))       # gcl_maps_build has validated the coordinates
        self._kmaps, self._identity, self._spatial, self._bitmap = {}, {}, {}, None
        return self

    def _native_level(self, t):
        nm = self.native
        l = int(t).bit_length() - 1
        if nm is None or (1 << l) != t or l >= nm.n_levels:
            return False
        d = nm.desc
        n, cap = int(d.n_rows[l]), int(d.cap[l])
        self._maps[t] = (nm.view(d.coords[l], (n, 4), torch.int32), nm.view(d.table[l], (cap, 2), torch.int64), cap)
        return True

    def _native_kernel_map(self, key):
        nm = self.native
        if nm is None or key not in nm.keys or key[1] == 1:
            return None
        m = nm.desc.maps[nm.keys.index(key)]
        K, n_in, n_out = int(m.K), int(m.n_in), int(m.n_out)
        i32 = torch.int32
        km = KernelMap.__new__(KernelMap)
        km.nbr, km.nbr_t = nm.view(m.nbr, (K, n_out), i32), nm.view(m.nbr_t, (K, n_in), i32)
        km._mgr = weakref.ref(self)
        km._t_in, km._t_out = key[0], key[0] * key[2]
        km.n_in, km.n_out, km.K = n_in, n_out, K
        km.same_map = km.nbr_t is None
        km._counts_dev = nm.view(m.counts, (K,), i32)
        if m.pair_in or int(m.n_pairs) > 0:
            km._counts = [int(m.counts_host[k]) for k in range(K)]
        else:     # built without pair lists (inference): the counts were never read back -- fetch them lazily like __init__
            km._counts = None
            km._counts_host = torch.empty(K, dtype=torch.int32, pin_memory=True)
            km._counts_host.copy_(km._counts_dev, non_blocking=True)
            km._counts_event = torch.cuda.Event()
            km._counts_event.record()
        km._sorted, km._pairs = {}, None
        for tag, (a, b, c), rows in (("n", (m.tbl_n, m.order_n, m.mask_n), n_out), ("t", (m.tbl_t, m.order_t, m.mask_t), n_in)):
            if a:
                tbl = nm.view(a, (K, rows), i32)
                tbl._gcl_flags = 0
                km._sorted[tag] = (tbl, nm.view(b, (rows,), i32), nm.view(c, ((rows + 31) // 32,), i32))
        if m.pair_in:
            seg = [int(m.seg_off[k]) for k in range(K + 1)]
            total = max(seg[-1], 1)
            km._pairs = (nm.view(m.pair_in, (total,), i32), nm.view(m.pair_out, (total,), i32), seg, _lib.host_i64(seg))
        return km

    def __init__(self, coordinates):
        self._segments = {}
        lib = _lib.require_gpu()
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N, 4] = (batch, x, y, z)")
        C = coordinates.to(torch.int32).contiguous()
        n = C.shape[0]
        if n == 0:
            raise ValueError("empty SparseTensor")
        self.device = C.device
        # the input level's hash table is made on first use (_input_table): an inference pass that hands the coordinates to
        # ONE native map build (ResUNet2._forward_eval -> build_native) never reads this manager's own table -- three
        # launches and two allocations per pass
        self._maps = {1: (C, None, _pow2_cap(n))}
        self._status = {}
        self._checked = set()
        self._kmaps = {}
        self._identity = {}
        self._spatial = {}
        self._bitmap = None

    # -- coordinate maps -----------------------------------------------------------------------------------
    def _input_table(self):
        """Inserts the input coordinates into their hash table (once; CoordinateManager.__init__ defers it)."""
        C, table, cap = self._maps[1]
        if table is None:
            lib = _lib.require_gpu()
            table = torch.empty((cap, 2), dtype=torch.int64, device=self.device)
            status = torch.empty(4, dtype=torch.int32, device=self.device)
            _lib.check(lib.gcl_coords_insert(_lib.ptr(C), C.shape[0], _lib.ptr(table), cap, _lib.ptr(status), _lib.stream()),
                       "gcl_coords_insert")
            self._maps[1] = (C, table, cap)
            self._status[1] = status
        return table

    def _check_status(self, t):
        if t in self._checked:
            return
        if t == 1:
            self._input_table()
        self._raise_on_status(self._status[t].tolist())
        self._checked.add(t)

    @staticmethod
    def _raise_on_status(st):
        if st[0]:
            raise ValueError(f"{st[0]} coordinates outside the packable range (batch < 65535, |x|,|y|,|z| < 32768)")
        if st[1]:
            raise ValueError(f"{st[1]} duplicate coordinates: ME.SparseTensor expects unique rows "
                             "(use ME.utils.sparse_quantize)")

    def get_coords(self, t):
        if t not in self._maps and not self._native_level(t):
            if self.native is not None:
                raise ValueError(f"tensor stride {t} was not part of the native map build")
            self._build_stride_maps(t)
        return self._maps[t][0]

    def num_rows(self, t):
        return self.get_coords(t).shape[0]

    def spatial_order(self, t):
        """Rows of the level at tensor stride ``t`` in (cloud, Morton cell) order (gcl_spatial_order); cached."""
        if t not in self._spatial:
            lib = _lib.load()
            C = self.get_coords(t)
            n = C.shape[0]
            scratch = torch.empty(lib.gcl_table_sort_scratch_len(n), dtype=torch.int32, device=self.device)
            order = torch.empty(n, dtype=torch.int32, device=self.device)
            _lib.check(lib.gcl_spatial_order(_lib.ptr(C), n, t, _lib.ptr(scratch), _lib.ptr(order), _lib.stream()),
                       "gcl_spatial_order")
            self._spatial[t] = order
        return self._spatial[t]

    def batch_segments(self, t):
        """[(first row, rows)] of every cloud at tensor stride ``t`` (InstanceNorm normalises per cloud).  Rows of one
        cloud must be contiguous, which holds for collated input and for every strided map derived from it (first-
        occurrence order); anything else is rejected.  One host read per level, cached."""
        if t not in self._segments:
            b = self.get_coords(t)[:, 0]
            vals, counts = torch.unique_consecutive(b, return_counts=True)
            vals, counts = vals.tolist(), counts.tolist()
            if len(set(vals)) != len(vals):
                raise NotImplementedError("MinkowskiInstanceNorm needs the rows of each cloud to be contiguous")
            seg, start = [], 0
            for c in counts:
                seg.append((start, c))
                start += c
            self._segments[t] = seg
        return self._segments[t]

    def _build_stride_maps(self, t):
        """Builds every missing power-of-two level up to max(t, 8) in ONE chain of launches: level 2s is built from
        level s with the row count of level s still on the device, and all counts come back in a single D2H read
        (one host sync per SparseTensor instead of one per level)."""
        lib = _lib.load()
        if self.native is None:
            self._input_table()
        levels = []
        s = max(self._maps)
        while s < max(t, 8):
            s *= 2
            levels.append(s)
        if t not in levels:
            raise ValueError(f"tensor stride {t} is not a power-of-two multiple of the existing maps")
        base_t = max(self._maps)
        Cb = self._maps[base_t][0]
        n_bound = Cb.shape[0]
        meta = torch.zeros(8 * len(levels), dtype=torch.int32, device=self.device)   # per level: [0]=n_out, [4:8]=status
        built = []
        n_dev = None
        for li, lv in enumerate(levels):
            cap = _pow2_cap(n_bound)
            table = torch.empty((cap, 2), dtype=torch.int64, device=self.device)
            scratch = torch.empty(lib.gcl_scan_scratch_len(n_bound), dtype=torch.int32, device=self.device)
            out = torch.empty((n_bound, 4), dtype=torch.int32, device=self.device)
            _lib.check(lib.gcl_stride_map(_lib.ptr(Cb), n_bound, n_dev, lv, _lib.ptr(table), cap, _lib.ptr(scratch),
                                          _lib.ptr(out), ctypes_offset(meta, 8 * li), ctypes_offset(meta, 8 * li + 4),
                                          _lib.stream()), "gcl_stride_map")
            built.append((lv, out, table, cap))
            Cb, n_dev = out, ctypes_offset(meta, 8 * li)
        first = 1 not in self._checked
        vals = (torch.cat([self._status[1], meta]) if first else meta).tolist()   # the ONE D2H sync of the maps
        if first:
            self._raise_on_status(vals[:4])
            self._checked.add(1)
            vals = vals[4:]
        m = vals
        for li, (lv, out, table, cap) in enumerate(built):
            if m[8 * li + 4]:
                raise ValueError(f"{m[8 * li + 4]} strided coordinates outside the packable range")
            self._maps[lv] = (out[:m[8 * li]], table, cap)
            self._checked.add(lv)

    # -- kernel maps -----------------------------------------------------------------------------------------
    def get_kernel_map(self, t_in, kernel_size, stride):
        key = (t_in, kernel_size, stride)
        if key in self._kmaps:
            return self._kmaps[key]
        if self.native is not None:
            km = self._native_kernel_map(key)
            if km is None:
                raise ValueError(f"kernel map {key} was not part of the native map build (model.native_map_specs)")
            self._kmaps[key] = km
            return km
        lib = _lib.load()
        if 1 not in self._checked:
            # first map of this tensor: build the whole stride pyramid now, so that the validity check of the input
            # coordinates and all level sizes share ONE host sync
            self._build_stride_maps(max(8, t_in * stride))
        t_out = t_in * stride
        self.get_coords(t_in)
        self._input_table()
        C_in, table_in, cap_in = self._maps[t_in]
        C_out = self.get_coords(t_out)
        n_in, n_out = C_in.shape[0], C_out.shape[0]
        K = kernel_size ** 3
        nbr = torch.empty((K, n_out), dtype=torch.int32, device=self.device)
        same = stride == 1
        nbr_t = None if same else torch.empty((K, n_in), dtype=torch.int32, device=self.device)
        counts = torch.empty(K, dtype=torch.int32, device=self.device)
        if self._bitmap is None:
            self._bitmap = {}
        bitmap_valid = t_in in self._bitmap            # one presence bitmap per coordinate table, filled once
        if not bitmap_valid:
            self._bitmap[t_in] = torch.empty(lib.gcl_kernel_map_bitmap_len(), dtype=torch.int32, device=self.device)
        scratch = torch.empty(lib.gcl_kernel_map_scratch_len(kernel_size, n_out), dtype=torch.int32, device=self.device)
        _lib.check(lib.gcl_kernel_map(_lib.ptr(C_out), n_out, _lib.ptr(table_in), cap_in, kernel_size, t_in,
                                      int(same) | (2 if bitmap_valid else 0), _lib.ptr(self._bitmap[t_in]),
                                      _lib.ptr(scratch), _lib.ptr(nbr),
                                      _lib.ptr(nbr_t), n_in, _lib.ptr(counts), _lib.stream()), "gcl_kernel_map")
        km = KernelMap(nbr, nbr_t, counts, n_in, n_out, K, self, t_in, t_out)
        self._kmaps[key] = km
        return km

    # -- loader-side prefetch -----------------------------------------------------------------------------------
    def prefetch(self, specs):
        """Builds, on the CURRENT stream, everything a network will ask this manager for: ``specs`` = iterable of
        (t_in, kernel_size, stride, tables, pairs) with ``tables`` a tuple of ``transposed`` flags for
        KernelMap.sorted_table and ``pairs`` whether the weight gradient's pair lists are needed.  A trainer calls this
        for batch i+1 on a side stream while batch i trains (the maps depend on the coordinates only), which takes the
        ~150 small integer launches and the level-size read-back off the training stream's critical path."""
        for t_in, ks, stride, tables, pairs in (s[:5] for s in specs):
            km = self.get_kernel_map(t_in, ks, stride)
            for tr in tables:
                km.sorted_table(transposed=bool(tr))
            if pairs:
                km.pairs()
        return self

    def device_tensors(self):
        """Every device tensor this manager holds (for ``record_stream`` when it was built on another stream)."""
        out = []
        if self.native is not None:      # one arena holds every map
            return [self.native.arena, self.native.coords]
        for C, table, _ in self._maps.values():
            out += [t for t in (C, table) if t is not None]
        out += list(self._status.values()) + list((self._bitmap or {}).values()) + list(self._spatial.values())
        for km in self._kmaps.values():
            out += [t for t in (km.nbr, km.nbr_t, km._counts_dev) if t is not None]
            for tup in km._sorted.values():
                out += [t for t in tup if t is not None]
            if km._pairs is not None:
                out += [km._pairs[0], km._pairs[1]]
        for p in self._identity.values():
            out.append(p[0])
        return [t for t in out if isinstance(t, torch.Tensor) and t.is_cuda]

    def identity_pairs(self, n):
        """Pair lists of a kernel_size-1 convolution (row i <-> row i), padded to GCL_PAIR_CHUNK."""
        if n not in self._identity and self.native is not None:
            nm = self.native
            for i, key in enumerate(nm.keys):
                m = nm.desc.maps[i]
                if key[1] == 1 and int(m.n_in) == n and m.pair_in:
                    seg = [0, int(m.seg_off[1])]
                    p = nm.view(m.pair_in, (seg[1],), torch.int32)
                    self._identity[n] = (p, p, seg, _lib.host_i64(seg))
                    break
        if n not in self._identity:
            ch = _lib.PAIR_CHUNK
            total = (n + ch - 1) // ch * ch
            p = torch.full((total,), -1, dtype=torch.int32, device=self.device)
            p[:n] = torch.arange(n, dtype=torch.int32, device=self.device)
            seg = [0, total]
            self._identity[n] = (p, p, seg, _lib.host_i64(seg))
        return self._identity[n]

    def kernel_map_triples(self, t_in, kernel_size, stride):
        """(k, in, out) triples on the host -- for parity tests against the oracle only."""
        km = self.get_kernel_map(t_in, kernel_size, stride)
        nbr = km.nbr.cpu()
        k, v = torch.nonzero(nbr >= 0, as_tuple=True)
        return torch.stack([k, nbr[k, v].long(), v], dim=1).numpy()


def ctypes_offset(t, elem):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + elem * t.element_size())


class SparseTensor:
    """``ME.SparseTensor``: features ``F`` [N, C] fp32 on coordinates ``C`` [N, 4] int32 (batch, x, y, z).

    Constructors used by the reference:
      SparseTensor(feats, coordinates=coords[, device=...])                  (lib/colocation_trainer.py:843-845)
      SparseTensor(feats, coordinate_map_key=key, coordinate_manager=mgr)    (model/resunet.py:227-230)
    Row i of ``F`` stays row i of ``C`` (the loss indexes F_out by loader row ids, :465).
    """

    def __init__(self, features, coordinates=None, coordinate_map_key=None, coordinate_manager=None, device=None,
                 tensor_stride=1):
        if coordinates is not None:
            if device is not None:
                features, coordinates = features.to(device), coordinates.to(device)
            elif coordinates.is_cuda and not features.is_cuda:
                features = features.to(coordinates.device)
            elif features.is_cuda and not coordinates.is_cuda:
                coordinates = coordinates.to(features.device)
            if not features.is_cuda:
                raise RuntimeError("gcl_amd.MinkowskiEngine has no CPU backend: move features/coordinates to the GPU "
                                   "(ME.SparseTensor(feats.to(device), coordinates=coords.to(device)))")
            if features.shape[0] != coordinates.shape[0]:
                raise ValueError("features and coordinates differ in length")
            if coordinate_manager is None:
                coordinate_manager = CoordinateManager(coordinates)
            elif coordinate_manager.num_rows(tensor_stride) != coordinates.shape[0]:     # a manager prefetched for these rows
                raise ValueError("coordinate_manager was built for a different coordinate set")
            coordinate_map_key = CoordinateMapKey(tensor_stride)
        elif coordinate_map_key is None or coordinate_manager is None:
            raise ValueError("either coordinates or (coordinate_map_key, coordinate_manager) is required")
        if features.dtype != torch.float32:
            features = features.float()
        self._F = features
        self.coordinate_map_key = coordinate_map_key
        self.coordinate_manager = coordinate_manager
        self._nonneg = False

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.get_coords(self.coordinate_map_key.tensor_stride)

    @property
    def coordinates(self):
        return self.C

    @property
    def features(self):
        return self._F

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def device(self):
        return self._F.device

    @property
    def shape(self):
        return self._F.shape

    def __len__(self):
        return self._F.shape[0]

    def _same_map(self, other):
        if (other.coordinate_manager is not self.coordinate_manager
                or other.coordinate_map_key != self.coordinate_map_key):
            raise ValueError("SparseTensors live on different coordinate maps")

    def __iadd__(self, other):          # `out += residual` (model/residual_block.py:50)
        self._same_map(other)
        from . import ops
        ops.tape_guard("SparseTensor.__iadd__", self._F, other.F)
        self._F = self._F + other.F
        self._nonneg = False
        self._bn_stats = None          # column sums published by a convolution describe the OLD features
        return self

    def __add__(self, other):
        self._same_map(other)
        from . import ops
        ops.tape_guard("SparseTensor.__add__", self._F, other.F)
        return SparseTensor(self._F + other.F, coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self.coordinate_manager)

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, {self.coordinate_map_key})"


def cat(*tensors):
    """``ME.cat(a, b)``: channel concatenation of tensors on the same coordinate map (model/resunet.py:203)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    a = tensors[0]
    for b in tensors[1:]:
        a._same_map(b)
    from . import ops
    out = SparseTensor(ops.cat_features([t.F for t in tensors]), coordinate_map_key=a.coordinate_map_key,
                       coordinate_manager=a.coordinate_manager)
    out._nonneg = all(t._nonneg for t in tensors)
    return out
