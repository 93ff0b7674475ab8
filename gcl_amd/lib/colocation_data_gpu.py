"""Loader work on the GPU: voxelisation and co-location group building.

Replaces, with identical results, the CPU side of ``ColocationKittiDataset.__getitem__``
(lib/colocation_data_loader.py:378-421) that dominates the reference's data time:

* ``sparse_quantize_gpu``      -- ``ME.utils.sparse_quantize(xyz / voxel_size, return_index=True)`` (:379, :388)
* ``colocation_groups_gpu``    -- ``get_matching_indices_colocation`` (util/pointcloud.py:69-132; one open3d KD-tree
  radius query per point per cloud in a Python loop) as two kernels over the voxel hash map of the sample
* ``build_sample_gpu`` / ``collate_gpu`` -- the per-sample tuple and the batch dict of ``collate_colocation_fn``
  (:424-475) with every tensor already resident on the device (``index_hash`` is not produced: the GPU loss does not
  need it, see gcl_amd/lib/colocation_trainer.py).
* ``build_batch_gpu`` (round 6) -- the same batch dict from the raw scans of ``batch_size`` samples in ONE pass: all clouds
  through one coordinate table (two host synchronisations per batch instead of ~9 per sample, no host round trip for the
  neighbours' centre-frame points); ``train_from_scans`` feeds ``FinestContrastiveLossTrainer.train_steps`` from raw scans
  with that pass running a step ahead on its own stream.
"""
import ctypes
import os

import numpy as np
import torch

from gcl_amd import _lib


def _cap(n):
    cap = 64
    while cap < 2 * n:
        cap *= 2
    return cap


def unique_coords_gpu(raw, return_table=False):
    """First-occurrence unique rows of int32 [P,4] voxel coordinates on the GPU (gcl_unique_coords): (coords [N,4],
    index int64 [N] ascending) -- the ``return_index=True`` half of ``ME.utils.sparse_quantize``."""
    lib = _lib.require_gpu()
    raw = raw.contiguous()
    P = raw.shape[0]
    dev = raw.device
    cap = _cap(P)
    table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    scratch = torch.empty(lib.gcl_scan_scratch_len(P), dtype=torch.int32, device=dev)
    coords = torch.empty((P, 4), dtype=torch.int32, device=dev)
    index = torch.empty(P, dtype=torch.int64, device=dev)
    meta = torch.empty(8, dtype=torch.int32, device=dev)
    off = lambda t, e: ctypes.c_void_p(t.data_ptr() + e * t.element_size())
    _lib.check(lib.gcl_unique_coords(_lib.ptr(raw, torch.int32), P, _lib.ptr(table), cap, _lib.ptr(scratch),
                                     _lib.ptr(coords), _lib.ptr(index), off(meta, 0), off(meta, 4), _lib.stream()),
               "gcl_unique_coords")
    m = meta.tolist()
    if m[4]:
        raise ValueError(f"{m[4]} points outside the packable voxel range")
    n = m[0]
    if return_table:
        return coords[:n], index[:n], (table, cap)
    return coords[:n], index[:n]


def sparse_quantize_gpu(xyz, voxel_size, batch_id=0, return_table=False):
    """``xyz`` float32 [P,3] on the GPU -> (coords int32 [N,4] with ``batch_id`` in column 0, index int64 [N]):
    coords = floor(xyz / voxel_size), first occurrence per voxel, index ascending -- exactly
    ``ME.utils.sparse_quantize(xyz / voxel_size, return_index=True)`` + the batch column."""
    lib = _lib.require_gpu()
    xyz = xyz.contiguous()
    P = xyz.shape[0]
    raw = torch.empty((P, 4), dtype=torch.int32, device=xyz.device)
    _lib.check(lib.gcl_voxel_coords(_lib.ptr(xyz, torch.float32), P, float(voxel_size), int(batch_id), _lib.ptr(raw),
                                    _lib.stream()), "gcl_voxel_coords")
    return unique_coords_gpu(raw, return_table)


def colocation_groups_gpu(xyz_own, xyz_cf, coords, n_center, n_clouds, list_M, voxel_size, radius, K=5):
    """Groups of one sample.  ``xyz_own`` / ``xyz_cf`` float32 [Ntot,3] (voxel representatives, centre cloud first, in
    their own sensor frames / in the centre frame), ``coords`` int32 [Ntot,4] = floor(xyz_own / voxel) with the cloud
    id in column 0, ``list_M`` the neighbour -> centre transforms (4x4).  Returns device tensors
    (group int32 [G], index int64 [sum], finest_flag bool [sum]) with the reference's member order."""
    lib = _lib.require_gpu()
    dev = xyz_own.device
    ntot = xyz_own.shape[0]
    cap = _cap(ntot)
    table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    status = torch.empty(4, dtype=torch.int32, device=dev)
    coords = coords.contiguous()          # named, not temporaries inside the call: a temporary's block is reused at once
    xyz_own, xyz_cf = xyz_own.contiguous(), xyz_cf.contiguous()
    _lib.check(lib.gcl_coords_insert(_lib.ptr(coords, torch.int32), ntot, _lib.ptr(table), cap,
                                     _lib.ptr(status), _lib.stream()), "gcl_coords_insert")
    to_cloud = np.zeros((n_clouds, 12), dtype=np.float64)
    to_cloud[0] = np.eye(4)[:3].reshape(-1)
    for j, M in enumerate(list_M):
        to_cloud[j + 1] = np.linalg.inv(np.asarray(M, dtype=np.float64))[:3].reshape(-1)
    hits = torch.empty((n_center, n_clouds, K), dtype=torch.int32, device=dev)
    cnt = torch.empty((n_center, n_clouds), dtype=torch.int32, device=dev)
    rng = torch.empty((n_center, n_clouds), dtype=torch.float64, device=dev)
    _lib.check(lib.gcl_colocation_hits(_lib.ptr(xyz_own, torch.float32),
                                       _lib.ptr(xyz_cf, torch.float32), n_center, n_clouds,
                                       to_cloud.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _lib.ptr(table), cap,
                                       float(1.0 / voxel_size), float(radius), K, _lib.ptr(hits), _lib.ptr(cnt),
                                       _lib.ptr(rng), _lib.stream()), "gcl_colocation_hits")
    scratch = torch.empty(4 * n_center + n_center // 2048 + 64, dtype=torch.int32, device=dev)
    group = torch.empty(n_center, dtype=torch.int32, device=dev)
    index = torch.empty(n_center * n_clouds * K, dtype=torch.int64, device=dev)
    finest = torch.empty(n_center * n_clouds * K, dtype=torch.uint8, device=dev)
    totals = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.check(lib.gcl_colocation_emit(_lib.ptr(hits), _lib.ptr(cnt), _lib.ptr(rng), n_center, n_clouds, K,
                                       _lib.ptr(scratch), _lib.ptr(group), _lib.ptr(index), _lib.ptr(finest),
                                       _lib.ptr(totals), _lib.stream()), "gcl_colocation_emit")
    g, e = totals.tolist()
    st = status.tolist()
    if st[0] or st[1]:
        raise ValueError("coordinates of the sample are not unique / out of range")
    return group[:g], index[:e], finest[:e].bool()


def build_sample_gpu(xyz_list, list_M, voxel_size, radius, device, K=5):
    """One training sample from raw sensor-frame clouds (centre first): voxelise every cloud, transform the neighbours
    into the centre frame (host fp64 -> fp32, as the reference), build the groups.  Returns a dict of device tensors."""
    dev = torch.device(device)
    coords, xyz_own, xyz_cf = [], [], []
    for c, xyz in enumerate(xyz_list):
        x = torch.as_tensor(xyz, dtype=torch.float32, device=dev)
        cc, idx = sparse_quantize_gpu(x, voxel_size, batch_id=c)
        xo = x[idx]
        coords.append(cc)
        xyz_own.append(xo)
        if c == 0:
            xyz_cf.append(xo)
        else:
            M = np.asarray(list_M[c - 1], dtype=np.float64)
            xh = xo.cpu().numpy()
            xyz_cf.append(torch.from_numpy((xh @ M[:3, :3].T + M[:3, 3]).astype(np.float32)).to(dev))
    C, XO, XC = torch.cat(coords), torch.cat(xyz_own), torch.cat(xyz_cf)
    group, index, finest = colocation_groups_gpu(XO, XC, C, len(coords[0]), len(coords), list_M, voxel_size, radius, K)
    return {"coords": C, "xyz_own": XO, "group": group, "index": index, "finest_flag": finest,
            "n_rows": C.shape[0], "n_center": len(coords[0])}


def collate_gpu(samples):
    """``collate_colocation_fn`` (lib/colocation_data_loader.py:424-475) on device tensors: batch ids increment per
    cloud, ``index`` gets the sample's row offset, features are ones."""
    Cs, idx, grp, fin, lengths = [], [], [], [], []
    start, cloud0 = 0, 0
    for s in samples:
        C = s["coords"].clone()
        C[:, 0] += cloud0
        cloud0 = int(C[-1, 0].item()) + 1 if len(C) else cloud0
        Cs.append(C)
        idx.append(s["index"] + start)
        grp.append(s["group"])
        fin.append(s["finest_flag"])
        start += s["n_rows"]
        lengths.append(s["n_rows"])
    C = torch.cat(Cs)
    return {"sinput_C": C, "sinput_F": torch.ones((len(C), 1), dtype=torch.float32, device=C.device),
            "group": torch.cat(grp), "index": torch.cat(idx), "finest_flag": torch.cat(fin), "index_hash": None,
            "batch_lengths": lengths}


class LoaderWorkspace:
    """Persistent buffers of ``build_batch_gpu`` for a loader that runs batch after batch on one stream
    (``train_from_scans``): fresh allocations per batch fall through to hipMalloc whenever the caching allocator has no free
    block for that stream (blocks another stream has just used are not reusable until its events pass) -- device-wide
    stalls of ~ms that land on the training stream.  ``shared``: everything only the loader's own stream touches (staging,
    tables, scratch), one set; ``out``: what a batch hands to the trainer, one set per SLOT, handed out again only after the
    consumer's ``release_batch`` (the ``_h2d_slot`` protocol of ``prefetch_to_device``: an event on the training stream that
    the loader's stream waits for before it overwrites the slot)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.shared, self.slots, self.age = {}, [], 0
        self.pinned = None

    @staticmethod
    def _get(store, name, numel, dtype, device):
        t = store.get(name)
        if t is None or t.dtype != dtype or t.numel() < numel:
            t = store[name] = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=device)
        return t[:numel]

    def take(self, name, shape, dtype):
        n = int(np.prod(shape))
        return self._get(self.shared, name, n, dtype, self.device).view(shape)

    def host(self, numel):
        if self.pinned is None or self.pinned.numel() < numel:
            self.pinned = torch.empty(int(numel * 1.25) + 64, dtype=torch.float32, pin_memory=True)
        return self.pinned[:numel]

    def host_aux(self, numel):
        """A second pinned block (the feature jitter).  Its previous upload was enqueued before the previous batch's last host
        read, which waited for the stream: it is free by the time the next batch asks."""
        if getattr(self, "pinned_aux", None) is None or self.pinned_aux.numel() < numel:
            self.pinned_aux = torch.empty(int(numel * 1.25) + 64, dtype=torch.float32, pin_memory=True)
        return self.pinned_aux[:numel]

    def next_slot(self, stream):
        slot = next((s_ for s_ in self.slots if not s_["busy"]), None)
        if slot is None:          # every slot belongs to a batch that is still pending: grow, never overwrite
            slot = {"bufs": {}, "busy": False, "free": None, "age": -1}
            self.slots.append(slot)
        if slot["free"] is not None:
            stream.wait_event(slot["free"])
        self.age += 1
        slot["busy"], slot["age"], slot["free"] = True, self.age, None
        return slot

    def out(self, slot, name, shape, dtype):
        n = int(np.prod(shape))
        return self._get(slot["bufs"], name, n, dtype, self.device).view(shape)


_COPY_POOL = [None, False]


def _copy_pool():
    """The threads that stage raw points into pinned memory (GCL_LOADER_COPY_THREADS, default 4; 0 = the calling thread)."""
    if not _COPY_POOL[1]:
        import concurrent.futures
        n = int(os.environ.get("GCL_LOADER_COPY_THREADS", "4"))
        _COPY_POOL[0] = concurrent.futures.ThreadPoolExecutor(n, thread_name_prefix="gcl-stage") if n > 0 else None
        _COPY_POOL[1] = True
    return _COPY_POOL[0]


def _copy_rows(dst, a, b, x):
    dst[a:b] = x


def build_batch_gpu(raw_samples, voxel_size, device, K=5, jitter=None, stream=None, workspace=None):
    """``ColocationKittiDataset.__getitem__`` x batch_size + ``collate_colocation_fn`` (lib/colocation_data_loader.py:315-475)
    from raw scans, on the device, in one pass over the whole batch.

    ``raw_samples``: list of dicts ``{"xyz": [centre, nghb_0, ...] float32 [P_c, 3] host arrays (after the loader's rotation /
    scale, :348-370), "list_M": [4 x 4 neighbour -> centre], "radius": the sample's matching radius (:361-365)}``; every
    sample has the same number of clouds.  ``jitter`` (optional): ``callable(sample_no, n_center) -> float32 [n_center, 1]``
    added to the centre cloud's ones (the reference's ``Jitter`` transform draws it after voxelisation, :414-415).
    Returns the batch dict of ``collate_gpu`` (device tensors; ``index_hash`` None) + ``"cloud_rows"`` (rows per cloud).
    Work is enqueued on ``stream`` (default: the current stream); the two host reads wait for that stream only.
    ``workspace`` (a ``LoaderWorkspace``): persistent buffers instead of fresh allocations; the returned tensors then live in
    one of its slots and the dict carries ``"_h2d_slot"`` -- the consumer calls ``colocation_trainer.release_batch`` when it
    is done enqueuing work on them (the trainer does)."""
    lib = _lib.require_gpu()
    dev = torch.device(device)
    n_s = len(raw_samples)
    n_c = len(raw_samples[0]["xyz"])
    if any(len(s["xyz"]) != n_c or len(s["list_M"]) != n_c - 1 for s in raw_samples):
        raise ValueError("every sample needs the same number of clouds and one transform per neighbour")
    n_clouds = n_s * n_c
    if n_clouds > 64:
        raise ValueError("at most 64 clouds per batch")
    sizes = [len(x) for s in raw_samples for x in s["xyz"]]
    offs = np.zeros(n_clouds + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    P = int(offs[-1])
    # one pinned staging block: the points of all clouds, then the neighbour -> centre transforms (identity for a centre)
    head = n_clouds * 24                     # the transforms first: their doubles stay 8-byte aligned for any P
    W = workspace
    import time
    tr = getattr(W, "trace", None)       # diagnostic (tools/micro/e2e_probe.py): wall time of the stages of a build
    t_0 = time.perf_counter()
    host = W.host(head + P * 3) if W is not None else torch.empty(head + P * 3, dtype=torch.float32, pin_memory=True)

    def tmp(name, shape, dtype):          # only this stream touches it
        return W.take(name, shape, dtype) if W is not None else torch.empty(shape, dtype=dtype, device=dev)

    hp = host[head:].view(P, 3).numpy()
    hm = host[:head].view(torch.float64).view(n_clouds, 12).numpy()
    ci = 0
    pool = _copy_pool()
    staged = []                         # per sample: the pending copies of its clouds into the pinned block
    for s in raw_samples:
        pend = []
        for c, x in enumerate(s["xyz"]):
            # numpy releases the interpreter lock for a plain copy of this size (36 MB per batch: 2.4 ms on one thread -- the
            # largest host item of a build; GCL_LOADER_COPY_THREADS = 4 copy threads share it, and every sample's H2D copy is
            # issued as soon as ITS clouds are staged, under the staging of the next sample).  torch's copy_ was measured here
            # too: its thread pool made a batch 54 ms in a process with the default thread count
            if pool is not None:
                pend.append(pool.submit(_copy_rows, hp, int(offs[ci]), int(offs[ci + 1]), x))
            else:
                hp[offs[ci]:offs[ci + 1]] = x
            hm[ci] = (np.eye(4) if c == 0 else np.asarray(s["list_M"][c - 1], dtype=np.float64))[:3].reshape(-1)
            ci += 1
        staged.append(pend)
    st_ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
    with torch.cuda.device(dev), st_ctx:
        st = _lib.stream()
        slot = W.next_slot(torch.cuda.current_stream()) if W is not None else None

        def out(name, shape, dtype):      # handed to the consumer
            return W.out(slot, name, shape, dtype) if W is not None else torch.empty(shape, dtype=dtype, device=dev)
        d = tmp("staging", (head + P * 3,), torch.float32)
        d[:head].copy_(host[:head], non_blocking=True)
        for si, pend in enumerate(staged):
            for f in pend:
                f.result()
            a, b = head + 3 * int(offs[si * n_c]), head + 3 * int(offs[(si + 1) * n_c])
            d[a:b].copy_(host[a:b], non_blocking=True)
        xyz_raw = d[head:]
        to_center = d[:head]
        raw = tmp("raw", (P, 4), torch.int32)
        _lib.check(lib.gcl_voxel_coords_multi(_lib.ptr(xyz_raw), P, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                              n_clouds, float(voxel_size), _lib.ptr(raw), st), "gcl_voxel_coords_multi")
        cap = _cap(P)
        table = tmp("table", (cap, 2), torch.int64)
        scratch = tmp("scratch", (lib.gcl_scan_scratch_len(P),), torch.int32)
        coords = out("coords", (P, 4), torch.int32)
        index = tmp("index", (P,), torch.int64)
        meta = tmp("meta", (8 + n_clouds + 1,), torch.int32)
        off = lambda t, e: ctypes.c_void_p(t.data_ptr() + e * t.element_size())
        _lib.check(lib.gcl_unique_coords(_lib.ptr(raw), P, _lib.ptr(table), cap, _lib.ptr(scratch), _lib.ptr(coords),
                                         _lib.ptr(index), off(meta, 0), off(meta, 4), st), "gcl_unique_coords")
        _lib.check(lib.gcl_cloud_row_starts(_lib.ptr(coords), P, off(meta, 0), n_clouds, off(meta, 8), st),
                   "gcl_cloud_row_starts")
        xyz_own = out("xyz_own", (P, 3), torch.float32)
        xyz_cf = out("xyz_cf", (P, 3), torch.float32)
        _lib.check(lib.gcl_loader_points(_lib.ptr(xyz_raw), _lib.ptr(index), _lib.ptr(coords), P, off(meta, 0),
                                         _lib.ptr(to_center), _lib.ptr(xyz_own), _lib.ptr(xyz_cf), st), "gcl_loader_points")
        t_1 = time.perf_counter()
        m = meta.tolist()                                        # host read 1: rows, status, the clouds' first rows
        t_2 = time.perf_counter()
        if m[4]:
            raise ValueError(f"{m[4]} points outside the packable voxel range")
        n = m[0]
        starts = m[8:8 + n_clouds + 1]
        for c in range(n_clouds - 1, -1, -1):                    # a cloud without rows starts where the next one does
            if starts[c] < 0:
                starts[c] = starts[c + 1]
        cloud_rows = [starts[c + 1] - starts[c] for c in range(n_clouds)]
        n_center = [cloud_rows[s * n_c] for s in range(n_s)]
        nct = sum(n_center)
        if min(n_center) <= 0:
            raise ValueError("a sample without centre voxels")
        hits = tmp("hits", (nct, n_c, K), torch.int32)
        cnt = tmp("cnt", (nct, n_c), torch.int32)
        rng = tmp("rng", (nct, n_c), torch.float64)
        c0 = 0
        for si, s in enumerate(raw_samples):
            to_cloud = np.zeros((n_c, 12), dtype=np.float64)
            to_cloud[0] = np.eye(4)[:3].reshape(-1)
            for j, M in enumerate(s["list_M"]):
                to_cloud[j + 1] = np.linalg.inv(np.asarray(M, dtype=np.float64))[:3].reshape(-1)
            _lib.check(lib.gcl_colocation_hits_at(_lib.ptr(xyz_own), _lib.ptr(xyz_cf), starts[si * n_c], n_center[si],
                                                  si * n_c, n_c, to_cloud.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                  _lib.ptr(table), cap, float(1.0 / voxel_size), float(s["radius"]), K,
                                                  off(hits, c0 * n_c * K), off(cnt, c0 * n_c), off(rng, c0 * n_c), st),
                       "gcl_colocation_hits_at")
            c0 += n_center[si]
        escr = tmp("escr", (4 * nct + nct // 2048 + 64,), torch.int32)
        group = out("group", (nct,), torch.int32)
        gidx = out("index", (nct * n_c * K,), torch.int64)
        finest = out("finest", (nct * n_c * K,), torch.uint8)
        totals = tmp("totals", (2,), torch.int32)
        _lib.check(lib.gcl_colocation_emit(_lib.ptr(hits), _lib.ptr(cnt), _lib.ptr(rng), nct, n_c, K, _lib.ptr(escr),
                                           _lib.ptr(group), _lib.ptr(gidx), _lib.ptr(finest), _lib.ptr(totals), st),
                   "gcl_colocation_emit")
        F = out("F", (n, 1), torch.float32)
        F.fill_(1.0)
        if jitter is not None:
            # drawn on the host once the centre clouds' sizes are known, uploaded from pinned memory without a host wait (a
            # pageable copy waits for everything queued on this low-priority stream: ~ a training step per sample)
            draws = [jitter(si, n_center[si]) for si in range(n_s)]
            if any(j is not None for j in draws):
                jh = W.host_aux(nct) if W is not None else torch.empty(nct, dtype=torch.float32, pin_memory=True)
                jn, c0 = jh.numpy(), 0
                for si, j in enumerate(draws):
                    jn[c0:c0 + n_center[si]] = 0.0 if j is None else np.asarray(j, dtype=np.float32).reshape(-1)
                    c0 += n_center[si]
                jd = tmp("jitter", (nct,), torch.float32)
                jd.copy_(jh, non_blocking=True)
                c0 = 0
                for si in range(n_s):
                    r0 = starts[si * n_c]
                    F[r0:r0 + n_center[si], 0] += jd[c0:c0 + n_center[si]]
                    c0 += n_center[si]
        t_3 = time.perf_counter()
        g, e = totals.tolist()                                   # host read 2: groups, members
        if tr is not None:
            t_4 = time.perf_counter()
            for k_, v_ in (("stage + enqueue 1", t_1 - t_0), ("host read 1", t_2 - t_1), ("enqueue 2 (+ jitter draws)", t_3 - t_2),
                           ("host read 2", t_4 - t_3), ("builds", 1.0)):
                tr[k_] = tr.get(k_, 0.0) + v_
        lengths = [starts[(si + 1) * n_c] - starts[si * n_c] for si in range(n_s)]
        res = {"sinput_C": coords[:n], "sinput_F": F, "group": group[:g], "index": gidx[:e], "finest_flag": finest[:e].bool(),
               "index_hash": None, "batch_lengths": lengths, "cloud_rows": cloud_rows, "xyz_own": xyz_own[:n],
               "xyz_cf": xyz_cf[:n]}
        if slot is not None:
            res["_h2d_slot"] = (slot, slot["age"])
        return res


def train_from_scans(trainer, raw_batches, voxel_size=None, K=5, jitter=None, depth=None, workers=None):
    """The reference's epoch loop from the loader's side (lib/colocation_trainer.py:838-846 pulls a batch from the DataLoader
    inside every iteration; the workers' ``__getitem__`` + collate are lib/colocation_data_loader.py:315-475): yields
    ``trainer.train_steps``' results for an iterable of RAW batches (lists of ``build_batch_gpu`` samples).  ``workers``
    helper threads (default ``GCL_LOADER_WORKERS`` = 2; the reference runs 4 loader processes, config.py) run
    ``build_batch_gpu`` -- H2D of the raw points included -- up to ``depth`` batches ahead, each on its own low-priority stream
    with its own ``LoaderWorkspace``; the trainer's own helpers (draws, coordinate maps) then work on a finished batch as on
    any prefetched one.  One worker is not enough beside a saturated GPU: a build waits twice for its stream (two host
    reads), and behind the training streams that latency is about a training step -- measured 12 - 16 ms per build
    (tools/micro/e2e_probe.py), i.e. the loader, not the step, set the pace.  Batches are yielded in the iterable's order.
    ``jitter``: ``callable(raw_samples) -> build_batch_gpu's jitter callable`` (e.g. ``synthetic.raw_sample_jitter``)."""
    import concurrent.futures
    import os
    import threading
    from collections import deque
    dev = trainer.device
    vs = float(voxel_size if voxel_size is not None else trainer.config.voxel_size)
    n_workers = max(1, int(workers if workers is not None else os.environ.get("GCL_LOADER_WORKERS", "2")))
    depth = max(n_workers, int(depth if depth is not None else os.environ.get("GCL_LOADER_DEPTH", str(n_workers + 1))))
    local = threading.local()

    def build(raw):
        with torch.cuda.device(dev):
            if getattr(local, "stream", None) is None:
                lo, hi = torch.cuda.Stream.priority_range()
                prio = {"low": lo, "high": hi}.get(os.environ.get("GCL_LOADER_PRIORITY", "low"), 0)
                local.stream = torch.cuda.Stream(device=dev, priority=prio)
                local.work = LoaderWorkspace(dev)
            b = build_batch_gpu(raw, vs, dev, K=K, jitter=jitter(raw) if jitter is not None else None, stream=local.stream,
                                workspace=local.work)
            ev = torch.cuda.Event()
            ev.record(local.stream)
        b["_h2d_event"] = ev        # what wait_for_batch / the map helper order their streams behind
        return b

    def batches():
        it = iter(raw_batches)
        with concurrent.futures.ThreadPoolExecutor(max_workers=n_workers) as pool:
            pending = deque()
            for raw in it:
                pending.append(pool.submit(build, raw))
                if len(pending) > depth:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()

    yield from trainer.train_steps(batches())
