"""Evaluation loop body of the reference's registration benchmark (interface of scripts/test_kitti.py:29-227).

``eval_pairs(model, pairs, matcher)`` composes, per pair and in the reference's order:
  2 x feature extraction (``:141-152``)  ->  ``find_corr`` on <= 5000 random rows per cloud + nearest-neighbour
  distances under the ground truth (``:154-155``)  ->  ``random_sample`` to exactly 5000 voxels (``:160-161``)  ->
  registration with the SC2-PCR ``Matcher.estimator`` (``:178-180``, the ``use_RANSAC false`` branch; open3d's RANSAC
  is a third-party CPU back-end and is not built)  ->  RTE / RRE / success meters (``:189-217``).

What differs by design: the clouds of ``batch_pairs`` pairs go through ONE forward pass (``forward_clouds``): in eval
mode the network treats the clouds of a batch independently (running BatchNorm statistics, per-cloud coordinate maps)
and every output row is accumulated in a fixed offset order, so the features equal the separate passes bit for bit at
a fraction of the launches of this launch-bound case.  All host random draws are made per pair in the reference's
order, so a seeded run selects the same rows whatever ``batch_pairs`` is.
"""
import os
import time

import numpy as np
import torch

import gcl_amd.MinkowskiEngine as ME
from gcl_amd.lib.eval import DeferredCorr, host_to_device


class AverageMeter:
    """lib/timer.py:6-26."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.sq_sum, self.count, self.var = 0, 0, 0.0, 0.0, 0, 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.sq_sum += val ** 2 * n
        self.var = self.sq_sum / self.count - self.avg ** 2


def apply_transform(pts, trans):
    """scripts/test_kitti.py:45-48."""
    return pts @ trans[:3, :3].t() + trans[:3, 3]


def evaluate_nn_dist(xyz0, xyz1, T_gth):
    """scripts/test_kitti.py:50-53."""
    xyz0 = apply_transform(xyz0, T_gth)
    return torch.sqrt(((torch.as_tensor(xyz0) - torch.as_tensor(xyz1)) ** 2).sum(1) + 1e-6).tolist()


def random_sample(pcd, feats, N):
    """Exactly N rows (scripts/test_kitti.py:55-75): a permutation prefix when there are more, draws WITH replacement
    when there are fewer.  ``feats`` may live on the GPU; the index is drawn on the host like the reference's."""
    n1 = pcd.shape[0]
    if n1 == N:
        return pcd, feats
    choice = np.random.permutation(n1)[:N] if n1 > N else np.random.choice(n1, N)
    sel = host_to_device(choice, feats.device) if isinstance(feats, torch.Tensor) else choice
    return pcd[choice], feats[sel]


def forward_clouds(model, clouds):
    """Features of several clouds from ONE forward pass.  ``clouds``: list of (F [N_i, C], coords int32 [N_i, 4] with any
    batch column); returns the list of per-cloud feature tensors.  Eval mode only (batch statistics would mix clouds)."""
    if model.training:
        raise RuntimeError("forward_clouds needs model.eval(): batch statistics would mix the clouds")
    if len(clouds) == 1:
        return [model(ME.SparseTensor(clouds[0][0], coordinates=clouds[0][1])).F]
    C = torch.cat([c for _, c in clouds])        # (a copy: the callers' batch columns stay as they are)
    off = 0
    for b, (_, c) in enumerate(clouds):
        C[off:off + len(c), 0] = b
        off += len(c)
    out = model(ME.SparseTensor(torch.cat([f for f, _ in clouds]), coordinates=C)).F
    return list(torch.split(out, [len(f) for f, _ in clouds]))


def forward_clouds_stream(model, cloud_sets, device=None, depth=3, exec_streams=None):
    """Features of successive cloud sets (an iterable of lists of (F, coords) as ``forward_clouds`` takes them), one forward
    pass per set, with the COORDINATE MAPS of the next sets built ahead: a helper thread copies set i + 1 to the device and
    makes its ``gcl_maps_build`` call on a side stream (the call carries the pass's host syncs -- level sizes -- and runs
    with the interpreter lock released) while the main thread enqueues the forward pass of set i.  A pass over one pair of
    35 k voxels is bound by exactly that chain (scripts/test_kitti.py:141-152 runs pair after pair); the features are those
    of ``forward_clouds`` bit for bit (same maps, same launches).  Yields one list of per-cloud feature tensors per set.
    ``exec_streams`` (default ``GCL_FWD_STREAMS`` = 3): the passes themselves alternate over that many streams -- a pass over
    one pair is a chain of ~25 dependent launches that leaves most of the chip idle, and the passes of different sets are
    independent; the caller's stream is made to wait for a set's pass before the set is yielded.  One pair per pass, M
    voxels/s (tools/micro/fwd_stream_probe.py): one call per pair 21.3; 1 / 2 / 3 / 4 streams at depth 3: 24.4 / 33.9 / 40.7 /
    26.9 (depth 2: 24.4 / 33.4 / 38.3 / 27.8)."""
    import concurrent.futures
    if model.training:
        raise RuntimeError("forward_clouds_stream needs model.eval(): batch statistics would mix the clouds")
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    specs = model.native_map_specs(training=False) if hasattr(model, "native_map_specs") else None
    sets = iter(cloud_sets)
    if specs is None or dev.type != "cuda":
        for clouds in sets:
            yield forward_clouds(model, [(f.to(dev), c.to(dev)) for f, c in clouds])
        return
    side = torch.cuda.Stream(device=dev)
    n_exec = int(os.environ.get("GCL_FWD_STREAMS", "3")) if exec_streams is None else int(exec_streams)
    execs = [torch.cuda.Stream(device=dev) for _ in range(n_exec)] if n_exec > 1 else [None]
    ring = [{"arena": None, "free": None} for _ in range(depth + max(1, n_exec) + 1)]
    # everything the caller has enqueued so far (parameter updates, the inputs themselves when they are device tensors) comes
    # first on the helper's stream and on the execution streams too
    with torch.cuda.device(dev):
        started = torch.cuda.Event()
        started.record(torch.cuda.current_stream())
        side.wait_event(started)
        for e in execs:
            if e is not None:
                e.wait_event(started)

    def build(clouds, slot, inputs_ready):
        with torch.cuda.device(dev), torch.cuda.stream(side):
            # device tensors of this set may have been produced on the caller's stream after the generator started (a lazy
            # ``((f.to(dev), c.to(dev)) for ...)``, or the output of earlier kernels): the side stream reads them only behind
            # everything the caller had enqueued when the set was pulled
            side.wait_event(inputs_ready)
            if slot["free"] is not None:
                side.wait_event(slot["free"])           # the pass that read this arena last has been enqueued and has run
            Fs, Cs = [], []
            for b, (f, c) in enumerate(clouds):
                cb = c.to(dev, non_blocking=True)
                if len(clouds) > 1:
                    cb = cb.clone() if cb is c else cb
                    cb[:, 0] = b
                Fs.append(f.to(dev, non_blocking=True))
                Cs.append(cb)
            F = Fs[0] if len(Fs) == 1 else torch.cat(Fs)
            C = Cs[0] if len(Cs) == 1 else torch.cat(Cs)
            mgr = ME.CoordinateManager.build_native(C, specs, arena=slot["arena"])
            slot["arena"] = mgr.native.arena
            ev = torch.cuda.Event()
            ev.record(side)
        return F, mgr, ev, [len(f) for f in Fs], slot

    ran = [0]

    def run(built):
        feats = _run_built(model, built, execs[ran[0] % len(execs)])
        if ran[0] == 0 and execs[0] is not None:
            # what a model's FIRST inference pass leaves behind for the later ones (packed kernels in the plan's state buffer,
            # the BatchNorm modules' folded scale / shift) is written on the first stream: the others start after it
            for e in execs[1:]:
                e.wait_event(built[4]["free"])
        ran[0] += 1
        return feats

    with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool, torch.cuda.device(dev):
        pending, k = [], 0
        for clouds in sets:
            pulled = torch.cuda.Event()
            pulled.record(torch.cuda.current_stream())
            pending.append(pool.submit(build, clouds, ring[k % len(ring)], pulled))
            k += 1
            if len(pending) > depth:
                yield run(pending.pop(0).result())
        while pending:
            yield run(pending.pop(0).result())


def _run_built(model, built, exec_stream=None):
    F, mgr, ev, sizes, slot = built
    caller = torch.cuda.current_stream()
    st = exec_stream if exec_stream is not None else caller
    with torch.cuda.stream(st):
        st.wait_event(ev)
        out = model(ME.SparseTensor(F, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
        slot["free"] = torch.cuda.Event()
        slot["free"].record(st)
        # allocated on the side stream, read by the pass just enqueued on this one: the allocator must not hand their blocks
        # to the helper's next allocations before that pass has run (the maps' arena is the ring's own, guarded by slot["free"])
        F.record_stream(st)
        mgr.native.coords.record_stream(st)
    if st is not caller:
        caller.wait_event(slot["free"])        # the caller's later work (and only that) is ordered behind this pass
        out.record_stream(caller)
    return list(torch.split(out, sizes)) if len(sizes) > 1 else [out]


def rotation_translation_error(T_est, T_gth):
    """(rte, rre in radians) with the reference's clamp of the trace diagonal (scripts/test_kitti.py:189-192)."""
    T_est, T_gth = torch.as_tensor(T_est).float().cpu(), torch.as_tensor(T_gth).float().cpu()
    rte = float(np.linalg.norm((T_est[:3, 3] - T_gth[:3, 3]).numpy()))
    m = T_est[:3, :3].t() @ T_gth[:3, :3]
    d = torch.min(torch.ones(3), torch.diagonal(m))
    with np.errstate(invalid="ignore"):
        rre = float(np.arccos((float(d.sum()) - 1) / 2))
    return rte, rre


def eval_pairs(model, pairs, matcher, device=None, batch_pairs=1, subsample_size=5000, n_points=5000,
               rte_thresh=2.0, rre_thresh=5.0, collect=False):
    """The loop of scripts/test_kitti.py:129-227 over ``pairs`` (dicts with the keys of ``collate_debug_pair_fn``:
    pcd0 / pcd1, sinput{0,1}_C, sinput{0,1}_F, T_gt).  Returns a dict with the three meters' summary, the per-pair
    lists (T_est, rte, rre, success, nn distances when ``collect``) and ``feat_enqueue_time`` / ``reg_enqueue_time``: the HOST
    time spent enqueuing the two stages, in seconds -- the loop is asynchronous, so these are not the reference's
    feat_timer / reg_timer (scripts/test_kitti.py:141-180), which include the device time behind a per-pair ``.to('cpu')``."""
    if matcher is None:
        raise NotImplementedError("only the SC2-PCR branch (use_RANSAC false) is built: open3d RANSAC is a CPU third party")
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    model.eval()
    success_meter, rte_meter, rre_meter = AverageMeter(), AverageMeter(), AverageMeter()
    out = dict(T_est=[], rte=[], rre=[], success=[], dists_nn=[], n_voxels=0)
    t_feat = t_reg = 0.0
    pairs = list(pairs)

    def finish(pending):
        """Host half of a chunk whose device work was enqueued one chunk ago: ONE wait for its transformations (a pinned
        copy behind an event), then the reference's per-pair bookkeeping (scripts/test_kitti.py:180-217)."""
        chunk, corrs, T_host, ev = pending
        ev.synchronize()
        for j, d in enumerate(chunk):
            T_est, T_gth = T_host[j].clone(), d["T_gt"]
            if collect:
                xyz0_corr, xyz1_corr = corrs[j].resolve()
                out["dists_nn"].append(evaluate_nn_dist(xyz0_corr, xyz1_corr, T_gth))
            rte, rre = rotation_translation_error(T_est, T_gth)
            if rte < rte_thresh:
                rte_meter.update(rte)
            if not np.isnan(rre) and rre < np.pi / 180 * rre_thresh:
                rre_meter.update(rre * 180 / np.pi)
            ok = rte < rte_thresh and not np.isnan(rre) and rre < np.pi / 180 * rre_thresh
            success_meter.update(1 if ok else 0)
            out["T_est"].append(T_est)
            out["rte"].append(rte)
            out["rre"].append(rre)
            out["success"].append(bool(ok))

    chunks = [pairs[b0:b0 + max(1, batch_pairs)] for b0 in range(0, len(pairs), max(1, batch_pairs))]
    n_streams = max(1, int(os.environ.get("GCL_EVAL_STREAMS", "1")))
    with torch.cuda.device(dev), torch.no_grad():
        pending = None
        main = torch.cuda.current_stream()
        # GCL_EVAL_STREAMS > 1 (measured, not the default): pair j's registration on stream j mod n_streams beside the others
        # and beside the next chunk's forward pass, a collector stream gathering the chunk's transformations -- 276 pairs/s on
        # one stream vs 259 - 261 on 2 - 4 at batch_pairs = 8 (216 vs 210 - 222 at 1): the registrations are bound by the
        # enqueuing thread and by kernels that fill the chip (the matrix build, the 1-NN), not by idle gaps between launches.
        # The host draws stay in the reference's per-pair order (they are made while enqueuing, serially).
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [main]
        collector = torch.cuda.Stream(device=dev) if n_streams > 1 else main
        # the maps of the coming chunks are built on a side stream while this chunk's kernels are enqueued
        feat_stream = forward_clouds_stream(model, ([(d[f"sinput{k}_F"], d[f"sinput{k}_C"]) for d in chunk for k in (0, 1)]
                                                    for chunk in chunks), device=dev)
        pair_no = 0
        for chunk in chunks:
            t0 = time.perf_counter()
            feats = next(feat_stream)
            feats_ready = torch.cuda.Event()
            feats_ready.record(main)
            t_feat += time.perf_counter() - t0
            corrs, Ts, dones = [], [], []
            for j, d in enumerate(chunk):
                st = streams[pair_no % len(streams)]
                pair_no += 1
                with torch.cuda.stream(st):
                    st.wait_event(feats_ready)
                    F0, F1 = feats[2 * j].detach(), feats[2 * j + 1].detach()
                    if st is not main:               # allocated on the main stream, read on this one
                        F0.record_stream(st)
                        F1.record_stream(st)
                    out["n_voxels"] += len(F0) + len(F1)
                    xyz0, xyz1 = d["pcd0"][0], d["pcd1"][0]
                    xyz0np, xyz1np = xyz0.numpy(), xyz1.numpy()
                    # draws + feature 1-NN now (the reference's order); its 5000 indices are read back only when ``collect``
                    corrs.append(DeferredCorr(xyz0, xyz1, F0, F1, subsample_size=subsample_size))
                    xyz0s, F0s = random_sample(xyz0np, F0, n_points)
                    xyz1s, F1s = random_sample(xyz1np, F1, n_points)
                    t0 = time.perf_counter()
                    x0, x1 = host_to_device(xyz0s, dev), host_to_device(xyz1s, dev)
                    T_est, _, _, _ = matcher.estimator(x0[None], x1[None], F0s[None], F1s[None])
                    Ts.append(T_est[0])
                    done = torch.cuda.Event()
                    done.record(st)
                    dones.append(done)
                    t_reg += time.perf_counter() - t0
            # the reference reads T on the host after every pair (:180); here the chunk's transformations leave the device
            # in ONE copy, and the host half of a chunk runs while the next chunk's kernels execute
            T_host = torch.empty((len(chunk), 4, 4), dtype=torch.float32, pin_memory=True)
            with torch.cuda.stream(collector):
                for done, T in zip(dones, Ts):
                    collector.wait_event(done)
                    if collector is not main:
                        T.record_stream(collector)
                T_host.copy_(torch.stack(Ts), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(collector)
            if pending is not None:
                finish(pending)
            pending = (chunk, corrs, T_host, ev)
        if pending is not None:
            finish(pending)
        if n_streams > 1:      # leave nothing of this loop in flight on streams the caller does not know
            for st in streams:
                main.wait_stream(st)
            main.wait_stream(collector)
    out.update(rte_avg=rte_meter.avg, rte_var=rte_meter.var, rre_avg=rre_meter.avg, rre_var=rre_meter.var,
               success_rate=success_meter.avg, n_pairs=success_meter.count, feat_enqueue_time=t_feat, reg_enqueue_time=t_reg)
    return out
