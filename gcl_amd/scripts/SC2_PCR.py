"""SC2-PCR registration back-end on the MI355X kernels (interface of scripts/SC2_PCR/SC2_PCR.py: class ``Matcher``).

``Matcher(**config_KITTI.json).estimator(src_keypts, tgt_keypts, src_features, tgt_features)`` as called by the eval
loop (scripts/test_kitti.py:172-180), batch size 1 (the reference asserts it too, :42, :249).  Putative
correspondences come from ``gcl_nn_rowmin`` (the reference's argmin of sqrt(2 - 2 f.g) over L2-normalised features
is the argmin of |f - g|^2); the registration itself is ONE C-ABI call (include/gcl_amd.h, gcl_sc2_register; with
``GCL_SC2_ONE_CALL=0`` the five staged calls with three small torch steps in between -- a stable sort for the seeds, an
argmax, the inlier labels -- that it replaced: same kernels, same results).  Nothing leaves the device until the caller
reads the result.
"""
import numpy as np
import torch

from gcl_amd import _lib
from gcl_amd.lib.eval import host_to_device
from gcl_amd.lib.metrics import pdist_min

import os

SPARSE_CONFIDENCE = os.environ.get("GCL_SC2_SPARSE", "1") != "0"      # gcl_sc2_confidence_sparse (round 5)
ONE_CALL = os.environ.get("GCL_SC2_ONE_CALL", "1") != "0"              # gcl_sc2_register: the stages below as one call


class _Stages:
    """``Matcher.last`` of a one-call registration: views into its output buffer, made when asked for."""

    def __init__(self, buf, fields):
        self._buf, self._fields = buf, fields

    def __getitem__(self, name):
        off, count, dtype, shape = self._fields[name]
        v = self._buf[off:off + count * dtype.itemsize].view(dtype)
        v = v.view(shape) if shape is not None else v
        return v[0] if name == "best" else v

    def keys(self):
        return self._fields.keys()


class Matcher:
    def __init__(self, inlier_threshold=0.10, num_node="all", use_mutual=True, d_thre=0.1, num_iterations=10,
                 ratio=0.2, nms_radius=0.1, max_points=8000, k1=30, k2=20, select_scene=None):
        self.inlier_threshold, self.num_node, self.use_mutual = inlier_threshold, num_node, use_mutual
        self.d_thre, self.num_iterations, self.ratio = d_thre, num_iterations, ratio
        self.max_points, self.nms_radius, self.k1, self.k2 = max_points, nms_radius, k1, k2

    # ---- scripts/SC2_PCR/SC2_PCR.py:281-302 --------------------------------------------------------------------
    def match_pair(self, src_keypts, tgt_keypts, src_features, tgt_features):
        N_src, N_tgt = src_features.shape[1], tgt_features.shape[1]
        dev = src_features.device
        if self.num_node == "all":
            src_sel, tgt_sel = None, None
        else:                                                    # with replacement, as the reference (:289-290)
            src_sel = host_to_device(np.random.choice(N_src, self.num_node), dev)      # pinned block, non-blocking copy
            tgt_sel = host_to_device(np.random.choice(N_tgt, self.num_node), dev)
        _, arg = pdist_min(src_features[0], tgt_features[0], "SquareL2", rows_a=src_sel, rows_b=tgt_sel)
        arg = arg.long()
        src_rows = src_sel if src_sel is not None else torch.arange(N_src, device=dev)
        tgt_rows = tgt_sel[arg] if tgt_sel is not None else arg
        return src_keypts[:, src_rows], tgt_keypts[:, tgt_rows]

    # ---- :304-381 ------------------------------------------------------------------------------------------------
    def SC2_PCR(self, src_keypts, tgt_keypts):
        lib = _lib.require_gpu()
        if src_keypts.shape[0] != 1:
            raise NotImplementedError("batch size 1 only (as the reference's pick_seeds / post_refinement)")
        src = src_keypts[0, :self.max_points].to(torch.float32).contiguous()
        tgt = tgt_keypts[0, :self.max_points].to(torch.float32).contiguous()
        n = src.shape[0]
        dev = src.device
        st = _lib.stream()
        n_seeds = int(n * self.ratio)
        if n_seeds < 1:
            raise ValueError("too few correspondences for SC2-PCR")
        k1, k2 = (self.k1, self.k2) if self.k1 <= n else (4, 4)                      # :75-77
        thr = 0.10 if self.inlier_threshold == 0.10 else 1.2
        if ONE_CALL and SPARSE_CONFIDENCE:
            # every stage below in ONE native call (same kernels, same results): two allocations, no torch operation between
            # the stages -- the loop over pairs was bound by this thread, not by the device
            fields, off = {}, 0
            for name, count, dtype, shape in (("out", 16, torch.float32, (1, 4, 4)), ("labels", n, torch.float32, (1, n)),
                                              ("conf", n, torch.float32, None), ("seeds", n_seeds, torch.int64, None),
                                              ("knn", n_seeds * k1, torch.int32, (n_seeds, k1)),
                                              ("seed_trans", n_seeds * 12, torch.float32, (n_seeds, 12)),
                                              ("fitness", n_seeds, torch.float32, None), ("best", 1, torch.int32, None)):
                fields[name] = (off, count, dtype, shape)
                off += (count * dtype.itemsize + 255) // 256 * 256
            buf = torch.empty(off, dtype=torch.uint8, device=dev)
            scratch = torch.empty(lib.gcl_sc2_register_scratch_bytes(n), dtype=torch.uint8, device=dev)
            base = buf.data_ptr()
            at = lambda name: base + fields[name][0]
            _lib.check(lib.gcl_sc2_register(_lib.ptr(src), _lib.ptr(tgt), n, float(self.d_thre), int(self.num_iterations),
                                            float(self.nms_radius), n_seeds, k1, k2, float(self.inlier_threshold), thr, 20,
                                            _lib.ptr(scratch), at("conf"), at("seeds"), at("knn"), at("seed_trans"),
                                            at("fitness"), at("best"), at("out"), at("labels"), st), "gcl_sc2_register")
            self.last = _Stages(buf, fields)
            self._labels = self.last["labels"]
            return self.last["out"]
        self._labels = None
        # confidence of every correspondence (:337-345)
        conf = torch.ones(n, dtype=torch.float32, device=dev)
        partial = torch.empty(lib.gcl_sc2_chunks() * n, dtype=torch.float32, device=dev)
        done = torch.zeros(1, dtype=torch.int32, device=dev)
        if SPARSE_CONFIDENCE:      # the matrix's non-zero entries kept from one build: bitwise the dense products' result
            scratch = torch.empty(lib.gcl_sc2_confidence_scratch_bytes(n), dtype=torch.uint8, device=dev)
            _lib.check(lib.gcl_sc2_confidence_sparse(_lib.ptr(src), _lib.ptr(tgt), n, float(self.d_thre),
                                                     int(self.num_iterations), _lib.ptr(partial), _lib.ptr(conf),
                                                     _lib.ptr(done), _lib.ptr(scratch), st), "gcl_sc2_confidence_sparse")
        else:
            _lib.check(lib.gcl_sc2_confidence(_lib.ptr(src), _lib.ptr(tgt), n, float(self.d_thre),
                                              int(self.num_iterations), _lib.ptr(partial), _lib.ptr(conf), _lib.ptr(done),
                                              st), "gcl_sc2_confidence")
        # seeds: local maxima first, by confidence (:32-58); ties -> lowest index
        is_max = torch.ones(n, dtype=torch.int32, device=dev)
        _lib.check(lib.gcl_sc2_local_max(_lib.ptr(src), _lib.ptr(conf), n, float(self.nms_radius), _lib.ptr(is_max),
                                         st), "gcl_sc2_local_max")
        seeds = torch.sort(-(conf * is_max.float()), stable=True)[1][:n_seeds].contiguous()
        # k1 most compatible correspondences of every seed under the second-order measure (:353-361, :85-86)
        bits = torch.empty(n * ((n + 63) // 64), dtype=torch.int64, device=dev)
        knn = torch.empty((n_seeds, k1), dtype=torch.int32, device=dev)
        _lib.check(lib.gcl_sc2_seed_knn(_lib.ptr(src), _lib.ptr(tgt), n, _lib.ptr(seeds), n_seeds, float(self.d_thre),
                                        k1, _lib.ptr(bits), _lib.ptr(knn), st), "gcl_sc2_seed_knn")
        # one hypothesis per seed and its inlier count (:88-161)
        trans = torch.empty((n_seeds, 12), dtype=torch.float32, device=dev)
        fitness = torch.empty(n_seeds, dtype=torch.float32, device=dev)
        _lib.check(lib.gcl_sc2_seed_trans(_lib.ptr(src), _lib.ptr(tgt), n, _lib.ptr(knn), n_seeds, k1, k2,
                                          float(self.d_thre), int(self.num_iterations), float(self.inlier_threshold),
                                          _lib.ptr(trans), _lib.ptr(fitness), st), "gcl_sc2_seed_trans")
        best = torch.sort(-fitness, stable=True)[1][0]
        T = trans[best].clone()
        # post refinement over all correspondences (:238-279)
        rpart = torch.empty(lib.gcl_sc2_refine_partial_len(), dtype=torch.float64, device=dev)
        state = torch.empty(2, dtype=torch.int32, device=dev)
        _lib.check(lib.gcl_sc2_refine(_lib.ptr(src), _lib.ptr(tgt), n, thr, 20, _lib.ptr(rpart), _lib.ptr(state),
                                      _lib.ptr(T), st), "gcl_sc2_refine")
        out = torch.zeros((1, 4, 4), dtype=torch.float32, device=dev)
        out[0, :3, :] = T.view(3, 4)
        out[0, 3, 3] = 1.0
        self.last = dict(conf=conf, seeds=seeds, knn=knn, seed_trans=trans, fitness=fitness, best=best)
        return out

    # ---- :383-410 ------------------------------------------------------------------------------------------------
    def estimator(self, src_keypts, tgt_keypts, src_features, tgt_features):
        src_corr, tgt_corr = self.match_pair(src_keypts, tgt_keypts, src_features, tgt_features)
        pred_trans = self.SC2_PCR(src_corr, tgt_corr)
        if self._labels is not None and src_corr.shape[1] <= self.max_points:      # made by the registration call itself
            return pred_trans, self._labels, src_corr, tgt_corr
        warped = src_corr @ pred_trans[:, :3, :3].transpose(1, 2) + pred_trans[:, None, :3, 3]
        distance = torch.sum((warped - tgt_corr) ** 2, dim=-1) ** 0.5
        pred_labels = (distance < self.inlier_threshold).float()
        return pred_trans, pred_labels, src_corr, tgt_corr
