"""ctypes binding of libgcl_hip.so (the C ABI declared in include/gcl_amd.h).

The product path has NO CPU fallback: if the library is missing or no GPU is visible, every operator raises.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GCL_LIB_PATH", os.path.join(CSRC, "libgcl_hip.so"))   # override: diagnostic builds only
SOURCES = ["coords.hip", "conv.hip", "norm.hip", "loss.hip", "data.hip", "sc2pcr.hip", "plan.hip"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "gcl_amd.h")

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

MAX_LEVELS, MAX_MAPS = 8, 16          # GCL_MAX_LEVELS, GCL_MAX_MAPS
MAPS_PINNED_BYTES = 512 * (MAX_MAPS + 1)
ERR_ARENA = -4


class MapSpec(ctypes.Structure):      # gcl_map_spec
    _fields_ = [("t_in", _i32), ("kernel_size", _i32), ("stride", _i32), ("tables", _i32), ("pairs", _i32)]


class MapDesc(ctypes.Structure):      # gcl_map_desc
    _fields_ = [("t_in", _i32), ("kernel_size", _i32), ("stride", _i32), ("K", _i32), ("level_in", _i32),
                ("level_out", _i32), ("n_in", _i64), ("n_out", _i64), ("n_pairs", _i64),
                ("nbr", _vp), ("nbr_t", _vp), ("counts", _vp),
                ("tbl_n", _vp), ("order_n", _vp), ("mask_n", _vp), ("tbl_t", _vp), ("order_t", _vp), ("mask_t", _vp),
                ("pair_in", _vp), ("pair_out", _vp), ("presence", _vp), ("dw_bounds", _vp), ("seg_off", _i64 * 128),
                ("counts_host", _i32 * 128)]


class MapsDesc(ctypes.Structure):     # gcl_maps_desc
    _fields_ = [("n_levels", _i32), ("n_maps", _i32), ("n_rows", _i64 * MAX_LEVELS), ("coords", _vp * MAX_LEVELS),
                ("table", _vp * MAX_LEVELS), ("cap", _i64 * MAX_LEVELS), ("status", _i32 * 4), ("arena_used", _i64),
                ("ready_event", _vp), ("late_mask", _i32), ("reserved", _i32), ("maps", MapDesc * MAX_MAPS)]


class PlanOp(ctypes.Structure):       # gcl_plan_op
    _fields_ = [("kind", _i32), ("x", _i32), ("x2", _i32), ("y", _i32), ("level_in", _i32), ("level_out", _i32),
                ("cin", _i32), ("cout", _i32), ("map", _i32), ("transpose", _i32), ("K", _i32), ("w", _i32),
                ("bias", _i32), ("bn_w", _i32), ("bn_b", _i32), ("bn", _i32), ("relu", _i32), ("momentum", _f32),
                ("eps", _f32)]


OP_CONVBN, OP_CONV, OP_RELU, OP_CAT, OP_ROWNORM = 1, 2, 3, 4, 5

# name -> (restype, argtypes); mirrors include/gcl_amd.h one to one (tests/test_abi.py checks both directions)
SIGNATURES = {
    "gcl_last_error": (ctypes.c_char_p, []),
    "gcl_version": (_i32, []),
    "gcl_device_count": (_i32, []),
    "gcl_coords_insert": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "gcl_scan_scratch_len": (_i64, [_i64]),
    "gcl_stride_map": (_i32, [_vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gcl_exclusive_scan_i32": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "gcl_voxel_coords": (_i32, [_vp, _i64, _f32, _i32, _vp, _vp]),
    "gcl_unique_coords": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_colocation_hits": (_i32, [_vp, _vp, _i64, _i32, ctypes.POINTER(ctypes.c_double), _vp, _i64, _f32,
                                    ctypes.c_double, _i32, _vp, _vp, _vp, _vp]),
    "gcl_colocation_emit": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_voxel_coords_multi": (_i32, [_vp, _i64, ctypes.POINTER(_i64), _i32, _f32, _vp, _vp]),
    "gcl_cloud_row_starts": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "gcl_loader_points": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gcl_colocation_hits_at": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, ctypes.POINTER(ctypes.c_double), _vp, _i64, _f32,
                                       ctypes.c_double, _i32, _vp, _vp, _vp, _vp]),
    "gcl_host_legacy_choice": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gcl_kernel_map_bitmap_len": (_i64, []),
    "gcl_kernel_map_scratch_len": (_i64, [_i32, _i64]),
    "gcl_kernel_map": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gcl_kernel_map_pairs": (_i32, [_vp, _i32, _i64, ctypes.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "gcl_pack_weights_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "gcl_amax": (_i32, [_vp, _i64, _vp, _i32, _vp]),
    "gcl_amax_multi": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "gcl_pack_weights_multi": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "gcl_conv_fwd_nb": (_i32, [_i64, _i32, _i32]),
    "gcl_pack_weights": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "gcl_table_sort_scratch_len": (_i64, [_i64]),
    "gcl_table_sort": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_table_sort_multi": (_i32, [_vp, _i32, _vp]),
    "gcl_spatial_order": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "gcl_table_sort_pre": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_conv_fwd_groups_scratch_len": (_i64, [_i64, _i32, _i32, _i32]),
    "gcl_conv_fwd_fused": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp,
                                  _i32, _vp, _vp, _vp, _i32, _vp]),
    "gcl_conv_fwd_fused_ld": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp,
                                     _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "gcl_split_planes": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "gcl_conv_fwd": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32,
                            _vp]),
    "gcl_conv_bwd_weight_scratch_len": (_i64, [_i32, _i32, _i32, _i64, _i64]),
    "gcl_conv_bwd_weight": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, ctypes.POINTER(_i64), _i32, _i32, _i32, _i32, _vp,
                                   _vp, _vp, _vp, _vp]),
    "gcl_conv_bwd_weight_bounds_len": (_i64, [_i32, _i64]),
    "gcl_conv_bwd_weight_bounds": (_i32, [_vp, ctypes.POINTER(_i64), _i32, _i64, _vp, _vp]),
    "gcl_conv_bwd_weight_rg": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, ctypes.POINTER(_i64), _i32, _i32, _i32, _i32, _vp,
                                      _vp, _vp, _vp, _vp, _vp]),
    "gcl_conv_bwd_weight_rows_scratch_len": (_i64, [_i32, _i32, _i32, _i64]),
    "gcl_conv_bwd_weight_rows": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_presence_bits": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "gcl_kernel_map_3_from_5": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "gcl_not_ones_rows": (_i32, [_vp, _i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    "gcl_stem_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "gcl_stem_bwd_weight_scratch_len": (_i64, [_i32, _i32, _i32, _i64]),
    "gcl_stem_bwd_weight": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_bn_scratch_len": (_i64, [_i64, _i32]),
    "gcl_bn_stats": (_i32, [_vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_bn_tiles_scratch_len": (_i64, [_i64, _i32]),
    "gcl_bn_stats_from_tiles_range": (_i32, [_vp, _i64, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                             _vp]),
    "gcl_bn_apply_planes": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_bn_bwd_reduce_range": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_bn_bwd_apply_planes": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_bn_stats_from_tiles": (_i32, [_vp, _i64, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcl_bn_apply": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_bn_apply_ld": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "gcl_bn_mask_len": (_i64, [_i64, _i32]),
    "gcl_bn_bwd_reduce": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_bn_bwd_apply": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_bn_bwd_reduce_ld": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_bn_bwd_apply_ld": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gcl_row_normalize_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "gcl_row_normalize_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "gcl_sgd_multi": (_i32, [_vp, _vp, _i32, _f32, _f32, _f32, _i32, _vp]),
    "gcl_col_sum": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "gcl_maps_arena_bytes": (_i64, [_i64, _vp, _i32, _i32]),
    "gcl_maps_build": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "gcl_maps_build_split": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "gcl_plan_create": (_vp, [_vp, _i32, _i32, _i32, _vp, _i32, _i32]),
    "gcl_plan_destroy": (None, [_vp]),
    "gcl_plan_state_bytes": (_i64, [_vp]),
    "gcl_plan_arena_bytes": (_i64, [_vp, _vp]),
    "gcl_plan_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gcl_plan_eval_state_bytes": (_i64, [_vp]),
    "gcl_plan_eval_arena_bytes": (_i64, [_vp, _vp]),
    "gcl_plan_forward_eval": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _vp]),
    "gcl_plan_backward": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "gcl_plan_release": (_i32, [_vp, _vp]),
    "gcl_plan_set_aux_stream": (_i32, [_vp, _vp]),
    "gcl_stream_create_cu_share": (_i32, [_i32, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    "gcl_stream_destroy": (_i32, [_vp]),
    "gcl_plan_profile": (_i32, [_vp, _i32]),
    "gcl_plan_profile_read": (_i32, [_vp, _vp, _i32]),
    "gcl_sc2_chunks": (_i32, []),
    "gcl_sc2_refine_partial_len": (_i32, []),
    "gcl_sc2_confidence": (_i32, [_vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "gcl_sc2_confidence_scratch_bytes": (_i64, [_i32]),
    "gcl_sc2_confidence_sparse": (_i32, [_vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_sc2_register_scratch_bytes": (_i64, [_i32]),
    "gcl_sc2_register": (_i32, [_vp, _vp, _i32, _f32, _i32, _f32, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp]),
    "gcl_sc2_local_max": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    "gcl_sc2_seed_knn": (_i32, [_vp, _vp, _i32, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    "gcl_sc2_seed_trans": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _f32, _vp, _vp, _vp]),
    "gcl_sc2_refine": (_i32, [_vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "gcl_group_loss_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "gcl_group_loss_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_circle_group_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gcl_circle_group_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp,
                                    _vp]),
    "gcl_nn_rowmin_scratch_len": (_i64, [_i32, _i32]),
    "gcl_nn_rowmin": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "gcl_neg_mask": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "gcl_neg_loss_fwd": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    "gcl_neg_loss_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp]),
    "gcl_loss_combine": (_i32, [_vp, _vp, _i32, _vp, _f32, _f32, _f32, _vp, _vp]),
    "gcl_loss_seed": (_i32, [_vp, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
}

PAIR_CHUNK = 128   # GCL_PAIR_CHUNK

_lib = None


TRAINING_STEP_SOURCES = ("coords.hip", "conv.hip", "norm.hip", "plan.hip")      # what a training step's kernels are built from


def source_hash():
    """sha256 (16 hex digits) over the sources of the TRAINING step's kernels (maps, convolutions, BatchNorm, the plan) + the
    shared device header: stamps profiles/pmc_summary.json, so that bench.py can tell whether the counter figures it quotes
    were collected on the kernels it is timing.  (loss.hip / data.hip / sc2pcr.hip hold no kernel the summary lists; the public
    header declares entry points, it holds no kernel code.)"""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in TRAINING_STEP_SOURCES] + [os.path.join(CSRC, "common.h")]:
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libgcl_hip.so (in-tree, so it travels to the GPU box)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "common.h"), HEADER]
    deps = srcs + hdrs
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # one object per translation unit, stale ones compiled concurrently, then one link
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objs, jobs = [], []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def load():
    """Load the library and bind every symbol of the header; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(gcl_amd has no CPU fallback; the HIP library is the only product path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_gpu_ok = False


def require_gpu():
    global _gpu_ok
    if not _gpu_ok:
        if not torch.cuda.is_available():
            raise RuntimeError("gcl_amd needs an AMD GPU (gfx950); torch.cuda.is_available() is False and there is "
                               "no CPU path")
        _gpu_ok = True
    return _lib if _lib is not None else load()


def check(rc, what=""):
    if rc != 0:
        msg = load().gcl_last_error()
        raise RuntimeError(f"libgcl_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("gcl_amd operators take GPU tensors only (no CPU path)")
    if t.device.index != torch._C._cuda_getDevice():
        # kernels are launched on the CURRENT device's stream: a tensor that lives elsewhere would be read / written by
        # the wrong GPU, unordered against torch's own work on its device
        raise RuntimeError(f"gcl_amd: tensor on cuda:{t.device.index} but the current device is "
                           f"cuda:{torch._C._cuda_getDevice()}; call torch.cuda.set_device({t.device.index}) (one "
                           "process per GPU) or wrap the call in `with torch.cuda.device(t.device):`")
    if not t.is_contiguous():
        raise RuntimeError("gcl_amd: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"gcl_amd: expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    """Raw handle of torch's current HIP stream on the current device (the fast C accessors: this is called once per
    launch, ~700 times per training step)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def host_i64(values):
    arr = (ctypes.c_int64 * len(values))(*[int(v) for v in values])
    return arr
