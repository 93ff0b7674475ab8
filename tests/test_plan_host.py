"""Host half of the native step runtime (no GPU): record validation in gcl_plan_create, the dry-run arena sizing of
gcl_maps_arena_bytes / gcl_plan_arena_bytes, and the bucket segmentation of NetworkPlan."""
import ctypes

import pytest

from gcl_amd import _lib


def _op(**kw):
    d = dict(kind=0, x=-1, x2=-1, y=-1, level_in=0, level_out=0, cin=0, cout=0, map=-1, transpose=0, K=1, w=-1, bias=-1,
             bn_w=-1, bn_b=-1, bn=-1, relu=0, momentum=0.0, eps=1e-5)
    d.update(kw)
    return d


def _mini_records():
    """stem convbn -> convbn 3^3 (relu) -> convbn 3^3 (+ residual, relu) -> stride-2 convbn -> transposed convbn ->
    cat -> 1x1 conv -> relu -> 1x1 conv + bias -> rownorm"""
    C = _lib.OP_CONVBN
    return [
        _op(kind=C, x=0, y=1, cin=1, cout=32, map=0, K=125, w=0, bn_w=1, bn_b=2, bn=0),
        _op(kind=C, x=1, y=2, cin=32, cout=32, map=1, K=27, w=3, bn_w=4, bn_b=5, bn=1, relu=1),
        _op(kind=C, x=2, x2=1, y=3, cin=32, cout=32, map=1, K=27, w=6, bn_w=7, bn_b=8, bn=2, relu=1),
        _op(kind=C, x=3, y=4, level_out=1, cin=32, cout=64, map=2, K=27, w=9, bn_w=10, bn_b=11, bn=3, relu=1),
        _op(kind=C, x=4, y=5, level_in=1, level_out=0, cin=64, cout=32, map=2, transpose=1, K=27, w=12, bn_w=13, bn_b=14,
            bn=4, relu=1),
        _op(kind=_lib.OP_CAT, x=5, x2=3, y=6, cin=32, cout=64),
        _op(kind=_lib.OP_CONV, x=6, y=7, cin=64, cout=64, map=3, K=1, w=15),
        _op(kind=_lib.OP_RELU, x=7, y=8, cin=64, cout=64),
        _op(kind=_lib.OP_CONV, x=8, y=9, cin=64, cout=32, map=3, K=1, w=16, bias=17),
        _op(kind=_lib.OP_ROWNORM, x=9, y=10, cin=32, cout=32),
    ]


def _create(records, n_tensors=11, n_params=18, worder=(3, 6, 9, 12, 15, 16)):
    lib = _lib.load()
    arr = (_lib.PlanOp * len(records))()
    for a, r in zip(arr, records):
        for k, v in r.items():
            setattr(a, k, v)
    wo = (ctypes.c_int32 * max(1, len(worder)))(*worder)
    return lib.gcl_plan_create(arr, len(records), n_tensors, n_params, wo, len(worder), 128)


def _maps(n=10000, n1=3000):
    d = _lib.MapsDesc()
    d.n_levels, d.n_maps = 2, 4
    d.n_rows[0], d.n_rows[1] = n, n1
    for i, (t, ks, st, li, lo) in enumerate([(1, 5, 1, 0, 0), (1, 3, 1, 0, 0), (1, 3, 2, 0, 1), (1, 1, 1, 0, 0)]):
        m = d.maps[i]
        m.t_in, m.kernel_size, m.stride, m.K, m.level_in, m.level_out = t, ks, st, ks ** 3, li, lo
        m.n_in, m.n_out = d.n_rows[li], d.n_rows[lo]
        m.n_pairs = 8 * m.n_out
        for k in range(m.K + 1):
            m.seg_off[k] = k * 1024
    return d


def test_plan_create_accepts_a_resunet_shaped_graph_and_sizes_its_arena():
    lib = _lib.load()
    h = _create(_mini_records())
    assert h, lib.gcl_last_error()
    h = ctypes.c_void_p(h)
    assert lib.gcl_plan_state_bytes(h) >= 6 * 18 * 8
    d = _maps()
    small = lib.gcl_plan_arena_bytes(h, ctypes.byref(d))
    big = lib.gcl_plan_arena_bytes(h, ctypes.byref(_maps(20000, 6000)))
    # activations alone: 10 k rows x (32 * 7 + 64 * 4) floats forward, and at least as much again backward
    assert small > 10000 * 4 * (32 * 7 + 64 * 4) and 1.8 * small < big < 2.2 * small
    d.maps[2].level_out = 0          # the records and the maps must agree on the levels
    assert lib.gcl_plan_arena_bytes(h, ctypes.byref(d)) < 0 and b"levels" in lib.gcl_last_error()
    # without a GPU the forward entry still validates its arguments before any HIP call
    assert lib.gcl_plan_forward(h, None, None, None, None, None, None, 0, None, None) == -1
    assert lib.gcl_plan_backward(h, None, None, (ctypes.c_void_p * 18)(), 0, 10, None) == -1
    assert b"no forward pass" in lib.gcl_last_error()
    lib.gcl_plan_destroy(h)


@pytest.mark.parametrize("mutate,why", [
    (lambda r: r[1].update(x=5), "consumes a tensor produced later"),
    (lambda r: r[2].update(w=3), "one parameter in two records (gradients are written, not added)"),
    (lambda r: r[6].update(cout=48), "generic shapes stay on the per-operator path"),
    (lambda r: r[5].update(x2=-1), "cat needs two inputs"),
    (lambda r: r[8].update(kind=_lib.OP_CONVBN), "bias with BatchNorm"),
    (lambda r: r[0].update(kind=9), "unknown kind"),
])
def test_plan_create_rejects_malformed_graphs(mutate, why):
    rec = _mini_records()
    mutate(rec)
    assert not _create(rec), why
    assert b"gcl_plan_create" in _lib.load().gcl_last_error()


def test_maps_arena_bound_grows_with_the_cloud_and_covers_every_spec():
    lib = _lib.load()
    specs = [(1, 5, 1, 0, 0), (1, 3, 1, 1, 1), (1, 3, 2, 3, 1), (2, 3, 1, 1, 1), (1, 1, 1, 0, 1)]
    arr = (_lib.MapSpec * len(specs))()
    for a, s in zip(arr, specs):
        a.t_in, a.kernel_size, a.stride, a.tables, a.pairs = s
    a = lib.gcl_maps_arena_bytes(100000, arr, len(specs), 4)
    b = lib.gcl_maps_arena_bytes(200000, arr, len(specs), 4)
    assert a > 100000 * 4 * (125 + 2 * 27 * 3) and 1.7 * a < b < 2.3 * a
    assert lib.gcl_maps_arena_bytes(0, arr, len(specs), 4) < 0
    arr[2].t_in = 16                                # outside the 4 levels
    assert lib.gcl_maps_arena_bytes(1000, arr, len(specs), 4) < 0


def test_bucket_segments_follow_the_lowest_record_of_each_bucket():
    from gcl_amd.MinkowskiEngine.native import NetworkPlan
    plan = NetworkPlan.__new__(NetworkPlan)
    plan.handle = None
    plan.records = _mini_records()
    plan.bucket_of_param, plan.on_bucket = None, None
    assert plan._segments() == [(0, 10, [])]
    # parameters 0..8 (records 0-2) in bucket 0, the rest in bucket 1: bucket 1 is complete once record 3 has run
    plan.bucket_of_param = {p: (0 if p < 9 else 1) for p in range(18)}
    plan.on_bucket = lambda b: None
    assert plan._segments() == [(3, 10, [1]), (0, 3, [0])]
