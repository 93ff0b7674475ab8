"""GPU parity tests proper: every HIP kernel, reached through the C ABI / the ME-compatible surface, against the
CPU oracle (oracle/) on the same seeded inputs, and against the golden vectors captured from the reference.

Bar: bit-exact for coordinate maps / kernel maps / indices; for fp32 features the north star's tolerance is
1e-4 relative L2 against the (fp64) oracle -- the per-operator bounds below are tighter and written in each test.
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_oracle as LO          # noqa: E402
from oracle import me_oracle as O             # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def random_cloud(seed, n=3000, extent=24, batch=2, sheet=True):
    """Unique int32 coords [N,4]; a mix of a thin sheet (LiDAR-like) and a blob, negative coordinates included."""
    rng = np.random.RandomState(seed)
    cs = []
    for b in range(batch):
        pts = rng.randint(-extent, extent, (n, 3))
        if sheet:
            pts[: n // 2, 2] = rng.randint(-1, 1, n // 2)
        c = np.unique(pts, axis=0)
        rng.shuffle(c)
        cs.append(np.concatenate([np.full((len(c), 1), b), c], axis=1))
    return np.concatenate(cs).astype(np.int32)


def make_mgr(C):
    import gcl_amd.MinkowskiEngine as ME
    return ME.CoordinateManager(torch.from_numpy(C).to(DEV))


# ---------------------------------------------------------------------------------------------------------------
# integer part: bit-exact
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,n,batch", [(0, 3000, 2), (1, 37, 1), (2, 20000, 3), (3, 1, 1)])
def test_stride_maps_bit_exact(seed, n, batch):
    C = random_cloud(seed, n=n, batch=batch)
    mgr = make_mgr(C)
    omgr = O.CoordinateManager(C)
    for t in (2, 4, 8):
        got = mgr.get_coords(t).cpu().numpy()
        ref = omgr.get_coords(t)
        assert got.shape == ref.shape, (t, got.shape, ref.shape)
        assert np.array_equal(got, ref), f"stride {t}: rows differ (first-occurrence order expected)"


@pytest.mark.parametrize("t_in,ks,stride", [(1, 3, 1), (1, 5, 1), (1, 3, 2), (2, 3, 1), (2, 3, 2), (4, 3, 2), (8, 3, 1)])
def test_kernel_maps_bit_exact(t_in, ks, stride):
    C = random_cloud(5, n=6000, batch=2)
    mgr = make_mgr(C)
    omgr = O.CoordinateManager(C)
    got = O.canonical(mgr.kernel_map_triples(t_in, ks, stride))
    ref = O.canonical(omgr.get_kernel_map(t_in, ks, stride))
    assert got.shape == ref.shape and np.array_equal(got, ref)
    km = mgr.get_kernel_map(t_in, ks, stride)
    assert km.n_pairs == len(ref)
    # transposed table and compacted pair lists describe the same set of triples
    if km.nbr_t is not None:
        nt = km.nbr_t.cpu()
        k, u = torch.nonzero(nt >= 0, as_tuple=True)
        tri_t = torch.stack([k, u, nt[k, u].long()], 1).numpy()
        assert np.array_equal(O.canonical(tri_t), ref)
    pin, pout, seg, _ = km.pairs()
    pin, pout = pin.cpu().numpy(), pout.cpu().numpy()
    tri_p = []
    for k in range(km.K):
        a, b = pin[seg[k]:seg[k + 1]], pout[seg[k]:seg[k + 1]]
        valid = a >= 0
        assert valid.sum() == km.counts[k] and (valid[: km.counts[k]]).all(), "padding must trail the segment"
        assert (np.diff(b[valid]) > 0).all(), "pairs of one offset are sorted by output row"
        tri_p.append(np.stack([np.full(valid.sum(), k), a[valid], b[valid]], 1))
    assert np.array_equal(O.canonical(np.concatenate(tri_p)), ref)


@pytest.mark.parametrize("ks,stride,n", [(3, 1, 6000), (5, 1, 3000), (3, 2, 9000), (3, 1, 40000)])
def test_kernel_map_count_and_bitmap_variants_agree(ks, stride, n):
    """gcl_kernel_map's round-6 variants through the C ABI: per-offset counts by integer atomics into zeroed counts (flag bit 2;
    builds of <= 64 blocks -- the 40 000-row case falls back to the ordered reduction) and no presence bitmap (tables that are
    cache-sized themselves) give the neighbour tables and counts of the plain call, bit for bit."""
    from gcl_amd import _lib
    lib = _lib.require_gpu()
    C = random_cloud(ks * 1000 + n, n=n, extent=40 if n > 10000 else 24, batch=2)
    mgr = make_mgr(C)
    t_in, t_out = 1, stride
    C_in, C_out = mgr.get_coords(t_in), mgr.get_coords(t_out)
    mgr._input_table()
    _, table_in, cap_in = mgr._maps[t_in]
    n_in, n_out, K, same = C_in.shape[0], C_out.shape[0], ks ** 3, stride == 1
    i32 = dict(dtype=torch.int32, device=DEV)
    outs = []
    for flag4, with_bitmap in ((0, True), (4, True), (4, False), (0, False)):
        nbr = torch.full((K, n_out), -7, **i32)
        nbr_t = None if same else torch.full((K, n_in), -7, **i32)
        counts = torch.zeros(K, **i32) if flag4 else torch.full((K,), -7, **i32)
        bitmap = torch.empty(lib.gcl_kernel_map_bitmap_len(), **i32) if with_bitmap else None
        scratch = torch.empty(lib.gcl_kernel_map_scratch_len(ks, n_out), **i32)
        _lib.check(lib.gcl_kernel_map(_lib.ptr(C_out), n_out, _lib.ptr(table_in), cap_in, ks, t_in, int(same) | flag4,
                                      _lib.ptr(bitmap), _lib.ptr(scratch), _lib.ptr(nbr), _lib.ptr(nbr_t), n_in, _lib.ptr(counts),
                                      _lib.stream()), "gcl_kernel_map")
        outs.append((nbr, nbr_t, counts))
    ref = outs[0]
    assert int(ref[2].sum()) == int((ref[0] >= 0).sum())
    for nbr, nbr_t, counts in outs[1:]:
        assert torch.equal(nbr, ref[0]) and torch.equal(counts, ref[2])
        assert (nbr_t is None and ref[1] is None) or torch.equal(nbr_t, ref[1])


@pytest.mark.parametrize("n", [1, 100, 4096, 8192, 8193, 40000, 524288, 524289, 700000])
def test_exclusive_scan_all_launch_shapes(n):
    """gcl_exclusive_scan_i32 against numpy over its three launch shapes: one workgroup (<= 8192 entries), block totals added
    up by the final pass itself (two launches, <= 256 blocks), block totals scanned by their own launch (three)."""
    from gcl_amd import _lib
    lib = _lib.require_gpu()
    rng = np.random.RandomState(n)
    v = rng.randint(0, 5, n).astype(np.int32)
    x = torch.from_numpy(v).to(DEV)
    out = torch.empty(n, dtype=torch.int32, device=DEV)
    scratch = torch.empty(lib.gcl_scan_scratch_len(n), dtype=torch.int32, device=DEV)
    _lib.check(lib.gcl_exclusive_scan_i32(_lib.ptr(x), n, _lib.ptr(out), _lib.ptr(scratch), _lib.stream()), "gcl_exclusive_scan_i32")
    ref = np.concatenate([[0], np.cumsum(v[:-1], dtype=np.int64)]).astype(np.int32)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_coordinate_errors():
    import gcl_amd.MinkowskiEngine as ME
    C = random_cloud(0, n=100, batch=1)
    dup = np.concatenate([C, C[:3]])
    with pytest.raises(ValueError, match="duplicate"):
        ME.CoordinateManager(torch.from_numpy(dup).to(DEV)).get_kernel_map(1, 3, 1)
    far = C.copy()
    far[0, 1] = 40000
    with pytest.raises(ValueError, match="range"):
        ME.CoordinateManager(torch.from_numpy(far).to(DEV)).get_kernel_map(1, 3, 1)
    with pytest.raises(RuntimeError, match="no CPU backend"):
        ME.SparseTensor(torch.ones(len(C), 1), coordinates=torch.from_numpy(C))


def test_extreme_coordinates_and_dense_block_maps_bit_exact():
    """Edge cases of the coordinate hash: the corners of the packable range (|x| = 32767, batch 32767), a FULL 20^3 block
    (every one of the 27 / 125 neighbours present: longest probe chains, no bitmap rejections) and its stride maps."""
    g = np.arange(-10, 10, dtype=np.int32)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    np.random.RandomState(0).shuffle(cube)
    corners = np.array([[32767, 32767, 32767], [-32767, -32767, -32767], [32767, -32767, 0], [32766, 32767, 32767]],
                       dtype=np.int32)
    C = np.concatenate([np.concatenate([np.zeros((len(cube), 1), np.int32), cube], 1),
                        np.concatenate([np.full((len(corners), 1), 32767, np.int32), corners], 1)]).astype(np.int32)
    mgr, omgr = make_mgr(C), O.CoordinateManager(C)
    for t in (2, 4, 8):
        assert np.array_equal(mgr.get_coords(t).cpu().numpy(), omgr.get_coords(t))
    for key in [(1, 3, 1), (1, 5, 1), (1, 3, 2), (2, 3, 1)]:
        got, ref = O.canonical(mgr.kernel_map_triples(*key)), O.canonical(omgr.get_kernel_map(*key))
        assert got.shape == ref.shape and np.array_equal(got, ref), key
    assert mgr.get_kernel_map(1, 3, 1).n_pairs > 26 * 18 ** 3          # interior voxels have all 27 neighbours


def test_empty_and_single_voxel_inputs():
    """No rows: a clean error, not a launch with an empty grid.  One voxel: every level has one row and the whole
    network still runs (eval statistics), matching the oracle."""
    import gcl_amd.MinkowskiEngine as ME
    with pytest.raises((ValueError, RuntimeError)):
        ME.SparseTensor(torch.ones(0, 1, device=DEV), coordinates=torch.zeros((0, 4), dtype=torch.int32, device=DEV))
    m, st = _model_and_state(3, 5)
    m.eval()
    C = torch.tensor([[0, 3, -2, 1]], dtype=torch.int32)
    with torch.no_grad():
        y = m(ME.SparseTensor(torch.ones(1, 1, device=DEV), coordinates=C.to(DEV))).F
    ref = O.resunet_forward(st, C.numpy(), torch.ones(1, 1, dtype=torch.float64), 5, True, False, 0.05)
    assert y.shape == (1, 32) and rel_l2(y.cpu(), ref.detach()) < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# convolution forward / input gradient / weight gradient
# ---------------------------------------------------------------------------------------------------------------
def _conv_case(cin, cout, ks, stride, transpose, seed=0, n=2500, bias=False):
    import gcl_amd.MinkowskiEngine as ME
    C = random_cloud(seed, n=n, batch=2)
    mgr = make_mgr(C)
    omgr = O.CoordinateManager(C)
    g = torch.Generator().manual_seed(seed)
    t_in = 2 if transpose else 1
    n_in = len(omgr.get_coords(t_in))
    x = torch.randn(n_in, cin, generator=g, dtype=torch.float64)
    cls = ME.MinkowskiConvolutionTranspose if transpose else ME.MinkowskiConvolution
    conv = cls(cin, cout, kernel_size=ks, stride=stride, bias=bias, dimension=3).to(DEV)
    W = conv.kernel.detach().cpu().double()
    xg = x.float().to(DEV).requires_grad_(cin > 4)
    st = ME.SparseTensor(xg, coordinate_map_key=ME.CoordinateMapKey(t_in), coordinate_manager=mgr)
    y = conv(st).F
    # oracle
    xo = x.clone().requires_grad_(True)
    Wo = W.clone().requires_grad_(True)
    if ks == 1:
        tri = np.stack([np.zeros(n_in, np.int64), np.arange(n_in), np.arange(n_in)], 1)
        n_out = n_in
    elif transpose:
        tri, n_out = omgr.get_kernel_map(t_in // stride, ks, stride), len(omgr.get_coords(t_in // stride))
    else:
        tri, n_out = omgr.get_kernel_map(t_in, ks, stride), len(omgr.get_coords(t_in * stride))
    bo = conv.bias.detach().cpu().double() if bias else None
    yo = O.sparse_conv(xo, Wo, tri, n_out, transpose=transpose, bias=bo)
    assert y.shape == yo.shape
    gy = torch.randn(yo.shape, generator=g, dtype=torch.float64)
    yo.backward(gy)
    y.backward(gy.float().to(DEV))
    return dict(y=(y.detach().cpu(), yo.detach()), dW=(conv.kernel.grad.cpu(), Wo.grad.reshape(conv.kernel.shape)),
                dx=(xg.grad.cpu() if cin > 4 else None, xo.grad),
                db=(conv.bias.grad.cpu() if bias else None, gy.sum(0, keepdim=True)))


CONV_CASES = [
    (32, 32, 3, 1, False), (32, 64, 3, 2, False), (64, 64, 3, 1, False), (64, 128, 3, 2, False),
    (128, 128, 3, 1, False), (128, 256, 3, 2, False), (256, 256, 3, 1, False),
    (256, 128, 3, 2, True), (256, 64, 3, 2, True), (128, 64, 3, 2, True),
    (96, 64, 1, 1, False), (64, 32, 1, 1, False),
    # generic shapes (exact-fp32 VALU kernels behind the same C-ABI entry points): demo.py:29's 16-dim head, channel
    # counts that are no multiple of 32 / 16 / 4, a 5^3 kernel on wide features, a transposed generic conv
    (64, 16, 1, 1, False), (48, 80, 3, 1, False), (16, 16, 3, 2, False), (30, 7, 3, 1, False), (32, 32, 5, 1, False),
    (80, 48, 3, 2, True),
]


# per-operator tolerance (relative L2 vs the fp64 oracle) of each MFMA arithmetic: exact f32 and the 3-plane bf16
# split are indistinguishable from fp32 rounding; the 2-plane split drops ~2^-17 of each product
PREC_TOL = {"f32": 2e-6, "bf16x6": 2e-6, "bf16x3": 3e-5, "fp16x3": 2e-6}


@pytest.fixture(params=["fp16x3", "bf16x6", "f32", "bf16x3"])
def precision(request):
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import ops
    old = ops.PRECISION
    ME.set_conv_precision(request.param)
    yield request.param
    ME.set_conv_precision(old)


@pytest.mark.parametrize("cin,cout,ks,stride,transpose", CONV_CASES)
def test_conv_fwd_bwd_vs_oracle(cin, cout, ks, stride, transpose, precision):
    r = _conv_case(cin, cout, ks, stride, transpose, bias=(cout in (16, 32) and ks == 1))
    tol = PREC_TOL[precision]
    assert rel_l2(*r["y"]) < tol, rel_l2(*r["y"])
    assert rel_l2(*r["dx"]) < tol, rel_l2(*r["dx"])
    assert rel_l2(*r["dW"]) < tol, rel_l2(*r["dW"])
    if r["db"][0] is not None:
        assert rel_l2(*r["db"]) < 2e-6


@pytest.mark.parametrize("cin,ks", [(1, 5), (1, 3), (3, 5)])
def test_stem_conv_vs_oracle(cin, ks):
    r = _conv_case(cin, 32, ks, 1, False, seed=3)
    assert rel_l2(*r["y"]) < 2e-6
    assert rel_l2(*r["dW"]) < 2e-6


@pytest.mark.parametrize("ks,n", [(5, 3000), (3, 700), (5, 1)])
def test_stem_occupancy_path_is_bitwise_the_general_path(ks, n):
    """Occupancy rows (features exactly 1.0: what the reference's loaders feed for every cloud but a training sample's
    jittered centre cloud, lib/colocation_data_loader.py:401-415): gcl_stem_fwd / gcl_stem_bwd_weight given presence words
    + per-row flags produce bit for bit what they produce from the neighbour table -- with every cloud all ones, with one
    cloud of the batch carrying other values (its rows walk the table, the other clouds' rows use the words), with all."""
    from gcl_amd import _lib
    lib = _lib.load()
    C = random_cloud(5, n=n, extent=12, batch=3) if n > 1 else np.zeros((1, 4), np.int32)
    mgr = make_mgr(C)
    kmap = mgr.get_kernel_map(1, ks, 1)
    K, n_out = kmap.nbr.shape[0], len(C)
    words = (K + 31) // 32
    batch_idx = torch.from_numpy(C[:, 0].astype(np.int64)).to(DEV)
    with torch.cuda.device(DEV):
        Cd = torch.from_numpy(np.ascontiguousarray(C, dtype=np.int32)).to(DEV)
        bits = torch.empty(n_out * words, dtype=torch.int32, device=DEV)
        _lib.check(lib.gcl_presence_bits(_lib.ptr(kmap.nbr), K, n_out, _lib.ptr(bits), _lib.stream()), "gcl_presence_bits")
        nb = kmap.nbr.cpu().numpy()
        want = np.zeros((n_out, words), np.uint32)
        for k in range(K):
            want[:, k // 32] |= (nb[k] >= 0).astype(np.uint32) << np.uint32(k % 32)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32).reshape(n_out, words), want)
        g = torch.Generator().manual_seed(ks)
        W = torch.randn(K, 1, 32, generator=g).to(DEV)
        dy = torch.randn(n_out, 32, generator=g).to(DEV)
        noise = 1.0 + 0.01 * torch.randn(n_out, 1, generator=g).to(DEV)
        n_clouds = int(C[:, 0].max()) + 1
        for jittered in ((), (n_clouds - 1,), tuple(range(n_clouds))):
            x = torch.ones(n_out, 1, device=DEV)
            for b in jittered:
                x = torch.where((batch_idx == b)[:, None], noise, x)
            for n_flags in (4096, 1):          # 1: batch indices beyond the scratch are flagged (table path), same bits
                cloud = torch.full((n_flags,), -1, dtype=torch.int32, device=DEV)
                rows = torch.full((n_out,), -1, dtype=torch.int32, device=DEV)
                _lib.check(lib.gcl_not_ones_rows(_lib.ptr(x), 1, _lib.ptr(Cd), n_out, _lib.ptr(cloud), n_flags, _lib.ptr(rows),
                                                 _lib.stream()), "gcl_not_ones_rows")
                want_rows = np.isin(C[:, 0], [b for b in jittered if (x[batch_idx == b] != 1).any()]) | (C[:, 0] >= n_flags)
                assert np.array_equal(rows.cpu().numpy(), want_rows.astype(np.int32)), (jittered, n_flags)
                ys, dws = [], []
                for pres in (None, bits):
                    y = torch.full((n_out, 32), float("nan"), device=DEV)
                    _lib.check(lib.gcl_stem_fwd(_lib.ptr(x), _lib.ptr(W), _lib.ptr(kmap.nbr), n_out, K, 1, 32, _lib.ptr(y),
                                                _lib.ptr(pres) if pres is not None else None,
                                                _lib.ptr(rows) if pres is not None else None, _lib.stream()), "gcl_stem_fwd")
                    scratch = torch.empty(lib.gcl_stem_bwd_weight_scratch_len(K, 1, 32, n_out), dtype=torch.float32, device=DEV)
                    dw = torch.empty(K, 1, 32, device=DEV)
                    _lib.check(lib.gcl_stem_bwd_weight(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(kmap.nbr), n_out, K, 1, 32,
                                                       _lib.ptr(scratch), _lib.ptr(dw),
                                                       _lib.ptr(pres) if pres is not None else None,
                                                       _lib.ptr(rows) if pres is not None else None, _lib.stream()),
                               "gcl_stem_bwd_weight")
                    ys.append(y), dws.append(dw)
                assert torch.equal(ys[0], ys[1]) and torch.equal(dws[0], dws[1]), (jittered, n_flags)


@pytest.mark.parametrize("n", [1, 31, 33, 127, 129, 1000])
def test_conv_ragged_row_counts(n, precision):
    """Tail handling: row counts around the 32-row wave tile and the 128-row workgroup tile."""
    import gcl_amd.MinkowskiEngine as ME
    rng = np.random.RandomState(n)
    pts = np.unique(rng.randint(-6, 6, (4 * n + 8, 3)), axis=0)[:n]
    C = np.concatenate([np.zeros((len(pts), 1), int), pts], 1).astype(np.int32)
    mgr, omgr = make_mgr(C), O.CoordinateManager(C)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(len(C), 32, generator=g, dtype=torch.float64)
    conv = ME.MinkowskiConvolution(32, 64, kernel_size=3, stride=1, dimension=3).to(DEV)
    xs = x.float().to(DEV).requires_grad_(True)
    y = conv(ME.SparseTensor(xs, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
    W = conv.kernel.detach().cpu().double().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = O.sparse_conv(xo, W, omgr.get_kernel_map(1, 3, 1), len(C))
    assert rel_l2(y.detach().cpu(), yo.detach()) < PREC_TOL[precision]
    gy = torch.randn(yo.shape, generator=g, dtype=torch.float64)
    yo.backward(gy)
    y.backward(gy.float().to(DEV))
    assert rel_l2(xs.grad.cpu(), xo.grad) < PREC_TOL[precision]
    assert rel_l2(conv.kernel.grad.cpu(), W.grad) < PREC_TOL[precision]


@pytest.mark.parametrize("cin,cout", [(128, 128), (256, 256), (128, 64), (256, 128)])
def test_inference_launch_with_offset_groups(cin, cout):
    """Flag GCL_CONV_TALL (inference launches): sixteen waves per workgroup, the offsets of a tile in four fixed groups
    (k mod 4) whose partial sums are added in group order.  Against the launch without the flag: equal to rounding (another summation
    order); repeatable bit for bit; and a row's result does not depend on what else is in the launch -- two clouds run
    together give, row for row, the bits they give one by one (fused epilogue included)."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    K = 27
    g = torch.Generator().manual_seed(cin + cout)
    ca, cb = random_cloud(21, n=1800, extent=14, batch=1), random_cloud(22, n=700, extent=9, batch=1)
    cb = cb.copy()
    cb[:, 0] = 1
    both = np.concatenate([ca, cb])

    def run(C, x, res, flags):
        mgr = make_mgr(C)
        tbl, order, mask = mgr.get_kernel_map(1, 3, 1).sorted_table()
        n_out = len(C)
        y = torch.full((n_out, cout), float("nan"), device=DEV)
        slot = ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_conv_fwd_fused(_lib.ptr(x), n_out, 0, _lib.ptr(wp), 4, _lib.ptr(xa), _lib.ptr(wa), _lib.ptr(tbl),
                                          _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, _lib.ptr(shift), _lib.ptr(scale),
                                          _lib.ptr(res), 1, _lib.ptr(slot), _lib.ptr(y), None, flags, _lib.stream()),
                   "gcl_conv_fwd_fused")
        return y

    with torch.cuda.device(DEV):
        x = torch.randn(len(both), cin, generator=g).to(DEV)
        res = torch.randn(len(both), cout, generator=g).to(DEV)
        W = (0.1 * torch.randn(K, cin, cout, generator=g)).to(DEV)
        scale, shift = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.1 * torch.randn(cout, generator=g)).to(DEV)
        xa, wa = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_amax(_lib.ptr(x), x.numel(), _lib.ptr(xa), 1, _lib.stream()), "gcl_amax")     # one scale for all runs
        _lib.check(lib.gcl_amax(_lib.ptr(W), W.numel(), _lib.ptr(wa), 1, _lib.stream()), "gcl_amax")
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, 4), dtype=torch.uint8, device=DEV)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(W), K, cin, cout, 0, 4, _lib.ptr(wa), _lib.ptr(wp), _lib.stream()), "pack")
        na = len(ca)
        plain = run(both, x, res, 0)
        tall = run(both, x, res, 4)
        assert torch.isfinite(tall).all() and torch.equal(tall, run(both, x, res, 4))
        assert not torch.equal(tall, plain)                      # the flag did select the other kernel
        assert rel_l2(tall.cpu().double(), plain.cpu().double()) < 1e-6      # two fp32 summation orders of the same products
        ya = run(ca, x[:na].contiguous(), res[:na].contiguous(), 4)
        yb = run(cb, x[na:].contiguous(), res[na:].contiguous(), 4)
        assert torch.equal(tall[:na], ya) and torch.equal(tall[na:], yb)


@pytest.mark.parametrize("cin,cout,n,use_planes", [(128, 128, 3000, True), (256, 256, 1500, True), (128, 256, 700, True),
                                                   (256, 128, 129, True), (128, 128, 1, True), (128, 64, 2500, True),
                                                   (256, 64, 900, True), (128, 32, 600, True), (64, 64, 3000, False),
                                                   (32, 32, 2000, False), (32, 64, 700, False), (64, 128, 1500, False),
                                                   (96, 64, 300, False), (64, 32, 1, False), (128, 128, 500, False)])
def test_lds_dma_forward_kernel_is_bitwise_the_register_staged_kernel(cin, cout, n, use_planes):
    """k_conv_fwd_dma (operands staged by `buffer_load ... lds`: no staging registers, no ds_write; missing neighbours read
    zero through the buffer resource; flag GCL_CONV_DMA) against k_conv_fwd_split (flag GCL_CONV_NO_DMA), on plane images and
    on fp32 rows.  Same products added in the same order: y and the BatchNorm column-sum partials are equal bit for bit, with
    and without the fused epilogue, for every column-block width (NB = 4, 2, 1) and for ragged / single-row launches."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    C = random_cloud(cin + n, n=n, extent=14, batch=1) if n > 1 else np.zeros((1, 4), np.int32)
    mgr = make_mgr(C)
    km = mgr.get_kernel_map(1, 3, 1)
    tbl, order, mask = km.sorted_table()
    n_out, K = len(C), 27
    g = torch.Generator().manual_seed(n)
    with torch.cuda.device(DEV):
        x = torch.randn(n_out, cin, generator=g).to(DEV)
        W = (0.1 * torch.randn(K, cin, cout, generator=g)).to(DEV)
        res = torch.randn(n_out, cout, generator=g).to(DEV)
        xa, wa = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_amax(_lib.ptr(x), x.numel(), _lib.ptr(xa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(W), W.numel(), _lib.ptr(wa), 1, _lib.stream()), "gcl_amax")
        planes = torch.empty((n_out, cin), dtype=torch.int32, device=DEV)
        _lib.check(lib.gcl_split_planes(_lib.ptr(x), n_out, cin, _lib.ptr(xa), _lib.ptr(planes), _lib.stream()), "split")
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, 4), dtype=torch.uint8, device=DEV)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(W), K, cin, cout, 0, 4, _lib.ptr(wa), _lib.ptr(wp), _lib.stream()), "pack")
        out = {}
        for flags in (8, 2):
            for fused in (False, True):
                y = torch.full((n_out, cout), float("nan"), device=DEV)
                stats = torch.full((4, cout, (n_out + 127) // 128), float("nan"), device=DEV)
                slot = ME.ops.amax_slot(x.device)
                _lib.check(lib.gcl_conv_fwd_fused(_lib.ptr(planes if use_planes else x), n_out, int(use_planes), _lib.ptr(wp), 4,
                                                  _lib.ptr(xa), _lib.ptr(wa),
                                                  _lib.ptr(tbl), _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, None,
                                                  None, _lib.ptr(res) if fused else None, int(fused),
                                                  _lib.ptr(slot) if fused else None, _lib.ptr(y), _lib.ptr(stats), flags,
                                                  _lib.stream()), "gcl_conv_fwd_fused")
                out[(flags, fused)] = (y, stats, ME.ops.amax_value(slot))
        for fused in (False, True):
            a, b = out[(8, fused)], out[(2, fused)]
            assert torch.isfinite(a[0]).all() and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert torch.equal(a[2], b[2])
        # against the fp64 product as well (the two kernels could be wrong together)
        nb = tbl.cpu().numpy()
        xs = x.double().cpu()[:, :]
        yo = torch.zeros(n_out, cout, dtype=torch.float64)
        src = km.nbr.cpu().numpy()
        for k in range(K):
            live = src[k] >= 0
            yo[live] += xs[src[k][live]] @ W[k].double().cpu()
        assert rel_l2(out[(2, False)][0].cpu(), yo) < 2e-6


@pytest.mark.parametrize("cin,cout,K", [(32, 64, 1), (64, 64, 27)])
def test_fused_epilogue_relu_forward_and_threshold_backward(cin, cout, K):
    """gcl_conv_fwd_fused's two ReLU modes against the plain launch: relu = 1 is max(conv, 0) with max|y| published (conv1_tr ->
    MEF.relu, model/resunet.py:222-223), relu = 2 passes conv where `residual` > 0 and writes 0 elsewhere -- the backward of a
    ReLU applied in the consumer's input-gradient epilogue (aten::threshold_backward(g, y, 0))."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    C = random_cloud(9, n=5000, extent=20, batch=2)
    mgr = make_mgr(C)
    n_out = len(C)
    g = torch.Generator().manual_seed(cin + K)
    with torch.cuda.device(DEV):
        tbl = order = mask = None
        if K > 1:
            tbl, order, mask = mgr.get_kernel_map(1, 3, 1).sorted_table()
        x = torch.randn(n_out, cin, generator=g).to(DEV)
        W = (0.1 * torch.randn(K, cin, cout, generator=g)).to(DEV)
        gate = torch.randn(n_out, cout, generator=g).to(DEV)          # the "ReLU output" whose sign gates mode 2
        xa, wa = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_amax(_lib.ptr(x), x.numel(), _lib.ptr(xa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(W), W.numel(), _lib.ptr(wa), 1, _lib.stream()), "gcl_amax")
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, 4), dtype=torch.uint8, device=DEV)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(W), K, cin, cout, 0, 4, _lib.ptr(wa), _lib.ptr(wp), _lib.stream()), "pack")

        def run(relu, residual):
            y = torch.full((n_out, cout), float("nan"), device=DEV)
            slot = ME.ops.amax_slot(x.device)
            _lib.check(lib.gcl_conv_fwd_fused(_lib.ptr(x), n_out, 0, _lib.ptr(wp), 4, _lib.ptr(xa), _lib.ptr(wa), _lib.ptr(tbl),
                                              _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, None, None,
                                              _lib.ptr(residual), relu, _lib.ptr(slot), _lib.ptr(y), None, 0, _lib.stream()),
                       "gcl_conv_fwd_fused")
            return y, ME.ops.amax_value(slot)
        plain, _ = run(0, None)
        y1, a1 = run(1, None)
        assert torch.equal(y1, torch.clamp_min(plain, 0.0)) and float(a1) == float(y1.abs().max())
        y2, a2 = run(2, gate)
        want = torch.where(gate > 0, plain, torch.zeros_like(plain))
        assert torch.equal(y2, want) and float(a2) == float(want.abs().max())
        assert lib.gcl_conv_fwd_fused(_lib.ptr(x), n_out, 0, _lib.ptr(wp), 4, _lib.ptr(xa), _lib.ptr(wa), _lib.ptr(tbl),
                                      _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, None, None, None, 2, None,
                                      _lib.ptr(y2), None, 0, _lib.stream()) != 0          # mode 2 without its gate tensor


@pytest.mark.parametrize("cin,cout,use_planes", [(128, 128, True), (64, 64, False)])
def test_lds_dma_forward_kernel_race_screen(cin, cout, use_planes):
    """The LDS-DMA kernel orders its reads behind its DMAs by its own `s_waitcnt vmcnt(0)` + the workgroup barrier, and its
    DMAs behind its reads by program order + `lgkmcnt` -- nothing the compiler checks.  A wrong placement shows as RARE wrong
    tiles that come and go with timing, so: 40 launches of a 60 k-row layer, half of them while a second stream keeps the
    memory system busy, every one compared bit for bit with the register-staged kernel's result."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    C = random_cloud(5, n=60000, extent=40, batch=1, sheet=False)
    mgr = make_mgr(C)
    km = mgr.get_kernel_map(1, 3, 1)
    tbl, order, mask = km.sorted_table()
    n_out, K = len(C), 27
    g = torch.Generator().manual_seed(3)
    with torch.cuda.device(DEV):
        x = torch.randn(n_out, cin, generator=g).to(DEV)
        W = (0.1 * torch.randn(K, cin, cout, generator=g)).to(DEV)
        xa, wa = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_amax(_lib.ptr(x), x.numel(), _lib.ptr(xa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(W), W.numel(), _lib.ptr(wa), 1, _lib.stream()), "gcl_amax")
        planes = torch.empty((n_out, cin), dtype=torch.int32, device=DEV)
        _lib.check(lib.gcl_split_planes(_lib.ptr(x), n_out, cin, _lib.ptr(xa), _lib.ptr(planes), _lib.stream()), "split")
        wp = torch.empty(lib.gcl_pack_weights_bytes(K, cin, cout, 4), dtype=torch.uint8, device=DEV)
        _lib.check(lib.gcl_pack_weights(_lib.ptr(W), K, cin, cout, 0, 4, _lib.ptr(wa), _lib.ptr(wp), _lib.stream()), "pack")
        xin = planes if use_planes else x

        def run(flags):
            y = torch.full((n_out, cout), float("nan"), device=DEV)
            stats = torch.full((4, cout, (n_out + 127) // 128), float("nan"), device=DEV)
            _lib.check(lib.gcl_conv_fwd(_lib.ptr(xin), n_out, int(use_planes), _lib.ptr(wp), 4, _lib.ptr(xa), _lib.ptr(wa),
                                        _lib.ptr(tbl), _lib.ptr(order), _lib.ptr(mask), n_out, K, cin, cout, None, _lib.ptr(y),
                                        _lib.ptr(stats), flags, _lib.stream()), "gcl_conv_fwd")
            return y, stats
        ref = run(8)
        side = torch.cuda.Stream()
        junk = torch.randn(64 << 20, device=DEV)
        for it in range(40):
            if it % 2:
                with torch.cuda.stream(side):           # streaming traffic beside the launch: other timing, other latencies
                    junk.mul_(1.0001)
            got = run(2)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), it
        torch.cuda.synchronize()


@pytest.mark.parametrize("cin,cout,stride,transpose", [(32, 32, 1, False), (64, 64, 1, False), (32, 64, 1, False),
                                                        (64, 32, 2, False), (64, 64, 2, True)])
def test_weight_gradient_range_grouped_mode(cin, cout, stride, transpose):
    """Levels with >= 32768 rows on the sorted side of the pair lists take the range-grouped weight-gradient launch
    (one (row range, offset) cell per workgroup, the K cells of a range on one XCD; csrc/conv.hip RG mode): dW against
    the fp64 oracle at the per-operator bound, bitwise reproducible, and equal to the k-major launch to rounding.  The
    stride-2 forward case has too few out rows and stays on the k-major launch (same assertions)."""
    import gcl_amd.MinkowskiEngine as ME
    assert ME.ops.PRECISION == "fp16x3"
    C = random_cloud(77, n=60000, extent=40, batch=1, sheet=False)
    assert len(C) > 40000
    mgr, omgr = make_mgr(C), O.CoordinateManager(C)
    g = torch.Generator().manual_seed(cin + cout)
    cls = ME.MinkowskiConvolutionTranspose if transpose else ME.MinkowskiConvolution
    conv = cls(cin, cout, kernel_size=3, stride=stride, dimension=3).to(DEV)
    t_in = stride if transpose else 1
    n_in = mgr.num_rows(t_in)
    x = torch.randn(n_in, cin, generator=g, dtype=torch.float64)
    W = conv.kernel.detach().cpu().double().requires_grad_(True)
    km = omgr.get_kernel_map(1, 3, stride)
    n_out = len(C) if transpose else len(omgr.get_coords(stride))
    yo = O.sparse_conv(x.float().double(), W, km, n_out, transpose=transpose)
    gy = torch.randn(yo.shape, generator=g, dtype=torch.float64)
    yo.backward(gy)
    grads = []
    for _ in range(2):
        conv.kernel.grad = None
        xs = x.float().to(DEV)
        y = conv(ME.SparseTensor(xs, coordinate_map_key=ME.CoordinateMapKey(t_in), coordinate_manager=mgr)).F
        y.backward(gy.float().to(DEV))
        grads.append(conv.kernel.grad.clone())
    assert torch.equal(grads[0], grads[1]), "deterministic"
    assert rel_l2(grads[0].cpu(), W.grad) < 2e-6


@pytest.mark.parametrize("ca,cb,n", [(96, 64, 70001), (64, 32, 70001), (96, 64, 1), (64, 32, 15), (32, 32, 16), (128, 32, 4099),
                                     (64, 64, 263), (96, 32, 8193), (32, 64, 530321)])
def test_weight_gradient_row_stream_kernel_size_1(ca, cb, n):
    """gcl_conv_bwd_weight_rows (kernel_size-1 convolutions -- conv1_tr 96 -> 64, final 64 -> 32, model/resunet.py:153-171:
    dW = x^T dy over ALL rows, both operands streamed once, no pair list): against the fp64 product at the per-operator
    bound, equal to the pair-list kernel on identity pairs to rounding, bitwise reproducible; single rows, row counts that
    are no multiple of the 16-row step, more workgroups than full steps, and the benchmark's row count."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + ca)
    with torch.cuda.device(DEV):
        a = torch.randn(n, ca, generator=g).to(DEV)
        b = (torch.randn(n, cb, generator=g) * 3e-3).to(DEV)          # gradients are small: a second scale
        aa, ba = ME.ops.amax_slot(a.device), ME.ops.amax_slot(a.device)
        _lib.check(lib.gcl_amax(_lib.ptr(a), a.numel(), _lib.ptr(aa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(b), b.numel(), _lib.ptr(ba), 1, _lib.stream()), "gcl_amax")
        ln = lib.gcl_conv_bwd_weight_rows_scratch_len(ca, cb, 4, n)
        assert ln > 0 and lib.gcl_conv_bwd_weight_rows_scratch_len(ca, cb, 0, n) == 0      # exact-f32: the pair-list path
        assert lib.gcl_conv_bwd_weight_rows_scratch_len(160, 64, 4, n) == 0                  # shapes the stream does not take
        outs = []
        for _ in range(2):
            scratch = torch.full((ln,), float("nan"), device=DEV)
            dw = torch.full((ca, cb), float("nan"), device=DEV)
            _lib.check(lib.gcl_conv_bwd_weight_rows(_lib.ptr(a), _lib.ptr(b), n, ca, cb, 4, _lib.ptr(aa), _lib.ptr(ba),
                                                    _lib.ptr(scratch), _lib.ptr(dw), _lib.stream()), "gcl_conv_bwd_weight_rows")
            outs.append(dw)
        assert torch.equal(outs[0], outs[1]), "deterministic"
        ref = a.double().T @ b.double()
        assert rel_l2(outs[0].double().cpu(), ref.cpu()) < 2e-6
        # the pair-list kernel on identity pairs (what kernel_size-1 convolutions ran until round 4)
        pa, pb, seg, seg_host = ME.CoordinateManager(torch.zeros((1, 4), dtype=torch.int32, device=DEV)).identity_pairs(n)
        scratch = torch.empty(lib.gcl_conv_bwd_weight_scratch_len(1, ca, cb, seg[-1], 0), device=DEV)
        dw2 = torch.empty((ca, cb), device=DEV)
        _lib.check(lib.gcl_conv_bwd_weight(_lib.ptr(a), n, _lib.ptr(b), n, 0, 0, _lib.ptr(pa), _lib.ptr(pb), seg_host, 1, ca, cb,
                                           4, _lib.ptr(aa), _lib.ptr(ba), _lib.ptr(scratch), _lib.ptr(dw2), _lib.stream()),
                   "gcl_conv_bwd_weight")
        assert rel_l2(outs[0].double().cpu(), dw2.double().cpu()) < 2e-6


@pytest.mark.parametrize("ca,cb,n,stride,transpose", [(128, 128, 3000, 1, False), (256, 128, 1500, 1, False),
                                                      (128, 256, 40, 1, False), (256, 256, 700, 2, False),
                                                      (128, 128, 1, 1, False), (128, 128, 9000, 1, False),
                                                      (128, 128, 24000, 1, False), (256, 256, 26000, 2, False),
                                                      (256, 128, 26000, 2, True)])
def test_weight_gradient_128_block_kernel(ca, cb, n, stride, transpose):
    """k_conv_bwd_weight_wg128 (plane images, Ca and Cb multiples of 128: a 128 x 128 block of dW[k] per workgroup, the rows
    gathered once into a shared double-buffered tile, every wave a 64 x 64 block; the default there) against the 64 x 64-block
    kernel (bit 1 of `planes`) and the fp64 product: equal to rounding (one accumulator per block walks the pairs in order
    instead of four interleaved ones), bitwise reproducible; ragged / single-pair lists and several offsets per workgroup.
    The 24000 / 26000-voxel cases (round 6) run at the row counts of the benchmark's stride-4 / stride-8 levels, on a strided
    map with 4 channel tiles, and with ``transpose`` the swapped lists of a transposed convolution (sorted side = operand A);
    they are also the parity cases of the parked range-grouped mode (profiles/patches/r06_rg128_dw.patch)."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    C = random_cloud(ca + n, n=n, extent=14 if n < 10000 else 40, batch=1 if n < 10000 else 2) if n > 1 \
        else np.zeros((1, 4), np.int32)
    mgr = make_mgr(C)
    km = mgr.get_kernel_map(1, 3, stride)
    pin, pout, seg, seg_host = km.pairs()
    n_in, n_out = len(C), mgr.num_rows(stride)
    K = 27
    g = torch.Generator().manual_seed(n + ca)
    with torch.cuda.device(DEV):
        # forward convolution: A = x over the map's in rows (pair_in), B = dy over its out rows (pair_out, ascending per
        # offset: sorted side 2); transposed: A = x over the out rows (pair_out: sorted side 1), B = dy over the in rows
        n_a, n_b = (n_out, n_in) if transpose else (n_in, n_out)
        pa, pb, side = (pout, pin, 1) if transpose else (pin, pout, 2)
        a = torch.randn(n_a, ca, generator=g).to(DEV)
        b = torch.randn(n_b, cb, generator=g).to(DEV)
        aa, ba = ME.ops.amax_slot(a.device), ME.ops.amax_slot(a.device)
        _lib.check(lib.gcl_amax(_lib.ptr(a), a.numel(), _lib.ptr(aa), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_amax(_lib.ptr(b), b.numel(), _lib.ptr(ba), 1, _lib.stream()), "gcl_amax")
        xa = torch.empty((n_a, ca), dtype=torch.int32, device=DEV)
        xb = torch.empty((n_b, cb), dtype=torch.int32, device=DEV)
        _lib.check(lib.gcl_split_planes(_lib.ptr(a), n_a, ca, _lib.ptr(aa), _lib.ptr(xa), _lib.stream()), "split")
        _lib.check(lib.gcl_split_planes(_lib.ptr(b), n_b, cb, _lib.ptr(ba), _lib.ptr(xb), _lib.stream()), "split")
        out = {}
        for legacy in (1, 0, 0):
            scratch = torch.full((lib.gcl_conv_bwd_weight_scratch_len(K, ca, cb, seg[-1], n_out),), float("nan"), device=DEV)
            dw = torch.full((K, ca, cb), float("nan"), device=DEV)
            _lib.check(lib.gcl_conv_bwd_weight(_lib.ptr(xa), n_a, _lib.ptr(xb), n_b, 1 | (2 * legacy), side, _lib.ptr(pa),
                                               _lib.ptr(pb), seg_host, K, ca, cb, 4, _lib.ptr(aa), _lib.ptr(ba),
                                               _lib.ptr(scratch), _lib.ptr(dw), _lib.stream()), "gcl_conv_bwd_weight")
            out.setdefault(legacy, []).append(dw)
        assert torch.isfinite(out[0][0]).all() and torch.equal(out[0][0], out[0][1]), "deterministic"
        src = km.nbr.cpu().numpy()
        ad, bd = a.double().cpu(), b.double().cpu()
        want = torch.zeros(K, ca, cb, dtype=torch.float64)
        for k in range(K):
            rows = np.nonzero(src[k] >= 0)[0]
            if len(rows):
                want[k] = (ad[rows].T @ bd[src[k][rows]]) if transpose else (ad[src[k][rows]].T @ bd[rows])
        assert rel_l2(out[0][0].cpu(), want) < 2e-6 and rel_l2(out[1][0].cpu(), want) < 2e-6
        assert rel_l2(out[0][0].cpu(), out[1][0].cpu()) < 2e-6


# ---------------------------------------------------------------------------------------------------------------
# batch norm (+ residual, + relu)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,n,res,relu,training", [(32, 5000, False, False, True), (64, 4097, False, True, True),
                                                   (128, 1500, True, True, True), (256, 333, True, True, True),
                                                   (64, 2000, True, True, False), (32, 1, False, True, False)])
def test_batch_norm_vs_torch(c, n, res, relu, training):
    from gcl_amd.MinkowskiEngine.ops import batch_norm
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, generator=g, dtype=torch.float64) * 2 + 3)
    r = torch.randn(n, c, generator=g, dtype=torch.float64) if res else None
    w = torch.rand(c, generator=g, dtype=torch.float64) + 0.5
    b = torch.randn(c, generator=g, dtype=torch.float64)
    rm, rv = torch.randn(c, generator=g, dtype=torch.float64), torch.rand(c, generator=g, dtype=torch.float64) + 0.5
    gy = torch.randn(n, c, generator=g, dtype=torch.float64)
    # reference: torch CPU fp64
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ro = r.clone().requires_grad_(True) if res else None
    rmo, rvo = rm.clone(), rv.clone()
    yo = torch.nn.functional.batch_norm(xo, rmo, rvo, wo, bo, training, 0.05, 1e-5)
    if res:
        yo = yo + ro
    if relu:
        yo = torch.relu(yo)
    yo.backward(gy)
    f = lambda t: t.float().to(DEV)
    xg, wg, bg = f(x).requires_grad_(True), f(w).requires_grad_(True), f(b).requires_grad_(True)
    rg = f(r).requires_grad_(True) if res else None
    rmg, rvg = f(rm), f(rv)
    y = batch_norm(xg, wg, bg, rmg, rvg, training, 0.05, 1e-5, rg, relu)
    y.backward(f(gy))
    assert rel_l2(y.detach().cpu(), yo.detach()) < 2e-6
    assert rel_l2(xg.grad.cpu(), xo.grad) < 1e-5
    assert rel_l2(wg.grad.cpu(), wo.grad) < 1e-5 and rel_l2(bg.grad.cpu(), bo.grad) < 1e-5
    if res:
        assert rel_l2(rg.grad.cpu(), ro.grad) < 2e-6
    if training and n > 1:
        assert rel_l2(rmg.cpu(), rmo) < 1e-6 and rel_l2(rvg.cpu(), rvo) < 1e-6


@pytest.mark.parametrize("c,n,res,relu,ld", [(128, 3000, False, False, 0), (128, 70001, True, True, 0), (256, 777, False, True, 0),
                                             (128, 5000, True, True, 256), (256, 129, True, False, 0)])
def test_batch_norm_plane_images_from_the_producer(c, n, res, relu, ld):
    """Bound mode (round 5, gcl_bn_stats_from_tiles_range / gcl_bn_apply_planes / gcl_bn_bwd_reduce_range /
    gcl_bn_bwd_apply_planes): from the convolution epilogue's column ranges the statistics launch bounds max|y| BEFORE the
    apply pass -- exactly the measured maximum without a residual, an upper bound (< 2 x + max|residual|) with one -- and
    the apply passes write the plane image themselves: bit for bit what gcl_split_planes makes of the same tensor at the
    same slot; y, mask, dx, dres unchanged by it.  ``ld``: the image of an ME.cat written in place (row pitch ld)."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, generator=g) * torch.rand(c, generator=g) * 3 + torch.randn(c, generator=g)).to(DEV)
    w, b = (torch.rand(c, generator=g) - 0.3).to(DEV), torch.randn(c, generator=g).to(DEV)      # some negative weights
    r = (torch.randn(n, c, generator=g) * 2).to(DEV) if res else None
    gy = (torch.randn(n, c, generator=g) * torch.rand(c, generator=g) * 1e-3).to(DEV)
    nt = (n + 127) // 128
    with torch.cuda.device(DEV):
        # the epilogue's partials, made here: per 128-row tile column sum, sum of squares, minimum, maximum
        pad = nt * 128 - n
        xp = torch.cat([x, torch.zeros(pad, c, device=DEV)]).view(nt, 128, c)
        big = torch.full((pad, c), 3.0e38, device=DEV)
        xmin = torch.cat([x, big]).view(nt, 128, c).amin(1)
        xmax = torch.cat([x, -big]).view(nt, 128, c).amax(1)
        part = torch.stack([xp.sum(1), (xp * xp).sum(1), xmin, xmax]).permute(0, 2, 1).contiguous()      # [4][c][nt]
        mean, rstd = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        xrange = torch.empty((2, c), device=DEV)
        slot, rslot = ME.ops.amax_slot(x.device), ME.ops.amax_slot(x.device)
        if res:
            _lib.check(lib.gcl_amax(_lib.ptr(r), r.numel(), _lib.ptr(rslot), 1, _lib.stream()), "gcl_amax")
        _lib.check(lib.gcl_bn_stats_from_tiles_range(_lib.ptr(part), nt, n, c, 1e-5, 0.05, None, None, _lib.ptr(mean), _lib.ptr(rstd),
                                                     _lib.ptr(xrange), _lib.ptr(w), _lib.ptr(b), int(relu),
                                                     _lib.ptr(rslot) if res else None, None, _lib.ptr(slot), _lib.stream()),
                   "gcl_bn_stats_from_tiles_range")
        assert torch.equal(xrange[0], x.amin(0)) and torch.equal(xrange[1], x.amax(0))
        bound = float(ME.ops.amax_value(slot))
        width = ld if ld else c
        outs = []
        for with_planes in (False, True):
            y = torch.full((n, width), 7.0, device=DEV)
            mask = torch.zeros(lib.gcl_bn_mask_len(n, c), dtype=torch.int64, device=DEV)
            s2 = slot.clone() if with_planes else ME.ops.amax_slot(x.device)
            img = torch.full((n, width), 0x7B7B7B7B, dtype=torch.int32, device=DEV) if with_planes else None
            _lib.check(lib.gcl_bn_apply_planes(_lib.ptr(x), n, c, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(w), _lib.ptr(b),
                                               _lib.ptr(r), int(relu), _lib.ptr(y), ld, _lib.ptr(mask) if relu else None,
                                               _lib.ptr(s2), _lib.ptr(img), _lib.stream()), "gcl_bn_apply_planes")
            outs.append((y, mask, s2, img))
        (y0, m0, s0, _), (y1, m1, s1, img) = outs
        assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(s1, slot), "the image changes nothing else"
        measured = float(y0[:, :c].abs().max())
        assert float(ME.ops.amax_value(s0)) == measured
        if res:
            assert measured <= bound <= 2.0 * measured + float(r.abs().max())
        else:
            assert bound == measured, "exact without a residual: max|y| is attained at a channel's extreme x"
        # the image == gcl_split_planes of y at the same slot (columns beyond c of a wider image are not touched)
        ref = torch.full((n, width), 0x7B7B7B7B, dtype=torch.int32, device=DEV)
        if ld:
            full = y1.clone()
            full[:, c:] = 0.0
            _lib.check(lib.gcl_split_planes(_lib.ptr(full), n, width, _lib.ptr(slot), _lib.ptr(ref), _lib.stream()), "split")
            sl = c // 32 * 32           # 32-column slices written by the pass: compare those
            assert torch.equal(img.view(n, width // 32, 32)[:, :c // 32], ref.view(n, width // 32, 32)[:, :c // 32])
            assert bool((img.view(n, width // 32, 32)[:, c // 32:] == 0x7B7B7B7B).all())
        else:
            _lib.check(lib.gcl_split_planes(_lib.ptr(y1), n, c, _lib.ptr(slot), _lib.ptr(ref), _lib.stream()), "split")
            assert torch.equal(img, ref)
        # backward: bound of max|dx| from the reduce launch, image from the apply pass
        sums = torch.empty((2, c), device=DEV)
        scratch = torch.empty(lib.gcl_bn_scratch_len(n, c), dtype=torch.float64, device=DEV)
        dslot = ME.ops.amax_slot(x.device)
        _lib.check(lib.gcl_bn_bwd_reduce_range(_lib.ptr(x), _lib.ptr(gy), 0, None, _lib.ptr(m0) if relu else None, n, c,
                                               _lib.ptr(mean), _lib.ptr(rstd), int(relu), _lib.ptr(scratch), _lib.ptr(sums[0]),
                                               _lib.ptr(sums[1]), _lib.ptr(xrange), _lib.ptr(w), _lib.ptr(dslot), _lib.stream()),
                   "gcl_bn_bwd_reduce_range")
        sums0 = torch.empty((2, c), device=DEV)
        _lib.check(lib.gcl_bn_bwd_reduce(_lib.ptr(x), _lib.ptr(gy), None, _lib.ptr(m0) if relu else None, n, c, _lib.ptr(mean),
                                         _lib.ptr(rstd), int(relu), _lib.ptr(scratch), _lib.ptr(sums0[0]), _lib.ptr(sums0[1]),
                                         _lib.stream()), "gcl_bn_bwd_reduce")
        assert torch.equal(sums, sums0)
        bouts = []
        for with_planes in (False, True):
            dx, dres = torch.empty_like(x), (torch.empty_like(x) if res else None)
            s2 = dslot.clone() if with_planes else ME.ops.amax_slot(x.device)
            img = torch.empty((n, c), dtype=torch.int32, device=DEV) if with_planes else None
            _lib.check(lib.gcl_bn_bwd_apply_planes(_lib.ptr(x), _lib.ptr(gy), 0, None, _lib.ptr(m0) if relu else None, n, c,
                                                   _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(w), _lib.ptr(sums[0]),
                                                   _lib.ptr(sums[1]), int(relu), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(s2),
                                                   _lib.ptr(img), _lib.stream()), "gcl_bn_bwd_apply_planes")
            bouts.append((dx, dres, s2, img))
        (dx0, dr0, t0, _), (dx1, dr1, t1, dimg) = bouts
        assert torch.equal(dx0, dx1) and (not res or torch.equal(dr0, dr1)) and torch.equal(t1, dslot)
        dmeasured, dbound = float(dx0.abs().max()), float(ME.ops.amax_value(dslot))
        assert float(ME.ops.amax_value(t0)) == dmeasured
        assert dmeasured <= dbound <= 4.0 * dmeasured, (dmeasured, dbound)
        dref = torch.empty((n, c), dtype=torch.int32, device=DEV)
        _lib.check(lib.gcl_split_planes(_lib.ptr(dx1), n, c, _lib.ptr(dslot), _lib.ptr(dref), _lib.stream()), "split")
        assert torch.equal(dimg, dref)


@pytest.mark.parametrize("c,n,ld,relu", [(64, 1000, 96, True), (32, 257, 160, False), (128, 70000, 256, True)])
def test_batch_norm_apply_into_a_column_slice(c, n, ld, relu):
    """gcl_bn_apply_ld (the decoder input of an ME.cat written straight into the cat's output): the slice holds bit for bit
    what gcl_bn_apply writes to a tensor of its own, the other columns are untouched, mask and max|y| are the same."""
    from gcl_amd import _lib
    import gcl_amd.MinkowskiEngine as ME
    lib = _lib.load()
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, generator=g) * 2 + 1).to(DEV)
    mean, rstd = torch.randn(c, generator=g).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
    w, b = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    res = torch.randn(n, c, generator=g).to(DEV)
    with torch.cuda.device(DEV):
        outs = []
        for y_ld in (0, ld):
            y = torch.full((n, y_ld if y_ld else c), 7.0, device=DEV)
            mask = torch.zeros(lib.gcl_bn_mask_len(n, c), dtype=torch.int64, device=DEV)
            slot = ME.ops.amax_slot(x.device)
            _lib.check(lib.gcl_bn_apply_ld(_lib.ptr(x), n, c, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(w), _lib.ptr(b),
                                           _lib.ptr(res), int(relu), _lib.ptr(y), y_ld, _lib.ptr(mask) if relu else None,
                                           _lib.ptr(slot), _lib.stream()), "gcl_bn_apply_ld")
            outs.append((y, mask, ME.ops.amax_value(slot)))
        (y0, m0, a0), (y1, m1, a1) = outs
        assert torch.equal(y1[:, :c], y0) and bool((y1[:, c:] == 7.0).all())
        assert torch.equal(m0, m1) and torch.equal(a0, a1)


# ---------------------------------------------------------------------------------------------------------------
# the whole network: ResUNetBN2C forward + backward vs the oracle (configs[0]-sized and a LiDAR-shaped cloud)
# ---------------------------------------------------------------------------------------------------------------
def _model_and_state(seed, conv1_kernel_size=5, out=32):
    from gcl_amd.model import load_model
    torch.manual_seed(seed)
    m = load_model("ResUNetBN2C")(1, out, bn_momentum=0.05, normalize_feature=True,
                                  conv1_kernel_size=conv1_kernel_size, D=3).to(DEV)
    with torch.no_grad():                      # non-trivial BN affine parameters
        for name, p in m.named_parameters():
            if name.endswith("bn.weight"):
                p.uniform_(0.5, 1.5)
            elif name.endswith("bn.bias"):
                p.uniform_(-0.1, 0.1)
    st = {k: v.detach().cpu().double().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
    return m, st


@pytest.mark.parametrize("kind", ["boxes5k", "lidar"])
def test_resunet_forward_backward_vs_oracle(kind, precision):
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    if kind == "boxes5k":      # BASELINE configs[0]: one 5k-point cloud, voxel 0.3 (demo plumbing case)
        xyz = synthetic.make_box_cloud(0, 5000)
        k1 = 3                 # demo.py:29 uses conv1_kernel_size=3
    else:
        xyz = synthetic.raycast(synthetic.make_scene(1, n_boxes=20), np.zeros(3), 7)[::3]
        k1 = 5
    coords, inds = ME.utils.sparse_quantize(xyz / 0.3, return_index=True)
    C = ME.utils.batched_coordinates([coords])
    feats = torch.ones(len(C), 1)
    m, st = _model_and_state(0, k1)
    m.train()
    out = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV)))
    assert np.array_equal(out.C.cpu().numpy(), C.numpy()), "row order of the stride-1 map must be preserved"
    F = out.F
    so = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    Fo = O.resunet_forward(so, C.numpy(), feats.double(), k1, True, True, 0.05)
    err = rel_l2(F.detach().cpu(), Fo.detach())
    print(f"[{kind}/{precision}] N={len(C)} feature rel-L2 vs fp64 oracle: {err:.3e}")
    from conftest import precision_log_path
    with open(precision_log_path(), "a") as fh:
        fh.write(f"{kind} {precision} N={len(C)} feature_rel_l2={err:.4e}\n")
    assert err < 1e-4            # north-star tolerance
    g = torch.Generator().manual_seed(1)
    gy = torch.randn(Fo.shape, generator=g, dtype=torch.float64)
    Fo.backward(gy)
    F.backward(gy.float().to(DEV))
    worst = 0.0
    for name, p in m.named_parameters():
        e = rel_l2(p.grad.cpu(), so[name].grad)
        worst = max(worst, e)
        # BatchNorm-parameter gradients of the deep levels (few rows) amplify fp32 rounding: 1e-2; kernels 2e-3
        assert e < (3e-2 if precision == "bf16x3" else (1e-2 if ".bn." in name else 2e-3)), (name, e)
    print(f"[{kind}] worst parameter-gradient rel-L2: {worst:.3e}")
    # BN running statistics were updated like BatchNorm1d's
    for name, b in m.named_buffers():
        if "running" in name:
            assert rel_l2(b.cpu(), so[name]) < 1e-4, name
    # eval mode uses the running statistics
    m.eval()
    with torch.no_grad():
        Fe = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F
    so2 = {k: v.detach() for k, v in so.items()}
    Feo = O.resunet_forward(so2, C.numpy(), feats.double(), k1, True, False, 0.05)
    assert rel_l2(Fe.cpu(), Feo) < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# loss / kNN against the golden vectors captured from the reference's own code
# ---------------------------------------------------------------------------------------------------------------
def _loss_case(z):
    """(draws, switches) recorded in a finest_loss_*.npz fixture (tests/golden/make_golden.py)."""
    sw = {k: bool(z[k]) for k in ("square_loss", "block_finest_gradient", "use_pair_group_positive_loss",
                                  "finest_term", "use_hard_negative") if k in z.files}
    draws = (z["pos_sel"], z["sel_hn1"], z["sel_hn2"], z["pair_pos"] if "pair_pos" in z.files else None)
    if "random_cols" in z.files:          # use_hard_negative == False: the drawn columns (:514)
        draws = draws + (z["random_cols"],)
    return draws, sw


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "finest_loss_*.npz"))))
def test_finest_contrastive_loss_golden(path):
    """The reference's own outputs (all config switches of finest_contrastive_loss, and location_contrastive_loss):
    loss triple within 2e-6 abs, dL/dF within 1e-5 rel-L2."""
    from gcl_amd.lib.colocation_trainer import finest_contrastive_loss
    z = np.load(path)
    draws, sw = _loss_case(z)
    F = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
    kw = dict(max_pos_cluster=int(z["max_pos_cluster"]), max_hn_samples=int(z["max_hn_samples"]), **sw)
    pos, fin, neg = finest_contrastive_loss(F, torch.from_numpy(z["group"]), torch.from_numpy(z["index"]),
                                            z["index_hash"], torch.from_numpy(z["finest_flag"]),
                                            draws=draws, **kw)
    assert abs(pos.item() - float(z["pos"])) < 2e-6 and abs(fin.item() - float(z["finest"])) < 2e-6
    if np.isnan(float(z["neg"])):
        assert np.isnan(neg.item())          # all hardest negatives were self matches: mean of empty
        (pos + fin).backward()
    else:
        assert abs(neg.item() - float(z["neg"])) < 2e-6
        (pos + fin + neg).backward()
        assert rel_l2(F.grad.cpu(), z["grad"]) < 1e-5
    # the trainer's one-node total (``total_weights``): same terms, total = the weighted sum, dL/dF = that of the sum
    if sw.get("use_hard_negative", True) and not np.isnan(float(z["neg"])):
        w = (0.7, 1.3, 0.9)
        F2 = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
        tot, p1, f1, n1 = finest_contrastive_loss(F2, torch.from_numpy(z["group"]), torch.from_numpy(z["index"]),
                                                  z["index_hash"], torch.from_numpy(z["finest_flag"]), draws=draws,
                                                  total_weights=w, **kw)
        assert not (p1.requires_grad or f1.requires_grad or n1.requires_grad) and tot.requires_grad
        assert abs(p1.item() - pos.item()) < 1e-6 and abs(f1.item() - fin.item()) < 1e-6 and n1.item() == neg.item()
        assert abs(tot.item() - (w[0] * pos.item() + w[1] * fin.item() + w[2] * neg.item())) < 2e-6
        (2.0 * tot).backward()
        F3 = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
        p3, f3, n3 = finest_contrastive_loss(F3, torch.from_numpy(z["group"]), torch.from_numpy(z["index"]),
                                             z["index_hash"], torch.from_numpy(z["finest_flag"]), draws=draws, **kw)
        (2.0 * (w[0] * p3 + w[1] * f3 + w[2] * n3)).backward()
        assert rel_l2(F2.grad.cpu(), F3.grad.cpu()) < 1e-6
        # ... and both equal the oracle's autograd with UNEQUAL weights (pos_weight != finest_weight used to read the
        # finest term's upstream gradient for both terms: two temporaries shared one address)
        Fo = torch.from_numpy(z["F_out"]).double().requires_grad_(True)
        po, fo, no = LO.finest_contrastive_loss(Fo, z["group"], z["index"], z["index_hash"], z["finest_flag"],
                                                draws=draws, **kw)
        (2.0 * (w[0] * po + w[1] * fo + w[2] * no)).backward()
        assert rel_l2(F3.grad.cpu(), Fo.grad) < 1e-5 and rel_l2(F2.grad.cpu(), Fo.grad) < 1e-5
    # drawing on the host from a seeded np.random reproduces the reference's selections
    np.random.seed(int(z["np_seed"]))
    p2, f2, n2 = finest_contrastive_loss(F.detach(), torch.from_numpy(z["group"]), torch.from_numpy(z["index"]),
                                         z["index_hash"], torch.from_numpy(z["finest_flag"]), **kw)
    assert abs(p2.item() - float(z["pos"])) < 2e-6 and abs(f2.item() - float(z["finest"])) < 2e-6
    if not np.isnan(float(z["neg"])):
        assert abs(n2.item() - float(z["neg"])) < 2e-6


def test_pdist_golden():
    from gcl_amd.lib.metrics import pdist, pdist_min
    z = np.load(os.path.join(G, "pdist.npz"))
    A, B = torch.from_numpy(z["A"]).to(DEV), torch.from_numpy(z["B"]).to(DEV)
    assert np.allclose(pdist(A[:96], B[:96], "L2").cpu().numpy(), z["L2_sub"], rtol=0, atol=2e-6)
    assert np.allclose(pdist(A[:96], B[:96], "SquareL2").cpu().numpy(), z["Sq_sub"], rtol=0, atol=2e-6)
    d, i = pdist_min(A, B, "L2")
    assert np.allclose(d.cpu().numpy(), z["L2_rowmin"], rtol=0, atol=2e-6)
    assert np.array_equal(i.cpu().numpy(), z["L2_rowarg"])


@pytest.mark.parametrize("nn_max_n", [-1, 500, 2000])
def test_find_nn_gpu_golden(nn_max_n):
    from gcl_amd.lib.eval import find_nn_gpu
    z = np.load(os.path.join(G, "find_nn.npz"))
    F0, F1 = torch.from_numpy(z["F0"]).to(DEV), torch.from_numpy(z["F1"]).to(DEV)
    idx, dist = find_nn_gpu(F0, F1, nn_max_n=nn_max_n, return_distance=True)
    assert idx.dtype == torch.int64 and not idx.is_cuda and dist.shape == (5000, 1)
    ref_i, ref_d = z[f"idx_{nn_max_n}"], z[f"dist_{nn_max_n}"]
    assert np.allclose(dist.numpy()[:, 0], ref_d, rtol=0, atol=2e-6)
    bad = np.nonzero(idx.numpy() != ref_i)[0]
    # an index may differ only where two candidates are tied to within fp32 rounding
    for r in bad:
        d_ref = float(((z["F0"][r] - z["F1"][ref_i[r]]) ** 2).sum())
        d_got = float(((z["F0"][r] - z["F1"][idx[r]]) ** 2).sum())
        assert abs(d_ref - d_got) < 1e-6
    assert len(bad) <= 2


def test_find_nn_repeated_calls_reuse_the_scratch_block():
    """The 1-NN search reads its pair-interleaved copy of B with SCALAR loads (csrc/loss.hip k_nn_rowmin): successive calls on
    DIFFERENT inputs land on the same scratch block of the caching allocator -- every call must see its own copy (scalar cache
    and L2 are made coherent at the kernel boundary), rows_a / rows_b selections included."""
    from gcl_amd.lib.metrics import pdist_min
    g = torch.Generator().manual_seed(11)
    for it in range(6):
        A = torch.nn.functional.normalize(torch.randn(700, 32, generator=g), dim=1)
        B = torch.nn.functional.normalize(torch.randn(901, 32, generator=g), dim=1)
        ra = torch.randperm(700, generator=g)[:333] if it % 2 else None
        rb = torch.randperm(901, generator=g)[:555] if it % 3 == 0 else None
        d, i = pdist_min(A.to(DEV), B.to(DEV), "SquareL2", rows_a=None if ra is None else ra.to(DEV),
                         rows_b=None if rb is None else rb.to(DEV))
        Ar, Br = (A if ra is None else A[ra]).double(), (B if rb is None else B[rb]).double()
        D = ((Ar[:, None, :] - Br[None]) ** 2).sum(-1)
        ref_d, ref_i = D.min(1)
        assert torch.equal(i.cpu().long(), ref_i), it
        assert float((d.cpu().double() - ref_d).abs().max()) < 2e-6


def test_find_nn_ragged_and_ties():
    from gcl_amd.lib.eval import find_nn_gpu
    g = torch.Generator().manual_seed(0)
    for ma, mb, c in [(1, 1, 32), (65, 3, 16), (130, 257, 64), (7, 1000, 32)]:
        A, B = torch.randn(ma, c, generator=g), torch.randn(mb, c, generator=g)
        ref = LO.find_nn(A, B)
        got = find_nn_gpu(A.to(DEV), B.to(DEV))
        assert np.array_equal(got.numpy(), ref.numpy())
    A = torch.zeros(5, 32)
    B = torch.zeros(300, 32)          # all tied: lowest index wins
    assert (find_nn_gpu(A.to(DEV), B.to(DEV)) == 0).all()


# ---------------------------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE-sized batch; the oracle would take minutes here)
# ---------------------------------------------------------------------------------------------------------------
def test_full_size_properties():
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    batch = synthetic.collate_train([synthetic.make_train_sample(11, num_neighborhood=6)])   # 7 clouds, ~1e5 voxels
    C = batch["sinput_C"].to(DEV)
    n = len(C)
    mgr = ME.CoordinateManager(C)
    km = mgr.get_kernel_map(1, 3, 1)
    assert km.counts[13] == n, "centre offset maps every voxel to itself"
    assert km.counts == km.counts[::-1], "offset k and its mirror K-1-k have the same number of pairs"
    assert (km.nbr[13].cpu() == torch.arange(n, dtype=torch.int32)).all()
    km2 = mgr.get_kernel_map(1, 3, 2)
    assert sum(km2.counts) >= n and (km2.nbr_t >= 0).sum().item() == sum(km2.counts)
    # every fine voxel has exactly one parent among the 27 offsets whose (even-aligned) cell contains it? No: at
    # least one -- its own cell -- so every column of nbr_t has a hit
    assert ((km2.nbr_t >= 0).sum(0) >= 1).all()
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(32, 64, kernel_size=3, stride=1, dimension=3).to(DEV)
    ones = torch.ones(n, 32, device=DEV)
    y = conv(ME.SparseTensor(ones, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
    # checksum of checksums: sum_v y[v] = sum_k count_k * colsum(W_k)
    cnt = torch.tensor(km.counts, dtype=torch.float64)
    ref = (cnt[:, None] * conv.kernel.detach().cpu().double().sum(1)).sum(0)
    assert rel_l2(y.detach().double().sum(0).cpu(), ref) < 1e-5
    # linearity + adjointness <conv(x), g> = <x, conv^T(g)> through the input-gradient kernel
    x = torch.randn(n, 32, device=DEV, requires_grad=True)
    gy = torch.randn(n, 64, device=DEV)
    y1 = conv(ME.SparseTensor(x, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
    y1.backward(gy)
    lhs = (y1.detach().double() * gy.double()).sum().item()
    rhs = (x.detach().double() * x.grad.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * abs(lhs) + 1e-3
    # dW by linearity in W: <dW, W> = <y, gy>
    dw_dot = (conv.kernel.grad.double() * conv.kernel.detach().double()).sum().item()
    assert abs(dw_dot - lhs) < 1e-5 * abs(lhs) + 1e-3
    # determinism: bitwise identical on a second run
    conv.kernel.grad = None
    x2 = x.detach().clone().requires_grad_(True)
    y2 = conv(ME.SparseTensor(x2, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
    y2.backward(gy)
    assert torch.equal(y1.detach(), y2.detach()) and torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("window", [0, 2048, 4096])
def test_table_sort_is_a_stable_mask_sort(window):
    """gcl_table_sort: order is a permutation sorted by presence mask (stable; inside windows of consecutive rows when
    a window is given), the permuted table and the tile masks are consistent with it."""
    from gcl_amd.MinkowskiEngine import core
    C = random_cloud(9, n=5000, batch=2)
    old = core.SORT_WINDOW
    core.SORT_WINDOW = window
    try:
        mgr = make_mgr(C)
        for key in [(1, 3, 1), (1, 3, 2)]:
            km = mgr.get_kernel_map(*key)
            for transposed in ([False, True] if km.nbr_t is not None else [False]):
                tbl = (km.nbr_t if transposed else km.nbr).cpu().numpy()
                ts, order, tmask = (t.cpu().numpy() for t in km.sorted_table(transposed))
                n = tbl.shape[1]
                mask = np.zeros(n, dtype=np.int64)
                for k in range(km.K):
                    mask |= (tbl[k] >= 0).astype(np.int64) << k
                win = np.arange(n) // window if window else np.zeros(n, dtype=np.int64)
                # the global sort's key: mask bits re-ordered by offset frequency in this table (rarest offset = most
                # significant bit, ties: lower offset lower), at most three 8-bit radix passes = its 24 top bits
                if window:
                    key = mask
                else:
                    cnt = np.array([int(((mask >> k) & 1).sum()) for k in range(km.K)])
                    pos = np.array([int(((cnt > cnt[k]) | ((cnt == cnt[k]) & (np.arange(km.K) < k))).sum())
                                    for k in range(km.K)])
                    key = np.zeros_like(mask)
                    for k in range(km.K):
                        key |= ((mask >> k) & 1) << pos[k]
                    key = key >> max(0, km.K - 24)
                ref_order = np.lexsort((np.arange(n), key, win))
                assert np.array_equal(order, ref_order)
                assert np.array_equal(ts, tbl[:, order])
                pad = (-n) % 32
                mt = np.concatenate([mask[order], np.zeros(pad, np.int64)]).reshape(-1, 32)
                assert np.array_equal(tmask.astype(np.int64), np.bitwise_or.reduce(mt, axis=1))
    finally:
        core.SORT_WINDOW = old


def test_extract_features_and_eval_pair_vs_oracle():
    """util/misc.extract_features (:58-130) and the eval-loop body (scripts/test_kitti.py:141-161): voxelise ->
    SparseTensor -> model(eval) -> F; find_corr with seeded subsampling returns the oracle's correspondences."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.lib.eval import find_corr
    from gcl_amd.util.misc import extract_features
    m, st = _model_and_state(3, 5)
    xyz = synthetic.raycast(synthetic.make_scene(2, n_boxes=15), np.zeros(3), 3)[::4]
    ret_xyz, F = extract_features(m, xyz, voxel_size=0.3, device=torch.device(DEV), skip_check=True)
    coords, inds = ME.utils.sparse_quantize(np.floor(xyz / 0.3), return_index=True)
    assert np.array_equal(ret_xyz, xyz[inds]) and F.shape == (len(inds), 32)
    C = ME.utils.batched_coordinates([coords]).numpy()
    Fo = O.resunet_forward(st, C, torch.ones(len(C), 1, dtype=torch.float64), 5, True, False, 0.05)
    assert rel_l2(F.detach().cpu(), Fo) < 1e-4
    # a second, shifted view of the same scene: correspondences through feature 1-NN
    xyz1 = synthetic.raycast(synthetic.make_scene(2, n_boxes=15), np.array([4.0, 0, 0]), 4)[::4]
    ret1, F1 = extract_features(m, xyz1, voxel_size=0.3, device=torch.device(DEV), skip_check=True)
    np.random.seed(5)
    a0, a1 = find_corr(ret_xyz, ret1, F.detach(), F1.detach(), subsample_size=1500)
    np.random.seed(5)
    i0 = np.random.choice(len(F), 1500, replace=False)
    i1 = np.random.choice(len(F1), 1500, replace=False)
    nn = LO.find_nn(F.detach().cpu()[i0], F1.detach().cpu()[i1], nn_max_n=500)
    assert np.array_equal(a0, ret_xyz[i0])
    same = (a1 == ret1[i1[nn.numpy()]]).all(axis=1)
    assert same.mean() > 0.995          # an index may differ only on fp32 near-ties of the distance


def test_train_step_matches_oracle_and_flat_ddp_is_transparent():
    """One optimizer step (a10): loss triple vs the oracle, and the flat-buffer DDP wrapper (world size 1) leaves the
    update unchanged (up to the last-bit order dependence of the loss backward's float atomics)."""
    from gcl_amd import ddp, synthetic
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    batch = synthetic.collate_train([synthetic.make_train_sample(21, num_neighborhood=2, n_boxes=10)])
    N, G = len(batch["sinput_C"]), len(batch["group"])
    rng = np.random.RandomState(0)
    draws = (rng.choice(G, min(G, 64), replace=False), rng.choice(N, 256, replace=False), rng.choice(N, 256, replace=False))
    cfg = make_config(batch_size=1, num_pos_per_batch=64, num_hn_samples_per_batch=256)
    outs = []
    for use_ddp in (False, True):
        torch.manual_seed(11)
        tr = FinestContrastiveLossTrainer(cfg, device=DEV, ddp=ddp.FlatDDP() if use_ddp else None)
        st0 = {k: v.detach().cpu().double().clone() for k, v in tr.model.state_dict().items() if "num_batches" not in k}
        loss, parts, n = tr.train_step(batch, draws=draws)
        outs.append((loss.item(), [p.item() for p in parts],
                     torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()))
    assert outs[0][0] == outs[1][0] and torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-6)
    Fo = O.resunet_forward(st0, batch["sinput_C"].numpy(), batch["sinput_F"].double(), 5, True, True, 0.05)
    ref = LO.finest_contrastive_loss(Fo, batch["group"].numpy(), batch["index"].numpy(), batch["index_hash"],
                                     batch["finest_flag"].numpy(), draws=draws)
    assert np.allclose(outs[0][1], [r.item() for r in ref], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["heavy_tail", "tiny", "huge", "zero"])
def test_fp16x3_dynamic_range(kind):
    """The default arithmetic scales every operand tensor by a power of two derived from its max-abs before splitting
    it into two fp16 planes: inputs with a wide dynamic range / extreme magnitudes must stay at fp32-level accuracy
    (error measured against the fp64 oracle, relative to the OUTPUT norm)."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import ops
    assert ops.PRECISION == "fp16x3"
    C = random_cloud(13, n=2000, batch=1)
    mgr, omgr = make_mgr(C), O.CoordinateManager(C)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(len(C), 64, generator=g, dtype=torch.float64)
    if kind == "heavy_tail":
        x = x * torch.exp(3.0 * torch.randn(len(C), 64, generator=g, dtype=torch.float64))     # ~ 8 decades
    elif kind == "tiny":
        x = x * 1e-30
    elif kind == "huge":
        x = x * 1e25
    elif kind == "zero":
        x = x * 0
    conv = ME.MinkowskiConvolution(64, 64, kernel_size=3, stride=1, dimension=3).to(DEV)
    xs = x.float().to(DEV).requires_grad_(True)
    y = conv(ME.SparseTensor(xs, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=mgr)).F
    W = conv.kernel.detach().cpu().double().requires_grad_(True)
    xo = x.float().double().requires_grad_(True)
    yo = O.sparse_conv(xo, W, omgr.get_kernel_map(1, 3, 1), len(C))
    gy = torch.randn(yo.shape, generator=g, dtype=torch.float64) * (1e-12 if kind != "zero" else 1.0)
    yo.backward(gy)
    y.backward(gy.float().to(DEV))
    assert torch.isfinite(y).all() and torch.isfinite(xs.grad).all() and torch.isfinite(conv.kernel.grad).all()
    if kind == "zero":
        assert (y == 0).all() and (conv.kernel.grad == 0).all()
        assert rel_l2(xs.grad.cpu(), xo.grad) < 2e-6
        return
    assert rel_l2(y.detach().cpu(), yo.detach()) < 2e-6
    assert rel_l2(xs.grad.cpu(), xo.grad) < 2e-6
    assert rel_l2(conv.kernel.grad.cpu(), W.grad) < 2e-6


def test_fp16x3_small_rows_elementwise():
    """The fp16x3 caveat as a number: the two fp16 planes of an operand are cut relative to
    the TENSOR maximum, so rows whose magnitude is 2^-20 of it keep only the bits above 2^-25 of the scaled range (the
    lo plane is subnormal there).  Every 16th input row is scaled by 2^-20 and only feeds outputs through its own
    centre offset weights -- reported per output row: error relative to that row's own norm next to the tensor-wide
    rel-L2.  Measured on MI355X: tensor rel-L2 1.1e-7, full-scale rows <= 1.9e-7, rows at 2^-20 of the maximum 1.4e-5 of
    THEIR OWN norm (fp32 arithmetic would give ~1e-7 there: this is the caveat, as a number), max |error| / max |y| 1.8e-7.
    Bounds asserted: 2e-6 tensor-wide and on full-scale rows, 1e-4 on the small rows, 2^-21 of the output maximum
    absolutely."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import ops
    assert ops.PRECISION == "fp16x3"
    n = 4096
    C = torch.zeros((n, 4), dtype=torch.int32)
    C[:, 1] = torch.arange(n) * 4                # isolated voxels: every output row sees only its own input row
    mgr, omgr = make_mgr(C.numpy()), O.CoordinateManager(C.numpy())
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 64, generator=g, dtype=torch.float64)
    small = torch.arange(n) % 16 == 0
    x[small] *= 2.0 ** -20
    conv = ME.MinkowskiConvolution(64, 64, kernel_size=3, stride=1, dimension=3).to(DEV)
    with torch.no_grad():
        y = conv(ME.SparseTensor(x.float().to(DEV), coordinate_map_key=ME.CoordinateMapKey(1),
                                 coordinate_manager=mgr)).F.cpu().double()
    yo = O.sparse_conv(x.float().double(), conv.kernel.detach().cpu().double(), omgr.get_kernel_map(1, 3, 1), n)
    err = (y - yo).norm(dim=1)
    row_rel = err / yo.norm(dim=1)
    total = rel_l2(y, yo)
    worst_small, worst_big = float(row_rel[small].max()), float(row_rel[~small].max())
    abs_vs_max = float((y - yo).abs().max() / yo.abs().max())
    print(f"fp16x3 element-wise: tensor rel-L2 {total:.2e}; rows at 2^-20 of the maximum: worst per-row relative error "
          f"{worst_small:.2e}; full-scale rows: {worst_big:.2e}; max |error| / max |y| = {abs_vs_max:.2e}")
    assert total < 2e-6 and worst_big < 2e-6
    assert worst_small < 1e-4
    assert abs_vs_max < 2.0 ** -21


def test_resunet_fat_variant_vs_oracle():
    """ResUNetFatBN (the reference script's default model, scripts/train_gcl_kitti.sh:13; model/resunet.py:263-266)
    runs through the same kernels: wider decoder (TR_CHANNELS 128,128,128,256), concat widths 160 / 192 / 384."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.model import load_model
    xyz = synthetic.raycast(synthetic.make_scene(4, n_boxes=12), np.zeros(3), 9)[::6]
    coords, _ = ME.utils.sparse_quantize(xyz / 0.3, return_index=True)
    C = ME.utils.batched_coordinates([coords])
    torch.manual_seed(2)
    m = load_model("ResUNetFatBN")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(DEV)
    st = {k: v.detach().cpu().double().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
    m.train()
    F = m(ME.SparseTensor(torch.ones(len(C), 1).to(DEV), coordinates=C.to(DEV))).F
    so = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    saved = (O.CHANNELS, O.TR_CHANNELS)
    Fo = O.resunet_forward(so, C.numpy(), torch.ones(len(C), 1, dtype=torch.float64), 5, True, True, 0.05)
    assert rel_l2(F.detach().cpu(), Fo.detach()) < 1e-4
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(Fo.shape, generator=g, dtype=torch.float64)
    Fo.backward(gy)
    F.backward(gy.float().to(DEV))
    # This case is ill-conditioned in fp32: the ORACLE ITSELF run in torch-CPU float32 differs from its float64 run by
    # 4.6e-3 ... 6.5e-3 rel-L2 on every encoder kernel / BatchNorm gradient (2e-3 in the decoder; measured with the
    # same state dict, cloud and output gradient) -- a small cloud whose BatchNorm backward cancels large sums.  The
    # HIP path lands at 1e-3 ... 8e-3; the bound is 3x the fp32 oracle's own distance.
    for name, p_ in m.named_parameters():
        tol = 2e-2
        err = rel_l2(p_.grad.cpu(), so[name].grad)
        assert err < tol, (name, err)


# ---------------------------------------------------------------------------------------------------------------
# "next" rows: voxelisation and co-location group building on the device -- bit-exact
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("voxel", [0.3, 0.025])
def test_sparse_quantize_gpu_bit_exact(voxel):
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_data_gpu import sparse_quantize_gpu
    xyz = synthetic.raycast(synthetic.make_scene(5, n_boxes=15), np.zeros(3), 1)
    if voxel < 0.1:
        xyz = synthetic.make_box_cloud(1, 5000)            # demo.py's 2.5 cm voxels
    xyz[:50] = xyz[50:100]                                  # exact duplicates
    ref_c, ref_i = ME.utils.sparse_quantize(xyz / np.float32(voxel), return_index=True)
    c, i = sparse_quantize_gpu(torch.from_numpy(xyz).to(DEV), voxel, batch_id=3)
    assert np.array_equal(i.cpu().numpy(), ref_i)
    assert np.array_equal(c.cpu().numpy()[:, 1:], ref_c) and (c[:, 0] == 3).all()


@pytest.mark.parametrize("seed,nn", [(31, 6), (32, 2)])
def test_colocation_groups_gpu_bit_exact(seed, nn):
    """Group sizes, member rows (nearest first, centre cloud first) and finest flags equal the CPU restatement of
    get_matching_indices_colocation (util/pointcloud.py:69-132) on a rotated + scaled synthetic sample."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_data_gpu import colocation_groups_gpu
    xyz_th, xyz_cmpl_th, coords, feats, group, index, finest, list_M = synthetic.make_train_sample(
        seed, 0.3, num_neighborhood=nn, n_boxes=20)
    # the generator scales the search radius 1.5 * 0.3 with the random scale it draws: replay it
    radius = synthetic.sample_search_radius(seed, 0.3, nn)
    xyz_own = np.concatenate([xyz_th] + xyz_cmpl_th).astype(np.float32)
    xyz_cf = np.concatenate([xyz_th] + [synthetic._apply(list_M[j], x) for j, x in enumerate(xyz_cmpl_th)])
    C = np.concatenate([np.concatenate([np.full((len(c), 1), b, np.int32), c], 1) for b, c in enumerate(coords)])
    g, idx, fl = colocation_groups_gpu(torch.from_numpy(xyz_own).to(DEV), torch.from_numpy(xyz_cf).to(DEV),
                                       torch.from_numpy(C).to(DEV), len(xyz_th), 1 + nn, list_M, 0.3, radius)
    assert np.array_equal(g.cpu().numpy(), np.asarray(group, dtype=np.int32))
    assert np.array_equal(idx.cpu().numpy(), np.asarray(index, dtype=np.int64))
    assert np.array_equal(fl.cpu().numpy(), np.asarray(finest, dtype=bool))
    assert len(group) > 100
    # ... and the per-point restatement of the reference's loop under oracle/ (brute-force radius searches), on the same
    # float32 centre-frame points the device builder receives
    from oracle.colocation_oracle import colocation_groups as oracle_groups
    og, oi, of = oracle_groups(xyz_th, xyz_cmpl_th, list_M, radius, K=5,
                               nghb_cf=[synthetic._apply(list_M[j], x) for j, x in enumerate(xyz_cmpl_th)])
    assert np.array_equal(g.cpu().numpy(), np.asarray(og, dtype=np.int32))
    assert np.array_equal(idx.cpu().numpy(), np.asarray(oi, dtype=np.int64))
    assert np.array_equal(fl.cpu().numpy(), np.asarray(of, dtype=bool))


def test_gpu_built_batch_trains():
    """End to end on the device: raw clouds -> voxelise -> groups -> collate -> training step (finite loss)."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_data_gpu import build_sample_gpu, collate_gpu
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    samples = []
    for s in (41, 42):
        scene = synthetic.make_scene(s, n_boxes=15)
        clouds = [synthetic.raycast(scene, np.array([x, 0.0, 0.0]), s * 7 + k)[::3] for k, x in enumerate((0.0, 6.0, 12.0))]
        Ms = [np.eye(4), np.eye(4)]
        Ms[0][0, 3], Ms[1][0, 3] = 6.0, 12.0
        samples.append(build_sample_gpu(clouds, Ms, 0.3, 0.45, DEV))
    batch = collate_gpu(samples)
    assert batch["group"].numel() > 50 and batch["finest_flag"].sum().item() == batch["group"].numel()
    tr = FinestContrastiveLossTrainer(make_config(batch_size=2, num_pos_per_batch=64, num_hn_samples_per_batch=128), device=DEV)
    np.random.seed(0)
    loss, parts, n = tr.train_step(batch)
    assert torch.isfinite(loss).item() and n == len(batch["sinput_C"])


@pytest.mark.parametrize("seeds,nn", [((61, 62), 2), ((71, 72, 73), 3)])
def test_build_batch_gpu_equals_the_cpu_loader(seeds, nn):
    """build_batch_gpu (round 6: every cloud of every sample through ONE coordinate table, the neighbours' centre-frame
    points computed on the device, two host reads per batch) against the CPU loader the synthetic generator restates
    (ColocationKittiDataset.__getitem__ + collate_colocation_fn, lib/colocation_data_loader.py:315-475): coordinates, features
    (the centre clouds' jitter), groups, member rows with the samples' offsets and finest flags, bit for bit; and against the
    per-point oracle on the device's own centre-frame points."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_data_gpu import build_batch_gpu
    raws = [synthetic.make_raw_sample(s, 0.3, num_neighborhood=nn, n_boxes=20) for s in seeds]
    want = synthetic.collate_train([synthetic.make_train_sample(s, 0.3, num_neighborhood=nn, n_boxes=20) for s in seeds])
    with torch.cuda.device(DEV):
        got = build_batch_gpu(raws, 0.3, DEV, jitter=synthetic.raw_sample_jitter(raws))
        torch.cuda.synchronize()
    assert torch.equal(got["sinput_C"].cpu(), want["sinput_C"])
    assert torch.equal(got["sinput_F"].cpu(), want["sinput_F"])
    assert got["batch_lengths"] == [int(v) for v in want["batch_lengths"]]
    assert torch.equal(got["group"].cpu(), want["group"]) and len(want["group"]) > 200
    assert torch.equal(got["index"].cpu(), want["index"])
    assert torch.equal(got["finest_flag"].cpu(), want["finest_flag"])
    # the centre-frame points: fp32(R p + t) from fp64 on the device == the host's numpy expression
    rows = np.cumsum([0] + got["cloud_rows"])
    xo, xc = got["xyz_own"].cpu().numpy(), got["xyz_cf"].cpu().numpy()
    n_c = nn + 1
    from oracle.colocation_oracle import colocation_groups as oracle_groups
    start, g0, i0 = 0, 0, 0
    for si, raw in enumerate(raws):
        cl = [xo[rows[si * n_c + c]:rows[si * n_c + c + 1]] for c in range(n_c)]
        cf = [xc[rows[si * n_c + c]:rows[si * n_c + c + 1]] for c in range(n_c)]
        assert np.array_equal(cf[0], cl[0])
        for j in range(nn):
            assert np.array_equal(cf[j + 1], synthetic._apply(raw["list_M"][j], cl[j + 1]))
        og, oi, of = oracle_groups(cl[0], cl[1:], raw["list_M"], raw["radius"], K=5, nghb_cf=cf[1:])
        ng, ni = len(og), len(oi)
        assert np.array_equal(got["group"][g0:g0 + ng].cpu().numpy(), np.asarray(og, dtype=np.int32))
        assert np.array_equal(got["index"][i0:i0 + ni].cpu().numpy(), np.asarray(oi, dtype=np.int64) + start)
        assert np.array_equal(got["finest_flag"][i0:i0 + ni].cpu().numpy(), np.asarray(of, dtype=bool))
        start, g0, i0 = start + got["batch_lengths"][si], g0 + ng, i0 + ni
    assert g0 == len(got["group"]) and i0 == len(got["index"])


def test_train_from_scans_equals_training_on_the_cpu_loaders_batches():
    """train_from_scans (raw scans -> build_batch_gpu a step ahead on its own stream -> train_steps) against train_steps on
    the batches the CPU loader makes from the same scans: the same losses step by step (lr = 0: every step is a function of
    its batch and draws)."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_data_gpu import train_from_scans
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config, prefetch_to_device
    seeds = [(81, 82), (83, 84), (85, 86)]
    raw_batches = [[synthetic.make_raw_sample(s, 0.3, num_neighborhood=2, n_boxes=12) for s in ss] for ss in seeds]
    cpu_batches = [synthetic.collate_train([synthetic.make_train_sample(s, 0.3, num_neighborhood=2, n_boxes=12) for s in ss])
                   for ss in seeds]
    keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
    cfg = make_config(batch_size=2, num_pos_per_batch=64, num_hn_samples_per_batch=128, lr=0.0, weight_decay=0.0)
    runs = []
    for mode in ("cpu", "scans"):
        torch.manual_seed(3)
        np.random.seed(3)
        tr = FinestContrastiveLossTrainer(cfg, device=DEV)
        order = [0, 1, 2, 0, 1]
        if mode == "cpu":
            host = [{k: v for k, v in b.items() if k in keys} for b in cpu_batches]
            steps = tr.train_steps(prefetch_to_device([host[i] for i in order], DEV, keys))
        else:
            steps = train_from_scans(tr, [raw_batches[i] for i in order], voxel_size=0.3, jitter=synthetic.raw_sample_jitter)
        runs.append([l.item() for l, _, _ in steps])
        torch.cuda.synchronize()
    assert runs[0] == runs[1] and len(set(runs[0])) >= 3, runs


def test_fused_amax_tags_equal_separate_pass():
    """BatchNorm apply / backward-apply publish max|y| / max|dx| themselves and the weight group refreshes all kernels
    in one launch: every tag must equal a separate gcl_amax pass, and in-place writes must void it."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import ops
    from gcl_amd import _lib
    lib = _lib.load()
    ME.set_conv_precision("fp16x3")

    def amax_ref(t):
        return t.detach().abs().max().view(1)

    val = ops.amax_value

    g = torch.Generator().manual_seed(5)
    x = (torch.randn(5000, 64, generator=g) * 3).to(DEV).requires_grad_(True)
    res = torch.randn(5000, 64, generator=g).to(DEV)
    bn = torch.nn.BatchNorm1d(64).to(DEV)
    y = ops.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 0.1, 1e-5, res, True)
    assert ops.known_amax(y) is not None and torch.equal(val(ops.known_amax(y)), amax_ref(y))
    seen = {}
    def hook(gx):
        tag = ops.known_amax(gx)
        seen["tag"] = (None if tag is None else val(tag), amax_ref(gx))

    x.register_hook(hook)
    (y * torch.randn(5000, 64, generator=g).to(DEV)).sum().backward()
    if seen["tag"][0] is not None:                       # the engine handed over the tagged tensor itself
        assert torch.equal(*seen["tag"])
    y2 = y.detach()
    ops.tag_amax(y2, ops.known_amax(y))
    y2.mul_(2.0)
    assert ops.known_amax(y2) is None                    # version bump voids the tag
    from gcl_amd.model import load_model
    model = load_model("ResUNetBN2C")(1, 32, normalize_feature=True, conv1_kernel_size=5, D=3).to(DEV)
    model._ensure_amax_group(model, None)
    ws = model._amax_group.params
    assert len(ws) == 22
    for w in (ws[0], ws[7], ws[-1]):
        assert torch.equal(val(ops.tensor_amax(lib, w)), amax_ref(w))
    with torch.no_grad():
        ws[7].mul_(3.0)
    assert ops.known_amax(ws[7]) is None and torch.equal(val(ops.tensor_amax(lib, ws[7])), amax_ref(ws[7]))


@pytest.mark.parametrize("n,c", [(5000, 32), (777, 64), (33, 16)])
def test_row_normalize_matches_torch_expression(n, c):
    """gcl_row_normalize_fwd/bwd == out.F / torch.norm(out.F, p=2, dim=1, keepdim=True) (model/resunet.py:226-230);
    tolerance 2e-6 rel-L2 (fp32 summation order differs)."""
    from gcl_amd.MinkowskiEngine import ops
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(n, c, generator=g).to(DEV)
    y = ops.l2_normalize_rows(x)
    (y * w).sum().backward()
    gx, x.grad = x.grad.clone(), None
    xr = x.detach().double().requires_grad_(True)
    yr = xr / torch.norm(xr, p=2, dim=1, keepdim=True)
    (yr * w.double()).sum().backward()
    assert rel_l2(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 2e-6
    assert rel_l2(gx.cpu().numpy(), xr.grad.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("overrides", [dict(block_finest_gradient=True), dict(square_loss=False),
                                       dict(use_pair_group_positive_loss=True), dict(finest_weight=0),
                                       dict(use_group_circle_loss=True, block_finest_gradient=True),
                                       dict(use_hard_negative=False)])
def test_trainer_accepts_the_other_loss_switches(overrides):
    """config.py:38-43 / :158 switches other than the training script's selection run through the trainer (the
    switch-by-switch values are pinned by the golden tests above, use_hard_negative=False included since round 3)."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    batch = synthetic.make_train_batch(77, batch_size=1, num_neighborhood=2, n_boxes=12)
    tr = FinestContrastiveLossTrainer(make_config(batch_size=1, num_pos_per_batch=64, num_hn_samples_per_batch=128,
                                                  **overrides), device=DEV)
    np.random.seed(1)
    loss, (pos, fin, neg), _ = tr.train_step(batch)
    assert torch.isfinite(loss).item() and pos.item() >= 0
    if overrides.get("finest_weight", 1) == 0:
        assert fin.item() == 0.0
    assert all(torch.isfinite(p).all() for p in tr.model.parameters())


def test_instance_norm_matches_per_cloud_formula():
    """ME.MinkowskiInstanceNorm semantics (per cloud and channel: biased variance, eps 1e-8, shared affine [1, C]) vs a
    fp64 torch restatement, forward and backward, with the fused residual + ReLU; tolerance 2e-6 rel-L2."""
    import gcl_amd.MinkowskiEngine as ME
    g = torch.Generator().manual_seed(11)
    sizes = [700, 33, 1500, 1]
    C = torch.cat([torch.cat([torch.full((n, 1), b, dtype=torch.int32),
                              torch.stack([torch.arange(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32),
                                           torch.full((n,), b, dtype=torch.int32)], 1)], 1)
                   for b, n in enumerate(sizes)]).to(DEV)
    n, c = C.shape[0], 64
    x = (torch.randn(n, c, generator=g) * 2 + 0.5).to(DEV).requires_grad_(True)
    r = torch.randn(n, c, generator=g).to(DEV).requires_grad_(True)
    wgt = torch.randn(n, c, generator=g).to(DEV)
    norm = ME.MinkowskiInstanceNorm(c, dimension=3).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(1, c, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(1, c, generator=g) * 0.1)
    xs = ME.SparseTensor(x, coordinates=C)
    rs = ME.SparseTensor(r, coordinate_map_key=xs.coordinate_map_key, coordinate_manager=xs.coordinate_manager)
    y = norm(xs, residual=rs, relu=True).F
    (y * wgt).sum().backward()
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    outs, start = [], 0
    for m in sizes:
        seg = xd[start:start + m]
        mu = seg.mean(0, keepdim=True)
        var = ((seg - mu) ** 2).mean(0, keepdim=True)
        outs.append((seg - mu) / torch.sqrt(var + 1e-8) * wd + bd)
        start += m
    yr = torch.relu(torch.cat(outs) + rd)
    (yr * wgt.double()).sum().backward()
    assert rel_l2(y.detach().cpu(), yr.detach().cpu()) < 2e-6
    # a 1-voxel cloud has variance 0: its gradient is scaled by 1 / sqrt(1e-8) -- compare the rest tightly
    keep = torch.ones(n, dtype=torch.bool)
    keep[-1] = False
    assert rel_l2(x.grad.cpu()[keep], xd.grad.cpu()[keep]) < 1e-5
    assert rel_l2(r.grad.cpu(), rd.grad.cpu()) < 2e-6
    assert rel_l2(norm.weight.grad.cpu(), wd.grad.cpu()) < 1e-5 and rel_l2(norm.bias.grad.cpu(), bd.grad.cpu()) < 1e-5


def test_in_model_variant_trains():
    """ResUNetIN2C (model/resunet.py:279-281): BatchNorm after the level convolutions, InstanceNorm in the blocks."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    batch = synthetic.make_train_batch(78, batch_size=1, num_neighborhood=2, n_boxes=12)
    tr = FinestContrastiveLossTrainer(make_config(model="ResUNetIN2C", batch_size=1, num_pos_per_batch=64,
                                                  num_hn_samples_per_batch=128), device=DEV)
    keys = tr.model.state_dict().keys()
    assert "block1.norm1.weight" in keys and "norm1.bn.weight" in keys and "block1.norm1.bn.weight" not in keys
    np.random.seed(2)
    l0 = tr.train_step(batch)[0].item()
    for _ in range(3):
        l1 = tr.train_step(batch)[0].item()
    assert np.isfinite(l0) and np.isfinite(l1)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "hardest_loss_*.npz"))))
def test_fcgf_hardest_contrastive_loss_golden(path):
    """lib/trainer.py:410-462 through gcl_nn_rowmin: (pos, neg) within 2e-6 abs of the reference's own output, gradients
    within 1e-5 rel-L2; host draws from a seeded np.random reproduce the reference's selections."""
    from gcl_amd.lib.trainer import contrastive_hardest_negative_loss
    z = np.load(path)
    F0 = torch.from_numpy(z["F0"]).to(DEV).requires_grad_(True)
    F1 = torch.from_numpy(z["F1"]).to(DEV).requires_grad_(True)
    kw = dict(num_pos=int(z["num_pos"]), num_hn_samples=int(z["num_hn"]))
    draws = (z["sel0"], z["sel1"], z["pos_sel"] if bool(z["subsampled"]) else None)
    pos, neg = contrastive_hardest_negative_loss(F0, F1, z["pairs"], draws=draws, **kw)
    assert abs(pos.item() - float(z["pos"])) < 2e-6 and abs(neg.item() - float(z["neg"])) < 2e-6
    (pos + neg).backward()
    assert rel_l2(F0.grad.cpu(), z["grad0"]) < 1e-5 and rel_l2(F1.grad.cpu(), z["grad1"]) < 1e-5
    np.random.seed(int(z["np_seed"]))
    p2, n2 = contrastive_hardest_negative_loss(F0.detach(), F1.detach(), z["pairs"], **kw)
    assert abs(p2.item() - float(z["pos"])) < 2e-6 and abs(n2.item() - float(z["neg"])) < 2e-6


# ---------------------------------------------------------------------------------------------------------------
# "next" row: SC2-PCR registration back-end
# ---------------------------------------------------------------------------------------------------------------
SC2_KEYS = ("inlier_threshold", "d_thre", "num_iterations", "ratio", "nms_radius", "max_points", "k1", "k2")


def _sc2_cfg(z):
    return {k: (int(z[k]) if k in ("num_iterations", "max_points", "k1", "k2") else float(z[k])) for k in SC2_KEYS}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "sc2pcr_*.npz"))))
def test_sc2pcr_matches_reference_output_and_oracle_stages(path):
    """Final transformation within 2e-3 of the reference's own output (tie order is unspecified there) and within
    2e-4 of the oracle, which shares the lowest-index tie rule; the integer stages (seeds, k1-neighbour sets, best
    hypothesis) equal the oracle's exactly, confidences within 1e-5."""
    from gcl_amd.scripts.SC2_PCR import Matcher
    from oracle.sc2pcr_oracle import sc2_pcr
    z = np.load(path)
    cfg = _sc2_cfg(z)
    m = Matcher(num_node="all", use_mutual=False, **cfg)
    src, tgt = torch.from_numpy(z["src"]).to(DEV), torch.from_numpy(z["tgt"]).to(DEV)
    T = m.SC2_PCR(src[None], tgt[None])[0].cpu().numpy()
    assert np.abs(T - z["T_ref"]).max() < 2e-3
    To, st = sc2_pcr(z["src"], z["tgt"], return_stages=True, **cfg)
    assert np.abs(T - To.numpy()).max() < 2e-4
    assert np.allclose(m.last["conf"].cpu().numpy(), st["conf"].numpy(), rtol=0, atol=1e-5)
    same_seeds = np.array_equal(m.last["seeds"].cpu().numpy(), st["seeds"].numpy())
    if same_seeds:        # confidences equal to ~1e-7: the seed order can only differ on near-ties
        assert np.array_equal(m.last["knn"].cpu().numpy(), st["knn"].numpy())
        assert np.array_equal(m.last["fitness"].cpu().numpy(), st["fitness"].numpy())
        assert int(m.last["best"]) == st["best"]
        # hypotheses of consistent seeds agree closely; seeds sitting on outliers give near-singular 3 x 3 problems
        good = st["fitness"].numpy() >= 0.5 * st["fitness"].numpy().max()
        d = np.abs(m.last["seed_trans"].cpu().numpy().reshape(-1, 3, 4) - st["seed_trans"][:, :3, :].numpy())
        assert good.sum() >= 5 and d[good].max() < 5e-3
    else:
        assert set(m.last["seeds"].cpu().numpy()[:50]) == set(st["seeds"].numpy()[:50])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "sc2pcr_*.npz"))))
def test_sc2pcr_one_call_equals_the_staged_calls(path, monkeypatch):
    """gcl_sc2_register (every stage of Matcher.SC2_PCR + the estimator's labels behind ONE C-ABI call, the seed order by a
    device sort) == the five staged calls with torch's stable sort / argmax between them: every stage and the transformation
    bit for bit, the labels equal to torch's expression except where a residual sits within rounding of the threshold."""
    import gcl_amd.scripts.SC2_PCR as S
    z = np.load(path)
    cfg = _sc2_cfg(z)
    src, tgt = torch.from_numpy(z["src"]).to(DEV), torch.from_numpy(z["tgt"]).to(DEV)
    res = []
    for one_call in (False, True):
        monkeypatch.setattr(S, "ONE_CALL", one_call)
        m = S.Matcher(num_node="all", use_mutual=False, **cfg)
        T = m.SC2_PCR(src[None], tgt[None]).clone()
        res.append((T, {k: m.last[k].clone() for k in ("conf", "seeds", "knn", "seed_trans", "fitness", "best")}, m._labels))
    assert res[0][2] is None and res[1][2] is not None
    assert torch.equal(res[0][0], res[1][0]) and res[1][0].shape == (1, 4, 4)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k].to(res[1][1][k].dtype), res[1][1][k]), k
    T = res[1][0]
    n = min(src.shape[0], cfg["max_points"])
    warped = src[None, :n] @ T[:, :3, :3].transpose(1, 2) + T[:, None, :3, 3]
    dist = torch.sum((warped - tgt[None, :n]) ** 2, dim=-1) ** 0.5
    ref = (dist < cfg["inlier_threshold"]).float()
    differ = ref != res[1][2]
    assert res[1][2].shape == ref.shape
    assert not bool(differ.any()) or float((dist[differ] - cfg["inlier_threshold"]).abs().max()) < 1e-5


def test_sc2pcr_seed_order_of_the_one_call_form_with_ties():
    """The device sort behind gcl_sc2_register's seeds: (value descending, index ascending) = torch.sort(-(conf * is_max),
    stable=True) -- on a problem whose confidences are massively tied (a regular grid: many equal entries, most of them
    suppressed to exactly zero)."""
    import gcl_amd.scripts.SC2_PCR as S
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    src = torch.from_numpy(g * 0.5).to(DEV)
    tgt = src + torch.tensor([1.0, 2.0, 0.5], device=DEV)
    cfg = dict(inlier_threshold=0.6, d_thre=0.1, num_iterations=10, ratio=0.5, nms_radius=0.6, max_points=8000, k1=30, k2=20)
    out = []
    for one_call in (False, True):
        S.ONE_CALL = one_call
        try:
            m = S.Matcher(num_node="all", use_mutual=False, **cfg)
            m.SC2_PCR(src[None], tgt[None])
            out.append((m.last["conf"].clone(), m.last["seeds"].clone()))
        finally:
            S.ONE_CALL = True
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert len(torch.unique(out[1][0])) < len(out[1][0]) // 2          # the ties are there


@pytest.mark.parametrize("n,inlier,noise", [(8000, 0.3, 0.03), (8000, 1.0, 0.0), (3000, 0.6, 0.02), (257, 0.5, 0.02), (1, 1.0, 0.0)])
def test_sc2pcr_confidence_sparse_equals_dense_bitwise(n, inlier, noise):
    """gcl_sc2_confidence_sparse (the compatibility matrix's non-zero entries kept from ONE build; 20 products over them)
    == gcl_sc2_confidence (every entry re-derived in every product), bit for bit: the same non-zero terms in the same order,
    the skipped ones exact zeros.  (8000, all inliers, no noise): every one of the 64 M entries is non-zero -- each segment
    of the ELL layout is full."""
    from gcl_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(n)
    src = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    src[:, 2] *= 0.1
    ang = np.deg2rad(9.0)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tgt = (src @ R.T + np.array([3.0, 1.0, 0.1]) + rng.normal(0, noise, (n, 3))).astype(np.float32)
    out = rng.rand(n) > inlier
    tgt[out] = rng.uniform(-40, 40, (int(out.sum()), 3)).astype(np.float32)
    with torch.cuda.device(DEV):
        s, t = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
        res = []
        for sparse in (False, True):
            conf = torch.ones(n, device=DEV)
            partial = torch.full((lib.gcl_sc2_chunks() * n,), float("nan"), device=DEV)
            done = torch.zeros(1, dtype=torch.int32, device=DEV)
            if sparse:
                scratch = torch.empty(lib.gcl_sc2_confidence_scratch_bytes(n), dtype=torch.uint8, device=DEV)
                _lib.check(lib.gcl_sc2_confidence_sparse(_lib.ptr(s), _lib.ptr(t), n, 0.1, 20, _lib.ptr(partial), _lib.ptr(conf),
                                                         _lib.ptr(done), _lib.ptr(scratch), _lib.stream()), "sparse")
            else:
                _lib.check(lib.gcl_sc2_confidence(_lib.ptr(s), _lib.ptr(t), n, 0.1, 20, _lib.ptr(partial), _lib.ptr(conf),
                                                  _lib.ptr(done), _lib.stream()), "dense")
            res.append((conf, partial, done))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
        assert bool(torch.isfinite(res[0][0]).all())
        if os.environ.get("GCL_SC2_FOLDED_NORMALIZE") == "0":      # every product followed by its own normalisation launch:
            assert torch.equal(res[0][1], res[1][1])              # then the last product's partial sums are the same buffer too


def test_sc2pcr_one_launch_refinement_and_sparse_confidence_vs_round_1_forms(tmp_path):
    """The round-5 forms of two SC2-PCR stages -- the refinement as ONE persistent launch (k_sc_refine_all) and the
    confidence's products over kept non-zero entries -- against the round-1 forms (GCL_SC2_REFINE_ONE_LAUNCH=0,
    GCL_SC2_SPARSE=0, selected in a fresh process: the switches are read once): the transformation of every golden problem
    agrees to 2e-6 (the confidence stage is bitwise, test above; the refinement's fp64 sums take another fixed order)."""
    import subprocess
    script = (
        "import sys, glob, os, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from gcl_amd.scripts.SC2_PCR import Matcher\n"
        "out = {}\n"
        "for path in sorted(glob.glob(os.path.join(%r, 'sc2pcr_*.npz'))):\n"
        "    z = np.load(path)\n"
        "    cfg = {k: (int(z[k]) if k in ('num_iterations', 'max_points', 'k1', 'k2') else float(z[k])) for k in %r}\n"
        "    m = Matcher(num_node='all', use_mutual=False, **cfg)\n"
        "    with torch.cuda.device(%r):\n"
        "        T = m.SC2_PCR(torch.from_numpy(z['src']).to(%r)[None], torch.from_numpy(z['tgt']).to(%r)[None])\n"
        "    out[os.path.basename(path)] = T[0].cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n") % (ROOT, G, SC2_KEYS, DEV, DEV, DEV)
    res = {}
    for tag, env in (("new", {}), ("old", {"GCL_SC2_REFINE_ONE_LAUNCH": "0", "GCL_SC2_SPARSE": "0"})):
        f = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", script, f], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    assert len(res["new"].files) >= 3
    for k in res["new"].files:      # (the one-launch refinement sums its fp64 terms in another fixed order: last-bit differences)
        assert np.abs(res["new"][k] - res["old"][k]).max() < 2e-6, k


def test_sc2pcr_estimator_end_to_end_at_kitti_size():
    """Matcher.estimator with config_KITTI.json (8000 sampled nodes from 5000 voxels): features -> correspondences ->
    transformation; 30 % of the voxels have a true match."""
    from gcl_amd.scripts.SC2_PCR import Matcher
    rng = np.random.RandomState(3)
    g = torch.Generator().manual_seed(3)
    n = 5000
    xyz0 = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    xyz0[:, 2] *= 0.1
    ang = np.deg2rad(12.0)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = np.array([8.0, -1.0, 0.2])
    perm = rng.permutation(n)
    xyz1 = (xyz0 @ R.T + t + rng.normal(0, 0.03, (n, 3))).astype(np.float32)[perm]
    F0 = torch.randn(n, 32, generator=g)
    F0 = F0 / F0.norm(dim=1, keepdim=True)
    F1 = F0.clone()
    bad = torch.rand(n, generator=g) > 0.3
    F1[bad] = torch.nn.functional.normalize(torch.randn(int(bad.sum()), 32, generator=g), dim=1)
    F1 = F1[torch.from_numpy(perm)]
    m = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                nms_radius=0.6, max_points=8000, k1=30, k2=20)
    np.random.seed(0)
    T, labels, s, tt = m.estimator(torch.from_numpy(xyz0).to(DEV)[None], torch.from_numpy(xyz1).to(DEV)[None],
                                   F0.to(DEV)[None], F1.to(DEV)[None])
    stages = {k: m.last[k].clone() for k in ("conf", "seeds", "knn", "seed_trans", "fitness", "best")}
    Tn = T[0].cpu().numpy()
    assert np.abs(Tn[:3, :3] - R).max() < 2e-3 and np.abs(Tn[:3, 3] - t).max() < 2e-2
    assert labels.shape == (1, 8000) and 0.2 < labels.mean().item() < 0.4 and s.shape == (1, 8000, 3)
    # the same problem through the five staged calls (GCL_SC2_ONE_CALL=0's path): at the maximum size too, every stage and the
    # transformation are those of the one-call form bit for bit (the tight bits come from the build pass there, from
    # k_sc_tight_bits here; the seed order from a rank count there, from torch.sort here)
    import gcl_amd.scripts.SC2_PCR as S
    S.ONE_CALL = False
    try:
        m2 = Matcher(inlier_threshold=0.6, num_node=8000, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                     nms_radius=0.6, max_points=8000, k1=30, k2=20)
        T2 = m2.SC2_PCR(s, tt)
    finally:
        S.ONE_CALL = True
    assert torch.equal(T2, T)
    for k, v in stages.items():
        assert torch.equal(m2.last[k].to(v.dtype), v), k


@pytest.mark.parametrize("cin,cout,n,stride,transpose", [(128, 128, 5000, 1, False), (256, 256, 2500, 1, False),
                                                          (128, 256, 6000, 2, False), (256, 128, 6000, 2, True),
                                                          (256, 64, 3000, 1, False), (128, 128, 300, 1, False)])
def test_inference_offset_group_launches_equal_the_sixteen_wave_kernel_bitwise(cin, cout, n, stride, transpose, monkeypatch):
    """The deep layers of an inference pass (GCL_CONV_TALL launches, Cin >= 128): the four fixed offset groups as four times
    as many ordinary workgroups + one sum / epilogue launch (round 6: k_conv_fwd_dma<2, false, false, GRP> + k_conv_groups_sum,
    taken when the caller hands over scratch) against the sixteen-wave kernel k_conv_fwd_tall -- the same products in the same
    order per group, the same group order, the same epilogue expression: bit for bit, with BatchNorm scale / shift, residual
    and ReLU in the epilogue, on stride-1, strided and transposed maps; and both against the fp64 oracle."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.MinkowskiEngine import ops
    C = random_cloud(cin + n, n=n, extent=16, batch=2)
    mgr, omgr = make_mgr(C), O.CoordinateManager(C)
    km = mgr.get_kernel_map(1, 3, stride)
    n_in, n_out_map = len(C), mgr.num_rows(stride)
    n_x, n_y = (n_out_map, n_in) if transpose else (n_in, n_out_map)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n_x, cin, generator=g)
    W = torch.randn(27, cin, cout, generator=g) * 0.05
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n_y, cout, generator=g)
    outs = []
    with torch.no_grad(), torch.cuda.device(DEV):
        xd, Wd, sd, hd, rd = (t.to(DEV) for t in (x, W, scale, shift, res))
        for groups in (True, False):
            monkeypatch.setattr(ops, "GROUP_LAUNCHES", groups)
            outs.append(ops.conv_bn_eval(xd, Wd, km, n_y, transpose, sd, hd, residual=rd, relu=True).clone())
            outs.append(ops.conv_bn_eval(xd, Wd, km, n_y, transpose, sd, hd).clone())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    yo = O.sparse_conv(x.double(), W.double(), omgr.get_kernel_map(1, 3, stride), n_y, transpose=transpose)
    assert rel_l2(outs[1].cpu(), yo * scale.double() + shift.double()) < 2e-6
    assert rel_l2(outs[0].cpu(), torch.relu(yo * scale.double() + shift.double() + res.double())) < 2e-6


def test_forward_pair_equals_two_forward_passes_bitwise():
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.lib.eval import forward_pair
    m, _ = _model_and_state(4, 5)
    m.eval()
    p = synthetic.make_eval_pair(5, baseline=15.0, n_boxes=15)
    F0, C0, F1, C1 = (p[k].to(DEV) for k in ("sinput0_F", "sinput0_C", "sinput1_F", "sinput1_C"))
    with torch.no_grad():
        a0 = m(ME.SparseTensor(F0, coordinates=C0)).F
        a1 = m(ME.SparseTensor(F1, coordinates=C1)).F
        b0, b1 = forward_pair(m, F0, C0, F1, C1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_split_map_build_pass_equals_the_one_stream_pass_bitwise(monkeypatch):
    """A one-call inference pass builds the deeper levels' maps on a side stream beside the first layers' convolutions
    (gcl_maps_build_split; the plan waits in front of the first record that uses one): the same launches on the same data, so
    the features, every level's coordinates and every sorted table equal the one-stream build bit for bit -- over repeated
    passes of alternating sizes (arena re-use between the streams), and with a consumer other than the plan (wait_ready)."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.MinkowskiEngine import native
    from gcl_amd.scripts.test_kitti import forward_clouds
    m, _ = _model_and_state(4, 5)
    m.eval()
    pairs = [synthetic.make_eval_pair(5 + s, baseline=15.0 + 5 * s, n_boxes=15 + 10 * (s % 2)) for s in range(3)]
    sets = [[(p[f"sinput{k}_F"].to(DEV), p[f"sinput{k}_C"].to(DEV)) for k in (0, 1)] for p in pairs]
    with torch.no_grad(), torch.cuda.device(DEV):
        monkeypatch.setenv("GCL_EVAL_SPLIT_MAPS", "0")
        forward_clouds(m, sets[0])                      # the first pass records the plan (per-operator path)
        ref = [[f.clone() for f in forward_clouds(m, s)] for s in sets]
        monkeypatch.setenv("GCL_EVAL_SPLIT_MAPS", "1")
        monkeypatch.setenv("GCL_EVAL_SPLIT_MIN_ROWS", "0")         # (default: passes of >= 100 000 rows)
        assert native.eval_side_stream(DEV, 10) is not None
        for rep in range(4):
            for s, r in zip(sets, ref):
                got = forward_clouds(m, s)
                assert all(torch.equal(a, b) for a, b in zip(got, r)), rep
        # the maps themselves, read by something that is not the plan
        C = torch.cat([c.clone() for _, c in sets[1]])
        C[: len(sets[1][0][1]), 0] = 0
        C[len(sets[1][0][1]):, 0] = 1
        specs = m.native_map_specs(training=False)
        one = ME.CoordinateManager.build_native(C, specs).native
        two = ME.CoordinateManager.build_native(C, specs, side_stream=native.eval_side_stream(DEV)).native
        assert two.ready is not None and two.desc.late_mask != 0 and one.desc.late_mask == 0
        two.wait_ready()
        torch.cuda.synchronize()
        for l in range(one.n_levels):
            assert one.num_rows(l) == two.num_rows(l)
            assert torch.equal(one.view(one.desc.coords[l], (one.num_rows(l), 4), torch.int32),
                               two.view(two.desc.coords[l], (two.num_rows(l), 4), torch.int32))
        for q in range(one.desc.n_maps):
            a, b = one.desc.maps[q], two.desc.maps[q]
            assert (a.K, a.n_in, a.n_out) == (b.K, b.n_in, b.n_out)
            for name, rows in (("nbr", a.n_out), ("nbr_t", a.n_in), ("tbl_n", a.n_out), ("tbl_t", a.n_in)):
                pa, pb = getattr(a, name), getattr(b, name)
                assert bool(pa) == bool(pb)
                if pa:
                    assert torch.equal(one.view(pa, (a.K, rows), torch.int32), two.view(pb, (b.K, rows), torch.int32)), (q, name)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "circle_loss_*.npz"))))
def test_location_circle_loss_golden(path):
    """lib/colocation_trainer.py:538-681 on the device vs the reference's own outputs: loss triple within 1e-5 rel,
    dL/dF within 1e-4 rel-L2 (exp / log chains in fp32); host draws reproduce the reference's selections."""
    from gcl_amd.lib.colocation_trainer import location_circle_loss
    z = np.load(path)
    sw = {k: bool(z[k]) for k in ("square_loss", "block_finest_gradient", "use_pair_group_positive_loss")}
    F = torch.from_numpy(z["F_out"]).to(DEV).requires_grad_(True)
    draws = (z["pos_sel"], z["pair_pos"] if "pair_pos" in z.files else None)
    kw = dict(max_pos_cluster=int(z["max_pos_cluster"]), points=torch.from_numpy(z["points"]),
              batch_lengths=z["batch_lengths"].tolist(), **sw)
    args = (torch.from_numpy(z["group"]), torch.from_numpy(z["index"]), None, torch.from_numpy(z["finest_flag"]))
    pos, fin, neg = location_circle_loss(F, *args, draws=draws, **kw)
    for got, key in ((pos, "pos"), (fin, "finest"), (neg, "neg")):
        assert abs(got.item() - float(z[key])) <= 1e-5 * max(1, abs(float(z[key]))), key
    (pos + fin + neg).backward()
    assert rel_l2(F.grad.cpu(), z["grad"]) < 1e-4
    np.random.seed(int(z["np_seed"]))
    p2, f2, n2 = location_circle_loss(F.detach(), *args, **kw)
    assert abs(p2.item() - float(z["pos"])) <= 1e-5 * max(1, abs(float(z["pos"])))
    assert abs(n2.item() - float(z["neg"])) <= 1e-5
