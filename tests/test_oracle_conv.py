"""Pins the ORACLE's sparse-conv semantics against an independent dense conv3d."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O


def _cloud(seed, n=400, extent=12, batch=1):
    rng = np.random.RandomState(seed)
    cs = []
    for b in range(batch):
        c = np.unique(rng.randint(-extent, extent, (n, 3)), axis=0)
        rng.shuffle(c)
        cs.append(np.concatenate([np.full((len(c), 1), b), c], axis=1))
    return np.concatenate(cs).astype(np.int32)


@pytest.mark.parametrize("ks,t_in,stride", [(3, 1, 1), (5, 1, 1), (3, 1, 2), (3, 2, 1), (3, 2, 2), (3, 4, 2)])
def test_kernel_map_np_equals_dict(ks, t_in, stride):
    C = _cloud(0, batch=2)
    mgr = O.CoordinateManager(C)
    Cin, Cout = mgr.get_coords(t_in), mgr.get_coords(t_in * stride)
    a = O.canonical(O.kernel_map_dict(Cin, Cout, ks, t_in))
    b = O.canonical(O.kernel_map_np(Cin, Cout, ks, t_in))
    assert a.shape == b.shape and (a == b).all()
    assert len(a) > 0


def test_stride_coords_floor_negative():
    C = np.array([[0, -1, -2, -3], [0, -4, 0, 1], [0, -3, -1, -4], [0, 3, 2, 1]], dtype=np.int32)
    out = O.stride_coords(C, 2)
    # floor toward -inf: -1 -> -2, -3 -> -4 ; rows 0 and 2 do NOT merge with row 1; first-occurrence order
    assert out.tolist() == [[0, -2, -2, -4], [0, -4, 0, 0], [0, -4, -2, -4], [0, 2, 2, 0]]


@pytest.mark.parametrize("ks", [3, 5])
def test_stride1_matches_dense(ks):
    C = _cloud(1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(len(C), 3, generator=g, dtype=torch.float64)
    W = torch.randn(ks ** 3, 3, 4, generator=g, dtype=torch.float64)
    tri = O.kernel_map_np(C, C, ks, 1)
    y = O.sparse_conv(x, W, tri, len(C))
    yd = O.dense_conv_reference(C, x, W, ks, 1, 1, C_out=C)
    assert (y - yd).abs().max() < 1e-12


@pytest.mark.parametrize("t_in", [1, 2])
def test_stride2_and_transpose_match_dense(t_in):
    C = _cloud(2, n=600)
    mgr = O.CoordinateManager(C)
    Cf, Cc = mgr.get_coords(t_in), mgr.get_coords(2 * t_in)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(len(Cf), 3, generator=g, dtype=torch.float64)
    W = torch.randn(27, 3, 5, generator=g, dtype=torch.float64)
    tri = mgr.get_kernel_map(t_in, 3, 2)
    y = O.sparse_conv(x, W, tri, len(Cc))
    yd = O.dense_conv_reference(Cf, x, W, 3, t_in, 2, C_out=Cc)
    assert (y - yd).abs().max() < 1e-12
    # transposed: coarse -> fine through the same map, swapped
    xc = torch.randn(len(Cc), 5, generator=g, dtype=torch.float64)
    Wt = torch.randn(27, 5, 3, generator=g, dtype=torch.float64)
    yt = O.sparse_conv(xc, Wt, tri, len(Cf), transpose=True)
    ytd = O.dense_conv_reference(Cc, xc, Wt, 3, 2 * t_in, 2, transpose=True, C_out=Cf)
    assert (yt - ytd).abs().max() < 1e-12
    # adjointness <conv(x; W), y> == <x, convT(y; W^T)>
    lhs = (y * xc).sum()
    rhs = (x * O.sparse_conv(xc, W.transpose(1, 2), tri, len(Cf), transpose=True)).sum()
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_resunet_forward_runs_and_is_normalised():
    C = _cloud(3, n=900, extent=16, batch=2)
    st = O.random_state(0, conv1_kernel_size=5)
    f = torch.ones(len(C), 1, dtype=torch.float64)
    out = O.resunet_forward(st, C, f, conv1_kernel_size=5)
    assert out.shape == (len(C), 32)
    assert torch.allclose(out.norm(dim=1), torch.ones(len(C), dtype=torch.float64), atol=1e-12)
    n_params = sum(int(np.prod(v.shape)) for k, v in st.items() if "running" not in k)
    assert n_params == 8753408


# ---------------------------------------------------------------------------------------------------------------
# the C / OpenMP restatement of ME's CPU path (oracle/me_cpu.c; bench.py's cpu_baseline) against the torch oracle
# ---------------------------------------------------------------------------------------------------------------
def test_c_restatement_maps_bit_exact():
    from oracle import me_cpu
    C = _cloud(3, n=1500, extent=14, batch=2)
    mo, mc = O.CoordinateManager(C), me_cpu.CoordinateManager(C)
    for t in (2, 4, 8):
        assert np.array_equal(mo.get_coords(t), mc.get_coords(t))
    for key in [(1, 3, 1), (1, 5, 1), (1, 3, 2), (2, 3, 1), (2, 3, 2), (4, 3, 2), (8, 3, 1)]:
        a, b = O.canonical(mo.get_kernel_map(*key)), O.canonical(mc.get_kernel_map(*key).triples())
        assert a.shape == b.shape and np.array_equal(a, b), key
    with pytest.raises(ValueError, match="duplicate"):
        me_cpu.CoordinateManager(np.concatenate([C, C[:2]]))


@pytest.mark.parametrize("cin,cout,ks,stride,transpose", [(32, 64, 3, 2, False), (64, 64, 3, 1, False),
                                                          (128, 64, 3, 2, True), (1, 32, 5, 1, False),
                                                          (48, 20, 3, 1, False)])
def test_c_restatement_conv_fwd_bwd(cin, cout, ks, stride, transpose):
    from oracle import me_cpu
    C = _cloud(4, n=1200, extent=10, batch=2)
    mo, mc = O.CoordinateManager(C), me_cpu.CoordinateManager(C)
    g = torch.Generator().manual_seed(1)
    t_in = 2 if transpose else 1
    key = (t_in // stride, ks, stride) if transpose else (t_in, ks, stride)
    n_in = len(mo.get_coords(t_in))
    n_out = len(mo.get_coords(t_in // stride if transpose else t_in * stride))
    x = torch.randn(n_in, cin, generator=g, dtype=torch.float64)
    W = torch.randn(ks ** 3, cin, cout, generator=g, dtype=torch.float64) / 8
    b = torch.randn(1, cout, generator=g, dtype=torch.float64)
    xo, Wo = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    yo = O.sparse_conv(xo, Wo, mo.get_kernel_map(*key), n_out, transpose=transpose, bias=b)
    xc, Wc = x.float().requires_grad_(True), W.float().requires_grad_(True)
    yc = O.sparse_conv(xc, Wc, mc.get_kernel_map(*key), n_out, transpose=transpose, bias=b.float())
    gy = torch.randn(yo.shape, generator=g, dtype=torch.float64)
    yo.backward(gy)
    yc.backward(gy.float())
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert rel(yc.detach(), yo.detach()) < 2e-6 and rel(xc.grad, xo.grad) < 2e-6 and rel(Wc.grad, Wo.grad) < 2e-6


def test_c_restatement_whole_network_equals_torch_oracle():
    from oracle import me_cpu
    C = _cloud(5, n=900, extent=10, batch=2)
    st = O.random_state(0, dtype=torch.float32)
    feats = torch.ones(len(C), 1)
    a = O.resunet_forward({k: v.clone() for k, v in st.items()}, C, feats, 5, True, True, 0.05)
    b = O.resunet_forward({k: v.clone() for k, v in st.items()}, C, feats, 5, True, True, 0.05,
                          mgr=me_cpu.CoordinateManager(C))
    assert float((a - b).norm() / a.norm()) < 1e-5
