"""GPU tests of the drop-in boundary (round 2): the UN-FUSED operator surface the reference's own model files would
use, generic channel counts (demo.py:29's 16-dim model), the wide-stem variants, the eval loop as a product function,
gradient accumulation, and the cache / tag invalidation rules of the host shim."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_oracle as LO          # noqa: E402
from oracle import me_oracle as O             # noqa: E402

DEV = "cuda:0"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))        # tests/ref_shaped_model.py


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _randomise_bn(m):
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("bn.weight"):
                p.uniform_(0.5, 1.5)
            elif name.endswith("bn.bias"):
                p.uniform_(-0.1, 0.1)


def _state64(m):
    return {k: v.detach().cpu().double().clone() for k, v in m.state_dict().items() if "num_batches" not in k}


def _lidar_cloud(seed, stride=3, n_boxes=20):
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    xyz = synthetic.raycast(synthetic.make_scene(seed, n_boxes=n_boxes), np.zeros(3), seed + 7)[::stride]
    coords, _ = ME.utils.sparse_quantize(xyz / 0.3, return_index=True)
    return ME.utils.batched_coordinates([coords])


# ---------------------------------------------------------------------------------------------------------------
# 1a. un-fused, reference-shaped forward: conv -> norm -> MEF.relu, `out += residual`, ME.cat, final SparseTensor
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k1", [5, 3])
def test_unfused_reference_shaped_model_matches_fused_model_and_oracle(k1):
    import gcl_amd.MinkowskiEngine as ME
    import gcl_amd.MinkowskiEngine.MinkowskiFunctional as MEF
    from gcl_amd.model import load_model
    from ref_shaped_model import build
    torch.manual_seed(0)
    fused = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=k1, D=3).to(DEV)
    _randomise_bn(fused)
    plain = build(ME, MEF, 1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=k1).to(DEV)
    missing = plain.load_state_dict(fused.state_dict(), strict=True)          # same names, same shapes
    assert not missing.missing_keys and not missing.unexpected_keys
    st = _state64(fused)
    C = _lidar_cloud(1)
    feats = torch.ones(len(C), 1)
    g = torch.Generator().manual_seed(1)
    gy = torch.randn(len(C), 32, generator=g, dtype=torch.float64)
    outs = {}
    for name, m in (("fused", fused), ("plain", plain)):
        m.train()
        out = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV)))
        assert np.array_equal(out.C.cpu().numpy(), C.numpy())
        out.F.backward(gy.float().to(DEV))
        outs[name] = (out.F.detach().cpu(), {n: p.grad.cpu().clone() for n, p in m.named_parameters()},
                      {n: b.cpu().clone() for n, b in m.named_buffers() if "running" in n})
    so = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    Fo = O.resunet_forward(so, C.numpy(), feats.double(), k1, True, True, 0.05)
    Fo.backward(gy)
    # the two host paths drive the same kernels: features agree to fp32 rounding of the (differently fused) BN passes
    assert rel_l2(outs["plain"][0], outs["fused"][0]) < 2e-6
    assert rel_l2(outs["plain"][0], Fo.detach()) < 1e-4 and rel_l2(outs["fused"][0], Fo.detach()) < 1e-4
    for n in outs["plain"][1]:
        assert rel_l2(outs["plain"][1][n], so[n].grad) < 2e-3, n
        assert rel_l2(outs["plain"][1][n], outs["fused"][1][n]) < 1e-3, n
    for n in outs["plain"][2]:
        assert rel_l2(outs["plain"][2][n], so[n]) < 1e-4, n
    # eval mode (running statistics): un-fused BatchNorm reads the buffers, the fused model uses conv+BN in one launch
    with torch.no_grad():
        fused.eval(), plain.eval()
        Fe_f = fused(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F.cpu()
        Fe_p = plain(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F.cpu()
    Feo = O.resunet_forward({k: v.detach() for k, v in so.items()}, C.numpy(), feats.double(), k1, True, False, 0.05)
    assert rel_l2(Fe_p, Feo) < 1e-4 and rel_l2(Fe_f, Feo) < 1e-4 and rel_l2(Fe_p, Fe_f) < 2e-6


def test_unfused_surface_pieces():
    """MEF.relu on a tensor that is NOT known to be non-negative, SparseTensor.__iadd__ dropping stale BN column sums,
    MinkowskiBatchNorm.forward(x) without residual, ME.cat order."""
    import gcl_amd.MinkowskiEngine as ME
    import gcl_amd.MinkowskiEngine.MinkowskiFunctional as MEF
    C = _lidar_cloud(2, stride=6)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(len(C), 32, generator=g)
    s = ME.SparseTensor(x.to(DEV), coordinates=C.to(DEV))
    r = MEF.relu(s)
    assert torch.equal(r.F.cpu(), torch.relu(x)) and r.coordinate_manager is s.coordinate_manager
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, stride=1, dimension=3).to(DEV)
    bn = ME.MinkowskiBatchNorm(32, momentum=0.05).to(DEV)
    conv.train(), bn.train()
    y = conv(s)
    assert getattr(y, "_bn_stats", None) is not None          # the conv epilogue published column sums of y
    y += s                                                    # ... which no longer describe y's features
    assert y._bn_stats is None
    z = bn(y)
    ref = torch.nn.functional.batch_norm(y.F.detach().cpu().double(), None, None, bn.bn.weight.detach().cpu().double(),
                                         bn.bn.bias.detach().cpu().double(), True, 0.05, 1e-5)
    assert rel_l2(z.F.detach().cpu(), ref) < 2e-6
    c = ME.cat(r, s)
    assert torch.equal(c.F[:, :32], r.F) and torch.equal(c.F[:, 32:], s.F)


# ---------------------------------------------------------------------------------------------------------------
# 1b. demo.py:29 -- ResUNetBN2C(1, 16, normalize_feature=True, conv1_kernel_size=3): Cout = 16 head
# ---------------------------------------------------------------------------------------------------------------
def test_demo_model_16dim_vs_oracle():
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.model import load_model
    torch.manual_seed(0)
    m = load_model("ResUNetBN2C")(1, 16, normalize_feature=True, conv1_kernel_size=3, D=3).to(DEV)     # demo.py:29
    _randomise_bn(m)
    st = _state64(m)
    assert st["final.kernel"].shape == (64, 16)
    xyz = synthetic.make_box_cloud(0, 5000)
    for voxel in (0.3, 0.025):                       # configs[0] and demo.py:66-70's default voxel size
        coords, _ = ME.utils.sparse_quantize(xyz / voxel, return_index=True)
        C = ME.utils.batched_coordinates([coords])
        feats = torch.ones(len(C), 1)
        m.train()
        F = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F
        so = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
        Fo = O.resunet_forward(so, C.numpy(), feats.double(), 3, True, True, 0.1)
        assert F.shape == (len(C), 16) and rel_l2(F.detach().cpu(), Fo.detach()) < 1e-4
        gy = torch.randn(Fo.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
        Fo.backward(gy)
        m.zero_grad()
        F.backward(gy.float().to(DEV))
        for n, p in m.named_parameters():
            # a 5k-point cloud leaves a few dozen rows at tensor stride 8: BatchNorm gradients over so few rows amplify
            # fp32 rounding (fp64 oracle) -- 1e-2 for the BN parameters, 2e-3 for the convolution kernels
            assert rel_l2(p.grad.cpu(), so[n].grad) < (1e-2 if ".bn." in n else 2e-3), n
        m.eval()
        with torch.no_grad():
            Fe = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F
        Feo = O.resunet_forward({k: v.detach() for k, v in so.items()}, C.numpy(), feats.double(), 3, True, False, 0.1)
        assert rel_l2(Fe.cpu(), Feo) < 1e-4
        st = _state64(m)            # running statistics moved


@pytest.mark.parametrize("name", ["ResUNetBN2E", "ResUNetBN2B", "ResUNetBN2D"])
def test_other_width_variants_vs_oracle(name):
    """ResUNetBN2E has CHANNELS[1] = 128: the first-layer kernels take any Cout that is a multiple of 32."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.model import load_model
    torch.manual_seed(1)
    m = load_model(name)(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(DEV)
    _randomise_bn(m)
    st = _state64(m)
    C = _lidar_cloud(3, stride=2)          # enough rows at tensor stride 8 for well-conditioned 256-channel BatchNorms
    feats = torch.ones(len(C), 1)
    m.train()
    F = m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F
    so = {k: v.clone().requires_grad_("running" not in k) for k, v in st.items()}
    Fo = O.resunet_forward(so, C.numpy(), feats.double(), 5, True, True, 0.05)
    assert rel_l2(F.detach().cpu(), Fo.detach()) < 1e-4
    gy = torch.randn(Fo.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    Fo.backward(gy)
    F.backward(gy.float().to(DEV))
    for n, p in m.named_parameters():     # fp32 gradients of these width variants sit at 2e-3 .. 4e-3 of the fp64 oracle
        assert rel_l2(p.grad.cpu(), so[n].grad) < 1e-2, n      # (deep levels of a single small cloud: few rows per BatchNorm)


def test_in_variant_with_wide_stem_runs():
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.model import load_model
    m = load_model("ResUNetIN2E")(3, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3).to(DEV)
    C = _lidar_cloud(4, stride=6)
    F = m(ME.SparseTensor(torch.rand(len(C), 3).to(DEV), coordinates=C.to(DEV))).F
    F.sum().backward()
    assert torch.isfinite(F).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())


# ---------------------------------------------------------------------------------------------------------------
# ADVICE: caches and tags
# ---------------------------------------------------------------------------------------------------------------
def test_eval_affine_cache_follows_running_statistics():
    """eval -> train forward WITHOUT optimizer step -> eval: the fused conv+BN inference launch must see the new running
    statistics (they are updated through raw pointers, which does not bump tensor versions)."""
    import gcl_amd.MinkowskiEngine as ME
    C = _lidar_cloud(5, stride=6)
    x = torch.randn(len(C), 32, generator=torch.Generator().manual_seed(0)) * 3 + 1
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, stride=1, dimension=3).to(DEV)
    bn = ME.MinkowskiBatchNorm(32, momentum=0.5).to(DEV)

    def both():
        conv.eval(), bn.eval()
        with torch.no_grad():
            s = ME.SparseTensor(x.to(DEV), coordinates=C.to(DEV))
            return ME.conv_bn(conv, bn, s, relu=True).F.cpu(), torch.relu(bn(conv(s)).F).cpu()

    f0, u0 = both()
    assert rel_l2(f0, u0) < 2e-6
    conv.train(), bn.train()
    bn(conv(ME.SparseTensor(x.to(DEV), coordinates=C.to(DEV))))       # moves running_mean / running_var
    f1, u1 = both()
    assert rel_l2(u1, u0) > 1e-2, "the running statistics did change"
    assert rel_l2(f1, u1) < 2e-6, "fused inference path used stale statistics"


def test_weight_writes_through_data_keep_the_fp16x3_scale_right():
    """p.data.mul_(...) does not bump p._version: a training-mode forward re-measures max|W| anyway; in eval mode
    ME.invalidate_amax() is the documented way."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd.model import load_model
    torch.manual_seed(2)
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3).to(DEV)
    C = _lidar_cloud(6, stride=6)
    feats = torch.ones(len(C), 1)
    run = lambda: m(ME.SparseTensor(feats.to(DEV), coordinates=C.to(DEV))).F.detach().cpu()
    m.train()
    run()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.data.mul_(300.0)         # far outside the fp16 range of the old scale
    F = run()
    Fo = O.resunet_forward(_state64_train(m), C.numpy(), feats.double(), 3, True, True, 0.05)
    assert torch.isfinite(F).all() and rel_l2(F, Fo.detach()) < 1e-4
    m.eval()
    with torch.no_grad():
        run()
        for p in m.parameters():
            if p.dim() >= 2:
                p.data.mul_(1.0 / 300.0)
        ME.invalidate_amax()
        Fe = run()
    Feo = O.resunet_forward(_state64(m), C.numpy(), feats.double(), 3, True, False, 0.05)
    assert rel_l2(Fe, Feo) < 1e-4


def _state64_train(m):
    """State BEFORE the forward that is being checked is not available any more; training-mode features do not depend
    on the running statistics, so the current buffers do."""
    return _state64(m)


def test_tensor_on_another_device_is_rejected():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import gcl_amd.MinkowskiEngine as ME
    C = _lidar_cloud(7, stride=8)
    with torch.cuda.device(0):
        with pytest.raises(RuntimeError, match="current device"):
            ME.SparseTensor(torch.ones(len(C), 1, device="cuda:1"), coordinates=C.to("cuda:1"))


# ---------------------------------------------------------------------------------------------------------------
# a10: gradient accumulation (iter_size), a17: the eval loop
# ---------------------------------------------------------------------------------------------------------------
def test_iter_size_two_accumulates_like_the_reference():
    """iter_size = 2 (lib/colocation_trainer.py:838, :875-879, :887): each micro-batch's loss terms divided by 2, two
    backward passes, one SGD step -- against the oracle's gradients of the same two batches."""
    from gcl_amd import ddp, synthetic
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    batches = [synthetic.collate_train([synthetic.make_train_sample(s, num_neighborhood=1, n_boxes=6)]) for s in (21, 22)]
    rng = np.random.RandomState(0)
    draws = []
    for b in batches:
        N, Gn = len(b["sinput_C"]), len(b["group"])
        draws.append((rng.choice(Gn, min(Gn, 64), replace=False), rng.choice(N, 256, replace=False),
                      rng.choice(N, 256, replace=False)))
    cfg = make_config(batch_size=1, num_pos_per_batch=64, num_hn_samples_per_batch=256, iter_size=2)
    results = []
    for use_ddp in (False, True):
        torch.manual_seed(11)
        tr = FinestContrastiveLossTrainer(cfg, device=DEV, ddp=ddp.FlatDDP() if use_ddp else None)
        st0 = _state64(tr.model)
        p0 = {n: p.detach().cpu().double().clone() for n, p in tr.model.named_parameters()}
        loss, parts, n = tr.train_step(batches, draws=draws)
        grads = {n: p.grad.detach().cpu().clone() for n, p in tr.model.named_parameters()}
        results.append((loss.item(), [p.item() for p in parts], grads,
                        {n: p.detach().cpu().clone() for n, p in tr.model.named_parameters()}))
    assert n == sum(len(b["sinput_C"]) for b in batches)
    so = {k: v.clone().requires_grad_("running" not in k) for k, v in st0.items()}
    total = 0.0
    ref_parts = np.zeros(3)
    for b, d in zip(batches, draws):
        Fo = O.resunet_forward(so, b["sinput_C"].numpy(), b["sinput_F"].double(), 5, True, True, 0.05)
        terms = LO.finest_contrastive_loss(Fo, b["group"].numpy(), b["index"].numpy(), b["index_hash"],
                                           b["finest_flag"].numpy(), draws=d)
        l = sum(t / 2 for t in terms)
        l.backward()
        total += l.item()
        ref_parts += np.array([t.item() / 2 for t in terms])
    for loss_v, parts_v, grads, params in results:
        assert abs(loss_v - total) < 2e-4 * max(1.0, abs(total)) and np.allclose(parts_v, ref_parts, rtol=2e-4, atol=2e-5)
        for name, gr in grads.items():
            assert rel_l2(gr, so[name].grad) < 5e-3, name
    # one SGD step (first step: momentum buffer = gradient): p1 = p0 - lr * (g + wd * p0)
    name = "block2_tr.conv2.kernel"
    want = p0[name] - 0.1 * (so[name].grad + 1e-4 * p0[name])
    assert rel_l2(results[0][3][name], want) < 1e-5


def _twin_pair(seed, shift_voxels=(8, 0, 0), voxel=0.3):
    """An eval pair whose second cloud is the first one moved by a multiple of 8 voxels (every U-Net level stays aligned):
    well-conditioned for an UNTRAINED network -- twin voxels get equal features, everything else is an outlier.  Input
    features carry a small per-voxel jitter so that no two voxels have tied features."""
    from gcl_amd import synthetic
    p = synthetic.make_eval_pair(seed, voxel_size=voxel, baseline=6.0, n_boxes=25)
    keep = torch.arange(0, len(p["sinput0_C"]), 3)                      # ~1/3 of the voxels: a few thousand rows
    C0 = p["sinput0_C"][keep].clone()
    xyz0 = p["pcd0"][0][keep].clone()
    sh = torch.tensor(shift_voxels, dtype=torch.int32)
    C1 = C0.clone()
    C1[:, 1:] += sh
    xyz1 = xyz0 + sh.float() * voxel
    F = 1.0 + 0.05 * torch.randn(len(C0), 1, generator=torch.Generator().manual_seed(seed))
    T = torch.eye(4)
    T[:3, 3] = sh.float() * voxel
    return {"pcd0": (xyz0,), "pcd1": (xyz1,), "sinput0_C": C0, "sinput1_C": C1, "sinput0_F": F, "sinput1_F": F.clone(),
            "T_gt": T}


def test_valid_epoch_matches_the_oracle_chain():
    """``_valid_epoch`` (lib/colocation_trainer.py:306-379): meters of the packaged validation step against the CPU chain
    me_oracle.resunet_forward -> loss_oracle.find_nn -> transform_oracle.est_quad_linear_robust on the same seeded
    draws (find_corr's two np.random.choice calls per pair, :385-388)."""
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    from oracle import transform_oracle as TO
    torch.manual_seed(5)
    tr = FinestContrastiveLossTrainer(make_config(hit_ratio_thresh=0.3), device=DEV)
    m = tr.model
    m.eval()
    st = {k: v.detach().cpu().double() for k, v in m.state_dict().items() if "num_batches" not in k}
    # small motions (the estimator is a local one): twins shifted by 8 voxels = 2.4 m would be outside its basin, so the
    # second cloud is the first one with the SAME voxel grid and the metric points moved by a small rigid motion
    pairs = []
    for seed, ang, t in [(70, 0.03, (0.25, -0.1, 0.05)), (71, -0.05, (-0.2, 0.15, 0.0))]:
        d = _twin_pair(seed, (0, 0, 0))
        c, s_ = np.cos(ang), np.sin(ang)
        T = torch.eye(4)
        T[:3, :3] = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
        T[:3, 3] = torch.tensor(t)
        d["pcd1"] = (d["pcd0"][0] @ T[:3, :3].t() + T[:3, 3],)
        d["T_gt"] = T
        pairs.append(d)
    np.random.seed(21)
    out = tr._valid_epoch(pairs)
    assert out["n"] == 2 and not m.training
    np.random.seed(21)
    loss, rte, rre, hit = [], [], [], []
    for d in pairs:
        Fs = [O.resunet_forward(st, d[f"sinput{k}_C"].numpy(), d[f"sinput{k}_F"].double(), 5, True, False, 0.05).float()
              for k in (0, 1)]
        xyz0, xyz1 = d["pcd0"][0].numpy(), d["pcd1"][0].numpy()
        n0, n1 = len(Fs[0]), len(Fs[1])
        if n0 > 5000:
            i0 = np.random.choice(n0, min(n0, 5000), replace=False)
            i1 = np.random.choice(n1, min(n1, 5000), replace=False)
        else:
            i0, i1 = np.arange(n0), np.arange(n1)
        nn = LO.find_nn(Fs[0][i0], Fs[1][i1], nn_max_n=-1).numpy()
        a, b = xyz0[i0], xyz1[i1[nn]]
        T_o = TO.est_quad_linear_robust(a, b)
        Tg = d["T_gt"].numpy()
        loss.append(TO.corr_dist(T_o, Tg, xyz0))
        rte.append(float(np.linalg.norm(T_o[:3, 3] - Tg[:3, 3])))
        with np.errstate(invalid="ignore"):
            r = float(np.arccos((np.trace(T_o[:3, :3].T @ Tg[:3, :3]) - 1) / 2))
        if not np.isnan(r):                                   # the reference's meter skips NaN (:349-351)
            rre.append(r)
        hit.append(TO.hit_ratio(a, b, Tg, 0.3))
    assert abs(out["loss"] - np.mean(loss)) < 2e-3 and abs(out["rte"] - np.mean(rte)) < 2e-3
    assert abs(out["hit_ratio"] - np.mean(hit)) < 2e-3 and out["feat_match_ratio"] == 1.0
    assert abs(out["rre"] - (np.mean(rre) if rre else 0.0)) < 4e-3
    assert out["rte"] < 0.05 and out["hit_ratio"] > 0.9          # an untrained net on twin clouds: the motion is found


def test_eval_pairs_matches_the_oracle_chain_and_batching_is_bitwise_neutral():
    """scripts/test_kitti.py:129-227 as gcl_amd.scripts.test_kitti.eval_pairs: composed T of every pair against the CPU
    chain me_oracle.resunet_forward -> loss_oracle.find_nn -> sc2pcr_oracle.sc2_pcr on the same seeded draws (same
    np.random call order as the reference's loop), RTE / RRE / success meters, and the B-pairs-per-forward mode
    returning the same transformations bit for bit."""
    from gcl_amd.model import load_model
    from gcl_amd.scripts.SC2_PCR import Matcher
    from gcl_amd.scripts.test_kitti import eval_pairs
    from oracle.sc2pcr_oracle import sc2_pcr
    torch.manual_seed(5)
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(DEV)
    _randomise_bn(m)
    m.eval()
    st = _state64(m)
    pairs = [_twin_pair(60, (8, 0, 0)), _twin_pair(61, (-8, 16, 0)), _twin_pair(62, (16, 8, 8))]
    cfgm = dict(inlier_threshold=0.6, num_node=1500, use_mutual=False, d_thre=0.1, num_iterations=20, ratio=0.2,
                nms_radius=0.6, max_points=8000, k1=30, k2=20)
    NPTS = 1500
    assert all(len(p["sinput0_C"]) > NPTS for p in pairs)
    np.random.seed(9)
    r1 = eval_pairs(m, pairs, Matcher(**cfgm), device=DEV, batch_pairs=1, subsample_size=NPTS, n_points=NPTS, collect=True)
    np.random.seed(9)
    r3 = eval_pairs(m, pairs, Matcher(**cfgm), device=DEV, batch_pairs=3, subsample_size=NPTS, n_points=NPTS)
    assert r1["n_pairs"] == 3 and len(r1["T_est"]) == 3 and len(r1["dists_nn"][0]) == NPTS
    for a, b in zip(r1["T_est"], r3["T_est"]):
        assert torch.equal(a, b), "batched forward must not change a single bit"
    assert r1["rte"] == r3["rte"] and r1["success"] == r3["success"]
    # the oracle chain with the same draws
    np.random.seed(9)
    for i, d in enumerate(pairs):
        Fs = [O.resunet_forward(st, d[f"sinput{k}_C"].numpy(), d[f"sinput{k}_F"].double(), 5, True, False, 0.05).float()
              for k in (0, 1)]
        xyz = [d["pcd0"][0].numpy(), d["pcd1"][0].numpy()]
        for k in (0, 1):                                  # find_corr's two draws (:33-34)
            np.random.choice(len(Fs[k]), min(len(Fs[k]), NPTS), replace=False)
        sel = [np.random.permutation(len(xyz[k]))[:NPTS] for k in (0, 1)]     # random_sample (:160-161), n1 > N
        x0, x1, F0, F1 = xyz[0][sel[0]], xyz[1][sel[1]], Fs[0][sel[0]], Fs[1][sel[1]]
        s_sel = np.random.choice(NPTS, cfgm["num_node"])                    # Matcher.match_pair (:289-290)
        t_sel = np.random.choice(NPTS, cfgm["num_node"])
        nn = LO.find_nn(F0[s_sel], F1[t_sel], nn_max_n=-1).numpy()
        T_o = sc2_pcr(x0[s_sel], x1[t_sel[nn]], inlier_threshold=0.6, d_thre=0.1, num_iterations=20, ratio=0.2,
                      nms_radius=0.6, max_points=8000, k1=30, k2=20)
        T_o = torch.as_tensor(T_o).float().reshape(4, 4)
        assert (r1["T_est"][i] - T_o).abs().max().item() < 2e-3, (i, r1["T_est"][i], T_o)
        assert (T_o - d["T_gt"]).abs().max().item() < 5e-2            # and both found the true motion
    assert r1["success_rate"] == 1.0 and r1["rte_avg"] < 0.05 and r1["rre_avg"] < 0.5
    assert r1["success"] == [True, True, True] and r1["n_voxels"] == sum(2 * len(p["sinput0_C"]) for p in pairs)


# ---------------------------------------------------------------------------------------------------------------
# configs[2] at the size BASELINE quotes: the full bs = 4 x 7-cloud batch against the committed fp64-oracle fixture
# ---------------------------------------------------------------------------------------------------------------
def test_full_size_batch_against_oracle_fixture():
    """tests/golden/full_bs4_sample.npz (tests/golden/make_full_fixture.py: fp64 CPU oracle on the benchmark batch, seed
    100, 530 321 voxels): 4096 sampled feature rows and the per-column sums within 1e-4 rel-L2, the loss triple for the
    fixture's draws within 2e-4."""
    import gcl_amd.MinkowskiEngine as ME
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_trainer import finest_contrastive_loss
    from gcl_amd.model import load_model
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "full_bs4_sample.npz"))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_full_fixture import BS, MODE, SEED, fixed_draws
    batch = synthetic.make_train_batch(SEED, batch_size=BS, group_mode=MODE)
    C = batch["sinput_C"]
    assert len(C) == int(z["n_voxels"]) and len(batch["group"]) == int(z["n_groups"])
    assert int(C.numpy().astype(np.int64).sum()) == int(z["coord_checksum"]), "the generator is not reproducible here"
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    st = O.random_state(0, dtype=torch.float32)
    missing = m.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys and all("num_batches" in k for k in missing.missing_keys)
    m = m.to(DEV)
    m.train()
    with torch.no_grad():
        F = m(ME.SparseTensor(batch["sinput_F"].to(DEV), coordinates=C.to(DEV))).F
    rows = torch.from_numpy(z["rows"]).to(DEV)
    err = rel_l2(F[rows].cpu(), z["feats"])
    err_sum = rel_l2(F.double().sum(0).cpu(), z["col_sum"])
    print(f"full-size batch: sampled-row rel-L2 {err:.3e}, column-sum rel-L2 {err_sum:.3e}")
    assert err < 1e-4 and err_sum < 1e-4
    assert rel_l2(F.double().abs().sum(0).cpu(), z["col_abs_sum"]) < 1e-5
    draws = fixed_draws(len(batch["group"]), len(C))
    pos, fin, neg = finest_contrastive_loss(F, batch["group"], batch["index"], batch["index_hash"], batch["finest_flag"],
                                            max_pos_cluster=1024, max_hn_samples=1024, draws=draws)
    got = np.array([pos.item(), fin.item(), neg.item()])
    assert np.allclose(got, z["loss"], rtol=2e-4, atol=2e-5), (got, z["loss"])
    # running statistics after this one training-mode forward (momentum 0.05)
    assert rel_l2(m.norm1.bn.running_mean.cpu(), z["running_mean_norm1"]) < 1e-4
    assert rel_l2(m.block4.norm2.bn.running_var.cpu(), z["running_var_block4"]) < 1e-4


# Bounds of the full-size gradient test = 2 x the larger of the two default arithmetics' measured worst (norm error,
# sampled-entry error) per parameter class -- profiles/r05_precision_errors_full_bs4.log: kernels 1.8e-5 / 1.5e-3 (exact f32;
# fp16x3 1.8e-5 / 1.1e-3), BatchNorm parameters 2.3e-4 / 2.0e-3 (fp16x3; f32 1.4e-4 / 1.0e-3).  bf16x6 (GCL_FULL_BWD_PRECISIONS)
# measured 2.7e-5 / 5.0e-3 and 2.0e-4 / 2.4e-3 and needs the wider pair.
FULL_BWD_BOUNDS = {"kernel": (4e-5, 3e-3), "bn": (5e-4, 4e-3)}
FULL_BWD_BOUNDS_BF16 = {"kernel": (6e-5, 1e-2), "bn": (5e-4, 5e-3)}


@pytest.mark.parametrize("precision", os.environ.get("GCL_FULL_BWD_PRECISIONS", "fp16x3,f32").split(","))
def test_full_size_training_step_gradients_against_oracle_fixture(precision):
    """tests/golden/full_bs4_backward.npz (make_full_fixture.py --backward: ONE fp64 oracle training step on the 530 321-voxel
    benchmark batch -- loss = pos + finest + neg, lib/colocation_trainer.py:875-887): the gradients the trainer's
    ``train_step`` leaves in its seats on the SECOND step -- per-parameter norm and 256 sampled entries per tensor, and the
    loss triple.  lr = 0, so both steps see the fixture's parameters.  Two arithmetics: the default ``fp16x3`` (second step
    through the native plan with the weight gradients on the aux stream and the range-grouped launches, >= 32768 rows) and
    the exact-f32 MFMA kernels as the CONTROL: the encoder's error growth must be the same in both, i.e. fp32 rounding
    amplified through 21 batch-statistics BatchNorms, not the split arithmetic's."""
    import gcl_amd.MinkowskiEngine as ME  # noqa: F401
    from conftest import precision_log_path
    from gcl_amd import synthetic
    from gcl_amd.MinkowskiEngine import native, ops
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config
    path = os.path.join(os.path.dirname(__file__), "golden", "full_bs4_backward.npz")
    z = np.load(path)
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_full_fixture import BS, MODE, SEED, fixed_draws
    batch = synthetic.make_train_batch(SEED, batch_size=BS, group_mode=MODE)
    assert len(batch["sinput_C"]) == int(z["n_voxels"])
    cfg = make_config(batch_size=BS, lr=0.0, momentum=0.0, weight_decay=0.0)
    old = ops.PRECISION
    ME.set_conv_precision(precision)
    try:
        with torch.cuda.device(DEV):
            tr = FinestContrastiveLossTrainer(cfg, device=torch.device(DEV))
            st = O.random_state(0, dtype=torch.float32)
            missing = tr.model.load_state_dict(st, strict=False)
            assert not missing.unexpected_keys and all("num_batches" in k for k in missing.missing_keys)
            draws = fixed_draws(len(batch["group"]), len(batch["sinput_C"]))
            keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
            dev_batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k in keys}
            for step in range(2):          # step 0 is recorded by the Tape, step 1 runs through the plan (fp16x3)
                b = dict(dev_batch)       # native maps in both steps: the Tape pass on them is what records the plan
                b["_coordinate_manager"] = ME.CoordinateManager.build_native(dev_batch["sinput_C"], tr.model.native_map_specs())
                loss, (pos, fin, neg), n = tr.train_step(b, draws=draws)
                torch.cuda.synchronize()
                got = np.array([pos.item(), fin.item(), neg.item()])
                assert np.allclose(got, z["loss"], rtol=2e-4, atol=2e-5), (step, got, z["loss"])
            if precision == "fp16x3":
                assert isinstance(tr.model.__dict__.get("_plan"), native.NetworkPlan)
                assert native.AUX_STREAM and tr.model._plan._aux is not None, "the plan's second pass must have used the aux stream"
            params = dict(tr.model.named_parameters())
            worst = {}
            for j, name in enumerate(z["names"]):
                g = params[str(name)].grad.detach().double().cpu().reshape(-1).numpy()
                ref_norm, idx, val = float(z[f"norm_{j}"]), z[f"idx_{j}"], z[f"val_{j}"]
                e_norm = abs(np.linalg.norm(g) - ref_norm) / max(ref_norm, 1e-30)
                e_val = np.linalg.norm(g[idx] - val) / max(np.linalg.norm(val), 1e-30)
                worst[str(name)] = (e_norm, e_val)
    finally:
        ME.set_conv_precision(old)
    log = precision_log_path()
    with open(log, "a") as fh:
        for name, (e_norm, e_val) in worst.items():
            fh.write(f"full_bs4_backward {precision} {name} norm_err={e_norm:.3e} sampled_err={e_val:.3e}\n")
    top = sorted(worst.items(), key=lambda kv: -max(kv[1]))[:5]
    print(f"full-size backward [{precision}]: worst parameters (norm err, sampled-entry err):", top)
    for name, (e_norm, e_val) in worst.items():
        b_norm, b_val = (FULL_BWD_BOUNDS_BF16 if precision.startswith("bf16") else FULL_BWD_BOUNDS)["bn" if ".bn." in name else "kernel"]
        assert e_norm < b_norm and e_val < b_val, (precision, name, e_norm, e_val, f"all parameters: {log}")


def test_prefetch_staging_slots_are_reused_only_after_release():
    """prefetch_to_device: device staging slots are handed out again only after release_batch (an event on the compute
    stream the copy stream waits for); a consumer that never releases gets one slot per batch (the pool grows, a pending
    batch is never overwritten: ADVICE round 2).  Every yielded batch holds its source's values when consumed."""
    from gcl_amd.lib.colocation_trainer import prefetch_to_device, release_batch, wait_for_batch
    g = torch.Generator().manual_seed(0)
    host = [{"sinput_C": torch.randint(0, 100, (1000 - 7 * i, 4), generator=g, dtype=torch.int32),     # never growing:
             "sinput_F": torch.randn(1000 - 7 * i, 1, generator=g), "tag": i} for i in range(12)]      # slots keep their buffers
    for release in (True, False):
        ptrs = set()
        with torch.cuda.device(DEV):
            for i, b in enumerate(prefetch_to_device(iter(host), DEV, keys=("sinput_C", "sinput_F"), ring=2)):
                wait_for_batch(b)
                assert b["tag"] == i and b["sinput_C"].is_cuda
                x = b["sinput_F"] * 2.0                      # work on the compute stream that reads the slot
                assert torch.equal(b["sinput_C"].cpu(), host[i]["sinput_C"])
                assert torch.equal(x.cpu(), host[i]["sinput_F"] * 2.0)
                ptrs.add(b["sinput_F"].data_ptr())
                if release:
                    release_batch(b)
        assert len(ptrs) <= (2 if release else 12), (release, len(ptrs))
        if not release:
            assert len(ptrs) == 12                           # one slot per pending batch, none recycled


def test_map_prefetch_on_side_stream_changes_nothing():
    """train_steps builds the coordinate manager of batch i+1 (maps, sorted tables, pair lists) on a side stream while
    batch i trains, from host batches copied by prefetch_to_device: losses and parameters equal the lazy path bit for
    bit."""
    from gcl_amd import synthetic
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer, make_config, prefetch_to_device
    batches = [synthetic.collate_train([synthetic.make_train_sample(s, num_neighborhood=2, n_boxes=10)]) for s in (31, 32, 33)]
    keys = ("sinput_C", "sinput_F", "group", "index", "finest_flag")
    host = [{k: v for k, v in b.items() if k in keys} for b in batches]
    # lr = 0: the parameters never move, so every step is a deterministic function of its batch and draws (the only
    # order-dependent arithmetic of a step, the float atomics of the loss backward, cannot feed back into later steps)
    cfg = make_config(batch_size=1, num_pos_per_batch=64, num_hn_samples_per_batch=128, lr=0.0, weight_decay=0.0)
    runs = []
    for prefetch in (False, True):
        torch.manual_seed(3)
        np.random.seed(3)
        tr = FinestContrastiveLossTrainer(cfg, device=DEV)
        tr.map_prefetch = prefetch
        seq = [host[i % 3] for i in range(5)]
        losses = [l.item() for l, _, _ in tr.train_steps(prefetch_to_device(seq, DEV, keys))]
        torch.cuda.synchronize()
        runs.append((losses, torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()))
    assert runs[0][0] == runs[1][0], "identical losses, step by step (steps 2.. run on maps built on the side stream)"
    assert torch.equal(runs[0][1], runs[1][1]) and len(set(runs[0][0])) >= 3


def test_fused_sgd_equals_torch_sgd():
    """gcl_amd.lib.optim.FusedSGD (one launch, gcl_sgd_multi) against torch.optim.SGD with the reference's settings
    (lr 0.1, momentum 0.8, weight decay 1e-4; lib/colocation_trainer.py:73-77) over three steps with an ExponentialLR."""
    from gcl_amd.lib.optim import FusedSGD
    g = torch.Generator().manual_seed(0)
    shapes = [(27, 64, 64), (64,), (1, 32), (125, 1, 32), (7,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    res = []
    for cls in (torch.optim.SGD, FusedSGD):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        opt = cls(ps, lr=0.1, momentum=0.8, weight_decay=1e-4)
        sch = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
        gg = torch.Generator().manual_seed(1)
        for _ in range(3):
            for p in ps:
                p.grad = torch.randn(p.shape, generator=gg).to(DEV)
            opt.step()
            sch.step()
        res.append(([p.detach().cpu() for p in ps], [opt.state[p]["momentum_buffer"].cpu() for p in ps]))
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-6)        # fma contraction differs in the last bit
