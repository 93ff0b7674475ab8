"""CPU ORACLE (test infrastructure, NOT product code) -- sparse-convolution half of the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package ``gcl_amd`` never does (it fails loudly without the HIP library).

What is restated here
---------------------
The arithmetic of this half lives in a THIRD-PARTY dependency that is absent from /root/reference:
``MinkowskiEngine`` (requirements.txt:8, un-pinned; README.md:77 "v0.5 or higher", README.md:130 installs
git master).  The reference holds no golden vector, checkpoint or test for it  =>  **parity unpinned**
against ME itself.  The operator semantics follow ME 0.5.x's published "generalized sparse convolution"
(Choy et al., CVPR'19, eq. 3) and are pinned instead by an INDEPENDENT dense oracle
(``dense_conv_reference`` below: ``torch.nn.functional.conv3d / conv_transpose3d`` sampled at the active
voxels; checked in tests/test_oracle_conv.py).  Topology follows the reference's own files:

* ``resunet_forward``   -- model/resunet.py:173-232 (ResUNet2.forward), widths :245-248 (ResUNetBN2C)
* ``basic_block``       -- model/residual_block.py:37-53
* ``batch_norm``        -- model/common.py:4-6 -> ME.MinkowskiBatchNorm == BatchNorm1d over rows

Operator semantics:
* coordinates int32 [N, 4] = (batch, x, y, z), unique; row order preserved at tensor-stride 1;
* strided conv: out coords = unique(floor(c / t_out) * t_out), floor toward -inf, rows in order of
  first occurrence in the input map (ME's own order is hash-order, i.e. unspecified; every comparison of
  maps is done on canonical sorted (k, in, out) triples);
* kernel offsets: k -> (ox, oy, oz) with x fastest, each in -(ks//2) .. +(ks//2), scaled by the INPUT
  tensor stride (x dilation); region centred on the output coordinate: out[v] = sum_k x[u(v + o_k)] W_k;
* transposed conv (stride 2): onto the existing finer map; kernel map = the fine->coarse forward map with
  in/out swapped and the same k.
"""
import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------
# integer part: coordinate maps / kernel maps
# --------------------------------------------------------------------------------------------------


def kernel_offsets(ks):
    """[K,3] int offsets, x fastest then y then z (last paragraph)."""
    r = ks // 2
    offs = []
    for k in range(ks ** 3):
        offs.append((k % ks - r, (k // ks) % ks - r, k // (ks * ks) - r))
    return np.asarray(offs, dtype=np.int64)


def stride_coords(C, t_out):
    """Unique floor(c / t_out) * t_out, in order of first occurrence (numpy, exact integers)."""
    C = np.asarray(C, dtype=np.int64)
    D = C.copy()
    D[:, 1:] = np.floor_divide(C[:, 1:], t_out) * t_out
    key = pack_keys(D)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return D[first].astype(np.int32)


def pack_keys(C):
    C = np.asarray(C, dtype=np.int64)
    off = 1 << 15
    assert C[:, 0].min(initial=0) >= 0 and C[:, 0].max(initial=0) < (1 << 15)
    assert C[:, 1:].min(initial=0) >= -off and C[:, 1:].max(initial=0) < off
    return (C[:, 0] << 48) | ((C[:, 1] + off) << 32) | ((C[:, 2] + off) << 16) | (C[:, 3] + off)


def kernel_map_dict(C_in, C_out, ks, t_in, dilation=1):
    """Obviously-correct pure-Python kernel map: list of (k, in_row, out_row) triples."""
    table = {tuple(int(v) for v in c): i for i, c in enumerate(np.asarray(C_in))}
    offs = kernel_offsets(ks) * (t_in * dilation)
    triples = []
    for v, c in enumerate(np.asarray(C_out)):
        for k, o in enumerate(offs):
            u = table.get((int(c[0]), int(c[1] + o[0]), int(c[2] + o[1]), int(c[3] + o[2])))
            if u is not None:
                triples.append((k, u, v))
    return np.asarray(triples, dtype=np.int64).reshape(-1, 3)


def kernel_map_np(C_in, C_out, ks, t_in, dilation=1):
    """Vectorised kernel map (sort + searchsorted); same triples as ``kernel_map_dict``, (k, out)-sorted."""
    C_in = np.asarray(C_in, dtype=np.int64)
    C_out = np.asarray(C_out, dtype=np.int64)
    kin = pack_keys(C_in)
    order = np.argsort(kin, kind="stable")
    skeys = kin[order]
    offs = kernel_offsets(ks) * (t_in * dilation)
    out = []
    for k, o in enumerate(offs):
        Q = C_out.copy()
        Q[:, 1:] += o
        ok = ((Q[:, 1:].min(axis=1, initial=0) >= -(1 << 15)) & (Q[:, 1:].max(axis=1, initial=0) < (1 << 15))) \
            if len(Q) else np.zeros(0, bool)     # the packable range is [-32768, 32767]
        q = pack_keys(np.where(ok[:, None], Q, 0)) if len(Q) else np.zeros(0, np.int64)
        pos = np.searchsorted(skeys, q)
        pos = np.minimum(pos, len(skeys) - 1) if len(skeys) else pos
        hit = ok & (skeys[pos] == q) if len(skeys) else np.zeros(len(q), bool)
        v = np.nonzero(hit)[0]
        u = order[pos[v]]
        out.append(np.stack([np.full(len(v), k, np.int64), u, v], axis=1))
    return np.concatenate(out) if out else np.zeros((0, 3), np.int64)


def canonical(triples):
    """Sort (k, in, out) triples lexicographically -- the form in which maps are compared bit-exactly."""
    t = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


class CoordinateManager:
    """Caches coordinate maps per tensor stride and kernel maps per (t_in, ks, stride, transpose)."""

    def __init__(self, C):
        self.coords = {1: np.asarray(C, dtype=np.int32)}
        self.kmaps = {}

    def get_coords(self, t):
        if t not in self.coords:
            base = self.coords[max(s for s in self.coords if s < t)]
            self.coords[t] = stride_coords(base, t)
        return self.coords[t]

    def get_kernel_map(self, t_in, ks, stride):
        """Forward (in stride t_in -> out stride t_in*stride) map as (k, in, out) triples."""
        key = (t_in, ks, stride)
        if key not in self.kmaps:
            self.kmaps[key] = kernel_map_np(self.get_coords(t_in), self.get_coords(t_in * stride), ks, t_in)
        return self.kmaps[key]


# --------------------------------------------------------------------------------------------------
# floating-point part: operators (torch CPU autograd gives the backward)
# --------------------------------------------------------------------------------------------------


def sparse_conv(x, W, triples, n_out, transpose=False, bias=None):
    """out[v] = sum_k x[u] @ W[k] over the triples (gather -> GEMM -> scatter-add per offset).

    ``transpose`` swaps the roles of the in/out columns (transposed convolution re-uses the forward map).
    W: [K, Cin, Cout] (or [Cin, Cout] when K == 1).
    """
    if hasattr(triples, "conv"):        # a kernel map of oracle/me_cpu.py: the C / OpenMP restatement does the work
        return triples.conv(x, W, n_out, transpose, bias)
    Wk = W if W.dim() == 3 else W.unsqueeze(0)
    out = x.new_zeros((n_out, Wk.shape[2]))
    tn = np.asarray(triples)
    if len(tn) and (np.diff(tn[:, 0]) < 0).any():
        tn = tn[np.argsort(tn[:, 0], kind="stable")]
    bounds = np.searchsorted(tn[:, 0], np.arange(Wk.shape[0] + 1))      # triples are grouped by k
    t = torch.as_tensor(tn)
    ci, co = (2, 1) if transpose else (1, 2)
    for k in range(Wk.shape[0]):
        a, b = int(bounds[k]), int(bounds[k + 1])
        if b > a:
            out = out.index_add(0, t[a:b, co], x[t[a:b, ci]] @ Wk[k])
    if bias is not None:
        out = out + bias
    return out


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum, eps=1e-5):
    """BatchNorm1d over the rows of [N, C] (statistics over ALL voxels of ALL clouds)."""
    return F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)


def basic_block(x, st, prefix, mgr, t, training, momentum):
    """model/residual_block.py:37-53: conv3-BN-relu-conv3-BN-(+x)-relu, both convs on one kernel map."""
    tri = mgr.get_kernel_map(t, 3, 1)
    n = len(mgr.get_coords(t))
    out = sparse_conv(x, st[prefix + ".conv1.kernel"], tri, n)
    out = _bn(out, st, prefix + ".norm1", training, momentum)
    out = torch.relu(out)
    out = sparse_conv(out, st[prefix + ".conv2.kernel"], tri, n)
    out = _bn(out, st, prefix + ".norm2", training, momentum)
    out = out + x
    return torch.relu(out)


def _bn(x, st, name, training, momentum):
    return batch_norm(x, st[name + ".bn.weight"], st[name + ".bn.bias"],
                      st[name + ".bn.running_mean"], st[name + ".bn.running_var"], training, momentum)


def resunet_forward(st, C, feats, conv1_kernel_size=5, normalize_feature=True, training=True,
                    bn_momentum=0.05, mgr=None, taps=None):
    """ResUNet2.forward (model/resunet.py:173-232) for the BN variants without the conv1_extra branch.

    ``st``: dict name -> tensor with the reference's parameter names (conv1.kernel, norm1.bn.weight,
    block1.conv1.kernel, ..., final.kernel, final.bias).  Running statistics are updated in place when
    ``training`` (as BatchNorm1d does).  ``taps`` (dict) receives intermediate activations when given.
    """
    mgr = mgr or CoordinateManager(np.asarray(C))
    n1, n2, n4, n8 = (len(mgr.get_coords(t)) for t in (1, 2, 4, 8))

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    def down(x, name, norm, t):
        y = sparse_conv(x, st[name + ".kernel"], mgr.get_kernel_map(t, 3, 2), len(mgr.get_coords(2 * t)))
        return _bn(y, st, norm, training, bn_momentum)

    def up(x, name, norm, t_fine):
        y = sparse_conv(x, st[name + ".kernel"], mgr.get_kernel_map(t_fine, 3, 2),
                        len(mgr.get_coords(t_fine)), transpose=True)
        return _bn(y, st, norm, training, bn_momentum)

    out_s1 = sparse_conv(feats, st["conv1.kernel"], mgr.get_kernel_map(1, conv1_kernel_size, 1), n1)
    out_s1 = tap("norm1", _bn(tap("conv1", out_s1), st, "norm1", training, bn_momentum))
    out_s1 = tap("block1", basic_block(out_s1, st, "block1", mgr, 1, training, bn_momentum))
    out = torch.relu(out_s1)

    out_s2 = down(out, "conv2", "norm2", 1)
    out_s2 = tap("block2", basic_block(out_s2, st, "block2", mgr, 2, training, bn_momentum))
    out = torch.relu(out_s2)

    out_s4 = down(out, "conv3", "norm3", 2)
    out_s4 = tap("block3", basic_block(out_s4, st, "block3", mgr, 4, training, bn_momentum))
    out = torch.relu(out_s4)

    out_s8 = down(out, "conv4", "norm4", 4)
    out_s8 = tap("block4", basic_block(out_s8, st, "block4", mgr, 8, training, bn_momentum))
    out = torch.relu(out_s8)

    out = up(out, "conv4_tr", "norm4_tr", 4)
    out = basic_block(out, st, "block4_tr", mgr, 4, training, bn_momentum)
    out_s4_tr = tap("block4_tr", torch.relu(out))
    out = torch.cat([out_s4_tr, out_s4], dim=1)

    out = up(out, "conv3_tr", "norm3_tr", 2)
    out = basic_block(out, st, "block3_tr", mgr, 2, training, bn_momentum)
    out_s2_tr = tap("block3_tr", torch.relu(out))
    out = torch.cat([out_s2_tr, out_s2], dim=1)

    out = up(out, "conv2_tr", "norm2_tr", 1)
    out = basic_block(out, st, "block2_tr", mgr, 1, training, bn_momentum)
    out_s1_tr = tap("block2_tr", torch.relu(out))
    out = torch.cat([out_s1_tr, out_s1], dim=1)

    out = tap("conv1_tr", out @ st["conv1_tr.kernel"])            # kernel_size 1: plain GEMM, no bias
    out = torch.relu(out)
    out = tap("final", out @ st["final.kernel"] + st["final.bias"])
    if normalize_feature:
        out = out / torch.norm(out, p=2, dim=1, keepdim=True)
    return out


# --------------------------------------------------------------------------------------------------
# the independent dense oracle that pins the sparse semantics
# --------------------------------------------------------------------------------------------------


def dense_conv_reference(C_in, x, W, ks, t_in, stride=1, transpose=False, C_out=None):
    """Same operator through torch's DENSE conv3d / conv_transpose3d on a zero-filled grid.

    Single batch.  Sparse conv == dense conv of the zero-embedded input, sampled at the output's active
    coordinates (missing neighbours contribute zero either way).
    Dense weights: conv3d Wd[Cout, Cin, x, y, z] = W_k^T ; conv_transpose3d Wd[Cin, Cout, x, y, z] = W_k
    with k = x + ks*y + ks^2*z (grid dims ordered x, y, z).
    """
    C_in = np.asarray(C_in, dtype=np.int64)
    C_out = np.asarray(C_out, dtype=np.int64)
    assert (C_in[:, 0] == 0).all() and (C_out[:, 0] == 0).all()
    Cin, Cout = W.shape[-2], W.shape[-1]
    Wk = W.reshape(ks, ks, ks, Cin, Cout)            # [z, y, x, Cin, Cout] because x is fastest in k
    Wk = Wk.permute(2, 1, 0, 3, 4)                   # [x, y, z, Cin, Cout]
    if not transpose:
        t_out = t_in * stride
        lo = (np.minimum(C_in[:, 1:].min(0), C_out[:, 1:].min(0)) // t_out) * t_out - t_out * ks
        gi = (C_in[:, 1:] - lo) // t_in
        size = gi.max(0) + 1 + 2 * ks
        size = size + (-size) % 2
        grid = x.new_zeros((1, Cin, *size))
        grid[0, :, gi[:, 0], gi[:, 1], gi[:, 2]] = x.t()
        Wd = Wk.permute(4, 3, 0, 1, 2).contiguous()
        y = F.conv3d(grid, Wd, stride=stride, padding=ks // 2)
        go = (C_out[:, 1:] - lo) // t_out
        return y[0, :, go[:, 0], go[:, 1], go[:, 2]].t()
    # transposed: input lives on the coarse grid (stride 2*t_fine), output on the fine grid (t_fine)
    t_fine = t_in // stride
    lo = (np.minimum(C_in[:, 1:].min(0), C_out[:, 1:].min(0)) // t_in) * t_in - t_in * ks
    gi = (C_in[:, 1:] - lo) // t_in
    size = gi.max(0) + 1 + ks
    grid = x.new_zeros((1, Cin, *size))
    grid[0, :, gi[:, 0], gi[:, 1], gi[:, 2]] = x.t()
    # out_fine[c_f] = sum_k x[c_c] W_k with c_f = c_c + o_k * t_fine  (o_k = -1..1)
    # conv_transpose3d: out[m*s - p + j] += x[m] Wd[j]  =>  j = o + 1, p = 1, output_padding = 1
    Wd = Wk.permute(3, 4, 0, 1, 2).contiguous()
    y = F.conv_transpose3d(grid, Wd, stride=stride, padding=ks // 2, output_padding=stride - 1)
    go = (C_out[:, 1:] - lo) // t_fine
    return y[0, :, go[:, 0], go[:, 1], go[:, 2]].t()


# --------------------------------------------------------------------------------------------------
# parameters with the reference's names / shapes / init
# --------------------------------------------------------------------------------------------------

CHANNELS = [None, 32, 64, 128, 256]          # model/resunet.py:247
TR_CHANNELS = [None, 64, 64, 64, 128]        # model/resunet.py:248


def param_shapes(in_channels=1, out_channels=32, conv1_kernel_size=5, channels=CHANNELS, tr=TR_CHANNELS):
    """name -> shape for every conv kernel / bias and BN of ResUNetBN2C (model/resunet.py:38-171)."""
    sh = {}

    def conv(name, ci, co, ks):
        sh[name + ".kernel"] = (ks ** 3, ci, co) if ks > 1 else (ci, co)

    def bn(name, c):
        sh[name + ".bn.weight"] = (c,)
        sh[name + ".bn.bias"] = (c,)
        sh[name + ".bn.running_mean"] = (c,)
        sh[name + ".bn.running_var"] = (c,)

    def block(name, c):
        conv(name + ".conv1", c, c, 3); bn(name + ".norm1", c)
        conv(name + ".conv2", c, c, 3); bn(name + ".norm2", c)

    conv("conv1", in_channels, channels[1], conv1_kernel_size); bn("norm1", channels[1]); block("block1", channels[1])
    conv("conv2", channels[1], channels[2], 3); bn("norm2", channels[2]); block("block2", channels[2])
    conv("conv3", channels[2], channels[3], 3); bn("norm3", channels[3]); block("block3", channels[3])
    conv("conv4", channels[3], channels[4], 3); bn("norm4", channels[4]); block("block4", channels[4])
    conv("conv4_tr", channels[4], tr[4], 3); bn("norm4_tr", tr[4]); block("block4_tr", tr[4])
    conv("conv3_tr", channels[3] + tr[4], tr[3], 3); bn("norm3_tr", tr[3]); block("block3_tr", tr[3])
    conv("conv2_tr", channels[2] + tr[3], tr[2], 3); bn("norm2_tr", tr[2]); block("block2_tr", tr[2])
    conv("conv1_tr", channels[1] + tr[2], tr[1], 1)
    conv("final", tr[1], out_channels, 1)
    sh["final.bias"] = (1, out_channels)
    return sh


def random_state(seed=0, dtype=torch.float64, **kw):
    """Random parameters (uniform +-1/sqrt(fan), BN weight ~ U(0.5,1.5)) for parity tests."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for name, shape in param_shapes(**kw).items():
        if name.endswith("running_mean"):
            st[name] = torch.zeros(shape, dtype=dtype)
        elif name.endswith("running_var"):
            st[name] = torch.ones(shape, dtype=dtype)
        elif name.endswith("bn.weight"):
            st[name] = (0.5 + torch.rand(shape, generator=g, dtype=torch.float64)).to(dtype)
        elif name.endswith("bn.bias") or name.endswith("final.bias"):
            st[name] = (0.2 * torch.rand(shape, generator=g, dtype=torch.float64) - 0.1).to(dtype)
        else:
            fan = shape[-2] * (shape[0] if len(shape) == 3 else 1)
            st[name] = ((2 * torch.rand(shape, generator=g, dtype=torch.float64) - 1) / fan ** 0.5).to(dtype)
    return st
