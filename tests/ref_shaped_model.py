"""A sparse residual U-Net written against the ME surface exactly the way the reference's model files use it
(test helper, NOT product code; written from SURVEY.md section 3.2 / 8b, no reference text).

The product model (gcl_amd/model/) calls the fused entry ``ME.conv_bn`` and the extra arguments of
``MinkowskiBatchNorm.forward``.  A user who points ``sys.modules['MinkowskiEngine']`` at gcl_amd's module and keeps
the reference's own model files goes through the UN-FUSED surface instead:

    out = self.conv(x); out = self.norm(out); out = MEF.relu(out)        (model/resunet.py:174-181)
    out += residual; out = MEF.relu(out)                                  (model/residual_block.py:50-51)
    out = ME.cat(a, b)                                                    (model/resunet.py:203,210,217)
    ME.SparseTensor(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True),
                    coordinate_map_key=out.coordinate_map_key,
                    coordinate_manager=out.coordinate_manager)            (model/resunet.py:226-230)

This file reproduces that call pattern (same parameter names as the reference, so state dicts interchange) and the
GPU tests compare it with the fused product model and with the fp64 oracle.
"""
import torch
import torch.nn as nn


def build(ME, MEF, in_channels=1, out_channels=32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5,
          channels=(None, 32, 64, 128, 256), tr_channels=(None, 64, 64, 64, 128), norm="BN", block_norm="BN"):
    """Returns an ``nn.Module`` built only from the 12 symbols of SURVEY.md 8b, taken from the module objects passed
    in (so the same code runs on any MinkowskiEngine-compatible module)."""

    def make_norm(kind, c):
        if kind == "BN":
            return ME.MinkowskiBatchNorm(c, momentum=bn_momentum)
        return ME.MinkowskiInstanceNorm(c, dimension=3)

    class Block(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.conv1 = ME.MinkowskiConvolution(c, c, kernel_size=3, stride=1, dimension=3)
            self.norm1 = make_norm(block_norm, c)
            self.conv2 = ME.MinkowskiConvolution(c, c, kernel_size=3, stride=1, dilation=1, bias=False, dimension=3)
            self.norm2 = make_norm(block_norm, c)

        def forward(self, x):
            residual = x
            out = self.conv1(x)
            out = self.norm1(out)
            out = MEF.relu(out)
            out = self.conv2(out)
            out = self.norm2(out)
            out += residual
            out = MEF.relu(out)
            return out

    class Net(ME.MinkowskiNetwork):
        def __init__(self):
            super().__init__(3)
            ch, tr = channels, tr_channels
            self.normalize_feature = normalize_feature
            conv = lambda ci, co, ks, s: ME.MinkowskiConvolution(in_channels=ci, out_channels=co, kernel_size=ks,
                                                                 stride=s, dilation=1, bias=False, dimension=3)
            up = lambda ci, co: ME.MinkowskiConvolutionTranspose(in_channels=ci, out_channels=co, kernel_size=3,
                                                                  stride=2, dilation=1, bias=False, dimension=3)
            # the reference's parameter names (state dicts interchange): conv{l} / norm{l} / block{l} going down,
            # conv{l}_tr / norm{l}_tr / block{l}_tr coming back up, then the two 1x1x1 heads
            down_in = {1: in_channels, 2: ch[1], 3: ch[2], 4: ch[3]}
            for lvl in (1, 2, 3, 4):
                first = conv(down_in[lvl], ch[lvl], conv1_kernel_size if lvl == 1 else 3, 1 if lvl == 1 else 2)
                self.add_module(f"conv{lvl}", first)
                self.add_module(f"norm{lvl}", make_norm(norm, ch[lvl]))
                self.add_module(f"block{lvl}", Block(ch[lvl]))
            up_in = {4: ch[4], 3: ch[3] + tr[4], 2: ch[2] + tr[3]}
            for lvl in (4, 3, 2):
                self.add_module(f"conv{lvl}_tr", up(up_in[lvl], tr[lvl]))
                self.add_module(f"norm{lvl}_tr", make_norm(norm, tr[lvl]))
                self.add_module(f"block{lvl}_tr", Block(tr[lvl]))
            self.conv1_tr = conv(ch[1] + tr[2], tr[1], 1, 1)
            self.final = ME.MinkowskiConvolution(in_channels=tr[1], out_channels=out_channels, kernel_size=1, stride=1,
                                                 dilation=1, bias=True, dimension=3)

        def _stage(self, suffix, x):
            """conv -> norm -> residual block as three separate module calls (the un-fused surface); the caller
            applies MEF.relu itself."""
            y = self._modules["conv" + suffix](x)
            y = self._modules["norm" + suffix](y)
            return self._modules["block" + suffix](y)

        def forward(self, x):
            kept, cur = {}, x
            for lvl in (1, 2, 3, 4):                      # encoder: keep the un-activated block output for the skip
                kept[lvl] = self._stage(str(lvl), cur)
                cur = MEF.relu(kept[lvl])
            for lvl in (4, 3, 2):                         # decoder: up-convolve, activate, concatenate with the skip
                cur = ME.cat(MEF.relu(self._stage(f"{lvl}_tr", cur)), kept[lvl - 1])
            cur = self.final(MEF.relu(self.conv1_tr(cur)))
            if not self.normalize_feature:
                return cur
            rows = cur.F
            return ME.SparseTensor(rows / torch.norm(rows, p=2, dim=1, keepdim=True),
                                   coordinate_map_key=cur.coordinate_map_key,
                                   coordinate_manager=cur.coordinate_manager)

    return Net()
