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
            self.conv1 = conv(in_channels, ch[1], conv1_kernel_size, 1)
            self.norm1 = make_norm(norm, ch[1])
            self.block1 = Block(ch[1])
            self.conv2 = conv(ch[1], ch[2], 3, 2)
            self.norm2 = make_norm(norm, ch[2])
            self.block2 = Block(ch[2])
            self.conv3 = conv(ch[2], ch[3], 3, 2)
            self.norm3 = make_norm(norm, ch[3])
            self.block3 = Block(ch[3])
            self.conv4 = conv(ch[3], ch[4], 3, 2)
            self.norm4 = make_norm(norm, ch[4])
            self.block4 = Block(ch[4])
            self.conv4_tr = up(ch[4], tr[4])
            self.norm4_tr = make_norm(norm, tr[4])
            self.block4_tr = Block(tr[4])
            self.conv3_tr = up(ch[3] + tr[4], tr[3])
            self.norm3_tr = make_norm(norm, tr[3])
            self.block3_tr = Block(tr[3])
            self.conv2_tr = up(ch[2] + tr[3], tr[2])
            self.norm2_tr = make_norm(norm, tr[2])
            self.block2_tr = Block(tr[2])
            self.conv1_tr = conv(ch[1] + tr[2], tr[1], 1, 1)
            self.final = ME.MinkowskiConvolution(in_channels=tr[1], out_channels=out_channels, kernel_size=1, stride=1,
                                                 dilation=1, bias=True, dimension=3)

        def forward(self, x):
            out_s1 = self.conv1(x)
            out_s1 = self.norm1(out_s1)
            out_s1 = self.block1(out_s1)
            out = MEF.relu(out_s1)

            out_s2 = self.conv2(out)
            out_s2 = self.norm2(out_s2)
            out_s2 = self.block2(out_s2)
            out = MEF.relu(out_s2)

            out_s4 = self.conv3(out)
            out_s4 = self.norm3(out_s4)
            out_s4 = self.block3(out_s4)
            out = MEF.relu(out_s4)

            out_s8 = self.conv4(out)
            out_s8 = self.norm4(out_s8)
            out_s8 = self.block4(out_s8)
            out = MEF.relu(out_s8)

            out = self.conv4_tr(out)
            out = self.norm4_tr(out)
            out = self.block4_tr(out)
            out_s4_tr = MEF.relu(out)
            out = ME.cat(out_s4_tr, out_s4)

            out = self.conv3_tr(out)
            out = self.norm3_tr(out)
            out = self.block3_tr(out)
            out_s2_tr = MEF.relu(out)
            out = ME.cat(out_s2_tr, out_s2)

            out = self.conv2_tr(out)
            out = self.norm2_tr(out)
            out = self.block2_tr(out)
            out_s1_tr = MEF.relu(out)
            out = ME.cat(out_s1_tr, out_s1)

            out = self.conv1_tr(out)
            out = MEF.relu(out)
            out = self.final(out)
            if self.normalize_feature:
                return ME.SparseTensor(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True),
                                       coordinate_map_key=out.coordinate_map_key,
                                       coordinate_manager=out.coordinate_manager)
            return out

    return Net()
