"""Residual block of the sparse U-Net (interface of model/residual_block.py:9-77).

Same computation as BasicBlockBase.forward (:37-53) -- conv3^3, norm, relu, conv3^3, norm, += residual, relu --
but the two norm sites call the fused HIP kernels: norm+relu in one pass, norm+residual+relu in one pass.
"""
import torch.nn as nn

import gcl_amd.MinkowskiEngine as ME
from gcl_amd.model.common import get_norm


class BasicBlockBase(nn.Module):
    expansion = 1
    NORM_TYPE = "BN"

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
        super().__init__()
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dimension=D)
        self.norm1 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, bias=False,
                                             dimension=D)
        self.norm2 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.downsample = downsample

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = ME.conv_bn(self.conv1, self.norm1, x, relu=True)
        return ME.conv_bn(self.conv2, self.norm2, y, residual=shortcut, relu=True)


class BasicBlockBN(BasicBlockBase):
    NORM_TYPE = "BN"


class BasicBlockIN(BasicBlockBase):
    NORM_TYPE = "IN"


_BLOCKS = {"BN": BasicBlockBN, "IN": BasicBlockIN}


def get_block(norm_type, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
    if norm_type not in _BLOCKS:
        raise ValueError(f"Type {norm_type}, not defined")
    return _BLOCKS[norm_type](inplanes, planes, stride, dilation, downsample, bn_momentum, D)
