"""Building blocks of the 4-stage IMHN (Identity-Mapping Hourglass Network), inference-only.

Module/attribute names follow /root/reference/models/layers_transposed.py so that a checkpoint written by the
reference (`checkpoint['weights']`, evaluate.py:308-309) loads with strict=True: Residual (:12-48), Conv (:90-122),
DilatedConv (:125-157), Backbone (:160-196), Hourglass (:199-286), SELayer (:289-310).  The convolutions themselves
run on PyTorch-ROCm (MIOpen / hipBLASLt); see posepaf/fused_model.py for the BN-folded, channels-last fp16 form
that the benchmark runs."""
import torch
import torch.nn.functional as F
from torch import nn

LEAK = 0.01


def _act():
    return nn.LeakyReLU(negative_slope=LEAK, inplace=True)


class Residual(nn.Module):
    """1x1 -> 3x3 -> 1x1 bottleneck (mid = outs // 2) with BN after every conv, optional 1x1+BN skip."""

    def __init__(self, ins, outs, bn=True, relu=True):
        super().__init__()
        mid = outs // 2
        self.relu_flag = relu
        self.convBlock = nn.Sequential(
            nn.Conv2d(ins, mid, 1, bias=False), nn.BatchNorm2d(mid), _act(),
            nn.Conv2d(mid, mid, 3, 1, 1, bias=False), nn.BatchNorm2d(mid), _act(),
            nn.Conv2d(mid, outs, 1, bias=False), nn.BatchNorm2d(outs))
        if ins != outs:
            self.skipConv = nn.Sequential(nn.Conv2d(ins, outs, 1, bias=False), nn.BatchNorm2d(outs))
        self.relu = _act()
        self.ins, self.outs = ins, outs

    def forward(self, x):
        y = self.convBlock(x)
        y = y + (self.skipConv(x) if self.ins != self.outs else x)
        return self.relu(y) if self.relu_flag else y


class _ConvBnAct(nn.Module):
    def __init__(self, inp_dim, out_dim, kernel_size, stride, padding, dilation, bn, relu):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, kernel_size, stride, padding=padding, dilation=dilation, bias=not bn)
        self.bn = nn.BatchNorm2d(out_dim) if bn else None
        self.relu = _act() if relu else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return x


class Conv(_ConvBnAct):
    """conv(k, 'same' padding) [+ BN] [+ LeakyReLU]; bias only when there is no BN."""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=True, relu=True, dropout=False, dialated=1):
        super().__init__(inp_dim, out_dim, kernel_size, stride, (kernel_size - 1) // 2, 1, bn, relu)


class DilatedConv(_ConvBnAct):
    """3x3 dilated conv, stride 1, padding = dilation."""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=True, relu=True, dropout=False, dialation=3):
        super().__init__(inp_dim, out_dim, kernel_size, stride, dialation, dialation, bn, relu)


class Backbone(nn.Module):
    """7x7/2 stem -> Residual(64,128) -> maxpool -> Residual(128,128) -> six dilated convs; output is
    concat(trunk, dilated) = 256 channels at 1/4 resolution."""

    def __init__(self, inplanes=3):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = _act()
        self.res1 = Residual(64, 128)
        self.pool = nn.MaxPool2d(2, 2)
        self.res2 = Residual(128, 128)
        self.dilation = nn.Sequential(*[DilatedConv(128, 128, dialation=d) for d in (3, 3, 4, 4, 5, 5)])

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.res2(self.pool(self.res1(x)))
        return torch.cat([x, self.dilation(x)], dim=1)


class Hourglass(nn.Module):
    """Order-`depth` hourglass whose channel count grows by `increase` per level; returns the full-resolution
    output followed by the four coarser maps it passes through (5 scales)."""

    def __init__(self, depth, nFeat, increase=128, bn=False):
        super().__init__()
        self.depth = depth
        levels = []
        for i in range(depth):
            c, cn = nFeat + increase * i, nFeat + increase * (i + 1)
            mods = [Residual(c, c, bn=bn), Residual(c, cn, bn=bn), Residual(cn, c, bn=bn), Conv(c, c, bn=bn)]
            if i == depth - 1:
                mods.append(Residual(cn, cn, bn=bn))
            levels.append(nn.ModuleList(mods))
        self.hg = nn.ModuleList(levels)
        self.downsample = nn.MaxPool2d(2, 2)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")

    def _level(self, i, x, coarse):
        up1 = self.hg[i][0](x)
        low = self.hg[i][1](self.downsample(x))
        low = self.hg[i][4](low) if i == self.depth - 1 else self._level(i + 1, low, coarse)
        coarse.append(low)
        return up1 + self.hg[i][3](self.upsample(self.hg[i][2](low)))

    def forward(self, x):
        coarse = []
        top = self._level(0, x, coarse)
        return [top] + coarse[::-1]


class SELayer(nn.Module):
    """Squeeze-and-excitation: global mean -> Linear -> LeakyReLU -> Linear -> Sigmoid -> channel scale."""

    def __init__(self, inp_dim, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(inp_dim, inp_dim // reduction), nn.LeakyReLU(inplace=True),
                                nn.Linear(inp_dim // reduction, inp_dim), nn.Sigmoid())

    def forward(self, x):
        b, c = x.shape[:2]
        return x * self.fc(self.avg_pool(x).view(b, c)).view(b, c, 1, 1)
