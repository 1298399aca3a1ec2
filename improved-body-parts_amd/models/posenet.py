"""PoseNet / NetworkEval: the 4-stage IMHN the reference evaluates (models/posenet.py:51-122, :184-202).

Same constructor signature, same forward contract (NHWC float image batch in [0,1] -> list[stage][scale] of
(N, 50, H/2^(2+s), W/2^(2+s))) and the same state_dict keys (`posenet.pre.conv1.weight` ...
`posenet.merge_preds.2.4.conv.bn.num_batches_tracked`, 1848 entries).  Inference only."""
import torch
from torch import nn

from models.layers_transposed import Backbone, Conv, Hourglass, SELayer


class Merge(nn.Module):
    """1x1 conv (no activation) that changes the channel count."""

    def __init__(self, x_dim, y_dim, bn=False):
        super().__init__()
        self.conv = Conv(x_dim, y_dim, 1, relu=False, bn=bn)

    def forward(self, x):
        return self.conv(x)


class Features(nn.Module):
    """Per scale: 3x3 conv -> 3x3 conv -> SE, all to `inp_dim` channels."""

    def __init__(self, inp_dim, increase=128, bn=False):
        super().__init__()
        self.before_regress = nn.ModuleList([
            nn.Sequential(Conv(inp_dim + i * increase, inp_dim, 3, bn=bn), Conv(inp_dim, inp_dim, 3, bn=bn),
                          SELayer(inp_dim)) for i in range(5)])

    def forward(self, fms):
        assert len(fms) == 5
        return [blk(f) for blk, f in zip(self.before_regress, fms)]


class PoseNet(nn.Module):
    def __init__(self, num_stages, inp_dim, oup_dim, bn=False, increase=128, init_weights=True, **kwargs):
        super().__init__()
        self.pre = Backbone()
        self.hourglass = nn.ModuleList()
        self.features = nn.ModuleList()
        self.outs = nn.ModuleList()
        self.merge_features = nn.ModuleList()
        self.merge_preds = nn.ModuleList()
        for t in range(num_stages):
            self.hourglass.append(Hourglass(depth=4, nFeat=inp_dim, increase=increase, bn=bn))
            self.features.append(Features(inp_dim=inp_dim, increase=increase, bn=bn))
            self.outs.append(nn.ModuleList([Conv(inp_dim, oup_dim, 1, relu=False, bn=False) for _ in range(5)]))
            if t < num_stages - 1:
                self.merge_features.append(
                    nn.ModuleList([Merge(inp_dim, inp_dim + j * increase, bn=bn) for j in range(5)]))
                self.merge_preds.append(
                    nn.ModuleList([Merge(oup_dim, inp_dim + j * increase, bn=bn) for j in range(5)]))
        self.num_stages = num_stages
        self.num_scales = 5
        if init_weights:
            self._initialize_weights()

    def forward(self, imgs):
        x = self.pre(imgs.permute(0, 3, 1, 2))
        preds, caches = [], None
        for t in range(self.num_stages):
            hg = self.hourglass[t](x)
            if caches is not None:
                hg = [a + c for a, c in zip(hg, caches)]
            feats = self.features[t](hg)
            stage_preds = [head(f) for head, f in zip(self.outs[t], feats)]
            if t != self.num_stages - 1:
                caches = [self.merge_preds[t][s](stage_preds[s]) + self.merge_features[t][s](feats[s])
                          for s in range(self.num_scales)]
                x = x + caches[0]
            preds.append(stage_preds)
        return preds

    def _initialize_weights(self):  # models/posenet.py:124-144
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.001)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                torch.nn.init.normal_(m.weight.data, 0, 0.01)
                m.bias.data.zero_()


class NetworkEval(nn.Module):
    """Inference wrapper (models/posenet.py:184-202): `opt` supplies nstack / hourglass_inp_dim / increase,
    `config` supplies num_layers (50)."""

    def __init__(self, opt, config, bn=False):
        super().__init__()
        self.posenet = PoseNet(opt.nstack, opt.hourglass_inp_dim, config.num_layers, bn=bn, init_weights=False,
                               increase=opt.increase)

    def forward(self, inp_imgs):
        if self.training:
            raise ValueError("\nOnly eval mode is available!!")
        return self.posenet(inp_imgs)
