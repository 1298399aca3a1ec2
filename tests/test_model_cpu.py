"""CPU: our IMHN definition against the reference's own module (golden vectors made by importing
/root/reference/models/posenet.py in the build container): identical state_dict keys/shapes (so the published
checkpoint loads with strict=True) and the same forward result under a shared deterministic init."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def model():
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf.model_init import deterministic_init
    m = NetworkEval(TrainingOpt(), GetConfig("Canonical"), bn=True).eval()
    deterministic_init(m, seed=7)
    return m


def test_state_dict_manifest(model):
    want = json.load(open(os.path.join(GOLDEN, "g5_state_dict_manifest.json")))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert len(got) == 1848
    assert got == want
    assert list(got)[0] == "posenet.pre.conv1.weight"
    assert sum(p.numel() for p in model.parameters()) == 128998760


def test_forward_matches_reference_module(model):
    g = np.load(os.path.join(GOLDEN, "g5_model_forward.npz"))
    assert int(g["n_params"]) == 128998760
    with torch.no_grad():
        out = model(torch.from_numpy(g["x"]))
    assert len(out) == 4 and len(out[0]) == 5
    for name, t in (("last_stage_scale0", out[-1][0]), ("last_stage_scale4", out[-1][4]), ("first_stage_scale0", out[0][0])):
        want = g[name]
        assert t.shape == want.shape
        # same ops in the same order on the same CPU kernels: tolerance covers only thread-count dependent
        # reduction order inside the convolutions
        assert np.allclose(t.numpy(), want, rtol=1e-4, atol=1e-5), name
    assert float(np.abs(g["last_stage_scale0"]).mean()) > 1e-3  # the fixture is not degenerate


def test_train_mode_is_refused(model):
    model.train()
    with pytest.raises(ValueError):
        model(torch.zeros(1, 64, 64, 3))
    model.eval()


def test_config_surface():
    from config.config import GetConfig, TrainingOpt
    c = json.load(open(os.path.join(GOLDEN, "constants.json")))
    cfg = GetConfig(TrainingOpt.config_name)
    assert cfg.limbs_conn.tolist() == c["limbs_conn"]
    assert cfg.flip_heat_ord.tolist() == c["flip_heat_ord"] and cfg.flip_paf_ord.tolist() == c["flip_paf_ord"]
    assert (cfg.paf_layers, cfg.heat_layers, cfg.num_layers, cfg.stride) == (c["paf_layers"], c["heat_layers"], c["num_layers"], c["stride"])
    assert cfg.parts == c["parts"]
    assert {str(k): v for k, v in cfg.dt_gt_mapping.items()} == c["dt_gt_mapping"]


# ---- the algebraic rewrites of posepaf/fused_model.py, checked as ALGEBRA in float64 on the CPU (the kernels that execute them are
# ---- compared with torch in tests/test_gpu_model.py)
def test_collapsed_upsample_weights_reproduce_the_convolution_of_the_upsampled_tensor():
    """conv3x3(upsample2(x)) (models/layers_transposed.py:270-275) == four 2x2 convolutions of x, one per output phase, with the
    tap sums of FConv._collapsed_weights (formed in fp32): identical up to that rounding, borders included."""
    import torch.nn.functional as F
    from posepaf import fused_model as fm
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(5, 7, 3, 1, 1, bias=True).double()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g, dtype=torch.float64))
        conv.bias.copy_(torch.randn(7, generator=g, dtype=torch.float64))
    f = fm.FConv(conv, None, False).double()
    x = torch.randn(2, 5, 6, 9, generator=g, dtype=torch.float64)
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), f.weight, f.bias, 1, 1)
    w4 = f._collapsed_weights()                                  # (4, K, 2, 2, C): phase (py, px), taps (a, b)
    assert w4.shape == (4, 7, 2, 2, 5) and w4.dtype == torch.float64
    got = torch.empty_like(want)
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        # phase (py, px) sees input rows y - 1 + py + a and columns x - 1 + px + b: pad one row / column on the side it reaches over
        xp = F.pad(x, (1 - px, px, 1 - py, py))
        got[:, :, py::2, px::2] = F.conv2d(xp, w4[ph].permute(0, 3, 1, 2).contiguous(), f.bias)
    assert (got - want).abs().max().item() <= 1e-6 * want.abs().max().item()      # (the tap sums are formed in fp32 by design)


def test_folded_prediction_merge_is_the_same_linear_map():
    """merge_features(f) + merge_preds(head(f)) (models/posenet.py:116-117; 1x1 convolutions, no activation between head and
    merge) == one 1x1 convolution with W' = Wf + Wp Wh, b' = bf + bp + Wp bh -- the fold FusedIMHN applies at load time."""
    g = torch.Generator().manual_seed(4)
    c, k, p = 12, 10, 5
    wf, bf = torch.randn(k, c, generator=g, dtype=torch.float64), torch.randn(k, generator=g, dtype=torch.float64)
    wp, bp = torch.randn(k, p, generator=g, dtype=torch.float64), torch.randn(k, generator=g, dtype=torch.float64)
    wh, bh = torch.randn(p, c, generator=g, dtype=torch.float64), torch.randn(p, generator=g, dtype=torch.float64)
    f = torch.randn(3, c, 4, 6, generator=g, dtype=torch.float64)
    conv = lambda t, w, b: torch.einsum("nchw,kc->nkhw", t, w) + b[None, :, None, None]
    want = conv(f, wf, bf) + conv(conv(f, wh, bh), wp, bp)
    got = conv(f, wf + wp @ wh, bf + bp + wp @ bh)
    assert (got - want).abs().max().item() <= 1e-12 * want.abs().max().item()


def test_skip_convolution_as_extra_input_channels():
    """conv3(t) + skip(x) (models/layers_transposed.py:12-48) == one 1x1 convolution of [t ; x] with the weights side by side
    (pp_pw_cat_f16's view of a residual block's tail)."""
    g = torch.Generator().manual_seed(5)
    t, x = torch.randn(2, 6, 3, 4, generator=g, dtype=torch.float64), torch.randn(2, 9, 3, 4, generator=g, dtype=torch.float64)
    w3, ws = torch.randn(8, 6, generator=g, dtype=torch.float64), torch.randn(8, 9, generator=g, dtype=torch.float64)
    b = torch.randn(8, generator=g, dtype=torch.float64)
    want = torch.einsum("nchw,kc->nkhw", t, w3) + torch.einsum("nchw,kc->nkhw", x, ws) + b[None, :, None, None]
    got = torch.einsum("nchw,kc->nkhw", torch.cat([t, x], 1), torch.cat([w3, ws], 1)) + b[None, :, None, None]
    assert (got - want).abs().max().item() <= 1e-12 * want.abs().max().item()
