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
