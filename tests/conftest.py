import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "improved-body-parts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference_cpp():
    """The reference's own C++ pafprocess, compiled into oracle/_ref (prebuilt file on the GPU box)."""
    from oracle.oracle import REF_SO, Reference
    if not os.path.exists(REF_SO) and not os.path.isdir("/root/reference"):
        pytest.skip("oracle/_ref not built and no reference tree")
    return Reference()


def load_scene(key):
    """Regenerate a golden scene's input and verify it against the stored checksum."""
    import hashlib
    import json

    import numpy as np
    from posepaf import synth
    meta = json.load(open(os.path.join(GOLDEN, "scenes.json")))[key]
    g = np.load(os.path.join(GOLDEN, f"g1_scene_{key}.npz"))
    net = synth.make_net_output(meta["P"], meta["seed"], dtype=np.float16 if meta["dtype"] == "f16" else np.float32)
    sha = np.frombuffer(hashlib.sha256(net.tobytes()).digest(), np.uint8)
    assert (sha == g["net_sha256"]).all(), \
        f"synthetic scene {key} no longer reproduces the input the golden vector was made from; " \
        "re-run tests/golden/make_golden.py in the build container"
    return net, g


def scene_keys():
    import json
    return sorted(json.load(open(os.path.join(GOLDEN, "scenes.json"))).keys())
