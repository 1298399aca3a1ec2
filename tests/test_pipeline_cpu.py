"""CPU: pre-processing formula (A0) against the reference's padRightDownCorner output (golden) and the
numpy expression of utils/parse_skeletons.py:54-72."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def test_u8_over_255_float32_equals_float64_division_rounded():
    x = np.arange(256, dtype=np.uint8)
    want = np.float32(x / 255)                      # what the reference computes (:60)
    got = (torch.from_numpy(x).to(torch.float32) / 255.0).numpy()
    assert np.array_equal(got, want)


def test_preprocess_matches_reference_padding_and_flip():
    from posepaf.pipeline import preprocess_batch
    g = np.load(os.path.join(GOLDEN, "g4_util.npz"))
    img, padded, pad = g["img"], g["padded"], g["pad"]          # util.padRightDownCorner(img, 64, 128) from the reference
    assert padded.shape == (64, 128, 3) and list(pad) == [0, 0, 27, 58]
    want0 = np.float32(padded / 255)                             # :60
    want1 = want0[:, ::-1, :]                                    # :69
    out = preprocess_batch(torch.from_numpy(img[None]), flip=True, dtype=torch.float32).numpy()
    assert out.shape == (2, 64, 128, 3)
    assert np.array_equal(out[0], want0) and np.array_equal(out[1], want1)
    single = preprocess_batch(torch.from_numpy(img[None]), flip=False).numpy()
    assert np.array_equal(single[0], want0)
    # already aligned sizes are left alone
    a = np.random.default_rng(0).integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    o = preprocess_batch(torch.from_numpy(a), flip=True).numpy()
    assert o.shape == (4, 64, 64, 3) and np.array_equal(o[2], np.float32(a[1] / 255)) and np.array_equal(o[3], o[2][:, ::-1])


def test_util_module_matches_reference_golden():
    """utils.util (package) against the reference's outputs stored in g4_util.npz."""
    from utils import util
    g = np.load(os.path.join(GOLDEN, "g4_util.npz"))
    padded, pad = util.padRightDownCorner(g["img"], 64, 128)
    assert np.array_equal(padded, g["padded"]) and list(pad) == list(g["pad"])
    kept = util.keypoint_heatmap_nms(torch.from_numpy(g["hm"]), kernel=3, thre=0.1).numpy()
    assert np.array_equal(kept, g["kept"])
    for (x, y), want in zip(g["anchors"], g["refined"]):
        got = util.refine_centroid(g["big"], (int(x), int(y)), 2)
        assert np.allclose(np.array(got, np.float64), want, rtol=1e-6, atol=1e-7)


def test_evaluate_buckets_images_by_padded_shape():
    """evaluate.py batches heterogeneous image sizes per padded shape (utils/util.py:44-65 pads to /64): same padded shape ->
    same bucket, order inside a bucket = image order, every image in exactly one bucket."""
    import importlib.util
    import os
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("pp_evaluate", os.path.join(PKG, "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    assert ev.padded_shape(512, 512) == (512, 512) and ev.padded_shape(427, 640) == (448, 640) and ev.padded_shape(1, 65) == (64, 128)
    shapes = [(480, 640), (427, 640), (512, 512), (440, 600), (640, 480), (448, 640), (500, 500)]
    g = ev.buckets_by_padded_shape(shapes)
    assert g == {(512, 640): [0], (448, 640): [1, 3, 5], (512, 512): [2, 6], (640, 512): [4]}
    assert sorted(k for v in g.values() for k in v) == list(range(len(shapes)))
