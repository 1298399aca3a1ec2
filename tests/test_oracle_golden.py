"""CPU: the oracle (plain-C restatement) against the golden vectors captured from the reference.

Golden vectors come from tests/golden/make_golden.py: the reference's compiled C++ process_paf, and the
reference's Python (find_peaks_refactor, heatmap_nms without refinement, util.*) imported in the build
container.  Integer results must be identical; float results identical too (same arithmetic), except
refine_centroid (float32 numpy reductions restated in double; tolerance stated there).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_scene, scene_keys


def test_tables_match_reference(oracle):
    c = json.load(open(os.path.join(GOLDEN, "constants.json")))
    lp, fh, fp = oracle.tables()
    assert lp.tolist() == c["limbs_conn"]
    assert fh.tolist() == c["flip_heat_ord"]
    assert fp.tolist() == c["flip_paf_ord"]
    assert (c["paf_layers"], c["heat_layers"], c["num_layers"], c["stride"]) == (30, 18, 50, 4)


def test_find_peaks_refactor_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "g2_find_peaks.npz"))
    n = len([k for k in g.files if k.startswith("map")])
    assert n >= 8
    for i in range(n):
        got = oracle.find_peaks(g[f"map{i}"], 0.1, "plus")
        assert np.array_equal(got, g[f"peaks{i}"]), f"case {i}"


@pytest.mark.parametrize("key", scene_keys())
def test_scene_against_reference_outputs(oracle, key):
    net, g = load_scene(key)
    heat, paf = oracle.flip_average(net)
    # NMS + ordering + ids, pinned by the reference's heatmap_nms(bool_refine_center=False)
    jl0, _ = oracle.heatmap_nms(heat, 4, refine=False)
    assert np.array_equal(jl0, g["joint_list_norefine"])
    # refined peaks: same oracle bicubic as when the fixture was made (regression guard; step itself unpinned)
    jl, _ = oracle.heatmap_nms(heat, 4, refine=True)
    assert np.array_equal(jl, g["joint_list"])
    assert np.array_equal(jl[:, 3:], jl0[:, 3:])
    # process_paf against the reference's compiled C++: bit-exact ids, scores and peak table
    up = oracle.upsample4_hwc(paf)
    res = oracle.process_paf(jl[None], up, 512)
    assert not res["sort_oob"]
    assert np.array_equal(res["ids"], g["cpp_ids"])
    assert np.array_equal(res["scores"], g["cpp_scores"])
    assert np.array_equal(res["peaks"], g["cpp_peaks"])
    # whole-path entry point agrees with the staged calls
    full = oracle.pipeline(net, 512)
    assert np.array_equal(full["ids"], g["cpp_ids"]) and np.array_equal(full["scores"], g["cpp_scores"])


def test_flip_average_matches_numpy_semantics(oracle):
    """utils/parse_skeletons.py:82-103 written with numpy on the same arrays (float16 and float32)."""
    from posepaf import skeleton as sk, synth
    for dt in (np.float16, np.float32):
        net = synth.make_net_output(3, 5, h=16, w=24, dtype=dt)
        o0 = net[0].transpose(1, 2, 0)
        o1 = net[1].transpose(1, 2, 0)
        paf = (o0[:, :, :30] + o1[:, :, :30][:, ::-1, :][:, :, sk.FLIP_PAF_ORD]) / 2
        heat = (o0[:, :, 30:50] + o1[:, :, 30:50][:, ::-1, :][:, :, sk.FLIP_HEAT_ORD]) / 2
        assert paf.dtype == dt
        h_, p_ = oracle.flip_average(net)
        assert np.array_equal(h_.transpose(1, 2, 0), heat.astype(np.float32))
        assert np.array_equal(p_.transpose(1, 2, 0), paf.astype(np.float32))
    # binary16 conversion helpers agree with numpy on every finite half and on a float sweep
    halves = np.arange(65536, dtype=np.uint16)
    f = halves.view(np.float16).astype(np.float32)
    ok = np.isfinite(f)
    got = np.array([oracle.L.orc_f16_to_f32(int(h)) for h in halves[ok][::97]], np.float32)
    assert np.array_equal(got, f[ok][::97])
    xs = np.random.default_rng(0).normal(0, 1, 20000).astype(np.float32) * np.float32(10.0) ** np.random.default_rng(1).integers(-9, 5, 20000)
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.L.orc_f32_to_f16(float(x)) for x in xs], np.uint16)
    assert np.array_equal(got, want)


def test_bicubic_restatement_properties(oracle):
    """OpenCV is absent (parity unpinned): check the restatement's defining properties instead.
    Coefficients for the four x4 phases are exact dyadic rationals summing to 1; a constant image stays
    constant; upsampling commutes with transposition; phase table matches the closed form."""
    for x in (0.625, 0.875, 0.125, 0.375):
        import ctypes
        c = (ctypes.c_float * 4)()
        oracle.L.orc_cubic_coeffs.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
        oracle.L.orc_cubic_coeffs(x, c)
        c = np.array(list(c), np.float64)
        A = -0.75
        want = [((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A, ((A + 2) * x - (A + 3)) * x * x + 1,
                ((A + 2) * (1 - x) - (A + 3)) * (1 - x) ** 2 + 1]
        want.append(1 - sum(want))
        assert np.array_equal(c, np.array(want))          # exact: all dyadic
        assert c.sum() == 1.0
    m = np.full((7, 9), 0.37, np.float32)
    assert np.allclose(oracle.resize_cubic(m, 4, 4), 0.37, atol=1e-6)
    r = np.random.default_rng(3).random((11, 13), dtype=np.float32)
    # horizontal-then-vertical is not bitwise symmetric under transposition, only to rounding
    assert np.allclose(oracle.resize_cubic(r, 4, 4).T, oracle.resize_cubic(np.ascontiguousarray(r.T), 4, 4), atol=1e-6)
    up = oracle.upsample4_hwc(r[None])
    assert np.array_equal(up[:, :, 0], oracle.resize_cubic(r, 4, 4))


def test_util_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "g4_util.npz"))
    # keypoint_heatmap_nms (3x3, >= thre): kept pixels == oracle's mode-B peaks
    hm, kept = g["hm"], g["kept"]
    for ch in range(hm.shape[1]):
        pk = oracle.find_peaks(hm[0, ch], 0.1, "3x3")
        mask = np.zeros(hm.shape[2:], bool)
        mask[pk[:, 1], pk[:, 0]] = True
        assert np.array_equal(mask, kept[0, ch] != 0), ch
        assert np.array_equal(kept[0, ch][mask], hm[0, ch][mask])
    # refine_centroid: reference mixes float32 reductions and float64 products; 1e-5 relative
    for (x, y), want in zip(g["anchors"], g["refined"]):
        got = oracle.refine_centroid(g["big"], int(x), int(y), 2)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("key", scene_keys())
def test_python_twins_against_reference_outputs(oracle, key):
    """A8: find_connections + find_humans restated (oracle) vs the reference's own Python run on the same inputs
    (golden G3: py_persons / py_n_connections, made by importing /root/reference/utils/parse_skeletons.py)."""
    net, g = load_scene(key)
    _, paf = oracle.flip_average(net)
    up = oracle.upsample4_hwc(paf)
    persons, ncn = oracle.py_find_humans(g["joint_list"], up, 512)
    assert np.array_equal(ncn, g["py_n_connections"])
    want = g["py_persons"]
    assert persons.shape == want.shape
    assert np.array_equal(persons[:, :, 0], want[:, :, 0])                       # ids, counts, totals' slot layout
    assert np.allclose(persons[:, :, 1], want[:, :, 1], rtol=0, atol=1e-9)       # limb scores / lengths
    assert np.allclose(persons[:, 18, 0], want[:, 18, 0], rtol=0, atol=1e-9)
