"""CPU: the in-repo COCO keypoint AP (pycocotools is absent: parity unpinned; these are hand-computed cases)."""
import numpy as np


def _gt(x0=100.0, y0=100.0):
    kp = np.zeros((17, 3))
    kp[:, 0] = x0 + np.arange(17) * 5
    kp[:, 1] = y0 + np.arange(17) * 3
    kp[:, 2] = 2
    return {"keypoints": kp.reshape(-1).tolist(), "area": 80.0 * 48.0, "bbox": [x0, y0, 80.0, 48.0], "num_keypoints": 17,
            "iscrowd": 0}


def test_oks_values():
    from posepaf import oks_eval as oe
    g = np.asarray(_gt()["keypoints"]).reshape(17, 3)
    assert oe.compute_oks(g, g, 3840.0) == 1.0
    d = g.copy()
    d[:, 0] += 10.0
    want = np.mean(np.exp(-(100.0) / ((oe.KPT_SIGMAS * 2) ** 2) / 3840.0 / 2))
    assert abs(oe.compute_oks(d, g, 3840.0) - want) < 1e-12
    g2 = g.copy()
    g2[5:, 2] = 0       # unlabelled keypoints do not count
    assert abs(oe.compute_oks(d, g2, 3840.0) - np.mean(np.exp(-(100.0) / ((oe.KPT_SIGMAS[:5] * 2) ** 2) / 3840.0 / 2))) < 1e-12


def test_ap_perfect_missing_and_false_positive():
    from posepaf import oks_eval as oe
    gts = {1: [_gt(), _gt(300, 200)], 2: [_gt(50, 50)]}
    perfect = {k: [{"keypoints": g["keypoints"], "score": 0.9 - 0.1 * i} for i, g in enumerate(v)] for k, v in gts.items()}
    r = oe.evaluate_keypoints(gts, perfect)
    assert abs(r["AP"] - 1.0) < 1e-9 and abs(r["AR"] - 1.0) < 1e-9 and r["n_gt"] == 3
    # one GT missed: recall 2/3 -> precision 1 up to recall 2/3, 0 beyond: 67 of 101 recall points
    miss = {1: perfect[1], 2: []}
    r = oe.evaluate_keypoints(gts, miss)
    assert abs(r["AP"] - 67 / 101) < 1e-9 and abs(r["AR"] - 2 / 3) < 1e-9
    # a high-scoring false positive first: precision at every recall level is capped at 3/4 after monotone smoothing
    fp = {1: perfect[1] + [{"keypoints": _gt(400, 400)["keypoints"], "score": 0.99}], 2: perfect[2]}
    r = oe.evaluate_keypoints(gts, fp)
    assert abs(r["AP50"] - 0.75) < 1e-9
    # empty detections
    r = oe.evaluate_keypoints(gts, {})
    assert r["AP"] == 0.0 and r["n_dt"] == 0


def test_gt_from_synth_joints():
    from posepaf import oks_eval as oe, synth
    _, joints = synth.make_scene(3, 5)
    gt = oe.gt_from_synth_joints(joints)
    assert 1 <= len(gt) <= 3
    for g in gt:
        kp = np.asarray(g["keypoints"]).reshape(17, 3)
        assert g["num_keypoints"] == int((kp[:, 2] > 0).sum()) and g["area"] > 0


def test_crowd_and_unlabelled_ground_truth_are_ignored_not_counted():
    """COCOeval: `ignore` = iscrowd or num_keypoints == 0 (cocoeval.py _prepare); a detection matched to an ignored GT is neither
    TP nor FP, a crowd GT may absorb several detections, ignored GT do not count towards recall."""
    from posepaf import oks_eval as oe
    real, crowd = _gt(), dict(_gt(300, 300), iscrowd=1)
    gts = {1: [crowd, real]}
    dts = {1: [{"keypoints": real["keypoints"], "score": 0.9},
               {"keypoints": crowd["keypoints"], "score": 0.8}, {"keypoints": crowd["keypoints"], "score": 0.7}]}
    r = oe.evaluate_keypoints(gts, dts)
    assert r["n_gt"] == 1 and abs(r["AP"] - 1.0) < 1e-9          # both crowd hits ignored: no false positive
    # the same two extra detections far from everything ARE false positives, but rank below the true positive: AP stays 1
    far = _gt(600, 600)["keypoints"]
    r = oe.evaluate_keypoints({1: [real]}, {1: [dts[1][0], {"keypoints": far, "score": 0.8}]})
    assert abs(r["AP"] - 1.0) < 1e-9
    # ... and cap precision when they outrank it
    r = oe.evaluate_keypoints({1: [real]}, {1: [{"keypoints": real["keypoints"], "score": 0.5}, {"keypoints": far, "score": 0.8}]})
    assert abs(r["AP"] - 0.5) < 1e-9
    # num_keypoints == 0: ignored; its OKS uses the distance to the doubled bounding box (0 inside)
    blank = dict(_gt(300, 300), num_keypoints=0)
    kp = np.asarray(blank["keypoints"]).reshape(17, 3).copy()
    kp[:, 2] = 0
    blank["keypoints"] = kp.reshape(-1).tolist()
    d_in = np.asarray(_gt(300, 300)["keypoints"]).reshape(17, 3)
    assert oe.compute_oks(d_in, kp, blank["area"], blank["bbox"]) == 1.0           # every joint inside the doubled box
    d_out = d_in.copy()
    d_out[:, 0] += 1000.0
    assert oe.compute_oks(d_out, kp, blank["area"], blank["bbox"]) < 1e-6
    r = oe.evaluate_keypoints({1: [real, blank]}, {1: [dts[1][0], {"keypoints": d_in.reshape(-1).tolist(), "score": 0.99}]})
    assert r["n_gt"] == 1 and abs(r["AP"] - 1.0) < 1e-9


def test_only_the_twenty_best_detections_per_image_count_and_area_comes_from_the_annotation():
    from posepaf import oks_eval as oe
    real = _gt()
    far = _gt(600, 600)["keypoints"]
    dts = {1: [{"keypoints": far, "score": 0.9 - 0.01 * i} for i in range(20)] + [{"keypoints": real["keypoints"], "score": 0.1}]}
    r = oe.evaluate_keypoints({1: [real]}, dts)
    assert r["n_dt"] == 20 and r["AP"] == 0.0                     # the true positive is the 21st: cut by maxDets = 20
    # OKS scales with the ANNOTATION's area (segment area in COCO), not with the box
    g = np.asarray(real["keypoints"]).reshape(17, 3)
    d = g.copy()
    d[:, 0] += 10
    small, big = oe.compute_oks(d, g, 1000.0), oe.compute_oks(d, g, 100000.0)
    assert small < big < 1.0
    lo = oe.evaluate_keypoints({1: [dict(real, area=200.0)]}, {1: [{"keypoints": d.reshape(-1).tolist(), "score": 1.0}]})
    hi = oe.evaluate_keypoints({1: [dict(real, area=1e6)]}, {1: [{"keypoints": d.reshape(-1).tolist(), "score": 1.0}]})
    assert lo["AP"] < hi["AP"] and abs(hi["AP"] - 1.0) < 1e-9
