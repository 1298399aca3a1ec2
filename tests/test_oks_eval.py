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
