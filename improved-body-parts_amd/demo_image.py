#!/usr/bin/env python3
"""demo_image.py -- the reference's single-image demo (demo_image.py:80-321) on the MI355X-native path.

    python demo_image.py --image photo.npy|.png --output result.png --run_refactor --run_cpp [-p checkpoint.pth]
    python demo_image.py --synthetic 6 --output result.npy --run_refactor --run_cpp     # offline: injected pose scene

Same flags as the reference (--image, --output, --run_refactor, --run_cpp): with --run_refactor the refactored path
(predict_refactor + heatmap_nms, then pafprocess with --run_cpp or find_connections / find_humans without), otherwise
the original path (predict + find_peaks + find_connections + find_humans) -- through the reference-named functions of
utils.parse_skeletons, so every stage runs on the GPU.  Rendering is NumPy (utils/draw.py); OpenCV / matplotlib are not
needed.  Images: .npy (BGR uint8) always, other formats through PIL when it is importable.  Offline there is no
checkpoint, so --synthetic P adds a synthetic P-people scene to the network output (as bench.py does) to have people to draw."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def load_image(path):
    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path), np.uint8)
    from PIL import Image      # RGB -> the BGR order cv2.imread returns (demo_image.py:81)
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def save_image(path, canvas):
    if path.endswith(".npy"):
        np.save(path, canvas)
    else:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(canvas[:, :, ::-1])).save(path)


def humans_from_python_rules(person_to_joint_assoc, joint_candidates):
    """demo_image.py:150-172"""
    from utils.common import BodyPart, Human
    humans = []
    for person_id, person in enumerate(person_to_joint_assoc[..., 0]):
        human = Human([])
        for part_idx, peak_id in enumerate(person[:18]):
            if peak_id < 0:
                continue
            x, y, s = joint_candidates[int(peak_id), :3]
            human.body_parts[part_idx] = BodyPart("%d-%d" % (person_id, part_idx), part_idx, x, y, s)
        if human.body_parts:
            human.score = person[-2] / person[-1]
            humans.append(human)
    return humans


def process(ori_img, model, test_cfg, model_cfg, config, run_refactor, run_cpp, inject=None):
    """-> (canvas, humans): the reference's process() (demo_image.py:80-243) with the image already loaded"""
    from posepaf import skeleton as sk
    from utils import draw, parse_skeletons as ps
    from utils.common import BodyPart, Human
    image_h = ori_img.shape[0]
    pairs = np.array(sk.LIMB_PAIRS)
    canvas = ori_img.copy()
    if run_refactor:
        heatmaps, pafs = ps.predict_refactor(ori_img, model, test_cfg, model_cfg, "", flip_avg=True, config=config)
        if inject is not None:
            heatmaps = heatmaps * 1e-3 + inject[0]     # random weights: keep their O(1) ripple out of the demonstration
            pafs = pafs * 1e-3 + inject[1]
        all_peaks = ps.heatmap_nms(heatmaps, model_cfg["stride"])
        # demo_image.py:94-172: x4 limb upsample + pafprocess (--run_cpp) or find_connections / find_humans (without): both rule
        # sets run inside the native batched path, which evaluates the x4 bicubic on the fly; humans come back as records
        from posepaf.api import PosePostProcessor, record_humans
        import torch
        assert len(all_peaks) == 18
        h, w = heatmaps.shape[:2]
        net = torch.zeros((1, 1, 50, h, w), dtype=torch.float32, device="cuda")
        net[0, 0, :30] = torch.from_numpy(np.ascontiguousarray(pafs.transpose(2, 0, 1))).cuda()
        net[0, 0, 30:50] = torch.from_numpy(np.ascontiguousarray(heatmaps.transpose(2, 0, 1))).cuda()
        post = PosePostProcessor(max_batch=1, max_h=h, max_w=w, max_peaks_per_part=64)
        rec = (post.process(net, image_h, flip=False) if run_cpp else post.process_py(net, image_h, flip=False))[0]
        post.close()
        humans = []
        for hid, hm in enumerate(record_humans(rec)):
            human = Human([])
            for part in range(18):
                if hm["ids"][part] >= 0:
                    human.body_parts[part] = BodyPart("%d-%d" % (hid, part), part, int(hm["x"][part]), int(hm["y"][part]),
                                                      float(hm["part_score"][part]))
            if human.body_parts:
                human.score = hm["score"]
                humans.append(human)
        canvas = draw.draw_humans(canvas, humans)
        return canvas, humans
    heatmaps, pafs = ps.predict(ori_img, model, test_cfg, model_cfg, "", flip_avg=True, config=config)
    if inject is not None:
        heatmaps = heatmaps * 1e-3 + inject[0]
        pafs = pafs * 1e-3 + inject[1]
    all_peaks = ps.find_peaks(heatmaps, test_cfg)
    connected, special = ps.find_connections(all_peaks, pafs.astype(np.float32), image_h, test_cfg, pairs)
    persons, cand = ps.find_humans(connected, special, all_peaks, test_cfg, pairs)
    humans = humans_from_python_rules(persons, cand)
    canvas = draw.draw_limbs_original(canvas, persons, cand, sk.LIMB_PAIRS, sk.DRAW_LIST)
    return canvas, humans


def main():
    ap = argparse.ArgumentParser(description="PoseNet demo (MI355X-native path)")
    ap.add_argument("--image", type=str, default=None, help="input image (.npy BGR uint8, or any format PIL reads)")
    ap.add_argument("--output", type=str, default="result.npy")
    ap.add_argument("--run_refactor", action="store_true")
    ap.add_argument("--run_cpp", action="store_true")
    ap.add_argument("--checkpoint_path", "-p", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="no --image: a random 512x512 image with this many injected people")
    a = ap.parse_args()
    if a.run_cpp and not a.run_refactor:
        raise SystemExit("--run_cpp only exists on the refactored path (demo_image.py:118)")
    import torch
    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf import skeleton as sk, synth
    from posepaf.fused_model import FusedIMHN
    from posepaf.model_init import deterministic_init
    opt, config = TrainingOpt(), GetConfig(TrainingOpt.config_name)
    net = NetworkEval(opt, config, bn=True).eval()
    if a.checkpoint_path:
        net.load_state_dict(torch.load(a.checkpoint_path, map_location="cpu", weights_only=True)["weights"])   # demo_image.py:292-293
    else:
        deterministic_init(net, 7)
    model = FusedIMHN.from_network(net).eval().cuda().half().to(memory_format=torch.channels_last)
    inject = None
    if a.image:
        img = load_image(a.image)
    else:
        img = np.random.default_rng(0).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    if a.synthetic:
        hp, wp = -(-img.shape[0] // 64) * 64, -(-img.shape[1] // 64) * 64
        scene = synth.make_net_output(a.synthetic, 4242, h=hp // 4, w=wp // 4, noise=0.0, dtype=np.float32, flip=False)[0]
        paf, heat = scene[:30].transpose(1, 2, 0), scene[30:50].transpose(1, 2, 0)
        if a.run_refactor:
            inject = (heat, paf)
        else:      # the original path works at image resolution: a smooth x4 up-sample of the scene, cropped to the image
            import torch.nn.functional as F

            def up(m):
                t = torch.from_numpy(np.ascontiguousarray(m.transpose(2, 0, 1)))[None]
                t = F.interpolate(t, scale_factor=4, mode="bicubic", align_corners=False)[0]
                return t.permute(1, 2, 0).numpy()[:img.shape[0], :img.shape[1]].astype(np.float64)
            inject = (up(heat), up(paf))
    t0 = time.time()
    canvas, humans = process(img, model, sk.default_test_cfg(), sk.default_model_cfg(), config, a.run_refactor, a.run_cpp, inject)
    print("processing time is %.5f, %d people" % (time.time() - t0, len(humans)))
    save_image(a.output, canvas)


if __name__ == "__main__":
    main()
