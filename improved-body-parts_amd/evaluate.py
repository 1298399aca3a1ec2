#!/usr/bin/env python3
"""evaluate.py -- the reference's evaluation loop (evaluate.py:235-330) on the MI355X-native path.

    python evaluate.py --run_refactor --run_cpp --synthetic 64 [--batch 16] [--dump_name results.json]
    python evaluate.py --gpus 8 --run_refactor --run_cpp --synthetic 5000          # launches its own 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --synthetic 5000 ...

What is kept from the reference: the flags --run_refactor / --run_cpp (:53-54; --run_refactor is required, --run_cpp
selects the C++ pafprocess rules -- the README's 65.8-AP / 7.3-fps row -- and its absence the pure-Python rules), the per-image result format (append_result :182-209) and the
COCO-style JSON dump (:269-270).  What changes: images are processed in batches, sharded i mod W over the ranks, with one
RCCL all-gather of the fixed-size records; nothing is copied to the host before the final records.

Offline there are no COCO images, annotations or pretrained weights, so the data source is synthetic: random uint8
images through the (randomly initialised or checkpoint-loaded) network, with ground-truth-style pose scenes injected
into the network output; the injected scenes' joints serve as ground truth for the in-repo OKS evaluation
(posepaf/oks_eval.py; pycocotools is absent).  With --checkpoint_path and --images DIR (npy files of BGR uint8 arrays)
the same loop runs on real inputs.

Image sizes may differ (COCO val2017 does): the reference pads every image to a multiple of 64 (utils/parse_skeletons.py:54,
utils/util.py:44-65) and runs it alone; here a rank's images are BUCKETED by padded shape and each bucket runs in batches --
the pad bytes (128) are written on the host, the per-image `img_h` of process_paf (evaluate.py:110) travels as a device
array.  --sizes HxW,HxW,... makes the synthetic set heterogeneous.
"""
import argparse
import glob
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from posepaf import coco, dist as pdist, oks_eval, synth  # noqa: E402
from posepaf._lib import RECORD_BYTES  # noqa: E402


def parse():
    ap = argparse.ArgumentParser(description="PoseNet evaluation (MI355X-native path)")
    ap.add_argument("--run_refactor", action="store_true")
    ap.add_argument("--run_cpp", action="store_true")
    ap.add_argument("--checkpoint_path", "-p", default=None, help="reference checkpoint (.pth with a 'weights' entry)")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic images")
    ap.add_argument("--sizes", default="512x512", help="synthetic image sizes HxW[,HxW...], cycled over the image index")
    ap.add_argument("--images", default=None, help="directory of .npy BGR uint8 images (any sizes)")
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without WORLD_SIZE: launch that many ranks (one per GPU)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dump_name", default="results.json")
    ap.add_argument("--people", type=int, nargs="*", default=[1, 2, 3, 4, 6, 8, 10, 5])
    ap.add_argument("--scales", type=float, nargs="*", default=None,
                    help="without --run_refactor: predict's `multiplier` list (the reference hard-codes [1.0], "
                         "utils/parse_skeletons.py:188); e.g. 0.5 1.0 1.5 for BASELINE config 5")
    return ap.parse_args()


def padded_shape(h, w, mult=64):
    """utils/util.py:44-65 padRightDownCorner: bottom / right up to the next multiple of `mult`"""
    return -(-h // mult) * mult, -(-w // mult) * mult


def buckets_by_padded_shape(shapes):
    """indices grouped by padded (Hp, Wp), each group in ascending index order; groups ordered by first appearance"""
    out = {}
    for k, (h, w) in enumerate(shapes):
        out.setdefault(padded_shape(h, w), []).append(k)
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:   # before anything touches the GPU; children are started, never exec'd
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                          f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                          os.path.abspath(__file__)] + sys.argv[1:], env=env))
    original = not a.run_refactor   # evaluate.py:81-84: predict + find_peaks + find_connections + find_humans
    if original and a.run_cpp:
        raise SystemExit("--run_cpp only exists on the refactored path (evaluate.py:97-129)")
    scales = a.scales or [1.0]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "evaluate.py needs MI355X GPUs"
    backend = os.environ.get("POSEPAF_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 control flow on fewer GPUs
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    torch.backends.cudnn.benchmark = True

    from config.config import GetConfig, TrainingOpt
    from models.posenet import NetworkEval
    from posepaf.api import PosePostProcessor
    from posepaf.fused_model import FusedIMHN
    from posepaf.model_init import deterministic_init
    from posepaf.pipeline import PosePipeline

    opt, config = TrainingOpt(), GetConfig(TrainingOpt.config_name)
    net = NetworkEval(opt, config, bn=True).eval()
    if a.checkpoint_path:
        ckpt = torch.load(a.checkpoint_path, map_location="cpu", weights_only=True)
        net.load_state_dict(ckpt["weights"])       # evaluate.py:308-309 (strict)
    else:
        deterministic_init(net, 7)
    model = FusedIMHN.from_network(net).eval().to(dev).half().to(memory_format=torch.channels_last)

    # ---- data
    if a.images:
        files = sorted(glob.glob(os.path.join(a.images, "*.npy")))
        n_images = len(files)
        load = lambda i: np.load(files[i])  # noqa: E731
        shape_of = lambda i: tuple(np.load(files[i], mmap_mode="r").shape[:2])  # noqa: E731
        image_ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
        gts, inject_for = {}, None
    else:
        n_images = a.synthetic
        sizes = [tuple(int(v) for v in t.lower().split("x")) for t in a.sizes.split(",")]
        image_ids = list(range(n_images))
        shape_of = lambda i: sizes[i % len(sizes)]  # noqa: E731
        load = lambda i: np.random.default_rng(10_000 + i).integers(0, 256, shape_of(i) + (3,), dtype=np.uint8)  # noqa: E731
        gts = {}
        scene_cache = {}

        def inject_for(i):
            """(network-output-shaped scene, ground-truth joints) for image i, rendered at ITS padded feature-map size"""
            p = a.people[i % len(a.people)]
            hp_, wp_ = padded_shape(*shape_of(i))
            key = (p, i % 64, hp_, wp_)
            if key not in scene_cache:
                scene_cache[key] = synth.make_scene(p, 20_000 + key[1], h=hp_ // 4, w=wp_ // 4, dtype=np.float16)
            return scene_cache[key]
    if n_images == 0:
        raise SystemExit("nothing to evaluate: pass --synthetic N or --images DIR")

    B = a.batch
    mine = pdist.shard_indices(n_images, rank, world)
    S = pdist.padded_shard_size(n_images, world)
    shapes = [shape_of(int(i)) for i in mine]                 # this rank's images, local order
    if original:
        groups = {}
        for k, hw in enumerate(shapes):                      # accumulators live at IMAGE resolution: bucket by exact size
            groups.setdefault(hw, []).append(k)
    else:
        groups = buckets_by_padded_shape(shapes)             # refactored path: bucket by padded shape
    all_shapes = [shape_of(i) for i in range(n_images)] if n_images <= 100_000 else shapes
    hp = max(padded_shape(h, w)[0] for h, w in all_shapes) // 4
    wp = max(padded_shape(h, w)[1] for h, w in all_shapes) // 4
    # feature maps up to hp x wp (area bound; pp_process_batch refuses a map that does not fit LDS, loudly)
    post = PosePostProcessor(max_batch=B, max_h=hp if not original else int(hp * max(scales) + 16), max_w=wp if not original else
                             int(wp * max(scales) + 16), max_peaks_per_part=64, device=local)
    pipe = PosePipeline(model, post)
    if original:
        from posepaf.original_path import OriginalPathProcessor, resize_images_u8
        from posepaf.pipeline import preprocess_batch
    local_recs = torch.zeros((max(S, 1), RECORD_BYTES), dtype=torch.uint8, device=dev)
    scale = torch.tensor(1e-3, dtype=torch.float16, device=dev)

    # untimed warm-up on each bucket shape: MIOpen's per-shape algorithm search happens on a shape's first call
    with torch.no_grad():
        for key in groups:
            if original:
                warm = torch.zeros((B,) + key + (3,), dtype=torch.uint8, device=dev)
                for sc in scales:
                    model(preprocess_batch(resize_images_u8(warm, float(sc)), True, torch.float16))
            else:
                pipe.forward_maps(torch.full((B,) + key + (3,), 128, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for key, members in groups.items():
        proc = OriginalPathProcessor(post, key[0], key[1], B) if original else None
        for b0 in range(0, len(members), B):
            loc = members[b0:b0 + B]                         # positions in this rank's local order
            idx = [int(mine[k]) for k in loc]                # global image indices
            n = len(idx)
            if original:
                H, W = key
                imgs = np.stack([load(i) for i in idx] + [np.zeros((H, W, 3), np.uint8)] * (B - n))
                dev_imgs = torch.from_numpy(imgs).to(dev, non_blocking=True)
                with torch.no_grad():
                    proc.reset()
                    for sc in scales:
                        scaled = resize_images_u8(dev_imgs, float(sc))
                        sh, sw = scaled.shape[1:3]
                        x = preprocess_batch(scaled, True, torch.float16)
                        ph, pw = x.shape[1:3]
                        maps = model(x).contiguous().view(B, 2, 50, ph // 4, pw // 4)
                        if inject_for is not None:   # the same synthetic people, rendered at this scale
                            szs = [(ph // 4, pw // 4, float(sc))]
                            inj = np.stack([synth.make_scene_at_scales(a.people[i % len(a.people)], 20_000 + i % 64, szs,
                                                                       img=H)[0][0] for i in idx] +
                                           [np.zeros((2, 50, ph // 4, pw // 4), np.float16)] * (B - n))
                            maps = torch.addcmul(torch.from_numpy(inj).to(dev), maps, scale)
                        proc.accumulate(maps, ph - sh, pw - sw, len(scales))
                    rec = proc.finish(B)
            else:
                Hp, Wp = key
                imgs = np.full((B, Hp, Wp, 3), 128, np.uint8)          # padValue 128 (utils/util.py:44-65), written on the host
                heights = np.full(B, Hp, np.int32)
                for j, i in enumerate(idx):
                    im = load(i)
                    imgs[j, :im.shape[0], :im.shape[1]] = im
                    heights[j] = im.shape[0]                            # `img_h` of process_paf / find_connections, per image
                dev_imgs = torch.from_numpy(imgs).to(dev, non_blocking=True)
                h_dev = torch.from_numpy(heights).to(dev, non_blocking=True)
                maps = pipe.forward_maps(dev_imgs)
                if inject_for is not None:
                    inj = np.stack([inject_for(i)[0] for i in idx] + [np.zeros((2, 50, Hp // 4, Wp // 4), np.float16)] * (B - n))
                    maps = torch.addcmul(torch.from_numpy(inj).to(dev), maps, scale)
                # --run_cpp: pafprocess rules (evaluate.py:105-129); without it: find_connections + find_humans (:88-89, :130-156)
                rec = post.process_async(maps, Hp, True, min_img_size_dev=h_dev) if a.run_cpp else \
                    post.process_py_async(maps, Hp, True, img_height_dev=h_dev)
            local_recs.index_copy_(0, torch.tensor(loc, dtype=torch.int64, device=dev), rec.view(B, RECORD_BYTES)[:n])
    local_recs = local_recs.view(-1)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0

    if world > 1:
        shard = local_recs[:S * RECORD_BYTES]
        merged = pdist.gather_records(shard if backend == "nccl" else shard.cpu(), len(mine))
    else:
        from posepaf.api import records_to_numpy
        merged = records_to_numpy(local_recs[:len(mine) * RECORD_BYTES])

    if rank == 0:
        results, dts = [], {}
        for i, rec in enumerate(merged):
            if int(rec["status"]) & 32:      # PP_ST_FLOAT_COORDS (original path): x / y are float32 bit patterns
                rec = rec.copy()
                fx, fy = rec["humans"]["x"].view(np.float32), rec["humans"]["y"].view(np.float32)
                rec["humans"]["x"], rec["humans"]["y"] = np.rint(fx).astype(np.int32), np.rint(fy).astype(np.int32)
            humans = coco.humans_from_record(rec)
            res = coco.coco_results(image_ids[i], humans)     # evaluate.py:182-209
            results.extend(res)
            dts[image_ids[i]] = [{"keypoints": r["keypoints"], "score": r["score"]} for r in res]
            if inject_for is not None:
                joints = synth.make_scene_at_scales(a.people[i % len(a.people)], 20_000 + i % 64, [(8, 8, 1.0)],
                                                    img=shape_of(i)[0])[1] if original else inject_for(i)[1]
                gts[image_ids[i]] = oks_eval.gt_from_synth_joints(joints)
        with open(a.dump_name, "w") as f:
            json.dump(results, f)
        summary = {"images": int(n_images), "world": world, "images_per_sec_rank0": len(mine) / dt_local,
                   "people_found": len(results), "status_or": int(np.bitwise_or.reduce(merged["status"])) if len(merged) else 0}
        if gts:
            summary["synthetic_oks"] = oks_eval.evaluate_keypoints(gts, dts)
        print(json.dumps(summary))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
