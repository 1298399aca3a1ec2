#!/usr/bin/env python3
"""evaluate.py -- the reference's evaluation loop (evaluate.py:235-330) on the MI355X-native path.

    python evaluate.py --run_refactor --run_cpp --synthetic 5000 [--batch 128] [--dump_name results.json]
    python evaluate.py --run_refactor --run_cpp -p weights.pth --ann_file person_keypoints_val2017.json --img_dir val2017
    python evaluate.py --gpus 8 --run_refactor --run_cpp --synthetic 5000          # launches its own 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --synthetic 5000 ...

What is kept from the reference: the flags --run_refactor / --run_cpp (:53-54; --run_cpp selects the C++ pafprocess rules --
the README's 65.8-AP / 7.3-fps row -- and its absence the pure-Python rules), the COCO annotation input with the
person-image filter (:237-254), the per-image result format (append_result :182-209), the COCO-style JSON dump (:269-270)
and the keypoint AP against the annotation's ground truth (:274-279, here posepaf/oks_eval.py because pycocotools is absent).

What changes: the refactored path runs on posepaf.engine.InferenceEngine -- THE SAME engine bench.py times: images are
decoded by background threads straight into pinned slots, uploaded on a copy stream under the previous batch's compute,
padded / normalised / mirrored by pp_preprocess_u8_ragged, and forward + post-processing replay from one HIP graph per
bucket.  Images are sharded i mod W over the ranks; the only exchange is one RCCL all-gather of the fixed-size records.
Reported speed = all images / MAX-over-ranks wall time of the loop (loading, upload, compute, record gather included).

Image sizes differ (COCO val2017 does): the reference pads every image to a multiple of 64 (utils/parse_skeletons.py:54,
utils/util.py:44-65) and runs it alone; here a rank's images are BUCKETED by padded shape and each bucket runs in batches;
the per-image `img_h` of process_paf (evaluate.py:110) travels with the batch as a device array.

Data sources:
  --ann_file F --img_dir D   COCO keypoint annotation JSON (read with `json`) + image directory (decoded with PIL, RGB -> BGR
                             like cv2.imread; libjpeg builds may differ in the last bit: unpinned)
  --images DIR               a directory of image files or .npy BGR uint8 arrays, no ground truth
  --synthetic N              random uint8 images; ground-truth-style pose scenes are injected into the network output (a
                             randomly initialised network emits no peaks) and serve as ground truth for the OKS evaluation
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
from posepaf.engine import padded_shape  # noqa: E402

IMAGE_EXT = (".jpg", ".jpeg", ".png", ".bmp")
SCENE_BANK = 64          # distinct synthetic scenes per bucket shape (image i shows scene i mod 64)


def parse(argv=None):
    ap = argparse.ArgumentParser(description="PoseNet evaluation (MI355X-native path)")
    ap.add_argument("--run_refactor", action="store_true")
    ap.add_argument("--run_cpp", action="store_true")
    ap.add_argument("--checkpoint_path", "-p", default=None, help="reference checkpoint (.pth with a 'weights' entry)")
    ap.add_argument("--ann_file", default=None, help="COCO keypoint annotation JSON (evaluate.py:237-246)")
    ap.add_argument("--img_dir", default=None, help="directory of the annotation's images (evaluate.py:263)")
    ap.add_argument("--all_images", action="store_true", help="with --ann_file: every image, not only those with a person "
                                                              "annotation (the reference's NUM_TEST_IMG > 0 branch, :247-248)")
    ap.add_argument("--inject_gt", action="store_true", help="with --ann_file and no trained weights: add GT-style maps rendered "
                                                             "from the annotation's keypoints to the network output")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic images")
    ap.add_argument("--sizes", default="512x512", help="synthetic image sizes HxW[,HxW...], cycled over the image index")
    ap.add_argument("--images", default=None, help="directory of image files / .npy BGR uint8 arrays (any sizes)")
    ap.add_argument("--limit", type=int, default=0, help="evaluate only the first N images of the set")
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without WORLD_SIZE: launch that many ranks (one per GPU)")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--workers", type=int, default=0, help="decode threads per rank (0 = CPU share of this rank, <= 16)")
    ap.add_argument("--no_graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--dump_name", default="results.json")
    ap.add_argument("--people", type=int, nargs="*", default=[1, 2, 3, 4, 6, 8, 10, 5])
    ap.add_argument("--scales", type=float, nargs="*", default=None,
                    help="without --run_refactor: predict's `multiplier` list (the reference hard-codes [1.0], "
                         "utils/parse_skeletons.py:188); e.g. 0.5 1.0 1.5 for BASELINE config 5")
    return ap.parse_args(argv)


def buckets_by_padded_shape(shapes):
    """indices grouped by padded (Hp, Wp), each group in ascending index order; groups ordered by first appearance"""
    out = {}
    for k, (h, w) in enumerate(shapes):
        out.setdefault(padded_shape(h, w), []).append(k)
    return out


def plan_batch(n_left: int, B: int) -> int:
    """batch of the plan that takes `n_left` (< B) trailing images of a bucket: B, B/2, B/4 ... >= 8, the smallest that fits"""
    b = B
    while b // 2 >= max(n_left, 8):
        b //= 2
    return b


def read_bgr(path: str) -> np.ndarray:
    """cv2.imread(path) (evaluate.py:72): BGR uint8 (H, W, 3)"""
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))[:, :, ::-1]


# ------------------------------------------------------------------------------------------------ data sources
class CocoSource:
    """The annotation file of evaluate.py:237-254 read with `json`: image list (person images only unless all_images),
    file names, sizes, and the ground truth COCOeval would use."""

    def __init__(self, ann_file, img_dir, all_images=False, limit=0, inject_gt=False):
        doc = json.load(open(ann_file))
        images = {im["id"]: im for im in doc["images"]}
        person = [c["id"] for c in doc.get("categories", []) if c.get("name") == "person"] or [1]
        anns = [a for a in doc.get("annotations", []) if a.get("category_id") in person]
        with_person = sorted({a["image_id"] for a in anns})          # getImgIds(catIds=person), :252-254
        ids = sorted(images) if all_images else with_person
        if limit:
            ids = ids[:limit]
        self.image_ids = ids
        self.files = [os.path.join(img_dir, images[i]["file_name"]) for i in ids]
        self._shapes = [(int(images[i]["height"]), int(images[i]["width"])) for i in ids]
        keep = set(ids)
        self.gts = {i: [] for i in ids}
        for a in anns:
            if a["image_id"] in keep and "keypoints" in a:
                self.gts[a["image_id"]].append({"keypoints": a["keypoints"], "area": float(a["area"]), "bbox": a.get("bbox"),
                                                "num_keypoints": int(a.get("num_keypoints", 0)),
                                                "iscrowd": int(a.get("iscrowd", 0))})
        self.has_scenes = bool(inject_gt)

    def __len__(self):
        return len(self.image_ids)

    def shape(self, i):
        return self._shapes[i]

    def load(self, i):
        im = read_bgr(self.files[i])
        if im.shape[:2] != self._shapes[i]:
            raise SystemExit(f"{self.files[i]}: decoded size {im.shape[:2]} != annotation {self._shapes[i]}")
        return im

    def bank_for_bucket(self, hp, wp, indices):
        """--inject_gt (no weights offline): one GT-style scene per image, rendered from the annotation's own keypoints
        (COCO order -> the 18 CMU parts, neck = shoulder midpoint as in the reference's data pipeline)."""
        from posepaf import skeleton as sk
        if len(indices) * 2 * 50 * (hp // 4) * (wp // 4) * 2 > 64 << 30:
            raise SystemExit("--inject_gt keeps one scene per image in HBM: use --limit")
        scenes = []
        for i in indices:
            people = []
            for g in self.gts[self.image_ids[i]]:
                kp = np.asarray(g["keypoints"], np.float64).reshape(17, 3)
                j = np.zeros((sk.NUM_PART, 3))
                j[:, 2] = 2
                for coco_i, cmu_i in enumerate(sk.ORDER_COCO):
                    if kp[coco_i, 2] > 0:
                        j[cmu_i] = (kp[coco_i, 0], kp[coco_i, 1], 1)
                if kp[5, 2] > 0 and kp[6, 2] > 0:
                    j[1] = ((kp[5, 0] + kp[6, 0]) / 2, (kp[5, 1] + kp[6, 1]) / 2, 1)
                if (j[:, 2] < 2).any():
                    people.append(j)
            base = synth.render_maps(np.stack(people) if people else np.zeros((0, sk.NUM_PART, 3)), hp // 4, wp // 4)
            scenes.append(np.stack([base, synth.mirror_sample(base)]).astype(np.float16))
        return np.stack(scenes), {int(i): k for k, i in enumerate(indices)}


class DirSource:
    def __init__(self, d, limit=0):
        files = sorted(f for f in glob.glob(os.path.join(d, "*")) if f.lower().endswith(IMAGE_EXT + (".npy",)))
        self.files = files[:limit] if limit else files
        self.image_ids = [os.path.splitext(os.path.basename(f))[0] for f in self.files]
        self.gts, self.has_scenes = {}, False
        self._shapes = {}

    def __len__(self):
        return len(self.files)

    def shape(self, i):
        if i not in self._shapes:
            f = self.files[i]
            if f.endswith(".npy"):
                self._shapes[i] = tuple(np.load(f, mmap_mode="r").shape[:2])
            else:
                from PIL import Image
                with Image.open(f) as im:       # header only
                    self._shapes[i] = (im.height, im.width)
        return self._shapes[i]

    def load(self, i):
        return read_bgr(self.files[i])


class SyntheticSource:
    """Random uint8 images; image i shows scene (i mod 64) of its bucket shape with people[i mod len(people)] persons."""

    def __init__(self, n, sizes, people):
        self.n, self.sizes, self.people = n, sizes, people
        self.image_ids = list(range(n))
        self.gts, self.has_scenes = {}, True
        self._scenes = {}

    def __len__(self):
        return self.n

    def shape(self, i):
        return self.sizes[i % len(self.sizes)]

    def load(self, i):
        h, w = self.shape(i)
        rng = np.random.default_rng(10_000 + i)
        return np.frombuffer(rng.bytes(h * w * 3), np.uint8).reshape(h, w, 3)

    def scene_slot(self, i):
        return i % SCENE_BANK

    def bank_for_bucket(self, hp, wp, indices):
        return np.stack([self.scene(s, hp, wp)[0] for s in range(SCENE_BANK)]), {int(i): self.scene_slot(int(i)) for i in indices}

    def n_people(self, slot):
        return self.people[slot % len(self.people)]

    def scene(self, slot, hp, wp):
        """(network-output-shaped scene, ground-truth joints) of bank slot `slot` at padded shape (hp, wp)"""
        key = (slot, hp, wp)
        if key not in self._scenes:
            self._scenes[key] = synth.make_scene(self.n_people(slot), 20_000 + slot, h=hp // 4, w=wp // 4, dtype=np.float16)
        return self._scenes[key]


# ------------------------------------------------------------------------------------------------ the two loops
def run_refactored(a, src, mine, model, post, dev, rank, world):
    """Refactored path on the shared engine.  -> (device uint8 records of this rank's images in local order, seconds)"""
    from posepaf import fused_model
    from posepaf.engine import BatchFeeder, InferenceEngine
    B = a.batch
    shapes = [src.shape(int(i)) for i in mine]
    groups = buckets_by_padded_shape(shapes)
    max_hw = (max([padded_shape(*s)[0] for s in shapes] + [64]), max([padded_shape(*s)[1] for s in shapes] + [64]))
    eng = InferenceEngine(model, post, B, dev.index, rules="cpp" if a.run_cpp else "py", use_graph=not a.no_graph,
                          inject_scale=1e-3 if src.has_scenes else None, max_image_hw=max_hw,
                          progress=(lambda m: print(f"[evaluate] {m}", file=sys.stderr, flush=True)) if rank == 0 else None)
    jobs = []
    for (hp, wp), members in groups.items():
        for b0 in range(0, len(members), B):
            loc = members[b0:b0 + B]
            jobs.append((eng.plan(hp, wp, B if len(loc) == B else plan_batch(len(loc), B)), loc))
    # untimed set-up: scene banks, kernel choice per layer shape (rank 0 tunes, the others take its table), graph capture
    bank_slot = {}
    if src.has_scenes:
        banks = {}
        for (hp, wp), members in groups.items():
            arr, where = src.bank_for_bucket(hp, wp, [int(mine[k]) for k in members])
            banks[(hp, wp)] = torch.from_numpy(arr).to(dev)
            bank_slot.update(where)
        for p in eng.plans.values():
            eng.set_bank(p, banks[(p.hp, p.wp)])
    fused_model.load_table()
    if world > 1:
        import torch.distributed as dist
        if rank == 0:
            for p in eng.plans.values():
                eng.prepare(p)
        box = [fused_model.table_entries() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        fused_model.install_entries(box[0])
    for p in eng.plans.values():
        eng.prepare(p)
    if rank == 0:
        fused_model.save_table()
    local = torch.zeros((max(len(mine), 1), RECORD_BYTES), dtype=torch.uint8, device=dev)

    def fill(k_local, j, sizes, idx, imgs):
        i = int(mine[k_local])
        im = src.load(i)
        h, w = im.shape[:2]
        imgs[j, :h, :w] = im
        sizes[0, j], sizes[1, j] = h, w
        if src.has_scenes:
            idx[j] = bank_slot[i]

    workers = a.workers or max(1, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8) // max(1, min(world, 8))))
    eng.sync()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    feeder = BatchFeeder(eng, jobs, fill, workers=workers, depth=2).start()
    for _, plan, loc, slot in feeder:
        rec = eng.submit(slot, plan)
        local.index_copy_(0, torch.as_tensor(loc, dtype=torch.int64).to(dev, non_blocking=True),
                          rec.view(plan.b, RECORD_BYTES)[: len(loc)])
    eng.sync()
    dt = time.perf_counter() - t0
    feeder.close()
    return local.view(-1), dt, {"plans": sorted(eng.plans), "decode_threads": workers,
                                "conv_table": fused_model.table_hash(), "launch": "eager" if a.no_graph else "hipGraph replay"}


def run_original(a, src, mine, model, post, dev):
    """evaluate.py:81-89 without --run_refactor: predict + find_peaks + find_connections + find_humans at image resolution,
    with a real scale search.  Accumulators live at image resolution, so images are bucketed by exact size; per batch the images
    are decoded by a thread pool into a pinned buffer (uploaded while the previous batch computes), every scale runs
    resize -> pad / normalise / mirror -> forward, and ALL scales are accumulated by one launch (pp_original_accumulate_all).
    Synthetic runs take their scenes from a device-resident bank per scale (64 scenes), like the refactored path."""
    from concurrent.futures import ThreadPoolExecutor
    from posepaf.original_path import OriginalPathProcessor, resize_images_u8, scaled_size
    from posepaf.pipeline import preprocess_batch
    B, scales = a.batch, a.scales or [1.0]
    shapes = [src.shape(int(i)) for i in mine]
    groups = {}
    for k, hw in enumerate(shapes):
        groups.setdefault(hw, []).append(k)
    local = torch.zeros((max(len(mine), 1), RECORD_BYTES), dtype=torch.uint8, device=dev)
    scale = torch.tensor(1e-3, dtype=torch.float16, device=dev)
    workers = a.workers or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
    pool = ThreadPoolExecutor(max_workers=workers)
    banks = {}
    with torch.no_grad():   # untimed set-up per image size: convolution shapes of every scale, scene banks
        for (H, W) in groups:
            warm = torch.zeros((B, H, W, 3), dtype=torch.uint8, device=dev)
            for sc in scales:
                x = preprocess_batch(resize_images_u8(warm, float(sc)), True, torch.float16)
                model(x)
                if src.has_scenes:
                    fh, fw = x.shape[1] // 4, x.shape[2] // 4
                    arr = np.stack([synth.make_scene_at_scales(src.n_people(s_), 20_000 + s_, [(fh, fw, float(sc))], img=H)[0][0]
                                    for s_ in range(SCENE_BANK)])
                    banks[(H, W, float(sc))] = torch.from_numpy(arr).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    copy_stream = torch.cuda.Stream(device=dev)
    for (H, W), members in groups.items():
        proc = OriginalPathProcessor(post, H, W, B)
        pinned = [torch.zeros((B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        staged = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        events = [None, None]
        batches = [members[b0:b0 + B] for b0 in range(0, len(members), B)]

        def stage(j):   # decode batch j into pinned[j % 2] and start its upload on the copy stream
            if j >= len(batches):
                return
            k = j % 2
            if events[k] is not None:
                events[k][1].synchronize()          # the compute that read staged[k] two batches ago has finished
            view = pinned[k].numpy()

            def put(t):
                view[t[0]] = src.load(int(mine[t[1]]))
            list(pool.map(put, enumerate(batches[j])))
            with torch.cuda.stream(copy_stream):
                staged[k].copy_(pinned[k], non_blocking=True)
                up = torch.cuda.Event()
                up.record(copy_stream)
            events[k] = [up, None]

        stage(0)
        for j, loc in enumerate(batches):
            k = j % 2
            idx = [int(mine[q]) for q in loc]
            n = len(idx)
            torch.cuda.current_stream(dev).wait_event(events[k][0])
            dev_imgs = staged[k]
            slots = torch.tensor([src.scene_slot(i) for i in idx] + [0] * (B - n), dtype=torch.int64, device=dev) if src.has_scenes else None
            with torch.no_grad():
                proc.reset()
                for sc in scales:
                    scaled = resize_images_u8(dev_imgs, float(sc))
                    sh, sw = scaled.shape[1:3]
                    x = preprocess_batch(scaled, True, torch.float16)
                    ph, pw = x.shape[1:3]
                    maps = model(x).contiguous().view(B, 2, 50, ph // 4, pw // 4)
                    if src.has_scenes:   # the same synthetic people, rendered at this scale
                        maps = torch.addcmul(banks[(H, W, float(sc))].index_select(0, slots), maps, scale)
                    proc.accumulate(maps, ph - sh, pw - sw, len(scales))
                rec = proc.finish(B)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            events[k][1] = done
            local.index_copy_(0, torch.tensor(loc, dtype=torch.int64, device=dev), rec.view(B, RECORD_BYTES)[:n])
            stage(j + 1)                            # the next batch is decoded and uploaded while this one computes
    torch.cuda.synchronize()
    pool.shutdown(wait=False)
    return local.view(-1), time.perf_counter() - t0, {"launch": "eager", "scales": scales, "decode_threads": workers}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:   # before anything touches the GPU; children are started, never exec'd
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                os.path.abspath(__file__)] + list(argv), env=env)
    original = not a.run_refactor   # evaluate.py:81-84: predict + find_peaks + find_connections + find_humans
    if original and a.run_cpp:
        raise SystemExit("--run_cpp only exists on the refactored path (evaluate.py:97-129)")
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
    from posepaf.api import PosePostProcessor, records_to_numpy
    from posepaf.fused_model import FusedIMHN
    from posepaf.model_init import deterministic_init

    opt, config = TrainingOpt(), GetConfig(TrainingOpt.config_name)
    net = NetworkEval(opt, config, bn=True).eval()
    if a.checkpoint_path:
        ckpt = torch.load(a.checkpoint_path, map_location="cpu", weights_only=True)
        net.load_state_dict(ckpt["weights"])       # evaluate.py:308-309 (strict)
    else:
        deterministic_init(net, 7)
    model = FusedIMHN.from_network(net).eval().to(dev).half().to(memory_format=torch.channels_last)

    # ---- data
    if a.ann_file:
        if not a.img_dir:
            raise SystemExit("--ann_file needs --img_dir")
        src = CocoSource(a.ann_file, a.img_dir, a.all_images, a.limit, a.inject_gt)
    elif a.images:
        src = DirSource(a.images, a.limit)
    else:
        sizes = [tuple(int(v) for v in t.lower().split("x")) for t in a.sizes.split(",")]
        src = SyntheticSource(a.limit or a.synthetic, sizes, a.people)
    n_images = len(src)
    if n_images == 0:
        raise SystemExit("nothing to evaluate: pass --ann_file F --img_dir D, --images DIR or --synthetic N")

    scales = a.scales or [1.0]
    mine = pdist.shard_indices(n_images, rank, world)
    S = pdist.padded_shard_size(n_images, world)
    all_shapes = [src.shape(int(i)) for i in (range(n_images) if n_images <= 200_000 else mine)]
    hp = max(padded_shape(h, w)[0] for h, w in all_shapes) // 4
    wp = max(padded_shape(h, w)[1] for h, w in all_shapes) // 4
    # feature maps up to hp x wp (area bound; pp_process_batch refuses a map that does not fit LDS, loudly)
    post = PosePostProcessor(max_batch=a.batch, max_h=hp if not original else int(hp * max(scales) + 16),
                             max_w=wp if not original else int(wp * max(scales) + 16), max_peaks_per_part=64, device=local)
    if original:
        local_recs, dt_local, info = run_original(a, src, mine, model, post, dev)
    else:
        local_recs, dt_local, info = run_refactored(a, src, mine, model, post, dev, rank, world)

    if world > 1:
        import torch.distributed as dist
        shard = torch.zeros(S * RECORD_BYTES, dtype=torch.uint8, device=dev)
        shard[: len(mine) * RECORD_BYTES] = local_recs[: len(mine) * RECORD_BYTES]
        merged = pdist.gather_records(shard if backend == "nccl" else shard.cpu(), len(mine))
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        merged = records_to_numpy(local_recs[:len(mine) * RECORD_BYTES])
        dt = dt_local

    if rank == 0:
        results, dts, gts = [], {}, dict(src.gts)
        for i, rec in enumerate(merged):
            if int(rec["status"]) & 32:      # PP_ST_FLOAT_COORDS (original path): x / y are float32 bit patterns
                rec = rec.copy()
                fx, fy = rec["humans"]["x"].view(np.float32), rec["humans"]["y"].view(np.float32)
                rec["humans"]["x"], rec["humans"]["y"] = np.rint(fx).astype(np.int32), np.rint(fy).astype(np.int32)
            humans = coco.humans_from_record(rec)
            res = coco.coco_results(src.image_ids[i], humans)     # evaluate.py:182-209
            results.extend(res)
            dts[src.image_ids[i]] = [{"keypoints": r["keypoints"], "score": r["score"]} for r in res]
            if src.has_scenes and not src.gts:
                slot = src.scene_slot(i)
                joints = synth.make_scene_at_scales(src.n_people(slot), 20_000 + slot, [(8, 8, 1.0)], img=src.shape(i)[0])[1] \
                    if original else src.scene(slot, *padded_shape(*src.shape(i)))[1]
                gts[src.image_ids[i]] = oks_eval.gt_from_synth_joints(joints)
        os.makedirs(os.path.dirname(os.path.abspath(a.dump_name)), exist_ok=True)
        with open(a.dump_name, "w") as f:
            json.dump(results, f)
        summary = {"images": int(n_images), "world": world, "images_per_sec": n_images / dt, "seconds": dt,
                   "images_per_gpu_per_step": a.batch, "rules": "original" if original else ("cpp" if a.run_cpp else "python"),
                   "people_found": len(results),
                   "status_or": int(np.bitwise_or.reduce(merged["status"])) if len(merged) else 0}
        summary.update(info)
        if gts:
            key = "keypoint_ap" if src.gts else "synthetic_oks"     # ground truth from the annotation file / from the injected scenes
            summary[key] = oks_eval.evaluate_keypoints(gts, dts)
        print(json.dumps(summary, default=str))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
