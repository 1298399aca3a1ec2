"""The batch engine of the refactored path: uint8 images in pinned host memory -> person records in HBM.

ONE engine serves `bench.py` (BASELINE's throughput line) and `evaluate.py` (the reference's evaluation loop,
evaluate.py:235-280), so the number the benchmark reports is the speed of the loop a user runs:

    pinned host slot --(copy stream, H2D)--> staging ring --(HBM -> HBM)--> the plan's input buffer
        --> [HIP graph of the plan: pad / 255 / mirror (pp_preprocess_u8_ragged) -> IMHN forward (fp16) ->
             (synthetic runs only: + scene bank[idx]) -> K_A -> K_B (+ assembly)] --> pp_record[b] of the plan

A *plan* belongs to one bucket `(Hp, Wp, b)`: `b` images whose sizes pad to the same multiple-of-64 shape
(utils/parse_skeletons.py:54, utils/util.py:44-65).  The reference runs every image alone at its own size; bucketing by
padded shape keeps its arithmetic (each image still sees exactly its own padded pixels) and lets a batch share launches.
Per-image height (`img_h` of process_paf, evaluate.py:110) travels in the slot header and is read on the device.

Host side: `acquire()` hands out pinned slots, any thread fills `slot.images(...)`, `slot.sizes`, `slot.bank_idx`, the
engine thread calls `submit()`, which never blocks on the GPU: the upload of batch k+1 rides under the compute of batch
k.  A slot is handed back at submit time together with its upload event; the next user waits on that event (not on the
compute) before writing.

There is no CPU path: the engine needs a HIP device and libposepaf.so.
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from . import skeleton as sk
from ._lib import RECORD_BYTES, PosePafError
from .fused_model import to_planes


def padded_shape(h: int, w: int, mult: int = sk.MAX_DOWNSAMPLE):
    """utils/util.py:44-65 padRightDownCorner: bottom / right up to the next multiple of `mult`"""
    return -(-h // mult) * mult, -(-w // mult) * mult


def header_bytes(b: int) -> int:
    """slot header: int32 heights[b], int32 widths[b], int64 bank_idx[b]; rounded to 256 B so the images stay aligned"""
    return -(-(16 * b) // 256) * 256


class Slot:
    """One pinned host batch: header (sizes, bank indices) followed by the images of a (b, Hp, Wp, 3) bucket."""

    def __init__(self, nbytes: int, index: int):
        self.index = index
        self.flat = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.np = self.flat.numpy()
        self.uploaded = None          # torch.cuda.Event of the last upload from this slot (None: never used)

    def wait_host_writable(self):
        if self.uploaded is not None:
            self.uploaded.synchronize()

    def views(self, b: int, hp: int, wp: int):
        """-> (sizes int32 (2, b), bank_idx int64 (b,), images uint8 (b, hp, wp, 3)) numpy views of the pinned bytes"""
        hb = header_bytes(b)
        sizes = self.np[: 8 * b].view(np.int32).reshape(2, b)
        idx = self.np[8 * b: 16 * b].view(np.int64)
        imgs = self.np[hb: hb + b * hp * wp * 3].reshape(b, hp, wp, 3)
        return sizes, idx, imgs


class _Plan:
    def __init__(self, eng, hp: int, wp: int, b: int):
        dev = eng.dev
        self.hp, self.wp, self.b = hp, wp, b
        self.nbytes = header_bytes(b) + b * hp * wp * 3
        self.flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.sizes = self.flat[: 8 * b].view(torch.int32).view(2, b)
        self.bank_idx = self.flat[8 * b: 16 * b].view(torch.int64)
        self.images = self.flat[header_bytes(b):].view(b, hp, wp, 3)
        self.sizes[0].fill_(hp)
        self.sizes[1].fill_(wp)
        self.records = torch.zeros(b * RECORD_BYTES, dtype=torch.uint8, device=dev)
        self.bank = None              # (K, 2, 50, hp/4, wp/4) fp16 synthetic scenes, or None
        self.graph = None
        self.maps = None              # the plan's last network output (kept for checks)


class InferenceEngine:
    """model: FusedIMHN-like module (NHWC fp16 in [0,1] -> (N, 50, h/4, w/4)); post: PosePostProcessor.
    rules: "cpp" = pafprocess rules (evaluate.py --run_cpp), "py" = find_connections + find_humans rules.
    inject_scale: synthetic runs add scene_bank[idx] to `inject_scale * network output` (a randomly initialised network emits
    no peaks); None = the network output alone (real weights)."""

    def __init__(self, model, post, batch: int, device: int, rules: str = "cpp", use_graph: bool = True,
                 inject_scale: float | None = None, max_image_hw=(512, 512), n_slots: int = 3, n_staging: int = 2,
                 postproc_only: bool = False, progress=None):
        if not torch.cuda.is_available():
            raise PosePafError("no HIP device: the engine has no CPU path")
        if rules not in ("cpp", "py"):
            raise PosePafError(f"rules must be 'cpp' or 'py', got {rules!r}")
        self.model, self.post, self.B, self.rules = model, post, int(batch), rules
        self.dev = torch.device("cuda", device)
        self.use_graph, self.postproc_only, self.progress = use_graph, postproc_only, progress
        self.inject_scale = None if inject_scale is None else torch.tensor(inject_scale, dtype=torch.float16, device=self.dev)
        hp, wp = padded_shape(*max_image_hw)
        self.max_bytes = header_bytes(self.B) + self.B * hp * wp * 3
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.staging = [torch.empty(self.max_bytes, dtype=torch.uint8, device=self.dev) for _ in range(n_staging)]
        self.consumed = [torch.cuda.Event() for _ in range(n_staging)]
        self.slots = [Slot(self.max_bytes, i) for i in range(n_slots)]
        self.free = queue.Queue()
        for s in self.slots:
            self.free.put(s)
        self.plans: dict = {}
        self.pool = None
        self.n_submit = 0
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ plans
    def plan(self, hp: int, wp: int, b: int | None = None) -> _Plan:
        b = self.B if b is None else int(b)
        key = (hp, wp, b)
        p = self.plans.get(key)
        if p is None:
            if b > self.B or header_bytes(b) + b * hp * wp * 3 > self.max_bytes:
                raise PosePafError(f"bucket {key} exceeds the engine's slot size (batch {self.B}, {self.max_bytes} bytes)")
            p = self.plans[key] = _Plan(self, hp, wp, b)
        return p

    def set_bank(self, plan: _Plan, scenes):
        """scenes: (K, 2, 50, hp/4, wp/4) float16 (numpy, or a device tensor shared between plans) -- the synthetic scene
        bank of this plan (before prepare())"""
        if plan.graph is not None:
            raise PosePafError("set_bank() after prepare(): the bank is part of the captured graph")
        want = (2, sk.NUM_CH, plan.hp // 4, plan.wp // 4)
        if isinstance(scenes, np.ndarray):
            scenes = torch.from_numpy(np.ascontiguousarray(scenes))
        if tuple(scenes.shape[1:]) != want or scenes.dtype != torch.float16:
            raise PosePafError(f"scene bank must be (K,) + {want} float16, got {tuple(scenes.shape)} {scenes.dtype}")
        plan.bank = scenes.to(self.dev).contiguous()

    def _body(self, p: _Plan):
        import ctypes as C
        from . import _lib
        b, hp, wp = p.b, p.hp, p.wp
        if self.postproc_only:
            maps = p.bank.index_select(0, p.bank_idx)
        else:
            x = torch.empty((2 * b, hp, wp, 3), dtype=torch.float16, device=self.dev)
            _lib.check(_lib.load().pp_preprocess_u8_ragged(
                C.c_void_p(p.images.data_ptr()), C.c_void_p(p.sizes.data_ptr()), C.c_void_p(x.data_ptr()), _lib.PP_F16, b, hp, wp,
                sk.PAD_VALUE, 1, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
            out = self.model(x)
            out = out[-1][0] if isinstance(out, (list, tuple)) else out
            maps = to_planes(out).view(b, 2, sk.NUM_CH, hp // 4, wp // 4)
            if p.bank is not None:
                maps = torch.addcmul(p.bank.index_select(0, p.bank_idx), maps, self.inject_scale)
        p.maps = maps
        heights = p.sizes[0]
        if self.rules == "cpp":
            self.post.process_async(maps, hp, True, min_img_size_dev=heights, records=p.records)
        else:
            self.post.process_py_async(maps, hp, True, img_height_dev=heights, records=p.records)
        return p.records

    def prepare(self, plan: _Plan, warmup: int = 2):
        """Eager warm-up of the plan (tunes every convolution shape of this batch geometry once, see fused_model), then
        capture of the whole per-batch path into one HIP graph.  Untimed set-up work."""
        if plan.graph is not None or not self.use_graph:
            if not self.use_graph and plan.maps is None:
                with torch.no_grad():
                    self._body(plan)
            return plan
        if self.progress:
            self.progress(f"plan {plan.hp}x{plan.wp} x{plan.b}: warm-up + HIP graph capture")
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._body(plan)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()   # plans replay one after another: they can share one memory pool
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool), torch.no_grad():
            self._body(plan)
        plan.graph = g
        return plan

    # ------------------------------------------------------------------ ingest + run
    def acquire(self) -> Slot:
        """A pinned slot whose previous upload has completed (blocks the CALLING thread only)."""
        s = self.free.get()
        s.wait_host_writable()
        return s

    def submit(self, slot: Slot, plan: _Plan, recycle: bool = True):
        """Enqueue upload + compute of the batch in `slot`; returns the plan's device record buffer (b * RECORD_BYTES uint8),
        valid on the compute stream until the plan's next submit.  Never blocks on the GPU."""
        k = self.n_submit % len(self.staging)
        self.n_submit += 1
        cur = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[k])         # compute has finished reading this staging buffer
            self.staging[k][: plan.nbytes].copy_(slot.flat[: plan.nbytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            slot.uploaded = ev
        if recycle:
            self.free.put(slot)
        cur.wait_event(ev)                                         # the batch has landed in HBM
        plan.flat.copy_(self.staging[k][: plan.nbytes], non_blocking=True)   # into the graph's input buffer (HBM -> HBM)
        self.consumed[k].record(cur)
        return self.run_resident(plan)

    def run_resident(self, plan: _Plan):
        """The compute part alone, on whatever already sits in the plan's input buffer."""
        if plan.graph is not None:
            plan.graph.replay()
            return plan.records
        with torch.no_grad():
            return self._body(plan)

    def sync(self):
        torch.cuda.synchronize(self.dev)


class BatchFeeder:
    """Background loader: `depth` producer threads fill pinned slots (each image of a batch by a pool of `workers` decode
    threads) and hand (job, slot) pairs to the engine thread.  jobs: list of (plan, [items]); fill(item, j, sizes, idx, imgs)
    writes item j of the batch into the slot views."""

    def __init__(self, engine: InferenceEngine, jobs, fill, workers: int = 8, depth: int = 2):
        from concurrent.futures import ThreadPoolExecutor
        self.eng, self.jobs, self.fill = engine, list(jobs), fill
        self.out = queue.Queue(maxsize=max(1, depth))
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.next = 0
        self.lock = threading.Lock()
        self.error = None
        self.threads = [threading.Thread(target=self._produce, daemon=True) for _ in range(max(1, depth))]

    def start(self):
        for t in self.threads:
            t.start()
        return self

    def _produce(self):
        try:
            while True:
                with self.lock:
                    j = self.next
                    self.next += 1
                if j >= len(self.jobs):
                    return
                plan, items = self.jobs[j]
                slot = self.eng.acquire()
                sizes, idx, imgs = slot.views(plan.b, plan.hp, plan.wp)
                sizes[0, :] = plan.hp        # unused tail entries of a short batch: a blank full-size image, bank scene 0
                sizes[1, :] = plan.wp
                idx[:] = 0
                list(self.pool.map(lambda t: self.fill(t[1], t[0], sizes, idx, imgs), enumerate(items)))
                self.out.put((j, slot))
        except BaseException as e:   # surfaced by the consumer
            self.error = e
            self.out.put((None, None))

    def __iter__(self):
        for _ in range(len(self.jobs)):
            j, slot = self.out.get()
            if j is None:
                raise self.error
            yield j, self.jobs[j][0], self.jobs[j][1], slot

    def close(self):
        self.pool.shutdown(wait=False)
