#!/usr/bin/env python3
"""bench.py -- end-to-end images/sec at 512x512 (BASELINE.json's metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

With --gpus N > 1 and no WORLD_SIZE in the environment the script LAUNCHES ITS OWN RANKS (one process per GPU through
`python -m torch.distributed.run`, rendezvous on 127.0.0.1) before anything touches the GPU, waits for them and exits with
their return code; under an outer torchrun (WORLD_SIZE set) it is one of the ranks.  Rank 0 prints ONE JSON line.

One "step" = one pass of the whole hot path over one batch of B synthetic 512x512 images per GPU:
pinned-host -> HBM upload of the uint8 batch (copy stream, double-buffered: the upload of batch k+1 overlaps the compute
of batch k; utils/parse_skeletons.py:73) -> uint8 pre-processing + flip -> 4-stage IMHN forward on (2B, 512, 512, 3) in
fp16 -> K_A heat-map NMS/refine -> K_BC limb line-integral scoring + matching + person assembly -> (N > 1) RCCL
all-gather of the per-image records.  Images shard across ranks (weak scaling: B per GPU); no other collective.

A randomly initialised network emits no peaks, so the post-processing load comes from synthetic ground-truth style scenes
(posepaf/synth.py) ADDED to the (down-scaled) network output: the forward is fully live and the kernels see realistic
peak/limb counts.  Nothing is skipped or cached inside the timed region.

`roofline` is for the longest hand-written post-processing kernel (HIP events on the launch stream) and carries the
chain-level fraction too; `roofline_forward` prices the step against the dense fp16 MFMA peak with the FLOPs the
inference model actually executes (tools/count_flops.py); `cpu_baseline` is the oracle (plain-C port of the reference
path) on all host cores, one image per core, with a per-stage single-core split.

POSEPAF_BENCH_STUB=1 replaces the GPU engine by a CPU stub (records filled by numpy) so that the control flow --
self-launch, process group, sharding, per-step exchange, MAX-over-ranks timing, status check, the JSON line -- can be
exercised on a machine without GPUs (tests/test_bench_control_flow.py, gloo, world size 2).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "improved-body-parts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

IMG = 512
FEAT = IMG // 4
SCENE_PEOPLE = (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 30, 2, 4, 6, 3)   # people per synthetic scene (mean 8.2)
HBM_PEAK_GBS = 8000.0                                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0                                           # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
FLOP_PER_FORWARD_REFERENCE = 529.39e9                                   # the reference module, 512x512
FLOP_PER_FORWARD_EXECUTED = 512.18e9                                    # tools/count_flops.py: the inference model (the last
#                                                                         stage's unused coarse heads are not computed)
FLOP_PER_IMAGE = 2 * FLOP_PER_FORWARD_EXECUTED                          # flip: two forwards per image
# the 3x3 convolutions behind a x2 upsample (4 stages x hourglass levels 0..3: 256 ch @128^2, 384 @64^2, 512 @32^2, 640 @16^2 outputs):
# 2 * H * W * C^2 * 9 each; the collapsed 2x2 form (posepaf/fused_model.py forward_up2) multiplies out 4/9 of that
UP2_FLOP_PER_FORWARD = 4 * 2 * 9 * (128 * 128 * 256 * 256 + 64 * 64 * 384 * 384 + 32 * 32 * 512 * 512 + 16 * 16 * 640 * 640)
UP2_FLOP_SAVED_PER_FORWARD = UP2_FLOP_PER_FORWARD * 5 / 9
ST_DEFINED, ST_SORT_UNDEFINED = 0x7F, 8                                 # include/posepaf.h:53-61


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("POSEPAF_BENCH_BATCH", "0")),
                    help="images per GPU per step (default 128; 32 with --multiscale)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="bound on the CPU baseline leg (all parts together)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = every host core this process may use)")
    ap.add_argument("--postproc-only", action="store_true", help="time only K_A..K_C (profiling aid)")
    ap.add_argument("--plain-model", action="store_true", help="unfused nn.Module forward instead of the fused one")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-ingest", action="store_true", help="leave the host->HBM upload out of the step (round-1 form)")
    ap.add_argument("--retune", action="store_true", help="ignore the saved kernel-choice table and time every layer shape again")
    ap.add_argument("--no-verify", action="store_true", help="skip the independent re-computation of the last batch")
    ap.add_argument("--multiscale", action="store_true",
                    help="BASELINE configs[4]: original path, scale search {0.5, 1.0, 1.5} x 512 + flip, float64 accumulation")
    a = ap.parse_args(argv)
    if a.batch <= 0:
        a.batch = 32 if a.multiscale else 128
    return a


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(a, argv):
    """--gpus N > 1 from a plain shell: start N ranks as CHILD processes (never exec: nothing here has touched the GPU, and
    nothing will in this parent), forward their output, return their exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU baseline
def build_scenes(batch, dtype=None):
    import numpy as np
    from posepaf import synth
    dtype = dtype or np.float16
    uniq = [synth.make_net_output(p, 9000 + i, dtype=dtype) for i, p in enumerate(SCENE_PEOPLE)]
    return np.stack([uniq[i % len(uniq)] for i in range(batch)]), uniq


_W_ORC, _W_SCENES = None, None


def _cpu_worker_init(scenes):
    global _W_ORC, _W_SCENES
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle.oracle import Oracle
    _W_ORC = Oracle()
    _W_SCENES = scenes


def _cpu_worker_run(args):
    """one worker = one core: whole post-processing of one image at a time until the deadline"""
    start, deadline = args
    n, k = 0, start
    while time.time() < deadline:
        _W_ORC.pipeline(_W_SCENES[k % len(_W_SCENES)], IMG)
        n += 1
        k += 1
    return n


def _time_loop(fn, seconds):
    n, t0 = 0, time.perf_counter()
    while True:
        fn(n)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds and n >= 3:
            return n / dt, n


def usable_cores():
    """host cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one"""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    return avail, quota


def cpu_baseline(seconds, cores=0, check=None):
    """The oracle -- the plain-C restatement of the reference's CPU path, pinned bit-exact against the reference's own
    compiled C++ and Python outputs (tests/test_oracle_*.py) -- timed on this machine's host cores, SURVEY.md 8(d):
      value      : whole post-processing (flip-average, heatmap_nms with x4 patch refinement, x4 limb upsample, process_paf),
                   one image per core on all cores, bench.py's scene mix;
      stages     : images/sec of each stage on ONE core, incl. the pure-Python rule set (find_connections + find_humans,
                   what the reference runs without --run_cpp) as its C port.
    The network forward is not part of it (fp32 CPU forward of the reference module: ~6.5 s per image on 8 threads,
    SURVEY.md section 6)."""
    import multiprocessing as mp
    import numpy as np
    from oracle.oracle import Oracle
    avail, quota = usable_cores()
    cores = cores or (min(avail, quota) if quota else avail)
    orc = Oracle()
    _, uniq = build_scenes(len(SCENE_PEOPLE))
    nu = len(uniq)
    part = seconds / 8.0
    # ---- single-core stage split
    single, _ = _time_loop(lambda i: orc.pipeline(uniq[i % nu], IMG), part)
    flips = [orc.flip_average(s) for s in uniq]
    st_flip, _ = _time_loop(lambda i: orc.flip_average(uniq[i % nu]), part / 2)
    st_nms, _ = _time_loop(lambda i: orc.heatmap_nms(flips[i % nu][0]), part)
    jls = [orc.heatmap_nms(f[0])[0] for f in flips]
    st_up, _ = _time_loop(lambda i: orc.upsample4_hwc(flips[i % nu][1]), part / 2)
    mid = [5, 7, 9, 11]                                    # 6, 10, 15 and 30 people (mean 15): four 31.5 MB up-sampled maps
    ups = [orc.upsample4_hwc(flips[k][1]) for k in mid]
    st_cpp, _ = _time_loop(lambda i: orc.process_paf(jls[mid[i % 4]][None], ups[i % 4], IMG), part)
    st_py, _ = _time_loop(lambda i: orc.py_find_humans(jls[mid[i % 4]], ups[i % 4], IMG), part)
    del ups
    # ---- all cores, one image per core
    par_seconds = max(2.0, seconds - 5.5 * part)
    ctx = mp.get_context("spawn")      # never fork a process that holds a HIP context
    with ctx.Pool(cores, initializer=_cpu_worker_init, initargs=(uniq,)) as pool:
        pool.map(_cpu_worker_run, [(i, time.time() + 0.2) for i in range(cores)])          # start-up + first touch
        t0 = time.time()
        counts = pool.map(_cpu_worker_run, [(3 * i, t0 + par_seconds) for i in range(cores)])
        dt = time.time() - t0
    n = int(np.sum(counts))
    checked = None
    if check is not None:   # the oracle as the CHECKER of the timed GPU run: same maps in, records must be equal
        maps, recs = check
        checked = "equal"
        for i in range(len(maps)):
            if int(recs[i]["status"]) & ST_SORT_UNDEFINED:   # the reference's own result is undefined there
                continue
            want = orc.pipeline(maps[i], IMG)
            k = int(recs[i]["n_humans"])
            if k != len(want["ids"]) or not np.array_equal(recs[i]["humans"]["peak_id"][:k], want["ids"]) \
                    or not np.array_equal(recs[i]["humans"]["score"][:k], want["scores"]):
                checked = "DIFFERENT"
    return {"value": n / dt, "unit": "images/sec", "cores": cores, "host_cores_available": avail, "cgroup_cpu_quota": quota,
            "records_vs_oracle": checked,
            "records_vs_oracle_sample": None if check is None else f"first {len(check[0])} images of the last timed batch: person count, peak ids and scores equal",
            "kind": "port",
            "sample": f"{n} images ({nu} scenes of {min(SCENE_PEOPLE)}-{max(SCENE_PEOPLE)} people, cycled) in {dt:.1f} s on "
                      f"{cores} worker processes, one image per core; post-processing only (flip-average, NMS + x4 patch "
                      "refinement, x4 limb upsample, process_paf); network forward excluded",
            "single_core": {"value": single, "unit": "images/sec"},
            "stages_single_core_images_per_sec": {
                "flip_average": st_flip, "heatmap_nms": st_nms, "limb_upsample_x4": st_up,
                "process_paf_cpp_rules": st_cpp, "find_connections+find_humans_python_rules_as_C": st_py,
                "matching_stage_scenes": "6, 10, 15 and 30 people"},
            "note": "the Python-rule figure is the C port of utils/parse_skeletons.py:324-600; the reference's interpreted "
                    "NumPy version of the same stage runs at 5.2 FPS on the authors' machine (README.md:86)"}


# ------------------------------------------------------------------------------------------------ engines
class StubEngine:
    """CPU stand-in for the GPU engine (control-flow tests): records with recognisable content, no HIP."""

    name = "stub"

    def __init__(self, a, rank, world):
        import numpy as np
        import torch
        from posepaf._lib import RECORD_BYTES, RECORD_DTYPE
        self.torch, self.B, self.rank = torch, a.batch, rank
        rec = np.zeros(a.batch, RECORD_DTYPE)
        rec["n_humans"] = 1 + rank
        rec["n_peaks"] = 100 * rank + np.arange(a.batch)
        rec["status"] = int(os.environ.get("POSEPAF_BENCH_STUB_STATUS", "0"), 0)
        self.rec = torch.from_numpy(rec.view(np.uint8).copy())
        assert self.rec.numel() == a.batch * RECORD_BYTES
        self.gather_device = torch.device("cpu")
        self.steps_run = 0

    def step(self):
        self.steps_run += 1
        time.sleep(0.002)
        return self.rec

    def sync(self):
        pass

    def verify(self, rec):
        return {"ok": os.environ.get("POSEPAF_BENCH_STUB_VERIFY", "ok") != "fail", "stub": True}

    def oracle_sample(self, rec, n=8):
        return None

    def extras(self, a, dt, world):
        return {"stub": True, "steps_run": self.steps_run}


class GpuEngine:
    """bench.py's step on posepaf.engine.InferenceEngine -- the same engine improved-body-parts_amd/evaluate.py runs on."""
    name = "gpu"

    def __init__(self, a, rank, world, local, backend):
        import numpy as np
        import torch
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
        from posepaf import fused_model
        from posepaf.api import PosePostProcessor
        from posepaf.engine import InferenceEngine
        from posepaf.fused_model import build_inference_model
        from posepaf.pipeline import PosePipeline
        self.torch, self.a, self.B, self.rank, self.world = torch, a, a.batch, rank, world
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        self.dev = dev = torch.device("cuda", local)
        self.local = local
        # MIOpen exhaustive find (miopenFindConvolutionForwardAlgorithm) per conv shape: +30 % over the default heuristic
        if os.environ.get("POSEPAF_CUDNN_BENCHMARK", "1") == "1":
            torch.backends.cudnn.benchmark = True
        self.gather_device = dev if backend == "nccl" else torch.device("cpu")
        B = a.batch
        self.post = PosePostProcessor(max_batch=B, max_h=FEAT if not a.multiscale else 192, max_w=FEAT if not a.multiscale else 192,
                                      max_peaks_per_part=64, device=local)
        _, self.uniq = build_scenes(len(SCENE_PEOPLE))
        say = (lambda msg: print(f"[bench] {msg}", file=sys.stderr, flush=True)) if rank == 0 else None
        if say:   # the warm-up times every layer shape once: say so on stderr, one line per shape
            fused_model.set_progress(say)
            say(f"building the model; warm-up tunes each convolution shape at batch {a.batch} unless the choice table "
                f"{fused_model.default_table_path()} exists (under a minute)")
        self.model = None if a.postproc_only else build_inference_model(dev, fused=not a.plain_model)
        self.table_loaded = fused_model.load_table() if not a.retune else 0
        self.eng = InferenceEngine(self.model, self.post, B, local, rules="cpp", use_graph=not (a.no_graph or a.multiscale),
                                   inject_scale=1e-3, max_image_hw=(IMG, IMG), n_slots=2, postproc_only=a.postproc_only,
                                   progress=say)
        if a.multiscale:
            self.scale = torch.tensor(1e-3, dtype=torch.float16, device=dev)
            g = torch.Generator(device="cpu").manual_seed(1234 + rank)
            self.ms_host = [torch.randint(0, 256, (B, IMG, IMG, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
            self.static_images = self.ms_host[0].to(dev)
            self._init_multiscale()
            self.k = 0
            self.ms_graph, self.ms_rec = None, None
            with torch.no_grad():
                for _ in range(2):                       # eager: tunes the three geometries of every convolution shape
                    self._body_multiscale()
            torch.cuda.synchronize()
            if rank == 0 and not a.plain_model:
                fused_model.save_table()
            if not a.no_graph:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.no_grad():
                    self._body_multiscale()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.ms_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.ms_graph), torch.no_grad():
                    self.ms_rec = self._body_multiscale()
            return
        self.plan = self.eng.plan(IMG, IMG, B)
        self.eng.set_bank(self.plan, np.stack(self.uniq))
        # host side of the ingest: two different pinned batches, uploaded alternately
        g = np.random.default_rng(1234 + rank)
        self.slots = []
        for _ in range(2):
            slot = self.eng.acquire()
            sizes, idx, imgs = slot.views(B, IMG, IMG)
            sizes[:] = IMG
            idx[:] = np.arange(B) % len(self.uniq)
            imgs[:] = g.integers(0, 256, imgs.shape, dtype=np.uint8)
            self.slots.append(slot)
        # kernel choice per layer shape: rank 0 tunes (or loads the table), every rank runs rank 0's table
        if world > 1 and not a.plain_model and not a.postproc_only:
            import torch.distributed as dist
            if rank == 0:
                self.eng.prepare(self.plan)
            box = [fused_model.table_entries() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            fused_model.install_entries(box[0])
        self.eng.prepare(self.plan)
        if rank == 0 and not a.plain_model and not a.postproc_only:
            fused_model.save_table()
        self.inject = self.plan.bank.index_select(0, torch.from_numpy(np.arange(B) % len(self.uniq)).to(dev))
        self.k = 0
        self.eng.submit(self.slots[0], self.plan, recycle=False)     # the resident batch of --no-ingest
        self.eng.sync()

    def _init_multiscale(self):
        from posepaf import synth
        import numpy as np
        from posepaf.original_path import OriginalPathProcessor
        torch = self.torch
        self.ms_sizes = [(64, 64, 0.5), (128, 128, 1.0), (192, 192, 1.5)]
        self.proc = OriginalPathProcessor(self.post, IMG, IMG, self.B)
        uniq = [synth.make_scene_at_scales(p, 9100 + i, self.ms_sizes)[0] for i, p in enumerate(SCENE_PEOPLE)]
        self.ms_inject = [torch.from_numpy(np.stack([uniq[i % len(uniq)][k] for i in range(self.B)])).to(self.dev)
                          for k in range(len(self.ms_sizes))]

    def _body_multiscale(self):
        from posepaf.original_path import resize_images_u8
        from posepaf.pipeline import preprocess_batch
        torch = self.torch
        self.proc.reset()
        for k, (h, w, sc) in enumerate(self.ms_sizes):
            scaled = resize_images_u8(self.static_images, sc)
            x = preprocess_batch(scaled, True, torch.float16)
            maps = self.model(x).view(self.B, 2, 50, h, w)
            maps = torch.addcmul(self.ms_inject[k], maps, self.scale)
            self.proc.accumulate(maps, 0, 0, len(self.ms_sizes))
        return self.proc.finish(self.B)

    def _extras_multiscale(self, a, dt):
        """roofline of the scale-accumulation kernel (k_accumulate_scales: every scale's flip-average, x4 bicubic, crop, resize and
        the float64 accumulation in ONE launch) -- HIP events on the launch stream around that launch alone."""
        torch = self.torch
        B = a.batch
        maps = [torch.addcmul(inj, torch.zeros_like(inj), self.scale) for inj in self.ms_inject]   # scene maps of the three scales
        iters, ev = 10, [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for k in range(iters + 2):
            if k == 2:
                ev[0].record()
            self.proc.reset()
            for m in maps:
                self.proc.accumulate(m, 0, 0, len(maps))
            self.proc._flush()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / iters
        bytes_in = sum(m.numel() * m.element_size() for m in maps)                        # 2 * 50 * (64^2 + 128^2 + 192^2) * 2 B per image
        bytes_out = B * 50 * IMG * IMG * 8                                                  # float64 accumulators, written once
        alg = bytes_in + bytes_out
        ach = alg / (ms * 1e-3) / 1e9
        tf = B * a.steps * FLOP_PER_FORWARD_EXECUTED * 2 * (0.25 + 1.0 + 2.25) / dt / 1e12
        return {"kernel_ms": {"k_accumulate_scales": ms},
                "roofline": {"bound": "hbm", "kernel": "k_accumulate_scales", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
                             "note": "network output of the three scales read once (fp16) + the 105 MB per image of float64 "
                                     "accumulators written once; the kernel is bound by its bicubic arithmetic, not by HBM"},
                "roofline_forward": {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": tf / MFMA_F16_PEAK_TFLOPS,
                                     "note": "three forwards per flip sample (inputs 256^2, 512^2, 768^2 = 3.5x the FLOPs of one 512^2 "
                                             "forward) over the whole step time"}}

    def step(self):
        if self.a.multiscale:
            if not self.a.no_ingest:   # the uint8 batch comes from pinned host memory every step (25 MB at 32 images)
                self.static_images.copy_(self.ms_host[self.k % 2], non_blocking=True)
                self.k += 1
            if self.ms_graph is not None:
                self.ms_graph.replay()
                return self.ms_rec
            with self.torch.no_grad():
                return self._body_multiscale()
        if self.a.no_ingest:
            return self.eng.run_resident(self.plan)
        rec = self.eng.submit(self.slots[self.k % 2], self.plan, recycle=False)   # batch k+1's upload rides under batch k's compute
        self.k += 1
        return rec

    def sync(self):
        self.torch.cuda.synchronize()

    def verify(self, rec):
        """Independent re-computation of the LAST batch (a status word alone does not prove the records):
          (1) the post-processing of the batch's network output once more with the person assembly as its OWN launch
              (pp_debug_set_mode 1, eager) -- every record byte must equal what the graph replay produced;
          (2) the fused fp16 forward (kernels picked by the tuner, as replayed) against the plain nn.Module on PyTorch-ROCm
              (MIOpen convolutions, BatchNorm unfolded) for the first two images of the batch;
          (3) the fused forward of the whole batch twice: bit-identical outputs (a race detector: the hand-counted LDS-DMA of the
              convolution kernels goes wrong only under load, and then differently every time).
        -> dict for the JSON line; ["ok"] False invalidates the run."""
        torch = self.torch
        import numpy as np
        a, p = self.a, getattr(self, "plan", None)
        if a.multiscale or p is None:
            return {"ok": True, "skipped": "multi-scale line: records are checked by the GPU tests only"}
        out = {"ok": True}
        self.sync()
        got = rec.cpu().numpy().copy()
        maps = p.maps.clone()
        self.post.set_mode(1)
        again = torch.zeros_like(p.records)
        self.post.process_async(maps, IMG, True, min_img_size_dev=p.sizes[0].clone(), records=again)
        self.sync()
        self.post.set_mode(0)
        same = bool(np.array_equal(got, again.cpu().numpy()))
        out["records_vs_separate_assembly_launch"] = "equal" if same else "DIFFERENT"
        out["ok"] &= same
        if self.model is not None and not a.plain_model:
            from posepaf.fused_model import build_inference_model
            from posepaf.pipeline import preprocess_batch
            nchk = min(2, p.b)
            with torch.no_grad():
                x = preprocess_batch(p.images, True, torch.float16)
                whole = self.model(x).clone()
                again = self.model(x)
                # (the kernels have no atomics on the data path: two passes over the same batch must agree to the bit -- a
                # difference means a kernel read operands that were still in flight)
                out["forward_twice_bit_identical"] = bool(torch.equal(whole, again))
                out["ok"] &= out["forward_twice_bit_identical"]
                fused = whole[: 2 * nchk].float()
                del whole, again
                bench_flag = torch.backends.cudnn.benchmark
                torch.backends.cudnn.benchmark = False
                plain = build_inference_model(self.dev, fused=False)
                ref = plain(x[: 2 * nchk].contiguous())[-1][0].float()
                torch.backends.cudnn.benchmark = bench_flag
                del plain
            err = float((fused - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
            out["forward_vs_plain_module_max_err_over_max"] = err
            out["forward_tolerance"] = 3e-2
            out["ok"] &= bool(err <= 3e-2) and bool(torch.isfinite(fused).all())
        return out

    def oracle_sample(self, rec, n=8):
        """(network output of the first n images of the last batch, their records) for the CPU-baseline leg's checker"""
        from posepaf.api import records_to_numpy
        p = getattr(self, "plan", None)
        if p is None or self.a.multiscale:
            return None
        self.sync()
        n = min(n, p.b)
        return p.maps[:n].cpu().numpy(), records_to_numpy(rec)[:n]

    def extras(self, a, dt, world):
        """roofline of the hand-written kernels (HIP events on the launch stream) + forward roofline"""
        B = a.batch
        out = {}
        if a.multiscale:
            return self._extras_multiscale(a, dt)
        ms = self.post.time_kernels(self.inject, IMG, True, iters=20)
        # algorithmic bytes per launch: every flip sample of every channel the kernel consumes, fp16, once
        # the chain is two launches: k_heat_peaks (+ image ordering) and k_limb_connect (+ the person assembly by each image's
        # last limb workgroup); k_assemble_wave in kernel_ms is the assembly alone as its own launch (diagnostic only)
        alg = {"k_heat_peaks": B * 18 * 2 * FEAT * FEAT * 2, "k_limb_connect": B * 30 * 2 * FEAT * FEAT * 2}
        dom = max(alg, key=lambda k: ms[k])
        chain_bytes = sum(alg.values())
        achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
        chain = chain_bytes / (ms["chain"] * 1e-3) / 1e9
        traffic, src = None, None
        for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get(f"batch{B}")
                    src = f"profiles/{name} (rocprofv3 --pmc passes of tools/pmc_traffic.py; not re-measured in this run)"
                except Exception:
                    traffic = None
                if traffic is not None:
                    break
        out["kernel_ms"] = ms
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                           "algorithmic_bytes_per_launch": alg[dom],
                           "chain": {"kernels": sorted(alg), "ms": ms["chain"],
                                     "algorithmic_bytes": chain_bytes, "achieved": chain, "frac": chain / HBM_PEAK_GBS}}
        if not a.postproc_only:
            tf = B * a.steps * FLOP_PER_IMAGE / dt / 1e12
            out["forward_tflops_per_gpu"] = tf
            out["roofline_forward"] = {
                "bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS,
                "flop_per_forward": FLOP_PER_FORWARD_EXECUTED,
                "flop_per_forward_multiplied_out": FLOP_PER_FORWARD_EXECUTED - UP2_FLOP_SAVED_PER_FORWARD,
                "note": "whole step time (ingest + forward + pre/post-processing) against the dense fp16 MFMA peak; ALGORITHMIC FLOPs: "
                        "those of the inference model's convolutions as the reference defines them (512.18 GFLOP per 512x512 forward, "
                        "tools/count_flops.py; the reference module's 529.39 include last-stage coarse heads that are computed and "
                        "discarded).  Since round 3 the hourglass' conv3x3(upsample2(x)) layers are evaluated as four 2x2 convolutions "
                        "of x (per-phase tap sums: the same real-number result with 2.25x fewer multiply-adds), so the matrix units "
                        "multiply out about flop_per_forward_multiplied_out per forward when every such site takes that form"}
            if not a.plain_model:
                from posepaf import fused_model
                ch = fused_model.conv_choices()
                # (the table may hold other geometries -- multi-scale runs, other batch sizes: count this step's 2 B samples only)
                up2 = {k: v for k, v in ch.items() if k[0] == "up2" and k[1] == 2 * B}     # upsample -> 3x3 -> add(s): 1 = one launch of the halo kernel
                dual = {k: v for k, v in ch.items() if k[0] == "dual" and k[1] == 2 * B}   # convolution with a second output y + other: 0 = separate add
                side = {k: v for k, v in ch.items() if k[0] in ("pool", "mean") and k[1] == 2 * B}   # pooled output / SE channel sums from the producer's epilogue
                ch = {k: v for k, v in ch.items() if isinstance(k[0], int) and k[0] == 2 * B}
                out["conv_layers"] = {"shapes_own_kernel": sum(1 for v in ch.values() if v >= 100),
                                      "shapes_ck_template_kernel": sum(1 for v in ch.values() if 0 <= v < 100),
                                      "shapes_miopen_plus_epilogue": sum(1 for v in ch.values() if v < 0),
                                      "upsample_conv_add_sites_fused": sum(1 for v in up2.values() if v),
                                      "upsample_conv_add_sites_separate": sum(1 for v in up2.values() if not v),
                                      "two_output_conv_sites_fused": sum(1 for v in dual.values() if v),
                                      "pooled_or_channel_sum_outputs_fused": sum(1 for v in side.values() if v),
                                      "shapes_streaming_1x1_kernel": sum(1 for v in ch.values() if v == 105),
                                      "choice_table_hash": fused_model.table_hash(),
                                      "choice_table": "loaded from " + fused_model.default_table_path() if self.table_loaded
                                      else "tuned in this run's warm-up"}
        return out


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, argv)            # before ANY GPU call in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    stub = os.environ.get("POSEPAF_BENCH_STUB", "0") == "1"
    backend = os.environ.get("POSEPAF_DIST_BACKEND", "gloo" if stub else "nccl")   # "gloo": control-flow rehearsal

    # one MIOpen user database per rank: eight processes tuning the same shapes must not serialise on one file lock
    os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/posepaf_miopen_db_rank{local}")
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/posepaf_miopen_cache_rank{local}")
    os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
    os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)

    import numpy as np
    import torch
    from posepaf._lib import RECORD_BYTES
    from posepaf.api import records_to_numpy

    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            ndev = torch.cuda.device_count()          # does not initialise the GPU
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % max(ndev, 1)))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    eng = StubEngine(a, rank, world) if stub else GpuEngine(a, rank, world, local, backend)
    B = a.batch
    gdev = eng.gather_device
    gathered = torch.empty(world * B * RECORD_BYTES, dtype=torch.uint8, device=gdev) if world > 1 else None

    def step():
        rec = eng.step()
        if world > 1:   # the path's only exchange: fixed-size per-image records, one collective per batch
            dist.all_gather_into_tensor(gathered, rec if rec.device == gdev else rec.to(gdev))
        return rec

    def fence():
        eng.sync()
        if world > 1:
            dist.barrier()
        eng.sync()

    for _ in range(a.warmup):
        rec = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rec = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # every rank checks its own records; the verdicts are combined so that rank 0's line and every exit code agree
    recs = records_to_numpy(rec)
    status_or = int(np.bitwise_or.reduce(recs["status"].astype(np.uint32)))
    humans = int(recs["n_humans"].sum())
    if world > 1:
        allrec = gathered.cpu().numpy().view(recs.dtype)
        status_or = int(np.bitwise_or.reduce(allrec["status"].astype(np.uint32)))
        humans = int(allrec["n_humans"].sum())
    allowed = ST_SORT_UNDEFINED | (32 if a.multiscale else 0)
    bad_status = bool(status_or & ~ST_DEFINED) or bool(status_or & ~allowed)
    verdict = {"ok": True, "skipped": "--no-verify"} if a.no_verify else eng.verify(rec)
    if world > 1:   # every rank re-computes its own last batch; one failing rank fails the job
        t = torch.tensor([0 if verdict["ok"] else 1], dtype=torch.int32, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        verdict["all_ranks_ok"] = not bool(t.item())
        verdict["ok"] = verdict["all_ranks_ok"]

    rc = 0
    if rank == 0:
        multi = a.multiscale
        out = {
            "metric": "end-to-end images/sec at 512×512" + (", multi-scale" if multi else ""),
            "value": world * B * a.steps / dt, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random-init IMHN weights, random uint8 images uploaded from pinned host memory every step, "
                    "synthetic GT-style pose scenes injected into the network output)",
            "config": {"workload": ("configs[4]: original path, scales {0.5,1.0,1.5} x 512 + flip, float64 accumulation at image "
                                    "resolution in HBM, find_peaks + Python-twin matching") if multi else
                                   ("configs[2] post-processing only" if a.postproc_only else
                                    "configs[2]+forward: 512x512, flip=on, host->HBM ingest + 4-stage IMHN forward (fp16) + full HIP "
                                    "pafprocess (NMS + limb line-integral + assembly)"),
                       "images_per_gpu_per_step": B, "people_per_scene": list(SCENE_PEOPLE),
                       "ingest": "off" if a.no_ingest else "pinned host -> HBM upload of each uint8 batch inside the step (copy stream, double-buffered)",
                       "launch": "eager" if (a.no_graph or stub) else "hipGraph replay",
                       "parallelism": f"image-sharded x{world}"},
            "humans_found_in_batch": humans, "status_or": status_or, "verify": verdict,
        }
        out.update(eng.extras(a, dt, world))
        if not a.no_cpu_baseline and world == 1 and not stub:   # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.cpu_cores, check=eng.oracle_sample(rec))
            if out["cpu_baseline"].get("records_vs_oracle") == "DIFFERENT":
                verdict["ok"] = False
        print(json.dumps(out), flush=True)
    if bad_status:
        print(f"bench.py: pp_record.status = {status_or:#010x}: "
              + ("bits outside include/posepaf.h:53-59 (corrupted records)" if status_or & ~ST_DEFINED else "capacity-overflow flags raised")
              + " -- the result is INVALID", file=sys.stderr, flush=True)
        rc = 3
    if not verdict["ok"]:
        print(f"bench.py: the independent re-computation of the last batch disagrees: {verdict} -- the result is INVALID",
              file=sys.stderr, flush=True)
        rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
