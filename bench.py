#!/usr/bin/env python3
"""bench.py -- end-to-end images/sec at 512x512 (BASELINE.json's metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of B synthetic 512x512 images per GPU, inputs already
resident in HBM: uint8 pre-processing + flip -> 4-stage IMHN forward on (2B, 512, 512, 3) in fp16 ->
K_A heat-map NMS/refine -> K_B limb line-integral scoring + matching -> K_C person assembly -> (N > 1) RCCL
all-gather of the per-image records.  Images shard across ranks (weak scaling: B per GPU); there is no other
collective on the path.

A randomly initialised network emits no peaks, so the post-processing load comes from synthetic ground-truth
style scenes (posepaf/synth.py) ADDED to the (down-scaled) network output: the forward is fully live and the
kernels see realistic peak/limb counts.  Nothing is skipped or cached inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written kernel (k_limb_connect), timed with
HIP events on the launch stream; `cpu_baseline` is the oracle (plain-C port of the reference path) on one host core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "improved-body-parts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# one MIOpen user database per rank: eight processes tuning the same shapes must not serialise on one file lock
_rank = os.environ.get("LOCAL_RANK", "0")
os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/posepaf_miopen_db_rank{_rank}")
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/posepaf_miopen_cache_rank{_rank}")
os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)

import numpy as np  # noqa: E402
import torch  # noqa: E402

IMG = 512
FEAT = IMG // 4
SCENE_PEOPLE = (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 30, 2, 4, 6, 3)   # people per synthetic scene (mean 8.2)
HBM_PEAK_GBS = 8000.0                                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0                                           # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
FLOP_PER_IMAGE = 2 * 529.4e9                                            # 529.4 GFLOP / forward, x2 flip


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("POSEPAF_BENCH_BATCH", "64")),
                    help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="bound on the CPU baseline leg")
    ap.add_argument("--postproc-only", action="store_true", help="time only K_A..K_C (profiling aid)")
    ap.add_argument("--plain-model", action="store_true", help="unfused nn.Module forward instead of the fused one")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--multiscale", action="store_true",
                    help="BASELINE configs[4]: original path, scale search {0.5, 1.0, 1.5} x 512 + flip, float64 accumulation")
    return ap.parse_args()


def build_scenes(batch, dtype=np.float16):
    from posepaf import synth
    uniq = [synth.make_net_output(p, 9000 + i, dtype=dtype) for i, p in enumerate(SCENE_PEOPLE)]
    return np.stack([uniq[i % len(uniq)] for i in range(batch)]), uniq


def cpu_baseline(uniq_scenes, seconds):
    """Oracle (plain-C restatement of flip-average + heatmap_nms + x4 limb upsample + process_paf) on ONE core,
    same synthetic scenes, bounded wall time.  The network forward is NOT included (fp32 CPU forward of the
    reference module: ~6.5 s per image on 8 threads)."""
    from oracle.oracle import Oracle
    orc = Oracle()
    n, t0 = 0, time.perf_counter()
    while True:
        for s in uniq_scenes:
            orc.pipeline(s, IMG)
            n += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": 1, "kind": "port",
            "sample": f"{n} images ({len(uniq_scenes)} scenes of {min(SCENE_PEOPLE)}-{max(SCENE_PEOPLE)} people, cycled) "
                      f"in {dt:.1f} s; post-processing only (flip-average, NMS+refine, x4 limb upsample, process_paf); "
                      "network forward excluded"}


def bench_multiscale(a, world, rank, dev, dist, backend, model, post, images):
    """configs[4]: predict's scale search (0.5, 1.0, 1.5) with flip, per-scale maps up-sampled, resized to the image and
    accumulated in float64 in HBM, then find_peaks + the Python-twin matching at image resolution."""
    from posepaf import synth
    from posepaf._lib import RECORD_BYTES
    from posepaf.api import records_to_numpy
    from posepaf.original_path import OriginalPathProcessor, resize_images_u8
    from posepaf.pipeline import preprocess_batch
    B = a.batch
    mult = (0.5, 1.0, 1.5)
    sizes = [(64, 64, 0.5), (128, 128, 1.0), (192, 192, 1.5)]
    proc = OriginalPathProcessor(post, IMG, IMG, B)
    uniq = [synth.make_scene_at_scales(p, 9100 + i, sizes)[0] for i, p in enumerate(SCENE_PEOPLE)]
    inject = [torch.from_numpy(np.stack([uniq[i % len(uniq)][k] for i in range(B)])).to(dev) for k in range(len(sizes))]
    scale = torch.tensor(1e-3, dtype=torch.float16, device=dev)
    gdev = dev if backend == "nccl" else torch.device("cpu")
    gathered = torch.empty(world * B * RECORD_BYTES, dtype=torch.uint8, device=gdev) if world > 1 else None

    def step():
        with torch.no_grad():
            proc.reset()
            for k, sc in enumerate(mult):
                scaled = resize_images_u8(images, sc)
                x = preprocess_batch(scaled, True, torch.float16)
                maps = model(x).view(B, 2, 50, sizes[k][0], sizes[k][1])
                maps = torch.addcmul(inject[k], maps, scale)
                proc.accumulate(maps, 0, 0, len(mult))
            rec = proc.finish(B)
        if world > 1:
            dist.all_gather_into_tensor(gathered, rec if backend == "nccl" else rec.cpu())
        return rec

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    recs = records_to_numpy(rec)
    if rank == 0:
        print(json.dumps({
            "metric": "end-to-end images/sec at 512×512, multi-scale", "value": world * B * a.steps / dt, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random-init IMHN weights, random uint8 images, the same synthetic people injected at every scale)",
            "config": {"workload": "configs[4]: original path, scales {0.5,1.0,1.5} x 512 + flip, float64 accumulation at image "
                                   "resolution in HBM, find_peaks + Python-twin matching", "images_per_gpu_per_step": B,
                       "people_per_scene": list(SCENE_PEOPLE), "parallelism": f"image-sharded x{world}"},
            "humans_found_in_batch": int(recs["n_humans"].sum()), "status_or": int(np.bitwise_or.reduce(recs["status"]))}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    backend = os.environ.get("POSEPAF_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N>1 control flow on one GPU
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MIOpen exhaustive find (miopenFindConvolutionForwardAlgorithm) per conv shape: +30 % over the default heuristic
    if os.environ.get("POSEPAF_CUDNN_BENCHMARK", "1") == "1":
        torch.backends.cudnn.benchmark = True
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from posepaf._lib import RECORD_BYTES
    from posepaf.api import PosePostProcessor, records_to_numpy
    from posepaf.pipeline import PosePipeline
    from posepaf.fused_model import build_inference_model

    B = a.batch
    post = PosePostProcessor(max_batch=B, max_h=FEAT, max_w=FEAT, max_peaks_per_part=64, device=local)
    scenes_np, uniq = build_scenes(B)
    inject = torch.from_numpy(scenes_np).to(dev)                                   # (B,2,50,128,128) fp16
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    images = torch.randint(0, 256, (B, IMG, IMG, 3), dtype=torch.uint8, generator=g).to(dev)

    model = None
    if not a.postproc_only:
        model = build_inference_model(dev, fused=not a.plain_model)
    pipe = PosePipeline(model, post, dtype=torch.float16, flip=True)
    gdev = dev if backend == "nccl" else torch.device("cpu")
    gathered = torch.empty(world * B * RECORD_BYTES, dtype=torch.uint8, device=gdev) if world > 1 else None

    scale = torch.tensor(1e-3, dtype=torch.float16, device=dev)
    static_images = images.clone()

    if a.multiscale:
        return bench_multiscale(a, world, rank, dev, dist, backend, model, post, images)

    def body():
        if a.postproc_only:
            return post.process_async(inject, IMG, True)
        maps = pipe.forward_maps(static_images)
        maps = torch.addcmul(inject, maps, scale)
        return post.process_async(maps, IMG, True)

    graph, static_rec = None, None
    if not a.no_graph:
        # HIP graph of the whole per-batch path (forward + K_A/K_B/K_C): ~1000 launches replayed as one
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            static_rec = body()

    def step():
        if graph is not None:
            static_images.copy_(images, non_blocking=True)   # a fresh batch lands in the graph's input buffer
            graph.replay()
            rec = static_rec
        else:
            with torch.no_grad():
                rec = body()
        if world > 1:   # the path's only exchange: fixed-size per-image records, one collective per batch
            dist.all_gather_into_tensor(gathered, rec if backend == "nccl" else rec.cpu())
        return rec

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    recs = records_to_numpy(rec)

    # ---- roofline of the dominant hand-written kernel, HIP events on the launch stream
    ms = post.time_kernels(inject, IMG, True, iters=20)
    alg_bytes = {"k_heat_peaks": B * 18 * 2 * FEAT * FEAT * 2, "k_limb_connect": B * 30 * 2 * FEAT * FEAT * 2}
    dom = "k_limb_connect"
    achieved = alg_bytes[dom] / (ms[dom] * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(dom, {}).get(f"batch{B}")
        except Exception:
            traffic = None

    conv_summary = None
    if model is not None and not a.plain_model:
        from posepaf.fused_model import conv_choices
        ch = conv_choices()
        conv_summary = {"shapes_fused_kernel": sum(1 for v in ch.values() if v >= 0),
                        "shapes_miopen_plus_epilogue": sum(1 for v in ch.values() if v < 0)}
    if rank == 0:
        out = {
            "metric": "end-to-end images/sec at 512×512", "value": world * B * a.steps / dt, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random-init IMHN weights, random uint8 images, synthetic GT-style pose scenes injected "
                    "into the network output)",
            "config": {"workload": "configs[2]+forward: 512x512, flip=on, 4-stage IMHN forward (fp16) + full HIP "
                                   "pafprocess (NMS + limb line-integral + assembly)" if not a.postproc_only else
                                   "configs[2] post-processing only", "images_per_gpu_per_step": B,
                       "people_per_scene": list(SCENE_PEOPLE), "parallelism": f"image-sharded x{world}"},
            "forward_tflops_per_gpu": None if a.postproc_only else B * a.steps * FLOP_PER_IMAGE / dt / 1e12,
            "roofline_forward": None if a.postproc_only else {
                "bound": "mfma", "achieved": B * a.steps * FLOP_PER_IMAGE / dt / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": B * a.steps * FLOP_PER_IMAGE / dt / 1e12 / MFMA_F16_PEAK_TFLOPS,
                "note": "whole step time (forward + pre/post-processing) against the dense fp16 MFMA peak"},
            "conv_layers": conv_summary,
            "humans_found_in_batch": int(recs["n_humans"].sum()), "status_or": int(np.bitwise_or.reduce(recs["status"])),
            "kernel_ms": ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes[dom]},
        }
        if not a.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(uniq, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
