#!/usr/bin/env python3
"""A/B probe (GPU only, diagnostic) of round 1's garbage pp_record.status, run against ROUND 1's library rebuilt with one
switch (tools/_ab/libposepaf_r1.so, PP_AB_STATUS): 'm' = the original protocol, hipMemsetAsync(d_status) captured as a
memset node + atomicOr; 'n' = the same kernels WITHOUT the memset node (d_status zeroed once at create).  64 images per batch
(bench.py's scene mix), pp_process_batch captured into a HIP graph and replayed, exactly like `bench.py --postproc-only`.
If the garbage disappears with 'n', the memset node (not a stray kernel write) produced it.

    python tools/status_ab_probe.py            # runs both arms in child processes, prints one JSON object"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "improved-body-parts_amd")]
LIB = os.path.join(ROOT, "tools", "_ab", "libposepaf_r1.so")


def arm(graph: bool):
    import numpy as np
    import torch
    import bench
    from posepaf._lib import RECORD_BYTES, RECORD_DTYPE
    L = C.CDLL(LIB)
    vp = C.c_void_p
    L.pp_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.pp_process_batch.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    B = 64
    ctx = vp()
    assert L.pp_create(C.byref(ctx), 0, B, 128, 128, 64) == 0
    scenes, _ = bench.build_scenes(B)
    inject = torch.from_numpy(scenes).cuda()
    static_in = inject.clone()
    rec = torch.empty(B * RECORD_BYTES, dtype=torch.uint8, device="cuda")

    def body():
        rc = L.pp_process_batch(ctx, B, vp(static_in.data_ptr()), 1, 128, 128, 1, 512, None, vp(rec.data_ptr()),
                                vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc

    g = None
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(); body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
    bad_replays, values, humans = 0, {}, set()
    for rep in range(100):
        static_in.copy_(inject, non_blocking=True)
        if g is not None:
            g.replay()
        else:
            body()
        r = rec.cpu().numpy().view(RECORD_DTYPE)
        st = r["status"].astype(np.uint32)
        humans.add(int(r["n_humans"].sum()))
        if (st & ~np.uint32(0x3F)).any():
            bad_replays += 1
            for i in np.nonzero(st & ~np.uint32(0x3F))[0][:4]:
                values[f"{int(st[i]):#010x}"] = values.get(f"{int(st[i]):#010x}", 0) + 1
    return {"replays": 100, "replays_with_undefined_status_bits": bad_replays, "sample_values": dict(list(values.items())[:12]),
            "humans_per_batch": sorted(humans)}


def main():
    if len(sys.argv) > 1:
        print(json.dumps(arm(sys.argv[1] == "graph")))
        return 0
    out = {}
    for mode in ("m", "n"):
        for launch in ("graph", "eager"):
            env = dict(os.environ, PP_AB_STATUS=mode)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), launch], env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[f"{'memset_node+atomicOr' if mode == 'm' else 'no_memset_node'}/{launch}"] = \
                json.loads(line[-1]) if line else {"error": r.stderr[-600:]}
    print(json.dumps({"library": "round-1 kernels (commit c5d9fdb) + PP_AB_STATUS switch", "arms": out}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
