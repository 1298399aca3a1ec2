#!/usr/bin/env python3
"""Root-cause probe for round 1's garbage pp_record.status (GPU only, diagnostic).

Round 1 zeroed the per-image status words with hipMemsetAsync(d_status, 0, 4 * batch) inside pp_process_batch; under
torch.cuda.graph capture that call becomes a MEMSET NODE of the HIP graph.  This probe captures exactly that -- one
hipMemsetAsync(buf, 0, nbytes) followed by a copy of buf -- for several sizes, replays it with unrelated kernels around
each replay, pre-fills buf with 0xAB before every replay, and reports what the copy saw:
    0x00000000  the memset node did its job
    0xabababab  the memset node did not run (or ran after the copy)
    other       the memset node wrote something else than the captured value
The product no longer uses a memset node (csrc/posepaf_kernels.hip or_flags); the probe documents why."""
import ctypes
import json
import sys

import torch


def main():
    assert torch.cuda.is_available()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    filler = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
    report = {}
    for nbytes in (64, 128, 256, 512, 4096):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty_like(buf)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
            out.copy_(buf)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            out.copy_(buf)
        seen = {}
        for rep in range(200):
            buf.fill_(0xAB)
            for _ in range(rep % 4):
                _ = filler @ filler
            g.replay()
            _ = filler @ filler
            words = out.cpu().numpy().view("<u4")
            for w in set(words.tolist()):
                seen[f"{w:#010x}"] = seen.get(f"{w:#010x}", 0) + 1
        report[f"{nbytes}B"] = seen
    print(json.dumps({"hipMemsetAsync_in_graph": report, "replays_per_size": 200,
                      "hip_runtime": torch.version.hip}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
