"""Deterministic, name-seeded initialisation for the IMHN (no pretrained weights are available offline).

Every tensor of the state_dict is filled from a generator seeded by CRC32(key) ^ seed, so two
implementations with the same keys get bit-identical weights.  Used for the model golden vector
(tests/golden/make_golden.py applies it to the REFERENCE module, tests apply it to ours) and for the
benchmark's random-init weights.  Scales (unit-gain fan-in) are chosen so that activations neither vanish nor explode
through 4 stages (the reference's own N(0, 0.001) init drives every output to ~0)."""
import math
import zlib

import torch


@torch.no_grad()
def deterministic_init(model: torch.nn.Module, seed: int = 0) -> None:
    sd = model.state_dict()
    for key, t in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("num_batches_tracked"):
            t.fill_(1)
        elif key.endswith("running_var"):
            t.copy_(0.6 + 0.8 * torch.rand(t.shape, generator=g))
        elif key.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif t.dim() == 4:  # conv weight
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(1.0 / fan_in))
        elif t.dim() == 2:  # linear weight
            t.copy_(torch.randn(t.shape, generator=g) * math.sqrt(1.0 / t.shape[1]))
        elif key.endswith("bn.weight") or ".bn1.weight" in key or (key.endswith(".weight") and t.dim() == 1):
            t.copy_(0.9 + 0.2 * torch.rand(t.shape, generator=g))
        else:  # biases
            t.copy_(0.05 * torch.randn(t.shape, generator=g))
