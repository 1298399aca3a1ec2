#!/usr/bin/env python3
"""Substitute the @PLACEHOLDER@ numbers of DESIGN.md / README.md from the bench lines kept under profiles/ (round 3).
Idempotent: the templates live in DESIGN.md.in / README.md.in next to the outputs."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])


d, m = line("r03_bench_default_b128.json"), line("r03_bench_multiscale_b32.json")
kb = d["kernel_ms"]["k_limb_connect"] * 1e3
vals = {
    "E2E": f"{d['value']:.0f}", "E2EMS": f"{d['ms_per_step']:.1f}", "MS": f"{m['value']:.0f}",
    "KB": f"{kb:.0f}", "KBFRAC": f"{d['roofline']['frac']:.2f}",
    "CHAIN": f"{d['roofline']['chain']['ms'] * 1e3:.0f}", "KA": f"{d['kernel_ms']['k_heat_peaks'] * 1e3:.0f}",
    "CHAINFRAC": f"{d['roofline']['chain']['frac']:.2f}",
    "FWD": f"{d['roofline_forward']['achieved']:.0f}", "FWDFRAC": f"{d['roofline_forward']['frac']:.3f}",
    "OWN": str(d["conv_layers"]["shapes_own_kernel"]), "CK": str(d["conv_layers"]["shapes_ck_template_kernel"]),
    "MSFWD": f"{m['roofline_forward']['frac']:.2f}", "CPU": f"{d['cpu_baseline']['value']:.0f}",
    "CPU1": f"{d['cpu_baseline']['single_core']['value']:.1f}",
}
for name in ("DESIGN.md", "README.md"):
    src = os.path.join(ROOT, name + ".in")
    if not os.path.exists(src):
        continue
    text = open(src).read()
    text = re.sub(r"@([A-Z0-9]+)@", lambda mo: vals.get(mo.group(1), mo.group(0)), text)
    left = re.findall(r"@[A-Z0-9]+@", text)
    assert not left, left
    open(os.path.join(ROOT, name), "w").write(text)
    print(name, "written")
