"""CPU: bench.py's REAL control flow with the GPU engine replaced by a stub (POSEPAF_BENCH_STUB=1): `--gpus 2` from a plain
shell launches its own two ranks (torch.distributed.run, gloo, 127.0.0.1), shards, exchanges the per-image records every
step, takes the MAX over ranks, prints ONE JSON line from rank 0, and turns corrupted status words into a failing exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"POSEPAF_BENCH_STUB": "1", "POSEPAF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_single_process_stub_line():
    r, lines = _run(["--steps", "3", "--warmup", "1", "--batch", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "images/sec"
    assert out["metric"].startswith("end-to-end images/sec at 512") and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["steps_run"] == 4 and out["humans_found_in_batch"] == 4 and out["status_or"] == 0
    assert abs(out["value"] - 4 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_gpus2_self_launches_two_ranks_and_gathers():
    r, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "5"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert len(lines) == 1, lines                             # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "image-sharded x2"
    assert out["humans_found_in_batch"] == 5 * 1 + 5 * 2     # the gathered records of BOTH ranks (stub: rank r finds 1 + r people)
    assert out["steps_run"] == 5
    assert abs(out["value"] - 2 * 5 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]   # whole-job aggregate


def test_outer_torchrun_form_is_one_of_the_ranks():
    """the driver's N > 1 form: python -m torch.distributed.run ... bench.py --gpus 2 (WORLD_SIZE set: no self-launch)"""
    env = dict(os.environ, POSEPAF_BENCH_STUB="1", POSEPAF_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.parametrize("status,code", [("0xfee07f8f", 3), ("0x10", 3), ("0x8", 0)])
def test_bad_status_words_fail_the_run(status, code):
    """round 1 printed status_or = 0xfee07f8f and exited 0; undefined bits or overflow flags now fail the run,
    PP_ST_SORT_UNDEFINED (the reference's own undefined behaviour on exact ties) alone does not"""
    r, lines = _run(["--steps", "1", "--warmup", "0", "--batch", "2"], POSEPAF_BENCH_STUB_STATUS=status)
    assert r.returncode == code, r.stderr[-1000:]
    assert len(lines) == 1 and json.loads(lines[0])["status_or"] == int(status, 0)
    if code:
        assert "INVALID" in r.stderr


def test_a_failed_recomputation_fails_the_run():
    """verify() -- records vs the separate-assembly launch, forward vs the plain module, forward twice bit-identical -- saying no
    on ANY rank invalidates the line: exit code 3, the verdict in the JSON"""
    r, lines = _run(["--steps", "1", "--warmup", "0", "--batch", "2"], POSEPAF_BENCH_STUB_VERIFY="fail")
    assert r.returncode == 3 and "INVALID" in r.stderr
    assert len(lines) == 1 and json.loads(lines[0])["verify"]["ok"] is False
    r, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2"], POSEPAF_BENCH_STUB_VERIFY="fail")
    assert r.returncode != 0 and len(lines) == 1 and json.loads(lines[0])["verify"]["ok"] is False


def test_gpus_mismatch_is_refused():
    env = dict(os.environ, POSEPAF_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
