"""CPU, world_size 2, gloo: image sharding and the record all-gather (the only exchange on the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, q):
    import sys
    from conftest import PKG, ROOT  # noqa: F401  (conftest puts the package on sys.path)
    import torch.distributed as dist
    from posepaf import dist as pdist
    from posepaf._lib import RECORD_BYTES, RECORD_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = pdist.shard_indices(n_images, rank, world)
    S = pdist.padded_shard_size(n_images, world)
    recs = np.zeros(S, RECORD_DTYPE)
    for k, i in enumerate(idx):   # a recognisable fake record per global image index
        recs[k]["n_humans"] = i % 5
        recs[k]["n_peaks"] = 1000 + i
        recs[k]["humans"][0]["score"] = i * 0.5
        recs[k]["humans"][0]["peak_id"][:] = i
    local = torch.from_numpy(recs.view(np.uint8).copy())
    assert local.numel() == S * RECORD_BYTES
    merged = pdist.gather_records(local, len(idx))
    ok = len(merged) == n_images and all(int(merged[i]["n_peaks"]) == 1000 + i for i in range(n_images)) and \
        all(float(merged[i]["humans"][0]["score"]) == i * 0.5 for i in range(n_images))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [7, 8, 1])
def test_shard_and_gather_gloo_world2(n_images):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_indices_cover_everything_once():
    from posepaf import dist as pdist
    for n in (0, 1, 5, 8, 5000):
        for w in (1, 2, 3, 8):
            allidx = np.concatenate([pdist.shard_indices(n, r, w) for r in range(w)])
            assert sorted(allidx.tolist()) == list(range(n))
            assert max(len(pdist.shard_indices(n, r, w)) for r in range(w)) <= pdist.padded_shard_size(n, w)


def test_coco_formatting_from_record():
    """evaluate.py:111-129 + :182-209 on a hand-made record."""
    from posepaf import coco, skeleton as sk
    from posepaf._lib import RECORD_DTYPE
    rec = np.zeros(1, RECORD_DTYPE)[0]
    rec["n_humans"] = 2
    rec["humans"][0]["peak_id"][:] = -1
    rec["humans"][0]["peak_id"][[0, 1, 5]] = [3, 4, 9]
    rec["humans"][0]["x"][[0, 1, 5]] = [10, 20, 30]
    rec["humans"][0]["y"][[0, 1, 5]] = [11, 21, 31]
    rec["humans"][0]["part_score"][[0, 1, 5]] = [0.9, 0.8, 0.7]
    rec["humans"][0]["score"] = 1.25
    rec["humans"][1]["peak_id"][:] = -1          # no parts: skipped (is_added False)
    humans = coco.humans_from_record(rec)
    assert len(humans) == 1 and humans[0].score == 1.25
    assert sorted(humans[0].body_parts) == [0, 1, 5]
    assert (humans[0].body_parts[5].x, humans[0].body_parts[5].y) == (30, 31)
    res = coco.coco_results(42, humans)
    assert len(res) == 1 and res[0]["image_id"] == 42 and res[0]["category_id"] == 1 and len(res[0]["keypoints"]) == 51
    kp = np.array(res[0]["keypoints"]).reshape(17, 3)
    assert tuple(kp[0]) == (10, 11, 1)                       # COCO 0 = nose = CMU 0
    assert tuple(kp[sk.ORDER_COCO.index(5)]) == (30, 31, 1)  # CMU 5 (Lsho) -> COCO 5
    assert kp[:, 2].sum() == 2                               # neck (CMU 1) is not a COCO keypoint
