"""CPU: the data-source side of improved-body-parts_amd/evaluate.py (reference evaluate.py:235-280) -- COCO annotation reading
with the person-image filter, PNG decoding to BGR, bucket/plan arithmetic, the slot layout of posepaf.engine."""
import importlib.util
import json
import os

import numpy as np

from conftest import PKG


def _evaluate_module():
    spec = importlib.util.spec_from_file_location("pp_evaluate", os.path.join(PKG, "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    return ev


def write_coco_fixture(root, sizes=((256, 256), (200, 300), (256, 256)), people=(2, 3, 1), seed=0, with_empty=True):
    """3 PNG images + a COCO keypoint annotation file whose persons are posepaf.synth people.  -> (ann_file, img_dir)"""
    from PIL import Image
    from posepaf import oks_eval, synth
    img_dir = os.path.join(root, "images")
    os.makedirs(img_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    images, anns, aid = [], [], 1
    for k, ((h, w), p) in enumerate(zip(sizes, people)):
        iid = 1000 + 7 * k
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(rgb).save(os.path.join(img_dir, f"{iid:012d}.png"))
        images.append({"id": iid, "file_name": f"{iid:012d}.png", "height": h, "width": w})
        joints = synth.random_people(p, np.random.default_rng(seed + 50 + k), img_h=h, img_w=w)
        for g in oks_eval.gt_from_synth_joints(joints):
            anns.append(dict(g, id=aid, image_id=iid, category_id=1))
            aid += 1
    if with_empty:   # an image WITHOUT person annotations: the reference's getImgIds(catIds=person) leaves it out
        Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(os.path.join(img_dir, "empty.png"))
        images.append({"id": 5, "file_name": "empty.png", "height": 64, "width": 64})
    doc = {"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}]}
    ann_file = os.path.join(root, "person_keypoints_fixture.json")
    json.dump(doc, open(ann_file, "w"))
    return ann_file, img_dir


def test_coco_source_reads_annotations_and_images(tmp_path):
    ev = _evaluate_module()
    ann, img_dir = write_coco_fixture(str(tmp_path))
    src = ev.CocoSource(ann, img_dir)
    assert src.image_ids == [1000, 1007, 1014]                         # person images only, ascending id (evaluate.py:252-254)
    assert len(ev.CocoSource(ann, img_dir, all_images=True)) == 4      # NUM_TEST_IMG > 0 branch: every image
    assert [src.shape(i) for i in range(3)] == [(256, 256), (200, 300), (256, 256)]
    from PIL import Image
    im = src.load(1)
    assert im.dtype == np.uint8 and im.shape == (200, 300, 3)
    rgb = np.asarray(Image.open(src.files[1]))
    assert np.array_equal(im, rgb[:, :, ::-1])                         # BGR like cv2.imread
    g = src.gts[1007]
    assert 1 <= len(g) <= 3 and all(len(x["keypoints"]) == 51 and x["area"] > 0 and x["iscrowd"] == 0 for x in g)
    # --inject_gt: scenes rendered from the annotation itself, one per image of the bucket
    src = ev.CocoSource(ann, img_dir, inject_gt=True)
    bank, where = src.bank_for_bucket(256, 256, [0, 2])
    assert bank.shape == (2, 2, 50, 64, 64) and bank.dtype == np.float16 and where == {0: 0, 2: 1}
    assert bank[0, 0, 30:48].max() > 0.9 and bank[:, :, 48:].max() == 0


def test_plan_batches_and_slot_layout():
    ev = _evaluate_module()
    assert [ev.plan_batch(n, 128) for n in (1, 8, 9, 16, 17, 33, 64, 65, 127)] == [8, 8, 16, 16, 32, 64, 64, 128, 128]
    assert ev.plan_batch(1, 2) == 2 and ev.plan_batch(3, 4) == 4
    from posepaf.engine import header_bytes, padded_shape
    assert header_bytes(128) == 2048 and header_bytes(2) == 256 and header_bytes(17) == 512
    assert padded_shape(427, 640) == (448, 640)


def test_directory_source_reads_png_and_npy(tmp_path):
    from PIL import Image
    ev = _evaluate_module()
    a = np.random.default_rng(1).integers(0, 256, (70, 90, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "b.png")
    np.save(tmp_path / "a.npy", a)
    src = ev.DirSource(str(tmp_path))
    assert src.image_ids == ["a", "b"] and src.shape(0) == src.shape(1) == (70, 90)
    assert np.array_equal(src.load(0), a) and np.array_equal(src.load(1), a[:, :, ::-1])
