"""CPU: the oracle's process_paf against the reference's compiled C++ (oracle/_ref) on fresh seeds.

Wider sweep than the committed golden scenes; needs oracle/_ref/libpafprocess_ref.so (built here from
/root/reference; travels prebuilt to the GPU box)."""
import numpy as np
import pytest


@pytest.mark.parametrize("people", [1, 3, 8, 20, 40])
def test_process_paf_bit_exact_against_compiled_reference(oracle, reference_cpp, people):
    from posepaf import synth
    checked = 0
    for seed in range(100, 104):
        for dt in (np.float16, np.float32):
            net = synth.make_net_output(people, seed, dtype=dt)
            heat, paf = oracle.flip_average(net)
            jl, _ = oracle.heatmap_nms(heat)
            if len(jl) == 0:
                continue
            up = oracle.upsample4_hwc(paf)
            a = oracle.process_paf(jl[None], up, 512)
            if a["sort_oob"]:
                continue  # the reference's sort reads out of bounds here: no defined result
            b = reference_cpp.process_paf(jl[None], up, 512)
            assert np.array_equal(a["ids"], b["ids"])
            assert np.array_equal(a["scores"], b["scores"])
            assert np.array_equal(a["peaks"], b["peaks"])
            checked += 1
    assert checked >= 4


def test_min_img_size_prior_and_small_image(oracle, reference_cpp):
    """long limbs are penalised when longer than half of min_img_size (pafprocess.cpp:92)."""
    from posepaf import synth
    net = synth.make_net_output(6, 11, dtype=np.float32)
    heat, paf = oracle.flip_average(net)
    jl, _ = oracle.heatmap_nms(heat)
    up = oracle.upsample4_hwc(paf)
    for size in (512, 200, 64):
        a = oracle.process_paf(jl[None], up, size)
        b = reference_cpp.process_paf(jl[None], up, size)
        assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["scores"], b["scores"])


def test_merge_that_sums_two_peak_ids(oracle, reference_cpp):
    """pafprocess.cpp:200-228: a skeleton holding peak id 0 is not seen as sharing that part, so a merge adds the two
    ids (1 + 0 + 1 = 2) and a LATER connection of the same limb matches the made-up id.  One person, nose id 2."""
    from posepaf import synth
    net = synth.make_id_sum_merge_scene()
    heat, paf = oracle.flip_average(net, flip=False)
    jl, _ = oracle.heatmap_nms(heat)
    up = oracle.upsample4_hwc(paf)
    a = oracle.process_paf(jl[None], up, 512)
    b = reference_cpp.process_paf(jl[None], up, 512)
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["scores"], b["scores"])
    want = np.full(18, -1, np.int32)
    want[[0, 1, 2, 14]] = (2, 3, 4, 6)
    assert a["ids"].shape == (1, 18) and np.array_equal(a["ids"][0], want)
    assert len(a["connections"][21]) == 2  # nose 0 - Rsho 4 first, then the dropped nose 2 - Rsho 5
