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
