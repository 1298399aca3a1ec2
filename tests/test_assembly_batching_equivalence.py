"""CPU: the order-preserving batching the assembly kernels apply (csrc/posepaf_kernels.hip: k_assemble for the drop-in path,
assemble_image_wave -- stable slots + counting classification -- for the batched path) against the plain
connection-by-connection loop of the reference (pafprocess.cpp:138-275), both restated in Python on random inputs that are
far nastier than real scenes (few peaks per part, so skeletons share peaks, merge, and the `id > 0` membership quirk sums
real ids into ids that belong to other peaks).  What is checked is the ARGUMENT the kernel relies on:

  within one limb type a connection that is alone on the skeleton(s) it touches commutes with all the others, so maximal
  runs of such connections can be applied at once from lookup tables built at the start of the limb, provided that
  (1) connections are still visited in order, (2) the others go through the full scan one by one, (3) skeleton indices held
  by later connections are shifted after an `erase`, and (4) after a merge that ADDS two ids in the limb's own columns every
  remaining connection of the limb scans.
"""
import random

import pytest

NUM_PART = 18
LIMBS = [(1, 0), (1, 14), (1, 15), (1, 16), (1, 17), (0, 14), (0, 15), (14, 16), (15, 17), (1, 2), (2, 3), (3, 4), (1, 5), (5, 6),
         (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (0, 2), (0, 5), (2, 8), (8, 12), (5, 11), (11, 9), (16, 2),
         (17, 5), (8, 11)]   # config/config.py limbs of the 18-part skeleton


def new_row(part1, part2, c, ps):
    ids, sc = [-1] * 20, [-1.0] * 20
    ids[part1], sc[part1] = c["id1"], c["score"]
    ids[part2], sc[part2] = c["id2"], c["score"]
    ids[19], sc[19] = 2, c["len"]
    sc[18] = ps[c["id1"]] + ps[c["id2"]] + c["score"]
    return [ids, sc]


def apply_found1(row, part2, c, ps):          # pafprocess.cpp:152-180
    ids, sc = row
    min_len = int(sc[19] * 16.0)
    if ids[part2] == -1 and min_len > c["len"]:
        ids[part2], sc[part2] = c["id2"], c["score"]
        ids[19] += 1
        sc[19] = max(sc[19], c["len"])
        sc[18] += ps[c["id2"]] + c["score"]
    elif (ids[part2] != c["id2"] and sc[part2] <= c["score"] and min_len > c["len"]) or \
            (ids[part2] == c["id2"] and sc[part2] <= c["score"]):
        ids[part2], sc[part2] = c["id2"], c["score"]
        t = ps[c["id2"]] + c["score"]
        sc[18] = (sc[18] - t) + t
        sc[19] = max(sc[19], c["len"])


def scan_and_apply(skel, part1, part2, c, ps):
    """One connection exactly as the reference does it.  -> (erased index or None, summed two ids in a limb column?)"""
    found = [i for i, (ids, _) in enumerate(skel) if ids[part1] == c["id1"] or ids[part2] == c["id2"]]
    if len(found) == 1:
        apply_found1(skel[found[0]], part2, c, ps)
    elif len(found) == 2:                       # :182-256
        (i1, f1), (i2, f2) = skel[found[0]], skel[found[1]]
        min_len = int(f1[19] * 16.0)
        member, min1, min2 = False, 0.0, 0.0
        for kp in range(NUM_PART):
            a1, a2 = i1[kp] > 0, i2[kp] > 0     # id 0 counts as unassigned
            if a1:
                min1 = f1[kp] if min1 == 0.0 else min(f1[kp], min1)
            if a2:
                min2 = f2[kp] if min2 == 0.0 else min(f2[kp], min2)
            member |= a1 and a2
        if not member and (c["score"] >= min(min1, min2) * 0.7 or c["len"] < min_len):
            odd = (i1[part1] >= 0 and i2[part1] >= 0) or (i1[part2] >= 0 and i2[part2] >= 0)
            for kp in range(NUM_PART):
                i1[kp] += i2[kp] + 1
                f1[kp] += f2[kp] + 1.0
            i1[19] += i2[19]
            f1[19] = max(f1[19], c["len"])
            f1[18] += f2[18] + c["score"]
            del skel[found[1]]
            return found[1], odd
    elif len(found) == 0:                       # :257-273
        skel.append(new_row(part1, part2, c, ps))
    return None, False


def assemble_reference(conns, ps):
    skel = []
    for limb, (part1, part2) in enumerate(LIMBS):
        for c in conns[limb]:
            scan_and_apply(skel, part1, part2, c, ps)
    return skel


def assemble_batched(conns, ps, off, cnt, stats):
    skel = []
    for limb, (part1, part2) in enumerate(LIMBS):
        cs = conns[limb]
        if not cs:
            continue
        # ---- tables at the start of the limb
        own1, own2 = {}, {}
        for s, (ids, _) in enumerate(skel):
            for own, part in ((own1, part1), (own2, part2)):
                r = ids[part] - off[part]
                if 0 <= r < cnt[part]:
                    own.setdefault(r, []).append(s)
        cb1 = {c["id1"] - off[part1]: k for k, c in enumerate(cs)}
        cb2 = {c["id2"] - off[part2]: k for k, c in enumerate(cs)}
        conf = [False] * len(cs)
        for s, (ids, _) in enumerate(skel):
            r1, r2 = ids[part1] - off[part1], ids[part2] - off[part2]
            k1 = cb1.get(r1) if 0 <= r1 < cnt[part1] else None
            k2 = cb2.get(r2) if 0 <= r2 < cnt[part2] else None
            multi1 = k1 is not None and len(own1[r1]) > 1
            multi2 = k2 is not None and len(own2[r2]) > 1
            two = k1 is not None and k2 is not None and k1 != k2
            if k1 is not None and (multi1 or two):
                conf[k1] = True
            if k2 is not None and (multi2 or two):
                conf[k2] = True
        idx = []
        for k, c in enumerate(cs):
            o1 = own1.get(c["id1"] - off[part1], [None])[-1]
            o2 = own2.get(c["id2"] - off[part2], [None])[-1]
            if o1 is not None and o2 is not None and o1 != o2:
                conf[k] = True
            idx.append(o1 if o1 is not None else o2)
        # ---- in order: runs of independent connections at once, the others through the scan
        pos = 0
        while pos < len(cs):
            nxt = next((k for k in range(pos, len(cs)) if conf[k]), len(cs))
            run = list(range(pos, nxt))
            for k in run:                      # "in parallel": found-1 updates touch distinct rows, new rows keep their order
                if idx[k] is not None:
                    apply_found1(skel[idx[k]], part2, cs[k], ps)
            for k in run:
                if idx[k] is None:
                    skel.append(new_row(part1, part2, cs[k], ps))
            stats["batched"] += len(run)
            if nxt < len(cs):
                erased, odd = scan_and_apply(skel, part1, part2, cs[nxt], ps)
                stats["scanned"] += 1
                if erased is not None:
                    idx = [i - 1 if (i is not None and i > erased) else i for i in idx]
                    if odd:
                        stats["odd"] += 1
                        conf[nxt + 1:] = [True] * (len(cs) - nxt - 1)
            pos = nxt + 1
    return skel


DEAD = -2


def assemble_wave_model(conns, ps, off, cnt, stats, slots=256):
    """The wave form (assemble_image_wave / assemble_pass): STABLE slots (an erased skeleton's slot is marked dead, nothing
    moves, first / second match = lowest / second-lowest live slot), classification by COUNTING -- every live skeleton looks
    its two limb peaks up in "end point -> connection" tables and counts itself into cnt[connection] (+ a shared flag when
    it is touched by two different connections) -- found-1 updates applied by the skeleton side, births by the connection
    side, per maximal run of unflagged connections; the others through the scan; everything after an id-summing merge
    through the scan."""
    skel = []                                   # slot -> [ids, sc] or None (dead)
    for limb, (part1, part2) in enumerate(LIMBS):
        cs = conns[limb]
        if not cs:
            continue
        cb1 = {c["id1"] - off[part1]: k for k, c in enumerate(cs)}
        cb2 = {c["id2"] - off[part2]: k for k, c in enumerate(cs)}
        count, shared, myk = [0] * len(cs), [False] * len(cs), {}
        for s_, row in enumerate(skel):
            if row is None:
                continue
            ids = row[0]
            r1, r2 = ids[part1] - off[part1], ids[part2] - off[part2]
            k1 = cb1.get(r1) if 0 <= r1 < cnt[part1] else None
            k2 = cb2.get(r2) if 0 <= r2 < cnt[part2] else None
            two = k1 is not None and k2 is not None and k1 != k2
            if k1 is not None:
                count[k1] += 1
                shared[k1] |= two
            if k2 is not None and k2 != k1:
                count[k2] += 1
                shared[k2] |= two
            if not two and (k1 is not None or k2 is not None):
                myk[s_] = k1 if k1 is not None else k2
        conf = [shared[k] or count[k] >= 2 for k in range(len(cs))]
        myk = {s_: k for s_, k in myk.items() if not conf[k]}
        pos = 0
        while pos < len(cs):
            nxt = next((k for k in range(pos, len(cs)) if conf[k]), len(cs))
            for s_, k in myk.items():           # skeleton side: found-1 updates of this run
                if pos <= k < nxt:
                    apply_found1(skel[s_], part2, cs[k], ps)
            for k in range(pos, nxt):           # connection side: births in connection order
                if count[k] == 0:
                    skel.append(new_row(part1, part2, cs[k], ps))
            stats["batched"] += nxt - pos
            if nxt < len(cs):
                c = cs[nxt]
                live = [i for i, row in enumerate(skel) if row is not None]
                found = [i for i in live if skel[i][0][part1] == c["id1"] or skel[i][0][part2] == c["id2"]]
                stats["scanned"] += 1
                if len(found) == 1:
                    apply_found1(skel[found[0]], part2, c, ps)
                elif len(found) == 0:
                    skel.append(new_row(part1, part2, c, ps))
                elif len(found) == 2:
                    pair = [skel[found[0]], skel[found[1]]]
                    erased, odd = scan_and_apply(pair, part1, part2, c, ps)     # the reference's two-row logic, unchanged
                    if erased is not None:
                        skel[found[1]] = None                                    # erase = mark dead
                        if odd:
                            stats["odd"] += 1
                            conf[nxt + 1:] = [True] * (len(cs) - nxt - 1)
                            myk = {}
            pos = nxt + 1
        assert len(skel) <= slots
    return [row for row in skel if row is not None]


def random_case(rng):
    cnt = [rng.choice([0, 1, 1, 2, 3, 5]) for _ in range(NUM_PART)]
    off, run = [], 0
    for c in cnt:
        off.append(run)
        run += c
    ps = [rng.choice([0.3, 0.6, 0.9, rng.random()]) for _ in range(run)]
    conns = []
    for part1, part2 in LIMBS:
        a = list(range(off[part1], off[part1] + cnt[part1]))
        b = list(range(off[part2], off[part2] + cnt[part2]))
        rng.shuffle(a)
        rng.shuffle(b)
        m = rng.randint(0, min(len(a), len(b)))
        conns.append([{"id1": a[i], "id2": b[i], "score": rng.choice([0.2, 0.5, 0.8, rng.random()]),
                       "len": rng.choice([5.0, 20.0, 60.0, 200.0 * rng.random()])} for i in range(m)])
    return conns, ps, off, cnt


@pytest.mark.parametrize("seed", range(8))
def test_batched_assembly_equals_connection_by_connection(seed):
    rng = random.Random(seed)
    stats = {"batched": 0, "scanned": 0, "odd": 0}
    for _ in range(400):
        conns, ps, off, cnt = random_case(rng)
        want = assemble_reference(conns, ps)
        got = assemble_batched(conns, ps, off, cnt, stats)
        assert got == want
    assert stats["batched"] > 1000 and stats["scanned"] > 300   # both paths are really exercised


def test_id_summing_merges_occur_in_the_random_cases():
    rng = random.Random(1234)
    stats = {"batched": 0, "scanned": 0, "odd": 0}
    for _ in range(3000):
        conns, ps, off, cnt = random_case(rng)
        assert assemble_batched(conns, ps, off, cnt, stats) == assemble_reference(conns, ps)
    assert stats["odd"] > 0


@pytest.mark.parametrize("seed", range(8))
def test_wave_form_assembly_equals_connection_by_connection(seed):
    """the form the batched path runs (stable slots + counting classification), same random adversarial inputs"""
    rng = random.Random(100 + seed)
    stats = {"batched": 0, "scanned": 0, "odd": 0}
    for _ in range(400):
        conns, ps, off, cnt = random_case(rng)
        assert assemble_wave_model(conns, ps, off, cnt, stats) == assemble_reference(conns, ps)
    assert stats["batched"] > 1000 and stats["scanned"] > 300


def test_wave_form_handles_id_summing_merges():
    rng = random.Random(4321)
    stats = {"batched": 0, "scanned": 0, "odd": 0}
    for _ in range(3000):
        conns, ps, off, cnt = random_case(rng)
        assert assemble_wave_model(conns, ps, off, cnt, stats) == assemble_reference(conns, ps)
    assert stats["odd"] > 0
