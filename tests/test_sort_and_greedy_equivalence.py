"""CPU: the lane-parallel formulation of libstdc++'s __unguarded_partition used by K_B (csrc/posepaf_kernels.hip,
StdSortGE::partition_wave) against the sequential loop it replaces (bits/stl_algo.h, as restated in
oracle/posepaf_oracle.c ls_unguarded_partition), with the reference's non-strict comparator a >= b
(pafprocess.cpp:333-335).  Pure Python on small arrays with many ties, windows inside longer arrays (so that the unguarded
scans can leave the window), both scans allowed to run off the array (flagged oob)."""
import random

import pytest


def sequential(keys, first, last, pivot):
    """bits/stl_algo.h __unguarded_partition with comp(a, b) = a >= b; scans stop at the array bounds (oob)."""
    a = list(keys)
    n, pv, oob = len(a), a[pivot], False
    while True:
        while True:
            if first >= n:
                oob = True
                break
            if not (a[first] >= pv):
                break
            first += 1
        last -= 1
        while True:
            if last < 0:
                oob = True
                break
            if not (pv >= a[last]):
                break
            last -= 1
        if not (first < last):
            return a, first, oob
        a[first], a[last] = a[last], a[first]
        first += 1


def stopper_lists(keys, first, last, pivot):
    """What partition_wave computes: left stoppers (< pivot) upwards, right stoppers (> pivot) downwards, k swaps, the cut."""
    a = list(keys)
    n, pv, oob = len(a), a[pivot], False
    L = [i for i in range(first, last) if a[i] < pv]
    R = [j for j in range(last - 1, first - 1, -1) if a[j] > pv]
    k = sum(1 for i in range(min(len(L), len(R))) if L[i] < R[i])
    for i in range(k):
        a[L[i]], a[R[i]] = a[R[i]], a[L[i]]
    if k >= 1:
        cut = L[k] if (k < len(L) and L[k] < R[k - 1]) else R[k - 1]
    elif L:
        cut = L[0]
    else:
        beyond = [i for i in range(last, n) if a[i] < pv]
        cut = beyond[0] if beyond else n
        oob |= not beyond
    if k == 0 and not R:
        oob |= not any(a[j] > pv for j in range(first - 1, -1, -1))
    return a, cut, oob


@pytest.mark.parametrize("seed", range(6))
def test_stopper_list_partition_equals_sequential_scan(seed):
    rng = random.Random(seed)
    for _ in range(1500):
        n = rng.randint(3, 90)
        levels = rng.choice([2, 3, 5, 40])              # few distinct keys -> many exact ties
        keys = [float(rng.randint(0, levels)) for _ in range(n)]
        pivot = rng.randint(0, n - 2)
        first = pivot + 1                                # std::__unguarded_partition_pivot: pivot sits just below the range
        last = rng.randint(first, n)
        want = sequential(keys, first, last, pivot)
        got = stopper_lists(keys, first, last, pivot)
        assert got == want, (keys, first, last, pivot)


def test_windows_where_a_scan_leaves_the_range():
    # every element of the window >= pivot: the left scan runs on into the neighbouring elements
    keys = [5.0, 7.0, 7.0, 9.0, 6.0, 1.0]
    assert stopper_lists(keys, 1, 4, 0) == sequential(keys, 1, 4, 0)
    # nothing smaller anywhere to the right: off the array
    keys = [5.0, 7.0, 7.0, 9.0]
    a, cut, oob = stopper_lists(keys, 1, 4, 0)
    assert (a, cut, oob) == sequential(keys, 1, 4, 0) and oob and cut == 4
    # nothing greater in the window or below it: the right scan passes the pivot and leaves the array
    keys = [5.0, 5.0, 3.0, 5.0]
    assert stopper_lists(keys, 1, 4, 0) == sequential(keys, 1, 4, 0)
    assert stopper_lists(keys, 1, 4, 0)[2]


def final_insertion_sort(keys):
    """bits/stl_algo.h __final_insertion_sort with comp(a, b) = a >= b on (key, position) pairs: guarded insertion sort of
    the first 16, __unguarded_linear_insert for the rest (runs off the front -> oob)."""
    a = [(k, i) for i, k in enumerate(keys)]
    n, oob = len(a), False

    def linear_insert(last, guarded_first):
        nonlocal oob
        val = a[last]
        nxt = last - 1
        while True:
            if nxt < 0:
                oob = True
                break
            if not (val[0] >= a[nxt][0]):
                break
            a[nxt + 1] = a[nxt]
            nxt -= 1
        a[nxt + 1] = val

    for i in range(1, min(n, 16)):              # __insertion_sort(first, first + 16)
        if a[i][0] >= a[0][0]:                  # comp(i, first): move_backward, no scan
            a[0:i + 1] = [a[i]] + a[0:i]
        else:
            linear_insert(i, True)
    for i in range(16, n):                      # __unguarded_insertion_sort
        linear_insert(i, False)
    return a, oob


@pytest.mark.parametrize("seed", range(4))
def test_final_insertion_sort_closed_form(seed):
    """K_B does not emulate the final insertion sort: the result is 'descending key, ties in REVERSE of their position in the
    array it starts from', and the unguarded insert of element p >= 16 leaves the array iff nothing strictly greater precedes it."""
    rng = random.Random(100 + seed)
    for _ in range(1500):
        n = rng.randint(1, 70)
        levels = rng.choice([1, 2, 4, 30])
        keys = [float(rng.randint(0, levels)) for _ in range(n)]
        got, oob = final_insertion_sort(keys)
        want_oob = any(p >= 16 and all(not (keys[q] > keys[p]) for q in range(p)) for p in range(n))
        assert oob == want_oob, keys
        if not oob:
            rank = [sum(1 for q in range(n) if keys[q] > keys[p]) + sum(1 for q in range(p + 1, n) if keys[q] == keys[p])
                    for p in range(n)]
            assert sorted(rank) == list(range(n))
            assert [pos for _, pos in got] == [p for _, p in sorted(zip(rank, range(n)))], keys


def greedy_sequential(cands):
    """pafprocess.cpp:113-129: candidates in sorted order; take one if neither end point is used yet."""
    used_a, used_b, out = set(), set(), []
    for rank, (a, b) in enumerate(cands):
        if a not in used_a and b not in used_b:
            used_a.add(a)
            used_b.add(b)
            out.append(rank)
    return out


def greedy_locally_dominant(cands):
    """K_B: repeat { every live candidate whose rank is the lowest among the live candidates sharing its a AND among those
    sharing its b is accepted; its end points retire; candidates touching a retired end point die }."""
    state = [0] * len(cands)           # 0 live, 1 accepted, 2 dead
    used_a, used_b = set(), set()
    for _ in range(len(cands) + 1):
        min_a, min_b, live = {}, {}, False
        for r, (a, b) in enumerate(cands):
            if state[r] == 0:
                if a in used_a or b in used_b:
                    state[r] = 2
                else:
                    min_a[a] = min(min_a.get(a, r), r)
                    min_b[b] = min(min_b.get(b, r), r)
                    live = True
        if not live:
            break
        for r, (a, b) in enumerate(cands):
            if state[r] == 0 and min_a[a] == r and min_b[b] == r:
                state[r] = 1
                used_a.add(a)
                used_b.add(b)
    return [r for r, s in enumerate(state) if s == 1]


@pytest.mark.parametrize("seed", range(4))
def test_locally_dominant_greedy_equals_sequential_greedy(seed):
    rng = random.Random(500 + seed)
    for _ in range(2000):
        na, nb = rng.randint(1, 9), rng.randint(1, 9)
        pairs = [(a, b) for a in range(na) for b in range(nb) if rng.random() < 0.6]
        rng.shuffle(pairs)             # position in the list = rank in the reference's sorted order (ranks are distinct)
        assert greedy_locally_dominant(pairs) == greedy_sequential(pairs)
