"""The decision by vote of k_wide's scanner waves (csrc/wide_kernel.inc: the node_num == 1 decision, `vote_lists`), as a model
in plain Python against the merge it replaced.  Every scanner wave publishes its best candidate(s) as (fp64 cost key, code)
entries, codes unique across the partition; the supervisor still takes the argmin / the k-way merge, a scanner wave only asks
"is MY entry part of the selection": nobody's entry comes before it (node_num == 1), or fewer than k entries do (lists).  The
model checks that the two give the same selection on random exchanges with heavy cost ties, short lists and empty waves —
the argument the kernel relies on (the GPU parity tests check the kernel itself).  Host-only."""
import numpy as np
import pytest

NONE = (np.iinfo(np.uint64).max, 0xFFFFFFFF)
T_BIT = 1 << 63


def exchange_single(rng, waves):
    """Per wave: an A entry (start now) or none, and a T entry (res_total) or none; codes unique."""
    codes = rng.permutation(waves * 8)
    out = []
    for w in range(waves):
        a = (int(rng.integers(0, 4)) << 40, int(codes[2 * w])) if rng.random() < 0.5 else None
        t = (int(rng.integers(0, 4)) << 40, int(codes[2 * w + 1])) if rng.random() < 0.8 else None
        out.append((a, t))
    return out


def entry_of(a, t):
    """What a wave stands for in the exchange: its A entry, else its T entry behind every A entry (bit 63), else nothing."""
    if a is not None:
        return a
    if t is not None:
        return (t[0] | T_BIT, t[1])
    return NONE


@pytest.mark.parametrize("seed", range(200))
def test_single_node_vote_is_the_argmin(seed):
    rng = np.random.default_rng(seed)
    waves = int(rng.choice([16, 32, 64]))
    ex = exchange_single(rng, waves)
    entries = [entry_of(a, t) for a, t in ex]
    best = min(entries)
    winners = [w for w, own in enumerate(entries) if own != NONE and not any(e < own for e in entries)]
    if best == NONE:
        assert winners == []          # nobody has a candidate: the job fails, nobody applies anything
    else:
        assert winners == [entries.index(best)]
        # start now iff some wave has an A entry (:6274 before :6335)
        assert (best[0] & T_BIT == 0) == any(a is not None for a, _ in ex)


def exchange_lists(rng, waves, k):
    """Per wave a sorted list of up to k (cost, code) entries; many equal costs, some waves empty."""
    codes = iter(rng.permutation(waves * k * 2))
    lists = []
    for _ in range(waves):
        n = int(rng.integers(0, k + 1)) if rng.random() < 0.7 else 0
        lst = sorted((int(rng.integers(0, 3)) << 40, int(next(codes))) for _ in range(n))
        lists.append(lst)
    return lists


@pytest.mark.parametrize("seed", range(200))
def test_list_vote_is_the_k_way_merge(seed):
    rng = np.random.default_rng(1000 + seed)
    waves = int(rng.choice([16, 32, 64]))
    k = int(rng.choice([2, 3, 4, 8]))
    lists = exchange_lists(rng, waves, k)
    everything = sorted(e for lst in lists for e in lst)
    merged = everything[:k]                       # merge_lists_w: the first k of all lists together
    nfound = min(k, len(everything))
    # vote_lists on every wave: own entries in order, stop at the first that is not selected
    selected = []
    for lst in lists:
        for own in lst:
            before = sum(1 for e in everything if e < own)
            if before >= k:
                break
            selected.append(own)
    assert nfound == len(merged)
    if nfound == k:                               # (a selection with fewer than k nodes is not applied: second exchange / failure)
        assert sorted(selected) == merged
    # the selected entries of a wave are a prefix of its list (what lets the kernel stop at the first miss)
    for lst in lists:
        flags = [sum(1 for e in everything if e < own) < k for own in lst]
        assert flags == sorted(flags, reverse=True)
