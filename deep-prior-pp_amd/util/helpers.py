"""Host helpers of the mains (API names of /root/reference/src/util/helpers.py: `shuffle_many_inplace` :87-108, `chunks` :145-153).

The mains shuffle five or six arrays of tens of thousands of 64 KB crops with one call; here the swap sequence is
composed into ONE permutation first and every array is gathered once, instead of two fancy-index row copies per array
and swap.  The permutation and the RandomState stream position are pinned by tests/golden/helpers.json (the reference's
own output)."""
import itertools

import numpy


def _swap_permutation(n, rng):
    """The permutation a descending Fisher-Yates walk applies: position oi = n-1 .. 1 is exchanged with a position
    drawn from [0, oi].  The draws are made one by one (the bound changes per draw, and the stream position after the
    call is part of what a seeded main observes), the exchanges are applied to an index vector only."""
    perm = numpy.arange(n)
    for oi in range(n - 1, 0, -1):
        ni = rng.randint(oi + 1)
        perm[oi], perm[ni] = perm[ni], perm[oi]
    return perm


def shuffle_many_inplace(arrays, random_state=None):
    """Shuffle every array of `arrays` with the same permutation of its first axis, in place."""
    if random_state is not None and not isinstance(random_state, numpy.random.RandomState):
        raise ValueError("random_state must be None or numpy RandomState")
    rng = numpy.random.mtrand._rand if random_state is None else random_state
    lengths = set(a.shape[0] for a in arrays)
    assert len(lengths) <= 1, "arrays differ in their first dimension: %s" % sorted(lengths)
    if not lengths:
        return
    perm = _swap_permutation(lengths.pop(), rng)
    for a in arrays:
        _gather_rows_inplace(a, perm)


def _gather_rows_inplace(a, perm):
    """a[i] <- a[perm[i]] with O(1) extra memory (the reference swaps rows in place; `a[...] = a[perm]` would hold a second copy of
    a 4.7 GB crop array for the duration of the call): the permutation is walked cycle by cycle with one saved row per cycle."""
    n = len(perm)
    if a.nbytes <= (64 << 20):                     # small arrays: one gather is faster than the row walk
        a[...] = a[perm]
        return
    done = numpy.zeros(n, dtype=bool)
    for start in range(n):
        if done[start] or perm[start] == start:
            done[start] = True
            continue
        saved = a[start].copy()
        i = start
        while True:
            done[i] = True
            j = int(perm[i])
            if j == start:
                a[i] = saved
                break
            a[i] = a[j]
            i = j


def chunks(l, n):
    """Iterate over consecutive slices of at most n items of the sequence l."""
    return (l[i:i + n] for i in itertools.islice(itertools.count(0, n), (len(l) + n - 1) // n))
