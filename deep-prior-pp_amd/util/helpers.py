"""Small helpers (API of /root/reference/src/util/helpers.py:87-153)."""
import numpy


def shuffle_many_inplace(arrays, random_state=None):
    """Fisher-Yates shuffle of several arrays consistently along the first axis (helpers.py:87-108)."""
    if random_state is None:
        rng = numpy.random.mtrand._rand
    elif isinstance(random_state, numpy.random.RandomState):
        rng = random_state
    else:
        raise ValueError("random_state must be None or numpy RandomState")
    assert all(i.shape[0] == arrays[0].shape[0] for i in arrays[1:])
    for oi in reversed(range(1, arrays[0].shape[0])):
        ni = rng.randint(oi + 1)
        for a in arrays:
            a[[oi, ni]] = a[[ni, oi]]


def chunks(l, n):
    """Successive n-sized chunks of l (helpers.py:145-153)."""
    for i in range(0, len(l), n):
        yield l[i:i + n]
