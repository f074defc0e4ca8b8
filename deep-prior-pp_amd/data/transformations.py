"""2-D point transforms used on joint labels (API of /root/reference/src/data/transformations.py:47-102: the four names,
argument order and return conventions).  Host helpers for callers (plotting, evaluation, the importers' crop labels); during
training the same arithmetic runs inside the augmentation kernel (csrc/augment.hip).

Written from the behaviour the fixtures pin (tests/golden/geometry.npz holds the reference's own outputs): the array forms are
the implementation -- one homogeneous matrix product / one rotation over ALL points -- and the single-point forms are one-row
calls of them.  Arithmetic is float64; results are cast the way the reference's assignments cast them (see each function)."""
import numpy


def _homography_rows(M):
    return numpy.asarray(M, dtype=numpy.float64).reshape(3, 3)


def transformPoints2D(pts, M):
    """Projective map of the (u, v) columns of `pts` (n, >=2) by the 3x3 matrix M; further columns (depth) are kept, and the result
    has the dtype of `pts`."""
    out = numpy.array(pts, copy=True)
    H = _homography_rows(M)
    uv1 = numpy.concatenate([numpy.asarray(pts, numpy.float64)[:, :2], numpy.ones((out.shape[0], 1))], axis=1)
    w = uv1 @ H.T                                   # rows (x', y', w') of every point at once
    out[:, :2] = w[:, :2] / w[:, 2:3]
    return out


def transformPoint2D(pt, M):
    """One (u, v) point through M: a float64 pair."""
    row = numpy.array([[pt[0], pt[1]]], dtype=numpy.float64)
    return transformPoints2D(row, M)[0]


def rotatePoints2D(pts, center, angle):
    """Rotate the (u, v) columns of `pts` (n, 3) about `center` by `angle` degrees (counter-clockwise in image coordinates with v
    down: u' = u cos - v sin); the depth column is kept; dtype of `pts`."""
    out = numpy.array(pts, copy=True)
    rad = numpy.float64(angle) * numpy.pi / 180.
    c, s = numpy.cos(rad), numpy.sin(rad)
    R = numpy.array([[c, -s], [s, c]])
    ctr = numpy.asarray(center, dtype=out.dtype)[:2]
    # (the offset from the centre is formed in the points' own precision, as `pp[0:2] -= center[0:2]` on a copy of p1 does)
    rel = (out[:, :2] - ctr).astype(numpy.float64)
    out[:, :2] = (rel @ R.T).astype(out.dtype) + ctr
    return out


def rotatePoint2D(p1, center, angle):
    """One (u, v, d) point; see rotatePoints2D."""
    return rotatePoints2D(numpy.asarray(p1)[None, :], center, angle)[0]
