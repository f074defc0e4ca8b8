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


# ---- 3-D rotations about a centre (/root/reference/src/data/transformations.py:105-166) ------------------------------------------------
def euler_rxyz_matrix(ax, ay, az):
    """3x3 matrix of `transforms3d.euler.euler2mat(ax, ay, az, 'rxyz')` (radians; scalars or equal-shaped arrays -> (..., 3, 3)), the call
    getRotationMatrix makes (transformations.py:118-119).  transforms3d (0.3, a dependency the reference does not vendor and this image
    does not have) is restated from its published algorithm: axes 'rxyz' = (first axis 2, parity 1, no repetition, rotating frame), i.e.
    i, j, k = z, y, x, the first and last angles swapped, all three negated, then Shoemake's products.  The result equals
    Rx(ax) @ Ry(ay) @ Rz(az) -- intrinsic rotations about x, then the new y, then the new z (checked in tests/test_oracle.py)."""
    ai, aj, ak = -numpy.asarray(az, numpy.float64), -numpy.asarray(ay, numpy.float64), -numpy.asarray(ax, numpy.float64)
    si, sj, sk = numpy.sin(ai), numpy.sin(aj), numpy.sin(ak)
    ci, cj, ck = numpy.cos(ai), numpy.cos(aj), numpy.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    i, j, k = 2, 1, 0
    M = numpy.zeros(numpy.shape(ai) + (3, 3), numpy.float64)
    M[..., i, i] = cj * ck
    M[..., i, j] = sj * sc - cs
    M[..., i, k] = sj * cc + ss
    M[..., j, i] = cj * sk
    M[..., j, j] = sj * ss + cc
    M[..., j, k] = sj * cs - sc
    M[..., k, i] = -sj
    M[..., k, j] = cj * si
    M[..., k, k] = cj * ci
    return M


def getRotationMatrix(angle_x, angle_y, angle_z):
    """4x4 homogeneous rotation for angles in degrees about x, y, z (rotating frame)."""
    R = numpy.eye(4)
    R[:3, :3] = euler_rxyz_matrix(angle_x * numpy.pi / 180., angle_y * numpy.pi / 180., angle_z * numpy.pi / 180.)
    return R


def rotatePoints3D(pts, center, angle_x, angle_y, angle_z):
    """Rotate the (n, 3) points about `center`; dtype of `pts`.  The offset from the centre is formed in the points' own precision (the
    reference subtracts in place on a copy of the point), the product and the re-centring are float64, the result is stored in the
    points' dtype."""
    out = numpy.array(pts, copy=True)
    R = getRotationMatrix(angle_x, angle_y, angle_z)[:3, :3]
    ctr = numpy.asarray(center)
    rel = out - ctr.astype(out.dtype)
    out[:] = rel.astype(numpy.float64) @ R.T + ctr.astype(numpy.float64)
    return out


def rotatePoint3D(p1, center, angle_x, angle_y, angle_z):
    """One point; a float64 triple like the reference's (its `ps` is the float64 product)."""
    p = numpy.asarray(p1)
    rel = p - numpy.asarray(center).astype(p.dtype)
    return rel.astype(numpy.float64) @ getRotationMatrix(angle_x, angle_y, angle_z)[:3, :3].T + numpy.asarray(center, numpy.float64)


def transformPoint3D(pt, M):
    """One (x, y, z) point through the 4x4 homogeneous matrix M."""
    q = numpy.asarray(M, numpy.float64).reshape(4, 4) @ numpy.array([pt[0], pt[1], pt[2], 1.], numpy.float64)
    return q[:3] / q[3]


def getTransformationMatrix(center, rot, trans, scale):
    """The six coefficients (row-major 2x3) of: translate by -trans - center, rotate by `rot` (radians), scale, move back to center."""
    c, s = numpy.cos(rot), numpy.sin(rot)
    dx, dy = -trans[0] - center[0], -trans[1] - center[1]
    return numpy.array([c * scale, -s * scale, scale * (c * dx - s * dy) + center[0],
                        s * scale, c * scale, scale * (c * dy + s * dx) + center[1]])
