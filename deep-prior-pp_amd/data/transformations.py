"""2-D point transforms used on joint labels (API of /root/reference/src/data/transformations.py:47-102).  Host helpers
for callers (plotting, evaluation); during training the same arithmetic runs inside the augmentation kernel."""
import numpy


def transformPoint2D(pt, M):
    pt2 = numpy.dot(numpy.asarray(M).reshape((3, 3)), numpy.asarray([pt[0], pt[1], 1]))
    return numpy.asarray([pt2[0] / pt2[2], pt2[1] / pt2[2]])


def transformPoints2D(pts, M):
    ret = pts.copy()
    for i in range(pts.shape[0]):
        ret[i, 0:2] = transformPoint2D(pts[i, 0:2], M)
    return ret


def rotatePoint2D(p1, center, angle):
    """Rotate (u, v, d) about `center` by `angle` degrees."""
    alpha = angle * numpy.pi / 180.
    pp = p1.copy()
    pp[0:2] -= center[0:2]
    pr = numpy.zeros_like(pp)
    pr[0] = pp[0] * numpy.cos(alpha) - pp[1] * numpy.sin(alpha)
    pr[1] = pp[0] * numpy.sin(alpha) + pp[1] * numpy.cos(alpha)
    pr[2] = pp[2]
    pr[0:2] += center[0:2]
    return pr


def rotatePoints2D(pts, center, angle):
    ret = pts.copy()
    for i in range(pts.shape[0]):
        ret[i] = rotatePoint2D(pts[i], center, angle)
    return ret
