"""
oracle/augment.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of the online depth-crop augmentation:

  augment_crop         NetTrainer.augmentCrop        /root/reference/src/trainer/nettrainer.py:919-997
  move_com             HandDetector.moveCoM          /root/reference/src/util/handdetector.py:678-710
  rotate_hand          HandDetector.rotateHand       handdetector.py:712-747
  scale_hand           HandDetector.scaleHand        handdetector.py:750-780
  recrop_hand          HandDetector.recropHand       handdetector.py:782-803
  com_to_bounds / com_to_transform                   handdetector.py:204-258
  Camera.*             DepthImporter / NYU / MSRA projections, /root/reference/src/data/importers.py:80-119,
                       756-793, 1187-1224
  rotate_point_2d      /root/reference/src/data/transformations.py:71-88
  pca_transform        sklearn PCA.transform as used at /root/reference/src/trainer/poseregnettrainer.py:262

PARITY UNPINNED for the two OpenCV calls (cv2 2.4.x is absent here; the reference holds no
vectors).  `warp_affine_nn` / `warp_perspective_nn` restate OpenCV 2.4's imgwarp.cpp:
  * warpAffine inverts the 2x3 matrix in double, then addresses with 10-bit fixed point:
      X = (cvRound((M01*y + M02)*1024) + 512 + cvRound(M00*x*1024)) >> 10   (same for Y)
  * warpPerspective inverts the 3x3 matrix in double (cofactor formula), walks the destination in
    64x16 blocks and rounds (X0 + M0*x1) / W with cvRound (round half to even)
  * BORDER_CONSTANT value 0, INTER_NEAREST.
Bit-exactness versus real cv2 is claimed only for pixels whose source coordinate is at least
2^-9 away from a rounding boundary.

`numpy.linalg.inv(M)` (LAPACK, on the float32 M of the reference: handdetector.py:701,772) is restated with
the same f64 cofactor inverse as cv::invert, and 3x3 products are written out in plain IEEE double, so that the
HIP kernel (compiled with -ffp-contract=off) can reproduce this oracle bit for bit.

Python-2 integer division in com_to_transform (handdetector.py:246,249) is kept with `//`.
NumPy-1.x scalar semantics are kept: float32 scalar (op) python float evaluates in float64 and
is rounded to float32 only when stored into the float32 result array.
"""
import numpy as np


def cv_round(v):
    """cvRound: round half to even (lrint)."""
    return np.rint(v).astype(np.int64)


# --------------------------------------------------------------------------- camera models
class Camera(object):
    """Pinhole (un)projection of the importers.  flip_y: NYU and MSRA negate the y axis."""

    def __init__(self, fx, fy, ux, uy, flip_y):
        self.fx, self.fy, self.ux, self.uy, self.flip_y = float(fx), float(fy), float(ux), float(uy), bool(flip_y)

    @staticmethod
    def icvl():      # importers.py:199
        return Camera(241.42, 241.42, 160., 120., False)

    @staticmethod
    def msra():      # importers.py:547
        return Camera(241.42, 241.42, 160., 120., True)

    @staticmethod
    def nyu():       # importers.py:891
        return Camera(588.03, 587.07, 320., 240., True)

    def jointImgTo3D(self, s):
        s = [float(v) for v in s]
        ret = np.zeros((3,), np.float32)
        ret[0] = (s[0] - self.ux) * s[2] / self.fx
        ret[1] = ((self.uy - s[1]) if self.flip_y else (s[1] - self.uy)) * s[2] / self.fy
        ret[2] = s[2]
        return ret

    def joint3DToImg(self, s):
        f32in = isinstance(s, np.ndarray) and s.dtype == np.float32
        s = [float(v) for v in s]
        ret = np.zeros((3,), np.float32)
        if s[2] == 0.:
            ret[0] = self.ux
            ret[1] = self.uy
            return ret
        q0, q1 = s[0] / s[2], s[1] / s[2]
        if f32in:       # float32 scalar / float32 scalar stays float32 (sample[0]/sample[2], importers.py:115)
            q0, q1 = float(np.float32(q0)), float(np.float32(q1))
        ret[0] = q0 * self.fx + self.ux
        ret[1] = (self.uy - q1 * self.fy) if self.flip_y else (q1 * self.fy + self.uy)
        ret[2] = s[2]
        return ret

    def jointsImgTo3D(self, pts):
        return np.stack([self.jointImgTo3D(p) for p in pts]).astype(np.float32)

    def joints3DToImg(self, pts):
        return np.stack([self.joint3DToImg(p) for p in pts]).astype(np.float32)


# --------------------------------------------------------------------------- correctly rounded cos / sin
# The reference calls libm: cv2.getRotationMatrix2D -> C cos() / sin(), rotatePoint2D -> numpy.cos / numpy.sin (NumPy 1.x
# forwards float64 to libm).  On its platform (glibc 2.19: the IBM Accurate Mathematical Library) both are CORRECTLY ROUNDED.
# Today's numpy.cos (SIMD kernels) and the device's ocml cos are only "< 1 ulp", i.e. each may differ from that value -- and from
# each other -- in the last bit.  So the oracle computes the correctly rounded value itself, with an algorithm that is plain IEEE
# arithmetic and therefore reproducible operation for operation on the device (csrc/augment.hip: dpp_sincos_cr): Cody-Waite
# reduction by pi/2 in three 33-bit parts, Taylor series in double-double (106 bits), result = the high word.  tests/test_oracle.py
# holds it against mpmath at 400 bits (every tested argument equal; the method can miss a rounding only when the true value lies
# within 2^-100 of a midpoint between two doubles).
_SIN_DD = [(float.fromhex(a), float.fromhex(b)) for a, b in (
    ('-0x1.5555555555555p-3', '-0x1.5555555555555p-57'), ('0x1.1111111111111p-7', '0x1.1111111111111p-63'),
    ('-0x1.a01a01a01a01ap-13', '-0x1.a01a01a01a01ap-73'), ('0x1.71de3a556c734p-19', '-0x1.c154f8ddc6c00p-73'),
    ('-0x1.ae64567f544e4p-26', '0x1.c062e06d1f209p-80'), ('0x1.6124613a86d09p-33', '0x1.f28e0cc748ebep-87'),
    ('-0x1.ae7f3e733b81fp-41', '-0x1.1d8656b0ee8cbp-97'), ('0x1.952c77030ad4ap-49', '0x1.ac981465ddc6cp-103'),
    ('-0x1.2f49b46814157p-57', '-0x1.2650f61dbdcb4p-112'), ('0x1.71b8ef6dcf572p-66', '-0x1.d043ae40c4647p-120'),
    ('-0x1.761b41316381ap-75', '0x1.3423c7d91404fp-130'), ('0x1.3f3ccdd165fa9p-84', '-0x1.58ddadf344487p-139'))]
_COS_DD = [(float.fromhex(a), float.fromhex(b)) for a, b in (
    ('-0x1.0000000000000p-1', '0x0.0p+0'), ('0x1.5555555555555p-5', '0x1.5555555555555p-59'),
    ('-0x1.6c16c16c16c17p-10', '0x1.f49f49f49f49fp-65'), ('0x1.a01a01a01a01ap-16', '0x1.a01a01a01a01ap-76'),
    ('-0x1.27e4fb7789f5cp-22', '-0x1.cbbc05b4fa99ap-76'), ('0x1.1eed8eff8d898p-29', '-0x1.2aec959e14c06p-83'),
    ('-0x1.93974a8c07c9dp-37', '-0x1.05d6f8a2efd1fp-92'), ('0x1.ae7f3e733b81fp-45', '0x1.1d8656b0ee8cbp-101'),
    ('-0x1.6827863b97d97p-53', '-0x1.eec01221a8b0bp-107'), ('0x1.e542ba4020225p-62', '0x1.ea72b4afe3c2fp-120'),
    ('-0x1.0ce396db7f853p-70', '0x1.aebcdbd20331cp-124'), ('0x1.f2cf01972f578p-80', '-0x1.9ada5fcc1ab14p-135'),
    ('-0x1.88e85fc6a4e5ap-89', '0x1.71c37ebd16540p-143'))]
_PIO2 = tuple(float.fromhex(h) for h in ('0x1.921fb54400000p+0', '0x1.0b4611a600000p-34', '0x1.3198a2e000000p-69', '0x1.b839a252049c1p-104'))
_TWO_OVER_PI = float.fromhex('0x1.45f306dc9c883p-1')


def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _fast_two_sum(a, b):          # |a| >= |b|
    s = a + b
    return s, b - (s - a)


def _two_prod(a, b):
    """p + e == a * b exactly (Dekker's splitting; the device gets the same pair from one fma)."""
    p = a * b
    t = 134217729.0 * a
    ah = t - (t - a)
    al = a - ah
    t = 134217729.0 * b
    bh = t - (t - b)
    bl = b - bh
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def _dd_mul(x, y):
    p, e = _two_prod(x[0], y[0])
    e = e + (x[0] * y[1] + x[1] * y[0])
    return _fast_two_sum(p, e)


def _dd_add(x, y):
    s, e = _two_sum(x[0], y[0])
    e = e + (x[1] + y[1])
    return _fast_two_sum(s, e)


def sincos_cr(a):
    """(sin(a), cos(a)) of the float64 `a`, |a| < 8, correctly rounded (see above)."""
    a = float(a)
    k = float(np.rint(a * _TWO_OVER_PI))                  # round half to even, like rint() on the device
    # r = a - k * pi/2 as a double-double: k * (33-bit part) is exact, and so is the first difference
    r = _two_sum(a - k * _PIO2[0], -(k * _PIO2[1]))
    r = _dd_add(r, (-(k * _PIO2[2]), -(k * _PIO2[3])))
    z = _dd_mul(r, r)
    ps = _SIN_DD[-1]
    for c in _SIN_DD[-2::-1]:
        ps = _dd_add(_dd_mul(ps, z), c)
    s = _dd_add(_dd_mul(_dd_mul(ps, z), r), r)             # r + r * z * P(z)
    pc = _COS_DD[-1]
    for c in _COS_DD[-2::-1]:
        pc = _dd_add(_dd_mul(pc, z), c)
    c = _dd_add(_dd_mul(pc, z), (1.0, 0.0))                # 1 + z * Q(z)
    q = int(k) & 3
    s, c = s[0], c[0]
    return ((s, c), (c, -s), (-s, -c), (-c, s))[q]


def rotate_point_2d(p1, center, angle):
    """rotatePoint2D, transformations.py:71-88 (angle in degrees, keeps the dtype of p1)."""
    alpha = angle * np.pi / 180.
    pp = p1.copy()
    pp[0:2] -= center[0:2]
    pr = np.zeros_like(pp)
    sa, ca = sincos_cr(alpha)
    pr[0] = pp[0] * ca - pp[1] * sa
    pr[1] = pp[0] * sa + pp[1] * ca
    pr[2] = pp[2]
    pr[0:2] += center[0:2]
    return pr


# --------------------------------------------------------------------------- crop geometry
def com_to_bounds(com, size, fx, fy):
    """comToBounds, handdetector.py:204-226 (the ill-defined-CoM branch is not on the augment path:
    moveCoM/scaleHand guard against z == 0 before calling)."""
    c0, c1, c2 = float(com[0]), float(com[1]), float(com[2])
    zstart = c2 - size[2] / 2.
    zend = c2 + size[2] / 2.
    xstart = int(np.floor((c0 * c2 / fx - size[0] / 2.) / c2 * fx + 0.5))
    xend = int(np.floor((c0 * c2 / fx + size[0] / 2.) / c2 * fx + 0.5))
    ystart = int(np.floor((c1 * c2 / fy - size[1] / 2.) / c2 * fy + 0.5))
    yend = int(np.floor((c1 * c2 / fy + size[1] / 2.) / c2 * fy + 0.5))
    return xstart, xend, ystart, yend, zstart, zend


def com_to_transform(com, size, fx, fy, dsize=(128, 128)):
    """comToTransform, handdetector.py:228-258: off . scale . trans (3x3 float64)."""
    xstart, xend, ystart, yend, _, _ = com_to_bounds(com, size, fx, fy)
    trans = np.eye(3)
    trans[0, 2] = -xstart
    trans[1, 2] = -ystart
    wb = (xend - xstart)
    hb = (yend - ystart)
    if wb > hb:
        scale = np.eye(3) * dsize[0] / float(wb)
        sz = (dsize[0], hb * dsize[0] // wb)          # py2 int division, handdetector.py:246
    else:
        scale = np.eye(3) * dsize[1] / float(hb)
        sz = (wb * dsize[1] // hb, dsize[1])          # handdetector.py:249
    scale[2, 2] = 1
    xs = int(np.floor(dsize[0] / 2. - sz[1] / 2.))    # the x/y swap of handdetector.py:252-253 is kept
    ys = int(np.floor(dsize[1] / 2. - sz[0] / 2.))
    # off . (scale . trans) written out (plain IEEE double products/sums, no BLAS fused multiply-adds)
    sc = float(scale[0, 0])
    return np.array([[sc, 0., sc * float(-xstart) + float(xs)],
                     [0., sc, sc * float(-ystart) + float(ys)],
                     [0., 0., 1.]], dtype=np.float64)


# --------------------------------------------------------------------------- cv2 restatements
def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D (center is a Point2f)."""
    a = angle_deg * np.pi / 180.
    sa, ca = sincos_cr(a)
    alpha = ca * scale
    beta = sa * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (imgwarp.cpp)."""
    M = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1. / D if D != 0 else 0.
    A11 = M[1, 1] * D
    A22 = M[0, 0] * D
    m = np.empty(6)
    m[0] = A11
    m[1] = M[0, 1] * (-D)
    m[3] = M[1, 0] * (-D)
    m[4] = A22
    m[2] = -m[0] * M[0, 2] - m[1] * M[1, 2]
    m[5] = -m[3] * M[0, 2] - m[4] * M[1, 2]
    return m


def warp_affine_coords(Minv6, w, h):
    """Source pixel (X, Y) for every destination pixel, warpAffine INTER_NEAREST fixed point."""
    m = Minv6
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = cv_round(m[0] * xs * 1024.)
    bdelta = cv_round(m[3] * xs * 1024.)
    X0 = cv_round((m[1] * ys + m[2]) * 1024.) + 512
    Y0 = cv_round((m[4] * ys + m[5]) * 1024.) + 512
    X = (X0[:, None] + adelta[None, :]) >> 10
    Y = (Y0[:, None] + bdelta[None, :]) >> 10
    return X, Y


def _gather(src, X, Y, border=0.):
    h, w = src.shape
    ok = (X >= 0) & (X < w) & (Y >= 0) & (Y < h)
    out = np.full(X.shape, border, dtype=src.dtype)
    out[ok] = src[Y[ok], X[ok]]
    return out


def warp_affine_nn(src, M23, border=0.):
    h, w = src.shape
    X, Y = warp_affine_coords(invert_affine(M23), w, h)
    return _gather(src, X, Y, border)


def invert_3x3(M):
    """cv::invert for a 3x3 double matrix: determinant + cofactors (matrix.cpp, n == 3 branch)."""
    S = np.array(M, dtype=np.float64).reshape(3, 3)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) -
         S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0]) +
         S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if d == 0:
        return np.zeros((3, 3))
    d = 1. / d
    t = np.empty(9)
    t[0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[3] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[4] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[5] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[6] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[7] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[8] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t.reshape(3, 3)


def mat3_mul(A, B):
    """3x3 product with python-float arithmetic in k = 0,1,2 order (one rounding per operation)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    C = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            acc = 0.0
            for k in range(3):
                acc = acc + float(A[i, k]) * float(B[k, j])
            C[i, j] = acc
    return C


WP_BLOCK_W = 64     # bw0 for a 128-wide image with BLOCK_SZ = 32 (imgwarp.cpp warpPerspective)


def warp_perspective_coords(Minv9, w, h):
    m = np.asarray(Minv9, dtype=np.float64).reshape(9)
    xs = np.arange(w)
    bx = (xs // WP_BLOCK_W) * WP_BLOCK_W
    x1 = (xs - bx).astype(np.float64)
    bx = bx.astype(np.float64)
    ys = np.arange(h, dtype=np.float64)[:, None]
    X0 = m[0] * bx[None, :] + m[1] * ys + m[2]
    Y0 = m[3] * bx[None, :] + m[4] * ys + m[5]
    W0 = m[6] * bx[None, :] + m[7] * ys + m[8]
    Wv = W0 + m[6] * x1[None, :]
    with np.errstate(divide='ignore'):
        Wv = np.where(Wv != 0, 1. / Wv, 0.)
    lim_lo, lim_hi = float(-2 ** 31), float(2 ** 31 - 1)
    fX = np.maximum(lim_lo, np.minimum(lim_hi, (X0 + m[0] * x1[None, :]) * Wv))
    fY = np.maximum(lim_lo, np.minimum(lim_hi, (Y0 + m[3] * x1[None, :]) * Wv))
    X = np.clip(cv_round(fX), -32768, 32767)       # saturate_cast<short>
    Y = np.clip(cv_round(fY), -32768, 32767)
    return X, Y


def warp_perspective_nn(src, M33, border=0.):
    h, w = src.shape
    X, Y = warp_perspective_coords(invert_3x3(M33), w, h)
    return _gather(src, X, Y, border)


# --------------------------------------------------------------------------- HandDetector warps
def recrop_hand(crop, M, Mnew, com, size, fx, fy, background_value=0., nv_val=32000.):
    """recropHand, handdetector.py:782-803 (argument names as in the reference: the warp matrix is
    dot(M, Mnew))."""
    warped = warp_perspective_nn(crop, mat3_mul(M, Mnew), border=float(background_value))
    warped[np.isclose(warped, nv_val)] = background_value
    _, _, _, _, zstart, zend = com_to_bounds(com, size, fx, fy)
    msk1 = np.logical_and(warped < zstart, warped != 0)
    msk2 = np.logical_and(warped > zend, warped != 0)
    warped[msk1] = zstart
    warped[msk2] = 0.
    return warped


def move_com(dpt, cube, com, off, joints3D, M, cam, fx, fy):
    """moveCoM, handdetector.py:678-710."""
    if np.allclose(off, 0.):
        return dpt, joints3D, com, M
    new_com = cam.joint3DToImg(cam.jointImgTo3D(com) + off)
    if not (np.allclose(com[2], 0.) or np.allclose(new_com[2], 0.)):
        Mnew = com_to_transform(new_com, cube, fx, fy, dpt.shape)
        new_dpt = recrop_hand(dpt, Mnew, invert_3x3(np.asarray(M, dtype=np.float64)), new_com, cube, fx, fy)
    else:
        Mnew = M
        new_dpt = dpt
    new_joints3D = joints3D + cam.jointImgTo3D(com) - cam.jointImgTo3D(new_com)
    return new_dpt, new_joints3D, new_com, Mnew


def rotate_hand(dpt, cube, com, rot, joints3D, cam):
    """rotateHand, handdetector.py:712-747."""
    if np.allclose(rot, 0.):
        return dpt, joints3D, rot
    rot = np.mod(rot, 360)
    M = rotation_matrix_2d((dpt.shape[1] // 2, dpt.shape[0] // 2), -rot, 1)
    new_dpt = warp_affine_nn(dpt, M, border=0.)
    com3D = cam.jointImgTo3D(com)
    joint_2D = cam.joints3DToImg(joints3D + com3D)
    data_2D = np.zeros_like(joint_2D)
    for k in range(data_2D.shape[0]):
        data_2D[k] = rotate_point_2d(joint_2D[k], com[0:2], rot)
    new_joints3D = (cam.jointsImgTo3D(data_2D) - com3D)
    return new_dpt, new_joints3D, rot


def scale_hand(dpt, cube, com, sc, joints3D, M, fx, fy):
    """scaleHand, handdetector.py:750-780 (z-threshold uses the OLD cube)."""
    if np.allclose(sc, 1.):
        return dpt, joints3D, cube, M
    new_cube = [s * sc for s in cube]
    if not np.allclose(com[2], 0.):
        Mnew = com_to_transform(com, new_cube, fx, fy, dpt.shape)
        new_dpt = recrop_hand(dpt, Mnew, invert_3x3(np.asarray(M, dtype=np.float64)), com, cube, fx, fy)
    else:
        Mnew = M
        new_dpt = dpt
    return new_dpt, joints3D, new_cube, Mnew


def augment_crop(img, gt3Dcrop, com, cube, M, mode, off, rot, sc, cam, fx, fy, normZeroOne=False):
    """
    augmentCrop, nettrainer.py:919-997, with the random draws (mode, off, rot, sc) passed in
    explicitly (the reference's worker RNG is re-seeded from OS entropy, nettrainer.py:611, so
    parity is defined per sample on explicit parameters).  mode in {'com','rot','sc','none'}.
    Returns (imgD, curLabel, cube, com, M, rot) like the reference's return tuple (minus the None).
    """
    assert len(img.shape) == 2
    cube = [float(c) for c in cube]
    if normZeroOne:
        img = img * cube[2] + (com[2] - (cube[2] / 2.))
    else:
        img = img * (cube[2] / 2.) + com[2]
    premax = img.max()
    if mode == 'com':
        rot, sc = 0., 1.
        imgD, new_joints3D, com, M = move_com(img.astype('float32'), cube, com, off, gt3Dcrop, M, cam, fx, fy)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif mode == 'rot':
        off, sc = np.zeros((3,)), 1.
        imgD, new_joints3D, rot = rotate_hand(img.astype('float32'), cube, com, rot, gt3Dcrop, cam)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif mode == 'sc':
        off, rot = np.zeros((3,)), 0.
        imgD, new_joints3D, cube, M = scale_hand(img.astype('float32'), cube, com, sc, gt3Dcrop, M, fx, fy)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif mode == 'none':
        off, sc, rot = np.zeros((3,)), 1., 0.
        imgD = img
        curLabel = gt3Dcrop / (cube[2] / 2.)
    else:
        raise NotImplementedError()
    imgD = np.array(imgD, dtype=np.float32, copy=True)
    # scalar (float64) thresholds are cast to the float32 array dtype by NumPy for comparison and assignment
    far = np.float32(float(com[2]) + (cube[2] / 2.))
    near = np.float32(float(com[2]) - (cube[2] / 2.))
    imgD[imgD == premax] = far
    imgD[imgD == 0] = far
    imgD[imgD >= far] = far
    imgD[imgD <= near] = near
    if normZeroOne:
        imgD -= near
        imgD /= np.float32(cube[2])
    else:
        imgD -= np.float32(com[2])
        imgD /= np.float32(cube[2] / 2.)
    return imgD, curLabel, np.asarray(cube), com, M, rot


def pca_transform(label, mean, components):
    """sklearn PCA.transform (no whitening): (x - mean_) . components_^T."""
    return (label.reshape(1, -1) - mean) @ components.T


# --------------------------------------------------------------------------- synthetic samples
def synthetic_augment_inputs(rng, n, cam, cube=(250., 250., 250.), joints=16, dsize=128):
    """SURVEY.md section 8(d) cfg 3: per sample com=(u,v,d), M = comToTransform(com, cube), joints ~
    N(0, 35^2 mm) clipped to +-cube/2, and a normalised crop with a blob."""
    from .nets import synthetic_crops
    fx, fy = abs(cam.fx), abs(cam.fy)
    imgs = synthetic_crops(rng, n, dsize, dsize, np.float32)[:, 0]
    coms = np.zeros((n, 3), np.float32)
    cubes = np.tile(np.asarray(cube, np.float32), (n, 1))
    Ms = np.zeros((n, 3, 3), np.float32)
    gts = np.zeros((n, joints, 3), np.float32)
    for i in range(n):
        com2d = np.array([rng.uniform(60, 260), rng.uniform(40, 200), rng.uniform(300, 600)], np.float32)
        coms[i] = cam.jointImgTo3D(com2d)
        Ms[i] = com_to_transform(cam.joint3DToImg(coms[i]), cube, fx, fy, (dsize, dsize))
        gts[i] = np.clip(rng.normal(0, 35., (joints, 3)), -cube[0] / 2., cube[0] / 2.)
    return imgs, coms, cubes, Ms, gts


def draw_params(rng, n, n_modes, sigma_com=5., sigma_sc=0.02, rot_range=180.):
    """The four draws of augmentCrop in the reference's order, nettrainer.py:954-957."""
    modes = np.zeros(n, np.int32)
    offs = np.zeros((n, 3))
    rots = np.zeros(n)
    scs = np.zeros(n)
    for i in range(n):
        modes[i] = rng.randint(0, n_modes)
        offs[i] = rng.randn(3) * sigma_com
        rots[i] = rng.uniform(-rot_range, rot_range)
        scs[i] = abs(1. + rng.randn() * sigma_sc)
    return modes, offs, rots, scs


# --------------------------------------------------------------------------- initial crop
def detector_preprocess(dpt):
    """HandDetector.__init__, handdetector.py:53-68: depth outside [max(10, min), min(1500, max)] is 'not defined' (0)."""
    d = np.asarray(dpt, np.float32).copy()
    max_depth = min(1500, d.max())
    min_depth = max(10, d.min())
    d[d > max_depth] = 0.
    d[d < min_depth] = 0.
    return d, min_depth, max_depth


def get_crop(dpt, xstart, xend, ystart, yend, zstart, zend, background=0.):
    """getCrop, handdetector.py:260-296: the window of the frame, zero-padded where it leaves the frame, then the
    z-threshold (nearer than the cube -> front face, farther -> 0)."""
    H, W = dpt.shape
    cropped = dpt[max(ystart, 0):min(yend, H), max(xstart, 0):min(xend, W)].copy()
    cropped = np.pad(cropped, ((abs(ystart) - max(ystart, 0), abs(yend) - min(yend, H)),
                               (abs(xstart) - max(xstart, 0), abs(xend) - min(xend, W))), mode='constant', constant_values=background)
    msk1 = np.logical_and(cropped < zstart, cropped != 0)
    msk2 = np.logical_and(cropped > zend, cropped != 0)
    cropped[msk1] = zstart
    cropped[msk2] = 0.
    return cropped


def resize_nn(src, dsize_wh):
    """cv2.resize(src, (w, h), interpolation=INTER_NEAREST) as in OpenCV 2.4 imgwarp.cpp resizeNN:
    inv_scale = (double)dsize / ssize, ifx = 1. / inv_scale, sx = min(cvFloor(x * ifx), ssize - 1)."""
    w, h = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = src.shape
    ifx = 1. / (float(w) / float(sw))
    ify = 1. / (float(h) / float(sh))
    sx = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    return src[sy][:, sx]


def crop_area_3d(dpt, com, size, fx, fy, dsize=(128, 128), nd_value=0.):
    """cropArea3D with docom=False, handdetector.py:382-490: metric cube around `com` (image coordinates, z in mm) ->
    dsize crop (mm, background = nd_value) + the crop transform M.  `dpt` is the detector-preprocessed frame."""
    xstart, xend, ystart, yend, zstart, zend = com_to_bounds(com, size, fx, fy)
    cropped = get_crop(dpt, xstart, xend, ystart, yend, zstart, zend)
    wb, hb = (xend - xstart), (yend - ystart)
    if wb > hb:
        sz = (dsize[0], hb * dsize[0] // wb)          # py2 integer division, handdetector.py:447-450
    else:
        sz = (wb * dsize[1] // hb, dsize[1])
    if cropped.shape[0] > cropped.shape[1]:
        sc = sz[1] / float(cropped.shape[0])
    else:
        sc = sz[0] / float(cropped.shape[1])
    rz = resize_nn(cropped, sz)
    ret = np.ones(dsize, np.float32) * np.float32(nd_value)
    xs = int(np.floor(dsize[0] / 2. - rz.shape[1] / 2.))
    ys = int(np.floor(dsize[1] / 2. - rz.shape[0] / 2.))
    ret[ys:ys + rz.shape[0], xs:xs + rz.shape[1]] = rz
    M = np.array([[sc, 0., sc * float(-xstart) + float(xs)], [0., sc, sc * float(-ystart) + float(ys)], [0., 0., 1.]], dtype=np.float64)
    return ret, M, com


def calculate_com(dpt, min_depth, max_depth):
    """calculateCoM, handdetector.py:91-108: (mean column, mean row, mean depth) of the pixels inside the depth range."""
    dc = np.asarray(dpt, np.float32).copy()
    dc[dc < min_depth] = 0
    dc[dc > max_depth] = 0
    ys, xs = np.nonzero(dc > 0)
    num = np.count_nonzero(dc)
    if num == 0:
        return np.array((0., 0., 0.))
    return np.array((xs.mean() * num, ys.mean() * num, dc.sum(dtype=np.float64)), np.float64) / num


def crop_area_3d_docom(dpt, com, size, fx, fy, min_depth, max_depth, dsize=(128, 128), nd_value=0.):
    """cropArea3D with docom=True and no refinement net, handdetector.py:413-427: re-centre on the CoM of the first window."""
    xstart, xend, ystart, yend, zstart, zend = com_to_bounds(com, size, fx, fy)
    cropped = get_crop(dpt, xstart, xend, ystart, yend, zstart, zend)
    com2 = calculate_com(cropped, min_depth, max_depth)
    if np.allclose(com2, 0.):
        com2[2] = cropped[cropped.shape[0] // 2, cropped.shape[1] // 2]
        if np.isclose(com2[2], 0):
            com2[2] = 300
    com2[0] += xstart
    com2[1] += ystart
    com2 = com2.astype(np.float32)           # the device keeps image coordinates in float32 (as the importers' arrays do)
    ret, M, _ = crop_area_3d(dpt, com2, size, fx, fy, dsize, nd_value)
    return ret, M, com2


def normalize_crop(crop_mm, com_z, cube_z):
    """Dataset.imgStackDepthOnly, dataset.py:97-103: undefined depth (0) -> far plane, then (d - com_z) / (cube_z / 2)."""
    d = np.asarray(crop_mm, np.float32).copy()
    d[d == 0] = com_z + (cube_z / 2.)
    d -= com_z
    d /= (cube_z / 2.)
    return d


def synthetic_frames(rng, n, cam, H=240, W=320, cube=(250., 250., 250.)):
    """Full depth frames for the crop tests: a far wall with holes (0 = not defined), a hand-sized blob around a CoM that
    may sit close to the image border, some pixels nearer / farther than the cube."""
    frames = np.zeros((n, H, W), np.float32)
    coms = np.zeros((n, 3), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        d = rng.uniform(350., 900.)
        u = rng.uniform(-10., W + 10.) if i % 3 == 0 else rng.uniform(60., W - 60.)
        v = rng.uniform(-10., H + 10.) if i % 3 == 0 else rng.uniform(50., H - 50.)
        f = np.full((H, W), 1400., np.float32) + rng.normal(0, 3., (H, W)).astype(np.float32)
        f[rng.uniform(size=(H, W)) < 0.05] = 0.
        r = cube[0] / 2. * cam.fx / d * rng.uniform(0.5, 0.9)
        blob = (xx - u) ** 2 + (yy - v) ** 2 < r * r
        f[blob] = (d + rng.normal(0, 30., (H, W)))[blob].astype(np.float32)
        stick = np.abs(xx - u - r / 2) < 3
        f[stick & (rng.uniform(size=(H, W)) < 0.5)] = np.float32(d - cube[2])       # nearer than the cube's front face
        f[rng.uniform(size=(H, W)) < 0.01] = 2500.                                      # beyond maxDepth
        frames[i] = f
        coms[i] = (u, v, d)
    return frames, coms


def refine_com(crop_mm, size, com, net_forward):
    """refineCoM, handdetector.py:634-676: normalise and clamp the crop to the cube, feed it with its 1/2 and 1/4 centre crops to the
    refinement net (`net_forward(list of (1,1,h,w) arrays) -> (1, 3)`), return the offset in mm (float32)."""
    imgD = np.asarray(crop_mm, np.float32).copy()
    imgD[imgD == 0] = com[2] + (size[2] / 2.)
    imgD[imgD >= com[2] + (size[2] / 2.)] = com[2] + (size[2] / 2.)
    imgD[imgD <= com[2] - (size[2] / 2.)] = com[2] - (size[2] / 2.)
    imgD -= com[2]
    imgD /= (size[2] / 2.)
    test_data = np.zeros((1, 1) + imgD.shape, np.float32)
    test_data[0, 0] = imgD
    ins = [test_data]
    for k in (2, 4):
        dsize = (int(test_data.shape[2] // k), int(test_data.shape[3] // k))
        xstart = int(test_data.shape[2] / 2 - dsize[0] / 2)
        ystart = int(test_data.shape[3] / 2 - dsize[1] / 2)
        ins.append(test_data[:, :, ystart:ystart + dsize[1], xstart:xstart + dsize[0]])
    jts = np.asarray(net_forward(ins), np.float32)
    return jts[0] * np.float32(size[2] / 2.)


def crop_area_3d_refined(dpt, com, size, cam, fx, fy, min_depth, max_depth, net_forward, dsize=(128, 128), nd_value=0., rsize=(128, 128),
                         com2_override=None):
    """cropArea3D with docom=True AND a refinement net, handdetector.py:413-440: re-centre on the CoM of the first window, show the
    net that window resized to `rsize` as it is (resizeCrop(cropped, dsize), :430 -- no aspect-preserving paste), move the centre by
    the regressed offset, crop again (the aspect-preserving crop of :446-490).  Returns (crop, M, com2, com1, net input crop).
    com2_override: crop around this centre instead of the one computed here (tests hand in the device's own float32 centre so
    that the final crops can be compared bit for bit although the net outputs differ in the last bits)."""
    xstart, xend, ystart, yend, zstart, zend = com_to_bounds(com, size, fx, fy)
    cropped = get_crop(dpt, xstart, xend, ystart, yend, zstart, zend)
    com1 = calculate_com(cropped, min_depth, max_depth)
    if np.allclose(com1, 0.):
        com1[2] = cropped[cropped.shape[0] // 2, cropped.shape[1] // 2]
        if np.isclose(com1[2], 0):
            com1[2] = 300
    com1[0] += xstart
    com1[1] += ystart
    com1 = com1.astype(np.float32)
    xstart, xend, ystart, yend, zstart, zend = com_to_bounds(com1, size, fx, fy)
    cropped = get_crop(dpt, xstart, xend, ystart, yend, zstart, zend)
    rz = resize_nn(cropped, rsize)
    newCom3D = refine_com(rz, size, com1, net_forward) + cam.jointImgTo3D(com1)
    com2 = cam.joint3DToImg(newCom3D)
    if np.allclose(com2, 0.):
        com2[2] = cropped[cropped.shape[0] // 2, cropped.shape[1] // 2]
    use = com2 if com2_override is None else np.asarray(com2_override, np.float32)
    ret, M, _ = crop_area_3d(dpt, use, size, fx, fy, dsize, nd_value)
    return ret, M, com2, com1, rz
