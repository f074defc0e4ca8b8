"""Fused augmentation kernels vs the oracle's restatement of augmentCrop / moveCoM / rotateHand / scaleHand on
explicit per-sample (mode, off, rot, sc).  Pixels: bit-exact (both sides evaluate the same IEEE-double coordinate
arithmetic without FMA contraction), rotations included: the rotation coefficients are the correctly rounded cos / sin on both
sides (oracle.augment.sincos_cr == csrc/augment.hip dpp_sincos_cr, operation for operation).  Labels: 1e-6 (f32 storage)."""
import numpy as np
import pytest

from hipdp import ops
from oracle import augment as A
from tests.backends import BACKENDS, get_runtime

MODES = {'none': 0, 'com': 1, 'rot': 2, 'sc': 3}


def _run(rt, cam, imgs, coms, cubes, Ms, gts, modes, offs, rots, scs, pca=None, norm01=False):
    B, J = gts.shape[0], gts.shape[1]
    f32 = lambda a: rt.upload(np.asarray(a, np.float32))
    d = dict(img=f32(imgs), com=f32(coms), cube=f32(cubes), M=f32(Ms.reshape(B, 9)), gt=f32(gts))
    mode = rt.upload(np.asarray(modes, np.int32))
    off, rot, sc = rt.upload(np.asarray(offs, np.float64)), rt.upload(np.asarray(rots, np.float64)), rt.upload(np.asarray(scs, np.float64))
    rec = rt.alloc(B * rt.lib.dpp_augment_record_bytes(), np.uint8)
    E = pca[1].shape[0] if pca else 0
    out_y = rt.alloc((B, E if pca else J * 3), zero=False)
    out = rt.alloc((B, 128, 128), zero=False)
    pm = f32(pca[0]) if pca else None
    pc = f32(pca[1]) if pca else None
    ops.augment_prepare(rt, d['img'], d['com'], d['cube'], d['M'], d['gt'], B, J, 128, (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y),
                        rec, out_y, mode=mode, off=off, rot=rot, sc=sc, pca_mean=pm, pca_comp=pc, E=E, norm_zero_one=norm01)(rt.stream)
    ops.augment_warp(rt, d['img'], rec, B, 128, out)(rt.stream)
    rt.synchronize()
    return out.get(), out_y.get()


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('camname', ['icvl', 'nyu'])
def test_augment_matches_oracle(backend, camname):
    rt = get_runtime(backend)
    rng = np.random.RandomState(21)
    cam = getattr(A.Camera, camname)()
    J = 16 if camname == 'icvl' else 14
    cube = (250.,) * 3 if camname == 'icvl' else (300.,) * 3
    B = 12
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=cube, joints=J)
    names = ['com', 'rot', 'sc', 'none'] * 3
    names[8], names[9], names[10] = 'com', 'rot', 'sc'
    modes = [MODES[n] for n in names]
    _, offs, rots, scs = A.draw_params(rng, B, 4)
    offs[8] = 0.0          # the reference's early-outs
    rots[9] = 0.0
    scs[10] = 1.0 + 5e-6   # inside numpy.allclose(sc, 1.)
    mean = rng.normal(0, 0.1, J * 3).astype(np.float32)
    comp = rng.normal(0, 0.2, (30, J * 3)).astype(np.float32)
    out, out_y = _run(rt, cam, imgs, coms, cubes, Ms, gts, modes, offs, rots, scs, pca=(mean, comp))
    for i in range(B):
        com2d = cam.joint3DToImg(coms[i])
        ref, lab, *_ = A.augment_crop(imgs[i].copy(), gts[i].copy(), com2d, cubes[i], Ms[i], names[i], offs[i], rots[i], scs[i],
                                      cam, abs(cam.fx), abs(cam.fy))
        nbad = int((out[i] != ref).sum())
        assert nbad == 0, (i, names[i], nbad)
        yref = A.pca_transform(lab.astype('f8'), mean.astype('f8'), comp.astype('f8'))[0]
        np.testing.assert_allclose(out_y[i], yref, rtol=0, atol=2e-6 * max(1.0, np.abs(yref).max()))
    # the warps really moved pixels (guards against a degenerate pass)
    assert (out[0] != imgs[0]).mean() > 0.01 and (out[1] != imgs[1]).mean() > 0.01 and (out[2] != imgs[2]).mean() > 0.001


@pytest.mark.parametrize('backend', BACKENDS)
def test_augment_raw_labels_and_device_rng(backend):
    rt = get_runtime(backend)
    rng = np.random.RandomState(22)
    cam = A.Camera.msra()
    B, J = 64, 21
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(200.,) * 3, joints=J)
    f32 = lambda a: rt.upload(np.asarray(a, np.float32))
    rec = rt.alloc(B * rt.lib.dpp_augment_record_bytes(), np.uint8)
    out_y, out = rt.alloc((B, J * 3), zero=False), rt.alloc((B, 128, 128), zero=False)
    out_mode = rt.alloc(B, np.int32)
    table = rt.upload(np.array([1, 2, 0], np.int32))          # aug_modes = ['com', 'rot', 'none']
    img = f32(imgs)
    ops.augment_prepare(rt, img, f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), B, J, 128,
                        (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y), rec, out_y, mode_table=table, n_modes=3, seed=1234, counter=7,
                        out_mode=out_mode)(rt.stream)
    ops.augment_warp(rt, img, rec, B, 128, out)(rt.stream)
    rt.synchronize()
    o, m = out.get(), out_mode.get()
    assert set(np.unique(m)) <= {0, 1, 2} and len(np.unique(m)) == 3
    assert np.isfinite(o).all() and o.min() >= -1.0 - 2e-6 and o.max() <= 1.0 + 2e-6
    none = np.where(m == 0)[0]
    np.testing.assert_allclose(o[none], imgs[none], atol=2e-6)
    np.testing.assert_allclose(out_y.get()[none], (gts[none] / 100.).reshape(len(none), -1), rtol=1e-6)
    # same (seed, counter) -> same draws; different counter -> different
    out2 = rt.alloc((B, 128, 128), zero=False)
    ops.augment_prepare(rt, img, f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), B, J, 128,
                        (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y), rec, out_y, mode_table=table, n_modes=3, seed=1234, counter=7)(rt.stream)
    ops.augment_warp(rt, img, rec, B, 128, out2)(rt.stream)
    rt.synchronize()
    np.testing.assert_array_equal(out2.get(), o)


@pytest.mark.parametrize('backend', BACKENDS)
def test_crop_area_3d_matches_oracle(backend):
    """The initial crop (cropArea3D + imgStackDepthOnly) bit for bit against the oracle, on frames whose
    crop window leaves the image, contains undefined pixels, and pixels nearer / farther than the cube."""
    from hipdp import runtime as R
    from util.handdetector import HandDetector, crop_frames
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    rng = np.random.RandomState(11)
    for cam, cube, (H, W) in ((A.Camera.icvl(), (250., 250., 250.), (240, 320)), (A.Camera.nyu(), (300., 300., 300.), (120, 160))):
        B = 7
        frames, coms = A.synthetic_frames(rng, B, cam, H, W, cube)
        cubes = np.tile(np.asarray(cube, np.float32), (B, 1))
        cubes[1] = (200., 220., 180.)                                   # a non-cubic, non-square window
        fx, fy = abs(cam.fx), abs(cam.fy)
        crops_n, Ms = crop_frames(frames, coms, cubes, fx, fy, 128, normalize=True, runtime=rt)
        crops_mm, _ = crop_frames(frames, coms, cubes, fx, fy, 128, normalize=False, runtime=rt)
        for i in range(B):
            d, _, _ = A.detector_preprocess(frames[i])
            ref, M, _ = A.crop_area_3d(d, coms[i], cubes[i], fx, fy)
            assert np.array_equal(crops_mm[i], ref), i
            assert np.array_equal(crops_n[i], A.normalize_crop(ref, coms[i][2], cubes[i][2])), i
            assert np.array_equal(Ms[i].astype(np.float32), M.astype(np.float32)), i
            # for a square pixel window the transform cropArea3D returns is comToTransform's (handdetector.py:228-258), the
            # one augmentation consumes; for a non-square window the reference's two functions differ by comToTransform's
            # x/y swap (handdetector.py:252-253), and so do the restatements
            xs_, xe_, ys_, ye_, _, _ = A.com_to_bounds(coms[i], cubes[i], fx, fy)
            if xe_ - xs_ == ye_ - ys_:
                np.testing.assert_array_equal(M, A.com_to_transform(coms[i], cubes[i], fx, fy))
            assert crops_n[i].min() >= -1.0 and crops_n[i].max() <= 1.0 + 1e-6
        # the reference's per-frame method
        hd = HandDetector(frames[2].copy(), fx, fy)
        crop, M, com = hd.cropArea3D(com=coms[2], size=tuple(cubes[2]), dsize=(128, 128))
        d, _, _ = A.detector_preprocess(frames[2])
        ref, Mref, _ = A.crop_area_3d(d, coms[2], cubes[2], fx, fy)
        assert np.array_equal(crop, ref) and np.allclose(M, Mref, rtol=1e-6) and hd.getNDValue() == 0.
        # com=None: the crop is centred on the centre of mass of the whole range-limited frame (handdetector.py:401-402)
        com0 = hd.calculateCoM(hd.dpt)
        crop0, M0, c0 = hd.cropArea3D(com=None, size=tuple(cubes[2]), dsize=(128, 128))
        ref0, Mref0, _ = A.crop_area_3d(d, com0.astype(np.float32), cubes[2], fx, fy)
        assert np.array_equal(crop0, ref0) and np.allclose(M0, Mref0, rtol=1e-6) and np.allclose(c0, com0, rtol=1e-6)
        with pytest.raises(NotImplementedError):
            hd.detect()


@pytest.mark.parametrize('backend', BACKENDS)
def test_crop_area_3d_docom(backend):
    """cropArea3D(docom=True): the crop is re-centred on the centre of mass of its first window (calculateCoM).  The device
    sums depth in float64 (the reference's float32 pairwise sum is not reproducible): CoM within 1e-3 mm / 1e-4 px, and the
    crop bit-exact whenever that does not move a window bound across a rounding boundary."""
    from hipdp import runtime as R
    from util.handdetector import HandDetector, crop_frames
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    rng = np.random.RandomState(12)
    cam, cube, (H, W) = A.Camera.icvl(), (250., 250., 250.), (240, 320)
    B = 9
    frames, coms = A.synthetic_frames(rng, B, cam, H, W, cube)
    coms[:, :2] += rng.uniform(-12, 12, (B, 2)).astype(np.float32)          # a sloppy initial centre
    frames[4] = 0.
    frames[4, 100, 100] = 700.                                               # an (almost) empty frame: the fall-back branch
    cubes = np.tile(np.asarray(cube, np.float32), (B, 1))
    fx, fy = abs(cam.fx), abs(cam.fy)
    crops, Ms, com2 = crop_frames(frames, coms, cubes, fx, fy, 128, normalize=False, runtime=rt, docom=True, return_com=True)
    exact = 0
    for i in range(B):
        d, mn, mx = A.detector_preprocess(frames[i])
        ref, M, cref = A.crop_area_3d_docom(d, coms[i], cubes[i], fx, fy, mn, mx)
        np.testing.assert_allclose(com2[i], cref, rtol=0, atol=1e-3)
        if A.com_to_bounds(com2[i], cubes[i], fx, fy) == A.com_to_bounds(cref, cubes[i], fx, fy):
            assert np.array_equal(crops[i], ref), i
            exact += 1
    assert exact >= B - 1
    hd = HandDetector(frames[1].copy(), fx, fy)
    crop, M, c = hd.cropArea3D(com=coms[1], size=cube, dsize=(128, 128), docom=True)
    np.testing.assert_allclose(c, com2[1], rtol=0, atol=1e-6)
    assert np.array_equal(crop, crops[1])


def test_host_com_helpers_match_oracle():
    """HandDetector.calculateCoM / getCrop / refineCoMIterative (host NumPy, handdetector.py:91-108, 260-296, 540-558)."""
    from util.handdetector import HandDetector
    cam = A.Camera.icvl()
    frames, coms = A.synthetic_frames(np.random.RandomState(21), 3, cam, 240, 320, (250., 250., 250.))
    for i in range(3):
        hd = HandDetector(frames[i].copy(), abs(cam.fx), abs(cam.fy))
        d, mn, mx = A.detector_preprocess(frames[i])
        assert hd.minDepth == mn and hd.maxDepth == mx and np.array_equal(hd.dpt, d)
        np.testing.assert_allclose(hd.calculateCoM(hd.dpt), A.calculate_com(d, mn, mx), rtol=1e-6)
        b = A.com_to_bounds(coms[i], (250., 250., 250.), abs(cam.fx), abs(cam.fy))
        assert np.array_equal(hd.getCrop(hd.dpt, *b), A.get_crop(d, *b))
        c5 = hd.refineCoMIterative(coms[i].astype(float), 5, (250., 250., 250.))
        assert np.isfinite(c5).all() and abs(c5[2] - coms[i][2]) < 125.


@pytest.mark.parametrize('backend', BACKENDS)
def test_augment_norm_zero_one(backend):
    """normZeroOne crops (values in [0, 1] from the cube's front face, nettrainer.py:948, 982-988): same kernels, other
    de-normalisation / re-normalisation constants, bit for bit against the oracle."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(31)
    cam = A.Camera.icvl()
    J, B = 16, 8
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(250.,) * 3, joints=J)
    imgs01 = ((imgs + 1.0) * 0.5).astype(np.float32)          # the same crops stored in [0, 1]
    names = ['com', 'rot', 'sc', 'none'] * 2
    modes = [MODES[n] for n in names]
    _, offs, rots, scs = A.draw_params(rng, B, 4)
    out, out_y = _run(rt, cam, imgs01, coms, cubes, Ms, gts, modes, offs, rots, scs, norm01=True)
    for i in range(B):
        ref, lab, *_ = A.augment_crop(imgs01[i].copy(), gts[i].copy(), cam.joint3DToImg(coms[i]), cubes[i], Ms[i], names[i], offs[i],
                                      rots[i], scs[i], cam, abs(cam.fx), abs(cam.fy), normZeroOne=True)
        nbad = int((out[i] != ref).sum())
        assert nbad == 0, (i, names[i], nbad)
        assert out[i].min() >= 0.0 and out[i].max() <= 1.0 + 1e-6
        np.testing.assert_allclose(out_y[i].reshape(J, 3), lab, rtol=0, atol=2e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_fused_augment_equals_two_stage_and_advances_counter(backend):
    """dpp_augment (ONE launch: prepare + warp + counter advance) against dpp_augment_prepare + dpp_augment_warp, bit for bit,
    for every split count; the device counter moves by one per launch (fresh draws per step for a recorded plan) and draws are
    keyed by the GLOBAL sample index, so two shards of a global batch reproduce the single-batch result."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(31)
    cam = A.Camera.nyu()
    B, J = 16, 14
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(300.,) * 3, joints=J)
    f32 = lambda a: rt.upload(np.asarray(a, np.float32))
    camt = (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y)
    mean = rng.normal(0, 0.1, J * 3).astype(np.float32)
    comp = rng.normal(0, 0.2, (30, J * 3)).astype(np.float32)
    pm, pc = f32(mean), f32(comp)
    table = rt.upload(np.array([1, 2, 3, 0], np.int32))
    img, com, cube, M, gt = f32(imgs), f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts)

    def two_stage(counter):
        rec = rt.alloc(B * rt.lib.dpp_augment_record_bytes(), np.uint8)
        y, x = rt.alloc((B, 30), zero=False), rt.alloc((B, 128, 128), zero=False)
        ops.augment_prepare(rt, img, com, cube, M, gt, B, J, 128, camt, rec, y, mode_table=table, n_modes=4, seed=77, counter=counter,
                            pca_mean=pm, pca_comp=pc, E=30)(rt.stream)
        ops.augment_warp(rt, img, rec, B, 128, x)(rt.stream)
        rt.synchronize()
        return x.get(), y.get()

    x0, y0 = two_stage(0)
    x1, y1 = two_stage(1)
    assert (x0 != x1).mean() > 0.01
    for splits in (0, 1, 4, 16):
        st = ops.AugmentState(rt, B, seed=77)
        xo, yo = rt.alloc((B, 128, 128), zero=False), rt.alloc((B, 30), zero=False)
        (launch,) = st.ops(img, com, cube, M, gt, J, 128, camt, xo, yo, mode_table=table, n_modes=4, pca_mean=pm, pca_comp=pc, E=30, splits=splits)
        launch(rt.stream)
        rt.synchronize()
        np.testing.assert_array_equal(xo.get(), x0)
        np.testing.assert_array_equal(yo.get(), y0)
        assert int(st.counter.get()[0]) == 1 and int(st.ticket.get()[0]) == 0
        launch(rt.stream)                                   # the same prepared launch again: the next step's draws
        rt.synchronize()
        np.testing.assert_array_equal(xo.get(), x1)
        np.testing.assert_array_equal(yo.get(), y1)
        assert int(st.counter.get()[0]) == 2
    # two ranks of 8 samples each == one batch of 16 (ragged B = 8 + an odd shard size 5 exercise the dead workgroups)
    for lo, n in ((0, 8), (8, 8), (3, 5)):
        st = ops.AugmentState(rt, n, seed=77, sample0=lo, global_batch=B)
        xo, yo = rt.alloc((n, 128, 128), zero=False), rt.alloc((n, 30), zero=False)
        v = lambda b, k: b.view(lo * k, (n * k,))          # noqa: E731
        (launch,) = st.ops(v(img, 128 * 128), v(com, 3), v(cube, 3), v(M, 9), v(gt, J * 3), J, 128, camt, xo, yo, mode_table=table,
                           n_modes=4, pca_mean=pm, pca_comp=pc, E=30)
        launch(rt.stream)
        rt.synchronize()
        np.testing.assert_array_equal(xo.get(), x0[lo:lo + n])
        np.testing.assert_array_equal(yo.get(), y0[lo:lo + n])


@pytest.mark.parametrize('backend', BACKENDS)
def test_rotation_coefficients_are_the_correctly_rounded_ones(backend):
    """rot mode over many angles (special ones included): the inverse affine map the device stores in its per-sample record
    (six float64) equals the oracle's invert_affine(getRotationMatrix2D) BIT FOR BIT -- cos / sin included, which the device
    evaluates with its own correctly rounded routine, not ocml's -- and so does every pixel of every rotated crop."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(77)
    cam = A.Camera.nyu()
    J, B = 14, 48
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(300.,) * 3, joints=J)
    rots = rng.uniform(-180, 180, B)
    rots[:12] = [90., -90., 180., -180., 45., 135., 1e-3, -1e-3, 30., 60., 179.999999, 0.5]
    offs, scs = np.zeros((B, 3)), np.ones(B)
    f32 = lambda a: rt.upload(np.asarray(a, np.float32))          # noqa: E731
    nrec = rt.lib.dpp_augment_record_bytes()
    rec = rt.alloc(B * nrec, np.uint8)
    out_y, out = rt.alloc((B, J * 3), zero=False), rt.alloc((B, 128, 128), zero=False)
    img = f32(imgs)
    ops.augment_prepare(rt, img, f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), B, J, 128, (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y),
                        rec, out_y, mode=rt.upload(np.full(B, MODES['rot'], np.int32)), off=rt.upload(offs), rot=rt.upload(rots),
                        sc=rt.upload(scs))(rt.stream)
    ops.augment_warp(rt, img, rec, B, 128, out)(rt.stream)
    rt.synchronize()
    m = rec.get().reshape(B, nrec)[:, :72].copy().view(np.float64).reshape(B, 9)      # AugRec.m leads the record
    got, lab = out.get(), out_y.get().reshape(B, J, 3)
    for i in range(B):
        r = np.mod(rots[i], 360)
        want = A.invert_affine(A.rotation_matrix_2d((64, 64), -r, 1))
        assert np.array_equal(m[i, :6], want), (i, rots[i], m[i, :6], want)
        ref, rlab, *_ = A.augment_crop(imgs[i].copy(), gts[i].copy(), cam.joint3DToImg(coms[i]), cubes[i], Ms[i], 'rot', offs[i], rots[i], 1.,
                                       cam, abs(cam.fx), abs(cam.fy))
        assert np.array_equal(got[i], ref), (i, rots[i], int((got[i] != ref).sum()))
        np.testing.assert_allclose(lab[i], rlab, rtol=0, atol=2e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_binarize_image_flag(backend):
    """augment_poses' binarizeImage (poseregnettrainer.py:255-257): the augmented crop thresholded at 0.5 -- with normZeroOne crops,
    where it separates hand from background -- as an epilogue flag of the fused kernel."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(41)
    cam = A.Camera.icvl()
    J, B = 16, 8
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(250.,) * 3, joints=J)
    imgs01 = ((imgs + 1.0) * 0.5).astype(np.float32)
    names = ['com', 'rot', 'sc', 'none'] * 2
    _, offs, rots, scs = A.draw_params(rng, B, 4)
    f32 = lambda a: rt.upload(np.asarray(a, np.float32))          # noqa: E731
    out_x, out_y = rt.alloc((B, 128, 128), zero=False), rt.alloc((B, J * 3), zero=False)
    ops.augment(rt, f32(imgs01), f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), B, J, 128, (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y),
                out_x, out_y, mode=rt.upload(np.array([MODES[n] for n in names], np.int32)), off=rt.upload(offs), rot=rt.upload(rots),
                sc=rt.upload(scs), norm_zero_one=True, binarize=True)(rt.stream)
    rt.synchronize()
    got = out_x.get()
    for i in range(B):
        ref, *_ = A.augment_crop(imgs01[i].copy(), gts[i].copy(), cam.joint3DToImg(coms[i]), cubes[i], Ms[i], names[i], offs[i], rots[i], scs[i],
                                 cam, abs(cam.fx), abs(cam.fy), normZeroOne=True)
        ref = ref.copy()
        ref[ref < 0.5] = 0
        ref[ref >= 0.5] = 1
        assert np.array_equal(got[i], ref), (i, names[i])
    assert set(np.unique(got)) == {0.0, 1.0}
