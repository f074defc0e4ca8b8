"""
HandDetector -- the augmentation slice of /root/reference/src/util/handdetector.py (comToBounds / comToTransform
:204-258, moveCoM / rotateHand / scaleHand / recropHand :678-803, sampleRandomPoses :805-909).

The crop warps run on the MI355X through the fused augmentation kernels (csrc/augment.hip); this class keeps the
reference's per-crop method signatures for callers and computes only the tiny 3x3 crop geometry on the host.
cropArea3D (handdetector.py:382-490, docom=False: the call the importers make for every frame) runs on the device too;
`crop_frames` is its batched form fused with Dataset.imgStackDepthOnly.  CoM refinement by a ScaleNet (refineCoM,
handdetector.py:634-676) goes through the net's computeOutput.  Whole-frame detection / tracking (detect, track) are not
provided yet.
"""
import numpy

from data.transformations import rotatePoints2D      # noqa: F401  (part of the module surface the reference exposes)


class HandDetector(object):
    RESIZE_BILINEAR = 0
    RESIZE_CV2_NN = 1
    RESIZE_CV2_LINEAR = 2

    def __init__(self, dpt, fx, fy, importer=None, refineNet=None):
        self.dpt = dpt
        self.maxDepth = min(1500, dpt.max())
        self.minDepth = max(10, dpt.min())
        self.dpt[self.dpt > self.maxDepth] = 0.
        self.dpt[self.dpt < self.minDepth] = 0.
        self.fx, self.fy = fx, fy
        self.refineNet = refineNet
        self.importer = importer
        self.resizeMethod = self.RESIZE_CV2_NN

    @staticmethod
    def detectionModeToString(com, refineNet):
        """Tag of the cache files (handdetector.py:72-89)."""
        if com is False and refineNet is False:
            return 'gt'
        if com is True and refineNet is False:
            return 'com'
        if com is True and refineNet is True:
            return 'comref'
        raise NotImplementedError("com {}, refineNet {}".format(com, refineNet))

    def checkImage(self, tol):
        """Is there some content in the image (handdetector.py:110-120)."""
        return not (numpy.std(self.dpt) < tol)

    # ---- crop geometry (host, a handful of flops) -----------------------------------------------------------
    def comToBounds(self, com, size):
        """Project the metric cube around the CoM back to pixel bounds (handdetector.py:204-226)."""
        if numpy.isclose(com[2], 0.):
            print("Warning: CoM ill-defined!")
            xstart = self.dpt.shape[0] // 4
            xend = xstart + self.dpt.shape[0] // 2
            ystart = self.dpt.shape[1] // 4
            yend = ystart + self.dpt.shape[1] // 2
            return xstart, xend, ystart, yend, self.minDepth, self.maxDepth
        c0, c1, c2 = float(com[0]), float(com[1]), float(com[2])
        zstart, zend = c2 - size[2] / 2., c2 + size[2] / 2.
        xstart = int(numpy.floor((c0 * c2 / self.fx - size[0] / 2.) / c2 * self.fx + 0.5))
        xend = int(numpy.floor((c0 * c2 / self.fx + size[0] / 2.) / c2 * self.fx + 0.5))
        ystart = int(numpy.floor((c1 * c2 / self.fy - size[1] / 2.) / c2 * self.fy + 0.5))
        yend = int(numpy.floor((c1 * c2 / self.fy + size[1] / 2.) / c2 * self.fy + 0.5))
        return xstart, xend, ystart, yend, zstart, zend

    def comToTransform(self, com, size, dsize=(128, 128)):
        """Affine crop transform off . scale . trans (handdetector.py:228-258); the reference's Python-2 integer
        divisions are kept as floor divisions."""
        xstart, xend, ystart, yend, _, _ = self.comToBounds(com, size)
        wb, hb = (xend - xstart), (yend - ystart)
        if wb > hb:
            s = dsize[0] / float(wb)
            sz = (dsize[0], hb * dsize[0] // wb)
        else:
            s = dsize[1] / float(hb)
            sz = (wb * dsize[1] // hb, dsize[1])
        xs = int(numpy.floor(dsize[0] / 2. - sz[1] / 2.))
        ys = int(numpy.floor(dsize[1] / 2. - sz[0] / 2.))
        return numpy.array([[s, 0., s * float(-xstart) + xs], [0., s, s * float(-ystart) + ys], [0., 0., 1.]])

    # ---- single-crop warps on the device ----------------------------------------------------------------------
    def _run(self, dpt_norm, cube, com, joints3D, M, mode, off=(0., 0., 0.), rot=0., sc=1.):
        from hipdp import ops
        from hipdp.augmenter import MODE_CODE, camera_tuple
        from hipdp.runtime import default_runtime
        rt = default_runtime()
        J = joints3D.shape[0]
        dsz = dpt_norm.shape[0]
        f32 = lambda a: rt.upload(numpy.ascontiguousarray(a, numpy.float32))       # noqa: E731
        com3d = self.importer.jointImgTo3D(com)
        rec = rt.alloc(rt.lib.dpp_augment_record_bytes(), numpy.uint8)
        out_y, out_x = rt.alloc((1, J * 3)), rt.alloc((1, dsz, dsz))
        img = f32(dpt_norm[None])
        ops.augment_prepare(rt, img, f32(com3d[None]), f32(numpy.asarray(cube)[None]), f32(numpy.asarray(M).reshape(1, 9)),
                            f32(joints3D.reshape(1, J, 3)), 1, J, dsz, camera_tuple(self.importer), rec, out_y,
                            mode=rt.upload(numpy.array([MODE_CODE[mode]], numpy.int32)), off=rt.upload(numpy.asarray(off, numpy.float64)),
                            rot=rt.upload(numpy.array([rot], numpy.float64)), sc=rt.upload(numpy.array([sc], numpy.float64)))(rt.stream)
        ops.augment_warp(rt, img, rec, 1, dsz, out_x)(rt.stream)
        rt.synchronize()
        return out_x.get()[0], out_y.get().reshape(J, 3)

    def _normalise(self, dpt, cube, com):
        d = numpy.asarray(dpt, numpy.float32).copy()
        d[d == 0] = com[2] + cube[2] / 2.
        return (d - com[2]) / (cube[2] / 2.)

    def moveCoM(self, dpt, cube, com, off, joints3D, M, pad_value=0):
        """Simulate a different CoM on an already cropped image (handdetector.py:678-710).  dpt is in mm like in the
        reference; returns (new_dpt [normalised to the NEW CoM, far plane filled], new_joints3D, new_com, Mnew)."""
        if numpy.allclose(off, 0.):
            return dpt, joints3D, com, M
        new_com = self.importer.joint3DToImg(self.importer.jointImgTo3D(com) + off)
        img, lab = self._run(self._normalise(dpt, cube, com), cube, com, numpy.asarray(joints3D, numpy.float32), M, 'com', off=off)
        Mnew = self.comToTransform(new_com, cube, dpt.shape) if not (numpy.allclose(com[2], 0.) or numpy.allclose(new_com[2], 0.)) else M
        return img, lab * (cube[2] / 2.), new_com, Mnew

    def rotateHand(self, dpt, cube, com, rot, joints3D, pad_value=0):
        """In-plane rotation about the crop centre (handdetector.py:712-747)."""
        if numpy.allclose(rot, 0.):
            return dpt, joints3D, rot
        M = self.comToTransform(com, cube, dpt.shape)
        img, lab = self._run(self._normalise(dpt, cube, com), cube, com, numpy.asarray(joints3D, numpy.float32), M, 'rot', rot=rot)
        return img, lab * (cube[2] / 2.), numpy.mod(rot, 360)

    def scaleHand(self, dpt, cube, com, sc, joints3D, M, pad_value=0):
        """Re-crop with a scaled metric cube (handdetector.py:750-780)."""
        if numpy.allclose(sc, 1.):
            return dpt, joints3D, cube, M
        new_cube = [s * sc for s in cube]
        img, _ = self._run(self._normalise(dpt, cube, com), cube, com, numpy.asarray(joints3D, numpy.float32), M, 'sc', sc=sc)
        Mnew = self.comToTransform(com, new_cube, dpt.shape) if not numpy.allclose(com[2], 0.) else M
        return img, joints3D, new_cube, Mnew

    # ---- pose-space sampling for the PCA prior (one-off set-up, host) ------------------------------------------------
    @staticmethod
    def sampleRandomPoses(importer, rng, base_poses, base_com, base_cube, num_poses, aug_modes, retall=False, rot3D=False,
                          sigma_com=None, sigma_sc=None, rot_range=None):
        """Random pose augmentation in label space only, used once to fit the 30-D PCA prior
        (handdetector.py:805-909; main_nyu_posereg_embedding.py:86-88).  Same draws from `rng`, in the same order."""
        sigma_com = 5. if sigma_com is None else sigma_com
        sigma_sc = 0.02 if sigma_sc is None else sigma_sc
        rot_range = 180. if rot_range is None else rot_range
        simple = ('none', 'rot', 'sc', 'com')
        combo = ('rot+com', 'com+rot')
        combo_sc = ('rot+com+sc', 'rot+sc+com')
        assert all(m in simple + combo + combo_sc + ('sc+rot+com', 'sc+com+rot', 'com+sc+rot', 'com+rot+sc') for m in aug_modes)
        n = int(num_poses)
        new_poses = numpy.zeros((n, base_poses.shape[1], base_poses.shape[2]), dtype=base_poses.dtype)
        new_com = numpy.zeros((n, 3), dtype=base_poses.dtype)
        new_cube = numpy.zeros((n, 3), dtype=base_poses.dtype)
        modes = rng.randint(0, len(aug_modes), n)
        ridxs = rng.randint(0, base_poses.shape[0], n)
        off = rng.randn(n, 3) * sigma_com
        sc = numpy.fabs(rng.randn(n) * sigma_sc + 1.)
        rot = rng.uniform(-rot_range, rot_range, size=(n, 3))
        if aug_modes == ['none']:
            out = base_poses / (base_cube[:, 2] / 2.)[:, None, None]
            return (out, base_com, base_cube) if retall else out
        # The reference loops over the n (= 1e6 in the scripts) samples in Python; the same arithmetic, in the same operation
        # order and precision (float64 products rounded to the arrays' float32 where the reference rounds), is done here per
        # augmentation mode on whole index sets.
        f32 = numpy.float32
        fx, fy, ux, uy, flip = float(importer.fx), float(importer.fy), float(importer.ux), float(importer.uy), bool(importer.flip_y)

        def to_img(p):                      # joints3DToImg on (..., 3), DepthImporter.joint3DToImg
            p = numpy.asarray(p, numpy.float64)
            z = p[..., 2]
            ok = z != 0.
            zz = numpy.where(ok, z, 1.)
            u = numpy.where(ok, p[..., 0] / zz * fx + ux, ux)
            v = numpy.where(ok, (uy - p[..., 1] / zz * fy) if flip else (p[..., 1] / zz * fy + uy), uy)
            return numpy.stack([u, v, numpy.where(ok, z, 0.)], axis=-1).astype(f32)

        def to_3d(q):                       # jointsImgTo3D on (..., 3)
            q = numpy.asarray(q, numpy.float64)
            x = (q[..., 0] - ux) * q[..., 2] / fx
            y = ((uy - q[..., 1]) if flip else (q[..., 1] - uy)) * q[..., 2] / fy
            return numpy.stack([x, y, q[..., 2]], axis=-1).astype(f32)

        def rot_2d(pts, center, angle):     # rotatePoints2D about per-sample centres, transformations.py:71-88
            alpha = (angle * numpy.pi / 180.)[:, None]
            pp0 = (pts[..., 0] - center[:, None, 0]).astype(f32)
            pp1 = (pts[..., 1] - center[:, None, 1]).astype(f32)
            r0 = (pp0.astype(numpy.float64) * numpy.cos(alpha) - pp1.astype(numpy.float64) * numpy.sin(alpha)).astype(f32)
            r1 = (pp0.astype(numpy.float64) * numpy.sin(alpha) + pp1.astype(numpy.float64) * numpy.cos(alpha)).astype(f32)
            out = pts.copy()
            out[..., 0] = r0 + center[:, None, 0]
            out[..., 1] = r1 + center[:, None, 1]
            return out

        def rot_3d(pts, center, angles):    # rotatePoints3D about per-sample centres with per-sample angles, transformations.py:105-155
            from data.transformations import euler_rxyz_matrix
            a = numpy.asarray(angles, numpy.float64) * numpy.pi / 180.
            R = euler_rxyz_matrix(a[:, 0], a[:, 1], a[:, 2])
            rel = (pts - center[:, None]).astype(numpy.float64)             # the offset is formed in the points' precision
            return (numpy.einsum('nab,njb->nja', R, rel) + center[:, None].astype(numpy.float64)).astype(pts.dtype)

        dt = base_poses.dtype
        mode_of = numpy.asarray([aug_modes[m] for m in range(len(aug_modes))])
        mname = mode_of[modes]
        cube_all, com_all, pose_all = base_cube[ridxs], base_com[ridxs], base_poses[ridxs]
        for mode in sorted(set(mname.tolist())):
            I = numpy.nonzero(mname == mode)[0]
            cube, com3D, pose = cube_all[I], com_all[I], pose_all[I]
            if mode == 'com':
                nc = (com3D + off[I]).astype(dt)
                new_com[I], new_cube[I] = nc, cube
                new_poses[I] = (pose + com3D[:, None] - nc[:, None]) / (new_cube[I][:, 2] / 2.)[:, None, None]
            elif mode == 'rot':
                new_com[I], new_cube[I] = com3D, cube
                nc = new_com[I]
                if rot3D:
                    new_poses[I] = (rot_3d(pose + nc[:, None], nc, rot[I]) - nc[:, None]) / (new_cube[I][:, 2] / 2.)[:, None, None]
                    continue
                joint_2D = to_img(pose + nc[:, None])
                data_2D = rot_2d(joint_2D, to_img(com3D), rot[I, 0])
                new_poses[I] = (to_3d(data_2D) - nc[:, None]) / (new_cube[I][:, 2] / 2.)[:, None, None]
            elif mode == 'sc':
                new_com[I] = com3D
                new_cube[I] = cube * sc[I].astype(f32)[:, None]          # float32 array x float64 scalar stays float32 in the reference's NumPy
                new_poses[I] = pose / (new_cube[I][:, 2] / 2.)[:, None, None]
            elif mode == 'none':
                new_com[I], new_cube[I] = com3D, cube
                new_poses[I] = pose / (new_cube[I][:, 2] / 2.)[:, None, None]
            elif mode in combo or mode in combo_sc:
                nc = (com3D + off[I]).astype(dt)
                new_com[I], new_cube[I] = nc, cube
                p = pose + com3D[:, None] - new_com[I][:, None]
                if mode in combo_sc:
                    p = p * sc[I].astype(f32)[:, None, None]
                if rot3D:                    # handdetector.py:891, 903: about the NEW centre, re-centred on it
                    nc = new_com[I]
                    new_poses[I] = (rot_3d(p + nc[:, None], nc, rot[I]) - nc[:, None]) / (new_cube[I][:, 2] / 2.)[:, None, None]
                    continue
                joint_2D = to_img(p + com3D[:, None])
                data_2D = rot_2d(joint_2D, to_img(new_com[I]), rot[I, 0])
                new_poses[I] = (to_3d(data_2D) - com3D[:, None]) / (new_cube[I][:, 2] / 2.)[:, None, None]
            else:
                raise NotImplementedError()
        return (new_poses, new_com, new_cube, rot) if retall else new_poses

    # ---- initial crop on the device ---------------------------------------------------------------------------------------
    def getNDValue(self):
        """Value of 'not defined' depth (handdetector.py:122-130): the most frequent out-of-range value -- 0 after the
        constructor zeroed everything outside [minDepth, maxDepth]."""
        lo, hi = self.dpt[self.dpt < self.minDepth], self.dpt[self.dpt > self.maxDepth]
        vals = lo if lo.shape[0] > hi.shape[0] else hi
        if vals.shape[0] == 0:
            return 0.
        u, c = numpy.unique(vals, return_counts=True)
        return u[numpy.argmax(c)]

    def cropArea3D(self, com=None, size=(250, 250, 250), dsize=(128, 128), docom=False):
        """Crop the metric cube `size` (mm) around `com` (image coordinates, z in mm) and resize it to `dsize`
        (handdetector.py:382-490).  Returns (crop in mm, crop transform M, com) like the reference."""
        if len(size) != 3 or len(dsize) != 2:
            raise ValueError("Size must be 3D and dsize 2D bounding box")
        if com is None:
            com = self.calculateCoM(self.dpt)            # handdetector.py:401-402: centre of mass of the whole (range-limited) frame
        if dsize[0] != dsize[1]:
            raise NotImplementedError("square destination sizes only")
        frame = numpy.asarray(self.dpt, numpy.float32)[None]
        cube = numpy.asarray(size, numpy.float32)[None]
        nd = self.getNDValue()
        crops, Ms, coms = crop_frames(frame, numpy.asarray(com, numpy.float32)[None], cube, self.fx, self.fy, dsize[0], normalize=False,
                                      nd_value=nd, docom=docom, return_com=True)
        if docom and self.refineNet is not None and self.importer is not None:
            # handdetector.py:429-440: a ScaleNet regresses the offset of the true CoM from the crop; crop again around it.  The net
            # looks at resizeCrop(cropped, dsize): the window resized to dsize AS IT IS (:430), not the aspect-preserving paste
            # (at the net's own input size: the reference passes dsize, which has to be that size there)
            dims = self.refineNet.cfgParams.inputDim
            rs = int((dims[0] if isinstance(dims[0], (list, tuple)) else dims)[2])
            rz, _ = crop_frames(frame, coms[0][None], cube, self.fx, self.fy, rs, normalize=False, nd_value=nd, stretch=True)
            newCom3D = self.refineCoM(rz[0], size, coms[0]) + self.importer.jointImgTo3D(coms[0])
            com2 = numpy.asarray(self.importer.joint3DToImg(newCom3D), numpy.float64)
            if numpy.allclose(com2, 0.):
                com2[2] = crops[0][crops[0].shape[0] // 2, crops[0].shape[1] // 2]
            crops, Ms, coms = crop_frames(frame, com2.astype(numpy.float32)[None], cube, self.fx, self.fy, dsize[0], normalize=False,
                                          nd_value=nd, return_com=True)
        return crops[0], Ms[0].astype(numpy.float64), (coms[0].astype(numpy.float64) if docom else com)

    def refineCoM(self, cropped, size, com):
        """Offset (mm) of the hand centre predicted by the refinement net from a crop around `com` (handdetector.py:634-676):
        normalise and clamp the crop to the cube, feed it with its 1/2 and 1/4 centre crops."""
        imgD = numpy.asarray(cropped, 'float32').copy()
        imgD[imgD == 0] = com[2] + (size[2] / 2.)
        imgD[imgD >= com[2] + (size[2] / 2.)] = com[2] + (size[2] / 2.)
        imgD[imgD <= com[2] - (size[2] / 2.)] = com[2] - (size[2] / 2.)
        imgD -= com[2]
        imgD /= (size[2] / 2.)
        test_data = numpy.zeros((1, 1, cropped.shape[0], cropped.shape[1]), dtype='float32')
        test_data[0, 0] = imgD
        inputs = [test_data]
        for k in (2, 4):
            dsize = (int(test_data.shape[2] // k), int(test_data.shape[3] // k))
            xstart = int(test_data.shape[2] / 2 - dsize[0] / 2)
            ystart = int(test_data.shape[3] / 2 - dsize[1] / 2)
            inputs.append(numpy.ascontiguousarray(test_data[:, :, ystart:ystart + dsize[1], xstart:xstart + dsize[0]]))
        if self.refineNet.cfgParams.numInputs == 1:
            jts = self.refineNet.computeOutput(test_data)
        elif self.refineNet.cfgParams.numInputs == 3:
            jts = self.refineNet.computeOutput(inputs)
        else:
            raise NotImplementedError("Number of inputs is {}".format(self.refineNet.cfgParams.numInputs))
        return jts[0] * (size[2] / 2.)

    def calculateCoM(self, dpt):
        """Centre of mass (mean column, mean row, mean depth) of the pixels inside [minDepth, maxDepth]
        (handdetector.py:91-108); host NumPy -- the per-crop version used by cropArea3D(docom=True) runs on the device."""
        dc = numpy.asarray(dpt).copy()
        dc[dc < self.minDepth] = 0
        dc[dc > self.maxDepth] = 0
        ys, xs = numpy.nonzero(dc > 0)
        num = numpy.count_nonzero(dc)
        if num == 0:
            return numpy.array((0, 0, 0), float)
        return numpy.array((xs.mean() * num, ys.mean() * num, dc.sum()), float) / num

    def getCrop(self, dpt, xstart, xend, ystart, yend, zstart, zend, thresh_z=True, background=0):
        """Window of the frame, zero-padded where it leaves the frame, z-thresholded (handdetector.py:260-296)."""
        if len(dpt.shape) != 2:
            raise NotImplementedError()
        H, W = dpt.shape
        cropped = dpt[max(ystart, 0):min(yend, H), max(xstart, 0):min(xend, W)].copy()
        cropped = numpy.pad(cropped, ((abs(ystart) - max(ystart, 0), abs(yend) - min(yend, H)),
                                      (abs(xstart) - max(xstart, 0), abs(xend) - min(xend, W))), mode='constant', constant_values=background)
        if thresh_z is True:
            msk1 = numpy.logical_and(cropped < zstart, cropped != 0)
            msk2 = numpy.logical_and(cropped > zend, cropped != 0)
            cropped[msk1] = zstart
            cropped[msk2] = 0.
        return cropped

    def refineCoMIterative(self, com, num_iter, size=(250, 250, 250)):
        """Re-centre the cube on the centre of mass of its own content, num_iter times (handdetector.py:540-558)."""
        for _ in range(num_iter):
            xstart, xend, ystart, yend, zstart, zend = self.comToBounds(com, size)
            cropped = self.getCrop(self.dpt, xstart, xend, ystart, yend, zstart, zend)
            com = self.calculateCoM(cropped)
            if numpy.allclose(com, 0.):
                com[2] = cropped[cropped.shape[0] // 2, cropped.shape[1] // 2]
            com[0] += max(xstart, 0)
            com[1] += max(ystart, 0)
        return com

    def detect(self, *args, **kwargs):
        raise NotImplementedError("hand detection / tracking (cv2.findContours slab analysis, handdetector.py:504-631) belongs to the realtime demo "
                                  "(util/realtimehandposepipeline.py is its only caller), which is out of scope (SURVEY.md section 2)")

    track = detect


def crop_frames(frames, coms, cubes, fx, fy, dsize=128, normalize=True, nd_value=0., runtime=None, docom=False, return_com=False,
                stretch=False):
    """Batched cropArea3D (+ Dataset.imgStackDepthOnly when normalize): frames (B, H, W) raw depth in mm, coms (B, 3) crop
    centres in image coordinates, cubes (B, 3) in mm -> (crops (B, dsize, dsize) float32, M (B, 3, 3) float32[, coms]).
    Two kernel launches for the whole batch (csrc/augment.hip: crop_prepare / crop_warp); docom=True re-centres every crop
    on the centre of mass of its first window (two more launches), as handdetector.py:413-427 does; stretch=True resizes the
    window to dsize x dsize as it is (resizeCrop(cropped, dsize), the refinement net's input, :430)."""
    from hipdp import ops
    from hipdp.runtime import default_runtime
    rt = runtime or default_runtime()
    frames = numpy.ascontiguousarray(frames, numpy.float32)
    B, H, W = frames.shape
    fr = rt.upload(frames)
    co = rt.upload(numpy.ascontiguousarray(coms, numpy.float32).reshape(B, 3))
    cu = rt.upload(numpy.ascontiguousarray(cubes, numpy.float32).reshape(B, 3))
    rec = rt.alloc(B * rt.lib.dpp_crop_record_bytes(), numpy.uint8)
    out, M = rt.alloc((B, dsize, dsize), zero=False), rt.alloc((B, 9), zero=False)
    ops.crop_prepare(rt, fr, B, H, W, co, cu, fx, fy, dsize, rec, M, stretch=stretch)(rt.stream)
    if docom:
        co2 = rt.alloc((B, 3), zero=False)
        ops.crop_com(rt, fr, rec, B, H, W, co2)(rt.stream)
        ops.crop_prepare(rt, fr, B, H, W, co2, cu, fx, fy, dsize, rec, M, stretch=stretch)(rt.stream)
        co = co2
    ops.crop_warp(rt, fr, rec, B, H, W, dsize, out, normalize=normalize, nd_value=nd_value)(rt.stream)
    rt.synchronize()
    if return_com:
        return out.get(), M.get().reshape(B, 3, 3), co.get()
    return out.get(), M.get().reshape(B, 3, 3)
