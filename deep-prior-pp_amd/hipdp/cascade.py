"""
The CoM-refinement cascade on the device: frame -> crop -> centre of mass -> crop -> ScaleNet offset -> refined centre -> final crop.

What the reference does per frame on the host, in `HandDetector.cropArea3D(docom=True)` with a `refineNet`
(/root/reference/src/util/handdetector.py:382-490, refineCoM :634-676; called for every frame by the importers when
`di.refineNet` is set, /root/reference/src/data/importers.py:382-396, and by the realtime / test scripts), is ONE launch plan
over a whole batch of frames here:

    crop_prepare(com0) -> crop_com -> crop_prepare(com1, stretch) -> crop_warp(rsz, normalised)   cropArea3D's docom branch, :430
    crop_center x2 -> the refinement net's deterministic forward plan                          refineCoM, :634-676
    crop_refine: com2 = joint3DToImg(out * cube_z/2 + jointImgTo3D(com1)) (+ the labels)       :431-433
    crop_prepare(com2) -> crop_warp(dsz, normalised or mm)                                     :435-440, dataset.py:97-103

No host step in between, so the plan can be put in front of a train step of the pose regressor (`step_plan(before=...)`):
BASELINE.json's config 5 ("com_refine + posereg cascade", bench.py --workload cascade).  refineCoM's input normalisation
(0 -> far plane, clamp to the cube, (d - com_z) / (cube_z/2)) is what crop_warp's normalisation produces: getCrop has already
clamped the window to [zstart, zend] (handdetector.py:260-296), so the two clamps of refineCoM are no-ops on it.
"""
import numpy as np

from . import engine, ops
from .augmenter import camera_tuple


class CascadeCropper(object):
    def __init__(self, rt, importer, refineNet, B, H, W, dsize=128, refine_size=None, normalize=True, nd_value=0., fx=None, fy=None,
                 frames=None, coms=None, cubes=None, out=None, gt3d=None, J=0, proj=None, out_y=None):
        """
        :param importer:   the dataset importer (camera; NYU / MSRA flip the y axis)
        :param refineNet:  a ScaleNet-like net whose output is the normalised 3-D offset of the hand centre (numInputs 1 or 3),
                           or None: plain docom re-centring
        :param B, H, W:    frames per call and their size
        :param dsize:      side of the final crop (128; 256 for the config-5 stress)
        :param refine_size side of the crop the refinement net looks at (default: the net's input size)
        :param frames, coms, cubes, out: device buffers to work on ([B][H][W], [B][3], [B][3], [B][dsize][dsize]); allocated if None
        :param gt3d, J, proj, out_y: optional labels of the final crop: gt3Dorig [B][J][3] -> out_y (see dpp_crop_refine)
        """
        self.rt, self.B, self.H, self.W, self.dsize = rt, int(B), int(H), int(W), int(dsize)
        self.normalize, self.nd_value = bool(normalize), float(nd_value)
        self.cam = camera_tuple(importer)
        self.fx = abs(float(importer.fx if fx is None else fx))              # HandDetector(dpt, abs(di.fx), abs(di.fy), ...)
        self.fy = abs(float(importer.fy if fy is None else fy))
        self.net = refineNet
        f32 = np.float32
        self.frames = frames if frames is not None else rt.alloc((B, H, W), f32, zero=False)
        self.com0 = coms if coms is not None else rt.alloc((B, 3), f32, zero=False)
        self.cube = cubes if cubes is not None else rt.alloc((B, 3), f32, zero=False)
        self.out = out if out is not None else rt.alloc((B, dsize, dsize), f32, zero=False)
        self.M = rt.alloc((B, 9), f32, zero=False)
        self.rec = rt.alloc(B * rt.lib.dpp_crop_record_bytes(), np.uint8)
        self.com1 = rt.alloc((B, 3), f32, zero=False)
        self.com2 = self.com1
        self.com3d = rt.alloc((B, 3), f32, zero=False)
        self.gt3d, self.J, self.out_y = gt3d, int(J), out_y
        self.gt3d_crop = rt.alloc((B, J, 3), f32, zero=False) if gt3d is not None else None
        pm = pc = None
        E = 0
        if proj is not None:
            pm = rt.upload(np.asarray(proj.mean_, f32))
            pc = rt.upload(np.asarray(proj.components_, f32))
            E = int(proj.components_.shape[0])
        plan = ops.Plan('cascade')
        fr, rec = self.frames, self.rec
        if refineNet is None:
            rs = self.dsize
        else:
            dims = refineNet.cfgParams.inputDim
            d0 = dims[0] if isinstance(dims[0], (list, tuple)) else dims
            rs = int(refine_size or d0[2])
        plan.add(ops.crop_prepare(rt, fr, B, H, W, self.com0, self.cube, self.fx, self.fy, rs, rec, None))
        plan.add(ops.crop_com(rt, fr, rec, B, H, W, self.com1))
        if refineNet is not None:
            nin = int(getattr(refineNet.cfgParams, 'numInputs', 1))
            if nin not in (1, 3):
                raise NotImplementedError("Number of inputs is {}".format(nin))
            self.eng = engine.CompiledNet(refineNet, train=False, runtime=rt)
            nb = self.eng.N
            if B < nb:
                raise ValueError("the cascade needs at least one net batch of frames (%d < %d): pad the batch" % (B, nb))
            if self.eng.out_dim != 3:
                raise ValueError("the refinement net must regress one 3-D offset")
            self.com2 = rt.alloc((B, 3), f32, zero=False)
            self.crop_r = rt.alloc((B, rs, rs), f32, zero=False)
            self.net_out = rt.alloc((B, 3), f32, zero=False)
            plan.add(ops.crop_prepare(rt, fr, B, H, W, self.com1, self.cube, self.fx, self.fy, rs, rec, None, stretch=True))
            plan.add(ops.crop_warp(rt, fr, rec, B, H, W, rs, self.crop_r, normalize=True, nd_value=0.0))
            x_ins = self.eng.x_ins
            # the net's batch is fixed: walk the frames in chunks of it (the last chunk may overlap the one before: a frame's
            # output does not depend on its batch in deterministic mode)
            starts = list(range(0, B - nb + 1, nb))
            if starts[-1] + nb < B:
                starts.append(B - nb)
            for s0 in starts:
                src = self.crop_r.view(s0 * rs * rs, (nb, rs, rs))
                plan.add(ops.copy2d(rt, src.reshape(nb * rs * rs), rs * rs, x_ins[0].buf.reshape(nb * rs * rs), rs * rs, nb, rs * rs,
                                    name='cascade_in0'))
                for k in range(1, nin):                   # 1/2 and 1/4 CENTRE crops, handdetector.py:654-666
                    f = 2 ** k
                    plan.add(ops.crop_center(rt, src, nb, rs, rs, x_ins[k].buf, rs // f, rs // f, name='cascade_in%d' % k))
                for op, side in self.eng.fwd.ops:
                    plan.add(op, side)
                plan.add(ops.copy2d(rt, self.eng.out.buf.reshape(nb * 3), 3, self.net_out.view(s0 * 3, (nb * 3,)), 3, nb, 3,
                                    name='cascade_out'))
            plan.add(ops.crop_refine(rt, fr, rec, B, H, W, self.com1, self.cube, self.net_out, self.cam, self.com2, gt3d_orig=gt3d, J=J,
                                     pca_mean=pm, pca_comp=pc, E=E, com3d_out=self.com3d, gt3d_crop=self.gt3d_crop, out_y=out_y))
        plan.add(ops.crop_prepare(rt, fr, B, H, W, self.com2, self.cube, self.fx, self.fy, self.dsize, rec, self.M))
        plan.add(ops.crop_warp(rt, fr, rec, B, H, W, self.dsize, self.out, normalize=self.normalize, nd_value=self.nd_value))
        self.plan = plan

    def run(self):
        self.plan.run(self.rt)

    def __call__(self, frames, coms, cubes):
        """Host arrays in, host arrays out: (crops [B][dsize][dsize], M [B][3][3], com [B][3] in image coordinates)."""
        self.frames.set(np.ascontiguousarray(frames, np.float32))
        self.com0.set(np.ascontiguousarray(coms, np.float32).reshape(self.B, 3))
        self.cube.set(np.ascontiguousarray(cubes, np.float32).reshape(self.B, 3))
        self.run()
        self.rt.synchronize()
        return self.out.get(), self.M.get().reshape(self.B, 3, 3), self.com2.get()
