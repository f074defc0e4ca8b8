"""
Camera models of the dataset importers (API of /root/reference/src/data/importers.py:47-119, 187-210, 529-566, 756-793,
878-914, 1187-1224): intrinsics, default crop cubes and the pinhole (un)projection that the augmentation and the
evaluation use.  Reading the ICVL / MSRA15 / NYU files from disk (`loadSequence`) is listed under "next" in
SURVEY.md section 8(f) and raises NotImplementedError.
"""
import numpy as np


class DepthImporter(object):
    flip_y = False

    def __init__(self, fx, fy, ux, uy, hand=None):
        self.fx, self.fy, self.ux, self.uy = fx, fy, ux, uy
        self.depth_map_size = (320, 240)
        self.refineNet = None
        self.crop_joint_idx = 0
        self.hand = hand

    def jointsImgTo3D(self, sample):
        ret = np.zeros((sample.shape[0], 3), np.float32)
        for i in range(sample.shape[0]):
            ret[i] = self.jointImgTo3D(sample[i])
        return ret

    def jointImgTo3D(self, sample):
        """(u, v, d) in image coordinates / mm -> metric 3-D (x, y, z); the y axis is negated for NYU / MSRA."""
        s0, s1, s2 = float(sample[0]), float(sample[1]), float(sample[2])
        ret = np.zeros((3,), np.float32)
        ret[0] = (s0 - self.ux) * s2 / self.fx
        ret[1] = ((self.uy - s1) if self.flip_y else (s1 - self.uy)) * s2 / self.fy
        ret[2] = s2
        return ret

    def joints3DToImg(self, sample):
        ret = np.zeros((sample.shape[0], 3), np.float32)
        for i in range(sample.shape[0]):
            ret[i] = self.joint3DToImg(sample[i])
        return ret

    def joint3DToImg(self, sample):
        s0, s1, s2 = float(sample[0]), float(sample[1]), float(sample[2])
        ret = np.zeros((3,), np.float32)
        if s2 == 0.:
            ret[0] = self.ux
            ret[1] = self.uy
            return ret
        ret[0] = s0 / s2 * self.fx + self.ux
        ret[1] = (self.uy - s1 / s2 * self.fy) if self.flip_y else (s1 / s2 * self.fy + self.uy)
        ret[2] = s2
        return ret

    def getCameraProjection(self):
        ret = np.zeros((4, 4), np.float32)
        ret[0, 0], ret[1, 1], ret[2, 2] = self.fx, self.fy, 1.
        ret[0, 2], ret[1, 2], ret[3, 2] = self.ux, self.uy, 1.
        return ret

    def loadSequence(self, *args, **kwargs):
        raise NotImplementedError("dataset readers are scheduled after the hot path (SURVEY.md section 8(f) rank 3)")


class ICVLImporter(DepthImporter):
    def __init__(self, basepath=None, useCache=True, cacheDir='./cache/', refineNet=None, hand=None):
        super(ICVLImporter, self).__init__(241.42, 241.42, 160., 120., hand)    # importers.py:199
        self.depth_map_size = (320, 240)
        self.basepath, self.useCache, self.cacheDir = basepath, useCache, cacheDir
        self.numJoints = 16
        self.crop_joint_idx = 0
        self.refineNet = refineNet
        self.default_cubes = {'train': (250, 250, 250), 'test_seq_1': (250, 250, 250), 'test_seq_2': (250, 250, 250)}
        self.sides = {'train': 'right', 'test_seq1': 'right', 'test_seq_2': 'right'}


class MSRA15Importer(DepthImporter):
    flip_y = True

    def __init__(self, basepath=None, useCache=True, cacheDir='./cache/', refineNet=None, detectorNet=None, derotNet=None, hand=None):
        super(MSRA15Importer, self).__init__(241.42, 241.42, 160., 120., hand)  # importers.py:547
        self.depth_map_size = (320, 240)
        self.basepath, self.useCache, self.cacheDir = basepath, useCache, cacheDir
        self.refineNet, self.derotNet, self.detectorNet = refineNet, derotNet, detectorNet
        self.numJoints = 21
        self.crop_joint_idx = 5
        self.default_cubes = {'P0': (200, 200, 200), 'P1': (200, 200, 200), 'P2': (200, 200, 200), 'P3': (180, 180, 180),
                              'P4': (180, 180, 180), 'P5': (180, 180, 180), 'P6': (170, 170, 170), 'P7': (160, 160, 160),
                              'P8': (150, 150, 150)}


class NYUImporter(DepthImporter):
    flip_y = True

    def __init__(self, basepath=None, useCache=True, cacheDir='./cache/', refineNet=None, allJoints=False, hand=None):
        super(NYUImporter, self).__init__(588.03, 587.07, 320., 240., hand)     # importers.py:891
        self.depth_map_size = (640, 480)
        self.basepath, self.useCache, self.cacheDir = basepath, useCache, cacheDir
        self.allJoints = allJoints
        self.numJoints = 36
        self.scales = {'train': 1., 'test_1': 1., 'test_2': 0.83, 'test': None, 'train_synth': 1.,
                       'test_synth_1': 1., 'test_synth_2': 0.83, 'test_synth': None}
        self.restrictedJointsEval = [0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32]
        self.refineNet = refineNet
        self.default_cubes = {'train': (300, 300, 300), 'test_1': (300, 300, 300), 'test_2': (250, 250, 250),
                              'test': (300, 300, 300), 'train_synth': (300, 300, 300), 'test_synth_1': (300, 300, 300),
                              'test_synth_2': (250, 250, 250), 'test_synth': (300, 300, 300)}
        self.sides = {'train': 'right', 'test_1': 'right', 'test_2': 'right', 'test': 'right', 'train_synth': 'right',
                      'test_synth_1': 'right', 'test_synth_2': 'right', 'test_synth': 'right'}
        self.crop_joint_idx = 13 if not allJoints else 32
