"""
Dataset importers (API of /root/reference/src/data/importers.py:47-119, 187-420, 529-800, 878-1224): intrinsics, default
crop cubes, the pinhole (un)projection that the augmentation and the evaluation use, and `loadSequence` for the ICVL, MSRA15
and NYU file formats.  Where the reference crops every frame in Python (HandDetector.cropArea3D per frame), the frames of a
sequence are cropped here in batches by the device kernels (util.handdetector.crop_frames); the per-sequence pickle cache
keeps the reference's file naming.
"""
import os
import pickle
import struct

import numpy as np

from data.basetypes import DepthFrame, NamedImgSequence
from data.transformations import transformPoints2D


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

    def getDepthMapNV(self):
        return 32001

    # ---- shared by the loadSequence of the three datasets -------------------------------------------------------------
    def _config(self, seqName, cube):
        if cube is None:
            return {'cube': self.default_cubes[seqName]}
        assert isinstance(cube, tuple) and len(cube) == 3
        return {'cube': cube}

    def _cache_file(self, *parts):
        return '{}/{}_cache.pkl'.format(self.cacheDir, '_'.join(str(p) for p in (self.__class__.__name__,) + parts))

    def _from_cache(self, fn, Nmax, shuffle, rng):
        if not (self.useCache and os.path.isfile(fn)):
            return None
        print("Loading cache data from {}".format(fn))
        with open(fn, 'rb') as f:
            seqName, data, config = pickle.load(f, encoding='latin1')
        data = [DepthFrame(*d) for d in data]
        if shuffle and rng is not None:
            print("Shuffling")
            rng.shuffle(data)
        return NamedImgSequence(seqName, data if np.isinf(Nmax) else data[0:int(Nmax)], config)

    def _to_cache(self, fn, seqName, data, config):
        if not self.useCache:
            return
        print("Save cache data to {}".format(fn))
        os.makedirs(os.path.dirname(fn) or '.', exist_ok=True)
        with open(fn, 'wb') as f:
            pickle.dump((seqName, [tuple(d) for d in data], config), f, protocol=2)

    def _crop_records(self, records, config, docom, side, chunk=256):
        """records: (dpt, gtorig, gt3Dorig, fileName, subSeqName) of the frames that survived the readers' checks.  Crops
        them like `hd.cropArea3D(com=gtorig[crop_joint_idx], size=cube, docom=docom)` does (importers.py:382-396) -- in
        batches on the device unless a refinement net has to look at every crop -- and builds the DepthFrames."""
        from util.handdetector import HandDetector, crop_frames
        data = []
        cube = np.asarray(config['cube'], np.float32)
        for i0 in range(0, len(records), chunk):
            part = records[i0:i0 + chunk]
            if docom and self.refineNet is not None:
                # crop -> CoM -> ScaleNet refinement -> crop (handdetector.py:413-440) for the whole chunk as one device plan
                crops, Ms, coms = self._crop_refined(part, cube)
            else:
                frames = np.stack([r[0] for r in part]).astype(np.float32)
                c0 = np.stack([r[1][self.crop_joint_idx] for r in part]).astype(np.float32)
                crops, Ms, coms = crop_frames(frames, c0, np.tile(cube, (len(part), 1)), self.fx, self.fy, 128, normalize=False,
                                              docom=docom, return_com=True)
                Ms = Ms.astype(np.float64)
            for (dpt, gtorig, gt3Dorig, fileName, subSeqName), c, M, com in zip(part, crops, Ms, coms):
                com3D = self.jointImgTo3D(com)
                gt3Dcrop = gt3Dorig - com3D                              # normalize to com
                gtcrop = transformPoints2D(gtorig, M)
                data.append(DepthFrame(np.asarray(c, np.float32), gtorig, gtcrop, M, gt3Dorig, gt3Dcrop, com3D, fileName, subSeqName,
                                       side, {}))
        return data

    def _crop_refined(self, part, cube):
        from hipdp.cascade import CascadeCropper
        from hipdp.runtime import default_runtime
        frames = np.stack([r[0] for r in part]).astype(np.float32)
        c0 = np.stack([r[1][self.crop_joint_idx] for r in part]).astype(np.float32)
        n = len(part)
        nb = int(self.refineNet.cfgParams.batch_size)
        if n < nb:                                        # a short last chunk: padded by repeating its last frame (netbase.py:285-290)
            frames = np.concatenate([frames, np.repeat(frames[-1:], nb - n, axis=0)])
            c0 = np.concatenate([c0, np.repeat(c0[-1:], nb - n, axis=0)])
        B, H, W = frames.shape
        key = (B, H, W, id(self.refineNet))
        cc = getattr(self, '_cascade', None)
        if cc is None or cc[0] != key:
            self.refineNet.setDeterministic()
            cc = (key, CascadeCropper(default_runtime(), self, self.refineNet, B, H, W, dsize=128, normalize=False))
            self._cascade = cc
        crops, Ms, coms = cc[1](frames, c0, np.tile(cube, (B, 1)))
        return crops[:n], Ms[:n].astype(np.float64), coms[:n]

    def _crop_stream(self, config, docom, side, chunk=256):
        """Incremental form of _crop_records: frames are cropped (on the device) as soon as `chunk` of them have been read and
        only the 128x128 crops are kept, like the reference, which crops frame by frame (importers.py:382-396) -- the raw
        640x480 frames of NYU's 72 757-frame training sequence would need 89 GB of host memory."""
        return _CropStream(self, config, docom, side, chunk)

    def _read_ahead(self, entries, window=32, workers=8):
        """(entry, depth map or None when the file is missing) in the order of `entries` = [(filename, ...), ...]: the frames are
        decoded by a small thread pool a window ahead of the consumer (PNG inflation and file reads release the GIL; the reference
        decodes its 72 757 NYU training frames one by one), so the device crops never wait for a file."""
        import collections
        from concurrent.futures import ThreadPoolExecutor

        def load(fn):
            return self.loadDepthMap(fn) if os.path.isfile(fn) else None
        pending = collections.deque()
        with ThreadPoolExecutor(workers) as pool:
            for e in entries:
                pending.append((e, pool.submit(load, e[0])))
                if len(pending) >= window:
                    e0, f = pending.popleft()
                    yield e0, f.result()
            while pending:
                e0, f = pending.popleft()
                yield e0, f.result()

    @staticmethod
    def _has_content(dpt, tol=1.):
        """HandDetector.checkImage on the detector-preprocessed frame (handdetector.py:53-68, 110-120)."""
        d = np.asarray(dpt, np.float32)
        hi, lo = min(1500, d.max()), max(10, d.min())
        d = np.where((d > hi) | (d < lo), np.float32(0), d)
        return not (np.std(d) < tol)

    def _finish(self, seqName, data, config, cache, shuffle, rng):
        self._to_cache(cache, seqName, data, config)
        if shuffle and rng is not None:
            print("Shuffling")
            rng.shuffle(data)
        return NamedImgSequence(seqName, data, config)


class _CropStream(object):
    def __init__(self, importer, config, docom, side, chunk):
        self.imp, self.config, self.docom, self.side, self.chunk = importer, config, docom, side, int(chunk)
        self.pending, self.data, self.count = [], [], 0

    def __len__(self):
        return self.count

    def append(self, record):
        self.pending.append(record)
        self.count += 1
        if len(self.pending) >= self.chunk:
            self._flush()

    def _flush(self):
        if self.pending:
            self.data.extend(self.imp._crop_records(self.pending, self.config, self.docom, self.side, chunk=self.chunk))
            self.pending = []

    def finish(self):
        self._flush()
        return self.data


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

    def loadDepthMap(self, filename):
        """16-bit single-channel PNG, depth in mm (importers.py:213-223)."""
        from PIL import Image
        img = Image.open(filename)
        assert len(img.getbands()) == 1                     # ensure depth image
        return np.asarray(img, np.float32)

    def _baseline_rows(self, filename, firstName):
        """The non-blank lines of a result file as a (frames, numJoints, 3) float32 array of (u, v, d) triples; with firstName the
        leading file-name token of every line is skipped."""
        off = 1 if firstName else 0
        with open(filename) as fh:
            rows = [ln.strip().split(' ') for ln in fh if ln.rstrip()]
        n = self.numJoints * 3
        return np.asarray([[float(v) for v in r[off:off + n]] for r in rows], np.float32).reshape(len(rows), self.numJoints, 3)

    def loadBaseline(self, filename, firstName=False):
        """Results of a baseline method (e.g. LRF_Results_seq_1.txt: one line of u v d triples per frame) as a list of metric (numJoints, 3)
        arrays (importers.py:422-456)."""
        return [self.jointsImgTo3D(ev) for ev in self._baseline_rows(filename, firstName)]

    def loadBaseline2D(self, filename, firstName=False):
        """The same file as image-plane (numJoints, 2) arrays (importers.py:458-484)."""
        return [np.array(ev[:, :2]) for ev in self._baseline_rows(filename, firstName)]

    def loadSequence(self, seqName, subSeq=None, Nmax=float('inf'), shuffle=False, rng=None, docom=False, cube=None):
        """<basepath>/<seqName>.txt: one line per frame, `relative/path.png u v d` x 16 joints; frames under <basepath>/Depth/
        (importers.py:232-420)."""
        from util.handdetector import HandDetector
        if (subSeq is not None) and (not isinstance(subSeq, list)):
            raise TypeError("subSeq must be None or list")
        config = self._config(seqName, cube)
        mode = HandDetector.detectionModeToString(docom, self.refineNet is not None)
        cache = self._cache_file(seqName, self.hand, mode, config['cube'][0]) if subSeq is None else \
            self._cache_file(seqName, ''.join(subSeq), self.hand, mode, config['cube'][0])
        cached = self._from_cache(cache, Nmax, shuffle, rng)
        if cached is not None:
            return cached
        if self.hand is not None and self.hand != self.sides[seqName]:
            raise NotImplementedError()
        objdir = '{}/Depth/'.format(self.basepath)
        # every ICVL frame is recorded as 'left' (importers.py:401-402); the table of sides -- whose key for the first test sequence
        # is misspelt in the reference, :211 -- is only consulted when a hand was asked for
        records = self._crop_stream(config, docom, 'left')
        entries = []
        with open('{}/{}.txt'.format(self.basepath, seqName)) as inputfile:
            for line in inputfile:
                part = line.split(' ')
                subSeqName = ''
                if subSeq is not None:
                    p = part[0].split('/')
                    long_name = len(p[0]) > 6               # frames of the un-rotated sequence "0" live in long-named folders
                    if (long_name and '0' not in subSeq) or (not long_name and p[0] not in subSeq):
                        continue
                    subSeqName = p[0] if not long_name else '0'
                entries.append(('{}/{}'.format(objdir, part[0]), part, subSeqName))
        for (dptFileName, part, subSeqName), dpt in self._read_ahead(entries):
            if len(records) >= Nmax:
                break
            if dpt is None:
                print("File {} does not exist!".format(dptFileName))
                continue
            gtorig = np.zeros((self.numJoints, 3), np.float32)
            for joint in range(self.numJoints):
                for xyz in range(0, 3):
                    gtorig[joint, xyz] = part[joint * 3 + xyz + 1]
            gt3Dorig = self.jointsImgTo3D(gtorig)        # normalized joints in 3D coordinates
            if not self._has_content(dpt, 1):
                print("Skipping image {}, no content".format(dptFileName))
                continue
            records.append((dpt, gtorig, gt3Dorig, dptFileName, subSeqName))
        print("Loaded {} samples.".format(len(records)))
        data = records.finish()
        return self._finish(seqName, data, config, cache, shuffle, rng)


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
        self.sides = {'P%d' % i: 'right' for i in range(9)}

    def loadDepthMap(self, filename):
        """Binary patch: int32 width, height, left, top, right, bottom, then float32 depth of the [top:bottom, left:right]
        window (importers.py:570-587)."""
        with open(filename, 'rb') as f:
            width, height, left, top, right, bottom = struct.unpack('6i', f.read(24))
            patch = np.fromfile(f, dtype='float32', sep="")
        imgdata = np.zeros((height, width), dtype='float32')
        imgdata[top:bottom, left:right] = patch.reshape([bottom - top, right - left])
        return imgdata

    def loadSequence(self, seqName, subSeq=None, Nmax=float('inf'), shuffle=False, rng=None, docom=False, cube=None):
        """<basepath>/<subject>/<gesture>/joint.txt (count, then 21 x (x y z) per line, z negated) next to
        NNNNNN_depth.bin (importers.py:596-700)."""
        from util.handdetector import HandDetector
        if (subSeq is not None) and (not isinstance(subSeq, list)):
            raise TypeError("subSeq must be None or list")
        config = self._config(seqName, cube)
        mode = HandDetector.detectionModeToString(docom, self.refineNet is not None)
        cache = self._cache_file(seqName, self.hand, mode, config['cube'][0]) if subSeq is None else \
            self._cache_file(seqName, self.hand, ''.join(subSeq), mode, config['cube'][0])
        cached = self._from_cache(cache, Nmax, shuffle, rng)
        if cached is not None:
            return cached
        objdir = '{}/{}/'.format(self.basepath, seqName)
        subdirs = sorted([name for name in os.listdir(objdir) if os.path.isdir(os.path.join(objdir, name))])
        records = self._crop_stream(config, docom, self.sides[seqName])
        entries = []
        for subdir in subdirs:
            subSeqName = ''
            if subSeq is not None:
                if subdir not in subSeq:
                    continue
                subSeqName = subdir
            with open('{}/{}/joint.txt'.format(objdir, subdir)) as inputfile:
                nImgs = int(inputfile.readline())
                for i in range(nImgs):
                    part = inputfile.readline().split(' ')
                    entries.append(('{}/{}/{}_depth.bin'.format(objdir, subdir, str(i).zfill(6)), part, subSeqName))
        for (dptFileName, part, subSeqName), dpt in self._read_ahead(entries):
            if len(records) >= Nmax:
                break
            if dpt is None:
                print("File {} does not exist!".format(dptFileName))
                continue
            gt3Dorig = np.zeros((self.numJoints, 3), np.float32)
            for joint in range(gt3Dorig.shape[0]):
                for xyz in range(0, 3):
                    gt3Dorig[joint, xyz] = part[joint * 3 + xyz]
            gt3Dorig[:, 2] *= (-1.)                  # the files hold -z
            gtorig = self.joints3DToImg(gt3Dorig)
            if self.hand is not None and self.hand != self.sides[seqName]:
                gtorig[:, 0] -= dpt.shape[1] / 2.
                gtorig[:, 0] *= (-1)
                gtorig[:, 0] += dpt.shape[1] / 2.
                gt3Dorig = self.jointsImgTo3D(gtorig)
                dpt = dpt[:, ::-1]
            if not self._has_content(dpt, 1.):
                print("Skipping image {}, no content".format(dptFileName))
                continue
            records.append((dpt, gtorig, gt3Dorig, dptFileName, subSeqName))
        print("Loaded {} samples.".format(len(records)))
        data = records.finish()
        return self._finish(seqName, data, config, cache, shuffle, rng)


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

    def loadDepthMap(self, filename):
        """RGB PNG with the depth in the green (high byte) and blue (low byte) channels (importers.py:917-931)."""
        from PIL import Image
        img = Image.open(filename)
        assert len(img.getbands()) == 3
        _, g, b = img.split()
        g, b = np.asarray(g, np.int32), np.asarray(b, np.int32)
        return np.asarray(np.bitwise_or(np.left_shift(g, 8), b), np.float32)

    @staticmethod
    def _predicted_uv(filename):
        """`pred_joint_uvconf` of camera 0 from Tompson et al.'s test_predictions.mat, with the all-zero (unused) joint slots of every frame
        dropped: (frames, numJoints, 3) of (u, v, confidence); numJoints = the length of `conv_joint_names`."""
        import scipy.io
        mat = scipy.io.loadmat(filename)
        numJoints = mat['conv_joint_names'][0].shape[0]
        joints = np.asarray(mat['pred_joint_uvconf'][0])
        out = np.zeros((joints.shape[0], numJoints, 3), np.float64)
        for f in range(joints.shape[0]):
            used = joints[f][np.count_nonzero(joints[f], axis=1) != 0]
            out[f, :used.shape[0]] = used[:numJoints]
        return numJoints, out

    def loadBaseline(self, filename, gt=None):
        """Baseline predictions as a list of metric (numJoints, 3) arrays (importers.py:1079-1145).  With `gt` (frames, >= 14 joints, 3:
        the ground truth in image coordinates) the file is the .mat of 2-D predictions: depth is read off the frame `depth_1_NNNNNNN.png`
        beside it at the predicted pixel and replaced by the ground-truth depth where it is more than 150 mm from that of joint 13 (the
        palm: a prediction that fell onto the background); frames whose image is missing are skipped.  Without `gt` the file is text, one
        line of u v d triples per frame."""
        if gt is None:
            with open(filename) as fh:
                rows = [ln.rstrip().split(' ') for ln in fh if ln.rstrip()]
            self.numJoints = len(rows[0]) // 3
            n = self.numJoints * 3
            return [self.jointsImgTo3D(np.asarray([float(v) for v in r[:n]], np.float32).reshape(self.numJoints, 3)) for r in rows]
        self.numJoints, pred = self._predicted_uv(filename)
        data = []
        for dat in range(min(pred.shape[0], gt.shape[0])):
            fname = '{0:s}/depth_1_{1:07d}.png'.format(os.path.split(filename)[0], dat + 1)
            if not os.path.isfile(fname):
                continue
            dm = self.loadDepthMap(fname)
            ev = np.zeros((self.numJoints, 3), np.float32)
            ev[:, :2] = pred[dat, :, :2]
            ev[:, 2] = dm[ev[:, 1].astype(int), ev[:, 0].astype(int)]
            far = np.abs(ev[:, 2] - gt[dat, 13, 2]) > 150.
            ev[far, 2] = np.asarray(gt[dat, :self.numJoints, 2])[far]
            data.append(self.jointsImgTo3D(ev))
        return data

    def loadBaseline2D(self, filename):
        """The .mat predictions as image-plane (numJoints, 2) arrays (importers.py:1147-1175)."""
        self.numJoints, pred = self._predicted_uv(filename)
        return [np.asarray(p[:, :2], np.float32) for p in pred]

    def loadSequence(self, seqName, Nmax=float('inf'), shuffle=False, rng=None, docom=False, cube=None):
        """<basepath>/<seqName>/joint_data.mat (joint_xyz, joint_uvd of camera 1) and depth_1_NNNNNNN.png
        (importers.py:943-1064); the 14 evaluation joints unless allJoints."""
        import scipy.io
        from util.handdetector import HandDetector
        config = self._config(seqName, cube)
        cache = self._cache_file(seqName, self.hand, self.allJoints, HandDetector.detectionModeToString(docom, self.refineNet is not None),
                                 config['cube'][0])
        cached = self._from_cache(cache, Nmax, shuffle, rng)
        if cached is not None:
            return cached
        if self.hand is not None and self.hand != self.sides[seqName]:
            raise NotImplementedError()
        objdir = '{}/{}/'.format(self.basepath, seqName)
        mat = scipy.io.loadmat('{}/{}/joint_data.mat'.format(self.basepath, seqName))
        joints3D = mat['joint_xyz'][0]
        joints2D = mat['joint_uvd'][0]
        eval_idxs = np.arange(36) if self.allJoints else np.asarray(self.restrictedJointsEval)
        self.numJoints = len(eval_idxs)
        records = self._crop_stream(config, docom, self.sides[seqName])
        entries = [('{0:s}/depth_1_{1:07d}.png'.format(objdir, line + 1), line) for line in range(joints3D.shape[0])]
        for (dptFileName, line), dpt in self._read_ahead(entries):
            if len(records) >= Nmax:
                break
            if dpt is None:
                print("File {} does not exist!".format(dptFileName))
                continue
            gtorig = np.asarray(joints2D[line, eval_idxs, 0:3], np.float32)     # joints in image coordinates
            gt3Dorig = np.asarray(joints3D[line, eval_idxs, 0:3], np.float32)   # normalized joints in 3D coordinates
            if not self._has_content(dpt, 1):
                print("Skipping image {}, no content".format(dptFileName))
                continue
            records.append((dpt, gtorig, gt3Dorig, dptFileName, ''))
        print("Loaded {} samples.".format(len(records)))
        data = records.finish()
        return self._finish(seqName, data, config, cache, shuffle, rng)
