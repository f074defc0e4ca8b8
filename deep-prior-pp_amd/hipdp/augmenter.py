"""
Device-side crop augmentation for a whole (macro-)batch: the fused kernel of csrc/augment.hip (dpp_augment) behind one object.
This is what replaces the 8 worker processes + shared-memory macro-batch protocol of the reference
(/root/reference/src/trainer/nettrainer.py:601-628, 666-689) and the per-sample loop of
PoseRegNetTrainer.augment_poses (/root/reference/src/trainer/poseregnettrainer.py:221-264).
"""
import numpy as np

from . import ops

MODE_CODE = {'none': 0, 'com': 1, 'rot': 2, 'sc': 3}


def camera_tuple(importer):
    """(fx, fy, ux, uy, flip_y) of an importer object (data.importers.*) -- flip_y for the NYU / MSRA conventions."""
    flip = bool(getattr(importer, 'flip_y', importer.__class__.__name__ in ('NYUImporter', 'MSRA15Importer')))
    return (float(importer.fx), float(importer.fy), float(importer.ux), float(importer.uy), flip)


class DeviceAugmenter(object):
    def __init__(self, rt, importer, aug_modes, n, J, dsz=128, proj=None, sigma_com=None, sigma_sc=None, rot_range=None, seed=0,
                 normZeroOne=False, binarize=False):
        for m in aug_modes:
            if m not in MODE_CODE:
                raise NotImplementedError("augmentation mode %r" % (m,))
        self.rt, self.n, self.J, self.dsz = rt, int(n), int(J), int(dsz)
        self.cam = camera_tuple(importer)
        self.table = rt.upload(np.array([MODE_CODE[m] for m in aug_modes], np.int32))
        self.n_modes = len(aug_modes)
        self.sigma_com = 5. if sigma_com is None else sigma_com          # nettrainer.py:939-946
        self.sigma_sc = 0.02 if sigma_sc is None else sigma_sc
        self.rot_range = 180. if rot_range is None else rot_range
        self.seed = int(seed)
        self.normZeroOne = bool(normZeroOne)
        self.binarize = bool(binarize)            # augment_poses' binarizeImage (poseregnettrainer.py:255-257)
        self.counter = rt.alloc(1, np.int64)
        self.ticket = rt.alloc(1, np.int32)
        self.sample0, self.global_batch = 0, self.n            # set_shard(): data-parallel ranks key draws by the global sample index
        self.pm = self.pc = None
        self.E = 0
        if proj is not None:                                             # sklearn PCA: (x - mean_) . components_^T
            self.pm = rt.upload(np.asarray(proj.mean_, np.float32))
            self.pc = rt.upload(np.asarray(proj.components_, np.float32))
            self.E = int(proj.components_.shape[0])
        self.out_dim = self.E if proj is not None else self.J * 3

    def set_shard(self, sample0, global_batch):
        """Data parallelism: this augmenter handles samples [sample0, sample0 + n) of a global (macro-)batch of `global_batch`
        samples; device draws are keyed by the global index, so they do not depend on the number of ranks."""
        self.sample0, self.global_batch = int(sample0), int(global_batch)

    def build(self, img, com3d, cube, Mcrop, gt3d, out_x, out_y, explicit=None, dp_layout=None):
        """The launch list (ONE fused launch) augmenting `n` crops from the *DB buffers into out_x / out_y.  explicit =
        dict(mode, off, rot, sc) of device buffers pins the draws (parity tests); otherwise they come from the device generator,
        whose counter the launch itself advances.
        dp_layout = (G, rank, B): the n local samples are this rank's slices of n / B global minibatches of G * B samples (the
        trainer's sharding, hipdp.parallel.DataParallel.shard): local sample k * B + j is global sample k * G * B + rank * B + j.
        One launch per local minibatch then, each keyed by its global position, all on the same draw counter (the last one advances
        it) -- what a sample draws is what it would draw in a single-process run over the global macro-batch."""
        rt = self.rt
        if dp_layout is None:
            chunks = [(0, self.n, self.sample0, self.global_batch, True)]
        else:
            G, rank, B = (int(v) for v in dp_layout)
            if self.n % B:
                raise ValueError("the local macro-batch (%d samples) is not a whole number of minibatches of %d" % (self.n, B))
            nk = self.n // B
            chunks = [(k * B, B, k * G * B + rank * B, self.n * G, k == nk - 1) for k in range(nk)]
        E = self.out_dim
        D = self.dsz * self.dsz

        def rows(buf, start, count, per):
            return buf.view(start * per, (count * per,))

        launches = []
        for (start, count, sample0, gbatch, last) in chunks:
            kw = dict(mode_table=self.table, n_modes=self.n_modes, seed=self.seed, counter=0, counter_dev=self.counter,
                      ticket=self.ticket if last else None, sample0=sample0, global_batch=gbatch)
            if explicit is not None:
                if dp_layout is not None:
                    raise NotImplementedError("explicit draws with a data-parallel layout")
                kw = dict(mode=explicit['mode'], off=explicit['off'], rot=explicit['rot'], sc=explicit['sc'])
            launches.append(ops.augment(rt, rows(img, start, count, D), rows(com3d, start, count, 3), rows(cube, start, count, 3),
                                        rows(Mcrop, start, count, 9), rows(gt3d, start, count, self.J * 3), count, self.J, self.dsz, self.cam,
                                        rows(out_x, start, count, D), rows(out_y, start, count, E),
                                        sigma_com=self.sigma_com, sigma_sc=self.sigma_sc, rot_range=self.rot_range, pca_mean=self.pm,
                                        pca_comp=self.pc, E=self.E, norm_zero_one=self.normZeroOne, binarize=self.binarize, **kw))
        return launches
