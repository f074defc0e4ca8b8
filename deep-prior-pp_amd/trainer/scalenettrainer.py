"""
ScaleNetTrainer / ScaleNetTrainerParams (API of /root/reference/src/trainer/scalenettrainer.py:44-251): trains the multi-scale
CoM-refinement regressor of main_nyu_com_refine.py.  The three network inputs are the crop (`train_data_x`) and its 1/2 and
1/4 centre crops (`train_data_x1`, `train_data_x2`, added with addManagedData like in the reference); the label is the
normalised 3-D offset of the crop joint (numJoints = 1, nDims = 3).  `augment_poses` augments the resident macro-batch on
the device and re-derives the two centre crops from it (scalenettrainer.py:239-251).
"""
from hipdp import ops
from hipdp.augmenter import DeviceAugmenter
from net.scalenet import ScaleNet, ScaleNetParams      # noqa: F401  (re-exported like the reference module)
from trainer.nettrainer import NetTrainerParams
from trainer.optimizer import Optimizer
from trainer.poseregnettrainer import PoseRegNetTrainer


class ScaleNetTrainerParams(NetTrainerParams):
    def __init__(self):
        super(ScaleNetTrainerParams, self).__init__()


class ScaleNetTrainer(PoseRegNetTrainer):
    def __init__(self, poseNet=None, cfgParams=None, rng=None, subfolder='./eval/', numChunks=1, runtime=None, dp=None):
        from trainer.nettrainer import NetTrainer
        NetTrainer.__init__(self, cfgParams, 8, subfolder, numChunks, runtime=runtime, dp=dp)         # memory factor 8, :61
        self.poseNet = poseNet
        if self.dp is not None and poseNet is not None:
            poseNet.dp = self.dp           # test-time computeOutput shards its batches over the ranks too (NetBase.computeOutput)
        self.rng = rng if rng is not None else self.rng
        if not isinstance(cfgParams, ScaleNetTrainerParams):
            raise ValueError("cfgParams must be an instance of ScaleNetTrainerParams")
        self.setupFunctions()

    def _inputs(self, prefix, index):
        B = self.cfgParams.batch_size
        names = [prefix + '_data_x'] + [prefix + '_data_x' + str(i) for i in range(1, self.poseNet.cfgParams.numInputs)]
        return [getattr(self, n).rows(index * B, B) for n in names]

    # ---- train ------------------------------------------------------------------------------------------
    def setupTrain(self):
        opt = self.optimizer = Optimizer(self.grads, self.params)
        self.updates = opt.ADAM(self.cfgParams.learning_rate)
        print("compiling train_model() ... ")
        wd = self.cfgParams.weightreg_factor if not self.poseNet.hasDropout() else 0.0
        self._compile_engines(wd)
        print("done.")
        B = self.cfgParams.batch_size
        te = self.train_engine

        def launch(index, learning_rate):
            for t, src in zip(te.x_ins, self._inputs('train', index)):       # givens_train of scalenettrainer.py:148-151
                self.rt.copy(t.buf, src)
            self.rt.copy(te.y_in, self.train_data_y.rows(index * B, B))
            self._train_step(learning_rate)

        def train_model(index, learning_rate):
            launch(index, learning_rate)
            return float(te.cost.get()[0])

        def train_model_async(index, learning_rate):
            launch(index, learning_rate)
            return self.rt.read_async(te.cost)

        self.train_model = train_model
        self.train_model_async = train_model_async
        self.test_model_on_train = lambda index: self._eval_multi('train', self.train_data_y, index)[1]

    def _eval_multi(self, prefix, ys, index):
        B = self.cfgParams.batch_size
        ee = self.eval_engine
        for t, src in zip(ee.x_ins, self._inputs(prefix, index)):
            self.rt.copy(t.buf, src)
        self.rt.copy(ee.y_in, ys.rows(index * B, B))
        ee.fwd.run(self.rt)
        ee.lossplan.run(self.rt)
        return self._reduce_eval(float(ee.cost.get()[0]), float(ee.err.get()[0]))

    # ---- validate ---------------------------------------------------------------------------------------
    def setupValidate(self):
        print("compiling validation_cost() ... ")
        self.validation_cost = lambda index: self._eval_multi('val', self.val_data_y, index)[0]
        print("done.")
        self.validation_observer.append(self.validation_cost)
        print("compiling validation_error() ... ")
        self.validation_error = lambda index: self._eval_multi('val', self.val_data_y, index)[1]
        print("done.")
        self.validation_observer.append(self.validation_error)

    def setupDebugFunctions(self):
        def compute_train_descr(index):
            ee = self.eval_engine
            for t, src in zip(ee.x_ins, self._inputs('train', index)):
                self.rt.copy(t.buf, src)
            ee.fwd.run(self.rt)
            return ee.out.buf.get()

        self.compute_train_descr = compute_train_descr

    # ---- augmentation hook ------------------------------------------------------------------------------
    def augment_poses(self, macro_params, macro_idx, last, tidxs, idxs, new_data):
        """scalenettrainer.py:211-251: augment crop + offset label (one "joint": the label times cube_z/2), then cut the two
        centre crops out of the augmented crop."""
        args = macro_params['args']
        n = len(idxs)
        if n == 0:
            return
        if list(tidxs) != list(range(tidxs[0], tidxs[0] + n)) or list(idxs) != list(range(idxs[0], idxs[0] + n)):
            raise NotImplementedError("augment_poses expects contiguous index ranges")
        H, W = self.train_data_xDB.shape[-2:]
        # un-augmented rows (resident as a whole, or the macro-batch loadMacroBatch just uploaded); the offset label in mm
        # (label * cube_z / 2, one "joint") is formed on the device
        src_x = self.source_rows('train_data_x', idxs, macro_idx, last)
        src_y = self.source_rows('train_data_y', idxs, macro_idx, last)
        com, cube, M = (self.source_rows(v, idxs, macro_idx, last) for v in ('train_data_com', 'train_data_cube', 'train_data_M'))
        akey = (n, H, tuple(args['aug_modes']))          # one augmenter for the run, launch lists per source buffer (see PoseRegNetTrainer)
        if self._augmenter is None or self._augmenter[0] != akey:
            aug = DeviceAugmenter(self.rt, args['di'], args['aug_modes'], n, 1, dsz=H, proj=None, sigma_com=args.get('sigma_com'),
                                  sigma_sc=args.get('sigma_sc'), rot_range=args.get('rot_range'), seed=int(self.rng.randint(1 << 30)),
                                  normZeroOne=bool(args.get('normZeroOne')))
            self._augmenter = (akey, aug, {})
        aug, cache = self._augmenter[1], self._augmenter[2]
        key = (src_x.ptr, tidxs[0], n)
        if key not in cache:
            x_out = self.train_data_x.rows(tidxs[0], n).reshape(n, H, W)
            gt = self.rt.alloc((n, 1, 3), zero=False)
            launches = [ops.rowscale(self.rt, src_y.reshape(n, 3), cube.reshape(n, 3), 2, 0.5, gt.reshape(n, 3), n, 3)]
            launches += aug.build(src_x.reshape(n, H, W), com, cube, M.reshape(n, 9), gt, x_out, self.train_data_y.rows(tidxs[0], n),
                                  dp_layout=self._dp_layout())
            for k, name in ((2, 'train_data_x1'), (4, 'train_data_x2')):
                launches.append(ops.crop_center(self.rt, x_out, n, H, W, getattr(self, name).rows(tidxs[0], n), H // k, W // k))
            cache[key] = launches
        for op in cache[key]:
            op(self.rt.stream)
