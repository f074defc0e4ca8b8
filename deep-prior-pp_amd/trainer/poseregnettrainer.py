"""
PoseRegNetTrainer / PoseRegNetTrainerParams (API of /root/reference/src/trainer/poseregnettrainer.py:44-264).

setupFunctions/compileFunctions build, instead of Theano functions, two compiled engines over the SAME device-resident
parameters: a training engine (forward with batch statistics, sum-squared-error cost, backward, ADAM) behind
`train_model(index, lr)`, and a deterministic engine behind the validation functions.  `augment_poses` augments the
whole resident macro-batch with the fused HIP kernels and writes `train_data_x` / `train_data_y` in place.
"""
import numpy

from hipdp import engine, ops
from hipdp.augmenter import DeviceAugmenter
from net.poseregnet import PoseRegNet, PoseRegNetParams      # noqa: F401  (re-exported like the reference module)
from trainer.nettrainer import NetTrainer, NetTrainerParams
from trainer.optimizer import Optimizer


class PoseRegNetTrainerParams(NetTrainerParams):
    def __init__(self):
        super(PoseRegNetTrainerParams, self).__init__()


class PoseRegNetTrainer(NetTrainer):
    def __init__(self, poseNet=None, cfgParams=None, rng=None, subfolder='./eval/', numChunks=1, runtime=None, dp=None):
        super(PoseRegNetTrainer, self).__init__(cfgParams, 5, subfolder, numChunks, runtime=runtime, dp=dp)
        self.poseNet = poseNet
        if self.dp is not None and poseNet is not None:
            poseNet.dp = self.dp           # test-time computeOutput shards its batches over the ranks too (NetBase.computeOutput)
        self.rng = rng if rng is not None else self.rng
        if not isinstance(cfgParams, PoseRegNetTrainerParams):
            raise ValueError("cfgParams must be an instance of PoseRegNetTrainerParams")
        self.setupFunctions()

    def setupFunctions(self):
        """Cost selection of poseregnettrainer.py:84-107: numJoints == 1 -> mean_n sum_d (out-y)^2 (the PCA-embedding
        target); otherwise mean_n mean_j sum_d; L2 weight decay only for nets without dropout."""
        cfg = self.poseNet.cfgParams
        if cfg.numJoints == 1 and cfg.nDims == 1:
            self.loss_cfg = dict(kind='scalar')           # y is a VECTOR: the (B, 1) output broadcasts against it (:84-85, 92-93)
        elif cfg.numJoints == 1:
            self.loss_cfg = dict(kind='embedding')
        else:
            self.loss_cfg = dict(kind='joints', numJoints=cfg.numJoints, nDims=cfg.nDims)
        self.params = self.poseNet.params
        self.grads = ['d(cost)/d(%s)' % p.name for p in self.params]
        self._augmenter = None

    def compileFunctions(self, compileDebugFcts=False):
        self.setupTrain()
        self.compileDebugFcts = compileDebugFcts
        if compileDebugFcts:
            self.setupDebugFunctions()
        self.setupValidate()

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
            self.rt.copy(te.x_in.buf, self.train_data_x.rows(index * B, B))
            self.rt.copy(te.y_in, self.train_data_y.rows(index * B, B))
            self._train_step(learning_rate)

        def train_model(index, learning_rate):
            launch(index, learning_rate)
            return float(te.cost.get()[0])

        def train_model_async(index, learning_rate):
            """Same step; the cost comes back through a handle (.get()) so that the epoch loop can queue the next minibatch
            before it reads this one's cost (no host round trip between steps)."""
            launch(index, learning_rate)
            return self.rt.read_async(te.cost)

        def test_model_on_train(index):
            return self._eval(self.train_data_x, self.train_data_y, index)[1]

        self.train_model_async = train_model_async
        self.train_model = train_model
        self.test_model_on_train = test_model_on_train

    def _compile_engines(self, wd):
        """The two compiled engines over the same device-resident parameters.  Data parallel (self.dp): the train engine's step
        all-reduces the flat gradient buffer (early FC1 bucket + the rest) between backward and the replicated ADAM, and its cost
        buffer -- this rank's share of the global minibatch's cost -- is summed over the ranks right after the step."""
        self.train_engine = engine.CompiledNet(self.poseNet, train=True, runtime=self.rt, loss=self.loss_cfg, weight_decay=wd, dp=self.dp,
                                               optimizer=getattr(getattr(self, 'optimizer', None), 'rule', None))
        self.eval_engine = engine.CompiledNet(self.poseNet, train=False, runtime=self.rt, loss=self.loss_cfg, weight_decay=wd)
        self._allreduce = None                         # (kept for callers that drive the engines by hand: a callable on the flat gradient)
        self._cost_sum = self.dp.allreduce_sum_op(self.train_engine.cost, 'cost_allreduce') if self.dp is not None else None

    def _train_step(self, learning_rate):
        self.train_engine.train_step_device(learning_rate, allreduce=self._allreduce)
        if self._cost_sum is not None:
            self._cost_sum(self.rt.stream)

    def _reduce_eval(self, cost, err):
        """Validation values of equally sized shards: the mean over the ranks is the value of the global batch."""
        if self.dp is None:
            return cost, err
        c, e = self.dp.mean_scalars([cost, err])
        return c, e

    def _eval(self, xs, ys, index):
        B = self.cfgParams.batch_size
        ee = self.eval_engine
        self.rt.copy(ee.x_in.buf, xs.rows(index * B, B))
        self.rt.copy(ee.y_in, ys.rows(index * B, B))
        ee.fwd.run(self.rt)
        ee.lossplan.run(self.rt)
        return self._reduce_eval(float(ee.cost.get()[0]), float(ee.err.get()[0]))

    # ---- validate ---------------------------------------------------------------------------------------
    def setupValidate(self):
        print("compiling validation_cost() ... ")
        self.validation_cost = lambda index: self._eval(self.val_data_x, self.val_data_y, index)[0]
        self.validation_observer.append(self.validation_cost)
        print("compiling validation_error() ... ")
        self.validation_error = lambda index: self._eval(self.val_data_x, self.val_data_y, index)[1]
        self.validation_observer.append(self.validation_error)
        print("compiling validation_error_avg() ... ")
        if hasattr(self, 'val_data_y3D'):
            # errors_avg / errors_max of poseregnettrainer.py:122-126: back-project through the PCA prior on the device
            B = self.cfgParams.batch_size
            E, D = self.pca_dataDB.shape
            self._pca_out = self.rt.alloc((B, D))
            self._err3d = self.rt.alloc(2)
            ee = self.eval_engine
            self._pca_gemm = ops.gemm(self.rt, ee.out.buf, self.pca_data.buf, self._pca_out, B, D, E, 1, 0, E, D, D, bias=self.mean_data.buf,
                                      name='pca_backproject')

            def _avgmax(index):
                self._eval(self.val_data_x, self.val_data_y, index)
                self._pca_gemm(self.rt.stream)
                ops.error_l2(self.rt, self._pca_out, self.val_data_y3D.rows(index * B, B), B * (D // 3), 3, self._err3d)(self.rt.stream)
                e = self._err3d.get()
                if self.dp is not None:                       # mean of the shard means; the max is a max over ranks
                    import torch
                    t = torch.tensor([float(e[0]) / self.dp.world, 0.0], dtype=torch.float64)
                    m = torch.tensor([float(e[1])], dtype=torch.float64)
                    if self.dp.dist.get_backend() == 'nccl':
                        t, m = t.cuda(), m.cuda()
                    self.dp.dist.all_reduce(t, op=self.dp.dist.ReduceOp.SUM)
                    self.dp.dist.all_reduce(m, op=self.dp.dist.ReduceOp.MAX)
                    return float(t[0].item()), float(m.item())
                return float(e[0]), float(e[1])

            self.validation_error_avg = lambda index: _avgmax(index)[0]
            self.validation_error_max = lambda index: _avgmax(index)[1]
            self.validation_observer.append(self.validation_error_avg)
            self.validation_observer.append(self.validation_error_max)
        print("done.")

    def setupDebugFunctions(self):
        B = self.cfgParams.batch_size

        def compute_train_descr(index):
            ee = self.eval_engine
            self.rt.copy(ee.x_in.buf, self.train_data_x.rows(index * B, B))
            ee.fwd.run(self.rt)
            return ee.out.buf.get()

        self.compute_train_descr = compute_train_descr

    # ---- augmentation hook ------------------------------------------------------------------------------
    def augment_poses(self, macro_params, macro_idx, last, tidxs, idxs, new_data):
        """Augment the samples `idxs` of the resident training set into positions `tidxs` of train_data_x / train_data_y
        (poseregnettrainer.py:221-264).  The reference calls this per worker on a slice; here the hook is invoked once
        per macro-batch with the full range, and two kernel launches process all of it."""
        args = macro_params['args']
        n = len(idxs)
        if n == 0:
            return
        if list(tidxs) != list(range(tidxs[0], tidxs[0] + n)) or list(idxs) != list(range(idxs[0], idxs[0] + n)):
            raise NotImplementedError("augment_poses expects contiguous index ranges")
        proj = args.get('proj')
        # un-augmented rows: resident as a whole next to the augmented window (the reference's *DB arrays), or -- for a paged
        # training set -- the macro-batch NetTrainer.loadMacroBatch just uploaded
        src_x = self.source_rows('train_data_x', idxs, macro_idx, last)
        if proj is not None or hasattr(self, 'train_gt3DcropDB'):
            gt = self.source_rows('train_gt3Dcrop', idxs, macro_idx, last)
            scale_labels = None
        else:
            # no PCA prior: the labels ARE the normalised joints; the mm-space joints the augmentation needs are
            # train_data_y * cube_z / 2 (poseregnettrainer.py:228-240), formed on the device from the un-augmented labels
            scale_labels = self.source_rows('train_data_y', idxs, macro_idx, last)
            gt = None
        com, cube, M = (self.source_rows(v, idxs, macro_idx, last) for v in ('train_data_com', 'train_data_cube', 'train_data_M'))
        dsz = self.train_data_xDB.shape[-1]
        J = (self.train_gt3DcropDB.shape[1] if gt is not None else int(numpy.prod(self.train_data_yDB.shape[1:])) // 3)
        # ONE augmenter (its seed drawn from self.rng once, its draw counter on the device) for the whole run; the launch list is
        # cached per source buffer: a paged training set alternates between two staging sets (nettrainer._src / _src_next)
        akey = (n, J, dsz, id(proj), tuple(args['aug_modes']), args.get('binarizeImage') is True)
        if self._augmenter is None or self._augmenter[0] != akey:
            aug = DeviceAugmenter(self.rt, args['di'], args['aug_modes'], n, J, dsz=dsz, proj=proj,
                                  sigma_com=args.get('sigma_com'), sigma_sc=args.get('sigma_sc'), rot_range=args.get('rot_range'),
                                  seed=int(self.rng.randint(1 << 30)), normZeroOne=bool(args.get('normZeroOne')),
                                  binarize=args.get('binarizeImage') is True)
            self._augmenter = (akey, aug, {})
        aug, cache = self._augmenter[1], self._augmenter[2]
        key = (src_x.ptr, tidxs[0], n)
        if key not in cache:
            launches = []
            if gt is None:
                gt = self.rt.alloc((n, J, 3), zero=False)
                launches.append(ops.rowscale(self.rt, scale_labels.reshape(n, J * 3), cube.reshape(n, 3), 2, 0.5, gt.reshape(n, J * 3), n, J * 3))
            launches += aug.build(src_x.reshape(n, dsz, dsz), com, cube, M.reshape(n, 9), gt.reshape(n, J, 3),
                                  self.train_data_x.rows(tidxs[0], n).reshape(n, dsz, dsz), self.train_data_y.rows(tidxs[0], n),
                                  dp_layout=self._dp_layout())
            cache[key] = launches
        for op in cache[key]:
            op(self.rt.stream)
