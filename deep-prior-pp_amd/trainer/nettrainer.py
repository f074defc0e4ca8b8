"""
NetTrainer / NetTrainerParams (API of /root/reference/src/trainer/nettrainer.py:47-997): data management, the
epoch / minibatch loop with validation, early stopping and snapshots, and the crop augmentation hook.

What differs from the reference is WHERE things run, not what they compute:
  * the training set, its labels and the augmentation side data (`train_data_cube/com/M`, `train_gt3Dcrop`) live on
    the MI355X (288 GB HBM holds NYU's 4.8 GB outright; the reference's host-RAM <-> device "macro-batch" paging is
    kept as arithmetic -- same helper methods, same padding with seeded random samples -- but a single resident
    macro-batch is required);
  * the 8 augmentation worker processes + shared-memory hand-off (nettrainer.py:601-628, 666-689) are replaced by the
    two fused HIP kernels of hipdp.augmenter, run over the whole macro-batch whenever the reference would swap in a
    freshly augmented one (nettrainer.py:528-558);
  * `train_model` / validation functions are launch plans compiled by hipdp.engine instead of theano.function.
"""
import time

import numpy

from hipdp.runtime import default_runtime


def _nanmean(values):
    """numpy.nanmean, NaN (without NumPy's empty-slice warning) for a validation set with no minibatch -- what the reference's
    observers return then (nettrainer.py:760-765 on an empty list)."""
    values = numpy.asarray(values, dtype=numpy.float64)
    return float('nan') if values.size == 0 or numpy.isnan(values).all() else float(numpy.nanmean(values))


class NetTrainerParams(object):
    def __init__(self):
        self.batch_size = 128
        self.momentum = 0.9                       # present in the reference, never used by its optimiser
        self.learning_rate = 0.01
        self.weightreg_factor = 0.001
        self.use_early_stopping = True
        # learning rate as a function of the epoch (nettrainer.py:54)
        self.lr_of_ep = lambda ep: numpy.float32(self.learning_rate / 10.) if ep <= 1 else \
            numpy.float32(self.learning_rate / 3.) if 1 < ep <= 2 else numpy.float32(self.learning_rate * numpy.exp(-0.04 * ep))
        self.snapshot_last = 5
        self.snapshot_freq = None
        self.para_augment = False                 # accepted for compatibility: augmentation always runs on the device
        self.para_num_proc = 8
        self.augment_fun_params = {'fun': None, 'args': {}}
        self.para_load = False
        self.load_fun_params = {'fun': None, 'args': {}}
        self.force_macrobatch_reload = False
        self.pad_random = True
        self.validation_frequency = 1000
        self.pre_epoch_fn = None
        self.post_epoch_fn = None
        self.pre_minibatch_fn = None
        self.post_minibatch_fn = None


class DeviceData(object):
    """Stand-in for a Theano shared variable holding a data array on the device."""

    @staticmethod
    def _floatX(value):
        """Floating-point data becomes float32 like every Theano shared variable of the reference (floatX): the kernels read
        and write raw float32 pointers, so a float64 array (sklearn's pca.mean_ / components_ / transform output, which the
        mains pass on uncast) must not reach the device as it is."""
        value = numpy.asarray(value)
        if value.dtype.kind == 'f' and value.dtype != numpy.float32:
            value = value.astype(numpy.float32)
        return numpy.ascontiguousarray(value)

    def __init__(self, rt, value, name):
        self.name = name
        self.rt = rt
        value = self._floatX(value)
        self.buf = rt.upload(value)
        self.shape = tuple(value.shape)
        self.dtype = value.dtype

    def set_value(self, value, borrow=False):
        value = self._floatX(value)
        if tuple(value.shape) != self.shape or value.dtype != self.dtype:
            self.buf = self.rt.upload(value)
            self.shape, self.dtype = tuple(value.shape), value.dtype
        else:
            self.buf.set(value)

    def get_value(self, borrow=False):
        return self.buf.get().reshape(self.shape)

    def rows(self, start, count):
        """Buffer view of `count` leading-dimension rows starting at `start` (no copy)."""
        per = int(numpy.prod(self.shape[1:])) if len(self.shape) > 1 else 1
        return self.buf.view(start * per, (count,) + tuple(self.shape[1:]))


class NetTrainer(object):
    SYNC_BATCH_FINISHED = 'batch_finished'
    SYNC_LOAD_FINISHED = 'load_finished'

    def __init__(self, cfgParams, memory_factor, subfolder='./eval/', numChunks=1, runtime=None, dp=None):
        """dp: None, a hipdp.parallel.DataParallel, or 'env' (torch.distributed from RANK / WORLD_SIZE / MASTER_*; one process per
        GPU, RCCL).  With G ranks `cfgParams.batch_size` is the PER-GPU batch and the global minibatch has G x that many samples:
        every rank keeps its contiguous slice of every global minibatch of the arrays handed to setData / addManagedData /
        addStaticData('val_*'), augments it on its own GPU with draws keyed by the global sample index,
        the gradients are all-reduced inside `train_model`, validation values are averaged over the ranks and rank 0 writes the
        snapshots.  The reference is single-device (nettrainer.py:778-907 is the loop this wraps)."""
        self.subfolder = subfolder
        self.cfgParams = cfgParams
        self.rng = numpy.random.RandomState(23455)
        if not isinstance(cfgParams, NetTrainerParams):
            raise ValueError("cfgParams must be an instance of NetTrainerParams")
        world = 1
        if dp == 'env':
            from hipdp import parallel
            _, world = parallel.init_from_env()        # selects the rank's GPU: before any runtime is created
        self.rt = runtime or default_runtime()
        if dp == 'env':
            parallel.check_runtime_device(self.rt)
            dp = parallel.DataParallel(self.rt) if world > 1 else None
        self.dp = dp
        self.memorySize = self._free_device_mb() / float(memory_factor)    # MB, nettrainer.py:100-112
        if cfgParams.para_load is True and numChunks == 1:
            raise ValueError("para_load is True but numChunks == 1, so we do not need para_load!")
        self._src = {}                 # var -> device buffer with the UN-augmented rows of the current macro-batch window
        self._src_next = {}            # the other staging set: the next macro-batch is uploaded into it while this one trains
        self._prefetched = None        # (macro_idx, handles) of the upload in flight
        self._load_thread = None
        self.currentMacroBatch = -1
        self.currentChunk = -1
        self.numChunks = numChunks
        self.trainSize = 0
        self.sampleSize = 0
        self.numTrainSamplesMB = 0
        self.numTrainSamples = 0
        self.numValSamples = 0
        self.epoch = 0
        self.managedVar = []
        self.trainingVar = []
        self.validation_observer = []

    def _free_device_mb(self):
        try:
            import torch
            if torch.cuda.is_available() and not getattr(self.rt, 'is_emulator', False):
                free, _ = torch.cuda.mem_get_info()
                return free / 1024. ** 2
        except Exception:       # noqa: BLE001
            pass
        import psutil
        return psutil.virtual_memory().available / 1024. ** 2

    # ---- data registration ---------------------------------------------------------------------------------
    def _publish(self, key, host):
        setattr(self, key + 'DB', host)
        if hasattr(self, key):
            print("Reusing shared variables!")
            getattr(self, key).set_value(host, borrow=True)
        else:
            setattr(self, key, DeviceData(self.rt, host, key))

    def addData(self, data):
        if not isinstance(data, dict):
            raise ValueError("Error: expected dictionary for data!")
        for key in data:
            self._publish(key, self.alignData(data[key]))

    def addStaticData(self, data):
        if not isinstance(data, dict):
            raise ValueError("Error: expected dictionary for data!")
        for key in data:
            val = numpy.asarray(data[key])
            if self.dp is not None and key.startswith('val_') and val.ndim >= 1 and val.shape[0] == getattr(self, '_global_val_n', -1):
                val = self.dp.shard(val, self.cfgParams.batch_size)        # per-sample validation arrays follow val_data_x
            self._publish(key, val)

    def addManagedData(self, data):
        if not isinstance(data, dict):
            raise ValueError("Error: expected dictionary for data!")
        for key in data:
            if data[key].shape[0] != (self.numTrainSamplesMB if self.dp is None else self._global_train_n):
                raise ValueError("Number of samples must be the same as number of labels.")
            self._publish_training(key, self._shard_train(numpy.asarray(data[key])))
            self.trainingVar.append(key)

    def _publish_training(self, key, data):
        """A per-training-sample array.  It fits one macro-batch (always, on 288 GB, for the datasets of the mains): aligned and
        resident on the device as a whole.  Otherwise the reference's paging layout (nettrainer.py:259-276, 193-204): host
        arrays keyDB (the full macro-batches) and keyDBlast (the last one, padded), and a device window of ONE macro-batch that
        loadMacroBatch refills."""
        nmb = self.getNumMacroBatches()
        if nmb == 1:
            self._publish(key, self.alignData(data))
            return
        spm = self.getNumSamplesPerMacroBatch()
        n_full = (nmb - 1) * spm
        setattr(self, key + 'DB', self._pinned(data[0:n_full]))
        setattr(self, key + 'DBlast', self._pinned(self.alignData(data[n_full:], fillData=data)))
        if key not in self.managedVar:
            self.managedVar.append(key)
        first = getattr(self, key + 'DB')[:spm]
        if hasattr(self, key):
            print("Reusing shared variables!")
            getattr(self, key).set_value(first, borrow=True)
        else:
            setattr(self, key, DeviceData(self.rt, first, key))

    def _pinned(self, arr):
        """Page-locked host copy (asynchronous uploads need it); a plain array on runtimes without pinned memory."""
        arr = DeviceData._floatX(arr)
        pin = getattr(self.rt, 'pinned_like', None)
        if pin is None:
            return arr
        out = pin(arr)
        out[...] = arr
        return out

    def _shard_train(self, data):
        """Data parallel: this rank's slice of every global minibatch, the global array first padded to whole global minibatches by
        alignData's rule (rows drawn by RandomState(n), or the last row repeated without pad_random); single process: the array as it is."""
        if self.dp is None:
            return data
        if self.cfgParams.pad_random:
            return self.dp.shard(data, self.cfgParams.batch_size, pad_rng_seed=data.shape[0])
        return self.dp.shard(data, self.cfgParams.batch_size, pad='last')

    def _dp_layout(self):
        """(G, rank, B) for the augmenter: how this rank's samples sit in the global macro-batch (None: single process)."""
        return None if self.dp is None else (self.dp.world, self.dp.rank, self.cfgParams.batch_size)

    def setData(self, train_data, train_y, val_data, val_y, max_train_size=0):
        if (train_data.shape[0] != train_y.shape[0]) or (val_data.shape[0] != val_y.shape[0]):
            raise ValueError("Number of samples must be the same as number of labels.")
        if self.dp is not None:
            self._global_train_n, self._global_val_n = train_data.shape[0], val_data.shape[0]
            max_train_size = max_train_size / float(self.dp.world)
            train_data, train_y = self._shard_train(train_data), self._shard_train(train_y)
            val_data, val_y = (self.dp.shard(v, self.cfgParams.batch_size, what='the validation set') for v in (val_data, val_y))
        self.trainSize = max(train_data.nbytes, train_y.nbytes, max_train_size) / 1024. / 1024.
        self.numTrainSamplesMB = train_data.shape[0]
        self.numTrainSamples = self.numTrainSamplesMB
        self.numValSamples = val_data.shape[0]
        self.sampleSize = self.trainSize / self.numTrainSamplesMB
        assert self.memorySize > self.sampleSize * self.cfgParams.batch_size, \
            "{} > {}".format(self.memorySize, self.sampleSize * self.cfgParams.batch_size)
        if self.getNumMacroBatches() == 1:
            # shrink the macro batch to the smallest possible (nettrainer.py:255-257)
            self.memorySize = self.sampleSize * numpy.ceil(self.numTrainSamplesMB / float(self.cfgParams.batch_size)) * \
                self.cfgParams.batch_size
        self._publish_training('train_data_x', train_data)
        self._publish_training('train_data_y', train_y)
        self.trainingVar.append('train_data_x')
        self.trainingVar.append('train_data_y')
        self._publish('val_data_x', val_data)
        self._publish('val_data_y', val_y)
        print("Train size: {}MB, Memory available: {}MB, sample size: {}MB, aligned memory: {}MB".format(
            self.trainSize, self.memorySize, self.sampleSize, self.getGPUMemAligned()))
        print("{} train samples, {} val samples, batch size {}".format(
            train_data.shape[0], val_data.shape[0], self.cfgParams.batch_size))
        print("{} macro batches, {} mini batches per macro, {} full mini batches total".format(
            self.getNumMacroBatches(), self.getNumMiniBatchesPerMacroBatch(), self.getNumMiniBatches()))
        print("{} data chunks, {} train samples total".format(self.numChunks, self.numTrainSamples))

    def replaceValData(self, val_data, val_y):
        self.val_data_x.set_value(val_data, borrow=True)
        self.val_data_y.set_value(val_y, borrow=True)

    def alignData(self, data, alignSize=None, out=None, fillData=None):
        """Pad to a whole number of minibatches with training samples drawn by RandomState(data.shape[0]) -- the same
        seed for every array, so data and labels pad consistently (nettrainer.py:365-413)."""
        if out is not None:
            raise NotImplementedError()
        if alignSize is None:
            alignSize = self.getNumSamplesPerMacroBatch()
        if alignSize < data.shape[0]:
            print("WARNING: aligned size < data size ({}<{})".format(alignSize, data.shape[0]))
        topad = 0 if data.shape[0] == alignSize else alignSize - data.shape[0] % alignSize
        padded = numpy.pad(data, [(0, topad)] + [(0, 0)] * (data.ndim - 1), mode='constant', constant_values=0)
        if fillData is None:
            fillData = data
        if (data.shape[0] % alignSize) != 0:
            n_fill = alignSize - (data.shape[0] % alignSize)
            if self.cfgParams.pad_random:
                rng = numpy.random.RandomState(data.shape[0])
                for i in range(0, n_fill):
                    padded[data.shape[0] + i] = fillData[rng.randint(0, fillData.shape[0])]
            else:
                for i in range(0, n_fill):
                    padded[data.shape[0] + i] = padded[data.shape[0] - 1]
        return padded

    # ---- size arithmetic (nettrainer.py:415-487) ---------------------------------------------------------
    def getSizeMiniBatch(self):
        return self.cfgParams.batch_size * self.sampleSize

    def getSizeMacroBatch(self):
        return self.getNumMacroBatches() * self.getSizeMiniBatch()

    def getNumFullMiniBatches(self):
        return self.getNumMiniBatches()

    def getNumMiniBatches(self):
        return int(numpy.ceil(self.numTrainSamples / float(self.cfgParams.batch_size)))

    def getNumMacroBatches(self):
        return int(numpy.ceil(self.trainSize / float(self.getGPUMemAligned())))

    def getNumMiniBatchesPerMacroBatch(self):
        return int(self.getGPUMemAligned() / self.sampleSize / self.cfgParams.batch_size)

    def getNumSamplesPerMacroBatch(self):
        return int(self.getNumMiniBatchesPerMacroBatch() * self.cfgParams.batch_size)

    def getNumMiniBatchesPerChunk(self):
        return int(self.getNumMiniBatchesPerMacroBatch() * self.getNumMacroBatches())

    def getNumSamplesPerChunk(self):
        return self.getNumMiniBatchesPerChunk() * self.cfgParams.batch_size

    def getGPUMemAligned(self):
        return self.sampleSize * self.cfgParams.batch_size * int(self.memorySize / float(self.sampleSize * self.cfgParams.batch_size))

    def isLastMacroBatch(self, macro_idx):
        return macro_idx >= self.getNumMacroBatches() - 1

    def _macro_range(self, mbi, use_all_last=True):
        """(first index, one past the last index, is-last) of macro-batch `mbi` in the indexing the augmentation hooks use: the
        full macro-batches index keyDB, the last one indexes keyDBlast from 0 -- all of it, or only the minibatches that hold real
        samples (nettrainer.py:726-739)."""
        spm = self.getNumSamplesPerMacroBatch()
        if not self.isLastMacroBatch(mbi):
            return mbi * spm, min((mbi + 1) * spm, self.train_data_xDB.shape[0]), False
        if use_all_last is True:
            return 0, spm, True
        B = self.cfgParams.batch_size
        real_minibatches = int(numpy.ceil(self.numTrainSamplesMB / float(B))) - self.getNumMiniBatchesPerMacroBatch() * (self.getNumMacroBatches() - 1)
        return 0, B * real_minibatches, True

    def chunksForMP(self, mbi, use_all_last=True):
        """(last, target index lists, source index lists) of macro-batch `mbi`, dealt into para_num_proc slices the way the reference
        hands them to its worker processes (nettrainer.py:726-744).  loadMacroBatch flattens the slices again: one kernel launch
        augments the whole range."""
        lo, hi, last = self._macro_range(mbi, use_all_last)
        per = int(numpy.ceil((hi - lo) / float(self.cfgParams.para_num_proc)))
        cuts = list(range(0, hi - lo, per)) if per > 0 else []
        tidxs = [list(range(c, min(c + per, hi - lo))) for c in cuts]
        return last, tidxs, [[lo + t for t in sl] for sl in tidxs]

    # ---- macro-batch handling ------------------------------------------------------------------------------
    def loadMiniBatch(self, mini_idx):
        macro_idx = int((mini_idx % self.getNumMiniBatchesPerChunk()) / self.getNumMiniBatchesPerMacroBatch())
        self.loadMacroBatch(macro_idx, mini_idx)
        return mini_idx % self.getNumMiniBatchesPerMacroBatch()

    def loadMacroBatch(self, macro_idx, mini_idx):
        """Make macro-batch `macro_idx` the one in the device window (nettrainer.py:500-599): on first use, or -- with
        force_macrobatch_reload and a single macro-batch -- just before the last minibatch of an epoch, which re-augments the
        resident data.  Paged training sets (more than one macro-batch) are uploaded from the pinned host arrays, the NEXT
        macro-batch already travelling over PCIe on a copy stream while this one trains; the augmentation hook then runs on the
        device from the uploaded rows into the window."""
        force_reload = (((mini_idx % self.getNumMiniBatchesPerChunk()) == self.getNumMiniBatchesPerMacroBatch() - 1) and
                        self.cfgParams.force_macrobatch_reload is True and (self.getNumMacroBatches() == 1))
        if macro_idx != self.currentMacroBatch or force_reload is True:
            fun = self.cfgParams.augment_fun_params['fun']
            nmb = self.getNumMacroBatches()
            if nmb > 1:
                self._page_in(macro_idx, augmenting=fun is not None)
            if fun is not None:
                last, tidx, idxs = self.chunksForMP(macro_idx)
                print("Loading macro batch {}, last {}, start idx {}, end idx {}".format(macro_idx, last, idxs[0][0], idxs[-1][-1]))
                getattr(self, fun)(self.cfgParams.augment_fun_params, macro_idx, last,
                                   [itm for sl in tidx for itm in sl], [itm for sl in idxs for itm in sl], None)
            self.currentMacroBatch = macro_idx
            self._para_swap(macro_idx)
            if nmb > 1:
                self._prefetch((macro_idx + 1) % nmb, augmenting=fun is not None)

    def _macro_rows(self, var, macro_idx):
        """Host rows of macro-batch `macro_idx` of a training array."""
        nmb = self.getNumMacroBatches()
        if nmb == 1:
            return getattr(self, var + 'DB')
        if self.isLastMacroBatch(macro_idx):
            return getattr(self, var + 'DBlast')
        spm = self.getNumSamplesPerMacroBatch()
        return getattr(self, var + 'DB')[macro_idx * spm:(macro_idx + 1) * spm]

    def _stage_target(self, var, augmenting, which):
        """Where the raw rows of `var` go: straight into the device window, or -- for the arrays the augmentation REWRITES
        (train_data_x / train_data_y) -- into a staging buffer the augmentation kernel reads."""
        rewritten = augmenting and var in ('train_data_x', 'train_data_y')
        store = self._src if which == 0 else self._src_next
        if var not in store:
            shape = (self.getNumSamplesPerMacroBatch(),) + tuple(getattr(self, var + 'DBlast').shape[1:])
            store[var] = DeviceData(self.rt, numpy.zeros(shape, getattr(self, var + 'DBlast').dtype), var + '_stage%d' % which)
        return store[var], rewritten

    def _prefetch(self, macro_idx, augmenting):
        up = getattr(self.rt, 'upload_async', None)
        if up is None:
            return                                           # a runtime without asynchronous copies pages in synchronously
        handles = []
        for var in self.trainingVar:
            tgt, _ = self._stage_target(var, augmenting, 1)
            handles.append(up(tgt.buf, self._macro_rows(var, macro_idx)))
        self._prefetched = (macro_idx, handles)

    def _page_in(self, macro_idx, augmenting):
        if self._prefetched is not None and self._prefetched[0] == macro_idx:
            for h in self._prefetched[1]:
                h.wait()                                     # the main stream waits for the copy stream's event
            self._last_paged_in = self._prefetched[1]        # (_para_swap makes the HOST wait for them before it rewrites their source)
            self._src, self._src_next = self._src_next, self._src
        else:
            for var in self.trainingVar:
                tgt, _ = self._stage_target(var, augmenting, 0)
                tgt.set_value(self._macro_rows(var, macro_idx))
        self._prefetched = None
        for var in self.trainingVar:
            stage, rewritten = self._stage_target(var, augmenting, 0)
            if not rewritten:
                self.rt.copy(getattr(self, var).buf, stage.buf)      # device-to-device into the window

    def source_rows(self, var, idxs, macro_idx, last):
        """Device rows holding the UN-augmented samples `idxs` (contiguous) of `var` for the augmentation hooks: idxs are indices
        into keyDB (or keyDBlast when `last`), exactly what the reference's augment_poses indexes (poseregnettrainer.py:223-241)."""
        n = len(idxs)
        if self.getNumMacroBatches() == 1:
            if var not in self._src:
                self._src[var] = getattr(self, var) if var not in ('train_data_x', 'train_data_y') else \
                    DeviceData(self.rt, getattr(self, var + 'DB'), var + '_orig')
            return self._src[var].rows(idxs[0], n)
        base = 0 if last else macro_idx * self.getNumSamplesPerMacroBatch()
        return self._src[var].rows(idxs[0] - base, n)

    # ---- chunked host loading (para_load, nettrainer.py:512-526, 630-655, 701-723) ---------------------------------------
    def setupDataLoading(self):
        """para_load: the training set is one of `numChunks` chunks; while chunk c trains, `load_fun_params['fun']`
        (a trainer method `fun(params, chunk_idx, last, data_queue)` that fills data_queue[var][:]) prepares chunk c+1 on a
        host thread (file reading and NumPy release the GIL; the reference forks a process because its loop also augments on the
        CPU).  Augmentation needs no workers here: it is a kernel launch."""
        if self.cfgParams.para_load is True:
            import queue
            import threading
            assert self.numChunks > 1, "Please set the number of chunks appropriately!"
            self.load_recv_queue, self.load_send_queue = queue.Queue(), queue.Queue()
            self.load_data_queue = {}
            for var in self.trainingVar:
                if not hasattr(self, var):
                    raise ValueError("Variable " + var + " not defined!")
                if var.startswith("train_"):
                    sz = list(getattr(self, var + 'DB').shape)
                    if hasattr(self, var + 'DBlast'):
                        sz[0] += getattr(self, var + 'DBlast').shape[0]
                    self.load_data_queue[var] = numpy.full(tuple(sz), numpy.nan, dtype=getattr(self, var + 'DB').dtype) \
                        if getattr(self, var + 'DB').dtype.kind == 'f' else numpy.zeros(tuple(sz), getattr(self, var + 'DB').dtype)
            self._load_thread = threading.Thread(target=self.loadDataMP, args=(self.load_recv_queue, self.load_send_queue, self.load_data_queue),
                                                 daemon=True)
            self._load_thread.start()
            print("Loading chunk {}, last {}".format(0, False))
            self.load_recv_queue.put((0, self.cfgParams.load_fun_params, False))

    def loadDataMP(self, recv_queue, send_queue, data_queue):
        while True:
            (chunk_idx, params, last) = recv_queue.get()
            if chunk_idx == -1:
                return
            assert params['fun'] is not None
            getattr(self, params['fun'])(params, chunk_idx, last, data_queue)
            send_queue.put((chunk_idx, self.SYNC_LOAD_FINISHED))

    def _para_swap(self, macro_idx):
        """do_para_swap (nettrainer.py:512-526): at the last macro-batch of a chunk the freshly loaded chunk replaces the host
        arrays (it is what the NEXT macro-batch loads read), and the following chunk is requested."""
        if self.cfgParams.para_load is not True or self._load_thread is None or not self.isLastMacroBatch(macro_idx):
            return
        (ci, msg) = self.load_send_queue.get()
        assert msg == self.SYNC_LOAD_FINISHED
        # the asynchronous upload of a prefetched macro-batch may still be READING the pinned host arrays that are rewritten below:
        # the host waits for those copies (stream-side waits do not protect host memory)
        if self._prefetched is not None:
            for h in self._prefetched[1]:
                sync = getattr(h, 'synchronize', None)
                if sync is not None:
                    sync()
        for h in getattr(self, '_last_paged_in', ()):
            sync = getattr(h, 'synchronize', None)
            if sync is not None:
                sync()
        nmb, spm = self.getNumMacroBatches(), self.getNumSamplesPerMacroBatch()
        for var in self.trainingVar:
            if not hasattr(self, var):
                raise ValueError("Variable " + var + " not defined!")
            if var not in self.load_data_queue:
                continue
            new = self.load_data_queue[var]
            if nmb > 1:
                getattr(self, var + 'DB')[:] = new[0:(nmb - 1) * spm]
                getattr(self, var + 'DBlast')[:] = self.alignData(new[(nmb - 1) * spm:], fillData=new)
            else:
                getattr(self, var + 'DB')[:] = new
                # single macro-batch: the resident copy the augmentation reads is refreshed as well
                if var in self._src and self._src[var] is not getattr(self, var):
                    self._src[var].set_value(getattr(self, var + 'DB'))
                elif self.cfgParams.augment_fun_params['fun'] is None or var not in ('train_data_x', 'train_data_y'):
                    getattr(self, var).set_value(getattr(self, var + 'DB'))
        self._prefetched = None                                  # it was read from the old chunk
        self.currentChunk = ci
        next_chunk = int(numpy.mod(ci + 1, self.numChunks))
        print("Received chunk {}, requesting {}".format(ci, next_chunk))
        self.load_recv_queue.put((next_chunk, self.cfgParams.load_fun_params, False))

    def unsetDataLoading(self):
        if self._load_thread is not None:
            self.load_recv_queue.put((-1, None, False))
            self._load_thread.join(timeout=60)
            self._load_thread = None

    # ---- training loop (nettrainer.py:778-907) ---------------------------------------------------------------
    def train(self, n_epochs=50, storeFilters=False):
        if len(self.validation_observer) < 1:
            raise ValueError("Require at least 1 validation function, that monitors validation cost!")
        if self.cfgParams.augment_fun_params['fun'] is not None or self.cfgParams.load_fun_params['fun'] is not None:
            self.setupDataLoading()
        wvals = []
        n_val_batches = self.val_data_xDB.shape[0] // self.cfgParams.batch_size
        best_validation_loss = numpy.inf
        bestParams = None
        best_device = None             # (store, device buffers) of the best weights so far
        bestParamsEp = -1
        start_time = time.time()
        train_costs = []
        validation_obs = [[] for _ in range(1, len(self.validation_observer))]
        self.epoch = 0

        self.poseNet.setDeterministic()
        for vi in range(1, len(self.validation_observer)):
            validation_obs[vi - 1].append(_nanmean([self.validation_observer[vi](i) for i in range(n_val_batches)]))
        self.poseNet.unsetDeterministic()

        class _Now(object):                       # a trainer without an asynchronous step: the value is already there
            def __init__(self, v):
                self.v = v

            def get(self):
                return numpy.asarray([self.v])

        train_async = getattr(self, 'train_model_async', None) or (lambda i, lr: _Now(self.train_model(i, lr)))
        pending = None

        def resolve(p):
            idx, handle = p
            minibatch_avg_cost = float(numpy.asarray(handle.get()).reshape(-1)[0])
            print("minibatch {0:4d}, average cost: {1}".format(idx, minibatch_avg_cost))
            if numpy.any(numpy.isnan(minibatch_avg_cost)):
                self.checkNaNs()
                assert False
            train_costs.append(minibatch_avg_cost)

        while self.epoch < n_epochs:
            writer = self.dp is None or self.dp.rank == 0          # replicas hold the same weights: rank 0 writes the snapshots
            if self.epoch % self.cfgParams.snapshot_last == 0 and writer:
                self._snapshot(self.subfolder + '/net_last.pkl', rewritten=True)
            if self.cfgParams.snapshot_freq is not None:
                if self.epoch % self.cfgParams.snapshot_freq == 0 and writer:
                    self._snapshot(self.subfolder + '/net_{}.pkl'.format(self.epoch))
            if self.cfgParams.pre_epoch_fn is not None:
                getattr(self, self.cfgParams.pre_epoch_fn)()
            self.epoch += 1
            learning_rate = self.cfgParams.lr_of_ep(self.epoch)
            for minibatch_index in range(self.getNumFullMiniBatches()):
                if self.cfgParams.pre_minibatch_fn is not None:
                    getattr(self, self.cfgParams.pre_minibatch_fn)()
                self.poseNet.unsetDeterministic()
                mini_idx = self.loadMiniBatch(minibatch_index)
                # The reference reads the cost of every minibatch right after the device call (nettrainer.py:844-849).  Here the
                # next minibatch is queued first and the previous cost is read while it runs: same values, same order, the NaN
                # check one step late -- the host round trip no longer sits between two steps.
                launched = (minibatch_index, train_async(mini_idx, learning_rate))
                if pending is not None:
                    resolve(pending)
                pending = launched
                if self.cfgParams.post_minibatch_fn is not None:
                    getattr(self, self.cfgParams.post_minibatch_fn)()
                iter_count = (self.epoch - 1) * self.getNumFullMiniBatches() + minibatch_index
                last_of_epoch = minibatch_index == self.getNumFullMiniBatches() - 1
                if last_of_epoch or (iter_count + 1) % self.cfgParams.validation_frequency == 0:
                    resolve(pending)
                    pending = None
                if (iter_count + 1) % self.cfgParams.validation_frequency == 0:
                    if storeFilters:
                        for lay in self.poseNet.layers:
                            if lay.__class__.__name__ in ('ConvPoolLayer', 'ConvLayer'):
                                wvals.append(lay.W.get_value())
                    self.poseNet.setDeterministic()
                    this_validation_loss = _nanmean([self.validation_observer[0](i) for i in range(n_val_batches)])
                    for vi in range(1, len(self.validation_observer)):
                        validation_obs[vi - 1].append(_nanmean([self.validation_observer[vi](i) for i in range(n_val_batches)]))
                    self.poseNet.unsetDeterministic()
                    print("{}: epoch {}, LR {}, minibatch {}/{}, validation cost {} error {}".format(
                        time.ctime(), self.epoch, learning_rate, minibatch_index + 1, self.getNumFullMiniBatches(),
                        this_validation_loss, [vo[-1] for vo in validation_obs]))
                    if this_validation_loss < best_validation_loss:
                        best_validation_loss = this_validation_loss
                        print("Best validation loss so far, store network weights!")
                        # (nettrainer.py:871-876 copies every array to the host; here the copy stays on the device when it can)
                        snap = getattr(self.poseNet, 'deviceWeightSnapshot', lambda into=None: None)(best_device)
                        if snap is not None:
                            best_device, bestParams = snap, None
                        else:
                            best_device, bestParams = None, self.poseNet.weightVals
                        bestParamsEp = self.epoch
            if self.cfgParams.post_epoch_fn is not None:
                getattr(self, self.cfgParams.post_epoch_fn)()

        getattr(self.poseNet, 'joinSave', lambda: None)()          # the last epoch's snapshot is on disk when train() returns
        end_time = time.time()
        print('Optimization complete with best validation score of %f,' % best_validation_loss)
        print('The code run for %d epochs, with %f epochs/sec' % (self.epoch, self.epoch / max(1e-9, end_time - start_time)))
        if (bestParams is not None or best_device is not None) and self.cfgParams.use_early_stopping is True:
            if best_device is not None:
                self.poseNet.restoreDeviceWeightSnapshot(best_device)
            else:
                self.poseNet.weightVals = bestParams
            print('Best params at epoch %d' % bestParamsEp)
        if self.dp is not None and hasattr(self.poseNet, 'syncRunningStatistics'):
            # per-GPU BatchNorm statistics: every rank's running mean / inv_std followed its own shards.  From here on the replicas are
            # ONE model again -- rank 0's, the one whose checkpoints were written (every rank runs train(), so this is a safe collective)
            self.poseNet.syncRunningStatistics(self.dp)
        if self.cfgParams.augment_fun_params['fun'] is not None or self.cfgParams.load_fun_params['fun'] is not None:
            self.unsetDataLoading()
        return train_costs, wvals, validation_obs[0] if len(validation_obs) == 1 else validation_obs

    def _snapshot(self, path, rewritten=False):
        """The per-epoch checkpoint (nettrainer.py:816-820) without stopping the epoch loop for the pickle: NetBase.saveAsync copies the
        parameters on the device, moves them to the host on the copy stream and writes the file from a worker thread.  A net whose
        save() was replaced (an instance attribute) or that has no saveAsync is saved the plain way.  rewritten: the file is overwritten
        every epoch (`net_last.pkl`): a snapshot is dropped while the previous one is still being written (epochs shorter than the
        write of a 75 MB pickle; NYU's 568-minibatch epochs are 5x longer than it); numbered snapshots are always written."""
        net = self.poseNet
        fn = getattr(net, 'saveAsync', None)
        if fn is None or 'save' in vars(net):
            net.save(path)
        else:
            fn(path, skip_if_busy=rewritten)

    def checkNaNs(self):
        for param_i in self.params:
            if numpy.any(numpy.isnan(param_i.get_value())):
                print("NaN in weights", param_i.name)

    # ---- single-crop augmentation (nettrainer.py:919-997) ------------------------------------------------------
    def augmentCrop(self, img, gt3Dcrop, com, cube, M, aug_modes, hd, normZeroOne=False, sigma_com=None, sigma_sc=None,
                    rot_range=None):
        """Same contract as the reference for ONE crop: draws (mode, off, rot, sc) from self.rng in the reference's order
        and runs the device kernels on a batch of one.  `com` is in image coordinates like in the reference.
        Returns (imgD, None, curLabel, cube, com, M, rot)."""
        from hipdp import ops
        from hipdp.augmenter import MODE_CODE, camera_tuple
        assert len(img.shape) == 2
        assert isinstance(aug_modes, list)
        sigma_com = 5. if sigma_com is None else sigma_com
        sigma_sc = 0.02 if sigma_sc is None else sigma_sc
        rot_range = 180. if rot_range is None else rot_range
        mode = self.rng.randint(0, len(aug_modes))
        off = self.rng.randn(3) * sigma_com
        rot = self.rng.uniform(-rot_range, rot_range)
        sc = abs(1. + self.rng.randn() * sigma_sc)
        rt = self.rt
        J = int(numpy.asarray(gt3Dcrop).reshape(-1, 3).shape[0])
        com3d = numpy.asarray(hd.importer.jointImgTo3D(com), numpy.float32)
        f32 = lambda a: rt.upload(numpy.ascontiguousarray(a, numpy.float32))       # noqa: E731
        rec = rt.alloc(rt.lib.dpp_augment_record_bytes(), numpy.uint8)
        out_y, out_x = rt.alloc((1, J * 3)), rt.alloc((1,) + img.shape)
        imgb = f32(img[None])
        ops.augment_prepare(rt, imgb, f32(com3d[None]), f32(numpy.asarray(cube)[None]), f32(numpy.asarray(M).reshape(1, 9)),
                            f32(numpy.asarray(gt3Dcrop).reshape(1, J, 3)), 1, J, img.shape[0], camera_tuple(hd.importer), rec, out_y,
                            mode=rt.upload(numpy.array([MODE_CODE[aug_modes[mode]]], numpy.int32)),
                            off=rt.upload(numpy.asarray(off, numpy.float64)), rot=rt.upload(numpy.array([rot], numpy.float64)),
                            sc=rt.upload(numpy.array([sc], numpy.float64)), norm_zero_one=bool(normZeroOne))(rt.stream)
        ops.augment_warp(rt, imgb, rec, 1, img.shape[0], out_x)(rt.stream)
        rt.synchronize()
        new_cube = numpy.asarray(cube, numpy.float64) * (sc if aug_modes[mode] == 'sc' else 1.0)
        return out_x.get()[0], None, out_y.get().reshape(J, 3), new_cube, com, M, rot
