"""
NetBase / NetBaseParams (API of /root/reference/src/net/netbase.py:52-477): the net container the scripts use --
`layers`, `output`, `params`, `weights`, `computeOutput`, deterministic-mode toggles, `weightVals`, and the
pickle checkpoint format {'class', 'network': str(net), '<layerNum>-values': [params..., params_nontrained...]}.

Execution is delegated to hipdp.engine, which compiles the graph reachable from `self.output` into HIP kernel
launches (and is recompiled whenever `output` / `layers` are re-pointed, as main_*_posereg_embedding.py does
when it appends the PCA-prior layer).
"""
import difflib
import gzip
import os
import pickle
import time

import numpy

from net.batchnormlayer import BatchNormLayer, BatchNormLayerParams
from net.convlayer import ConvLayer, ConvLayerParams
from net.convpoollayer import ConvPoolLayer, ConvPoolLayerParams
from net.dropoutlayer import DropoutLayer, DropoutLayerParams
from net.hiddenlayer import HiddenLayer, HiddenLayerParams
from net.nonlinearitylayer import NonlinearityLayer, NonlinearityLayerParams
from net.poollayer import PoolLayer, PoolLayerParams

_LAYER_CLASSES = dict(ConvPoolLayer=ConvPoolLayer, ConvLayer=ConvLayer, HiddenLayer=HiddenLayer, PoolLayer=PoolLayer,
                      DropoutLayer=DropoutLayer, BatchNormLayer=BatchNormLayer, NonlinearityLayer=NonlinearityLayer)


class NetBaseParams(object):
    def __init__(self):
        self.numInputs = 1
        self.numOutputs = 1
        self.layers = []
        self.inputDim = None
        self.outputDim = None
        self.loadFile = None

    def getMemoryRequirement(self):
        return sum(l.getMemoryRequirement() for l in self.layers)


def _shared_variables(layers, attr):
    """The shared variables the layers list under `attr` ('params' | 'weights'), each once (twins and shared towers list the
    same variable several times), in layer order."""
    found = {}
    for layer in layers:
        for var in getattr(layer, attr):
            found.setdefault(var.auto_name, var)
    return list(found.values())


def _blocklist(attr, what, match):
    """The `<attr>` / `<attr>_filter` property pair of netbase.py:157-216: `<attr>` enumerates the layers' shared variables without
    the ones on the block list, assigning the block list checks that every entry belongs to the model (UserWarning otherwise).
    `match` names the attribute of a variable that is compared with the blocked auto_names (the reference compares `auto_name` for
    the parameters and `name` for the weights)."""
    slot = '_%s_filter' % attr

    def current(net):
        return net.__dict__.setdefault(slot, [])

    def listed(net):
        hidden = set(b.auto_name for b in current(net))
        return [v for v in _shared_variables(net.layers, attr) if getattr(v, match) not in hidden]

    def assign(net, blocked):
        known = set(v.auto_name for v in _shared_variables(net.layers, attr))
        for b in blocked:
            if b.auto_name not in known:
                raise UserWarning("%s {} not in model!".format(b) % what)
        net.__dict__[slot] = blocked

    return property(listed), property(current, assign)


class NetBase(object):
    def __init__(self, rng, inputVar, cfgParams, twin=None):
        self.inputVar = inputVar
        self.cfgParams = cfgParams
        self.rng = rng
        self._twin = twin                 # layers built with copyLayer = twin.layers[i] (netbase.py:117): one parameter store for both
        self.layers = []
        for num, spec in enumerate(cfgParams.layers):
            build = _LAYER_CLASSES[type(spec).__name__[:-len('Params')]]            # '<X>LayerParams' -> '<X>Layer'
            self.layers.append(build(rng, inputVar=self._input_of(num, spec), cfgParams=spec,
                                     copyLayer=twin.layers[num] if twin is not None else None, layerNum=num))
        self.output = self.layers[-1].output
        self.load(self.cfgParams.loadFile)

    def _input_of(self, num, spec):
        """What layer `num` reads: the net's input, or the previous layer's output -- flattened in front of the first dense layer,
        reshaped back to a map in front of a convolution that follows one (netbase.py:100-113)."""
        if num == 0:
            return self.inputVar
        before = self.layers[-1]
        rank_out, rank_in = len(before.cfgParams.outputDim), len(spec.inputDim)
        if rank_out == rank_in:
            return before.output
        if (rank_out, rank_in) == (4, 2):
            var = before.output.flatten(2)
        elif (rank_out, rank_in) == (2, 4):
            var = before.output.reshape(spec.inputDim, ndim=4)
        else:
            return before.output
        var.name = "input_layer_{}".format(num)
        return var

    def __str__(self):
        # (part of the checkpoint format: load() compares it with the stored description)
        rows = ["Layer {}: {} with {} \n".format(num, type(layer).__name__, layer) for num, layer in enumerate(self.layers)]
        return "Network configuration:\n" + "".join(rows)

    # ---- parameter enumeration -------------------------------------------------------------------------
    all_params = property(lambda self: _shared_variables(self.layers, 'params'))
    all_weights = property(lambda self: _shared_variables(self.layers, 'weights'))
    params, params_filter = _blocklist('params', 'Param', 'auto_name')
    weights, weights_filter = _blocklist('weights', 'Weight', 'name')

    # ---- inference ---------------------------------------------------------------------------------------
    def _engine(self, runtime=None):
        """The compiled inference engine for the CURRENT graph (recompiled when output/layers were re-pointed)."""
        from hipdp import engine as _engine
        key = (id(self.output), len(self.layers), tuple(self.cfgParams.outputDim))
        eng = getattr(self, '_compiled', None)
        if eng is None or eng.key != key or (runtime is not None and eng.rt is not runtime):
            eng = _engine.CompiledNet(self, train=False, runtime=runtime)
            eng.key = key
            self._compiled = eng
        return eng

    def computeOutput(self, inputs, timeit=False, dp=None):
        """Batched deterministic forward; the last batch is padded by repeating the last sample
        (/root/reference/src/net/netbase.py:217-316).
        dp (a hipdp.parallel.DataParallel; default: the one a data-parallel trainer attached to the net as `net.dp`): the test set is
        dealt to the ranks batch by batch -- rank r evaluates batches r, r + G, r + 2G, ... on its own GPU -- and the outputs are
        all-gathered, so every rank returns the full array ("Validation / computeOutput: shard batches,
        all-gather outputs").  A COLLECTIVE: every rank has to call it, with the same inputs -- a rank-conditional call (evaluation
        under `if writer:`) deadlocks; pass dp=False for a private, un-sharded evaluation on one rank.
        The trained parameters of the replicas are identical (replicated ADAM on the all-reduced gradient), the BatchNorm RUNNING
        statistics are not: with per-GPU batch statistics (DataParallel(sync_bn=False), the default) every rank's EMA follows its own
        shards.  Rank 0's statistics -- the ones in the checkpoint rank 0 writes -- are therefore broadcast before the batches are
        dealt out, so a batch goes through the same numbers and the same kernels whichever rank runs it and the result equals the
        single-process evaluation of rank 0's checkpoint bit for bit.  The rank's OWN statistics are restored when the call returns
        (round 6, ADVICE r5: the broadcast used to overwrite them for good)."""
        if not isinstance(inputs, list):
            inputs = [inputs]
        assert all(i.shape[0] == inputs[0].shape[0] for i in inputs[1:])
        if isinstance(self.output, list):
            raise NotImplementedError("multi-output nets")
        if not self.isDeterministic():
            print("WARNING: network is probabilistic for testing, DISABLING")
            self.setDeterministic()
        if dp is None:
            dp = getattr(self, 'dp', None)
        if dp is False:
            dp = None
        G, rank = (dp.world, dp.rank) if dp is not None else (1, 0)
        batch_size = self.cfgParams.batch_size
        nSamp = inputs[0].shape[0]
        padSize = int(batch_size * numpy.ceil(nSamp / float(batch_size)))
        outSize = list(self.cfgParams.outputDim)
        outSize[0] = padSize
        out = numpy.zeros(tuple(outSize), dtype='float32')
        eng = self._engine()
        local_nt = None
        if G > 1:
            # rank 0's running statistics for the evaluation only: the rank's own are put back afterwards (an inference call in the
            # middle of training must not rewrite the state of the replicas -- train()'s epilogue is where they are made one model)
            store = self._live_store()
            if store is not None:
                st = self.__dict__.setdefault('_nt_stage', {})
                if st.get('store') is not store:
                    st.clear()
                    st.update(store=store, buf=store.rt.alloc(store.n_nt, zero=False))
                local_nt = st['buf']
                store.rt.copy(local_nt, store.nt)
            self.syncRunningStatistics(dp)
        n_test_batches = padSize // batch_size
        start = time.time()

        def batch_of(i):
            chunks = []
            for x in inputs:                              # one array per network input (three for ScaleNet)
                chunk = x[i * batch_size:(i + 1) * batch_size]
                if chunk.shape[0] < batch_size:
                    pad = numpy.zeros((batch_size,) + tuple(x.shape[1:]), dtype=x.dtype)
                    pad[0:chunk.shape[0]] = chunk
                    pad[chunk.shape[0]:] = x[-1]
                    chunk = pad
                chunks.append(chunk)
            return chunks
        mine = list(range(rank, n_test_batches, G))
        if len(mine) > 1 and hasattr(eng.rt, 'staged_upload') and len(inputs) == len(eng.x_ins):
            self._compute_output_pipelined(eng, mine, batch_of, out, batch_size)
        else:
            for i in mine:
                chunks = batch_of(i)
                o = eng.forward(chunks if len(chunks) > 1 else chunks[0])
                out[i * batch_size:(i + 1) * batch_size] = o.reshape(self.cfgParams.outputDim)
        if G > 1:
            # every rank sends the rows of ITS batches (padded to the same count) and places the others' by batch index
            per = -(-n_test_batches // G)
            mine = numpy.zeros((per * batch_size,) + tuple(outSize[1:]), dtype='float32')
            for k, i in enumerate(range(rank, n_test_batches, G)):
                mine[k * batch_size:(k + 1) * batch_size] = out[i * batch_size:(i + 1) * batch_size]
            for r, rows in enumerate(dp.all_gather_host(mine)):
                for k, i in enumerate(range(r, n_test_batches, G)):
                    out[i * batch_size:(i + 1) * batch_size] = rows[k * batch_size:(k + 1) * batch_size]
        if local_nt is not None:
            store = self._live_store()
            store.rt.copy(store.nt, local_nt)
        end = time.time()
        if timeit:
            print("{} in {}s, {}ms per frame".format(padSize, end - start, (end - start) * 1000. / padSize))
        return out[0:nSamp]

    def _compute_output_pipelined(self, eng, batches, batch_of, out, batch_size):
        """The batch loop of computeOutput with the PCIe transfer of batch i + 1 under the evaluation of batch i.  The reference uploads a
        batch, runs the compiled function and reads the result back, one after the other (netbase.py:286-310); done that way here the device
        waits 0.2 ms per batch for its input (0.845 ms per batch of 128 against 0.64 ms of device time).  Two device staging buffers per
        network input: batch i + 1 goes into one of them on the copy stream -- ordered only behind the device-to-device copy that last read
        that buffer -- while the main stream copies the other one into the engine's input and evaluates it; outputs come back through
        asynchronous reads that are resolved one batch late.  Same kernels, same values."""
        from hipdp import layout as _layout
        rt = eng.rt
        eng.store.check_live()
        stages = [[rt.alloc(t.shape, zero=False) for t in eng.x_ins] for _ in range(2)]
        free = [None, None]
        held = [None, None]              # the upload events (and through them the host source arrays) of the batch in each staging slot
        pending = None

        def to_device_layout(a, t):
            a = numpy.asarray(a, numpy.float32)
            want = (t.shape[0], t.shape[3], t.shape[1], t.shape[2])
            if tuple(a.shape) != want:
                raise ValueError("input shape %s, expected %s" % (a.shape, want))
            return a if a.shape[1] == 1 else _layout.nchw_to_nhwc(a)
        for n, i in enumerate(batches):
            k = n & 1
            ups = [rt.staged_upload(stages[k][j], to_device_layout(a, t), free[k]) for j, (a, t) in enumerate(zip(batch_of(i), eng.x_ins))]
            for j, t in enumerate(eng.x_ins):
                rt.wait_event(ups[j])
                rt.copy(t.buf, stages[k][j])
            held[k] = ups
            free[k] = rt.record_event()
            eng.fwd.run(rt)
            handle = rt.read_async(eng.out.buf)
            if pending is not None:
                pi, ph = pending
                out[pi * batch_size:(pi + 1) * batch_size] = ph.get().reshape(self.cfgParams.outputDim)
            pending = (i, handle)
        pi, ph = pending
        out[pi * batch_size:(pi + 1) * batch_size] = ph.get().reshape(self.cfgParams.outputDim)

    def syncRunningStatistics(self, dp):
        """Rank 0's non-trained parameters (BatchNorm running mean / inv_std) on every rank.  A collective (one broadcast of the flat
        statistics buffer, a few KB).  No-op before the net has a device store."""
        store = self._live_store()
        if store is not None and dp is not None and dp.world > 1:
            dp.broadcast(store.nt)

    # ---- mode toggles ------------------------------------------------------------------------------------
    def unsetDeterministic(self):
        for layer in self.layers:
            if isinstance(layer, (DropoutLayer, BatchNormLayer)):
                layer.unsetDeterministic()

    def setDeterministic(self):
        for layer in self.layers:
            if isinstance(layer, (DropoutLayer, BatchNormLayer)):
                layer.setDeterministic()

    def isDeterministic(self):
        for layer in self.layers:
            if isinstance(layer, (DropoutLayer, BatchNormLayer)) and not layer.isDeterministic():
                return False
        return True

    def hasDropout(self):
        return any(isinstance(layer, DropoutLayer) for layer in self.layers)

    # ---- weight values -----------------------------------------------------------------------------------
    @property
    def weightVals(self):
        return self.recGetWeightVals(self.all_params)

    @weightVals.setter
    def weightVals(self, value):
        self.recSetWeightVals(self.all_params, value)

    def recSetWeightVals(self, param, value):
        if isinstance(value, list):
            assert isinstance(param, list), "tried to assign a list of weights to params, which is not a list {}".format(type(param))
            assert len(param) == len(value), "tried to assign unequal list of weights {} != {}".format(len(param), len(value))
            for i in range(len(value)):
                self.recSetWeightVals(param[i], value[i])
        else:
            param.set_value(value)

    def recGetWeightVals(self, param):
        if isinstance(param, list):
            return [self.recGetWeightVals(p) for p in param]
        return param.get_value()

    # ---- checkpoints -------------------------------------------------------------------------------------
    def _live_store(self):
        """The device parameter store the net's parameters are bound to (None before the first compile / after a rebuild)."""
        store = getattr(self, '_param_store', None)
        return store if store is not None and not store.released else None

    def deviceWeightSnapshot(self, into=None):
        """A device-resident copy of all weights and running statistics (None when the net has no device store yet): the epoch loop
        keeps its best weights this way instead of pulling ~270 arrays to the host at every improved validation."""
        store = self._live_store()
        return None if store is None else (store, store.snapshot(into[1] if into is not None and into[0] is store else None))

    def restoreDeviceWeightSnapshot(self, snap):
        store, bufs = snap
        store.restore(bufs)

    def _checkpoint_state(self, bulk):
        """The checkpoint dictionary (netbase.py:318-346 of the reference: class, description, per-layer value lists); `bulk`:
        {auto_name: value} of the parameters that live on the device."""
        state = dict([('class', self.__class__.__name__), ('network', self.__str__())])

        def value(p):
            v = bulk.get(getattr(p, 'auto_name', None))
            return numpy.array(p.get_value() if v is None else v)
        for layer in self.layers:
            key = '{}-values'.format(layer.layerNum)
            state[key] = [value(p) for p in layer.params]
            state[key].extend([value(p) for p in layer.params_nontrained])
        return state

    @staticmethod
    def _write_checkpoint(state, filename):
        opener = gzip.open if filename.lower().endswith('.gz') else open
        with opener(filename, 'wb') as handle:
            pickle.dump(state, handle, 2)          # protocol 2 = what cPickle wrote; readable by the reference
        print('Saved model parameter to {}'.format(filename))

    def save(self, filename):
        self.joinSave()
        store = self._live_store()
        bulk = store.bulk_values() if store is not None else {}      # two device -> host copies for the whole net
        self._write_checkpoint(self._checkpoint_state(bulk), filename)

    def saveAsync(self, filename, skip_if_busy=False):
        """save() off the calling thread (the epoch loop's per-epoch `net_last.pkl`, /root/reference/src/trainer/nettrainer.py:816-820:
        a 75 MB protocol-2 pickle is 0.6 s of interpreter time per epoch of the 128x128 ResNet).  The parameters are copied
        device-to-device into a staging buffer on the current stream (50 us; training may go on changing the live ones), that buffer goes
        to page-locked host memory on the copy stream, a helper thread dumps it raw (GIL released) and a writer PROCESS
        (hipdp/ckpt_writer.py: NumPy only) converts the layouts and writes the SAME bytes save() writes -- a writer thread would hold the
        GIL for the pickle and stall the training thread with it.  joinSave() -- called by the next saveAsync / save and at the end of
        train() -- waits for it.
        skip_if_busy: for a file that is rewritten over and over (`net_last.pkl`): when the writer of the previous snapshot is still at
        work -- epochs shorter than a checkpoint write -- this snapshot is dropped instead of stalling the epoch loop behind it; the
        file then holds the state of the previous epoch until the next one lands.  Returns True when a snapshot was started / written."""
        store = self._live_store()
        if store is None or not hasattr(store.rt, 'download_async'):
            self.save(filename)
            return True
        names = [[getattr(p, 'auto_name', None) for p in layer.params + layer.params_nontrained] for layer in self.layers]
        if any(n not in store.by_param for ns in names for n in ns):
            self.save(filename)                  # a parameter that does not live in the device store: the plain path reads it
            return True
        pend = self.__dict__.get('_save_pending')
        if skip_if_busy and pend is not None and pend[0].is_alive():
            # remembered: joinSave() writes the LATEST dropped snapshot request once the running writer is done (the reference's
            # net_last.pkl holds the state at the start of the last epoch, nettrainer.py:816-820; without this the file could end up
            # several epochs stale)
            self._save_dropped = filename
            print('Snapshot {} skipped (previous checkpoint write still running); it is written when that one finishes'.format(filename))
            return False
        self.__dict__.pop('_save_dropped', None)          # superseded by the snapshot taken now
        self.joinSave()
        import subprocess
        import sys
        import tempfile
        import threading
        rt = store.rt
        st = self.__dict__.setdefault('_save_stage', {})
        if st.get('store') is not store:
            st.clear()
            st.update(store=store, w=rt.alloc(store.n_w, zero=False), nt=rt.alloc(store.n_nt, zero=False), hw=None, hnt=None)
        rt.copy(st['w'], store.w)
        rt.copy(st['nt'], store.nt)
        rw, st['hw'] = rt.download_async(st['w'], st['hw'])
        rn, st['hnt'] = rt.download_async(st['nt'], st['hnt'])
        # everything that reads the net object is done NOW, on the caller's thread
        meta = {'class': self.__class__.__name__, 'network': self.__str__(), 'n_w': int(store.n_w), 'n_nt': int(store.n_nt),
                'layers': [(layer.layerNum, ns) for layer, ns in zip(self.layers, names)],
                'slots': [(s['param'].auto_name, 'w' if s['trained'] else 'nt', s['off'], s['size'], s['kind'], s['info'], tuple(s['shape']))
                          for s in store.slots]}
        tmpdir = self._scratch_dir(4 * (int(store.n_w) + int(store.n_nt)) + (1 << 20))
        tag = 'dpp_ckpt_%d_%d' % (os.getpid(), id(self) & 0xffffff)
        pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # the directory that holds hipdp/
        err = []

        def work():
            # A per-epoch snapshot is best effort and must never end a training run that a plain save() would have survived: when the
            # hand-over to the writer process fails (scratch directory full -- Docker's default /dev/shm is 64 MB, the 128x128 net dumps
            # 75 MB --, no process to be had, the writer itself dying) the scratch files are removed and the SAME state is written the
            # synchronous way, in this thread, from the host copies already made.  Only if that fails too is the error kept for joinSave().
            host = {}
            meta_path = raw_path = None
            try:
                host['w'] = numpy.ascontiguousarray(rw.get(), numpy.float32).reshape(-1)
                host['nt'] = numpy.ascontiguousarray(rn.get(), numpy.float32).reshape(-1)
                if tmpdir is None:
                    raise OSError("no scratch directory with room for the parameter dump")
                meta_path, raw_path = os.path.join(tmpdir, tag + '.meta'), os.path.join(tmpdir, tag + '.f32')
                with open(meta_path, 'wb') as fh:
                    pickle.dump(meta, fh, 4)
                with open(raw_path, 'wb') as fh:          # ndarray.tofile releases the GIL while it writes
                    host['w'].tofile(fh)
                    host['nt'].tofile(fh)
                env = dict(os.environ, PYTHONPATH=pkg + os.pathsep + os.environ.get('PYTHONPATH', ''))
                r = subprocess.run([sys.executable, '-m', 'hipdp.ckpt_writer', meta_path, raw_path, filename], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                if r.returncode != 0:
                    raise RuntimeError("checkpoint writer failed for %s: %s" % (filename, r.stdout.decode(errors='replace')[-2000:]))
                print('Saved model parameter to {}'.format(filename))
            except BaseException as e:           # noqa: BLE001
                for f in (meta_path, raw_path, filename + '.part'):
                    try:
                        if f is not None:
                            os.remove(f)
                    except OSError:
                        pass
                try:
                    if 'nt' not in host:
                        raise e
                    from hipdp import ckpt_writer
                    print('WARNING: asynchronous checkpoint of {} failed ({}: {}); writing it synchronously'.format(filename, type(e).__name__, e))
                    ckpt_writer.write(ckpt_writer.build_state(meta, host), filename)
                    print('Saved model parameter to {}'.format(filename))
                except BaseException as e2:      # noqa: BLE001  (raised by joinSave(strict=True); reported otherwise)
                    err.append(e2)
        th = threading.Thread(target=work, name='dpp-checkpoint')
        th.start()
        self._save_pending = (th, err, filename)
        return True

    @staticmethod
    def _scratch_dir(need_bytes):
        """A directory with room for the raw parameter dump: /dev/shm when it has the space, else the temporary directory, else None."""
        import shutil
        import tempfile
        for d in ('/dev/shm', tempfile.gettempdir()):
            try:
                if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need_bytes:
                    return d
            except OSError:
                pass
        return None

    def joinSave(self, strict=False):
        """Wait for the checkpoint a saveAsync() started (no-op otherwise), then write the snapshot request that was dropped while it ran
        (skip_if_busy), if any.  A snapshot whose asynchronous AND synchronous write both failed is reported, not raised (the epoch loop's
        `net_last.pkl` is best effort: a full disk must not end the run) -- unless strict."""
        pend = self.__dict__.pop('_save_pending', None)
        if pend is not None:
            pend[0].join()
            if pend[1]:
                if strict:
                    raise pend[1][0]
                print('WARNING: checkpoint {} was NOT written: {}: {}'.format(pend[2], type(pend[1][0]).__name__, pend[1][0]))
        dropped = self.__dict__.pop('_save_dropped', None)
        if dropped is not None:
            try:
                self.save(dropped)
            except (OSError, RuntimeError) as e:
                if strict:
                    raise
                print('WARNING: checkpoint {} was NOT written: {}: {}'.format(dropped, type(e).__name__, e))

    def load(self, filename, raise_on_error=True):
        if filename is None:
            return
        print('Loading model parameters from {}'.format(filename))
        opener = gzip.open if filename.lower().endswith('.gz') else open
        with opener(filename, 'rb') as handle:
            saved = pickle.load(handle, encoding='latin1')     # py2 cPickle files
        if saved['network'] != self.__str__():
            print("Possibly not matching network configuration!")
            differences = list(difflib.Differ().compare(saved['network'].splitlines(), self.__str__().splitlines()))
            print("Differences are:")
            print("\n".join(differences))
        for layer in self.layers:
            key = '{}-values'.format(layer.layerNum)
            if key not in saved:
                if raise_on_error:
                    raise ImportError("{} not in saved variables!".format(key))
                print("WARNING: {} not in saved variables!".format(key))
                continue
            mine = layer.params + layer.params_nontrained
            if len(mine) != len(saved[key]):
                print("Warning: Layer parameters for layer {} do not match. Trying to fit on shape!".format(layer.layerNum))
                n_assigned = 0
                for p in mine:
                    for v in saved[key]:
                        if p.get_value().shape == v.shape:
                            p.set_value(v)
                            n_assigned += 1
                if n_assigned != len(mine):
                    if raise_on_error:
                        raise ImportError("Could not load all necessary variables!")
                    print("WARNING: Could not load all necessary variables!")
                else:
                    print("Found fitting parameters!")
            else:
                for p, v in zip(mine, saved[key]):
                    if p.get_value().shape == v.shape:
                        p.set_value(v)
                    elif raise_on_error:
                        raise ImportError("Skipping parameter for {}! Shape {} does not fit {}.".format(p.name, p.get_value().shape, v.shape))
                    else:
                        print("WARNING: Skipping parameter for {}! Shape {} does not fit {}.".format(p.name, p.get_value().shape, v.shape))
        print('Done')
