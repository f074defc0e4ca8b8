"""
Data parallelism for the train step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node; "gloo" in the CPU tests).  The reference is single-device; this is new design:

  * every rank owns a contiguous shard of the global minibatch and augments it on its own GPU (no exchange);
  * the cost is normalised by the GLOBAL batch, so per-rank gradients are partial sums and an all-reduce (sum) of the
    flat fp32 gradient buffer (74.9 MB for the 128x128 ResNet) between backward and ADAM reproduces the single-device
    gradient; ADAM is replicated.  The buffer goes in two buckets: FC1's filter gradient (67 MB, 90 % of the bytes) is
    complete a few kernels into the backward pass and is all-reduced from the side stream while the rest of the
    backward pass runs; the remaining 8 MB follow after the last reduction;
  * BatchNorm: `sync_bn=False` uses per-GPU batch statistics (fast mode); `sync_bn=True` all-gathers the per-block
    (mean, M2) / (sum g, sum g*xhat) partials of every BatchNorm (2 x C floats x blocks) so that statistics and their
    gradients are those of the global batch (parity mode: N-GPU gradients == 1-GPU gradients on the same global batch).
"""
import os


class CollectiveOp(object):
    """A step of a launch plan that is a collective instead of a kernel launch."""
    meta = None

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, stream):
        self.fn()


class _Works(object):
    """Several asynchronous collectives as one work object."""

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class _SideJoin(object):
    """Stand-in for the work object of an asynchronous collective whose result was written on the side stream: wait() makes the
    current (main) stream wait for that stream."""

    def __init__(self, rt):
        self.rt = rt

    def wait(self):
        self.rt.main_wait_side()


class DataParallel(object):
    def __init__(self, rt, sync_bn=False):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (use hipdp.parallel.init_from_env)")
        self.dist = dist
        self.rt = rt
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.sync_bn = bool(sync_bn)
        # gloo with device tensors (several ranks sharing ONE MI355X: the control-flow / parity tests of the GPU tier,
        # DPP_DIST_BACKEND=gloo): the collectives run on host copies.  Blocking, correct against the HIP streams (the device -> host
        # copy waits for the stream it is issued on), and nothing a measured run ever takes: bench.py insists on RCCL.
        self.host_staged = dist.get_backend() == 'gloo' and not getattr(rt, 'is_emulator', False)
        # How a gradient bucket is summed (round 6: an A/B switch for the 8-GPU box, which this build never met):
        #   DPP_ALLREDUCE=allreduce (default)  one all_reduce of the bucket -- RCCL picks ring / tree;
        #   DPP_ALLREDUCE=rs_ag                reduce_scatter + all_gather on the flat buffer: every rank sums ONE 1/world slice of the
        #                                      bucket (in place: its slice of the buffer is the output) and the slices are gathered
        #                                      back -- the "direct" schedule of SURVEY.md section 5 (xGMI is point-to-point: 7 links per
        #                                      GPU, a ring is bound by one link), the tail that does not divide by world goes through a
        #                                      small all_reduce.  Same sums in another order of additions (float32: last bits).
        self.schedule = os.environ.get('DPP_ALLREDUCE', 'allreduce')
        if self.schedule not in ('allreduce', 'rs_ag'):
            raise ValueError("DPP_ALLREDUCE must be 'allreduce' or 'rs_ag', got %r" % self.schedule)
        # bench.py --gpus N: with measure_exposed set, every join with a collective is bracketed by two events on the main stream
        # (exposed_events); their distance is the time the main stream stood still for the exchange
        self.measure_exposed = False
        self.exposed_events = []

    def _collective(self, fn, *bufs):
        """fn(*tensors) on the buffers -- directly (RCCL on device memory; gloo on the emulator's host memory) or on host copies that
        are written back (gloo, device memory)."""
        ts = [self.rt.tensor(b) for b in bufs]
        if not self.host_staged:
            return fn(*ts)
        hs = [t.cpu() for t in ts]
        out = fn(*hs)
        for t, h in zip(ts, hs):
            t.copy_(h)
        return out

    def _sum_tensor(self, t, async_op=False):
        """Sum tensor t (flat, float32) over the ranks in place by the configured schedule; returns a work object when async_op."""
        dist = self.dist
        if self.schedule == 'allreduce':
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
        flat = t.view(-1)
        n, G = flat.numel(), self.world
        per = n // G
        m = per * G
        rccl = dist.get_backend() == 'nccl'
        # RCCL orders the collectives of a communicator on its stream, so the three may be queued back to back (and reduce in place: a
        # rank's output is its own slice of the input).  The host backend of the tests has no such order: blocking calls on copies.
        queued = async_op and rccl
        works = []
        if per:
            mine = flat[self.rank * per:(self.rank + 1) * per]
            works.append(dist.reduce_scatter_tensor(mine, flat[:m] if rccl else flat[:m].clone(), op=dist.ReduceOp.SUM, async_op=queued))
            works.append(dist.all_gather_into_tensor(flat[:m], mine if rccl else mine.clone(), async_op=queued))
        if n > m:
            works.append(dist.all_reduce(flat[m:], op=dist.ReduceOp.SUM, async_op=queued))
        return _Works(works if queued else []) if async_op else None

    def _bracket(self, fn):
        """fn() between two events on the main stream when the exposed time of the exchanges is being measured."""
        if not self.measure_exposed or not hasattr(self.rt, 'torch'):
            return fn()
        tc = self.rt.torch.cuda
        e0, e1 = tc.Event(enable_timing=True), tc.Event(enable_timing=True)
        e0.record(tc.current_stream())
        out = fn()
        e1.record(tc.current_stream())
        self.exposed_events.append((e0, e1))
        return out

    def exposed_ms(self):
        """Sum of the bracketed intervals since the last call (synchronise the device first)."""
        ms = sum(a.elapsed_time(b) for a, b in self.exposed_events)
        self.exposed_events = []
        return ms

    def allreduce_sum_op(self, buf, name='allreduce_sum'):
        return CollectiveOp(lambda: self._bracket(lambda: self._collective(lambda t: self._sum_tensor(t), buf)), name)

    def allreduce_sum_async_op(self, buf, handle, name='allreduce_sum_async'):
        """Start summing `buf` over the ranks WITHOUT blocking the stream the step continues on: issued from the side
        stream (right after the kernel that produced `buf`), RCCL runs it on its own stream behind the side stream's work;
        `handle` (a one-element list) receives the work object that `wait_op` later joins into the main stream."""
        t = self.rt.tensor(buf)

        def start():
            side = getattr(self.rt, '_side', None)
            two = side is not None and getattr(self.rt, 'has_side_stream', False)
            if self.host_staged:
                if two:
                    with self.rt.torch.cuda.stream(side):
                        self._collective(lambda h: self._sum_tensor(h), buf)
                    handle[0] = _SideJoin(self.rt)
                else:
                    self._collective(lambda h: self._sum_tensor(h), buf)
                    handle[0] = None
            elif two:
                with self.rt.torch.cuda.stream(side):
                    handle[0] = self._sum_tensor(t, async_op=True)
            else:
                handle[0] = self._sum_tensor(t, async_op=True)
        return CollectiveOp(start, name)

    def wait_op(self, handle, name='allreduce_wait'):
        """The current (main) stream waits for the collective started by allreduce_sum_async_op."""
        def wait():
            if handle[0] is not None:
                self._bracket(handle[0].wait)
                handle[0] = None
        return CollectiveOp(wait, name)

    def all_gather_op(self, src, dst, name='all_gather'):
        """dst (world * src.size floats) <- concatenation over ranks of src."""
        ts, td = self.rt.tensor(src), self.rt.tensor(dst)
        assert td.numel() == self.world * ts.numel()
        return CollectiveOp(lambda: self._collective(lambda d, s_: self.dist.all_gather_into_tensor(d, s_), dst, src), name)

    def broadcast(self, buf, src=0):
        self._collective(lambda t: self.dist.broadcast(t, src=src), buf)

    def all_gather_host(self, arr):
        """[rank 0's array, rank 1's, ...] of equally shaped host arrays (test-time outputs: NetBase.computeOutput(dp=)); through the
        device under RCCL, on the host tensors under gloo.  A collective."""
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.dist.get_backend() == 'nccl':
            t = t.to(getattr(self.rt, 'device', 'cuda'))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]

    def mean_scalars(self, values):
        """Mean over the ranks of a few host floats (validation costs / errors of equally sized shards); a collective."""
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if self.dist.get_backend() == 'nccl':
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) / self.world for v in t.cpu()]

    def shard(self, data, batch, pad_rng_seed=None, pad='cut', what='array'):
        """This rank's rows of a per-sample array under the minibatch layout of SURVEY.md section 8(e): the global minibatch k is rows
        [k*G*B, (k+1)*G*B) and rank r owns its contiguous slice [r*B, (r+1)*B).  How an array that is not a whole number of global
        minibatches is completed first:
          pad='cut'     dropped remainder (validation data: the reference drops it too, nettrainer.py:793) -- an array shorter than
                        one global minibatch would leave every rank without a single validation batch, which is refused;
          pad='random'  rows drawn by RandomState(pad_rng_seed) (alignData's pad_random rule, nettrainer.py:365-413; passing
                        pad_rng_seed alone selects it);
          pad='last'    the last row repeated (alignData with pad_random=False, nettrainer.py:404-411)."""
        import numpy as np
        data = np.asarray(data)
        G, B = self.world, int(batch)
        n, gb = data.shape[0], G * B
        if pad_rng_seed is not None:
            pad = 'random'
        if pad == 'cut':
            if n < gb:
                raise ValueError("data parallel: %s has %d samples, fewer than one global minibatch (%d ranks x %d): every rank's shard "
                                 "would be empty" % (what, n, G, B))
            data = data[:(n // gb) * gb]
        elif n % gb:
            fill_n = gb - n % gb
            if pad == 'random':
                rng = np.random.RandomState(pad_rng_seed)
                fill = np.stack([data[rng.randint(0, n)] for _ in range(fill_n)])
            elif pad == 'last':
                fill = np.repeat(data[n - 1:n], fill_n, axis=0)
            else:
                raise ValueError("pad must be 'cut', 'random' or 'last'")
            data = np.concatenate([data, fill], axis=0)
        return np.ascontiguousarray(data.reshape((-1, G, B) + data.shape[1:])[:, self.rank].reshape((-1,) + data.shape[1:]))

    def broadcast_store(self, store):
        """Replicas start from rank 0's parameters and running statistics."""
        self.broadcast(store.w)
        self.broadcast(store.nt)


def local_device_index():
    """The GPU this process owns under torchrun (LOCAL_RANK, wrapped when several ranks share the visible GPUs: the gloo control-flow
    tests put two ranks on one MI355X); None outside a multi-process launch or without a GPU."""
    import torch
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or not torch.cuda.is_available():
        return None
    return int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()


def bind_local_device():
    """Make the rank's GPU the current device.  Everything that creates device state -- TorchHipRuntime binds
    torch.cuda.current_device() and creates its streams there -- has to come after this; `hipdp.runtime.default_runtime()` calls it
    itself, so the importers / PCA / pose sampling of a main that run before the trainer exists land on the right GPU too."""
    idx = local_device_index()
    if idx is not None:
        import torch
        torch.cuda.set_device(idx)
    return idx


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun); returns (rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        backend = os.environ.get('DPP_DIST_BACKEND', backend)      # e.g. gloo: several ranks on ONE GPU to test the control flow
        # NO device_id in init_process_group: it makes torch create the RCCL communicator eagerly, BEFORE the engine's
        # streams exist, and the side stream then shares a hardware queue with the main stream (measured on the MI355X: the
        # two-stream step drops from 5.16 to 6.82 ms).  Created lazily by the first collective, after TorchHipRuntime() has
        # its streams, it does not disturb them.
        bind_local_device()
        dist.init_process_group(backend=backend)
    elif world > 1:
        bind_local_device()
    return rank, world


def check_runtime_device(rt):
    """A runtime created before the rank's GPU was selected holds buffers and streams on another GPU than the collectives and the
    launches of this process will use: refuse it instead of hanging in RCCL."""
    idx = local_device_index()
    dev = getattr(rt, 'device', None)
    if idx is not None and dev is not None and getattr(dev, 'index', idx) != idx:
        raise RuntimeError("data parallel: this rank owns cuda:%d but its runtime was created on %s -- call hipdp.parallel.init_from_env() "
                           "(or hipdp.runtime.default_runtime()) before creating device state" % (idx, dev))
