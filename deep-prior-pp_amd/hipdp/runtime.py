"""
Device-memory plumbing for the kernel library.  PyTorch-ROCm is used for allocation, host<->device
copies and the stream only; every computation goes through libdpp_hip.so.

`TorchHipRuntime` is the product runtime and refuses to start without a GPU (no CPU fallback).
The CPU-side kernel-logic tests inject `tests/emu/emu_runtime.EmuRuntime` (host memory + the emulator
build of the same kernel sources) through the same small interface.
"""
import os

import numpy as np

from . import lib as _lib


class Buffer(object):
    """A typed view of device memory: `ptr` (int), `shape`, `dtype`; `owner` (the allocation's tensor) keeps the allocation alive,
    `keep` whatever else has to outlive the kernels queued on it (the inputs of an asynchronous producer)."""
    __slots__ = ('ptr', 'shape', 'dtype', 'owner', 'rt', 'keep')

    def __init__(self, rt, ptr, shape, dtype, owner):
        self.rt, self.ptr, self.shape, self.dtype, self.owner = rt, int(ptr), tuple(int(s) for s in shape), np.dtype(dtype), owner
        self.keep = None

    @property
    def size(self):
        return int(np.prod(self.shape)) if len(self.shape) else 1

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def view(self, offset, shape, dtype=None):
        """A sub-view starting `offset` ELEMENTS in, with a new shape."""
        dt = np.dtype(dtype or self.dtype)
        n = int(np.prod(shape)) if len(shape) else 1
        assert offset >= 0 and offset * self.dtype.itemsize + n * dt.itemsize <= self.nbytes, "view out of range"
        b = Buffer(self.rt, self.ptr + offset * self.dtype.itemsize, shape, dt, self.owner)
        b.keep = self.keep
        return b

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        assert int(np.prod(shape)) == self.size
        b = Buffer(self.rt, self.ptr, shape, self.dtype, self.owner)
        b.keep = self.keep
        return b

    def get(self):
        return self.rt.download(self)

    def set(self, arr):
        self.rt.copy_in(self, arr)

    def zero(self):
        self.rt.zero(self)


class _AsyncRead(object):
    def __init__(self, host, ev, shape):
        self.host, self.ev, self.shape = host, ev, shape

    def get(self):
        self.ev.synchronize()
        return self.host.numpy().reshape(self.shape).copy()


class TorchHipRuntime(object):
    def __init__(self, device=None, lib_path=None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.DppError("no MI355X visible to PyTorch-ROCm: the DeepPrior++ HIP path needs a GPU "
                                "(there is no CPU fallback)")
        self.torch = torch
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.lib = _lib.load(lib_path)
        self.is_emulator = False
        self.has_side_stream = os.environ.get('DPP_NO_SIDE_STREAM', '0') != '1'
        self._side = torch.cuda.Stream(self.device)

    _TD = {'float32': 'float32', 'int32': 'int32', 'uint8': 'uint8', 'uint16': 'uint16', 'float64': 'float64', 'int64': 'int64'}

    def alloc(self, shape, dtype=np.float32, zero=True):
        t = self.torch
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        td = getattr(t, self._TD[np.dtype(dtype).name])
        ten = (t.zeros if zero else t.empty)(max(1, int(np.prod(shape))), dtype=td, device=self.device)
        return Buffer(self, ten.data_ptr(), shape, dtype, ten)

    def upload(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype or arr.dtype)
        b = self.alloc(arr.shape, arr.dtype, zero=False)
        self.copy_in(b, arr)
        return b

    def _tensor(self, buf):
        base = buf.owner
        off = (buf.ptr - base.data_ptr()) // base.element_size()
        if buf.dtype.itemsize != base.element_size():
            raise ValueError("dtype-changing views are not downloadable")
        return base[off:off + buf.size]

    def tensor(self, buf):
        """A flat torch tensor aliasing the buffer (for torch.distributed collectives)."""
        return self._tensor(buf)

    def copy_in(self, buf, arr):
        arr = np.ascontiguousarray(arr, dtype=buf.dtype).reshape(-1)
        assert arr.size == buf.size, (arr.shape, buf.shape)
        self._tensor(buf).copy_(self.torch.from_numpy(arr), non_blocking=False)

    def download(self, buf):
        return self._tensor(buf).cpu().numpy().reshape(buf.shape).copy()

    def zero(self, buf):
        self._tensor(buf).zero_()

    def read_async(self, buf):
        """Start a device -> pinned-host copy of `buf` behind the work queued so far and return a handle whose .get() waits
        for THAT copy only (an event), not for work queued afterwards: lets the host read step k's cost while step k+1 is
        already in flight."""
        t = self.torch
        src = self._tensor(buf)
        host = t.empty(src.shape, dtype=src.dtype, pin_memory=True)
        host.copy_(src, non_blocking=True)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.device))
        return _AsyncRead(host, ev, buf.shape)

    def copy(self, dst, src):
        self._tensor(dst).copy_(self._tensor(src))

    def download_async(self, buf, host=None):
        """Start a device -> page-locked host copy of `buf` on the COPY stream (behind the work queued so far on the current stream, but
        not in front of what is queued afterwards: the copy engine moves it while later kernels run).  Returns (handle, host tensor);
        handle.get() waits for that copy only.  `host`: a pinned tensor of a previous call to reuse.  The caller keeps `buf` unchanged
        until get() returns (checkpoints copy the live parameters into a staging buffer first)."""
        t = self.torch
        if getattr(self, '_copy', None) is None:
            self._copy = t.cuda.Stream(self.device)
        src = self._tensor(buf)
        if host is None or host.numel() != src.numel() or host.dtype != src.dtype:
            host = t.empty(src.shape, dtype=src.dtype, pin_memory=True)
        self._copy.wait_stream(t.cuda.current_stream(self.device))
        with t.cuda.stream(self._copy):
            host.copy_(src, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self._copy)
        return _AsyncRead(host, ev, buf.shape), host

    def pinned_like(self, arr):
        """An uninitialised page-locked host array of arr's shape / dtype (source of asynchronous uploads)."""
        t = self.torch
        ten = t.empty(tuple(arr.shape), dtype=t.from_numpy(np.empty(0, arr.dtype)).dtype, pin_memory=True)
        return ten.numpy()

    def upload_async(self, buf, host):
        """Start copying `host` (ideally page-locked: then the copy engine moves it while kernels run) into the first
        host.size elements of `buf` on a dedicated copy stream; returns a handle whose wait() makes the CURRENT stream wait for
        the copy (no host block).  The copy itself first waits for the work queued so far, which may still read `buf`."""
        t = self.torch
        if getattr(self, '_copy', None) is None:
            self._copy = t.cuda.Stream(self.device)
        cur = t.cuda.current_stream(self.device)
        src = t.from_numpy(np.ascontiguousarray(host, dtype=buf.dtype).reshape(-1))
        dst = self._tensor(buf)[:src.numel()]
        self._copy.wait_stream(cur)
        with t.cuda.stream(self._copy):
            dst.copy_(src, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self._copy)

        class _Copy(object):
            def __init__(self, ev, keep):
                self.ev, self.keep = ev, keep

            def wait(self_inner):
                t.cuda.current_stream(self.device).wait_event(self_inner.ev)

            def synchronize(self_inner):
                """Block the HOST until the copy has read its source (before the host buffer is rewritten)."""
                self_inner.ev.synchronize()
        return _Copy(ev, src)

    def staged_upload(self, buf, host, free_event=None):
        """Host array -> device buffer `buf` on the COPY stream, ordered only behind `free_event` (the point on the main stream after which
        `buf` may be overwritten: the caller recorded it behind the last reader) -- not behind everything queued on the main stream, as
        upload_async orders it.  Returns the event to make the main stream wait for before it reads `buf`.  What lets computeOutput move
        batch i + 1 over PCIe while batch i is being evaluated."""
        t = self.torch
        if getattr(self, '_copy', None) is None:
            self._copy = t.cuda.Stream(self.device)
        src = t.from_numpy(np.ascontiguousarray(host, dtype=buf.dtype).reshape(-1))
        dst = self._tensor(buf)[:src.numel()]
        if free_event is not None:
            self._copy.wait_event(free_event)
        with t.cuda.stream(self._copy):
            dst.copy_(src, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(self._copy)
        ev.keep = src                      # the host tensor outlives the copy: the caller holds the event until it has been waited for
        return ev

    def record_event(self):
        """An event at the current end of the main stream."""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def wait_event(self, ev):
        self.torch.cuda.current_stream(self.device).wait_event(ev)

    @property
    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    @property
    def side_stream(self):
        return self._side.cuda_stream

    def side_wait_main(self):
        self._side.wait_stream(self.torch.cuda.current_stream(self.device))

    def main_wait_side(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self._side)

    def capture(self, fn, warmup=1):
        """Capture everything `fn` launches (on the current stream and on the forked side stream) into a hipGraph and
        return it; `graph.replay()` re-issues the whole step with one host call."""
        t = self.torch
        cur = t.cuda.current_stream(self.device)
        s = t.cuda.Stream(self.device)
        s.wait_stream(cur)
        with t.cuda.stream(s):
            for _ in range(warmup):
                fn()
        cur.wait_stream(s)
        t.cuda.synchronize(self.device)
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g):
            fn()
        return g

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


_default = None


def default_runtime():
    """The process-wide product runtime (one per process / GPU)."""
    global _default
    if _default is None:
        from . import parallel
        parallel.bind_local_device()          # under torchrun: the rank's own GPU, whoever asks for the runtime first
        _default = TorchHipRuntime()
    return _default


def set_default_runtime(rt):
    global _default
    _default = rt
