"""TEST INFRASTRUCTURE ONLY: a runtime with the same interface as hipdp.runtime.TorchHipRuntime whose
"device" memory is host memory and whose library is the emulator build of the kernel sources
(tests/emu/_build/libdpp_emu.so).  Lets the CPU-only test tier drive the real kernel code."""
import os
import subprocess

import numpy as np

from hipdp import lib as _lib
from hipdp.runtime import Buffer

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU_LIB = os.path.join(ROOT, 'tests', 'emu', '_build', 'libdpp_emu.so')


def build_emulator():
    """make emu (a no-op when the library is newer than the kernel sources), under a file lock: several test processes may ask."""
    import fcntl
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(EMU_LIB), '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(['make', '-s', '-j8', '-C', os.path.join(ROOT, 'deep-prior-pp_amd', 'csrc'), 'emu'])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return EMU_LIB


class EmuRuntime(object):
    def __init__(self):
        build_emulator()
        self.lib = _lib.load(EMU_LIB)
        self.stream = None
        self.is_emulator = True
        self.has_side_stream = False

    def alloc(self, shape, dtype=np.float32, zero=True):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = max(1, int(np.prod(shape)))
        raw = np.zeros(n * np.dtype(dtype).itemsize + 64, np.uint8)
        off = (-raw.ctypes.data) % 64
        arr = raw[off:off + n * np.dtype(dtype).itemsize].view(dtype)
        if not zero:
            arr.view(np.uint8)[:] = 0x7F          # poison: 0x7F7F7F7F = 3.4e38 as float32 (0x7F for uint8)
        return Buffer(self, arr.ctypes.data, shape, dtype, (raw, arr))

    def upload(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype or arr.dtype)
        b = self.alloc(arr.shape, arr.dtype, zero=False)
        self.copy_in(b, arr)
        return b

    def _arr(self, buf):
        raw, arr = buf.owner
        off = buf.ptr - arr.ctypes.data
        return arr.view(np.uint8)[off:off + buf.nbytes].view(buf.dtype)

    def tensor(self, buf):
        import torch
        return torch.from_numpy(self._arr(buf))

    def copy_in(self, buf, arr):
        arr = np.ascontiguousarray(arr, dtype=buf.dtype).reshape(-1)
        assert arr.size == buf.size, (arr.shape, buf.shape)
        self._arr(buf)[:] = arr

    def download(self, buf):
        return self._arr(buf).reshape(buf.shape).copy()

    def zero(self, buf):
        self._arr(buf)[:] = 0

    def read_async(self, buf):
        val = self.download(buf)            # the emulator has already executed everything queued

        class _Done(object):
            def get(self_inner):
                return val
        return _Done()

    def copy(self, dst, src):
        self._arr(dst)[:] = self._arr(src)

    def download_async(self, buf, host=None):
        return self.read_async(buf), None

    # the stream-ordered upload interface of NetBase._compute_output_pipelined: everything executes in issue order here, so the events
    # are placeholders -- what the CPU tier exercises is the control flow (staging slots, padding of the last batch, late output reads)
    def staged_upload(self, buf, host, free_event=None):
        self.copy_in(buf, np.asarray(host).reshape(buf.shape) if np.asarray(host).size == buf.size else host)
        return object()

    def record_event(self):
        return object()

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass
