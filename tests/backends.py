"""Backends for the kernel-parity tests: the SIMT emulator build (CPU tier) and the real gfx950 library
(`-m gpu` tier).  The same test bodies run on both; on the GPU box they call through the C ABI of
deep-prior-pp_amd/lib/libdpp_hip.so."""
import pytest

_rts = {}


def get_runtime(kind):
    if kind not in _rts:
        if kind == 'emu':
            from tests.emu.emu_runtime import EmuRuntime
            _rts[kind] = EmuRuntime()
        else:
            from hipdp.runtime import TorchHipRuntime
            _rts[kind] = TorchHipRuntime()
    return _rts[kind]


BACKENDS = [pytest.param('emu', id='emu'), pytest.param('hip', marks=pytest.mark.gpu, id='hip')]
