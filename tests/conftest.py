import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'deep-prior-pp_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    if not hasattr(config, 'workerinput') and getattr(config.option, 'markexpr', '') == 'not gpu':
        # build the emulator library once, in the controlling process, before any worker asks for it
        try:
            from tests.emu.emu_runtime import build_emulator
            build_emulator()
        except Exception as e:          # noqa: BLE001  (the tests that need it will report the failure themselves)
            sys.stderr.write("emulator build failed: %r\n" % (e,))


def pytest_cmdline_main(config):
    """The CPU tier (`-m "not gpu"`: the kernels run under the single-threaded SIMT emulator, ~7.5 min in one process) is spread
    over a few pytest-xdist workers when the plugin is there and no `-n` was given.  DPP_TEST_WORKERS=0 keeps it in one
    process.  The GPU tier always runs in ONE process: the round-end harness records the libraries that process loads."""
    if hasattr(config, 'workerinput') or os.environ.get('DPP_TEST_WORKERS', '') == '0':
        return None                     # an xdist worker runs this hook too: it must never become a controller itself
    if getattr(config.option, 'markexpr', '') != 'not gpu':
        return None
    if not config.pluginmanager.hasplugin('xdist') or getattr(config.option, 'numprocesses', None):
        return None
    n = int(os.environ.get('DPP_TEST_WORKERS', max(1, min(6, (os.cpu_count() or 1) - 2))))
    if n > 1:                           # what xdist's own (tryfirst) hook derives from `-n <n>`
        os.environ['DPP_TEST_WORKERS'] = '0'    # inherited by the popen workers: second guard against recursive spawning
        config.option.numprocesses = n
        config.option.dist = 'load'
        config.option.tx = ['popen'] * n
    return None
