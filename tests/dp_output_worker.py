"""Worker of tests/test_data_parallel.py::test_compute_output_shards_batches_*: one rank of a data-parallel test-time forward.
Usage: dp_output_worker.py <out.npz>; DPP_WORKER_BACKEND = (unset: gloo on the emulator) | hip-gloo (gloo ranks sharing the MI355X)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'deep-prior-pp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from hipdp import parallel  # noqa: E402
from hipdp import runtime as R  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from oracle import nets  # noqa: E402


def main():
    out = sys.argv[1]
    rank, world = parallel.init_from_env('gloo')
    if os.environ.get('DPP_WORKER_BACKEND') == 'hip-gloo':
        from hipdp.runtime import TorchHipRuntime
        rt = TorchHipRuntime()
    else:
        from tests.emu.emu_runtime import EmuRuntime
        rt = EmuRuntime()
    R.set_default_runtime(rt)
    B, n = int(os.environ.get('DPP_WORKER_BATCH', '4')), int(os.environ.get('DPP_WORKER_SAMPLES', '11'))
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=1, wIn=32, hIn=32, batchSize=B, numJoints=14, nDims=3))
    x = nets.synthetic_crops(np.random.RandomState(5), n, 32, 32, np.float32)
    net.setDeterministic()
    dp = parallel.DataParallel(rt) if world > 1 else None
    if os.environ.get('DPP_WORKER_DIVERGE') == '1':
        # what per-GPU BatchNorm statistics leave behind after a data-parallel training run: every rank's running mean / inv_std
        # followed its own shards (the trained parameters are replicated).  The single process plays rank 0.
        net.computeOutput(x[:1], dp=False)                # compiles the net: the parameters move into the device store
        prng = np.random.RandomState(100 + rank)
        for l in net.layers:
            for p_ in l.params_nontrained:
                v = p_.get_value()
                p_.set_value((v + prng.normal(0, 0.05, v.shape) * (1.0 if 'mean' in p_.name else 0.2)).astype(np.float32))
    bn0 = [l for l in net.layers if l.__class__.__name__ == 'BatchNormLayer'][0]
    before = bn0.mean.get_value().copy() if os.environ.get('DPP_WORKER_DIVERGE') == '1' else None
    res = dict(out=net.computeOutput(x, dp=dp))
    if before is not None:
        # the sharded call evaluated with rank 0's statistics and put this rank's own back (round 6)
        res['bn_mean_before'], res['bn_mean_after'] = before, bn0.mean.get_value().copy()
    if dp is not None:
        net.dp = dp                                   # what a data-parallel trainer attaches: the plain call shards as well
        res['out_attr'] = net.computeOutput(x)
        res['few'] = net.computeOutput(x[:3])         # fewer batches than ranks: rank 1 only receives
    else:
        res['few'] = net.computeOutput(x[:3])
    np.savez(out, **res)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
