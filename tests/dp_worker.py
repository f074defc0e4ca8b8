"""Worker of tests/test_data_parallel.py: one data-parallel rank (gloo, emulator runtime) computing the gradients of its
shard of a global batch with sync-BN, then one full train step.  Usage: dp_worker.py <out.npz> <sync_bn 0|1> [weight_decay]
DPP_WORKER_BACKEND=nccl: the same on the MI355X through RCCL (the GPU tier runs it with a world of ONE rank: every collective of
the step -- early FC1 bucket from the side stream, its wait, the second bucket, the sync-BN all-gathers, the cost reduction and
the parameter broadcast -- goes through RCCL and its stream ordering, and must leave the single-process result untouched)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'deep-prior-pp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


import numpy as np  # noqa: E402

from hipdp import heuristics  # noqa: E402
from hipdp import engine, parallel  # noqa: E402

heuristics.EARLY_BUCKET_MIN = 1 << 18      # the test net's FC layers (1 M weights) take the overlapped early-bucket path
from net.resnet import ResNet, ResNetParams  # noqa: E402
from oracle import nets  # noqa: E402


def main():
    out, sync = sys.argv[1], bool(int(sys.argv[2]))
    wd = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    if os.environ.get('DPP_WORKER_BACKEND') == 'nccl':
        import torch
        import torch.distributed as dist
        from hipdp.runtime import TorchHipRuntime
        torch.cuda.set_device(0)
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)     # also for a world of one
        assert dist.get_backend() == 'nccl'
        rt = TorchHipRuntime()
    elif os.environ.get('DPP_WORKER_BACKEND') == 'hip-gloo':      # several gloo ranks sharing the one MI355X (host-staged collectives)
        import torch
        from hipdp.runtime import TorchHipRuntime
        torch.cuda.set_device(0)
        rank, world = parallel.init_from_env('gloo')
        rt = TorchHipRuntime()
    else:
        from tests.emu.emu_runtime import EmuRuntime
        rank, world = parallel.init_from_env('gloo')
        rt = EmuRuntime()
    B = int(os.environ.get('DPP_WORKER_BATCH', '4'))
    net = ResNet(np.random.RandomState(23455 + rank), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=B, numJoints=1, nDims=30))
    dp = parallel.DataParallel(rt, sync_bn=sync)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), dp=dp, weight_decay=wd)   # broadcasts rank 0's parameters
    rng = np.random.RandomState(99)
    x = nets.synthetic_crops(rng, B * world, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (B * world, 30)).astype(np.float32)
    xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
    cost, _ = eng.cost_and_grads(xs, ys)
    assert eng._early_slice is not None and eng._early_work[0] is not None      # the early bucket is in flight
    eng.allreduce_grads()
    assert eng._early_work[0] is None
    G = {}
    for i, l in enumerate(net.layers):
        for s, p in enumerate(l.params):
            G['g_%d_%d' % (i, s)] = eng.store.read_grad(p)
    G['cost'] = np.array([cost])
    G['global_cost'] = np.array([eng.global_cost()])            # collective: the reference's cost of the global minibatch
    eng.train_step(xs, ys, 1e-3)
    last = net.layers[-1]
    G['w_last'] = last.W.get_value()
    G['bn_mean'] = [l for l in net.layers if l.__class__.__name__ == 'BatchNormLayer'][0].mean.get_value()
    np.savez(out, **G)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
