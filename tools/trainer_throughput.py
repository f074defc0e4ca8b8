#!/usr/bin/env python3
"""What a user of the class API gets: crops/s of PoseRegNetTrainer.train() itself (the reference's epoch loop: per-epoch re-augmentation of
the resident training set, one train_model call + cost read-back per minibatch, validation at the end of every epoch) on the bs128
ResNet, against bench.py's plan-only figure.      python tools/trainer_throughput.py [n_crops] [epochs]"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from tools import synth  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
EPOCHS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, S, J = 128, 128, 14
di, imgs, coms, cubes, Ms, gts, pca_mean, pca_comp = synth.crop_db(N, S, J)
rng = np.random.RandomState(23455)


class Proj(object):
    mean_, components_ = pca_mean, pca_comp

    @staticmethod
    def transform(x):
        return (np.asarray(x, np.float64) - pca_mean) @ pca_comp.T


labels = (gts / (cubes[:, 2] / 2.)[:, None, None]).astype(np.float32)
embed = Proj.transform(labels.reshape(N, -1)).astype(np.float32)
net = ResNet(rng, cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=B, numJoints=1, nDims=30))
p = PoseRegNetTrainerParams()
p.batch_size = B
p.learning_rate = 1e-3
p.weightreg_factor = 0.0        # as the pose-regression mains set it (/root/reference/src/main_nyu_posereg_embedding.py:104); the default 0.001 adds two
                                # launches over all parameters (0.17 ms per step)
p.force_macrobatch_reload = True
p.para_augment = True
p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': ['com', 'rot', 'none'], 'proj': Proj,
                                                         'hd': HandDetector(imgs[0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
os.makedirs('/tmp/dpp_trainer_throughput', exist_ok=True)
tr = PoseRegNetTrainer(net, p, rng, '/tmp/dpp_trainer_throughput')
tr.setData(imgs[:, None], embed, imgs[:B, None], embed[:B])
tr.addStaticData({'val_data_y3D': labels[:B]})
tr.addStaticData({'pca_data': pca_comp, 'mean_data': pca_mean})
tr.addManagedData({'train_data_cube': cubes, 'train_data_com': coms, 'train_data_M': Ms, 'train_gt3Dcrop': gts})
tr.compileFunctions()


def run(epochs):
    sink = io.StringIO()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink):            # the per-minibatch cost lines (kept: the reference prints them too)
        costs, _, _ = tr.train(n_epochs=epochs)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, len(costs)


run(1)
dt, steps = run(EPOCHS)
print('PoseRegNetTrainer.train(): %d epochs x %d minibatches of %d in %.3f s = %.3f ms per minibatch, %.0f crops/s (incl. per-epoch '
      're-augmentation, snapshots, cost read-back, validation)' % (EPOCHS, steps // EPOCHS, B, dt, dt / steps * 1e3, steps * B / dt))
if os.environ.get('DPP_TRAINER_PROFILE') == '1':
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    run(EPOCHS)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
if os.environ.get('DPP_TRAINER_PROFILE') == '2':
    def timed(name, fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        print('%-40s %.3f ms per call' % (name, (time.perf_counter() - t0) / n * 1e3))
    te = tr.train_engine
    timed('train_engine.run_step_plans()', lambda i: te.run_step_plans())
    timed('train_model_async (no resolve)', lambda i: tr.train_model_async(i % 8, 1e-3))
    timed('train_model (sync cost)', lambda i: tr.train_model(i % 8, 1e-3))
    timed('loadMiniBatch', lambda i: tr.loadMiniBatch(i % 8))
    hs = []
    timed('train_model_async + read previous', lambda i: (hs.append(tr.train_model_async(i % 8, 1e-3)), hs[-2].get() if len(hs) > 1 else None))
    timed('rt.read_async(cost).get()', lambda i: tr.rt.read_async(te.cost).get())
if os.environ.get('DPP_TRAINER_PROFILE') == '3':
    from hipdp import engine
    def timed(name, fn, n=20):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        print('%-60s %.3f ms per call' % (name, (time.perf_counter() - t0) / n * 1e3))
    te = tr.train_engine
    print('trainer engine: launches fwd %d loss %d bwd %d upd %d; loss_cfg %r wd %r optimizer %r' % (
        len(te.fwd.launches()), len(te.lossplan.launches()), len(te.bwd.launches()), len(te.upd.launches()), tr.loss_cfg,
        getattr(te, 'weight_decay', None), getattr(getattr(tr, 'optimizer', None), 'rule', None)))
    timed('trainer engine step', lambda i: te.run_step_plans())
    timed('  fwd', lambda i: te.fwd.run(tr.rt))
    timed('  loss', lambda i: te.lossplan.run(tr.rt))
    timed('  bwd', lambda i: te.bwd.run(tr.rt))
    timed('  upd', lambda i: te.upd.run(tr.rt))
    net2 = ResNet(np.random.RandomState(1), cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=B, numJoints=1, nDims=30))
    e2 = engine.CompiledNet(net2, train=True, runtime=tr.rt, loss=dict(kind='embedding'))
    e2.set_lr(1e-3)
    timed('fresh engine (bench.py style) step', lambda i: e2.run_step_plans())
    timed('trainer engine step again', lambda i: te.run_step_plans())
