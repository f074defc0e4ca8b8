"""Worker of tests/test_data_parallel.py (trainer level): the main-script flow of main_*_posereg_embedding.py -- PoseRegNetTrainer.setData /
addStaticData / addManagedData / compileFunctions / train -- as ONE data-parallel rank, or as the single process it is compared with.
Usage: dp_trainer_worker.py <out.npz> <sync_bn 0|1> <global_batch> <epochs>
  DPP_WORKER_BACKEND = emu   gloo ranks on the SIMT emulator (CPU tier)
                       hip   the real kernels on cuda:0; several gloo ranks may share the one MI355X (DPP_DIST_BACKEND=gloo)
  WORLD_SIZE unset or 1: a single process WITHOUT dp on the whole global batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'deep-prior-pp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
from sklearn.decomposition import PCA  # noqa: E402

from data.importers import ICVLImporter  # noqa: E402
from hipdp import heuristics  # noqa: E402
from hipdp import engine, parallel  # noqa: E402
from hipdp import runtime as R  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402

heuristics.EARLY_BUCKET_MIN = 1 << 18      # the test net's FC1 (1 M weights) takes the overlapped early-bucket path


def main():
    out, sync, GB, epochs = sys.argv[1], bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
    from oracle import augment as A
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('DPP_WORKER_BACKEND', 'emu') == 'hip':
        import torch
        from hipdp.runtime import TorchHipRuntime
        torch.cuda.set_device(0)
        if world > 1:
            parallel.init_from_env('gloo' if os.environ.get('DPP_DIST_BACKEND') == 'gloo' else 'nccl')
        rt = TorchHipRuntime()
    else:
        from tests.emu.emu_runtime import EmuRuntime
        if world > 1:
            parallel.init_from_env('gloo')
        rt = EmuRuntime()
    R.set_default_runtime(rt)
    dp = parallel.DataParallel(rt, sync_bn=sync) if world > 1 else None
    B = GB // world
    size, J, E, n_train, n_val = 32, 16, 8, 3 * GB - 3, 2 * GB
    di = ICVLImporter('../data/ICVL/')
    cam = A.Camera.icvl()
    rng = np.random.RandomState(23455)
    imgs, train_com, train_cube, train_M, train_gt3Dcrop = A.synthetic_augment_inputs(np.random.RandomState(1), n_train, cam, cube=(250.,) * 3,
                                                                                     joints=J, dsize=size)
    vimgs, _, val_cube, _, val_gt3Dcrop = A.synthetic_augment_inputs(np.random.RandomState(2), n_val, cam, cube=(250.,) * 3, joints=J, dsize=size)
    train_data, val_data = imgs[:, None], vimgs[:, None]
    train_gt3D = (train_gt3Dcrop / (train_cube[:, 2] / 2.)[:, None, None]).astype('float32')
    val_gt3D = (val_gt3Dcrop / (val_cube[:, 2] / 2.)[:, None, None]).astype('float32')
    pca = PCA(n_components=E)
    pca.fit(HandDetector.sampleRandomPoses(di, rng, train_gt3Dcrop, train_com, train_cube, 200, ['com', 'rot', 'none']).reshape((-1, J * 3)))
    train_embed = pca.transform(train_gt3D.reshape((-1, J * 3))).astype('float32')
    val_embed = pca.transform(val_gt3D.reshape((-1, J * 3))).astype('float32')
    net = ResNet(rng, cfgParams=ResNetParams(type=0, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=1, nDims=E))
    p = PoseRegNetTrainerParams()
    p.batch_size = B
    p.learning_rate = 0.001
    p.weightreg_factor = 0.001
    p.force_macrobatch_reload = True
    p.use_early_stopping = False
    p.validation_frequency = 2
    p.snapshot_last = 1
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': ['com', 'rot', 'none'],
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di),
                                                             'proj': pca}}
    sub = os.path.join(os.path.dirname(out), 'snap_w%d_r%d' % (world, dp.rank if dp else 0))
    os.makedirs(sub, exist_ok=True)
    tr = PoseRegNetTrainer(net, p, rng, sub, dp=dp)
    tr.setData(train_data, train_embed, val_data, val_embed)
    tr.addStaticData({'val_data_y3D': val_gt3D})
    tr.addStaticData({'pca_data': pca.components_.astype('float32'), 'mean_data': pca.mean_.astype('float32')})
    tr.addManagedData({'train_data_cube': train_cube, 'train_data_com': train_com, 'train_data_M': train_M.astype('float32'),
                       'train_gt3Dcrop': train_gt3Dcrop})
    tr.compileFunctions()
    res = dict(shard_x=tr.train_data_xDB.copy(), shard_com=tr.train_data_comDB.copy(), shard_val=tr.val_data_xDB.copy())
    costs, _, val = tr.train(n_epochs=epochs)
    res.update(costs=np.asarray(costs), val=np.asarray(val, np.float64), aug_x=tr.train_data_x.get_value(), aug_y=tr.train_data_y.get_value(),
               w=tr.train_engine.store.w.get(), snapshot=np.array([os.path.isfile(os.path.join(sub, 'net_last.pkl'))]))
    np.savez(out, **res)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
