#!/usr/bin/env python3
"""
DeepPrior++ pose regressor on the ICVL hand-pose dataset with the MI355X path -- the Python-3 counterpart of the reference's driver for
BASELINE.json configs[2] (what /root/reference/src/main_icvl_posereg_embedding.py does, written against the class API of
deep-prior-pp_amd/): the un-rotated training sub-sequence ('0') -> crops -> 30-D PCA prior on augmented poses -> ResNet / PoseRegNet
with the fused online augmentation (rotation / centre jitter / none as in the reference's script; --aug-modes com,rot,sc,none adds the
cube scaling BASELINE configs[2] lists) -> prior layer -> test_seq_1.

    python examples/main_icvl_posereg_embedding.py --data ../data/ICVL/ [--batch 256] [--aug-modes com,rot,sc,none] [--epochs 100]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))

import numpy  # noqa: E402

from data.dataset import ICVLDataset  # noqa: E402
from data.importers import ICVLImporter  # noqa: E402
from net.hiddenlayer import HiddenLayer, HiddenLayerParams  # noqa: E402
from net.poseregnet import PoseRegNet, PoseRegNetParams  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402
from util.handpose_evaluation import DeviceHandposeEvaluation  # noqa: E402
from util.pcaprior import DevicePCA, sample_random_poses_device  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', default='../data/ICVL/')
    ap.add_argument('--net', choices=['resnet', 'poseregnet'], default='resnet')
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--batch', type=int, default=128, help='crops per GPU (BASELINE configs[2]: 256)')
    ap.add_argument('--embedding', type=int, default=30)
    ap.add_argument('--aug-modes', default='com,rot,none')
    ap.add_argument('--train-subseq', default='0', help="comma-separated training sub-sequences ('0' = the un-rotated recordings)")
    ap.add_argument('--prior-poses', type=float, default=1e6)
    ap.add_argument('--dp', action='store_true', help='data parallel: started under torchrun, one rank per GPU')
    ap.add_argument('--out', default='./eval/icvl_posereg_embedding')
    ap.add_argument('--cache', default='./cache/')
    args = ap.parse_args(argv)
    if args.dp:
        from hipdp import parallel
        parallel.init_from_env()          # selects this rank's GPU: before the importers / PCA / trainer create device state
    os.makedirs(args.out, exist_ok=True)
    rng = numpy.random.RandomState(23455)
    aug_modes = args.aug_modes.split(',')

    di = ICVLImporter(args.data, cacheDir=args.cache)
    train = di.loadSequence('train', args.train_subseq.split(','), shuffle=True, rng=rng, docom=False)
    test = di.loadSequence('test_seq_1', docom=False)
    train_data, train_gt3D = ICVLDataset([train]).imgStackDepthOnly('train')
    train_cube = numpy.asarray([train.config['cube']] * train_data.shape[0], 'float32')
    train_com = numpy.asarray([f.com for f in train.data], 'float32')
    train_M = numpy.asarray([f.T for f in train.data], 'float32')
    train_gt3Dcrop = numpy.asarray([f.gt3Dcrop for f in train.data], 'float32')
    test_data, test_gt3D = ICVLDataset([test]).imgStackDepthOnly('test_seq_1')
    J = train_gt3D.shape[1]
    print("{} training crops, {} test crops, {} joints, augmentation modes {}".format(train_data.shape[0], test_data.shape[0], J, aug_modes))

    pca = DevicePCA(n_components=args.embedding)
    pca.fit(sample_random_poses_device(di, rng, train_gt3Dcrop, train_com, train_cube, int(args.prior_poses), aug_modes).reshape((-1, J * 3)))
    train_embed = pca.transform(train_gt3D.reshape((-1, J * 3))).astype('float32')
    val_embed = pca.transform(test_gt3D.reshape((-1, J * 3))).astype('float32')

    size = train_data.shape[2]
    Net, Params = (ResNet, ResNetParams) if args.net == 'resnet' else (PoseRegNet, PoseRegNetParams)
    net = Net(rng, cfgParams=Params(type=0, nChan=train_data.shape[1], wIn=size, hIn=size, batchSize=args.batch, numJoints=1, nDims=args.embedding))
    p = PoseRegNetTrainerParams()
    p.batch_size = args.batch
    p.learning_rate = 0.001
    p.weightreg_factor = 0.0
    p.force_macrobatch_reload = True
    p.para_augment = True
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': aug_modes, 'proj': pca,
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    trainer = PoseRegNetTrainer(net, p, rng, args.out, dp='env' if args.dp else None)
    trainer.setData(train_data, train_embed, test_data, val_embed)
    trainer.addStaticData({'val_data_y3D': test_gt3D})
    trainer.addStaticData({'pca_data': pca.components_.astype('float32'), 'mean_data': pca.mean_.astype('float32')})
    trainer.addManagedData({'train_data_cube': train_cube, 'train_data_com': train_com, 'train_data_M': train_M, 'train_gt3Dcrop': train_gt3Dcrop})
    trainer.compileFunctions()
    costs, _, val_errs = trainer.train(n_epochs=args.epochs)
    writer = trainer.dp is None or trainer.dp.rank == 0
    if writer:
        net.save(os.path.join(args.out, 'net_{}.pkl'.format(net.__class__.__name__)))

    prior = HiddenLayer(rng, net.layers[-1].output, HiddenLayerParams(inputDim=(args.batch, args.embedding), outputDim=(args.batch, J * 3),
                                                                        activation=None), layerNum=len(net.layers))
    prior.W.set_value(pca.components_.astype('float32'))
    prior.b.set_value(pca.mean_.astype('float32'))
    net.layers.append(prior)
    net.output = prior.output
    net.cfgParams.numJoints, net.cfgParams.nDims = J, 3
    net.cfgParams.outputDim = (args.batch, J * 3)
    if writer:
        net.save(os.path.join(args.out, 'network_prior.pkl'))

    net.setDeterministic()
    com = numpy.asarray([f.com for f in test.data], 'float32')
    half = test.config['cube'][2] / 2.
    joints = net.computeOutput(test_data).reshape((-1, J, 3)) * half + com[:, None, :]
    gt = numpy.asarray([f.gt3Dorig for f in test.data], 'float32')
    ev = DeviceHandposeEvaluation(gt, joints)
    print("test_seq_1: mean joint error {:.2f} mm, max {:.2f} mm".format(ev.getMeanError(), ev.getMaxError()))
    print("per joint: {}".format([round(float(ev.getJointMeanError(j)), 2) for j in range(J)]))
    return costs, {'test_seq_1': (float(ev.getMeanError()), float(ev.getMaxError()))}


if __name__ == '__main__':
    main()
