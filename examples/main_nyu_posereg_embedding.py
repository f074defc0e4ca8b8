#!/usr/bin/env python3
"""
DeepPrior++ pose regressor on the NYU hand-pose dataset with the MI355X path -- the Python-3 counterpart of the reference's
driver for this workload (what /root/reference/src/main_nyu_posereg_embedding.py does, written against the class API of
deep-prior-pp_amd/, not transcribed): sequences -> cropped / normalised stacks -> 30-D PCA prior fitted on augmented poses ->
ResNet (or PoseRegNet) regressing the embedding -> PoseRegNetTrainer with online device augmentation -> append the prior as a
linear layer -> joints of the test sequences -> mean / max joint error in mm.

    python examples/main_nyu_posereg_embedding.py --data ../data/NYU/ [--net resnet|poseregnet] [--epochs 100]
                                                  [--refine eval/com_refine/net_ScaleNet.pkl] [--train-frames N] [--out ./eval/nyu]
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/main_nyu_posereg_embedding.py --data ... --dp
        (one rank per MI355X: every rank keeps its slice of each global minibatch of 8 x 128 crops, gradients all-reduced over RCCL)

Dataset layout (as the reference expects it): <data>/train, <data>/test_1, <data>/test_2, each with depth_1_%07d.png + joint_data.mat; the
importer caches the cropped sequences as pickles under --cache with the reference's file names.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))

import numpy  # noqa: E402

from data.dataset import NYUDataset  # noqa: E402
from data.importers import NYUImporter  # noqa: E402
from net.hiddenlayer import HiddenLayer, HiddenLayerParams  # noqa: E402
from net.poseregnet import PoseRegNet, PoseRegNetParams  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402
from util.handpose_evaluation import DeviceHandposeEvaluation  # noqa: E402
from util.pcaprior import DevicePCA, sample_random_poses_device  # noqa: E402


def stack(seqs, dataset):
    """Crops, normalised labels and the per-sample geometry the online augmentation needs, over all sequences."""
    data, labels = zip(*[dataset.imgStackDepthOnly(s.name) for s in seqs])
    com = numpy.concatenate([numpy.asarray([f.com for f in s.data], 'float32') for s in seqs])
    cube = numpy.concatenate([numpy.asarray([s.config['cube']] * len(s.data), 'float32') for s in seqs])
    M = numpy.concatenate([numpy.asarray([f.T for f in s.data], 'float32') for s in seqs])
    gt3Dcrop = numpy.concatenate([numpy.asarray([f.gt3Dcrop for f in s.data], 'float32') for s in seqs])
    return numpy.concatenate(data), numpy.concatenate(labels), com, cube, M, gt3Dcrop


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', default='../data/NYU/')
    ap.add_argument('--net', choices=['resnet', 'poseregnet'], default='resnet')
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--batch', type=int, default=128, help='crops per GPU')
    ap.add_argument('--embedding', type=int, default=30)
    ap.add_argument('--train-frames', type=float, default=float('inf'))
    ap.add_argument('--refine', default=None, help='checkpoint of a trained ScaleNet: crop centres are refined by it (com_refine cascade)')
    ap.add_argument('--dp', action='store_true', help='data parallel: started under torchrun, one rank per GPU')
    ap.add_argument('--out', default='./eval/nyu_posereg_embedding')
    ap.add_argument('--cache', default='./cache/')
    ap.add_argument('--prior-poses', type=float, default=1e6, help='augmented poses the PCA prior is fitted on')
    args = ap.parse_args(argv)
    if args.dp:
        from hipdp import parallel
        parallel.init_from_env()          # selects this rank's GPU: before the importers / PCA / trainer create device state
    os.makedirs(args.out, exist_ok=True)
    rng = numpy.random.RandomState(23455)
    aug_modes = ['com', 'rot', 'none']

    di = NYUImporter(args.data, cacheDir=args.cache)
    docom = False
    if args.refine is not None:
        from net.scalenet import ScaleNet, ScaleNetParams
        refine = ScaleNet(numpy.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, nChan=1, wIn=128, hIn=128, batchSize=64,
                                                                                   resizeFactor=2, numJoints=1, nDims=3))
        refine.load(args.refine)
        refine.setDeterministic()
        di.refineNet = refine                      # loadSequence(docom=True) then crops every frame through the device cascade
        docom = True
    train_seqs = [di.loadSequence('train', Nmax=args.train_frames, shuffle=True, rng=rng, docom=docom)]
    test_seqs = [di.loadSequence('test_1', docom=docom), di.loadSequence('test_2', docom=docom)]

    train_data, train_gt3D, train_com, train_cube, train_M, train_gt3Dcrop = stack(train_seqs, NYUDataset(train_seqs))
    test_sets = [stack([s], NYUDataset([s])) for s in test_seqs]
    val_data, val_gt3D = test_sets[0][0], test_sets[0][1]
    J = train_gt3D.shape[1]
    print("{} training crops, {} / {} test crops, {} joints".format(train_data.shape[0], test_sets[0][0].shape[0], test_sets[1][0].shape[0], J))

    # the pose prior: PCA of 1e6 augmented training poses (label-space augmentation on the device, eigen-decomposition on the device)
    pca = DevicePCA(n_components=args.embedding)
    pca.fit(sample_random_poses_device(di, rng, train_gt3Dcrop, train_com, train_cube, int(args.prior_poses), aug_modes).reshape((-1, J * 3)))
    train_embed = pca.transform(train_gt3D.reshape((-1, J * 3))).astype('float32')
    val_embed = pca.transform(val_gt3D.reshape((-1, J * 3))).astype('float32')

    size = train_data.shape[2]
    Net, Params = (ResNet, ResNetParams) if args.net == 'resnet' else (PoseRegNet, PoseRegNetParams)
    net = Net(rng, cfgParams=Params(type=0, nChan=train_data.shape[1], wIn=size, hIn=size, batchSize=args.batch, numJoints=1, nDims=args.embedding))

    p = PoseRegNetTrainerParams()
    p.batch_size = args.batch
    p.learning_rate = 0.001
    p.weightreg_factor = 0.0
    p.force_macrobatch_reload = True            # a freshly augmented copy of the resident training set every epoch
    p.para_augment = True
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': aug_modes, 'proj': pca,
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    trainer = PoseRegNetTrainer(net, p, rng, args.out, dp='env' if args.dp else None)
    trainer.setData(train_data, train_embed, val_data, val_embed)
    trainer.addStaticData({'val_data_y3D': val_gt3D})
    trainer.addStaticData({'pca_data': pca.components_.astype('float32'), 'mean_data': pca.mean_.astype('float32')})
    trainer.addManagedData({'train_data_cube': train_cube, 'train_data_com': train_com, 'train_data_M': train_M, 'train_gt3Dcrop': train_gt3Dcrop})
    trainer.compileFunctions()
    costs, _, val_errs = trainer.train(n_epochs=args.epochs)
    writer = trainer.dp is None or trainer.dp.rank == 0
    if writer:
        net.save(os.path.join(args.out, 'net_{}.pkl'.format(net.__class__.__name__)))

    # the prior as the last layer: joints = embedding . components + mean
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
    results = {}
    for seq, (data, gt3D, com, cube, _, _) in zip(test_seqs, test_sets):
        joints = net.computeOutput(data).reshape((-1, J, 3)) * (cube[:, 2] / 2.)[:, None, None] + com[:, None, :]
        gt = gt3D * (cube[:, 2] / 2.)[:, None, None] + com[:, None, :]
        ev = DeviceHandposeEvaluation(gt, joints)
        print("{}: mean joint error {:.2f} mm, max {:.2f} mm".format(seq.name, ev.getMeanError(), ev.getMaxError()))
        results[seq.name] = (float(ev.getMeanError()), float(ev.getMaxError()))
    return costs, results


if __name__ == '__main__':
    main()
