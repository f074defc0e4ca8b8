#!/usr/bin/env python3
"""
Leave-one-subject-out cross-validation of the DeepPrior++ pose regressor on MSRA15 with the MI355X path -- the Python-3 counterpart of
the reference's driver for BASELINE.json configs[3] (what /root/reference/src/main_msra15_posereg_embedding_crossval.py does, written
against the class API of deep-prior-pp_amd/): for every subject P0..P8, train on the other eight (crops, 30-D PCA prior fitted on
augmented poses, online device augmentation), append the prior as a linear layer, test on the held-out subject; at the end the
per-fold and the overall mean joint error.

    python examples/main_msra15_posereg_embedding_crossval.py --data ../data/MSRA15/ [--net resnet|poseregnet] [--epochs 100] [--folds 0,1,..]
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/main_msra15_posereg_embedding_crossval.py --data ... --dp
        (one rank per MI355X, 128 crops each: a global minibatch of 1 024; the gradients are all-reduced over RCCL)

The folds are independent of each other; --folds selects a subset so that several nodes can share the nine.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))

import numpy  # noqa: E402

from data.dataset import MSRA15Dataset  # noqa: E402
from data.importers import MSRA15Importer  # noqa: E402
from net.hiddenlayer import HiddenLayer, HiddenLayerParams  # noqa: E402
from net.poseregnet import PoseRegNet, PoseRegNetParams  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402
from util.handpose_evaluation import DeviceHandposeEvaluation  # noqa: E402
from util.helpers import shuffle_many_inplace  # noqa: E402
from util.pcaprior import DevicePCA, sample_random_poses_device  # noqa: E402


def stack(seqs):
    ds = MSRA15Dataset(seqs, localCache=False)
    data, labels = zip(*[ds.imgStackDepthOnly(s.name) for s in seqs])
    per_frame = lambda get: numpy.concatenate([numpy.asarray([get(f, s) for f in s.data], 'float32') for s in seqs])     # noqa: E731
    return (numpy.concatenate(data), numpy.concatenate(labels), per_frame(lambda f, s: f.com), per_frame(lambda f, s: s.config['cube']),
            per_frame(lambda f, s: f.T), per_frame(lambda f, s: f.gt3Dcrop))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', default='../data/MSRA15/')
    ap.add_argument('--subjects', default='P0,P1,P2,P3,P4,P5,P6,P7,P8')
    ap.add_argument('--folds', default=None, help='comma-separated indices into --subjects to run (default: all)')
    ap.add_argument('--net', choices=['resnet', 'poseregnet'], default='resnet')
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--batch', type=int, default=128, help='crops per GPU')
    ap.add_argument('--embedding', type=int, default=30)
    ap.add_argument('--prior-poses', type=float, default=1e6)
    ap.add_argument('--dp', action='store_true', help='data parallel: started under torchrun, one rank per GPU')
    ap.add_argument('--out', default='./eval/msra15_posereg_embedding_cv')
    ap.add_argument('--cache', default='./cache/')
    args = ap.parse_args(argv)
    if args.dp:
        from hipdp import parallel
        parallel.init_from_env()          # selects this rank's GPU: before the importers / PCA / trainer create device state
    os.makedirs(args.out, exist_ok=True)
    rng = numpy.random.RandomState(23455)
    aug_modes = ['com', 'rot', 'none']

    di = MSRA15Importer(args.data, cacheDir=args.cache)
    seqs = [di.loadSequence(name, shuffle=True, rng=rng, docom=False) for name in args.subjects.split(',')]
    folds = range(len(seqs)) if args.folds is None else [int(f) for f in args.folds.split(',')]
    results, costs = {}, {}
    all_gt, all_joints = [], []
    for icv in folds:
        train_seqs, test_seq = [s for i, s in enumerate(seqs) if i != icv], seqs[icv]
        print("fold {}: training on {}, testing on {}".format(icv, ' '.join(s.name for s in train_seqs), test_seq.name))
        train_data, train_gt3D, train_com, train_cube, train_M, train_gt3Dcrop = stack(train_seqs)
        shuffle_many_inplace([train_data, train_gt3D, train_cube, train_com, train_gt3Dcrop, train_M], random_state=rng)
        test_data, test_gt3D, test_com, test_cube, _, _ = stack([test_seq])
        J = train_gt3D.shape[1]

        pca = DevicePCA(n_components=args.embedding)
        pca.fit(sample_random_poses_device(di, rng, train_gt3Dcrop, train_com, train_cube, int(args.prior_poses), aug_modes).reshape((-1, J * 3)))
        train_embed = pca.transform(train_gt3D.reshape((-1, J * 3))).astype('float32')
        val_embed = pca.transform(test_gt3D.reshape((-1, J * 3))).astype('float32')

        size = train_data.shape[2]
        Net, Params = (ResNet, ResNetParams) if args.net == 'resnet' else (PoseRegNet, PoseRegNetParams)
        net = Net(rng, cfgParams=Params(type=0, nChan=train_data.shape[1], wIn=size, hIn=size, batchSize=args.batch, numJoints=1,
                                        nDims=args.embedding))
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
        trainer.addManagedData({'train_data_cube': train_cube, 'train_data_com': train_com, 'train_data_M': train_M, 'train_gt3Dcrop': train_gt3Dcrop})
        trainer.compileFunctions(compileDebugFcts=False)
        costs[test_seq.name] = trainer.train(n_epochs=args.epochs)[0]
        writer = trainer.dp is None or trainer.dp.rank == 0
        if writer:
            net.save(os.path.join(args.out, 'net_{}.pkl'.format(icv)))

        prior = HiddenLayer(rng, net.layers[-1].output, HiddenLayerParams(inputDim=(args.batch, args.embedding), outputDim=(args.batch, J * 3),
                                                                            activation=None), layerNum=len(net.layers))
        prior.W.set_value(pca.components_.astype('float32'))
        prior.b.set_value(pca.mean_.astype('float32'))
        net.layers.append(prior)
        net.output = prior.output
        net.cfgParams.numJoints, net.cfgParams.nDims = J, 3
        net.cfgParams.outputDim = (args.batch, J * 3)
        if writer:
            net.save(os.path.join(args.out, 'network_prior_{}.pkl'.format(icv)))

        net.setDeterministic()
        half = (test_cube[:, 2] / 2.)[:, None, None]
        joints = net.computeOutput(test_data).reshape((-1, J, 3)) * half + test_com[:, None, :]
        gt = test_gt3D * half + test_com[:, None, :]
        ev = DeviceHandposeEvaluation(gt, joints)
        print("fold {} ({}): mean joint error {:.2f} mm, max {:.2f} mm".format(icv, test_seq.name, ev.getMeanError(), ev.getMaxError()))
        results[test_seq.name] = (float(ev.getMeanError()), float(ev.getMaxError()))
        all_gt.append(gt)
        all_joints.append(joints)
    ev = DeviceHandposeEvaluation(numpy.concatenate(all_gt), numpy.concatenate(all_joints))
    print("{} folds: mean joint error {:.2f} mm, max {:.2f} mm".format(len(results), ev.getMeanError(), ev.getMaxError()))
    results['all'] = (float(ev.getMeanError()), float(ev.getMaxError()))
    return costs, results


if __name__ == '__main__':
    main()
