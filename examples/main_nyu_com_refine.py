#!/usr/bin/env python3
"""
The CoM-refinement net (ScaleNet) on the NYU hand-pose dataset with the MI355X path -- the Python-3 counterpart of the reference's
driver for this workload (what /root/reference/src/main_nyu_com_refine.py does, written against the class API of deep-prior-pp_amd/):
the training sequence twice -- cropped around the annotated hand centre and around the centre of mass -- as 128x128 crops plus their
64x64 and 32x32 centre crops, ScaleNet (three towers, 3-D output = offset of the crop joint from the crop centre) trained by
ScaleNetTrainer with online device augmentation, error of the refined centre on the test sequences against the unrefined centre.

    python examples/main_nyu_com_refine.py --data ../data/NYU/ [--epochs 100] [--train-frames N] [--out ./eval/nyu_com_refine]

The checkpoint it writes (<out>/net_ScaleNet.pkl) is what examples/main_nyu_posereg_embedding.py --refine takes: the importer then crops
every frame through the device cascade crop -> centre of mass -> ScaleNet -> crop (hipdp/cascade.py).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))

import numpy  # noqa: E402

from data.dataset import NYUDataset  # noqa: E402
from data.importers import NYUImporter  # noqa: E402
from net.scalenet import ScaleNet, ScaleNetParams  # noqa: E402
from trainer.scalenettrainer import ScaleNetTrainer, ScaleNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector  # noqa: E402
from util.handpose_evaluation import DeviceHandposeEvaluation  # noqa: E402
from util.helpers import shuffle_many_inplace  # noqa: E402


def centre(x, factor):
    """The centre crop of side / factor of a stack [N][C][H][W] (the second and third input of the net)."""
    h, w = x.shape[2] // factor, x.shape[3] // factor
    y0, x0 = x.shape[2] // 2 - h // 2, x.shape[3] // 2 - w // 2
    return numpy.ascontiguousarray(x[:, :, y0:y0 + h, x0:x0 + w])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', default='../data/NYU/')
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--train-frames', type=float, default=float('inf'))
    ap.add_argument('--out', default='./eval/nyu_com_refine')
    ap.add_argument('--cache', default='./cache/')
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    rng = numpy.random.RandomState(23455)
    aug_modes = ['com', 'rot', 'none']

    di = NYUImporter(args.data, cacheDir=args.cache)
    train_seqs = [di.loadSequence('train', Nmax=args.train_frames, shuffle=True, rng=rng, docom=False)._replace(name='train_gt'),
                  di.loadSequence('train', Nmax=args.train_frames, shuffle=True, rng=rng, docom=True)._replace(name='train_com')]
    test_seqs = [di.loadSequence('test_1', docom=True), di.loadSequence('test_2', docom=True)]

    ds = NYUDataset(train_seqs)
    stacks = [ds.imgStackDepthOnly(s.name) for s in train_seqs]
    train_data = numpy.concatenate([d for d, _ in stacks])
    train_gt3D = numpy.concatenate([g for _, g in stacks])
    train_cube = numpy.concatenate([numpy.asarray([s.config['cube']] * len(s.data), 'float32') for s in train_seqs])
    train_com = numpy.concatenate([numpy.asarray([f.com for f in s.data], 'float32') for s in train_seqs])
    train_M = numpy.concatenate([numpy.asarray([f.T for f in s.data], 'float32') for s in train_seqs])
    shuffle_many_inplace([train_data, train_gt3D, train_com, train_cube, train_M], random_state=rng)
    tds = NYUDataset(test_seqs)
    test_sets = [tds.imgStackDepthOnly(s.name) for s in test_seqs]
    val_data, val_gt3D = test_sets[0]
    cj = di.crop_joint_idx
    print("{} training crops ({} frames, each around the annotated centre and around the centre of mass), {} / {} test crops".format(
        train_data.shape[0], len(train_seqs[0].data), test_sets[0][0].shape[0], test_sets[1][0].shape[0]))

    size = train_data.shape[2]
    net = ScaleNet(rng, cfgParams=ScaleNetParams(type=1, nChan=train_data.shape[1], wIn=size, hIn=size, batchSize=args.batch, resizeFactor=2,
                                                 numJoints=1, nDims=3))
    p = ScaleNetTrainerParams()
    p.use_early_stopping = False
    p.batch_size = args.batch
    p.learning_rate = 0.0005
    p.weightreg_factor = 0.0001
    p.force_macrobatch_reload = True
    p.para_augment = True
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': aug_modes,
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    trainer = ScaleNetTrainer(net, p, rng, args.out)
    trainer.setData(train_data, train_gt3D[:, cj, :], val_data, val_gt3D[:, cj, :])
    trainer.addStaticData({'val_data_x1': centre(val_data, 2), 'val_data_x2': centre(val_data, 4)})
    trainer.addManagedData({'train_data_x1': centre(train_data, 2), 'train_data_x2': centre(train_data, 4)})
    trainer.addManagedData({'train_data_com': train_com, 'train_data_cube': train_cube, 'train_data_M': train_M, 'train_gt3D': train_gt3D})
    trainer.compileFunctions()
    costs, _, val_errs = trainer.train(n_epochs=args.epochs)
    net.save(os.path.join(args.out, 'net_ScaleNet.pkl'))

    # the refined centre (net output * cube/2 + crop centre) against the crop joint, and the unrefined centre for comparison
    net.setDeterministic()
    results = {}
    for seq, (data, _) in zip(test_seqs, test_sets):
        gt = numpy.asarray([f.gt3Dorig[cj] for f in seq.data], 'float32').reshape((-1, 1, 3))
        com3D = numpy.asarray([f.com for f in seq.data], 'float32')                # the crop centres, camera space (mm)
        out = net.computeOutput([data, centre(data, 2), centre(data, 4)]).reshape((-1, 1, 3))
        refined = out * (seq.config['cube'][2] / 2.) + com3D[:, None, :]
        e_ref = DeviceHandposeEvaluation(gt, refined).getMeanError()
        e_com = DeviceHandposeEvaluation(gt, com3D.reshape((-1, 1, 3))).getMeanError()
        print("{}: crop joint {:.2f} mm from the refined centre, {:.2f} mm from the centre of mass".format(seq.name, e_ref, e_com))
        results[seq.name] = (float(e_ref), float(e_com))
    return costs, results


if __name__ == '__main__':
    main()
