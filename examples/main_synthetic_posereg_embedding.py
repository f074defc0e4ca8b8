#!/usr/bin/env python3
"""
The training script of the reference, /root/reference/src/main_nyu_posereg_embedding.py:44-180, run end to end on the
MI355X against SYNTHETIC data (there is no dataset in the build environment): same calls, same order -- sequences ->
cropped / normalised stacks -> PCA prior on sampled poses -> PoseRegNet (or ResNet) -> PoseRegNetTrainer with online
augmentation -> train -> append the prior layer -> computeOutput -> mean joint error in mm.  The only substitution is the
data source: `SyntheticImporter.loadSequence` renders depth frames of a 16-joint "hand" (spheres on a 6-D linear pose
manifold) with ICVL's camera and crops them with HandDetector's device crop, where the reference's importers read PNGs.

    python examples/main_synthetic_posereg_embedding.py [--net poseregnet|resnet] [--frames 2048] [--epochs 8]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))        # instead of the reference's src/ 

import numpy  # noqa: E402

from data.basetypes import DepthFrame, NamedImgSequence  # noqa: E402
from data.dataset import Dataset  # noqa: E402
from data.importers import ICVLImporter  # noqa: E402
from net.hiddenlayer import HiddenLayer, HiddenLayerParams  # noqa: E402
from net.poseregnet import PoseRegNet, PoseRegNetParams  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams  # noqa: E402
from util.handdetector import HandDetector, crop_frames  # noqa: E402
from util.handpose_evaluation import DeviceHandposeEvaluation as HandposeEvaluation  # noqa: E402  (same metrics, computed on the device)
from util.pcaprior import DevicePCA, sample_random_poses_device  # noqa: E402


class SyntheticImporter(ICVLImporter):
    """ICVL's camera and conventions; frames are rendered instead of read."""

    def __init__(self, seed=7):
        super(SyntheticImporter, self).__init__('../data/SYNTH/')
        r = numpy.random.RandomState(seed)
        self.mean_pose = r.uniform(-60, 60, (self.numJoints, 3))
        self.mean_pose[self.crop_joint_idx] = 0.                       # the crop joint is the CoM
        self.basis = r.normal(0, 1, (6, self.numJoints, 3)) * 14.
        self.basis[:, self.crop_joint_idx] = 0.

    def loadSequence(self, seqName, Nmax=1024, shuffle=False, rng=None, docom=False):
        r = numpy.random.RandomState(abs(hash(seqName)) % (2 ** 31))
        W, H = self.depth_map_size
        cube = self.default_cubes['train']
        z = r.normal(0, 1, (Nmax, 6))
        gt3Dcrop = (self.mean_pose[None] + numpy.einsum('nk,kjd->njd', z, self.basis)).astype('float32')
        com3D = numpy.stack([r.uniform(-120, 120, Nmax), r.uniform(-80, 80, Nmax), r.uniform(380, 650, Nmax)], axis=1).astype('float32')
        gt3Dorig = gt3Dcrop + com3D[:, None]
        frames = numpy.zeros((Nmax, H, W), 'float32')
        for i in range(Nmax):                                          # depth of the nearest sphere surface per pixel
            d = frames[i]
            for j in range(self.numJoints):
                X, Y, Z = gt3Dorig[i, j]
                u, v = X / Z * self.fx + self.ux, Y / Z * self.fy + self.uy
                rad = 14. * self.fx / Z
                x0, x1 = max(0, int(u - rad)), min(W, int(u + rad) + 2)
                y0, y1 = max(0, int(v - rad)), min(H, int(v + rad) + 2)
                if x0 >= x1 or y0 >= y1:
                    continue
                yy, xx = numpy.mgrid[y0:y1, x0:x1].astype('float32')
                r2 = ((xx - u) ** 2 + (yy - v) ** 2) / (rad * rad)
                dz = (Z - 14. * numpy.sqrt(numpy.clip(1. - r2, 0., 1.))).astype('float32')
                win = d[y0:y1, x0:x1]
                take = (r2 < 1.) & ((win == 0) | (dz < win))
                win[take] = dz[take]
        coms = numpy.stack([self.joint3DToImg(c) for c in com3D]).astype('float32')
        cubes = numpy.tile(numpy.asarray(cube, 'float32'), (Nmax, 1))
        crops, Ms = crop_frames(frames, coms, cubes, self.fx, self.fy, 128, normalize=False)     # cropArea3D on the device
        data = []
        for i in range(Nmax):
            # like the reference's importers (importers.py:388-396): .com is the crop centre in metric 3-D
            data.append(DepthFrame(crops[i], gt3Dorig[i], None, Ms[i], gt3Dorig[i], gt3Dcrop[i], com3D[i], '', seqName, 'right', {}))
        if shuffle and rng is not None:
            rng.shuffle(data)
        return NamedImgSequence(seqName, data, {'cube': cube})


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', choices=['poseregnet', 'resnet'], default='poseregnet')
    ap.add_argument('--frames', type=int, default=2048)
    ap.add_argument('--epochs', type=int, default=8)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--embedding', type=int, default=30)
    ap.add_argument('--prior-poses', type=int, default=20000)
    ap.add_argument('--out', default='./eval/SYNTH_EMB')
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    rng = numpy.random.RandomState(23455)

    print("create data")
    aug_modes = ['com', 'rot', 'none']
    di = SyntheticImporter()
    t0 = time.time()
    Seq1 = di.loadSequence('train', Nmax=args.frames, shuffle=True, rng=rng)
    Seq2 = di.loadSequence('test_1', Nmax=max(2 * args.batch, args.frames // 8))
    print("  rendered and cropped {} frames in {:.1f} s".format(len(Seq1.data) + len(Seq2.data), time.time() - t0))
    trainDataSet, testDataSet = Dataset([Seq1]), Dataset([Seq2])
    train_data, train_gt3D = trainDataSet.imgStackDepthOnly('train')
    train_data_cube = numpy.asarray([Seq1.config['cube']] * train_data.shape[0], dtype='float32')
    train_data_com = numpy.asarray([d.com for d in Seq1.data], dtype='float32')
    train_data_M = numpy.asarray([da.T for da in Seq1.data], dtype='float32')
    train_gt3Dcrop = numpy.asarray([d.gt3Dcrop for d in Seq1.data], dtype='float32')
    val_data, val_gt3D = testDataSet.imgStackDepthOnly('test_1')
    print("data size: {}Mb".format(train_data.nbytes // (1024 * 1024)))

    # convert data to embedding
    # (the reference: sklearn PCA on HandDetector.sampleRandomPoses(...), main_nyu_posereg_embedding.py:86-88; here the sampling,
    # the scatter matrix and its eigen-decomposition run on the device, the samples never visit the host)
    pca = DevicePCA(n_components=args.embedding)
    pca.fit(sample_random_poses_device(di, rng, train_gt3Dcrop, train_data_com, train_data_cube, args.prior_poses, aug_modes, keep_on_device=True))
    train_gt3D_embed = pca.transform(train_gt3D.reshape((train_gt3D.shape[0], -1))).astype('float32')
    val_gt3D_embed = pca.transform(val_gt3D.reshape((val_gt3D.shape[0], -1))).astype('float32')

    print("create network")
    batchSize = args.batch
    Net, NetParams = (PoseRegNet, PoseRegNetParams) if args.net == 'poseregnet' else (ResNet, ResNetParams)
    poseNetParams = NetParams(type=0, nChan=train_data.shape[1], wIn=train_data.shape[3], hIn=train_data.shape[2], batchSize=batchSize,
                              numJoints=1, nDims=train_gt3D_embed.shape[1])
    poseNet = Net(rng, cfgParams=poseNetParams)

    poseNetTrainerParams = PoseRegNetTrainerParams()
    poseNetTrainerParams.batch_size = batchSize
    poseNetTrainerParams.learning_rate = 0.001
    poseNetTrainerParams.weightreg_factor = 0.0
    poseNetTrainerParams.force_macrobatch_reload = True
    poseNetTrainerParams.para_augment = True
    poseNetTrainerParams.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': aug_modes,
                                                                                'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di),
                                                                                'proj': pca}}
    print("setup trainer")
    poseNetTrainer = PoseRegNetTrainer(poseNet, poseNetTrainerParams, rng, args.out)
    poseNetTrainer.setData(train_data, train_gt3D_embed, val_data, val_gt3D_embed)
    poseNetTrainer.addStaticData({'val_data_y3D': val_gt3D})
    poseNetTrainer.addStaticData({'pca_data': pca.components_.astype('float32'), 'mean_data': pca.mean_.astype('float32')})
    poseNetTrainer.addManagedData({'train_data_cube': train_data_cube, 'train_data_com': train_data_com,
                                   'train_data_M': train_data_M, 'train_gt3Dcrop': train_gt3Dcrop})
    poseNetTrainer.compileFunctions(compileDebugFcts=False)

    t0 = time.time()
    train_res = poseNetTrainer.train(n_epochs=args.epochs)
    dt = time.time() - t0
    train_costs = train_res[0]
    nb = len(train_costs)
    print("trained {} minibatches of {} in {:.2f} s ({:.0f} crops/s incl. augmentation and validation); cost {:.4f} -> {:.4f}".format(
        nb, batchSize, dt, nb * batchSize / dt, float(numpy.mean(train_costs[:4])), float(numpy.mean(train_costs[-4:]))))
    poseNet.save("{}/net_SYNTH_EMB.pkl".format(args.out))

    # add prior to network
    cfg = HiddenLayerParams(inputDim=(batchSize, train_gt3D_embed.shape[1]), outputDim=(batchSize, int(numpy.prod(train_gt3D.shape[1:]))),
                            activation=None)
    pcalayer = HiddenLayer(rng, poseNet.layers[-1].output, cfg, layerNum=len(poseNet.layers))
    pcalayer.W.set_value(pca.components_.astype('float32'))
    pcalayer.b.set_value(pca.mean_.astype('float32'))
    poseNet.layers.append(pcalayer)
    poseNet.output = pcalayer.output
    poseNet.cfgParams.numJoints = train_gt3D.shape[1]
    poseNet.cfgParams.nDims = train_gt3D.shape[2]
    poseNet.cfgParams.outputDim = pcalayer.cfgParams.outputDim
    poseNet.save("{}/network_prior.pkl".format(args.out))

    print("Testing ...")
    gt3D = [j.gt3Dorig for j in Seq2.data]
    jts = poseNet.computeOutput(val_data)
    joints = [jts[i].reshape((-1, 3)) * (Seq2.config['cube'][2] / 2.) + Seq2.data[i].com for i in range(val_data.shape[0])]
    hpe = HandposeEvaluation(gt3D, joints)
    mean_pose_err = HandposeEvaluation(gt3D, [di.mean_pose + s.com for s in Seq2.data]).getMeanError()
    print("Mean error: {:.2f}mm, max error: {:.2f}mm  (predicting the mean pose: {:.2f}mm)".format(hpe.getMeanError(), hpe.getMaxError(), mean_pose_err))
    return train_costs, (float(hpe.getMeanError()), float(hpe.getMaxError()), float(mean_pose_err))


if __name__ == '__main__':
    main()
