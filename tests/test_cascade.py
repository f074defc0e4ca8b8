"""The CoM-refinement cascade as ONE device plan (hipdp/cascade.py; /root/reference/src/util/handdetector.py:413-440, 634-676):
frame -> crop -> centre of mass -> stretched crop -> ScaleNet offset -> refined centre -> final crop (+ labels) -- stage by stage
against the oracle's restatement, on the emulator and on the GPU; then BASELINE.json's config 5 at its full per-GPU size on the GPU:
128 frames of 640x480 -> refinement -> 256x256 crops -> one ResNet train step, fp32 and bf16, as one plan."""
import numpy as np
import pytest

from data.importers import ICVLImporter, NYUImporter
from hipdp import engine
from hipdp import runtime as R
from hipdp.cascade import CascadeCropper
from net.scalenet import ScaleNet, ScaleNetParams
from oracle import augment as A
from oracle import nets
from tests.backends import BACKENDS, get_runtime
from tests.test_scalenet import make as make_scalenet


def _oracle_net_forward(onet, P64):
    def fwd(ins):
        B = onet['batch_size']
        rep = [np.repeat(a.astype(np.float64), B, axis=0) for a in ins]          # computeOutput pads by repeating the last sample
        out, _ = nets.forward(onet, P64, rep, train=False)
        return out[:1]
    return fwd


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('dsize', [128, 256])
def test_cascade_stages_match_oracle(backend, dsize):
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    di = ICVLImporter('../data/ICVL/')
    cam = A.Camera.icvl()
    nb, B = 2, 5                                           # two full net batches and an overlapping last chunk
    net, onet, P = make_scalenet(rt, nb)
    net.setDeterministic()
    cube = (250., 250., 250.)
    frames, coms = A.synthetic_frames(np.random.RandomState(3), B, cam, 240, 320, cube)
    J = 16
    rng = np.random.RandomState(4)
    gt3d = (np.stack([cam.jointImgTo3D(c) for c in coms])[:, None, :] + rng.normal(0, 30., (B, J, 3))).astype(np.float32)
    comp, _ = np.linalg.qr(rng.normal(size=(J * 3, 30)))

    class Proj(object):
        mean_ = rng.normal(0, 0.05, J * 3)
        components_ = comp.T

    out_y = rt.alloc((B, 30), zero=False)
    cc = CascadeCropper(rt, di, net, B, 240, 320, dsize=dsize, gt3d=rt.upload(gt3d), J=J, proj=Proj, out_y=out_y)
    crops, Ms, com2 = cc(frames, coms, np.tile(np.float32(cube), (B, 1)))
    com1, net_in, y = cc.com1.get(), cc.crop_r.get(), out_y.get()
    fwd = _oracle_net_forward(onet, nets.cast_params(P, np.float64))
    for i in range(B):
        d, lo, hi = A.detector_preprocess(frames[i])
        ref, M, c2, c1, rz = A.crop_area_3d_refined(d, coms[i], cube, cam, abs(cam.fx), abs(cam.fy), lo, hi, fwd, dsize=(dsize, dsize),
                                                    com2_override=com2[i])
        assert np.array_equal(com1[i], c1), (i, com1[i], c1)                     # centre of mass of the first window: exact
        assert np.array_equal(net_in[i], A.normalize_crop(rz, c1[2], cube[2])), i    # what the net sees: exact
        np.testing.assert_allclose(com2[i], c2, rtol=0, atol=2e-4)               # float32 net vs float64 oracle: < 1e-3 px / mm
        assert np.array_equal(crops[i], A.normalize_crop(ref, com2[i][2], cube[2])), i   # the final crop around the device's centre: exact
        np.testing.assert_allclose(Ms[i], M, rtol=1e-6, atol=1e-4)
        lab = (gt3d[i] - cam.jointImgTo3D(com2[i])) / np.float32(cube[2] / 2.)
        np.testing.assert_allclose(y[i], A.pca_transform(lab.reshape(1, -1).astype('f8'), Proj.mean_, Proj.components_)[0], rtol=0, atol=3e-6)
    # the host path of the same call (HandDetector.cropArea3D with a refineNet, one frame at a time) agrees with the plan
    from util.handdetector import HandDetector
    hd = HandDetector(frames[1].copy(), abs(di.fx), abs(di.fy), importer=di, refineNet=net)
    c, M, com = hd.cropArea3D(com=coms[1], size=cube, dsize=(dsize, dsize), docom=True)
    np.testing.assert_allclose(com, com2[1], rtol=0, atol=1e-4)
    d = A.normalize_crop(c, com[2], cube[2])
    assert (d != crops[1]).mean() < 1e-3                    # (a last-bit difference of the centre may move a window bound)


def _synthetic_nyu(B, seed=7):
    cam = A.Camera.nyu()
    cube = (300., 300., 300.)
    frames, coms = A.synthetic_frames(np.random.RandomState(seed), B, cam, 480, 640, cube)
    return cam, cube, frames, coms


@pytest.mark.gpu
@pytest.mark.parametrize('bf16', [False, True])
def test_config5_cascade_is_one_plan_at_full_size(bf16):
    """configs[4] on one GPU: 128 NYU-sized frames -> ScaleNet refinement -> 256x256 crops + 30-D labels -> ResNet (69 M parameters)
    train step, ONE plan.  Size-independent properties: the plan's crops / labels equal the staged path (crop_frames, computeOutput,
    the host formula), the cost equals the train engine run on those crops alone, and a second run of the plan is bit-identical."""
    from net.resnet import ResNet, ResNetParams
    from util.handdetector import crop_frames
    rt = get_runtime('hip')
    R.set_default_runtime(rt)
    B, S, J = 128, 256, 14
    di = NYUImporter('../data/NYU/')
    cam, cube, frames, coms = _synthetic_nyu(B)
    rnet = ScaleNet(np.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, nChan=1, wIn=128, hIn=128, batchSize=64, resizeFactor=2,
                                                                         numJoints=1, nDims=3))
    rnet.setDeterministic()
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=B, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=bf16)
    rng = np.random.RandomState(4)
    gt3d = (np.stack([cam.jointImgTo3D(c) for c in coms])[:, None, :] + rng.normal(0, 40., (B, J, 3))).astype(np.float32)
    comp, _ = np.linalg.qr(rng.normal(size=(J * 3, 30)))

    class Proj(object):
        mean_ = rng.normal(0, 0.05, J * 3)
        components_ = comp.T

    cc = CascadeCropper(rt, di, rnet, B, 480, 640, dsize=S, gt3d=rt.upload(gt3d), J=J, proj=Proj, out=eng.x_in.buf.reshape(B, S, S), out_y=eng.y_in)
    cc.frames.set(frames)
    cc.com0.set(coms)
    cc.cube.set(np.tile(np.float32(cube), (B, 1)))
    eng.set_lr(1e-3)
    w0 = eng.store.w.get().copy()
    plan = eng.step_plan(before=cc.plan)
    plan.run(rt)
    rt.synchronize()
    cost = float(eng.cost.get()[0])
    x, y, com2 = eng.x_in.buf.get().reshape(B, S, S), eng.y_in.get(), cc.com2.get()
    assert np.isfinite(cost) and np.isfinite(x).all() and x.min() >= -1.0 - 1e-6 and x.max() <= 1.0 + 1e-6
    # staged path: docom crop (stretched to 128) -> computeOutput -> host formula -> crop at 256
    cubes = np.tile(np.float32(cube), (B, 1))
    _, _, c1 = crop_frames(frames, coms, cubes, abs(di.fx), abs(di.fy), 128, normalize=False, docom=True, return_com=True)
    assert np.array_equal(c1, cc.com1.get())
    rz, _ = crop_frames(frames, c1, cubes, abs(di.fx), abs(di.fy), 128, normalize=True, stretch=True)
    off = rnet.computeOutput(nets.scalenet_inputs(rz[:, None]))
    for i in range(B):
        n3 = off[i] * np.float32(cube[2] / 2.) + di.jointImgTo3D(c1[i])
        np.testing.assert_allclose(com2[i], di.joint3DToImg(n3), rtol=0, atol=1e-4)
    x2, _ = crop_frames(frames, com2, cubes, abs(di.fx), abs(di.fy), S, normalize=True)
    assert np.array_equal(x, x2)
    lab = (gt3d - np.stack([di.jointImgTo3D(c) for c in com2])[:, None, :]) / np.float32(cube[2] / 2.)
    np.testing.assert_allclose(y, (lab.reshape(B, -1).astype('f8') - Proj.mean_) @ Proj.components_.T, rtol=0, atol=3e-6)
    # the train step saw exactly these crops: the same engine on them alone, from the same weights, gives the same cost and weights
    w1 = eng.store.w.get().copy()
    assert np.abs(w1 - w0).max() > 1e-4
    eng.store.w.set(w0)
    eng.reset_optimizer(1e-3)
    c2 = eng.train_step(x2[:, None], y, 1e-3)
    assert c2 == cost, (c2, cost)
    assert any(('bf16' in (l.meta or {}).get('kernel', '')) for _, l in eng.all_launches()) == bf16
