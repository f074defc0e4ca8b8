"""Synthetic depth data of the shapes BASELINE.json's configs name, built on the PRODUCT's importer /
HandDetector classes: what bench.py and the tools feed the kernels.  (The oracle has its own generators for the parity tests;
nothing here imports oracle/.)"""
import numpy as np


def importer_of(name):
    from data.importers import ICVLImporter, MSRA15Importer, NYUImporter
    return {'nyu': NYUImporter, 'icvl': ICVLImporter, 'msra': MSRA15Importer}[name]('../data/' + name.upper() + '/')


def blob_crops(rng, n, h=128, w=128):
    """Normalised crops: far-plane background +1.0 (dataset.py:98-100) and a random ellipse covering 25-45 % of the pixels with
    values U(-1, 0.6)."""
    x = np.ones((n, h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        area = rng.uniform(0.25, 0.45) * h * w
        ar = rng.uniform(0.6, 1.6)
        a = np.sqrt(area * ar / np.pi)
        b = area / (np.pi * a)
        cy, cx = h / 2. + rng.uniform(-8, 8), w / 2. + rng.uniform(-8, 8)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        msk = (u / a) ** 2 + (v / b) ** 2 <= 1.0
        vals = rng.uniform(-1, 0.6, size=(h, w)).astype(np.float32)
        x[i][msk] = vals[msk]
    return x


def crop_db(n, size, J=14, seed=23455, dataset='nyu', cube=(300., 300., 300.)):
    """A device-resident training set in the trainer's layout: normalised crops + per-sample (com3D, cube, M, gt3Dcrop) + a 30-D
    PCA prior.  Returns (importer, imgs, coms, cubes, Ms, gts, pca_mean, pca_components)."""
    from util.handdetector import HandDetector
    rng = np.random.RandomState(seed)
    di = importer_of(dataset)
    hd = HandDetector(np.ones((8, 8), np.float32) * 500., abs(di.fx), abs(di.fy), importer=di)
    imgs = blob_crops(rng, n, size, size)
    coms = np.zeros((n, 3), np.float32)
    cubes = np.tile(np.asarray(cube, np.float32), (n, 1))
    Ms = np.zeros((n, 3, 3), np.float32)
    gts = np.zeros((n, J, 3), np.float32)
    for i in range(n):
        com2d = np.array([rng.uniform(60, 260), rng.uniform(40, 200), rng.uniform(300, 600)], np.float32)
        coms[i] = di.jointImgTo3D(com2d)
        Ms[i] = hd.comToTransform(di.joint3DToImg(coms[i]), cube, (size, size))
        gts[i] = np.clip(rng.normal(0, 35., (J, 3)), -cube[0] / 2., cube[0] / 2.)
    pca_mean = rng.normal(0, 0.05, J * 3).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(J * 3, 30)))
    return di, imgs, coms, cubes, Ms, gts, pca_mean, q.T.astype(np.float32)


def depth_frames(rng, n, di, H=480, W=640, cube=(300., 300., 300.), J=14):
    """Full depth frames (mm): a far wall with holes (0 = not defined), a hand-sized blob around a centre, some pixels nearer /
    farther than the cube; per frame the annotated centre (u, v, d) and J joints (mm, camera space) scattered around it."""
    frames = np.zeros((n, H, W), np.float32)
    coms = np.zeros((n, 3), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    fx = abs(float(di.fx))
    for i in range(n):
        d = rng.uniform(450., 900.)
        u, v = rng.uniform(0.25 * W, 0.75 * W), rng.uniform(0.25 * H, 0.75 * H)
        f = np.full((H, W), 1400., np.float32) + rng.normal(0, 3., (H, W)).astype(np.float32)
        f[rng.uniform(size=(H, W)) < 0.05] = 0.
        r = cube[0] / 2. * fx / d * rng.uniform(0.5, 0.9)
        blob = (xx - u) ** 2 + (yy - v) ** 2 < r * r
        f[blob] = (d + rng.normal(0, 30., (H, W)))[blob].astype(np.float32)
        f[rng.uniform(size=(H, W)) < 0.01] = 2500.
        frames[i] = f
        coms[i] = (u, v, d)
    gt3d = (np.stack([di.jointImgTo3D(c) for c in coms])[:, None, :] + rng.normal(0, 35., (n, J, 3))).astype(np.float32)
    return frames, coms, gt3d
