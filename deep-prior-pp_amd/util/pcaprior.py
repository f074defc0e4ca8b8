"""The PCA pose prior on the device: what the embedding scripts do once before training,
/root/reference/src/main_nyu_posereg_embedding.py:86-92 --

    pca = PCA(n_components=30)
    pca.fit(HandDetector.sampleRandomPoses(di, rng, train_gt3Dcrop, train_data_com, train_data_cube, 1e6, aug_modes).reshape((-1, J*3)))

-- with the 1e6 label-space augmentations (/root/reference/src/util/handdetector.py:805-909), the column means, the scatter
matrix and its eigen-decomposition all computed by HIP kernels (csrc/prior.hip).  The random DRAWS stay on the host: they come
from the script's numpy RandomState in the reference's order, so the sampled poses are the reference's.

`DevicePCA` offers the attributes and methods of sklearn.decomposition.PCA that the scripts and the trainer use (`mean_`,
`components_`, `explained_variance_`, `transform`, `inverse_transform`); it can be handed to PoseRegNetTrainer as `proj`."""
import numpy

from hipdp.lib import check
from hipdp.runtime import default_runtime

MODE_CODE = {'none': 0, 'com': 1, 'rot': 2, 'sc': 3, 'rot+com': 4, 'com+rot': 4, 'rot+com+sc': 5, 'rot+sc+com': 5}


def sample_random_poses_device(importer, rng, base_poses, base_com, base_cube, num_poses, aug_modes, sigma_com=None, sigma_sc=None,
                               rot_range=None, runtime=None, keep_on_device=False, rot3D=False):
    """HandDetector.sampleRandomPoses(...) with the per-sample arithmetic on the device (rot3D: the rotation modes turn the pose in 3-D
    about its centre, /root/reference/src/util/handdetector.py:870, 891, 903; the per-sample rotation matrices are formed on the host).  Returns the (n, J, 3) float32
    poses (a device Buffer when keep_on_device, else a NumPy array).  Draws: exactly the reference's five rng calls."""
    rt = runtime or default_runtime()
    sigma_com = 5. if sigma_com is None else sigma_com
    sigma_sc = 0.02 if sigma_sc is None else sigma_sc
    rot_range = 180. if rot_range is None else rot_range
    for m in aug_modes:
        if m not in MODE_CODE:
            raise NotImplementedError("augmentation mode %r" % (m,))
    n = int(num_poses)
    base_poses = numpy.ascontiguousarray(base_poses, numpy.float32)
    J = base_poses.shape[1]
    modes = rng.randint(0, len(aug_modes), n)
    ridxs = rng.randint(0, base_poses.shape[0], n)
    off = rng.randn(n, 3) * sigma_com
    sc = numpy.fabs(rng.randn(n) * sigma_sc + 1.)
    rot = rng.uniform(-rot_range, rot_range, size=(n, 3))
    if list(aug_modes) == ['none']:
        out = base_poses / (numpy.asarray(base_cube, numpy.float32)[:, 2] / 2.)[:, None, None]
        return rt.upload(out) if keep_on_device else out
    code = numpy.asarray([MODE_CODE[m] for m in aug_modes], numpy.int32)[modes]
    up = rt.upload
    d_poses, d_com, d_cube = up(base_poses), up(numpy.ascontiguousarray(base_com, numpy.float32)), up(numpy.ascontiguousarray(base_cube, numpy.float32))
    d_mode, d_ridx = up(code.astype(numpy.int32)), up(ridxs.astype(numpy.int32))
    if rot3D:
        from data.transformations import euler_rxyz_matrix
        a = rot * numpy.pi / 180.
        rot_arg = euler_rxyz_matrix(a[:, 0], a[:, 1], a[:, 2]).reshape(n, 9)
    else:
        rot_arg = rot[:, 0]
    d_off, d_sc, d_rot = up(numpy.ascontiguousarray(off, numpy.float64)), up(numpy.ascontiguousarray(sc, numpy.float64)), \
        up(numpy.ascontiguousarray(rot_arg, numpy.float64))
    out = rt.alloc((n, J, 3), zero=False)
    flip = bool(getattr(importer, 'flip_y', importer.__class__.__name__ in ('NYUImporter', 'MSRA15Importer')))
    fn = rt.lib.dpp_pose_sample_rot3d if rot3D else rt.lib.dpp_pose_sample
    check(fn(d_poses.ptr, d_com.ptr, d_cube.ptr, base_poses.shape[0], J, d_mode.ptr, d_ridx.ptr, d_off.ptr, d_sc.ptr,
             d_rot.ptr, n, float(importer.fx), float(importer.fy), float(importer.ux), float(importer.uy), int(flip),
             out.ptr, None, None, rt.stream), 'dpp_pose_sample')
    if keep_on_device:
        out.keep = (d_poses, d_com, d_cube, d_mode, d_ridx, d_off, d_sc, d_rot)      # inputs alive until the kernel has run
        return out
    rt.synchronize()
    return out.get()


class DevicePCA(object):
    """sklearn.decomposition.PCA's fit / transform / inverse_transform on the device (no whitening), for X of shape (N, D),
    D <= 80 (26 joints).  `fit` accepts a NumPy array or a device Buffer (e.g. from sample_random_poses_device(keep_on_device=True))."""

    def __init__(self, n_components=None, runtime=None):
        self.n_components = n_components
        self.rt = runtime or default_runtime()

    def fit(self, X, y=None):
        rt = self.rt
        buf = X if hasattr(X, 'ptr') else rt.upload(numpy.ascontiguousarray(X, numpy.float32))
        N = int(buf.shape[0])
        D = int(numpy.prod(buf.shape[1:]))
        k = D if self.n_components is None else int(self.n_components)
        if not (0 < k <= D):
            raise ValueError("n_components=%r must be between 1 and n_features=%d" % (self.n_components, D))
        ws = rt.alloc(int(rt.lib.dpp_pca_workspace_bytes(N, D)), numpy.uint8, zero=False)
        mean, evals, comps = rt.alloc(D, numpy.float64), rt.alloc(D, numpy.float64), rt.alloc((D, D), numpy.float64)
        check(rt.lib.dpp_pca_fit(buf.ptr, N, D, ws.ptr, mean.ptr, evals.ptr, comps.ptr, rt.stream), 'dpp_pca_fit')
        rt.synchronize()
        ev = evals.get()
        self.n_samples_, self.n_features_in_ = N, D
        self.mean_ = mean.get()
        self.components_ = comps.get()[:k]
        self.explained_variance_ = ev[:k]
        tot = ev.sum()
        self.explained_variance_ratio_ = ev[:k] / tot if tot > 0 else numpy.zeros(k)
        self.singular_values_ = numpy.sqrt(numpy.maximum(ev[:k], 0) * (N - 1))
        self.noise_variance_ = float(ev[k:].mean()) if k < D else 0.0
        self.n_components_ = k
        return self

    def transform(self, X):
        X = numpy.asarray(X)
        return (X - self.mean_) @ self.components_.T

    def fit_transform(self, X, y=None):
        self.fit(X)
        return self.transform(X.get() if hasattr(X, 'ptr') else X)

    def inverse_transform(self, Y):
        return numpy.asarray(Y) @ self.components_ + self.mean_
