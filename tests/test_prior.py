"""SURVEY.md section 8(f) rank 4 on the device (csrc/prior.hip): sampleRandomPoses against the reference's own output (the golden
fixture tests/golden/poses.npz was produced by /root/reference/src/util/handdetector.py:805-909 itself) and against the host
restatement, the PCA fit against scikit-learn's, the evaluation metrics against the NumPy formulas of
/root/reference/src/util/handpose_evaluation.py:92-228 (including NaN joints)."""
import os

import numpy as np
import pytest
from sklearn.decomposition import PCA

from data.importers import ICVLImporter, NYUImporter
from hipdp import runtime as R
from tests.backends import BACKENDS, get_runtime
from util.handdetector import HandDetector
from util.handpose_evaluation import DeviceHandposeEvaluation, HandposeEvaluation
from util.pcaprior import DevicePCA, sample_random_poses_device

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _base(rng, n, J, cube=250.):
    poses = rng.normal(0, 35., (n, J, 3)).astype(np.float32)
    com = np.stack([rng.uniform(-80, 80, n), rng.uniform(-60, 60, n), rng.uniform(300, 600, n)], axis=1).astype(np.float32)
    cubes = np.tile(np.float32([cube, cube, cube]), (n, 1))
    return poses, com, cubes


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('modes', [['com', 'rot', 'none'], ['com', 'rot', 'sc', 'none'], ['rot+com', 'rot+com+sc', 'sc'], ['none']])
def test_device_pose_sampling_equals_the_host_restatement(backend, modes):
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    for imp, J in ((ICVLImporter('../data/ICVL/'), 16), (NYUImporter('../data/NYU/'), 14)):
        poses, com, cubes = _base(np.random.RandomState(3), 9, J)
        host = HandDetector.sampleRandomPoses(imp, np.random.RandomState(7), poses, com, cubes, 500, modes)
        dev = sample_random_poses_device(imp, np.random.RandomState(7), poses, com, cubes, 500, modes, runtime=rt)
        assert dev.shape == host.shape == ((500, J, 3) if modes != ['none'] else (9, J, 3))
        # same draws, same operation order and precision: equal up to the device's cos / sin (an ulp of the rotated pixel)
        np.testing.assert_allclose(dev, host, rtol=0, atol=3e-6 * max(1.0, np.abs(host).max()))
        if 'rot' not in ''.join(modes):
            assert np.array_equal(dev, host)


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_pose_sampling_matches_the_reference_fixture(backend):
    """poses.npz: inputs and outputs of the reference's own sampleRandomPoses (tests/golden/make_golden.py)."""
    rt = get_runtime(backend)
    g = np.load(os.path.join(GOLD, 'poses.npz'))
    for nm, di in (('icvl', ICVLImporter('x')), ('nyu', NYUImporter('x'))):
        args = (g['%s_gt' % nm], g['%s_com' % nm], g['%s_cube' % nm], 300)
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            got = sample_random_poses_device(di, np.random.RandomState(9), *args, modes, runtime=rt)
            want = g['%s_%s' % (nm, tag)]
            assert got.shape == want.shape and got.dtype == np.float32
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)       # the bar tests/test_oracle.py holds the host version to


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_pose_sampling_rot3d_matches_host_and_reference_fixture(backend):
    """rot3D=True (/root/reference/src/util/handdetector.py:870, 891, 903): the device path against the host
    restatement (bit for bit: the rotation matrices are the host's, the arithmetic is IEEE without contraction) and against rot3d.npz,
    the reference's own output (tests/golden/make_golden_r6.py)."""
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    g = np.load(os.path.join(GOLD, 'rot3d.npz'))
    for nm, di in (('icvl', ICVLImporter('x')), ('nyu', NYUImporter('x'))):
        args = (g['%s_gt' % nm], g['%s_com' % nm], g['%s_cube' % nm], 300)
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            got = sample_random_poses_device(di, np.random.RandomState(9), *args, modes, runtime=rt, rot3D=True)
            host = HandDetector.sampleRandomPoses(di, np.random.RandomState(9), *args, modes, rot3D=True)
            want = g['%s_%s' % (nm, tag)]
            assert got.shape == want.shape and got.dtype == np.float32
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
            assert np.array_equal(got, host)
            flat = sample_random_poses_device(di, np.random.RandomState(9), *args, modes, runtime=rt)
            assert not np.array_equal(flat, got)          # the in-plane rotation is a different sample


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('J', [14, 21])
def test_device_pca_matches_sklearn(backend, J):
    rt = get_runtime(backend)
    rng = np.random.RandomState(5)
    D, N, k = J * 3, 3000, 30
    # poses with a decaying spectrum and an offset mean, like joint coordinates
    basis, _ = np.linalg.qr(rng.normal(size=(D, D)))
    X = ((rng.normal(size=(N, D)) * (2.0 * 0.8 ** np.arange(D))) @ basis.T + rng.normal(0, 0.3, D)).astype(np.float32)
    ref = PCA(n_components=k, svd_solver='full').fit(X.astype(np.float64))
    dev = DevicePCA(n_components=k, runtime=rt).fit(X)
    np.testing.assert_allclose(dev.mean_, ref.mean_, rtol=0, atol=1e-6)
    np.testing.assert_allclose(dev.explained_variance_, ref.explained_variance_, rtol=1e-6)
    np.testing.assert_allclose(dev.explained_variance_ratio_, ref.explained_variance_ratio_, rtol=1e-6)
    # eigenvectors up to sign (sklearn flips by the left singular vectors, the kernel by the largest entry)
    dots = np.abs(np.sum(dev.components_ * ref.components_, axis=1))
    assert dots.min() > 1 - 1e-6, dots.min()
    Y = rng.normal(size=(7, D)).astype(np.float32)
    np.testing.assert_allclose(np.abs(dev.transform(Y)), np.abs(ref.transform(Y)), rtol=0, atol=1e-5)
    np.testing.assert_allclose(dev.inverse_transform(dev.transform(Y)), ref.inverse_transform(ref.transform(Y)), rtol=0, atol=1e-5)
    assert all(c[np.argmax(np.abs(c))] > 0 for c in dev.components_)
    # a device buffer goes in without a host round trip
    dev2 = DevicePCA(n_components=k, runtime=rt).fit(rt.upload(X))
    assert np.array_equal(dev2.components_, dev.components_)
    with pytest.raises(ValueError):
        DevicePCA(n_components=D + 1, runtime=rt).fit(X)


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_pca_converges_at_the_largest_dimension(backend):
    """D = 78 (26 joints; dpp_pca_fit takes D <= 80) with clustered eigenvalues: the Jacobi sweeps run until the off-diagonal mass is at rounding
    level (not a fixed count), so the eigenpairs satisfy C v = lambda v to f64 precision."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(6)
    D, N = 78, 1200
    basis, _ = np.linalg.qr(rng.normal(size=(D, D)))
    spectrum = np.concatenate([np.full(20, 3.0), 3.0 - 1e-3 * np.arange(20), 0.5 ** np.arange(38)])
    X = ((rng.normal(size=(N, D)) * np.sqrt(spectrum)) @ basis.T).astype(np.float32)
    dev = DevicePCA(n_components=D, runtime=rt).fit(X)
    Xc = X.astype(np.float64) - X.astype(np.float64).mean(0)
    C = Xc.T @ Xc / (N - 1)
    V = dev.components_.astype(np.float64)
    lam = dev.explained_variance_.astype(np.float64)
    assert np.all(np.diff(lam) <= 1e-12)
    with pytest.raises(Exception):
        DevicePCA(n_components=3, runtime=rt).fit(np.zeros((100, 81), np.float32))        # beyond the LDS-resident limit: refused
    resid = np.abs(C @ V.T - V.T * lam[None, :]).max()
    assert resid < 1e-12 * lam[0] * D, resid
    np.testing.assert_allclose(V @ V.T, np.eye(D), rtol=0, atol=1e-12)
    np.testing.assert_allclose(lam, np.linalg.eigvalsh(C)[::-1], rtol=0, atol=1e-12 * lam[0])


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_evaluation_equals_numpy_metrics(backend):
    rt = get_runtime(backend)
    rng = np.random.RandomState(9)
    N, J = 700, 14
    gt = rng.normal(0, 60, (N, J, 3)).astype(np.float32)
    pr = (gt + rng.normal(0, 8, (N, J, 3))).astype(np.float32)
    pr[5, 3, 1] = np.nan                    # a missing joint
    pr[40] = np.nan                         # a frame without any prediction
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        host = HandposeEvaluation(gt.astype(np.float64), pr.astype(np.float64))     # the kernels form the errors in f64
        dev = DeviceHandposeEvaluation(gt, pr, runtime=rt)
        np.testing.assert_allclose(dev.getMeanError(), host.getMeanError(), rtol=1e-12)
        np.testing.assert_allclose(dev.getStdError(), host.getStdError(), rtol=1e-10)
        np.testing.assert_allclose(dev.getMaxError(), host.getMaxError(), rtol=1e-12)
        np.testing.assert_allclose(dev.getMedianError(), host.getMedianError(), rtol=1e-12)
        np.testing.assert_allclose(dev.getMeanErrorOverSeq(), host.getMeanErrorOverSeq(), rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(dev.getMaxErrorOverSeq(), host.getMaxErrorOverSeq(), rtol=1e-12, equal_nan=True)
        for j in (0, 3, 13):
            np.testing.assert_allclose(dev.getJointMeanError(j), host.getJointMeanError(j), rtol=1e-12)
            np.testing.assert_allclose(dev.getJointStdError(j), host.getJointStdError(j), rtol=1e-10)
            np.testing.assert_allclose(dev.getJointMaxError(j), host.getJointMaxError(j), rtol=1e-12)
        for d in (0, 10, 20, 35.5, 80):
            assert dev.getNumFramesWithinMaxDist(d) == host.getNumFramesWithinMaxDist(d)
            assert dev.getNumFramesWithinMeanDist(d) == host.getNumFramesWithinMeanDist(d)


@pytest.mark.parametrize('backend', BACKENDS)
def test_poses_kept_on_the_device_are_ordinary_buffers(backend):
    """sample_random_poses_device(keep_on_device=True) hands DevicePCA a device buffer; it must still download, view and feed
    fit_transform like any other buffer (its `owner` had been replaced by a tuple)."""
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    imp = ICVLImporter('../data/ICVL/')
    poses, com, cubes = _base(np.random.RandomState(3), 9, 16)
    modes = ['com', 'rot', 'none']
    host = sample_random_poses_device(imp, np.random.RandomState(7), poses, com, cubes, 400, modes, runtime=rt)
    dev = sample_random_poses_device(imp, np.random.RandomState(7), poses, com, cubes, 400, modes, runtime=rt, keep_on_device=True)
    assert dev.keep is not None and np.array_equal(dev.get(), host)
    flat = dev.reshape((400, 48))
    assert flat.keep is dev.keep and np.array_equal(flat.view(48, (48,)).get(), host[1].reshape(-1))
    a = DevicePCA(n_components=10, runtime=rt).fit(flat)
    b = DevicePCA(n_components=10, runtime=rt).fit(host.reshape(400, 48))
    np.testing.assert_allclose(a.components_, b.components_, rtol=0, atol=1e-6)
    np.testing.assert_allclose(DevicePCA(n_components=10, runtime=rt).fit_transform(flat), b.transform(host.reshape(400, 48)), rtol=0, atol=1e-3)
