"""Data-parallel train step on 2 ranks (gloo, CPU; the real kernels under the SIMT emulator): with sync-BN the
all-reduced gradients equal the single-process gradients on the same global batch; with local BN the replicas still
stay bit-identical to each other after a full step (same all-reduced gradient, replicated ADAM)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from hipdp import engine
from net.resnet import ResNet, ResNetParams
from oracle import nets
from tests.backends import get_runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(tmp_path, sync, world=2, wd=0.0, **extra_env):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.update(extra_env)
        out = os.path.join(str(tmp_path), 'rank%d_%d.npz' % (r, int(sync)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), out, str(int(sync)), repr(wd)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log = p.communicate(timeout=900)[0].decode()
        assert p.returncode == 0, log[-3000:]
    return [np.load(o) for o in outs]


@pytest.mark.parametrize('wd', [1e-3])
def test_sync_bn_gradients_equal_single_process(tmp_path, wd):
    """wd > 0: the weight-decay gradient 2*wd*W is not a per-shard partial sum -- it has to enter ONCE, after the all-reduce."""
    r0, r1 = _run_ranks(tmp_path, sync=True, wd=wd)
    rt = get_runtime('emu')
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=8, numJoints=1, nDims=30))
    # the single-process reference runs the same stand-alone BatchNorm kernels as the sync-BN ranks: in this deep, tiny-batch
    # net f32 rounding differences between kernel variants are amplified enough to flip ReLU masks of near-zero elements
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), fuse_bn=False, weight_decay=wd)
    rng = np.random.RandomState(99)
    x = nets.synthetic_crops(rng, 8, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (8, 30)).astype(np.float32)
    cost, _ = eng.cost_and_grads(x, y)
    # the per-rank cost is the rank's share of the global mean
    assert abs(float(r0['cost'][0] + r1['cost'][0]) - cost) < 1e-5 * abs(cost)
    assert abs(float(r0['global_cost'][0]) - cost) < 1e-5 * abs(cost) and r0['global_cost'][0] == r1['global_cost'][0]
    gmax = 0.0
    ref = {}
    for i, l in enumerate(net.layers):
        for s, p in enumerate(l.params):
            ref['g_%d_%d' % (i, s)] = eng.store.read_grad(p)
            gmax = max(gmax, np.abs(ref['g_%d_%d' % (i, s)]).max())
    for k, g in ref.items():
        assert np.array_equal(r0[k], r1[k]), k                         # both ranks hold the same reduced gradient
        np.testing.assert_allclose(r0[k], g, rtol=0, atol=2e-4 * max(np.abs(g).max(), 5e-3 * gmax), err_msg=k)
    eng.train_step(x, y, 1e-3)
    np.testing.assert_allclose(r0['bn_mean'], [l for l in net.layers if l.__class__.__name__ == 'BatchNormLayer'][0].mean.get_value(),
                               rtol=1e-5, atol=1e-6)
    assert np.array_equal(r0['w_last'], r1['w_last'])


def test_local_bn_replicas_stay_identical(tmp_path):
    r0, r1 = _run_ranks(tmp_path, sync=False)
    for k in r0.files:
        if k.startswith('g_') or k == 'w_last':
            assert np.array_equal(r0[k], r1[k]), k
    assert not np.array_equal(r0['bn_mean'], r1['bn_mean'])            # per-GPU running statistics differ by design


def test_reduce_scatter_all_gather_schedule_equals_the_all_reduce(tmp_path):
    """DPP_ALLREDUCE=rs_ag (round 6): the gradient buckets summed by reduce_scatter + all_gather on the flat
    buffer (+ an all_reduce of the tail that does not divide by the world size) -- two gloo ranks on the emulator: both ranks hold the
    same gradient, and it is the all_reduce schedule's gradient up to the order of two additions (with two ranks: the same bits)."""
    a0, a1 = _run_ranks(tmp_path, sync=False)
    sub = tmp_path / 'rs_ag'
    sub.mkdir()
    b0, b1 = _run_ranks(sub, sync=False, DPP_ALLREDUCE='rs_ag')
    n = 0
    for k in a0.files:
        if k.startswith('g_') or k == 'w_last':
            assert np.array_equal(b0[k], b1[k]), k
            assert np.array_equal(a0[k], b0[k]), k            # two ranks: a + b in either schedule
            n += 1
    assert n > 50 and float(a0['cost'][0]) == float(b0['cost'][0])


@pytest.mark.gpu
def test_rccl_world_of_one_reduce_scatter_all_gather_schedule(tmp_path):
    """The same switch through RCCL on the MI355X with a world of one: in-place reduce_scatter / all_gather of the early FC1 bucket
    started from the side stream, their wait, the second bucket -- identities that must leave gradients and weights bit-identical to the
    all_reduce schedule (what is tested is their ordering against the engine's two HIP streams)."""
    (a,) = _run_ranks(tmp_path, sync=False, world=1, wd=1e-3, DPP_WORKER_BACKEND='nccl', DPP_WORKER_BATCH='8')
    sub = tmp_path / 'rs_ag'
    sub.mkdir()
    (b,) = _run_ranks(sub, sync=False, world=1, wd=1e-3, DPP_WORKER_BACKEND='nccl', DPP_WORKER_BATCH='8', DPP_ALLREDUCE='rs_ag')
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('sync', [False, True])
def test_rccl_world_of_one_equals_single_process(tmp_path, sync):
    """The data-parallel step through RCCL on the MI355X, one rank: the all-reduces / all-gathers are identities, so gradients,
    cost and the updated weights must be BIT-identical to the engine without `dp` -- what is tested is that the collectives are
    ordered correctly against the engine's two HIP streams (a race shows up as a stale or half-reduced FC1 gradient)."""
    B = 8
    (r0,) = _run_ranks(tmp_path, sync=sync, world=1, wd=1e-3, DPP_WORKER_BACKEND='nccl', DPP_WORKER_BATCH=str(B))
    rt = get_runtime('hip')
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=B, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), fuse_bn=not sync, weight_decay=1e-3)
    rng = np.random.RandomState(99)
    x = nets.synthetic_crops(rng, B, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    cost, _ = eng.cost_and_grads(x, y)
    assert float(r0['cost'][0]) == cost and float(r0['global_cost'][0]) == cost
    for i, l in enumerate(net.layers):
        for s, p in enumerate(l.params):
            assert np.array_equal(r0['g_%d_%d' % (i, s)], eng.store.read_grad(p)), (i, s)
    eng.train_step(x, y, 1e-3)
    assert np.array_equal(r0['w_last'], net.layers[-1].W.get_value())


# ---- data parallelism behind the drop-in API: PoseRegNetTrainer(..., dp=) ---------------------------------------------------------
def _run_trainer(tmp_path, world, sync, GB, epochs, backend='emu', **extra_env):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, DPP_WORKER_BACKEND=backend)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.update(extra_env)
        out = os.path.join(str(tmp_path), 'trainer_w%d_r%d_%d.npz' % (world, r, int(sync)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_trainer_worker.py'), out, str(int(sync)), str(GB), str(epochs)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log = p.communicate(timeout=1500)[0].decode()
        assert p.returncode == 0, log[-4000:]
    return [np.load(o) for o in outs]


def _check_trainer_ranks_against_single_process(ranks, single, G, B, tight):
    GB = G * B
    n_pad = single['shard_x'].shape[0]
    assert n_pad % GB == 0
    glob = np.arange(n_pad).reshape(-1, G, B)
    for r, res in enumerate(ranks):
        rows = glob[:, r].reshape(-1)
        # the shard is the rank's slice of every global minibatch (padding rows included: the same seeded draws as alignData's)
        assert np.array_equal(res['shard_x'], single['shard_x'][rows]) and np.array_equal(res['shard_com'], single['shard_com'][rows])
        vrows = np.arange(single['shard_val'].shape[0]).reshape(-1, G, B)[:, r].reshape(-1)
        assert np.array_equal(res['shard_val'], single['shard_val'][vrows])
        # device draws are keyed by the GLOBAL sample index: the augmented crops / labels of the last macro-batch reload are the rows
        # the single process produced, bit for bit
        assert np.array_equal(res['aug_x'], single['aug_x'][rows]) and np.array_equal(res['aug_y'], single['aug_y'][rows])
        assert (res['aug_x'] != res['shard_x']).mean() > 0.005
    a, b = ranks[0], ranks[1]
    assert np.array_equal(a['w'], b['w']) and np.array_equal(a['costs'], b['costs']) and np.array_equal(a['val'], b['val'])   # replicas
    assert bool(a['snapshot'][0]) and not bool(b['snapshot'][0])                 # rank 0 writes the snapshots
    assert len(a['costs']) == len(single['costs']) and np.all(np.isfinite(a['costs']))
    # the first minibatch: the same global batch, sync-BN statistics, one cost
    assert abs(a['costs'][0] - single['costs'][0]) < tight * abs(single['costs'][0]), (a['costs'][0], single['costs'][0])
    # later steps: ADAM turns round-off-sized gradient differences into +-lr steps (in the reference's float32 graph too)
    np.testing.assert_allclose(a['costs'], single['costs'], rtol=3e-2)
    np.testing.assert_allclose(a['val'], single['val'], rtol=3e-2)


def test_trainer_on_two_ranks_equals_one_rank_on_the_same_global_batch(tmp_path):
    """PoseRegNetTrainer(..., dp=DataParallel(sync_bn=True)) on 2 gloo ranks x batch 4 against the plain trainer at batch 8 (emulator)."""
    ranks = _run_trainer(tmp_path, 2, True, 8, 1)
    (single,) = _run_trainer(tmp_path, 1, True, 8, 1)
    _check_trainer_ranks_against_single_process(ranks, single, 2, 4, tight=2e-5)


@pytest.mark.gpu
def test_trainer_two_gloo_ranks_share_the_gpu_and_equal_one_process(tmp_path):
    """The same on the real kernels: two ranks (gloo, host-staged collectives) sharing the one MI355X against a single HIP process:
    sharding, global-index augmentation draws, gradient all-reduce (early FC1 bucket from the side stream + the rest), sync-BN
    all-gathers, replicated ADAM, averaged validation, rank-0 snapshots."""
    ranks = _run_trainer(tmp_path, 2, True, 16, 2, backend='hip', DPP_DIST_BACKEND='gloo')
    (single,) = _run_trainer(tmp_path, 1, True, 16, 2, backend='hip')
    _check_trainer_ranks_against_single_process(ranks, single, 2, 8, tight=2e-5)


def _check_local_bn_sum_of_shard_gradients(r0, B, wd, backend='hip'):
    """The FAST data-parallel mode (local BatchNorm statistics, fused-statistics kernels -- what bench.py --gpus N measures): rank r's
    gradient is the single-process gradient of ITS shard under the global-batch cost normalisation, so the all-reduced gradient must be
    (g(shard 0) + g(shard 1)) / G with g() from a plain single-process engine at batch B (per-shard statistics, the same fused kernels),
    plus the weight decay 2 wd W added ONCE after the all-reduce; the global cost is the mean of the shard costs plus the regulariser."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(99)
    x = nets.synthetic_crops(rng, 2 * B, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (2 * B, 30)).astype(np.float32)
    shard_g, shard_cost = [], []
    net = None
    for r in range(2):
        net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=B, numJoints=1, nDims=30))
        eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))        # fuse_bn default, no weight decay
        cost, _ = eng.cost_and_grads(x[r * B:(r + 1) * B], y[r * B:(r + 1) * B])
        shard_cost.append(cost)
        shard_g.append({(i, s): eng.store.read_grad(p) for i, l in enumerate(net.layers) for s, p in enumerate(l.params)})
    reg = sum(float((l.W.get_value().astype(np.float64) ** 2).sum()) for l in net.layers if hasattr(l, 'W'))
    want_cost = 0.5 * (shard_cost[0] + shard_cost[1]) + wd * reg
    assert abs(float(r0['global_cost'][0]) - want_cost) < 2e-5 * abs(want_cost), (float(r0['global_cost'][0]), want_cost)
    gmax = max(np.abs(v).max() for v in shard_g[0].values())
    for (i, s_), g0 in shard_g[0].items():
        want = 0.5 * (g0.astype(np.float64) + shard_g[1][(i, s_)])
        l = net.layers[i]
        if hasattr(l, 'W') and s_ == 0:
            want = want + 2.0 * wd * l.W.get_value().astype(np.float64)
        np.testing.assert_allclose(r0['g_%d_%d' % (i, s_)], want, rtol=0, atol=2e-5 * max(np.abs(want).max(), 5e-3 * gmax), err_msg='%d %d' % (i, s_))


def test_local_bn_allreduced_gradient_is_the_sum_of_the_shard_gradients(tmp_path):
    """Emulator form of the check below (2 gloo ranks, local BatchNorm, batch 4 per rank)."""
    r0, r1 = _run_ranks(tmp_path, sync=False, world=2, wd=1e-3)
    for k in r0.files:
        if k.startswith('g_'):
            assert np.array_equal(r0[k], r1[k]), k
    _check_local_bn_sum_of_shard_gradients(r0, 4, wd=1e-3, backend='emu')


@pytest.mark.gpu
@pytest.mark.parametrize('sync', [False, True])
def test_two_gloo_ranks_on_one_gpu_gradients(tmp_path, sync):
    """Engine level, two ranks on the real kernels (DPP_DIST_BACKEND=gloo on one MI355X): with sync-BN the all-reduced gradients equal
    the single-process HIP gradients on the same global batch at the 2e-4 bar; with local BN the replicas stay bit-identical."""
    B = 8
    r0, r1 = _run_ranks(tmp_path, sync=sync, world=2, wd=1e-3, DPP_WORKER_BACKEND='hip-gloo', DPP_DIST_BACKEND='gloo', DPP_WORKER_BATCH=str(B))
    for k in r0.files:
        if k.startswith('g_') or k == 'w_last':
            assert np.array_equal(r0[k], r1[k]), k
    if not sync:
        assert not np.array_equal(r0['bn_mean'], r1['bn_mean'])
        _check_local_bn_sum_of_shard_gradients(r0, B, wd=1e-3)
        return
    rt = get_runtime('hip')
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=2 * B, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), fuse_bn=False, weight_decay=1e-3)
    rng = np.random.RandomState(99)
    x = nets.synthetic_crops(rng, 2 * B, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (2 * B, 30)).astype(np.float32)
    cost, _ = eng.cost_and_grads(x, y)
    assert abs(float(r0['global_cost'][0]) - cost) < 1e-5 * abs(cost)
    gmax = max(np.abs(eng.store.read_grad(p)).max() for l in net.layers for p in l.params)
    for i, l in enumerate(net.layers):
        for s, p in enumerate(l.params):
            g = eng.store.read_grad(p)
            np.testing.assert_allclose(r0['g_%d_%d' % (i, s)], g, rtol=0, atol=2e-4 * max(np.abs(g).max(), 5e-3 * gmax), err_msg='%d %d' % (i, s))


# ---- test-time inference under data parallelism: NetBase.computeOutput(dp=) ------------------------------------------------------------
def _run_output_ranks(tmp_path, world, **extra_env):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.update(extra_env)
        out = os.path.join(str(tmp_path), 'output_w%d_r%d.npz' % (world, r))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_output_worker.py'), out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        log = p.communicate(timeout=900)[0].decode()
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(o)) for o in outs]          # (read now: a later run with the same world size reuses the file names)


def _check_sharded_outputs(ranks, single):
    for res in ranks:                       # every rank returns the FULL array, equal to the single-process one bit for bit
        assert res['out'].shape == single['out'].shape == (11, 42)
        assert np.array_equal(res['out'], single['out']) and np.array_equal(res['out_attr'], single['out'])
        assert np.array_equal(res['few'], single['few']) and res['few'].shape == (3, 42)
    assert np.abs(single['out']).max() > 0.01


def test_compute_output_shards_batches_over_two_ranks(tmp_path):
    """SURVEY.md section 8(e), "computeOutput: shard batches, all-gather outputs" (/root/reference/src/net/netbase.py:217-316 is the
    single-device loop): 11 samples at batch 4 = 3 batches, rank 0 evaluates batches 0 and 2, rank 1 batch 1 (the last one padded by
    repeating the last sample); 2 gloo ranks on the emulator against one process."""
    _check_sharded_outputs(_run_output_ranks(tmp_path, 2), _run_output_ranks(tmp_path, 1)[0])


def test_compute_output_uses_rank0_running_statistics_after_local_bn_training(tmp_path):
    """with per-GPU BatchNorm statistics the ranks' running mean / inv_std diverge during training; the sharded
    computeOutput must still be ONE model -- rank 0's, whose checkpoint is the one written -- on every rank and for every batch."""
    two = _run_output_ranks(tmp_path, 2, DPP_WORKER_DIVERGE='1')
    one = _run_output_ranks(tmp_path, 1, DPP_WORKER_DIVERGE='1')[0]
    plain = _run_output_ranks(tmp_path, 1)[0]
    assert not np.array_equal(one['out'], plain['out'])           # the perturbed statistics matter
    _check_sharded_outputs(two, one)
    # ... and a sharded evaluation leaves every rank's own running statistics as they were (an inference call in the middle
    # of training must not rewrite the replicas' state)
    for r in two:
        assert np.array_equal(r['bn_mean_before'], r['bn_mean_after'])
    assert not np.array_equal(two[0]['bn_mean_after'], two[1]['bn_mean_after'])


@pytest.mark.gpu
def test_compute_output_two_gloo_ranks_share_the_gpu(tmp_path):
    """The same on the real kernels: two gloo ranks sharing the one MI355X against a single HIP process, bit for bit."""
    _check_sharded_outputs(_run_output_ranks(tmp_path, 2, DPP_WORKER_BACKEND='hip-gloo', DPP_DIST_BACKEND='gloo'),
                           _run_output_ranks(tmp_path, 1, DPP_WORKER_BACKEND='hip-gloo')[0])


# ---- host logic of the data-parallel layer (no process group needed) ------------------------------------------------------------------
def _fake_dp(world, rank):
    from hipdp.parallel import DataParallel
    dp = DataParallel.__new__(DataParallel)
    dp.world, dp.rank = world, rank
    return dp


def test_shard_padding_modes_follow_align_data():
    """Training arrays are completed to whole GLOBAL minibatches the way alignData completes them (/root/reference/src/trainer/
    nettrainer.py:365-413): rows drawn by RandomState(n) with pad_random, the last row repeated without; validation arrays are cut, and
    a validation set shorter than one global minibatch is refused instead of becoming an empty shard on every rank (NaN validation loss)."""
    G, B, n = 2, 4, 21
    data = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    parts = {m: [_fake_dp(G, r).shard(data, B, **kw) for r in range(G)] for m, kw in
             (('last', dict(pad='last')), ('random', dict(pad_rng_seed=n)), ('cut', dict()))}
    for m, want_n in (('last', 24), ('random', 24), ('cut', 16)):
        glob = np.empty((want_n, 3), np.float32)
        idx = np.arange(want_n).reshape(-1, G, B)
        for r in range(G):
            glob[idx[:, r].reshape(-1)] = parts[m][r]
        assert np.array_equal(glob[:min(n, want_n)], data[:min(n, want_n)])
        if m == 'last':
            assert np.array_equal(glob[n:], np.repeat(data[-1:], 3, axis=0))
        if m == 'random':
            rng = np.random.RandomState(n)
            assert np.array_equal(glob[n:], np.stack([data[rng.randint(0, n)] for _ in range(3)]))
    with pytest.raises(ValueError, match='fewer than one global minibatch'):
        _fake_dp(G, 0).shard(data[:7], B, what='the validation set')
    # the trainer picks the mode from cfgParams.pad_random
    from trainer.nettrainer import NetTrainer

    class _T(object):
        pass
    for pad_random, mode in ((True, 'random'), (False, 'last')):
        t = _T()
        t.dp, t.cfgParams = _fake_dp(G, 1), _T()
        t.cfgParams.batch_size, t.cfgParams.pad_random = B, pad_random
        assert np.array_equal(NetTrainer._shard_train(t, data), parts[mode][1])


def test_default_runtime_binds_the_ranks_own_gpu(monkeypatch):
    """Under torchrun whoever asks for the process-wide runtime first (the importers' device crops, DevicePCA, the trainer) gets one on
    cuda:LOCAL_RANK; a trainer handed a runtime on another GPU with dp='env' refuses (every rank used to end up on cuda:0)."""
    import torch
    from hipdp import parallel, runtime
    state = dict(cur=0)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda i: state.__setitem__('cur', int(i)))
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: state['cur'])

    class _Rt(object):
        def __init__(self):
            self.device = torch.device('cuda', torch.cuda.current_device())
    monkeypatch.setattr(runtime, 'TorchHipRuntime', _Rt)
    monkeypatch.setattr(runtime, '_default', None)
    monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert parallel.local_device_index() == 3
    rt = runtime.default_runtime()
    assert rt.device.index == 3 and state['cur'] == 3
    parallel.check_runtime_device(rt)
    stale = _Rt.__new__(_Rt)
    stale.device = torch.device('cuda', 0)
    with pytest.raises(RuntimeError, match='owns cuda:3'):
        parallel.check_runtime_device(stale)
    monkeypatch.setenv('LOCAL_RANK', '11')                 # several ranks per visible GPU wrap (two gloo ranks on one MI355X)
    assert parallel.local_device_index() == 3
    monkeypatch.setenv('WORLD_SIZE', '1')
    assert parallel.local_device_index() is None
