"""bench.py's command line: `--gpus N` without a launcher starts the N ranks itself (torch.distributed.run) and rank 0 prints
ONE JSON line carrying n_gpus = N.  The CPU tier drives the real control flow (spawn, rendezvous on 127.0.0.1, data-parallel
engine, gradient all-reduce, max-over-ranks timing) with two gloo ranks on the SIMT emulator at a toy size (DPP_BENCH_EMU=1);
on the one-GPU box `--gpus 2` has to fail loudly instead of quietly timing one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=1500):
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None)
    e.pop('RANK', None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def test_gpus_2_spawns_two_ranks_and_reports_them():
    rc, out, err = _bench(['--gpus', '2', '--batch', '4', '--size', '32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                          env=dict(DPP_BENCH_EMU='1'))
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['parallelism'] == 'dp2' and res['config']['global_batch'] == 8
    assert res['scaling'] == 'weak' and res['unit'] == 'depth-crops/sec' and res['value'] > 0
    assert res['config']['final_cost'] == res['config']['final_cost'] and res['config']['final_cost'] > 0      # finite
    d = res['config']['dist']                     # who took part, gathered from all ranks
    assert d['backend'] == 'gloo' and d['world'] == 2 and len(d['devices']) == 2 and len(d['per_rank_ms_per_step']) == 2
    assert all(v > 0 for v in d['per_rank_ms_per_step']) and max(d['per_rank_ms_per_step']) <= res['ms_per_step'] * 1.001


@pytest.mark.gpu
def test_gpus_2_with_gloo_ranks_sharing_the_one_mi355x():
    """the multi-rank bench path on the real kernels.  Two gloo ranks (DPP_DIST_BACKEND=gloo: collectives on host
    copies) share the one MI355X of the test box, each with its own 128-crop shard: n_gpus = 2, global batch 256, a finite cost, both
    ranks' devices and clocks in config.dist.  (RCCL itself needs a second GPU; what it must then show -- one distinct device per
    rank -- is asserted by bench.py.)"""
    rc, out, err = _bench(['--gpus', '2', '--steps', '5', '--warmup', '2', '--no-cpu-baseline'], env=dict(DPP_DIST_BACKEND='gloo'), timeout=900)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 256 and res['config']['parallelism'] == 'dp2'
    assert np.isfinite(res['config']['final_cost']) and res['config']['final_cost'] > 0 and res['value'] > 0
    d = res['config']['dist']
    assert d['backend'] == 'gloo' and d['world'] == 2 and len(d['devices']) == 2 and len(d['per_rank_ms_per_step']) == 2


@pytest.mark.gpu
def test_one_gpu_line_carries_the_forward_only_and_no_augment_legs():
    """"crops/s (fwd only; fwd+bwd+ADAM; with/without augmentation)" in ONE line, plus the device's own clock and
    the participating device."""
    rc, out, err = _bench(['--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-trainer'], timeout=600)
    assert rc == 0, err[-3000:]
    res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
    fo, na = res['forward_only'], res['no_augment']
    assert fo['finite'] and fo['value'] > res['value'] and fo['launches'] > 0 and fo['batch'] == 128
    assert na['value'] > 0.8 * res['value']
    assert 0.5 * res['ms_per_step'] < res['config']['hip_event_ms_per_step'] <= res['ms_per_step'] * 1.05
    d = res['config']['dist']
    assert d['world'] == 1 and len(d['devices']) == 1 and d['devices'][0].startswith('cuda:0')


def test_world_size_mismatch_is_an_error():
    """Started under a launcher with another rank count than --gpus says: refuse, do not silently time something else."""
    rc, out, err = _bench(['--gpus', '2', '--batch', '4', '--size', '32', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                          env=dict(DPP_BENCH_EMU='1', WORLD_SIZE='1', RANK='0'))
    assert rc != 0 and '--gpus 2' in (out + err)


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has %d GPUs' % torch.cuda.device_count())
    rc, out, err = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'])
    assert rc != 0 and 'needs 2 visible' in err and not [l for l in out.splitlines() if l.startswith('{')]


def test_pipelined_augmentation_trains_on_the_same_minibatches():
    """Default: step i's minibatch is augmented by step i - 1 on the gradient-branch stream (engine.step_plan(prefetch=));
    --augment-inline does it at the start of step i.  Same slices, same draw counters: the cost after the run is identical."""
    costs = {}
    for flag in ([], ['--augment-inline']):
        rc, out, err = _bench(['--batch', '4', '--size', '32', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'] + flag,
                              env=dict(DPP_BENCH_EMU='1'))
        assert rc == 0, err[-3000:]
        res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
        assert res['config']['augment_pipelined'] == (not flag)
        costs[bool(flag)] = res['config']['final_cost']
    assert costs[True] == costs[False] and costs[True] > 0


@pytest.mark.gpu
def test_plan_runner_modes_train_the_same_net_on_the_gpu():
    """The plan runner's issue modes only change HOW the two lanes are ordered against each other -- forks / joins riding on kernel
    completion signals (default), as recorded events (DPP_FORK_STOP_EVENT=0), the gradient branch dealt to two side streams
    (DPP_SIDE_STREAMS=2), the augmentation inline instead of prefetched -- so a few steps of the bs128 bench must end on the same cost
    in every mode (a missing dependency shows up as a different or non-finite cost)."""
    costs = {}
    for name, env, flags in (('default', {}, []), ('recorded_events', dict(DPP_FORK_STOP_EVENT='0'), []),
                             ('two_side_streams', dict(DPP_SIDE_STREAMS='2'), []), ('inline_augment', {}, ['--augment-inline'])):
        rc, out, err = _bench(['--steps', '6', '--warmup', '2', '--no-cpu-baseline'] + flags, env=env, timeout=600)
        assert rc == 0, err[-3000:]
        costs[name] = json.loads([l for l in out.splitlines() if l.startswith('{')][0])['config']['final_cost']
    assert len(set(costs.values())) == 1 and costs['default'] > 0, costs


def test_ablation_variable_is_refused_and_knobs_are_stamped():
    """DPP_WHATIF_SKIP drops launches from the timed plan (tools/whatif.sh): bench.py must not report a number with it set unless told
    so, and any other DPP_* experiment variable present in the environment is recorded in the output line."""
    args = ['--batch', '4', '--size', '32', '--steps', '1', '--warmup', '0', '--no-cpu-baseline']
    rc, out, err = _bench(args, env=dict(DPP_BENCH_EMU='1', DPP_WHATIF_SKIP='bn_finalize'))
    assert rc != 0 and 'DPP_WHATIF_SKIP' in (out + err) and not [l for l in out.splitlines() if l.startswith('{')]
    rc, out, err = _bench(args, env=dict(DPP_BENCH_EMU='1', DPP_EXPERIMENT='1', DPP_KSPLIT='0'))
    assert rc == 0, err[-3000:]
    res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
    assert res['config']['knobs'] == {'DPP_EXPERIMENT': '1', 'DPP_KSPLIT': '0'} and res['config']['ablation'] is None
    rc, out, err = _bench(args, env=dict(DPP_BENCH_EMU='1'))
    assert rc == 0 and json.loads([l for l in out.splitlines() if l.startswith('{')][0])['config']['knobs'] is None


def test_cascade_workload_control_flow():
    """`--workload cascade` (BASELINE configs[4]) at a toy size on the emulator: frames -> refinement cascade -> crops + labels ->
    train step, pipelined and inline, end on the same cost."""
    costs = {}
    for flag in ([], ['--augment-inline']):
        rc, out, err = _bench(['--workload', 'cascade', '--batch', '2', '--size', '32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'] + flag,
                              env=dict(DPP_BENCH_EMU='1'))
        assert rc == 0, err[-3000:]
        res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
        assert 'cascade' in res['config']['workload'] and 'cpu_baseline' not in res
        costs[bool(flag)] = res['config']['final_cost']
    assert costs[True] == costs[False] and costs[True] > 0


@pytest.mark.gpu
def test_cascade_workload_on_the_gpu_full_size():
    """configs[4] on one GPU as bench.py times it: 128 frames of 640x480 per step, 256x256 crops, bf16."""
    rc, out, err = _bench(['--workload', 'cascade', '--size', '256', '--dtype', 'bf16', '--steps', '4', '--warmup', '2', '--no-cpu-baseline'], timeout=900)
    assert rc == 0, err[-3000:]
    res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
    assert res['value'] > 1000 and res['dtype'] == 'bf16' and res['config']['final_cost'] > 0
