#!/usr/bin/env python3
"""
bench.py -- the hot-path benchmark of BASELINE.json on MI355X.

One "step" = one pass of the DeepPrior++ hot path over one minibatch: fused crop augmentation (rot / CoM / none,
the mains' aug_modes) of 128 device-resident synthetic depth crops + PCA-prior label projection, then train_model
= forward + sum-squared-error loss + backward + the reference's ADAM of the ResNet pose regressor
(`ResNet(type=0, numJoints=1, nDims=30)`, 18.7 M parameters, fp32), i.e. BASELINE.json configs[1]
"NYU posereg_embedding ResNet-50, 30-dim PCA prior, bs128 fp32, 1xMI355X".  Inputs are resident in HBM before the
timed region.  The whole step is ONE call into libdpp_hip.so (a recorded launch plan, include/dpp_hip.h dpp_plan_*).

`--gpus N` (N > 1): one rank per GPU over RCCL.  Started by the driver under torch.distributed.run the script finds
RANK / WORLD_SIZE in the environment; started plainly (`python bench.py --gpus 4`) it launches the N ranks itself.  Every
rank processes its own 128-crop shard of the global minibatch (weak scaling), the flat fp32 gradient buffer is all-reduced
(sum of per-shard partial gradients of the global-batch cost) between backward and the replicated ADAM.

`--workload cascade` (with `--size 256 [--dtype bf16]`) is BASELINE.json configs[4] on this GPU: every step cuts its minibatch out of
640x480 depth frames through the CoM-refinement cascade (crop -> centre of mass -> ScaleNet -> crop, hipdp/cascade.py) and trains the
256x256 ResNet on it -- one plan per step.

Prints ONE JSON line (rank 0).  `roofline` describes the kernel family with the largest share of the step time,
timed live with HIP events on the launch stream; `cpu_baseline` times the oracle's PyTorch-CPU restatement of the same
train step on the host cores (reported only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'deep-prior-pp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

FLOP_PER_CROP = {128: 722.6e6, 256: 2.9e9}     # fwd + dgrad + wgrad, SURVEY.md section 8(d)
PEAK_MFMA_F32 = 157.3e12       # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
PEAK_MFMA_BF16 = 2.5e15        # dense
PEAK_HBM = 8.0e12


def csrc_sha16():
    """Fingerprint of the kernel sources: rocprofv3 counter files under profiles/ record it, and `roofline.traffic` is only taken
    from a file measured on THESE sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, 'deep-prior-pp_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'deep-prior-pp_amd', 'csrc', '*.h'))):
        with open(fn, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# environment variables that select a measured configuration through a command-line flag of this script (recorded in `config` by
# that flag) or plumbing; every other DPP_* variable is an experiment / opt-in knob and is stamped into config.knobs
PLUMBING_ENV = ('DPP_LAUNCH_MODE', 'DPP_BF16', 'DPP_DIST_BACKEND', 'DPP_BENCH_EMU', 'DPP_TEST_WORKERS', 'DPP_ALLREDUCE')
ABLATION_ENV = ('DPP_WHATIF_SKIP',)            # drops launches from the timed plan: results are wrong on purpose


def _time_cpu_steps(batch, size, min_steps, max_steps, budget_s):
    """(seconds per step over the timed steps, their median, the count) of the oracle's PyTorch-CPU train step at `batch`."""
    import torch
    from oracle import nets, torch_ref
    onet = nets.build_resnet(type=0, wIn=size, hIn=size, batchSize=batch, numJoints=1, nDims=30)
    P = nets.init_params(onet, np.random.RandomState(23455), np.float32)
    tr = torch_ref.TorchTrainer(onet, P)
    rng = np.random.RandomState(3)
    x = torch.tensor(nets.synthetic_crops(rng, batch, size, size, np.float32))
    y = torch.tensor(rng.normal(0, 0.3, (batch, 30)).astype(np.float32))
    t_w = time.time()
    tr.step(x, y, 1e-3)                      # warm-up (thread pools, allocator, oneDNN primitive caches: the first TWO steps are slow)
    if time.time() - t_w < 3.0:              # (not a second one where a step takes tens of seconds: the oversubscribed all-thread probe)
        tr.step(x, y, 1e-3)
    times = []
    t_start = time.time()
    while len(times) < min_steps or (time.time() - t_start < budget_s and len(times) < max_steps):
        t0 = time.time()
        tr.step(x, y, 1e-3)
        times.append(time.time() - t0)
    return sum(times) / len(times), float(np.median(times)), len(times), times


def _augment_worker(job):
    """One of the reference's augmentation worker processes (nettrainer.py:601-628 starts para_num_proc = 8 of them): `n` crops through
    the oracle's NumPy restatement of NetTrainer.augmentCrop."""
    seed, n = job
    from oracle import augment as A
    rng = np.random.RandomState(seed)
    cam = A.Camera.nyu()
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, 8, cam, cube=(300., 300., 300.), joints=14)
    modes = ['com', 'rot', 'none']
    mi, offs, rots, scs = A.draw_params(np.random.RandomState(seed + 1), n, len(modes))
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 8
        A.augment_crop(imgs[k].copy(), gts[k].copy(), cam.joint3DToImg(coms[k]), cubes[k], Ms[k], modes[mi[i]], offs[i], rots[i], scs[i],
                       cam, abs(cam.fx), abs(cam.fy))
    return time.perf_counter() - t0


def cpu_augment_baseline(procs, crops_per_proc=2048):
    """The online augmentation on the host the way the reference runs it: NumPy per crop, in `procs` worker processes (started and
    warmed before the timed region: the reference keeps its workers alive across epochs)."""
    import multiprocessing as mp
    if procs == 1:
        _augment_worker((3, 8))
        t0 = time.perf_counter()
        _augment_worker((11, crops_per_proc))
        dt = time.perf_counter() - t0
    else:
        with mp.get_context('fork').Pool(procs) as pool:
            pool.map(_augment_worker, [(3 + k, 8) for k in range(procs)])
            t0 = time.perf_counter()
            pool.map(_augment_worker, [(11 + 7 * k, crops_per_proc) for k in range(procs)], chunksize=1)
            dt = time.perf_counter() - t0
    return dict(value=round(procs * crops_per_proc / dt, 1), unit='augmented depth-crops/sec', processes=procs, kind='port',
                sample='%d crops per process through the NumPy restatement of augmentCrop (oracle/augment.py; modes com / rot / none, NYU '
                       'geometry)' % crops_per_proc)


def forward_parity_mm(rt, size, frames=8):
    """Part of the cpu_baseline leg (the checker, after the timed region): the deterministic HIP forward (NetBase.computeOutput) of the
    ResNet with the PCA-prior layer (type 1, NYU's 14 joints) on `frames` synthetic crops against the float64 oracle on the same
    weights, in mm on a 300 mm cube -- the north_star's "within 1e-3 mm" figure, the one tests/test_full_size.py holds to its bar.
    Weights: He initialisation, BatchNorm parameters perturbed, running statistics = the batch statistics of a calibration batch
    (a net whose running statistics normalise its activations, as a trained one's do), last layer scaled to O(0.3) outputs."""
    from hipdp import runtime as R
    from net.resnet import ResNet, ResNetParams
    from oracle import nets
    R.set_default_runtime(rt)
    B, seed = frames, 23455
    net = ResNet(np.random.RandomState(seed), cfgParams=ResNetParams(type=1, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=14, nDims=3))
    onet = nets.build_resnet(type=1, wIn=size, hIn=size, batchSize=B, numJoints=14, nDims=3)
    P = nets.perturb_bn(nets.init_params(onet, np.random.RandomState(seed), np.float32), onet, np.random.RandomState(seed + 1))
    xc = nets.synthetic_crops(np.random.RandomState(seed + 2), B, size, size, np.float64)
    out_c, cache = nets.forward(onet, nets.cast_params(P, np.float64), xc, True)
    for i, l in enumerate(onet['layers']):
        if l['kind'] == 'bn':
            P[i][2], P[i][3] = cache[i][1].astype(np.float32), cache[i][2].astype(np.float32)
    last = max(P)
    P[last][0] = (P[last][0] * (0.3 / max(1e-6, np.abs(out_c).max()))).astype(np.float32)
    for i, l in enumerate(net.layers):
        if i in P:
            for prm, v in zip(l.params + l.params_nontrained, P[i]):
                prm.set_value(v)
    x = nets.synthetic_crops(np.random.RandomState(5), frames, size, size, np.float32)
    net.setDeterministic()
    out = net.computeOutput(x, dp=False)
    ref = nets.compute_output(onet, nets.cast_params(P, np.float64), x.astype(np.float64))
    d = np.abs(out.astype(np.float64) - ref) * 150.0
    return dict(max=float('%.3g' % d.max()), mean=float('%.3g' % d.mean()), frames=frames, bar=1e-3,
                against='float64 NumPy oracle (oracle/nets.py), 14 joints through the PCA-prior layer, 300 mm cube')


def _cpu_leg_child(argv):
    """`bench.py --cpu-leg batch size threads min_steps max_steps budget_s`: the child process of cpu_baseline -- started with
    OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES in its environment (they are read when the OpenMP runtime starts, so they cannot be set
    inside the parent) -- times the oracle's PyTorch-CPU train step and prints the per-step seconds as one JSON line."""
    batch, size, threads, min_steps, max_steps = (int(v) for v in argv[:5])
    budget = float(argv[5])
    import torch
    torch.set_num_threads(threads)
    import torch.nn.functional  # noqa: F401
    times = _time_cpu_steps(batch, size, min_steps, max_steps, budget)[3]
    print(json.dumps(dict(times=times, threads=int(torch.get_num_threads()))))


def _cpu_leg(batch, size, threads, min_steps, max_steps, budget_s):
    """Per-step seconds of the CPU train step on `threads` OpenMP threads pinned to neighbouring cores, measured in a child process."""
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND='close', OMP_PLACES='cores',
               OMP_WAIT_POLICY='active')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-leg', str(batch), str(size), str(threads), str(min_steps),
                        str(max_steps), str(budget_s)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if r.returncode != 0:
        raise RuntimeError('cpu leg failed: ' + r.stderr.decode(errors='replace')[-1500:])
    return json.loads(r.stdout.decode().strip().splitlines()[-1])['times']


def cpu_baseline(batch, size, budget_s=45.0):
    """The oracle's PyTorch-CPU restatement of the identical fp32 train step on bounded samples, as legs of
    the SAME JSON line.  PyTorch-CPU does NOT get faster with every core on these small convolutions (round 4 measured 19.8 crops/s
    on 128 threads against 38.3 on ONE thread at batch 16), so the headline leg is the best of {all threads, 32, 16, 8} at the
    benchmarked batch -- chosen on the median of 2-3 probe steps each (all candidates are printed), then timed for the rest of the
    budget -- and the all-thread and one-thread figures are reported beside it, plus the NumPy augmentation in 1 and 8 processes (the
    reference's para_num_proc = 8, nettrainer.py:59).  With forward_parity_mm the only place this script touches oracle/.
    Round 6: every timing runs in a CHILD process started with OMP_NUM_THREADS = the candidate count,
    OMP_PROC_BIND=close and OMP_PLACES=cores -- the threads of a candidate sit on neighbouring cores and stay there.  Unpinned, the same
    16 threads measured 72 and 383 crops/s inside one process (round 5: placement decided per parallel region).  `value` is the median of
    the longer run; the probe's median at the chosen count is reported beside it."""
    import torch
    threads = int(torch.get_num_threads())
    extra = {}
    probes = {}
    for t in sorted(set([threads, min(threads, 32), min(threads, 16), min(threads, 8)]), reverse=True):
        probes[t] = float(np.median(_cpu_leg(batch, size, t, 2, 3, 6.0)))       # 3 steps (2 where a step takes > 3 s: the oversubscribed counts)
    best = min(probes, key=probes.get)
    left = max(10.0, budget_s - 4.0 * sum(probes.values()))
    times = _cpu_leg(batch, size, best, 5, 12, left)
    med_s, mean_s, n = float(np.median(times)), float(np.mean(times)), len(times)
    cand = {str(t): round(batch / v, 2) for t, v in sorted(probes.items())}
    out = dict(value=round(batch / med_s, 2), unit='depth-crops/sec', cores=best, kind='port', candidates_crops_per_s=cand,
               medians_s_per_step=dict(probe=round(probes[best], 3), run=round(med_s, 3)),
               pinning='child process per timing: OMP_NUM_THREADS=<cores>, OMP_PROC_BIND=close, OMP_PLACES=cores',
               sample='%d train steps of batch %d (PyTorch-CPU fp32 restatement, not Theano) on %d of %d threads pinned to neighbouring cores '
                      '(the fastest of %s, each on the median of 2-3 probe steps in a process of its own): median %.2f s/step, mean %.2f; '
                      'value = that median' % (n, batch, best, threads, sorted(probes), med_s, mean_s))
    extra['cpu_baseline_all_threads'] = dict(value=round(batch / probes[threads], 2), unit='depth-crops/sec', cores=threads, kind='port',
                                             sample='median of 2-3 train steps of batch %d after a warm-up step: %.2f s' % (batch, probes[threads]))
    b1 = min(16, batch)
    t1 = _cpu_leg(b1, size, 1, 2, 6, 15.0)
    extra['cpu_baseline_1thread'] = dict(value=round(b1 / float(np.median(t1)), 3), unit='depth-crops/sec', cores=1, kind='port',
                                         sample='%d train steps of batch %d of the same graph on ONE thread: median %.2f s/step' % (
                                             len(t1), b1, float(np.median(t1))))
    if size == 128:
        extra['cpu_augment_baseline'] = [cpu_augment_baseline(1), cpu_augment_baseline(8)]
    return out, extra


def trainer_ms_per_minibatch(rt, B, S, epochs=2, minibatches=32):
    """What the reference's class API delivers on this build: PoseRegNetTrainer.train() (per-epoch re-augmentation of the resident set,
    one train_model + cost read-back per minibatch, validation every epoch, weightreg_factor = 0 as in the pose-regression mains -- rounds 3-4 ran
    this leg with the trainer's default regulariser 0.001, 0.17 ms per step that the engine step it was compared with did not have)
    on epochs of `minibatches` minibatches of resident crops, WITHOUT the per-epoch pickle snapshot (net.save is a no-op here;
    tools/trainer_throughput.py measures the loop WITH the snapshots, which a writer process takes off the training thread).  Round 5: epochs of
    32 minibatches instead of 8 -- the per-epoch work (validation pass, re-augmentation, bookkeeping: ~2 ms) spread over 8 minibatches was most
    of the "+6 % of the class API" the round-4 line showed; NYU's epochs are 568 minibatches long."""
    import contextlib
    import io
    import tempfile
    import torch
    from hipdp import runtime as R
    from net.resnet import ResNet, ResNetParams
    from tools import synth
    from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams
    from util.handdetector import HandDetector
    R.set_default_runtime(rt)
    N, J = minibatches * B, 14
    di, imgs, coms, cubes, Ms, gts, pca_mean, pca_comp = synth.crop_db(N, S, J)
    rng = np.random.RandomState(23455)

    class Proj(object):
        mean_, components_ = pca_mean, pca_comp

        @staticmethod
        def transform(x):
            return (np.asarray(x, np.float64) - pca_mean) @ pca_comp.T

    labels = (gts / (cubes[:, 2] / 2.)[:, None, None]).astype(np.float32)
    embed = Proj.transform(labels.reshape(N, -1)).astype(np.float32)
    net = ResNet(rng, cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=B, numJoints=1, nDims=30))
    net.save = lambda filename: None
    p = PoseRegNetTrainerParams()
    p.batch_size, p.learning_rate, p.force_macrobatch_reload, p.para_augment = B, 1e-3, True, True
    p.weightreg_factor = 0.0          # as the pose-regression mains set it (/root/reference/src/main_nyu_posereg_embedding.py:104); the timed engine step has none either
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': ['com', 'rot', 'none'], 'proj': Proj,
                                                             'hd': HandDetector(imgs[0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        tr = PoseRegNetTrainer(net, p, rng, tmp, runtime=rt)
        tr.setData(imgs[:, None], embed, imgs[:B, None], embed[:B])
        tr.addStaticData({'val_data_y3D': labels[:B]})
        tr.addStaticData({'pca_data': pca_comp, 'mean_data': pca_mean})
        tr.addManagedData({'train_data_cube': cubes, 'train_data_com': coms, 'train_data_M': Ms, 'train_gt3Dcrop': gts})
        tr.compileFunctions()
        tr.train(n_epochs=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        costs, _, _ = tr.train(n_epochs=epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return round(dt / len(costs) * 1e3, 3)


def _device_id(torch, emu):
    """What tells two GPUs apart in the line: the HIP device index plus the PCI bus id (uuid where PyTorch exposes one)."""
    if emu or not torch.cuda.is_available():
        return 'emulator:%d' % os.getpid()
    i = torch.cuda.current_device()
    pr = torch.cuda.get_device_properties(i)
    tag = getattr(pr, 'uuid', None) or getattr(pr, 'pci_bus_id', None)
    return 'cuda:%d %s%s' % (i, pr.name, (' ' + str(tag)) if tag is not None else '')


def _rccl_version(torch):
    try:
        return '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                      # noqa: BLE001  (not built / not queryable without a GPU)
        return None


def forward_only_leg(rt, torch, net, B, S, x_host, steps=50, warmup=10):
    """`computeOutput`'s compiled function on the device (/root/reference/src/net/netbase.py:257-310, the reference's second profiled
    call site): the deterministic forward (stored BatchNorm statistics) of the SAME net on one resident batch, timed with HIP events
    on the launch stream.  Not the headline; reported beside it ("fwd only; fwd+bwd+ADAM")."""
    from hipdp import engine
    ev_eng = engine.CompiledNet(net, train=False, runtime=rt)
    ev_eng.set_input(x_host)
    for _ in range(warmup):
        ev_eng.fwd.run(rt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(steps):
        ev_eng.fwd.run(rt)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    out = ev_eng.out.buf.get()
    names, groups = {}, {}
    for o in ev_eng.fwd.launches():
        m = o.meta or dict(kernel=o.name, flops=0.0, bytes=0.0)
        names[m['kernel']] = names.get(m['kernel'], 0) + 1
        g = groups.setdefault(m['kernel'], dict(ops=[], flops=0.0, bytes=0.0))
        g['ops'].append(o)
        g['flops'] += m['flops']
        g['bytes'] += m['bytes']
    # the leg's own roofline: its dominant kernel family re-issued back to back between two HIP events (as the train step's families are)
    from hipdp import ops
    fam = {}
    for kname, g in groups.items():
        plan = ops.NativePlan(rt, [(o, False) for o in g['ops']], mode='native')
        plan.run(rt)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(torch.cuda.current_stream())
        for _ in range(5):
            plan.run(rt)
        f1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        fam[kname] = f0.elapsed_time(f1) / 5
    dom = max(fam, key=fam.get)
    g = groups[dom]
    t = fam[dom] * 1e-3
    # HBM bytes per launch of that family from the counter passes over the deterministic forward (tools/refresh_profiles.sh), from a
    # file measured on THESE kernel sources only
    traffic, traffic_src = None, None
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic_forward.json')), reverse=True):
        try:
            with open(fn) as fh:
                pj = json.load(fh)
        except (OSError, ValueError):
            continue
        kk = [k for k in pj if not k.startswith('_') and dom.split('_mfma')[0] in k]
        if pj.get('_csrc_sha16') == csrc_sha16() and kk and B == 128 and S == 128:
            traffic = round(sum(pj[k]['bytes_per_launch'] * pj[k]['launches_per_step'] for k in kk) / sum(pj[k]['launches_per_step'] for k in kk))
            traffic_src = 'profiles/' + os.path.basename(fn).replace('.json', '.txt')
            break
    roof = dict(kernel=dom, launches_per_batch=len(g['ops']), avg_launch_us=round(t / len(g['ops']) * 1e6, 2), share_of_batch=round(fam[dom] / sum(fam.values()), 3),
                traffic=traffic, traffic_source=traffic_src, algorithmic_bytes=round(g['bytes'] / len(g['ops'])),
                mfma=dict(achieved=round(g['flops'] / t / 1e12, 2), peak=PEAK_MFMA_F32 / 1e12, unit='TFLOP/s', frac=round(g['flops'] / t / PEAK_MFMA_F32, 4)),
                hbm=dict(achieved=round(g['bytes'] / t / 1e9, 1), peak=PEAK_HBM / 1e9, unit='GB/s', frac=round(g['bytes'] / t / PEAK_HBM, 4)))
    # one frame at a time (the reference's realtime use, netbase.py:286-310 with a batch of one; the paper quotes 30 fps): the same
    # architecture compiled for batch 1, latency per frame with the frame resident
    from net.resnet import ResNet, ResNetParams
    net1 = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=1, numJoints=1, nDims=30))
    e1 = engine.CompiledNet(net1, train=False, runtime=rt)
    e1.set_input(x_host[:1])
    for _ in range(warmup):
        e1.fwd.run(rt)
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(torch.cuda.current_stream())
    for _ in range(steps):
        e1.fwd.run(rt)
    g1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return dict(value=round(B / (ms * 1e-3), 1), unit='depth-crops/sec', ms_per_batch=round(ms, 4), batch=B, launches=len(ev_eng.fwd),
                launches_by_family=names, finite=bool(np.isfinite(out).all()), roofline=roof,
                single_frame_ms=round(g0.elapsed_time(g1) / steps, 4),
                mode='deterministic forward (stored BatchNorm statistics), inputs resident, HIP events over %d batches; single_frame_ms: the '
                     'same net compiled for a batch of one' % steps)


def _mfma_busy_from_profiles(kernel_family, B, S, args):
    """`roofline.mfma_busy`: the matrix-core busy fraction of the dominant family's kernels, SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES
    (launch-weighted), from the rocprofv3 --pmc pass of tools/refresh_profiles.sh over this workload.  Counters cannot be read inside
    this process; like `traffic` the figure is taken only from a file measured on THESE kernel sources (`_csrc_sha16`)."""
    import glob
    if not (B == 128 and S == 128 and args.dtype == 'f32' and args.workload == 'train'):
        return {}
    sha = csrc_sha16()
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_busy.json')), reverse=True):
        try:
            with open(fn) as fh:
                pj = json.load(fh)
        except (OSError, ValueError):
            continue
        fam = pj.get('families', {}).get(kernel_family)
        if pj.get('_csrc_sha16') == sha and fam:
            return dict(mfma_busy=fam['mfma_busy'], mfma_busy_source='profiles/' + os.path.basename(fn) + ' (kernel sources ' + sha + ')')
    return dict(mfma_busy=None)


def _sub_bench(extra_args, timeout=900):
    """One more configuration of this script in a process of its own (own device allocations, own DPP_BF16): its JSON line."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--headline-only', '--no-cpu-baseline', '--no-trainer'] + extra_args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        return dict(error=r.stderr.decode(errors='replace')[-800:])
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def _pmc_step_bytes(tag, sha):
    """HBM bytes per step of a non-headline configuration from the rocprofv3 --pmc passes of tools/refresh_profiles.sh
    (profiles/r*_hbm_traffic_<tag>.json), only from a file measured on THESE kernel sources."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic_%s.json' % tag)), reverse=True):
        try:
            with open(fn) as fh:
                pj = json.load(fh)
        except (OSError, ValueError):
            continue
        if pj.get('_csrc_sha16') == sha and pj.get('_total_bytes_per_step'):
            return float(pj['_total_bytes_per_step']), 'profiles/' + os.path.basename(fn)
    return None, None


def config5_leg(steps=10, warmup=3):
    """BASELINE.json configs[4] on this GPU, driver-timed: the 256x256 bf16 train step, bs128 -- bf16-stored
    activations / gradients, bf16 MFMA operands wherever the plan has them.  `bf16_mfma_flop_share` = the share of the step's MFMA FLOPs
    issued as v_mfma_f32_16x16x32_bf16; `hbm_frac` = PMC bytes per step / step time / 8 TB/s (null without a fingerprint-matching
    counter file)."""
    j = _sub_bench(['--size', '256', '--dtype', 'bf16', '--steps', str(steps), '--warmup', str(warmup)])
    if 'error' in j:
        return j
    nbytes, src = _pmc_step_bytes('256_bf16', j['config']['kernel_sources'])
    out = dict(ms_per_step=j['ms_per_step'], value=j['value'], unit=j['unit'], steps=steps, warmup=warmup, dtype='bf16',
               workload=j['config']['workload'], launches=j['config']['launches'], final_cost=j['config']['final_cost'],
               hip_event_ms_per_step=j['config']['hip_event_ms_per_step'],
               bf16_mfma_flop_share=j['config'].get('bf16_mfma_flop_share'),
               step_flop_rate_tflops=round(j['value'] * FLOP_PER_CROP[256] / 1e12, 1),
               step_bf16_mfma_frac=round(j['value'] * FLOP_PER_CROP[256] / PEAK_MFMA_BF16, 4),
               hbm_bytes_per_step=nbytes, hbm_source=src,
               hbm_frac=round(nbytes / (j['ms_per_step'] * 1e-3) / PEAK_HBM, 4) if nbytes else None,
               roofline=j.get('roofline'))
    return out


def config3_leg(rt, torch, iters=200):
    """BASELINE.json configs[2]: ICVL geometry (16 joints, 250 mm cube), the fused online augmentation (rot / scale / trans / CoM jitter +
    labels + PCA projection) as ONE launch over 256 crops, timed alone -- replayed `iters` times as one hipGraph chain between two HIP
    events (includes the kernel boundary) -- and the bs256 train step with that augmentation in it (a process of its own)."""
    from hipdp import ops
    from hipdp.augmenter import MODE_CODE, camera_tuple
    from tools import synth
    B, J, E, S = 256, 16, 30, 128
    di, imgs, coms, cubes, Ms, gts, pca_mean, pca_comp = synth.crop_db(B, S, J, dataset='icvl', cube=(250., 250., 250.))
    f32 = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))       # noqa: E731
    modes = ['com', 'rot', 'sc', 'none']
    table = rt.upload(np.array([MODE_CODE[m] for m in modes], np.int32))
    x_out, y_out = rt.alloc((B, S, S)), rt.alloc((B, E))
    st = ops.AugmentState(rt, B, seed=1234)
    (launch,) = st.ops(f32(imgs), f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), J, S, camera_tuple(di), x_out, y_out, mode_table=table,
                       n_modes=len(modes), pca_mean=f32(pca_mean), pca_comp=f32(pca_comp), E=E)
    plan = ops.NativePlan(rt, [(launch, False)] * iters, mode='graph1')
    plan.run(rt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(3):
        plan.run(rt)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * iters)
    out = dict(augment_us_bs256=round(us, 2), augment_crops_per_s=round(B / (us * 1e-6), 0),
               augment_hbm_frac=round(B * 2.0 * S * S * 4 / (us * 1e-6) / PEAK_HBM, 4), modes=modes, joints=J,
               finite=bool(np.isfinite(x_out.get()).all() and np.isfinite(y_out.get()).all()),
               algorithmic_bytes_per_crop=2 * S * S * 4)
    j = _sub_bench(['--batch', '256', '--steps', '20', '--warmup', '5'])
    if 'error' in j:
        out['train_step'] = j
    else:
        out.update(ms_per_step_bs256=j['ms_per_step'], value_bs256=j['value'], unit=j['unit'])
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run (one per GPU, RCCL)."""
    import torch
    emu = os.environ.get('DPP_BENCH_EMU') == '1'
    shared = os.environ.get('DPP_DIST_BACKEND') == 'gloo'          # control-flow runs: several ranks may share one GPU
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not emu and have < (1 if shared else n):
        sys.stderr.write('bench.py: --gpus %d needs %d visible MI355X GPUs, PyTorch-ROCm sees %d\n' % (n, n, have))
        sys.exit(2)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=128, help='crops per GPU')
    ap.add_argument('--size', type=int, default=128, help='crop side: 128 (configs 2-4) or 256 (config 5 stress)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help='bf16: bf16 MFMA operands / fp32 accumulation in FC1 and the 3x3 convolutions (config 5)')
    ap.add_argument('--no-augment', action='store_true')
    ap.add_argument('--augment-inline', action='store_true',
                    help='augment the minibatch at the start of its own step on the main stream (default: the previous step does it '
                         'on the gradient-branch stream while ADAM runs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--headline-only', action='store_true', help='the timed steps and the roofline only: none of the extra legs (forward_only, '
                    'no_augment, config3, config5); what the extra legs themselves run')
    ap.add_argument('--no-trainer', action='store_true', help="skip config.trainer_ms_per_minibatch (a short PoseRegNetTrainer.train() run)")
    ap.add_argument('--sync-bn', action='store_true', help='all-gather BatchNorm partial statistics across ranks (parity mode)')
    ap.add_argument('--launch', choices=['native', 'python', 'graph', 'graph1'], default=None,
                    help='native = the step is one C call that issues every launch on two HIP streams (default); python = one '
                         'ctypes call per launch; graph / graph1 = explicit hipGraph with two lanes / one chain')
    ap.add_argument('--profile-ops', action='store_true', help='print the per-kernel-family time table to stderr')
    ap.add_argument('--workload', choices=['train', 'cascade'], default='train',
                    help="train (default): BASELINE configs[1]; cascade: configs[4] on this GPU -- every step crops its minibatch from "
                         "640x480 frames through the CoM-refinement cascade (crop -> CoM -> ScaleNet -> crop at --size) and trains the "
                         "ResNet on it (use with --size 256 [--dtype bf16])")
    ap.add_argument('--allow-ablation', action='store_true',
                    help='run although an ablation variable (DPP_WHATIF_SKIP) is set: the timed plan then misses launches and its '
                         'results are wrong; tools/whatif.sh only')
    args = ap.parse_args()
    ablation = sorted(k for k in ABLATION_ENV if os.environ.get(k))
    if ablation and not args.allow_ablation:
        raise SystemExit('bench.py: %s is set -- an ablation that drops launches from the timed plan; refusing to report a number '
                         '(pass --allow-ablation for tools/whatif.sh)' % ', '.join(ablation))
    knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith('DPP_') and k not in PLUMBING_ENV}

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus)
    if args.launch is not None:
        os.environ['DPP_LAUNCH_MODE'] = args.launch
    if args.dtype == 'bf16':
        os.environ['DPP_BF16'] = '1'

    import torch
    from hipdp import engine, ops, parallel
    from net.resnet import ResNet, ResNetParams

    emu = os.environ.get('DPP_BENCH_EMU') == '1'      # CPU-tier control-flow test of the multi-rank path (tests/test_bench_cli.py)
    rank, world = parallel.init_from_env('gloo' if emu else 'nccl')          # "nccl" is RCCL on ROCm; one rank per GPU
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = dist.get_backend()
        if backend != 'nccl' and os.environ.get('DPP_DIST_BACKEND') != backend and not emu:
            raise SystemExit('bench.py: multi-GPU runs use RCCL (backend nccl), got %s' % backend)
    elif not emu:
        torch.cuda.set_device(0)

    if emu:
        from tests.emu.emu_runtime import EmuRuntime
        rt = EmuRuntime()
        sync_dev = lambda: None          # noqa: E731
    else:
        from hipdp.runtime import TorchHipRuntime
        rt = TorchHipRuntime()
        sync_dev = torch.cuda.synchronize
    B, S = args.batch, args.size
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=S, hIn=S, batchSize=B,
                                                                      numJoints=1, nDims=30))
    dp = parallel.DataParallel(rt, sync_bn=args.sync_bn) if world > 1 else None
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), dp=dp)

    # ---- device-resident data: rank r owns samples [r*NDB, (r+1)*NDB) of the global set ----
    from hipdp.augmenter import camera_tuple
    from tools import synth
    NDB, J = (8 if not emu else 2) * B, 14
    f32 = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))       # noqa: E731
    x_out = eng.x_in.buf.reshape(B, S, S)
    nsl = NDB // B
    step_plans = {}
    aug_plans = {}
    if args.workload == 'cascade':
        # config 5: the minibatch of every step comes out of the refinement cascade (hipdp/cascade.py): 128 frames of 640x480 ->
        # docom crop -> ScaleNet (3 scales, deterministic) -> refined centre -> crop at SxS + 30-D labels, then the train step
        from hipdp.cascade import CascadeCropper
        from net.scalenet import ScaleNet, ScaleNetParams
        di = synth.importer_of('nyu')
        H, W = (480, 640) if not emu else (120, 160)
        frames, coms0, gt3d = synth.depth_frames(np.random.RandomState(23455 + rank), NDB, di, H, W, (300., 300., 300.), J)
        rnet = ScaleNet(np.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, nChan=1, wIn=128, hIn=128, batchSize=B, resizeFactor=2,
                                                                             numJoints=1, nDims=3))
        rnet.setDeterministic()
        rng = np.random.RandomState(99)

        class Proj(object):
            mean_ = rng.normal(0, 0.05, J * 3)
            components_ = np.linalg.qr(rng.normal(size=(J * 3, 30)))[0].T

        pca_comp = np.asarray(Proj.components_, np.float32)
        db = dict(frames=f32(frames), com=f32(coms0), cube=f32(np.tile(np.float32((300., 300., 300.)), (NDB, 1))), gt=f32(gt3d))
        imgs = None
        # The cascade never touches x_in / y_in: it crops into staging buffers, so that the cascade of step i + 1 runs on the gradient
        # branch BESIDE the forward pass of step i (engine.step_plan(early=)) and only two copies (34 MB) sit under the ADAM update.
        x_stage, y_stage = rt.alloc((B, S, S), zero=False), rt.alloc((B, 30), zero=False)
        stage_copy = ops.Plan('cascade_to_input')
        stage_copy.add(ops.copy2d(rt, x_stage.reshape(B * S * S), S * S, x_out.reshape(B * S * S), S * S, B, S * S, name='load_crops'))
        stage_copy.add(ops.copy2d(rt, y_stage.reshape(B * 30), 30, eng.y_in.reshape(B * 30), 30, B, 30, name='load_labels'))
        for sl in range(nsl):
            o = sl * B
            cc = CascadeCropper(rt, di, rnet, B, H, W, dsize=S, frames=db['frames'].view(o * H * W, (B, H, W)), coms=db['com'].view(o * 3, (B, 3)),
                                cubes=db['cube'].view(o * 3, (B, 3)), out=x_stage, gt3d=db['gt'].view(o * J * 3, (B, J, 3)), J=J, proj=Proj,
                                out_y=y_stage)
            aug_plans[sl] = cc.plan
    else:
        di, imgs, coms, cubes, Ms, gts, pca_mean, pca_comp = synth.crop_db(NDB, S, J, seed=23455 + rank)
        db = dict(img=f32(imgs), com=f32(coms), cube=f32(cubes), M=f32(Ms.reshape(NDB, 9)), gt=f32(gts))
        pm, pc = f32(pca_mean), f32(pca_comp)
        table = rt.upload(np.array([1, 2, 0], np.int32))                     # aug_modes = ['com', 'rot', 'none']
        camt = camera_tuple(di)

        def slice_views(sl):
            o = sl * B
            return (db['img'].view(o * S * S, (B, S, S)), db['com'].view(o * 3, (B, 3)), db['cube'].view(o * 3, (B, 3)),
                    db['M'].view(o * 9, (B, 9)), db['gt'].view(o * J * 3, (B, J, 3)))

        # device-resident draw counter of the augmentation RNG: draws are keyed by (seed, step, GLOBAL sample index), so what a
        # sample gets does not depend on how many GPUs the global minibatch is spread over
        aug = ops.AugmentState(rt, B, seed=1234, sample0=rank * B, global_batch=world * B)
        for sl in range(nsl):
            im, co, cu, mm, gt = slice_views(sl)
            p = ops.Plan('augment')
            if args.no_augment:
                p.add(ops.copy2d(rt, im.reshape(B * S * S), S * S, x_out.reshape(B * S * S), S * S, B, S * S, name='load_crops'))
            else:
                for o in aug.ops(im, co, cu, mm, gt, J, S, camt, x_out, eng.y_in, mode_table=table, n_modes=3, pca_mean=pm, pca_comp=pc, E=30):
                    p.add(o)
            aug_plans[sl] = p
    # Step i trains on slice i % nsl.  Pipelined (default): its minibatch was augmented into x_in / y_in by step i - 1, on the gradient
    # branch under that step's ADAM update (engine.step_plan(prefetch=)), and step i does the same for slice i + 1 -- the reference's
    # background augmentation workers (nettrainer.py:601-628) as one kernel on an idle stream; every step still launches exactly one
    # augmentation.  --augment-inline: the augmentation of slice i is the first launch of step i, on the main stream.
    pipelined = not args.augment_inline
    if args.workload == 'cascade' and not pipelined:
        for sl in range(nsl):
            aug_plans[sl] = ops.Plan.concat('cascade_inline', [aug_plans[sl], stage_copy])
    for sl in range(nsl):
        if pipelined and args.workload == 'cascade':
            step_plans[sl] = (aug_plans[sl], eng.step_plan(prefetch=stage_copy, early=aug_plans[(sl + 1) % nsl]))
        elif pipelined:
            step_plans[sl] = (aug_plans[sl], eng.step_plan(prefetch=aug_plans[(sl + 1) % nsl]))
        else:
            step_plans[sl] = (aug_plans[sl], eng.step_plan(before=aug_plans[sl]))
    # Record every step plan NOW (host work only: recording a plan launches nothing).  A plan is recorded into its native form on its first
    # run, ~0.3 ms of host time during which the GPU idles; with 8 data slices and the driver's 5 warm-up steps, three of those recordings
    # used to fall into the timed region (round 5: 3.467 ms over 20 steps against 3.42 over 200).
    for sl in range(nsl):
        getattr(step_plans[sl][1], '_compile', lambda rt_: None)(rt)
    if pipelined:
        aug_plans[0].run(rt)              # the minibatch of the first step
        if args.workload == 'cascade':
            stage_copy.run(rt)
    if args.no_augment and args.workload == 'train':
        eng.y_in.set(np.random.RandomState(5).normal(0, 0.3, (B, 30)).astype(np.float32))
    eng.set_lr(1e-3)

    step_no = [0]

    def step():
        step_plans[step_no[0] % nsl][1].run(rt)
        step_no[0] += 1

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    sync_dev()
    # second clock: two HIP events on the launch stream around the same K steps (the driver's 5-s GPU sampler cannot see a 0.07 s
    # region; the device's own timestamps can)
    ev = None
    if not emu:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if ev is not None:
        ev[1].record(torch.cuda.current_stream())
    sync_dev()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    my_ms = elapsed / args.steps * 1e3
    hip_event_ms = ev[0].elapsed_time(ev[1]) / args.steps if ev is not None else None
    dist_info = dict(backend=None, world=1, devices=[_device_id(torch, emu)], rccl_version=_rccl_version(torch), per_rank_ms_per_step=[round(my_ms, 3)])
    if dist is not None:
        te = torch.tensor([elapsed], device='cpu' if emu or dist.get_backend() == 'gloo' else 'cuda', dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        # who took part: every rank's device and its own clock (a straggler shows here; under RCCL every rank must own another GPU)
        rows = [None] * world
        dist.all_gather_object(rows, (_device_id(torch, emu), round(my_ms, 3)))
        dist_info = dict(backend='rccl' if dist.get_backend() == 'nccl' else dist.get_backend(), world=world, devices=[r[0] for r in rows],
                         rccl_version=_rccl_version(torch), per_rank_ms_per_step=[r[1] for r in rows])
        if dist.get_backend() == 'nccl' and len(set(dist_info['devices'])) != world:
            raise SystemExit('bench.py: %d RCCL ranks on %d distinct GPUs (%s): one rank per GPU is the contract' % (
                world, len(set(dist_info['devices'])), dist_info['devices']))
    # how much of the step the main stream stood still for the gradient exchange (round 6): ten MORE steps, after the timed region,
    # with an event pair around every join with a collective (an event is a marker packet: not something the timed steps carry)
    exposed_ms = 0.0
    if dp is not None and not emu:
        dp.measure_exposed = True
        for _ in range(10):
            step()
        sync_dev()
        exposed_ms = dp.exposed_ms() / 10.0
        dp.measure_exposed = False
    seg = step_plans[0][1].segment_counts() if hasattr(step_plans[0][1], 'segment_counts') else (1, 0)
    dist_info.update(allreduce_ms_exposed=[round(exposed_ms, 4)], launch_segments_per_step=seg[0], host_issued_collectives_per_step=seg[1],
                     allreduce_schedule=getattr(dp, 'schedule', None))
    if dist is not None:
        rows = [None] * world
        dist.all_gather_object(rows, round(exposed_ms, 4))
        dist_info['allreduce_ms_exposed'] = rows
    cost = float(eng.cost.get()[0])
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- per-kernel-family timing with HIP events on the launch stream (after the timed region, rank 0) ----
    # Every family's launches (in plan order) are re-issued back to back by the C++ plan runner between TWO events on the launch
    # stream: the interval is the sum of the family's kernel durations plus the 1.6 us boundary behind each.  (Rounds 1-2 put an
    # event behind every launch; an event is a marker packet of its own that costs the queue ~5 us -- profiles/r02_join_probe.txt --
    # which inflated the average launch of the 10-20 us GEMMs by a third against the rocprofv3 kernel trace.)
    fam = {}
    if rank == 0 and not emu:
        stream = torch.cuda.current_stream()
        allops = [('aug', o) for o in step_plans[0][0].launches()] + eng.all_launches()
        groups = {}
        for ph, o in allops:
            m = o.meta or dict(kernel=o.name, flops=0.0, bytes=0.0)
            g = groups.setdefault(m['kernel'], dict(ops=[], flops=0.0, bytes=0.0))
            g['ops'].append(o)
            g['flops'] += m['flops']
            g['bytes'] += m['bytes']
        reps = 3
        for kname, g in groups.items():
            plan = ops.NativePlan(rt, [(o, False) for o in g['ops']], mode='native')
            plan.run(rt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                plan.run(rt)
            e1.record(stream)
            torch.cuda.synchronize()
            fam[kname] = dict(ms=e0.elapsed_time(e1) / reps, n=len(g['ops']), flops=g['flops'], bytes=g['bytes'])
        tot = sum(f['ms'] for f in fam.values())
        if args.profile_ops:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(allops) + 1)]        # per-launch detail: event after every launch
            evs[0].record(stream)
            for k, (_, o) in enumerate(allops):
                o(rt.stream)
                evs[k + 1].record(stream)
            torch.cuda.synchronize()
            for k, (ph, o) in enumerate(allops):          # (each interval carries ~5 us of event-marker cost)
                m = o.meta or dict(kernel=o.name, flops=0.0, bytes=0.0)
                t = evs[k].elapsed_time(evs[k + 1])
                print('OP %-4s %-22s %-22s %8.2f us %8.2f TF/s %8.1f GB/s' % (ph, o.name, m['kernel'], t * 1e3,
                      m['flops'] / (t * 1e-3) / 1e12 if t else 0, m['bytes'] / (t * 1e-3) / 1e9 if t else 0), file=sys.stderr)
            for kname, f in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
                print('%-26s n=%4d  %8.3f ms  %5.1f%%  %8.2f TFLOP/s  %8.1f GB/s' % (
                    kname, f['n'], f['ms'], 100 * f['ms'] / tot, f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] else 0,
                    f['bytes'] / (f['ms'] * 1e-3) / 1e9 if f['ms'] else 0), file=sys.stderr)
            print('sum of the family intervals (one stream, back to back): %.3f ms (step %.3f ms)' % (tot, ms), file=sys.stderr)

    if rank == 0:
        roof = None
        if fam:
            dom_name, dom = max(fam.items(), key=lambda kv: kv[1]['ms'])
            mfma = 'mfma' in dom_name
            peak_mfma = PEAK_MFMA_BF16 if 'bf16' in dom_name else PEAK_MFMA_F32
            per_launch_t = dom['ms'] * 1e-3 / dom['n']
            if mfma:
                ach = dom['flops'] / dom['n'] / per_launch_t / 1e12
                roof = dict(bound='mfma', kernel=dom_name, achieved=round(ach, 3), peak=peak_mfma / 1e12, unit='TFLOP/s',
                            frac=round(ach * 1e12 / peak_mfma, 4), traffic=None)
            else:
                ach = dom['bytes'] / dom['n'] / per_launch_t / 1e9
                roof = dict(bound='hbm', kernel=dom_name, achieved=round(ach, 1), peak=PEAK_HBM / 1e9, unit='GB/s',
                            frac=round(ach * 1e9 / PEAK_HBM, 4), traffic=None)
            # HBM bytes per launch of that kernel family from the PMC counters: they cannot be read inside this process, so the number
            # is the one measured with rocprofv3 --pmc on this workload (separate FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py).
            # The file records the fingerprint of the kernel sources it was measured on (`_csrc_sha16`); it is used only when that is the
            # fingerprint of the sources this run was built from -- otherwise `traffic` stays null (stale counters are not evidence).
            pmc_family = {'gemm_mfma_f32': ('gemm_kernel', 'gemm_ksplit_kernel', 'gemm_stream16_kernel', 'gemm_expand_kernel', 'gemm_rowstream_kernel',
                                            'fc_stream_kernel', 'fc_gemm_kernel', 'wgrad_stream_kernel', 'fc_wgrad_stream_kernel'),
                          'conv3x3_mfma_f32': ('conv3x3_kernel',), 'bn_bwd_apply': ('bn_bwd_apply_kernel',), 'adam': ('adam_kernel',)}.get(dom_name, ())
            sha = csrc_sha16()
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')), reverse=True):
                try:
                    with open(fn) as fh:
                        pj = json.load(fh)
                    ks = [k for k in pmc_family if k in pj]
                    if pj.get('_csrc_sha16') == sha and B == 128 and S == 128 and args.dtype == 'f32' and args.workload == 'train' and ks:
                        # launch-weighted mean over the kernels the family's launches run on
                        nl = sum(pj[k]['launches_per_step'] for k in ks)
                        roof['traffic'] = round(sum(pj[k]['bytes_per_launch'] * pj[k]['launches_per_step'] for k in ks) / nl)
                        roof['traffic_source'] = 'profiles/' + os.path.basename(fn).replace('.json', '.txt') + ' (kernel sources ' + sha + ')'
                        break
                except (OSError, ValueError):
                    pass
            roof['algorithmic_bytes'] = round(dom['bytes'] / dom['n'])
            roof['hbm_frac'] = round(dom['bytes'] / dom['n'] / per_launch_t / PEAK_HBM, 4)
            roof['launches_per_step'] = dom['n']
            roof['avg_launch_us'] = round(per_launch_t * 1e6, 2)
            roof['share_of_step'] = round(dom['ms'] / sum(f['ms'] for f in fam.values()), 3)
        flop = FLOP_PER_CROP.get(S, FLOP_PER_CROP[128] * (S / 128.0) ** 2)
        res = dict(metric='depth-crops/sec (NYU ResNet50 bs128) 1/2/4/8 GPU; mean 3D joint err (mm)', value=round(value, 1),
                   unit='depth-crops/sec', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic',
                   config=dict(workload=('NYU posereg_embedding ResNet (type 0, 30-D PCA prior) train step: fused augment + fwd + bwd + ADAM, '
                                         'bs%d/GPU %s, %dx%dx1 crops' if args.workload == 'train' else
                                         'NYU com_refine + posereg cascade: per step %d frames of 640x480 -> crop -> CoM -> ScaleNet refinement -> '
                                         'crop + 30-D labels, then the ResNet (type 0) train step fwd + bwd + ADAM, %s, %dx%dx1 crops')
                               % (B, 'fp32' if args.dtype == 'f32' else 'bf16 MFMA operands / fp32 accumulate', S, S),
                               global_batch=world * B, parallelism='dp%d' % world, augment=not args.no_augment, augment_pipelined=pipelined,
                               bn='sync (global batch statistics)' if (world > 1 and args.sync_bn) else 'local per-GPU batch statistics',
                               launches=eng.num_launches(), launch_mode=ops.LAUNCH_MODE,
                               step_mfma_frac=round(value / world * flop / PEAK_MFMA_F32, 4), final_cost=round(cost, 5),
                               knobs=knobs or None, ablation=ablation or None, kernel_sources=csrc_sha16(),
                               hip_event_ms_per_step=round(hip_event_ms, 4) if hip_event_ms is not None else None, dist=dist_info),
                   roofline=roof)
        if roof is not None:
            roof.update(_mfma_busy_from_profiles(roof.get('kernel'), B, S, args))
        if not emu:
            fl = [((o.meta or {}).get('kernel', ''), (o.meta or {}).get('flops', 0.0)) for _, o in eng.all_launches()]
            mf = sum(f for k, f in fl if 'mfma' in k)
            res['config']['bf16_mfma_flop_share'] = round(sum(f for k, f in fl if 'mfma_bf16' in k) / mf, 4) if mf else None
        if world == 1 and not emu and args.workload == 'train' and not args.headline_only:
            # the other legs BASELINE.md section 3 lists, after the timed region: the train step WITHOUT the augmentation launch, and the
            # deterministic forward alone (computeOutput's device function)
            plain = eng.step_plan()
            for _ in range(5):
                plain.run(rt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            for _ in range(args.steps):
                plain.run(rt)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            na_ms = e0.elapsed_time(e1) / args.steps
            res['no_augment'] = dict(value=round(B / (na_ms * 1e-3), 1), unit='depth-crops/sec', ms_per_step=round(na_ms, 4),
                                     mode='fwd + bwd + ADAM on a resident minibatch, no augmentation launch, HIP events over %d steps' % args.steps)
            xb = np.ascontiguousarray(eng.x_in.buf.get().reshape(B, 1, S, S), np.float32)
            res['forward_only'] = forward_only_leg(rt, torch, net, B, S, xb)
        if args.dtype == 'bf16':
            # forward error of the bf16 path against the fp32 path on the SAME (just trained) weights, deterministic mode (SURVEY.md
            # section 8(d) cfg 5: "parity reported vs fp32, not vs the 1e-3 mm bar").  The number to read is RELATIVE: the embedding of
            # this synthetic run is untrained and its PCA prior a random orthonormal basis, so "mm" obtained by pushing the difference
            # through that prior scale with an arbitrary embedding magnitude (reported as nominal_mm for continuity with rounds 2-3 only;
            # the physical figure is measured on a trained net by tests/test_configs.py / test_engine.py).
            xb = np.ascontiguousarray((imgs[:B] if imgs is not None else eng.x_in.buf.get()).reshape(B, 1, S, S), np.float32)
            o32 = engine.CompiledNet(net, train=False, runtime=rt, bf16=False).forward(xb)
            o16 = engine.CompiledNet(net, train=False, runtime=rt, bf16=True).forward(xb)
            dj = (o16.astype(np.float64) - o32.astype(np.float64)) @ pca_comp.astype(np.float64)
            d32 = o32.astype(np.float64)
            rel_rms = float(np.sqrt(((o16 - d32) ** 2).sum() / max(1e-30, (d32 ** 2).sum())))
            res['config']['bf16_forward_error_vs_fp32'] = dict(relative_embedding_max=round(float(np.abs(o16 - d32).max() / max(1e-30, np.abs(d32).max())), 5),
                                                                relative_embedding_rms=round(rel_rms, 5),
                                                                nominal_mm=dict(max=round(float(np.abs(dj).max() * 150.0), 4),
                                                                                mean=round(float(np.abs(dj).mean() * 150.0), 4),
                                                                                note='untrained embedding through a random orthonormal prior: not a physical error'))
        if world == 1 and not emu and args.workload == 'train' and S == 128 and args.dtype == 'f32' and not args.no_trainer:
            # the drop-in class API next to the engine's number (same GPU, same build, right after the timed region)
            res['config']['trainer_ms_per_minibatch'] = trainer_ms_per_minibatch(rt, B, S)
        if world == 1 and not emu and args.workload == 'train' and S == 128 and B == 128 and args.dtype == 'f32' and not args.headline_only:
            # the other single-GPU configurations of BASELINE.json, timed by whoever runs this script
            res['config3'] = config3_leg(rt, torch)
            res['config5'] = config5_leg()
        if world == 1 and not args.no_cpu_baseline and not emu and args.workload == 'train':
            res['cpu_baseline'], extra = cpu_baseline(B, S)
            res.update(extra)
            if args.dtype == 'f32':
                res['config']['forward_parity_mm'] = forward_parity_mm(rt, S)
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-leg':
        _cpu_leg_child(sys.argv[2:])
    else:
        main()
