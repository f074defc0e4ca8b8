#!/usr/bin/env python3
"""
bench.py -- the hot-path benchmark of BASELINE.json on MI355X.

One "step" = one pass of the DeepPrior++ hot path over one minibatch: fused crop augmentation (rot / CoM / none,
the mains' aug_modes) of 128 device-resident synthetic depth crops + PCA-prior label projection, then train_model
= forward + sum-squared-error loss + backward + the reference's ADAM of the ResNet pose regressor
(`ResNet(type=0, numJoints=1, nDims=30)`, 18.7 M parameters, fp32), i.e. BASELINE.json configs[1]
"NYU posereg_embedding ResNet-50, 30-dim PCA prior, bs128 fp32, 1xMI355X".  Inputs are resident in HBM before the
timed region.  With --gpus N the driver launches one rank per GPU (torch.distributed, RCCL): every rank processes its
own 128-crop shard (weak scaling) and the flat fp32 gradient buffer is all-reduced (mean) between backward and ADAM.

Prints ONE JSON line (rank 0).  `roofline` describes the kernel family with the largest share of the step time,
timed live with HIP events on the launch stream; `cpu_baseline` times the oracle's PyTorch-CPU restatement of the same
train step on the host cores (reported only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'deep-prior-pp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

FLOP_PER_CROP = 722.6e6        # fwd + dgrad + wgrad, SURVEY.md section 8(d)
PEAK_MFMA_F32 = 157.3e12       # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
PEAK_HBM = 8.0e12


def synthetic_db(n, J=14, seed=23455):
    """Device-resident training set: normalised crops + the per-sample geometry augmentCrop needs (NYU camera)."""
    from oracle import augment as A
    rng = np.random.RandomState(seed)
    cam = A.Camera.nyu()
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, n, cam, cube=(300., 300., 300.), joints=J)
    pca_mean = rng.normal(0, 0.05, J * 3).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(J * 3, 30)))
    return cam, imgs, coms, cubes, Ms, gts, pca_mean, q.T.astype(np.float32)


def cpu_baseline(batch, budget_s=12.0):
    """The oracle's PyTorch-CPU restatement of the identical fp32 train step, bounded sample."""
    import torch
    from oracle import nets, torch_ref
    onet = nets.build_resnet(type=0, batchSize=batch, numJoints=1, nDims=30)
    P = nets.init_params(onet, np.random.RandomState(23455), np.float32)
    tr = torch_ref.TorchTrainer(onet, P)
    rng = np.random.RandomState(3)
    x = torch.tensor(nets.synthetic_crops(rng, batch, 128, 128, np.float32))
    y = torch.tensor(rng.normal(0, 0.3, (batch, 30)).astype(np.float32))
    tr.step(x, y, 1e-3)                      # warm-up (thread pools, allocator)
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < budget_s and n < 8):
        tr.step(x, y, 1e-3)
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=round(batch / dt, 2), unit='depth-crops/sec', cores=int(torch.get_num_threads()), kind='port',
                sample='%d train steps of batch %d (PyTorch-CPU fp32 restatement, not Theano), %.2f s/step' % (n, batch, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--no-augment', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sync-bn', action='store_true', help='all-gather BatchNorm partial statistics across ranks (parity mode)')
    ap.add_argument('--launch', choices=['auto', 'eager', 'graph'], default='eager',
                    help='eager = every kernel launched from Python (two concurrent HIP streams); graph = replay captured hipGraphs; auto = probe both')
    ap.add_argument('--eager', action='store_true', help='same as --launch eager')
    ap.add_argument('--profile-ops', action='store_true', help='print the per-kernel-family time table to stderr')
    args = ap.parse_args()
    if args.eager:
        args.launch = 'eager'

    import torch
    from hipdp import engine, ops, parallel
    from hipdp.runtime import TorchHipRuntime
    from net.resnet import ResNet, ResNetParams

    rank, world = parallel.init_from_env('nccl')          # "nccl" is RCCL on ROCm; one rank per GPU
    if world > 1:
        import torch.distributed as dist
    else:
        dist = None
        torch.cuda.set_device(0)

    rt = TorchHipRuntime()
    B = args.batch
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=128, hIn=128, batchSize=B,
                                                                      numJoints=1, nDims=30))
    dp = parallel.DataParallel(rt, sync_bn=args.sync_bn) if world > 1 else None
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), dp=dp)

    # ---- device-resident data (different per rank: each rank owns its shard of the global minibatch) ----
    NDB, J = 8 * B, 14
    cam, imgs, coms, cubes, Ms, gts, pca_mean, pca_comp = synthetic_db(NDB, J, seed=23455 + rank)
    f32 = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))       # noqa: E731
    db = dict(img=f32(imgs), com=f32(coms), cube=f32(cubes), M=f32(Ms.reshape(NDB, 9)), gt=f32(gts))
    pm, pc = f32(pca_mean), f32(pca_comp)
    rec = rt.alloc(B * rt.lib.dpp_augment_record_bytes(), np.uint8)
    table = rt.upload(np.array([1, 2, 0], np.int32))                     # aug_modes = ['com', 'rot', 'none']
    camt = (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y)
    x_out = eng.x_in.buf.reshape(B, 128, 128)

    def slice_views(i):
        o = (i % (NDB // B)) * B
        return (db['img'].view(o * 128 * 128, (B, 128, 128)), db['com'].view(o * 3, (B, 3)), db['cube'].view(o * 3, (B, 3)),
                db['M'].view(o * 9, (B, 9)), db['gt'].view(o * J * 3, (B, J, 3)))

    step_no = [0]
    ctr_dev = rt.alloc(1, np.int64)              # device-resident draw counter of the augmentation RNG
    aug_cache = {}

    def augment_ops(i):
        sl = i % (NDB // B)
        if sl not in aug_cache:
            im, co, cu, mm, gt = slice_views(sl)
            if args.no_augment:
                aug_cache[sl] = None
            else:
                aug_cache[sl] = [ops.augment_prepare(rt, im, co, cu, mm, gt, B, J, 128, camt, rec, eng.y_in, mode_table=table, n_modes=3,
                                                     seed=1234 + rank, counter=0, pca_mean=pm, pca_comp=pc, E=30, counter_dev=ctr_dev),
                                 ops.augment_warp(rt, im, rec, B, 128, x_out), ops.counter_add(rt, ctr_dev, 1)]
        return aug_cache[sl]

    def augment(i):
        lst = augment_ops(i)
        if lst is None:
            rt.copy(x_out, slice_views(i % (NDB // B))[0])
            return []
        for o in lst:
            o(rt.stream)
        return lst

    if args.no_augment:
        eng.y_in.set(np.random.RandomState(5).normal(0, 0.3, (B, 30)).astype(np.float32))

    eng.set_lr(1e-3)

    def step_body(i):
        augment(i)
        eng.run_step_plans()          # with --gpus N the flat-gradient all-reduce (RCCL) is a step of the update plan

    # One hipGraph per resident data slice (pointers are baked into a graph; draw counter and ADAM step count live on the
    # device, so every replay sees fresh augmentation draws and the right bias correction).  Multi-GPU runs stay eager:
    # the RCCL all-reduce sits between backward and ADAM.
    graphs = {}
    mode = 'eager'
    if world == 1 and args.launch != 'eager':
        try:
            for sl in range(NDB // B):
                graphs[sl] = rt.capture(lambda sl=sl: step_body(sl))
            mode = 'hipgraph'
        except Exception as e:          # noqa: BLE001
            print('graph capture failed, running eager: %r' % (e,), file=sys.stderr)
            graphs = {}
    if graphs and args.launch == 'auto':
        # eager launches keep the two HIP streams genuinely concurrent; a replayed graph removes the host cost per launch.
        # Which wins depends on the driver: time a few untimed steps of each and keep the faster mode.
        def _probe(use_graph, n=6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for k in range(n):
                if use_graph:
                    graphs[k % (NDB // B)].replay()
                else:
                    step_body(k)
            torch.cuda.synchronize()
            return time.perf_counter() - t
        _probe(True, 2), _probe(False, 2)
        if _probe(False) < _probe(True):
            graphs, mode = {}, 'eager'

    def step():
        i = step_no[0]
        step_no[0] += 1
        if graphs:
            graphs[i % (NDB // B)].replay()
        else:
            step_body(i)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    cost = float(eng.cost.get()[0])
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- per-kernel-family timing with HIP events on the launch stream (after the timed region) ----
    fam = {}
    if rank == 0:
        stream = torch.cuda.current_stream()
        allops = eng.all_launches()
        aug_ops = augment(step_no[0])
        allops = [('aug', o) for o in aug_ops] + allops
        eng.set_lr(1e-3)
        reps = 3
        for r in range(reps):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(allops) + 1)]
            evs[0].record(stream)
            for k, (_, o) in enumerate(allops):
                o(rt.stream)
                evs[k + 1].record(stream)
            torch.cuda.synchronize()
            for k, (ph, o) in enumerate(allops):
                m = o.meta or dict(kernel=o.name, flops=0.0, bytes=0.0)
                f = fam.setdefault(m['kernel'], dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
                f['ms'] += evs[k].elapsed_time(evs[k + 1]) / reps
                if r == 0:
                    f['n'] += 1
                    f['flops'] += m['flops']
                    f['bytes'] += m['bytes']
        tot = sum(f['ms'] for f in fam.values())
        if args.profile_ops:
            # per-launch detail of the last repetition
            for k, (ph, o) in enumerate(allops):
                m = o.meta or dict(kernel=o.name, flops=0.0, bytes=0.0)
                t = evs[k].elapsed_time(evs[k + 1])
                print('OP %-4s %-22s %-22s %8.2f us %8.2f TF/s %8.1f GB/s' % (ph, o.name, m['kernel'], t * 1e3,
                      m['flops'] / (t * 1e-3) / 1e12 if t else 0, m['bytes'] / (t * 1e-3) / 1e9 if t else 0), file=sys.stderr)
            for kname, f in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
                print('%-26s n=%4d  %8.3f ms  %5.1f%%  %8.2f TFLOP/s  %8.1f GB/s' % (
                    kname, f['n'], f['ms'], 100 * f['ms'] / tot, f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] else 0,
                    f['bytes'] / (f['ms'] * 1e-3) / 1e9 if f['ms'] else 0), file=sys.stderr)
            print('sum of per-launch event intervals: %.3f ms (step %.3f ms)' % (tot, ms), file=sys.stderr)

    if rank == 0:
        dom_name, dom = max(fam.items(), key=lambda kv: kv[1]['ms'])
        mfma = 'mfma' in dom_name
        per_launch_t = dom['ms'] * 1e-3 / dom['n']
        if mfma:
            ach = dom['flops'] / dom['n'] / per_launch_t / 1e12
            roof = dict(bound='mfma', kernel=dom_name, achieved=round(ach, 3), peak=PEAK_MFMA_F32 / 1e12, unit='TFLOP/s',
                        frac=round(ach * 1e12 / PEAK_MFMA_F32, 4), traffic=None)
        else:
            ach = dom['bytes'] / dom['n'] / per_launch_t / 1e9
            roof = dict(bound='hbm', kernel=dom_name, achieved=round(ach, 1), peak=PEAK_HBM / 1e9, unit='GB/s',
                        frac=round(ach * 1e9 / PEAK_HBM, 4), traffic=None)
        # HBM bytes per launch of that kernel family from the PMC counters: they cannot be read inside this process, so the
        # number is the one measured with rocprofv3 --pmc on this same build and workload (separate FETCH_SIZE / WRITE_SIZE
        # passes, calibrated on adam_kernel; tools/pmc_summary.py -> profiles/r01_hbm_traffic.json)
        pmc_family = {'gemm_mfma_f32': 'gemm_kernel', 'conv3x3_mfma_f32': 'conv3x3_kernel', 'bn_bwd_apply': 'bn_bwd_apply_kernel',
                      'adam': 'adam_kernel'}.get(dom_name)
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')) as fh:
                pj = json.load(fh)
            if B == 128 and pmc_family in pj:
                roof['traffic'] = round(pj[pmc_family]['bytes_per_launch'])
                roof['traffic_source'] = 'profiles/r01_hbm_traffic.txt'
        except (OSError, ValueError):
            pass
        roof['algorithmic_bytes'] = round(dom['bytes'] / dom['n'])
        roof['hbm_frac'] = round(dom['bytes'] / dom['n'] / per_launch_t / PEAK_HBM, 4)
        roof['launches_per_step'] = dom['n']
        roof['avg_launch_us'] = round(per_launch_t * 1e6, 2)
        roof['share_of_step'] = round(dom['ms'] / sum(f['ms'] for f in fam.values()), 3)
        res = dict(metric='depth-crops/sec (NYU ResNet50 bs128) 1/2/4/8 GPU; mean 3D joint err (mm)', value=round(value, 1),
                   unit='depth-crops/sec', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='NYU posereg_embedding ResNet (type 0, 30-D PCA prior) train step: fused augment + fwd + bwd + ADAM, '
                                        'bs%d/GPU fp32, 128x128x1 crops' % B,
                               global_batch=world * B, parallelism='dp%d' % world, augment=not args.no_augment,
                               bn='sync (global batch statistics)' if (world > 1 and args.sync_bn) else 'local per-GPU batch statistics', launches=eng.num_launches(), launch_mode=mode,
                               step_mfma_frac=round(value / world * FLOP_PER_CROP / PEAK_MFMA_F32, 4), final_cost=round(cost, 5)),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(B)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
