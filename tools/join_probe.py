"""What a cross-stream dependency costs on the critical path (MI355X): chains of ~8 us kernels issued by the native plan runner
(dpp_plan_run, the host stays ahead of the GPU) with (a) nothing else, (b) a fork per kernel (the side stream waits for main and
runs a tiny kernel; main never waits), (c) fork + join per kernel (main waits for the side kernel), (d) fork per kernel and ONE
join at the end.  Prints microseconds per main-chain kernel (HIP events around the whole plan)."""
import sys
import time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/deep-prior-pp_amd')
import numpy as np
import torch
from hipdp import ops
from hipdp.runtime import TorchHipRuntime

rt = TorchHipRuntime()
n = 24 << 20          # ~35 us per kernel: the host (3-4 us per API call) stays ahead in every mode
x, y = rt.alloc(n), rt.alloc(n)
st = rt.alloc(8)
N = 200


def plan(mode):
    p = ops.Plan('probe')
    for i in range(N):
        p.add(ops.scale(rt, x, y, n, a=1.0))
        if mode >= 1:
            p.fork()
            p.add(ops.adam_tick(rt, st), side=True)
        if mode == 2:
            p.join()
    if mode == 3:
        p.join()
    return p


main = torch.cuda.current_stream()
for mode, name in ((0, 'main chain only'), (1, 'fork per kernel, no join'), (2, 'fork + join per kernel'), (3, 'fork per kernel, one join at the end')):
    p = plan(mode)
    for rep in range(3):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main)
        t0 = time.perf_counter()
        p.run(rt)
        host = (time.perf_counter() - t0) * 1e6 / N
        e1.record(main)
        torch.cuda.synchronize()
    print('%-40s %6.2f us per main kernel (host issue %5.2f us)' % (name, e0.elapsed_time(e1) * 1e3 / N, host))
