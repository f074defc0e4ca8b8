"""Which stream does the end of the backward pass wait for?  Runs the bs128 ResNet train step with events recorded on the
main stream (after its last backward kernel) and on the parameter-gradient stream (after its last kernel), both just before
the join in front of the fused reduction, plus progress marks every 10 % of the backward plan."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/deep-prior-pp_amd')
import numpy as np, torch
from net.resnet import ResNet, ResNetParams
from hipdp import engine, ops
from hipdp.runtime import TorchHipRuntime

rt = TorchHipRuntime()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, batchSize=B, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
eng.set_lr(1e-3)
for _ in range(5):
    eng.run_step_plans()
torch.cuda.synchronize()
main = torch.cuda.current_stream(rt.device)
side = rt._side
E = lambda: torch.cuda.Event(enable_timing=True)


def run_bwd_marked():
    plan = eng.bwd
    last_join = max(i for i, (op, _) in enumerate(plan.ops) if isinstance(op, ops.Join))
    marks = []
    t_begin = E(); t_begin.record(main)
    n = len(plan.ops)
    for i, (op, on_side) in enumerate(plan.ops):
        if i == last_join:
            em, es = E(), E()
            em.record(main); es.record(side)
        if isinstance(op, ops.Fork):
            rt.side_wait_main()
        elif isinstance(op, ops.Join):
            rt.main_wait_side()
        else:
            op(rt.side_stream if on_side else rt.stream)
        if i % (n // 10) == 0:
            a, b = E(), E()
            a.record(main); b.record(side)
            marks.append((i, a, b))
    t_end = E(); t_end.record(main)
    return t_begin, em, es, t_end, marks


for rep in range(3):
    eng.fwd.run(rt)
    eng.lossplan.run(rt)
    t_begin, em, es, t_end, marks = run_bwd_marked()
    eng.upd.run(rt)
    torch.cuda.synchronize()
    print("backward: main chain done at %.3f ms, gradient branch done at %.3f ms, plan end (after reduce) %.3f ms" %
          (t_begin.elapsed_time(em), t_begin.elapsed_time(es), t_begin.elapsed_time(t_end)))
    print("  progress (plan op index: main ms / side ms): " +
          '  '.join("%d: %.2f/%.2f" % (i, t_begin.elapsed_time(a), t_begin.elapsed_time(b)) for i, a, b in marks))
