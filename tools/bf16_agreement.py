#!/usr/bin/env python3
"""The un-pinned agreement statistics of the bf16 train step (tests/test_engine.py:bf16_gradients_vs_pinned_oracle) tensor by tensor, for
bisecting which launches move them:  python tools/bf16_agreement.py [size 256] [batch 32]   (uses oracle/: a test tool, not product code)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
from hipdp import engine  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from oracle import nets, torch_ref  # noqa: E402
from tests.pinning import device_grad_pins, device_masks, device_quant, device_store, store_agreement  # noqa: E402
from tests.test_engine import make_net  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rt = TorchHipRuntime()
net, onet, P = make_net(rt, 0, B, S, 1, 30, calib_batch=4)
rng = np.random.RandomState(8)
x = nets.synthetic_crops(rng, B, S, S, np.float32)
y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=True)
cost, out = eng.cost_and_grads(x, y)
quant = device_quant(eng, net)
store = device_store(eng, net)
own, own_g = {}, {}
gpins = device_grad_pins(eng, net)
torch_ref.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64), masks=device_masks(eng, net),
                         quant=quant, store=store, stored_out=own, grad_pins=gpins, stored_grads=own_g)
print('knobs', {k: v for k, v in os.environ.items() if k.startswith('DPP_')})
rows = []
for (kind, i), mine in own_g.items():
    pin = gpins[0][i] if kind == 'g' else gpins[1][i]
    rows.append((float((np.asarray(mine, np.float32) == pin).mean()), kind, i, net.layers[i].__class__.__name__, pin.shape))
for r in sorted(rows)[:12]:
    print('grad  %.5f %s %d %s %s' % r)
ag = store_agreement(store, own)
for i, a in sorted(ag.items(), key=lambda kv: kv[1])[:6]:
    print('store %.5f %d' % (a, i))
names = {}
for _, l in eng.all_launches():
    if 'bf16' in (l.meta or {}).get('kernel', ''):
        names.setdefault(l.name.rsplit('_', 1)[0], []).append(l.name.rsplit('_', 1)[-1])
print({k: len(v) for k, v in names.items()})
for (kind, i) in (('g', 44), ('dv', 43), ('g', 35), ('dv', 34)):
    if (kind, i) in own_g:
        pin = gpins[0][i] if kind == 'g' else gpins[1][i]
        mine = np.asarray(own_g[(kind, i)], np.float32)
        d = np.abs(mine - pin)
        print(kind, i, 'agree %.5f' % float((mine == pin).mean()), 'max|d| %.3e' % d.max(), 'max|pin| %.3e' % np.abs(pin).max(),
              'differing elements: median |pin| %.3e' % (np.median(np.abs(pin[mine != pin])) if (mine != pin).any() else 0.0))
    else:
        print(kind, i, 'not pinned / not bf16-stored')
for nm in ('dgrad1x1_46', 'dgrad1x1_37', 'conv1x1_46'):
    for _, l in eng.all_launches():
        if l.name == nm:
            d = l.keep[0]
            print(nm, 'variant', d.variant, 'M N K', d.M, d.N, d.K, 'store', d.store, 'prec', d.precision, 'actA', d.actA.mode, 'bn_x', bool(d.epi.bn_x), 'residual', bool(d.residual))
