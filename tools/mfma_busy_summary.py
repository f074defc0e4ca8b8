#!/usr/bin/env python3
"""Matrix-core busy fraction per kernel from a rocprofv3 SQ counter pass (counters only + --kernel-trace):
   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o run -- python tools/step_profile.py 3
   python tools/mfma_busy_summary.py <results.db> [--json profiles/rNN_mfma_busy.json] > profiles/rNN_mfma_busy.txt
north_star: "kernel choices are evidenced by rocprof MFMA utilisation".  Two normalisations are printed, because ROCm 7.2 ships no gfx950
section for the derived MfmaUtil metric (/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots"):
  busy/SQ     SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, both summed over their hardware instances (what VERDICT r4 asked for);
  busy/time   SQ_VALU_MFMA_BUSY_CYCLES / (traced kernel duration x 2.4 GHz x 1024 SIMDs): the fraction of all matrix pipes' cycles the
              kernel kept busy -- comparable with `achieved / peak` of the roofline (f32 16x16x4: 32 busy cycles per instruction).
The counter's unit is checked on a kernel of known arithmetic: busy cycles per MFMA instruction (SQ_INSTS_MFMA counts per wave) must be
32 for the f32 kernels."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import pretty  # noqa: E402

FAMILY = {'gemm_mfma_f32': ('gemm_kernel', 'gemm_ksplit_kernel', 'gemm_stream16_kernel', 'gemm_expand_kernel', 'gemm_rowstream_kernel',
                            'fc_stream_kernel', 'fc_gemm_kernel', 'wgrad_stream_kernel', 'fc_wgrad_stream_kernel'),
          'conv3x3_mfma_f32': ('conv3x3_kernel', 'conv3x3_stream_kernel'), 'conv3x3_wgrad_mfma_f32': ('conv3x3_wgrad_kernel', 'wgrad3_stream_kernel'),
          'stem_fwd_mfma_f32': ('stem_fwd_kernel',), 'stem_wgrad': ('stem_wgrad_kernel',), 'resblock_eval_mfma_f32': ('resblock_eval_kernel',)}
CLOCK_GHZ, SIMDS = 2.4, 1024.0


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    db = sqlite3.connect(argv[0])
    q = ("select k.name, (k.grid_x/k.workgroup_x) || ',' || (k.grid_y/k.workgroup_y) || ',' || (k.grid_z/k.workgroup_z), p.counter_name, sum(p.v), "
         "count(*), sum(k.duration) from (select dispatch_id, counter_name, sum(counter_value) as v from pmc_events group by 1, 2) p "
         "join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2, 3")
    data = {}
    for name, g, cname, s, n, d in db.execute(q):
        rec = data.setdefault((pretty(name), str(g)), {})
        rec[cname] = float(s)
        rec['_n'], rec['_dur_ns'] = int(n), float(d)
    print("matrix-core busy per (kernel, grid): sums over the dispatches of the traced steps; dur = traced duration with counters on")
    print("%-64s %6s %9s %10s %10s %12s" % ('kernel <workgroups>', 'calls', 'dur us', 'busy/SQ', 'busy/time', 'busy cyc/MFMA'))
    rows, fam = [], {}
    for (name, g), r in data.items():
        busy = r.get('SQ_VALU_MFMA_BUSY_CYCLES')
        if busy is None:
            continue
        sqb, insts, n, dur = r.get('SQ_BUSY_CYCLES', 0.0), r.get('SQ_INSTS_MFMA', 0.0), r['_n'], r['_dur_ns']
        rows.append((dur, name, g, n, busy, sqb, insts))
        base = name.split('<')[0].strip()
        for f, members in FAMILY.items():
            if base in members:
                a = fam.setdefault(f, [0.0, 0.0, 0.0, 0])
                a[0] += busy
                a[1] += sqb
                a[2] += dur
                a[3] += n
    for dur, name, g, n, busy, sqb, insts in sorted(rows, reverse=True):
        label = name[:60 - len(g)] + ' <' + g + '>'
        print("%-64s %6d %9.2f %10.4f %10.4f %12.1f" % (label, n, dur / n / 1e3, busy / sqb if sqb else float('nan'),
                                                       busy / (dur * CLOCK_GHZ * SIMDS) if dur else float('nan'), busy / insts if insts else float('nan')))
    print()
    print("per kernel family (bench.py's roofline families): launch-weighted")
    out = {}
    for f, (busy, sqb, dur, n) in sorted(fam.items(), key=lambda kv: -kv[1][2]):
        bt = busy / (dur * CLOCK_GHZ * SIMDS) if dur else float('nan')
        bs = busy / sqb if sqb else float('nan')
        print("  %-28s %6d launches  busy/SQ %.4f  busy/time %.4f" % (f, n, bs, bt))
        out[f] = dict(mfma_busy=round(bt, 4), mfma_busy_over_sq_busy=round(bs, 4), launches=n)
    if '--json' in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        js = dict(families=out, _csrc_sha16=bench.csrc_sha16(),
                  _source='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA (tools/mfma_busy_summary.py); mfma_busy = busy cycles / '
                          '(kernel duration x 2.4 GHz x 1024 SIMDs)')
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as fh:
            json.dump(js, fh, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
