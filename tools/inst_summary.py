#!/usr/bin/env python3
"""Dynamic instruction counts per wave, by kernel and grid, from rocprofv3 SQ counter passes (counters only, --kernel-trace):
   python tools/inst_summary.py <results.db> [<results2.db> ...] > profiles/rNN_instruction_mix.txt
Every counter is summed over the dispatch; divided by SQ_WAVES it is the number of instructions ONE wave issues on its way
through the kernel.  A SIMD issues roughly one instruction of a wave per 4 cycles (tools/probes/icache_probe.hip: 4.3 cycles
per dependent scalar instruction, and two waves on one SIMD take twice as long), so instructions per wave x waves per SIMD x
~4.3 cycles is a floor on the kernel's duration that no memory system can hide."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import pretty  # noqa: E402


def short(name):
    return pretty(name)


def main():
    data = {}          # (kernel, grid) -> {counter: (sum, n), '_dur': (sum, n)}
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        # pmc_events holds one row per (dispatch, counter, hardware instance): total per dispatch first, then average per (kernel, grid)
        q = ("select k.name, (k.grid_x/k.workgroup_x) || ',' || (k.grid_y/k.workgroup_y) || ',' || (k.grid_z/k.workgroup_z), p.counter_name, "
             "sum(p.v), count(*), sum(k.duration) from (select dispatch_id, counter_name, sum(counter_value) as v from pmc_events group by 1, 2) p "
             "join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2, 3")
        for name, g, cname, s, n, d in db.execute(q):
            rec = data.setdefault((short(name), str(g)), {})
            rec[cname] = (rec.get(cname, (0.0, 0))[0] + s, rec.get(cname, (0.0, 0))[1] + n)
            rec['_dur'] = (d, n)
            rec['_wgs'] = eval(g.replace(',', '*'))
    counters = sorted(set(c for rec in data.values() for c in rec if not c.startswith('_') and c != 'SQ_WAVES'))
    labels = [c.replace('SQ_INSTS_', '').replace('SQ_', '') for c in counters]
    print("per wave: counter / SQ_WAVES, averaged over the dispatches of that (kernel, grid); dur = traced kernel duration with counters on")
    print("%-58s %6s %7s %7s %7s " % ('kernel <workgroups>', 'calls', 'waves', 'dur us', 'w/SIMD') + ' '.join('%9s' % l[:9] for l in labels))
    rows = []
    for (name, g), rec in data.items():
        if 'SQ_WAVES' not in rec:
            continue
        wsum, n = rec['SQ_WAVES']
        waves = wsum / n
        dur = rec['_dur'][0] / rec['_dur'][1] / 1e3 if rec['_dur'][1] else 0.0
        vals = [(rec[c][0] / rec[c][1]) / waves if c in rec else float('nan') for c in counters]
        rows.append((dur * n, name, g, n, waves, dur, vals))
    for _, name, g, n, waves, dur, vals in sorted(rows, reverse=True):
        label = (name[:56 - len(g) - 3] + ' <' + g + '>')
        print("%-58s %6d %7d %7.2f %7.2f " % (label, n, waves, dur, waves / 1024.0) + ' '.join('%9.1f' % v for v in vals))


if __name__ == '__main__':
    main()
