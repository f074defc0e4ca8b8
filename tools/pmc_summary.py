#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in SEPARATE runs, see
/opt/skills/guides/MI355X_MICROARCH.md "HBM" and "rocprofv3 PMC slots"):
   python tools/pmc_summary.py <fetch_results.db> <write_results.db> [steps] > profiles/rNN_hbm_traffic.txt
Both counters are in KiB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads; instead of
assuming the factor, both counters are CALIBRATED on adam_kernel, whose traffic is known exactly (reads w, g, m, v and
writes w, m, v: 16 and 12 bytes per parameter, each element touched once, far larger than any cache)."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import pretty  # noqa: E402

N_PARAMS_PAD = None


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), avg(counter_value), sum(counter_value), avg(duration) from pmc_events "
                      "where counter_name = ? group by name", (counter,)).fetchall()
    return {r[0]: r[1:] for r in rows}


def short(name):
    return pretty(name)[:70]


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    steps = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith('--') else 1.0
    n_adam = 18714452.0      # parameters of the bs128 NYU ResNet (type 0, 30-D output); --params N for another net (256x256: 69046100)
    if '--params' in sys.argv:
        n_adam = float(sys.argv[sys.argv.index('--params') + 1])
    adam = [k for k in fetch if 'adam_kernel' in k]
    fcal = wcal = 1.0
    if '--cal-from' in sys.argv:
        # a pass without the ADAM update (the deterministic forward): the calibration factors of the train-step pass of the same run
        import json as _json
        with open(sys.argv[sys.argv.index('--cal-from') + 1]) as fh:
            cj = _json.load(fh)
        fcal, wcal = float(cj['_fetch_cal']), float(cj['_write_cal'])
        adam = []
    if adam:
        fcal = 16.0 * n_adam / (fetch[adam[0]][1] * 1024.0)
        wcal = 12.0 * n_adam / (write[adam[0]][1] * 1024.0)
    print("HBM traffic per kernel from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %g steps" % steps)
    print("calibration on adam_kernel (%.0f parameters: 16 B read, 12 B written each): true/FETCH_SIZE = %.3f, true/WRITE_SIZE = %.3f"
          % (n_adam, fcal, wcal))
    print("%-70s %8s %12s %12s %12s %14s" % ('kernel', 'calls', 'fetch MB', 'write MB', 'HBM MB', 'MB per step'))
    tot = 0.0
    out = []
    for k in fetch:
        calls = fetch[k][0]
        f = fetch[k][1] * 1024.0 * fcal / 1e6
        w = (write[k][1] if k in write else 0.0) * 1024.0 * wcal / 1e6
        out.append((calls * (f + w), short(k), calls, f, w))
    for t, name, calls, f, w in sorted(out, reverse=True):
        tot += t
        print("%-70s %8d %12.3f %12.3f %12.3f %14.1f" % (name, calls, f, w, f + w, t / steps))
    print("total HBM traffic per step: %.1f MB" % (tot / steps))
    print("note: the FETCH_SIZE calibration holds for 16 B / lane streaming reads (adam_kernel); kernels that read 4 B / lane")
    print("      (reduce_multi, reduce_partials, bn finalizes) may be over-stated by up to 2x.")
    fam = {}
    for t, name, calls, f, w in out:
        key = name.split('<')[0]
        a = fam.setdefault(key, [0, 0.0])
        a[0] += calls
        a[1] += t
    print("per kernel family (all template instances):")
    for key, (calls, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s %7.1f launches/step %10.3f MB/launch %10.1f MB/step" % (key, calls / steps, t / calls, t / steps))
    if '--json' in sys.argv:
        import json
        js = {key: dict(launches_per_step=calls / steps, bytes_per_launch=t / calls * 1e6) for key, (calls, t) in fam.items()}
        js['_total_bytes_per_step'] = tot / steps * 1e6
        js['_fetch_cal'], js['_write_cal'] = fcal, wcal
        js['_source'] = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), calibrated on adam_kernel'
        # fingerprint of the kernel sources of the tree this ran in (= the tree that was profiled): bench.py only quotes the file for
        # a build of the same sources
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        js['_csrc_sha16'] = bench.csrc_sha16()
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as fh:
            json.dump(js, fh, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
