#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd SQLite) as a per-kernel table:
   python tools/prof_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.txt"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import pretty  # noqa: E402

by_grid = '--by-grid' in sys.argv                # one line per (kernel, grid) instead of per kernel
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
db = sqlite3.connect(argv[0])
steps = float(argv[1]) if len(argv) > 1 else 1.0
key = "name || ' <' || (grid_x/workgroup_x) || ',' || (grid_y/workgroup_y) || ',' || (grid_z/workgroup_z) || '>'" if by_grid else "name"
# Residency of a launch (--by-grid).  The registers a wave is ALLOCATED are the code object's .vgpr_count (architectural + accumulation
# registers, granules of 8 of the 512 a SIMD lane has) -- NOT the trace's vgpr column, which is lower for every MFMA kernel (conv3x3_kernel: 60
# there, 112 in the code object, and four workgroups per CU is what the phase stamps of tools/conv3_micro.py count).  The counts are read from the
# library's objects (deep-prior-pp_amd/lib/obj/*.o: fat binary -> gfx950 code object -> llvm-readelf --notes); workgroups per CU = what registers
# and the 160 KB of LDS allow; rounds = workgroups of the launch / (that x 256 CUs).  Columns are blank when the tools are missing.
rows = db.execute("select %s, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                  "max(vgpr_count), max(lds_size), max(workgroup_x*workgroup_y*workgroup_z), "
                  "max((grid_x/workgroup_x)*(grid_y/workgroup_y)*(grid_z/workgroup_z)) from kernels group by 1 order by 3 desc" % key).fetchall()


def elf_resources():
    import glob
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    llvm = '/opt/rocm/lib/llvm/bin'
    recs = []
    try:
        for obj in sorted(glob.glob(os.path.join(root, 'deep-prior-pp_amd', 'lib', 'obj', '*.o'))):
            with tempfile.TemporaryDirectory() as td:
                fb, co = os.path.join(td, 'x.fatbin'), os.path.join(td, 'x.co')
                if subprocess.call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fb], stderr=subprocess.DEVNULL) != 0:
                    continue
                if subprocess.call([llvm + '/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fb,
                                    '--output=' + co, '--unbundle'], stderr=subprocess.DEVNULL) != 0:
                    continue
                cur = {}
                for line in subprocess.check_output([llvm + '/llvm-readelf', '--notes', co], stderr=subprocess.DEVNULL).decode().splitlines():
                    m = re.match(r'\s*-?\s*\.(agpr_count|name|private_segment_fixed_size|vgpr_count):\s*(\S+)', line)
                    if not m:
                        continue
                    if m.group(1) == 'agpr_count':
                        cur = {}
                    cur[m.group(1)] = m.group(2)
                    if m.group(1) == 'vgpr_count' and 'name' in cur:
                        recs.append((cur['name'], int(cur['vgpr_count']), int(cur.get('private_segment_fixed_size', 0))))
        if not recs:
            return {}
        dem = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in recs), capture_output=True, text=True).stdout.splitlines()
    except (OSError, subprocess.CalledProcessError):
        return {}
    out = {}
    for (mangled, v, scr), d in zip(recs, dem):
        for nm in (pretty(mangled), pretty(d)):
            out[nm] = (v, scr)
    return out


ELF = elf_resources() if by_grid else {}


def residency(name, lds, threads, wgs):
    if name not in ELF:
        return None
    alloc = max(8, (ELF[name][0] + 7) // 8 * 8)
    waves_simd = min(8, 512 // alloc)
    per_wg = max(1, -(-(threads or 64) // 64))
    wg_cu = max(1, min(waves_simd * 4 // per_wg, (160 * 1024 // lds) if lds else 32, 32))
    return ELF[name][0], wg_cu, (wgs or 1) / (wg_cu * 256.0)


tot = sum(r[2] for r in rows)
# a train step launches adam_kernel exactly once: when the trace holds it, the number of traced steps is COUNTED, not taken from the
# command line (round 3's header said 28 steps for a trace with 29 -- warm-up, timed region and the family re-issues of bench.py)
adam_calls = sum(r[1] for r in rows if re.search(r'\badam_kernel\b', r[0]))
if adam_calls and not by_grid and adam_calls != steps:
    print("# steps: %g given on the command line, %d adam_kernel launches in the trace -- using %d" % (steps, adam_calls, adam_calls))
if adam_calls:
    steps = float(adam_calls)
print("rocprofv3 --kernel-trace summary: %d kernels, %d dispatches, total %.1f us (%.1f us per step over %g steps)" %
      (len(rows), sum(r[1] for r in rows), tot, tot / steps, steps))
print("%-66s %7s %11s %9s %9s %9s %6s %5s %6s" % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%', 'vgpr', 'lds') +
      ("  %5s %5s %6s" % ('alloc', 'wg/cu', 'rounds') if by_grid else ''))
for r in rows:
    name, grid = r[0], ''
    if by_grid:
        name, grid = name.rsplit(' <', 1)
        grid = ' <' + grid
    name = pretty(name)
    name = name[:66 - len(grid)] + grid
    line = "%-66s %7d %11.1f %9.2f %9.2f %9.2f %6.1f %5d %6d" % (name[:66], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0)
    if by_grid:
        res = residency(pretty(r[0].rsplit(' <', 1)[0]), r[7], r[8], r[9])
        line += ("  %5d %5d %6.2f" % res) if res else ''
    print(line)

# per kernel family (all template instances / grids together): the figure bench.py's roofline.avg_launch_us corresponds to
fam = {}
for r in rows:
    name = pretty(r[0].rsplit(' <', 1)[0] if by_grid else r[0])
    key = re.sub(r'[<( ].*$', '', name)
    a = fam.setdefault(key, [0, 0.0])
    a[0] += r[1]
    a[1] += r[2]
print()
print("per kernel family:")
print("%-44s %9s %12s %9s %12s" % ('family', 'calls', 'total_us', 'avg_us', 'calls/step'))
for key, (calls, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %9d %12.1f %9.2f %12.1f" % (key[:44], calls, t, t / calls, calls / steps))
