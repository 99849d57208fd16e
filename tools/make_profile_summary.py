#!/usr/bin/env python
"""Builds profiles/rNN_* from a gpurun_out/ collection:
   gpurun_out/bench_final.json                     (python bench.py --steps 100 --warmup 5)
   gpurun_out/prof_final/*/*kernel_stats.csv       (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline)
   gpurun_out/pmc_final/<set>/*/*counter_collection.csv   (one rocprofv3 --kernel-trace --pmc <set> pass per counter set)
usage: python tools/make_profile_summary.py r02 [gpurun_out sub-directory]
Also writes profiles/<tag>_pmc_traffic.json: HBM bytes per launch by kernel family (FETCH_SIZE x 2 -- gfx950 reports half of a
wide streaming read, MI355X_MICROARCH.md section HBM -- + WRITE_SIZE, both KB in rocprofv3), which bench.py reports as
roofline.traffic."""
import csv, glob, json, os, re, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
go = os.path.join(root, 'gpurun_out', *(sys.argv[2:3]))
line = [l for l in open(os.path.join(go, 'bench_final.json')) if l.startswith('{')][-1]
open(os.path.join(out, tag + '_bench_line.json'), 'w').write(line)
bench = json.loads(line)
ks = (glob.glob(os.path.join(go, 'prof_final', '*', '*kernel_stats.csv')) + glob.glob(os.path.join(go, 'prof_final', '*kernel_stats.csv')))[0]
shutil.copy(ks, os.path.join(out, tag + '_full_step_kernel_stats.csv'))


def short(n):
    m = re.search(r'k_[a-z0-9_]+(<[^>]*>)?', n)
    return m.group(0) if m else n[:40]


rows = list(csv.DictReader(open(ks)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(go, 'pmc_final', '*', '*', '*counter_collection.csv')) + glob.glob(os.path.join(go, 'pmc_final', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        pmc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
md = ['# Round %s profile summary (MI355X, rocprofv3)\n' % tag[1:],
      '* `%s_bench_line.json` -- `python bench.py --steps 100 --warmup 5`: **%.2f full steps/s** (one replayed hipGraph per full step, the layout step '
      'as a parallel branch of the shape step; each loop alone: shape %.2f ms, layout %.3f ms per step), CPU oracle baseline %.5f steps/s on %d host threads; `roofline.achieved` %.1f TFLOP/s for `k_conv_ws` '
      '(avg launch %.1f us, %d launches per step).' % (
          tag, bench['value'], bench['config']['shape']['ms_per_step'], bench['config']['layout']['ms_per_step'],
          bench['cpu_baseline']['value'], bench['cpu_baseline']['cores'], bench['roofline']['achieved'],
          bench['roofline'].get('avg_launch_us') or 0, bench['roofline'].get('launches_per_step') or 0),
      '* `%s_full_step_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` '
      '(23 executions of the fused step, plus the stand-alone loops bench.py times for its per-loop record).' % tag,
      '* PMC: separate `rocprofv3 --kernel-trace --pmc <set>` passes of `python bench.py --steps 6 --warmup 2 --no-cpu-baseline` '
      '(FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY).\n',
      '| kernel | calls | avg us | % of GPU time | FETCH_SIZE KB/launch (raw; gfx950 reports 1/2 of wide streaming reads) | WRITE_SIZE KB/launch | MFMA busy / (GUI_ACTIVE/8 x 1024 SIMDs) | LDS active / CU-cycles | wave-cycles waiting |',
      '|---|---|---|---|---|---|---|---|---|']
for r in rows[:18]:
    n = short(r['Name'])
    d = pmc.get(n, {})
    avg = lambda c: (sum(d[c]) / len(d[c])) if c in d and d[c] else None
    gui = avg('GRBM_GUI_ACTIVE')
    mf = avg('SQ_VALU_MFMA_BUSY_CYCLES')
    lds = avg('SQ_LDS_IDX_ACTIVE')
    wc, wa = avg('SQ_WAVE_CYCLES'), avg('SQ_WAIT_ANY')
    f = lambda x, fmt='%.0f': '-' if x is None else fmt % x
    md.append('| `%s` | %s | %.1f | %.1f | %s | %s | %s | %s | %s |' % (
        n, r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot, f(avg('FETCH_SIZE')), f(avg('WRITE_SIZE')),
        '-' if not (gui and mf is not None) else '%.3f' % (mf / (gui / 8 * 1024)),
        '-' if not (gui and lds is not None) else '%.3f' % (lds / (gui / 8 * 256)),
        '-' if not (wc and wa is not None) else '%.2f' % (wa / wc)))
ws = [r for r in rows if 'k_conv_ws' in r['Name'] or 'k_linear_ws' in r['Name']]
if ws:
    wt, wc = sum(float(r['TotalDurationNs']) for r in ws), sum(int(r['Calls']) for r in ws)
    md.append('')
    md.append('Cross-check of `roofline.avg_launch_us`: rocprofv3 average over all `k_conv_ws` / `k_linear_ws` variants = %.1f us (%d calls); `bench.py` '
              '(HIP events, launches replayed back to back, split-K launches including their reduction kernel) = %.1f us.'
              % (wt / wc / 1e3, wc, bench['roofline'].get('avg_launch_us') or 0))
traffic = {}
fam = collections.defaultdict(lambda: collections.defaultdict(list))
for n, d in pmc.items():
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        fam[n.split('<')[0]][c] += d.get(c, [])
for n, d in fam.items():
    if d['FETCH_SIZE'] and d['WRITE_SIZE']:
        fe, wr = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        traffic[n] = {'hbm_bytes_per_launch': int((2 * fe + wr) * 1024), 'fetch_kb_raw': round(fe, 1), 'write_kb_raw': round(wr, 1),
                      'launches_sampled': len(d['FETCH_SIZE'])}
json.dump(traffic, open(os.path.join(out, tag + '_pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
extra = os.path.join(out, tag + '_notes.md')
if os.path.exists(extra):
    md += ['', open(extra).read()]
open(os.path.join(out, tag + '_pmc_summary.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md[:12]))
