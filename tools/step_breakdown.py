"""Per-step kernel breakdown from a rocprofv3 kernel trace (csv): the last N shape steps (delimited by k_ddim_update), kernels
aggregated by name, summed kernel time against the wall-clock step time, and the gaps between consecutive kernels.
usage: python tools/step_breakdown.py <kernel_trace.csv> [nsteps]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r['Start_Timestamp']))
upd = [i for i, r in enumerate(rows) if 'k_ddim_update' in r['Kernel_Name']]
i0, i1 = upd[-nst - 1], upd[-1]
seg = rows[i0 + 1:i1 + 1]
T = (int(rows[i1]['End_Timestamp']) - int(rows[i0]['End_Timestamp'])) / 1e3 / nst
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    agg[k][0] += 1
    agg[k][1] += d
    busy += d
print('step %.1f us wall, %.1f kernels per step, summed kernel time %.1f us per step' % (T, len(seg) / nst, busy / nst))
for k, (n, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:30]:
    print('%-66s n/step %5.1f  avg %7.1f us  per step %7.1f us  %4.1f%%' % (k, n / nst, d / n, d / nst, 100 * d / busy))
gaps, end = 0.0, int(seg[0]['End_Timestamp'])
for r in seg[1:]:
    s = int(r['Start_Timestamp'])
    if s > end:
        gaps += (s - end) / 1e3
    end = max(end, int(r['End_Timestamp']))
print('idle gaps between kernels: %.1f us per step' % (gaps / nst))
