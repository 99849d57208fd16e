#!/usr/bin/env python
"""Per-launch-shape breakdown of ONE denoising step from a rocprofv3 --kernel-trace csv
(the step = launches between the last two k_ddim_update / k_ddpm_update).
usage: python tools/step_breakdown.py <kernel_trace.csv> [ddim|ddpm] [top]"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mark = 'k_%s_update' % (sys.argv[2] if len(sys.argv) > 2 else 'ddim')
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
seg = rows[idx[-2] + 1:idx[-1] + 1]
agg, byk, tot = collections.OrderedDict(), collections.Counter(), 0.0
for r in seg:
    n = r['Kernel_Name']
    m = re.search(r'k_[a-z0-9_]+(ILi\d+ELi\d+E|<[^>]*>)?', n)
    n = m.group(0) if m else n[:30]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    byk[n] += d
    wg = int(r['Workgroup_Size_X'])
    key = (n, int(r['Grid_Size_X']) // wg, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), wg)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-28s grid %6d %3d %2d wg %4d  n=%3d  tot %8.1f us  avg %7.1f' % (k[0], k[1], k[2], k[3], k[4], v[0], v[1], v[1] / v[0]))
print('--- by kernel')
for k, v in byk.most_common():
    print('%-28s %9.1f us  %5.1f %%' % (k, v, 100 * v / tot))
print('launches %d  busy %.1f us  span %.1f us' % (len(seg), tot, (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3))
