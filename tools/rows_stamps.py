"""Timeline of ONE layout denoising step from inside the rows kernels (instrumented build: ES_BUILD_TAG=_stamp ES_BUILD_FLAGS=-DES_STAMP
python -m echoscene_amd.build --force; this tool loads it with ES_LIB_TAG=_stamp).  Every workgroup's wave 0 stamps the 100 MHz wall clock at:
0 kernel entry, 1 kernarg block read, 2 operand staged (its loads returned, prologue applied, LDS written), 3 barrier passed,
4 MFMA chain done + partials in LDS, 5 second barrier passed, 6 epilogue stored.  Per launch: span = last exit - first entry; gap = first
entry of the NEXT rows launch - last exit (the launch boundary plus any non-rows kernel in between).
usage: ES_LIB_TAG=_stamp python tools/rows_stamps.py [O] [out.txt]"""
import os, sys, ctypes as C
os.environ.setdefault('ES_LIB_TAG', '_stamp')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from echoscene_amd import hip

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
raw = C.CDLL(hip.LIB_PATH)
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)
NL = 256
stamps = torch.zeros(NL * 1024 * 8, dtype=torch.int64, device=dev)
log = '/tmp/rows_launches.txt'
raw.es_debug_rows_stamp.argtypes = [C.c_void_p, C.c_char_p]
assert raw.es_debug_rows_stamp(C.c_void_p(stamps.data_ptr()), log.encode()) == 0
den.sample(obj_embed, triples, noise=None, n_steps=40)          # the plan is captured with the stamp pointer in its kernargs
torch.cuda.synchronize()
assert raw.es_debug_rows_stamp(None, None) == 0
launches = [l.split(None, 4) for l in open(log).read().splitlines()]
s = stamps.cpu().numpy().reshape(NL, 1024, 8)
rows = []
for l in launches:
    lid, gx, gy, n = int(l[0]), int(l[1]), int(l[2]), int(l[3])
    nwg = min(gx * gy, 1024)
    st = s[lid, :nwg].astype(np.float64)
    st = st[st[:, 0] > 0]                      # workgroups that exited at entry (folded remainder) wrote nothing
    if not len(st):
        continue
    rows.append((lid, gx, gy, n, l[4], st))
t_first = rows[0][5][:, 0].min()
print('# layout step, O = %d: %d rows launches; times in us (100 MHz clock: 10 ns resolution)' % (O, len(rows)))
print('# id  grid   wgs | first entry (since step start) | entry spread | kernarg | stage | barrier | mfma | barrier2 | store | span | gap to next | desc')
tot = dict(kernarg=0.0, stage=0.0, bar=0.0, mfma=0.0, bar2=0.0, store=0.0, span=0.0, gap=0.0, spread=0.0)
for i, (lid, gx, gy, n, desc, st) in enumerate(rows):
    t = st / 100.0
    e0, end = t[:, 0].min(), t[:, 6].max()
    ph = np.median(t[:, 1:7] - t[:, 0:6], axis=0)
    # critical workgroup = the one that exits last
    crit = t[np.argmax(t[:, 6])]
    gap = (rows[i + 1][5][:, 0].min() / 100.0 - end) if i + 1 < len(rows) else float('nan')
    spread = t[:, 0].max() - e0
    print('%3d %4dx%d %4d | %8.2f | %5.2f | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | span %5.2f | gap %5.2f | crit wg: %s | %s'
          % (lid, gx, gy, len(st), e0 - t_first / 100.0, spread, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], end - e0, gap,
             ' '.join('%.2f' % v for v in (crit[1:7] - crit[0:6])), desc))
    for k, v in zip(('kernarg', 'stage', 'bar', 'mfma', 'bar2', 'store'), ph):
        tot[k] += v
    tot['span'] += end - e0
    tot['spread'] += spread
    if gap == gap:
        tot['gap'] += gap
print('# sums over %d launches (median workgroup per launch): ' % len(rows) + '  '.join('%s %.1f' % kv for kv in tot.items()))
print('# step (first entry of launch 0 -> last exit of the last rows launch): %.1f us' % (rows[-1][5][:, 6].max() / 100.0 - t_first / 100.0))
# XCD of workgroup 0 of every launch vs "the dispatcher continues round-robin where the previous launch stopped"
xs, pred, tot = [], [], 0
for (lid, gx, gy, n, desc, st) in rows:
    xs.append(int(s[lid, 0, 7] & 0xf))
    pred.append(tot % 8)
    tot += gx * gy
print('# XCC_ID of workgroup 0 per launch:        ', xs)
print('# (sum of earlier rows grids) % 8 per launch:', pred)
