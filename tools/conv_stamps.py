"""Phase stamps of k_conv_ws (build with ES_BUILD_FLAGS=-DES_STAMP): where does the per-tile time outside the K loop go?
Stamps per wave (100 MHz wall clock): 0 kernel entry, 1 set-up done (producer: masks / offsets; consumer: at the first barrier),
2 unit 0 published, 3 K loop done, 4 epilogue done."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
L = hip.lib()
raw = C.CDLL(hip.LIB_PATH)
O, dims, N = 32, (16, 16, 16), 224
M = O * 4096
for (cin, taps, res_on, tag) in [(224, 27, True, '3x3x3 224->224 f32+res'), (224, 27, False, '3x3x3 224->224 f32'),
                                 (32, 27, False, '3x3x3 32->224 f32'), (448, 1, True, '1x1 448->224 f32+res')]:
    b = Builder(dev)
    x = b.buf(M, cin, dtype=torch.float16); x.normal_()
    w = torch.randn(N, cin, 3, 3, 3) if taps == 27 else torch.randn(N, cin)
    pc = PackedConv(w / (cin * taps) ** 0.5, torch.zeros(N), dev)
    o32 = b.buf(M, N)
    res = b.buf(M, N) if res_on else None
    if res is not None:
        res.normal_()
    b.conv(x, pc, O, dims, res=res, out_f32=o32, out_f16=None)
    plan = b.finish()
    nwg = (M // 256)
    stamps = torch.zeros(nwg * 12 * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        plan.run()
    torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(stamps.data_ptr())) == 0
    plan.run()
    torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(0)) == 0
    s = stamps.cpu().numpy().reshape(nwg, 12, 8).astype(np.float64) / 100.0      # us
    t0 = s[:, :, 0].min()
    cons, prod = s[:, 0, :], s[:, 8, :]
    print('== %s: %d workgroups' % (tag, nwg))
    start = s[:, :, 0].min(1) - t0
    end = s[:, :, 4].max(1) - t0
    order = np.argsort(start)
    r1, r2 = order[:256], order[256:]
    print('  launch span %.1f us; round 1 starts %.1f..%.1f us, ends %.1f..%.1f; round 2 starts %.1f..%.1f, ends %.1f..%.1f'
          % (end.max(), start[r1].min(), start[r1].max(), end[r1].min(), end[r1].max(),
             start[r2].min() if len(r2) else 0, start[r2].max() if len(r2) else 0, end[r2].min() if len(r2) else 0, end[r2].max() if len(r2) else 0))
    for name, rr in (('round 1', r1), ('round 2', r2)):
        if not len(rr):
            continue
        c, p = cons[rr], prod[rr]
        print('  %s consumer wave 0: entry->first barrier %.2f us, wait for unit 0 %.2f, K loop %.2f, epilogue %.2f  | producer wave: set-up %.2f, '
              'set-up->unit 0 published %.2f, K loop %.2f, (epilogue barrier) %.2f'
              % (name, (c[:, 1] - c[:, 0]).mean(), (c[:, 2] - c[:, 1]).mean(), (c[:, 3] - c[:, 2]).mean(), (c[:, 4] - c[:, 3]).mean(),
                 (p[:, 1] - p[:, 0]).mean(), (p[:, 2] - p[:, 1]).mean(), (p[:, 3] - p[:, 2]).mean(), (p[:, 4] - p[:, 3]).mean()))
        wave_start = s[rr][:, :, 0]
        print('  %s: wave entry skew inside a workgroup %.2f us (max - min)' % (name, (wave_start.max(1) - wave_start.min(1)).mean()))
    if len(r2):
        # gap between a round-1 workgroup's end and the next workgroup's start on ... (CU unknown): compare distributions
        print('  median round-1 end %.1f us vs median round-2 start %.1f us' % (np.median(end[r1]), np.median(start[r2])))
