"""Phase stamps of the few-objects 3x3x3 launch (k_conv_ws<128, 4, 8, ..., 5>, 4 objects, 16^3, 224 -> 224, S = 2; build with
ES_BUILD_FLAGS=-DES_STAMP ES_BUILD_TAG=_stamp, run with ES_LIB_TAG=_stamp): where do the ~20 us outside the K loop go?
Stamps per wave (100 MHz wall clock): 0 kernel entry, 1 set-up done, 2 unit 0 published, 3 K loop done, 4 epilogue done."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
raw = C.CDLL(hip.LIB_PATH)
O = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (dims, cin, N, tag) in [((16, 16, 16), 224, 224, '16^3 224->224'), ((16, 8, 8), 448, 448, '16x8x8 448->448'), ((16, 4, 4), 672, 672, '16x4x4 672->672')]:
    M = O * dims[0] * dims[1] * dims[2]
    b = Builder(dev)
    x = b.buf(M, cin, dtype=torch.float16); x.normal_()
    pc = PackedConv(torch.randn(N, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.zeros(N), dev)
    o32 = b.buf(M, N)
    res = b.buf(M, N); res.normal_()
    idx = b.conv(x, pc, O, dims, res=res, out_f32=o32)
    S = hip.lib().es_conv_split_of(C.byref(b.ops[idx].u.conv))
    plan = b.finish()
    ntn = (N + 223) // 224
    nwg = ((M + 127) // 128) * ntn * S
    stamps = torch.zeros(nwg * 12 * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(stamps.data_ptr())) == 0
    plan.run()
    torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(0)) == 0
    s = stamps.cpu().numpy().reshape(nwg, 12, 8).astype(np.float64) / 100.0      # us
    if s[:, :, 4].min() == 0:
        print('== %s: S = %d, %d workgroups: stamps incomplete (another kernel took the launch?)' % (tag, S, nwg)); continue
    t0 = s[:, :, 0].min()
    cons, prod = s[:, 0, :], s[:, 4, :]
    start, end = s[:, :, 0].min(1) - t0, s[:, :, 4].max(1) - t0
    print('== %s, %d objects: S = %d, %d workgroups, launch + reduction %.1f us (events); conv kernel span %.1f us; workgroup starts %.1f..%.1f, ends %.1f..%.1f'
          % (tag, O, S, nwg, e0.elapsed_time(e1) * 1e3, end.max(), start.min(), start.max(), end.min(), end.max()))
    c, p = cons, prod
    print('  consumer wave 0: entry->first barrier %.2f us, wait for unit 0 %.2f, K loop %.2f, epilogue %.2f  | producer wave 0: set-up %.2f, '
          'set-up->unit 0 published %.2f, K loop %.2f, (epilogue barrier) %.2f'
          % ((c[:, 1] - c[:, 0]).mean(), (c[:, 2] - c[:, 1]).mean(), (c[:, 3] - c[:, 2]).mean(), (c[:, 4] - c[:, 3]).mean(),
             (p[:, 1] - p[:, 0]).mean(), (p[:, 2] - p[:, 1]).mean(), (p[:, 3] - p[:, 2]).mean(), (p[:, 4] - p[:, 3]).mean()))
    ws = s[:, :, 0]
    print('  wave entry skew inside a workgroup %.2f us; workgroup lifetime %.2f us (mean), %.2f (max)' % ((ws.max(1) - ws.min(1)).mean(), (end - start).mean(), (end - start).max()))
