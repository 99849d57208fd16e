"""Phase stamps of k_linear_ws (build with ES_BUILD_FLAGS=-DES_STAMP): per workgroup (one 256-row tile walking ncb column tiles) the
consumers' K-loop and epilogue times of the first three column tiles.  Stamps per wave (100 MHz wall clock): 0 entry,
1 + 2 cb K loop of column tile cb done, 2 + 2 cb its epilogue done."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
raw = C.CDLL(hip.LIB_PATH)
O = 32
for name, dims, K, N, geglu, res_on in (('qkv 448->1344 @16x8x8 (f16 out)', (16, 8, 8), 448, 1344, False, False),
                                        ('ff1 GEGLU 448->3584 @16x8x8', (16, 8, 8), 448, 3584, True, False),
                                        ('ff2 1792->448 @16x8x8 (f16 out + f32 residual)', (16, 8, 8), 1792, 448, False, True)):
    M = O * dims[0] * dims[1] * dims[2]
    b = Builder(dev)
    x = b.buf(M, K, dtype=torch.float16); x.normal_()
    pc = PackedConv(torch.randn(N, K) / K ** 0.5, torch.zeros(N), dev, geglu=geglu)
    if geglu:
        out = b.buf(M, N // 2, dtype=torch.float16)
        b.conv(x, pc, O, dims, out_f16=out, epilogue=hip.EPI_GEGLU, out_ld=N // 2)
    else:
        out = b.buf(M, N, dtype=torch.float16)
        res = None
        if res_on:
            res = b.buf(M, N); res.normal_()
        b.conv(x, pc, O, dims, out_f16=out, res=res)
    plan = b.finish()
    nwg_max = 4096
    stamps = torch.zeros(nwg_max * 12 * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(stamps.data_ptr())) == 0
    plan.run()
    torch.cuda.synchronize()
    assert raw.es_debug_set_stamp(C.c_void_p(0)) == 0
    s = stamps.cpu().numpy().reshape(nwg_max, 12, 8).astype(np.float64) / 100.0      # us
    used = s[:, 0, 0] > 0
    s = s[used]
    t0 = s[:, :, 0].min()
    c = s[:, 0, :]                      # consumer wave 0
    print('== %s: %.1f us per launch, %d workgroups stamped' % (name, e0.elapsed_time(e1) * 1e3, s.shape[0]))
    start = c[:, 0] - t0
    print('  workgroup entry %.1f .. %.1f us (median %.1f)' % (start.min(), start.max(), np.median(start)))
    prev = c[:, 0]
    for cb in range(3):
        k, e = c[:, 1 + 2 * cb], c[:, 2 + 2 * cb]
        ok = k > 0
        if not ok.any():
            break
        print('  column tile %d: (wait +) K loop %.2f us, epilogue %.2f us   [%d workgroups]' % (cb, (k[ok] - prev[ok]).mean(), (e[ok] - k[ok]).mean(), ok.sum()))
        prev = e
    last = s[:, :8, :].max(2).max(1) - t0
    print('  last stamped event of a workgroup: median %.1f us, max %.1f us after the first entry' % (np.median(last), last.max()))
