"""A/B of a conv-dispatcher environment switch on one box: the launches that dominate the shape step, random operands,
20 launches back to back in one plan.  usage: python tools/microbench_ab.py VAR  -> runs itself with VAR=0 / VAR=1 alternately
(the switches are read once per process)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (name, O, dims, Cin, N, taps, kind): kind 'res' = fp32 residual + fp32 output (ResBlock conv2), 'f16' = fp16 output only (qkv), 'geglu' = FeedForward
CASES = [('3x3x3 224->224 @16^3', 32, (16, 16, 16), 224, 224, 27, 'res'), ('3x3x3 448->448 @16x8x8', 32, (16, 8, 8), 448, 448, 27, 'res'),
         ('3x3x3 672->448 @16x8x8', 32, (16, 8, 8), 672, 448, 27, 'res'), ('qkv 448->1344 @16x8x8', 32, (16, 8, 8), 448, 1344, 1, 'f16'),
         ('ff2 1792->448 @16x8x8', 32, (16, 8, 8), 1792, 448, 1, 'res'), ('ff1 GEGLU 448->3584 @16x8x8', 32, (16, 8, 8), 448, 3584, 1, 'geglu'),
         ('ff1 GEGLU 672->5376 @16x4x4', 32, (16, 4, 4), 672, 5376, 1, 'geglu'),
         ('3x3x3 672->672 @16x4x4', 32, (16, 4, 4), 672, 672, 27, 'res'), ('3x3x3 1344->672 @16x4x4', 32, (16, 4, 4), 1344, 672, 27, 'res')]


def child():
    import torch
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    dev = torch.device('cuda')
    out = []
    from echoscene_amd import hip
    for name, O, dims, cin, cout, taps, kind in CASES:
        V = dims[0] * dims[1] * dims[2]
        M = O * V
        b = Builder(dev)
        x = b.buf(M, cin, dtype=torch.float16); x.normal_()
        k = 3 if taps == 27 else 1
        wt = torch.randn(cout, cin, k, k, k) / (cin * taps) ** 0.5
        pc = PackedConv(wt if taps == 27 else wt.reshape(cout, cin), torch.zeros(cout), dev, geglu=kind == 'geglu')
        if kind == 'res':
            o = b.buf(M, cout); res = b.buf(M, cout); res.normal_()
        elif kind == 'f16':
            o = b.buf(M, cout, dtype=torch.float16)
        else:
            o = b.buf(M, cout // 2, dtype=torch.float16)
        for _ in range(20):
            if kind == 'res':
                b.conv(x, pc, O, dims, res=res, out_f32=o)
            elif kind == 'f16':
                b.conv(x, pc, O, dims, out_f16=o)
            else:
                b.conv(x, pc, O, dims, out_f16=o, epilogue=hip.EPI_GEGLU, out_ld=cout // 2)
        plan = b.finish()
        plan.run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        out.append('%s %.1f us %.0f TF' % (name, best, 2.0 * M * cout * cin * taps / best / 1e6))
    print(' | '.join(out), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child()
    else:
        var = sys.argv[1]
        for rnd in range(3):
            for v in ('0', '1'):
                e = dict(os.environ); e[var] = v
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=e, capture_output=True, text=True, timeout=600)
                print('%s=%s  %s' % (var, v, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
