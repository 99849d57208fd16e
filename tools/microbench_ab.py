"""A/B of a conv-dispatcher environment switch on one box: the launches that dominate the shape step, random operands,
20 launches back to back in one plan.  usage: python tools/microbench_ab.py VAR  -> runs itself with VAR=0 / VAR=1 alternately
(the switches are read once per process)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [('3x3x3 224->224 @16^3', 32, (16, 16, 16), 224, 224, 27), ('3x3x3 448->448 @16x8x8', 32, (16, 8, 8), 448, 448, 27),
         ('3x3x3 672->448 @16x8x8', 32, (16, 8, 8), 672, 448, 27), ('1x1 448->1344 @16x8x8', 32, (16, 8, 8), 448, 1344, 1),
         ('1x1 1792->448 @16x8x8', 32, (16, 8, 8), 1792, 448, 1)]


def child():
    import torch
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    dev = torch.device('cuda')
    out = []
    for name, O, dims, cin, cout, taps in CASES:
        V = dims[0] * dims[1] * dims[2]
        M = O * V
        b = Builder(dev)
        x = b.buf(M, cin, dtype=torch.float16); x.normal_()
        k = 3 if taps == 27 else 1
        pc = PackedConv(torch.randn(cout, cin, k, k, k) / (cin * taps) ** 0.5, torch.zeros(cout), dev)
        o = b.buf(M, cout); res = b.buf(M, cout); res.normal_()
        for _ in range(20):
            b.conv(x, pc, O, dims, res=res, out_f32=o)
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
