"""Micro-benchmarks of the rows path on the GPU: graph-node floor and per-op cost of k_linear_rows.
Chains are DEPENDENT launches replayed from one hipGraph (as in a layout denoising step); `split` = K split over
workgroups with slab outputs that the next op of the chain sums while staging (round 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder, PackedLinear, View, seg, norm_segs

dev = torch.device('cuda')


def timeit(plan, step, reps=20):
    plan.sample(step, 0, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.sample(step, 0, reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps     # us per replay


def chain(M, K, N, n_ops, pro=0, split=None):
    """n_ops dependent products; op i reads op i-1's output (tiled / truncated to K columns by views of width N)."""
    b = Builder(dev)
    step = b.buf(1, dtype=torch.int32, zero=True)
    x = b.buf(M, max(K, N))
    x.normal_()
    ga, be = b.buf(K), b.buf(K)
    ga.fill_(1.0); be.zero_()
    cur = View(x, width=N)
    nseg = (K + N - 1) // N if K > N else 1
    for i in range(n_ops):
        W = torch.randn(N, K) / K ** 0.5
        pl = PackedLinear(W, torch.zeros(N), dev)
        if pro == hip.PRO_GEGLU:
            src = b.buf(M, 2 * K); src.normal_()
            cur = b.linear([seg(View(src, ld=2 * K, width=K), pro=pro)], pl, M, split=split)
            continue
        if K <= N:
            views = [cur.cols(0, K)]
        else:
            views = [cur] * (K // N)              # the same tensor as several K segments (skip-concat shapes)
        if pro in (hip.PRO_GN, hip.PRO_GN_SILU):
            segs = norm_segs(views, ga, be, 1e-5, pro == hip.PRO_GN_SILU, C=K)
        elif pro == hip.PRO_LN:
            segs = [seg(views[0], pro=pro, gamma=ga, beta=be, eps=1e-5, gs=K)]
        else:
            segs = [seg(v, pro=pro) for v in views]
        cur = b.linear(segs, pl, M, split=split)
    return b.finish(), step, cur


def cold():
    """the real step streams 335 MB of weights per pass (> the 256 MB Infinity Cache): chains long enough that weights are cold"""
    import os
    for (M, K, N, pro, name) in [(32, 512, 512, 0, 'plain512'), (32, 512, 512, 3, 'gn_silu512'), (32, 1024, 512, 3, 'gn_silu1024'),
                                 (32, 512, 512, 4, 'ln512 '), (32, 1536, 512, 0, 'plain1536')]:
        n = max(60, int(420e6 / (K * N * 4)))
        for split in (False, None) if pro != 4 else (False, 16):
            plan, step, cur = chain(M, K, N, n, pro, split)
            t = timeit(plan, step, reps=5) / n
            print('cold %-12s K=%d N=%d n_ops=%d S=%d dbg=%s: %.2f us/op' % (name, K, N, n, cur.nslab, os.environ.get('ES_ROWS_DBG', '0'), t), flush=True)
            del plan
            torch.cuda.empty_cache()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'cold':
        cold()
        sys.exit(0)
    cases = [(32, 16, 16, 0, 'tiny'), (32, 512, 512, 0, 'plain512'), (32, 512, 512, 3, 'gn_silu512'),
             (32, 512, 512, 4, 'ln512'), (32, 1024, 512, 3, 'gn_silu1024'), (32, 1536, 512, 0, 'plain1536'),
             (32, 2048, 512, 5, 'geglu2048'), (32, 512, 4096, 4, 'ln512->4096'),
             (128, 1536, 256, 0, 'gcn_l1-like')]
    modes = [('nosplit', False), ('auto', None)] + [('kbps%d' % k, k) for k in (4, 8, 16)]
    for (M, K, N, pro, name) in cases:
        line = '%-14s M=%d K=%d N=%d :' % (name, M, K, N)
        for mname, split in modes:
            if K < 64 and split not in (False, None):
                continue
            try:
                plan, step, cur = chain(M, K, N, 60, pro, split)
                t = timeit(plan, step) / 60
                line += '  %s(S=%d) %.2f us' % (mname, cur.nslab, t)
            except Exception as e:       # noqa
                line += '  %s ERR %s' % (mname, str(e)[:60])
        print(line, flush=True)
