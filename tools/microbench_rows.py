"""Micro-benchmarks of the rows path on the GPU: graph-node floor and per-op cost of k_linear_rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder, PackedLinear, View, seg

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


def chain(M, K, N, n_ops, pro=0, dep=True):
    b = Builder(dev)
    step = b.buf(1, dtype=torch.int32, zero=True)
    x = b.buf(M, K)
    x.normal_()
    ga, be = b.buf(K), b.buf(K)
    ga.fill_(1.0); be.zero_()
    cur = x
    for i in range(n_ops):
        W = torch.randn(N, K) / K ** 0.5
        pl = PackedLinear(W, torch.zeros(N), dev)
        out = b.buf(M, N)
        src = cur if (dep and N == K) else x
        if pro == hip.PRO_GEGLU:
            src = b.buf(M, 2 * K); src.normal_()
            b.linear([seg(View(src, ld=2 * K, width=K))], pl, M, View(out), prologue=pro)
        else:
            b.linear([seg(View(src))], pl, M, View(out), prologue=pro, gamma=ga, beta=be, eps=1e-5)
        cur = out
    return b.finish(), step


if __name__ == '__main__':
    for (M, K, N, pro, name) in [(32, 16, 16, 0, 'tiny'), (32, 512, 512, 0, 'plain512'), (32, 512, 512, 3, 'gn_silu512'),
                                 (32, 512, 512, 4, 'ln512'), (32, 1024, 512, 3, 'gn_silu1024'),
                                 (32, 2048, 512, 5, 'geglu2048'), (32, 512, 4096, 4, 'ln512->4096'),
                                 (32, 2048, 11264, 1, 'emb_all'), (124, 1664, 256, 0, 'gcn_l1')]:
        n = 100
        plan, step = chain(M, K, N, n, pro)
        t = timeit(plan, step)
        print('%-14s M=%d K=%d N=%d : %.2f us/op  (%.1f GB/s weights)' % (name, M, K, N, t / n, N * K * 4 / (t / n) / 1e3), flush=True)
