"""Dry run of the layout planner on the HOST (no GPU): emits one UNet1D step with a CPU Builder, lists the launches the runtime
would make (fused groups) and checks, from the pointers alone, that
  * the problems of one launch are independent (none reads or overwrites what another writes), and
  * every operand a launch reads was produced by an EARLIER launch (or is an input of the step).
Usage: python tools/plan_dryrun.py [model_channels] [nodes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def emit_layout_step_cpu(mc=128, O=8, enable_t_emb=True, concat=False, seed=3):
    from echoscene_amd import synth, config as escfg, hip
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.plan import Builder, GraphIndex, UNet1DWeights, emit_unet1d_step
    from echoscene_amd.samplers import _cap, _cpu_sd
    dev = torch.device('cpu')
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(mc, enable_t_emb=enable_t_emb, concat=concat))
    synth.seeded_fill_(net, prefix='dry.')
    w = UNet1DWeights(_cpu_sd(net), net, dev)
    _, triples = synth.synthetic_graph(O, seed=seed)
    g = GraphIndex(triples, O, dev, capacity=_cap(triples.shape[0]))
    b = Builder(dev)
    x = b.buf(O, net.in_channels)
    eps = b.buf(O, net.out_channels)
    step = b.buf(1, dtype=torch.int32, zero=True)
    oe = b.dev(torch.randn(O, 640))
    n_steps = 4
    tables = dict(emb=None, emb_all=torch.zeros(n_steps, w.emb_all.N), t_lin=torch.zeros(n_steps, 64) if enable_t_emb else None)
    temb = torch.zeros(n_steps, mc)
    emit_unet1d_step(b, w, g, x, oe, temb, step, eps, tables=tables)
    return b, dict(x=x, eps=eps, step=step)


def _linear_io(a, hip):
    """(reads, writes) of one linear problem as lists of (ptr, bytes) ranges -- generous upper bounds"""
    reads, writes = [], []
    rows_max = 1 << 20
    for s in range(a.nseg):
        sg = a.seg[s]
        ns = max(sg.nslab, 1)
        for j in range(ns):
            base = (sg.ptr or 0) + 4 * j * sg.slab_stride
            span = 4 * (max(sg.ld, 0) * ((a.M - 1) if sg.mode == hip.SEG_DIRECT else rows_max) + sg.width)
            reads.append((base, span, sg.mode != hip.SEG_DIRECT))
    ln_attn = a.nseg == 1 and a.seg[0].pro == hip.PRO_LN_ATTN        # res is an OUTPUT (the formed row), u sits gs columns behind t0
    if ln_attn:
        sg = a.seg[0]
        for j in range(max(sg.nslab, 1)):
            reads.append(((sg.ptr or 0) + 4 * (j * sg.slab_stride + sg.gs), 4 * (sg.ld * (a.M - 1) + sg.width), False))
    # (ES_PRO_LN_ATTN: res2 is an operand of the prologue, as wide as the segment -- with the launch's N = 8 C here the range ran
    #  past the end of the cross-attention vectors and, depending on where the allocator had put the next buffer, into a tensor a
    #  later launch writes: a spurious finding about once in three full test runs)
    rw = a.seg[0].width if ln_attn else a.N
    for ptr, ld, ns, ss in ((None if ln_attn else a.res, a.res_ld, a.res_nslab, a.res_slab_stride), (a.res2, a.res2_ld, a.res2_nslab, a.res2_slab_stride)):
        if ptr:
            for j in range(max(ns, 1)):
                reads.append((ptr + 4 * j * ss, 4 * (ld * (a.M - 1) + rw), False))
    Nout = a.N // 2 if a.act == hip.ACT_GEGLU else a.N
    S = 1
    if a.kb_per_slice > 0:
        import ctypes as C
        got = C.c_int(0)
        S = hip.lib().es_linear_rows_slices(C.byref(a), C.byref(got))       # (the library's rule: segment-aligned cuts since round 5)
    for j in range(S):
        writes.append(Rect(a.out + 4 * j * a.out_slab_stride, 4 * (a.out_ld * (a.M - 1) + Nout), 4 * a.out_ld, 4 * Nout))
    if ln_attn:
        writes.append(Rect(a.res, 4 * (a.res_ld * (a.M - 1) + a.seg[0].width), 4 * a.res_ld, 4 * a.seg[0].width))
    return reads, writes


class Rect(tuple):
    """a written range (ptr, bytes) that also knows its row pitch and row width: two column blocks of one matrix do not overlap"""

    def __new__(cls, ptr, nbytes, pitch=0, width=0):
        r = super().__new__(cls, (ptr, nbytes))
        r.pitch, r.width = pitch, width
        return r


def rects_overlap(a, b):
    if not (a[0] < b[0] + b[1] and b[0] < a[0] + a[1]):
        return False
    pa, pb = getattr(a, 'pitch', 0), getattr(b, 'pitch', 0)
    if pa and pa == pb and a.width <= pa and b.width <= pa:
        # only the DIFFERENCE of the two starts is meaningful (the matrix base need not be pitch-aligned: taking each address modulo
        # the pitch made this check depend on where the allocator put the buffer): b's columns start d bytes right of a's, cyclically
        d = (b[0] - a[0]) % pa
        if d >= a.width and d + b.width <= pa:
            return False
    return True


def launches(b):
    """the runtime's launch grouping (es_plan_run): [(first op index, [op, ...])]"""
    from echoscene_amd import hip
    out, i, ops = [], 0, b.ops
    while i < len(ops):
        op = ops[i]
        grp = [op]
        if op.kind == hip.OP_LINEAR and op.u.linear.fuse_next:
            q = i
            grp = []
            while len(grp) < 3 and q < len(ops) and ops[q].kind == hip.OP_LINEAR and ops[q].lane == op.lane:
                grp.append(ops[q])
                if not ops[q].u.linear.fuse_next:
                    break
                q += 1
        out.append((i, grp))
        i += len(grp)
    return out


def check(b, inputs=()):
    """returns (n_launches, problems): dependency violations found from the pointers"""
    from echoscene_amd import hip
    L = launches(b)
    problems = []
    written = []                     # (lo, hi, launch index)

    def overlaps(lo, hi, lo2, hi2):
        return lo < hi2 and lo2 < hi

    all_ios = []
    for li, (i0, grp) in enumerate(L):
        ios = []
        for op in grp:
            if op.kind == hip.OP_LINEAR:
                ios.append(_linear_io(op.u.linear, hip))
            elif op.kind == hip.OP_ROWSEL:
                a = op.u.rowsel
                ios.append(([], [(a.out, 4 * (a.out_ld * (a.rows - 1) + a.n))]))
            else:
                ios.append(([], []))
        all_ios.append(ios)
    # every intermediate is written by exactly one launch of the step: a launch that reads it must come later
    writers = [(wp, wp + wn, li) for li, ios in enumerate(all_ios) for (_, wr) in ios for (wp, wn) in wr]
    for li, ios in enumerate(all_ios):
        for (rd, _) in ios:
            for (rp, rn, gathered) in rd:
                for (w0, w1, wl) in writers:
                    hit = (w0 <= rp < w1) if gathered else overlaps(rp, rp + rn, w0, w1)
                    if hit and wl > li:
                        problems.append('launch %d reads a range that launch %d (later) writes' % (li, wl))
    for li, (i0, grp) in enumerate(L):
        ios = all_ios[li]
        for k, (rd, wr) in enumerate(ios):
            for k2, (rd2, wr2) in enumerate(ios):
                if k2 == k:
                    continue
                for (wp, wn) in wr:
                    for (rp, rn, gathered) in rd2:
                        # gathered operands: only the base pointer is known exactly -> test the row range of the written tensor
                        if overlaps(wp, wp + wn, rp, rp + (4 if gathered else rn)) or (gathered and wp <= rp < wp + wn):
                            problems.append('launch %d (op %d): problem %d reads what problem %d of the same launch writes' % (li, i0, k2, k))
                    for (wp2, wn2) in wr2:
                        if k2 > k and rects_overlap(next(w for w in wr if w[0] == wp), next(w for w in wr2 if w[0] == wp2)):
                            problems.append('launch %d (op %d): problems %d and %d write overlapping ranges' % (li, i0, k, k2))
        for (rd, wr) in ios:
            for (wp, wn) in wr:
                written.append((wp, wp + wn, li))
    return len(L), problems


def describe(b):
    from echoscene_amd import hip
    lines = []
    for li, (i0, grp) in enumerate(launches(b)):
        parts = []
        for op in grp:
            if op.kind == hip.OP_LINEAR:
                a = op.u.linear
                pro = ','.join(str(a.seg[s].pro) for s in range(a.nseg))
                modes = ','.join(str(a.seg[s].mode) for s in range(a.nseg))
                parts.append('lin M%d K%d N%d kbps%d pro[%s] mode[%s] act%d%s%s' % (a.M, a.K, a.N, a.kb_per_slice, pro, modes, a.act,
                                                                                 ' res' if a.res else '', ' res2' if a.res2 else ''))
            else:
                parts.append('kind%d' % op.kind)
        lines.append('%3d: %s' % (li, ' || '.join(parts)))
    return lines


if __name__ == '__main__':
    mc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    O = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    b, io = emit_layout_step_cpu(mc, O)
    print('\n'.join(describe(b)))
    n, problems = check(b)
    print('%d ops, %d launches, %d dependency problems' % (len(b.ops), n, len(problems)))
    for p in problems:
        print('  ' + p)
