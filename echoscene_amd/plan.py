"""Plan compiler: turns a reference-keyed ``state_dict`` + a scene graph into the op list that the
native runtime (csrc/es_runtime.hip) enqueues / graph-replays.

Python runs only at build time (once per model and once per graph size); the per-step hot loop is
``es_sampler_run`` in C++.  No arithmetic of the hot path is done by PyTorch here: torch is used
for device allocations, host-side weight preparation (BatchNorm folding, centre-tap extraction,
concatenation of projection matrices) and host->device copies.
"""
import ctypes as C
from ctypes import byref as _byref
import os

import numpy as np
import torch

from . import hip
from .hip import Seg, LinearArgs, UpdateArgs, CopyArgs, RowSelArgs, Op


# K-slice choices of the rows planner.  They change where a K sum is cut -- i.e. the fp32 bits of the results -- so they are
# CONSTANTS of the build, not environment switches (two processes / ranks with different environments would break the
# bit-for-bit claims: collated batch == single scene, plan reuse, model-file replay).  A/B figures from round 3, one box, layout steps/s:
ROWS_GCN_SLICES = 2      # K slices of the triple-row GCN products: 8 -> 1048, 4 -> 1066, 2 -> ~1090
ROWS_CAV_SLICES = 1      # cross-attention-vector product: 1 / 2 / 3 / 4 slices -> 1124 / 1125 / 1118 / 1108
ROWS_LN_SPLIT = 2        # slices of a product whose consumer is a LayerNorm (whole-row statistics): 1 -> 1067, 2 -> 1093
ROWS_VO1_SLICES = 2      # attention-with-one-token product: 2 / 1 -> 1124 / 1122
ROWS_SKIP_EARLY = False  # a ResBlock's skip projection on its first conv's launch: measured -1.1 %
# Round 5 (second half): the one-token self-attention of a transformer block, t2 = vo1(LayerNorm1(t0)) + t0 + cav, is linear in t0 up to ONE
# row statistic: LayerNorm1's mean subtraction is the fixed projection P = I - 11^T / C, so vo1(LN1(t0)) = rstd(t0) (W1 P) t0 + b, and
# u = (W1 P) t0 = GN(x) (W1 P Wp)^T + W1 P bp comes out of the input projection's OWN launch (weights [Wp ; W1 P Wp], folded in fp64; the
# bias b joins the cross-attention vector's bias).  The feed-forward launch then forms t2 = rstd u + t0 + cav in its prologue
# (ES_PRO_LN_ATTN) and the self-attention product is no launch of its own: 11 dependent launches less per layout step.
ROWS_FOLD_ATTN1 = True
# The head of the UNet1D trunk (conv_in ... the first transformer's proj_in: 9 dependent products that do not need the GCN output)
# rides on the launches of the GCN chain instead of following it: 9 launches off the critical path of a layout step.  Not a numerical
# choice (every product keeps its own K slices: same bits), a planner constant (not an environment switch): 0 = off, 1 = conv_in on the box embedding's
# launch + one product on net2's output launch of every GCN layer, 2 = also one on net1's second Linear (320 + 256 workgroups).
# Same-box A/B (profiles/r04_layout_ride_ab.txt, tools/ab_layout_ride.py sets this attribute): 0 / 1 / 2 -> 810-823 / 801 / 795 us per step.
ROWS_RIDE = 2


# Round 5: the K cuts of a rows product (planner constants: they decide where an fp32 sum is cut, i.e. the bits; A/B tools set the attributes).
# A consumer workgroup loads its 16-row A slice once per slab of the producer, whatever the slice count: per-workgroup A bytes are
# 64 x K and the L2 -> L1 traffic of a launch is that times its workgroups (round-5 stamps: a launch with 384 workgroups spends 1.5 us
# ISSUING loads) -- so at most one workgroup per CU and at most 4 slabs.
ROWS_MAX_SLABS = 4
ROWS_MAX_WGS = 256


def seg_aligned_kbps(seg_widths, K, N, kalign_cols=16):
    """k-blocks (16 columns) per K slice of a rows product whose output may be a slab tensor -- round 5 rule: slices never straddle two
    segments (the low-latency kernel k_rows_x reads ONE segment per workgroup): every segment is cut into ceil(width / slice) slices.
    The smallest slice (a multiple of the largest GroupNorm group, at least 128 columns) that keeps the slab count <= ROWS_MAX_SLABS and
    the workgroups of two 16-row tiles <= ROWS_MAX_WGS.  A function of the segment widths and N only -- never of M.  0 = one slice."""
    if K % 16 or any(w % 16 for w in seg_widths):
        return 0
    wk = [w // 16 for w in seg_widths]
    kal = max(1, kalign_cols // 16)
    nct = (N + 15) // 16
    smax = min(ROWS_MAX_SLABS, ROWS_MAX_WGS // (2 * nct))
    if smax < 2:
        return 0
    for kbps in range(8, max(wk) + 1):
        if kbps % kal:
            continue
        S = sum((w + kbps - 1) // kbps for w in wk)
        if S <= smax:
            # (slices of more than 96 k-blocks are beyond k_rows_x -- 12 blocks per wave --, and k_linear_rows cuts K uniformly only)
            return kbps if S >= 2 and kbps <= 96 else 0
    return 0


class View:
    """Pointer + leading dimension into a device matrix (column slices without copies).
    ``nslab`` > 1: a SLAB tensor -- the split-K output of a rows launch, whose value is the fixed-order sum of ``nslab``
    matrices ``slab_stride`` floats apart; whoever reads it (A segment, residual, DDPM update) sums the slabs."""

    def __init__(self, t, col=0, ld=None, width=None, row=0, nslab=1, slab_stride=0, step=None, step_stride=0):
        # step / step_stride: the view is row (*step) of a per-schedule table, step_stride floats apart (residual operands only)
        self.step, self.step_stride = step, step_stride
        self.t = t
        self.col = col
        self.row = row
        self.ld = (t.shape[-1] if t.dim() > 1 else 0) if ld is None else ld
        self.width = (t.shape[-1] - col) if width is None else width
        self.nslab = nslab
        self.slab_stride = slab_stride

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * (self.col + self.row * self.ld)

    def cols(self, col, width):
        """column slice of this view (slab tensors stay slab tensors)"""
        return View(self.t, col=self.col + col, ld=self.ld, width=width, row=self.row, nslab=self.nslab,
                    slab_stride=self.slab_stride)

    def value(self):
        """the matrix this view denotes, as a fresh tensor (tests / debugging only: sums the slabs in torch)"""
        rows = self.t.shape[-2] if self.t.dim() > 1 else 1
        base = self.t.storage_offset() + self.col + self.row * self.ld
        out = None
        for j in range(max(self.nslab, 1)):
            m = self.t.as_strided((rows, self.width), (max(self.ld, 0), 1), base + j * self.slab_stride)
            out = m.clone() if out is None else out + m
        return out


def seg(view, mode=hip.SEG_DIRECT, idx=None, ent_row=None, ent_off=None, step=None, step_stride=0, width=None,
        pro=hip.PRO_NONE, gamma=None, beta=None, eps=0.0, gs=0, pre_act=hip.ACT_NONE, goff=0, ent_wt=None):
    """One K segment of a rows launch.  pro/gamma/beta/eps/gs: the segment's own prologue (GroupNorm in groups of ``gs``
    channels, LayerNorm, SiLU); ``goff``: offset of the segment's first channel inside gamma/beta; ``pre_act``: activation
    applied to the (slab-summed) source first."""
    s = Seg()
    s.ptr = view.ptr
    s.idx = idx.data_ptr() if idx is not None else None
    s.ent_row = ent_row.data_ptr() if ent_row is not None else None
    s.ent_off = ent_off.data_ptr() if ent_off is not None else None
    s.step = step.data_ptr() if step is not None else None
    s.step_stride = step_stride
    s.ld = view.ld
    s.width = view.width if width is None else width
    s.mode = mode
    s.nslab, s.slab_stride = view.nslab, view.slab_stride
    s.pre_act = pre_act
    s.pro = pro
    s.gamma = gamma.data_ptr() + 4 * goff if gamma is not None else None
    s.beta = beta.data_ptr() + 4 * goff if beta is not None else None
    s.eps = eps
    s.gs = gs
    s.ent_wt = ent_wt.ptr if ent_wt is not None else None            # SEG_CSRWAVG: View of the [T, 2] weights
    s._keep = (view.t, idx, ent_row, ent_off, step, gamma, beta, ent_wt.t if ent_wt is not None else None)
    return s


def rows_gn_in_registers(C, groups=32):
    """True when the rows kernels' GroupNorm prologue takes ``groups`` groups over C channels: it reduces a group inside one wave with
    power-of-two lane strides (es_rows_x.h), i.e. group sizes 4, 8, 16, 32 -- the shipped layout denoisers' 512 / 1024 channels."""
    gs = C // groups
    return C % groups == 0 and 4 <= gs <= 32 and gs & (gs - 1) == 0


def norm_segs(views, gamma, beta, eps, silu, C=None, groups=32, b=None, M=None):
    """segments of GroupNorm32(+SiLU) over the channel concatenation of ``views`` (th.cat([h, hs.pop()]) followed by
    normalization(ch), openai_model_3d.py / denoise_net.py ResBlock.in_layers): groups never straddle a source.

    Any other group size (GroupNorm32(32, channels) takes every channels % 32 == 0, ldm_diffusion_util.py:222-239; e.g.
    model_channels = 384: groups of 12, or the 36-channel groups of a 768 + 384 concatenation): the norm is its own launch in front
    of the product -- the channels-last GroupNorm of the volume path over M rows of ONE voxel each, fp32 in / fp32 out
    (es_groupnorm_vol: sums in fp32, statistics in double) -- and the product reads the normalised rows as a plain segment.  The
    sources must then be whole matrices (``Builder.allow_split`` off: emit_unet1d_step decides that for the whole plan)."""
    C = sum(v.width for v in views) if C is None else C
    gs = C // groups
    if C % groups:
        raise ValueError('GroupNorm32 over %d channels: not a multiple of %d groups' % (C, groups))
    if not rows_gn_in_registers(C, groups) or any(v.width % gs for v in views):
        if b is None or M is None:
            raise ValueError('GroupNorm32 over %d channels: group size %d needs the separate-launch route (pass the builder)' % (C, gs))
        assert 1 <= len(views) <= 2 and C <= 2048 and groups <= 64, (len(views), C, groups)
        for v in views:
            assert v.nslab <= 1 and v.col == 0 and v.row == 0 and v.ld == v.width and v.width % 8 == 0, \
                'the separate GroupNorm launch reads whole fp32 matrices (allow_split off)'
        from .hip import GNArgs
        if gamma is None:                            # the affine is folded into the product's weights (fold_affine): unit scale, no shift
            gamma, beta = b.dev(torch.ones(C)), b.dev(torch.zeros(C))
        y = b.buf(M, C, scratch=True)
        a = GNArgs()
        a.x1, a.C1 = views[0].ptr, views[0].width
        a.x2, a.C2 = (views[1].ptr, views[1].width) if len(views) > 1 else (None, 0)
        a.O, a.V, a.groups, a.eps = M, 1, groups, eps
        a.gamma, a.beta, a.silu = gamma.data_ptr(), beta.data_ptr(), 1 if silu else 0
        st = getattr(b, '_gn_stats', None)           # (Builder.groupnorm's scratch: one for all GroupNorms of a plan, same stream)
        need = M * groups * 2 + M * groups * 2
        if st is None or st.numel() < need:
            st = b._gn_stats = b.buf(need, scratch=True)
        a.stats = st.data_ptr()
        a.y_f16, a.y_is_f32 = y.data_ptr(), 1
        b.keep += [gamma, beta, y] + [v.t for v in views]
        b._push(hip.OP_GN, 'gn', a)
        return [seg(View(y))]
    out, off = [], 0
    for v in views:
        assert v.width % gs == 0 and off % gs == 0
        out.append(seg(v, pro=hip.PRO_GN_SILU if silu else hip.PRO_GN, gamma=gamma, beta=beta, eps=eps, gs=gs, goff=off))
        off += v.width
    return out


def own(t, device, dtype=torch.float32):
    """Contiguous ``dtype`` copy of t on ``device`` that the plan OWNS.  The weight planners read a model's parameters in place when
    the model already sits on the GPU (samplers.state_dict_for): whatever a plan keeps must not alias a live parameter."""
    r = t.detach().to(device=device, dtype=dtype).contiguous()
    return r.clone() if r.data_ptr() == t.data_ptr() else r


def mm64(A, B):
    """A @ B in fp64 for the weight folds.  Host tensors: torch.  Device tensors: the library's es_matmul_f64 -- a fixed left fold
    over k per element, so that every process (object shards fold their own copy) derives bit-identical weights; a BLAS call on the
    GPU promises no summation order.  A vector B gives a vector."""
    A = A.detach().double()
    B = B.detach().double()
    if not A.is_cuda:
        return A @ B.to(A.device)
    vec = B.dim() == 1
    Bm = (B.unsqueeze(1) if vec else B).to(A.device).contiguous()
    A = A.contiguous()
    assert A.dim() == 2 and Bm.dim() == 2 and A.shape[1] == Bm.shape[0], (A.shape, B.shape)
    with torch.cuda.device(A.device):
        out = torch.empty(A.shape[0], Bm.shape[1], dtype=torch.float64, device=A.device)
        hip.check(hip.lib().es_matmul_f64(hip.ptr(A), hip.ptr(Bm), hip.ptr(out), A.shape[0], A.shape[1], Bm.shape[1], hip.current_stream()),
                  'es_matmul_f64')
    return out[:, 0] if vec else out


class PackedLinear:
    """Device image of one (possibly fused) linear layer in MFMA fragment order.
    ``geglu=True``: W = [value rows | gate rows] of a GEGLU projection; rows are interleaved per 16-row tile so
    that the kernel's epilogue (ES_ACT_GEGLU) can emit value*gelu(gate) directly (N_out = N/2)."""

    def __init__(self, W, b, device, geglu=False):
        self.N, self.K = W.shape
        self.geglu = geglu
        L = hip.lib()
        n = L.es_pack_linear_f32_size(self.N, self.K)
        if torch.device(device).type == 'cuda':
            # the re-layout runs on the GPU from the uploaded fp32 weight (es_pack_linear_f32_dev; bit-identical to the host loop)
            with torch.cuda.device(device):
                Wd = W.detach().to(device=device, dtype=torch.float32).contiguous()
                if geglu:                    # [value rows | gate rows] -> per 16-row tile 8 value rows + their 8 gate rows
                    Nh = self.N // 2
                    if Nh % 8:
                        raise ValueError('GEGLU projection with %d outputs (a multiple of 8 is needed)' % Nh)
                    idx = (torch.arange(Nh, device=device).view(-1, 1, 8) + torch.tensor([0, Nh], device=device).view(1, 2, 1)).reshape(-1)
                    Wd = Wd.index_select(0, idx)
                    b = None if b is None else b.detach().to(device=device, dtype=torch.float32).index_select(0, idx)
                out = torch.empty(n, dtype=torch.float32, device=device)
                hip.check(L.es_pack_linear_f32_dev(hip.ptr(Wd), self.N, self.K, hip.ptr(out), hip.current_stream()), 'es_pack_linear_f32_dev')
            self.w = out
            self.b = None if b is None else own(b, device)
            self.weight_bytes = self.N * self.K * 4
            self.nbatch = 1
            return
        W = W.detach().to(torch.float32).contiguous().cpu()
        out = torch.empty(n, dtype=torch.float32)
        if geglu:
            bh = None if b is None else b.detach().to(torch.float32).contiguous().cpu()
            bo = None if b is None else torch.empty_like(bh)
            hip.check(L.es_pack_linear_geglu_f32(C.c_void_p(W.data_ptr()), None if bh is None else C.c_void_p(bh.data_ptr()),
                                                 self.N // 2, self.K, C.c_void_p(out.data_ptr()),
                                                 None if bo is None else C.c_void_p(bo.data_ptr())), 'es_pack_linear_geglu_f32')
            b = bo
        else:
            hip.check(L.es_pack_linear_f32(C.c_void_p(W.data_ptr()), self.N, self.K, C.c_void_p(out.data_ptr())),
                      'es_pack_linear_f32')
        self.w = out.to(device)
        self.b = None if b is None else b.detach().to(torch.float32).contiguous().to(device)
        self.weight_bytes = self.N * self.K * 4
        self.nbatch = 1


class PackedLinearBatch(PackedLinear):
    """Several linears of identical shape stored back to back (one batched launch, grid.z = len)."""

    def __init__(self, Ws, bs, device):
        parts = [PackedLinear(W, b, device) for W, b in zip(Ws, bs)]
        self.N, self.K, self.geglu = parts[0].N, parts[0].K, False
        assert all(p.N == self.N and p.K == self.K for p in parts)
        self.w = torch.cat([p.w for p in parts])
        self.b = torch.cat([p.b for p in parts]) if parts[0].b is not None else None
        self.weight_bytes = sum(p.weight_bytes for p in parts)
        self.nbatch = len(parts)


def fold_bn(sd, lin, bn):
    """Linear followed by eval-mode BatchNorm1d -> one affine map (fp64 fold, fp32 result).
    y = ((xW^T + b) - mean) * gamma / sqrt(var + eps) + beta   (model/layers.py:27-31)."""
    W = sd[lin + '.weight'].double()
    b = sd[lin + '.bias'].double()
    s = sd[bn + '.weight'].double() / torch.sqrt(sd[bn + '.running_var'].double() + 1e-5)
    return (W * s[:, None]).float(), ((b - sd[bn + '.running_mean'].double()) * s + sd[bn + '.bias'].double()).float()


def fold_affine64(W, b, gamma, beta):
    """fold_affine in fp64, unrounded (for folds that continue)"""
    Wd = W.detach().double()
    g, be = gamma.detach().double().to(Wd.device), beta.detach().double().to(Wd.device)
    bb = (b.detach().double().to(Wd.device) if b is not None else torch.zeros(W.shape[0], dtype=torch.float64, device=Wd.device)) + mm64(Wd, be)
    return Wd * g[None, :], bb


def fold_affine(W, b, gamma, beta):
    """Linear(norm(x) * gamma + beta) = (W diag(gamma)) norm(x) + (W beta + b): the affine of a GroupNorm / LayerNorm that feeds a
    product directly (no SiLU in between) moves into the weights (fp64 fold) -- the kernels then neither load nor apply it
    (round 5: the norm prologue of 32 launches per layout step lost a third of its loads)."""
    Wd = W.detach().double()
    g, be = gamma.detach().double().to(Wd.device), beta.detach().double().to(Wd.device)
    bb = (b.detach().double().to(Wd.device) if b is not None else torch.zeros(W.shape[0], dtype=torch.float64, device=Wd.device)) + mm64(Wd, be)
    return (Wd * g[None, :]).float(), bb.float()


def centre_tap(w):
    """Conv1d(k=3, pad=1) on a length-1 signal touches only the centre tap;
    k=1 convs pass through."""
    return w[:, :, w.shape[2] // 2] if w.dim() == 3 else w


class GraphIndex:
    """Device index arrays of one scene graph: gather indices for s/o and the CSR that turns
    scatter_add + average (model/graph.py:172-199) into a deterministic segmented mean whose
    summation order equals the reference's (s-messages in triple order, then o-messages).

    ``capacity``: number of triple ROWS the plans built on this index process (>= the number of triples; the extra rows
    gather node 0 and are listed in no node's CSR segment, so they never reach a result).  Plans bake pointers and row
    counts, not indices: ``update()`` rewrites the arrays in place for another graph with the same node count and a triple
    count within the capacity, and every plan / captured hipGraph built on this object stays valid (eval_3dfront.py visits a
    different scene graph on every call)."""

    def __init__(self, triples, num_objs, device, capacity=None):
        self.O = int(num_objs)
        self.device = device
        n = int(triples.reshape(-1, 3).shape[0])
        self.T = max(int(capacity or 0), n)                      # rows the plans run over
        self.s = torch.zeros(max(self.T, 1), dtype=torch.int32, device=device)
        self.o = torch.zeros(max(self.T, 1), dtype=torch.int32, device=device)
        self._csr = {}
        self.update(triples)

    def update(self, triples):
        tri = triples.detach().cpu().numpy().astype(np.int64).reshape(-1, 3)
        n = int(tri.shape[0])
        if n > self.T:
            raise ValueError('graph has %d triples, capacity %d' % (n, self.T))
        if n and (tri[:, [0, 2]].min() < 0 or tri[:, [0, 2]].max() >= self.O):
            raise IndexError('triple endpoint out of range')
        self.n_triples = n
        pad = np.zeros((self.T - n, 3), dtype=np.int64)
        full = np.concatenate([tri, pad], 0) if self.T > n else tri
        self.s[:self.T].copy_(torch.from_numpy(full[:, 0].astype(np.int32)))
        self.o[:self.T].copy_(torch.from_numpy(full[:, 2].astype(np.int32)))
        self.p_host = full[:, 1].copy()                          # [capacity] predicate ids (padding rows: predicate 0)
        self.tri_host = tri                                      # the real triples
        for key in list(self._csr):
            self._fill_csr(key)

    def _fill_csr(self, key):
        off_s, off_o = key
        tri = self.tri_host
        rows, offs, ptr = [], [], [0]
        for n in range(self.O):
            ts = np.nonzero(tri[:, 0] == n)[0]
            to = np.nonzero(tri[:, 2] == n)[0]
            rows += ts.tolist() + to.tolist()
            offs += [off_s] * len(ts) + [off_o] * len(to)
            ptr.append(len(rows))
        cap = max(2 * self.T, 1)                                 # every triple is listed twice (subject + object side)
        if key not in self._csr:
            z = lambda k: torch.zeros(k, dtype=torch.int32, device=self.device)
            self._csr[key] = (z(self.O + 1), z(cap), z(cap))
        dp, dr, do = self._csr[key]
        dp.copy_(torch.tensor(ptr, dtype=torch.int32))
        dr[:len(rows)].copy_(torch.tensor(rows, dtype=torch.int32))
        do[:len(offs)].copy_(torch.tensor(offs, dtype=torch.int32))

    def csr(self, off_s, off_o):
        key = (off_s, off_o)
        if key not in self._csr:
            self._fill_csr(key)
        return self._csr[key]


class Rider:
    """Hands out the ops of an op-emitting generator ONE at a time, so that a chain of products which does not depend on another
    chain can ride on that chain's launches (``Builder.ride``: the runtime launches an op marked ``fuse_next`` and its successor
    as one grid).  The generator yields None after every op it has emitted and the string 'CTX' -- having emitted nothing --
    in front of the first op that needs the other chain's result; from then on nothing rides."""

    def __init__(self, gen):
        self.gen, self.blocked, self.done = gen, False, False

    def step(self):
        """emit ONE more op if it is independent of the host chain; True when an op was appended"""
        if self.blocked or self.done:
            return False
        try:
            r = next(self.gen)
        except StopIteration:
            self.done = True
            return False
        if r == 'CTX':
            self.blocked = True
            return False
        return True

    def finish(self):
        """emit everything that is left"""
        for _ in self.gen:
            pass
        self.done = True


class Builder:
    """Accumulates ops + keeps every referenced device tensor alive."""

    def __init__(self, device):
        self.device = device
        self.ops = []
        self.keep = []
        self.weight_bytes = 0      # algorithmic weight bytes streamed per plan execution
        self.tags = {}             # name -> View of an intermediate (parity debugging)
        # side-stream graph branches: measured SLOWER on MI355X (2.27 vs 2.06 ms per layout step) -- off by default
        self.use_lanes = os.environ.get('ES_LANES', '0') != '0'
        self.flops = 0
        self.allow_split = True    # K split over workgroups with slab outputs (Builder.linear(out=None))
        self.scratch = set()       # data_ptr of buffers the plan fully writes before it reads them (save_model does not dump them)

    def buf(self, *shape, dtype=torch.float32, zero=False, scratch=False):
        """``scratch=True``: every byte of the buffer is written by an op of the plan before any op reads it (activations, split-K
        workspace, statistics) and nothing writes it at build time -- model files list it without its contents."""
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        if scratch and not zero:
            self.scratch.add(t.data_ptr())
        return t

    def dev(self, t, dtype=torch.float32):
        t = t.detach().to(dtype).contiguous().to(self.device)
        self.keep.append(t)
        return t

    def fork(self, lane):
        if not self.use_lanes:
            return
        op = Op()
        op.kind, op.lane = hip.OP_FORK, lane
        self.ops.append(op)

    def join(self, lane):
        if not self.use_lanes:
            return
        op = Op()
        op.kind, op.lane = hip.OP_JOIN, lane
        self.ops.append(op)

    def linear(self, segs, pl, M, out=None, prologue=hip.PRO_NONE, gamma=None, beta=None, eps=0.0, act=hip.ACT_NONE,
               res=None, res2=None, lane=0, use_bias=True, a_bstride=0, out_bstride=0, split=None, fuse_next=False):
        """out = act(prologue(A) @ W^T + b) + res (+ res2).  ``out`` None: the output buffer is allocated here and, when the op
        allows it (no activation epilogue, no batching) and ``split`` is not False, K is split over workgroups -- the returned
        View is then a slab tensor.  ``split`` = int: k-blocks (16 columns) per slice; True / None: the library's choice.
        ``fuse_next``: the NEXT linear op emitted is independent of this one -- the runtime launches both as one grid."""
        a = LinearArgs()
        for i, s in enumerate(segs):
            a.seg[i] = s
            self.keep.append(getattr(s, '_keep', None))
        a.nseg = len(segs)
        a.M, a.K, a.N = M, pl.K, pl.N
        a.wpack = pl.w.data_ptr()
        a.bias = pl.b.data_ptr() if (pl.b is not None and use_bias) else None
        a.prologue = prologue
        a.gamma = gamma.data_ptr() if gamma is not None else None
        a.beta = beta.data_ptr() if beta is not None else None
        a.eps = eps
        a.act = hip.ACT_GEGLU if pl.geglu else act
        a.nbatch, a.a_bstride, a.out_bstride = pl.nbatch, a_bstride, out_bstride
        a.res = res.ptr if res is not None else None
        a.res_ld = res.ld if res is not None else 0
        a.res_nslab, a.res_slab_stride = (res.nslab, res.slab_stride) if res is not None else (0, 0)
        if res is not None and getattr(res, 'step', None) is not None:
            a.res_step, a.res_step_stride = res.step.data_ptr(), res.step_stride
            self.keep.append(res.step)
        a.res2 = res2.ptr if res2 is not None else None
        a.res2_ld = res2.ld if res2 is not None else 0
        a.res2_nslab, a.res2_slab_stride = (res2.nslab, res2.slab_stride) if res2 is not None else (0, 0)
        if out is None:
            Nout = pl.N // 2 if pl.geglu else pl.N
            S, kbps = 1, 0
            if split is not False and a.act == hip.ACT_NONE and pl.nbatch == 1 and self.allow_split:
                if isinstance(split, int) and not isinstance(split, bool):
                    kbps = split
                else:
                    kal = max([sg.gs for sg in segs if sg.pro in (hip.PRO_GN, hip.PRO_GN_SILU)] +
                              ([pl.K // 32] if prologue in (hip.PRO_GN, hip.PRO_GN_SILU) else []) + [16])
                    kbps = seg_aligned_kbps([sg.width for sg in segs], pl.K, pl.N, kal)
                a.kb_per_slice = kbps
                a.seg_slices = 1 if all(sg.width % 16 == 0 for sg in segs) else 0
                if a.seg_slices and kbps and not (isinstance(split, int) and not isinstance(split, bool)):
                    # a GroupNorm(+SiLU) segment next to plain ones is cut twice as fine while the slab / workgroup bounds allow it: the
                    # norm makes its k-blocks the expensive ones of the launch (the K = 1536 conv2 + skip products: 16 | 16 | 32 | 32 k-blocks)
                    gn = [sg.pro in (hip.PRO_GN, hip.PRO_GN_SILU) for sg in segs]
                    if any(gn) and not all(gn):
                        cuts = lambda half: sum((sg.width // 16 + (kbps + 1) // 2 - 1) // ((kbps + 1) // 2) if h else (sg.width // 16 + kbps - 1) // kbps
                                                for sg, h in zip(segs, half))
                        nct_ = (pl.N + 15) // 16
                        if ((kbps + 1) // 2) % max(1, kal // 16) == 0 and (kbps + 1) // 2 >= 8 and \
                                cuts(gn) <= min(ROWS_MAX_SLABS, ROWS_MAX_WGS // (2 * nct_)):
                            a.seg_slices |= sum(2 << i for i, h in enumerate(gn) if h)
                got = C.c_int(0)
                S = hip.lib().es_linear_rows_slices(C.byref(a), C.byref(got))
                a.kb_per_slice = got.value if S > 1 else 0
            buf = self.buf(S, M, Nout)
            out = View(buf[0], nslab=S, slab_stride=M * Nout if S > 1 else 0)
            a.out_slab_stride = M * Nout if S > 1 else 0
        else:
            assert out.nslab <= 1
        a.out = out.ptr
        a.out_ld = out.ld
        a.fuse_next = 1 if (fuse_next and not self.use_lanes) else 0
        op = Op()
        op.kind, op.lane = hip.OP_LINEAR, (lane if self.use_lanes else 0)
        op.u.linear = a
        self.ops.append(op)
        self.keep += [pl, gamma, beta, out.t, res.t if res is not None else None, res2.t if res2 is not None else None]
        self.weight_bytes += pl.weight_bytes
        self.flops += 2 * M * pl.K * pl.N * pl.nbatch
        return out

    def ride(self, rider):
        """The op emitted LAST (a rows product, or the last member of a fused group of two) takes the next independent op of
        ``rider`` onto its launch: the rider's op is appended right behind it and the pair is marked for one grid.  Problems of one
        launch must be independent -- the rider's chain only reads its own earlier ops and inputs of the step."""
        if rider is None or self.use_lanes or not ROWS_RIDE or not self.allow_split:
            return False
        i = len(self.ops) - 1
        host = self.ops[i]
        if host.kind != hip.OP_LINEAR or host.u.linear.nbatch > 1 or host.u.linear.act == hip.ACT_GEGLU:
            return False
        n_group = 1
        if i > 0 and self.ops[i - 1].kind == hip.OP_LINEAR and self.ops[i - 1].u.linear.fuse_next:
            n_group = 2
            if i > 1 and self.ops[i - 2].kind == hip.OP_LINEAR and self.ops[i - 2].u.linear.fuse_next:
                return False                       # three problems already
        if not rider.step():
            return False
        assert len(self.ops) == i + 2 and self.ops[i + 1].kind == hip.OP_LINEAR and not self.ops[i + 1].u.linear.fuse_next, \
            'a rider step emits exactly one rows product'
        self.ops[i].u.linear.fuse_next = 1
        return n_group + 1

    def update(self, kind, x, eps, coef, step, noise=None, noise_stride=0, inc_step=True, clip_x0=False):
        """eps: tensor or (slab) View"""
        a = UpdateArgs()
        if isinstance(eps, View):
            assert eps.ld == eps.width and eps.col == 0
            a.x, a.eps, a.eps_nslab, a.eps_slab_stride = x.data_ptr(), eps.ptr, eps.nslab, eps.slab_stride
            self.keep.append(eps.t)
        else:
            a.x, a.eps = x.data_ptr(), eps.data_ptr()
        a.noise = noise.ptr if noise is not None else None
        a.noise_stride = noise_stride
        a.coef, a.coef_stride = coef.data_ptr(), coef.shape[1]
        a.step = step.data_ptr()
        a.n = x.numel()
        a.inc_step = 1 if inc_step else 0
        a.clip_x0 = 1 if clip_x0 else 0
        op = Op()
        op.kind, op.lane = kind, 0
        op.u.update = a
        self.ops.append(op)

    def rowsel(self, table, step, out, rows=1):
        """out[r, :n] = table[*step, :n] for r < rows (out: View; table: 2-D tensor [n_steps, n])"""
        a = RowSelArgs()
        a.table, a.stride, a.step = table.data_ptr(), table.shape[1], step.data_ptr()
        a.out, a.out_ld, a.rows, a.n = out.ptr, out.ld, rows, table.shape[1]
        op = Op()
        op.kind, op.lane = hip.OP_ROWSEL, 0
        op.u.rowsel = a
        self.ops.append(op)
        self.keep.append(table)

    def copy(self, dst, src, nbytes, rows=0, dst_pitch=0, src_pitch=0):
        """flat copy of nbytes, or (rows > 1) a 2-D copy of rows x nbytes with byte pitches"""
        a = CopyArgs()
        a.dst, a.src, a.bytes = dst, src, nbytes
        a.rows, a.dst_pitch, a.src_pitch = rows, dst_pitch, src_pitch
        op = Op()
        op.kind, op.lane = hip.OP_COPY, 0
        op.u.copy = a
        self.ops.append(op)

    def finish(self):
        return Plan(self)


class Plan:
    def __init__(self, b):
        self.keep = b.keep
        self.tags = b.tags
        self.n_ops = len(b.ops)
        self.weight_bytes = b.weight_bytes
        self.flops = b.flops
        self.scratch = set(getattr(b, 'scratch', ()))
        arr = (Op * len(b.ops))(*b.ops)
        self._arr = arr
        self.handle = hip.lib().es_plan_create(arr, len(b.ops))
        if not self.handle:
            raise RuntimeError('es_plan_create: ' + hip.lib().es_last_error().decode())

    @property
    def n_launches(self):
        """kernel launches of one execution: the runtime's grouping of rows products marked ``fuse_next`` (es_plan_run: up to three
        independent products per grid), one launch per other op, + the step increment behind a sampler update"""
        ops, i, n = self._arr, 0, 0
        while i < len(ops):
            op, k = ops[i], 1
            if op.kind == hip.OP_LINEAR and op.u.linear.fuse_next:
                while k < 3 and ops[i + k - 1].u.linear.fuse_next and i + k < len(ops) and ops[i + k].kind == hip.OP_LINEAR \
                        and ops[i + k].lane == op.lane:
                    k += 1
            if op.kind not in (hip.OP_FORK, hip.OP_JOIN):
                n += 1
            if op.kind in (hip.OP_DDPM, hip.OP_DDIM) and op.u.update.inc_step and not (op.kind == hip.OP_DDPM and op.u.update.n <= 4096):
                n += 1                           # (the one-workgroup DDPM update advances the step counter itself)
            i += k
        return n

    def run(self):
        hip.check(hip.lib().es_plan_run(C.c_void_p(self.handle), hip.current_stream()), 'es_plan_run')

    def poison_scratch(self):
        """Verification of the ``scratch=True`` claim (model files store such buffers without their contents and the loader
        zero-fills them): overwrite every scratch buffer with NaN bit patterns.  A plan whose results after this differ from before
        -- or are not finite -- reads a byte no op of it wrote.  Returns the number of bytes poisoned."""
        n = 0
        for t in self.keep:
            if isinstance(t, torch.Tensor) and t.data_ptr() in self.scratch and t.numel():
                t.view(torch.uint8).fill_(0xFF)          # 0xFFFFFFFF / 0xFFFF: a NaN in fp32 and fp16, -1 in the integer types
                n += t.numel() * t.element_size()
        return n

    def sample(self, step, first_step, n_steps, use_graph=True):
        hip.check(hip.lib().es_sampler_run(C.c_void_p(self.handle), C.c_void_p(step.data_ptr()), first_step,
                                           n_steps, 1 if use_graph else 0, hip.current_stream()), 'es_sampler_run')

    def __del__(self):
        try:
            if self.handle:
                hip.lib().es_plan_destroy(C.c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass


def save_model(plan, path, regions):
    """Serialise ``plan`` and every device allocation it references into a model file that a host WITHOUT Python loads with
    ``es_model_load`` and runs with ``es_layout_sample`` / ``es_shape_sample`` / ``es_vq_decode`` (include/echoscene_hip.h).
    ``regions``: name -> tensor, the caller's I/O areas ("x", "noise", "step", "z", "sdf").  The pointer fields of the ops are
    found through ``es_op_pointer_offsets`` (one table, in the library) and mapped to the allocator blocks that contain them;
    the library rewrites them as (buffer, offset) and dumps the buffers."""
    import bisect
    L = hip.lib()
    torch.cuda.synchronize()
    ptrs = set()
    offs = (C.c_size_t * 64)()
    for op in plan._arr:
        n = L.es_op_pointer_offsets(op.kind, offs, 64)
        if n < 0:
            raise RuntimeError('save_model: op kind %d has no pointer table' % op.kind)
        base = C.addressof(op)
        for j in range(n):
            v = C.c_uint64.from_address(base + offs[j]).value
            if v:
                ptrs.add(v)
    for t in regions.values():
        ptrs.add(t.data_ptr())
    blocks = []
    for sg in torch.cuda.memory_snapshot():
        a = sg['address']
        for blk in sg['blocks']:
            if blk['state'] == 'active_allocated':
                blocks.append((a, blk['size']))
            a += blk['size']
    blocks.sort()
    starts = [b0 for b0, _ in blocks]
    used = {}
    for p in sorted(ptrs):
        i = bisect.bisect_right(starts, p) - 1
        if i < 0 or p >= blocks[i][0] + blocks[i][1]:
            raise RuntimeError('save_model: pointer 0x%x is not inside a live torch allocation' % p)
        used[i] = blocks[i]
    bl = [used[i] for i in sorted(used)]
    bufs = (hip.BufferDesc * len(bl))()
    region_ptrs = {t.data_ptr() for t in regions.values()}
    stored = 0
    for k, (a, n) in enumerate(bl):
        # scratch blocks (Builder.buf(scratch=True): written by the plan before they are read) are listed, not dumped
        sc = a in plan.scratch and a not in region_ptrs
        bufs[k].ptr, bufs[k].bytes = a, n | ((1 << 63) if sc else 0)
        stored += 0 if sc else n
    regs = (hip.RegionDesc * len(regions))()
    for k, (name, t) in enumerate(regions.items()):
        regs[k].name = name.encode()
        regs[k].ptr, regs[k].bytes = t.data_ptr(), t.numel() * t.element_size()
    hip.check(L.es_model_save(str(path).encode(), C.c_void_p(plan.handle), bufs, len(bl), regs, len(regions)), 'es_model_save')
    return stored


def combine_plans(device, main, side, side_repeat=1):
    """One plan = ``main``'s ops on lane 0 and ``side_repeat`` copies of ``side``'s ops on lane 1 (forked before, joined after):
    captured as ONE hipGraph with two parallel branches, so the latency-bound layout chain (32 workgroups per launch) can run
    in the gaps of the MFMA-bound shape step instead of after it.  Both plans keep their own buffers / step counters."""
    import copy
    b = Builder(device)
    b.use_lanes = True
    fork, join = Op(), Op()
    fork.kind, fork.lane = hip.OP_FORK, 1
    join.kind, join.lane = hip.OP_JOIN, 1
    ops = [fork]
    for _ in range(side_repeat):
        for op in side._arr:
            if op.kind in (hip.OP_FORK, hip.OP_JOIN):
                continue                   # the side plan's own lanes collapse into lane 1 (an inner JOIN would stall the main stream)
            o = Op()
            C.memmove(C.byref(o), C.byref(op), C.sizeof(Op))
            o.lane = 1
            ops.append(o)
    for op in main._arr:
        o = Op()
        C.memmove(C.byref(o), C.byref(op), C.sizeof(Op))
        ops.append(o)
    ops.append(join)
    b.ops = ops
    b.keep = [main, side]
    b.weight_bytes, b.flops = main.weight_bytes + side.weight_bytes * side_repeat, main.flops + side.flops * side_repeat
    return b.finish()


# ------------------------------------------------------------------------------------------------
# GraphTripleConvNet  (reference model/graph.py:89-250)
# ------------------------------------------------------------------------------------------------
def Dp_of(L):
    """predicate width of a packed GraphTripleConv layer: net1's output is [H | Dp | H] wide (graph.py:108)"""
    return L['n1b'].N - 2 * L['H']


class GCNWeights:
    """Packed weights of one GraphTripleConvNet: BatchNorm folded into the Linears."""

    def __init__(self, sd, prefix, device, pooling='avg'):
        if pooling not in ('avg', 'sum', 'wAvg'):
            raise ValueError('Invalid pooling "%s"' % pooling)              # graph.py:105
        self.pooling = pooling
        self.layers = []
        i = 0
        while f'{prefix}.gconvs.{i}.net1.0.weight' in sd:
            p = f'{prefix}.gconvs.{i}'
            bn = (p + '.net1.1.running_mean') in sd
            j2 = 3 if bn else 2                     # index of the second Linear inside build_mlp

            def lin(name, idx):
                q = f'{p}.{name}.{idx}'
                if bn:
                    W, b = fold_bn(sd, q, f'{p}.{name}.{idx + 1}')
                else:
                    W, b = sd[q + '.weight'], sd[q + '.bias']
                return PackedLinear(W, b, device)

            L = dict(n1a=lin('net1', 0), n1b=lin('net1', j2), n2a=lin('net2', 0), n2b=lin('net2', j2))
            L['H'] = L['n2a'].K
            L['Dout'] = L['n2b'].N
            L['Dobj'] = None
            if (p + '.linear_projection.weight') in sd:
                L['proj'] = PackedLinear(sd[p + '.linear_projection.weight'], sd[p + '.linear_projection.bias'], device)
                L['projp'] = PackedLinear(sd[p + '.linear_projection_pred.weight'],
                                          sd[p + '.linear_projection_pred.bias'], device)
                L['Dobj'] = L['proj'].K
            if pooling == 'wAvg':
                # WeightNetGCN (graph.py:37-86): two down-sampling Linears to 128 features, then one head per triple slot
                # (Linear(384, 64) + ReLU + Linear(64, 1) + Sigmoid).  The two heads read the same features: their first Linears
                # are stacked into one [128, 384] product and their second ones into a block-diagonal [2, 128] product (the zero
                # blocks add exact zeros, every head's 64 terms keep their order).
                q = p + '.weightNet'
                L['wn_obj'] = PackedLinear(sd[q + '.down_sample_obj.weight'], sd[q + '.down_sample_obj.bias'], device)
                L['wn_pred'] = PackedLinear(sd[q + '.down_sample_pred.weight'], sd[q + '.down_sample_pred.bias'], device)
                if (q + '.Net_s.0.weight') in sd:
                    heads = ('Net_s', 'Net_o')
                else:                                                       # separate_s_o=False: one head for both slots
                    heads = ('Net', 'Net')
                W1 = torch.cat([sd[f'{q}.{h}.0.weight'].float() for h in heads], 0)
                b1 = torch.cat([sd[f'{q}.{h}.0.bias'].float() for h in heads], 0)
                nh = sd[f'{q}.{heads[0]}.0.weight'].shape[0]
                W2 = torch.zeros(2, 2 * nh)
                W2[0, :nh] = sd[f'{q}.{heads[0]}.2.weight'].float()[0]
                W2[1, nh:] = sd[f'{q}.{heads[1]}.2.weight'].float()[0]
                b2 = torch.cat([sd[f'{q}.{h}.2.bias'].float() for h in heads], 0)
                L['wn_h'] = PackedLinear(W1, b1, device)
                L['wn_w'] = PackedLinear(W2, b2, device)
                if L['wn_pred'].K != Dp_of(L):
                    raise ValueError("pooling='wAvg': the weighting net down-samples predicates of width %d, the layer's are %d wide "
                                     "(the reference's WeightNetGCN(hidden_dim, output_dim, 128) needs output_dim == input_dim_pred)"
                                     % (L['wn_pred'].K, Dp_of(L)))
            self.layers.append(L)
            i += 1
        if not self.layers:
            raise KeyError('no GraphTripleConvNet under ' + prefix)


def emit_gcn(b, gw, g, obj, Dobj, pred, Dp, out=None, want_pred=False, rider=None):
    """obj: View [O, Dobj]; pred: View [T, Dp]; returns View of the last layer's object output
    (and the predicate output when ``want_pred`` -- the samplers never consume it).
    ``rider`` (Rider): an independent chain of products that rides on this chain's launches -- one op on net1's second Linear
    and one on net2's second Linear of every layer (the gather / segmented-mean launches keep their lean kernels).
    The two hidden products of a layer (net1's first Linear over the gathered triples, net2's first Linear over the pooled
    messages) split K over workgroups: their ReLU is applied by the consumer to the slab sum (``pre_act``)."""
    O, T = g.O, g.T
    n = len(gw.layers)
    b.keep.append(g)            # the plan references the graph's device index arrays
    for li, L in enumerate(gw.layers):
        H, Dout = L['H'], L['Dout']
        W2 = 2 * H + Dp
        last = li == n - 1
        has_proj = 'proj' in L
        need_newp = has_proj and (not last or want_pred)
        # net1's first Linear over the gathered triples and the residual projection of the node vectors depend only on the
        # layer input: ONE launch (fuse_next); likewise net2's first Linear over the pooled messages and the predicate projection.
        # (the triple products have ~4x the rows of the node products, i.e. 4x the workgroups per slice: 2-4 slices instead of the
        #  8 the (K, N) rule would pick -- the consumer reads S slabs of [T x H]; a planner constant, not a function of M)
        gs_ = ROWS_GCN_SLICES
        nkb1 = (2 * Dobj + Dp + 15) // 16
        kb1 = max(8, (nkb1 + gs_ - 1) // gs_) if gs_ > 1 else 0
        seg_ok = Dobj % 16 == 0 and Dp % 16 == 0
        t1 = b.linear([seg(obj, hip.SEG_GATHER, idx=g.s, width=Dobj), seg(pred, width=Dp),
                       seg(obj, hip.SEG_GATHER, idx=g.o, width=Dobj)], L['n1a'], T, fuse_next=has_proj,
                      split=kb1 if kb1 and (not seg_ok or max(Dobj, Dp) // 16 <= 96) else False)             # relu deferred
        if has_proj:
            b.fork(2)
            # (one K slice: with the triple-row product the launch then stays within the 512 workgroups that are resident at once)
            proj = b.linear([seg(obj, width=Dobj)], L['proj'], O, lane=2, split=False)
        else:
            proj = None
        t2 = View(b.buf(T, W2))
        b.linear([seg(t1, pre_act=hip.ACT_RELU)], L['n1b'], T, t2, act=hip.ACT_RELU)
        if ROWS_RIDE >= 2:
            b.ride(rider)
        ptr, rows, offs = g.csr(0, H + Dp)
        pooling = getattr(gw, 'pooling', 'avg')
        wts = None
        if pooling == 'wAvg':
            # s_weights, o_weights = weightNet(new_s, new_p, new_o) (graph.py:165-167): feat = [down(s) | down(o) | down(p)]
            feat = b.buf(T, 3 * L['wn_obj'].N)
            nf = L['wn_obj'].N
            b.linear([seg(View(t2.t, col=0, ld=W2, width=H))], L['wn_obj'], T, View(feat, col=0, width=nf))
            b.linear([seg(View(t2.t, col=H + Dp, ld=W2, width=H))], L['wn_obj'], T, View(feat, col=nf, width=nf))
            b.linear([seg(View(t2.t, col=H, ld=W2, width=Dp))], L['wn_pred'], T, View(feat, col=2 * nf, width=nf))
            hid = View(b.buf(T, L['wn_h'].N))
            b.linear([seg(View(feat))], L['wn_h'], T, hid, act=hip.ACT_RELU)
            wts = View(b.buf(T, 2))
            b.linear([seg(hid)], L['wn_w'], T, wts, act=hip.ACT_SIGMOID)
        mode = {'avg': hip.SEG_CSRMEAN, 'sum': hip.SEG_CSRSUM, 'wAvg': hip.SEG_CSRWAVG}[pooling]
        n1 = b.linear([seg(View(t2.t, col=0, ld=W2, width=H), mode, idx=ptr, ent_row=rows, ent_off=offs, ent_wt=wts)],
                      L['n2a'], O)                                                           # relu deferred
        dst = out if (last and out is not None) else View(b.buf(O, Dout))
        if proj is not None:
            b.join(2)
        # net2's second Linear and the predicate projection (it needs net1's predicate columns and the layer's predicates, nothing of
        # net2) are independent: ONE launch.  (Round 5: the projection sat on the pooling launch before; a fused group takes ONE kernel
        # family, the pooling is k_linear_rows', and a product must take the same route -- the same bits -- whatever it is fused with.)
        b.linear([seg(n1, pre_act=hip.ACT_RELU)], L['n2b'], O, dst, act=hip.ACT_RELU, res=proj, fuse_next=need_newp)
        if need_newp:
            newp = View(b.buf(T, Dp))
            b.linear([seg(pred, width=Dp)], L['projp'], T, newp, res=View(t2.t, col=H, ld=W2, width=Dp), lane=2)
        elif not has_proj:
            newp = View(t2.t, col=H, ld=W2, width=Dp)
        b.ride(rider)
        obj, Dobj = dst, Dout
        if not last or want_pred:
            pred = newp
    return (obj, pred) if want_pred else obj


# ------------------------------------------------------------------------------------------------
# UNet1DModel -- the layout denoiser as a chain of fused row-linears
# (reference denoise_net.py:773-806; block semantics :293-313, attention.py:172-245, 385-396)
# ------------------------------------------------------------------------------------------------
def time_tables(w, temb, t_lin, device):
    """Timestep-dependent, node-independent products tabulated for every step of the schedule, on the HIP path:
    emb = time_embed(temb[i]) [n, 4mc]; emb_all = all ResBlock emb_layers(silu(emb)) [n, sum cout];
    t_lin(emb) [n, 64] (box_time_emb / shape_time_emb) or None."""
    n, mc = temb.shape
    b = Builder(device)
    e1 = View(b.buf(n, 4 * mc))
    b.linear([seg(View(temb))], w.te0, n, e1, act=hip.ACT_SILU)
    emb = View(b.buf(n, 4 * mc))
    b.linear([seg(e1)], w.te2, n, emb)
    emb_all = b.buf(n, w.emb_all.N)
    b.linear([seg(emb)], w.emb_all, n, View(emb_all), prologue=hip.PRO_SILU)
    tl = None
    if t_lin is not None:
        tl = b.buf(n, t_lin.N)
        b.linear([seg(emb)], t_lin, n, View(tl))
    b.finish().run()
    torch.cuda.synchronize()
    return dict(emb=emb.t, emb_all=emb_all, t_lin=tl)


class UNet1DWeights:
    def __init__(self, sd, net, device):
        """sd: state_dict of the UNet1DModel holder ``net`` (keys without prefix)."""
        self.device = device
        self.mc = net.model_channels
        self.topo = net.topo
        self.enable_t_emb = net.enable_t_emb
        self.in_ch, self.out_ch = net.in_channels, net.out_channels
        self.concat = bool(getattr(net, 'concat', False))
        self.heads = net.num_heads
        dv = lambda k: own(sd[k], device)
        P = lambda w, bname: PackedLinear(centre_tap(sd[w]), sd[bname] if bname else None, device)
        self.te0 = P('time_embed.0.weight', 'time_embed.0.bias')
        self.te2 = P('time_embed.2.weight', 'time_embed.2.bias')
        self.box_emb = P('box_embeddings.weight', 'box_embeddings.bias')
        self.box_t = P('box_time_emb.weight', 'box_time_emb.bias') if net.enable_t_emb else None
        self.pred_table = sd['pred_embeddings.weight'].detach().float().cpu()
        self.gcn = GCNWeights(sd, 'box_graph_cov', device)
        self.ctx_dim = self.gcn.layers[-1]['Dout']
        inp, mid, out = net.topo
        names = [(f'input_blocks.{i}.{j}', it) for i, blk in enumerate(inp) for j, it in enumerate(blk)]
        names += [(f'middle_block.{j}', it) for j, it in enumerate(mid)]
        names += [(f'output_blocks.{i}.{j}', it) for i, blk in enumerate(out) for j, it in enumerate(blk)]
        self.items = {}
        # channel counts the trunk's GroupNorms run over (ResBlock in / out layers, attention / transformer input norms, the output norm)
        self.gn_channels = sorted({c for _, it in names for c in (it[1:3] if it[0] == 'res' else it[1:2] if it[0] == 'attn' else ())}
                                  | {self.mc})
        emb_w, emb_b, self.emb_slices, off = [], [], {}, 0
        ca_v, ca_b, self.ca = [], [], {}
        for name, it in names:
            kind = it[0]
            d = {}
            if kind == 'conv_in':
                d['conv'] = P(name + '.weight', name + '.bias')
            elif kind == 'res':
                d['gn1'] = (dv(name + '.in_layers.0.weight'), dv(name + '.in_layers.0.bias'))
                d['conv1'] = P(name + '.in_layers.2.weight', name + '.in_layers.2.bias')
                d['gn2'] = (dv(name + '.out_layers.0.weight'), dv(name + '.out_layers.0.bias'))
                d['conv2'] = P(name + '.out_layers.3.weight', name + '.out_layers.3.bias')
                if (name + '.skip_connection.weight') in sd:
                    d['skip'] = P(name + '.skip_connection.weight', name + '.skip_connection.bias')
                    # out = conv2(silu(GN2(h1))) + skip(x): with the norm done by conv1's epilogue both are plain
                    # products -> ONE op over the K-concatenation [h1n | x] with weights [W2 | Wskip]
                    d['conv2skip'] = PackedLinear(torch.cat([centre_tap(sd[name + '.out_layers.3.weight']),
                                                             centre_tap(sd[name + '.skip_connection.weight'])], 1),
                                                  sd[name + '.out_layers.3.bias'] + sd[name + '.skip_connection.bias'], device)
                emb_w.append(sd[name + '.emb_layers.1.weight'])
                emb_b.append(sd[name + '.emb_layers.1.bias'])
                self.emb_slices[name] = (off, it[2])
                off += it[2]
            elif kind == 'attn' and self.concat:
                # AttentionBlock on ONE token: softmax over one key == 1, so the block is
                # x + proj_out(V-rows of qkv(GroupNorm(x)))  (QKVAttentionLegacy rows are [head][q|k|v][ch]);
                # the two 1x1 convs are folded into one matrix in fp64.
                Cc = it[1]
                ch = Cc // self.heads
                vrows = torch.cat([torch.arange(h * 3 * ch + 2 * ch, h * 3 * ch + 3 * ch) for h in range(self.heads)])
                Wv = centre_tap(sd[name + '.qkv.weight']).double()[vrows]
                bv = sd[name + '.qkv.bias'].double()[vrows]
                Wp = centre_tap(sd[name + '.proj_out.weight']).double()
                d['gn'] = (None, None)               # (affine folded into the product)
                d['av'] = PackedLinear(*fold_affine(mm64(Wp, Wv), mm64(Wp, bv) + sd[name + '.proj_out.bias'].double(),
                                                    sd[name + '.norm.weight'], sd[name + '.norm.bias']), device)
            elif kind == 'attn':
                tb = name + '.transformer_blocks.0'
                # the three norms of a transformer block feed a product directly: their affines are folded into its weights
                d['gn'] = (None, None)
                d['proj_in'] = PackedLinear(*fold_affine(centre_tap(sd[name + '.proj_in.weight']), sd[name + '.proj_in.bias'],
                                                         sd[name + '.norm.weight'], sd[name + '.norm.bias']), device)
                d['ln1'] = (None, None)
                d['ln3'] = (None, None)
                # one token, one key: softmax == 1, so attention(x) = to_out(to_v(.)) exactly
                # ... and the two linears fold into one matrix (fp64): to_out . to_v
                d['vo1'] = PackedLinear(*fold_affine(mm64(sd[tb + '.attn1.to_out.0.weight'], sd[tb + '.attn1.to_v.weight']),
                                                     sd[tb + '.attn1.to_out.0.bias'], sd[tb + '.norm1.weight'], sd[tb + '.norm1.bias']), device)
                d['ff1'] = PackedLinear(*fold_affine(sd[tb + '.ff.net.0.proj.weight'], sd[tb + '.ff.net.0.proj.bias'],
                                                     sd[tb + '.norm3.weight'], sd[tb + '.norm3.bias']), device, geglu=True)
                if ROWS_FOLD_ATTN1 and it[1] <= 512 and it[1] % 16 == 0:
                    # [t0 | u] = GN(x) [Wp ; W1 P Wp]^T + [bp ; W1 P bp],  t2 = rstd(t0) u + t0 + (cav + b): W1 P = W1 with the mean of
                    # every row removed; the bias b of the self-attention rides in the cross-attention vector's bias (below) --
                    # a plan that keeps the self-attention launch for such a block (kernel family 0, the separate-GroupNorm route)
                    # runs it without its own bias (d['a1_bias_in_cav'])
                    Wp64, bp64 = fold_affine64(centre_tap(sd[name + '.proj_in.weight']), sd[name + '.proj_in.bias'],
                                               sd[name + '.norm.weight'], sd[name + '.norm.bias'])
                    W164, b164 = fold_affine64(mm64(sd[tb + '.attn1.to_out.0.weight'], sd[tb + '.attn1.to_v.weight']),
                                               sd[tb + '.attn1.to_out.0.bias'], sd[tb + '.norm1.weight'], sd[tb + '.norm1.bias'])
                    W1P = W164 - W164.mean(dim=1, keepdim=True)
                    d['proj_in_u'] = PackedLinear(torch.cat([Wp64, mm64(W1P, Wp64)], 0).float(), torch.cat([bp64, mm64(W1P, bp64)], 0).float(), device)
                    d['a1_bias_in_cav'] = b164
                # x_out = proj_out(ff2(g) + b2 + t2) + x_in is linear in (g, t2): ONE op over the K-concatenation [g | t2] with
                # [Wpo.Wff2 | Wpo] (folded in fp64) -- one dependent launch less per transformer block
                Wpo = centre_tap(sd[name + '.proj_out.weight']).double()
                Wf2 = sd[tb + '.ff.net.2.weight'].double()
                d['ff2po'] = PackedLinear(torch.cat([mm64(Wpo, Wf2), Wpo], 1).float(),
                                          (mm64(Wpo, sd[tb + '.ff.net.2.bias']) + sd[name + '.proj_out.bias'].double()).float(),
                                          device)
                # cross-attention with ONE key: out = to_out2(to_v2(ctx)) -- linear in ctx, so the two matrices fold into one
                # [C x ctx_dim] matrix per block (fp64): all blocks' vectors are ONE product per step
                self.ca[name] = (len(ca_v), it[1])
                ca_v.append(mm64(sd[tb + '.attn2.to_out.0.weight'], sd[tb + '.attn2.to_v.weight']).float())
                ca_b.append(sd[tb + '.attn2.to_out.0.bias'] if 'a1_bias_in_cav' not in d else
                            (sd[tb + '.attn2.to_out.0.bias'].detach().double().to(d['a1_bias_in_cav'].device) + d['a1_bias_in_cav']).float())
            elif kind == 'down':
                d['conv'] = P(name + '.op.weight', name + '.op.bias')
            elif kind == 'up':
                d['conv'] = P(name + '.conv.weight', name + '.conv.bias')
            self.items[name] = d
        # all ResBlock emb projections / all cross-attention value projections as ONE product each
        self.emb_all = PackedLinear(torch.cat(emb_w, 0), torch.cat(emb_b, 0), device)
        if not self.concat:
            self.cav_all = PackedLinear(torch.cat(ca_v, 0), torch.cat(ca_b, 0), device)
        self.out_gn = (dv('out.0.weight'), dv('out.0.bias'))
        self.out_conv = P('out.2.weight', 'out.2.bias')


def emit_unet1d_step(b, w, g, x, obj_embed_dev, temb, step, eps_out, tables=None):
    """One UNet1DModel.forward on x [O, in_ch] -> eps_out [O, out_ch].
    ``tables`` (time_tables): the time MLP / emb projections are read from per-schedule tables by the step counter
    instead of being recomputed every step (all nodes share t)."""
    O, mc = g.O, w.mc
    E = 4 * mc
    gdim = 64
    emb_ld = w.emb_all.N
    if tables is not None:
        emb = None
        # (round 5: the consumers read row *step of the table in place -- es_linear_args.res_step -- instead of a row-select launch)
        emb_all = (tables['emb_all'], step)
        emb_ld = 0                               # one row, broadcast over the nodes
        b.fork(1)
    else:
        # timestep MLP (all nodes share t: the table row is broadcast with ld = 0)
        e1 = View(b.buf(O, E))
        b.linear([seg(View(temb, ld=0, width=mc), step=step, step_stride=mc)], w.te0, O, e1, act=hip.ACT_SILU)
        emb = View(b.buf(O, E))
        b.linear([seg(e1)], w.te2, O, emb)
        # side lane 1: all 22 ResBlock time projections (92 MB of weights) overlap the GCN chain
        emb_all = b.buf(O, w.emb_all.N)
        b.fork(1)
        b.linear([seg(emb)], w.emb_all, O, View(emb_all), prologue=hip.PRO_SILU, lane=1)
    # GCN input  [obj_embed | box_embeddings(x_t) | box_time_emb(emb)]   (denoise_net.py:758-771)
    Dobj = obj_embed_dev.shape[1] + gdim + (gdim if w.enable_t_emb else 0)
    objbuf = b.buf(O, Dobj)
    oe_w = obj_embed_dev.shape[1]
    objbuf[:, :oe_w].copy_(obj_embed_dev)        # constant over the loop: written once at plan build
    # The trunk below is emitted by a generator: its head (conv_in ... the first transformer's proj_in) needs x and the time
    # tables only, not the GCN output, and rides on the launches of the GCN chain (Rider / Builder.ride) -- conv_in on the box
    # embedding's launch (both read x_t), then one product per net1 / net2 output launch of the GCN layers.
    box = {}
    if not all(rows_gn_in_registers(c) for c in w.gn_channels):
        # a GroupNorm of the trunk has a group size the rows kernels do not reduce in registers (model_channels = 384: groups of
        # 12): those norms are separate launches over whole matrices (norm_segs) -- no K-split slab outputs, nothing rides
        b.allow_split = False
    rider = Rider(_trunk(b, w, O, x, emb_all, emb_ld, box, eps_out))
    if w.enable_t_emb and tables is not None:
        # the time slot of the node vectors = row *step of the t_lin table, broadcast: an IDENTITY product over a step-indexed segment
        # (exact: sums of x * 1 and zeros), a third independent problem of the box-embedding launch instead of a row-select launch
        if not hasattr(w, 'eye_t'):
            w.eye_t = PackedLinear(torch.eye(gdim), None, b.device)
        b.linear([seg(View(tables['t_lin'], ld=0, width=gdim), step=step, step_stride=gdim)], w.eye_t, O,
                 View(objbuf, col=oe_w + gdim, ld=Dobj, width=gdim), fuse_next=True)
    b.linear([seg(View(x))], w.box_emb, O, View(objbuf, col=oe_w, ld=Dobj, width=gdim))
    b.ride(rider)
    if w.enable_t_emb and tables is None:
        b.linear([seg(emb)], w.box_t, O, View(objbuf, col=oe_w + gdim, ld=Dobj, width=gdim))
    pred = b.pred_rows = b.dev(w.pred_table[torch.from_numpy(g.p_host)])     # refreshed in place for a new graph
    ctx = emit_gcn(b, w.gcn, g, View(objbuf), Dobj, View(pred), pred.shape[1], rider=rider)
    box['ctx'] = ctx
    b.tags.update(ctx=ctx, gcn_in=View(objbuf))
    if emb is not None:
        b.tags['emb'] = emb
    cavo = box['cavo'] = {}
    if not w.concat:
        # the cross-attention vectors of all transformer blocks: one product (folded to_out2 . to_v2 matrices)
        cs_ = ROWS_CAV_SLICES
        if cs_ > 1:
            cavv = b.linear([seg(ctx)], w.cav_all, O, split=max(8, ((w.cav_all.K + 15) // 16 + cs_ - 1) // cs_))
        else:
            cavv = View(b.buf(O, w.cav_all.N))
            b.linear([seg(ctx)], w.cav_all, O, cavv)
        coff = 0
        for name, (k, Cc) in w.ca.items():
            cavo[name] = cavv.cols(coff, Cc)
            coff += Cc
    b.join(1)                                  # emb_all (side lane, forked after the time MLP)
    rider.finish()                             # the rest of the trunk
    return objbuf


def _trunk(b, w, O, x, emb_all, emb_ld, box, eps_out):
    """The UNet1D trunk (input / middle / output blocks and the output conv) as a GENERATOR of rows products for ``Rider``: it
    yields None after every product it has emitted and 'CTX' -- without emitting -- in front of a product that needs the GCN
    output: ``box['ctx']`` (concat conditioning: the very first product) or ``box['cavo']`` (crossattn: the attention product of
    the first transformer block), both filled in by emit_unet1d_step once the GCN chain is emitted."""
    mc = w.mc
    # Every trunk product writes a slab tensor (K split over workgroups) unless its consumer needs whole rows cheaply:
    # the two LayerNorm operands of a transformer block (t0, t2) are produced with ``ln_split`` slices.
    ln_split = ROWS_LN_SPLIT

    def ln_kbps(K):
        nkb = (K + 15) // 16
        return False if ln_split <= 1 else max(8, (nkb + ln_split - 1) // ln_split)

    def run_block(name_prefix, blk, h_segs, hC):
        """h_segs: list of Views forming the (possibly concatenated) input; returns (Views, C)."""
        for j, it in enumerate(blk):
            name = f'{name_prefix}.{j}'
            d = w.items[name]
            kind = it[0]
            if kind == 'conv_in':
                o = b.linear([seg(v) for v in h_segs], d['conv'], O)
                yield
                h_segs, hC = [o], mc
            elif kind == 'res':
                cin, cout = it[1], it[2]
                assert cin == hC
                eo, _ = w.emb_slices[name]
                skip_early = 'skip' in d and ROWS_SKIP_EARLY and b.allow_split      # (not behind a separate GroupNorm launch)
                if skip_early:
                    yield 'CTX'                  # (this variant fuses conv1 with the skip projection itself: nothing rides from here on)
                h1 = b.linear(norm_segs(h_segs, d['gn1'][0], d['gn1'][1], 1e-5, True, C=cin, b=b, M=O), d['conv1'], O,
                              res=(View(emb_all[0], col=eo, ld=0, width=cout, step=emb_all[1], step_stride=emb_all[0].shape[1])
                                   if isinstance(emb_all, tuple) else View(emb_all, col=eo, ld=emb_ld, width=cout)), fuse_next=skip_early)
                if not skip_early:
                    yield
                gn2 = norm_segs([h1], d['gn2'][0], d['gn2'][1], 1e-5, True, C=cout, b=b, M=O)
                if skip_early:
                    # skip_connection(x) depends only on the block input: it rides on conv1's launch as a second, independent problem
                    # (one grid), and conv2 shrinks from K = cout + cin to K = cout with the projection as its (slab) residual
                    resv = b.linear([seg(v) for v in h_segs], d['skip'], O)
                    o = b.linear(gn2, d['conv2'], O, res=resv)
                elif 'skip' in d and len(h_segs) <= 2:
                    # out = conv2(silu(GN2(h1))) + skip(x): ONE op over the K-concatenation [h1 | x] with weights [W2 | Wskip],
                    # the norm being the prologue of the first segment only
                    o = b.linear(gn2 + [seg(v) for v in h_segs], d['conv2skip'], O)
                else:
                    if 'skip' in d:
                        resv = b.linear([seg(v) for v in h_segs], d['skip'], O)
                        yield
                    else:
                        assert len(h_segs) == 1
                        resv = h_segs[0]
                    o = b.linear(gn2, d['conv2'], O, res=resv)
                yield
                h_segs, hC = [o], cout
            elif kind == 'attn' and w.concat:
                C = it[1]
                xin = h_segs[0]
                o = b.linear(norm_segs([xin], d['gn'][0], d['gn'][1], 1e-5, False, C=C, b=b, M=O), d['av'], O, res=xin)
                yield
                h_segs, hC = [o], C
            elif kind == 'attn':
                C = it[1]
                xin = h_segs[0]
                fold = ROWS_FOLD_ATTN1 and 'proj_in_u' in d and b.allow_split
                t0 = b.linear(norm_segs([xin], d['gn'][0], d['gn'][1], 1e-6, False, C=C, b=b, M=O), d['proj_in_u' if fold else 'proj_in'], O,
                              split=ln_kbps(C))
                if fold:
                    t0 = t0.cols(0, C)               # (u = t0 W1^T sits C columns further in the same slabs)
                yield
                yield 'CTX'                      # the next product adds the cross-attention vector of the GCN output
                cavo = box['cavo']
                if fold:
                    # the feed-forward launch forms t2 = attn1(norm1(t0)) + t0 + attn2 in its prologue and publishes it (ES_PRO_LN_ATTN)
                    t2 = View(b.buf(O, C, scratch=True))
                    n_ops, acct = len(b.ops), (b.weight_bytes, b.flops)
                    gl = b.linear([seg(t0, pro=hip.PRO_LN_ATTN, eps=1e-5, gs=C)], d['ff1'], O, res=t2, res2=cavo[name])
                    if hip.lib().es_linear_rows_takes_ln_attn(_byref(b.ops[-1].u.linear)) != 1:
                        del b.ops[n_ops:]            # (no kernel for this shape: the self-attention product stays a launch of its own)
                        b.weight_bytes, b.flops = acct
                        fold = False
                if fold:
                    yield
                    b.tags[name + '.transformer_blocks.0:in'] = t0
                    b.tags[name + '.transformer_blocks.0:attn2'] = t2
                    o = b.linear([seg(gl), seg(t2)], d['ff2po'], O, res=xin)
                    yield
                    h_segs, hC = [o], C
                    b.tags[name] = h_segs[0]
                    continue
                # x = attn1(norm1(x)) + x ; x = attn2(norm2(x), ctx) + x.  With one token attn1 is the folded matrix
                # to_out.to_v applied to LN1(x); with one key the second line adds the per-node vector
                # to_out2(to_v2(ctx)) (precomputed above) -> second residual of the same op.
                vs_ = ROWS_VO1_SLICES
                t2 = b.linear([seg(t0, pro=hip.PRO_LN, gamma=d['ln1'][0], beta=d['ln1'][1], eps=1e-5, gs=C)], d['vo1'], O,
                              res=t0, res2=cavo[name], use_bias='a1_bias_in_cav' not in d,      # (else: the bias already sits in cavo)
                              split=(False if vs_ <= 1 else max(8, ((C + 15) // 16 + vs_ - 1) // vs_)))
                yield
                b.tags[name + '.transformer_blocks.0:in'] = t0
                b.tags[name + '.transformer_blocks.0:attn2'] = t2
                # GEGLU applied in the ff1 epilogue (needs finished sums: one slice, 16 * 4C / 16 column tiles)
                gl = b.linear([seg(t2, pro=hip.PRO_LN, gamma=d['ln3'][0], beta=d['ln3'][1], eps=1e-5, gs=C)], d['ff1'], O)
                yield
                o = b.linear([seg(gl), seg(t2)], d['ff2po'], O, res=xin)
                yield
                h_segs, hC = [o], C
            elif kind in ('down', 'up'):
                o = b.linear([seg(v) for v in h_segs], d['conv'], O)
                yield
                h_segs = [o]
            b.tags[name] = h_segs[0]
        return h_segs, hC

    inp, mid, out = w.topo
    hs = []
    h_segs, hC = [View(x)], w.in_ch
    if w.concat:                               # box vector and GCN output as input channels (denoise_net.py:789-790)
        yield 'CTX'
        h_segs, hC = [View(x), box['ctx']], w.in_ch + w.ctx_dim
    for i, blk in enumerate(inp):
        h_segs, hC = yield from run_block(f'input_blocks.{i}', blk, h_segs, hC)
        hs.append((h_segs[0], hC))
    h_segs, hC = yield from run_block('middle_block', mid, h_segs, hC)
    for i, blk in enumerate(out):
        sk, sC = hs.pop()
        h_segs, hC = yield from run_block(f'output_blocks.{i}', blk, [h_segs[0], sk], hC + sC)
    # (eps stays an ordinary tensor: it is also the result of the step-level API; one slice)
    # eps: a slab tensor too when the caller takes it as a View (eps_out None: the DDPM update sums the slabs) -- two workgroups
    # multiplying the whole K range were the slowest launch of the trunk's tail
    if eps_out is None:
        b.tags['eps'] = b.linear(norm_segs(h_segs, w.out_gn[0], w.out_gn[1], 1e-5, True, C=hC, b=b, M=O), w.out_conv, O)
    else:
        b.linear(norm_segs(h_segs, w.out_gn[0], w.out_gn[1], 1e-5, True, C=hC, b=b, M=O), w.out_conv, O, View(eps_out))
        b.tags['eps'] = View(eps_out)
    yield


# the volume-path ops (conv, groupnorm, attention, ...) are attached to Builder by plan_vol; importing it here makes
# `from echoscene_amd.plan import Builder` complete on its own (a test that ran alone found Builder without them)
from . import plan_vol  # noqa: E402,F401
