"""Plan compiler for the 3-D latent-SDF denoiser (reference UNet3DModel.forward,
model/networks/diffusion_shape/openai_model_3d.py:816-863, block semantics :294-314 and
attention.py:172-245, 335-351).

Per-object vectors (time MLP, conv-pool stem code, shape GCN, ResBlock time projections,
cross-attention-with-one-key vectors) run on the fp32 "rows" path; everything with a voxel
dimension runs on the "volume" path (channels-last, fp16 MFMA operands, fp32 accumulate and
fp32 residual stream).
"""
import ctypes as C
import os
import torch

from . import hip
from .hip import ConvArgs, GNArgs, LNArgs, AttnArgs, GegluArgs, ToClArgs, StemArgs, Op
from .plan import (Builder, PackedLinear, GCNWeights, View, seg, emit_gcn, own, mm64)


# Planner constants of the volume path (round 5: module constants, not environment switches -- they decide which products a plan
# holds and which tensors exist, i.e. the bits; tests / A-B tools set the attributes).
VOL_FOLD_FFO = True        # FeedForward output + proj_out as ONE K-concatenated product (round 4: -11 launches per step)
VOL_GN_F16 = True          # a ResBlock's conv1 -> GroupNorm -> conv2 intermediate written once, as f16, with the conv's row-group sums
VOL_GN_PART_FUSED = True     # a split-K conv's reduction kernel also forms the next GroupNorm's partial sums (same bits, one launch less)
VOL_GN_RG_ANY = False      # tests: ask every producing conv for the row-group sums, whatever route it takes


def _gn_rg():
    """the library's route option `gn_rg` (es_vol_set_option): GroupNorm statistics from the producing conv's row-group sums"""
    buf = C.create_string_buffer(512)
    hip.lib().es_vol_options(buf, 512)
    opts = dict(kv.split('=') for kv in buf.value.decode().strip(';').split(';'))
    return int(opts.get('gn_rg', '1')) != 0


class PackedConv:
    """f16 [Npad][taps][Cin32] image of a conv / linear weight + fp32 bias on the device."""

    def __init__(self, W, b, device, geglu=False):
        on_gpu = torch.device(device).type == 'cuda'
        W = W.detach().float().contiguous()
        W = W.to(device) if on_gpu else W.cpu()
        self.geglu = False
        if geglu and W.dim() == 2 and (W.shape[0] // 2) % 112 == 0 and b is not None:
            # GEGLU fused into the contraction epilogue (ES_EPI_GEGLU): every 16-row group of the packed weight = the value rows
            # of 8 outputs, then their 8 gate rows (value = first half of the projection, attention.py:39-46), so the value and
            # the gate of an output land in lanes i16 and i16 ^ 8 of one MFMA tile (echoscene_hip.h: es_conv_args.epilogue)
            C4 = W.shape[0] // 2
            idx = (torch.arange(C4).view(-1, 1, 8) + torch.tensor([0, C4]).view(1, 2, 1)).reshape(-1)
            W = W.index_select(0, idx.to(W.device)).contiguous()
            b = b.detach().float().cpu()[idx].contiguous()
            self.geglu = True
        self.N, cin = W.shape[0], W.shape[1]
        self.taps = 1 if W.dim() == 2 else int(W[0, 0].numel())
        if self.taps not in (1, 27):
            raise ValueError('conv kernel must be 1x1x1 or 3x3x3')
        self.Cin = (cin + 31) // 32 * 32
        L = hip.lib()
        if self.N <= 4 and self.taps == 27 and self.Cin <= 64:   # consumed by the small-N kernels (VQ-VAE conv_out)
            W = W.cpu()
            out = torch.empty(self.N * self.taps * self.Cin, dtype=torch.int16)
            hip.check(L.es_pack_conv_rows_f16(C.c_void_p(W.data_ptr()), self.N, cin, self.taps,
                                              C.c_void_p(out.data_ptr())), 'es_pack_conv_rows_f16')
        elif on_gpu:
            # the tiled image is formed on the GPU from the uploaded fp32 weight (es_pack_conv_f16_dev: bit-identical to the host loop,
            # which was 6 of the 9 s of a process's first scene call)
            with torch.cuda.device(device):
                out = torch.empty(L.es_pack_conv_f16_size(self.N, self.Cin, self.taps), dtype=torch.int16, device=device)
                hip.check(L.es_pack_conv_f16_dev(hip.ptr(W), self.N, cin, self.taps, hip.ptr(out), hip.current_stream()), 'es_pack_conv_f16_dev')
        else:
            n = L.es_pack_conv_f16_size(self.N, self.Cin, self.taps)
            out = torch.empty(n, dtype=torch.int16)
            hip.check(L.es_pack_conv_f16(C.c_void_p(W.data_ptr()), self.N, cin, self.taps, C.c_void_p(out.data_ptr())),
                      'es_pack_conv_f16')
        self.w = out.to(device)
        self.b = None if b is None else own(b, device)
        self.weight_bytes = self.N * cin * self.taps * 2
        self.cin_true = cin


class PackedConv32:
    """fp32 [N][taps][Cin16] image of a conv / linear weight (the fp32-operand validation route, csrc/es_vol32.hip).  A GEGLU projection
    keeps its value | gate row order (the fp32 route applies GEGLU with its own kernel)."""

    def __init__(self, W, b, device, geglu=False):
        W = W.detach().float().contiguous().cpu()
        self.geglu = False
        self.N, cin = W.shape[0], W.shape[1]
        self.taps = 1 if W.dim() == 2 else int(W[0, 0].numel())
        if self.taps not in (1, 27):
            raise ValueError('conv kernel must be 1x1x1 or 3x3x3')
        self.Cin = (cin + 15) // 16 * 16
        L = hip.lib()
        out = torch.empty(L.es_pack_conv_f32_size(self.N, cin, self.taps), dtype=torch.float32)
        hip.check(L.es_pack_conv_f32(C.c_void_p(W.data_ptr()), self.N, cin, self.taps, C.c_void_p(out.data_ptr())), 'es_pack_conv_f32')
        self.w = out.to(device)
        self.b = None if b is None else own(b, device)
        self.weight_bytes = self.N * cin * self.taps * 4
        self.cin_true = cin


class PackedConvX3(PackedConv):
    """Split-operand weight image (round 6, precision 'fp32x'): w = w_hi + w_lo in f16 (22 bits of the mantissa), packed as ONE f16
    image over 3 Cin channels [w_hi | w_hi | w_lo] -- against the activation image [a_hi | a_lo | a_hi] (es_split_f16x3) the ordinary
    f16 contraction accumulates a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32.  ``cin_part``: channels of one part (Cin of the fp32
    operand, padded to 32); GEGLU is applied by its own kernel, as on the fp32 route."""

    def __init__(self, W, b, device, geglu=False):
        W = W.detach().float()
        cin = W.shape[1]
        cp = (cin + 31) // 32 * 32
        if cp != cin:
            pad = torch.zeros((W.shape[0], cp - cin) + tuple(W.shape[2:]), dtype=W.dtype, device=W.device)
            W = torch.cat([W, pad], 1)
        hi = W.half().float()
        lo = (W - hi).half().float()
        super().__init__(torch.cat([hi, hi, lo], 1), b, device, geglu=False)
        self.cin_part, self.cin_true = cp, cin
        self.weight_bytes = self.N * cin * self.taps * 6


class UNet3DWeights:
    def __init__(self, sd, net, device, precision='fp16'):
        """sd: state_dict of the UNet3DModel holder ``net`` (keys without 'diffusion_net.').  ``precision='fp32'``: fp32 weight
        images for the fp32-operand validation route."""
        self.precision = precision
        PackedConv = globals()[{'fp32': 'PackedConv32', 'fp32x': 'PackedConvX3'}.get(precision, 'PackedConv')]
        self.device, self.mc, self.topo = device, net.model_channels, net.topo
        self.enable_t_emb, self.mp = net.enable_t_emb, net.messsage_passing
        self.heads = net.num_heads
        self.concat = bool(getattr(net, 'concat', False))
        dv = lambda k: own(sd[k], device)
        PL = lambda w, b: PackedLinear(sd[w], sd[b] if b else None, device)
        PC = lambda w, b: PackedConv(sd[w], sd[b] if b else None, device)
        self.te0 = PL('time_embed.0.weight', 'time_embed.0.bias')
        self.te2 = PL('time_embed.2.weight', 'time_embed.2.bias')
        self.shape_t = None
        if self.mp:
            self.stem = [dv('shape_embeddings.0.weight'), dv('shape_embeddings.0.bias'),
                         dv('shape_embeddings.2.weight'), dv('shape_embeddings.2.bias')]
            self.stem_lin = PL('shape_embeddings.5.weight', 'shape_embeddings.5.bias')
            self.shape_t = PL('shape_time_emb.weight', 'shape_time_emb.bias') if net.enable_t_emb else None
            self.pred_table = sd['pred_embeddings.weight'].detach().float().cpu()
            self.gcn = GCNWeights(sd, 'shape_code_graph_cov', device)
        inp, mid, out = net.topo
        names = [(f'input_blocks.{i}.{j}', it) for i, blk in enumerate(inp) for j, it in enumerate(blk)]
        names += [(f'middle_block.{j}', it) for j, it in enumerate(mid)]
        names += [(f'output_blocks.{i}.{j}', it) for i, blk in enumerate(out) for j, it in enumerate(blk)]
        self.items, self.emb_slices, self.ca = {}, {}, {}
        emb_w, emb_b, ca_v, ca_b, off = [], [], [], [], 0
        for name, it in names:
            kind, d = it[0], {}
            if kind == 'conv_in':
                d['conv'] = PC(name + '.weight', name + '.bias')
            elif kind == 'res':
                d['gn1'] = (dv(name + '.in_layers.0.weight'), dv(name + '.in_layers.0.bias'))
                d['conv1'] = PC(name + '.in_layers.2.weight', name + '.in_layers.2.bias')
                d['gn2'] = (dv(name + '.out_layers.0.weight'), dv(name + '.out_layers.0.bias'))
                d['conv2'] = PC(name + '.out_layers.3.weight', name + '.out_layers.3.bias')
                if (name + '.skip_connection.weight') in sd:
                    d['skip'] = PackedConv(sd[name + '.skip_connection.weight'].flatten(1), None, device)
                    # both biases are added once in the fused epilogue
                    d['bias2'] = (sd[name + '.out_layers.3.bias'].float() +
                                  sd[name + '.skip_connection.bias'].float()).contiguous().to(device)
                emb_w.append(sd[name + '.emb_layers.1.weight'])
                emb_b.append(sd[name + '.emb_layers.1.bias'])
                self.emb_slices[name] = (off, it[2])
                off += it[2]
            elif kind == 'attn' and self.concat:
                # AttentionBlock (openai_model_3d.py:317-363).  QKVAttentionLegacy orders the qkv rows
                # [head][q|k|v][ch]; the attention kernel wants [q | k | v] with heads inside each: row permutation.
                Cc = it[1]
                ch = Cc // self.heads
                perm = torch.cat([torch.cat([torch.arange(h * 3 * ch + part * ch, h * 3 * ch + (part + 1) * ch)
                                             for h in range(self.heads)]) for part in range(3)])
                d['gn'] = (dv(name + '.norm.weight'), dv(name + '.norm.bias'))
                d['qkv'] = PackedConv(sd[name + '.qkv.weight'].flatten(1)[perm], sd[name + '.qkv.bias'][perm], device)
                d['proj_out'] = PackedConv(sd[name + '.proj_out.weight'].flatten(1), sd[name + '.proj_out.bias'], device)
            elif kind == 'attn':
                tb = name + '.transformer_blocks.0'
                d['gn'] = (dv(name + '.norm.weight'), dv(name + '.norm.bias'))
                d['proj_in'] = PackedConv(sd[name + '.proj_in.weight'].flatten(1), sd[name + '.proj_in.bias'], device)
                d['ln1'] = (dv(tb + '.norm1.weight'), dv(tb + '.norm1.bias'))
                d['ln3'] = (dv(tb + '.norm3.weight'), dv(tb + '.norm3.bias'))
                d['qkv'] = PackedConv(torch.cat([sd[tb + '.attn1.to_q.weight'], sd[tb + '.attn1.to_k.weight'],
                                                 sd[tb + '.attn1.to_v.weight']], 0), None, device)
                d['o1'] = PC(tb + '.attn1.to_out.0.weight', tb + '.attn1.to_out.0.bias')
                d['o2'] = PL(tb + '.attn2.to_out.0.weight', tb + '.attn2.to_out.0.bias')     # rows path
                d['ff1'] = PackedConv(sd[tb + '.ff.net.0.proj.weight'], sd[tb + '.ff.net.0.proj.bias'], device, geglu=True)
                d['ff2'] = PC(tb + '.ff.net.2.weight', tb + '.ff.net.2.bias')
                d['proj_out'] = PackedConv(sd[name + '.proj_out.weight'].flatten(1), sd[name + '.proj_out.bias'], device)
                # FeedForward output + proj_out as ONE K-concatenated product (round 4): with t3 = ff2(gg) + t2 (attention.py:243-245)
                # and out = proj_out(t3) + x_in (:385-396),  out = (Wp W2) gg + Wp t2 + (Wp b2 + bp) + x_in -- the kernel's second
                # contraction phase (a2 / w2) carries the t2 term; the [M, C] tensor t3 and one HBM-bound launch per block disappear
                if VOL_FOLD_FFO:
                    W2, b2 = sd[tb + '.ff.net.2.weight'].double(), sd[tb + '.ff.net.2.bias'].double()
                    Wp, bp = sd[name + '.proj_out.weight'].flatten(1).double(), sd[name + '.proj_out.bias'].double()
                    d['ffo'] = PackedConv(mm64(Wp, W2).float(), None, device)
                    d['po'] = PackedConv(Wp.float(), None, device)
                    d['ffo_bias'] = (mm64(Wp, b2) + bp).float().contiguous().to(device)
                # cross-attention with one key: to_out2(to_v2(ctx)) folded into one matrix per block (fp64)
                self.ca[name] = (len(ca_v), it[1])
                ca_v.append(mm64(sd[tb + '.attn2.to_out.0.weight'], sd[tb + '.attn2.to_v.weight']).float())
                ca_b.append(sd[tb + '.attn2.to_out.0.bias'])
            elif kind == 'down':
                d['conv'] = PC(name + '.op.weight', name + '.op.bias')
            elif kind == 'up':
                d['conv'] = PC(name + '.conv.weight', name + '.conv.bias')
            self.items[name] = d
        self.emb_all = PackedLinear(torch.cat(emb_w, 0), torch.cat(emb_b, 0), device)
        self.cav_all = None if self.concat else PackedLinear(torch.cat(ca_v, 0), torch.cat(ca_b, 0), device)
        self.out_gn = (dv('out.0.weight'), dv('out.0.bias'))
        self.out_conv = PC('out.2.weight', 'out.2.bias')
        self.in_ch, self.out_ch = net.in_channels, net.out_channels


class VolBuilderMixin:
    """Volume-path op emitters, mixed into plan.Builder."""

    def _push(self, kind, field, a):
        op = Op()
        op.kind, op.lane = kind, 0
        setattr(op.u, field, a)
        self.ops.append(op)
        return len(self.ops) - 1

    def conv(self, a_f16, pc, O, dims, mode=hip.CONV_SAME, bias=None, rowvec=None, res=None, out_f32=None,
             out_f16=None, skip=None, ncdhw=False, splitk=None, epilogue=0, out_ld=None):
        """dims = (D,H,W) of the OUTPUT grid. skip = (raw_f16 tensor, PackedConv) for the fused 1x1 skip."""
        D, H, W = dims
        if getattr(self, 'fp32x', False) and a_f16.dtype == torch.float32:
            # split-operand route: the fp32 operand(s) -> [hi | lo | hi] f16 images, then the ordinary f16 launch on 3 Cin channels; an
            # f16 operand copy of the output is the fp32 output itself, as on the fp32 route
            assert isinstance(pc, PackedConvX3) and not epilogue
            a3 = self.split3(a_f16, pc.cin_part)
            if skip is not None:
                skip = (self.split3(skip[0], skip[1].cin_part), skip[1])
            out = out_f32 if out_f32 is not None else out_f16
            assert out is not None and out.dtype == torch.float32 and (out_f32 is None or out_f16 is None)
            return self.conv(a3, pc, O, dims, mode=mode, bias=bias, rowvec=rowvec, res=res, out_f32=out, skip=skip, ncdhw=ncdhw,
                             splitk=splitk, out_ld=out_ld)
        if getattr(self, 'fp32', False) and not getattr(self, 'fp32x', False):
            return self._conv32(a_f16, pc, O, dims, mode, bias, rowvec, res, out_f32, out_f16, skip, ncdhw, out_ld)
        a = ConvArgs()
        a.a, a.w = a_f16.data_ptr(), pc.w.data_ptr()
        a.O, a.D, a.H, a.W = O, D, H, W
        a.Cin, a.N, a.taps, a.mode = pc.Cin, pc.N, pc.taps, mode
        if skip is not None:
            a.a2, a.w2, a.Cin2 = skip[0].data_ptr(), skip[1].w.data_ptr(), skip[1].Cin
            self.weight_bytes += skip[1].weight_bytes
            self.flops += 2 * O * D * H * W * skip[1].Cin * pc.N
        bt = pc.b if bias is None else bias
        a.bias = bt.data_ptr() if bt is not None else None
        if rowvec is not None:
            a.rowvec, a.rowvec_ld = rowvec.ptr, rowvec.ld
        a.res = res.data_ptr() if res is not None else None
        a.out_f32 = out_f32.data_ptr() if out_f32 is not None else None
        a.out_f16 = out_f16.data_ptr() if out_f16 is not None else None
        a.out_ld = -1 if ncdhw else (pc.N if out_ld is None else out_ld)
        a.epilogue = epilogue
        a.O_hint = int(getattr(self, 'o_hint', 0) or 0)      # object sharding: tile / split choices of the whole problem (> 0) / of the reference shard (< 0)
        # split-K scratch shared by all convs of the plan (ops are stream-ordered): the slabs of the launch that splits widest
        M = O * D * H * W
        # eligibility from the WHOLE (or reference) problem's row count (O_hint): a shard and the unsharded run must take the same
        # can_split decision, or the library picks S from the global tile count for one and S = 1 for the other
        Mh = (-a.O_hint if a.O_hint < 0 else max(a.O_hint, O)) * D * H * W
        if not ncdhw and Mh * pc.N <= 8192 * 5376 and pc.N % 4 == 0 and not epilogue:
            a.splitk = -1 if splitk is None else splitk
            a.workspace = 1                                  # (any non-NULL value: es_conv_split_of only asks whether there is one)
            S = hip.lib().es_conv_split_of(C.byref(a))       # round 6: the library says how many slabs it will write
            if S < 0:
                raise RuntimeError('es_conv_split_of: ' + hip.lib().es_last_error().decode())
            need = max(S, splitk or 0, 1) * M * pc.N
            if getattr(self, '_ws', None) is None or self._ws.numel() < need:
                self._ws = self.buf(need, scratch=True)
                for op in self.ops:
                    if op.kind == hip.OP_CONV and op.u.conv.workspace:
                        op.u.conv.workspace = self._ws.data_ptr()
            a.workspace = self._ws.data_ptr()
        self.keep += [pc, bt, skip]
        self.weight_bytes += pc.weight_bytes
        self.flops += 2 * O * D * H * W * pc.cin_true * pc.taps * pc.N
        idx = self._push(hip.OP_CONV, 'conv', a)
        if out_f32 is not None and not ncdhw and not epilogue and (out_ld is None or out_ld == pc.N):
            # a GroupNorm that reads this tensor later asks the conv for its row-group sums (groupnorm() below)
            if not hasattr(self, '_conv_of'):
                self._conv_of = {}
            self._conv_of[out_f32.data_ptr()] = (idx, M, pc.N)
        return idx

    def split3(self, x32, C):
        """f16 [rows, 3 C] split-operand image of the fp32 tensor x32 [rows, C] (es_split_f16x3), as one op of the plan"""
        rows = x32.numel() // C
        out = self.buf(rows, 3 * C, dtype=torch.float16, scratch=True)
        a = ToClArgs()
        a.x, a.O, a.C, a.V, a.Cpad, a.out, a.out_is_f32 = x32.data_ptr(), rows, C, 1, C, out.data_ptr(), 2
        self._push(hip.OP_TO_CL, 'tocl', a)
        return out

    def _conv32(self, a32, pc, O, dims, mode, bias, rowvec, res, out_f32, out_f16, skip, ncdhw, out_ld):
        """fp32-operand validation route: the same launch on es_conv_f32 (fp32 activations / weights, exact-fp32 MFMA).  A request
        for an f16 operand copy (``out_f16``) is served by the fp32 output itself: in this mode every "f16" buffer of the plan is an
        fp32 tensor."""
        D, H, W = dims
        assert isinstance(pc, PackedConv32) and a32.dtype == torch.float32
        a = ConvArgs()
        a.a, a.w = a32.data_ptr(), pc.w.data_ptr()
        a.O, a.D, a.H, a.W = O, D, H, W
        a.Cin, a.N, a.taps, a.mode = pc.Cin, pc.N, pc.taps, mode
        if skip is not None:
            a.a2, a.w2, a.Cin2 = skip[0].data_ptr(), skip[1].w.data_ptr(), skip[1].Cin
            self.flops += 2 * O * D * H * W * skip[1].Cin * pc.N
        bt = pc.b if bias is None else bias
        a.bias = bt.data_ptr() if bt is not None else None
        if rowvec is not None:
            a.rowvec, a.rowvec_ld = rowvec.ptr, rowvec.ld
        a.res = res.data_ptr() if res is not None else None
        out = out_f32 if out_f32 is not None else out_f16
        assert out is not None and out.dtype == torch.float32 and (out_f32 is None or out_f16 is None)
        a.out_f32 = out.data_ptr()
        a.out_ld = -1 if ncdhw else (pc.N if out_ld is None else out_ld)
        self.keep += [pc, bt, skip]
        self.weight_bytes += pc.weight_bytes
        self.flops += 2 * O * D * H * W * pc.cin_true * pc.taps * pc.N
        return self._push(hip.OP_CONV_F32, 'conv', a)

    def _rowgroup_producer(self, x, Cx, M, V):
        """The conv op of this plan that writes the fp32 tensor x and would form its row-group sums in its own epilogue
        (es_conv_emits_gn_stats), or None: x comes from elsewhere (stem output, another plan), the shapes do not allow it, or the
        conv takes a route behind which the sums would cost a pass over the output -- what the GroupNorm's statistics pass costs."""
        if not _gn_rg() or V % 64 or x is None:
            return None
        ent = getattr(self, '_conv_of', {}).get(x.data_ptr())
        if ent is None or ent[1] != M or ent[2] != Cx:
            return None
        op = self.ops[ent[0]]
        if op.kind != hip.OP_CONV or op.u.conv.out_f32 != x.data_ptr():
            return None
        # (VOL_GN_RG_ANY, tests: ask whatever the route -- the sums then come from k_rowgroup_stats behind the conv)
        if not VOL_GN_RG_ANY:
            q = ConvArgs.from_buffer_copy(op.u.conv)
            if q.O_hint > q.O or q.O_hint < 0:
                # a deterministic shard takes the decision of the WHOLE problem (round 6: of the reference shard, O_hint < 0):
                # where that run reduces row-group sums every run must too (its own launch may take a route that cannot form them
                # -- k_rowgroup_stats then leaves the same bits), or two runs would normalise with differently rounded statistics
                q.O, q.O_hint = abs(q.O_hint), 0
            if hip.lib().es_conv_emits_gn_stats(C.byref(q)) != 1:
                return None
        return op

    def _rowgroup_stats(self, op, x, Cx, M):
        """Row-group sums [2][M/64][Cx] of x, requested from its producer op (es_conv_args.gn_stats_out)."""
        st = getattr(self, '_rg_stats', None)
        if st is None:
            st = self._rg_stats = {}
        ent = st.get(x.data_ptr())
        if ent is None or ent[0] is not op:
            # keyed by (producer op, tensor): a buffer that a LATER conv rewrites gets fresh sums from that conv, never the stale
            # ones of the first producer
            buf = self.buf(2 * (M // 64) * Cx, scratch=True)
            op.u.conv.gn_stats_out = buf.data_ptr()
            st[x.data_ptr()] = ent = (op, buf)
        assert op.u.conv.gn_stats_out == ent[1].data_ptr() and op.u.conv.out_f32 == x.data_ptr()
        return ent[1]

    def conv_gn_intermediate(self, a_f16, pc, O, dims, rowvec=None):
        """A conv whose output is read ONLY by a GroupNorm -- the conv1 -> GroupNorm -> conv2 intermediate of a ResBlock
        (openai_model_3d.py:294-314, vqvae_modules.py:67-126), never part of the residual stream.  It is written ONCE, as f16 (the
        operand precision the GroupNorm emits anyway), with the GroupNorm's statistics formed from the conv's fp32 values before the
        rounding (es_conv_args.gn_stats_out): 6 B per element less HBM traffic than fp32 out + fp32 in.
        Only where the launch forms those sums itself (es_conv_emits_gn_stats); a deterministic shard follows the decision of
        the WHOLE problem and, when its own route cannot form them, writes fp32 as well (the sums then come from a pass over it:
        same order, same bits; the GroupNorm still reads the f16 tensor).  Returns the tensor the GroupNorm is to read."""
        D, H, W = dims
        M, V = O * D * H * W, D * H * W
        if getattr(self, 'fp32', False):
            h32 = self.buf(M, pc.N, scratch=True)
            self.conv(a_f16, pc, O, dims, rowvec=rowvec, out_f32=h32)
            return h32
        h16 = self.buf(M, pc.N, dtype=torch.float16, scratch=True)
        idx = self.conv(a_f16, pc, O, dims, rowvec=rowvec, out_f16=h16)
        op = self.ops[idx]
        L = hip.lib()
        if V % 64 == 0 and _gn_rg() and VOL_GN_F16:
            whole = ConvArgs.from_buffer_copy(op.u.conv)
            if whole.O_hint > whole.O or whole.O_hint < 0:
                whole.O, whole.O_hint = abs(whole.O_hint), 0
            if L.es_conv_emits_gn_stats(C.byref(whole)) == 1:
                st = self.buf(2 * (M // 64) * pc.N, scratch=True)
                op.u.conv.gn_stats_out = st.data_ptr()
                if L.es_conv_emits_gn_stats(C.byref(op.u.conv)) != 1:
                    h32 = self.buf(M, pc.N, scratch=True)
                    op.u.conv.out_f32 = h32.data_ptr()
                    self.keep.append(h32)
                if not hasattr(self, '_f16_stats'):
                    self._f16_stats = {}
                self._f16_stats[h16.data_ptr()] = st
                return h16
        h32 = self.buf(M, pc.N, scratch=True)
        op.u.conv.out_f32, op.u.conv.out_f16 = h32.data_ptr(), None
        # the f16 tensor is not used on this route: release it (it was M x N x 2 bytes of HBM held for nothing)
        self.scratch.discard(h16.data_ptr())
        self.keep[:] = [t for t in self.keep if t is not h16]
        del h16
        if not hasattr(self, '_conv_of'):
            self._conv_of = {}
        self._conv_of[h32.data_ptr()] = (idx, M, pc.N)
        return h32

    def groupnorm(self, x1, C1, x2, C2, O, V, gamma, beta, eps, silu, y_f16, raw_f16=None, groups=32):
        a = GNArgs()
        a.x1, a.C1 = x1.data_ptr(), C1
        a.x2, a.C2 = (x2.data_ptr(), C2) if x2 is not None else (None, 0)
        a.O, a.V, a.groups, a.eps = O, V, groups, eps
        a.gamma, a.beta, a.silu = gamma.data_ptr(), beta.data_ptr(), int(silu)
        need = O * ((V + 7) // 8) * groups * 2 + O * groups * 2     # es_groupnorm_vol: partials at the smallest voxel tile (8) + final stats
        st = getattr(self, '_gn_stats', None)           # one scratch shared by all GroupNorms (same stream, in order)
        if st is None or st.numel() < need:
            st = self._gn_stats = self.buf(need, scratch=True)
        a.stats = st.data_ptr()
        a.y_f16 = y_f16.data_ptr()
        a.raw_f16 = raw_f16.data_ptr() if raw_f16 is not None else None
        a.O_hint = int(getattr(self, 'o_hint', 0) or 0)
        if getattr(self, 'fp32', False):            # fp32-operand validation route: fp32 operand out, statistics by a pass over x
            assert y_f16.dtype == torch.float32 and (raw_f16 is None or raw_f16.dtype == torch.float32)
            a.y_is_f32 = 1
            self.keep += [gamma, beta]
            return self._push(hip.OP_GN, 'gn', a)
        # statistics from the producing convs' epilogues instead of a pass over x1 / x2 (both sources must have them)
        if x1.dtype == torch.float16:               # the f16-only output of conv_gn_intermediate(): statistics from that conv's sums
            assert x2 is None and raw_f16 is None
            a.x1_is_f16 = 1
            a.stats1 = self._f16_stats[x1.data_ptr()].data_ptr()
            self.keep += [gamma, beta]
            return self._push(hip.OP_GN, 'gn', a)
        p1 = self._rowgroup_producer(x1, C1, O * V, V)
        p2 = self._rowgroup_producer(x2, C2, O * V, V) if x2 is not None else None
        if p1 is not None and (x2 is None or p2 is not None):
            a.stats1 = self._rowgroup_stats(p1, x1, C1, O * V).data_ptr()
            a.stats2 = self._rowgroup_stats(p2, x2, C2, O * V).data_ptr() if x2 is not None else None
        elif x2 is None and VOL_GN_PART_FUSED:
            # x1 written by a split-K conv of this plan: its reduction kernel leaves this GroupNorm's per-tile partial sums
            # (es_conv_args.gn_part_out; k_gn_partial's own order, the same bits) and the statistics launch is dropped
            ent = getattr(self, '_conv_of', {}).get(x1.data_ptr())
            if ent is not None and ent[1] == O * V and ent[2] == C1:
                op = self.ops[ent[0]]
                cv = op.u.conv
                if op.kind == hip.OP_CONV and cv.out_f32 == x1.data_ptr() and not cv.gn_part_out and not cv.gn_stats_out:
                    cv.gn_part_groups = groups
                    if hip.lib().es_conv_emits_gn_part(C.byref(cv)) == 1:
                        pb = self.buf(O * ((V + 7) // 8) * groups * 2, scratch=True)
                        cv.gn_part_out = pb.data_ptr()
                        a.part_in = pb.data_ptr()
                        self.keep.append(pb)
                    else:
                        cv.gn_part_groups = 0
        self.keep += [gamma, beta]
        return self._push(hip.OP_GN, 'gn', a)

    def layernorm(self, x, M, Cc, gamma, beta, y_f16):
        a = LNArgs()
        a.x, a.M, a.C, a.eps = x.data_ptr(), M, Cc, 1e-5
        a.gamma, a.beta, a.y_f16 = gamma.data_ptr(), beta.data_ptr(), y_f16.data_ptr()
        a.y_is_f32 = 1 if y_f16.dtype == torch.float32 else 0
        self.keep += [gamma, beta]
        return self._push(hip.OP_LN, 'ln', a)

    def attention(self, qkv, B, Ntok, heads, dhead, out_f16):
        a = AttnArgs()
        a.qkv, a.B, a.Ntok, a.heads, a.dhead = qkv.data_ptr(), B, Ntok, heads, dhead
        a.scale = float(dhead) ** -0.5
        a.out_f16 = out_f16.data_ptr()
        self.flops += 4 * B * heads * Ntok * Ntok * dhead
        return self._push(hip.OP_ATTN_F32 if getattr(self, 'fp32', False) else hip.OP_ATTN, 'attn', a)

    def geglu(self, h_f32, M, C4, out_f16):
        a = GegluArgs()
        a.h_f32, a.M, a.C4, a.out_f16 = h_f32.data_ptr(), M, C4, out_f16.data_ptr()
        a.out_is_f32 = 1 if out_f16.dtype == torch.float32 else 0
        return self._push(hip.OP_GEGLU, 'geglu', a)

    def to_cl(self, x, O, Cc, V, Cpad, out):
        a = ToClArgs()
        a.x, a.O, a.C, a.V, a.Cpad, a.out = x.data_ptr(), O, Cc, V, Cpad, out.data_ptr()
        a.out_is_f32 = 1 if out.dtype == torch.float32 else 0
        return self._push(hip.OP_TO_CL, 'tocl', a)

    def stem(self, x, w, scratch, out, O, cin=3, ostride=0):
        a = StemArgs()
        a.x = x.data_ptr()
        a.w0, a.b0, a.w1, a.b1 = [t.data_ptr() for t in w]
        a.scratch, a.out, a.O = scratch.data_ptr(), out.data_ptr(), O
        a.Cin, a.x_ostride = cin, ostride
        self.keep += list(w)
        return self._push(hip.OP_STEM, 'stem', a)


def emit_unet3d_step(b, w, g, x, uc_dev, temb, step, eps_out, dims=(16, 16, 16), lo=0, hi=None, c_dev=None, tables=None,
                     gather_rows=None):
    """One UNet3DModel.forward: x f32 [Ol,3,D,H,W] (NCDHW) -> eps_out f32 [Ol,3,D,H,W].

    'concat' family (``w.concat``; c_dev f32 [Ol, V] = this rank's rows of c_s): the network input is the 5-channel
    NCDHW staging tensor xc = [x_t (3) | c_s (1) | GCN output (1)] (diffusion_shape/network.py:26-28,
    openai_model_3d.py:841-842); its first four channels feed the conv-pool stem.

    Object sharding (multi-GPU): this rank owns objects [lo, hi) of the O-node graph; x / eps_out hold only
    those.  The per-object vector ops (time MLP, GCN over the FULL graph, projections) are computed for all O
    rows on every rank -- they are ~0.1 % of the FLOPs -- while every voxel op runs on the local objects only.
    The one cross-rank dependency is the conv-pool stem code of the other ranks' objects (the "echo"): the ops
    up to ``b.split`` produce this rank's codes, the caller all-gathers them into ``objbuf`` and runs the rest."""
    Ofull, mc, E, gdim = g.O, w.mc, 4 * w.mc, 64
    hi = Ofull if hi is None else hi
    Ol = hi - lo
    O = Ofull
    D0, H0, W0 = dims
    V0 = D0 * H0 * W0
    b.fp32x = getattr(w, 'precision', 'fp16') == 'fp32x'   # round 6: fp32 activations, split-operand f16 contractions (3 x the K)
    b.fp32 = b.fp32x or getattr(w, 'precision', 'fp16') == 'fp32'     # fp32-operand routes: every operand buffer of the plan is fp32
    f16 = torch.float32 if b.fp32 else torch.float16       # dtype of every contraction OPERAND buffer of the plan
    # ---- per-object (rows path) ----
    emb = None
    if tables is None:
        e1 = View(b.buf(O, E))
        b.linear([seg(View(temb, ld=0, width=mc), step=step, step_stride=mc)], w.te0, O, e1, act=hip.ACT_SILU)
        emb = View(b.buf(O, E))
        b.linear([seg(e1)], w.te2, O, emb)
    row0 = lo                                  # first local row inside the per-object row buffers
    xc = None
    cavo, coff = {}, 0
    side0 = len(b.ops)                         # first op of the per-object (echo) chain, see 'side branch' below
    keep_main = []                             # ops of that region that must stay on the main branch
    if w.mp:
        ucw = uc_dev.shape[1]
        Dobj = ucw + gdim + (gdim if w.enable_t_emb else 0)
        objbuf = b.buf(O, Dobj)
        objbuf[:, :ucw].copy_(uc_dev)
        code512 = b.buf(Ol, 512)
        if w.concat:
            assert c_dev is not None and ucw == V0 and tuple(c_dev.shape) == (Ol, V0), 'concat: uc_s / c_s must be [O, D*H*W]'
            xc = b.buf(Ol, 5, V0)
            xc[:, 3].copy_(c_dev)                  # constant over the loop: written once at plan build
            b.copy(xc.data_ptr(), x.data_ptr(), 3 * V0 * 4, rows=Ol, dst_pitch=5 * V0 * 4, src_pitch=3 * V0 * 4)
            b.stem(xc, w.stem, b.buf(Ol, 32 * 512), code512, Ol, cin=4, ostride=5 * V0)
            b.xc = xc
        else:
            b.stem(x, w.stem, b.buf(Ol, 32 * 512), code512, Ol)
        if Ol == Ofull and not getattr(b, 'force_exchange', False):
            b.linear([seg(View(code512))], w.stem_lin, Ol, View(objbuf, col=ucw, ld=Dobj, width=gdim))
            b.codes_local = None
            b.codes_all = None
        else:
            # sharded: this rank's codes go to the (zero-padded, equal-size) send block of the echo all-gather; the gathered
            # [world * block, 64] buffer is copied into the GCN input by the first op after the exchange -- no torch op and
            # no allocation between the two graph launches of a step
            nrows = gather_rows or Ofull
            blk = max(Ol, getattr(b, 'shard_block', Ol))
            b.codes_local = b.buf(blk, gdim, zero=True)
            b.linear([seg(View(code512))], w.stem_lin, Ol, View(b.codes_local))
            b.codes_all = b.buf(max(nrows, Ofull), gdim, zero=True)
        b.split = len(b.ops)                       # <- all-gather point of the multi-GPU loop
        b.code_cols = (ucw, gdim)
        if b.codes_all is not None:
            b.copy(objbuf.data_ptr() + ucw * 4, b.codes_all.data_ptr(), gdim * 4, rows=Ofull, dst_pitch=Dobj * 4,
                   src_pitch=gdim * 4)
        if w.enable_t_emb:
            if tables is not None:
                b.rowsel(tables['t_lin'], step, View(objbuf, col=ucw + gdim, ld=Dobj, width=gdim), rows=O)
            else:
                b.linear([seg(emb)], w.shape_t, O, View(objbuf, col=ucw + gdim, ld=Dobj, width=gdim))
        pred = b.pred_rows = b.dev(w.pred_table[torch.from_numpy(g.p_host)])     # refreshed in place for a new graph
        ctx = emit_gcn(b, w.gcn, g, View(objbuf), Dobj, View(pred), pred.shape[1])
        b.tags.update(ctx=ctx, code=View(objbuf, col=ucw, ld=Dobj, width=gdim))
        emb_ld = w.emb_all.N
        if tables is not None:                     # all objects share t: one table row, broadcast (rowvec_ld = 0)
            emb_all = b.buf(1, w.emb_all.N)
            b.rowsel(tables['emb_all'], step, View(emb_all))
            keep_main.append(len(b.ops) - 1)
            emb_ld = 0
        else:
            b.tags['emb'] = emb
            emb_all = b.buf(O, w.emb_all.N)
            b.linear([seg(emb)], w.emb_all, O, View(emb_all), prologue=hip.PRO_SILU)
        if w.concat:
            # GCN output rows of the local objects -> fifth input channel
            assert ctx.width == V0
            b.copy(xc[:, 4].data_ptr(), ctx.ptr + lo * ctx.ld * 4, V0 * 4, rows=Ol, dst_pitch=5 * V0 * 4, src_pitch=ctx.ld * 4)
        else:
            cav = b.buf(O, w.cav_all.N)
            b.linear([seg(ctx)], w.cav_all, O, View(cav))          # all blocks' cross-attention vectors: ONE product
            for name, (k, Cc) in w.ca.items():
                cavo[name] = View(cav, col=coff, ld=w.cav_all.N, width=Cc)
                coff += Cc
    else:
        # No echo message passing (sdfusion-txt2shape.yaml / sdfusion-txt2shape_concat.yaml): objects are independent and
        # c_s (c_dev, this rank's rows) is the one cross-attention key ('crossattn') or the fourth input channel
        # ('concat').  The per-object row buffers hold the local objects only; nothing is exchanged between ranks.
        assert c_dev is not None, 'shape denoiser without message passing needs the conditioning c_s'
        O, row0, objbuf = Ol, 0, None
        b.split, b.codes_local, b.code_cols, b.codes_all = 0, None, None, None
        emb_ld = w.emb_all.N
        if tables is not None:
            emb_all = b.buf(1, w.emb_all.N)
            b.rowsel(tables['emb_all'], step, View(emb_all))
            emb_ld = 0
        else:
            b.tags['emb'] = emb
            emb_all = b.buf(O, w.emb_all.N)
            b.linear([seg(emb)], w.emb_all, O, View(emb_all), prologue=hip.PRO_SILU)
        if w.concat:
            assert tuple(c_dev.shape) == (Ol, V0), 'concat: c_s must be [O, D*H*W]'
            xc = b.buf(Ol, 4, V0)
            xc[:, 3].copy_(c_dev)
            b.copy(xc.data_ptr(), x.data_ptr(), 3 * V0 * 4, rows=Ol, dst_pitch=4 * V0 * 4, src_pitch=3 * V0 * 4)
            b.xc = xc
        else:
            ctx = View(b.dev(c_dev))
            b.cdev = ctx.t                       # refreshed by the caller for every sample
            cav = b.buf(O, w.cav_all.N)
            b.linear([seg(ctx)], w.cav_all, O, View(cav))          # all blocks' cross-attention vectors: ONE product
            for name, (k, Cc) in w.ca.items():
                cavo[name] = View(cav, col=coff, ld=w.cav_all.N, width=Cc)
                coff += Cc

    # ---- side branch: the echo chain (conv-pool stem, 5-layer GCN, cross-attention vectors: ~45 dependent launches of a few
    # microseconds) feeds nothing before the FIRST transformer block, while conv_in and the first ResBlocks only need the
    # latent and the time-embedding table row.  In the single-GPU 'crossattn' plan it therefore runs as a parallel graph
    # branch (lane 2) and is joined right before the first SpatialTransformer3D -- ~0.4 ms off the critical path per step.
    side_join = {'pending': False}
    # (not with ES_LANES=1: emit_gcn's own FORK/JOIN pairs would then be nested inside this branch and every inner JOIN would
    #  make the main stream wait for the whole echo chain)
    if (w.mp and not w.concat and Ol == Ofull and tables is not None and len(b.ops) > side0 and not b.use_lanes
            and not getattr(b, 'force_exchange', False)):
        for k in range(side0, len(b.ops)):
            if k not in keep_main:
                b.ops[k].lane = 2
        fk = Op()
        fk.kind, fk.lane = hip.OP_FORK, 2
        b.ops.insert(side0, fk)
        b.split += 1 if b.split > side0 else 0
        side_join['pending'] = True

    def join_side():
        if side_join['pending']:
            jn = Op()
            jn.kind, jn.lane = hip.OP_JOIN, 2
            b.ops.append(jn)
            side_join['pending'] = False

    # ---- volume path ----
    sbuf = lambda *sh, **kw: b.buf(*sh, scratch=True, **kw)    # activations: written by the step before they are read (not stored in model files)
    state = dict(h=None, C=0, dims=dims, last_op=None, h16=None)
    O = Ol                                     # ---- from here on: local objects only ----

    def V_(dm):
        return dm[0] * dm[1] * dm[2]

    def need_f16():
        """f16 copy of the current fp32 activation (for convs that read it un-normalised)."""
        if b.fp32:
            return state['h']                      # the fp32 activation is the operand
        if state['h16'] is None:
            t = sbuf(O * V_(state['dims']), state['C'], dtype=f16)
            op = b.ops[state['last_op']]
            op.u.conv.out_f16 = t.data_ptr()
            state['h16'] = t
        return state['h16']

    def run_block(prefix, blk, skip=None):
        for j, it in enumerate(blk):
            name, kind = f'{prefix}.{j}', it[0]
            d = w.items[name]
            dm = state['dims']
            M = O * V_(dm)
            if kind == 'conv_in':
                cpad = getattr(d['conv'], 'cin_part', d['conv'].Cin)
                xcl = sbuf(O * V0, cpad, dtype=f16)                 # (channels padded to the weight image's Cin: 32, fp32 route 16)
                b.to_cl(xc if w.concat else x, O, w.in_ch, V0, cpad, xcl)
                o = sbuf(M, mc)
                state['last_op'] = b.conv(xcl, d['conv'], O, dm, out_f32=o)
                state.update(h=o, C=mc, h16=None)
            elif kind == 'res':
                cin, cout = it[1], it[2]
                x1, C1 = state['h'], state['C']
                x2, C2 = (skip if skip is not None else (None, 0))
                assert C1 + C2 == cin, (name, C1, C2, cin)
                eo, _ = w.emb_slices[name]
                y1 = sbuf(M, cin, dtype=f16)
                raw = sbuf(M, cin, dtype=f16) if 'skip' in d else None
                b.groupnorm(x1, C1, x2, C2, O, V_(dm), d['gn1'][0], d['gn1'][1], 1e-5, True, y1, raw)
                h1 = b.conv_gn_intermediate(y1, d['conv1'], O, dm, rowvec=View(emb_all, col=eo, ld=emb_ld, width=cout, row=row0 if emb_ld else 0))
                y2 = sbuf(M, cout, dtype=f16)
                b.groupnorm(h1, cout, None, 0, O, V_(dm), d['gn2'][0], d['gn2'][1], 1e-5, True, y2)
                o = sbuf(M, cout)
                if 'skip' in d:
                    state['last_op'] = b.conv(y2, d['conv2'], O, dm, bias=d['bias2'], skip=(raw, d['skip']), out_f32=o)
                else:
                    assert x2 is None
                    state['last_op'] = b.conv(y2, d['conv2'], O, dm, res=x1, out_f32=o)
                state.update(h=o, C=cout, h16=None)
                skip = None
            elif kind == 'attn' and w.concat:
                Cc = it[1]
                xin = state['h']
                yn = sbuf(M, Cc, dtype=f16)
                b.groupnorm(xin, Cc, None, 0, O, V_(dm), d['gn'][0], d['gn'][1], 1e-5, False, yn)
                qkv = sbuf(M, 3 * Cc, dtype=f16)
                b.conv(yn, d['qkv'], O, dm, out_f16=qkv)
                at = sbuf(M, Cc, dtype=f16)
                b.attention(qkv, O, V_(dm), w.heads, Cc // w.heads, at)
                o = sbuf(M, Cc)
                state['last_op'] = b.conv(at, d['proj_out'], O, dm, res=xin, out_f32=o)
                state.update(h=o, h16=None)
            elif kind == 'attn':
                join_side()                        # the cross-attention vectors (cavo) are read from here on
                Cc = it[1]
                xin = state['h']
                yn = sbuf(M, Cc, dtype=f16)
                b.groupnorm(xin, Cc, None, 0, O, V_(dm), d['gn'][0], d['gn'][1], 1e-6, False, yn)
                t0 = sbuf(M, Cc)
                b.conv(yn, d['proj_in'], O, dm, out_f32=t0)
                l1 = sbuf(M, Cc, dtype=f16)
                b.layernorm(t0, M, Cc, d['ln1'][0], d['ln1'][1], l1)
                qkv = sbuf(M, 3 * Cc, dtype=f16)
                b.conv(l1, d['qkv'], O, dm, out_f16=qkv)
                at = sbuf(M, Cc, dtype=f16)
                b.attention(qkv, O, V_(dm), w.heads, Cc // w.heads, at)
                # x = attn1(norm1(x)) + x ; x = attn2(norm2(x), ctx) + x  (one key: + per-object vector)
                t2 = sbuf(M, Cc)
                t2h = sbuf(M, Cc, dtype=f16) if ('ffo' in d and not b.fp32) else None      # operand copy of t2 for the folded ff2 + proj_out product
                b.conv(at, d['o1'], O, dm, rowvec=View(cavo[name].t, col=cavo[name].col, ld=cavo[name].ld, width=cavo[name].width, row=row0), res=t0, out_f32=t2,
                       out_f16=t2h)
                l3 = sbuf(M, Cc, dtype=f16)
                b.layernorm(t2, M, Cc, d['ln3'][0], d['ln3'][1], l3)
                gg = sbuf(M, 4 * Cc, dtype=f16)
                if d['ff1'].geglu:               # GEGLU in the contraction epilogue: the [M, 8C] fp32 tensor never exists
                    b.conv(l3, d['ff1'], O, dm, out_f16=gg, epilogue=hip.EPI_GEGLU, out_ld=4 * Cc)
                else:
                    gl = sbuf(M, 8 * Cc)
                    b.conv(l3, d['ff1'], O, dm, out_f32=gl)
                    b.geglu(gl, M, 4 * Cc, gg)
                o = sbuf(M, Cc)
                if 'ffo' in d:
                    state['last_op'] = b.conv(gg, d['ffo'], O, dm, bias=d['ffo_bias'], skip=(t2 if b.fp32 else t2h, d['po']), res=xin, out_f32=o)
                else:
                    t3 = sbuf(M, Cc, dtype=f16)
                    b.conv(gg, d['ff2'], O, dm, res=t2, out_f16=t3)
                    state['last_op'] = b.conv(t3, d['proj_out'], O, dm, res=xin, out_f32=o)
                b.tags[name + '.transformer_blocks.0:in'] = View(t0)
                b.tags[name + '.transformer_blocks.0:attn2'] = View(t2)
                state.update(h=o, h16=None)
            elif kind == 'down':
                a16 = need_f16()
                nd = (dm[0] // 2, dm[1] // 2, dm[2] // 2) if w.concat else (dm[0], dm[1] // 2, dm[2] // 2)
                o = sbuf(O * V_(nd), state['C'])
                state['last_op'] = b.conv(a16, d['conv'], O, nd, mode=hip.CONV_DOWN_DHW if w.concat else hip.CONV_DOWN_HW, out_f32=o)
                state.update(h=o, dims=nd, h16=None)
            elif kind == 'up':
                a16 = need_f16()
                nd = (dm[0] * 2, dm[1] * 2, dm[2] * 2) if w.concat else (dm[0], dm[1] * 2, dm[2] * 2)
                o = sbuf(O * V_(nd), state['C'])
                state['last_op'] = b.conv(a16, d['conv'], O, nd, mode=hip.CONV_UP_DHW if w.concat else hip.CONV_UP_HW, out_f32=o)
                state.update(h=o, dims=nd, h16=None)
            b.tags[name] = View(state['h'])

    inp, mid, out = w.topo
    hs = []
    for i, blk in enumerate(inp):
        run_block(f'input_blocks.{i}', blk)
        hs.append((state['h'], state['C']))
    run_block('middle_block', mid)
    for i, blk in enumerate(out):
        run_block(f'output_blocks.{i}', blk, skip=hs.pop())
    join_side()                                # (a topology without transformer blocks: join before the step ends)
    dm = state['dims']
    yo = sbuf(O * V_(dm), state['C'], dtype=f16)
    b.groupnorm(state['h'], state['C'], None, 0, O, V_(dm), w.out_gn[0], w.out_gn[1], 1e-5, True, yo)
    b.conv(yo, w.out_conv, O, dm, out_f32=eps_out, ncdhw=True)
    return objbuf


for _n in ('_push', 'conv', '_conv32', 'split3', 'conv_gn_intermediate', '_rowgroup_producer', '_rowgroup_stats', 'groupnorm', 'layernorm', 'attention', 'geglu', 'to_cl', 'stem'):
    setattr(Builder, _n, getattr(VolBuilderMixin, _n))


# ------------------------------------------------------------------------------------------------
# VQ-VAE decode epilogue: VQVAE.decode_no_quant (vqvae_networks/network.py:95-103) ->
# VectorQuantizer nearest-code lookup (quantizer.py:68-119) -> post_quant_conv -> Decoder3D.forward
# (vqvae_modules.py:376-409; ResnetBlock :67-126, AttnBlock :128-176, Upsample :24-39)
# ------------------------------------------------------------------------------------------------
def _vq_groups(C):
    return C // 4 if C <= 32 else (32 if C % 32 == 0 else 30)          # Normalize(), vqvae_modules.py:13-21


class VQWeights:
    def __init__(self, sd, device):
        dv = lambda k: own(sd[k], device)
        PC = lambda w, b: PackedConv(sd[w], sd[b] if b else None, device)
        E = sd['quantize.embedding.weight'].detach().double()
        Wp = sd['post_quant_conv.weight'].detach().double().flatten(1)          # [3,3]
        self.codebook = own(E, device)
        self.lut = (mm64(E, Wp.t()) + sd['post_quant_conv.bias'].double()).float().contiguous().to(device)
        self.n_embed = E.shape[0]
        if E.shape[1] != 3:
            raise NotImplementedError('embed_dim != 3')
        d = {k[len('decoder.'):]: v for k, v in sd.items() if k.startswith('decoder.')}
        self.conv_in = PackedConv(d['conv_in.weight'], d['conv_in.bias'], device)

        def res(p):
            r = dict(gn1=(own(d[p + '.norm1.weight'], device), own(d[p + '.norm1.bias'], device)),
                     conv1=PackedConv(d[p + '.conv1.weight'], d[p + '.conv1.bias'], device),
                     gn2=(own(d[p + '.norm2.weight'], device), own(d[p + '.norm2.bias'], device)),
                     conv2=PackedConv(d[p + '.conv2.weight'], d[p + '.conv2.bias'], device))
            r['cin'], r['cout'] = d[p + '.conv1.weight'].shape[1], d[p + '.conv1.weight'].shape[0]
            if (p + '.nin_shortcut.weight') in d:
                r['skip'] = PackedConv(d[p + '.nin_shortcut.weight'].flatten(1), None, device)
                r['bias2'] = (d[p + '.conv2.bias'].float() + d[p + '.nin_shortcut.bias'].float()).contiguous().to(device)
            return r

        self.mid1, self.mid2 = res('mid.block_1'), res('mid.block_2')
        p = 'mid.attn_1'
        Cc = d[p + '.q.weight'].shape[0]
        self.attn = dict(
            C=Cc, gn=(own(d[p + '.norm.weight'], device), own(d[p + '.norm.bias'], device)),
            qkv=PackedConv(torch.cat([d[p + '.q.weight'].flatten(1), d[p + '.k.weight'].flatten(1),
                                      d[p + '.v.weight'].flatten(1)], 0),
                           torch.cat([d[p + '.q.bias'], d[p + '.k.bias'], d[p + '.v.bias']], 0), device),
            proj=PackedConv(d[p + '.proj_out.weight'].flatten(1), d[p + '.proj_out.bias'], device))
        n_lvl = 1 + max(int(k.split('.')[1]) for k in d if k.startswith('up.'))
        self.levels = []
        for lvl in reversed(range(n_lvl)):
            blocks, bi = [], 0
            while f'up.{lvl}.block.{bi}.norm1.weight' in d:
                blocks.append(res(f'up.{lvl}.block.{bi}'))
                bi += 1
            up = None
            if (f'up.{lvl}.upsample.conv.weight') in d:
                up = PackedConv(d[f'up.{lvl}.upsample.conv.weight'], d[f'up.{lvl}.upsample.conv.bias'], device)
            self.levels.append((blocks, up))
        self.out_gn = (own(d['norm_out.weight'], device), own(d['norm_out.bias'], device))
        self.conv_out = PackedConv(d['conv_out.weight'], d['conv_out.bias'], device)


def emit_vq_decode(b, w, z, sdf_out, Oc, zdims=(16, 16, 16)):
    """z f32 [Oc,3,16,16,16] -> sdf_out f32 [Oc,1,64,64,64] (one chunk of objects)."""
    from .hip import VQArgs
    f16 = torch.float16
    sbuf = lambda *sh, **kw: b.buf(*sh, scratch=True, **kw)    # activations: not stored in model files
    st = dict(h=None, C=0, dims=tuple(zdims), last=None, h16=None)
    V_ = lambda dm: dm[0] * dm[1] * dm[2]
    V0 = V_(zdims)
    zq = sbuf(Oc * V0, 32, dtype=f16)
    a = VQArgs()
    a.z, a.codebook, a.lut = z.data_ptr(), w.codebook.data_ptr(), w.lut.data_ptr()
    a.O, a.V, a.n_embed, a.Cpad = Oc, V0, w.n_embed, 32
    a.idx_out = None
    a.out_f16 = zq.data_ptr()
    b._push(hip.OP_VQ, 'vq', a)
    b.keep.append(w)
    o = sbuf(Oc * V0, w.conv_in.N)
    st['last'] = b.conv(zq, w.conv_in, Oc, zdims, out_f32=o)
    st.update(h=o, C=w.conv_in.N)

    def gn(x, Cc, ga, be, act, y, raw=None):
        b.groupnorm(x, Cc, None, 0, Oc, V_(st['dims']), ga, be, 1e-6, act, y, raw, groups=_vq_groups(Cc))

    def res(r):
        dm, M = st['dims'], Oc * V_(st['dims'])
        x, cin, cout = st['h'], r['cin'], r['cout']
        y1 = sbuf(M, cin, dtype=f16)
        raw = sbuf(M, cin, dtype=f16) if 'skip' in r else None
        gn(x, cin, r['gn1'][0], r['gn1'][1], 1, y1, raw)
        h1 = b.conv_gn_intermediate(y1, r['conv1'], Oc, dm)
        y2 = sbuf(M, cout, dtype=f16)
        gn(h1, cout, r['gn2'][0], r['gn2'][1], 1, y2)
        o = sbuf(M, cout)
        if 'skip' in r:
            st['last'] = b.conv(y2, r['conv2'], Oc, dm, bias=r['bias2'], skip=(raw, r['skip']), out_f32=o)
        else:
            st['last'] = b.conv(y2, r['conv2'], Oc, dm, res=x, out_f32=o)
        st.update(h=o, C=cout, h16=None)

    res(w.mid1)
    # AttnBlock: single head of C channels over all voxels
    dm, M, Cc = st['dims'], Oc * V_(st['dims']), w.attn['C']
    x = st['h']
    yn = sbuf(M, Cc, dtype=f16)
    gn(x, Cc, w.attn['gn'][0], w.attn['gn'][1], 0, yn)
    qkv = sbuf(M, 3 * Cc, dtype=f16)
    b.conv(yn, w.attn['qkv'], Oc, dm, out_f16=qkv)
    at = sbuf(M, Cc, dtype=f16)
    b.attention(qkv, Oc, V_(dm), 1, Cc, at)
    o = sbuf(M, Cc)
    st['last'] = b.conv(at, w.attn['proj'], Oc, dm, res=x, out_f32=o)
    st.update(h=o, h16=None)
    res(w.mid2)
    for blocks, up in w.levels:
        for r in blocks:
            res(r)
        if up is not None:
            dm = st['dims']
            t = sbuf(Oc * V_(dm), st['C'], dtype=f16)           # f16 copy of the current activation
            b.ops[st['last']].u.conv.out_f16 = t.data_ptr()
            nd = (dm[0] * 2, dm[1] * 2, dm[2] * 2)
            o = sbuf(Oc * V_(nd), st['C'])
            st['last'] = b.conv(t, up, Oc, nd, mode=hip.CONV_UP_DHW, out_f32=o)
            st.update(h=o, dims=nd, h16=None)
    dm = st['dims']
    yo = sbuf(Oc * V_(dm), st['C'], dtype=f16)
    gn(st['h'], st['C'], w.out_gn[0], w.out_gn[1], 2, yo)        # norm_out -> GELU
    b.conv(yo, w.conv_out, Oc, dm, out_f32=sdf_out, ncdhw=True)
    return dm
