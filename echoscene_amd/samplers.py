"""Loop owners on the host side: thin objects that own packed weights + per-graph plans and call
the native sampling loop.  They mirror ``EchoToLayout`` / ``EchoToShape`` of the reference
(model/networks/diffusion_layout/echo2layout.py, diffusion_shape/echo2shape.py)."""
import os

import torch

from . import hip
from .plan import (Builder, GraphIndex, View, GCNWeights, UNet1DWeights, emit_gcn, emit_unet1d_step, time_tables)
from .plan_vol import UNet3DWeights, emit_unet3d_step, VQWeights, emit_vq_decode
from .schedules import LayoutSchedule, ShapeSchedule, timestep_embedding_table


def _cap(n_triples):
    """triple-row capacity class of a plan (GraphIndex): the count rounded up to a multiple of 32"""
    return max(32, (int(n_triples) + 31) // 32 * 32)


def state_dict_for(module, device=None):
    """The module's state dict as the weight planners read it.  Entries that already live on ``device`` (a GPU) stay there: the
    fp64 folds and the weight re-layouts then run on the GPU from the parameters in place -- a model moved with ``.cuda()`` used to be
    downloaded tensor by tensor (1867 copies), folded on the host and uploaded again (1300 copies), 1.5 s of a process's first scene
    call.  Everything else comes to the host.  The planners never keep an entry itself (plan.own)."""
    dev = None if device is None else torch.device(device)
    on_dev = lambda v: dev is not None and dev.type == 'cuda' and v.is_cuda and (dev.index is None or v.device.index == dev.index)
    return {k: (v.detach() if on_dev(v) else v.detach().cpu()) for k, v in module.state_dict().items()}


def _cpu_sd(module):
    return state_dict_for(module, None)


def gcn_forward(sd, prefix, obj, pred, triples, device=None, weights=None, pooling='avg'):
    """GraphTripleConvNet.forward on the HIP path (setup GCNs and unit tests).
    obj f32[O,Dobj], pred f32[T,Dp] (any device), triples int64[T,3] -> (obj_out, pred_out) on device."""
    device = device or torch.device('cuda')
    gw = weights or GCNWeights(sd, prefix, device, pooling)
    g = GraphIndex(triples, obj.shape[0], device)
    b = Builder(device)
    o = b.dev(obj)
    p = b.dev(pred)
    oo, po = emit_gcn(b, gw, g, View(o), obj.shape[1], View(p), pred.shape[1], want_pred=True)
    plan = b.finish()
    plan.run()
    torch.cuda.synchronize()
    out_p = po.t if po.col == 0 and po.ld == po.t.shape[1] else po.t[:, po.col:po.col + po.width].contiguous()
    return oo.t, out_p


class LayoutDenoiser:
    """UNet1DModel + GaussianDiffusion sampling on the HIP path (loop A of SURVEY.md section 3.1)."""

    def __init__(self, net, diffusion_kwargs, device=None):
        self.device = device or torch.device('cuda')
        self.net = net
        self.w = UNet1DWeights(state_dict_for(net, self.device), net, self.device)
        dk = dict(diffusion_kwargs)
        # every parameterisation GaussianDiffusion can sample with is a coefficient table of the same update op (schedules.py)
        self.sched = LayoutSchedule(dk.get('time_num', 1000), dk.get('beta_start', 1e-4), dk.get('beta_end', 0.02),
                                    dk.get('schedule_type', 'linear'), dk.get('model_mean_type', 'eps'),
                                    dk.get('model_var_type', 'fixedsmall'))
        self.T = self.sched.time_num
        self.temb = timestep_embedding_table(self.sched.timesteps, net.model_channels).to(self.device)
        self.coef = self.sched.coef.to(self.device)
        # time MLP / emb projections / box_time_emb for every step of the schedule (node-independent)
        self.tables = time_tables(self.w, self.temb, self.w.box_t, self.device)
        self._plans, self._last, self.max_plans = {}, None, 4

    def _plan_for(self, obj_embed, triples, clip=False):
        """``clip``: clip_denoised=True of p_sample_loop_sg (the predicted x0 clamped to [-1, 1], diffusion_ddpm.py:243-244) -- a
        property of the plan's update op, so it is part of the cache key.
        Plans are cached by (node count, triple-row capacity): a NEW scene graph of the same size class only rewrites the
        index arrays and the predicate-embedding rows in place (GraphIndex.update) -- no plan rebuild, no graph re-capture
        (0.3 s per scene in round 1).  Capacity = triple count rounded up to a multiple of 32."""
        O = obj_embed.shape[0]
        cap = _cap(triples.shape[0])
        key = (O, cap) if not clip else (O, cap, 'clip')
        sig = hash(triples.detach().cpu().numpy().tobytes())
        st = self._plans.get(key)
        if st is None:
            g = GraphIndex(triples, O, self.device, capacity=cap)
            b = Builder(self.device)
            D = self.net.in_channels
            x = b.buf(O, D)
            step = b.buf(1, dtype=torch.int32, zero=True)
            noise = b.buf(self.T + 1, O, D)
            oe = b.dev(obj_embed)
            objbuf = emit_unet1d_step(b, self.w, g, x, oe, self.temb, step, None, tables=self.tables)
            eps = b.tags['eps']                                  # View: the output conv's K slices (the update sums the slabs)
            n_eps_ops = len(b.ops)
            b.update(hip.OP_DDPM, x, eps, self.coef, step, noise=View(noise[1:].reshape(self.T, O * D), ld=O * D),
                     noise_stride=O * D, inc_step=True, clip_x0=clip)
            plan = b.finish()
            # eps-only plan (same ops minus the update) for step-level parity tests
            b2 = Builder(self.device)
            b2.ops = b.ops[:n_eps_ops]
            b2.keep = b.keep
            b2.tags = b.tags
            st = dict(plan=plan, eps_plan=b2.finish(), x=x, eps=eps, step=step, noise=noise, objbuf=objbuf,
                      oe_w=oe.shape[1], g=g, pred=b.pred_rows, sig=sig)
            if len(self._plans) >= self.max_plans:               # a few size classes stay resident
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = st
        elif st['sig'] != sig:
            st['g'].update(triples)
            st['pred'].copy_(self.w.pred_table[torch.from_numpy(st['g'].p_host)])
            st['sig'] = sig
        st['objbuf'][:, :st['oe_w']].copy_(obj_embed.to(self.device))
        self._last = st
        return st

    @property
    def weight_bytes_per_step(self):
        return self._last['plan'].weight_bytes

    def save_model(self, path, obj_embed, triples):
        """The layout loop of this scene graph as a model file for hosts without Python (es_model_load + es_layout_sample)."""
        from .plan import save_model
        st = self._plan_for(obj_embed, triples)
        return save_model(st['plan'], path, dict(x=st['x'], noise=st['noise'], step=st['step'], coef=self.coef))

    def eps(self, x, obj_embed, triples, iteration):
        """One UNet1DModel.forward at loop iteration ``iteration`` (t = T-1-iteration)."""
        st = self._plan_for(obj_embed, triples)
        st['x'].copy_(x.to(self.device))
        st['eps_plan'].sample(st['step'], int(iteration), 1, use_graph=False)
        torch.cuda.synchronize()
        return st['eps'].value()

    def sample(self, obj_embed, triples, noise=None, n_steps=None, use_graph=True, clip_denoised=False):
        """p_sample_loop_sg: returns x_0 [O, 8].  ``noise`` f32[T+1, O, 8] (row 0 = x_T, row 1+i = draw of
        iteration i) makes the run reproducible against the CPU oracle; None draws it on the device.
        ``clip_denoised``: clamp the predicted x0 to [-1, 1] in every step (diffusion_ddpm.py:243-244; the shipped call passes False)."""
        st = self._plan_for(obj_embed, triples, clip=bool(clip_denoised))
        O, D = st['x'].shape
        n_steps = self.T if n_steps is None else n_steps
        if noise is None:
            st['noise'].normal_()
        else:
            st['noise'][:noise.shape[0]].copy_(noise.to(self.device))
        st['x'].copy_(st['noise'][0])
        st['plan'].sample(st['step'], 0, n_steps, use_graph=use_graph)
        return st['x'].clone()


CANON_OBJECTS = 4      # objects of the reference shard of ShapeDenoiser(deterministic=True): 32 objects over the 8 GPUs of one node


class ShapeDenoiser:
    """UNet3DModel + DDIM sampling on the HIP path (loop B of SURVEY.md section 3.1):
    ``EchoToShape.rel2shape`` without the VQ-VAE decode (echo2shape.py:484-521).

    ``rank`` / ``world``: object sharding over GPUs (echoscene_amd/parallel.py).  world == 1 runs the whole
    step as one hipGraph; world > 1 splits each step at the echo all-gather."""

    def __init__(self, df, model_params=None, ddim_steps=100, device=None, z_shape=(3, 16, 16, 16), rank=0, world=1,
                 group=None, deterministic=False, force_exchange=False, precision='fp16', ddim_eta=0.0):
        """``force_exchange``: build the sharded step structure (stem plan -> code exchange -> main plan) even at world == 1 --
        the one-GPU test of the captured RCCL exchange (tests/test_hip_scene.py).
        ``precision``: 'fp16' (product: fp16 MFMA operands, fp32 accumulate) or 'fp32' -- the VALIDATION route: fp32 activations and
        weights on the exact-fp32 matrix instruction (csrc/es_vol32.hip, 1/16 of the fp16 matrix rate), i.e. the reference's own
        arithmetic (openai_model_3d.py:816-863 is fp32 everywhere), so that the cost of operand rounding is a measurement."""
        # ``ddim_eta`` != 0: stochastic DDIM -- every step adds sigma_t * randn (samplers/ddim.py:256-260); the draws of a run are a
        # device table [S, objects x latent] filled per sample() (or handed in: ``step_noise``)
        # 'fp32x' (round 6): fp32 activations, fp32 attention / norms, and every contraction on the f16 matrix pipe with SPLIT operands
        # (x = hi + lo in f16, three partial products accumulated in fp32: plan_vol.PackedConvX3) -- the reference's arithmetic to
        # ~2^-21 per product at 3x the K of the product route, instead of the 1/16 matrix rate of 'fp32'
        if precision not in ('fp16', 'fp32', 'fp32x'):
            raise ValueError("precision must be 'fp16', 'fp32' or 'fp32x'")
        self.precision = precision
        self.force_exchange = bool(force_exchange)
        self.device = device or torch.device('cuda')
        self.df = df
        net = df.diffusion_net
        self.net = net
        sd = {k[len('diffusion_net.'):]: v for k, v in state_dict_for(df, self.device).items()}
        self.w = UNet3DWeights(sd, net, self.device, precision)
        mp = dict(model_params or {})
        self.ddim_eta = float(ddim_eta)
        self.sched = ShapeSchedule(ddim_steps, mp.get('timesteps', 1000), mp.get('linear_start', 0.00085),
                                   mp.get('linear_end', 0.012), eta=self.ddim_eta)
        self.S = len(self.sched.timesteps)
        self.z_shape = tuple(z_shape)
        self.temb = timestep_embedding_table(self.sched.timesteps, net.model_channels).to(self.device)
        self.coef = self.sched.coef.to(self.device)
        self.rank, self.world, self.group = rank, world, group
        # deterministic=True ("the 8-GPU result must equal the 1-GPU result bit for bit"): the CANONICAL
        # arithmetic -- every K split and partial-sum tiling is a function of the layer alone: the one a reference shard of
        # CANON_OBJECTS objects takes (es_conv_args.O_hint < 0), on every rank of every world size, 1 included.  All world sizes
        # then produce the same bits, and the shard sizes of the 8-GPU partition are the fast ones.  (Rounds 4-5 took the unsharded
        # run's splits as the canon: its small shards then ran K chains tuned for 8x their rows -- x1.9 at 8 GPUs.)  The price
        # is paid where it hurts least: a canonical run with MORE objects per GPU than the reference splits K more than it needs.
        # deterministic=False (the default since round 6): every rank tunes its splits to its own share; results differ between
        # world sizes in fp32 summation order only (5e-4 relative after 4 steps, inside the 2e-2 parity budget).  At 4 objects
        # per GPU the two modes are the same arithmetic.
        self.deterministic = deterministic
        self.tables = time_tables(self.w, self.temb, self.w.shape_t, self.device)
        self._plans, self.max_plans = {}, 2

    def _insert_plan(self, key, st):
        """bounded plan cache: evicting a plan also drops the fused (layout + shape) graph built on it -- each resident plan
        owns its activation buffers, split-K workspace and captured graph (several GB at O = 32)"""
        while len(self._plans) >= self.max_plans:
            old = self._plans.pop(next(iter(self._plans)))
            fk = getattr(self, '_fused_key', None)
            if fk is not None and old.get('plan') is not None and fk[1] == id(old['plan']):
                self._fused, self._fused_key = None, None
        self._plans[key] = st

    def _plan_for(self, uc, triples, c=None):
        from .parallel import partition
        uc = uc.reshape(uc.shape[0], -1)
        O = uc.shape[0]
        concat = self.w.concat
        need_c = concat or not self.w.mp        # c_s is unused only by 'crossattn' WITH message passing
        if need_c:
            if c is None:
                raise ValueError("this shape denoiser needs the conditioning c_s (concat: [O, 4096], echo2shape.py:234-235; "
                                 "no message passing: the cross-attention key [O, 1, context_dim])")
            c = c.reshape(O, -1).to(self.device).float()
        cap = _cap(triples.shape[0])
        key = (O, cap)
        sig = hash(triples.detach().cpu().numpy().tobytes())
        st = self._plans.get(key)
        if st is None:
            lo, hi, block = partition(O, self.world, self.rank)
            if hi == lo:
                # more ranks than object blocks: this rank owns nothing.  It runs no kernels but keeps joining the
                # collectives of the loop (a rank that raised here would leave the others blocked in the all-gather).
                z = lambda *sh: torch.zeros(*sh, device=self.device)
                st = dict(empty=True, x=z(0, *self.z_shape), eps=z(0, *self.z_shape), lo=lo, hi=hi, O=O,
                          codes_local=z(block, 64), codes_all=z(block * self.world, 64), objbuf=None, cdev=None, xc=None)
                st['sig'] = sig
                self._insert_plan(key, st)
                return st
            g = GraphIndex(triples, O, self.device, capacity=cap)
            b = Builder(self.device)
            b.shard_block = block
            b.force_exchange = self.force_exchange
            if self.deterministic:
                b.o_hint = -CANON_OBJECTS   # split-K / partial-sum tiling of the reference shard -> bit-identical latents at every world size
            x = b.buf(hi - lo, *self.z_shape)
            eps = b.buf(hi - lo, *self.z_shape)
            step = b.buf(1, dtype=torch.int32, zero=True)
            ucd = b.dev(uc)
            objbuf = emit_unet3d_step(b, self.w, g, x, ucd, self.temb, step, eps, dims=self.z_shape[1:], lo=lo, hi=hi,
                                      c_dev=c[lo:hi] if need_c else None, tables=self.tables,
                                      gather_rows=block * self.world)
            n_eps_ops = len(b.ops)
            snoise = None
            if self.ddim_eta != 0.0:
                nz = x.numel()
                snoise = b.buf(self.S, nz)
                b.update(hip.OP_DDIM, x, eps, self.coef, step, noise=View(snoise, ld=nz), noise_stride=nz, inc_step=True)
            else:
                b.update(hip.OP_DDIM, x, eps, self.coef, step, inc_step=True)
            st = dict(x=x, eps=eps, step=step, snoise=snoise, objbuf=objbuf, ucw=ucd.shape[1], lo=lo, hi=hi, O=O,
                      codes_local=b.codes_local, codes_all=getattr(b, 'codes_all', None), code_cols=b.code_cols,
                      xc=getattr(b, 'xc', None), g=g, pred=getattr(b, 'pred_rows', None), sig=sig,
                      cdev=getattr(b, 'cdev', None))

            def sub(ops):
                b2 = Builder(self.device)
                b2.ops, b2.keep, b2.tags = ops, b.keep, b.tags
                b2.weight_bytes, b2.flops = b.weight_bytes, b.flops
                return b2.finish()
            st['plan'] = b.finish()
            st['eps_plan'] = sub(b.ops[:n_eps_ops])
            if (self.world > 1 or self.force_exchange) and self.w.mp:
                st['stem_plan'] = sub(b.ops[:b.split])
                st['main_plan'] = sub(b.ops[b.split:])
            self._insert_plan(key, st)
        elif st.get('sig') != sig and not st.get('empty'):
            # same size class, another scene graph: rewrite indices / predicate rows in place, keep plans and graphs
            st['g'].update(triples)
            if st['pred'] is not None:
                st['pred'].copy_(self.w.pred_table[torch.from_numpy(st['g'].p_host)])
            st['sig'] = sig
        if st.get('empty'):
            return st
        if st['objbuf'] is not None:
            st['objbuf'][:, :st['ucw']].copy_(uc.to(self.device))
        if concat:
            st['xc'][:, 3].copy_(c[st['lo']:st['hi']])
        elif st['cdev'] is not None:
            st['cdev'].copy_(c[st['lo']:st['hi']])
        return st

    # -- shard backend protocol of parallel.sharded_ddim_loop ------------------------------------------------
    def codes_local(self, i):
        st = self._cur
        if not st.get('empty'):
            st['stem_plan'].sample(st['step'], int(i), 1, use_graph=self._use_graph)     # captured like the main part
        return st['codes_local']

    def gather_buffers(self):
        """(send block [block, 64], receive buffer [world * block, 64]) of the per-step echo all-gather: pre-allocated, the
        stem plan writes the first, the main plan's first op reads the second."""
        st = self._cur
        return st['codes_local'], st['codes_all']

    def step(self, i, codes_all):
        st = self._cur
        if st.get('empty'):
            return
        if codes_all.data_ptr() != st['codes_all'].data_ptr():      # single-process emulations hand in their own tensor
            st['codes_all'][:codes_all.shape[0]].copy_(codes_all)
        st['main_plan'].sample(st['step'], int(i), 1, use_graph=self._use_graph)

    def latents_local(self):
        return self._cur['x']

    def save_model(self, path, uc, triples, c=None):
        """The DDIM loop of this scene (single GPU) as a model file for hosts without Python (es_model_load + es_shape_sample)."""
        from .plan import save_model
        assert self.world == 1 and not self.force_exchange
        st = self._plan_for(uc, triples, c)
        regions = dict(x=st['x'], step=st['step'], coef=self.coef)
        if st.get('snoise') is not None:
            # ddim_eta != 0: the per-step draws [S, objects x latent] are an INPUT of the loop -- a named region the host fills before
            # es_shape_sample (es_model_region(m, "step_noise")); the file stores whatever the table holds now
            regions['step_noise'] = st['snoise']
        return save_model(st['plan'], path, regions)

    def step_graph(self, group=None):
        """ONE graph per DDIM step of the sharded loop ("RCCL on a dedicated stream, or direct peer
        writes for the 8 KB"): this rank's stem ops -> the RCCL all-gather of the [block, 64] codes -> everything else, captured
        together (torch.cuda.CUDAGraph: ProcessGroupNCCL collectives are capturable), so a step is a single graph launch with no
        host round trip between its three parts.  The step counter lives on the device and is advanced by the captured DDIM
        update.

        OPT-IN (``ES_STEP_GRAPH=1``): the captured exchange has only ever run on a 1-rank NCCL group (one GPU per box here), so
        the default is the three-call step (graph launch / collective / graph launch), which the 2-rank tests do cover.  When it
        is enabled the graph-or-eager decision is COLLECTIVE: every rank attempts the capture, then all ranks all-reduce (MIN) a
        success flag and ALL fall back to the three-call step if any rank failed -- a rank must never replay captured all-gathers
        while a peer issues eager ones after a failed capture.  A rank without objects captures nothing and issues
        one eager all-gather per step; that matches its peers' captured ones call for call.
        Returns the graph, or None (three-call step)."""
        st = self._cur
        if not self._use_graph or ('stem_plan' not in st and not st.get('empty')) or os.environ.get('ES_STEP_GRAPH', '0') != '1':
            return None
        if 'step_graph' in st:
            return st['step_graph']
        st['step_graph'] = None
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_backend(group) != 'nccl':
            return None                          # (decided from the backend alone: the same on every rank)
        send, recv = st['codes_local'], st['codes_all']
        g, ok = None, 1
        # communicator set-up outside any capture; every rank (an empty shard too) takes part (recv is scratch)
        dist.all_gather_into_tensor(recv, send, group=group)
        torch.cuda.synchronize(self.device)
        if not st.get('empty'):
            try:
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        st['stem_plan'].run()
                        dist.all_gather_into_tensor(recv, send, group=group)
                        st['main_plan'].run()
                torch.cuda.current_stream(self.device).wait_stream(s)
            except Exception as e:                 # noqa: BLE001 -- any capture problem on this rank
                import warnings
                warnings.warn('sharded DDIM step: the RCCL exchange could not be captured on rank %d (%s)' % (self.rank, e))
                g, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)      # collective decision: all ranks replay, or none does
        if int(flag.item()) != 1:
            g = None
        st['step_graph'] = g
        return g

    def begin(self, first_step):
        """set the device step counter (the captured step graph does not re-set it every replay)"""
        self._cur['step'].fill_(int(first_step))

    def eps(self, x, uc, triples, iteration, c=None):
        assert self.world == 1
        st = self._plan_for(uc, triples, c)
        st['x'].copy_(x.to(self.device))
        st['eps_plan'].sample(st['step'], int(iteration), 1, use_graph=False)
        return st['eps'].clone()

    def sample(self, uc, triples, noise1=None, n_steps=None, use_graph=True, c=None, step_noise=None):
        """DDIM loop; ``noise1`` f32[1,C,D,H,W] is shared by all objects as in the reference
        (echo2shape.py:507-510); None draws it on the device (world > 1: pass it, or every rank draws its own).
        ``step_noise`` (ddim_eta != 0 only) f32[S, O, C,D,H,W]: the per-step draws of p_sample_ddim, one per OBJECT (noise_like without
        repeat, ldm_diffusion_util.py:289-292); None draws them on the device.
        Returns the latents of ALL objects [O,C,D,H,W] (all-gathered when sharded)."""
        from .parallel import sharded_ddim_loop
        st = self._plan_for(uc, triples, c)
        n_steps = self.S if n_steps is None else n_steps
        if noise1 is None:
            noise1 = torch.randn((1,) + self.z_shape, device=self.device)
        st['x'].copy_(noise1.to(self.device).expand(st['hi'] - st['lo'], *self.z_shape))
        if st.get('snoise') is not None or (self.ddim_eta != 0.0 and st.get('empty')):
            if step_noise is None and self.world > 1:
                # one table for the WHOLE scene from this rank's generator, then this rank's objects: ranks seeded alike (as they
                # must be for noise1) draw what the unsharded run draws -- a local normal_() gave every world size its own noise
                # EVERY rank draws the table, also one that owns no object (more ranks than objects): a rank that skipped the draw left
                # its generator behind the others', and the next call's noise1 differed between ranks.  One call for the
                # whole table, as the unsharded run's normal_() -- a chunked draw is another Philox stream (transient: S*O*latent floats)
                per = 1
                for d in self.z_shape:
                    per *= int(d)
                full = torch.randn(self.S, st['O'], per, device=self.device)
                if st['hi'] > st['lo']:
                    st['snoise'].copy_(full[:, st['lo']:st['hi']].reshape(self.S, -1))
                del full
            elif st.get('snoise') is None:
                pass                                 # (an empty shard with a caller-given table: nothing to fill)
            elif step_noise is None:
                st['snoise'].normal_()
            else:
                sn = step_noise.to(self.device).float()[:, st['lo']:st['hi']]
                st['snoise'][:sn.shape[0]].copy_(sn.reshape(sn.shape[0], -1))
        if st.get('empty'):
            from .parallel import all_gather_rows, sharded_ddim_loop as loop
            if not self.w.mp:
                return all_gather_rows(st['x'], st['O'], self.world, self.group).clone()
            self._cur, self._use_graph = st, use_graph
            return loop(self, st['O'], n_steps, self.world, self.group).clone()
        if (self.world == 1 and not self.force_exchange) or not self.w.mp:
            st['plan'].sample(st['step'], 0, n_steps, use_graph=use_graph)
            if self.world == 1:
                return st['x'].clone()
            from .parallel import all_gather_rows         # no message passing: ranks only meet at the end
            return all_gather_rows(st['x'], st['O'], self.world, self.group).clone()
        self._cur, self._use_graph = st, use_graph
        return sharded_ddim_loop(self, st['O'], n_steps, self.world, self.group).clone()


def sample_layout_and_shape(lay, shp, obj_embed, triples, uc, c=None, layout_noise=None, shape_noise=None, use_graph=True):
    """Both sampling loops of one scene (EchoScene.py:402-419 runs them back to back) as ONE replayed hipGraph: every replay = one
    DDIM shape step on the main branch and ``T_layout // S_shape`` (= 10) ancestral layout steps on a parallel branch
    (plan.combine_plans), so the latency-bound layout chain -- 131 launches of 32 workgroups per step -- runs inside the gaps
    of the MFMA-bound shape step instead of after it (measured on the bench: 24.3 -> 23.0 ms per full step, i.e. the layout
    step disappears).  Left-over layout steps (T not a multiple of S) run afterwards.  Returns (boxes x_0 [O, 8], latents z_0)."""
    from .plan import combine_plans
    if shp.world != 1:
        return lay.sample(obj_embed, triples, noise=layout_noise, use_graph=use_graph), \
            shp.sample(uc, triples, noise1=shape_noise, c=c, use_graph=use_graph)
    st = lay._plan_for(obj_embed, triples)
    if layout_noise is None:
        st['noise'].normal_()
    else:
        st['noise'][:layout_noise.shape[0]].copy_(layout_noise.to(lay.device))
    st['x'].copy_(st['noise'][0])
    ss = shp._plan_for(uc, triples, c)
    if shape_noise is None:
        shape_noise = torch.randn((1,) + shp.z_shape, device=shp.device)
    ss['x'].copy_(shape_noise.to(shp.device).expand(ss['hi'] - ss['lo'], *shp.z_shape))
    r = lay.T // shp.S
    done = 0
    if r >= 1 and use_graph:
        key = (id(st['plan']), id(ss['plan']), r)
        if getattr(shp, '_fused_key', None) != key:
            shp._fused, shp._fused_key = combine_plans(shp.device, ss['plan'], st['plan'], side_repeat=r), key
        st['step'].zero_()
        shp._fused.sample(ss['step'], 0, shp.S, use_graph=True)
        done = r * shp.S
    else:
        ss['plan'].sample(ss['step'], 0, shp.S, use_graph=use_graph)
    if done < lay.T:
        st['plan'].sample(st['step'], done, lay.T - done, use_graph=use_graph)
    return st['x'].clone(), ss['x'].clone()


class VQDecoder:
    """VQVAE.decode_no_quant on the HIP path: latents [O,3,16,16,16] -> SDF [O,1,64,64,64]
    (the once-per-sample epilogue of rel2shape, echo2shape.py:522).  Objects are decoded in chunks so
    that the 64^3-resolution activations stay bounded."""

    def __init__(self, vqvae, device=None, chunk=8):
        self.device = device or torch.device('cuda')
        self.w = VQWeights(state_dict_for(vqvae, self.device), self.device)
        self.chunk = chunk
        self._plans = {}

    def _plan(self, Oc, zdims):
        key = (Oc, zdims)
        if key not in self._plans:
            b = Builder(self.device)
            z = b.buf(Oc, 3, *zdims)
            od = tuple(4 * d for d in zdims)         # placeholder; real dims come from the emitter
            sdf = b.buf(Oc, 1, *od)
            dm = emit_vq_decode(b, self.w, z, sdf, Oc, zdims)
            assert tuple(dm) == od, (dm, od)
            self._plans = {key: dict(plan=b.finish(), z=z, sdf=sdf)}
        return self._plans[key]

    def save_model(self, path, n_objects, zdims=(16, 16, 16)):
        """The decoder for ``n_objects`` latents as a model file for hosts without Python (es_model_load + es_vq_decode)."""
        from .plan import save_model
        st = self._plan(n_objects, tuple(zdims))
        return save_model(st['plan'], path, dict(z=st['z'], sdf=st['sdf']))

    def decode_no_quant(self, z, sync=True):
        """``sync=False``: everything is only enqueued on the current stream (the caller orders consumers)."""
        z = z.to(self.device).float().contiguous()
        O = z.shape[0]
        zdims = tuple(z.shape[2:])
        n_up = len([1 for _, up in self.w.levels if up is not None])
        out = torch.empty(O, 1, *[d * 2 ** n_up for d in zdims], device=self.device)
        for i in range(0, O, self.chunk):
            n = min(self.chunk, O - i)
            st = self._plan(n, zdims)
            st['z'].copy_(z[i:i + n])
            st['plan'].run()
            out[i:i + n].copy_(st['sdf'])
        if sync:
            torch.cuda.synchronize()
        return out
