"""Host-side mirror of the reference's scene-level API:

    model/SGDiff.py       SGDiff                      (facade used by scripts/eval_3dfront.py:387-395,271)
    model/EchoScene.py    Sg2ScDiffModel              (layout + shape)
    model/EchoLayout.py   Sg2BoxDiffModel             (layout only)
    diffusion_layout/echo2layout.py  EchoToLayout     diffusion_shape/echo2shape.py  EchoToShape

Same constructor arguments, attribute names (=> same checkpoint keys), method
names, return structure and dtypes.  The arithmetic is NOT here: every method below prepares small
host-side index/concat glue and calls the HIP samplers (echoscene_amd/samplers.py); training
entry points raise NotImplementedError (out of scope).
"""
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import config as escfg
from .params import _Holder, Emb, mlp
from .graph import GraphTripleConvNet
from .unet import UNet1DModel, DiffusionUNet
from .vqvae import VQVAE


def _dev(module):
    for p in module.parameters():
        return p.device
    return torch.device('cpu')


def _hip_device(d):
    if d.type != 'cuda':
        raise RuntimeError('the sampling path runs on the MI355X HIP library only: move the model to the GPU '
                           '(.cuda()) first -- there is no CPU fallback')
    return d


class DiffusionPoint(_Holder):
    """Key prefix holder: ``LayoutDiff.df.model.*`` (diffusion_layout/diffusion_ddpm.py:550-561)."""

    def __init__(self, denoise_net, diffusion_kwargs):
        super().__init__()
        self.model = denoise_net
        self.diffusion_kwargs = dict(diffusion_kwargs)


class EchoToLayout(nn.Module):
    def __init__(self, config, n_classes=None):
        super().__init__()
        lb = config.layout_branch
        if lb.denoiser != 'unet1d':
            raise NotImplementedError(lb.denoiser)
        # relation_condition (echo2layout.py:12,105): when true the reference hands the relation embeddings to the denoiser as `context`,
        # when false None -- and UNet1DModel.forward overwrites that argument with the GCN output ('crossattn', denoise_net.py:791-792)
        # or never reads it ('concat'): the flag has no effect on SAMPLING.  (Its only effect is in training, where false raises
        # NotImplementedError, echo2layout.py:93-96; training is out of scope here.)  Both values are accepted.
        self.rel_condition = bool(lb.get('relation_condition', True))
        self.df = DiffusionPoint(UNet1DModel(**lb.denoiser_kwargs), lb.diffusion_kwargs)
        self.config = config
        self.translation_dim = lb.get('translation_dim', 3)
        self.size_dim = lb.get('size_dim', 3)
        self.angle_dim = lb.angle_dim
        self.bbox_dim = self.translation_dim + self.size_dim + self.angle_dim
        self.trainable_params = list(self.df.parameters())
        self.scene_ids = None
        self._den = None

    def invalidate(self):
        self._den = None

    def set_input(self, data_dict):
        self.preds = data_dict['preds']
        self.rel = data_dict['c_b']
        self.uc_rel = data_dict['uc_b']

    def forward(self, *a, **k):
        raise NotImplementedError('training (EchoToLayout.forward / losses) is out of scope of this build')

    def _denoiser(self):
        if self._den is None:
            from ..samplers import LayoutDenoiser
            self._den = LayoutDenoiser(self.df.model, self.df.diffusion_kwargs, _hip_device(_dev(self.df)))
        return self._den

    @torch.no_grad()
    def generate_layout_sg(self, box_dim, text=None, ret_traj=False, ddim=False, clip_denoised=False,
                           batch_seeds=None, noise=None):
        """echo2layout.py:112-126.  ``noise`` (optional, f32[T+1,O,box_dim]) makes the run reproducible.
        ``clip_denoised`` reaches the loop (gen_samples_sg, echo2layout.py:108; diffusion_ddpm.py:243-244).  ``ret_traj``, ``ddim``,
        ``text`` and ``batch_seeds`` are accepted and -- exactly as in the reference, whose ``sample`` (echo2layout.py:102-110)
        passes none of them on -- have no effect."""
        samples = self._denoiser().sample(self.uc_rel, self.preds, noise=noise, clip_denoised=bool(clip_denoised))
        s, t = self.size_dim, self.translation_dim
        return {'sizes': samples[:, 0:s].contiguous(),
                'translations': samples[:, s:s + t].contiguous(),
                'angles': samples[:, s + t:self.bbox_dim].contiguous()}


class EchoToShape(object):
    """Plain object, *not* an nn.Module, exactly like the reference: its
    networks are checkpointed under 'shape_df' / 'vqvae' and are not reached by ``SGDiff.cuda()``."""

    def __init__(self, opt):
        self.opt = opt
        self.isTrain = opt.hyper.isTrain
        self.device = opt.hyper.device
        df_conf = escfg.resolve_nested(opt.shape_branch.df_cfg)
        vq_conf = escfg.resolve_nested(opt.shape_branch.vq_cfg)
        dd = vq_conf.model.params.ddconfig
        z_sp = dd.resolution // (2 ** (len(dd.ch_mult) - 1))
        self.z_shape = (dd.z_channels, z_sp, z_sp, z_sp)
        self.df_conf = df_conf
        self.df = DiffusionUNet(df_conf.unet.params, conditioning_key=df_conf.model.params.conditioning_key)
        mp = vq_conf.model.params
        self.vqvae = VQVAE(dict(mp.ddconfig), mp.n_embed, mp.embed_dim)
        ck = opt.shape_branch.get('vq_ckpt', None)
        if ck is not None and os.path.exists(str(ck)):
            sd = torch.load(ck, map_location='cpu')
            self.vqvae.load_state_dict(sd['vqvae'] if 'vqvae' in sd else sd)
        self.df.to(self.device)
        self.vqvae.to(self.device)
        self.df_module, self.vqvae_module = self.df, self.vqvae
        self.trainable_params = list(self.df.parameters())
        self.ddim_steps = 100                           # hard-coded by the reference (echo2shape.py:118)
        if opt.misc.get('debug', 0) == 1:
            self.ddim_steps = 7
        self.uc_scale = 3.0                             # read but never applied (ddim.py:207-228)
        self._den = None
        self._dec = None
        self.triples = None

    def invalidate(self):
        self._den = None
        self._dec = None

    def set_input(self, input=None):
        self.rel = input['c_s']
        self.uc_rel = input['uc_s']
        self.triples = input.get('triples', None)

    def switch_eval(self):
        self.df.eval()
        self.vqvae.eval()

    def forward(self, *a, **k):
        raise NotImplementedError('training (EchoToShape.forward / p_losses) is out of scope of this build')

    def _denoiser(self, ddim_eta=0.0):
        if self._den is None or self._den.S != self._expected_steps() or self._den.ddim_eta != float(ddim_eta):
            from ..samplers import ShapeDenoiser
            self._den = ShapeDenoiser(self.df, self.df_conf.model.params, ddim_steps=self.ddim_steps,
                                      device=_hip_device(_dev(self.df)), z_shape=self.z_shape, ddim_eta=ddim_eta)
        return self._den

    def _expected_steps(self):
        return len(range(0, self.df_conf.model.params.timesteps,
                         self.df_conf.model.params.timesteps // self.ddim_steps))

    def _decoder(self):
        if self._dec is None:
            from ..samplers import VQDecoder
            self._dec = VQDecoder(self.vqvae, _hip_device(_dev(self.vqvae)))
        return self._dec

    @torch.no_grad()
    def rel2shape(self, data, ddim_eta=0.0, noise=None, sync=True, step_noise=None):
        """echo2shape.py:484-525: one latent noise shared by all objects, 100-step DDIM (eta 0, no CFG),
        then VQ-VAE decode_no_quant -> SDF [O,1,64,64,64].  ``noise`` f32[1,C,D,H,W] replaces the
        reference's wall-clock seeded draw (``torch.manual_seed(int(time.time()))``, :502)."""
        self.switch_eval()
        self.set_input(data)
        den = self._denoiser(ddim_eta)             # ddim_eta != 0: sigma_t * randn per step and object (``step_noise`` f32[S,O,C,D,H,W] or drawn)
        if noise is None:
            g = torch.Generator(device=den.device).manual_seed(int(time.time()))
            noise = torch.randn((1,) + tuple(self.z_shape), device=den.device, generator=g)
        # 'concat': c_s is a constant fourth input channel (echo2shape.py:234-235, network.py:26-28); 'crossattn' + mp
        # ignores it (the GCN output overwrites the context, openai_model_3d.py:843-844)
        z = den.sample(self.uc_rel, self.triples, noise1=noise, step_noise=step_noise,
                       c=self.rel if (self.df.conditioning_key == 'concat' or
                                      not self.df.diffusion_net.messsage_passing) else None)
        self.gen_z = z                      # the latents handed to the VQ-VAE (echo2shape.py:521-522), kept like gen_df
        self.gen_df = self._decoder().decode_no_quant(z, sync=sync)
        return self.gen_df


class _SceneModel(nn.Module):
    """Common graph front-end of Sg2ScDiffModel / Sg2BoxDiffModel (EchoScene.py:14-113, EchoLayout.py:9-93)."""

    def __init__(self, vocab, diff_opt, diffusion_bs, embedding_dim, batch_size, gconv_pooling, gconv_num_layers,
                 mlp_normalization, separated, replace_latent, residual, use_angles, use_clip, with_shape):
        super().__init__()
        diff_opt = escfg.to_plain(diff_opt)
        g = embedding_dim
        self.replace_all_latent = replace_latent
        self.batch_size, self.embedding_dim, self.vocab = batch_size, g, vocab
        self.use_angles, self.clip = use_angles, use_clip
        add = 512 if use_clip else 0
        self.obj_classes_list = list(set(vocab['object_idx_to_name']))
        self.edge_list = list(set(vocab['pred_idx_to_name']))
        num_objs, num_preds = len(self.obj_classes_list), len(self.edge_list)
        self.obj_embeddings_ec = Emb(num_objs + 1, 2 * g)
        self.pred_embeddings_ec = Emb(num_preds, 2 * g)
        self.obj_embeddings_dc = Emb(num_objs + 1, 2 * g)
        self.pred_embeddings_man_dc = Emb(num_preds, 2 * g)
        self.out_dim_ini_encoder = 2 * g + add
        self.out_dim_manipulator = 2 * g + add
        kw = dict(hidden_dim=4 * g, pooling=gconv_pooling, mlp_normalization=mlp_normalization, residual=residual)
        self.gconv_net_ec = GraphTripleConvNet(2 * g + add, 2 * g + add, num_layers=gconv_num_layers,
                                               output_dim=self.out_dim_ini_encoder, **kw)
        self.gconv_net_manipulation = GraphTripleConvNet(self.out_dim_ini_encoder + g + 2 * g + add, 2 * g + add,
                                                         num_layers=min(gconv_num_layers, 5),
                                                         output_dim=self.out_dim_manipulator, **kw)
        self.diff_cfg = diff_opt
        self.diffusion_bs = diffusion_bs if diff_opt.hyper.batch_size is None else diff_opt.hyper.batch_size
        self.s_l_separated = separated
        if separated:   # present in checkpoints, never used by sampling
            rel = dict(num_layers=gconv_num_layers, output_dim=self.out_dim_manipulator, **kw)
            if with_shape:
                self.gconv_net_ec_rel_s = GraphTripleConvNet(self.out_dim_manipulator + 2 * g + add, 2 * g + add, **rel)
            self.gconv_net_ec_rel_l = GraphTripleConvNet(self.out_dim_manipulator + 2 * g + add, 2 * g + add, **rel)
        if with_shape:
            self.ShapeDiff = EchoToShape(diff_opt)
            layers = [2 * g + add, 960, 1280]
            if self.ShapeDiff.df.conditioning_key == 'concat':
                layers = [2 * g + add, 1280, 4096]
            self.rel_s_mlp = mlp(layers, mlp_normalization, final_nonlinearity=False)
        self.LayoutDiff = EchoToLayout(diff_opt)
        self.lr_init, self.lr_step, self.lr_evo = diff_opt.hyper.lr_init, diff_opt.hyper.lr_step, diff_opt.hyper.lr_evo
        self._setup_w = None

    # -- bookkeeping the caller touches (eval_3dfront.py:390, SGDiff.py:29,49-84) ---------------------
    def lr_lambda(self, counter):
        if counter < self.lr_step[0]:
            return 1.0
        if counter < self.lr_step[1]:
            return self.lr_evo[0] / self.lr_init
        if counter < self.lr_step[2]:
            return self.lr_evo[1] / self.lr_init
        return self.lr_evo[2] / self.lr_init

    def optimizer_ini(self):
        params = list(self.parameters())
        if hasattr(self, 'ShapeDiff'):
            params += self.ShapeDiff.trainable_params
        self.optimizerFULL = optim.AdamW(params, lr=1e-4)
        self.scheduler = optim.lr_scheduler.LambdaLR(self.optimizerFULL, lr_lambda=self.lr_lambda)
        self.optimizers = [self.optimizerFULL]

    def invalidate(self):
        """Drop packed device images (call after weights change)."""
        self._setup_w = None
        self.LayoutDiff.invalidate()
        if hasattr(self, 'ShapeDiff'):
            self.ShapeDiff.invalidate()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def forward(self, *a, **k):
        raise NotImplementedError('training (forward / select_sdfs) is out of scope of this build')

    # -- setup GCNs (once per sample) -----------------------------------------------------------------
    def _setup(self, enc_objs, enc_triples, enc_text, enc_rel, dec_objs, dec_triples, dec_text, dec_rel,
               change_rows=(), added_rows=(), manip_pred_table='pred_embeddings_ec'):
        """init_encoder + manipulate (EchoScene.py:143-157,181-195) on the HIP rows path.
        Returns obj_embed_ (uc), latent_obj_vecs after splicing (c), device tensors."""
        from ..plan import Builder, GraphIndex, GCNWeights, View, emit_gcn
        dev = _hip_device(_dev(self))
        if self._setup_w is None:
            # packed GCN weights + the three embedding tables the glue reads; dropped by invalidate()
            from ..plan import own
            # (parameters that already live on the device are read in place: samplers.state_dict_for)
            here = lambda v: v.detach() if (v.is_cuda and v.device == dev) else v.detach().cpu()
            sd = {k: here(v) for k, v in nn.Module.state_dict(self).items()
                  if k.startswith(('gconv_net_ec.', 'gconv_net_manipulation.', 'obj_embeddings_ec.', 'pred_embeddings_ec.',
                                   'pred_embeddings_man_dc.'))}
            # (the embedding tables live on the device: the lookups below are device-side row gathers, the CLIP features are
            #  never copied to the host -- VERDICT r2 #9)
            pool = self.gconv_net_ec.gconvs[0].pooling          # 'avg' (every shipped args.json) or 'sum'
            self._setup_w = (GCNWeights(sd, 'gconv_net_ec', dev, pool), GCNWeights(sd, 'gconv_net_manipulation', dev, pool),
                             {k: own(sd[k + '.weight'], dev) for k in ('obj_embeddings_ec', 'pred_embeddings_ec',
                                                                             'pred_embeddings_man_dc')})
        w_ec, w_man, tabs = self._setup_w
        sd = {k + '.weight': v for k, v in tabs.items()}
        g = self.embedding_dim

        def embed(objs, triples, text, rel, ptab):
            p = triples[:, 1].to(dev).long()
            oe = sd['obj_embeddings_ec.weight'].index_select(0, objs.to(dev).long())
            pe = sd[ptab + '.weight'].index_select(0, p)
            if self.clip:
                oe = torch.cat([text.detach().to(dev).float(), oe], 1)
                pe = torch.cat([rel.detach().to(dev).float(), pe], 1)
            return oe.contiguous(), pe.contiguous()

        enc_oe, enc_pe = embed(enc_objs, enc_triples, enc_text, enc_rel, 'pred_embeddings_ec')
        dec_oe, dec_pe = embed(dec_objs, dec_triples, dec_text, dec_rel, manip_pred_table)
        Oe, Od = enc_oe.shape[0], dec_oe.shape[0]
        D = self.out_dim_ini_encoder
        b = Builder(dev)
        g_enc = GraphIndex(enc_triples, Oe, dev)
        latent = emit_gcn(b, w_ec, g_enc, View(b.dev(enc_oe)), enc_oe.shape[1], View(b.dev(enc_pe)), enc_pe.shape[1])
        plan = b.finish()
        plan.run()
        latent = latent.t                                           # [Oe, D] on device
        # append zero rows for added nodes (EchoScene.py:478-486) and the 64-d change slot (:428-435,:393-398)
        if len(added_rows):
            rows = [latent[i:i + 1] for i in range(Oe)]
            for ad in added_rows:
                rows.insert(ad, torch.zeros(1, D, device=dev))
            latent = torch.cat(rows, 0)
        # the reference walks the node index upwards and draws for every index that is IN the list (EchoScene.py:428-435):
        # ascending order, one draw per distinct node, out-of-range entries never match
        change = torch.zeros(Od, g)
        for i in sorted({int(r) for r in change_rows if 0 <= int(r) < Od}):
            change[i] = torch.from_numpy(np.random.normal(0, 1, g)).float()
        man_in = torch.cat([latent, change.to(dev), dec_oe], 1).contiguous()
        b2 = Builder(dev)
        g_dec = GraphIndex(dec_triples, Od, dev)
        latent_m = emit_gcn(b2, w_man, g_dec, View(b2.dev(man_in)), man_in.shape[1], View(b2.dev(dec_pe)),
                            dec_pe.shape[1])
        plan2 = b2.finish()
        plan2.run()
        torch.cuda.synchronize()
        return dec_oe, latent, latent_m.t

    def _layout(self, triples, obj_embed_, relation_cond, noise=None):
        self.LayoutDiff.set_input({'preds': triples, 'box': None, 'uc_b': obj_embed_, 'c_b': relation_cond,
                                   'obj_id_to_scene': None})
        return self.LayoutDiff.generate_layout_sg(box_dim=self.diff_cfg.layout_branch.denoiser_kwargs.in_channels,
                                                  noise=noise)


class Sg2ScDiffModel(_SceneModel):
    def __init__(self, vocab, diff_opt, diffusion_bs=8, embedding_dim=128, batch_size=32, gconv_pooling='avg',
                 gconv_num_layers=5, mlp_normalization='none', separated=False, replace_latent=False,
                 residual=False, use_angles=False, use_clip=True):
        super().__init__(vocab, diff_opt, diffusion_bs, embedding_dim, batch_size, gconv_pooling, gconv_num_layers,
                         mlp_normalization, separated, replace_latent, residual, use_angles, use_clip, True)
        self.sample_obj = self.diff_cfg.shape_branch.sampling
        self._rel_w = None

    def invalidate(self):
        super().invalidate()
        self._rel_w = None

    def _rel_s(self, feat):
        """rel_s_mlp + unsqueeze(1) (EchoScene.py:413-416) on the rows path."""
        from ..plan import Builder, PackedLinear, View, seg, fold_bn
        from .. import hip
        dev = feat.device
        if self._rel_w is None:
            sd = {k: v.detach().cpu() for k, v in self.rel_s_mlp.state_dict().items()}
            if '1.running_mean' in sd:
                W0, b0 = fold_bn(sd, '0', '1')
                last = '3'
            else:
                W0, b0 = sd['0.weight'], sd['0.bias']
                last = '2'
            self._rel_w = (PackedLinear(W0, b0, dev), PackedLinear(sd[last + '.weight'], sd[last + '.bias'], dev))
        l0, l1 = self._rel_w
        b = Builder(dev)
        O = feat.shape[0]
        h = View(b.buf(O, l0.N))
        b.linear([seg(View(feat.contiguous()))], l0, O, h, act=hip.ACT_RELU)
        o = View(b.buf(O, l1.N))
        b.linear([seg(h)], l1, O, o)
        b.keep.append(feat)
        b.finish().run()
        torch.cuda.synchronize()
        return o.t.unsqueeze(1)

    def _shapes(self, gen_shape, dec_objs, dec_triples, obj_embed_, latent, shape_noise=None):
        if not gen_shape:
            return None
        uc = self._rel_s(obj_embed_)
        c = self._rel_s(latent)
        return self.ShapeDiff.rel2shape({'obj_cat': dec_objs, 'triples': dec_triples, 'c_s': c, 'uc_s': uc},
                                        noise=shape_noise)

    def _layout_and_shapes(self, gen_shape, dec_objs, dec_triples, obj_embed_, latent, layout_noise, shape_noise):
        """The two loops only share the setup (the reference runs them back to back, EchoScene.py:402-419).  Here they are
        ONE replayed hipGraph: each replay is a DDIM shape step with ten ancestral layout steps on a parallel branch
        (samplers.sample_layout_and_shape), then the VQ-VAE decode."""
        if not gen_shape:
            return None, self._layout(dec_triples, obj_embed_, latent, layout_noise)
        from ..samplers import sample_layout_and_shape
        uc = self._rel_s(obj_embed_)
        c = self._rel_s(latent)
        L, S = self.LayoutDiff, self.ShapeDiff
        L.set_input({'preds': dec_triples, 'box': None, 'uc_b': obj_embed_, 'c_b': latent, 'obj_id_to_scene': None})
        S.switch_eval()
        S.set_input({'obj_cat': dec_objs, 'triples': dec_triples, 'c_s': c, 'uc_s': uc})
        sden = S._denoiser()
        need_c = S.df.conditioning_key == 'concat' or not S.df.diffusion_net.messsage_passing
        if shape_noise is None:       # the reference seeds this draw from the wall clock (echo2shape.py:502)
            g = torch.Generator(device=sden.device).manual_seed(int(time.time()))
            shape_noise = torch.randn((1,) + tuple(S.z_shape), device=sden.device, generator=g)
        x, z = sample_layout_and_shape(L._denoiser(), sden, obj_embed_, dec_triples, uc, c if need_c else None,
                                       layout_noise=layout_noise, shape_noise=shape_noise)
        s_, t_ = L.size_dim, L.translation_dim
        boxes = {'sizes': x[:, 0:s_].contiguous(), 'translations': x[:, s_:s_ + t_].contiguous(),
                 'angles': x[:, s_ + t_:L.bbox_dim].contiguous()}
        S.gen_z = z
        S.gen_df = S._decoder().decode_no_quant(z, sync=True)
        return S.gen_df, boxes

    @torch.no_grad()
    def sample(self, dec_objs, dec_triplets, dec_text_feat, dec_rel_feat, gen_shape=False, layout_noise=None,
               shape_noise=None):
        """EchoScene.py:388-420."""
        oe, _, latent_m = self._setup(dec_objs, dec_triplets, dec_text_feat, dec_rel_feat,
                                      dec_objs, dec_triplets, dec_text_feat, dec_rel_feat)
        sdf, boxes = self._layout_and_shapes(gen_shape, dec_objs, dec_triplets, oe, latent_m, layout_noise, shape_noise)
        return {'shapes': sdf}, boxes

    def _edited(self, enc, dec, touched, added, gen_shape, layout_noise, shape_noise):
        oe, latent, latent_m = self._setup(*enc, *dec, change_rows=touched, added_rows=added)
        if not self.replace_all_latent:
            latent = latent.clone()
            for t in sorted(touched):                       # take original nodes when untouched (:440-448)
                if 0 <= int(t) < latent.shape[0]:
                    latent[t] = latent_m[t]
        else:
            latent = latent_m
        sdf, boxes = self._layout_and_shapes(gen_shape, dec[0], dec[1], oe, latent, layout_noise, shape_noise)
        keep = torch.ones(len(boxes['translations']), 1, device=oe.device)
        for t in touched:
            if 0 <= int(t) < keep.shape[0]:
                keep[int(t)] = 0
        return keep, {'shapes': sdf}, boxes

    @torch.no_grad()
    def sample_with_changes(self, enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triplets,
                            dec_text_feat, dec_rel_feat, manipulated_nodes, gen_shape=False, layout_noise=None,
                            shape_noise=None):
        """EchoScene.py:422-472."""
        return self._edited((enc_objs, enc_triples, enc_text_feat, enc_rel_feat),
                            (dec_objs, dec_triplets, dec_text_feat, dec_rel_feat),
                            list(manipulated_nodes), [], gen_shape, layout_noise, shape_noise)

    @torch.no_grad()
    def sample_with_additions(self, enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triplets,
                              dec_text_feat, dec_rel_feat, missing_nodes, gen_shape=False, layout_noise=None,
                              shape_noise=None):
        """EchoScene.py:474-532: zero rows inserted at ``missing_nodes[i] + i``; note the reference draws the
        change noise for rows listed in ``missing_nodes`` (:489-494) but splices / masks ``nodes_added``."""
        added = [m + i for i, m in enumerate(missing_nodes)]
        oe, latent, latent_m = self._setup(enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triplets,
                                           dec_text_feat, dec_rel_feat, change_rows=list(missing_nodes),
                                           added_rows=added)
        if not self.replace_all_latent:
            latent = latent.clone()
            for t in sorted(added):
                latent[t] = latent_m[t]
        else:
            latent = latent_m
        sdf, boxes = self._layout_and_shapes(gen_shape, dec_objs, dec_triplets, oe, latent, layout_noise, shape_noise)
        keep = torch.ones(len(boxes['translations']), 1, device=oe.device)
        for t in added:
            keep[t] = 0
        return keep, {'shapes': sdf}, boxes

    def state_dict(self, epoch=None, counter=None, **kw):
        """EchoScene.py:534-543 when called with (epoch, counter); plain nn.Module.state_dict otherwise."""
        sd = super().state_dict(**kw)
        if epoch is None and counter is None:
            return sd
        sd.update({'epoch': epoch, 'counter': counter, 'opt': self.optimizerFULL.state_dict(),
                   'vqvae': self.ShapeDiff.vqvae_module.state_dict(),
                   'shape_df': self.ShapeDiff.df_module.state_dict()})
        return sd


class Sg2BoxDiffModel(_SceneModel):
    def __init__(self, vocab, diff_opt, diffusion_bs=8, embedding_dim=128, batch_size=32, gconv_pooling='avg',
                 gconv_num_layers=5, mlp_normalization='none', separated=False, replace_latent=False,
                 residual=False, use_angles=False, use_clip=True):
        super().__init__(vocab, diff_opt, diffusion_bs, embedding_dim, batch_size, gconv_pooling, gconv_num_layers,
                         mlp_normalization, separated, replace_latent, residual, use_angles, use_clip, False)

    # EchoLayout's manipulator uses pred_embeddings_man_dc (EchoLayout.py:154), EchoScene pred_embeddings_ec
    @torch.no_grad()
    def sampleBoxes(self, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, layout_noise=None):
        """EchoLayout.py:291-307."""
        oe, _, latent_m = self._setup(dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                                      dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                                      manip_pred_table='pred_embeddings_man_dc')
        return self._layout(dec_triplets, oe, latent_m, layout_noise)

    @torch.no_grad()
    def sampleBoxes_with_changes(self, enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triples,
                                 dec_text_feat, dec_rel_feat, manipulated_nodes, layout_noise=None):
        touched = list(manipulated_nodes)
        oe, latent, latent_m = self._setup(enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triples,
                                           dec_text_feat, dec_rel_feat, change_rows=touched,
                                           manip_pred_table='pred_embeddings_man_dc')
        if not self.replace_all_latent:
            latent = latent.clone()
            for t in sorted(touched):
                if 0 <= int(t) < latent.shape[0]:
                    latent[t] = latent_m[t]
        else:
            latent = latent_m
        boxes = self._layout(dec_triples, oe, latent, layout_noise)
        # f32 [O,1] tensor on the model's device (EchoLayout.py:342-348); only the _with_additions variant returns a list
        keep = torch.ones(len(boxes['translations']), 1, device=oe.device)
        for t in touched:
            if 0 <= int(t) < keep.shape[0]:
                keep[int(t)] = 0
        return keep, boxes

    @torch.no_grad()
    def sampleBoxes_with_additions(self, enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triples,
                                   dec_text_feat, dec_rel_feat, missing_nodes, layout_noise=None):
        added = [m + i for i, m in enumerate(missing_nodes)]
        oe, latent, latent_m = self._setup(enc_objs, enc_triples, enc_text_feat, enc_rel_feat, dec_objs, dec_triples,
                                           dec_text_feat, dec_rel_feat, change_rows=added,   # sic: nodes_added here,
                                           added_rows=added, manip_pred_table='pred_embeddings_man_dc')   # EchoLayout.py:367-372
        if not self.replace_all_latent:
            latent = latent.clone()
            for t in sorted(added):
                latent[t] = latent_m[t]
        else:
            latent = latent_m
        boxes = self._layout(dec_triples, oe, latent, layout_noise)
        keep = [0 if i in added else 1 for i in range(len(boxes['translations']))]
        return keep, boxes

    def state_dict(self, epoch=None, counter=None, **kw):
        sd = super().state_dict(**kw)
        if epoch is None and counter is None:
            return sd
        sd.update({'epoch': epoch, 'counter': counter, 'opt': self.optimizerFULL.state_dict()})
        return sd


class SGDiff(nn.Module):
    """model/SGDiff.py:6-129 -- the facade ``scripts/eval_3dfront.py`` constructs and calls."""

    def __init__(self, type, diff_opt, vocab, replace_latent=False, with_changes=True, residual=False,
                 gconv_pooling='avg', with_angles=False, clip=True, separated=False):
        super().__init__()
        assert type in ['echoscene', 'echolayout'], '{} is not included'.format(type)
        assert replace_latent is not None and with_changes is not None
        self.type_, self.vocab, self.with_angles, self.epoch, self.diff_opt = type, vocab, with_angles, 0, diff_opt
        cls = Sg2ScDiffModel if type == 'echoscene' else Sg2BoxDiffModel
        self.diff = cls(vocab, diff_opt, diffusion_bs=16, embedding_dim=64, mlp_normalization='batch',
                        separated=separated, gconv_num_layers=5, gconv_pooling=gconv_pooling, use_angles=with_angles,
                        replace_latent=replace_latent, residual=residual, use_clip=clip)
        self.diff.optimizer_ini()
        self.counter = 0

    def forward_mani(self, *a, **k):
        raise NotImplementedError('training (forward_mani) is out of scope of this build')

    def load_networks(self, exp, epoch, strict=True, restart_optim=False, load_shape_branch=True):
        """SGDiff.py:49-84: same checkpoint layout ('opt', 'vqvae', 'shape_df', 'epoch', 'counter' + module keys)."""
        ckpt = torch.load(os.path.join(exp, 'checkpoint', 'model{}.pth'.format(epoch)), map_location='cpu')
        opt_state = ckpt.pop('opt', None)
        if load_shape_branch and self.type_ == 'echoscene':
            if 'vqvae' in ckpt and 'shape_df' in ckpt:
                self.diff.ShapeDiff.vqvae.load_state_dict(ckpt.pop('vqvae'))
                self.diff.ShapeDiff.df.load_state_dict(ckpt.pop('shape_df'))
                self.diff.ShapeDiff.df_module = self.diff.ShapeDiff.df
                self.diff.ShapeDiff.vqvae_module = self.diff.ShapeDiff.vqvae
                self.diff.ShapeDiff.invalidate()
                print('[*] shape branch has successfully been restored from: %s'
                      % os.path.join(exp, 'checkpoint', 'model{}.pth'.format(epoch)))
            else:
                print('no vqvae or shape_df recorded. Assume it is only the layout branch')
        if 'epoch' in ckpt and 'counter' in ckpt:
            self.epoch, self.counter = ckpt.pop('epoch'), ckpt.pop('counter')
        else:
            print('no epoch or counter recorded.')
        ckpt.pop('vqvae', None)
        ckpt.pop('shape_df', None)
        self.diff.load_state_dict(ckpt, strict=strict)
        print('[*] GCN and layout branch has successfully been restored from: %s'
              % os.path.join(exp, 'checkpoint', 'model{}.pth'.format(epoch)))
        if not restart_optim:
            self.diff.optimizerFULL.load_state_dict(opt_state)
            self.diff.scheduler = optim.lr_scheduler.LambdaLR(self.diff.optimizerFULL, lr_lambda=self.diff.lr_lambda,
                                                              last_epoch=int(self.counter - 1))

    def cuda(self, device=None):
        r = super().cuda(device)
        if self.type_ == 'echoscene':         # the reference places the shape nets by cfg.hyper.device (echo2shape.py:77);
            dev = _dev(self)                  # keeping them on the same GPU as the rest is what that amounts to
            self.diff.ShapeDiff.df.to(dev)
            self.diff.ShapeDiff.vqvae.to(dev)
        self.diff.invalidate()
        return r

    def sample_box_and_shape(self, dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat, gen_shape=False,
                             **noise):
        if self.type_ == 'echolayout':
            return self.diff.sampleBoxes(dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                                         layout_noise=noise.get('layout_noise'))
        shape_dict, layout_dict = self.diff.sample(dec_objs, dec_triplets, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                   gen_shape=gen_shape, **noise)
        return {**shape_dict, **layout_dict}

    def sample_boxes_and_shape_with_changes(self, enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat,
                                            dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                            manipulated_nodes, gen_shape=False, **noise):
        if self.type_ == 'echolayout':
            return self.diff.sampleBoxes_with_changes(enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat,
                                                      dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                                      manipulated_nodes, layout_noise=noise.get('layout_noise'))
        keep, shape_dict, layout_dict = self.diff.sample_with_changes(
            enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat, dec_objs, dec_triples,
            encoded_dec_text_feat, encoded_dec_rel_feat, manipulated_nodes, gen_shape=gen_shape, **noise)
        return keep, {**shape_dict, **layout_dict}

    def sample_boxes_and_shape_with_additions(self, enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat,
                                              dec_objs, dec_triples, encoded_dec_text_feat, encoded_dec_rel_feat,
                                              missing_nodes, gen_shape=False, **noise):
        if self.type_ == 'echolayout':
            keep, layout_dict = self.diff.sampleBoxes_with_additions(
                enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat, dec_objs, dec_triples,
                encoded_dec_text_feat, encoded_dec_rel_feat, missing_nodes, layout_noise=noise.get('layout_noise'))
            return layout_dict                 # sic: the reference drops ``keep`` here (SGDiff.py:113-115)
        keep, shape_dict, layout_dict = self.diff.sample_with_additions(
            enc_objs, enc_triples, encoded_enc_text_feat, encoded_enc_rel_feat, dec_objs, dec_triples,
            encoded_dec_text_feat, encoded_dec_rel_feat, missing_nodes, gen_shape=gen_shape, **noise)
        return keep, {**shape_dict, **layout_dict}

    def save(self, exp, outf, epoch, counter=None):
        torch.save(self.diff.state_dict(epoch, counter), os.path.join(exp, outf, 'model{}.pth'.format(epoch)))
