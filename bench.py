#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: denoising steps/sec of EchoScene's sampling loops.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A "step" is one denoising iteration over one synthetic 32-node scene graph (BASELINE.json /
SURVEY.md section 8(d)): with ``--workload layout`` one ``p_sample_sg`` of the 1000-step DDPM
box loop (BASELINE configs[1]); with ``--workload full`` one layout step + one DDIM shape step
over the same O objects (the metric's "layout+SDF" step; needs the volume path).
Inputs (weights, graph, noise tables) are resident in HBM before the timed region starts.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).
  --scaling strong (default; BASELINE configs[3]): ONE 32-node scene, its objects block-partitioned over the N GPUs (the
      per-object shape UNet is 99.98 % of the step FLOPs).  The path has a real exchange step here: every DDIM step
      all-gathers the 64-d conv-pool codes ("echo" message passing, 8 KB) and every rank runs the small shape GCN on the
      full graph; the layout branch (1 % of the work) is replicated.  value = steps / max-over-ranks time -- the same
      quantity as at N = 1.
  --scaling weak (BASELINE configs[4] shape): --scenes-per-gpu scenes per GPU, partitioned BY SCENE (the collated batch graph
      is block diagonal, threedfront_dataset.py:698-701), so no rank needs another rank's codes: no data-path collective at
      all; value = total scenes x steps / max-over-ranks time (scene-steps/s).
At N = 1 the line also carries the BASELINE configs[1] (layout only) and configs[2] (16-node full step) results as
``sub_records``.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FUSE_DEFAULT = True            # --fuse-loops default: measured 24.34 -> 23.02 ms per full step (profiles/r02_notes.md)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no sparsity)


def build_layout(dev, O, seed, graph_seed=None):
    from echoscene_amd import synth, config as escfg
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(512))
    synth.seeded_fill_(net, prefix='bench.layout.')
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    objs, triples = synth.synthetic_graph(O, seed=seed if graph_seed is None else graph_seed)
    rs = torch.Generator().manual_seed(seed if graph_seed is None else graph_seed)
    obj_embed = torch.randn(O, 640, generator=rs)
    return net, den, obj_embed, triples


def build_shape(dev, O, seed, triples, rank=0, world=1, deterministic=False):
    from echoscene_amd import synth, config as escfg
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    conf = escfg.shape_df_conf(224)
    df = DiffusionUNet(conf.unet.params, conditioning_key='crossattn')
    synth.seeded_fill_(df, prefix='bench.shape.')
    den = ShapeDenoiser(df, conf.model.params, ddim_steps=100, device=dev, rank=rank, world=world, deterministic=deterministic)
    uc = torch.randn(O, 1, 1280, generator=torch.Generator().manual_seed(seed + 1))
    return df, den, uc


def _stats(xs, nd=4):
    """median / min / max of a list of repetitions (the JSON shows the spread)"""
    xs = sorted(float(x) for x in xs)
    n = len(xs)
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return {'median': round(med, nd), 'min': round(xs[0], nd), 'max': round(xs[-1], nd), 'reps': n}


def time_dominant_kernel(ss, dev, reps=5):
    """roofline.achieved for the dominant kernel: the step's conv launches that the library dispatches to the
    warp-specialised 256-row-tile kernel k_conv_ws -- >= 256 tiles of 256 rows (the 16^3 and 16x8x8 levels), or fewer tiles with K
    split over them (the 16x4x4 level; the rule below mirrors es_conv_mfma_f16) -- are replayed as their own plan and timed with
    HIP events on the stream they are launched on (split launches include their fixed-order reduction kernel).
    ``reps`` separately timed replays; returns (TFLOP/s at the median replay, median avg us per launch, launches, _stats of the avg us)."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder
    plan = ss['plan']
    ops, flops = [], 0.0
    for op in list(plan._arr):
        if op.kind != hip.OP_CONV:
            continue
        c = op.u.conv
        M = c.O * c.D * c.H * c.W
        if (c.N <= 4 and c.Cin <= 64) or c.Cin == 32 or c.N % 4 or c.out_ld < 0:
            continue                             # direct small-N kernel / NCDHW output / (Cin == 32: the zero-padded 3-channel input conv)
        tiles = ((M + 255) // 256) * ((c.N + 223) // 224)
        if tiles < 256:
            nks = c.taps * (c.Cin // 32) + (c.Cin2 // 32 if c.a2 else 0)
            s2 = min(256 // tiles, 16 if M * c.N <= (1 << 22) else 8)
            while s2 > 1 and nks // s2 < 24:
                s2 -= 1
            if not (s2 >= 2 and tiles * s2 >= 160 and c.epilogue == 0 and c.workspace):
                continue                         # stays on the 128- / 64-row k_conv_lean tiles
        ops.append(op)
        flops += 2.0 * M * c.N * (c.Cin * c.taps + c.Cin2)
    if not ops:
        return None
    b = Builder(dev)
    b.ops, b.keep = ops, plan.keep
    sub = b.finish()
    sub.run()
    torch.cuda.synchronize()
    per = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sub.run()
        e1.record()
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) * 1e3)
    st = _stats([u / len(ops) for u in per], 2)
    us = st['median'] * len(ops)
    return flops / us / 1e6, us / len(ops), len(ops), st


PMC_TRAFFIC_FILES = ('r06_pmc_traffic.json', 'r05b_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json')


def _pmc_file():
    for name in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            return path
    return None


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` as measured by the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each in
    its own pass, FETCH doubled as the MI355X guide prescribes for wide streaming reads on gfx950): bench.py cannot collect
    counters itself, so it reports the number recorded under profiles/ for this same command -- or null."""
    path = _pmc_file()
    try:
        with open(path) as f:
            d = json.load(f)
        if kernel == 'k_conv_ws' and 'k_conv_ws3' in d and 'k_conv_ws' in d:
            # round 6: the dominant kernel's launches are k_conv_ws and (3x3x3 SAME convs) its shared-A-tile variant k_conv_ws3:
            # the launch-weighted mean of the two families
            a, b = d['k_conv_ws'], d['k_conv_ws3']
            na, nb = a.get('launches_sampled', 0), b.get('launches_sampled', 0)
            return int(round((a['hbm_bytes_per_launch'] * na + b['hbm_bytes_per_launch'] * nb) / max(na + nb, 1)))
        return d.get(kernel, {}).get('hbm_bytes_per_launch')
    except Exception:
        return None


def route_options_string():
    """the library's route options (es_options_string: what a model file records) -- the run's kernel routes, for the record"""
    import ctypes
    from echoscene_amd import hip
    buf = ctypes.create_string_buffer(1024)
    hip.lib().es_options_string(buf, 1024)
    return buf.value.decode()


def pmc_traffic_source():
    """file + git blob id of the committed counter summary ``roofline.traffic`` is read from (so that a stale file is visible:
    the id changes whenever the passes are re-collected), or null"""
    import hashlib
    path = _pmc_file()
    if path is None:
        return None
    data = open(path, 'rb').read()
    return {'file': os.path.relpath(path, ROOT), 'git_blob': hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()}


def sub_records(dev, lay, use_graph, a):
    """BASELINE configs[1] and configs[2] from the same command (N = 1): the layout loop alone (already timed on its stream) and
    the full step of a 16-node scene."""
    recs = [{'config': 'configs[1]: EchoLayout box diffusion, 32-node graph, 1000-step DDPM', 'metric': 'layout steps/s',
             'value': lay['steps_per_s'], 'ms_per_step': lay['ms_per_step'], 'kernels_per_step': lay['kernels_per_step'],
             'launches_per_step': lay['launches_per_step'], 'hbm_GBps_algorithmic': lay['hbm_GBps_algorithmic']}]
    O = 16
    net, den, obj_embed, triples = build_layout(dev, O, seed=116)
    df, sden, uc = build_shape(dev, O, 116, triples)
    n = max(10, min(a.steps, 30))
    den.sample(obj_embed, triples, noise=None, n_steps=3, use_graph=use_graph)
    noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    sden.sample(uc, triples, noise1=noise1, n_steps=2, use_graph=use_graph)
    st, ss = next(iter(den._plans.values())), next(iter(sden._plans.values()))
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    st['plan'].sample(st['step'], 0, n, use_graph=use_graph)
    e[1].record()
    ss['plan'].sample(ss['step'], 0, n, use_graph=use_graph)
    e[2].record()
    torch.cuda.synchronize()
    tl, tsh = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    recs.append({'config': 'configs[2]: EchoScene full (layout+SDF) 16-node graph, 64^3 SDF', 'metric': 'full steps/s',
                 'value': round(1e3 / (tl + tsh), 3), 'ms_per_step': round(tl + tsh, 4), 'layout_ms': round(tl, 4),
                 'shape_ms': round(tsh, 4), 'steps_timed': n,
                 'shape_TFLOPs': round(ss['plan'].flops / (tsh * 1e-3) / 1e12, 1)})
    del den, sden, st, ss
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free > 120 * 2 ** 30:
        recs.append(configs4_record(dev, use_graph))
    return recs


MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32: exact fp32 in / fp32 accumulate (MI355X_MICROARCH.md)


def fp32_route_record(dev, df, sden, uc, triples, conf_params, use_graph, n=2):
    """The fp32-OPERAND validation route (ShapeDenoiser(precision='fp32'), csrc/es_vol32.hip: the reference's own arithmetic on the
    same plan) at the benchmarked shape: its step rate with a roofline against the fp32 matrix peak, and -- the point of the route --
    the difference between one evaluation of the fp16-operand product path and the fp32-operand path on the SAME weights and inputs,
    i.e. the cost of operand rounding as a measurement (the product path's parity tolerance is 2e-2)."""
    from echoscene_amd.samplers import ShapeDenoiser
    O = uc.shape[0]
    den32 = ShapeDenoiser(df, conf_params, ddim_steps=100, device=dev, precision='fp32')
    x = torch.randn(O, 3, 16, 16, 16, generator=torch.Generator().manual_seed(21))
    e32 = den32.eps(x, uc, triples, iteration=50)
    e16 = sden.eps(x, uc, triples, iteration=50)
    d = (e16 - e32).double()
    rel_max = (d.abs().max() / e32.double().abs().max()).item()
    rel_rms = (d.pow(2).mean().sqrt() / e32.double().pow(2).mean().sqrt()).item()
    ss = next(iter(den32._plans.values()))
    ss['plan'].sample(ss['step'], 0, 1, use_graph=use_graph)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record(); ss['plan'].sample(ss['step'], 0, n, use_graph=use_graph); e[1].record()
    torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / n
    tf = ss['plan'].flops / (ms * 1e-3) / 1e12
    rec = {'config': 'fp32-operand validation route of the shape step (precision=fp32: fp32 activations and weights on v_mfma_f32_16x16x4_f32), '
                     '%d objects, same plan structure as the product path' % O,
           'metric': 'shape steps/s', 'value': round(1e3 / ms, 4), 'ms_per_step': round(ms, 2), 'steps_timed': n, 'dtype': 'f32',
           'roofline': {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'kernel': 'k_conv_f32 (whole step)'},
           'fp16_product_vs_fp32_route_eps': {'max_abs_diff_over_max_abs': rel_max, 'rel_rms': rel_rms,
                                              'what': 'one UNet3D + echo-GCN evaluation at DDIM iteration 50, same weights / inputs'}}
    del den32, ss
    torch.cuda.empty_cache()
    return rec


def fp32x_route_record(dev, df, sden, uc, triples, conf_params, use_graph, n=3):
    """Round 6: the SPLIT-OPERAND route (ShapeDenoiser(precision='fp32x')) at the benchmarked shape -- fp32 activations,
    fp32 attention / norms, every contraction as three f16 partial products (hi x hi + lo x hi + hi x lo, fp32 accumulate) on the
    product kernels: the reference's arithmetic to ~2^-21 per product as a USABLE mode (the 'fp32' route runs the 1/16-rate fp32
    matrix instruction).  Its step time, the matrix rate on the 3x FLOPs it executes, and its difference to the 'fp32' route."""
    from echoscene_amd.samplers import ShapeDenoiser
    O = uc.shape[0]
    denx = ShapeDenoiser(df, conf_params, ddim_steps=100, device=dev, precision='fp32x')
    x = torch.randn(O, 3, 16, 16, 16, generator=torch.Generator().manual_seed(21))
    ex = denx.eps(x, uc, triples, iteration=50)
    e16 = sden.eps(x, uc, triples, iteration=50)
    d = (e16 - ex).double()
    rel_rms = (d.pow(2).mean().sqrt() / ex.double().pow(2).mean().sqrt()).item()
    ss = next(iter(denx._plans.values()))
    ss['plan'].sample(ss['step'], 0, 1, use_graph=use_graph)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record(); ss['plan'].sample(ss['step'], 0, n, use_graph=use_graph); e[1].record()
    torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / n
    tf = 3.0 * ss['plan'].flops / (ms * 1e-3) / 1e12           # executed: three f16 partial products per product of the model
    rec = {'config': 'split-operand route of the shape step (precision=fp32x: fp32 activations, contractions as 3 f16 partial products with fp32 '
                     'accumulate on the product kernels), %d objects' % O,
           'metric': 'shape steps/s', 'value': round(1e3 / ms, 4), 'ms_per_step': round(ms, 2), 'steps_timed': n, 'dtype': 'f32 activations, 2 x f16 split operands, f32 accumulate',
           'roofline': {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tf / MFMA_F16_PEAK_TFLOPS, 4), 'kernel': 'whole step, executed FLOPs (3 x the model FLOPs)'},
           'fp16_product_vs_fp32x_route_eps': {'rel_rms': rel_rms, 'what': 'one UNet3D + echo-GCN evaluation at DDIM iteration 50, same weights / inputs'}}
    del denx, ss
    torch.cuda.empty_cache()
    return rec


def configs4_record(dev, use_graph, scenes=8, O=32, n=5):
    """BASELINE configs[4] as ONE rank of the 8-GPU batch run sees it (64 scenes x 32 nodes over 8 GPUs = 8 scenes = 256 objects per
    GPU, partitioned by scene: no collective inside the steps), measured on this GPU: scene-steps/s and the roofline of the shape
    step at that shape (every conv launch has >= 2048 tiles of 256 rows)."""
    from echoscene_amd import synth
    graphs = [synth.synthetic_graph(O, seed=400 + s) for s in range(scenes)]
    _, triples = synth.collate_graphs(graphs)
    net, den, _, _ = build_layout(dev, O, seed=100)
    obj_embed = torch.randn(O * scenes, 640, generator=torch.Generator().manual_seed(401))
    df, sden, uc = build_shape(dev, O * scenes, 400, triples)
    den.sample(obj_embed, triples, noise=None, n_steps=2, use_graph=use_graph)
    noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    sden.sample(uc, triples, noise1=noise1, n_steps=2, use_graph=use_graph)
    st, ss = next(iter(den._plans.values())), next(iter(sden._plans.values()))
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    st['plan'].sample(st['step'], 0, n, use_graph=use_graph)
    e[1].record()
    ss['plan'].sample(ss['step'], 0, n, use_graph=use_graph)
    e[2].record()
    torch.cuda.synchronize()
    tl, tsh = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    dom = time_dominant_kernel(ss, dev, reps=1)
    flops = ss['plan'].flops
    rec = {'config': 'configs[4] rank shape: %d scenes x %d nodes = %d objects on ONE GPU (1/8 of the batch-64 run; scenes are '
                     'independent, no per-step collective)' % (scenes, O, scenes * O),
           'metric': 'scene-steps/s (full step)', 'value': round(scenes * 1e3 / (tl + tsh), 3), 'ms_per_step': round(tl + tsh, 3),
           'layout_ms': round(tl, 3), 'shape_ms': round(tsh, 3), 'steps_timed': n,
           'shape_TFLOPs': round(flops / (tsh * 1e-3) / 1e12, 1),
           'roofline': None if dom is None else {'bound': 'mfma', 'kernel': 'k_conv_ws', 'achieved': round(dom[0], 1),
                                                 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(dom[0] / MFMA_F16_PEAK_TFLOPS, 4),
                                                 'avg_launch_us': round(dom[1], 1), 'launches_per_step': dom[2],
                                                 'whole_shape_step_TFLOPs': round(flops / (tsh * 1e-3) / 1e12, 1)}}
    del den, sden, st, ss
    torch.cuda.empty_cache()
    return rec


def cpu_baseline_shape(df, uc, triples, O_sample, timed=2):
    """1 warm-up + ``timed`` DDIM steps of the CPU oracle on the first O_sample objects (cost is linear in objects);
    returns seconds per step."""
    from oracle import echoscene_oracle as orc
    from echoscene_amd import synth
    sd = {k[len('diffusion_net.'):]: v.detach() for k, v in df.state_dict().items()}
    keep = (triples[:, 0] < O_sample) & (triples[:, 2] < O_sample)
    tri = triples[keep]
    orc.shape_sample_loop(sd, uc[:O_sample], tri, synth.shape_noise(seed=7), S=100, n_steps=1)       # warm-up
    t0 = time.perf_counter()
    orc.shape_sample_loop(sd, uc[:O_sample], tri, synth.shape_noise(seed=7), S=100, n_steps=timed)
    return (time.perf_counter() - t0) / timed


def cpu_baseline_layout(net, obj_embed, triples, O, budget_s=15.0):
    """Times the CPU oracle (torch fp32, the current thread count) on a bounded sample of the same workload."""
    from oracle import echoscene_oracle as orc
    from echoscene_amd import synth
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    noise = synth.layout_noise(O, 8, 1000, seed=7)
    orc.layout_sample_loop(sd, obj_embed, triples, noise, 1000, n_steps=2)       # warm-up
    n = 0
    t0 = time.perf_counter()
    while True:
        orc.layout_sample_loop(sd, obj_embed, triples, noise, 1000, n_steps=5)
        n += 5
        if time.perf_counter() - t0 > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t0
    return n / dt, n


def cpu_baseline(net, obj_embed, triples, O, full, df=None, uc=None):
    """The CPU oracle on the box's host cores, SURVEY 8(d) protocol.  Round 5: the thread count is SWEPT upwards over
    {8, 16, 32, 64, all logical CPUs} until more threads are clearly worse (or 75 s are spent) -- a quick pass per count (layout: a bounded loop after a warm-up; shape: 1 warm-up + 1 timed DDIM
    step on 4 of the O objects) -- and the best count is then timed properly: shape = 1 warm-up + 2 timed DDIM steps on 8 objects, scaled
    by O / 8 (the per-object cost ratio O = 8 / O = 4 is recorded).  ``value`` / ``cores`` are those of the best thread count."""
    ncores = os.cpu_count() or torch.get_num_threads()
    default_threads = torch.get_num_threads()
    sweep = sorted({t for t in (8, 16, 32, 64, ncores) if t <= ncores} | {min(8, ncores)})
    key = 'full_steps_per_s_estimate' if full else 'layout_steps_per_s'
    res, not_run = {}, []
    t_sweep = time.perf_counter()
    for nt in sweep:
        # the sweep ends as soon as more threads are clearly worse (torch's CPU kernels thrash beyond 16-32 threads on these hosts:
        # 256 threads measured 0.001 steps/s, minutes per shape step) or its wall-clock budget is spent -- the default run of this
        # script must finish within minutes on any box; the counts not run are listed
        if res and (res[max(res)][key] < 0.7 * max(r[key] for r in res.values()) or time.perf_counter() - t_sweep > 75.0):
            not_run.append(nt)
            continue
        torch.set_num_threads(nt)
        v, n = cpu_baseline_layout(net, obj_embed, triples, O, budget_s=2.5 if full else 4.0)
        r = {'threads': nt, 'layout_steps_per_s': round(v, 3), 'layout_steps_timed': n}
        if full:
            ts = cpu_baseline_shape(df, uc, triples, 4, timed=1)
            r['shape_s_per_step_O4_quick'] = round(ts, 3)
            r['full_steps_per_s_estimate'] = round(1.0 / (1.0 / v + ts * (O / 4)), 5)
        res[nt] = r
    best = max(res.values(), key=lambda r: r[key])
    nt = best['threads']
    torch.set_num_threads(nt)
    v, n = cpu_baseline_layout(net, obj_embed, triples, O, budget_s=6.0 if full else 10.0)
    final = {'threads': nt, 'layout_steps_per_s': round(v, 3), 'layout_steps_timed': n}
    value = v
    Os = 8
    if full:
        ts8 = cpu_baseline_shape(df, uc, triples, Os, timed=2)
        final['shape_s_per_step_O%d' % Os] = round(ts8, 3)
        final['per_object_cost_ratio_O8_vs_O4'] = round(ts8 / (2 * best['shape_s_per_step_O4_quick']), 3)
        value = 1.0 / (1.0 / v + ts8 * (O / Os))
        final['full_steps_per_s'] = round(value, 5)
    torch.set_num_threads(default_threads)
    out = {'value': round(value, 5), 'unit': 'steps/s', 'cores': nt, 'kind': 'port', 'host_logical_cpus': ncores,
           'sample': ('layout: bounded loop of the same %d-node graph after a 2-step warm-up; ' % O) +
                     (('shape: 1 warm-up + 2 timed DDIM steps on %d of the %d objects, scaled x%d (cost is linear in objects); '
                       % (Os, O, O // Os)) if full else '') +
                     'torch-CPU oracle fp32 at the best thread count of the sweep below (by_threads: the quick pass per count)',
           'measured_at_best': final, 'by_threads': [res[k] for k in sorted(res)],
           'thread_counts_not_run': not_run}
    if full:
        out.update({'objects_timed': Os, 'objects_of_workload': O, 'extrapolated': True,
                    'extrapolation': 'shape step timed on %d objects and scaled x%d; SURVEY.md 8(d) probe of the reference itself at O = 32 on 8 '
                                     'threads of another host: 57.3 s per shape step (0.0175 steps/s)' % (Os, O // Os)})
    if 8 in res:
        out['cores_8_quick'] = res[8][key]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--nodes', type=int, default=32)
    ap.add_argument('--workload', default='full', choices=['full', 'layout'])
    ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'])
    ap.add_argument('--scenes-per-gpu', type=int, default=2,
                    help='weak scaling: scenes per GPU (configs[4] is 8 per GPU; 2 keeps the default run inside ~20 GB)')
    ap.add_argument('--no-sub-records', action='store_true')
    ap.add_argument('--deterministic', action='store_true',
                    help='the CANONICAL arithmetic (ShapeDenoiser(deterministic=True)): split-K / GroupNorm tiling of a 4-object reference '
                         'shard on every rank of every world size, 1 included -- the latents are the same BIT FOR BIT for N = 1, 2, 4, 8 '
                         '(SURVEY.md 8(e)).  Default (off, round 6): every rank tunes the tiling to its own share of the objects; at 4 '
                         'objects per GPU the two modes coincide')
    ap.add_argument('--tuned', action='store_true', help='(the default since round 6; kept for old command lines)')
    ap.add_argument('--fuse-loops', type=int, default=-1,
                    help='1: one hipGraph per full step with the layout step as a parallel branch of the shape step; 0: two streams; '
                         '-1: the default of this build')
    ap.add_argument('--reps', type=int, default=5, help='repetitions of the timed K-step region (value = the median repetition)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-mode', action='store_true', help='N > 1: skip the record of the other sharding mode (other_sharding_mode)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--check', action='store_true',
                    help='seeded initial latents (the same for every world size) and a CRC32 of the final latents of the whole scene in the '
                         'JSON line ("check"): with --deterministic the N-rank run reproduces the 1-rank CRC (of a --deterministic run) bit for bit')
    a = ap.parse_args()

    backend = os.environ.get('ES_DIST_BACKEND', 'nccl')     # 'gloo' only for single-GPU multi-rank smoke tests
    if a.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` (the documented contract, line 4 of this file): launched without a rendezvous environment, so this
        # process becomes the launcher -- one rank per GPU under torch.distributed.run on the loopback address, same arguments.  (Until
        # round 5 --gpus was parsed and never read: the plain-python form silently ran ONE rank and printed n_gpus 1.)
        ndev = torch.cuda.device_count()
        if backend == 'nccl' and ndev < a.gpus:
            sys.exit('bench.py: --gpus %d asked for, %d visible device(s) (one rank per GPU over RCCL; ES_DIST_BACKEND=gloo lets several '
                     'ranks share a device for smoke tests)' % (a.gpus, ndev))
        import socket, subprocess
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus != world:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE is %d: launch `python bench.py --gpus N` (it starts the ranks itself) or '
                 '`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`' % (a.gpus, world))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    O = a.nodes
    full = a.workload == 'full'
    weak = a.scaling == 'weak'
    scenes_local = a.scenes_per_gpu if weak else 1
    scenes = scenes_local * world if weak else 1
    # strong: ONE scene, objects block-partitioned over the ranks; the layout branch (1 % of
    #         the work, couples all nodes of a scene every step) is replicated on every rank.
    # weak:   every rank owns `scenes_local` whole scenes, collated into one block-diagonal graph of its own: both loops
    #         run rank-locally, nothing is exchanged during the steps.
    from echoscene_amd import synth
    if weak:
        graphs = [synth.synthetic_graph(O, seed=100 + rank * scenes_local + s) for s in range(scenes_local)]
        _, triples = synth.collate_graphs(graphs)
        net, den, _, _ = build_layout(dev, O, seed=100)
        obj_embed = torch.randn(O * scenes_local, 640, generator=torch.Generator().manual_seed(100 + rank))
    else:
        net, den, obj_embed, triples = build_layout(dev, O, seed=100)
    triples_all = triples
    O_all = O * scenes_local
    sh_rank, sh_world = (0, 1) if weak else (rank, world)
    use_graph = not a.no_graph
    # untimed warm-up (also builds the plans and captures the graphs)
    den.sample(obj_embed, triples, noise=None, n_steps=max(a.warmup, 1), use_graph=use_graph)
    st = next(iter(den._plans.values()))
    st['noise'].normal_()
    st['x'].copy_(st['noise'][0])
    if full:
        df, sden, uc = build_shape(dev, O_all, 100, triples_all, sh_rank, sh_world, deterministic=a.deterministic)
        noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
        sden.sample(uc, triples_all, noise1=noise1, n_steps=max(min(a.warmup, 3), 1), use_graph=use_graph)
        ss = next(iter(sden._plans.values()))
        ss['x'].normal_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    want_fuse = full and (FUSE_DEFAULT if a.fuse_loops < 0 else a.fuse_loops == 1)
    fuse = want_fuse and sh_world == 1
    fused = None
    fused_sharded = False
    if want_fuse and sh_world > 1 and not ss.get('empty'):
        # sharded scene: a DDIM step is stem graph -> all-gather -> main graph; the (replicated) layout step rides on the main
        # graph as a parallel branch, exactly as in the single-GPU case (two streams serialise: section 5 of DESIGN.md)
        from echoscene_amd.plan import combine_plans
        ss['main_plan_alone'] = ss['main_plan']
        ss['main_plan'] = combine_plans(dev, ss['main_plan_alone'], st['plan'])
        ss.pop('step_graph', None)               # (a captured step graph of the warm-up holds the old main plan)
        st['step'].zero_()
        ss['main_plan'].sample(ss['step'], 0, 1, use_graph=use_graph)      # capture outside the timed region
        st['step'].zero_()
        torch.cuda.synchronize()
        fused_sharded = True
    if fuse:
        # ONE hipGraph per full step: the layout step is a parallel branch of the shape step (plan.combine_plans), so its 131
        # small launches run in the gaps of the shape step's kernels instead of after them
        from echoscene_amd.plan import combine_plans
        fused = combine_plans(dev, ss['plan'], st['plan'])
        st['step'].zero_()
        fused.sample(ss['step'], 0, 2, use_graph=use_graph)          # capture outside the timed region
        st['step'].zero_()
        torch.cuda.synchronize()
    # The two loops are independent given the setup (the reference runs them back to back); they are enqueued on
    # two HIP streams so the latency-bound layout chain (few CUs busy) overlaps the MFMA-bound shape loop.
    if full and a.check:                        # (after every capture run above: those advance the state)
        xall = torch.randn(O_all, 3, 16, 16, 16, generator=torch.Generator().manual_seed(11))
        ss['x'].copy_(xall[ss.get('lo', 0):ss.get('hi', O_all)])
        torch.cuda.synchronize()
    s_lay, s_shp = torch.cuda.Stream(), torch.cuda.Stream()

    def reset_state():
        """fresh noise states before every repetition (outside the timed bracket): a loop restarted at iteration 0 on the previous
        repetition's RESULT treats a clean sample as noise, and on random weights a few such restarts overflow"""
        if a.check:
            return                               # (--check seeds the latents itself, one repetition)
        st['noise'].normal_()
        st['x'].copy_(st['noise'][0])
        if full:
            ss['x'].normal_()

    def timed_region():
        """EXACTLY a.steps steps, bracketed by a barrier + device synchronisation on both sides; returns (wall s, layout-loop ms,
        shape-loop ms) -- in the fused case both event pairs bracket the fused loop."""
        reset_state()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        if fused is not None:
            with torch.cuda.stream(s_shp):
                ev[0].record(); ev[2].record()
                done = 0
                while done < a.steps:
                    n = min(a.steps - done, sden.S)
                    fused.sample(ss['step'], 0, n, use_graph=use_graph)
                    done += n
                ev[1].record(); ev[3].record()
        with torch.cuda.stream(s_lay):
            ev[0].record() if fused is None else None
            done = a.steps if (fused is not None or fused_sharded) else 0
            while done < a.steps:                   # the layout loop is 1000 iterations long; K may exceed it
                n = min(a.steps - done, den.T)
                st['plan'].sample(st['step'], 0, n, use_graph=use_graph)
                done += n
            ev[1].record() if fused is None else None
        with torch.cuda.stream(s_shp):
            ev[2].record() if fused is None else None
            done = a.steps if fused is not None else 0
            while full and done < a.steps:          # the DDIM loop is 100 iterations long
                n = min(a.steps - done, sden.S)
                if sh_world == 1:
                    ss['plan'].sample(ss['step'], 0, n, use_graph=use_graph)
                else:                               # per-step echo all-gather over RCCL (parallel.sharded_ddim_loop)
                    from echoscene_amd.parallel import sharded_ddim_loop
                    sden._cur, sden._use_graph = ss, use_graph
                    sharded_ddim_loop(sden, O_all, n, sh_world)
                done += n
            ev[3].record() if fused is None else None
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        w = time.perf_counter() - t0
        tm = torch.tensor([w], device=dev if backend == 'nccl' else 'cpu')
        if dist is not None:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)          # max over ranks
        return float(tm.item()), ev[0].elapsed_time(ev[1]), (ev[2].elapsed_time(ev[3]) if full else 0.0)

    # The K-step region is timed a.reps times (each repetition with its own barrier / synchronise bracket); `value` is the MEDIAN
    # repetition, min / max go into the line next to it (boxes and runs differ by a few per cent: VERDICT r3 #5).
    reps = 1 if a.check else max(1, a.reps)
    regions = [timed_region() for _ in range(reps)]
    walls = sorted(r[0] for r in regions)
    wall = walls[len(walls) // 2] if len(walls) % 2 else 0.5 * (walls[len(walls) // 2 - 1] + walls[len(walls) // 2])
    mid = min(regions, key=lambda r: abs(r[0] - wall))
    lay_ms, shp_ms = mid[1], mid[2]
    check = None
    if full and a.check:                        # (before the per-loop replays below advance the state again)
        import zlib
        z = ss['x'].detach().cpu().contiguous()
        if dist is not None and not weak:
            parts = [None] * world
            dist.all_gather_object(parts, z)
            z = torch.cat(parts, 0)
        check = {'latents_crc32': zlib.crc32(z.numpy().tobytes()), 'objects': int(z.shape[0]),
                 'abs_sum': float(z.double().abs().sum())}
    lay_reps, shp_reps = [r[1] / a.steps for r in regions], [r[2] / a.steps for r in regions]
    if fused is not None:
        # per-loop figures for the record (outside the timed region): each loop alone on the idle GPU, `reps` times each
        # (the layout loop: 20 untimed steps first, then up to 400 timed ones -- 50 steps right behind a shape loop measured the first
        #  replays of its graph on a GPU still clocked for the matrix work: 0.786 ms where `--workload layout` measures 0.666)
        nn = min(a.steps, 50)
        nl = max(1, min(max(a.steps, 1) * 4, 400, den.T - 20))
        lay_reps, shp_reps = [], []
        for _ in range(reps):
            reset_state()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            st['plan'].sample(st['step'], 0, 20, use_graph=use_graph)
            e[0].record(); st['plan'].sample(st['step'], 20, nl, use_graph=use_graph)
            e[1].record()
            e[2].record(); ss['plan'].sample(ss['step'], 0, nn, use_graph=use_graph)
            e[3].record(); torch.cuda.synchronize()
            lay_reps.append(e[0].elapsed_time(e[1]) / nl)
            shp_reps.append(e[2].elapsed_time(e[3]) / nn)
        solo = (_stats(lay_reps)['median'] * a.steps, _stats(shp_reps)['median'] * a.steps)
    if fused_sharded:
        # the layout steps ran inside the sharded main graphs; for the record (outside the timed region): the layout loop alone
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        nn = min(a.steps, 50)
        e[0].record(); st['plan'].sample(st['step'], 0, nn, use_graph=use_graph); e[1].record(); torch.cuda.synchronize()
        lay_ms = e[0].elapsed_time(e[1]) * a.steps / nn
        lay_reps = [lay_ms / a.steps]
    fused_ms = None
    if fused is not None:
        fused_ms, (lay_ms, shp_ms) = lay_ms, solo
    assert torch.isfinite(st['x']).all(), 'non-finite layout state'
    alt = None
    if full and sh_world > 1 and not a.check and not a.no_alt_mode:
        # The OTHER sharding mode on the same ranks, shape loop only, same K steps inside the same barrier bracket: the line's `value`
        # is the tuned default (or --deterministic); this record puts the other curve next to it, so that one multi-GPU run of the
        # driver shows both (round 6: the canonical arithmetic is that of a 4-object shard, so at N = 8 the two agree).
        from echoscene_amd.parallel import sharded_ddim_loop
        _, sden2, _ = build_shape(dev, O_all, 100, triples_all, sh_rank, sh_world, deterministic=not a.deterministic)
        sden2.sample(uc, triples_all, noise1=noise1, n_steps=2, use_graph=use_graph)
        ss2 = next(iter(sden2._plans.values()))
        ws2 = []
        for _ in range(max(1, min(reps, 3))):
            ss2['x'].normal_()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = 0
            while done < a.steps:
                n = min(a.steps - done, sden2.S)
                sden2._cur, sden2._use_graph = ss2, use_graph
                sharded_ddim_loop(sden2, O_all, n, sh_world)
                done += n
            torch.cuda.synchronize()
            dist.barrier()
            tm = torch.tensor([time.perf_counter() - t0], device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ws2.append(float(tm.item()))
        alt = {'mode': 'tuned shards (K splits chosen from the local object count)' if a.deterministic else 'bit-exact shards (--deterministic: the canonical K splits of a 4-object shard on every rank)',
               'what': 'shape loop only, %d steps, max over ranks, median of %d' % (a.steps, len(ws2)),
               'shape_ms_per_step': _stats([w * 1e3 / a.steps for w in ws2])}
        del sden2, ss2
    if full:
        assert torch.isfinite(ss['x']).all(), 'non-finite shape latent'

    if rank == 0:
        ms_per_step = wall * 1e3 / a.steps
        value = scenes * a.steps / wall           # scene-steps per second (scenes == 1 unless weak scaling)
        T = int(triples.shape[0])
        rep = {'wall_ms_per_step': _stats([w * 1e3 / a.steps for w in walls]),
               'value': _stats([scenes * a.steps / w for w in walls]),
               'layout_ms_per_step': _stats(lay_reps), 'shape_ms_per_step': _stats(shp_reps) if full else None}
        lay = {'steps_per_s': round(a.steps / (lay_ms * 1e-3), 2), 'ms_per_step': round(lay_ms / a.steps, 4),
               'ms_per_step_min_max': [rep['layout_ms_per_step']['min'], rep['layout_ms_per_step']['max']],
               'kernels_per_step': st['plan'].n_ops, 'launches_per_step': st['plan'].n_launches,
               'weight_bytes_per_step': st['plan'].weight_bytes,
               'hbm_GBps_algorithmic': round(st['plan'].weight_bytes / (lay_ms * 1e-3 / a.steps) / 1e9, 1)}
        if full:
            flops = ss['plan'].flops          # this rank's share of the step
            ach_step = flops / (shp_ms * 1e-3 / a.steps) / 1e12
            dom = time_dominant_kernel(ss, dev)
            ach, dom_us, dom_n, dom_st = dom if dom else (ach_step, None, 0, None)
            t_sum = lay_ms / a.steps + shp_ms / a.steps              # full_steps/s = 1 / (t_layout_step + t_shape_step)
            note = None
            if fused_ms is not None and ms_per_step < shp_ms / a.steps:
                note = ('the fused full step (%.3f ms) is shorter than the shape loop timed alone (%.3f ms): the two are separate '
                        'measurements, %d repetitions each, and run-to-run variation on one box (see `repetitions`) exceeds the ~1 ms '
                        'the layout branch adds' % (ms_per_step, shp_ms / a.steps, reps))
            out = {
                'metric': 'denoising steps/sec (layout+SDF) for 32-node scene-graph, 64^3 SDF (3x16^3 latent), '
                          'full step = one DDPM layout step + one DDIM shape step over all objects; `value` = the FUSED step '
                          '(both loops as one hipGraph per step, the layout step on a parallel branch: what the K timed steps ran); '
                          '`full_steps_per_s_sum` = SURVEY 8(d)\'s literal 1 / (t_layout_step + t_shape_step), each loop alone',
                'value': round(value, 4), 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': a.scaling,
                'value_min_max': [rep['value']['min'], rep['value']['max']],
                'full_steps_per_s_sum': round(1e3 / t_sum, 4),
                'repetitions': rep, 'note': note,
                'vs_baseline': None, 'dtype': 'f16 MFMA operands / f32 accumulate (shape UNet); f32 (layout, GCN)',
                'data': 'synthetic',
                'config': {'workload': 'EchoScene full (layout+SDF) %d-node synthetic graph (T=%d), 3x16^3 latent -> '
                                       '64^3 SDF, layout 1000-step DDPM + shape 100-step DDIM schedules; %s'
                                       % (O, T, ('configs[4] shape: %d scenes per GPU x %d GPU(s), partitioned by scene (block-diagonal '
                                                 'batch graph): no per-step collective' % (scenes_local, world)) if weak
                                          else ('1 scene (configs[3] when N > 1: objects sharded over %d GPU(s), echo all-gather of '
                                                '[O,64] codes every DDIM step over RCCL)' % world)),
                           'scenes': scenes, 'hip_graph': use_graph, 'deterministic_shards': a.deterministic if sh_world > 1 else None,
                           'deterministic': a.deterministic,   # ShapeDenoiser(deterministic=...): the canonical arithmetic, bit-exact across world sizes (opt-in since round 6)
                           'step_graph': os.environ.get('ES_STEP_GRAPH', '0') == '1',      # sharded DDIM step captured as ONE graph (opt-in)
                           'route_options': route_options_string(),
                           'loops': ('one hipGraph per full step, layout step as a parallel branch '
                                                                              '(%.3f ms per step); layout / shape below: each loop alone' % (fused_ms / a.steps))
                           if fused_ms is not None else ('layout step as a parallel branch of the sharded main graph' if fused_sharded else 'two HIP streams'), 'layout': lay,
                           'shape': {'steps_per_s': round(a.steps / (shp_ms * 1e-3), 3),
                                     'ms_per_step': round(shp_ms / a.steps, 3),
                                     'ms_per_step_min_max': [rep['shape_ms_per_step']['min'], rep['shape_ms_per_step']['max']],
                                     'kernels_per_step': ss['plan'].n_ops,
                                     'algorithmic_TFLOP_per_step': round(flops / 1e12, 3)}},
                'roofline': {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': round(ach / MFMA_F16_PEAK_TFLOPS, 4), 'traffic': pmc_traffic('k_conv_ws'), 'traffic_source': pmc_traffic_source(), 'kernel': 'k_conv_ws (3x3x3 SAME launches: its shared-A-tile variant k_conv_ws3)',
                             'launches_per_step': dom_n, 'avg_launch_us': None if dom_us is None else round(dom_us, 1),
                             'avg_launch_us_min_max': None if dom_st is None else [dom_st['min'], dom_st['max']],
                             'whole_shape_step_TFLOPs': round(ach_step, 1),
                             'note': 'achieved = algorithmic FLOPs of the k_conv_ws launches of one shape step / their '
                                     'duration (HIP events on the launch stream, launches replayed back to back); '
                                     'whole_shape_step = all FLOPs of the step / shape-step time (all kernels); traffic = HBM '
                                     'bytes per launch from the committed rocprofv3 --pmc passes of this command (traffic_source: file + git blob id; '
                                     'FETCH_SIZE x2 on gfx950 + WRITE_SIZE); null when that file is absent'},
            }
        else:
            ach = lay['hbm_GBps_algorithmic']
            out = {
                'metric': 'denoising steps/sec (layout box-denoiser loop, 32-node scene graph, 1000-step DDPM)',
                'value': round(value, 2), 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                'ms_per_step': round(ms_per_step, 5), 'higher_is_better': True, 'scaling': a.scaling,
                'value_min_max': [rep['value']['min'], rep['value']['max']], 'repetitions': rep,
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': 'configs[1]: EchoLayout box diffusion, %d-node synthetic graph (T=%d triples), '
                                       '1000-step DDPM, HIP denoiser + graph conv' % (O, T),
                           'scenes_per_gpu': 1, 'hip_graph': use_graph, 'layout': lay},
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic('k_rows_x'), 'traffic_source': pmc_traffic_source(), 'kernel': 'k_rows_x (the rows products of the layout step)'},
            }
        if world == 1 and not weak and not a.no_sub_records and full:
            out['sub_records'] = sub_records(dev, lay, use_graph, a)
            try:
                from echoscene_amd import config as escfg
                out['sub_records'].append(fp32x_route_record(dev, df, sden, uc, triples_all, escfg.shape_df_conf(224).model.params, use_graph))
                out['sub_records'].append(fp32_route_record(dev, df, sden, uc, triples_all, escfg.shape_df_conf(224).model.params, use_graph))
            except Exception as exc:                  # (a validation figure must not take the benchmark line down)
                out['sub_records'].append({'config': 'fp32-operand validation route', 'error': repr(exc)})
        if world == 1 and not weak and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(net, obj_embed, triples, O, full, df if full else None, uc if full else None)
        if check is not None:
            out['check'] = check
        if alt is not None:
            out['other_sharding_mode'] = alt
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
