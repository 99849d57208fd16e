#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: denoising steps/sec of EchoScene's sampling loops.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A "step" is one denoising iteration over one synthetic 32-node scene graph (BASELINE.json /
SURVEY.md section 8(d)): with ``--workload layout`` one ``p_sample_sg`` of the 1000-step DDPM
box loop (BASELINE configs[1]); with ``--workload full`` one layout step + one DDIM shape step
over the same O objects (the metric's "layout+SDF" step; needs the volume path).
Inputs (weights, graph, noise tables) are resident in HBM before the timed region starts.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank samples its own
scene (the path shards over scenes/objects with no data-path collective -> weak scaling);
value = total steps of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def build_layout(dev, O, seed):
    from echoscene_amd import synth, config as escfg
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(512))
    synth.seeded_fill_(net, prefix='bench.layout.')
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    objs, triples = synth.synthetic_graph(O, seed=seed)
    rs = torch.Generator().manual_seed(seed)
    obj_embed = torch.randn(O, 640, generator=rs)
    return net, den, obj_embed, triples


def cpu_baseline_layout(net, obj_embed, triples, O, budget_s=15.0):
    """Times the CPU oracle (torch fp32, all host threads) on a bounded sample of the same workload."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--nodes', type=int, default=32)
    ap.add_argument('--workload', default='layout', choices=['layout'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    O = a.nodes
    net, den, obj_embed, triples = build_layout(dev, O, seed=100 + rank)
    use_graph = not a.no_graph
    # untimed warm-up (also builds the plan and captures the graph)
    den.sample(obj_embed, triples, noise=None, n_steps=max(a.warmup, 1), use_graph=use_graph)
    st = next(iter(den._plans.values()))
    st['noise'].normal_()
    st['x'].copy_(st['noise'][0])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    done = 0
    while done < a.steps:                       # the loop is 1000 iterations long; K may exceed it
        n = min(a.steps - done, den.T)
        st['plan'].sample(st['step'], 0, n, use_graph=use_graph)
        done += n
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall], device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    assert torch.isfinite(st['x']).all(), 'non-finite layout state'

    if rank == 0:
        ms_per_step = wall * 1e3 / a.steps
        value = world * a.steps / wall
        n_launch = st['plan'].n_ops
        wbytes = st['plan'].weight_bytes
        ach = wbytes / (dev_ms * 1e-3 / a.steps) / 1e9
        out = {
            'metric': 'denoising steps/sec (layout box-denoiser loop, 32-node scene graph, 1000-step DDPM)',
            'value': round(value, 2), 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: EchoLayout box diffusion, %d-node synthetic graph (T=%d triples), '
                                   '1000-step DDPM, HIP denoiser + graph conv' % (O, triples.shape[0]),
                       'scenes_per_gpu': 1, 'hip_graph': use_graph, 'kernels_per_step': n_launch},
            'roofline': {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': None,
                         'kernel': 'k_linear_rows', 'algorithmic_bytes_per_step': wbytes,
                         'launches_per_step': n_launch,
                         'avg_launch_us_incl_gaps': round(dev_ms * 1e3 / a.steps / n_launch, 3)},
        }
        if world == 1 and not a.no_cpu_baseline:
            v, n = cpu_baseline_layout(net, obj_embed, triples, O)
            out['cpu_baseline'] = {'value': round(v, 3), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'sample': '%d layout denoising steps of the same %d-node graph '
                                   '(torch-CPU oracle, fp32)' % (n, O)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
