#!/usr/bin/env python
"""Experiment: run the layout loop and the shape loop CONCURRENTLY on CU-masked HIP streams
(hipExtStreamCreateWithCUMask).  The conv kernels take every VGPR of a CU, so without a mask the latency-bound
layout kernels only get CUs at kernel boundaries and the two loops serialise.
usage: python tools/probe_cu_mask.py [--layout-cus 16] [--steps 100] [--layout-steps 1000]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--layout-cus', type=int, default=16)
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--layout-steps', type=int, default=1000)
a = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
hiprt = C.CDLL('libamdhip64.so')


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) enabled for the stream"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= (1 << (b % 32))
    st = C.c_void_p()
    rc = hiprt.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


O = 32
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)
den.sample(obj_embed, triples, noise=None, n_steps=3)
st = next(iter(den._plans.values()))
df, sden, uc = bench.build_shape(dev, O, 100, triples, 0, 1)
sden.sample(uc, triples, noise1=torch.randn(1, 3, 16, 16, 16, device=dev), n_steps=2)
ss = next(iter(sden._plans.values()))
torch.cuda.synchronize()


def run(s_lay, s_shp, tag):
    st['noise'].normal_(); st['x'].copy_(st['noise'][0]); ss['x'].normal_()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    with torch.cuda.stream(s_shp):
        ev[2].record(); ss['plan'].sample(ss['step'], 0, a.steps); ev[3].record()
    with torch.cuda.stream(s_lay):
        ev[0].record(); st['plan'].sample(st['step'], 0, a.layout_steps); ev[1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('%-44s wall %.3f s   shape %.3f s (%.2f ms/step)   layout %.3f s (%.3f ms/step)' % (
        tag, wall, ev[2].elapsed_time(ev[3]) / 1e3, ev[2].elapsed_time(ev[3]) / a.steps,
        ev[0].elapsed_time(ev[1]) / 1e3, ev[0].elapsed_time(ev[1]) / a.layout_steps), flush=True)


run(torch.cuda.Stream(), torch.cuda.Stream(), 'two plain streams')
n = a.layout_cus
# hypothesis A: mask bit i -> XCD (i % 8): the first n bits are n/8 CUs of every XCD
lay_bits = list(range(n))
run(masked_stream(lay_bits), masked_stream([b for b in range(256) if b not in lay_bits]), 'masked: layout = bits 0..%d' % (n - 1))
# hypothesis B: contiguous per XCD (bit i -> XCD i // 32): take n/8 CUs from each block of 32
lay_bits = [x * 32 + j for x in range(8) for j in range(n // 8)]
run(masked_stream(lay_bits), masked_stream([b for b in range(256) if b not in lay_bits]), 'masked: layout = %d bits per block of 32' % (n // 8))
run(masked_stream(lay_bits), torch.cuda.Stream(), 'layout masked, shape unmasked')


def run_interleaved(s_lay, s_shp, tag, chunk=1):
    """one host thread feeds both queues in proportion (the runtime blocks the host when a queue is full)"""
    st['noise'].normal_(); st['x'].copy_(st['noise'][0]); ss['x'].normal_()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ratio = a.layout_steps // a.steps
    t0 = time.perf_counter()
    with torch.cuda.stream(s_shp):
        ev[2].record()
    with torch.cuda.stream(s_lay):
        ev[0].record()
    for i in range(0, a.steps, chunk):
        with torch.cuda.stream(s_shp):
            ss['plan'].sample(ss['step'], i, chunk)
        with torch.cuda.stream(s_lay):
            st['plan'].sample(st['step'], i * ratio, chunk * ratio)
    with torch.cuda.stream(s_shp):
        ev[3].record()
    with torch.cuda.stream(s_lay):
        ev[1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('%-44s wall %.3f s (host enqueue %.3f)  shape %.3f s   layout %.3f s' % (
        tag, wall, t_enq, ev[2].elapsed_time(ev[3]) / 1e3, ev[0].elapsed_time(ev[1]) / 1e3), flush=True)


run_interleaved(torch.cuda.Stream(), torch.cuda.Stream(), 'interleaved launches, plain streams')
run_interleaved(torch.cuda.Stream(), torch.cuda.Stream(), 'interleaved launches, chunk 5', chunk=5)
run_interleaved(torch.cuda.Stream(priority=-1), torch.cuda.Stream(), 'interleaved, layout stream high priority')
