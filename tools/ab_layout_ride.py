"""Same-box A/B of the layout step's launch grouping (configs[1]: 32 nodes, model_channels 512): plan.ROWS_RIDE = 0 / 1 / 2 (the head
of the trunk riding on the GCN chain's launches) in ONE process, each mode with its own plan and captured graph; median of `reps`
timings of `steps` replayed steps.  The final boxes of a seeded 50-step run must be BIT-identical across the modes.
ES_ROWS_U1 (read once per process by the library: 0 / 1 / 2) selects the two-workgroups-per-CU variants -- run the tool once per value.
usage: python tools/ab_layout_ride.py [steps] [reps] [modes, e.g. 2,2]
The CRC32 of the seeded 50-step result is printed so that runs of different processes (ES_ROWS_U1 / ES_ROWS_NT2 values) can be compared."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from echoscene_amd import plan, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = [int(m) for m in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2, 0, 1, 2]
dev = torch.device('cuda')
O = 32
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)
noise = synth.layout_noise(O, 8, 50)
ref = None
print('ES_ROWS_U1=%s ES_ROWS_NT2=%s' % (os.environ.get('ES_ROWS_U1', '(default 1)'), os.environ.get('ES_ROWS_NT2', '(default 1)')))
for mode in modes:
    plan.ROWS_RIDE = mode
    den._plans.clear()
    x = den.sample(obj_embed, triples, noise=noise, n_steps=50, use_graph=True).cpu()
    if ref is None:
        ref = x
    same = torch.equal(x, ref)
    import zlib
    crc = zlib.crc32(x.numpy().tobytes())
    st = next(iter(den._plans.values()))
    ts = []
    for _ in range(reps):
        st['noise'].normal_()
        st['x'].copy_(st['noise'][0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st['plan'].sample(st['step'], 0, steps, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    print('ride=%d  ops=%d  step %.1f us (min %.1f max %.1f)  %.1f steps/s  bit-identical to the first mode: %s  crc %08x'
          % (mode, st['plan'].n_ops, statistics.median(ts), min(ts), max(ts), 1e6 / statistics.median(ts), same, crc), flush=True)
    assert same
