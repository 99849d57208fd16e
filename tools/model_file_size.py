#!/usr/bin/env python
"""Size of the model files of the benchmarked scene (O = 32, shipped widths) against the packed weights they carry: since round 4
(ABI 5) activation scratch is listed, not stored ("O = 32 model files < 1.2x the packed-weight bytes").
usage: python tools/model_file_size.py [O]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import synth

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)
df, sden, uc = bench.build_shape(dev, O, 100, triples)
tmp = tempfile.mkdtemp(prefix='esm_')
for name, fn, w in (('layout', lambda p: den.save_model(p, obj_embed, triples), sum(p.numel() * 4 for p in net.parameters())),
                    ('shape', lambda p: sden.save_model(p, uc, triples), sum(p.numel() * 2 for p in df.parameters()))):
    path = os.path.join(tmp, name + '.esm')
    stored = fn(path)
    size = os.path.getsize(path)
    print('%s model file: %.1f MB on disk (%.1f MB of buffer contents); reference parameters at the stored precision: %.1f MB -> x%.2f'
          % (name, size / 2 ** 20, stored / 2 ** 20, w / 2 ** 20, size / w), flush=True)
    os.remove(path)
