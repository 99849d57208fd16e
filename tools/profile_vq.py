"""VQ-VAE decode of 32 latents (the once-per-sample epilogue of a scene): wall time; run under rocprofv3 --kernel-trace --stats for
the per-kernel split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd import synth, config as escfg
from echoscene_amd.model.vqvae import VQVAE
from echoscene_amd.samplers import VQDecoder
c = escfg.vqvae_conf(64).model.params
vq = VQVAE(dict(c.ddconfig), 8192, c.embed_dim)
synth.seeded_fill_(vq, prefix='vqvae_full.')
dec = VQDecoder(vq, torch.device('cuda'))
z = torch.randn(32, 3, 16, 16, 16, device='cuda') * 0.6
dec.decode_no_quant(z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    dec.decode_no_quant(z)
torch.cuda.synchronize()
print('decode of 32 objects: %.1f ms' % ((time.perf_counter() - t0) / 3 * 1e3))
