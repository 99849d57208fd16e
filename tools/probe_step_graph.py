"""One-GPU check of the captured sharded DDIM step (stem -> RCCL all-gather -> main as ONE graph): a 1-rank NCCL process group
on this GPU, the sharded step structure forced at world == 1 (ShapeDenoiser(force_exchange=True)).  Prints STEP_GRAPH_OK when
the exchange was captured and the latents equal the ordinary single-graph run bit for bit.  usage: python tools/probe_step_graph.py [mc]"""
import os, sys, socket
os.environ['ES_STEP_GRAPH'] = '1'          # the captured exchange is opt-in (samplers.ShapeDenoiser.step_graph)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from echoscene_amd import synth, config as escfg
from echoscene_amd.model.unet import DiffusionUNet
from echoscene_amd.samplers import ShapeDenoiser

mc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
p = escfg.shape_unet_params(mc)
ctx = 64 if mc < 224 else 1280
p['context_dim'] = ctx
df = DiffusionUNet(p)
synth.seeded_fill_(df, prefix='stepgraph.')
O = 5
objs, triples = synth.synthetic_graph(O, seed=12)
uc = torch.randn(O, 1, ctx, generator=torch.Generator().manual_seed(3))
noise1 = synth.shape_noise(seed=7)
mpar = escfg.shape_df_conf().model.params
ref = ShapeDenoiser(df, mpar, ddim_steps=4, device=dev).sample(uc, triples, noise1)
den = ShapeDenoiser(df, mpar, ddim_steps=4, device=dev, force_exchange=True)
z = den.sample(uc, triples, noise1)
st = next(iter(den._plans.values()))
captured = st.get('step_graph') is not None
z2 = den.sample(uc, triples, noise1)                 # replays the cached step graph
print('captured exchange: %s; max |z - ref| = %.3e; replay equal: %s' % (captured, (z - ref).abs().max().item(), torch.equal(z, z2)))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    den.sample(uc, triples, noise1)
torch.cuda.synchronize(); t1 = time.perf_counter()
print('sharded-structure loop: %.3f ms per DDIM step (captured=%s)' % ((t1 - t0) * 1e3 / 20, captured))
if captured and torch.equal(z, ref) and torch.equal(z, z2):
    print('STEP_GRAPH_OK')
dist.destroy_process_group()
