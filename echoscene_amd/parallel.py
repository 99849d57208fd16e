"""Multi-GPU sharding of the shape branch: one process per GPU (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-GPU sampling path; this is new design
(SURVEY.md section 8(e)): objects of one scene are block-partitioned over the ranks.  Each object's
UNet3D forward depends on the other objects only through the 64-d conv-pool code of their current
latents that feeds the shape GCN ("echo" message passing, openai_model_3d.py:800-814), so the only
per-step exchange is an all-gather of [O_local, 64] floats (8 KB per step at O=32), after which
every rank runs the tiny GCN redundantly on the full graph.  At the end the latents (or decoded
SDFs) are all-gathered.  Every kernel treats objects independently and the GCN is computed on the full graph on every rank, so
the only thing that can differ between world sizes is WHERE a K sum is cut.  ``ShapeDenoiser(deterministic=True)``: the canonical
cuts of a 4-object reference shard on every rank of every world size -- the results are identical BIT FOR BIT for world = 1, 2, 4,
8, ....  ``deterministic=False`` (the default since round 6) lets each rank tune the cuts to its share: same values up to fp32
summation order; at 4 objects per GPU the two modes coincide.
"""
import torch


def partition(num_objects, world, rank):
    """Contiguous block partition with equal padded block size.  Returns (lo, hi, block).  With more ranks than blocks the
    trailing ranks get lo == hi (no objects): they run no voxel work but must still join every collective."""
    block = (num_objects + world - 1) // world
    lo = min(rank * block, num_objects)
    hi = min(lo + block, num_objects)
    return lo, hi, block


def all_gather_rows(local, num_rows, world, group=None, out=None):
    """All-gather row blocks of a [rows_local, C] tensor into [num_rows, C] (blocks padded to equal size).
    ``out`` ([world * block, C], pre-allocated) with ``local`` already a full zero-padded block makes the call
    allocation-free (the per-step echo exchange)."""
    import torch.distributed as dist
    block = (num_rows + world - 1) // world
    C = local.shape[1:]
    if out is not None and local.shape[0] == block:
        pad = local
    else:
        pad = torch.zeros((block,) + tuple(C), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        out = torch.empty((world * block,) + tuple(C), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) != 'nccl':
        # test-only path (several ranks sharing one GPU under gloo): stage through host memory
        outc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(outc, pad.cpu(), group=group)
        out.copy_(outc)
    else:
        dist.all_gather_into_tensor(out, pad, group=group)
    return out[:num_rows]


def sharded_ddim_loop(backend, num_objects, n_steps, world, group=None):
    """The DDIM loop with the per-step echo exchange.

    ``backend`` provides (HIP: samplers.ShapeDenoiser shard; tests: an oracle-based shard):
        codes_local(i)        -> [O_local, 64] conv-pool codes of this rank's current latents
        step(i, codes_all)    -> advance this rank's latents by DDIM iteration i given all objects' codes
        latents_local()       -> [O_local, C, D, H, W]
        gather_buffers()      -> optional: pre-allocated (send block, receive buffer) of the exchange
    Per step on the HIP path: graph launch (stem), all-gather of [block, 64] floats per rank, graph launch (rest) -- no allocation,
    no torch op in between; with ES_STEP_GRAPH=1 and every rank able to capture it (a collective decision, backend.step_graph)
    ONE captured graph = stem ops, the RCCL all-gather, everything else.  Returns the full latents [O, C, D, H, W] on every
    rank.  A rank without objects (more ranks than objects) still joins every collective."""
    exchange = world > 1 or getattr(backend, 'force_exchange', False)
    bufs = backend.gather_buffers() if (exchange and hasattr(backend, 'gather_buffers')) else None
    g = backend.step_graph(group) if (exchange and hasattr(backend, 'step_graph')) else None
    if g is not None:
        # one captured graph per step: stem -> RCCL all-gather -> main, no host work between the steps
        backend.begin(0)
        for i in range(n_steps):
            g.replay()
        zl = backend.latents_local()
        return all_gather_rows(zl, num_objects, world, group) if world > 1 else zl
    for i in range(n_steps):
        cl = backend.codes_local(i)
        if exchange:
            ca = all_gather_rows(cl, num_objects, world, group, out=bufs[1] if bufs else None)
        else:
            ca = cl
        backend.step(i, ca)
    zl = backend.latents_local()
    return all_gather_rows(zl, num_objects, world, group) if world > 1 else zl
