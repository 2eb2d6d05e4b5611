"""Multi-GPU sharding of the denoising path: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

The path shards by IMAGE: the reference's default pipeline processes one latent per call, its std
group is that image's (uncond, cond) rows x heads, so images are independent and each rank runs its own images with a
full weight replica.  The only collectives are outside the step loop: one broadcast of the text embeddings (+ the
token ids / region tables, a few KB) from rank 0 before generation and an optional gather of the final latents
(32 KB per 512x512 image).  No per-step communication exists, so nothing overlaps with compute and bucket sizes do
not matter; with xGMI's point-to-point links a 236 KB broadcast is latency-bound (microseconds).
"""
import torch
import torch.distributed as dist


def shard_image_indices(n_images, rank, world):
    """image i -> rank i % world (round-robin keeps ranks balanced to within one image)"""
    return [i for i in range(n_images) if i % world == rank]


def broadcast_generation_inputs(*tensors, src=0):
    """broadcast text embeddings / id tensors / region tables from `src`; returns them (single tensor if one given)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)
    return tensors[0] if len(tensors) == 1 else tensors


def broadcast_region_state(region_state, device, src=0):
    """region tables {L: fp32 [Bw, L, S]} are built on rank `src` only (host-side rasterisation) and broadcast"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return region_state
    obj = [None]
    if dist.get_rank() == src:
        obj[0] = {int(L): tuple(t.shape) for L, t in region_state.items()} if isinstance(region_state, dict) else None
    dist.broadcast_object_list(obj, src=src)
    if obj[0] is None:
        return region_state if dist.get_rank() == src else torch.FloatTensor(0)
    out = {}
    for L, shape in sorted(obj[0].items()):
        t = region_state[L].to(device) if dist.get_rank() == src else torch.empty(shape, dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        out[L] = t
    return out


def gather_latents(latents, dst=0):
    """final latents of every rank -> list on `dst` (None elsewhere); equal per-rank counts assumed"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [latents]
    bufs = [torch.empty_like(latents) for _ in range(dist.get_world_size())] if dist.get_rank() == dst else None
    dist.gather(latents, bufs, dst=dst)
    return bufs
