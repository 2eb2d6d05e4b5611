"""world_size-2 gloo test of the multi-GPU plumbing (runs on CPU): image sharding, the embedding / region-table
broadcast and the latent gather.  The per-image arithmetic is rank-independent by construction (no per-step
collective), so what is checked is that every rank ends up with rank 0's inputs and the right image subset."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusionspatialcontrol_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        emb = torch.full((2, 77, 8), float(rank + 1))
        emb = parallel.broadcast_generation_inputs(emb, src=0)
        rs = {64: torch.full((2, 64, 77), 0.5), 16: torch.full((2, 16, 77), -0.25)} if rank == 0 else None
        rs = parallel.broadcast_region_state(rs, torch.device("cpu"), src=0)
        mine = parallel.shard_image_indices(5, rank, world)
        lat = torch.stack([torch.full((4, 2, 2), float(i)) for i in mine[:2]])
        got = parallel.gather_latents(lat, dst=0)
        q.put((rank, float(emb.mean()), sorted(rs.keys()), float(rs[64].mean()), float(rs[16].mean()), mine,
               None if got is None else [float(g.mean()) for g in got]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, emb_mean, keys, m64, m16, mine, gathered in res:
        assert emb_mean == 1.0                       # everyone holds rank 0's embeddings
        assert keys == [16, 64] and m64 == 0.5 and m16 == -0.25
    assert res[0][5] == [0, 2, 4] and res[1][5] == [1, 3]
    assert res[0][6] == [1.0, 2.0] and res[1][6] is None      # rank0 images {0,2} mean 1.0; rank1 images {1,3} mean 2.0


def test_single_process_is_a_noop():
    t = torch.ones(3)
    assert parallel.broadcast_generation_inputs(t) is t
    assert parallel.shard_image_indices(4, 0, 1) == [0, 1, 2, 3]
    assert parallel.gather_latents(t) == [t]
