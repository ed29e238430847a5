"""world_size-2 CPU (gloo) coverage of the multi-process path: the one collective of the path (flat checkpoint
broadcast from rank 0) and the rank sharding of clips keyed by GLOBAL clip index."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from tests.helpers import clip_batch
    from versband_amd import synth
    shapes = [synth.hifigan_shapes(synth.HifiGanConfig(upsample_initial_channel=32)),
              {"a.weight": ((3, 5), ("u", 1.0)), "b.bias": ((7,), ("norm",))}]
    if rank == 0:
        sds = [synth.make_state_dict(s, 11 + i) for i, s in enumerate(shapes)]
    else:
        sds = [{k: torch.empty(shp) for k, (shp, _) in s.items()} for s in shapes]
    got = bench.broadcast_state(sds, rank, world, torch.device("cpu"))
    ref = [synth.make_state_dict(s, 11 + i) for i, s in enumerate(shapes)]
    ok = all(torch.equal(got[i][k], ref[i][k]) for i in range(2) for k in ref[i])
    # clip sharding: rank r owns global clips [r*B, (r+1)*B)
    B, T = 2, 8
    mine = clip_batch(B, T, 4, clip0=rank * B)
    torch.save({"ok": ok, "x": mine["x_latent"], "midi": mine["midi"]}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, nprocs=world, args=(world, port, str(tmp_path)), join=True)
    from tests.helpers import clip_batch
    whole = clip_batch(4, 8, 4, clip0=0)
    parts = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert all(p["ok"] for p in parts), "weights differ after broadcast"
    assert torch.equal(torch.cat([p["x"] for p in parts]), whole["x_latent"])      # shards == single-process batch, bitwise
    assert torch.equal(torch.cat([p["midi"] for p in parts]), whole["midi"])
