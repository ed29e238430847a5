"""Multi-GPU side of the path: one process per GPU, clips sharded by GLOBAL clip index, and the ONE collective of the whole path -
a flat broadcast of rank 0's checkpoints over RCCL / xGMI.

The reference shards items with DistributedSampler(shuffle=False) under an NCCL process group and issues no collective at all: every
rank reads the checkpoints from disk itself (scripts/test_final.py:351-357, 467-477).  north_star replaces the N disk reads by one
broadcast ("RCCL broadcast of weights over xGMI only"); nothing is exchanged per step, per clip or at the end.

    init(rank, world, device)                 process group: "nccl" (= RCCL on ROCm) one rank per GPU; "gloo" on CPU or when all ranks of
                                              a functional test share one device (VB_ONE_DEVICE / VB_BENCH_ONE_DEVICE)
    shard_indices(n_items, rank, world)       the items rank r generates (rank::world, DistributedSampler(shuffle=False) order)
    rank0_guarded(fn, what)                   rank 0 runs a loader, every rank learns whether it worked before the data collectives start
    broadcast_state(state, src, device)       state dict(s) from rank `src` to every rank: tensors are packed per dtype into flat
                                              buffers (one broadcast each - the fp32 checkpoints of this path make ONE), rebuilt as views;
                                              returns (state, info) with the bytes moved, the time and a checksum comparison across ranks
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Sequence, Tuple, Union  # noqa: F401

import torch

State = Dict[str, torch.Tensor]


def one_device() -> bool:
    """functional tests of the N > 1 code path on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL"""
    return bool(os.environ.get("VB_ONE_DEVICE") or os.environ.get("VB_BENCH_ONE_DEVICE"))


def init(rank: int, world: int, device: torch.device, master_addr: str = "127.0.0.1", master_port: int = 54189) -> str:
    """join the process group (no-op for world == 1); returns the backend name.  The rendezvous defaults mirror the reference's
    tcp://localhost:54189 (scripts/test_final.py:351-353); MASTER_ADDR / MASTER_PORT from a launcher win."""
    import torch.distributed as dist
    if world <= 1:
        return "none"
    if dist.is_initialized():
        return dist.get_backend()
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port))
    if device.type == "cuda" and not one_device():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        return "nccl"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return "gloo"


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """DistributedSampler(shuffle=False) without its padding: rank r takes items r, r + world, ... (scripts/test_final.py:356)"""
    return list(range(n_items))[rank::world]


def rank0_guarded(fn, what: str = "load"):
    """Run fn() on rank 0 only and tell every rank whether it worked BEFORE the data collectives: a checkpoint that fails to load on
    rank 0 would otherwise leave the other ranks blocked in the broadcast until the process-group timeout.  Rank 0 passes the loader and
    gets its result; the other ranks pass None and get None; on failure every rank raises (rank 0 re-raises the original error)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not multi:
        return fn() if fn is not None else None
    result, err = None, None
    if dist.get_rank() == 0:
        try:
            result = fn()
        except BaseException as e:          # noqa: BLE001 - reported to the other ranks, then re-raised here
            err = e
    box = [None if err is None else f"{type(err).__name__}: {err}"]
    dist.broadcast_object_list(box, src=0)
    if box[0] is not None:
        if err is not None:
            raise err
        raise RuntimeError(f"rank 0 failed to {what}: {box[0]}")
    return result


def broadcast_state(state: Union[State, Sequence[State]], src: int = 0, device: torch.device = torch.device("cpu"),
                    check: bool = True, force: bool = False) -> Tuple[Union[State, List[State]], dict]:
    """Every rank returns rank `src`'s tensors, bit for bit.  Ranks other than `src` may pass None / empty dicts: names, shapes and
    dtypes travel first (one small object broadcast), then one flat buffer per dtype.  With world == 1 (or no process group) the
    state comes back unchanged.  check: every rank folds its received bytes into a 64-bit sum and the sums are compared (all_gather) -
    a broadcast that left any rank with different weights raises instead of generating different audio on that rank.
    force: run the collectives even in a process group of ONE rank (tests/test_gpu_multirank.py: the only way to execute the RCCL code
    path - communicator set-up, device-buffer broadcast, all_gather of the checksums - on a 1-GPU box)."""
    import torch.distributed as dist
    single = isinstance(state, dict)
    states: List[State] = [state] if single else [s or {} for s in (state or [])]
    info = {"bytes": 0, "ms": 0.0, "buffers": 0, "checked": False, "backend": None}
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return state, info
    rank = dist.get_rank()
    info["backend"] = dist.get_backend()
    # names, shapes, dtypes - and whether rank src passed one dict or a list of them (the other ranks pass None)
    meta = [[(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items() if torch.is_tensor(v)] for sd in states] if rank == src else None
    # entries that are not tensors (python scalars / strings a checkpoint may carry) travel inside the same object broadcast
    extra = [{k: v for k, v in sd.items() if not torch.is_tensor(v)} for sd in states] if rank == src else None
    box = [(single, meta, extra)]
    dist.broadcast_object_list(box, src=src)
    single, meta, extra = box[0]
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out: List[State] = [dict(e) for e in extra]
    by_dtype: Dict[str, List[Tuple[int, str, tuple]]] = {}
    for i, lst in enumerate(meta):
        for k, shp, dt in lst:
            by_dtype.setdefault(dt, []).append((i, k, shp))
    bufs = []
    for dt in sorted(by_dtype):
        items = by_dtype[dt]
        sizes = [int(torch.Size(shp).numel()) for _, _, shp in items]
        buf = torch.empty(sum(sizes), dtype=getattr(torch, dt), device=device)
        if rank == src:
            off = 0
            for (i, k, _), n in zip(items, sizes):
                buf[off:off + n].copy_(states[i][k].reshape(-1))
                off += n
        dist.broadcast(buf, src=src)
        off = 0
        for (i, k, shp), n in zip(items, sizes):
            out[i][k] = buf[off:off + n].view(shp)
            off += n
        info["bytes"] += buf.numel() * buf.element_size()
        info["buffers"] += 1
        bufs.append(buf)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    info["ms"] = 1e3 * (time.perf_counter() - t0)          # packing on rank src + the broadcast(s); the check below is not in it
    if check and bufs:
        sums = []
        for buf in bufs:
            raw = buf.view(torch.uint8)
            pad = (-raw.numel()) % 8
            if pad:
                raw = torch.cat([raw, raw.new_zeros(pad)])
            sums.append(raw.view(torch.int64).sum().reshape(1))          # wrap-around 64-bit sum of the bytes as received
        mine = torch.cat(sums)
        gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
        if not all(torch.equal(g, gathered[src]) for g in gathered):
            raise RuntimeError("broadcast_state: a rank holds different weights than rank %d after the broadcast" % src)
        info["checked"] = True
    # (the returned tensors are views of the flat per-dtype buffers: a caller that keeps only some of them keeps the whole buffer alive -
    #  load_state_dict / the engines' packers copy what they need, after which the dicts can be dropped)
    return (out[0] if single else out), info
