"""Clip / window parallelism across the GPUs of one node.

The hot path shards over INDEPENDENT forwards -- whole clips, or the sliding windows the reference's
driver cuts a long video into (scripts/inference_with_video_mesh.py:176-216; windows share only frame 0).
One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests), identical
weights, NO collective inside the forward; the only exchange is an all-gather of the per-item outputs
(`[T, N, 3]` fp32 = 786 KB for the BASELINE clip) at the end.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


# Run the collectives even in a process group of ONE rank (they are identities there).  The `-m gpu` tests set it to take
# forward_frame_parallel / the bucketed gradient all-reduce / the window gather through torch.distributed's "nccl" backend
# (= RCCL) on the single GPU of the test box: async work handles, side streams and record_stream as on a full node.
ALWAYS_COLLECT = False


def collectives_on(world: int) -> bool:
    return world > 1 or (ALWAYS_COLLECT and dist.is_available() and dist.is_initialized())


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def partition(n_items: int, world: int, rank: int) -> range:
    """Contiguous, balanced shard of range(n_items): the first n_items % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def counts(n_items: int, world: int) -> List[int]:
    return [len(partition(n_items, world, r)) for r in range(world)]


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out[r * n : (r + 1) * n] = rank r's `inp` (n = inp.numel(); both contiguous): one all_gather_into_tensor straight
    into `out`, whatever its shape ([world, ...] stacked or concatenated along dim 0).  `out` is handed over in the
    concatenated form, the only one every backend accepts (RCCL takes both; gloo -- the transport of the single-device
    rehearsals and tests -- only this one)."""
    world = dist.get_world_size(group)
    if out.numel() != world * inp.numel() or not (out.is_contiguous() and inp.is_contiguous()):
        raise ValueError(f"all_gather_into: out {tuple(out.shape)} must be contiguous with world x {tuple(inp.shape)} elements")
    shape = (world * inp.shape[0],) + tuple(inp.shape[1:]) if inp.dim() > 0 else (world,)
    return dist.all_gather_into_tensor(out.view(shape), inp, group=group, async_op=async_op)


def all_gather_items(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """local: [n_local, ...] outputs of this rank's shard (partition order).  Returns [n_items, ...] on every
    rank.  Shards may be uneven: each rank pads to the largest shard, one all_gather moves everything."""
    rank, world = world_info(group)
    if not collectives_on(world):
        assert local.shape[0] == n_items
        return local
    cnt = counts(n_items, world)
    assert local.shape[0] == cnt[rank], (local.shape, cnt, rank)
    m = max(cnt)
    item_shape = tuple(local.shape[1:])
    padded = local.new_zeros((m,) + item_shape)
    padded[:cnt[rank]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded.contiguous(), group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, cnt)], dim=0)


def map_items(fn: Callable[[int], torch.Tensor], n_items: int, item_shape: Sequence[int], device, dtype=torch.float32,
              group=None) -> torch.Tensor:
    """Runs fn(i) -> tensor[item_shape] for this rank's shard of range(n_items) and all-gathers the results."""
    rank, world = world_info(group)
    mine = partition(n_items, world, rank)
    outs = [fn(i) for i in mine]
    local = torch.stack(outs, dim=0) if outs else torch.zeros((0,) + tuple(item_shape), dtype=dtype, device=device)
    return all_gather_items(local, n_items, group)
