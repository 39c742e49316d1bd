"""Tensor-parallel helpers of the path: head-wise sharding and the two exchange steps.

Attention itself needs no collective: rank r owns ``num_heads/tp`` query heads, the KV heads
they read and the matching slice of the paged cache (the layout of
vllm/model_executor/models/llama.py:117-130); block tables, slot mappings and kv lengths are
replicated.  The exchange steps are
  * the all-reduce (sum) of the row-parallel o_proj output that follows attention
    (vllm/model_executor/layers/linear.py:1099-1100), and
  * making every rank evict the same block when the victim depends on data and the scores are
    not rank-invariant (the reference-compatible flat cache view): the deciding rank's indices
    are broadcast, which is what the reference effectively does by committing only the driver
    rank's layer-0 decision (vllm/worker/model_runner.py:1797-1802);
  * value_l2 scores average over ALL KV heads: each rank sums the norms of its heads per token,
    the f32 token sums are all-reduced and every rank finishes the score itself
    (``all_reduce_sum`` on the f32 sums), so all ranks hold identical score tables and take
    identical decisions - the same ones a single GPU takes.
Both use torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests; a
GPU tensor handed to a gloo group is staged through the host).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    """Heads owned by one tensor-parallel rank."""
    tp_size: int
    rank: int
    num_heads: int          # query heads on this rank
    num_kv_heads: int       # KV heads on this rank (>= 1; replicated when tp_size > total KV heads)
    q_head_start: int       # first global query head
    kv_head_start: int      # first global KV head

    @property
    def q_heads(self) -> range:
        return range(self.q_head_start, self.q_head_start + self.num_heads)

    @property
    def kv_heads(self) -> range:
        return range(self.kv_head_start, self.kv_head_start + self.num_kv_heads)


def shard_heads(total_num_heads: int, total_num_kv_heads: int, tp_size: int, rank: int) -> HeadShard:
    if total_num_heads % tp_size:
        raise ValueError(f"{total_num_heads} query heads cannot be split over {tp_size} ranks")
    if not 0 <= rank < tp_size:
        raise ValueError("rank out of range")
    nh = total_num_heads // tp_size
    if total_num_kv_heads >= tp_size:
        if total_num_kv_heads % tp_size:
            raise ValueError(f"{total_num_kv_heads} KV heads cannot be split over {tp_size} ranks")
        nkv = total_num_kv_heads // tp_size
        kv_start = rank * nkv
    else:
        # fewer KV heads than ranks: each KV head is replicated on tp_size/total_kv ranks
        if tp_size % total_num_kv_heads:
            raise ValueError(f"{tp_size} ranks must be a multiple of {total_num_kv_heads} KV heads")
        nkv = 1
        kv_start = rank // (tp_size // total_num_kv_heads)
    return HeadShard(tp_size, rank, nh, nkv, rank * nh, kv_start)


def broadcast_victims(rmv_idx: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                      src: int = 0) -> torch.Tensor:
    """Make every rank of the group use rank ``src``'s victim indices (int32 [batch], -1 = none)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        root = dist.get_global_rank(group, src) if group is not None else src
        if _host_staged(rmv_idx, group):
            tmp = rmv_idx.cpu()
            dist.broadcast(tmp, src=root, group=group)
            rmv_idx.copy_(tmp)
        else:
            dist.broadcast(rmv_idx, src=root, group=group)
    return rmv_idx


def all_reduce_sum(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """In-place sum over the tensor-parallel group (the o_proj partial-output reduction)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(x, group):
            tmp = x.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            x.copy_(tmp)
        else:
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def _host_staged(x: torch.Tensor, group: Optional[dist.ProcessGroup]) -> bool:
    return x.is_cuda and dist.get_backend(group) == "gloo"


def group_rank(group: Optional[dist.ProcessGroup] = None) -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group)
    return 0


def group_size(group: Optional[dist.ProcessGroup] = None) -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1
