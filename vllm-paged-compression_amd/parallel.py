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


class OneShotAllReduce:
    """All-reduce over peer-mapped buffers for decode-sized messages, in one stage or - for the
    larger ones - two (the role of the reference's CustomAllreduce,
    vllm/distributed/device_communicators/custom_all_reduce.py, which ``GroupCoordinator`` tries
    before NCCL, vllm/distributed/parallel_state.py:357-379).

    Every rank allocates one shared buffer, exports it with HIP IPC, and maps its peers' buffers
    (the handles travel through ``dist.all_gather_object`` on ``group``, any backend).  Ranks
    must live on one node; they may even share a device (how the single-GPU tests run it).
    ``all_reduce`` is in place and must be called by all ranks in the same order."""

    MAX_RANKS = 8

    def __init__(self, group: Optional[dist.ProcessGroup], device, max_bytes: int = 1 << 20):
        import ctypes
        from . import _lib
        self._lib = lib = _lib.load()
        self._check = _lib.check
        self.group = group
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > self.MAX_RANKS:
            raise ValueError(f"one-shot all-reduce serves up to {self.MAX_RANKS} ranks")
        self.max_bytes = int(max_bytes)
        # the two-stage form keeps a rank's reduced chunk behind its input in the slot
        self.two_stage_from = (None if self.world <= 2 else
                               512 * 1024 if self.world <= 4 else 256 * 1024)
        self.force_two_stage = False       # tests: every message through the two-stage form
        slot = self.max_bytes
        if self.world > 1:
            slot += -(-self.max_bytes // (16 * self.world)) * 16
        self.buffer_bytes = lib.pe_ar_buffer_bytes(slot)
        self.seq = 0
        self.calls_two_stage = 0
        self._own = ctypes.c_void_p()
        self._peers = []
        with torch.cuda.device(self.device):
            self._check(lib.pe_ar_alloc(ctypes.byref(self._own), self.buffer_bytes), "ar_alloc")
            hb = lib.pe_ar_ipc_handle_bytes()
            handle = ctypes.create_string_buffer(hb)
            self._check(lib.pe_ar_get_ipc_handle(self._own, handle), "ar_get_ipc_handle")
            handles = [None] * self.world
            dist.all_gather_object(handles, handle.raw, group=group)
            ptrs = (ctypes.c_void_p * self.world)()
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs[r] = self._own.value
                    continue
                p = ctypes.c_void_p()
                self._check(lib.pe_ar_open_ipc_handle(ctypes.create_string_buffer(h, hb), ctypes.byref(p)),
                            "ar_open_ipc_handle")
                self._peers.append(p)
                ptrs[r] = p.value
            self._ptrs = ptrs
        dist.barrier(group=group)   # every rank has mapped every buffer before the first call

    def should_use(self, x: torch.Tensor) -> bool:
        """Message small enough for the shared slot, 16-byte granular, contiguous, on our device."""
        n = x.numel() * x.element_size()
        return (x.is_cuda and x.device == self.device and x.is_contiguous() and 0 < n <= self.max_bytes
                and n % 16 == 0 and x.data_ptr() % 16 == 0
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32))

    def two_stage(self, nbytes: int) -> bool:
        """The reference's choice between its two kernels (csrc/custom_all_reduce.cuh:476-483):
        one stage for two ranks and for small messages, two stages (reduce-scatter + all-gather)
        from 512 KiB with up to four ranks and from 256 KiB with up to eight."""
        if self.force_two_stage:
            return True
        return self.two_stage_from is not None and nbytes >= self.two_stage_from

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        from . import _lib
        code = {torch.float32: _lib.PE_F32, torch.float16: _lib.PE_F16, torch.bfloat16: _lib.PE_BF16}[x.dtype]
        self.seq += 1
        two = self.two_stage(x.numel() * x.element_size())
        fn = self._lib.pe_allreduce_twostage if two else self._lib.pe_allreduce_oneshot
        self.calls_two_stage += int(two)
        self._check(fn(x.data_ptr(), x.data_ptr(), x.numel(), code, self._ptrs, self.rank, self.world,
                       self.buffer_bytes, self.seq, torch.cuda.current_stream(self.device).cuda_stream),
                    "allreduce_twostage" if two else "allreduce_oneshot")
        return x

    def timeouts(self) -> int:
        """Waits that gave up because a peer never arrived (synchronises the device)."""
        import ctypes
        n = ctypes.c_uint(0)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            self._check(self._lib.pe_ar_timeouts(self._own, ctypes.byref(n)), "ar_timeouts")
        return int(n.value)

    def close(self) -> None:
        if self._own is None:
            return
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            dist.barrier(group=self.group)      # nobody is still reading our buffer
            for p in self._peers:
                self._lib.pe_ar_close_ipc_handle(p)
            self._lib.pe_ar_free(self._own)
        self._own, self._peers = None, []


_one_shot = {}


def enable_one_shot_all_reduce(group: Optional[dist.ProcessGroup], device,
                               max_bytes: int = 8 << 20) -> OneShotAllReduce:
    """Register a peer-buffer communicator for ``group``: ``all_reduce_sum`` then serves messages
    of up to ``max_bytes`` (default: the reference's max_size of 8 MiB,
    custom_all_reduce.py:56) through it and leaves larger ones to the group's backend (RCCL)."""
    comm = OneShotAllReduce(group, device, max_bytes)
    _one_shot[id(group) if group is not None else None] = comm
    return comm


def disable_one_shot_all_reduce(group: Optional[dist.ProcessGroup] = None) -> None:
    comm = _one_shot.pop(id(group) if group is not None else None, None)
    if comm is not None:
        comm.close()


def all_reduce_sum(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """In-place sum over the tensor-parallel group (the o_proj partial-output reduction).
    Dispatch order as the reference's (parallel_state.py:357-379): the one-shot communicator
    first when one is registered for the group and the message fits it, then the group's own
    backend."""
    comm = _one_shot.get(id(group) if group is not None else None)
    if comm is not None and comm.should_use(x):
        return comm.all_reduce(x)
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
