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


class AllReduceError(RuntimeError):
    """The peer-buffer exchange lost a rank: an output was filled with NaN instead of a sum."""


class OneShotAllReduce:
    """All-reduce over peer-mapped buffers for decode-sized messages, in one stage or - for the
    larger ones - two (the role of the reference's CustomAllreduce,
    vllm/distributed/device_communicators/custom_all_reduce.py, which ``GroupCoordinator`` tries
    before NCCL, vllm/distributed/parallel_state.py:357-379).

    Every rank allocates one shared buffer, exports it with HIP IPC, and maps its peers' buffers
    (the handles travel through ``dist.all_gather_object`` on ``group``, any backend).  Ranks
    must live on one node; they may even share a device (how the single-GPU tests run it).
    ``all_reduce`` is in place and must be called by all ranks in the same order.

    Set-up verifies the mappings before anything relies on them: every rank writes tagged
    patterns into its own and - through the mappings - into its peers' buffers and checks what
    it reads back (``pe_ar_selfcheck``), then one exchange of each kind is run on a known input
    with a short wait bound.  If any rank sees anything wrong, ALL ranks switch the communicator
    off (``enabled`` False, reason on stderr) and ``all_reduce_sum`` uses the group's own backend.

    A rank that misses a call cannot make the others return a partial sum: their wait gives up
    after ``timeout_s`` seconds, the output is filled with NaN, and the status words the kernel
    raises in pinned host memory make the next ``all_reduce`` / ``check`` / ``close`` raise
    ``AllReduceError`` (the reference's barrier would spin forever instead,
    csrc/custom_all_reduce.cuh:140-200)."""

    MAX_RANKS = 8
    SEQ_WRAP = 0xFFFFFFFE      # even: the call after the wrap (1) uses the other slot

    def __init__(self, group: Optional[dist.ProcessGroup], device, max_bytes: int = 1 << 20,
                 timeout_s: Optional[float] = None, self_check: bool = True):
        import ctypes
        import os
        from . import _lib
        self._lib = lib = _lib.load()
        self._check = _lib.check
        self.group = group
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > self.MAX_RANKS:
            raise ValueError(f"one-shot all-reduce serves up to {self.MAX_RANKS} ranks")
        self.max_bytes = int(max_bytes)
        if timeout_s is None:
            timeout_s = float(os.environ.get("PE_AR_TIMEOUT_S", "10"))
        self.timeout_us = max(1, int(timeout_s * 1e6))
        # the two-stage form keeps a rank's reduced chunk behind its input in the slot.  The
        # crossover sizes are the reference's (NVLink) ones, unmeasured on xGMI:
        # PE_AR_TWO_STAGE_BYTES overrides them
        self.two_stage_from = (None if self.world <= 2 else
                               512 * 1024 if self.world <= 4 else 256 * 1024)
        if self.world > 2 and os.environ.get("PE_AR_TWO_STAGE_BYTES"):
            self.two_stage_from = int(os.environ["PE_AR_TWO_STAGE_BYTES"])
        self.force_two_stage = False       # tests: every message through the two-stage form
        slot = self.max_bytes
        if self.world > 1:
            slot += -(-self.max_bytes // (16 * self.world)) * 16
        self.buffer_bytes = lib.pe_ar_buffer_bytes(slot)
        self.seq = 0
        self.calls_two_stage = 0
        self.enabled = True
        self.disabled_reason: Optional[str] = None
        self._own = ctypes.c_void_p()
        self._status = ctypes.c_void_p()
        self._status_words = None
        self._peers = []
        with torch.cuda.device(self.device):
            self._check(lib.pe_ar_alloc(ctypes.byref(self._own), self.buffer_bytes), "ar_alloc")
            self._check(lib.pe_ar_status_alloc(ctypes.byref(self._status)), "ar_status_alloc")
            self._status_words = (ctypes.c_uint32 * 16).from_address(self._status.value)
            hb = lib.pe_ar_ipc_handle_bytes()
            handle = ctypes.create_string_buffer(hb)
            self._check(lib.pe_ar_get_ipc_handle(self._own, handle), "ar_get_ipc_handle")
            handles = [None] * self.world
            dist.all_gather_object(handles, handle.raw, group=group)
            ptrs = (ctypes.c_void_p * self.world)()
            problem = None
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs[r] = self._own.value
                    continue
                p = ctypes.c_void_p()
                rc = lib.pe_ar_open_ipc_handle(ctypes.create_string_buffer(h, hb), ctypes.byref(p))
                if rc != 0:
                    problem = f"mapping rank {r}'s buffer failed: {lib.pe_last_error().decode()}"
                    break
                self._peers.append(p)
                ptrs[r] = p.value
            self._ptrs = ptrs
        # every rank has mapped every buffer before the first call - or nobody uses them
        if not self._agree(problem) and self_check:
            self._agree(self._verify_mappings())
        if self.enabled and self_check and self.world > 1:
            self._agree(self._verify_exchange())

    # ---- set-up checks -------------------------------------------------------------------
    def _agree(self, problem: Optional[str]) -> bool:
        """Collective: switch the communicator off on every rank if any rank reports a problem.
        Returns True when it was switched off."""
        import sys
        reports = [None] * self.world
        dist.all_gather_object(reports, problem, group=self.group)
        bad = [(r, p) for r, p in enumerate(reports) if p]
        if bad and self.enabled:
            self.enabled = False
            self.disabled_reason = "; ".join(f"rank {r}: {p}" for r, p in bad)
            if self.rank == 0:
                print("paged_evict: peer-buffer all-reduce disabled, the group's own backend is used "
                      f"instead ({self.disabled_reason})", file=sys.stderr, flush=True)
        return bool(bad)

    def _verify_mappings(self) -> Optional[str]:
        lib = self._lib
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            bad = torch.zeros(1, dtype=torch.int32, device=self.device)
            rc = lib.pe_ar_selfcheck(self._ptrs, self.rank, self.world, 0, None, stream)
            torch.cuda.synchronize()
            dist.barrier(group=self.group)          # every rank's patterns are written
            if rc == 0:
                rc = lib.pe_ar_selfcheck(self._ptrs, self.rank, self.world, 1, bad.data_ptr(), stream)
            torch.cuda.synchronize()
            dist.barrier(group=self.group)
            if rc != 0:
                return f"self-check launch failed: {lib.pe_last_error().decode()}"
            n = int(bad.item())
        return f"{n} threads read a wrong pattern through the peer mappings" if n else None

    def _verify_exchange(self) -> Optional[str]:
        """One exchange of each kind on a known input, with a wait bound of a few seconds."""
        saved = self.timeout_us, self.force_two_stage
        self.timeout_us = min(self.timeout_us, 5_000_000)
        problem = None
        try:
            n = min(self.max_bytes, 64 * 1024) // 4
            n -= n % (4 * self.world)
            expect = float(self.world * (self.world + 1) // 2)
            for two in (False, True):
                if two and self.world <= 2 and self.two_stage_from is None:
                    break
                self.force_two_stage = two
                x = torch.full((max(n, 4 * self.world),), float(self.rank + 1), dtype=torch.float32,
                               device=self.device)
                self._launch(x)
                torch.cuda.synchronize(self.device)
                if self._status_words[0] != 0:
                    problem = (f"a peer did not arrive in the {'two-stage' if two else 'one-stage'} "
                               f"test exchange (missing rank {int(self._status_words[2]) - 1})")
                    break
                if not bool((x == expect).all()):
                    problem = f"the {'two-stage' if two else 'one-stage'} test exchange gave a wrong sum"
                    break
        finally:
            self.timeout_us, self.force_two_stage = saved
            self.calls_two_stage = 0
        return problem

    # ---- the exchange ---------------------------------------------------------------------
    def should_use(self, x: torch.Tensor) -> bool:
        """Message small enough for the shared slot, 16-byte granular, contiguous, on our device."""
        n = x.numel() * x.element_size()
        return (self.enabled and x.is_cuda and x.device == self.device and x.is_contiguous()
                and 0 < n <= self.max_bytes and n % 16 == 0 and x.data_ptr() % 16 == 0
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32))

    def two_stage(self, nbytes: int) -> bool:
        """The reference's choice between its two kernels (csrc/custom_all_reduce.cuh:476-483):
        one stage for two ranks and for small messages, two stages (reduce-scatter + all-gather)
        from 512 KiB with up to four ranks and from 256 KiB with up to eight."""
        if self.force_two_stage:
            return True
        return self.two_stage_from is not None and nbytes >= self.two_stage_from

    def path(self, nbytes: int) -> str:
        """Which exchange a message of ``nbytes`` takes: 'oneshot', 'twostage' or 'backend'."""
        if not self.enabled or not 0 < nbytes <= self.max_bytes or nbytes % 16:
            return "backend"
        return "twostage" if self.two_stage(nbytes) else "oneshot"

    def _launch(self, x: torch.Tensor) -> None:
        from . import _lib
        code = {torch.float32: _lib.PE_F32, torch.float16: _lib.PE_F16, torch.bfloat16: _lib.PE_BF16}[x.dtype]
        # 1, 2, ..., SEQ_WRAP, 1, ...: the kernels compare flags by signed difference, and the
        # wrap keeps consecutive calls on alternating slots
        self.seq = self.seq % self.SEQ_WRAP + 1
        two = self.two_stage(x.numel() * x.element_size())
        fn = self._lib.pe_allreduce_twostage if two else self._lib.pe_allreduce_oneshot
        self.calls_two_stage += int(two)
        self._check(fn(x.data_ptr(), x.data_ptr(), x.numel(), code, self._ptrs, self.rank, self.world,
                       self.buffer_bytes, self.seq, self._status, self.timeout_us,
                       torch.cuda.current_stream(self.device).cuda_stream),
                    "allreduce_twostage" if two else "allreduce_oneshot")

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self._status_words[0] != 0:      # plain read of pinned host memory: no device call
            self._raise_lost_peer()
        self._launch(x)
        return x

    def _raise_lost_peer(self) -> None:
        w = self._status_words
        raise AllReduceError(
            f"peer-buffer all-reduce on rank {self.rank}: rank {int(w[2]) - 1} did not arrive at call "
            f"{int(w[1])} within {self.timeout_us / 1e6:g} s; the output of that call was filled with NaN "
            "and the communicator must not be used again")

    def check(self) -> None:
        """Wait for the queued exchanges and raise if any of them lost a peer."""
        torch.cuda.synchronize(self.device)
        if self._status_words is not None and self._status_words[0] != 0:
            self._raise_lost_peer()

    def timeouts(self) -> int:
        """Waits that gave up because a peer never arrived (synchronises the device)."""
        import ctypes
        n = ctypes.c_uint(0)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            self._check(self._lib.pe_ar_timeouts(self._own, ctypes.byref(n)), "ar_timeouts")
        return int(n.value)

    def close(self) -> None:
        """Unmap and free (collective).  Raises AllReduceError afterwards if an exchange lost a
        peer and nobody has been told yet."""
        if self._own is None:
            return
        lost = False
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            lost = self._status_words[0] != 0
            if lost:
                w = [int(v) for v in self._status_words[:3]]
            dist.barrier(group=self.group)      # nobody is still reading our buffer
            for p in self._peers:
                self._lib.pe_ar_close_ipc_handle(p)
            self._lib.pe_ar_free(self._own)
            self._status_words = None
            self._lib.pe_ar_status_free(self._status)
        self._own, self._peers, self.enabled = None, [], False
        if lost:
            raise AllReduceError(f"peer-buffer all-reduce on rank {self.rank}: rank {w[2] - 1} did not "
                                 f"arrive at call {w[1]}; outputs of that call were NaN")


_one_shot = {}


def enable_one_shot_all_reduce(group: Optional[dist.ProcessGroup], device,
                               max_bytes: int = 8 << 20, **kwargs) -> Optional[OneShotAllReduce]:
    """Register a peer-buffer communicator for ``group``: ``all_reduce_sum`` then serves messages
    of up to ``max_bytes`` (default: the reference's max_size of 8 MiB,
    custom_all_reduce.py:56) through it and leaves larger ones to the group's backend (RCCL).
    Collective.  PE_AR_DISABLE=1 in the environment keeps every message on the group's backend
    (returns None); a communicator whose set-up checks fail registers itself switched off."""
    import os
    if os.environ.get("PE_AR_DISABLE", "0") not in ("", "0"):
        return None
    comm = OneShotAllReduce(group, device, max_bytes, **kwargs)
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
