"""Eviction-aware block bookkeeping — the host side of the path ("eviction scheduler").

Restates, as a small self-contained module, what the fork changes in vLLM's block manager:

* ``EvictBlockTable``         vllm/core/block/block_table.py:42-66,94-126,136-196,309-353,456-466
                              two counters: ``num_full_slots`` (logical length) and
                              ``num_cached_slots`` (tokens physically held), pruned prompt
                              allocation, ``free_prunned_blocks``
* ``EvictBlockSpaceManager``  vllm/core/block_manager.py:113-156,537-543 (admission with the pruned
                              size, per-sequence kv length, freeing a victim block)
* ``EvictionScheduler``       vllm/core/scheduler.py:1338-1361,1879-1890 and the engine commit
                              vllm/engine/llm_engine.py:1179-1193: per-step ``block_tables`` /
                              ``seq_kv_lens`` and ``update_block_tables(request_id, rmv_idx)``,
                              here an O(1) dictionary lookup instead of a scan over the running
                              queue

Method names keep the reference's spelling (``free_prunned_blocks``).
"""
from __future__ import annotations

from collections import deque
from typing import Deque, Dict, List, Tuple

from .page_evict_kv_util import cdiv, get_num_required_blocks_after_prune_promt


class BlockAllocator:
    """Free-list allocator of physical block ids [0, num_blocks) with the hand-out order of the
    reference's NaiveBlockAllocator (vllm/core/block/naive_block.py:127-145): ids are taken from
    the front of the list and a freed id goes back to the FRONT, so the block freed last is the
    next one handed out."""

    def __init__(self, num_blocks: int):
        self.num_blocks = num_blocks
        self._free: Deque[int] = deque(range(num_blocks))

    def get_num_free_blocks(self) -> int:
        return len(self._free)

    def allocate(self) -> int:
        if not self._free:
            raise RuntimeError("out of KV-cache blocks")
        return self._free.popleft()

    def free(self, block_id: int) -> None:
        self._free.appendleft(block_id)


class EvictBlockTable:

    def __init__(self, block_size: int, allocator: BlockAllocator, paged_evict_config=None):
        self._block_size = block_size
        self._allocator = allocator
        self.paged_evict_config = paged_evict_config
        self._blocks: List[int] = []
        self._block_fill: List[int] = []  # tokens held by each block
        self._num_full_slots = 0
        self._num_cached_slots = 0

    # ---- prompt ------------------------------------------------------------------------
    def allocate(self, num_tokens: int) -> None:
        assert not self._blocks and num_tokens > 0
        if self.paged_evict_config is None:
            kept = num_tokens
        else:
            _, kept = get_num_required_blocks_after_prune_promt(
                num_tokens, self.paged_evict_config, self._block_size)
        n_blocks = cdiv(kept, self._block_size)
        self._blocks = [self._allocator.allocate() for _ in range(n_blocks)]
        self._block_fill = [self._block_size] * n_blocks
        if kept % self._block_size:
            self._block_fill[-1] = kept % self._block_size
        self._num_full_slots = num_tokens
        if self.paged_evict_config is None:
            self._num_cached_slots = num_tokens
        else:
            rem = num_tokens % self._block_size
            self._num_cached_slots = (n_blocks * self._block_size if rem == 0 else
                                      (n_blocks - 1) * self._block_size + rem)

    # ---- decode --------------------------------------------------------------------------
    def append_token_ids(self, num_tokens: int = 1) -> None:
        assert self._blocks, "no blocks have been allocated"
        for _ in range(num_tokens):
            if self._num_empty_slots < 1:
                self._blocks.append(self._allocator.allocate())
                self._block_fill.append(0)
            self._block_fill[self._num_cached_slots // self._block_size] += 1
            self._num_full_slots += 1
            self._num_cached_slots += 1

    @property
    def _num_empty_slots(self) -> int:
        return len(self._blocks) * self._block_size - self._num_cached_slots

    def get_seq_kv_len(self) -> int:
        return self._num_cached_slots

    @property
    def num_full_slots(self) -> int:
        return self._num_full_slots

    @property
    def physical_block_ids(self) -> List[int]:
        return self._blocks

    def free_prunned_blocks(self, rmv_block_idx: int) -> None:
        self._allocator.free(self._blocks[rmv_block_idx])
        self._num_cached_slots -= self._block_fill[rmv_block_idx]
        del self._blocks[rmv_block_idx]
        del self._block_fill[rmv_block_idx]

    def free(self) -> None:
        for b in self._blocks:
            self._allocator.free(b)
        self._blocks, self._block_fill = [], []
        self._num_cached_slots = 0


class EvictBlockSpaceManager:

    def __init__(self, block_size: int, num_gpu_blocks: int, paged_evict_config=None,
                 watermark: float = 0.01):
        self.block_size = block_size
        self.paged_evict_config = paged_evict_config
        self.allocator = BlockAllocator(num_gpu_blocks)
        self.num_total_gpu_blocks = num_gpu_blocks
        self.watermark_blocks = int(watermark * num_gpu_blocks)
        self.block_tables: Dict[int, EvictBlockTable] = {}

    def num_required_blocks(self, prompt_len: int) -> int:
        if self.paged_evict_config is None:
            return cdiv(prompt_len, self.block_size)
        return get_num_required_blocks_after_prune_promt(prompt_len, self.paged_evict_config,
                                                         self.block_size)[0]

    def can_allocate(self, prompt_len: int) -> str:
        """"OK" / "LATER" / "NEVER" (AllocStatus of the reference)."""
        need = self.num_required_blocks(prompt_len)
        if self.num_total_gpu_blocks - need < self.watermark_blocks:
            return "NEVER"
        if self.allocator.get_num_free_blocks() - need >= self.watermark_blocks:
            return "OK"
        return "LATER"

    def allocate(self, seq_id: int, prompt_len: int) -> None:
        bt = EvictBlockTable(self.block_size, self.allocator, self.paged_evict_config)
        bt.allocate(prompt_len)
        self.block_tables[seq_id] = bt

    def can_append_slots(self, seq_id: int) -> bool:
        bt = self.block_tables[seq_id]
        return bt._num_empty_slots >= 1 or self.allocator.get_num_free_blocks() >= 1

    def append_slots(self, seq_id: int, num_tokens: int = 1) -> None:
        self.block_tables[seq_id].append_token_ids(num_tokens)

    def get_block_table(self, seq_id: int) -> List[int]:
        return self.block_tables[seq_id].physical_block_ids

    def get_seq_kv_len(self, seq_id: int) -> int:
        return self.block_tables[seq_id].get_seq_kv_len()

    def free_prunned_blocks(self, seq_id: int, rmv_block_idx: int) -> None:
        self.block_tables[seq_id].free_prunned_blocks(rmv_block_idx)

    def free(self, seq_id: int) -> None:
        self.block_tables.pop(seq_id).free()

    def get_num_free_gpu_blocks(self) -> int:
        return self.allocator.get_num_free_blocks()

    def gpu_cache_usage(self) -> float:
        """1 - free/total, the reference's "GPU KV cache usage" (llm_engine.py:1602-1608)."""
        return 1.0 - self.allocator.get_num_free_blocks() / max(self.num_total_gpu_blocks, 1)


class EvictionScheduler:
    """Per-step view for the worker and the post-step commit of eviction decisions."""

    def __init__(self, block_manager: EvictBlockSpaceManager):
        self.block_manager = block_manager
        self._req_to_seq: Dict[str, int] = {}

    def add_request(self, request_id: str, seq_id: int, prompt_len: int) -> None:
        self.block_manager.allocate(seq_id, prompt_len)
        self._req_to_seq[request_id] = seq_id

    def finish_request(self, request_id: str) -> None:
        self.block_manager.free(self._req_to_seq.pop(request_id))

    def schedule_decode(self, request_ids: List[str]) -> Tuple[Dict[int, List[int]], Dict[int, int]]:
        """append one slot per running request; returns (block_tables, seq_kv_lens) by seq id
        (scheduler.py:1338-1361: kv lens are the *physical* lengths, before this step's
        eviction adjustment, which the model runner applies, model_runner.py:538-543)."""
        tables, kv_lens = {}, {}
        for rid in request_ids:
            sid = self._req_to_seq[rid]
            self.block_manager.append_slots(sid)
            tables[sid] = self.block_manager.get_block_table(sid)
            kv_lens[sid] = self.block_manager.get_seq_kv_len(sid)
        return tables, kv_lens

    def update_block_tables(self, request_id: str, rmv_block_idx: int) -> None:
        self.block_manager.free_prunned_blocks(self._req_to_seq[request_id], rmv_block_idx)

    def commit(self, reqids_to_rmv_block_idx: Dict[int, Tuple[str, int]]) -> None:
        """Apply a step's ``get_rmv_block_idx()`` result ({batch_idx: (request_id, block_idx)})."""
        cfg = self.block_manager.paged_evict_config
        # percentage prune type: a decode prune frees num_block_merge-1 consecutive blocks
        # starting at the reported index (page_evict_kv_util_bak.py:66-106)
        n = cfg.num_block_merge - 1 if cfg is not None and cfg.cache_prune_type == "percentage" else 1
        for _, (request_id, idx) in reqids_to_rmv_block_idx.items():
            for _ in range(n):
                self.update_block_tables(request_id, idx)
