"""Vectorised eviction scheduler: the bookkeeping of ``block_table.EvictionScheduler`` for a whole
decode batch at a time.

``EvictBlockTable`` keeps one Python object and two Python lists per sequence, which is what
the reference does (vllm/core/block/block_table.py) and costs a few microseconds per sequence
per step; at several hundred running sequences that is more host time than the GPU needs for
the step.  Here the state is a struct of arrays - block ids ``int32 [rows, max_blocks]``, block
counts and the two slot counters of the fork (logical ``num_full_slots`` and physical
``num_cached_slots``, block_table.py:309-353) - and one decode step is a handful of numpy
operations: append a slot to every running sequence, hand out blocks to the few that need one,
gather the batch's table matrix, and apply a step's eviction decisions (block_table.py:456-466,
scheduler.py:1879-1890, llm_engine.py:1179-1193) with one indexed shift.  Semantics are those of
``EvictionScheduler``; tests/test_abi_and_host.py walks both side by side.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .page_evict_kv_util import cdiv, get_num_required_blocks_after_prune_promt


@dataclass
class DecodeBatch:
    """One decode step's view of the running sequences (arrays are in batch order)."""
    request_ids: List[str]
    seq_ids: List[int]
    rows: np.ndarray          # state rows of the sequences
    block_tables: np.ndarray  # int32 [B, M] (M = longest table), zero padded
    table_lens: np.ndarray    # int32 [B]
    kv_lens: np.ndarray       # int64 [B] physical lengths incl. this step's token
    seq_lens: np.ndarray      # int64 [B] logical lengths incl. this step's token


class BatchEvictionScheduler:

    def __init__(self, block_size: int, num_gpu_blocks: int, paged_evict_config=None,
                 max_seqs: int = 256, max_blocks_per_seq: int = 512):
        self.block_size = block_size
        self.paged_evict_config = paged_evict_config
        self.num_total_gpu_blocks = num_gpu_blocks
        # free list: FIFO ring (blocks are handed out in the order they were freed, like the
        # per-request allocator's queue)
        self._free = np.arange(num_gpu_blocks, dtype=np.int32)
        self._head = 0
        self._n_free = num_gpu_blocks
        self.tables = np.zeros((max_seqs, max_blocks_per_seq), dtype=np.int32)
        self.n_blocks = np.zeros(max_seqs, dtype=np.int32)
        self.num_full = np.zeros(max_seqs, dtype=np.int64)
        self.num_cached = np.zeros(max_seqs, dtype=np.int64)
        self._free_rows = list(range(max_seqs - 1, -1, -1))
        self._row_of_req: Dict[str, int] = {}
        self._seq_of_req: Dict[str, int] = {}
        self._batch_key: Optional[Tuple[str, ...]] = None
        self._batch_rows: Optional[np.ndarray] = None
        self._batch_seq_ids: Optional[List[int]] = None

    # ---- blocks ------------------------------------------------------------------------------
    def get_num_free_gpu_blocks(self) -> int:
        return self._n_free

    def _take(self, n: int) -> np.ndarray:
        if n > self._n_free:
            raise RuntimeError("out of KV-cache blocks")
        cap = len(self._free)
        out = self._free[(self._head + np.arange(n)) % cap]
        self._head = (self._head + n) % cap
        self._n_free -= n
        return out

    def _give(self, ids: np.ndarray) -> None:
        n = len(ids)
        cap = len(self._free)
        self._free[(self._head + self._n_free + np.arange(n)) % cap] = ids
        self._n_free += n

    # ---- requests ----------------------------------------------------------------------------
    def add_request(self, request_id: str, seq_id: int, prompt_len: int) -> None:
        """Pruned prompt allocation (page_evict_kv_util.py:6-36, block_table.py:94-126)."""
        bs = self.block_size
        kept = prompt_len
        if self.paged_evict_config is not None:
            _, kept = get_num_required_blocks_after_prune_promt(prompt_len, self.paged_evict_config, bs)
        n = cdiv(kept, bs)
        row = self._free_rows.pop()
        self.tables[row, :n] = self._take(n)
        self.n_blocks[row] = n
        self.num_full[row] = prompt_len
        if self.paged_evict_config is None:
            self.num_cached[row] = prompt_len
        else:
            rem = prompt_len % bs
            self.num_cached[row] = n * bs if rem == 0 else (n - 1) * bs + rem
        self._row_of_req[request_id] = row
        self._seq_of_req[request_id] = seq_id
        self._batch_key = None

    def finish_request(self, request_id: str) -> None:
        row = self._row_of_req.pop(request_id)
        self._seq_of_req.pop(request_id)
        n = int(self.n_blocks[row])
        self._give(self.tables[row, :n].copy())
        self.tables[row, :n] = 0
        self.n_blocks[row] = 0
        self.num_full[row] = self.num_cached[row] = 0
        self._free_rows.append(row)
        self._batch_key = None

    def get_block_table(self, request_id: str) -> List[int]:
        row = self._row_of_req[request_id]
        return self.tables[row, :self.n_blocks[row]].tolist()

    def get_seq_kv_len(self, request_id: str) -> int:
        return int(self.num_cached[self._row_of_req[request_id]])

    # ---- one decode step -----------------------------------------------------------------------
    def _rows_for(self, request_ids: Sequence[str]) -> np.ndarray:
        key = tuple(request_ids)
        if key != self._batch_key:  # the batch composition rarely changes between steps
            self._batch_rows = np.fromiter((self._row_of_req[r] for r in request_ids), dtype=np.int64,
                                           count=len(request_ids))
            self._batch_seq_ids = [self._seq_of_req[r] for r in request_ids]
            self._batch_key = key
        return self._batch_rows

    def schedule_decode(self, request_ids: Sequence[str]) -> DecodeBatch:
        """Append one slot to every request (block_table.py:136-196) and return the step's view."""
        bs = self.block_size
        rows = self._rows_for(request_ids)
        self.num_full[rows] += 1
        self.num_cached[rows] += 1
        need = np.nonzero(self.num_cached[rows] > self.n_blocks[rows].astype(np.int64) * bs)[0]
        if len(need):
            r = rows[need]
            self.tables[r, self.n_blocks[r]] = self._take(len(need))
            self.n_blocks[r] += 1
        lens = self.n_blocks[rows]
        m = int(lens.max()) if len(rows) else 0
        return DecodeBatch(list(request_ids), self._batch_seq_ids, rows,
                           self.tables[rows, :m], lens.copy(), self.num_cached[rows],
                           self.num_full[rows])

    def commit(self, reqids_to_rmv_block_idx: Dict[int, Tuple[str, int]]) -> None:
        """Apply a step's ``get_rmv_block_idx()`` result: free the victim blocks and close the
        gaps (block_table.py:456-466)."""
        if not reqids_to_rmv_block_idx:
            return
        cfg = self.paged_evict_config
        # percentage prune type: num_block_merge-1 consecutive blocks go, from the reported index
        n = cfg.num_block_merge - 1 if cfg is not None and cfg.cache_prune_type == "percentage" else 1
        for _ in range(n):
            self._free_column(reqids_to_rmv_block_idx)

    def _free_column(self, reqids_to_rmv_block_idx: Dict[int, Tuple[str, int]]) -> None:
        bs = self.block_size
        items = list(reqids_to_rmv_block_idx.values())
        rows = np.fromiter((self._row_of_req[r] for r, _ in items), dtype=np.int64, count=len(items))
        idx = np.fromiter((i for _, i in items), dtype=np.int64, count=len(items))
        n = self.n_blocks[rows].astype(np.int64)
        # tokens held by the victim: a full block unless it is the (possibly partial) last one
        fill = np.where(idx < n - 1, bs, self.num_cached[rows] - (n - 1) * bs)
        self._give(self.tables[rows, idx].copy())
        m = int(n.max())
        cols = np.arange(m - 1, dtype=np.int64)[None, :]
        src = cols + (cols >= idx[:, None])
        self.tables[rows[:, None], cols] = self.tables[rows[:, None], src]
        self.tables[rows, n - 1] = 0  # keep the padding past a row's end zero
        self.n_blocks[rows] -= 1
        self.num_cached[rows] -= fill
