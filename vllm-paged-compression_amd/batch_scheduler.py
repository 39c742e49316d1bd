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
    block_tables: Optional[np.ndarray]  # int32 [B, M] (M = longest table), zero padded; None when
    #                           the tables live on the device (``device_state``)
    table_lens: np.ndarray    # int32 [B]
    kv_lens: np.ndarray       # int64 [B] physical lengths incl. this step's token
    seq_lens: np.ndarray      # int64 [B] logical lengths incl. this step's token
    device_state: Optional["DeviceTableState"] = None
    host_tables: Optional[np.ndarray] = None   # the scheduler's table array (rows = ``rows``)

    def table_row(self, i: int) -> List[int]:
        """Block ids of batch entry i (host copy)."""
        n = int(self.table_lens[i])
        if self.block_tables is not None:
            return self.block_tables[i, :n].tolist()
        return self.host_tables[self.rows[i], :n].tolist()


class DeviceTableState:
    """The running batch's block tables and length counters, resident on the GPU in batch order
    (row b = b-th sequence of the decode batch).

    A row goes up only when the batch composition changes (``load``); a decode step sends one
    int32 per sequence - the id of a block allocated for it this step, or -1 - and the device
    appends it, advances both length counters and derives the step's slot mapping and the kv
    lengths attention must see (``advance``, one launch).  The eviction launch of the step then
    closes the victim's column inside the row.  ``h2d_bytes`` counts what crossed PCIe.

    The per-step entries go through a ring of pinned host rows that the launch reads in place
    (sequence b's entry sits in the row its new logical length selects, so the launch has the same
    arguments at every step): no copy is queued, and the launch can be a node of a replayed HIP
    graph.  With ``defer_advance`` set, ``advance`` only stores the step's entries and the engine
    issues the launch itself - ``launch_advance()``, eagerly or captured at the head of the
    step's graph.  A graph that captured the launch reads this object's buffers at every replay:
    the state must outlive the graph (a new scheduler adopts it through ``device_state=``)."""

    RING = 64      # steps the host may run ahead of the device, kept below this by events
    _GUARD = 16    # an event every so many steps

    def __init__(self, capacity: int, max_blocks: int, device):
        import torch
        self.device = torch.device(device)
        self.capacity, self.max_blocks = capacity, max_blocks
        z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=self.device)  # noqa: E731
        self.tables = z(capacity, max_blocks)
        self.num_blocks, self.kv_lens, self.seq_lens = z(capacity), z(capacity), z(capacity)
        self.slot_mapping = z(capacity, dt=torch.int64)
        self.attn_kv_lens, self.last_blocks = z(capacity), z(capacity)
        self.batch = 0
        self.key: Optional[Tuple[str, ...]] = None
        self.h2d_bytes = 0
        self.loads = 0
        self.score_rows = None       # device int32 [batch] cached by the backend per batch key
        self.score_rows_key = None
        self.ring = torch.full((self.RING, capacity), -1, dtype=torch.int32).pin_memory()
        self._ring_np = self.ring.numpy()
        self._cols = np.arange(capacity)
        self._steps = 0
        self._events: Dict[int, "torch.cuda.Event"] = {}
        self._advance_args = None
        self.defer_advance = False

    def load(self, key, tables: np.ndarray, n_blocks: np.ndarray, kv_lens: np.ndarray,
             seq_lens: np.ndarray) -> None:
        """(Re)load rows 0..B-1 from host arrays: the batch composition changed."""
        import torch
        from . import staging
        B, m = tables.shape
        if B > self.capacity or m > self.max_blocks:
            raise RuntimeError("DeviceTableState: batch or table wider than the device state")
        if self._steps:
            # launches still queued read ring entries of the rows' previous owners
            torch.cuda.current_stream(self.device).synchronize()
            self._events.clear()
            self._steps = 0
        t, nb, kv, sl = staging.upload([np.ascontiguousarray(tables, dtype=np.int32),
                                        n_blocks.astype(np.int32), kv_lens.astype(np.int32),
                                        seq_lens.astype(np.int32)], self.device)
        self.h2d_bytes += tables.nbytes + 12 * B
        self.loads += 1
        self.tables[:B].zero_()
        if m:
            self.tables[:B, :m].copy_(t)
        self.num_blocks[:B].copy_(nb)
        self.kv_lens[:B].copy_(kv)
        self.seq_lens[:B].copy_(sl)
        self.batch, self.key = B, key

    def advance(self, new_blocks: np.ndarray, logical_lens: np.ndarray, evict_rule: int,
                cache_budget: int, block_size: int, evict_freq: int = 0, num_block_merge: int = 2,
                initial_blocks: int = 1) -> None:
        """``logical_lens``: the sequences' logical lengths with this step's token (what the
        device's counters will read after the step)."""
        import torch
        B = len(new_blocks)
        t = self._steps
        if t % self._GUARD == 0:
            # everything submitted so far covers the launches of the steps before this one
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._events[t] = ev
            old = self._events.pop(t - 2 * self._GUARD, None)
            if old is not None:
                old.synchronize()     # the device is at most RING - _GUARD steps behind the host
        self._steps = t + 1
        self._ring_np[np.asarray(logical_lens) % self.RING, self._cols[:B]] = new_blocks
        self.h2d_bytes += 4 * B
        self._advance_args = (evict_rule, cache_budget, block_size, evict_freq, num_block_merge,
                              initial_blocks)
        if not self.defer_advance:
            self.launch_advance()

    def launch_advance(self) -> None:
        """The step's launch on the current stream (or into the graph being captured)."""
        from . import ops
        ops.table_state_advance(self.tables, self.num_blocks, self.kv_lens, self.seq_lens, self.ring,
                                self.batch, self.slot_mapping, self.attn_kv_lens, self.last_blocks,
                                *self._advance_args)


class BatchEvictionScheduler:

    def __init__(self, block_size: int, num_gpu_blocks: int, paged_evict_config=None,
                 max_seqs: int = 256, max_blocks_per_seq: int = 512, device=None,
                 device_state: Optional[DeviceTableState] = None):
        """``device``: keep the running batch's tables on that GPU (``DeviceTableState``): decode
        steps then upload one int per sequence instead of the table matrix.  The arrays below
        remain the host's record of who owns which block (free-list accounting, the blocks to
        release when a request finishes); they are brought up to date by ``commit``.
        ``device_state``: a ``DeviceTableState`` of the same geometry to adopt instead of
        allocating a new one."""
        self.device_state: Optional[DeviceTableState] = None
        if device is not None:
            if device_state is not None and (device_state.capacity, device_state.max_blocks) == \
                    (max_seqs, max_blocks_per_seq) and device_state.device == __import__("torch").device(device):
                # adopt an existing state (launches captured into HIP graphs keep reading its
                # buffers); whatever it holds belongs to nobody any more
                device_state.key = device_state.score_rows_key = None
                device_state.batch = 0
                device_state.defer_advance = False
                self.device_state = device_state
            else:
                self.device_state = DeviceTableState(max_seqs, max_blocks_per_seq, device)
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
        """Freed ids go back to the FRONT of the list one after the other, as the reference's
        allocator does (vllm/core/block/naive_block.py:139-145): the id freed last is handed out
        first."""
        n = len(ids)
        cap = len(self._free)
        self._head = (self._head - n) % cap
        self._free[(self._head + np.arange(n)) % cap] = np.asarray(ids)[::-1]
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
        self._invalidate_batch()

    def finish_request(self, request_id: str) -> None:
        row = self._row_of_req.pop(request_id)
        self._seq_of_req.pop(request_id)
        n = int(self.n_blocks[row])
        self._give(self.tables[row, :n].copy())
        self.tables[row, :n] = 0
        self.n_blocks[row] = 0
        self.num_full[row] = self.num_cached[row] = 0
        self._free_rows.append(row)
        self._invalidate_batch()

    def _invalidate_batch(self) -> None:
        """The host record changed outside a decode step (a request came or went): neither the
        cached batch rows nor the device-resident rows may be reused, even if the next decode
        batch lists the same request ids (an id finished and re-added, a lone request preempted
        and recomputed) - its blocks and lengths are not the resident ones any more."""
        self._batch_key = None
        st = self.device_state
        if st is not None:
            st.key = None
            st.score_rows_key = None

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
        st = self.device_state
        if st is not None and st.key != self._batch_key:
            # batch composition changed: the device rows are rebuilt from the host's record
            # (taken before this step's append, which the device applies itself)
            lens0 = self.n_blocks[rows]
            m0 = int(lens0.max()) if len(rows) else 0
            st.load(self._batch_key, self.tables[rows, :m0], lens0, self.num_cached[rows],
                    self.num_full[rows])
        self.num_full[rows] += 1
        self.num_cached[rows] += 1
        need = np.nonzero(self.num_cached[rows] > self.n_blocks[rows].astype(np.int64) * bs)[0]
        new_blocks = None
        if st is not None:
            new_blocks = np.full(len(rows), -1, dtype=np.int32)
        if len(need):
            r = rows[need]
            ids = self._take(len(need))
            self.tables[r, self.n_blocks[r]] = ids
            self.n_blocks[r] += 1
            if new_blocks is not None:
                new_blocks[need] = ids
        lens = self.n_blocks[rows]
        if st is not None:
            cfg = self.paged_evict_config
            if cfg is not None and cfg.cache_prune_type == "percentage":
                # num_block_merge-1 blocks leave every evict_freq blocks (percentage_decode_trigger)
                st.advance(new_blocks, self.num_full[rows], 2, 0, bs, cfg.evict_freq, cfg.num_block_merge,
                           cfg.initial_blocks)
            else:
                rule = 0
                if cfg is not None and cfg.evict_method in ("streamingLLM", "streamingLLM-1", "global",
                                                            "local"):
                    rule = 1
                st.advance(new_blocks, self.num_full[rows], rule,
                           cfg.cache_budget if cfg is not None else 0, bs)
            return DecodeBatch(list(request_ids), self._batch_seq_ids, rows, None, lens.copy(),
                               self.num_cached[rows], self.num_full[rows], device_state=st,
                               host_tables=self.tables)
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
        freed = [self._free_column(reqids_to_rmv_block_idx) for _ in range(n)]
        # back to the free list request by request (all of one request's blocks, then the next
        # request's), the order in which the engine's loop frees them (llm_engine.py:1189-1193)
        self._give(np.stack(freed, axis=1).reshape(-1))

    def _free_column(self, reqids_to_rmv_block_idx: Dict[int, Tuple[str, int]]) -> np.ndarray:
        """Close the reported column of every listed request; returns the block ids it held."""
        bs = self.block_size
        items = list(reqids_to_rmv_block_idx.values())
        rows = np.fromiter((self._row_of_req[r] for r, _ in items), dtype=np.int64, count=len(items))
        idx = np.fromiter((i for _, i in items), dtype=np.int64, count=len(items))
        n = self.n_blocks[rows].astype(np.int64)
        # tokens held by the victim: a full block unless it is the (possibly partial) last one
        fill = np.where(idx < n - 1, bs, self.num_cached[rows] - (n - 1) * bs)
        ids = self.tables[rows, idx].copy()
        m = int(n.max())
        cols = np.arange(m - 1, dtype=np.int64)[None, :]
        src = cols + (cols >= idx[:, None])
        self.tables[rows[:, None], cols] = self.tables[rows[:, None], src]
        self.tables[rows, n - 1] = 0  # keep the padding past a row's end zero
        self.n_blocks[rows] -= 1
        self.num_cached[rows] -= fill
        return ids
