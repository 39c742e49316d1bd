"""PagedEvictBackend — the attention backend of the paged-KV-eviction path on MI355X.

Drop-in for the reference's ``TorchCUDASDPABackend`` family
(vllm/attention/backends/torch_cuda_sdpa.py:43-1586) behind the ``vllm.attention`` backend
API (vllm/attention/backends/abstract.py:32-258): same static methods, the same metadata
fields (``seq_kv_lens``, ``seq_kv_lens_tensor``, ``reqid_mapping_blocktables`` included, so the
driver->worker broadcast of metadata kwargs round-trips), the same builder entry points and
``AttentionImpl.forward`` / ``get_rmv_block_idx`` contract.  When vLLM is importable the
classes subclass its abstract bases; otherwise they are duck-typed.

What differs from the reference by design:
  * prefill, decode, KV write, prompt prune, block scoring, victim choice and block-table
    compaction are HIP kernels (one launch each) instead of eager torch + Python loops;
  * the decode-step eviction decision is taken once per step (first layer that runs) and
    shared by all layers through the step's metadata object; ``CompatFlags.per_layer_decision``
    restores the reference's one-decision-per-layer behaviour;
  * evicted slots are zeroed by the KV-write kernel (``zero_slots``) instead of appending zero
    rows from a TmpCacheSingleton scratch tensor (vllm/worker/tmp_cache_singleton.py).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, fields
from itertools import accumulate
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np
import torch

from . import ops, parallel, staging
from .block_l2norm_manager import BlockL2NormManager
from .config import CompatFlags, PagedEvictConfig
from .page_evict_kv_util import adjust_seq_kv_lens
from .paged_attn import PagedAttention
from .pruner import KVCachePruner

try:  # pragma: no cover - exercised only inside a vLLM installation
    from vllm.attention.backends.abstract import (AttentionBackend as _BackendBase,
                                                  AttentionImpl as _ImplBase,
                                                  AttentionMetadata as _MetadataBase,
                                                  AttentionMetadataBuilder as _BuilderBase)
    from vllm.attention.backends.utils import CommonAttentionState as _StateCls
    _HAVE_VLLM = True
except Exception:  # vLLM absent: plain base classes
    _BackendBase = _ImplBase = _BuilderBase = object
    _MetadataBase = object
    _StateCls = None
    _HAVE_VLLM = False

PAD_SLOT_ID = -1


class AttentionType:
    DECODER = "decoder"
    ENCODER = "encoder"
    ENCODER_ONLY = "encoder_only"
    ENCODER_DECODER = "encoder_decoder"


class PagedEvictBackend(_BackendBase):

    @staticmethod
    def get_name() -> str:
        return "PAGED_EVICT_HIP"

    @staticmethod
    def get_impl_cls() -> Type["PagedEvictImpl"]:
        return PagedEvictImpl

    @staticmethod
    def get_metadata_cls() -> Type["PagedEvictMetadata"]:
        return PagedEvictMetadata

    @staticmethod
    def get_state_cls():
        if _StateCls is None:
            raise RuntimeError("CommonAttentionState needs a vLLM installation")
        return _StateCls

    @staticmethod
    def get_builder_cls() -> Type["PagedEvictMetadataBuilder"]:
        return PagedEvictMetadataBuilder

    @classmethod
    def make_metadata(cls, *args, **kwargs) -> "PagedEvictMetadata":
        return cls.get_metadata_cls()(*args, **kwargs)

    @classmethod
    def make_metadata_builder(cls, *args, **kwargs) -> "PagedEvictMetadataBuilder":
        return cls.get_builder_cls()(*args, **kwargs)

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        PagedAttention.copy_blocks(kv_caches, src_to_dists)


# ------------------------------------------------------------------------------------------
@dataclass
class PagedEvictMetadata(_MetadataBase):
    """Per-step attention metadata; field names follow TorchCUDASDPAMetadata
    (torch_cuda_sdpa.py:91-162)."""
    num_prefills: int
    num_prefill_tokens: int
    num_decode_tokens: int
    slot_mapping: Optional[torch.Tensor]
    multi_modal_placeholder_index_maps: Optional[Dict[str, Any]] = None
    seq_lens_tensor: Optional[torch.Tensor] = None
    max_decode_seq_len: int = 0
    block_tables: Optional[torch.Tensor] = None
    max_prefill_seq_len: int = 0
    use_cuda_graph: bool = False
    seq_lens: Optional[List[int]] = None
    seq_start_loc: Optional[torch.Tensor] = None
    context_lens_tensor: Optional[torch.Tensor] = None
    max_query_len: Optional[int] = None
    max_decode_query_len: Optional[int] = None
    query_start_loc: Optional[torch.Tensor] = None
    _cached_prefill_metadata: Optional["PagedEvictMetadata"] = None
    _cached_decode_metadata: Optional["PagedEvictMetadata"] = None
    encoder_seq_lens: Optional[List[int]] = None
    encoder_seq_lens_tensor: Optional[torch.Tensor] = None
    max_encoder_seq_len: Optional[int] = None
    num_encoder_tokens: Optional[int] = None
    cross_slot_mapping: Optional[torch.Tensor] = None
    cross_block_tables: Optional[torch.Tensor] = None
    # eviction: physical KV length per sequence (list + device tensor) and the request ->
    # {seq_id: block table} map of the step
    seq_kv_lens: Optional[List[int]] = None
    seq_kv_lens_tensor: Optional[torch.Tensor] = None
    reqid_mapping_blocktables: Optional[Dict[str, Dict[int, List[int]]]] = None

    def __post_init__(self):
        self.attn_bias = None
        # per-step scratch shared by all layers (decision, kept-row index, prefix sums)
        self._pe_step: Dict[str, Any] = {}
        # set by build_decode_batch: the step's DecodeBatch (request ids, table lengths as arrays)
        self._pe_batch = None

    def asdict_zerocopy(self, skip_fields=None) -> Dict[str, Any]:
        skip = skip_fields or set()
        return {f.name: getattr(self, f.name) for f in fields(self) if f.name not in skip}

    @property
    def prefill_metadata(self) -> Optional["PagedEvictMetadata"]:
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is not None:
            return self._cached_prefill_metadata
        n, nt = self.num_prefills, self.num_prefill_tokens

        def cut(x, k):
            return None if x is None else x[:k]

        self._cached_prefill_metadata = PagedEvictMetadata(
            num_prefills=n, num_prefill_tokens=nt, num_decode_tokens=0,
            slot_mapping=cut(self.slot_mapping, nt),
            multi_modal_placeholder_index_maps=self.multi_modal_placeholder_index_maps,
            seq_lens=cut(self.seq_lens, n), seq_lens_tensor=cut(self.seq_lens_tensor, n),
            max_query_len=self.max_query_len, max_prefill_seq_len=self.max_prefill_seq_len,
            max_decode_seq_len=0,
            query_start_loc=cut(self.query_start_loc, n + 1),
            context_lens_tensor=cut(self.context_lens_tensor, n),
            block_tables=cut(self.block_tables, n), use_cuda_graph=False,
            seq_kv_lens=cut(self.seq_kv_lens, n),
            seq_kv_lens_tensor=cut(self.seq_kv_lens_tensor, n),
            reqid_mapping_blocktables=self.reqid_mapping_blocktables)
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["PagedEvictMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is not None:
            return self._cached_decode_metadata
        n, nt = self.num_prefills, self.num_prefill_tokens

        def tail(x, k):
            return None if x is None else x[k:]

        self._cached_decode_metadata = PagedEvictMetadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
            slot_mapping=tail(self.slot_mapping, nt), multi_modal_placeholder_index_maps=None,
            seq_lens=tail(self.seq_lens, n), seq_lens_tensor=tail(self.seq_lens_tensor, n),
            max_prefill_seq_len=0, max_decode_seq_len=self.max_decode_seq_len,
            block_tables=tail(self.block_tables, n), use_cuda_graph=self.use_cuda_graph,
            seq_kv_lens=tail(self.seq_kv_lens, n),
            seq_kv_lens_tensor=tail(self.seq_kv_lens_tensor, n),
            reqid_mapping_blocktables=self.reqid_mapping_blocktables)
        return self._cached_decode_metadata

    def get_seq_lens(self, attn_type: str = AttentionType.DECODER):
        """(query lengths, kv lengths); kv lengths are the pruned ones under eviction."""
        if attn_type not in (AttentionType.DECODER, AttentionType.ENCODER_ONLY):
            raise AttributeError(f"Invalid attention type {attn_type} for paged eviction")
        return self.seq_lens, (self.seq_lens if self.seq_kv_lens is None else self.seq_kv_lens)

    def get_seq_len_block_table_args(self, attn_type: str = AttentionType.DECODER):
        if attn_type not in (AttentionType.DECODER, AttentionType.ENCODER_ONLY):
            raise AttributeError(f"Invalid attention type {attn_type} for paged eviction")
        return self.seq_lens_tensor, self.max_decode_seq_len, self.block_tables


# ------------------------------------------------------------------------------------------
def compute_slot_mapping(is_profile_run: bool, is_prompt: bool, slot_mapping: List[int],
                         seq_id: int, query_len: int, context_len: int, block_size: int,
                         block_tables: Dict[int, List[int]]) -> None:
    """Slot ids of a sequence under eviction (torch_cuda_sdpa.py:565-598): a prompt fills every
    slot of its (already pruned) block table, the last block partially when the prompt length
    is not a multiple of the block size; a decode step writes slot context_len % block_size of
    the last block."""
    if is_profile_run:
        slot_mapping.extend([PAD_SLOT_ID] * query_len)
        return
    table = block_tables[seq_id]
    if is_prompt:
        rem = query_len % block_size
        full = table if rem == 0 else table[:-1]
        for b in full:
            slot_mapping.extend(range(b * block_size, (b + 1) * block_size))
        if rem:
            slot_mapping.extend(range(table[-1] * block_size, table[-1] * block_size + rem))
    else:
        assert query_len == 1, "decode request should have query_len=1 when paged_evict is enabled"
        slot_mapping.append(table[-1] * block_size + context_len % block_size)


class PagedEvictMetadataBuilder(_BuilderBase):
    """Builds PagedEvictMetadata from the model runner's per-sequence-group data
    (torch_cuda_sdpa.py:461-697).  Reads from ``input_builder``: ``runner`` (``device``,
    ``cache_config.paged_evict_config``), ``block_size``, ``sliding_window``,
    ``chunked_prefill_enabled`` and ``inter_data_list[*].{request_id, is_prompt, block_tables,
    seq_kv_lens, seq_ids, input_tokens, orig_seq_lens, seq_lens, query_lens, context_lens}``."""

    def __init__(self, input_builder) -> None:
        self.slot_mapping: List[int] = []
        self.prefill_seq_lens: List[int] = []
        self.context_lens: List[int] = []
        self.block_tables: List[List[int]] = []
        self.curr_seq_lens: List[int] = []
        self.num_prefills = 0
        self.num_prefill_tokens = 0
        self.num_decode_tokens = 0
        self.input_builder = input_builder
        self.runner = input_builder.runner
        self.sliding_window = input_builder.sliding_window
        self.block_size = input_builder.block_size
        self.paged_evict_config = self.runner.cache_config.paged_evict_config
        self.seq_kv_lens: List[int] = []
        self.reqid_mapping_blocktables: Dict[str, Dict[int, List[int]]] = {}

    def _add_seq_group(self, inter_data, chunked_prefill_enabled: bool) -> None:
        if chunked_prefill_enabled and self.paged_evict_config is not None:
            raise NotImplementedError(
                "Chunked prefill is not supported when paged eviction is enabled.")
        if self.sliding_window is not None and self.paged_evict_config is not None:
            raise AssertionError("Sliding window is not supported when paged_evict is enabled")
        is_prompt = inter_data.is_prompt
        block_tables = inter_data.block_tables
        self.reqid_mapping_blocktables[inter_data.request_id] = block_tables
        is_profile_run = block_tables is None or all(not t for t in block_tables.values())
        for seq_id, tokens, seq_len, curr_seq_len, query_len, context_len in zip(
                inter_data.seq_ids, inter_data.input_tokens, inter_data.orig_seq_lens,
                inter_data.seq_lens, inter_data.query_lens, inter_data.context_lens):
            self.context_lens.append(context_len)
            if is_prompt:
                self.num_prefills += 1
                self.num_prefill_tokens += len(tokens)
                self.prefill_seq_lens.append(seq_len)
            else:
                assert query_len == 1, (
                    f"seq_len: {seq_len}, context_len: {context_len}, query_len: {query_len}")
                self.num_decode_tokens += query_len
                self.curr_seq_lens.append(curr_seq_len)
            table: List[int] = []
            if not is_prompt and block_tables is not None:
                table = block_tables[seq_id]
            self.block_tables.append(table)
            compute_slot_mapping(is_profile_run, is_prompt, self.slot_mapping, seq_id, query_len,
                                 context_len, self.block_size, block_tables)
        if getattr(inter_data, "seq_kv_lens", None) is not None:
            self.seq_kv_lens.extend(inter_data.seq_kv_lens.values())

    def build(self, seq_lens: List[int], query_lens: List[int], cuda_graph_pad_size: int,
              batch_size: int) -> PagedEvictMetadata:
        if cuda_graph_pad_size != -1:
            # the reference's graph-capture metadata carries no seq_kv_lens either
            # (vllm/attention/backends/utils.py:316-336); every launch script enforces eager
            raise NotImplementedError("paged eviction runs in eager mode (--enforce-eager)")
        for inter_data in self.input_builder.inter_data_list:
            self._add_seq_group(inter_data,
                                getattr(self.input_builder, "chunked_prefill_enabled", False))
        device = self.runner.device
        # block tables as one padded int32 matrix; every array of the step goes up in one
        # asynchronous copy through pinned memory
        max_blocks = max((len(t) for t in self.block_tables), default=0)
        pad = [0] * max_blocks
        bt_np = np.array([t + pad[len(t):] for t in self.block_tables], dtype=np.int32).reshape(
            len(self.block_tables), max_blocks)
        i32 = np.int32
        (slot_mapping_t, seq_lens_t, query_start_t, seq_start_t, context_lens_t, bt,
         seq_kv_lens_t) = staging.upload([
             np.asarray(self.slot_mapping, dtype=np.int64),
             np.asarray(seq_lens, dtype=i32),
             np.asarray(list(accumulate(query_lens, initial=0)), dtype=i32),
             np.asarray(list(accumulate(seq_lens, initial=0)), dtype=i32),
             np.asarray(self.context_lens, dtype=i32),
             bt_np,
             np.asarray(self.seq_kv_lens, dtype=i32)], device)

        return PagedEvictMetadata(
            num_prefills=self.num_prefills,
            slot_mapping=slot_mapping_t,
            multi_modal_placeholder_index_maps=None,
            num_prefill_tokens=self.num_prefill_tokens,
            num_decode_tokens=self.num_decode_tokens,
            seq_lens=seq_lens,
            seq_lens_tensor=seq_lens_t,
            max_query_len=max(query_lens),
            max_prefill_seq_len=max(self.prefill_seq_lens, default=0),
            max_decode_seq_len=max(self.curr_seq_lens, default=0),
            query_start_loc=query_start_t,
            seq_start_loc=seq_start_t,
            context_lens_tensor=context_lens_t,
            block_tables=bt,
            use_cuda_graph=False,
            seq_kv_lens=self.seq_kv_lens,
            seq_kv_lens_tensor=seq_kv_lens_t,
            reqid_mapping_blocktables=self.reqid_mapping_blocktables)


    def build_decode_batch(self, batch, seq_kv_lens=None) -> PagedEvictMetadata:
        """Metadata of a pure decode step straight from a ``batch_scheduler.DecodeBatch`` - the
        same fields ``build`` derives sequence by sequence, computed with array operations.
        ``seq_kv_lens``: the physical lengths attention reads; by default ``batch.kv_lens`` minus
        what this step evicts, the adjustment the model runner makes before attention
        (model_runner.py:538-543, ``page_evict_kv_util.adjust_seq_kv_lens``)."""
        bs = self.block_size
        if seq_kv_lens is None:
            seq_kv_lens = adjust_seq_kv_lens(self.paged_evict_config, batch.seq_lens, batch.kv_lens, bs)
        B = len(batch.request_ids)
        st = getattr(batch, "device_state", None)
        if st is not None:
            # device-resident tables: the step's append, slot mapping and kv lengths were computed
            # on the GPU by the scheduler's advance launch; nothing is uploaded here.  The prefix
            # sums and context lengths, which no decode launch reads, are left out.
            kv_list = np.asarray(seq_kv_lens, dtype=np.int64).tolist()
            meta = PagedEvictMetadata(
                num_prefills=0, slot_mapping=st.slot_mapping[:B], multi_modal_placeholder_index_maps=None,
                num_prefill_tokens=0, num_decode_tokens=B, seq_lens=batch.seq_lens.tolist(),
                seq_lens_tensor=st.seq_lens[:B], max_query_len=1, max_prefill_seq_len=0,
                max_decode_seq_len=int(batch.seq_lens.max()) if B else 0, query_start_loc=None,
                seq_start_loc=None, context_lens_tensor=None, block_tables=st.tables[:B],
                use_cuda_graph=False, seq_kv_lens=kv_list, seq_kv_lens_tensor=st.attn_kv_lens[:B],
                reqid_mapping_blocktables=None)
            meta._pe_batch = batch
            return meta
        device = self.runner.device
        i32 = np.int32
        kv = np.asarray(seq_kv_lens, dtype=np.int64)
        ctx = batch.kv_lens - 1
        last = batch.block_tables[np.arange(B), batch.table_lens - 1].astype(np.int64)
        slots = last * bs + ctx % bs                      # torch_cuda_sdpa.py:590-597
        seq_lens = batch.seq_lens
        (slot_mapping_t, seq_lens_t, query_start_t, seq_start_t, context_lens_t, bt,
         seq_kv_lens_t) = staging.upload([
             slots, seq_lens.astype(i32), np.arange(B + 1, dtype=i32),
             np.concatenate(([0], np.cumsum(seq_lens))).astype(i32), ctx.astype(i32),
             np.ascontiguousarray(batch.block_tables, dtype=i32), kv.astype(i32)], device)
        meta = PagedEvictMetadata(
            num_prefills=0, slot_mapping=slot_mapping_t, multi_modal_placeholder_index_maps=None,
            num_prefill_tokens=0, num_decode_tokens=B, seq_lens=seq_lens.tolist(),
            seq_lens_tensor=seq_lens_t, max_query_len=1, max_prefill_seq_len=0,
            max_decode_seq_len=int(seq_lens.max()) if B else 0, query_start_loc=query_start_t,
            seq_start_loc=seq_start_t, context_lens_tensor=context_lens_t, block_tables=bt,
            use_cuda_graph=False, seq_kv_lens=kv.tolist(), seq_kv_lens_tensor=seq_kv_lens_t,
            reqid_mapping_blocktables=None)
        meta._pe_batch = batch
        return meta


# ------------------------------------------------------------------------------------------
class _LayerCaches:
    """The KV caches of the layers that share one cache configuration, as each layer's forward
    last saw them, and their addresses as device arrays: lets the first layer of a decode step
    zero the step's evicted slots in every layer's cache with one launch (the slots are the same
    for all layers), so that each layer keeps its one-launch decode step."""

    def __init__(self):
        self.caches: Dict[str, torch.Tensor] = {}      # by layer name (unique per layer and model)
        self._built = None      # (key ptrs, value ptrs, set of cache addresses, shape, dtype)

    def note(self, layer: str, kv_cache: torch.Tensor) -> None:
        old = self.caches.get(layer)
        if old is None or old.data_ptr() != kv_cache.data_ptr() or old.shape != kv_cache.shape:
            self.caches[layer] = kv_cache
            self._built = None

    def arrays(self, like: torch.Tensor):
        """Pointer arrays over the registered caches that have ``like``'s shape, dtype and device."""
        b = self._built
        if b is None or b[3] != like.shape or b[4] != like.dtype or b[5] != like.device:
            same = [c for _, c in sorted(self.caches.items())
                    if c.shape == like.shape and c.dtype == like.dtype and c.device == like.device
                    and c.is_contiguous()]
            v_off = like.stride(0) * like.element_size()
            kp = torch.tensor([c.data_ptr() for c in same], dtype=torch.int64).to(like.device)
            vp = torch.tensor([c.data_ptr() + v_off for c in same], dtype=torch.int64).to(like.device)
            b = self._built = (kp, vp, {c.data_ptr() for c in same}, like.shape, like.dtype, like.device)
        return b[0], b[1], b[2]


def _layer_caches_of(cache_config) -> Optional[_LayerCaches]:
    """The registry lives on the cache configuration object the layers share (and dies with it)."""
    reg = getattr(cache_config, "_pe_layer_caches", None)
    if reg is None and cache_config is not None:
        try:
            reg = cache_config._pe_layer_caches = _LayerCaches()
        except (AttributeError, TypeError):     # a configuration object that takes no attributes
            return None
    return reg


class DeviceScoreTable:
    """Block scores of the running sequences, resident on the GPU: f32 [rows, width] plus a
    count per row; (request, sequence) pairs are mapped to rows on the host.  The device-side
    counterpart of BlockL2NormManager's per-request lists."""

    def __init__(self, rows: int, width: int, device):
        self.scores = torch.zeros((rows, width), dtype=torch.float32, device=device)
        self.counts = torch.zeros(rows, dtype=torch.int32, device=device)
        self._free = list(range(rows - 1, -1, -1))
        self._row: Dict[Tuple[str, int], int] = {}
        self._new: List[int] = []

    def row_of(self, key: Tuple[str, int]) -> int:
        """Row of a sequence; a sequence seen for the first time gets a free row whose count is
        cleared by the next ``flush_new_rows``."""
        r = self._row.get(key)
        if r is None:
            if not self._free:
                raise RuntimeError("DeviceScoreTable: more live sequences than rows")
            r = self._free.pop()
            self._row[key] = r
            self._new.append(r)
        return r

    def flush_new_rows(self) -> None:
        """Zero the counts of rows handed out since the last flush (one launch for all of them)."""
        if self._new:
            idx = staging.upload([np.asarray(self._new, dtype=np.int64)], self.counts.device)[0]
            self.counts.index_fill_(0, idx, 0)
            self._new = []

    def set_counts(self, rows: List[int], counts: List[int]) -> None:
        idx, val = staging.upload([np.asarray(rows, dtype=np.int64),
                                   np.asarray(counts, dtype=np.int32)], self.counts.device)
        self.counts.index_copy_(0, idx, val)
        self._new = [r for r in self._new if r not in set(rows)]

    def knows(self, key: Tuple[str, int]) -> bool:
        return key in self._row

    def retain_only(self, live: List[Tuple[str, int]]) -> None:
        live_set = set(live)
        for key in [k for k in self._row if k not in live_set]:
            self._free.append(self._row.pop(key))


class PagedEvictImpl(_ImplBase):
    """AttentionImpl of the path.  Constructor and ``forward`` arguments as in
    abstract.py:222-258 / torch_cuda_sdpa.py:730-836; ``kwargs['cache_config']`` carries the
    block size and the PagedEvictConfig, ``kwargs['prefix']`` the layer name."""

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 blocksparse_params: Optional[Dict[str, Any]] = None,
                 logits_soft_cap: Optional[float] = None, **kwargs) -> None:
        if blocksparse_params is not None:
            raise ValueError("Paged-eviction attention does not support block-sparse attention.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.alibi_slopes = (None if alibi_slopes is None else
                             torch.tensor(alibi_slopes, dtype=torch.float32))
        self.sliding_window = sliding_window
        self.kv_cache_dtype = kv_cache_dtype
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        if head_size not in PagedAttention.get_supported_head_sizes():
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {PagedAttention.get_supported_head_sizes()}.")
        if kv_cache_dtype != "auto":
            raise NotImplementedError("Paged-eviction attention does not support FP8 KV cache.")
        if alibi_slopes is not None or sliding_window is not None:
            # prefill with a bias is outside the eviction path (Llama/OPT use neither)
            raise NotImplementedError("ALiBi / sliding-window prefill is not implemented")

        self.cache_config = kwargs.get("cache_config", None)
        self.compat: CompatFlags = kwargs.get("compat", None) or CompatFlags()
        self.tp_group = kwargs.get("tp_group", None)
        self.enabled_paged_eviction = False
        self.kv_cache_pruner: Optional[KVCachePruner] = None
        self.paged_evict_config: Optional[PagedEvictConfig] = None
        if self.cache_config is not None and self.cache_config.paged_evict_config is not None:
            self.enabled_paged_eviction = True
            self.paged_evict_config = self.cache_config.paged_evict_config
            # README-era command lines name the prompt and the decode method separately
            # (--prompt-evict-method / --decode-evict-method); HEAD has one evict_method for both
            pec = self.paged_evict_config
            if pec.cache_prune_type == "percentage":
                self.kv_cache_pruner = KVCachePruner(
                    "percentage", pec.prompt_evict_method, self.cache_config.block_size,
                    pec.cache_budget, pec.initial_blocks,
                    cross_head_compat=self.compat.prompt_prune_cross_head, evict_freq=pec.evict_freq,
                    num_block_merge=pec.num_block_merge,
                    decode_evict_method=pec.decode_evict_method)
            else:
                prompt_method = kwargs.get("prompt_evict_method") or pec.evict_method
                self.kv_cache_pruner = KVCachePruner(
                    pec.cache_prune_type, prompt_method, self.cache_config.block_size,
                    pec.cache_budget, pec.initial_blocks,
                    cross_head_compat=self.compat.prompt_prune_cross_head)
        self.block_l2norm_mgr = BlockL2NormManager()
        self.prefix: str = kwargs.get("prefix", "")
        self.layer_idx = _layer_index(self.prefix)
        random.seed(1234)  # the reference seeds the global RNG in every impl (:796)
        self.reqids_mapping_rmv_blockIdx: Dict[int, Tuple[str, int]] = {}
        self._pending_rmv: Optional[torch.Tensor] = None
        self._pending_readback: Optional[staging.AsyncReadback] = None
        self._pending_reqs: Optional[List[str]] = None
        self._score_table: Optional[DeviceScoreTable] = None
        # tensor parallel: KV heads of the whole model, and whether this rank is the one that
        # accounts for its KV heads when they are replicated (more ranks than KV heads)
        self.total_num_kv_heads = int(kwargs.get("total_num_kv_heads", 0)) or (
            self.num_kv_heads * (parallel.group_size(self.tp_group) if self.tp_group is not None else 1))
        self.kv_head_owner = bool(kwargs.get("kv_head_owner", True))
        self.max_num_seqs = int(kwargs.get("max_num_seqs", 256))
        self.max_blocks_per_seq = int(kwargs.get("max_blocks_per_seq", 4096))

    # ---------------------------------------------------------------------------------- API
    def get_rmv_block_idx(self) -> Dict[int, Tuple[str, int]]:
        """{batch index: (request id, block index to free)} decided in the last forward."""
        if self._pending_rmv is not None:
            # the copy was queued right behind the decision kernel: wait for it alone, not for
            # the attention launches of the step queued after it
            rb = self._pending_readback
            rmv = rb.tolist() if rb is not None else self._pending_rmv.cpu().tolist()
            self._pending_rmv = self._pending_readback = None
            reqs = self._pending_reqs
            self.reqids_mapping_rmv_blockIdx = {i: (reqs[i], r) for i, r in enumerate(rmv) if r >= 0}
        return self.reqids_mapping_rmv_blockIdx

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: torch.Tensor, attn_metadata: PagedEvictMetadata, k_scale: float = 1.0,
                v_scale: float = 1.0, attn_type: str = AttentionType.DECODER,
                output: Optional[torch.Tensor] = None,
                rope: Optional[Tuple[torch.Tensor, torch.Tensor, bool]] = None) -> torch.Tensor:
        """query [T, Hq*D], key/value [T, Hkv*D], kv_cache [2, nb, bs*Hkv*D] (numel()==0 in the
        profiling run) -> [T, Hq*D].

        ``rope`` = (positions int64 [T], cos_sin_cache [max_position, rot_dim], is_neox) hands
        the rotary embedding that precedes attention in the model
        (vllm/model_executor/models/llama.py:202-203) to the backend: query and key then arrive
        UNROTATED.  Prompt rows are rotated in place before the prefill path; a decode layer on
        the one-launch path rotates inside the attention kernel (pe_decode_step_rope) and leaves
        query/key as they are, any other decode layer rotates in place first."""
        assert k_scale == 1.0 and v_scale == 1.0
        if attn_type != AttentionType.DECODER:
            raise NotImplementedError("Paged eviction is only enabled for decoder attention")
        self.reqids_mapping_rmv_blockIdx = {}
        self._pending_rmv = self._pending_readback = None
        T = query.shape[0]
        # decode layers after the first of a step: metadata already validated, decision taken
        launcher = attn_metadata._pe_step.get("launcher")
        if launcher is not None and launcher.matches(query, key, value, kv_cache):
            dec = attn_metadata._pe_step.get("decode")
            if dec is not None:
                self._pending_rmv, self._pending_reqs = dec["rmv"], dec.get("req_ids")
                self._pending_readback = dec.get("readback")
                self.reqids_mapping_rmv_blockIdx = dec["host_rmv"] or {}
                if dec.get("merge") is not None:
                    kc, vc = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)
                    self.kv_cache_pruner.merge_blocks_in_cache(kc, vc, dec["merge_tables"], dec["merge"])
            zeroed = attn_metadata._pe_step.get("zeroed")
            if zeroed is not None:
                self._note_cache(kv_cache)
                if kv_cache.data_ptr() not in zeroed[0]:
                    # this layer's cache was not known when the step's slots were zeroed
                    kc, vc = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)
                    ops.reshape_and_cache(key[:0].view(0, self.num_kv_heads, self.head_size),
                                          value[:0].view(0, self.num_kv_heads, self.head_size), kc, vc,
                                          zeroed[1][:0], zero_slots=zeroed[1])
            o = torch.empty_like(query) if output is None else output
            launcher.run(query, key, value, kv_cache, o, rope)
            return o
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size)
        is_profile_run = kv_cache.numel() == 0
        out = (torch.empty_like(query) if output is None else
               output.view(-1, self.num_heads, self.head_size))
        key_cache = value_cache = None
        if not is_profile_run:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
        if attn_metadata.prefill_metadata is not None and attn_metadata.decode_metadata is not None \
                and self.enabled_paged_eviction:
            raise NotImplementedError(
                "Chunked prefill is not supported when paged eviction is enabled.")
        if self.enabled_paged_eviction and not is_profile_run and (
                self.compat.per_layer_decision or self.layer_idx <= 0
                or "mgr" not in attn_metadata._pe_step):
            # only the deciding layer tracks the batch composition
            if attn_metadata.reqid_mapping_blocktables is not None:
                self.block_l2norm_mgr.update_reqs(attn_metadata.reqid_mapping_blocktables)
            attn_metadata._pe_step.setdefault("mgr", self.block_l2norm_mgr)

        if self.enabled_paged_eviction and not is_profile_run \
                and self.paged_evict_config.cache_prune_type == "budget" \
                and self.paged_evict_config.evict_method in ("streamingLLM-1", "inverse_key_l2"):
            self._note_cache(kv_cache)      # the methods that zero evicted slots (all layers at once)
        rope_dec = rope_pre = None
        if rope is not None:
            positions, cos_sin, is_neox = rope
            nt = attn_metadata.num_prefill_tokens
            rope_pre = (positions[:nt], cos_sin, is_neox)
            rope_dec = (positions[nt:], cos_sin, is_neox)
        if (pm := attn_metadata.prefill_metadata) is not None:
            self._forward_prefill(out, query, key, value, key_cache, value_cache, attn_metadata, pm,
                                  is_profile_run, rope_pre)
        if (dm := attn_metadata.decode_metadata) is not None:
            self._forward_decode(out, query, key, value, key_cache, value_cache, attn_metadata, dm,
                                 kv_cache, rope_dec)
        return out.view(T, self.num_heads * self.head_size)

    # ------------------------------------------------------------------------------ prefill
    def _forward_prefill(self, out, query, key, value, key_cache, value_cache, meta, pm,
                         is_profile_run, rope=None) -> None:
        """``rope`` = (positions of the prompt rows, cos_sin_cache, is_neox): the prompt rows arrive
        unrotated and are rotated in place here - in the same launch as the KV write
        (pe_rope_reshape_and_cache: the two launches that precede attention in the model,
        vllm/model_executor/models/llama.py:202-205, as one) whenever the rows the cache keeps are
        known without looking at the data: no prune, or the streamingLLM prune with attention over
        the full prompt (the kept rows depend on the lengths only).  A token-level prune scores
        the ROTATED keys, and a prune in front of attention gathers rotated rows: those rotate in
        a launch of their own first."""
        nt = meta.num_prefill_tokens
        q, k, v = query[:nt], key[:nt], value[:nt]
        seq_lens = pm.seq_lens
        step = meta._pe_step
        cu_q = step.get("cu_q")
        if cu_q is None:
            cu_q = pm.query_start_loc if pm.query_start_loc is not None else _cumsum_t(seq_lens, q.device)
            step["cu_q"] = cu_q
        max_q = max(seq_lens)
        cfg = self.paged_evict_config
        pruner = self.kv_cache_pruner
        prune = self.enabled_paged_eviction and any(pruner.prunes(L) for L in seq_lens)
        fuse_rope_write = False
        if rope is not None and nt > 0:
            fuse_rope_write = not is_profile_run and (
                not prune or (cfg.disable_evict_prefill and not self.compat.prompt_prune_cross_head
                              and pruner.evict_method in ("streamingLLM", "streamingLLM-1")))
            if not fuse_rope_write:
                ops.rotary_embedding(rope[0], q, k, self.head_size, rope[1], rope[2])

        src_index = kept = None
        if prune:
            # streamingLLM index depends on lengths only: built once per step, shared by layers
            shared = pruner.evict_method in ("streamingLLM", "streamingLLM-1")
            if shared and "src_index" in step:
                src_index, kept, cu_kept = step["src_index"], step["kept"], step["cu_kept"]
            else:
                kept = [pruner.get_pruned_length(L) for L in seq_lens]
                cu_kept = step.get("cu_kept")
                if cu_kept is None:
                    cu_kept = _cumsum_t(kept, q.device)
                    step["cu_kept"] = cu_kept
                    step["kept"] = kept
                src_index, _ = pruner.keep_index(k, v, seq_lens, cu_q, cu_kept)
                if shared:
                    step["src_index"] = src_index
            pairs = src_index.dim() == 2 and self.compat.prompt_prune_cross_head

        if prune and not cfg.disable_evict_prefill:
            # evict before attention: attention sees the pruned K/V (top-left causal mask)
            pk, pv = ops.prompt_prune_gather(k, v, src_index, mode=2 if pairs else None)
            ops.prefill_causal(q, pk, pv, cu_q, cu_kept, max_q, self.scale, out=out[:nt])
            if not is_profile_run:
                PagedAttention.write_to_paged_cache(pk, pv, key_cache, value_cache,
                                                    meta.slot_mapping[:pk.shape[0]],
                                                    self.kv_cache_dtype, 1.0, 1.0)
                self._prompt_block_scores(pv, kept, meta, pk)
        else:
            if fuse_rope_write:
                # one launch: rotation of the prompt's query/key rows + KV write of the rows the
                # cache keeps (a pruned-away row has slot -1: rotated for attention, not cached)
                slots = meta.slot_mapping[:nt]
                if prune:
                    slots = step.get("rope_slots")
                    if slots is None:
                        n_kept = src_index.shape[0]
                        slots = torch.full((nt,), -1, dtype=torch.int64, device=q.device)
                        slots.index_copy_(0, src_index.to(torch.int64), meta.slot_mapping[:n_kept])
                        step["rope_slots"] = slots      # lengths only: shared by the step's layers
                ops.rotary_embedding_and_cache(rope[0], q, k, v, key_cache, value_cache, slots,
                                               self.head_size, rope[1], rope[2])
            ops.prefill_causal(q, k, v, cu_q, cu_q, max_q, self.scale, out=out[:nt])
            if not is_profile_run:
                if fuse_rope_write:
                    if prune:
                        if self._needs_scores():
                            pk, pv = ops.prompt_prune_gather(k, v, src_index, mode=None)
                            self._prompt_block_scores(pv, kept, meta, pk)
                    elif self.enabled_paged_eviction:
                        self._prompt_block_scores(v, list(seq_lens), meta, k)
                elif prune:
                    n_kept = src_index.shape[0]
                    PagedAttention.write_to_paged_cache(
                        k, v, key_cache, value_cache, meta.slot_mapping[:n_kept],
                        self.kv_cache_dtype, 1.0, 1.0, src_index=src_index, src_pairs=pairs)
                    if self._needs_scores():
                        pk, pv = ops.prompt_prune_gather(k, v, src_index, mode=2 if pairs else None)
                        self._prompt_block_scores(pv, kept, meta, pk)
                else:
                    PagedAttention.write_to_paged_cache(k, v, key_cache, value_cache,
                                                        meta.slot_mapping[:nt], self.kv_cache_dtype,
                                                        1.0, 1.0)
                    if self.enabled_paged_eviction:
                        self._prompt_block_scores(v, list(seq_lens), meta, k)

    def _step_keys(self, meta) -> List[Tuple[str, int]]:
        """(request id, sequence id) of every sequence of the step, in batch order."""
        step = meta._pe_step
        keys = step.get("keys")
        if keys is None:
            batch = getattr(meta, "_pe_batch", None)
            if batch is not None:
                keys = list(zip(batch.request_ids, batch.seq_ids))
            elif self.compat.request_order_first_seen or meta.reqid_mapping_blocktables is None:
                mgr = step.get("mgr", self.block_l2norm_mgr)
                keys = list(mgr.flat_sequence_map)
            else:
                keys = [(r, s) for r, tables in meta.reqid_mapping_blocktables.items() for s in tables]
            step["keys"] = keys
        return keys

    def _tp_size(self) -> int:
        return parallel.group_size(self.tp_group) if self.tp_group is not None else 1

    def _tp_exact_scores(self) -> bool:
        """Tensor parallel with rank-invariant scores: every rank keeps the full-head score table
        (token sums all-reduced).  The flat cache view mixes heads and tokens of the local shard,
        so that compatibility mode falls back to rank 0 deciding and broadcasting."""
        return self._tp_size() > 1 and not self.compat.score_flat_cache_view

    def _reduce_token_sums(self, sums: torch.Tensor) -> torch.Tensor:
        if not self.kv_head_owner:
            sums.zero_()
        return parallel.all_reduce_sum(sums, self.tp_group)

    def _needs_scores(self) -> bool:
        if self._tp_size() > 1 and not self._tp_exact_scores() \
                and parallel.group_rank(self.tp_group) != 0:
            return False
        return (self.enabled_paged_eviction and not self.compat.random_victim
                and self.paged_evict_config.cache_prune_type == "budget"
                and self.paged_evict_config.evict_method in ("local", "global")
                and (self.layer_idx <= 0 or self.compat.per_layer_decision))

    def _table(self, device) -> DeviceScoreTable:
        if self._score_table is None:
            self._score_table = DeviceScoreTable(2 * self.max_num_seqs, self.max_blocks_per_seq, device)
        return self._score_table

    def _score_formula(self) -> str:
        """Score formula of global/local: the pruner's sub method (kvcache_prunner.py:13,51-66)."""
        f = getattr(self.kv_cache_pruner, "sub_evict_method", "value_l2")
        if f not in ops.SCORE_FORMULAS:
            raise ValueError(f"Unknown sub eviction method: {f}")
        return f

    def _norm_block_scores(self, tensor: torch.Tensor, ids: torch.Tensor, n: int, layout: int,
                           block_stride: int) -> torch.Tensor:
        """L2-norm block scores (norm, head mean, token sum) of n blocks of the key or the value
        tensor: one launch, or under rank-invariant tensor-parallel scoring the token sums of
        the local heads, all-reduced, then finished."""
        bs = self.cache_config.block_size
        H, D = self.num_kv_heads, self.head_size
        if self._tp_exact_scores():
            sums = ops.block_token_norm_sums(tensor, ids, n, layout, H, D, bs, block_stride=block_stride)
            self._reduce_token_sums(sums)
            return ops.block_score_from_token_sums(sums, n, bs, self.total_num_kv_heads, tensor.dtype)
        return ops.block_score_value_l2(tensor, ids, n, layout, H, D, bs, block_stride=block_stride)

    def _formula_block_scores(self, keys: Optional[torch.Tensor], values: torch.Tensor,
                              ids: torch.Tensor, n: int, key_layout: int, value_layout: int,
                              block_stride: int) -> torch.Tensor:
        """Scores of n blocks under the score formula in force (f32 [n])."""
        f = self._score_formula()
        vs = ks = None
        if f in ("value_l2", "value_l2_plus_key_l2"):
            vs = self._norm_block_scores(values, ids, n, value_layout, block_stride)
        if f != "value_l2":
            if keys is None:
                raise RuntimeError(f"score formula {f} needs the keys")
            ks = self._norm_block_scores(keys, ids, n, key_layout, block_stride)
        return ops.combine_block_scores(f, vs, ks, values.dtype)

    def _new_block_scores(self, key_cache: torch.Tensor, value_cache: torch.Tensor,
                          last_blocks: torch.Tensor, n: int) -> Optional[torch.Tensor]:
        """Scores of the sequences' just-filled blocks when the decision launch cannot compute
        them itself (it scores value norms of all KV heads of this rank): rank-invariant
        tensor-parallel scoring, or a formula with key norms.  None = let the launch do it."""
        if self._score_formula() == "value_l2" and not self._tp_exact_scores():
            return None
        flat = self.compat.score_flat_cache_view
        return self._formula_block_scores(
            key_cache, value_cache, last_blocks, n,
            ops.SCORE_CACHE_FLAT if flat else ops.SCORE_KEY_CACHE,
            ops.SCORE_CACHE_FLAT if flat else ops.SCORE_CACHE, value_cache.stride(0))

    def _prompt_block_scores(self, values: torch.Tensor, kept: List[int], meta,
                             keys_dense: Optional[torch.Tensor] = None) -> None:
        """Scores of every whole block of each (pruned) prompt -> the sequence's score row
        (torch_cuda_sdpa.py:1278-1288)."""
        if not self._needs_scores():
            return
        bs = self.cache_config.block_size
        table = self._table(values.device)
        keys = self._step_keys(meta)
        if self.compat.request_order_first_seen:
            # HEAD's manager forgets every request absent from a step, and a prefill step holds
            # the new prompts only (block_l2norm_manager.py:48-53)
            table.retain_only(keys)
        H, D = values.shape[1], values.shape[2]
        if values.stride(2) != 1 or values.stride(1) != D:
            raise RuntimeError("prompt values must be dense [tokens, heads, head_size] rows")
        # One launch for the whole batch: block i of the launch starts at token offs[i] of the
        # dense value tensor (block stride = one token row, block id = token offset) and its
        # score goes to flat position idx[i] of the score table.
        width = table.scores.shape[1]
        offs, idx, rows, counts = [], [], [], []
        off = 0
        for i, n in enumerate(kept):
            nb = n // bs
            row = table.row_of(keys[i])
            if nb > 0:
                offs.extend(range(off, off + nb * bs, bs))
                idx.extend(range(row * width, row * width + nb))
            rows.append(row)
            counts.append(nb)
            off += n
        if offs:
            dev = values.device
            n_all = len(offs)
            ids = torch.tensor(offs, dtype=torch.int32, device=dev)
            dst = torch.tensor(idx, dtype=torch.int64, device=dev)
            if keys_dense is not None and (keys_dense.stride(2) != 1 or keys_dense.stride(1) != D
                                           or keys_dense.stride(0) != values.stride(0)):
                raise RuntimeError("prompt keys must be dense rows laid out like the values")
            sc = self._formula_block_scores(keys_dense, values, ids, n_all, ops.SCORE_TOKEN_MAJOR,
                                            ops.SCORE_TOKEN_MAJOR, values.stride(0))
            table.scores.view(-1).index_copy_(0, dst, sc)
        table.set_counts(rows, counts)

    # ------------------------------------------------------------------------------- decode
    def _forward_decode(self, out, query, key, value, key_cache, value_cache, meta, dm,
                        kv_cache=None, rope=None) -> None:
        n0 = meta.num_prefill_tokens
        q, k, v = query[n0:], key[n0:], value[n0:]
        block_tables = dm.block_tables
        seq_lens_arg = dm.seq_lens_tensor
        zero_slots = None
        if self.enabled_paged_eviction:
            seq_lens_arg = dm.seq_kv_lens_tensor
            dec = self._decode_decision(meta, dm, value_cache, key_cache)
            if dec["tables"] is not None:
                block_tables = dec["tables"]
            zero_slots = dec["zero_slots"]
            if zero_slots is not None and kv_cache is not None and not self.compat.per_layer_decision:
                zero_slots = self._zero_in_all_layers(meta, zero_slots, kv_cache, key_cache, value_cache,
                                                      dm.slot_mapping)
            self._pending_rmv, self._pending_reqs = dec["rmv"], dec.get("req_ids")
            self._pending_readback = dec.get("readback")
            self.reqids_mapping_rmv_blockIdx = dec["host_rmv"] or {}
            if dec.get("merge") is not None:
                # percentage type, value_l2: this layer's blocks are merged before attention reads
                # them (the step's token goes to the last block, which is never one of them)
                self.kv_cache_pruner.merge_blocks_in_cache(key_cache, value_cache,
                                                           dec["merge_tables"], dec["merge"])
        # the decode kernels take a maximum length only to size their split-KV work: give them
        # the longest PHYSICAL length (the reference passes the logical one, :446-447, which
        # over-states what is read once blocks have been evicted)
        max_len = dm.max_decode_seq_len
        if self.enabled_paged_eviction and dm.seq_kv_lens:
            max_len = max(dm.seq_kv_lens)
        # every layer of the step launches with the same metadata: validate once (first layer of
        # the step), then reuse the argument list
        step = meta._pe_step
        launcher = step.get("launcher")
        if launcher is None and "launcher" not in step and kv_cache is not None and n0 == 0 \
                and not self.compat.per_layer_decision:
            try:
                launcher = ops.DecodeLayerLauncher(q, k, v, kv_cache, self.num_heads, self.num_kv_heads,
                                                   self.head_size, dm.slot_mapping, zero_slots,
                                                   block_tables, seq_lens_arg, max_len, self.scale)
                launcher.tables, launcher.zero = block_tables, zero_slots
                if self.compat.streaming_decode_original_table:
                    # HEAD reads kv_len tokens of the ORIGINAL table, which leaves the step's own
                    # token out: the fused launch (new token attended from registers) cannot
                    # express that
                    launcher.fused = False
            except RuntimeError:
                launcher = None  # shape outside the launcher's contract: per-op path below
            step["launcher"] = launcher
        if (launcher is not None and launcher.tables is block_tables and launcher.zero is zero_slots
                and kv_cache is not None and launcher.matches(q, k, v, kv_cache)):
            launcher.run(q, k, v, kv_cache, out[n0:], rope)
            return
        if rope is not None and q.shape[0] > 0:
            ops.rotary_embedding(rope[0], q, k, self.head_size, rope[1], rope[2])
        PagedAttention.write_to_paged_cache(k, v, key_cache, value_cache, dm.slot_mapping,
                                            self.kv_cache_dtype, 1.0, 1.0, zero_slots=zero_slots)
        PagedAttention.forward_decode(q, key_cache, value_cache, block_tables, seq_lens_arg,
                                      max_len, self.kv_cache_dtype, self.num_kv_heads,
                                      self.scale, None, 1.0, 1.0, output=out[n0:])

    def _note_cache(self, kv_cache: torch.Tensor) -> None:
        if self.prefix and kv_cache.numel() > 0:
            reg = _layer_caches_of(self.cache_config)
            if reg is not None:
                reg.note(self.prefix, kv_cache)

    def _zero_in_all_layers(self, meta, zero_slots, kv_cache, key_cache, value_cache, slot_mapping):
        """The step's evicted slots (streamingLLM-1, inverse_key_l2: the same for every layer) are
        zeroed in all known layers' caches by the first layer that gets here; returns what is left
        for this layer to zero itself (None when its cache was covered)."""
        step = meta._pe_step
        self._note_cache(kv_cache)
        zeroed = step.get("zeroed")
        if zeroed is None:
            reg = _layer_caches_of(self.cache_config)
            if reg is None:
                return zero_slots
            kp, vp, covered = reg.arrays(kv_cache)
            if not covered:
                return zero_slots
            nb = kv_cache.shape[1]
            ops.zero_slots_layers(kp, vp, zero_slots, self.num_kv_heads, self.head_size,
                                  self.cache_config.block_size, nb, kv_cache.dtype)
            zeroed = step["zeroed"] = (covered, zero_slots)
        return None if kv_cache.data_ptr() in zeroed[0] else zero_slots

    def _decode_decision(self, meta, dm, value_cache, key_cache=None) -> Dict[str, Any]:
        """Eviction decision of this decode step: taken by the first layer that runs and reused
        by the others (one launch per step), unless per-layer decisions are requested."""
        step = meta._pe_step
        if not self.compat.per_layer_decision and "decode" in step:
            return step["decode"]
        cfg = self.paged_evict_config
        bs = self.cache_config.block_size
        method = cfg.evict_method
        seq_lens = dm.seq_lens
        dev = dm.block_tables.device
        mgr = step.get("mgr", self.block_l2norm_mgr)
        batch = getattr(meta, "_pe_batch", None)
        n_seqs = len(seq_lens)
        keys = self._step_keys(meta)
        req_ids = [k[0] for k in keys]
        if batch is not None:       # array view of the step (vectorised scheduler)
            L = batch.seq_lens
            table_lens = batch.table_lens.astype(np.int64)
            table_of = batch.table_row
        else:                       # per-request mapping of the reference's builder
            L = np.asarray(seq_lens, dtype=np.int64)
            mapping = meta.reqid_mapping_blocktables
            table_of = lambda i: mapping[keys[i][0]][keys[i][1]]  # noqa: E731
            table_lens = np.fromiter((len(table_of(i)) for i in range(n_seqs)), dtype=np.int64,
                                     count=n_seqs)
        if cfg.cache_prune_type == "percentage":
            dec = self._decode_decision_percentage(dm, L, table_lens, req_ids,
                                                   getattr(batch, "device_state", None))
            if not self.compat.per_layer_decision:
                step["decode"] = dec
            return dec
        if self._score_table is not None:
            # a decode step lists every running sequence: whoever is absent has finished
            self._score_table.retain_only(keys)
        if self.compat.request_order_first_seen and self._needs_scores():
            # HEAD: a request its manager does not know (new, or forgotten at a prefill step it
            # was absent from) is listed with one zero score per block of its table, at the step
            # it appears (block_l2norm_manager.py:39-44)
            table = self._table(dev)
            fresh = [i for i, k in enumerate(keys) if not table.knows(k)]
            if fresh:
                rows_f = [table.row_of(keys[i]) for i in fresh]
                table.scores[staging.upload([np.asarray(rows_f, dtype=np.int64)], dev)[0]] = 0.0
                table.set_counts(rows_f, [int(table_lens[i]) for i in fresh])
        st = getattr(batch, "device_state", None) if batch is not None else None
        if st is not None and not (self.compat.per_layer_decision or self.compat.block_table_slice_in_loop
                                   or self.compat.streaming_decode_original_table
                                   or self.compat.compacted_table_strided_view):
            dec = self._decode_decision_device(st, dm, L, table_of, keys, req_ids, value_cache,
                                               key_cache)
            step["decode"] = dec
            return dec
        if st is not None:
            raise NotImplementedError("device-resident block tables take one decision per step and "
                                      "compact in place: not available with the per-layer / "
                                      "in-loop-slice / original-table compatibility switches")
        over_np = L > cfg.cache_budget
        trig_np = over_np & (L % bs == 0)
        over, trig = over_np.tolist(), trig_np.tolist()
        any_trig = bool(trig_np.any())
        dec: Dict[str, Any] = {"rmv": None, "tables": None, "zero_slots": None, "host_rmv": None,
                               "req_ids": req_ids if any_trig else None}
        # width of the compacted table: the longest row once the victims are gone (the
        # reference always uses M-1 and so can cut a full-length row that evicts nothing)
        M = dm.block_tables.shape[1]
        width = M - 1
        if not self.compat.block_table_slice_in_loop and any_trig:
            width = int((table_lens - trig_np).max())
            width = min(max(width, 1), M)
        if method == "inverse_key_l2":
            # zero one random middle slot per over-budget sequence (torch_cuda_sdpa.py:1435-1475)
            slots = []
            for i, o in enumerate(over):
                if not o:
                    continue
                ids = table_of(i)
                slots.append(ids[random.randint(1, len(ids) - 2)] * bs + random.randint(0, bs - 1))
            if slots:
                dec["zero_slots"] = staging.upload([np.asarray(slots, dtype=np.int64)], dev)[0]
        elif method in ("streamingLLM", "streamingLLM-1"):
            need_zero = method == "streamingLLM-1" and bool((over_np & ~trig_np).any())
            if any_trig or need_zero:
                rmv, tables, zero = ops.evict_select_compact(
                    dm.block_tables, dm.seq_lens_tensor, dm.seq_lens_tensor, method,
                    cfg.cache_budget, bs, compat_q1=self.compat.block_table_slice_in_loop,
                    compact=any_trig, out_width=width)
                dec["tables"] = tables if any_trig else None
                if self.compat.streaming_decode_original_table:
                    dec["tables"] = None   # HEAD: the compacted table is built and then ignored
                dec["zero_slots"] = zero
                dec["host_rmv"] = {i: (req_ids[i], 1) for i, t in enumerate(trig) if t}
        elif any_trig:
            nblk = staging.upload_i32(table_lens, dev)
            override = None
            if self.compat.random_victim:
                ov = []
                for i, t in enumerate(trig):
                    if not t:
                        ov.append(-1)
                        continue
                    n = int(table_lens[i])
                    end = 1 + cfg.topk_blocks if method == "local" else n - 1
                    ov.append(random.randint(1, end - 1))
                override = staging.upload_i32(ov, dev)
                rmv, tables, _ = ops.evict_select_compact(
                    dm.block_tables, dm.seq_lens_tensor, nblk, method, cfg.cache_budget, bs,
                    cfg.topk_blocks, victim_override=override,
                    compat_q1=self.compat.block_table_slice_in_loop, out_width=width)
                dec["host_rmv"] = {i: (req_ids[i], r) for i, r in enumerate(ov) if r >= 0}
            elif self._tp_size() > 1 and not self._tp_exact_scores() \
                    and parallel.group_rank(self.tp_group) != 0:
                # data-dependent victims are decided by the first rank of the TP group and
                # broadcast; this rank only applies them
                rmv = torch.empty(len(seq_lens), dtype=torch.int32, device=dev)
                parallel.broadcast_victims(rmv, self.tp_group, 0)
                _, tables, _ = ops.evict_select_compact(
                    dm.block_tables, dm.seq_lens_tensor, nblk, method, cfg.cache_budget, bs,
                    cfg.topk_blocks, victim_override=rmv,
                    compat_q1=self.compat.block_table_slice_in_loop, out_width=width)
                dec["rmv"] = rmv
            else:
                table = self._table(dev)
                rows = staging.upload_i32([table.row_of(k) for k in keys], dev)
                table.flush_new_rows()
                new_scores = None
                if self.compat.request_order_first_seen:
                    # HEAD scores the last block of the table of the request batch row i is
                    # RESOLVED to (torch_cuda_sdpa.py:1501,1520), which is another request's once
                    # the manager's order has left the batch order
                    last = staging.upload_i32([table_of(i)[-1] for i in range(n_seqs)], dev)
                    flat = self.compat.score_flat_cache_view
                    new_scores = self._formula_block_scores(
                        key_cache, value_cache, last, n_seqs,
                        ops.SCORE_CACHE_FLAT if flat else ops.SCORE_KEY_CACHE,
                        ops.SCORE_CACHE_FLAT if flat else ops.SCORE_CACHE, value_cache.stride(0))
                elif self._tp_exact_scores() or self._score_formula() != "value_l2":
                    # score of each sequence's just-filled block outside the decision launch: over
                    # ALL KV heads under tensor parallelism (local token sums -> all-reduce ->
                    # finish; identical on every rank, so no broadcast), or with key norms
                    last = dm.block_tables.gather(
                        1, (nblk.to(torch.int64) - 1).clamp_(min=0).unsqueeze(1)).squeeze(1).contiguous()
                    new_scores = self._new_block_scores(key_cache, value_cache, last, len(seq_lens))
                rmv, tables, _ = ops.evict_select_compact(
                    dm.block_tables, dm.seq_lens_tensor, nblk, method, cfg.cache_budget, bs,
                    cfg.topk_blocks, drop_victim_score=not self.compat.keep_victim_score,
                    compat_q1=self.compat.block_table_slice_in_loop,
                    score_layout=(ops.SCORE_CACHE_FLAT if self.compat.score_flat_cache_view
                                  else ops.SCORE_CACHE),
                    score_table=table.scores, score_counts=table.counts, score_rows=rows,
                    value_cache=value_cache, out_width=width, new_scores=new_scores)
                if self._tp_size() > 1 and new_scores is None:
                    parallel.broadcast_victims(rmv, self.tp_group, 0)
                dec["rmv"] = rmv
            if self.compat.compacted_table_strided_view and tables is not None and tables.shape[1] == M - 1:
                # what the reference's kernels see: the [B, M] buffer (compacted columns + the
                # untouched last column) read with a row pitch of M-1
                full = torch.cat([tables, dm.block_tables[:, M - 1:M]], dim=1)
                tables = full.flatten()[:tables.shape[0] * (M - 1)].view(tables.shape[0], M - 1).contiguous()
            dec["tables"] = tables
            if dec["rmv"] is not None:
                dec["readback"] = staging.AsyncReadback(dec["rmv"])
        if not self.compat.per_layer_decision:
            step["decode"] = dec
        return dec

    def _decode_decision_device(self, st, dm, L: np.ndarray, table_of, keys, req_ids,
                                value_cache, key_cache=None) -> Dict[str, Any]:
        """The budget-type decision on device-resident tables (``DeviceTableState``): the same
        launch as ``_decode_decision`` takes, but the victim's column is closed inside the
        resident row, so there is no second table and nothing about the tables is uploaded.
        ``dm.block_tables`` IS the resident array: every layer of the step reads it after this
        launch, with ``dm.seq_kv_lens_tensor`` (already shortened by the scheduler's advance)."""
        cfg = self.paged_evict_config
        bs = self.cache_config.block_size
        method = cfg.evict_method
        dev = st.tables.device
        B = len(req_ids)
        over_np = L > cfg.cache_budget
        trig_np = over_np & (L % bs == 0)
        trig = trig_np.tolist()
        any_trig = bool(trig_np.any())
        dec: Dict[str, Any] = {"rmv": None, "tables": None, "zero_slots": None, "host_rmv": None,
                               "req_ids": req_ids if any_trig else None}
        if method == "inverse_key_l2":
            slots = []
            for i, o in enumerate(over_np.tolist()):
                if not o:
                    continue
                ids = table_of(i)
                slots.append(ids[random.randint(1, len(ids) - 2)] * bs + random.randint(0, bs - 1))
            if slots:
                dec["zero_slots"] = staging.upload([np.asarray(slots, dtype=np.int64)], dev)[0]
            return dec
        if method in ("streamingLLM", "streamingLLM-1"):
            need_zero = method == "streamingLLM-1" and bool((over_np & ~trig_np).any())
            if any_trig or need_zero:
                rmv, _, zero = ops.evict_select_compact_inplace(
                    st.tables, st.num_blocks, st.kv_lens, st.seq_lens, B, method, cfg.cache_budget, bs)
                dec["zero_slots"] = zero
                dec["host_rmv"] = {i: (req_ids[i], 1) for i, t in enumerate(trig) if t}
                dec["rmv_dev"] = rmv
            return dec
        if not any_trig:
            return dec
        override = None
        score_kw: Dict[str, Any] = {}
        if self.compat.random_victim:
            ov = []
            for i, t in enumerate(trig):
                if not t:
                    ov.append(-1)
                    continue
                n = len(table_of(i))
                end = 1 + cfg.topk_blocks if method == "local" else n - 1
                ov.append(random.randint(1, end - 1))
            override = staging.upload_i32(ov, dev)
            dec["host_rmv"] = {i: (req_ids[i], r) for i, r in enumerate(ov) if r >= 0}
        elif self._tp_size() > 1 and not self._tp_exact_scores():
            raise NotImplementedError("device-resident tables under tensor parallelism need the "
                                      "rank-invariant scores (CompatFlags.score_flat_cache_view off)")
        else:
            table = self._table(dev)
            if st.score_rows_key != st.key or st.score_rows is None:
                st.score_rows = staging.upload_i32([table.row_of(k) for k in keys], dev)
                st.score_rows_key = st.key
            table.flush_new_rows()
            new_scores = self._new_block_scores(key_cache, value_cache, st.last_blocks[:B], B)
            score_kw = dict(drop_victim_score=not self.compat.keep_victim_score,
                            score_layout=(ops.SCORE_CACHE_FLAT if self.compat.score_flat_cache_view
                                          else ops.SCORE_CACHE),
                            score_table=table.scores, score_counts=table.counts,
                            score_rows=st.score_rows, value_cache=value_cache, new_scores=new_scores)
        rmv, _, _ = ops.evict_select_compact_inplace(
            st.tables, st.num_blocks, st.kv_lens, st.seq_lens, B, method, cfg.cache_budget, bs,
            cfg.topk_blocks, victim_override=override, **score_kw)
        dec["rmv_dev"] = rmv     # the victims on the device (a caller replaying this step from a
        #                          HIP graph reads them back itself after each replay)
        if dec["host_rmv"] is None:
            dec["rmv"] = rmv
            if not torch.cuda.is_current_stream_capturing():
                dec["readback"] = staging.AsyncReadback(rmv)
        return dec

    def _decode_decision_percentage(self, dm, L: np.ndarray, table_lens: np.ndarray,
                                    req_ids: List[str], st=None) -> Dict[str, Any]:
        """Decode step of the percentage prune type (kvcache_prunner_old.py:264-295,
        page_evict_kv_util_bak.py:66-106, trigger torch_cuda_sdpa.py:1310): nothing depends on
        the data, so the decision is host arithmetic; the device compacts the tables and, for
        value_l2, every layer merges its own blocks.  ``st``: the tables are device-resident
        (``DeviceTableState``): the freed columns are closed inside the resident rows, the
        merges read a copy of the columns they need taken before that."""
        cfg = self.paged_evict_config
        bs = self.cache_config.block_size
        ib, nbm = cfg.initial_blocks, cfg.num_block_merge
        # on a trigger step the last block has just been filled, so a sequence holds
        # table_len * block_size tokens: "more than initial_blocks + num_block_merge blocks"
        trig_np = (L % (bs * cfg.evict_freq) == 0) & (table_lens > ib + nbm)
        dec: Dict[str, Any] = {"rmv": None, "tables": None, "zero_slots": None, "host_rmv": None,
                               "req_ids": None, "merge": None}
        # under HIP-graph capture the launches must serve later steps too: the trigger of each row
        # is then derived on the device from the resident counters (same arithmetic), and a step
        # on which some row stands on a multiple of evict_freq blocks issues the launches whether
        # or not that row holds enough blocks yet (the other steps launch nothing, captured or not:
        # an engine keeps one graph for each of the two kinds of step)
        capturing = st is not None and torch.cuda.is_current_stream_capturing() and \
            bool((L % (bs * cfg.evict_freq) == 0).any())
        if not trig_np.any() and not capturing:
            return dec
        dev = dm.block_tables.device
        value_l2 = cfg.decode_evict_method != "streamingLLM"     # a scored method: blocks are merged
        first_freed = ib + 1 if value_l2 else ib
        n_remove = nbm - 1
        M = dm.block_tables.shape[1]
        B = len(req_ids)
        dec["host_rmv"] = percentage_decode_victims(cfg, L, table_lens, req_ids, bs)
        if capturing:
            trig_d = (st.seq_lens[:B] % (bs * cfg.evict_freq) == 0) & (st.num_blocks[:B] > ib + nbm)
            start = torch.where(trig_d, first_freed, -1).to(torch.int32)
            rows_d = torch.where(trig_d, torch.arange(B, device=dev), -1).to(torch.int32)
        else:
            start = staging.upload_i32(np.where(trig_np, first_freed, -1), dev)
            rows_d = None
        if value_l2:
            dec["merge"] = rows_d if capturing else staging.upload_i32(np.nonzero(trig_np)[0], dev)
            dec["merge_tables"] = dm.block_tables
        if st is not None:
            if self.compat.per_layer_decision:
                raise NotImplementedError("device-resident block tables take one decision per step")
            if value_l2:
                # the merged group's block ids, before their columns close
                dec["merge_tables"] = st.tables[:B, :ib + nbm].clone()
            # a host-chosen victim column per triggered row, once per freed block (the next one
            # has moved into the same column).  The launch applies it on block boundaries past its
            # budget argument; a triggered row holds more than initial_blocks + num_block_merge
            # >= 3 blocks, so the smallest budget the entry point takes lets every one through.
            for _ in range(n_remove):
                ops.evict_select_compact_inplace(st.tables, st.num_blocks, st.kv_lens, st.seq_lens, B,
                                                 "global", 3 * bs, bs, victim_override=start)
            return dec
        width = min(max(int((table_lens - trig_np * n_remove).max()), 1), M)
        dec["tables"] = ops.block_table_remove_range(dm.block_tables, start, n_remove, width)
        return dec


def percentage_decode_victims(cfg, logical_lens, table_lens, req_ids, block_size: int) -> Dict[int, Tuple[str, int]]:
    """What ``get_rmv_block_idx()`` reports on a decode step of the percentage prune type
    ({batch index: (request id, first freed column)}): host arithmetic on the step's logical
    lengths and table lengths (trigger torch_cuda_sdpa.py:1310; columns
    page_evict_kv_util_bak.py:66-106).  An engine that replays the step's launches from a HIP graph
    calls it for the commit."""
    ib, nbm = cfg.initial_blocks, cfg.num_block_merge
    L = np.asarray(logical_lens)
    trig = (L % (block_size * cfg.evict_freq) == 0) & (np.asarray(table_lens) > ib + nbm)
    first_freed = ib + 1 if cfg.decode_evict_method != "streamingLLM" else ib
    return {int(i): (req_ids[i], first_freed) for i in np.nonzero(trig)[0]}


def _layer_index(prefix: str) -> int:
    """First integer component of a dotted layer name ('model.layers.3.self_attn.attn' -> 3)."""
    for part in prefix.split("."):
        if part.isdigit():
            return int(part)
    return -1


def _cumsum_t(lens, device) -> torch.Tensor:
    return staging.upload_i32(list(accumulate(lens, initial=0)), device)


def get_attn_backend_for_eviction(cache_config) -> Optional[Type[PagedEvictBackend]]:
    """Selector hook: the backend class to use when ``--enable-paged-eviction`` is set
    (replaces the TORCH_SDPA branch of vllm/attention/selector.py:138-146, which ROCm's
    platform override vllm/platforms/rocm.py:47-57 never reaches)."""
    if cache_config is not None and getattr(cache_config, "paged_evict_config", None) is not None:
        return PagedEvictBackend
    return None
