"""Torch-level operators over the C ABI of libpaged_evict.so.

The function names, argument order and argument meaning of ``reshape_and_cache``,
``paged_attention_v1/v2``, ``copy_blocks`` and ``swap_blocks`` mirror the reference's
``vllm/_custom_ops.py`` wrappers of ``torch.ops._C`` / ``torch.ops._C_cache_ops``
(schemas: csrc/torch_bindings.cpp:27-51,430-468), so a caller of those can switch by
changing the import.  The same callables are registered as ``torch.ops.paged_evict.*``.

Validation lives here (dtypes, devices, contiguity); the kernels get raw device
pointers and the current HIP stream.  There is no CPU implementation: tensors that are
not on a GPU raise.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.PE_F32, torch.float16: _lib.PE_F16, torch.bfloat16: _lib.PE_BF16}

EVICT_METHODS = {"streamingLLM": 0, "streamingLLM-1": 1, "local": 2, "global": 3,
                 "inverse_key_l2": 4}
SCORE_TOKEN_MAJOR, SCORE_CACHE, SCORE_CACHE_FLAT, SCORE_KEY_CACHE = 0, 1, 2, 3
# KVCachePruner.sub_evict_method (kvcache_prunner.py:13,29-66) -> pe_score_formula
SCORE_FORMULAS = {"value_l2": 0, "key_l2": 1, "value_l2_plus_key_l2": 2, "inverse_key_l2": 3}
# token scores of the README-era class, by its method names (get_score,
# vllm/attention/kvcache_prunner_old.py:77-134) -> pe_score_formula; "streamingLLM" as a score is
# the token's position
TOKEN_SCORE_FORMULAS = dict(SCORE_FORMULAS, key_l1=4, value_l1=5, inverse_key_l1=6,
                            key_l1_div_value_l1=7, key_l1_div_value_l2=8, key_l2_div_value_l1=9,
                            key_l2_div_value_l2=10, value_l1_div_key_l1=11, value_l1_div_key_l2=12,
                            value_l2_div_key_l1=13, value_l2_div_key_l2=14, value_l1_plus_key_l1=15,
                            cosine=16, streamingLLM=17)
_SCORE_FORMULA_SHIFT = 8


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"paged_evict: unsupported dtype {t.dtype}") from None


def _need_gpu(*tensors: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("paged_evict ops run on the GPU only (got a %s tensor); "
                               "there is no CPU fallback" % t.device.type)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("paged_evict: tensors on different devices")
    assert dev is not None
    return dev


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_workspaces: Dict[Tuple[str, int], torch.Tensor] = {}


def _workspace(dev: torch.device, nbytes: int, tag: str = "decode") -> Optional[torch.Tensor]:
    """Per-device scratch buffer, grown on demand (allocated by torch, outside the C call)."""
    if nbytes <= 0:
        return None
    key = (tag, dev.index if dev.index is not None else torch.cuda.current_device())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


def reload_env() -> None:
    """Have the library read its PE_* tunables from the environment again (it reads each once)."""
    _lib.load().pe_reload_env()


def _check_kv_cache_args(kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
    if kv_cache_dtype != "auto":
        # the reference backend rejects fp8 KV (torch_cuda_sdpa.py:769-772)
        raise NotImplementedError("paged_evict does not support FP8 KV cache "
                                  f"(kv_cache_dtype={kv_cache_dtype!r})")
    if k_scale != 1.0 or v_scale != 1.0:
        raise RuntimeError("paged_evict: k_scale and v_scale must be 1.0")


DIAG_DECODE_LEN_CLAMPED, DIAG_SCORE_ROW_FULL, DIAG_TABLE_STATE_MISMATCH = 1, 2, 4


def device_diagnostics(reset: bool = True) -> int:
    """Flags the kernels set on the current device when they had to clamp inconsistent arguments
    (PE_DIAG_* of include/paged_evict.h).  Synchronises the device: for tests and debugging."""
    import ctypes
    flags = ctypes.c_uint(0)
    lib = _lib.load()
    _lib.check(lib.pe_device_diagnostics(ctypes.addressof(flags), 1 if reset else 0),
               "device_diagnostics")
    return int(flags.value)


def check_device_diagnostics() -> None:
    """Raise if any kernel since the last check clamped a sequence length to its block table or
    dropped a block score because the score row was full."""
    flags = device_diagnostics(reset=True)
    if flags:
        what = []
        if flags & DIAG_DECODE_LEN_CLAMPED:
            what.append("a decode launch got a seq_lens entry beyond its block-table row")
        if flags & DIAG_SCORE_ROW_FULL:
            what.append("a block score was dropped because the sequence's score row was full")
        if flags & DIAG_TABLE_STATE_MISMATCH:
            what.append("device-resident block tables disagree with the host's counters")
        raise RuntimeError("paged_evict device diagnostics: " + "; ".join(what))


# --------------------------------------------------------------------------------------
def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                      kv_cache_dtype: str = "auto", k_scale: float = 1.0, v_scale: float = 1.0,
                      *, src_index: Optional[torch.Tensor] = None, src_pairs: bool = False,
                      zero_slots: Optional[torch.Tensor] = None) -> None:
    """key/value [T, H_kv, D]; key_cache [nb, H_kv, D/x, bs, x]; value_cache [nb, H_kv, D, bs];
    slot_mapping int64[T].  ``src_index`` (int32 [T] or [T, H_kv]) fuses a row gather into the
    write (``src_pairs``: [T, H_kv] entries are row*H_kv+head pairs); ``zero_slots``
    (int64 [n]) zeroes those slots after the write."""
    _check_kv_cache_args(kv_cache_dtype, k_scale, v_scale)
    dev = _need_gpu(key, value, key_cache, value_cache, slot_mapping, src_index, zero_slots)
    if key_cache.dim() != 5 or value_cache.dim() != 4:
        raise RuntimeError("reshape_and_cache: key_cache must be 5-d and value_cache 4-d")
    if not (key_cache.is_contiguous() and value_cache.is_contiguous()):
        raise RuntimeError("reshape_and_cache: caches must be contiguous")
    nb, H, D, bs = value_cache.shape
    if key.dim() != 3 or key.shape[1:] != (H, D) or value.shape[1:] != (H, D):
        raise RuntimeError(f"reshape_and_cache: key/value must be [T, {H}, {D}]")
    if key.dtype != key_cache.dtype or value.dtype != value_cache.dtype or key.dtype != value.dtype:
        raise RuntimeError("reshape_and_cache: dtype mismatch between key/value and the caches")
    if key.stride(2) != 1 or key.stride(1) != D or value.stride(2) != 1 or value.stride(1) != D:
        raise RuntimeError("reshape_and_cache: key/value rows must be dense")
    slot_mapping = slot_mapping.flatten()
    if slot_mapping.dtype != torch.int64 or not slot_mapping.is_contiguous():
        raise RuntimeError("reshape_and_cache: slot_mapping must be contiguous int64")
    T = slot_mapping.numel()
    head_stride = 0
    if src_index is not None:
        if src_index.dtype != torch.int32 or not src_index.is_contiguous():
            raise RuntimeError("reshape_and_cache: src_index must be contiguous int32")
        if src_index.dim() == 2:
            if src_index.shape != (T, H):
                raise RuntimeError("reshape_and_cache: src_index must be [T] or [T, H_kv]")
            head_stride = 2 if src_pairs else 1
        elif src_index.numel() != T:
            raise RuntimeError("reshape_and_cache: src_index must be [T] or [T, H_kv]")
    elif key.shape[0] < T or value.shape[0] < T:
        raise RuntimeError("reshape_and_cache: fewer key/value rows than slots")
    nz = 0
    if zero_slots is not None:
        if zero_slots.dtype != torch.int64 or not zero_slots.is_contiguous():
            raise RuntimeError("reshape_and_cache: zero_slots must be contiguous int64")
        nz = zero_slots.numel()
    lib = _lib.load()
    _lib.check(lib.pe_reshape_and_cache(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), T,
        key.stride(0), value.stride(0), _p(src_index), head_stride, _p(zero_slots), nz,
        H, D, bs, nb, _dtype_code(key), _stream(dev)), "reshape_and_cache")


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """In-place rotary embedding of query [T, Hq*D] (or [T, Hq, D]) and key; names and arguments
    of ``vllm._custom_ops.rotary_embedding``."""
    rotary_embedding_and_cache(positions, query, key, None, None, None, None, head_size,
                               cos_sin_cache, is_neox)


def rotary_embedding_and_cache(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                               value: Optional[torch.Tensor], key_cache: Optional[torch.Tensor],
                               value_cache: Optional[torch.Tensor],
                               slot_mapping: Optional[torch.Tensor], head_size: int,
                               cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """Rotary embedding of query and key in place and, in the same launch, the KV write of the
    rotated key and of value into the paged cache (``slot_mapping`` int64 [T], one slot per token,
    negative = not cached).  With ``key_cache=None`` only the rotation runs."""
    dev = _need_gpu(positions, query, key, value, key_cache, value_cache, slot_mapping, cos_sin_cache)
    T = positions.numel()
    q2 = query.reshape(T, -1) if query.dim() != 2 else query
    k2 = key.reshape(T, -1) if key.dim() != 2 else key
    if q2.data_ptr() != query.data_ptr() or k2.data_ptr() != key.data_ptr():
        raise RuntimeError("rotary_embedding: query/key must be viewable as [num_tokens, heads*head_size]")
    if q2.stride(1) != 1 or k2.stride(1) != 1 or q2.shape[1] % head_size or k2.shape[1] % head_size:
        raise RuntimeError("rotary_embedding: rows must be dense multiples of head_size")
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rotary_embedding: positions must be contiguous int64")
    if cos_sin_cache.dim() != 2 or not cos_sin_cache.is_contiguous() or cos_sin_cache.dtype != query.dtype \
            or key.dtype != query.dtype:
        raise RuntimeError("rotary_embedding: cos_sin_cache [max_position, rot_dim] of the query dtype expected")
    Hq, Hkv = q2.shape[1] // head_size, k2.shape[1] // head_size
    bs = nb = 0
    v_stride = 0
    if key_cache is not None:
        if value is None or value_cache is None or slot_mapping is None:
            raise RuntimeError("rotary_embedding_and_cache: value, value_cache and slot_mapping needed")
        v2 = value.reshape(T, -1) if value.dim() != 2 else value
        if v2.data_ptr() != value.data_ptr() or v2.stride(1) != 1 or v2.shape[1] != Hkv * head_size \
                or value.dtype != query.dtype:
            raise RuntimeError("rotary_embedding_and_cache: value must be [num_tokens, kv_heads*head_size]")
        if key_cache.dim() != 5 or value_cache.dim() != 4 or not key_cache.is_contiguous() \
                or not value_cache.is_contiguous() or key_cache.dtype != query.dtype:
            raise RuntimeError("rotary_embedding_and_cache: contiguous 5-d key_cache / 4-d value_cache expected")
        nb, H, D, bs = value_cache.shape
        if (H, D) != (Hkv, head_size):
            raise RuntimeError("rotary_embedding_and_cache: cache geometry does not match key")
        if slot_mapping.dtype != torch.int64 or slot_mapping.numel() != T or not slot_mapping.is_contiguous():
            raise RuntimeError("rotary_embedding_and_cache: slot_mapping must be contiguous int64 [num_tokens]")
        v_stride = v2.stride(0)
        value = v2
    lib = _lib.load()
    _lib.check(lib.pe_rope_reshape_and_cache(
        _p(q2), _p(k2), _p(value), _p(key_cache), _p(value_cache), _p(positions), _p(cos_sin_cache),
        _p(slot_mapping), T, q2.stride(0), k2.stride(0), v_stride, Hq, Hkv, head_size,
        cos_sin_cache.shape[1], 1 if is_neox else 0, cos_sin_cache.shape[0], bs, nb,
        _dtype_code(query), _stream(dev)), "rope_reshape_and_cache")


def _paged_attention(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                     seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                     v_scale, blocksparse_vert_stride) -> None:
    _check_kv_cache_args(kv_cache_dtype, k_scale, v_scale)
    if blocksparse_vert_stride is not None and blocksparse_vert_stride > 1:
        raise ValueError("paged_evict does not support block-sparse attention")
    dev = _need_gpu(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes)
    if query.dim() != 3 or out.shape != query.shape:
        raise RuntimeError("paged_attention: query/out must be [num_seqs, num_heads, head_size]")
    S, Hq, D = query.shape
    if key_cache.dim() != 5 or value_cache.dim() != 4:
        raise RuntimeError("paged_attention: key_cache must be 5-d and value_cache 4-d")
    nb, Hkv, Dv, bs = value_cache.shape
    if Hkv != num_kv_heads or Dv != D or bs != block_size:
        raise RuntimeError("paged_attention: cache shape does not match num_kv_heads/head_size/"
                           "block_size")
    if not (query.dtype == key_cache.dtype == value_cache.dtype == out.dtype):
        raise RuntimeError("paged_attention: dtype mismatch")
    if query.stride(2) != 1 or query.stride(1) != D or not out.is_contiguous():
        raise RuntimeError("paged_attention: query rows must be dense and out contiguous")
    if key_cache.stride(0) != value_cache.stride(0) or key_cache.stride(1) != value_cache.stride(1):
        raise RuntimeError("paged_attention: key and value cache strides differ")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention: block_tables and seq_lens must be int32")
    if block_tables.dim() != 2 or block_tables.shape[0] != S or block_tables.stride(1) != 1 \
            or block_tables.stride(0) != block_tables.shape[1]:
        raise RuntimeError("paged_attention: block_tables must be contiguous [num_seqs, max_blocks]")
    if seq_lens.numel() != S or not seq_lens.is_contiguous():
        raise RuntimeError("paged_attention: seq_lens must be contiguous [num_seqs]")
    if alibi_slopes is not None and (alibi_slopes.dtype != torch.float32 or alibi_slopes.numel() != Hq):
        raise RuntimeError("paged_attention: alibi_slopes must be float32[num_heads]")
    lib = _lib.load()
    need = lib.pe_paged_decode_workspace_bytes(S, Hq, num_kv_heads, D, block_size, max_seq_len)
    ws = _workspace(dev, need)
    _lib.check(lib.pe_paged_decode(
        _p(out), _p(query), _p(key_cache), _p(value_cache), _p(block_tables), _p(seq_lens),
        _p(alibi_slopes), S, Hq, num_kv_heads, D, block_size, block_tables.shape[1],
        query.stride(0), key_cache.stride(0), key_cache.stride(1), float(scale), int(max_seq_len),
        _dtype_code(query), _p(ws), 0 if ws is None else ws.numel(), _stream(dev)),
        "paged_attention")


def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype="auto",
                       k_scale=1.0, v_scale=1.0, tp_rank=0, blocksparse_local_blocks=0,
                       blocksparse_vert_stride=0, blocksparse_block_size=64,
                       blocksparse_head_sliding_step=0) -> None:
    _paged_attention(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                     seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                     v_scale, blocksparse_vert_stride)


def paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                       num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes, kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0, tp_rank=0,
                       blocksparse_local_blocks=0, blocksparse_vert_stride=0,
                       blocksparse_block_size=64, blocksparse_head_sliding_step=0) -> None:
    """Same kernel as v1: the split-KV partition count is chosen inside the library and its
    partials live in the library's workspace; exp_sums / max_logits / tmp_out are accepted for
    signature compatibility and left untouched."""
    _paged_attention(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                     seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                     v_scale, blocksparse_vert_stride)


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    """block_mapping int64 [num_pairs, 2] (src, dst); copies in every layer's K and V cache."""
    if len(key_caches) != len(value_caches):
        raise RuntimeError("copy_blocks: key_caches and value_caches differ in length")
    if not key_caches or block_mapping.numel() == 0:
        return
    dev = _need_gpu(*key_caches, *value_caches)
    bm = block_mapping.to(device=dev, dtype=torch.int64).contiguous()
    kp = torch.tensor([t.data_ptr() for t in key_caches], dtype=torch.int64).to(dev)
    vp = torch.tensor([t.data_ptr() for t in value_caches], dtype=torch.int64).to(dev)
    lib = _lib.load()
    _lib.check(lib.pe_copy_blocks(_p(kp), _p(vp), _p(bm), len(key_caches), bm.shape[0],
                                  key_caches[0][0].numel(), _dtype_code(key_caches[0]),
                                  _stream(dev)), "copy_blocks")


def zero_slots_layers(key_ptrs: torch.Tensor, value_ptrs: torch.Tensor, zero_slots: torch.Tensor,
                      num_kv_heads: int, head_size: int, block_size: int, num_blocks: int,
                      dtype: torch.dtype) -> None:
    """Zero the K and V rows of ``zero_slots`` (int64) in the caches of all layers with one launch.
    ``key_ptrs`` / ``value_ptrs``: device int64 tensors holding each layer's cache address."""
    dev = _need_gpu(key_ptrs, value_ptrs, zero_slots)
    if key_ptrs.dtype != torch.int64 or value_ptrs.dtype != torch.int64 or zero_slots.dtype != torch.int64 \
            or key_ptrs.numel() != value_ptrs.numel():
        raise RuntimeError("zero_slots_layers: int64 pointer arrays of equal length and int64 slots expected")
    lib = _lib.load()
    _lib.check(lib.pe_zero_slots_layers(_p(key_ptrs), _p(value_ptrs), key_ptrs.numel(), _p(zero_slots),
                                        zero_slots.numel(), num_kv_heads, head_size, block_size,
                                        num_blocks, _DT[dtype], _stream(dev)), "zero_slots_layers")


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    """block_mapping: CPU int64 [num_pairs, 2]; src/dst may live on the GPU or the host."""
    if block_mapping.device.type != "cpu":
        raise RuntimeError("block_mapping must be on CPU")
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")
        kind, dev = 0, src.device
    elif src.is_cuda and dst.device.type == "cpu":
        kind, dev = 1, src.device
    elif src.device.type == "cpu" and dst.is_cuda:
        kind, dev = 2, dst.device
    else:
        raise RuntimeError("Invalid device combination")
    bm = block_mapping.to(torch.int64).contiguous()
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.pe_swap_blocks(_p(src), _p(dst), _p(bm), bm.shape[0],
                                      src[0].numel() * src.element_size(), kind, _stream(dev)),
                   "swap_blocks")


# --------------------------------------------------------------------------------------
# measurement hook: when set to a list, prefill_causal brackets its launch with events on the
# launch stream and appends (start, end)
prefill_event_sink = None


def prefill_causal(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                   cu_q: torch.Tensor, cu_kv: torch.Tensor, max_q_len: int, scale: float,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Varlen causal attention; query [Tq, Hq, D], key/value [Tkv, Hkv, D]; cu_q/cu_kv int32
    prefix sums on the GPU.  Query i of a request attends kv rows 0..min(i, Lkv-1)."""
    dev = _need_gpu(query, key, value, cu_q, cu_kv)
    if query.dim() != 3 or key.dim() != 3 or value.shape != key.shape:
        raise RuntimeError("prefill_causal: query [Tq,Hq,D], key/value [Tkv,Hkv,D] expected")
    Tq, Hq, D = query.shape
    Hkv = key.shape[1]
    if key.shape[2] != D or Hq % Hkv:
        raise RuntimeError("prefill_causal: head layout mismatch")
    for t in (query, key, value):
        if t.stride(2) != 1 or t.stride(1) != D:
            raise RuntimeError("prefill_causal: rows must be dense")
    if not (query.dtype == key.dtype == value.dtype):
        raise RuntimeError("prefill_causal: dtype mismatch")
    if cu_q.dtype != torch.int32 or cu_kv.dtype != torch.int32 or cu_q.numel() != cu_kv.numel():
        raise RuntimeError("prefill_causal: cu_q/cu_kv must be int32 of equal length")
    if out is None:
        out = torch.empty((Tq, Hq, D), dtype=query.dtype, device=dev)
    lib = _lib.load()
    sink = prefill_event_sink
    if sink is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
    _lib.check(lib.pe_prefill_causal(
        _p(out), _p(query), _p(key), _p(value), _p(cu_q), _p(cu_kv), cu_q.numel() - 1,
        int(max_q_len), Hq, Hkv, D, query.stride(0), key.stride(0), value.stride(0),
        out.stride(0), float(scale), _dtype_code(query), _stream(dev)), "prefill_causal")
    if sink is not None:
        e_ev.record()
        sink.append((s_ev, e_ev))
    return out


def prompt_keep_index_streaming(cu_q: torch.Tensor, cu_kept: torch.Tensor, total_kept: int,
                                cache_budget: int, block_size: int) -> torch.Tensor:
    dev = _need_gpu(cu_q, cu_kept)
    idx = torch.empty(total_kept, dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.pe_prompt_keep_index_streaming(_p(idx), _p(cu_q), _p(cu_kept),
                                                  cu_q.numel() - 1, cache_budget, block_size,
                                                  _stream(dev)), "prompt_keep_index_streaming")
    return idx


def prompt_token_select(key: torch.Tensor, value: torch.Tensor, cu_q: torch.Tensor,
                        cu_kept: torch.Tensor, total_kept: int, cache_budget: int,
                        block_size: int, method: str, compat_q2: bool = False,
                        tail_blocks: int = 1, formula: str = "value_l2") -> torch.Tensor:
    """Per-(token, head) kept-row index [total_kept, H_kv] int32 of the token-level prompt prune.
    cache_budget 0: percentage prune type (the last ``tail_blocks`` blocks stay, the pruned count
    of a prompt is its length minus its kept length).  ``formula``: the score of "local" /
    "global" (SCORE_FORMULAS; "inverse_key_l2" as a method always uses its own)."""
    dev = _need_gpu(key, value, cu_q, cu_kept)
    if formula not in TOKEN_SCORE_FORMULAS:
        raise ValueError(f"Unknown score formula: {formula}")
    T, H, D = key.shape
    idx = torch.empty((total_kept, H), dtype=torch.int32, device=dev)
    lib = _lib.load()
    need = lib.pe_prompt_token_select_workspace_bytes(T, H)
    ws = _workspace(dev, need, "select")
    _lib.check(lib.pe_prompt_token_select(
        _p(idx), _p(key), _p(value), _p(cu_q), _p(cu_kept), cu_q.numel() - 1, T, cache_budget,
        block_size, EVICT_METHODS[method] | (TOKEN_SCORE_FORMULAS[formula] << _SCORE_FORMULA_SHIFT),
        1 if compat_q2 else 0, tail_blocks, key.stride(0),
        value.stride(0),
        H, D, _dtype_code(key), _p(ws), 0 if ws is None else ws.numel(), _stream(dev)),
        "prompt_token_select")
    return idx


def prompt_prune_gather(key: torch.Tensor, value: torch.Tensor, src_index: torch.Tensor,
                        mode: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense gather of kept rows: src_index int32 [n] (rows) or [n, H_kv] (rows per head;
    ``mode=2``: entries are row*H_kv+head pairs)."""
    dev = _need_gpu(key, value, src_index)
    T, H, D = key.shape
    n = src_index.shape[0]
    if mode is None:
        mode = 0 if src_index.dim() == 1 else 1
    ok = torch.empty((n, H, D), dtype=key.dtype, device=dev)
    ov = torch.empty((n, H, D), dtype=key.dtype, device=dev)
    lib = _lib.load()
    _lib.check(lib.pe_prompt_prune_gather(_p(key), _p(value), _p(ok), _p(ov), _p(src_index), mode,
                                          n, key.stride(0), value.stride(0), H, D,
                                          _dtype_code(key), _stream(dev)), "prompt_prune_gather")
    return ok, ov


def block_score_value_l2(values: torch.Tensor, block_ids: Optional[torch.Tensor], n_blocks: int,
                         layout: int, num_kv_heads: int, head_size: int, block_size: int,
                         block_stride: Optional[int] = None,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dev = _need_gpu(values, block_ids)
    if out is None:
        out = torch.empty(n_blocks, dtype=torch.float32, device=dev)
    if block_stride is None:
        block_stride = block_size * num_kv_heads * head_size
    lib = _lib.load()
    _lib.check(lib.pe_block_score_value_l2(_p(out), _p(values), _p(block_ids), n_blocks, layout,
                                           block_stride, num_kv_heads, head_size, block_size,
                                           _dtype_code(values), _stream(dev)),
               "block_score_value_l2")
    return out


def combine_block_scores(formula: str, value_scores: Optional[torch.Tensor],
                         key_scores: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """Block scores of a score formula from the L2-norm block scores of the value and key tensors
    (f32 tensors holding storage-type values, as block_score_value_l2 returns them), with the
    roundings torch applies to 0-d tensors of the storage type (kvcache_prunner.py:51-66)."""
    if formula == "value_l2":
        return value_scores
    if formula == "key_l2":
        return key_scores
    if formula == "value_l2_plus_key_l2":
        return (value_scores.to(dtype) + key_scores.to(dtype)).float()
    if formula == "inverse_key_l2":
        return (1 / (key_scores.to(dtype) + 1e-8)).float()
    raise ValueError(f"Unknown score formula: {formula}")


def block_token_norm_sums(values: torch.Tensor, block_ids: Optional[torch.Tensor], n_blocks: int,
                          layout: int, num_kv_heads: int, head_size: int, block_size: int,
                          block_stride: Optional[int] = None,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Step 1 of the head-sharded score: f32 [n_blocks * block_size], per token the sum over the
    local heads of the rounded norms (to be summed over the tensor-parallel ranks)."""
    dev = _need_gpu(values, block_ids)
    if out is None:
        out = torch.empty(n_blocks * block_size, dtype=torch.float32, device=dev)
    if block_stride is None:
        block_stride = block_size * num_kv_heads * head_size
    lib = _lib.load()
    _lib.check(lib.pe_block_token_norm_sums(_p(out), _p(values), _p(block_ids), n_blocks, layout,
                                            block_stride, num_kv_heads, head_size, block_size,
                                            _dtype_code(values), _stream(dev)),
               "block_token_norm_sums")
    return out


def block_score_from_token_sums(sums: torch.Tensor, n_blocks: int, block_size: int,
                                total_kv_heads: int, dtype: torch.dtype,
                                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Step 2: scores f32 [n_blocks] from the rank-summed token sums; ``dtype`` is the cache's
    storage type (its rounding is applied to the head mean and to the token sum)."""
    dev = _need_gpu(sums)
    if sums.dtype != torch.float32 or not sums.is_contiguous() or sums.numel() < n_blocks * block_size:
        raise RuntimeError("block_score_from_token_sums: contiguous f32 [n_blocks*block_size] expected")
    if out is None:
        out = torch.empty(n_blocks, dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.pe_block_score_from_token_sums(_p(out), _p(sums), n_blocks, block_size,
                                                  total_kv_heads, _DT[dtype], _stream(dev)),
               "block_score_from_token_sums")
    return out


def evict_select_compact(block_tables: torch.Tensor, seq_lens: torch.Tensor,
                         num_blocks: torch.Tensor, method: str, cache_budget: int,
                         block_size: int, topk_blocks: int = -1, *,
                         victim_override: Optional[torch.Tensor] = None,
                         drop_victim_score: bool = True, compat_q1: bool = False,
                         score_layout: int = SCORE_CACHE,
                         score_table: Optional[torch.Tensor] = None,
                         score_counts: Optional[torch.Tensor] = None,
                         score_rows: Optional[torch.Tensor] = None,
                         value_cache: Optional[torch.Tensor] = None,
                         compact: bool = True, out_width: Optional[int] = None,
                         new_scores: Optional[torch.Tensor] = None,
                         dtype: Optional[torch.dtype] = None):
    """One launch per decode step: returns (rmv_idx int32[B], compacted tables int32[B, out_width]
    or None, zero_slots int64[B] or None).  out_width defaults to the reference's M-1; pass the
    longest surviving row when a row that evicts nothing may be M blocks long.  ``new_scores``
    (f32 [B]): scores of the just-filled blocks computed by the caller (head-sharded caches)
    instead of in the launch."""
    dev = _need_gpu(block_tables, seq_lens, num_blocks, victim_override, score_table,
                    score_counts, score_rows, value_cache, new_scores)
    B, M = block_tables.shape
    for t in (block_tables, seq_lens, num_blocks):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("evict_select_compact: int32 contiguous tensors expected")
    m = EVICT_METHODS[method]
    rmv = torch.empty(B, dtype=torch.int32, device=dev)
    if out_width is None:
        out_width = max(M - 1, 1)
    out_tables = torch.empty((B, out_width), dtype=torch.int32, device=dev) if compact else None
    zero = torch.empty(B, dtype=torch.int64, device=dev) if m == 1 else None
    Hkv = D = 0
    kv_stride = 0
    dt = _lib.PE_BF16
    width = 0
    if m in (2, 3):
        if victim_override is None or score_table is not None:
            if score_table is None or score_counts is None or score_rows is None or \
                    (value_cache is None and new_scores is None):
                raise RuntimeError("evict_select_compact: local/global need score_table, "
                                   "score_counts, score_rows and value_cache (or new_scores)")
        if new_scores is not None and (new_scores.dtype != torch.float32 or new_scores.numel() != B
                                       or not new_scores.is_contiguous()):
            raise RuntimeError("evict_select_compact: new_scores must be contiguous f32 [batch]")
        if value_cache is not None:
            _, Hkv, D, bs = value_cache.shape
            if bs != block_size:
                raise RuntimeError("evict_select_compact: value_cache block size mismatch")
            kv_stride = value_cache.stride(0)
            dt = _dtype_code(value_cache)
        if score_table is not None:
            width = score_table.shape[1]
    lib = _lib.load()
    _lib.check(lib.pe_evict_select_compact(
        _p(rmv), _p(out_tables), _p(zero), _p(block_tables), _p(seq_lens), _p(num_blocks),
        _p(victim_override), B, M, out_width, m, cache_budget, block_size, topk_blocks,
        1 if drop_victim_score else 0, 1 if compat_q1 else 0, score_layout, _p(score_table),
        _p(score_counts), _p(score_rows), width, _p(value_cache), kv_stride, Hkv, D,
        _p(new_scores), dt if dtype is None else _DT[dtype], _stream(dev)),
        "evict_select_compact")
    return rmv, out_tables, zero


def table_state_advance(tables: torch.Tensor, num_blocks: torch.Tensor, kv_lens: torch.Tensor,
                        seq_lens: torch.Tensor, new_blocks: torch.Tensor, batch: int,
                        slot_mapping_out: torch.Tensor, attn_kv_lens_out: torch.Tensor,
                        last_blocks_out: Optional[torch.Tensor], evict_rule: int, cache_budget: int,
                        block_size: int, evict_freq: int = 0, num_block_merge: int = 2,
                        initial_blocks: int = 1) -> None:
    """One decode step's append on device-resident tables (see pe_table_state_advance): rows
    0..batch-1 of the state arrays are the step's sequences in batch order.  A two-dimensional
    ``new_blocks`` [ring_len, >= batch] is a ring of per-step rows (pe_table_state_advance_ring):
    sequence b takes its entry from row (seq_lens[b] + 1) % ring_len; it may be a pinned host
    tensor, which the launch reads in place."""
    ring = new_blocks.dim() == 2
    dev = _need_gpu(tables, num_blocks, kv_lens, seq_lens, None if ring and not new_blocks.is_cuda
                    else new_blocks, slot_mapping_out, attn_kv_lens_out, last_blocks_out)
    for t in (tables, num_blocks, kv_lens, seq_lens, new_blocks, attn_kv_lens_out):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("table_state_advance: int32 contiguous tensors expected")
    if slot_mapping_out.dtype != torch.int64 or not slot_mapping_out.is_contiguous():
        raise RuntimeError("table_state_advance: slot_mapping_out must be contiguous int64")
    if tables.dim() != 2 or min(tables.shape[0], num_blocks.numel(), kv_lens.numel(), seq_lens.numel(),
                                new_blocks.shape[-1], slot_mapping_out.numel(),
                                attn_kv_lens_out.numel()) < batch:
        raise RuntimeError("table_state_advance: state arrays shorter than the batch")
    lib = _lib.load()
    if ring:
        if not new_blocks.is_cuda and not new_blocks.is_pinned():
            raise RuntimeError("table_state_advance: a host ring must be pinned memory")
        if new_blocks.shape[0] < 2:
            raise RuntimeError("table_state_advance: a ring has at least 2 rows")
        _lib.check(lib.pe_table_state_advance_ring(
            _p(tables), _p(num_blocks), _p(kv_lens), _p(seq_lens), _p(new_blocks), new_blocks.shape[0],
            new_blocks.shape[1], batch, tables.shape[1], _p(slot_mapping_out), _p(attn_kv_lens_out),
            _p(last_blocks_out), evict_rule, cache_budget, block_size, evict_freq, num_block_merge,
            initial_blocks, _stream(dev)), "table_state_advance_ring")
        return
    _lib.check(lib.pe_table_state_advance(
        _p(tables), _p(num_blocks), _p(kv_lens), _p(seq_lens), _p(new_blocks), batch, tables.shape[1],
        _p(slot_mapping_out), _p(attn_kv_lens_out), _p(last_blocks_out), evict_rule, cache_budget,
        block_size, evict_freq, num_block_merge, initial_blocks, _stream(dev)), "table_state_advance")


def evict_select_compact_inplace(block_tables: torch.Tensor, num_blocks: torch.Tensor,
                                 kv_lens: torch.Tensor, seq_lens: torch.Tensor, batch: int,
                                 method: str, cache_budget: int, block_size: int,
                                 topk_blocks: int = -1, *,
                                 victim_override: Optional[torch.Tensor] = None,
                                 drop_victim_score: bool = True, score_layout: int = SCORE_CACHE,
                                 score_table: Optional[torch.Tensor] = None,
                                 score_counts: Optional[torch.Tensor] = None,
                                 score_rows: Optional[torch.Tensor] = None,
                                 value_cache: Optional[torch.Tensor] = None,
                                 new_scores: Optional[torch.Tensor] = None,
                                 dtype: Optional[torch.dtype] = None):
    """pe_evict_select_compact on device-resident tables: rows 0..batch-1 of ``block_tables``
    lose their victim's column in place, ``num_blocks`` / ``kv_lens`` are updated.  Returns
    (rmv_idx int32[batch], freed block ids int32[batch] (-1 none), zero_slots int64[batch] or None)."""
    dev = _need_gpu(block_tables, num_blocks, kv_lens, seq_lens, victim_override, score_table,
                    score_counts, score_rows, value_cache, new_scores)
    for t in (block_tables, num_blocks, kv_lens, seq_lens):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("evict_select_compact_inplace: int32 contiguous tensors expected")
    if block_tables.dim() != 2 or min(block_tables.shape[0], num_blocks.numel(), kv_lens.numel(),
                                      seq_lens.numel()) < batch:
        raise RuntimeError("evict_select_compact_inplace: state arrays shorter than the batch")
    m = EVICT_METHODS[method]
    rmv = torch.empty(batch, dtype=torch.int32, device=dev)
    freed = torch.empty(batch, dtype=torch.int32, device=dev)
    zero = torch.empty(batch, dtype=torch.int64, device=dev) if m == 1 else None
    Hkv = D = 0
    kv_stride = 0
    dt = _lib.PE_BF16
    width = 0
    if m in (2, 3):
        if victim_override is None or score_table is not None:
            if score_table is None or score_counts is None or score_rows is None or \
                    (value_cache is None and new_scores is None):
                raise RuntimeError("evict_select_compact_inplace: local/global need score_table, "
                                   "score_counts, score_rows and value_cache (or new_scores)")
        if value_cache is not None:
            _, Hkv, D, bs = value_cache.shape
            if bs != block_size:
                raise RuntimeError("evict_select_compact_inplace: value_cache block size mismatch")
            kv_stride = value_cache.stride(0)
            dt = _dtype_code(value_cache)
        if score_table is not None:
            width = score_table.shape[1]
    lib = _lib.load()
    _lib.check(lib.pe_evict_select_compact_inplace(
        _p(rmv), _p(freed), _p(zero), _p(block_tables), _p(num_blocks), _p(kv_lens), _p(seq_lens),
        _p(victim_override), batch, block_tables.shape[1], m, cache_budget, block_size, topk_blocks,
        1 if drop_victim_score else 0, score_layout, _p(score_table), _p(score_counts),
        _p(score_rows), width, _p(value_cache), kv_stride, Hkv, D, _p(new_scores),
        dt if dtype is None else _DT[dtype], _stream(dev)), "evict_select_compact_inplace")
    return rmv, freed, zero


def token_scores(key: torch.Tensor, value: torch.Tensor, formula: str) -> torch.Tensor:
    """get_score of the README-era class (vllm/attention/kvcache_prunner_old.py:77-134) on dense
    [T, H, D] key / value tensors: [T, H] scores in the tensors' dtype."""
    dev = _need_gpu(key, value)
    if formula not in TOKEN_SCORE_FORMULAS:
        raise ValueError(f"Unknown score formula: {formula}")
    T, H, D = key.shape
    if value.shape != key.shape or key.stride(2) != 1 or value.stride(2) != 1 \
            or key.stride(1) != D or value.stride(1) != D:
        raise RuntimeError("token_scores: dense [tokens, heads, head_size] rows expected")
    out = torch.empty((T, H), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().pe_token_scores(_p(out), _p(key), _p(value), T, key.stride(0), value.stride(0),
                                           H, D, TOKEN_SCORE_FORMULAS[formula], _dtype_code(key),
                                           _stream(dev)), "token_scores")
    return out.to(key.dtype)


def merge_blocks_value_l2(key_cache: torch.Tensor, value_cache: torch.Tensor,
                          block_tables: torch.Tensor, rows: torch.Tensor, first_block: int,
                          num_block_merge: int, formula: str = "value_l2") -> None:
    """In place, for every batch row listed in ``rows`` (int32) and every KV head: merge the
    ``num_block_merge`` blocks from table column ``first_block`` into the first of them, keeping
    the block_size tokens with the highest score (``formula``: ||V||_2 by default, or any method
    name get_score knows) in their original order (the decode-time merge of the percentage prune
    type, kvcache_prunner_old.py:264-315)."""
    if formula not in TOKEN_SCORE_FORMULAS:
        raise ValueError(f"Unknown score formula: {formula}")
    dev = _need_gpu(key_cache, value_cache, block_tables, rows)
    nb, Hkv, D, bs = value_cache.shape
    if key_cache.dtype != value_cache.dtype or key_cache.shape[0] != nb or key_cache.shape[1] != Hkv:
        raise RuntimeError("merge_blocks_value_l2: key/value cache mismatch")
    if value_cache.stride(3) != 1 or value_cache.stride(2) != bs or \
            key_cache.stride(0) != value_cache.stride(0) or key_cache.stride(1) != value_cache.stride(1):
        raise RuntimeError("merge_blocks_value_l2: paged cache layout expected")
    for t in (block_tables, rows):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("merge_blocks_value_l2: int32 contiguous tensors expected")
    lib = _lib.load()
    _lib.check(lib.pe_merge_blocks_scored(
        _p(key_cache), _p(value_cache), _p(block_tables), _p(rows), rows.numel(),
        block_tables.shape[1], first_block, num_block_merge, Hkv, D, bs, value_cache.stride(0),
        value_cache.stride(1), TOKEN_SCORE_FORMULAS[formula], _dtype_code(value_cache), _stream(dev)),
        "merge_blocks_scored")


def block_table_remove_range(block_tables: torch.Tensor, start: torch.Tensor, n_remove: int,
                             out_width: Optional[int] = None) -> torch.Tensor:
    """Rows of ``block_tables`` without columns [start[b], start[b]+n_remove) (start[b] < 0: row
    kept as is); int32 [B, out_width], out_width defaults to M - n_remove."""
    dev = _need_gpu(block_tables, start)
    B, M = block_tables.shape
    for t in (block_tables, start):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("block_table_remove_range: int32 contiguous tensors expected")
    if start.numel() != B:
        raise RuntimeError("block_table_remove_range: one start column per row expected")
    if out_width is None:
        out_width = max(M - n_remove, 1)
    out = torch.empty((B, out_width), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.pe_block_table_remove_range(_p(out), _p(block_tables), _p(start), n_remove, B, M,
                                               out_width, _stream(dev)), "block_table_remove_range")
    return out


def decode_step(out, query, key, value, key_cache, value_cache, slot_mapping, num_kv_heads, scale,
                block_tables, seq_lens, block_size, max_seq_len, alibi_slopes=None) -> None:
    """KV write of a decode batch + paged attention in ONE launch (see pe_decode_step).  query
    [S, Hq, D], key/value [S, Hkv, D], slot_mapping int64[S] (each slot in its sequence's last
    block), seq_lens int32[S] counting the new token."""
    dev = _need_gpu(out, query, key, value, key_cache, value_cache, slot_mapping, block_tables, seq_lens)
    S, Hq, D = query.shape
    nb = value_cache.shape[0]
    if key.shape != (S, num_kv_heads, D) or value.shape != key.shape:
        raise RuntimeError("decode_step: key/value must be [num_seqs, num_kv_heads, head_size]")
    for t in (query, key, value):
        if t.stride(2) != 1 or t.stride(1) != D or t.dtype != key_cache.dtype:
            raise RuntimeError("decode_step: dense rows of the cache dtype expected")
    if slot_mapping.dtype != torch.int64 or slot_mapping.numel() != S or not slot_mapping.is_contiguous():
        raise RuntimeError("decode_step: slot_mapping must be contiguous int64[num_seqs]")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32 or \
            not block_tables.is_contiguous() or not seq_lens.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("decode_step: int32 contiguous block_tables/seq_lens, contiguous out expected")
    lib = _lib.load()
    if not lib.pe_paged_decode_supports_fused_write(Hq, num_kv_heads, D, block_size, _dtype_code(query)):
        raise RuntimeError("decode_step: shape not supported by the fused kernel; use "
                           "reshape_and_cache + paged_attention_v1")
    need = lib.pe_paged_decode_workspace_bytes(S, Hq, num_kv_heads, D, block_size, max_seq_len)
    ws = _workspace(dev, need)
    _lib.check(lib.pe_decode_step(
        _p(out), _p(query), _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping),
        _p(block_tables), _p(seq_lens), _p(alibi_slopes), S, Hq, num_kv_heads, D, block_size, nb,
        block_tables.shape[1], query.stride(0), key.stride(0), value.stride(0), key_cache.stride(0),
        key_cache.stride(1), float(scale), int(max_seq_len), _dtype_code(query), _p(ws),
        0 if ws is None else ws.numel(), _stream(dev)), "decode_step")


def decode_step_rope(out, positions, query, key, value, key_cache, value_cache, slot_mapping,
                     num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                     cos_sin_cache, is_neox, alibi_slopes=None) -> None:
    """Rotary embedding of the step's query/key rows + KV write + paged attention in ONE launch
    (see pe_decode_step_rope): what ``rotary_embedding`` followed by ``decode_step`` computes,
    except that ``query`` and ``key`` are given unrotated and stay as they are.  positions int64
    [S], cos_sin_cache [max_position, rot_dim] of the query dtype; ``positions=None``:
    cos_sin_cache holds the step's rows already, row s for sequence s."""
    dev = _need_gpu(out, positions, query, key, value, key_cache, value_cache, slot_mapping,
                    block_tables, seq_lens, cos_sin_cache)
    S, Hq, D = query.shape
    nb = value_cache.shape[0]
    if key.shape != (S, num_kv_heads, D) or value.shape != key.shape:
        raise RuntimeError("decode_step_rope: key/value must be [num_seqs, num_kv_heads, head_size]")
    for t in (query, key, value):
        if t.stride(2) != 1 or t.stride(1) != D or t.dtype != key_cache.dtype:
            raise RuntimeError("decode_step_rope: dense rows of the cache dtype expected")
    if slot_mapping.dtype != torch.int64 or slot_mapping.numel() != S or not slot_mapping.is_contiguous():
        raise RuntimeError("decode_step_rope: slot_mapping must be contiguous int64[num_seqs]")
    if positions is None:
        if cos_sin_cache.dim() != 2 or cos_sin_cache.shape[0] < S:
            raise RuntimeError("decode_step_rope: without positions cos_sin_cache must hold one row per sequence")
    elif positions.dtype != torch.int64 or positions.numel() != S or not positions.is_contiguous():
        raise RuntimeError("decode_step_rope: positions must be contiguous int64[num_seqs]")
    if cos_sin_cache.dim() != 2 or not cos_sin_cache.is_contiguous() or cos_sin_cache.dtype != query.dtype:
        raise RuntimeError("decode_step_rope: cos_sin_cache [max_position, rot_dim] of the query dtype expected")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32 or \
            not block_tables.is_contiguous() or not seq_lens.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("decode_step_rope: int32 contiguous block_tables/seq_lens, contiguous out expected")
    lib = _lib.load()
    rot_dim = cos_sin_cache.shape[1]
    if not decode_step_rope_supported(Hq, num_kv_heads, D, block_size, query.dtype, rot_dim, is_neox):
        raise RuntimeError("decode_step_rope: shape or rot_dim not supported by the fused kernel; use "
                           "rotary_embedding_and_cache + paged_attention_v1")
    need = lib.pe_paged_decode_workspace_bytes(S, Hq, num_kv_heads, D, block_size, max_seq_len)
    ws = _workspace(dev, need)
    _lib.check(lib.pe_decode_step_rope(
        _p(out), _p(query), _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping),
        _p(block_tables), _p(seq_lens), _p(alibi_slopes), _p(positions), _p(cos_sin_cache), rot_dim,
        1 if is_neox else 0, cos_sin_cache.shape[0], S, Hq, num_kv_heads, D, block_size, nb,
        block_tables.shape[1], query.stride(0), key.stride(0), value.stride(0), key_cache.stride(0),
        key_cache.stride(1), float(scale), int(max_seq_len), _dtype_code(query), _p(ws),
        0 if ws is None else ws.numel(), _stream(dev)), "decode_step_rope")


def decode_step_rope_supported(num_heads: int, num_kv_heads: int, head_size: int, block_size: int,
                               dtype: torch.dtype, rot_dim: int, is_neox: bool) -> bool:
    """Whether ``decode_step_rope`` serves this geometry (otherwise: the separate launches)."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    lib = _lib.load()
    code = _lib.PE_BF16 if dtype == torch.bfloat16 else _lib.PE_F16
    if not lib.pe_paged_decode_supports_fused_write(num_heads, num_kv_heads, head_size, block_size, code):
        return False
    return (block_size in (8, 16, 32) and head_size in (64, 128) and 0 < rot_dim <= head_size
            and rot_dim % (64 if is_neox else 16) == 0)


# --------------------------------------------------------------------------------------
def _prewrap(fn, first: int, tail: tuple) -> tuple:
    """``tail`` (arguments ``first``.. of ``fn``) as instances of the declared C types."""
    types = fn.argtypes[first:first + len(tail)]
    return tuple(a if a is None else t(a) for a, t in zip(tail, types))


class DecodeLayerLauncher:
    """Pre-validated launcher of one decode layer (KV write + paged attention).

    All layers of a decode step share the step's metadata (slot mapping, block tables, kv
    lengths) and the shapes of q/k/v/kv_cache; validating them once per step and re-using the
    argument list keeps the per-layer host cost to two foreign calls, which matters because a
    decode layer is only tens of microseconds of GPU time.  ``matches`` tells whether a layer's
    tensors have the signature the launcher was built for."""

    def __init__(self, query, key, value, kv_cache, num_heads, num_kv_heads, head_size,
                 slot_mapping, zero_slots, block_tables, seq_lens, max_seq_len, scale):
        dev = _need_gpu(query, key, value, kv_cache, slot_mapping, zero_slots, block_tables, seq_lens)
        lib = _lib.load()
        self._lib = lib
        self.sig = self.signature(query, key, value, kv_cache)
        es = kv_cache.element_size()
        if kv_cache.dim() != 3 or kv_cache.shape[0] != 2 or not kv_cache.is_contiguous():
            raise RuntimeError("kv_cache must be a contiguous [2, num_blocks, block_elems] tensor")
        nb = kv_cache.shape[1]
        bs = kv_cache.shape[2] // (num_kv_heads * head_size)
        if slot_mapping.dtype != torch.int64 or block_tables.dtype != torch.int32 \
                or seq_lens.dtype != torch.int32:
            raise RuntimeError("decode launcher: slot_mapping int64, block_tables/seq_lens int32 expected")
        if not (slot_mapping.is_contiguous() and block_tables.is_contiguous() and seq_lens.is_contiguous()):
            raise RuntimeError("decode launcher: metadata tensors must be contiguous")
        S = query.shape[0]
        if block_tables.shape[0] != S or seq_lens.numel() != S or slot_mapping.numel() != S:
            raise RuntimeError("decode launcher: metadata does not match the batch")
        if query.dtype != kv_cache.dtype or key.dtype != kv_cache.dtype or value.dtype != kv_cache.dtype:
            raise RuntimeError("decode launcher: dtype mismatch")
        self.v_off = kv_cache.stride(0) * es
        self._out_numel = S * num_heads * head_size
        self._out_dtype = kv_cache.dtype
        dt = _dtype_code(query)
        stream = _stream(dev)
        nz = 0 if zero_slots is None else zero_slots.numel()
        need = lib.pe_paged_decode_workspace_bytes(S, num_heads, num_kv_heads, head_size, bs, max_seq_len)
        ws = _workspace(dev, need)
        self._keep = (slot_mapping, zero_slots, block_tables, seq_lens, ws)
        self._rc_tail = (_p(slot_mapping), S, key.stride(0), value.stride(0), None, 0, _p(zero_slots), nz,
                         num_kv_heads, head_size, bs, nb, dt, stream)
        kv_head_stride = head_size * bs
        self._pa_tail = (_p(block_tables), _p(seq_lens), None, S, num_heads, num_kv_heads, head_size, bs,
                         block_tables.shape[1], query.stride(0), kv_cache.stride(1), kv_head_stride,
                         float(scale), int(max_seq_len), dt, _p(ws), 0 if ws is None else ws.numel(), stream)
        # one launch per layer (KV write fused into the attention kernel) when nothing has to be
        # zeroed and the shape runs on the MFMA decode path
        self.fused = nz == 0 and bool(lib.pe_paged_decode_supports_fused_write(
            num_heads, num_kv_heads, head_size, bs, dt)) and query.stride(0) % 8 == 0 \
            and key.stride(0) % 8 == 0 and value.stride(0) % 8 == 0
        self._fs_tail = (_p(slot_mapping), _p(block_tables), _p(seq_lens), None, S, num_heads,
                         num_kv_heads, head_size, bs, nb, block_tables.shape[1], query.stride(0),
                         key.stride(0), value.stride(0), kv_cache.stride(1), kv_head_stride,
                         float(scale), int(max_seq_len), dt, _p(ws), 0 if ws is None else ws.numel(),
                         stream)
        # constant arguments converted to their C types once: the per-layer foreign call then
        # converts only the six tensor pointers
        # the rotating form takes the same arguments with (positions, cos_sin_cache, rot_dim,
        # is_neox, max_position) after alibi_slopes
        self._fr_head = _prewrap(lib.pe_decode_step_rope, 6, self._fs_tail[:4])
        self._fr_rest = _prewrap(lib.pe_decode_step_rope, 15, self._fs_tail[4:])
        self._rope_geom = (num_heads, num_kv_heads, head_size, bs, S)
        self._rope_ok = {}   # (rot_dim, is_neox) -> whether the rotating launch serves it
        self._step_cs = self._step_cs_key = None   # this step's rows of the cos/sin table
        self._rc_tail = _prewrap(lib.pe_reshape_and_cache, 4, self._rc_tail)
        self._pa_tail = _prewrap(lib.pe_paged_decode, 4, self._pa_tail)
        self._fs_tail = _prewrap(lib.pe_decode_step, 6, self._fs_tail)

    @staticmethod
    def signature(query, key, value, kv_cache):
        # rows / elements / row strides: the same for the [T, H*D] tensors a layer receives and
        # their [T, H, D] views
        return (query.shape[0], query.numel(), query.stride(0), key.numel(), key.stride(0),
                value.stride(0), kv_cache.shape, kv_cache.dtype, query.dtype, query.device)

    def matches(self, query, key, value, kv_cache) -> bool:
        return self.sig == self.signature(query, key, value, kv_cache)

    # measurement hook: when set to a list, run() brackets the attention launch with events on
    # the launch stream and appends (start, end)
    event_sink = None

    def run(self, query, key, value, kv_cache, out, rope=None) -> None:
        """``rope`` = (positions int64 [S], cos_sin_cache [max_position, rot_dim], is_neox): the
        layer's query/key rows arrive unrotated.  On the one-launch path the rotation runs inside
        the attention kernel and query/key stay as they are; otherwise they are rotated in place
        first, as the reference's rotary_embedding does."""
        # the kernels write a dense [S, Hq, D] tensor of the cache dtype at out's address
        if out.numel() != self._out_numel or out.dtype != self._out_dtype or not out.is_contiguous():
            raise RuntimeError("decode launcher: output must be a contiguous tensor of "
                               f"{self._out_numel} {self._out_dtype} elements")
        kc = kv_cache.data_ptr()
        vc = kc + self.v_off
        lib = self._lib
        sink = DecodeLayerLauncher.event_sink
        if rope is not None:
            positions, cos_sin, is_neox = rope
            Hq, Hkv, D, bs, S = self._rope_geom
            if positions.dtype != torch.int64 or positions.numel() != S or not positions.is_contiguous() \
                    or cos_sin.dim() != 2 or cos_sin.dtype != self._out_dtype or not cos_sin.is_contiguous():
                raise RuntimeError("decode launcher: positions int64 [num_seqs] and cos_sin_cache "
                                   "[max_position, rot_dim] of the cache dtype expected")
            rot_key = (cos_sin.shape[1], bool(is_neox))
            in_launch = self._rope_ok.get(rot_key)
            if in_launch is None:
                in_launch = self._rope_ok[rot_key] = decode_step_rope_supported(
                    Hq, Hkv, D, bs, self._out_dtype, cos_sin.shape[1], is_neox)
            if self.fused and in_launch:
                if sink is not None:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                # all layers of the step rotate by the same positions: their rows of the model's
                # table are gathered once (first layer) and handed to the kernel without positions
                step_cs = self._step_cs
                if step_cs is None or self._step_cs_key != (positions.data_ptr(), cos_sin.data_ptr()):
                    step_cs = cos_sin.index_select(0, positions.clamp(0, cos_sin.shape[0] - 1))
                    self._step_cs, self._step_cs_key = step_cs, (positions.data_ptr(), cos_sin.data_ptr())
                _lib.check(lib.pe_decode_step_rope(
                    out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), kc, vc,
                    *self._fr_head, None, step_cs.data_ptr(), cos_sin.shape[1],
                    1 if is_neox else 0, S, *self._fr_rest), "decode_step_rope")
                if sink is not None:
                    e.record()
                    sink.append((s, e))
                return
            rotary_embedding(positions, query, key, D, cos_sin, is_neox)
        if self.fused:
            if sink is not None:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
            _lib.check(lib.pe_decode_step(out.data_ptr(), query.data_ptr(), key.data_ptr(),
                                          value.data_ptr(), kc, vc, *self._fs_tail), "decode_step")
            if sink is not None:
                e.record()
                sink.append((s, e))
            return
        _lib.check(lib.pe_reshape_and_cache(key.data_ptr(), value.data_ptr(), kc, vc, *self._rc_tail),
                   "reshape_and_cache")
        if sink is None:
            _lib.check(lib.pe_paged_decode(out.data_ptr(), query.data_ptr(), kc, vc, *self._pa_tail),
                       "paged_attention")
        else:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _lib.check(lib.pe_paged_decode(out.data_ptr(), query.data_ptr(), kc, vc, *self._pa_tail),
                       "paged_attention")
            e.record()
            sink.append((s, e))


# -------------------------------------------------------------------------------------- torch.library
_registered = False


def register_torch_ops() -> None:
    """Register the operators as torch.ops.paged_evict.* with the reference's schemas
    (csrc/torch_bindings.cpp:27-51,433-450).  Idempotent."""
    global _registered
    if _registered:
        return
    lib = torch.library.Library("paged_evict", "DEF")
    lib.define("paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache,"
               " int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens,"
               " int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype,"
               " float k_scale, float v_scale, int tp_rank, int blocksparse_local_blocks,"
               " int blocksparse_vert_stride, int blocksparse_block_size,"
               " int blocksparse_head_sliding_step) -> ()")
    lib.define("paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits,"
               " Tensor! tmp_out, Tensor query, Tensor key_cache, Tensor value_cache,"
               " int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens,"
               " int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype,"
               " float k_scale, float v_scale, int tp_rank, int blocksparse_local_blocks,"
               " int blocksparse_vert_stride, int blocksparse_block_size,"
               " int blocksparse_head_sliding_step) -> ()")
    lib.define("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache,"
               " Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype,"
               " float k_scale, float v_scale) -> ()")
    lib.define("copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches,"
               " Tensor block_mapping) -> ()")
    lib.define("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()")
    # the step in front of the path (csrc/torch_bindings.cpp:141-145)
    lib.define("rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size,"
               " Tensor cos_sin_cache, bool is_neox) -> ()")
    lib.impl("rotary_embedding", rotary_embedding, "CUDA")
    lib.impl("paged_attention_v1", paged_attention_v1, "CUDA")
    lib.impl("paged_attention_v2", paged_attention_v2, "CUDA")
    lib.impl("reshape_and_cache", lambda k, v, kc, vc, sm, dt, ks, vs: reshape_and_cache(
        k, v, kc, vc, sm, dt, ks, vs), "CUDA")
    lib.impl("copy_blocks", copy_blocks, "CUDA")
    lib.impl("swap_blocks", swap_blocks, "CompositeExplicitAutograd")
    register_torch_ops._lib = lib  # keep alive
    _registered = True
