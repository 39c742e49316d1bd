"""TEST INFRASTRUCTURE ONLY — CPU oracle for the paged-KV-eviction attention path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package never does.

Two halves:

* ctypes wrappers over ``oracle/libpe_oracle.so`` (``oracle/pe_oracle.c``): KV write,
  block copy, single-query paged attention, block score, block-table compaction,
  varlen causal prefill.  Each C function cites the reference lines it restates.
* small pure-Python / torch-CPU restatements of the integer and index logic of
  the eviction path (prompt-prune index maps, pruned lengths, decode triggers,
  victim choice, slot mapping, block-table counters).  Citations are relative to
  the reference checkout.

Pinning of this oracle against the reference's own code and fixtures is done by
``tests/test_oracle_golden.py`` (golden vectors made by importing the reference's
``KVCachePruner`` / ``BlockL2NormManager``) and ``tests/test_oracle_vs_ref.py``
(the reference's CPU kernels compiled from its sources into ``oracle/_ref``).
"""
from __future__ import annotations

import ctypes
import os
import random
import subprocess
from typing import List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PE_ORACLE_LIB: a differently built copy of the same source (tools/sanitize_oracle.sh builds one
# with the undefined-behaviour sanitizer and runs the CPU tests through it)
_LIB_PATH = os.environ.get("PE_ORACLE_LIB") or os.path.join(_HERE, "libpe_oracle.so")
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def build(force: bool = False) -> str:
    """Compile oracle/pe_oracle.c with gcc (a few seconds)."""
    src = os.path.join(_HERE, "pe_oracle.c")
    if os.environ.get("PE_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-fno-fast-math", "-ffp-contract=off",
             "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_block_score.restype = ctypes.c_float
        L.orc_block_score_from_token_sums.restype = ctypes.c_float
        L.orc_bf16_to_f32.restype = ctypes.c_float
        L.orc_f16_to_f32.restype = ctypes.c_float
        L.orc_f32_to_bf16.restype = ctypes.c_uint16
        L.orc_f32_to_f16.restype = ctypes.c_uint16
        L.orc_f32_to_bf16.argtypes = [ctypes.c_float]
        L.orc_f32_to_f16.argtypes = [ctypes.c_float]
        L.orc_bf16_to_f32.argtypes = [ctypes.c_uint16]
        L.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
        _lib = L
    return _lib


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _i64(v):
    return ctypes.c_int64(int(v))


# --------------------------------------------------------------------------- C wrappers

def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping) -> None:
    """key/value [T, H, D]; key_cache [nb, H, D/x, bs, x]; value_cache [nb, H, D, bs]."""
    assert key_cache.is_contiguous() and value_cache.is_contiguous()
    slot_mapping = slot_mapping.to(torch.int64).contiguous()
    T, H, D = key.shape
    bs, x = key_cache.shape[3], key_cache.shape[4]
    lib().orc_reshape_and_cache(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), _i64(T),
        _i64(key.stride(0)), _i64(value.stride(0)), ctypes.c_int(H), ctypes.c_int(D),
        ctypes.c_int(bs), ctypes.c_int(x), ctypes.c_int(_DT[key.dtype]))


def copy_blocks(key_cache, value_cache, block_mapping) -> None:
    block_mapping = block_mapping.to(torch.int64).contiguous()
    lib().orc_copy_blocks(_p(key_cache), _p(value_cache), _p(block_mapping),
                          _i64(block_mapping.shape[0]), _i64(key_cache[0].numel()),
                          ctypes.c_int(_DT[key_cache.dtype]))


def paged_attention(query, key_cache, value_cache, block_tables, seq_lens, scale,
                    alibi_slopes=None) -> torch.Tensor:
    """query [S, Hq, D] -> out [S, Hq, D] (same dtype)."""
    S, Hq, D = query.shape
    nb, Hkv, _, bs, x = key_cache.shape
    block_tables = block_tables.to(torch.int32).contiguous()
    seq_lens = seq_lens.to(torch.int32).contiguous()
    out = torch.zeros((S, Hq, D), dtype=query.dtype)
    if alibi_slopes is not None:
        alibi_slopes = alibi_slopes.to(torch.float32).contiguous()
    lib().orc_paged_attention(
        _p(out), _p(query), _p(key_cache), _p(value_cache), _p(block_tables), _p(seq_lens),
        _p(alibi_slopes), ctypes.c_int(S), ctypes.c_int(Hq), ctypes.c_int(Hkv),
        ctypes.c_int(D), ctypes.c_int(bs), ctypes.c_int(x),
        ctypes.c_int(block_tables.shape[1]), _i64(query.stride(0)),
        _i64(key_cache.stride(0)), _i64(key_cache.stride(1)), ctypes.c_float(scale),
        ctypes.c_int(_DT[query.dtype]))
    return out


def block_score(values: torch.Tensor, block_size: int, num_kv_heads: int, head_size: int,
                layout: int) -> float:
    """layout 0: [bs,H,D] token-major; 1: cache block [H,D,bs] logical rows;
    2: cache block re-viewed as [-1,H,D] the way the reference's decode call site does."""
    values = values.contiguous()
    return float(lib().orc_block_score(_p(values), ctypes.c_int(block_size),
                                       ctypes.c_int(num_kv_heads), ctypes.c_int(head_size),
                                       ctypes.c_int(layout), ctypes.c_int(_DT[values.dtype])))


def rotary_embedding(positions: torch.Tensor, rows: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> torch.Tensor:
    """Rotated copy of rows [T, heads*head_size] (or [T, heads, head_size])."""
    T = positions.numel()
    out = rows.reshape(T, -1).contiguous().clone()
    cs = cos_sin_cache.contiguous()
    pos = positions.to(torch.int64).contiguous()
    lib().orc_rotary_embedding(_p(pos), _p(out), _p(cs), ctypes.c_int(T),
                               ctypes.c_int(out.shape[1] // head_size), ctypes.c_int(head_size),
                               ctypes.c_int64(out.shape[1]), ctypes.c_int(cs.shape[1]),
                               ctypes.c_int(1 if is_neox else 0), ctypes.c_int(_DT[rows.dtype]))
    return out.view(rows.shape)


def block_token_norm_sums(values: torch.Tensor, block_size: int, num_kv_heads: int, head_size: int,
                          layout: int) -> torch.Tensor:
    """Per token of one block, the f32 sum over the given heads of the rounded norms."""
    values = values.contiguous()
    out = torch.zeros(block_size, dtype=torch.float32)
    lib().orc_block_token_norm_sums(_p(values), _p(out), ctypes.c_int(block_size),
                                    ctypes.c_int(num_kv_heads), ctypes.c_int(head_size),
                                    ctypes.c_int(layout), ctypes.c_int(_DT[values.dtype]))
    return out


def block_score_from_token_sums(sums: torch.Tensor, total_kv_heads: int, dtype: torch.dtype) -> float:
    sums = sums.to(torch.float32).contiguous()
    return float(lib().orc_block_score_from_token_sums(_p(sums), ctypes.c_int(sums.numel()),
                                                       ctypes.c_int(total_kv_heads),
                                                       ctypes.c_int(_DT[dtype])))


def compact_block_tables(tables: torch.Tensor, rmv_idx: Sequence[int], compat_q1: bool = False,
                         out_width: Optional[int] = None):
    """out_width None: the reference's shape [B, M-1]."""
    tables = tables.to(torch.int32).contiguous()
    B, M = tables.shape
    w = max(M - 1, 0) if out_width is None else out_width
    rmv = torch.tensor(list(rmv_idx), dtype=torch.int32)
    out = torch.zeros((B, w), dtype=torch.int32)
    lib().orc_compact_block_tables(_p(tables), _p(rmv), _p(out), ctypes.c_int(B),
                                   ctypes.c_int(M), ctypes.c_int(w),
                                   ctypes.c_int(1 if compat_q1 else 0))
    return out


def surviving_width(table_lens: Sequence[int], rmv_idx: Sequence[int]) -> int:
    """Longest block-table row after this step's removals."""
    return max(n - (1 if r >= 0 else 0) for n, r in zip(table_lens, rmv_idx))


def prefill_attention(query, key, value, q_lens: Sequence[int], kv_lens: Sequence[int], scale):
    """query [Tq,Hq,D], key/value [Tkv,Hkv,D]; causal, top-left aligned when Lq != Lkv."""
    query, key, value = query.contiguous(), key.contiguous(), value.contiguous()
    cu_q = torch.tensor([0] + list(torch.tensor(list(q_lens)).cumsum(0)), dtype=torch.int32)
    cu_kv = torch.tensor([0] + list(torch.tensor(list(kv_lens)).cumsum(0)), dtype=torch.int32)
    out = torch.zeros_like(query)
    lib().orc_prefill_attention(
        _p(out), _p(query), _p(key), _p(value), _p(cu_q), _p(cu_kv), ctypes.c_int(len(q_lens)),
        ctypes.c_int(query.shape[1]), ctypes.c_int(key.shape[1]), ctypes.c_int(query.shape[2]),
        ctypes.c_float(scale), ctypes.c_int(_DT[query.dtype]))
    return out


# --------------------------------------------------------------------------- prompt prune

def prompt_slices(q_len: int, budget: int, block_size: int) -> Tuple[int, int, int]:
    """(end_first, end_middle, n_prune) of KVCachePruner.prune_prompt,
    vllm/attention/kvcache_prunner.py:78-96 (identical in :118-135, :185-200)."""
    rem = q_len % block_size
    end_first = block_size
    end_mid = q_len - block_size - rem
    mid_tokens = end_mid - end_first
    mid_keep = budget - 2 * block_size
    n_prune = mid_tokens - mid_keep
    if n_prune < 0:
        n_prune = 0
    return end_first, end_mid, n_prune


def pruned_length(seq_len: int, budget: int, block_size: int) -> int:
    """KVCachePruner.get_pruned_length, kvcache_prunner.py:284-326 (same for every method)."""
    if seq_len <= budget:
        return seq_len
    rem = seq_len % block_size
    end_mid = seq_len - block_size - rem
    mid_keep = max(budget - block_size - block_size, 0)
    return block_size + mid_keep + (seq_len - end_mid)


def num_required_blocks_after_prune(q_len: int, budget: int, block_size: int,
                                    initial_blocks: int = 1) -> Tuple[int, int]:
    """get_num_required_blocks_after_prune_promt, vllm/core/page_evict_kv_util.py:6-36."""
    if q_len < budget:
        return -(-q_len // block_size), q_len
    rem = q_len % block_size
    end_first = initial_blocks * block_size
    end_mid = q_len - block_size - rem
    mid_keep = max(budget - initial_blocks * block_size - block_size, 0)
    total = end_first + mid_keep + (q_len - end_mid)
    return -(-total // block_size), total


def streaming_keep_indices(q_len: int, budget: int, block_size: int) -> List[int]:
    """Kept token positions of the streamingLLM prompt prune, kvcache_prunner.py:76-115."""
    if q_len <= budget:
        return list(range(q_len))
    end_first, end_mid, n_prune = prompt_slices(q_len, budget, block_size)
    return (list(range(0, end_first)) + list(range(end_first + n_prune, end_mid))
            + list(range(end_mid, q_len)))


def token_scores(keys: torch.Tensor, values: torch.Tensor, method: str,
                 formula: str = "value_l2") -> torch.Tensor:
    """KVCachePruner.get_token_score, kvcache_prunner.py:29-49; ``formula`` is the sub method
    of global/local (:13 sets value_l2; the other arms are :35-43)."""
    if method in ("global", "local"):
        if formula == "value_l2":
            return torch.norm(values, p=2, dim=-1)
        if formula == "key_l2":
            return torch.norm(keys, p=2, dim=-1)
        if formula == "value_l2_plus_key_l2":      # value_l2(keys) + key_l2(values), :26-27
            return torch.norm(keys, p=2, dim=-1) + torch.norm(values, p=2, dim=-1)
        if formula == "inverse_key_l2":
            return torch.div(1, torch.norm(keys, p=2, dim=-1) + 1e-8)
        raise ValueError(formula)
    if method == "inverse_key_l2":
        return torch.div(1, torch.norm(keys, p=2, dim=-1) + 1e-8)
    raise ValueError(method)


def block_score_formula(keys: torch.Tensor, values: torch.Tensor, formula: str) -> torch.Tensor:
    """KVCachePruner.get_block_score for one [bs, H, D] block, kvcache_prunner.py:51-66: 0-d
    tensor of the input dtype.  The norm / head-mean / token-sum chain is the C restatement
    (orc_block_score); the sum and the reciprocal of the key arms are torch's 0-d arithmetic."""
    bs, H, D = values.shape
    dt = values.dtype

    def nrm(x):
        return torch.tensor(block_score(x, bs, H, D, 0), dtype=torch.float32).to(dt)
    if formula == "value_l2":
        return nrm(values)
    if formula == "key_l2":
        return nrm(keys)
    if formula == "value_l2_plus_key_l2":
        return nrm(values) + nrm(keys)
    if formula == "inverse_key_l2":
        return 1 / (nrm(keys) + 1e-8)
    raise ValueError(formula)


def topk_keep_mask(scores: torch.Tensor, n_prune: int) -> torch.Tensor:
    """[T,H] bool keep-mask: per head drop the n_prune lowest scores
    (kvcache_prunner.py:141-145).  Ties are broken towards the lower token index
    (a stable ascending sort); torch.topk leaves the tie order unspecified, so tie
    cases are outside what the reference pins."""
    T, H = scores.shape
    keep = torch.ones((T, H), dtype=torch.bool)
    if n_prune <= 0:
        return keep
    order = torch.sort(scores.float(), dim=0, stable=True).indices  # ascending
    keep.scatter_(0, order[:n_prune], False)
    return keep


def prompt_prune(keys: torch.Tensor, values: torch.Tensor, method: str, budget: int,
                 block_size: int, compat_q2: bool = False, formula: str = "value_l2"):
    """KVCachePruner.prune_prompt, kvcache_prunner.py:69-165.  Returns (keys, values).

    compat_q2=False: per-head compaction (kept tokens of head h stay in head h).
    compat_q2=True: the reference's ``middle[mask].view(-1, H, D)`` (:148-149), which
    re-packs kept (token, head) rows in row-major order across heads."""
    L, H, D = keys.shape
    if L <= budget:
        return keys, values
    if method in ("streamingLLM", "streamingLLM-1"):
        idx = torch.tensor(streaming_keep_indices(L, budget, block_size), dtype=torch.long)
        return keys[idx], values[idx]
    end_first, end_mid, n_prune = prompt_slices(L, budget, block_size)
    mk, mv = keys[end_first:end_mid], values[end_first:end_mid]
    if n_prune > 0:
        keep = topk_keep_mask(token_scores(mk, mv, method, formula), n_prune)
        if compat_q2:
            pk, pv = mk[keep].view(-1, H, D), mv[keep].view(-1, H, D)
        else:
            n_keep = mk.shape[0] - n_prune
            pk = torch.empty((n_keep, H, D), dtype=keys.dtype)
            pv = torch.empty((n_keep, H, D), dtype=keys.dtype)
            for h in range(H):
                sel = keep[:, h].nonzero().flatten()
                pk[:, h] = mk[sel, h]
                pv[:, h] = mv[sel, h]
    else:
        pk, pv = mk, mv
    return (torch.cat([keys[:end_first], pk, keys[end_mid:]], dim=0),
            torch.cat([values[:end_first], pv, values[end_mid:]], dim=0))


# --------------------------------------------------------------------------- slot mapping

def prompt_slot_mapping(block_table: Sequence[int], query_len: int, block_size: int) -> List[int]:
    """TorchCUDASDPAMetadataBuilder._compute_slot_mapping, prompt arm,
    vllm/attention/backends/torch_cuda_sdpa.py:577-589."""
    out: List[int] = []
    rem = query_len % block_size
    if rem == 0:
        for b in block_table:
            out.extend(b * block_size + i for i in range(block_size))
    else:
        for b in block_table[:-1]:
            out.extend(b * block_size + i for i in range(block_size))
        out.extend(block_table[-1] * block_size + i for i in range(rem))
    return out


def decode_slot(block_table: Sequence[int], context_len: int, block_size: int) -> int:
    """Decode arm of the same function, torch_cuda_sdpa.py:591-598."""
    return block_table[-1] * block_size + context_len % block_size


# --------------------------------------------------------------------------- decode decisions

def decode_trigger(seq_len: int, budget: int, block_size: int) -> bool:
    """Eviction trigger, torch_cuda_sdpa.py:1390-1403,1491-1498 and
    vllm/worker/model_runner.py:541."""
    return seq_len > budget and seq_len % block_size == 0


def decode_streaming(seq_lens: Sequence[int], block_tables: Sequence[Sequence[int]], budget: int,
                     block_size: int, method: str):
    """_prune_decode_reqs_streamingLLM, torch_cuda_sdpa.py:1375-1433.
    Returns (rmv_idx per sequence (-1 none), zero-slot ids)."""
    rmv = [-1] * len(seq_lens)
    zero_slots: List[int] = []
    for i, L in enumerate(seq_lens):
        if L <= budget:
            continue
        if L % block_size != 0:
            if method == "streamingLLM-1":
                zero_slots.append(block_tables[i][1] * block_size + L % block_size - 1)
        else:
            rmv[i] = 1
    return rmv, zero_slots


def decode_inverse_key_l2(seq_lens: Sequence[int], block_tables: Sequence[Sequence[int]],
                          budget: int, block_size: int, rng) -> List[int]:
    """_prune_decode_reqs_inverseKeyL2, torch_cuda_sdpa.py:1435-1475: every sequence whose
    logical length exceeds the budget gets one slot zeroed per step - a random block of the
    middle of its table (first and last excluded, :1451), then a random slot of it (:1452); no
    block is removed and kv lengths are not adjusted (vllm/worker/model_runner.py:545-547).
    ``rng`` supplies ``randint`` (the reference draws from the global ``random`` module, seeded
    1234 by the impl constructor, :796).  Returns the slot ids in batch order."""
    slots: List[int] = []
    for L, ids in zip(seq_lens, block_tables):
        if L <= budget:
            continue
        b = rng.randint(1, len(ids) - 2)
        s = rng.randint(0, block_size - 1)
        slots.append(ids[b] * block_size + s)
    return slots


def argmin_first(scores: Sequence[float], start: int, end: int) -> int:
    """Arg-min with first-min-wins (strict <), torch_cuda_sdpa.py:1536-1541."""
    r = start
    for i in range(start, end):
        if scores[i] < scores[r]:
            r = i
    return r


def decode_topk_victim(scores: List[float], new_score: float, method: str, topk_blocks: int,
                       drop_victim_score: bool) -> int:
    """Score-based victim of _prune_decode_reqs_topk (random off),
    torch_cuda_sdpa.py:1517-1544: append the just-filled block's score, then arg-min over
    [1, 1+topk) (local) or [1, n-1) (global).  ``drop_victim_score`` deletes the
    victim's entry afterwards (the behaviour of the commented-out call at :1546);
    False keeps HEAD's behaviour, where the list is never shortened."""
    scores.append(new_score)
    if method == "local":
        s, e = 1, 1 + topk_blocks
    else:
        s, e = 1, len(scores) - 1
    e = min(e, len(scores))
    v = argmin_first(scores, s, e)
    if drop_victim_score:
        del scores[v]
    return v


class RandomVictimStream:
    """HEAD's random victim choice (enable_random_evict=True, torch_cuda_sdpa.py:802,
    1502-1515): every layer's impl draws from the *global* ``random`` module, seeded
    1234 by the last impl constructed (:796); only layer 0's draw is committed
    (vllm/worker/model_runner.py:1797-1802).  Layer 0 draws first in each step, the
    other ``num_layers-1`` layers then consume the same number of draws with the same
    arguments."""

    def __init__(self, num_layers: int, seed: int = 1234):
        self.rng = random.Random(seed)
        self.num_layers = num_layers

    def step(self, ranges: Sequence[Tuple[int, int]]) -> List[int]:
        first = [self.rng.randint(a, b) for a, b in ranges]
        for _ in range(self.num_layers - 1):
            for a, b in ranges:
                self.rng.randint(a, b)
        return first


# --------------------------------------------------------------------------- percentage prune type

def percentage_prompt_slices(q_len: int, block_size: int, evict_freq: int, initial_blocks: int = 1,
                             num_block_merge: int = 2) -> Tuple[int, int, int]:
    """(end_first, end_middle, n_prune) of the README-era prompt prune,
    vllm/attention/kvcache_prunner_old.py:143-176 (allocation side:
    vllm/core/page_evict_kv_util_bak.py:10-31); n_prune = 0 for a prompt too short to prune."""
    if q_len <= (initial_blocks + num_block_merge) * block_size:
        return q_len, q_len, 0
    rem = q_len % block_size
    end_first = initial_blocks * block_size
    end_mid = q_len - num_block_merge * block_size - rem
    n_prune = int((end_mid - end_first) / evict_freq)
    n_prune = block_size * (n_prune // block_size)
    return end_first, end_mid, n_prune


def percentage_keep_indices(q_len: int, block_size: int, evict_freq: int, initial_blocks: int = 1,
                            num_block_merge: int = 2) -> List[int]:
    """Kept token positions of that prune with the streamingLLM prompt method
    (kvcache_prunner_old.py:176-196): the oldest n_prune middle tokens go."""
    end_first, _, n_prune = percentage_prompt_slices(q_len, block_size, evict_freq, initial_blocks,
                                                     num_block_merge)
    return list(range(0, end_first)) + list(range(end_first + n_prune, q_len))


def percentage_prompt_prune(keys: torch.Tensor, values: torch.Tensor, method: str, block_size: int,
                            evict_freq: int, initial_blocks: int = 1, num_block_merge: int = 2,
                            compat_q2: bool = False):
    """prune_prompt of the percentage type, kvcache_prunner_old.py:137-262: streamingLLM drops
    the oldest middle tokens, "value_l2" / "inverse_key_l2" the lowest-score ones per head
    (compat_q2: the reference's cross-head re-packing ``middle[mask].view(-1, H, D)``, :246-247)."""
    L, H, D = keys.shape
    end_first, end_mid, n_prune = percentage_prompt_slices(L, block_size, evict_freq, initial_blocks,
                                                           num_block_merge)
    if n_prune == 0:
        return keys, values
    if method == "streamingLLM":
        idx = torch.tensor(percentage_keep_indices(L, block_size, evict_freq, initial_blocks,
                                                   num_block_merge), dtype=torch.long)
        return keys[idx], values[idx]
    mk, mv = keys[end_first:end_mid], values[end_first:end_mid]
    if method in ("value_l2", "inverse_key_l2"):
        scores = token_scores(mk, mv, "global" if method == "value_l2" else method)
    else:
        scores = readme_token_scores(mk, mv, method)
    keep = topk_keep_mask(scores, n_prune)
    if compat_q2:
        pk, pv = mk[keep].view(-1, H, D), mv[keep].view(-1, H, D)
    else:
        n_keep = mk.shape[0] - n_prune
        pk = torch.empty((n_keep, H, D), dtype=keys.dtype)
        pv = torch.empty((n_keep, H, D), dtype=keys.dtype)
        for h in range(H):
            sel = keep[:, h].nonzero().flatten()
            pk[:, h] = mk[sel, h]
            pv[:, h] = mv[sel, h]
    return (torch.cat([keys[:end_first], pk, keys[end_mid:]], dim=0),
            torch.cat([values[:end_first], pv, values[end_mid:]], dim=0))


def percentage_decode_blocks(seq_len: int, table_len: int, block_size: int, evict_freq: int,
                             method: str, initial_blocks: int = 1, num_block_merge: int = 2):
    """Decode step of the percentage type: (first freed table column or -1, columns freed,
    columns merged).  Trigger: logical length a multiple of evict_freq blocks
    (torch_cuda_sdpa.py:1310) and more than initial_blocks + num_block_merge blocks held
    (page_evict_kv_util_bak.py:69,88).  streamingLLM frees [ib, ib+nbm-1) (:72-74); value_l2
    merges [ib, ib+nbm) into block ib (kvcache_prunner_old.py:281-284) and frees [ib+1, ib+nbm)
    (page_evict_kv_util_bak.py:91-93)."""
    if seq_len % (block_size * evict_freq) != 0 or table_len <= initial_blocks + num_block_merge:
        return -1, 0, None
    if method == "streamingLLM":
        return initial_blocks, num_block_merge - 1, None
    return initial_blocks + 1, num_block_merge - 1, (initial_blocks, initial_blocks + num_block_merge)


# get_score of the README-era pruner, vllm/attention/kvcache_prunner_old.py:23-134, by method name:
# (operation, first norm, second norm) over the row norms kl1 = ||K||_1, kl2 = ||K||_2, vl1 = ||V||_1,
# vl2 = ||V||_2.  AS WRITTEN there: value_*_div_key_* / value_l1_plus_key_l1 hand (key, value) to
# helpers that apply the "value" norm to their first argument and the "key" norm to the second
# (:59-72), and key_l1_div_value_l1 takes the key's L2 norm (:47-48).
README_SCORES = {
    "value_l2": ("single", "vl2", None), "key_l2": ("single", "kl2", None),
    "key_l1": ("single", "kl1", None), "value_l1": ("single", "vl1", None),
    "value_l2_plus_key_l2": ("plus", "kl2", "vl2"), "value_l1_plus_key_l1": ("plus", "kl1", "vl1"),
    "inverse_key_l2": ("inverse", "kl2", None), "inverse_key_l1": ("inverse", "kl1", None),
    "key_l1_div_value_l1": ("div", "kl2", "vl1"), "key_l1_div_value_l2": ("div", "kl1", "vl2"),
    "key_l2_div_value_l1": ("div", "kl2", "vl1"), "key_l2_div_value_l2": ("div", "kl2", "vl2"),
    "value_l1_div_key_l1": ("div", "kl1", "vl1"), "value_l1_div_key_l2": ("div", "kl1", "vl2"),
    "value_l2_div_key_l1": ("div", "kl2", "vl1"), "value_l2_div_key_l2": ("div", "kl2", "vl2"),
    "cosine": ("cosine", None, None), "streamingLLM": ("position", None, None),
}


def readme_token_scores(keys: torch.Tensor, values: torch.Tensor, method: str) -> torch.Tensor:
    """[T, H] scores of get_score (kvcache_prunner_old.py:77-134) in the input dtype.  Every
    reduction runs in f32 in ascending element order and every operation's result is rounded to
    the storage type where torch rounds it (after each norm, after ``+ 1e-8``, after the
    division / sum; for the cosine after each quotient, each product and the final sum) - the
    arithmetic the kernels use; torch's own vectorised sums may differ in the last f32 bit
    before that rounding."""
    import numpy as np
    dt = keys.dtype
    T, H, D = keys.shape
    f32 = np.float32

    def rt(a):       # round an f32 array to the storage type
        return torch.from_numpy(np.ascontiguousarray(a, dtype=f32)).to(dt).float().numpy()

    def seq_sum(a):  # f32 sum over the last axis, one element after the other
        return np.cumsum(a.astype(f32), axis=-1, dtype=f32)[..., -1]
    k = keys.float().numpy().astype(f32)
    v = values.float().numpy().astype(f32)
    op, a, b = README_SCORES[method]
    if op == "position":
        return torch.arange(1, T + 1, dtype=torch.float32).view(-1, 1).expand(-1, H).to(dt)
    if op == "cosine":
        kn = np.maximum(rt(np.sqrt(seq_sum(k * k))), f32(1e-12))
        vn = np.maximum(rt(np.sqrt(seq_sum(v * v))), f32(1e-12))
        prod = rt(rt(k / kn[..., None]) * rt(v / vn[..., None]))
        return torch.from_numpy(rt(seq_sum(prod))).to(dt)

    def norm(which):
        x = k if which[0] == "k" else v
        return rt(seq_sum(np.abs(x))) if which.endswith("1") else rt(np.sqrt(seq_sum(x * x)))
    na = norm(a)
    with np.errstate(divide="ignore", invalid="ignore"):   # fp16: 0 + 1e-8 rounds to 0, 1/0 = inf as in torch
        if op == "single":
            out = na
        elif op == "inverse":
            out = rt(f32(1.0) / rt(na + f32(1e-8)))
        elif op == "plus":
            out = rt(na + norm(b))
        else:
            out = rt(na / rt(norm(b) + f32(1e-8)))
    return torch.from_numpy(out).to(dt)


def merge_keep_sets(values: torch.Tensor, n_keep: int, keys: Optional[torch.Tensor] = None,
                    method: str = "value_l2") -> List[List[int]]:
    """Per head, the token indices that survive prune_oldest_block
    (kvcache_prunner_old.py:302-309): scores ||V[t, h, :]||_2 (sum of squares in ascending
    element order, f32, rounded to the storage type) or, with ``keys`` and ``method``, any other
    score of get_score; the T - n_keep lowest go, equal scores lowest token first (torch.topk
    leaves that order unspecified)."""
    import numpy as np
    T, H, D = values.shape
    if method != "value_l2":
        score = readme_token_scores(keys, values, method).float()
        order = torch.sort(score, dim=0, stable=True).indices
        keep = torch.ones((T, H), dtype=torch.bool)
        keep.scatter_(0, order[:T - n_keep], False)
        return [keep[:, h].nonzero().flatten().tolist() for h in range(H)]
    sq = values.float().numpy().astype(np.float32)
    sq = sq * sq
    ss = np.cumsum(sq, axis=-1, dtype=np.float32)[..., -1]
    score = torch.from_numpy(np.sqrt(ss)).to(values.dtype).float()
    order = torch.sort(score, dim=0, stable=True).indices  # ascending, [T, H]
    keep = torch.ones((T, H), dtype=torch.bool)
    keep.scatter_(0, order[:T - n_keep], False)
    return [keep[:, h].nonzero().flatten().tolist() for h in range(H)]


def merge_blocks_in_cache(key_cache: torch.Tensor, value_cache: torch.Tensor,
                          block_ids: Sequence[int], method: str = "value_l2") -> None:
    """The decode-time merge on the paged layouts (key_cache [nb, H, D/x, bs, x], value_cache
    [nb, H, D, bs]): logical rows of the listed blocks -> per-head selection -> the kept rows,
    in order, into block_ids[0] (the write-back of torch_cuda_sdpa.py:1350-1363).  Per-head
    selection on logical rows is what kvcache_prunner_old.py:302-313 intends; as written it
    works on a flat re-view of the blocks, repacks rows across heads and returns the kept keys
    as values."""
    nb, H, Dx, bs, x = key_cache.shape
    D = Dx * x
    ids = torch.tensor(list(block_ids), dtype=torch.long)
    # [n, H, D/x, bs, x] -> [n, bs, H, D]
    k = key_cache[ids].permute(0, 3, 1, 2, 4).reshape(len(ids) * bs, H, D)
    v = value_cache[ids].permute(0, 3, 1, 2).reshape(len(ids) * bs, H, D)
    keep = merge_keep_sets(v, bs, k, method)
    for h in range(H):
        sel = torch.tensor(keep[h], dtype=torch.long)
        key_cache[ids[0], h] = k[sel, h].reshape(bs, Dx, x).permute(1, 0, 2)
        value_cache[ids[0], h] = v[sel, h].t()


def table_remove_range(tables: torch.Tensor, start: Sequence[int], n_remove: int,
                       out_width: int) -> torch.Tensor:
    """Rows without columns [start, start+n_remove) (start < 0: unchanged), zero-filled."""
    B, M = tables.shape
    out = torch.zeros((B, out_width), dtype=torch.int32)
    for b in range(B):
        row = tables[b].tolist()
        if start[b] >= 0:
            row = row[:start[b]] + row[start[b] + n_remove:]
        row = (row + [0] * out_width)[:out_width]
        out[b] = torch.tensor(row, dtype=torch.int32)
    return out


# --------------------------------------------------------------------------- block-table counters

class NaiveAllocator:
    """Hand-out order of NaiveBlockAllocator (vllm/core/block/naive_block.py:127-145): ids come
    off the front of a deque that starts as 0..n-1, a freed id returns to the front."""

    def __init__(self, num_blocks: int):
        self.free_ids = list(range(num_blocks))

    def take(self) -> int:
        return self.free_ids.pop(0)

    def give(self, block_id: int) -> None:
        self.free_ids.insert(0, block_id)


class BlockTableCounters:
    """The two counters the fork adds to BlockTable (vllm/core/block/block_table.py:
    58-66,119-126,188-196,456-466): ``num_full_slots`` (logical tokens) and
    ``num_cached_slots`` (tokens physically cached), plus the block-id list.  With
    ``allocator`` (a NaiveAllocator shared by several tables) the ids are the reference's."""

    def __init__(self, block_size: int, budget: Optional[int], allocator: Optional[NaiveAllocator] = None):
        self.bs = block_size
        self.budget = budget
        self.blocks: List[int] = []
        self.num_full_slots = 0
        self.num_cached_slots = 0
        self._next = 0
        self.freed: List[int] = []
        self.allocator = allocator

    def _new_block(self) -> int:
        if self.allocator is not None:
            return self.allocator.take()
        if self.freed:
            return self.freed.pop(0)
        b = self._next
        self._next += 1
        return b

    def free_all(self) -> None:
        """BlockTable.free, block_table.py:260-271: blocks are freed front to back."""
        for b in self.blocks:
            if self.allocator is not None:
                self.allocator.give(b)
        self.blocks = []

    def allocate(self, n_tokens: int) -> None:
        if self.budget is None:
            n_blocks, kept = -(-n_tokens // self.bs), n_tokens
        else:
            n_blocks, kept = num_required_blocks_after_prune(n_tokens, self.budget, self.bs)
        self.blocks = [self._new_block() for _ in range(n_blocks)]
        self.num_full_slots = n_tokens
        if self.budget is None:
            self.num_cached_slots = n_tokens
        else:
            rem = n_tokens % self.bs
            self.num_cached_slots = n_blocks * self.bs if rem == 0 else (n_blocks - 1) * self.bs + rem

    def append_token(self) -> None:
        if len(self.blocks) * self.bs - self.num_cached_slots < 1:
            self.blocks.append(self._new_block())
        self.num_full_slots += 1
        self.num_cached_slots += 1

    def free_pruned_block(self, idx: int) -> None:
        if self.allocator is not None:
            self.allocator.give(self.blocks[idx])
        else:
            self.freed.append(self.blocks[idx])
        self.num_cached_slots -= self.bs
        del self.blocks[idx]


def simulate_streaming_generation(prompt_len: int, out_len: int, budget: int, block_size: int):
    """Trace of (seq_len, kv_len passed to attention, n_blocks in table passed to
    attention, rmv_idx) per decode step for the streamingLLM method, composing
    block_table.py (allocate/append/free), model_runner.py:538-543 and
    torch_cuda_sdpa.py:1375-1433 + llm_engine.py:1179-1193."""
    bt = BlockTableCounters(block_size, budget)
    bt.allocate(prompt_len)
    trace = []
    for _ in range(out_len):
        bt.append_token()
        seq_len = bt.num_full_slots
        kv_len = bt.num_cached_slots
        rmv = -1
        if decode_trigger(seq_len, budget, block_size):
            kv_len -= block_size
            rmv = 1
        trace.append((seq_len, kv_len, len(bt.blocks) - (1 if rmv >= 0 else 0), rmv))
        if rmv >= 0:
            bt.free_pruned_block(rmv)
    return trace
