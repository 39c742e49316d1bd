"""Prompt length after pruning, for block allocation, and the blocks a decode prune frees.
Same names and return values as the reference's vllm/core/page_evict_kv_util.py:6-36 (budget
prune type) and vllm/core/page_evict_kv_util_bak.py:6-106 (percentage prune type)."""
from typing import Tuple


def cdiv(a: int, b: int) -> int:
    return -(a // -b)


def percentage_pruned_prompt(q_len: int, block_size: int, evict_freq: int, initial_blocks: int,
                             num_block_merge: int) -> Tuple[int, int]:
    """(tokens pruned, tokens kept) of a q_len-token prompt under the percentage prune type
    (page_evict_kv_util_bak.py:10-31, kvcache_prunner_old.py:143-176): the first
    initial_blocks blocks and the last num_block_merge blocks (+ the remainder) stay, and of
    the middle the oldest block_size*floor(int(middle/evict_freq)/block_size) tokens go."""
    if q_len <= (initial_blocks + num_block_merge) * block_size:
        return 0, q_len
    rem = q_len % block_size
    middle = (q_len - num_block_merge * block_size - rem) - initial_blocks * block_size
    pruned = block_size * (int(middle / evict_freq) // block_size)
    return pruned, q_len - pruned


def get_num_required_blocks_after_prune_promt(q_len: int, paged_evict_config,
                                              block_size: int) -> Tuple[int, int]:
    """(number of blocks, number of tokens) a q_len-token prompt occupies once pruned.
    Budget: first block + budgeted middle + last block (+ remainder)."""
    cfg = paged_evict_config
    if cfg.cache_prune_type == "percentage":
        _, kept = percentage_pruned_prompt(q_len, block_size, cfg.evict_freq, cfg.initial_blocks,
                                           cfg.num_block_merge)
        return cdiv(kept, block_size), kept
    if q_len < cfg.cache_budget:
        return cdiv(q_len, block_size), q_len
    if cfg.evict_method not in ("streamingLLM", "streamingLLM-1", "inverse_key_l2", "global",
                                "local"):
        raise ValueError(f"Unsupported eviction_method: {cfg.evict_method} for cache_prune_type: "
                         f"{cfg.cache_prune_type}")
    if cfg.cache_prune_type != "budget":
        raise ValueError(f"Unsupported cache_prune_type: {cfg.cache_prune_type} for evict_method: "
                         f"{cfg.evict_method}")
    head = cfg.initial_blocks * block_size
    end_middle = q_len - block_size - q_len % block_size
    middle_kept = max(cfg.cache_budget - head - block_size, 0)
    # never more than the prompt holds (a budget that is not a whole number of blocks, which
    # PagedEvictConfig rounds away, would otherwise overshoot for prompts just above it)
    total = min(q_len, head + middle_kept + (q_len - end_middle))
    return cdiv(total, block_size), total


def get_blocks_to_prune(paged_evict_config, seq_kv_len: int, block_size: int) -> Tuple[int, int, int]:
    """(first block index, end block index, tokens pruned) of the block-table columns a decode
    prune of the percentage type frees, (-1, -1, 0) for a sequence too short to prune
    (page_evict_kv_util_bak.py:66-94; there the short-sequence result is overwritten by the
    lines that follow it - the early return is what those lines are for).
    streamingLLM frees [initial_blocks, initial_blocks + num_block_merge - 1); value_l2 merges
    num_block_merge blocks into block initial_blocks and frees the num_block_merge - 1 after it."""
    cfg = paged_evict_config
    if cfg.cache_prune_type != "percentage":
        raise ValueError("get_blocks_to_prune describes the percentage prune type")
    ib, nbm = cfg.initial_blocks, cfg.num_block_merge
    if seq_kv_len <= (ib + nbm) * block_size:
        return -1, -1, 0
    if cfg.decode_evict_method == "streamingLLM":
        return ib, ib + nbm - 1, (nbm - 1) * block_size
    return ib + 1, ib + nbm, (nbm - 1) * block_size


def percentage_decode_trigger(paged_evict_config, seq_len: int, seq_kv_len: int,
                              block_size: int) -> bool:
    """A decode step of the percentage type prunes when the logical length (the step's token
    included) reaches a multiple of evict_freq blocks (torch_cuda_sdpa.py:1310) and the
    sequence physically holds more than initial_blocks + num_block_merge blocks."""
    cfg = paged_evict_config
    return (seq_len % (block_size * cfg.evict_freq) == 0
            and seq_kv_len > (cfg.initial_blocks + cfg.num_block_merge) * block_size)


def budget_decode_trigger(paged_evict_config, seq_len: int, block_size: int) -> bool:
    """A decode step of the budget type evicts one block when the logical length (the step's
    token included) exceeds the budget and sits on a block boundary
    (vllm/worker/model_runner.py:540-541, torch_cuda_sdpa.py:1390-1403,1491-1498).
    inverse_key_l2 never removes a block at decode (model_runner.py:545-547)."""
    cfg = paged_evict_config
    return (cfg.evict_method in ("streamingLLM", "streamingLLM-1", "global", "local")
            and seq_len > cfg.cache_budget and seq_len % block_size == 0)


def evicted_tokens(paged_evict_config, seq_len: int, seq_kv_len: int, block_size: int) -> int:
    """Tokens that leave the attention view at the decode step whose logical length is seq_len
    and whose physical length (before the step's eviction) is seq_kv_len: what the model runner
    subtracts from seq_kv_lens before attention runs (model_runner.py:538-543).  0 with eviction
    off (config None)."""
    cfg = paged_evict_config
    if cfg is None:
        return 0
    if cfg.cache_prune_type == "percentage":
        return ((cfg.num_block_merge - 1) * block_size
                if percentage_decode_trigger(cfg, seq_len, seq_kv_len, block_size) else 0)
    return block_size if budget_decode_trigger(cfg, seq_len, block_size) else 0


def adjust_seq_kv_len(paged_evict_config, seq_len: int, seq_kv_len: int, block_size: int) -> int:
    """seq_kv_len as the attention kernels must see it on this decode step."""
    return seq_kv_len - evicted_tokens(paged_evict_config, seq_len, seq_kv_len, block_size)


def adjust_seq_kv_lens(paged_evict_config, seq_lens, seq_kv_lens, block_size: int):
    """Array form of adjust_seq_kv_len for a decode batch (numpy int arrays in, int64 array out)."""
    import numpy as np
    L = np.asarray(seq_lens, dtype=np.int64)
    kv = np.asarray(seq_kv_lens, dtype=np.int64)
    cfg = paged_evict_config
    if cfg is None:
        return kv.copy()
    if cfg.cache_prune_type == "percentage":
        trig = (L % (block_size * cfg.evict_freq) == 0) & \
               (kv > (cfg.initial_blocks + cfg.num_block_merge) * block_size)
        return kv - trig * ((cfg.num_block_merge - 1) * block_size)
    if cfg.evict_method not in ("streamingLLM", "streamingLLM-1", "global", "local"):
        return kv.copy()
    trig = (L > cfg.cache_budget) & (L % block_size == 0)
    return kv - trig * block_size
