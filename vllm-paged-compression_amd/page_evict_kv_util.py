"""Prompt length after pruning, for block allocation.  Same name and return value as the
reference's vllm/core/page_evict_kv_util.py:6-36."""
from typing import Tuple


def cdiv(a: int, b: int) -> int:
    return -(a // -b)


def get_num_required_blocks_after_prune_promt(q_len: int, paged_evict_config,
                                              block_size: int) -> Tuple[int, int]:
    """(number of blocks, number of tokens) a q_len-token prompt occupies once pruned to
    the cache budget: first block + budgeted middle + last block (+ remainder)."""
    cfg = paged_evict_config
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
    total = head + middle_kept + (q_len - end_middle)
    return cdiv(total, block_size), total
