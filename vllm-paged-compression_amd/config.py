"""Eviction flags: the validated config object of the path.

``PagedEvictConfig`` has the constructor arguments, attributes and validation of the
reference's vllm/config.py:933-991 (flags: vllm/engine/arg_utils.py:971-1036), so an
instance built by the fork's ``EngineArgs`` can be used as is, and vice versa.  Extra,
build-specific switches live in ``CompatFlags`` (all default to the intended semantics).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Optional

logger = logging.getLogger(__name__)

EVICT_METHODS = ("streamingLLM", "streamingLLM-1", "inverse_key_l2", "local", "global")
# methods of the "percentage" prune type (README-era --prompt-evict-method / --decode-evict-method)
# token scores get_score of the README-era pruner knows (vllm/attention/kvcache_prunner_old.py:77-134)
TOKEN_SCORE_METHODS = ("cosine", "key_l1", "key_l2", "value_l1", "value_l2", "key_l1_div_value_l1",
                       "key_l1_div_value_l2", "key_l2_div_value_l1", "key_l2_div_value_l2",
                       "value_l1_div_key_l1", "value_l1_div_key_l2", "value_l2_div_key_l1",
                       "value_l2_div_key_l2", "value_l1_plus_key_l1", "value_l2_plus_key_l2",
                       "inverse_key_l1", "inverse_key_l2")
PERCENTAGE_PROMPT_METHODS = ("streamingLLM",) + TOKEN_SCORE_METHODS
PERCENTAGE_DECODE_METHODS = ("streamingLLM",) + TOKEN_SCORE_METHODS


class PagedEvictConfig:
    """--enable-paged-eviction flags.

    disable_evict_prefill: attend the full prompt, prune when writing the cache
    cache_prune_type:      "budget", or "percentage" when ``evict_freq`` is given (the reference's
                           HEAD parses "percentage" and then rejects it: the five positional
                           arguments alone behave exactly like that)
    evict_method:          one of EVICT_METHODS
    cache_budget:          tokens kept per sequence (clamped up to 128)
    topk_blocks:           candidate window of "local" (> 1); forced to -1 for "global"

    Keyword-only, the flags HEAD's constructor has commented out (config.py:945-958,
    engine/arg_utils.py:1004-1006,1138-1141) and the README's command lines still pass:
    evict_freq:            "percentage": the prompt prune drops 1/evict_freq of the middle tokens
                           (whole blocks) and a decode step prunes every evict_freq blocks
    num_block_merge:       blocks a decode prune works on: num_block_merge-1 of them are freed
    prompt_evict_method / decode_evict_method: the two method flags of that era
                           ("streamingLLM" or a token score of TOKEN_SCORE_METHODS, the
                           names get_score of that era's pruner knows); default evict_method
    """

    def __init__(self, disable_evict_prefill: bool, cache_prune_type: str, evict_method: str,
                 cache_budget: int, topk_blocks: int, *, evict_freq: Optional[int] = None,
                 num_block_merge: int = 2, prompt_evict_method: Optional[str] = None,
                 decode_evict_method: Optional[str] = None) -> None:
        self.disable_evict_prefill = disable_evict_prefill
        self.cache_prune_type = cache_prune_type
        self.evict_method = evict_method
        self.cache_budget = cache_budget
        self.topk_blocks = topk_blocks
        self.initial_blocks = 1  # blocks at the head of the sequence that are never evicted
        self.evict_freq = evict_freq
        self.num_block_merge = num_block_merge
        self.prompt_evict_method = prompt_evict_method or evict_method
        self.decode_evict_method = decode_evict_method or evict_method
        self._verify_args()

    def _verify_args(self) -> None:
        if self.cache_prune_type not in ("percentage", "budget"):
            raise ValueError("Cache prune type must be either 'percentage' or 'budget'. Got "
                             f"{self.cache_prune_type}.")
        if self.cache_prune_type == "percentage" and self.evict_freq is not None:
            if self.evict_freq < 1:
                raise ValueError(f"Evict frequency must be positive. Got {self.evict_freq}.")
            if self.num_block_merge < 2:
                raise ValueError("Number of blocks to merge must be at least 2. Got "
                                 f"{self.num_block_merge}.")
            if self.prompt_evict_method not in PERCENTAGE_PROMPT_METHODS:
                raise ValueError(f"Unknown prompt evict method {self.prompt_evict_method!r} for the "
                                 f"percentage prune type; expected one of {PERCENTAGE_PROMPT_METHODS}.")
            if self.decode_evict_method not in PERCENTAGE_DECODE_METHODS:
                raise ValueError(f"Unknown decode evict method {self.decode_evict_method!r} for the "
                                 f"percentage prune type; expected one of {PERCENTAGE_DECODE_METHODS}.")
            return
        assert self.cache_prune_type == "budget", \
            "Cache prune type must be 'budget' for paged compression now."
        if self.evict_method not in EVICT_METHODS:
            raise ValueError(f"Unknown evict method {self.evict_method!r}; expected one of "
                             f"{EVICT_METHODS}.")
        if self.cache_budget < 0:
            raise ValueError(f"Cache budget must be positive. Got {self.cache_budget}.")
        if self.cache_budget < 128:
            logger.warning("Cache budget is set to less than 128 tokens. "
                           "Setting cache budget to 128 tokens.")
            self.cache_budget = 128
        if self.cache_budget % 16 != 0:
            # every block size the backend runs (8, 16, 32 in the reference's scripts) divides a
            # multiple of 16 only if it is <= 16; KVCachePruner re-checks against the real block
            # size.  A budget that is not a whole number of blocks makes the pruned-length formula
            # return more tokens than a prompt just above the budget has
            # (vllm/attention/kvcache_prunner.py:284-326; the reference then fails in copy_)
            rounded = self.cache_budget // 16 * 16
            logger.warning("Cache budget %d is not a multiple of 16; using %d.", self.cache_budget,
                           rounded)
            self.cache_budget = rounded
        if self.evict_method == "local" and self.topk_blocks <= 1:
            raise ValueError("Topk blocks must be > 1 when using 'local' evict method. "
                             f"Got {self.topk_blocks}.")
        if self.evict_method == "global":
            self.topk_blocks = -1

    def __repr__(self) -> str:
        extra = ""
        if self.cache_prune_type == "percentage":
            extra = (f", evict_freq={self.evict_freq}, num_block_merge={self.num_block_merge}, "
                     f"prompt_evict_method={self.prompt_evict_method!r}, "
                     f"decode_evict_method={self.decode_evict_method!r}")
        return (f"PagedEvictConfig(disable_evict_prefill={self.disable_evict_prefill}, "
                f"cache_prune_type={self.cache_prune_type!r}, evict_method={self.evict_method!r}, "
                f"cache_budget={self.cache_budget}, topk_blocks={self.topk_blocks}{extra})")


def budget_from_percentage(evict_fraction: float, input_len: int, block_size: int) -> int:
    """Budget that keeps (1 - evict_fraction) of an input_len-token prompt, rounded down to a
    whole number of blocks (the benchmark configs are worded as "25 % / 50 % eviction")."""
    if not 0.0 <= evict_fraction < 1.0:
        raise ValueError("evict_fraction must be in [0, 1)")
    return max(int((1.0 - evict_fraction) * input_len) // block_size * block_size, 3 * block_size)


@dataclass
class CompatFlags:
    """Switches that reproduce behaviours of the reference's HEAD that differ from the
    intended algorithm.  Defaults = intended semantics."""
    # in-loop slice of _create_updated_block_tables (torch_cuda_sdpa.py:1583)
    block_table_slice_in_loop: bool = False
    # middle[mask].view(-1, H, D) re-packing across heads (kvcache_prunner.py:251-252)
    prompt_prune_cross_head: bool = False
    # decode-time block score read through a flat re-view of the cache block (:1521-1524)
    score_flat_cache_view: bool = False
    # never delete the victim's score (:1546 commented out), so the list outgrows the table
    keep_victim_score: bool = False
    # random victim instead of arg-min (enable_random_evict = True, :802)
    random_victim: bool = False
    # every layer takes its own decision; only layer 0's is committed (model_runner.py:1797-1802)
    per_layer_decision: bool = False
    # batch index -> request through BlockL2NormManager's flat map, which lists requests in the
    # order they were first seen and forgets every request absent from a step - a prefill step
    # of new prompts drops all running ones (block_l2norm_manager.py:15-61); the default maps
    # index i to the i-th sequence of the step's own metadata and keeps scores until a request
    # is absent from a decode step
    request_order_first_seen: bool = False
    # streamingLLM / streamingLLM-1 decode at HEAD computes the compacted table but never uses it:
    # the flag it sets (is_pruned_decode, torch_cuda_sdpa.py:903-911) is not the one the attention
    # call tests (is_prune_decode, :858,1046), so attention reads the ORIGINAL table with the kv
    # length shortened by one block - the victim block stays visible and the last block (with
    # the step's own token) is cut off.  The default attends the compacted table.
    streaming_decode_original_table: bool = False
    # the compacted table HEAD hands to paged attention is the slice [:, :M-1] of an [B, M] tensor
    # (torch_cuda_sdpa.py:1583), i.e. a view whose rows are M entries apart, and the attention
    # kernels index it with size(1) = M-1 as the row pitch (csrc/attention/paged_attention_v1.cu:63,81,
    # attention_kernels.cuh:207): row i is read from flat offset i*(M-1) of the underlying buffer,
    # so on an eviction step every sequence but the first attends a table shifted by i entries
    # that starts with other rows' block ids.  The default hands the kernel a dense table.
    compacted_table_strided_view: bool = False

    @classmethod
    def reference_head(cls) -> "CompatFlags":
        return cls(True, True, True, True, True, True, True, True, True)


@dataclass
class CacheConfigView:
    """The two attributes of vLLM's CacheConfig this path reads."""
    block_size: int
    paged_evict_config: Optional[PagedEvictConfig] = None
