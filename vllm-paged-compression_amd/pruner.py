"""KVCachePruner: prompt pruning and block scoring, with the constructor and method names of
the reference's vllm/attention/kvcache_prunner.py:6-326, over the HIP operators.  The
"percentage" prune type (README-era flags; vllm/attention/kvcache_prunner_old.py:137-315) is
the same object with ``evict_freq`` / ``num_block_merge`` given.

Length arithmetic (``get_pruned_length``) runs on the host; everything that touches K/V
runs on the GPU: the kept rows are described by an int32 index built on the device
(streamingLLM: a closed-form map; local/global/inverse_key_l2: per-head radix select of the
lowest scores) and applied either as a dense gather (``prune_prompt``) or fused into the
KV-cache write (``ops.reshape_and_cache(..., src_index=...)``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import ops
from .page_evict_kv_util import percentage_pruned_prompt

_STREAMING = ("streamingLLM", "streamingLLM-1")
_TOKEN_LEVEL = ("local", "global", "inverse_key_l2")


class KVCachePruner:

    def __init__(self, cache_prune_type: str, evict_method: str, block_size: int,
                 cache_budget: int, initial_blocks: int, *, cross_head_compat: bool = False,
                 evict_freq: Optional[int] = None, num_block_merge: int = 2,
                 decode_evict_method: Optional[str] = None):
        self.cache_prune_type = cache_prune_type
        self.evict_method = evict_method
        self.orig_block_size = block_size
        self.cache_budget = cache_budget
        # score used by "global" and "local" (kvcache_prunner.py:13 hard-wires "value_l2"; the
        # other formulas of :29-66 are reached by assigning the attribute, as in the reference)
        self.sub_evict_method = "value_l2"
        self.initial_blocks = initial_blocks
        self.cross_head_compat = cross_head_compat
        self.evict_frequency = evict_freq
        self.num_block_merge = num_block_merge
        self.prompt_evict_method = evict_method
        self.decode_evict_method = decode_evict_method or evict_method
        if cache_prune_type == "percentage":
            assert evict_freq is not None and evict_freq >= 1 and num_block_merge >= 2
            assert initial_blocks == 1, "the kept-row index jumps after the first block"
            if evict_method not in ops.TOKEN_SCORE_FORMULAS:
                raise ValueError(f"Unknown prompt eviction method for the percentage type: {evict_method}")
            if self.decode_evict_method not in ops.TOKEN_SCORE_FORMULAS:
                raise ValueError("Unknown decode eviction method for the percentage type: "
                                 f"{self.decode_evict_method}")
            return
        assert cache_prune_type == "budget"
        assert cache_budget >= 3 * block_size, \
            "Cache budget must be at least 3 times the original block size"
        if cache_budget % block_size != 0:
            # floor(L/bs)*bs < budget < L would make get_pruned_length exceed L: the kept-row
            # index would then address rows past the prompt
            raise ValueError(f"cache_budget {cache_budget} must be a multiple of the block size "
                             f"{block_size}")
        if evict_method not in _STREAMING + _TOKEN_LEVEL:
            raise ValueError(f"Unknown eviction method: {evict_method}")

    # ------------------------------------------------------------------ host arithmetic
    def get_pruned_length(self, seq_len: int) -> int:
        """Tokens kept of a seq_len-token prompt.  Budget: first block + budgeted middle + last
        block (+ remainder).  Percentage: everything but the oldest 1/evict_freq of the middle
        (whole blocks)."""
        if self.cache_prune_type == "percentage":
            return percentage_pruned_prompt(seq_len, self.orig_block_size, self.evict_frequency,
                                            self.initial_blocks, self.num_block_merge)[1]
        if seq_len <= self.cache_budget:
            return seq_len
        bs = self.orig_block_size
        end_middle = seq_len - bs - seq_len % bs
        return min(seq_len, bs + max(self.cache_budget - 2 * bs, 0) + (seq_len - end_middle))

    def prunes(self, seq_len: int) -> bool:
        """Whether a prompt of this length loses tokens."""
        if self.cache_prune_type == "percentage":
            return self.get_pruned_length(seq_len) < seq_len
        return seq_len > self.cache_budget

    def get_blocks_to_prune_and_merge_decode(self, seq_kv_len: int) -> Tuple[int, int, int]:
        """(first, end, tokens pruned) of the block-table columns a decode prune of the
        percentage type works on: num_block_merge blocks from initial_blocks, of which
        num_block_merge-1 blocks' worth of tokens go (kvcache_prunner_old.py:264-295)."""
        assert self.cache_prune_type == "percentage"
        s = self.initial_blocks
        return s, s + self.num_block_merge, (self.num_block_merge - 1) * self.orig_block_size

    def merge_blocks_in_cache(self, key_cache: torch.Tensor, value_cache: torch.Tensor,
                              block_tables: torch.Tensor, rows: torch.Tensor) -> None:
        """Scored decode prune (value_l2 or any other token score of get_score) for the batch rows
        listed in ``rows``: per KV head the block_size highest-score tokens of the merged blocks
        survive, in order, in the first of them (prune_oldest_block + write-back,
        kvcache_prunner_old.py:297-315 and torch_cuda_sdpa.py:1331-1363, in place on the paged
        cache)."""
        s, _, _ = self.get_blocks_to_prune_and_merge_decode(0)
        ops.merge_blocks_value_l2(key_cache, value_cache, block_tables, rows, s, self.num_block_merge,
                                  formula=self.decode_evict_method)

    def get_score(self, block_keys: torch.Tensor, block_values: torch.Tensor, evict_method: str):
        """[T, H] token scores of dense [T, H, D] rows under any method name of the README-era
        class (get_score, kvcache_prunner_old.py:77-134)."""
        return ops.token_scores(block_keys.contiguous(), block_values.contiguous(), evict_method)

    # ------------------------------------------------------------------ device index
    def keep_index(self, key: torch.Tensor, value: torch.Tensor, seq_lens: Sequence[int],
                   cu_q: Optional[torch.Tensor] = None,
                   cu_kept: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, List[int]]:
        """Kept-row index for a batch of prompts laid out back to back in key/value
        ([sum(seq_lens), H, D]).  Returns (int32 index, kept length per prompt); the index is
        [n_kept] for the streamingLLM methods and [n_kept, H] (per head) otherwise."""
        kept = [self.get_pruned_length(L) for L in seq_lens]
        dev = key.device
        if cu_q is None:
            cu_q = _cumsum(seq_lens, dev)
        if cu_kept is None:
            cu_kept = _cumsum(kept, dev)
        total = sum(kept)
        if self.cache_prune_type == "percentage" and self.evict_method != "streamingLLM":
            # token-level scores of that era (kvcache_prunner_old.py:201-262): per head the
            # lowest-score middle tokens go, under the score the method names
            idx = ops.prompt_token_select(key, value, cu_q, cu_kept, total, 0, self.orig_block_size,
                                          "global", compat_q2=self.cross_head_compat,
                                          tail_blocks=self.num_block_merge, formula=self.evict_method)
        elif self.cache_prune_type == "percentage":
            # budget 0: the pruned count of each prompt is its length minus its kept length
            idx = ops.prompt_keep_index_streaming(cu_q, cu_kept, total, 0, self.orig_block_size)
        elif self.evict_method in _STREAMING:
            idx = ops.prompt_keep_index_streaming(cu_q, cu_kept, total, self.cache_budget,
                                                  self.orig_block_size)
        else:
            idx = ops.prompt_token_select(key, value, cu_q, cu_kept, total, self.cache_budget,
                                          self.orig_block_size, self.evict_method,
                                          compat_q2=self.cross_head_compat,
                                          formula=self._formula())
        return idx, kept

    def _gather(self, key, value, idx):
        mode = None
        if idx.dim() == 2 and self.cross_head_compat:
            mode = 2
        return ops.prompt_prune_gather(key, value, idx, mode=mode)

    # ------------------------------------------------------------------ reference-shaped API
    def prune_prompt(self, key_tensor: torch.Tensor, value_tensor: torch.Tensor):
        """Prune one request's prompt K/V ([L, H, D]) -> ([n_kept, H, D], same)."""
        L = key_tensor.shape[0]
        if not self.prunes(L):
            return key_tensor, value_tensor
        idx, _ = self.keep_index(key_tensor, value_tensor, [L])
        return self._gather(key_tensor, value_tensor, idx)

    def prune_prompt_inplace(self, key_tensor, value_tensor, output_key_tensor,
                             output_value_tensor) -> None:
        """Same, writing into caller-provided [n_kept, H, D] tensors."""
        k, v = self.prune_prompt(key_tensor, value_tensor)
        output_key_tensor.copy_(k)
        output_value_tensor.copy_(v)

    def prune_prompts_batch(self, key: torch.Tensor, value: torch.Tensor,
                            seq_lens: Sequence[int]):
        """All prompts of a prefill batch in one go (the reference loops over requests on 8
        side streams, torch_cuda_sdpa.py:1222-1296).  Returns (keys, values, kept lengths)."""
        if not any(self.prunes(L) for L in seq_lens):
            return key, value, list(seq_lens)
        idx, kept = self.keep_index(key, value, seq_lens)
        k, v = self._gather(key, value, idx)
        return k, v, kept

    def _formula(self) -> str:
        if self.sub_evict_method not in ops.SCORE_FORMULAS:
            raise ValueError(f"Unknown sub eviction method: {self.sub_evict_method}")
        return self.sub_evict_method

    def get_token_score(self, keys: torch.Tensor, values: torch.Tensor, evict_method: str):
        """[T, H] scores (kvcache_prunner.py:29-49): the sub method's formula for global/local
        (||V||_2 by default), 1/(||K||_2+1e-8) for inverse_key_l2."""
        if evict_method in ("global", "local"):
            f = self._formula()
            if f == "value_l2":
                return torch.norm(values, p=2, dim=-1)
            if f == "key_l2":
                return torch.norm(keys, p=2, dim=-1)
            if f == "value_l2_plus_key_l2":
                return torch.norm(keys, p=2, dim=-1) + torch.norm(values, p=2, dim=-1)
            return torch.div(1, torch.norm(keys, p=2, dim=-1) + 1e-8)
        if evict_method == "inverse_key_l2":
            return torch.div(1, torch.norm(keys, p=2, dim=-1) + 1e-8)
        raise ValueError(f"Unknown eviction method: {evict_method}")

    def get_block_score(self, block_keys: torch.Tensor, block_values: torch.Tensor,
                        evict_method: str) -> torch.Tensor:
        """Score of one [block_size, H, D] block under the sub method (0-d tensor of the input
        dtype; kvcache_prunner.py:51-66)."""
        assert evict_method in ("global", "local"), \
            "Only global and local eviction methods are supported for block score calculation"
        bs = block_values.shape[0]
        s = self.get_block_scores_dense(block_values, 1, keys=block_keys, block_size=bs)
        return s[0].to(block_values.dtype)

    def get_block_scores_dense(self, values: torch.Tensor, n_blocks: int,
                               out: Optional[torch.Tensor] = None,
                               keys: Optional[torch.Tensor] = None,
                               block_size: Optional[int] = None) -> torch.Tensor:
        """Scores of the first n_blocks whole blocks of dense [T, H, D] value (and key) tensors
        (the prompt-time loop of torch_cuda_sdpa.py:1278-1288 as one launch per tensor); f32
        [n_blocks].  ``keys`` is needed by every formula but value_l2."""
        T, H, D = values.shape
        bs = block_size or self.orig_block_size
        f = self._formula()
        if f == "value_l2":
            return ops.block_score_value_l2(values.contiguous(), None, n_blocks, ops.SCORE_TOKEN_MAJOR,
                                            H, D, bs, out=out)
        if keys is None:
            raise ValueError(f"score formula {f} needs the keys")
        ks = ops.block_score_value_l2(keys.contiguous(), None, n_blocks, ops.SCORE_TOKEN_MAJOR, H, D, bs)
        vs = None
        if f == "value_l2_plus_key_l2":
            vs = ops.block_score_value_l2(values.contiguous(), None, n_blocks, ops.SCORE_TOKEN_MAJOR,
                                          H, D, bs)
        sc = ops.combine_block_scores(f, vs, ks, values.dtype)
        if out is not None:
            out.copy_(sc)
            return out
        return sc


def _cumsum(lens: Sequence[int], device) -> torch.Tensor:
    out = [0]
    for n in lens:
        out.append(out[-1] + int(n))
    return torch.tensor(out, dtype=torch.int32, device=device)
