"""PagedAttention static wrapper with the interface of the reference's
vllm/attention/ops/paged_attn.py:16-255, over the HIP operators."""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import ops


@dataclass
class PagedAttentionMetadata:
    """Metadata for PagedAttention."""
    seq_lens_tensor: Optional[torch.Tensor]   # (batch,) KV length per sequence
    max_decode_seq_len: int                   # 0 for a prefill-only batch
    block_tables: Optional[torch.Tensor]      # (batch, max_blocks_per_seq) int32


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [32, 64, 80, 96, 112, 120, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                             kv_cache_dtype: str, k_scale: float, v_scale: float, *,
                             src_index: Optional[torch.Tensor] = None, src_pairs: bool = False,
                             zero_slots: Optional[torch.Tensor] = None) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(),
                              kv_cache_dtype, k_scale, v_scale, src_index=src_index,
                              src_pairs=src_pairs, zero_slots=zero_slots)

    @staticmethod
    def forward_decode(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int,
                       kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], k_scale: float, v_scale: float,
                       tp_rank: int = 0, blocksparse_local_blocks: int = 0,
                       blocksparse_vert_stride: int = 0, blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0,
                       output: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One kernel family replaces the reference's v1 / v2+reduce pair; the split-KV
        partitioning that its heuristic (paged_attn.py:117-127) chooses between is decided
        inside the library from the batch shape."""
        if output is None:
            output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                               block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                               kv_cache_dtype, k_scale, v_scale, tp_rank, blocksparse_local_blocks,
                               blocksparse_vert_stride, blocksparse_block_size,
                               blocksparse_head_sliding_step)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        key_caches = [kv_cache[0] for kv_cache in kv_caches]
        value_caches = [kv_cache[1] for kv_cache in kv_caches]
        ops.copy_blocks(key_caches, value_caches, src_to_dists)
