"""MI355X-native paged-KV-eviction attention path (gfx950 HIP kernels behind a C ABI).

Layout
  csrc/                 HIP kernels + the extern "C" entry points (include/paged_evict.h)
  _lib.py               ctypes binding; no fallback when the shared library is missing
  ops.py                torch-level operators mirroring the reference's vllm._custom_ops
  paged_attn.py         PagedAttention static wrapper (mirror of vllm/attention/ops/paged_attn.py)
  config.py             PagedEvictConfig (mirror of vllm/config.py:933-991)
  pruner.py             KVCachePruner host logic over the HIP ops
  block_l2norm_manager.py   per-request block-score bookkeeping
  block_table.py        eviction-aware block table / block manager counters
  backend.py            PagedEvictBackend: AttentionBackend / Impl / Metadata / Builder
"""
from . import _lib  # noqa: F401
from ._lib import LIB_PATH, PagedEvictLibraryError, load as load_library  # noqa: F401

__all__ = ["load_library", "LIB_PATH", "PagedEvictLibraryError"]
__version__ = "0.1.0"
