"""ctypes binding of libpaged_evict.so (the C ABI declared in include/paged_evict.h).

There is no fallback: if the shared library is missing or an entry point fails, the
caller gets an exception.  `python -m vllm_paged_compression_amd.build` builds it.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpaged_evict.so")

PE_F32, PE_F16, PE_BF16 = 0, 1, 2

# name -> (restype, argtypes); must list every symbol include/paged_evict.h declares
SIGNATURES = {
    "pe_last_error": (ctypes.c_char_p, []),
    "pe_version": (c_int, []),
    "pe_reload_env": (None, []),
    "pe_device_diagnostics": (c_int, [c_void_p, c_int]),
    "pe_reshape_and_cache": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int,
                                     c_int, c_int, c_int64, c_int, c_void_p]),
    "pe_paged_decode_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pe_paged_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                                c_int64, c_int64, c_float, c_int, c_int, c_void_p, c_int64,
                                c_void_p]),
    "pe_paged_decode_noalloc_max_bytes": (c_int64, []),
    "pe_paged_decode_noalloc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                                        c_int64, c_int64, c_float, c_int, c_int, c_void_p]),
    "pe_paged_decode_supports_fused_write": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pe_decode_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_float,
                               c_int, c_int, c_void_p, c_int64, c_void_p]),
    "pe_decode_step_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                    c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_int64,
                                    c_float, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "pe_prefill_causal": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                  c_int64, c_float, c_int, c_void_p]),
    "pe_prompt_prune_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "pe_prompt_keep_index_streaming": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_void_p]),
    "pe_prompt_token_select_workspace_bytes": (c_int64, [c_int64, c_int]),
    "pe_prompt_token_select": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                       c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "pe_block_score_value_l2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                        c_int, c_int, c_int, c_int, c_void_p]),
    "pe_block_token_norm_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                         c_int, c_int, c_int, c_int, c_void_p]),
    "pe_block_score_from_token_sums": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                               c_void_p]),
    "pe_evict_select_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_int,
                                        c_void_p]),
    "pe_table_state_advance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "pe_table_state_advance_ring": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                            c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_void_p]),
    "pe_evict_select_compact_inplace": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                                                c_void_p, c_int, c_void_p]),
    "pe_merge_blocks_value_l2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_int64, c_int64, c_int,
                                         c_void_p]),
    "pe_block_table_remove_range": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p]),
    "pe_merge_blocks_scored": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_int64, c_int64, c_int, c_int,
                                       c_void_p]),
    "pe_token_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int,
                                c_int, c_int, c_void_p]),
    "pe_rope_reshape_and_cache": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                          c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_int64,
                                          c_int, c_void_p]),
    "pe_ar_buffer_bytes": (c_int64, [c_int64]),
    "pe_ar_alloc": (c_int, [c_void_p, c_int64]),
    "pe_ar_free": (c_int, [c_void_p]),
    "pe_ar_ipc_handle_bytes": (c_int, []),
    "pe_ar_get_ipc_handle": (c_int, [c_void_p, c_void_p]),
    "pe_ar_open_ipc_handle": (c_int, [c_void_p, c_void_p]),
    "pe_ar_close_ipc_handle": (c_int, [c_void_p]),
    "pe_ar_timeouts": (c_int, [c_void_p, c_void_p]),
    "pe_ar_status_alloc": (c_int, [c_void_p]),
    "pe_ar_status_free": (c_int, [c_void_p]),
    "pe_ar_selfcheck": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pe_allreduce_oneshot": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int,
                                     c_int64, ctypes.c_uint, c_void_p, c_int64, c_void_p]),
    "pe_allreduce_twostage": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int,
                                      c_int64, ctypes.c_uint, c_void_p, c_int64, c_void_p]),
    "pe_zero_slots_layers": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int,
                                     c_int64, c_int, c_void_p]),
    "pe_copy_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                               c_void_p]),
    "pe_swap_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
}

_lib = None


class PagedEvictLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the shared library and bind every declared entry point (raises if any is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PagedEvictLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python -m vllm_paged_compression_amd.build`); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PagedEvictLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    """Non-zero status -> RuntimeError (the counterpart of TORCH_CHECK in the reference ops)."""
    if status != 0:
        msg = load().pe_last_error()
        raise RuntimeError(f"paged_evict {what} failed (status {status}): "
                           f"{msg.decode() if msg else ''}")
