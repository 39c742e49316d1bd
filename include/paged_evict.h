/*
 * paged_evict.h — C ABI of libpaged_evict.so: MI355X (gfx950) kernels for the
 * paged-KV-eviction attention path of vllm-Paged-Compression.
 *
 * Conventions
 *  - every pointer named *_dev / documented "device" is a device (HBM) pointer owned
 *    by the caller; the library never allocates or frees memory in these calls (the set-up
 *    calls of the one-shot all-reduce, pe_ar_alloc / pe_ar_free / pe_ar_*_ipc_handle, are the
 *    exception and say so);
 *  - `stream` is a hipStream_t (NULL = the null stream); all work is enqueued on it
 *    and the call returns without synchronising;
 *  - return value: PE_OK (0) or a PE_ERR_* code; pe_last_error() then returns a
 *    thread-local message.  The Python side turns a non-zero status into
 *    RuntimeError, mirroring TORCH_CHECK in the reference's ops;
 *  - `dtype` is a pe_dtype; K/V cache element type == query/key/value element type
 *    (kv_cache_dtype "auto"; the reference backend rejects fp8 KV,
 *    vllm/attention/backends/torch_cuda_sdpa.py:769-772);
 *  - paged cache layouts are the reference's (vllm/attention/ops/paged_attn.py:40-62,
 *    csrc/cache_kernels.cu:184-191):
 *        key_cache   [num_blocks, num_kv_heads, head_size/x, block_size, x], x = 16/sizeof(T)
 *        value_cache [num_blocks, num_kv_heads, head_size, block_size]
 *
 * Reference citations are relative to the reference checkout.
 */
#ifndef PAGED_EVICT_H_
#define PAGED_EVICT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { PE_F32 = 0, PE_F16 = 1, PE_BF16 = 2 } pe_dtype;

enum {
  PE_OK = 0,
  PE_ERR_INVALID_ARG = 1, /* shape / pointer / range check failed               */
  PE_ERR_UNSUPPORTED = 2, /* valid request the kernels do not implement         */
  PE_ERR_LAUNCH = 3,      /* HIP reported an error at launch                    */
  PE_ERR_WORKSPACE = 4    /* caller's workspace is too small                    */
};

/* eviction methods of PagedEvictConfig.evict_method (vllm/config.py:933-991,
 * vllm/engine/arg_utils.py:971-1036) */
typedef enum {
  PE_EVICT_STREAMING = 0,    /* "streamingLLM"   */
  PE_EVICT_STREAMING_1 = 1,  /* "streamingLLM-1" */
  PE_EVICT_LOCAL = 2,        /* "local"          */
  PE_EVICT_GLOBAL = 3,       /* "global"         */
  PE_EVICT_INVERSE_KEY_L2 = 4 /* "inverse_key_l2" */
} pe_evict_method;

/* score formula of "local" / "global": KVCachePruner.sub_evict_method
 * (kvcache_prunner.py:13,29-66; HEAD hard-wires "value_l2").  Carried in bits 8..15 of the
 * `method` argument of pe_prompt_token_select: method | (formula << PE_SCORE_FORMULA_SHIFT).
 * From PE_FORMULA_KEY_L1 on: the token scores of the README-era class (get_score,
 * vllm/attention/kvcache_prunner_old.py:23-134), which the percentage prune type's prompt prune
 * and decode-time block merge select by method name.  They are implemented as that file WRITES
 * them, which for some is not what the name says (noted per entry). */
typedef enum {
  PE_FORMULA_VALUE_L2 = 0,             /* ||V||_2                      */
  PE_FORMULA_KEY_L2 = 1,               /* ||K||_2                      */
  PE_FORMULA_VALUE_L2_PLUS_KEY_L2 = 2, /* ||V||_2 + ||K||_2            */
  PE_FORMULA_INVERSE_KEY_L2 = 3,       /* 1 / (||K||_2 + 1e-8)         */
  PE_FORMULA_KEY_L1 = 4,               /* ||K||_1                      */
  PE_FORMULA_VALUE_L1 = 5,             /* ||V||_1                      */
  PE_FORMULA_INVERSE_KEY_L1 = 6,       /* 1 / (||K||_1 + 1e-8)         */
  PE_FORMULA_KEY_L1_DIV_VALUE_L1 = 7,  /* as written (:47-48): ||K||_2 / (||V||_1 + 1e-8) */
  PE_FORMULA_KEY_L1_DIV_VALUE_L2 = 8,  /* ||K||_1 / (||V||_2 + 1e-8)   */
  PE_FORMULA_KEY_L2_DIV_VALUE_L1 = 9,  /* ||K||_2 / (||V||_1 + 1e-8)   */
  PE_FORMULA_KEY_L2_DIV_VALUE_L2 = 10, /* ||K||_2 / (||V||_2 + 1e-8)   */
  PE_FORMULA_VALUE_L1_DIV_KEY_L1 = 11, /* as written (:59-60): ||K||_1 / (||V||_1 + 1e-8) */
  PE_FORMULA_VALUE_L1_DIV_KEY_L2 = 12, /* as written (:62-63): ||K||_1 / (||V||_2 + 1e-8) */
  PE_FORMULA_VALUE_L2_DIV_KEY_L1 = 13, /* as written (:65-66): ||K||_2 / (||V||_1 + 1e-8) */
  PE_FORMULA_VALUE_L2_DIV_KEY_L2 = 14, /* as written (:68-69): ||K||_2 / (||V||_2 + 1e-8) */
  PE_FORMULA_VALUE_L1_PLUS_KEY_L1 = 15, /* ||K||_1 + ||V||_1           */
  PE_FORMULA_COSINE = 16,              /* sum_d normalize(K)_d * normalize(V)_d (:23-27) */
  PE_FORMULA_POSITION = 17             /* "streamingLLM" as a score: 1-based position in the
                                          scored slice (:124-126)      */
} pe_score_formula;
#define PE_SCORE_FORMULA_SHIFT 8

/* layouts accepted by pe_block_score_value_l2 */
typedef enum {
  PE_SCORE_TOKEN_MAJOR = 0, /* dense [block_size, num_kv_heads, head_size] per block */
  PE_SCORE_CACHE = 1,       /* paged value cache block, logical V[token, head, :]   */
  PE_SCORE_CACHE_FLAT = 2,  /* paged block re-viewed flat as [-1, heads, head_size],
                               the reference's decode-time view (quirk)             */
  PE_SCORE_KEY_CACHE = 3    /* paged KEY cache block [heads][head_size/x][block_size][x],
                               logical K[token, head, :] (x = 16 / element size)     */
} pe_score_layout;

const char* pe_last_error(void);
int pe_version(void);
/* Tunables read from the environment (PE_DECODE_SPLITS, PE_DECODE_WAVES,
 * PE_EVICT_FAST, PE_PREFILL_SHAPE, ...: experiment and test switches, none is needed in
 * production) are read once, on first use, so that no launch pays for getenv.  A process that
 * changes one of them afterwards calls pe_reload_env() to have them read again. */
void pe_reload_env(void);

/* Conditions a kernel survives by clamping but that mean the caller passed inconsistent
 * arguments (the reference's kernels would read out of bounds instead).  The kernels record
 * them in device memory; pe_device_diagnostics copies the flags of the current device to the
 * host (it synchronises the device: a debugging / test call, not for the serving loop) and
 * clears them when reset != 0. */
enum {
  PE_DIAG_DECODE_LEN_CLAMPED = 1, /* pe_paged_decode / pe_decode_step: a seq_lens entry exceeded
                                     max_num_blocks_per_seq * block_size (or, on the generic
                                     path, max_seq_len) and was clamped                        */
  PE_DIAG_SCORE_ROW_FULL = 2,     /* pe_evict_select_compact: a sequence's score row already
                                     held score_table_width entries; the new block's score was
                                     not appended                                              */
  PE_DIAG_TABLE_STATE_MISMATCH = 4 /* pe_table_state_advance: the step's token does not fall into
                                     the last block of the device-resident row (host and device
                                     counters disagree, or the row is full); the token's slot
                                     was reported as -1 (not cached)                           */
};
int pe_device_diagnostics(unsigned int* flags_out, int reset);

/* ------------------------------------------------------------------------------------
 * KV write.  Replaces torch.ops._C_cache_ops.reshape_and_cache
 * (csrc/torch_bindings.cpp:444-450, kernel csrc/cache_kernels.cu:152-204,263-289;
 * caller vllm/attention/ops/paged_attn.py:65-84).
 *   key/value      device [num_tokens, num_kv_heads, head_size], token stride given in
 *                  elements (head and element strides are dense)
 *   slot_mapping   device int64[num_tokens]; slot < 0 = padding, skipped
 *   src_index      optional device int32: row of key/value to read for output token i.
 *                  NULL: row i.  src_index_head_stride == 0: one row per token
 *                  (src_index[i]); == 1: one row per (token, head)
 *                  (src_index[i*num_kv_heads + h]); == 2: one (row, head) pair per
 *                  (token, head), encoded row*num_kv_heads + head (the compat_q2 output of
 *                  pe_prompt_token_select) — the prompt-prune gather fused into
 *                  the write (replaces the prune_prompt_inplace + reshape_and_cache pair,
 *                  vllm/attention/backends/torch_cuda_sdpa.py:888-892,1020-1024)
 *   zero_slots     optional device int64[num_zero_slots]: slots whose K and V rows are
 *                  set to zero after the write (negative entries skipped).  Replaces the
 *                  zero rows the reference appends from TmpCacheSingleton
 *                  (torch_cuda_sdpa.py:919-933,962-976; vllm/worker/tmp_cache_singleton.py)
 *   num_blocks     blocks in the cache, for the range check of slots (a slot beyond the
 *                  cache is skipped)
 * ---------------------------------------------------------------------------------- */
int pe_reshape_and_cache(const void* key, const void* value, void* key_cache,
                         void* value_cache, const int64_t* slot_mapping,
                         int64_t num_tokens, int64_t key_stride, int64_t value_stride,
                         const int32_t* src_index, int src_index_head_stride,
                         const int64_t* zero_slots, int64_t num_zero_slots,
                         int num_kv_heads, int head_size, int block_size,
                         int64_t num_blocks, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Decode attention over the surviving blocks.  Replaces torch.ops._C.paged_attention_v1
 * and paged_attention_v2 + reduce (csrc/torch_bindings.cpp:27-51,
 * csrc/attention/attention_kernels.cuh:86-669; dispatch heuristic
 * vllm/attention/ops/paged_attn.py:87-191) for the non-fp8, non-block-sparse arms.
 *   out            device [num_seqs, num_heads, head_size] (dense)
 *   query          device [num_seqs, num_heads, head_size], seq stride q_stride (elements)
 *   block_tables   device int32[num_seqs, max_num_blocks_per_seq]
 *   seq_lens       device int32[num_seqs]: KV length per sequence (seq_kv_lens_tensor
 *                  under eviction, torch_cuda_sdpa.py:1044-1045)
 *   alibi_slopes   optional device float[num_heads]
 *   max_seq_len    upper bound of seq_lens (sizes the split-KV partition count only)
 *   workspace      device scratch of at least pe_paged_decode_workspace_bytes(...) bytes;
 *                  may be NULL when that function returns 0.  Calls that overlap on one
 *                  workspace (different streams) race on it.
 * softmax follows the CUDA kernel: p = exp(l - max) / (sum + 1e-6) (:342).
 * Kernel choice: 16-bit dtypes with block_size 16 (every head size of the reference's op,
 * vllm/attention/ops/paged_attn.py:37: 32, 64, 80, 96, 112, 120, 128, 192, 256) or block_size
 * 8/32 (head sizes that are multiples of 32), up to 16 query heads per KV head and strides that are multiples of 8
 * elements run the matrix-core kernel, which reads K/V once per KV head; everything else (f32,
 * other block sizes, more query heads per KV head) runs a general kernel with the same results
 * and no tuning.
 * ---------------------------------------------------------------------------------- */
int64_t pe_paged_decode_workspace_bytes(int num_seqs, int num_heads, int num_kv_heads,
                                        int head_size, int block_size, int max_seq_len);
int pe_paged_decode(void* out, const void* query, const void* key_cache,
                    const void* value_cache, const int32_t* block_tables,
                    const int32_t* seq_lens, const float* alibi_slopes, int num_seqs,
                    int num_heads, int num_kv_heads, int head_size, int block_size,
                    int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride,
                    int64_t kv_head_stride, float scale, int max_seq_len, int dtype,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* One decode layer in one launch: the KV write of this step's token (what
 * pe_reshape_and_cache does for a decode batch: key/value [num_seqs, num_kv_heads, head_size],
 * slot_mapping int64[num_seqs], negative = skip) fused in front of the attention of
 * pe_paged_decode.  Replaces the reshape_and_cache + paged_attention_v1/v2 pair that the
 * reference issues per layer in its decode branch
 * (vllm/attention/backends/torch_cuda_sdpa.py:936,947,1049-1062).  The kernel takes the step's
 * token from new_key/new_value in registers as logical position seq_lens[s]-1, reads only the
 * seq_lens[s]-1 older tokens from the cache, and stores the new row to slot_mapping[s] without
 * reading it back.  Requirements beyond pe_paged_decode's:
 *   - the shape must be one pe_paged_decode_supports_fused_write() accepts;
 *   - seq_lens[s] counts the new token;
 *   - slot_mapping[s] must be the slot of that position in the table the call is given:
 *       block_tables[s][(seq_lens[s]-1) / block_size] * block_size + (seq_lens[s]-1) % block_size
 *     (a decode step's slot mapping satisfies this, also on an eviction step, where the call
 *     gets the compacted table and the length shortened by one block: the step's token is the
 *     last token of the sequence's last block in both views).  The kernel does not check it: a
 *     slot that is inconsistent with the table leaves this step's output correct and the cache
 *     wrong for later steps;
 *   - a block that receives a new token in this call must not appear in another sequence's
 *     table: that sequence's workgroups read the block while the row is being stored, so its
 *     output would depend on timing.  A block manager never builds such tables (a block shared
 *     by two sequences is copied before either writes into it, pe_copy_blocks); with
 *     pe_reshape_and_cache + pe_paged_decode the write is complete before any read.
 * A negative slot (padding) or one beyond num_blocks*block_size skips the fusion for that
 * sequence: nothing is stored and all seq_lens[s] positions are read from the cache, as
 * pe_paged_decode does.  A shape or stride the fused kernel cannot serve returns
 * PE_ERR_UNSUPPORTED and writes nothing. */
int pe_paged_decode_supports_fused_write(int num_heads, int num_kv_heads, int head_size,
                                         int block_size, int dtype);
int pe_decode_step(void* out, const void* query, const void* new_key, const void* new_value,
                   void* key_cache, void* value_cache, const int64_t* slot_mapping,
                   const int32_t* block_tables, const int32_t* seq_lens,
                   const float* alibi_slopes, int num_seqs, int num_heads, int num_kv_heads,
                   int head_size, int block_size, int64_t num_blocks,
                   int max_num_blocks_per_seq, int64_t q_stride, int64_t new_k_stride,
                   int64_t new_v_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                   float scale, int max_seq_len, int dtype, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* pe_decode_step_rope: pe_decode_step with the rotary embedding of the step's query and key rows
 * done inside the launch.  Replaces the three launches the reference issues per layer of a decode
 * step: rotary_embedding (csrc/pos_encoding_kernels.cu:124-162, called from
 * vllm/model_executor/models/llama.py:202-203), reshape_and_cache and paged_attention_v1/v2
 * (vllm/attention/backends/torch_cuda_sdpa.py:936,947,1049-1062).  query and new_key are given
 * UNROTATED and are left as they are (the reference rotates them in place; nothing after
 * attention reads them); the rotated key row is what the cache receives.  positions int64
 * [num_seqs] (clamped to [0, max_position)), cos_sin_cache [max_position, rot_dim] of the storage
 * type (a row = rot_dim/2 cosines then rot_dim/2 sines), is_neox as in rotary_embedding.
 * positions == NULL: cos_sin_cache is a per-step table [max_position >= num_seqs, rot_dim] whose
 * row s belongs to sequence s (the rows of the model's table gathered by the step's positions,
 * once for all layers): the kernel then requests the row together with the query instead of
 * behind a dependent load of the position, which takes the rotation off the launch's start-up
 * path.  The
 * arithmetic is that of pe_rope_reshape_and_cache (each product and their sum rounded to the
 * storage type), so the result and the cache bytes equal those of pe_rope_reshape_and_cache
 * followed by pe_paged_decode.  A sequence whose slot is negative is not cached, its query is
 * still rotated and its new key row is ignored, as in pe_decode_step.  rot_dim must be a multiple
 * of 64 (NeoX pairing) or 16 (GPT-J pairing), head_size 64 or 128 (block_size 8, 16 or 32), and
 * the shape one pe_paged_decode_supports_fused_write accepts; otherwise PE_ERR_UNSUPPORTED, nothing
 * written. */
int pe_decode_step_rope(void* out, const void* query, const void* new_key, const void* new_value,
                        void* key_cache, void* value_cache, const int64_t* slot_mapping,
                        const int32_t* block_tables, const int32_t* seq_lens,
                        const float* alibi_slopes, const int64_t* positions,
                        const void* cos_sin_cache, int rot_dim, int is_neox, int64_t max_position,
                        int num_seqs, int num_heads, int num_kv_heads, int head_size,
                        int block_size, int64_t num_blocks, int max_num_blocks_per_seq,
                        int64_t q_stride, int64_t new_k_stride, int64_t new_v_stride,
                        int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                        int max_seq_len, int dtype, void* workspace, int64_t workspace_bytes,
                        void* stream);

/* pe_paged_decode_noalloc: pe_paged_decode without a caller workspace.  It has no counterpart
 * in the reference; it exists for callers that cannot allocate device memory around the call
 * (code under HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc).  The
 * split-KV partial results go to a scratch array that is part of the library's device image:
 * pe_paged_decode_noalloc_max_bytes() bytes (8 MiB in the shipped build), held once by every
 * GPU the library runs on until the process ends.  The call allocates and frees nothing; a call
 * whose partials would not fit returns PE_ERR_WORKSPACE and launches nothing.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (different streams or threads) race
 * on it and must be serialised by the caller.  Every byte of the array that a call reads was
 * written by the same call, so nothing has to be cleared between calls.  The array used is that of the device `stream` belongs to (of the current device for
 * the NULL stream), which must be the current device.  pe_paged_decode_noalloc_max_bytes() also
 * resolves the array's address on the current device: call it once per device at set-up, outside
 * any stream capture, so that the first pe_paged_decode_noalloc on that device may run under
 * capture (the address lookup loads the device image and is not a capturable operation).
 * Arguments and results are those of pe_paged_decode. */
int64_t pe_paged_decode_noalloc_max_bytes(void);
int pe_paged_decode_noalloc(void* out, const void* query, const void* key_cache,
                            const void* value_cache, const int32_t* block_tables,
                            const int32_t* seq_lens, const float* alibi_slopes, int num_seqs,
                            int num_heads, int num_kv_heads, int head_size, int block_size,
                            int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride,
                            int64_t kv_head_stride, float scale, int max_seq_len, int dtype,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Prefill attention.  Replaces TorchCUDASDPABackendImpl._run_sdpa_forward
 * (vllm/attention/backends/torch_cuda_sdpa.py:1069-1127): per request causal attention
 * of q[Lq] over k/v[Lkv] with grouped KV heads (no repeat_interleave copy) and the
 * mask alignment of scaled_dot_product_attention(is_causal=True): query i attends kv
 * rows 0..min(i, Lkv-1) (top-left aligned when Lq != Lkv, i.e. when the prompt was
 * pruned before attention).
 *   q [total_q, num_heads, head_size], k/v [total_kv, num_kv_heads, head_size], token
 *   strides in elements; cu_q / cu_kv device int32[num_seqs+1] prefix sums;
 *   max_q_len = longest query (sizes the grid).
 * ---------------------------------------------------------------------------------- */
int pe_prefill_causal(void* out, const void* q, const void* k, const void* v,
                      const int32_t* cu_q, const int32_t* cu_kv, int num_seqs,
                      int max_q_len, int num_heads, int num_kv_heads, int head_size,
                      int64_t q_stride, int64_t k_stride, int64_t v_stride,
                      int64_t out_stride, float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Zero rows for every layer at once.  The reference zeroes the evicted slots of streamingLLM-1 and
 * inverse_key_l2 by appending zero rows to each layer's KV write (TmpCacheSingleton,
 * vllm/worker/tmp_cache_singleton.py:8-88, torch_cuda_sdpa.py:919-933,962-976); the slots are the
 * same for all layers of a step, so one launch at the start of the step zeroes them in every
 * layer's cache and each layer keeps its one-launch decode step.
 *   key_cache_ptrs / value_cache_ptrs  DEVICE arrays of num_layers device pointers (dense caches
 *                                      of the layout above, all of one shape)
 *   zero_slots                         device int64[num_zero_slots]; negative or out-of-range
 *                                      entries are skipped
 * ---------------------------------------------------------------------------------- */
int pe_zero_slots_layers(void* const* key_cache_ptrs, void* const* value_cache_ptrs, int num_layers,
                         const int64_t* zero_slots, int64_t num_zero_slots, int num_kv_heads,
                         int head_size, int block_size, int64_t num_blocks, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Prompt prune, dense form.  Replaces KVCachePruner.prune_prompt_inplace
 * (vllm/attention/kvcache_prunner.py:167-281) driven per request by
 * _prune_prompts_keys_and_values_new (torch_cuda_sdpa.py:1222-1296): gathers the kept
 * rows of key/value into out_key/out_value [n_out, num_kv_heads, head_size] (dense).
 * src_index as in pe_reshape_and_cache.
 * ---------------------------------------------------------------------------------- */
int pe_prompt_prune_gather(const void* key, const void* value, void* out_key,
                           void* out_value, const int32_t* src_index,
                           int src_index_head_stride, int64_t n_out, int64_t key_stride,
                           int64_t value_stride, int num_kv_heads, int head_size,
                           int dtype, void* stream);

/* Kept-row index of the streamingLLM prompt prune for a batch of prompts
 * (kvcache_prunner.py:183-218): for request r with query rows [cu_q[r], cu_q[r+1]) and
 * kept rows [cu_kept[r], cu_kept[r+1]) writes src_index[cu_kept[r] + n] = absolute row
 * of the n-th kept token.  cu_q/cu_kept device int32[num_seqs+1].
 * cache_budget == 0: the number of pruned tokens of request r is its length minus its kept
 * length - the "percentage" prune type, whose prompt prune drops the oldest
 * block_size*floor(int(middle/evict_freq)/block_size) tokens after the first block
 * (vllm/attention/kvcache_prunner_old.py:137-200). */
int pe_prompt_keep_index_streaming(int32_t* src_index, const int32_t* cu_q,
                                   const int32_t* cu_kept, int num_seqs, int cache_budget,
                                   int block_size, void* stream);

/* Token-level prompt prune for "local"/"global" (score = ||V||_2, or the pe_score_formula in
 * bits 8..15 of `method`) and "inverse_key_l2" (score = 1/(||K||_2+1e-8))
 * (kvcache_prunner.py:29-49,220-257): per request and per KV
 * head drops the n_prune lowest-score middle tokens and writes the per-(token, head)
 * kept-row index src_index[(cu_kept[r]+n)*num_kv_heads + h].  Scores are rounded to the
 * storage type like torch.norm's result; equal scores are dropped lowest token first.
 * compat_q2 != 0 lays the index out the way the reference's `middle[mask].view(-1,H,D)`
 * re-packs rows across heads (kvcache_prunner.py:251-252); then each entry is
 * row*num_kv_heads + head of the source (token, head) pair.
 * cache_budget == 0 selects the geometry of the "percentage" prune type
 * (vllm/attention/kvcache_prunner_old.py:201-262): the last tail_blocks (= num_block_merge)
 * blocks and the remainder stay, and request r loses as many middle tokens as its length
 * exceeds its kept length; tail_blocks is ignored otherwise (the budget type keeps one block).
 * total_q = number of key/value rows (= cu_q[num_seqs]).
 * workspace: pe_prompt_token_select_workspace_bytes(total_q, num_kv_heads) bytes. */
int64_t pe_prompt_token_select_workspace_bytes(int64_t total_q, int num_kv_heads);
int pe_prompt_token_select(int32_t* src_index, const void* key, const void* value,
                           const int32_t* cu_q, const int32_t* cu_kept, int num_seqs,
                           int64_t total_q, int cache_budget, int block_size, int method, int compat_q2,
                           int tail_blocks, int64_t key_stride, int64_t value_stride, int num_kv_heads,
                           int head_size, int dtype, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Per-block L2-norm score.  Replaces KVCachePruner.get_block_score
 * (kvcache_prunner.py:51-56): norm over head_size, mean over heads, sum over the
 * block's tokens, each reduction rounded to the storage type as torch does.  The key-norm
 * formulas of kvcache_prunner.py:57-66 are the same launch on the key tensor (dense rows, or
 * PE_SCORE_KEY_CACHE for a paged key block) combined by the caller.
 *   layout PE_SCORE_TOKEN_MAJOR: values = dense [*, num_kv_heads, head_size]; block i
 *     covers rows [block_ids[i]*block_size, +block_size) (block_ids NULL: block i = i)
 *     — the prompt-time call site torch_cuda_sdpa.py:1278-1288;
 *   layout PE_SCORE_CACHE / PE_SCORE_CACHE_FLAT: values = paged value cache, block i =
 *     physical block block_ids[i] — the decode-time call site torch_cuda_sdpa.py:1518-1526.
 *   scores_out device float[n_blocks].
 * ---------------------------------------------------------------------------------- */
int pe_block_score_value_l2(float* scores_out, const void* values,
                            const int32_t* block_ids, int n_blocks, int layout,
                            int64_t block_stride, int num_kv_heads, int head_size,
                            int block_size, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * The same score in two steps, for head-sharded (tensor-parallel) caches: each rank holds
 * num_kv_heads of the total_kv_heads KV heads, so the head mean of
 * kvcache_prunner.py:51-56 spans ranks.
 *   pe_block_token_norm_sums: sums_out[i*block_size + t] = sum over the LOCAL heads of the
 *     (storage-type rounded) norms of token t of block i — f32, not rounded;
 *   the caller adds the ranks' arrays (all-reduce, one f32 [n_blocks*block_size]
 *     message);
 *   pe_block_score_from_token_sums: scores_out[i] = round(sum_t round(sums[i][t] /
 *     total_kv_heads)) with the rounding of the storage type.
 * With every head on one rank the two steps give exactly pe_block_score_value_l2; across
 * ranks they give the same bits whenever the f32 head sums are exact (norms are 8- or 11-bit
 * significands, so always unless one token's norms span more than 2^13 across heads).
 * Arguments as pe_block_score_value_l2 (PE_SCORE_CACHE_FLAT is not head-separable and is
 * rejected).
 * ---------------------------------------------------------------------------------- */
int pe_block_token_norm_sums(float* sums_out, const void* values, const int32_t* block_ids,
                             int n_blocks, int layout, int64_t block_stride,
                             int num_kv_heads, int head_size, int block_size, int dtype,
                             void* stream);
int pe_block_score_from_token_sums(float* scores_out, const float* sums, int n_blocks,
                                   int block_size, int total_kv_heads, int dtype,
                                   void* stream);

/* ------------------------------------------------------------------------------------
 * Decode-step eviction decision + block-table compaction, one launch per step.
 * Replaces _prune_decode_reqs_streamingLLM / _prune_decode_reqs_topk /
 * _create_updated_block_tables (torch_cuda_sdpa.py:1375-1433,1478-1586) and the score
 * bookkeeping of BlockL2NormManager (vllm/attention/block_l2norm_manager.py:64-75).
 *
 * For sequence b with logical length seq_lens[b] (trigger: seq_len > cache_budget and
 * seq_len % block_size == 0, :1390-1403,1491-1498):
 *   streamingLLM / streamingLLM-1: victim = block index 1; streamingLLM-1 additionally
 *     reports, on non-boundary steps past the budget, the slot
 *     block_tables[b][1]*block_size + seq_len%block_size - 1 in zero_slots_out[b];
 *   local / global: the just-filled block (table entry num_blocks[b]-1) is scored from the
 *     value cache (or taken from new_scores[b] when new_scores is non-NULL: the
 *     tensor-parallel caller computes it with the two-step form above; value_cache may
 *     then be NULL), appended to the sequence's score list
 *     score_table[score_rows[b]][0..score_counts[..]), and the victim is the first
 *     arg-min over [1, 1+topk_blocks) (local) or [1, n_scores-1) (global).  If
 *     victim_override is non-NULL and victim_override[b] >= 0 that index is used instead
 *     (the reference's random mode, :1502-1515, draws it on the host).  With
 *     drop_victim_score != 0 the victim's entry is deleted from the list so that the
 *     list keeps tracking the table (the call the reference leaves commented out, :1546).
 * Outputs: rmv_idx_out[b] = victim index or -1; out_tables[b] = row b without that
 * column, out_width entries per row (zero-filled past the row's end).  The reference's
 * shape is out_width = max_blocks-1, which cuts the last block off a row that evicts nothing
 * but is max_blocks long; a caller that knows the row lengths passes the longest surviving
 * row instead.  compat_q1 != 0 reproduces the reference's in-loop slice (:1583), needs
 * out_width == max_blocks-1 and runs the rows serially;
 * zero_slots_out[b] = slot to zero or -1 (may be NULL unless streamingLLM-1).
 * out_tables may be NULL to skip compaction.
 * ---------------------------------------------------------------------------------- */
int pe_evict_select_compact(int32_t* rmv_idx_out, int32_t* out_tables,
                            int64_t* zero_slots_out, const int32_t* block_tables,
                            const int32_t* seq_lens, const int32_t* num_blocks,
                            const int32_t* victim_override, int batch, int max_blocks,
                            int out_width, int method, int cache_budget, int block_size, int topk_blocks,
                            int drop_victim_score, int compat_q1, int score_layout,
                            float* score_table, int32_t* score_counts,
                            const int32_t* score_rows, int score_table_width,
                            const void* value_cache, int64_t kv_block_stride,
                            int num_kv_heads, int head_size, const float* new_scores,
                            int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Device-resident block tables of the running decode batch (the "eviction scheduler" half that
 * lives on the GPU).  State, caller-owned device arrays in batch order:
 *     tables     int32 [capacity, max_blocks]  block ids of batch slot b, zero padded
 *     num_blocks int32 [capacity]              entries in use
 *     kv_lens    int32 [capacity]              tokens physically held (BlockTable._num_cached_slots,
 *                                              vllm/core/block/block_table.py:119-126,183-196)
 *     seq_lens   int32 [capacity]              logical length (_num_full_slots)
 * The host uploads a row only when a sequence joins the batch; per step it sends one int32 per
 * sequence (the id of a newly allocated block, or -1), so the bytes crossing PCIe per step do
 * not depend on the table width.
 *
 * pe_table_state_advance: one decode step's append for every batch slot - replaces, on the
 *   device, BlockTable.append_token_ids (block_table.py:136-196), the decode arm of
 *   _compute_slot_mapping (vllm/attention/backends/torch_cuda_sdpa.py:591-598) and the
 *   kv-length adjustment of ModelInputForGPUBuilder._compute_lens
 *   (vllm/worker/model_runner.py:538-543).  new_blocks[b] >= 0 is appended to row b; both
 *   lengths grow by one; slot_mapping_out[b] = last block * block_size + (kv_len-1) % block_size;
 *   attn_kv_lens_out[b] = the physical length attention must see this step: kv_len minus the
 *   tokens this step evicts (evict_rule 0: none; 1: budget type, block_size tokens when
 *   seq_len > cache_budget and seq_len % block_size == 0 - pass 0 for inverse_key_l2, which never
 *   removes a block; 2: percentage type, (num_block_merge-1)*block_size tokens when
 *   seq_len % (block_size*evict_freq) == 0 and kv_len > (initial_blocks+num_block_merge)*block_size).
 *   last_blocks_out (optional) = id of each row's last block (what a head-sharded caller scores).
 * pe_table_state_advance_ring: the same step with the per-sequence entries taken from a ring of
 *   rows, new_blocks_ring int32 [ring_len, ring_stride]: sequence b reads row
 *   (seq_lens[b] + 1) % ring_len - its logical length after the step - so the launch has the same
 *   arguments at every step and can be a node of a replayed HIP graph.  The ring may be pinned
 *   host memory mapped into the device (then nothing is copied per step: the host stores the
 *   step's entries, -1 where no block was allocated, before it submits the launch).  The caller
 *   keeps the host fewer than ring_len steps ahead of the device and lets the pending launches
 *   drain before it gives a batch slot to another sequence.
 * pe_evict_select_compact_inplace: pe_evict_select_compact on that state - the victim's column
 *   is closed inside the row (no second table), num_blocks[b] and kv_lens[b] are updated, and
 *   freed_blocks_out[b] = the victim's block id (-1: none) for the host's free list.  Victim
 *   choice, score bookkeeping and zero_slots_out exactly as pe_evict_select_compact.  Every layer
 *   of the step then reads block_tables with attn_kv_lens.
 * ---------------------------------------------------------------------------------- */
int pe_table_state_advance(int32_t* tables, int32_t* num_blocks, int32_t* kv_lens,
                           int32_t* seq_lens, const int32_t* new_blocks, int batch, int max_blocks,
                           int64_t* slot_mapping_out, int32_t* attn_kv_lens_out,
                           int32_t* last_blocks_out, int evict_rule, int cache_budget,
                           int block_size, int evict_freq, int num_block_merge, int initial_blocks,
                           void* stream);
int pe_table_state_advance_ring(int32_t* tables, int32_t* num_blocks, int32_t* kv_lens,
                                int32_t* seq_lens, const int32_t* new_blocks_ring, int ring_len,
                                int64_t ring_stride, int batch, int max_blocks,
                                int64_t* slot_mapping_out, int32_t* attn_kv_lens_out,
                                int32_t* last_blocks_out, int evict_rule, int cache_budget,
                                int block_size, int evict_freq, int num_block_merge,
                                int initial_blocks, void* stream);
int pe_evict_select_compact_inplace(int32_t* rmv_idx_out, int32_t* freed_blocks_out,
                                    int64_t* zero_slots_out, int32_t* block_tables,
                                    int32_t* num_blocks, int32_t* kv_lens, const int32_t* seq_lens,
                                    const int32_t* victim_override, int batch, int max_blocks,
                                    int method, int cache_budget, int block_size, int topk_blocks,
                                    int drop_victim_score, int score_layout, float* score_table,
                                    int32_t* score_counts, const int32_t* score_rows,
                                    int score_table_width, const void* value_cache,
                                    int64_t kv_block_stride, int num_kv_heads, int head_size,
                                    const float* new_scores, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Decode step of the "percentage" prune type (README-era flags --cache-prune-type percentage
 * --evict-freq N; code kept in vllm/attention/kvcache_prunner_old.py:264-315,
 * vllm/core/page_evict_kv_util_bak.py:66-106 and the decode loop left commented in
 * vllm/attention/backends/torch_cuda_sdpa.py:1299-1373).  Every evict_freq blocks a sequence
 * gives up num_block_merge-1 blocks: streamingLLM drops table columns
 * [initial_blocks, initial_blocks+num_block_merge-1); value_l2 first merges the num_block_merge
 * blocks from column initial_blocks into the first of them and drops the others.
 *
 * pe_merge_blocks_value_l2: for every listed row b (rows = device int32[n_rows] of batch
 *   indices; a negative entry is skipped, so a list built on the device can switch rows off) and
 *   every KV head, reads the num_block_merge blocks block_tables[b][first_block ..)
 *   of both caches, scores each of their tokens with ||V[token, head, :]||_2 (rounded to the
 *   storage type), keeps the block_size highest (equal scores: the lowest token goes first),
 *   and writes the kept K and V rows, in their original order, into block
 *   block_tables[b][first_block].  Replaces prune_oldest_block + the write-back through
 *   reshape_and_cache; the reference selects on a flat re-view of the cache blocks, repacks the
 *   kept rows across heads and stores the kept keys as values (kvcache_prunner_old.py:312-313)
 *   - this entry point implements the per-head selection those lines intend.
 *   At most 256 tokens and 160 KiB of K+V per (row, head) merge.
 * pe_block_table_remove_range: out_tables[b] = row b of block_tables without columns
 *   [start[b], start[b]+n_remove) (start[b] < 0: row copied), out_width entries per row,
 *   zero-filled past the source row.  The compaction step of that decode loop for any
 *   num_block_merge (pe_evict_select_compact removes one column).
 * ---------------------------------------------------------------------------------- */
int pe_merge_blocks_value_l2(void* key_cache, void* value_cache, const int32_t* block_tables,
                             const int32_t* rows, int n_rows, int max_blocks, int first_block,
                             int num_block_merge, int num_kv_heads, int head_size,
                             int block_size, int64_t kv_block_stride, int64_t kv_head_stride,
                             int dtype, void* stream);
int pe_block_table_remove_range(int32_t* out_tables, const int32_t* block_tables,
                                const int32_t* start, int n_remove, int batch, int max_blocks,
                                int out_width, void* stream);
/* pe_merge_blocks_scored: pe_merge_blocks_value_l2 with the token score chosen by `formula`
 * (a pe_score_formula): the decode-time merge for every --decode-evict-method name get_score
 * knows (vllm/attention/kvcache_prunner_old.py:77-134,297-315).  Scores of a token come from its
 * key and value rows of the KV head at hand; the kept tokens are those with the HIGHEST scores.
 * pe_token_scores: scores_out[t * num_kv_heads + h] = that score of row t, head h of dense
 * key / value tensors [num_tokens, num_kv_heads, head_size] (f32 holding a value of the storage
 * type; PE_FORMULA_POSITION counts rows from 1) - get_score itself, for callers and tests. */
int pe_merge_blocks_scored(void* key_cache, void* value_cache, const int32_t* block_tables,
                           const int32_t* rows, int n_rows, int max_blocks, int first_block,
                           int num_block_merge, int num_kv_heads, int head_size, int block_size,
                           int64_t kv_block_stride, int64_t kv_head_stride, int formula, int dtype,
                           void* stream);
int pe_token_scores(float* scores_out, const void* key, const void* value, int64_t num_tokens,
                    int64_t key_stride, int64_t value_stride, int num_kv_heads, int head_size,
                    int formula, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Rotary position embedding fused with the KV write - the two launches that precede the
 * attention path in the model (vllm/model_executor/models/llama.py:202-205):
 * rotary_embedding (csrc/pos_encoding_kernels.cu:10-35,124-162; schema
 * csrc/torch_bindings.cpp "rotary_embedding(Tensor positions, Tensor! query, Tensor! key,
 * int head_size, Tensor cos_sin_cache, bool is_neox)") followed by reshape_and_cache
 * (csrc/cache_kernels.cu:152-289).
 *   query [num_tokens, num_heads, head_size] and key [num_tokens, num_kv_heads, head_size]
 *     (row strides q_stride / k_stride, e.g. views of the fused qkv projection) are rotated
 *     in place: element pairs (i, i + rot_dim/2) for is_neox != 0, (2i, 2i+1) otherwise,
 *     with cos = cos_sin_cache[pos][i], sin = cos_sin_cache[pos][rot_dim/2 + i]; the products
 *     and their sum are each rounded to the storage type, as the reference kernel does.
 *     Elements past rot_dim pass through.  positions are clamped to [0, max_position).
 *   With key_cache != NULL the rotated key row and the value row of token t are also written
 *     to slot slot_mapping[t] (< 0 or out of range: token not cached) in the layouts of
 *     pe_reshape_and_cache; with key_cache == NULL only the rotation is done.
 * ---------------------------------------------------------------------------------- */
int pe_rope_reshape_and_cache(void* query, void* key, const void* value, void* key_cache,
                              void* value_cache, const int64_t* positions,
                              const void* cos_sin_cache, const int64_t* slot_mapping,
                              int64_t num_tokens, int64_t q_stride, int64_t k_stride,
                              int64_t v_stride, int num_heads, int num_kv_heads, int head_size,
                              int rot_dim, int is_neox, int64_t max_position, int block_size,
                              int64_t num_blocks, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * One-shot all-reduce (sum) for decode-sized messages: the exchange behind the row-parallel
 * o_proj that follows attention under tensor parallelism
 * (vllm/model_executor/layers/linear.py:1099-1100).  MI355X counterpart of the reference's
 * one-stage custom all-reduce (csrc/custom_all_reduce.cu:13-144, csrc/custom_all_reduce.cuh),
 * which vllm/distributed/parallel_state.py:357-379 tries before NCCL: every rank owns a buffer
 * all ranks of the node have mapped, publishes its input there, flags its peers, and sums the
 * N inputs itself (rank order, f32 accumulation: every rank gets the same bits).  Larger
 * messages belong to RCCL.
 *   pe_ar_buffer_bytes(max_message_bytes): size of a rank's shared buffer (header + 2 slots).
 *   pe_ar_alloc / pe_ar_free: the buffer, in uncached device memory, zeroed.  These two and
 *     the IPC calls are set-up calls: they allocate / map and synchronise the device.
 *   pe_ar_get_ipc_handle / pe_ar_open_ipc_handle / pe_ar_close_ipc_handle: export the buffer to
 *     the other ranks (handles are pe_ar_ipc_handle_bytes() opaque bytes, exchanged by the
 *     caller over any host channel) and map a peer's.
 *   pe_allreduce_oneshot: out[i] = sum over ranks of in[i], `count` elements of `dtype`
 *     (message a multiple of 16 bytes and at most the slot size, else PE_ERR_WORKSPACE;
 *     out may be in).  peer_buffers: HOST array of `world` device pointers, entry r = rank r's
 *     buffer as mapped in this process (own buffer at index `rank`).  seq: 1, 2, 3, ... - the
 *     number of this call, the same on every rank (all ranks issue the same calls in the same
 *     order; it may wrap from 0xfffffffe to 1, never to an equal parity).  A workgroup whose
 *     peers do not show up within timeout_us microseconds (<= 0: ten seconds) stops waiting
 *     instead of hanging the GPU, stores NaN for its part of the output rather than a partial
 *     sum, counts the event in the buffer header (pe_ar_timeouts, a synchronising read) and,
 *     when status_host is given, sets status_host[0] != 0 ([1] = seq of the call, [2] = 1 + rank
 *     of a missing peer).  The reference's barrier spins until the peer arrives
 *     (csrc/custom_all_reduce.cuh:140-200) and so can hang but not return a partial sum.
 *   pe_ar_status_alloc / pe_ar_status_free: 16 uint32 status words in pinned, device-mapped host
 *     memory (set-up calls: they allocate).  The host reads them as plain memory before a call,
 *     without touching the device; a communicator whose word 0 is set must not be used again.
 *   pe_ar_selfcheck: verifies the peer mappings once at set-up.  phase 0 (all ranks, then a host
 *     barrier with every device synchronised): each rank fills a pattern area of its own buffer
 *     and writes tagged words into every rank's buffer through the mapping; phase 1: each rank
 *     reads all peers' patterns through the mappings and the words its peers wrote into its own
 *     buffer, and adds the number of mismatching threads to *mismatches_dev (device uint32).
 *   pe_allreduce_twostage: the same sum (same bits) as reduce-scatter + all-gather over the same
 *     buffers - the reference's cross_device_reduce_2stage (csrc/custom_all_reduce.cuh:235-283),
 *     which its dispatch takes from 512 KiB with up to 4 ranks and from 256 KiB with up to 8
 *     (:476-483): rank r sums the r-th of `world` chunks, every rank collects the reduced chunks.
 *     Arguments as pe_allreduce_oneshot; the reduced chunk lives behind the input in the slot, so
 *     the message may take count*size + ceil(count*size / world) bytes of it at most.  The two
 *     forms share the call sequence: seq counts the calls of both.
 * ---------------------------------------------------------------------------------- */
int64_t pe_ar_buffer_bytes(int64_t max_message_bytes);
int pe_ar_alloc(void** ptr_out, int64_t bytes);
int pe_ar_free(void* ptr);
int pe_ar_ipc_handle_bytes(void);
int pe_ar_get_ipc_handle(void* ptr, void* handle_out);
int pe_ar_open_ipc_handle(const void* handle, void** ptr_out);
int pe_ar_close_ipc_handle(void* ptr);
int pe_ar_timeouts(const void* own_buffer, unsigned int* count_out);
int pe_ar_status_alloc(void** host_words_out);
int pe_ar_status_free(void* host_words);
int pe_ar_selfcheck(void* const* peer_buffers, int rank, int world, int phase,
                    unsigned int* mismatches_dev, void* stream);
int pe_allreduce_oneshot(void* out, const void* in, int64_t count, int dtype,
                         void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                         unsigned int seq, void* status_host, int64_t timeout_us, void* stream);
int pe_allreduce_twostage(void* out, const void* in, int64_t count, int dtype,
                          void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                          unsigned int seq, void* status_host, int64_t timeout_us, void* stream);

/* ------------------------------------------------------------------------------------
 * Block copy / swap — backend-API obligations (vllm/attention/backends/abstract.py:79-94).
 * pe_copy_blocks replaces torch.ops._C_cache_ops.copy_blocks
 * (csrc/cache_kernels.cu:68-148): cache_ptrs device arrays of num_layers device pointers,
 * block_mapping device int64[num_pairs, 2] (src, dst).
 * pe_swap_blocks replaces swap_blocks (csrc/cache_kernels.cu:24-63): block_mapping is a
 * HOST int64[num_pairs, 2]; kind 0 = device->device, 1 = device->host, 2 = host->device.
 * ---------------------------------------------------------------------------------- */
int pe_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                   const int64_t* block_mapping, int num_layers, int num_pairs,
                   int64_t numel_per_block, int dtype, void* stream);
int pe_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host,
                   int64_t num_pairs, int64_t block_bytes, int kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAGED_EVICT_H_ */
