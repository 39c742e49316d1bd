// Decode-time block merge of the "percentage" prune type on gfx950: every evict_freq blocks a
// sequence merges num_block_merge consecutive blocks of its table into the first of them by
// keeping, per KV head, the block_size tokens with the highest ||V||_2 (the reference's
// get_blocks_to_prune_and_merge_decode + prune_oldest_block,
// vllm/attention/kvcache_prunner_old.py:264-315, driven per request by the decode loop at
// vllm/attention/backends/torch_cuda_sdpa.py:1299-1373), and drops the emptied blocks from the
// table (vllm/core/page_evict_kv_util_bak.py:66-106).
//
//   merge_blocks_kernel        grid (sequence to merge, KV head), 256 threads: the K and V
//                              slabs of the merged blocks are copied to LDS (the destination
//                              block is one of the sources), one thread per token computes the
//                              score, ranks it among the others, and the kept tokens are
//                              written back in their original order
//   table_remove_range_kernel  row b without columns [start[b], start[b] + n_remove)
#include "pe_common.h"
#include "score_formulas.h"

namespace pe {

constexpr int MERGE_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(MERGE_THREADS) void merge_blocks_kernel(
    T* __restrict__ key_cache, T* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ rows, int max_blocks,
    int first_block, int num_merge, int head_size, int block_size, int64_t kv_block_stride,
    int64_t kv_head_stride, int formula) {
  constexpr int X = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int slab = head_size * block_size;  // elements of one (block, head) slab of K or of V
  T* k_s = reinterpret_cast<T*>(smem);                       // [num_merge][slab]
  T* v_s = k_s + (size_t)num_merge * slab;                   // [num_merge][slab]
  float* score = reinterpret_cast<float*>(v_s + (size_t)num_merge * slab);  // [n_tok]
  int* dest = reinterpret_cast<int*>(score + num_merge * block_size);       // [n_tok]
  const int b = rows[blockIdx.x];
  if (b < 0) return;  // an entry switched off by the caller (the whole workgroup leaves)
  const int h = blockIdx.y;
  const int tid = threadIdx.x;
  const int n_tok = num_merge * block_size;
  const int32_t* table = block_tables + (int64_t)b * max_blocks + first_block;

  // ---- 1. slabs -> LDS (16-byte pieces, cache layout kept) -------------------------------
  const int pieces = slab / X;
  for (int i = tid; i < num_merge * pieces; i += MERGE_THREADS) {
    const int m = i / pieces, p = i % pieces;
    const int64_t base = (int64_t)table[m] * kv_block_stride + (int64_t)h * kv_head_stride;
    reinterpret_cast<u32x4*>(k_s + (size_t)m * slab)[p] =
        reinterpret_cast<const u32x4*>(key_cache + base)[p];
    reinterpret_cast<u32x4*>(v_s + (size_t)m * slab)[p] =
        reinterpret_cast<const u32x4*>(value_cache + base)[p];
  }
  __syncthreads();

  // ---- 2. score of token t: ||V[t, h, :]||_2 rounded to the storage type, or another formula
  //         of get_score (kvcache_prunner_old.py:77-134) over the token's key and value rows ------
  if (tid < n_tok) {
    const T* v = v_s + (size_t)(tid / block_size) * slab + tid % block_size;  // V[d][tok]
    if (formula == PE_FORMULA_VALUE_L2) {
      float ss = 0.f;
      for (int d = 0; d < head_size; ++d) {
        const float x = to_f32(v[d * block_size]);
        ss += x * x;
      }
      score[tid] = round_to<T>(sqrtf(ss));
    } else {
      const T* k = k_s + (size_t)(tid / block_size) * slab + (tid % block_size) * X;  // K[d/X][tok][d%X]
      score[tid] = formula_score<T>(
          formula, head_size, tid + 1,
          [&](int d) { return to_f32(k[(size_t)(d / X) * block_size * X + d % X]); },
          [&](int d) { return to_f32(v[d * block_size]); });
    }
  }
  __syncthreads();

  // ---- 3. drop the n_tok - block_size lowest scores (equal scores: lowest token first) ------
  const int n_drop = n_tok - block_size;
  bool keep = false;
  if (tid < n_tok) {
    const float s = score[tid];
    int below = 0;
    for (int u = 0; u < n_tok; ++u) {
      const float o = score[u];
      below += (o < s || (o == s && u < tid)) ? 1 : 0;
    }
    keep = below >= n_drop;
    dest[tid] = keep ? 1 : 0;
  }
  __syncthreads();
  int pos = 0;
  if (tid < n_tok)
    for (int u = 0; u < tid; ++u) pos += dest[u];
  __syncthreads();
  if (tid < n_tok) dest[tid] = keep ? pos : -1;
  __syncthreads();

  // ---- 4. kept tokens, in their original order, into the first block of the range ----------
  const int64_t out_base = (int64_t)table[0] * kv_block_stride + (int64_t)h * kv_head_stride;
  T* k_out = key_cache + out_base;    // [head_size/X][block_size][X]
  T* v_out = value_cache + out_base;  // [head_size][block_size]
  const int chunks = head_size / X;
  for (int i = tid; i < n_tok * chunks; i += MERGE_THREADS) {
    const int t = i % n_tok, c = i / n_tok;
    const int p = dest[t];
    if (p < 0) continue;
    const T* src = k_s + (size_t)(t / block_size) * slab + ((size_t)c * block_size + t % block_size) * X;
    *reinterpret_cast<u32x4*>(k_out + ((size_t)c * block_size + p) * X) =
        *reinterpret_cast<const u32x4*>(src);
  }
  for (int i = tid; i < n_tok * head_size; i += MERGE_THREADS) {
    const int t = i % n_tok, d = i / n_tok;
    const int p = dest[t];
    if (p < 0) continue;
    v_out[(size_t)d * block_size + p] =
        v_s[(size_t)(t / block_size) * slab + (size_t)d * block_size + t % block_size];
  }
}

__global__ void table_remove_range_kernel(int32_t* __restrict__ out_tables,
                                          const int32_t* __restrict__ block_tables,
                                          const int32_t* __restrict__ start, int n_remove,
                                          int max_blocks, int out_width) {
  const int b = blockIdx.x;
  const int32_t* row = block_tables + (int64_t)b * max_blocks;
  int32_t* o = out_tables + (int64_t)b * out_width;
  const int s = start[b];
  for (int j = threadIdx.x; j < out_width; j += blockDim.x) {
    const int src = (s >= 0 && j >= s) ? j + n_remove : j;
    o[j] = src < max_blocks ? row[src] : 0;
  }
}

}  // namespace pe

extern "C" {

int pe_merge_blocks_value_l2(void* key_cache, void* value_cache, const int32_t* block_tables,
                             const int32_t* rows, int n_rows, int max_blocks, int first_block,
                             int num_block_merge, int num_kv_heads, int head_size, int block_size,
                             int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                             void* stream) {
  return pe_merge_blocks_scored(key_cache, value_cache, block_tables, rows, n_rows, max_blocks,
                                first_block, num_block_merge, num_kv_heads, head_size, block_size,
                                kv_block_stride, kv_head_stride, PE_FORMULA_VALUE_L2, dtype, stream);
}

int pe_merge_blocks_scored(void* key_cache, void* value_cache, const int32_t* block_tables,
                           const int32_t* rows, int n_rows, int max_blocks, int first_block,
                           int num_block_merge, int num_kv_heads, int head_size, int block_size,
                           int64_t kv_block_stride, int64_t kv_head_stride, int formula, int dtype,
                           void* stream) {
  using namespace pe;
  PE_REQUIRE(formula_known(formula), PE_ERR_INVALID_ARG, "merge_blocks: unknown score formula %d", formula);
  PE_REQUIRE(n_rows >= 0, PE_ERR_INVALID_ARG, "negative n_rows");
  if (n_rows == 0) return PE_OK;
  PE_REQUIRE(key_cache && value_cache && block_tables && rows, PE_ERR_INVALID_ARG,
             "merge_blocks: null pointer");
  PE_REQUIRE(num_block_merge >= 2 && first_block >= 0 && first_block + num_block_merge <= max_blocks,
             PE_ERR_INVALID_ARG, "merge_blocks: blocks [%d, %d) outside a table of %d columns",
             first_block, first_block + num_block_merge, max_blocks);
  PE_REQUIRE(num_kv_heads > 0 && num_kv_heads <= 65535 && head_size > 0 && block_size > 0,
             PE_ERR_INVALID_ARG, "merge_blocks: bad sizes");
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(head_size % x == 0 && kv_block_stride % x == 0 && kv_head_stride % x == 0 &&
                 (((uintptr_t)key_cache | (uintptr_t)value_cache) % 16) == 0,
             PE_ERR_INVALID_ARG, "merge_blocks: cache slabs must be 16-byte aligned");
  const int n_tok = num_block_merge * block_size;
  PE_REQUIRE(n_tok <= MERGE_THREADS, PE_ERR_UNSUPPORTED,
             "merge_blocks: at most %d tokens per merge (got %d)", MERGE_THREADS, n_tok);
  const size_t lds = (size_t)2 * num_block_merge * head_size * block_size * elem_size(dtype) +
                     (size_t)n_tok * 8;
  PE_REQUIRE(lds <= 160 * 1024, PE_ERR_UNSUPPORTED,
             "merge_blocks: %d blocks of head size %d need %zu bytes of LDS (limit 163840)",
             num_block_merge, head_size, lds);
#define PE_LAUNCH_MERGE(T)                                                                        \
  do {                                                                                            \
    if (lds > 64 * 1024)                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_blocks_kernel<T>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    hipLaunchKernelGGL(merge_blocks_kernel<T>, dim3((unsigned)n_rows, (unsigned)num_kv_heads),    \
                       dim3(MERGE_THREADS), lds, static_cast<hipStream_t>(stream), (T*)key_cache, \
                       (T*)value_cache, block_tables, rows, max_blocks, first_block,              \
                       num_block_merge, head_size, block_size, kv_block_stride, kv_head_stride,   \
                       formula);                                                                  \
  } while (0)
  PE_DISPATCH_ALL(dtype, T, PE_LAUNCH_MERGE(T));
#undef PE_LAUNCH_MERGE
  return check_launch("merge_blocks_value_l2");
}

int pe_block_table_remove_range(int32_t* out_tables, const int32_t* block_tables,
                                const int32_t* start, int n_remove, int batch, int max_blocks,
                                int out_width, void* stream) {
  using namespace pe;
  PE_REQUIRE(batch >= 0, PE_ERR_INVALID_ARG, "negative batch");
  if (batch == 0) return PE_OK;
  PE_REQUIRE(out_tables && block_tables && start, PE_ERR_INVALID_ARG,
             "block_table_remove_range: null pointer");
  PE_REQUIRE(n_remove >= 1 && max_blocks >= 1 && out_width >= 1 && out_width <= max_blocks,
             PE_ERR_INVALID_ARG, "block_table_remove_range: need n_remove >= 1 and out_width in [1, max_blocks=%d]",
             max_blocks);
  hipLaunchKernelGGL(table_remove_range_kernel, dim3((unsigned)batch), dim3(64), 0,
                     static_cast<hipStream_t>(stream), out_tables, block_tables, start, n_remove,
                     max_blocks, out_width);
  return check_launch("block_table_remove_range");
}

}  // extern "C"
