/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement of the reference algorithms on the
 * paged-KV-eviction attention path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path never
 * does.
 *
 * Every function names the reference lines it restates (paths relative to the
 * reference checkout).  The restatement is scalar C: it follows the reference's
 * arithmetic (what is accumulated in f32, what is rounded to the storage type and
 * where) rather than its thread decomposition.
 *
 * Pinning: orc_reshape_and_cache / orc_copy_blocks are checked bit-exact, and
 * orc_paged_attention within tolerance, against the reference's own CPU kernels
 * compiled from its sources (oracle/_ref, tests/test_oracle_vs_ref.py) and against
 * the recipes of the reference's tests/kernels/test_cache.py:111-202 and
 * tests/kernels/test_attention.py:47-115.  orc_block_score is checked against
 * golden vectors produced by the reference's KVCachePruner.get_block_score
 * (tests/golden/, made by tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_BF16 = 2 };

/* ---- storage-type conversions (round to nearest even, like the GPU casts) ---- */
static float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else {
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
  }
  if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* < 2^-25 (rounds to 0) */
    return (uint16_t)sign;
  }
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  if (e < -14) { /* subnormal half */
    int shift = -14 - e + 13; /* bits to drop from the 24-bit significand */
    uint32_t half = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) half++;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
  return (uint16_t)(sign | half);
}

static inline float load_elem(const void* p, int64_t i, int dtype) {
  switch (dtype) {
    case ORC_F32:
      return ((const float*)p)[i];
    case ORC_F16:
      return f16_to_f32(((const uint16_t*)p)[i]);
    default:
      return bf16_to_f32(((const uint16_t*)p)[i]);
  }
}
static inline float round_to(float v, int dtype) {
  switch (dtype) {
    case ORC_F32:
      return v;
    case ORC_F16:
      return f16_to_f32(f32_to_f16(v));
    default:
      return bf16_to_f32(f32_to_bf16(v));
  }
}
static inline void store_elem(void* p, int64_t i, float v, int dtype) {
  switch (dtype) {
    case ORC_F32:
      ((float*)p)[i] = v;
      break;
    case ORC_F16:
      ((uint16_t*)p)[i] = f32_to_f16(v);
      break;
    default:
      ((uint16_t*)p)[i] = f32_to_bf16(v);
      break;
  }
}
static inline int elem_size(int dtype) { return dtype == ORC_F32 ? 4 : 2; }

/* exported for tests of the conversions themselves */
uint16_t orc_f32_to_bf16(float f) { return f32_to_bf16(f); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

/*
 * KV write.  Restates reshape_and_cache_kernel, csrc/cache_kernels.cu:152-204:
 * token i goes to slot slot_mapping[i] (negative = padding, skipped, :166-169);
 * key cache layout [num_blocks, num_heads, head_size/x, block_size, x] (:184-187),
 * value cache layout [num_blocks, num_heads, head_size, block_size] (:188-191).
 * Pure byte moves: bit-exact.
 */
void orc_reshape_and_cache(const void* key, const void* value, void* key_cache,
                           void* value_cache, const int64_t* slot_mapping,
                           int64_t num_tokens, int64_t key_stride,
                           int64_t value_stride, int num_heads, int head_size,
                           int block_size, int x, int dtype) {
  const int es = elem_size(dtype);
  const char* k = (const char*)key;
  const char* v = (const char*)value;
  char* kc = (char*)key_cache;
  char* vc = (char*)value_cache;
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t blk = slot / block_size;
    const int64_t off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      const int h = i / head_size;
      const int d = i % head_size;
      const int64_t tk =
          blk * num_heads * (head_size / x) * block_size * x +
          (int64_t)h * (head_size / x) * block_size * x +
          (int64_t)(d / x) * block_size * x + off * x + d % x;
      const int64_t tv = blk * num_heads * head_size * block_size +
                         (int64_t)h * head_size * block_size +
                         (int64_t)d * block_size + off;
      memcpy(kc + tk * es, k + (t * key_stride + i) * es, es);
      memcpy(vc + tv * es, v + (t * value_stride + i) * es, es);
    }
  }
}

/*
 * Block copy.  Restates copy_blocks_kernel, csrc/cache_kernels.cu:68-94, for one
 * layer: for each (src, dst) pair copy numel_per_block elements of K and of V.
 */
void orc_copy_blocks(void* key_cache, void* value_cache,
                     const int64_t* block_mapping, int64_t num_pairs,
                     int64_t numel_per_block, int dtype) {
  const int64_t nb = numel_per_block * elem_size(dtype);
  for (int64_t p = 0; p < num_pairs; ++p) {
    const int64_t s = block_mapping[2 * p], d = block_mapping[2 * p + 1];
    memmove((char*)key_cache + d * nb, (char*)key_cache + s * nb, nb);
    memmove((char*)value_cache + d * nb, (char*)value_cache + s * nb, nb);
  }
}

/*
 * Single-query paged attention.  Restates paged_attention_kernel,
 * csrc/attention/attention_kernels.cuh:86-500 (v1, no partitioning), for the
 * non-block-sparse, non-fp8 arms:
 *   logits[t] = scale * <q, k_t> (f32)  (+ alibi_slope * (t - seq_len + 1), :297)
 *   m = max_t logits, e_t = exp(logits[t] - m), S = sum e_t          (:310-339)
 *   p_t = e_t * (1 / (S + 1e-6))                                    (:342-345)
 *   p_t is rounded to the storage type before the P.V product        (:398-400)
 *   out[d] = sum_t p_t * v_t[d] accumulated in f32, stored as the
 *   storage type                                                     (:431,:485+)
 * K/V are fetched through block_tables with the layouts of orc_reshape_and_cache.
 * seq_len == 0 leaves the output row zero.
 */
void orc_paged_attention(void* out, const void* query, const void* key_cache,
                         const void* value_cache, const int32_t* block_tables,
                         const int32_t* seq_lens, const float* alibi_slopes,
                         int num_seqs, int num_heads, int num_kv_heads,
                         int head_size, int block_size, int x,
                         int max_num_blocks_per_seq, int64_t q_stride,
                         int64_t kv_block_stride, int64_t kv_head_stride,
                         float scale, int dtype) {
  const int qpk = num_heads / num_kv_heads;
  int max_len = 0;
  for (int s = 0; s < num_seqs; ++s)
    if (seq_lens[s] > max_len) max_len = seq_lens[s];
  float* logits = (float*)malloc(sizeof(float) * (size_t)(max_len > 0 ? max_len : 1));
  float* qf = (float*)malloc(sizeof(float) * (size_t)head_size);
  float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
  for (int s = 0; s < num_seqs; ++s) {
    const int seq_len = seq_lens[s];
    const int32_t* table = block_tables + (int64_t)s * max_num_blocks_per_seq;
    for (int h = 0; h < num_heads; ++h) {
      const int kvh = h / qpk;
      const float slope = alibi_slopes ? alibi_slopes[h] : 0.f;
      for (int d = 0; d < head_size; ++d)
        qf[d] = load_elem(query, s * q_stride + (int64_t)h * head_size + d, dtype);
      float m = -3.402823466e+38f;
      for (int t = 0; t < seq_len; ++t) {
        const int64_t blk = table[t / block_size];
        const int off = t % block_size;
        const int64_t base = blk * kv_block_stride + (int64_t)kvh * kv_head_stride;
        float dot = 0.f;
        for (int d = 0; d < head_size; ++d) {
          const int64_t ki = base + (int64_t)(d / x) * block_size * x + (int64_t)off * x + d % x;
          dot += qf[d] * load_elem(key_cache, ki, dtype);
        }
        float l = scale * dot;
        if (slope != 0.f) l += slope * (float)(t - seq_len + 1);
        logits[t] = l;
        if (l > m) m = l;
      }
      float sum = 0.f;
      for (int t = 0; t < seq_len; ++t) {
        logits[t] = expf(logits[t] - m);
        sum += logits[t];
      }
      const float inv = 1.f / (sum + 1e-6f);
      for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
      for (int t = 0; t < seq_len; ++t) {
        const float p = round_to(logits[t] * inv, dtype);
        const int64_t blk = table[t / block_size];
        const int off = t % block_size;
        const int64_t base = blk * kv_block_stride + (int64_t)kvh * kv_head_stride;
        for (int d = 0; d < head_size; ++d)
          acc[d] += p * load_elem(value_cache, base + (int64_t)d * block_size + off, dtype);
      }
      for (int d = 0; d < head_size; ++d)
        store_elem(out, ((int64_t)s * num_heads + h) * head_size + d, acc[d], dtype);
    }
  }
  free(logits);
  free(qf);
  free(acc);
}

/*
 * Block score ("value_l2").  Restates KVCachePruner.get_block_score,
 * vllm/attention/kvcache_prunner.py:51-56:
 *     torch.norm(block_values, p=2, dim=-1).mean(dim=1).sum(dim=0)
 * on a [block_size, num_kv_heads, head_size] view.  torch evaluates each
 * reduction in f32 and rounds its result to the tensor dtype, so the restatement
 * rounds after the norm, after the head mean and after the token sum.
 *
 * layout == 0: `values` is token-major [block_size, num_kv_heads, head_size]
 *              (prompt-time call site, torch_cuda_sdpa.py:1281-1287).
 * layout == 1: `values` is one block of the paged value cache
 *              [num_kv_heads, head_size, block_size]; the score is of the logical
 *              V[token, head, :] rows (the intended decode-time score).
 * layout == 2: `values` is one block of the paged value cache read the way the
 *              reference's decode call site views it (torch_cuda_sdpa.py:1521-1524):
 *              the flat block re-viewed as [-1, num_kv_heads, head_size].
 */
float orc_block_score(const void* values, int block_size, int num_kv_heads,
                      int head_size, int layout, int dtype) {
  float tok_sum = 0.f;
  for (int t = 0; t < block_size; ++t) {
    float head_sum = 0.f;
    for (int h = 0; h < num_kv_heads; ++h) {
      float ss = 0.f;
      for (int d = 0; d < head_size; ++d) {
        int64_t idx;
        if (layout == 1)
          idx = ((int64_t)h * head_size + d) * block_size + t;
        else
          idx = ((int64_t)t * num_kv_heads + h) * head_size + d;
        const float v = load_elem(values, idx, dtype);
        ss += v * v;
      }
      head_sum += round_to(sqrtf(ss), dtype);
    }
    tok_sum += round_to(head_sum / (float)num_kv_heads, dtype);
  }
  return round_to(tok_sum, dtype);
}

/*
 * Rotary position embedding, csrc/pos_encoding_kernels.cu:10-35 (apply_token_rotary_embedding)
 * and :38-71 (all heads of query and key of one token).  The reference computes in the storage
 * type: x*cos, y*sin and their difference / sum are each rounded to it.  rows [num_tokens][heads]
 * [head_size] with row stride `stride` (elements); cos_sin [max_position][rot_dim].
 */
void orc_rotary_embedding(const int64_t* positions, void* rows, const void* cos_sin, int num_tokens,
                          int num_heads, int head_size, int64_t stride, int rot_dim, int is_neox,
                          int dtype) {
  const int embed = rot_dim / 2;
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t cbase = positions[t] * rot_dim;
    for (int h = 0; h < num_heads; ++h) {
      const int64_t base = (int64_t)t * stride + (int64_t)h * head_size;
      for (int i = 0; i < embed; ++i) {
        const int xi = is_neox ? i : 2 * i;
        const int yi = is_neox ? embed + i : 2 * i + 1;
        const float c = load_elem(cos_sin, cbase + i, dtype);
        const float s = load_elem(cos_sin, cbase + embed + i, dtype);
        const float x = load_elem(rows, base + xi, dtype);
        const float y = load_elem(rows, base + yi, dtype);
        volatile float xc = round_to(x * c, dtype), ys = round_to(y * s, dtype);
        volatile float yc = round_to(y * c, dtype), xs = round_to(x * s, dtype);
        store_elem(rows, base + xi, xc - ys, dtype);
        store_elem(rows, base + yi, yc + xs, dtype);
      }
    }
  }
}

/*
 * The same score split at the head mean, for caches whose KV heads are spread over
 * tensor-parallel ranks (each rank passes its own heads): orc_block_token_norm_sums gives,
 * per token, the f32 sum over the given heads of the rounded norms; the ranks' arrays are
 * added; orc_block_score_from_token_sums applies the mean over ALL heads, its rounding, the
 * token sum and its rounding (kvcache_prunner.py:51-56 restated around an all-reduce).
 */
void orc_block_token_norm_sums(const void* values, float* sums, int block_size, int num_kv_heads,
                               int head_size, int layout, int dtype) {
  for (int t = 0; t < block_size; ++t) {
    float head_sum = 0.f;
    for (int h = 0; h < num_kv_heads; ++h) {
      float ss = 0.f;
      for (int d = 0; d < head_size; ++d) {
        int64_t idx;
        if (layout == 1)
          idx = ((int64_t)h * head_size + d) * block_size + t;
        else
          idx = ((int64_t)t * num_kv_heads + h) * head_size + d;
        const float v = load_elem(values, idx, dtype);
        ss += v * v;
      }
      head_sum += round_to(sqrtf(ss), dtype);
    }
    sums[t] = head_sum;
  }
}

float orc_block_score_from_token_sums(const float* sums, int block_size, int total_kv_heads,
                                      int dtype) {
  float tok_sum = 0.f;
  for (int t = 0; t < block_size; ++t)
    tok_sum += round_to(sums[t] / (float)total_kv_heads, dtype);
  return round_to(tok_sum, dtype);
}

/*
 * Block-table compaction, intended semantics of
 * TorchCUDASDPABackendImpl._create_updated_block_tables,
 * vllm/attention/backends/torch_cuda_sdpa.py:1562-1586: row b loses column
 * rmv_idx[b] (rmv_idx[b] < 0: nothing removed), the rest shifts left.  The reference's
 * output width is max_blocks-1, which silently drops the last block of a row that
 * evicts nothing but is max_blocks long; out_width lets the caller keep such rows
 * whole (out_width = longest surviving row; entries past a row's end are 0).
 * compat_q1 != 0 reproduces the reference's loop exactly, including the slice
 * taken inside the loop (:1583), which for rows b >= 1 of a full-width table
 * drops the last block and repeats the one before it.
 */
void orc_compact_block_tables(const int32_t* tables, const int32_t* rmv_idx,
                              int32_t* out, int batch, int max_blocks,
                              int out_width, int compat_q1) {
  if (!compat_q1) {
    for (int b = 0; b < batch; ++b) {
      const int32_t* row = tables + (int64_t)b * max_blocks;
      int32_t* o = out + (int64_t)b * out_width;
      const int r = rmv_idx[b];
      for (int j = 0; j < out_width; ++j) {
        const int src = (r >= 0 && j >= r) ? j + 1 : j;
        o[j] = src < max_blocks ? row[src] : 0;
      }
    }
    return;
  }
  const int w = max_blocks - 1; /* the reference's shape; out_width must equal it */
  if (w <= 0 || out_width != w) return;
  /* literal walk of the reference loop on a working copy whose width shrinks
     after the first iteration */
  int32_t* work = (int32_t*)malloc(sizeof(int32_t) * (size_t)batch * (size_t)max_blocks);
  memcpy(work, tables, sizeof(int32_t) * (size_t)batch * (size_t)max_blocks);
  int cur_w = max_blocks; /* logical width of `work` rows (row pitch stays max_blocks) */
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_blocks);
  for (int b = 0; b < batch; ++b) {
    int32_t* row = work + (int64_t)b * max_blocks;
    int n;
    const int r = rmv_idx[b];
    if (r >= 0) {
      n = 0;
      for (int j = 0; j < cur_w; ++j)
        if (j != r) tmp[n++] = row[j];
    } else {
      n = cur_w < w ? cur_w : w;
      for (int j = 0; j < n; ++j) tmp[j] = row[j];
    }
    for (int j = 0; j < n; ++j) row[j] = tmp[j];
    cur_w = w;
  }
  for (int b = 0; b < batch; ++b)
    memcpy(out + (int64_t)b * w, work + (int64_t)b * max_blocks, sizeof(int32_t) * (size_t)w);
  free(work);
  free(tmp);
}

/*
 * Varlen causal prefill attention with the reference's mask alignment.  Restates
 * TorchCUDASDPABackendImpl._run_sdpa_forward, torch_cuda_sdpa.py:1069-1127: per
 * request, scaled_dot_product_attention(q[Lq], k[Lkv], v[Lkv], is_causal=True)
 * with GQA by repeating KV heads (:1078-1080).  With is_causal=True and
 * Lq != Lkv PyTorch applies a top-left aligned lower-triangular mask: query i
 * attends kv rows 0..min(i, Lkv-1).  Arithmetic: logits, softmax and the P.V sum
 * in f32; output rounded to the storage type.  (The reference's fused SDPA kernel
 * is a third-party PyTorch kernel whose internal rounding is not pinned by any
 * reference test: parity for prefill is tolerance-based.)
 */
void orc_prefill_attention(void* out, const void* query, const void* key,
                           const void* value, const int32_t* cu_q,
                           const int32_t* cu_kv, int num_seqs, int num_heads,
                           int num_kv_heads, int head_size, float scale,
                           int dtype) {
  const int qpk = num_heads / num_kv_heads;
  int max_kv = 1;
  for (int s = 0; s < num_seqs; ++s)
    if (cu_kv[s + 1] - cu_kv[s] > max_kv) max_kv = cu_kv[s + 1] - cu_kv[s];
  float* logits = (float*)malloc(sizeof(float) * (size_t)max_kv);
  float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
  for (int s = 0; s < num_seqs; ++s) {
    const int lq = cu_q[s + 1] - cu_q[s], lkv = cu_kv[s + 1] - cu_kv[s];
    for (int h = 0; h < num_heads; ++h) {
      const int kvh = h / qpk;
      for (int i = 0; i < lq; ++i) {
        const int64_t qb = ((int64_t)(cu_q[s] + i) * num_heads + h) * head_size;
        int n = i + 1 < lkv ? i + 1 : lkv;
        float m = -3.402823466e+38f;
        for (int t = 0; t < n; ++t) {
          const int64_t kb = ((int64_t)(cu_kv[s] + t) * num_kv_heads + kvh) * head_size;
          float dot = 0.f;
          for (int d = 0; d < head_size; ++d)
            dot += load_elem(query, qb + d, dtype) * load_elem(key, kb + d, dtype);
          logits[t] = dot * scale;
          if (logits[t] > m) m = logits[t];
        }
        float sum = 0.f;
        for (int t = 0; t < n; ++t) {
          logits[t] = expf(logits[t] - m);
          sum += logits[t];
        }
        for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
        for (int t = 0; t < n; ++t) {
          const int64_t vb = ((int64_t)(cu_kv[s] + t) * num_kv_heads + kvh) * head_size;
          const float p = logits[t] / sum;
          for (int d = 0; d < head_size; ++d) acc[d] += p * load_elem(value, vb + d, dtype);
        }
        for (int d = 0; d < head_size; ++d)
          store_elem(out, qb + d, n > 0 ? acc[d] : 0.f, dtype);
      }
    }
  }
  free(logits);
  free(acc);
}
