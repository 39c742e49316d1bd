// Token-level prompt prune ("local" / "global": score ||V||_2 or another formula of
// pe_score_formula, "inverse_key_l2": 1/(||K||_2 + 1e-8)) on gfx950: per request and per KV head, drop the n_prune lowest-score
// tokens of the middle slice and emit the kept-row index used by the gather kernels.
//
// Three small launches per call (prefill-time, cold path):
//   1. token_score_kernel      one lane per (token, head): f32 score, rounded to the storage
//                              type exactly where torch rounds (after norm, +eps, reciprocal)
//   2. token_select_kernel     one workgroup per (head, request): 4-pass 8-bit radix select of
//                              the n_prune-th smallest score (scores are >= 0, so their f32 bit
//                              patterns order like the values); ties at the threshold are
//                              dropped lowest token first; then an ordered compaction writes
//                              the per-head kept index (or keep flags in compat mode)
//   3. token_index_compat_kernel  (compat only) row-major (token, head) repacking of the kept
//                              pairs, the effect of the reference's `middle[mask].view(-1,H,D)`
#include "pe_common.h"
#include "score_formulas.h"

namespace pe {

constexpr int SEL_THREADS = 256;

// sortable != 0: the score is stored as an unsigned key with the order of the values (what
// token_select_kernel's radix select walks; the cosine formula is signed); 0: the f32 score itself.
template <typename T>
__global__ void token_score_kernel(float* __restrict__ scores, const T* __restrict__ key,
                                   const T* __restrict__ value, int64_t total, int64_t key_stride,
                                   int64_t value_stride, int num_heads, int head_size, int method,
                                   int sortable) {
  constexpr int X = 16 / sizeof(T);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t t = i / num_heads;
  const int h = (int)(i % num_heads);
  // formula: the method's own (inverse_key_l2), else the one carried in bits 8..15
  const int formula = (method & 0xff) == PE_EVICT_INVERSE_KEY_L2
                          ? PE_FORMULA_INVERSE_KEY_L2
                          : (method >> PE_SCORE_FORMULA_SHIFT) & 0xff;
  auto row_norm = [&](const T* p) {  // ||row||_2 rounded to the storage type, ascending d
    float ss = 0.f;
    for (int c = 0; c < head_size / X; ++c) {
      union {
        u32x4 v;
        T e[X];
      } u;
      u.v = *reinterpret_cast<const u32x4*>(p + c * X);
#pragma unroll
      for (int e = 0; e < X; ++e) {
        const float x = to_f32(u.e[e]);
        ss += x * x;
      }
    }
    return round_to<T>(sqrtf(ss));
  };
  const T* kp = key + t * key_stride + (int64_t)h * head_size;
  const T* vp = value + t * value_stride + (int64_t)h * head_size;
  float s;
  if (formula == PE_FORMULA_VALUE_L2) {
    s = row_norm(vp);
  } else if (formula == PE_FORMULA_KEY_L2) {
    s = row_norm(kp);
  } else if (formula == PE_FORMULA_VALUE_L2_PLUS_KEY_L2) {
    s = round_to<T>(row_norm(kp) + row_norm(vp));
  } else if (formula == PE_FORMULA_INVERSE_KEY_L2) {
    s = round_to<T>(row_norm(kp) + 1e-8f);
    s = round_to<T>(1.0f / s);
  } else {
    // the README-era formulas (cold path: element-wise reads)
    s = formula_score<T>(formula, head_size, (int)t + 1, [&](int d) { return to_f32(kp[d]); },
                         [&](int d) { return to_f32(vp[d]); });
  }
  scores[i] = sortable ? __uint_as_float(sortable_bits(s)) : s;
}

// exclusive prefix count of `flag` over the 256 threads of the workgroup; *total gets the sum.
__device__ __forceinline__ int block_excl_count(bool flag, int* wave_tot /* LDS[4] */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  const int before = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();  // protect wave_tot from the previous use
  if (lane == 0) wave_tot[wave] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int c = wave_tot[w];
    if (w < wave) off += c;
    tot += c;
  }
  *total = tot;
  return off + before;
}

struct SliceGeom {
  int q0, L, k0, mid, n_prune, end_mid;
};

// budget > 0: the budget prune type (first block, budgeted middle, last block + remainder).
// budget == 0: the percentage prune type - the last tail_blocks blocks (+ remainder) stay and the
// number of pruned middle tokens is the request's length minus its kept length
// (kvcache_prunner_old.py:203-233).
__device__ __forceinline__ SliceGeom slice_geom(const int32_t* cu_q, const int32_t* cu_kept, int r,
                                                int budget, int bs, int tail_blocks) {
  SliceGeom g;
  g.q0 = cu_q[r];
  g.L = cu_q[r + 1] - g.q0;
  g.k0 = cu_kept[r];
  g.mid = 0;
  g.n_prune = 0;
  g.end_mid = g.L;
  if (budget == 0) {
    const int n_prune = g.L - (cu_kept[r + 1] - g.k0);
    if (n_prune > 0) {
      g.end_mid = g.L - tail_blocks * bs - g.L % bs;
      g.mid = g.end_mid - bs;
      g.n_prune = n_prune;
    }
  } else if (g.L > budget) {
    const int rem = g.L % bs;
    g.end_mid = g.L - bs - rem;
    g.mid = g.end_mid - bs;
    g.n_prune = g.mid - (budget - 2 * bs);
    if (g.n_prune < 0) g.n_prune = 0;
  }
  return g;
}

__global__ __launch_bounds__(SEL_THREADS) void token_select_kernel(
    int32_t* __restrict__ src_index, uint8_t* __restrict__ keep_flags,
    const float* __restrict__ scores, const int32_t* __restrict__ cu_q,
    const int32_t* __restrict__ cu_kept, int budget, int bs, int num_heads, int compat_q2,
    int tail_blocks) {
  __shared__ int hist[256];
  __shared__ int wave_tot[4];
  __shared__ int sh_bin, sh_k;
  const int h = blockIdx.x, r = blockIdx.y;
  const int tid = threadIdx.x;
  const SliceGeom g = slice_geom(cu_q, cu_kept, r, budget, bs, tail_blocks);
  const int H = num_heads;

  if (g.n_prune == 0) {
    // nothing pruned (short prompt, or the middle already fits): identity index
    if (!compat_q2) {
      for (int n = tid; n < g.L; n += SEL_THREADS) src_index[(int64_t)(g.k0 + n) * H + h] = g.q0 + n;
    } else {
      for (int n = tid; n < g.L; n += SEL_THREADS)
        src_index[(int64_t)(g.k0 + n) * H + h] = (g.q0 + n) * H + h;
    }
    return;
  }
  const float* sc = scores + (int64_t)(g.q0 + bs) * H + h;  // middle token i -> sc[i*H]

  // ---- radix select: value of the n_prune-th smallest score and how many ties to drop ----
  uint32_t prefix = 0, mask = 0;
  int k = g.n_prune;  // rank (1-based) searched among the elements matching prefix
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < g.mid; i += SEL_THREADS) {
      const uint32_t u = __float_as_uint(sc[(int64_t)i * H]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] >= k) break;
        cum += hist[b];
      }
      sh_bin = b;
      sh_k = k - cum;
    }
    __syncthreads();
    prefix |= (uint32_t)sh_bin << shift;
    mask |= 255u << shift;
    k = sh_k;
    __syncthreads();
  }
  const uint32_t thr = prefix;  // exact bit pattern of the threshold score
  const int ties_to_drop = k;   // among scores == thr, drop the first `k` by token index

  // ---- ordered compaction ------------------------------------------------------------------
  int tie_base = 0, keep_base = 0;
  for (int base = 0; base < g.mid; base += SEL_THREADS) {
    const int i = base + tid;
    uint32_t u = 0xffffffffu;
    if (i < g.mid) u = __float_as_uint(sc[(int64_t)i * H]);
    const bool is_tie = i < g.mid && u == thr;
    int tie_tot;
    const int tie_rank = tie_base + block_excl_count(is_tie, wave_tot, &tie_tot);
    const bool keep = i < g.mid && (u > thr || (is_tie && tie_rank >= ties_to_drop));
    int keep_tot;
    const int keep_rank = keep_base + block_excl_count(keep, wave_tot, &keep_tot);
    if (i < g.mid) {
      if (compat_q2) {
        keep_flags[(int64_t)(g.q0 + bs + i) * H + h] = keep ? 1 : 0;
      } else if (keep) {
        src_index[(int64_t)(g.k0 + bs + keep_rank) * H + h] = g.q0 + bs + i;
      }
    }
    tie_base += tie_tot;
    keep_base += keep_tot;
  }
  if (!compat_q2) {
    // first block and tail are kept whole
    const int n_keep_mid = g.mid - g.n_prune;
    for (int n = tid; n < bs; n += SEL_THREADS) src_index[(int64_t)(g.k0 + n) * H + h] = g.q0 + n;
    const int tail = g.L - g.end_mid;
    for (int n = tid; n < tail; n += SEL_THREADS)
      src_index[(int64_t)(g.k0 + bs + n_keep_mid + n) * H + h] = g.q0 + g.end_mid + n;
  }
}

// compat: kept (token, head) pairs of the middle slice re-packed in row-major order
__global__ __launch_bounds__(SEL_THREADS) void token_index_compat_kernel(
    int32_t* __restrict__ src_index, const uint8_t* __restrict__ keep_flags,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_kept, int budget, int bs,
    int num_heads, int tail_blocks) {
  __shared__ int wave_sum_s[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int H = num_heads;
  const SliceGeom g = slice_geom(cu_q, cu_kept, r, budget, bs, tail_blocks);
  if (g.n_prune == 0) return;  // identity already written by token_select_kernel
  const int n_keep_mid = g.mid - g.n_prune;
  for (int n = tid; n < bs * H; n += SEL_THREADS)
    src_index[(int64_t)g.k0 * H + n] = g.q0 * H + n;
  const int tail = g.L - g.end_mid;
  for (int n = tid; n < tail * H; n += SEL_THREADS)
    src_index[(int64_t)(g.k0 + bs + n_keep_mid) * H + n] = (g.q0 + g.end_mid) * H + n;
  int base_rank = 0;
  for (int base = 0; base < g.mid; base += SEL_THREADS) {
    const int i = base + tid;
    int cnt = 0;
    const uint8_t* f = keep_flags + (int64_t)(g.q0 + bs + i) * H;
    if (i < g.mid)
      for (int h = 0; h < H; ++h) cnt += f[h];
    // block exclusive scan of cnt
    const int lane = tid & 63, wave = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wave_sum_s[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) off += wave_sum_s[w];
      tot += wave_sum_s[w];
    }
    int rank = base_rank + off + incl - cnt;
    if (i < g.mid) {
      for (int h = 0; h < H; ++h) {
        if (f[h]) {
          src_index[(int64_t)(g.k0 + bs) * H + rank] = (g.q0 + bs + i) * H + h;
          ++rank;
        }
      }
    }
    base_rank += tot;
  }
}

}  // namespace pe

extern "C" {

int64_t pe_prompt_token_select_workspace_bytes(int64_t total_q, int num_kv_heads) {
  if (total_q <= 0 || num_kv_heads <= 0) return 0;
  const int64_t n = total_q * num_kv_heads;
  return ((n * 4 + 255) / 256) * 256 + n;  // f32 scores, then one keep flag per (token, head)
}

int pe_prompt_token_select(int32_t* src_index, const void* key, const void* value,
                           const int32_t* cu_q, const int32_t* cu_kept, int num_seqs,
                           int64_t total_q, int cache_budget, int block_size, int method,
                           int compat_q2, int tail_blocks,
                           int64_t key_stride, int64_t value_stride, int num_kv_heads,
                           int head_size, int dtype, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  using namespace pe;
  PE_REQUIRE(num_seqs >= 0, PE_ERR_INVALID_ARG, "negative num_seqs");
  if (num_seqs == 0) return PE_OK;
  PE_REQUIRE(src_index && key && value && cu_q && cu_kept, PE_ERR_INVALID_ARG,
             "prompt_token_select: null pointer");
  {
    const int base = method & 0xff, formula = (method >> PE_SCORE_FORMULA_SHIFT) & 0xff;
    PE_REQUIRE((base == PE_EVICT_LOCAL || base == PE_EVICT_GLOBAL || base == PE_EVICT_INVERSE_KEY_L2) &&
                   (method >> 16) == 0,
               PE_ERR_INVALID_ARG, "token-level prune is defined for local/global/inverse_key_l2");
    PE_REQUIRE(formula_known(formula), PE_ERR_INVALID_ARG, "unknown score formula %d", formula);
  }
  PE_REQUIRE(block_size > 0 && (cache_budget >= 3 * block_size || (cache_budget == 0 && tail_blocks >= 1)),
             PE_ERR_INVALID_ARG,
             "cache budget must be at least 3 blocks (kvcache_prunner.py:15), or 0 with tail_blocks >= 1");
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(head_size % x == 0 && key_stride % x == 0 && value_stride % x == 0,
             PE_ERR_UNSUPPORTED, "rows must be made of 16-byte chunks");
  PE_REQUIRE(num_seqs <= 65535 && num_kv_heads <= 65535, PE_ERR_UNSUPPORTED, "grid too large");
  PE_REQUIRE(total_q > 0, PE_ERR_INVALID_ARG, "prompt_token_select: total_q must be positive");
  const int64_t n_pairs = total_q * num_kv_heads;
  PE_REQUIRE(workspace && workspace_bytes >= pe_prompt_token_select_workspace_bytes(total_q, num_kv_heads),
             PE_ERR_WORKSPACE, "prompt_token_select: workspace too small");
  float* scores = static_cast<float*>(workspace);
  uint8_t* flags = static_cast<uint8_t*>(workspace) + ((n_pairs * 4 + 255) / 256) * 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t total = total_q * num_kv_heads;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(token_score_kernel<T>, dim3((unsigned)((total + 255) / 256)),
                                     dim3(256), 0, st, scores, (const T*)key, (const T*)value, total,
                                     key_stride, value_stride, num_kv_heads, head_size, method, 1));
  if (int rc = check_launch("token_score")) return rc;
  hipLaunchKernelGGL(token_select_kernel, dim3((unsigned)num_kv_heads, (unsigned)num_seqs),
                     dim3(SEL_THREADS), 0, st, src_index, flags, scores, cu_q, cu_kept, cache_budget,
                     block_size, num_kv_heads, compat_q2, tail_blocks);
  if (int rc = check_launch("token_select")) return rc;
  if (compat_q2) {
    hipLaunchKernelGGL(token_index_compat_kernel, dim3((unsigned)num_seqs), dim3(SEL_THREADS), 0, st,
                       src_index, flags, cu_q, cu_kept, cache_budget, block_size, num_kv_heads,
                       tail_blocks);
    if (int rc = check_launch("token_index_compat")) return rc;
  }
  return PE_OK;
}

int pe_token_scores(float* scores_out, const void* key, const void* value, int64_t num_tokens,
                    int64_t key_stride, int64_t value_stride, int num_kv_heads, int head_size,
                    int formula, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_tokens >= 0, PE_ERR_INVALID_ARG, "negative num_tokens");
  if (num_tokens == 0) return PE_OK;
  PE_REQUIRE(scores_out && key && value, PE_ERR_INVALID_ARG, "token_scores: null pointer");
  PE_REQUIRE(formula_known(formula), PE_ERR_INVALID_ARG, "unknown score formula %d", formula);
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && head_size % x == 0 && key_stride % x == 0 &&
                 value_stride % x == 0,
             PE_ERR_UNSUPPORTED, "token_scores: rows must be made of 16-byte chunks");
  const int64_t total = num_tokens * num_kv_heads;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(token_score_kernel<T>, dim3((unsigned)((total + 255) / 256)),
                                     dim3(256), 0, static_cast<hipStream_t>(stream), scores_out,
                                     (const T*)key, (const T*)value, total, key_stride, value_stride,
                                     num_kv_heads, head_size,
                                     PE_EVICT_GLOBAL | (formula << PE_SCORE_FORMULA_SHIFT), 0));
  return check_launch("token_scores");
}

}  // extern "C"
