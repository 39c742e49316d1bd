// Eviction scoring and the per-step eviction decision on gfx950:
//   pe_block_score_value_l2   per-block value_l2 score
//   pe_block_token_norm_sums / pe_block_score_from_token_sums   the same score in two steps
//                             around an all-reduce, for head-sharded caches
//   pe_evict_select_compact   trigger + victim choice + score bookkeeping + block-table
//                             compaction (+ streamingLLM-1 zero-slot ids), one launch/step
// Integer / small-FP work: one 64-lane wave per block or per sequence; the byte traffic is
// one 32 KiB value block per scored block plus the block tables.
#include <cstdlib>

#include "pe_common.h"

namespace pe {

constexpr int SCORE_MAX_PAIRS = 64 * 64;  // block_size * num_kv_heads held in LDS
constexpr int SCORE_THREADS = 256;         // workgroup of the scoring kernels
constexpr int SCORE_STAGE_BYTES = 40 * 1024;  // LDS staging area for paged value slabs (8 padded 128 x 16 slabs of 16-bit values in one pass)

// Set when a sequence's score row is full and a new score could not be appended (the caller
// sized the table too narrow); read back through pe_device_diagnostics.
static __device__ unsigned int evict_diag_word;

// LDS the scoring code needs: norms[bs*H] + means[bs] (f32) + the staging area of the paged
// layout (whole head slabs, each padded by one row of block_size elements).
__host__ __device__ inline int score_heads_per_pass(int num_kv_heads, int head_size, int block_size,
                                                    int elem) {
  const int slab = (head_size * block_size + block_size) * elem;
  int g = SCORE_STAGE_BYTES / slab;
  if (g > num_kv_heads) g = num_kv_heads;
  return g;  // 0: a slab does not fit, the strided path is used
}
inline size_t score_lds_bytes(int layout, int num_kv_heads, int head_size, int block_size, int elem) {
  size_t n = (size_t)(block_size * num_kv_heads + block_size) * 4;
  n = (n + 15) & ~(size_t)15;
  if (layout == PE_SCORE_CACHE) {
    const int g = score_heads_per_pass(num_kv_heads, head_size, block_size, elem);
    n += (size_t)g * (head_size * block_size + block_size) * elem;
  }
  return n;
}

// norms[t*H + h] = ||V[t, h, :]||_2 rounded to the storage type, for one block, computed by the
// whole workgroup.  The sum of squares of a (token, head) pair runs over d in ascending order
// in one thread (products of 16-bit values are exact in f32, so only this order matters) -
// that is torch's result for norm(V, dim=-1) on a low-precision tensor.
//   PE_SCORE_CACHE: the block is [H][D][bs]; whole head slabs are copied to LDS with 16-byte
//   loads (a slab is contiguous, so the copy is fully coalesced) and each thread then walks
//   the d axis of its pair in LDS - consecutive threads hold consecutive tokens, the slabs are
//   padded by one row so that the heads of a wave fall into different banks.
//   dense rows (token-major, or the flat re-view of a cache block): 16-byte loads along the row.
template <typename T>
__device__ void block_pair_norms(const T* __restrict__ base, int layout, int num_kv_heads,
                                 int head_size, int block_size, float* norms, char* stage_raw) {
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int pairs = block_size * num_kv_heads;
  constexpr int X = 16 / sizeof(T);
  if (layout == PE_SCORE_CACHE) {
    const int slab = head_size * block_size;  // elements of one head
    const int g = score_heads_per_pass(num_kv_heads, head_size, block_size, (int)sizeof(T));
    const bool vec_ok = g > 0 && slab % X == 0 && block_size % X == 0 && ((uintptr_t)base % 16) == 0;
    if (vec_ok) {
      T* stage = reinterpret_cast<T*>(stage_raw);
      const int lds_slab = slab + block_size;
      const int chunks_per_head = slab / X;
      for (int h0 = 0; h0 < num_kv_heads; h0 += g) {
        const int nh = min(g, num_kv_heads - h0);
        for (int c = tid; c < nh * chunks_per_head; c += nthr) {
          const int hh = c / chunks_per_head, w = c % chunks_per_head;
          *reinterpret_cast<u32x4*>(stage + hh * lds_slab + w * X) =
              *reinterpret_cast<const u32x4*>(base + (int64_t)(h0 + hh) * slab + w * X);
        }
        __syncthreads();
        for (int p = tid; p < nh * block_size; p += nthr) {
          const int hh = p / block_size, t = p % block_size;
          const T* v = stage + hh * lds_slab + t;
          float ss = 0.f;
          // eight LDS reads in flight, then their squares added in ascending d: the additions
          // form one dependent chain either way, the read latency is paid once per eight
          int d = 0;
          for (; d + 8 <= head_size; d += 8) {
            T x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = v[(d + e) * block_size];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = to_f32(x8[e]);
              ss += x * x;
            }
          }
          for (; d < head_size; ++d) {
            const float x = to_f32(v[d * block_size]);
            ss += x * x;
          }
          norms[t * num_kv_heads + h0 + hh] = round_to<T>(sqrtf(ss));
        }
        __syncthreads();
      }
      return;
    }
    for (int p = tid; p < pairs; p += nthr) {
      const int t = p / num_kv_heads, h = p % num_kv_heads;
      const T* v = base + (int64_t)h * slab + t;  // V[h][d][t]
      float ss = 0.f;
      for (int d = 0; d < head_size; ++d) {
        const float x = to_f32(v[(int64_t)d * block_size]);
        ss += x * x;
      }
      norms[p] = round_to<T>(sqrtf(ss));
    }
    __syncthreads();
    return;
  }
  if (layout == PE_SCORE_KEY_CACHE) {
    // paged key block [H][D/X][bs][X]: the X elements of a 16-byte chunk are consecutive d of one
    // token, consecutive threads take consecutive tokens (coalesced), d ascends with the chunk
    for (int p = tid; p < pairs; p += nthr) {
      const int h = p / block_size, t = p % block_size;
      const T* k = base + ((int64_t)h * (head_size / X) * block_size + t) * X;
      float ss = 0.f;
      for (int c = 0; c < head_size / X; ++c) {
        union {
          u32x4 raw;
          T e[X];
        } u;
        u.raw = *reinterpret_cast<const u32x4*>(k + (int64_t)c * block_size * X);
#pragma unroll
        for (int e = 0; e < X; ++e) {
          const float x = to_f32(u.e[e]);
          ss += x * x;
        }
      }
      norms[t * num_kv_heads + h] = round_to<T>(sqrtf(ss));
    }
    __syncthreads();
    return;
  }
  for (int p = tid; p < pairs; p += nthr) {
    const T* v = base + (int64_t)p * head_size;  // dense rows (token-major or flat re-view)
    float ss = 0.f;
    if (head_size % X == 0 && ((uintptr_t)v % 16) == 0) {
      // 16-byte loads, elements still accumulated in ascending order
      for (int d = 0; d < head_size; d += X) {
        union {
          u32x4 raw;
          T e[X];
        } u;
        u.raw = *reinterpret_cast<const u32x4*>(v + d);
#pragma unroll
        for (int e = 0; e < X; ++e) {
          const float x = to_f32(u.e[e]);
          ss += x * x;
        }
      }
    } else {
      for (int d = 0; d < head_size; ++d) {
        const float x = to_f32(v[d]);
        ss += x * x;
      }
    }
    norms[p] = round_to<T>(sqrtf(ss));
  }
  __syncthreads();
}

// Score of one block (every thread of the workgroup returns it).  Follows torch's evaluation of
//   norm(V, dim=-1).mean(dim=1).sum(dim=0)
// on a low-precision tensor: every reduction accumulates in f32 and its result is rounded to
// the storage type.
template <typename T>
__device__ float block_score_wg(const T* __restrict__ base, int layout, int num_kv_heads,
                                int head_size, int block_size, float* norms /* LDS [bs*H] */,
                                float* means /* LDS [bs] */, char* stage) {
  block_pair_norms<T>(base, layout, num_kv_heads, head_size, block_size, norms, stage);
  for (int t = threadIdx.x; t < block_size; t += blockDim.x) {
    float hs = 0.f;
    int h = 0;
    for (; h + 4 <= num_kv_heads; h += 4) {  // reads first, sums in ascending h
      float x4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) x4[e] = norms[t * num_kv_heads + h + e];
#pragma unroll
      for (int e = 0; e < 4; ++e) hs += x4[e];
    }
    for (; h < num_kv_heads; ++h) hs += norms[t * num_kv_heads + h];
    means[t] = round_to<T>(hs / (float)num_kv_heads);
  }
  __syncthreads();
  float total = 0.f;
  int t = 0;
  for (; t + 8 <= block_size; t += 8) {  // reads first, sums in ascending t
    float x8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x8[e] = means[t + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) total += x8[e];
  }
  for (; t < block_size; ++t) total += means[t];
  __syncthreads();  // norms/means may be reused by the caller's next block
  return round_to<T>(total);
}

__device__ inline char* score_stage_ptr(float* norms, int num_kv_heads, int block_size) {
  size_t n = (size_t)(block_size * num_kv_heads + block_size) * 4;
  n = (n + 15) & ~(size_t)15;
  return reinterpret_cast<char*>(norms) + n;
}

template <typename T>
__global__ __launch_bounds__(SCORE_THREADS) void block_score_kernel(
    float* __restrict__ scores, const T* __restrict__ values, const int32_t* __restrict__ block_ids,
    int layout, int64_t block_stride, int num_kv_heads, int head_size, int block_size) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* norms = reinterpret_cast<float*>(smem);
  float* means = norms + block_size * num_kv_heads;
  const int i = blockIdx.x;
  const int64_t blk = block_ids ? block_ids[i] : i;
  const float s = block_score_wg<T>(values + blk * block_stride, layout, num_kv_heads, head_size,
                                    block_size, norms, means,
                                    score_stage_ptr(norms, num_kv_heads, block_size));
  if (threadIdx.x == 0) scores[i] = s;
}

// Head-sharded scoring, step 1: per token, the sum over this rank's heads of the rounded norms.
template <typename T>
__global__ __launch_bounds__(SCORE_THREADS) void block_token_norm_sums_kernel(
    float* __restrict__ sums, const T* __restrict__ values, const int32_t* __restrict__ block_ids,
    int layout, int64_t block_stride, int num_kv_heads, int head_size, int block_size) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* norms = reinterpret_cast<float*>(smem);
  const int i = blockIdx.x;
  const int64_t blk = block_ids ? block_ids[i] : i;
  block_pair_norms<T>(values + blk * block_stride, layout, num_kv_heads, head_size, block_size,
                      norms, score_stage_ptr(norms, num_kv_heads, block_size));
  for (int t = threadIdx.x; t < block_size; t += blockDim.x) {
    float hs = 0.f;
    for (int h = 0; h < num_kv_heads; ++h) hs += norms[t * num_kv_heads + h];
    sums[(int64_t)i * block_size + t] = hs;
  }
}

// Step 2: head mean over all ranks' heads, rounded; token sum, rounded.
template <typename T>
__global__ __launch_bounds__(64) void block_score_from_sums_kernel(float* __restrict__ scores,
                                                                   const float* __restrict__ sums,
                                                                   int n_blocks, int block_size,
                                                                   int total_kv_heads) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_blocks) return;
  float total = 0.f;
  for (int t = 0; t < block_size; ++t)
    total += round_to<T>(sums[(int64_t)i * block_size + t] / (float)total_kv_heads);
  scores[i] = round_to<T>(total);
}

// One workgroup per sequence (grid = batch), or a single workgroup walking all rows serially
// when the reference's in-loop slice has to be reproduced (compat_q1).  All waves score the
// just-filled block together and copy the table; the selection over the score list is
// computed redundantly by every wave (same inputs, same result, so control flow stays uniform)
// and written by wave 0.
template <typename T>
__global__ __launch_bounds__(SCORE_THREADS) void evict_select_compact_kernel(
    int32_t* __restrict__ rmv_idx_out, int32_t* out_tables,
    int64_t* __restrict__ zero_slots_out, const int32_t* block_tables,
    const int32_t* __restrict__ seq_lens, const int32_t* num_blocks,
    const int32_t* __restrict__ victim_override, int batch, int max_blocks, int out_width,
    int method, int cache_budget, int block_size, int topk_blocks, int drop_victim_score, int compat_q1,
    int score_layout, float* __restrict__ score_table, int32_t* __restrict__ score_counts,
    const int32_t* __restrict__ score_rows, int score_table_width,
    const T* __restrict__ value_cache, int64_t kv_block_stride, int num_kv_heads, int head_size,
    const float* __restrict__ new_scores, int32_t* n_blocks_io, int32_t* kv_lens_io,
    int32_t* __restrict__ freed_out) {
  // n_blocks_io != nullptr selects the in-place form (device-resident tables): out_tables is
  // block_tables itself, the victim's column is closed inside the row, the row's block count
  // and physical length are updated and the victim's block id is reported in freed_out.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* norms = reinterpret_cast<float*>(smem);
  float* means = norms + block_size * num_kv_heads;
  char* stage = score_stage_ptr(norms, num_kv_heads, block_size);
  const int lane = threadIdx.x & 63;
  const bool writer = threadIdx.x < 64;  // wave 0
  const int b_begin = compat_q1 ? 0 : blockIdx.x;
  const int b_end = compat_q1 ? batch : blockIdx.x + 1;
  const int w = out_width;

  for (int b = b_begin; b < b_end; ++b) {
    // everything the decision may need is requested up front, so that the dependent chain
    // (length -> last block id -> value block -> score list) is as short as it can be
    const int seq_len = seq_lens[b];
    const int nblk_b = num_blocks[b];
    const int srow_b = score_rows ? score_rows[b] : 0;
    const int32_t* row = block_tables + (int64_t)b * max_blocks;
    const int last_blk_b = row[max(min(nblk_b, max_blocks) - 1, 0)];
    const int count_b = score_counts ? score_counts[srow_b] : 0;
    int rmv = -1;
    int64_t zslot = -1;
    const bool over = seq_len > cache_budget;
    const bool boundary = (seq_len % block_size) == 0;
    if (over && !boundary && method == PE_EVICT_STREAMING_1) {
      zslot = (int64_t)row[1] * block_size + (seq_len % block_size) - 1;
    }
    if (over && boundary && method != PE_EVICT_INVERSE_KEY_L2) {
      if (method == PE_EVICT_STREAMING || method == PE_EVICT_STREAMING_1) {
        rmv = 1;
      } else {
        const int ov = victim_override ? victim_override[b] : -1;
        if (score_table != nullptr) {
          // score the just-filled block and append it to this sequence's list
          const int srow = srow_b;
          float* list = score_table + (int64_t)srow * score_table_width;
          int n = count_b;
          float s;
          if (new_scores != nullptr) {
            s = new_scores[b];  // scored by the caller (head-sharded caches)
          } else {
            const int last_blk = last_blk_b;
            s = block_score_wg<T>(value_cache + (int64_t)last_blk * kv_block_stride, score_layout,
                                  num_kv_heads, head_size, block_size, norms, means, stage);
          }
          // the appended entry is kept in a register (every lane holds s): entries written
          // in this launch are never read back through memory
          int n_new = -1;
          if (n < score_table_width) {
            if (threadIdx.x == 0) list[n] = s;
            n_new = n;
            n += 1;
          } else if (threadIdx.x == 0) {
            atomicOr(&evict_diag_word, (unsigned)PE_DIAG_SCORE_ROW_FULL);
          }
          int start = 1, end = (method == PE_EVICT_LOCAL) ? 1 + topk_blocks : n - 1;
          if (end > n) end = n;
          // first arg-min over [start, end): strict '<' keeps the earliest minimum
          int best = start;
          if (ov >= 0) {
            best = ov;
          } else {
            float bv = (start < n) ? (start == n_new ? s : list[start]) : 0.f;
            int bi = start;
            for (int i = start + lane; i < end; i += 64) {
              const float v = (i == n_new) ? s : list[i];
              if (v < bv || (v == bv && i < bi)) {
                bv = v;
                bi = i;
              }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
              const float ov2 = __shfl_xor(bv, m);
              const int oi = __shfl_xor(bi, m);
              if (ov2 < bv || (ov2 == bv && oi < bi)) {
                bv = ov2;
                bi = oi;
              }
            }
            best = bi;
          }
          rmv = best;
          if (drop_victim_score && best < n) {
            // delete the victim's entry: shift the tail left by one (chunked so that reads of a
            // chunk complete before its writes)
            for (int base = best; base < n - 1; base += 64) {
              const int i = base + lane;
              float v = 0.f;
              if (i < n - 1) v = (i + 1 == n_new) ? s : list[i + 1];
              __syncthreads();
              if (writer && i < n - 1) list[i] = v;
              __syncthreads();
            }
            n -= 1;
          }
          if (threadIdx.x == 0) score_counts[srow] = n;
        } else {
          rmv = ov;  // host-chosen victim (random mode) without score bookkeeping
        }
      }
    }
    if (threadIdx.x == 0) {
      rmv_idx_out[b] = rmv;
      if (zero_slots_out) zero_slots_out[b] = zslot;
    }
    if (n_blocks_io != nullptr) {
      int32_t* r = out_tables + (int64_t)b * max_blocks;  // == row, writable
      const int n = num_blocks[b];
      if (rmv >= 0 && rmv < n) {
        const int32_t victim = r[rmv];
        // shift [rmv+1, n) left by one, a chunk at a time: all reads of a chunk complete before
        // its writes, and a later chunk only reads entries no earlier chunk has written
        for (int base = rmv; base < n - 1; base += blockDim.x) {
          const int j = base + threadIdx.x;
          int32_t v = 0;
          if (j < n - 1) v = r[j + 1];
          __syncthreads();
          if (j < n - 1) r[j] = v;
          __syncthreads();
        }
        if (threadIdx.x == 0) {
          r[n - 1] = 0;
          n_blocks_io[b] = n - 1;
          kv_lens_io[b] -= block_size;
          freed_out[b] = victim;
        }
      } else if (threadIdx.x == 0) {
        freed_out[b] = -1;
      }
    } else if (out_tables != nullptr && w > 0) {
      int32_t* o = out_tables + (int64_t)b * w;
      if (!compat_q1 || b == 0) {
        for (int j = threadIdx.x; j < w; j += blockDim.x) {
          const int src = (rmv >= 0 && j >= rmv) ? j + 1 : j;
          o[j] = src < max_blocks ? row[src] : 0;
        }
      } else {
        // rows after the first see a table already cut to width w by the reference's slice
        // inside its loop: removing a column leaves w-1 new entries and the old last one stays
        for (int j = threadIdx.x; j < w; j += blockDim.x) {
          int32_t v;
          if (rmv >= 0 && rmv < w) {
            v = (j < rmv) ? row[j] : (j < w - 1 ? row[j + 1] : row[w - 1]);
          } else {
            v = row[j];
          }
          o[j] = v;
        }
      }
    }
  }
}

// The same decision for the ordinary case (one workgroup per sequence, no reference-compatible
// in-loop slice) with the memory round trips taken out of the dependent chain: the table row and
// the lengths are requested together when the kernel starts, the score list and its count
// together with the value block, every thread keeps the entries it loaded in registers, and
// both shifts (score list, table row) are written from those registers - an entry is never read
// again after something was written.  FAST_SLOTS entries per thread bound the row widths
// this form takes; wider tables go through evict_select_compact_kernel.
constexpr int FAST_SLOTS = 8;
template <typename T>
__global__ __launch_bounds__(SCORE_THREADS) void evict_select_fast_kernel(
    int32_t* __restrict__ rmv_idx_out, int32_t* out_tables, int64_t* __restrict__ zero_slots_out,
    const int32_t* block_tables, const int32_t* __restrict__ seq_lens, const int32_t* num_blocks,
    const int32_t* __restrict__ victim_override, int max_blocks, int out_width, int method,
    int cache_budget, int block_size, int topk_blocks, int drop_victim_score, int score_layout,
    float* score_table, int32_t* score_counts, const int32_t* __restrict__ score_rows,
    int score_table_width, const T* __restrict__ value_cache, int64_t kv_block_stride,
    int num_kv_heads, int head_size, const float* __restrict__ new_scores, int32_t* n_blocks_io,
    int32_t* kv_lens_io, int32_t* __restrict__ freed_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* norms = reinterpret_cast<float*>(smem);
  float* means = norms + block_size * num_kv_heads;
  char* stage = score_stage_ptr(norms, num_kv_heads, block_size);
  __shared__ int sh_last;
  __shared__ float red_v[SCORE_THREADS / 64];
  __shared__ int red_i[SCORE_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;

  // ---- first round trip: lengths, row ------------------------------------------------------
  const int seq_len = seq_lens[b];
  const int nblk = num_blocks[b];
  const int srow = score_rows ? score_rows[b] : 0;
  const int ov = victim_override ? victim_override[b] : -1;
  const int32_t* row = block_tables + (int64_t)b * max_blocks;
  const int row1 = row[1];
  int32_t my_row[FAST_SLOTS];
#pragma unroll
  for (int k = 0; k < FAST_SLOTS; ++k) {
    const int j = tid + k * SCORE_THREADS;
    my_row[k] = j < max_blocks ? row[j] : 0;
  }

  int rmv = -1;
  int64_t zslot = -1;
  const bool over = seq_len > cache_budget;
  const bool boundary = (seq_len % block_size) == 0;
  if (over && !boundary && method == PE_EVICT_STREAMING_1)
    zslot = (int64_t)row1 * block_size + (seq_len % block_size) - 1;
  if (over && boundary && method != PE_EVICT_INVERSE_KEY_L2) {
    if (method == PE_EVICT_STREAMING || method == PE_EVICT_STREAMING_1) {
      rmv = 1;
    } else if (score_table == nullptr) {
      rmv = ov;  // host-chosen victim (random mode) without score bookkeeping
    } else {
      // ---- second round trip: score list and count, value block ----------------------------
      float* list = score_table + (int64_t)srow * score_table_width;
      const int count = score_counts[srow];
      float my_list[FAST_SLOTS];
#pragma unroll
      for (int k = 0; k < FAST_SLOTS; ++k) {
        const int i = tid + k * SCORE_THREADS;
        my_list[k] = i < score_table_width ? list[i] : 0.f;
      }
      float s;
      if (new_scores != nullptr) {
        s = new_scores[b];  // scored by the caller (head-sharded caches)
      } else {
        const int idx_last = max(min(nblk, max_blocks) - 1, 0);
#pragma unroll
        for (int k = 0; k < FAST_SLOTS; ++k)
          if (tid + k * SCORE_THREADS == idx_last) sh_last = my_row[k];
        __syncthreads();
        s = block_score_wg<T>(value_cache + (int64_t)sh_last * kv_block_stride, score_layout,
                              num_kv_heads, head_size, block_size, norms, means, stage);
      }
      int n = count, n_new = -1;
      if (n < score_table_width) {
        n_new = n;
        n += 1;
      } else if (tid == 0) {
        atomicOr(&evict_diag_word, (unsigned)PE_DIAG_SCORE_ROW_FULL);
      }
      const int start = 1;
      int end = (method == PE_EVICT_LOCAL) ? 1 + topk_blocks : n - 1;
      if (end > n) end = n;
      int best = start;
      if (ov >= 0) {
        best = ov;
      } else {
        // first arg-min over [start, end): strict '<' keeps the earliest minimum
        float bv = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < FAST_SLOTS; ++k) {
          const int i = tid + k * SCORE_THREADS;
          if (i >= start && i < end) {
            const float v = (i == n_new) ? s : my_list[k];
            if (bi == 0x7fffffff || v < bv) {  // a thread's own entries come in ascending i
              bv = v;
              bi = i;
            }
          }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          const float ov2 = __shfl_xor(bv, m);
          const int oi = __shfl_xor(bi, m);
          if (oi != 0x7fffffff && (bi == 0x7fffffff || ov2 < bv || (ov2 == bv && oi < bi))) {
            bv = ov2;
            bi = oi;
          }
        }
        if (lane == 0) {
          red_v[wave] = bv;
          red_i[wave] = bi;
        }
        __syncthreads();
        bv = red_v[0];
        bi = red_i[0];
#pragma unroll
        for (int w2 = 1; w2 < SCORE_THREADS / 64; ++w2) {
          const float ov2 = red_v[w2];
          const int oi = red_i[w2];
          if (oi != 0x7fffffff && (bi == 0x7fffffff || ov2 < bv || (ov2 == bv && oi < bi))) {
            bv = ov2;
            bi = oi;
          }
        }
        if (bi != 0x7fffffff) best = bi;
      }
      rmv = best;
      // every thread holds its list entries in registers before any of them is overwritten (with
      // a caller-chosen victim and caller-supplied scores no barrier lies on the way to here)
      __syncthreads();
      if (drop_victim_score && best < n) {
        // delete the victim's entry: every entry behind it moves one place to the front
#pragma unroll
        for (int k = 0; k < FAST_SLOTS; ++k) {
          const int i = tid + k * SCORE_THREADS;
          if (i > best && i < n) list[i - 1] = (i == n_new) ? s : my_list[k];
        }
        n -= 1;
      }
      if (tid == 0) {
        if (n_new >= 0) list[n_new] = s;  // the shift above never writes this slot
        score_counts[srow] = n;
      }
    }
  }
  if (tid == 0) {
    rmv_idx_out[b] = rmv;
    if (zero_slots_out) zero_slots_out[b] = zslot;
  }
  __syncthreads();  // every wave's row loads have completed before the row is written
  if (n_blocks_io != nullptr) {
    int32_t* r = out_tables + (int64_t)b * max_blocks;  // == row, writable
    const int n = min(nblk, max_blocks);
    if (rmv >= 0 && rmv < n) {
#pragma unroll
      for (int k = 0; k < FAST_SLOTS; ++k) {
        const int j = tid + k * SCORE_THREADS;
        if (j == rmv) freed_out[b] = my_row[k];
        if (j > rmv && j < n) r[j - 1] = my_row[k];
      }
      if (tid == 0) {
        r[n - 1] = 0;
        n_blocks_io[b] = nblk - 1;
        kv_lens_io[b] -= block_size;
      }
    } else if (tid == 0) {
      freed_out[b] = -1;
    }
  } else if (out_tables != nullptr && out_width > 0) {
    int32_t* o = out_tables + (int64_t)b * out_width;
#pragma unroll
    for (int k = 0; k < FAST_SLOTS; ++k) {
      const int j = tid + k * SCORE_THREADS;
      if (j < max_blocks && j != rmv) {
        const int dst = (rmv >= 0 && j > rmv) ? j - 1 : j;
        if (dst < out_width) o[dst] = my_row[k];
      }
    }
    // the column a removal frees at the end of a full-width output
    if (tid == 0 && rmv >= 0 && rmv < max_blocks && out_width == max_blocks) o[max_blocks - 1] = 0;
  }
}

// this file's diagnostic word (see pe_device_diagnostics)
int evict_diag_fetch(unsigned int* out, int reset) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(evict_diag_word));
  if (e == hipSuccess) e = hipMemcpy(out, p, sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset && *out != 0) e = hipMemset(p, 0, sizeof(unsigned int));
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "evict diagnostics: %s", hipGetErrorString(e));
  return PE_OK;
}

}  // namespace pe

extern "C" {

int pe_block_score_value_l2(float* scores_out, const void* values, const int32_t* block_ids,
                            int n_blocks, int layout, int64_t block_stride, int num_kv_heads,
                            int head_size, int block_size, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(n_blocks >= 0, PE_ERR_INVALID_ARG, "negative n_blocks");
  if (n_blocks == 0) return PE_OK;
  PE_REQUIRE(scores_out && values, PE_ERR_INVALID_ARG, "block_score: null pointer");
  PE_REQUIRE(layout >= 0 && layout <= PE_SCORE_KEY_CACHE, PE_ERR_INVALID_ARG, "block_score: bad layout %d",
             layout);
  PE_REQUIRE(layout != PE_SCORE_KEY_CACHE || head_size % (16 / elem_size(dtype)) == 0, PE_ERR_UNSUPPORTED,
             "block_score: key cache rows must be made of 16-byte chunks");
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && block_size > 0, PE_ERR_INVALID_ARG,
             "block_score: bad sizes");
  PE_REQUIRE(block_size * num_kv_heads <= SCORE_MAX_PAIRS, PE_ERR_UNSUPPORTED,
             "block_size*num_kv_heads too large for the score kernel");
  const size_t lds = score_lds_bytes(layout, num_kv_heads, head_size, block_size, elem_size(dtype));
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(block_score_kernel<T>, dim3((unsigned)n_blocks),
                                     dim3(SCORE_THREADS), lds,
                                     static_cast<hipStream_t>(stream), scores_out, (const T*)values,
                                     block_ids, layout, block_stride, num_kv_heads, head_size,
                                     block_size));
  return check_launch("block_score_value_l2");
}

int pe_block_token_norm_sums(float* sums_out, const void* values, const int32_t* block_ids,
                             int n_blocks, int layout, int64_t block_stride, int num_kv_heads,
                             int head_size, int block_size, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(n_blocks >= 0, PE_ERR_INVALID_ARG, "negative n_blocks");
  if (n_blocks == 0) return PE_OK;
  PE_REQUIRE(sums_out && values, PE_ERR_INVALID_ARG, "block_token_norm_sums: null pointer");
  PE_REQUIRE(layout == PE_SCORE_TOKEN_MAJOR || layout == PE_SCORE_CACHE || layout == PE_SCORE_KEY_CACHE,
             PE_ERR_INVALID_ARG, "block_token_norm_sums: layout %d is not head-separable", layout);
  PE_REQUIRE(layout != PE_SCORE_KEY_CACHE || head_size % (16 / elem_size(dtype)) == 0, PE_ERR_UNSUPPORTED,
             "block_token_norm_sums: key cache rows must be made of 16-byte chunks");
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && block_size > 0, PE_ERR_INVALID_ARG,
             "block_token_norm_sums: bad sizes");
  PE_REQUIRE(block_size * num_kv_heads <= SCORE_MAX_PAIRS, PE_ERR_UNSUPPORTED,
             "block_size*num_kv_heads too large for the score kernel");
  const size_t lds = score_lds_bytes(layout, num_kv_heads, head_size, block_size, elem_size(dtype));
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(block_token_norm_sums_kernel<T>, dim3((unsigned)n_blocks),
                                     dim3(SCORE_THREADS), lds, static_cast<hipStream_t>(stream), sums_out,
                                     (const T*)values, block_ids, layout, block_stride, num_kv_heads,
                                     head_size, block_size));
  return check_launch("block_token_norm_sums");
}

int pe_block_score_from_token_sums(float* scores_out, const float* sums, int n_blocks,
                                   int block_size, int total_kv_heads, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(n_blocks >= 0, PE_ERR_INVALID_ARG, "negative n_blocks");
  if (n_blocks == 0) return PE_OK;
  PE_REQUIRE(scores_out && sums, PE_ERR_INVALID_ARG, "block_score_from_token_sums: null pointer");
  PE_REQUIRE(block_size > 0 && total_kv_heads > 0, PE_ERR_INVALID_ARG,
             "block_score_from_token_sums: bad sizes");
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(block_score_from_sums_kernel<T>,
                                     dim3((unsigned)((n_blocks + 63) / 64)), dim3(64), 0,
                                     static_cast<hipStream_t>(stream), scores_out, sums, n_blocks,
                                     block_size, total_kv_heads));
  return check_launch("block_score_from_token_sums");
}

static int evict_select_impl(int32_t* rmv_idx_out, int32_t* out_tables, int64_t* zero_slots_out,
                             const int32_t* block_tables, const int32_t* seq_lens,
                             const int32_t* num_blocks, const int32_t* victim_override, int batch,
                             int max_blocks, int out_width, int method, int cache_budget,
                             int block_size, int topk_blocks, int drop_victim_score, int compat_q1,
                             int score_layout, float* score_table, int32_t* score_counts,
                             const int32_t* score_rows, int score_table_width,
                             const void* value_cache, int64_t kv_block_stride, int num_kv_heads,
                             int head_size, const float* new_scores, int dtype, void* stream,
                             int32_t* n_blocks_io, int32_t* kv_lens_io, int32_t* freed_out) {
  using namespace pe;
  PE_REQUIRE(batch >= 0, PE_ERR_INVALID_ARG, "negative batch");
  if (batch == 0) return PE_OK;
  PE_REQUIRE(rmv_idx_out && block_tables && seq_lens && num_blocks, PE_ERR_INVALID_ARG,
             "evict_select_compact: null pointer");
  PE_REQUIRE(method >= PE_EVICT_STREAMING && method <= PE_EVICT_INVERSE_KEY_L2, PE_ERR_INVALID_ARG,
             "unknown eviction method %d", method);
  PE_REQUIRE(block_size > 0 && max_blocks >= 2 && cache_budget >= 3 * block_size,
             PE_ERR_INVALID_ARG,
             "need block_size > 0, a table at least 2 blocks wide and a budget of >= 3 blocks");
  PE_REQUIRE(out_tables == nullptr || (out_width >= 1 && out_width <= max_blocks),
             PE_ERR_INVALID_ARG, "out_width %d must be in [1, max_blocks=%d]", out_width, max_blocks);
  PE_REQUIRE(!compat_q1 || out_tables == nullptr || out_width == max_blocks - 1, PE_ERR_INVALID_ARG,
             "the reference-compatible compaction has width max_blocks-1");
  PE_REQUIRE(method != PE_EVICT_STREAMING_1 || zero_slots_out, PE_ERR_INVALID_ARG,
             "streamingLLM-1 needs zero_slots_out");
  size_t lds = 0;
  if (method == PE_EVICT_LOCAL || method == PE_EVICT_GLOBAL) {
    PE_REQUIRE(method != PE_EVICT_LOCAL || topk_blocks > 1, PE_ERR_INVALID_ARG,
               "Topk blocks must be > 1 when using 'local' evict method");
    PE_REQUIRE(score_table || victim_override, PE_ERR_INVALID_ARG,
               "local/global need a score table or host-chosen victims");
    if (score_table) {
      PE_REQUIRE(score_counts && score_rows && (value_cache || new_scores) && score_table_width > 2,
                 PE_ERR_INVALID_ARG, "score bookkeeping arguments missing");
      if (!new_scores) {
        PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && block_size * num_kv_heads <= SCORE_MAX_PAIRS,
                   PE_ERR_INVALID_ARG, "bad head layout for block scoring");
        PE_REQUIRE(score_layout == PE_SCORE_CACHE || score_layout == PE_SCORE_CACHE_FLAT,
                   PE_ERR_INVALID_ARG, "decode-time scores read the paged cache (layout 1 or 2)");
        lds = score_lds_bytes(score_layout, num_kv_heads, head_size, block_size, elem_size(dtype));
      }
    }
  }
  // PE_EVICT_FAST=0 sends every call through the general kernel (differential tests)
  static EnvInt fast_env("PE_EVICT_FAST", 1);
  const bool fast_ok = fast_env.get() != 0;
  if (fast_ok && !compat_q1 && max_blocks <= FAST_SLOTS * SCORE_THREADS &&
      score_table_width <= FAST_SLOTS * SCORE_THREADS) {
    PE_DISPATCH_ALL(dtype, T,
                    hipLaunchKernelGGL(evict_select_fast_kernel<T>, dim3((unsigned)batch),
                                       dim3(SCORE_THREADS), lds, static_cast<hipStream_t>(stream),
                                       rmv_idx_out, out_tables, zero_slots_out, block_tables, seq_lens,
                                       num_blocks, victim_override, max_blocks, out_width, method,
                                       cache_budget, block_size, topk_blocks, drop_victim_score,
                                       score_layout, score_table, score_counts, score_rows,
                                       score_table_width, (const T*)value_cache, kv_block_stride,
                                       num_kv_heads, head_size, new_scores, n_blocks_io, kv_lens_io,
                                       freed_out));
    return check_launch("evict_select_compact");
  }
  const unsigned grid = compat_q1 ? 1u : (unsigned)batch;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(evict_select_compact_kernel<T>, dim3(grid), dim3(SCORE_THREADS), lds,
                                     static_cast<hipStream_t>(stream), rmv_idx_out, out_tables,
                                     zero_slots_out, block_tables, seq_lens, num_blocks,
                                     victim_override, batch, max_blocks, out_width, method, cache_budget,
                                     block_size, topk_blocks, drop_victim_score, compat_q1,
                                     score_layout, score_table, score_counts, score_rows,
                                     score_table_width, (const T*)value_cache, kv_block_stride,
                                     num_kv_heads, head_size, new_scores, n_blocks_io, kv_lens_io,
                                     freed_out));
  return check_launch("evict_select_compact");
}

int pe_evict_select_compact(int32_t* rmv_idx_out, int32_t* out_tables, int64_t* zero_slots_out,
                            const int32_t* block_tables, const int32_t* seq_lens,
                            const int32_t* num_blocks, const int32_t* victim_override, int batch,
                            int max_blocks, int out_width, int method, int cache_budget, int block_size,
                            int topk_blocks, int drop_victim_score, int compat_q1,
                            int score_layout, float* score_table, int32_t* score_counts,
                            const int32_t* score_rows, int score_table_width,
                            const void* value_cache, int64_t kv_block_stride, int num_kv_heads,
                            int head_size, const float* new_scores, int dtype, void* stream) {
  return evict_select_impl(rmv_idx_out, out_tables, zero_slots_out, block_tables, seq_lens,
                           num_blocks, victim_override, batch, max_blocks, out_width, method,
                           cache_budget, block_size, topk_blocks, drop_victim_score, compat_q1,
                           score_layout, score_table, score_counts, score_rows, score_table_width,
                           value_cache, kv_block_stride, num_kv_heads, head_size, new_scores, dtype,
                           stream, nullptr, nullptr, nullptr);
}

int pe_evict_select_compact_inplace(int32_t* rmv_idx_out, int32_t* freed_blocks_out,
                                    int64_t* zero_slots_out, int32_t* block_tables,
                                    int32_t* num_blocks, int32_t* kv_lens, const int32_t* seq_lens,
                                    const int32_t* victim_override, int batch, int max_blocks,
                                    int method, int cache_budget, int block_size, int topk_blocks,
                                    int drop_victim_score, int score_layout, float* score_table,
                                    int32_t* score_counts, const int32_t* score_rows,
                                    int score_table_width, const void* value_cache,
                                    int64_t kv_block_stride, int num_kv_heads, int head_size,
                                    const float* new_scores, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(batch <= 0 || (freed_blocks_out && num_blocks && kv_lens && block_tables),
             PE_ERR_INVALID_ARG, "evict_select_compact_inplace: null pointer");
  return evict_select_impl(rmv_idx_out, block_tables, zero_slots_out, block_tables, seq_lens,
                           num_blocks, victim_override, batch, max_blocks, max_blocks, method,
                           cache_budget, block_size, topk_blocks, drop_victim_score, 0, score_layout,
                           score_table, score_counts, score_rows, score_table_width, value_cache,
                           kv_block_stride, num_kv_heads, head_size, new_scores, dtype, stream,
                           num_blocks, kv_lens, freed_blocks_out);
}

// ---------------------------------------------------------------------------------------
// Device-resident decode state: one thread per batch slot appends the step's token.
}  // extern "C"

namespace pe {
static __device__ unsigned int state_diag_word;

__global__ void table_state_advance_kernel(
    int32_t* __restrict__ tables, int32_t* __restrict__ n_blocks, int32_t* __restrict__ kv_lens,
    int32_t* __restrict__ seq_lens, const int32_t* __restrict__ new_blocks, int batch,
    int max_blocks, int64_t* __restrict__ slot_mapping_out, int32_t* __restrict__ attn_kv_lens_out,
    int32_t* __restrict__ last_blocks_out, int rule, int cache_budget, int block_size,
    int evict_freq, int num_block_merge, int initial_blocks, int ring_len, int64_t ring_stride) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  int32_t* row = tables + (int64_t)b * max_blocks;
  int nb = n_blocks[b];
  const int kv = kv_lens[b] + 1;   // physical length with this step's token
  const int L = seq_lens[b] + 1;   // logical length with this step's token
  // ring form: the step's entry of sequence b sits in the ring row its new logical length selects,
  // so the same launch (a node of a replayed graph) finds a new row at every step
  const int nblk_new = new_blocks[(ring_len > 1 ? (int64_t)(L % ring_len) * ring_stride : 0) + b];
  bool bad = false;
  if (nblk_new >= 0) {
    if (nb < max_blocks) {
      row[nb] = nblk_new;
      nb += 1;
    } else {
      bad = true;
    }
  }
  // the token's position must fall into the row's last block (host and device count alike)
  if (nb <= 0 || (kv - 1) / block_size != nb - 1) bad = true;
  if (bad) {
    atomicOr(&state_diag_word, (unsigned)PE_DIAG_TABLE_STATE_MISMATCH);
    slot_mapping_out[b] = -1;  // the token is not cached rather than written somewhere else
  } else {
    slot_mapping_out[b] = (int64_t)row[nb - 1] * block_size + (kv - 1) % block_size;
  }
  int attn = kv;
  if (rule == 1) {  // budget type: one block leaves on a boundary past the budget
    if (L > cache_budget && L % block_size == 0) attn -= block_size;
  } else if (rule == 2) {  // percentage type: num_block_merge-1 blocks every evict_freq blocks
    if (L % (block_size * evict_freq) == 0 && kv > (initial_blocks + num_block_merge) * block_size)
      attn -= (num_block_merge - 1) * block_size;
  }
  n_blocks[b] = nb;
  kv_lens[b] = kv;
  seq_lens[b] = L;
  attn_kv_lens_out[b] = attn;
  if (last_blocks_out) last_blocks_out[b] = nb > 0 ? row[nb - 1] : 0;
}

int state_diag_fetch(unsigned int* out, int reset) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(state_diag_word));
  if (e == hipSuccess) e = hipMemcpy(out, p, sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset && *out != 0) e = hipMemset(p, 0, sizeof(unsigned int));
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "state diagnostics: %s", hipGetErrorString(e));
  return PE_OK;
}
}  // namespace pe

extern "C" {

static int table_state_advance_impl(int32_t* tables, int32_t* num_blocks, int32_t* kv_lens,
                                    int32_t* seq_lens, const int32_t* new_blocks, int ring_len,
                                    int64_t ring_stride, int batch, int max_blocks,
                                    int64_t* slot_mapping_out, int32_t* attn_kv_lens_out,
                                    int32_t* last_blocks_out, int evict_rule, int cache_budget,
                                    int block_size, int evict_freq, int num_block_merge,
                                    int initial_blocks, void* stream) {
  using namespace pe;
  PE_REQUIRE(batch >= 0, PE_ERR_INVALID_ARG, "negative batch");
  if (batch == 0) return PE_OK;
  PE_REQUIRE(tables && num_blocks && kv_lens && seq_lens && new_blocks && slot_mapping_out &&
                 attn_kv_lens_out,
             PE_ERR_INVALID_ARG, "table_state_advance: null pointer");
  PE_REQUIRE(max_blocks > 0 && block_size > 0 && evict_rule >= 0 && evict_rule <= 2,
             PE_ERR_INVALID_ARG, "table_state_advance: bad sizes");
  PE_REQUIRE(evict_rule != 2 || (evict_freq > 0 && num_block_merge >= 2), PE_ERR_INVALID_ARG,
             "table_state_advance: percentage rule needs evict_freq and num_block_merge");
  PE_REQUIRE(ring_len >= 1 && (ring_len == 1 || ring_stride >= batch), PE_ERR_INVALID_ARG,
             "table_state_advance: ring rows shorter than the batch");
  hipLaunchKernelGGL(table_state_advance_kernel, dim3((unsigned)((batch + 127) / 128)), dim3(128), 0,
                     static_cast<hipStream_t>(stream), tables, num_blocks, kv_lens, seq_lens,
                     new_blocks, batch, max_blocks, slot_mapping_out, attn_kv_lens_out,
                     last_blocks_out, evict_rule, cache_budget, block_size, evict_freq,
                     num_block_merge, initial_blocks, ring_len, ring_stride);
  return check_launch("table_state_advance");
}

int pe_table_state_advance(int32_t* tables, int32_t* num_blocks, int32_t* kv_lens,
                           int32_t* seq_lens, const int32_t* new_blocks, int batch, int max_blocks,
                           int64_t* slot_mapping_out, int32_t* attn_kv_lens_out,
                           int32_t* last_blocks_out, int evict_rule, int cache_budget,
                           int block_size, int evict_freq, int num_block_merge, int initial_blocks,
                           void* stream) {
  return table_state_advance_impl(tables, num_blocks, kv_lens, seq_lens, new_blocks, 1, 0, batch,
                                  max_blocks, slot_mapping_out, attn_kv_lens_out, last_blocks_out,
                                  evict_rule, cache_budget, block_size, evict_freq, num_block_merge,
                                  initial_blocks, stream);
}

int pe_table_state_advance_ring(int32_t* tables, int32_t* num_blocks, int32_t* kv_lens,
                                int32_t* seq_lens, const int32_t* new_blocks_ring, int ring_len,
                                int64_t ring_stride, int batch, int max_blocks,
                                int64_t* slot_mapping_out, int32_t* attn_kv_lens_out,
                                int32_t* last_blocks_out, int evict_rule, int cache_budget,
                                int block_size, int evict_freq, int num_block_merge,
                                int initial_blocks, void* stream) {
  using namespace pe;
  PE_REQUIRE(ring_len >= 2, PE_ERR_INVALID_ARG, "table_state_advance_ring: a ring has at least 2 rows");
  return table_state_advance_impl(tables, num_blocks, kv_lens, seq_lens, new_blocks_ring, ring_len,
                                  ring_stride, batch, max_blocks, slot_mapping_out, attn_kv_lens_out,
                                  last_blocks_out, evict_rule, cache_budget, block_size, evict_freq,
                                  num_block_merge, initial_blocks, stream);
}

}  // extern "C"
