// Single-query ("decode") attention over a paged KV cache on gfx950.
//
// Fast path (16-bit types, block_size 16 [8 and 32 with head sizes 64/128], head_size in {64,128,256}, up to 16 query
// heads per KV head): one workgroup per (KV split, KV head, sequence); all query heads
// that share the KV head are processed together, so every K and V byte is read from
// HBM once.  K and V go straight from global memory into MFMA operand registers —
// the reference's paged layouts already are MFMA fragment layouts for 16-token blocks:
//
//   key_cache block  [D/8][16 tok][8]   : lane l of the wave reads 16 bytes at
//                    kstep*1024 + l*16  = A fragment (row = token l&15,
//                    k = 8*(l>>4)+j) of v_mfma_f32_16x16x32 for S^T = K.Q^T;
//   value_cache block [D][16 tok]       : lane l reads 8 bytes (4 tokens 4*(l>>4)..)
//                    of row d = 16*dt + (l&15) from each of two blocks = A fragment of
//                    O^T += V^T.P^T, with the 32 k-slots of the MFMA ordered
//                    (lane group g, j<4) -> block A token 4g+j, (g, j>=4) -> block B
//                    token 4g+j-4, which is exactly how the S^T accumulators of the two
//                    blocks sit in the lanes, so P needs no cross-lane movement.
//
// Softmax is the online form in f32; the final normalisation is 1/(sum + 1e-6) as in
// the reference kernel (csrc/attention/attention_kernels.cuh:342) and P is rounded to
// the storage type before P.V (:398-400).
//
// Generic path (any head size that is a multiple of the 16-byte chunk, any block size,
// f32 too): one workgroup per (head, sequence), logits staged in LDS like the reference.
#include <cfloat>
#include <cstdlib>
#include <mutex>

#include "paged_decode_kernel.h"

namespace pe {

// merge of the split partials: grid (num_heads, num_seqs), one thread per d.  All partials of
// a thread are requested before any is used (one memory round trip instead of one per pass).
template <typename T, int MAX_SPLITS>
__global__ void paged_decode_merge_kernel(T* __restrict__ out, const float* __restrict__ ws,
                                          int num_heads, int head_size, int num_splits) {
  const int head = blockIdx.x, seq = blockIdx.y, num_seqs = gridDim.y;
  const int64_t stride = ws_stride(head_size);
  const int64_t split_stride = (int64_t)num_seqs * num_heads * stride;
  const float* base = ws + ((int64_t)seq * num_heads + head) * stride;
  for (int d = threadIdx.x; d < head_size; d += blockDim.x) {
    float m[MAX_SPLITS], l[MAX_SPLITS], o[MAX_SPLITS];
#pragma unroll
    for (int s = 0; s < MAX_SPLITS; ++s) {
      const bool on = s < num_splits;
      const float* w = base + (on ? s : 0) * split_stride;
      m[s] = on ? w[0] : NEG_BIG;
      l[s] = on ? w[1] : 0.f;
      o[s] = on ? w[2 + d] : 0.f;
    }
    float m_tot = NEG_BIG;
#pragma unroll
    for (int s = 0; s < MAX_SPLITS; ++s) m_tot = fmaxf(m_tot, m[s]);
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_SPLITS; ++s) {
      const float f = __expf(m[s] - m_tot);
      acc += f * o[s];
      lsum += f * l[s];
    }
    out[((int64_t)seq * num_heads + head) * head_size + d] = (T)(acc * (1.f / (lsum + 1e-6f)));
  }
}

// More than 16 splits (only a forced split count gets there): the same merge as a plain loop -
// the maximum first, then the weighted sums in split order.
template <typename T>
__global__ void paged_decode_merge_loop_kernel(T* __restrict__ out, const float* __restrict__ ws,
                                               int num_heads, int head_size, int num_splits) {
  const int head = blockIdx.x, seq = blockIdx.y, num_seqs = gridDim.y;
  const int64_t stride = ws_stride(head_size);
  const int64_t split_stride = (int64_t)num_seqs * num_heads * stride;
  const float* base = ws + ((int64_t)seq * num_heads + head) * stride;
  for (int d = threadIdx.x; d < head_size; d += blockDim.x) {
    float m_tot = NEG_BIG;
    for (int s = 0; s < num_splits; ++s) m_tot = fmaxf(m_tot, base[s * split_stride]);
    float acc = 0.f, lsum = 0.f;
    for (int s = 0; s < num_splits; ++s) {
      const float* w = base + s * split_stride;
      const float f = __expf(w[0] - m_tot);
      acc += f * w[2 + d];
      lsum += f * w[1];
    }
    out[((int64_t)seq * num_heads + head) * head_size + d] = (T)(acc * (1.f / (lsum + 1e-6f)));
  }
}

// ---------------------------------------------------------------------------------------
// Generic path: the reference's three passes (logits -> softmax -> P.V), one workgroup of
// 256 threads per (head, sequence), logits in dynamic LDS.
template <typename T>
__global__ __launch_bounds__(256) void paged_decode_generic_kernel(
    T* __restrict__ out, const T* __restrict__ query, const T* __restrict__ key_cache,
    const T* __restrict__ value_cache, const int32_t* __restrict__ block_tables,
    const int32_t* __restrict__ seq_lens, const float* __restrict__ alibi_slopes, int num_heads,
    int num_kv_heads, int head_size, int block_size, int max_num_blocks_per_seq, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, float scale, int max_len) {
  constexpr int X = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* q_s = reinterpret_cast<float*>(smem);       // [head_size]
  float* red = q_s + ((head_size + 3) & ~3);          // [8]
  float* logits = red + 8;                            // [seq_len]
  const int head = blockIdx.x, seq = blockIdx.y;
  const int kvh = head / (num_heads / num_kv_heads);
  // clamped to what the table row addresses and to the LDS logits buffer (reported through the
  // diagnostic word)
  const int raw_len = seq_lens[seq];
  const int seq_len = min(raw_len, min(max_num_blocks_per_seq * block_size, max_len));
  if (raw_len > seq_len && threadIdx.x == 0 && head == 0)
    atomicOr(&decode_diag_word, (unsigned)PE_DIAG_DECODE_LEN_CLAMPED);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  const float slope = alibi_slopes ? alibi_slopes[head] : 0.f;
  for (int d = tid; d < head_size; d += 256)
    q_s[d] = to_f32(query[(int64_t)seq * q_stride + (int64_t)head * head_size + d]);
  __syncthreads();

  float mx = -FLT_MAX;
  for (int t = tid; t < seq_len; t += 256) {
    const int64_t blk = table[t / block_size];
    const int off = t % block_size;
    const T* kp = key_cache + blk * kv_block_stride + (int64_t)kvh * kv_head_stride + (int64_t)off * X;
    float dot = 0.f;
    for (int c = 0; c < head_size / X; ++c) {
      union {
        u32x4 v;
        T e[X];
      } u;
      u.v = *reinterpret_cast<const u32x4*>(kp + (int64_t)c * block_size * X);
#pragma unroll
      for (int e = 0; e < X; ++e) dot += q_s[c * X + e] * to_f32(u.e[e]);
    }
    float x = scale * dot;
    if (slope != 0.f) x += slope * (float)(t - seq_len + 1);
    logits[t] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = tid; t < seq_len; t += 256) {
    const float e = __expf(logits[t] - mx);
    logits[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7] + 1e-6f);
  for (int d = tid; d < head_size; d += 256) {
    float acc = 0.f;
    for (int t = 0; t < seq_len; ++t) {
      const int64_t blk = table[t / block_size];
      const int off = t % block_size;
      const float p = round_to<T>(logits[t] * inv);
      acc += p * to_f32(value_cache[blk * kv_block_stride + (int64_t)kvh * kv_head_stride +
                                    (int64_t)d * block_size + off]);
    }
    out[((int64_t)seq * num_heads + head) * head_size + d] = from_f32<T>(acc);
  }
}


// Upper limit of the split count for a maximum length: a split has at least one block pair per
// wave of the narrowest workgroup.  pe_paged_decode_workspace_bytes sizes the workspace with it
// and choose_shape never exceeds it, whatever the tunables say.
static int split_cap(int max_seq_len) {
  const int n_pairs = (max_seq_len + 31) / 32;
  int cap = n_pairs / DEC_WAVES;
  if (cap < 1) cap = 1;
  if (cap > 64) cap = 64;
  return cap;
}

// Number of KV splits of the throughput shape (4 waves).  A split costs a second (merge) launch
// and a round trip of the partial results, about 2.5 us, so it only pays when the workgroups are
// few AND long.  Rule: aim at one workgroup per CU, at least 12 pairs per split while the chip is
// less than a quarter full, 24 beyond, at most 16 splits.
static int choose_splits(int num_seqs, int num_kv_heads, int max_seq_len) {
  const int n_pairs = (max_seq_len + 31) / 32;
  const int cap = split_cap(max_seq_len);
  static EnvInt min_pairs_var("PE_DEC_SPLIT_MIN_PAIRS", 0);
  const int min_pairs_env = min_pairs_var.get();
  const int64_t wgs = (int64_t)num_seqs * num_kv_heads;
  const int64_t target = 256;  // one workgroup per CU
  const int min_pairs = min_pairs_env > 0 ? min_pairs_env : (wgs * 4 <= target ? 12 : 24);
  int max_splits = n_pairs / min_pairs;
  if (max_splits < 1) max_splits = 1;
  if (max_splits > 16) max_splits = 16;
  if (max_splits > cap) max_splits = cap;
  int splits = (int)(target / wgs);
  if (splits < 1) splits = 1;
  if (splits > max_splits) splits = max_splits;
  return splits;
}

// LDS of the in-workgroup merge: o[waves][gqa][head_size + 4] + m, l [waves][16] (f32)
static size_t merge_lds_bytes(int waves, int gqa, int head_size) {
  const int padded = (head_size + 15) / 16 * 16;
  return ((size_t)waves * gqa * (padded + 4) + 2 * (size_t)waves * 16) * 4;
}

// Workgroup width and split count of a launch.
//   Small launches are bound by latency, not bandwidth: a wave that walks its block pairs one
//   memory round trip (~1.7 us) after the other is what takes the time, and one CU cannot pull
//   more than ~100 GB/s however many waves it runs.  While the whole launch has at most 2048
//   block pairs (the wave slots of the chip at two waves per SIMD) every pair gets a wave of its
//   own: 8-wave workgroups, one split per 8 pairs (at most 16; the merge launch reads all
//   partials in one round trip).  Measured on MI355X, graph-replayed, us per launch before ->
//   after: 1 sequence x 8 KV heads, kv 776: 11.4 -> 8.1; OPT-125m geometry, kv 1074: 10.1 -> 7.4;
//   8 sequences: 12.3 -> 9.9; 64 sequences x 1 KV head (one rank of TP=8), kv 776: 13.6 -> 10.5;
//   1 sequence of kv 4104: 16.3 -> 11.5.
//   Larger launches keep the throughput shape: 4 waves (two workgroups per CU, two tiles in
//   flight per wave), splits only while the chip is not full.
//   Experiment overrides: PE_DECODE_WAVES (4/8/16), PE_DECODE_SPLITS.
static void choose_shape(int num_seqs, int num_heads, int num_kv_heads, int head_size,
                         int max_seq_len, int* waves, int* splits) {
  const int n_pairs = (max_seq_len + 31) / 32;
  const int cap = split_cap(max_seq_len);
  const int64_t wgs = (int64_t)num_seqs * num_kv_heads;
  static EnvInt latency_pairs_var("PE_DEC_LATENCY_PAIRS", 2048);
  const int64_t latency_pairs = latency_pairs_var.get();
  *waves = DEC_WAVES;
  if (wgs * n_pairs <= latency_pairs && n_pairs > DEC_WAVES) {
    *waves = 8;
    int sp = (n_pairs + 7) / 8;
    if (sp > 16) sp = 16;
    if (sp > cap) sp = cap;
    *splits = sp;
  } else {
    *splits = choose_splits(num_seqs, num_kv_heads, max_seq_len);
  }
  static EnvInt forced_w_var("PE_DECODE_WAVES", 0), forced_s_var("PE_DECODE_SPLITS", 0);
  const int forced_w = forced_w_var.get();
  if (forced_w == 4 || forced_w == 8 || forced_w == 16) *waves = forced_w;
  const int forced_s = forced_s_var.get();
  if (forced_s > 0) *splits = forced_s < cap ? forced_s : cap;
  while (*waves > DEC_WAVES &&
         merge_lds_bytes(*waves, num_heads / num_kv_heads, head_size) > 64 * 1024)
    *waves /= 2;
}

static bool mfma_path_ok(int num_heads, int num_kv_heads, int head_size, int block_size,
                         int dtype, int64_t q_stride, int64_t kv_block_stride,
                         int64_t kv_head_stride) {
  static EnvInt force_generic("PE_DECODE_FORCE_GENERIC", 0);
  if (force_generic.get()) return false;
  if (dtype != PE_BF16 && dtype != PE_F16) return false;
  if (block_size != 16 && block_size != 8 && block_size != 32) return false;
  // the head sizes the reference's op accepts (vllm/attention/ops/paged_attn.py:37)
  if (head_size != 32 && head_size != 64 && head_size != 80 && head_size != 96 && head_size != 112 &&
      head_size != 120 && head_size != 128 && head_size != 192 && head_size != 256)
    return false;
  // 8- and 32-token blocks (the page sizes of scripts/diff_page_sizes.sh:16) are built for the
  // head sizes whose rows are whole 32-element k-steps (32, 64, 96, 128, 192, 256)
  if (block_size != 16 && head_size % 32 != 0) return false;
  if (num_heads / num_kv_heads > 16) return false;
  if (q_stride % 8 || kv_block_stride % 8 || kv_head_stride % 8) return false;
  return true;
}

// Instantiation choice for head sizes 64 / 128 / 256: waves per workgroup (4, 8, or - plain form,
// 16-token blocks, head sizes up to 128 - 16), rotating form or not, tokens per block.
template <typename T, int HS, int B>
static void dispatch_dec_block(const DecLaunch& la, int waves) {
  if (la.rope != nullptr) {
    if constexpr (HS <= 128) {
      if (waves == 8) launch_dec<T, HS, 8, true, B>(la);
      else launch_dec<T, HS, DEC_WAVES, true, B>(la);
    }
    return;
  }
  if constexpr (B == 16 && HS <= 128) {
    if (waves == 16) {
      launch_dec<T, HS, 16, false, 16>(la);
      return;
    }
  }
  if (waves == 8) launch_dec<T, HS, 8, false, B>(la);
  else launch_dec<T, HS, DEC_WAVES, false, B>(la);
}

template <typename T, int HS>
static void dispatch_dec(const DecLaunch& la, int waves, int block_size) {
  if constexpr (HS <= 128) {
    if (block_size == 32) return dispatch_dec_block<T, HS, 32>(la, waves);
    if (block_size == 8) return dispatch_dec_block<T, HS, 8>(la, waves);
  }
  dispatch_dec_block<T, HS, 16>(la, waves);
}

int decode_diag_fetch(unsigned int* out, int reset) {
  unsigned int a = 0, b = 0;
  if (int rc = decode_diag_fetch_tu(&a, reset)) return rc;
  if (int rc = decode_hs_diag_fetch(&b, reset)) return rc;
  unsigned int c = 0;
  if (int rc = decode_blk_diag_fetch(&c, reset)) return rc;
  b |= c;
  *out = a | b;
  return PE_OK;
}

}  // namespace pe

extern "C" {

int64_t pe_paged_decode_workspace_bytes(int num_seqs, int num_heads, int num_kv_heads,
                                        int head_size, int block_size, int max_seq_len) {
  (void)block_size;
  if (num_seqs <= 0 || num_heads <= 0 || num_kv_heads <= 0) return 0;
  // sized for the largest split count choose_splits can pick, so a caller may allocate once
  const int max_splits = pe::split_cap(max_seq_len);
  if (max_splits <= 1) return 0;
  return (int64_t)max_splits * num_seqs * num_heads * pe::ws_stride(head_size) * 4;
}

static int decode_impl(void* out, const void* query, const void* key_cache, const void* value_cache,
                       const int32_t* block_tables, const int32_t* seq_lens,
                       const float* alibi_slopes, int num_seqs, int num_heads, int num_kv_heads,
                       int head_size, int block_size, int max_num_blocks_per_seq, int64_t q_stride,
                       int64_t kv_block_stride, int64_t kv_head_stride, float scale, int max_seq_len,
                       int dtype, void* workspace, int64_t workspace_bytes, void* stream,
                       const void* new_key, const void* new_value, const int64_t* new_slots,
                       int64_t new_k_stride, int64_t new_v_stride, int64_t num_slots,
                       const pe::RopeArgs* rope = nullptr) {
  using namespace pe;
  PE_REQUIRE(num_seqs >= 0, PE_ERR_INVALID_ARG, "negative num_seqs");
  if (num_seqs == 0) return PE_OK;
  PE_REQUIRE(out && query && key_cache && value_cache && block_tables && seq_lens,
             PE_ERR_INVALID_ARG, "paged_decode: null pointer");
  PE_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
             PE_ERR_INVALID_ARG, "num_heads %d must be a positive multiple of num_kv_heads %d",
             num_heads, num_kv_heads);
  PE_REQUIRE(head_size > 0 && block_size > 0 && max_num_blocks_per_seq > 0 && max_seq_len >= 0,
             PE_ERR_INVALID_ARG, "paged_decode: bad sizes");
  // max_seq_len is an upper bound used for sizing only: under eviction the reference passes the
  // logical (unpruned) maximum here (torch_cuda_sdpa.py:446-447), which may exceed what the
  // compacted table addresses
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(head_size % x == 0, PE_ERR_UNSUPPORTED, "head_size %d not a multiple of %d", head_size, x);
  PE_REQUIRE(num_seqs <= 65535 && num_kv_heads <= 65535, PE_ERR_UNSUPPORTED, "grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);

  if (mfma_path_ok(num_heads, num_kv_heads, head_size, block_size, dtype, q_stride,
                   kv_block_stride, kv_head_stride)) {
    int splits = 1, waves = DEC_WAVES;
    choose_shape(num_seqs, num_heads, num_kv_heads, head_size, max_seq_len, &waves, &splits);
    // the rotating form, the 8-/32-token-block forms and head size 256 (whose 16-wave form would
    // spill) are built for 4 and 8 waves
    if ((rope != nullptr || block_size != 16 || head_size > 128) && waves > 8) waves = 8;
    float* ws = static_cast<float*>(workspace);
    if (splits > 1) {
      const int64_t need = (int64_t)splits * num_seqs * num_heads * ws_stride(head_size) * 4;
      PE_REQUIRE(workspace && workspace_bytes >= need, PE_ERR_WORKSPACE,
                 "paged_decode needs %lld workspace bytes, got %lld", (long long)need,
                 (long long)workspace_bytes);
    }
    dim3 grid((unsigned)splits, (unsigned)num_kv_heads, (unsigned)num_seqs);
    const size_t lds = merge_lds_bytes(waves, num_heads / num_kv_heads, head_size);
    DecLaunch la{out, ws, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes,
                 num_heads, num_kv_heads, max_num_blocks_per_seq, q_stride, kv_block_stride,
                 kv_head_stride, scale, splits, new_key, new_value, new_slots, new_k_stride,
                 new_v_stride, num_slots, rope, grid, lds, st};
    if (block_size != 16 && head_size != 64 && head_size != 128) {
      // 8- / 32-token blocks of the other head sizes: no rotating form, 4- or 8-wave workgroups
      if (int rc = launch_decode_other_blocks(la, head_size, block_size, waves >= 8 ? 8 : DEC_WAVES, dtype))
        return rc;
    } else if (head_size != 64 && head_size != 128 && head_size != 256) {
      // the other head sizes: 16-token blocks, no rotation (mfma_path_ok / the entry points
      // have seen to that), 4- or 8-wave workgroups
      if (int rc = launch_decode_other_heads(la, head_size, waves >= 8 ? 8 : DEC_WAVES, dtype)) return rc;
    } else {
      PE_DISPATCH_16(dtype, T, {
        if (head_size == 64) dispatch_dec<T, 64>(la, waves, block_size);
        else if (head_size == 128) dispatch_dec<T, 128>(la, waves, block_size);
        else dispatch_dec<T, 256>(la, waves, block_size);
      });
    }
    if (int rc = check_launch("paged_decode")) return rc;
    if (splits > 1) {
#define PE_LAUNCH_MERGE(T, MS)                                                                   \
  hipLaunchKernelGGL((paged_decode_merge_kernel<T, MS>), dim3((unsigned)num_heads, (unsigned)num_seqs), \
                     dim3(head_size <= 64 ? 64 : 128), 0, st, (T*)out, ws, num_heads, head_size, splits)
      PE_DISPATCH_16(dtype, T, {
        if (splits <= 4) PE_LAUNCH_MERGE(T, 4);
        else if (splits <= 8) PE_LAUNCH_MERGE(T, 8);
        else if (splits <= 16) PE_LAUNCH_MERGE(T, 16);
        else
          hipLaunchKernelGGL((paged_decode_merge_loop_kernel<T>),
                             dim3((unsigned)num_heads, (unsigned)num_seqs),
                             dim3(head_size <= 64 ? 64 : 128), 0, st, (T*)out, ws, num_heads, head_size,
                             splits);
      });
#undef PE_LAUNCH_MERGE
      if (int rc = check_launch("paged_decode_merge")) return rc;
    }
    return PE_OK;
  }

  // generic path: attention only - it cannot take the step's token from registers, so a fused
  // call that lands here must fail instead of silently leaving the cache unwritten
  PE_REQUIRE(new_key == nullptr, PE_ERR_UNSUPPORTED,
             "decode_step: fused KV write needs the MFMA path (16-bit dtype, block_size 16 with "
             "head_size 64/128/256 or block_size 8/32 with head_size 64/128, strides multiples of "
             "8); ask "
             "pe_paged_decode_supports_fused_write first");
  int gen_len = max_seq_len;
  if ((int64_t)max_num_blocks_per_seq * block_size < gen_len) gen_len = max_num_blocks_per_seq * block_size;
  const size_t lds = (size_t)(((head_size + 3) & ~3) + 8 + gen_len) * 4;
  PE_REQUIRE(lds <= 160 * 1024, PE_ERR_UNSUPPORTED,
             "generic decode path holds the logits in LDS: max_seq_len %d too long for "
             "head_size %d / block_size %d / dtype %d",
             max_seq_len, head_size, block_size, dtype);
  PE_REQUIRE(num_heads <= 0x7fffffff, PE_ERR_UNSUPPORTED, "grid too large");
  dim3 grid((unsigned)num_heads, (unsigned)num_seqs);
#define PE_LAUNCH_GEN(T)                                                                        \
  do {                                                                                          \
    if (lds > 64 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&paged_decode_generic_kernel<T>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL(paged_decode_generic_kernel<T>, grid, dim3(256), lds, st, (T*)out,       \
                       (const T*)query, (const T*)key_cache, (const T*)value_cache,             \
                       block_tables, seq_lens, alibi_slopes, num_heads, num_kv_heads, head_size, \
                       block_size, max_num_blocks_per_seq, q_stride, kv_block_stride,           \
                       kv_head_stride, scale, gen_len);                                         \
  } while (0)
  PE_DISPATCH_ALL(dtype, T, PE_LAUNCH_GEN(T));
#undef PE_LAUNCH_GEN
  return check_launch("paged_decode_generic");
}

int pe_paged_decode(void* out, const void* query, const void* key_cache, const void* value_cache,
                    const int32_t* block_tables, const int32_t* seq_lens,
                    const float* alibi_slopes, int num_seqs, int num_heads, int num_kv_heads,
                    int head_size, int block_size, int max_num_blocks_per_seq, int64_t q_stride,
                    int64_t kv_block_stride, int64_t kv_head_stride, float scale, int max_seq_len,
                    int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  return decode_impl(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes,
                     num_seqs, num_heads, num_kv_heads, head_size, block_size,
                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, scale,
                     max_seq_len, dtype, workspace, workspace_bytes, stream, nullptr, nullptr,
                     nullptr, 0, 0, 0);
}

int pe_paged_decode_supports_fused_write(int num_heads, int num_kv_heads, int head_size,
                                         int block_size, int dtype) {
  return pe::mfma_path_ok(num_heads, num_kv_heads, head_size, block_size, dtype, 8, 8, 8) ? 1 : 0;
}

int pe_decode_step(void* out, const void* query, const void* new_key, const void* new_value,
                   void* key_cache, void* value_cache, const int64_t* slot_mapping,
                   const int32_t* block_tables, const int32_t* seq_lens, const float* alibi_slopes,
                   int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                   int64_t num_blocks, int max_num_blocks_per_seq, int64_t q_stride,
                   int64_t new_k_stride, int64_t new_v_stride, int64_t kv_block_stride,
                   int64_t kv_head_stride, float scale, int max_seq_len, int dtype, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_seqs == 0 || (new_key && new_value && slot_mapping), PE_ERR_INVALID_ARG,
             "decode_step: null new_key/new_value/slot_mapping");
  PE_REQUIRE(new_k_stride % 8 == 0 && new_v_stride % 8 == 0 &&
                 (((uintptr_t)new_key | (uintptr_t)new_value) % 16 == 0),
             PE_ERR_INVALID_ARG, "decode_step: new key/value rows must be 16-byte aligned");
  PE_REQUIRE(num_blocks > 0, PE_ERR_INVALID_ARG, "decode_step: bad num_blocks");
  return decode_impl(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes,
                     num_seqs, num_heads, num_kv_heads, head_size, block_size,
                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, scale,
                     max_seq_len, dtype, workspace, workspace_bytes, stream, new_key, new_value,
                     slot_mapping, new_k_stride, new_v_stride, num_blocks * block_size);
}

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
                        void* stream) {
  using namespace pe;
  PE_REQUIRE(num_seqs == 0 || (new_key && new_value && slot_mapping && cos_sin_cache),
             PE_ERR_INVALID_ARG, "decode_step_rope: null pointer");
  PE_REQUIRE(positions != nullptr || max_position >= num_seqs, PE_ERR_INVALID_ARG,
             "decode_step_rope: without positions the table needs one row per sequence (%lld < %d)",
             (long long)max_position, num_seqs);
  PE_REQUIRE(new_k_stride % 8 == 0 && new_v_stride % 8 == 0 &&
                 (((uintptr_t)new_key | (uintptr_t)new_value | (uintptr_t)cos_sin_cache) % 16 == 0),
             PE_ERR_INVALID_ARG, "decode_step_rope: rows must be 16-byte aligned");
  PE_REQUIRE(num_blocks > 0 && max_position > 0, PE_ERR_INVALID_ARG,
             "decode_step_rope: bad num_blocks / max_position");
  PE_REQUIRE(mfma_path_ok(num_heads, num_kv_heads, head_size, block_size, dtype, q_stride,
                          kv_block_stride, kv_head_stride),
             PE_ERR_UNSUPPORTED,
             "decode_step_rope needs the MFMA decode path (16-bit dtype, block_size 8/16/32, head_size "
             "64/128, strides multiples of 8): use pe_rope_reshape_and_cache + pe_paged_decode");
  PE_REQUIRE(head_size == 64 || head_size == 128, PE_ERR_UNSUPPORTED,
             "decode_step_rope is built for head sizes 64 and 128 (head_size %d): use "
             "pe_rope_reshape_and_cache + pe_paged_decode", head_size);
  // both elements of a rotation pair must sit in one lane of the fragment layout
  PE_REQUIRE(rot_dim > 0 && rot_dim <= head_size &&
                 (is_neox ? rot_dim % 64 == 0 : rot_dim % 16 == 0),
             PE_ERR_UNSUPPORTED,
             "decode_step_rope: rot_dim %d (head_size %d) must be a multiple of %d; use "
             "pe_rope_reshape_and_cache + pe_paged_decode",
             rot_dim, head_size, is_neox ? 64 : 16);
  const RopeArgs rope{positions, cos_sin_cache, rot_dim, is_neox ? 1 : 0, max_position};
  return decode_impl(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes,
                     num_seqs, num_heads, num_kv_heads, head_size, block_size,
                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, scale,
                     max_seq_len, dtype, workspace, workspace_bytes, stream, new_key, new_value,
                     slot_mapping, new_k_stride, new_v_stride, num_blocks * block_size, &rope);
}

// ---------------------------------------------------------------------------------------
// pe_paged_decode_noalloc: the split-KV partials live in a file-scope device array instead of a
// caller workspace.  Host code never uses the array's name as an address: its device address
// is looked up once per device with hipGetSymbolAddress and kept in noalloc_scratch_ptr[].
static __device__ __align__(256) unsigned char scratch[PE_NOALLOC_SCRATCH_BYTES];

namespace pe {
constexpr int MAX_DEVICES = 64;
static void* noalloc_scratch_ptr[MAX_DEVICES];
static std::mutex noalloc_mutex;

// Device address of a GPU's copy of `scratch` (every device holds its own copy at its own
// address).  The device is the one `stream` belongs to (the current device for the null stream)
// and must be current: a symbol resolves on the current device only.  Resolved on the first call
// per device - pe_paged_decode_noalloc_max_bytes does it at set-up, outside stream capture.
static int noalloc_scratch(void* stream, void** ptr_out) {
  int cur = -1;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  int dev = cur;
  if (stream != nullptr) {
    e = hipStreamGetDevice(static_cast<hipStream_t>(stream), &dev);
    if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "hipStreamGetDevice: %s", hipGetErrorString(e));
    PE_REQUIRE(dev == cur, PE_ERR_INVALID_ARG,
               "pe_paged_decode_noalloc: the stream belongs to device %d but device %d is current",
               dev, cur);
  }
  PE_REQUIRE(dev >= 0 && dev < MAX_DEVICES, PE_ERR_UNSUPPORTED, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(noalloc_mutex);
  if (noalloc_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr)
      return fail(PE_ERR_LAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev,
                  hipGetErrorString(e));
    noalloc_scratch_ptr[dev] = p;
  }
  *ptr_out = noalloc_scratch_ptr[dev];
  return PE_OK;
}
}  // namespace pe

int64_t pe_paged_decode_noalloc_max_bytes(void) {
  void* p = nullptr;
  (void)pe::noalloc_scratch(nullptr, &p);  // set-up: resolve the current device's array now
  return (int64_t)PE_NOALLOC_SCRATCH_BYTES;
}

int pe_paged_decode_noalloc(void* out, const void* query, const void* key_cache,
                            const void* value_cache, const int32_t* block_tables,
                            const int32_t* seq_lens, const float* alibi_slopes, int num_seqs,
                            int num_heads, int num_kv_heads, int head_size, int block_size,
                            int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride,
                            int64_t kv_head_stride, float scale, int max_seq_len, int dtype,
                            void* stream) {
  using namespace pe;
  void* ws = nullptr;
  if (int rc = noalloc_scratch(stream, &ws)) return rc;
  // decode_impl checks the partials of the split count it picks against the size given here and
  // reports PE_ERR_WORKSPACE when they do not fit; nothing in it allocates
  const int rc = decode_impl(out, query, key_cache, value_cache, block_tables, seq_lens,
                             alibi_slopes, num_seqs, num_heads, num_kv_heads, head_size, block_size,
                             max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride,
                             scale, max_seq_len, dtype, ws, (int64_t)PE_NOALLOC_SCRATCH_BYTES,
                             stream, nullptr, nullptr, nullptr, 0, 0, 0);
  if (rc == PE_ERR_WORKSPACE)
    return fail(PE_ERR_WORKSPACE,
                "pe_paged_decode_noalloc: the split-KV partials of this call (%d sequences, %d "
                "heads, head size %d, max_seq_len %d) exceed the built-in scratch of %lld bytes; "
                "use pe_paged_decode with a caller workspace",
                num_seqs, num_heads, head_size, max_seq_len, (long long)PE_NOALLOC_SCRATCH_BYTES);
  return rc;
}

}  // extern "C"
