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

#include "pe_common.h"

namespace pe {

// Build-time tunables of the MFMA path (the defaults are the measured optimum on MI355X)
#ifndef PE_DEC_WAVES
#define PE_DEC_WAVES 4  // waves per workgroup of the default (throughput) shape
#endif
#ifndef PE_DEC_TILES
#define PE_DEC_TILES 2  // 32-token tiles in flight per wave (head sizes up to 128)
#endif
#ifndef PE_DEC_OCC
#define PE_DEC_OCC 2  // waves per SIMD the register budget is set for (head sizes up to 128)
#endif
// bytes of the file-scope scratch array behind pe_paged_decode_noalloc (held by every GPU the
// library runs on until the process ends)
#ifndef PE_NOALLOC_SCRATCH_BYTES
#define PE_NOALLOC_SCRATCH_BYTES (8 << 20)
#endif
constexpr int DEC_WAVES = PE_DEC_WAVES;
constexpr float NEG_BIG = -1.0e30f;

// Set when a sequence length exceeds what its block-table row addresses and the kernel clamps
// it (a caller bug the launch itself survives); read back through pe_device_diagnostics.
static __device__ unsigned int decode_diag_word;

template <typename T, int HEAD_SIZE>
struct DecTile {
  typename Traits<T>::vec8 k[2][HEAD_SIZE / 32];
  typename Traits<T>::vec4 v[2][HEAD_SIZE / 16];
};

// Rotary-embedding inputs of pe_decode_step_rope
struct RopeArgs {
  const int64_t* positions;  // [num_seqs]
  const void* cos_sin;       // [max_position, rot_dim], first half of a row cos, second half sin
  int rot_dim;
  int is_neox;
  int64_t max_position;
};

// Partial results of one split: [split][seq][head] -> m, l, then O[D] (all f32)
__host__ __device__ inline int64_t ws_stride(int head_size) { return head_size + 2; }

// WAVES: waves per workgroup.  4 (two workgroups per CU) is the throughput shape; 8 and 16 are
// the latency shapes for launches with few (sequence, KV head) pairs: the block pairs of a
// sequence spread over more waves of ONE workgroup, which merges them through LDS, instead of
// over KV splits that need a second launch to be merged.  16 waves leave 128 registers per
// wave, enough for one tile in flight.
//
// ROPE: the rotary embedding of the step's query and key rows runs inside the launch
// (pe_decode_step_rope).  The fragment layout keeps both elements of a rotation pair in one lane
// (NeoX: element d and d + rot_dim/2 sit in k-steps ks and ks + rot_dim/64 of the same lane when
// rot_dim is a multiple of 64; GPT-J: adjacent elements), so Q is rotated in registers right after
// it is loaded and the new K row before it is used and stored; neither tensor is written back.
//
// BLK: tokens per cache block (8, 16 or 32; the reference's page-size study runs all three,
// scripts/diff_page_sizes.sh:16).  A tile is always 32 tokens - two 16-token halves, each one MFMA
// row block: with 16-token blocks a half is one block and its K slab is the A-operand image as it
// lies in memory; a 32-token block holds both halves (chunk rows of 512 bytes, a half is 256
// contiguous bytes of each); with 8-token blocks a half spans two blocks and the lanes of a half
// pick theirs (K: by token, V: by lane group).
template <typename T, int HEAD_SIZE, int WAVES, bool ROPE, int BLK = 16>
__global__ __launch_bounds__(WAVES * 64, (WAVES >= 16 ? 4 : (HEAD_SIZE <= 128 ? PE_DEC_OCC : 2)))
void paged_decode_mfma_kernel(
    T* __restrict__ out, float* __restrict__ ws, const T* __restrict__ query,
    const T* __restrict__ key_cache, const T* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ seq_lens,
    const float* __restrict__ alibi_slopes, int num_heads, int num_kv_heads,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, float scale, int num_splits, const T* __restrict__ new_key,
    const T* __restrict__ new_value, const int64_t* __restrict__ new_slots, int64_t new_k_stride,
    int64_t new_v_stride, int64_t num_slots, T* key_cache_w, T* value_cache_w,
    const int64_t* __restrict__ rope_positions, const T* __restrict__ rope_cos_sin, int rot_dim,
    int rope_neox, int64_t rope_max_position) {
  // key_cache_w / value_cache_w: the same caches through non-restrict pointers, used only by
  // the fused KV write (the written slot is never loaded by this launch).
  using TR = Traits<T>;
  using vec8 = typename TR::vec8;
  using vec4 = typename TR::vec4;
  constexpr int KS = HEAD_SIZE / 32;  // k-steps of S^T = K.Q^T
  constexpr int DT = HEAD_SIZE / 16;  // d-tiles of O^T
  constexpr int BS = 16;              // tokens per tile half
  static_assert(BLK == 8 || BLK == 16 || BLK == 32, "block size");

  const int split = blockIdx.x;
  const int kvh = blockIdx.y;
  const int seq = blockIdx.z;
  const int num_seqs = gridDim.z;
  const int gqa = num_heads / num_kv_heads;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;   // lane group: tokens 4g..4g+3 of a block
  const int hc = lane & 15;  // query head within the group (column of S^T / O^T)

  // Block-table window in registers: lane l holds entry tbl_base + l of the row.  The first
  // window is requested before the sequence length is known (the row is max_num_blocks_per_seq
  // entries wide whatever the length), so the length -> table -> K/V chain of dependent loads
  // at the start of the kernel is one link shorter, and the main loop picks its block ids with
  // v_readlane instead of a dependent global load per pair.
  const int32_t* table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  int tbl_base = 0;
  int tbl_reg = lane < max_num_blocks_per_seq ? table[lane] : 0;

  // a length beyond what the table row can address is clamped (never read past the row) and
  // reported through the diagnostic word
  const int raw_len = seq_lens[seq];
  const int seq_len = min(raw_len, max_num_blocks_per_seq * BLK);
  if (raw_len > max_num_blocks_per_seq * BLK && threadIdx.x == 0 && split == 0 && kvh == 0)
    atomicOr(&decode_diag_word, (unsigned)PE_DIAG_DECODE_LEN_CLAMPED);
  // Fused KV write: the step's new token (position seq_len-1) is taken from new_key/new_value
  // directly and only the seq_len-1 older tokens are read from the cache, so the store of the
  // new row into the cache needs no ordering against this kernel's loads.
  bool has_new = false;
  int64_t new_slot = -1;
  if (new_key != nullptr && seq_len > 0) {
    new_slot = new_slots[seq];
    has_new = new_slot >= 0 && new_slot < num_slots;
  }
  const int cached_len = has_new ? seq_len - 1 : seq_len;
  const int n_blocks = (cached_len + BLK - 1) / BLK;
  const int n_pairs = (cached_len + 2 * BS - 1) / (2 * BS);  // 32-token tiles
  const int pairs_per_split = (n_pairs + num_splits - 1) / num_splits;
  const int pair_begin = split * pairs_per_split;
  const int pair_end = min(n_pairs, pair_begin + pairs_per_split);
  auto block_id = [&](int i) -> int64_t {  // i is wave-uniform and never decreases
    if (i >= tbl_base + 64) {
      tbl_base = i & ~63;
      const int e = tbl_base + lane;
      tbl_reg = e < max_num_blocks_per_seq ? table[e] : 0;
    }
    return __builtin_amdgcn_readlane(tbl_reg, i - tbl_base);
  };

  // ---- Q fragments: B operand, lane holds Q[head hc][32*ks + 8g .. +8) ------------
  vec8 qf[KS];
  {
    const int head = kvh * gqa + hc;
    const T* qp = query + (int64_t)seq * q_stride + (int64_t)head * HEAD_SIZE + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (hc < gqa) raw = *reinterpret_cast<const u32x4*>(qp + 32 * ks);
      qf[ks] = __builtin_bit_cast(vec8, raw);
    }
  }
  // Rotation of a row held as fragments (lane: elements 32*ks + 8g .. +8 of every k-step), with
  // the arithmetic of the stand-alone kernel (rot_pair), so that the fused step is bit-identical
  // to rotary embedding followed by pe_decode_step.  The cosines and sines of the lane's elements
  // are requested here, in front of the first K/V tiles, and consumed once those are under way
  // (the position -> cos/sin chain of dependent loads then runs beside the table -> K/V chain
  // instead of in front of it).  NeoX, rot_dim = 64*hs: cs[ks] = cosines and cs[hs + ks] = sines
  // of k-step ks < hs, which pairs with k-step ks + hs; GPT-J: cs[ks] = 4 cosines, 4 sines of the
  // chunk's adjacent pairs.  The loops are unrolled over every hs the head size admits, so all
  // fragment indices are constants.
  auto rope_load = [&](u32x4 (&cs)[KS]) {
    // positions == nullptr: the table already holds one row per sequence of this launch (gathered
    // once per step for all layers), so the cosines and sines are requested together with Q
    // instead of behind a dependent load of the position
    int64_t pos = seq;
    if (rope_positions != nullptr) {
      pos = rope_positions[seq];
      pos = pos < 0 ? 0 : (pos >= rope_max_position ? rope_max_position - 1 : pos);
    }
    const T* cs_row = rope_cos_sin + pos * rot_dim;
    const int half = rot_dim / 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) cs[ks] = u32x4{0u, 0u, 0u, 0u};
    if (rope_neox) {
#pragma unroll
      for (int hs = 1; hs <= KS / 2; ++hs) {
        if (rot_dim != 64 * hs) continue;
#pragma unroll
        for (int ks = 0; ks < hs; ++ks) {
          cs[ks] = *reinterpret_cast<const u32x4*>(cs_row + 32 * ks + 8 * g);
          cs[hs + ks] = *reinterpret_cast<const u32x4*>(cs_row + half + 32 * ks + 8 * g);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d0 = 32 * ks + 8 * g;  // rot_dim is a multiple of 16: a chunk is rotated whole
        if (d0 >= rot_dim) continue;
        const u32x2 c4 = *reinterpret_cast<const u32x2*>(cs_row + d0 / 2);
        const u32x2 s4 = *reinterpret_cast<const u32x2*>(cs_row + half + d0 / 2);
        cs[ks] = u32x4{c4[0], c4[1], s4[0], s4[1]};
      }
    }
  };
  auto rope_apply = [&](vec8 (&f)[KS], const u32x4 (&cs)[KS]) {
    if (rope_neox) {
#pragma unroll
      for (int hs = 1; hs <= KS / 2; ++hs) {
        if (rot_dim != 64 * hs) continue;
#pragma unroll
        for (int ks = 0; ks < hs; ++ks) {
          const vec8 c8 = __builtin_bit_cast(vec8, cs[ks]);
          const vec8 s8 = __builtin_bit_cast(vec8, cs[hs + ks]);
          vec8 x = f[ks], y = f[ks + hs];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            T ox, oy;
            rot_pair<T>(to_f32(x[j]), to_f32(y[j]), to_f32(c8[j]), to_f32(s8[j]), ox, oy);
            x[j] = ox;
            y[j] = oy;
          }
          f[ks] = x;
          f[ks + hs] = y;
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (32 * ks + 8 * g >= rot_dim) continue;
        const vec8 cs8 = __builtin_bit_cast(vec8, cs[ks]);  // [0..4) cosines, [4..8) sines
        vec8 x = f[ks];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          T ox, oy;
          rot_pair<T>(to_f32(x[j]), to_f32(x[j + 1]), to_f32(cs8[j / 2]), to_f32(cs8[4 + j / 2]), ox, oy);
          x[j] = ox;
          x[j + 1] = oy;
        }
        f[ks] = x;
      }
    }
  };
  static_assert(!ROPE || HEAD_SIZE <= 128, "the rotating form is built for head sizes 64 and 128");
  u32x4 cs_q[KS];
  bool q_rotated = !ROPE;
  if constexpr (ROPE) rope_load(cs_q);
  const float slope =
      (alibi_slopes != nullptr && hc < gqa) ? alibi_slopes[kvh * gqa + hc] : 0.f;

  float m_run = NEG_BIG;  // running max of head hc (same in the 4 lanes sharing hc)
  float l_run = 0.f;      // this lane's share of the running sum
  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t head_off = (int64_t)kvh * kv_head_stride;
  const int k_lane_off = lane * 8;               // elements
  const int v_lane_off = hc * BS + 4 * g;        // elements, + dt*16*BS

  auto issue = [&](DecTile<T, HEAD_SIZE>& t, int pair) {
    if constexpr (BLK == 32) {
      // one block holds the tile: half b = tokens 16b .. 16b+15 of every chunk row / value row
      const int64_t pa = block_id(pair);
      const T* kp = key_cache + pa * kv_block_stride + head_off + g * (BLK * 8) + hc * 8;
      const T* vp = value_cache + pa * kv_block_stride + head_off + hc * BLK + 4 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        t.k[0][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kp + ks * (4 * BLK * 8))));
        t.k[1][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kp + ks * (4 * BLK * 8) + BS * 8)));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        t.v[0][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vp + dt * 16 * BLK)));
        t.v[1][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vp + dt * 16 * BLK + BS)));
      }
      return;
    }
    if constexpr (BLK == 8) {
      // four blocks per tile (past the end: the last block again, masked); the K row of token hc
      // of a half lies in the half's first block for hc < 8, the value tokens 4g .. 4g+3 in its
      // first block for g < 2
      const int last = n_blocks - 1;
      const int64_t p0 = block_id(min(4 * pair, last));
      const int64_t p1 = block_id(min(4 * pair + 1, last));
      const int64_t p2 = block_id(min(4 * pair + 2, last));
      const int64_t p3 = block_id(min(4 * pair + 3, last));
      const int koff = g * (BLK * 8) + (hc & 7) * 8;
      const int voff = hc * BLK + 4 * (g & 1);
      const T* ka = key_cache + (hc < 8 ? p0 : p1) * kv_block_stride + head_off + koff;
      const T* kb = key_cache + (hc < 8 ? p2 : p3) * kv_block_stride + head_off + koff;
      const T* va = value_cache + (g < 2 ? p0 : p1) * kv_block_stride + head_off + voff;
      const T* vb = value_cache + (g < 2 ? p2 : p3) * kv_block_stride + head_off + voff;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        t.k[0][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ka + ks * (4 * BLK * 8))));
        t.k[1][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + ks * (4 * BLK * 8))));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        t.v[0][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(va + dt * 16 * BLK)));
        t.v[1][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vb + dt * 16 * BLK)));
      }
      return;
    }
    const int ia = 2 * pair;
    const int ib = min(2 * pair + 1, n_blocks - 1);  // odd tail: re-read A's neighbour, masked
    const int64_t pa = block_id(ia);
    const int64_t pb = block_id(ib);
    const T* ka = key_cache + pa * kv_block_stride + head_off + k_lane_off;
    const T* kb = key_cache + pb * kv_block_stride + head_off + k_lane_off;
    const T* va = value_cache + pa * kv_block_stride + head_off + v_lane_off;
    const T* vb = value_cache + pb * kv_block_stride + head_off + v_lane_off;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      t.k[0][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ka + ks * 512)));
      t.k[1][ks] = __builtin_bit_cast(vec8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + ks * 512)));
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      t.v[0][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(va + dt * 16 * BS)));
      t.v[1][dt] = __builtin_bit_cast(vec4, __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vb + dt * 16 * BS)));
    }
  };

  // tile tokens are tok_base + b*16 + 4g + r; tokens >= limit are masked
  auto compute = [&](const DecTile<T, HEAD_SIZE>& t, int tok_base, int limit) {
    f32x4 s[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      s[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[b] = TR::mfma16(t.k[b][ks], qf[ks], s[b]);
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int tok0 = tok_base + b * BS + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = tok0 + r;
        float x = s[b][r] * scale + slope * (float)(tok - seq_len + 1);
        x = tok < limit ? x : NEG_BIG;
        s[b][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    vec8 pf;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[b][r] - m_new);
        psum += p;
        pf[b * 4 + r] = (T)p;
      }
    }
    l_run = l_run * alpha + psum;
    // slots past seq_len in the last block may hold anything (NaN included): their V is
    // replaced by zero, as the reference does (attention_kernels.cuh:420-430)
    const bool tail = tok_base + 2 * BS > limit;  // wave-uniform
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      acc[dt] *= alpha;
      vec8 vf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vf[j] = t.v[0][dt][j];
        vf[4 + j] = t.v[1][dt][j];
      }
      if (tail) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int tok = tok_base + (j >> 2) * BS + 4 * g + (j & 3);
          vf[j] = tok < limit ? vf[j] : (T)0.f;
        }
      }
      acc[dt] = TR::mfma16(vf, pf, acc[dt]);
    }
  };

  // ---- main loop: waves interleave over the split's block pairs ----------------------
  if constexpr (HEAD_SIZE <= 128 && PE_DEC_TILES == 2 && WAVES < 16) {
    // two tiles in flight per wave (the next pair's 16 KB is requested before the current
    // pair is consumed)
    DecTile<T, HEAD_SIZE> t0, t1;
    int p = pair_begin + wave;
    if (p < pair_end) {
      issue(t0, p);
      if constexpr (ROPE) {
        rope_apply(qf, cs_q);
        q_rotated = true;
      }
      while (true) {
        if (p + WAVES < pair_end) issue(t1, p + WAVES);
        compute(t0, 2 * p * BS, cached_len);
        p += WAVES;
        if (p >= pair_end) break;
        if (p + WAVES < pair_end) issue(t0, p + WAVES);
        compute(t1, 2 * p * BS, cached_len);
        p += WAVES;
        if (p >= pair_end) break;
      }
    }
  } else {
    // head_size 256: one tile per wave fits the register file; occupancy hides latency
    DecTile<T, HEAD_SIZE> t0;
    for (int p = pair_begin + wave; p < pair_end; p += WAVES) {
      issue(t0, p);
      if constexpr (ROPE) {
        if (!q_rotated) {
          rope_apply(qf, cs_q);
          q_rotated = true;
        }
      }
      compute(t0, 2 * p * BS, cached_len);
    }
  }

  // ---- the step's new token: attention contribution from registers + store into the cache --
  if (has_new && split == 0 &&
      wave == ((pair_end > pair_begin ? pair_end - pair_begin : 0) % WAVES)) {
    const T* ksrc = new_key + (int64_t)seq * new_k_stride + (int64_t)kvh * HEAD_SIZE;
    const T* vsrc = new_value + (int64_t)seq * new_v_stride + (int64_t)kvh * HEAD_SIZE;
    DecTile<T, HEAD_SIZE> tn;
    // the new key row sits at the query's position: the same cosines and sines, requested again
    // together with the row itself (they were not kept across the main loop)
    u32x4 cs_n[KS];
    if constexpr (ROPE) rope_load(cs_n);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (hc == 0) raw = *reinterpret_cast<const u32x4*>(ksrc + 32 * ks + 8 * g);  // row = token 0
      tn.k[0][ks] = __builtin_bit_cast(vec8, raw);
      tn.k[1][ks] = __builtin_bit_cast(vec8, u32x4{0u, 0u, 0u, 0u});
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      vec4 v4 = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
      if (g == 0) v4[0] = vsrc[16 * dt + hc];  // k-slot (g=0, j=0) = token 0 of the tile
      tn.v[0][dt] = v4;
      tn.v[1][dt] = vec4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    }
    if constexpr (ROPE) {
      if (!q_rotated) rope_apply(qf, cs_n);  // a wave that had no block pair of its own
      rope_apply(tn.k[0], cs_n);             // zero rows (hc != 0) stay zero
    }
    compute(tn, seq_len - 1, seq_len);
    // store the row into its cache slot (same bytes pe_reshape_and_cache writes)
    const int64_t blk = new_slot / BLK;
    const int off = (int)(new_slot % BLK);
    T* kdst = key_cache_w + blk * kv_block_stride + (int64_t)kvh * kv_head_stride;
    T* vdst = value_cache_w + blk * kv_block_stride + (int64_t)kvh * kv_head_stride;
    if constexpr (ROPE) {
      // the rotated row only exists in the fragments: lane (g, hc = 0) holds 16-byte chunk 4*ks + g
      if (hc == 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          *reinterpret_cast<u32x4*>(kdst + ((4 * ks + g) * BLK + off) * 8) =
              __builtin_bit_cast(u32x4, tn.k[0][ks]);
      }
    } else {
      for (int c = lane; c < HEAD_SIZE / 8; c += 64)
        *reinterpret_cast<u32x4*>(kdst + (c * BLK + off) * 8) =
            *reinterpret_cast<const u32x4*>(ksrc + c * 8);
    }
    for (int d = lane; d < HEAD_SIZE; d += 64) vdst[d * BLK + off] = vsrc[d];
  }

  // ---- merge the 4 lane groups' sums, then the waves through LDS -------------------
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);

  // dynamic LDS: o[WAVES][gqa][HEAD_SIZE + 4], m[WAVES][16], l[WAVES][16] (f32)
  extern __shared__ __attribute__((aligned(16))) float dec_smem[];
  constexpr int OROW = HEAD_SIZE + 4;
  float* sm_o = dec_smem;
  float* sm_m = dec_smem + WAVES * gqa * OROW;
  float* sm_l = sm_m + WAVES * 16;
  if (hc < gqa) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<f32x4*>(&sm_o[(wave * gqa + hc) * OROW + 16 * dt + 4 * g]) = acc[dt];
    if (g == 0) {
      sm_m[wave * 16 + hc] = m_run;
      sm_l[wave * 16 + hc] = l_run;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < gqa * HEAD_SIZE; i += WAVES * 64) {
    const int h = i / HEAD_SIZE, d = i % HEAD_SIZE;
    float m_tot = sm_m[h];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) m_tot = fmaxf(m_tot, sm_m[w * 16 + h]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float f = __expf(sm_m[w * 16 + h] - m_tot);
      o += f * sm_o[(w * gqa + h) * OROW + d];
      l += f * sm_l[w * 16 + h];
    }
    const int head = kvh * gqa + h;
    if (num_splits == 1) {
      out[((int64_t)seq * num_heads + head) * HEAD_SIZE + d] = (T)(o * (1.f / (l + 1e-6f)));
    } else {
      float* w = ws + (((int64_t)split * num_seqs + seq) * num_heads + head) * ws_stride(HEAD_SIZE);
      if (d == 0) {
        w[0] = m_tot;
        w[1] = l;
      }
      w[2 + d] = o;
    }
  }
}

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
  return ((size_t)waves * gqa * (head_size + 4) + 2 * (size_t)waves * 16) * 4;
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
  if (head_size != 64 && head_size != 128 && head_size != 256) return false;
  // 8- and 32-token blocks (the page sizes of scripts/diff_page_sizes.sh:16) are built for the
  // head sizes of the models that study runs (64, 128)
  if (block_size != 16 && head_size == 256) return false;
  if (num_heads / num_kv_heads > 16) return false;
  if (q_stride % 8 || kv_block_stride % 8 || kv_head_stride % 8) return false;
  return true;
}

int decode_diag_fetch(unsigned int* out, int reset) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(decode_diag_word));
  if (e == hipSuccess) e = hipMemcpy(out, p, sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset && *out != 0) e = hipMemset(p, 0, sizeof(unsigned int));
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "decode diagnostics: %s", hipGetErrorString(e));
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
#define PE_LAUNCH_DEC_WR(T, HS, W, R)                                                              \
  if (lds > 64 * 1024)                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&paged_decode_mfma_kernel<T, HS, W, R>), \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
  hipLaunchKernelGGL((paged_decode_mfma_kernel<T, HS, W, R>), grid, dim3(W * 64), lds, st, (T*)out, \
                     ws, (const T*)query, (const T*)key_cache, (const T*)value_cache,              \
                     block_tables, seq_lens, alibi_slopes, num_heads, num_kv_heads,                \
                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, scale,     \
                     splits, (const T*)new_key, (const T*)new_value, new_slots, new_k_stride,       \
                     new_v_stride, num_slots, (T*)key_cache, (T*)value_cache,                      \
                     R ? rope->positions : nullptr, R ? (const T*)rope->cos_sin : nullptr,         \
                     R ? rope->rot_dim : 0, R ? rope->is_neox : 0, R ? rope->max_position : 0)
#define PE_LAUNCH_DEC_W(T, HS, W) PE_LAUNCH_DEC_WR(T, HS, W, false)
#define PE_LAUNCH_DEC_B(T, HS, W, B)                                                               \
  if (lds > 64 * 1024)                                                                             \
    (void)hipFuncSetAttribute(                                                                     \
        reinterpret_cast<const void*>(&paged_decode_mfma_kernel<T, HS, W, false, B>),              \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
  hipLaunchKernelGGL((paged_decode_mfma_kernel<T, HS, W, false, B>), grid, dim3(W * 64), lds, st,  \
                     (T*)out, ws, (const T*)query, (const T*)key_cache, (const T*)value_cache,     \
                     block_tables, seq_lens, alibi_slopes, num_heads, num_kv_heads,                \
                     max_num_blocks_per_seq, q_stride, kv_block_stride, kv_head_stride, scale,     \
                     splits, (const T*)new_key, (const T*)new_value, new_slots, new_k_stride,       \
                     new_v_stride, num_slots, (T*)key_cache, (T*)value_cache, nullptr, nullptr, 0, \
                     0, 0)
#define PE_LAUNCH_DEC(T, HS)                               \
  do {                                                     \
    if (block_size != 16) {                                \
      if constexpr (HS <= 128) {                           \
        if (block_size == 32) {                            \
          if (waves == 8) {                                \
            PE_LAUNCH_DEC_B(T, HS, 8, 32);                 \
          } else {                                         \
            PE_LAUNCH_DEC_B(T, HS, DEC_WAVES, 32);         \
          }                                                \
        } else {                                           \
          if (waves == 8) {                                \
            PE_LAUNCH_DEC_B(T, HS, 8, 8);                  \
          } else {                                         \
            PE_LAUNCH_DEC_B(T, HS, DEC_WAVES, 8);          \
          }                                                \
        }                                                  \
      }                                                    \
    } else if (rope != nullptr) {                          \
      if constexpr (HS <= 128) {                           \
        if (waves == 8) {                                  \
          PE_LAUNCH_DEC_WR(T, HS, 8, true);                \
        } else {                                           \
          PE_LAUNCH_DEC_WR(T, HS, DEC_WAVES, true);        \
        }                                                  \
      }                                                    \
    } else if (waves == 16) {                              \
      if constexpr (HS <= 128) {                           \
        PE_LAUNCH_DEC_W(T, HS, 16);                        \
      }                                                    \
    } else if (waves == 8) {                               \
      PE_LAUNCH_DEC_W(T, HS, 8);                           \
    } else {                                               \
      PE_LAUNCH_DEC_W(T, HS, DEC_WAVES);                   \
    }                                                      \
  } while (0)
    PE_DISPATCH_16(dtype, T, {
      if (head_size == 64) PE_LAUNCH_DEC(T, 64);
      else if (head_size == 128) PE_LAUNCH_DEC(T, 128);
      else PE_LAUNCH_DEC(T, 256);
    });
#undef PE_LAUNCH_DEC
#undef PE_LAUNCH_DEC_B
#undef PE_LAUNCH_DEC_W
#undef PE_LAUNCH_DEC_WR
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
             "decode_step_rope needs the MFMA decode path (16-bit dtype, block_size 16, head_size "
             "64/128, strides multiples of 8): use pe_rope_reshape_and_cache + pe_paged_decode");
  PE_REQUIRE(block_size == 16, PE_ERR_UNSUPPORTED,
             "decode_step_rope is built for 16-token blocks; use pe_rope_reshape_and_cache + "
             "pe_paged_decode");
  PE_REQUIRE(head_size <= 128, PE_ERR_UNSUPPORTED,
             "decode_step_rope: head_size %d has no registers left for the rotation; use "
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
