// The matrix-core decode kernel (template) shared by the translation units that instantiate it:
// paged_decode.hip (head sizes 64, 128, 256, all block sizes, the rotating form) and
// paged_decode_hs.hip (the other head sizes of the reference's list).  See paged_decode.hip for
// the description of the data layout and the algorithm.
#pragma once
#include <cfloat>

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

// A K/V fragment load with or without the non-temporal hint
template <bool NT, typename V>
__device__ __forceinline__ V ldg(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <typename T, int HEAD_SIZE>
struct DecTile {
  typename Traits<T>::vec8 k[2][(HEAD_SIZE + 31) / 32];
  typename Traits<T>::vec4 v[2][(HEAD_SIZE + 15) / 16];
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
  // Head sizes that are not a multiple of 32 (80, 112, 120: the reference's list,
  // vllm/attention/ops/paged_attn.py:37) run with the last k-step / d-tile partly empty: the
  // lanes whose 16-byte chunk (K, Q) or row (V) lies past the head load nothing and hold zeros.
  constexpr int KS = (HEAD_SIZE + 31) / 32;  // k-steps of S^T = K.Q^T
  constexpr int DT = (HEAD_SIZE + 15) / 16;  // d-tiles of O^T
  constexpr bool RAGGED_K = HEAD_SIZE % 32 != 0;
  constexpr bool RAGGED_V = HEAD_SIZE % 16 != 0;
  constexpr int BS = 16;              // tokens per tile half
  // K/V loads of the throughput shape carry the non-temporal hint: a launch that streams more
  // than the caches hold keeps nothing worth keeping (measured: 35.4 against 39.7 us at 200 MB
  // per launch).  The latency shapes serve launches of a few megabytes, which run faster with
  // plain loads (one sequence of kv 776: 7.7 against 8.3 us).
  constexpr bool STREAM = WAVES < 8;
  static_assert(BLK == 8 || BLK == 16 || BLK == 32, "block size");
  static_assert(HEAD_SIZE % 8 == 0 && (BLK == 16 || HEAD_SIZE % 32 == 0), "head size");

  const int split = blockIdx.x;
  const int kvh = blockIdx.y;
  const int seq = blockIdx.z;
  const int num_seqs = gridDim.z;
  const int gqa = num_heads / num_kv_heads;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;   // lane group: tokens 4g..4g+3 of a block
  const int hc = lane & 15;  // query head within the group (column of S^T / O^T)
  // this lane's chunk of the last k-step / row of the last d-tile lies inside the head
  const bool k_last_ok = !RAGGED_K || (4 * (KS - 1) + g) * 8 < HEAD_SIZE;
  const bool v_last_ok = !RAGGED_V || 16 * (DT - 1) + hc < HEAD_SIZE;

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
      if (hc < gqa && (ks < KS - 1 || k_last_ok)) raw = *reinterpret_cast<const u32x4*>(qp + 32 * ks);
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
  static_assert(!ROPE || (HEAD_SIZE <= 128 && HEAD_SIZE % 32 == 0),
                "the rotating form is built for head sizes 64 and 128");
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
        t.k[0][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(kp + ks * (4 * BLK * 8))));
        t.k[1][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(kp + ks * (4 * BLK * 8) + BS * 8)));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        t.v[0][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(vp + dt * 16 * BLK)));
        t.v[1][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(vp + dt * 16 * BLK + BS)));
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
        t.k[0][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(ka + ks * (4 * BLK * 8))));
        t.k[1][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(kb + ks * (4 * BLK * 8))));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        t.v[0][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(va + dt * 16 * BLK)));
        t.v[1][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(vb + dt * 16 * BLK)));
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
      if (RAGGED_K && ks == KS - 1 && !k_last_ok) {
        t.k[0][ks] = t.k[1][ks] = __builtin_bit_cast(vec8, u32x4{0u, 0u, 0u, 0u});
        continue;
      }
      t.k[0][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(ka + ks * 512)));
      t.k[1][ks] = __builtin_bit_cast(vec8, ldg<STREAM>(reinterpret_cast<const u32x4*>(kb + ks * 512)));
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (RAGGED_V && dt == DT - 1 && !v_last_ok) {
        t.v[0][dt] = t.v[1][dt] = __builtin_bit_cast(vec4, u32x2{0u, 0u});
        continue;
      }
      t.v[0][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(va + dt * 16 * BS)));
      t.v[1][dt] = __builtin_bit_cast(vec4, ldg<STREAM>(reinterpret_cast<const u32x2*>(vb + dt * 16 * BS)));
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
      if (hc == 0 && (ks < KS - 1 || k_last_ok))
        raw = *reinterpret_cast<const u32x4*>(ksrc + 32 * ks + 8 * g);  // row = token 0
      tn.k[0][ks] = __builtin_bit_cast(vec8, raw);
      tn.k[1][ks] = __builtin_bit_cast(vec8, u32x4{0u, 0u, 0u, 0u});
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      vec4 v4 = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
      if (g == 0 && (dt < DT - 1 || v_last_ok)) v4[0] = vsrc[16 * dt + hc];  // k-slot (g=0, j=0) = token 0
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

  // dynamic LDS: o[WAVES][gqa][16 * DT + 4], m[WAVES][16], l[WAVES][16] (f32)
  extern __shared__ __attribute__((aligned(16))) float dec_smem[];
  constexpr int OROW = 16 * DT + 4;
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

// Arguments of one launch of the MFMA kernel, as the dispatch of pe_paged_decode assembles them
struct DecLaunch {
  void* out;
  float* ws;
  const void *query, *key_cache, *value_cache;
  const int32_t *block_tables, *seq_lens;
  const float* alibi_slopes;
  int num_heads, num_kv_heads, max_num_blocks_per_seq;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  float scale;
  int num_splits;
  const void *new_key, *new_value;
  const int64_t* new_slots;
  int64_t new_k_stride, new_v_stride, num_slots;
  const RopeArgs* rope;
  dim3 grid;
  size_t lds;
  hipStream_t stream;
};

template <typename T, int HS, int W, bool R, int B>
inline void launch_dec(const DecLaunch& a) {
  if (a.lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&paged_decode_mfma_kernel<T, HS, W, R, B>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
  hipLaunchKernelGGL((paged_decode_mfma_kernel<T, HS, W, R, B>), a.grid, dim3(W * 64), a.lds, a.stream,
                     (T*)a.out, a.ws, (const T*)a.query, (const T*)a.key_cache, (const T*)a.value_cache,
                     a.block_tables, a.seq_lens, a.alibi_slopes, a.num_heads, a.num_kv_heads,
                     a.max_num_blocks_per_seq, a.q_stride, a.kv_block_stride, a.kv_head_stride, a.scale,
                     a.num_splits, (const T*)a.new_key, (const T*)a.new_value, a.new_slots, a.new_k_stride,
                     a.new_v_stride, a.num_slots, (T*)a.key_cache, (T*)a.value_cache,
                     R ? a.rope->positions : nullptr, R ? (const T*)a.rope->cos_sin : nullptr,
                     R ? a.rope->rot_dim : 0, R ? a.rope->is_neox : 0, R ? a.rope->max_position : 0);
}

// this translation unit's copy of the diagnostic word
static int decode_diag_fetch_tu(unsigned int* out, int reset) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(decode_diag_word));
  if (e == hipSuccess) e = hipMemcpy(out, p, sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset && *out != 0) e = hipMemset(p, 0, sizeof(unsigned int));
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "decode diagnostics: %s", hipGetErrorString(e));
  return PE_OK;
}

// paged_decode_hs.hip: head sizes 32, 80, 96, 112, 120, 192 (16-token blocks, 4 or 8 waves)
int launch_decode_other_heads(const DecLaunch& a, int head_size, int waves, int dtype);
int decode_hs_diag_fetch(unsigned int* out, int reset);
// paged_decode_blk.hip: head sizes 32, 96, 192, 256 at 8- and 32-token blocks (4 or 8 waves)
int launch_decode_other_blocks(const DecLaunch& a, int head_size, int block_size, int waves, int dtype);
int decode_blk_diag_fetch(unsigned int* out, int reset);

}  // namespace pe
