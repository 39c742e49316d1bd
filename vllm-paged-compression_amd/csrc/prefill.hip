// Varlen causal prefill attention on gfx950 (flash-style, no L_q x L_kv matrix in memory).
//
// Mask: query row i of a request attends kv rows 0..min(i, L_kv-1) — the top-left aligned
// lower triangle that scaled_dot_product_attention(is_causal=True) applies, which is what
// the reference runs when the prompt was pruned before attention (L_kv < L_q).
// Grouped KV heads are addressed directly (no repeat_interleave copy).
//
// MFMA path (16-bit types; head sizes 64 and 128 as they are, 256 with 8-wave workgroups, and the
// other sizes of the reference's list - 32, 80, 96, 112, 120, 192, vllm/attention/ops/paged_attn.py:37 -
// in the next larger of those row pitches with the chunks past the head never used):
// a workgroup of 4 waves owns 128 query rows of
// one (request, head); K/V tiles of 64 tokens stream global -> LDS directly (LDS-DMA loads,
// double-buffered: tile j+1 is in flight during the math of tile j).
//   S^T = K.Q^T   v_mfma_f32_16x16x32: A = K rows from LDS (ds_read_b128, XOR-swizzled
//                 chunks), B = Q fragments held in registers; a lane ends up with 4 tokens
//                 x 1 query row, so the row max/sum is a register reduction + 2 shuffles.
//   O^T += V^T.P^T  A = V^T read with ds_read_b64_tr_b16 from the row-major V tile, B = P
//                 taken straight from the S^T accumulators (k-slot order chosen to match).
// The finished O tile goes through LDS so that global stores are whole 16-byte row chunks.
#include "pe_common.h"

#include <cstdlib>
#include <type_traits>

namespace pe {

constexpr int PF_BN = 64;   // kv rows per tile
constexpr int PF_DEFAULT_SHAPE = 24;
constexpr float PF_NEG = -1.0e30f;
constexpr float PF_LAZY_LOG2 = 6.0f;  // weights may reach 2^6 before the reference point moves

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int D>
__device__ __forceinline__ int swz_off(int row, int chunk) {
  // byte offset of 16-byte chunk `chunk` of row `row` in the [rows][D] 16-bit K tile: the 16 rows
  // a ds_read_b128 lane group touches land on 16 different 16-byte slots of the 256-byte bank row
  constexpr int CPR = D / 8;
  const int x = (CPR >= 16) ? (row & 15) : ((row >> 1) & 7);
  return row * (D * 2) + ((chunk ^ x) << 4);
}

template <int D>
__device__ __forceinline__ int swz_off_v(int row, int chunk) {
  // V tile: a 32-lane half of ds_read_b64_tr_b16 takes 8 consecutive rows x 32 bytes (one chunk
  // pair per row); XOR-ing the pair index with the row puts the 8 rows on 8 different 32-byte
  // spans of the bank row
  constexpr int CPR = D / 8;
  const int x = (CPR >= 16) ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
  return row * (D * 2) + ((chunk ^ x) << 4);
}

// D: row pitch of the K/V tiles in LDS and width of the fragment loops (64, 128 or 256 elements);
// the head itself has head_size <= D elements (a multiple of 8).  NR: 16-row query tiles per wave;
// WAVES: waves per workgroup (query rows per workgroup = 16 * NR * WAVES); WPE: waves per SIMD the
// register budget is cut for.
// max over the four lanes l, l^16, l^32, l^48 with the row-swap permutes (VALU, no LDS round
// trip as a ds_bpermute shuffle would take): v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second, v_permlane16_swap the odd 16-lane rows of the
// first with the even rows of the second; with both operands equal the two results hold, in
// every lane, the lane's own value and its partner's.
__device__ __forceinline__ float max_xor_16_32(float x) {
  unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(x);
  auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <typename T, int D, int NR, int WAVES, int WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void prefill_mfma_kernel(
    T* __restrict__ out, const T* __restrict__ q, const T* __restrict__ k,
    const T* __restrict__ v, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_kv,
    int num_heads, int num_kv_heads, int head_size, int64_t q_stride, int64_t k_stride,
    int64_t v_stride, int64_t out_stride, float scale_log2e) {
  using TR = Traits<T>;
  using vec8 = typename TR::vec8;
  constexpr int KS = D / 32;
  constexpr int DT = D / 16;
  constexpr int CPR = D / 8;                        // 16-byte chunks per row
  constexpr int TILE_BYTES = PF_BN * D * 2;         // one K or V tile
  constexpr int WROWS = 16 * NR;                    // query rows per wave
  constexpr int BM = WROWS * WAVES;                 // query rows per workgroup
  static_assert(BM % PF_BN == 0 && BM * D * 2 <= 4 * TILE_BYTES, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][K|V][tile]
  auto k_tile = [&](int buf) { return smem + buf * 2 * TILE_BYTES; };
  auto v_tile = [&](int buf) { return smem + buf * 2 * TILE_BYTES + TILE_BYTES; };

  // grid (head, request, query tile): workgroups are dispatched x-fastest, so every request's
  // heaviest (last) query tiles start first and the short ones fill the tail.  Consecutive
  // workgroup ids go to consecutive XCDs; the (head, request) plane is re-enumerated so that all
  // query heads of one (request, KV head) - which read the same K/V - get ids that are equal
  // modulo 8 and share one XCD's L2 (needs num_kv_heads * num_requests to be a multiple of 8).
  const int gqa = num_heads / num_kv_heads;
  const int lid = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int n_groups = num_kv_heads * (int)gridDim.y;
  int grp, hg;
  if (n_groups % 8 == 0) {
    grp = (lid / (8 * gqa)) * 8 + lid % 8;
    hg = (lid / 8) % gqa;
  } else {
    grp = lid / gqa;
    hg = lid % gqa;
  }
  const int seq = grp / num_kv_heads;
  const int kvh = grp % num_kv_heads;
  const int head = kvh * gqa + hg;
  const int q0 = cu_q[seq], lq = cu_q[seq + 1] - q0;
  const int kv0 = cu_kv[seq], lkv = cu_kv[seq + 1] - kv0;
  const int qt = (int)gridDim.z - 1 - (int)blockIdx.z;
  const int m0 = qt * BM;
  if (m0 >= lq) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int wrow0 = m0 + wave * WROWS;  // first query row of this wave
  const int hchunks = head_size >> 3;   // 16-byte chunks of a head row that exist (<= CPR)

  // ---- Q fragments (B operand): lane holds Q[row][32ks + 8g .. +8) for its row tiles ----
  vec8 qf[NR][KS];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int row = min(wrow0 + n * 16 + c, lq - 1);
    const T* qp = q + (int64_t)(q0 + row) * q_stride + (int64_t)head * head_size + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // a chunk past the head is zero in Q, so whatever its K slot holds contributes nothing
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (4 * ks + g < hchunks) raw = *reinterpret_cast<const u32x4*>(qp + 32 * ks);
      qf[n][ks] = __builtin_bit_cast(vec8, raw);
    }
  }

  // The row sums ride on the matrix pipe: one more MFMA per P fragment with an all-ones A operand
  // adds the (storage-type rounded) probabilities of a query row over the fragment's 32 keys, in
  // every lane of the row's column - no vector adds per element, no cross-lane sum at the end,
  // and numerator and denominator use the same rounded weights.
  float m_run[NR];
  f32x4 acc_l[NR];
  f32x4 acc[DT][NR];
  vec8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    m_run[n] = PF_NEG;
    acc_l[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int kv_end = min(lkv, m0 + BM);  // rows past the last query row are never attended
  const int n_tiles = (kv_end + PF_BN - 1) / PF_BN;

  // Staging: K/V tiles go global -> LDS directly (global_load_lds_dwordx4, no registers, no
  // ds_write).  One wave-instruction fills 1 KiB of LDS contiguously in lane order: RPI whole tile
  // rows, lane l -> row l / CPR, 16-byte slot l % CPR.  The XOR swizzle of the LDS image is applied
  // to the per-lane SOURCE chunk instead (slot s of a row holds chunk s ^ x(row)).  Wave w issues
  // instruction i for rows i * RSTEP + w * RPI ..., so x(row) does not depend on i (RSTEP is a
  // multiple of 16) and the per-lane source offset is one constant; tile and i go into the scalar
  // base.  Only the ragged last tile of a request clamps its rows per lane.  The loads are asm
  // statements the compiler does not count: stage_wait() + a barrier must separate them from the
  // reads of the tile.
  constexpr int RPI = 64 / CPR;            // tile rows per wave-instruction
  constexpr int RSTEP = WAVES * RPI;       // rows between a wave's consecutive instructions
  constexpr int NINST = PF_BN / RSTEP;     // instructions per wave per operand tile
  static_assert(RSTEP % 16 == 0 && NINST >= 1 && NINST * RSTEP == PF_BN, "staging shape");
  const int lrow = wave * RPI + lane / CPR;  // row within a RSTEP group
  const int lslot = lane % CPR;
  const char* k_seq = reinterpret_cast<const char*>(k + (int64_t)kv0 * k_stride + (int64_t)kvh * head_size);
  const char* v_seq = reinterpret_cast<const char*>(v + (int64_t)kv0 * v_stride + (int64_t)kvh * head_size);
  const uint32_t k_row_bytes = (uint32_t)k_stride * 2u, v_row_bytes = (uint32_t)v_stride * 2u;
  // source chunk for this lane's slot: the inverse of swz_off / swz_off_v (XOR is an involution)
  // (a slot whose chunk lies past the head is filled with the row's chunk 0 instead: finite data
  //  that meets the zero chunk of Q, or - in V - feeds output columns nobody stores)
  uint32_t k_chunk16 = (uint32_t)(swz_off<D>(lrow, lslot) - lrow * (D * 2));
  uint32_t v_chunk16 = (uint32_t)(swz_off_v<D>(lrow, lslot) - lrow * (D * 2));
  if ((int)(k_chunk16 >> 4) >= hchunks) k_chunk16 = 0;
  if ((int)(v_chunk16 >> 4) >= hchunks) v_chunk16 = 0;
  const uint32_t k_lane_off = (uint32_t)lrow * k_row_bytes + k_chunk16;
  const uint32_t v_lane_off = (uint32_t)lrow * v_row_bytes + v_chunk16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem +
                        (uint32_t)wave * 1024u;
  // M0 carries the LDS destination of an LDS-DMA load and nothing else in this kernel (the
  // compiler keeps no value in it: every access to M0 in the generated code is one of the moves
  // below).  A load may not read an SGPR that a vector
  // instruction wrote fewer than 5 issue slots earlier: the first load of a tile waits that out;
  // the operands of the following ones are scalar sums of values that existed before it.
  auto glds16 = [](const char* base, uint32_t lane_off, uint32_t lds_dst, auto first_tag) {
    if constexpr (decltype(first_tag)::value) {
      asm volatile(
          "s_nop 4\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %0, %1"
          :
          : "v"(lane_off), "s"(base), "s"(lds_dst)
          : "memory");
    } else {
      asm volatile(
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %0, %1"
          :
          : "v"(lane_off), "s"(base), "s"(lds_dst)
          : "memory");
    }
  };
  auto stage = [&](int tile, int buf) {
    const int t0 = tile * PF_BN;
    const char* kb = k_seq + (int64_t)t0 * k_row_bytes;
    const char* vb = v_seq + (int64_t)t0 * v_row_bytes;
    const uint32_t kd = lds0 + (uint32_t)buf * (2u * TILE_BYTES);
    const uint32_t vd = kd + TILE_BYTES;
    if (t0 + PF_BN <= lkv) {
      glds16(kb, k_lane_off, kd, std::true_type{});
      glds16(vb, v_lane_off, vd, std::false_type{});
#pragma unroll
      for (int i = 1; i < NINST; ++i) {
        glds16(kb + (int64_t)(i * RSTEP) * k_row_bytes, k_lane_off, kd + i * RSTEP * D * 2, std::false_type{});
        glds16(vb + (int64_t)(i * RSTEP) * v_row_bytes, v_lane_off, vd + i * RSTEP * D * 2, std::false_type{});
      }
    } else {
      const int last = lkv - 1 - t0;  // >= 0: a staged tile starts inside the request
#pragma unroll
      for (int i = 0; i < NINST; ++i) {
        const uint32_t row = (uint32_t)min(lrow + i * RSTEP, last);
        // (the row offsets are per-lane values, the scalar operands are the first load's)
        if (i == 0) {
          glds16(kb, row * k_row_bytes + k_chunk16, kd, std::true_type{});
        } else {
          glds16(kb, row * k_row_bytes + k_chunk16, kd + i * RSTEP * D * 2, std::false_type{});
        }
        glds16(vb, row * v_row_bytes + v_chunk16, vd + i * RSTEP * D * 2, std::false_type{});
      }
    }
  };
  auto stage_wait = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  auto compute = [&](int tile, int buf, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int n0 = tile * PF_BN;
    const char* kt = k_tile(buf);
    const char* vt = v_tile(buf);
    // ---- S^T = K.Q^T ---------------------------------------------------------------
    // The K fragments of row group sub+1 are requested before the MFMAs of group sub are issued
    // (the scheduling fences keep that order), so that an MFMA waits for reads issued a whole
    // group earlier instead of for the read right in front of it.
    f32x4 s[4][NR];
    vec8 kf[2][KS];
    auto load_k = [&](int sub, vec8* dst) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        dst[ks] = __builtin_bit_cast(
            vec8, *reinterpret_cast<const u32x4*>(kt + swz_off<D>(sub * 16 + c, 4 * ks + g)));
    };
    load_k(0, kf[0]);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      if (sub + 1 < 4) load_k(sub + 1, kf[(sub + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NR; ++n) s[sub][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int n = 0; n < NR; ++n) s[sub][n] = TR::mfma16(kf[sub & 1][ks], qf[n][ks], s[sub][n]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- mask, online softmax (the scale rides in the exponent's fma; maxima are kept in
    //      scaled log2 units, which needs scale > 0 - the launcher guarantees it) ----------
    vec8 pf[NR][2];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      float mx = PF_NEG;
      if constexpr (MASKED) {
        const int qrow = wrow0 + n * 16 + c;
        const int limit = min(qrow, lkv - 1) - n0 - 4 * g;  // last attended (sub*16 + r)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[sub][n][r] = (sub * 16 + r <= limit) ? s[sub][n][r] : PF_NEG;
      }
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][n][r]);
      mx = max_xor_16_32(mx);
      // The reference point of the exponentials only has to be NEAR the row's maximum: it moves
      // (and the accumulators are rescaled) when some row of the wave outgrew it by more than
      // 2^PF_LAZY_LOG2, so the weights stay below that factor and the rescale, frequent with a
      // strict running maximum, becomes rare.  The normalised output does not depend on it.
      const float m_tile = mx * scale_log2e;
      if (__builtin_amdgcn_ballot_w64(m_tile > m_run[n] + PF_LAZY_LOG2) != 0) {  // wave-uniform
        const float m_new = fmaxf(m_run[n], m_tile);
        const float alpha = __builtin_amdgcn_exp2f(m_run[n] - m_new);
        m_run[n] = m_new;
        acc_l[n] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt][n] *= alpha;
      }
      const float m_new = m_run[n];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][n][r], scale_log2e, -m_new));
          pf[n][sub >> 1][(sub & 1) * 4 + r] = (T)p;
        }
      }
    }
    // ---- O^T += V^T.P^T ------------------------------------------------------------------
    // transposed read: lane i of each 16-lane group addresses row (4g + i/4), columns
    // 16dt + 4*(i%4) .. +4 and receives column 16dt+i of the group's 4 rows
    const int tq = c >> 2, tp = c & 3;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int n = 0; n < NR; ++n) acc_l[n] = TR::mfma16(ones, pf[n][k2], acc_l[n]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int row_lo = (2 * k2) * 16 + 4 * g + tq;
        const int row_hi = row_lo + 16;
        const int ch = 2 * dt + (tp >> 1);
        const int sub8 = (tp & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + swz_off_v<D>(row_lo, ch) + sub8));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + swz_off_v<D>(row_hi, ch) + sub8));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const vec8 vf = __builtin_bit_cast(vec8, both);
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[dt][n] = TR::mfma16(vf, pf[n][k2], acc[dt][n]);
      }
    }
  };

  // ---- main loop: tiles every wave attends in full first (no mask code), then the diagonal /
  //      ragged tiles.  Tile t+1 streams into the other LDS buffer while tile t is computed. -----
  auto run_tiles = [&](int t_begin, int t_end, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int buf = tile & 1;
      if (tile + 1 < n_tiles) stage(tile + 1, buf ^ 1);
      // wave-uniform: skip a tile that is fully masked for all of this wave's rows
      if (!MASKED || tile * PF_BN <= wrow0 + WROWS - 1) compute(tile, buf, masked_tag);
      stage_wait();
      __syncthreads();
    }
  };
  if (n_tiles > 0) {
    const int n_full = min(min(m0, lkv) / PF_BN, n_tiles);
    stage(0, 0);
    // Retire the Q loads here: an empty asm that "rewrites" the fragments makes the compiler wait
    // for them before the loop instead of carrying them as pending into its wait-count
    // bookkeeping of every iteration.
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[n][ks]));
    stage_wait();
    __syncthreads();
    run_tiles(0, n_full, std::false_type{});
    run_tiles(n_full, n_tiles, std::true_type{});
  }

  // ---- epilogue: normalise, stage the wave's WROWS x D tile in LDS, store whole rows ---------
  // The 16 lanes of a group write 8 bytes each to 16 different rows: 8-byte unit u of row r goes
  // to unit u ^ (r & 15) so that they cover the 128-byte bank window once.
  char* o_lds = smem + wave * (WROWS * D * 2);
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const float l = acc_l[n][0];  // every output row of the ones-MFMA holds the column's sum
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      typename TR::vec4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (T)(acc[dt][n][r] * inv);
      *reinterpret_cast<typename TR::vec4*>(o_lds + (n * 16 + c) * (D * 2) + (((4 * dt + g) ^ c) << 3)) = o4;
    }
  }
  __syncthreads();
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
#pragma unroll
  for (int i = 0; i < WROWS * CPR / 64; ++i) {
    const int idx = lane + i * 64;
    const int row = idx / CPR, ch = idx % CPR;
    const int qrow = wrow0 + row;
    const int x = row & 15;
    const u32x2 lo = *reinterpret_cast<const u32x2*>(o_lds + row * (D * 2) + (((2 * ch) ^ x) << 3));
    const u32x2 hi = *reinterpret_cast<const u32x2*>(o_lds + row * (D * 2) + (((2 * ch + 1) ^ x) << 3));
    if (qrow < lq && ch < hchunks) {
      *reinterpret_cast<u32x4*>(out + (int64_t)(q0 + qrow) * out_stride + (int64_t)head * head_size + ch * 8) =
          u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
}

// Generic path: one wave per (query row, head); logits of the row in LDS.
template <typename T>
__global__ __launch_bounds__(64) void prefill_generic_kernel(
    T* __restrict__ out, const T* __restrict__ q, const T* __restrict__ k,
    const T* __restrict__ v, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_kv,
    int num_heads, int num_kv_heads, int head_size, int64_t q_stride, int64_t k_stride,
    int64_t v_stride, int64_t out_stride, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);
  float* logits = qs + ((head_size + 3) & ~3);
  const int seq = blockIdx.z, head = blockIdx.y, i = blockIdx.x;
  const int kvh = head / (num_heads / num_kv_heads);
  const int q0 = cu_q[seq], lq = cu_q[seq + 1] - q0;
  const int kv0 = cu_kv[seq], lkv = cu_kv[seq + 1] - kv0;
  if (i >= lq) return;
  const int lane = threadIdx.x;
  const int n = min(i + 1, lkv);
  const T* qp = q + (int64_t)(q0 + i) * q_stride + (int64_t)head * head_size;
  for (int d = lane; d < head_size; d += 64) qs[d] = to_f32(qp[d]);
  __syncthreads();
  float mx = PF_NEG;
  for (int t = lane; t < n; t += 64) {
    const T* kp = k + (int64_t)(kv0 + t) * k_stride + (int64_t)kvh * head_size;
    float dot = 0.f;
    for (int d = 0; d < head_size; ++d) dot += qs[d] * to_f32(kp[d]);
    dot *= scale;
    logits[t] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int t = lane; t < n; t += 64) {
    const float e = __expf(logits[t] - mx);
    logits[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  T* op = out + (int64_t)(q0 + i) * out_stride + (int64_t)head * head_size;
  for (int d = lane; d < head_size; d += 64) {
    float a = 0.f;
    for (int t = 0; t < n; ++t)
      a += logits[t] * to_f32(v[(int64_t)(kv0 + t) * v_stride + (int64_t)kvh * head_size + d]);
    op[d] = from_f32<T>(a * inv);
  }
}

}  // namespace pe

extern "C" int pe_prefill_causal(void* out, const void* q, const void* k, const void* v,
                                 const int32_t* cu_q, const int32_t* cu_kv, int num_seqs,
                                 int max_q_len, int num_heads, int num_kv_heads, int head_size,
                                 int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                 int64_t out_stride, float scale, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_seqs >= 0 && max_q_len >= 0, PE_ERR_INVALID_ARG, "prefill: negative size");
  if (num_seqs == 0 || max_q_len == 0) return PE_OK;
  PE_REQUIRE(out && q && k && v && cu_q && cu_kv, PE_ERR_INVALID_ARG, "prefill: null pointer");
  PE_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 && head_size > 0,
             PE_ERR_INVALID_ARG, "prefill: bad head layout");
  PE_REQUIRE(num_seqs <= 65535 && num_heads <= 65535, PE_ERR_UNSUPPORTED, "prefill: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool hs_ok = head_size == 32 || head_size == 64 || head_size == 80 || head_size == 96 ||
                     head_size == 112 || head_size == 120 || head_size == 128 || head_size == 192 ||
                     head_size == 256;
  static EnvInt force_generic("PE_PREFILL_FORCE_GENERIC", 0);
  const bool mfma = (dtype == PE_BF16 || dtype == PE_F16) && hs_ok && !force_generic.get() &&
                    scale > 0.f && max_q_len / 64 < 65535 &&
                    k_stride > 0 && v_stride > 0 && k_stride < (1 << 24) && v_stride < (1 << 24) &&
                    q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 &&
                    out_stride % 8 == 0 &&
                    (((uintptr_t)out | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0);
  if (mfma) {
    // Tile shape <NR><WAVES> = rows per wave / waves per workgroup: 24 = 32 rows x 4 waves (128-row
    // workgroups, two per CU), 18 = 16 x 8 (128 rows, four waves per SIMD), 28 = 32 x 8 (256 rows,
    // one workgroup per CU), 14 = 16 x 4.  All give the same bits.  Measured on MI355X (32 query /
    // 8 KV heads x 128 unless noted, us per launch, 24 / 18 / 28): 32 x 256 tokens 53.8 / 49.4 / 69.3;
    // 16 x 512 70.4 / 68.0 / 83.8 (12 heads x 64: 20.9 / 19.5 / 21.4); 8 x 1024 100 / 104 / 117;
    // 8 x 1536 200 / 212 / 220; 8 x 2048 381 / - / 378; 8 x 4096 1243 / - / 1221; 8 x 8192 4238 / - / 4073;
    // 1 x 32768 7572 / - / 7292; 8 x 4096 with 8 query heads on 1 KV head 320 / 351 / 326; one
    // prompt alone, 24 / 14 / 28: 1 x 1024 26.3 / 25.0 / 33.9 (12 heads x 64: 19.7 / 12.6 / 24.9,
    // 8 heads on 1 KV head: 24.2 / 21.2 / 31.7), 2 x 1024 32.2 / 36.6 / 37.2, 1 x 4096
    // 177 / 224 / 167, 1 x 8192 581 / 780 / 568; 18 / 14: 1 x 512 15.6 / 13.2, 2 x 512 16.8 / 15.5.
    // So: a launch with no more 128-row tiles than the chip has CUs takes 64-row workgroups (twice
    // as many of them); other short prompts take the four-waves-per-SIMD shape (their workgroups
    // are mostly start-up and diagonal tiles, which more resident waves hide better); long prompts take the
    // 256-row shape (half the K/V tile traffic per flop) from 4096 tokens on with up to 4 query
    // heads per KV head, or from 2048 with enough tiles to fill the chip several times over;
    // everything else the 128-row default.  PE_PREFILL_SHAPE overrides (tuning runs).
    static EnvInt shape_env("PE_PREFILL_SHAPE", 0);
    int shape = shape_env.get();
    if (shape == 0) {
      const int64_t tiles256 = (int64_t)num_seqs * num_heads * ((max_q_len + 255) / 256);
      const int64_t tiles128 = (int64_t)num_seqs * num_heads * ((max_q_len + 127) / 128);
      if (tiles128 <= 256) shape = 14;
      else if (max_q_len <= 512) shape = 18;
      else if ((max_q_len >= 4096 && num_heads / num_kv_heads <= 4) ||
               (max_q_len >= 2048 && tiles256 >= 2048)) shape = 28;
      else shape = PF_DEFAULT_SHAPE;
    }
    const float sl2 = scale * 1.4426950408889634f;
#define PE_LAUNCH_PF(T, HS, NR, WAVES, WPE)                                                        \
  do {                                                                                             \
    const int bm = 16 * NR * WAVES;                                                                \
    dim3 grid((unsigned)num_heads, (unsigned)num_seqs, (unsigned)((max_q_len + bm - 1) / bm));     \
    const size_t lds = 4 * (size_t)PF_BN * HS * 2;                                                 \
    if (lds > 64 * 1024)                                                                           \
      (void)hipFuncSetAttribute(                                                                   \
          reinterpret_cast<const void*>(&prefill_mfma_kernel<T, HS, NR, WAVES, WPE>),              \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((prefill_mfma_kernel<T, HS, NR, WAVES, WPE>), grid, dim3(WAVES * 64), lds,  \
                       st, (T*)out, (const T*)q, (const T*)k, (const T*)v, cu_q, cu_kv, num_heads, \
                       num_kv_heads, head_size, q_stride, k_stride, v_stride, out_stride, sl2);    \
  } while (0)
#define PE_LAUNCH_PF_SHAPE(T, HS)                                                                  \
  do {                                                                                             \
    switch (shape) {                                                                               \
      case 18: PE_LAUNCH_PF(T, HS, 1, 8, 4); break;                                                \
      case 28: PE_LAUNCH_PF(T, HS, 2, 8, 2); break;                                                \
      case 14: PE_LAUNCH_PF(T, HS, 1, 4, 4); break;                                                \
      default: PE_LAUNCH_PF(T, HS, 2, 4, 2); break;                                                \
    }                                                                                              \
  } while (0)
    // row pitch 256 (head sizes 192, 256): four 64-row tiles of K and V take 128 KiB of LDS, so
    // one workgroup per CU - 8 waves x 16 rows, whose staging writes 16 tile rows per step
    PE_DISPATCH_16(dtype, T, {
      if (head_size <= 64) PE_LAUNCH_PF_SHAPE(T, 64);
      else if (head_size <= 128) PE_LAUNCH_PF_SHAPE(T, 128);
      else PE_LAUNCH_PF(T, 256, 1, 8, 2);
    });
#undef PE_LAUNCH_PF_SHAPE
#undef PE_LAUNCH_PF
    return check_launch("prefill_causal");
  }
  const size_t lds = (size_t)(((head_size + 3) & ~3) + max_q_len) * 4;
  PE_REQUIRE(lds <= 160 * 1024, PE_ERR_UNSUPPORTED,
             "generic prefill path keeps a row of logits in LDS: max_q_len %d too long", max_q_len);
  dim3 grid((unsigned)max_q_len, (unsigned)num_heads, (unsigned)num_seqs);
#define PE_LAUNCH_PFG(T)                                                                          \
  do {                                                                                            \
    if (lds > 64 * 1024)                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_generic_kernel<T>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    hipLaunchKernelGGL(prefill_generic_kernel<T>, grid, dim3(64), lds, st, (T*)out, (const T*)q,   \
                       (const T*)k, (const T*)v, cu_q, cu_kv, num_heads, num_kv_heads, head_size,  \
                       q_stride, k_stride, v_stride, out_stride, scale);                          \
  } while (0)
  PE_DISPATCH_ALL(dtype, T, PE_LAUNCH_PFG(T));
#undef PE_LAUNCH_PFG
  return check_launch("prefill_generic");
}
