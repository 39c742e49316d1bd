// Rotary position embedding fused with the KV write on gfx950:
//   pe_rope_reshape_and_cache   rotate Q and K in place (NeoX or GPT-J pairing) and, in the same
//                               pass, scatter the rotated K and V into the paged cache
// The step that precedes the attention path in the model (rotary_embedding, then
// reshape_and_cache) reads and writes K twice and launches twice; here K is read once, rotated in
// registers, written back for prefill attention and written to its cache slot.  HBM-bound: every
// global access is a 16-byte chunk per lane.
//
// Arithmetic follows the reference kernel (csrc/pos_encoding_kernels.cu:10-35), which works in
// the storage type: x*cos and y*sin are each rounded to it, and so is their sum/difference.
#include "pe_common.h"

namespace pe {

constexpr int RP_TOKENS = 16;   // tokens per workgroup tile
constexpr int RP_THREADS = 256;

template <typename T>
union Chunk {
  u32x4 v;
  T e[16 / sizeof(T)];
};

// Rotates one row (head) in place; returns nothing.  Work item = one 16-byte chunk pair (NeoX:
// chunk c and chunk c + rot/2) or one chunk (GPT-J: pairs are adjacent elements).
// cos / sin of one work item: NeoX - the chunk of 8 (4) pairs starting at pair item * X; GPT-J - the
// X / 2 pairs of the chunk's adjacent elements.
template <typename T, bool NEOX>
struct CosSin {
  static constexpr int X = 16 / sizeof(T);
  T c[X], s[X];
  __device__ __forceinline__ void load(const T* cs, int rot_dim, int item) {
    const int half = rot_dim / 2;
    if (NEOX) {
      Chunk<T> cc, ss;
      cc.v = *reinterpret_cast<const u32x4*>(cs + item * X);
      ss.v = *reinterpret_cast<const u32x4*>(cs + half + item * X);
#pragma unroll
      for (int e = 0; e < X; ++e) {
        c[e] = cc.e[e];
        s[e] = ss.e[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < X / 2; ++e) {
        c[e] = cs[item * (X / 2) + e];
        s[e] = cs[half + item * (X / 2) + e];
      }
    }
  }
};

// Rotates one work item of one row in place: a 16-byte chunk pair (NeoX: chunk `item` and chunk
// `item` + rot/2) or one chunk (GPT-J: pairs are adjacent elements).  ``cache_row`` (K rows with a
// slot): base of the row's chunk 0 in the paged K cache; chunk c lives ``cache_chunk_stride``
// elements further per chunk.
template <typename T, bool NEOX>
__device__ __forceinline__ void rotate_item(T* row, const CosSin<T, NEOX>& cs, int rot_dim, int item,
                                            T* cache_row, int64_t cache_chunk_stride) {
  constexpr int X = 16 / sizeof(T);
  const int half = rot_dim / 2;
  if (NEOX) {
    Chunk<T> x, y;
    x.v = *reinterpret_cast<const u32x4*>(row + item * X);
    y.v = *reinterpret_cast<const u32x4*>(row + half + item * X);
#pragma unroll
    for (int e = 0; e < X; ++e)
      rot_pair<T>(to_f32(x.e[e]), to_f32(y.e[e]), to_f32(cs.c[e]), to_f32(cs.s[e]), x.e[e], y.e[e]);
    *reinterpret_cast<u32x4*>(row + item * X) = x.v;
    *reinterpret_cast<u32x4*>(row + half + item * X) = y.v;
    if (cache_row != nullptr) {
      *reinterpret_cast<u32x4*>(cache_row + item * cache_chunk_stride) = x.v;
      *reinterpret_cast<u32x4*>(cache_row + (half / X + item) * cache_chunk_stride) = y.v;
    }
  } else {
    Chunk<T> x;
    x.v = *reinterpret_cast<const u32x4*>(row + item * X);
#pragma unroll
    for (int e = 0; e < X; e += 2)
      rot_pair<T>(to_f32(x.e[e]), to_f32(x.e[e + 1]), to_f32(cs.c[e / 2]), to_f32(cs.s[e / 2]), x.e[e],
                  x.e[e + 1]);
    *reinterpret_cast<u32x4*>(row + item * X) = x.v;
    if (cache_row != nullptr) *reinterpret_cast<u32x4*>(cache_row + item * cache_chunk_stride) = x.v;
  }
}

// One workgroup = RP_TOKENS consecutive tokens x one KV head (and the query heads that share it).
template <typename T, bool NEOX>
__global__ __launch_bounds__(RP_THREADS) void rope_cache_kernel(
    T* __restrict__ query, T* __restrict__ key, const T* __restrict__ value,
    T* __restrict__ key_cache, T* __restrict__ value_cache, const int64_t* __restrict__ positions,
    const T* __restrict__ cos_sin, const int64_t* __restrict__ slot_mapping, int64_t num_tokens,
    int64_t q_stride, int64_t k_stride, int64_t v_stride, int gqa, int num_kv_heads, int head_size,
    int rot_dim, int block_size, int64_t num_slots, int64_t max_position) {
  constexpr int X = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* vt = reinterpret_cast<T*>(smem);  // [head_size][RP_TOKENS]
  int64_t* s_slot = reinterpret_cast<int64_t*>(smem + (size_t)head_size * RP_TOKENS * sizeof(T));
  int64_t* s_pos = s_slot + RP_TOKENS;
  int32_t* s_blk = reinterpret_cast<int32_t*>(s_pos + RP_TOKENS);  // slot / block_size
  int32_t* s_off = s_blk + RP_TOKENS;                              // slot % block_size

  const int h = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * RP_TOKENS;
  const int tid = threadIdx.x;
  if (tid < RP_TOKENS) {
    const int64_t t = t0 + tid;
    int64_t slot = -1, pos = 0;
    if (t < num_tokens) {
      if (slot_mapping != nullptr) {
        slot = slot_mapping[t];
        if (slot >= num_slots) slot = -1;
      }
      pos = positions[t];
      pos = pos < 0 ? 0 : (pos >= max_position ? max_position - 1 : pos);  // stay inside the table
    }
    s_slot[tid] = slot;
    s_pos[tid] = pos;
    s_blk[tid] = slot >= 0 ? (int32_t)(slot / block_size) : 0;
    s_off[tid] = slot >= 0 ? (int32_t)(slot % block_size) : 0;
  }
  __syncthreads();

  // ---- rotate: rows = (token, K head h) and (token, each query head of h) ------------------
  const int chunks = head_size / X;
  const int items = NEOX ? rot_dim / (2 * X) : rot_dim / X;  // work items per row
  // a thread keeps its (token, item) and walks the token's rows: cos / sin are loaded once
  for (int i = tid; i < RP_TOKENS * items; i += RP_THREADS) {
    const int j = i / items;
    const int item = i - j * items;
    const int64_t t = t0 + j;
    if (t >= num_tokens) continue;
    CosSin<T, NEOX> cs;
    cs.load(cos_sin + s_pos[j] * rot_dim, rot_dim, item);
    // K row: rotated in place (prefill attention reads it) and, from the same registers,
    // written to its cache slot: key_cache[blk][h][c][off][0..X)
    T* cache_row = nullptr;
    if (s_slot[j] >= 0)
      cache_row = key_cache + ((int64_t)s_blk[j] * num_kv_heads + h) * (int64_t)chunks * block_size * X +
                  s_off[j] * X;
    rotate_item<T, NEOX>(key + t * k_stride + (int64_t)h * head_size, cs, rot_dim, item, cache_row,
                         (int64_t)block_size * X);
    T* qrow = query + t * q_stride + (int64_t)h * gqa * head_size;
    for (int r = 0; r < gqa; ++r)
      rotate_item<T, NEOX>(qrow + (int64_t)r * head_size, cs, rot_dim, item, nullptr, 0);
  }
  if (key_cache == nullptr) return;

  // ---- K chunks past rot_dim: copied as they are; V: staged transposed into LDS ---------------
  const int rot_chunks = rot_dim / X;
  for (int i = tid; i < RP_TOKENS * chunks; i += RP_THREADS) {
    const int j = i % RP_TOKENS;
    const int c = i / RP_TOKENS;
    const int64_t slot = s_slot[j];
    if (slot < 0) continue;
    const int64_t t = t0 + j;
    const int64_t blk = s_blk[j];
    const int off = s_off[j];
    if (c >= rot_chunks) {
      T* dst = key_cache + ((blk * num_kv_heads + h) * chunks + c) * (int64_t)block_size * X +
               (int64_t)off * X;
      *reinterpret_cast<u32x4*>(dst) =
          *reinterpret_cast<const u32x4*>(key + t * k_stride + (int64_t)h * head_size + c * X);
    }
    Chunk<T> u;
    u.v = *reinterpret_cast<const u32x4*>(value + t * v_stride + (int64_t)h * head_size + c * X);
#pragma unroll
    for (int e = 0; e < X; ++e) vt[(c * X + e) * RP_TOKENS + j] = u.e[e];
  }
  __syncthreads();
  constexpr int GROUPS = RP_TOKENS / X;
  for (int i = tid; i < head_size * GROUPS; i += RP_THREADS) {
    const int g = i % GROUPS;
    const int d = i / GROUPS;
    const int j0 = g * X;
    const int64_t s0 = s_slot[j0];
    bool run = s0 >= 0 && (s0 % X) == 0 && (s0 % block_size) + X <= block_size;
#pragma unroll
    for (int e = 1; e < X; ++e) run = run && (s_slot[j0 + e] == s0 + e);
    if (run) {
      const int64_t blk = s_blk[j0];
      const int off = s_off[j0];
      T* dst = value_cache + ((blk * num_kv_heads + h) * head_size + d) * (int64_t)block_size + off;
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(vt + d * RP_TOKENS + j0);
    } else {
#pragma unroll
      for (int e = 0; e < X; ++e) {
        if (s_slot[j0 + e] < 0) continue;
        const int64_t blk = s_blk[j0 + e];
        const int off = s_off[j0 + e];
        value_cache[((blk * num_kv_heads + h) * head_size + d) * (int64_t)block_size + off] =
            vt[d * RP_TOKENS + j0 + e];
      }
    }
  }
}

}  // namespace pe

extern "C" int pe_rope_reshape_and_cache(void* query, void* key, const void* value, void* key_cache,
                                         void* value_cache, const int64_t* positions,
                                         const void* cos_sin_cache, const int64_t* slot_mapping,
                                         int64_t num_tokens, int64_t q_stride, int64_t k_stride,
                                         int64_t v_stride, int num_heads, int num_kv_heads,
                                         int head_size, int rot_dim, int is_neox,
                                         int64_t max_position, int block_size, int64_t num_blocks,
                                         int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_tokens >= 0, PE_ERR_INVALID_ARG, "negative num_tokens");
  if (num_tokens == 0) return PE_OK;
  PE_REQUIRE(query && key && positions && cos_sin_cache, PE_ERR_INVALID_ARG,
             "rope_reshape_and_cache: null pointer");
  const bool to_cache = key_cache != nullptr;
  PE_REQUIRE(!to_cache || (value && value_cache && slot_mapping), PE_ERR_INVALID_ARG,
             "rope_reshape_and_cache: the cache write needs value, value_cache and slot_mapping");
  PE_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 && head_size > 0,
             PE_ERR_INVALID_ARG, "rope_reshape_and_cache: bad head layout");
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(head_size % x == 0 && q_stride % x == 0 && k_stride % x == 0 &&
                 (!to_cache || v_stride % x == 0),
             PE_ERR_UNSUPPORTED, "rows must be made of 16-byte chunks");
  PE_REQUIRE(rot_dim > 0 && rot_dim <= head_size && rot_dim % (2 * x) == 0, PE_ERR_UNSUPPORTED,
             "rot_dim %d must be a multiple of %d and at most head_size", rot_dim, 2 * x);
  PE_REQUIRE(max_position > 0, PE_ERR_INVALID_ARG, "max_position must be positive");
  PE_REQUIRE(!to_cache || (block_size > 0 && num_blocks > 0 && block_size % x == 0),
             PE_ERR_INVALID_ARG, "rope_reshape_and_cache: bad cache geometry");
  PE_REQUIRE((((uintptr_t)query | (uintptr_t)key | (uintptr_t)cos_sin_cache |
               (uintptr_t)(to_cache ? value : query) | (uintptr_t)(to_cache ? key_cache : query) |
               (uintptr_t)(to_cache ? value_cache : query)) % 16) == 0,
             PE_ERR_UNSUPPORTED, "rope_reshape_and_cache: pointers must be 16-byte aligned");
  const int64_t tiles = (num_tokens + RP_TOKENS - 1) / RP_TOKENS;
  PE_REQUIRE(tiles <= 0x7fffffff && num_kv_heads <= 65535, PE_ERR_UNSUPPORTED, "grid too large");
  dim3 grid((unsigned)tiles, (unsigned)num_kv_heads);
  const size_t lds = (size_t)head_size * RP_TOKENS * elem_size(dtype) + 3 * RP_TOKENS * sizeof(int64_t);
  const int gqa = num_heads / num_kv_heads;
  const int64_t num_slots = to_cache ? num_blocks * (int64_t)block_size : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define PE_LAUNCH_ROPE(T, NEOX)                                                                    \
  hipLaunchKernelGGL((rope_cache_kernel<T, NEOX>), grid, dim3(RP_THREADS), lds, st, (T*)query,      \
                     (T*)key, (const T*)value, (T*)key_cache, (T*)value_cache, positions,          \
                     (const T*)cos_sin_cache, slot_mapping, num_tokens, q_stride, k_stride,        \
                     v_stride, gqa, num_kv_heads, head_size, rot_dim, block_size, num_slots,       \
                     max_position)
  PE_DISPATCH_ALL(dtype, T, {
    if (is_neox) PE_LAUNCH_ROPE(T, true);
    else PE_LAUNCH_ROPE(T, false);
  });
#undef PE_LAUNCH_ROPE
  return check_launch("rope_reshape_and_cache");
}
