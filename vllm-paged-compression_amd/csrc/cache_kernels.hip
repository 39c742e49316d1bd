// KV-cache data movement for the paged-eviction path on gfx950:
//   pe_reshape_and_cache      KV write (+ fused prompt-prune gather, + slot zeroing)
//   pe_prompt_prune_gather    dense gather of the kept prompt rows
//   pe_prompt_keep_index_streaming
//   pe_copy_blocks / pe_swap_blocks
// All of it is HBM-bound byte movement: the kernels are organised so that global loads
// and stores are 16 bytes per lane and contiguous across a wave wherever the paged
// layout allows it.
#include <atomic>

#include "pe_common.h"

namespace pe {

thread_local char g_last_error[512] = "";
char* last_error_buf() { return g_last_error; }

static std::atomic<int> g_env_generation{0};
int env_generation() { return g_env_generation.load(std::memory_order_relaxed); }

constexpr int RC_TOKENS = 16;   // tokens per workgroup tile
constexpr int RC_THREADS = 256;

// One workgroup = RC_TOKENS consecutive tokens x one KV head.
//  K: 16-byte chunk c of token j goes to key_cache[blk][h][c][off][0..x): lanes that
//     differ in j write consecutive 16-byte pieces when the slots are consecutive.
//  V: value_cache[blk][h][d][off] has the token innermost, so the tile is transposed
//     through LDS and written as 16-byte runs of 8 (x) tokens whenever those tokens
//     occupy consecutive, aligned slots of one block; otherwise element by element.
template <typename T>
__global__ __launch_bounds__(RC_THREADS) void reshape_and_cache_kernel(
    const T* __restrict__ key, const T* __restrict__ value, T* __restrict__ key_cache,
    T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    const int32_t* __restrict__ src_index, int src_head_stride, int64_t num_tokens,
    int64_t key_stride, int64_t value_stride, int num_heads, int head_size, int block_size,
    int64_t num_slots) {
  constexpr int X = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* vt = reinterpret_cast<T*>(smem);                       // [head_size][RC_TOKENS]
  int64_t* s_slot = reinterpret_cast<int64_t*>(smem + (size_t)head_size * RC_TOKENS * sizeof(T));
  int64_t* s_src = s_slot + RC_TOKENS;   // element offset of the source V row of (token, head)
  int64_t* s_srck = s_src + RC_TOKENS;   // same for K

  const int h = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * RC_TOKENS;
  const int tid = threadIdx.x;
  if (tid < RC_TOKENS) {
    const int64_t t = t0 + tid;
    int64_t slot = -1, src = 0;
    if (t < num_tokens) {
      slot = slot_mapping[t];
      if (slot >= num_slots) slot = -1;
      int64_t src_h = h;
      if (src_index == nullptr) {
        src = t;
      } else if (src_head_stride == 0) {
        src = src_index[t];
      } else if (src_head_stride == 1) {
        src = src_index[t * num_heads + h];
      } else {  // (row, head) pair
        const int64_t pr = src_index[t * num_heads + h];
        src = pr / num_heads;
        src_h = pr % num_heads;
      }
      s_srck[tid] = src * key_stride + src_h * head_size;
      src = src * value_stride + src_h * head_size;
    }
    s_slot[tid] = slot;
    s_src[tid] = src;
  }
  __syncthreads();

  const int chunks = head_size / X;  // 16-byte chunks per (token, head) row
  // ---- K: straight 16-byte copies; V: staged transposed into LDS -------------------
  // token fastest across lanes: consecutive slots give contiguous K stores, and the
  // 2-byte LDS writes of neighbouring lanes fall into neighbouring banks.
  for (int i = tid; i < RC_TOKENS * chunks; i += RC_THREADS) {
    const int j = i % RC_TOKENS;
    const int c = i / RC_TOKENS;
    const int64_t slot = s_slot[j];
    if (slot < 0) continue;
    const int64_t blk = slot / block_size;
    const int off = (int)(slot % block_size);
    const u32x4 kv = *reinterpret_cast<const u32x4*>(key + s_srck[j] + c * X);
    union {
      u32x4 v;
      T e[X];
    } u;
    u.v = *reinterpret_cast<const u32x4*>(value + s_src[j] + c * X);
    T* dst = key_cache + ((blk * num_heads + h) * chunks + c) * (int64_t)block_size * X +
             (int64_t)off * X;
    *reinterpret_cast<u32x4*>(dst) = kv;
#pragma unroll
    for (int e = 0; e < X; ++e) vt[(c * X + e) * RC_TOKENS + j] = u.e[e];
  }
  __syncthreads();
  // ---- V stores: one 16-byte run of X tokens per (d, group) -----------------------
  constexpr int GROUPS = RC_TOKENS / X;
  for (int i = tid; i < head_size * GROUPS; i += RC_THREADS) {
    const int g = i % GROUPS;
    const int d = i / GROUPS;
    const int j0 = g * X;
    const int64_t s0 = s_slot[j0];
    bool run = s0 >= 0 && (s0 % X) == 0 && (s0 % block_size) + X <= block_size;
#pragma unroll
    for (int e = 1; e < X; ++e) run = run && (s_slot[j0 + e] == s0 + e);
    if (run) {
      const int64_t blk = s0 / block_size;
      const int off = (int)(s0 % block_size);
      T* dst = value_cache + ((blk * num_heads + h) * head_size + d) * (int64_t)block_size + off;
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(vt + d * RC_TOKENS + j0);
    } else {
#pragma unroll
      for (int e = 0; e < X; ++e) {
        const int64_t s = s_slot[j0 + e];
        if (s < 0) continue;
        const int64_t blk = s / block_size;
        const int off = (int)(s % block_size);
        value_cache[((blk * num_heads + h) * head_size + d) * (int64_t)block_size + off] =
            vt[d * RC_TOKENS + j0 + e];
      }
    }
  }
}

// zero the K and V rows of the given slots (one workgroup per slot)
template <typename T>
__global__ void zero_slots_kernel(T* __restrict__ key_cache, T* __restrict__ value_cache,
                                  const int64_t* __restrict__ zero_slots, int num_heads,
                                  int head_size, int block_size, int64_t num_slots) {
  constexpr int X = 16 / sizeof(T);
  const int64_t slot = zero_slots[blockIdx.x];
  if (slot < 0 || slot >= num_slots) return;
  const int64_t blk = slot / block_size;
  const int off = (int)(slot % block_size);
  const int chunks = head_size / X;
  for (int i = threadIdx.x; i < num_heads * head_size; i += blockDim.x) {
    const int h = i / head_size, d = i % head_size;
    key_cache[((blk * num_heads + h) * chunks + d / X) * (int64_t)block_size * X +
              (int64_t)off * X + d % X] = (T)0.f;
    value_cache[((blk * num_heads + h) * head_size + d) * (int64_t)block_size + off] = (T)0.f;
  }
}

// the same for the caches of every layer in one launch: grid (slot, layer)
template <typename T>
__global__ void zero_slots_layers_kernel(void* const* __restrict__ key_ptrs,
                                         void* const* __restrict__ value_ptrs,
                                         const int64_t* __restrict__ zero_slots, int num_heads,
                                         int head_size, int block_size, int64_t num_slots) {
  constexpr int X = 16 / sizeof(T);
  const int64_t slot = zero_slots[blockIdx.x];
  if (slot < 0 || slot >= num_slots) return;
  T* key_cache = static_cast<T*>(key_ptrs[blockIdx.y]);
  T* value_cache = static_cast<T*>(value_ptrs[blockIdx.y]);
  const int64_t blk = slot / block_size;
  const int off = (int)(slot % block_size);
  const int chunks = head_size / X;
  for (int i = threadIdx.x; i < num_heads * head_size; i += blockDim.x) {
    const int h = i / head_size, d = i % head_size;
    key_cache[((blk * num_heads + h) * chunks + d / X) * (int64_t)block_size * X +
              (int64_t)off * X + d % X] = (T)0.f;
    value_cache[((blk * num_heads + h) * head_size + d) * (int64_t)block_size + off] = (T)0.f;
  }
}

// out[i][h][:] = in[src(i,h)][h][:], 16 bytes per thread
template <typename T>
__global__ void prune_gather_kernel(const T* __restrict__ key, const T* __restrict__ value,
                                    T* __restrict__ out_key, T* __restrict__ out_value,
                                    const int32_t* __restrict__ src_index, int src_head_stride,
                                    int64_t n_out, int64_t key_stride, int64_t value_stride,
                                    int num_heads, int head_size) {
  constexpr int X = 16 / sizeof(T);
  const int chunks_per_row = num_heads * head_size / X;
  const int64_t total = n_out * chunks_per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / chunks_per_row;
    const int c = (int)(i % chunks_per_row);
    const int h = c * X / head_size;
    int64_t src, sc = c;
    if (src_head_stride == 0) {
      src = src_index[row];
    } else if (src_head_stride == 1) {
      src = src_index[row * num_heads + h];
    } else {  // (row, head) pair: read the chunk at the same offset inside the source head
      const int64_t pr = src_index[row * num_heads + h];
      src = pr / num_heads;
      sc = (pr % num_heads) * (head_size / X) + (c - h * (head_size / X));
    }
    *reinterpret_cast<u32x4*>(out_key + row * (int64_t)num_heads * head_size + c * X) =
        *reinterpret_cast<const u32x4*>(key + src * key_stride + sc * X);
    *reinterpret_cast<u32x4*>(out_value + row * (int64_t)num_heads * head_size + c * X) =
        *reinterpret_cast<const u32x4*>(value + src * value_stride + sc * X);
  }
}

// kept-row index of the streamingLLM prompt prune; grid.y = request
__global__ void keep_index_streaming_kernel(int32_t* __restrict__ src_index,
                                            const int32_t* __restrict__ cu_q,
                                            const int32_t* __restrict__ cu_kept, int budget,
                                            int bs) {
  const int r = blockIdx.y;
  const int q0 = cu_q[r], L = cu_q[r + 1] - q0;
  const int k0 = cu_kept[r], n_kept = cu_kept[r + 1] - k0;
  int n_prune = 0;
  if (budget == 0) {
    // pruned count given by the caller's kept lengths (percentage prune type: the oldest
    // middle tokens go, kvcache_prunner_old.py:166-176)
    n_prune = L > n_kept ? L - n_kept : 0;
  } else if (L > budget) {
    const int rem = L % bs;
    const int end_mid = L - bs - rem;
    n_prune = (end_mid - bs) - (budget - 2 * bs);
    if (n_prune < 0) n_prune = 0;
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_kept; n += gridDim.x * blockDim.x) {
    // kept = [0, bs) U [bs + n_prune, end_mid) U [end_mid, L): a single jump after bs
    const int src = n >= bs ? n + n_prune : n;
    src_index[k0 + n] = q0 + src;
  }
}

template <int VEC>
__global__ void copy_blocks_kernel(void* const* __restrict__ key_ptrs,
                                   void* const* __restrict__ value_ptrs,
                                   const int64_t* __restrict__ mapping, int64_t block_bytes) {
  const int layer = blockIdx.y, pair = blockIdx.x;
  const int64_t src = mapping[2 * pair], dst = mapping[2 * pair + 1];
  char* kc = static_cast<char*>(key_ptrs[layer]);
  char* vc = static_cast<char*>(value_ptrs[layer]);
  if (VEC == 16) {
    const int64_t n = block_bytes / 16;
    const u32x4* ks = reinterpret_cast<const u32x4*>(kc + src * block_bytes);
    u32x4* kd = reinterpret_cast<u32x4*>(kc + dst * block_bytes);
    const u32x4* vs = reinterpret_cast<const u32x4*>(vc + src * block_bytes);
    u32x4* vd = reinterpret_cast<u32x4*>(vc + dst * block_bytes);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      kd[i] = ks[i];
      vd[i] = vs[i];
    }
  } else {
    for (int64_t i = threadIdx.x; i < block_bytes; i += blockDim.x) {
      kc[dst * block_bytes + i] = kc[src * block_bytes + i];
      vc[dst * block_bytes + i] = vc[src * block_bytes + i];
    }
  }
}

}  // namespace pe

extern "C" {

const char* pe_last_error(void) { return pe::g_last_error; }
int pe_version(void) { return 105; }  // 105: pe_reload_env, all-reduce status word + self-check
void pe_reload_env(void) { pe::g_env_generation.fetch_add(1, std::memory_order_relaxed); }

int pe_device_diagnostics(unsigned int* flags_out, int reset) {
  PE_REQUIRE(flags_out != nullptr, PE_ERR_INVALID_ARG, "device_diagnostics: null pointer");
  unsigned int a = 0, b = 0, c = 0;
  if (int rc = pe::decode_diag_fetch(&a, reset)) return rc;
  if (int rc = pe::evict_diag_fetch(&b, reset)) return rc;
  if (int rc = pe::state_diag_fetch(&c, reset)) return rc;
  *flags_out = a | b | c;
  return PE_OK;
}

int pe_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                         const int64_t* slot_mapping, int64_t num_tokens, int64_t key_stride,
                         int64_t value_stride, const int32_t* src_index,
                         int src_index_head_stride, const int64_t* zero_slots,
                         int64_t num_zero_slots, int num_kv_heads, int head_size, int block_size,
                         int64_t num_blocks, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_tokens >= 0 && num_zero_slots >= 0, PE_ERR_INVALID_ARG, "negative count");
  PE_REQUIRE(key_cache && value_cache, PE_ERR_INVALID_ARG, "reshape_and_cache: null cache");
  PE_REQUIRE(num_tokens == 0 || (key && value && slot_mapping), PE_ERR_INVALID_ARG,
             "reshape_and_cache: null key/value/slot_mapping");
  PE_REQUIRE(num_zero_slots == 0 || zero_slots, PE_ERR_INVALID_ARG, "null zero_slots");
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && block_size > 0 && num_blocks > 0,
             PE_ERR_INVALID_ARG, "reshape_and_cache: bad sizes");
  const int es = elem_size(dtype);
  const int x = 16 / es;
  PE_REQUIRE(head_size % x == 0, PE_ERR_UNSUPPORTED,
             "head_size %d is not a multiple of %d (16-byte chunks)", head_size, x);
  PE_REQUIRE(key_stride % x == 0 && value_stride % x == 0, PE_ERR_UNSUPPORTED,
             "key/value token stride must be a multiple of %d elements", x);
  PE_REQUIRE(((uintptr_t)key | (uintptr_t)value | (uintptr_t)key_cache | (uintptr_t)value_cache) % 16 == 0,
             PE_ERR_INVALID_ARG, "pointers must be 16-byte aligned");
  PE_REQUIRE(src_index_head_stride >= 0 && src_index_head_stride <= 2, PE_ERR_INVALID_ARG,
             "src_index_head_stride must be 0, 1 or 2");
  PE_REQUIRE(num_kv_heads <= 65535, PE_ERR_UNSUPPORTED, "too many heads");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t num_slots = num_blocks * block_size;
  if (num_tokens > 0) {
    const int64_t tiles = (num_tokens + RC_TOKENS - 1) / RC_TOKENS;
    PE_REQUIRE(tiles <= 0x7fffffff, PE_ERR_UNSUPPORTED, "too many tokens");
    dim3 grid((unsigned)tiles, (unsigned)num_kv_heads);
    const size_t lds = (size_t)head_size * RC_TOKENS * es + 3 * RC_TOKENS * sizeof(int64_t);
    PE_DISPATCH_ALL(dtype, T,
                    hipLaunchKernelGGL(reshape_and_cache_kernel<T>, grid, dim3(RC_THREADS), lds, st,
                                       (const T*)key, (const T*)value, (T*)key_cache,
                                       (T*)value_cache, slot_mapping, src_index,
                                       src_index_head_stride, num_tokens, key_stride, value_stride,
                                       num_kv_heads, head_size, block_size, num_slots));
    if (int rc = check_launch("reshape_and_cache")) return rc;
  }
  if (num_zero_slots > 0) {
    PE_DISPATCH_ALL(dtype, T,
                    hipLaunchKernelGGL(zero_slots_kernel<T>, dim3((unsigned)num_zero_slots),
                                       dim3(256), 0, st, (T*)key_cache, (T*)value_cache, zero_slots,
                                       num_kv_heads, head_size, block_size, num_slots));
    if (int rc = check_launch("zero_slots")) return rc;
  }
  return PE_OK;
}

int pe_zero_slots_layers(void* const* key_cache_ptrs, void* const* value_cache_ptrs, int num_layers,
                         const int64_t* zero_slots, int64_t num_zero_slots, int num_kv_heads,
                         int head_size, int block_size, int64_t num_blocks, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_layers >= 0 && num_zero_slots >= 0, PE_ERR_INVALID_ARG, "zero_slots_layers: negative count");
  if (num_layers == 0 || num_zero_slots == 0) return PE_OK;
  PE_REQUIRE(key_cache_ptrs && value_cache_ptrs && zero_slots, PE_ERR_INVALID_ARG,
             "zero_slots_layers: null pointer");
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0 && block_size > 0 && num_blocks > 0 &&
                 head_size % (16 / elem_size(dtype)) == 0,
             PE_ERR_INVALID_ARG, "zero_slots_layers: bad sizes");
  PE_REQUIRE(num_layers <= 65535, PE_ERR_UNSUPPORTED, "zero_slots_layers: grid too large");
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(zero_slots_layers_kernel<T>,
                                     dim3((unsigned)num_zero_slots, (unsigned)num_layers), dim3(256), 0,
                                     static_cast<hipStream_t>(stream), key_cache_ptrs, value_cache_ptrs,
                                     zero_slots, num_kv_heads, head_size, block_size,
                                     num_blocks * block_size));
  return check_launch("zero_slots_layers");
}

int pe_prompt_prune_gather(const void* key, const void* value, void* out_key, void* out_value,
                           const int32_t* src_index, int src_index_head_stride, int64_t n_out,
                           int64_t key_stride, int64_t value_stride, int num_kv_heads,
                           int head_size, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(n_out >= 0, PE_ERR_INVALID_ARG, "negative n_out");
  if (n_out == 0) return PE_OK;
  PE_REQUIRE(key && value && out_key && out_value && src_index, PE_ERR_INVALID_ARG,
             "prompt_prune_gather: null pointer");
  PE_REQUIRE(num_kv_heads > 0 && head_size > 0, PE_ERR_INVALID_ARG, "prompt_prune_gather: bad sizes");
  PE_REQUIRE(((uintptr_t)key | (uintptr_t)value | (uintptr_t)out_key | (uintptr_t)out_value) % 16 == 0,
             PE_ERR_INVALID_ARG, "prompt_prune_gather: pointers must be 16-byte aligned");
  const int x = 16 / elem_size(dtype);
  PE_REQUIRE(head_size % x == 0 && key_stride % x == 0 && value_stride % x == 0,
             PE_ERR_UNSUPPORTED, "rows must be made of 16-byte chunks");
  PE_REQUIRE(src_index_head_stride >= 0 && src_index_head_stride <= 2, PE_ERR_INVALID_ARG,
             "src_index_head_stride must be 0, 1 or 2");
  const int64_t total = n_out * (num_kv_heads * head_size / x);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(prune_gather_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                     static_cast<hipStream_t>(stream), (const T*)key,
                                     (const T*)value, (T*)out_key, (T*)out_value, src_index,
                                     src_index_head_stride, n_out, key_stride, value_stride,
                                     num_kv_heads, head_size));
  return check_launch("prompt_prune_gather");
}

int pe_prompt_keep_index_streaming(int32_t* src_index, const int32_t* cu_q, const int32_t* cu_kept,
                                   int num_seqs, int cache_budget, int block_size, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_seqs >= 0, PE_ERR_INVALID_ARG, "negative num_seqs");
  if (num_seqs == 0) return PE_OK;
  PE_REQUIRE(src_index && cu_q && cu_kept, PE_ERR_INVALID_ARG, "keep_index: null pointer");
  PE_REQUIRE(block_size > 0 && (cache_budget == 0 || cache_budget >= 3 * block_size),
             PE_ERR_INVALID_ARG,
             "cache budget must be at least 3 blocks (kvcache_prunner.py:15), or 0 when the kept "
             "lengths alone define the prune");
  PE_REQUIRE(num_seqs <= 65535, PE_ERR_UNSUPPORTED, "too many prompts in one call");
  hipLaunchKernelGGL(keep_index_streaming_kernel, dim3(32, (unsigned)num_seqs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src_index, cu_q, cu_kept, cache_budget,
                     block_size);
  return check_launch("keep_index_streaming");
}

int pe_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                   const int64_t* block_mapping, int num_layers, int num_pairs,
                   int64_t numel_per_block, int dtype, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_layers >= 0 && num_pairs >= 0 && numel_per_block > 0, PE_ERR_INVALID_ARG,
             "copy_blocks: bad sizes");
  if (num_layers == 0 || num_pairs == 0) return PE_OK;
  PE_REQUIRE(key_cache_ptrs && value_cache_ptrs && block_mapping, PE_ERR_INVALID_ARG,
             "copy_blocks: null pointer");
  PE_REQUIRE(num_layers <= 65535, PE_ERR_UNSUPPORTED, "too many layers");
  const int64_t bytes = numel_per_block * elem_size(dtype);
  dim3 grid((unsigned)num_pairs, (unsigned)num_layers);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bytes % 16 == 0)
    hipLaunchKernelGGL(copy_blocks_kernel<16>, grid, dim3(256), 0, st, key_cache_ptrs,
                       value_cache_ptrs, block_mapping, bytes);
  else
    hipLaunchKernelGGL(copy_blocks_kernel<1>, grid, dim3(256), 0, st, key_cache_ptrs,
                       value_cache_ptrs, block_mapping, bytes);
  return check_launch("copy_blocks");
}

int pe_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host,
                   int64_t num_pairs, int64_t block_bytes, int kind, void* stream) {
  using namespace pe;
  PE_REQUIRE(num_pairs >= 0 && block_bytes > 0, PE_ERR_INVALID_ARG, "swap_blocks: bad sizes");
  PE_REQUIRE(kind >= 0 && kind <= 2, PE_ERR_INVALID_ARG, "swap_blocks: invalid device combination");
  if (num_pairs == 0) return PE_OK;
  PE_REQUIRE(src && dst && block_mapping_host, PE_ERR_INVALID_ARG, "swap_blocks: null pointer");
  const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice
                                    : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
  for (int64_t i = 0; i < num_pairs; ++i) {
    const int64_t s = block_mapping_host[2 * i], d = block_mapping_host[2 * i + 1];
    hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + d * block_bytes,
                                  static_cast<const char*>(src) + s * block_bytes, block_bytes, k,
                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "swap_blocks: %s", hipGetErrorString(e));
  }
  return PE_OK;
}

}  // extern "C"
