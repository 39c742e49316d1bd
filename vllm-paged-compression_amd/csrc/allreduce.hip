// One-shot and two-stage all-reduce (sum) for decode-sized messages over peer-mapped memory on gfx950.
//
// The exchange that follows attention in a tensor-parallel model is the sum of the row-parallel
// o_proj partial outputs ([tokens, hidden], vllm/model_executor/layers/linear.py:1099-1100).  At
// decode it is a few hundred KiB: a ring collective spends its time in per-hop latency, while on
// a fully connected xGMI node every rank can simply read all its peers once.  This file is the
// MI355X counterpart of the reference's one-stage custom all-reduce
// (csrc/custom_all_reduce.cuh, dispatched before NCCL in
// vllm/distributed/parallel_state.py:357-379): every rank owns one buffer that all ranks of the
// node have mapped (HIP IPC), publishes its input there, raises a flag in every peer's buffer,
// waits for the peers' flags and sums the N inputs itself, in rank order, in f32 - so all ranks
// produce the same bits.
//
// Buffer layout (per rank, uncached device memory so that stores and flag writes are visible
// to the peers without cache maintenance):
//     [0, AR_HEADER_BYTES)            flags[AR_MAX_BLOCKS][AR_MAX_RANKS] uint32 + status words
//     [AR_HEADER_BYTES, +2*slot)      two data slots used alternately (call sequence parity):
//                                     a rank can only overwrite a slot two calls later, and it
//                                     cannot start that call before every peer has raised the
//                                     flag of the call in between, i.e. has finished reading
// Workgroup b of every rank handles the same slice of the message and synchronises only with
// workgroup b of the peers (flags[b][src]), so no grid-wide barrier is needed.
//
// Two-stage form (the reference's cross_device_reduce_2stage, custom_all_reduce.cuh:235-283, which
// its dispatch picks from 512 KiB with up to 4 ranks and from 256 KiB with up to 8, :476-483): the
// message is cut into one chunk per rank; after the inputs are published, rank r sums chunk r of
// all ranks (rank order, f32) into a result area behind its input slot, a second flag round
// tells the peers, and every rank collects the N reduced chunks.  A rank then pulls
// 2(N-1)/N of the message over its links instead of N-1 times the message, and the bits are the
// one-shot form's.  The result area is part of the slot, so a message of b bytes needs
// b + ceil(b/N) bytes of it.
// A wait that sees no flag for about two seconds gives up, records it in the header
// (pe_ar_timeouts) and lets the launch finish: a missing peer must not hang the GPU.
#include "pe_common.h"

#include <cstring>

namespace pe {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 256;
constexpr int64_t AR_HEADER_BYTES = 8192;  // flags (2 KiB) + status, second flag round at 4 KiB
constexpr int AR_FLAGS_B = 1024;           // uint32 index of the second flag round
constexpr int AR_STATUS_WORD = AR_MAX_BLOCKS * AR_MAX_RANKS;  // uint32 index of the timeout count
constexpr unsigned long long AR_TIMEOUT_TICKS = 200000000ull;  // s_memrealtime runs at 100 MHz

struct ArPeers {
  char* buf[AR_MAX_RANKS];
};

template <typename T>
__device__ __forceinline__ void ar_accumulate(float* acc, const u32x4& raw) {
  constexpr int X = 16 / sizeof(T);
  union {
    u32x4 v;
    T e[X];
  } u;
  u.v = raw;
#pragma unroll
  for (int i = 0; i < X; ++i) acc[i] += to_f32(u.e[i]);
}

// Raise flag[b][rank] of a flag round (uint32 index `base`) in every rank's buffer, own included,
// and wait for every rank's flag here.  Called by all threads of the workgroup; the stores before
// the call are made visible first.
__device__ __forceinline__ void ar_flag_round(const ArPeers& peers, int base, int b, int rank,
                                              int world, uint32_t seq) {
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < world) {
    uint32_t* theirs =
        reinterpret_cast<uint32_t*>(peers.buf[threadIdx.x]) + base + b * AR_MAX_RANKS + rank;
    __hip_atomic_store(theirs, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* here =
        reinterpret_cast<uint32_t*>(peers.buf[rank]) + base + b * AR_MAX_RANKS + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // flags only grow: a peer that is already one call ahead has raised a larger value
    while ((int32_t)(__hip_atomic_load(here, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > AR_TIMEOUT_TICKS) {
        atomicAdd(reinterpret_cast<uint32_t*>(peers.buf[rank]) + AR_STATUS_WORD, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

// sum of vector i of all ranks' input slots, rank order, f32
template <typename T>
__device__ __forceinline__ u32x4 ar_sum_ranks(const ArPeers& peers, int64_t slot_off, int64_t i,
                                              int world) {
  constexpr int X = 16 / sizeof(T);
  float acc[X];
#pragma unroll
  for (int j = 0; j < X; ++j) acc[j] = 0.f;
  u32x4 part[AR_MAX_RANKS];
#pragma unroll
  for (int r = 0; r < AR_MAX_RANKS; ++r)
    if (r < world) part[r] = reinterpret_cast<const u32x4*>(peers.buf[r] + slot_off)[i];
#pragma unroll
  for (int r = 0; r < AR_MAX_RANKS; ++r)
    if (r < world) ar_accumulate<T>(acc, part[r]);
  union {
    u32x4 v;
    T e[X];
  } o;
#pragma unroll
  for (int j = 0; j < X; ++j) o.e[j] = from_f32<T>(acc[j]);
  return o.v;
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void allreduce_twostage_kernel(
    T* __restrict__ out, const T* __restrict__ in, int64_t n_vec /* 16-byte vectors */,
    int64_t chunk_vec /* vectors per rank chunk */, ArPeers peers, int rank, int world,
    int64_t slot_bytes, uint32_t seq) {
  const int b = blockIdx.x;
  const int64_t per_block = (chunk_vec + gridDim.x - 1) / gridDim.x;
  const int64_t slot_off = AR_HEADER_BYTES + (int64_t)(seq & 1u) * slot_bytes;
  const int64_t res_off = slot_off + n_vec * 16;  // the owning rank's reduced chunk
  // workgroup b's part of chunk c: the same vectors on every rank
  const int64_t sub_lo = min(b * per_block, chunk_vec), sub_hi = min((b + 1) * per_block, chunk_vec);
  auto lo_of = [&](int c) { return min(n_vec, c * chunk_vec + sub_lo); };
  auto hi_of = [&](int c) { return min(n_vec, c * chunk_vec + sub_hi); };

  // 1. publish this workgroup's part of every chunk of the input
  u32x4* mine = reinterpret_cast<u32x4*>(peers.buf[rank] + slot_off);
  const u32x4* src = reinterpret_cast<const u32x4*>(in);
  for (int c = 0; c < world; ++c)
    for (int64_t i = lo_of(c) + threadIdx.x; i < hi_of(c); i += AR_THREADS) mine[i] = src[i];
  ar_flag_round(peers, 0, b, rank, world, seq);
  // 2. reduce this rank's chunk: into the result area for the peers, and straight into out
  u32x4* dst = reinterpret_cast<u32x4*>(out);
  u32x4* res = reinterpret_cast<u32x4*>(peers.buf[rank] + res_off);
  const int64_t c0 = (int64_t)rank * chunk_vec;
  for (int64_t i = lo_of(rank) + threadIdx.x; i < hi_of(rank); i += AR_THREADS) {
    const u32x4 v = ar_sum_ranks<T>(peers, slot_off, i, world);
    res[i - c0] = v;
    dst[i] = v;
  }
  ar_flag_round(peers, AR_FLAGS_B, b, rank, world, seq);
  // 3. collect the other ranks' reduced chunks
  for (int c = 0; c < world; ++c) {
    if (c == rank) continue;
    const u32x4* theirs = reinterpret_cast<const u32x4*>(peers.buf[c] + res_off);
    const int64_t cc = (int64_t)c * chunk_vec;
    for (int64_t i = lo_of(c) + threadIdx.x; i < hi_of(c); i += AR_THREADS) dst[i] = theirs[i - cc];
  }
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(
    T* __restrict__ out, const T* __restrict__ in, int64_t n_vec /* 16-byte vectors */,
    ArPeers peers, int rank, int world, int64_t slot_bytes, uint32_t seq) {
  const int b = blockIdx.x;
  const int64_t per_block = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t lo = b * per_block;
  const int64_t hi = min(n_vec, lo + per_block);
  const int64_t slot_off = AR_HEADER_BYTES + (int64_t)(seq & 1u) * slot_bytes;

  // 1. publish this rank's slice
  u32x4* mine = reinterpret_cast<u32x4*>(peers.buf[rank] + slot_off);
  const u32x4* src = reinterpret_cast<const u32x4*>(in);
  for (int64_t i = lo + threadIdx.x; i < hi; i += AR_THREADS) mine[i] = src[i];
  // 2. raise flag[b][rank] in every rank's buffer (own included), 3. wait for all of them here
  ar_flag_round(peers, 0, b, rank, world, seq);
  // 4. sum the N slices in rank order (identical on every rank), f32 accumulation
  u32x4* dst = reinterpret_cast<u32x4*>(out);
  for (int64_t i = lo + threadIdx.x; i < hi; i += AR_THREADS)
    dst[i] = ar_sum_ranks<T>(peers, slot_off, i, world);
}

}  // namespace pe

extern "C" {

int64_t pe_ar_buffer_bytes(int64_t max_message_bytes) {
  if (max_message_bytes <= 0) return 0;
  const int64_t slot = (max_message_bytes + 255) / 256 * 256;
  return pe::AR_HEADER_BYTES + 2 * slot;
}

int pe_ar_alloc(void** ptr_out, int64_t bytes) {
  using namespace pe;
  PE_REQUIRE(ptr_out && bytes > AR_HEADER_BYTES, PE_ERR_INVALID_ARG, "ar_alloc: bad arguments");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "ar_alloc: %s", hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(PE_ERR_LAUNCH, "ar_alloc: %s", hipGetErrorString(e));
  }
  *ptr_out = p;
  return PE_OK;
}

int pe_ar_free(void* ptr) {
  using namespace pe;
  if (ptr == nullptr) return PE_OK;
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "ar_free: %s", hipGetErrorString(e));
  return PE_OK;
}

int pe_ar_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int pe_ar_get_ipc_handle(void* ptr, void* handle_out) {
  using namespace pe;
  PE_REQUIRE(ptr && handle_out, PE_ERR_INVALID_ARG, "ar_get_ipc_handle: null pointer");
  hipError_t e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), ptr);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
  return PE_OK;
}

int pe_ar_open_ipc_handle(const void* handle, void** ptr_out) {
  using namespace pe;
  PE_REQUIRE(handle && ptr_out, PE_ERR_INVALID_ARG, "ar_open_ipc_handle: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  return PE_OK;
}

int pe_ar_close_ipc_handle(void* ptr) {
  using namespace pe;
  if (ptr == nullptr) return PE_OK;
  hipError_t e = hipIpcCloseMemHandle(ptr);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
  return PE_OK;
}

int pe_ar_timeouts(const void* own_buffer, unsigned int* count_out) {
  using namespace pe;
  PE_REQUIRE(own_buffer && count_out, PE_ERR_INVALID_ARG, "ar_timeouts: null pointer");
  hipError_t e = hipMemcpy(count_out, static_cast<const uint32_t*>(own_buffer) + AR_STATUS_WORD,
                           sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "ar_timeouts: %s", hipGetErrorString(e));
  return PE_OK;
}

int pe_allreduce_oneshot(void* out, const void* in, int64_t count, int dtype,
                         void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                         unsigned int seq, void* stream) {
  using namespace pe;
  PE_REQUIRE(count >= 0, PE_ERR_INVALID_ARG, "allreduce: negative count");
  if (count == 0) return PE_OK;
  PE_REQUIRE(out && in && peer_buffers, PE_ERR_INVALID_ARG, "allreduce: null pointer");
  PE_REQUIRE(world >= 1 && world <= AR_MAX_RANKS && rank >= 0 && rank < world, PE_ERR_INVALID_ARG,
             "allreduce: rank %d / world %d out of range (at most %d ranks)", rank, world,
             AR_MAX_RANKS);
  PE_REQUIRE(seq != 0, PE_ERR_INVALID_ARG, "allreduce: the call sequence number starts at 1");
  const int64_t bytes = count * elem_size(dtype);
  PE_REQUIRE(bytes % 16 == 0 && (((uintptr_t)out | (uintptr_t)in) % 16) == 0, PE_ERR_UNSUPPORTED,
             "allreduce: message and pointers must be multiples of 16 bytes");
  const int64_t slot = (buffer_bytes - AR_HEADER_BYTES) / 2 / 256 * 256;
  PE_REQUIRE(buffer_bytes > AR_HEADER_BYTES && bytes <= slot, PE_ERR_WORKSPACE,
             "allreduce: message of %lld bytes exceeds the %lld-byte slot of the shared buffer",
             (long long)bytes, (long long)slot);
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) peers.buf[r] = nullptr;
  for (int r = 0; r < world; ++r) {
    PE_REQUIRE(peer_buffers[r] != nullptr, PE_ERR_INVALID_ARG, "allreduce: peer buffer %d is null", r);
    peers.buf[r] = static_cast<char*>(peer_buffers[r]);
  }
  const int64_t n_vec = bytes / 16;
  // one workgroup per 16 KiB, at most AR_MAX_BLOCKS: the slice boundaries are the same on all ranks
  int blocks = (int)((n_vec + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(allreduce_oneshot_kernel<T>, dim3((unsigned)blocks),
                                     dim3(AR_THREADS), 0, static_cast<hipStream_t>(stream), (T*)out,
                                     (const T*)in, n_vec, peers, rank, world, slot, (uint32_t)seq));
  return check_launch("allreduce_oneshot");
}

int pe_allreduce_twostage(void* out, const void* in, int64_t count, int dtype,
                          void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                          unsigned int seq, void* stream) {
  using namespace pe;
  PE_REQUIRE(count >= 0, PE_ERR_INVALID_ARG, "allreduce: negative count");
  if (count == 0) return PE_OK;
  PE_REQUIRE(out && in && peer_buffers, PE_ERR_INVALID_ARG, "allreduce: null pointer");
  PE_REQUIRE(world >= 1 && world <= AR_MAX_RANKS && rank >= 0 && rank < world, PE_ERR_INVALID_ARG,
             "allreduce: rank %d / world %d out of range (at most %d ranks)", rank, world,
             AR_MAX_RANKS);
  PE_REQUIRE(seq != 0, PE_ERR_INVALID_ARG, "allreduce: the call sequence number starts at 1");
  const int64_t bytes = count * elem_size(dtype);
  PE_REQUIRE(bytes % 16 == 0 && (((uintptr_t)out | (uintptr_t)in) % 16) == 0, PE_ERR_UNSUPPORTED,
             "allreduce: message and pointers must be multiples of 16 bytes");
  const int64_t n_vec = bytes / 16;
  const int64_t chunk_vec = (n_vec + world - 1) / world;
  const int64_t slot = (buffer_bytes - AR_HEADER_BYTES) / 2 / 256 * 256;
  PE_REQUIRE(buffer_bytes > AR_HEADER_BYTES && bytes + chunk_vec * 16 <= slot, PE_ERR_WORKSPACE,
             "allreduce: a two-stage message of %lld bytes needs %lld bytes of the %lld-byte slot",
             (long long)bytes, (long long)(bytes + chunk_vec * 16), (long long)slot);
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) peers.buf[r] = nullptr;
  for (int r = 0; r < world; ++r) {
    PE_REQUIRE(peer_buffers[r] != nullptr, PE_ERR_INVALID_ARG, "allreduce: peer buffer %d is null", r);
    peers.buf[r] = static_cast<char*>(peer_buffers[r]);
  }
  // one workgroup per 16 KiB of a chunk, at most AR_MAX_BLOCKS
  int blocks = (int)((chunk_vec + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(allreduce_twostage_kernel<T>, dim3((unsigned)blocks),
                                     dim3(AR_THREADS), 0, static_cast<hipStream_t>(stream), (T*)out,
                                     (const T*)in, n_vec, chunk_vec, peers, rank, world, slot,
                                     (uint32_t)seq));
  return check_launch("allreduce_twostage");
}

}  // extern "C"
