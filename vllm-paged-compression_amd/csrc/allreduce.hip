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
// A missing peer must neither hang the GPU nor pass for a result: a wait that sees no flag within
// the caller's bound (default 10 s) gives up, fills this workgroup's part of the output with NaN
// instead of a partial sum, counts the event in the buffer header (pe_ar_timeouts) and raises the
// communicator's status word in pinned host memory (pe_ar_status_alloc), which the host reads
// before every call without touching the device.
#include "pe_common.h"

#include <cstring>

namespace pe {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 256;
constexpr int64_t AR_HEADER_BYTES = 8192;  // flags (2 KiB) + status, second flag round at 4 KiB
constexpr int AR_FLAGS_B = 1024;           // uint32 index of the second flag round
constexpr int AR_STATUS_WORD = AR_MAX_BLOCKS * AR_MAX_RANKS;  // uint32 index of the timeout count
constexpr int AR_CHECK_WORD = 1536;        // uint32 index of the mapping self-check area (512 words)
constexpr int AR_CHECK_OWN = 256;          // words a rank fills in its own buffer; peers' words follow
constexpr int AR_CHECK_PEER = 8;           // words every rank writes into every buffer
constexpr int64_t AR_DEFAULT_TIMEOUT_US = 10000000;  // s_memrealtime runs at 100 MHz: 100 ticks / us

// status words in pinned host memory (uint32 each): [0] != 0: a wait gave up, [1] seq of that
// call, [2] 1 + rank of a peer that was missing
struct ArStatus {
  uint32_t* host;                 // device-visible address of the host words, or nullptr
  unsigned long long timeout_ticks;
};

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
// the call are made visible first.  Returns false (to every thread) when a peer's flag did not
// arrive in time.
__device__ __forceinline__ bool ar_flag_round(const ArPeers& peers, int base, int b, int rank,
                                              int world, uint32_t seq, const ArStatus& status) {
  __threadfence_system();
  __syncthreads();
  int gave_up = 0;
  if ((int)threadIdx.x < world) {
    uint32_t* theirs =
        reinterpret_cast<uint32_t*>(peers.buf[threadIdx.x]) + base + b * AR_MAX_RANKS + rank;
    __hip_atomic_store(theirs, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* here =
        reinterpret_cast<uint32_t*>(peers.buf[rank]) + base + b * AR_MAX_RANKS + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // flags only grow: a peer that is already one call ahead has raised a larger value
    while ((int32_t)(__hip_atomic_load(here, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > status.timeout_ticks) {
        gave_up = 1;
        atomicAdd(reinterpret_cast<uint32_t*>(peers.buf[rank]) + AR_STATUS_WORD, 1u);
        if (status.host != nullptr) {
          __hip_atomic_store(status.host + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(status.host + 2, (uint32_t)threadIdx.x + 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(status.host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  return __syncthreads_or(gave_up) == 0;
}

// 16 bytes of quiet NaNs of the element type: what a workgroup stores instead of a sum it could
// not complete
template <typename T>
__device__ __forceinline__ u32x4 ar_poison() {
  const uint32_t w = sizeof(T) == 4 ? 0x7fc00000u
                                    : (Traits<T>::code == PE_BF16 ? 0x7fc07fc0u : 0x7e007e00u);
  return u32x4{w, w, w, w};
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
    T* out, const T* in /* may be out */, int64_t n_vec /* 16-byte vectors */,
    int64_t chunk_vec /* vectors per rank chunk */, ArPeers peers, int rank, int world,
    int64_t slot_bytes, uint32_t seq, ArStatus status) {
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
  const bool ok1 = ar_flag_round(peers, 0, b, rank, world, seq, status);
  // 2. reduce this rank's chunk: into the result area for the peers, and straight into out
  //    (a round that gave up publishes NaN, so the peers that collect this chunk see it too)
  u32x4* dst = reinterpret_cast<u32x4*>(out);
  u32x4* res = reinterpret_cast<u32x4*>(peers.buf[rank] + res_off);
  const int64_t c0 = (int64_t)rank * chunk_vec;
  for (int64_t i = lo_of(rank) + threadIdx.x; i < hi_of(rank); i += AR_THREADS) {
    const u32x4 v = ok1 ? ar_sum_ranks<T>(peers, slot_off, i, world) : ar_poison<T>();
    res[i - c0] = v;
    dst[i] = v;
  }
  const bool ok2 = ar_flag_round(peers, AR_FLAGS_B, b, rank, world, seq, status);
  // 3. collect the other ranks' reduced chunks
  for (int c = 0; c < world; ++c) {
    if (c == rank) continue;
    const u32x4* theirs = reinterpret_cast<const u32x4*>(peers.buf[c] + res_off);
    const int64_t cc = (int64_t)c * chunk_vec;
    for (int64_t i = lo_of(c) + threadIdx.x; i < hi_of(c); i += AR_THREADS)
      dst[i] = ok2 ? theirs[i - cc] : ar_poison<T>();
  }
}

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(
    T* out, const T* in /* may be out */, int64_t n_vec /* 16-byte vectors */, ArPeers peers,
    int rank, int world, int64_t slot_bytes, uint32_t seq, ArStatus status) {
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
  const bool ok = ar_flag_round(peers, 0, b, rank, world, seq, status);
  // 4. sum the N slices in rank order (identical on every rank), f32 accumulation
  u32x4* dst = reinterpret_cast<u32x4*>(out);
  for (int64_t i = lo + threadIdx.x; i < hi; i += AR_THREADS)
    dst[i] = ok ? ar_sum_ranks<T>(peers, slot_off, i, world) : ar_poison<T>();
}

// Mapping self-check, run once when a communicator is set up.  phase 0: every rank fills the own
// part of its check area with a rank-tagged pattern and writes AR_CHECK_PEER tagged words into
// every rank's buffer THROUGH THE MAPPING (what the flag writes of the exchange do).  The host
// then synchronises all ranks.  phase 1: every rank reads every peer's own pattern through the
// mapping (what the slot reads do) and, in its own buffer, the words each peer wrote; mismatches
// are counted into *bad_out.
__device__ __forceinline__ uint32_t ar_check_pattern(int writer, int target, int j) {
  return 0xa5000000u ^ ((uint32_t)writer << 20) ^ ((uint32_t)target << 16) ^ (uint32_t)(j * 2654435761u >> 16);
}
__global__ __launch_bounds__(AR_THREADS) void allreduce_selfcheck_kernel(ArPeers peers, int rank, int world,
                                                                         int phase, uint32_t* bad_out) {
  const int t = threadIdx.x;
  if (phase == 0) {
    uint32_t* own = reinterpret_cast<uint32_t*>(peers.buf[rank]) + AR_CHECK_WORD;
    if (t < AR_CHECK_OWN)
      __hip_atomic_store(own + t, ar_check_pattern(rank, rank, t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t < world * AR_CHECK_PEER) {
      const int r = t / AR_CHECK_PEER, k = t % AR_CHECK_PEER;
      uint32_t* theirs = reinterpret_cast<uint32_t*>(peers.buf[r]) + AR_CHECK_WORD + AR_CHECK_OWN +
                         rank * AR_CHECK_PEER + k;
      __hip_atomic_store(theirs, ar_check_pattern(rank, r, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    return;
  }
  int bad = 0;
  for (int r = 0; r < world; ++r) {
    const uint32_t* theirs = reinterpret_cast<const uint32_t*>(peers.buf[r]) + AR_CHECK_WORD;
    if (t < AR_CHECK_OWN &&
        __hip_atomic_load(theirs + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ar_check_pattern(r, r, t))
      bad = 1;
  }
  if (t < world * AR_CHECK_PEER) {
    const int r = t / AR_CHECK_PEER, k = t % AR_CHECK_PEER;
    const uint32_t* mine = reinterpret_cast<const uint32_t*>(peers.buf[rank]) + AR_CHECK_WORD +
                           AR_CHECK_OWN + r * AR_CHECK_PEER + k;
    if (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ar_check_pattern(r, rank, k))
      bad = 1;
  }
  if (bad) atomicAdd(bad_out, 1u);
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

int pe_ar_status_alloc(void** host_words_out) {
  using namespace pe;
  PE_REQUIRE(host_words_out, PE_ERR_INVALID_ARG, "ar_status_alloc: null pointer");
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, 64, hipHostMallocMapped);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "ar_status_alloc: %s", hipGetErrorString(e));
  memset(p, 0, 64);
  *host_words_out = p;
  return PE_OK;
}

int pe_ar_status_free(void* host_words) {
  using namespace pe;
  if (host_words == nullptr) return PE_OK;
  hipError_t e = hipHostFree(host_words);
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "ar_status_free: %s", hipGetErrorString(e));
  return PE_OK;
}

namespace pe {
// the kernels' view of the caller's status words and wait bound
static int ar_status(void* status_host, int64_t timeout_us, ArStatus* st) {
  st->host = nullptr;
  if (status_host != nullptr) {
    void* d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, status_host, 0);
    if (e != hipSuccess)
      return fail(PE_ERR_INVALID_ARG, "allreduce: status words are not mapped pinned host memory (%s)",
                  hipGetErrorString(e));
    st->host = static_cast<uint32_t*>(d);
  }
  if (timeout_us <= 0) timeout_us = AR_DEFAULT_TIMEOUT_US;
  st->timeout_ticks = (unsigned long long)timeout_us * 100ull;
  return PE_OK;
}
}  // namespace pe

int pe_ar_selfcheck(void* const* peer_buffers, int rank, int world, int phase,
                    unsigned int* mismatches_dev, void* stream) {
  using namespace pe;
  PE_REQUIRE(peer_buffers && world >= 1 && world <= AR_MAX_RANKS && rank >= 0 && rank < world,
             PE_ERR_INVALID_ARG, "ar_selfcheck: bad arguments");
  PE_REQUIRE(phase == 0 || (phase == 1 && mismatches_dev), PE_ERR_INVALID_ARG,
             "ar_selfcheck: phase 0 writes, phase 1 verifies into mismatches_dev");
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) peers.buf[r] = nullptr;
  for (int r = 0; r < world; ++r) {
    PE_REQUIRE(peer_buffers[r] != nullptr, PE_ERR_INVALID_ARG, "ar_selfcheck: peer buffer %d is null", r);
    peers.buf[r] = static_cast<char*>(peer_buffers[r]);
  }
  hipLaunchKernelGGL(allreduce_selfcheck_kernel, dim3(1), dim3(AR_THREADS), 0,
                     static_cast<hipStream_t>(stream), peers, rank, world, phase, mismatches_dev);
  return check_launch("ar_selfcheck");
}

int pe_allreduce_oneshot(void* out, const void* in, int64_t count, int dtype,
                         void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                         unsigned int seq, void* status_host, int64_t timeout_us, void* stream) {
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
  ArStatus status;
  if (int rc = ar_status(status_host, timeout_us, &status)) return rc;
  const int64_t n_vec = bytes / 16;
  // one workgroup per 16 KiB, at most AR_MAX_BLOCKS: the slice boundaries are the same on all ranks
  int blocks = (int)((n_vec + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(allreduce_oneshot_kernel<T>, dim3((unsigned)blocks),
                                     dim3(AR_THREADS), 0, static_cast<hipStream_t>(stream), (T*)out,
                                     (const T*)in, n_vec, peers, rank, world, slot, (uint32_t)seq,
                                     status));
  return check_launch("allreduce_oneshot");
}

int pe_allreduce_twostage(void* out, const void* in, int64_t count, int dtype,
                          void* const* peer_buffers, int rank, int world, int64_t buffer_bytes,
                          unsigned int seq, void* status_host, int64_t timeout_us, void* stream) {
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
  ArStatus status;
  if (int rc = ar_status(status_host, timeout_us, &status)) return rc;
  // one workgroup per 16 KiB of a chunk, at most AR_MAX_BLOCKS
  int blocks = (int)((chunk_vec + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  PE_DISPATCH_ALL(dtype, T,
                  hipLaunchKernelGGL(allreduce_twostage_kernel<T>, dim3((unsigned)blocks),
                                     dim3(AR_THREADS), 0, static_cast<hipStream_t>(stream), (T*)out,
                                     (const T*)in, n_vec, chunk_vec, peers, rank, world, slot,
                                     (uint32_t)seq, status));
  return check_launch("allreduce_twostage");
}

}  // extern "C"
