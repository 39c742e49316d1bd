// What does it cost to hand a partial result from one workgroup to another INSIDE a launch?
// (The decode kernel merges its KV splits in a second launch of ~3-4 us; an in-launch merge has
// to beat that.)  A producer workgroup stores a 2 KiB payload and then a flag; a consumer
// workgroup that is already polling the flag reads the payload when it sees it.  Times are
// s_memrealtime ticks (100 MHz, the same clock on every XCD): flag store -> flag seen, and
// flag seen -> payload read.
//   scope 1: workgroup-scope accesses (sc0: bypass the CU's L1, served by the XCD's L2) - only
//            meaningful when both workgroups run on ONE XCD
//   scope 2: agent-scope accesses (sc1), the scope HIP's memory model needs across XCDs
//   placement: consumer on the producer's XCD (workgroup id + 8) or on the next XCD (id + 1)
//   load: idle chip, or every CU also running a loop that streams random 4 KiB slabs
// Every spin is bounded; a pair that gives up is counted, not waited for.
// Build and run: hipcc --offload-arch=gfx950 -O3 -o xwg_handoff xwg_handoff.hip && ./xwg_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int SCOPE>
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
  if (SCOPE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int SCOPE>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
  if (SCOPE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Rec { uint64_t t_store, t_seen, t_read; uint32_t xcc_p, xcc_c, ok, gave_up; };

// pairs: workgroup ids [0, 8*npairs_per_xcd) are producers or consumers by (id / dist) & 1;
// workgroups past n_pair_wgs stream memory (the load)
template <int SCOPE>
__global__ __launch_bounds__(256) void handoff(uint32_t* flags, uint32_t* payload, Rec* rec, int dist,
                                               int n_pair_wgs, uint32_t tag, const char* slabs,
                                               const int* table, int pairs, uint32_t* sink) {
  const int wg = blockIdx.x;
  if (wg >= n_pair_wgs) {  // background load: the decode kernel's tile loop without the arithmetic
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int* row = table + (int64_t)(wg - n_pair_wgs) * pairs;
    uint32_t s = 0;
    for (int p = w; p < pairs; p += 4) {
      const char* a = slabs + (int64_t)row[p] * 4096;
      u32x4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + j * 1024 + lane * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) s ^= x[j][0] ^ x[j][1] ^ x[j][2] ^ x[j][3];
    }
    if (s == 0x12345678u) sink[threadIdx.x] = s;
    return;
  }
  const int role = (wg / dist) & 1;               // 0 producer, 1 consumer
  const int pair = role ? wg - dist : wg;         // the producer's id names the pair
  uint32_t* flag = flags + pair * 32;             // one 128-byte line per pair
  uint32_t* data = payload + (int64_t)pair * 512; // 2 KiB
  const int t = threadIdx.x;
  if (role == 0) {
    if (t == 0) rec[pair].xcc_p = xcc_id();
    // let the consumer reach its polling loop first
    const uint64_t t0 = now();
    while (now() - t0 < 400) {}  // 4 us
#pragma unroll
    for (int j = 0; j < 2; ++j) st<SCOPE>(data + j * 256 + t, tag + j * 256 + t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      rec[pair].t_store = now();
      st<SCOPE>(flag, tag);
    }
  } else {
    __shared__ uint32_t seen;
    if (t == 0) {
      rec[pair].xcc_c = xcc_id();
      uint32_t v = 0;
      int spins = 0;
      while ((v = ld<SCOPE>(flag)) != tag && ++spins < 200000) {}
      rec[pair].t_seen = now();
      rec[pair].gave_up = v != tag;
      seen = v == tag;
    }
    __syncthreads();
    uint32_t bad = 0;
    if (seen) {
#pragma unroll
      for (int j = 0; j < 2; ++j) bad |= ld<SCOPE>(data + j * 256 + t) != tag + j * 256 + t;
    }
    const int any_bad = __syncthreads_or((int)bad);
    if (t == 0) {
      rec[pair].t_read = now();
      rec[pair].ok = seen && !any_bad;
    }
  }
}

template <int SCOPE>
void run(const char* what, int dist, bool loaded, uint32_t* flags, uint32_t* payload, Rec* rec,
         const char* slabs, const int* table, uint32_t* sink, uint32_t& tag) {
  const int n_pair_wgs = 256;  // 128 pairs, one workgroup per CU
  const int bg = loaded ? 512 : 0, pairs = 200;
  std::vector<double> seen, read;
  int ok = 0, gave_up = 0, same = 0, n = 0;
  for (int rep = 0; rep < 12; ++rep) {
    tag += 1000;
    (void)hipMemset(rec, 0, sizeof(Rec) * 256);
    handoff<SCOPE><<<n_pair_wgs + bg, 256>>>(flags, payload, rec, dist, n_pair_wgs, tag, slabs, table, pairs, sink);
    (void)hipDeviceSynchronize();
    std::vector<Rec> h(256);
    (void)hipMemcpy(h.data(), rec, sizeof(Rec) * 256, hipMemcpyDeviceToHost);
    if (rep < 2) continue;
    for (int wg = 0; wg < n_pair_wgs; ++wg) {
      if ((wg / dist) & 1) continue;
      const Rec& r = h[wg];
      ++n;
      gave_up += r.gave_up;
      ok += r.ok;
      same += r.xcc_p == r.xcc_c;
      if (r.ok && r.t_seen >= r.t_store) {
        seen.push_back((r.t_seen - r.t_store) * 0.01);
        read.push_back((r.t_read - r.t_seen) * 0.01);
      }
    }
  }
  auto med = [](std::vector<double>& v, double q) {
    if (v.empty()) return -1.0;
    std::sort(v.begin(), v.end());
    return v[(size_t)(q * (v.size() - 1))];
  };
  printf("%-44s %-6s pairs %4d same-XCD %4d correct %4d gave up %3d | flag seen after: median %.2f us, p90 %.2f | payload read: median %.2f us, p90 %.2f\n",
         what, loaded ? "loaded" : "idle", n, same, ok, gave_up, med(seen, 0.5), med(seen, 0.9), med(read, 0.5), med(read, 0.9));
}

int main() {
  uint32_t *flags, *payload, *sink; Rec* rec; char* slabs; int* table;
  (void)hipMalloc(&flags, 256 * 128); (void)hipMemset(flags, 0, 256 * 128);
  (void)hipMalloc(&payload, 256 * 2048); (void)hipMalloc(&rec, sizeof(Rec) * 256);
  (void)hipMalloc(&sink, 4096);
  const int nslabs = 1 << 18;
  (void)hipMalloc(&slabs, (size_t)nslabs * 4096); (void)hipMemset(slabs, 1, (size_t)nslabs * 4096);
  std::vector<int> h(512 * 200);
  srand(3);
  for (auto& x : h) x = (int)(((unsigned)rand() * 2654435761u) % nslabs);
  (void)hipMalloc(&table, h.size() * 4);
  (void)hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  uint32_t tag = 1;
  for (int loaded = 0; loaded < 2; ++loaded) {
    run<1>("workgroup-scope accesses, consumer on the producer's XCD", 8, loaded, flags, payload, rec, slabs, table, sink, tag);
    run<2>("agent-scope accesses, consumer on the producer's XCD", 8, loaded, flags, payload, rec, slabs, table, sink, tag);
    run<2>("agent-scope accesses, consumer on the next XCD", 1, loaded, flags, payload, rec, slabs, table, sink, tag);
    run<1>("workgroup-scope accesses, consumer on the next XCD", 1, loaded, flags, payload, rec, slabs, table, sink, tag);
  }
  return 0;
}
