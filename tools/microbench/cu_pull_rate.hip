// How fast does ONE compute unit pull paged K/V slabs, and does the shape of the loads matter?
// Every workgroup walks its own list of block pairs (random 4 KiB K slabs and 4 KiB V slabs, the
// decode kernel's tile: two 16-token blocks of a 128-wide head in bf16 = 16 KiB), its waves
// interleaved over the pairs with two tiles in flight per wave, and only xors what it loaded.
//   pattern 0: sixteen 16-byte-per-lane loads, lanes contiguous (1 KiB per instruction)
//   pattern 1: the decode kernel's: K as eight contiguous 16-byte-per-lane loads, V as sixteen
//              8-byte-per-lane loads (row (lane&15) of a 16-row group, 8 bytes of its 32)
// Launches rotate over eight disjoint slab sets (the layers of a model): a set read again by the
// next launch would sit in the 256 MB memory-side cache.  "nontemporal" = the loads carry the nt bit.
//   pattern 2: V as eight 16-byte-per-lane loads (half a row each, lane groups 0/1 one block,
//              2/3 the other) and K with the token order that layout of P needs (each load
//              touches half of sixteen 128-byte lines)
// Build and run: hipcc --offload-arch=gfx950 -O3 -o cu_pull_rate cu_pull_rate.hip && ./cu_pull_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int SLAB = 4096;  // bytes of one (block, head) slab of K or of V

template <int PAT, bool NT>
struct Tile {
  u32x4 k[8];
  u32x4 v4[PAT == 1 ? 1 : 8];
  u32x2 v2[PAT == 1 ? 16 : 1];
  template <typename V>
  static __device__ V ld(const V* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
  }
  __device__ void load(const char* kc, const char* vc, int ba, int bb, int lane) {
    const char* ka = kc + (int64_t)ba * SLAB;
    const char* kb = kc + (int64_t)bb * SLAB;
    const char* va = vc + (int64_t)ba * SLAB;
    const char* vb = vc + (int64_t)bb * SLAB;
    const int i = lane & 15, g = lane >> 4;
    if (PAT == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k[j] = ld(reinterpret_cast<const u32x4*>(ka + j * 1024 + lane * 16));
        k[4 + j] = ld(reinterpret_cast<const u32x4*>(kb + j * 1024 + lane * 16));
        v4[j] = ld(reinterpret_cast<const u32x4*>(va + j * 1024 + lane * 16));
        v4[4 + j] = ld(reinterpret_cast<const u32x4*>(vb + j * 1024 + lane * 16));
      }
    } else if (PAT == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k[j] = ld(reinterpret_cast<const u32x4*>(ka + j * 1024 + lane * 16));
        k[4 + j] = ld(reinterpret_cast<const u32x4*>(kb + j * 1024 + lane * 16));
      }
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        v2[2 * dt] = ld(reinterpret_cast<const u32x2*>(va + (16 * dt + i) * 32 + 8 * g));
        v2[2 * dt + 1] = ld(reinterpret_cast<const u32x2*>(vb + (16 * dt + i) * 32 + 8 * g));
      }
    } else {
      const char* kblk = (i >> 3) ? kb : ka;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int token = (i & 3) + 8 * ((i >> 2) & 1) + 4 * m;
          k[4 * m + ks] = ld(reinterpret_cast<const u32x4*>(kblk + (4 * ks + g) * 256 + token * 16));
        }
      const char* vblk = (g >> 1) ? vb : va;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
        v4[dt] = ld(reinterpret_cast<const u32x4*>(vblk + (16 * dt + i) * 32 + 16 * (g & 1)));
    }
  }
  __device__ uint32_t fold() const {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s ^= k[j][0] ^ k[j][1] ^ k[j][2] ^ k[j][3];
    if (PAT == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s ^= v2[j][0] ^ v2[j][1];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) s ^= v4[j][0] ^ v4[j][1] ^ v4[j][2] ^ v4[j][3];
    }
    return s;
  }
};

template <int PAT, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void pull(const char* kc, const char* vc, const int* table,
                                                   int pairs, uint32_t* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int* row = table + (int64_t)blockIdx.x * 2 * pairs;
  Tile<PAT, NT> a, b;
  uint32_t s = 0;
  int p = w;
  if (p < pairs) a.load(kc, vc, row[2 * p], row[2 * p + 1], lane);
  while (p < pairs) {
    const int p1 = p + WAVES, p2 = p + 2 * WAVES;
    if (p1 < pairs) b.load(kc, vc, row[2 * p1], row[2 * p1 + 1], lane);
    s ^= a.fold();
    if (p2 < pairs) a.load(kc, vc, row[2 * p2], row[2 * p2 + 1], lane);
    if (p1 < pairs) s ^= b.fold();
    p = p2;
  }
  if (s == 0x12345678u) sink[threadIdx.x] = s;
}

constexpr int LAYERS = 8;          // launches rotate over this many disjoint slab sets, as the layers
constexpr int64_t TABLE_STRIDE = 512 * 2 * 97;   // of a model do: the 256 MB memory-side cache cannot help

template <int PAT, int WAVES, bool NT = false>
double run(const char* kc, const char* vc, const int* table, int wgs, int pairs, uint32_t* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) pull<PAT, WAVES, NT><<<wgs, WAVES * 64>>>(kc, vc, table, pairs, sink);
  const int reps = 24;
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    pull<PAT, WAVES, NT><<<wgs, WAVES * 64>>>(kc, vc, table + (i % LAYERS) * TABLE_STRIDE, pairs, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  const int nblocks = 1 << 19;                 // 2 GiB of K and 2 GiB of V
  char *kc, *vc; int* table; uint32_t* sink;
  (void)hipMalloc(&kc, (size_t)nblocks * SLAB); (void)hipMalloc(&vc, (size_t)nblocks * SLAB);
  (void)hipMemset(kc, 1, (size_t)nblocks * SLAB); (void)hipMemset(vc, 2, (size_t)nblocks * SLAB);
  (void)hipMalloc(&sink, 4096 * 4);
  const int max_wgs = 512, max_pairs = 97;
  std::vector<int> h((size_t)LAYERS * TABLE_STRIDE);
  srand(7);
  for (auto& x : h) x = (int)(((unsigned)rand() * 2654435761u) % nblocks);
  (void)hipMalloc(&table, h.size() * 4);
  (void)hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  // paged-cache addressing: a slab is (block, head) with 8 heads per block, so head h only ever
  // touches addresses = h * 4 KiB mod 32 KiB.  Workgroups are dealt to the 8 XCDs round-robin:
  // with head = workgroup % 8 every XCD pulls one residue class only; with the head rotated by
  // the sequence every XCD pulls all of them.
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<int> hh(h.size());
    for (size_t x = 0; x < hh.size(); ++x) {
      const int w = (int)((x % TABLE_STRIDE) / (2 * max_pairs));
      const int head = mode == 0 ? w % 8 : (w + w / 8) % 8;
      hh[x] = (h[x] / 8) * 8 + head;
    }
    (void)hipMemcpy(table, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    for (int wgs : {256, 512}) {
      const int pairs = 25;
      const double bytes = (double)pairs * 4 * SLAB;
      const double us4 = run<1, 4, true>(kc, vc, table, wgs, pairs, sink);
      const double us8 = run<1, 8, true>(kc, vc, table, wgs, pairs, sink);
      printf("head-strided slabs, head %s: %d workgroups x %d pairs: 4 waves %.2f us (%.2f TB/s), 8 waves %.2f us (%.2f TB/s)\n",
             mode == 0 ? "= workgroup %% 8 (one residue class per XCD)" : "rotated by the sequence",
             wgs, pairs, us4, bytes * wgs / us4 * 1e-6, us8, bytes * wgs / us8 * 1e-6);
    }
  }
  (void)hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("pattern waves workgroups pairs |   us    GB/s per workgroup   TB/s total\n");
  const int shapes[][2] = {{64, 97}, {64, 25}, {128, 49}, {256, 25}, {512, 1}, {512, 13}, {512, 25}, {512, 49}};
  for (auto& sh : shapes) {
    const int wgs = sh[0], pairs = sh[1];
    const double bytes = (double)pairs * 4 * SLAB;
#define RUN(P, W)                                                                              \
  {                                                                                            \
    const double us = run<P, W>(kc, vc, table, wgs, pairs, sink);                              \
    printf("%d %2d %4d %3d | %7.2f  %7.1f  %6.2f\n", P, W, wgs, pairs, us, bytes / us * 1e-3, \
           bytes * wgs / us * 1e-6);                                                           \
  }
    RUN(0, 4) RUN(1, 4) RUN(2, 4) RUN(0, 8) RUN(1, 8) RUN(2, 8)
    {
      const double a = run<1, 4, true>(kc, vc, table, wgs, pairs, sink), b = run<1, 8, true>(kc, vc, table, wgs, pairs, sink);
      printf("1 nontemporal %4d %3d | 4 waves %7.2f us %6.2f TB/s, 8 waves %7.2f us %6.2f TB/s\n", wgs, pairs, a,
             bytes * wgs / a * 1e-6, b, bytes * wgs / b * 1e-6);
    }
  }
  return 0;
}
