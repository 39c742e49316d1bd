// LDS read-rate microbenchmark: bytes per clock per CU for the read instructions the prefill kernel uses.
// Build and run:  hipcc --offload-arch=gfx950 -O3 -o lds_read_rate lds_read_rate.hip && ./lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int g = lane >> 4, c = lane & 15;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (MODE == 0) {         // ds_read_b128, K pattern: row c (256 B rows), chunk (4j+g)^c
        const int off = ((j >> 2) * 16 + c) * 256 + ((((j & 3) * 4 + g) ^ c) << 4) + wave * 8192;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + off);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      } else if constexpr (MODE == 1) {  // ds_read_b64_tr_b16, V pattern
        const int tq = c >> 2, tp = c & 3;
        const int row = 4 * g + tq + (j & 1) * 16;
        const int dt = j >> 1;
        const int ch = 2 * dt + (tp >> 1);
        const int x = (row & 7) << 1;
        const int off = row * 256 + ((ch ^ x) << 4) + (tp & 1) * 8 + wave * 8192;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
        acc ^= (uint32_t)v[0] ^ ((uint32_t)v[1] << 3) ^ ((uint32_t)v[2] << 5) ^ ((uint32_t)v[3] << 7);
      } else {                            // plain ds_read_b64, same addresses as MODE 1
        const int tq = c >> 2, tp = c & 3;
        const int row = 4 * g + tq + (j & 1) * 16;
        const int dt = j >> 1;
        const int ch = 2 * dt + (tp >> 1);
        const int x = (row & 7) << 1;
        const int off = row * 256 + ((ch ^ x) << 4) + (tp & 1) * 8 + wave * 8192;
        u32x2 v = *reinterpret_cast<const u32x2*>(smem + off);
        acc ^= v[0] ^ v[1];
      }
    }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x12345678u) out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
void run(const char* name, int bytes_per_inst) {
  uint32_t* out; hipMalloc(&out, 512 * 256 * 4);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  rate_kernel<MODE><<<512, 256, 65536>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  rate_kernel<MODE><<<512, 256, 65536>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double bytes_per_cu = 8.0 /*waves*/ * iters * 8 * bytes_per_inst;   // 2 WGs x 4 waves per CU
  printf("%-22s %.3f ms  %.1f B/clk/CU at %d MHz (nominal), %.1f at 2.0 GHz\n", name, ms,
         bytes_per_cu / (ms * 1e-3 * clk_khz * 1e3), clk_khz / 1000, bytes_per_cu / (ms * 1e-3 * 2.0e9));
  hipFree(out);
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<0>("ds_read_b128 (K)", 1024);
  run<1>("ds_read_b64_tr_b16 (V)", 512);
  run<2>("ds_read_b64 (V addrs)", 512);
  return 0;
}
