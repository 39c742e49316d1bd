// How much do the matrix pipe and the vector pipe of a SIMD overlap?  One workgroup on one CU.
// Build and run:  hipcc --offload-arch=gfx950 -O3 -o matrix_vector_overlap matrix_vector_overlap.hip && ./matrix_vector_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// MODE 0: MFMA only.  1: v_fma only.  2: v_exp only.  3: per MFMA, KV fma in the same wave.
// 4: per MFMA, KV exp in the same wave.  5: even waves MFMA only, odd waves (other slot of the same SIMD) fma only.
// 6: like 5 with exp.
template <int MODE, int KV>
__global__ void k(uint64_t* out, float* sink, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-4f + i;
  const bool do_m = MODE == 0 || MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && (wave >> 2) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || ((MODE == 5 || MODE == 6) && (wave >> 2) == 1);
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  if (MODE == 3 || MODE == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < KV; ++q) {
          const int r = (j * KV + q) & 7;
          if (MODE == 3) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
          else v[r] = __builtin_amdgcn_exp2f(v[r]) * 0.f + v[r];
        }
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4 * KV; ++q) {
        const int r = q & 7;
        if (MODE == 1 || MODE == 5) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        else v[r] = __builtin_amdgcn_exp2f(v[r]);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int MODE, int KV>
void run(const char* name, int threads) {
  uint64_t* out; float* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 4096);
  const int iters = 4000;
  k<MODE, KV><<<1, threads>>>(out, sink, 10);
  k<MODE, KV><<<1, threads>>>(out, sink, iters);
  hipDeviceSynchronize();
  uint64_t h[16];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s", name);
  for (int w = 0; w < threads / 64; w += (threads > 256 ? 4 : 1))
    printf("  wave%d: %.1f clk/group", w, (double)h[w] / (iters * 4.0));
  printf("\n");
  hipFree(out); hipFree(sink);
}
int main() {
  printf("group = 1 MFMA 16x16x32 bf16 and/or KV vector ops; s_memtime ticks (100 MHz?) or shader clocks per group\n");
  run<0, 1>("MFMA only, 1 wave/SIMD", 256);
  run<1, 1>("1 fma per group, 1 wave/SIMD", 256);
  run<1, 4>("4 fma per group", 256);
  run<2, 1>("1 exp per group", 256);
  run<2, 4>("4 exp per group", 256);
  run<3, 1>("same wave: MFMA + 1 fma", 256);
  run<3, 2>("same wave: MFMA + 2 fma", 256);
  run<3, 3>("same wave: MFMA + 3 fma", 256);
  run<3, 4>("same wave: MFMA + 4 fma", 256);
  run<3, 6>("same wave: MFMA + 6 fma", 256);
  run<4, 1>("same wave: MFMA + 1 exp(+fma)", 256);
  run<4, 2>("same wave: MFMA + 2 exp(+fma)", 256);
  run<0, 1>("MFMA only, 2 waves/SIMD", 512);
  run<5, 2>("wave A MFMA | wave B 2 fma per group", 512);
  run<5, 3>("wave A MFMA | wave B 3 fma per group", 512);
  run<5, 4>("wave A MFMA | wave B 4 fma per group", 512);
  run<6, 1>("wave A MFMA | wave B 1 exp per group", 512);
  run<6, 2>("wave A MFMA | wave B 2 exp per group", 512);
  return 0;
}
