// Does a wave's vector work run in the shadow of its own matrix instructions when the matrix
// accumulators live in the AccVGPR half of the register file (so that the matrix instruction's
// C/D traffic does not compete for the ports the vector instructions read)?  One workgroup of
// 4 waves (one per SIMD) on one CU; shader clocks per group of 1 MFMA + KV vector instructions.
// Build and run: hipcc --offload-arch=gfx950 -O3 -o mfma_agpr_overlap mfma_agpr_overlap.hip && ./mfma_agpr_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ACC: 0 = accumulators in VGPRs (builtin), 1 = in AGPRs (inline asm, "a" constraint)
// VOP: 0 = v_fma_f32, 1 = v_exp_f32, 2 = ds_read_b128 (LDS fragment read)
template <int ACC, int VOP, int KV, bool MFMA>
__global__ void k(uint64_t* out, float* sink, int iters) {
  __shared__ f32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  f32x4 ld[4] = {};
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-4f + i;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MFMA) {
        if (ACC == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int q = 0; q < KV; ++q) {
        const int r = (j * KV + q) & 7;
        if (VOP == 0) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(v[r]));
        else if (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(j * KV + q) & 3]) : "v"((threadIdx.x * 16 + 1024 * q) & 16383));
      }
    }
    if (VOP == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + ld[j][0];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int ACC, int VOP, int KV, bool MFMA>
double run() {
  uint64_t* out; float* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 4096);
  const int iters = 4000;
  k<ACC, VOP, KV, MFMA><<<1, 256>>>(out, sink, 10);
  k<ACC, VOP, KV, MFMA><<<1, 256>>>(out, sink, iters);
  (void)hipDeviceSynchronize();
  uint64_t h[4];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(out); (void)hipFree(sink);
  return (double)h[0] / (iters * 4.0);
}

template <int VOP>
void table(const char* name) {
  printf("%-14s alone (no MFMA)  KV=2 %5.1f  KV=4 %5.1f\n", name, run<0, VOP, 2, false>(), run<0, VOP, 4, false>());
  printf("%-14s + MFMA acc VGPR  KV=0 %5.1f  KV=1 %5.1f  KV=2 %5.1f  KV=3 %5.1f  KV=4 %5.1f\n", name,
         run<0, VOP, 0, true>(), run<0, VOP, 1, true>(), run<0, VOP, 2, true>(), run<0, VOP, 3, true>(), run<0, VOP, 4, true>());
  printf("%-14s + MFMA acc AGPR  KV=0 %5.1f  KV=1 %5.1f  KV=2 %5.1f  KV=3 %5.1f  KV=4 %5.1f\n", name,
         run<1, VOP, 0, true>(), run<1, VOP, 1, true>(), run<1, VOP, 2, true>(), run<1, VOP, 3, true>(), run<1, VOP, 4, true>());
}

int main() {
  printf("clocks per group (1 v_mfma_f32_16x16x32_bf16 + KV ops of the same wave), one wave per SIMD\n");
  table<0>("v_fma_f32");
  table<1>("v_exp_f32");
  table<2>("ds_read_b128");
  return 0;
}
