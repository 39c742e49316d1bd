// Shared host/device helpers for the paged-eviction kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/paged_evict.h"

namespace pe {

// ---- error reporting: every entry point returns a status and leaves a message ----
char* last_error_buf();
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
#define PE_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return ::pe::fail((code), __VA_ARGS__); \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PE_OK;
}

inline int elem_size(int dtype) { return dtype == PE_F32 ? 4 : 2; }

// ---- tunables from the environment, read once ------------------------------------
// A launch path must not call getenv (a linear scan of the environment) per launch.  An EnvInt
// reads its variable on first use and again after pe_reload_env() (tests and sweep tools that
// change a tunable inside one process call that).
int env_generation();  // bumped by pe_reload_env (defined in cache_kernels.hip)
struct EnvInt {
  const char* name;
  int dflt;
  int gen = -1;
  int val = 0;
  EnvInt(const char* n, int d) : name(n), dflt(d) {}
  int get() {
    const int g = env_generation();
    if (gen != g) {
      const char* v = getenv(name);
      val = v ? atoi(v) : dflt;
      gen = g;
    }
    return val;
  }
};

// per-file diagnostic words behind pe_device_diagnostics (defined in paged_decode.hip / evict.hip)
int decode_diag_fetch(unsigned int* out, int reset);
int evict_diag_fetch(unsigned int* out, int reset);
int state_diag_fetch(unsigned int* out, int reset);

// ---- storage types ---------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <typename T>
struct Traits;
template <>
struct Traits<__bf16> {
  using vec8 = bf16x8;
  using vec4 = bf16x4;
  static constexpr int code = PE_BF16;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Traits<_Float16> {
  using vec8 = f16x8;
  using vec4 = f16x4;
  static constexpr int code = PE_F16;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Traits<float> {
  static constexpr int code = PE_F32;
};

template <typename T>
__device__ __forceinline__ float to_f32(T v) {
  return (float)v;
}
template <typename T>
__device__ __forceinline__ T from_f32(float v) {
  return (T)v;
}
// Round an f32 value to the storage type and back (what torch does after each reduction on a
// low-precision tensor, and what a kernel computing in the storage type does after each
// operation).  The 16-bit value is made opaque to the optimiser: where the rounded value feeds
// straight into further f32 arithmetic the compiler otherwise drops the down/up conversion pair
// and contracts the surrounding multiply-add.
template <typename T>
__device__ __forceinline__ float round_to(float v) {
  if constexpr (sizeof(T) == 4) {
    asm volatile("" : "+v"(v));
    return v;
  } else {
    uint32_t bits = __builtin_bit_cast(uint16_t, (T)v);
    asm volatile("" : "+v"(bits));
    return (float)__builtin_bit_cast(T, (uint16_t)bits);
  }
}
template <typename T>
__device__ __forceinline__ float round_to_strict(float v) {
  return round_to<T>(v);
}

// One rotary-embedding pair in the storage type's arithmetic, as the reference kernel computes it
// (csrc/pos_encoding_kernels.cu:10-35): x*cos, y*sin, y*cos and x*sin are each rounded to T, and so
// are their difference and sum (separate multiplies and add/sub, no contraction).
template <typename T>
__device__ __forceinline__ void rot_pair(float x, float y, float c, float s, T& ox, T& oy) {
  const float xc = round_to_strict<T>(x * c);
  const float ys = round_to_strict<T>(y * s);
  const float yc = round_to_strict<T>(y * c);
  const float xs = round_to_strict<T>(x * s);
  ox = from_f32<T>(xc - ys);
  oy = from_f32<T>(yc + xs);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

}  // namespace pe

// dtype dispatch over the two 16-bit storage types (+ f32 where a kernel has it)
#define PE_DISPATCH_16(dtype, T, ...)                                   \
  switch (dtype) {                                                      \
    case PE_BF16: {                                                     \
      using T = __bf16;                                                 \
      __VA_ARGS__;                                                      \
    } break;                                                            \
    case PE_F16: {                                                      \
      using T = _Float16;                                               \
      __VA_ARGS__;                                                      \
    } break;                                                            \
    default:                                                            \
      return ::pe::fail(PE_ERR_UNSUPPORTED, "dtype %d not supported here", (int)(dtype)); \
  }
#define PE_DISPATCH_ALL(dtype, T, ...)                                  \
  switch (dtype) {                                                      \
    case PE_BF16: {                                                     \
      using T = __bf16;                                                 \
      __VA_ARGS__;                                                      \
    } break;                                                            \
    case PE_F16: {                                                      \
      using T = _Float16;                                               \
      __VA_ARGS__;                                                      \
    } break;                                                            \
    case PE_F32: {                                                      \
      using T = float;                                                  \
      __VA_ARGS__;                                                      \
    } break;                                                            \
    default:                                                            \
      return ::pe::fail(PE_ERR_UNSUPPORTED, "unknown dtype %d", (int)(dtype)); \
  }
