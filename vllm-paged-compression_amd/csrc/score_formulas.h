// Token score formulas of the token-level prune and of the decode-time block merge.
//
// HEAD's KVCachePruner has four (kvcache_prunner.py:29-49); the README-era class adds L1 norms,
// key/value ratios, a cosine similarity and the token position (get_score,
// vllm/attention/kvcache_prunner_old.py:23-134).  A formula is a small recipe over the four row
// norms ||K||_1, ||K||_2, ||V||_1, ||V||_2 - each rounded to the storage type, as torch.norm on a
// 16-bit tensor returns it - with the result of every further operation rounded again, which is
// where torch rounds.  The recipes are the reference's AS WRITTEN: several of its helpers pass
// (key, value) to functions that apply the "value" norm to the first argument and the "key" norm to
// the second (kvcache_prunner_old.py:59-72), and key_l1_div_value_l1 takes the L2 norm of the key
// (:47-48); the names below are the reference's method names, the comments say what is computed.
#pragma once
#include "pe_common.h"

namespace pe {

enum { FN_KL1 = 0, FN_KL2 = 1, FN_VL1 = 2, FN_VL2 = 3 };
enum { FOP_SINGLE = 0, FOP_PLUS = 1, FOP_DIV = 2, FOP_INVERSE = 3, FOP_COSINE = 4, FOP_POSITION = 5 };

struct FormulaSpec {
  int op, a, b;
};

__host__ __device__ inline FormulaSpec formula_spec(int f) {
  switch (f) {
    case PE_FORMULA_VALUE_L2: return {FOP_SINGLE, FN_VL2, 0};
    case PE_FORMULA_KEY_L2: return {FOP_SINGLE, FN_KL2, 0};
    case PE_FORMULA_VALUE_L2_PLUS_KEY_L2: return {FOP_PLUS, FN_KL2, FN_VL2};
    case PE_FORMULA_INVERSE_KEY_L2: return {FOP_INVERSE, FN_KL2, 0};
    case PE_FORMULA_KEY_L1: return {FOP_SINGLE, FN_KL1, 0};
    case PE_FORMULA_VALUE_L1: return {FOP_SINGLE, FN_VL1, 0};
    case PE_FORMULA_INVERSE_KEY_L1: return {FOP_INVERSE, FN_KL1, 0};
    case PE_FORMULA_KEY_L1_DIV_VALUE_L1: return {FOP_DIV, FN_KL2, FN_VL1};  // ||K||_2 / ||V||_1 (:47-48)
    case PE_FORMULA_KEY_L1_DIV_VALUE_L2: return {FOP_DIV, FN_KL1, FN_VL2};
    case PE_FORMULA_KEY_L2_DIV_VALUE_L1: return {FOP_DIV, FN_KL2, FN_VL1};
    case PE_FORMULA_KEY_L2_DIV_VALUE_L2: return {FOP_DIV, FN_KL2, FN_VL2};
    case PE_FORMULA_VALUE_L1_DIV_KEY_L1: return {FOP_DIV, FN_KL1, FN_VL1};  // ||K||_1 / ||V||_1 (:59-60)
    case PE_FORMULA_VALUE_L1_DIV_KEY_L2: return {FOP_DIV, FN_KL1, FN_VL2};  // ||K||_1 / ||V||_2 (:62-63)
    case PE_FORMULA_VALUE_L2_DIV_KEY_L1: return {FOP_DIV, FN_KL2, FN_VL1};  // ||K||_2 / ||V||_1 (:65-66)
    case PE_FORMULA_VALUE_L2_DIV_KEY_L2: return {FOP_DIV, FN_KL2, FN_VL2};  // ||K||_2 / ||V||_2 (:68-69)
    case PE_FORMULA_VALUE_L1_PLUS_KEY_L1: return {FOP_PLUS, FN_KL1, FN_VL1};
    case PE_FORMULA_COSINE: return {FOP_COSINE, 0, 0};
    case PE_FORMULA_POSITION: return {FOP_POSITION, 0, 0};
    default: return {-1, 0, 0};
  }
}

__host__ __device__ inline bool formula_known(int f) { return formula_spec(f).op >= 0; }

// Score of one (token, head) row pair.  K(d), V(d): element d of the key / value row as f32
// (d ascending: sums run in that order).  pos1: 1-based position of the token in the scored slice.
template <typename T, typename KF, typename VF>
__device__ __forceinline__ float formula_score(int formula, int head_size, int pos1, KF K, VF V) {
  const FormulaSpec sp = formula_spec(formula);
  if (sp.op == FOP_POSITION) return (float)pos1;
  if (sp.op == FOP_COSINE) {
    // F.normalize(x, dim=-1) = x / max(||x||_2, 1e-12), every quotient rounded; the products
    // rounded; their sum accumulated in f32 and rounded (kvcache_prunner_old.py:23-27)
    float kk = 0.f, vv = 0.f;
    for (int d = 0; d < head_size; ++d) {
      const float x = K(d), y = V(d);
      kk += x * x;
      vv += y * y;
    }
    const float kn = fmaxf(round_to<T>(sqrtf(kk)), 1e-12f), vn = fmaxf(round_to<T>(sqrtf(vv)), 1e-12f);
    float acc = 0.f;
    for (int d = 0; d < head_size; ++d)
      acc += round_to<T>(round_to<T>(K(d) / kn) * round_to<T>(V(d) / vn));
    return round_to<T>(acc);
  }
  auto norm = [&](int which) {
    float s = 0.f;
    const bool key = which == FN_KL1 || which == FN_KL2;
    const bool l1 = which == FN_KL1 || which == FN_VL1;
    for (int d = 0; d < head_size; ++d) {
      const float x = key ? K(d) : V(d);
      s += l1 ? fabsf(x) : x * x;
    }
    return round_to<T>(l1 ? s : sqrtf(s));
  };
  const float a = norm(sp.a);
  if (sp.op == FOP_SINGLE) return a;
  if (sp.op == FOP_INVERSE) return round_to<T>(1.0f / round_to<T>(a + 1e-8f));
  const float b = norm(sp.b);
  if (sp.op == FOP_PLUS) return round_to<T>(a + b);
  return round_to<T>(a / round_to<T>(b + 1e-8f));
}

// f32 bits -> unsigned key with the order of the values (negative scores exist: the cosine)
__host__ __device__ inline uint32_t sortable_bits(float v) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(v);
#else
  __builtin_memcpy(&u, &v, 4);
#endif
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

}  // namespace pe
