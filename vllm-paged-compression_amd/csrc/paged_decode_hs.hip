// Matrix-core decode kernel for the head sizes of the reference's list
// (vllm/attention/ops/paged_attn.py:37) that the models of its scripts do not use: 32, 80, 96,
// 112, 120, 192 - 16-token blocks, 4- or 8-wave workgroups, KV splits and fused KV write as for
// the other sizes.  A translation unit of its own so that the instantiations compile beside
// those of paged_decode.hip.
#include "paged_decode_kernel.h"

namespace pe {

int launch_decode_other_heads(const DecLaunch& a, int head_size, int waves, int dtype) {
#define PE_HS(T, HS)                                     \
  do {                                                   \
    if (waves == 8) launch_dec<T, HS, 8, false, 16>(a);  \
    else launch_dec<T, HS, DEC_WAVES, false, 16>(a);     \
  } while (0)
  PE_DISPATCH_16(dtype, T, {
    switch (head_size) {
      case 32: PE_HS(T, 32); break;
      case 80: PE_HS(T, 80); break;
      case 96: PE_HS(T, 96); break;
      case 112: PE_HS(T, 112); break;
      case 120: PE_HS(T, 120); break;
      case 192: PE_HS(T, 192); break;
      default: return fail(PE_ERR_UNSUPPORTED, "decode: head size %d has no matrix-core kernel", head_size);
    }
  });
#undef PE_HS
  return PE_OK;
}

int decode_hs_diag_fetch(unsigned int* out, int reset) { return decode_diag_fetch_tu(out, reset); }

}  // namespace pe
