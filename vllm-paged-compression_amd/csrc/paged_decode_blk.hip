// Matrix-core decode kernel at 8- and 32-token blocks (the other page sizes of the reference's
// scripts/diff_page_sizes.sh:16) for the head sizes beyond 64 / 128 whose rows are whole k-steps:
// 32, 96, 192, 256 (vllm/attention/ops/paged_attn.py:37).  4- or 8-wave workgroups, KV splits and
// fused KV write as everywhere else; no rotating form.  A translation unit of its own so that the
// instantiations compile beside the others.
#include "paged_decode_kernel.h"

namespace pe {

int launch_decode_other_blocks(const DecLaunch& a, int head_size, int block_size, int waves, int dtype) {
#define PE_HB(T, HS, B)                                 \
  do {                                                  \
    if (waves == 8) launch_dec<T, HS, 8, false, B>(a);  \
    else launch_dec<T, HS, DEC_WAVES, false, B>(a);     \
  } while (0)
#define PE_HSB(T, HS)               \
  do {                              \
    if (block_size == 8) PE_HB(T, HS, 8); \
    else PE_HB(T, HS, 32);          \
  } while (0)
  if (block_size != 8 && block_size != 32)
    return fail(PE_ERR_UNSUPPORTED, "decode: block size %d has no matrix-core kernel here", block_size);
  PE_DISPATCH_16(dtype, T, {
    switch (head_size) {
      case 32: PE_HSB(T, 32); break;
      case 96: PE_HSB(T, 96); break;
      case 192: PE_HSB(T, 192); break;
      case 256: PE_HSB(T, 256); break;
      default:
        return fail(PE_ERR_UNSUPPORTED, "decode: head size %d has no matrix-core kernel at %d-token blocks",
                    head_size, block_size);
    }
  });
#undef PE_HSB
#undef PE_HB
  return PE_OK;
}

int decode_blk_diag_fetch(unsigned int* out, int reset) { return decode_diag_fetch_tu(out, reset); }

}  // namespace pe
