// temporary: entry points still being written
#include "pe_common.h"
extern "C" {
int pe_prefill_causal(void*, const void*, const void*, const void*, const int32_t*, const int32_t*,
                      int, int, int, int, int, int64_t, int64_t, int64_t, int64_t, float, int, void*) {
  return pe::fail(PE_ERR_UNSUPPORTED, "pe_prefill_causal: not built yet");
}
int64_t pe_prompt_token_select_workspace_bytes(int64_t, int) { return 0; }
int pe_prompt_token_select(int32_t*, const void*, const void*, const int32_t*, const int32_t*, int,
                           int, int, int, int, int64_t, int64_t, int, int, int, void*, int64_t, void*) {
  return pe::fail(PE_ERR_UNSUPPORTED, "pe_prompt_token_select: not built yet");
}
}
