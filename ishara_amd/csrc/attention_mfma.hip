// MFMA flash-attention forward (bf16).  Placeholder until the kernel lands: fails loudly.
#include "kernels.h"
int launch_attn_fwd_mfma(const void*, const void*, const void*, void*, float*, int, int, int, int, float, DropSpec, hipStream_t) {
    ishara_set_error("attention impl 1 (MFMA) is not built in this revision; use attn_impl=0");
    return -1;
}
