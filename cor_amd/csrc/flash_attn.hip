// cor_amd — bf16 MFMA flash-attention kernels (hd = 64) for gfx950. Placeholder: entry points report
// "no kernel for this shape" so callers take the exact row-per-lane kernel (attention.hip).
#include "common.h"

int cor_flash_plain_bf16(const void*, long, long, const void*, long, long, const void*, long, long, void*, long, long, int, int, int,
                         int, int, float, hipStream_t) { return COR_ENOSUPPORT; }
int cor_flash_sam_bf16(const void*, void*, int, const void*, const float*, const float*, int, int, int, int, hipStream_t) {
  return COR_ENOSUPPORT;
}
