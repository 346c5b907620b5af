// cor_amd — input pre-processing on the GPU: Pillow-BILINEAR resize of uint8 images (what
// torchvision.transforms.Resize does to a PIL image), ToTensor, Normalize. ref: utils/dataloader.py:266-293, 349-350.
//
// Integer work, bit-exact with Pillow (src/libImaging/Resample.c, 8 bits per channel): the host computes Pillow's
// fixed-point coefficient tables in double precision (cor_amd/preprocess.py: precompute_coeffs + normalize_coeffs_8bpc,
// PRECISION_BITS = 22); pass 1 resamples rows, rounds to uint8, pass 2 resamples columns of that uint8 image, rounds to
// uint8, then (x / 255 - mean) / std in IEEE float32. HBM-bound byte work: one thread per output pixel, taps walk
// consecutive bytes (pass 1) or one byte per row with the threads of a wave on consecutive columns (pass 2: coalesced).
#include "common.h"

namespace {

constexpr int PB = 22;                         // Pillow: PRECISION_BITS = 32 - 8 - 2
__device__ __forceinline__ unsigned char clip8(int acc) {
  const int v = acc >> PB;                     // arithmetic shift, then clamp: Pillow's clip8 lookup
  return (unsigned char)min(max(v, 0), 255);
}

// in [H, W, C] u8 -> out [H, OW, C] u8
template <int C>
__global__ void __launch_bounds__(256) resample_h_kernel(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk,
                                                         int ksize, int H, int W, int OW) {
  const long total = (long)H * OW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int y = (int)(i / OW), xx = (int)(i - (long)y * OW);
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const unsigned char* p = in + ((long)y * W + x0) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (PB - 1);
    for (int t = 0; t < n; ++t) {
      const int kv = k[t];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)p[t * C + c] * kv;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[i * C + c] = clip8(acc[c]);
  }
}

// in [H, W, C] u8 -> out_f32 [C, OH, W] ((v/255 - mean)/std, or v/255 when mean == nullptr) and / or out_u8 [OH, W, C]
template <int C>
__global__ void __launch_bounds__(256) resample_v_kernel(const unsigned char* in, float* out_f32, unsigned char* out_u8, const int* bounds,
                                                         const int* kk, int ksize, int H, int W, int OH, const float* mean, const float* stdv) {
  const long total = (long)OH * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int yy = (int)(i / W), x = (int)(i - (long)yy * W);
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    const unsigned char* p = in + ((long)y0 * W + x) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (PB - 1);
    for (int t = 0; t < n; ++t) {
      const int kv = k[t];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)p[(long)t * W * C + c] * kv;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const unsigned char v = clip8(acc[c]);
      if (out_u8) out_u8[i * C + c] = v;
      if (out_f32) {
        float f = (float)v / 255.0f;                                   // ToTensor
        if (mean) f = (f - mean[c]) / stdv[c];                         // Normalize (IEEE float32 ops, as torch does them)
        out_f32[((long)c * OH + yy) * W + x] = f;
      }
    }
  }
}

int grid_for(long total) { return (int)min((total + 255) / 256, 65536L); }

}  // namespace

extern "C" int cor_resample_rows_u8(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, int ksize, int H, int W,
                                    int C, int OW, void* stream) {
  if (!in || !out || !bounds || !kk || ksize <= 0 || H <= 0 || W <= 0 || OW <= 0) return COR_EINVAL;
  const long total = (long)H * OW;
  if (C == 3) hipLaunchKernelGGL(resample_h_kernel<3>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, bounds, kk, ksize, H, W, OW);
  else if (C == 1) hipLaunchKernelGGL(resample_h_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, bounds, kk, ksize, H, W, OW);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_resample_cols_u8(const unsigned char* in, float* out_f32, unsigned char* out_u8, const int* bounds, const int* kk, int ksize,
                                    int H, int W, int C, int OH, const float* mean, const float* stdv, void* stream) {
  if (!in || (!out_f32 && !out_u8) || !bounds || !kk || ksize <= 0 || H <= 0 || W <= 0 || OH <= 0 || ((mean == nullptr) != (stdv == nullptr)))
    return COR_EINVAL;
  const long total = (long)OH * W;
  if (C == 3) hipLaunchKernelGGL(resample_v_kernel<3>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out_f32, out_u8, bounds, kk, ksize, H, W, OH, mean, stdv);
  else if (C == 1) hipLaunchKernelGGL(resample_v_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out_f32, out_u8, bounds, kk, ksize, H, W, OH, mean, stdv);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}
