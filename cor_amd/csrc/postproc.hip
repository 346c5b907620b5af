// cor_amd — mask post-processing and segmentation metrics of the inference harness (gfx950). HBM-bound reductions:
// one block per sample, wave-shuffle + LDS block reductions, coalesced row access.
//   ref: utils/vailder.py:426-430 (sigmoid, per-sample min-max), :459-473 (cv2.resize INTER_LINEAR to the GT size,
//        > 0.5, uint8 * 255); utils/trainer_v3_g.py:381-443 (compute_dice / mae / iou / mdice / miou).
#include "common.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// out = (sigmoid(x) - min) / (max - min + 1e-8) per sample
__global__ void __launch_bounds__(256) prob_minmax_kernel(const float* logits, float* out, int HW) {
  __shared__ float red[4];
  const float* x = logits + (long)blockIdx.x * HW;
  float* o = out + (long)blockIdx.x * HW;
  float mx = -INFINITY, mn = INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) { const float p = 1.0f / (1.0f + expf(-x[i])); mx = fmaxf(mx, p); mn = fminf(mn, p); }
  mx = block_reduce(mx, red, true);
  mn = -block_reduce(-mn, red, true);
  const float inv = 1.0f / (mx - mn + 1e-8f);
  for (int i = threadIdx.x; i < HW; i += 256) o[i] = (1.0f / (1.0f + expf(-x[i])) - mn) * inv;
}

// bilinear (half-pixel centres, edge clamp: cv2.INTER_LINEAR == align_corners=False without antialias) + threshold
__global__ void __launch_bounds__(256) resize_binarize_kernel(const float* p, unsigned char* out, int H, int W, int OH, int OW, float thr) {
  const float* src = p + (long)blockIdx.y * H * W;
  unsigned char* dst = out + (long)blockIdx.y * OH * OW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)OH * OW; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % OW), oy = (int)(i / OW);
    float sy = fmaxf(((float)oy + 0.5f) * ((float)H / (float)OH) - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * ((float)W / (float)OW) - 0.5f, 0.f);
    const int y0 = min((int)floorf(sy), H - 1), x0 = min((int)floorf(sx), W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
    const float top = src[y0 * W + x0] * (1.f - wx) + src[y0 * W + x1] * wx;
    const float bot = src[y1 * W + x0] * (1.f - wx) + src[y1 * W + x1] * wx;
    const float v = top * (1.f - wy) + bot * wy;
    // thr >= 0: hard mask (> thr ? 255 : 0); thr < 0: soft mask, (v * 255).astype(uint8) = truncation (utils/vailder.py:621)
    dst[i] = thr >= 0.f ? (v > thr ? 255 : 0) : (unsigned char)fminf(fmaxf(v * 255.f, 0.f), 255.f);
  }
}

// per sample: [dice, mae, iou, mdice, miou] of a soft prediction against the ground truth (both in [0,1])
__global__ void __launch_bounds__(256) mask_metrics_kernel(const float* pred, const float* gt, float* out, int HW, float smooth) {
  __shared__ float red[4];
  const float* p = pred + (long)blockIdx.x * HW; const float* g = gt + (long)blockIdx.x * HW;
  float pg = 0.f, ps = 0.f, gs = 0.f, ad = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) { const float a = p[i], b = g[i]; pg += a * b; ps += a; gs += b; ad += fabsf(a - b); }
  pg = block_reduce(pg, red, false); ps = block_reduce(ps, red, false); gs = block_reduce(gs, red, false); ad = block_reduce(ad, red, false);
  if (threadIdx.x == 0) {
    const float n = (float)HW;
    const float dice = (2.f * pg + smooth) / (ps + gs + smooth);
    const float iou = (pg + smooth) / (ps + gs - pg + smooth);
    // background: (1-p)(1-g) summed = n - ps - gs + pg ; sums n - ps, n - gs
    const float bpg = n - ps - gs + pg, bps = n - ps, bgs = n - gs;
    const float bdice = (2.f * bpg + smooth) / (bps + bgs + smooth);
    const float biou = (bpg + smooth) / (bps + bgs - bpg + smooth);
    float* o = out + blockIdx.x * 5;
    o[0] = dice; o[1] = ad / n; o[2] = iou; o[3] = 0.5f * (dice + bdice); o[4] = 0.5f * (iou + biou);
  }
}

}  // namespace

extern "C" int cor_mask_prob_minmax(const float* logits, float* out, int B, int HW, void* stream) {
  if (!logits || !out || B <= 0 || HW <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(prob_minmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, out, HW);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_resize_binarize(const float* prob, unsigned char* out, int B, int H, int W, int OH, int OW, float threshold, void* stream) {
  if (!prob || !out || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return COR_EINVAL;
  long blocks = ((long)OH * OW + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(resize_binarize_kernel, dim3((int)blocks, B), dim3(256), 0, (hipStream_t)stream, prob, out, H, W, OH, OW, threshold);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_resize_gray(const float* prob, unsigned char* out, int B, int H, int W, int OH, int OW, void* stream) {
  return cor_resize_binarize(prob, out, B, H, W, OH, OW, -1.0f, stream);
}

extern "C" int cor_mask_metrics(const float* pred, const float* gt, float* out, int B, int HW, float smooth, void* stream) {
  if (!pred || !gt || !out || B <= 0 || HW <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(mask_metrics_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pred, gt, out, HW, smooth);
  COR_CHECK_LAUNCH();
  return 0;
}
