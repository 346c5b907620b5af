// cor_amd — small HBM/latency-bound kernels of the support branch, prompt encoder and mask decoder (gfx950).
// None of these is GEMM-shaped enough for MFMA (tiny channel counts or pure gathers); they are written for
// coalesced channels-last access, wave-shuffle reductions and LDS-broadcast weights.
#include "common.h"

namespace {

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

// ---------------------------------------------------------------- bilinear (align_corners=False, no antialias)
__device__ __forceinline__ void bil_taps(int o, int n_in, int n_out, int& i0, int& i1, float& w1) {
  float s = ((float)o + 0.5f) * ((float)n_in / (float)n_out) - 0.5f;
  s = fmaxf(s, 0.0f);
  i0 = min((int)floorf(s), n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  w1 = s - (float)i0;
}

__global__ void __launch_bounds__(256) bilinear_kernel(const float* x, float* out, int planes, int H, int W, int OH, int OW, int clamp01) {
  const long total = (long)planes * OH * OW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % OW); long t = i / OW; const int oy = (int)(t % OH); const long pl = t / OH;
    int y0, y1, x0, x1; float wy, wx;
    bil_taps(oy, H, OH, y0, y1, wy); bil_taps(ox, W, OW, x0, x1, wx);
    const float* p = x + pl * H * W;
    const float top = p[y0 * W + x0] * (1.f - wx) + p[y0 * W + x1] * wx;
    const float bot = p[y1 * W + x0] * (1.f - wx) + p[y1 * W + x1] * wx;
    float v = top * (1.f - wy) + bot * wy;
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = v;
  }
}

// ---------------------------------------------------------------- 3x3 stride-2 pad-1 conv, tiny channels
__global__ void __launch_bounds__(256) conv3x3s2_kernel(const float* x, int cl, const float* w, const float* bias, float* out,
                                                        int B, int Cin, int Cout, int H, int W, int OH, int OW) {
  const long total = (long)B * OH * OW * Cout;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int co = (int)(i % Cout); long t = i / Cout;
    const int ox = (int)(t % OW); t /= OW; const int oy = (int)(t % OH); const int b = (int)(t / OH);
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int y = 2 * oy + ky - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = 2 * ox + kx - 1;
          if (xx < 0 || xx >= W) continue;
          const float v = cl ? x[(((long)b * H + y) * W + xx) * Cin + ci] : x[(((long)b * Cin + ci) * H + y) * W + xx];
          acc = fmaf(v, w[((co * Cin + ci) * 3 + ky) * 3 + kx], acc);
        }
      }
    out[i] = acc;
  }
}

// ---------------------------------------------------------------- depthwise 7x7 pad 3, channels-last; w_t [49, C]
template <typename TO>
__global__ void __launch_bounds__(256) dwconv7_kernel(const float* x, const float* w_t, const float* bias, TO* out, int B, int H, int W, int C) {
  const long total = (long)B * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C); long t = i / C;
    const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int b = (int)(t / H);
    float acc = bias[c];
    for (int ky = 0; ky < 7; ++ky) {
      const int sy = y + ky - 3;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int sx = xx + kx - 3;
        if (sx < 0 || sx >= W) continue;
        acc = fmaf(x[(((long)b * H + sy) * W + sx) * C + c], w_t[(ky * 7 + kx) * C + c], acc);
      }
    }
    st<TO>(out + i, acc);
  }
}

// four channels per thread (16-B loads of x and w): the scalar form issued 98 4-byte loads per output and ran at the L1 rate
template <typename TO>
__global__ void __launch_bounds__(256) dwconv7_vec4_kernel(const float* x, const float* w_t, const float* bias, TO* out, int B, int H, int W, int C) {
  const int c4n = C >> 2;
  const long total = (long)B * H * W * c4n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4; long t = i / c4n;
    const int xx = (int)(t % W); t /= W; const int y = (int)(t % H); const int b = (int)(t / H);
    f32x4 acc = *(const f32x4*)(bias + c);
    for (int ky = 0; ky < 7; ++ky) {
      const int sy = y + ky - 3;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int sx = xx + kx - 3;
        if (sx < 0 || sx >= W) continue;
        const f32x4 xv = *(const f32x4*)(x + (((long)b * H + sy) * W + sx) * C + c), wv = *(const f32x4*)(w_t + (ky * 7 + kx) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);       // same order per channel as the scalar form
      }
    }
    st4<TO>(out + (((long)b * H + y) * W + xx) * C + c, acc);
  }
}

// ---------------------------------------------------------------- mask-adapter pooling
// grid (ceil(D/256), B); every block recomputes the P x M softmax weights (P*M <= 8K values) in LDS.
__global__ void __launch_bounds__(256) adapter_pool_kernel(const float* maps, const float* feat, float* out, int P, int M, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // ls[P*M] | wgt[P] | red[2*M]
  float* ls = sm; float* wgt = sm + P * M; float* red = wgt + P;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* mp = maps + (long)b * P * M;
  for (int i = tid; i < P * M; i += 256) {
    const float v = mp[i];
    ls[i] = fminf(v, 0.f) - log1pf(expf(-fabsf(v)));          // logsigmoid
  }
  __syncthreads();
  for (int m = wave; m < M; m += 4) {                          // one wave reduces one map
    float mx = -INFINITY;
    for (int p = lane; p < P; p += 64) mx = fmaxf(mx, ls[p * M + m]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int p = lane; p < P; p += 64) s += expf(ls[p * M + m] - mx);
    s = wave_sum(s);
    if (lane == 0) { red[2 * m] = mx; red[2 * m + 1] = 1.0f / s; }
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    float a = 0.f;
    for (int m = 0; m < M; ++m) a += expf(ls[p * M + m] - red[2 * m]) * red[2 * m + 1];
    wgt[p] = a / (float)M;
  }
  __syncthreads();
  const int d = blockIdx.x * 256 + tid;
  if (d >= D) return;
  const float* fp = feat + (long)b * P * D + d;
  float acc = 0.f;
  for (int p = 0; p < P; ++p) acc = fmaf(wgt[p], fp[(long)p * D], acc);
  out[(long)b * D + d] = acc;
}

// ---------------------------------------------------------------- masked average pooling (+clamp, +L2 norm); one block per sample
__global__ void __launch_bounds__(256) masked_pool_kernel(const float* feat, int nchw, const float* mask, float* out, int P, int D,
                                                          int clamp01, int l2norm) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // mk[P] | res[D] | red[8]
  float* mk = sm; float* res = sm + P; float* red = res + D;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ms = 0.f;
  for (int p = tid; p < P; p += 256) {
    float v = mask[(long)b * P + p];
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    mk[p] = v; ms += v;
  }
  ms = wave_sum(ms);
  if (lane == 0) red[wave] = ms;
  __syncthreads();
  const float denom = red[0] + red[1] + red[2] + red[3] + 1e-8f;
  if (nchw) {                                                  // feat [B,D,P]: a wave per channel, p contiguous
    for (int d = wave; d < D; d += 4) {
      const float* fp = feat + ((long)b * D + d) * P;
      float acc = 0.f;
      for (int p = lane; p < P; p += 64) acc = fmaf(fp[p], mk[p], acc);
      acc = wave_sum(acc);
      if (lane == 0) res[d] = acc / denom;
    }
  } else {                                                     // feat [B,P,D]: a thread per channel, d contiguous
    for (int d = tid; d < D; d += 256) {
      const float* fp = feat + (long)b * P * D + d;
      float acc = 0.f;
      for (int p = 0; p < P; ++p) acc = fmaf(fp[(long)p * D], mk[p], acc);
      res[d] = acc / denom;
    }
  }
  __syncthreads();
  float inv = 1.f;
  if (l2norm) {
    float q = 0.f;
    for (int d = tid; d < D; d += 256) q += res[d] * res[d];
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    inv = 1.0f / fmaxf(sqrtf(red[4] + red[5] + red[6] + red[7]), 1e-12f);
  }
  for (int d = tid; d < D; d += 256) out[(long)b * D + d] = res[d] * inv;
}

// ---------------------------------------------------------------- gated fusion pieces
__global__ void __launch_bounds__(256) fuse_gate_kernel(const float* img, const float* txt, const float* aI, const float* aT, float* cat, int N, int D) {
  const long total = (long)N * D;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / D; const int d = (int)(i - n * D);
    cat[n * 2 * D + d] = aI[i] * img[i];
    cat[n * 2 * D + D + d] = aT[i] * txt[i];
  }
}

__global__ void __launch_bounds__(256) fuse_mix_kernel(const float* cat, const float* dyn, float* out, int N, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float a = dyn[row];
  const float* c = cat + (long)row * 2 * D;
  float q = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = a * c[d] + (1.f - a) * c[D + d]; q += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  for (int d = lane; d < D; d += 64) out[(long)row * D + d] = (a * c[d] + (1.f - a) * c[D + d]) * inv;
}

// ---------------------------------------------------------------- dense random-Fourier PE tokens [size*size, 2F]
__global__ void __launch_bounds__(256) dense_pe_kernel(const float* G, float* out, int size, int F) {
  const long total = (long)size * size * F;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int f = (int)(i % F); const int t = (int)(i / F); const int y = t / size, x = t - y * size;
    const float cx = 2.f * (((float)x + 0.5f) / (float)size) - 1.f, cy = 2.f * (((float)y + 0.5f) / (float)size) - 1.f;
    const float ang = 6.283185307179586f * (cx * G[f] + cy * G[F + f]);
    out[(long)t * 2 * F + f] = sinf(ang);
    out[(long)t * 2 * F + F + f] = cosf(ang);
  }
}

// ---------------------------------------------------------------- ConvT(2x2,s2) pixel shuffle (+bias) + LN over channels + act
// y [B*H*W, 4*C] with col = (dy*2+dx)*C + co  ->  out [B,2H,2W,C]; one wave per output pixel, lane = channel (+64 j).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) upscale_shuffle_kernel(const TI* y, const float* bias, const float* ln_w, const float* ln_b,
                                                              float eps, int act, TO* out, int B, int H, int W, int C) {
  const int lane = threadIdx.x & 63;
  const long pix = blockIdx.x * 4L + (threadIdx.x >> 6);
  const long npix = (long)B * 4 * H * W;
  if (pix >= npix) return;
  const int OW = 2 * W, OH = 2 * H;
  const int ox = (int)(pix % OW); long t = pix / OW; const int oy = (int)(t % OH); const int b = (int)(t / OH);
  const long tok = ((long)b * H + (oy >> 1)) * W + (ox >> 1);
  const TI* src = y + tok * 4L * C + (long)(((oy & 1) * 2 + (ox & 1))) * C;
  float v[4]; float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    v[j] = 0.f;
    if (c < C) { v[j] = ld<TI>(src + c) + (bias ? bias[c] : 0.f); s += v[j]; }
  }
  if (ln_w) {
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < C) { const float d = v[j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; if (c < C) v[j] = (v[j] - mean) * rstd * ln_w[c] + ln_b[c]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; if (c < C) st<TO>(out + pix * C + c, apply_act(v[j], act)); }
}

// C == 64 (the SAM decoder: transformer_dim / 4): four output pixels per wave, 16 lanes x 4 channels per pixel, 8 / 16-byte
// accesses instead of one 2-byte element per lane (0.8 TB/s); the LayerNorm sums run over the 16 lanes of a pixel.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) upscale_shuffle64_kernel(const TI* y, const float* bias, const float* ln_w, const float* ln_b,
                                                                float eps, int act, TO* out, int B, int H, int W) {
  constexpr int C = 64;
  const int lane = threadIdx.x & 63, sub = lane >> 4, c = (lane & 15) * 4;
  const long pix = (blockIdx.x * 4L + (threadIdx.x >> 6)) * 4 + sub;
  const long npix = (long)B * 4 * H * W;
  const bool ok = pix < npix;
  const long pc = ok ? pix : npix - 1;               // clamped: all 64 lanes take part in the shuffles
  const int OW = 2 * W, OH = 2 * H;
  const int ox = (int)(pc % OW); long t = pc / OW; const int oy = (int)(t % OH); const int b = (int)(t / OH);
  const long tok = ((long)b * H + (oy >> 1)) * W + (ox >> 1);
  const TI* src = y + tok * 4L * C + (long)(((oy & 1) * 2 + (ox & 1))) * C + c;
  f32x4 v = ld4<TI>(src);
  if (bias) v += *(const f32x4*)(bias + c);
  if (ln_w) {
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / C + eps);
    const f32x4 wv = *(const f32x4*)(ln_w + c), bv = *(const f32x4*)(ln_b + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * wv[e] + bv[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
  if (ok) st4<TO>(out + pix * C + c, v);
}

// ---------------------------------------------------------------- fused second ConvT + GELU + hypernetwork dot
// thread = one INPUT pixel (its Cin values in registers); weights [Cin,Cout,2,2] and hyper rows in LDS (broadcast reads).
template <typename T, int CIN, int COUT>
__global__ void __launch_bounds__(256) upscale_hyper_kernel(const T* x, const float* w, const float* bias, const float* hyper,
                                                            long hyper_bs, float* masks, int H, int W, int Kmask) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // w_s[4][COUT][CIN] | b_s[COUT] | h_s[Kmask*COUT]
  float* w_s = sm; float* b_s = sm + 4 * COUT * CIN; float* h_s = b_s + COUT;
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < CIN * COUT * 4; i += 256) {            // source index i = (ci*COUT + co)*4 + dd
    const int dd = i & 3, co = (i >> 2) % COUT, ci = (i >> 2) / COUT;
    w_s[(dd * COUT + co) * CIN + ci] = w[i];
  }
  for (int i = tid; i < COUT; i += 256) b_s[i] = bias[i];
  for (int i = tid; i < Kmask * COUT; i += 256) h_s[i] = hyper[(long)b * hyper_bs + i];
  __syncthreads();
  const int pix = blockIdx.x * 256 + tid;
  if (pix >= H * W) return;
  const int yy = pix / W, xx = pix - yy * W;
  float xi[CIN];
  const T* xp = x + ((long)b * H * W + pix) * CIN;
#pragma unroll
  for (int i = 0; i < CIN / 4; ++i) { const f32x4 v = ld4<T>(xp + 4 * i); xi[4 * i] = v[0]; xi[4 * i + 1] = v[1]; xi[4 * i + 2] = v[2]; xi[4 * i + 3] = v[3]; }
  for (int k0 = 0; k0 < Kmask; k0 += 4) {
    const int nk = min(4, Kmask - k0);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      float m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};       // dx = 0 / 1
      for (int co = 0; co < COUT; ++co) {
        const float* w0 = w_s + ((dy * 2 + 0) * COUT + co) * CIN;
        const float* w1 = w_s + ((dy * 2 + 1) * COUT + co) * CIN;
        float a0 = b_s[co], a1 = b_s[co];
#pragma unroll
        for (int ci = 0; ci < CIN; ci += 4) {
          const f32x4 wa = *(const f32x4*)(w0 + ci), wb = *(const f32x4*)(w1 + ci);
          a0 = fmaf(xi[ci], wa[0], a0); a0 = fmaf(xi[ci + 1], wa[1], a0); a0 = fmaf(xi[ci + 2], wa[2], a0); a0 = fmaf(xi[ci + 3], wa[3], a0);
          a1 = fmaf(xi[ci], wb[0], a1); a1 = fmaf(xi[ci + 1], wb[1], a1); a1 = fmaf(xi[ci + 2], wb[2], a1); a1 = fmaf(xi[ci + 3], wb[3], a1);
        }
        a0 = gelu_erf_f(a0); a1 = gelu_erf_f(a1);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nk) { const float hv = h_s[(k0 + k) * COUT + co]; m0[k] = fmaf(hv, a0, m0[k]); m1[k] = fmaf(hv, a1, m1[k]); }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < nk) {
        float* dst = masks + (((long)b * Kmask + k0 + k) * 2 * H + 2 * yy + dy) * 2L * W + 2 * xx;
        *(float2*)dst = make_float2(m0[k], m1[k]);
      }
    }
  }
}

// best[b] = argmax over iou[b, k_off : k_off+Ksel] (first maximum wins); hyper_sel[b,:] = hyper[b, k_off+best, :]
__global__ void iou_select_kernel(const float* iou, const float* hyper, int B, int Kall, int k_off, int Ksel, int C,
                                  long long* best, float* hyper_sel) {
  const int b = blockIdx.x;
  int bi = 0; float bv = iou[(long)b * Kall + k_off];
  for (int k = 1; k < Ksel; ++k) { const float v = iou[(long)b * Kall + k_off + k]; if (v > bv) { bv = v; bi = k; } }
  if (threadIdx.x == 0 && best) best[b] = bi;
  for (int c = threadIdx.x; c < C; c += blockDim.x) hyper_sel[(long)b * C + c] = hyper[((long)b * Kall + k_off + bi) * C + c];
}

// The five output MLPs of the mask decoder (four hyper-network MLPs on mask tokens 1..4, the IoU head on token 0; each 256 -> 256 -> 256 ->
// 32 | 4 with ReLU) in ONE launch: as 15 GEMM launches on 6 tokens per image they were 143 us of pure launch latency on the step's critical
// path. Block = (group of NS samples, MLP m); thread n owns output feature n of the two hidden layers: it streams row n of the weight
// (16-byte loads; the 128 KB of a layer come out of L2) against the NS activation vectors in LDS (broadcast reads) - fp32 FMA chains over
// k = 0..255 in order, so a sample's result does not depend on the batch it is in. Hidden activations are rounded to the operand type
// like the GEMM outputs they replace.
constexpr int DH_NS = 8;
template <typename TW> __device__ __forceinline__ void dh_row8(const TW* w, float (&f)[8]);
template <> __device__ __forceinline__ void dh_row8<bf16_t>(const bf16_t* w, float (&f)[8]) {
  const uint4 u = *(const uint4*)w;
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u); f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u); f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void dh_row8<float>(const float* w, float (&f)[8]) {
  const f32x4 a = *(const f32x4*)w, b = *(const f32x4*)(w + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
template <typename TW>
__device__ __forceinline__ void dh_dot(const TW* wrow, const float (*x)[256], float (&acc)[DH_NS]) {
#pragma unroll 4
  for (int k = 0; k < 256; k += 8) {
    float w[8];
    dh_row8<TW>(wrow + k, w);
#pragma unroll
    for (int s = 0; s < DH_NS; ++s) {
      const f32x4 a = *(const f32x4*)&x[s][k], b = *(const f32x4*)&x[s][k + 4];
      acc[s] = fmaf(w[0], a[0], acc[s]); acc[s] = fmaf(w[1], a[1], acc[s]); acc[s] = fmaf(w[2], a[2], acc[s]); acc[s] = fmaf(w[3], a[3], acc[s]);
      acc[s] = fmaf(w[4], b[0], acc[s]); acc[s] = fmaf(w[5], b[1], acc[s]); acc[s] = fmaf(w[6], b[2], acc[s]); acc[s] = fmaf(w[7], b[3], acc[s]);
    }
  }
}
template <typename TW>
__global__ void __launch_bounds__(256) decoder_heads_kernel(const TW* hs, const TW* w01, const float* b01, const TW* w2, const float* b2,
                                                            float* hyper, float* iou, int B) {
  __shared__ __attribute__((aligned(16))) float xa[DH_NS][256], xb[DH_NS][256];
  const int tid = threadIdx.x, m = blockIdx.y, s0 = blockIdx.x * DH_NS;
  const int tok = m < 4 ? 1 + m : 0;
#pragma unroll
  for (int s = 0; s < DH_NS; ++s) xa[s][tid] = s0 + s < B ? ld<TW>(hs + ((long)(s0 + s) * 6 + tok) * 256 + tid) : 0.f;
  __syncthreads();
#pragma unroll 1
  for (int layer = 0; layer < 2; ++layer) {
    const float (*xi)[256] = layer == 0 ? xa : xb;
    float (*xo)[256] = layer == 0 ? xb : xa;
    float acc[DH_NS];
    const float bias = b01[(m * 2 + layer) * 256 + tid];
#pragma unroll
    for (int s = 0; s < DH_NS; ++s) acc[s] = bias;
    dh_dot<TW>(w01 + ((long)(m * 2 + layer) * 256 + tid) * 256, xi, acc);
#pragma unroll
    for (int s = 0; s < DH_NS; ++s) {
      float v = fmaxf(acc[s], 0.f);
      if (sizeof(TW) == 2) v = bf2f(f2bf(v));
      xo[s][tid] = v;
    }
    __syncthreads();
  }
  const int nout = m < 4 ? 32 : 4;
  if (tid < nout) {
    const int row = (m < 4 ? 32 * m : 128) + tid;
    float acc[DH_NS];
#pragma unroll
    for (int s = 0; s < DH_NS; ++s) acc[s] = b2[row];
    dh_dot<TW>(w2 + (long)row * 256, xa, acc);
#pragma unroll
    for (int s = 0; s < DH_NS; ++s)
      if (s0 + s < B) {
        if (m < 4) hyper[(long)(s0 + s) * 128 + row] = acc[s];
        else iou[(long)(s0 + s) * 4 + tid] = acc[s];
      }
  }
}

}  // namespace

extern "C" int cor_decoder_heads(const void* hs, const void* w01, const float* b01, const void* w2, const float* b2, int dtype, float* hyper,
                                 float* iou, int B, void* stream) {
  if (!hs || !w01 || !b01 || !w2 || !b2 || !hyper || !iou || B <= 0) return COR_EINVAL;
  if ((((uintptr_t)hs | (uintptr_t)w01 | (uintptr_t)w2) & 15) != 0) return COR_EINVAL;
  const dim3 grid(cdiv(B, DH_NS), 5);
  if (dtype == COR_BF16)
    hipLaunchKernelGGL(decoder_heads_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hs, (const bf16_t*)w01, b01, (const bf16_t*)w2, b2, hyper, iou, B);
  else if (dtype == COR_F32)
    hipLaunchKernelGGL(decoder_heads_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)hs, (const float*)w01, b01, (const float*)w2, b2, hyper, iou, B);
  else
    return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_bilinear(const float* x, float* out, int planes, int H, int W, int OH, int OW, int clamp01, void* stream) {
  if (!x || !out || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((long)planes * OH * OW)), dim3(256), 0, (hipStream_t)stream, x, out, planes, H, W, OH, OW, clamp01);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_conv3x3s2_small(const float* x, int x_channels_last, const float* w, const float* bias, float* out, int B, int Cin,
                                   int Cout, int H, int W, void* stream) {
  if (!x || !w || !out || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return COR_EINVAL;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(conv3x3s2_kernel, dim3(grid_for((long)B * OH * OW * Cout)), dim3(256), 0, (hipStream_t)stream, x, x_channels_last, w, bias, out, B, Cin, Cout, H, W, OH, OW);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_dwconv7x7(const float* x, const float* w_t, const float* bias, void* out, int out_dtype, int B, int H, int W, int C, void* stream) {
  if (!x || !w_t || !bias || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return COR_EINVAL;
  if ((C & 3) == 0 && (((uintptr_t)x | (uintptr_t)w_t | (uintptr_t)bias | (uintptr_t)out) & 15) == 0) {
    const dim3 g4(grid_for((long)B * H * W * (C >> 2)));
    if (out_dtype == COR_F32) hipLaunchKernelGGL((dwconv7_vec4_kernel<float>), g4, dim3(256), 0, (hipStream_t)stream, x, w_t, bias, (float*)out, B, H, W, C);
    else if (out_dtype == COR_BF16) hipLaunchKernelGGL((dwconv7_vec4_kernel<bf16_t>), g4, dim3(256), 0, (hipStream_t)stream, x, w_t, bias, (bf16_t*)out, B, H, W, C);
    else return COR_ENOSUPPORT;
    COR_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(grid_for((long)B * H * W * C));
  if (out_dtype == COR_F32) hipLaunchKernelGGL((dwconv7_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x, w_t, bias, (float*)out, B, H, W, C);
  else if (out_dtype == COR_BF16) hipLaunchKernelGGL((dwconv7_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, w_t, bias, (bf16_t*)out, B, H, W, C);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_adapter_pool(const float* maps, const float* feat, float* out, int B, int P, int M, int D, void* stream) {
  if (!maps || !feat || !out || B <= 0 || P <= 0 || M <= 0 || D <= 0) return COR_EINVAL;
  const size_t lds = ((size_t)P * M + P + 2 * M) * sizeof(float);
  if (lds > 64 * 1024) return COR_ENOSUPPORT;
  hipLaunchKernelGGL(adapter_pool_kernel, dim3(cdiv(D, 256), B), dim3(256), lds, (hipStream_t)stream, maps, feat, out, P, M, D);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_masked_pool(const float* feat, int feat_nchw, const float* mask, float* out, int B, int P, int D, int clamp01,
                               int l2norm, void* stream) {
  if (!feat || !mask || !out || B <= 0 || P <= 0 || D <= 0) return COR_EINVAL;
  const size_t lds = ((size_t)P + D + 8) * sizeof(float);
  if (lds > 64 * 1024) return COR_ENOSUPPORT;
  hipLaunchKernelGGL(masked_pool_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, feat, feat_nchw, mask, out, P, D, clamp01, l2norm);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_fuse_gate(const float* img, const float* txt, const float* aI, const float* aT, float* cat, int N, int D, void* stream) {
  if (!img || !txt || !aI || !aT || !cat || N <= 0 || D <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(fuse_gate_kernel, dim3(grid_for((long)N * D)), dim3(256), 0, (hipStream_t)stream, img, txt, aI, aT, cat, N, D);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_fuse_mix(const float* cat, const float* dyn, float* out, int N, int D, void* stream) {
  if (!cat || !dyn || !out || N <= 0 || D <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(fuse_mix_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, cat, dyn, out, N, D);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_dense_pe(const float* gauss, float* out, int size, int F, void* stream) {
  if (!gauss || !out || size <= 0 || F <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(dense_pe_kernel, dim3(grid_for((long)size * size * F)), dim3(256), 0, (hipStream_t)stream, gauss, out, size, F);
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_upscale_shuffle(const void* y, int y_dtype, const float* bias, const float* ln_w, const float* ln_b, float eps, int act,
                                   void* out, int out_dtype, int B, int H, int W, int Cout, void* stream) {
  if (!y || !out || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 256) return COR_EINVAL;
  if ((ln_w == nullptr) != (ln_b == nullptr)) return COR_EINVAL;
  const bool c64 = Cout == 64 && (((uintptr_t)y | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0;
  const dim3 grid(c64 ? cdiv((long)B * 4 * H * W, 16) : cdiv((long)B * 4 * H * W, 4));
#define CALL(TI, TO)                                                                                                                     \
  if (c64) hipLaunchKernelGGL((upscale_shuffle64_kernel<TI, TO>), grid, dim3(256), 0, (hipStream_t)stream, (const TI*)y, bias, ln_w, ln_b, eps, act, (TO*)out, B, H, W); \
  else hipLaunchKernelGGL((upscale_shuffle_kernel<TI, TO>), grid, dim3(256), 0, (hipStream_t)stream, (const TI*)y, bias, ln_w, ln_b, eps, act, (TO*)out, B, H, W, Cout)
  if (y_dtype == COR_F32 && out_dtype == COR_F32) { CALL(float, float); }
  else if (y_dtype == COR_BF16 && out_dtype == COR_BF16) { CALL(bf16_t, bf16_t); }
  else if (y_dtype == COR_BF16 && out_dtype == COR_F32) { CALL(bf16_t, float); }
  else if (y_dtype == COR_F32 && out_dtype == COR_BF16) { CALL(float, bf16_t); }
  else return COR_ENOSUPPORT;
#undef CALL
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_upscale_hyper(const void* x, int dtype, const float* w, const float* bias, const float* hyper, long hyper_bs,
                                 float* masks, int B, int H, int W, int Cin, int Cout, int Kmask, void* stream) {
  if (!x || !w || !bias || !hyper || !masks || B <= 0 || H <= 0 || W <= 0 || Kmask <= 0 || Kmask > 16) return COR_EINVAL;
  if (Cin != 64 || Cout != 32) return COR_ENOSUPPORT;   // SAM decoder: transformer_dim/4 -> /8 (mask_decoder.py:58)
  const size_t lds = (4 * 32 * 64 + 32 + (size_t)Kmask * 32) * sizeof(float);
  const dim3 grid(cdiv(H * W, 256), B);
  if (dtype == COR_F32) hipLaunchKernelGGL((upscale_hyper_kernel<float, 64, 32>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)x, w, bias, hyper, hyper_bs, masks, H, W, Kmask);
  else if (dtype == COR_BF16) hipLaunchKernelGGL((upscale_hyper_kernel<bf16_t, 64, 32>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, w, bias, hyper, hyper_bs, masks, H, W, Kmask);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_iou_select(const float* iou, const float* hyper, int B, int Kall, int k_off, int Ksel, int C, long long* best,
                              float* hyper_sel, void* stream) {
  if (!iou || !hyper || !hyper_sel || B <= 0 || Kall <= 0 || k_off < 0 || Ksel <= 0 || k_off + Ksel > Kall || C <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(iou_select_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, iou, hyper, B, Kall, k_off, Ksel, C, best, hyper_sel);
  COR_CHECK_LAUNCH();
  return 0;
}
