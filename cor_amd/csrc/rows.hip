// cor_amd — HBM-bound row kernels for gfx950: LayerNorm, L2-normalise, elementwise add, casts/copies,
// patch gather (im2col), layout changes, token embedding. One wave per row where a row reduction is needed
// (shuffle reduction over 64 lanes, no LDS), 4-element vector accesses on the contiguous axis.
#include "common.h"

namespace {

// ---------------------------------------------------------------- LayerNorm: one wave per row, row cached in registers
template <typename TI, typename TO, int MAXV>   // MAXV = ceil(C / 256) vectors of 4 per lane
__global__ void __launch_bounds__(256) layernorm_kernel(const TI* x, TO* y, const float* w, const float* b, int rows, int C,
                                                        float eps, int act, int rev) {
  const int lane = threadIdx.x & 63;
  const int row = (rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (long)row * C;
  const int nv = C >> 2;          // C % 4 == 0 (checked by the launcher)
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) { v[i] = ld4<TI>(xr + 4 * j); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / C + eps);
  TO* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) {
      const f32x4 wv = *(const f32x4*)(w + 4 * j), bv = *(const f32x4*)(b + 4 * j);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = apply_act((v[i][e] - mean) * rstd * wv[e] + bv[e], act);
      st4<TO>(yr + 4 * j, o);
    }
  }
}

// Half a wave per row (two rows per wave, eight per block): twice the rows in flight per wave, 5-step reductions; NT = non-temporal loads
template <typename TI, typename TO, int NV, bool NT>   // NV vectors of 4 per lane: C = 128 * NV
__global__ void __launch_bounds__(256) layernorm_half_kernel(const TI* x, TO* y, const float* w, const float* b, int rows, int C,
                                                             float eps, int act, int rev) {
  const int lane = threadIdx.x & 31;
  const int row = (rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 8 + (threadIdx.x >> 5);
  if (row >= rows) return;
  const TI* xr = x + (long)row * C;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if constexpr (NT && sizeof(TI) == 4) v[i] = __builtin_nontemporal_load((const f32x4*)(xr + 4 * (lane + 32 * i)));
    else v[i] = ld4<TI>(xr + 4 * (lane + 32 * i));
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  auto hsum = [](float t) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  const float mean = hsum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(hsum(q) / C + eps);
  TO* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 32 * i;
    const f32x4 wv = *(const f32x4*)(w + 4 * j), bv = *(const f32x4*)(b + 4 * j);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = apply_act((v[i][e] - mean) * rstd * wv[e] + bv[e], act);
    st4<TO>(yr + 4 * j, o);
  }
}

#ifdef COR_PROBES
// Probe ("bf16 delta"): the residual add moved out of the proj GEMM's epilogue into the LayerNorm pass.
// x (fp32, in place) += delta (bf16); y = LN(x). Same half-wave-per-row form as layernorm_half_kernel.
template <int NV, bool WRITE_X>
__global__ void __launch_bounds__(256) layernorm_delta_probe(float* x, const bf16_t* dl, bf16_t* y, const float* w, const float* b, int rows, int C, float eps) {
  const int lane = threadIdx.x & 31;
  const int row = ((int)gridDim.x - 1 - (int)blockIdx.x) * 8 + (threadIdx.x >> 5);
  if (row >= rows) return;
  float* xr = x + (long)row * C;
  const bf16_t* dr = dl + (long)row * C;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = __builtin_nontemporal_load((const f32x4*)(xr + 4 * (lane + 32 * i)));
    const f32x4 d = ld4<bf16_t>(dr + 4 * (lane + 32 * i));
    v[i] += d;
    if (WRITE_X) *(f32x4*)(xr + 4 * (lane + 32 * i)) = v[i];
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  auto hsum = [](float t) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  const float mean = hsum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(hsum(q) / C + eps);
  bf16_t* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 32 * i;
    const f32x4 wv = *(const f32x4*)(w + 4 * j), bv = *(const f32x4*)(b + 4 * j);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
    st4<bf16_t>(yr + 4 * j, o);
  }
}
#endif

template <typename TI, typename TO>
int launch_ln(const void* x, void* y, const float* w, const float* b, int rows, int C, float eps, int act, int rev, hipStream_t s) {
  // wide fp32 rows (the residual streams: C = 768 / 1024 / 1152 / 1280): half a wave per row, non-temporal loads (x is read once
  // here and next by a GEMM epilogue a millisecond later): 131072 x 768 -> bf16 in 98-110 us against 113-121 us for the
  // wave-per-row form (5.3 -> 6.1 TB/s stand-alone; +0.1 ... 0.9 % on the bench step, where the reverse work order already
  // serves half of x from the Infinity Cache)
  if (sizeof(TI) == 4 && C % 128 == 0 && C >= 768 && C <= 1280) {   // (the choice depends on C only: a sample's result must not depend on the batch size)
    const dim3 g2(cdiv(rows, 8)), b2(256);
#define LNH_CASE(NV_) hipLaunchKernelGGL((layernorm_half_kernel<TI, TO, NV_, true>), g2, b2, 0, s, (const TI*)x, (TO*)y, w, b, rows, C, eps, act, rev)
    switch (C / 128) {
      case 6: LNH_CASE(6); break;
      case 7: LNH_CASE(7); break;
      case 8: LNH_CASE(8); break;
      case 9: LNH_CASE(9); break;
      default: LNH_CASE(10); break;
    }
#undef LNH_CASE
    COR_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(cdiv(rows, 4)), block(256);
  const int nv = cdiv(C, 256);
#define LN_CASE(MV) hipLaunchKernelGGL((layernorm_kernel<TI, TO, MV>), grid, block, 0, s, (const TI*)x, (TO*)y, w, b, rows, C, eps, act, rev)
  if (nv <= 1) LN_CASE(1);
  else if (nv <= 2) LN_CASE(2);
  else if (nv <= 3) LN_CASE(3);
  else if (nv <= 4) LN_CASE(4);
  else if (nv <= 5) LN_CASE(5);
  else if (nv <= 8) LN_CASE(8);
  else return COR_ENOSUPPORT;
#undef LN_CASE
  COR_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- row L2 normalise
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) l2norm_kernel(const TI* x, TO* y, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (long)row * C;
  float q = 0.f;
  for (int j = lane; j < C; j += 64) { const float v = ld<TI>(xr + j); q += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), eps);
  TO* yr = y + (long)row * C;
  for (int j = lane; j < C; j += 64) st<TO>(yr + j, ld<TI>(xr + j) * inv);
}

// ---------------------------------------------------------------- elementwise add with periodic second operand
template <typename TA, typename TB, typename TO>
__global__ void __launch_bounds__(256) add_kernel(const TA* a, const TB* b, TO* out, long n4, long period4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 av = ld4<TA>(a + 4 * i), bv = ld4<TB>(b + 4 * (i % period4));
    st4<TO>(out + 4 * i, av + bv);
  }
}

// 4 elements per thread (16-B fp32 / 8-B bf16 accesses) when C, both leading dimensions and both base addresses allow it:
// the element-wise form moved the 131072 x 768 fp32 -> bf16 cast before the neck GEMM at 1.5 TB/s.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) copy_rows4_kernel(const TI* in, long ld_in, TO* out, long ld_out, int rows, int C) {
  const int c4 = C >> 2;
  const long n = (long)rows * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / c4; const int c = (int)(i - r * c4) * 4;
    st4<TO>(out + r * ld_out + c, ld4<TI>(in + r * ld_in + c));
  }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) copy_rows_kernel(const TI* in, long ld_in, TO* out, long ld_out, int rows, int C) {
  const long n = (long)rows * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / C; const int c = (int)(i - r * C);
    st<TO>(out + r * ld_out + c, ld<TI>(in + r * ld_in + c));
  }
}

// ---------------------------------------------------------------- layout: tokens [B,HW,C] <-> NCHW [B,C,HW] via a 32x32 LDS tile
template <typename T>
__global__ void __launch_bounds__(256) tokens_to_nchw_kernel(const T* x, float* out, int HW, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (p < HW && c < C) ? ld<T>(x + ((long)b * HW + p) * C + c) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    if (c < C && p < HW) out[((long)b * C + c) * HW + p] = tile[tx][ty + 8 * i];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) nchw_to_tokens_kernel(const float* x, T* out, int HW, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    tile[ty + 8 * i][tx] = (p < HW && c < C) ? x[((long)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    if (c < C && p < HW) st<T>(out + ((long)b * HW + p) * C + c, tile[tx][ty + 8 * i]);
  }
}

// ---------------------------------------------------------------- patch gather: NCHW fp32 image -> [B*gh*gw, Kpad]
// thread -> 4 consecutive dx of one (patch, c, dy): 16-B loads along the image row, 4-element stores along k.
template <typename TO>
__global__ void __launch_bounds__(256) patchify_kernel(const float* img, TO* out, int B, int C, int H, int W, int p, int Kpad) {
  const int gh = H / p, gw = W / p, K = C * p * p;
  const int kq = Kpad >> 2;                       // groups of 4 along k
  const long total = (long)B * gh * gw * kq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / kq; const int k = (int)(i - row * kq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      const int c = k / (p * p), rem = k - c * p * p, dy = rem / p, dx = rem - dy * p;
      const int b = (int)(row / (gh * gw)), pr = (int)(row - (long)b * gh * gw), py = pr / gw, px = pr - py * gw;
      const float* src = img + (((long)b * C + c) * H + py * p + dy) * W + px * p + dx;
      if (dx + 3 < p && (((uintptr_t)src) & 15) == 0) v = *(const f32x4*)src;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // slow path: group straddles an image row (p % 4 != 0)
          const int kk = k + e;
          if (kk < K) {
            const int c2 = kk / (p * p), r2 = kk - c2 * p * p, dy2 = r2 / p, dx2 = r2 - dy2 * p;
            v[e] = img[(((long)b * C + c2) * H + py * p + dy2) * W + px * p + dx2];
          }
        }
      }
    }
    st4<TO>(out + row * Kpad + k, v);
  }
}

// ---------------------------------------------------------------- 3x3 pad-1 im2col on channels-last tokens
template <typename T>
__global__ void __launch_bounds__(256) im2col3x3_kernel(const T* x, T* out, int B, int H, int W, int C) {
  const int c4 = C >> 2;
  const long total = (long)B * H * W * 9 * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % c4); long t = i / c4;
    const int tap = (int)(t % 9); t /= 9;
    const int xw = (int)(t % W); t /= W;
    const int yh = (int)(t % H); const int b = (int)(t / H);
    const int sy = yh + tap / 3 - 1, sx = xw + tap % 3 - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = ld4<T>(x + (((long)b * H + sy) * W + sx) * C + 4 * cg);
    st4<T>(out + (((long)b * H + yh) * W + xw) * 9L * C + (long)tap * C + 4 * cg, v);
  }
}

// 16-byte form (row bytes % 16 == 0): one thread per 16-B chunk of one (token, tap); blockIdx.y = image row (b, yh), so the
// per-thread index math is two small divisions; raw copies, no dtype conversion (the 8-B convert-and-store form ran at 2 TB/s).
__global__ void __launch_bounds__(256) im2col3x3_raw16_kernel(const uint4* x, uint4* out, int H, int W, int cb) {
  const int by = blockIdx.y;                                  // b * H + yh
  const int yh = by % H;
  const int per_row = W * 9 * cb;                             // chunks of one image row of the output
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per_row; i += gridDim.x * 256) {
    const int c = i % cb; const int t = i / cb;
    const int tap = t % 9, xw = t / 9;
    const int sy = yh + tap / 3 - 1, sx = xw + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[((long)(by - yh + sy) * W + sx) * cb + c];
    out[((long)by * W + xw) * 9L * cb + (long)tap * cb + c] = v;
  }
}

__global__ void __launch_bounds__(256) embed_kernel(const long long* ids, const float* table, const float* pos, float* out,
                                                    int rows, int ctx, int D, int vocab) {
  const long n = (long)rows * (D >> 2);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / (D >> 2)), c = (int)(i - (long)r * (D >> 2)) * 4;
    long long id = ids[r];
    const bool oob = id < 0 || id >= vocab;              // nn.Embedding raises here (a wrong tokenizer: SigLIP vs SigLIP2 vocabulary);
    id = oob ? 0 : id;                                   // no fault, but no plausible garbage either: the row becomes NaN and poisons
    f32x4 e = *(const f32x4*)(table + id * D + c);       // that sample's text feature visibly
    const f32x4 p = *(const f32x4*)(pos + (long)(r % ctx) * D + c);
    if (oob) { const float q = __builtin_nanf(""); e = f32x4{q, q, q, q}; }
    *(f32x4*)(out + (long)r * D + c) = e + p;
  }
}

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

}  // namespace

#ifdef COR_PROBES
extern "C" int cor_probe_layernorm_delta(void* x, const void* delta, void* y, const float* w, const float* b, int rows, int C, float eps, int write_x, void* stream) {
  if (C != 768) return COR_ENOSUPPORT;
  const dim3 g2(cdiv(rows, 8)), b2(256);
  if (write_x) hipLaunchKernelGGL((layernorm_delta_probe<6, true>), g2, b2, 0, (hipStream_t)stream, (float*)x, (const bf16_t*)delta, (bf16_t*)y, w, b, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_delta_probe<6, false>), g2, b2, 0, (hipStream_t)stream, (float*)x, (const bf16_t*)delta, (bf16_t*)y, w, b, rows, C, eps);
  COR_CHECK_LAUNCH();
  return 0;
}
#endif

#define DISPATCH2(dt_a, dt_b, CALL)                                             \
  if (dt_a == COR_F32 && dt_b == COR_F32) { CALL(float, float); }               \
  else if (dt_a == COR_F32 && dt_b == COR_BF16) { CALL(float, bf16_t); }        \
  else if (dt_a == COR_BF16 && dt_b == COR_F32) { CALL(bf16_t, float); }        \
  else if (dt_a == COR_BF16 && dt_b == COR_BF16) { CALL(bf16_t, bf16_t); }      \
  else return COR_ENOSUPPORT;

extern "C" int cor_version(void) { return 1; }

extern "C" int cor_layernorm(const void* x, int x_dtype, void* y, int y_dtype, const float* w, const float* b, int rows, int C,
                             float eps, int act, void* stream) {
  if (!x || !y || !w || !b || rows <= 0 || C <= 0 || (C & 3)) return COR_EINVAL;
  const int rev = (act & COR_ORDER_REVERSE) ? 1 : 0;   // rows from the last to the first (see COR_ORDER_REVERSE)
  act &= ~COR_ORDER_REVERSE;
#define CALL(TI, TO) return launch_ln<TI, TO>(x, y, w, b, rows, C, eps, act, rev, (hipStream_t)stream)
  DISPATCH2(x_dtype, y_dtype, CALL)
#undef CALL
}

extern "C" int cor_l2norm_rows(const void* x, int x_dtype, void* y, int y_dtype, int rows, int C, float eps, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0) return COR_EINVAL;
#define CALL(TI, TO) hipLaunchKernelGGL((l2norm_kernel<TI, TO>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const TI*)x, (TO*)y, rows, C, eps)
  DISPATCH2(x_dtype, y_dtype, CALL)
#undef CALL
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_add(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, long n, long b_period,
                       void* stream) {
  if (!a || !b || !out || n <= 0 || b_period <= 0 || (n & 3) || (b_period & 3)) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long n4 = n >> 2, p4 = b_period >> 2;
#define ADD3(TA, TB, TO) hipLaunchKernelGGL((add_kernel<TA, TB, TO>), dim3(grid_for(n4)), dim3(256), 0, s, (const TA*)a, (const TB*)b, (TO*)out, n4, p4)
  const int key = a_dtype * 4 + b_dtype * 2 + out_dtype;
  switch (key) {
    case 0: ADD3(float, float, float); break;
    case 1: ADD3(float, float, bf16_t); break;
    case 2: ADD3(float, bf16_t, float); break;
    case 3: ADD3(float, bf16_t, bf16_t); break;
    case 4: ADD3(bf16_t, float, float); break;
    case 5: ADD3(bf16_t, float, bf16_t); break;
    case 6: ADD3(bf16_t, bf16_t, float); break;
    case 7: ADD3(bf16_t, bf16_t, bf16_t); break;
    default: return COR_ENOSUPPORT;
  }
#undef ADD3
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_copy_rows(const void* in, long ld_in, int in_dtype, void* out, long ld_out, int out_dtype, int rows, int C,
                             void* stream) {
  if (!in || !out || rows <= 0 || C <= 0 || (ld_in != 0 && ld_in < C) || ld_out < C) return COR_EINVAL;  // ld_in == 0: broadcast one row
  const bool vec4 = (C & 3) == 0 && (ld_in & 3) == 0 && (ld_out & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  if (vec4) {
#define CALL(TI, TO) hipLaunchKernelGGL((copy_rows4_kernel<TI, TO>), dim3(grid_for((long)rows * (C >> 2))), dim3(256), 0, (hipStream_t)stream, (const TI*)in, ld_in, (TO*)out, ld_out, rows, C)
    DISPATCH2(in_dtype, out_dtype, CALL)
#undef CALL
  } else {
#define CALL(TI, TO) hipLaunchKernelGGL((copy_rows_kernel<TI, TO>), dim3(grid_for((long)rows * C)), dim3(256), 0, (hipStream_t)stream, (const TI*)in, ld_in, (TO*)out, ld_out, rows, C)
    DISPATCH2(in_dtype, out_dtype, CALL)
#undef CALL
  }
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_tokens_to_nchw(const void* x, int dtype, float* out, int B, int HW, int C, void* stream) {
  if (!x || !out || B <= 0 || HW <= 0 || C <= 0) return COR_EINVAL;
  const dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
  if (dtype == COR_F32) hipLaunchKernelGGL((tokens_to_nchw_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, HW, C);
  else if (dtype == COR_BF16) hipLaunchKernelGGL((tokens_to_nchw_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, HW, C);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_nchw_to_tokens(const float* x, void* out, int out_dtype, int B, int HW, int C, void* stream) {
  if (!x || !out || B <= 0 || HW <= 0 || C <= 0) return COR_EINVAL;
  const dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
  if (out_dtype == COR_F32) hipLaunchKernelGGL((nchw_to_tokens_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x, (float*)out, HW, C);
  else if (out_dtype == COR_BF16) hipLaunchKernelGGL((nchw_to_tokens_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, HW, C);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int p, int Kpad, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || p <= 0 || H < p || W < p || Kpad < C * p * p || (Kpad & 3)) return COR_EINVAL;
  const long total = (long)B * (H / p) * (W / p) * (Kpad >> 2);
  if (out_dtype == COR_F32) hipLaunchKernelGGL((patchify_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, (float*)out, B, C, H, W, p, Kpad);
  else if (out_dtype == COR_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, B, C, H, W, p, Kpad);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_im2col3x3(const void* x, int dtype, void* out, int B, int H, int W, int C, void* stream) {
  if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return COR_EINVAL;
  const long total = (long)B * H * W * 9 * (C >> 2);
  const int esz = dtype == COR_F32 ? 4 : 2;
  if ((dtype == COR_F32 || dtype == COR_BF16) && (C * esz) % 16 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 && (long)B * H < 65536) {
    const int cb = C * esz / 16, per_row = W * 9 * cb;
    hipLaunchKernelGGL(im2col3x3_raw16_kernel, dim3(cdiv(per_row, 256), B * H), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)out, H, W, cb);
    COR_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == COR_F32) hipLaunchKernelGGL((im2col3x3_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, B, H, W, C);
  else if (dtype == COR_BF16) hipLaunchKernelGGL((im2col3x3_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, B, H, W, C);
  else return COR_ENOSUPPORT;
  COR_CHECK_LAUNCH();
  return 0;
}

extern "C" int cor_embed_tokens(const long long* ids, const float* table, const float* pos, float* out, int rows, int ctx, int D,
                                int vocab, void* stream) {
  if (!ids || !table || !pos || !out || rows <= 0 || ctx <= 0 || D <= 0 || (D & 3) || vocab <= 0) return COR_EINVAL;
  hipLaunchKernelGGL(embed_kernel, dim3(grid_for((long)rows * (D >> 2))), dim3(256), 0, (hipStream_t)stream, ids, table, pos, out, rows, ctx, D, vocab);
  COR_CHECK_LAUNCH();
  return 0;
}
