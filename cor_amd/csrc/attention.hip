// cor_amd — attention kernels for gfx950.
//
// attn_rowlane: exact-fp32 flash-style attention, ONE QUERY PER LANE. q and the output accumulator live in
// registers (2*HD VGPRs); keys/values are wave-uniform addresses, so they arrive through the scalar cache
// (s_load) or as broadcast vector loads and cost no LDS and no barrier; online softmax over chunks of 8 keys.
// It serves (a) the fp32 parity mode of every attention on the path, (b) the decoder's tiny attentions
// (6 tokens <-> 4096 tokens, hd 16/32) in both modes, (c) head sizes the MFMA kernel does not take (hd 72).
// SAM's decomposed relative position bias is folded in: per key ROW kh the lane computes q.Rh[qh-kh+S-1] once,
// and q.Rw[qw-kw+S-1] for all kw is precomputed per lane into LDS ([S][128] floats, lane-strided: conflict-free).
// Window partition / zero padding / unpartition are pure addressing (padded tokens read the qkv bias row).
//
// The bf16 MFMA flash kernels (hd = 64) live in flash_attn.hip.
#include "common.h"

namespace {

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o;
  long q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;   // element strides: batch, token
  int H, Tq, Tk;
  float scale;
  // SAM
  const void* pad_row;                 // [3*H*HD] (q|k|v) in T
  const float* rel_h; const float* rel_w;
  int S;                               // rel-pos grid (window size, or grid for global)
  int grid, nW;                        // window mode: image grid and windows per side
};

constexpr int NT = 128;                // threads (= queries) per block

template <typename T, int HD>
__device__ __forceinline__ float dot_q(const float (&q)[HD], const T* __restrict__ p) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) {
    const f32x4 kv = ld4<T>(p + 4 * i);
    a0 = fmaf(q[4 * i + 0], kv[0], a0); a1 = fmaf(q[4 * i + 1], kv[1], a1);
    a0 = fmaf(q[4 * i + 2], kv[2], a0); a1 = fmaf(q[4 * i + 3], kv[3], a1);
  }
  return a0 + a1;
}

// MODE 0: plain strided MHA. MODE 1: SAM global (rel-pos, S = grid). MODE 2: SAM windowed (rel-pos, S = window).
template <typename T, typename TO, int HD, int MODE>
__global__ void __launch_bounds__(NT) attn_rowlane(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rw_lds[];   // [S][NT], MODE != 0
  const int tid = threadIdx.x;
  const int h = blockIdx.y;
  const int bz = blockIdx.z;
  const int S = a.S;
  int t = blockIdx.x * NT + tid;
  bool valid = t < a.Tq;
  if (!valid) t = a.Tq - 1;

  const T* qbase = (const T*)a.q; const T* kbase = (const T*)a.k; const T* vbase = (const T*)a.v;
  const T* pad = (const T*)a.pad_row;
  const int d3 = 3 * a.H * HD;         // SAM qkv row length
  int qh = 0, qw = 0;
  long orow = 0;                       // output row (tokens) for SAM modes
  const T* qp;
  int b = bz, wy = 0, wx = 0;
  if (MODE == 0) {
    qp = qbase + bz * a.q_sb + (long)t * a.q_st + h * HD;
  } else if (MODE == 1) {
    qh = t / S; qw = t - qh * S;
    orow = (long)bz * a.Tq + t;
    qp = qbase + orow * d3 + h * HD;
  } else {
    const int nw2 = a.nW * a.nW;
    b = bz / nw2; const int wi = bz - b * nw2; wy = wi / a.nW; wx = wi - wy * a.nW;
    qh = t / S; qw = t - qh * S;
    const int y = wy * S + qh, x = wx * S + qw;
    if (y >= a.grid || x >= a.grid) valid = false;
    orow = ((long)b * a.grid + min(y, a.grid - 1)) * a.grid + min(x, a.grid - 1);
    qp = valid ? qbase + orow * d3 + h * HD : pad + h * HD;
  }

  float q[HD], o[HD];
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) {
    const f32x4 v4 = ld4<T>(qp + 4 * i);
    q[4 * i] = v4[0]; q[4 * i + 1] = v4[1]; q[4 * i + 2] = v4[2]; q[4 * i + 3] = v4[3];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;

  if (MODE != 0) {
    for (int kw = 0; kw < S; ++kw) rw_lds[kw * NT + tid] = dot_q<float, HD>(q, a.rel_w + (long)(qw - kw + S - 1) * HD);
    // each lane reads back only its own column: no barrier needed
  }

  float m = -INFINITY, l = 0.f;
  const int rows = MODE == 0 ? 1 : S;
  const int cols = MODE == 0 ? a.Tk : S;
  for (int kh = 0; kh < rows; ++kh) {
    float rh = 0.f;
    if (MODE != 0) rh = dot_q<float, HD>(q, a.rel_h + (long)(qh - kh + S - 1) * HD);
    for (int c0 = 0; c0 < cols; c0 += 8) {
      const int n = min(8, cols - c0);
      float s[8];
      const T* vps[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        s[jj] = -INFINITY; vps[jj] = vbase;
        if (jj < n) {                  // wave-uniform
          const int kw = c0 + jj;
          const T* kp;
          if (MODE == 0) {
            kp = kbase + bz * a.k_sb + (long)kw * a.k_st + h * HD;
            vps[jj] = vbase + bz * a.v_sb + (long)kw * a.v_st + h * HD;
          } else if (MODE == 1) {
            const long krow = (long)bz * a.Tk + kh * S + kw;
            kp = kbase + krow * d3 + h * HD; vps[jj] = vbase + krow * d3 + h * HD;
          } else {
            const int y = wy * S + kh, x = wx * S + kw;
            if (y < a.grid && x < a.grid) {
              const long krow = ((long)b * a.grid + y) * a.grid + x;
              kp = kbase + krow * d3 + h * HD; vps[jj] = vbase + krow * d3 + h * HD;
            } else {
              kp = pad + a.H * HD + h * HD; vps[jj] = pad + 2 * a.H * HD + h * HD;
            }
          }
          float sc = dot_q<T, HD>(q, kp) * a.scale;
          if (MODE != 0) sc += rh + rw_lds[kw * NT + tid];
          s[jj] = sc;
        }
      }
      float mx = m;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) mx = fmaxf(mx, s[jj]);
      const float alpha = __expf(m - mx);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj < n) {
          const float p = __expf(s[jj] - mx);
          l += p;
          const T* vp = vps[jj];
#pragma unroll
          for (int i = 0; i < HD / 4; ++i) {
            const f32x4 vv = ld4<T>(vp + 4 * i);
            o[4 * i] = fmaf(p, vv[0], o[4 * i]); o[4 * i + 1] = fmaf(p, vv[1], o[4 * i + 1]);
            o[4 * i + 2] = fmaf(p, vv[2], o[4 * i + 2]); o[4 * i + 3] = fmaf(p, vv[3], o[4 * i + 3]);
          }
        }
      }
      m = mx;
    }
  }
  if (!valid) return;
  const float inv = 1.0f / l;
  TO* op = MODE == 0 ? (TO*)a.o + bz * a.o_sb + (long)t * a.o_st + h * HD : (TO*)a.o + orow * (long)(a.H * HD) + h * HD;
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) {
    f32x4 r = {o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv};
    st4<TO>(op + 4 * i, r);
  }
}

// ---------------------------------------------------------------------------------------------------------
// attn_fewq: a handful of queries (<= 8) against thousands of keys — the decoder's token -> image cross attention
// (6 tokens x 4096 image keys, 8 heads x 16; lib/sam_model/transformer.py:163-166,99-100). One query per lane would
// leave 58 of 64 lanes idle and walk 4096 keys serially (measured 2.6 ms per launch); here the KEYS are spread over
// the 256 lanes of a block (one block per (batch, head)), every lane keeps a private online-softmax state for all
// queries, and the partial states are merged once at the end (wave shuffles, then LDS across the 4 waves).
template <typename T, typename TO, int HD, int TQ>
__global__ void __launch_bounds__(256) attn_fewq(const AttnArgs a) {
  __shared__ float qs[TQ][HD];
  __shared__ float red[4][TQ][HD + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const T* qb = (const T*)a.q + b * a.q_sb + h * HD;
  const T* kb = (const T*)a.k + b * a.k_sb + h * HD;
  const T* vb = (const T*)a.v + b * a.v_sb + h * HD;
  for (int i = tid; i < TQ * HD; i += 256) {
    const int qi = i / HD, d = i - qi * HD;
    qs[qi][d] = qi < a.Tq ? ld<T>(qb + (long)qi * a.q_st + d) * a.scale : 0.f;
  }
  __syncthreads();
  float m[TQ], l[TQ], o[TQ][HD];
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) {
    m[qi] = -INFINITY; l[qi] = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[qi][d] = 0.f;
  }
  for (int j = tid; j < a.Tk; j += 256) {
    float kv[HD], vv[HD];
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const f32x4 k4 = ld4<T>(kb + (long)j * a.k_st + 4 * i), v4 = ld4<T>(vb + (long)j * a.v_st + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kv[4 * i + e] = k4[e]; vv[4 * i + e] = v4[e]; }
    }
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) sc = fmaf(qs[qi][d], kv[d], sc);
      const float mn = fmaxf(m[qi], sc);
      const float alpha = __expf(m[qi] - mn), p = __expf(sc - mn);
      l[qi] = l[qi] * alpha + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[qi][d] = fmaf(p, vv[d], o[qi][d] * alpha);
      m[qi] = mn;
    }
  }
  // merge the 64 lane states of a wave, then the 4 waves
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) {
    const float mw = wave_max(m[qi]);
    const float f = m[qi] == -INFINITY ? 0.f : __expf(m[qi] - mw);
    const float lw = wave_sum(l[qi] * f);
#pragma unroll
    for (int d = 0; d < HD; ++d) o[qi][d] = wave_sum(o[qi][d] * f);
    if (lane == 0) {
      red[wave][qi][HD] = mw; red[wave][qi][HD + 1] = lw;
#pragma unroll
      for (int d = 0; d < HD; ++d) red[wave][qi][d] = o[qi][d];
    }
  }
  __syncthreads();
  for (int i = tid; i < a.Tq * HD; i += 256) {
    const int qi = i / HD, d = i - qi * HD;
    float mm = -INFINITY;
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[w][qi][HD]);
    float num = 0.f, den = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float f = red[w][qi][HD] == -INFINITY ? 0.f : __expf(red[w][qi][HD] - mm);
      num += red[w][qi][d] * f; den += red[w][qi][HD + 1] * f;
    }
    st<TO>((TO*)a.o + b * a.o_sb + (long)qi * a.o_st + h * HD + d, num / den);
  }
}

// Round 5, the same arithmetic for SMALL grids (B * H < 128 blocks: batch 1..15, where this kernel sits three times on the forward's
// critical path and a lane's chain of 16 keys x 8 queries x ~38 dependent instructions takes 40 us however few blocks run): one thread per
// (query, key slot) instead of one per key slot - 4 queries x 256 slots = 16 waves per block, two blocks per (batch, head). Slot s still
// walks keys s, s + 256, ... in order, a wave still merges the same 64 slots by the same butterflies and the four waves of a query are
// merged in the same order, so the result is BIT-IDENTICAL to attn_fewq (a sample's output does not depend on the batch size).
template <typename T, typename TO, int HD, int TQ>
__global__ void __launch_bounds__(1024) attn_fewq_wide(const AttnArgs a) {
  __shared__ float red[16][HD + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = (wave >> 2) + (TQ / 2) * blockIdx.z, slot = (wave & 3) * 64 + lane;
  const int h = blockIdx.x, b = blockIdx.y;
  const T* qb = (const T*)a.q + b * a.q_sb + h * HD;
  const T* kb = (const T*)a.k + b * a.k_sb + h * HD;
  const T* vb = (const T*)a.v + b * a.v_sb + h * HD;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = qi < a.Tq ? ld<T>(qb + (long)qi * a.q_st + d) * a.scale : 0.f;
  float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  if (qi < a.Tq) {                                                     // (wave-uniform)
    for (int j = slot; j < a.Tk; j += 256) {
      float kv[HD], vv[HD];
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        const f32x4 k4 = ld4<T>(kb + (long)j * a.k_st + 4 * i), v4 = ld4<T>(vb + (long)j * a.v_st + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kv[4 * i + e] = k4[e]; vv[4 * i + e] = v4[e]; }
      }
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) sc = fmaf(q[d], kv[d], sc);
      const float mn = fmaxf(m, sc);
      const float alpha = __expf(m - mn), p = __expf(sc - mn);
      l = l * alpha + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(p, vv[d], o[d] * alpha);
      m = mn;
    }
  }
  {
    const float mw = wave_max(m);
    const float f = m == -INFINITY ? 0.f : __expf(m - mw);
    const float lw = wave_sum(l * f);
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = wave_sum(o[d] * f);
    if (lane == 0) {
      red[wave][HD] = mw; red[wave][HD + 1] = lw;
#pragma unroll
      for (int d = 0; d < HD; ++d) red[wave][d] = o[d];
    }
  }
  __syncthreads();
  for (int i = tid; i < (TQ / 2) * HD; i += 1024) {
    const int ql = i / HD, d = i - ql * HD, qq = ql + (TQ / 2) * blockIdx.z;
    if (qq >= a.Tq) continue;
    float mm = -INFINITY;
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[4 * ql + w][HD]);
    float num = 0.f, den = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float f = red[4 * ql + w][HD] == -INFINITY ? 0.f : __expf(red[4 * ql + w][HD] - mm);
      num += red[4 * ql + w][d] * f; den += red[4 * ql + w][HD + 1] * f;
    }
    st<TO>((TO*)a.o + b * a.o_sb + (long)qq * a.o_st + h * HD + d, num / den);
  }
}

template <typename T, typename TO>
int launch_fewq(const AttnArgs& a, int B, hipStream_t s) {
  if ((long)a.H * B < 128) hipLaunchKernelGGL((attn_fewq_wide<T, TO, 16, 8>), dim3(a.H, B, 2), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL((attn_fewq<T, TO, 16, 8>), dim3(a.H, B), dim3(256), 0, s, a);
  COR_CHECK_LAUNCH();
  return 0;
}

template <typename T, typename TO, int HD, int MODE>
int launch_rowlane(const AttnArgs& a, int nb, hipStream_t s) {
  const size_t lds = MODE == 0 ? 0 : (size_t)a.S * NT * sizeof(float);
  if (lds > 64 * 1024) return COR_ENOSUPPORT;
  hipLaunchKernelGGL((attn_rowlane<T, TO, HD, MODE>), dim3(cdiv(a.Tq, NT), a.H, nb), dim3(NT), lds, s, a);
  COR_CHECK_LAUNCH();
  return 0;
}

template <int HD, int MODE>
int dispatch_types(const AttnArgs& a, int nb, int dtype, int out_dtype, hipStream_t s) {
  if (dtype == COR_F32 && out_dtype == COR_F32) return launch_rowlane<float, float, HD, MODE>(a, nb, s);
  if (dtype == COR_BF16 && out_dtype == COR_BF16) return launch_rowlane<bf16_t, bf16_t, HD, MODE>(a, nb, s);
  if (dtype == COR_BF16 && out_dtype == COR_F32) return launch_rowlane<bf16_t, float, HD, MODE>(a, nb, s);
  if (dtype == COR_F32 && out_dtype == COR_BF16) return launch_rowlane<float, bf16_t, HD, MODE>(a, nb, s);
  return COR_ENOSUPPORT;
}

}  // namespace

// bf16 MFMA kernels (flash_attn.hip); return COR_ENOSUPPORT when the shape is not theirs.
int cor_flash_plain_bf16(const void* q, long q_sb, long q_st, const void* k, long k_sb, long k_st, const void* v, long v_sb,
                         long v_st, void* out, long o_sb, long o_st, int out_dtype, int B, int H, int Tq, int Tk, int hd, float scale,
                         hipStream_t s);
int cor_flash_sam_bf16(const void* qkv, void* out, int out_dtype, const void* pad_row, const float* rel_h, const float* rel_w,
                       int B, int H, int hd, int grid, int window, float q_prescale, int variant, hipStream_t s);

// Which kernel family the dispatchers below pick for 16-byte-aligned operands (tests assert that head_dim 72 / 80 and the SAM
// shapes run on the matrix cores, not on the row-per-lane VALU kernel). sam_window: -1 = cor_attention, 0 = cor_sam_attention
// global, > 0 = windowed. Returns COR_KERNEL_* or COR_ENOSUPPORT.
extern "C" int cor_attention_kernel_id(int dtype, int hd, int Tq, int Tk, int sam_window, int grid) {
  if (sam_window < 0) {
    if (dtype == COR_BF16 && (hd == 64 || hd == 72 || hd == 80) && Tq >= 64 && Tk >= 64) return COR_KERNEL_FLASH_MFMA;
    if (hd == 16 && Tq <= 8 && Tk >= 512) return COR_KERNEL_FEWQ;
    return (hd == 16 || hd == 32 || hd == 64 || hd == 72 || hd == 80) ? COR_KERNEL_ROWLANE : COR_ENOSUPPORT;
  }
  if (dtype == COR_BF16 && (hd == 64 || hd == 80)) {
    if (sam_window == 0 && grid == 64) return hd == 64 ? COR_KERNEL_FLASH_PIPELINED : COR_KERNEL_FLASH_MFMA;
    if (sam_window == 14) return hd == 64 ? COR_KERNEL_WINDOW_BLOCK : COR_KERNEL_FLASH_MFMA;
  }
  return (hd == 16 || hd == 32 || hd == 64 || hd == 80) ? COR_KERNEL_ROWLANE : COR_ENOSUPPORT;
}

extern "C" int cor_attention(const void* q, long q_sb, long q_st, const void* k, long k_sb, long k_st, const void* v, long v_sb,
                             long v_st, int dtype, void* out, long o_sb, long o_st, int out_dtype, int B, int H, int Tq, int Tk,
                             int hd, float scale, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == COR_BF16 && (hd == 64 || hd == 72 || hd == 80) && Tq >= 64 && Tk >= 64) {     // bf16 MFMA flash kernels
    const int rc = cor_flash_plain_bf16(q, q_sb, q_st, k, k_sb, k_st, v, v_sb, v_st, out, o_sb, o_st, out_dtype, B, H, Tq, Tk, hd, scale, s);
    if (rc != COR_ENOSUPPORT) return rc;
  }
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = out;
  a.q_sb = q_sb; a.q_st = q_st; a.k_sb = k_sb; a.k_st = k_st; a.v_sb = v_sb; a.v_st = v_st; a.o_sb = o_sb; a.o_st = o_st;
  a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.S = 0;
  if (hd == 16 && Tq <= 8 && Tk >= 512) {
    if (dtype == COR_F32 && out_dtype == COR_F32) return launch_fewq<float, float>(a, B, s);
    if (dtype == COR_BF16 && out_dtype == COR_BF16) return launch_fewq<bf16_t, bf16_t>(a, B, s);
    if (dtype == COR_BF16 && out_dtype == COR_F32) return launch_fewq<bf16_t, float>(a, B, s);
    if (dtype == COR_F32 && out_dtype == COR_BF16) return launch_fewq<float, bf16_t>(a, B, s);
  }
  switch (hd) {
    case 16: return dispatch_types<16, 0>(a, B, dtype, out_dtype, s);
    case 32: return dispatch_types<32, 0>(a, B, dtype, out_dtype, s);
    case 64: return dispatch_types<64, 0>(a, B, dtype, out_dtype, s);
    case 72: return dispatch_types<72, 0>(a, B, dtype, out_dtype, s);
    case 80: return dispatch_types<80, 0>(a, B, dtype, out_dtype, s);
    default: return COR_ENOSUPPORT;
  }
}

template <int HD>
static int sam_rowlane(AttnArgs a, int B, int grid, int window, int dtype, int out_dtype, hipStream_t s) {
  a.scale = 1.0f / sqrtf((float)HD);
  if (window == 0) {
    a.S = grid; a.Tq = a.Tk = grid * grid; a.nW = 1;
    return dispatch_types<HD, 1>(a, B, dtype, out_dtype, s);
  }
  a.S = window; a.Tq = a.Tk = window * window; a.nW = (grid + window - 1) / window;
  return dispatch_types<HD, 2>(a, B * a.nW * a.nW, dtype, out_dtype, s);
}

extern "C" int cor_sam_attention(const void* qkv, int dtype, void* out, int out_dtype, const void* pad_row, const float* rel_h,
                                 const float* rel_w, int B, int H, int hd, int grid, int window, float q_prescale, int variant, void* stream) {
  if (!qkv || !out || !rel_h || !rel_w || B <= 0 || H <= 0 || grid <= 0 || window < 0 || variant < 0) return COR_EINVAL;
  if (!(q_prescale > 0.f)) return COR_EINVAL;
  if (window > 0 && !pad_row) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == COR_BF16 && (hd == 64 || hd == 80)) {
    const int rc = cor_flash_sam_bf16(qkv, out, out_dtype, pad_row, rel_h, rel_w, B, H, hd, grid, window, q_prescale, variant, s);
    if (rc != COR_ENOSUPPORT) return rc;
  }
  if (q_prescale != 1.0f) return COR_ENOSUPPORT;       // the row-per-lane kernels take the raw q
  const int HD = hd;
  const long d = (long)H * HD;
  const size_t esz = dtype == COR_F32 ? 4 : 2;
  AttnArgs a{};
  a.q = qkv; a.k = (const char*)qkv + d * esz; a.v = (const char*)qkv + 2 * d * esz; a.o = out;
  a.H = H; a.pad_row = pad_row; a.rel_h = rel_h; a.rel_w = rel_w; a.grid = grid;
  switch (hd) {
    case 16: return sam_rowlane<16>(a, B, grid, window, dtype, out_dtype, s);
    case 32: return sam_rowlane<32>(a, B, grid, window, dtype, out_dtype, s);
    case 64: return sam_rowlane<64>(a, B, grid, window, dtype, out_dtype, s);
    case 80: return sam_rowlane<80>(a, B, grid, window, dtype, out_dtype, s);   // SAM ViT-H (1280 / 16 heads)
    default: return COR_ENOSUPPORT;
  }
}
