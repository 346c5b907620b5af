// cor_amd — gallery similarity + top-k for gfx950 (the retrieval end of the path).
//
//   score[b,g] = Q[b,:] . G[g,:]   (unit vectors: cosine, utils/loss_func.py:84)  ->  per query top-k by
//   (score desc, global index asc).
//
// The [Bq, Ng] score matrix is never written to HBM. One WAVE owns 32 queries (their K-fragments stay in
// registers for the whole kernel) and streams a slice of gallery rows through the MFMA as the A operand:
// D[row = gallery row, col = query] puts each query on a lane, so every lane keeps a private sorted top-k list
// of its own 16 scores per 32x32 tile (compare against the list minimum; insertion is rare after warm-up).
// Partial lists (2 lane-halves x nsplit gallery slices) go to a small workspace; a second kernel merges them.
//
// fp32 gallery: v_mfma_f32_32x32x2_f32 — bitwise a k-ordered fmaf chain. With 16-B operand loads the chain
// order is, for c = 0..C/8-1, i = 0..3:  k = 8c+i  then  k = 8c+4+i. oracle/c/sim_chain.c restates exactly this
// chain, so fp32 scores (and hence top-k indices) can be checked BITWISE against the CPU.
// bf16 / fp16 gallery: v_mfma_f32_32x32x16_{bf16,f16}; Q is rounded to the gallery dtype in registers.
#include <limits.h>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct TopkPlan { int nqt, nsplit, tiles_per_split, nparts, kmax; };

inline TopkPlan make_plan(int Bq, int Ng, int k) {
  TopkPlan p;
  p.kmax = k <= 8 ? 8 : 32;
  p.nqt = cdiv(Bq, 32);
  const int tiles = cdiv(Ng, 32);
  int want = cdiv(4096, p.nqt);                       // ~16 waves per CU over the whole chip
  const int cap = (4096 / p.kmax) / 2;                // merge kernel holds nparts*kmax <= 4096 candidates (32 KiB) in LDS
  if (want > cap) want = cap;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  p.tiles_per_split = cdiv(tiles, want);
  p.nsplit = cdiv(tiles, p.tiles_per_split);
  p.nparts = 2 * p.nsplit;
  return p;
}

template <int KMAX>
__device__ __forceinline__ void topk_insert(float (&ls)[KMAX], int (&li)[KMAX], float s, int idx) {
  if (s > ls[KMAX - 1]) {
    ls[KMAX - 1] = s; li[KMAX - 1] = idx;
#pragma unroll
    for (int j = KMAX - 1; j > 0; --j) {
      if (ls[j] > ls[j - 1]) {       // strict: an equal, earlier (smaller index) entry stays ahead
        const float ts = ls[j]; ls[j] = ls[j - 1]; ls[j - 1] = ts;
        const int ti = li[j]; li[j] = li[j - 1]; li[j - 1] = ti;
      }
    }
  }
}

// TG: float (exact chain), bf16_t, _Float16. C (embedding dim) <= 256, multiple of 16.
template <typename TG, int KMAX>
__global__ void __launch_bounds__(256) sim_topk_partial(const float* __restrict__ Q, const TG* __restrict__ G, int Bq, int Ng, int C,
                                                        TopkPlan plan, float* ws_s, int* ws_i) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= plan.nqt * plan.nsplit) return;
  const int qt = wid % plan.nqt, split = wid / plan.nqt;
  const int q = qt * 32 + r;
  const float* qrow = Q + (long)min(q, Bq - 1) * C;

  constexpr bool F32 = sizeof(TG) == 4;
  constexpr int CH = F32 ? 8 : 16;          // k per chunk pair
  constexpr int NCH = 256 / CH;             // max chunks held in registers
  const int nch = C / CH;
  uint4 qf[NCH];                            // fp32: 4 floats ; 16-bit: 8 values
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c < nch) {
      if (F32) {
        qf[c] = *(const uint4*)(qrow + c * 8 + 4 * h);
      } else {
        const f32x4 lo = *(const f32x4*)(qrow + c * 16 + 8 * h), hi = *(const f32x4*)(qrow + c * 16 + 8 * h + 4);
        if (sizeof(TG) == 2 && __is_same(TG, bf16_t)) {
          qf[c].x = (uint32_t)f2bf(lo[0]) | ((uint32_t)f2bf(lo[1]) << 16); qf[c].y = (uint32_t)f2bf(lo[2]) | ((uint32_t)f2bf(lo[3]) << 16);
          qf[c].z = (uint32_t)f2bf(hi[0]) | ((uint32_t)f2bf(hi[1]) << 16); qf[c].w = (uint32_t)f2bf(hi[2]) | ((uint32_t)f2bf(hi[3]) << 16);
        } else {
          f16x8 t = {(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
          qf[c] = __builtin_bit_cast(uint4, t);
        }
      }
    }
  }

  float ls[KMAX]; int li[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { ls[j] = -INFINITY; li[j] = INT_MAX; }

  const int t0 = split * plan.tiles_per_split;
  const int t1 = min(t0 + plan.tiles_per_split, cdiv(Ng, 32));
  for (int t = t0; t < t1; ++t) {
    const int g0 = t * 32;
    const char* grow = (const char*)(G + (long)min(g0 + r, Ng - 1) * C) + 16 * h;   // lane's 16-B column inside each 32-B chunk pair
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nch) {
        const uint4 a = *(const uint4*)(grow + c * 32);
        if (F32) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(qf[c].x), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(qf[c].y), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(qf[c].z), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(qf[c].w), acc, 0, 0, 0);
        } else if (__is_same(TG, bf16_t)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, qf[c]), acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int g = g0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const float s = g < Ng ? acc[e] : -INFINITY;
      topk_insert<KMAX>(ls, li, s, g);
    }
  }
  if (q < Bq) {
    const long base = ((long)q * plan.nparts + split * 2 + h) * KMAX;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { ws_s[base + j] = ls[j]; ws_i[base + j] = li[j]; }
  }
}

__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// one block per query: k rounds of block-wide arg-best over nparts*KMAX candidates held in LDS.
__global__ void __launch_bounds__(256) sim_topk_merge(const float* ws_s, const int* ws_i, int n, int k, long long g_offset,
                                                      float* out_s, long long* out_i) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  float* cs = (float*)smraw; int* ci = (int*)(cs + n);
  __shared__ float rs[4]; __shared__ int ri[4]; __shared__ int rp[4];
  const int qb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += 256) { cs[i] = ws_s[(long)qb * n + i]; ci[i] = ws_i[(long)qb * n + i]; }
  __syncthreads();
  for (int round = 0; round < k; ++round) {
    float bs = -INFINITY; int bi = INT_MAX, bp = -1;
    for (int i = tid; i < n; i += 256)
      if (bp < 0 || better(cs[i], ci[i], bs, bi)) { bs = cs[i]; bi = ci[i]; bp = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64); const int oi = __shfl_xor(bi, o, 64), op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || better(os, oi, bs, bi) || (os == bs && oi == bi && op < bp))) { bs = os; bi = oi; bp = op; }
    }
    if (lane == 0) { rs[wave] = bs; ri[wave] = bi; rp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = rs[0]; int fi = ri[0], fp = rp[0];
      for (int w = 1; w < 4; ++w)
        if (rp[w] >= 0 && (fp < 0 || better(rs[w], ri[w], fs, fi) || (rs[w] == fs && ri[w] == fi && rp[w] < fp))) { fs = rs[w]; fi = ri[w]; fp = rp[w]; }
      out_s[(long)qb * k + round] = fs;
      out_i[(long)qb * k + round] = (fi == INT_MAX) ? -1LL : (long long)fi + g_offset;
      if (fp >= 0) { cs[fp] = -INFINITY; ci[fp] = INT_MAX; }
    }
    __syncthreads();
  }
}

template <typename TG>
int launch_topk(const float* Q, const void* G, int Bq, int Ng, int C, int k, long long g_offset, float* out_s, long long* out_i,
                void* workspace, hipStream_t s) {
  const TopkPlan p = make_plan(Bq, Ng, k);
  float* ws_s = (float*)workspace;
  int* ws_i = (int*)(ws_s + (long)Bq * p.nparts * p.kmax);
  const int nwaves = p.nqt * p.nsplit;
  if (p.kmax == 8) hipLaunchKernelGGL((sim_topk_partial<TG, 8>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  else hipLaunchKernelGGL((sim_topk_partial<TG, 32>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  COR_CHECK_LAUNCH();
  const int n = p.nparts * p.kmax;
  hipLaunchKernelGGL(sim_topk_merge, dim3(Bq), dim3(256), (size_t)n * 8, s, ws_s, ws_i, n, k, g_offset, out_s, out_i);
  COR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" long cor_topk_workspace_bytes(int Bq, int Ng, int k) {
  if (Bq <= 0 || Ng <= 0 || k <= 0 || k > 32) return COR_EINVAL;
  const TopkPlan p = make_plan(Bq, Ng, k);
  return (long)Bq * p.nparts * p.kmax * 8;
}

extern "C" int cor_similarity_topk(const float* Q, const void* G, int g_dtype, int Bq, int Ng, int C, int k, long long g_offset,
                                   float* out_scores, long long* out_idx, void* workspace, void* stream) {
  if (!Q || !G || !out_scores || !out_idx || !workspace || Bq <= 0 || Ng <= 0 || k <= 0) return COR_EINVAL;
  if (k > 32 || C > 256 || (C & 15)) return COR_ENOSUPPORT;
  if (((uintptr_t)Q & 15) || ((uintptr_t)G & 15)) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (g_dtype) {
    case COR_F32: return launch_topk<float>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, s);
    case COR_BF16: return launch_topk<bf16_t>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, s);
    case COR_F16: return launch_topk<_Float16>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, s);
    default: return COR_ENOSUPPORT;
  }
}
