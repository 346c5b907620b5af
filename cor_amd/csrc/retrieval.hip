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

// Offer the 16 scores of one 32x32 accumulator tile (this lane's query, gallery rows g0 + row(e,h)) to the lane's list.
// After warm-up almost no score beats the list minimum, so the insertion bubble (a few hundred predicated moves once
// unrolled) must not even be ISSUED in the common case: one max over the tile and one wave-uniform ballot skip it all,
// and inside, each element is again guarded by a ballot (hipcc otherwise predicates the bubble without a branch and a
// tile cost ~15k cycles of v_mov with an empty EXEC mask).
template <int KMAX>
__device__ __forceinline__ void topk_offer_tile(float (&ls)[KMAX], int (&li)[KMAX], const f32x16& acc, int g0, int h, int Ng) {
  float sc[16];
  float tmax = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int g = g0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    sc[e] = g < Ng ? acc[e] : -INFINITY;
    tmax = fmaxf(tmax, sc[e]);
  }
  if (__builtin_amdgcn_ballot_w64(tmax > ls[KMAX - 1]) == 0) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (__builtin_amdgcn_ballot_w64(sc[e] > ls[KMAX - 1]) != 0) topk_insert<KMAX>(ls, li, sc[e], g0 + (e & 3) + 8 * (e >> 2) + 4 * h);
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
    topk_offer_tile<KMAX>(ls, li, acc, g0, h, Ng);
  }
  if (q < Bq) {
    const long base = ((long)q * plan.nparts + split * 2 + h) * KMAX;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { ws_s[base + j] = ls[j]; ws_i[base + j] = li[j]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2 (16-bit galleries): the MFMA-bound form for B_tot >= ~256. One block per CU = 4 waves x 64 queries: every wave keeps
// the K-fragments of TWO 32-query blocks in registers (128 VGPRs) for the whole kernel, the gallery slice streams through
// LDS in 32-row tiles (16 KiB, register-staged, double-buffered, shared by the 4 waves => each gallery byte is fetched
// once per 256 queries), and each A fragment read from LDS feeds two MFMAs (one per query block). LDS rows are 512 B, so
// chunk c of row r sits in slot c ^ (r & 15): the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-B slots.
struct TopkPlan2 { int nqg, nsplit, tiles_per_split, nparts, kmax; };

inline TopkPlan2 make_plan2(int Bq, int Ng, int k, int n_cu) {
  TopkPlan2 p;
  p.kmax = k <= 8 ? 8 : (k <= 16 ? 16 : 32);
  p.nqg = cdiv(Bq, 256);
  const int tiles = cdiv(Ng, 32);
  int want = n_cu / p.nqg;                            // one resident block per CU
  const int cap = (8192 / p.kmax) / 2;                // merge kernel: nparts*kmax <= 8192 candidates in LDS
  if (want > cap) want = cap;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  p.tiles_per_split = cdiv(tiles, want);
  p.nsplit = cdiv(tiles, p.tiles_per_split);
  p.nparts = 2 * p.nsplit;
  return p;
}

template <typename TG>
__device__ __forceinline__ uint4 q_frag16(const float* qrow, int c, int h) {
  const f32x4 lo = *(const f32x4*)(qrow + c * 16 + 8 * h), hi = *(const f32x4*)(qrow + c * 16 + 8 * h + 4);
  if (__is_same(TG, bf16_t)) {
    uint4 u;
    u.x = (uint32_t)f2bf(lo[0]) | ((uint32_t)f2bf(lo[1]) << 16); u.y = (uint32_t)f2bf(lo[2]) | ((uint32_t)f2bf(lo[3]) << 16);
    u.z = (uint32_t)f2bf(hi[0]) | ((uint32_t)f2bf(hi[1]) << 16); u.w = (uint32_t)f2bf(hi[2]) | ((uint32_t)f2bf(hi[3]) << 16);
    return u;
  }
  f16x8 t = {(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
  return __builtin_bit_cast(uint4, t);
}

template <typename TG, int KMAX>
__global__ void __launch_bounds__(256, 1) sim_topk_v2(const float* __restrict__ Q, const TG* __restrict__ G, int Bq, int Ng,
                                                      TopkPlan2 plan, float* ws_s, int* ws_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 16 KiB gallery tiles
  constexpr int C = 256, TILE = 32 * C * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int qg = blockIdx.x % plan.nqg, split = blockIdx.x / plan.nqg;
  const int q0 = qg * 256 + wave * 64;
  const bool active = q0 < Bq;                                  // wave-uniform: idle waves still stage and barrier

  uint4 qf[2][16];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float* qrow = Q + (long)min(q0 + qb * 32 + r, Bq - 1) * C;
#pragma unroll
    for (int c = 0; c < 16; ++c) qf[qb][c] = q_frag16<TG>(qrow, c, h);
  }
  float ls[2][KMAX]; int li[2][KMAX];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { ls[qb][j] = -INFINITY; li[qb][j] = INT_MAX; }

  // staging: thread owns chunks c = tid + 256 i (i < 4) of the 32 x 32-chunk tile
  int st_row[4], st_dst[4], st_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 5, ch = c & 31;
    st_row[i] = row; st_src[i] = ch * 8; st_dst[i] = row * 512 + ((ch ^ (row & 15)) << 4);
  }
  int rd[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) rd[c] = r * 512 + (((2 * c + h) ^ (r & 15)) << 4);

  const int ntiles = cdiv(Ng, 32);
  const int t0 = split * plan.tiles_per_split, t1 = min(t0 + plan.tiles_per_split, ntiles);
  uint4 rg0, rg1, rg2, rg3;
#define SIM_GLOAD(T_)                                                                              \
  {                                                                                                \
    const int g0_ = (T_) * 32;                                                                     \
    rg0 = *(const uint4*)(G + (long)min(g0_ + st_row[0], Ng - 1) * C + st_src[0]);                 \
    rg1 = *(const uint4*)(G + (long)min(g0_ + st_row[1], Ng - 1) * C + st_src[1]);                 \
    rg2 = *(const uint4*)(G + (long)min(g0_ + st_row[2], Ng - 1) * C + st_src[2]);                 \
    rg3 = *(const uint4*)(G + (long)min(g0_ + st_row[3], Ng - 1) * C + st_src[3]);                 \
  }
#define SIM_LSTORE(B_)                                                                             \
  {                                                                                                \
    char* d_ = smem + (B_) * TILE;                                                                 \
    *(uint4*)(d_ + st_dst[0]) = rg0; *(uint4*)(d_ + st_dst[1]) = rg1;                              \
    *(uint4*)(d_ + st_dst[2]) = rg2; *(uint4*)(d_ + st_dst[3]) = rg3;                              \
  }
  if (t0 < t1) {
    SIM_GLOAD(t0)
    SIM_LSTORE(0)
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const bool more = t + 1 < t1;
    if (more) SIM_GLOAD(t + 1)
    const char* buf = smem + ((t - t0) & 1) * TILE;
    if (active) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint4 a = *(const uint4*)(buf + rd[c]);
        if (__is_same(TG, bf16_t)) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[0][c]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[1][c]), acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, qf[0][c]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, qf[1][c]), acc1, 0, 0, 0);
        }
      }
      topk_offer_tile<KMAX>(ls[0], li[0], acc0, t * 32, h, Ng);
      topk_offer_tile<KMAX>(ls[1], li[1], acc1, t * 32, h, Ng);
    }
    if (more) SIM_LSTORE((t + 1 - t0) & 1)
    __syncthreads();
  }
#undef SIM_GLOAD
#undef SIM_LSTORE
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 32 + r;
    if (q < Bq) {
      const long base = ((long)q * plan.nparts + split * 2 + h) * KMAX;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) { ws_s[base + j] = ls[qb][j]; ws_i[base + j] = li[qb][j]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3 (16-bit galleries, large shards): threshold-and-append. Per-lane sorted lists cost one divergent insertion bubble
// (~500 wave cycles) per accepted score, and a stream of n scores accepts ~k ln(n/k) of them per lane: measured 9x the
// MFMA time. Instead:
//   1. DENSE pass over a strided sample of S gallery rows writes the [Bq, S] scores (same MFMA code, same arithmetic);
//   2. the k-th best sample score of a query, tau_q, is a valid LOWER bound of its k-th best overall score;
//   3. APPEND pass over the whole shard: a lane only compares its tile maximum with tau_q and appends the rare
//      scores >= tau_q (expected k*Ng/S per query) to a per-query candidate list (atomic slot counter);
//   4. exact selection over the candidates by (score desc, index asc).
// Every row of the global top-k has score >= tau_q, so the result is exact; a candidate-list overflow (pathological
// score distributions) is reported through index -2 and the caller re-runs the list kernel.
struct ScanArgs {
  int Bq, Ng, nqg, nsplit, tiles_per_split;
  long row_stride;                 // gallery rows between consecutive scanned rows (DENSE sample: > 1)
  int n_rows;                      // rows scanned (DENSE: S; APPEND: Ng)
  float* dense; long dense_ld;     // DENSE: out[q * dense_ld + j]
  const float* tau; int* cnt; float* cand_s; int* cand_i; int cap;   // APPEND
};

template <typename TG, bool DENSE>
__global__ void __launch_bounds__(256, 2) sim_scan(const float* __restrict__ Q, const TG* __restrict__ G, const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 16 KiB gallery tiles
  constexpr int C = 256, TILE = 32 * C * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int qg = blockIdx.x % a.nqg, split = blockIdx.x / a.nqg;
  const int q0 = qg * 256 + wave * 64;
  const bool active = q0 < a.Bq;

  uint4 qf[2][16];
  float tau[2] = {0.f, 0.f};
  int ncand[2] = {0, 0};
  const int nstreams = a.nsplit * 2;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = min(q0 + qb * 32 + r, a.Bq - 1);
    const float* qrow = Q + (long)q * C;
#pragma unroll
    for (int c = 0; c < 16; ++c) qf[qb][c] = q_frag16<TG>(qrow, c, h);
    if (!DENSE) tau[qb] = a.tau[q];
  }
  // Gallery stream: LDS-DMA ring of NS 16-KiB tiles, AHEAD tiles in flight (80 KiB per CU): with one tile in flight the
  // scan ran at HBM LATENCY (16 KiB per ~4.5 us per CU = 0.9 TB/s chip-wide). The LDS image is lane-linear per wave
  // (64 lanes x 16 B = two 512-B rows), so slot (row, sl) is fed from source chunk sl ^ (row & 15).
  constexpr int NS = 4, AHEAD = 3;                 // 64 KiB of LDS: two blocks (8 waves) per CU
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  const unsigned wbase = __builtin_amdgcn_readfirstlane(tid & ~63) * 16;       // this wave's first slot (bytes) per 4-KiB pass
  int st_row[4], st_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 5, sl = c & 31;
    st_row[i] = row; st_src[i] = (sl ^ (row & 15)) * 8;
  }
  int rd[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) rd[c] = r * 512 + (((2 * c + h) ^ (r & 15)) << 4);

  const int ntiles = cdiv(a.n_rows, 32);
  const int t0 = split * a.tiles_per_split, t1 = min(t0 + a.tiles_per_split, ntiles);
  auto issue = [&](int t) {
    const int g0 = t * 32;
    const unsigned dst = lds0 + ((t - t0) % NS) * TILE + wbase;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(G + (long)min(g0 + st_row[i], a.n_rows - 1) * a.row_stride * C + st_src[i], dst + 256 * 16 * i);
  };
  for (int t = t0; t < min(t0 + AHEAD, t1); ++t) issue(t);
  for (int t = t0; t < t1; ++t) {
    // tile t has landed once at most 4 * (tiles issued after t) of this wave's DMA are still outstanding
    const int later = min(t1 - 1 - t, AHEAD - 1);
    switch (later) {
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();                                   // every wave's part of tile t is in LDS; tile t-1 is fully consumed
    if (t + AHEAD < t1) issue(t + AHEAD);              // -> ring slot of tile t-1
    const char* buf = smem + ((t - t0) % NS) * TILE;
    if (active) {
      f32x16 acc[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
      // A fragments two K-steps ahead of the MFMAs that consume them (one wave per SIMD: no other wave hides LDS latency)
      uint4 af[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) af[c] = *(const uint4*)(buf + rd[c]);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint4 av = af[c & 3];
        if (c + 4 < 16) af[c & 3] = *(const uint4*)(buf + rd[c + 4]);
        if (__is_same(TG, bf16_t)) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, qf[0][c]), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, qf[1][c]), acc[1], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, qf[0][c]), acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, qf[1][c]), acc[1], 0, 0, 0);
        }
      }
      const int g0 = t * 32;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 32 + r;
        if (DENSE) {
          if (q < a.Bq) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int g = g0 + (e & 3) + 8 * (e >> 2) + 4 * h;
              if (g < a.n_rows) a.dense[(long)q * a.dense_ld + g] = acc[qb][e];
            }
          }
        } else {
          float tmax = -INFINITY;
#pragma unroll
          for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, acc[qb][e]);
          if (__builtin_amdgcn_ballot_w64(tmax >= tau[qb]) != 0) {             // rare after the sample threshold
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int g = g0 + (e & 3) + 8 * (e >> 2) + 4 * h;
              if (acc[qb][e] >= tau[qb] && g < a.n_rows && q < a.Bq) {
                // private list of this (query, gallery slice, lane half): a register counter, no atomic round trip
                // (a returning atomicAdd per accepted score serialised ~2 us each: 7x the MFMA time at 1M rows)
                if (ncand[qb] < a.cap) {
                  const long o = ((long)q * nstreams + split * 2 + h) * a.cap + ncand[qb];
                  a.cand_s[o] = acc[qb][e]; a.cand_i[o] = g;
                }
                ++ncand[qb];
              }
            }
          }
        }
      }
    }
  }
  if (!DENSE && active) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + qb * 32 + r;
      if (q < a.Bq) a.cnt[(long)q * nstreams + split * 2 + h] = ncand[qb];
    }
  }
}

// k-th best of each query's S sample scores -> tau (block per query; k rounds of block-wide max with removal in LDS)
__global__ void __launch_bounds__(256) sim_sample_tau(const float* dense, long ld, int S, int k, float* tau, int* cnt) {
  extern __shared__ __attribute__((aligned(16))) float sv[];
  __shared__ float rs[4]; __shared__ int rp[4];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < S; i += 256) sv[i] = dense[(long)q * ld + i];
  (void)cnt;
  __syncthreads();
  float kth = -INFINITY;
  for (int round = 0; round < k; ++round) {
    float bs = -INFINITY; int bp = -1;
    for (int i = tid; i < S; i += 256) if (bp < 0 || sv[i] > bs) { bs = sv[i]; bp = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64); const int op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || os > bs || (os == bs && op < bp))) { bs = os; bp = op; }
    }
    if (lane == 0) { rs[wave] = bs; rp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = rs[0]; int fp = rp[0];
      for (int w = 1; w < 4; ++w) if (rp[w] >= 0 && (fp < 0 || rs[w] > fs || (rs[w] == fs && rp[w] < fp))) { fs = rs[w]; fp = rp[w]; }
      rs[0] = fs;
      if (fp >= 0) sv[fp] = -INFINITY;
    }
    __syncthreads();
    kth = rs[0];
    __syncthreads();
  }
  if (tid == 0) tau[q] = (S >= k) ? kth : -INFINITY;
}

// exact top-k of a query's candidates (nstreams private lists of <= cap entries), (score desc, index asc).
// The lists are first compacted into LDS (LDS atomic slot counter), then k rounds of block-wide arg-best run on-chip.
constexpr int FS_MAX = 8192;                           // candidates per query held in LDS (64 KiB)
__global__ void __launch_bounds__(256) sim_final_select(const float* cand_s, const int* cand_i, const int* cnt, int nstreams, int cap,
                                                        int k, long long g_offset, float* out_s, long long* out_i) {
  extern __shared__ __attribute__((aligned(16))) char fsraw[];
  float* cs = (float*)fsraw; int* ci = (int*)(cs + FS_MAX);
  __shared__ float rs[4]; __shared__ int ri[4]; __shared__ int rp[4]; __shared__ int ovf, total;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { ovf = 0; total = 0; }
  __syncthreads();
  for (int st = tid; st < nstreams; st += 256) {
    const int c = cnt[(long)q * nstreams + st];
    if (c > cap) ovf = 1;
    const int n = min(c, cap);
    if (n > 0) {
      const int base = atomicAdd(&total, n);
      const long src = ((long)q * nstreams + st) * cap;
      for (int j = 0; j < n; ++j)
        if (base + j < FS_MAX) { cs[base + j] = cand_s[src + j]; ci[base + j] = cand_i[src + j]; }
    }
  }
  __syncthreads();
  if (total > FS_MAX) ovf = 1;
  const int n = min(total, FS_MAX);
  const bool overflow = ovf != 0;
  __syncthreads();
  for (int round = 0; round < k; ++round) {
    float bs = -INFINITY; int bi = INT_MAX, bp = -1;
    for (int i = tid; i < n; i += 256) {
      if (ci[i] == INT_MAX) continue;                  // already taken
      if (bp < 0 || cs[i] > bs || (cs[i] == bs && ci[i] < bi)) { bs = cs[i]; bi = ci[i]; bp = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64); const int oi = __shfl_xor(bi, o, 64), op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; bp = op; }
    }
    if (lane == 0) { rs[wave] = bs; ri[wave] = bi; rp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = rs[0]; int fi = ri[0], fp = rp[0];
      for (int w = 1; w < 4; ++w) if (rp[w] >= 0 && (fp < 0 || rs[w] > fs || (rs[w] == fs && ri[w] < fi))) { fs = rs[w]; fi = ri[w]; fp = rp[w]; }
      out_s[(long)q * k + round] = fp >= 0 ? fs : -INFINITY;
      out_i[(long)q * k + round] = overflow ? -2LL : (fp >= 0 ? (long long)fi + g_offset : -1LL);
      if (fp >= 0) ci[fp] = INT_MAX;
    }
    __syncthreads();
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

int g_topk_force_lists = 0;       // cor_topk_set_mode(1): always use the per-lane list kernels (fallback after an overflow)
constexpr int V3_MIN_ROWS = 32768;
struct V3Plan { int S, stride, cap, nsplit, tiles_per_split, nstreams; };
int device_cus();
inline V3Plan make_v3(int Bq, int Ng, int k, int) {
  V3Plan p;
  p.S = 4096;
  p.stride = Ng / p.S;                                 // strided sample: robust to ordered galleries
  const int nqg = cdiv(Bq, 256), tiles = cdiv(Ng, 32);
  int want = 2 * device_cus() / nqg;                   // two resident blocks per CU
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  p.tiles_per_split = cdiv(tiles, want);
  p.nsplit = cdiv(tiles, p.tiles_per_split);
  p.nstreams = 2 * p.nsplit;
  const long expect = (long)k * Ng / p.S / p.nstreams;  // expected accepted scores per private list
  p.cap = (int)(4 * expect + 32);
  return p;
}

int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

int launch_merge(const float* ws_s, const int* ws_i, int Bq, int n, int k, long long g_offset, float* out_s, long long* out_i, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sim_topk_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    attr_set = true;
  }
  hipLaunchKernelGGL(sim_topk_merge, dim3(Bq), dim3(256), (size_t)n * 8, s, ws_s, ws_i, n, k, g_offset, out_s, out_i);
  COR_CHECK_LAUNCH();
  return 0;
}

template <typename TG>
int launch_topk(const float* Q, const void* G, int Bq, int Ng, int C, int k, long long g_offset, float* out_s, long long* out_i,
                void* workspace, hipStream_t s) {
  float* ws_s = (float*)workspace;
  if (sizeof(TG) == 2 && C == 256 && Ng >= V3_MIN_ROWS && !g_topk_force_lists) {   // threshold-and-append (exact)
    const V3Plan p = make_v3(Bq, Ng, k, device_cus());
    const size_t lds = 4 * 32 * 256 * 2;                // NS ring slots
    static bool scan_attr = false;
    if (!scan_attr) {
      (void)hipFuncSetAttribute((const void*)sim_scan<TG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)sim_scan<TG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      scan_attr = true;
    }
    char* w = (char*)workspace;
    float* dense = (float*)w; w += (size_t)Bq * p.S * 4;
    float* tau = (float*)w; w += (size_t)Bq * 4;
    int* cnt = (int*)w; w += (size_t)Bq * p.nstreams * 4;
    float* cand_s = (float*)w; w += (size_t)Bq * p.nstreams * p.cap * 4;
    int* cand_i = (int*)w;
    ScanArgs a{};
    a.Bq = Bq; a.Ng = Ng; a.nqg = cdiv(Bq, 256);
    // 1. dense scores of a strided sample
    a.row_stride = p.stride; a.n_rows = p.S; a.dense = dense; a.dense_ld = p.S;
    { const int tiles = cdiv(p.S, 32); int want = device_cus() / a.nqg; if (want > tiles) want = tiles; if (want < 1) want = 1;
      a.tiles_per_split = cdiv(tiles, want); a.nsplit = cdiv(tiles, a.tiles_per_split); }
    hipLaunchKernelGGL((sim_scan<TG, true>), dim3(a.nqg * a.nsplit), dim3(256), lds, s, Q, (const TG*)G, a);
    COR_CHECK_LAUNCH();
    // 2. tau_q = k-th best sample score
    hipLaunchKernelGGL(sim_sample_tau, dim3(Bq), dim3(256), (size_t)p.S * 4, s, dense, (long)p.S, p.S, k, tau, cnt);
    COR_CHECK_LAUNCH();
    // 3. full scan: every (query, slice, lane half) stream keeps the scores >= tau_q in its private list
    a.row_stride = 1; a.n_rows = Ng; a.tau = tau; a.cnt = cnt; a.cand_s = cand_s; a.cand_i = cand_i; a.cap = p.cap;
    a.tiles_per_split = p.tiles_per_split; a.nsplit = p.nsplit;
    hipLaunchKernelGGL((sim_scan<TG, false>), dim3(a.nqg * a.nsplit), dim3(256), lds, s, Q, (const TG*)G, a);
    COR_CHECK_LAUNCH();
    // 4. exact selection
    static bool fs_attr = false;
    if (!fs_attr) {
      (void)hipFuncSetAttribute((const void*)sim_final_select, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX * 8);
      fs_attr = true;
    }
    hipLaunchKernelGGL(sim_final_select, dim3(Bq), dim3(256), FS_MAX * 8, s, cand_s, cand_i, cnt, p.nstreams, p.cap, k, g_offset, out_s, out_i);
    COR_CHECK_LAUNCH();
    return 0;
  }
  if (sizeof(TG) == 2 && C == 256) {                   // MFMA-bound form
    const TopkPlan2 p = make_plan2(Bq, Ng, k, device_cus());
    int* ws_i = (int*)(ws_s + (long)Bq * p.nparts * p.kmax);
    const dim3 grid(p.nqg * p.nsplit), block(256);
    const size_t lds = 2 * 32 * 256 * 2;
#define SIM_V2(KM) hipLaunchKernelGGL((sim_topk_v2<TG, KM>), grid, block, lds, s, Q, (const TG*)G, Bq, Ng, p, ws_s, ws_i)
    if (p.kmax == 8) SIM_V2(8); else if (p.kmax == 16) SIM_V2(16); else SIM_V2(32);
#undef SIM_V2
    COR_CHECK_LAUNCH();
    return launch_merge(ws_s, ws_i, Bq, p.nparts * p.kmax, k, g_offset, out_s, out_i, s);
  }
  const TopkPlan p = make_plan(Bq, Ng, k);
  int* ws_i = (int*)(ws_s + (long)Bq * p.nparts * p.kmax);
  const int nwaves = p.nqt * p.nsplit;
  if (p.kmax == 8) hipLaunchKernelGGL((sim_topk_partial<TG, 8>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  else hipLaunchKernelGGL((sim_topk_partial<TG, 32>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  COR_CHECK_LAUNCH();
  return launch_merge(ws_s, ws_i, Bq, p.nparts * p.kmax, k, g_offset, out_s, out_i, s);
}

}  // namespace

extern "C" int cor_topk_set_mode(int force_lists) {
  g_topk_force_lists = force_lists ? 1 : 0;
  return 0;
}

extern "C" long cor_topk_workspace_bytes(int Bq, int Ng, int k) {
  if (Bq <= 0 || Ng <= 0 || k <= 0 || k > 32) return COR_EINVAL;
  const TopkPlan p = make_plan(Bq, Ng, k);
  const TopkPlan2 p2 = make_plan2(Bq, Ng, k, device_cus());
  const long a = (long)Bq * p.nparts * p.kmax * 8, b = (long)Bq * p2.nparts * p2.kmax * 8;
  long m = a > b ? a : b;
  if (Ng >= V3_MIN_ROWS) {
    const V3Plan v = make_v3(Bq, Ng, k, 0);
    const long c = (long)Bq * v.S * 4 + (long)Bq * 4 + (long)Bq * v.nstreams * 4 + (long)Bq * v.nstreams * v.cap * 8;
    if (c > m) m = c;
  }
  return m;
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
