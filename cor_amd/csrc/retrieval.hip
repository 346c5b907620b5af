// cor_amd — gallery similarity + top-k for gfx950 (the retrieval end of the path).
//
//   score[b,g] = Q[b,:] . G[g,:]   (unit vectors: cosine, utils/loss_func.py:84)  ->  per query top-k by
//   (score desc, global index asc).
//
// The [Bq, Ng] score matrix is never written to HBM. 16-bit galleries with C = 256 (the CORE embedding width) take the
// threshold-and-append pipeline further down (sim_scan: MFMA scan + exact fp32-chain re-scoring of the short list =>
// top-k indices and scores BIT-IDENTICAL to the CPU chain oracle); what follows first are the per-lane sorted-list kernels
// (fp32 galleries: the exact-chain form; 16-bit: the always-available fallback). One WAVE owns 32 queries (their K-fragments stay in
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
__device__ __forceinline__ uint4 q_frag16_vals(const f32x4 lo, const f32x4 hi) {       // 8 fp32 values -> one 16-byte K-fragment piece
  if (__is_same(TG, bf16_t)) {
    uint4 u;
    u.x = (uint32_t)f2bf(lo[0]) | ((uint32_t)f2bf(lo[1]) << 16); u.y = (uint32_t)f2bf(lo[2]) | ((uint32_t)f2bf(lo[3]) << 16);
    u.z = (uint32_t)f2bf(hi[0]) | ((uint32_t)f2bf(hi[1]) << 16); u.w = (uint32_t)f2bf(hi[2]) | ((uint32_t)f2bf(hi[3]) << 16);
    return u;
  }
  f16x8 t = {(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
  return __builtin_bit_cast(uint4, t);
}
template <typename TG>
__device__ __forceinline__ uint4 q_frag16(const float* qrow, int c, int h) {
  return q_frag16_vals<TG>(*(const f32x4*)(qrow + c * 16 + 8 * h), *(const f32x4*)(qrow + c * 16 + 8 * h + 4));
}

template <typename TG, int KMAX>
__global__ void __launch_bounds__(256, 1) sim_topk_v2(const float* __restrict__ Q, const TG* __restrict__ G, int Bq, int Ng,
                                                      TopkPlan2 plan, float* ws_s, int* ws_i, const int* gate) {
  if (gate && *gate == 0) return;                               // device-side fallback: runs only after a candidate overflow
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

// ---------------------------------------------------------------------------------------------------------------------
// v3 (16-bit galleries, C = 256, every shard size): threshold-and-append with EXACT re-scoring.
// Per-lane sorted lists cost one divergent insertion bubble (~500 wave cycles) per accepted score, and a stream of n scores
// accepts ~k ln(n/k) of them per lane: measured 9x the MFMA time. Instead, four launches (sim_prep + these three):
//   A. sim_scan<SAMPLE>: MFMA scores of a strided SAMPLE of 32-row tiles; every (gallery slice, lane quarter) group keeps only
//      its MAXIMUM per query and folds it (atomic max on order-preserving keys) into one of 32 SUPER-GROUPS per query. The
//      super-groups are disjoint row sets, so the k-th largest of their maxima is a LOWER bound of the query's k-th best score over
//      the whole shard (no dense score matrix, no selection pass).
//   B. tau_q = that k-th largest maximum minus delta_q: ranked in the PROLOGUE of pass C (thread <-> query, a 32-element bitonic
//      network in registers). Rounds 2-3 ranked all 512 group maxima in a launch of their own (sim_tau, 5-7 us, launch-bound); the
//      32 unions admit ~15 % more candidates and cost the prologue ~3 us: one launch fewer, 0-2 % faster in a same-box A/B.
//   C. sim_scan<APPEND>: MFMA scores of the WHOLE shard; a lane compares its tile maximum with tau_q and appends the rare
//      scores >= tau_q to the private list of its (query, gallery slice, lane quarter) stream (register counter, no atomics).
//   D. sim_final: per query, the candidates are compacted into LDS, the k-th best MFMA score T is bounded by a 2-pass radix
//      select (16 key bits), the SHORT LIST (MFMA score >= T - delta_q, ~k entries) is RE-SCORED with the exact fp32 fmaf chain of
//      oracle/c/sim_chain.c over the stored 16-bit values (the query rounded to the gallery dtype), and ranked by
//      (chain score desc, index asc).
// Exactness. Let s~ be the MFMA score (fp32 accumulation of exact bf16/fp16 products in the matrix core's order) and s the
// chain score. Both are fp32 summations of the same 256 exact products, so |s~ - s| <= eps_q := 2^-23 * 255/... bounded by
// 2 * 255 * 2^-24 * sum|q_k g_k| <= 3.1e-5 * |q| |g| (one rounding per addition either way; doubled for a truncating
// adder). If g is in the chain top-k then s~_g >= T - 2 eps_q (the k rows of the MFMA top-k have chain scores
// >= T - eps_q, hence the k-th best chain score is >= T - eps_q), so with delta_q = 6.4e-5 * max(1, |q|) * 1.01 (gallery rows
// are unit vectors: cosine similarity) the short list contains every row of the exact answer, tau_q - delta_q admits all
// of them in pass C, and the output (scores AND indices) is bit-identical to ranking the chain scores of the whole shard.
// Overflows (a stream list, the LDS candidate buffer or the short list: pathological score distributions such as tens of
// thousands of identical rows) are detected ON THE DEVICE: sim_final flags the query and the gated list kernels
// (sim_topk_v2 + sim_topk_merge, launched behind it, a few microseconds when idle) recompute exactly those queries:
// no host synchronisation, no process-global mode.
constexpr float SIM_DELTA = 6.4e-5f * 1.01f;
constexpr int FS_MAX = 8192;                           // candidates per query held in LDS by sim_final (64 KiB)
constexpr int SL_MAX = 512;                            // short list (re-scored exactly)
constexpr int SCAN_NS = 5, SCAN_AHEAD = 4;             // LDS-DMA ring of 32-KiB gallery super-tiles (64 rows): 160 KiB = all of LDS, one block per CU

struct ScanArgs {
  int Bq, Ng, nqg, nsplit, tiles_per_split;
  int ntiles;                      // 64-row super-tiles this launch walks (SAMPLE: sample super-tiles; APPEND: all of them)
  int tile_stride;                 // super-tiles between consecutive walked super-tiles (SAMPLE: >= 1; APPEND: 1)
  const uint4* qimg;               // queries rounded to the gallery dtype, fragment-major (sim_prep)
  unsigned* sg; int Bqp;           // 32 SUPER-GROUP maxima per query as order-preserving keys, sg[g * Bqp + q] (zeroed by sim_prep; 0 = empty):
                                   // SAMPLE: atomic max of the group (slice, lane quarter) maximum into super-group (split * 4 + rq) & 31; APPEND: read
  const float* dq; int k;          // APPEND: delta_q (sim_prep) and k: tau_q = k-th largest super-group maximum - delta_q, computed in the prologue
  float tau_add;                   // 0; timing-only ablation (COR_TOPK_DEBUG_NOCAND): +1e30 = no candidate ever passes
  int probe_same;                  // COR_PROBES (timing only): every block streams the same 8 super-tiles (cache-resident gallery)
  unsigned long long* stamps;      // COR_PROBES: cycle stamps of waves 0 and 4 of block 0 (tools/sim_stamps.py scan)
  float* tau; int* cnt; float* rec_s; int* rec_g; int cap;           // APPEND: tau_q (written by the blocks of slice 0 for the selection kernel); record i of stream (q, slice, lane quarter): 8 scores + first row
};

// One block = 8 waves = 256 * QB queries (wave w owns queries q0 + 32 * QB * w ..): with QB = 2 all 512 queries of an
// 8-GPU all-gather (8 x 64) sit in ONE block, so every gallery byte is fetched from HBM once per launch (512 MB at 1M rows)
// and the kernel is MFMA-bound (arithmetic intensity = queries per block FLOP/B against a ridge of ~400); the K-fragments of
// the wave's queries stay in registers for the whole kernel (QB * 64 VGPRs) and each A fragment read from LDS feeds 2 QB MFMAs
// (v_mfma_f32_16x16x32 since round 5: see the kernel).
// Gallery super-tiles (64 rows x 512 B = two 32-row tiles) stream through a 5-slot LDS-DMA ring, 4 of them (128 KiB per CU) in
// flight: counted s_waitcnt vmcnt, ONE barrier per 64 rows. The LDS image is lane-linear per wave-instruction (64 lanes x 16 B = two 512-B
// rows), so slot (row, sl) is fed from source chunk sl ^ (row & 15) and the reads apply the same XOR (conflict-free b128).
// max of three. NOT inline asm: hipcc pads no MFMA -> VALU read hazard for an asm statement, so a v_max3_f32 in asm read the
// accumulators before the matrix pipe had written them (sporadic missed candidates). The file is built with -fno-honor-nans
// (Makefile) so that fmaxf needs no canonicalising v_max x,x,x per operand and folds to v_max3_f32 by itself.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// order-preserving float <-> unsigned keys (0 is below every float's key: "empty")
__device__ __forceinline__ unsigned f2key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f_floor(unsigned key) {      // the smallest float whose order-preserving key is >= key's prefix
  const unsigned u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
  return __uint_as_float(u);
}

template <typename TG, int QB, bool SAMPLE>
__global__ void __launch_bounds__(512, 2) sim_scan(const TG* __restrict__ G, const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 256, TILE = 32 * C * 2, STILE = 2 * TILE, NS = SCAN_NS, AHEAD = SCAN_AHEAD;
  // Round 5: v_mfma_f32_16x16x32 instead of 32x32x16. A wave's 32 rows x 32 QB queries are 2 x NQ blocks of 16 x 16 = 2 NQ INDEPENDENT
  // accumulators of four registers (the 32 x 32 form had QB = 2 dependent chains: a wave alone issued an MFMA every ~43 cycles instead of 32),
  // and the chip holds a higher clock on this shape (MI355X guide, DVFS item 7): 512 x 1M 280-300 -> 185-200 us in the timing probe. Operand
  // lane (n16 = lane & 15: gallery row of a 16-row block / query of a 16-query block, rq = lane >> 4: K quarter of a 32-deep step);
  // accumulator lane: query n16, rows 4 rq + e of the row block. Same LDS image, same number of ds_read_b128, same register count.
  constexpr int NQ = 2 * QB;                           // 16-query blocks per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n16 = lane & 15, rq = lane >> 4;
  const int qg = blockIdx.x % a.nqg, split = blockIdx.x / a.nqg;
  const int q0 = qg * (256 * QB) + wave * (32 * QB);
  const bool active = q0 < a.Bq;                     // wave-uniform: idle waves still stage and barrier
  // Waves w and w + 4 share a SIMD and, running the same program between the same barriers, would reach their MFMAs, their
  // VALU epilogues and the barrier TOGETHER: the matrix pipe then idles through every epilogue. The younger half (waves 4-7)
  // therefore runs the epilogue of a tile AFTER the next barrier, in front of its next MFMAs, i.e. beside the MFMAs of its
  // SIMD partner (MI355X guide, "two waves per SIMD", item 9); the accumulators are not overwritten until then, so no
  // double buffering is needed.
  const bool late = wave >= 4;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  // A super-tile = 64 rows x 512 B; wave w copies rows 8w .. 8w+7 with FOUR LDS-DMA instructions of two rows each under ONE M0 setting: the
  // instruction offsets 0 / 1024 / 2048 / 3072 move the LDS and the global address alike (rows are contiguous in both), the per-lane
  // offset carries the row of the pair and the swizzled chunk. (Round 5 stamps: per-lane 64-bit addresses + an M0 save / set / restore per
  // instruction cost a wave 520-630 cycles per super-tile, both waves of a SIMD at the same time.)
  const unsigned wbase = (unsigned)wave * 4096u;       // this wave's first row (bytes) in a ring slot
  const int st_e = lane >> 5, st_sl = lane & 31;
  unsigned st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) st_off[i] = (unsigned)(st_e * 512 + ((st_sl ^ ((8 * wave + 2 * i + st_e) & 15)) << 4));
  const int t0 = split * a.tiles_per_split, t1 = min(t0 + a.tiles_per_split, a.ntiles);     // super-tiles of 64 rows
  auto issue = [&](int t) {
#ifdef COR_PROBES
    const long g0 = (long)((a.probe_same & 1) ? (t & 7) : t) * a.tile_stride * 64;
#else
    const long g0 = (long)t * a.tile_stride * 64;
#endif
    const unsigned dst = lds0 + ((t - t0) % NS) * STILE + wbase;
    if (g0 + 64 <= a.Ng) {                             // (scalar) every row of the super-tile is inside the shard
      const char* sb = (const char*)G + (g0 + 8 * wave) * 512;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                   "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(st_off[0]), "v"(st_off[1]), "v"(st_off[2]), "v"(st_off[3]), "s"(sb), "s"(dst) : "memory");
    } else {                                           // the shard's last super-tile: clamped duplicates (masked to -inf in the epilogue)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * wave + 2 * i + st_e;
        glds16(G + min(g0 + row, (long)a.Ng - 1) * C + ((st_sl ^ (row & 15)) << 3), dst + 1024 * i);
      }
    }
  };
  // The ring is filled FIRST: the HBM round trip of the first super-tiles runs under the prologue below (threshold network, 32 KiB of
  // K-fragments per wave); tau_s sits in the slot these copies do not touch.
  for (int t = t0; t < min(t0 + AHEAD, t1); ++t) issue(t);

  // APPEND: tau_q in the prologue (round 4: no sim_tau launch). Thread <-> query of this block: its 32 super-group maxima, a 32-element
  // bitonic network in registers (the K-fragments are not loaded yet), tau_q = k-th largest - delta_q (fewer than k non-empty super-groups:
  // -inf, every row is a candidate). Any lower bound of the k-th best score is valid, so 32 unions of the (slice, lane quarter) groups serve
  // as well as the 512 groups the launch ranked (~15 % more candidates).
  float* tau_s = (float*)(smem + (SCAN_NS - 1) * 2 * 32 * 256 * 2);   // in the LAST ring slot: first written by the copies behind the loop's first barrier
  if (!SAMPLE) {
    const int ql = tid;                                // 256 * QB queries per block, 512 threads
    if (ql < 256 * QB) {
      const int qg_ = min(qg * (256 * QB) + ql, a.Bq - 1);
      float v[32];
#pragma unroll
      for (int g = 0; g < 32; ++g) { const unsigned key = a.sg[(long)g * a.Bqp + qg_]; v[g] = key == 0u ? -INFINITY : key2f_floor(key); }
#pragma unroll
      for (int kk = 2; kk <= 32; kk <<= 1)
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            const int l = i ^ j;
            if (l > i) {
              const float x = v[i], y = v[l];
              if ((i & kk) == 0) { v[i] = fmaxf(x, y); v[l] = fminf(x, y); }
              else               { v[i] = fminf(x, y); v[l] = fmaxf(x, y); }
            }
          }
      float kth = v[0];
#pragma unroll
      for (int c = 1; c < 32; ++c) kth = (c == a.k - 1) ? v[c] : kth;
      const float t = kth - a.dq[qg_];                 // -inf stays -inf
      tau_s[ql] = t;
      if (split == 0 && qg * (256 * QB) + ql < a.Bq) a.tau[qg * (256 * QB) + ql] = t;      // for the selection kernel
    }
    __syncthreads();
  }
  uint4 qf[NQ][8];
  float tau[NQ], gmax[NQ];
  int ncand[NQ];
  const int nstreams = a.nsplit * 4;                   // a stream = (query, gallery slice, lane quarter rq)
#pragma unroll
  for (int qb = 0; qb < NQ; ++qb) {
    // K-fragments from the fragment-major image sim_prep wrote (16-query block, 32-deep K-step, lane) x 16 B: 1 KiB per wave-instruction,
    // 8 of them per query block (reading the fp32 rows cost every CU 512 KiB of L2 traffic per launch: ~7 us)
    const uint4* qimg = a.qimg + ((long)(q0 / 16 + qb) * 8) * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[qb][kk] = active ? qimg[kk * 64] : make_uint4(0, 0, 0, 0);
    tau[qb] = SAMPLE ? 0.f : tau_s[wave * (32 * QB) + qb * 16 + n16] + a.tau_add;
    gmax[qb] = -INFINITY; ncand[qb] = 0;
  }
  // read address of 32-deep K-step kk, row block rb: row rb * 16 + n16, chunk (4 kk + rq) ^ n16 = ((kk ^ (n16 >> 2)) << 2) | (rq ^ (n16 & 3)): one XOR
  // per read. Conflict-free ds_read_b128: every 16-lane service group holds 16 different n16 (the two K quarters it mixes map onto disjoint slots).
  const int rd_base = n16 * 512 + ((rq ^ (n16 & 3)) << 4), rd_x = n16 >> 2;
#define SIM_RD(rb_, kk_) (rd_base + (rb_) * 8192 + ((((kk_) ^ rd_x)) << 6))

  f32x4 acc[NQ][2];                                    // [16-query block][16-row block]: register e = row 4 rq + e of the row block
  // epilogue of one 32-row tile (rows g0 ..): SAMPLE keeps the group maximum; APPEND compares the tile maximum with tau and
  // appends the rare scores >= tau to the lane's private stream list
  const int r4 = 4 * rq;
  auto epilogue = [&](int g0) {
    const int lim = a.Ng - g0;                         // rows of this tile inside the shard (scalar); < 32 only at the shard's end
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      if (lim < 32) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (16 * rb + r4 + e >= lim) acc[qb][rb][e] = -INFINITY;                 // clamped duplicate rows never count
      }
      float tmax = max3f(acc[qb][0][0], acc[qb][0][1], acc[qb][0][2]);
      tmax = max3f(tmax, acc[qb][0][3], acc[qb][1][0]);
      tmax = max3f(tmax, acc[qb][1][1], acc[qb][1][2]);
      tmax = fmaxf(tmax, acc[qb][1][3]);
      if (SAMPLE) {
        gmax[qb] = fmaxf(gmax[qb], tmax);
      } else if (__builtin_amdgcn_ballot_w64(tmax >= tau[qb]) != 0) {
        // Some lane of the wave has a candidate in this tile. Testing the registers one by one cost ~80 instructions per triggered tile
        // (70 us of a 270-us scan in round 2); instead a lane whose tile maximum passes appends its WHOLE 8-score column as one record
        // (2 x 16-byte stores + the tile's first row) to its private stream list, and sim_final filters the records against tau.
        int q = q0 + qb * 16 + n16;
        asm volatile("" : "+v"(q));                    // opaque: keeps the list address arithmetic HERE (hoisted out of the tile loop
        if (tmax >= tau[qb] && q < a.Bq) {             // it was spilled, and a spill reload inside the loop drains the DMA ring)
          if (ncand[qb] < a.cap) {
            const long rec = ((long)q * nstreams + split * 4 + rq) * a.cap + ncand[qb];
            f32x4* dst = (f32x4*)(a.rec_s + rec * 8);
            dst[0] = acc[qb][0]; dst[1] = acc[qb][1];
            a.rec_g[rec] = g0;
          }
          ++ncand[qb];
        }
      }
    }
  };
  auto mfma_tile = [&](const char* buf) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};           // C operand of the first K-step: the inline constant 0 (no v_mov per tile)
    uint4 af[4];                                       // A fragments two K-steps (four reads) ahead of the MFMAs that consume them
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const uint4*)(buf + SIM_RD(i & 1, i >> 1));
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const uint4 av = af[(2 * kk + rb) & 3];
        if (kk + 2 < 8) af[(2 * kk + rb) & 3] = *(const uint4*)(buf + SIM_RD(rb, kk + 2));
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
          if (__is_same(TG, bf16_t))
            acc[qb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, qf[qb][kk]), kk == 0 ? zero : acc[qb][rb], 0, 0, 0);
          else
            acc[qb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, qf[qb][kk]), kk == 0 ? zero : acc[qb][rb], 0, 0, 0);
        }
      }
  };

  // (s_setprio 1 for the younger half - waves 4-7 - before the loop, the guide's static-priority item, and a sample stride of 32 instead of 16
  // at 1M rows: no difference in a same-box A/B on the round-4 kernel. Round 5, measured with cycle stamps: priority 2 for the younger half
  // during its FIRST tile only balances the two halves, below.)
  int g_pending = -1;                                  // late waves: tile whose epilogue is still owed
#ifdef COR_PROBES
#define SC_STAMP(i_) do { if (!SAMPLE && a.stamps && blockIdx.x == 0 && (tid & 255) == 0 && t - t0 < 24) { __builtin_amdgcn_sched_barrier(0); \
    a.stamps[((tid >> 8) * 24 + (t - t0)) * 8 + (i_)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SC_STAMP(i_) do { } while (0)
#endif
  for (int t = t0; t < t1; ++t) {
    // super-tile t has landed once at most 4 * (super-tiles issued after t) of this wave's DMA are still outstanding (VMEM
    // retires in order; candidate stores issued in between only make the wait stricter)
    // Round 5 (cycle stamps, tools/sim_stamps.py): VMEM retires in order, so the ~5 candidate stores of a record appended in the last period
    // count among "the 4 (AHEAD - 1) youngest operations" and the wait below then also waits for copies issued one period ago (500-1 000
    // cycles per super-tile with a ring of 4). A ring of FIVE (all 160 KiB of LDS; tau_s lives in the last slot until the first barrier) puts
    // one more period between a copy's issue and the wait that can be forced onto it.
    const int later = min(t1 - 1 - t, AHEAD - 1);
    if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SC_STAMP(0);
    __builtin_amdgcn_s_barrier();                      // every wave's part of super-tile t is in LDS; t-1 is fully consumed
    SC_STAMP(1);
    if (t + AHEAD < t1) issue(t + AHEAD);              // -> ring slot of super-tile t-1
#ifdef COR_PROBES
    if (a.probe_same & 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // probe: how deep is the backlog? (everything but the newest super-tile)
    if (a.probe_same & 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    SC_STAMP(2);
    const char* buf = smem + ((t - t0) % NS) * STILE;
    const int g0 = t * a.tile_stride * 64;
    if (active) {
      if (late && g_pending >= 0) epilogue(g_pending);
      SC_STAMP(3);
      if (late) __builtin_amdgcn_s_setprio(2);
      mfma_tile(buf);
      if (late) __builtin_amdgcn_s_setprio(0);
      SC_STAMP(4);
      epilogue(g0);
      SC_STAMP(5);
      mfma_tile(buf + TILE);
      SC_STAMP(6);
      if (late) g_pending = g0 + 32; else epilogue(g0 + 32);
      SC_STAMP(7);
    }
  }
  if (active && late && g_pending >= 0) epilogue(g_pending);
#undef SIM_RD
  if (active) {
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const int q = q0 + qb * 16 + n16;
      if (q < a.Bq) {
        if (SAMPLE) atomicMax(a.sg + (long)((split * 4 + rq) & 31) * a.Bqp + q, f2key(gmax[qb]));    // (a group without a tile: key(-inf) > 0 = empty)
        else a.cnt[(long)q * nstreams + split * 4 + rq] = ncand[qb];
      }
    }
  }
}

// Queries rounded to the gallery dtype in the fragment order of v_mfma_f32_16x16x32: image[(qblk16 * 8 + kk) * 64 + lane] = 16 B =
// q[qblk16*16 + (lane&15)][32 kk + 8 (lane>>4) .. +8]; rows beyond Bq repeat the last query (their results are never written). Also
// clears the per-call overflow flags. One block of 256 threads per 32 queries.
template <typename TG>
__global__ void __launch_bounds__(256) sim_prep(const float* __restrict__ Q, int Bq, uint4* img, int* flags, int* ovf_q, unsigned* sg, int Bqp,
                                                float* dq) {
  __shared__ float part[32][33];                       // |q|^2 partials of the block's 32 queries: [query][K-step * 2 + lane half]
  const int qblk = blockIdx.x, tid = threadIdx.x;
  if (qblk == 0 && tid == 0) flags[0] = 0;
  if (tid < 32 && qblk * 32 + tid < Bq) ovf_q[qblk * 32 + tid] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) sg[(long)(tid >> 5 | (i << 3)) * Bqp + qblk * 32 + (tid & 31)] = 0u;     // the 32 super-group maxima of these queries: empty
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int item = tid + 256 * i, c = item >> 6, lane = item & 63, r = lane & 31, h = lane >> 5;
    const float* qrow = Q + (long)min(qblk * 32 + r, Bq - 1) * 256;
    const uint4 f = q_frag16<TG>(qrow, c, h);           // chunk ch = 2 c + h of the query: values 8 ch .. 8 ch + 7
    // v_mfma_f32_16x16x32 fragment order: image[(qblk16 * 8 + kk) * 64 + lane'] with lane' = 16 * (ch & 3) + (query & 15), kk = ch >> 2
    { const int ch = 2 * c + h; img[((long)(qblk * 2 + (r >> 4)) * 8 + (ch >> 2)) * 64 + 16 * (ch & 3) + (r & 15)] = f; }
    float n2 = 0.f;                                    // of the values as the MFMA sees them
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo, hi;
      if (__is_same(TG, bf16_t)) { lo = __uint_as_float(w[j] << 16); hi = __uint_as_float(w[j] & 0xffff0000u); }
      else { typedef _Float16 h2 __attribute__((ext_vector_type(2))); const h2 t2 = __builtin_bit_cast(h2, w[j]); lo = (float)t2[0]; hi = (float)t2[1]; }
      n2 = fmaf(lo, lo, n2); n2 = fmaf(hi, hi, n2);
    }
    part[r][c * 2 + h] = n2;
  }
  __syncthreads();
  if (tid < 32 && qblk * 32 + tid < Bq) {
    float n2 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) n2 += part[tid][j];   // fixed order: delta_q is reproducible
    dq[qblk * 32 + tid] = SIM_DELTA * fmaxf(1.f, sqrtf(n2));
  }
}

template <typename TG> __device__ __forceinline__ float round_to(float x);
template <> __device__ __forceinline__ float round_to<bf16_t>(float x) { return bf2f(f2bf(x)); }
template <> __device__ __forceinline__ float round_to<_Float16>(float x) { return (float)(_Float16)x; }

// chain score of gallery row `idx` against the query in LDS (oracle/c/sim_chain.c order: chunk c of 8: k = 8c+i then 8c+4+i)
template <typename TG>
__device__ __forceinline__ float chain_score(const TG* __restrict__ G, long idx, const float* qs) {
  const uint4* row = (const uint4*)(G + idx * 256);
  float acc = 0.f;
#pragma unroll 4
  for (int c = 0; c < 32; ++c) {
    const uint4 v = row[c];
    float g[8];
    if (__is_same(TG, bf16_t)) {
      g[0] = __uint_as_float(v.x << 16); g[1] = __uint_as_float(v.x & 0xffff0000u); g[2] = __uint_as_float(v.y << 16); g[3] = __uint_as_float(v.y & 0xffff0000u);
      g[4] = __uint_as_float(v.z << 16); g[5] = __uint_as_float(v.z & 0xffff0000u); g[6] = __uint_as_float(v.w << 16); g[7] = __uint_as_float(v.w & 0xffff0000u);
    } else {
      const f16x8 hv = __builtin_bit_cast(f16x8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = (float)hv[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = fmaf(g[i], qs[8 * c + i], acc);
      acc = fmaf(g[4 + i], qs[8 * c + 4 + i], acc);
    }
  }
  return acc;
}

// Exact fallback of ONE query (block of 256 threads): every row's chain score, per-thread sorted top-k lists in LDS (thread t
// owns slots [t*k, t*k + k) of cs/ci: 256 * k <= FS_MAX), then k rounds of block-wide arg-best by (score desc, index asc).
template <typename TG, int NT = 256>
__device__ void brute_force_topk(const TG* __restrict__ G, int Ng, const float* qs, float* cs, int* ci, int k, long long g_offset,
                                 float* out_s, long long* out_i) {
  constexpr int NW = NT / 64;
  __shared__ float rs[NW]; __shared__ int ri[NW], rp[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, base = tid * k;
  for (int j = 0; j < k; ++j) { cs[base + j] = -INFINITY; ci[base + j] = INT_MAX; }
  for (int g = tid; g < Ng; g += NT) {                  // ascending rows: a tie keeps the earlier (smaller) index ahead
    const float sc = chain_score<TG>(G, g, qs);
    if (sc > cs[base + k - 1]) {
      int j = k - 1;
      while (j > 0 && sc > cs[base + j - 1]) { cs[base + j] = cs[base + j - 1]; ci[base + j] = ci[base + j - 1]; --j; }
      cs[base + j] = sc; ci[base + j] = g;
    }
  }
  __syncthreads();
  const int n = NT * k;
  for (int round = 0; round < k; ++round) {
    float bs = -INFINITY; int bi = INT_MAX, bp = -1;
    for (int i = tid; i < n; i += NT)
      if (ci[i] != INT_MAX && (bp < 0 || cs[i] > bs || (cs[i] == bs && ci[i] < bi))) { bs = cs[i]; bi = ci[i]; bp = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64); const int oi = __shfl_xor(bi, o, 64), op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; bp = op; }
    }
    if (lane == 0) { rs[wave] = bs; ri[wave] = bi; rp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = rs[0]; int fi = ri[0], fp = rp[0];
      for (int w = 1; w < NW; ++w) if (rp[w] >= 0 && (fp < 0 || rs[w] > fs || (rs[w] == fs && ri[w] < fi))) { fs = rs[w]; fi = ri[w]; fp = rp[w]; }
      out_s[round] = fp >= 0 ? fs : -INFINITY;
      out_i[round] = fp >= 0 ? (long long)fi + g_offset : -1LL;
      if (fp >= 0) ci[fp] = INT_MAX;
    }
    __syncthreads();
  }
}

// chain score of a gallery row staged in LDS (32 chunks of 16 B; chunk c sits in slot c ^ x): the order of oracle/c/sim_chain.c
template <typename TG>
__device__ __forceinline__ float chain_score_lds(const char* row, int x, const float* qs) {
  float acc = 0.f;
#pragma unroll 4
  for (int c = 0; c < 32; ++c) {
    const uint4 v = *(const uint4*)(row + ((c ^ x) << 4));
    const f32x4 qa = *(const f32x4*)(qs + 8 * c), qb = *(const f32x4*)(qs + 8 * c + 4);
    float g[8];
    if (__is_same(TG, bf16_t)) {
      g[0] = __uint_as_float(v.x << 16); g[1] = __uint_as_float(v.x & 0xffff0000u); g[2] = __uint_as_float(v.y << 16); g[3] = __uint_as_float(v.y & 0xffff0000u);
      g[4] = __uint_as_float(v.z << 16); g[5] = __uint_as_float(v.z & 0xffff0000u); g[6] = __uint_as_float(v.w << 16); g[7] = __uint_as_float(v.w & 0xffff0000u);
    } else {
      const f16x8 hv = __builtin_bit_cast(f16x8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = (float)hv[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = fmaf(g[i], qa[i], acc);
      acc = fmaf(g[4 + i], qb[i], acc);
    }
  }
  return acc;
}

// D. exact top-k of one query's candidates. LDS: cs/ci [FS_MAX] | sl_s/sl_i [SL_MAX] | qs[256] | hist[256].
template <typename TG>
__global__ void __launch_bounds__(256) sim_final(const float* __restrict__ Q, const TG* __restrict__ G, const float* rec_s, const int* rec_g,
                                                 const float* tau, const int* cnt, int nstreams, int cap, int Ng, int k, long long g_offset,
                                                 float* out_s, long long* out_i, int* flags, int* ovf_q, int no_fallback) {
  extern __shared__ __attribute__((aligned(16))) char fsraw[];
  float* cs = (float*)fsraw; int* ci = (int*)(cs + FS_MAX);
  float* sl_s = (float*)(ci + FS_MAX); int* sl_i = (int*)(sl_s + SL_MAX);
  float* qs = (float*)(sl_i + SL_MAX); int* hist = (int*)(qs + 256);
  __shared__ int ovf, total, nsl, sel_bin, k_rem;
  __shared__ float qn2[4];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { ovf = 0; total = 0; nsl = 0; }
  const float qv = round_to<TG>(Q[(long)q * 256 + tid]);          // the query as the MFMA saw it
  qs[tid] = qv;
  const float part = wave_sum(qv * qv);
  if (lane == 0) qn2[wave] = part;
  __syncthreads();
  const float delta = SIM_DELTA * fmaxf(1.f, sqrtf(qn2[0] + qn2[1] + qn2[2] + qn2[3]));
  // 1. the private stream lists hold whole 8-score columns (records: two 16-row blocks x four rows of one lane quarter): keep the scores
  // >= tau_q (the admission threshold of the scan, delta already subtracted) and compact them into LDS. Score [g][i] of lane quarter rq is
  // row g0 + 16 g + 4 rq + i.
  const float tq = tau[q];
  auto take = [&](const f32x4 (&v)[2], int g0, int r4) {            // ONE returning LDS atomic per record (round 2: one per passing score,
    int np = 0;                                                       // up to 16 dependent LDS round trips per record)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) np += (v[g][i] >= tq && v[g][i] > -INFINITY) ? 1 : 0;
    if (np == 0) return;
    int p = atomicAdd(&total, np);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (v[g][i] >= tq && v[g][i] > -INFINITY) {
          if (p < FS_MAX) { cs[p] = v[g][i]; ci[p] = g0 + i + 16 * g + r4; }
          ++p;
        }
  };
  // Two round trips: the thread's (up to four: 1024 streams) COUNTS first - contiguous per query, coalesced - then record 0 of its non-empty
  // streams, all issued together. (Rounds 3-4 fetched record 0 WITH the count, one trip: with 1000 mostly empty streams per query that is
  // 2 000 scattered 32-byte sectors per query, 64 MB per 512-query search: 5.5 us slower than this at 32k-125k rows.)
  for (int st0 = 0; st0 < nstreams; st0 += 1024) {
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = st0 + 256 * u + tid;
      cc[u] = st < nstreams ? cnt[(long)q * nstreams + st] : 0;
    }
    f32x4 v0[4][2]; int g00[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = st0 + 256 * u + tid;
      if (cc[u] > 0) {
        const long rec0 = ((long)q * nstreams + st) * cap;
        const f32x4* src = (const f32x4*)(rec_s + rec0 * 8);
        v0[u][0] = src[0]; v0[u][1] = src[1];
        g00[u] = rec_g[rec0];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = st0 + 256 * u + tid;
      if (cc[u] > cap) ovf = 1;
      const int n = min(cc[u], cap), r4 = 4 * (st & 3);
      if (n > 0) take(v0[u], g00[u], r4);
      if (n > 1) {
        const long rec0 = ((long)q * nstreams + st) * cap;
        for (int j = 1; j < n; ++j) {
          const f32x4* sj = (const f32x4*)(rec_s + (rec0 + j) * 8);
          const f32x4 vj[2] = {sj[0], sj[1]};
          take(vj, rec_g[rec0 + j], r4);
        }
      }
    }
  }
  __syncthreads();
  if (total > FS_MAX) ovf = 1;
  const int n = min(total, FS_MAX);
  __syncthreads();
  // 2. T_lo <= T = k-th best MFMA score (2-pass radix select over order-preserving keys); n <= k: every candidate is in
  float T = -INFINITY;
  if (!ovf && n > k) {
    unsigned prefix = 0;
    if (tid == 0) k_rem = k;
    for (int pass = 0; pass < 2; ++pass) {               // 16 key bits: sign, exponent, 7 mantissa bits
      const int shift = 24 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += 256) {
        const unsigned key = f2key(cs[i]);
        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
      }
      __syncthreads();
      if (wave == 0) {                                   // bins 4*lane .. 4*lane+3; suffix sums from the top bin down
        const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const int mine = h0 + h1 + h2 + h3;
        int above = mine;                                // inclusive suffix sum over lanes >= this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(above, o, 64); if (lane + o < 64) above += t; }
        above -= mine;                                   // elements in bins above this lane's four
        const int kr = k_rem;
        if (above < kr && kr <= above + mine) {          // exactly one lane
          int a3 = above, b = 3;
          if (kr <= a3 + h3) b = 3; else { a3 += h3; if (kr <= a3 + h2) b = 2; else { a3 += h2; if (kr <= a3 + h1) b = 1; else { a3 += h1; b = 0; } } }
          sel_bin = 4 * lane + b; k_rem = kr - a3;
        }
      }
      __syncthreads();
      prefix = (prefix << 8) | (unsigned)sel_bin;
      __syncthreads();
    }
    // T_lo = the smallest float whose key starts with the k-th best score's 16 bits: T_lo <= T (within 2^-7 relative), so the
    // short list cut T_lo - delta admits a superset of what T - delta would (exactness is unaffected; two passes saved)
    const unsigned key_lo = prefix << 16;
    const unsigned u = (key_lo & 0x80000000u) ? (key_lo & 0x7fffffffu) : ~key_lo;
    T = __uint_as_float(u);
  }
  // 3. short list: MFMA score >= T - delta
  if (!ovf) {
    const float cut = T - delta;
    for (int i = tid; i < n; i += 256) {
      if (cs[i] >= cut) {
        const int p = atomicAdd(&nsl, 1);
        if (p < SL_MAX) sl_i[p] = ci[i];
      }
    }
  }
  __syncthreads();
  if (nsl > SL_MAX) ovf = 1;
  __syncthreads();
  if (ovf) {
    // Candidate overflow (a stream list, the LDS buffer or the short list: pathological score distributions such as tens of
    // thousands of identical rows). No host round trip and no second launch: THIS block ranks the whole shard for its query
    // with the exact chain (256 threads x Ng / 256 rows x 256 dependent fmaf: ~2 ms per query at 1M rows - slow, exact, rare).
    if (tid == 0) { ovf_q[q] = 1; atomicOr(flags, 1); }
    if (no_fallback) {                                   // tests: expose the raw overflow marker
      for (int j = tid; j < k; j += 256) { out_s[(long)q * k + j] = -INFINITY; out_i[(long)q * k + j] = -2LL; }
      return;
    }
    __syncthreads();
    brute_force_topk<TG>(G, Ng, qs, cs, ci, k, g_offset, out_s + (long)q * k, out_i + (long)q * k);
    return;
  }
  const int m = nsl;
  // 4. exact re-scoring: the fmaf chain of oracle/c/sim_chain.c (chunk c = 0..31 of 8: k = 8c+i then 8c+4+i, i = 0..3)
  // The rows arrive by LDS-DMA in the candidate buffer (dead from here: 64 rows x 512 B per round, wave w stages rows 16 w ..; slot
  // (lane & 31) of row j is fed from source chunk (lane & 31) ^ (j & 31)), all in flight at once; then thread j runs the chain over
  // its row. Round 2 read the rows from global memory inside the chain: 32 loads, four in flight = eight dependent round trips.
  {
    const int lane_ = tid & 63, wave_ = tid >> 6;
    const unsigned ldsR = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)cs)) + (unsigned)wave_ * 8192u;
    for (int base = 0; base < m; base += 64) {
      const int nr = min(64, m - base);
      __syncthreads();                                   // the candidate buffer / the previous round's rows are no longer read
#pragma unroll 1
      for (int i = 0; i < 8; ++i) {
        const int jj = wave_ * 16 + 2 * i + (lane_ >> 5);
        if (wave_ * 16 + 2 * i < nr) {                   // wave-uniform
          const long idx = sl_i[base + min(jj, nr - 1)];
          glds16(G + idx * 256 + (((lane_ & 31) ^ (jj & 31)) << 3), __builtin_amdgcn_readfirstlane(ldsR + 1024u * i));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < nr) sl_s[base + tid] = chain_score_lds<TG>((const char*)cs + tid * 512, tid & 31, qs);
    }
  }
  __syncthreads();
  // 5. rank by (chain score desc, index asc); entries beyond the short list (Ng < k) are (-inf, -1)
  for (int j = tid; j < m; j += 256) {
    const float s = sl_s[j]; const int idx = sl_i[j];
    int rank = 0;
    for (int i = 0; i < m; ++i) rank += (sl_s[i] > s || (sl_s[i] == s && sl_i[i] < idx)) ? 1 : 0;
    if (rank < k) { out_s[(long)q * k + rank] = s; out_i[(long)q * k + rank] = (long long)idx + g_offset; }
  }
  for (int j = m + tid; j < k; j += 256) { out_s[(long)q * k + j] = -INFINITY; out_i[(long)q * k + j] = -1LL; }
}

__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// one block per query: k rounds of block-wide arg-best over nparts*KMAX candidates held in LDS.
__global__ void __launch_bounds__(256) sim_topk_merge(const float* ws_s, const int* ws_i, int n, int k, long long g_offset,
                                                      float* out_s, long long* out_i, const int* gate_q) {
  if (gate_q && gate_q[blockIdx.x] == 0) return;                // fallback: only the queries whose candidate lists overflowed
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

inline int device_cus() { return cor_device_cus(); }

// ---------------------------------------------------------------------------------------------------------------------
// v4 (16-bit galleries, C = 256, SMALL shards - the 8-GPU shard shapes of BASELINE configs[2], 256..512 x 12.5k - and few queries
// against up to ~130k rows): TWO launches and LOCAL thresholds instead of prep -> sample scan -> tau -> append scan -> final.
//   sim_block_scan: a block = 32 * QB queries x one gallery slice of 256 * T rows. Every wave owns ALL the block's queries (their
//     K-fragments in registers) and 32 * T rows of its own, which reach its PRIVATE 16-KiB LDS buffer by LDS-DMA in K-half tiles
//     (no block barrier inside the scan); the 16 * T * QB scores of a lane STAY IN REGISTERS. The block then knows, per query, 64
//     disjoint row classes' maxima; their k-th largest is a lower bound of the query's k-th best score over the whole shard (k
//     classes each hold a row at least that good), so rows below it minus delta_q cannot be in the answer: the lanes append the
//     few registers that pass (~k per query and slice) to per-(query, slice) lists. No sample pass, no global threshold, no
//     second scan, no atomics on global memory.
//   sim_final_wave: ONE WAVE per query gathers its lists, finds the k-th best MFMA score by a counting binary search, re-scores
//     the short list with the exact fp32 fmaf chain (rows staged by LDS-DMA, all in flight at once) and ranks it.
// Exactness (same argument as v3, with the slice's own threshold): let T_b be the k-th best MFMA score inside slice b. The k best
// rows of slice b have chain scores >= T_b - eps, so the shard's k-th best chain score is >= T_b - eps for EVERY b; a row g of the
// exact answer therefore has MFMA score >= T_b(g) - 2 eps >= tau_b(g) - delta and is appended by its block. Every row of the MFMA
// top-k is appended too (global k-th best >= local k-th best), so the final's T is the shard's k-th best MFMA score and the short
// list {MFMA score >= T - delta} holds the exact answer. A kernel boundary between the two costs ~1.5 us; keeping the selection in
// the scan launch would need an agent-scope release + counter + acquire (~3.4 us by the MI355X guide's price list) and one block
// ranking 32-64 queries: measured choice, see DESIGN 3.4.
constexpr int SB_NC = 4096;                            // candidates per query held in LDS by sim_final_wave (32 KiB)
constexpr int SB_SL = 64;                              // short list (one entry per lane)
constexpr int SB_MAXSL = 192;                          // slices per query the wave final can gather (make_small keeps nslices below)

struct SmallPlan {
  bool ok;
  int qb, T, nqg, nslices, cap, grid, xcd_map;
  size_t off_flags, off_ovf, off_cnt, off_cand, bytes;
};
inline SmallPlan make_small(int Bq, int Ng, int k) {
  SmallPlan p{};
  p.qb = Bq > 32 ? 2 : 1;
  p.nqg = cdiv(Bq, 32 * p.qb);
  const int tmax = p.qb == 2 ? 2 : 4;                  // accumulators: 16 * T * QB <= 64 registers
  // A block's time is a LATENCY chain (~11 us: query conversion, two DMA round trips, threshold, append), not throughput, and the
  // selection kernel's time grows with the candidates (~12 per slice and query): the fewest, largest slices that still fit one
  // round of blocks. Shapes that need a second round (512 x 32 000: 55 us here against 48 on the global-threshold pipeline) or more
  // than 32 slices per query stay on the global-threshold pipeline.
  int t = cdiv(Ng, 256);
  p.T = t <= 1 ? 1 : (t == 2 ? 2 : tmax);              // instantiated: 1, 2, (4 for QB = 1)
  if (p.T > tmax) p.T = tmax;
  p.nslices = cdiv(Ng, 256 * p.T);
  p.cap = k <= 12 ? 128 : 256;                         // entries per (query, slice) list: ~k + 2 expected; a sparse last slice: all its rows
  p.xcd_map = p.nslices >= 8 ? 1 : 0;                  // blocks that share a gallery slice share an XCD (its L2)
  p.grid = p.nqg * (p.xcd_map ? ((p.nslices + 7) & ~7) : p.nslices);
  // k <= 16: the threshold is the k-th largest of 32 class maxima; near k = 32 it is their MINIMUM, a weak and volatile bound (one query
  // of 300 overflowed a 256-entry list at 300 x 5 000, k = 32): such searches keep the global threshold
  p.ok = k <= 16 && p.nslices <= 32 && p.nqg * p.nslices <= device_cus() && (long)p.nslices * (k + 8) * 13 / 10 <= 3072 && p.nslices <= SB_MAXSL;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 255) & ~(size_t)255; return at; };
  p.off_flags = take(16);
  p.off_ovf = take((size_t)Bq * 4);
  p.off_cnt = take((size_t)Bq * p.nslices * 4);
  p.off_cand = take((size_t)Bq * p.nslices * p.cap * 8);
  p.bytes = o;
  return p;
}


template <typename TG, int QB, int T>
__global__ void __launch_bounds__(512, 2) sim_block_scan(const float* __restrict__ Q, const TG* __restrict__ G, int Bq, int Ng, int k, int nqg,
                                                         int nslices, int xcd_map, int cap, int* __restrict__ cnt, uint2* __restrict__ cand,
                                                         int* __restrict__ flags, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef COR_PROBES
  unsigned long long ts_[10]; int nts_ = 0;
#define SB_STAMP() do { __builtin_amdgcn_sched_barrier(0); ts_[nts_++] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SB_STAMP() do { } while (0)
#endif
  constexpr int QPB = 32 * QB, ZB = QB * 16 * 64 * 16, TPQ = 512 / QPB, FPT = 256 / TPQ, VPT = 64 / TPQ, CS = QPB + 1;
  constexpr int RPW = 32 * T, RPB = 8 * RPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int slice, qg;
  if (xcd_map) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; slice = (j / nqg) * 8 + x; qg = j % nqg; }
  else { slice = blockIdx.x / nqg; qg = blockIdx.x % nqg; }
  if (slice >= nslices) return;                        // padding blocks of the XCD mapping (whole block, before any barrier)
  if (blockIdx.x == 0 && tid == 0) flags[0] = 0;       // per-call overflow flag (the final kernel runs behind this one)
  const int q0 = qg * QPB;
  // A short (last) slice spreads its rows EVENLY over the eight waves (floor(rows / 8) each, the remainder one apiece, instead of 32 T), so
  // that the 32 row classes below stay populated: a wave with >= 8 rows fills its four classes, hence every slice with >= 64 rows has
  // all 32 (k <= 32 of them are needed), and a slice with fewer rows may well declare every row a candidate - the list holds >= 128.
  // (Seen at 32 x 12 500: 212 rows in two waves, 8 classes < k, threshold -inf, 212 candidates per query, every query on the brute-force
  // fallback; and at 300 x 5 000, k = 32: 392 rows in seven waves, 28 classes, the same.)
  const long slice0 = (long)slice * RPB;
  const int rows_here = (int)(Ng - slice0 < RPB ? Ng - slice0 : RPB);
  const int rpw = rows_here >> 3, rem = rows_here & 7; // floor(rows / 8) per wave, the first `rem` waves one more: >= 8 each from 64 rows up
  const long w0 = slice0 + wave * rpw + (wave < rem ? wave : rem);          // this wave's first gallery row
  const int own = rpw + (wave < rem ? 1 : 0);          // rows this wave owns (wave-uniform): 32 T in a full slice
  const int ntw = (own + 31) >> 5;                     // 32-row tiles this wave scans (<= T)
  const int nsteps = 2 * ntw;                          // K-half tiles

  char* Z = smem;                                      // query image (prologue), then class maxima / thresholds / list counters
  char* Aw = smem + ZB + wave * 16384;                 // this wave's two 8-KiB K-half buffers (32 rows x 256 B)
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)Aw));
  const int rowl = lane >> 4, sl = lane & 15;
  auto issue = [&](int s) {                            // K-half tile s = 2 t + p -> buffer p. LDS image lane-linear: 4 rows x 256 B per
    const int t = s >> 1, p = s & 1;                   // wave-instruction; slot sl of row `row` holds source chunk sl ^ (row & 15)
    const long g0 = w0 + 32 * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + rowl;
      long gr = g0 + row;
      gr = gr < Ng ? gr : (long)Ng - 1;                // clamped duplicates are masked to -inf below
      glds16(G + gr * 256 + ((sl ^ (row & 15)) + 16 * p) * 8, __builtin_amdgcn_readfirstlane(ldsA + p * 8192 + 1024 * i));
    }
  };
  SB_STAMP();                                          // 0: start
  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);

  // queries -> gallery dtype, MFMA-fragment order in Z: image[(qblk * 16 + c) * 64 + lane'] = q[qblk*32 + (lane'&31)][16c + 8(lane'>>5) ..+8]
  // thread (query qq, part pp) converts FPT consecutive values: whole 16-value K-steps c = pp * FPT / 16 ...
  const int qq = tid / TPQ, pp = tid % TPQ;
  float nrm2 = 0.f;
  {
    const float* qrow = Q + (long)min(q0 + qq, Bq - 1) * 256 + pp * FPT;
#pragma unroll
    for (int j = 0; j < FPT / 8; ++j) {                // one 16-byte fragment piece = 8 values
      const f32x4 lo = *(const f32x4*)(qrow + 8 * j), hi = *(const f32x4*)(qrow + 8 * j + 4);
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float x = round_to<TG>(v[i]); nrm2 = fmaf(x, x, nrm2); }
      const int kpos = pp * FPT + 8 * j, c = kpos >> 4, hh = (kpos >> 3) & 1;
      ((uint4*)Z)[((qq >> 5) * 16 + c) * 64 + hh * 32 + (qq & 31)] = q_frag16_vals<TG>(lo, hi);
    }
#pragma unroll
    for (int o = 1; o < TPQ; o <<= 1) nrm2 += __shfl_xor(nrm2, o, 64);   // the TPQ threads of a query are adjacent lanes: every one holds |q|^2
  }
  __syncthreads();
  uint4 qf[QB][16];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int c = 0; c < 16; ++c) qf[qb][c] = ((const uint4*)Z)[(qb * 16 + c) * 64 + lane];
  __syncthreads();                                     // Z is free from here on
  SB_STAMP();                                          // 1: queries converted, fragments in registers
  float* cmx = (float*)Z;                              // [32 classes][CS]: class = (wave * 2 + h) * 2 + j ; padded stride: conflict-free both ways
  float* thr = cmx + 32 * CS;                          // [QPB]
  int* lcnt = (int*)(thr + QPB);                       // [QPB]
  float* dlt = (float*)(lcnt + QPB);                   // [QPB] delta_q
  float* smx = dlt + QPB;                              // [QPB] kth: the k-th largest class maximum (the level an overflowing list is re-cut ABOVE)
  int* lcnt2 = (int*)(smx + QPB);                      // [QPB] entries of the re-cut list
  if (pp == 0) dlt[qq] = SIM_DELTA * fmaxf(1.f, sqrtf(nrm2));

  f32x16 acc[QB][T];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[qb][t][e] = 0.f;
  const int rd_base = r * 256 + (((h ^ r) & 1) << 4), rd_x = (r >> 1) & 7;     // chunk (2c' + h) ^ (r & 15) of row r
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t < ntw) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int s = 2 * t + p;
        // K-half tile s has landed once at most the 8 copies of tile s + 1 are outstanding (VMEM retires in order)
        if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* buf = Aw + p * 8192;
        uint4 af[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) af[c] = *(const uint4*)(buf + rd_base + ((c ^ rd_x) << 5));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint4 av = af[c & 3];
          if (c + 4 < 8) af[c & 3] = *(const uint4*)(buf + rd_base + (((c + 4) ^ rd_x) << 5));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            if (__is_same(TG, bf16_t))
              acc[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, qf[qb][8 * p + c]), acc[qb][t], 0, 0, 0);
            else
              acc[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, qf[qb][8 * p + c]), acc[qb][t], 0, 0, 0);
          }
        }
        if (s + 2 < nsteps) {                          // buffer p is consumed (its reads fed the MFMAs above): refill it two steps ahead
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          issue(s + 2);
        }
      }
    }
  }
  SB_STAMP();                                          // 2: scan done
  // rows beyond the wave's share (the next wave's rows, clamped duplicates, absent tiles) never count: only a short slice has any
  if (own < RPW) {                                     // (a full slice: own == 32 T)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (32 * t + (e & 3) + 8 * (e >> 2) + 4 * h >= own) acc[qb][t][e] = -INFINITY;
  }
  // class maxima: 32 disjoint row classes per query and block = (wave, lane half, register parity); their k-th largest is a lower
  // bound of the query's k-th best score in the shard (k classes each hold a row at least that good)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float c0 = -INFINITY, c1 = -INFINITY;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 16; e += 2) { c0 = fmaxf(c0, acc[qb][t][e]); c1 = fmaxf(c1, acc[qb][t][e + 1]); }
    cmx[((wave * 2 + h) * 2 + 0) * CS + qb * 32 + r] = c0;
    cmx[((wave * 2 + h) * 2 + 1) * CS + qb * 32 + r] = c1;
  }
  __syncthreads();
  SB_STAMP();                                          // 3: class maxima exchanged
  // tau_q = k-th largest of the 32 class maxima: lane <-> query, a 32-element bitonic network in registers (the K-fragments are dead),
  // 240 compare-exchanges for all 64 queries at once. (First forms: TPQ lanes per query with ds_bpermute count reductions, ~3 us;
  // one scalar counting search per query, 130 cycles per step of a serial v_cmp -> s_bcnt1 -> s_cselect chain: 8 us.)
  if (wave == 0 && lane < QPB) {
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = cmx[c * CS + lane];
#pragma unroll
    for (int kk = 2; kk <= 32; kk <<= 1)
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int l = i ^ j;
          if (l > i) {
            const float x = v[i], y = v[l];
            if ((i & kk) == 0) { v[i] = fmaxf(x, y); v[l] = fminf(x, y); }      // descending
            else               { v[i] = fminf(x, y); v[l] = fmaxf(x, y); }
          }
        }
    float kth = v[0];
#pragma unroll
    for (int c = 1; c < 32; ++c) kth = (c == k - 1) ? v[c] : kth;
    smx[lane] = kth;
    lcnt2[lane] = 0;
    // (fewer than k classes hold a row: -inf, every real row is a candidate; the floor keeps masked -inf scores out with ONE compare)
    thr[lane] = fmaxf(kth - dlt[lane], -3.0e38f);
    lcnt[lane] = 0;
  }
  __syncthreads();
  SB_STAMP();                                          // 4: thresholds
  // append the registers that pass to the block's per-query lists (the A buffers are free: every wave is past its scan). One LDS
  // atomic per lane and query block reserves room for the lane's passing registers; the stores are EXEC-masked, not branched: hipcc
  // turned `if (pass) store` into 64 taken branches per wave (9 000 cycles of a 45 000-cycle block), so each register is one
  // v_cmp -> s_and_saveexec -> ds_write2_b32 (score, row) -> address += 8 -> restore exec, in inline asm.
  uint2* lst = (uint2*)(smem + ZB);                    // [QPB][cap]
  const unsigned lds_lst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(smem + ZB));
  // pass 0: everything >= kth - delta. pass 1 (round 5), only for queries whose pass-0 list overflowed: a slice of duplicates ties
  // hundreds of rows AT its own k-th class maximum for every query; the list is re-cut to the rows strictly ABOVE kth (few: they sit in
  // fewer than k of the 32 row classes), and kth stays behind as the level of what was left out. sim_final_wave skips the left-out
  // rows when that level is below the query's short-list cut (exact: every row of the answer has an MFMA score >= the cut), so only a
  // query the duplicated row is really close to still takes the brute-force path (rounds 1-4: every query of the call did).
  auto append = [&](const int pass) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int ql = qb * 32 + r;
      float tq = (q0 + ql < Bq) ? thr[ql] : INFINITY;              // padding queries never append
      if (pass == 1) tq = (q0 + ql < Bq && lcnt[ql] > cap) ? __uint_as_float(__float_as_uint(smx[ql]) + (smx[ql] >= 0.f ? 1u : -1u)) : INFINITY;   // next float above kth
      int* counter = pass == 0 ? lcnt : lcnt2;
      int np = 0;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) np += acc[qb][t][e] >= tq ? 1 : 0;
      const int pos0 = np > 0 ? atomicAdd(&counter[ql], np) : 0;
      if (pos0 + np > cap - pass) tq = INFINITY;                   // the list is full: this lane stores nothing; the count flags the query
      unsigned addr = lds_lst + (unsigned)(ql * cap + pos0) * 8u;
      const unsigned rowb = (unsigned)w0 + 4u * h;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned rowv = rowb + (unsigned)(32 * t + (e & 3) + 8 * (e >> 2));
          unsigned long long sv;
          asm volatile("v_cmp_ge_f32 vcc, %2, %3\n\ts_and_saveexec_b64 %1, vcc\n\tds_write2_b32 %0, %2, %4 offset1:1\n\tv_add_u32 %0, 8, %0\n\ts_mov_b64 exec, %1"
                       : "+v"(addr), "=&s"(sv) : "v"(acc[qb][t][e]), "v"(tq), "v"(rowv) : "vcc", "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the asm stores are invisible to hipcc's counter bookkeeping
    __syncthreads();
  };
  append(0);
  {
    bool over = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) over = over || (q0 + qb * 32 + r < Bq && lcnt[qb * 32 + r] > cap);
    int* any_over = lcnt2 + QPB;                                   // (inside Z: the kernel's LDS is allocated to the last byte at QB = 2)
    if (tid == 0) *any_over = 0;
    __syncthreads();
    if (over) *any_over = 1;
    __syncthreads();
    if (*any_over) append(1);                                      // (block-uniform)
  }
  SB_STAMP();                                          // 5: lists filled
  if (q0 + qq < Bq) {
    const int n_raw = lcnt[qq], n = min(n_raw, cap);
    const long li = (long)(q0 + qq) * nslices + slice;
    if (n_raw > cap) {
      // overflow: cnt = cap + 1 + n2 (n2 <= cap - 1 entries strictly above kth follow entry 0 = {kth, marker}); 2 cap + 1: even the re-cut
      // list overflowed (nothing usable: the query takes the brute-force path)
      const int n2 = lcnt2[qq];
      const bool hard = n2 > cap - 1;
      if (pp == 0) { cnt[li] = hard ? 2 * cap + 1 : cap + 1 + n2; cand[li * cap] = make_uint2(__float_as_uint(smx[qq]), 0xFFFFFFFFu); }
      if (!hard)
        for (int j = pp; j < n2; j += TPQ) cand[li * cap + 1 + j] = lst[qq * cap + j];
    } else {
      if (pp == 0) cnt[li] = n_raw;
      for (int j = pp; j < n; j += TPQ) cand[li * cap + j] = lst[qq * cap + j];
    }
  }
#ifdef COR_PROBES
  SB_STAMP();                                          // 6: written out
  if (stamps && lane == 0) {
    for (int i = 0; i < nts_; ++i) stamps[((long)blockIdx.x * 8 + wave) * 8 + i] = ts_[i];
  }
#endif
#undef SB_STAMP
}

// ONE WAVE per query: gather the candidates, select, re-score exactly, rank. Two gather front ends: RECORDS = false: the
// per-(query, slice) entry lists of sim_block_scan (cnt / cand; nl = slices); RECORDS = true: the per-(query, stream) records of
// sim_scan<APPEND> (cnt / rec_s / rec_g; nl = streams; a record = the 8 scores of one lane's accumulator column + the tile's
// first row; the scores >= tau_q are kept).
template <typename TG, bool RECORDS>
__global__ void __launch_bounds__(64) sim_final_wave(const float* __restrict__ Q, const TG* __restrict__ G, const int* __restrict__ cnt,
                                                     const uint2* __restrict__ cand, const float* __restrict__ rec_s, const int* __restrict__ rec_g,
                                                     const float* __restrict__ tau, int nslices, int cap, int Ng, int k, long long g_offset,
                                                     float* out_s, long long* out_i, int* flags, int* ovf_q, int no_fallback) {
  __shared__ __attribute__((aligned(16))) float cs[SB_NC];
  __shared__ __attribute__((aligned(16))) int ci[SB_NC];
  __shared__ __attribute__((aligned(16))) char rows[32 * 512];
  __shared__ __attribute__((aligned(16))) float qs[256];
  __shared__ float sl_s[SB_SL];
  __shared__ int sl_i[SB_SL];
  const int q = blockIdx.x, lane = threadIdx.x;
  // (the first 64 list lengths are requested together with the query row: one global round trip instead of two)
  const int c_first = (!RECORDS && lane < nslices) ? cnt[(long)q * nslices + lane] : 0;
  {
    const f32x4 v = *(const f32x4*)(Q + (long)q * 256 + 4 * lane);
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float x = round_to<TG>(v[i]); qs[4 * lane + i] = x; n2 = fmaf(x, x, n2); }
    n2 = wave_sum(n2);
    sl_s[lane] = n2;                                   // (parked: read back as delta below, after the barrier)
  }
  int n = 0;
  bool ovf = false, lists_over = false;
  __shared__ bool sovf[SB_MAXSL];
  if (RECORDS) {
    // 1r. lane <-> stream; record 0 of every stream is fetched TOGETHER with the stream's count (most streams hold 0 or 1 records):
    // one global round trip instead of three dependent ones. Score [g][i] of lane quarter r4 / 4 is row g0 + 16 g + r4 + i.
    __shared__ int total;
    if (lane == 0) total = 0;
    __syncthreads();
    const float tq = tau[q];
    auto take = [&](const f32x4 (&v)[2], int g0, int r4) {          // one LDS atomic per record (not per score), then predicated stores
      int np = 0;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) np += (v[g][i] >= tq && v[g][i] > -INFINITY) ? 1 : 0;
      if (np == 0) return;
      int pos = atomicAdd(&total, np);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (v[g][i] >= tq && v[g][i] > -INFINITY) {
            if (pos < SB_NC) { cs[pos] = v[g][i]; ci[pos] = g0 + i + 16 * g + r4; }
            ++pos;
          }
    };
    // the loads of up to 8 streams per lane (512 of up to 1024 streams) are issued together: one global round trip, not one per 64 streams
    constexpr int RS = 8;
    for (int sb = 0; sb < nslices; sb += 64 * RS) {
      f32x4 v0[RS][2]; int g00[RS], cc[RS];
#pragma unroll
      for (int u = 0; u < RS; ++u) {
        const int st = sb + 64 * u + lane;
        const long rec0 = ((long)q * nslices + (st < nslices ? st : 0)) * cap;
        const f32x4* src = (const f32x4*)(rec_s + rec0 * 8);
        v0[u][0] = src[0]; v0[u][1] = src[1];
        g00[u] = rec_g[rec0];
        cc[u] = st < nslices ? cnt[(long)q * nslices + st] : 0;
      }
#pragma unroll
      for (int u = 0; u < RS; ++u) {
        const int st = sb + 64 * u + lane;
        if (cc[u] > cap) ovf = true;
        const int nrec = min(cc[u], cap), r4 = 4 * (st & 3);
        if (nrec > 0) take(v0[u], g00[u], r4);
        if (nrec > 1) {
          const long rec0 = ((long)q * nslices + st) * cap;
          for (int j = 1; j < nrec; ++j) {
            const f32x4* sj = (const f32x4*)(rec_s + (rec0 + j) * 8);
            const f32x4 vj[2] = {sj[0], sj[1]};
            take(vj, rec_g[rec0 + j], r4);
          }
        }
      }
    }
    __syncthreads();
    n = total;
  }
  // 1. gather. Pass 1: lane <-> slice, list lengths and their exclusive prefix (soff). Pass 2: lane <-> ENTRY of the flattened
  // list (its slice by a binary search over soff): four independent loads per lane and round, so the whole gather is about one
  // global round trip (a first form walked entry j of every slice per round: one dependent round trip per entry, ~12 us)
  if (!RECORDS) {
    __shared__ int soff[SB_MAXSL + 1];
    for (int s0 = 0; s0 < nslices; s0 += 64) {
      const int s = s0 + lane;
      int c = s0 == 0 ? c_first : (s < nslices ? cnt[(long)q * nslices + s] : 0);
      // an overflowing list was re-cut by sim_block_scan to the rows strictly above its k-th class maximum (entries 1 ..); entry 0 holds
      // that level, judged against the cut below
      if (s < nslices) sovf[s] = c > cap;
      if (c > cap) { lists_over = true; if (c > 2 * cap) ovf = true; c = c > 2 * cap ? 0 : c - cap - 1; }   // the re-cut list (entries 1 ..), or nothing
      int incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      if (s < nslices) soff[s] = n + incl - c;
      n += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) soff[nslices] = n;
    __syncthreads();
    const uint2* qc = cand + (long)q * nslices * cap;
    const int nn = min(n, SB_NC);
    for (int i0 = 0; i0 < nn; i0 += 256) {
      uint2 e[4]; int idx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        idx[u] = i0 + 64 * u + lane;
        const int i = min(idx[u], nn - 1);
        int lo = 0, hi = nslices;                          // largest s with soff[s] <= i (empty slices repeat an offset: the search skips them)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (soff[mid] <= i) lo = mid; else hi = mid; }
        e[u] = qc[(long)lo * cap + (i - soff[lo]) + (sovf[lo] ? 1 : 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (idx[u] < nn) { cs[idx[u]] = __uint_as_float(e[u].x); ci[idx[u]] = (int)e[u].y; }
    }
    __syncthreads();
  }
  n = __builtin_amdgcn_readfirstlane(n);
  if (__builtin_amdgcn_ballot_w64(ovf) != 0 || n > SB_NC) ovf = true; else ovf = false;
  __syncthreads();
  const float delta = SIM_DELTA * fmaxf(1.f, sqrtf(sl_s[0]));
  __syncthreads();
  // 2. T_lo <= T = k-th best MFMA score: counting binary search over the top 16 key bits (n <= k: every candidate is in)
  float T = -INFINITY;
  int m = 0;
  if (!ovf) {
    if (n > k) {
      // keys of the first 512 candidates in registers (8 per lane; typical n: a few hundred), the rest re-read from LDS; every
      // step of the search is compares + ballots + scalar bit counts (no cross-lane reduction, no LDS round trip)
      unsigned kr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) kr[j] = lane + 64 * j < n ? f2key(cs[lane + 64 * j]) >> 16 : 0u;
      unsigned lo = 0x007Fu, hi = 0x10000u;
#pragma unroll 1
      for (int it = 0; it < 16; ++it) {
        const unsigned mid = (lo + hi) >> 1;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kr[j] >= mid));
        for (int i0 = 512; i0 < n; i0 += 64)
          c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i0 + lane < n && (f2key(cs[min(i0 + lane, n - 1)]) >> 16) >= mid));
        if (c >= k) lo = mid; else hi = mid;
      }
      // T_lo = the smallest float whose key starts with the k-th best score's 16 bits (within 2^-7 relative below T): the short list
      // cut T_lo - delta admits a superset of what T - delta would; exactness is unaffected
      T = lo == 0x007Fu ? -INFINITY : key2f_floor(lo << 16);
    }
    // 3. short list: MFMA score >= T - delta (ballot compaction; order irrelevant: the ranking below is total)
    const float cut = T - delta;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const bool pass = i < n && cs[i] >= cut && cs[i] > -INFINITY;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
      const int pos = m + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (pass && pos < SB_SL) sl_i[pos] = ci[i];
      m += __builtin_popcountll(bal);
    }
    if (m > SB_SL) ovf = true;
    // slices whose lists overflowed: the rows left out of the re-cut list score at most the level in entry 0; they can hold a row of
    // the answer only if that level reaches the cut (every row of the exact answer has an MFMA score >= cut: the short list's own
    // criterion). Below it they are skipped - exact; at or above it the query takes the brute-force path as before.
    if (!RECORDS && __builtin_amdgcn_ballot_w64(lists_over) != 0) {
      for (int s0 = 0; s0 < nslices; s0 += 64) {
        const int s = s0 + lane;
        const bool bad = s < nslices && sovf[s] && !(__uint_as_float(cand[((long)q * nslices + s) * cap].x) < cut);
        if (__builtin_amdgcn_ballot_w64(bad) != 0) ovf = true;
      }
    }
  }
  __syncthreads();
  if (ovf) {
    if (lane == 0) { ovf_q[q] = 1; atomicOr(flags, 1); }
    if (no_fallback) {
      for (int j = lane; j < k; j += 64) { out_s[(long)q * k + j] = -INFINITY; out_i[(long)q * k + j] = -2LL; }
      return;
    }
    brute_force_topk<TG, 64>(G, Ng, qs, cs, ci, k, g_offset, out_s + (long)q * k, out_i + (long)q * k);
    return;
  }
  if (lane == 0) ovf_q[q] = 0;
  // 4. exact re-scoring, 32 rows per round: the rows arrive by LDS-DMA (2 rows per wave-instruction, all in flight at once;
  // slot (lane & 31) of row j is fed from source chunk (lane & 31) ^ j), then lane j runs the chain over its row
  const unsigned ldsR = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)rows));
  for (int base = 0; base < m; base += 32) {
    const int nr = min(32, m - base);
#pragma unroll 1
    for (int i = 0; 2 * i < nr; ++i) {
      const int jj = 2 * i + (lane >> 5);
      const long idx = sl_i[base + min(jj, nr - 1)];
      glds16(G + idx * 256 + (((lane & 31) ^ jj) << 3), __builtin_amdgcn_readfirstlane(ldsR + 1024 * i));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane < nr) sl_s[base + lane] = chain_score_lds<TG>(rows + lane * 512, lane, qs);
    __syncthreads();
  }
  // 5. rank by (chain score desc, index asc); entries beyond the short list (Ng < k) are (-inf, -1)
  if (lane < m) {
    const float s = sl_s[lane]; const int idx = sl_i[lane];
    int rank = 0;
    for (int i = 0; i < m; ++i) rank += (sl_s[i] > s || (sl_s[i] == s && sl_i[i] < idx)) ? 1 : 0;
    if (rank < k) { out_s[(long)q * k + rank] = s; out_i[(long)q * k + rank] = (long long)idx + g_offset; }
  }
  for (int j = m + lane; j < k; j += 64) { out_s[(long)q * k + j] = -INFINITY; out_i[(long)q * k + j] = -1LL; }
}


// plan of the threshold-and-append pipeline (host; shared by the launcher and cor_topk_workspace_bytes)
struct V3Plan {
  int qb, nqg;                       // query blocks per wave (1 | 2), query groups of 256 * qb
  int tiles, nsplit, tiles_per_split, nstreams, cap;                 // APPEND pass
  int s_stride, s_tiles, s_nsplit, s_tiles_per_split, ngroups;       // SAMPLE pass (ngroups == 0: no sample, tau = -inf)
  size_t off_img, off_sg, off_dq, off_tau, off_flags, off_ovf, off_cnt, off_recs, off_recg, off_lists, bytes;
};
inline V3Plan make_v3(int Bq, int Ng, int k) {
  V3Plan p{};
  p.qb = Bq > 256 ? 2 : 1;
  p.nqg = cdiv(Bq, 256 * p.qb);
  p.tiles = cdiv(Ng, 64);                              // 64-row super-tiles
  int want = device_cus() / p.nqg;                     // one resident block per CU
  if (want > 256) want = 256;                          // nstreams <= 1024
  if (want > p.tiles) want = p.tiles;
  if (want < 1) want = 1;
  p.tiles_per_split = cdiv(p.tiles, want);
  p.nsplit = cdiv(p.tiles, p.tiles_per_split);
  p.nstreams = 4 * p.nsplit;                           // (query, slice, lane quarter)
  long expect;                                         // expected accepted scores per query over the whole shard
  if (Ng <= 4096) {                                    // tiny shard: no sample pass, every row is a candidate (<= FS_MAX)
    p.ngroups = 0; p.s_tiles = 0; p.s_stride = 1; p.s_nsplit = 0; p.s_tiles_per_split = 0;
    p.cap = 2 * p.tiles_per_split;                     // records (32-row tiles) per (slice, lane quarter) stream: every tile is one
  } else {
    // every 16th super-tile, at least ~128 of them: the sample pass costs 1/stride of the full scan plus a fixed ~8 us; the
    // candidates it admits (~2 k stride per query) must stay rare per wave and tile (the append path is divergent)
    p.s_stride = p.tiles / 16 >= 128 ? 16 : (p.tiles / 128 > 1 ? p.tiles / 128 : 1);
    p.s_tiles = cdiv(p.tiles, p.s_stride);
    int sw = device_cus() / p.nqg;
    if (sw > 256) sw = 256;
    if (sw > p.s_tiles) sw = p.s_tiles;
    if (sw < 1) sw = 1;
    p.s_tiles_per_split = cdiv(p.s_tiles, sw);
    p.s_nsplit = cdiv(p.s_tiles, p.s_tiles_per_split);
    p.ngroups = 4 * p.s_nsplit;
    expect = 3L * k * p.s_stride;                      // ~ k * Ng / sample rows, x3 for group-maximum slack
    // k > 16: tau is the k-th largest of only 32 super-group maxima - near k = 32 their MINIMUM, which admits ~4x the candidates of the
    // tighter bound the factor 3 was sized for: without this most searches overflow some stream and fall back (exact, slow)
    if (k > 16) expect *= 4;
    // records per stream: Poisson with mean ~ expect / nstreams (< 1 on large shards); +10 keeps P(overflow) per search < 1e-4
    p.cap = (int)(2 * expect / p.nstreams) + 10;
  }
  const TopkPlan2 p2 = make_plan2(Bq, Ng, k, device_cus());
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 255) & ~(size_t)255; return at; };
  p.off_img = take((size_t)p.nqg * 8 * p.qb * 16 * 64 * 16);        // whole query blocks of every group (rows past Bq: copies)
  p.off_sg = take((size_t)32 * p.nqg * 256 * p.qb * 4);               // 32 super-group maxima per query slot (whole query blocks)
  p.off_dq = take((size_t)Bq * 4);
  p.off_tau = take((size_t)Bq * 4);
  p.off_flags = take(16);
  p.off_ovf = take((size_t)Bq * 4);
  p.off_cnt = take((size_t)Bq * p.nstreams * 4);
  p.off_recs = take((size_t)Bq * p.nstreams * p.cap * 32);            // a record = 8 scores
  p.off_recg = take((size_t)Bq * p.nstreams * p.cap * 4);
  p.off_lists = take((size_t)Bq * p2.nparts * p2.kmax * 8);          // fallback list kernels
  p.bytes = o;
  return p;
}

int launch_merge(const float* ws_s, const int* ws_i, int Bq, int n, int k, long long g_offset, float* out_s, long long* out_i,
                 const int* gate_q, hipStream_t s) {
  static DevOnce once;
  cor_max_dyn_lds((const void*)sim_topk_merge, 8192 * 8, once);
  hipLaunchKernelGGL(sim_topk_merge, dim3(Bq), dim3(256), (size_t)n * 8, s, ws_s, ws_i, n, k, g_offset, out_s, out_i, gate_q);
  COR_CHECK_LAUNCH();
  return 0;
}

template <typename TG>
int launch_lists_v2(const float* Q, const TG* G, int Bq, int Ng, int k, long long g_offset, float* out_s, long long* out_i, float* ws_s,
                    const int* gate, const int* gate_q, hipStream_t s) {
  const TopkPlan2 p = make_plan2(Bq, Ng, k, device_cus());
  int* ws_i = (int*)(ws_s + (long)Bq * p.nparts * p.kmax);
  const dim3 grid(p.nqg * p.nsplit), block(256);
  const size_t lds = 2 * 32 * 256 * 2;
#define SIM_V2(KM) hipLaunchKernelGGL((sim_topk_v2<TG, KM>), grid, block, lds, s, Q, G, Bq, Ng, p, ws_s, ws_i, gate)
  if (p.kmax == 8) SIM_V2(8); else if (p.kmax == 16) SIM_V2(16); else SIM_V2(32);
#undef SIM_V2
  COR_CHECK_LAUNCH();
  return launch_merge(ws_s, ws_i, Bq, p.nparts * p.kmax, k, g_offset, out_s, out_i, gate_q, s);
}

template <typename TG, int QB>
int launch_v3(const float* Q, const TG* G, int Bq, int Ng, int k, long long g_offset, float* out_s, long long* out_i, char* w,
              const V3Plan& p, int flags, hipStream_t s) {
  constexpr size_t lds = (size_t)SCAN_NS * 64 * 256 * 2;
  static DevOnce once_s, once_a, once_f;
  cor_max_dyn_lds((const void*)sim_scan<TG, QB, true>, (int)lds, once_s);
  cor_max_dyn_lds((const void*)sim_scan<TG, QB, false>, (int)lds, once_a);
  constexpr size_t fs_lds = (size_t)FS_MAX * 8 + SL_MAX * 8 + 256 * 4 + 256 * 4;
  cor_max_dyn_lds((const void*)sim_final<TG>, (int)fs_lds, once_f);
  unsigned* sg = (unsigned*)(w + p.off_sg); float* dq = (float*)(w + p.off_dq); float* tau = (float*)(w + p.off_tau);
  const int Bqp = p.nqg * 256 * QB;
  int* dflags = (int*)(w + p.off_flags); int* ovf_q = (int*)(w + p.off_ovf); int* cnt = (int*)(w + p.off_cnt);
  float* rec_s = (float*)(w + p.off_recs); int* rec_g = (int*)(w + p.off_recg);
  uint4* qimg = (uint4*)(w + p.off_img);
  // 0. queries -> gallery dtype, fragment-major; clears the per-call overflow flags. (Folding this conversion into the SAMPLE pass -
  // every wave converting its own 64 fp32 rows, the blocks of slice 0 publishing the image - was built and measured in round 4: the
  // SAMPLE pass grew from 8-10 to 21.6 us at 125k rows and from 22.7 to 37 us at 1M, against the 5.6-us launch it replaced: reverted. A
  // second form - COALESCED 1-KiB row loads, transposed into fragments through the wave's 16 KiB of the idle gallery ring - costs the
  // SAMPLE pass +4.5 us (every block still pulls 512 KiB of fp32 rows through its CU's L2 port against 256 KiB of the image) and saves
  // the 5-us launch + its boundary: NO difference in a same-box A/B (three alternating rounds: 512 x 125k 84.6-88.4 vs 86.7-87.9 us,
  // 512 x 1M 319-321 vs 316-327, 256 x 100k 56.3-56.9 vs 55.6-56.1). The launch stays: it is the simpler code.)
  hipLaunchKernelGGL((sim_prep<TG>), dim3(p.nqg * 8 * QB), dim3(256), 0, s, Q, Bq, qimg, dflags, ovf_q, sg, Bqp, dq);
  COR_CHECK_LAUNCH();
  ScanArgs a{};
  a.Bq = Bq; a.Ng = Ng; a.nqg = p.nqg; a.qimg = qimg; a.sg = sg; a.Bqp = Bqp;
  if (p.ngroups > 0) {                                  // A. group maxima of the strided sample
    a.nsplit = p.s_nsplit; a.tiles_per_split = p.s_tiles_per_split; a.ntiles = p.s_tiles; a.tile_stride = p.s_stride;
    hipLaunchKernelGGL((sim_scan<TG, QB, true>), dim3(p.nqg * p.s_nsplit), dim3(512), lds, s, G, a);
    COR_CHECK_LAUNCH();
  }
  // B. (round 4: no sim_tau launch - the APPEND pass ranks the 32 super-group maxima in its prologue; no sample pass: all empty, tau = -inf)
  // C. full scan
  a.nsplit = p.nsplit; a.tiles_per_split = p.tiles_per_split; a.ntiles = p.tiles; a.tile_stride = 1;
  a.dq = dq; a.k = k;
  a.tau = tau; a.cnt = cnt; a.rec_s = rec_s; a.rec_g = rec_g; a.cap = p.cap; a.tau_add = (flags & 4) ? 1e30f : 0.f;
#ifdef COR_PROBES
  a.stamps = (flags & 32) ? (unsigned long long*)(w + cor_topk_workspace_bytes(Bq, Ng, k)) : nullptr;   // tools/sim_stamps.py: 2 MiB behind the workspace
  a.probe_same = ((flags & 64) ? 1 : 0) | ((flags & 128) ? 2 : 0) | ((flags & 256) ? 4 : 0);
#endif
  hipLaunchKernelGGL((sim_scan<TG, QB, false>), dim3(p.nqg * p.nsplit), dim3(512), lds, s, G, a);
  COR_CHECK_LAUNCH();
  // D. exact selection: one 256-thread block per query. (COR_TOPK_WAVE_FINAL: the one-wave-per-query kernel of the small-shard path fed
  // from the records - measured 52-59 us against 25 here: 512 streams are eight dependent gather rounds for one wave. A/B partner only.)
  if (flags & COR_TOPK_WAVE_FINAL)
    hipLaunchKernelGGL((sim_final_wave<TG, true>), dim3(Bq), dim3(64), 0, s, Q, G, cnt, nullptr, rec_s, rec_g, tau, p.nstreams, p.cap, Ng, k, g_offset,
                       out_s, out_i, dflags, ovf_q, (flags & COR_TOPK_NO_FALLBACK) ? 1 : 0);
  else
    hipLaunchKernelGGL((sim_final<TG>), dim3(Bq), dim3(256), fs_lds, s, Q, G, rec_s, rec_g, tau, cnt, p.nstreams, p.cap, Ng, k, g_offset, out_s, out_i,
                       dflags, ovf_q, (flags & COR_TOPK_NO_FALLBACK) ? 1 : 0);
  COR_CHECK_LAUNCH();
  return 0;                                             // (an overflowed query was ranked exactly inside sim_final: no second launch)
}

template <typename TG, int QB, int T>
int launch_small_t(const float* Q, const TG* G, int Bq, int Ng, int k, long long g_offset, float* out_s, long long* out_i, char* w,
                   const SmallPlan& p, int flags, hipStream_t s) {
  constexpr size_t lds = (size_t)QB * 16 * 64 * 16 + 8 * 16384;
  static DevOnce once;
  cor_max_dyn_lds((const void*)sim_block_scan<TG, QB, T>, (int)lds, once);
  int* dflags = (int*)(w + p.off_flags); int* ovf_q = (int*)(w + p.off_ovf); int* cnt = (int*)(w + p.off_cnt);
  uint2* cand = (uint2*)(w + p.off_cand);
#ifdef COR_PROBES
  unsigned long long* stamps = (flags & 32) ? (unsigned long long*)(w + cor_topk_workspace_bytes(Bq, Ng, k)) : nullptr;   // tools/sim_stamps.py: 2 MiB behind the workspace
#else
  unsigned long long* stamps = nullptr;
#endif
  hipLaunchKernelGGL((sim_block_scan<TG, QB, T>), dim3(p.grid), dim3(512), lds, s, Q, G, Bq, Ng, k, p.nqg, p.nslices, p.xcd_map, p.cap, cnt, cand, dflags, stamps);
  COR_CHECK_LAUNCH();
  hipLaunchKernelGGL((sim_final_wave<TG, false>), dim3(Bq), dim3(64), 0, s, Q, G, cnt, cand, nullptr, nullptr, nullptr, p.nslices, p.cap, Ng, k, g_offset,
                     out_s, out_i, dflags, ovf_q, (flags & COR_TOPK_NO_FALLBACK) ? 1 : 0);
  COR_CHECK_LAUNCH();
  return 0;
}
template <typename TG>
int launch_small(const float* Q, const TG* G, int Bq, int Ng, int k, long long g_offset, float* out_s, long long* out_i, char* w,
                 const SmallPlan& p, int flags, hipStream_t s) {
#define SB_GO(QB_, T_) return launch_small_t<TG, QB_, T_>(Q, G, Bq, Ng, k, g_offset, out_s, out_i, w, p, flags, s)
  if (p.qb == 1) { if (p.T == 1) SB_GO(1, 1); if (p.T == 2) SB_GO(1, 2); SB_GO(1, 4); }
  if (p.T == 1) SB_GO(2, 1);
  SB_GO(2, 2);
#undef SB_GO
}

template <typename TG>
int launch_topk(const float* Q, const void* G, int Bq, int Ng, int C, int k, long long g_offset, float* out_s, long long* out_i,
                void* workspace, int flags, hipStream_t s) {
  float* ws_s = (float*)workspace;
  const bool force_lists = (flags & COR_TOPK_FORCE_LISTS) != 0;
  if constexpr (sizeof(TG) == 2) {
    if (C == 256 && !force_lists && !(flags & COR_TOPK_FORCE_GLOBAL_THRESHOLD)) {   // small shards: two launches, local thresholds
      const SmallPlan sp = make_small(Bq, Ng, k);
      if (sp.ok) return launch_small<TG>(Q, (const TG*)G, Bq, Ng, k, g_offset, out_s, out_i, (char*)workspace, sp, flags, s);
    }
    if (C == 256 && !force_lists) {                     // threshold-and-append + exact re-scoring
      const V3Plan p = make_v3(Bq, Ng, k);
      if (p.qb == 2) return launch_v3<TG, 2>(Q, (const TG*)G, Bq, Ng, k, g_offset, out_s, out_i, (char*)workspace, p, flags, s);
      return launch_v3<TG, 1>(Q, (const TG*)G, Bq, Ng, k, g_offset, out_s, out_i, (char*)workspace, p, flags, s);
    }
    if (C == 256) return launch_lists_v2<TG>(Q, (const TG*)G, Bq, Ng, k, g_offset, out_s, out_i, ws_s, nullptr, nullptr, s);
  }
  const TopkPlan p = make_plan(Bq, Ng, k);
  int* ws_i = (int*)(ws_s + (long)Bq * p.nparts * p.kmax);
  const int nwaves = p.nqt * p.nsplit;
  if (p.kmax == 8) hipLaunchKernelGGL((sim_topk_partial<TG, 8>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  else hipLaunchKernelGGL((sim_topk_partial<TG, 32>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, Q, (const TG*)G, Bq, Ng, C, p, ws_s, ws_i);
  COR_CHECK_LAUNCH();
  return launch_merge(ws_s, ws_i, Bq, p.nparts * p.kmax, k, g_offset, out_s, out_i, nullptr, s);
}

}  // namespace

extern "C" long cor_topk_workspace_bytes(int Bq, int Ng, int k) {
  if (Bq <= 0 || Ng <= 0 || k <= 0 || k > 32) return COR_EINVAL;
  const TopkPlan p = make_plan(Bq, Ng, k);
  const TopkPlan2 p2 = make_plan2(Bq, Ng, k, device_cus());
  const long a = (long)Bq * p.nparts * p.kmax * 8, b = (long)Bq * p2.nparts * p2.kmax * 8;
  long m = a > b ? a : b;
  const long c = (long)make_v3(Bq, Ng, k).bytes;
  if (c > m) m = c;
  const long d = (long)make_small(Bq, Ng, k).bytes;
  return d > m ? d : m;
}

extern "C" int cor_similarity_topk(const float* Q, const void* G, int g_dtype, int Bq, int Ng, int C, int k, long long g_offset,
                                   float* out_scores, long long* out_idx, void* workspace, int flags, void* stream) {
  if (!Q || !G || !out_scores || !out_idx || !workspace || Bq <= 0 || Ng <= 0 || k <= 0) return COR_EINVAL;
  if (k > 32 || C > 256 || (C & 15)) return COR_ENOSUPPORT;
  if (((uintptr_t)Q & 15) || ((uintptr_t)G & 15) || ((uintptr_t)workspace & 255)) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (g_dtype) {
    case COR_F32: return launch_topk<float>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, flags, s);
    case COR_BF16: return launch_topk<bf16_t>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, flags, s);
    case COR_F16: return launch_topk<_Float16>(Q, G, Bq, Ng, C, k, g_offset, out_scores, out_idx, workspace, flags, s);
    default: return COR_ENOSUPPORT;
  }
}
