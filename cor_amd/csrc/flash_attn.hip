// cor_amd — bf16 MFMA flash attention (head_dim 64) for gfx950.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   S^T = K . Q^T   (v_mfma_f32_32x32x16_bf16, A = K tile from LDS, B = the wave's Q fragments held in registers)
// puts a query on each lane (col = lane&31), so the online softmax is lane-local plus one half-wave exchange, and
// the exponentiated tile P^T is already the B operand of   O^T += V^T . P^T   (accumulator registers 8s..8s+7 are the
// k-step-s fragment: no LDS round trip for P). V^T fragments come from the row-major V tile in LDS through
// ds_read_b64_tr_b16 (hardware transpose). K/V tiles of 64 keys are register-staged and double-buffered
// (global loads of tile t+1 are issued before the MFMAs of tile t), one barrier per tile.
//
// SAM's decomposed relative-position bias (from the UNSCALED q) is produced by the same MFMA:
//   T^T[j][q] = Rtable[j,:] . Q[q,:]  for every table row j, then  bias[q][key] = Th[qh-kh+S-1][q] + Tw[qw-kw+S-1][q].
//   global (S = 64, one key row per 64-key tile): the Tw part of all 64 key columns sits in 32 registers per lane
//   for the whole kernel, the Th part is one LDS scalar per lane per tile;
//   windowed (S = 14): both parts are looked up per score in two small per-wave LDS tables.
// Window partition, zero padding (a padded token's q/k/v is the qkv bias row) and un-partition are addressing only.
// Scores are kept in the log2 domain (scale*log2e folded in) so the exponential is a bare v_exp_f32.
#include <type_traits>

#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

struct FlashArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; void* o;
  long q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;   // MODE 0: element strides (batch, token)
  int H, Tq, Tk;
  float scale_log2;                                     // softmax scale * log2(e)
  const bf16_t* pad_row; const float* rel_h; const float* rel_w;
  int grid, S, nW, d3;                                  // SAM: image grid, rel-pos size, windows per side, 3*H*64
  int nqt;                                              // query tiles (128 queries) per (batch, head)
};

constexpr int KT = 64;                       // keys per tile
constexpr int TILE_B = KT * 128;             // one K or V tile: 64 rows x 128 B
constexpr int KV_BYTES = 4 * TILE_B;         // K0 V0 K1 V1 = 32 KiB
constexpr int AUX_PER_WAVE = 8192;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// value of lane (l ^ 32): v_permlane32_swap (VALU) instead of a ds_bpermute round trip through the LDS pipe
__device__ __forceinline__ float other_half(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = low-half values, r[1] = high-half values
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32 per pair (two scalar casts + shift/or cost 4 instructions)
__device__ __forceinline__ uint32_t pk2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint4 pack8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
  uint4 u;
  u.x = pk2(f0, f1); u.y = pk2(f2, f3); u.z = pk2(f4, f5); u.w = pk2(f6, f7);
  return u;
}

// A fragment (rows j of an fp32 [rows, 64] table, k-step c) converted to bf16
__device__ __forceinline__ uint4 table_frag(const float* tbl, int j, int c, int h) {
  const float* p = tbl + (long)j * 64 + 16 * c + 8 * h;
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  return pack8(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
}

template <int MODE, typename TO>
__global__ void __launch_bounds__(256, 2) flash_fwd(const FlashArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware placement: the nqt query tiles of one (batch, head) re-read the same K/V (1 MiB at 4096 keys); hand each
  // XCD (blocks with equal id % 8) a contiguous run of work items so those re-reads hit ITS L2 instead of HBM/MALL.
  const int wi_ = xcd_remap(blockIdx.x, gridDim.x);
  const int qt_ = wi_ % a.nqt, hb_ = wi_ / a.nqt;
  const int head = hb_ % a.H, bz = hb_ / a.H;
  const int S = a.S;

  // ---- problem geometry
  int b = bz, wy = 0, wx = 0;
  if (MODE == 2) { const int nw2 = a.nW * a.nW; b = bz / nw2; const int wi = bz - b * nw2; wy = wi / a.nW; wx = wi - wy * a.nW; }
  const int g2 = a.grid * a.grid;

  // (K,V) row pointers of key index kidx (already clamped to [0, Tk))
  auto kv_src = [&](int kidx, const bf16_t*& kp, const bf16_t*& vp) {
    if (MODE == 0) {
      kp = a.k + bz * a.k_sb + (long)kidx * a.k_st + head * 64;
      vp = a.v + bz * a.v_sb + (long)kidx * a.v_st + head * 64;
    } else {
      long row; bool ok = true;
      if (MODE == 1) row = (long)b * g2 + kidx;
      else {
        const int ky = kidx / S, kx = kidx - ky * S, y = wy * S + ky, x = wx * S + kx;
        ok = y < a.grid && x < a.grid;
        row = (long)b * g2 + (long)y * a.grid + x;
      }
      const bf16_t* base = ok ? a.q + row * a.d3 : a.pad_row;
      kp = base + a.H * 64 + head * 64;
      vp = base + 2 * a.H * 64 + head * 64;
    }
  };

  // ---- this lane's query
  int tq = qt_ * 128 + wave * 32 + r;
  bool qvalid = tq < a.Tq;
  tq = min(tq, a.Tq - 1);
  const int qh = MODE == 0 ? 0 : tq / S, qw = MODE == 0 ? 0 : tq - (tq / S) * S;
  const bf16_t* qp;
  long orow = 0;
  if (MODE == 0) qp = a.q + bz * a.q_sb + (long)tq * a.q_st + head * 64;
  else if (MODE == 1) { orow = (long)b * g2 + tq; qp = a.q + orow * a.d3 + head * 64; }
  else {
    const int y = wy * S + qh, x = wx * S + qw;
    const bool ok = y < a.grid && x < a.grid;
    qvalid = qvalid && ok;
    orow = (long)b * g2 + (long)min(y, a.grid - 1) * a.grid + min(x, a.grid - 1);
    qp = ok ? a.q + orow * a.d3 + head * 64 : a.pad_row + head * 64;
  }
  uint4 qf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qf[c] = *(const uint4*)(qp + 16 * c + 8 * h);

  // ---- relative-position tables (log2 domain)
  float* aux = (float*)(smem + KV_BYTES + wave * AUX_PER_WAVE);
  float wreg[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) wreg[i][e] = 0.f;
  if (MODE == 1) {
    float* scr = (float*)(smem + wave * AUX_PER_WAVE);            // aliases the K/V buffers: barrier before staging
#pragma unroll 1
    for (int tbl = 0; tbl < 2; ++tbl) {
      const float* table = tbl == 0 ? a.rel_h : a.rel_w;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        f32x16 acc[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[jb][e] = 0.f;
          const int j = min(64 * half + jb * 32 + r, 2 * S - 2);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, table_frag(table, j, c, h)),
                                                              __builtin_bit_cast(bf16x8, qf[c]), acc[jb], 0, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 16; ++e) scr[(jb * 32 + acc_row(e, h)) * 32 + r] = acc[jb][e] * LOG2E;
        if (tbl == 0) {
#pragma unroll 4
          for (int i = 0; i < 32; ++i) {
            const int kh = 32 * h + i, j = qh + (S - 1) - kh;
            if ((j >> 6) == half) aux[kh * 32 + r] = scr[(j & 63) * 32 + r];
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int kw = kb * 32 + acc_row(e, h), j = qw + (S - 1) - kw;
              if ((j >> 6) == half) wreg[kb][e] = scr[(j & 63) * 32 + r];
            }
        }
      }
    }
    __syncthreads();
  } else if (MODE == 2) {
#pragma unroll 1
    for (int tbl = 0; tbl < 2; ++tbl) {
      const float* table = tbl == 0 ? a.rel_h : a.rel_w;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const int j = min(r, 2 * S - 2);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, table_frag(table, j, c, h)),
                                                      __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) aux[tbl * 1024 + acc_row(e, h) * 32 + r] = acc[e] * LOG2E;
    }
    // column part of the bias for this lane's 32 score registers: constant over tiles -> registers
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int loc = min(kb * 32 + acc_row(e, h), 55);
        const int kw = loc - 14 * ((loc * 4682) >> 16);
        wreg[kb][e] = aux[1024 + (qw - kw + 13) * 32 + r];
      }
  }

  // ---- staging map. Windowed mode: a tile is 4 whole key rows of the 14x14 window (56 keys; LDS rows 56..63 are
  // loaded but masked), so a score's (key row, key column) inside the tile is a compile-time property of its register.
  constexpr int TSTRIDE = MODE == 2 ? 56 : KT;
  const int srow = tid >> 3, sch = tid & 7;                 // rows srow and srow+32
  const int k_st0 = srow * 128 + ((sch ^ ((srow >> 1) & 7)) << 4);
  const int k_st1 = (srow + 32) * 128 + ((sch ^ (((srow + 32) >> 1) & 7)) << 4);
  const int v_st0 = srow * 128 + sch * 16, v_st1 = (srow + 32) * 128 + sch * 16;
  uint4 rk0, rk1, rv0, rv1;
#define FA_GLOAD(T_)                                                                   \
  {                                                                                    \
    const bf16_t *kp_, *vp_;                                                           \
    kv_src(min((T_) * TSTRIDE + srow, a.Tk - 1), kp_, vp_);                            \
    rk0 = *(const uint4*)(kp_ + sch * 8); rv0 = *(const uint4*)(vp_ + sch * 8);        \
    kv_src(min((T_) * TSTRIDE + srow + 32, a.Tk - 1), kp_, vp_);                       \
    rk1 = *(const uint4*)(kp_ + sch * 8); rv1 = *(const uint4*)(vp_ + sch * 8);        \
  }
#define FA_LSTORE(BUF_)                                                                \
  {                                                                                    \
    char* Ks_ = smem + (BUF_) * 2 * TILE_B; char* Vs_ = Ks_ + TILE_B;                  \
    *(uint4*)(Ks_ + k_st0) = rk0; *(uint4*)(Ks_ + k_st1) = rk1;                        \
    *(uint4*)(Vs_ + v_st0) = rv0; *(uint4*)(Vs_ + v_st1) = rv1;                        \
  }

  // ---- fragment read maps
  const int sw = (lane >> 1) & 7;
  int kch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kch[c] = r * 128 + (((2 * c + h) ^ sw) << 4);
  // transposed V read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block
  const int v_tr = (4 * h + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int nt = (a.Tk + TSTRIDE - 1) / TSTRIDE;
  // one K/V tile; TAIL = the last, partially filled tile (the only one that needs per-key masking)
  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const bool more = t + 1 < nt;
    if (more) FA_GLOAD(t + 1)
    const char* Ks = smem + (t & 1) * 2 * TILE_B; const char* Vs = Ks + TILE_B;

    // S^T = K . Q^T
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 kf = *(const uint4*)(Ks + kb * 32 * 128 + kch[c]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[c]), s[kb], 0, 0, 0);
      }
    }
    // logits in the log2 domain. MODE 1: the per-tile row term rh is the same for all 64 keys of the tile, so it is
    // folded into the running-max bookkeeping instead of being added to 32 registers. MODE 2: the tile holds key rows
    // 4t..4t+3; the row term is one of 4 per-tile LDS scalars, selected per register at compile time (per lane half).
    const float rh = MODE == 1 ? aux[t * 32 + r] : 0.f;
    float thv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) thv[j] = aux[max(qh - (4 * t + j) + 13, 0) * 32 + r];
    }
    float mloc = -INFINITY;
    if (MODE != 2) {                                     // scale (+ column bias) two registers per v_pk_fma_f32
      const f32x2_t sc2 = {a.scale_log2, a.scale_log2};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          f32x2_t v = {s[kb][e], s[kb][e + 1]};
          const f32x2_t w = {MODE == 1 ? wreg[kb][e] : 0.f, MODE == 1 ? wreg[kb][e + 1] : 0.f};
          v = v * sc2 + w;
          s[kb][e] = v[0]; s[kb][e + 1] = v[1];
        }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float x;
        if (MODE == 2) {
          const int l0 = kb * 32 + (e & 3) + 8 * (e >> 2), l1 = l0 + 4;      // key inside the tile for h = 0 / 1
          const int lim = TAIL ? 28 : 56;                                    // last tile: key rows 12, 13 only
          if (l0 >= lim) x = -INFINITY;                                      // (then l1 >= lim too)
          else {
            const float th0 = thv[l0 / 14], th1 = thv[l1 < 56 ? l1 / 14 : 0];
            x = fmaf(s[kb][e], a.scale_log2, wreg[kb][e]) + (h ? th1 : th0);
            if (l1 >= lim && h) x = -INFINITY;
          }
        } else {
          x = s[kb][e];                                  // already scaled + biased (packed, below)
          if (TAIL && t * KT + kb * 32 + acc_row(e, h) >= a.Tk) x = -INFINITY;
        }
        s[kb][e] = x;
        mloc = fmaxf(mloc, x);
      }
    mloc = fmaxf(mloc, other_half(mloc)) + rh;  // true tile max (x + rh)
    const float mnew = fmaxf(m, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    const float msub = mnew - rh;                       // p = 2^(x + rh - mnew)
    m = mnew;
    l *= alpha;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    uint4 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 p;
      const f32x2_t ms2 = {msub, msub};
      f32x2_t lacc = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2_t d = f32x2_t{s[kb][e], s[kb][e + 1]} - ms2;
        const f32x2_t pe = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
        p[e] = pe[0]; p[e + 1] = pe[1];
        lacc += pe;
      }
      l += lacc[0] + lacc[1];
      pf[kb][0] = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
      pf[kb][1] = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
    }
    // O^T += V^T . P^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int krow = kb * 32 + ks * 16;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vb = Vs + krow * 128 + db * 64 + v_tr;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
          const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
          const uint4 vf = make_uint4(u0.x, u0.y, u1.x, u1.y);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[kb][ks]), o[db], 0, 0, 0);
        }
      }
    if (more) FA_LSTORE((t + 1) & 1)
    __syncthreads();
  };
  FA_GLOAD(0)
  FA_LSTORE(0)
  __syncthreads();
  const int nfull = (a.Tk % TSTRIDE) ? nt - 1 : nt;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < nt) tile(nt - 1, std::true_type{});

  l += other_half(l);
  if (!qvalid) return;
  const float inv = 1.0f / l;
  TO* op = MODE == 0 ? (TO*)a.o + bz * a.o_sb + (long)tq * a.o_st + head * 64 : (TO*)a.o + orow * (long)(a.H * 64) + head * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v4 = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
      st4<TO>(op + db * 32 + 8 * g + 4 * h, v4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Global SAM attention, 64 queries per wave (two 32-query blocks), ONE wave per SIMD (block = 4 waves = 256 queries).
// Every K fragment (ds_read_b128) and every transposed V fragment (2 x ds_read_b64_tr_b16) now feeds TWO MFMAs, which
// halves LDS traffic and barriers per flop, and the two query blocks give each wave two independent MFMA->softmax->MFMA
// chains, so the matrix pipe works on one block while the VALU runs the other block's softmax (with 32 queries per wave
// the loop was barrier/latency bound: MFMA busy 25 %, waves waiting 54 % of their cycles). Needs ~300 registers, hence
// one wave per SIMD; LDS = 32 KiB K/V double buffer + 4 x 16 KiB row-bias tables.
template <typename TO>
__global__ void __launch_bounds__(256, 1) flash_global64(const FlashArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QB = 2, AUX64 = 2 * AUX_PER_WAVE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi_ = xcd_remap(blockIdx.x, gridDim.x);
  const int qt_ = wi_ % a.nqt, hb_ = wi_ / a.nqt;
  const int head = hb_ % a.H, b = hb_ / a.H;
  const int S = a.S;                                   // 64
  const int g2 = a.grid * a.grid;
  const bf16_t* kvbase = a.q + (long)b * g2 * a.d3 + head * 64;

  int qh[QB], qw[QB]; long orow[QB]; bool qvalid[QB];
  uint4 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int tq = qt_ * 256 + wave * 64 + qb * 32 + r;
    qvalid[qb] = tq < a.Tq;
    tq = min(tq, a.Tq - 1);
    qh[qb] = tq / S; qw[qb] = tq - qh[qb] * S;
    orow[qb] = (long)b * g2 + tq;
    const bf16_t* qp = a.q + orow[qb] * a.d3 + head * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[qb][c] = *(const uint4*)(qp + 16 * c + 8 * h);
  }

  // ---- relative-position tables (log2 domain): Th part -> aux[kh][qb*32 + q], Tw part -> registers
  float* aux = (float*)(smem + KV_BYTES + wave * AUX64);
  float* scr = (float*)(smem + wave * AUX_PER_WAVE);   // aliases the K/V buffers: barrier before staging
  float wreg[QB][2][16];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll 1
    for (int tbl = 0; tbl < 2; ++tbl) {
      const float* table = tbl == 0 ? a.rel_h : a.rel_w;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        f32x16 acc[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[jb][e] = 0.f;
          const int j = min(64 * half + jb * 32 + r, 2 * S - 2);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, table_frag(table, j, c, h)),
                                                              __builtin_bit_cast(bf16x8, qf[qb][c]), acc[jb], 0, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 16; ++e) scr[(jb * 32 + acc_row(e, h)) * 32 + r] = acc[jb][e] * LOG2E;
        if (tbl == 0) {
#pragma unroll 4
          for (int i = 0; i < 32; ++i) {
            const int kh = 32 * h + i, j = qh[qb] + (S - 1) - kh;
            if ((j >> 6) == half) aux[kh * 64 + qb * 32 + r] = scr[(j & 63) * 32 + r];
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int kw = kb * 32 + acc_row(e, h), j = qw[qb] + (S - 1) - kw;
              if ((j >> 6) == half) wreg[qb][kb][e] = scr[(j & 63) * 32 + r];
            }
        }
      }
    }
  }
  __syncthreads();

  // ---- staging (register-staged double buffering, as flash_fwd)
  const int srow = tid >> 3, sch = tid & 7;
  const int k_st0 = srow * 128 + ((sch ^ ((srow >> 1) & 7)) << 4);
  const int k_st1 = (srow + 32) * 128 + ((sch ^ (((srow + 32) >> 1) & 7)) << 4);
  const int v_st0 = srow * 128 + sch * 16, v_st1 = (srow + 32) * 128 + sch * 16;
  uint4 rk0, rk1, rv0, rv1;
#define G64_GLOAD(T_)                                                                              \
  {                                                                                                \
    const bf16_t* p0_ = kvbase + (long)((T_) * KT + srow) * a.d3 + sch * 8;                        \
    const bf16_t* p1_ = p0_ + 32L * a.d3;                                                          \
    rk0 = *(const uint4*)(p0_ + a.H * 64); rv0 = *(const uint4*)(p0_ + 2 * a.H * 64);             \
    rk1 = *(const uint4*)(p1_ + a.H * 64); rv1 = *(const uint4*)(p1_ + 2 * a.H * 64);             \
  }
#define G64_LSTORE(BUF_)                                                                           \
  {                                                                                                \
    char* Ks_ = smem + (BUF_) * 2 * TILE_B; char* Vs_ = Ks_ + TILE_B;                              \
    *(uint4*)(Ks_ + k_st0) = rk0; *(uint4*)(Ks_ + k_st1) = rk1;                                    \
    *(uint4*)(Vs_ + v_st0) = rv0; *(uint4*)(Vs_ + v_st1) = rv1;                                    \
  }
  const int sw = (lane >> 1) & 7;
  int kch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kch[c] = r * 128 + (((2 * c + h) ^ sw) << 4);
  const int v_tr = (4 * h + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  f32x16 o[QB][2];
  float m[QB], l[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m[qb] = -INFINITY; l[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] = 0.f;
  }
  const int nt = a.Tk / KT;                            // 4096 / 64, no tail
  G64_GLOAD(0)
  G64_LSTORE(0)
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;
    if (more) G64_GLOAD(t + 1)
    const char* Ks = smem + (t & 1) * 2 * TILE_B; const char* Vs = Ks + TILE_B;
    // S^T = K . Q^T for both query blocks: one fragment read, two MFMAs
    f32x16 s[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[qb][kb][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 kf = *(const uint4*)(Ks + kb * 32 * 128 + kch[c]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[qb][c]), s[qb][kb], 0, 0, 0);
      }
    uint4 pf[QB][2][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float rh = aux[t * 64 + qb * 32 + r];
      const f32x2_t sc2 = {a.scale_log2, a.scale_log2};
      float mloc = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          f32x2_t v = {s[qb][kb][e], s[qb][kb][e + 1]};
          const f32x2_t w = {wreg[qb][kb][e], wreg[qb][kb][e + 1]};
          v = v * sc2 + w;
          s[qb][kb][e] = v[0]; s[qb][kb][e + 1] = v[1];
          mloc = fmaxf(mloc, fmaxf(v[0], v[1]));
        }
      mloc = fmaxf(mloc, other_half(mloc)) + rh;
      const float mnew = fmaxf(m[qb], mloc);
      const float alpha = __builtin_amdgcn_exp2f(m[qb] - mnew);
      const float msub = mnew - rh;
      m[qb] = mnew;
      l[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[qb][db][e] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 p;
        const f32x2_t ms2 = {msub, msub};
        f32x2_t lacc = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const f32x2_t d = f32x2_t{s[qb][kb][e], s[qb][kb][e + 1]} - ms2;
          const f32x2_t pe = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          p[e] = pe[0]; p[e + 1] = pe[1];
          lacc += pe;
        }
        l[qb] += lacc[0] + lacc[1];
        pf[qb][kb][0] = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
        pf[qb][kb][1] = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
      }
    }
    // O^T += V^T . P^T: one transposed fragment, two MFMAs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int krow = kb * 32 + ks * 16;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vb = Vs + krow * 128 + db * 64 + v_tr;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
          const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
          const uint4 vf = make_uint4(u0.x, u0.y, u1.x, u1.y);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[qb][kb][ks]), o[qb][db], 0, 0, 0);
        }
      }
    if (more) G64_LSTORE((t + 1) & 1)
    __syncthreads();
  }
#undef G64_GLOAD
#undef G64_LSTORE
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float lt = l[qb] + other_half(l[qb]);
    if (!qvalid[qb]) continue;
    const float inv = 1.0f / lt;
    TO* op = (TO*)a.o + orow[qb] * (long)(a.H * 64) + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4 = {o[qb][db][4 * g] * inv, o[qb][db][4 * g + 1] * inv, o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv};
        st4<TO>(op + db * 32 + 8 * g + 4 * h, v4);
      }
  }
}

template <typename TO>
int launch_global64(const FlashArgs& a, int nb, hipStream_t s) {
  const size_t lds = KV_BYTES + 4 * 2 * AUX_PER_WAVE;  // 96 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)flash_global64<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  FlashArgs b = a;
  b.nqt = cdiv(a.Tq, 256);
  hipLaunchKernelGGL((flash_global64<TO>), dim3(b.nqt * a.H * nb), dim3(256), lds, s, b);
  COR_CHECK_LAUNCH();
  return 0;
}

template <int MODE, typename TO>
int launch(const FlashArgs& a, int nb, hipStream_t s) {
  const size_t lds = KV_BYTES + (MODE == 0 ? 0 : 4 * AUX_PER_WAVE);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)flash_fwd<MODE, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  FlashArgs b = a;
  b.nqt = cdiv(a.Tq, 128);
  hipLaunchKernelGGL((flash_fwd<MODE, TO>), dim3(b.nqt * a.H * nb), dim3(256), lds, s, b);
  COR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int g_flash_global_variant = 0;      // 0 (default): 32 queries per wave, 2 waves per SIMD (2.66 ms at B=32); 1: flash_global64 (3.37 ms: hipcc does not overlap the two chains)
extern "C" int cor_flash_set_variant(int v) { g_flash_global_variant = v ? 1 : 0; return 0; }

int cor_flash_plain_bf16(const void* q, long q_sb, long q_st, const void* k, long k_sb, long k_st, const void* v, long v_sb, long v_st,
                         void* out, long o_sb, long o_st, int out_dtype, int B, int H, int Tq, int Tk, float scale, hipStream_t s) {
  // 16-B fragment loads: every row start must be 16-B aligned
  if ((q_st | k_st | v_st | q_sb | k_sb | v_sb) & 7) return COR_ENOSUPPORT;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return COR_ENOSUPPORT;
  if ((o_st | o_sb) & 3) return COR_ENOSUPPORT;
  FlashArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = out;
  a.q_sb = q_sb; a.q_st = q_st; a.k_sb = k_sb; a.k_st = k_st; a.v_sb = v_sb; a.v_st = v_st; a.o_sb = o_sb; a.o_st = o_st;
  a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale_log2 = scale * LOG2E; a.S = 1; a.grid = 1; a.nW = 1;
  if (out_dtype == COR_BF16) return launch<0, bf16_t>(a, B, s);
  if (out_dtype == COR_F32) return launch<0, float>(a, B, s);
  return COR_ENOSUPPORT;
}

int cor_flash_sam_bf16(const void* qkv, void* out, int out_dtype, const void* pad_row, const float* rel_h, const float* rel_w, int B,
                       int H, int grid, int window, hipStream_t s) {
  if (((uintptr_t)qkv & 15) || ((uintptr_t)rel_h & 15) || ((uintptr_t)rel_w & 15)) return COR_ENOSUPPORT;
  FlashArgs a{};
  a.q = (const bf16_t*)qkv; a.o = out; a.H = H; a.scale_log2 = 0.125f * LOG2E;
  a.pad_row = (const bf16_t*)pad_row; a.rel_h = rel_h; a.rel_w = rel_w; a.grid = grid; a.d3 = 3 * H * 64;
  if (window == 0) {
    if (grid != 64) return COR_ENOSUPPORT;            // one key row per 64-key tile
    a.S = 64; a.Tq = a.Tk = grid * grid; a.nW = 1;
    if (g_flash_global_variant == 1) {
      if (out_dtype == COR_BF16) return launch_global64<bf16_t>(a, B, s);
      if (out_dtype == COR_F32) return launch_global64<float>(a, B, s);
    }
    if (out_dtype == COR_BF16) return launch<1, bf16_t>(a, B, s);
    if (out_dtype == COR_F32) return launch<1, float>(a, B, s);
    return COR_ENOSUPPORT;
  }
  if (window != 14 || ((uintptr_t)pad_row & 15)) return COR_ENOSUPPORT;
  a.S = 14; a.Tq = a.Tk = 196; a.nW = (grid + 13) / 14;
  if (out_dtype == COR_BF16) return launch<2, bf16_t>(a, B * a.nW * a.nW, s);
  if (out_dtype == COR_F32) return launch<2, float>(a, B * a.nW * a.nW, s);
  return COR_ENOSUPPORT;
}
