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
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

struct FlashArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; void* o;
  long q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;   // MODE 0: element strides (batch, token)
  int H, Tq, Tk;
  float scale_log2;                                     // what (q . k) is multiplied by: softmax scale * log2(e) / q_prescale
  float tbl_scale;                                      // what (Rtable . q) is multiplied by: log2(e) / q_prescale
  const bf16_t* pad_row; const float* rel_h; const float* rel_w;
  int grid, S, nW, d3;                                  // SAM: image grid, rel-pos size, windows per side, 3*H*64
  int nqt;                                              // query tiles (128 queries) per (batch, head)
  int rev;                                              // 1: work items from the last to the first (COR_ORDER_REVERSE)
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

// acc + both bf16 halves of a packed pair (v_dot2c_f32_bf16 against (1, 1)): the row sum of P costs one instruction per TWO keys; it
// is taken over the bf16-rounded P, the values the PV product uses
__device__ __forceinline__ float sum2_bf16(uint32_t pair, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pair), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc, false);
}
__device__ __forceinline__ float max3f_(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// A fragment (rows j of an fp32 [rows, 64] table, k-step c) converted to bf16
__device__ __forceinline__ uint4 table_frag(const float* tbl, int j, int c, int h) {
  const float* p = tbl + (long)j * 64 + 16 * c + 8 * h;
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  return pack8(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
}

// Output store. O^T leaves a query on each lane (32 queries x HD values per wave): storing it from there is 8-byte stores that
// touch 32 different output rows per instruction - a store-ISSUE-bound tail (MI355X guide: ~9k cycles; 61 % of a windowed
// block's lifetime was prologue + this tail). Instead the wave transposes its tile through its own LDS (row stride = row bytes
// + 16: conflict-free for both passes) and each instruction stores whole rows. HD = 64 (2 column blocks of 32), 72 or 80
// (3 blocks, the third partly empty). off(j) = element offset of local query j's output row, or -1 when that query does not exist.
template <typename TO, int HD, typename OFF>
__device__ __forceinline__ void store_o_rows(const f32x16 (&o)[(HD + 31) / 32], float inv, char* stg, TO* out, int lane, OFF&& off) {
  constexpr int DB = (HD + 31) / 32;
  const int r = lane & 31, h = lane >> 5;
  if constexpr (sizeof(TO) == 2) {
    constexpr int SR = HD * 2 + 16, NCHO = HD * 2 / 16;           // staging row stride (B), 16-B pieces per output row
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (db * 32 + 8 * g + 8 <= HD) {                            // columns db*32 + 8g + 4h .. +3 exist (same for both halves)
          uint2 u;
          u.x = pk2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv); u.y = pk2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *(uint2*)(stg + r * SR + (db * 32 + 8 * g + 4 * h) * 2) = u;
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // lanes exchange data through LDS (compiler-only fence)
#pragma unroll
    for (int i = 0; i < (32 * NCHO + 63) / 64; ++i) {
      const int idx = lane + 64 * i, j = idx / NCHO, ch = idx - j * NCHO;
      if (idx < 32 * NCHO) {
        const uint4 v = *(const uint4*)(stg + j * SR + ch * 16);
        const long e = off(j);
        if (e >= 0) *(uint4*)(out + e + ch * 8) = v;
      }
    }
  } else {
#pragma unroll
    for (int db = 0; db < DB; ++db) {                   // fp32 rows: one 128-B block of 32 columns per pass
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (db * 32 + 8 * g + 8 <= HD) {
          const f32x4 v4 = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
          *(f32x4*)(stg + r * 144 + (8 * g + 4 * h) * 4) = v4;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = (lane >> 3) + 8 * i, ch = lane & 7;
        const f32x4 v = *(const f32x4*)(stg + j * 144 + ch * 16);
        const long e = off(j);
        if (e >= 0 && db * 32 + 4 * ch + 4 <= HD) *(f32x4*)(out + e + db * 32 + ch * 4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// A fragment of an fp32 [rows, HD] rel-pos table for K-step c, zero beyond HD (HD = 72: the fifth K-step is half empty)
template <int HD>
__device__ __forceinline__ uint4 table_frag_hd(const float* tbl, int j, int c, int h) {
  if (16 * c + 8 * h + 8 > HD) return make_uint4(0, 0, 0, 0);
  const float* p = tbl + (long)j * HD + 16 * c + 8 * h;
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  return pack8(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
}

// HD = 64 (SAM-B/L, SigLIP-B/L), 72 (SigLIP SO400M/14: the factory's default tower) or 80 (SAM-H). For 72 / 80 the score
// product runs 5 K-steps (the operand rows are zero-padded to 80 values) and O^T has 3 column blocks of 32; K / V tiles sit in
// LDS with 208-byte rows (13 sixteen-byte slots: any 16 consecutive rows hit 16 distinct slots, so no swizzle is needed).
template <int MODE, typename TO, int HD = 64>
__global__ void __launch_bounds__(256, 2) flash_fwd(const FlashArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = (HD + 15) / 16, DB = (HD + 31) / 32, NCH = HD * 2 / 16, NCHP = 2 * KS;
  constexpr int ROWB = HD == 64 ? 128 : 208, TILEB = KT * ROWB, KVB = 4 * TILEB;
  constexpr int NP = (KT * NCHP + 255) / 256;               // staging passes of 256 sixteen-byte pieces
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware placement: the nqt query tiles of one (batch, head) re-read the same K/V (1 MiB at 4096 keys); hand each
  // XCD (blocks with equal id % 8) a contiguous run of work items so those re-reads hit ITS L2 instead of HBM/MALL.
  const int wi0_ = xcd_remap(blockIdx.x, gridDim.x);
  const int wi_ = a.rev ? (int)gridDim.x - 1 - wi0_ : wi0_;
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
      kp = a.k + bz * a.k_sb + (long)kidx * a.k_st + head * HD;
      vp = a.v + bz * a.v_sb + (long)kidx * a.v_st + head * HD;
    } else {
      long row; bool ok = true;
      if (MODE == 1) row = (long)b * g2 + kidx;
      else {
        const int ky = kidx / S, kx = kidx - ky * S, y = wy * S + ky, x = wx * S + kx;
        ok = y < a.grid && x < a.grid;
        row = (long)b * g2 + (long)y * a.grid + x;
      }
      const bf16_t* base = ok ? a.q + row * a.d3 : a.pad_row;
      kp = base + a.H * HD + head * HD;
      vp = base + 2 * a.H * HD + head * HD;
    }
  };

  // ---- this lane's query
  int tq = qt_ * 128 + wave * 32 + r;
  bool qvalid = tq < a.Tq;
  tq = min(tq, a.Tq - 1);
  const int qh = MODE == 0 ? 0 : tq / S, qw = MODE == 0 ? 0 : tq - (tq / S) * S;
  const bf16_t* qp;
  long orow = 0;
  if (MODE == 0) qp = a.q + bz * a.q_sb + (long)tq * a.q_st + head * HD;
  else if (MODE == 1) { orow = (long)b * g2 + tq; qp = a.q + orow * a.d3 + head * HD; }
  else {
    const int y = wy * S + qh, x = wx * S + qw;
    const bool ok = y < a.grid && x < a.grid;
    qvalid = qvalid && ok;
    orow = (long)b * g2 + (long)min(y, a.grid - 1) * a.grid + min(x, a.grid - 1);
    qp = ok ? a.q + orow * a.d3 + head * HD : a.pad_row + head * HD;
  }
  uint4 qf[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) qf[c] = (16 * c + 8 * h + 8 <= HD) ? *(const uint4*)(qp + 16 * c + 8 * h) : make_uint4(0, 0, 0, 0);

  // ---- staging map. Windowed mode: a tile is 4 whole key rows of the 14x14 window (56 keys; LDS rows 56..63 are
  // loaded but masked), so a score's (key row, key column) inside the tile is a compile-time property of its register.
  // Piece idx = tid + 256 p covers LDS row idx / NCHP, 16-byte slot idx % NCHP; slots >= NCH (HD = 72: slot 9) are zero.
  constexpr int TSTRIDE = MODE == 2 ? 56 : KT;
  int st_row[NP], st_ch[NP], st_k[NP], st_v[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + 256 * p, row = idx / NCHP, ch = idx - row * NCHP;
    st_row[p] = row; st_ch[p] = ch;
    st_k[p] = row * ROWB + (HD == 64 ? ((ch ^ ((row >> 1) & 7)) << 4) : ch * 16);
    st_v[p] = row * ROWB + ch * 16;
  }
  uint4 rk[NP], rv[NP];
#define FA_GLOAD(T_)                                                                         \
  {                                                                                          \
    _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) {                                      \
      rk[p_] = make_uint4(0, 0, 0, 0); rv[p_] = make_uint4(0, 0, 0, 0);                      \
      if ((KT * NCHP) % 256 == 0 || tid + 256 * p_ < KT * NCHP) {                            \
        const bf16_t *kp_, *vp_;                                                             \
        kv_src(min((T_) * TSTRIDE + st_row[p_], a.Tk - 1), kp_, vp_);                        \
        if (NCH == NCHP || st_ch[p_] < NCH) { rk[p_] = *(const uint4*)(kp_ + st_ch[p_] * 8); rv[p_] = *(const uint4*)(vp_ + st_ch[p_] * 8); } \
      }                                                                                      \
    }                                                                                        \
  }
#define FA_LSTORE(BUF_)                                                                      \
  {                                                                                          \
    char* Ks_ = smem + (BUF_) * 2 * TILEB; char* Vs_ = Ks_ + TILEB;                          \
    _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) {                                      \
      if ((KT * NCHP) % 256 == 0 || tid + 256 * p_ < KT * NCHP) { *(uint4*)(Ks_ + st_k[p_]) = rk[p_]; *(uint4*)(Vs_ + st_v[p_]) = rv[p_]; } \
    }                                                                                        \
  }
  // tile 0's global loads are issued here, before the rel-pos table work, so that their latency (~2 us, a fifth of a
  // windowed block's lifetime) overlaps it; the registers are written to LDS after the tables (which alias the K/V buffers).
  FA_GLOAD(0)

  // ---- relative-position tables (log2 domain)
  float* aux = (float*)(smem + KVB + wave * AUX_PER_WAVE);
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
          for (int c = 0; c < KS; ++c)
            acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, table_frag_hd<HD>(table, j, c, h)),
                                                              __builtin_bit_cast(bf16x8, qf[c]), acc[jb], 0, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 16; ++e) scr[(jb * 32 + acc_row(e, h)) * 32 + r] = acc[jb][e] * a.tbl_scale;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
      for (int c = 0; c < KS; ++c)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, table_frag_hd<HD>(table, j, c, h)),
                                                      __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) aux[tbl * 1024 + acc_row(e, h) * 32 + r] = acc[e] * a.tbl_scale;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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

  // ---- fragment read maps
  const int sw = (lane >> 1) & 7;
  int kch[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) kch[c] = r * ROWB + (HD == 64 ? (((2 * c + h) ^ sw) << 4) : (2 * c + h) * 16);
  // transposed V read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block
  const int v_tr = (4 * h + ((lane & 15) >> 2)) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -INFINITY;
  f32x16 lsum;
#pragma unroll
  for (int e = 0; e < 16; ++e) lsum[e] = 0.f;
  const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);   // eight bf16 1.0

  const int nt = (a.Tk + TSTRIDE - 1) / TSTRIDE;
  // one K/V tile; TAIL = the last, partially filled tile (the only one that needs per-key masking)
  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const bool more = t + 1 < nt;
    if (more) FA_GLOAD(t + 1)
    const char* Ks = smem + (t & 1) * 2 * TILEB; const char* Vs = Ks + TILEB;

    // S^T = K . Q^T. All K fragments of a 32-key block are read before its MFMAs (a read placed right before its MFMA exposes the
    // LDS latency once per MFMA; see flash_global_pipe).
    f32x16 s[2];
    uint4 kfr[2 * KS];
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) kfr[i] = *(const uint4*)(Ks + (i / KS) * 32 * ROWB + kch[i % KS]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
      for (int c = 0; c < KS; ++c)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfr[kb * KS + c]), __builtin_bit_cast(bf16x8, qf[c]), s[kb], 0, 0, 0);
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
    // Scalar f32 VALU on purpose (and -fno-slp-vectorize for this file): beside MFMAs a v_pk_add/mul/fma_f32 costs 17-30
    // cycles of issue against 4 per scalar op (MI355X guide, cycle constants), and hipcc packs adjacent scalar ops by itself.
    if (MODE != 2) {                                     // scale (+ column bias)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kb][e] = MODE == 1 ? fmaf(s[kb][e], a.scale_log2, wreg[kb][e]) : s[kb][e] * a.scale_log2;
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
    // Lazy rescaling: the reference point m only moves when some row's tile max exceeds it by more than 8 (log2 domain), so
    // p <= 256 and the 33 multiplies of the O / l rescale run a few times per row instead of once per tile; O / l is unchanged
    // mathematically (same m in numerator and denominator). The softmax VALU, not the MFMA, bounds these kernels.
    if (__builtin_amdgcn_ballot_w64(mloc > m + 8.0f) != 0) {
      const float mnew = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m - mnew);
      m = mnew;
      lsum[1] *= alpha; lsum[2] *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    }
    const float msub = m - rh;                          // p = 2^(x + rh - m)
    uint4 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 p;
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(s[kb][e] - msub);
      // row sums on the VALU (this lane's 16 keys of the block; the lane halves are added once at the end): beside MFMAs a
      // vector instruction costs ~2 cycles, a ones-row MFMA 32 (profiles/archive/r02_valu_mfma_issue_probe.jsonl)
      lsum[1] += (p[0] + p[1]) + (p[2] + p[3]); lsum[2] += (p[4] + p[5]) + (p[6] + p[7]);
      lsum[1] += (p[8] + p[9]) + (p[10] + p[11]); lsum[2] += (p[12] + p[13]) + (p[14] + p[15]);
      pf[kb][0] = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
      pf[kb][1] = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
    }
    // O^T += V^T . P^T, one 16-key K-step at a time (its DB transposed V fragments first)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kb = kk >> 1, ks = kk & 1;
      uint4 vfr[DB];
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const char* vb = Vs + (kb * 32 + ks * 16) * ROWB + db * 64 + v_tr;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * ROWB));
        const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
        vfr[db] = make_uint4(u0.x, u0.y, u1.x, u1.y);
      }
#pragma unroll
      for (int db = 0; db < DB; ++db)
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfr[db]), __builtin_bit_cast(bf16x8, pf[kb][ks]), o[db], 0, 0, 0);
    }
    if (more) FA_LSTORE((t + 1) & 1)
    __syncthreads();
  };
  FA_LSTORE(0)
  __syncthreads();
  const int nfull = (a.Tk % TSTRIDE) ? nt - 1 : nt;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < nt) tile(nt - 1, std::true_type{});
#undef FA_GLOAD
#undef FA_LSTORE

  const float lhalf = lsum[1] + lsum[2];
  const float inv = 1.0f / (lhalf + other_half(lhalf)); // full row sum: both lane halves
  // staging: the K/V buffers (every wave is past the last tile's barrier), 8 KiB per wave
  auto off = [&](int j) -> long {
    const int tj = qt_ * 128 + wave * 32 + j;
    if (tj >= a.Tq) return -1;
    if (MODE == 0) return bz * a.o_sb + (long)tj * a.o_st + head * HD;
    if (MODE == 1) return ((long)b * g2 + tj) * (long)(a.H * HD) + head * HD;
    const int y = wy * S + tj / S, x = wx * S + (tj - (tj / S) * S);
    if (y >= a.grid || x >= a.grid) return -1;
    return ((long)b * g2 + (long)y * a.grid + x) * (long)(a.H * HD) + head * HD;
  };
  store_o_rows<TO, HD>(o, inv, smem + wave * 8192, (TO*)a.o, lane, off);
}

// ---------------------------------------------------------------------------------------------------------------------
// Global SAM attention, SOFTWARE-PIPELINED over key tiles (variant 1, default). flash_fwd<1> runs each wave's tile as a chain
// QK^T MFMAs -> softmax VALU -> PV MFMAs, so the matrix pipe idles during the softmax and the VALU during the MFMAs.
// Here iteration t interleaves, in ONE instruction stream (an MFMA holds the VALU issue for 8 of its 32 cycles),
//     phase A:  O^T += V(t-1)^T . P(t-1)^T  (8 MFMAs)                     with   scale + column bias + max of S(t)  (VALU)
//     phase B:  S(t+1)^T = K(t+1) . Q^T     (8 MFMAs)                     with   exp2 and bf16 packing of P(t)       (VALU)
// pinned with sched_group_barrier; every fragment of a phase is read from LDS at the top of the phase (a read placed right
// before its MFMA exposed ~130 cycles of LDS latency sixteen times per tile).
//   * row sums of P on the VALU (32 adds per tile and lane, lane halves added once at the end). Round 1 had them on the matrix
//     pipe (ones . P^T, 4 MFMAs per tile); tools/probes/valu_probe.hip then measured what an instruction costs here: an MFMA 32
//     cycles, a vector instruction between MFMAs ~2 - so 128 cycles bought 70 (measured: 2.29 -> 2.21 ms; variant 6 = old form);
//     the sum is over the fp32 P before its bf16 rounding;
//   * lazy rescaling: the reference point m only moves when some row's tile max exceeds it by more than 8 (log2 domain), so
//     p <= 256 and the 33 multiplies of the O / l rescale run a few times per row, not once per tile (same m in numerator
//     and denominator: O / l is mathematically unchanged);
//   * K / V tiles by LDS-DMA into rings of three 8-KiB tiles each, issued TWO iterations before use (a register-staged tile
//     had one iteration, about its own global-load latency); one counted wait and one barrier per iteration.
// LDS = 48 KiB rings + 32 KiB row-bias tables = 80 KiB: two blocks per CU. Tile -1 is a zero V tile with P = 0; the last
// iteration's S(nt) comes from a stale K tile and is dropped.
// Variants tried and measured slower (round 2, profiles/archive/r02_attention_fold_ablation.jsonl, r02_attention_pingpong_stamps.jsonl; their
// code is in the history up to commit 5d3e0f7): bias / running reference as extra k-steps of the score MFMA (+1 % / +12 %), 8-wave
// blocks (+5 %), an explicit ping-pong form with alternating matrix / softmax segments (+5 %), row sums of P by a ones-row MFMA (+4 %).
// STAMP (COR_PROBES builds only, tools/attn_stamps.py): per-section cycle sums are written INSTEAD of the outputs.
// CB (round 5): q arrives pre-scaled (scale_log2 == 1), so the column bias is the C OPERAND of the first score MFMA of each chain
// (accumulator start value) instead of one fma per score: 32 vector instructions fewer per tile and wave.
template <typename TO, bool STAMP = false, bool CB = false>
__global__ void __launch_bounds__(256, 2) flash_global_pipe(const FlashArgs a) {   // 4 waves x 32 queries per block
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K_BYTES = 3 * TILE_B, V_BYTES = 3 * TILE_B;         // rings of three 8-KiB tiles each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi0_ = xcd_remap(blockIdx.x, gridDim.x);
  const int wi_ = a.rev ? (int)gridDim.x - 1 - wi0_ : wi0_;
  const int qt_ = wi_ % a.nqt, hb_ = wi_ / a.nqt;
  const int head = hb_ % a.H, b = hb_ / a.H;
  const int S = a.S;                                   // 64
  const int g2 = a.grid * a.grid;
  const bf16_t* kvbase = a.q + (long)b * g2 * a.d3 + head * 64;

  int tq = qt_ * (NW * 32) + wave * 32 + r;
  const bool qvalid = tq < a.Tq;
  tq = min(tq, a.Tq - 1);
  const int qh = tq / S, qw = tq - qh * S;
  const long orow = (long)b * g2 + tq;
  const bf16_t* qp = a.q + orow * a.d3 + head * 64;
  uint4 qf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qf[c] = *(const uint4*)(qp + 16 * c + 8 * h);

  // ---- relative-position tables (log2 domain): row part -> aux[kh][q] (LDS), column part -> 32 registers.
  // Round 5 (the fixed part of a block was 16 % of the kernel: four sequential rounds of table loads -> MFMA -> LDS -> 32 conditional
  // copies): the 32 queries of a wave share ONE grid row qh (32 | 64), so
  //   * row part: the A operand's rows are the table rows in the order the tiles need them, row i <- table row qh + 63 - i (i = kh):
  //     the accumulator IS Th[kh][q] in score layout and goes to aux directly (8 MFMAs, 32 LDS writes, no scratch);
  //   * column part: a lane needs table rows j = qw + 63 - kw, qw = 32 pw + r: local rows r + 63 - kw in [0, 94] of the 96-row window
  //     that starts at 32 pw -> 12 MFMAs, 48 LDS writes, 32 conflict-free LDS reads along the lane's diagonal (no conditions);
  //   * every table fragment load is issued before the first MFMA (straight-line code): one load latency instead of four.
  float* aux = (float*)(smem + K_BYTES + V_BYTES + wave * AUX_PER_WAVE);
  f32x16 wreg[2];
  uint4 fh[2][4];
  {
    float* scr = (float*)(smem + wave * (3 * 4096));    // 96 rows x 32 queries x 4 B: aliases the K/V rings (barrier before staging)
    const int pw = (qt_ * NW + wave) & 1;               // which half of the grid row the wave's queries are (qw = 32 pw + r)
    uint4 fw[3][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c = 0; c < 4; ++c) fh[kb][c] = table_frag(a.rel_h, qh + (S - 1) - (kb * 32 + r), c, h);
#pragma unroll
    for (int jb = 0; jb < 3; ++jb)
#pragma unroll
      for (int c = 0; c < 4; ++c) fw[jb][c] = table_frag(a.rel_w, min(32 * pw + jb * 32 + r, 2 * S - 2), c, h);
#pragma unroll
    for (int jb = 0; jb < 3; ++jb) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[jb][c]), __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) scr[(jb * 32 + acc_row(e, h)) * 32 + r] = acc[e] * a.tbl_scale;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // lanes exchange data through LDS (compiler-only fence)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) wreg[kb][e] = scr[(r + (S - 1) - (kb * 32 + acc_row(e, h))) * 32 + r];
  }
  __syncthreads();

  // ---- staging: LDS-DMA (global_load_lds_dwordx4), no registers. A tile is 64 rows x 128 B = 512 chunks of 16 B; thread tid
  // copies chunks tid and tid+256 (rows tid>>3 and 32 + (tid>>3)); the LDS image is lane-linear, so the K swizzle (chunk c of
  // row r at slot c ^ ((r>>1)&7), for conflict-free ds_read_b128) is applied to the SOURCE chunk; V stays linear for
  // ds_read_b64_tr_b16. Rings: K(t) and V(t) live in slot t % 3. Iteration t reads K(t+1), V(t-1) and issues K(t+3), V(t+1).
  char* Kring = smem; char* Vring = smem + K_BYTES;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  const unsigned wofs = __builtin_amdgcn_readfirstlane(wave) * 1024u;
  const int nt = a.Tk / KT;
  const int srow = tid >> 3, sch = tid & 7;
  const int kswz = (sch ^ ((srow >> 1) & 7)) * 8;          // source chunk (elements) of the K copy; (32 + srow)>>1 & 7 is the same
  // V: ds_read_b64_tr_b16 reads a 64-byte column window of key rows rho, rho+1, rho+2, rho+3 at once; rows are 128 B, so rows rho
  // and rho+2 share their banks (2-way conflict on every V read: SQ_LDS_BANK_CONFLICT was 30 % of the kernel's LDS cycles). Rows
  // with bit 1 set store their two 64-byte halves exchanged (source chunk ^ 4), the reader picks window db ^ bit 1 of its row.
  const int vswz = (sch ^ (((srow >> 1) & 1) << 2)) * 8;
  // "saddr" copies: wave-uniform 64-bit tile base (SGPRs, one scalar add per tile) + per-lane 32-bit byte offsets that never change
  // (row srow / 32 + srow of a tile, K or V plane, swizzled chunk): no vector address arithmetic per copy
  const unsigned vk0 = (unsigned)((srow * a.d3 + a.H * 64 + kswz) * 2), vk1 = vk0 + (unsigned)(32 * a.d3 * 2);
  const unsigned vv0 = (unsigned)((srow * a.d3 + 2 * a.H * 64 + vswz) * 2), vv1 = vv0 + (unsigned)(32 * a.d3 * 2);
  const long tile_bytes = (long)KT * a.d3 * 2;
  // (both pieces of a tile under one M0 setting: glds16_so_pair4k, 5 instructions instead of 10 per tile and wave)
  const unsigned vk0p = vk0 + 2048u, vk1m = vk1 - 2048u, vv0p = vv0 + 2048u, vv1m = vv1 - 2048u;
  auto issue_k = [&](int t, int slot) {
    const char* base = (const char*)kvbase + (long)min(t, nt - 1) * tile_bytes;
    glds16_so_pair4k(base, vk0p, vk1m, lds0 + slot * TILE_B + wofs + 2048u);
  };
  auto issue_v = [&](int t, int slot) {
    const char* base = (const char*)kvbase + (long)min(t, nt - 1) * tile_bytes;
    glds16_so_pair4k(base, vv0p, vv1m, lds0 + K_BYTES + slot * TILE_B + wofs + 2048u);
  };
  // the loop's copies: running source pointers (K(t+3), V(t+1); they stop at the last tile, which is then re-read and dropped) instead of
  // a 64-bit multiply per copy (the loop carried 49 scalar instructions per tile for addresses)
  const char* ksrc = (const char*)kvbase + (long)min(3, nt - 1) * tile_bytes;
  const char* vsrc = (const char*)kvbase + (long)min(1, nt - 1) * tile_bytes;
  const unsigned ldsK = lds0 + wofs + 2048u, ldsV = lds0 + K_BYTES + wofs + 2048u;
  // (the converted row-table fragments are pinned BEFORE the copies: hipcc waits for its own loads with vmcnt(0) when it cannot
  // count the asm copies behind them, which would drain the copies too)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(fh[kb][c].x), "+v"(fh[kb][c].y), "+v"(fh[kb][c].z), "+v"(fh[kb][c].w));
  issue_k(0, 0); issue_k(1, 1); issue_k(2, 2); issue_v(0, 0);
  *(uint4*)(Vring + 2 * TILE_B + tid * 16) = make_uint4(0, 0, 0, 0);          // V tile "-1" (slot 2)
  *(uint4*)(Vring + 2 * TILE_B + 4096 + tid * 16) = make_uint4(0, 0, 0, 0);
  // the row part of the bias while the first tiles are in flight (it needs no scratch: accumulator layout = aux layout)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[kb][c]), __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) aux[(kb * 32 + acc_row(e, h)) * 32 + r] = acc[e] * a.tbl_scale;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- fragment read maps
  const int sw = (lane >> 1) & 7;
  int kch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kch[c] = r * 128 + (((2 * c + h) ^ sw) << 4);
  const int v_tr = (4 * h + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int v_trd[2] = {v_tr + ((lane >> 3) & 1) * 64, v_tr + (1 - ((lane >> 3) & 1)) * 64};   // window db of this lane's rows (halves exchanged where row bit 1 is set)

  f32x16 o[2], s[2], lsum;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) lsum[e] = 0.f;
  float m = -INFINITY;
  uint4 pf[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) { pf[kb][0] = make_uint4(0, 0, 0, 0); pf[kb][1] = make_uint4(0, 0, 0, 0); }

  // S(0)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[kb][e] = CB ? wreg[kb][e] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint4 kf = *(const uint4*)(Kring + kb * 32 * 128 + kch[c]);
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[c]), s[kb], 0, 0, 0);
    }
  }
  __syncthreads();                                      // every wave is done with K tile 0 before iteration 0 refills its slot

  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};      // STAMP: cycles in [DMA issue | phase A | reference update | phase B | vmcnt wait | barrier]
  // one key tile; S(t) arrives in `s`, S(t+1) leaves in `sn`
  // C3 = t % 3 and TAIL (the source pointers may have to stop at the last tile) are compile-time: six tiles per loop trip make every ring
  // offset an immediate (the loop carried ~10 scalar instructions per tile for t % 3 and 6 vector adds for the fragment addresses)
  auto tile_step = [&](auto C3_, auto TAIL_, const int t, f32x16 (&s)[2], f32x16 (&sn)[2]) __attribute__((always_inline)) {
    constexpr int c3 = decltype(C3_)::value;
    constexpr bool TAIL = decltype(TAIL_)::value;
    constexpr int c3p1 = c3 == 2 ? 0 : c3 + 1, c3p2 = c3 == 0 ? 2 : c3 - 1;  // (t+1) % 3, (t+2) % 3 = (t-1) % 3
    unsigned long long ts[8];
    auto stamp = [&](int i) { if (STAMP) { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } };
    stamp(0);
    // (the slots are free from the top of the iteration: K(t) and V(t-2) were last read in iteration t-1; issuing the four copies
    // after phase A, after phase B or split between them instead measured +-0.5 %)
    glds16_so_pair4k(ksrc, vk0p, vk1m, ldsK + c3 * TILE_B);       // K(t+3) over K(t), last read in iteration t-1
    glds16_so_pair4k(vsrc, vv0p, vv1m, ldsV + c3p1 * TILE_B);    // V(t+1) over V(t-2), last read in iteration t-1
    if (!TAIL || t + 4 < nt) ksrc += tile_bytes;
    if (!TAIL || t + 2 < nt) vsrc += tile_bytes;
    stamp(1);
    const char* Vs = Vring + c3p2 * TILE_B;             // V(t-1)
    const char* Ks = Kring + c3p1 * TILE_B;             // K(t+1)
    const float rh = aux[t * 32 + r];
    // ---- phase A: PV(t-1) and row-sum MFMAs beside (scale + bias +) max of S(t)
    uint4 vf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kb = i >> 2, ks = (i >> 1) & 1, db = i & 1;
      const char* vb = Vs + (kb * 32 + ks * 16) * 128 + v_trd[db];
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
      const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
      vf[i] = make_uint4(u0.x, u0.y, u1.x, u1.y);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (round 5, measured and not kept: with the bias fma gone phase A has ~25 vector instructions for 8 MFMAs and phase B ~100 for 8;
    // moving the PV MFMAs of the second key block to phase B - 4 + 12, O^T rescaled at the end of the iteration - took 1.941 ms against
    // 1.925: between the two waves of a SIMD such moves are zero-sum. NPA = 4 selects that form.)
    constexpr int NPA = 8;
    auto pv = [&](int i) __attribute__((always_inline)) {
      const int kb = i >> 2, ks = (i >> 1) & 1, db = i & 1;
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[i]), __builtin_bit_cast(bf16x8, pf[kb][ks]), o[db], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NPA; ++i) pv(i);
    float mloc;
    {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) if (!CB) s[kb][e] = fmaf(s[kb][e], a.scale_log2, wreg[kb][e]);
      // tile maximum of the lane's 32 scores as a tree of v_max3_f32: 16 instructions (a running max3 chain came out as 17 v_max + 8 v_max3)
      float m4[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int e = 8 * g;
          const float t0 = max3f_(s[kb][e], s[kb][e + 1], s[kb][e + 2]), t1 = max3f_(s[kb][e + 3], s[kb][e + 4], s[kb][e + 5]);
          m4[kb * 2 + g] = max3f_(max3f_(t0, t1, s[kb][e + 6]), s[kb][e + 7], -INFINITY);
        }
      mloc = fmaxf(max3f_(m4[0], m4[1], m4[2]), m4[3]);
#pragma unroll
      for (int i = 0; i < (NPA == 4 ? 4 : 12); ++i) {   // 1 MFMA : 4 VALU (CB: 2)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NPA == 4 ? 6 : (CB ? 2 : 4), 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);
    float msub = 0.f, alpha = 1.f;
    bool resc;
    {
      mloc = fmaxf(mloc, other_half(mloc)) + rh;        // true tile max (x + rh)
      resc = __builtin_amdgcn_ballot_w64(mloc > m + 8.0f) != 0;     // lazy rescale (see the header)
      if (resc) {
        const float mnew = fmaxf(m, mloc);
        alpha = __builtin_amdgcn_exp2f(m - mnew);
        m = mnew;
        lsum[0] *= alpha; lsum[1] *= alpha; lsum[2] *= alpha;
        if (NPA == 8) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
        }
      }
      msub = m - rh;                                    // p = 2^(x + rh - m)
    }
    stamp(3);
    // ---- phase B: QK(t+1) MFMAs beside exp2 and packing of P(t)
    uint4 kf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = *(const uint4*)(Ks + (i >> 2) * 32 * 128 + kch[i & 3]);
    if (!CB) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sn[kb][e] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = NPA; i < 8; ++i) pv(i);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (CB && c == 0) {
          // the chain's first MFMA takes the column bias as its C operand and writes a DIFFERENT register block: hipcc's builtin ties C to D
          // and copied the 32 start values per tile (34 v_mov in the phase); in asm D and C are separate operands. The next MFMA of the
          // chain reads D as its C (same block, same opcode: issues back to back, as the compiler's own chains do).
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(sn[kb]) : "v"(__builtin_bit_cast(u32x4, kf[kb * 4])), "v"(__builtin_bit_cast(u32x4, qf[0])), "v"(wreg[kb]));
        } else
          sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[kb * 4 + c]), __builtin_bit_cast(bf16x8, qf[c]), sn[kb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 p;
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(s[kb][e] - msub);
      pf[kb][0] = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
      pf[kb][1] = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
      // row sums: this lane's 16 keys of the block (the other lane half holds the other 16), two keys per instruction
      lsum[1] = sum2_bf16(pf[kb][0].y, sum2_bf16(pf[kb][0].x, lsum[1])); lsum[2] = sum2_bf16(pf[kb][0].w, sum2_bf16(pf[kb][0].z, lsum[2]));
      lsum[1] = sum2_bf16(pf[kb][1].y, sum2_bf16(pf[kb][1].x, lsum[1])); lsum[2] = sum2_bf16(pf[kb][1].w, sum2_bf16(pf[kb][1].z, lsum[2]));
    }
#pragma unroll
    for (int i = 0; i < (NPA == 4 ? 10 : (CB ? 6 : 8)); ++i) {   // 1 MFMA : 12 VALU (32 sub + 32 exp2 + 16 cvt_pk + 16 dot2); CB: two of the eight score MFMAs are asm
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NPA == 4 ? 10 : (CB ? 16 : 12), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (NPA < 8) {
      asm volatile("" :: "v"(lsum[1]), "v"(lsum[2]));   // (keeps the pure softmax arithmetic above the branch: LLVM sinks it otherwise)
      if (resc) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
      }
    }
    stamp(4);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // K(t+2), V(t) (issued one iteration ago) have landed; this iteration's 4 stay in flight
    stamp(5);
    __syncthreads();
    stamp(6);
    if (STAMP) {
#pragma unroll
      for (int i = 0; i < 6; ++i) tsum[i] += ts[i + 1] - ts[i];
    }
  };
  f32x16 s2[2];
  {
    // six tiles per trip: t % 3 (ring slots) and t % 2 (which of the two score register sets holds S(t): the copy S <- S(t+1) at the
    // end of every tile was 32 v_mov per tile and wave) are compile-time. The trips cover the tiles whose copies need no clamping (every
    // tile of a trip has K(t+4) to advance to); the tail runs the same sequence one tile at a time with the clamps.
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    int t = 0;
#pragma unroll 1
    for (; t + 9 < nt; t += 6) {                        // tiles t .. t+5 with t + 5 + 4 < nt: no clamp
      tile_step(I0{}, std::false_type{}, t, s, s2);     tile_step(I1{}, std::false_type{}, t + 1, s2, s);
      tile_step(I2{}, std::false_type{}, t + 2, s, s2); tile_step(I0{}, std::false_type{}, t + 3, s2, s);
      tile_step(I1{}, std::false_type{}, t + 4, s, s2); tile_step(I2{}, std::false_type{}, t + 5, s2, s);
    }
    // tail (up to 11 tiles, one at a time, t % 6 == 0 at its start): the same sequence with the clamps
#pragma unroll 1
    for (; t < nt; t += 6) {
      if (t < nt) tile_step(I0{}, std::true_type{}, t, s, s2);
      if (t + 1 < nt) tile_step(I1{}, std::true_type{}, t + 1, s2, s);
      if (t + 2 < nt) tile_step(I2{}, std::true_type{}, t + 2, s, s2);
      if (t + 3 < nt) tile_step(I0{}, std::true_type{}, t + 3, s2, s);
      if (t + 4 < nt) tile_step(I1{}, std::true_type{}, t + 4, s, s2);
      if (t + 5 < nt) tile_step(I2{}, std::true_type{}, t + 5, s2, s);
    }
  }
  // PV(nt-1) and its row sums
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Vs = Vring + ((nt - 1) % 3) * TILE_B;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int krow = kb * 32 + ks * 16;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vb = Vs + krow * 128 + v_trd[db];
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
          const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
          const uint4 vfr = make_uint4(u0.x, u0.y, u1.x, u1.y);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfr), __builtin_bit_cast(bf16x8, pf[kb][ks]), o[db], 0, 0, 0);
        }
      }
  }
  { const float l = lsum[1] + lsum[2]; lsum[0] = l + other_half(l); }

  if (STAMP) {                                          // probe build: no outputs, six cycle sums per wave at the start of `o`
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) ((unsigned long long*)a.o)[((long)blockIdx.x * NW + wave) * 8 + i] = tsum[i];
      ((unsigned long long*)a.o)[((long)blockIdx.x * NW + wave) * 8 + 6] = (unsigned long long)(o[0][0] + lsum[0] != 12345.f);   // keep the arithmetic alive
    }
    return;
  }
  const float inv = 1.0f / lsum[0];                     // full row sum (the MFMA already summed both lane halves)
  auto off = [&](int j) -> long {
    const int tj = qt_ * (NW * 32) + wave * 32 + j;
    return tj < a.Tq ? ((long)b * g2 + tj) * (long)(a.H * 64) + head * 64 : -1L;
  };
  store_o_rows<TO, 64>(o, inv, (char*)aux, (TO*)a.o, lane, off);   // the wave's row-bias table is dead: 8 KiB of private staging
}

// ---------------------------------------------------------------------------------------------------------------------
// Global SAM attention, 64 QUERIES PER WAVE at ONE wave per SIMD (round 5; the default when q arrives pre-scaled, i.e. the
// engine's path). flash_global_pipe above runs two 32-query waves per SIMD; its two waves' matrix and vector work barely overlap
// (stamps: a 64-key tile costs about the SUM of both waves' demands) and every K / V fragment is read from LDS once per 32 queries.
// Here a wave owns one whole grid row of queries (64 = two 32-query blocks qb) and the whole 512-register file:
//   * K(t+1) / V(t-1) fragments are read once per 64 queries (half the LDS reads, half the LDS-DMA issues per score);
//   * q is pre-scaled (scale * log2 e folded into the qkv weight at pack time), so the column bias is the C OPERAND of the first
//     score MFMA of each chain (accumulator start value = wreg, kept as f32x16): no vector instruction per score for scale + bias;
//   * one instruction stream per SIMD: the 32 MFMAs of an iteration (PV(t-1): 16, QK(t+1): 16) are spread over the ~230 vector
//     instructions of the softmax of S(t) in proportion (sched_group_barrier), so neither pipe waits for a partner wave's phase;
//   * V ring of four (issued two iterations ahead, like K): one wave per SIMD has nothing else to hide a late tile.
// Row bias: one LDS scalar per lane, q-block and tile, as above. Lazy rescale as above (one ballot for both q-blocks).
// LDS = 32 KiB K ring + 32 KiB V ring + 4 x 16 KiB row-bias tables = 128 KiB: one block per CU.
// DBG (probe builds of the timing ablations only; results are garbage when set): 1 no copies in the loop, 2 no wait / barrier,
// 4 no fragment reads, 8 no exp2, 16 no tile maxima, 32 no PV MFMAs, 64 no score MFMAs
template <typename TO, int DBG = 0>
__global__ void __launch_bounds__(256, 1) flash_global_w64(const FlashArgs a) {   // 4 waves x 64 queries per block
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K_BYTES = 4 * TILE_B, V_BYTES = 4 * TILE_B, AUXW = 64 * 64 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi0_ = xcd_remap(blockIdx.x, gridDim.x);
  const int wi_ = a.rev ? (int)gridDim.x - 1 - wi0_ : wi0_;
  const int qt_ = wi_ % a.nqt, hb_ = wi_ / a.nqt;
  const int head = hb_ % a.H, b = hb_ / a.H;
  constexpr int S = 64;
  const int g2 = S * S;
  const bf16_t* kvbase = a.q + (long)b * g2 * a.d3 + head * 64;
  const int qh = qt_ * NW + wave;                       // the wave's grid row; query (qb, r) sits at column qw = 32 qb + r

  u32x4 qf[2][4];                                       // B operands of every score MFMA: 128-bit values (they live in AGPRs)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const bf16_t* qp = a.q + ((long)b * g2 + qh * S + qb * 32 + r) * a.d3 + head * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[qb][c] = *(const u32x4*)(qp + 16 * c + 8 * h);
  }

  // ---- relative-position tables (log2 domain): row part -> aux[kh][64 queries] (LDS), column part -> 2 x 32 registers.
  // As in flash_global_pipe (round 5): the row part's A operand holds table rows qh + 63 - i, so the accumulator is Th[kh][q] in score
  // layout (straight to aux); the column part of q-block qb needs the 96-row window that starts at table row 32 qb; the row-table
  // fragments serve both q-blocks, the column table's four 32-row blocks too (blocks qb .. qb + 2); all loads before the first MFMA.
  float* aux = (float*)(smem + K_BYTES + V_BYTES + wave * AUXW);
  f32x16 wreg[2][2];
  {
    float* scr = (float*)(smem + wave * (3 * 4096));    // 96 rows x 32 queries x 4 B: aliases the rings (barrier before staging)
    uint4 fh[2][4], fw[4][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c = 0; c < 4; ++c) fh[kb][c] = table_frag(a.rel_h, qh + (S - 1) - (kb * 32 + r), c, h);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int c = 0; c < 4; ++c) fw[jb][c] = table_frag(a.rel_w, min(jb * 32 + r, 2 * S - 2), c, h);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[kb][c]), __builtin_bit_cast(bf16x8, qf[qb][c]), acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) aux[(kb * 32 + acc_row(e, h)) * 64 + qb * 32 + r] = acc[e] * a.tbl_scale;
      }
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[qb + jb][c]), __builtin_bit_cast(bf16x8, qf[qb][c]), acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) scr[(jb * 32 + acc_row(e, h)) * 32 + r] = acc[e] * a.tbl_scale;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // lanes exchange data through LDS (compiler-only fence)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) wreg[qb][kb][e] = scr[(r + (S - 1) - (kb * 32 + acc_row(e, h))) * 32 + r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // the second q-block reuses the scratch
    }
  }
  __syncthreads();

  // ---- staging by LDS-DMA as in flash_global_pipe (same LDS images, same source swizzles); rings of FOUR: K(t) and V(t) live in
  // slot t & 3. Iteration t reads K(t+1), V(t-1) and issues K(t+3), V(t+2).
  char* Kring = smem; char* Vring = smem + K_BYTES;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  const unsigned wofs = __builtin_amdgcn_readfirstlane(wave) * 1024u;
  const int nt = a.Tk / KT;                             // 64 (the launcher requires a 64 x 64 grid)
  const int srow = tid >> 3, sch = tid & 7;
  const int kswz = (sch ^ ((srow >> 1) & 7)) * 8;
  const int vswz = (sch ^ (((srow >> 1) & 1) << 2)) * 8;
  const unsigned vk0 = (unsigned)((srow * a.d3 + a.H * 64 + kswz) * 2), vk1 = vk0 + (unsigned)(32 * a.d3 * 2);
  const unsigned vv0 = (unsigned)((srow * a.d3 + 2 * a.H * 64 + vswz) * 2), vv1 = vv0 + (unsigned)(32 * a.d3 * 2);
  const long tile_bytes = (long)KT * a.d3 * 2;
  auto issue_k = [&](int t, int slot) {
    const char* base = (const char*)kvbase + (long)t * tile_bytes;
    const unsigned d = lds0 + slot * TILE_B + wofs;
    glds16_so(base, vk0, d);
    glds16_so(base, vk1, d + 4096);
  };
  auto issue_v = [&](int t, int slot) {
    const char* base = (const char*)kvbase + (long)t * tile_bytes;
    const unsigned d = lds0 + K_BYTES + slot * TILE_B + wofs;
    glds16_so(base, vv0, d);
    glds16_so(base, vv1, d + 4096);
  };
  issue_k(0, 0); issue_k(1, 1); issue_k(2, 2); issue_v(0, 0); issue_v(1, 1);
  *(uint4*)(Vring + 3 * TILE_B + tid * 16) = make_uint4(0, 0, 0, 0);          // V tile "-1" (slot 3)
  *(uint4*)(Vring + 3 * TILE_B + 4096 + tid * 16) = make_uint4(0, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- fragment read maps (flash_global_pipe's)
  const int sw = (lane >> 1) & 7;
  int kch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kch[c] = r * 128 + (((2 * c + h) ^ sw) << 4);
  const int v_tr = (4 * h + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int v_trd[2] = {v_tr + ((lane >> 3) & 1) * 64, v_tr + (1 - ((lane >> 3) & 1)) * 64};

  f32x16 o[2][2], s[2][2];
  float lsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float m[2] = {-INFINITY, -INFINITY};
  u32x4 pf[2][2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][x][e] = 0.f;
      pf[qb][x][0] = u32x4{0, 0, 0, 0}; pf[qb][x][1] = u32x4{0, 0, 0, 0};
    }

  // S(0) (accumulator start value = column bias)
  {
    uint4 kf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = *(const uint4*)(Kring + (i >> 2) * 32 * 128 + kch[i & 3]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s[qb][kb] = wreg[qb][kb];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[kb * 4 + c]), __builtin_bit_cast(bf16x8, qf[qb][c]), s[qb][kb], 0, 0, 0);
      }
  }
  __syncthreads();                                      // (rings of four: not needed for the K slot, kept as the loop's entry rendezvous)

  // ---- register files. hipcc selects the AGPR form for EVERY builtin MFMA of a 512-register kernel, and the VALU cannot read AGPRs:
  // scores left in AGPRs cost 64 v_accvgpr_read per tile, and mixed use makes the allocator shuttle O^T between the files around
  // every branch. So every MFMA of the loop is inline asm with the files spelled out: scores S, their start values (the C operand:
  // C and D share one file bit) and P in arch VGPRs; O^T, the Q fragments and the K / V fragments (LDS reads straight into AGPRs)
  // in accumulator registers. What the compiler then no longer does for these instructions (MI355X guide 5.7): hazard padding.
  // S is read by the VALU only in the NEXT iteration (hundreds of cycles after the last score MFMA); O^T is read by the rescale
  // (placed behind the whole softmax of the tile) and by the epilogue (behind explicit s_nops).
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+a"(qf[qb][c]));
#pragma unroll
    for (int x = 0; x < 2; ++x) { asm volatile("" : "+v"(s[qb][x])); asm volatile("" : "+v"(wreg[qb][x])); asm volatile("" : "+a"(o[qb][x])); }
  }
  const char* ksrc = (const char*)kvbase + 3 * tile_bytes;   // K(t+3) of iteration t (clamped to the last tile: re-read, dropped)
  const char* vsrc = (const char*)kvbase + 2 * tile_bytes;   // V(t+2)
  // both copies of a tile in one M0 setting: M0 = destination + 2048, instruction offsets -2048 / +2048 (the offset moves the
  // LDS address AND the global address, so the per-lane source offsets carry the opposite 2048)
  const unsigned vk0p = vk0 + 2048u, vk1p = vk1 - 2048u, vv0p = vv0 + 2048u, vv1p = vv1 - 2048u;

#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    // the two copies of a tile under one M0 setting; where in the iteration they are issued is free (their slots were last read one / two
    // iterations ago): DBG bit 512 = both tiles at the top (the first form), else K beside softmax step 2 and V beside step 10 - measured
    // equal (2.27 / 2.28 ms); without the copies the kernel takes 1.84 ms: an LDS-DMA instruction costs its wave ~130 cycles wherever it
    // stands, and at one wave per SIMD nothing else issues meanwhile (register staging - 4 loads + 4 ds_write_b128 per tile - took 2.40)
    auto dma_tile = [&](const unsigned dst, const unsigned v0, const unsigned v1, const char* src) __attribute__((always_inline)) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %4 offset:-2048\n\tglobal_load_lds_dwordx4 %3, %4 offset:2048\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst), "v"(v0), "v"(v1), "s"(src) : "memory");
    };
    const unsigned kd = lds0 + (unsigned)((t + 3) & 3) * TILE_B + wofs + 2048u;                // over K(t-1), last read in iteration t-2
    const unsigned vd = lds0 + K_BYTES + (unsigned)((t + 2) & 3) * TILE_B + wofs + 2048u;      // over V(t-2), last read in iteration t-1
    const char *ksrc_t = ksrc, *vsrc_t = vsrc;
    if (t + 4 < nt) ksrc += tile_bytes;
    if (t + 3 < nt) vsrc += tile_bytes;
    if (!(DBG & 1) && (DBG & 512)) { dma_tile(kd, vk0p, vk1p, ksrc_t); dma_tile(vd, vv0p, vv1p, vsrc_t); }
    const char *Vs = Vring + ((t + 3) & 3) * TILE_B, *Ks = Kring + ((t + 1) & 3) * TILE_B;   // V(t-1), K(t+1)
    const float rh0 = aux[t * 64 + r], rh1 = aux[t * 64 + 32 + r];
    u32x4 vf[8], kf[8];
    if (DBG & 4) { Vs = Vring; Ks = Kring; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kb = i >> 2, ks = (i >> 1) & 1, db = i & 1;
      if ((DBG & 4) && t > 0) { vf[i] = u32x4{(unsigned)t, (unsigned)i, 3u, 4u}; continue; }
      const char* vb = Vs + (kb * 32 + ks * 16) * 128 + v_trd[db];
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
      vf[i] = __builtin_bit_cast(u32x4, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((DBG & 4) && t > 0) { kf[i] = u32x4{(unsigned)t, (unsigned)i, 1u, 2u}; continue; }
      kf[i] = *(const u32x4*)(Ks + (i >> 2) * 32 * 128 + kch[i & 3]);
    }
    __builtin_amdgcn_sched_barrier(0);

    // PV(t-1), (qb, kb)-major: P(t-1) of a block is dead once its four MFMAs are issued, before the block's new P is packed
    auto pv = [&](int n) __attribute__((always_inline)) {
      const int qb = n >> 3, kb = (n >> 2) & 1, ks = (n >> 1) & 1, db = n & 1;
      if (DBG & 32) { asm volatile("" : "+a"(o[qb][db]) : "a"(vf[kb * 4 + ks * 2 + db]), "v"(pf[qb][kb][ks])); return; }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "a"(vf[kb * 4 + ks * 2 + db]), "v"(pf[qb][kb][ks]));
    };
    // S(t+1) block g = (qb, kb), k-step c, in place of S(t)'s consumed block
    auto qk = [&](int g, int c) __attribute__((always_inline)) {
      const int qb = g >> 1, kb = g & 1;
      if (DBG & 64) { asm volatile("" : "+v"(s[qb][kb]) : "a"(kf[kb * 4 + c]), "a"(qf[qb][c]), "v"(wreg[qb][kb])); return; }
      if (c == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[qb][kb]) : "a"(kf[kb * 4 + c]), "a"(qf[qb][c]), "v"(wreg[qb][kb]));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[qb][kb]) : "a"(kf[kb * 4 + c]), "a"(qf[qb][c]));
    };
    // ---- part 1: tile maxima of S(t) (the fragment reads above land meanwhile), then the first two PV(t-1) MFMAs beside the rest
    float mloc[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float m4[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int e = 8 * g;
          const f32x16& x = s[qb][kb];
          if (DBG & 16) { m4[kb * 2 + g] = x[e]; continue; }
          const float t0 = max3f_(x[e], x[e + 1], x[e + 2]), t1 = max3f_(x[e + 3], x[e + 4], x[e + 5]);
          m4[kb * 2 + g] = max3f_(max3f_(t0, t1, x[e + 6]), x[e + 7], -INFINITY);
        }
      mloc[qb] = fmaxf(max3f_(m4[0], m4[1], m4[2]), m4[3]);
      __builtin_amdgcn_sched_barrier(0);
      pv(qb);
    }
    float msub[2], al0 = 1.f, al1 = 1.f;
    bool resc;
    {
      const float ml0 = fmaxf(mloc[0], other_half(mloc[0])) + rh0, ml1 = fmaxf(mloc[1], other_half(mloc[1])) + rh1;   // true tile maxima
      // lazy rescale (see flash_global_pipe). Fourteen of the sixteen PV(t-1) MFMAs, whose P is relative to the OLD reference, are still
      // to be issued: the row sums (complete up to tile t-1) move to the new reference now, O^T only at the END of the iteration
      resc = __builtin_amdgcn_ballot_w64(ml0 > m[0] + 8.0f || ml1 > m[1] + 8.0f) != 0;
      if (resc) {
        const float mn0 = fmaxf(m[0], ml0), mn1 = fmaxf(m[1], ml1);
        al0 = __builtin_amdgcn_exp2f(m[0] - mn0); al1 = __builtin_amdgcn_exp2f(m[1] - mn1);
        m[0] = mn0; m[1] = mn1;
        lsum[0][0] *= al0; lsum[0][1] *= al0; lsum[1][0] *= al1; lsum[1][1] *= al1;
      }
      msub[0] = m[0] - rh0; msub[1] = m[1] - rh1;       // p = 2^(x + rh - m)
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- part 2: the softmax of S(t) in 16 steps of four scores (4 sub + 4 exp2, and 2 cvt_pk + 2 dot2 for the previous step's four),
    // block g = (qb, kb) in steps 4g .. 4g+3; once a block's scores are consumed its registers take S(t+1) IN PLACE (4 MFMAs, start
    // value = column bias). MFMAs per step: PV 2..9 two per step beside block 0; PV 10..13 + QK block 0 beside block 1; PV 14, 15 +
    // QK block 1 beside block 2; QK block 2 beside block 3; QK block 3 runs under the wait, the barrier and the next iteration's
    // copies and reads. One MFMA per half step: sched_barrier pins the order (sched_group_barrier does not see asm as an MFMA).
    float pe[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c <= 16; ++c) {
      float pn[4];
      uint32_t d0 = 0, d1 = 0;
      const int cp = c - 1, gp = cp >> 2, qbp = gp >> 1, kbp = gp & 1, k4 = cp & 3;
      if (c < 16) {
        const int g = c >> 2, qb = g >> 1, kb = g & 1, e0 = 4 * (c & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) pn[i] = (DBG & 8) ? s[qb][kb][e0 + i] - msub[qb] : (DBG & 256) ? __builtin_amdgcn_exp2f(s[qb][kb][e0 + i]) : __builtin_amdgcn_exp2f(s[qb][kb][e0 + i] - msub[qb]);
      }
      if (c > 0) { d0 = pk2(pe[0], pe[1]); if (!(DBG & 128)) lsum[qbp][k4 & 1] = sum2_bf16(d0, lsum[qbp][k4 & 1]); else if (c == 1) lsum[qbp][k4 & 1] += __uint_as_float(d0); }
      if (!(DBG & 1) && !(DBG & 512)) {
        if (c == 2) dma_tile(kd, vk0p, vk1p, ksrc_t);
        if (c == 10) dma_tile(vd, vv0p, vv1p, vsrc_t);
      }
      // first MFMA of the step
      if (c < 4) pv(2 + 2 * c);
      else if (c < 8) pv(6 + c);
      else if (c < 10) pv(6 + c);
      else if (c < 12) qk(1, c - 8);
      else if (c < 16) qk(2, c - 12);
      __builtin_amdgcn_sched_barrier(0);
      if (c < 16) {
        const int g = c >> 2, qb = g >> 1, kb = g & 1, e0 = 4 * (c & 3);
#pragma unroll
        for (int i = 2; i < 4; ++i) pn[i] = (DBG & 8) ? s[qb][kb][e0 + i] - msub[qb] : (DBG & 256) ? __builtin_amdgcn_exp2f(s[qb][kb][e0 + i]) : __builtin_amdgcn_exp2f(s[qb][kb][e0 + i] - msub[qb]);
      }
      if (c > 0) {
        d1 = pk2(pe[2], pe[3]); if (!(DBG & 128)) lsum[qbp][k4 & 1] = sum2_bf16(d1, lsum[qbp][k4 & 1]);
        u32x4& dst = pf[qbp][kbp][k4 >> 1];
        if (k4 & 1) { dst[2] = d0; dst[3] = d1; } else { dst[0] = d0; dst[1] = d1; }
      }
      // second MFMA of the step
      if (c < 4) pv(3 + 2 * c);
      else if (c < 8) qk(0, c - 4);
      else if (c < 10) qk(1, c - 8);
      else if (c == 16) { qk(3, 0); qk(3, 1); qk(3, 2); qk(3, 3); }
#pragma unroll
      for (int i = 0; i < 4; ++i) pe[i] = pn[i];
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the softmax arithmetic above is pure: without this use LLVM sinks all of it below the branch that follows, away from the MFMAs)
    asm volatile("" :: "v"(lsum[0][0]), "v"(lsum[0][1]), "v"(lsum[1][0]), "v"(lsum[1][1]));
    if (resc) {                                          // every PV(t-1) MFMA was issued steps ago, no PV(t) yet: O^T moves to the new reference
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float x0 = o[0][db][e], x1 = o[1][db][e], t0, t1;
          asm volatile("v_accvgpr_read_b32 %2, %0\n\tv_accvgpr_read_b32 %3, %1\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %5\n\t"
                       "v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3"
                       : "+a"(x0), "+a"(x1), "=&v"(t0), "=&v"(t1) : "v"(al0), "v"(al1));
          o[0][db][e] = x0; o[1][db][e] = x1;
        }
    }
    if (!(DBG & 2)) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K(t+2) (issued one iteration ago) has landed, V(t) long since; the younger six stay in flight
      __syncthreads();
    }
  }
  // PV(nt-1)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Vs = Vring + ((nt - 1) & 3) * TILE_B;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vb = Vs + (kb * 32 + ks * 16) * 128 + v_trd[db];
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
          const u32x4 vfr = __builtin_bit_cast(u32x4, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "a"(vfr), "v"(pf[qb][kb][ks]));
        }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMA's 16 passes + write-back before O^T is read (no compiler padding behind asm)
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l = lsum[qb][0] + lsum[qb][1];
    const float inv = 1.0f / (l + other_half(l));       // full row sum: both lane halves
    auto off = [&](int j) -> long { return ((long)b * g2 + qh * S + qb * 32 + j) * (long)(a.H * 64) + head * 64; };
    store_o_rows<TO, 64>(o[qb], inv, (char*)aux, (TO*)a.o, lane, off);   // the wave's row-bias table is dead: private staging
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Windowed SAM attention (14x14 windows), ONE block per (window, head): 7 waves x 32 queries (224 slots for the 196 queries,
// 12.5 % padding; flash_fwd<2> ran two 128-query blocks per (window, head): 23 % padding and every K/V byte staged twice).
//   * K and V of the whole window (2 x 224 rows x 128 B = 56 KiB) are staged ONCE by LDS-DMA while the waves build their
//     rel-pos tables; one barrier, then no synchronisation until the block ends;
//   * LDS row R = 16 * kh + kw (kw = 14, 15: finite duplicates of kw = 13, masked through the bias), so the 32-key MFMA block
//     `blk` holds window rows 2 blk and 2 blk + 1 and accumulator register e of lane half h is key (kh, kw) =
//     (2 blk + (e >> 3), (e & 3) + 8 ((e >> 2) & 1) + 4 h): the row index is a compile-time property of the register, the
//     column pattern is the same in every block -> 8 column-bias registers per lane for the whole kernel;
//   * the softmax VALU bounds this kernel (196 keys amortise nothing), so the bias and the running reference are folded
//     INTO the score MFMA: the accumulator is initialised with  colbias[kw] + rowbias[kh] - m  (one add per score; -inf
//     for masked slots) and Q is pre-scaled by scale * log2(e), so the MFMA result is already  x - m  in the log2 domain
//     and a score costs  add + max3/2 + exp2 + cvt_pk/2  instead of  fma + add + select + max + sub + exp2 + cvt_pk/2;
//   * lazy rescaling as in the other kernels (reference moves only when a block maximum exceeds it by 2^8), row sums of P on
//     the matrix pipe, P^T accumulators are directly the B operand of O^T += V^T . P^T, V^T by ds_read_b64_tr_b16.
// LDS = 56 KiB K/V + 7 x 3456 B (per-wave rel-pos table, later the output staging) = 81536 B: two blocks (14 waves) per CU.
// Where lanes of a wave exchange data through LDS, wavefront-scope fences tell hipcc so (no instruction is emitted).
constexpr int WIN_ROWS = 224, WIN_KV = WIN_ROWS * 128, WIN_T = 27 * 32 * 4, WIN_LDS = 2 * WIN_KV + 7 * WIN_T;

template <typename TO>
__global__ void __launch_bounds__(448, 4) win_attn(const FlashArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int bid = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int head = bid % a.H, win = bid / a.H;
  const int nw2 = a.nW * a.nW, b = win / nw2, wi = win - b * nw2, wy = wi / a.nW, wx = wi - wy * a.nW;
  const int g2 = a.grid * a.grid;
  char* Kt = smem; char* Vt = smem + WIN_KV;
  float* T = (float*)(smem + 2 * WIN_KV + wave * WIN_T);

  // ---- 1. K / V of the window -> LDS (LDS-DMA: no registers, in flight during steps 2-4). 56 wave-instructions of 8 rows =
  // half a window row each: instruction i covers key row kh = (i % 28) / 2 (wave-uniform) and columns kw = 8 (i & 1) + lane / 8,
  // so the per-lane part of the address (column, chunk, image-border test in x) is computed once for the two column halves and
  // the per-instruction part (row y, K or V plane) is scalar.
  {
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    const int sl = lane & 7;
    long colp[2]; bool okx[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kw = min(8 * hf + (lane >> 3), 13), x = wx * 14 + kw;
      okx[hf] = x < a.grid;
      colp[hf] = ((long)b * g2 + x) * a.d3 + head * 64;          // element offset of (row y = 0, column x) of this head's q slice
    }
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) {
      const int i = wave * 8 + i8;                       // wave-uniform
      const bool isv = i >= 28;
      const int ii = isv ? i - 28 : i, kh = ii >> 1, hf = ii & 1, y = wy * 14 + kh;
      // K: source chunk swizzled for conflict-free ds_read_b128 (LDS row R = 16 kh + 8 hf + lane / 8: (R >> 1) & 7 = 4 hf + lane / 16)
      // V: rows with bit 1 set store their 64-byte halves exchanged (conflict-free ds_read_b64_tr_b16, see flash_global_pipe): bit 1 of R = lane bit 4
      const int chunk = isv ? (sl ^ (((lane >> 4) & 1) << 2)) : (sl ^ (4 * hf + (lane >> 4)));
      const long plane = (long)(isv ? 2 : 1) * a.H * 64 + chunk * 8;
      const bf16_t* src = (y < a.grid && okx[hf]) ? a.q + colp[hf] + (long)y * a.grid * a.d3 + plane : a.pad_row + head * 64 + plane;
      glds16(src, lds0 + (isv ? WIN_KV : 0) + ii * 1024);
    }
  }

  // ---- 2. this lane's query (lanes r and r + 32 hold the same query: k halves of every 16-wide K-step)
  const int tq = min(wave * 32 + r, 195);
  const int qh = (tq * 4682) >> 16, qw = tq - 14 * qh;     // tq / 14 for tq < 224
  uint4 qf[4];
  {
    const int y = wy * 14 + qh, x = wx * 14 + qw;
    const bf16_t* qp = (y < a.grid && x < a.grid) ? a.q + ((long)b * g2 + (long)y * a.grid + x) * a.d3 + head * 64 : a.pad_row + head * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *(const uint4*)(qp + 16 * c + 8 * h);
  }

  // ---- 3. rel-pos tables from the UNSCALED q, log2 domain: T[j][q] = log2e * Rtable[j,:] . q. Column table first (gathered
  // into 8 registers), then the row table, which stays in the wave's LDS region for the main loop.
  float colb[8];
  uint4 tf[2][4];                                       // both tables' fragments are loaded before the first product (one load latency, not two)
#pragma unroll
  for (int tbl = 0; tbl < 2; ++tbl)
#pragma unroll
    for (int c = 0; c < 4; ++c) tf[tbl][c] = table_frag(tbl == 0 ? a.rel_w : a.rel_h, min(r, 26), c, h);
#pragma unroll
  for (int tbl = 0; tbl < 2; ++tbl) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tf[tbl][c]), __builtin_bit_cast(bf16x8, qf[c]), acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = acc_row(e, h);
      if (row < 27) T[row * 32 + r] = acc[e] * a.tbl_scale;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // lanes exchange table entries through LDS (compiler-only fence)
    if (tbl == 0) {
#pragma unroll
      for (int e8 = 0; e8 < 8; ++e8) {
        const int kw = (e8 & 3) + 8 * (e8 >> 2) + 4 * h;
        colb[e8] = kw < 14 ? T[(qw - min(kw, 13) + 13) * 32 + r] : -INFINITY;      // kw = 14, 15: padding slots
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... before the row table overwrites the column table
    }
  }

  // ---- 4. the score MFMA works directly in the log2 domain: q must carry scale * log2(e). The engine folds that factor into
  // the q rows of the qkv weight at pack time (q_prescale: no run-time cost, ONE rounding of the scaled weight); a caller that
  // passes raw q (scale_log2 != 1) gets it re-scaled here (bf16 -> bf16: one more rounding on the q side).
  if (a.scale_log2 != 1.0f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t w[4] = {qf[c].x, qf[c].y, qf[c].z, qf[c].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pk2(__uint_as_float(w[i] << 16) * a.scale_log2, __uint_as_float(w[i] & 0xffff0000u) * a.scale_log2);
      qf[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's K/V pieces have landed
  __syncthreads();                                      // ... and everybody else's

  // ---- 5. main loop: 7 blocks of 32 keys (two window rows each), no barrier
  const int sw = (r >> 1) & 7;
  const int k_rd = r * 128, v_tr = (4 * h + ((lane & 15) >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int v_trd[2] = {v_tr + ((lane >> 3) & 1) * 64, v_tr + (1 - ((lane >> 3) & 1)) * 64};
  const int t_rd = (qh + 13) * 32 + r;                  // row table index of key row kh: t_rd - 32 * kh
  f32x16 o[2], lsum;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; lsum[e] = 0.f; }
  float m = 0.f;
  const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);   // eight bf16 1.0
#pragma unroll
  for (int blk = 0; blk < 7; ++blk) {
    const char* Kb = Kt + blk * 32 * 128 + k_rd;
    uint4 kf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) kf[c] = *(const uint4*)(Kb + (((2 * c + h) ^ sw) << 4));
    const float rm0 = T[t_rd - 32 * (2 * blk)] - m, rm1 = T[t_rd - 32 * (2 * blk + 1)] - m;
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = colb[e & 7] + (e < 8 ? rm0 : rm1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[c]), __builtin_bit_cast(bf16x8, qf[c]), s, 0, 0, 0);
    float mloc = max3f_(max3f_(s[0], s[1], s[2]), max3f_(s[3], s[4], s[5]), max3f_(s[6], s[7], s[8]));   // 8 v_max3 / v_max instead of 15 v_max
    mloc = fmaxf(max3f_(mloc, max3f_(s[9], s[10], s[11]), max3f_(s[12], s[13], s[14])), s[15]);
    mloc = fmaxf(mloc, other_half(mloc));               // block maximum of x - m for this query
    // reference update: always after the first block (m = its maximum), later only when a block maximum exceeds it by 2^8
    if (blk == 0 || __builtin_amdgcn_ballot_w64(mloc > 8.0f) != 0) {
      const float d = blk == 0 ? mloc : fmaxf(mloc, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m += d;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] -= d;
      if (blk != 0) {
        lsum[1] *= alpha; lsum[2] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
      }
    }
    uint4 pf[2];
    {
      f32x16 p;
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(s[e]);
      pf[0] = pack8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
      pf[1] = pack8(p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]);
      // row sums on the VALU, two keys per instruction on the packed pairs (see flash_global_pipe)
      lsum[1] = sum2_bf16(pf[0].y, sum2_bf16(pf[0].x, lsum[1])); lsum[2] = sum2_bf16(pf[0].w, sum2_bf16(pf[0].z, lsum[2]));
      lsum[1] = sum2_bf16(pf[1].y, sum2_bf16(pf[1].x, lsum[1])); lsum[2] = sum2_bf16(pf[1].w, sum2_bf16(pf[1].z, lsum[2]));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const char* vb = Vt + (blk * 32 + ks * 16) * 128 + v_trd[db];
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * 128));
        const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
        const uint4 vfr = make_uint4(u0.x, u0.y, u1.x, u1.y);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfr), __builtin_bit_cast(bf16x8, pf[ks]), o[db], 0, 0, 0);
      }
    }
  }

  // ---- 6. O / l, transposed through the wave's own LDS region (the row table is dead), one 32-column half (bf16) or 16-column
  // quarter (fp32) of the 32 query rows at a time (row stride 80 B, 2560 B), stored as 64-byte row pieces, 16 rows per instruction.
  // EVERY lane writes before any lane reads and wavefront-scope fences separate the two: lanes exchange data through LDS here,
  // which single-thread reasoning does not see (a conditional write followed by a read let hipcc forward a stale register).
  const float lhalf = lsum[1] + lsum[2];
  const float inv = 1.0f / (lhalf + other_half(lhalf));
  char* stg = (char*)T;
  TO* out = (TO*)a.o;
  auto off = [&](int jq) -> long {
    const int tj = wave * 32 + jq;
    if (tj >= 196) return -1;
    const int jh = (tj * 4682) >> 16, y = wy * 14 + jh, x = wx * 14 + (tj - 14 * jh);
    if (y >= a.grid || x >= a.grid) return -1;
    return ((long)b * g2 + (long)y * a.grid + x) * (long)(a.H * 64) + head * 64;
  };
  const long e0 = off(lane >> 2), e1 = off((lane >> 2) + 16);
  const int rd0 = (lane >> 2) * 80 + (lane & 3) * 16;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if constexpr (sizeof(TO) == 2) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 u;
        u.x = pk2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv); u.y = pk2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *(uint2*)(stg + r * 80 + (8 * g + 4 * h) * 2) = u;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint4 v0 = *(const uint4*)(stg + rd0), v1 = *(const uint4*)(stg + rd0 + 16 * 80);
      if (e0 >= 0) *(uint4*)(out + e0 + db * 32 + (lane & 3) * 8) = v0;
      if (e1 >= 0) *(uint4*)(out + e1 + db * 32 + (lane & 3) * 8) = v1;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  } else {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {                    // columns 16 qd .. 16 qd + 15: accumulator half db = qd >> 1, registers 8 (qd & 1) ..
      const int db = qd >> 1, g0 = 2 * (qd & 1);
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = g0 + gg;
        const f32x4 v4 = {o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
        *(f32x4*)(stg + r * 80 + (8 * gg + 4 * h) * 4) = v4;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const f32x4 v0 = *(const f32x4*)(stg + rd0), v1 = *(const f32x4*)(stg + rd0 + 16 * 80);
      if (e0 >= 0) *(f32x4*)(out + e0 + qd * 16 + (lane & 3) * 4) = v0;
      if (e1 >= 0) *(f32x4*)(out + e1 + qd * 16 + (lane & 3) * 4) = v1;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

template <typename TO>
int launch_win(const FlashArgs& a, int nwin, hipStream_t s) {
  static DevOnce once;
  cor_max_dyn_lds((const void*)win_attn<TO>, WIN_LDS, once);
  hipLaunchKernelGGL((win_attn<TO>), dim3(nwin * a.H), dim3(448), WIN_LDS, s, a);
  COR_CHECK_LAUNCH();
  return 0;
}

template <typename TO, bool STAMP = false, bool CB = false>
int launch_global_pipe(const FlashArgs& a, int nb, hipStream_t s) {
  const size_t lds = 6 * TILE_B + 4 * AUX_PER_WAVE;    // 80 KiB: two blocks per CU
  static DevOnce once;
  cor_max_dyn_lds((const void*)flash_global_pipe<TO, STAMP, CB>, (int)lds, once);
  FlashArgs b = a;
  b.nqt = cdiv(a.Tq, 128);
  hipLaunchKernelGGL((flash_global_pipe<TO, STAMP, CB>), dim3(b.nqt * a.H * nb), dim3(256), lds, s, b);
  COR_CHECK_LAUNCH();
  return 0;
}

template <typename TO, int DBG = 0>
int launch_global_w64(const FlashArgs& a, int nb, hipStream_t s) {
  const size_t lds = 8 * TILE_B + 4 * 64 * 64 * 4;     // 128 KiB: one block (4 waves x 64 queries) per CU
  static DevOnce once;
  cor_max_dyn_lds((const void*)flash_global_w64<TO, DBG>, (int)lds, once);
  FlashArgs b = a;
  b.nqt = a.Tq / 256;
  hipLaunchKernelGGL((flash_global_w64<TO, DBG>), dim3(b.nqt * a.H * nb), dim3(256), lds, s, b);
  COR_CHECK_LAUNCH();
  return 0;
}

template <int MODE, typename TO, int HD = 64>
int launch(const FlashArgs& a, int nb, hipStream_t s) {
  constexpr int ROWB = HD == 64 ? 128 : 208;
  const size_t lds = 4 * KT * ROWB + (MODE == 0 ? 0 : 4 * AUX_PER_WAVE);      // K0 V0 K1 V1 (+ per-wave rel-pos tables)
  static DevOnce once;
  cor_max_dyn_lds((const void*)flash_fwd<MODE, TO, HD>, (int)lds, once);
  FlashArgs b = a;
  b.nqt = cdiv(a.Tq, 128);
  hipLaunchKernelGGL((flash_fwd<MODE, TO, HD>), dim3(b.nqt * a.H * nb), dim3(256), lds, s, b);
  COR_CHECK_LAUNCH();
  return 0;
}
template <int MODE, int HD>
int launch_t(const FlashArgs& a, int nb, int out_dtype, hipStream_t s) {
  if (out_dtype == COR_BF16) return launch<MODE, bf16_t, HD>(a, nb, s);
  if (out_dtype == COR_F32) return launch<MODE, float, HD>(a, nb, s);
  return COR_ENOSUPPORT;
}

}  // namespace

// `variant` (per call): 0 (default) = flash_global_pipe (global; software-pipelined over key tiles; with a pre-scaled q the column
// bias is the score MFMA's C operand) / win_attn (windowed; one 7-wave block per (window, head)); 1 = flash_fwd<1> / flash_fwd<2>,
// the round-1 chain forms of the same arithmetic, kept as the in-process A/B and parity partners (tests/test_gpu_parity.py,
// tools/attn_bench.py); 2 = flash_global_pipe with the bias as one fma per score (round 4's default; A/B partner); 4 =
// flash_global_w64 (64 queries per wave, one wave per SIMD: built in round 5, parity-green, slower - DESIGN 3.2). Anything else is COR_EINVAL in the
// production library; a COR_PROBES build (make probes -> tools/probes/libcor_probes.so) adds 9 = the timing probe of the default
// global kernel (per-wave s_memtime sums per loop section INSTEAD of the outputs: tools/attn_stamps.py).
int cor_flash_plain_bf16(const void* q, long q_sb, long q_st, const void* k, long k_sb, long k_st, const void* v, long v_sb, long v_st,
                         void* out, long o_sb, long o_st, int out_dtype, int B, int H, int Tq, int Tk, int hd, float scale, hipStream_t s) {
  // 16-B fragment loads: every row start must be 16-B aligned
  if ((q_st | k_st | v_st | q_sb | k_sb | v_sb) & 7) return COR_ENOSUPPORT;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return COR_ENOSUPPORT;
  if ((o_st | o_sb) & 7) return COR_ENOSUPPORT;         // 16-byte row stores
  FlashArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = out;
  a.q_sb = q_sb; a.q_st = q_st; a.k_sb = k_sb; a.k_st = k_st; a.v_sb = v_sb; a.v_st = v_st; a.o_sb = o_sb; a.o_st = o_st;
  a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale_log2 = scale * LOG2E; a.tbl_scale = LOG2E; a.S = 1; a.grid = 1; a.nW = 1;
  if (hd == 64) return launch_t<0, 64>(a, B, out_dtype, s);
  if (hd == 72) return launch_t<0, 72>(a, B, out_dtype, s);       // SigLIP SO400M/14 (the factory's default tower)
  if (hd == 80) return launch_t<0, 80>(a, B, out_dtype, s);
  return COR_ENOSUPPORT;
}

int cor_flash_sam_bf16(const void* qkv, void* out, int out_dtype, const void* pad_row, const float* rel_h, const float* rel_w, int B,
                       int H, int hd, int grid, int window, float q_prescale, int variant, hipStream_t s) {
  if (hd != 64 && hd != 80) return COR_ENOSUPPORT;     // SAM-B/L: 64, SAM-H: 80
  const int rev = (variant & COR_ORDER_REVERSE) ? 1 : 0;
  variant &= ~COR_ORDER_REVERSE;
#ifdef COR_PROBES
  if (variant != 0 && variant != 1 && variant != 2 && variant != 4 && variant != 9 && !(variant >= 16 && variant < 4096)) return COR_EINVAL;
#else
  if (variant != 0 && variant != 1 && variant != 2 && variant != 4) return COR_EINVAL;   // no probe / experimental kernels in the production library
#endif
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)rel_h & 15) || ((uintptr_t)rel_w & 15)) return COR_ENOSUPPORT;
  FlashArgs a{};
  a.q = (const bf16_t*)qkv; a.o = out; a.H = H; a.rev = rev;
  a.scale_log2 = LOG2E / sqrtf((float)hd) / q_prescale; a.tbl_scale = LOG2E / q_prescale;
  if (hd == 64 && fabsf(a.scale_log2 - 1.0f) < 1e-6f) { a.scale_log2 = 1.0f; a.tbl_scale = 8.0f; }   // q_prescale = 0.125 * log2(e): exact constants
  a.pad_row = (const bf16_t*)pad_row; a.rel_h = rel_h; a.rel_w = rel_w; a.grid = grid; a.d3 = 3 * H * hd;
  if (window == 0) {
    if (grid != 64) return COR_ENOSUPPORT;            // one key row per 64-key tile
    a.S = 64; a.Tq = a.Tk = grid * grid; a.nW = 1;
#ifdef COR_PROBES
    if (const char* e = getenv("COR_ATTN_TK")) a.Tk = atoi(e);   // timing probe: fewer key tiles (garbage results): prologue + epilogue share
#endif
    if (hd == 80) return launch_t<1, 80>(a, B, out_dtype, s);     // SAM-H: the chain form (the pipelined kernel is head_dim 64 only)
#ifdef COR_PROBES
    if (variant >= 16 && out_dtype == COR_BF16 && a.scale_log2 == 1.0f) {   // timing ablations of flash_global_w64 (garbage results)
      switch (variant - 16) {
        case 1: return launch_global_w64<bf16_t, 1>(a, B, s);
        case 2: return launch_global_w64<bf16_t, 2>(a, B, s);
        case 3: return launch_global_w64<bf16_t, 3>(a, B, s);
        case 4: return launch_global_w64<bf16_t, 4>(a, B, s);
        case 8: return launch_global_w64<bf16_t, 8>(a, B, s);
        case 16: return launch_global_w64<bf16_t, 16>(a, B, s);
        case 32: return launch_global_w64<bf16_t, 32>(a, B, s);
        case 64: return launch_global_w64<bf16_t, 64>(a, B, s);
        case 96: return launch_global_w64<bf16_t, 96>(a, B, s);
        case 7: return launch_global_w64<bf16_t, 7>(a, B, s);
        case 127: return launch_global_w64<bf16_t, 127>(a, B, s);
        case 128: return launch_global_w64<bf16_t, 128>(a, B, s);
        case 256: return launch_global_w64<bf16_t, 256>(a, B, s);
        case 384: return launch_global_w64<bf16_t, 384>(a, B, s);
        case 512: return launch_global_w64<bf16_t, 512>(a, B, s);
        default: return COR_EINVAL;
      }
    }
    if (variant == 9) {
      if (out_dtype == COR_BF16) return launch_global_pipe<bf16_t, true>(a, B, s);
      if (out_dtype == COR_F32) return launch_global_pipe<float, true>(a, B, s);
    }
#endif
    if (variant == 4 && a.scale_log2 == 1.0f) {        // experimental: 64 queries per wave at one wave per SIMD (measured slower)
      if (out_dtype == COR_BF16) return launch_global_w64<bf16_t>(a, B, s);
      if (out_dtype == COR_F32) return launch_global_w64<float>(a, B, s);
    }
    if (variant == 0 && a.scale_log2 == 1.0f) {        // pre-scaled q (the engine's path): column bias as the score MFMA's C operand
      if (out_dtype == COR_BF16) return launch_global_pipe<bf16_t, false, true>(a, B, s);
      if (out_dtype == COR_F32) return launch_global_pipe<float, false, true>(a, B, s);
    }
    if (variant == 0 || variant == 2 || variant == 4) {
      if (out_dtype == COR_BF16) return launch_global_pipe<bf16_t>(a, B, s);
      if (out_dtype == COR_F32) return launch_global_pipe<float>(a, B, s);
    }
    if (out_dtype == COR_BF16) return launch<1, bf16_t>(a, B, s);
    if (out_dtype == COR_F32) return launch<1, float>(a, B, s);
    return COR_ENOSUPPORT;
  }
  if (window != 14 || ((uintptr_t)pad_row & 15)) return COR_ENOSUPPORT;
  a.S = 14; a.Tq = a.Tk = 196; a.nW = (grid + 13) / 14;
  if (hd == 80) return launch_t<2, 80>(a, B * a.nW * a.nW, out_dtype, s);
  if (variant != 1) {                                   // one 7-wave block per (window, head)
    if (out_dtype == COR_BF16) return launch_win<bf16_t>(a, B * a.nW * a.nW, s);
    if (out_dtype == COR_F32) return launch_win<float>(a, B * a.nW * a.nW, s);
    return COR_ENOSUPPORT;
  }
  if (out_dtype == COR_BF16) return launch<2, bf16_t>(a, B * a.nW * a.nW, s);
  if (out_dtype == COR_F32) return launch<2, float>(a, B * a.nW * a.nW, s);
  return COR_ENOSUPPORT;
}
