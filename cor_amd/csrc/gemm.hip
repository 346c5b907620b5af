// cor_amd — NT GEMM with fused epilogue for gfx950 (MI355X).
//
//   C[M,N] = residual + col_scale * act(A[M,K] . W[N,K]^T + bias)
//
// Both operands are K-contiguous (activation rows, nn.Linear weight rows), which is exactly the MFMA A/B fragment
// order, so one LDS image serves both dtypes:
//   * block tile 128x128, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs);
//   * one K-step = 128 BYTES of K per row (64 bf16 / 32 fp32): a tile row is 8 chunks of 16 B;
//   * lane (r = lane&31, h = lane>>5) reads chunk 2s+h of row r for s = 0..3 with ds_read_b128:
//       bf16: the chunk is the 8 k-values of v_mfma_f32_32x32x16_bf16's operand  -> 1 MFMA per chunk pair;
//       fp32: the chunk holds 4 k-values, fed one at a time to v_mfma_f32_32x32x2_f32 -> 4 MFMAs per chunk pair
//             (k order inside a step is permuted identically for A and W, which a dot product does not see);
//   * LDS rows are 128 B, so chunk c of row r is stored at chunk slot c ^ ((r>>1)&7): the 16 rows of one
//     ds_read_b128 lane group then hit 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//   * register-staged double buffering (global_load_dwordx4 for tile t+1 issued before the MFMAs of tile t,
//     written to the other LDS buffer after them), one barrier per K-step;
//   * XCD-aware tile order: blocks that share an XCD walk N-tiles of the same A row-panel (L2 reuse of A).
#include "common.h"

namespace {

struct GemmArgs {
  const char* A; const char* W; char* C;
  long lda_b, ldw_b;   // row strides in BYTES
  long ldc;            // elements
  int M, N, Kb;        // Kb = K in BYTES
  int tm, tn;
  const float* bias; const float* col_scale; const float* residual;
  long ldr; int res_row_mod; int act;
};

template <typename TA> struct Mfma;
template <> struct Mfma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

constexpr int BM = 128, BN = 128, ROWB = 128;          // tile rows / bytes of K per row per step
constexpr int TILE_BYTES = BM * ROWB;                  // 16 KiB per operand per buffer
constexpr int GEMM_LDS = 4 * TILE_BYTES;               // A0 B0 A1 B1 = 64 KiB

template <typename TA, typename TO>
__global__ void __launch_bounds__(256, 2) gemm_nt_mfma(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (swz / g.tn) * BM, n0 = (swz % g.tn) * BN;

  // ---- staging map: thread handles chunk (tid + 256 i), i = 0..3, of each operand tile
  const char* a_src[4]; const char* b_src[4]; int lds_st[4]; int kofs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
    const int ar = min(m0 + row, g.M - 1), br = min(n0 + row, g.N - 1);   // clamp: rows past the edge are never stored
    a_src[i] = g.A + (long)ar * g.lda_b + ch * 16;
    b_src[i] = g.W + (long)br * g.ldw_b + ch * 16;
    lds_st[i] = row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
    kofs[i] = ch * 16;
  }
  // ---- fragment read map
  const int r = lane & 31, h = lane >> 5, sw = (lane >> 1) & 7;
  const int a_rd = (wm * 64 + r) * ROWB, b_rd = (wn * 64 + r) * ROWB;
  int ch_rd[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ch_rd[s] = ((2 * s + h) ^ sw) << 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nkt = (g.Kb + ROWB - 1) / ROWB;
  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int kb = kt * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = kb + kofs[i] + 16 <= g.Kb;       // K tail: zero fill (K bytes is a multiple of 16)
      ra[i] = ok ? *(const uint4*)(a_src[i] + kb) : make_uint4(0, 0, 0, 0);
      rb[i] = ok ? *(const uint4*)(b_src[i] + kb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    char* As = smem + buf * 2 * TILE_BYTES; char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) { *(uint4*)(As + lds_st[i]) = ra[i]; *(uint4*)(Bs + lds_st[i]) = rb[i]; }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) gload(kt + 1);
    const char* As = smem + (kt & 1) * 2 * TILE_BYTES; const char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 a0 = *(const uint4*)(As + a_rd + ch_rd[s]);
      const uint4 a1 = *(const uint4*)(As + a_rd + 32 * ROWB + ch_rd[s]);
      const uint4 b0 = *(const uint4*)(Bs + b_rd + ch_rd[s]);
      const uint4 b1 = *(const uint4*)(Bs + b_rd + 32 * ROWB + ch_rd[s]);
      Mfma<TA>::run(a0, b0, acc[0][0]);
      Mfma<TA>::run(a0, b1, acc[0][1]);
      Mfma<TA>::run(a1, b0, acc[1][0]);
      Mfma<TA>::run(a1, b1, acc[1][1]);
    }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  TO* C = (TO*)g.C;
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int n = n0 + wn * 64 + nj * 32 + r;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.0f;
    const float sc = g.col_scale ? g.col_scale[n] : 1.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = apply_act(acc[mi][nj][e] + bv, g.act) * sc;
        if (g.residual) {
          const int rr = g.res_row_mod > 0 ? m % g.res_row_mod : m;
          v += g.residual[(long)rr * g.ldr + n];
        }
        st<TO>(C + (long)m * g.ldc + n, v);
      }
    }
  }
}

// Generic fallback for shapes the MFMA path cannot take (K bytes not a multiple of 16, unaligned rows):
// 32x32 output tile per block of 256 threads, fp32 FMA, every access bounds-checked.
template <typename TA, typename TO>
__global__ void __launch_bounds__(256) gemm_nt_small(const TA* A, long lda, const TA* W, long ldw, TO* C, long ldc,
                                                     int M, int N, int K, const float* bias, int act,
                                                     const float* col_scale, const float* residual, long ldr,
                                                     int res_row_mod) {
  __shared__ float As[32][33], Ws[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  float acc[4] = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rrow = ty + 8 * i, k = k0 + tx;
      As[rrow][tx] = (m0 + rrow < M && k < K) ? ld<TA>(A + (long)(m0 + rrow) * lda + k) : 0.0f;
      Ws[rrow][tx] = (n0 + rrow < N && k < K) ? ld<TA>(W + (long)(n0 + rrow) * ldw + k) : 0.0f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const float w = Ws[tx][kk];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(As[ty + 8 * i][kk], w, acc[i]);
    }
    __syncthreads();
  }
  const int n = n0 + tx;
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.0f, sc = col_scale ? col_scale[n] : 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i;
    if (m >= M) continue;
    float v = apply_act(acc[i] + bv, act) * sc;
    if (residual) v += residual[(long)(res_row_mod > 0 ? m % res_row_mod : m) * ldr + n];
    st<TO>(C + (long)m * ldc + n, v);
  }
}

template <typename TA, typename TO>
int launch_gemm(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                const float* bias, int act, const float* col_scale, const float* residual, long ldr, int res_row_mod,
                hipStream_t s) {
  const long esz = sizeof(TA);
  const bool fast = (K * esz) % 16 == 0 && (lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0 &&
                    ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0);
  if (fast) {
    GemmArgs g;
    g.A = (const char*)A; g.W = (const char*)W; g.C = (char*)C;
    g.lda_b = lda * esz; g.ldw_b = ldw * esz; g.ldc = ldc;
    g.M = M; g.N = N; g.Kb = (int)(K * esz);
    g.tm = cdiv(M, BM); g.tn = cdiv(N, BN);
    g.bias = bias; g.col_scale = col_scale; g.residual = residual; g.ldr = ldr; g.res_row_mod = res_row_mod; g.act = act;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_mfma<TA, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_mfma<TA, TO>), dim3(g.tm * g.tn), dim3(256), GEMM_LDS, s, g);
  } else {
    hipLaunchKernelGGL((gemm_nt_small<TA, TO>), dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, s, (const TA*)A, lda,
                       (const TA*)W, ldw, (TO*)C, ldc, M, N, K, bias, act, col_scale, residual, ldr, res_row_mod);
  }
  COR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int cor_gemm(const void* A, long lda, const void* W, long ldw, int ab_dtype, void* C, long ldc, int c_dtype,
                        int M, int N, int K, const float* bias, int act, const float* col_scale,
                        const float* residual, long ldr, int res_row_mod, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N) return COR_EINVAL;
  if (residual && ldr < N) return COR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (ab_dtype == COR_F32 && c_dtype == COR_F32)
    return launch_gemm<float, float>(A, lda, W, ldw, C, ldc, M, N, K, bias, act, col_scale, residual, ldr, res_row_mod, s);
  if (ab_dtype == COR_BF16 && c_dtype == COR_BF16)
    return launch_gemm<bf16_t, bf16_t>(A, lda, W, ldw, C, ldc, M, N, K, bias, act, col_scale, residual, ldr, res_row_mod, s);
  if (ab_dtype == COR_BF16 && c_dtype == COR_F32)
    return launch_gemm<bf16_t, float>(A, lda, W, ldw, C, ldc, M, N, K, bias, act, col_scale, residual, ldr, res_row_mod, s);
  if (ab_dtype == COR_F32 && c_dtype == COR_BF16)
    return launch_gemm<float, bf16_t>(A, lda, W, ldw, C, ldc, M, N, K, bias, act, col_scale, residual, ldr, res_row_mod, s);
  return COR_ENOSUPPORT;
}
