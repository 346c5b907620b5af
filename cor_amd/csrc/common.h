// cor_amd — device-side helpers shared by all gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "../../include/cor_amd.h"

typedef uint16_t bf16_t;  // storage type for bfloat16

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define COR_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4-wide vector load/store (16 B for fp32, 8 B for bf16); pointers must be suitably aligned.
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  uint2 u = *(const uint2*)p;
  f32x4 r;
  r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
  r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
  return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  *(uint2*)p = u;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off level), ~15 VALU ops instead of libm's ~40:
// the GELU epilogue of the MLP GEMMs runs once per output element.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// erf GELU for values that are ROUNDED TO BF16 right after (GEMM epilogues with a bf16 output): erf(x / sqrt 2) as the odd
// polynomial x * P(x^2) on |x| <= 4 (Lawson minimax fit constrained to reach exactly 1 at the clamp, so x >= 4 gives x and
// x <= -4 gives 0), 13 full-rate instructions and no transcendental instead of 14 + rcp + exp2. |erf error| <= 6.4e-5,
// |GELU error| <= 1.3e-4 (below half a bf16 ulp for |y| > 0.07; fp32 outputs keep the 1.5e-7 form above).
__device__ __forceinline__ float gelu_erf_bf16out_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float t = xc * xc;
  float p = fmaf(-2.557373525739815e-09f, t, 2.0897032832641423e-07f);
  p = fmaf(p, t, -7.421273508272735e-06f);
  p = fmaf(p, t, 0.00015240515430200944f);
  p = fmaf(p, t, -0.0020422501798044567f);
  p = fmaf(p, t, 0.01916329039530596f);
  p = fmaf(p, t, -0.13212890465728208f);
  p = fmaf(p, t, 0.7976113602924678f);
  const float hx = 0.5f * x;
  return fmaf(hx, xc * p, hx);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case COR_ACT_GELU_ERF: return gelu_erf_f(v);
    case COR_ACT_RELU: return fmaxf(v, 0.0f);
    case COR_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case COR_ACT_GELU_TANH: return gelu_tanh_f(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware block remap: blocks with equal (bid % 8) share an XCD (speed only, never correctness);
// give each such group a contiguous chunk of the tile space so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// global -> LDS direct copy of 16 B per lane (LDS-DMA): LDS destination = lds_dst (wave-uniform, in M0) + lane*16.
// M0 is compiler-reserved, so it is saved, set, used and restored inside ONE asm statement. The copy is invisible to
// hipcc's s_waitcnt bookkeeping: the caller drains it (s_waitcnt vmcnt) before the barrier that publishes the data.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Same copy, "saddr" form: wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: one VGPR per
// address instead of two and no 64-bit VALU add per issue.
__device__ __forceinline__ void glds16_so(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// Two copies of a tile whose destinations are 4096 B apart: ONE M0 setting (destination + 2048) and the instruction offsets -2048 / +2048. The
// offset moves the LDS address AND the global address, so the caller passes per-lane source offsets that carry the opposite 2048
// (voff0 + 2048, voff1 - 2048): 5 instructions instead of 10.
__device__ __forceinline__ void glds16_so_pair4k(const void* sbase, unsigned voff0_plus2k, unsigned voff1_minus2k, unsigned lds_dst_plus2k) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %3 offset:-2048\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff0_plus2k), "v"(voff1_minus2k), "s"(sbase), "s"(lds_dst_plus2k) : "memory");
}

__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- per-DEVICE one-time host state -----------------------------------------------------------------------------------
// The > 64 KiB dynamic-LDS attribute of a kernel and the CU count belong to a device, not to the process: a second GPU in
// the same process (model on cuda:1) must get its own hipFuncSetAttribute. Everything is keyed by the CURRENT device id;
// the Python front end guarantees the operands live on that device (cor_amd/ops.py:_dev).
constexpr int COR_MAX_DEVICES = 64;
static inline int cor_cur_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < COR_MAX_DEVICES) ? d : 0;
}
// Several host threads may launch concurrently (the header promises it): the flags are atomics. The guarded calls are idempotent
// (the same attribute value / the same property), so two threads that both find a flag clear may both make the call; what the
// acquire / release pair guarantees is that a thread which sees the flag set also sees the call completed.
struct DevOnce { std::atomic<bool> done[COR_MAX_DEVICES] = {}; };
static inline void cor_max_dyn_lds(const void* fn, int bytes, DevOnce& once) {
  const int d = cor_cur_device();
  if (!once.done[d].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    once.done[d].store(true, std::memory_order_release);
  }
}
static inline int cor_device_cus() {
  static std::atomic<int> n[COR_MAX_DEVICES] = {};
  const int d = cor_cur_device();
  int v = n[d].load(std::memory_order_acquire);
  if (v == 0) {
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    n[d].store(v, std::memory_order_release);
  }
  return v;
}
