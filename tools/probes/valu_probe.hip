// Stand-alone micro-benchmark (not part of libcor_amd.so): issue cost of vector instructions on gfx950 as a wave sees it
//   (a) alone on its SIMD, (b) beside a second VALU wave, (c) beside a partner wave issuing back-to-back MFMAs.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: v_fma_f32 independent (32 accumulators); 1: v_max_f32; 2: v_exp_f32; 3: v_cvt_pk_bf16; 4: dependent v_fma chain
template <int MODE>
__device__ __forceinline__ void valu_body(float (&x)[32], float a, float b) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (MODE == 0) x[i] = fmaf(x[i], a, b);
    else if (MODE == 1) x[i] = fmaxf(x[i], x[(i + 7) & 31] * 0.f + b);
    else if (MODE == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
    else if (MODE == 4) x[0] = fmaf(x[0], a, b);
  }
}

// grid = #CUs blocks; block = W waves. Waves with id < nvalu run the VALU loop, the others issue MFMAs (partner = 1) or exit.
template <int MODE>
__global__ void __launch_bounds__(512) probe(unsigned long long* out, int nvalu_waves, int partner, int iters, float a, float b, int prio, int gap) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = (float)(lane + i) * 1e-3f;
  f32x16 acc = {0};
  const uint4 fa = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < nvalu_waves) {
    if (prio) __builtin_amdgcn_s_setprio(1);
    for (int it = 0; it < iters; ++it) { valu_body<MODE>(x, a, b); __builtin_amdgcn_sched_barrier(0); }
  } else if (partner == 1) {
    // the partner outlasts the VALU waves (3x the iterations): the VALU waves' time is all "beside MFMA"
    for (int it = 0; it < 3 * iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fa), acc, 0, 0, 0);
        if (gap == 1) asm volatile("s_nop 7");
        if (gap == 2) { asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); asm volatile("s_nop 7"); }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = acc[0];
#pragma unroll
  for (int i = 0; i < 32; ++i) sum += x[i];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)(sum != 1.2345f); }
}

// One instruction stream per wave: 1 MFMA followed by NV independent v_fma_f32 (or exp2 when EXP), `waves` waves per block.
template <int NV, bool EXP>
__global__ void __launch_bounds__(512) interleave(unsigned long long* out, int iters, float a, float b) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (float)(lane + i) * 1e-3f;
  f32x16 acc0 = {0}, acc1 = {0};
  const uint4 fa = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fa), acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fa), acc0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) x[(g * NV + i) & 15] = EXP ? __builtin_amdgcn_exp2f(x[(g * NV + i) & 15]) : fmaf(x[(g * NV + i) & 15], a, b);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = acc0[0] + acc1[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += x[i];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)(sum != 1.2345f); }
}
template <int NV, bool EXP>
void run_il(unsigned long long* d, int waves) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((interleave<NV, EXP>), dim3(blocks), dim3(waves * 64), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((interleave<NV, EXP>), dim3(blocks), dim3(waves * 64), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), d, blocks * 8 * 16, hipMemcpyDeviceToHost);
  double v = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { v += h[(b * 8 + w) * 2]; ++n; }
  printf("{\"test\": \"1 MFMA + %d %s per group, same wave\", \"waves_per_SIMD\": %d, \"ticks_per_group_per_wave\": %.1f, \"ns_per_group_per_SIMD_wall\": %.2f}\n",
         NV, EXP ? "v_exp_f32" : "v_fma_f32", waves / 4, v / n / (iters * 8.0), ms * 1e6 / (iters * 8.0) / (waves / 4));
}

template <int MODE>
void run(const char* name, unsigned long long* d, int waves, int nvalu, int partner, int prio = 0, int gap = 0) {
  const int iters = 2000, blocks = 256;
  hipMemset(d, 0, blocks * 8 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(waves * 64), 0, 0, d, nvalu, partner, iters, 1.0001f, 0.5f, prio, gap);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(waves * 64), 0, 0, d, nvalu, partner, iters, 1.0001f, 0.5f, prio, gap);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), d, blocks * 8 * 16, hipMemcpyDeviceToHost);
  double v = 0, p = 0; int nv = 0, np = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < waves; ++w) { if (w < nvalu) { v += h[(b * 8 + w) * 2]; ++nv; } else { p += h[(b * 8 + w) * 2]; ++np; } }
  const double per_instr_ticks = v / nv / (iters * 32.0);
  // wall: the VALU waves' loop is the kernel when no partner outlasts it
  printf("{\"test\": \"%s\", \"waves_per_block\": %d, \"valu_waves\": %d, \"partner\": \"%s\", \"ticks_per_valu_instr\": %.2f, \"partner_ticks_per_mfma\": %.2f, \"kernel_us\": %.1f, \"ns_per_valu_instr_wall\": %.3f}\n",
         name, waves, nvalu, partner == 1 ? "mfma" : "none", per_instr_ticks, np ? p / np / (3 * iters * 8.0) : 0.0, ms * 1e3, ms * 1e6 / (iters * 32.0));
}

int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8 * 16);
  // one wave per SIMD (4 waves per block, one block per CU: 512-thread launch bound keeps 1 block/CU? use 4 waves)
  run<0>("fma, 1 wave/SIMD", d, 4, 4, 0);
  run<0>("fma, 2 waves/SIMD both VALU", d, 8, 8, 0);
  run<0>("fma beside back-to-back MFMA partner", d, 8, 4, 1);
  run<0>("fma (s_setprio 1) beside back-to-back MFMA partner", d, 8, 4, 1, 1);
  run<0>("fma beside MFMA partner with s_nop 7 after each MFMA", d, 8, 4, 1, 0, 1);
  run<0>("fma beside MFMA partner with 4 x s_nop 7 after each MFMA", d, 8, 4, 1, 0, 2);
  run<2>("exp2, 1 wave/SIMD", d, 4, 4, 0);
  run<2>("exp2 beside back-to-back MFMA partner", d, 8, 4, 1);
  run<4>("dependent fma chain, 1 wave/SIMD", d, 4, 4, 0);
  run<4>("dependent fma chain beside back-to-back MFMA partner", d, 8, 4, 1);
  run_il<0, false>(d, 4); run_il<0, false>(d, 8);
  run_il<4, false>(d, 4); run_il<4, false>(d, 8);
  run_il<6, false>(d, 4); run_il<6, false>(d, 8);
  run_il<8, false>(d, 4); run_il<8, false>(d, 8);
  run_il<12, false>(d, 4); run_il<12, false>(d, 8);
  run_il<2, true>(d, 4); run_il<2, true>(d, 8);
  run_il<4, true>(d, 4); run_il<4, true>(d, 8);
  return 0;
}
