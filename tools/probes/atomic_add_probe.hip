// Bandwidth of non-returning fp32 atomic adds (global_atomic_add_f32, performed at L2) against a load-add-store pass over the same
// 131072 x 768 fp32 matrix (the residual stream): is "fire and forget" residual accumulation an option for the GEMM epilogue?
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Wno-unused-value -o /tmp/atomic_add_probe tools/probes/atomic_add_probe.hip && /tmp/atomic_add_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_atomic(float* x, long n, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) unsafeAtomicAdd(x + i, v);           // no return value used: global_atomic_add_f32 without glc
}
__global__ void k_rmw(float* x, long n, float v) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) { float4 t = *(float4*)(x + i); t.x += v; t.y += v; t.z += v; t.w += v; *(float4*)(x + i) = t; }
}
// the epilogue's access pattern: a wave adds 32 rows x 32 columns (128 B per row) per instruction group, rows 3072 B apart
__global__ void k_atomic_tile(float* x, int rows, int cols, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = cols / 32, tiles = (rows / 32) * tiles_n;
  for (int t = blockIdx.x * (blockDim.x >> 6) + wave; t < tiles; t += gridDim.x * (blockDim.x >> 6)) {
    const int r0 = (t / tiles_n) * 32, c0 = (t % tiles_n) * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                 // lane -> (row 2i + lane/32, column lane%32): two 128-B row pieces per instruction
      unsafeAtomicAdd(x + (long)(r0 + 2 * i + (lane >> 5)) * cols + c0 + (lane & 31), v);
    }
  }
}

int main() {
  const int rows = 131072, cols = 768;
  const long n = (long)rows * cols;
  float* x;
  hipMalloc(&x, n * 4);
  hipMemset(x, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("{\"kernel\": \"%s\", \"us\": %.1f, \"payload_GBps\": %.0f, \"hbm_GBps_if_read_plus_write\": %.0f}\n", name, best * 1e3, n * 4 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
  };
  run("atomic_add_f32 linear, 2048 blocks", [&] { hipLaunchKernelGGL(k_atomic, dim3(2048), dim3(256), 0, 0, x, n, 1.0f); }, 2.0 * n * 4);
  run("atomic_add_f32 linear, 8192 blocks", [&] { hipLaunchKernelGGL(k_atomic, dim3(8192), dim3(256), 0, 0, x, n, 1.0f); }, 2.0 * n * 4);
  run("load + add + store float4, 4096 blocks", [&] { hipLaunchKernelGGL(k_rmw, dim3(4096), dim3(256), 0, 0, x, n, 1.0f); }, 2.0 * n * 4);
  run("atomic_add_f32 in 32x32 tiles (epilogue pattern), 2048 blocks", [&] { hipLaunchKernelGGL(k_atomic_tile, dim3(2048), dim3(256), 0, 0, x, rows, cols, 1.0f); }, 2.0 * n * 4);
  std::vector<float> h(8);
  hipMemcpy(h.data(), x, 32, hipMemcpyDeviceToHost);
  printf("{\"check_first_element\": %.1f}\n", h[0]);
  return 0;
}
