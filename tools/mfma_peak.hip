// Calibration microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on the whole chip and the tick rates of
// s_memtime / s_memrealtime.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int nacc>
__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* stamps, int iters) {
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    if (nacc > 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
    if (nacc > 2) {
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks_full = 256;
  float* out; unsigned long long* st;
  hipMalloc(&out, sizeof(float) * 256 * 1024); hipMalloc(&st, sizeof(unsigned long long) * 2 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int blocks, threads, nacc, iters; const char* name; };
  Cfg cfgs[] = {{1, 64, 1, 200000, "1 wave, 1 dependent accumulator"}, {1, 64, 4, 50000, "1 wave, 4 accumulators"},
                {1, 64, 2, 100000, "1 wave, 2 accumulators"},
                {blocks_full, 256, 1, 100000, "256 blocks x 4 waves (1/SIMD), 1 acc"},
                {blocks_full, 256, 4, 50000, "256 blocks x 4 waves (1/SIMD), 4 acc"},
                {blocks_full, 512, 1, 100000, "256 blocks x 8 waves (2/SIMD), 1 acc"},
                {blocks_full, 512, 4, 50000, "256 blocks x 8 waves (2/SIMD), 4 acc"},
                {blocks_full, 256, 4, 400000, "256 blocks x 4 waves, 4 acc, long (DVFS)"}};
  for (auto& c : cfgs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (c.nacc == 1)
        hipLaunchKernelGGL(mfma_loop<1>, dim3(c.blocks), dim3(c.threads), 0, 0, out, st, c.iters);
      else if (c.nacc == 2)
        hipLaunchKernelGGL(mfma_loop<2>, dim3(c.blocks), dim3(c.threads), 0, 0, out, st, c.iters);
      else
        hipLaunchKernelGGL(mfma_loop<4>, dim3(c.blocks), dim3(c.threads), 0, 0, out, st, c.iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * c.blocks);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * c.blocks, hipMemcpyDeviceToHost);
    double nm = (double)c.iters * (c.nacc > 2 ? 4 : c.nacc);
    double waves = (double)c.blocks * c.threads / 64;
    double tf = nm * waves * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-45s event %.3f ms  memtime ticks/MFMA(wave) %.1f  memtime %.0f MHz  memrealtime %.1f MHz  => %.1f TFLOP/s\n", c.name, ms,
           h[0] / nm, h[0] / (ms * 1e3), h[1] / (ms * 1e3), tf);
  }
  return 0;
}
