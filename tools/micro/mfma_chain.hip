// Diagnostic micro-benchmark (not part of the product): what does a v_mfma_f32_32x32x16_f16 cost per wave on a LOADED chip?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
// Every CU runs W waves per SIMD (256 registers each at W = 2); a wave issues N MFMAs from registers on CH independent accumulators
// (CH = 1: one dependent chain, the tile products of ddp_conv_rows; CH = 2: the am / ac pair of the 2048-scaled form).  Reported:
// wall-clock TFLOP/s, s_memtime ticks per MFMA and wave, and the shader clock the ticks imply (s_memtime / s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CH>
__global__ __launch_bounds__(256, 2) void chain(float* out, unsigned long long* clk, int iters, int pad_regs, const h8* __restrict__ rnd) {
  extern __shared__ float lds_pad[];     // (70 KiB per workgroup: at most two workgroups per CU, i.e. exactly W waves per SIMD)
  if (pad_regs == 12345) lds_pad[threadIdx.x] = 1.f;
  const int lane = threadIdx.x & 63;
  h8 a[12], b[12];
#pragma unroll
  for (int k = 0; k < 12; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[k][i] = (_Float16)(0.001f * (lane + k + i)); b[k][i] = (_Float16)(0.002f * (lane - k + i)); }
  if (rnd) {     // operands of random bits (what real data looks like to the multipliers)
#pragma unroll
    for (int k = 0; k < 12; ++k) { a[k] = rnd[(2 * k) * 64 + lane]; b[k] = rnd[(2 * k + 1) * 64 + lane]; }
  }
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[0], 0, 0, 0);
      acc[CH - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + 1) % 12], acc[CH - 1], 0, 0, 0);
      acc[CH - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 1) % 12], b[k], acc[CH - 1], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  (void)pad_regs;
}

int main() {
  const int iters = 4000;
  float* out; unsigned long long* clk;
  const int maxwg = 256 * 2;
  CK(hipMalloc(&out, sizeof(float) * maxwg * 256));
  CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * maxwg));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  h8* rnd;
  {
    _Float16 hr[24 * 64 * 8];
    srand(3);
    for (auto& v : hr) v = (_Float16)((rand() / (float)RAND_MAX) * 4.f - 2.f);
    CK(hipMalloc(&rnd, sizeof(hr)));
    CK(hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice));
  }
  for (int random = 0; random <= 1; ++random)
  for (int ch = 1; ch <= 2; ++ch)
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
      const int nwg = 256 * wgs_per_cu;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (ch == 1) hipLaunchKernelGGL(chain<1>, dim3(nwg), dim3(256), 70 * 1024, 0, out, clk, iters, 0, random ? rnd : nullptr);
        else hipLaunchKernelGGL(chain<2>, dim3(nwg), dim3(256), 70 * 1024, 0, out, clk, iters, 0, random ? rnd : nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[2 * 512]; CK(hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost));
      double ticks = 0, real = 0; for (int i = 0; i < nwg; ++i) { ticks += h[2 * i]; real += h[2 * i + 1]; }
      ticks /= nwg; real /= nwg;
      const double nm = 36.0 * iters, fl = nm * 32768.0 * nwg * 4;
      printf("%s operands, chains %d, %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s fp16, %.1f ticks per MFMA and wave, clock %.0f MHz (ticks / realtime x 100 MHz)\n", random ? "random" : "smooth", ch, wgs_per_cu, ms,
             fl / ms * 1e-9, ticks / nm, ticks / real * 100.0);
    }
  return 0;
}
