// Diagnostic micro-benchmark (not part of the product): an fp32 product on the fp16 matrix cores with BOTH operands split into
// two halves, v = hi + lo / 2048 (hi = fp16(v), lo = fp16((v - hi) * 2048): 22 significant bits, the low part scaled back into
// fp16's normal range), three products per 16 k on v_mfma_f32_32x32x16_f16:
//     x w ~ xh wh + (xh wl + xl wh) / 2048            dropped: xl wl / 2^22
// (1) accuracy against an fp64 host product, next to the exact fp32 MFMA chain; (2) sustained rate of both forms on random data.
//   hipcc --offload-arch=gfx950 -O3 -o f16x2_mfma f16x2_mfma.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int K = 192;

__device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * 2048.f);
}

// one wave: C[32][32] = A[32][K] B[K][32], both forms
__global__ void acc_kernel(const float* A, const float* B, float* C32, float* C16) {
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  f32x16 c = {0};
  for (int k = 0; k < K; k += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + hh], B[(k + hh) * 32 + r], c, 0, 0, 0);
  f32x16 m = {0}, s = {0};
  for (int k = 0; k < K; k += 16) {
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      _Float16 x, y;
      split(A[r * K + k + 8 * hh + j], x, y); ah[j] = x; al[j] = y;
      split(B[(k + 8 * hh + j) * 32 + r], x, y); bh[j] = x; bl[j] = y;
    }
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, s, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
    C32[row * 32 + r] = c[i];
    C16[row * 32 + r] = m[i] + s[i] * (1.f / 2048.f);
  }
}

// rate: every wave issues NIT x (K/2 fp32 MFMAs | 3 K/16 fp16 MFMAs) on register operands (random bits, finite)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const float* seed, float* sink, int nit, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = seed[(lane * 8 + j) & 1023]; b[j] = seed[(lane * 8 + j + 512) & 1023]; }
  h8 ah, al, bh, bl;
  for (int j = 0; j < 8; ++j) { _Float16 x, y; split(a[j], x, y); ah[j] = x; al[j] = y; split(b[j], x, y); bh[j] = x; bl[j] = y; }
  f32x16 c0 = {0}, c1 = {0};
  unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < nit; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < K / 2; ++k) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 7], b[(k + 3) & 7], c0, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < K / 16; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c1, 0, 0, 0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
  if (acc == 12345.678f) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  std::vector<float> A(32 * K), B(K * 32);
  srand(1);
  auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
  for (int i = 0; i < 32 * K; ++i) { float v = rnd() * 3.f; A[i] = v > 0 ? v : 0.f; if (i % 7 == 0) A[i] *= 1e-5f; }   // relu-like, some tiny
  for (int i = 0; i < K * 32; ++i) { B[i] = rnd() * 0.2f; if (i % 11 == 0) B[i] *= 1e-4f; }
  float *dA, *dB, *dC32, *dC16;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC32, 4096)); CK(hipMalloc(&dC16, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC32, dC16);
  std::vector<float> C32(1024), C16(1024);
  CK(hipMemcpy(C32.data(), dC32, 4096, hipMemcpyDeviceToHost));
  CK(hipMemcpy(C16.data(), dC16, 4096, hipMemcpyDeviceToHost));
  double e32 = 0, e16 = 0, e32r = 0, e16r = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, sab = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; sab += fabs((double)A[i * K + k] * B[k * 32 + j]); }
      e32 = fmax(e32, fabs(C32[i * 32 + j] - ref) / sab); e16 = fmax(e16, fabs(C16[i * 32 + j] - ref) / sab);
      e32r += fabs(C32[i * 32 + j] - ref) / sab / 1024; e16r += fabs(C16[i * 32 + j] - ref) / sab / 1024;
    }
  printf("accuracy vs fp64, |err| / sum|a b|:  fp32 MFMA max %.3e mean %.3e   fp16 hi/lo max %.3e mean %.3e   (2^-21 = %.3e)\n", e32, e32r, e16, e16r, pow(2.0, -21));

  float *seed, *sink; unsigned long long* clk;
  std::vector<float> S(1024);
  for (auto& v : S) v = rnd();
  CK(hipMalloc(&seed, 4096)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clk, 64));
  CK(hipMemcpy(seed, S.data(), 4096, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int wpc : {4, 8}) {     // waves per CU (1 or 2 per SIMD)
      const int nit = 4000, grid = 256 * wpc / 4;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(grid), dim3(256), 0, 0, seed, sink, nit, clk);
        else hipLaunchKernelGGL((rate_kernel<1>), dim3(grid), dim3(256), 0, 0, seed, sink, nit, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      const double flop_equiv = 2.0 * 32 * 32 * K * (double)nit * grid * 4;      // fp32-equivalent FLOPs of the products
      printf("%s, %d waves/CU: %.3f ms, %.1f TFLOP/s fp32-equivalent, in-kernel clock %.2f GHz\n", mode ? "fp16 hi/lo (3 MFMA per 16 k)" : "fp32 MFMA 32x32x2          ",
             wpc, best, flop_equiv / best / 1e9, (double)c[0] / c[1] * 0.1);
    }
  return 0;
}
