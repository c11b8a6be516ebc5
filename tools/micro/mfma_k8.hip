// how many cycles does the legacy v_mfma_f32_32x32x8_f16 take on gfx950?  (candidate for the 4-of-16 valid last k-step of K = 180)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int K8>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
  h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (K8) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* clk, h[256];
  hipMalloc(&out, 4 * 256 * 256); hipMalloc(&clk, 8 * 256);
  const int iters = 2000;
  for (int v = 0; v < 2; ++v) {
    if (v) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, clk, iters); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize(); hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= 256;
    printf("%s: %.1f ticks per instruction (one wave per SIMD, dependent chain)\n", v ? "v_mfma_f32_32x32x8_f16 " : "v_mfma_f32_32x32x16_f16", m / (16.0 * iters));
  }
  return 0;
}
