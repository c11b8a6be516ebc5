// Semantics of the gfx950 fp8 conversions the 3-byte G plane relies on (round 6): v_cvt_pk_fp8_f32 (rounding, saturation, subnormals of OCP
// e4m3) and v_cvt_scalef32_pk_f16_fp8 (does the scale multiply or divide? is it exact?).  Prints a table; run on the GPU box.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/fp8_cvt tools/micro/fp8_cvt.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, unsigned* packed, float* back, float scale) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const float a = in[i], b = -in[i];
  const int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  packed[i] = (unsigned)p & 0xffffu;
  const h2 r = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)p, scale, false);
  back[2 * i] = (float)r[0];
  back[2 * i + 1] = (float)r[1];
}
int main() {
  const float vals[] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 0.3f, 3.3f, 240.f, 447.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1000.f, 1e30f,
                        0.015625f, 0.0078125f, 0.001953125f, 0.0009765625f, 0.0029f, 1e-4f};
  const int n = sizeof(vals) / sizeof(float);
  float *d_in, *d_back;
  unsigned* d_p;
  hipMalloc(&d_in, n * 4);
  hipMalloc(&d_back, n * 8);
  hipMalloc(&d_p, n * 4);
  hipMemcpy(d_in, vals, n * 4, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 0.001953125f, 512.0f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, n, d_p, d_back, scale);
    float back[2 * 64];
    unsigned p[64];
    hipMemcpy(back, d_back, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(p, d_p, n * 4, hipMemcpyDeviceToHost);
    printf("scale %g\n", scale);
    for (int i = 0; i < n; ++i) printf("  in %-12g fp8 bytes %02x %02x  -> f16 %-14g %-14g\n", vals[i], p[i] & 0xff, (p[i] >> 8) & 0xff, back[2 * i], back[2 * i + 1]);
  }
  return 0;
}
