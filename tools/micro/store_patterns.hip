// Diagnostic micro-benchmark (not part of the product): HBM WRITE rate of the stage-A store pattern against other shapes of
// the same 4.5 GB (44440 rows x 12672 floats x 2 slices, row pitch 50688 B).   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
//   P0  linear stream (each wave-instruction 1 KiB contiguous, consecutive waves consecutive KiB)
//   P1  stage A today: a wave-instruction writes 8 rows x 128 B; a wave covers 4 column tiles of 32 floats, a workgroup 512 rows
//   P2  2 rows x 512 B per wave-instruction (a wave's 128 columns of a row at once)
//   P3  1 row x 1 KiB per wave-instruction (256 columns per wave, 4 KiB per workgroup and row)
//   P4  like P1 with 64-B segments (16 rows x 64 B): the direct 16x16x4 C/D store without a transpose
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NROWS = 44440, NCOLS = 12672, NZ = 2;

__global__ __launch_bounds__(256) void p0(float* out, size_t n4) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<f32x4*>(out)[i] = v;
}

// grid (ceil(NCOLS / (WC * 4 waves)), ceil(NROWS / 512), NZ); SEG = floats per contiguous segment of a wave-instruction
template <int SEG>
__global__ __launch_bounds__(256) void pat(float* out, int mrows) {
  constexpr int LPS = SEG / 4;          // lanes per segment
  constexpr int RPI = 64 / LPS;         // rows per wave-instruction
  constexpr int WC = (SEG >= 128) ? SEG : 128;   // columns a wave covers per row
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = ((int)blockIdx.x * 4 + wave) * WC;
  float* ob = out + (size_t)blockIdx.z * NROWS * NCOLS;
  const int R0 = (int)blockIdx.y * mrows, R1 = min(NROWS, R0 + mrows);
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int row0 = R0; row0 < R1; row0 += 32) {
    for (int cs = 0; cs < WC; cs += SEG) {            // segments of the wave's column range
      for (int p = 0; p < 32 / RPI; ++p) {
        const int rr = row0 + p * RPI + lane / LPS, c = c0 + cs + 4 * (lane % LPS);
        if (rr < R1 && c < NCOLS) *reinterpret_cast<f32x4*>(&ob[(size_t)rr * NCOLS + c]) = v;
      }
    }
  }
}

template <typename F>
static float timeit(F f, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const size_t n = (size_t)NZ * NROWS * NCOLS;
  float* out;
  CK(hipMalloc(&out, n * 4));
  const double gb = n * 4 / 1e9;
  float ms = timeit([&] { hipLaunchKernelGGL(p0, dim3(256 * 16), dim3(256), 0, 0, out, n / 4); });
  printf("P0 linear                      %.3f ms  %.2f TB/s\n", ms, gb / ms);
  for (int mrows : {512, 128}) {
    const dim3 gy((NCOLS + 511) / 512, (NROWS + mrows - 1) / mrows, NZ);
    ms = timeit([&] { hipLaunchKernelGGL((pat<32>), gy, dim3(256), 0, 0, out, mrows); });
    printf("P1 8 rows x 128 B  (mrows %3d)  %.3f ms  %.2f TB/s\n", mrows, ms, gb / ms);
    ms = timeit([&] { hipLaunchKernelGGL((pat<128>), gy, dim3(256), 0, 0, out, mrows); });
    printf("P2 2 rows x 512 B  (mrows %3d)  %.3f ms  %.2f TB/s\n", mrows, ms, gb / ms);
    const dim3 g3((NCOLS + 1023) / 1024, (NROWS + mrows - 1) / mrows, NZ);
    ms = timeit([&] { hipLaunchKernelGGL((pat<256>), g3, dim3(256), 0, 0, out, mrows); });
    printf("P3 1 row  x 1 KiB  (mrows %3d)  %.3f ms  %.2f TB/s\n", mrows, ms, gb / ms);
    ms = timeit([&] { hipLaunchKernelGGL((pat<16>), gy, dim3(256), 0, 0, out, mrows); });
    printf("P4 16 rows x 64 B  (mrows %3d)  %.3f ms  %.2f TB/s\n", mrows, ms, gb / ms);
  }
  return 0;
}
