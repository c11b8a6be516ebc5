// Diagnostic micro-benchmark: what does HBM take for a pure streaming WRITE on this chip (stage A's ceiling: it writes G, ~0.9 GB per launch)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int NT>
__global__ __launch_bounds__(256) void fill(f32x4* __restrict__ p, size_t nq, float v) {
  const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
  }
}
// half lines: every lane writes 16 bytes at stride 32 (the hi pieces of stage A's plane drain before the lo pieces complete the lines)
__global__ __launch_bounds__(256) void fill_half(f32x4* __restrict__ p, size_t nq, float v) {
  const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; 2 * i + 1 < nq; i += (size_t)gridDim.x * 256) { p[2 * i] = x; p[2 * i + 1] = x; }
}
// stage A's store pattern: a wave writes a 32-row x 128-byte tile with four instructions of 8 rows x 128 bytes, rows LD bytes apart; a
// workgroup of 4 waves x 3 tiles covers 1.5 KiB of 32 rows; the grid walks the columns of a row block first.
__global__ __launch_bounds__(256) void fill_tiles(float* __restrict__ p, int nrows, int ld, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_per_row = ld / 32, groups = tiles_per_row / 12;
  const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
  for (int job = blockIdx.x; job < (nrows / 32) * groups; job += gridDim.x) {
    const int rb = job / groups, cg = job % groups;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t row = (size_t)rb * 32 + 8 * q + (lane >> 3);
        *reinterpret_cast<f32x4*>(p + row * ld + (size_t)(cg * 12 + wave * 3 + t) * 32 + 4 * (lane & 7)) = x;
      }
  }
}
int main() {
  const size_t bytes = (size_t)4 << 30, nq = bytes / 16;
  f32x4* p; CK(hipMalloc(&p, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {2048, 16384}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), 0, 0, p, nq, 1.f);
        else if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(grid), dim3(256), 0, 0, p, nq, 1.f);
        else if (mode == 2) hipLaunchKernelGGL(fill_half, dim3(grid), dim3(256), 0, 0, p, nq, 1.f);
        else CK(hipMemsetAsync(p, 0x11, bytes, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      const char* nm[] = {"16-byte stores, lane-linear", "... nontemporal", "two 16-byte stores per lane (32-byte lane stride)", "hipMemsetAsync"};
      printf("%-52s grid %5d: %.3f ms for 4 GiB = %.2f TB/s\n", nm[mode], grid, best, bytes / best * 1e-9);
    }
  {
    const int ld = 13344 / 384 * 384 + 384, nrows = (int)(bytes / 4 / ld) / 32 * 32;     // (13440 floats per row: 35 groups of 12 tiles)
    for (int grid : {2048, 16384}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill_tiles, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float*>(p), nrows, ld, 1.f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      printf("%-52s grid %5d: %.3f ms for %.2f GiB = %.2f TB/s\n", "32-row x 128-byte tiles, rows 53 KB apart (stage A)", grid, best, (double)nrows * ld * 4 / (1 << 30), (double)nrows * ld * 4 / best * 1e-9);
    }
  }
  return 0;
}
