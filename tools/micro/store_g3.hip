// Diagnostic micro-benchmark (round 6, not part of the product): HBM write rate of stage A's drain in the two plane forms of G.
// Rows of 40192 B (plane form 1) / 53376 B (plane form 0), 44440 rows x 2 slices; a wave covers 4 product blocks (16 G columns) of 32-row
// tiles, a workgroup 512 rows, as ddp_stage_a_h2_kernel<60, GH> does.   hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_g3 tools/micro/store_g3.hip
//   F0  form 0 as shipped: per block and half tile a 16-row x 64-B hi store and a 16-row x 64-B lo store that interleave into whole 128-B lines
//   G0  form 1 as first built: per block and half tile a 16-row x 64-B hi store (dwordx4) and a 16-row x 32-B lo store (dwordx2), hi and lo regions apart
//   G1  ... the lo store as dwordx4 from every second lane (pairs of columns)
//   G2  ... the stores of a row tile's 4 blocks grouped: 8 hi stores back to back, then 8 lo stores (what registers would have to hold)
//   G3  ... lo pieces padded to 16 B (no byte saving: is it the 32-B segment?)
//   G4  G0 with non-temporal stores
//   H0  24-byte units (hi 16 B + lo 8 B side by side: a wave's 16 units = 384 contiguous bytes per row): dwordx4 + dwordx2 per block and half tile
//   H1  ... as 12-byte pieces (4 values: 4 fp16 + 4 bytes), one dwordx3 store per lane, 8 rows x 96 B per instruction, 4 per block
//   H2  H1 with the 4 blocks of a row group back to back (what holding a row tile's four blocks would allow)
//   Every mode also with the stores spaced as in the kernel (SL > 0: s_sleep behind every store instruction, two workgroups per CU through LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NROWS = 44440, NZ = 2, N8 = 23, GCP = 72;
constexpr int LD0 = 13344, LD1 = 10048;     // floats per row

// a wave's 16 G columns at k8 group `g8`: byte offsets inside a row.  Parts are ignored (one tile of GCP columns): the access shape is the same
typedef float f32x3 __attribute__((ext_vector_type(3)));
template <int MODE, int SL = 0>
__global__ __launch_bounds__(256) void drain(float* out, int mrows) {
  extern __shared__ float dyn_lds[];
  if (mrows < 0) dyn_lds[threadIdx.x] = 0.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int unit0 = ((int)blockIdx.x * 4 + wave) * 16;       // first (k8, c) unit of the wave: 16 consecutive units
  if (unit0 >= N8 * GCP) return;
  const int ld = (MODE == 0 || MODE == 4) ? LD0 : LD1;
  char* ob = reinterpret_cast<char*>(out) + (size_t)blockIdx.z * NROWS * ld * 4;
  const int R0 = (int)blockIdx.y * mrows, R1 = min(NROWS, R0 + mrows);
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  const f32x2 v2 = {1.f, (float)lane};
  const int g = lane & 3;
  for (int row0 = R0; row0 < R1; row0 += 32) {
    if (MODE == 3) {        // grouped: all hi stores of the row tile, then all lo stores
      for (int t = 0; t < 4; ++t)
        for (int h = 0; h < 2; ++h) {
          const int rr = min(row0 + 16 * h + (lane >> 2), R1 - 1), u = unit0 + 4 * t + g;
          *reinterpret_cast<f32x4*>(ob + (size_t)rr * ld * 4 + (size_t)u * 16) = v;
        }
      for (int t = 0; t < 4; ++t)
        for (int h = 0; h < 2; ++h) {
          const int rr = min(row0 + 16 * h + (lane >> 2), R1 - 1), u = unit0 + 4 * t + g;
          *reinterpret_cast<f32x2*>(ob + (size_t)rr * ld * 4 + (size_t)N8 * GCP * 16 + (size_t)u * 8) = v2;
        }
      continue;
    }
    if (MODE == 7 || MODE == 8) {      // 12-byte pieces: lane (row 8 p + lane / 8, piece lane % 8) of quarter p of block t
      const f32x3 v3 = {1.f, 2.f, (float)lane};
      if (MODE == 7) {
        for (int t = 0; t < 4; ++t)
          for (int p = 0; p < 4; ++p) {
            const int rr = min(row0 + 8 * p + (lane >> 3), R1 - 1);
            *reinterpret_cast<f32x3*>(ob + (size_t)rr * ld * 4 + (size_t)(unit0 + 4 * t) * 24 + 12 * (lane & 7)) = v3;
            if (SL) __builtin_amdgcn_s_sleep(SL);
          }
      } else {
        for (int p = 0; p < 4; ++p)
          for (int t = 0; t < 4; ++t) {
            const int rr = min(row0 + 8 * p + (lane >> 3), R1 - 1);
            *reinterpret_cast<f32x3*>(ob + (size_t)rr * ld * 4 + (size_t)(unit0 + 4 * t) * 24 + 12 * (lane & 7)) = v3;
            if (SL) __builtin_amdgcn_s_sleep(SL);
          }
      }
      continue;
    }
    for (int t = 0; t < 4; ++t)
      for (int h = 0; h < 2; ++h) {
        const int rr = min(row0 + 16 * h + (lane >> 2), R1 - 1), u = unit0 + 4 * t + g;
        char* rowp = ob + (size_t)rr * ld * 4;
        if (MODE == 6) {
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 24) = v;
          if (SL) __builtin_amdgcn_s_sleep(SL);
          *reinterpret_cast<f32x2*>(rowp + (size_t)u * 24 + 16) = v2;
          if (SL) __builtin_amdgcn_s_sleep(SL);
        } else if (MODE == 0) {
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 32) = v;
          if (SL) __builtin_amdgcn_s_sleep(SL);
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 32 + 16) = v;
          if (SL) __builtin_amdgcn_s_sleep(SL);
        } else if (MODE == 1) {
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 16) = v;
          if (SL) __builtin_amdgcn_s_sleep(SL);
          *reinterpret_cast<f32x2*>(rowp + (size_t)N8 * GCP * 16 + (size_t)u * 8) = v2;
          if (SL) __builtin_amdgcn_s_sleep(SL);
        } else if (MODE == 2) {
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 16) = v;
          if ((lane & 1) == 0) *reinterpret_cast<f32x4*>(rowp + (size_t)N8 * GCP * 16 + (size_t)u * 8) = v;
        } else if (MODE == 4) {   // lo padded to 16 B: a second full-size region (the row is longer: reads LD0-sized rows)
          *reinterpret_cast<f32x4*>(rowp + (size_t)u * 16) = v;
          *reinterpret_cast<f32x4*>(rowp + (size_t)N8 * GCP * 16 + (size_t)u * 16) = v;
        } else if (MODE == 5) {
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(rowp + (size_t)u * 16));
          __builtin_nontemporal_store(v2, reinterpret_cast<f32x2*>(rowp + (size_t)N8 * GCP * 16 + (size_t)u * 8));
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
  float* out;
  CK(hipMalloc(&out, (size_t)NZ * NROWS * LD0 * 4));
  const dim3 grid((N8 * GCP + 63) / 64, (NROWS + 511) / 512, NZ);
  const double gb0 = (double)NZ * NROWS * N8 * GCP * 32 / 1e9, gb1 = (double)NZ * NROWS * N8 * GCP * 24 / 1e9;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL((drain<0>), grid, dim3(256), 0, 0, out, 512); });
  printf("F0 form 0: hi + lo 16-byte pieces interleaved (whole lines per store pair)   %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb0, gb0 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<1>), grid, dim3(256), 0, 0, out, 512); });
  printf("G0 form 1: 16 rows x 64 B hi (x4) + 16 rows x 32 B lo (x2)                    %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<2>), grid, dim3(256), 0, 0, out, 512); });
  printf("G1 form 1: lo as dwordx4 from every second lane                                %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<3>), grid, dim3(256), 0, 0, out, 512); });
  printf("G2 form 1: a row tile's 8 hi stores back to back, then its 8 lo stores          %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<4>), grid, dim3(256), 0, 0, out, 512); });
  printf("G3 hi region + lo region of 16-byte pieces (no byte saving)                    %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb0, gb0 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<5>), grid, dim3(256), 0, 0, out, 512); });
  printf("G4 form 1 with non-temporal stores                                             %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<6>), grid, dim3(256), 0, 0, out, 512); });
  printf("H0 24-byte units: dwordx4 + dwordx2 per block and half tile                    %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<7>), grid, dim3(256), 0, 0, out, 512); });
  printf("H1 24-byte units as 12-byte pieces: one dwordx3 per lane, 8 rows x 96 B        %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<8>), grid, dim3(256), 0, 0, out, 512); });
  printf("H2 ... the four blocks of a row group back to back                             %.3f ms  %.2f GB  %.2f TB/s\n", ms, gb1, gb1 / ms);
  // the same with the stores spaced as in the kernel: two workgroups per CU (70 KiB of dynamic LDS each), s_sleep(2) = ~128 cycles behind every store
  const size_t lds = 70 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(drain<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(drain<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(drain<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(drain<7, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(drain<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ms = timeit([&] { hipLaunchKernelGGL((drain<0, 2>), grid, dim3(256), lds, 0, out, 512); });
  printf("spaced, two workgroups per CU:  F0 %.3f ms %.2f TB/s", ms, gb0 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<1, 2>), grid, dim3(256), lds, 0, out, 512); });
  printf(" | G0 %.3f ms %.2f TB/s", ms, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<6, 2>), grid, dim3(256), lds, 0, out, 512); });
  printf(" | H0 %.3f ms %.2f TB/s", ms, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<7, 2>), grid, dim3(256), lds, 0, out, 512); });
  printf(" | H1 %.3f ms %.2f TB/s", ms, gb1 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((drain<8, 2>), grid, dim3(256), lds, 0, out, 512); });
  printf(" | H2 %.3f ms %.2f TB/s\n", ms, gb1 / ms);
  return 0;
}
