// Diagnostic micro-benchmark (round 6, not part of the product): the stream-tile loop of ddp_conv_rows in the OTHER fp16 MFMA shape.
// Question ("next" 1): the kernel's tile products run on v_mfma_f32_32x32x16_f16; the MI355X guide (DVFS item 7)
// measured the 16x16x32 shape at 1.12 - 1.15 x the FLOP/s of 32x32x16 at equal cycles, because the chip holds a higher clock under it.
// Does that hold for THIS instruction mix - unified fp16 hi/lo planes (three MFMAs per product on one accumulator), weights once per
// workgroup through a three-slot LDS ring filled by buffer loads to LDS, bare barriers, the C-component feature contraction per tile?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_16 tools/micro/stream_16.hip && ./tools/micro/stream_16
// Form A = tools/micro/stream_wide.hip's form A (the shipped organisation: 32 edges per wave, two 4-wave workgroups per CU, third-tile ring).
// Form S = the same work per wave on v_mfma_f32_16x16x32_f16: the wave's 32 edges as two 16-row tiles, a 32-column weight tile as two
// 16-column tiles, six k-steps of 32 instead of twelve of 16: 12 MFMAs of 16 cycles per 32 k where form A has 6 of 32 cycles - the same
// pipe cycles, the same operand registers (96) and accumulator registers (16), the same LDS bytes per product; the feature rows of the
// epilogue are read as the 16x16 C layout wants them (lane = column l % 16, rows 4 (l / 16) .. + 3).  Random operands, results summed into
// a sink (both forms compute the same number of product FLOPs; their VALUES differ: the operand images are not permuted to match).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NS = 12, NF = 2 * NS, TILE_Q = NF * 64, NW = 4, NT = 256, FEAT_FLOATS = 60 * 36, NP = 3;

__device__ __forceinline__ f32x16 splat(float v) { f32x16 r; for (int i = 0; i < 16; ++i) r[i] = v; return r; }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

// SHAPE 32: acc 32x32 (16 registers), 3 MFMAs of 32x32x16 per 16 k.  SHAPE 16: four 16x16 accumulators (4 registers each), 12 MFMAs of
// 16x16x32 per 32 k.  The ring, the barriers, the waits and the piece size are the same.
template <int SHAPE, int C>
__global__ __launch_bounds__(NT, 2) void tile_kernel(const f32x4* __restrict__ w, const h8* __restrict__ a, float* __restrict__ out, int ntiles, int nseg,
                                                      unsigned long long* clk, unsigned long long* rclk) {
  constexpr int KPP = NS / NP, PIECE_Q = 2 * KPP * 64, FPW = 2 * KPP / NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float* feat = lds + 3 * PIECE_Q * 4 + wave * FEAT_FLOATS;
  for (int i = lane; i < FEAT_FLOATS; i += 64) feat[i] = 1e-3f * (float)((i * 7 + wave) & 31);
  // the A operand: 2 NS fragments of 16 bytes per lane either way (32x32x16: [ks][plane]; 16x16x32: [row tile][k32 step][plane])
  h8 ah[NS], al[NS];
  {
    const h8* ap = a + ((size_t)(blockIdx.x * NW + wave) * 2 * NS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) { ah[ks] = ap[(2 * ks) * 64]; al[ks] = ap[(2 * ks + 1) * 64]; }
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(w), 0, ntiles * TILE_Q * 16, 0x00020000);
  const int npieces = ntiles * NP;
  auto request = [&](int j, int slot) {
    const int off = min(j, npieces - 1) * (PIECE_Q * 16);
#pragma unroll
    for (int f = 0; f < FPW; ++f)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ring + slot * PIECE_Q + (wave + NW * f) * 64), 16, ((wave + NW * f) * 64 + lane) * 16, off, 0, 0);
  };
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  request(0, 0);
  request(1, 1);
  float sink = 0.f;
  const int tps = ntiles / nseg;
  int j = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    f32x16 res[C];
#pragma unroll
    for (int c = 0; c < C; ++c) res[c] = splat(0.f);
    for (int t = 0; t < tps; ++t) {
      f32x16 acc = splat(0.25f);               // (SHAPE 16: registers 4 s .. 4 s + 3 = sub-tile s = 2 rt + ct)
#pragma unroll
      for (int p = 0; p < NP; ++p, ++j) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        request(j + 2, (j + 2) % 3);
        const f32x4* slot = ring + (j % 3) * PIECE_Q;
        if constexpr (SHAPE == 32) {
          f32x4 b0 = slot[lane], b1 = slot[64 + lane];
#pragma unroll
          for (int k = 0; k < KPP; ++k) {
            const h8 bh = __builtin_bit_cast(h8, b0), bl = __builtin_bit_cast(h8, b1);
            if (k + 1 < KPP) { b0 = slot[(2 * k + 2) * 64 + lane]; b1 = slot[(2 * k + 3) * 64 + lane]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p * KPP + k], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[p * KPP + k], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[p * KPP + k], bh, acc, 0, 0, 0);
          }
        } else {
          // a piece = KPP k16 steps = KPP / 2 k32 steps; per k32 step and column tile one hi and one lo fragment of 1 KiB (the piece's
          // 2 KPP fragments: [k32 step][column tile][plane])
          static_assert(KPP % 2 == 0, "a piece holds whole k32 steps");
          f32x4 acc4[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) acc4[s] = f32x4{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
#pragma unroll
          for (int k = 0; k < KPP / 2; ++k) {
            const int kk = p * (KPP / 2) + k;          // k32 step of the tile
            h8 b[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) b[ct][pl] = __builtin_bit_cast(h8, slot[((2 * k + ct) * 2 + pl) * 64 + lane]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const h8 xh = ah[2 * kk + rt], xl = al[2 * kk + rt];      // (the wave's A image: [k32 step][row tile])
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) {
                f32x4& d = acc4[2 * rt + ct];
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, b[ct][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, b[ct][1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, b[ct][0], d, 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * s + q] = acc4[s][q];
        }
      }
      // the feature contraction of the tile: out[c][i] += F[u c][row of register i] * acc[i]
      const int u = t % 10;
      if constexpr (SHAPE == 32) {
        const float* frow = feat + u * C * 36 + 4 * (lane >> 5);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 36 + 8 * q4);
#pragma unroll
            for (int q = 0; q < 4; ++q) res[c][4 * q4 + q] += f[q] * acc[4 * q4 + q];
          }
      } else {
        const float* frow = feat + u * C * 36 + 4 * (lane >> 4);       // rows 4 (l / 16) .. + 3 of a 16-row tile
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 36 + 16 * rt);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int q = 0; q < 4; ++q) res[c][4 * (2 * rt + ct) + q] += f[q] * acc[4 * (2 * rt + ct) + q];
          }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) sink += res[c][i];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  out[(size_t)blockIdx.x * NT + tid] = sink;
  if (tid == 0) { clk[blockIdx.x] = t1 - t0; rclk[blockIdx.x] = r1 - r0; }
}

template <int SHAPE, int C>
static void run(const char* name, const f32x4* w, const h8* a, float* out, unsigned long long* clk, unsigned long long* rclk, int ntiles, int nseg, int wgs) {
  const size_t ring_b = (size_t)3 * (2 * (NS / NP) * 64) * 16, ldsb = ring_b + (size_t)NW * FEAT_FLOATS * 4;
  const size_t lds_use = ldsb < 70 * 1024 ? 70 * 1024 : ldsb;     // exactly two workgroups per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<SHAPE, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((tile_kernel<SHAPE, C>), dim3(wgs), dim3(NT), lds_use, 0, w, a, out, ntiles, nseg, clk, rclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  CK(hipGetLastError());
  std::vector<unsigned long long> c(wgs), r(wgs);
  CK(hipMemcpy(c.data(), clk, 8 * wgs, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), rclk, 8 * wgs, hipMemcpyDeviceToHost));
  double mean = 0, ghz = 0; int n = 0;
  for (int i = 0; i < wgs; ++i) { mean += (double)c[i] / wgs; if (r[i] > 0) { ghz += (double)c[i] / (double)r[i] * 0.1; ++n; } }
  const double flop = 2.0 * 32 * 32 * 192 * ntiles * NW * wgs;          // product FLOPs (each costs 3 fp16 MFMA FLOPs)
  printf("%-64s wgs %5d: %.3f ms, %6.0f k ticks per workgroup, in-kernel clock %.2f GHz, %6.1f TFLOP/s fp32-equivalent = %.2f PFLOP/s of fp16 MFMA issue\n", name, wgs, best,
         mean / 1e3, n ? ghz / n : 0.0, flop / best / 1e9, 3.0 * flop / best / 1e12);
}

int main() {
  const int ntiles = 60, nseg = 6, wgs_max = 256 * 8;
  std::vector<_Float16> W((size_t)ntiles * TILE_Q * 8), A((size_t)wgs_max * NW * 2 * NS * 64 * 8);
  srand(2);
  for (auto& v : W) v = (_Float16)((rand() / (float)RAND_MAX) * 0.4f - 0.2f);
  for (auto& v : A) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f);
  f32x4* w; h8* a; float* out; unsigned long long *clk, *rclk;
  CK(hipMalloc(&w, W.size() * 2)); CK(hipMalloc(&a, A.size() * 2)); CK(hipMalloc(&out, (size_t)wgs_max * NT * 4)); CK(hipMalloc(&clk, 8 * wgs_max)); CK(hipMalloc(&rclk, 8 * wgs_max));
  CK(hipMemcpy(w, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  // warm the chip up (the clock it holds depends on what ran before): a few seconds of the shipped form
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((tile_kernel<32, 1>), dim3(2048), dim3(NT), 70 * 1024, 0, w, a, out, ntiles, nseg, clk, rclk);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 1>("A: v_mfma_f32_32x32x16_f16 (shipped form), C = 1", w, a, out, clk, rclk, ntiles, nseg, 2048);
    run<16, 1>("S: v_mfma_f32_16x16x32_f16, C = 1", w, a, out, clk, rclk, ntiles, nseg, 2048);
    run<32, 3>("A: v_mfma_f32_32x32x16_f16 (shipped form), C = 3", w, a, out, clk, rclk, ntiles, nseg, 2048);
    run<16, 3>("S: v_mfma_f32_16x16x32_f16, C = 3", w, a, out, clk, rclk, ntiles, nseg, 2048);
  }
  return 0;
}
