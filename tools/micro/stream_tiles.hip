// Diagnostic micro-benchmark (not part of the product): the tile loop of a "row-stationary" conv kernel.
// Question: can 8 waves of ONE workgroup per CU (two per SIMD), each holding the A operand (the fp16 hi/lo planes of h for its own
// 32 edges: 96 registers) in registers, run the fp16 hi/lo split products near the matrix-pipe rate when the weight tiles are
// streamed ONCE per workgroup through an LDS ring (register-staged: every wave loads 3 of a tile's 24 one-KiB fragments, one
// barrier per tile) and every wave reads its B operands from LDS?  Per 256 edges the weights then leave L2 once (today: once per
// 32 edges, ~2 MB per workgroup = the L2 -> CU rate the chip sustains).
//   hipcc --offload-arch=gfx950 -O3 -o stream_tiles stream_tiles.hip && ./stream_tiles
// Modes:  G = private "G tiles" per wave (B operand straight from global memory through a register ring, per-wave addresses),
//         EPI = feature contraction per tile (C components, features from a wave-private LDS area)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NS = 12, NF = 2 * NS, TILE_Q = NF * 64;   // f32x4 quads per tile (24 fragments of 1 KiB)
constexpr int NW = 8, NT = 512;
constexpr int FPW = NF / NW;                             // fragments a wave stages per tile
constexpr int FEAT_FLOATS = 60 * 36;                     // per-wave feature area

__device__ __forceinline__ f32x16 splat(float v) { f32x16 r; for (int i = 0; i < 16; ++i) r[i] = v; return r; }

template <int C>
__device__ __forceinline__ void epilogue(const f32x16& am, const f32x16& ac, const float* feat, int u, int hh, f32x16* out) {
  const float* frow = feat + u * C * 36 + 4 * hh;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    float tq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tq[q] = am[4 * q4 + q] + ac[4 * q4 + q] * (1.f / 2048.f);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 36 + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] += f[q] * tq[q];
    }
  }
}

// one tile product from B fragments in LDS (24 KiB, fragment q = 2 ks + plane at q * 1 KiB + lane * 16)
__device__ __forceinline__ void tile_from_lds(const f32x4* slot, const h8 (&ah)[NS], const h8 (&al)[NS], int lane, f32x16& am, f32x16& ac) {
  f32x4 b0 = slot[lane], b1 = slot[64 + lane];
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) {
    const h8 bh = __builtin_bit_cast(h8, b0), bl = __builtin_bit_cast(h8, b1);
    if (ks + 1 < NS) {
      b0 = slot[(2 * ks + 2) * 64 + lane];
      b1 = slot[(2 * ks + 3) * 64 + lane];
    }
    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, am, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, ac, 0, 0, 0);
  }
}

template <int C, int NG, int GRING>
__global__ __launch_bounds__(NT, 1) void stream_kernel(const f32x4* __restrict__ w, const f32x4* __restrict__ g, const h8* __restrict__ a,
                                                        float* __restrict__ out, int ntiles, int nseg, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);                 // 2 slots of TILE_Q quads
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5;
  float* feat = lds + 2 * TILE_Q * 4 + wave * FEAT_FLOATS;
  for (int i = lane; i < FEAT_FLOATS; i += 64) feat[i] = 1e-3f * (float)((i * 7 + wave) & 31);
  h8 ah[NS], al[NS];
  const h8* ap = a + ((size_t)(blockIdx.x * NW + wave) * 2 * NS) * 64 + lane;
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) { ah[ks] = ap[(2 * ks) * 64]; al[ks] = ap[(2 * ks + 1) * 64]; }
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  // stage tile 0 into slot 0, request tile 1
  f32x4 st[FPW];
#pragma unroll
  for (int f = 0; f < FPW; ++f) st[f] = w[(size_t)(wave * FPW + f) * 64 + lane];
#pragma unroll
  for (int f = 0; f < FPW; ++f) ring[(wave * FPW + f) * 64 + lane] = st[f];
#pragma unroll
  for (int f = 0; f < FPW; ++f) st[f] = w[(size_t)min(1, ntiles - 1) * TILE_Q + (wave * FPW + f) * 64 + lane];
  f32x16 res[3];
  float sink = 0.f;
  const int tps = ntiles / nseg;     // stream tiles per segment
  int t = 0;
  for (int sg = 0; sg < nseg; ++sg) {
#pragma unroll
    for (int c = 0; c < C; ++c) res[c] = splat(0.f);
    if constexpr (NG > 0) {
      // private G tiles of this segment: B fragments straight from global memory (per-wave rows), register ring of GRING fragments
      const f32x4* gp = g + ((size_t)((blockIdx.x * NW + wave) * nseg + sg) * NG) * TILE_Q + lane;
      f32x4 gr[GRING];
#pragma unroll
      for (int k = 0; k < GRING; ++k) gr[k] = gp[k * 64];
      for (int j = 0; j < NG; ++j) {
        const f32x4* gn = gp + ((j + 1 < NG) ? TILE_Q : 0);
        f32x16 am = splat(0.f), ac = splat(0.f);
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          const h8 bh = __builtin_bit_cast(h8, gr[(2 * ks) % GRING]), bl = __builtin_bit_cast(h8, gr[(2 * ks + 1) % GRING]);
          const int q0 = 2 * ks + GRING, q1 = q0 + 1;
          gr[(2 * ks) % GRING] = (q0 < NF) ? gp[q0 * 64] : gn[(q0 - NF) * 64];
          gr[(2 * ks + 1) % GRING] = (q1 < NF) ? gp[q1 * 64] : gn[(q1 - NF) * 64];
          __builtin_amdgcn_sched_barrier(0);
          am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, am, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, ac, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, ac, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        gp = gn;
        epilogue<C>(am, ac, feat, j % 10, hh, res);
      }
    }
    for (int j = 0; j < tps; ++j, ++t) {
      __syncthreads();     // tile t is in slot t & 1 (every wave's part), everyone is done with tile t - 1
      // park tile t + 1 (requested a tile ago) into the other slot, request tile t + 2
      f32x4* nslot = ring + ((t + 1) & 1) * TILE_Q;
#pragma unroll
      for (int f = 0; f < FPW; ++f) nslot[(wave * FPW + f) * 64 + lane] = st[f];
      const f32x4* wn = w + (size_t)min(t + 2, ntiles - 1) * TILE_Q;
#pragma unroll
      for (int f = 0; f < FPW; ++f) st[f] = wn[(wave * FPW + f) * 64 + lane];
      f32x16 am = splat(0.25f), ac = splat(0.f);
      tile_from_lds(ring + (t & 1) * TILE_Q, ah, al, lane, am, ac);
      epilogue<C>(am, ac, feat, j % 10, hh, res);
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) sink += res[c][i];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[(size_t)blockIdx.x * NT + tid] = sink + st[0][0];
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int C, int NG, int GRING>
static void run(const char* name, const f32x4* w, const f32x4* g, const h8* a, float* out, unsigned long long* clk, int ntiles, int nseg, int wgs) {
  const size_t ldsb = (size_t)(2 * TILE_Q * 4 + NW * FEAT_FLOATS) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<C, NG, GRING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<C, NG, GRING>), dim3(wgs), dim3(NT), ldsb, 0, w, g, a, out, ntiles, nseg, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  CK(hipGetLastError());
  std::vector<unsigned long long> c(wgs);
  CK(hipMemcpy(c.data(), clk, 8 * wgs, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : c) mean += (double)v / wgs;
  const double tiles_per_wave = ntiles + (double)NG * nseg;
  const double pipe = tiles_per_wave * 36 * 32 * 2;          // matrix-pipe cycles per SIMD (two waves) per workgroup
  const double flop = 2.0 * 32 * 32 * 192 * tiles_per_wave * NW * wgs;
  printf("%-34s wgs %5d: %.3f ms, %.0f k ticks per workgroup (pipe %.0f k = %.0f %%), %.1f TFLOP/s fp32-equivalent, %.2f MB L2->CU per 32 edges\n", name, wgs, best, mean / 1e3,
         pipe / 1e3, 100.0 * pipe / mean, flop / best / 1e9, (ntiles * 24.576e-3 / NW + NG * nseg * 24.576e-3));
}

int main() {
  const int ntiles = 60, nseg = 6, wgs_max = 256 * 8;
  std::vector<_Float16> W((size_t)ntiles * TILE_Q * 8), A((size_t)wgs_max * NW * 2 * NS * 64 * 8);
  srand(2);
  for (auto& v : W) v = (_Float16)((rand() / (float)RAND_MAX) * 0.4f - 0.2f);
  for (auto& v : A) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f);
  const size_t gq = (size_t)wgs_max * NW * nseg * 3 * TILE_Q;     // up to 3 private G tiles per wave and segment (1.7 GB at 2048 workgroups)
  f32x4 *w, *g; h8* a; float* out; unsigned long long* clk;
  CK(hipMalloc(&w, W.size() * 2)); CK(hipMalloc(&a, A.size() * 2)); CK(hipMalloc(&g, gq * 16)); CK(hipMalloc(&out, (size_t)wgs_max * NT * 4)); CK(hipMalloc(&clk, 8 * wgs_max));
  CK(hipMemcpy(w, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(g, 0x11, gq * 16));
  for (int wgs : {256, 2048}) {
    run<1, 0, 4>("stream only, C = 1", w, g, a, out, clk, ntiles, nseg, wgs);
    run<3, 0, 4>("stream only, C = 3", w, g, a, out, clk, ntiles, nseg, wgs);
    run<1, 2, 4>("stream + 2 G tiles/seg, ring 4", w, g, a, out, clk, ntiles, nseg, wgs);
    run<1, 2, 8>("stream + 2 G tiles/seg, ring 8", w, g, a, out, clk, ntiles, nseg, wgs);
    run<3, 2, 8>("stream + 2 G tiles/seg, ring 8, C=3", w, g, a, out, clk, ntiles, nseg, wgs);
    run<1, 3, 8>("stream + 3 G tiles/seg, ring 8", w, g, a, out, clk, ntiles, nseg, wgs);
  }
  return 0;
}
