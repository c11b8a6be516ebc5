// Diagnostic micro-benchmark (not part of the product): the stream-tile loop of ddp_conv_rows at ONE 512-register wave per SIMD.
// Question ("next" 1): the shipped kernel runs two 256-register waves per SIMD, each with the A operand of 32
// edges in registers; a wave issues one MFMA per ~100 cycles.  What does ONE wave per SIMD reach when it owns RT = 2 row tiles (64
// edges: every B fragment read from LDS feeds two independent accumulation chains, half the LDS reads and barriers per product)?
//   hipcc --offload-arch=gfx950 -O3 -o stream_wide stream_wide.hip && ./stream_wide
// Both forms: unified fp16 hi/lo planes (3 MFMAs per k-step and row tile on one accumulator), weight tiles once per workgroup through a
// three-slot LDS ring filled by buffer loads to LDS, bare barriers, C-component feature contraction per tile from LDS.
//   form A (shipped): RT = 1, 4 waves per workgroup, two workgroups per CU, ring of third tiles (3 barriers per tile)
//   form B:           RT = 2, 4 waves per workgroup, one workgroup per CU, ring of whole tiles (1 barrier per tile)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NS = 12, NF = 2 * NS, TILE_Q = NF * 64, NW = 4, NT = 256, FEAT_FLOATS = 60 * 36;

__device__ __forceinline__ f32x16 splat(float v) { f32x16 r; for (int i = 0; i < 16; ++i) r[i] = v; return r; }
template <int C>
__device__ __forceinline__ void epilogue(const f32x16& acc, const float* feat, int u, int hh, f32x16* out) {
  const float* frow = feat + u * C * 36 + 4 * hh;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 36 + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] += f[q] * acc[4 * q4 + q];
    }
}

// NP pieces per tile (a ring slot holds one piece), RT row tiles per wave, C components, WPC workgroups per CU (launch bound)
template <int RT, int NP, int C, int WPC>
__global__ __launch_bounds__(NT, WPC) void wide_kernel(const f32x4* __restrict__ w, const h8* __restrict__ a, float* __restrict__ out, int ntiles, int nseg,
                                                        unsigned long long* clk) {
  constexpr int KPP = NS / NP, PIECE_Q = 2 * KPP * 64, FPW = 2 * KPP / NW;     // fragments a wave copies per piece
  static_assert(NS % NP == 0 && (2 * KPP) % NW == 0, "pieces");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);                 // 3 slots of PIECE_Q quads
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5;
  float* feat = lds + 3 * PIECE_Q * 4 + wave * FEAT_FLOATS;
  for (int i = lane; i < FEAT_FLOATS; i += 64) feat[i] = 1e-3f * (float)((i * 7 + wave) & 31);
  h8 ah[RT][NS], al[RT][NS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const h8* ap = a + ((size_t)((blockIdx.x * NW + wave) * RT + rt) * 2 * NS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) { ah[rt][ks] = ap[(2 * ks) * 64]; al[rt][ks] = ap[(2 * ks + 1) * 64]; }
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(w), 0, ntiles * TILE_Q * 16, 0x00020000);
  const int npieces = ntiles * NP;
  auto request = [&](int j, int slot) {
    const int off = min(j, npieces - 1) * (PIECE_Q * 16);
#pragma unroll
    for (int f = 0; f < FPW; ++f)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ring + slot * PIECE_Q + (wave + NW * f) * 64), 16, ((wave + NW * f) * 64 + lane) * 16, off, 0, 0);
  };
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  request(0, 0);
  request(1, 1);
  float sink = 0.f;
  const int tps = ntiles / nseg;
  int j = 0;     // piece counter
  for (int sg = 0; sg < nseg; ++sg) {
    f32x16 res[RT][C];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < C; ++c) res[rt][c] = splat(0.f);
    for (int t = 0; t < tps; ++t) {
      f32x16 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = splat(0.25f);
#pragma unroll
      for (int p = 0; p < NP; ++p, ++j) {
        if constexpr (FPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (FPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        request(j + 2, (j + 2) % 3);
        const f32x4* slot = ring + (j % 3) * PIECE_Q;
        f32x4 b0 = slot[lane], b1 = slot[64 + lane];
#pragma unroll
        for (int k = 0; k < KPP; ++k) {
          const h8 bh = __builtin_bit_cast(h8, b0), bl = __builtin_bit_cast(h8, b1);
          if (k + 1 < KPP) { b0 = slot[(2 * k + 2) * 64 + lane]; b1 = slot[(2 * k + 3) * 64 + lane]; }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt][p * KPP + k], bh, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt][p * KPP + k], bl, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rt][p * KPP + k], bh, acc[rt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epilogue<C>(acc[rt], feat, (t + rt) % 10, hh, res[rt]);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) sink += res[rt][c][i];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[(size_t)blockIdx.x * NT + tid] = sink;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int RT, int NP, int C, int WPC>
static void run(const char* name, const f32x4* w, const h8* a, float* out, unsigned long long* clk, int ntiles, int nseg, int wgs) {
  const size_t ring_b = (size_t)3 * (2 * (NS / NP) * 64) * 16, ldsb = ring_b + (size_t)NW * FEAT_FLOATS * 4;
  // (form A: pad the workgroup's LDS so that exactly WPC workgroups share a CU)
  const size_t lds_use = (WPC == 2) ? (ldsb < 70 * 1024 ? 70 * 1024 : ldsb) : (ldsb < 100 * 1024 ? 100 * 1024 : ldsb);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_kernel<RT, NP, C, WPC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((wide_kernel<RT, NP, C, WPC>), dim3(wgs), dim3(NT), lds_use, 0, w, a, out, ntiles, nseg, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  CK(hipGetLastError());
  std::vector<unsigned long long> c(wgs);
  CK(hipMemcpy(c.data(), clk, 8 * wgs, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : c) mean += (double)v / wgs;
  const double flop = 2.0 * 32 * 32 * 192 * ntiles * RT * NW * wgs;          // product FLOPs (each costs 3 fp16 MFMA FLOPs)
  const double edges = 32.0 * RT * NW * wgs;
  printf("%-60s wgs %5d: %.3f ms, %6.0f k ticks per workgroup, %6.1f ticks per MFMA and wave, %6.1f TFLOP/s fp32-equivalent, %.2f us per 1000 edges\n", name, wgs,
         best, mean / 1e3, mean / (36.0 * ntiles * RT), flop / best / 1e9, best * 1e3 / (edges / 1000.0));
}

int main() {
  const int ntiles = 60, nseg = 6, wgs_max = 256 * 8;
  std::vector<_Float16> W((size_t)ntiles * TILE_Q * 8), A((size_t)wgs_max * NW * 2 * 2 * NS * 64 * 8);
  srand(2);
  for (auto& v : W) v = (_Float16)((rand() / (float)RAND_MAX) * 0.4f - 0.2f);
  for (auto& v : A) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f);
  f32x4* w; h8* a; float* out; unsigned long long* clk;
  CK(hipMalloc(&w, W.size() * 2)); CK(hipMalloc(&a, A.size() * 2)); CK(hipMalloc(&out, (size_t)wgs_max * NT * 4)); CK(hipMalloc(&clk, 8 * wgs_max));
  CK(hipMemcpy(w, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  for (int mult : {1, 4}) {
    run<1, 3, 1, 2>("A: 32 edges per wave, 2 workgroups per CU, third-tile ring, C = 1", w, a, out, clk, ntiles, nseg, 512 * mult);
    run<1, 3, 3, 2>("A: ... C = 3", w, a, out, clk, ntiles, nseg, 512 * mult);
    run<2, 3, 1, 1>("B: 64 edges per wave, 1 workgroup per CU, third-tile ring, C = 1", w, a, out, clk, ntiles, nseg, 256 * mult);
    run<2, 1, 1, 1>("B: ... whole-tile ring (1 barrier per tile), C = 1", w, a, out, clk, ntiles, nseg, 256 * mult);
    run<2, 1, 3, 1>("B: ... whole-tile ring, C = 3", w, a, out, clk, ntiles, nseg, 256 * mult);
  }
  return 0;
}
