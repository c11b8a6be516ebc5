// ddp_gemm.hip - stage A of the source-node factorisation (include/ddp_hip.h, ddp_stage_a):
//   out[b][j * ldo + n] = sum_u x[j, off[b] + u] * w[b][u, n]      b < nbatch, j < nrows, n < ncols, K = n_in (= ns <= 64)
// i.e. G = x_scalar @ Wg per conv and G slot, [n_src, ncols] with ncols = hg * g_cols (12600 at
// ns = 60): 120 FLOPs and 4 bytes written per output element, so the product is bound by the HBM write of G.
//
// Three forms, chosen by ddp_stage_a() from the shape:
//  * ddp_stage_a_mfma_kernel (what the score model's products run: ncols >= 512, K in {60, 64, 32, 24, 16}): weight-stationary
//    on v_mfma_f32_32x32x2_f32, exact fp32; details above the kernel;
//  * ddp_stage_a_kernel (narrow or odd shapes): weight-stationary on the VECTOR ALU - a lane owns two (four) output columns and
//    keeps their K weights in registers, the x row of the current node is wave-uniform and arrives through the SCALAR cache
//    (s_load_dwordx16) as an SGPR pair of v_pk_fma_f32: no LDS, no per-lane x traffic, one 8 / 16-byte store per lane and row;
//    measured at ~1/3 of its nominal rate on the wide products, hence the MFMA form there;
//  * ddp_stage_a_x3_kernel (an option that stays off, model.stage_a_bf16x3): the MFMA form on bf16 with both operands split in
//    three terms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DDP_GEMM_THREADS 256
#define DDP_GEMM_ROWS 128   // rows per workgroup (the weight columns are re-read from L2 once per 128 rows)

struct GemmOffs {
  int off[DDP_MAX_GEMM_BATCH];
};

// KT > 0: K is the compile-time constant KT (fully unrolled, no guards); KT == 0: any even K <= 64 (guarded pairs).
// CPL = output columns per lane: 4 (K = 60: 240 weight registers, two waves per SIMD, 16-byte stores; needs ncols % 4
// == 0) halves the scalar-cache traffic per output byte against 2, which is what bounds the 2-column form.
// x / w / out are direct __restrict__ kernel arguments: only then does the compiler treat the x row as invariant and
// fetch it with scalar loads (through a pointer read from a struct it falls back to 60 per-lane loads per row: 5x slower).
template <int KT, int CPL>
__global__ __launch_bounds__(DDP_GEMM_THREADS, 2) void ddp_stage_a_kernel(const float* __restrict__ x, int ldx, int nrows,
                                                                           const int32_t* __restrict__ rows,
                                                                           const int32_t* __restrict__ nrows_dev, int out_rows,
                                                                           const GemmOffs offs, const float* __restrict__ w,
                                                                           int k, int ncols, float* __restrict__ out, int ldo) {
  constexpr int KP = (KT > 0) ? KT / 2 : 32;                  // k pairs held in registers
  const int K = (KT > 0) ? KT : k;
  const int z = (int)blockIdx.z;
  if (nrows_dev) nrows = min(nrows, *nrows_dev);              // device-side length of the row list (nrows = its capacity)
  const int r0 = (int)blockIdx.y * DDP_GEMM_ROWS;
  if (r0 >= nrows) return;
  const int r1 = min(nrows, r0 + DDP_GEMM_ROWS);
  const int col = ((int)blockIdx.x * DDP_GEMM_THREADS + (int)threadIdx.x) * CPL;
  bool act[CPL];
  int cc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    act[q] = col + q < ncols;
    cc[q] = act[q] ? col + q : 0;
  }
  const float* __restrict__ W = w + (size_t)z * K * ncols;
  f32x2 wr[CPL][KP];
#pragma unroll
  for (int u = 0; u < KP; ++u) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      if (KT > 0 || 2 * u < K)
        wr[q][u] = f32x2{W[(size_t)(2 * u) * ncols + cc[q]], W[(size_t)(2 * u + 1) * ncols + cc[q]]};
      else
        wr[q][u] = f32x2{0.f, 0.f};
    }
  }
  const bool vec = act[CPL - 1] && ((ldo & (CPL - 1)) == 0);     // aligned CPL-wide store
  for (int j = r0; j < r1; ++j) {
    const int row = rows ? rows[j] : j;                         // (wave-uniform: x row j arrives through scalar loads)
    float* __restrict__ o = out + ((size_t)z * out_rows + row) * ldo + col;
    const float* __restrict__ xr = x + (size_t)row * ldx + offs.off[z];
    f32x2 acc[CPL];                                             // even / odd k partial sums
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = f32x2{0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      if (KT > 0 || 2 * u < K) {
        const f32x2 xv = {xr[2 * u], xr[2 * u + 1]};
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = __builtin_elementwise_fma(xv, wr[q][u], acc[q]);
      }
    }
    if (vec) {
      if constexpr (CPL == 4)
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], acc[2][0] + acc[2][1], acc[3][0] + acc[3][1]};
      else
        *reinterpret_cast<f32x2*>(o) = f32x2{acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]};
    } else {
#pragma unroll
      for (int q = 0; q < CPL; ++q)
        if (act[q]) o[q] = acc[q][0] + acc[q][1];
    }
  }
}

// ---- MFMA form (used for wide outputs): weight-stationary on v_mfma_f32_32x32x2_f32.
// A wave owns CT = 4 column tiles of 32 and keeps their K x 32 weights as B operands in registers (K/2 VGPRs per tile:
// lane (r, hh) holds w[k = hh*K/2 + s][col r], the k order of the K/2 MFMA steps is (s, K/2 + s)); it streams 32-row tiles
// of x as the A operand (lane (r, hh): x[row r][off + hh*K/2 + s], K/2 contiguous floats, next tile requested one tile
// ahead) and stores each accumulator register as two 128-byte row segments.  4 independent accumulators per wave keep
// the matrix pipe issuing; per 32 x 128 output block K/2 * 4 MFMAs = 7680 cycles for 16 KiB at K = 60, i.e. the same
// 8.5 B/clk/CU bound as above, but the VALU form stalls at ~1/3 of its nominal rate (1.5 TB/s measured, rocBLAS 2.2-2.4).
// Measured on the 44440 x 60 x 12608 product (4.5 GB): MFMA side alone 1.0 ms, stores alone 0.88 ms, together 1.53 ms.
// Tried without gain: 16-byte stores through an LDS transpose, non-temporal stores, 2 / 3 / 5 / 6 column tiles per wave,
// 512 rows per workgroup, and a producer / consumer split (4 MFMA waves handing blocks through LDS to 4 store waves).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef DDP_GEMM_MROWS
#define DDP_GEMM_MROWS 512
#endif //  // rows per workgroup of the MFMA form

#ifndef DDP_SA_CT
#define DDP_SA_CT 4      // column tiles (of 32) per wave
#define DDP_SA_WPE 2     // waves per SIMD the register budget is set for
#endif
template <int KT>
__global__ __launch_bounds__(DDP_GEMM_THREADS, DDP_SA_WPE) void ddp_stage_a_mfma_kernel(const float* __restrict__ x, int ldx, int nrows,
                                                                                const int32_t* __restrict__ rows,
                                                                                const int32_t* __restrict__ nrows_dev, int out_rows,
                                                                                int mrows, const GemmOffs offs, const float* __restrict__ w,
                                                                                int ncols, float* __restrict__ out, int ldo) {
  // mrows: rows per workgroup (a multiple of 32).  DDP_GEMM_MROWS for large products - the weight registers of a workgroup are
  // loaded once per 512 rows - and 128 for small ones (a strong-scaling shard has 185 ligand rows: more, shorter workgroups)
  if (nrows_dev) nrows = min(nrows, *nrows_dev);              // device-side length of the row list (nrows = its capacity)
  if ((int)blockIdx.y * mrows >= nrows) return;
  constexpr int KH = KT / 2, CT = DDP_SA_CT;
  constexpr bool LAG = (4 + 3 * (KH / 4) <= KH - 1);   // room for the lagged stores between a block's MFMAs
  constexpr int XS = KT + 1;                                    // odd LDS row stride: conflict-free ds_read_b32 down a column
  constexpr int NV = (32 * KT / 4 + DDP_GEMM_THREADS - 1) / DDP_GEMM_THREADS;   // 16-byte pieces of an x tile per thread
  constexpr int TS = 36;                                        // LDS row stride of the per-wave store tile
  __shared__ float xt[2][32 * XS];
  __shared__ __attribute__((aligned(16))) float st[4][2][32 * TS];
  const bool wide = ((ncols & 3) == 0) && ((ldo & 3) == 0) && ((reinterpret_cast<size_t>(out) & 15) == 0);
  const int z = (int)blockIdx.z, tid = (int)threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int col0 = ((int)blockIdx.x * 4 + wave) * (32 * CT);
  const float* __restrict__ W = w + (size_t)z * KT * ncols;
  float wr[CT][KH];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = min(col0 + 32 * t + r, ncols - 1);
#pragma unroll
    for (int s2 = 0; s2 < KH; ++s2) wr[t][s2] = W[(size_t)(hh * KH + s2) * ncols + c];
  }
  const float* __restrict__ xb = x + offs.off[z];
  // Row tiles of this workgroup: blockIdx.y, + gridDim.y, ...  Dense products get one tile per workgroup; a ROW LIST (whose grid
  // is sized for the list's capacity, ~10 x its usual length) is launched with a bounded gridDim.y and walks its tiles, keeping
  // the weight registers: far fewer workgroups that only find out they have nothing to do
  for (int R0 = (int)blockIdx.y * mrows; R0 < nrows; R0 += (int)gridDim.y * mrows) {
  const int R1 = min(nrows, R0 + mrows);
  const bool al4 = ((ldx | offs.off[z]) & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;

  // The 32 x K tile of x rows is fetched ONCE per workgroup with coalesced 16-byte loads (each wave fetching its own A
  // operand straight from memory is 64 scattered 4-byte pieces per load instruction, 4x redundant: the L1 tag rate then
  // bounds the kernel at ~40 % of the matrix pipe), parked in LDS (double buffered) and read per lane as the A operand.
  f32x4 xv[NV];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;                 // piece i = (row i / (KT/4), quad i % (KT/4))
      const int rr = min(i / (KT / 4), 31), q = i % (KT / 4);
      const int ri = min(row0 + rr, nrows - 1);
      const float* __restrict__ p = xb + (size_t)(rows ? rows[ri] : ri) * ldx + 4 * q;
      if (al4)
        xv[v] = *reinterpret_cast<const f32x4*>(p);
      else
        xv[v] = f32x4{p[0], p[1], p[2], p[3]};
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;
      if (i < 32 * (KT / 4)) {
        const int rr = i / (KT / 4), q = i % (KT / 4);
        float* d = &xt[buf][rr * XS + 4 * q];
        d[0] = xv[v][0]; d[1] = xv[v][1]; d[2] = xv[v][2]; d[3] = xv[v][3];
      }
    }
  };
  // pending block of this wave (parked in LDS, not yet written out)
  float* pend_ob = nullptr;
  int pend_c0 = 0, pend_rows = 0, pend_buf = 0, pbuf = 0;
  int pend_ri[4] = {0, 0, 0, 0};   // out rows of this lane's four 16-byte pieces of the pending block
  f32x4 dv = {0.f, 0.f, 0.f, 0.f};
  auto drain_read = [&](int p) {
    if (pend_ob) dv = *reinterpret_cast<const f32x4*>(&st[wave][pend_buf][(8 * p + (lane >> 3)) * TS + 4 * (lane & 7)]);
  };
  auto drain_store = [&](int p) {
    if (pend_ob) {
      const int rr = 8 * p + (lane >> 3), c = pend_c0 + 4 * (lane & 7);
      if (rr < pend_rows && c < ncols) *reinterpret_cast<f32x4*>(&pend_ob[(size_t)pend_ri[p] * ldo + c]) = dv;
    }
  };
  fetch(R0);
  park(0);
  __syncthreads();
  int buf = 0;
  for (int row0 = R0; row0 < R1; row0 += 32, buf ^= 1) {
#if defined(DDP_SA_ABL) && (DDP_SA_ABL == 5 || DDP_SA_ABL == 6)   // timing only: no loads inside the row-tile loop (stores alone on the vmcnt queue)
    const bool more = false;
#else
    const bool more = row0 + 32 < R1;
#endif
    if (more) fetch(row0 + 32);                                 // in flight during this tile's MFMAs
    float a[KH];
    {
      const float* ar = &xt[buf][r * XS + hh * KH];
#pragma unroll
      for (int s2 = 0; s2 < KH; ++s2) a[s2] = ar[s2];
    }
    // one column tile at a time: its 32 x 32 block is stored while the next tile's MFMAs run (a single accumulator
    // chain already issues at the full rate: issue interval = dependent latency = 64 cycles); C/D layout: register i of
    // lane (r, hh) = out[row0 + (i&3) + 8*(i>>2) + 4*hh][col r], i.e. every store instruction writes two 128-byte row pieces
    float* __restrict__ ob = out + (size_t)z * out_rows * ldo;   // the batch slice; row indices below are absolute
    int blk_ri[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ri = min(row0 + 8 * p + (lane >> 3), R1 - 1);
      blk_ri[p] = rows ? rows[ri] : ri;
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
#if defined(DDP_SA_ABL) && DDP_SA_ABL == 3
      for (int s2 = 0; s2 < 1; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], wr[t][s2], acc, 0, 0, 0);
#else
      for (int s2 = 0; s2 < KH; ++s2) {
#if defined(DDP_SA_ABL) && (DDP_SA_ABL == 4 || DDP_SA_ABL == 6)   // timing only: 1/5 of the matrix work, the store schedule unchanged
        if (s2 % 5 == 0)
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], wr[t][s2], acc, 0, 0, 0);
        if (wide && LAG) {   // quarter p of the pending block: LDS read at MFMA 2 + p*KH/4, store two MFMAs later
          if ((s2 - 2) % (KH / 4) == 0 && (s2 - 2) / (KH / 4) < 4 && s2 >= 2) drain_read((s2 - 2) / (KH / 4));
          if ((s2 - 4) % (KH / 4) == 0 && (s2 - 4) / (KH / 4) < 4 && s2 >= 4) drain_store((s2 - 4) / (KH / 4));
        }
      }
#endif
      if (wide) {
        // park the block in this wave's LDS tile (double buffered); its four 16-byte-per-lane stores (8 rows x 128
        // bytes each) are issued one at a time between the MFMAs of the NEXT block (drain() calls in the loop above), so
        // the writes leave the CU as a steady stream instead of a 16-KiB burst per workgroup after every block
        float* tl = st[wave][pbuf];
#pragma unroll
        for (int i = 0; i < 16; ++i) tl[((i & 3) + 8 * (i >> 2) + 4 * hh) * TS + r] = acc[i];
        pend_ob = ob;
#pragma unroll
        for (int p = 0; p < 4; ++p) pend_ri[p] = blk_ri[p];
        pend_c0 = col0 + 32 * t;
        pend_rows = R1 - row0;
        pend_buf = pbuf;
        pbuf ^= 1;
        if (!LAG) {   // (short K: no room between the MFMAs, write the block out right away)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            drain_read(p);
            drain_store(p);
          }
          pend_ob = nullptr;
        }
      } else {
        const int c = col0 + 32 * t + r;
        if (c < ncols) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (row0 + rr < R1) ob[(size_t)(rows ? rows[row0 + rr] : row0 + rr) * ldo + c] = acc[i];
          }
        }
      }
    }
    if (more) park(buf ^ 1);
    __syncthreads();
  }
  if (wide && LAG) {   // the last block
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      drain_read(p);
      drain_store(p);
    }
  }
  __syncthreads();   // the next row tile reuses the LDS tiles
  }
}

// ---- bf16x3 form of the MFMA kernel: the same product on v_mfma_f32_32x32x16_bf16 (16 x the fp32 rate) with BOTH operands
// split into three bfloat16 terms, v = hi + mid + lo (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): 24 significant
// bits, every residual exact in fp32), and the six products whose weight is >= 2^-16 of the leading one accumulated in fp32:
//   x w ~ xh wh + (xh wm + xm wh) + (xm wm + xh wl + xl wh)            dropped: xm wl, xl wm, xl wl  (<= 2^-24 |x w| each)
// i.e. fp32-class accuracy (measured against an fp64 product: tests/test_gpu_parity.py::test_stage_a_bf16x3_error) at 6 MFMAs
// of 32 cycles per 16 k instead of 8 of 64: the matrix side of the product shrinks 2.7 x and the kernel is left with the HBM
// write of G.  Not bitwise the fp32 form (different rounding points), deterministic like it (a row's result does not depend
// on the tile it sits in).  Weights arrive pre-split from the host (packing.split_bf16x3): [batch][plane][k/16][k/8 % 2][col][8].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define DDP_SA3_CT 2   // column tiles per wave: 3 planes x 4 k-steps x 4 registers each = 96 weight registers (3 tiles: 256 + spills)

template <int KT>
__global__ __launch_bounds__(DDP_GEMM_THREADS, 2) void ddp_stage_a_x3_kernel(const float* __restrict__ x, int ldx, int nrows,
                                                                              const int32_t* __restrict__ rows,
                                                                              const int32_t* __restrict__ nrows_dev, int out_rows,
                                                                              int mrows, const GemmOffs offs,
                                                                              const __bf16* __restrict__ w3, int ncols,
                                                                              float* __restrict__ out, int ldo) {
  if (nrows_dev) nrows = min(nrows, *nrows_dev);
  if ((int)blockIdx.y * mrows >= nrows) return;
  constexpr int KP = (KT + 15) / 16 * 16, NS = KP / 16, CT = DDP_SA3_CT;
  constexpr int XS = KP + 8;                                   // bf16 per LDS row of an x plane (16-byte aligned rows)
  constexpr int NV = (32 * KT / 4 + DDP_GEMM_THREADS - 1) / DDP_GEMM_THREADS;
  constexpr int TS = 36;
  constexpr int NM = NS * 6;                                    // MFMAs per block
  __shared__ __attribute__((aligned(16))) __bf16 xt[2][3][32 * XS];
  __shared__ __attribute__((aligned(16))) float st[4][2][32 * TS];
  const int z = (int)blockIdx.z, tid = (int)threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int col0 = ((int)blockIdx.x * 4 + wave) * (32 * CT);
  bf16x8 wr[CT][3][NS];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = min(col0 + 32 * t + r, ncols - 1);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
        wr[t][p][s2] = *reinterpret_cast<const bf16x8*>(w3 + (((((size_t)z * 3 + p) * NS + s2) * 2 + hh) * ncols + c) * 8);
  }
  // the k padding [KT, KP) of both x buffers is zero for good
  if constexpr (KP > KT) {
    for (int i = tid; i < 2 * 3 * 32 * (KP - KT); i += DDP_GEMM_THREADS) {
      const int kk = i % (KP - KT), rr = (i / (KP - KT)) % 32, bp = i / ((KP - KT) * 32);
      xt[bp / 3][bp % 3][rr * XS + KT + kk] = (__bf16)0.f;
    }
  }
  const float* __restrict__ xb = x + offs.off[z];
  for (int R0 = (int)blockIdx.y * mrows; R0 < nrows; R0 += (int)gridDim.y * mrows) {
  const int R1 = min(nrows, R0 + mrows);
  const bool al4 = ((ldx | offs.off[z]) & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
  f32x4 xv[NV];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;
      const int rr = min(i / (KT / 4), 31), q = i % (KT / 4);
      const int ri = min(row0 + rr, nrows - 1);
      const float* __restrict__ p = xb + (size_t)(rows ? rows[ri] : ri) * ldx + 4 * q;
      if (al4)
        xv[v] = *reinterpret_cast<const f32x4*>(p);
      else
        xv[v] = f32x4{p[0], p[1], p[2], p[3]};
    }
  };
  auto park = [&](int buf) {   // split into the three bf16 planes on the way into LDS (once per workgroup, not per wave)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;
      if (i < 32 * (KT / 4)) {
        const int rr = i / (KT / 4), q = i % (KT / 4);
        bf16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = xv[v][e];
          h[e] = (__bf16)f;
          const float r1 = f - (float)h[e];
          m[e] = (__bf16)r1;
          l[e] = (__bf16)(r1 - (float)m[e]);
        }
        *reinterpret_cast<bf16x4*>(&xt[buf][0][rr * XS + 4 * q]) = h;
        *reinterpret_cast<bf16x4*>(&xt[buf][1][rr * XS + 4 * q]) = m;
        *reinterpret_cast<bf16x4*>(&xt[buf][2][rr * XS + 4 * q]) = l;
      }
    }
  };
  float* pend_ob = nullptr;
  int pend_c0 = 0, pend_rows = 0, pend_buf = 0, pbuf = 0;
  int pend_ri[4] = {0, 0, 0, 0};
  f32x4 dv = {0.f, 0.f, 0.f, 0.f};
  auto drain_read = [&](int p) {
    if (pend_ob) dv = *reinterpret_cast<const f32x4*>(&st[wave][pend_buf][(8 * p + (lane >> 3)) * TS + 4 * (lane & 7)]);
  };
  auto drain_store = [&](int p) {
    if (pend_ob) {
      const int rr = 8 * p + (lane >> 3), c = pend_c0 + 4 * (lane & 7);
      if (rr < pend_rows && c < ncols) *reinterpret_cast<f32x4*>(&pend_ob[(size_t)pend_ri[p] * ldo + c]) = dv;
    }
  };
  fetch(R0);
  park(0);
  __syncthreads();
  int buf = 0;
  for (int row0 = R0; row0 < R1; row0 += 32, buf ^= 1) {
    const bool more = row0 + 32 < R1;
    if (more) fetch(row0 + 32);
    bf16x8 a[3][NS];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) a[p][s2] = *reinterpret_cast<const bf16x8*>(&xt[buf][p][r * XS + 16 * s2 + 8 * hh]);
    float* __restrict__ ob = out + (size_t)z * out_rows * ldo;
    int blk_ri[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ri = min(row0 + 8 * p + (lane >> 3), R1 - 1);
      blk_ri[p] = rows ? rows[ri] : ri;
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      // smallest terms first; quarter q of the pending block leaves between the MFMAs (LDS read at MFMA q NM/4, store 2 later)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // (xl wh) (xh wl) (xm wm) (xm wh) (xh wm) (xh wh)
        const int pr = m / NS, s2 = m % NS;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pr]][s2], wr[t][PB[pr]][s2], acc, 0, 0, 0);
        static_assert(NM / 4 >= 3 && NM % 4 == 0, "quarter schedule");
        if (m % (NM / 4) == 0) drain_read(m / (NM / 4));
        if (m % (NM / 4) == 2) drain_store(m / (NM / 4));
      }
      float* tl = st[wave][pbuf];
#pragma unroll
      for (int i = 0; i < 16; ++i) tl[((i & 3) + 8 * (i >> 2) + 4 * hh) * TS + r] = acc[i];
      pend_ob = ob;
#pragma unroll
      for (int p = 0; p < 4; ++p) pend_ri[p] = blk_ri[p];
      pend_c0 = col0 + 32 * t;
      pend_rows = R1 - row0;
      pend_buf = pbuf;
      pbuf ^= 1;
    }
    if (more) park(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {   // the last block
    drain_read(p);
    drain_store(p);
  }
  __syncthreads();
  }
}

// ---- fp16 hi/lo split form ("h2", round 4): the same product on v_mfma_f32_32x32x16_f16 with BOTH operands split into two
// halves, v = hi + lo / 2048 (hi = fp16(v), lo = fp16((v - hi) * 2048): 22 significant bits), three products per 16 k -
//   x w ~ xh wh + (xh wl + xl wh) / 2048                      dropped: xl wl / 2^22
// - accumulated in fp32 in two registers sets (main, correction), exactly as the conv kernels do (csrc/ddp_conv.hip, h2 form:
// the fp32 chain's error class: <= 2^-20 sum|x w| over the test products, tools/micro/f16x2_mfma.hip).  12 MFMAs of 32 cycles per 32 x 32 block at K = 60
// instead of 30 of 64: the matrix side shrinks 5 x and the kernel is left with its stores.  Deterministic, row results do not
// depend on the tile a row sits in.  Weights arrive pre-split from the host (packing.split_h2): [batch][plane][k/16][k/8 % 2][col][8].
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#ifndef DDP_SAH_CT_GH
#define DDP_SAH_CT_GH 4     // (round 5, end: the one-accumulator form of the GH path freed the registers of a fourth tile's weights: 253 VGPRs)
#endif
#ifndef DDP_SAH_CT
#define DDP_SAH_CT 4   // column tiles per wave: 2 planes x 4 k-steps x 4 registers each = 32 weight registers per tile
#endif

// GH (round 5): the G part of a row leaves as fp16 hi/lo operand planes - the layout ddp_conv_rows reads G in
// (include/ddp_hip.h, ddp_conv_task_t::gh): a lane then drains 8 consecutive columns of a parked row (one 8-k group of one G column)
// as two 16-byte pieces, hi plane and lo plane, where the fp32 form stores two 16-byte quads - the same bytes, the same number of
// store instructions.  Where the two pieces of every 8-column group go is a table (gh_dest[batch][ncols / 8][2], float offsets inside
// the row, built by the host from the parts of the G array; bit 0 of the first = a plane group): the drain has no arithmetic of its
// own.  The other groups are fp32 columns (Gb, padding) and leave unconverted.  Needs ncols % 32 == 0.
// G3 (round 6, with GH; plane form 1 of include/ddp_hip.h, ddp_conv_task_t::gh_fmt = 1): a plane group leaves as 24 bytes - 8 fp16 hi words =
// V TRUNCATED to fp16 after rounding V to 19 significant bits, then 8 bytes = the next 8 mantissa bits of each value - and EVERY 8-column
// group g of the product sits at byte 24 g of the row (fp32 groups - Gb, padding - carry 6 values: product columns 0, 1, 4, 5, 2, 6 of the
// group in that order, columns 3 and 7 are not stored).  The drain is a different one (profiles/r06_store_g3_micro.txt: partial lines that
// complete a block later cost a quarter of the rate, a row group's bytes written by consecutive instructions do not): a lane converts its
// accumulator quads - 4 consecutive columns of its row, the transposed product - straight out of the registers into the wave's image of
// the row tile in LDS (32 rows x the 384 bytes of the wave's 16 groups: no fp32 park, no re-read for the conversion), and behind the row
// tile's fourth block the wave copies the image out with twelve 16-byte store instructions of WHOLE 128-byte lines (8 rows x 384 bytes per
// three instructions).  Needs ncols % 128 == 0 (a wave's four blocks are all there or none is) and ldo = 6 ncols / 8.
// (Measured and dropped: the waves running FREE - every wave fetching and splitting its own operand fragments, no shared x tile, no barrier
// in the loop - 1.05 instead of 0.89 ms on the atom stack, the step 15.2 instead of 14.3 ms: four times the split work, 22 spilled registers;
// the image leaving one row group behind each k-step of the NEXT tile's first block instead of as a burst - 0.92 ms, the step 14.7 ms.)
template <int KT, bool GH = false, bool G3 = false>
__global__ __launch_bounds__(DDP_GEMM_THREADS, 2) void ddp_stage_a_h2_kernel(const float* __restrict__ x, int ldx, int nrows,
                                                                              const int32_t* __restrict__ rows,
                                                                              const int32_t* __restrict__ nrows_dev, int out_rows,
                                                                              int mrows, const GemmOffs offs,
                                                                              const _Float16* __restrict__ wh, int ncols,
                                                                              float* __restrict__ out, int ldo, int32_t* range_flag,
                                                                              const int32_t* __restrict__ gh_dest = nullptr) {
  if (nrows_dev) nrows = min(nrows, *nrows_dev);
  if ((int)blockIdx.y * mrows >= nrows) return;
  // (GH: four column tiles per wave since the block has ONE accumulator - 512 contiguous bytes per row and wave; with the 2048-scaled
  // planes' two accumulators three tiles were what fitted)
  constexpr int KP = (KT + 15) / 16 * 16, NS = KP / 16, CT = GH ? DDP_SAH_CT_GH : DDP_SAH_CT;
  constexpr int XS = KP + 8;                                   // halves per LDS row of an x plane (16-byte aligned rows)
  constexpr int NV = (32 * KT / 4 + DDP_GEMM_THREADS - 1) / DDP_GEMM_THREADS;
  constexpr int TS = 36;
  constexpr int RS3 = 392;                                     // G3: bytes per row of a wave's row-tile image (384 + 8: 98 words, two-way banks at most)
  __shared__ __attribute__((aligned(16))) _Float16 xt[2][2][32 * XS];
  __shared__ __attribute__((aligned(16))) float st[G3 ? 1 : 4][2][G3 ? 4 : 32 * TS];
  __shared__ __attribute__((aligned(16))) char g3t[G3 ? 4 : 1][G3 ? 32 * RS3 : 16];
  const int z = (int)blockIdx.z, tid = (int)threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int col0 = ((int)blockIdx.x * 4 + wave) * (32 * CT);
  h8 wr[CT][2][NS];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = min(col0 + 32 * t + r, ncols - 1);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
        wr[t][p][s2] = *reinterpret_cast<const h8*>(wh + (((((size_t)z * 2 + p) * NS + s2) * 2 + hh) * ncols + c) * 8);
  }
  // the k padding [KT, KP) of both x buffers is zero for good
  if constexpr (KP > KT) {
    for (int i = tid; i < 2 * 2 * 32 * (KP - KT); i += DDP_GEMM_THREADS) {
      const int kk = i % (KP - KT), rr = (i / (KP - KT)) % 32, bp = i / ((KP - KT) * 32);
      xt[bp / 2][bp % 2][rr * XS + KT + kk] = (_Float16)0.f;
    }
  }
  const float* __restrict__ xb = x + offs.off[z];
  float* __restrict__ ob = out + (size_t)z * out_rows * ldo;   // the batch slice; row indices below are absolute
  // Column quad of this lane inside a 32-column block, CLAMPED to the block's last valid quad (ncols % 4 == 0): a lane beyond the
  // array's last column re-stores its neighbour's quad (same address, same data) instead of sitting under a branch.  Blocks that
  // start beyond ncols are skipped as a whole (wave-uniform).
  int cq[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) cq[t] = max(0, min(4 * (lane & 7), ncols - 4 - (col0 + 32 * t)));
  // GH: a lane owns ONE 8-column group of a row - lane (row 16 h + lane / 4, group lane & 3) for the two halves h of the tile -, converts
  // it once and stores its hi words with one instruction and its lo words with the next (the two 16-byte pieces of a group are neighbours:
  // the pair of instructions completes the 128-byte lines).  (Until the end of round 5 two lanes converted the same group and kept one
  // plane each: twice the LDS reads and conversions per byte.)  gh_o[t][0 / 1]: float offsets (inside a row) of the group's two pieces;
  // gh_sp[t]: a plane group (else 8 fp32 columns: first and second quad as they are)
  int gh_o[CT][2];
  bool gh_sp[CT];
  float gh_max = 0.f;                   // GH: largest |G value| split (outside the fp16 range: reported through range_flag)
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    gh_o[t][0] = gh_o[t][1] = 0;
    gh_sp[t] = false;
    if constexpr (GH) {
      const int g = min(col0 + 32 * t + 8 * (lane & 3), ncols - 8) >> 3;
      const int32_t* __restrict__ d = gh_dest + ((size_t)z * (ncols >> 3) + g) * 2;
      gh_o[t][0] = d[0] & ~3;
      gh_o[t][1] = d[1];
      gh_sp[t] = (d[0] & 1) != 0;
    }
  }
  // G3: which of the wave's 16 groups are plane groups (bit 4 t + q), and where lane i = 64 j + lane of a drain instruction j < 3 reads /
  // writes: 8 rows x 24 16-byte pieces per three instructions
  unsigned sp3 = 0u;
  int g3_rr[3] = {0, 0, 0}, g3_pc[3] = {0, 0, 0};
  if constexpr (G3) {
    const int g = min(col0 + 8 * (lane & 15), ncols - 8) >> 3;
    sp3 = (unsigned)__ballot((gh_dest[((size_t)z * (ncols >> 3) + g) * 2] & 1) != 0) & 0xffffu;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g3_rr[j] = (64 * j + lane) / 24;
      g3_pc[j] = (64 * j + lane) % 24;
    }
  }
  for (int R0 = (int)blockIdx.y * mrows; R0 < nrows; R0 += (int)gridDim.y * mrows) {
  const int R1 = min(nrows, R0 + mrows);
  const bool al4 = ((ldx | offs.off[z]) & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
  f32x4 xv[NV];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;
      const int rr = min(i / (KT / 4), 31), q = i % (KT / 4);
      const int ri = min(row0 + rr, nrows - 1);
      const float* __restrict__ p = xb + (size_t)(rows ? rows[ri] : ri) * ldx + 4 * q;
      if (al4)
        xv[v] = *reinterpret_cast<const f32x4*>(p);
      else
        xv[v] = f32x4{p[0], p[1], p[2], p[3]};
    }
  };
  auto park = [&](int buf) {   // split into the two fp16 planes on the way into LDS (once per workgroup, not per wave)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * DDP_GEMM_THREADS;
      if (i < 32 * (KT / 4)) {
        const int rr = i / (KT / 4), q = i % (KT / 4);
        h4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // GH: UNIFIED planes (hi and lo at one scale, DDP_GH_SX x: the weights carry 1 / DDP_GH_SX, so that the ONE accumulator of a
          // block is the plane value itself - no second accumulator, no join); otherwise v = hi + lo / 2048
          const float f = GH ? xv[v][e] * (float)DDP_GH_SX : xv[v][e];
          if (!(fabsf(f) <= 65504.f) && range_flag) *range_flag = 1;     // outside the fp16 range (or NaN): reported, not saturated
          h[e] = (_Float16)f;
          l[e] = (_Float16)(GH ? f - (float)h[e] : (f - (float)h[e]) * 2048.f);
        }
        *reinterpret_cast<h4*>(&xt[buf][0][rr * XS + 4 * q]) = h;
        *reinterpret_cast<h4*>(&xt[buf][1][rr * XS + 4 * q]) = l;
      }
    }
  };
  // The pending block of this wave: parked in LDS, written out quarter by quarter between the MFMAs of the NEXT block, WITHOUT a
  // condition in the loop - rows and columns beyond the array are clamped onto valid ones (duplicate stores of identical data), and
  // the wave's very first block, which has no predecessor, is peeled (DRAIN = false).  (Under `if (pending)` / `if (row < rows)`
  // hipcc's waitcnt insertion drained the store queue at the branch joins.)
  size_t pend_off[4] = {0, 0, 0, 0};    // element offset of this lane's 16-byte piece of quarter p of the pending block
  int pend_lds[4] = {0, 0, 0, 0};       // ... and where it sits in the parked tile
  bool pend_sp = false;                 // GH: the pending block's groups are plane groups
  int pbuf = 0;
  f32x4 dv = {0.f, 0.f, 0.f, 0.f};
  // GH: drain step p of the pending tile.  Even p: the 8 columns of this lane's group of half p / 2 are read and converted, dv = their hi
  // words (fp32 groups: first quad), dv2 = the lo words (second quad); odd p: dv = dv2.  Packed conversions, the range check as one
  // running max, no branch (a NaN needs an inf or a NaN in x, which the x split reports)
  f32x4 dv2 = {0.f, 0.f, 0.f, 0.f};
  auto gh_read = [&](const float* pt, int p) {
    if (p & 1) {
      dv = dv2;
      return;
    }
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    const f32x4 a = *reinterpret_cast<const f32x4*>(&pt[pend_lds[p]]), b = *reinterpret_cast<const f32x4*>(&pt[pend_lds[p] + 4]);
    const f32x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const h8 hi = __builtin_convertvector(f, h8);
    const f32x8 rest = f - __builtin_convertvector(hi, f32x8);     // (unified planes: lo at hi's scale, include/ddp_hip.h DDP_ROWS_S*)
    const h8 lo = __builtin_convertvector(rest, h8);
    const float m8 = fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))), fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
    gh_max = fmaxf(gh_max, pend_sp ? m8 : 0.f);
    const f32x4 vh = __builtin_bit_cast(f32x4, hi), vl = __builtin_bit_cast(f32x4, lo);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dv[e] = pend_sp ? vh[e] : a[e];
      dv2[e] = pend_sp ? vl[e] : b[e];
    }
  };
  auto block = [&](auto drain_tag, int t, const h8 (&a)[2][NS], const int (&blk_ri)[4], int nr) {
    constexpr bool DRAIN = decltype(drain_tag)::value;
    f32x16 am, ac;
#pragma unroll
    for (int i = 0; i < 16; ++i) { am[i] = 0.f; ac[i] = 0.f; }
    const float* pt = st[wave][pbuf ^ 1];     // the pending block's tile
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      if constexpr (DRAIN && NS == 4) {
        if constexpr (GH)
          gh_read(pt, s2);
        else
          dv = *reinterpret_cast<const f32x4*>(&pt[pend_lds[s2]]);
      }
      if constexpr (GH) {
        // the TRANSPOSED product (A = the weights' fragment, B = the rows'): lane (row r, hh) then holds 4 consecutive columns per group
        // of accumulator registers - the tile is parked with four 16-byte LDS writes instead of sixteen 4-byte ones
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][0][s2], a[0][s2], am, 0, 0, 0);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][1][s2], a[0][s2], am, 0, 0, 0);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][0][s2], a[1][s2], am, 0, 0, 0);
      } else {
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s2], wr[t][0][s2], am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s2], wr[t][1][s2], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][s2], wr[t][0][s2], ac, 0, 0, 0);
      }
      if constexpr (DRAIN && NS == 4) {
        {
          // (the hi and the lo store of a group leave one k-step apart and complete whole 128-byte lines together; as CONSECUTIVE
          // instructions behind the odd k-step - measured, round 6 - the atom stack takes 1.20 - 1.23 ms against 1.22: no difference)
          *reinterpret_cast<f32x4*>(&ob[pend_off[s2]]) = dv;
        }
      }
    }
    if constexpr (DRAIN && NS != 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if constexpr (GH)
          gh_read(pt, p);
        else
          dv = *reinterpret_cast<const f32x4*>(&pt[pend_lds[p]]);
        *reinterpret_cast<f32x4*>(&ob[pend_off[p]]) = dv;
      }
    }
    float* tl = st[wave][pbuf];
    if constexpr (GH) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<f32x4*>(&tl[r * TS + 8 * q4 + 4 * hh]) = f32x4{am[4 * q4], am[4 * q4 + 1], am[4 * q4 + 2], am[4 * q4 + 3]};
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) tl[((i & 3) + 8 * (i >> 2) + 4 * hh) * TS + r] = am[i] + ac[i] * (1.f / 2048.f);
    }
    if constexpr (GH) {
#if defined(DDP_GH_ABL) && DDP_GH_ABL == 1     // timing only: the plane destinations without the conversion
      pend_sp = false;
#else
      pend_sp = gh_sp[t];
#endif
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int rr = min(16 * (p >> 1) + (lane >> 2), nr - 1);        // clamped row of the tile (blk_ri is clamped the same way)
        pend_lds[p] = rr * TS + 8 * (lane & 3);
        pend_off[p] = (size_t)blk_ri[p >> 1] * ldo + gh_o[t][p & 1];
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int rr = min(8 * p + (lane >> 3), nr - 1);        // clamped row of the tile (blk_ri is clamped the same way)
        pend_lds[p] = rr * TS + cq[t];
        pend_off[p] = (size_t)blk_ri[p] * ldo + (col0 + 32 * t + cq[t]);
      }
    }
    pbuf ^= 1;
  };
  // G3: block t of the row tile - the transposed product, then this lane's four accumulator quads (row r, product columns 32 t + 8 q + 4 hh
  // + 0..3: half of group q) into the wave's image: 8 bytes of hi words at 24 (4 t + q) + 8 hh, 4 continuation bytes at + 16 + 4 hh
  auto block3 = [&](int t, const h8 (&a)[2][NS]) {
    f32x16 am;
#pragma unroll
    for (int i = 0; i < 16; ++i) am[i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][0][s2], a[0][s2], am, 0, 0, 0);
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][1][s2], a[0][s2], am, 0, 0, 0);
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[t][0][s2], a[1][s2], am, 0, 0, 0);
    }
    char* tl = g3t[G3 ? wave : 0] + r * RS3 + 96 * t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool sp = ((sp3 >> (4 * t + q)) & 1u) != 0u;      // (wave-uniform)
      const float v0 = am[4 * q], v1 = am[4 * q + 1], v2 = am[4 * q + 2], v3 = am[4 * q + 3];
      gh_max = fmaxf(gh_max, sp ? fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))) : 0.f);
      // V rounded to 19 significant bits (half of bit 5 of the fp32 mantissa added to the bit pattern: magnitude rounding, carries into the
      // exponent where it must), hi = its truncation to fp16, byte = mantissa bits 12 .. 5 (below the fp16 normal range the byte means
      // nothing and the reader multiplies it by the hi word's zero exponent)
      const uint32_t b0 = __builtin_bit_cast(uint32_t, v0) + 0x10u, b1 = __builtin_bit_cast(uint32_t, v1) + 0x10u;
      const uint32_t b2 = __builtin_bit_cast(uint32_t, v2) + 0x10u, b3 = __builtin_bit_cast(uint32_t, v3) + 0x10u;
      typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
      const pk2 h01 = __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
      const pk2 h23 = __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, b2), __builtin_bit_cast(float, b3));
      const uint32_t lo = ((b0 >> 5) & 0xffu) | (((b1 >> 5) & 0xffu) << 8) | (((b2 >> 5) & 0xffu) << 16) | ((b3 >> 5) << 24);
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      u32x2 w8;
      w8[0] = sp ? __builtin_bit_cast(uint32_t, h01) : __builtin_bit_cast(uint32_t, v0);
      w8[1] = sp ? __builtin_bit_cast(uint32_t, h23) : __builtin_bit_cast(uint32_t, v1);
      *reinterpret_cast<u32x2*>(tl + 24 * q + 8 * hh) = w8;
      *reinterpret_cast<uint32_t*>(tl + 24 * q + 16 + 4 * hh) = sp ? lo : __builtin_bit_cast(uint32_t, v2);
    }
  };
  // G3: the row tile's image leaves - twelve 16-byte stores of whole lines; rows behind the tile's last repeat it (identical data)
  auto drain3 = [&](int myrow, int nr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const char* img = g3t[G3 ? wave : 0];
    float* __restrict__ oc = ob + 6 * (col0 >> 3);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int rl = min(8 * p + g3_rr[j], nr - 1);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 lo8 = *reinterpret_cast<const u32x2*>(img + rl * RS3 + 16 * g3_pc[j]);
        const u32x2 hi8 = *reinterpret_cast<const u32x2*>(img + rl * RS3 + 16 * g3_pc[j] + 8);
        const int ri = __shfl(myrow, rl);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(oc + (size_t)ri * ldo + 4 * g3_pc[j]) = u32x4{lo8[0], lo8[1], hi8[0], hi8[1]};
      }
    __builtin_amdgcn_wave_barrier();
  };
  fetch(R0);
  park(0);
  __syncthreads();
  int buf = 0;
  bool first = true;
  for (int row0 = R0; row0 < R1; row0 += 32, buf ^= 1) {
    const bool more = row0 + 32 < R1;
    if (more) fetch(row0 + 32);
    h8 a[2][NS];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) a[p][s2] = *reinterpret_cast<const h8*>(&xt[buf][p][r * XS + 16 * s2 + 8 * hh]);
    int blk_ri[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // (GH: entries 0, 1 = the lane's rows 16 h + lane / 4 of the tile's two halves)
      const int ri = min(row0 + (GH ? 16 * (p & 1) + (lane >> 2) : 8 * p + (lane >> 3)), R1 - 1);
      blk_ri[p] = rows ? rows[ri] : ri;
    }
    const int nr = R1 - row0;
    if constexpr (G3) {
      if (col0 < ncols) {                              // (wave-uniform; ncols % 128 == 0: all four blocks or none)
        const int ri = min(row0 + r, R1 - 1);
        const int myrow = rows ? rows[ri] : ri;
#pragma unroll
        for (int t = 0; t < CT; ++t) block3(t, a);
        drain3(myrow, nr);
      }
      if (more) park(buf ^ 1);
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      if (col0 + 32 * t >= ncols) continue;           // (wave-uniform: a column block beyond the array)
      if (t == 0 && first)
        block(std::false_type{}, t, a, blk_ri, nr);
      else
        block(std::true_type{}, t, a, blk_ri, nr);
      first = false;
    }
    if (more) park(buf ^ 1);
    __syncthreads();
  }
  if (!first) {   // the last block
    const float* pt = st[wave][pbuf ^ 1];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (GH)
        gh_read(pt, p);
      else
        dv = *reinterpret_cast<const f32x4*>(&pt[pend_lds[p]]);
      *reinterpret_cast<f32x4*>(&ob[pend_off[p]]) = dv;
    }
  }
  __syncthreads();
  }
  if constexpr (GH) {
    if (!(gh_max <= 65504.f) && range_flag) *range_flag = 1;
  }
}

static int stage_a_impl(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows,
                        const int32_t* offs, int nbatch, const float* w, const void* w_bf16x3, const void* w_h2, int k, int ncols, float* out,
                        int ldo, int32_t* range_flag, void* stream, const int32_t* gh_dest = nullptr, int gh_fmt = 0) {
  if (!rows) out_rows = nrows;                       // dense: out[b] has one row per x row
  if (out_rows < 1 && nrows > 0) return ddp_fail(DDP_EINVAL, "ddp_stage_a: out_rows");
  if (nbatch < 0 || nbatch > DDP_MAX_GEMM_BATCH) return ddp_fail(DDP_ELIMIT, "ddp_stage_a: nbatch > DDP_MAX_GEMM_BATCH");
  if (k < 2 || k > 64 || (k & 1)) return ddp_fail(DDP_ELIMIT, "ddp_stage_a: K must be even and in [2, 64]");
  // (plane form 1 - fp16 hi + a continuation byte - writes 24 bytes per 8 product columns: its rows are shorter than ncols floats)
  if (ncols < 1 || nrows < 0 || (ldo < ncols && !(gh_dest && gh_fmt == 1 && ldo >= (ncols / 8) * 6)))
    return ddp_fail(DDP_EINVAL, "ddp_stage_a: ncols / nrows / ldo");
  if (gh_dest && gh_fmt == 1 && ((ncols & 127) != 0 || (ldo & 31) != 0))
    return ddp_fail(DDP_EINVAL, "ddp_stage_a_gh3: ncols % 128 == 0 (a wave's four column blocks), ldo % 32 == 0 (whole 128-byte lines)");
  if (nbatch == 0 || nrows == 0) return 0;
  if (!x || !offs || !w || !out) return ddp_fail(DDP_EINVAL, "ddp_stage_a: null argument");
  if ((reinterpret_cast<size_t>(x) & 3) || (reinterpret_cast<size_t>(out) & 7))   // scalar loads: dword aligned
    return ddp_fail(DDP_EINVAL, "ddp_stage_a: x must be 4-byte aligned, out 8-byte aligned");
  GemmOffs O;
  for (int i = 0; i < DDP_MAX_GEMM_BATCH; ++i) O.off[i] = (i < nbatch) ? offs[i] : 0;
  hipStream_t s = (hipStream_t)stream;
  const int nry = (nrows + DDP_GEMM_ROWS - 1) / DDP_GEMM_ROWS;
#define DDP_GEMM_LAUNCH(KT, CPL)                                                                                 \
  hipLaunchKernelGGL((ddp_stage_a_kernel<KT, CPL>), dim3((ncols + CPL * DDP_GEMM_THREADS - 1) / (CPL * DDP_GEMM_THREADS), nry, nbatch), \
                     dim3(DDP_GEMM_THREADS), 0, s, x, ldx, nrows, rows, nrows_dev, out_rows, O, w, k, ncols, out, ldo)
  const bool wide = ((ncols | ldo) & 3) == 0 && ncols >= 4 * DDP_GEMM_THREADS && (reinterpret_cast<size_t>(out) & 15) == 0;
#define DDP_GEMM_MFMA(KT)                                                                                        \
  hipLaunchKernelGGL((ddp_stage_a_mfma_kernel<KT>), dim3((ncols + 128 * DDP_SA_CT - 1) / (128 * DDP_SA_CT), gy, nbatch), \
                     dim3(DDP_GEMM_THREADS), 0, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O, w, ncols, out, ldo)
  // a row list with a device-side length is usually a small part of its capacity: 128-row tiles (as many workgroups as an
  // exactly-sized launch of the actual list would get), a bounded grid, the kernel walks its tiles
  const bool listed = rows && nrows_dev;
  // rows per workgroup, measured with the h2 kernel (tools/bench_stage_a_rows.py, us for 128 -> 256 rows per workgroup): 695 x 6
  // products 56 -> 43, 1480 x 6 112 -> 101, 5560 x 6 408 -> 386, 5555 x 2 149 -> 145, a 12000-row list of 44440 307 -> 294 (a
  // 2903-row one 73 -> 74); 185 x 6: 23 -> 20 with 64.  A list of a small array is usually short: 128 stays
  int mrows = listed ? (nrows >= 16384 ? 256 : 128) : (nrows >= 8192 ? DDP_GEMM_MROWS : nrows >= 384 ? 256 : 64);
  int gy = (nrows + mrows - 1) / mrows;
  if (listed && gy > 96) gy = 96;
#ifdef DDP_SA_TUNE   // diagnostic builds: rows per workgroup of products below 8192 rows / the grid cap of listed products from the environment
  if (nrows < 8192 || listed) {
    if (const char* e = getenv("DDP_SA_MROWS")) mrows = atoi(e);
    gy = (nrows + mrows - 1) / mrows;
    const char* c = getenv("DDP_SA_GYCAP");
    const int cap = c ? atoi(c) : 96;
    if (listed && gy > cap) gy = cap;
  }
#endif
  static const bool no_mfma = getenv("DDP_STAGE_A_VALU") != nullptr;   // diagnostic: force the VALU form
  // h2 form (fp16 hi/lo split of both operands): wide, 16-byte aligned outputs only, K = 60 / 32 / 24 / 16 (KP = 64 / 32 / 32 / 16)
  if (gh_dest) {
    if (!(w_h2 && wide && (k == 60 || k == 32 || k == 24 || k == 16) && (reinterpret_cast<size_t>(w_h2) & 15) == 0 && (ncols & 31) == 0 &&
          (reinterpret_cast<size_t>(gh_dest) & 7) == 0))
      return ddp_fail(DDP_EINVAL, "ddp_stage_a_gh: plane output needs the h2 path (w_h2, k in {60, 32, 24, 16}) and ncols % 32 == 0");
    const dim3 grid((ncols + 128 * DDP_SAH_CT_GH - 1) / (128 * DDP_SAH_CT_GH), gy, nbatch);
    // occupancy shaping (ddp_set_occupancy_shaping): extra dynamic LDS on top of the kernel's static 55 KB = one workgroup per CU
    const int pad = ddp_shape_stage_a_pad;
    static int pad_have[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DDP_GEMM_GH(KT, I)                                                                                       \
    {                                                                                                            \
      if (pad > 0) {                                                                                             \
        const hipError_t ep = (gh_fmt == 1) ? ddp_need_lds(reinterpret_cast<const void*>(ddp_stage_a_h2_kernel<KT, true, true>), pad, &pad_have[4 + I]) \
                                            : ddp_need_lds(reinterpret_cast<const void*>(ddp_stage_a_h2_kernel<KT, true>), pad, &pad_have[I]); \
        if (ep != hipSuccess) return ddp_fail_hip(ep, "hipFuncSetAttribute(stage A)");                           \
      }                                                                                                          \
      if (gh_fmt == 1)                                                                                           \
        hipLaunchKernelGGL((ddp_stage_a_h2_kernel<KT, true, true>), grid, dim3(DDP_GEMM_THREADS), pad, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O, \
                           reinterpret_cast<const _Float16*>(w_h2), ncols, out, ldo, range_flag, gh_dest);       \
      else                                                                                                       \
        hipLaunchKernelGGL((ddp_stage_a_h2_kernel<KT, true>), grid, dim3(DDP_GEMM_THREADS), pad, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O, \
                           reinterpret_cast<const _Float16*>(w_h2), ncols, out, ldo, range_flag, gh_dest);       \
    }
    switch (k) {
      case 60: DDP_GEMM_GH(60, 0); break;
      case 32: DDP_GEMM_GH(32, 1); break;
      case 24: DDP_GEMM_GH(24, 2); break;
      default: DDP_GEMM_GH(16, 3); break;
    }
#undef DDP_GEMM_GH
    const hipError_t eg = hipGetLastError();
    if (eg != hipSuccess) return ddp_fail_hip(eg, "ddp_stage_a_gh launch");
    return 0;
  }
  if (w_h2 && wide && (k == 60 || k == 32 || k == 24 || k == 16) && !no_mfma && (reinterpret_cast<size_t>(w_h2) & 15) == 0) {
    const dim3 grid((ncols + 128 * DDP_SAH_CT - 1) / (128 * DDP_SAH_CT), gy, nbatch);
#define DDP_GEMM_H2(KT)                                                                                          \
    hipLaunchKernelGGL((ddp_stage_a_h2_kernel<KT>), grid, dim3(DDP_GEMM_THREADS), 0, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O, \
                       reinterpret_cast<const _Float16*>(w_h2), ncols, out, ldo, range_flag)
    switch (k) {
      case 60: DDP_GEMM_H2(60); break;
      case 32: DDP_GEMM_H2(32); break;
      case 24: DDP_GEMM_H2(24); break;
      default: DDP_GEMM_H2(16); break;
    }
#undef DDP_GEMM_H2
    const hipError_t eh = hipGetLastError();
    if (eh != hipSuccess) return ddp_fail_hip(eh, "ddp_stage_a (h2) launch");
    return 0;
  }
  // bf16x3 form: wide, 16-byte aligned outputs only (its blocks always leave through the LDS transpose)
  if (w_bf16x3 && wide && (k == 60 || k == 32) && !no_mfma && (reinterpret_cast<size_t>(w_bf16x3) & 15) == 0) {
    const dim3 grid((ncols + 128 * DDP_SA3_CT - 1) / (128 * DDP_SA3_CT), gy, nbatch);
    if (k == 60)
      hipLaunchKernelGGL((ddp_stage_a_x3_kernel<60>), grid, dim3(DDP_GEMM_THREADS), 0, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O,
                         reinterpret_cast<const __bf16*>(w_bf16x3), ncols, out, ldo);
    else
      hipLaunchKernelGGL((ddp_stage_a_x3_kernel<32>), grid, dim3(DDP_GEMM_THREADS), 0, s, x, ldx, nrows, rows, nrows_dev, out_rows, mrows, O,
                         reinterpret_cast<const __bf16*>(w_bf16x3), ncols, out, ldo);
    const hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) return ddp_fail_hip(e3, "ddp_stage_a (bf16x3) launch");
    return 0;
  }
  if (ncols >= 512 && !no_mfma && (k == 60 || k == 64 || k == 32 || k == 24 || k == 16)) {
    switch (k) {
      case 60: DDP_GEMM_MFMA(60); break;
      case 64: DDP_GEMM_MFMA(64); break;
      case 32: DDP_GEMM_MFMA(32); break;
      case 24: DDP_GEMM_MFMA(24); break;
      default: DDP_GEMM_MFMA(16); break;
    }
  } else
  switch (k) {
    case 60: if (wide) DDP_GEMM_LAUNCH(60, 4); else DDP_GEMM_LAUNCH(60, 2); break;
    case 32: DDP_GEMM_LAUNCH(32, 2); break;
    case 24: DDP_GEMM_LAUNCH(24, 2); break;
    case 16: DDP_GEMM_LAUNCH(16, 2); break;
    default: DDP_GEMM_LAUNCH(0, 2); break;
  }
#undef DDP_GEMM_LAUNCH
#undef DDP_GEMM_MFMA
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_stage_a launch");
  return 0;
}

extern "C" int ddp_stage_a(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows,
                           const int32_t* offs, int nbatch, const float* w, const void* w_bf16x3, int k, int ncols, float* out, int ldo,
                           void* stream) {
  return stage_a_impl(x, ldx, nrows, rows, nrows_dev, out_rows, offs, nbatch, w, w_bf16x3, nullptr, k, ncols, out, ldo, nullptr, stream);
}

// ddp_stage_a with the weights ALSO given as fp16 hi/lo planes (packing.split_h2): the product then runs on the fp16 matrix cores
// with both operands split (x in the kernel), when the shape allows it (wide 16-byte aligned outputs, K in {60, 32, 24, 16}); other
// shapes fall back to the exact fp32 forms on `w`.
extern "C" int ddp_stage_a_h2(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows,
                              const int32_t* offs, int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo,
                              int32_t* range_flag, void* stream) {
  return stage_a_impl(x, ldx, nrows, rows, nrows_dev, out_rows, offs, nbatch, w, nullptr, w_h2, k, ncols, out, ldo, range_flag, stream);
}

extern "C" int ddp_stage_a_gh(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows,
                              const int32_t* offs, int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo,
                              int32_t* range_flag, const int32_t* dest, void* stream) {
  if (!dest) return ddp_fail(DDP_EINVAL, "ddp_stage_a_gh: dest");
  return stage_a_impl(x, ldx, nrows, rows, nrows_dev, out_rows, offs, nbatch, w, nullptr, w_h2, k, ncols, out, ldo, range_flag, stream, dest);
}

// ddp_stage_a_gh with plane form 1 of ddp_conv_task_t::gh (fp16 hi + a continuation byte: 24 bytes per 8 values, 19 significant bits; ABI 17).
// `ldo` (floats per output row) is then the row length of that form (packing.gh3_ld = 6 ncols / 8), smaller than ncols; of `dest` only bit 0 of
// entry [group][0] is read (a plane group or an fp32 group): every group sits at byte 24 g of the row.
extern "C" int ddp_stage_a_gh3(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows,
                               const int32_t* offs, int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo,
                               int32_t* range_flag, const int32_t* dest, void* stream) {
  if (!dest) return ddp_fail(DDP_EINVAL, "ddp_stage_a_gh3: dest");
  return stage_a_impl(x, ldx, nrows, rows, nrows_dev, out_rows, offs, nbatch, w, nullptr, w_h2, k, ncols, out, ldo, range_flag, stream, dest, 1);
}
