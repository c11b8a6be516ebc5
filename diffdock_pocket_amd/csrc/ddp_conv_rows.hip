// ddp_conv_rows.hip - the factorised conv (fc -> FasterTensorProduct -> message, csrc/ddp_conv.hip's contract) through 256-edge,
// ROW-STATIONARY workgroups for gfx950 (round 5).
//
// Replaces, per conv (reference file:line):  edge_attr_ = cat(...)  models/all_atom_score_model.py:273-312;  w = fc(edge_attr_)
// models/score_model.py:100-105,114;  msg = FasterTensorProduct(x[src], sh, w)  models/layers.py:34-85.
//
// Why: the 32-edge kernel (ddp_conv32_kernel<60, h2>) streams the whole packed fc.3 weight set (1.47 MB) and ~0.5 MB of G rows
// through every 32-edge workgroup: 278 M L2 requests of 128 B per 2.3-ms launch = 15.3 TB/s of L2 -> CU traffic, L2 busy 90 %
// (profiles/r05_pmc_conv32_h2_l1_l2.json) against the 16.8 - 18.8 TB/s the chip delivers from its XCD L2s - the matrix pipe sat at
// 31 %.  Here the weights leave L2 once per 256 edges:
//   workgroup = 8 waves (two per SIMD, 256 registers each), ONE per CU, 256 consecutive (source-ordered) edges of one conv;
//   wave w owns the 32 edges [32 w, 32 w + 32) for the whole kernel and keeps h = relu(fc1) of them as the A-operand fragments
//     of v_mfma_f32_32x32x16_f16 in REGISTERS (fp16 hi/lo planes: 8 NS registers);
//   the weight tiles (task.wsh: the fc.0 tiles, then the fc.3 tiles segment by segment) are staged once per workgroup through a
//     two-slot LDS ring - every wave loads 1/8 of the next tile, one barrier per tile - and all eight waves read their B operands
//     from LDS (conflict-free lane-linear 16-byte reads);
//   h comes from the TRANSPOSED fc1 product (A = fc.0 tile, B = edge_attr_ fragments gathered straight from the three row
//     segments): the accumulator of column tile ct leaves lane (edge, hh) with 16 h values of its own edge, which ARE the k-groups
//     (2 ct, hh) and (2 ct + 1, hh) of the next products in the permuted k order DDP_ROWS_KPERM (the host packs fc.3 and G in it);
//   a segment = one 32-column part of one weight block's output columns: a wave accumulates EVERYTHING that lands there in
//     registers - first the factorised features (one pass of the same tile product per run of edges with one source node, B = the
//     node's G tile in plane form, task.gh, straight from memory through a 4-fragment register ring; rows outside the run are
//     masked in the epilogue), then the segment's stream tiles (the vector-input features) - and stores the message columns;
//     no message tile in LDS, no cross-wave reduction, no atomics.
// Per 32 edges: 0.18 MB of weights + ~0.26 MB of G (whole runs: 2.5 per 32 edges at 3dpf) from L2 instead of 2.0 MB.
// Summation order of a message element: G runs in edge order, then the stream tiles in feature order, then (blocks with several
// features per tile) the lane groups in order: fixed, bitwise reproducible; within fp32 rounding of ddp_conv_messages.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#undef DDP_STAMPS   // (the in-kernel stamps of tools/stamp_conv.py belong to ddp_conv.hip)
#include "ddp_conv_common.h"

#define ROWS_NW 8
#define ROWS_NT 512
#define ROWS_ET 256
#define ROWS_FS 36     // floats per feature row F[u * C + c][edge]
#ifndef DDP_ROWS_GRING
#define DDP_ROWS_GRING 4
#endif

struct RowsLaunch {
  ConvLaunch L;
  int nts;         // stream tiles per conv (fc.0 tiles + fc.3 tiles of all segments)
  int priv_bytes;  // LDS bytes of a wave's private area
  int aux_off;     // byte offset of the per-edge tables inside it
};

static_assert(sizeof(ConvLaunch) + 16 <= 4096, "the launch descriptor travels as a kernel argument");
// per-edge tables of a wave (behind its feature rows)
struct RowsAux {
  float shT[4][32];   // harmonics, component-major: the "feature rows" of the factorised features
  float sh[32][4];    // ... edge-major (build_features)
  int src[32], pos[32], rid[32];
};

// out[c][i] += F[(u * C + c)][row_i] * (am[i] + ac[i] / 2048), rows in the MFMA C/D layout: reg i <-> row (i & 3) + 8 (i >> 2) + 4 hh
template <int C>
__device__ __forceinline__ void rows_epilogue(const f32x16& am, const f32x16& ac, const float* frow, int cstride, f32x16* out) {
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    float tq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tq[q] = am[4 * q4 + q] + ac[4 * q4 + q] * DDP_H2_INV;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * cstride + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] += f[q] * tq[q];
    }
  }
}

// the same with the rows outside run `run` masked out (G tiles: the B operand was ONE source node's G)
template <int C>
__device__ __forceinline__ void rows_epilogue_run(const f32x16& am, const f32x16& ac, const float* frow, const int* rid, int run, f32x16* out) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const i32x4 id = *reinterpret_cast<const i32x4*>(rid + 8 * q4);
    float tq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tq[q] = (id[q] == run) ? am[4 * q4 + q] + ac[4 * q4 + q] * DDP_H2_INV : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 32 + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] += f[q] * tq[q];
    }
  }
}

// One stream step: tile t is complete in slot t & 1 and nobody reads tile t - 1 any more (barrier); the staged registers (tile
// t + 1, requested a tile ago) go to the other slot, tile t + 2 is requested.
template <int NF>
__device__ __forceinline__ void rows_stream_step(f32x4* ring, const f32x4* __restrict__ wsh, int t, int nts, int wave, int lane, f32x4 (&st)[(NF + ROWS_NW - 1) / ROWS_NW]) {
  constexpr int FPW = (NF + ROWS_NW - 1) / ROWS_NW, TILE_Q = NF * 64;
  __syncthreads();
  f32x4* nslot = ring + ((t + 1) & 1) * TILE_Q;
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wave + ROWS_NW * f < NF) nslot[(wave + ROWS_NW * f) * 64 + lane] = st[f];
  const f32x4* __restrict__ wn = wsh + (size_t)min(t + 2, nts - 1) * TILE_Q;
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wave + ROWS_NW * f < NF) st[f] = wn[(wave + ROWS_NW * f) * 64 + lane];
}

// acc += A(regs) x B(tile in LDS): 3 split products per 16 k, B fragments read one k-step ahead
template <int NS>
__device__ __forceinline__ void rows_tile_lds(const f32x4* slot, const h8 (&ah)[NS], const h8 (&al)[NS], int lane, f32x16& am, f32x16& ac) {
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

// One segment = the 32-column part `part` of block B: G runs, stream tiles, store.  Returns the stream position behind it.
template <int NS, int C>
__device__ __forceinline__ int rows_segment(const RowsLaunch& RL, const ddp_block_t& B, int part, const ddp_conv_task_t& T, const h8 (&ah)[NS],
                                            const h8 (&al)[NS], f32x4* ring, f32x4 (&st)[(2 * NS + ROWS_NW - 1) / ROWS_NW], int t, const float* F,
                                            const RowsAux* aux, unsigned rmask, int src_reg, int nvw, int wave, int lane) {
  constexpr int NF = 2 * NS, TILE_Q = NF * 64, GR = DDP_ROWS_GRING;
  static_assert(NF % GR == 0, "fragment f of every G tile lives in ring slot f % GR");
  const ddp_conv_shape_t& S = RL.L.shape;
  const int r = lane & 31, hh = lane >> 5;
  const f32x4* __restrict__ wsh = reinterpret_cast<const f32x4*>(T.wsh);
  // lane -> (output channel, feature slot) of the segment's tiles (tile_lane_map of ddp_conv.hip)
  int ncol, us;
  bool valid;
  if (B.nsub > 1) {
    ncol = 32 * part + r;
    us = 0;
    valid = ncol < B.n;
  } else {
    us = r / B.n;
    ncol = r - us * B.n;
    valid = us < B.ups;
  }
  f32x16 res[C];
#pragma unroll
  for (int c = 0; c < C; ++c) res[c] = splat16(0.f);

  // ---- factorised features: one tile product per run of edges with one source node, B = the node's G tile (plane form)
  if (B.g_slot >= 0 && rmask != 0u) {
    const int gc = S.g_cols[B.g_slot];
    const int n8 = (S.hid + 7) >> 3;
    const int nmine = min(32, B.n - 32 * part);                                   // G columns of this part
    const int gcol = B.g_col0 + 32 * part + ((r < nmine) ? r : 0);
    const size_t gld = (size_t)DDP_GH_LD(S.hid, gc) / 4;                          // node stride in 16-byte units
    const f32x4* __restrict__ G4 = reinterpret_cast<const f32x4*>(T.gh[B.g_slot]);
    const float* __restrict__ Gb = reinterpret_cast<const float*>(T.gh[B.g_slot]) + 8 * (size_t)n8 * gc;   // Gb[c] behind the planes
    // fragment q = 2 ks + plane of lane (r, hh): 16-byte unit ((2 k8 + plane) gc + gcol), k8 = min(2 ks + hh, n8 - 1)
    const int o_main = 2 * hh * gc + gcol;                                         // + (4 ks + plane) gc
    const int k8l = min(2 * (NS - 1) + hh, n8 - 1);
    const int o_last = 2 * k8l * gc + gcol;                                        // + plane gc
    unsigned m = rmask;
    int a0 = __builtin_ctz(m);
    int node = __builtin_amdgcn_readlane(src_reg, a0);
    const f32x4* __restrict__ gp = G4 + (size_t)node * gld;
    float bias = Gb[(size_t)node * (4 * gld) + gcol];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 gr[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) gr[k] = gp[((k >> 1) == NS - 1 ? o_last : o_main + 4 * (k >> 1) * gc) + (k & 1) * gc];
    __builtin_amdgcn_sched_barrier(0);
    int run = 0;
    const float* shrow = &aux->shT[(C == 1) ? 0 : 1][4 * hh];
    const int* ridrow = &aux->rid[4 * hh];
    while (m != 0u) {
      m &= m - 1u;
      const int an = (m != 0u) ? __builtin_ctz(m) : a0;
      const int node_n = __builtin_amdgcn_readlane(src_reg, an);
      const f32x4* __restrict__ gpn = G4 + (size_t)node_n * gld;
      const float bias_n = Gb[(size_t)node_n * (4 * gld) + gcol];
      f32x16 am = splat16(bias), ac = splat16(0.f);
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
        const h8 bh = __builtin_bit_cast(h8, gr[(2 * ks) % GR]), bl = __builtin_bit_cast(h8, gr[(2 * ks + 1) % GR]);
        {
          const int q0 = 2 * ks + GR;                      // the pair of fragments that takes the two slots this step frees
          const int kq = (q0 < NF) ? (q0 >> 1) : ((q0 - NF) >> 1);
          const f32x4* __restrict__ src4 = (q0 < NF) ? gp : gpn;
          const int o = (kq == NS - 1) ? o_last : o_main + 4 * kq * gc;
          gr[(2 * ks) % GR] = src4[o];
          gr[(2 * ks + 1) % GR] = src4[o + gc];
        }
        __builtin_amdgcn_sched_barrier(0);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, ac, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (lanes behind the part's last G column hold a clamped column's product: they add nothing - with several features per tile
      // their registers are summed into the first lane group at the end)
      rows_epilogue_run<C>(am, ac, shrow, ridrow, (r < nmine) ? run : -2, res);
      gp = gpn;
      bias = bias_n;
      a0 = an;
      ++run;
    }
  }

  // ---- the segment's stream tiles (vector-input features)
  const int cnt = (B.ntiles == 0 || B.U == 0) ? 0 : (B.nsub > 1 ? B.U : (B.U + B.ups - 1) / B.ups);
  if (cnt > 0) {
    float bias = T.bsp[(size_t)t * 32 + r];
    for (int j = 0; j < cnt; ++j, ++t) {
      rows_stream_step<NF>(ring, wsh, t, RL.nts, wave, lane, st);
      f32x16 am = splat16(bias), ac = splat16(0.f);
      bias = T.bsp[(size_t)min(t + 1, RL.nts - 1) * 32 + r];
      rows_tile_lds<NS>(ring + (t & 1) * TILE_Q, ah, al, lane, am, ac);
      int u = (B.nsub > 1) ? j : j * B.ups + us;
      if (!(valid && u < B.U)) u = 0;
      rows_epilogue<C>(am, ac, F + (u * C) * ROWS_FS + 4 * hh, ROWS_FS, res);
    }
  }

  // ---- several features per tile (n <= 16): the lane groups us = 1, 2, .. are added to group 0 in order
  if (B.nsub == 1 && B.ups > 1 && cnt > 0) {
    for (int s = 1; s < B.ups; ++s) {
      const int from = (hh << 5) + min(r + s * B.n, 31);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = __shfl(res[c][i], from);
          if (us == 0) res[c][i] += v;
        }
    }
  }

  // ---- the message columns of the segment
  if (valid && us == 0) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    float* __restrict__ mo = T.msg + B.out_off + ncol * C;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const i32x4 pq = *reinterpret_cast<const i32x4*>(&aux->pos[4 * hh + 8 * q4]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * hh + 8 * q4 + q < nvw) {
#pragma unroll
          for (int c = 0; c < C; ++c) mo[(size_t)pq[q] * S.d_out + c] = res[c][4 * q4 + q];
        }
    }
  }
  return t;
}

template <int SZ>
__global__ __launch_bounds__(ROWS_NT, 1) void ddp_conv_rows_kernel(const RowsLaunch RL) {
  constexpr int NS = H2Class<SZ>::NS, NF = 2 * NS, TILE_Q = NF * 64, FPW = (NF + ROWS_NW - 1) / ROWS_NW;
  constexpr int NCT1 = (3 * SZ + 31) / 32, NQ = SZ / 4;     // fc.0 column tiles; 16-byte quads per edge_attr_ segment (ns floats each)
  static_assert(NS > 0 && SZ % 4 == 0, "size classes with an h2 form");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ConvLaunch& L = RL.L;
  const ddp_conv_shape_t& S = L.shape;
  const int tid = threadIdx.x;
  int ti, p0, nvalid;
  if (!conv_tile<ROWS_ET>(L, ti, p0, nvalid)) return;
  const ddp_conv_task_t& T = L.task[ti];
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, hh = lane >> 5;
  f32x4* ring = reinterpret_cast<f32x4*>(lds);
  char* priv = reinterpret_cast<char*>(lds) + 2 * TILE_Q * 16 + (size_t)wave * RL.priv_bytes;
  const f32x4* __restrict__ wsh = reinterpret_cast<const f32x4*>(T.wsh);
  const int nvw = max(0, min(32, nvalid - 32 * wave));      // valid edges of this wave

  // ---- the wave's edges (rows behind the last valid one repeat it: every load stays in bounds, nothing of theirs is stored)
  const int pr = p0 + min(32 * wave + r, nvalid - 1);
  const int src = T.src[pr], eid = T.eid[pr];
  const int pos = T.pos ? T.pos[pr] : pr;
  const f32x4 shv = reinterpret_cast<const f32x4*>(T.sh)[eid];
  const float* __restrict__ xb0 = T.seg_ptr[0] + (size_t)T.seg_idx[0][pr] * T.seg_ld[0];
  const float* __restrict__ xb1 = T.seg_ptr[1] + (size_t)T.seg_idx[1][pr] * T.seg_ld[1];
  const float* __restrict__ xb2 = T.seg_ptr[2] + (size_t)T.seg_idx[2][pr] * T.seg_ld[2];

  // ---- stage tile 0, request tile 1
  f32x4 st[FPW];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wave + ROWS_NW * f < NF) st[f] = wsh[(wave + ROWS_NW * f) * 64 + lane];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wave + ROWS_NW * f < NF) ring[(wave + ROWS_NW * f) * 64 + lane] = st[f];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wave + ROWS_NW * f < NF) st[f] = wsh[(size_t)min(1, RL.nts - 1) * TILE_Q + (wave + ROWS_NW * f) * 64 + lane];

  // ---- edge_attr_ of the wave's edges as B-operand fragments: lane (edge r, hh) holds k = 16 ks + 8 hh + i.  hi plane in registers,
  // lo plane in the wave's private LDS area (each lane reads back what it wrote)
  h8 xh[NS];
  f32x4* xlo = reinterpret_cast<f32x4*>(priv);
  {
    f32x4 xv[NS][2];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kq = 4 * ks + 2 * hh + q;                 // quad index inside edge_attr_ = cat(seg0, seg1, seg2), NQ quads each
        const int sg = kq / NQ, off = kq - sg * NQ;
        const float* __restrict__ b = (sg == 0) ? xb0 : (sg == 1) ? xb1 : xb2;
        xv[ks][q] = (sg < 3) ? reinterpret_cast<const f32x4*>(b)[off] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      h4 h0, l0, h1, l1;
      split_h2(xv[ks][0], h0, l0);
      split_h2(xv[ks][1], h1, l1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h2_range_check(xv[ks][0][i], T.h2_range_flag);
        h2_range_check(xv[ks][1][i], T.h2_range_flag);
      }
      h8 lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[ks][i] = h0[i];
        xh[ks][4 + i] = h1[i];
        lo[i] = l0[i];
        lo[4 + i] = l1[i];
      }
      xlo[ks * 64 + lane] = __builtin_bit_cast(f32x4, lo);
    }
  }

  // ---- fc1, transposed: D[h column m][edge n] = sum_k W1[k][32 ct + m] x[n][k]; lane (edge r, hh) ends with the h columns
  // 32 ct + (j & 3) + 8 (j >> 2) + 4 hh, j < 16, of its own edge = the k-groups (2 ct, hh) and (2 ct + 1, hh) of DDP_ROWS_KPERM
  h8 ah[NS], al[NS];
  int t = 0;
#pragma unroll
  for (int ct = 0; ct < NCT1; ++ct, ++t) {
    f32x16 am, ac = splat16(0.f);
    {
      const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(T.bsp + (size_t)t * 32 + 4 * hh);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 b = bp[2 * q4];
#pragma unroll
        for (int q = 0; q < 4; ++q) am[4 * q4 + q] = b[q];
      }
    }
    rows_stream_step<NF>(ring, wsh, t, RL.nts, wave, lane, st);
    const f32x4* slot = ring + (t & 1) * TILE_Q;
    f32x4 w0 = slot[lane], w1 = slot[64 + lane];
    f32x4 xl = xlo[lane];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const h8 wh = __builtin_bit_cast(h8, w0), wl = __builtin_bit_cast(h8, w1), xlk = __builtin_bit_cast(h8, xl);
      if (ks + 1 < NS) {
        w0 = slot[(2 * ks + 2) * 64 + lane];
        w1 = slot[(2 * ks + 3) * 64 + lane];
        xl = xlo[(ks + 1) * 64 + lane];
      }
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], am, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xlk, ac, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], ac, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pre = am[j] + ac[j] * DDP_H2_INV;
      h2_range_check(pre, T.h2_range_flag);     // (before the relu: fmaxf drops a NaN)
      const float v = fmaxf(pre, 0.f);
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)((v - (float)hi) * DDP_H2_SCALE);
      constexpr int dummy = 0;
      (void)dummy;
      if (2 * ct + (j >> 3) < NS) {
        ah[2 * ct + (j >> 3)][j & 7] = hi;
        al[2 * ct + (j >> 3)][j & 7] = lo;
      }
    }
  }

  // ---- per-edge tables of the wave (the private area is free: the lo plane of edge_attr_ is dead)
  float* F = reinterpret_cast<float*>(priv);
  RowsAux* aux = reinterpret_cast<RowsAux*>(priv + RL.aux_off);
  unsigned rmask;
  {
    const int prev = __shfl_up(src, 1);
    const bool rowv = (hh == 0) && (r < nvw);
    const bool runstart = rowv && (r == 0 || src != prev);
    rmask = (unsigned)(__ballot(runstart) & 0xffffffffull);
    const unsigned upto = (r == 31) ? ~0u : ((2u << r) - 1u);
    if (hh == 0) {
      aux->src[r] = src;
      aux->pos[r] = pos;
      aux->rid[r] = rowv ? (int)__popc(rmask & upto) - 1 : -1;
      aux->sh[r][0] = shv[0]; aux->sh[r][1] = shv[1]; aux->sh[r][2] = shv[2]; aux->sh[r][3] = shv[3];
      aux->shT[0][r] = shv[0]; aux->shT[1][r] = shv[1]; aux->shT[2][r] = shv[2]; aux->shT[3][r] = shv[3];
    }
  }

  // ---- the segments: blocks in order, the 32-column parts of a block in order
  for (int bi = 0; bi < S.nblocks; ++bi) {
    const ddp_block_t& B = S.blk[bi];
    if (B.ntiles > 0 && B.U > 0) build_features<32, 2>(B, T, aux->src, aux->sh, F, lane);
    const int nparts = (B.n + 31) >> 5;
    for (int part = 0; part < nparts; ++part) {
      if (B.C == 1)
        t = rows_segment<NS, 1>(RL, B, part, T, ah, al, ring, st, t, F, aux, rmask, src, nvw, wave, lane);
      else
        t = rows_segment<NS, 3>(RL, B, part, T, ah, al, ring, st, t, F, aux, rmask, src, nvw, wave, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
extern "C" int ddp_conv_rows(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, void* stream) {
  if (!shape || !tasks) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: null argument");
  if (ntasks < 0 || ntasks > DDP_MAX_TASKS) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: ntasks > DDP_MAX_TASKS");
  if (shape->nblocks < 1 || shape->nblocks > DDP_MAX_BLOCKS) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: nblocks");
  if (shape->f_in != shape->hid || shape->hid != 180) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: shapes of the size class ns = 60 (f_in = hid = 180) only");
  RowsLaunch RL;
  ConvLaunch& L = RL.L;
  L.shape = *shape;
  L.r1_floats = 0;
  L.tv_off = 0;
  L.ntasks = 0;
  L.dev_counts = 0;
  const int NS = 12, nct1 = shape->nct1;
  if (nct1 != (shape->hid + 31) / 32) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: nct1");
  int nts = nct1, frows = 0;
  for (int b = 0; b < shape->nblocks; ++b) {
    const ddp_block_t& B = shape->blk[b];
    if (B.C != 1 && B.C != 3) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: block C must be 1 or 3");
    if (B.n < 1 || B.n > 64 || (B.C == 3 && B.n > 32)) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: block n too large");
    if (B.nsub < 1 || B.nsub > 2 || B.ups < 1 || (B.nsub > 1) != (B.n > 32) || (B.nsub == 1 && B.ups != 32 / B.n))
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: nsub / ups");
    if (B.g_slot > 1 || (B.g_slot >= 0 && (shape->g_cols[B.g_slot] < B.g_col0 + B.n)))
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: factorised block outside its G row");
    if (B.out_off < 0 || B.out_off + B.n * B.C > shape->d_out) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: block outside the message row");
    if (B.nseg < 0 || B.nseg > DDP_MAX_SEGS) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: nseg");
    if (B.ntiles > 0 && B.U > 0) {
      nts += ((B.n + 31) / 32) * (B.nsub > 1 ? B.U : (B.U + B.ups - 1) / B.ups);
      if (B.U * B.C > frows) frows = B.U * B.C;
    }
  }
  int tiles = 0;
  for (int i = 0; i < ntasks; ++i) {
    const ddp_conv_task_t& T = tasks[i];
    if (T.n_edges <= 0) continue;  // an empty conv sends no message (models/score_model.py:109-111)
    if (T.n_edges_dev) L.dev_counts = 1;
    if (!T.wsh || !T.bsp || (reinterpret_cast<size_t>(T.wsh) & 15) || (reinterpret_cast<size_t>(T.bsp) & 15))
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: task.wsh / bsp missing (or not 16-byte aligned)");
    for (int gs = 0; gs < 2; ++gs)
      if (shape->g_cols[gs] > 0 && (!T.gh[gs] || (reinterpret_cast<size_t>(T.gh[gs]) & 15)))
        return ddp_fail(DDP_EINVAL, "ddp_conv_rows: factorised shape but task.gh is null (or not 16-byte aligned)");
    // edge_attr_ = three segments of ns floats, gathered as 16-byte quads
    for (int sg = 0; sg < 3; ++sg)
      if (T.seg_n[sg] != shape->f_in / 3 || (T.seg_ld[sg] & 3) || (reinterpret_cast<size_t>(T.seg_ptr[sg]) & 15) || !T.seg_idx[sg])
        return ddp_fail(DDP_EINVAL, "ddp_conv_rows: edge_attr_ must be three 16-byte aligned segments of f_in / 3 columns");
    if (reinterpret_cast<size_t>(T.sh) & 15) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: sh must be 16-byte aligned");
    L.tile_start[L.ntasks] = tiles;
    L.task[L.ntasks] = T;
    tiles += (T.n_edges + ROWS_ET - 1) / ROWS_ET;
    ++L.ntasks;
  }
  L.tile_start[L.ntasks] = tiles;
  if (tiles == 0) return 0;
  RL.nts = nts;
  int fbytes = frows * ROWS_FS * 4;
  fbytes = (fbytes + 127) / 128 * 128;
  RL.aux_off = fbytes;
  int priv = fbytes + (int)sizeof(RowsAux);
  if (priv < NS * 1024) priv = NS * 1024;          // the lo plane of edge_attr_ during fc1
  priv = (priv + 127) / 128 * 128;
  RL.priv_bytes = priv;
  const size_t lds_bytes = (size_t)2 * (2 * NS * 1024) + (size_t)ROWS_NW * priv;
  if (lds_bytes > 160 * 1024 - 1024) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: LDS budget exceeded (too many vector features per block)");
  static int lds_have = 0;
  hipError_t err = ddp_need_lds(reinterpret_cast<const void*>(ddp_conv_rows_kernel<60>), (int)lds_bytes, &lds_have);
  if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(conv rows)");
  hipLaunchKernelGGL(ddp_conv_rows_kernel<60>, dim3(tiles), dim3(ROWS_NT), lds_bytes, (hipStream_t)stream, RL);
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_conv_rows launch");
  return 0;
}
