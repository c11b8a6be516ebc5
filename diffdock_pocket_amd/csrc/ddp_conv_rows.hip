// ddp_conv_rows.hip - the factorised conv (fc -> FasterTensorProduct -> message, csrc/ddp_conv.hip's contract) through 128-edge,
// ROW-STATIONARY workgroups for gfx950 (round 5).
//
// Replaces, per conv (reference file:line):  edge_attr_ = cat(...)  models/all_atom_score_model.py:273-312;  w = fc(edge_attr_)
// models/score_model.py:100-105,114;  msg = FasterTensorProduct(x[src], sh, w)  models/layers.py:34-85.
//
// Why: the 32-edge kernel (ddp_conv32_kernel<60, h2>) streams the whole packed fc.3 weight set (1.47 MB) and ~0.5 MB of G rows
// through every 32-edge workgroup: 278 M L2 requests of 128 B per 2.3-ms launch = 15.3 TB/s of L2 -> CU traffic, L2 busy 90 %
// (profiles/r05_pmc_conv32_h2_l1_l2.json) against the 16.8 - 18.8 TB/s the chip delivers from its XCD L2s - the matrix pipe sat at
// 31 %.  Here the weights leave L2 once per 128 edges:
//   workgroup = 4 waves (one per SIMD, 256 registers each), TWO per CU, 128 consecutive (source-ordered) edges of one conv: the two
//     co-resident workgroups are independent, so one's memory-bound G runs overlap the other's matrix-bound stream tiles (one
//     8-wave workgroup per CU ran every phase in lock-step behind its barriers: 480 k ticks per 256 edges, G runs 45 % of them);
//   wave w owns the 32 edges [32 w, 32 w + 32) for the whole kernel and keeps h = relu(fc1) of them as the A-operand fragments
//     of v_mfma_f32_32x32x16_f16 in REGISTERS (fp16 hi/lo planes: 8 NS registers);
//   every operand is a pair of UNIFIED fp16 planes (V = v 2^s = hi + lo, both halves at one scale; DDP_ROWS_S* of include/ddp_hip.h):
//     the three split products of a k-step accumulate into ONE register tile (the form v = hi + lo / 2048 of the other kernels needs
//     two, and a multiply-add per element to join them);
//   the weight tiles (task.wsh: the fc.0 tiles, then the fc.3 tiles segment by segment) are staged once per workgroup through a
//     three-slot LDS ring of THIRD tiles (8 KiB each) by LDS-DMA (buffer loads to LDS) - every wave moves 1/4 of the piece after
//     next, one BARE barrier per piece - and all four waves read their B operands from LDS (conflict-free lane-linear 16-byte reads);
//   h comes from the TRANSPOSED fc1 product (A = fc.0 tile, B = edge_attr_ fragments gathered straight from the three row
//     segments): the accumulator of column tile ct leaves lane (edge, hh) with 16 h values of its own edge, which ARE the k-groups
//     (2 ct, hh) and (2 ct + 1, hh) of the next products in the permuted k order DDP_ROWS_KPERM (the host packs fc.3 and G in it);
//   a segment = one 32-column part of one weight block's output columns: a wave accumulates EVERYTHING that lands there in
//     registers - first the factorised features (one pass of the same tile product per run of edges with one source node, B = the
//     node's G tile in plane form, task.gh, straight from memory through a register ring: 12 fragments by buffer loads for the
//     scalar segments, 8 by global loads for the merged pair of vector blocks; the rows of a run are selected from its product),
//     then the segment's stream tiles (the vector-input features) - and stores the message columns; no message tile in LDS, no
//     cross-wave reduction, no atomics.
// Per 32 edges: 0.18 MB of weights + ~0.26 MB of G (whole runs: 2.5 per 32 edges at 3dpf) from L2 instead of 2.0 MB.
// Summation order of a message element: G runs in edge order, then the stream tiles in feature order, then (blocks with several
// features per tile) the lane groups in order: fixed, bitwise reproducible; within fp32 rounding of ddp_conv_messages.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#undef DDP_STAMPS   // (the in-kernel stamps of tools/stamp_conv.py belong to ddp_conv.hip)
#include "ddp_conv_common.h"

#define ROWS_NW 4
#define ROWS_NT 256
#define ROWS_ET 128
#define ROWS_FS 36     // floats per feature row F[u * C + c][edge]
#ifndef DDP_ROWS_GRING1
#define DDP_ROWS_GRING1 12
#endif
#ifndef DDP_ROWS_GRING3
#define DDP_ROWS_GRING3 8
#endif

// Operand planes of this kernel ("unified" fp16 hi/lo planes, include/ddp_hip.h DDP_ROWS_S*): V = v * 2^s = hi + lo with hi = fp16(V),
// lo = fp16(V - hi) at the SAME scale, so that the three split products hh wh + hh wl + hl wh land in ONE accumulator (the common form
// v = hi + lo / 2048 of ddp_conv_common.h needs two and a multiply-add per element to join them: 32 registers of every tile product
// here).  22 significant bits while lo is a normal fp16 number (|V| >= 0.125), an absolute 2^-25 / 2^s below; |V| <= 65504 or the
// range flag is raised.  The accumulators carry 2^(sa + sb); the feature rows / harmonics they are multiplied with carry the inverse.
#define ROWS_SX ((float)DDP_ROWS_SX)
#define ROWS_SW ((float)DDP_ROWS_SW)
#define ROWS_SH ((float)DDP_ROWS_SH)
#define ROWS_SG ((float)DDP_ROWS_SG)
__device__ __forceinline__ void rows_split(const f32x4 v, float scale, h4& hi, h4& lo, int32_t* flag) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float V = v[i] * scale;
    h2_range_check(V, flag);
    hi[i] = (_Float16)V;
    lo[i] = (_Float16)(V - (float)hi[i]);
  }
}

// Diagnostic build only (-DDDP_ROWS_STAMPS, tools/stamp_rows.py): lane 0 of every wave records s_memtime at the phase boundaries
#ifdef DDP_ROWS_STAMPS
#define RS_SLOTS 32
#define RS_WGS 16384
__device__ unsigned long long ddp_rows_stamp_buf[(size_t)RS_WGS * ROWS_NW * RS_SLOTS];
#define RSTAMP(k)                                                                                          \
  do {                                                                                                     \
    unsigned long long t_;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (lane == 0 && blockIdx.x < RS_WGS) ddp_rows_stamp_buf[((size_t)blockIdx.x * ROWS_NW + wave) * RS_SLOTS + (k)] = t_; \
  } while (0)
#define RSTAMP_VAL(k, v)                                                                                   \
  do {                                                                                                     \
    if (lane == 0 && blockIdx.x < RS_WGS) ddp_rows_stamp_buf[((size_t)blockIdx.x * ROWS_NW + wave) * RS_SLOTS + (k)] = (unsigned long long)(v); \
  } while (0)
extern "C" int ddp_debug_read_rows_stamps(unsigned long long* host_dst, int n_wgs) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ddp_rows_stamp_buf), sizeof(unsigned long long) * RS_SLOTS * ROWS_NW * (size_t)n_wgs);
}
#else
#define RSTAMP(k) do {} while (0)
#define RSTAMP_VAL(k, v) do {} while (0)
#endif

struct RowsLaunch {
  ConvLaunch L;
  int nts;         // stream tiles per conv (fc.0 tiles + fc.3 tiles of all segments)
  int bias_bytes;  // LDS bytes of the bias table behind the ring
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

// out[c][i] += F[(u * C + c)][row_i] * acc[i], rows in the MFMA C/D layout: reg i <-> row (i & 3) + 8 (i >> 2) + 4 hh
// (acc carries the operands' plane scales; the feature rows carry their inverse)
template <int C>
__device__ __forceinline__ void rows_epilogue(const f32x16& acc, const float* frow, int cstride, f32x16* out) {
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * cstride + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] += f[q] * acc[4 * q4 + q];
    }
  }
}

// G tiles: every row of the wave's tile belongs to exactly ONE run, so the runs' tile products are only SELECTED into tg[row] = the
// product of the row's own run (16 registers whatever C) and multiplied by the harmonics once, behind the last run
__device__ __forceinline__ void rows_select_run(const f32x16& acc, const int* rid, int run, f32x16& tg) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const i32x4 id = *reinterpret_cast<const i32x4*>(rid + 8 * q4);
#pragma unroll
    for (int q = 0; q < 4; ++q) tg[4 * q4 + q] = (id[q] == run) ? acc[4 * q4 + q] : tg[4 * q4 + q];
  }
}
template <int C>
__device__ __forceinline__ void rows_apply_harmonics(const f32x16& tg, const float* frow, f32x16* out) {
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * 32 + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[c][4 * q4 + q] = f[q] * tg[4 * q4 + q];
    }
}

// The weight stream: a tile travels as ROWS_NP pieces of NS / ROWS_NP k-steps (8 KiB at NS = 12), piece p of every tile through slot p of
// a three-slot LDS ring.  One stream step j = ROWS_NP t + p: every wave's part of piece j has landed (the wave waits for its own LDS-DMA
// copies of piece j - those of piece j + 1 stay in flight - then the barrier), nobody reads piece j - 1 any more, so piece j + 2 is
// requested into its slot (global_load_lds_dwordx4: no staging registers; every wave moves 2 NS / (ROWS_NP ROWS_NW) fragments of 1 KiB,
// lane-linear in LDS).  A copy has two piece products to land (one was not enough: ~1 k ticks of every 3.4 k-tick tile waited for it).
#define ROWS_NP 3
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
// (the copies are BUFFER loads to LDS, not global_load_lds: hipcc books a global_load_lds as a flat access to both address spaces, and while
// one is pending every wait for an ordinary load becomes vmcnt(0))
typedef __amdgpu_buffer_rsrc_t RowsStream;
__device__ __forceinline__ RowsStream rows_stream_of(const void* wsh, int nts, int tile_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wsh), 0, nts * tile_bytes, 0x00020000);
}
template <int NS>
__device__ __forceinline__ void rows_request_piece(f32x4* ring, RowsStream wsh, int jn, int npieces, int slot, int wave, int lane) {
  constexpr int FPP = 2 * NS / ROWS_NP, FPW = FPP / ROWS_NW, PIECE_Q = FPP * 64;
  static_assert(NS % ROWS_NP == 0 && FPP % ROWS_NW == 0, "every wave moves the same number of fragments per piece");
  f32x4* nslot = ring + slot * PIECE_Q;
  const int piece_off = min(jn, npieces - 1) * (PIECE_Q * 16);
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsh, (lds_ptr_t)(nslot + (wave + ROWS_NW * f) * 64), 16, ((wave + ROWS_NW * f) * 64 + lane) * 16, piece_off, 0, 0);
}
template <int NS, int P>
__device__ __forceinline__ void rows_stream_step(f32x4* ring, RowsStream wsh, int t, int nts, int wave, int lane) {
  constexpr int FPW = 2 * NS / ROWS_NP / ROWS_NW;
  // (hipcc does NOT wait for an LDS-DMA in front of a barrier: without this a wave can pass while its part of the piece is in flight.
  // vmcnt counts in order: "at most FPW outstanding" = everything older than the copies of piece j + 1 has landed)
  static_assert(FPW == 2 || FPW == 1, "the literals below");
  if constexpr (FPW == 2)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  // the bare barrier, not __syncthreads(): its workgroup fence makes hipcc wait vmcnt(0) whenever an ordinary load is in flight.  What
  // the barrier orders here is LDS only: this wave's reads of the slot that is requested next (and, once, the bias table's writes) are
  // complete (lgkmcnt(0)), the copies it waits for are counted above; the asm statements keep the compiler from moving LDS accesses
  // across it.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  rows_request_piece<NS>(ring, wsh, ROWS_NP * t + P + 2, ROWS_NP * nts, (P + 2) % ROWS_NP, wave, lane);
}

// acc += A(regs, k-steps KS0 ..) x B(piece in LDS): 3 split products per 16 k on ONE accumulator (unified planes), B fragments read one
// k-step ahead (two or four k-steps ahead: the same step time, 18.15 ms)
template <int NS, int KS0>
__device__ __forceinline__ void rows_piece_lds(const f32x4* slot, const h8 (&ah)[NS], const h8 (&al)[NS], int lane, f32x16& acc) {
  constexpr int NK = NS / ROWS_NP;
  f32x4 b0 = slot[lane], b1 = slot[64 + lane];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const h8 bh = __builtin_bit_cast(h8, b0), bl = __builtin_bit_cast(h8, b1);
    if (k + 1 < NK) {
      b0 = slot[(2 * k + 2) * 64 + lane];
      b1 = slot[(2 * k + 3) * 64 + lane];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[KS0 + k], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[KS0 + k], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[KS0 + k], bh, acc, 0, 0, 0);
  }
}

// The basis features of a block's vector-input segments (DOT, VEC_S0, CROSS; build_features of ddp_conv_common.h restated for one wave):
// ALL loads of a segment first - clamped, unconditional - then the arithmetic.  build_features issues one load per feature inside a
// runtime loop, and with the stream's LDS-DMA copies in flight hipcc waits vmcnt(0) at every use: ~700 ticks per feature, 6 - 13 k per block.
template <int MAXI>
__device__ __forceinline__ void rows_build_features(const ddp_block_t& B, const ddp_conv_task_t& T, const RowsAux* aux, float* F, int lane) {
  constexpr int FS = ROWS_FS;
  const int e = lane & 31, half = lane >> 5;
  const float* __restrict__ xrow = T.x_src + (size_t)aux->src[e] * T.ldx_src;
  // (aux->sh carries 1 / (DDP_ROWS_SH DDP_ROWS_SW): the stream tiles' accumulators carry the planes' scales)
  const float s0 = aux->sh[e][0], sx = aux->sh[e][1], sy = aux->sh[e][2], sz = aux->sh[e][3];
  const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt2 = 0.70710678118654752f;
  int ubase = 0;
  for (int si = 0; si < B.nseg; ++si) {
    const int kind = B.seg[si].kind, off = B.seg[si].in_off, cnt = B.seg[si].count;
    float ax[MAXI], ay[MAXI], az[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int ul = max(min(half + 2 * i, cnt - 1), 0);      // (an empty segment: the load stays inside the row, nothing is stored)
      ax[i] = xrow[off + 3 * ul];
      ay[i] = xrow[off + 3 * ul + 1];
      az[i] = xrow[off + 3 * ul + 2];
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int ul = half + 2 * i;
      if (ul < cnt) {
        const int u = ubase + ul;
        if (kind == DDP_F_DOT) {
          F[u * FS + e] = (ax[i] * sx + ay[i] * sy + az[i] * sz) * inv_sqrt3;
        } else if (kind == DDP_F_VEC_S0) {
          F[(u * 3 + 0) * FS + e] = ax[i] * s0;
          F[(u * 3 + 1) * FS + e] = ay[i] * s0;
          F[(u * 3 + 2) * FS + e] = az[i] * s0;
        } else {  // DDP_F_CROSS: a x s1 / sqrt(2)
          F[(u * 3 + 0) * FS + e] = (ay[i] * sz - az[i] * sy) * inv_sqrt2;
          F[(u * 3 + 1) * FS + e] = (az[i] * sx - ax[i] * sz) * inv_sqrt2;
          F[(u * 3 + 2) * FS + e] = (ax[i] * sy - ay[i] * sx) * inv_sqrt2;
        }
      }
    }
    ubase += cnt;
  }
}

// Where the G tile of segment (block bi, part) sits inside a node's row of task.gh[slot]: byte offset of the tile, padded width, padded
// columns of the slot and in front of the part (include/ddp_hip.h, ddp_conv_task_t::gh).  Returns false if the block has no G part.
__device__ __forceinline__ bool rows_gpart(const ddp_conv_shape_t& S, int bi, int part, int& wp, int& cumw, int& gcp) {
  const ddp_block_t& B = S.blk[bi];
  wp = cumw = gcp = 0;
  if (B.g_slot < 0) return false;
  for (int bj = 0; bj < S.nblocks; ++bj) {
    const ddp_block_t& Bj = S.blk[bj];
    if (Bj.g_slot != B.g_slot) continue;
    for (int pj = 0; pj < ((Bj.n + 31) >> 5); ++pj) {
      const int wj = (min(32, Bj.n - 32 * pj) + 3) & ~3;
      if (bj < bi || (bj == bi && pj < part)) cumw += wj;
      if (bj == bi && pj == part) wp = wj;
      gcp += wj;
    }
  }
  return true;
}

// The G runs of one segment: per run of edges with one source node ONE tile product h[32 x 16 NS] @ G[node][16 NS x 32], B = the node's G
// tile (plane form, task.gh) straight from memory through a register ring of GR fragments; every row belongs to exactly one run, so the
// products are SELECTED into tg[row] (rows_select_run).  Returns tg (0 in the lanes behind the tile's last column).
// G of a source node and slot: the column parts of the slot's blocks one after the other, each a CONTIGUOUS tile [k8][wp columns][plane]
// [8 halves] (wp = the part's width rounded up to 4), then Gb per padded column - a run reads one tile as one linear stream, and stage A
// fills it in whole 128-byte lines.  Fragment q = 2 ks + plane of lane (r, hh): 16-byte unit 2 (k8 wp + column) + plane, k8 = min(2 ks +
// hh, n8 - 1), addressed as (node base + uniform fragment offset) + a 32-bit per-lane offset (per-fragment 64-bit lane addresses cost ~40
// registers).
// MERGE (two vector blocks of the same width n <= 16 with one G part each, e.g. 1o and 1e at nv = 10): lanes [0, n) take block A's
// columns, lanes [n, 2 n) block B's (another G array, the same node): ONE tile product per run for both - in block B's own tiles lane
// n + j is feature group 1 of output column j, i.e. where its lane-group sum expects that column.
struct RowsGPart {
  const char* base;      // the part's tile inside node 0's row of its G array
  size_t gldb;           // node stride in bytes
  int wp, nmine, bias_off;   // padded width, columns, byte offset of Gb[column 0] (plane form 1: of the Gb region) from `base`
  int cumw;              // padded columns of the slot in front of the part
};
// Plane form 1 of a G array (ddp_conv_task_t::gh_fmt = 1, round 6: "G3"; ABI 17): the unit (k8, c) of a part's tile is 24 bytes - 8 fp16 hi
// words (V truncated), then 8 continuation bytes (19 significant bits; include/ddp_hip.h) - instead of 32; Gb per padded column c of the
// slot sits behind the units of all parts in 24-byte groups of six fp32: 24 (c / 6) + 4 (c % 6) bytes.
template <int GF>
__device__ __forceinline__ RowsGPart rows_gpart_of(const ddp_conv_shape_t& S, const ddp_conv_task_t& T, int bi, int part) {
  RowsGPart P;
  int wp, cumw, gcp;
  rows_gpart(S, bi, part, wp, cumw, gcp);
  const int n8 = (S.hid + 7) >> 3;
  if constexpr (GF == 1) {
    P.base = reinterpret_cast<const char*>(T.gh[S.blk[bi].g_slot]) + (size_t)(n8 * cumw) * 24;
    P.gldb = (size_t)DDP_GH3_LD(S.hid, gcp) * 4;
    P.bias_off = n8 * (gcp - cumw) * 24;
  } else {
    P.base = reinterpret_cast<const char*>(T.gh[S.blk[bi].g_slot]) + (size_t)(2 * n8 * cumw) * 16;
    P.gldb = (size_t)DDP_GH_LD(S.hid, gcp) * 4;
    P.bias_off = (8 * n8 * gcp + cumw) * 4 - (2 * n8 * cumw) * 16;
  }
  P.wp = wp;
  P.nmine = min(32, S.blk[bi].n - 32 * part);
  P.cumw = cumw;
  return P;
}
// per-lane byte offsets of a G part's fragments and of its Gb word (cl = the lane's column of the part)
template <int GF>
struct RowsGLane {
  unsigned h_main, h_last;    // hi (plane form 0: hi and lo) fragment of k-step kq < NS - 1: + fragment offset; of the last k-step
  unsigned l_main, l_last;    // plane form 1: the 8-byte lo pieces
  unsigned bias;
};
template <int GF>
__device__ __forceinline__ RowsGLane<GF> rows_glane(const RowsGPart& P, int n8, int NS, int hh, int cl) {
  RowsGLane<GF> o;
  const int gc = P.wp;
  const int k8l = min(2 * (NS - 1) + hh, n8 - 1);
  if constexpr (GF == 1) {
    o.h_main = (unsigned)(hh * gc + cl) * 24u;
    o.h_last = (unsigned)(k8l * gc + cl) * 24u;
    o.l_main = o.h_main + 16u;
    o.l_last = o.h_last + 16u;
    const int c = P.cumw + cl;
    o.bias = (unsigned)(P.bias_off + 24 * (c / 6) + 4 * (c % 6));
  } else {
    o.h_main = (unsigned)(2 * hh * gc + 2 * cl) * 16u;
    o.h_last = (unsigned)(2 * k8l * gc + 2 * cl) * 16u;
    o.l_main = o.h_main + 16u;
    o.l_last = o.h_last + 16u;
    o.bias = (unsigned)(P.bias_off + 4 * cl);
  }
  return o;
}
// uniform byte offset of k-step kq's fragments inside a part's tile (the last k-step is addressed by the lane offsets alone)
template <int GF>
__device__ __forceinline__ constexpr unsigned rows_gfrag_hi(int kq, int NS, int gc) {
  return (kq == NS - 1) ? 0u : (GF == 1 ? (unsigned)(2 * kq * gc) * 24u : (unsigned)(4 * kq * gc) * 16u);
}
template <int GF>
__device__ __forceinline__ constexpr unsigned rows_gfrag_lo(int kq, int NS, int gc) {
  return (kq == NS - 1) ? 0u : (GF == 1 ? (unsigned)(2 * kq * gc) * 24u : (unsigned)(4 * kq * gc) * 16u);
}
// the lo plane of a fragment as the B operand: plane form 0 holds the 8 fp16 words, plane form 1 eight continuation bytes - lo = sign(hi)
// 2^E(hi) u8 / 2^18: the byte in the mantissa of 2^-8 (0x1C00 | u8), minus 2^-8, times the hi word's sign-and-exponent bits; four packed
// instructions per pair of values (csrc/ddp_conv_rows16.hip, r16_lo_of: the same)
typedef unsigned int f32x2r __attribute__((ext_vector_type(2)));      // (eight continuation bytes: two words, never used as floats)
template <int GF>
struct RowsLoT { typedef f32x4 type; };
template <>
struct RowsLoT<1> { typedef f32x2r type; };
template <int GF>
__device__ __forceinline__ h8 rows_lo_operand(const typename RowsLoT<GF>::type v, const f32x4 hi) {
  if constexpr (GF == 1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
    const u32x4r hw = __builtin_bit_cast(u32x4r, hi);
    const h2 c = {(_Float16)0.00390625f, (_Float16)0.00390625f};
    u32x4r out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned x = __builtin_amdgcn_perm(0x1c1c1c1cu, v[i >> 1], (i & 1) ? 0x07030602u : 0x05010400u);
      const h2 y = __builtin_bit_cast(h2, x) - c;
      out[i] = __builtin_bit_cast(unsigned, y * __builtin_bit_cast(h2, hw[i] & 0xfc00fc00u));
    }
    return __builtin_bit_cast(h8, out);
  } else {
    return __builtin_bit_cast(h8, v);
  }
}
template <int NS, int GR, bool MERGE, int G3>
__device__ __forceinline__ f32x16 rows_g_runs(const ddp_conv_shape_t& S, const RowsGPart& PA, const RowsGPart& PB, const h8 (&ah)[NS], const h8 (&al)[NS],
                                              const RowsAux* aux, unsigned rmask, int src_reg, int lane) {
  constexpr int NF = 2 * NS, GK = GR / 2;      // the ring holds GK k-steps: a hi and a lo fragment each
  static_assert(NF % GR == 0 && GR % 2 == 0, "k-step ks of every G tile lives in ring slot ks % (GR / 2)");
  typedef typename RowsLoT<G3>::type lo_t;
  const int r = lane & 31, hh = lane >> 5;
  const int n8 = (S.hid + 7) >> 3;
  const int ncols = MERGE ? PA.nmine + PB.nmine : PA.nmine;
  const bool inb = MERGE && r >= PA.nmine;                                          // this lane reads block B's array
  const int cl = (r < ncols) ? (inb ? r - PA.nmine : r) : 0;
  const int gc = PA.wp;                                                             // (MERGE: the same for both parts)
  const RowsGLane<G3> LA = rows_glane<G3>(PA, n8, NS, hh, cl), LB = rows_glane<G3>(PB, n8, NS, hh, cl);
  const unsigned lo_bias = inb ? LB.bias : LA.bias;
  const unsigned h_main = LA.h_main, h_last = LA.h_last, l_main = LA.l_main, l_last = LA.l_last;     // (the same for both parts: one width)
#define ROWS_NODE(a) __builtin_amdgcn_readlane(src_reg, (a))
  // node base of this lane's array: wave-uniform without MERGE (SGPRs), a per-lane select of two uniform bases with it
#define ROWS_GBASE(a) (inb ? PB.base + (size_t)ROWS_NODE(a) * PB.gldb : PA.base + (size_t)ROWS_NODE(a) * PA.gldb)
#define ROWS_GFRAG_HI(base, kq) (*reinterpret_cast<const f32x4*>((base) + rows_gfrag_hi<G3>((kq), NS, gc) + (((kq) == NS - 1) ? h_last : h_main)))
#define ROWS_GFRAG_LO(base, kq) (*reinterpret_cast<const lo_t*>((base) + rows_gfrag_lo<G3>((kq), NS, gc) + (((kq) == NS - 1) ? l_last : l_main)))
  unsigned m = rmask;
  int a0 = __builtin_ctz(m);
  const char* __restrict__ gp = ROWS_GBASE(a0);
  float bias = *reinterpret_cast<const float*>(gp + lo_bias);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 grh[GK];
  lo_t grl[GK];
#pragma unroll
  for (int k = 0; k < GK; ++k) {
    grh[k] = ROWS_GFRAG_HI(gp, k);
    grl[k] = ROWS_GFRAG_LO(gp, k);
  }
  __builtin_amdgcn_sched_barrier(0);
  int run = 0;
  const int* ridrow = &aux->rid[4 * hh];
  f32x16 tg = splat16(0.f);
  while (m != 0u) {
    m &= m - 1u;
    const int an = (m != 0u) ? __builtin_ctz(m) : a0;
    const char* __restrict__ gpn = ROWS_GBASE(an);
    const float bias_n = *reinterpret_cast<const float*>(gpn + lo_bias);
    f32x16 acc = splat16(bias);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const h8 bh = __builtin_bit_cast(h8, grh[ks % GK]), bl = rows_lo_operand<G3>(grl[ks % GK], grh[ks % GK]);
      {
        const int q0 = ks + GK;                          // the k-step that takes the slot this step frees
        const int kq = (q0 < NS) ? q0 : q0 - NS;
        const char* __restrict__ srcb = (q0 < NS) ? gp : gpn;
        grh[ks % GK] = ROWS_GFRAG_HI(srcb, kq);
        grl[ks % GK] = ROWS_GFRAG_LO(srcb, kq);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (lanes behind the tile's last G column hold a clamped column's product: they select nothing)
    rows_select_run(acc, ridrow, (r < ncols) ? run : -2, tg);
    gp = gpn;
    bias = bias_n;
    a0 = an;
    ++run;
  }
#undef ROWS_GFRAG_HI
#undef ROWS_GFRAG_LO
#undef ROWS_GBASE
#undef ROWS_NODE
  return tg;
}

// The G runs of a SCALAR segment (C = 1, one G array): like rows_g_runs, but
//  * the fragments are BUFFER loads - descriptor of the node's tile (4 SGPRs, rebuilt per run) + the fragment's uniform offset (an SGPR) +
//    ONE lane offset register; as global loads hipcc kept a 64-bit lane address per fragment of the run in registers (24 of them);
//  * the register ring holds half a tile (DDP_ROWS_GRING1 = 12 fragments = 6 k-steps, 12 KiB in flight per wave: the unified planes freed an
//    accumulator per product): while run i is multiplied, run i + 1 arrives in the slots its steps free - a run costs the larger of
//    its 36 MFMAs and ONE memory round trip, not three;
//  * the last run's steps load from an EMPTY buffer (num_records = 0: the loads return zeros without touching memory) instead of
//    prefetching a tile nobody reads;
//  * a run's product goes straight into the segment's accumulator (rows of the run selected, times the harmonic), no tg registers.
struct RowsGSeq {        // (scalars only: the ring and the accumulators are separate locals, so that everything stays in registers)
  RowsStream rs;         // the current run's node
  RowsStream rsn;        // the next run's (behind the last run: an empty buffer)
  const char* base;
  size_t gldb;
  float bias;            // Gb of the current run's column (added when the run's product is added to the accumulator)
  unsigned m;            // runs not yet started (bit = first row)
  int run, nruns;
  unsigned h_main, h_last, l_main, l_last, lo_bias;
  int gc;
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int G3, int NS>
__device__ __forceinline__ f32x4 rows_gseq_hi(const RowsGSeq& G, RowsStream R, int kq) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R, (kq == NS - 1) ? G.h_last : G.h_main, (int)rows_gfrag_hi<G3>(kq, NS, G.gc), 0));
}
template <int G3, int NS>
__device__ __forceinline__ typename RowsLoT<G3>::type rows_gseq_lo(const RowsGSeq& G, RowsStream R, int kq) {
  if constexpr (G3 == 1)
    return __builtin_bit_cast(f32x2r, __builtin_amdgcn_raw_buffer_load_b64(R, (kq == NS - 1) ? G.l_last : G.l_main, (int)rows_gfrag_lo<G3>(kq, NS, G.gc), 0));
  else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R, (kq == NS - 1) ? G.l_last : G.l_main, (int)rows_gfrag_lo<G3>(kq, NS, G.gc), 0));
}
__device__ __forceinline__ RowsStream rows_gseq_node(const RowsGSeq& G, int src_reg, int row) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(G.base + (size_t)__builtin_amdgcn_readlane(src_reg, row) * G.gldb), 0, (int)G.gldb, 0x00020000);
}
// a run starts: its bias word is requested now and read when the run's last k-step is done (a load whose value is needed at once is the
// youngest operation in flight: waiting for it drains the whole in-order queue)
__device__ __forceinline__ void rows_gseq_next(RowsGSeq& G, int src_reg, f32x16& gacc) {
  G.m &= G.m - 1u;
  G.rsn = (G.m != 0u) ? rows_gseq_node(G, src_reg, __builtin_ctz(G.m)) : __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(G.base), 0, 0, 0x00020000);
  G.bias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(G.rs, G.lo_bias, 0, 0));
  gacc = splat16(0.f);
}
template <int NS, int GK, int G3>
__device__ __forceinline__ void rows_gseq_init(RowsGSeq& G, f32x4 (&grh)[GK], typename RowsLoT<G3>::type (&grl)[GK], f32x16& gacc,
                                               const ddp_conv_shape_t& S, const RowsGPart& PA, unsigned rmask, int src_reg, int lane) {
  const int r = lane & 31, hh = lane >> 5;
  const int n8 = (S.hid + 7) >> 3;
  const int cl = (r < PA.nmine) ? r : 0;
  G.gc = PA.wp;
  const RowsGLane<G3> LA = rows_glane<G3>(PA, n8, NS, hh, cl);
  G.h_main = LA.h_main;
  G.h_last = LA.h_last;
  G.l_main = LA.l_main;
  G.l_last = LA.l_last;
  G.lo_bias = LA.bias;
  G.base = PA.base;
  G.gldb = PA.gldb;
  G.m = rmask;
  G.nruns = __builtin_amdgcn_readfirstlane(__popc(rmask));
  G.run = 0;
  G.rs = rows_gseq_node(G, src_reg, __builtin_ctz(rmask));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < GK; ++k) {
    grh[k] = rows_gseq_hi<G3, NS>(G, G.rs, k);
    grl[k] = rows_gseq_lo<G3, NS>(G, G.rs, k);
  }
  __builtin_amdgcn_sched_barrier(0);
  rows_gseq_next(G, src_reg, gacc);
}
// k-step KS of the current run: three split products from ring slot KS % GK, which then takes the fragments GK k-steps on
template <int NS, int GK, int KS, int G3>
__device__ __forceinline__ void rows_gseq_step(RowsGSeq& G, f32x4 (&grh)[GK], typename RowsLoT<G3>::type (&grl)[GK], f32x16& gacc, const h8 (&ah)[NS],
                                               const h8 (&al)[NS]) {
  constexpr int q0 = KS + GK, kq = (q0 < NS) ? q0 : q0 - NS;
  __builtin_amdgcn_sched_barrier(0);
  {
    const h8 bh = __builtin_bit_cast(h8, grh[KS % GK]), bl = rows_lo_operand<G3>(grl[KS % GK], grh[KS % GK]);
    gacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[KS], bh, gacc, 0, 0, 0);
    gacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[KS], bl, gacc, 0, 0, 0);
    gacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[KS], bh, gacc, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const RowsStream srcb = (q0 < NS) ? G.rs : G.rsn;
  grh[KS % GK] = rows_gseq_hi<G3, NS>(G, srcb, kq);
  grl[KS % GK] = rows_gseq_lo<G3, NS>(G, srcb, kq);
  __builtin_amdgcn_sched_barrier(0);
}
// the run's last k-step is done: res[row] += sh0[row] * product for the rows of THIS run (lanes behind the tile's last column: nothing)
__device__ __forceinline__ void rows_gseq_finish(RowsGSeq& G, f32x16& gacc, const int* ridrow, const float* shrow, bool mine, int src_reg,
                                                 f32x16& res) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  {
    const int sel = mine ? G.run : -2;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const i32x4 id = *reinterpret_cast<const i32x4*>(ridrow + 8 * q4);
      const f32x4 f = *reinterpret_cast<const f32x4*>(shrow + 8 * q4);
#pragma unroll
      for (int q = 0; q < 4; ++q) res[4 * q4 + q] = (id[q] == sel) ? res[4 * q4 + q] + f[q] * (gacc[4 * q4 + q] + G.bias) : res[4 * q4 + q];
    }
  }
  G.rs = G.rsn;
  ++G.run;
  rows_gseq_next(G, src_reg, gacc);
}

// gmode: 0 = the segment runs its own G tiles; 1 = it runs the MERGED tiles of its block and the next one (block B's products are
// parked behind the wave's per-edge tables); 2 = its G products were computed by the segment before (lanes [n, 2 n))
template <int NS, int C, int G3>
__device__ __forceinline__ int rows_segment(const RowsLaunch& RL, const ddp_block_t& B, int bi, int part, const ddp_conv_task_t& T, const h8 (&ah)[NS],
                                            const h8 (&al)[NS], f32x4* ring, const float* lbias, int t, const float* F,
                                            const RowsAux* aux, unsigned rmask, int src_reg, int nvw, int wave, int lane, int sgi,
                                            int gmode) {
  constexpr int NF = 2 * NS, GRW = (C == 1) ? DDP_ROWS_GRING1 : DDP_ROWS_GRING3;
  constexpr int GR = (NF % GRW == 0) ? GRW : (NF % 6 == 0) ? 6 : (NF % 4 == 0) ? 4 : 2;
  (void)sgi;
  const ddp_conv_shape_t& S = RL.L.shape;
  const int r = lane & 31, hh = lane >> 5;
  const RowsStream wsh = rows_stream_of(T.wsh, RL.nts, 2 * NS * 1024);
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

  // ---- scalar segment with factorised features: whole-tile ring, buffer loads (RowsGSeq)
  bool gdone = false;
  if constexpr (C == 1) {
    if (B.g_slot >= 0 && rmask != 0u) {
      constexpr int GK1 = ((DDP_ROWS_GRING1 >= NF) ? NF : GR) / 2;      // k-steps in the ring
      const RowsGPart PA = rows_gpart_of<G3>(S, T, bi, part);
      RowsGSeq G;
      f32x4 grh[GK1];
      typename RowsLoT<G3>::type grl[GK1];
      f32x16 gacc;
      rows_gseq_init<NS, GK1, G3>(G, grh, grl, gacc, S, PA, rmask, src_reg, lane);
      const int* ridrow = &aux->rid[4 * hh];
      const float* shrow = &aux->shT[0][4 * hh];
      const bool mine = r < PA.nmine;
      while (G.run < G.nruns) {
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          // (static k-steps: the chain is resolved at compile time)
          if (ks == 0) rows_gseq_step<NS, GK1, 0, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 1) rows_gseq_step<NS, GK1, 1 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 2) rows_gseq_step<NS, GK1, 2 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 3) rows_gseq_step<NS, GK1, 3 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 4) rows_gseq_step<NS, GK1, 4 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 5) rows_gseq_step<NS, GK1, 5 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 6) rows_gseq_step<NS, GK1, 6 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 7) rows_gseq_step<NS, GK1, 7 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 8) rows_gseq_step<NS, GK1, 8 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 9) rows_gseq_step<NS, GK1, 9 % NS, G3>(G, grh, grl, gacc, ah, al);
          else if (ks == 10) rows_gseq_step<NS, GK1, 10 % NS, G3>(G, grh, grl, gacc, ah, al);
          else rows_gseq_step<NS, GK1, 11 % NS, G3>(G, grh, grl, gacc, ah, al);
        }
        rows_gseq_finish(G, gacc, ridrow, shrow, mine, src_reg, res[0]);
      }
      gdone = true;
    }
  }

  // ---- factorised features (G runs), multiplied by the harmonics once
  if (!gdone && B.g_slot >= 0 && rmask != 0u) {
    const float* shrow = &aux->shT[(C == 1) ? 0 : 1][4 * hh];
    // (the pair's products of block B wait in the wave's private LDS area, not in 16 registers across block A's stream tiles)
    f32x4* pair = reinterpret_cast<f32x4*>(const_cast<RowsAux*>(aux) + 1) + (hh * 16 + min(max(r - B.n, 0), 15)) * 4;
    const bool in_b = r >= B.n && r < 2 * B.n;
    if (gmode == 2) {
      f32x16 tsel;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 v = pair[q4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tsel[4 * q4 + q] = in_b ? v[q] : 0.f;
      }
      rows_apply_harmonics<C>(tsel, shrow, res);
    } else if (gmode == 1) {
      const RowsGPart PA = rows_gpart_of<G3>(S, T, bi, part), PB = rows_gpart_of<G3>(S, T, bi + 1, 0);
      const f32x16 tg = rows_g_runs<NS, GR, true, G3>(S, PA, PB, ah, al, aux, rmask, src_reg, lane);
      if (in_b) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) pair[q4] = f32x4{tg[4 * q4], tg[4 * q4 + 1], tg[4 * q4 + 2], tg[4 * q4 + 3]};
      }
      f32x16 tsel;
#pragma unroll
      for (int i = 0; i < 16; ++i) tsel[i] = (r < B.n) ? tg[i] : 0.f;
      rows_apply_harmonics<C>(tsel, shrow, res);
    } else {
      const RowsGPart PA = rows_gpart_of<G3>(S, T, bi, part);
      const f32x16 tg = rows_g_runs<NS, GR, false, G3>(S, PA, PA, ah, al, aux, rmask, src_reg, lane);
      rows_apply_harmonics<C>(tg, shrow, res);
    }
  }

  RSTAMP(5 + 3 * sgi);
  // ---- the segment's stream tiles (vector-input features)
  const int cnt = (B.ntiles == 0 || B.U == 0) ? 0 : (B.nsub > 1 ? B.U : (B.U + B.ups - 1) / B.ups);
  if (cnt > 0) {
    for (int j = 0; j < cnt; ++j, ++t) {
      constexpr int KPP = NS / ROWS_NP, PIECE_Q = 2 * KPP * 64;
      f32x16 acc;
      rows_stream_step<NS, 0>(ring, wsh, t, RL.nts, wave, lane);
      acc = splat16(lbias[t * 32 + r]);
      rows_piece_lds<NS, 0>(ring, ah, al, lane, acc);                      // (piece p of every tile sits in slot p)
      rows_stream_step<NS, 1>(ring, wsh, t, RL.nts, wave, lane);
      rows_piece_lds<NS, KPP>(ring + PIECE_Q, ah, al, lane, acc);
      rows_stream_step<NS, 2>(ring, wsh, t, RL.nts, wave, lane);
      rows_piece_lds<NS, 2 * KPP>(ring + 2 * PIECE_Q, ah, al, lane, acc);
      int u = (B.nsub > 1) ? j : j * B.ups + us;
      if (!(valid && u < B.U)) u = 0;
      rows_epilogue<C>(acc, F + (u * C) * ROWS_FS + 4 * hh, ROWS_FS, res);
    }
  }

  RSTAMP(6 + 3 * sgi);
  // ---- several features per tile (n <= 16): the lane groups us = 1, 2, .. are added to group 0 in order
  if (B.nsub == 1 && B.ups > 1 && (cnt > 0 || gmode == 2)) {     // (gmode 2: the G products sit in lane group 1)
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
  RSTAMP(7 + 3 * sgi);
  return t;
}

template <int SZ, int G3>
__global__ __launch_bounds__(ROWS_NT, 2) void ddp_conv_rows_kernel(const RowsLaunch RL) {
  constexpr int NS = H2Class<SZ>::NS, RING_Q = 2 * NS * 64;     // the ring holds one tile's worth of pieces
  constexpr int NCT1 = (3 * SZ + 31) / 32, NQ = SZ / 4;     // fc.0 column tiles; 16-byte quads per edge_attr_ segment (ns floats each)
  static_assert(NS > 0 && NS % ROWS_NP == 0 && SZ % 4 == 0, "size classes with an h2 form whose k16 steps split into ROWS_NP pieces");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ConvLaunch& L = RL.L;
  const ddp_conv_shape_t& S = L.shape;
  const int tid = threadIdx.x;
  int ti, p0, nvalid;
  if (!conv_tile<ROWS_ET>(L, ti, p0, nvalid)) return;
  const ddp_conv_task_t& T = L.task[ti];
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, hh = lane >> 5;
  f32x4* ring = reinterpret_cast<f32x4*>(lds);
  float* lbias = lds + RING_Q * 4;                                        // [nts][32] bias words of the stream tiles
  char* priv = reinterpret_cast<char*>(lds) + RING_Q * 16 + RL.bias_bytes + (size_t)wave * RL.priv_bytes;
  const RowsStream wsh = rows_stream_of(T.wsh, RL.nts, 2 * NS * 1024);
  const int nvw = max(0, min(32, nvalid - 32 * wave));      // valid edges of this wave
  RSTAMP(0);

  // ---- the wave's edges (rows behind the last valid one repeat it: every load stays in bounds, nothing of theirs is stored)
  const int pr = p0 + min(32 * wave + r, nvalid - 1);
  const int src = T.src[pr], eid = T.eid[pr];
  const int pos = T.pos ? T.pos[pr] : pr;
  const f32x4 shv = reinterpret_cast<const f32x4*>(T.sh)[eid];
  const float* __restrict__ xb0 = T.seg_ptr[0] + (size_t)T.seg_idx[0][pr] * T.seg_ld[0];
  const float* __restrict__ xb1 = T.seg_ptr[1] + (size_t)T.seg_idx[1][pr] * T.seg_ld[1];
  const float* __restrict__ xb2 = T.seg_ptr[2] + (size_t)T.seg_idx[2][pr] * T.seg_ld[2];
  // the source rows' vector irreps (the features of the blocks: first read ~25 k ticks from here) are touched now, one word per
  // 128-byte line and lane half: they arrive beside the edge_attr_ gather and wait in L2
  float vtouch[2];
  {
    int lo = 1 << 30, hi = 0;
    for (int bi = 0; bi < S.nblocks; ++bi)
      if (S.blk[bi].ntiles > 0)
        for (int si = 0; si < S.blk[bi].nseg; ++si) {
          lo = min(lo, S.blk[bi].seg[si].in_off);
          hi = max(hi, S.blk[bi].seg[si].in_off + 3 * S.blk[bi].seg[si].count);
        }
    const float* __restrict__ xs = T.x_src + (size_t)src * T.ldx_src;
#pragma unroll
    for (int i = 0; i < 2; ++i) vtouch[i] = (hi > lo) ? xs[min(lo + 32 * (2 * i + hh), hi - 1)] : 0.f;
  }

  // ---- stage tile 0, request tile 1
  // ---- request tile 0; the tiles' bias words: one table in LDS for the whole kernel (no global load inside the tile loops)
  rows_request_piece<NS>(ring, wsh, 0, ROWS_NP * RL.nts, 0, wave, lane);
  rows_request_piece<NS>(ring, wsh, 1, ROWS_NP * RL.nts, 1, wave, lane);
  for (int i = tid; i < RL.nts * 32; i += ROWS_NT) lbias[i] = T.bsp[i];
  RSTAMP(1);
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
      rows_split(xv[ks][0], ROWS_SX, h0, l0, T.h2_range_flag);
      rows_split(xv[ks][1], ROWS_SX, h1, l1, T.h2_range_flag);
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

  RSTAMP(2);
  asm volatile("" ::"v"(vtouch[0]), "v"(vtouch[1]));     // (the touched words: never used)
  // ---- fc1, transposed: D[h column m][edge n] = sum_k W1[k][32 ct + m] x[n][k]; lane (edge r, hh) ends with the h columns
  // 32 ct + (j & 3) + 8 (j >> 2) + 4 hh, j < 16, of its own edge = the k-groups (2 ct, hh) and (2 ct + 1, hh) of DDP_ROWS_KPERM
  h8 ah[NS], al[NS];
  int t = 0;
#pragma unroll
  for (int ct = 0; ct < NCT1; ++ct, ++t) {
    f32x16 acc;
#pragma unroll
    for (int pc = 0; pc < ROWS_NP; ++pc) {
      constexpr int KPP = NS / ROWS_NP, PIECE_Q = 2 * KPP * 64;
      if (pc == 0) rows_stream_step<NS, 0>(ring, wsh, t, RL.nts, wave, lane);
      else if (pc == 1) rows_stream_step<NS, 1>(ring, wsh, t, RL.nts, wave, lane);
      else rows_stream_step<NS, 2>(ring, wsh, t, RL.nts, wave, lane);
      if (pc == 0) {
        const f32x4* bp = reinterpret_cast<const f32x4*>(lbias + t * 32 + 4 * hh);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 b = bp[2 * q4];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * q4 + q] = b[q];
        }
      }
      const f32x4* slot = ring + pc * PIECE_Q;
      f32x4 w0 = slot[lane], w1 = slot[64 + lane];
      f32x4 xl = xlo[(pc * KPP) * 64 + lane];
#pragma unroll
      for (int k = 0; k < KPP; ++k) {
        const int ks = pc * KPP + k;
        const h8 wh = __builtin_bit_cast(h8, w0), wl = __builtin_bit_cast(h8, w1), xlk = __builtin_bit_cast(h8, xl);
        if (k + 1 < KPP) {
          w0 = slot[(2 * k + 2) * 64 + lane];
          w1 = slot[(2 * k + 3) * 64 + lane];
          xl = xlo[(ks + 1) * 64 + lane];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xlk, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pre = acc[j] * (ROWS_SH / (ROWS_SW * ROWS_SX));     // the h plane's scale over the accumulator's
      h2_range_check(pre, T.h2_range_flag);     // (before the relu: fmaxf drops a NaN)
      const float v = fmaxf(pre, 0.f);
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      constexpr int dummy = 0;
      (void)dummy;
      if (2 * ct + (j >> 3) < NS) {
        ah[2 * ct + (j >> 3)][j & 7] = hi;
        al[2 * ct + (j >> 3)][j & 7] = lo;
      }
    }
  }

  RSTAMP(3);
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
      // the harmonics carry the inverse of the accumulators' scales: 1 / (SH SW) for the stream tiles' features, 1 / (SH SG) for G
      constexpr float fs = 1.f / (ROWS_SH * ROWS_SW), gs = 1.f / (ROWS_SH * ROWS_SG);
      aux->sh[r][0] = shv[0] * fs; aux->sh[r][1] = shv[1] * fs; aux->sh[r][2] = shv[2] * fs; aux->sh[r][3] = shv[3] * fs;
      aux->shT[0][r] = shv[0] * gs; aux->shT[1][r] = shv[1] * gs; aux->shT[2][r] = shv[2] * gs; aux->shT[3][r] = shv[3] * gs;
    }
  }

  RSTAMP(4);
  RSTAMP_VAL(30, __popc(rmask));
  RSTAMP_VAL(31, p0 / ROWS_ET);
  // ---- the segments: blocks in order, the 32-column parts of a block in order
  int sgi = 0;
  for (int bi = 0; bi < S.nblocks; ++bi) {
    const ddp_block_t& B = S.blk[bi];
    if (B.ntiles > 0 && B.U > 0) {
      bool fast = true;     // (vector-input segments of at most 16 features: the factorised shapes of nv <= 16)
      for (int si = 0; si < B.nseg; ++si)
        fast = fast && B.seg[si].count <= 16 && (B.seg[si].kind == DDP_F_DOT || B.seg[si].kind == DDP_F_VEC_S0 || B.seg[si].kind == DDP_F_CROSS);
      if (fast)
        rows_build_features<8>(B, T, aux, F, lane);
      else
        build_features<32, 2>(B, T, aux->src, aux->sh, F, lane);
    }
    const int nparts = (B.n + 31) >> 5;
    for (int part = 0; part < nparts; ++part, ++sgi) {
      // two neighbouring vector blocks of one width with a single G part each (1o, 1e): one merged G tile product per run (rows_g_runs)
      int gmode = 0;
      if (B.C == 3 && B.g_slot >= 0 && nparts == 1 && 2 * B.n <= 32) {
        const bool with_next = bi + 1 < S.nblocks && S.blk[bi + 1].C == 3 && S.blk[bi + 1].g_slot >= 0 && S.blk[bi + 1].n == B.n;
        const bool with_prev = bi > 0 && S.blk[bi - 1].C == 3 && S.blk[bi - 1].g_slot >= 0 && S.blk[bi - 1].n == B.n;
        // (pairs are (1, 2): a block that is the second of a pair is never the first of another)
        const bool prev_is_second = with_prev && bi > 1 && S.blk[bi - 2].C == 3 && S.blk[bi - 2].g_slot >= 0 && S.blk[bi - 2].n == B.n;
        if (with_prev && !prev_is_second) gmode = 2;
        else if (with_next) gmode = 1;
      }
      if (B.C == 1)
        t = rows_segment<NS, 1, G3>(RL, B, bi, part, T, ah, al, ring, lbias, t, F, aux, rmask, src, nvw, wave, lane, sgi, 0);
      else
        t = rows_segment<NS, 3, G3>(RL, B, bi, part, T, ah, al, ring, lbias, t, F, aux, rmask, src, nvw, wave, lane, sgi, gmode);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
extern "C" int ddp_conv_rows(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, void* stream) {
  if (!shape || !tasks) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: null argument");
  if (ntasks < 0 || ntasks > DDP_MAX_TASKS) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: ntasks > DDP_MAX_TASKS");
  if (shape->nblocks < 1 || shape->nblocks > DDP_MAX_BLOCKS) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: nblocks");
  const int sc = (shape->f_in != shape->hid) ? 0 : (shape->hid == 180) ? 60 : (shape->hid == 96) ? 32 : 0;
  if (sc == 0) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: shapes of the size classes ns = 60 / 32 (f_in = hid = 180 / 96) only");
  RowsLaunch RL;
  ConvLaunch& L = RL.L;
  L.shape = *shape;
  L.r1_floats = 0;
  L.tv_off = 0;
  L.ntasks = 0;
  L.dev_counts = 0;
  const int NS = (sc == 60) ? 12 : 6, nct1 = shape->nct1;
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
  int tiles = 0, gfmt = -1, form = -1;
  for (int i = 0; i < ntasks; ++i) {
    const ddp_conv_task_t& T = tasks[i];
    if (T.n_edges <= 0) continue;  // an empty conv sends no message (models/score_model.py:109-111)
    if (T.gh_fmt != 0 && T.gh_fmt != 1) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: task.gh_fmt must be 0 or 1");
    if (gfmt >= 0 && T.gh_fmt != gfmt) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: the tasks of a launch carry G in ONE plane form");
    gfmt = T.gh_fmt;
    if (T.rows_form != 0 && T.rows_form != 1) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: task.rows_form must be 0 or 1");
    if ((T.rows_bias_k != 0 || T.rows_seg0 != 0 || T.rows_seg1 != 0 || T.rows_nts != 0) && T.rows_form != 1)
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: task.rows_bias_k / rows_seg0 / rows_seg1 / rows_nts need rows_form 1");
    if (form >= 0 && T.rows_form != form) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: the tasks of a launch carry ONE form of operand images");
    form = T.rows_form;
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
  if (form == 1) return ddp_conv_rows16_launch(shape, tasks, ntasks, sc, stream);     // operand images of v_mfma_f32_16x16x32_f16
  RL.nts = nts;
  int fbytes = frows * ROWS_FS * 4;
  fbytes = (fbytes + 127) / 128 * 128;
  RL.aux_off = fbytes;
  int priv = fbytes + (int)sizeof(RowsAux) + 2048;      // (+ the parked G products of a merged pair of vector blocks)
  if (priv < NS * 1024) priv = NS * 1024;          // the lo plane of edge_attr_ during fc1
  priv = (priv + 127) / 128 * 128;
  RL.priv_bytes = priv;
  RL.bias_bytes = (nts * 128 + 127) / 128 * 128;
  size_t lds_bytes = (size_t)2 * (NS * 1024) + RL.bias_bytes + (size_t)ROWS_NW * priv;
  if (2 * lds_bytes > 160 * 1024) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: LDS budget of two workgroups per CU exceeded (too many vector features per block)");
  // occupancy shaping (ddp_set_occupancy_shaping): a launch that is to leave one 256-register wave slot per SIMD to another kernel asks
  // for more LDS than two workgroups per CU can have
  if ((size_t)ddp_shape_rows_min_lds > lds_bytes) lds_bytes = (size_t)ddp_shape_rows_min_lds;
  static int lds_have[4] = {0, 0, 0, 0};
  hipError_t err;
#define ROWS_LAUNCH(SZ_, G3_, I_)                                                                                                       \
  {                                                                                                                                     \
    err = ddp_need_lds(reinterpret_cast<const void*>(ddp_conv_rows_kernel<SZ_, G3_>), (int)lds_bytes, &lds_have[I_]);                  \
    if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(conv rows)");                                                 \
    hipLaunchKernelGGL((ddp_conv_rows_kernel<SZ_, G3_>), dim3(tiles), dim3(ROWS_NT), lds_bytes, (hipStream_t)stream, RL);               \
  }
  if (sc == 60) {
    if (gfmt == 1) ROWS_LAUNCH(60, 1, 2) else ROWS_LAUNCH(60, 0, 0)
  } else {
    if (gfmt == 1) ROWS_LAUNCH(32, 1, 3) else ROWS_LAUNCH(32, 0, 1)
  }
#undef ROWS_LAUNCH
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_conv_rows launch");
  return 0;
}
