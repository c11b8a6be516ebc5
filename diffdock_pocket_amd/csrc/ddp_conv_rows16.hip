// ddp_conv_rows16.hip - the row-stationary factorised conv of ddp_conv_rows.hip with every tile product on v_mfma_f32_16x16x32_f16
// (round 6).  Same contract (csrc/ddp_conv.hip), same organisation - 128-edge workgroups of four waves, two per CU, a wave owns 32
// source-ordered edges and keeps h = relu(fc1) of them in registers, weights once per workgroup through a three-slot LDS ring filled by
// LDS-DMA, unified fp16 hi/lo planes (three products per k-step on one accumulator), G tiles straight from memory, no message tile in
// LDS - selected per task by ddp_conv_task_t::rows_form = 1 (ddp_conv_rows.hip keeps form 0: v_mfma_f32_32x32x16_f16).
//
// Replaces, per conv (reference file:line):  edge_attr_ = cat(...)  models/all_atom_score_model.py:273-312;  w = fc(edge_attr_)
// models/score_model.py:100-105,114;  msg = FasterTensorProduct(x[src], sh, w)  models/layers.py:34-85.
//
// Why another shape: both instructions do 1024 FLOP per cycle and SIMD, but under the 16x16x32 shape the chip holds a higher clock.  The
// stripped stream-tile loop of the kernel - the same ring, barriers, operand planes and feature contraction - runs 1.17 x faster on it
// (tools/micro/stream_16.hip, profiles/r06_stream_16_micro.txt: 0.386 against 0.454 ms, 1.92 against 1.70 GHz in-kernel, 4 % fewer ticks),
// and its 16-row / 16-column tiles fit the work: a run of edges with one source node is at most 16 rows for most convs (atom<-atom:
// exactly 8), a vector block's 10 columns are one 16-column tile.
//
// Operand images (lane l: n = l % 16, g = l / 16):
//   A  16 rows x 32 k: lane holds row n, k = 8 g + i, i < 8;    B  32 k x 16 columns: lane holds column n, k = 8 g + i;
//   D  16 x 16: lane holds column n, rows 4 g + j, j < 4 (four registers).
//   The wave's 32 edges are TWO 16-row tiles (rt), a 32-column weight tile TWO 16-column tiles (ct): four accumulators of four
//   registers per 32 x 32 tile product, twelve MFMAs of 16 cycles per 32 k where the 32x32x16 form has six of 32 cycles.
//   h without a transpose: fc1 runs transposed (A = fc.0 tile, B = edge_attr_ fragments gathered by the lane), so the accumulator of
//   (16 h columns mt, 16 edges et) leaves lane (edge n, g) with h columns 4 g + j of sub-tile mt - and the two sub-tiles of stream tile t
//   ARE the A fragment of k32 step t of the later products: element i of lane (n, g) = position 16 (i / 4) + 4 g + i % 4 of the tile.
//   The host places h column 32 t + 8 g + i at that position of fc.0's stream tile (packing.rows_stream, form 1), so the k order of h,
//   of the fc.3 tiles and of G is the NATURAL one: G keeps the byte layout ddp_stage_a_gh writes (k8 group = k / 8), only its k's are not
//   permuted.
// Summation order of a message element: G runs in edge order, then the stream tiles in feature order, then (blocks with several
// features per tile) the lane groups in order: fixed, bitwise reproducible; within fp32 rounding of the other conv kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#undef DDP_STAMPS
#include "ddp_conv_common.h"

#define R16_NW 4
#define R16_NT 256
#define R16_ET 128
#define R16_FS 36     // floats per feature row F[u * C + c][edge]
#define R16_NP 3      // pieces per stream tile (one ring slot each)
#define R16_FROWS 72  // feature rows a wave holds at a time (a block with more - the direct convs: 80 features x 3 components - builds them in chunks)
#define R16_SX ((float)DDP_ROWS_SX)
#define R16_SW ((float)DDP_ROWS_SW)
#define R16_SH ((float)DDP_ROWS_SH)
#define R16_SG ((float)DDP_ROWS_SG)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* r16_lds_ptr_t;
typedef __amdgpu_buffer_rsrc_t R16Stream;

struct R16Launch {
  ConvLaunch L;
  int nts;         // stream tiles per conv (fc.0 tiles + fc.3 tiles of all segments)
  int bias_tiles;  // stream tiles whose bias words sit in the LDS table (all, or fc.0's: ddp_conv_task_t::rows_bias_k)
  int bias_bytes;  // LDS bytes of the bias table behind the ring
  int priv_bytes;  // LDS bytes of a wave's private area
  int aux_off;     // byte offset of the per-edge tables inside it
  int frows;       // feature rows of the private area (<= R16_FROWS)
};
static_assert(sizeof(ConvLaunch) + 16 <= 4096, "the launch descriptor travels as a kernel argument");
struct R16Aux {
  float shT[4][32];   // harmonics, component-major (the "feature rows" of the factorised features), x 1 / (SH SG)
  float sh[32][4];    // ... edge-major, x 1 / (SH SW) (the stream tiles' features)
  int src[32], pos[32], rid[32];
};

__device__ __forceinline__ f32x4 r16_splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ void r16_split(const f32x4 v, float scale, h4& hi, h4& lo, int32_t* flag) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float V = v[i] * scale;
    h2_range_check(V, flag);
    hi[i] = (_Float16)V;
    lo[i] = (_Float16)(V - (float)hi[i]);
  }
}

// ---- the weight stream (ddp_conv_rows.hip's: a tile travels as R16_NP pieces through a three-slot LDS ring; one BARE barrier per piece)
__device__ __forceinline__ R16Stream r16_stream_of(const void* wsh, int nts, int tile_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wsh), 0, nts * tile_bytes, 0x00020000);
}
template <int NS>
__device__ __forceinline__ void r16_request_piece(f32x4* ring, R16Stream wsh, int jn, int npieces, int slot, int wave, int lane) {
  constexpr int FPP = 2 * NS / R16_NP, FPW = FPP / R16_NW, PIECE_Q = FPP * 64;
  static_assert(NS % R16_NP == 0 && FPP % R16_NW == 0, "every wave moves the same number of fragments per piece");
  f32x4* nslot = ring + slot * PIECE_Q;
  const int piece_off = min(jn, npieces - 1) * (PIECE_Q * 16);
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsh, (r16_lds_ptr_t)(nslot + (wave + R16_NW * f) * 64), 16, ((wave + R16_NW * f) * 64 + lane) * 16, piece_off, 0, 0);
}
template <int NS, int P>
__device__ __forceinline__ void r16_stream_step(f32x4* ring, R16Stream wsh, int t, int nts, int wave, int lane) {
  constexpr int FPW = 2 * NS / R16_NP / R16_NW;
  static_assert(FPW == 2 || FPW == 1, "the literals below");
  // (hipcc does not wait for an LDS-DMA in front of a barrier: "at most FPW outstanding" = everything older than the copies of piece j + 1 has landed)
  if constexpr (FPW == 2)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  r16_request_piece<NS>(ring, wsh, R16_NP * t + P + 2, R16_NP * nts, (P + 2) % R16_NP, wave, lane);
}

// acc[2 rt + ct] += A(h of row tile rt, k32 steps of piece P) x B(piece in LDS: fragments [k32 step][ct][plane]); three split products
// per k-step on ONE accumulator (unified planes)
template <int NS, int P>
__device__ __forceinline__ void r16_piece(const f32x4* slot, const h8 (&ah)[NS], const h8 (&al)[NS], int lane, f32x4 (&acc)[4]) {
  constexpr int KP2 = NS / R16_NP / 2;     // k32 steps per piece
  static_assert((NS / R16_NP) % 2 == 0, "a piece holds whole k32 steps");
#pragma unroll
  for (int k = 0; k < KP2; ++k) {
    constexpr int dummy = 0;
    (void)dummy;
    const int s = P * KP2 + k;
    h8 b[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) b[ct][pl] = __builtin_bit_cast(h8, slot[((2 * k + ct) * 2 + pl) * 64 + lane]);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 d = acc[2 * rt + ct];
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[2 * s + rt], b[ct][0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[2 * s + rt], b[ct][1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[2 * s + rt], b[ct][0], d, 0, 0, 0);
        acc[2 * rt + ct] = d;
      }
  }
}

// The basis features of a block's vector-input segments (DOT, VEC_S0, CROSS; ddp_conv_rows.hip's rows_build_features: independent of the
// MFMA shape - feature rows are indexed by edge)
template <int MAXI>
__device__ __forceinline__ void r16_build_features(const ddp_block_t& B, const ddp_conv_task_t& T, const R16Aux* aux, float* F, int lane) {
  constexpr int FS = R16_FS;
  const int e = lane & 31, half = lane >> 5;
  const float* __restrict__ xrow = T.x_src + (size_t)aux->src[e] * T.ldx_src;
  const float s0 = aux->sh[e][0], sx = aux->sh[e][1], sy = aux->sh[e][2], sz = aux->sh[e][3];
  const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt2 = 0.70710678118654752f;
  int ubase = 0;
  for (int si = 0; si < B.nseg; ++si) {
    const int kind = B.seg[si].kind, off = B.seg[si].in_off, cnt = B.seg[si].count;
    float ax[MAXI], ay[MAXI], az[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int ul = max(min(half + 2 * i, cnt - 1), 0);
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

// ... the features [u0, u1) of a block of ANY kinds, rows (u - u0) C + c: the chunks of a block whose features do not fit the private area at
// once (the direct convs; build_features of ddp_conv_common.h over a range, for one wave: lane = (edge, half))
__device__ __forceinline__ void r16_build_features_range(const ddp_block_t& B, const ddp_conv_task_t& T, const R16Aux* aux, float* F, int lane,
                                                         int u0, int u1) {
  constexpr int FS = R16_FS;
  const int e = lane & 31, half = lane >> 5;
  const float* __restrict__ xrow = T.x_src + (size_t)aux->src[e] * T.ldx_src;
  const float s0 = aux->sh[e][0], sx = aux->sh[e][1], sy = aux->sh[e][2], sz = aux->sh[e][3];
  const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt2 = 0.70710678118654752f;
  int ubase = 0;
  for (int si = 0; si < B.nseg; ++si) {
    const int kind = B.seg[si].kind, off = B.seg[si].in_off, cnt = B.seg[si].count;
    const int lo = max(0, u0 - ubase), hi = min(cnt, u1 - ubase);
    for (int ul = lo + half; ul < hi; ul += 2) {
      const int u = ubase + ul - u0;
      if (kind == DDP_F_SCALAR_S0) {
        F[u * FS + e] = xrow[off + ul] * s0;
      } else if (kind == DDP_F_DOT) {
        F[u * FS + e] = (xrow[off + 3 * ul] * sx + xrow[off + 3 * ul + 1] * sy + xrow[off + 3 * ul + 2] * sz) * inv_sqrt3;
      } else if (kind == DDP_F_SCALAR_S1) {
        const float a = xrow[off + ul];
        F[(u * 3 + 0) * FS + e] = a * sx;
        F[(u * 3 + 1) * FS + e] = a * sy;
        F[(u * 3 + 2) * FS + e] = a * sz;
      } else if (kind == DDP_F_VEC_S0) {
        F[(u * 3 + 0) * FS + e] = xrow[off + 3 * ul] * s0;
        F[(u * 3 + 1) * FS + e] = xrow[off + 3 * ul + 1] * s0;
        F[(u * 3 + 2) * FS + e] = xrow[off + 3 * ul + 2] * s0;
      } else {  // DDP_F_CROSS: a x s1 / sqrt(2)
        const float ax = xrow[off + 3 * ul], ay = xrow[off + 3 * ul + 1], az = xrow[off + 3 * ul + 2];
        F[(u * 3 + 0) * FS + e] = (ay * sz - az * sy) * inv_sqrt2;
        F[(u * 3 + 1) * FS + e] = (az * sx - ax * sz) * inv_sqrt2;
        F[(u * 3 + 2) * FS + e] = (ax * sy - ay * sx) * inv_sqrt2;
      }
    }
    ubase += cnt;
  }
}

// Where the G tile of segment (block bi, part) sits inside a node's row of task.gh[slot] (plane form 0: [k8][wp columns][plane][8 halves] per
// part, then Gb per padded column; include/ddp_hip.h)
struct R16GPart {
  const char* base;      // the part's tile inside node 0's row of its G array
  size_t gldb;           // node stride in bytes
  int wp, nmine, bias_off;   // padded width, columns, byte offset of Gb[column 0] from `base` (plane form 1: of Gb's first group)
  int cumw;                  // padded columns of the slot in front of the part
};
// Plane form GF (ddp_conv_task_t::gh_fmt): 0 = [k8][c][plane][8 halves], 32 bytes per unit (k8, c); 1 = 24-byte units [8 hi halves | 8
// continuation bytes], Gb in 24-byte groups of six fp32 behind the units of all parts
template <int GF>
__device__ __forceinline__ R16GPart r16_gpart_of(const ddp_conv_shape_t& S, const ddp_conv_task_t& T, int bi, int part) {
  const ddp_block_t& B = S.blk[bi];
  int wp = 0, cumw = 0, gcp = 0;
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
  const int n8 = (S.hid + 7) >> 3;
  R16GPart P;
  if constexpr (GF == 1) {
    P.base = reinterpret_cast<const char*>(T.gh[B.g_slot]) + (size_t)(n8 * cumw) * 24;
    P.gldb = (size_t)DDP_GH3_LD(S.hid, gcp) * 4;
    P.bias_off = 24 * n8 * gcp - 24 * n8 * cumw;
  } else {
    P.base = reinterpret_cast<const char*>(T.gh[B.g_slot]) + (size_t)(2 * n8 * cumw) * 16;
    P.gldb = (size_t)DDP_GH_LD(S.hid, gcp) * 4;
    P.bias_off = (8 * n8 * gcp + cumw) * 4 - (2 * n8 * cumw) * 16;
  }
  P.cumw = cumw;
  P.wp = wp;
  P.nmine = min(32, B.n - 32 * part);
  return P;
}

// The G runs of one segment, NCT 16-column tiles wide (1: the part has at most 16 columns): per run of edges with one source node the tile
// product h[32 x 16 NS] @ G[node][16 NS x 16 NCT], B = the node's G tile straight from memory through a register ring of GK k32 steps
// (buffer loads: descriptor of the node's row + a uniform fragment offset + one lane offset per column tile); while run i is multiplied
// run i + 1 arrives in the slots its steps free; behind the last run the loads hit an EMPTY buffer.  Every row belongs to exactly one
// run: the run's rows are selected from its product (rid), times the harmonic(s), into the segment's accumulators.
template <int NCT>
struct R16GSeq {
  R16Stream rs, rsn;     // the current run's node, the next run's (behind the last run: an empty buffer)
  const char* base;
  size_t gldb;
  float bias[NCT];       // Gb of the current run's columns
  unsigned m;            // runs not yet started (bit = first row)
  int run, nruns;
  unsigned l_main[NCT], l_last[NCT], l_bias[NCT];
  int gc;
};
template <int NCT>
__device__ __forceinline__ R16Stream r16_gseq_node(const R16GSeq<NCT>& G, int src_reg, int row) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(G.base + (size_t)__builtin_amdgcn_readlane(src_reg, row) * G.gldb), 0, (int)G.gldb, 0x00020000);
}
template <int NCT>
__device__ __forceinline__ void r16_gseq_next(R16GSeq<NCT>& G, int src_reg, f32x4 (&gacc)[4]) {
  G.m &= G.m - 1u;
  G.rsn = (G.m != 0u) ? r16_gseq_node(G, src_reg, __builtin_ctz(G.m)) : __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(G.base), 0, 0, 0x00020000);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) G.bias[ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(G.rs, G.l_bias[ct], 0, 0));
#pragma unroll
  for (int i = 0; i < 4; ++i) gacc[i] = r16_splat4(0.f);
}
// fragment (k32 step kq, column tile ct, plane) of the tile behind descriptor R: k8 group 4 kq + g (the last step: clamped to the row's last group)
template <int GF>
struct R16Lo {      // what a lane keeps of a fragment's lo plane in the ring: 8 fp16 words, or 8 continuation bytes (decoded when the step multiplies)
  typedef f32x4 T;
};
typedef uint32_t r16_u32x2 __attribute__((ext_vector_type(2)));
template <>
struct R16Lo<1> {
  typedef r16_u32x2 T;
};
template <int NCT, int NS2, int GF>
__device__ __forceinline__ f32x4 r16_gfrag_hi(const R16GSeq<NCT>& G, R16Stream R, int kq, int ct) {
  constexpr int UB = GF == 1 ? 24 : 32;
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R, ((kq == NS2 - 1) ? G.l_last[ct] : G.l_main[ct]),
                                                                          (kq == NS2 - 1) ? 0 : 4 * kq * G.gc * UB, 0));
}
template <int NCT, int NS2, int GF>
__device__ __forceinline__ typename R16Lo<GF>::T r16_gfrag_lo(const R16GSeq<NCT>& G, R16Stream R, int kq, int ct) {
  if constexpr (GF == 1)
    return __builtin_bit_cast(r16_u32x2, __builtin_amdgcn_raw_buffer_load_b64(R, ((kq == NS2 - 1) ? G.l_last[ct] : G.l_main[ct]) + 16,
                                                                               (kq == NS2 - 1) ? 0 : 4 * kq * G.gc * 24, 0));
  else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R, ((kq == NS2 - 1) ? G.l_last[ct] : G.l_main[ct]) + 16,
                                                                            (kq == NS2 - 1) ? 0 : 4 * kq * G.gc * 32, 0));
}
// plane form 1: the lo words of a fragment from its hi words and continuation bytes.  V = hi + sign(hi) 2^E u8 / 2^18 (E = hi's exponent;
// hi = V truncated): the byte lands in the mantissa of 2^-8 (0x1C00 | u8 = 2^-8 + u8 2^-18), minus 2^-8, times hi's sign-and-exponent
// word - four packed instructions per pair of values; a zero exponent (|V| < 2^-14) gives lo = 0
__device__ __forceinline__ h8 r16_lo_of(const f32x4 hi, const r16_u32x2 by) {
  typedef _Float16 r16_h2 __attribute__((ext_vector_type(2)));
  typedef uint32_t r16_u32x4 __attribute__((ext_vector_type(4)));
  const r16_u32x4 hw = __builtin_bit_cast(r16_u32x4, hi);
  const r16_h2 c = {(_Float16)0.00390625f, (_Float16)0.00390625f};
  r16_u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x = __builtin_amdgcn_perm(0x1c1c1c1cu, by[i >> 1], (i & 1) ? 0x07030602u : 0x05010400u);
    const r16_h2 y = __builtin_bit_cast(r16_h2, x) - c;
    const r16_h2 pw = __builtin_bit_cast(r16_h2, hw[i] & 0xfc00fc00u);
    out[i] = __builtin_bit_cast(uint32_t, y * pw);
  }
  return __builtin_bit_cast(h8, out);
}
template <int NS, int NCT, int GK, int GF>
__device__ __forceinline__ void r16_gseq_init(R16GSeq<NCT>& G, f32x4 (&gh)[GK][NCT], typename R16Lo<GF>::T (&gl)[GK][NCT], f32x4 (&gacc)[4],
                                              const ddp_conv_shape_t& S, const R16GPart& PA, unsigned rmask, int src_reg, int lane) {
  constexpr int NS2 = NS / 2;
  const int n = lane & 15, g = lane >> 4;
  const int n8 = (S.hid + 7) >> 3;
  G.gc = PA.wp;
  const int k8l = min(4 * (NS2 - 1) + g, n8 - 1);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int cl = min(16 * ct + n, PA.nmine - 1);       // (lanes behind the part's last column read a valid one: they select nothing)
    constexpr unsigned UB = GF == 1 ? 24u : 32u;
    G.l_main[ct] = (unsigned)(g * G.gc + cl) * UB;
    G.l_last[ct] = (unsigned)(k8l * G.gc + cl) * UB;
    if constexpr (GF == 1)
      G.l_bias[ct] = (unsigned)(PA.bias_off + 24 * ((PA.cumw + cl) / 6) + 4 * ((PA.cumw + cl) % 6));
    else
      G.l_bias[ct] = (unsigned)(PA.bias_off + 4 * cl);
  }
  G.base = PA.base;
  G.gldb = PA.gldb;
  G.m = rmask;
  G.nruns = __builtin_amdgcn_readfirstlane(__popc(rmask));
  G.run = 0;
  G.rs = r16_gseq_node(G, src_reg, __builtin_ctz(rmask));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < GK; ++k)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      gh[k][ct] = r16_gfrag_hi<NCT, NS2, GF>(G, G.rs, k, ct);
      gl[k][ct] = r16_gfrag_lo<NCT, NS2, GF>(G, G.rs, k, ct);
    }
  __builtin_amdgcn_sched_barrier(0);
  r16_gseq_next(G, src_reg, gacc);
}
// k32 step KS of the current run: the split products of both row tiles against ring slot KS % GK, which then takes the fragments GK steps on.
// (Multiplying only the row tile a run lies in - a run of at most 16 edges that does not straddle: half the matrix work of its product - was
// built and measured: three straight-line variants of the chain cost the 12-fragment ring its registers, and the step got 0.3 ms LONGER,
// 15.75 - 15.88 against 15.47 - 15.50 ms; the runs are bound by their loads, not by their MFMAs.  profiles/r06_rows16_ab.txt)
template <int NS, int NCT, int GK, int KS, int GF>
__device__ __forceinline__ void r16_gseq_step(R16GSeq<NCT>& G, f32x4 (&gh)[GK][NCT], typename R16Lo<GF>::T (&gl)[GK][NCT], f32x4 (&gacc)[4],
                                              const h8 (&ah)[NS], const h8 (&al)[NS]) {
  constexpr int NS2 = NS / 2, q0 = KS + GK, kq = (q0 < NS2) ? q0 : q0 - NS2;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const h8 bh = __builtin_bit_cast(h8, gh[KS % GK][ct]);
    h8 bl;
    if constexpr (GF == 1)
      bl = r16_lo_of(gh[KS % GK][ct], gl[KS % GK][ct]);
    else
      bl = __builtin_bit_cast(h8, gl[KS % GK][ct]);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x4 d = gacc[2 * rt + ct];
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[2 * KS + rt], bh, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[2 * KS + rt], bl, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[2 * KS + rt], bh, d, 0, 0, 0);
      gacc[2 * rt + ct] = d;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const R16Stream srcb = (q0 < NS2) ? G.rs : G.rsn;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    gh[KS % GK][ct] = r16_gfrag_hi<NCT, NS2, GF>(G, srcb, kq, ct);
    gl[KS % GK][ct] = r16_gfrag_lo<NCT, NS2, GF>(G, srcb, kq, ct);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// the run's last k-step is done: res[c][row, column] += harmonic_c[row] * (product + Gb) for the rows of THIS run
template <int C, int NCT>
__device__ __forceinline__ void r16_gseq_finish(R16GSeq<NCT>& G, f32x4 (&gacc)[4], const R16Aux* aux, int g, const bool (&mine)[NCT], int src_reg,
                                                f32x16 (&res)[C]) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const i32x4 id = *reinterpret_cast<const i32x4*>(&aux->rid[16 * rt + 4 * g]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(&aux->shT[(C == 1) ? 0 : 1 + c][16 * rt + 4 * g]);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int sel = mine[ct] ? G.run : -2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * (2 * rt + ct) + j;
          res[c][i] = (id[j] == sel) ? res[c][i] + f[j] * (gacc[2 * rt + ct][j] + G.bias[ct]) : res[c][i];
        }
      }
    }
  }
  G.rs = G.rsn;
  ++G.run;
  r16_gseq_next(G, src_reg, gacc);
}
template <int NS, int C, int NCT, int GF>
__device__ __forceinline__ void r16_g_runs(const ddp_conv_shape_t& S, const R16GPart& PA, const h8 (&ah)[NS], const h8 (&al)[NS], const R16Aux* aux,
                                           unsigned rmask, int src_reg, int lane, f32x16 (&res)[C]) {
  // k32 steps in the ring: half a tile at NS = 12 (the whole tile at NS = 6) for the scalar segments - 12 fragments at two column tiles, like
  // ddp_conv_rows.hip -; a vector segment holds three accumulator sets, so its ring is 6 fragments at one column tile (n <= 16: the shapes
  // of nv <= 16) and one k32 step at two
  constexpr int NS2 = NS / 2, GK = (C == 3 && NCT == 2) ? 1 : ((NS2 % 3 == 0) ? 3 : 1);
  const int n = lane & 15, g = lane >> 4;
  R16GSeq<NCT> G;
  f32x4 gh[GK][NCT], gacc[4];
  typename R16Lo<GF>::T gl[GK][NCT];
  r16_gseq_init<NS, NCT, GK, GF>(G, gh, gl, gacc, S, PA, rmask, src_reg, lane);
  bool mine[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) mine[ct] = 16 * ct + n < PA.nmine;
  while (G.run < G.nruns) {
#pragma unroll
    for (int ks = 0; ks < NS2; ++ks) {
      // (static k-steps: the chain is resolved at compile time)
      if (ks == 0) r16_gseq_step<NS, NCT, GK, 0, GF>(G, gh, gl, gacc, ah, al);
      else if (ks == 1) r16_gseq_step<NS, NCT, GK, 1 % NS2, GF>(G, gh, gl, gacc, ah, al);
      else if (ks == 2) r16_gseq_step<NS, NCT, GK, 2 % NS2, GF>(G, gh, gl, gacc, ah, al);
      else if (ks == 3) r16_gseq_step<NS, NCT, GK, 3 % NS2, GF>(G, gh, gl, gacc, ah, al);
      else if (ks == 4) r16_gseq_step<NS, NCT, GK, 4 % NS2, GF>(G, gh, gl, gacc, ah, al);
      else r16_gseq_step<NS, NCT, GK, 5 % NS2, GF>(G, gh, gl, gacc, ah, al);
    }
    r16_gseq_finish<C, NCT>(G, gacc, aux, g, mine, src_reg, res);
  }
}

// One segment = one 32-column part of one weight block's output columns: the factorised features (G runs), then the segment's stream
// tiles (vector-input features), then the message columns
template <int NS, int C, int GF, bool DIRECT>
__device__ __forceinline__ int r16_segment(const R16Launch& RL, const ddp_block_t& B, int bi, int part, const ddp_conv_task_t& T, const h8 (&ah)[NS],
                                           const h8 (&al)[NS], f32x4* ring, const float* lbias, int t, float* F, const R16Aux* aux, unsigned rmask,
                                           int src_reg, int nvw, int wave, int lane, int nts) {
  const ddp_conv_shape_t& S = RL.L.shape;
  const int n = lane & 15, g = lane >> 4;
  const R16Stream wsh = r16_stream_of(T.wsh, nts, 2 * NS * 1024);
  // lane -> (output channel, feature slot) of the segment's tiles, per 16-column tile ct: column 16 ct + n of the 32-column tile
  int ncol[2], us[2];
  bool valid[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = 16 * ct + n;
    if (B.nsub > 1) {
      ncol[ct] = 32 * part + c;
      us[ct] = 0;
      valid[ct] = ncol[ct] < B.n;
    } else {
      us[ct] = c / B.n;
      ncol[ct] = c - us[ct] * B.n;
      valid[ct] = us[ct] < B.ups;
    }
  }
  f32x16 res[C];      // register 4 (2 rt + ct) + j <-> edge row 16 rt + 4 g + j, column 16 ct + n
#pragma unroll
  for (int c = 0; c < C; ++c) res[c] = splat16(0.f);

  // ---- factorised features
  if constexpr (!DIRECT) {
    if (B.g_slot >= 0 && rmask != 0u) {
      const R16GPart PA = r16_gpart_of<GF>(S, T, bi, part);
      if (PA.nmine > 16)
        r16_g_runs<NS, C, 2, GF>(S, PA, ah, al, aux, rmask, src_reg, lane, res);
      else
        r16_g_runs<NS, C, 1, GF>(S, PA, ah, al, aux, rmask, src_reg, lane, res);
    }
  }

  // ---- the segment's stream tiles (vector-input features)
  const int cnt = (B.ntiles == 0 || B.U == 0) ? 0 : (B.nsub > 1 ? B.U : (B.U + B.ups - 1) / B.ups);
  // a block whose U C feature rows do not fit the private area builds them chunk by chunk: tpc tiles (tpc upt features) per chunk
  const int upt = (B.nsub > 1) ? 1 : B.ups;                      // features per tile
  const bool chunked = B.U * C > RL.frows;
  const int tpc = chunked ? max(1, RL.frows / (C * upt)) : cnt;
  int u0 = 0;
  for (int j = 0; j < cnt; ++j, ++t) {
    constexpr int KPP = NS / R16_NP, PIECE_Q = 2 * KPP * 64;
    if (chunked && j % tpc == 0) {
      u0 = j * upt;
      r16_build_features_range(B, T, aux, F, lane, u0, min(B.U, u0 + tpc * upt));
    }
    f32x4 acc[4];
    r16_stream_step<NS, 0>(ring, wsh, t, nts, wave, lane);
    {
      // (rows_bias_k: the tile's bias is its k row `hid`, times h[hid] = 1)
      const bool bt = t < RL.bias_tiles;
      const float b0 = bt ? lbias[t * 32 + n] : 0.f, b1 = bt ? lbias[t * 32 + 16 + n] : 0.f;
      acc[0] = r16_splat4(b0);
      acc[1] = r16_splat4(b1);
      acc[2] = r16_splat4(b0);
      acc[3] = r16_splat4(b1);
    }
    r16_piece<NS, 0>(ring, ah, al, lane, acc);                      // (piece p of every tile sits in slot p)
    r16_stream_step<NS, 1>(ring, wsh, t, nts, wave, lane);
    r16_piece<NS, 1>(ring + PIECE_Q, ah, al, lane, acc);
    r16_stream_step<NS, 2>(ring, wsh, t, nts, wave, lane);
    r16_piece<NS, 2>(ring + 2 * PIECE_Q, ah, al, lane, acc);
    // the feature contraction in the D layout: out[c][row, column] += F[(u c)][row] * acc[row, column]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      int u = (B.nsub > 1) ? j : j * B.ups + us[ct];
      if (!(valid[ct] && u < B.U)) u = 0;
      const float* frow = F + ((u - u0) * C) * R16_FS + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const f32x4 f = *reinterpret_cast<const f32x4*>(frow + c * R16_FS + 16 * rt);
#pragma unroll
          for (int q = 0; q < 4; ++q) res[c][4 * (2 * rt + ct) + q] += f[q] * acc[2 * rt + ct][q];
        }
    }
  }

  // ---- several features per tile (n <= 16: the output columns all sit in column tile 0): lane groups us = 1, 2, .. are added to group 0 in order
  if (B.nsub == 1 && B.ups > 1 && cnt > 0) {
    for (int s = 1; s < B.ups; ++s) {
      const int csrc = min(n + s * B.n, 31);             // column of the 32-column tile that holds feature group s of this lane's channel
      const int from = (csrc & 15) + 16 * g;
      const bool hi_tile = csrc >= 16;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v0 = __shfl(res[c][4 * (2 * rt) + q], from), v1 = __shfl(res[c][4 * (2 * rt + 1) + q], from);
            if (us[0] == 0) res[c][4 * (2 * rt) + q] += hi_tile ? v1 : v0;
          }
    }
  }

  // ---- the message columns of the segment
  float* __restrict__ mo = T.msg + B.out_off;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    if (valid[ct] && us[ct] == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const i32x4 pq = *reinterpret_cast<const i32x4*>(&aux->pos[16 * rt + 4 * g]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (16 * rt + 4 * g + q < nvw) {
#pragma unroll
            for (int c = 0; c < C; ++c) mo[(size_t)pq[q] * S.d_out + ncol[ct] * C + c] = res[c][4 * (2 * rt + ct) + q];
          }
      }
    }
  }
  return t;
}

// (DIRECT: a shape without factorised blocks - the layers' direct convs, every feature a stream tile: the G runs are compiled out and the
// kernel carries a name of its own, ddp_conv_rows16_direct_kernel, so that profiles keep the two kinds of launch apart)
template <int SZ, int GF, bool DIRECT>
__device__ __forceinline__ void r16_body(const R16Launch& RL) {
  constexpr int NS = H2Class<SZ>::NS, NS2 = NS / 2, RING_Q = 2 * NS * 64;     // the ring holds one tile's worth of pieces
  constexpr int NQ = SZ / 4;     // 16-byte quads per edge_attr_ segment (ns floats each)
  static_assert(NS > 0 && NS % (2 * R16_NP) == 0 && SZ % 4 == 0, "size classes whose k16 steps split into R16_NP pieces of whole k32 steps");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ConvLaunch& L = RL.L;
  const ddp_conv_shape_t& S = L.shape;
  const int tid = threadIdx.x;
  int ti, p0, nvalid;
  if (!conv_tile<R16_ET>(L, ti, p0, nvalid)) return;
  const ddp_conv_task_t& T = L.task[ti];
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, n = lane & 15, g = lane >> 4;
  f32x4* ring = reinterpret_cast<f32x4*>(lds);
  float* lbias = lds + RING_Q * 4;                                        // [nts][32] bias words of the stream tiles
  char* priv = reinterpret_cast<char*>(lds) + RING_Q * 16 + RL.bias_bytes + (size_t)wave * RL.priv_bytes;
  const int nts = (T.rows_nts > 0) ? T.rows_nts : RL.nts;      // stream tiles of THIS task (a task of a segment range has a stream of its own)
  const R16Stream wsh = r16_stream_of(T.wsh, nts, 2 * NS * 1024);
  const int nvw = max(0, min(32, nvalid - 32 * wave));      // valid edges of this wave

  // ---- the wave's edges (rows behind the last valid one repeat it: every load stays in bounds, nothing of theirs is stored).  Per-edge
  // scalars by lane r = l % 32 (both halves), the edge_attr_ gather by lane (n, g) for the edges 16 et + n of the two edge tiles
  const int pr = p0 + min(32 * wave + r, nvalid - 1);
  const int src = T.src[pr], eid = T.eid[pr];
  const int pos = T.pos ? T.pos[pr] : pr;
  const f32x4 shv = reinterpret_cast<const f32x4*>(T.sh)[eid];
  const float* __restrict__ xb[2][3];
#pragma unroll
  for (int et = 0; et < 2; ++et) {
    const int pe = p0 + min(32 * wave + 16 * et + n, nvalid - 1);
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) xb[et][sg] = T.seg_ptr[sg] + (size_t)T.seg_idx[sg][pe] * T.seg_ld[sg];
  }
  // the source rows' vector irreps (the features of the blocks) are touched now, one word per 128-byte line: they arrive beside the
  // edge_attr_ gather and wait in L2
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
    for (int i = 0; i < 2; ++i) vtouch[i] = (hi > lo) ? xs[min(lo + 32 * (2 * i + (lane >> 5)), hi - 1)] : 0.f;
  }

  // ---- request tiles 0 / 1 of the stream; the tiles' bias words: one table in LDS for the whole kernel
  r16_request_piece<NS>(ring, wsh, 0, R16_NP * nts, 0, wave, lane);
  r16_request_piece<NS>(ring, wsh, 1, R16_NP * nts, 1, wave, lane);
  for (int i = tid; i < RL.bias_tiles * 32; i += R16_NT) lbias[i] = T.bsp[i];
  // ---- edge_attr_ of the wave's edges as B-operand fragments: lane (edge n of tile et, g) holds k = 32 s + 8 g + i.  hi plane in registers,
  // lo plane in the wave's private LDS area (each lane reads back what it wrote); image index 2 s + et
  h8 xh[NS];
  f32x4* xlo = reinterpret_cast<f32x4*>(priv);
  {
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
      f32x4 xv[2][2];
#pragma unroll
      for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int kq = 8 * s + 2 * g + q;                   // quad index inside edge_attr_ = cat(seg0, seg1, seg2), NQ quads each
          const int sg = kq / NQ, off = kq - sg * NQ;
          const float* __restrict__ b = (sg == 0) ? xb[et][0] : (sg == 1) ? xb[et][1] : xb[et][2];
          xv[et][q] = (sg < 3) ? reinterpret_cast<const f32x4*>(b)[off] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int et = 0; et < 2; ++et) {
        h4 h0, l0, h1, l1;
        r16_split(xv[et][0], R16_SX, h0, l0, T.h2_range_flag);
        r16_split(xv[et][1], R16_SX, h1, l1, T.h2_range_flag);
        h8 lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[2 * s + et][i] = h0[i];
          xh[2 * s + et][4 + i] = h1[i];
          lo[i] = l0[i];
          lo[4 + i] = l1[i];
        }
        xlo[(2 * s + et) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
      }
    }
  }
  asm volatile("" ::"v"(vtouch[0]), "v"(vtouch[1]));     // (the touched words: never used)

  // ---- fc1, transposed: D[h position m of sub-tile mt][edge n of tile et] = sum_k W1[k][.] x[edge][k]; lane (edge n, g) ends with the h values at
  // positions 16 mt + 4 g + j of stream tile t = the elements i = 4 mt + j of the A fragment (k32 step t, row tile et) of the later products
  h8 ah[NS], al[NS];
  int t = 0;
#pragma unroll
  for (int ct = 0; ct < NS2; ++ct, ++t) {
    f32x4 acc[4];      // [2 mt + et]
#pragma unroll
    for (int pc = 0; pc < R16_NP; ++pc) {
      constexpr int KPP = NS / R16_NP, KP2 = KPP / 2, PIECE_Q = 2 * KPP * 64;
      if (pc == 0) r16_stream_step<NS, 0>(ring, wsh, t, nts, wave, lane);
      else if (pc == 1) r16_stream_step<NS, 1>(ring, wsh, t, nts, wave, lane);
      else r16_stream_step<NS, 2>(ring, wsh, t, nts, wave, lane);
      if (pc == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + t * 32 + 16 * mt + 4 * g);
          acc[2 * mt] = b;
          acc[2 * mt + 1] = b;
        }
      }
      const f32x4* slot = ring + pc * PIECE_Q;
#pragma unroll
      for (int k = 0; k < KP2; ++k) {
        const int s = pc * KP2 + k;
        h8 w[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) w[mt][pl] = __builtin_bit_cast(h8, slot[((2 * k + mt) * 2 + pl) * 64 + lane]);
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const h8 xl = __builtin_bit_cast(h8, xlo[(2 * s + et) * 64 + lane]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x4 d = acc[2 * mt + et];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[mt][0], xh[2 * s + et], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[mt][0], xl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[mt][1], xh[2 * s + et], d, 0, 0, 0);
            acc[2 * mt + et] = d;
          }
        }
      }
    }
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pre = acc[2 * (i >> 2) + et][i & 3] * (R16_SH / (R16_SW * R16_SX));     // the h plane's scale over the accumulator's
        h2_range_check(pre, T.h2_range_flag);     // (before the relu: fmaxf drops a NaN)
        const float v = fmaxf(pre, 0.f);
        const _Float16 hi = (_Float16)v;
        ah[2 * ct + et][i] = hi;
        al[2 * ct + et][i] = (_Float16)(v - (float)hi);
      }
  }

  // ---- per-edge tables of the wave (the private area is free: the lo plane of edge_attr_ is dead)
  float* F = reinterpret_cast<float*>(priv);
  R16Aux* aux = reinterpret_cast<R16Aux*>(priv + RL.aux_off);
  unsigned rmask;
  {
    const int prev = __shfl_up(src, 1);
    const bool rowv = (lane < 32) && (r < nvw);
    const bool runstart = rowv && (r == 0 || src != prev);
    rmask = (unsigned)(__ballot(runstart) & 0xffffffffull);
    const unsigned upto = (r == 31) ? ~0u : ((2u << r) - 1u);
    if (lane < 32) {
      aux->src[r] = src;
      aux->pos[r] = pos;
      aux->rid[r] = rowv ? (int)__popc(rmask & upto) - 1 : -1;
      // the harmonics carry the inverse of the accumulators' scales: 1 / (SH SW) for the stream tiles' features, 1 / (SH SG) for G
      constexpr float fs = 1.f / (R16_SH * R16_SW), gs = 1.f / (R16_SH * R16_SG);
      aux->sh[r][0] = shv[0] * fs; aux->sh[r][1] = shv[1] * fs; aux->sh[r][2] = shv[2] * fs; aux->sh[r][3] = shv[3] * fs;
      aux->shT[0][r] = shv[0] * gs; aux->shT[1][r] = shv[1] * gs; aux->shT[2][r] = shv[2] * gs; aux->shT[3][r] = shv[3] * gs;
    }
  }

  // ---- the segments: blocks in order, the 32-column parts of a block in order; a task of a segment RANGE (ddp_conv_task_t::rows_seg0 / 1)
  // walks only those - its weight stream holds only their tiles
  const int seg0 = T.rows_seg0, seg1 = (T.rows_seg1 > 0) ? T.rows_seg1 : 0x7fffffff;
  int sgi = 0;
  for (int bi = 0; bi < S.nblocks; ++bi) {
    const ddp_block_t& B = S.blk[bi];
    const int nparts = (B.n + 31) >> 5;
    if (sgi >= seg1 || sgi + nparts <= seg0) {      // (none of the block's parts is this task's)
      sgi += nparts;
      continue;
    }
    if (B.ntiles > 0 && B.U > 0 && B.U * B.C <= RL.frows) {
      bool fast = true;     // (vector-input segments of at most 16 features: the factorised shapes of nv <= 16)
      for (int si = 0; si < B.nseg; ++si)
        fast = fast && B.seg[si].count <= 16 && (B.seg[si].kind == DDP_F_DOT || B.seg[si].kind == DDP_F_VEC_S0 || B.seg[si].kind == DDP_F_CROSS);
      if (fast)
        r16_build_features<8>(B, T, aux, F, lane);
      else
        build_features<32, 2>(B, T, aux->src, aux->sh, F, lane);
    }
    for (int part = 0; part < nparts; ++part, ++sgi) {
      if (sgi < seg0 || sgi >= seg1) continue;
      if (B.C == 1)
        t = r16_segment<NS, 1, GF, DIRECT>(RL, B, bi, part, T, ah, al, ring, lbias, t, F, aux, rmask, src, nvw, wave, lane, nts);
      else
        t = r16_segment<NS, 3, GF, DIRECT>(RL, B, bi, part, T, ah, al, ring, lbias, t, F, aux, rmask, src, nvw, wave, lane, nts);
    }
  }
}

template <int SZ, int GF>
__global__ __launch_bounds__(R16_NT, 2) void ddp_conv_rows16_kernel(const R16Launch RL) {
  r16_body<SZ, GF, false>(RL);
}
template <int SZ>
__global__ __launch_bounds__(R16_NT, 2) void ddp_conv_rows16_direct_kernel(const R16Launch RL) {
  r16_body<SZ, 0, true>(RL);
}

// ------------------------------------------------------------------------------------------------ host (called by ddp_conv_rows for rows_form = 1)
int ddp_conv_rows16_launch(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, int sc, void* stream) {
  R16Launch RL;
  ConvLaunch& L = RL.L;
  L.shape = *shape;
  L.r1_floats = 0;
  L.tv_off = 0;
  L.ntasks = 0;
  L.dev_counts = 0;
  const int NS = (sc == 60) ? 12 : 6, nct1 = shape->nct1;
  int nts = nct1, frows = 0;
  for (int b = 0; b < shape->nblocks; ++b) {
    const ddp_block_t& B = shape->blk[b];
    if (B.ntiles > 0 && B.U > 0) {
      nts += ((B.n + 31) / 32) * (B.nsub > 1 ? B.U : (B.U + B.ups - 1) / B.ups);
      if (B.U * B.C > frows) frows = B.U * B.C;
    }
  }
  int tiles = 0;
  for (int i = 0; i < ntasks; ++i) {
    const ddp_conv_task_t& T = tasks[i];
    if (T.n_edges <= 0) continue;
    if (T.gh_fmt != tasks[0].gh_fmt || (unsigned)T.gh_fmt > 1u) return ddp_fail(DDP_EINVAL, "ddp_conv_rows: the tasks of a launch carry one plane form of G (gh_fmt 0 or 1)");
    if (T.rows_seg0 < 0 || T.rows_seg1 < 0 || T.rows_nts < 0 || (T.rows_seg1 > 0 && (T.rows_seg1 <= T.rows_seg0 || T.rows_nts < nct1)) ||
        (T.rows_nts > nts) || (T.rows_seg1 > 0 && !T.rows_bias_k))      // (a range's bias table is fc.0's: rows_bias_k)
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: task.rows_seg0 / rows_seg1 / rows_nts");
    if (T.rows_bias_k != tasks[0].rows_bias_k || (unsigned)T.rows_bias_k > 1u || (T.rows_bias_k && (shape->hid & 15) == 0))
      return ddp_fail(DDP_EINVAL, "ddp_conv_rows: rows_bias_k is 0 or 1 for all tasks of a launch and needs hid % 16 != 0");
    if (T.n_edges_dev) L.dev_counts = 1;
    L.tile_start[L.ntasks] = tiles;
    L.task[L.ntasks] = T;
    tiles += (T.n_edges + R16_ET - 1) / R16_ET;
    ++L.ntasks;
  }
  L.tile_start[L.ntasks] = tiles;
  if (tiles == 0) return 0;
  RL.nts = nts;
  if (frows > R16_FROWS) frows = R16_FROWS;          // (larger blocks build their features in chunks)
  RL.frows = frows;
  int fbytes = frows * R16_FS * 4;
  fbytes = (fbytes + 127) / 128 * 128;
  RL.aux_off = fbytes;
  int priv = fbytes + (int)sizeof(R16Aux);
  if (priv < NS * 1024) priv = NS * 1024;          // the lo plane of edge_attr_ during fc1
  priv = (priv + 127) / 128 * 128;
  RL.priv_bytes = priv;
  RL.bias_tiles = tasks[0].rows_bias_k ? nct1 : nts;
  RL.bias_bytes = (RL.bias_tiles * 128 + 127) / 128 * 128;
  size_t lds_bytes = (size_t)2 * (NS * 1024) + RL.bias_bytes + (size_t)R16_NW * priv;
  if (2 * lds_bytes > 160 * 1024) return ddp_fail(DDP_ELIMIT, "ddp_conv_rows: LDS budget of two workgroups per CU exceeded (too many vector features per block)");
  if ((size_t)ddp_shape_rows_min_lds > lds_bytes) lds_bytes = (size_t)ddp_shape_rows_min_lds;
  static int lds_have[6] = {0, 0, 0, 0, 0, 0};
  hipError_t err;
  const int gf = tasks[0].gh_fmt;
  const bool direct = shape->g_cols[0] == 0 && shape->g_cols[1] == 0;
#define R16_LAUNCH(SZ_, GF_, I_)                                                                                             \
  {                                                                                                                          \
    err = ddp_need_lds(reinterpret_cast<const void*>(ddp_conv_rows16_kernel<SZ_, GF_>), (int)lds_bytes, &lds_have[I_]);      \
    if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(conv rows16)");                                    \
    hipLaunchKernelGGL((ddp_conv_rows16_kernel<SZ_, GF_>), dim3(tiles), dim3(R16_NT), lds_bytes, (hipStream_t)stream, RL);   \
  }
#define R16_LAUNCH_D(SZ_, I_)                                                                                                \
  {                                                                                                                          \
    err = ddp_need_lds(reinterpret_cast<const void*>(ddp_conv_rows16_direct_kernel<SZ_>), (int)lds_bytes, &lds_have[I_]);    \
    if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(conv rows16)");                                    \
    hipLaunchKernelGGL((ddp_conv_rows16_direct_kernel<SZ_>), dim3(tiles), dim3(R16_NT), lds_bytes, (hipStream_t)stream, RL); \
  }
  if (direct) {
    if (sc == 60) R16_LAUNCH_D(60, 4) else R16_LAUNCH_D(32, 5)
  } else if (sc == 60) {
    if (gf == 1) R16_LAUNCH(60, 1, 2) else R16_LAUNCH(60, 0, 0)
  } else {
    if (gf == 1) R16_LAUNCH(32, 1, 3) else R16_LAUNCH(32, 0, 1)
  }
#undef R16_LAUNCH
#undef R16_LAUNCH_D
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_conv_rows (16x16x32 form) launch");
  return 0;
}
