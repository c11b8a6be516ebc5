// ddp_conv.hip - fused per-edge fc -> Clebsch-Gordan tensor product -> message kernel for gfx950 (MI355X).
//
// Replaces, per conv (reference file:line):
//   edge_attr_ = cat(edge_attr, x_recv[:, :ns], x_src[:, :ns])     models/all_atom_score_model.py:273-312
//   w = fc(edge_attr_) = Linear -> ReLU -> Linear   [E, weight_numel]  models/score_model.py:100-105,114
//   msg = FasterTensorProduct(x[src], sh, w)                        models/layers.py:34-85
// The reference materialises w (40 kB per edge at ns=60); here a workgroup owns 64 edges (CSR order of the
// receiving node), keeps h = relu(fc1) for them in LDS, streams the packed fc2 weight through fp32 MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains) one 32-column tile at a time and contracts each tile with
// the tensor-product basis features straight from the accumulator registers:  w never leaves the CU.
//
// Work decomposition (one launch = the <=9 convs of a layer, they share one shape):
//   workgroup = 512 threads = 8 waves (two per SIMD, 256 registers each), 64 edges.
//   phase 0  stage edge_attr_ rows (3 gathers) into LDS
//   phase 1  h = relu(edge_attr_ @ W1 + b1)  via MFMA, to LDS [64][hs]
//   per weight block (0e,1o,1e,0o):
//     phase 2  basis features F[u][c][e] from gathered x[src] and sh  -> LDS
//     phase 3  waves split the block's column tiles; for a tile: acc[e, col] = h @ W2p[:, tile] + b2p (MFMA,
//              K = hid), then out[e, n(,c)] += F[e, u(,c)] * acc[e, (u,n)] in registers (C/D layout of the MFMA)
//     phase 4  ordered (deterministic) cross-wave / cross-lane reduction in LDS, coalesced store of the
//              block's message columns.
// Packed weight layout (built once on the host, ddp_pack.py): tile-major, inside a tile K is interleaved so that
// one global_load_dwordx4 per lane (1 KiB per wave, fully coalesced) feeds 4 consecutive MFMA k-steps:
//   w2p[((tile * (hp/8) + m) * 2 + hh) * 32 + j][i] = W[k = 8m + 4hh + i][column(tile, j)]
// and the A operand is read from LDS with one ds_read_b128 per lane for the same 4 k-steps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FS 68  // LDS row stride (floats) of the feature buffer F[u*C + c][e], e < 64
#define DDP_CONV_THREADS 512  // 8 waves: two per SIMD

struct ConvLaunch {
  ddp_conv_shape_t shape;
  int ntasks;
  int tile_start[DDP_MAX_TASKS + 1];
  ddp_conv_task_t task[DDP_MAX_TASKS];
};

__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v;
  return r;
}

// ------------------------------------------------------------------------------------------------ phase 2
__device__ __forceinline__ void build_features(const ddp_block_t& B, const ddp_conv_task_t& T, const int* s_src,
                                               const int* s_eid, float* fbuf, int tid) {
  const int e = tid & 63, wave = tid >> 6;
  const float* xrow = T.x_src + (size_t)s_src[e] * T.ldx_src;
  const f32x4 shv = reinterpret_cast<const f32x4*>(T.sh)[s_eid[e]];
  const float s0 = shv[0], sx = shv[1], sy = shv[2], sz = shv[3];
  const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt2 = 0.70710678118654752f;
  int ubase = 0;
  for (int si = 0; si < B.nseg; ++si) {
    const int kind = B.seg[si].kind, off = B.seg[si].in_off, cnt = B.seg[si].count;
    for (int ul = wave; ul < cnt; ul += DDP_CONV_THREADS / 64) {
      const int u = ubase + ul;
      if (kind == DDP_F_SCALAR_S0) {
        fbuf[u * FS + e] = xrow[off + ul] * s0;
      } else if (kind == DDP_F_DOT) {
        const float ax = xrow[off + 3 * ul], ay = xrow[off + 3 * ul + 1], az = xrow[off + 3 * ul + 2];
        fbuf[u * FS + e] = (ax * sx + ay * sy + az * sz) * inv_sqrt3;
      } else if (kind == DDP_F_SCALAR_S1) {
        const float a = xrow[off + ul];
        fbuf[(u * 3 + 0) * FS + e] = a * sx;
        fbuf[(u * 3 + 1) * FS + e] = a * sy;
        fbuf[(u * 3 + 2) * FS + e] = a * sz;
      } else if (kind == DDP_F_VEC_S0) {
        fbuf[(u * 3 + 0) * FS + e] = xrow[off + 3 * ul] * s0;
        fbuf[(u * 3 + 1) * FS + e] = xrow[off + 3 * ul + 1] * s0;
        fbuf[(u * 3 + 2) * FS + e] = xrow[off + 3 * ul + 2] * s0;
      } else {  // DDP_F_CROSS: a x s1 / sqrt(2)
        const float ax = xrow[off + 3 * ul], ay = xrow[off + 3 * ul + 1], az = xrow[off + 3 * ul + 2];
        fbuf[(u * 3 + 0) * FS + e] = (ay * sz - az * sy) * inv_sqrt2;
        fbuf[(u * 3 + 1) * FS + e] = (az * sx - ax * sz) * inv_sqrt2;
        fbuf[(u * 3 + 2) * FS + e] = (ax * sy - ay * sx) * inv_sqrt2;
      }
    }
    ubase += cnt;
  }
}

// column -> (feature u, output channel n, valid) of lane column r in tile t of block B
__device__ __forceinline__ void tile_lane_map(const ddp_block_t& B, int t, int r, int& u, int& ncol, int& us, bool& valid) {
  if (B.nsub > 1) {
    u = t / B.nsub;
    const int sub = t - u * B.nsub;
    ncol = sub * 32 + r;
    us = 0;
    valid = (ncol < B.n) && (u < B.U);
  } else {
    us = r / B.n;
    ncol = r - us * B.n;
    u = t * B.ups + us;
    valid = (us < B.ups) && (u < B.U);
  }
  if (!valid) { u = 0; ncol = 0; }
}

// ------------------------------------------------------------------------------------------------ phases 3+4
// 8 waves = 2 per SIMD.  Wave w owns the 32 edges of row-tile rt = w >> 2 and, inside the block, the tile groups
// g = (w & 3), (w & 3) + 4, ...  (so the two waves of a SIMD cover each other's waits and epilogues).
// C = 1: scalar block, tiles in pairs (1x2 register blocking: 32 edges x 64 columns per wave step)
// C = 3: vector block, single tiles with three output accumulators (x,y,z) per edge row
template <int C>
__device__ __forceinline__ void run_block(const ddp_conv_shape_t& S, const ddp_block_t& B, const ddp_conv_task_t& T,
                                          const float* hbuf, float* fbuf, int tid, int p0, int nvalid) {
  constexpr int CT = (C == 1) ? 2 : 1;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int rt = wave >> 2, wq = wave & 3;
  const int nm = S.hp >> 3;
  const int ngroups = B.ntiles / CT;  // host pads scalar blocks to an even tile count
  const f32x4* __restrict__ w2p = reinterpret_cast<const f32x4*>(T.w2p);
  const float* arow = &hbuf[(rt * 32 + r) * S.hs + 4 * hh];

  f32x16 out[CT][C];
#pragma unroll
  for (int s = 0; s < CT; ++s)
#pragma unroll
    for (int c = 0; c < C; ++c) out[s][c] = splat16(0.f);

  // B-operand prefetch, flattened over (group, m)
  f32x4 bnext[CT];
  if (wq < ngroups) {
#pragma unroll
    for (int s = 0; s < CT; ++s) {
      const int tile = B.tile0 + wq * CT + s;
      bnext[s] = w2p[((size_t)tile * nm * 2 + hh) * 32 + r];
    }
  }
  f32x4 anext = *reinterpret_cast<const f32x4*>(arow);
  for (int g = wq; g < ngroups; g += 4) {
    f32x16 acc[CT];
#pragma unroll
    for (int s = 0; s < CT; ++s) acc[s] = splat16(T.b2p[(B.tile0 + g * CT + s) * 32 + r]);
    for (int m = 0; m < nm; ++m) {
      f32x4 bcur[CT];
#pragma unroll
      for (int s = 0; s < CT; ++s) bcur[s] = bnext[s];
      {  // prefetch the next (group, m)
        int gn = g, mn = m + 1;
        if (mn == nm) { gn = g + 4; mn = 0; }
        if (gn < ngroups) {
#pragma unroll
          for (int s = 0; s < CT; ++s) {
            const int tile = B.tile0 + gn * CT + s;
            bnext[s] = w2p[(((size_t)tile * nm + mn) * 2 + hh) * 32 + r];
          }
        }
      }
      const f32x4 a = anext;
      anext = *reinterpret_cast<const f32x4*>(arow + 8 * ((m + 1 == nm) ? 0 : m + 1));  // h is tile independent: wrap
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < CT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[s][i], acc[s], 0, 0, 0);
    }
    // contraction with the basis features, in the MFMA C/D layout: reg i <-> edge row (i&3) + 8*(i>>2) + 4*hh
#pragma unroll
    for (int s = 0; s < CT; ++s) {
      int u, ncol, us;
      bool valid;
      tile_lane_map(B, g * CT + s, r, u, ncol, us, valid);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float* frow = &fbuf[(u * C + c) * FS + rt * 32 + 4 * hh];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 f = *reinterpret_cast<const f32x4*>(frow + 8 * q4);
#pragma unroll
          for (int q = 0; q < 4; ++q) out[s][c][4 * q4 + q] += f[q] * acc[s][4 * q4 + q];
        }
      }
    }
  }

  // ---- phase 4: deterministic cross-wave / cross-lane reduction.  Every wave parks its 32-row partial tile in its own
  // LDS region with plain stores (all waves concurrently); then all threads sum the 4 regions of a row-tile (and, for
  // n <= 32, the `ups` lane groups and both tile slots) in a FIXED order and store the block's message columns
  // coalesced.  Vector blocks do this in two passes (row-tile 0, then 1) to stay inside fbuf.
  constexpr int NP = (C == 1) ? 1 : 2;            // passes
  constexpr int RW = CT * 32 * C;                 // floats per edge row in a wave region: [c][slot*32 + r]
  constexpr int REGION = 32 * RW;                 // 2048 (scalar) / 3072 (vector) floats per wave
  const int nc = B.n * C;
  float* part = fbuf;
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    __syncthreads();  // F (or the previous pass's partials) no longer needed
    if (NP == 1 || rt == pass) {
      float* mine = part + ((NP == 1) ? wave : wq) * REGION;
#pragma unroll
      for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
            mine[row * RW + c * (CT * 32) + s * 32 + r] = out[s][c][i];
          }
    }
    __syncthreads();
    const int rows = (NP == 1) ? 64 : 32;
    for (int idx = tid; idx < rows * nc; idx += DDP_CONV_THREADS) {
      const int el = idx / nc, cc = idx - el * nc;
      const int ncol = cc / C, c = cc - ncol * C;
      const int e = (NP == 1) ? el : pass * 32 + el;
      const int wbase = (NP == 1) ? (e >> 5) * 4 : 0;
      float sum = 0.f;
      for (int w = 0; w < 4; ++w) {
        const float* reg = part + (wbase + w) * REGION + (e & 31) * RW + c * (CT * 32);
        if (B.nsub > 1) {
          sum += reg[ncol];                       // column = sub*32 + r = ncol
        } else {
          for (int sl = 0; sl < CT; ++sl)
            for (int k = 0; k < B.ups; ++k) sum += reg[sl * 32 + k * B.n + ncol];
        }
      }
      if (e < nvalid) T.msg[(size_t)(p0 + e) * S.d_out + B.out_off + cc] = sum;
    }
  }
  __syncthreads();  // fbuf is rewritten by the next block's features
}

// ------------------------------------------------------------------------------------------------ kernel
__global__ __launch_bounds__(DDP_CONV_THREADS, 2) void ddp_conv_messages_kernel(const ConvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_src[64], s_eid[64];
  const ddp_conv_shape_t& S = L.shape;
  const int tid = threadIdx.x;
  int t = 0;
  while (t + 1 < L.ntasks && (int)blockIdx.x >= L.tile_start[t + 1]) ++t;
  const ddp_conv_task_t& T = L.task[t];
  const int p0 = ((int)blockIdx.x - L.tile_start[t]) * DDP_EDGE_TILE;
  const int nvalid = min(DDP_EDGE_TILE, T.n_edges - p0);
  float* hbuf = lds;
  float* fbuf = lds + 64 * S.hs;
  float* xa = fbuf;  // edge_attr_ staging aliases the feature buffer

  // ---- phase 0: indices + edge_attr_ rows
  if (tid < 64) {
    const int p = p0 + min(tid, nvalid - 1);
    s_src[tid] = T.src[p];
    s_eid[tid] = T.eid[p];
  }
  int col0 = 0;
#pragma unroll
  for (int sg = 0; sg < DDP_MAX_SEGS; ++sg) {
    const int n = T.seg_n[sg];
    if (n > 0) {
      const float* __restrict__ ptr = T.seg_ptr[sg];
      const int* __restrict__ idx = T.seg_idx[sg];
      const int ld = T.seg_ld[sg];
      for (int i = tid; i < 64 * n; i += DDP_CONV_THREADS) {
        const int e = i / n, c = i - e * n;
        const int row = idx[p0 + min(e, nvalid - 1)];
        xa[e * S.hs + col0 + c] = ptr[(size_t)row * ld + c];
      }
      col0 += n;
    }
  }
  {
    const int npad = S.kp1 - S.f_in;
    for (int i = tid; i < 64 * npad; i += DDP_CONV_THREADS) {
      const int e = i / npad, c = i - e * npad;
      xa[e * S.hs + S.f_in + c] = 0.f;
    }
  }
  __syncthreads();

  // ---- phase 1: h = relu(edge_attr_ @ W1 + b1); wave w: row-tile w >> 2, column tiles (w & 3), (w & 3) + 4, ...
  {
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int rt = wave >> 2;
    const int nm1 = S.kp1 >> 3;
    const f32x4* __restrict__ w1p = reinterpret_cast<const f32x4*>(T.w1p);
    for (int ct = wave & 3; ct < S.nct1; ct += 4) {
      f32x16 acc = splat16(T.b1p[ct * 32 + r]);
      for (int m = 0; m < nm1; ++m) {
        const f32x4 b = w1p[(((size_t)ct * nm1 + m) * 2 + hh) * 32 + r];
        const f32x4 a = *reinterpret_cast<const f32x4*>(&xa[(rt * 32 + r) * S.hs + 8 * m + 4 * hh]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
      }
      const int col = ct * 32 + r;
      if (col < S.hp) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          hbuf[row * S.hs + col] = fmaxf(acc[i], 0.f);
        }
      }
    }
  }
  __syncthreads();

  // ---- per weight block
  for (int bi = 0; bi < S.nblocks; ++bi) {
    const ddp_block_t& B = S.blk[bi];
    build_features(B, T, s_src, s_eid, fbuf, tid);
    __syncthreads();
    if (B.C == 1)
      run_block<1>(S, B, T, hbuf, fbuf, tid, p0, nvalid);
    else
      run_block<3>(S, B, T, hbuf, fbuf, tid, p0, nvalid);
  }
}

// ------------------------------------------------------------------------------------------------ host
extern "C" int ddp_conv_messages(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, void* stream) {
  if (!shape || !tasks) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: null argument");
  if (ntasks < 0 || ntasks > DDP_MAX_TASKS) return ddp_fail(DDP_ELIMIT, "ddp_conv_messages: ntasks > DDP_MAX_TASKS");
  if (shape->nblocks < 1 || shape->nblocks > DDP_MAX_BLOCKS) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: nblocks");
  if ((shape->kp1 & 7) || (shape->hp & 7) || (shape->hs & 3) || shape->hs < shape->kp1 || shape->hs < shape->hp)
    return ddp_fail(DDP_EINVAL, "ddp_conv_messages: kp1/hp must be multiples of 8 and hs >= both");
  for (int b = 0; b < shape->nblocks; ++b) {
    const ddp_block_t& B = shape->blk[b];
    if (B.C != 1 && B.C != 3) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: block C must be 1 or 3");
    if (B.n < 1 || B.n > 64 || (B.C == 3 && B.n > 32)) return ddp_fail(DDP_ELIMIT, "ddp_conv_messages: block n too large");
    if (B.nsub < 1 || B.nsub > 2 || B.ups < 1) return ddp_fail(DDP_ELIMIT, "ddp_conv_messages: nsub/ups");
    if (B.C == 1 && (B.ntiles & 1)) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: scalar blocks need an even tile count");
    if (B.U * B.C * FS > shape->fbuf_floats || 4 * 4096 > shape->fbuf_floats)
      return ddp_fail(DDP_EINVAL, "ddp_conv_messages: fbuf_floats too small");
    if (B.nseg < 0 || B.nseg > DDP_MAX_SEGS) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: nseg");
  }
  if (64 * shape->hs > shape->fbuf_floats) return ddp_fail(DDP_EINVAL, "ddp_conv_messages: fbuf_floats < staging tile");
  ConvLaunch L;
  L.shape = *shape;
  L.ntasks = 0;
  int tiles = 0;
  for (int i = 0; i < ntasks; ++i) {
    if (tasks[i].n_edges <= 0) continue;  // an empty conv sends no message (models/score_model.py:109-111)
    L.tile_start[L.ntasks] = tiles;
    L.task[L.ntasks] = tasks[i];
    tiles += (tasks[i].n_edges + DDP_EDGE_TILE - 1) / DDP_EDGE_TILE;
    ++L.ntasks;
  }
  L.tile_start[L.ntasks] = tiles;
  if (tiles == 0) return 0;
  const size_t lds_bytes = (size_t)(64 * shape->hs + shape->fbuf_floats) * sizeof(float);
  if (lds_bytes > 160 * 1024 - 1024) return ddp_fail(DDP_ELIMIT, "ddp_conv_messages: LDS budget exceeded");
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(ddp_conv_messages_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(conv)");
  hipLaunchKernelGGL(ddp_conv_messages_kernel, dim3(tiles), dim3(DDP_CONV_THREADS), lds_bytes, (hipStream_t)stream, L);
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_conv_messages launch");
  return 0;
}
