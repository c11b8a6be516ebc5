// ddp_conv_common.h - what the conv kernels of ddp_conv.hip (32- / 64-edge workgroups) and ddp_conv_rows.hip (256-edge, row-stationary
// workgroups) share: vector types, size classes, the fp16 hi/lo split of an operand, the launch descriptor, the workgroup -> tile
// map and the basis features of FasterTensorProduct (reference models/layers.py:40-53).
#ifndef DDP_CONV_COMMON_H
#define DDP_CONV_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


#include "ddp_conv_diag.h"   // STAMP / GSTAMP / DDP_ABL_* : no-ops in the product build


// Size classes: the conv kernels are instantiated per scalar multiplicity ns of the released architectures, with the k-group
// counts of their loops as compile-time constants (fully unrolled tile loops, static register rings: section 4.3 of DESIGN.md).
// One instantiation per class keeps every kernel's register allocation to ITS loops (with all variants inside one kernel the
// allocator spills in the ns = 60 path).  SZ = ns when f_in = hid = 3 ns for ns in {60, 32, 24, 16}; SZ = 0: any shape (runtime loops).
template <int SZ> struct SizeClass { static constexpr int NM = 0; static constexpr bool TAIL = false; };
template <> struct SizeClass<60> { static constexpr int NM = 23; static constexpr bool TAIL = true; };   // hp = 184 = 8 x 23, hid = 180
template <> struct SizeClass<32> { static constexpr int NM = 12; static constexpr bool TAIL = false; };  // README small score model
template <> struct SizeClass<24> { static constexpr int NM = 9; static constexpr bool TAIL = false; };   // confidence model
template <> struct SizeClass<16> { static constexpr int NM = 6; static constexpr bool TAIL = false; };   // BASELINE configs[0]

// ---- fp16 hi/lo split ("h2") forms of the dense fc products.  Both operands of h @ W are split into two
// halves, v = hi + lo / 2048 (hi = fp16(v), lo = fp16((v - hi) * 2048): 22 significant bits, the low part scaled back into fp16's
// normal range), and three products per 16 k run on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//     h w  ~  hh wh + (hh wl + hl wh) / 2048                   dropped: hl wl / 2^22
// Operands of 22 significant bits: a product is off by <= (2^-21 + 2^-22) |h w| in the worst case; measured against fp64 on the fc
// shapes (tools/micro/f16x2_mfma.hip, K = 192): |err| <= 0.85e-7 sum|h w| where the exact fp32 MFMA chain shows 1.8e-7 (the f16
// instruction sums its 16 products before it rounds) - at 3.75 x the fp32 MFMA rate (580 against 155 TFLOP/s sustained).
// The operands keep their byte counts: a weight fragment of 16 k is two 16-byte loads (hi, lo) like two fp32 k-groups.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define DDP_H2_SCALE 2048.f
#define DDP_H2_INV (1.f / 2048.f)
// k16 steps of f_in = hid = 3 SZ (0: no h2 form for this class); LDS row stride of an operand plane = 16 NS + 8 halves
// (16-byte aligned rows whose 16-lane ds_read_b128 groups fall on distinct banks for NS = 12, 6, 5, 3)
template <int SZ> struct H2Class { static constexpr int NS = 0; };
template <> struct H2Class<60> { static constexpr int NS = 12; };
template <> struct H2Class<32> { static constexpr int NS = 6; };
template <> struct H2Class<24> { static constexpr int NS = 5; };
template <> struct H2Class<16> { static constexpr int NS = 3; };

__device__ __forceinline__ void split_h2(const f32x4 v, h4& hi, h4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)((v[i] - (float)hi[i]) * DDP_H2_SCALE);
  }
}
// a value the h2 form cannot split (outside the fp16 range, or NaN) is REPORTED, never silently saturated: ddp_conv_task_t::h2_range_flag
__device__ __forceinline__ void h2_range_check(float v, int32_t* flag) {
  if (!(fabsf(v) <= 65504.f) && flag) *flag = 1;
}

struct ConvLaunch {
  ddp_conv_shape_t shape;
  int r1_floats;   // h2 kernels: floats of the first LDS region (operand planes of h; 32-edge kernel: later h in fp32)
  int tv_off;   // 32-edge kernels: row stride (floats) of the LDS message tile; unused by the 64-edge kernel
  int ntasks;
  int dev_counts;   // some task carries n_edges_dev: tile table rebuilt on the device (conv_tile)
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
template <int ET, int NG = 8>
__device__ __forceinline__ void build_features(const ddp_block_t& B, const ddp_conv_task_t& T, const int* s_src,
                                               const float (*s_sh)[4], float* fbuf, int tid) {
  constexpr int FS = ET + 4;
  const int e = tid & (ET - 1), wave = tid / ET;   // NG thread groups of ET threads
  const float* xrow = T.x_src + (size_t)s_src[e] * T.ldx_src;
  const float s0 = s_sh[e][0], sx = s_sh[e][1], sy = s_sh[e][2], sz = s_sh[e][3];
  const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt2 = 0.70710678118654752f;
  int ubase = 0;
  for (int si = 0; si < B.nseg; ++si) {
    const int kind = B.seg[si].kind, off = B.seg[si].in_off, cnt = B.seg[si].count;
    for (int ul = wave; ul < cnt; ul += NG) {
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

// workgroup id -> tile.  XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so
// id -> tile is remapped to give every XCD one contiguous range of tiles: neighbouring tiles share source nodes (G rows,
// x rows) and all tiles of a conv share its packed weights.
__device__ __forceinline__ int xcd_tile(int ntl) {
  const int q = ntl >> 3, rem = ntl & 7, x = (int)blockIdx.x & 7;
  return ((x < rem) ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + ((int)blockIdx.x >> 3);
}

// workgroup -> (task, first edge, valid edges).  Host-side counts: the tile table of the launch.  Device-side counts
// (ConvLaunch::dev_counts; include/ddp_hip.h "Device-side counts"): the grid covers the tasks' CAPACITIES, every workgroup
// reads the actual edge counts, rebuilds the tile table from them and leaves if it lies behind the last tile - the XCD-aware
// order is then the one of a launch of exactly that many tiles.
template <int ET>
__device__ __forceinline__ bool conv_tile(const ConvLaunch& L, int& t, int& p0, int& nvalid) {
  if (!L.dev_counts) {
    const int tile = xcd_tile((int)gridDim.x);
    t = 0;
    while (t + 1 < L.ntasks && tile >= L.tile_start[t + 1]) ++t;
    p0 = (tile - L.tile_start[t]) * ET;
    nvalid = min(ET, L.task[t].n_edges - p0);
    return true;
  }
  // all counts first (independent scalar loads, one round trip), then the prefix arithmetic in registers
  int cnt[DDP_MAX_TASKS];
#pragma unroll
  for (int i = 0; i < DDP_MAX_TASKS; ++i) {
    int n = 0;
    if (i < L.ntasks) {
      const ddp_conv_task_t& T = L.task[i];
      n = T.n_edges_dev ? max(0, min(*T.n_edges_dev, T.n_edges)) : T.n_edges;
    }
    cnt[i] = n;
  }
  int total = 0;
#pragma unroll
  for (int i = 0; i < DDP_MAX_TASKS; ++i) total += (cnt[i] + ET - 1) / ET;
  if ((int)blockIdx.x >= total) return false;
  const int tile = xcd_tile(total);
  int base = 0;
  t = 0;
  p0 = 0;
  nvalid = 0;
#pragma unroll
  for (int i = 0; i < DDP_MAX_TASKS; ++i) {
    const int nt = (cnt[i] + ET - 1) / ET;
    if (tile >= base && tile < base + nt) {
      t = i;
      p0 = (tile - base) * ET;
      nvalid = min(ET, cnt[i] - p0);
    }
    base += nt;
  }
  return true;
}

#endif /* DDP_CONV_COMMON_H */
