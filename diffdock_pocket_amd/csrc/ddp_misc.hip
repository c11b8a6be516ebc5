// ddp_misc.hip - the HBM-bound kernels around the fused conv (gfx950):
//   ddp_segment_reduce   CSR segmented mean + e3nn BatchNorm(eval) + residual accumulate
//   ddp_edge_featurize   edge vector -> RBF -> 2-layer MLP, spherical harmonics (lmax = 1)
//   ddp_torsion_sh       closed-form 1o block of FullTensorProduct(sh, Y2(bond))
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ segment reduce
// Replaces torch_scatter.scatter(reduce='mean') (reference models/score_model.py:117), e3nn BatchNorm in eval
// mode (:123-124) and the pad + add residual (models/all_atom_score_model.py:315-324).
// One workgroup per receiving node, one thread per output channel: the node's incoming messages are consecutive
// rows of `msg` (CSR order), so every row read is a fully coalesced d_out*4-byte burst and the sum order is the
// CSR order (deterministic).  Algorithmic bytes: 4*d_out per edge read + 8*d_out per node (x read + write).
struct ReduceLaunch {
  ddp_reduce_src_t src[3];
  int nsrc;
};

__global__ void ddp_segment_reduce_kernel(float* __restrict__ x, int ldx, int n_nodes, int d_out, ReduceLaunch L,
                                          int accumulate, int n_rep, int rep_stride) {
  const int node = blockIdx.x;
  const int ch = threadIdx.x;
  if (ch >= d_out) return;
  float* dst = x + (size_t)node * ldx + ch;
  // same association as the reference's `x + u_a + u_b + u_c` (all_atom_score_model.py:316,320,324)
  float total = (accumulate && n_rep <= 1) ? *dst : 0.f;
  // all row pointers first, then the message rows in batches of 8 independent loads that are summed IN ORDER: a node has
  // ~8 incoming edges per conv, so the whole segment is one memory round trip instead of a chain of them
  int p0[3], p1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool on = k < L.nsrc;
    p0[k] = on ? L.src[k].rowptr[node] : 0;
    p1[k] = on ? L.src[k].rowptr[node + 1] : 0;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // (a conv whose device-side edge count is 0 contributes exactly 0, like one the host drops: score_model.py:109-111)
    if (k < L.nsrc && !(L.src[k].n_edges_dev && *L.src[k].n_edges_dev <= 0)) {
      const ddp_reduce_src_t& s = L.src[k];
      const float* __restrict__ m = s.msg + ch;
      const int32_t* __restrict__ rm = s.rowmap;
      float sum = 0.f;
      for (int p = p0[k]; p < p1[k]; p += 8) {
        float v[8];
        int row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = min(p + i, p1[k] - 1);
          row[i] = rm ? rm[q] : q;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = m[(size_t)row[i] * d_out];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (p + i < p1[k]) sum += v[i];
      }
      const int cnt = p1[k] - p0[k];
      const float mean = sum / (float)(cnt > 1 ? cnt : 1);
      total += mean * s.bn_scale[ch] + s.bn_shift[ch];
    }
  }
  if (n_rep <= 1) {
    *dst = total;
  } else {   // the update (summed from 0) added to the node's copy in every graph of the batch
    for (int g = 0; g < n_rep; ++g) {
      float* d = dst + (size_t)g * rep_stride * ldx;
      *d = *d + total;
    }
  }
}

// The same reduction with four channels per thread (16-byte loads and stores): thread -> (node, channel quad), consecutive lanes
// read consecutive quads of a message row.  Every element sees exactly the operations of the kernel above, in the same order
// (bitwise the same result); four times the bytes per load instruction in flight.  Needs d_out, ldx multiples of 4 and 16-byte
// aligned arrays.
typedef float red4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ddp_segment_reduce4_kernel(float* __restrict__ x, int ldx, int n_nodes, int d_out, ReduceLaunch L,
                                                                  int accumulate, int n_rep, int rep_stride) {
  const int nq = d_out >> 2;
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  const int node = (int)(item / nq), ch = 4 * (int)(item - (long long)node * nq);
  if (node >= n_nodes) return;
  float* dst = x + (size_t)node * ldx + ch;
  red4 total = (accumulate && n_rep <= 1) ? *reinterpret_cast<const red4*>(dst) : red4{0.f, 0.f, 0.f, 0.f};
  int p0[3], p1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool on = k < L.nsrc;
    p0[k] = on ? L.src[k].rowptr[node] : 0;
    p1[k] = on ? L.src[k].rowptr[node + 1] : 0;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k < L.nsrc && !(L.src[k].n_edges_dev && *L.src[k].n_edges_dev <= 0)) {
      const ddp_reduce_src_t& s = L.src[k];
      const float* __restrict__ m = s.msg + ch;
      const int32_t* __restrict__ rm = s.rowmap;
      red4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int p = p0[k]; p < p1[k]; p += 8) {
        red4 v[8];
        int row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = min(p + i, p1[k] - 1);
          row[i] = rm ? rm[q] : q;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const red4*>(m + (size_t)row[i] * d_out);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (p + i < p1[k]) sum += v[i];
      }
      const int cnt = p1[k] - p0[k];
      const float den = (float)(cnt > 1 ? cnt : 1);
      const red4 sc = *reinterpret_cast<const red4*>(s.bn_scale + ch), sh = *reinterpret_cast<const red4*>(s.bn_shift + ch);
#pragma unroll
      for (int c = 0; c < 4; ++c) total[c] += (sum[c] / den) * sc[c] + sh[c];
    }
  }
  if (n_rep <= 1) {
    *reinterpret_cast<red4*>(dst) = total;
  } else {
    for (int g = 0; g < n_rep; ++g) {
      red4* d = reinterpret_cast<red4*>(dst + (size_t)g * rep_stride * ldx);
      *d = *d + total;
    }
  }
}

extern "C" int ddp_segment_reduce(float* x, int ldx, int n_nodes, int d_out, const ddp_reduce_src_t* srcs, int nsrc,
                                  int accumulate, int n_rep, int rep_stride, void* stream) {
  if (!x || (!srcs && nsrc > 0)) return ddp_fail(DDP_EINVAL, "ddp_segment_reduce: null argument");
  if (nsrc < 0 || nsrc > 3) return ddp_fail(DDP_ELIMIT, "ddp_segment_reduce: nsrc > 3");
  if (d_out < 1 || d_out > 1024) return ddp_fail(DDP_ELIMIT, "ddp_segment_reduce: d_out");
  if (n_nodes <= 0) return 0;
  ReduceLaunch L;
  L.nsrc = 0;
  for (int i = 0; i < nsrc; ++i)
    if (srcs[i].n_edges > 0) L.src[L.nsrc++] = srcs[i];  // an empty conv contributes exactly 0 (score_model.py:109-111)
  if (L.nsrc == 0 && (accumulate || n_rep > 1)) return 0;
  if (n_rep > 1 && rep_stride < n_nodes) return ddp_fail(DDP_EINVAL, "ddp_segment_reduce: rep_stride < n_nodes");
  bool wide = ((d_out | ldx) & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
  for (int i = 0; i < L.nsrc; ++i)
    wide = wide && ((reinterpret_cast<size_t>(L.src[i].msg) | reinterpret_cast<size_t>(L.src[i].bn_scale) | reinterpret_cast<size_t>(L.src[i].bn_shift)) & 15) == 0;
  static const bool narrow_only = getenv("DDP_REDUCE_NARROW") != nullptr;   // diagnostic: the one-channel-per-thread form
  if (wide && !narrow_only) {
    const long long items = (long long)n_nodes * (d_out >> 2);
    hipLaunchKernelGGL(ddp_segment_reduce4_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, n_nodes,
                       d_out, L, accumulate, n_rep, rep_stride);
  } else {
    const int threads = ((d_out + 63) / 64) * 64;
    hipLaunchKernelGGL(ddp_segment_reduce_kernel, dim3(n_nodes), dim3(threads), 0, (hipStream_t)stream, x, ldx, n_nodes,
                       d_out, L, accumulate, n_rep, rep_stride);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_segment_reduce launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ edge featurise
// One thread per edge; the (zero-padded to 64) MLP weights live in LDS and are read as wave-wide broadcasts.
// Output rows are transposed through LDS so the [E, ns] store is coalesced.
#define EF_NS 64
__global__ __launch_bounds__(256) void ddp_edge_featurize_kernel(
    const float* __restrict__ pos_a, const int* __restrict__ ia, const float* __restrict__ pos_b,
    const int* __restrict__ ib, int n_edges, const int* __restrict__ n_edges_dev, const float* __restrict__ offset, int k_rbf,
    float coeff, const float* __restrict__ pre, const int* __restrict__ pre_idx, int ld_pre, const float* __restrict__ pre2,
    int n_pre2, int ld_pre2, const float* __restrict__ w1d, const float* __restrict__ w2, const float* __restrict__ b2, int ns,
    float* __restrict__ out, float* __restrict__ sh) {
  if (n_edges_dev) n_edges = min(n_edges, *n_edges_dev);   // device-side edge count: n_edges is then the capacity
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w1 = smem;                       // [k_rbf][64]
  float* s_w2 = s_w1 + k_rbf * EF_NS;       // [64][64]
  float* s_b2 = s_w2 + EF_NS * EF_NS;       // [64]
  float* s_off = s_b2 + EF_NS;              // [k_rbf]
  float* s_out = s_off + ((k_rbf + 3) & ~3);  // [4 waves][64 edges][65]
  const int tid = threadIdx.x;
  for (int i = tid; i < k_rbf * EF_NS; i += 256) s_w1[i] = w1d[i];
  for (int i = tid; i < EF_NS * EF_NS; i += 256) s_w2[i] = w2[i];
  if (tid < EF_NS) s_b2[tid] = b2[tid];
  for (int i = tid; i < k_rbf; i += 256) s_off[i] = offset[i];
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  float* my_out = s_out + wave * 64 * 65;
  for (int base = blockIdx.x * 256; base < n_edges; base += gridDim.x * 256) {
    const int e = base + tid;
    const bool live = e < n_edges;
    const int ec = live ? e : n_edges - 1;
    const int a = ia[ec], b = ib[ec];
    const float vx = pos_b[3 * b] - pos_a[3 * a], vy = pos_b[3 * b + 1] - pos_a[3 * a + 1],
                vz = pos_b[3 * b + 2] - pos_a[3 * a + 2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);  // sqrt(3) / max(|v|, eps)  (F.normalize eps)
    if (live) reinterpret_cast<f32x4*>(sh)[e] = f32x4{1.0f, vx * inv, vy * inv, vz * inv};

    float hid[EF_NS];
    const float* prow = pre + (size_t)pre_idx[ec] * ld_pre;
#pragma unroll
    for (int j = 0; j < EF_NS; ++j) hid[j] = (j < ns) ? prow[j] : 0.f;
    if (ec < n_pre2) {
      const float* p2 = pre2 + (size_t)ec * ld_pre2;
#pragma unroll
      for (int j = 0; j < EF_NS; ++j)
        if (j < ns) hid[j] = hid[j] + p2[j];
    }
    for (int k = 0; k < k_rbf; ++k) {
      const float t = d - s_off[k];
      const float rb = expf(coeff * (t * t));
      const f32x4* wrow = reinterpret_cast<const f32x4*>(s_w1 + k * EF_NS);
#pragma unroll
      for (int j4 = 0; j4 < EF_NS / 4; ++j4) {
        const f32x4 w = wrow[j4];
        hid[4 * j4 + 0] += rb * w[0];
        hid[4 * j4 + 1] += rb * w[1];
        hid[4 * j4 + 2] += rb * w[2];
        hid[4 * j4 + 3] += rb * w[3];
      }
    }
    // second layer, 16 outputs at a time to bound register use; results go to LDS transposed
#pragma unroll
    for (int oc = 0; oc < EF_NS / 16; ++oc) {
      float o[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = s_b2[16 * oc + q];
#pragma unroll
      for (int j = 0; j < EF_NS; ++j) {
        const float hj = fmaxf(hid[j], 0.f);
        const f32x4* wrow = reinterpret_cast<const f32x4*>(s_w2 + j * EF_NS + 16 * oc);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 w = wrow[q4];
          o[4 * q4 + 0] += hj * w[0];
          o[4 * q4 + 1] += hj * w[1];
          o[4 * q4 + 2] += hj * w[2];
          o[4 * q4 + 3] += hj * w[3];
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) my_out[lane * 65 + 16 * oc + q] = o[q];
    }
    __builtin_amdgcn_wave_barrier();
    // coalesced store of this wave's 64 rows x ns columns
    const int ebase = base + wave * 64;
    for (int i = lane; i < 64 * ns; i += 64) {
      const int row = i / ns, col = i - row * ns;
      if (ebase + row < n_edges) out[(size_t)(ebase + row) * ns + col] = my_out[row * 65 + col];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// MFMA form of the same op for k_rbf in {8, 16, .., 64}: a wave owns 32 edges; both Linears run on v_mfma_f32_32x32x2_f32
// (exact fp32 FMA chains) with the packed weights in LDS:  lane (r = lane & 31, hh = lane >> 5) feeds A[r][8m + 4hh + i] -
// for the first layer the Gaussian of edge r evaluated on the fly (each half-wave evaluates half of the k's of an edge, no
// exp is computed twice) - and B[8m + 4hh + i][32 ct + r]; C/D register i <-> edge row (i&3) + 8(i>>2) + 4hh, column r.
// The first layer's accumulators start from the `pre` rows (sigma / bond-type part of the Linear + bias), relu(hidden)
// goes through LDS once to turn the C/D layout into the A layout of the second layer.
typedef float ef_f32x16 __attribute__((ext_vector_type(16)));
#define EF_HS 68   // LDS row stride of the hidden tile
struct FeatLaunch {
  int njobs;
  int blk_start[DDP_MAX_FEATURIZE_JOBS + 1];
  ddp_featurize_job_t job[DDP_MAX_FEATURIZE_JOBS];
};

// One launch for several edge sets (ddp_edge_featurize_jobs): a workgroup belongs to one job - the block ranges of
// FeatLaunch::blk_start - stages that job's weights and walks that job's edges.
__global__ __launch_bounds__(256) void ddp_edge_featurize_mfma_kernel(const FeatLaunch L) {
  int j = 0;
  while (j + 1 < L.njobs && (int)blockIdx.x >= L.blk_start[j + 1]) ++j;
  const ddp_featurize_job_t& J = L.job[j];
  const int blk = (int)blockIdx.x - L.blk_start[j], nblk = L.blk_start[j + 1] - L.blk_start[j];
  const float* __restrict__ pos_a = J.pos_a;
  const int* __restrict__ ia = J.ia;
  const float* __restrict__ pos_b = J.pos_b;
  const int* __restrict__ ib = J.ib;
  const float* __restrict__ offset = J.offset;
  const float* __restrict__ pre = J.pre;
  const int* __restrict__ pre_idx = J.pre_idx;
  const float* __restrict__ pre2 = J.pre2;
  const float* __restrict__ w1d = J.w1d;
  const float* __restrict__ w2 = J.w2;
  const float* __restrict__ b2 = J.b2;
  float* __restrict__ out = J.out;
  float* __restrict__ sh = J.sh;
  const int k_rbf = J.k_rbf, ld_pre = J.ld_pre, n_pre2 = J.n_pre2, ld_pre2 = J.ld_pre2, ns = J.ns;
  const float coeff = J.coeff;
  int n_edges = J.n_edges;
  if (J.n_edges_dev) n_edges = min(n_edges, *J.n_edges_dev);   // device-side edge count: n_edges is then the capacity
  if (blk * 128 >= n_edges) return;                              // (a capacity-sized grid: nothing to do behind the count)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w1 = smem;                         // packed [k_rbf/8][2 hh][64 cols][4]
  float* s_w2 = s_w1 + k_rbf * EF_NS;         // packed [8][2][64][4]
  float* s_b2 = s_w2 + EF_NS * EF_NS;         // [64]
  float* s_off = s_b2 + EF_NS;                // [k_rbf]
  float* s_h = s_off + 64;                    // [4 waves][32][EF_HS]
  const int tid = threadIdx.x;
  for (int i = tid; i < k_rbf * EF_NS; i += 256) {
    const int k = i / EF_NS, c = i - k * EF_NS;
    s_w1[(((k >> 3) * 2 + ((k >> 2) & 1)) * EF_NS + c) * 4 + (k & 3)] = w1d[i];
  }
  for (int i = tid; i < EF_NS * EF_NS; i += 256) {
    const int k = i / EF_NS, c = i - k * EF_NS;
    s_w2[(((k >> 3) * 2 + ((k >> 2) & 1)) * EF_NS + c) * 4 + (k & 3)] = w2[i];
  }
  if (tid < EF_NS) s_b2[tid] = b2[tid];
  for (int i = tid; i < k_rbf; i += 256) s_off[i] = offset[i];
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  float* my_h = s_h + wave * 32 * EF_HS;
  const f32x4* w1q = reinterpret_cast<const f32x4*>(s_w1);
  const f32x4* w2q = reinterpret_cast<const f32x4*>(s_w2);
  const int nm1 = k_rbf >> 3;
  for (int base = (blk * 4 + wave) * 32; base < n_edges; base += nblk * 128) {
    const int e = base + r;
    const int ec = min(e, n_edges - 1);
    const int a = ia[ec], b = ib[ec];
    const float vx = pos_b[3 * b] - pos_a[3 * a], vy = pos_b[3 * b + 1] - pos_a[3 * a + 1],
                vz = pos_b[3 * b + 2] - pos_a[3 * a + 2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    if (hh == 0 && e < n_edges) {
      const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);  // sqrt(3) / max(|v|, eps)  (F.normalize eps)
      reinterpret_cast<f32x4*>(sh)[e] = f32x4{1.0f, vx * inv, vy * inv, vz * inv};
    }
    // first layer: accumulators start from the pre rows of the 16 edge rows this lane holds in the C/D layout
    ef_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int er = min(base + row, n_edges - 1);
      const float* prow = pre + (size_t)pre_idx[er] * ld_pre;
      acc0[i] = (r < ns) ? prow[r] : 0.f;
      acc1[i] = (32 + r < ns) ? prow[32 + r] : 0.f;
      if (er < n_pre2) {   // rows of a second table added to the first n_pre2 edges (bond-type columns of lig_edge_embedding)
        const float* p2 = pre2 + (size_t)er * ld_pre2;
        if (r < ns) acc0[i] = acc0[i] + p2[r];
        if (32 + r < ns) acc1[i] = acc1[i] + p2[32 + r];
      }
    }
    for (int m = 0; m < nm1; ++m) {
      const f32x4 b0 = w1q[(m * 2 + hh) * EF_NS + r], b1 = w1q[(m * 2 + hh) * EF_NS + 32 + r];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = d - s_off[8 * m + 4 * hh + i];
        const float rb = expf(coeff * (t * t));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, b0[i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, b1[i], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      my_h[row * EF_HS + r] = fmaxf(acc0[i], 0.f);
      my_h[row * EF_HS + 32 + r] = fmaxf(acc1[i], 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tile is complete before it is read back in the A layout
    {
      const float bb0 = s_b2[r], bb1 = s_b2[32 + r];
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = bb0; acc1[i] = bb1; }
    }
    const float* hrow = my_h + r * EF_HS + 4 * hh;
#pragma unroll
    for (int m = 0; m < EF_NS / 8; ++m) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(hrow + 8 * m);
      const f32x4 b0 = w2q[(m * 2 + hh) * EF_NS + r], b1 = w2q[(m * 2 + hh) * EF_NS + 32 + r];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0[i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1[i], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (base + row < n_edges) {
        if (r < ns) out[(size_t)(base + row) * ns + r] = acc0[i];
        if (32 + r < ns) out[(size_t)(base + row) * ns + 32 + r] = acc1[i];
      }
    }
    __builtin_amdgcn_wave_barrier();   // my_h is rewritten by the next tile
  }
}

extern "C" int ddp_edge_featurize(const float* pos_a, const int32_t* ia, const float* pos_b, const int32_t* ib,
                                  int n_edges, const int32_t* n_edges_dev, const float* offset, int k_rbf, float coeff,
                                  const float* pre, const int32_t* pre_idx, int ld_pre, const float* pre2, int n_pre2, int ld_pre2,
                                  const float* w1d, const float* w2, const float* b2, int ns, float* out, float* sh, void* stream) {
  if (n_edges <= 0) return 0;
  if (n_pre2 > 0 && !pre2) return ddp_fail(DDP_EINVAL, "ddp_edge_featurize: n_pre2 > 0 but pre2 is null");
  if (!pre2) n_pre2 = 0;
  if (!pos_a || !ia || !pos_b || !ib || !offset || !pre || !pre_idx || !w1d || !w2 || !b2 || !out || !sh)
    return ddp_fail(DDP_EINVAL, "ddp_edge_featurize: null argument");
  if (ns < 1 || ns > EF_NS) return ddp_fail(DDP_ELIMIT, "ddp_edge_featurize: ns > 64");
  if (k_rbf < 2 || k_rbf > 256) return ddp_fail(DDP_ELIMIT, "ddp_edge_featurize: k_rbf");
  if ((k_rbf & 7) == 0 && k_rbf <= 64) {   // both Linears on the matrix cores
    ddp_featurize_job_t J = {pos_a, ia, pos_b, ib, n_edges, n_edges_dev, offset, k_rbf, coeff, pre, pre_idx, ld_pre, pre2, n_pre2, ld_pre2,
                             w1d, w2, b2, ns, out, sh};
    return ddp_edge_featurize_jobs(&J, 1, stream);
  }
  const size_t lds = (size_t)(k_rbf * EF_NS + EF_NS * EF_NS + EF_NS + ((k_rbf + 3) & ~3) + 4 * 64 * 65) * sizeof(float);
  static int lds_have = 0;
  hipError_t err = ddp_need_lds(reinterpret_cast<const void*>(ddp_edge_featurize_kernel), (int)lds, &lds_have);
  if (err != hipSuccess) return ddp_fail_hip(err, "hipFuncSetAttribute(edge_featurize)");
  int blocks = (n_edges + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ddp_edge_featurize_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, pos_a, ia, pos_b, ib,
                     n_edges, n_edges_dev, offset, k_rbf, coeff, pre, pre_idx, ld_pre, pre2, n_pre2, ld_pre2, w1d, w2, b2, ns, out, sh);
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_edge_featurize launch");
  return 0;
}

extern "C" int ddp_edge_featurize_jobs(const ddp_featurize_job_t* jobs, int njobs, void* stream) {
  if (njobs < 0 || njobs > DDP_MAX_FEATURIZE_JOBS) return ddp_fail(DDP_ELIMIT, "ddp_edge_featurize_jobs: njobs");
  if (njobs == 0) return 0;
  if (!jobs) return ddp_fail(DDP_EINVAL, "ddp_edge_featurize_jobs: null jobs");
  FeatLaunch L;
  L.njobs = 0;
  int blocks = 0, kmax = 0;
  for (int i = 0; i < njobs; ++i) {
    ddp_featurize_job_t J = jobs[i];
    if (J.n_edges <= 0) continue;
    if (J.n_pre2 > 0 && !J.pre2) return ddp_fail(DDP_EINVAL, "ddp_edge_featurize: n_pre2 > 0 but pre2 is null");
    if (!J.pre2) J.n_pre2 = 0;
    if (!J.pos_a || !J.ia || !J.pos_b || !J.ib || !J.offset || !J.pre || !J.pre_idx || !J.w1d || !J.w2 || !J.b2 || !J.out || !J.sh)
      return ddp_fail(DDP_EINVAL, "ddp_edge_featurize: null argument");
    if (J.ns < 1 || J.ns > EF_NS) return ddp_fail(DDP_ELIMIT, "ddp_edge_featurize: ns > 64");
    if (J.k_rbf < 8 || J.k_rbf > 64 || (J.k_rbf & 7))
      return ddp_fail(DDP_ELIMIT, "ddp_edge_featurize_jobs: k_rbf must be a multiple of 8 in [8, 64] (ddp_edge_featurize takes any)");
    int nb = (J.n_edges + 127) / 128;
    if (nb > 1024) nb = 1024;
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = J;
    blocks += nb;
    if (J.k_rbf > kmax) kmax = J.k_rbf;
  }
  L.blk_start[L.njobs] = blocks;
  if (blocks == 0) return 0;
  const size_t lds_m = (size_t)(kmax * EF_NS + EF_NS * EF_NS + EF_NS + 64 + 4 * 32 * EF_HS) * sizeof(float);
  static int lds_have_m = 0;
  hipError_t e2 = ddp_need_lds(reinterpret_cast<const void*>(ddp_edge_featurize_mfma_kernel), (int)lds_m, &lds_have_m);
  if (e2 != hipSuccess) return ddp_fail_hip(e2, "hipFuncSetAttribute(edge_featurize_mfma)");
  hipLaunchKernelGGL(ddp_edge_featurize_mfma_kernel, dim3(blocks), dim3(256), lds_m, (hipStream_t)stream, L);
  e2 = hipGetLastError();
  if (e2 != hipSuccess) return ddp_fail_hip(e2, "ddp_edge_featurize (mfma) launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ torsion sh
// t = sqrt(3/2) * (3 (n.v) v - n): the 1o block of o3.FullTensorProduct(sh(lmax=1), "2e") applied to
// (sh(edge), Y2(bond)) (reference models/all_atom_score_model.py:394-395,418-419).
__global__ void ddp_torsion_sh_kernel(const float* __restrict__ sh_edge, const float* __restrict__ bond_vec,
                                      const int* __restrict__ bond_of_edge, int n_edges, const int* __restrict__ n_edges_dev,
                                      float* __restrict__ out, int edge_blocks, const float* __restrict__ x, int ldx, int ns,
                                      const int* __restrict__ b0, const int* __restrict__ b1, int n_bonds,
                                      float* __restrict__ bond_attr) {
  if ((int)blockIdx.x >= edge_blocks) {   // bond_attr = x[b0, :ns] + x[b1, :ns]
    const int i = (blockIdx.x - edge_blocks) * blockDim.x + threadIdx.x;
    if (i >= n_bonds * ns) return;
    const int b = i / ns, c = i - b * ns;
    bond_attr[i] = x[(size_t)b0[b] * ldx + c] + x[(size_t)b1[b] * ldx + c];
    return;
  }
  if (n_edges_dev) n_edges = min(n_edges, *n_edges_dev);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const f32x4 s = reinterpret_cast<const f32x4*>(sh_edge)[e];
  const float inv3 = 0.57735026918962576f;
  const float nx = s[1] * inv3, ny = s[2] * inv3, nz = s[3] * inv3;  // unit edge vector
  const int b = bond_of_edge[e];
  float vx = bond_vec[3 * b], vy = bond_vec[3 * b + 1], vz = bond_vec[3 * b + 2];
  const float vin = 1.0f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
  vx *= vin; vy *= vin; vz *= vin;
  const float dot3 = 3.0f * (nx * vx + ny * vy + nz * vz);
  const float k = 1.2247448713915890f;  // sqrt(3/2)
  reinterpret_cast<f32x4*>(out)[e] = f32x4{0.f, k * (dot3 * vx - nx), k * (dot3 * vy - ny), k * (dot3 * vz - nz)};
}

extern "C" int ddp_torsion_sh(const float* sh_edge, const float* bond_vec, const int32_t* bond_of_edge, int n_edges,
                              const int32_t* n_edges_dev, float* out, const float* x, int ldx, int ns, const int32_t* b0,
                              const int32_t* b1, int n_bonds, float* bond_attr, void* stream) {
  if (n_edges < 0) n_edges = 0;
  if (n_edges > 0 && (!sh_edge || !bond_vec || !bond_of_edge || !out)) return ddp_fail(DDP_EINVAL, "ddp_torsion_sh: null argument");
  int attr = 0;
  if (bond_attr && n_bonds > 0) {
    if (!x || !b0 || !b1 || ns < 1 || ldx < ns) return ddp_fail(DDP_EINVAL, "ddp_torsion_sh: bond_attr arguments");
    attr = n_bonds * ns;
  }
  const int eb = (n_edges + 255) / 256, ab = (attr + 255) / 256;
  if (eb + ab == 0) return 0;
  hipLaunchKernelGGL(ddp_torsion_sh_kernel, dim3(eb + ab), dim3(256), 0, (hipStream_t)stream, sh_edge, bond_vec, bond_of_edge,
                     n_edges, n_edges_dev, out, eb, x, ldx, ns, b0, b1, n_bonds, bond_attr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_torsion_sh launch");
  return 0;
}
