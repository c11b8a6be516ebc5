// ddp_heads.hip - the per-graph / per-bond scalar work around the convs of a forward (include/ddp_hip.h):
//   ddp_step_prologue   t -> sigma (utils/diffusion_utils.py:22-34), the dynamic cross cutoff (all_atom_score_model.py:548-550),
//                       the graphs' sigma embedding (:371) and ligand centres (:571-576), bond centres and bond vectors of the
//                       two torsion heads (:589-592,613-616,392,416), the bond rows of the ligand edge list (:462-468)
//   ddp_trrot_head      tr / rot read-out of the final conv (:357-384)
//   ddp_tor_head        tor_final_layer / sc_tor_final_layer + the torus score norm (:400-410,424-434)
// Each replaces 15 - 30 PyTorch launches of a few hundred bytes; in a replayed hipGraph a dependent launch costs ~5 us whatever
// it does, so for small batches (cfg1, the 5-sample shard) these launches WERE the step.  All sums run in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

#pragma clang fp contract(off)

#define DDP_HEAD_THREADS 64
#define DDP_HEAD_MAX_SD 64   // widest sigma embedding the read-out kernel stages in LDS

// wave-wide sum in a fixed order (all lanes get it)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

struct PrologueLaunch {
  ddp_prologue_args_t a;
  int blk_graph, blk_bond[2], blk_copy[2];   // first block of every section behind the graph section
};

__global__ __launch_bounds__(DDP_HEAD_THREADS) void ddp_step_prologue_kernel(PrologueLaunch L) {
  const ddp_prologue_args_t& A = L.a;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b < L.blk_bond[0]) {   // ---- one wave per graph (its loops are chains of dependent loads for a single thread)
    const int g = b, lane = tid;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // component k on lane k
      if (lane == k && A.sigma[k]) {
        float s = 0.f;
        if (A.t[k] && A.sig_max[k] > 0.f) {   // (sig_max <= 0: sigma[k] is an INPUT - the caller's own t_to_sigma made it)
          const float t = A.t[k][(size_t)g * A.t_stride[k]];
          s = powf(A.sig_min[k], 1.0f - t) * powf(A.sig_max[k], t);   // sigma_min^(1-t) * sigma_max^t
          A.sigma[k][g] = s;
        } else {
          s = A.sigma[k][g];
        }
        if (k == 0 && A.cut) A.cut[g] = s * A.cut_mul + A.cut_add;
      }
    }
    if (A.graph_emb) {   // [sin(scale t w) | cos(scale t w) | 0]  (utils/diffusion_utils.py:73-84), t = the tr time
      const float st = A.emb_scale * A.t[0][(size_t)g * A.t_stride[0]];
      const int half = A.sd / 2;
      float* o = A.graph_emb + (size_t)g * A.sd;
      for (int s = lane; s < half; s += DDP_HEAD_THREADS) {
        const float arg = st * A.freq[s];
        o[s] = sinf(arg);
        o[half + s] = cosf(arg);
      }
      if ((A.sd & 1) && lane == 0) o[A.sd - 1] = 0.f;
    }
    if (A.center) {      // mean position of the graph's ligand atoms: lane l sums atoms l, l + 64, ... in order, then a fixed tree
      const int p0 = A.graph_ptr[g], p1 = A.graph_ptr[g + 1];
      float x = 0.f, y = 0.f, z = 0.f;
      for (int p = p0 + lane; p < p1; p += DDP_HEAD_THREADS) {
        x += A.lig_pos[3 * p];
        y += A.lig_pos[3 * p + 1];
        z += A.lig_pos[3 * p + 2];
      }
      x = wave_sum(x); y = wave_sum(y); z = wave_sum(z);
      if (lane == 0) {
        const float n = (float)(p1 - p0);
        A.center[3 * g] = __fdiv_rn(x, n);
        A.center[3 * g + 1] = __fdiv_rn(y, n);
        A.center[3 * g + 2] = __fdiv_rn(z, n);
      }
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // ---- one thread per bond: centre and direction
    const int b_end = h == 0 ? L.blk_bond[1] : L.blk_copy[0];
    if (b >= L.blk_bond[h] && b < b_end) {
      const int i = (b - L.blk_bond[h]) * DDP_HEAD_THREADS + tid;
      if (i >= A.bonds[h].n) return;
      const float* p = A.bonds[h].pos;
      const int i0 = A.bonds[h].b0[i], i1 = A.bonds[h].b1[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float u = p[3 * (size_t)i0 + c], v = p[3 * (size_t)i1 + c];
        if (A.bonds[h].mid) A.bonds[h].mid[3 * (size_t)i + c] = (u + v) / 2.0f;
        if (A.bonds[h].vec) A.bonds[h].vec[3 * (size_t)i + c] = v - u;
      }
      return;
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // ---- int32 copies
    const int b_end = h == 0 ? L.blk_copy[1] : (int)gridDim.x;
    if (b >= L.blk_copy[h] && b < b_end) {
      const int i = (b - L.blk_copy[h]) * DDP_HEAD_THREADS + tid;
      if (i < A.copy[h].n) A.copy[h].dst[i] = A.copy[h].src[i];
      return;
    }
  }
}

extern "C" int ddp_step_prologue(const ddp_prologue_args_t* args, void* stream) {
  if (!args) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: null argument");
  const ddp_prologue_args_t& A = *args;
  if (A.n_graphs < 0) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: n_graphs");
  if (A.cut && !A.sigma[0]) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: cut needs the tr sigma");
  if (A.graph_emb && (!A.t[0] || !A.freq || A.sd < 2)) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: graph_emb");
  if (A.center && (!A.lig_pos || !A.graph_ptr)) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: center");
  PrologueLaunch L;
  L.a = A;
  auto blocks = [](int n) { return n > 0 ? (n + DDP_HEAD_THREADS - 1) / DDP_HEAD_THREADS : 0; };
  int nb = A.n_graphs;      // one wave per graph
  for (int h = 0; h < 2; ++h) {
    if (A.bonds[h].n > 0 && (!A.bonds[h].pos || !A.bonds[h].b0 || !A.bonds[h].b1))
      return ddp_fail(DDP_EINVAL, "ddp_step_prologue: bond job");
    L.blk_bond[h] = nb;
    nb += blocks(A.bonds[h].n);
  }
  for (int h = 0; h < 2; ++h) {
    if (A.copy[h].n > 0 && (!A.copy[h].src || !A.copy[h].dst)) return ddp_fail(DDP_EINVAL, "ddp_step_prologue: copy job");
    L.blk_copy[h] = nb;
    nb += blocks(A.copy[h].n);
  }
  L.blk_graph = 0;
  if (nb == 0) return 0;
  hipLaunchKernelGGL(ddp_step_prologue_kernel, dim3(nb), dim3(DDP_HEAD_THREADS), 0, (hipStream_t)stream, L);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_step_prologue launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ tr / rot read-out
// One wave per graph; lane j = hidden unit j of both read-out MLPs Linear(1 + sd, ns) -> ReLU -> Linear(ns, 1).
__global__ __launch_bounds__(DDP_HEAD_THREADS) void ddp_trrot_head_kernel(ddp_trrot_args_t A) {
  const int g = blockIdx.x, j = threadIdx.x;
  // the first Linear's weights go through LDS: coalesced, independent loads (read straight from memory, lane j walks row j: one
  // dependent 4-byte load per product, 33 memory round trips)
  __shared__ float wl[2][DDP_HEAD_THREADS * (DDP_HEAD_MAX_SD + 1)];
  const int k_in = 1 + A.sd;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    for (int i = j; i < A.ns * k_in; i += DDP_HEAD_THREADS) wl[h][i] = A.w1[h][i];
  const float* gp = A.gp + (size_t)g * A.ld_gp;
  // (:362-363) the 1o and 1e halves of the final conv's output are added
  float v[2][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    v[0][c] = gp[c] + gp[6 + c];
    v[1][c] = gp[3 + c] + gp[9 + c];
  }
  const float* emb = A.graph_emb + (size_t)g * A.sd;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float norm = sqrtf(v[h][0] * v[h][0] + v[h][1] * v[h][1] + v[h][2] * v[h][2]);
    float hid = 0.f;
    if (j < A.ns) {
      const float* w = wl[h] + j * k_in;
      float acc = w[0] * norm;
      for (int k = 0; k < A.sd; ++k) acc += w[1 + k] * emb[k];
      acc += A.b1[h][j];
      hid = fmaxf(acc, 0.f) * A.w2[h][j];
    }
    const float mlp = wave_sum(hid) + A.b2[h][0];
    float scale = 1.0f;
    if (A.sigma[h]) {
      const float s = A.sigma[h][g];
      if (h == 1) {   // so3.score_norm (utils/so3.py:85-89): table over log10(sigma), fp32 index arithmetic
        float idx = __fdiv_rn(log10f(s) - A.so3_lo, A.so3_span) * (float)A.so3_n;
        int i = (int)rintf(idx);
        i = min(max(i, 0), A.so3_n - 1);
        scale = A.so3_table[i];
      }
    }
    if (j < 3) {
      const float vj = j == 0 ? v[h][0] : (j == 1 ? v[h][1] : v[h][2]);
      float o = __fdiv_rn(vj, norm) * mlp;
      if (A.sigma[h]) o = (h == 0) ? __fdiv_rn(o, A.sigma[h][g]) : o * scale;
      A.out[h][3 * (size_t)g + j] = o;
    }
  }
}

extern "C" int ddp_trrot_head(const ddp_trrot_args_t* args, void* stream) {
  if (!args) return ddp_fail(DDP_EINVAL, "ddp_trrot_head: null argument");
  const ddp_trrot_args_t& A = *args;
  if (A.n_graphs <= 0) return 0;
  if (!A.gp || !A.graph_emb || A.ld_gp < 12 || A.ns < 1 || A.ns > DDP_HEAD_THREADS || A.sd < 0 || A.sd > DDP_HEAD_MAX_SD)
    return ddp_fail(DDP_EINVAL, "ddp_trrot_head: arguments");
  for (int h = 0; h < 2; ++h)
    if (!A.w1[h] || !A.b1[h] || !A.w2[h] || !A.b2[h] || !A.out[h]) return ddp_fail(DDP_EINVAL, "ddp_trrot_head: null weights");
  if (A.sigma[1] && (!A.so3_table || A.so3_n < 1)) return ddp_fail(DDP_EINVAL, "ddp_trrot_head: so3 table");
  hipLaunchKernelGGL(ddp_trrot_head_kernel, dim3(A.n_graphs), dim3(DDP_HEAD_THREADS), 0, (hipStream_t)stream, A);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_trrot_head launch");
  return 0;
}

// ------------------------------------------------------------------------------------------------ torsion read-out
// One wave per rotatable bond; lane j = hidden unit j of Linear(2 ns, ns, no bias) -> tanh -> Linear(ns, 1, no bias).
__global__ __launch_bounds__(DDP_HEAD_THREADS) void ddp_tor_head_kernel(ddp_tor_args_t A) {
  const int b = blockIdx.x, j = threadIdx.x;
  __shared__ float h_in[2 * DDP_HEAD_THREADS];
  const int k_in = 2 * A.ns;
  for (int k = j; k < k_in; k += DDP_HEAD_THREADS) h_in[k] = A.h[(size_t)b * A.ld_h + k];
  __syncthreads();
  float hid = 0.f;
  if (j < A.ns) {
    const float* w = A.w1 + (size_t)j * k_in;
    float acc = 0.f;
    for (int k = 0; k < k_in; ++k) acc += w[k] * h_in[k];
    hid = tanhf(acc) * A.w2[j];
  }
  float o = wave_sum(hid);
  if (A.sigma) {   // torus.score_norm (utils/torus.py:78-82): sqrt of the table over ln(sigma / pi)
    const float s = A.sigma[A.graph_of_bond[b]];
    float x = logf(__fdiv_rn(s, 3.14159274101257324f));
    x = __fdiv_rn(x - A.torus_lo, A.torus_span) * (float)A.torus_n;
    x = fminf(fmaxf(x, 0.f), (float)A.torus_n);
    const int i = (int)rintf(x);
    o = o * sqrtf(A.torus_table[i]);
  }
  if (j == 0) A.out[b] = o;
}

extern "C" int ddp_tor_head(const ddp_tor_args_t* args, void* stream) {
  if (!args) return ddp_fail(DDP_EINVAL, "ddp_tor_head: null argument");
  const ddp_tor_args_t& A = *args;
  if (A.n_bonds <= 0) return 0;
  if (!A.h || !A.w1 || !A.w2 || !A.out || A.ns < 1 || A.ns > DDP_HEAD_THREADS || A.ld_h < 2 * A.ns)
    return ddp_fail(DDP_EINVAL, "ddp_tor_head: arguments");
  if (A.sigma && (!A.graph_of_bond || !A.torus_table || A.torus_n < 1)) return ddp_fail(DDP_EINVAL, "ddp_tor_head: score norm");
  hipLaunchKernelGGL(ddp_tor_head_kernel, dim3(A.n_bonds), dim3(DDP_HEAD_THREADS), 0, (hipStream_t)stream, A);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_tor_head launch");
  return 0;
}
