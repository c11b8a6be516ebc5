// ddp_node.hip - node encoders and the sigma-dependent part of the edge-embedding MLPs on gfx950 (MI355X).
//
// Replaces (reference file:line):
//   sinusoidal_embedding / get_timestep_embedding            utils/diffusion_utils.py:73-84,104-109
//   AtomEncoder.forward / OldAtomEncoder.forward             models/score_model.py:54-82 / :17-52
//   the node_sigma_emb columns of the first Linear of every *_edge_embedding MLP (models/all_atom_score_model.py:71-81,
//   164-169,187-192,212-217: `Linear(cat([.., node_sigma_emb[edge end], rbf]))` - the part that depends on the node only)
// as ONE launch per forward: every job is a gathered-row Linear
//   out[n, :ncols] = (emb_mode == 2 ? emb_sum(n) : 0) + bias + [ emb_sum(n) | dense0(n) | dense1(n) | sigma_emb(t(n)) ] @ W
// whose A rows are assembled in LDS (embedding-table gathers summed in feature order, float rows copied, sin / cos of
// scale * t * freq evaluated in place) and multiplied on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
// Workgroup = 32 node rows x 256 threads; K is walked in chunks of 128; wave w owns the 32-column tiles w, w + 4 of the output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

typedef float nd_f32x16 __attribute__((ext_vector_type(16)));

#define ND_ROWS 32
#define ND_KC 128
#define ND_LD (ND_KC + 1)   // LDS row stride of the A chunk: lane r reads column k of row r -> bank r + k, conflict free

struct NodeLaunch {
  int njobs;
  int tile_start[DDP_MAX_NODE_JOBS + 1];
  ddp_node_job_t job[DDP_MAX_NODE_JOBS];
};

__device__ __forceinline__ float nd_emb_sum(const ddp_node_job_t& J, const int* catrow, int col) {
  // x_embedding = 0; x_embedding += table_f[x[:, f]] in feature order (models/score_model.py:75-76): 0 + e0 is exact
  float s = 0.f;
  for (int f = 0; f < J.n_cat; ++f) s = s + J.table[(size_t)catrow[f] * J.emb_dim + col];
  return s;
}

__global__ __launch_bounds__(256) void ddp_node_linear_kernel(const NodeLaunch L) {
  __shared__ float a_lds[ND_ROWS * ND_LD];
  __shared__ int cat_lds[ND_ROWS][DDP_MAX_NODE_CAT];
  __shared__ float t_lds[ND_ROWS];
  const int tid = threadIdx.x;
  int j = 0;
  while (j + 1 < L.njobs && (int)blockIdx.x >= L.tile_start[j + 1]) ++j;
  const ddp_node_job_t& J = L.job[j];
  const int row0 = ((int)blockIdx.x - L.tile_start[j]) * ND_ROWS;
  const int nvalid = min(ND_ROWS, J.n_rows - row0);

  // per-row table rows of the categorical features and the (scaled) diffusion time
  for (int i = tid; i < ND_ROWS * J.n_cat; i += 256) {
    const int r = i / J.n_cat, f = i - r * J.n_cat;
    const int row = row0 + min(r, nvalid - 1);
    cat_lds[r][f] = J.feat_off[f] + J.cat[(size_t)row * J.ld_cat + f];
  }
  if (tid < ND_ROWS && J.sd > 0 && J.t) {
    const int row = row0 + min(tid, nvalid - 1);
    t_lds[tid] = J.scale * J.t[(size_t)row * J.t_stride];   // embedding_scale * t (utils/diffusion_utils.py:106)
  }
  __syncthreads();

  const int k_emb = (J.emb_mode == 1) ? J.emb_dim : 0;
  const int k_d0 = k_emb + J.n_dense[0], k_d1 = k_d0 + J.n_dense[1], K = k_d1 + J.sd;
  const int half = J.sd >> 1;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int ntile = (J.ncols + 31) >> 5;
  nd_f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;

  for (int c0 = 0; c0 < K; c0 += ND_KC) {
    const int kc = min(ND_KC, K - c0);
    const int kc2 = (kc + 1) & ~1;
    // A chunk: element (row, c0 + c); a thread's 16 elements are fetched together (independent loads in flight), then stored
    float av[ND_ROWS * ND_KC / 256];
#pragma unroll
    for (int u = 0; u < ND_ROWS * ND_KC / 256; ++u) {
      const int i = tid + 256 * u;
      const int rr = i / kc2, c = i - rr * kc2, k = c0 + c;
      float v = 0.f;
      if (i < ND_ROWS * kc2 && c < kc) {
        const int row = row0 + min(rr, nvalid - 1);
        if (k < k_emb) v = nd_emb_sum(J, cat_lds[rr], k);
        else if (k < k_d0) v = J.dense[0][(size_t)row * J.ld_dense[0] + (k - k_emb)];
        else if (k < k_d1) v = J.dense[1][(size_t)row * J.ld_dense[1] + (k - k_d0)];
        else {
          const int s = k - k_d1;
          if (J.sig_emb) v = J.sig_emb[(size_t)row * J.ld_sig + s];
          else if (s < half) v = sinf(t_lds[rr] * J.freq[s]);             // [sin | cos | 0 pad] (diffusion_utils.py:73-84)
          else if (s < 2 * half) v = cosf(t_lds[rr] * J.freq[s - half]);
          if (J.sig_out && rr < nvalid) J.sig_out[(size_t)row * J.ld_sig_out + s] = v;   // data[node type].node_sigma_emb
        }
      }
      av[u] = v;
    }
#pragma unroll
    for (int u = 0; u < ND_ROWS * ND_KC / 256; ++u) {
      const int i = tid + 256 * u;
      if (i < ND_ROWS * kc2) a_lds[(i / kc2) * ND_LD + (i % kc2)] = av[u];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tile = wave + 4 * q;
      if (tile < ntile) {
        const int col = tile * 32 + r;
        const bool cok = col < J.ncols;
        const float* wcol = J.w + (size_t)c0 * J.ncols + (cok ? col : 0);
        // eight k-steps at a time: their eight weight loads are issued together (one memory round trip per group instead of
        // one per MFMA - a lone workgroup of the 1404-row receptor Linear is a latency chain otherwise)
        for (int kk = 0; kk < kc2; kk += 16) {
          float a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = kk + 2 * u + hh;
            b[u] = (cok && k < kc) ? wcol[(size_t)k * J.ncols] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = kk + 2 * u + hh;
            a[u] = (k < kc2) ? a_lds[r * ND_LD + k] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (kk + 2 * u < kc2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: C/D register i <-> row (i&3) + 8(i>>2) + 4hh, column tile*32 + r
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + 4 * q;
    const int col = tile * 32 + r;
    if (tile < ntile && col < J.ncols) {
      const float b = J.bias ? J.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (rr < nvalid) {
          float v = acc[q][i] + b;                                  // Linear: x W^T + b
          if (J.emb_mode == 2) v = nd_emb_sum(J, cat_lds[rr], col) + v;   // x_embedding += linear(...) (score_model.py:47)
          if (J.add) v = J.add[(size_t)(row0 + rr) * J.ld_add + col] + v;
          J.out[(size_t)(row0 + rr) * J.ld_out + col] = v;
        }
      }
    }
  }
  const int nz = J.zero_to - J.ncols;
  for (int i = tid; i < nvalid * nz; i += 256) {
    const int rr = i / nz, c = i - rr * nz;
    J.out[(size_t)(row0 + rr) * J.ld_out + J.ncols + c] = 0.f;
  }
}

extern "C" int ddp_node_linear(const ddp_node_job_t* jobs, int njobs, void* stream) {
  if (!jobs) return ddp_fail(DDP_EINVAL, "ddp_node_linear: null argument");
  if (njobs < 0 || njobs > DDP_MAX_NODE_JOBS) return ddp_fail(DDP_ELIMIT, "ddp_node_linear: njobs > DDP_MAX_NODE_JOBS");
  NodeLaunch L;
  L.njobs = 0;
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    const ddp_node_job_t& J = jobs[i];
    if (J.n_rows <= 0) continue;
    if (!J.w || !J.out || J.ncols < 1 || J.ncols > 256) return ddp_fail(DDP_EINVAL, "ddp_node_linear: w / out / ncols (1..256)");
    if (J.n_cat < 0 || J.n_cat > DDP_MAX_NODE_CAT || (J.n_cat > 0 && (!J.cat || !J.table || J.emb_dim < 1)))
      return ddp_fail(DDP_EINVAL, "ddp_node_linear: categorical part");
    if (J.emb_mode < 0 || J.emb_mode > 2 || (J.emb_mode != 0 && J.n_cat == 0)) return ddp_fail(DDP_EINVAL, "ddp_node_linear: emb_mode");
    if (J.emb_mode == 2 && J.emb_dim < J.ncols) return ddp_fail(DDP_EINVAL, "ddp_node_linear: pass-through embedding narrower than the output");
    for (int d = 0; d < 2; ++d)
      if (J.n_dense[d] < 0 || (J.n_dense[d] > 0 && !J.dense[d])) return ddp_fail(DDP_EINVAL, "ddp_node_linear: dense part");
    if (J.sd < 0 || (J.sd > 0 && !J.sig_emb && (!J.t || !J.freq))) return ddp_fail(DDP_EINVAL, "ddp_node_linear: sigma part");
    if (J.zero_to > 0 && J.zero_to < J.ncols) return ddp_fail(DDP_EINVAL, "ddp_node_linear: zero_to < ncols");
    L.tile_start[L.njobs] = tiles;
    L.job[L.njobs] = J;
    if (J.zero_to <= 0) L.job[L.njobs].zero_to = J.ncols;
    tiles += (J.n_rows + ND_ROWS - 1) / ND_ROWS;
    ++L.njobs;
  }
  L.tile_start[L.njobs] = tiles;
  if (tiles == 0) return 0;
  hipLaunchKernelGGL(ddp_node_linear_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, L);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_node_linear launch");
  return 0;
}
