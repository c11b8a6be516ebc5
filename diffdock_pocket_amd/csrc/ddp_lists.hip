// ddp_lists.hip - index-list primitives with DEVICE-SIDE counts (gfx950), so that a denoising step needs no host
// synchronisation: the sizes of the pose-dependent edge lists (radius graphs, their CSR / source-order views, the pruned and
// "touched" sub-lists of the exact work eliminations) live in device memory, every consumer is launched on a grid sized for
// the list's CAPACITY and reads the actual count itself.  All primitives are batched: one launch serves up to
// DDP_MAX_LIST_JOBS independent jobs (a step has ~10 views / ~6 searches / ~8 sub-lists; each would otherwise be its own
// chain of small launches).  Everything is bitwise deterministic: no result depends on the order in which atomics land.
//
//   ddp_scan_jobs      exclusive prefix sum of per-item values (flags, counts, CSR row lengths), optional compaction of the
//                      flagged indices; one workgroup of 1024 threads per job
//   ddp_mark_jobs      mask[idx[i]] = 1
//   ddp_rowcopy_jobs   compaction of whole CSR rows (keep mask per row)
//   ddp_select_jobs    stable compaction of the items with maskA[idxA[i]] | maskB[idxB[i]] and of their payloads
//   ddp_gather_rows    out[i, :] = x[idx[i], :]
//   ddp_clean_pair_maps  the two index maps of the layer-1 clean-pair sharing (score_model)
// The reference has no counterpart: it calls torch_cluster / torch.sort / boolean indexing per forward and synchronises with
// the host every time (models/all_atom_score_model.py:444-636); these kernels stand for that host-side list handling.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

template <typename J>
struct ListLaunch {
  int njobs;
  int blk_start[DDP_MAX_LIST_JOBS + 1];
  J job[DDP_MAX_LIST_JOBS];
};

__device__ __forceinline__ int list_job_of(const int* blk_start, int njobs, int b) {
  int j = 0;
  while (j + 1 < njobs && b >= blk_start[j + 1]) ++j;
  return j;
}

__device__ __forceinline__ int dev_count(int cap, const int32_t* n_dev) {
  if (!n_dev) return cap;
  const int n = *n_dev;
  return n < cap ? (n < 0 ? 0 : n) : cap;
}

// ------------------------------------------------------------------------------------------------ scan
__device__ __forceinline__ int scan_value(const ddp_scan_job_t& J, int i) {
  int v = 1;
  if (J.val) v = J.val[i];
  else if (J.rowptr) v = J.rowptr[i + 1] - J.rowptr[i];
  if (J.flag && J.flag[i] == 0) v = 0;
  return v;
}

// A lane owns EPL = 4 CONSECUTIVE items per chunk and moves them with 16-byte accesses: a wave instruction then covers 1 KiB of
// contiguous memory.  (One workgroup does a whole job, so everything goes through ONE CU's memory pipe: with 8 items per lane
// fetched as 8 separate dwords - every instruction touching 16 cache lines for 256 useful bytes - a 44 K-item scan took
// 46 - 74 us; the arithmetic is nothing.)  `vec`: every array of the job is 16-byte aligned (the usual case).
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define DDP_SCAN_EPL 4
#define DDP_SCAN_NCH 12   // chunks of 64 * EPL items a wave keeps in registers (small form: n <= 16 * NCH * 256 = 48 K)

__device__ __forceinline__ void scan_load4(const ddp_scan_job_t& J, int i, int hi, bool vec, int v[4]) {
  if (i + 4 <= hi && vec) {
    if (J.val) {
      const i32x4 a = *reinterpret_cast<const i32x4*>(J.val + i);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    } else if (J.rowptr) {
      const i32x4 a = *reinterpret_cast<const i32x4*>(J.rowptr + i);
      const int nx = J.rowptr[i + 4];
      v[0] = a[1] - a[0]; v[1] = a[2] - a[1]; v[2] = a[3] - a[2]; v[3] = nx - a[3];
    } else {
      v[0] = v[1] = v[2] = v[3] = 1;
    }
    if (J.flag) {
      const i32x4 f = *reinterpret_cast<const i32x4*>(J.flag + i);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (f[u] == 0) v[u] = 0;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u < hi) ? scan_value(J, i + u) : 0;
  }
}

// writes the exclusive prefixes ex, ex + v0, ... of the lane's four items (and the compacted indices)
__device__ __forceinline__ void scan_store4(const ddp_scan_job_t& J, int i, int hi, bool vec, const int v[4], int ex) {
  const int e[4] = {ex, ex + v[0], ex + v[0] + v[1], ex + v[0] + v[1] + v[2]};
  if (i + 4 <= hi && vec) {
    const i32x4 o = {e[0], e[1], e[2], e[3]};
    if (J.excl) *reinterpret_cast<i32x4*>(J.excl + i) = o;
    if (J.excl2) *reinterpret_cast<i32x4*>(J.excl2 + i) = o;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u < hi) {
        if (J.excl) J.excl[i + u] = e[u];
        if (J.excl2) J.excl2[i + u] = e[u];
      }
  }
  if (J.list) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u < hi && v[u] != 0) J.list[e[u]] = i + u;   // (weights are 0 / 1 when a list is asked for)
  }
}

__device__ __forceinline__ bool scan_aligned(const ddp_scan_job_t& J) {
  return ((reinterpret_cast<size_t>(J.val) | reinterpret_cast<size_t>(J.rowptr) | reinterpret_cast<size_t>(J.flag) |
           reinterpret_cast<size_t>(J.excl) | reinterpret_cast<size_t>(J.excl2)) & 15) == 0;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(v, off);
    if (lane >= off) v += up;
  }
  return v;
}

// General form, any n: one workgroup of 1024 threads per job, a contiguous segment per wave, two passes over it.
__global__ __launch_bounds__(1024) void ddp_scan_jobs_kernel(const ListLaunch<ddp_scan_job_t> L) {
  __shared__ int seg_total[16];
  constexpr int EPL = DDP_SCAN_EPL, CH = 64 * EPL;
  const ddp_scan_job_t& J = L.job[blockIdx.x];
  const int n = dev_count(J.n, J.n_dev);
  const bool vec = scan_aligned(J);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int S = (((n + 15) / 16) + CH - 1) / CH * CH;      // contiguous segment per wave, a multiple of the chunk
  const int lo = min(wave * S, n), hi = min(lo + S, n);
  int run = 0;
  for (int base = lo; base < hi; base += CH) {
    int v[4];
    scan_load4(J, base + lane * EPL, hi, vec, v);
    int t = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
    run += t;
  }
  if (lane == 0) seg_total[wave] = run;
  __syncthreads();
  int offset = J.base;
  for (int w = 0; w < wave; ++w) offset += seg_total[w];
  run = offset;
  for (int base = lo; base < hi; base += CH) {
    int v[4];
    scan_load4(J, base + lane * EPL, hi, vec, v);
    const int mine = v[0] + v[1] + v[2] + v[3];
    const int incl = wave_incl_scan(mine, lane);
    scan_store4(J, base + lane * EPL, hi, vec, v, run + incl - mine);
    run += __shfl(incl, 63);
  }
  if (threadIdx.x == 0) {
    int total = J.base;
    for (int w = 0; w < 16; ++w) total += seg_total[w];
    if (J.excl) J.excl[n] = total;
    if (J.total) *J.total = total;
  }
}

// Small form (host-chosen by the jobs' capacities: every job <= 48 K items, i.e. every per-node scan of a 40-sample batch): a
// wave's whole segment is fetched at once - NCH independent 16-byte loads per lane and array, ONE memory round trip - and stays
// in registers for the second pass.
__global__ __launch_bounds__(1024) void ddp_scan_jobs_small_kernel(const ListLaunch<ddp_scan_job_t> L) {
  __shared__ int seg_total[16];
  constexpr int EPL = DDP_SCAN_EPL, CH = 64 * EPL, NCH = DDP_SCAN_NCH;
  const ddp_scan_job_t& J = L.job[blockIdx.x];
  const int n = dev_count(J.n, J.n_dev);
  const bool vec = scan_aligned(J);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int S = (((n + 15) / 16) + CH - 1) / CH * CH;
  const int lo = min(wave * S, n), hi = min(lo + S, n);
  const int nch = S / CH;
  int val[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c < nch) {
      scan_load4(J, lo + c * CH + lane * EPL, hi, vec, val[c]);
    } else {
      val[c][0] = val[c][1] = val[c][2] = val[c][3] = 0;
    }
  }
  int run = 0;
  int excl[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int mine = val[c][0] + val[c][1] + val[c][2] + val[c][3];
    const int incl = wave_incl_scan(mine, lane);
    excl[c] = run + incl - mine;           // exclusive prefix of the lane's first item inside the wave's segment
    run += __shfl(incl, 63);
  }
  if (lane == 0) seg_total[wave] = run;
  __syncthreads();
  int offset = J.base;
  for (int w = 0; w < wave; ++w) offset += seg_total[w];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (c < nch) scan_store4(J, lo + c * CH + lane * EPL, hi, vec, val[c], offset + excl[c]);
  if (threadIdx.x == 0) {
    int total = J.base;
    for (int w = 0; w < 16; ++w) total += seg_total[w];
    if (J.excl) J.excl[n] = total;
    if (J.total) *J.total = total;
  }
}

static int check_jobs(int njobs, const void* jobs, const char* what) {
  if (njobs < 0 || njobs > DDP_MAX_LIST_JOBS) return ddp_fail(DDP_ELIMIT, what);
  if (njobs > 0 && !jobs) return ddp_fail(DDP_EINVAL, what);
  return 0;
}

static int launch_ok(const char* what) {
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : ddp_fail_hip(err, what);
}

extern "C" int ddp_scan_jobs(const ddp_scan_job_t* jobs, int njobs, void* stream) {
  if (int rc = check_jobs(njobs, jobs, "ddp_scan_jobs: njobs")) return rc;
  if (njobs == 0) return 0;
  ListLaunch<ddp_scan_job_t> L;
  L.njobs = njobs;
  for (int i = 0; i < njobs; ++i) {
    if (jobs[i].n < 0 || (jobs[i].list && (jobs[i].val || jobs[i].rowptr)))
      return ddp_fail(DDP_EINVAL, "ddp_scan_jobs: n < 0, or a list asked for with weights other than 0 / 1");
    L.job[i] = jobs[i];
  }
  bool small = true;
  for (int i = 0; i < njobs; ++i) small = small && jobs[i].n <= 16 * DDP_SCAN_NCH * 64 * DDP_SCAN_EPL;
  if (small)
    hipLaunchKernelGGL(ddp_scan_jobs_small_kernel, dim3(njobs), dim3(1024), 0, (hipStream_t)stream, L);
  else
    hipLaunchKernelGGL(ddp_scan_jobs_kernel, dim3(njobs), dim3(1024), 0, (hipStream_t)stream, L);
  return launch_ok("ddp_scan_jobs launch");
}

// ------------------------------------------------------------------------------------------------ mark
__global__ __launch_bounds__(256) void ddp_mark_jobs_kernel(const ListLaunch<ddp_mark_job_t> L) {
  const int j = list_job_of(L.blk_start, L.njobs, blockIdx.x);
  const ddp_mark_job_t& J = L.job[j];
  const int n = dev_count(J.n, J.n_dev);
  const int i = ((int)blockIdx.x - L.blk_start[j]) * 256 + (int)threadIdx.x;
  if (i < n) J.mask[J.idx ? J.idx[i] : i] = 1;      // (every writer stores the same value)
}

extern "C" int ddp_mark_jobs(const ddp_mark_job_t* jobs, int njobs, void* stream) {
  if (int rc = check_jobs(njobs, jobs, "ddp_mark_jobs: njobs")) return rc;
  ListLaunch<ddp_mark_job_t> L;
  L.njobs = 0;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    if (jobs[i].n <= 0) continue;
    if (!jobs[i].mask) return ddp_fail(DDP_EINVAL, "ddp_mark_jobs: null mask");
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = jobs[i];
    blocks += (jobs[i].n + 255) / 256;
  }
  L.blk_start[L.njobs] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(ddp_mark_jobs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  return launch_ok("ddp_mark_jobs launch");
}

// ------------------------------------------------------------------------------------------------ row copy
// one wave per row of the old CSR list: a kept row's entries move, in order, to their place in the new list
__global__ __launch_bounds__(256) void ddp_rowcopy_jobs_kernel(const ListLaunch<ddp_rowcopy_job_t> L) {
  const int j = list_job_of(L.blk_start, L.njobs, blockIdx.x);
  const ddp_rowcopy_job_t& J = L.job[j];
  const int row = ((int)blockIdx.x - L.blk_start[j]) * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (row >= J.n_rows || J.keep[row] == 0) return;
  const int ob = J.old_rowptr[row], m = J.old_rowptr[row + 1] - ob, nb = J.new_rowptr[row];
  for (int a = lane; a < m; a += 64) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (J.out[k]) J.out[k][nb + a] = J.in[k][ob + a];
  }
}

extern "C" int ddp_rowcopy_jobs(const ddp_rowcopy_job_t* jobs, int njobs, void* stream) {
  if (int rc = check_jobs(njobs, jobs, "ddp_rowcopy_jobs: njobs")) return rc;
  ListLaunch<ddp_rowcopy_job_t> L;
  L.njobs = 0;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    const ddp_rowcopy_job_t& J = jobs[i];
    if (J.n_rows <= 0) continue;
    if (!J.keep || !J.old_rowptr || !J.new_rowptr) return ddp_fail(DDP_EINVAL, "ddp_rowcopy_jobs: null argument");
    for (int k = 0; k < 3; ++k)
      if (J.out[k] && !J.in[k]) return ddp_fail(DDP_EINVAL, "ddp_rowcopy_jobs: output without input");
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = J;
    blocks += (J.n_rows + 3) / 4;
  }
  L.blk_start[L.njobs] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(ddp_rowcopy_jobs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  return launch_ok("ddp_rowcopy_jobs launch");
}

// ------------------------------------------------------------------------------------------------ select
// Stable compaction in three launches: per-block counts (blocks of DDP_SELECT_BLOCK consecutive items), an exclusive scan of
// the block counts (ddp_scan_jobs_kernel), and the scatter, in which every block recomputes its flags, ranks them with wave
// ballots in item order and writes the kept indices and payloads behind its block offset.
#define DDP_SELECT_BLOCK 2048   // items per workgroup of 256 threads: wave w owns items [512 w, 512 w + 512) in 8 rounds of 64

__device__ __forceinline__ bool select_flag(const ddp_select_job_t& J, int i) {
  bool f = false;
  if (J.mask_a) f = J.mask_a[J.idx_a ? J.idx_a[i] : i] != 0;
  if (J.mask_b) f = f || (J.mask_b[J.idx_b ? J.idx_b[i] : i] != 0);
  return f;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void ddp_select_jobs_kernel(const ListLaunch<ddp_select_job_t> L) {
  __shared__ int wave_total[4];
  const int j = list_job_of(L.blk_start, L.njobs, blockIdx.x);
  const ddp_select_job_t& J = L.job[j];
  const int n = dev_count(J.n, J.n_dev);
  const int blk = (int)blockIdx.x - L.blk_start[j];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int w0 = blk * DDP_SELECT_BLOCK + wave * 512;
  if (blk * DDP_SELECT_BLOCK >= n) {                 // beyond the actual count: an empty block
    if (!SCATTER && threadIdx.x == 0) J.block_count[blk] = 0;
    return;
  }
  unsigned long long mm[8];
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = w0 + 64 * r + lane;
    mm[r] = __ballot(i < n && select_flag(J, i));
    cnt += __popcll(mm[r]);
  }
  if (lane == 0) wave_total[wave] = cnt;
  __syncthreads();
  if (!SCATTER) {
    if (threadIdx.x == 0) J.block_count[blk] = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    return;
  }
  int o = J.block_off[blk];
  for (int w = 0; w < wave; ++w) o += wave_total[w];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = w0 + 64 * r + lane;
    if ((mm[r] >> lane) & 1ull) {
      const int p = o + __popcll(mm[r] & ((1ull << lane) - 1ull));
      if (J.out_idx) J.out_idx[p] = i;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (J.out[k]) J.out[k][p] = J.pay[k][i] + J.pay_add[k];
    }
    o += __popcll(mm[r]);
  }
}

extern "C" int ddp_select_jobs(const ddp_select_job_t* jobs, int njobs, void* stream) {
  if (int rc = check_jobs(njobs, jobs, "ddp_select_jobs: njobs")) return rc;
  ListLaunch<ddp_select_job_t> L;
  ListLaunch<ddp_scan_job_t> S;
  L.njobs = 0;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    const ddp_select_job_t& J = jobs[i];
    if (J.n <= 0) {                                  // an empty list: only its count is produced
      if (J.total) {
        const hipError_t e = hipMemsetAsync(J.total, 0, sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) return ddp_fail_hip(e, "ddp_select_jobs memset");
      }
      continue;
    }
    if (!J.block_count || !J.block_off || (!J.mask_a && !J.mask_b)) return ddp_fail(DDP_EINVAL, "ddp_select_jobs: null argument");
    for (int k = 0; k < 4; ++k)
      if (J.out[k] && !J.pay[k]) return ddp_fail(DDP_EINVAL, "ddp_select_jobs: output without payload");
    const int nb = (J.n + DDP_SELECT_BLOCK - 1) / DDP_SELECT_BLOCK;
    ddp_scan_job_t& sj = S.job[L.njobs];
    sj = ddp_scan_job_t{};
    sj.n = nb;
    sj.val = J.block_count;
    sj.excl = J.block_off;
    sj.total = J.total;
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = J;
    blocks += nb;
  }
  L.blk_start[L.njobs] = blocks;
  S.njobs = L.njobs;
  if (blocks == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ddp_select_jobs_kernel<false>, dim3(blocks), dim3(256), 0, st, L);
  hipLaunchKernelGGL(ddp_scan_jobs_kernel, dim3(S.njobs), dim3(1024), 0, st, S);
  hipLaunchKernelGGL(ddp_select_jobs_kernel<true>, dim3(blocks), dim3(256), 0, st, L);
  return launch_ok("ddp_select_jobs launch");
}

// ------------------------------------------------------------------------------------------------ gather rows
__global__ __launch_bounds__(256) void ddp_gather_rows_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ idx,
                                                              int n, const int32_t* __restrict__ n_dev, float* __restrict__ out,
                                                              int ldo, int ncols) {
  n = dev_count(n, n_dev);
  const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (row >= n) return;
  const float* __restrict__ src = x + (size_t)idx[row] * ldx;
  float* __restrict__ dst = out + (size_t)row * ldo;
  for (int c = lane; c < ncols; c += 64) dst[c] = src[c];
}

extern "C" int ddp_gather_rows(const float* x, int ldx, const int32_t* idx, int n, const int32_t* n_dev, float* out, int ldo,
                               int ncols, void* stream) {
  if (n <= 0 || ncols <= 0) return 0;
  if (!x || !idx || !out) return ddp_fail(DDP_EINVAL, "ddp_gather_rows: null argument");
  hipLaunchKernelGGL(ddp_gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, n, n_dev, out,
                     ldo, ncols);
  return launch_ok("ddp_gather_rows launch");
}

// ------------------------------------------------------------------------------------------------ clean-pair maps
// score_model, layer 1 atom<-atom of a sampling batch of one rigid complex (B samples x n0 atoms, E = B * e0 edges in CSR
// order): `touched[a]` marks the atoms a ligand message reached.
//   rowmap[p]  = p                      if the edge at CSR position p has a touched end (its message is computed per sample)
//              = E + p % e0             otherwise (the message computed once on the complex's own edge list)
//   rows_v[a0] = g * n0 + a0 for the first sample g in which atom a0 is untouched (g = 0 if there is none: then unused)
__global__ __launch_bounds__(256) void ddp_clean_pair_maps_kernel(const int32_t* __restrict__ touched, const int32_t* __restrict__ recv,
                                                                  const int32_t* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                                  int E, int e0, int B, int n0,
                                                                  int32_t* __restrict__ rowmap, int32_t* __restrict__ rows_v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < E) {
    // the same edge in the complex's own list: position i % e0 when every sample has the same list (rigid receptor); in general
    // the receiver's row in sample 0 at the same offset (an untouched receiver's row is, entry by entry, its copy's)
    const int r = recv[i];
    const int ref = rowptr ? rowptr[r % n0] + (i - rowptr[r]) : i % e0;
    rowmap[i] = (touched[r] | touched[src[i]]) ? i : E + ref;
  }
  if (i < n0) {
    int g = 0;
    for (int b = 0; b < B; ++b)
      if (touched[b * n0 + i] == 0) {
        g = b;
        break;
      }
    rows_v[i] = g * n0 + i;
  }
}

extern "C" int ddp_clean_pair_maps(const int32_t* touched, const int32_t* recv, const int32_t* src, const int32_t* rowptr, int n_edges,
                                   int e0, int n_graphs, int n0, int32_t* rowmap, int32_t* rows_v, void* stream) {
  if (n_edges <= 0 || e0 <= 0 || n_graphs <= 0 || n0 <= 0 || n_edges != e0 * n_graphs)
    return ddp_fail(DDP_EINVAL, "ddp_clean_pair_maps: sizes");
  if (!touched || !recv || !src || !rowmap || !rows_v) return ddp_fail(DDP_EINVAL, "ddp_clean_pair_maps: null argument");
  const int n = n_edges > n0 ? n_edges : n0;
  hipLaunchKernelGGL(ddp_clean_pair_maps_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, touched, recv, src,
                     rowptr, n_edges, e0, n_graphs, n0, rowmap, rows_v);
  return launch_ok("ddp_clean_pair_maps launch");
}

// ------------------------------------------------------------------------------------------------ partial sharing with moving atoms
// A batch of B poses of ONE complex whose side chains move (flexible residues): node (s, i) = copy i of sample s, n nodes and e0
// edges per sample, every sample's edge list stored sample after sample.  An atom is "off" in sample s when its position there
// differs bitwise from its position in sample 0 (flag == NULL), or when flag[(s, i)] != 0 (a mask made by an earlier pass).
// For an edge list (a[e], b[e]) of per-sample node ids this marks
//   mark[a[e]] = 1  if b[e] is off - or, with a_too, a[e] itself is off                                  (edges of sample s)
//   mark[(s, a0)] = 1  for every edge (a0, b0) of SAMPLE 0's list whose b0 is off in sample s             (ref_list)
//   mark[(0, i)] = 1  for every node of sample 0                                                          (the reference copy)
// engine._lists runs it twice over the atom kNN graph (first a = the query atom of an edge, b = the neighbour it found; then
// a = the receiver, b = the query, off = the first pass's marks) and once over receptor<-atom; the proof that an unmarked
// receiver has, edge by edge, bitwise the inputs of its copy in sample 0 is there.
__global__ __launch_bounds__(256) void ddp_flex_mark_kernel(const float* __restrict__ pos, const int32_t* __restrict__ flag,
                                                            int n_b_per_graph, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                            int n_edges, int e0, int n_a_per_graph, int a_too, int ref_list,
                                                            int32_t* __restrict__ mark) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n_a_per_graph) mark[e] = 1;
  if (e >= n_edges) return;
  const int s = e / e0;
  if (s == 0) return;
  auto off = [&](int x) {   // node x = (s, j) of the b kind (atoms)
    if (flag) return flag[x] != 0;
    const int x0 = x - s * n_b_per_graph;
    return pos[3 * (size_t)x] != pos[3 * (size_t)x0] || pos[3 * (size_t)x + 1] != pos[3 * (size_t)x0 + 1] ||
           pos[3 * (size_t)x + 2] != pos[3 * (size_t)x0 + 2];
  };
  const int ai = a[e], bi = b[e];
  bool m = off(bi);
  if (a_too) m = m || off(ai);
  if (m) mark[ai] = 1;
  if (ref_list) {   // sample 0's edge number e - s e0, seen with sample s's flags / positions
    const int l = e - s * e0;
    if (off(b[l] + s * n_b_per_graph)) mark[a[l] + s * n_a_per_graph] = 1;
  }
}

extern "C" int ddp_flex_mark(const float* pos, const int32_t* flag, int n_b_per_graph, const int32_t* a, const int32_t* b, int n_edges,
                             int e0, int n_a_per_graph, int a_too, int ref_list, int32_t* mark, void* stream) {
  if (n_edges <= 0 || e0 <= 0 || n_edges % e0 || n_b_per_graph <= 0 || n_a_per_graph <= 0)
    return ddp_fail(DDP_EINVAL, "ddp_flex_mark: sizes");
  if ((!pos && !flag) || !a || !b || !mark) return ddp_fail(DDP_EINVAL, "ddp_flex_mark: null argument");
  if (a_too && n_a_per_graph != n_b_per_graph) return ddp_fail(DDP_EINVAL, "ddp_flex_mark: a_too needs both ends of one node kind");
  const int n = n_edges > n_a_per_graph ? n_edges : n_a_per_graph;
  hipLaunchKernelGGL(ddp_flex_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, flag, n_b_per_graph, a, b,
                     n_edges, e0, n_a_per_graph, a_too, ref_list, mark);
  return launch_ok("ddp_flex_mark launch");
}

// Message row of every position of the FULL receiver-CSR list when only the rows of the marked receivers were kept
// (ddp_rowcopy_jobs: new_rowptr) and an unmarked receiver (s, i) reads the messages of its copy (0, i):
//   rowmap[p] = new_rowptr[t] + (p - old_rowptr[r]),   r = recv[p],   t = mark[r] ? r : r % n_recv_per_graph
__global__ __launch_bounds__(256) void ddp_fallback_rowmap_kernel(const int32_t* __restrict__ mark, const int32_t* __restrict__ recv,
                                                                  const int32_t* __restrict__ old_rowptr,
                                                                  const int32_t* __restrict__ new_rowptr, int n_edges,
                                                                  int n_recv_per_graph, int32_t* __restrict__ rowmap) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_edges) return;
  const int r = recv[p];
  const int t = mark[r] ? r : r % n_recv_per_graph;
  rowmap[p] = new_rowptr[t] + (p - old_rowptr[r]);
}

extern "C" int ddp_fallback_rowmap(const int32_t* mark, const int32_t* recv, const int32_t* old_rowptr, const int32_t* new_rowptr,
                                   int n_edges, int n_recv_per_graph, int32_t* rowmap, void* stream) {
  if (n_edges <= 0) return 0;
  if (n_recv_per_graph <= 0) return ddp_fail(DDP_EINVAL, "ddp_fallback_rowmap: sizes");
  if (!mark || !recv || !old_rowptr || !new_rowptr || !rowmap) return ddp_fail(DDP_EINVAL, "ddp_fallback_rowmap: null argument");
  hipLaunchKernelGGL(ddp_fallback_rowmap_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, (hipStream_t)stream, mark, recv, old_rowptr,
                     new_rowptr, n_edges, n_recv_per_graph, rowmap);
  return launch_ok("ddp_fallback_rowmap launch");
}
