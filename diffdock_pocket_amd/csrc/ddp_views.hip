// ddp_views.hip - CSR / source-order views of an edge list on the device (include/ddp_hip.h: ddp_group_by_key).
//
// The conv kernels consume the edges of a conv direction grouped by RECEIVING node (CSR; the segmented mean of reference
// models/score_model.py:117 sums a node's messages in edge order) and, on the factorised path, grouped by SOURCE node.
// Both are a STABLE sort of the edge list by an integer key < n_keys plus a row-pointer array.  The PyTorch formulation
// (graph.py: torch.sort(stable=True), index_add, cumsum, gathers, dtype conversions) is ~20 small launches per view and
// eight views per forward, all in the host-paced front of the step; this is 5 launches for ALL views of a batch of jobs
// (ddp_group_by_key_jobs; item counts may live in device memory, the grids are sized for the capacities) with identical results:
//   1 zero the row counters            2 histogram of the keys (atomics: a count does not depend on their order)
//   3 exclusive scan -> rowptr         4 every item takes a slot of its row with an atomic (arbitrary order inside a row)
//   5 a wave per row puts the row's items into ascending ORIGINAL index = the stable order (rank counting, rows are
//     short: <= ~150 items) and writes the permutation plus the gathered payload arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

struct GroupLaunch {
  int njobs;
  int blk_start[DDP_MAX_LIST_JOBS + 1];
  ddp_group_job_t job[DDP_MAX_LIST_JOBS];
};

__device__ __forceinline__ int group_job_of(const GroupLaunch& L, int b) {
  int j = 0;
  while (j + 1 < L.njobs && b >= L.blk_start[j + 1]) ++j;
  return j;
}

__device__ __forceinline__ int group_count(const ddp_group_job_t& J) {
  if (!J.n_items_dev) return J.n_items;
  const int n = *J.n_items_dev;
  return n < J.n_items ? (n < 0 ? 0 : n) : J.n_items;
}

// phase 0 / 1 / 3 run over ITEMS (grid sized for the capacity): ZERO the row counters (blocks over keys), HIST, SLOT
template <int PHASE>
__global__ __launch_bounds__(256) void ddp_group_items_kernel(const GroupLaunch L) {
  const int j = group_job_of(L, blockIdx.x);
  const ddp_group_job_t& J = L.job[j];
  const int i = ((int)blockIdx.x - L.blk_start[j]) * 256 + (int)threadIdx.x;
  if (PHASE == 0) {
    if (i < J.n_keys) J.scratch[i] = 0;
    return;
  }
  if (i >= group_count(J)) return;
  if (PHASE == 1) atomicAdd(&J.scratch[J.key[i]], 1);                          // a count does not depend on the atomics' order
  else (J.scratch + J.n_keys)[atomicAdd(&J.scratch[J.key[i]], 1)] = i;          // a slot of the row, arbitrary order inside it
}

// one wave per row: items of the row (original indices, arbitrary order in tmp) -> ascending order by rank counting
__global__ __launch_bounds__(256) void ddp_row_order_kernel(const GroupLaunch L) {
  const int j = group_job_of(L, blockIdx.x);
  const ddp_group_job_t& J = L.job[j];
  const int row = ((int)blockIdx.x - L.blk_start[j]) * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (row >= J.n_keys) return;
  const int32_t* __restrict__ tmp = J.scratch + J.n_keys;
  const int b = J.rowptr[row], m = J.rowptr[row + 1] - b;
  const int okey = J.key_map ? J.key_map[row] : row;
  for (int a = lane; a < m; a += 64) {
    const int v = tmp[b + a];
    int rank = 0;
    for (int q = 0; q < m; ++q) rank += tmp[b + q] < v;      // indices are distinct
    const int p = b + rank;
    J.perm[p] = v;
    if (J.out_key) J.out_key[p] = okey;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (J.out[k]) J.out[k][p] = J.pay[k][v];
  }
}

template <typename K>
static void group_launch(K kernel, GroupLaunch& L, const ddp_group_job_t* jobs, int njobs, int per_block, bool by_keys, bool need_perm,
                         hipStream_t st) {
  L.njobs = 0;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    const int n = by_keys ? jobs[i].n_keys : jobs[i].n_items;
    if (n <= 0 || (need_perm && !jobs[i].perm)) continue;
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = jobs[i];
    blocks += (n + per_block - 1) / per_block;
  }
  L.blk_start[L.njobs] = blocks;
  if (blocks > 0) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, st, L);
}

extern "C" int ddp_group_by_key_jobs(const ddp_group_job_t* jobs, int njobs, void* stream) {
  if (njobs < 0 || njobs > DDP_MAX_LIST_JOBS) return ddp_fail(DDP_ELIMIT, "ddp_group_by_key_jobs: njobs");
  if (njobs == 0) return 0;
  if (!jobs) return ddp_fail(DDP_EINVAL, "ddp_group_by_key_jobs: null jobs");
  ddp_scan_job_t scans[DDP_MAX_LIST_JOBS];
  for (int i = 0; i < njobs; ++i) {
    const ddp_group_job_t& J = jobs[i];
    if (J.n_items < 0 || J.n_keys < 1) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: n_items / n_keys");
    if (!J.rowptr || !J.scratch) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: null rowptr / scratch");
    if (J.n_items > 0 && !J.key) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: null key");
    if (!J.perm && (J.out_key || J.out[0] || J.out[1] || J.out[2])) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: outputs without perm");
    for (int k = 0; k < 3; ++k)
      if (J.out[k] && !J.pay[k]) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: output without payload");
    ddp_scan_job_t& S = scans[i];
    S = ddp_scan_job_t{};
    S.n = J.n_keys;
    S.val = J.scratch;        // the row counters ...
    S.excl = J.rowptr;
    S.excl2 = J.scratch;      // ... become the rows' cursors
  }
  hipStream_t st = (hipStream_t)stream;
  GroupLaunch L;
  group_launch(ddp_group_items_kernel<0>, L, jobs, njobs, 256, true, false, st);
  group_launch(ddp_group_items_kernel<1>, L, jobs, njobs, 256, false, false, st);
  if (int rc = ddp_scan_jobs(scans, njobs, stream)) return rc;
  group_launch(ddp_group_items_kernel<2>, L, jobs, njobs, 256, false, true, st);   // perm == NULL: row pointers only
  group_launch(ddp_row_order_kernel, L, jobs, njobs, 4, true, true, st);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_group_by_key launch");
  return 0;
}

extern "C" int ddp_group_by_key(const int32_t* key, int n_items, int n_keys, const int32_t* pay0, const int32_t* pay1,
                                const int32_t* pay2, int32_t* rowptr, int32_t* perm, int32_t* out_key, int32_t* out0,
                                int32_t* out1, int32_t* out2, int32_t* scratch, void* stream) {
  ddp_group_job_t J = {};
  J.key = key; J.n_items = n_items; J.n_keys = n_keys;
  J.pay[0] = pay0; J.pay[1] = pay1; J.pay[2] = pay2;
  J.rowptr = rowptr; J.perm = perm; J.out_key = out_key;
  J.out[0] = out0; J.out[1] = out1; J.out[2] = out2;
  J.scratch = scratch;
  return ddp_group_by_key_jobs(&J, 1, stream);
}
