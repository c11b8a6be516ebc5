// ddp_views.hip - CSR / source-order views of an edge list on the device (include/ddp_hip.h: ddp_group_by_key).
//
// The conv kernels consume the edges of a conv direction grouped by RECEIVING node (CSR; the segmented mean of reference
// models/score_model.py:117 sums a node's messages in edge order) and, on the factorised path, grouped by SOURCE node.
// Both are a STABLE sort of the edge list by an integer key < n_keys plus a row-pointer array.  The PyTorch formulation
// (graph.py: torch.sort(stable=True), index_add, cumsum, gathers, dtype conversions) is ~20 small launches per view and
// eight views per forward, all in the host-paced front of the step; this is 5 launches per view with identical results:
//   1 zero the row counters            2 histogram of the keys (atomics: a count does not depend on their order)
//   3 exclusive scan -> rowptr         4 every item takes a slot of its row with an atomic (arbitrary order inside a row)
//   5 a wave per row puts the row's items into ascending ORIGINAL index = the stable order (rank counting, rows are
//     short: <= ~150 items) and writes the permutation plus the gathered payload arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

__global__ void ddp_hist_kernel(const int32_t* __restrict__ key, int E, int32_t* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) atomicAdd(&counts[key[i]], 1);
}

// exclusive prefix sum of counts[0..n) into rowptr[0..n], one workgroup of 1024 threads = 16 waves; cursor[k] = rowptr[k]
// on exit.  Wave w owns the contiguous segment [w S, (w+1) S): it walks it 64 items at a time (coalesced), scanning each
// group with shuffles and carrying the running total; the 16 segment totals are combined through LDS and added in a
// second coalesced pass.
__global__ __launch_bounds__(1024) void ddp_scan_kernel(int32_t* __restrict__ counts_cursor, int n, int32_t* __restrict__ rowptr) {
  __shared__ int seg_total[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int S = (((n + 15) / 16) + 63) & ~63;
  const int lo = min(wave * S, n), hi = min(lo + S, n);
  int run = 0;
  for (int base = lo; base < hi; base += 64) {
    const int i = base + lane;
    const int c = (i < hi) ? counts_cursor[i] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (i < hi) rowptr[i] = run + incl - c;      // exclusive inside the segment
    run += __shfl(incl, 63);
  }
  if (lane == 0) seg_total[wave] = run;
  __syncthreads();
  int offset = 0;
  for (int w = 0; w < wave; ++w) offset += seg_total[w];
  for (int i = lo + lane; i < hi; i += 64) {
    const int v = rowptr[i] + offset;
    rowptr[i] = v;
    counts_cursor[i] = v;
  }
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < 16; ++w) total += seg_total[w];
    rowptr[n] = total;
  }
}

__global__ void ddp_slot_kernel(const int32_t* __restrict__ key, int E, int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) tmp[atomicAdd(&cursor[key[i]], 1)] = i;
}

// one wave per row: items of the row (original indices, arbitrary order in tmp) -> ascending order by rank counting
__global__ __launch_bounds__(256) void ddp_row_order_kernel(const int32_t* __restrict__ rowptr, int n_keys,
                                                            const int32_t* __restrict__ tmp, const int32_t* __restrict__ pay0,
                                                            const int32_t* __restrict__ pay1, const int32_t* __restrict__ pay2,
                                                            int32_t* __restrict__ perm, int32_t* __restrict__ out_key,
                                                            int32_t* __restrict__ out0, int32_t* __restrict__ out1,
                                                            int32_t* __restrict__ out2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_keys) return;
  const int b = rowptr[row], m = rowptr[row + 1] - b;
  for (int a = lane; a < m; a += 64) {
    const int v = tmp[b + a];
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += tmp[b + j] < v;      // indices are distinct
    const int p = b + rank;
    perm[p] = v;
    if (out_key) out_key[p] = row;
    if (out0) out0[p] = pay0[v];
    if (out1) out1[p] = pay1[v];
    if (out2) out2[p] = pay2[v];
  }
}

extern "C" int ddp_group_by_key(const int32_t* key, int n_items, int n_keys, const int32_t* pay0, const int32_t* pay1,
                                const int32_t* pay2, int32_t* rowptr, int32_t* perm, int32_t* out_key, int32_t* out0,
                                int32_t* out1, int32_t* out2, int32_t* scratch, void* stream) {
  if (n_items < 0 || n_keys < 1) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: n_items / n_keys");
  if (!rowptr || !scratch) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: null rowptr / scratch");
  if (n_items > 0 && !key) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: null key");
  if (!perm && (out_key || out0 || out1 || out2)) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: outputs without perm");
  if ((out0 && !pay0) || (out1 && !pay1) || (out2 && !pay2)) return ddp_fail(DDP_EINVAL, "ddp_group_by_key: output without payload");
  hipStream_t st = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(scratch, 0, sizeof(int32_t) * (size_t)n_keys, st);
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_group_by_key memset");
  int32_t* tmp = scratch + n_keys;     // scratch: n_keys counters followed by n_items slots
  if (n_items > 0)
    hipLaunchKernelGGL(ddp_hist_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, key, n_items, scratch);
  hipLaunchKernelGGL(ddp_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, n_keys, rowptr);
  if (n_items > 0 && perm) {   // perm == NULL: row pointers only (items already grouped)
    hipLaunchKernelGGL(ddp_slot_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, key, n_items, scratch, tmp);
    hipLaunchKernelGGL(ddp_row_order_kernel, dim3((n_keys + 3) / 4), dim3(256), 0, st, (const int32_t*)rowptr, n_keys,
                       (const int32_t*)tmp, pay0, pay1, pay2, perm, out_key, out0, out1, out2);
  }
  err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_group_by_key launch");
  return 0;
}
