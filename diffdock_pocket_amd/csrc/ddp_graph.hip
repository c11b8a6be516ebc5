// ddp_graph.hip - neighbour search of the score-model forward on the device (include/ddp_hip.h: ddp_radius_count /
// ddp_radius_fill / ddp_knn), the counterpart of the torch_cluster calls inside the reference forward
// (models/all_atom_score_model.py:457,524,545-564,607,627; conventions restated in SURVEY Appendix B.3):
//   radius(x, y, r, batch_x, batch_y, max_num_neighbors): for every query y all x of the SAME graph with |x - y|^2 < r^2
//     (strict), query-major, ascending x; more matches than the cap -> the FIRST cap matches in ascending x index (what
//     torch_cluster's CUDA kernel, the one the reference runs on a GPU, keeps), or with DDP_RADIUS_NEAREST the cap nearest
//     (ties at the cut distance kept)
//   knn_graph(x, k, batch): the k nearest other nodes of the same graph, nearest first
// Graphs are contiguous node ranges (x_ptr[g] .. x_ptr[g+1]).  radius and kNN: one wave per query; the
// points of a graph are a few kB and stay in L1/L2.  Distances are formed
// exactly like the dense PyTorch formulation this replaces (differences, then (dx^2 + dy^2) + dz^2 with separate roundings:
// no FMA contraction), so the strict comparisons select the same pairs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

#define DDP_KNN_MAX 32

__device__ __forceinline__ float sqdist(const float* __restrict__ a, const float* __restrict__ b) {
  // contraction off: with it hipcc fuses some instances of this expression into FMAs and not others, so the SAME pair
  // evaluates to values one ulp apart in the counting and the selecting loop (seen as off-by-one cuts), and no instance
  // matches the separately rounded products of the dense formulation
#pragma clang fp contract(off)
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}

// point j of `p`, divided by `div` when the search runs on per-graph scaled coordinates (the reference's dynamic cross
// cutoff divides both point sets by 3 sigma + 20 and searches with r = 1, models/all_atom_score_model.py:548-556: the same
// correctly rounded divisions, so the strict comparison sees the same values)
__device__ __forceinline__ void load_pt(const float* __restrict__ p, size_t j, bool scaled, float div, float out[3]) {
  out[0] = p[3 * j];
  out[1] = p[3 * j + 1];
  out[2] = p[3 * j + 2];
  if (scaled) {
    out[0] = __fdiv_rn(out[0], div);
    out[1] = __fdiv_rn(out[1], div);
    out[2] = __fdiv_rn(out[2], div);
  }
}

__device__ __forceinline__ float sqdist_to(const float* __restrict__ yq, const float* __restrict__ x, size_t j, bool scaled, float div) {
  float xj[3];
  load_pt(x, j, scaled, div, xj);
  return sqdist(yq, xj);
}

// cut distance of a capped query: the cap-th smallest squared distance among the matches (d2 < r2); matches with
// d2 <= cut are kept.  cap <= DDP_CUT_LIST: one pass that keeps the cap smallest distances in a sorted register list
// (the side-chain torsion head searches 1111-atom graphs with cap 32).  Larger caps (they only bind on graphs with more
// than that many neighbours inside r, which the default 10000 never does): O(matches * n) counting.
#define DDP_CUT_LIST 64
__device__ float radius_cut(const float* __restrict__ x, int j0, int j1, const float* __restrict__ yq, float r2, int cap, bool scaled,
                            float div) {
  if (cap <= DDP_CUT_LIST) {
    float best[DDP_CUT_LIST];
#pragma unroll
    for (int i = 0; i < DDP_CUT_LIST; ++i) best[i] = __builtin_inff();
    float worst = __builtin_inff();                 // = best[cap - 1]
    for (int j = j0; j < j1; ++j) {
      float d = sqdist_to(yq, x, (size_t)j, scaled, div);
      if (!(d < r2) || !(d < worst)) continue;
      bool shifting = false;
#pragma unroll
      for (int i = 0; i < DDP_CUT_LIST; ++i) {
        if (i < cap && (shifting || d < best[i])) {
          const float t = best[i];
          best[i] = d;
          d = t;
          shifting = true;
        }
        if (i == cap - 1) worst = best[i];
      }
    }
    return worst;
  }
  float cut = r2;
  for (int i = j0; i < j1; ++i) {
    const float di = sqdist_to(yq, x, (size_t)i, scaled, div);
    if (!(di < r2) || !(di < cut)) continue;
    int le = 0;
    for (int j = j0; j < j1; ++j) {
      const float dj = sqdist_to(yq, x, (size_t)j, scaled, div);
      le += (dj < r2 && dj <= di) ? 1 : 0;
    }
    if (le >= cap) cut = di;   // smallest d with #(d2 <= d) >= cap
  }
  return cut;
}

// One WAVE per query (the ligand-side searches have ~1.5 k queries against ~1.1 k points each: one thread per query
// leaves the chip empty and every thread in a 1111-iteration latency chain): lane l tests points j0 + l, + 64, ...;
// a 64-point chunk's matches are placed with a ballot / prefix-popcount, which keeps the ascending-x order.
// One launch serves several searches (jobs); a fill pass never writes behind a job's capacity.
struct RadiusLaunch {
  int njobs;
  int blk_start[DDP_MAX_LIST_JOBS + 1];
  ddp_radius_job_t job[DDP_MAX_LIST_JOBS];
};

template <bool FILL>
__global__ __launch_bounds__(256) void ddp_radius_kernel(const RadiusLaunch L) {
  int jb_ = 0;
  while (jb_ + 1 < L.njobs && (int)blockIdx.x >= L.blk_start[jb_ + 1]) ++jb_;
  const ddp_radius_job_t& J = L.job[jb_];
  const int q = ((int)blockIdx.x - L.blk_start[jb_]) * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (q >= J.ny) return;
  if (FILL && !J.out_x) return;                      // a count-only job
  const float* __restrict__ x = J.x;
  const int g = J.y_batch[q];
  const int j0 = J.x_ptr[g], j1 = J.x_ptr[g + 1];
  const bool scaled = J.graph_div != nullptr;
  const float div = scaled ? J.graph_div[g] : 1.f;
  const float r2 = J.r * J.r;
  const int cap = J.max_neighbors, flags = J.flags;
  float yq[3];
  load_pt(J.y, (size_t)q, scaled, div, yq);
  int n = 0;
  for (int j = j0 + lane; j < j1; j += 64) n += (sqdist_to(yq, x, (size_t)j, scaled, div) < r2) ? 1 : 0;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
  const bool drop_self = (flags & DDP_RADIUS_DROP_SELF) != 0, nearest = (flags & DDP_RADIUS_NEAREST) != 0;
  float lim = r2;           // keep d2 < r2 ...
  bool capped = false;
  if (n > cap && nearest) { // ... or, capped to the nearest, d2 <= cut (rare: one lane selects, the wave takes its answer)
    float c = 0.f;
    if (lane == 0) c = radius_cut(x, j0, j1, yq, r2, cap, scaled, div);
    lim = __shfl(c, 0);
    capped = true;
  }
  int kept = 0, seen = 0;   // pairs emitted / matches met so far (the self pair counts towards the cap, then is dropped)
  const int obase = FILL ? J.offsets[q] : 0;
  for (int jb = j0; jb < j1; jb += 64) {
    const int j = jb + lane;
    bool match = false;
    if (j < j1) {
      const float d = sqdist_to(yq, x, (size_t)j, scaled, div);
      match = capped ? (d < r2 && d <= lim) : (d < r2);
    }
    const unsigned long long mm = __ballot(match);
    if (!nearest) match = match && (seen + __popcll(mm & ((1ull << lane) - 1ull)) < cap);   // first `cap` in index order
    seen += __popcll(mm);
    const bool ok = match && !(drop_self && j == q);
    const unsigned long long mk = __ballot(ok);
    if (FILL && ok) {
      const int o = obase + kept + __popcll(mk & ((1ull << lane) - 1ull));
      if (o < J.capacity) {
        J.out_query[o] = q;
        J.out_x[o] = j;
      } else if (J.overflow) {
        *J.overflow = 1;
      }
    }
    kept += __popcll(mk);
    if (!nearest && seen >= cap) break;   // (wave-uniform)
  }
  if (!FILL && lane == 0) J.counts[q] = kept;
}

// kNN: one wave per query.  Round t selects the t-th nearest: every lane scans its share of the graph's points for the
// smallest (distance, index) pair that comes strictly after the previous selection in that lexicographic order, a
// shuffle butterfly takes the wave-wide minimum.  Equal distances: lower index first (the order of a stable sort by
// distance, i.e. of the dense top-k formulation); NaN / infinite distances are never selected (-1 in the output).
// Distances are re-evaluated per round from L1 (k * n / 64 evaluations per lane, 144 for the 1111-atom pocket at k = 8).
__global__ __launch_bounds__(256) void ddp_knn_kernel(const float* __restrict__ x, const int32_t* __restrict__ x_ptr,
                                                      const int32_t* __restrict__ batch, int n, int k,
                                                      int32_t* __restrict__ out_nb /*[n][k], -1 padded*/) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= n) return;
  const int g = batch[q];
  const int j0 = x_ptr[g], j1 = x_ptr[g + 1];
  const float yq[3] = {x[3 * (size_t)q], x[3 * (size_t)q + 1], x[3 * (size_t)q + 2]};
  const int none = 0x7fffffff;
  float pd = -1.f;   // previous selection (squared distances are >= 0)
  int pj = -1;
  // graphs of up to 64 * CAP points (the 1111-atom pocket of 3dpf: 18 per lane): a lane's distances are evaluated ONCE and kept in
  // registers; the k selection rounds then only compare (same comparisons, same result as the loop below)
  constexpr int CAP = 24;
  if (j1 - j0 <= 64 * CAP) {
    float dl[CAP];
#pragma unroll
    for (int c = 0; c < CAP; ++c) {
      const int j = j0 + lane + 64 * c;
      float d = __builtin_inff();
      if (j < j1 && j != q) d = sqdist(yq, x + 3 * (size_t)j);
      dl[c] = (d < __builtin_inff()) ? d : __builtin_inff();        // NaN or inf: never selected
    }
    for (int t = 0; t < k; ++t) {
      float bd = __builtin_inff();
      int bj = none;
#pragma unroll
      for (int c = 0; c < CAP; ++c) {
        const int j = j0 + lane + 64 * c;
        const float d = dl[c];
        const bool after = (d > pd) || (d == pd && j > pj);
        if (d < __builtin_inff() && after && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const int oj = __shfl_xor(bj, off);
        if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
      }
      if (bj == none) {                                              // fewer than k candidates: pad the rest
        for (int u = t + lane; u < k; u += 64) out_nb[(size_t)q * k + u] = -1;
        return;
      }
      if (lane == 0) out_nb[(size_t)q * k + t] = bj;
      pd = bd;
      pj = bj;
    }
    return;
  }
  for (int t = 0; t < k; ++t) {
    float bd = __builtin_inff();
    int bj = none;
    for (int j = j0 + lane; j < j1; j += 64) {
      if (j == q) continue;
      const float d = sqdist(yq, x + 3 * (size_t)j);
      if (!(d < __builtin_inff())) continue;                       // NaN or inf
      const bool after = (d > pd) || (d == pd && j > pj);
      if (after && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float od = __shfl_xor(bd, off);
      const int oj = __shfl_xor(bj, off);
      if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
    }
    if (bj == none) {                                              // fewer than k candidates: pad the rest
      for (int u = t + lane; u < k; u += 64) out_nb[(size_t)q * k + u] = -1;
      return;
    }
    if (lane == 0) out_nb[(size_t)q * k + t] = bj;
    pd = bd;
    pj = bj;
  }
}

static int radius_job_ok(const ddp_radius_job_t& J) {
  if (J.ny < 0 || J.max_neighbors < 1 || !(J.r > 0.f)) return ddp_fail(DDP_EINVAL, "ddp_radius: ny / max_neighbors / r");
  if (J.ny > 0 && (!J.x || !J.x_ptr || !J.y || !J.y_batch)) return ddp_fail(DDP_EINVAL, "ddp_radius: null argument");
  return 0;
}

template <bool FILL>
static int radius_launch(const ddp_radius_job_t* jobs, int njobs, void* stream) {
  RadiusLaunch L;
  L.njobs = 0;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    if (jobs[i].ny <= 0 || (FILL && !jobs[i].out_x)) continue;
    L.blk_start[L.njobs] = blocks;
    L.job[L.njobs++] = jobs[i];
    blocks += (jobs[i].ny + 3) / 4;
  }
  L.blk_start[L.njobs] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL((ddp_radius_kernel<FILL>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_radius launch");
  return 0;
}

extern "C" int ddp_radius_count(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_batch, int ny, float r,
                                int max_neighbors, int flags, int32_t* counts, void* stream) {
  ddp_radius_job_t J = {};
  J.x = x; J.x_ptr = x_ptr; J.y = y; J.y_batch = y_batch; J.ny = ny; J.r = r; J.max_neighbors = max_neighbors; J.flags = flags;
  J.counts = counts;
  if (int rc = radius_job_ok(J)) return rc;
  if (ny == 0) return 0;
  if (!counts) return ddp_fail(DDP_EINVAL, "ddp_radius_count: null counts");
  return radius_launch<false>(&J, 1, stream);
}

extern "C" int ddp_radius_fill(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_batch, int ny, float r,
                               int max_neighbors, int flags, const int32_t* offsets, int32_t* out_query, int32_t* out_x,
                               void* stream) {
  ddp_radius_job_t J = {};
  J.x = x; J.x_ptr = x_ptr; J.y = y; J.y_batch = y_batch; J.ny = ny; J.r = r; J.max_neighbors = max_neighbors; J.flags = flags;
  J.offsets = const_cast<int32_t*>(offsets); J.out_query = out_query; J.out_x = out_x; J.capacity = 0x7fffffff;
  if (int rc = radius_job_ok(J)) return rc;
  if (ny == 0) return 0;
  if (!offsets || !out_query || !out_x) return ddp_fail(DDP_EINVAL, "ddp_radius_fill: null argument");
  return radius_launch<true>(&J, 1, stream);
}

// Several searches without a host round trip: count pass -> exclusive scan of the per-query counts (offsets[q] = base +
// matches of the queries before q; offsets[ny] and *total = base + all matches) -> fill pass.  Three launches for all jobs.
extern "C" int ddp_radius_search_jobs(const ddp_radius_job_t* jobs, int njobs, void* stream) {
  if (njobs < 0 || njobs > DDP_MAX_LIST_JOBS) return ddp_fail(DDP_ELIMIT, "ddp_radius_search_jobs: njobs");
  if (njobs == 0) return 0;
  if (!jobs) return ddp_fail(DDP_EINVAL, "ddp_radius_search_jobs: null jobs");
  ddp_scan_job_t scans[DDP_MAX_LIST_JOBS];
  int ns = 0;
  for (int i = 0; i < njobs; ++i) {
    const ddp_radius_job_t& J = jobs[i];
    if (int rc = radius_job_ok(J)) return rc;
    if (!J.counts) return ddp_fail(DDP_EINVAL, "ddp_radius_search_jobs: null counts");
    if (J.out_x && (!J.out_query || !J.offsets || J.capacity < 0)) return ddp_fail(DDP_EINVAL, "ddp_radius_search_jobs: fill outputs");
    if (!J.offsets && !J.total) continue;
    ddp_scan_job_t& S = scans[ns++];
    S = ddp_scan_job_t{};
    S.n = J.ny;
    S.val = J.counts;
    S.base = J.base;
    S.excl = J.offsets;
    S.total = J.total;
  }
  if (int rc = radius_launch<false>(jobs, njobs, stream)) return rc;
  if (int rc = ddp_scan_jobs(scans, ns, stream)) return rc;
  return radius_launch<true>(jobs, njobs, stream);
}

extern "C" int ddp_knn(const float* x, const int32_t* x_ptr, const int32_t* batch, int n, int k, int32_t* out_neighbors,
                       void* stream) {
  if (n < 0 || k < 1 || k > DDP_KNN_MAX) return ddp_fail(DDP_ELIMIT, "ddp_knn: k must be in [1, 32]");
  if (n == 0) return 0;
  if (!x || !x_ptr || !batch || !out_neighbors) return ddp_fail(DDP_EINVAL, "ddp_knn: null argument");
  hipLaunchKernelGGL(ddp_knn_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, x_ptr, batch, n, k, out_neighbors);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_knn launch");
  return 0;
}
