// ddp_pose.hip - the pose update between two score-model calls (include/ddp_hip.h, ddp_pose_update): one launch for
//   modify_conformer(pos, tr_update, rot_update, torsion_updates)          reference utils/diffusion_utils.py:37-60
// = rigid move about the centre of mass (axis-angle -> matrix, utils/geometry.py:72-86), the rotatable-bond torsions applied
//   one after the other (utils/torsion.py:68-94) and the Kabsch re-alignment of the twisted conformer onto the rigid one
//   (utils/geometry.py:209-243), for all samples of the batch.  The reference does this per sample on the CPU with
//   numpy / scipy; the PyTorch-ROCm form of it is ~100 tiny launches per step and is bound by the host.
// One workgroup per sample, positions in LDS.  The optimal rotation is taken from Horn's quaternion form (largest
// eigenvector of the 4x4 matrix built from the 3x3 covariance, cyclic Jacobi in fp64 by one thread): it IS the Kabsch
// rotation including its reflection fix (always a proper rotation), without the SVD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

#define DDP_POSE_THREADS 128

__device__ __forceinline__ void rotvec_to_matrix(float vx, float vy, float vz, float* R) {
  // Rodrigues; small-angle series below 1e-6 as in sampler.rotvec_to_matrix / scipy
  const float ang = sqrtf(vx * vx + vy * vy + vz * vz);
  float a, b;
  if (ang < 1e-6f) {
    a = 1.0f - ang * ang / 6.0f;
    b = 0.5f - ang * ang / 24.0f;
  } else {
    a = sinf(ang) / ang;
    b = (1.0f - cosf(ang)) / (ang * ang);
  }
  // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  R = I + a K + b K^2
  const float xx = vx * vx, yy = vy * vy, zz = vz * vz, xy = vx * vy, xz = vx * vz, yz = vy * vz;
  R[0] = 1.f - b * (yy + zz); R[1] = -a * vz + b * xy;     R[2] = a * vy + b * xz;
  R[3] = a * vz + b * xy;     R[4] = 1.f - b * (xx + zz);  R[5] = -a * vx + b * yz;
  R[6] = -a * vy + b * xz;    R[7] = a * vx + b * yz;      R[8] = 1.f - b * (xx + yy);
}

// block-wide sum of three floats (all threads get the result); red: LDS scratch of 3 * DDP_POSE_THREADS floats
__device__ __forceinline__ void block_sum3(float& x, float& y, float& z, float* red, int tid) {
  red[tid] = x; red[DDP_POSE_THREADS + tid] = y; red[2 * DDP_POSE_THREADS + tid] = z;
  __syncthreads();
  for (int s = DDP_POSE_THREADS / 2; s > 0; s >>= 1) {
    if (tid < s) {
      red[tid] += red[tid + s];
      red[DDP_POSE_THREADS + tid] += red[DDP_POSE_THREADS + tid + s];
      red[2 * DDP_POSE_THREADS + tid] += red[2 * DDP_POSE_THREADS + tid + s];
    }
    __syncthreads();
  }
  x = red[0]; y = red[DDP_POSE_THREADS]; z = red[2 * DDP_POSE_THREADS];
  __syncthreads();
}

// largest-eigenvalue eigenvector of the symmetric 4x4 matrix A (cyclic Jacobi, fp64)
__device__ void max_eigvec4(double A[4][4], double q[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int r = 0; r < 4; ++r) (p == r ? diag : off) += A[p][r] * A[p][r];
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < 3; ++p)
      for (int r = p + 1; r < 4; ++r) {
        if (fabs(A[p][r]) < 1e-300) continue;
        const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {   // A <- A J
          const double akp = A[k][p], akr = A[k][r];
          A[k][p] = c * akp - s * akr;
          A[k][r] = s * akp + c * akr;
        }
        for (int k = 0; k < 4; ++k) {   // A <- J^T A
          const double apk = A[p][k], ark = A[r][k];
          A[p][k] = c * apk - s * ark;
          A[r][k] = s * apk + c * ark;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkr = V[k][r];
          V[k][p] = c * vkp - s * vkr;
          V[k][r] = s * vkp + c * vkr;
        }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > A[best][best]) best = k;
  for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}

__global__ __launch_bounds__(DDP_POSE_THREADS) void ddp_pose_update_kernel(const float* __restrict__ pos_in, int n_atoms,
                                                                           const float* __restrict__ tr, const float* __restrict__ rot,
                                                                           const float* __restrict__ tor, int n_tor,
                                                                           const int32_t* __restrict__ bonds,
                                                                           const uint8_t* __restrict__ mask_rotate,
                                                                           float* __restrict__ pos_out) {
  extern __shared__ float lds[];
  float* f = lds;                       // [n][3] twisted conformer
  float* rg = lds + 3 * n_atoms;        // [n][3] rigid conformer
  float* red = rg + 3 * n_atoms;        // 3 * DDP_POSE_THREADS
  __shared__ float M[12];               // a 3x3 matrix (+ a translation) published by thread 0
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ pin = pos_in + (size_t)b * n_atoms * 3;
  const float inv_n = 1.0f / (float)n_atoms;

  // rigid move about the centre: (pos - c) R^T + tr + c
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS) {
    const float x = pin[3 * i], y = pin[3 * i + 1], z = pin[3 * i + 2];
    f[3 * i] = x; f[3 * i + 1] = y; f[3 * i + 2] = z;
    sx += x; sy += y; sz += z;
  }
  block_sum3(sx, sy, sz, red, tid);
  const float cx = sx * inv_n, cy = sy * inv_n, cz = sz * inv_n;
  if (tid == 0) rotvec_to_matrix(rot[3 * b], rot[3 * b + 1], rot[3 * b + 2], M);
  __syncthreads();
  {
    const float tx = tr[3 * b] + cx, ty = tr[3 * b + 1] + cy, tz = tr[3 * b + 2] + cz;
    for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS) {
      const float x = f[3 * i] - cx, y = f[3 * i + 1] - cy, z = f[3 * i + 2] - cz;
      const float ox = M[0] * x + M[1] * y + M[2] * z + tx, oy = M[3] * x + M[4] * y + M[5] * z + ty,
                  oz = M[6] * x + M[7] * y + M[8] * z + tz;
      f[3 * i] = ox; f[3 * i + 1] = oy; f[3 * i + 2] = oz;
      rg[3 * i] = ox; rg[3 * i + 1] = oy; rg[3 * i + 2] = oz;
    }
  }
  __syncthreads();
  if (tor == nullptr || n_tor == 0) {
    for (int i = tid; i < 3 * n_atoms; i += DDP_POSE_THREADS) pos_out[(size_t)b * n_atoms * 3 + i] = rg[i];
    return;
  }

  // torsions, in bond order: atoms of mask_rotate[j] turn about the axis pos[u] - pos[v] through pos[v]
  for (int j = 0; j < n_tor; ++j) {
    const int u = bonds[2 * j], v = bonds[2 * j + 1];
    const float pvx = f[3 * v], pvy = f[3 * v + 1], pvz = f[3 * v + 2];
    if (tid == 0) {
      const float ax = f[3 * u] - pvx, ay = f[3 * u + 1] - pvy, az = f[3 * u + 2] - pvz;
      const float k = tor[(size_t)b * n_tor + j] / sqrtf(ax * ax + ay * ay + az * az);
      rotvec_to_matrix(ax * k, ay * k, az * k, M);
    }
    __syncthreads();
    const uint8_t* __restrict__ mk = mask_rotate + (size_t)j * n_atoms;
    for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS)
      if (mk[i]) {
        const float x = f[3 * i] - pvx, y = f[3 * i + 1] - pvy, z = f[3 * i + 2] - pvz;
        f[3 * i] = M[0] * x + M[1] * y + M[2] * z + pvx;
        f[3 * i + 1] = M[3] * x + M[4] * y + M[5] * z + pvy;
        f[3 * i + 2] = M[6] * x + M[7] * y + M[8] * z + pvz;
      }
    __syncthreads();
  }

  // Kabsch alignment of f (A) onto rg (B): centroids, covariance S[x][y] = sum (a - ca)_x (b - cb)_y
  float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
  for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS) {
    ax += f[3 * i]; ay += f[3 * i + 1]; az += f[3 * i + 2];
    bx += rg[3 * i]; by += rg[3 * i + 1]; bz += rg[3 * i + 2];
  }
  block_sum3(ax, ay, az, red, tid);
  block_sum3(bx, by, bz, red, tid);
  const float cax = ax * inv_n, cay = ay * inv_n, caz = az * inv_n, cbx = bx * inv_n, cby = by * inv_n, cbz = bz * inv_n;
  float S0[3] = {0.f, 0.f, 0.f}, S1[3] = {0.f, 0.f, 0.f}, S2[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS) {
    const float x = f[3 * i] - cax, y = f[3 * i + 1] - cay, z = f[3 * i + 2] - caz;
    const float p = rg[3 * i] - cbx, q = rg[3 * i + 1] - cby, r = rg[3 * i + 2] - cbz;
    S0[0] += x * p; S0[1] += x * q; S0[2] += x * r;
    S1[0] += y * p; S1[1] += y * q; S1[2] += y * r;
    S2[0] += z * p; S2[1] += z * q; S2[2] += z * r;
  }
  block_sum3(S0[0], S0[1], S0[2], red, tid);
  block_sum3(S1[0], S1[1], S1[2], red, tid);
  block_sum3(S2[0], S2[1], S2[2], red, tid);
  if (tid == 0) {
    const double Sxx = S0[0], Sxy = S0[1], Sxz = S0[2], Syx = S1[0], Syy = S1[1], Syz = S1[2], Szx = S2[0], Szy = S2[1], Szz = S2[2];
    double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double q[4];
    max_eigvec4(N, q);
    const double nq = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] * nq, x = q[1] * nq, y = q[2] * nq, z = q[3] * nq;
    const float R[9] = {(float)(1 - 2 * (y * y + z * z)), (float)(2 * (x * y - z * w)), (float)(2 * (x * z + y * w)),
                        (float)(2 * (x * y + z * w)), (float)(1 - 2 * (x * x + z * z)), (float)(2 * (y * z - x * w)),
                        (float)(2 * (x * z - y * w)), (float)(2 * (y * z + x * w)), (float)(1 - 2 * (x * x + y * y))};
    for (int k = 0; k < 9; ++k) M[k] = R[k];
    // t = cb - R ca
    M[9] = cbx - (R[0] * cax + R[1] * cay + R[2] * caz);
    M[10] = cby - (R[3] * cax + R[4] * cay + R[5] * caz);
    M[11] = cbz - (R[6] * cax + R[7] * cay + R[8] * caz);
  }
  __syncthreads();
  float* __restrict__ po = pos_out + (size_t)b * n_atoms * 3;
  for (int i = tid; i < n_atoms; i += DDP_POSE_THREADS) {
    const float x = f[3 * i], y = f[3 * i + 1], z = f[3 * i + 2];
    po[3 * i] = M[0] * x + M[1] * y + M[2] * z + M[9];
    po[3 * i + 1] = M[3] * x + M[4] * y + M[5] * z + M[10];
    po[3 * i + 2] = M[6] * x + M[7] * y + M[8] * z + M[11];
  }
}

extern "C" int ddp_pose_update(const float* pos_in, int n_samples, int n_atoms, const float* tr, const float* rot,
                               const float* tor, int n_tor, const int32_t* bonds, const uint8_t* mask_rotate, float* pos_out,
                               void* stream) {
  if (n_samples <= 0 || n_atoms <= 0) return 0;
  if (!pos_in || !tr || !rot || !pos_out) return ddp_fail(DDP_EINVAL, "ddp_pose_update: null argument");
  if (n_tor < 0 || (n_tor > 0 && tor && (!bonds || !mask_rotate))) return ddp_fail(DDP_EINVAL, "ddp_pose_update: torsion arguments");
  const size_t lds = (size_t)(6 * n_atoms + 3 * DDP_POSE_THREADS) * sizeof(float);
  if (lds > 60 * 1024) return ddp_fail(DDP_ELIMIT, "ddp_pose_update: more than 2400 atoms per sample");
  hipLaunchKernelGGL(ddp_pose_update_kernel, dim3(n_samples), dim3(DDP_POSE_THREADS), lds, (hipStream_t)stream, pos_in, n_atoms,
                     tr, rot, tor, n_tor, bonds, mask_rotate, pos_out);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_pose_update launch");
  return 0;
}

// ---- side-chain torsion update (modify_sidechains, utils/diffusion_utils.py:63-70 / utils/torsion.py:251-278): for every
// sample the flexible-residue bonds are applied one after the other; bond j turns the atoms subcomponents[map[j][0] ..
// map[j][1]) by angles[s][j] about pos[u_j] - pos[v_j] through pos[v_j].  One workgroup per sample; the PyTorch form is
// ~9 launches per bond.
__global__ __launch_bounds__(64) void ddp_sidechain_update_kernel(const float* __restrict__ pos_in, int n_atoms,
                                                                  const float* __restrict__ angles, int n_bonds,
                                                                  const int32_t* __restrict__ edge_idx,
                                                                  const int32_t* __restrict__ subcomponents,
                                                                  const int32_t* __restrict__ mapping,
                                                                  float* __restrict__ pos_out) {
  __shared__ float M[12];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ pin = pos_in + (size_t)b * n_atoms * 3;
  float* po = pos_out + (size_t)b * n_atoms * 3;
  if (pin != po)
    for (int i = tid; i < 3 * n_atoms; i += 64) po[i] = pin[i];
  __syncthreads();
  for (int j = 0; j < n_bonds; ++j) {
    const int u = edge_idx[2 * j], v = edge_idx[2 * j + 1];
    if (tid == 0) {
      const float pvx = po[3 * v], pvy = po[3 * v + 1], pvz = po[3 * v + 2];
      const float ax = po[3 * u] - pvx, ay = po[3 * u + 1] - pvy, az = po[3 * u + 2] - pvz;
      const float k = angles[(size_t)b * n_bonds + j] / sqrtf(ax * ax + ay * ay + az * az);
      rotvec_to_matrix(ax * k, ay * k, az * k, M);
      M[9] = pvx; M[10] = pvy; M[11] = pvz;
    }
    __syncthreads();
    const int m0 = mapping[2 * j], m1 = mapping[2 * j + 1];
    for (int i = m0 + tid; i < m1; i += 64) {
      const int a = subcomponents[i];
      const float x = po[3 * a] - M[9], y = po[3 * a + 1] - M[10], z = po[3 * a + 2] - M[11];
      po[3 * a] = M[0] * x + M[1] * y + M[2] * z + M[9];
      po[3 * a + 1] = M[3] * x + M[4] * y + M[5] * z + M[10];
      po[3 * a + 2] = M[6] * x + M[7] * y + M[8] * z + M[11];
    }
    __syncthreads();   // block-wide visibility of the moved atoms before the next bond reads its axis
  }
}

extern "C" int ddp_sidechain_update(const float* pos_in, int n_samples, int n_atoms, const float* angles, int n_bonds,
                                    const int32_t* edge_idx, const int32_t* subcomponents, const int32_t* mapping, float* pos_out,
                                    void* stream) {
  if (n_samples <= 0 || n_atoms <= 0) return 0;
  if (!pos_in || !pos_out || (n_bonds > 0 && (!angles || !edge_idx || !subcomponents || !mapping)))
    return ddp_fail(DDP_EINVAL, "ddp_sidechain_update: null argument");
  hipLaunchKernelGGL(ddp_sidechain_update_kernel, dim3(n_samples), dim3(64), 0, (hipStream_t)stream, pos_in, n_atoms, angles,
                     n_bonds, edge_idx, subcomponents, mapping, pos_out);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_sidechain_update launch");
  return 0;
}

// ---- scores -> pose updates: the reverse SDE / ODE step of utils/sampling.py:146-193 for the four components at once,
//   out[k][i] = a[k] * score[k][i] + b[k] * z[k][i],   coef = [a_tr b_tr a_rot b_rot a_tor b_tor a_sc b_sc] in DEVICE memory
// (a = g^2 dt (lambda + temp psi / 2), b = g sqrt(dt (1 + psi)) with low-temperature sampling; the host writes them per step
// together with the noise, so a captured step needs no kernel arguments that change).  Products and sum are rounded
// separately, as the PyTorch expression `a * score + b * z` does.
struct SdeLaunch {
  ddp_sde_args_t a;
};
__global__ __launch_bounds__(256) void ddp_sde_update_kernel(const float* __restrict__ coef, const SdeLaunch L) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i < L.a.n[k]) {
      const float p = coef[2 * k] * L.a.score[k][i];
      const float q = L.a.z[k] ? coef[2 * k + 1] * L.a.z[k][i] : 0.f;
      L.a.out[k][i] = L.a.z[k] ? p + q : p;
    }
  }
}

extern "C" int ddp_sde_update(const float* coef, const ddp_sde_args_t* args, void* stream) {
  if (!coef || !args) return ddp_fail(DDP_EINVAL, "ddp_sde_update: null argument");
  int nmax = 0;
  for (int k = 0; k < 4; ++k) {
    if (args->n[k] < 0 || (args->n[k] > 0 && (!args->score[k] || !args->out[k]))) return ddp_fail(DDP_EINVAL, "ddp_sde_update: component");
    if (args->n[k] > nmax) nmax = args->n[k];
  }
  if (nmax == 0) return 0;
  SdeLaunch L;
  L.a = *args;
  hipLaunchKernelGGL(ddp_sde_update_kernel, dim3((nmax + 255) / 256), dim3(256), 0, (hipStream_t)stream, coef, L);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ddp_fail_hip(err, "ddp_sde_update launch");
  return 0;
}
