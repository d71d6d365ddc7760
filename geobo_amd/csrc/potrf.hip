// potrf.hip -- blocked lower Cholesky + explicit L^-1 on gfx950 (replaces scipy.linalg.cholesky /
// solve_triangular, inversion.py:100,105,114), plus u = L^-1 y and the log-likelihood statistics
// (inversion.py:105-110).
//
// Right-looking, block size 128:
//   potf2_inv_kernel  one workgroup factors the 128x128 diagonal block inside LDS (132 KiB of the CU's
//                     160 KiB), writes L_kk, then inverts it in place in LDS (dtrti2 order) and writes
//                     L_kk^-1 straight into the diagonal block of Linv;
//   panel solve       P = A[k+1:,k] * (L_kk^-1)^T           -> geobo_gemm_nt on the fp64 MFMA core (in place)
//   trailing update   A[k+1:,k+1:] -= P P^T (lower tiles)   -> geobo_gemm_nt, lower_only
// L^-1 is then assembled by recursive halving, two MFMA GEMMs per merge:
//   Linv[hi,lo] = -Linv[hi,hi] * (L[hi,lo] * Linv[lo,lo]).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

constexpr int NB = 128;
constexpr int LS = NB + 1;  // LDS row stride (doubles): conflict-free column walks

__global__ void __launch_bounds__(256) potf2_inv_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ Linv,
                                                        int64_t ldi, int kb_global, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [NB][LS]
  const int tid = threadIdx.x;
  // load the block (rows are 1 KiB contiguous)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx >> 7, c = idx & (NB - 1);
    S[i * LS + c] = A[(int64_t)i * ld + c];
  }
  const int ti = tid >> 4, tj = tid & 15;
  bool bad_seen = false;
  for (int j = 0; j < NB; ++j) {
    __syncthreads();
    const double d = S[j * LS + j];
    if (!(d > 0.0) && !bad_seen) {  // non-positive or NaN pivot: LAPACK dpotrf's info = j (1-based)
      bad_seen = true;
      if (tid == 0 && *info == 0) *info = kb_global + j + 1;
    }
    const double r = sqrt(d);
    const double rinv = 1.0 / r;
    __syncthreads();
    if (tid < NB) {
      if (tid > j) S[tid * LS + j] *= rinv;
      else if (tid == j) S[j * LS + j] = r;
    }
    __syncthreads();
    for (int i = j + 1 + ti; i < NB; i += 16) {
      const double lij = S[i * LS + j];
      for (int c = j + 1 + tj; c <= i; c += 16) S[i * LS + c] -= lij * S[c * LS + j];
    }
  }
  __syncthreads();
  // write L_kk (upper part zeroed, like scipy's lower=True result)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx >> 7, c = idx & (NB - 1);
    A[(int64_t)i * ld + c] = (c <= i) ? S[i * LS + c] : 0.0;
  }
  // in-place inverse of the lower triangle, last column first
  for (int j = NB - 1; j >= 0; --j) {
    __syncthreads();
    const double ajj = 1.0 / S[j * LS + j];
    double acc = 0.0;
    if (tid < NB && tid > j) {
      for (int k = j + 1; k <= tid; ++k) acc = __builtin_fma(S[tid * LS + k], S[k * LS + j], acc);
    }
    __syncthreads();
    if (tid < NB) {
      if (tid > j) S[tid * LS + j] = -ajj * acc;
      else if (tid == j) S[j * LS + j] = ajj;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx >> 7, c = idx & (NB - 1);
    Linv[(int64_t)i * ldi + c] = (c <= i) ? S[i * LS + c] : 0.0;
  }
}

// u[i] = sum_{k<=i} Linv[i,k] y[k]: one wavefront per row, shuffle-tree reduction
__global__ void __launch_bounds__(256) trmv_lower_kernel(int64_t m, const double* __restrict__ Linv, int64_t ldi,
                                                         const double* __restrict__ y, double* __restrict__ u) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const double* Lr = Linv + row * ldi;
  double acc = 0.0;
  for (int64_t k = lane; k <= row; k += 64) acc = __builtin_fma(Lr[k], y[k], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) u[row] = acc;
}

// stats[0] = u.u, stats[1] = sum log(L_ii^2); single workgroup, fixed summation order (deterministic)
__global__ void __launch_bounds__(256) logl_stats_kernel(int64_t m, const double* __restrict__ u,
                                                         const double* __restrict__ L, int64_t ld,
                                                         double* __restrict__ stats) {
  __shared__ double s0[256], s1[256];
  const int tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int64_t i = tid; i < m; i += 256) {
    const double ui = u[i], d = L[i * ld + i];
    a = __builtin_fma(ui, ui, a);
    b += log(d * d);
  }
  s0[tid] = a; s1[tid] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { s0[tid] += s0[tid + w]; s1[tid] += s1[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) { stats[0] = s0[0]; stats[1] = s1[0]; }
}

int build_inverse(int lo, int hi, const double* L, int64_t ld, double* Linv, int64_t ldi, double* T, void* st) {
  if (hi - lo <= 1) return GEOBO_OK;
  const int mid = (lo + hi) / 2;
  int rc = build_inverse(lo, mid, L, ld, Linv, ldi, T, st);
  if (rc) return rc;
  rc = build_inverse(mid, hi, L, ld, Linv, ldi, T, st);
  if (rc) return rc;
  const int64_t r = (int64_t)(hi - mid) * NB, c = (int64_t)(mid - lo) * NB;
  const int64_t o_lo = (int64_t)lo * NB, o_mid = (int64_t)mid * NB;
  // T = L[mid:hi, lo:mid] * Linv[lo:mid, lo:mid]           (Y lower triangular)
  rc = geobo_gemm_nn(r, c, c, 1.0, L + o_mid * ld + o_lo, ld, Linv + o_lo * ldi + o_lo, ldi, 0.0, T, c, 0, 1, st);
  if (rc) return rc;
  // Linv[mid:hi, lo:mid] = -Linv[mid:hi, mid:hi] * T         (X lower triangular)
  return geobo_gemm_nn(r, c, r, -1.0, Linv + o_mid * ldi + o_mid, ldi, T, c, 0.0, Linv + o_mid * ldi + o_lo, ldi, 1, 0, st);
}

}  // namespace

extern "C" size_t geobo_potrf_ws_bytes(int64_t m) {
  const int64_t nb = (m + NB - 1) / NB;
  const int64_t half = ((nb + 1) / 2) * NB;
  return (size_t)(half * half) * sizeof(double);
}

extern "C" int geobo_potrf_inv(int64_t m, double* A, int64_t ld, double* Linv, int64_t ldi, int* info, void* ws,
                               size_t ws_bytes, void* stream) {
  if (!A || !Linv || !info || !ws) return GEOBO_E_ARG;
  if (m <= 0 || m % NB || (ld & 1) || (ldi & 1) || ld < m || ldi < m) return GEOBO_E_ALIGN;
  if (ws_bytes < geobo_potrf_ws_bytes(m)) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set = false;
  constexpr size_t lds = sizeof(double) * NB * LS;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_set = true;
  }
  if (hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return GEOBO_E_LAUNCH;
  if (hipMemset2DAsync(Linv, (size_t)ldi * sizeof(double), 0, (size_t)m * sizeof(double), (size_t)m, st) != hipSuccess)
    return GEOBO_E_LAUNCH;
  for (int64_t kb = 0; kb < m; kb += NB) {
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), lds, st, A + kb * ld + kb, ld, Linv + kb * ldi + kb, ldi,
                       (int)kb, info);
    if (hipGetLastError() != hipSuccess) return GEOBO_E_LAUNCH;
    const int64_t rem = m - kb - NB;
    if (rem > 0) {
      double* P = A + (kb + NB) * ld + kb;
      int rc = geobo_gemm_nt(rem, NB, NB, 1.0, P, ld, Linv + kb * ldi + kb, ldi, 0.0, P, ld, 0, stream);
      if (rc) return rc;
      rc = geobo_gemm_nt(rem, rem, NB, -1.0, P, ld, P, ld, 1.0, A + (kb + NB) * ld + (kb + NB), ld, 1, stream);
      if (rc) return rc;
    }
  }
  return build_inverse(0, (int)(m / NB), A, ld, Linv, ldi, (double*)ws, stream);
}

extern "C" int geobo_trmv_stats(int64_t m, const double* Linv, int64_t ldi, const double* y, const double* L,
                                int64_t ld, double* u, double* stats, void* stream) {
  if (!Linv || !y || !L || !u || !stats || m <= 0) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(trmv_lower_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, m, Linv, ldi, y, u);
  hipLaunchKernelGGL(logl_stats_kernel, dim3(1), dim3(256), 0, st, m, u, L, ld, stats);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
