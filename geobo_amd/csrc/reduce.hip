// reduce.hip -- the two streaming reductions of the shape-independent forms of the structured algorithm: what the fused n = 64
// kernels of xz2d_fold.hip do inside their epilogues, as stand-alone HBM-bound passes for every other grid extent.
//
//   geobo_sumsq_accum   ss[slot][c] += sum_{r = slot (mod slots)} (a[r][c] + b[r][c])^2
//                       diag(K - V^T V) of inversion.py:117,238 in the transposed order V = (L^-1 A3) K: a batch of rows of V comes out
//                       of the storing covariance product (any transform path), is squared and summed over the rows here.
//                       Deterministic: a (slot, column) pair is owned by one thread, rows are added in ascending order.
//   geobo_lamdot_z      out[b][o] = sum_z D[b][o][z] * lam[b % planes][o][z]
//                       x step of the lattice Gram (AkA on a lattice survey, inversion.py:96) behind a batched GEMM D = Gx X: the
//                       eigenvalue scaling and the channel (z) sum.  Eight lanes share one (plane, o) row of nz doubles: 16-byte loads,
//                       128 contiguous bytes per lane group, three shuffle steps.
//   geobo_rowgemv       out[r] = sum_c X[r][c] * v[c]
//                       a forward operator applied to a model, data = A rho (simcube.py:147-150; the synthetic surveys of bench.py and of
//                       the tests): one workgroup per row, 16-byte loads, a fixed reduction tree (deterministic).  HBM read bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) sumsq_accum_kernel(int64_t rows, int64_t n2, const double* __restrict__ a, int64_t lda,
                                                          const double* __restrict__ b, int64_t ldb, int slots, double* __restrict__ ss,
                                                          int64_t ld_ss) {
  const int64_t c2 = (int64_t)blockIdx.x * 256 + threadIdx.x;      // column pair
  if (c2 >= n2) return;
  const int slot = blockIdx.y;
  v2d acc = *reinterpret_cast<const v2d*>(ss + slot * ld_ss + 2 * c2);
  if (b) {
    for (int64_t r = slot; r < rows; r += slots) {
      const v2d x = *reinterpret_cast<const v2d*>(a + r * lda + 2 * c2) + *reinterpret_cast<const v2d*>(b + r * ldb + 2 * c2);
      acc.x = __builtin_fma(x.x, x.x, acc.x);
      acc.y = __builtin_fma(x.y, x.y, acc.y);
    }
  } else {
    for (int64_t r = slot; r < rows; r += slots) {
      const v2d x = *reinterpret_cast<const v2d*>(a + r * lda + 2 * c2);
      acc.x = __builtin_fma(x.x, x.x, acc.x);
      acc.y = __builtin_fma(x.y, x.y, acc.y);
    }
  }
  *reinterpret_cast<v2d*>(ss + slot * ld_ss + 2 * c2) = acc;
}

// 8 lanes per (batch plane, o) row; 32 rows per 256-thread workgroup
__global__ void __launch_bounds__(256) lamdot_z_kernel(int64_t nrows, int planes, int px, int nz, const double* __restrict__ D,
                                                       const double* __restrict__ lam, double* __restrict__ out) {
  const int sub = threadIdx.x & 7;
  for (int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); row < nrows; row += (int64_t)gridDim.x * 32) {
    const int64_t bp = row / px;                                   // batch plane (r, p)
    const int o = (int)(row - bp * px);
    const double* d = D + row * nz;
    const double* l = lam + ((int64_t)(bp % planes) * px + o) * nz;
    double acc = 0.0;
    for (int z = 2 * sub; z < nz; z += 16) {
      const v2d x = *reinterpret_cast<const v2d*>(d + z), y = *reinterpret_cast<const v2d*>(l + z);
      acc = __builtin_fma(x.x, y.x, acc);
      acc = __builtin_fma(x.y, y.y, acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (sub == 0) out[row] = acc;
  }
}

__global__ void __launch_bounds__(256) rowgemv_kernel(int64_t m, int64_t n2, const double* __restrict__ X, int64_t ld,
                                                      const double* __restrict__ v, double* __restrict__ out) {
  __shared__ double part[4];
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    const v2d* x = reinterpret_cast<const v2d*>(X + r * ld);
    const v2d* w = reinterpret_cast<const v2d*>(v);
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < n2; c += 256) {
      const v2d a = x[c], b = w[c];
      acc = __builtin_fma(a.x, b.x, acc);
      acc = __builtin_fma(a.y, b.y, acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[r] = (part[0] + part[1]) + (part[2] + part[3]);
    __syncthreads();
  }
}

}  // namespace

extern "C" int geobo_rowgemv(int64_t m, int64_t n, const double* X, int64_t ld, const double* v, double* out, void* stream) {
  if (!X || !v || !out) return GEOBO_E_ARG;
  if (m <= 0 || n <= 0) return GEOBO_OK;
  if ((n & 1) || (ld & 1) || ld < n || ((uintptr_t)X & 15) || ((uintptr_t)v & 15)) return GEOBO_E_ALIGN;
  const int64_t nb = m < 256 * 16 ? m : 256 * 16;
  hipLaunchKernelGGL(rowgemv_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, m, n / 2, X, ld, v, out);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_sumsq_accum(int64_t rows, int64_t n, const double* a, int64_t lda, const double* b, int64_t ldb, int slots,
                                 double* ss, int64_t ld_ss, void* stream) {
  if (!a || !ss) return GEOBO_E_ARG;
  if (rows <= 0 || n <= 0) return GEOBO_OK;
  if (slots < 1 || slots > 65535) return GEOBO_E_ARG;
  if ((n & 1) || (lda & 1) || (ldb & 1) || (ld_ss & 1) || lda < n || (b && ldb < n) || ld_ss < n || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) ||
      ((uintptr_t)ss & 15))
    return GEOBO_E_ALIGN;
  const int64_t n2 = n / 2;
  hipLaunchKernelGGL(sumsq_accum_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)slots), dim3(256), 0, (hipStream_t)stream, rows, n2, a,
                     lda, b, ldb, slots, ss, ld_ss);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_lamdot_z(int64_t batch, int planes, int px, int nz, const double* D, const double* lam, double* out, void* stream) {
  if (!D || !lam || !out) return GEOBO_E_ARG;
  if (batch <= 0) return GEOBO_OK;
  if (planes <= 0 || px <= 0 || nz <= 0) return GEOBO_E_ARG;
  if ((nz & 15) || ((uintptr_t)D & 15) || ((uintptr_t)lam & 15)) return GEOBO_E_ALIGN;
  const int64_t nrows = batch * px;
  int64_t nb = (nrows + 31) / 32;
  if (nb > 256 * 64) nb = 256 * 64;
  hipLaunchKernelGGL(lamdot_z_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, nrows, planes, px, nz, D, lam, out);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
