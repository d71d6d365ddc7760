// assembly.hip -- materialising covariance-block assembly and the gravity / magnetic forward operator.
//
//   k_block_kernel   out[r,c] = w*amp*k(|P_r - Q_c|^2)    one block of create_cov (kernels.py:183-195) straight
//                    from voxel coordinates: D2 (kernels.py:45-61) is folded in and never stored.
//                    HBM-write bound: 8 B written per element, coordinates read once per tile.
//   k_eval_kernel    elementwise k(d2) on a caller-supplied D2 array (the literal gpkernel(D2, gamma) API).
//   a_sens_kernel    A_sens (sensormodel.py:29-93): node potentials + 8-corner stencil, one workgroup per
//                    (sensor, x-range); two node planes live in LDS and each potential is evaluated once.
//   mfma_peak_kernel v_mfma_f64_16x16x4_f64 issue-rate micro-benchmark (roofline denominator check).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "covfun.h"
#include "geobo_hip.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

// ---- k_block ---------------------------------------------------------------------------------------------
// grid: (ceil(nc / 512), ceil(nr / KB_ROWS)); 256 threads; each thread owns two adjacent columns (16-byte
// stores, 1 KiB per wavefront store instruction) and walks KB_ROWS rows whose coordinates are wave-uniform.
constexpr int KB_ROWS = 32;

template <int ID, typename OUT>
__global__ void __launch_bounds__(256) k_block_kernel(const double* __restrict__ rx, const double* __restrict__ ry,
                                                      const double* __restrict__ rz, int64_t nr,
                                                      const double* __restrict__ cx, const double* __restrict__ cy,
                                                      const double* __restrict__ cz, int64_t nc, const CovParams p,
                                                      OUT* __restrict__ out, int64_t ld) {
  const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * KB_ROWS;
  if (c0 >= nc) return;
  const bool two = (c0 + 1) < nc;
  const double qx0 = cx[c0], qy0 = cy[c0], qz0 = cz[c0];
  const double qx1 = two ? cx[c0 + 1] : qx0, qy1 = two ? cy[c0 + 1] : qy0, qz1 = two ? cz[c0 + 1] : qz0;
  typedef OUT pair_t __attribute__((ext_vector_type(2)));   // fp64: 16-byte stores; fp32 (config-5 assembly): 8-byte stores
  const bool vec = two && ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & (2 * sizeof(OUT) - 1)) == 0);
  const int rows = (int)((nr - r0) < KB_ROWS ? (nr - r0) : KB_ROWS);
  for (int i = 0; i < rows; ++i) {
    const int64_t r = r0 + i;
    const double px = rx[r], py = ry[r], pz = rz[r];  // uniform -> scalar loads
    const double v0 = p.scale * cov_eval<ID>(p, sqdist3(px, py, pz, qx0, qy0, qz0));
    const double v1 = p.scale * cov_eval<ID>(p, sqdist3(px, py, pz, qx1, qy1, qz1));
    OUT* dst = out + r * ld + c0;
    if (vec) {
      *reinterpret_cast<pair_t*>(dst) = (pair_t){(OUT)v0, (OUT)v1};
    } else {
      dst[0] = (OUT)v0;
      if (two) dst[1] = (OUT)v1;
    }
  }
}

// covariance sampled on the difference lattice of a regular grid, z axis MIRRORED so that ascending contraction index
// means ascending table index: table[(diy*nx + dix)*2nz + (dz + nz - 1)] = scale*k(|P(0,0,0) - P(diy,dix,|dz|)|^2),
// dz in [-(nz-1), nz-1] (entry 2nz-1 of each row is padding), P = (i+1)*voxel size exactly as calcGridPoints3D builds
// it (kernels.py:36-38) -- i.e. the value the coordinate path produces for the voxel pair (0, d).
template <int ID>
__global__ void __launch_bounds__(256) cov_table_kernel(int nx, int ny, int nz, double sx, double sy, double sz,
                                                        const CovParams p, double* __restrict__ table) {
  const int nz2 = 2 * nz;
  const int64_t n = (int64_t)nx * ny * nz2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int zi = (int)(i % nz2);
    const int64_t t = i / nz2;
    const int dx = (int)(t % nx), dy = (int)(t / nx);
    int dz = zi - (nz - 1);
    dz = dz < 0 ? -dz : dz;
    if (dz > nz - 1) dz = nz - 1;  // padding entry
    const double d2 = sqdist3(1.0 * sx, 1.0 * sy, 1.0 * sz, (double)(dx + 1) * sx, (double)(dy + 1) * sy, (double)(dz + 1) * sz);
    table[i] = p.scale * cov_eval<ID>(p, d2);
  }
}

// ---- k_block on the regular grid: a gather from the difference-lattice table ------------------------------------------
// out[r, c] = table[(|iy_r - iy_c| nx + |ix_r - ix_c|) 2nz + (iz_c - iz_r + nz - 1)]  for row voxel rows[r] (or row0 + r) and column
// voxel col0 + c, voxel index p = (iy nx + ix) nz + iz.  No exp / sqrt / coordinates: integer index arithmetic and one 8-byte read
// of a table that lives in L2 (4 MB at 64^3) per element, so the launch is bound by the HBM store of the block -- the
// "materialised kernel assembly" regime of SURVEY.md section 8(d).  Along a row, consecutive columns of one (iy, ix) voxel column are
// consecutive table entries: the gather is a sequence of nz-long contiguous copies.
// grid: (ceil(ncols / 512), ceil(nr / KB_ROWS)); 256 threads, two adjacent columns per thread (nz even, col0 even: both in the
// same voxel column), KB_ROWS rows per workgroup with wave-uniform row decoding on the scalar unit.
template <typename OUT, int CPT>
__global__ void __launch_bounds__(256) k_block_grid_kernel(const double* __restrict__ table, int nx, int ny, int nz,
                                                           const int64_t* __restrict__ rows, int64_t row0, int64_t nr, int64_t col0,
                                                           int64_t ncols, OUT* __restrict__ out, int64_t ld) {
  // CPT adjacent columns per thread, all in one voxel column (nz % CPT == 0, col0 % CPT == 0): one 16-byte store per thread and row
  // for fp64 (CPT = 2) and for fp32 (CPT = 4) alike -- with 8-byte stores the fp32 block went no faster than the fp64 one
  const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CPT;
  const int64_t r0 = (int64_t)blockIdx.y * KB_ROWS;
  if (c0 >= ncols) return;
  const int nv = (int)((ncols - c0) < CPT ? (ncols - c0) : CPT);
  const int64_t pc = col0 + c0;
  const int izc = (int)(pc % nz);
  const int64_t tc = pc / nz;
  const int ixc = (int)(tc % nx), iyc = (int)(tc / nx);
  const int nz2 = 2 * nz;
  typedef OUT vec_t __attribute__((ext_vector_type(CPT)));
  const bool vec = nv == CPT && (ld % CPT == 0) && ((reinterpret_cast<uintptr_t>(out) & (CPT * sizeof(OUT) - 1)) == 0);
  const int nrow = (int)((nr - r0) < KB_ROWS ? (nr - r0) : KB_ROWS);
  for (int i = 0; i < nrow; ++i) {
    const int64_t r = r0 + i;
    const int64_t pr = rows ? rows[r] : row0 + r;            // uniform -> scalar load and scalar decode
    const int izr = (int)(pr % nz);
    const int64_t tr = pr / nz;
    const int ixr = (int)(tr % nx), iyr = (int)(tr / nx);
    const int dy = iyc > iyr ? iyc - iyr : iyr - iyc, dx = ixc > ixr ? ixc - ixr : ixr - ixc;
    const double* tp = table + ((int64_t)(dy * nx + dx) * nz2 + (izc - izr + nz - 1));
    double v[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) v[k] = tp[k < nv ? k : 0];
    OUT* dst = out + r * ld + c0;
    if (vec) {
      vec_t o;
#pragma unroll
      for (int k = 0; k < CPT; ++k) o[k] = (OUT)v[k];
      *reinterpret_cast<vec_t*>(dst) = o;
    } else {
#pragma unroll
      for (int k = 0; k < CPT; ++k)
        if (k < nv) dst[k] = (OUT)v[k];
    }
  }
}

// ---- weighted column sums: out[c] = sum_r X[r, c] v[r] ---------------------------------------------------------------------
// w = Linv^T u and mu = (A K)^T w of the posterior mean (inversion.py:114-116: mu = V^T u = (A K)^T L^-T L^-1 y): a pure stream over
// X.  grid (ceil(n / 512), RS): block (bx, rs) sums its slice of the rows into part[rs][c] (two adjacent columns per thread, v[r]
// wave-uniform), colsum_finish adds the RS slices in a fixed order.
__global__ void __launch_bounds__(256) colgemv_kernel(const double* __restrict__ X, int64_t ld, int64_t m, int64_t n,
                                                      const double* __restrict__ v, int64_t rows_per, double* __restrict__ part) {
  const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (c0 >= n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = r0 + rows_per < m ? r0 + rows_per : m;
  v2d a0 = (v2d){0., 0.}, a1 = (v2d){0., 0.}, a2 = (v2d){0., 0.}, a3 = (v2d){0., 0.};
  const double* p = X + r0 * ld + c0;
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4, p += 4 * ld) {
    const v2d x0 = *reinterpret_cast<const v2d*>(p), x1 = *reinterpret_cast<const v2d*>(p + ld);
    const v2d x2 = *reinterpret_cast<const v2d*>(p + 2 * ld), x3 = *reinterpret_cast<const v2d*>(p + 3 * ld);
    a0 += x0 * v[r]; a1 += x1 * v[r + 1]; a2 += x2 * v[r + 2]; a3 += x3 * v[r + 3];
  }
  for (; r < r1; ++r, p += ld) a0 += *reinterpret_cast<const v2d*>(p) * v[r];
  *reinterpret_cast<v2d*>(part + (int64_t)blockIdx.y * n + c0) = (a0 + a1) + (a2 + a3);
}

__global__ void __launch_bounds__(256) colsum_finish_kernel(const double* __restrict__ part, int rs, int64_t n, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double a = 0.;
  for (int s = 0; s < rs; ++s) a += part[(int64_t)s * n + c];
  out[c] = a;
}

int colgemv_splits(int64_t m, int64_t n) {
  const int64_t bx = (n + 511) / 512;
  int64_t rs = (2048 + bx - 1) / bx;           // ~8 workgroups per CU in flight
  if (rs > 64) rs = 64;
  if (rs > m / 64) rs = m / 64 > 0 ? m / 64 : 1;
  return (int)(rs < 1 ? 1 : rs);
}

// ---- transposed lattice application: spectral planes  W[r][iz][ky][kx] = Lambda3[iz][ky][kx] * lhat[r][ky][kx] --------------------
// (lattice_gram.LatticeGram: rows of L^-1 restricted to an operator's columns, convolved with the operator's stencil table through
//  its (y, x) eigen-decomposition.)  One workgroup per (row, iz) plane of Py x Px doubles; lhat[r] (128 KB) and Lambda3 (8 MB) stay in
//  cache, the launch is bound by its HBM writes.  mode 1: the older layout W[r][kx][iz][ky] = lamW[kx][iz][ky] * lhat[r][ky][kx] that feeds
//  the two batched-GEMM inverse steps (grids the fused inverse transform has no instance for).
// Blocking: a workgroup keeps ZG z-planes of a 2048-double chunk of Lambda3 in registers and sweeps its share of the rows, so that
// the table is read 8 times per launch and lhat once per ZG planes (one workgroup per (row, iz) plane re-read the 8 MB table for
// every row out of L2 / MALL -- as many bytes as the launch writes: 3.5 TB/s; this form: the write-only ceiling).
constexpr int WP_ZG = 8, WP_CH = 1024;      // z-planes per workgroup; v2d elements per chunk (4 per thread)
__global__ void __launch_bounds__(256) lattice_wplanes_kernel(const double* __restrict__ lam3, const double* __restrict__ lhat, int64_t plane,
                                                              int nz, int64_t rows, double* __restrict__ W) {
  const int64_t nch = plane / 2 / WP_CH;
  const int64_t ch = blockIdx.x % nch;
  const int z0 = (int)(blockIdx.x / nch) * WP_ZG;
  const int nzg = nz - z0 < WP_ZG ? nz - z0 : WP_ZG;
  v2d lam[WP_ZG][4];
#pragma unroll
  for (int g = 0; g < WP_ZG; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      lam[g][k] = g < nzg ? reinterpret_cast<const v2d*>(lam3 + (int64_t)(z0 + g) * plane)[ch * WP_CH + k * 256 + threadIdx.x] : (v2d){0.0, 0.0};
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const v2d* lh = reinterpret_cast<const v2d*>(lhat + r * plane) + ch * WP_CH + threadIdx.x;
    v2d l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = lh[k * 256];
    v2d* w = reinterpret_cast<v2d*>(W + (r * nz + z0) * plane) + ch * WP_CH + threadIdx.x;
#pragma unroll
    for (int g = 0; g < WP_ZG; ++g) {
      if (g < nzg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[(int64_t)g * (plane / 2) + k * 256] = lam[g][k] * l[k];
      }
    }
  }
}

__global__ void __launch_bounds__(256) lattice_wbuild_kernel(const double* __restrict__ lamW, const double* __restrict__ lhat, int Py, int Px,
                                                             int nz, double* __restrict__ W) {
  const int kx = blockIdx.x, r = blockIdx.y;
  const double* lh = lhat + (int64_t)r * Py * Px + kx;
  const double* lw = lamW + (int64_t)kx * nz * Py;
  double* w = W + ((int64_t)r * Px + kx) * nz * Py;
  const int ky = threadIdx.x % Py, ph = threadIdx.x / Py, nph = 256 / Py;     // 256 threads = (256 / Py) iz phases x Py ky lanes
  if (ph >= nph) return;
  const double l = lh[(int64_t)ky * Px];
  for (int iz = ph; iz < nz; iz += nph) w[iz * Py + ky] = lw[iz * Py + ky] * l;
}

// 2-D strided precision conversion (rows x cols, cols even): the fp32-assembly mode keeps A K in fp32 in HBM and hands the fp64
// MFMA kernels fp64 panels.  Pure streaming: 12 B per element.
template <typename SRC, typename DST>
__global__ void __launch_bounds__(256) convert_kernel(const SRC* __restrict__ src, int64_t lds_, DST* __restrict__ dst, int64_t ldd,
                                                      int64_t rows, int64_t cols2) {
  typedef SRC s2 __attribute__((ext_vector_type(2)));
  typedef DST d2 __attribute__((ext_vector_type(2)));
  const int64_t total = rows * cols2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols2, c = (i - r * cols2) * 2;
    const s2 v = *reinterpret_cast<const s2*>(src + r * lds_ + c);
    *reinterpret_cast<d2*>(dst + r * ldd + c) = (d2){(DST)v[0], (DST)v[1]};
  }
}

// in-place round trip through fp32 (what storing K in fp32 does to it)
__global__ void __launch_bounds__(256) round_f32_kernel(double* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = (double)(float)x[i];
}

template <int ID>
__global__ void __launch_bounds__(256) k_eval_kernel(const double* __restrict__ d2, int64_t n, const CovParams p,
                                                     double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = p.scale * cov_eval<ID>(p, d2[i]);
}

// ---- forward operator ----------------------------------------------------------------------------------------
// Reference arithmetic is reproduced operation by operation with FMA contraction OFF: the padded planes
// (iy = 0 and iy = ny) carry potentials of ~1.5e7 whose 8-corner differences cancel catastrophically, so the
// rounding sequence matters (hard part 1).
__device__ __forceinline__ double grav_potential(double x, double y, double z) {
#pragma clang fp contract(off)
  const double r = sqrt((x * x + y * y) + z * z);
  return (x * log(y + r) + y * log(x + r)) - z * atan((x * y) / (z * r + 1e-9));  // sensormodel.py:107-109
}

__device__ __forceinline__ double magn_potential(double x, double y, double z, double bx, double by, double bz,
                                                 double inv_norm_b) {
#pragma clang fp contract(off)
  const double r = sqrt((x * x + y * y) + z * z);
  // sensormodel.py:130-133, evaluated left to right
  double f = (2. * by * bz) * log(x + r);
  f = f + (2. * bz * bx) * log(y + r);
  f = f + (2. * by * bx) * log(z + r);
  f = f + (bz * bz - by * by) * atan((x * z) / (y * r));
  f = f + (bz * bz - bx * bx) * atan((y * z) / (x * r));
  return -(inv_norm_b * f);
}

struct SensArgs {
  const double* loc; int64_t Ms; int nx, ny, nz, wx, iy0, iy1;
  const double *xe, *ye, *ze;
  double bx, by, bz, inv_norm_b, scale_mul, scale_div;
  double* A; int64_t ld;
};

template <int FUNC>
__global__ void __launch_bounds__(512) a_sens_kernel(const SensArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double planes[];  // [2][(wx+1)*(nz+1)]
  const int n = blockIdx.x;
  const int jx0 = blockIdx.y * a.wx;
  const int wx = (a.nx - jx0) < a.wx ? (a.nx - jx0) : a.wx;  // voxels of this x-range
  const int nzp = a.nz + 1;
  const int pn = (wx + 1) * nzp;
  const double sx = a.loc[3 * (int64_t)n + 0], sy = a.loc[3 * (int64_t)n + 1], sz = a.loc[3 * (int64_t)n + 2];
  const double far = 1e6;
  double* row_out = a.A + (int64_t)n * a.ld;
  for (int i = a.iy0; i <= a.iy1; ++i) {   // node planes iy0..iy1 -> voxel slabs iy0..iy1-1
    double* cur = planes + (i & 1) * pn;
    const double* prev = planes + ((i & 1) ^ 1) * pn;
    double y0 = a.ye[i] - sy;
    if (i == 0) y0 -= far;
    if (i == a.ny) y0 += far;
    for (int t = threadIdx.x; t < pn; t += blockDim.x) {
      const int j = t / nzp, k = t - j * nzp;
      double x0 = a.xe[jx0 + j] - sx;
      if (i == 0) x0 -= far;      // sensormodel.py:63-68: the padding acts on axis 0 (= iy) of BOTH x0 and y0
      if (i == a.ny) x0 += far;
      const double z0 = a.ze[k] - sz;
      cur[t] = (FUNC == GEOBO_F_GRAV) ? grav_potential(x0, y0, z0)
                                      : magn_potential(x0, y0, z0, a.bx, a.by, a.bz, a.inv_norm_b);
    }
    __syncthreads();
    if (i > a.iy0) {
      const int iy = i - 1;
      const int nv = wx * a.nz;
      for (int t = threadIdx.x; t < nv; t += blockDim.x) {
        const int j = t / a.nz, k = t - j * a.nz;
        const double* hi = cur + j * nzp + k;    // e[i+1][j..][k..]   (i+1 = current plane)
        const double* lo = prev + j * nzp + k;   // e[i][..]
        const double h = ((hi[nzp + 1] - hi[nzp]) - hi[1]) + hi[0];   // sensormodel.py:85-86, left to right
        const double l = ((lo[nzp + 1] - lo[nzp]) - lo[1]) + lo[0];
        const double s = -(h - l);
        row_out[((int64_t)iy * a.nx + (jx0 + j)) * a.nz + k] = (a.scale_mul * s) / a.scale_div;
      }
    }
    __syncthreads();
  }
}

// ---- lattice form of A_sens ---------------------------------------------------------------------------------------------
// Sensors on a lattice commensurate with the voxel columns (the survey resampled to the cube's x-y grid, the reference's own
// workflow): the node offsets xe[j] - sx depend on j - jx_s only, so away from the +-1e6-padded planes iy = 0 and ny the
// 8-corner stencil is translation invariant.  P: potentials on the offset lattice (~(2ny)(2nx)(nz+1) evaluations instead of
// Ms (ny+1)(nx+1)(nz+1)); Q: the stencil of P, i.e. the operator entry for voxel-offset (ddy, ddx, k); every (sensor, iy) slab
// of A is then one contiguous copy out of Q.  Same expressions and evaluation order as a_sens_kernel (the caller verifies
// that the offsets are bit-identical for every pair), so the result is identical.
template <int FUNC>
__global__ void __launch_bounds__(256) lattice_potential_kernel(const double* __restrict__ dxv, int ndx, const double* __restrict__ dyv,
                                                                int ndy, const double* __restrict__ dzv, int ndz, double bx, double by,
                                                                double bz, double inv_norm_b, double* __restrict__ P) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)ndy * ndx * ndz;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int k = (int)(t % ndz);
    const int64_t r = t / ndz;
    const int j = (int)(r % ndx), i = (int)(r / ndx);
    P[t] = (FUNC == GEOBO_F_GRAV) ? grav_potential(dxv[j], dyv[i], dzv[k]) : magn_potential(dxv[j], dyv[i], dzv[k], bx, by, bz, inv_norm_b);
  }
}

__global__ void __launch_bounds__(256) lattice_stencil_kernel(const double* __restrict__ P, int ndx, int ndz, int nqy, int nqx, int nz,
                                                              double scale_mul, double scale_div, double* __restrict__ Q) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)nqy * nqx * nz;
  const int64_t plane = (int64_t)ndx * ndz;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int k = (int)(t % nz);
    const int64_t r = t / nz;
    const int qj = (int)(r % nqx), qi = (int)(r / nqx);
    const double* lo = P + qi * plane + (int64_t)qj * ndz + k;   // plane iy (node offsets ddy, ddx)
    const double* hi = lo + plane;                              // plane iy + 1
    const double h = ((hi[ndz + 1] - hi[ndz]) - hi[1]) + hi[0];  // sensormodel.py:85-86, left to right
    const double l = ((lo[ndz + 1] - lo[ndz]) - lo[1]) + lo[0];
    const double sres = -(h - l);
    Q[t] = (scale_mul * sres) / scale_div;
  }
}

__global__ void __launch_bounds__(256) lattice_gather_kernel(const double* __restrict__ Q, int nqx, int nx, int ny, int nz,
                                                             const int* __restrict__ jxs, const int* __restrict__ jys, int ia,
                                                             double* __restrict__ A, int64_t ld) {
  const int64_t s = blockIdx.x;
  const int iy = ia + blockIdx.y;
  const int64_t slab = (int64_t)nx * nz;
  const double* src = Q + ((int64_t)(iy - jys[s] + ny - 2) * nqx + (nx - 1 - jxs[s])) * nz;   // contiguous nx*nz doubles
  double* dst = A + s * ld + (int64_t)iy * slab;
  for (int64_t t = threadIdx.x; t < slab / 2; t += 256) reinterpret_cast<v2d*>(dst)[t] = reinterpret_cast<const v2d*>(src)[t];
}

__global__ void __launch_bounds__(256) scale_broadcast_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                              int64_t n2, int64_t nb2, double* __restrict__ out) {
  // 16-byte accesses; nb (the broadcast period) is even, so a pair never straddles the period
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const v2d x = reinterpret_cast<const v2d*>(a)[i];
    const v2d y = reinterpret_cast<const v2d*>(b)[i % nb2];
    reinterpret_cast<v2d*>(out)[i] = x * y;
  }
}

__global__ void __launch_bounds__(256) scale_broadcast2_kernel(const double* __restrict__ a, const double* __restrict__ b0,
                                                               const double* __restrict__ b1, int64_t n2, int64_t nb2,
                                                               double* __restrict__ out0, double* __restrict__ out1) {
  // one read of the spectrum, two eigenvalue tables, two outputs (both property blocks of one operator)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const v2d x = reinterpret_cast<const v2d*>(a)[i];
    const int64_t j = i % nb2;
    reinterpret_cast<v2d*>(out0)[i] = x * reinterpret_cast<const v2d*>(b0)[j];
    reinterpret_cast<v2d*>(out1)[i] = x * reinterpret_cast<const v2d*>(b1)[j];
  }
}

template <int FUNC>
__global__ void __launch_bounds__(256) potential_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                        const double* __restrict__ z, int64_t n, double bx, double by,
                                                        double bz, double inv_norm_b, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (FUNC == GEOBO_F_GRAV) ? grav_potential(x[i], y[i], z[i])
                                    : magn_potential(x[i], y[i], z[i], bx, by, bz, inv_norm_b);
}

// 8 independent accumulators (64 VGPRs) so that up to 4 wavefronts fit per SIMD; iters x 16 MFMAs per wave
__global__ void __launch_bounds__(256, 4) mfma_peak_kernel(int iters, double* __restrict__ out) {
  v4d acc[8];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0., 0., 0., 0.};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 7], 0, 0, 0);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// co-issue probe: NV independent VALU ops of a given flavour after every MFMA (same wave)
template <int MODE, int NV>
__global__ void __launch_bounds__(256, 4) mfma_mix_kernel(int iters, double* __restrict__ out) {
  v4d acc[8];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  double xd[8];
  float xf[8];
  unsigned xi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = (v4d){0., 0., 0., 0.}; xd[i] = a + i; xf[i] = (float)(a + i); xi[i] = threadIdx.x + i; }
  const double cd = 0.999999, dd = 1e-7;
  const float cf = 0.999999f, df = 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (MODE == 1) xd[v & 7] = __builtin_fma(xd[v & 7], cd, dd);
        if constexpr (MODE == 2) xf[v & 7] = __builtin_fmaf(xf[v & 7], cf, df);
        if constexpr (MODE == 3) xi[v & 7] = xi[v & 7] * 1664525u + 1013904223u;
      }
    }
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + xd[i] + xf[i] + xi[i];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

extern "C" int geobo_mfma_mix(int mode, int nv, int blocks, int iters, double* out, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
#define MIX(M, N) if (mode == M && nv == N) { hipLaunchKernelGGL((mfma_mix_kernel<M, N>), dim3(blocks), dim3(256), 0, st, iters, out); return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH; }
  MIX(0, 0) MIX(1, 2) MIX(1, 4) MIX(1, 8) MIX(1, 16) MIX(2, 4) MIX(2, 8) MIX(2, 16) MIX(2, 32) MIX(3, 8) MIX(3, 16) MIX(3, 32)
#undef MIX
  return GEOBO_E_UNSUPPORTED;
}

extern "C" int geobo_version(void) { return GEOBO_VERSION; }
extern "C" int64_t geobo_pad_m(int64_t m) { return (m + GEOBO_PAD_M - 1) / GEOBO_PAD_M * GEOBO_PAD_M; }
extern "C" int64_t geobo_pad_n(int64_t n) { return (n + GEOBO_PAD_N - 1) / GEOBO_PAD_N * GEOBO_PAD_N; }

extern "C" int geobo_k_block(int kernel_id, const double* rx, const double* ry, const double* rz, int64_t nr,
                             const double* cx, const double* cy, const double* cz, int64_t nc, double l1, double l2,
                             double w, double amp, double* out, int64_t ld, void* stream) {
  if (!rx || !ry || !rz || !cx || !cy || !cz || !out) return GEOBO_E_ARG;
  if (nr <= 0 || nc <= 0) return GEOBO_OK;
  if (ld < nc) return GEOBO_E_ARG;
  const CovParams p = make_cov(kernel_id, l1, l2, w, amp);
  const dim3 grid((unsigned)((nc + 511) / 512), (unsigned)((nr + KB_ROWS - 1) / KB_ROWS));
  hipStream_t st = (hipStream_t)stream;
#define GEOBO_KB(ID) hipLaunchKernelGGL((k_block_kernel<ID, double>), grid, dim3(256), 0, st, rx, ry, rz, nr, cx, cy, cz, nc, p, out, ld)
  COV_DISPATCH(kernel_id, GEOBO_KB);
#undef GEOBO_KB
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_k_block_f32(int kernel_id, const double* rx, const double* ry, const double* rz, int64_t nr,
                                 const double* cx, const double* cy, const double* cz, int64_t nc, double l1, double l2,
                                 double w, double amp, float* out, int64_t ld, void* stream) {
  if (!rx || !ry || !rz || !cx || !cy || !cz || !out) return GEOBO_E_ARG;
  if (nr <= 0 || nc <= 0) return GEOBO_OK;
  if (ld < nc) return GEOBO_E_ARG;
  const CovParams p = make_cov(kernel_id, l1, l2, w, amp);
  const dim3 grid((unsigned)((nc + 511) / 512), (unsigned)((nr + KB_ROWS - 1) / KB_ROWS));
  hipStream_t st = (hipStream_t)stream;
#define GEOBO_KB(ID) hipLaunchKernelGGL((k_block_kernel<ID, float>), grid, dim3(256), 0, st, rx, ry, rz, nr, cx, cy, cz, nc, p, out, ld)
  COV_DISPATCH(kernel_id, GEOBO_KB);
#undef GEOBO_KB
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_k_block_grid(int nx, int ny, int nz, const double* table, const int64_t* rows, int64_t row0, int64_t nr,
                                  int64_t col0, int64_t ncols, int out_f32, void* out, int64_t ld, void* stream) {
  if (!table || !out) return GEOBO_E_ARG;
  if (nr <= 0 || ncols <= 0) return GEOBO_OK;
  const int64_t N = (int64_t)nx * ny * nz;
  if (nx <= 0 || ny <= 0 || nz <= 0 || (nz & 1) || (col0 & 1) || col0 < 0 || col0 + ncols > N || ld < ncols) return GEOBO_E_ARG;
  if (!rows && (row0 < 0 || row0 + nr > N)) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const unsigned gy = (unsigned)((nr + KB_ROWS - 1) / KB_ROWS);
  if (out_f32 && nz % 4 == 0 && col0 % 4 == 0)
    hipLaunchKernelGGL((k_block_grid_kernel<float, 4>), dim3((unsigned)((ncols + 1023) / 1024), gy), dim3(256), 0, st, table, nx, ny, nz, rows,
                       row0, nr, col0, ncols, (float*)out, ld);
  else if (out_f32)
    hipLaunchKernelGGL((k_block_grid_kernel<float, 2>), dim3((unsigned)((ncols + 511) / 512), gy), dim3(256), 0, st, table, nx, ny, nz, rows,
                       row0, nr, col0, ncols, (float*)out, ld);
  else
    hipLaunchKernelGGL((k_block_grid_kernel<double, 2>), dim3((unsigned)((ncols + 511) / 512), gy), dim3(256), 0, st, table, nx, ny, nz, rows,
                       row0, nr, col0, ncols, (double*)out, ld);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" size_t geobo_colgemv_ws_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  return (size_t)colgemv_splits(m, n) * (size_t)n * sizeof(double);
}

extern "C" int geobo_colgemv(int64_t m, int64_t n, const double* X, int64_t ld, const double* v, double* out, void* ws,
                             size_t ws_bytes, void* stream) {
  if (!X || !v || !out || !ws) return GEOBO_E_ARG;
  if (m <= 0 || n <= 0) return GEOBO_OK;
  if ((n & 1) || (ld & 1) || ld < n || ((uintptr_t)X & 15) || ((uintptr_t)ws & 15)) return GEOBO_E_ALIGN;
  if (ws_bytes < geobo_colgemv_ws_bytes(m, n)) return GEOBO_E_ARG;
  const int rs = colgemv_splits(m, n);
  const int64_t rows_per = (m + rs - 1) / rs;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colgemv_kernel, dim3((unsigned)((n + 511) / 512), (unsigned)rs), dim3(256), 0, st, X, ld, m, n, v, rows_per, (double*)ws);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)ws, rs, n, out);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_lattice_wbuild(int64_t rows, int Py, int Px, int nz, const double* lamW, const double* lhat, double* W,
                                    void* stream) {
  if (!lamW || !lhat || !W) return GEOBO_E_ARG;
  if (rows <= 0) return GEOBO_OK;
  if (Py <= 0 || Px <= 0 || nz <= 0 || Py > 256 || rows > 65535) return GEOBO_E_UNSUPPORTED;
  hipLaunchKernelGGL(lattice_wbuild_kernel, dim3((unsigned)Px, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, lamW, lhat, Py, Px, nz, W);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_lattice_wplanes(int64_t rows, int Py, int Px, int nz, const double* lam3, const double* lhat, double* W,
                                     void* stream) {
  if (!lam3 || !lhat || !W) return GEOBO_E_ARG;
  if (rows <= 0) return GEOBO_OK;
  if (Py <= 0 || Px <= 0 || nz <= 0 || ((Py * Px) & 1) || rows > 65535 || ((uintptr_t)lam3 & 15) || ((uintptr_t)lhat & 15) ||
      ((uintptr_t)W & 15))
    return GEOBO_E_ALIGN;
  const int64_t plane = (int64_t)Py * Px;
  if (plane % (2 * WP_CH)) return GEOBO_E_UNSUPPORTED;
  const unsigned gx = (unsigned)(plane / 2 / WP_CH * ((nz + WP_ZG - 1) / WP_ZG));
  const unsigned gy = (unsigned)(rows < 32 ? rows : 32);                     // 64 x 32 workgroups at 64^3: 8 rows each for a 256-row batch
  hipLaunchKernelGGL(lattice_wplanes_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, lam3, lhat, plane, nz, rows, W);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_convert(int to_f32, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                             void* stream) {
  if (!src || !dst) return GEOBO_E_ARG;
  if (rows <= 0 || cols <= 0) return GEOBO_OK;
  if ((cols & 1) || (ld_src & 1) || (ld_dst & 1) || ld_src < cols || ld_dst < cols) return GEOBO_E_ALIGN;
  int64_t nb = (rows * (cols / 2) + 255) / 256;
  if (nb > 256 * 64) nb = 256 * 64;
  hipStream_t st = (hipStream_t)stream;
  if (to_f32)
    hipLaunchKernelGGL((convert_kernel<double, float>), dim3((unsigned)nb), dim3(256), 0, st, (const double*)src, ld_src, (float*)dst,
                       ld_dst, rows, cols / 2);
  else
    hipLaunchKernelGGL((convert_kernel<float, double>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)src, ld_src, (double*)dst,
                       ld_dst, rows, cols / 2);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_round_f32(double* x, int64_t n, void* stream) {
  if (!x) return GEOBO_E_ARG;
  if (n <= 0) return GEOBO_OK;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(round_f32_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_cov_table(int kernel_id, int nx, int ny, int nz, double sx, double sy, double sz, double l1, double l2,
                               double w, double amp, double* table, void* stream) {
  if (!table || nx <= 0 || ny <= 0 || nz <= 0) return GEOBO_E_ARG;
  const CovParams p = make_cov(kernel_id, l1, l2, w, amp);
  const int64_t n = (int64_t)nx * ny * nz * 2;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipStream_t st = (hipStream_t)stream;
#define GEOBO_CT(ID) hipLaunchKernelGGL(cov_table_kernel<ID>, dim3((unsigned)nb), dim3(256), 0, st, nx, ny, nz, sx, sy, sz, p, table)
  COV_DISPATCH(kernel_id, GEOBO_CT);
#undef GEOBO_CT
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_k_eval(int kernel_id, const double* d2, int64_t n, double l1, double l2, double w, double amp,
                            double* out, void* stream) {
  if (!d2 || !out) return GEOBO_E_ARG;
  if (n <= 0) return GEOBO_OK;
  const CovParams p = make_cov(kernel_id, l1, l2, w, amp);
  int64_t nb = (n + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
#define GEOBO_KE(ID) hipLaunchKernelGGL(k_eval_kernel<ID>, dim3((unsigned)nb), dim3(256), 0, st, d2, n, p, out)
  COV_DISPATCH(kernel_id, GEOBO_KE);
#undef GEOBO_KE
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_a_sens_slab(int func_id, const double* B3_host, const double* loc, int64_t Ms, int nx, int ny, int nz,
                                 const double* xe, const double* ye, const double* ze, double scale_mul, double scale_div,
                                 int iy0, int iy1, double* A, int64_t ld, int64_t col_origin, void* stream);

// A[n, 0] stands for voxel column col_origin; the requested slab's columns [iy0*nx*nz, iy1*nx*nz) must lie inside one row of the
// buffer, [col_origin, col_origin + ld): nothing outside the caller's rows is ever addressed
static bool slab_origin_ok(int nx, int ny, int nz, int iy0, int iy1, int64_t ld, int64_t col_origin) {
  const int64_t plane = (int64_t)nx * nz;
  (void)ny;
  return col_origin >= 0 && col_origin <= plane * iy0 && plane * iy1 - col_origin <= ld;
}

extern "C" size_t geobo_a_sens_lattice_ws_bytes(int nx, int ny, int nz) {
  if (nx <= 0 || ny < 3 || nz <= 0) return 0;
  const size_t p = (size_t)(2 * ny - 2) * (2 * nx) * (nz + 1), q = (size_t)(2 * ny - 3) * (2 * nx - 1) * nz;
  return (p + q) * sizeof(double);
}

extern "C" int geobo_a_sens_lattice(int func_id, const double* B3_host, int64_t Ms, int nx, int ny, int nz, const double* dxv,
                                    const double* dyv, const double* dzv, const int* jxs, const int* jys, double scale_mul,
                                    double scale_div, int iy0, int iy1, double* A, int64_t ld, int64_t col_origin, void* ws,
                                    size_t ws_bytes, void* stream) {
  if (!B3_host || !dxv || !dyv || !dzv || !jxs || !jys || !A || !ws) return GEOBO_E_ARG;
  if (iy0 < 0 || iy1 > ny || iy0 >= iy1) return GEOBO_E_ARG;
  if (Ms <= 0 || nx <= 0 || ny < 3 || nz <= 0 || (nz & 1) || (ld & 1) || !slab_origin_ok(nx, ny, nz, iy0, iy1, ld, col_origin)) return GEOBO_E_ARG;
  A -= col_origin;                    // the kernels address columns absolutely and touch the requested slab only
  if (func_id != GEOBO_F_GRAV && func_id != GEOBO_F_MAGN) return GEOBO_E_UNSUPPORTED;
  if (ws_bytes < geobo_a_sens_lattice_ws_bytes(nx, ny, nz)) return GEOBO_E_ARG;
  const int ia = iy0 > 1 ? iy0 : 1, ib = iy1 < ny - 1 ? iy1 : ny - 1;   // interior slabs of the request
  if (ia >= ib) return GEOBO_OK;
  const int ndx = 2 * nx, ndy = 2 * ny - 2, ndz = nz + 1, nqy = 2 * ny - 3, nqx = 2 * nx - 1;
  double* P = (double*)ws;
  double* Q = P + (size_t)ndy * ndx * ndz;
  const double bx = B3_host[0], by = B3_host[1], bz = B3_host[2];
  const double inb = 1. / sqrt(bx * bx + by * by + bz * bz);
  hipStream_t st = (hipStream_t)stream;
  const int64_t np = (int64_t)ndy * ndx * ndz, nq = (int64_t)nqy * nqx * nz;
  const unsigned gp = (unsigned)((np + 255) / 256 < 65536 ? (np + 255) / 256 : 65536);
  const unsigned gq = (unsigned)((nq + 255) / 256 < 65536 ? (nq + 255) / 256 : 65536);
  if (func_id == GEOBO_F_GRAV)
    hipLaunchKernelGGL(lattice_potential_kernel<GEOBO_F_GRAV>, dim3(gp), dim3(256), 0, st, dxv, ndx, dyv, ndy, dzv, ndz, bx, by, bz, inb, P);
  else
    hipLaunchKernelGGL(lattice_potential_kernel<GEOBO_F_MAGN>, dim3(gp), dim3(256), 0, st, dxv, ndx, dyv, ndy, dzv, ndz, bx, by, bz, inb, P);
  hipLaunchKernelGGL(lattice_stencil_kernel, dim3(gq), dim3(256), 0, st, (const double*)P, ndx, ndz, nqy, nqx, nz, scale_mul, scale_div, Q);
  hipLaunchKernelGGL(lattice_gather_kernel, dim3((unsigned)Ms, (unsigned)(ib - ia)), dim3(256), 0, st, (const double*)Q, nqx, nx, ny, nz,
                     jxs, jys, ia, A, ld);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_a_sens(int func_id, const double* B3_host, const double* loc, int64_t Ms, int nx, int ny, int nz,
                            const double* xe, const double* ye, const double* ze, double scale_mul, double scale_div,
                            double* A, int64_t ld, void* stream) {
  return geobo_a_sens_slab(func_id, B3_host, loc, Ms, nx, ny, nz, xe, ye, ze, scale_mul, scale_div, 0, ny, A, ld, 0, stream);
}

extern "C" int geobo_a_sens_slab(int func_id, const double* B3_host, const double* loc, int64_t Ms, int nx, int ny, int nz,
                                 const double* xe, const double* ye, const double* ze, double scale_mul, double scale_div,
                                 int iy0, int iy1, double* A, int64_t ld, int64_t col_origin, void* stream) {
  if (!B3_host || !loc || !xe || !ye || !ze || !A) return GEOBO_E_ARG;
  if (iy0 < 0 || iy1 > ny || iy0 >= iy1) return GEOBO_E_ARG;
  if (Ms <= 0 || nx <= 0 || ny <= 0 || nz <= 0 || !slab_origin_ok(nx, ny, nz, iy0, iy1, ld, col_origin)) return GEOBO_E_ARG;
  A -= col_origin;
  if (func_id != GEOBO_F_GRAV && func_id != GEOBO_F_MAGN) return GEOBO_E_UNSUPPORTED;
  SensArgs a;
  a.loc = loc; a.Ms = Ms; a.nx = nx; a.ny = ny; a.nz = nz;
  a.xe = xe; a.ye = ye; a.ze = ze;
  a.bx = B3_host[0]; a.by = B3_host[1]; a.bz = B3_host[2];
  a.inv_norm_b = 1. / sqrt(a.bx * a.bx + a.by * a.by + a.bz * a.bz);  // sensormodel.py:129 (inf for B = 0: grav ignores it)
  a.scale_mul = scale_mul; a.scale_div = scale_div; a.A = A; a.ld = ld; a.iy0 = iy0; a.iy1 = iy1;
  // two node planes of (wx+1)*(nz+1) doubles must fit the LDS budget
  const int budget = 128 * 1024;
  int wx = budget / (16 * (nz + 1)) - 1;
  if (wx < 1) return GEOBO_E_UNSUPPORTED;
  if (wx > nx) wx = nx;
  a.wx = wx;
  const size_t lds = (size_t)2 * (wx + 1) * (nz + 1) * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)Ms, (unsigned)((nx + wx - 1) / wx));
  if (func_id == GEOBO_F_GRAV) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(a_sens_kernel<GEOBO_F_GRAV>), hipFuncAttributeMaxDynamicSharedMemorySize, budget) != hipSuccess) return GEOBO_E_LAUNCH;
    hipLaunchKernelGGL(a_sens_kernel<GEOBO_F_GRAV>, grid, dim3(512), lds, st, a);
  } else {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(a_sens_kernel<GEOBO_F_MAGN>), hipFuncAttributeMaxDynamicSharedMemorySize, budget) != hipSuccess) return GEOBO_E_LAUNCH;
    hipLaunchKernelGGL(a_sens_kernel<GEOBO_F_MAGN>, grid, dim3(512), lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_potential(int func_id, const double* B3_host, const double* x, const double* y, const double* z,
                               int64_t n, double* out, void* stream) {
  if (!B3_host || !x || !y || !z || !out) return GEOBO_E_ARG;
  if (n <= 0) return GEOBO_OK;
  const double bx = B3_host[0], by = B3_host[1], bz = B3_host[2];
  const double inb = 1. / sqrt(bx * bx + by * by + bz * bz);
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (func_id == GEOBO_F_GRAV)
    hipLaunchKernelGGL(potential_kernel<GEOBO_F_GRAV>, dim3((unsigned)nb), dim3(256), 0, st, x, y, z, n, bx, by, bz, inb, out);
  else if (func_id == GEOBO_F_MAGN)
    hipLaunchKernelGGL(potential_kernel<GEOBO_F_MAGN>, dim3((unsigned)nb), dim3(256), 0, st, x, y, z, n, bx, by, bz, inb, out);
  else
    return GEOBO_E_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_scale_broadcast(const double* a, const double* b, int64_t n, int64_t nb, double* out, void* stream) {
  if (!a || !b || !out) return GEOBO_E_ARG;
  if (n <= 0) return GEOBO_OK;
  if ((n & 1) || (nb & 1) || nb <= 0 || n % nb) return GEOBO_E_ALIGN;
  int64_t nblk = (n / 2 + 255) / 256;
  if (nblk > 256 * 32) nblk = 256 * 32;
  hipLaunchKernelGGL(scale_broadcast_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a, b, n / 2, nb / 2, out);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_scale_broadcast2(const double* a, const double* b0, const double* b1, int64_t n, int64_t nb, double* out0,
                                      double* out1, void* stream) {
  if (!a || !b0 || !b1 || !out0 || !out1) return GEOBO_E_ARG;
  if (n <= 0) return GEOBO_OK;
  if ((n & 1) || (nb & 1) || nb <= 0 || n % nb) return GEOBO_E_ALIGN;
  int64_t nblk = (n / 2 + 255) / 256;
  if (nblk > 256 * 32) nblk = 256 * 32;
  hipLaunchKernelGGL(scale_broadcast2_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a, b0, b1, n / 2, nb / 2,
                     out0, out1);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_mfma_f64_peak(int blocks, int iters, double* out, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return GEOBO_E_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
