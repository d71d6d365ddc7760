/* geobo_hip.h -- C ABI of libgeobo_hip.so: the MI355X (gfx950) implementation of GeoBO's GP
 * joint-inversion hot path.
 *
 * The reference (sebhaan/geobo) has no FFI: its boundary is the Python API of geobo/kernels.py,
 * geobo/sensormodel.py and geobo/inversion.py.  The package `geobo_amd` keeps that Python surface
 * and binds the entry points below with ctypes (geobo_amd/_lib.py); INTEGRATION.md shows the same
 * stub applied to the reference itself.  Each entry point names the reference call site it replaces
 * (file:line relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (torch.float64 / torch.int32 CUDA tensors), row-major,
 *     leading dimensions in elements; `stream` is a hipStream_t passed as void*;
 *   - functions are stream-ordered, never allocate, never synchronise, never throw; the return value
 *     is 0 on success or a negative GEOBO_E_* code (argument validation / launch failure);
 *   - process-global state: ONE exception to "none" -- kernels that need more than 64 KiB of dynamic LDS record, in a
 *     per-kernel atomic bit mask indexed by device (`attr_done` in gemm_f64.hip, xz2d.hip, xz2d_fold.hip), that
 *     hipFuncSetAttribute(MaxDynamicSharedMemorySize) has been issued for that device.  The flag is idempotent, lock free,
 *     never cleared and carries no data: concurrent first calls at worst issue the attribute twice;
 *   - PADDING CONTRACT: matrix operands are allocated with their dimensions rounded up to
 *     GEOBO_PAD_M (rows of M-like dims) / GEOBO_PAD_N (voxel-like dims) and the padding is ZERO
 *     (identity on the diagonal of matrices that get factorised).  Kernels then run without edge
 *     predication.  `geobo_pad_m/n` give the rounded sizes;
 *   - voxel order is the reference's: flat index p = (iy*nx + ix)*nz + iz (kernels.py:40-42);
 *   - property blocks: 0 density, 1 magnetic susceptibility, 2 drill property; block (i,j) of the
 *     3x3 prior is k_auto(l_i) if i==j else w_ij * k_cross(l_i, l_j) (kernels.py:183-195).
 */
#ifndef GEOBO_HIP_H
#define GEOBO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOBO_VERSION 212 /* 201: flag word of geobo_gemm_nt, (64, 32) instance of geobo_xz2d, workspace layout of geobo_potrf_inv; 202: geobo_ymul, geobo_xz2d_fold_lattice; 203: geobo_sumsq_accum, geobo_lamdot_z, geobo_toeplitz_y2t, geobo_xz2d_fold_quad; 204: geobo_toeplitz_y3_add; 205: workspace layout of geobo_potrf_inv (one T buffer per tree node); 206: geobo_potrf_inv as one persistent tile-DAG launch from m = 1024 (workspace: + counters); 207: geobo_gemm_fold; 208: geobo_gemm_fold_lamdot; 209: geobo_toeplitz_y2s; 210: geobo_xz2d_fold* dense planes take the quarter-period group order of the basis (radix 4); 211: geobo_ymul_fold; 212: geobo_spectral_y, geobo_spectral_y2s, geobo_spectral_y_basis, geobo_spectral_y3, geobo_spectral_y3t, geobo_spectral_axis, geobo_rowgemv */

#define GEOBO_PAD_M 256 /* row padding of M-like dimensions (observation rows)            */
#define GEOBO_PAD_N 128 /* padding of voxel-like dimensions (columns / contraction index) */

enum { GEOBO_OK = 0, GEOBO_E_ARG = -1, GEOBO_E_ALIGN = -2, GEOBO_E_LAUNCH = -3, GEOBO_E_UNSUPPORTED = -4 };

/* covariance families (kernels.py): the *_X members are the cross-covariance forms ("2" suffix) */
enum {
  GEOBO_K_D2 = 0,        /* squared distance itself            kernels.py:45-61   */
  GEOBO_K_EXP = 1,       /* gpkernel                           kernels.py:81-88   */
  GEOBO_K_EXP_X = 2,     /* gpkernel2                          kernels.py:90-99   */
  GEOBO_K_MATERN32 = 3,  /* gpkernel_matern32                  kernels.py:140-146 */
  GEOBO_K_MATERN32_X = 4,/* gpkernel_matern32_2                kernels.py:148-156 */
  GEOBO_K_SPARSE = 5,    /* gpkernel_sparse                    kernels.py:101-114 */
  GEOBO_K_SPARSE_X = 6   /* gpkernel_sparse2                   kernels.py:116-138 */
};

enum { GEOBO_F_GRAV = 0, GEOBO_F_MAGN = 1 }; /* sensormodel.py:96-110 / :113-133 */

int geobo_version(void);
int64_t geobo_pad_m(int64_t m);
int64_t geobo_pad_n(int64_t n);

/* out[r, c] = w * amp * k(|rows_r - cols_c|^2; l1, l2)            (replaces kernels.py:45-61 + :81-156 per block;
 * one call per block of create_cov, kernels.py:183-195).  Coordinates are SoA: x[], y[], z[].
 * nr, nc arbitrary (this kernel is predicated); ld >= nc.  For *_X families l1,l2 are the two lengths;
 * for auto families l2 is ignored.  The sparse equal-length offset l2 += 1e-3*l2 (kernels.py:125-126)
 * is applied inside. */
int geobo_k_block(int kernel_id, const double* rx, const double* ry, const double* rz, int64_t nr,
                  const double* cx, const double* cy, const double* cz, int64_t nc,
                  double l1, double l2, double w, double amp, double* out, int64_t ld, void* stream);

/* The same block written in fp32 ("fp32 kernel assembly" of BASELINE config 5: 128^3 x 3 properties does not fit fp64 storage):
 * covariance evaluated in fp64, rounded once on the store; 4 B written per element. */
int geobo_k_block_f32(int kernel_id, const double* rx, const double* ry, const double* rz, int64_t nr,
                      const double* cx, const double* cy, const double* cz, int64_t nc,
                      double l1, double l2, double w, double amp, float* out, int64_t ld, void* stream);

/* The same block on the REGULAR GRID of calcGridPoints3D (kernels.py:27-42) as a gather from the difference-lattice table that
 * geobo_cov_table builds once per block pair (N*2 doubles, cache resident): out[r, c] = table[(|diy| nx + |dix|) 2nz + (dz + nz-1)]
 * for the row voxel rows[r] (device int64 array; NULL = row0 + r) and the column voxel col0 + c, voxel p = (iy nx + ix) nz + iz.
 * No exp / sqrt per element: the launch is bound by the HBM store of the block (SURVEY.md 8(d) regime (i), "materialised kernel
 * assembly"); geobo_k_block stays the generator for irregular point sets.  nz even, col0 even, col0 + ncols <= nx ny nz,
 * ld >= ncols; out_f32 != 0: out is float* (config 5's fp32 assembly; the table is then the fp32-rounded one). */
int geobo_k_block_grid(int nx, int ny, int nz, const double* table, const int64_t* rows, int64_t row0, int64_t nr, int64_t col0,
                       int64_t ncols, int out_f32, void* out, int64_t ld, void* stream);

/* Weighted column sums  out[c] = sum_(r < m) X[r, c] * v[r]  of a row-major matrix (n even, ld even, X 16-byte aligned): the two
 * streams of the posterior MEAN in its transposed form (inversion.py:114-116, mu = V^T u with V = L^-1 (A K)):
 *     w = Linv^T u  (m = n = M),   mu = (A K)^T w  (m = M, n = the voxel-property columns)  --  V is not needed for the mean.
 * ws: geobo_colgemv_ws_bytes(m, n) (row-slice partials, summed in a fixed order: deterministic). */
size_t geobo_colgemv_ws_bytes(int64_t m, int64_t n);
int geobo_colgemv(int64_t m, int64_t n, const double* X, int64_t ld, const double* v, double* out, void* ws, size_t ws_bytes,
                  void* stream);

/* Step of the TRANSPOSED lattice application (rows of L^-1 restricted to one operator's columns -> rows of L^-1 A on a lattice
 * survey, geobo_amd/lattice_gram.py::apply_transpose):  W[r][kx][iz][ky] = lamW[kx][iz][ky] * lhat[r][ky][kx]  for r < rows, with
 * lhat the (y, x) real-DFT of the row's sensor image (Py x Px) and lamW the eigen-data of the operator's stencil table; the two
 * inverse transforms that follow are geobo_gemm_batched launches.  Py <= 256, rows <= 65535. */
int geobo_lattice_wbuild(int64_t rows, int Py, int Px, int nz, const double* lamW, const double* lhat, double* W, void* stream);
/* The same products laid out as spectral PLANES, W[r][iz][ky][kx] = lam3[iz][ky][kx] * lhat[r][ky][kx]: every (r, iz) plane then
 * goes through the fused inverse two-axis transform (geobo_xz2d_fold_inv_strided) straight into the row of L^-1 A.
 * Py * Px a multiple of 2048 (GEOBO_E_UNSUPPORTED otherwise), rows <= 65535; bound by its HBM writes. */
int geobo_lattice_wplanes(int64_t rows, int Py, int Px, int nz, const double* lam3, const double* lhat, double* W, void* stream);

/* fp32-assembly mode (config 5): A K lives in HBM as fp32 and the fp64 MFMA kernels are handed fp64 panels.
 *   geobo_convert: 2-D strided precision conversion, to_f32 = 1: dst(float)[r*ld_dst + c] = (float)src(double)[r*ld_src + c],
 *                  to_f32 = 0: float -> double; cols and both leading dimensions even, 8-byte aligned bases;
 *   geobo_round_f32: x[i] = (double)(float)x[i] in place (a covariance table as fp32 storage would hold it). */
int geobo_convert(int to_f32, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                  void* stream);
int geobo_round_f32(double* x, int64_t n, void* stream);

/* out[i] = w * amp * k(d2[i]; l1, l2), elementwise on a caller-supplied squared-distance array
 * (the literal signature of kernels.py:81-156: gpkernel(D2, gamma) ...). */
int geobo_k_eval(int kernel_id, const double* d2, int64_t n, double l1, double l2, double w, double amp,
                 double* out, void* stream);

/* Forward operator, sensormodel.py:29-93 (A_sens).  xe[nx+1], ye[ny+1], ze[nz+1] are the node
 * coordinates along each axis (ze already negated, inversion.py:61-66); loc is (Ms,3) row-major.
 * Writes A[n, p] for n < Ms, p < nx*ny*nz with leading dimension ld; reproduces the +-1e6 m padding on
 * the iy axis (sensormodel.py:63-68), the left-to-right 8-corner sum (:81-86) and `scale`
 * (c_MILLIGALS_UNITS/fcor_grav or 1/fcor_mag, :88-91: grav = (c*s)/f, magn = s/f -> pass mul, div). */
int geobo_a_sens(int func_id, const double* B3_host, const double* loc, int64_t Ms, int nx, int ny, int nz,
                 const double* xe, const double* ye, const double* ze, double scale_mul, double scale_div,
                 double* A, int64_t ld, void* stream);

/* The same operator restricted to the voxel slab iy0 <= iy < iy1 (columns p = (iy*nx+ix)*nz+iz of that slab only; the rest of
 * A is not touched): a rank of a column-sharded run only needs its own y-slab of every sensor row.  A[n, 0] stands for voxel
 * column col_origin: 0 for a full-width operator, iy0*nx*nz for a caller that holds ONLY the slab; the requested columns must
 * lie inside one buffer row, col_origin <= iy0*nx*nz and iy1*nx*nz - col_origin <= ld (GEOBO_E_ARG otherwise).  The same holds
 * for geobo_a_sens_lattice.  (Streamed operators of the 128^3 configuration: A is 275 GB per type and never resident.) */
int geobo_a_sens_slab(int func_id, const double* B3_host, const double* loc, int64_t Ms, int nx, int ny, int nz,
                      const double* xe, const double* ye, const double* ze, double scale_mul, double scale_div,
                      int iy0, int iy1, double* A, int64_t ld, int64_t col_origin, void* stream);

/* Lattice form of the same operator for sensors on a lattice commensurate with the voxel columns (sensor n at lattice column
 * jxs[n], jys[n]; all at one height): away from the +-1e6-padded node planes iy = 0 and ny the 8-corner stencil of
 * sensormodel.py:81-86 depends on (iy - jys, ix - jxs, iz) only.  dxv[2nx]: xe[j] - sx for j - jxs = -(nx-1) .. nx;
 * dyv[2ny-2]: ye[i] - sy for i - jys = 2-ny .. ny-1 (planes 1 .. ny-1); dzv[nz+1]: ze[k] - sz.  Fills the INTERIOR slabs
 * max(iy0,1) <= iy < min(iy1, ny-1) of A (the two boundary slabs come from geobo_a_sens_slab); ~2000x fewer potential
 * evaluations at 64^3, bit-identical to geobo_a_sens when the offsets are exact for every (node, sensor) pair (the caller's
 * check: geobo_amd/hip.py::lattice_plan).  nz even, ny >= 3; ws: geobo_a_sens_lattice_ws_bytes(nx, ny, nz). */
size_t geobo_a_sens_lattice_ws_bytes(int nx, int ny, int nz);
int geobo_a_sens_lattice(int func_id, const double* B3_host, int64_t Ms, int nx, int ny, int nz, const double* dxv,
                         const double* dyv, const double* dzv, const int* jxs, const int* jys, double scale_mul,
                         double scale_div, int iy0, int iy1, double* A, int64_t ld, int64_t col_origin, void* ws, size_t ws_bytes,
                         void* stream);

/* Node potential itself, elementwise: out[i] = grav_func(x,y,z) (sensormodel.py:96-110) or
 * magn_func(x,y,z,B) (sensormodel.py:113-133); same arithmetic as inside geobo_a_sens. */
int geobo_potential(int func_id, const double* B3_host, const double* x, const double* y, const double* z, int64_t n,
                    double* out, void* stream);

/* Fused covariance-assembly x forward-operator product (replaces inversion.py:92 + :114's np.dot(Asens3, kcov),
 * never materialising K):
 *     AK[r, c] = sum_p A[r, p] * (w*amp*k(|P_p - P_(col0+c)|^2; l1, l2)),   r < Ms_pad, c < ncols, p < N_pad
 * A is (Ms_pad x N_pad, lda) zero padded, x/y/z have N_pad entries (padding = any finite value),
 * ncols multiple of GEOBO_PAD_N, Ms_pad multiple of GEOBO_PAD_M.  FP64 MFMA contraction. */
int geobo_ak_fused(int kernel_id, const double* A, int64_t Ms_pad, int64_t N_pad, int64_t lda,
                   const double* x, const double* y, const double* z, int64_t col0, int64_t ncols,
                   double l1, double l2, double w, double amp, double* AK, int64_t ldak, void* stream);

/* Regular-grid form of the same product.  On the grid of calcGridPoints3D (kernels.py:27-42) every covariance is a
 * function of the index difference (|diy|,|dix|,|diz|), so the generator stage becomes an integer-indexed gather from
 * a lattice table (nx*ny*nz doubles, L2 resident) instead of exp/sqrt on the FP pipe that the fp64 MFMA also needs:
 *   geobo_cov_table:     table[(diy*nx + dix)*2nz + (dz + nz-1)] = w*amp*k(|P(0,0,0) - P(diy,dix,|dz|)|^2; l1, l2),
 *                        dz = -(nz-1)..nz-1 (z axis mirrored, 2*nx*ny*nz doubles), P = (i+1)*voxel
 *   geobo_ak_fused_grid: AK[r, c] = sum_p A[r, p] * table[|iy_p-iy_q|, |ix_p-ix_q|, iz_p-iz_q],  q = col0 + c
 * Same padding contract as geobo_ak_fused; requires nz >= 16 and even. */
int geobo_cov_table(int kernel_id, int nx, int ny, int nz, double sx, double sy, double sz, double l1, double l2,
                    double w, double amp, double* table, void* stream);
int geobo_ak_fused_grid(const double* A, int64_t Ms_pad, int64_t N_pad, int64_t lda, int nx, int ny, int nz,
                        const double* table, int64_t col0, int64_t ncols, double* AK, int64_t ldak, void* stream);

/* C = alpha * X * Y^T + beta * C   (X: m x k, Y: n x k, both k-contiguous).  inversion.py:96 (AkA = (A K) A^T),
 * Cholesky panel/trailing updates.  m % 256 == 0, n % 128 == 0, k % 16 == 0.
 * lower_only: flag word.  GEOBO_GEMM_LOWER_ONLY (1): tiles strictly above the diagonal are skipped (SYRK-style).
 * GEOBO_GEMM_SMALL_TILES (2): 128-row workgroup tiles (256 threads) even when m % 256 == 0 -- for short launches that must
 * find room next to a concurrent kernel on another stream (two such workgroups share a CU; a 512-thread workgroup needs a
 * whole CU to drain first).  The Cholesky panel solves use it.
 * m_valid > 0: rows >= m_valid of X are zero padding -- they are neither contracted (64-row groups that lie entirely in
 * the padding issue no MFMAs) nor stored (those rows of C are left untouched); 0 = all m rows. */
#define GEOBO_GEMM_LOWER_ONLY 1
#define GEOBO_GEMM_SMALL_TILES 2
int geobo_gemm_nt(int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                  const double* Y, int64_t ldy, double beta, double* C, int64_t ldc, int lower_only, int64_t m_valid,
                  void* stream);

/* Host-side view of the order in which a memory-mode launch with few, long tiles walks its (row tile, column tile) items
 * (pure arithmetic, evaluated per workgroup on the device: no tile lists are allocated, copied or cached by a launch).
 * Writes bi << 16 | bj for up to `capacity` items into the HOST array out_host (may be NULL) and returns the item count.
 * lower_only: tiles strictly above the diagonal are skipped; x_lower / y_lower: triangular-operand launches (longest
 * contraction first).  Used by the tests to prove that every tile is visited exactly once. */
int64_t geobo_tile_order(int nbi, int nbj, int tm, int tn, int lower_only, int x_lower, int y_lower, int* out_host,
                         int64_t capacity);

/* C = X * Y^T with the contraction split into `splits` slices that run concurrently (more, shorter workgroups: fills the
 * chip when m/256 * n/128 is only a few hundred tiles, as in AkA) and are summed in fixed order afterwards (deterministic).
 * ws: splits * m * n doubles.  k % (16 * splits) == 0.  m_valid as in geobo_gemm_nt (rows >= m_valid of C become 0). */
int geobo_gemm_nt_splitk(int64_t m, int64_t n, int64_t k, int splits, const double* X, int64_t ldx, const double* Y,
                         int64_t ldy, double* C, int64_t ldc, int lower_only, int64_t m_valid, void* ws, size_t ws_bytes,
                         void* stream);

/* C = alpha * X * Y + beta * C      (X: m x k k-contiguous, Y: k x n n-contiguous).
 * x_lower != 0: X is lower triangular (k range clipped to k < row_end of each tile);
 * y_lower != 0: Y is lower triangular (k range starts at the tile's first column). */
int geobo_gemm_nn(int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                  const double* Y, int64_t ldy, double beta, double* C, int64_t ldc,
                  int x_lower, int y_lower, void* stream);

/* Batched form of the two products above with store predicates: for b < batch
 *     C_b[:m_valid, :n_valid] = alpha * X_b * op(Y_b) + beta * C_b,   X_b = X + b*strideX etc. (stride 0 = shared operand)
 * y_is_kn = 0: Y_b is n x k (k-contiguous, "NT");  y_is_kn = 1: Y_b is k x n (n-contiguous, "NN").
 * m, n are the COMPUTE extents (multiples of 128; operands must be readable over them), m_valid <= m and n_valid <= n
 * the stored extents.  Used by the spectral (real-DFT) form of the covariance product: every axis pass of the 3-D
 * transform is a batch of small-k MFMA GEMMs against a fixed cosine/sine matrix. */
int geobo_gemm_batched(int y_is_kn, int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                       int64_t strideX, const double* Y, int64_t ldy, int64_t strideY, double beta, double* C, int64_t ldc,
                       int64_t strideC, int64_t m_valid, int64_t n_valid, int batch, void* stream);

/* Radix-2 ("folded") form of a geobo_gemm_batched axis pass whose matrix operand is the pair-interleaved real eigenvector basis G
 * (P x n, P = 2n; row 2b+1 = (-1)^i row 2b: geobo_amd/spectral.py forward_matrix) or its transpose -- the (x, z) transforms of the
 * covariance products A_s K_sj and (L^-1 A) K (kernels.py:158-195, inversion.py:96,114-117) and of the lattice Gram / lattice
 * convolution on every grid extent that has no fused two-axis kernel (geobo_xz2d_fold: 64 x 64 planes only).  SAME operands, strides
 * and extents as the geobo_gemm_batched call it replaces (alpha = 1, beta = 0), half the multiply-adds:
 *     inverse = 0 (analysis, k = n inputs -> P outputs):  out[2b], out[2b+1] = E_b +- O_b, E / O = even- / odd-input sums against row 2b
 *     inverse = 1 (synthesis, k = P -> n outputs):        out[i] = sum_b G[2b][i] (s[2b] + (-1)^i s[2b+1])
 *     y_is_kn = 0: the data is X (m x k, k-contiguous), the matrix is Y (n x k): G for the analysis, G^T for the synthesis;
 *     y_is_kn = 1: the data is Y (k x n, n-contiguous), the matrix is X (m x k);
 *     inverse = 2 (y_is_kn = 0 only): synthesis with the matrix G^T as X (m x k) and the data as Y in the n x k layout, C = G^T S^T.
 * m, n: compute extents (multiples of 128; operands readable over them, as for geobo_gemm_batched), k % 16 == 0, batch <= 65535.
 * Agrees with the plain product up to summation order.  GEOBO_E_ALIGN for odd strides / an odd valid extent along the pair axis. */
int geobo_gemm_fold(int y_is_kn, int inverse, int64_t m, int64_t n, int64_t k, const double* X, int64_t ldx, int64_t strideX,
                    const double* Y, int64_t ldy, int64_t strideY, double* C, int64_t ldc, int64_t strideC, int64_t m_valid,
                    int64_t n_valid, int64_t batch, void* stream);

/* x step of the lattice Gram (AkA = (A K) A^T on a lattice survey, inversion.py:96) for extents without the fused kernel
 * (geobo_xcorr_reduce(_fold): nx = nz = 64), in ONE launch: the radix-2 analysis D_b = G X_b of geobo_gemm_fold (y_is_kn = 1; G: px x k
 * basis rows, X_b = Y + b * strideY: k x nz) with the eigenvalue scaling and the channel sum of geobo_lamdot_z as its epilogue,
 *     out[b][o] = sum_(z < nz) D_b[o][z] * lam[(batch0 + b) % planes][o][z],      b < batch, o < px
 * -- the intermediate D (px * nz doubles per batch element: 67 MB per sensor row at 128^3) is never written.  nz <= 128 (one column
 * tile holds every channel; GEOBO_E_UNSUPPORTED otherwise: geobo_gemm_fold + geobo_lamdot_z), px even, k % 16 == 0, batch <= 65535. */
int geobo_gemm_fold_lamdot(int64_t px, int64_t nz, int64_t k, const double* G, int64_t ldg, const double* Y, int64_t ldy, int64_t strideY,
                           const double* lam, int planes, int64_t batch0, double* out, int64_t batch, void* stream);

/* out[i] = a[i] * b[i % nb]   (spectrum x eigenvalue table, broadcast over the batch) */
int geobo_scale_broadcast(const double* a, const double* b, int64_t n, int64_t nb, double* out, void* stream);
/* the same for two tables at once: out0 = a .* b0, out1 = a .* b1 (one read of the spectrum for both property blocks) */
int geobo_scale_broadcast2(const double* a, const double* b0, const double* b1, int64_t n, int64_t nb, double* out0,
                           double* out1, void* stream);

/* Fused (x, z) real-DFT passes of the structured product: every plane (r, p), r < rows,
 * p < planes_per_row, at in + r*in_row + p*in_plane goes  X -> Mx X Mz^T  to out + r*out_row + p*out_plane.
 * inverse = 0: X is nx x nz, Mx = G_x (2nx x nx), Mz = G_z (2nz x nz), result 2nx x 2nz (into the spectrum);
 * inverse = 1: X is 2nx x 2nz, Mx = G_x^T (nx x 2nx), Mz = G_z^T (nz x 2nz), result nx x nz (back, cropped).
 * Planes are dense row-major; strides in doubles, even; in 16-byte aligned.  (nx, nz) in {(48, 64), (64, 64), (64, 32)}
 * (GEOBO_E_UNSUPPORTED otherwise: use two geobo_gemm_batched passes).  The matrices are the caller's: 32 x 32 planes go
 * through (64, 32) two at a time, consecutive planes stacked along x with Mx = diag(Mx32, Mx32) (geobo_amd/spectral.py). */
int geobo_xz2d(int inverse, int nx, int nz, int64_t rows, int planes_per_row, const double* in, int64_t in_row,
               int64_t in_plane, const double* Mx, int64_t ldmx, const double* Mz, int64_t ldmz, double* out,
               int64_t out_row, int64_t out_plane, void* stream);

/* Radix-2 / radix-4 ("folded") form of geobo_xz2d for the pair-interleaved spectral basis (geobo_amd/spectral.py: position 2b = base
 * row g_b, position 2b+1 = (-1)^i g_b): per pair one even-input and one odd-input partial sum, outputs E + O and E - O -- half
 * the MFMAs of the plain matrix product.  Fx, Fz: [n][n/2][2] = (Fe[b][j], Fo[b][j]) = (g_b[2j], g_b[2j+1]) of the
 * row axis and of the contiguous axis.  inverse = 0: planes n x n -> 2n x 2n; inverse = 1: 2n x 2n -> n x n (cropped).
 * n = 64 (GEOBO_E_UNSUPPORTED otherwise: use geobo_xz2d); strides even, 16-byte aligned bases.
 * BASIS ORDER (version 210).  The dense-plane entry points (this one, _lattice, _inv_ss, _inv_strided, _inv_mul) additionally take
 * the base rows in the order of spectral.base_modes -- groups of four per frequency w < n/4: b = 4w + (cos w, sin w, cos(n/2 - w),
 * sin(n/2 - w)); b = 0..3: cos 0, the middle pair, cos(n/4), sin(n/4) -- so that the eight spectral positions 8w .. 8w+7 are the orbit
 * of w under a quarter-period shift.  On the inputs / outputs i = 4j + rho of one residue class the rows of frequency n/2 -+ w are
 * +-(cos | sin) of frequency w: the forward x step and both inverse steps contract ONE cosine and ONE sine row per group and class
 * (RADIX 4, a quarter of the plain multiply-adds; the forward z step stays radix 2) and meet the eight values of a group through
 * signed sums in one lane.  Only rows b = 4w and 4w + 1 of F are read on those steps (w = 0: row 0 and its alternating-sign copy);
 * a column scaling of the basis (F = folded(G diag(f))) is honoured.  geobo_xz2d_fold_quad (block-diagonal matrices) is radix 2. */
int geobo_xz2d_fold(int inverse, int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row, int64_t in_plane,
                    const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane, void* stream);

/* AkA = (A K) A^T (inversion.py:96) on a lattice survey ("lattice Gram"): the x step of a (y, x)
 * correlation of the rows of A K with the operator's stencil table, the z axis acting as a channel.  For plane (r, p), r < rows, p < planes, at in + r*in_row + p*in_plane (nx x nz, row-major):
 *     out[r*out_row + p*out_plane + o] = sum_z lamT[(p*nz + z)*2nx + o] * sum_x Mx[o][x] * in[x][z],   o < 2nx
 * (Mx = G_x, 2nx x nx; lamT = eigenvalues of the even stencil table per (y-mode p, channel z, x-mode o)).  nx = nz = 64. */
int geobo_xcorr_reduce(int nx, int nz, int64_t rows, int planes, const double* in, int64_t in_row, int64_t in_plane,
                       const double* Mx, int64_t ldmx, const double* lamT, double* out, int64_t out_row, int64_t out_plane,
                       void* stream);

/* Forward form of geobo_xz2d_fold fed straight from the stencil table of a lattice survey (geobo_a_sens_lattice): plane y of
 * operator row r is the contiguous window  Q + row_off[r] + y*q_plane  (n*n doubles) of the table, except the first and the last
 * plane of a row -- the 1e6-padded boundary slabs -- which are read from  edge + r*edge_row  and  edge + r*edge_row + n*n.  The
 * operator rows themselves (8.6 GB per operator at 64^3) are neither written nor read; the table (63 MB) stays in cache.
 * row_off: device array of `rows` offsets in doubles (even).  n = 64, planes_per_row >= 3. */
int geobo_xz2d_fold_lattice(int n, int64_t rows, int planes_per_row, const double* Q, const int64_t* row_off, int64_t q_plane,
                            const double* edge, int64_t edge_row, const double* Fx, const double* Fz, double* out,
                            int64_t out_row, int64_t out_plane, void* stream);

/* Inverse radix-2 transform FUSED WITH THE SUM OF SQUARES over the rows (the posterior variance, inversion.py:117 / :238:
 * diag(K - V^T V) needs sum_m V[m, q]^2 only): for every plane (r, y), r < rows, y < planes_per_row, the n x n result
 * X_r[y] = Fx^T-step(Fz^T-step(S_r[y])) is squared and accumulated, never stored.  Rows r >= r2_first are the SUM of two
 * spectra (in and in2: V rows that receive contributions of two operators) -- both contractions are linear, the first step
 * accumulates over the terms.  ss: geobo_xz2d_fold_inv_ss_slots(n, rows, planes_per_row) partial cubes [slot][y][n*n] that the call
 * ADDS to (zero them once per reduction; a slot / y pair is owned by one workgroup per launch: no atomics, deterministic);
 * sum over the slots afterwards.  n = 64; in2 may be NULL. */
/* Inverse radix-2 transform whose output planes need not be dense: row i of plane (r, p) goes to
 * out + r*out_row + p*out_plane + i*out_rowstride (n contiguous doubles).  With out_plane = n and out_rowstride = planes*n the planes
 * of a row interleave: the transposed lattice application writes L^-1 A rows as [iy][iz][ix] this way (one inverse (y, x) transform
 * per z channel).  n = 64. */
int geobo_xz2d_fold_inv_strided(int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row, int64_t in_plane,
                                const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane,
                                int64_t out_rowstride, void* stream);

/* The same with the input plane (r, p) given as the elementwise PRODUCT  a[p] * b[r]  of two planes of 2n x 2n doubles (a + p*a_plane:
 * one per plane index, b + r*b_row: one per row): the transposed lattice application has W[r][iz] = Lambda[iz] * lhat_r as its
 * input, and both factors stay in cache where W (8.4 MB per row at 64^3) would be written to HBM and read back.  The kernel stages
 * its chunks through registers (two loads, one multiply, one LDS store per 16 bytes) instead of LDS-DMA and is bound by the matrix
 * pipe.  Same arithmetic as geobo_lattice_wplanes + geobo_xz2d_fold_inv_strided up to the summation order of the first contraction
 * (1e-15).  n = 64. */
int geobo_xz2d_fold_inv_mul(int n, int64_t rows, int planes_per_row, const double* a, int64_t a_plane, const double* b, int64_t b_row,
                            const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane,
                            int64_t out_rowstride, void* stream);

/* geobo_xz2d_fold for planes of HALF the extent (n = 32) on the n = 64 kernels: a group of four consecutive planes y .. y+3 of a row is one
 * kernel plane [[y, y+1], [y+2, y+3]] (the kernel reads / writes half-length memory rows; right half and bottom rows come from the
 * neighbouring planes), Fx / Fz are the folded matrices of diag(G_32, G_32) ((64, 32, 2)): four independent transforms per kernel
 * plane, half of the MFMAs on zero blocks -- the passes are HBM bound either way (config 2's 32^3 grid: 0.55 / 0.8 ms per 2048 rows where
 * the stacked-pair form of geobo_xz2d took 1.2 / 3.4).  in_plane / out_plane: stride between consecutive SMALL planes of a row
 * (inverse = 0: n x n in, 2n x 2n out; inverse = 1 the other way round); groups_per_row = planes per row / 4.  n = 32. */
int geobo_xz2d_fold_quad(int inverse, int n, int64_t rows, int groups_per_row, const double* in, int64_t in_row, int64_t in_plane,
                         const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane, void* stream);

int geobo_xz2d_fold_inv_ss_slots(int n, int64_t rows, int planes_per_row);
int geobo_xz2d_fold_inv_ss(int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row, int64_t in_plane,
                           const double* in2, int64_t in2_row, int64_t r2_first, const double* Fx, const double* Fz, double* ss,
                           void* stream);

/* y step of the lattice Gram: the same (m x k) matrix G from the left of every row,  out[r][i][c] = sum_j G[i][j] in[r][j][c]
 * for r < rows, c < C (in: rows of k x C at in + r*in_row, out: rows of m x C at out + r*out_row; strides in doubles).
 * m = 128, k = 64 (GEOBO_E_UNSUPPORTED otherwise: geobo_gemm_batched does the same), C % 64 == 0, in 16-byte aligned. */
int geobo_ymul(int m, int k, int64_t C, int64_t rows, const double* G, int64_t ldg, const double* in, int64_t in_row,
               double* out, int64_t out_row, void* stream);
/* The same product for a PAIR-INTERLEAVED matrix (row 2b+1 = (-1)^j row 2b: the basis of geobo_amd/spectral.py, also with columns zeroed
 * or scaled) in radix 2: per pair one even-input and one odd-input sum, out[2b] = E + O, out[2b+1] = E - O -- half the MFMAs of
 * geobo_ymul, which sat on the matrix pipe and on HBM at once.  Only the rows 2b of G are read.  Same shapes and errors. */
int geobo_ymul_fold(int m, int k, int64_t C, int64_t rows, const double* G, int64_t ldg, const double* in, int64_t in_row,
                    double* out, int64_t out_row, void* stream);

/* Folded form of geobo_xcorr_reduce for the pair-interleaved basis in the quarter-period group order of geobo_xz2d_fold (version 210):
 * the x product in RADIX 4 -- one cosine and one sine row per frequency group and residue class of x mod 4, a quarter of the MFMAs of
 * the plain product (rounds 2-4: radix 2, half); F = [n][n/2][2] folded matrices of the x axis (rows 4w, 4w + 1 are read); lamT and out
 * in spectral position order as before.  A workgroup keeps one of the `planes` y-modes for the whole launch (planes <= 2048).  n = 64. */
int geobo_xcorr_reduce_fold(int n, int64_t rows, int planes, const double* in, int64_t in_row, int64_t in_plane,
                            const double* F, const double* lamT, double* out, int64_t out_row, int64_t out_plane, void* stream);

/* y-axis stage of the structured product on a regular grid: for every mode c < C (the (x, z)
 * spectral index, contiguous) and row r < R,   out_j[r][y - y0][c] = sum_{y'} tab_j[|y - y'|][c] * in[r][y'][c]
 * for y in [y0, y1) -- the symmetric Toeplitz blocks of create_cov's K_sj (kernels.py:158-195) applied directly.
 * in: [R][ny][plane]; tab_j: [ny][C]; out_j: [R][y1-y0][plane]; plane >= C is the stride between the y-planes of in and out in
 * doubles (even): with plane == C a power of two (16384 at 64^3) the ny planes a lane walks sit on the same HBM channels, which
 * costs a quarter of the bandwidth (profiles/r03_hbm_copy_runs.txt) -- callers pad it.  nprop = 1 or 2 property blocks per
 * sweep (tab1/out1 unused for 1).  ny in {16, 32, 48, 64} run here; {80, 96, 112, 128} are forwarded to geobo_toeplitz_y3
 * (GEOBO_E_UNSUPPORTED otherwise), C % 64 == 0, ny*plane*8 < 2^31. */
int geobo_toeplitz_y(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* tab0, const double* tab1,
                     double* out0, double* out1, int y0, int y1, void* stream);

/* The same stage for rows that are the SUM of two covariance products (the rows of the transposed posterior behind the gravity block,
 * V = Z_g K_0j + Z_m K_1j, inversion.py:114-117 re-associated), two property blocks per sweep, full height:
 *     out_j[r][y][c] = sum_{y'} tab_gj[|y - y'|][c] * in_g[r][y'][c] + tab_mj[|y - y'|][c] * in_m[r][y'][c],   j = 0, 1
 * One workgroup of eight waves (term, block, half of the outputs) shares both input rows; the second term's partial sums reach their
 * partners through the consumed input buffer in LDS: one output stream per block instead of two (and one input stream less for the
 * inverse transform that follows).  ny in {32, 48, 64} (GEOBO_E_UNSUPPORTED otherwise), C % 64 == 0, plane >= C even. */
int geobo_toeplitz_y2t(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_g0,
                       const double* tab_g1, const double* tab_m0, const double* tab_m1, double* out0, double* out1, void* stream);

/* The same two-term rows where the cross blocks coincide, K_10 = K_01 -- create_cov's prior is symmetric (kernels.py:181-195: block
 * (i, j) = W[i][j] k_cross(l_i, l_j) with W symmetric and every cross kernel even in the exchange of its lengths) -- as THREE products
 * per mode instead of four:
 *     out_0 = T(tab_d0) in_g + T(tab_x)(in_g + in_m),     out_1 = T(tab_d1) in_m + T(tab_x)(in_g + in_m)
 * with tab_x = the shared generator t_01, tab_d0 = t_00 - t_01, tab_d1 = t_11 - t_01 (formed by the caller, once per step).  Eight waves
 * of two sizes -- four own a half of the outputs of a difference product, four a quarter of the shared one -- so that every SIMD issues
 * 3/4 of the multiply-adds of geobo_toeplitz_y2t; the shared sums reach their two consumers through a dedicated exchange area
 * (the kernel takes all 160 KiB of a CU's LDS at ny = 64).  Same shapes and errors as geobo_toeplitz_y2t. */
int geobo_toeplitz_y2s(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_d0,
                       const double* tab_x, const double* tab_d1, double* out0, double* out1, void* stream);

/* The same stage for up to THREE property blocks per sweep of the input (tabs / outs: HOST arrays of nprop device pointers) and
 * for ny in {80, 96, 112, 128} (128: BASELINE config 5, 128^3 x 3 properties): there the ny table values of a mode no longer fit a
 * lane's registers; a lane owns one mode, one half of the inputs and a chunk of 16 outputs, whose distances form a window of ny/2 + 15 table values with
 * static register indices; a workgroup of eight waves shares one staged input row between 4 / 2 / 1 output chunks (1 / 2 / 3 blocks;
 * three blocks run as 2 + 1), so a full-height product stages every row ny / (16 x chunks) times.
 * ny in {16, 32, 48, 64} is forwarded to geobo_toeplitz_y two blocks at a time. */
int geobo_toeplitz_y3(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                      double* const* outs, int y0, int y1, void* stream);
/* outs[j] += (the same sums): the second term of a two-term row V = Z_g K_0j + Z_m K_1j adds into the first term's spectrum, so that
 * ONE inverse transform per block follows (shape-independent form of the transposed posterior; what geobo_toeplitz_y2t does for
 * ny <= 64).  ny in {80, 96, 112, 128}; GEOBO_E_UNSUPPORTED otherwise. */
int geobo_toeplitz_y3_add(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                          double* const* outs, int y0, int y1, void* stream);

/* The y stage as an IN-KERNEL SPECTRAL PRODUCT on the fp64 matrix pipe (round 6; csrc/spectral_y.hip): the same sums as
 * geobo_toeplitz_y / geobo_toeplitz_y2s from the same arguments (tab_j are the Toeplitz generators, [ny][C]), computed through the
 * y axis's own spectrum without leaving the registers.  T_c is the leading block of a skew-circulant of size 2 ny, diagonalised by
 * cos / sin of the half-integer frequencies 1/2 .. ny - 1/2; radix 4 over the orbits {k, ny - k, ny/2 + k, ny/2 - k} of a
 * quarter-period shift: per mode ny^2 / 2 multiply-adds per transform on MFMA (the transform matrix is the same for every mode: the
 * 16 MFMA columns are 16 modes, 128-byte segments) plus 16 additions per orbit.  One term and two blocks cost 1.5 ny^2 (direct:
 * 2 ny^2), two-term rows with the shared cross block 2 ny^2 (direct, three products: 3 ny^2).  Results agree with the direct kernels
 * to rounding (a few 1e-16 of sum |t| |x|), not bit for bit.
 * basis: geobo_spectral_y_basis_doubles(ny) doubles filled ONCE per ny by geobo_spectral_y_basis (transform fragments in lane order:
 * the library keeps no state of its own).  ny in {32, 48, 64} here, {80, 96, 112, 128} through geobo_spectral_y3 (GEOBO_E_UNSUPPORTED otherwise; 0 doubles), C % 16 == 0, plane >= C,
 * ny*plane*8 < 2^31.  Replaces the same reference lines as geobo_toeplitz_y: kernels.py:158-195, inversion.py:96,114-117. */
int64_t geobo_spectral_y_basis_doubles(int ny);
int geobo_spectral_y_basis(int ny, double* basis, void* stream);
int geobo_spectral_y(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* tab0, const double* tab1,
                     double* out0, double* out1, int y0, int y1, const double* basis, void* stream);
int geobo_spectral_y2s(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_d0,
                       const double* tab_x, const double* tab_d1, double* out0, double* out1, const double* basis, void* stream);

/* The same in-kernel spectral product for up to THREE property blocks per sweep (tabs / outs: HOST arrays of nprop device pointers) and
 * for the long y axes ny in {80, 96, 112, 128} (128: BASELINE config 5), the drop-in for geobo_toeplitz_y3 / geobo_toeplitz_y3_add
 * (accumulate != 0: outs[j] += the sums -- the second term of a two-term row).  There a lane can no longer hold a tile's spectrum and its
 * eigenvalues: FOUR waves share one tile of 16 modes -- each analyses one tile of eight orbits with its own eigenvalues, the scaled
 * class sums change hands through 32 KiB of LDS (two barriers per block), each wave synthesises one residue class of the outputs.
 * Per mode (n_in + n_out) ny^2 / 2 multiply-adds on MFMA: 2 ny^2 for one term and three blocks (direct: 3 ny^2).
 * ny <= 64 is forwarded to geobo_spectral_y two blocks at a time (accumulate: GEOBO_E_UNSUPPORTED there).  Same basis blob. */
int geobo_spectral_y3(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                      double* const* outs, int y0, int y1, int accumulate, const double* basis, void* stream);

/* Two-term rows on the long y axes in ONE pass (the drop-in for geobo_toeplitz_y3 followed by geobo_toeplitz_y3_add):
 *     outs[j] = T(tabs_g[j]) in_g + T(tabs_m[j]) in_m,   j < nprop <= 3,  full height
 * -- the terms meet in the y spectrum (lambda_gj x^_g + lambda_mj x^_m): two analyses + nprop syntheses instead of 2 (1 + nprop)
 * transforms, and no read-modify-write of the outputs.  ny in {80, 96, 112, 128}; ny <= 64: geobo_spectral_y2s. */
int geobo_spectral_y3t(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in_g, const double* in_m,
                       const double* const* tabs_g, const double* const* tabs_m, double* const* outs, const double* basis, void* stream);

/* Axis passes of the (x, z) transforms along a STRIDED axis for extents without a fused two-axis kernel (round 6; the x step of the
 * covariance product's transforms, kernels.py:158-195 / inversion.py:96,114 through the spectral route):
 *     inverse == 0 (analysis):   out[item][p][c] = sum_i G[p][i] in[item][i][c],   p < 2n, i < n
 *     inverse != 0 (synthesis):  out[item][i][c] = sum_p G[p][i] in[item][p][c]
 * with G the real eigenvector basis of size n on HALF-INTEGER frequencies in orbit order (geobo_amd/spectral.py forward_matrix /
 * half_modes: spectral positions 8 w .. 8 w + 7 = sqrt(2) x {cos k, cos(n - k), sin k, -sin(n - k), cos(n/2 + k), cos(n/2 - k),
 * sin(n/2 + k), -sin(n/2 - k)}, k = w + 1/2) -- the drop-in for geobo_gemm_fold / geobo_gemm_batched with that matrix on the X side,
 * radix 4 on MFMA (n^2 / 2 multiply-adds per item and mode), data straight from global memory into the operand layout.
 * c < C contiguous modes (C % 16 == 0); planes plane_in / plane_out doubles apart, items item_in / item_out doubles apart.
 * mask_ends != 0 (analysis only): the input planes 0 and n - 1 count as zero -- the y step of the lattice Gram, whose boundary slabs
 * (the +-1e6 m padding of A_sens) do not enter the stencil correlation.
 * basis: the blob of geobo_spectral_y_basis(n).  n in {80, 96, 112, 128}; GEOBO_E_UNSUPPORTED otherwise. */
int geobo_spectral_axis(int inverse, int n, int64_t C, int64_t plane_in, int64_t plane_out, int64_t item_in, int64_t item_out, int64_t items,
                        const double* in, double* out, const double* basis, int mask_ends, void* stream);

/* In-place lower Cholesky of the (m x m, ld) matrix A, m % 256 == 0 (padding rows/cols = identity);
 * scipy.linalg.cholesky(AkA, lower=True), inversion.py:100.  Also writes Linv = L^-1 (m x m, ldi; lower,
 * upper part zeroed), used instead of the two solve_triangular calls (inversion.py:105,114).
 * info (device int32): 0 ok, j>0 = first non-positive / NaN pivot (1-based), like LAPACK dpotrf.
 * ws: workspace of geobo_potrf_ws_bytes(m) bytes.
 * From m = 1024 the call is ONE persistent launch (round 5, potrf.hip "tile DAG"): one 256-thread workgroup per CU draws the 128 x 128
 * tiles of L (left-looking: a tile is accumulated in registers over its whole contraction) and of L^-1 (forward substitution by block
 * rows) in dependency order from a global counter and synchronises through agent-scope counters in ws; every spin is bounded -- info
 * = -7 reports a scheduler time-out (never seen; the result is then undefined).  ctx is not used by that form.  Up to 40 block columns
 * (m <= 5120, round 6) ONE workgroup walks the latency chain -- diagonal tile, tile under it, update of the next diagonal tile -- with
 * the other workgroups parking the partial sums it needs; larger matrices keep every link a task of its own.  Below m = 1024, or
 * with GEOBO_POTRF=streams in the environment, the stream schedule of rounds 2-4 runs:
 * ctx: fork context or NULL.  With a context (three internal streams, ordered after / before `stream` by events) the
 * trailing update of every step runs one step behind on the first stream (look-ahead), and the L^-1 tree -- dozens of small
 * merges -- is built under the factorisation: every node's two GEMMs are queued on the second stream as soon as the columns of L
 * they read are final (27 -> 17 ms at m = 8448; one T buffer per node in ws).
 * A context is made ONCE at set-up time (geobo_potrf_ctx_create: the only entry points of this library that
 * create runtime objects, never called from a launch path), belongs to the device that was current then, and serves one
 * factorisation at a time: concurrent factorisations (other streams, other threads, other devices) each bring their own.
 * The library keeps no process-global streams, events, caches or locks. */
size_t geobo_potrf_ws_bytes(int64_t m);
int geobo_potrf_ctx_create(void** ctx);
int geobo_potrf_ctx_destroy(void* ctx);
int geobo_potrf_inv(int64_t m, double* A, int64_t ld, double* Linv, int64_t ldi, int* info, void* ws,
                    size_t ws_bytes, void* ctx, void* stream);

/* Row dot products  out[r] = sum_(c < n) X[r, c] * v[c]  of a row-major matrix (n even, ld even, X and v 16-byte aligned): a forward
 * operator applied to a model, data = A rho (simcube.py:147-150 -- the synthetic surveys of bench.py and of the tests; np.dot there).
 * One workgroup per row, fixed reduction tree: deterministic.  HBM read bound. */
int geobo_rowgemv(int64_t m, int64_t n, const double* X, int64_t ld, const double* v, double* out, void* stream);

/* Sum of squares over the rows of a batch of V = (L^-1 A3) K (the transposed order of inversion.py:114-117; diag of K - V^T V,
 * inversion.py:238) for grids without the fused inverse-transform reduction (geobo_xz2d_fold_inv_ss, n = 64):
 *     ss[slot][c] += sum_{r < rows, r % slots == slot} (a[r][c] + b[r][c])^2,   c < n      (b may be NULL: one-term rows)
 * a, b: row-major (rows x n, lda / ldb), what the storing covariance product wrote; ss: `slots` partial vectors ld_ss apart that the
 * call ADDS to (zero them once per reduction, sum over the slots afterwards).  A (slot, column) pair is owned by one thread and
 * the rows are added in ascending order: deterministic.  n, lda, ldb, ld_ss even; pointers 16-byte aligned.  HBM bound. */
int geobo_sumsq_accum(int64_t rows, int64_t n, const double* a, int64_t lda, const double* b, int64_t ldb, int slots, double* ss,
                      int64_t ld_ss, void* stream);

/* Eigenvalue scaling + channel sum of the lattice Gram's x step (AkA on a lattice survey, inversion.py:96) for grids without the
 * fused kernel (geobo_xcorr_reduce(_fold), nx = nz = 64):  out[b][o] = sum_(z < nz) D[b][o][z] * lam[b % planes][o][z]
 * for b < batch, o < px, with D = Gx X_b from a geobo_gemm_batched launch ([batch][px][nz], dense) and lam the stencil table's
 * eigen-data as [planes][px][nz].  nz % 16 == 0; D, lam 16-byte aligned. */
int geobo_lamdot_z(int64_t batch, int planes, int px, int nz, const double* D, const double* lam, double* out, void* stream);

/* Posterior mean/variance without storing V = L^-1 (A K)   (inversion.py:114-117 + :238's np.diag):
 *     V = Linv * AK (tile by tile, MFMA),  mu[c] = sum_m V[m,c] u[m],  var[c] = prior_var - sum_m V[m,c]^2
 * AK: (m x ncols, ldak); u: m; ws: geobo_posterior_ws_bytes(m, ncols). m % 256 == 0, ncols % 128 == 0.
 * m_valid > 0: rows >= m_valid are padding (identity rows of Linv over zero rows of AK, so V = 0 there): their 64-row
 * groups are skipped; 0 = all rows. */
size_t geobo_posterior_ws_bytes(int64_t m, int64_t ncols);
int geobo_posterior_reduce(int64_t m, int64_t ncols, const double* Linv, int64_t ldi, const double* AK, int64_t ldak,
                           const double* u, double prior_var, double* mu, double* var, int64_t m_valid, void* ws,
                           size_t ws_bytes, void* stream);

/* u = Linv * y (lower-triangular mat-vec, wavefront shuffle reduction); also
 * stats[0] = u.u, stats[1] = sum_i log(L_ii^2)   (inversion.py:105-110).  Ldiag = L (m x m, ld). */
int geobo_trmv_stats(int64_t m, const double* Linv, int64_t ldi, const double* y, const double* L, int64_t ld,
                     double* u, double* stats, void* stream);

/* micro-benchmark used by bench.py to pin the fp64 MFMA ceiling on the box: each of `blocks` workgroups
 * (256 threads) issues iters x 16 independent v_mfma_f64_16x16x4_f64; out receives a checksum. */
int geobo_mfma_f64_peak(int blocks, int iters, double* out, void* stream);

/* co-issue probe: `nv` VALU ops (mode 1 fp64 fma, 2 fp32 fma, 3 int mad; 0 none) after every fp64 MFMA of the same
 * wave; used to decide what the generator stage may cost (DESIGN.md section 4 "what shares the fp64 pipe"). */
int geobo_mfma_mix(int mode, int nv, int blocks, int iters, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOBO_HIP_H */
