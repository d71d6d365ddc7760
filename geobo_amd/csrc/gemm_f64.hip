// gemm_f64.hip -- fp64 MFMA contraction core for gfx950 and the kernels built on it:
//
//   * ak_fused        AK = A * K        K tile GENERATED in-kernel from voxel coordinates (never stored);
//                                        replaces create_cov + np.dot(Asens3, kcov)  (kernels.py:158-195,
//                                        inversion.py:92,96,114)
//   * gemm_nt         C = a X Y^T + b C  (AkA = (A K) A^T, inversion.py:96; Cholesky panel / trailing updates)
//   * gemm_nn         C = a X Y   + b C  (L^-1 assembly; optional triangular k-range clipping)
//   * posterior_reduce  V = Linv * AK tile by tile, reduced on the fly to mu = V^T u and sum_m V^2
//                                        (inversion.py:114-117, :238) -- V is never written.
//
// Design (MI355X): one workgroup = WM x WN wavefronts of 64 lanes, each wave owns a 64x64 block of C as a
// 4x4 grid of v_mfma_f64_16x16x4_f64 accumulators (16 x 4 f64 = 128 VGPRs).  The contraction index is
// walked in chunks of 16; per chunk the X tile ([TM][16], k-contiguous rows) and the Y tile are staged
// in double-buffered LDS (one barrier per chunk).  Inside a chunk MFMA step t contracts
// k in {t, 4+t, 8+t, 12+t}: lane (r = lane&15, g = lane>>4) supplies X[row r][4g+t] and Y[col r][4g+t], so
// every lane reads ONE contiguous 32-byte run per 16-row group instead of four strided f64 -- legal because
// the A and B operands use the same permutation of k.
// f64 MFMA fragment layout (differs from every other dtype): A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D: col = lane&15, row = (lane>>4) + 4*reg.
// The generator stage computes 16 x TN covariances per chunk on the VALU (4-8 per thread) while the matrix
// pipe runs the previous chunk's 64 MFMAs per wave; p-coordinates are wave-uniform and come in through
// scalar loads, q-coordinates live in registers for the whole kernel.
// Workgroup -> tile map is XCD aware: block b runs on XCD b%8, so consecutive slots of one XCD get the SAME
// row tile (they stream the same A rows through that XCD's L2) and different column tiles.
#include <hip/hip_runtime.h>
#include <atomic>
#include <type_traits>
#include <stdint.h>
#include "covfun.h"
#include "geobo_hip.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16;   // contraction chunk
constexpr int XS = 16;   // LDS row stride in doubles for [rows][16] tiles: 128-byte rows, XOR-swizzled 16-byte slots
constexpr int NST = 3;   // LDS ring depth of the 8-wave configuration; 4-wave workgroups use 2 stages (two fit one CU)

// 16-byte slot swizzle of a [rows][8 slots] tile: slot' = slot ^ swz(row).  With 128-byte rows a wave's ds_read_b128
// (lane = (row r = lane&15, k-group g = lane>>4), slot 2g+h) then touches 16 distinct 16-byte positions of the 256-byte
// bank row in every hardware lane group {0-3,12-15,20-27},{4-11,16-19,28-31},... -> conflict free without padding.
__device__ __forceinline__ int swz(int row) {
  const int p = (row >> 1) & 7;
  return (p & 1) | (((p >> 2) & 1) * 6);
}

enum { Y_GEN = 0, Y_NT = 1, Y_NN = 2, Y_TAB = 3 };
enum { EPI_STORE = 0, EPI_REDUCE = 1 };
enum { TRI_LOWER_ONLY = 1, TRI_X_LOWER = 2, TRI_Y_LOWER = 4 };

struct GemmArgs {
  const double* X; int64_t ldx;
  const double* Y; int64_t ldy;
  double* C; int64_t ldc;
  int64_t k;
  double alpha, beta;
  int nbi, nbj, tri, xcd_map;
  // xcd_map 3: balanced item order (tile_of_item), consumed 32 consecutive items per XCD at a time; ntiles = items per batch
  int ntiles;
  // batching (blockIdx.y) and store predicates (the compute tile grid may overhang the valid m x n region)
  int64_t sXb, sYb, sCb, m_valid, n_valid; int batch;
  // generator
  const double *px, *py, *pz; int64_t gcol0;
  CovParams cov;
  // lattice-table generator (Y_TAB): table[(|diy|*gnx + |dix|)*gnz + |diz|] = w*amp*k on the grid's difference lattice
  const double* table; int gnx, gny, gnz; int64_t gN;
  // reduce epilogue
  const double* u; double* part_mu; double* part_ss; int64_t ncols;
  // row tiles start at bi * TM - row_shift (a multiple of 64): the under-filled tile of a ragged row count is the FIRST one
  int64_t row_shift;
};

// Item order of memory-mode launches with few, long tiles (AkA, split-K slices, Cholesky trailing updates, L^-1 merges).
// Pure arithmetic on wave-uniform values (SALU), shared by host (item count) and device (decode): nothing is allocated,
// copied or cached for a launch.
//   plain / lower_only: bands of four row tiles, column-major inside a band, tiles strictly above the diagonal skipped, so
//     32 consecutive items form a 4 x 8 supertile (4 X + 8 Y panels per XCD L2) and every workgroup has a full tile of work;
//   triangular X (k range [0, row_end)): longest contraction first = row tiles descending, equal lengths adjacent;
//   triangular Y (k range [col0, k)):    longest first = column tiles ascending.
__host__ __device__ inline int lower_cols(int bi, int nbj, int tm, int tn) {   // column tiles of row tile bi at or below the diagonal
  const int64_t c = ((int64_t)(bi + 1) * tm + tn - 1) / tn;
  return c < nbj ? (int)c : nbj;
}
__host__ __device__ inline int64_t tile_items(int nbi, int nbj, int tm, int tn, int tri) {
  if (!(tri & TRI_LOWER_ONLY)) return (int64_t)nbi * nbj;
  int64_t n = 0;
  for (int bi = 0; bi < nbi; ++bi) n += lower_cols(bi, nbj, tm, tn);
  return n;
}
__host__ __device__ __forceinline__ void tile_of_item(int t, int nbi, int nbj, int tm, int tn, int tri, int& bi, int& bj) {
  if (tri & TRI_X_LOWER) { bi = nbi - 1 - t / nbj; bj = t % nbj; return; }
  if (tri & TRI_Y_LOWER) { bj = t / nbi; bi = t % nbi; return; }
  if (!(tri & TRI_LOWER_ONLY)) {
    const int band = t / (4 * nbj), r0 = 4 * band;
    const int rows = nbi - r0 < 4 ? nbi - r0 : 4;
    t -= band * 4 * nbj;
    bj = t / rows; bi = r0 + t % rows;
    return;
  }
  int r0 = 0;
  for (;; r0 += 4) {                                   // <= nbi / 4 iterations (lower_only launches have nbi <= a few dozen)
    int cnt = 0;
    for (int i = r0; i < r0 + 4 && i < nbi; ++i) cnt += lower_cols(i, nbj, tm, tn);
    if (t < cnt || r0 + 4 >= nbi) break;
    t -= cnt;
  }
  const int r1 = r0 + 4 < nbi ? r0 + 4 : nbi;
  for (bj = 0; bj < nbj; ++bj) {                        // column bj holds the band's rows bi >= floor(bj tn / tm)
    const int first = (int)(((int64_t)bj * tn) / tm);
    const int lo = first > r0 ? first : r0;
    const int c = r1 - lo;
    if (c <= 0) continue;
    if (t < c) { bi = lo + t; return; }
    t -= c;
  }
  bi = nbi; bj = 0;                                     // past the end (not reached: the grid is cut at the item count)
}

template <int WM, int WN>
constexpr int ring_depth() { return (WM * WN >= 8) ? NST : 2; }

template <int WM, int WN, int YMODE>
constexpr int lds_doubles() {
  constexpr int TM = 64 * WM, TN = 64 * WN;
  constexpr int ybuf = (YMODE == Y_NN) ? BK * (TN + 4) : TN * XS;
  return ring_depth<WM, WN>() * (TM * XS + ybuf);
}

template <int WM, int WN, int YMODE, int EPI, int KID>
__global__ void __launch_bounds__(64 * WM * WN, 2) gemm_f64_kernel(const GemmArgs a) {
  constexpr int TM = 64 * WM, TN = 64 * WN, NT = 64 * WM * WN;
  constexpr int YS_NN = TN + 4;
  constexpr int XBUF = TM * XS;
  constexpr int YBUF = (YMODE == Y_NN) ? BK * YS_NN : TN * XS;
  constexpr int XU = TM * 8 / NT;                                    // 16-byte units of X per thread per chunk
  constexpr int YU = (YMODE == Y_NN) ? (8 * TN / NT) : (TN * 8 / NT);  // same for Y (memory / lattice modes)
  constexpr int EPT = BK * TN / NT;                                  // generated covariances per thread per chunk
  static_assert(NT % TN == 0 && EPT >= 2 && EPT % 2 == 0, "generator mapping");

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const Xs = smem;
  constexpr int NSTG = ring_depth<WM, WN>();
  double* const Ys = smem + NSTG * XBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Waves w and w + 4 share a SIMD.  With four 64-row groups per tile the pairs are (group 0, group 3) and (group 1, group 2):
  // whenever the groups carry unequal work -- the diagonal block of a triangular X operand gives group g a contraction
  // g + 1 quarter-blocks deep, the last row tile has only its first groups valid -- both SIMD pairs get the same total.
  const int wrow = wave / WN, wn = wave % WN;
  const int wm = (WM == 4) ? (wrow < 2 ? wrow : 5 - wrow) : wrow;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- workgroup -> tile ------------------------------------------------------------------------------
  int bi, bj;
  int batch_idx = blockIdx.y;
  {
    const int b = blockIdx.x;
    if (a.xcd_map == 1) {          // one row tile per XCD at a time (generator modes: only the X operand is streamed)
      const int xcd = b & 7, s = b >> 3;
      bj = s % a.nbj;
      bi = xcd + 8 * (s / a.nbj);
    } else if (a.xcd_map == 2) {   // memory modes: every XCD works on a 4 x 8 supertile (32 workgroups share 4 X and 8 Y panels in its L2)
      const int xcd = b & 7, s = b >> 3;
      const int within = s & 31, g = (s >> 5) * 8 + xcd;
      const int nsi = (a.nbi + 3) >> 2;
      if (a.tri & TRI_X_LOWER) {
        // contraction length grows with the row tile: the 32 workgroups an XCD runs together (and the 8 XCDs' groups
        // dispatched together) must weigh the same -- mixed lengths desynchronise the k sweeps (no operand sharing in L2)
        // and stall the in-order dispatcher.  One row tile x 32 column tiles, rows heavy-first: measured 67.4 -> 73.0 TF/s
        // on the posterior shape (a 4 x 8 supertile mixes four lengths; 2 x 16 gives 71.3).
        const int nsj = (a.nbj + 31) >> 5;
        bi = g / nsj;
        bj = 32 * (g % nsj) + within;
      } else {
        bi = 4 * (g % nsi) + (within & 3);
        bj = 8 * (g / nsi) + (within >> 2);
      }
      if (bj >= a.nbj) return;
    } else if (a.xcd_map == 3) {
      // (batch, tile) items in the balanced order of tile_of_item, 32 consecutive items per XCD at a time, 1-D grid: the dispatcher
      // places workgroup b on XCD b % 8, so every XCD gets the same number of full 32-item groups over the WHOLE launch
      // (with a (tiles, batch) grid the same XCDs draw the partial groups in every slice and split-K cannot shorten the tail)
      const int xcd = b & 7, s = b >> 3;
      const int64_t item = ((int64_t)((s >> 5) * 8 + xcd) << 5) + (s & 31);
      if (item >= (int64_t)a.ntiles * a.batch) return;
      batch_idx = (int)(item / a.ntiles);
      tile_of_item((int)(item - (int64_t)batch_idx * a.ntiles), a.nbi, a.nbj, TM, TN, a.tri, bi, bj);
    } else {
      bi = b % a.nbi;
      bj = b / a.nbi;
    }
  }
  if (bi >= a.nbi) return;
  const double* const Xp = a.X + (int64_t)batch_idx * a.sXb;
  const double* const Yp = a.Y + (int64_t)batch_idx * a.sYb;
  double* const Cp = a.C + (int64_t)batch_idx * a.sCb;
  if ((a.tri & TRI_X_LOWER) && a.xcd_map != 3) bi = a.nbi - 1 - bi;  // triangular X: the longest contraction ranges are dispatched first
  const int64_t row0 = (int64_t)bi * TM - a.row_shift, col0 = (int64_t)bj * TN;
  if ((a.tri & TRI_LOWER_ONLY) && col0 >= row0 + TM) return;
  int64_t kb = (a.tri & TRI_Y_LOWER) ? col0 : 0;
  int64_t ke = a.k;
  if ((a.tri & TRI_X_LOWER) && ke > row0 + TM) ke = row0 + TM;

  v4d acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (v4d){0., 0., 0., 0.};
  // C <- alpha X Y^T + beta C with |beta| = |alpha| (rank-k updates of the Cholesky trailing matrix, accumulating panels): the
  // accumulators START as (beta / alpha) C -- exact -- so the loads of C fly together with the first operand chunks instead of
  // sitting, latency exposed, between the last MFMA and the stores (K = 128 update at m = 8192: 378 -> 2xx us).
  bool c_in_acc = false;
  if constexpr (EPI == EPI_STORE && (YMODE == Y_NT || YMODE == Y_NN)) {
    c_in_acc = a.beta != 0.0 && (a.beta == a.alpha || a.beta == -a.alpha);
    if (c_in_acc) {
      const double sgn = (a.beta == a.alpha) ? 1.0 : -1.0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // branch-free (clamped address + select): under a per-element branch the compiler waits for every load on its own
            const int64_t row = row0 + wm * 64 + m * 16 + lg + 4 * r;
            const int64_t col = col0 + wn * 64 + n * 16 + lr;
            const bool ok = row < a.m_valid && col < a.n_valid;
            const double c = Cp[(row < a.m_valid ? row : a.m_valid - 1) * a.ldc + (col < a.n_valid ? col : a.n_valid - 1)];
            acc[m][n][r] = ok ? sgn * c : 0.0;
          }
    }
  }

  // ---- staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4) -------------------------------------------
  // No VGPR round trip and no ds_write: a wave instruction moves 64 x 16 B = 8 tile rows; the LDS image is lane-linear
  // (uniform base + lane*16), so the XOR slot swizzle is applied to the per-lane SOURCE address and again by the
  // fragment reads (same involution).  Measured on MI355X: the register path's ds_write_b128 traffic cost ~8 % of the
  // matrix-pipe time, the DMA path costs nothing but issue slots.
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto dma16 = [&](const void* src, double* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst_wave_uniform, 16, 0, 0);
  };
  constexpr int RPI = NT / 8;  // tile rows covered by one pass of the whole workgroup (8 lanes per 128-byte row)
  const int srow = tid >> 3;   // this thread's row within a pass
  const char* const Xgb = reinterpret_cast<const char*>(Xp);
  int64_t xsrc[XU];
#pragma unroll
  for (int i = 0; i < XU; ++i) {
    const int row = i * RPI + srow;
    const int64_t grow = row0 + row < 0 ? 0 : row0 + row;   // rows in front of a shifted first tile: any valid row (their waves idle)
    xsrc[i] = (grow * a.ldx + 2 * ((tid & 7) ^ swz(row))) * 8;
  }
  auto stage_x = [&](int64_t k0, int st) {
#pragma unroll
    for (int i = 0; i < XU; ++i) dma16(Xgb + xsrc[i] + k0 * 8, Xs + st * XBUF + (i * RPI + wave * 8) * XS);
  };

  // Y operand, memory modes
  int64_t ysrc[(YMODE == Y_NT) ? YU : 1];
  const char* const Ygb = (YMODE == Y_NT) ? reinterpret_cast<const char*>(Yp + col0 * a.ldy) : nullptr;
  if constexpr (YMODE == Y_NT) {
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int row = i * RPI + srow;
      ysrc[i] = ((int64_t)row * a.ldy + 2 * ((tid & 7) ^ swz(row))) * 8;
    }
  }

  // generator state (coordinate mode): this thread's output column q and its wave-uniform slice of the chunk
  double qx = 0., qy = 0., qz = 0.;
  const int gq = tid % TN;
  const int gsub = __builtin_amdgcn_readfirstlane(tid / TN);
  v2d yr[EPT / 2];
  if constexpr (YMODE == Y_GEN) {
    const int64_t q = a.gcol0 + col0 + gq;
    qx = a.px[q]; qy = a.py[q]; qz = a.pz[q];
  }

  // lattice-table mode: per staged 16-byte slot this thread serves (q-row, k-pair); all index arithmetic is 32-bit
  // integer VALU (v_sad_u32 / v_mul_u32_u24 / v_cndmask), which co-issues with the fp64 MFMA pipe (FP VALU does not).
  int tqy[YU], tqx[YU], tqz[YU], tsl[YU];
  int piy = 0, pix = 0, piz = 0;   // wave-uniform voxel of the staged chunk's first contraction index
  int64_t pk = kb;
  if constexpr (YMODE == Y_TAB) {
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int row = i * RPI + srow;
      int64_t q = a.gcol0 + col0 + row;
      if (q >= a.gN) q = a.gN - 1;
      tqz[i] = (int)(q % a.gnz);
      const int64_t t = q / a.gnz;
      tqx[i] = (int)(t % a.gnx);
      tqy[i] = (int)(t / a.gnx);
      tsl[i] = (tid & 7) ^ swz(row);
    }
    piz = (int)(kb % a.gnz);
    const int64_t tp = kb / a.gnz;
    pix = (int)(tp % a.gnx);
    piy = (int)(tp / a.gnx);
  }

  auto stage_y = [&](int64_t k0, int st) {
    if constexpr (YMODE == Y_NT) {
#pragma unroll
      for (int i = 0; i < YU; ++i) dma16(Ygb + ysrc[i] + k0 * 8, Ys + st * YBUF + (i * RPI + wave * 8) * XS);
    } else if constexpr (YMODE == Y_NN) {
      // [16 k-rows][TN] tile: one wave instruction = one 1-KiB k-row (TN = 128 columns), padded row stride in LDS
#pragma unroll
      for (int i = 0; i < YU; ++i) {
        const int kr = i * (NT / 64) + wave;
        dma16(Yp + (k0 + kr) * a.ldy + col0 + 2 * lane, Ys + st * YBUF + kr * YS_NN);
      }
    } else if constexpr (YMODE == Y_TAB) {
      // advance the uniform voxel position by (k0 - pk) in {0, 16}; at most one carry per axis because gnz >= 16
      const int adv = (int)(k0 - pk);
      pk = k0;
      piz += adv;
      int c = piz >= a.gnz ? 1 : 0;
      piz -= c ? a.gnz : 0;
      pix += c;
      c = pix >= a.gnx ? 1 : 0;
      pix -= c ? a.gnx : 0;
      piy += c;
      // the 16 contraction indices of the chunk span at most two z-columns: A = (piy,pix) and its successor B
      int bx = pix + 1, by = piy;
      if (bx >= a.gnx) { bx = 0; by += 1; }
      const int ay = piy < a.gny ? piy : a.gny - 1;   // padded contraction indices: any in-range entry (A column is 0)
      by = by < a.gny ? by : a.gny - 1;
      const char* const tb = reinterpret_cast<const char*>(a.table);
      const unsigned nz2 = 2u * (unsigned)a.gnz;
#pragma unroll
      for (int i = 0; i < YU; ++i) {
        int pz = piz + 2 * tsl[i];
        const bool carry = pz >= a.gnz;
        pz -= carry ? a.gnz : 0;
        const unsigned cy = carry ? by : ay, cx = carry ? bx : pix;
        const unsigned rowb = __umul24(__usad((unsigned)tqx[i], cx, __umul24(__usad((unsigned)tqy[i], cy, 0u), (unsigned)a.gnx)), nz2);
        // mirrored z axis: entry (dz + nz - 1), dz = z_p - z_q; the pair (p, p+1) is contiguous and ascending
        const unsigned idx = rowb + (unsigned)(pz - tqz[i] + a.gnz - 1);
        dma16(tb + (idx << 3), Ys + st * YBUF + (i * RPI + wave * 8) * XS);
      }
    }
  };
  auto gen_y = [&](int64_t k0) {  // coordinate generator (prologue form; the main loop interleaves it with the MFMAs)
    const int64_t p0 = k0 + gsub * EPT;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double d2 = sqdist3(a.px[p0 + e], a.py[p0 + e], a.pz[p0 + e], qx, qy, qz);
      yr[e >> 1][e & 1] = a.cov.scale * cov_eval<KID>(a.cov, d2);
    }
  };
  auto store_gen = [&](int st) {
    double* buf = Ys + st * YBUF;
#pragma unroll
    for (int e = 0; e < EPT / 2; ++e)
      *reinterpret_cast<v2d*>(buf + gq * XS + 2 * ((gsub * (EPT / 2) + e) ^ swz(gq))) = yr[e];
  };

  // ---- main loop ---------------------------------------------------------------------------------------
  // Three-stage LDS ring, one barrier per 16-deep chunk, fragments in two rotating half sets:
  //   iteration c:  LDS-DMA of chunk c+2 issued (lands before this iteration's barrier)
  //                 set1 <- LDS(chunk c,   k-half 1)      | 32 MFMAs on set0 (chunk c, k-half 0)
  //                 set0 <- LDS(chunk c+1, k-half 0)      | 32 MFMAs on set1
  //                 barrier
  // Chunk c+1 became visible at the barrier of iteration c-1, so its first fragments are already in registers when
  // the barrier of iteration c releases: the matrix pipe restarts without waiting for LDS.  The body is one basic
  // block (the tail re-stages the last chunk instead of branching).  In coordinate-generator mode (Y_GEN) the
  // covariances of chunk c+2 are computed between the MFMA k-steps and written with ds_write; FP VALU competes with
  // the fp64 MFMA pipe, which is why the lattice-table mode is the production path on regular grids.
  if (kb < ke) {
    const int xoff0 = 2 * ((2 * lg + 0) ^ swz(lr)), xoff1 = 2 * ((2 * lg + 1) ^ swz(lr));
    const double* const xfrag = Xs + (wm * 64 + lr) * XS;
    const double* const yfrag = (YMODE == Y_NN) ? (Ys + wn * 64 + lr) : (Ys + (wn * 64 + lr) * XS);
    auto read_half = [&](int stage, int h, v2d (&av)[4], v2d (&bv)[4]) {
      const double* xb = xfrag + stage * XBUF + (h ? xoff1 : xoff0);
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const v2d*>(xb + m * 16 * XS);
      if constexpr (YMODE == Y_NN) {
        const double* yb = yfrag + stage * YBUF + (4 * lg + 2 * h) * YS_NN;
#pragma unroll
        for (int n = 0; n < 4; ++n) { bv[n][0] = yb[n * 16]; bv[n][1] = yb[YS_NN + n * 16]; }
      } else {
        const double* yb = yfrag + stage * YBUF + (h ? xoff1 : xoff0);
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = *reinterpret_cast<const v2d*>(yb + n * 16 * XS);
      }
    };
    if constexpr (NSTG == 2) {
      // 4-wave workgroups (128-row tiles: Cholesky / L^-1 pieces and the small-k axis passes of the spectral route):
      // plain double buffer, all fragments of chunk c read after the barrier; two such workgroups share a CU and
      // cover each other's barrier / LDS latency.
      stage_x(kb, 0);
      if constexpr (YMODE == Y_GEN) { gen_y(kb); store_gen(0); } else { stage_y(kb, 0); }
      __syncthreads();
      int cur = 0;
      for (int64_t k0 = kb; k0 < ke; k0 += BK) {
        int64_t kn = k0 + BK;
        if (kn >= ke) kn = ke - BK;
        stage_x(kn, cur ^ 1);
        if constexpr (YMODE != Y_GEN) stage_y(kn, cur ^ 1);
        v2d a0[4], b0[4], a1[4], b1[4];
        read_half(cur, 0, a0, b0);
        read_half(cur, 1, a1, b1);
        if constexpr (YMODE == Y_GEN) gen_y(kn);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(t < 2 ? a0[m][t & 1] : a1[m][t & 1],
                                                               t < 2 ? b0[n][t & 1] : b1[n][t & 1], acc[m][n], 0, 0, 0);
        if constexpr (YMODE == Y_GEN) store_gen(cur ^ 1);
        __syncthreads();
        cur ^= 1;
      }
    } else {
    {
        const int64_t k1 = (kb + BK < ke) ? kb + BK : kb;
        stage_x(kb, 0);
        stage_x(k1, 1);
        if constexpr (YMODE == Y_GEN) {
          gen_y(kb); store_gen(0);
          gen_y(k1); store_gen(1);
        } else {
          stage_y(kb, 0);
          stage_y(k1, 1);
        }
      }
      __syncthreads();
      int s0 = 0, s1 = 1, s2 = 2;  // ring stages of chunk c, c+1, c+2
      // Per-wave end of the useful contraction range:
      //   * 64 rows entirely in the zero padding behind m_valid: nothing to do (kw = kb);
      //   * triangular X (L^-1 in geobo_posterior_reduce): row group g of the diagonal block is zero beyond column
      //     row0 + 64 (g + 1) -- 3.4 % of the executed flop of a full sweep at 64^3 were spent on those zeros.
      // Past kw the wave keeps staging its share of the operand tiles and keeps the barriers, but reads no fragments and
      // issues no MFMAs: the matrix pipe of its SIMD is left to the co-resident wave.
      int64_t kw = ke;
      if constexpr (YMODE != Y_GEN) {
        if ((a.tri & TRI_X_LOWER) && kw > row0 + 64 * (wm + 1)) kw = row0 + 64 * (wm + 1);
        if (row0 + wm * 64 >= a.m_valid || row0 + wm * 64 < 0 || kw < kb) kw = kb;
      }
      // The wave's last 64 contraction indices under a triangular X are its own diagonal block: 16-row group m of the wave is
      // zero beyond column 16 (m + 1) of that block, so chunk c of the four (kd + 16 c) only needs the groups m >= c.
      int64_t kd = kw;                                   // first index of the peeled diagonal chunks (kw: none)
      if constexpr (YMODE != Y_GEN) {
        if ((a.tri & TRI_X_LOWER) && kw == row0 + 64 * (wm + 1) && kw - 64 >= kb) kd = kw - 64;
      }
      if (kw > kb) {
      v2d a0[4], b0[4], a1[4], b1[4];
      read_half(0, 0, a0, b0);
      for (int64_t k0 = kb; k0 < kd; k0 += BK) {
        int64_t kn = k0 + 2 * BK;
        if (kn >= ke) kn = ke - BK;
        stage_x(kn, s2);
        if constexpr (YMODE != Y_GEN) stage_y(kn, s2);
        const int64_t p0 = kn + gsub * EPT;  // generator: wave-uniform -> scalar loads of the p coordinates
        read_half(s0, 1, a1, b1);
  #pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t == 2) read_half(s1, 0, a0, b0);   // next chunk's first half (after the MFMAs that still read a0/b0)
  #pragma unroll
          for (int m = 0; m < 4; ++m)
  #pragma unroll
            for (int n = 0; n < 4; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(t < 2 ? a0[m][t & 1] : a1[m][t & 1],
                                                               t < 2 ? b0[n][t & 1] : b1[n][t & 1], acc[m][n], 0, 0, 0);
          if constexpr (YMODE == Y_GEN) {
  #pragma unroll
            for (int e = t * (EPT / 4); e < (t + 1) * (EPT / 4); ++e) {
              const double d2 = sqdist3(a.px[p0 + e], a.py[p0 + e], a.pz[p0 + e], qx, qy, qz);
              yr[e >> 1][e & 1] = a.cov.scale * cov_eval<KID>(a.cov, d2);
            }
          }
        }
        if constexpr (YMODE != Y_GEN) {
          // Issue-order hints under the matrix pipe (integer VALU, SALU, VMEM and DS issue do not compete with v_mfma_f64
          // for the FP pipe): 64 x {1 MFMA, <= 4 others}.  Tuned on MI355X against tighter patterns (<= 2 / <= 3 others, DS
          // reads pinned to the first MFMAs of each half): this one keeps the LDS-DMA issue spread over the whole chunk.
  #pragma unroll
          for (int i = 0; i < 64; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x126, 4, 0);
          }
        }
        if constexpr (YMODE == Y_GEN && (KID <= COV_MATERN32_X)) {
          // 64 x { 1 MFMA, up to GEN_VALU VALU }: interleave the generator into the matrix-pipe shadow
          constexpr int GEN_VALU = (KID == COV_MATERN32_X) ? 7 : ((KID == COV_D2) ? 1 : 5);
  #pragma unroll
          for (int i = 0; i < 64; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, GEN_VALU * (EPT / 4), 0);
          }
        }
        if constexpr (YMODE == Y_GEN) store_gen(s2);
        __syncthreads();
        const int ts = s0; s0 = s1; s1 = s2; s2 = ts;
      }
      if constexpr (YMODE != Y_GEN) {
        if (kd < kw) {
          auto diag_chunk = [&](auto mlo, int64_t k0) {
            constexpr int MLO = decltype(mlo)::value;
            int64_t kn = k0 + 2 * BK;
            if (kn >= ke) kn = ke - BK;
            stage_x(kn, s2);
            stage_y(kn, s2);
            read_half(s0, 1, a1, b1);
  #pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (t == 2) read_half(s1, 0, a0, b0);
  #pragma unroll
              for (int m = MLO; m < 4; ++m)
  #pragma unroll
                for (int n = 0; n < 4; ++n)
                  acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(t < 2 ? a0[m][t & 1] : a1[m][t & 1],
                                                                   t < 2 ? b0[n][t & 1] : b1[n][t & 1], acc[m][n], 0, 0, 0);
            }
  #pragma unroll
            for (int i = 0; i < 16 * (4 - MLO); ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x126, 6, 0);
            }
            __syncthreads();
            const int ts = s0; s0 = s1; s1 = s2; s2 = ts;
          };
          diag_chunk(std::integral_constant<int, 0>{}, kd);
          diag_chunk(std::integral_constant<int, 1>{}, kd + BK);
          diag_chunk(std::integral_constant<int, 2>{}, kd + 2 * BK);
          diag_chunk(std::integral_constant<int, 3>{}, kd + 3 * BK);
        }
      }
      }
      if constexpr (YMODE != Y_GEN) {
        // idle tail: this wave's rows have no operand left, the workgroup's other waves still do
        for (int64_t k0 = kw; k0 < ke; k0 += BK) {
          int64_t kn = k0 + 2 * BK;
          if (kn >= ke) kn = ke - BK;
          stage_x(kn, s2);
          stage_y(kn, s2);
          __syncthreads();
          const int ts = s0; s0 = s1; s1 = s2; s2 = ts;
        }
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  if constexpr (EPI == EPI_STORE) {
    // tiles that lie inside the valid region (all but the last row / column of tiles): no per-element predicate
    if (row0 + TM <= a.m_valid && col0 + TN <= a.n_valid && (a.beta == 0.0 || c_in_acc)) {
      double* const cw = Cp + (row0 + wm * 64 + lg) * a.ldc + col0 + wn * 64 + lr;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) cw[(int64_t)(m * 16 + 4 * r) * a.ldc + n * 16] = a.alpha * acc[m][n][r];
      return;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + wm * 64 + m * 16 + lg + 4 * r;
          const int64_t col = col0 + wn * 64 + n * 16 + lr;
          if (row < a.m_valid && col < a.n_valid) {
            double v = a.alpha * acc[m][n][r];
            double* dst = Cp + row * a.ldc + col;
            if (a.beta != 0.0 && !c_in_acc) v += a.beta * (*dst);
            *dst = v;
          }
        }
  } else {
    // column reductions over this tile's rows: mu-part = sum_r V[r,c] u[r],  ss-part = sum_r V[r,c]^2
    double smu[4], sss[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { smu[n] = 0.; sss[n] = 0.; }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t ri = row0 + wm * 64 + m * 16 + lg + 4 * r;
        const double ur = a.u[ri < 0 ? 0 : ri];             // rows in front of a shifted first tile: acc = 0
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const double v = acc[m][n][r];
          smu[n] = __builtin_fma(v, ur, smu[n]);
          sss[n] = __builtin_fma(v, v, sss[n]);
        }
      }
    // wavefront shuffle tree over the four 16-lane row groups
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      smu[n] += __shfl_xor(smu[n], 16); smu[n] += __shfl_xor(smu[n], 32);
      sss[n] += __shfl_xor(sss[n], 16); sss[n] += __shfl_xor(sss[n], 32);
    }
    double* red = smem;  // [2][WM][TN]; the main loop's last barrier has already been passed by every wave
    if (lg == 0) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        red[(0 * WM + wm) * TN + wn * 64 + n * 16 + lr] = smu[n];
        red[(1 * WM + wm) * TN + wn * 64 + n * 16 + lr] = sss[n];
      }
    }
    __syncthreads();
    if (tid < TN) {
      double m_ = 0., s_ = 0.;
#pragma unroll
      for (int w = 0; w < WM; ++w) { m_ += red[(0 * WM + w) * TN + tid]; s_ += red[(1 * WM + w) * TN + tid]; }
      a.part_mu[(int64_t)bi * a.ncols + col0 + tid] = m_;
      a.part_ss[(int64_t)bi * a.ncols + col0 + tid] = s_;
    }
  }
}

// out[r, c] = sum_s part[s][r][c]  (fixed order -> deterministic), 16-byte accesses
__global__ void __launch_bounds__(256) sum_slices_kernel(const double* __restrict__ part, int splits, int64_t m, int64_t n,
                                                         double* __restrict__ out, int64_t ldo) {
  const int64_t n2 = n >> 1, total = m * n2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n2, c = (i - r * n2) * 2;
    v2d acc = *reinterpret_cast<const v2d*>(part + r * n + c);
    for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const v2d*>(part + (int64_t)s * m * n + r * n + c);
    *reinterpret_cast<v2d*>(out + r * ldo + c) = acc;
  }
}

// deterministic final pass of the posterior reduction
__global__ void posterior_finish_kernel(const double* part_mu, const double* part_ss, int nbi, int64_t ncols,
                                        double prior_var, double* mu, double* var) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double m = 0., s = 0.;
  for (int b = 0; b < nbi; ++b) { m += part_mu[(int64_t)b * ncols + c]; s += part_ss[(int64_t)b * ncols + c]; }
  mu[c] = m;
  var[c] = prior_var - s;
}

template <int WM, int WN, int YMODE, int EPI, int KID>
int launch(GemmArgs& a, hipStream_t st) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = sizeof(double) * lds_doubles<WM, WN, YMODE>();
  auto kern = gemm_f64_kernel<WM, WN, YMODE, EPI, KID>;
  // opt-in to > 64 KiB dynamic LDS, once per DEVICE (the attribute lives with the device's code object); idempotent
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  if (a.m_valid <= 0) a.m_valid = (int64_t)1 << 62;
  if (a.n_valid <= 0) a.n_valid = (int64_t)1 << 62;
  a.xcd_map = (a.nbi >= 8) ? 1 : 0;
  // supertiles help when both operands stream (measured: posterior_reduce 59 -> 66 TF/s); with lower_only skipping they
  // unbalance the tail (AkA 420 -> 499 ms), so triangular-output launches keep the row-per-XCD map
  if ((YMODE == Y_NT || YMODE == Y_NN) && !(a.tri & TRI_LOWER_ONLY) && a.nbi >= 4 && a.nbj >= 8) a.xcd_map = 2;
  a.ntiles = 0;
  if (a.batch <= 0) a.batch = 1;
  const int64_t items = (int64_t)a.nbi * a.nbj * a.batch;
  // up to a few dozen tiles per CU: the tail of the launch matters -> balanced item list (plain and lower_only NT/NN)
  // (also the triangular-operand merges of the L^-1 build: longest contraction ranges first, equal lengths together)
  if ((YMODE == Y_NT || YMODE == Y_NN) && EPI == EPI_STORE && a.batch <= 64 &&
      a.nbi * a.nbj >= ((a.tri & (TRI_X_LOWER | TRI_Y_LOWER)) ? 16 : 64) && items <= 65536 && a.nbi <= 4096 &&
      (a.tri & (TRI_X_LOWER | TRI_Y_LOWER)) != (TRI_X_LOWER | TRI_Y_LOWER)) {
    a.ntiles = (int)tile_items(a.nbi, a.nbj, 64 * WM, 64 * WN, a.tri);
    if (a.ntiles > 0) a.xcd_map = 3;
  }
  int nblocks = a.nbi * a.nbj;
  if (a.xcd_map == 3) {
    const int64_t groups = ((int64_t)a.ntiles * a.batch + 31) / 32;
    nblocks = (int)(8 * 32 * ((groups + 7) / 8));
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, st, a);
    return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
  }
  if (a.xcd_map == 1) nblocks = 8 * ((a.nbi + 7) / 8) * a.nbj;
  if (a.xcd_map == 2) {
    int nst = ((a.nbi + 3) / 4) * ((a.nbj + 7) / 8);
    if (a.tri & TRI_X_LOWER) nst = a.nbi * ((a.nbj + 31) / 32);
    nblocks = 8 * 32 * ((nst + 7) / 8);
  }
  hipLaunchKernelGGL(kern, dim3(nblocks, a.batch), dim3(NT), lds, st, a);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int YMODE, int EPI, int KID>
int launch_by_rows(GemmArgs& a, int64_t m, int64_t n, hipStream_t st, bool small_tiles = false) {
  if (n % 128) return GEOBO_E_ALIGN;
  a.nbj = (int)(n / 128);
  if (m % 256 == 0 && !small_tiles) { a.nbi = (int)(m / 256); return launch<4, 2, YMODE, EPI, KID>(a, st); }
  if (m % 128 == 0) { a.nbi = (int)(m / 128); return launch<2, 2, YMODE, EPI, KID>(a, st); }
  return GEOBO_E_ALIGN;
}

}  // namespace

extern "C" int64_t geobo_tile_order(int nbi, int nbj, int tm, int tn, int lower_only, int x_lower, int y_lower, int* out_host,
                                    int64_t capacity) {
  if (nbi <= 0 || nbj <= 0 || tm <= 0 || tn <= 0 || (x_lower && y_lower)) return GEOBO_E_ARG;
  const int tri = (lower_only ? TRI_LOWER_ONLY : 0) | (x_lower ? TRI_X_LOWER : 0) | (y_lower ? TRI_Y_LOWER : 0);
  const int64_t n = tile_items(nbi, nbj, tm, tn, tri);
  if (out_host)
    for (int64_t t = 0; t < n && t < capacity; ++t) {
      int bi, bj;
      tile_of_item((int)t, nbi, nbj, tm, tn, tri, bi, bj);
      out_host[t] = bi << 16 | bj;
    }
  return n;
}

extern "C" int geobo_ak_fused(int kernel_id, const double* A, int64_t Ms_pad, int64_t N_pad, int64_t lda,
                              const double* x, const double* y, const double* z, int64_t col0, int64_t ncols,
                              double l1, double l2, double w, double amp, double* AK, int64_t ldak, void* stream) {
  if (!A || !x || !y || !z || !AK) return GEOBO_E_ARG;
  if (Ms_pad % 128 || N_pad % BK || ncols % 128 || (lda & 1) || col0 < 0 || col0 + ncols > N_pad) return GEOBO_E_ALIGN;
  GemmArgs a{};
  a.X = A; a.ldx = lda; a.Y = nullptr; a.ldy = 0; a.C = AK; a.ldc = ldak; a.k = N_pad;
  a.alpha = 1.0; a.beta = 0.0; a.tri = 0;
  a.px = x; a.py = y; a.pz = z; a.gcol0 = col0;
  a.cov = make_cov(kernel_id, l1, l2, w, amp);
  hipStream_t st = (hipStream_t)stream;
#define GEOBO_FUSED(ID) return launch_by_rows<Y_GEN, EPI_STORE, ID>(a, Ms_pad, ncols, st)
  COV_DISPATCH(kernel_id, GEOBO_FUSED);
#undef GEOBO_FUSED
  return GEOBO_E_ARG;
}

extern "C" int geobo_ak_fused_grid(const double* A, int64_t Ms_pad, int64_t N_pad, int64_t lda, int nx, int ny, int nz,
                                   const double* table, int64_t col0, int64_t ncols, double* AK, int64_t ldak,
                                   void* stream) {
  if (!A || !table || !AK) return GEOBO_E_ARG;
  if (nx <= 0 || ny <= 0 || nz < BK || (nz & 1)) return GEOBO_E_UNSUPPORTED;  // one carry per chunk: nz >= 16; k-pairs: nz even
  if ((int64_t)nx * ny * nz * 16 >= (int64_t)1 << 32) return GEOBO_E_UNSUPPORTED;  // 32-bit byte offsets into the table
  const int64_t N = (int64_t)nx * ny * nz;
  if (Ms_pad % 128 || N_pad % BK || ncols % 128 || (lda & 1) || col0 < 0 || col0 + ncols > N_pad || N_pad < N) return GEOBO_E_ALIGN;
  GemmArgs a{};
  a.X = A; a.ldx = lda; a.Y = nullptr; a.ldy = 0; a.C = AK; a.ldc = ldak; a.k = N_pad;
  a.alpha = 1.0; a.beta = 0.0; a.tri = 0;
  a.gcol0 = col0; a.table = table; a.gnx = nx; a.gny = ny; a.gnz = nz; a.gN = N;
  return launch_by_rows<Y_TAB, EPI_STORE, COV_D2>(a, Ms_pad, ncols, (hipStream_t)stream);
}

extern "C" int geobo_gemm_nt(int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                             const double* Y, int64_t ldy, double beta, double* C, int64_t ldc, int lower_only,
                             int64_t m_valid, void* stream) {
  if (!X || !Y || !C) return GEOBO_E_ARG;
  if (k % BK || (ldx & 1) || (ldy & 1)) return GEOBO_E_ALIGN;
  GemmArgs a{};
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.C = C; a.ldc = ldc; a.k = k;
  a.alpha = alpha; a.beta = beta; a.tri = (lower_only & GEOBO_GEMM_LOWER_ONLY) ? TRI_LOWER_ONLY : 0;
  a.m_valid = m_valid;
  return launch_by_rows<Y_NT, EPI_STORE, COV_D2>(a, m, n, (hipStream_t)stream, (lower_only & GEOBO_GEMM_SMALL_TILES) != 0);
}

extern "C" int geobo_gemm_nt_splitk(int64_t m, int64_t n, int64_t k, int splits, const double* X, int64_t ldx, const double* Y,
                                    int64_t ldy, double* C, int64_t ldc, int lower_only, int64_t m_valid, void* ws,
                                    size_t ws_bytes, void* stream) {
  if (!X || !Y || !C || !ws || splits < 1 || splits > 64) return GEOBO_E_ARG;
  if (k % (BK * splits) || (ldx & 1) || (ldy & 1) || (ldc & 1) || (n & 1)) return GEOBO_E_ALIGN;
  if (ws_bytes < (size_t)splits * m * n * sizeof(double)) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, (size_t)splits * m * n * sizeof(double), st) != hipSuccess) return GEOBO_E_LAUNCH;
  GemmArgs a{};
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.C = (double*)ws; a.ldc = n; a.k = k / splits;
  a.alpha = 1.0; a.beta = 0.0; a.tri = lower_only ? TRI_LOWER_ONLY : 0;
  a.sXb = k / splits; a.sYb = k / splits; a.sCb = m * n; a.batch = splits;
  a.m_valid = m_valid;
  int rc = launch_by_rows<Y_NT, EPI_STORE, COV_D2>(a, m, n, st);
  if (rc) return rc;
  int64_t nblk = (m * (n / 2) + 255) / 256;
  if (nblk > 256 * 16) nblk = 256 * 16;
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const double*)ws, splits, m, n, C, ldc);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_gemm_nn(int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                             const double* Y, int64_t ldy, double beta, double* C, int64_t ldc, int x_lower,
                             int y_lower, void* stream) {
  if (!X || !Y || !C) return GEOBO_E_ARG;
  if (k % BK || (ldx & 1) || (ldy & 1)) return GEOBO_E_ALIGN;
  GemmArgs a{};
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.C = C; a.ldc = ldc; a.k = k;
  a.alpha = alpha; a.beta = beta; a.tri = ((x_lower & 1) ? TRI_X_LOWER : 0) | ((y_lower & 1) ? TRI_Y_LOWER : 0);
  return launch_by_rows<Y_NN, EPI_STORE, COV_D2>(a, m, n, (hipStream_t)stream, ((x_lower | y_lower) & GEOBO_GEMM_SMALL_TILES) != 0);
}

extern "C" int geobo_gemm_batched(int y_is_kn, int64_t m, int64_t n, int64_t k, double alpha, const double* X, int64_t ldx,
                                  int64_t strideX, const double* Y, int64_t ldy, int64_t strideY, double beta, double* C,
                                  int64_t ldc, int64_t strideC, int64_t m_valid, int64_t n_valid, int batch, void* stream) {
  if (!X || !Y || !C || batch <= 0 || batch > 65535) return GEOBO_E_ARG;
  if (k % BK || (ldx & 1) || (ldy & 1) || (strideX & 1) || (strideY & 1)) return GEOBO_E_ALIGN;
  GemmArgs a{};
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.C = C; a.ldc = ldc; a.k = k;
  a.alpha = alpha; a.beta = beta; a.tri = 0;
  a.sXb = strideX; a.sYb = strideY; a.sCb = strideC; a.m_valid = m_valid; a.n_valid = n_valid; a.batch = batch;
  if (y_is_kn) return launch_by_rows<Y_NN, EPI_STORE, COV_D2>(a, m, n, (hipStream_t)stream);
  return launch_by_rows<Y_NT, EPI_STORE, COV_D2>(a, m, n, (hipStream_t)stream);
}

extern "C" size_t geobo_posterior_ws_bytes(int64_t m, int64_t ncols) {
  return (size_t)2 * (size_t)((m + 127) / 128) * (size_t)ncols * sizeof(double);
}

extern "C" int geobo_posterior_reduce(int64_t m, int64_t ncols, const double* Linv, int64_t ldi, const double* AK,
                                      int64_t ldak, const double* u, double prior_var, double* mu, double* var,
                                      int64_t m_valid, void* ws, size_t ws_bytes, void* stream) {
  if (!Linv || !AK || !u || !mu || !var || !ws) return GEOBO_E_ARG;
  if (m % 128 || ncols % 128 || (ldi & 1) || (ldak & 1)) return GEOBO_E_ALIGN;
  if (ws_bytes < geobo_posterior_ws_bytes(m, ncols)) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.X = Linv; a.ldx = ldi; a.Y = AK; a.ldy = ldak; a.C = nullptr; a.ldc = 0; a.k = m;
  a.alpha = 1.0; a.beta = 0.0; a.tri = TRI_X_LOWER;
  a.u = u; a.ncols = ncols;
  a.m_valid = (m_valid > 0 && m_valid < m) ? m_valid : m;
  const int nbi_max = (int)((m + 127) / 128);
  a.part_mu = (double*)ws;
  a.part_ss = (double*)ws + (size_t)nbi_max * ncols;
  // 256-row tiles of four 64-row wavefront groups, aligned to the END of the valid rows: a row count that does not fill its
  // last tile would leave that tile -- the one with the LONGEST contraction -- to a quarter of the workgroup (64^3 with 50
  // drill rows: 3 % of the launch for 0.6 % of the flop).  Shifted, the under-filled tile is the first one (contraction <= 192).
  const int64_t groups = (a.m_valid + 63) / 64;
  a.row_shift = 64 * ((4 - groups % 4) % 4);
  a.nbi = (int)((groups * 64 + a.row_shift) / 256);
  a.nbj = (int)(ncols / 128);
  int rc = launch<4, 2, Y_NN, EPI_REDUCE, COV_D2>(a, st);
  if (rc) return rc;
  hipLaunchKernelGGL(posterior_finish_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, st, a.part_mu,
                     a.part_ss, a.nbi, ncols, prior_var, mu, var);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
