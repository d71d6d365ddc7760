// gemm_fold.hip -- radix-2 ("folded") form of the batched axis passes of the spectral route, for every extent that has no fused
// two-axis kernel (xz2d_fold.hip exists for 64 x 64 planes only): the (x, z) transforms of A K = A_s K_sj and of V = (L^-1 A) K
// (kernels.py:158-195, inversion.py:96,114-117 re-associated: DESIGN.md section 2), the lattice Gram's and the lattice convolution's
// transforms.  Replaces geobo_gemm_batched launches whose matrix operand is the real eigenvector basis G (P x n, P = 2n) of
// spectral.forward_matrix or its transpose: the basis is PAIR-INTERLEAVED, row 2b+1 = (-1)^i row 2b, so
//     analysis (n -> P):   out[2b] = E_b + O_b,  out[2b+1] = E_b - O_b,   E_b / O_b = sum over the even / odd inputs of G[2b][i] x[i]
//     synthesis (P -> n):  out[i]  = sum_b G[2b][i] (s[2b] + (-1)^i s[2b+1])
// -- half the multiply-adds of the plain product (what xz2d_fold.hip does inside one 64 x 64 plane, here as a GEMM of any extent that
// is a multiple of 16).  The operands are EXACTLY those of the geobo_gemm_batched call it replaces (same matrices, same strides):
//   * analysis: the MFMA step t of a 16-deep chunk contracts k = {t, 4+t, 8+t, 12+t} -- steps 0, 2 see even inputs only, steps 1, 3
//     odd ones -- so E and O are two accumulator sets fed from the rows 2b of G alone (the staging reads every second row of the
//     matrix), at half the tile width; the butterfly is the epilogue (analysis along the contiguous axis: one 16-byte store per pair);
//   * synthesis: a 16-byte fragment of the data holds a pair (s[2b], s[2b+1]); its sum feeds the tiles of the even outputs, its
//     difference those of the odd ones (the staging permutes the output rows of a 128-tile: even ones to the first 64 tile rows, so
//     that the choice is wave-uniform), against the even columns of G^T alone: one MFMA per pair instead of two.
// Same 2 x 2 waves of 64 x 64, LDS-DMA staging and XOR slot swizzle as gemm_f64.hip; two 66 KB workgroups per CU.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "geobo_hip.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16, XS = 16, TM = 128, TN = 128;
constexpr int XBUF = TM * XS;                 // [128 rows][16 k], 128-byte rows, XOR-swizzled 16-byte slots
constexpr int YS_NN = TN + 4;                 // row stride of a [16 k][128 cols] chunk
constexpr int YBUF = BK * YS_NN;              // >= TN * XS
constexpr int STAGE = XBUF + YBUF;
constexpr size_t LDS_BYTES = 2 * STAGE * sizeof(double);      // 66 560 B

// FWD / INV: analysis / synthesis; _Z: the data is X (m x k), the matrix Y (n x k); _X: the matrix is X (m x k), the data Y (k x n);
// INV_XT: synthesis with the matrix as X and the data as Y in the n x k layout (C = G^T S^T: the last pass of the lattice convolution)
enum { FWD_Z = 0, FWD_X = 1, INV_Z = 2, INV_X = 3, INV_XT = 4 };

struct FoldArgs {
  const double* X; int64_t ldx, sX;
  const double* Y; int64_t ldy, sY;
  double* C; int64_t ldc, sC;
  int64_t k, m_valid, n_valid;
  int nbi, nbj;
  // FWD_X with the lattice Gram's x-step epilogue (LAM): C[batch][o] = sum_z (G X_batch)[o][z] * lam[(batch0 + batch) % planes][o][z]
  const double* lam; int planes; int64_t batch0;
};

__device__ __forceinline__ int swz(int row) {
  const int p = (row >> 1) & 7;
  return (p & 1) | (((p >> 2) & 1) * 6);
}

template <int MODE, bool LAM = false>
__global__ void __launch_bounds__(256, 2) gemm_fold_kernel(const FoldArgs a) {
  static_assert(!LAM || MODE == FWD_X, "the eigenvalue / channel-sum epilogue belongs to the analysis along a strided axis");
  constexpr bool NN = MODE == FWD_X || MODE == INV_X, INV = MODE >= INV_Z, BX = MODE == INV_X || MODE == INV_XT;
  constexpr int XROWS = (MODE == FWD_X) ? 64 : 128;           // rows of the X tile in LDS (analysis, matrix on the X side: base rows only)
  constexpr int YROWS = (MODE == FWD_Z) ? 64 : 128;           // rows of an NT Y tile
  constexpr int MT = (MODE == FWD_X) ? 2 : 4, NTL = (MODE == FWD_Z) ? 2 : 4;   // 16-row / 16-column tiles per wave
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4, srow = tid >> 3;
  const int bi = blockIdx.x % a.nbi, bj = blockIdx.x / a.nbi;
  const int64_t row0 = (int64_t)bi * TM, col0 = (int64_t)bj * TN;
  const char* const Xb = reinterpret_cast<const char*>(a.X + (int64_t)blockIdx.y * a.sX);
  const char* const Yb = reinterpret_cast<const char*>(a.Y + (int64_t)blockIdx.y * a.sY);
  double* const Cp = a.C + (int64_t)blockIdx.y * a.sC;

  // ---- staging sources: LDS tile row r' <- memory row --------------------------------------------------------------------------
  //   data (X of the NT modes): row0 + r';   analysis matrix: every second row, 2 (64 tile + r');   synthesis matrix: the 128 rows of
  //   the tile with the even ones first, tile0 + 2 (r' & 63) + (r' >> 6)
  int64_t xsrc[XROWS / 32], ysrc[4];
#pragma unroll
  for (int i = 0; i < XROWS / 32; ++i) {
    const int r = i * 32 + srow;
    const int64_t g = (MODE == FWD_X) ? 2 * ((int64_t)bi * 64 + r) : BX ? row0 + 2 * (r & 63) + (r >> 6) : row0 + r;
    xsrc[i] = (g * a.ldx + 2 * ((tid & 7) ^ swz(r))) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (NN) {
      ysrc[i] = ((int64_t)(i * 4 + wave) * a.ldy + col0 + 2 * lane) * 8;            // one 1-KiB k-row per wave instruction
    } else {
      const int r = i * 32 + srow;
      const int64_t g = (MODE == FWD_Z) ? 2 * ((int64_t)bj * 64 + r) : (MODE == INV_Z) ? col0 + 2 * (r & 63) + (r >> 6) : col0 + r;
      ysrc[i] = (g * a.ldy + 2 * ((tid & 7) ^ swz(r))) * 8;
    }
  }
  auto stage = [&](int64_t k0, int st) {
    double* const xs = smem + st * STAGE;
    double* const ys = xs + XBUF;
#pragma unroll
    for (int i = 0; i < XROWS / 32; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(Xb + xsrc[i] + k0 * 8), (lds_ptr_t)(xs + (i * 32 + wave * 8) * XS), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < (NN ? 4 : YROWS / 32); ++i) {
      if constexpr (NN)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(Yb + ysrc[i] + k0 * a.ldy * 8), (lds_ptr_t)(ys + (i * 4 + wave) * YS_NN), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(Yb + ysrc[i] + k0 * 8), (lds_ptr_t)(ys + (i * 32 + wave * 8) * XS), 16, 0, 0);
    }
  };

  // ---- accumulators ------------------------------------------------------------------------------------------------------------
  // analysis: E / O over the base rows (FWD_Z: 4 x 2 tiles, FWD_X: 2 x 4); synthesis: one set of 4 x 4
  v4d acc[2][4][2];          // [E | O][.][.] for the analysis modes; viewed as [m][n] = acc[m >> 1][..] below for the synthesis modes
  v4d accs[4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int l = 0; l < 2; ++l) acc[i][j][l] = (v4d){0., 0., 0., 0.};
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) accs[m][n] = (v4d){0., 0., 0., 0.};

  const int xo0 = 2 * ((2 * lg + 0) ^ swz(lr)), xo1 = 2 * ((2 * lg + 1) ^ swz(lr));
  const int xbase = (MODE == FWD_X ? wm * 32 : wm * 64) + lr;
  const int ybase = NN ? (wn * 64 + lr) : ((MODE == FWD_Z ? wn * 32 : wn * 64) + lr) * XS;
  const double sgnz = (MODE == INV_Z && wn) ? -1.0 : 1.0, sgnx = (BX && wm) ? -1.0 : 1.0;

  // 16-row / 16-column tiles of this wave that hold no valid output are skipped (wave-uniform masks: extents that are not multiples of
  // the 128 x 128 tile -- P = 192 at n = 96 -- otherwise spend a quarter of their MFMAs on padding)
  unsigned mmask = 0, nmask = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int64_t r = (MODE == FWD_X) ? 2 * ((int64_t)bi * 64 + wm * 32 + m * 16) : BX ? row0 + 32 * m : row0 + wm * 64 + m * 16;
    mmask |= (unsigned)(r < a.m_valid && (MODE != FWD_X || m < 2)) << m;
  }
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int64_t c = (MODE == FWD_Z) ? 2 * ((int64_t)bj * 64 + wn * 32 + n * 16) : (MODE == INV_Z) ? col0 + 32 * n : col0 + wn * 64 + n * 16;
    nmask |= (unsigned)(c < a.n_valid && (MODE != FWD_Z || n < 2)) << n;
  }
  mmask = __builtin_amdgcn_readfirstlane(mmask);
  nmask = __builtin_amdgcn_readfirstlane(nmask);
  const bool full = (mmask == (MODE == FWD_X ? 3u : 15u)) && (nmask == (MODE == FWD_Z ? 3u : 15u));

  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.k; k0 += BK) {
    const int64_t kn = (k0 + BK < a.k) ? k0 + BK : k0;        // (the last iteration re-stages its own chunk: no branch around the DMA)
    stage(kn, cur ^ 1);
    const double* const xs = smem + cur * STAGE + xbase * XS;
    const double* const ys = smem + cur * STAGE + XBUF + ybase;
    v2d av[2][MT], bv[2][NTL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MT; ++m) av[h][m] = *reinterpret_cast<const v2d*>(xs + m * 16 * XS + (h ? xo1 : xo0));
      if constexpr (NN) {
        const double* yb = ys + (4 * lg + 2 * h) * YS_NN;
#pragma unroll
        for (int n = 0; n < NTL; ++n) { bv[h][n][0] = yb[n * 16]; bv[h][n][1] = yb[YS_NN + n * 16]; }
      } else {
#pragma unroll
        for (int n = 0; n < NTL; ++n) bv[h][n] = *reinterpret_cast<const v2d*>(ys + n * 16 * XS + (h ? xo1 : xo0));
      }
    }
    if constexpr (!INV) {
      // k = 4 lg + 2 h + p: parity p selects E (even inputs) or O
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
              if (!full && !(((mmask >> m) & 1) && ((nmask >> n) & 1))) continue;
              v4d& c = (MODE == FWD_Z) ? acc[p][m][n] : acc[p][n][m];
              c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][m][p], bv[h][n][p], c, 0, 0, 0);
            }
    } else if constexpr (MODE == INV_Z) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double uv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) uv[m] = __builtin_fma(sgnz, av[h][m][1], av[h][m][0]);     // s[2b] +- s[2b+1]
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (!full && !(((mmask >> m) & 1) && ((nmask >> n) & 1))) continue;
            accs[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv[m], bv[h][n][0], accs[m][n], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double uv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) uv[n] = __builtin_fma(sgnx, bv[h][n][1], bv[h][n][0]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (!full && !(((mmask >> m) & 1) && ((nmask >> n) & 1))) continue;
            accs[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][m][0], uv[n], accs[m][n], 0, 0, 0);
          }
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue (D layout: col = lane & 15, row = (lane >> 4) + 4 reg) -----------------------------------------------------------
  if constexpr (MODE == FWD_Z) {
    // pairs along the contiguous axis: out[row][2b], out[row][2b+1] = E +- O as one 16-byte store
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + wm * 64 + m * 16 + lg + 4 * r;
          const int64_t col = 2 * ((int64_t)bj * 64 + wn * 32 + n * 16 + lr);
          if (row < a.m_valid && col < a.n_valid) {
            const double e = acc[0][m][n][r], o = acc[1][m][n][r];
            *reinterpret_cast<v2d*>(Cp + row * a.ldc + col) = (v2d){e + o, e - o};
          }
        }
  } else if constexpr (MODE == FWD_X && LAM) {
    // x step of the lattice Gram (AkA = (A K) A^T on a lattice survey, inversion.py:96) without its intermediate: the tile holds
    // D[o][z] = (G_x X)[o][z] for 128 spectral rows o and ALL channels z (one column tile); scale by the stencil's eigen-data, sum over
    // z -- 4 column tiles per lane, 16 lanes by shuffles, the two column halves of the tile through LDS -- and store 128 sums.  The
    // batched form wrote D (Px nz doubles per (row, ky) plane: 67 MB per row at 128^3) and read it back in geobo_lamdot_z.
    const double* const lamp = a.lam + ((a.batch0 + blockIdx.y) % a.planes) * (a.m_valid * a.n_valid);
    double* const red = smem;          // [2 column halves][128 rows]; the loop's last barrier has been passed by every wave
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 2 * (wm * 32 + m * 16 + lg + 4 * r);            // local row of the pair's first output
        const int64_t row = (int64_t)bi * 128 + rl;
        double pe = 0.0, po = 0.0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int64_t col = wn * 64 + n * 16 + lr;
          const bool ok = row < a.m_valid && col < a.n_valid;
          const int64_t ri = ok ? row : 0, ci = ok ? col : 0;
          const double le = lamp[ri * a.n_valid + ci], lo = lamp[(ri + 1) * a.n_valid + ci];
          const double e = acc[0][n][m][r], o = acc[1][n][m][r];
          pe = __builtin_fma(ok ? e + o : 0.0, le, pe);
          po = __builtin_fma(ok ? e - o : 0.0, lo, po);
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { pe += __shfl_xor(pe, off); po += __shfl_xor(po, off); }
        if (lr == 0) { red[wn * 128 + rl] = pe; red[wn * 128 + rl + 1] = po; }
      }
    __syncthreads();
    if (tid < 128) {
      const int64_t row = (int64_t)bi * 128 + tid;
      if (row < a.m_valid) Cp[row] = red[tid] + red[128 + tid];
    }
  } else if constexpr (MODE == FWD_X) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = 2 * ((int64_t)bi * 64 + wm * 32 + m * 16 + lg + 4 * r);
          const int64_t col = col0 + wn * 64 + n * 16 + lr;
          if (row < a.m_valid && col < a.n_valid) {
            const double e = acc[0][n][m][r], o = acc[1][n][m][r];
            Cp[row * a.ldc + col] = e + o;
            Cp[(row + 1) * a.ldc + col] = e - o;
          }
        }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = BX ? row0 + 2 * (m * 16 + lg + 4 * r) + wm : row0 + wm * 64 + m * 16 + lg + 4 * r;
          const int64_t col = (MODE == INV_Z) ? col0 + 2 * (n * 16 + lr) + wn : col0 + wn * 64 + n * 16 + lr;
          if (row < a.m_valid && col < a.n_valid) Cp[row * a.ldc + col] = accs[m][n][r];
        }
  }
}

template <int MODE, bool LAM = false>
int launch_fold(const FoldArgs& a, int64_t batch, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};     // opt-in to > 64 KiB dynamic LDS, once per DEVICE (as in gemm_f64.hip)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fold_kernel<MODE, LAM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_BYTES) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_fold_kernel<MODE, LAM>), dim3((unsigned)(a.nbi * a.nbj), (unsigned)batch), dim3(256), LDS_BYTES, st, a);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

}  // namespace

extern "C" int geobo_gemm_fold(int y_is_kn, int inverse, int64_t m, int64_t n, int64_t k, const double* X, int64_t ldx, int64_t strideX,
                               const double* Y, int64_t ldy, int64_t strideY, double* C, int64_t ldc, int64_t strideC, int64_t m_valid,
                               int64_t n_valid, int64_t batch, void* stream) {
  if (!X || !Y || !C) return GEOBO_E_ARG;
  if (m <= 0 || n <= 0 || k <= 0 || batch <= 0 || batch > 65535) return GEOBO_E_ARG;
  if (m % 128 || n % 128 || k % 16 || (ldx & 1) || (ldy & 1) || (strideX & 1) || (strideY & 1)) return GEOBO_E_ALIGN;
  if (!y_is_kn && !inverse && ((ldc & 1) || (strideC & 1))) return GEOBO_E_ALIGN;        // 16-byte pair stores
  if (m_valid <= 0) m_valid = m;
  if (n_valid <= 0) n_valid = n;
  // the pair partner of a valid output is valid too (P = 2 n_in is even; output extents of the synthesis are unconstrained)
  if (!inverse && ((y_is_kn ? m_valid : n_valid) & 1)) return GEOBO_E_ALIGN;
  FoldArgs a;
  a.X = X; a.ldx = ldx; a.sX = strideX;
  a.Y = Y; a.ldy = ldy; a.sY = strideY;
  a.C = C; a.ldc = ldc; a.sC = strideC;
  a.k = k; a.m_valid = m_valid; a.n_valid = n_valid;
  a.nbi = (int)(m / 128); a.nbj = (int)(n / 128);
  a.lam = nullptr; a.planes = 1; a.batch0 = 0;
  hipStream_t st = (hipStream_t)stream;
  if (inverse == 2) return y_is_kn ? GEOBO_E_UNSUPPORTED : launch_fold<INV_XT>(a, batch, st);
  switch ((y_is_kn ? 1 : 0) | (inverse ? 2 : 0)) {
    case FWD_Z: return launch_fold<FWD_Z>(a, batch, st);
    case FWD_X: return launch_fold<FWD_X>(a, batch, st);
    case INV_Z: return launch_fold<INV_Z>(a, batch, st);
    default: return launch_fold<INV_X>(a, batch, st);
  }
}

extern "C" int geobo_gemm_fold_lamdot(int64_t px, int64_t nz, int64_t k, const double* G, int64_t ldg, const double* Y, int64_t ldy,
                                      int64_t strideY, const double* lam, int planes, int64_t batch0, double* out, int64_t batch,
                                      void* stream) {
  if (!G || !Y || !lam || !out) return GEOBO_E_ARG;
  if (px <= 0 || nz <= 0 || k <= 0 || planes <= 0 || batch0 < 0 || batch <= 0 || batch > 65535) return GEOBO_E_ARG;
  if (nz > 128) return GEOBO_E_UNSUPPORTED;                   // one column tile holds every channel
  if ((px & 1) || k % 16 || (ldg & 1) || (ldy & 1) || (strideY & 1)) return GEOBO_E_ALIGN;
  FoldArgs a;
  a.X = G; a.ldx = ldg; a.sX = 0;
  a.Y = Y; a.ldy = ldy; a.sY = strideY;
  a.C = out; a.ldc = 0; a.sC = px;
  a.k = k; a.m_valid = px; a.n_valid = nz;
  a.nbi = (int)((px + 127) / 128); a.nbj = 1;
  a.lam = lam; a.planes = planes; a.batch0 = batch0;
  return launch_fold<FWD_X, true>(a, batch, (hipStream_t)stream);
}
