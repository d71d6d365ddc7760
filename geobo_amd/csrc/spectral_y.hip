// spectral_y.hip -- the y-axis stage of the structured covariance product as an IN-KERNEL spectral product on the fp64 matrix pipe
// (round 6).  Same operation as toeplitz.hip (kernels.py:158-195 builds K_sj densely; inversion.py:96,114 contract it):
//
//     out_j[r][y][c] = sum_{y'} t_{j,c}(|y - y'|) in[r][y'][c]        per mode c of the (x, z)-spectrum, j = property block
//
// toeplitz.hip applies T_c directly on the vector pipe (ny^2 multiply-adds per mode, block and term: the matrix differs per mode, so
// no operand is shared along a tile edge).  Here the y axis goes through its OWN spectrum without leaving the registers:
//
//   * T_c (ny x ny, symmetric Toeplitz) is the leading block of a SKEW-circulant of size P = 2 ny (first column t_0 .. t_{ny-1}, *,
//     -t_{ny-1} .. -t_1), which the real functions cos / sin(2 pi kappa y / P) of the HALF-INTEGER frequencies kappa = 1/2 .. ny - 1/2
//     diagonalise with the real eigenvalues  lambda_c(kappa) = (1/ny) sum_d w_d t_c(d) cos(2 pi kappa d / P)  (w_0 = 1, else 2).
//     Half-integer frequencies have no self-paired members (no constant / alternating / middle rows): all ny frequencies fall into
//     ny / 4 ORBITS {kappa, ny - kappa, ny/2 + kappa, ny/2 - kappa}, kappa = omega + 1/2, omega < ny/4, of the shift by a quarter
//     period.  On the inputs y = 4 j + rho of one residue class the cos / sin rows of the whole orbit are +-(cos | sin) of kappa:
//     RADIX 4 -- the analysis is, per class, a (2 ny/4) x (ny/4) matrix product against G_rho (the SAME matrix for every mode:
//     the 16 columns of an MFMA are 16 modes, 128-byte segments of the spectrum), then 16 additions per orbit; the synthesis
//     is the transpose.  ny^2 / 2 multiply-adds per transform and mode.
//   * one term, two blocks: 1 analysis + 2 syntheses = 1.5 ny^2 (direct: 2 ny^2); two-term rows with the shared cross block:
//     2 + 2 = 2 ny^2 (three direct products: 3 ny^2) -- the terms meet in the spectrum, lambda_00 x^_g + lambda_01 x^_m.
//   * a wave owns 16 modes and sweeps the rows: its 16 eigenvalues per lane and table stay in registers (computed in the prologue
//     from the SAME Toeplitz generator tables toeplitz.hip takes: a drop-in), the transform fragments sit in LDS (shared by the
//     workgroup), the inputs come straight from global memory into the B operand layout (lane = (y' mod 16 / 4, mode): four
//     128-byte segments per instruction, one row ahead), the D tiles of the synthesis go straight back (the same segments).
//     No per-lane table, no workgroup barrier in the row loop.
//
// Fragment / register layout (v_mfma_f64_16x16x4: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15]); g = lane >> 4, c = lane & 15:
//   analysis tile T of class rho: rows m < 8 = cos of omega = 8 T + m, rows m >= 8 = sin of omega = 8 T + m - 8, so that a lane
//     holds C_rho (reg h) and S_rho (reg 2 + h) of ITS orbits omega = 8 T + 4 h + g, h = 0, 1: the butterflies are lane local;
//   k-step s of the analysis contracts j = 4 s + g, i.e. input planes y' = 16 s + 4 g + rho;
//   synthesis of class rho: k-step (T, r) takes register r of tile T as its B operand (k <-> g): (trig = r >> 1, omega = 8 T + 4 (r & 1) + g);
//     D register r' holds output j = 16 mj + g + 4 r', y = 4 j + rho;
//   eigenvalues: tile tau = 2 T + h, register f = member of the orbit: one MFMA sweep over the table per block in the prologue.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(*static_cast<rsrc_t*>(nullptr), 0, 0, 0));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_double;

__device__ __forceinline__ rsrc_t make_rsrc(const double* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
}
// Cache policy of the row streams (aux bit 1 = nt): every spectrum is read once and written once per launch, GBs of it -- nothing to
// keep in L2.  Measured on the 64^3 batch shape (profiles/r06_spectral_y_ab.txt): two blocks 1.282 -> 1.250 ms, two-term rows
// 1.875 -> 1.851 ms with non-temporal loads and stores.  The eigenvalue tables and the fragment blob are read with the default policy.
#ifndef SY_LD_AUX
#define SY_LD_AUX 2
#endif
#ifndef SY_ST_AUX
#define SY_ST_AUX 2
#endif
__device__ __forceinline__ double ld_lane(rsrc_t rs, unsigned voff, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
__device__ __forceinline__ double ld_stream(rsrc_t rs, unsigned voff, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, SY_LD_AUX));
}
__device__ __forceinline__ void st_lane(rsrc_t rs, unsigned voff, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, SY_ST_AUX);
}

template <int NY>
struct Shape {
  static constexpr int P = 2 * NY, NW = NY / 4, NJ = NY / 4, NT = (NW + 7) / 8, MJ = (NJ + 15) / 16, KS = NJ / 4, KE = NY / 4;
  static constexpr int NF_FWD = 4 * KS * NT, NF_INV = 4 * NT * 4 * MJ, NF_EIG = 2 * NT * KE;
  static constexpr int NF = NF_FWD + NF_INV + NF_EIG;                 // fragments of 64 doubles
  static_assert(NY % 16 == 0, "ny");
  // fragment indices
  static constexpr int fwd(int rho, int s, int T) { return (rho * KS + s) * NT + T; }
  static constexpr int inv(int rho, int T, int r, int mj) { return NF_FWD + ((rho * NT + T) * 4 + r) * MJ + mj; }
  static constexpr int eig(int tau, int s) { return NF_FWD + NF_INV + tau * KE + s; }
};

// ---- the basis blob: every fragment of the three transforms in lane order (filled once per ny by geobo_spectral_y_basis) ---------------
template <int NY>
__global__ void basis_kernel(double* out) {
  using S = Shape<NY>;
  const int f = blockIdx.x, lane = threadIdx.x, m = lane & 15, g = lane >> 4;
  auto trig = [](int t, int om, int y) {      // cos / sin(2 pi (om + 1/2) y / P) with the phase reduced in integers
    const int idx = ((2 * om + 1) * y) % (2 * S::P);
    double s, c;
    sincospi((double)idx / (double)S::P, &s, &c);
    return t == 0 ? c : s;
  };
  double v = 0.0;
  if (f < S::NF_FWD) {
    const int T = f % S::NT, s = (f / S::NT) % S::KS, rho = f / (S::NT * S::KS);
    const int y = 4 * (4 * s + g) + rho, om = 8 * T + (m & 7);
    if (om < S::NW) v = trig(m >> 3, om, y);
  } else if (f < S::NF_FWD + S::NF_INV) {
    const int q = f - S::NF_FWD, mj = q % S::MJ, r = (q / S::MJ) % 4, T = (q / (4 * S::MJ)) % S::NT, rho = q / (4 * S::MJ * S::NT);
    const int j = 16 * mj + m, om = 8 * T + 4 * (r & 1) + g;
    if (om < S::NW && j < S::NJ) v = trig(r >> 1, om, 4 * j + rho);
  } else {
    const int q = f - S::NF_FWD - S::NF_INV, s = q % S::KE, tau = q / S::KE;
    const int T = tau >> 1, h = tau & 1, fm = m >> 2, gg = m & 3;        // row m of tile tau = 4 f + g': member f of the orbit of lane group g'
    const int om = 8 * T + 4 * h + gg, d = 4 * s + g;
    if (om < S::NW) {
      const int f2 = fm == 0 ? 2 * om + 1 : fm == 1 ? 2 * NY - (2 * om + 1) : fm == 2 ? NY + 2 * om + 1 : NY - (2 * om + 1);   // twice the frequency
      const int idx = (int)(((int64_t)f2 * d) % (2 * S::P));
      double sn, cs;
      sincospi((double)idx / (double)S::P, &sn, &cs);
      v = (d == 0 ? 1.0 : 2.0) * cs / (double)NY;
    }
  }
  out[(size_t)f * 64 + lane] = v;
}

struct SYArgs {
  const double* in[2];      // [R][NY][S] per term
  const double* tab[3];     // Toeplitz generators [NY][C]: NIN = 1: one per output block; NIN = 2: D0, X, D1 (geobo_toeplitz_y2s)
  double* out[2];           // [R][y1 - y0][S] per property block
  const double* basis;      // Shape<NY>::NF fragments
  int64_t C, S, R;
  int y0, y1;
};

// one orbit's analysis butterfly: class sums C[rho], S[rho] -> (a_f, b_f), f = kappa, ny - kappa, ny/2 + kappa, ny/2 - kappa
__device__ __forceinline__ void bfly_fwd(const double (&C)[4], const double (&S)[4], double (&a)[4], double (&b)[4]) {
  const double p0 = C[0] + C[2], p1 = C[0] - C[2], p2 = C[1] + C[3], p3 = C[1] - C[3];
  const double q0 = S[0] + S[2], q1 = S[0] - S[2], q2 = S[1] + S[3], q3 = S[1] - S[3];
  a[0] = p0 + p2; a[1] = p0 - p2; b[0] = q0 + q2; b[1] = q2 - q0;
  a[2] = p1 - q3; a[3] = p1 + q3; b[2] = q1 + p3; b[3] = p3 - q1;
}

template <int NY, int NIN, int NOUT, bool FULL>
__global__ void __launch_bounds__(256, 1) spectral_y_kernel(SYArgs g) {
  using S = Shape<NY>;
  constexpr int NT = S::NT, MJ = S::MJ, KS = S::KS, NTAB = NIN == 2 ? 3 : NOUT;
  static_assert((NIN == 1 && NOUT >= 1 && NOUT <= 2) || (NIN == 2 && NOUT == 2), "forms");
  extern __shared__ __attribute__((aligned(16))) double frag[];           // [NF][64]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, gq = lane >> 4;
  for (int i = threadIdx.x; i < S::NF * 64; i += 256) frag[i] = g.basis[i];
  __syncthreads();
  lds_double* const fl0 = (lds_double*)frag + lane;
  const int64_t Sd = g.S, m0 = ((int64_t)blockIdx.x * 4 + w) * 16;
  if (m0 >= g.C) return;
  const int S8 = (int)(Sd * 8);
  const unsigned voff = (unsigned)((4 * gq) * S8 + c * 8);                // plane 4 g of a group of 16, mode c of the tile

  // ---- eigenvalues of this lane's orbits: lam[tab][tau = 2 T + h][f] = (E t)(orbit member f of omega = 8 T + 4 h + g) ------------
  d4 lam[NTAB][2 * NT];
  {
    const int T8 = (int)(g.C * 8);
    const unsigned tvoff = (unsigned)(gq * T8 + c * 8);
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
      const rsrc_t tr = make_rsrc(g.tab[t] + m0, NY * T8);
#pragma unroll
      for (int tau = 0; tau < 2 * NT; ++tau) lam[t][tau] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int s = 0; s < S::KE; ++s) {
        const double tv = ld_lane(tr, tvoff, 4 * s * T8);
#pragma unroll
        for (int tau = 0; tau < 2 * NT; ++tau)
          lam[t][tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(fl0[S::eig(tau, s) * 64], tv, lam[t][tau], 0, 0, 0);
      }
    }
  }

  int64_t r = blockIdx.y;
  if (r >= g.R) return;
  const int64_t rstep = gridDim.y, rowlen = (int64_t)NY * Sd;
  const int ny_out = g.y1 - g.y0;
  const int64_t ostep = (int64_t)ny_out * Sd;
  const int in_bytes = NY * S8;

  auto load_row = [&](double (&x)[NIN][4 * KS], int64_t row) {
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
      const rsrc_t rs = make_rsrc(g.in[t] + row * rowlen + m0, in_bytes);
#pragma unroll
      for (int rho = 0; rho < 4; ++rho)
#pragma unroll
        for (int s = 0; s < KS; ++s) x[t][rho * KS + s] = ld_stream(rs, voff, (16 * s + rho) * S8);
    }
  };

  auto body = [&](double (&x)[NIN][4 * KS], double (&xn)[NIN][4 * KS], int64_t row) {
    if (row + rstep < g.R) load_row(xn, row + rstep);
#ifdef SY_ROW_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    // the fragment base is made opaque per row: the fragments are READ from LDS next to every MFMA (one ds_read_b64 each), not hoisted
    // out of the row loop into 128-200 registers (accumulation registers at that: every use would then cost two v_accvgpr_read)
    lds_double* fl = fl0;
    asm volatile("" : "+v"(fl));
    // ---- analysis: class sums per tile, then the orbit butterflies ----------------------------------------------------------
    double a[NIN][NT][2][4], b[NIN][NT][2][4];
#pragma unroll
    for (int t = 0; t < NIN; ++t) {
      d4 acc[4][NT];
#pragma unroll
      for (int rho = 0; rho < 4; ++rho)
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[rho][T] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int T = 0; T < NT; ++T)
            acc[rho][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[S::fwd(rho, s, T) * 64], x[t][rho * KS + s], acc[rho][T], 0, 0, 0);
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double Cc[4] = {acc[0][T][h], acc[1][T][h], acc[2][T][h], acc[3][T][h]};
          const double Ss[4] = {acc[0][T][2 + h], acc[1][T][2 + h], acc[2][T][2 + h], acc[3][T][2 + h]};
          bfly_fwd(Cc, Ss, a[t][T][h], b[t][T][h]);
        }
    }
    // ---- per output block: eigenvalue scaling folded into the first level of the synthesis butterfly, synthesis, store --------
#pragma unroll
    for (int jb = 0; jb < NOUT; ++jb) {
      asm volatile("" : "+v"(fl));      // (per block as well: the synthesis fragments of block 0 must not stay live through block 1)
      double Y[4][NT][4];
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double sa, da, sb, db, tb0, tb1, tb2, tb3;
          if constexpr (NIN == 1) {
            const d4 l = lam[jb][2 * T + h];
            const double (&aa)[4] = a[0][T][h];
            const double (&bb)[4] = b[0][T][h];
            const double t0 = l[0] * aa[0], t1 = l[2] * aa[2], u0 = l[0] * bb[0], u1 = l[2] * bb[2];
            sa = __builtin_fma(l[1], aa[1], t0); da = __builtin_fma(-l[1], aa[1], t0);
            sb = __builtin_fma(l[3], aa[3], t1); db = __builtin_fma(-l[3], aa[3], t1);
            tb1 = __builtin_fma(l[1], bb[1], u0); tb0 = __builtin_fma(-l[1], bb[1], u0);
            tb3 = __builtin_fma(l[3], bb[3], u1); tb2 = __builtin_fma(-l[3], bb[3], u1);
          } else {
            // V_0 = T(D0) x_g + T(X)(x_g + x_m), V_1 = T(D1) x_m + T(X)(x_g + x_m): in the spectrum, per member of the orbit
            const d4 ld = lam[jb == 0 ? 0 : 2][2 * T + h], lx = lam[1][2 * T + h];
            double ya[4], yb[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              const double pa = lx[f] * (a[0][T][h][f] + a[1][T][h][f]), pb = lx[f] * (b[0][T][h][f] + b[1][T][h][f]);
              ya[f] = __builtin_fma(ld[f], a[jb][T][h][f], pa);
              yb[f] = __builtin_fma(ld[f], b[jb][T][h][f], pb);
            }
            sa = ya[0] + ya[1]; da = ya[0] - ya[1]; sb = ya[2] + ya[3]; db = ya[2] - ya[3];
            tb0 = yb[0] - yb[1]; tb1 = yb[0] + yb[1]; tb2 = yb[2] - yb[3]; tb3 = yb[2] + yb[3];
          }
          Y[0][T][h] = sa + sb; Y[2][T][h] = sa - sb; Y[1][T][h] = da + tb3; Y[3][T][h] = da - tb3;
          Y[0][T][2 + h] = tb0 + tb2; Y[2][T][2 + h] = tb0 - tb2; Y[1][T][2 + h] = tb1 - db; Y[3][T][2 + h] = tb1 + db;
        }
      d4 o[4][MJ];
#pragma unroll
      for (int rho = 0; rho < 4; ++rho)
#pragma unroll
        for (int mj = 0; mj < MJ; ++mj) o[rho][mj] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int mj = 0; mj < MJ; ++mj)
              o[rho][mj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[S::inv(rho, T, rr, mj) * 64], Y[rho][T][rr], o[rho][mj], 0, 0, 0);
      // (slab [y0, y1): the descriptor's base is moved y0 planes in front of the row, so that plane y sits at offset y S8 whatever the
      // slab -- scalar offsets are unsigned; only the lanes with y0 <= y < y1 store, all of them inside the row)
      const rsrc_t dst = make_rsrc(g.out[jb] + row * ostep + m0 - (int64_t)g.y0 * Sd, g.y1 * S8);
#pragma unroll
      for (int rho = 0; rho < 4; ++rho)
#pragma unroll
        for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            // y = 4 (16 mj + g + 4 rr) + rho = (64 mj + 16 rr + rho) + 4 g: the lane part is the input's voff
            const int yb = 64 * mj + 16 * rr + rho;
            if (16 * mj + 4 * rr >= S::NJ) continue;                        // (padded outputs j >= ny/4 of an extent that is not a multiple of 64)
            if constexpr (FULL) st_lane(dst, voff, yb * S8, o[rho][mj][rr]);
            else {
              const int y = yb + 4 * gq;
              if (y >= g.y0 && y < g.y1) st_lane(dst, voff, yb * S8, o[rho][mj][rr]);
            }
          }
    }
  };

  double xa[NIN][4 * KS], xb[NIN][4 * KS];
  load_row(xa, r);
  while (true) {
    body(xa, xb, r);
    r += rstep;
    if (r >= g.R) break;
    body(xb, xa, r);
    r += rstep;
    if (r >= g.R) break;
  }
}

template <int NY, int NIN, int NOUT, bool FULL>
int launch_full(const SYArgs& g, hipStream_t st) {
  using S = Shape<NY>;
  constexpr size_t lds = (size_t)S::NF * 64 * sizeof(double);
  static_assert(lds <= 163840, "LDS");
  auto kern = spectral_y_kernel<NY, NIN, NOUT, FULL>;
  static std::atomic<uint64_t> attr_done{0};       // per-device "large-LDS attribute set" bits (include/geobo_hip.h, conventions)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  const int64_t nbx = (g.C + 63) / 64;
  // a wave keeps its 16 modes for R / gy rows.  Two property blocks / two terms take more than 256 registers: one 4-wave workgroup per
  // CU, one round; a single block fits 256 -- two workgroups per CU, 0.850 -> 0.808 ms (profiles/r06_spectral_y_ab.txt)
  const int64_t target = (NIN == 1 && NOUT == 1) ? 512 : 256;
  int64_t gy = 1;
  while (nbx * gy < target && gy < g.R) ++gy;
  hipLaunchKernelGGL(kern, dim3((unsigned)nbx, (unsigned)gy), dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int NY, int NIN, int NOUT>
int launch(const SYArgs& g, hipStream_t st) {
  return (g.y0 == 0 && g.y1 == NY) ? launch_full<NY, NIN, NOUT, true>(g, st) : launch_full<NY, NIN, NOUT, false>(g, st);
}

// ---- long y axes (ny = 80 .. 128; 128 = BASELINE config 5): FOUR waves share one tile of 16 modes ------------------------------------------
// At ny = 128 a lane of the kernel above would hold 64 spectral values per term, 32 eigenvalues per table (three to six tables) and 64
// values per block on the way back: over the register file.  Here the work on a tile is split between the four waves of a workgroup,
// and the split changes twice on the way through the spectrum:
//   analysis    wave rho reads and contracts the inputs of ITS residue class y' = 4 j + rho (a quarter of the tile: nothing is read
//               twice) against every orbit tile: NT x ny/16 MFMAs -> the class sums C_rho, S_rho of all orbits;
//   exchange 1  the class sums of orbit tile T go to wave T (LDS, 32 KiB, one barrier);
//   butterfly   wave T owns the orbits omega = 8 T .. 8 T + 7 (two per lane) and THEIR eigenvalues: 2 x 4 per table and lane -- a quarter;
//   exchange 2  the scaled class sums Y[rho] of every output block go back to wave rho (32 KiB per block, one barrier for all blocks);
//   synthesis   wave rho owns the outputs y = 4 j + rho: (ny/64 tiles of j) x (4 NT k-steps), final -- no partial sums to add -- and stores
//               them (optionally adding into the output: the second term of a two-term row, geobo_toeplitz_y3_add's form).
// ny = 80 / 96 have three orbit tiles: wave 3 sits out the butterflies.  A wave's fragments (analysis and synthesis of its class) stay in
// registers; those of the eigenvalue transform are read from the blob in the prologue.
struct SY3Args {
  const double* in[2];      // [R][NY][S] per term (NIN = 2: V_j = T(tab[0][j]) in[0] + T(tab[1][j]) in[1], the two-term rows)
  const double* tab[2][3];
  double* out[3];
  const double* basis;
  int64_t C, S, R;
  int y0, y1;
};

template <int NY, int NIN, int NOUT, bool ACC>
__global__ void __launch_bounds__(256, 1) spectral_y_split_kernel(SY3Args g) {
  using S = Shape<NY>;
  constexpr int NT = S::NT, MJ = S::MJ, KS = S::KS;
  static_assert(NT <= 4 && NOUT >= 1 && NOUT <= 3 && NIN >= 1 && NIN <= 2 && !(ACC && NIN == 2), "shape");
  // LDS holds the two exchange areas only: ex1[dest T][src rho][reg][lane] (class sums), ex2[block][dest rho][src T][reg][lane] (scaled
  // class sums).  The fragments a wave needs -- analysis of ITS class, synthesis of ITS class: 2 x ny/4 doubles per lane -- stay in
  // registers for the whole launch (in LDS they cost 128 KiB at ny = 128 and a read next to every MFMA; with the exchange areas apart
  // from each other and from the fragments a row needs TWO barriers instead of 2 + 2 per block).
  extern __shared__ __attribute__((aligned(16))) double exch[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, gq = lane >> 4;
  lds_double* const ex1 = (lds_double*)exch + lane;
  lds_double* const ex2 = ex1 + NIN * 4 * 4 * 4 * 64;                     // ex1: [term][dest T][src rho][reg]
  const int64_t Sd = g.S, m0 = (int64_t)blockIdx.x * 16;
  const int S8 = (int)(Sd * 8);
  const unsigned voff = (unsigned)((4 * gq) * S8 + c * 8);
  const bool orb = w < NT;                                                // this wave owns an orbit tile

  double ffrag[KS * NT], ifrag[NT * 4 * MJ];
  {
    const double* bf = g.basis + lane;
#pragma unroll
    for (int i = 0; i < KS * NT; ++i) ffrag[i] = bf[(size_t)(w * KS * NT + i) * 64];                       // fwd(rho = w, s, T)
#pragma unroll
    for (int i = 0; i < NT * 4 * MJ; ++i) ifrag[i] = bf[(size_t)(S::NF_FWD + w * NT * 4 * MJ + i) * 64];   // inv(rho = w, T, r, mj)
  }
  d4 lam[NIN][NOUT][2];
  if (orb) {
    const int T8 = (int)(g.C * 8);
    const unsigned tvoff = (unsigned)(gq * T8 + c * 8);
    const double* ef = g.basis + ((size_t)S::eig(2 * w, 0)) * 64 + lane;    // eigenvalue fragments of tau = 2 w, 2 w + 1
#pragma unroll
    for (int u = 0; u < NIN; ++u)
#pragma unroll
      for (int t = 0; t < NOUT; ++t) {
        const rsrc_t tr = make_rsrc(g.tab[u][t] + m0, NY * T8);
        lam[u][t][0] = d4{0., 0., 0., 0.};
        lam[u][t][1] = d4{0., 0., 0., 0.};
#pragma unroll 4
        for (int s = 0; s < S::KE; ++s) {
          const double tv = ld_lane(tr, tvoff, 4 * s * T8);
          lam[u][t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ef[(size_t)s * 64], tv, lam[u][t][0], 0, 0, 0);
          lam[u][t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ef[(size_t)(S::KE + s) * 64], tv, lam[u][t][1], 0, 0, 0);
        }
      }
  }

  int64_t r = blockIdx.y;
  if (r >= g.R) return;
  const int64_t rstep = gridDim.y, rowlen = (int64_t)NY * Sd;
  const int64_t ostep = (int64_t)(g.y1 - g.y0) * Sd;
  const int in_bytes = NY * S8;

  auto load_row = [&](double (&x)[NIN][KS], int64_t row) {     // this wave's residue class: planes y' = 16 s + 4 g + w
#pragma unroll
    for (int u = 0; u < NIN; ++u) {
      const rsrc_t rs = make_rsrc(g.in[u] + row * rowlen + m0, in_bytes);
#pragma unroll
      for (int s = 0; s < KS; ++s) x[u][s] = ld_stream(rs, voff, (16 * s + w) * S8);
    }
  };

  auto body = [&](double (&x)[NIN][KS], double (&xn)[NIN][KS], int64_t row) {
    if (row + rstep < g.R) load_row(xn, row + rstep);
    // accumulating form: the outputs this wave will add into are requested HERE, a whole row of arithmetic ahead of their use (requested
    // next to the synthesis they cost a memory latency per block: 5.6 ms against 3.5 for three blocks at ny = 128)
    double prev[ACC ? NOUT : 1][MJ][4];
    if constexpr (ACC) {
#pragma unroll
      for (int jb = 0; jb < NOUT; ++jb) {
        const rsrc_t dst = make_rsrc(g.out[jb] + row * ostep + m0 - (int64_t)g.y0 * Sd, g.y1 * S8);
#pragma unroll
        for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int yb = 64 * mj + 16 * rr + w, y = yb + 4 * gq;
            prev[jb][mj][rr] = (16 * mj + 4 * rr < S::NJ && y >= g.y0 && y < g.y1) ? ld_lane(dst, voff, yb * S8) : 0.0;
          }
      }
    }
#pragma unroll
    for (int u = 0; u < NIN; ++u) {
      d4 acc[NT];
#pragma unroll
      for (int T = 0; T < NT; ++T) acc[T] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(ffrag[s * NT + T], x[u][s], acc[T], 0, 0, 0);
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) ex1[(((u * 4 + T) * 4 + w) * 4 + q) * 64] = acc[T][q];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): this wave's exchange writes have landed
    __builtin_amdgcn_s_barrier();                                          // (1) every class's sums of this row are in ex1
    if (orb) {
      double a[NIN][2][4], b[NIN][2][4];
#pragma unroll
      for (int u = 0; u < NIN; ++u) {
        double cs[4][4];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[rho][q] = ex1[(((u * 4 + w) * 4 + rho) * 4 + q) * 64];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double Cc[4] = {cs[0][h], cs[1][h], cs[2][h], cs[3][h]};
          const double Ss[4] = {cs[0][2 + h], cs[1][2 + h], cs[2][2 + h], cs[3][2 + h]};
          bfly_fwd(Cc, Ss, a[u][h], b[u][h]);
        }
      }
#pragma unroll
      for (int jb = 0; jb < NOUT; ++jb) {
        double Y[4][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double sa, da, sb, db, tb0, tb1, tb2, tb3;
          if constexpr (NIN == 1) {
            const d4 l = lam[0][jb][h];
            const double t0 = l[0] * a[0][h][0], t1 = l[2] * a[0][h][2], u0 = l[0] * b[0][h][0], u1 = l[2] * b[0][h][2];
            sa = __builtin_fma(l[1], a[0][h][1], t0); da = __builtin_fma(-l[1], a[0][h][1], t0);
            sb = __builtin_fma(l[3], a[0][h][3], t1); db = __builtin_fma(-l[3], a[0][h][3], t1);
            tb1 = __builtin_fma(l[1], b[0][h][1], u0); tb0 = __builtin_fma(-l[1], b[0][h][1], u0);
            tb3 = __builtin_fma(l[3], b[0][h][3], u1); tb2 = __builtin_fma(-l[3], b[0][h][3], u1);
          } else {
            // the two terms meet in the spectrum: Y^ = lambda_g x^_g + lambda_m x^_m per member of the orbit
            const d4 lg = lam[0][jb][h], lm = lam[1][jb][h];
            double ya[4], yb[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              ya[f] = __builtin_fma(lm[f], a[1][h][f], lg[f] * a[0][h][f]);
              yb[f] = __builtin_fma(lm[f], b[1][h][f], lg[f] * b[0][h][f]);
            }
            sa = ya[0] + ya[1]; da = ya[0] - ya[1]; sb = ya[2] + ya[3]; db = ya[2] - ya[3];
            tb0 = yb[0] - yb[1]; tb1 = yb[0] + yb[1]; tb2 = yb[2] - yb[3]; tb3 = yb[2] + yb[3];
          }
          Y[0][h] = sa + sb; Y[2][h] = sa - sb; Y[1][h] = da + tb3; Y[3][h] = da - tb3;
          Y[0][2 + h] = tb0 + tb2; Y[2][2 + h] = tb0 - tb2; Y[1][2 + h] = tb1 - db; Y[3][2 + h] = tb1 + db;
        }
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) ex2[(((jb * 4 + rho) * 4 + w) * 4 + rr) * 64] = Y[rho][rr];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                          // (A) every orbit tile's scaled class sums, all blocks, are in ex2
    // (no further barrier: ex1 is rewritten only by waves that are past (A) -- every read of it lies in front of (A) -- and ex2 only
    // behind the next row's (1), which a wave reaches after its own reads of ex2 below)
#pragma unroll
    for (int jb = 0; jb < NOUT; ++jb) {
      // slab [y0, y1): the descriptor's base sits y0 planes in front of the row (see the kernel above)
      const rsrc_t dst = make_rsrc(g.out[jb] + row * ostep + m0 - (int64_t)g.y0 * Sd, g.y1 * S8);
      double yin[NT][4];
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) yin[T][rr] = ex2[(((jb * 4 + w) * 4 + T) * 4 + rr) * 64];
      // (all sixteen reads in flight, ONE wait, then the MFMAs back to back: interleaved read -> wait -> MFMA pairs leave an LDS latency
      // per pair uncovered on a SIMD that runs a single wave)
      __builtin_amdgcn_sched_barrier(0);
      d4 o[MJ];
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj) o[mj] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int mj = 0; mj < MJ; ++mj)
            o[mj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ifrag[(T * 4 + rr) * MJ + mj], yin[T][rr], o[mj], 0, 0, 0);
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (16 * mj + 4 * rr >= S::NJ) continue;
          const int yb = 64 * mj + 16 * rr + w;                            // y = 4 (16 mj + g + 4 rr) + rho, rho = w
          const int y = yb + 4 * gq;
          if (y >= g.y0 && y < g.y1) {
            double v = o[mj][rr];
            if constexpr (ACC) v += prev[jb][mj][rr];
            st_lane(dst, voff, yb * S8, v);
          }
        }
    }
  };

  double xa[NIN][KS], xb[NIN][KS];
  load_row(xa, r);
  while (true) {
    body(xa, xb, r);
    r += rstep;
    if (r >= g.R) break;
    body(xb, xa, r);
    r += rstep;
    if (r >= g.R) break;
  }
}

// Pipelined long-axis kernel (round 6, second half; one-term forms).  Ablations of the four-wave kernel above at ny = 128, three blocks
// (profiles/r06_spectral_y_ab.txt): without its loads and stores 3.26 of 3.57 ms remain; without its MFMAs 2.14 ms (then HBM-bound,
// 4.0 TB/s); its MFMAs alone are 1.86 ms of matrix-pipe time.  So a row costs the pipe 8192 cycles of MFMA and about 6000 more in which
// nothing is issued: the three phases of a row (analysis | orbit butterflies + eigenvalues | synthesis) are separated by workgroup
// barriers and the middle one is pure latency -- LDS round trips and a short vector chain on NT of the waves.  Splitting every role over
// two waves per SIMD changed nothing (2.86 / 3.62 ms: both waves of a SIMD sit in the same phase).  Here the phases of DIFFERENT steps
// overlap instead: waves 0..3 (one per SIMD) only do matrix work, waves 4..7 only the orbit tiles; a step is one (row, block) pair,
//     step t:  matrix waves  -- synthesis of pair t - 1 (ex2[(t-1) & 1]), stores;  behind a row's last block: analysis of the next row
//                               into ex1[(k+1) & 1]
//              orbit waves   -- pair t: (first block of a row: class sums from ex1[k & 1] -> spectrum, kept in registers) eigenvalues,
//                               inverse butterfly -> ex2[t & 1]
//              ONE barrier
// with both exchange areas double buffered: 4 x 32 KiB for any number of blocks.  Same arithmetic and summation order per output as the
// four-wave kernel: bit-identical results.
template <int NY, int NIN, int NOUT, bool ACC>
__global__ void __launch_bounds__(512, 1) spectral_y_pipe_kernel(SY3Args g) {
  using S = Shape<NY>;
  constexpr int NT = S::NT, MJ = S::MJ, KS = S::KS, EX = 4 * 4 * 4 * 64;
  static_assert(NT <= 4 && NOUT >= 1 && NOUT <= 3 && NIN >= 1 && NIN <= 2 && !(ACC && NIN == 2), "shape");
  // ex1 needs its second buffer only where a row is ONE step (one block): with two or three blocks the next row's analysis runs in the
  // row's LAST step and the orbit waves read ex1 in its FIRST
  constexpr int E1B = NOUT == 1 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double exch[];
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = w8 & 3;                                                   // matrix waves: residue class rho; orbit waves: orbit tile T
  const bool mat = w8 < 4, orb = !mat && w < NT;
  const int c = lane & 15, gq = lane >> 4;
  lds_double* const ex1 = (lds_double*)exch + lane;                       // [parity][term][dest T][src rho][reg]
  lds_double* const ex2 = ex1 + E1B * NIN * EX;                           // [parity][dest rho][src T][reg]
  const int64_t Sd = g.S, m0 = (int64_t)blockIdx.x * 16;
  const int S8 = (int)(Sd * 8);
  const unsigned voff = (unsigned)((4 * gq) * S8 + c * 8);
  const int64_t r0 = blockIdx.y, rstep = gridDim.y, rowlen = (int64_t)NY * Sd;
  if (r0 >= g.R) return;
  const int K = (int)((g.R - r0 + rstep - 1) / rstep);                    // rows of this workgroup
  const int nsteps = K * NOUT;
  const int64_t ostep = (int64_t)(g.y1 - g.y0) * Sd;
  const int in_bytes = NY * S8;

  if (mat) {
    double ffrag[KS * NT], ifrag[NT * 4 * MJ];
    {
      const double* bf = g.basis + lane;
#pragma unroll
      for (int i = 0; i < KS * NT; ++i) ffrag[i] = bf[(size_t)(w * KS * NT + i) * 64];
#pragma unroll
      for (int i = 0; i < NT * 4 * MJ; ++i) ifrag[i] = bf[(size_t)(S::NF_FWD + w * NT * 4 * MJ + i) * 64];
    }
    auto load_row = [&](double (&x)[NIN][KS], int k) {
#pragma unroll
      for (int u = 0; u < NIN; ++u) {
        const rsrc_t rs = make_rsrc(g.in[u] + (r0 + k * rstep) * rowlen + m0, in_bytes);
#pragma unroll
        for (int s = 0; s < KS; ++s) x[u][s] = ld_stream(rs, voff, (16 * s + w) * S8);
      }
    };
    auto analysis = [&](const double (&x)[NIN][KS], int k) {
      const int par = E1B == 2 ? (k & 1) : 0;
#pragma unroll
      for (int u = 0; u < NIN; ++u) {
        d4 acc[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T] = d4{0., 0., 0., 0.};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int T = 0; T < NT; ++T) acc[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(ffrag[s * NT + T], x[u][s], acc[T], 0, 0, 0);
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int q = 0; q < 4; ++q) ex1[(par * NIN + u) * EX + ((T * 4 + w) * 4 + q) * 64] = acc[T][q];
      }
    };
    auto load_prev = [&](double (&pv)[MJ][4], int k, auto JB) {            // accumulating form: the outputs pair (k, JB) will be added into
      constexpr int jb = decltype(JB)::value;
      const rsrc_t dst = make_rsrc(g.out[jb] + (r0 + k * rstep) * ostep + m0 - (int64_t)g.y0 * Sd, g.y1 * S8);
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int yb = 64 * mj + 16 * rr + w, y = yb + 4 * gq;
          pv[mj][rr] = (16 * mj + 4 * rr < S::NJ && y >= g.y0 && y < g.y1) ? ld_lane(dst, voff, yb * S8) : 0.0;
        }
    };
    double x[NIN][KS], prev[ACC ? MJ : 1][4], pnext[ACC ? MJ : 1][4];
    load_row(x, 0);
    analysis(x, 0);
    if (K > 1) load_row(x, 1);
    if constexpr (ACC) load_prev(prev, 0, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    auto synthesis = [&](int kk, auto JB, int par) {                       // pair (row kk, block JB) out of ex2[par]
      constexpr int jb = decltype(JB)::value;
      const rsrc_t dst = make_rsrc(g.out[jb] + (r0 + kk * rstep) * ostep + m0 - (int64_t)g.y0 * Sd, g.y1 * S8);
      lds_double* const e2 = ex2 + par * EX;
      double yin[NT][4];
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) yin[T][rr] = e2[((w * 4 + T) * 4 + rr) * 64];
      __builtin_amdgcn_sched_barrier(0);
      d4 o[MJ];
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj) o[mj] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int mj = 0; mj < MJ; ++mj)
            o[mj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ifrag[(T * 4 + rr) * MJ + mj], yin[T][rr], o[mj], 0, 0, 0);
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (16 * mj + 4 * rr >= S::NJ) continue;
          const int yb = 64 * mj + 16 * rr + w, y = yb + 4 * gq;
          if (y >= g.y0 && y < g.y1) {
            double v = o[mj][rr];
            if constexpr (ACC) v += prev[mj][rr];
            st_lane(dst, voff, yb * S8, v);
          }
        }
    };
    auto step_end = [&]() {
      if constexpr (ACC) {
#pragma unroll
        for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) prev[mj][rr] = pnext[mj][rr];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
    };
    using std::integral_constant;
    for (int kk = 0; kk < K; ++kk) {
      // step (kk, 0): the previous row's last pair
      if constexpr (ACC) { if (kk > 0) load_prev(pnext, kk, integral_constant<int, 0>{}); }
      if (kk > 0) synthesis(kk - 1, integral_constant<int, NOUT - 1>{}, (kk * NOUT - 1) & 1);
      if constexpr (NOUT == 1) {
        if (kk + 1 < K) { analysis(x, kk + 1); if (kk + 2 < K) load_row(x, kk + 2); }
      }
      if (kk > 0 || !ACC) step_end(); else { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
      if constexpr (NOUT >= 2) {                                           // step (kk, 1): pair (kk, 0)
        if constexpr (ACC) load_prev(pnext, kk, integral_constant<int, 1>{});
        synthesis(kk, integral_constant<int, 0>{}, (kk * NOUT) & 1);
        if constexpr (NOUT == 2) {
          if (kk + 1 < K) { analysis(x, kk + 1); if (kk + 2 < K) load_row(x, kk + 2); }
        }
        step_end();
      }
      if constexpr (NOUT >= 3) {                                           // step (kk, 2): pair (kk, 1)
        if constexpr (ACC) load_prev(pnext, kk, integral_constant<int, 2>{});
        synthesis(kk, integral_constant<int, 1>{}, (kk * NOUT + 1) & 1);
        if (kk + 1 < K) { analysis(x, kk + 1); if (kk + 2 < K) load_row(x, kk + 2); }
        step_end();
      }
    }
    synthesis(K - 1, integral_constant<int, NOUT - 1>{}, (K * NOUT - 1) & 1);   // drain
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  } else {
    d4 lam[NIN][NOUT][2];
    if (orb) {
      const int T8 = (int)(g.C * 8);
      const unsigned tvoff = (unsigned)(gq * T8 + c * 8);
      const double* ef = g.basis + ((size_t)S::eig(2 * w, 0)) * 64 + lane;
#pragma unroll
      for (int u = 0; u < NIN; ++u)
#pragma unroll
        for (int t = 0; t < NOUT; ++t) {
          const rsrc_t tr = make_rsrc(g.tab[u][t] + m0, NY * T8);
          lam[u][t][0] = d4{0., 0., 0., 0.};
          lam[u][t][1] = d4{0., 0., 0., 0.};
#pragma unroll 4
          for (int s = 0; s < S::KE; ++s) {
            const double tv = ld_lane(tr, tvoff, 4 * s * T8);
            lam[u][t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ef[(size_t)s * 64], tv, lam[u][t][0], 0, 0, 0);
            lam[u][t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ef[(size_t)(S::KE + s) * 64], tv, lam[u][t][1], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                          // row 0's class sums are in ex1[0]
    double a[NIN][2][4], b[NIN][2][4];
    for (int kk = 0; kk < K; ++kk) {
#pragma unroll
      for (int jb = 0; jb < NOUT; ++jb) {
        if (orb) {
          if (jb == 0) {
#pragma unroll
            for (int u = 0; u < NIN; ++u) {
              lds_double* const e1 = ex1 + ((E1B == 2 ? (kk & 1) : 0) * NIN + u) * EX;
              double cs[4][4];
#pragma unroll
              for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                for (int q = 0; q < 4; ++q) cs[rho][q] = e1[((w * 4 + rho) * 4 + q) * 64];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const double Cc[4] = {cs[0][h], cs[1][h], cs[2][h], cs[3][h]};
                const double Ss[4] = {cs[0][2 + h], cs[1][2 + h], cs[2][2 + h], cs[3][2 + h]};
                bfly_fwd(Cc, Ss, a[u][h], b[u][h]);
              }
            }
          }
          double Y[4][4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double sa, da, sb, db, tb0, tb1, tb2, tb3;
            if constexpr (NIN == 1) {
              const d4 l = lam[0][jb][h];
              const double t0 = l[0] * a[0][h][0], t1 = l[2] * a[0][h][2], u0 = l[0] * b[0][h][0], u1 = l[2] * b[0][h][2];
              sa = __builtin_fma(l[1], a[0][h][1], t0); da = __builtin_fma(-l[1], a[0][h][1], t0);
              sb = __builtin_fma(l[3], a[0][h][3], t1); db = __builtin_fma(-l[3], a[0][h][3], t1);
              tb1 = __builtin_fma(l[1], b[0][h][1], u0); tb0 = __builtin_fma(-l[1], b[0][h][1], u0);
              tb3 = __builtin_fma(l[3], b[0][h][3], u1); tb2 = __builtin_fma(-l[3], b[0][h][3], u1);
            } else {
              // the two terms meet in the spectrum: Y^ = lambda_g x^_g + lambda_m x^_m per member of the orbit
              const d4 lg = lam[0][jb][h], lm = lam[1][jb][h];
              double ya[4], yb[4];
#pragma unroll
              for (int f = 0; f < 4; ++f) {
                ya[f] = __builtin_fma(lm[f], a[1][h][f], lg[f] * a[0][h][f]);
                yb[f] = __builtin_fma(lm[f], b[1][h][f], lg[f] * b[0][h][f]);
              }
              sa = ya[0] + ya[1]; da = ya[0] - ya[1]; sb = ya[2] + ya[3]; db = ya[2] - ya[3];
              tb0 = yb[0] - yb[1]; tb1 = yb[0] + yb[1]; tb2 = yb[2] - yb[3]; tb3 = yb[2] + yb[3];
            }
            Y[0][h] = sa + sb; Y[2][h] = sa - sb; Y[1][h] = da + tb3; Y[3][h] = da - tb3;
            Y[0][2 + h] = tb0 + tb2; Y[2][2 + h] = tb0 - tb2; Y[1][2 + h] = tb1 - db; Y[3][2 + h] = tb1 + db;
          }
          lds_double* const e2 = ex2 + ((kk * NOUT + jb) & 1) * EX;
#pragma unroll
          for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) e2[((rho * 4 + w) * 4 + rr) * 64] = Y[rho][rr];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                          // (the matrix waves' drain step)
  }
}

#ifndef SY_PIPE
#define SY_PIPE 1        // 0: the four-wave kernel for the one-term forms as well: the A/B
#endif
template <int NY, int NIN, int NOUT, bool ACC>
int launch_split(const SY3Args& g, hipStream_t st) {
  using S = Shape<NY>;
  constexpr bool PIPE = SY_PIPE && !(NIN == 2 && NOUT == 1);   // the pipelined kernel (eight waves) wherever its exchange areas fit
  constexpr size_t lds = (size_t)((PIPE ? (NOUT == 1 ? 2 : 1) * NIN + 2 : NIN + NOUT) * 4 * 4 * 4 * 64) * sizeof(double);
  static_assert(lds <= 163840, "LDS");
  const void* kern;
  if constexpr (PIPE) kern = reinterpret_cast<const void*>(spectral_y_pipe_kernel<NY, NIN, NOUT, ACC>);
  else kern = reinterpret_cast<const void*>(spectral_y_split_kernel<NY, NIN, NOUT, ACC>);
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  const int64_t nbx = g.C / 16;
  int64_t gy = 1;                                   // one workgroup per CU: a workgroup keeps its 16 modes for R / gy rows
  while (nbx * gy < 512 && gy < g.R) ++gy;
  if constexpr (PIPE) hipLaunchKernelGGL((spectral_y_pipe_kernel<NY, NIN, NOUT, ACC>), dim3((unsigned)nbx, (unsigned)gy), dim3(512), lds, st, g);
  else hipLaunchKernelGGL((spectral_y_split_kernel<NY, NIN, NOUT, ACC>), dim3((unsigned)nbx, (unsigned)gy), dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int NY>
int launch_split_by(const SY3Args& g, int nprop, bool acc, hipStream_t st) {
  if (acc) return nprop == 1 ? launch_split<NY, 1, 1, true>(g, st) : nprop == 2 ? launch_split<NY, 1, 2, true>(g, st) : launch_split<NY, 1, 3, true>(g, st);
  return nprop == 1 ? launch_split<NY, 1, 1, false>(g, st) : nprop == 2 ? launch_split<NY, 1, 2, false>(g, st) : launch_split<NY, 1, 3, false>(g, st);
}

template <int NY>
int launch_split2_by(const SY3Args& g, int nprop, hipStream_t st) {
  return nprop == 1 ? launch_split<NY, 2, 1, false>(g, st) : nprop == 2 ? launch_split<NY, 2, 2, false>(g, st) : launch_split<NY, 2, 3, false>(g, st);
}

// ---- axis passes of the (x, z) transforms for extents without a fused two-axis kernel (round 6) --------------------------------------------
// The batched axis passes of spectral.py along a STRIDED axis (the x step of the covariance product's transforms: planes of Pz contiguous
// modes, n or 2n of them per (row, y) item) are the analysis / synthesis halves of the kernel above on their own:
//     analysis   out[item][p][c] = sum_i G[p][i] in[item][i][c]      (n planes -> P = 2n spectral planes)
//     synthesis  out[item][i][c] = sum_p G[p][i] in[item][p][c]      (P -> n)
// with G = spectral.forward_matrix(n) on the HALF-INTEGER basis (spectral.half_modes: spectral positions 8 w .. 8 w + 7 = the orbit of
// kappa = w + 1/2, rows sqrt(2) cos / sin): radix 4, n^2 / 2 multiply-adds per item and mode, inputs straight from global memory into the
// B operand layout, no LDS staging of the data.  Four waves per tile of 16 modes as above (wave = residue class on the spatial side,
// wave = orbit tile on the spectral side, ONE exchange through LDS, double buffered: one barrier per item); the geobo_gemm_fold passes
// they replace stage both operands through LDS and run at 2.7-3.9 TB/s on both roofs at once (n = 96, 128).
struct AxisArgs {
  const double* in;
  double* out;
  const double* basis;
  int64_t C, Si, So, item_in, item_out, R;     // modes per plane, plane strides, item strides (doubles), items
  int mask_ends;                               // analysis: input planes 0 and n - 1 count as zero (the lattice Gram's boundary slabs)
};

template <int N, bool INV>
__global__ void __launch_bounds__(256, 2) spectral_axis_kernel(AxisArgs g) {
  using S = Shape<N>;
  constexpr int NT = S::NT, MJ = S::MJ, KS = S::KS;
  static_assert(NT <= 4, "shape");
  __shared__ __attribute__((aligned(16))) double exch[2 * 4 * 4 * 4 * 64];      // [parity][dest][src][reg][lane]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, gq = lane >> 4;
  lds_double* const ex = (lds_double*)exch + lane;
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  const int Si8 = (int)(g.Si * 8), So8 = (int)(g.So * 8);
  const bool orb = w < NT;
  // spatial side: plane 16 s + 4 g + rho (rho = w); spectral side: plane 8 (8 T + 4 h + g) + m (T = w)
  const unsigned vsp = (unsigned)((4 * gq) * (INV ? So8 : Si8) + c * 8);
  const unsigned vsk = (unsigned)((8 * gq) * (INV ? Si8 : So8) + c * 8);
  constexpr double R2 = 1.4142135623730951;
  double frag[INV ? NT * 4 * MJ : KS * NT];
  {
    const double* bf = g.basis + lane;
#pragma unroll
    for (int i = 0; i < (INV ? NT * 4 * MJ : KS * NT); ++i)
      frag[i] = R2 * bf[(size_t)((INV ? S::NF_FWD + w * NT * 4 * MJ : w * KS * NT) + i) * 64];
  }
  const int in_bytes = (INV ? 2 * N : N) * Si8, out_bytes = (INV ? N : 2 * N) * So8;
  int par = 0;
  for (int64_t r = blockIdx.y; r < g.R; r += gridDim.y, par ^= 1) {
    const rsrc_t src = make_rsrc(g.in + r * g.item_in + m0, in_bytes);
    const rsrc_t dst = make_rsrc(g.out + r * g.item_out + m0, out_bytes);
    lds_double* const eb = ex + par * (4 * 4 * 4 * 64);
    if constexpr (!INV) {
      double x[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) x[s] = ld_stream(src, vsp, (16 * s + w) * Si8);
      if (g.mask_ends) {                                     // plane 16 s + 4 g + rho: 0 = (s 0, g 0, rho 0), n - 1 = (s KS - 1, g 3, rho 3)
        if (w == 0 && gq == 0) x[0] = 0.0;
        if (w == 3 && gq == 3) x[KS - 1] = 0.0;
      }
      d4 acc[NT];
#pragma unroll
      for (int T = 0; T < NT; ++T) acc[T] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[s * NT + T], x[s], acc[T], 0, 0, 0);
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) eb[((T * 4 + w) * 4 + q) * 64] = acc[T][q];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      if (orb) {
        double cs[4][4];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[rho][q] = eb[((w * 4 + rho) * 4 + q) * 64];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (8 * w + 4 * h >= S::NW) continue;                // (orbit tiles of an extent that is not a multiple of 32: wave uniform)
          const double Cc[4] = {cs[0][h], cs[1][h], cs[2][h], cs[3][h]};
          const double Ss[4] = {cs[0][2 + h], cs[1][2 + h], cs[2][2 + h], cs[3][2 + h]};
          double a[4], b[4];
          bfly_fwd(Cc, Ss, a, b);
          const double v[8] = {a[0], a[1], b[0], -b[1], a[2], a[3], b[2], -b[3]};
#pragma unroll
          for (int m = 0; m < 8; ++m) st_lane(dst, vsk, (8 * (8 * w + 4 * h) + m) * So8, v[m]);
        }
      }
    } else {
      if (orb) {
        double Y[4][4];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int q = 0; q < 4; ++q) Y[rho][q] = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (8 * w + 4 * h >= S::NW) continue;
          double v[8];
#pragma unroll
          for (int m = 0; m < 8; ++m) v[m] = ld_stream(src, vsk, (8 * (8 * w + 4 * h) + m) * Si8);
          const double a0 = v[0], a1 = v[1], b0 = v[2], b1 = -v[3], a2 = v[4], a3 = v[5], b2 = v[6], b3 = -v[7];
          const double sa = a0 + a1, da = a0 - a1, sb = a2 + a3, db = a2 - a3;
          const double t0 = b0 - b1, t1 = b0 + b1, t2 = b2 - b3, t3 = b2 + b3;
          Y[0][h] = sa + sb; Y[2][h] = sa - sb; Y[1][h] = da + t3; Y[3][h] = da - t3;
          Y[0][2 + h] = t0 + t2; Y[2][2 + h] = t0 - t2; Y[1][2 + h] = t1 - db; Y[3][2 + h] = t1 + db;
        }
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int q = 0; q < 4; ++q) eb[((rho * 4 + w) * 4 + q) * 64] = Y[rho][q];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      double yin[NT][4];
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) yin[T][q] = eb[((w * 4 + T) * 4 + q) * 64];
      d4 o[MJ];
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj) o[mj] = d4{0., 0., 0., 0.};
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mj = 0; mj < MJ; ++mj) o[mj] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[(T * 4 + q) * MJ + mj], yin[T][q], o[mj], 0, 0, 0);
#pragma unroll
      for (int mj = 0; mj < MJ; ++mj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (16 * mj + 4 * q >= S::NJ) continue;
          st_lane(dst, vsp, (64 * mj + 16 * q + w) * So8, o[mj][q]);
        }
    }
  }
}

template <int N>
int launch_axis(const AxisArgs& g, bool inverse, hipStream_t st) {
  const int64_t nbx = g.C / 16;
  int64_t gy = 1;                                   // two workgroups per CU (64 KiB of LDS, < 128 registers): ~2048 workgroups in flight
  while (nbx * gy < 2048 && gy < g.R) ++gy;
  if (gy > 65535) gy = 65535;
  if (inverse) hipLaunchKernelGGL((spectral_axis_kernel<N, true>), dim3((unsigned)nbx, (unsigned)gy), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((spectral_axis_kernel<N, false>), dim3((unsigned)nbx, (unsigned)gy), dim3(256), 0, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int NY>
int fill_basis(double* basis, hipStream_t st) {
  hipLaunchKernelGGL(basis_kernel<NY>, dim3(Shape<NY>::NF), dim3(64), 0, st, basis);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

}  // namespace

extern "C" int64_t geobo_spectral_y_basis_doubles(int ny) {
  switch (ny) {
    case 64: return (int64_t)Shape<64>::NF * 64;
    case 48: return (int64_t)Shape<48>::NF * 64;
    case 32: return (int64_t)Shape<32>::NF * 64;
    case 80: return (int64_t)Shape<80>::NF * 64;
    case 96: return (int64_t)Shape<96>::NF * 64;
    case 112: return (int64_t)Shape<112>::NF * 64;
    case 128: return (int64_t)Shape<128>::NF * 64;
    default: return 0;
  }
}

extern "C" int geobo_spectral_y_basis(int ny, double* basis, void* stream) {
  if (!basis) return GEOBO_E_ARG;
  switch (ny) {
    case 64: return fill_basis<64>(basis, (hipStream_t)stream);
    case 48: return fill_basis<48>(basis, (hipStream_t)stream);
    case 32: return fill_basis<32>(basis, (hipStream_t)stream);
    case 80: return fill_basis<80>(basis, (hipStream_t)stream);
    case 96: return fill_basis<96>(basis, (hipStream_t)stream);
    case 112: return fill_basis<112>(basis, (hipStream_t)stream);
    case 128: return fill_basis<128>(basis, (hipStream_t)stream);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

extern "C" int geobo_spectral_y(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* tab0,
                                const double* tab1, double* out0, double* out1, int y0, int y1, const double* basis, void* stream) {
  if (!in || !tab0 || !out0 || !basis || (nprop == 2 && (!tab1 || !out1))) return GEOBO_E_ARG;
  if (nprop < 1 || nprop > 2 || R <= 0 || y0 < 0 || y1 > ny || y1 <= y0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 16 || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  SYArgs g;
  g.in[0] = in; g.in[1] = in; g.tab[0] = tab0; g.tab[1] = nprop == 2 ? tab1 : tab0; g.tab[2] = tab0;
  g.out[0] = out0; g.out[1] = nprop == 2 ? out1 : out0; g.basis = basis; g.C = C; g.S = plane; g.R = R; g.y0 = y0; g.y1 = y1;
  hipStream_t st = (hipStream_t)stream;
  switch (ny) {
    case 64: return nprop == 2 ? launch<64, 1, 2>(g, st) : launch<64, 1, 1>(g, st);
    case 48: return nprop == 2 ? launch<48, 1, 2>(g, st) : launch<48, 1, 1>(g, st);
    case 32: return nprop == 2 ? launch<32, 1, 2>(g, st) : launch<32, 1, 1>(g, st);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

extern "C" int geobo_spectral_y2s(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_d0,
                                  const double* tab_x, const double* tab_d1, double* out0, double* out1, const double* basis, void* stream) {
  if (!in_g || !in_m || !tab_d0 || !tab_x || !tab_d1 || !out0 || !out1 || !basis) return GEOBO_E_ARG;
  if (R <= 0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 16 || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  SYArgs g;
  g.in[0] = in_g; g.in[1] = in_m; g.tab[0] = tab_d0; g.tab[1] = tab_x; g.tab[2] = tab_d1;
  g.out[0] = out0; g.out[1] = out1; g.basis = basis; g.C = C; g.S = plane; g.R = R; g.y0 = 0; g.y1 = ny;
  hipStream_t st = (hipStream_t)stream;
  switch (ny) {
    case 64: return launch<64, 2, 2>(g, st);
    case 48: return launch<48, 2, 2>(g, st);
    case 32: return launch<32, 2, 2>(g, st);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

extern "C" int geobo_spectral_y3(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                                 double* const* outs, int y0, int y1, int accumulate, const double* basis, void* stream) {
  if (!in || !tabs || !outs || !basis || nprop < 1 || nprop > 3) return GEOBO_E_ARG;
  for (int j = 0; j < nprop; ++j)
    if (!tabs[j] || !outs[j]) return GEOBO_E_ARG;
  if (R <= 0 || y0 < 0 || y1 > ny || y1 <= y0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 16 || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  if (ny <= 64) {
    if (accumulate) return GEOBO_E_UNSUPPORTED;        // (ny <= 64 has the two-term form geobo_spectral_y2s instead)
    for (int j = 0; j < nprop; j += 2) {
      const int n = nprop - j >= 2 ? 2 : 1;
      const int rc = geobo_spectral_y(ny, C, plane, R, n, in, tabs[j], n == 2 ? tabs[j + 1] : nullptr, outs[j], n == 2 ? outs[j + 1] : nullptr, y0, y1,
                                      basis, stream);
      if (rc) return rc;
    }
    return GEOBO_OK;
  }
  SY3Args g;
  g.in[0] = g.in[1] = in; g.basis = basis; g.C = C; g.S = plane; g.R = R; g.y0 = y0; g.y1 = y1;
  for (int j = 0; j < 3; ++j) { g.tab[0][j] = g.tab[1][j] = tabs[j < nprop ? j : 0]; g.out[j] = outs[j < nprop ? j : 0]; }
  hipStream_t st = (hipStream_t)stream;
  switch (ny) {
    case 80: return launch_split_by<80>(g, nprop, accumulate != 0, st);
    case 96: return launch_split_by<96>(g, nprop, accumulate != 0, st);
    case 112: return launch_split_by<112>(g, nprop, accumulate != 0, st);
    case 128: return launch_split_by<128>(g, nprop, accumulate != 0, st);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

extern "C" int geobo_spectral_y3t(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in_g, const double* in_m,
                                  const double* const* tabs_g, const double* const* tabs_m, double* const* outs, const double* basis, void* stream) {
  if (!in_g || !in_m || !tabs_g || !tabs_m || !outs || !basis || nprop < 1 || nprop > 3) return GEOBO_E_ARG;
  for (int j = 0; j < nprop; ++j)
    if (!tabs_g[j] || !tabs_m[j] || !outs[j]) return GEOBO_E_ARG;
  if (R <= 0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 16 || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  SY3Args g;
  g.in[0] = in_g; g.in[1] = in_m; g.basis = basis; g.C = C; g.S = plane; g.R = R; g.y0 = 0; g.y1 = ny;
  for (int j = 0; j < 3; ++j) { g.tab[0][j] = tabs_g[j < nprop ? j : 0]; g.tab[1][j] = tabs_m[j < nprop ? j : 0]; g.out[j] = outs[j < nprop ? j : 0]; }
  hipStream_t st = (hipStream_t)stream;
  switch (ny) {
    case 80: return launch_split2_by<80>(g, nprop, st);
    case 96: return launch_split2_by<96>(g, nprop, st);
    case 112: return launch_split2_by<112>(g, nprop, st);
    case 128: return launch_split2_by<128>(g, nprop, st);
    default: return GEOBO_E_UNSUPPORTED;   // (ny <= 64: geobo_spectral_y2s)
  }
}

extern "C" int geobo_spectral_axis(int inverse, int n, int64_t C, int64_t plane_in, int64_t plane_out, int64_t item_in, int64_t item_out, int64_t items,
                                   const double* in, double* out, const double* basis, int mask_ends, void* stream) {
  if (!in || !out || !basis || items <= 0) return GEOBO_E_ARG;
  if (C <= 0 || C % 16 || plane_in < C || plane_out < C) return GEOBO_E_ALIGN;
  if ((int64_t)2 * n * (plane_in > plane_out ? plane_in : plane_out) * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  AxisArgs g;
  g.in = in; g.out = out; g.basis = basis; g.C = C; g.Si = plane_in; g.So = plane_out; g.item_in = item_in; g.item_out = item_out; g.R = items;
  g.mask_ends = (mask_ends && !inverse) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  switch (n) {
    case 80: return launch_axis<80>(g, inverse != 0, st);
    case 96: return launch_axis<96>(g, inverse != 0, st);
    case 112: return launch_axis<112>(g, inverse != 0, st);
    case 128: return launch_axis<128>(g, inverse != 0, st);
    default: return GEOBO_E_UNSUPPORTED;
  }
}
