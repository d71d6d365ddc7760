// potrf.hip -- blocked lower Cholesky + explicit L^-1 on gfx950 (replaces scipy.linalg.cholesky /
// solve_triangular, inversion.py:100,105,114), plus u = L^-1 y and the log-likelihood statistics
// (inversion.py:105-110).
//
// Right-looking, block size 128:
//   potf2_inv_kernel  one workgroup factors the 128x128 diagonal block in registers (2-D cyclic over 256 threads),
//                     carrying the inverse along by forward substitution, and writes L_kk and L_kk^-1 (the latter
//                     straight into the diagonal block of Linv);
//   panel solve       P = A[k+1:,k] * (L_kk^-1)^T           -> geobo_gemm_nt on the fp64 MFMA core (in place)
//   trailing update   A[k+1:,k+1:] -= P P^T (lower tiles)   -> geobo_gemm_nt, lower_only
// L^-1 is then assembled by recursive halving, two MFMA GEMMs per merge:
//   Linv[hi,lo] = -Linv[hi,hi] * (L[hi,lo] * Linv[lo,lo]).
#include <hip/hip_runtime.h>
#include <new>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

constexpr int NB = 128;

// The 128x128 diagonal block AND the running inverse live in registers: thread (ti, tj) of a 16 x 16 grid owns the
// 2-D cyclic elements (ti + 16 p, tj + 16 q), p, q < 8.  Column step j broadcasts column j of the block and row j of
// the inverse through a double-buffered 2 KiB LDS line (ONE barrier per column), then every thread applies the
// rank-1 updates   A[i][c] -= l_i l_c  (i, c > j)   and   X[i][c] -= l_i X[j][c]  (i > j >= c)   on its registers.
// X starts as the identity, so after the last column X = L_kk^-1 (forward substitution fused into the
// factorisation).  The outer loop over 16-column groups is unrolled, so all register indices are static and the
// update ranges shrink with the group index.  (A version with the block in LDS spent ~90 % of its 0.35 ms in
// ~640 barriers and LDS round trips.)
template <int JB>
__device__ __forceinline__ void potf2_group(double (&a)[8][8], double (&x)[8][8], double (&colbuf)[2][NB],
                                            double (&rowbuf)[2][NB], int tid, int ti, int tj, int kb_global,
                                            int* __restrict__ info, bool& bad_seen) {
#pragma unroll 1
  for (int jj = 0; jj < 16; ++jj) {
    const int j = JB * 16 + jj, buf = jj & 1;
    if (tj == jj) {
#pragma unroll
      for (int p = JB; p < 8; ++p) colbuf[buf][ti + 16 * p] = a[p][JB];
    }
    if (ti == jj) {
#pragma unroll
      for (int q = 0; q <= JB; ++q) rowbuf[buf][tj + 16 * q] = x[JB][q];
    }
    __syncthreads();
    const double d = colbuf[buf][j];
    if (!(d > 0.0) && !bad_seen) {  // non-positive or NaN pivot: LAPACK dpotrf's info = j (1-based)
      bad_seen = true;
      if (tid == 0 && *info == 0) *info = kb_global + j + 1;
    }
    const double r = sqrt(d);
    const double rinv = 1.0 / r;
    double li[8], lc[8], xj[8];
#pragma unroll
    for (int p = JB; p < 8; ++p) li[p] = (ti + 16 * p > j) ? colbuf[buf][ti + 16 * p] * rinv : 0.0;
#pragma unroll
    for (int q = JB; q < 8; ++q) lc[q] = (tj + 16 * q > j) ? colbuf[buf][tj + 16 * q] * rinv : 0.0;
#pragma unroll
    for (int q = 0; q <= JB; ++q) xj[q] = (tj + 16 * q <= j) ? rowbuf[buf][tj + 16 * q] * rinv : 0.0;
    if (tj == jj) {  // column j of L: scaled sub-diagonal, sqrt on the diagonal
#pragma unroll
      for (int p = JB; p < 8; ++p) {
        const int i = ti + 16 * p;
        a[p][JB] = (i > j) ? li[p] : (i == j ? r : a[p][JB]);
      }
    }
    if (ti == jj) {  // row j of the inverse is final
#pragma unroll
      for (int q = 0; q <= JB; ++q) x[JB][q] = xj[q];
    }
#pragma unroll
    for (int p = JB; p < 8; ++p) {
#pragma unroll
      for (int q = JB; q < 8; ++q) a[p][q] = __builtin_fma(-li[p], lc[q], a[p][q]);
#pragma unroll
      for (int q = 0; q <= JB; ++q) x[p][q] = __builtin_fma(-li[p], xj[q], x[p][q]);
    }
  }
}

__global__ void __launch_bounds__(256) potf2_inv_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ Linv,
                                                        int64_t ldi, int kb_global, int* __restrict__ info) {
  __shared__ double colbuf[2][NB], rowbuf[2][NB];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  double a[8][8], x[8][8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      a[p][q] = (c <= i) ? A[(int64_t)i * ld + c] : 0.0;
      x[p][q] = (c == i) ? 1.0 : 0.0;
    }
  bool bad_seen = false;
  potf2_group<0>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<1>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<2>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<3>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<4>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<5>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<6>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  potf2_group<7>(a, x, colbuf, rowbuf, tid, ti, tj, kb_global, info, bad_seen);
  // L_kk with the upper part zeroed (like scipy's lower=True result) and L_kk^-1
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      A[(int64_t)i * ld + c] = (c <= i) ? a[p][q] : 0.0;
      Linv[(int64_t)i * ldi + c] = (c <= i) ? x[p][q] : 0.0;
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

// L^-1 by recursive halving.  The two halves of a node are independent until their merge, and the merges near the
// leaves are a handful of tiles each (latency of ONE tile's k sweep, 0.2-0.5 ms, whatever the chip could do in
// parallel): with a fork context (geobo_potrf_ctx_create) the top two levels of the tree fork onto the context's streams, so
// four subtrees run concurrently, and join by events before their parent's merge.  Everything is ordered after / before
// the caller's stream by the same events.  Without a context the whole tree runs on the caller's stream.
struct InvCtx {
  const double* L; int64_t ld; double* Linv; int64_t ldi; double* ws; size_t ws_doubles;
  hipStream_t s[4]; hipEvent_t ev[6]; int nev; bool fork;
};

int build_inverse(InvCtx& c, int lo, int hi, int depth, int sidx) {
  if (hi - lo <= 1) return GEOBO_OK;
  // split on a multiple of two 128-blocks where possible: the merge GEMMs then run on 256-row tiles
  int mid = (lo + hi) / 2;
  if (hi - lo > 2 && ((mid - lo) & 1)) ++mid;
  int rc;
  if (c.fork && depth < 2 && hi - lo >= 8) {
    const int other = sidx + (depth == 0 ? 2 : 1);
    hipEvent_t fork = c.ev[c.nev++], join = c.ev[c.nev++];
    if (hipEventRecord(fork, c.s[sidx]) != hipSuccess || hipStreamWaitEvent(c.s[other], fork, 0) != hipSuccess) return GEOBO_E_LAUNCH;
    rc = build_inverse(c, lo, mid, depth + 1, sidx);
    if (rc) return rc;
    rc = build_inverse(c, mid, hi, depth + 1, other);
    if (rc) return rc;
    if (hipEventRecord(join, c.s[other]) != hipSuccess || hipStreamWaitEvent(c.s[sidx], join, 0) != hipSuccess) return GEOBO_E_LAUNCH;
  } else {
    rc = build_inverse(c, lo, mid, depth + 1, sidx);
    if (rc) return rc;
    rc = build_inverse(c, mid, hi, depth + 1, sidx);
    if (rc) return rc;
  }
  // scratch for T: the root owns the whole workspace, its children a half each, everything below a quarter per stream
  double* T = depth == 0 ? c.ws : depth == 1 ? c.ws + (sidx / 2) * (c.ws_doubles / 2) : c.ws + sidx * (c.ws_doubles / 4);
  void* st = c.s[sidx];
  const int64_t r = (int64_t)(hi - mid) * NB, cc = (int64_t)(mid - lo) * NB;
  const int64_t o_lo = (int64_t)lo * NB, o_mid = (int64_t)mid * NB;
  // T = L[mid:hi, lo:mid] * Linv[lo:mid, lo:mid]           (Y lower triangular)
  rc = geobo_gemm_nn(r, cc, cc, 1.0, c.L + o_mid * c.ld + o_lo, c.ld, c.Linv + o_lo * c.ldi + o_lo, c.ldi, 0.0, T, cc, 0, 1, st);
  if (rc) return rc;
  // Linv[mid:hi, lo:mid] = -Linv[mid:hi, mid:hi] * T         (X lower triangular)
  return geobo_gemm_nn(r, cc, r, -1.0, c.Linv + o_mid * c.ldi + o_mid, c.ldi, T, cc, 0.0, c.Linv + o_mid * c.ldi + o_lo, c.ldi, 1, 0, st);
}

// Fork context (geobo_potrf_ctx_create): three streams + a few events on the device that was current at creation, owned by
// the caller.  Nothing here is process-global: two engines (or two devices, or two threads) each bring their own.
constexpr int NEV = 14;   // 6 for the L^-1 tree, 2 x 4 for the look-ahead rings of the factorisation
struct PotrfCtx { int dev; hipStream_t s[3]; hipEvent_t ev[NEV]; };

}  // namespace

extern "C" size_t geobo_potrf_ws_bytes(int64_t m) {
  const int64_t nb = (m + NB - 1) / NB;
  const int64_t half = ((nb + 1) / 2) * NB;
  return (size_t)(half * half) * sizeof(double);
}

extern "C" int geobo_potrf_ctx_create(void** ctx) {
  if (!ctx) return GEOBO_E_ARG;
  PotrfCtx* c = new (std::nothrow) PotrfCtx();
  if (!c) return GEOBO_E_LAUNCH;
  int ns = 0, ne = 0;   // streams / events created so far
  bool ok = hipGetDevice(&c->dev) == hipSuccess;
  while (ok && ns < 3) { ok = hipStreamCreateWithFlags(&c->s[ns], hipStreamNonBlocking) == hipSuccess; ns += ok; }
  while (ok && ne < NEV) { ok = hipEventCreateWithFlags(&c->ev[ne], hipEventDisableTiming) == hipSuccess; ne += ok; }
  if (!ok) {
    for (int i = 0; i < ne; ++i) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < ns; ++i) (void)hipStreamDestroy(c->s[i]);
    delete c;
    return GEOBO_E_LAUNCH;
  }
  *ctx = c;
  return GEOBO_OK;
}

extern "C" int geobo_potrf_ctx_destroy(void* ctx) {
  if (!ctx) return GEOBO_OK;
  PotrfCtx* c = (PotrfCtx*)ctx;
  for (int i = 0; i < NEV; ++i) (void)hipEventDestroy(c->ev[i]);
  for (int i = 0; i < 3; ++i) (void)hipStreamDestroy(c->s[i]);
  delete c;
  return GEOBO_OK;
}

extern "C" int geobo_potrf_inv(int64_t m, double* A, int64_t ld, double* Linv, int64_t ldi, int* info, void* ws,
                               size_t ws_bytes, void* ctx, void* stream) {
  if (!A || !Linv || !info || !ws) return GEOBO_E_ARG;
  if (ctx) {   // the fork streams belong to one device: refuse a context made for another one
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != ((PotrfCtx*)ctx)->dev) return GEOBO_E_ARG;
  }
  if (m <= 0 || m % NB || (ld & 1) || (ldi & 1) || ld < m || ldi < m) return GEOBO_E_ALIGN;
  if (ws_bytes < geobo_potrf_ws_bytes(m)) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) return GEOBO_E_LAUNCH;
  if (hipMemset2DAsync(Linv, (size_t)ldi * sizeof(double), 0, (size_t)m * sizeof(double), (size_t)m, st) != hipSuccess)
    return GEOBO_E_LAUNCH;
  // Right-looking factorisation with ONE STEP OF LOOK-AHEAD when a fork context is given.  The trailing update of step c is
  // split into (a) the next panel's column block and (b) everything right of it; (b) runs on the context's first stream while
  // the caller's stream already factors the next diagonal block and solves the next panel, which need (a)_c and (b)_(c-2) only:
  //     caller's stream:  [wait (b)_(c-2)] potf2(c), panel(c) -> event P_c;  [wait (b)_(c-1): same column block] (a)_c
  //     side stream:      [wait P_c] (b)_c -> event B_c
  // The two small latency-bound launches of a step (86 + 74 us at M = 8448) start while the previous step's (b) still runs.
  // Measured gain is small (22.8 -> 22.3 ms): (b) occupies every CU with one workgroup, so the critical-path kernels wait for
  // a tile of it to retire (~80 us; stream priorities do not change that) -- the chain potf2 + panel + (a) = 66 x ~200 us is what
  // remains.  Every element still receives the same updates in the same order: bit-identical to the serial schedule.
  const PotrfCtx* pc = (const PotrfCtx*)ctx;
  hipStream_t side = pc ? pc->s[0] : st;
  const hipEvent_t* Pev = pc ? pc->ev + 6 : nullptr;
  const hipEvent_t* Bev = pc ? pc->ev + 10 : nullptr;
  int step = 0, last_b = -1, prev_b = -1;   // steps whose (b) was launched most recently
  if (pc) {   // the side stream starts after everything already queued on the caller's stream (the memsets above, the producer of A)
    if (hipEventRecord(Pev[3], st) != hipSuccess || hipStreamWaitEvent(side, Pev[3], 0) != hipSuccess) return GEOBO_E_LAUNCH;
  }
  for (int64_t kb = 0; kb < m; kb += NB, ++step) {
    if (pc && prev_b >= 0 && hipStreamWaitEvent(st, Bev[prev_b & 3], 0) != hipSuccess) return GEOBO_E_LAUNCH;
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, st, A + kb * ld + kb, ld, Linv + kb * ldi + kb, ldi,
                       (int)kb, info);
    if (hipGetLastError() != hipSuccess) return GEOBO_E_LAUNCH;
    const int64_t rem = m - kb - NB;
    if (rem > 0) {
      double* P = A + (kb + NB) * ld + kb;
      // panel and (a) are short and run next to (b) of the previous step: 128-row tiles find room as soon as HALF a CU drains
      // (a 512-thread workgroup waits for a whole CU: behind 256-thread (b) tiles that only happens when (b) ends)
      const int crit = pc ? GEOBO_GEMM_SMALL_TILES : 0;
      int rc = geobo_gemm_nt(rem, NB, NB, 1.0, P, ld, Linv + kb * ldi + kb, ldi, 0.0, P, ld, crit, 0, stream);
      if (rc) return rc;
      if (!pc) {
        rc = geobo_gemm_nt(rem, rem, NB, -1.0, P, ld, P, ld, 1.0, A + (kb + NB) * ld + (kb + NB), ld, 1, 0, stream);
        if (rc) return rc;
        continue;
      }
      if (hipEventRecord(Pev[step & 3], st) != hipSuccess) return GEOBO_E_LAUNCH;
      if (last_b >= 0 && hipStreamWaitEvent(st, Bev[last_b & 3], 0) != hipSuccess) return GEOBO_E_LAUNCH;
      // (a)_c: column block c+1, rows >= c+1
      rc = geobo_gemm_nt(rem, NB, NB, -1.0, P, ld, P, ld, 1.0, A + (kb + NB) * ld + (kb + NB), ld, crit, 0, stream);
      if (rc) return rc;
      prev_b = last_b;
      if (rem > NB) {
        // (b)_c: columns >= c+2, lower tiles
        if (hipStreamWaitEvent(side, Pev[step & 3], 0) != hipSuccess) return GEOBO_E_LAUNCH;
        double* P2 = P + NB * ld;
        rc = geobo_gemm_nt(rem - NB, rem - NB, NB, -1.0, P2, ld, P2, ld, 1.0, A + (kb + 2 * NB) * ld + (kb + 2 * NB), ld, 1, 0, side);
        if (rc) return rc;
        if (hipEventRecord(Bev[step & 3], side) != hipSuccess) return GEOBO_E_LAUNCH;
        last_b = step;
      }
    }
  }
  if (pc && last_b >= 0 && hipStreamWaitEvent(st, Bev[last_b & 3], 0) != hipSuccess) return GEOBO_E_LAUNCH;
  InvCtx c;
  c.L = A; c.ld = ld; c.Linv = Linv; c.ldi = ldi; c.ws = (double*)ws; c.ws_doubles = geobo_potrf_ws_bytes(m) / sizeof(double);
  c.s[0] = st;
  c.nev = 0;
  c.fork = ctx != nullptr;
  if (c.fork) {
    const PotrfCtx* pc = (const PotrfCtx*)ctx;
    for (int i = 0; i < 3; ++i) c.s[i + 1] = pc->s[i];
    for (int i = 0; i < 6; ++i) c.ev[i] = pc->ev[i];
  }
  return build_inverse(c, 0, (int)(m / NB), 0, 0);
}

extern "C" int geobo_trmv_stats(int64_t m, const double* Linv, int64_t ldi, const double* y, const double* L,
                                int64_t ld, double* u, double* stats, void* stream) {
  if (!Linv || !y || !L || !u || !stats || m <= 0) return GEOBO_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(trmv_lower_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, m, Linv, ldi, y, u);
  hipLaunchKernelGGL(logl_stats_kernel, dim3(1), dim3(256), 0, st, m, u, L, ld, stats);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
