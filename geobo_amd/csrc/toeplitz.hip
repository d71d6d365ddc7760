// toeplitz.hip -- the y-axis stage of the structured covariance product  AK = A_s K_sj  (kernels.py:158-195 builds
// K_sj densely; on the grid of kernels.py:27-42 it is symmetric three-level Toeplitz).
//
// After the real-DFT passes over z and x (geobo_gemm_batched against fixed cosine/sine matrices) every mode
// c = (ox, oz) of a sensor row is an independent ny-vector, and the covariance acts on it as a symmetric Toeplitz
// matrix  T_c[y, y'] = t_c(|y - y'|),  t_c = the (x, z)-transform of the lattice table.  Carrying the y axis through
// the spectrum as well (transform to 2ny, scale, transform back) costs five sweeps over a 2ny-long spectrum; applying
// T_c directly costs ONE read and one write of the ny-long data:
//
//     out_j[r][y][c] = sum_{y'} t_{j,c}(|y - y'|) in[r][y'][c]        j = property block
//
// The matrix differs per mode, so this is not a GEMM (no operand is shared along a tile edge): it runs on the fp64
// VALU, one mode per lane (c is the contiguous index: every load/store is a 512-byte wave access), the NY table
// values of the lane held in registers for all the rows the wave sweeps, NY*OC fused multiply-adds per chunk of OC
// outputs with every index static.  8 flop per byte of traffic; VALU-bound at ~NY^2 * 4 cycles per wave-row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "geobo_hip.h"

namespace {

struct ToeplitzArgs {
  const double* in;       // [R][NY][C]
  const double* tab[2];   // [NY][C] per property block
  double* out[2];         // [R][y1-y0][C] per property block
  int64_t C, R;
  int nprop, y0, y1, row_step;
};

// Buffer addressing: wave-uniform resource (row base, 4 SGPRs) + scalar byte offset (y * C * 8) + one per-lane
// VGPR offset (lane * 8) -- no 64-bit per-lane address arithmetic next to the table and the accumulators.
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(*static_cast<rsrc_t*>(nullptr), 0, 0, 0));
__device__ __forceinline__ rsrc_t make_rsrc(const double* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double ld_lane(rsrc_t rs, unsigned lane8, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane8, soff, 0));
}
__device__ __forceinline__ void st_lane(rsrc_t rs, unsigned lane8, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, lane8, soff, 0);
}

// One group of GS = 8 inputs of the row sweep: q-th of NQ = (NY/OC) * (NY/8) groups (chunk h = q / NG of OC outputs,
// input group gq = q % NG).  A template (not a loop) so that every table index is a compile-time constant.
template <int NY, int OC, int Q>
__device__ __forceinline__ void toeplitz_group(const ToeplitzArgs& g, const double (&t)[NY], double (&acc)[OC],
                                               double (&xb)[2][8], rsrc_t src, rsrc_t nxt, rsrc_t dst, int C8,
                                               unsigned lane8, int y0, int y1) {
  constexpr int GS = 8, NG = NY / GS, NQ = (NY / OC) * NG;
  constexpr int h = Q / NG, gq = Q % NG;
  // launder the row pitch: the byte offsets y * C8 are recomputed where they are used (a handful of scalar ops)
  // instead of being hoisted out of the row loop into >128 live SGPRs (same for the 64 store predicates)
  asm volatile("" : "+s"(C8), "+s"(y0), "+s"(y1));
  {  // prefetch the next group: of this chunk, of the next chunk (re-reads the row: L1/L2 hits), or of the next row
    const rsrc_t rs = (Q + 1 == NQ) ? nxt : src;
    constexpr int y_first = ((Q + 1) % NG) * GS;
#pragma unroll
    for (int i = 0; i < GS; ++i) xb[(Q + 1) & 1][i] = ld_lane(rs, lane8, (y_first + i) * C8);
  }
  // compiler barrier: keeps the scheduler from hoisting every load of the row (which would spill the table)
  asm volatile("" ::: "memory");
  if (gq == 0) {
#pragma unroll
    for (int o = 0; o < OC; ++o) acc[o] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    const int yp = gq * GS + i;
    const double x = xb[Q & 1][i];
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      const int d = h * OC + o - yp;
      acc[o] = __builtin_fma(t[d < 0 ? -d : d], x, acc[o]);
    }
  }
  if (gq == NG - 1) {
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      const int y = h * OC + o;
      if (y >= y0 && y < y1) st_lane(dst, lane8, (y - y0) * C8, acc[o]);
    }
  }
  if constexpr (Q + 1 < NQ) toeplitz_group<NY, OC, Q + 1>(g, t, acc, xb, src, nxt, dst, C8, lane8, y0, y1);
}

template <int NY, int OC>
__global__ void __launch_bounds__(256, 2) toeplitz_y_kernel(ToeplitzArgs g) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane8 = (unsigned)lane * 8u;
  const int prop = w % g.nprop, rlane = w / g.nprop;
  const int64_t C = g.C, c0 = (int64_t)blockIdx.x * 64;
  const int C8 = (int)(C * 8), in_bytes = NY * C8, out_bytes = (g.y1 - g.y0) * C8;
  double t[NY];
  {
    const rsrc_t tr = make_rsrc(g.tab[prop] + c0, in_bytes);
#pragma unroll
    for (int d = 0; d < NY; ++d) t[d] = ld_lane(tr, lane8, d * C8);
  }
  const int64_t rstep = (int64_t)gridDim.y * g.row_step;
  int64_t r = (int64_t)blockIdx.y * g.row_step + rlane;
  if (r >= g.R) return;
  const int64_t ostep = (int64_t)(g.y1 - g.y0) * C;  // output rows hold the slab [y0, y1) only
  double* po = g.out[prop] + c0 + r * ostep;
  const double* ps = g.in + r * NY * C + c0;
  rsrc_t src = make_rsrc(ps, in_bytes);
  double xb[2][8];  // inputs stream through a two-deep ring of 8-value groups
#pragma unroll
  for (int i = 0; i < 8; ++i) xb[0][i] = ld_lane(src, lane8, i * C8);
  for (; r < g.R; r += rstep) {
    if (r + rstep < g.R) ps += rstep * NY * C;
    const rsrc_t nxt = make_rsrc(ps, in_bytes);
    const rsrc_t dst = make_rsrc(po, out_bytes);
    double acc[OC];
    toeplitz_group<NY, OC, 0>(g, t, acc, xb, src, nxt, dst, C8, lane8, g.y0, g.y1);
    src = nxt;
    po += rstep * ostep;
  }
}

template <int NY, int OC>
int launch(const ToeplitzArgs& g, hipStream_t st) {
  const int rlanes = 4 / g.nprop;
  int64_t gy = (g.R + rlanes - 1) / rlanes;
  const int64_t want = (4096 + g.C / 64 - 1) / (g.C / 64);  // ~16 workgroups per CU over the whole launch
  if (gy > want) gy = want;
  if (gy < 1) gy = 1;
  ToeplitzArgs a = g;
  a.row_step = rlanes;
  hipLaunchKernelGGL((toeplitz_y_kernel<NY, OC>), dim3((unsigned)(g.C / 64), (unsigned)gy), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

}  // namespace

extern "C" int geobo_toeplitz_y(int ny, int64_t C, int64_t R, int nprop, const double* in, const double* tab0,
                                const double* tab1, double* out0, double* out1, int y0, int y1, void* stream) {
  if (!in || !tab0 || !out0 || (nprop == 2 && (!tab1 || !out1))) return GEOBO_E_ARG;
  if (nprop < 1 || nprop > 2 || R <= 0 || y0 < 0 || y1 > ny || y1 <= y0) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (int64_t)ny * C * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  ToeplitzArgs g;
  g.in = in; g.tab[0] = tab0; g.tab[1] = nprop == 2 ? tab1 : tab0; g.out[0] = out0; g.out[1] = nprop == 2 ? out1 : out0;
  g.C = C; g.R = R; g.nprop = nprop; g.y0 = y0; g.y1 = y1; g.row_step = 1;
  hipStream_t st = (hipStream_t)stream;
  switch (ny) {
    case 16: return launch<16, 16>(g, st);
    case 32: return launch<32, 32>(g, st);
    case 48: return launch<48, 24>(g, st);
    case 64: {
      static const int oc = getenv("GEOBO_TOEP_OC") ? atoi(getenv("GEOBO_TOEP_OC")) : 32;
      return oc == 16 ? launch<64, 16>(g, st) : launch<64, 32>(g, st);
    }
    default: return GEOBO_E_UNSUPPORTED;  // longer y axes: carry y through the spectrum instead (spectral.py)
  }
}
