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
#include <atomic>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

struct ToeplitzArgs {
  const double* in;       // [R][NY][S]: C valid modes per plane, planes S >= C doubles apart
  const double* tab[2];   // [NY][C] per property block
  double* out[2];         // [R][y1-y0][S] per property block
  int64_t C, S, R;        // S: plane stride of in and out.  A power-of-two stride (C = 16384 doubles = 128 KiB at 64^3) puts the ny
                          // planes a lane walks on the same HBM channels: 3.95 TB/s for a copy with this access pattern against
                          // 5.2 TB/s with S = C + 256 (profiles/r03_hbm_copy_runs.txt)
  int nprop, y0, y1;
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

// s_waitcnt immediate for vmcnt(v) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14)
constexpr int vmcnt_only(int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// NY x OC multiply-adds of one wave-row, all table indices compile-time constants: outputs O0 .. O0 + OC - 1 of the row.
// (Rounds 2-4 shared ONE body between the two halves of the outputs -- T is persymmetric, the second half ran the same code on
// the row read backwards, `xstep` < 0; the kernels of this file now take the IMM form below.)
// The LDS reads of the row are inline asm: for LDS reads it can see, the compiler first waits for EVERY outstanding LDS-DMA
// (vmcnt(0): it cannot tell the two ring halves apart), i.e. for the NEXT row that was requested a moment ago -- no prefetch
// at all.  Groups of GX inputs, double buffered: group g+1 is requested before the FMAs of group g.
constexpr int GX = 4;

// IMM (round 5): the inputs are read in their stored order -- row y' at xaddr + 512 y', an instruction immediate -- and the chunk of
// outputs is named by O0 alone (one body per chunk, selected by a wave-uniform branch).  The runtime-stride form shares ONE body between the
// two halves of the outputs (the second half reads the row backwards: T is persymmetric) at the price of a 32-bit address add per read
// -- a VALU instruction like any other on this machine: 3 % of the issue slots of a half wave, 10 % of a quarter wave.
template <int NY, int OC, int G, int O0 = 0, bool IMM = false>
__device__ __forceinline__ void toeplitz_group(const double (&t)[NY], double (&acc)[OC], double (&xb)[2][GX], unsigned xaddr, int xstep) {
  constexpr int NGX = NY / GX;
  if constexpr (G == 0) {
    // the first group is requested HERE, inside the (possibly branched-to) body that consumes it: registers an inline-asm read is
    // still filling must not cross a branch -- the compiler may copy them at the join before the data has landed
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      if constexpr (IMM) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[0][i]) : "v"(xaddr), "n"(i * 512));
      else asm volatile("ds_read_b64 %0, %1" : "=v"(xb[0][i]) : "v"(xaddr + (unsigned)(i * xstep)));
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the groups apart: interleaving them costs registers the table needs
  if constexpr (G + 1 < NGX) {
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      if constexpr (IMM) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[(G + 1) & 1][i]) : "v"(xaddr), "n"(((G + 1) * GX + i) * 512));
      else asm volatile("ds_read_b64 %0, %1" : "=v"(xb[(G + 1) & 1][i]) : "v"(xaddr + (unsigned)(((G + 1) * GX + i) * xstep)));
    }
    // group G (requested one group earlier) is complete once only the GX reads just issued are outstanding
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]) : "n"(GX));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]));
  }
#pragma unroll
  for (int i = 0; i < GX; ++i) {
    const int yp = G * GX + i;
    const double x = xb[G & 1][i];
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      const int d = O0 + o - yp;
      acc[o] = (yp == 0) ? t[d < 0 ? -d : d] * x : __builtin_fma(t[d < 0 ? -d : d], x, acc[o]);
    }
  }
  if constexpr (G + 1 < NGX) toeplitz_group<NY, OC, G + 1, O0, IMM>(t, acc, xb, xaddr, xstep);
}

// Workgroup = 2 * nprop * Q waves sharing ONE row at a time: wave (prop, part q of a half, half).  The next row streams into the
// other half of a 2 x NY x 512 B LDS ring by LDS-DMA (no registers, issued a whole row -- ~7 us of arithmetic -- ahead), so the
// VALU never waits on HBM and each input is fetched once per workgroup for both property blocks.  Q = 2 (a single property block:
// every half of the outputs split between two waves) keeps four waves per workgroup -- the LDS ring allows two workgroups per CU,
// and with two waves each the one-block launches of the symmetric A K plan ran at 3.5 TB/s against 4.4 for the two-block ones.
template <int NY, int Q>
__global__ void __launch_bounds__(256, 2) toeplitz_y_kernel(ToeplitzArgs g) {
  constexpr int OC = NY / 2 / Q;
  __shared__ __attribute__((aligned(16))) double xs[2][NY][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = 2 * g.nprop * Q;
  const unsigned lane8 = (unsigned)lane * 8u;
  const int prop = w / (2 * Q), half = w & 1, qq = (w >> 1) % Q;
  const int64_t C = g.S, c0 = (int64_t)blockIdx.x * 64;     // C: plane stride of the data from here on (the table's is g.C)
  const int C8 = (int)(C * 8), out_bytes = (g.y1 - g.y0) * C8;
  double t[NY];
  {
    const int T8 = (int)(g.C * 8);
    const rsrc_t tr = make_rsrc(g.tab[prop] + c0, NY * T8);
#pragma unroll
    for (int d = 0; d < NY; ++d) t[d] = ld_lane(tr, lane8, d * T8);
  }
  int64_t r = blockIdx.y;
  if (r >= g.R) return;
  const int64_t rstep = gridDim.y;
  const int64_t ostep = (int64_t)(g.y1 - g.y0) * C;  // output rows hold the slab [y0, y1) only
  double* po = g.out[prop] + c0 + r * ostep;
  const double* ps = g.in + r * NY * C + c0;
  // one DMA instruction moves two y-planes (2 x 64 modes x 8 B): lanes 0-31 plane 2i, lanes 32-63 plane 2i+1
  const int64_t dma_lane = (int64_t)(lane >> 5) * C + (lane & 31) * 2;
  auto stage = [&](const double* row, int b) {
    for (int i = w; i < NY / 2; i += nw)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row + (int64_t)(2 * i) * C + dma_lane), (lds_ptr_t)&xs[b][2 * i][0], 16, 0, 0);
  };
  stage(ps, 0);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this wave's planes have landed
  __syncthreads();
  const int ch = half * Q + qq;          // this wave's chunk of OC outputs: y = ch * OC + o (one body per chunk: immediate-offset reads)
  int b = 0;
  for (; r < g.R; r += rstep) {
    const bool more = r + rstep < g.R;
    if (more) {
      ps += rstep * NY * C;
      stage(ps, b ^ 1);
    }
    double acc[OC], xb[2][GX];
    const unsigned xaddr = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][0][0] + lane8;
    if constexpr (Q == 1) {
      if (ch == 0) toeplitz_group<NY, OC, 0, 0, true>(t, acc, xb, xaddr, 0);
      else toeplitz_group<NY, OC, 0, OC, true>(t, acc, xb, xaddr, 0);
    } else {
      if (ch == 0) toeplitz_group<NY, OC, 0, 0, true>(t, acc, xb, xaddr, 0);     // (the distances o - y' index the register table: static)
      else if (ch == 1) toeplitz_group<NY, OC, 0, OC, true>(t, acc, xb, xaddr, 0);
      else if (ch == 2) toeplitz_group<NY, OC, 0, 2 * OC, true>(t, acc, xb, xaddr, 0);
      else toeplitz_group<NY, OC, 0, 3 * OC, true>(t, acc, xb, xaddr, 0);
    }
    // pin the sums here: otherwise the tail of every sum is sunk into its (predicated) store block, which keeps the last
    // inputs and half the table live across all of them (spills; scratch reloads are VMEM and drain the prefetch)
#pragma unroll
    for (int o = 0; o < OC; ++o) asm volatile("" : "+v"(acc[o]));
    // the next row was requested a whole row of arithmetic ago: drain it BEFORE the stores (free), so that the barrier below
    // does not have to wait for the stores as well
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
    const rsrc_t dst = make_rsrc(po, out_bytes);
    int y0 = g.y0, y1 = g.y1, pitch = C8;
    // laundered per row: keeps the NY/2 store offsets and predicates from being hoisted into ~100 live SGPRs
    asm volatile("" : "+s"(y0), "+s"(y1), "+s"(pitch));
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      const int y = ch * OC + o;
      if (y >= y0 && y < y1) st_lane(dst, lane8, (y - y0) * pitch, acc[o]);
    }
    po += rstep * ostep;
    __builtin_amdgcn_s_barrier();   // every wave has drained its share of the next row and is done reading this one
    b ^= 1;
  }
}

// ---- two-term rows (round 4): V = Z_g K_0j + Z_m K_1j in ONE pass -------------------------------------------------------------------
// The rows of the transposed posterior behind the gravity block are sums of two covariance products (engine._posterior_zpath).  As two
// launches of the kernel above each term writes its own output spectrum (2 x 4.3 GB per 256-row batch at 64^3) and the inverse
// transform reads both.  Here one workgroup of EIGHT waves -- (term, property block, half of the outputs) -- shares the two input rows
// of a sensor row; the four waves of the second term hand their 32 partial sums per lane to their partners through the input buffer
// that has just been consumed (all of it is dead after the row's barrier: 4 waves x 32 x 512 B = the 64 KiB of the two staged rows), the
// partners add and store ONE output per property block.  Same FMAs as before (the stage is bound by the fp64 VALU either way); what
// goes away is one output stream of the y stage and one input stream of the inverse transform.  LDS: 2 stages x 2 terms x NY x 512 B
// = 128 KiB, one workgroup per CU (the same eight waves per CU as two workgroups of the kernel above).
struct Toeplitz2Args {
  const double* in[2];        // [R][NY][S] per term
  const double* tab[2][2];    // [term][block]: [NY][C]
  double* out[2];             // [R][NY][S] per property block
  int64_t C, S, R;
};

template <int NY>
__global__ void __launch_bounds__(512, 1) toeplitz_y2_kernel(Toeplitz2Args g) {
  constexpr int OC = NY / 2;
  extern __shared__ __attribute__((aligned(16))) double xs2_dyn[];          // [2 stages][2 terms][NY][64]
  double (*xs)[2][NY][64] = reinterpret_cast<double (*)[2][NY][64]>(xs2_dyn);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int term = w >> 2, prop = (w >> 1) & 1, half = w & 1;
  const unsigned lane8 = (unsigned)lane * 8u;
  const int64_t C = g.S, c0 = (int64_t)blockIdx.x * 64;
  const int C8 = (int)(C * 8);
  double t[NY];
  {
    const int T8 = (int)(g.C * 8);
    const rsrc_t tr = make_rsrc(g.tab[term][prop] + c0, NY * T8);
#pragma unroll
    for (int d = 0; d < NY; ++d) t[d] = ld_lane(tr, lane8, d * T8);
  }
  int64_t r = blockIdx.y;
  if (r >= g.R) return;
  const int64_t rstep = gridDim.y, rowlen = (int64_t)NY * C;
  double* po = g.out[prop] + c0 + r * rowlen;
  const double* ps0 = g.in[0] + r * rowlen + c0;
  const double* ps1 = g.in[1] + r * rowlen + c0;
  const int64_t dma_lane = (int64_t)(lane >> 5) * C + (lane & 31) * 2;   // one DMA instruction moves two y-planes of 64 modes
  auto stage = [&](const double* row0, const double* row1, int b) {
    for (int i = w; i < NY / 2; i += 8) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row0 + (int64_t)(2 * i) * C + dma_lane), (lds_ptr_t)&xs[b][0][2 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row1 + (int64_t)(2 * i) * C + dma_lane), (lds_ptr_t)&xs[b][1][2 * i][0], 16, 0, 0);
    }
  };
  stage(ps0, ps1, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  int b = 0;
  for (; r < g.R; r += rstep) {
    const bool more = r + rstep < g.R;
    if (more) {
      ps0 += rstep * rowlen;
      ps1 += rstep * rowlen;
      stage(ps0, ps1, b ^ 1);
    }
    double acc[OC], xb[2][GX];
    const unsigned xaddr = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][term][0][0] + lane8;
    if (half == 0) toeplitz_group<NY, OC, 0, 0, true>(t, acc, xb, xaddr, 0);         // outputs y = half * NY/2 + o, immediate-offset reads
    else toeplitz_group<NY, OC, 0, OC, true>(t, acc, xb, xaddr, 0);
#pragma unroll
    for (int o = 0; o < OC; ++o) asm volatile("" : "+v"(acc[o]));
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));      // this wave's share of the next rows has landed (and last row's stores are out)
    __builtin_amdgcn_s_barrier();                   // (1) every wave is done reading the two rows of stage b: their 64 KiB are free
    // exchange area = stage b, [pair = 2 prop + half][o][lane]
    double* const ex = &xs[b][0][0][0] + (size_t)((2 * prop + half) * OC) * 64 + lane;
    if (term == 1) {
#pragma unroll
      for (int o = 0; o < OC; ++o) ex[o * 64] = acc[o];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // (2) the second term's partial sums are in LDS
    if (term == 0) {
#pragma unroll
      for (int o = 0; o < OC; ++o) acc[o] += ex[o * 64];
      const rsrc_t dst = make_rsrc(po, NY * C8);
      int pitch = C8;
      asm volatile("" : "+s"(pitch));
#pragma unroll
      for (int o = 0; o < OC; ++o) st_lane(dst, lane8, (half * OC + o) * pitch, acc[o]);
    }
    po += rstep * rowlen;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // (3) the exchange has been read: stage b may be refilled by the next iteration
    b ^= 1;
  }
}

template <int NY>
int launch2(const Toeplitz2Args& g, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * 2 * NY * 64 * sizeof(double);
  auto kern = toeplitz_y2_kernel<NY>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  int64_t gy = g.R < 2 ? g.R : 2;      // one workgroup per CU: 256 column blocks x 2 at 64^3 = two rounds of the chip
  while ((g.C / 64) * gy < 512 && gy < g.R) ++gy;
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.C / 64), (unsigned)gy), dim3(512), lds, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

// ---- two-term rows with a SHARED cross block (round 5): three products instead of four --------------------------------------------------
// create_cov's prior is symmetric, K_10 = K_01 (kernels.py:181-195: block (i, j) = W[i][j] k_cross(l_i, l_j), W symmetric, k_cross even in
// the exchange of its lengths), so the two-term rows  V_0 = Z_g K_00 + Z_m K_10,  V_1 = Z_g K_01 + Z_m K_11  share one Toeplitz block X:
//     V_0 = T(D0) x_g + T(X)(x_g + x_m),     V_1 = T(D1) x_m + T(X)(x_g + x_m),     D0 = t_00 - t_01,  D1 = t_11 - t_01
// -- three NY x NY products per mode instead of four.  Eight waves again, but of two sizes: waves 0-3 = (D0 on x_g | D1 on x_m) x (half of
// the outputs), NY x NY/2 multiply-adds each as before; waves 4-7 = the shared product on x_g + x_m, split into QUARTERS of the outputs
// (half, part), NY x NY/4 multiply-adds each.  Wave w runs on SIMD w % 4, so every SIMD carries one wave of each size: 3/4 of the issue
// slots of the four-product kernel.  The quarter waves hand their sums to both full waves of their half through a dedicated 32 KiB of
// LDS (2 stages x 2 terms x NY x 512 B + 2 x NY/2 x 512 B = 160 KiB at NY = 64: the whole LDS of a CU, one workgroup per CU as before):
// two barriers per row (sums written | sums read) instead of three.
struct Toeplitz2sArgs {
  const double* in[2];        // [R][NY][S] per term
  const double* tab[3];       // D0, X, D1: [NY][C]
  double* out[2];             // [R][NY][S] per property block
  int64_t C, S, R;
};

// toeplitz_group for inputs that are the SUM of the two staged rows (x = xs[term 0] + xs[term 1]: two LDS reads and one addition per input)
// (immediate-offset reads: row y' of the first term at xaddr + 512 y', of the second term NY rows behind it)
template <int NY, int OC, int G, int O0>
__device__ __forceinline__ void toeplitz_group_sum(const double (&t)[NY], double (&acc)[OC], double (&xb)[2][2 * GX], unsigned xaddr) {
  constexpr int NGX = NY / GX;
  if constexpr (G == 0) {
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[0][i]) : "v"(xaddr), "n"(i * 512));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[0][GX + i]) : "v"(xaddr), "n"((NY + i) * 512));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (G + 1 < NGX) {
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[(G + 1) & 1][i]) : "v"(xaddr), "n"(((G + 1) * GX + i) * 512));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[(G + 1) & 1][GX + i]) : "v"(xaddr), "n"((NY + (G + 1) * GX + i) * 512));
    }
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]), "+v"(xb[G & 1][4]), "+v"(xb[G & 1][5]),
                   "+v"(xb[G & 1][6]), "+v"(xb[G & 1][7])
                 : "n"(2 * GX));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]), "+v"(xb[G & 1][4]), "+v"(xb[G & 1][5]),
                   "+v"(xb[G & 1][6]), "+v"(xb[G & 1][7]));
  }
#pragma unroll
  for (int i = 0; i < GX; ++i) {
    const int yp = G * GX + i;
    const double x = xb[G & 1][i] + xb[G & 1][GX + i];
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      const int d = O0 + o - yp;
      acc[o] = (yp == 0) ? t[d < 0 ? -d : d] * x : __builtin_fma(t[d < 0 ? -d : d], x, acc[o]);
    }
  }
  if constexpr (G + 1 < NGX) toeplitz_group_sum<NY, OC, G + 1, O0>(t, acc, xb, xaddr);
}

template <int NY>
__global__ void __launch_bounds__(512, 1) toeplitz_y2s_kernel(Toeplitz2sArgs g) {
  static_assert(GX == 4 && NY % 16 == 0, "shape");
  constexpr int OCF = NY / 2, OCQ = NY / 4;
  extern __shared__ __attribute__((aligned(16))) double xs2s_dyn[];         // [2 stages][2 terms][NY][64] | exchange [2 halves][NY/2][64]
  double (*xs)[2][NY][64] = reinterpret_cast<double (*)[2][NY][64]>(xs2s_dyn);
  double* const exch = xs2s_dyn + (size_t)2 * 2 * NY * 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (the pairing matters: with equal roles on the two waves of a SIMD -- full + full, quarter + quarter -- the launch takes 2.89 ms
  // instead of 2.59)
  const bool shared = w >= 4;
  const int pos = w & 3, half = pos & 1, sel = pos >> 1;       // full waves: sel = block (0: D0 on x_g, 1: D1 on x_m); quarter waves: part
  const unsigned lane8 = (unsigned)lane * 8u;
  const int64_t C = g.S, c0 = (int64_t)blockIdx.x * 64;
  const int C8 = (int)(C * 8);
  double t[NY];
  {
    const int T8 = (int)(g.C * 8);
    const rsrc_t tr = make_rsrc(g.tab[shared ? 1 : 2 * sel] + c0, NY * T8);
#pragma unroll
    for (int d = 0; d < NY; ++d) t[d] = ld_lane(tr, lane8, d * T8);
  }
  int64_t r = blockIdx.y;
  if (r >= g.R) return;
  const int64_t rstep = gridDim.y, rowlen = (int64_t)NY * C;
  double* po = g.out[sel] + c0 + r * rowlen;                   // (full waves only)
  const double* ps0 = g.in[0] + r * rowlen + c0;
  const double* ps1 = g.in[1] + r * rowlen + c0;
  const int64_t dma_lane = (int64_t)(lane >> 5) * C + (lane & 31) * 2;   // one DMA instruction moves two y-planes of 64 modes
  auto stage = [&](const double* row0, const double* row1, int b) {
    for (int i = w; i < NY / 2; i += 8) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row0 + (int64_t)(2 * i) * C + dma_lane), (lds_ptr_t)&xs[b][0][2 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row1 + (int64_t)(2 * i) * C + dma_lane), (lds_ptr_t)&xs[b][1][2 * i][0], 16, 0, 0);
    }
  };
  stage(ps0, ps1, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  double* const ex = exch + (size_t)(half * OCF) * 64 + lane;   // [o of the half][lane]: output y = half * NY/2 + o
  int b = 0;
  for (; r < g.R; r += rstep) {
    const bool more = r + rstep < g.R;
    if (more) {
      ps0 += rstep * rowlen;
      ps1 += rstep * rowlen;
      stage(ps0, ps1, b ^ 1);
    }
    // (immediate-offset reads, one body per chunk of outputs: no address arithmetic on the vector pipe)
    if (!shared) {
      double acc[OCF], xb[2][GX];
      const unsigned xaddr = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][sel][0][0] + lane8;
      if (half == 0) toeplitz_group<NY, OCF, 0, 0, true>(t, acc, xb, xaddr, 0);
      else toeplitz_group<NY, OCF, 0, OCF, true>(t, acc, xb, xaddr, 0);
#pragma unroll
      for (int o = 0; o < OCF; ++o) asm volatile("" : "+v"(acc[o]));
      __builtin_amdgcn_s_waitcnt(vmcnt_only(0));    // this wave's share of the next rows has landed (and last row's stores are out)
      __builtin_amdgcn_s_barrier();                 // (A) the shared product's sums are in the exchange area; stage b is free
#pragma unroll
      for (int o = 0; o < OCF; ++o) acc[o] += ex[o * 64];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // (B) the exchange area has been read: the next row's sums may overwrite it
      const rsrc_t dst = make_rsrc(po, NY * C8);
      int pitch = C8;
      asm volatile("" : "+s"(pitch));
#pragma unroll
      for (int o = 0; o < OCF; ++o) st_lane(dst, lane8, (half * OCF + o) * pitch, acc[o]);
      po += rstep * rowlen;
    } else {
      double acc[OCQ], xb[2][2 * GX];
      const unsigned xa0 = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][0][0][0] + lane8;
      if (pos == 0) toeplitz_group_sum<NY, OCQ, 0, 0>(t, acc, xb, xa0);                   // pos = half + 2 part: outputs y = half * NY/2 + part * NY/4 + o
      else if (pos == 2) toeplitz_group_sum<NY, OCQ, 0, OCQ>(t, acc, xb, xa0);
      else if (pos == 1) toeplitz_group_sum<NY, OCQ, 0, 2 * OCQ>(t, acc, xb, xa0);
      else toeplitz_group_sum<NY, OCQ, 0, 3 * OCQ>(t, acc, xb, xa0);
#pragma unroll
      for (int o = 0; o < OCQ; ++o) asm volatile("" : "+v"(acc[o]));
      __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
      double* const exq = ex + (size_t)(sel * OCQ) * 64;
#pragma unroll
      for (int o = 0; o < OCQ; ++o) exq[o * 64] = acc[o];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // (A)
      __builtin_amdgcn_s_barrier();                 // (B)
    }
    b ^= 1;
  }
}

template <int NY>
int launch2s(const Toeplitz2sArgs& g, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * 2 * NY + NY) * 64 * sizeof(double);
  static_assert(lds <= 163840, "LDS");
  auto kern = toeplitz_y2s_kernel<NY>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  int64_t gy = g.R < 2 ? g.R : 2;      // one workgroup per CU: 256 column blocks x 2 at 64^3 = two rounds of the chip
  while ((g.C / 64) * gy < 512 && gy < g.R) ++gy;
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.C / 64), (unsigned)gy), dim3(512), lds, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

// ---- long y axes (NY = 128: the 128^3 configuration): the NY table values of a lane no longer fit its registers ---------------
// A lane owns ONE mode and ONE HALF of the inputs (lanes 0-31: y' < NY/2, lanes 32-63 the same 32 modes with y' >= NY/2) and a chunk of
// OC consecutive outputs o = ob .. ob+OC-1.  The distances d = o - y' it meets form a window of NY/2 + OC - 1 consecutive values, so
//     acc[oo] += t[oo - i + NY/2 - 1] * x[i]        (i: input inside the half, oo: output inside the chunk)
// has static register indices into a window of 79 table values (NY = 128, OC = 16) loaded once per workgroup with |d| folded in at
// load time; the two halves of a mode meet in one __shfl_xor(.., 32) per output -- no second LDS buffer, no second barrier.
// Workgroup = 2 * nprop * cw waves (output chunk, property block, 32-mode group) sharing one input row at a time through the same
// LDS-DMA ring as above (2 x NY x 512 B = 128 KiB: one workgroup per CU); grid = (64-mode blocks, groups of cw output chunks, row
// groups).  cw = the output chunks that fit eight waves (4 for one property block, 2 for two, 1 for three): with one chunk per
// workgroup (round 3) a one-block launch ran TWO waves per CU and a full-height product staged every input row NY / OC = 8 times;
// with cw chunks it is eight waves and NY / (OC cw) stagings.
struct ToeplitzWinArgs {
  const double* in;       // [R][NY][S]
  const double* tab[3];   // [NY][C] per property block
  double* out[3];         // [R][y1-y0][S] per property block
  int64_t C, S, R;
  int nprop, y0, y1, cw;
  int ngroups, nbx;       // grid decode (round 5): chunk groups per row, 64-mode blocks
};

template <int NH, int OC, int G>
__device__ __forceinline__ void toeplitz_win_group(const double (&t)[NH + OC - 1], double (&acc)[OC], double (&xb)[2][GX], unsigned xaddr) {
  constexpr int NGX = NH / GX;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (G + 1 < NGX) {
#pragma unroll
    for (int i = 0; i < GX; ++i)
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[(G + 1) & 1][i]) : "v"(xaddr), "n"(((G + 1) * GX + i) * 512));
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]) : "n"(GX));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[G & 1][0]), "+v"(xb[G & 1][1]), "+v"(xb[G & 1][2]), "+v"(xb[G & 1][3]));
  }
#pragma unroll
  for (int i = 0; i < GX; ++i) {
    const int yi = G * GX + i;
    const double x = xb[G & 1][i];
#pragma unroll
    for (int o = 0; o < OC; ++o) acc[o] = (yi == 0) ? t[o - yi + NH - 1] * x : __builtin_fma(t[o - yi + NH - 1], x, acc[o]);
  }
  if constexpr (G + 1 < NGX) toeplitz_win_group<NH, OC, G + 1>(t, acc, xb, xaddr);
}

template <int NY, int OC, bool ACC>
__global__ void __launch_bounds__(512, 1) toeplitz_y_win_kernel(ToeplitzWinArgs g) {
  constexpr int NH = NY / 2, WIN = NH + OC - 1;
  static_assert(GX == 4 && NH % GX == 0 && NH * 512 < 65536, "shape");
  extern __shared__ __attribute__((aligned(16))) double xs_dyn[];      // [2][NY][64]
  double (*xs)[NY][64] = reinterpret_cast<double (*)[NY][64]>(xs_dyn);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = 2 * g.nprop * g.cw;
  const int prop = (w >> 1) % g.nprop, ck = (w >> 1) / g.nprop, mg = w & 1, h = lane >> 5;
  const int mode = 32 * mg + (lane & 31);                                // this lane's mode inside the 64-mode block
  // Workgroup -> (mode block, chunk group, row group).  The chunk groups of one (mode block, row group) stage the SAME input rows:
  // with a (mode block, chunk group, row group) grid they were 1024 workgroups apart in dispatch order -- never resident together, so
  // every group fetched its rows from HBM again (two blocks at ny = 128: four times; the launch ran at 3.5 TB/s of real traffic for
  // 1.8 of algorithmic).  1-D grid, block b on XCD b % 8 (observed dispatch rule): the groups of one (mode block, row group) are
  // consecutive workgroups of ONE XCD, run together and share the staged rows through its L2.
  // (mode-block counts that are not a multiple of 8 -- tests, tiny grids -- keep the groups adjacent without the XCD interleave)
  const bool x8 = (g.nbx & 7) == 0;
  const int bid = blockIdx.x, bs = x8 ? bid >> 3 : bid, nbxs = x8 ? g.nbx >> 3 : g.nbx;
  const int by = bs % g.ngroups, bt = bs / g.ngroups;
  const int bx = x8 ? (bt % nbxs) * 8 + (bid & 7) : bt % nbxs;
  const int64_t bz = bt / nbxs, gzn = (int64_t)gridDim.x / ((int64_t)g.nbx * g.ngroups);
  const int64_t S = g.S, c0 = (int64_t)bx * 64;
  const int S8 = (int)(S * 8), out_bytes = (g.y1 - g.y0) * S8;
  const int ob = g.y0 + OC * (by * g.cw + ck);                           // first output of this wave's chunk (may lie behind y1:
                                                                         // such a wave only helps staging the rows)
  double t[WIN];
  {
    const int dmin = ob - (NH * h + NH - 1);
    const double* tp = g.tab[prop] + c0 + mode;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      int d = dmin + k;
      d = d < 0 ? -d : d;
      d = d > NY - 1 ? NY - 1 : d;                                       // (outputs behind the last plane: never stored)
      t[k] = tp[(int64_t)d * g.C];
    }
  }
  int64_t r = bz;
  if (r >= g.R) return;
  const int64_t rstep = gzn;
  const int64_t ostep = (int64_t)(g.y1 - g.y0) * S;
  double* po = g.out[prop] + c0 + r * ostep + (int64_t)(ob - g.y0) * S;
  const double* ps = g.in + r * NY * S + c0;
  const int64_t dma_lane = (int64_t)(lane >> 5) * S + (lane & 31) * 2;   // one DMA instruction moves two y-planes of 64 modes
  auto stage = [&](const double* row, int b) {
    for (int i = w; i < NY / 2; i += nw)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(row + (int64_t)(2 * i) * S + dma_lane), (lds_ptr_t)&xs[b][2 * i][0], 16, 0, 0);
  };
  stage(ps, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const unsigned lane8 = (unsigned)(lane & 31) * 8u + (unsigned)(32 * mg) * 8u;
  int nout = g.y1 - ob;
  nout = nout > OC ? OC : nout;
  int b = 0;
  for (; r < g.R; r += rstep) {
    const bool more = r + rstep < g.R;
    if (more) {
      ps += rstep * NY * S;
      stage(ps, b ^ 1);
    }
    double acc[OC], xb[2][GX];
    const unsigned xaddr = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][NH * h][mode];
#pragma unroll
    for (int i = 0; i < GX; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xb[0][i]) : "v"(xaddr), "n"(i * 512));
    toeplitz_win_group<NH, OC, 0>(t, acc, xb, xaddr);
#pragma unroll
    for (int o = 0; o < OC; ++o) acc[o] += __shfl_xor(acc[o], 32);      // the other half of the inputs of the same mode
#pragma unroll
    for (int o = 0; o < OC; ++o) asm volatile("" : "+v"(acc[o]));
    __builtin_amdgcn_s_waitcnt(vmcnt_only(0));                           // the next row has landed (free here), then the stores
    if (h == 0) {
      const rsrc_t dst = make_rsrc(po, out_bytes);
      int n = nout, pitch = S8;
      asm volatile("" : "+s"(n), "+s"(pitch));
      if constexpr (ACC) {     // out += ...: the second term of a two-term row adds into the first term's spectrum (geobo_toeplitz_y3_add)
        double prev[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) prev[o] = (o < n) ? ld_lane(dst, lane8, o * pitch) : 0.0;
#pragma unroll
        for (int o = 0; o < OC; ++o) acc[o] += prev[o];
      }
#pragma unroll
      for (int o = 0; o < OC; ++o)
        if (o < n) st_lane(dst, lane8, o * pitch, acc[o]);
    }
    po += rstep * ostep;
    __builtin_amdgcn_s_barrier();
    b ^= 1;
  }
}

template <int NY, int OC, bool ACC = false>
int launch_win(const ToeplitzWinArgs& g, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * NY * 64 * sizeof(double);
  auto kern = toeplitz_y_win_kernel<NY, OC, ACC>;
  static std::atomic<uint64_t> attr_done{0};      // per-device "large-LDS attribute set" bits (include/geobo_hip.h, conventions)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if (!((attr_done.load(std::memory_order_acquire) >> dev) & 1)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GEOBO_E_LAUNCH;
    attr_done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  }
  const int nchunks = (g.y1 - g.y0 + OC - 1) / OC;
  if (g.nprop == 3 && nchunks >= 2) {
    // three blocks are six waves with ONE chunk each; as 2 + 1 blocks every launch has eight waves and two / four chunks per
    // staged row (measured at ny = 128: 6.5 ms against 3.7 + 1.8)
    ToeplitzWinArgs a2 = g, a1 = g;
    a2.nprop = 2;
    a1.nprop = 1; a1.tab[0] = g.tab[2]; a1.out[0] = g.out[2];
    const int rc = launch_win<NY, OC, ACC>(a2, st);
    return rc ? rc : launch_win<NY, OC, ACC>(a1, st);
  }
  ToeplitzWinArgs a = g;
  a.cw = g.nprop == 1 ? 4 : g.nprop == 2 ? 2 : 1;  // eight waves (256 VGPRs each: the table window alone is 158)
  if (g.nprop == 1 && nchunks % 4 != 0 && nchunks % 3 == 0) a.cw = 3;   // ny = 96: 6 chunks as 3 + 3 (six waves), not 4 + 2 with two waves idle
  if (a.cw > nchunks) a.cw = nchunks;
  const int ngroups = (nchunks + a.cw - 1) / a.cw;
  int64_t gz = 1;                                   // one workgroup per CU: a few waves of workgroups, each sweeping R / gz rows
  while ((g.C / 64) * ngroups * gz < 1024 && gz < g.R) ++gz;
  a.ngroups = ngroups;
  a.nbx = (int)(g.C / 64);
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.nbx * ngroups * gz)), dim3(128 * g.nprop * a.cw), lds, st, a);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int NY>
int launch(const ToeplitzArgs& g, hipStream_t st) {
  int64_t gy = g.R < 4 ? g.R : 4;  // 256 column blocks x 4 at 64^3: every workgroup sweeps R/4 rows with one table load
  while ((g.C / 64) * gy < 1024 && gy < g.R) ++gy;
  if (g.nprop == 1 && NY % 4 == 0 && NY >= 32)
    hipLaunchKernelGGL((toeplitz_y_kernel<NY, 2>), dim3((unsigned)(g.C / 64), (unsigned)gy), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((toeplitz_y_kernel<NY, 1>), dim3((unsigned)(g.C / 64), (unsigned)gy), dim3(128 * g.nprop), 0, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

}  // namespace

extern "C" int geobo_toeplitz_y(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* tab0,
                                const double* tab1, double* out0, double* out1, int y0, int y1, void* stream) {
  if (!in || !tab0 || !out0 || (nprop == 2 && (!tab1 || !out1))) return GEOBO_E_ARG;
  if (nprop < 1 || nprop > 2 || R <= 0 || y0 < 0 || y1 > ny || y1 <= y0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (plane & 1) || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (ny > 64) {
    const double* tabs[3] = {tab0, nprop == 2 ? tab1 : nullptr, nullptr};
    double* outs[3] = {out0, nprop == 2 ? out1 : nullptr, nullptr};
    return geobo_toeplitz_y3(ny, C, plane, R, nprop, in, tabs, outs, y0, y1, stream);
  }
  ToeplitzArgs g;
  g.in = in; g.tab[0] = tab0; g.tab[1] = nprop == 2 ? tab1 : tab0; g.out[0] = out0; g.out[1] = nprop == 2 ? out1 : out0;
  g.C = C; g.S = plane; g.R = R; g.nprop = nprop; g.y0 = y0; g.y1 = y1;
  switch (ny) {
    case 16: return launch<16>(g, st);
    case 32: return launch<32>(g, st);
    case 48: return launch<48>(g, st);
    case 64: return launch<64>(g, st);
    default: return GEOBO_E_UNSUPPORTED;  // other y extents: carry y through the spectrum instead (spectral.py)
  }
}

extern "C" int geobo_toeplitz_y2t(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_g0,
                                  const double* tab_g1, const double* tab_m0, const double* tab_m1, double* out0, double* out1, void* stream) {
  if (!in_g || !in_m || !tab_g0 || !tab_g1 || !tab_m0 || !tab_m1 || !out0 || !out1) return GEOBO_E_ARG;
  if (R <= 0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (plane & 1) || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  Toeplitz2Args g;
  g.in[0] = in_g; g.in[1] = in_m; g.tab[0][0] = tab_g0; g.tab[0][1] = tab_g1; g.tab[1][0] = tab_m0; g.tab[1][1] = tab_m1;
  g.out[0] = out0; g.out[1] = out1; g.C = C; g.S = plane; g.R = R;
  switch (ny) {
    case 64: return launch2<64>(g, (hipStream_t)stream);
    case 48: return launch2<48>(g, (hipStream_t)stream);
    case 32: return launch2<32>(g, (hipStream_t)stream);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

extern "C" int geobo_toeplitz_y2s(int ny, int64_t C, int64_t plane, int64_t R, const double* in_g, const double* in_m, const double* tab_d0,
                                  const double* tab_x, const double* tab_d1, double* out0, double* out1, void* stream) {
  if (!in_g || !in_m || !tab_d0 || !tab_x || !tab_d1 || !out0 || !out1) return GEOBO_E_ARG;
  if (R <= 0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (plane & 1) || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  Toeplitz2sArgs g;
  g.in[0] = in_g; g.in[1] = in_m; g.tab[0] = tab_d0; g.tab[1] = tab_x; g.tab[2] = tab_d1;
  g.out[0] = out0; g.out[1] = out1; g.C = C; g.S = plane; g.R = R;
  switch (ny) {
    case 64: return launch2s<64>(g, (hipStream_t)stream);
    case 48: return launch2s<48>(g, (hipStream_t)stream);
    case 32: return launch2s<32>(g, (hipStream_t)stream);
    default: return GEOBO_E_UNSUPPORTED;
  }
}

namespace {
template <bool ACC>
int launch_win_by_ny(int ny, const ToeplitzWinArgs& g, hipStream_t st) {
  switch (ny) {     // the windowed kernel: a lane's half of the inputs + 16 outputs = a window of ny/2 + 15 table values in registers
    case 128: return launch_win<128, 16, ACC>(g, st);
    case 112: return launch_win<112, 16, ACC>(g, st);
    case 96: return launch_win<96, 16, ACC>(g, st);
    case 80: return launch_win<80, 16, ACC>(g, st);
    default: return GEOBO_E_UNSUPPORTED;   // other y extents: carry y through the spectrum instead (spectral.py)
  }
}
}  // namespace

extern "C" int geobo_toeplitz_y3_add(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                                     double* const* outs, int y0, int y1, void* stream) {
  if (!in || !tabs || !outs || nprop < 1 || nprop > 3) return GEOBO_E_ARG;
  for (int j = 0; j < nprop; ++j)
    if (!tabs[j] || !outs[j]) return GEOBO_E_ARG;
  if (R <= 0 || y0 < 0 || y1 > ny || y1 <= y0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (plane & 1) || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  if (ny <= 64) return GEOBO_E_UNSUPPORTED;   // (the register-table kernel has the two-term form geobo_toeplitz_y2t instead)
  ToeplitzWinArgs g;
  g.in = in; g.C = C; g.S = plane; g.R = R; g.nprop = nprop; g.y0 = y0; g.y1 = y1;
  for (int j = 0; j < 3; ++j) { g.tab[j] = tabs[j < nprop ? j : 0]; g.out[j] = outs[j < nprop ? j : 0]; }
  return launch_win_by_ny<true>(ny, g, (hipStream_t)stream);
}

extern "C" int geobo_toeplitz_y3(int ny, int64_t C, int64_t plane, int64_t R, int nprop, const double* in, const double* const* tabs,
                                 double* const* outs, int y0, int y1, void* stream) {
  if (!in || !tabs || !outs || nprop < 1 || nprop > 3) return GEOBO_E_ARG;
  for (int j = 0; j < nprop; ++j)
    if (!tabs[j] || !outs[j]) return GEOBO_E_ARG;
  if (R <= 0 || y0 < 0 || y1 > ny || y1 <= y0 || plane < C) return GEOBO_E_ARG;
  if (C <= 0 || C % 64 || (plane & 1) || (int64_t)ny * plane * 8 >= (1ll << 31)) return GEOBO_E_ALIGN;
  if (ny <= 64) {
    // shorter axes: the register-table kernel, two property blocks per sweep
    for (int j = 0; j < nprop; j += 2) {
      const int n = nprop - j >= 2 ? 2 : 1;
      const int rc = geobo_toeplitz_y(ny, C, plane, R, n, in, tabs[j], n == 2 ? tabs[j + 1] : nullptr, outs[j], n == 2 ? outs[j + 1] : nullptr,
                                      y0, y1, stream);
      if (rc) return rc;
    }
    return GEOBO_OK;
  }
  ToeplitzWinArgs g;
  g.in = in; g.C = C; g.S = plane; g.R = R; g.nprop = nprop; g.y0 = y0; g.y1 = y1;
  for (int j = 0; j < 3; ++j) { g.tab[j] = tabs[j < nprop ? j : 0]; g.out[j] = outs[j < nprop ? j : 0]; }
  return launch_win_by_ny<false>(ny, g, (hipStream_t)stream);
}
