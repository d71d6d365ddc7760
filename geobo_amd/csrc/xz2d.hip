// xz2d.hip -- fused two-axis real-DFT passes of the structured covariance product (DESIGN.md section 2;
// replaces the dense K_sj blocks of kernels.py:158-195 on the grid of kernels.py:27-42).
//
// Every (sensor row, y) plane of the forward operator goes   X (nx x nz)  ->  Gx X Gz^T (2nx x 2nz)   into the (x, z)
// spectrum, and comes back   S (2nx x 2nz)  ->  Gx^T S Gz (nx x nz)   after the y stage (toeplitz.hip).  As two separate
// batched GEMMs each pass stores its intermediate to HBM and reloads it, and every small tile pays load -> compute ->
// store in sequence (measured: 155 ms per 64^3 step, half HBM time and half MFMA time, not overlapped).  Here one
// persistent workgroup carries a plane through BOTH contractions with the intermediate in registers:
//
//   step 1   T[ix][oz] = sum_iz In[ix][iz] Mz[oz][iz]     wave w owns 16*CT output columns oz and ALL rows ix;
//   step 2   O[ox][oz] = sum_ix Mx[ox][ix] T[ix][oz]      the D fragments of step 1 ARE the B fragments of step 2:
//            v_mfma_f64_16x16x4 returns D[(lane>>4) + 4 reg][lane & 15] and reads B[k = lane>>4][j = lane & 15], so
//            register `reg` of row tile rt is k-step (rt, reg) with k <-> ix = 16 rt + 4 reg + (lane>>4)  -- no LDS
//            round trip, no shuffle; the rows of Mx are stored in LDS in that k order.
//
// In streams through a ring of 16-row chunks filled by LDS-DMA three chunks ahead (16-byte slots XOR-swizzled by row & 15 to suit ds_read_b128's lane groups, b128
// fragment reads, two k values per read); Mz fragments live in registers, Mx (64 KiB) in LDS, both loaded once per
// workgroup.  Per plane and wave: 384 MFMAs (64^3) against 64 + 128 LDS reads; HBM traffic is the plane in and the plane
// out, overlapped with the matrix pipe.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "geobo_hip.h"

namespace {

// opt-in to > 64 KiB dynamic LDS, once per DEVICE (not per process: a second device needs its own call); idempotent
int ensure_lds_attr(std::atomic<uint64_t>& done, const void* kern, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if ((done.load(std::memory_order_acquire) >> dev) & 1) return GEOBO_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GEOBO_E_LAUNCH;
  done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  return GEOBO_OK;
}

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct XZArgs {
  const double* in; int64_t in_row, in_plane;    // plane (r, p) at in + r*in_row + p*in_plane, IN_X x IN_Z row-major
  double* out; int64_t out_row, out_plane;       // plane (r, p) at out + r*out_row + p*out_plane, OUT_X x OUT_Z
  const double* Mz; int64_t ldmz;                // OUT_Z x IN_Z
  const double* Mx; int64_t ldmx;                // OUT_X x IN_X
  int ppr;                                       // planes per row
  int64_t nplanes;
};

constexpr int RING = 4;  // chunks in the LDS ring (prefetch distance RING-1)

// s_waitcnt immediate for vmcnt(v) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14)
constexpr int vmcnt_imm(int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); }

constexpr int xz_threads(int out_z) { return out_z / 16 >= 8 ? 512 : (out_z / 16 >= 4 ? 256 : 128); }

template <int IN_X, int IN_Z, int OUT_X, int OUT_Z>
struct XZCfg {
  static constexpr int NW = xz_threads(OUT_Z) / 64;    // waves per workgroup: two per SIMD when there are 8 column tiles
  static constexpr int CT = OUT_Z / 16 / NW;          // column tiles per wave
  static constexpr int RT1 = IN_X / 16;               // row tiles of T = chunks per plane
  static constexpr int KP = IN_Z / 8;                 // k-step pairs of step 1 (one b128 read each)
  static constexpr int RT2 = OUT_X / 16;
  static constexpr int ROWB = IN_Z * 8;               // bytes per input row
  static constexpr int CHB = 16 * ROWB;               // bytes per chunk
  static constexpr int ND = CHB / 1024 / NW;          // DMA instructions per wave per chunk
  static constexpr int LPR = ROWB / 16;               // lanes (16-byte slots) per row
  static constexpr int RPI = 64 / LPR;                // rows per DMA instruction
  static constexpr int MXS = (IN_X + 31) / 32 * 32;   // row stride of Mx in LDS (doubles): whole blocks of 16 XOR-swizzled 16-byte slots
  static constexpr int NS = RT2 * CT * 4;             // stores per wave per plane
  static constexpr size_t LDS = (size_t)RING * CHB + (size_t)OUT_X * MXS * 8;
  static_assert(CT >= 1 && OUT_Z % (16 * NW) == 0 && CHB % (1024 * NW) == 0 && IN_X % 16 == 0 && IN_Z % 32 == 0 && IN_Z <= 128 && OUT_X % 16 == 0, "shape");
  static_assert(ND >= 1 && RT1 >= RING - 1, "chunking");
};

// step-1 MFMAs of one chunk; fragment t is consumed once the LDS queue has drained down to the KP-1-t reads behind it.
// NA partial accumulators per tile (summed by the caller) so that at least four independent MFMA chains interleave:
// dependent MFMAs on one accumulator do not issue back to back.
template <int KP, int CT, int NA, int T>
__device__ __forceinline__ void step1(v2d (&a)[KP], const double (&gz)[CT][2 * KP], v4d (&d)[NA][CT]) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[T]) : "n"(KP - 1 - T));
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    d[(2 * T) % NA][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].x, gz[ct][2 * T], d[(2 * T) % NA][ct], 0, 0, 0);
    d[(2 * T + 1) % NA][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T].y, gz[ct][2 * T + 1], d[(2 * T + 1) % NA][ct], 0, 0, 0);
  }
  if constexpr (T + 1 < KP) step1<KP, CT, NA, T + 1>(a, gz, d);
}

template <int IN_X, int IN_Z, int OUT_X, int OUT_Z>
__global__ void __launch_bounds__(xz_threads(OUT_Z), 1) xz2d_kernel(XZArgs g) {
  using K = XZCfg<IN_X, IN_Z, OUT_X, OUT_Z>;
  constexpr int CT = K::CT, RT1 = K::RT1, KP = K::KP, RT2 = K::RT2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* const ring = reinterpret_cast<char*>(smem);
  double* const mx = smem + RING * K::CHB / 8;
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring;   // LDS byte address of the ring
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, q = lane >> 4;

  // ---- constants: Mx into LDS in the chained k order, Mz fragments into registers ------------------------------------
  for (int idx = tid; idx < OUT_X * IN_X; idx += 64 * K::NW) {
    const int ox = idx / IN_X, ix = idx % IN_X;
    const int rem = ix & 15;                                  // ix = 16 rt + 4 reg + q  ->  position 16 rt + 4 q + reg
    const int pos = (ix & ~15) + 4 * (rem & 3) + (rem >> 2);
    mx[ox * K::MXS + ((((pos >> 1) ^ (ox & 15)) << 1) | (pos & 1))] = g.Mx[(int64_t)ox * g.ldmx + ix];
  }
  double gz[CT][2 * KP];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < KP; ++t) {
      const double* p = g.Mz + (int64_t)(16 * (w * CT + ct) + lr) * g.ldmz + 8 * t + 2 * q;
      gz[ct][2 * t] = p[0];
      gz[ct][2 * t + 1] = p[1];
    }

  // ---- chunk stream: chunk c of plane n -> ring slot (n * RT1 + c) % RING ----------------------------------------------
  const int64_t first = blockIdx.x, pstep = gridDim.x;
  if (first >= g.nplanes) return;
  const int drow = lane / K::LPR, dpos = lane % K::LPR;       // this lane's row / 16-byte slot within a DMA instruction
  auto plane_ptr = [&](int64_t p) { return g.in + (p / g.ppr) * g.in_row + (p % g.ppr) * g.in_plane; };
  auto stage = [&](const double* plane, int c, int slot) {
#pragma unroll
    for (int j = 0; j < K::ND; ++j) {
      const int ii = w + K::NW * j;                           // DMA instruction index within the chunk (1 KiB each)
      const int row = ii * K::RPI + drow;                     // row within the chunk
      const char* src = reinterpret_cast<const char*>(plane) + (int64_t)(16 * c + row) * K::ROWB + ((dpos ^ (row & 15)) << 4);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(ring + slot * K::CHB + ii * 1024), 16, 0, 0);
    }
  };
  // prologue: chunks 0 .. RING-2 of the stream (they all belong to the first plane: RT1 >= RING-1)
  const double* cur = plane_ptr(first);
  __syncthreads();                                            // Mx visible
#pragma unroll
  for (int c = 0; c < RING - 1; ++c) stage(cur, c, c);
  int slot0 = 0;                                              // ring slot of chunk 0 of the current plane
  bool warm = false;                                          // false until a plane's stores have been issued
  for (int64_t p = first; p < g.nplanes; p += pstep) {
    const int64_t pn = p + pstep < g.nplanes ? p + pstep : p; // past the end: re-stage the last plane (keeps the counts static)
    const double* nxt = plane_ptr(pn);
    v4d d1[RT1][CT];
#pragma unroll
    for (int c = 0; c < RT1; ++c) {
      // (1) this wave's share of chunk c has landed.  Chunks 0 .. RING-2 of a plane were requested during the previous plane
      //     and waited for there (the vmcnt(0) in front of its stores); for the others -- and in the first plane -- everything
      //     issued after chunk c is RING-2 newer chunks.  The waits never have to see past stores: vmcnt counts loads and
      //     stores together, and nothing here relies on the two completing in issue order relative to each other.
      if (!(c <= RING - 2 && warm)) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * K::ND));
      // (2) every share landed; every wave is done with the chunk staged RING-1 ago.  A bare s_barrier: __syncthreads()
      //     adds a fence that drains vmcnt to 0, i.e. waits for the prefetched chunks and the output stores as well.
      __builtin_amdgcn_s_barrier();
      {                 // (3) refill the slot that was just released
        const int cn = c + RING - 1;
        if (cn < RT1) stage(cur, cn, (slot0 + cn) % RING);
        else stage(nxt, cn - RT1, (slot0 + cn) % RING);
      }
      // (4) step 1 on row tile c.  The fragment reads are inline asm: for LDS reads it can see, the compiler waits for
      //     EVERY outstanding LDS-DMA first (vmcnt(0): it cannot tell ring slots apart), which would collapse the
      //     three-chunk prefetch to one; the waits that matter are (1) and the lgkmcnt below.
      const unsigned xs = ring_lds + ((slot0 + c) % RING) * K::CHB + lr * K::ROWB;
      constexpr int NA = CT >= 2 ? 2 : 4;
      v4d dd[NA][CT];
#pragma unroll
      for (int e = 0; e < NA; ++e)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dd[e][ct] = (v4d){0., 0., 0., 0.};
      v2d a[KP];
#pragma unroll
      for (int t = 0; t < KP; ++t) {
        const unsigned addr = xs + (((4 * t + q) ^ lr) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[t]) : "v"(addr));
      }
      step1<KP, CT, NA, 0>(a, gz, dd);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if constexpr (NA == 4) d1[c][ct] = (dd[0][ct] + dd[1][ct]) + (dd[2][ct] + dd[3][ct]);
        else d1[c][ct] = dd[0][ct] + dd[1][ct];
      }
    }
    // ---- step 2 + stores, half the output row tiles at a time (register budget) ------------------------------------------
    double* const op = g.out + (p / g.ppr) * g.out_row + (p % g.ppr) * g.out_plane + 16 * (w * CT) + lr;
    constexpr int HALF = (RT2 * CT > 8 && RT2 % 2 == 0) ? RT2 / 2 : RT2;  // <= 8 accumulator tiles at a time
#pragma unroll
    for (int h0 = 0; h0 < RT2; h0 += HALF) {
      v4d acc[HALF][CT];
#pragma unroll
      for (int m = 0; m < HALF; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[m][ct] = (v4d){0., 0., 0., 0.};
#pragma unroll
      for (int rt = 0; rt < RT1; ++rt) {
        v2d a01[HALF], a23[HALF];
#pragma unroll
        for (int m = 0; m < HALF; ++m) {
          // 16-byte slot 8 rt + 2 q (+1), XOR row & 15: the 16 lanes of every ds_read_b128 lane group hit 16 different slots
          const double* ap = mx + (16 * (h0 + m) + lr) * K::MXS;
          a01[m] = *reinterpret_cast<const v2d*>(ap + (((8 * rt + 2 * q) ^ lr) << 1));
          a23[m] = *reinterpret_cast<const v2d*>(ap + (((8 * rt + 2 * q + 1) ^ lr) << 1));
        }
        // k-step outermost: consecutive MFMAs go to different accumulators (HALF*CT independent chains)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int m = 0; m < HALF; ++m)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(kk == 0 ? a01[m].x : kk == 1 ? a01[m].y : kk == 2 ? a23[m].x : a23[m].y,
                                                                d1[rt][ct][kk], acc[m][ct], 0, 0, 0);
      }
      // the next plane's first RING-1 chunks were requested at least one step 2 ago: drain them here (free), so that no
      // later wait has this plane's stores between itself and the chunk it waits for
      if (h0 == 0) __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
#pragma unroll
      for (int m = 0; m < HALF; ++m)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            op[(int64_t)(16 * (h0 + m) + q + 4 * r) * OUT_Z + 16 * ct] = acc[m][ct][r];
    }
    warm = true;
    slot0 = (slot0 + RT1) % RING;
    cur = nxt;
  }
}

template <int IN_X, int IN_Z, int OUT_X, int OUT_Z>
int launch(const XZArgs& g, hipStream_t st) {
  using K = XZCfg<IN_X, IN_Z, OUT_X, OUT_Z>;
  auto kern = xz2d_kernel<IN_X, IN_Z, OUT_X, OUT_Z>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), K::LDS)) return rc;
  int64_t nwg = g.nplanes < 1024 ? g.nplanes : 1024;  // persistent: 4 workgroups per CU over the launch, >= 8 planes each at 64^3
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * K::NW), K::LDS, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

// ---- lattice Gram, x step (geobo_xcorr_reduce) ----------------------------------------------------------------------------
// For every (row r, y-mode p) plane X (nx x nz):   out[r][p][o] = sum_z lamT[p][z][o] * sum_x Gx[o][x] X[x][z],  o < 2nx.
// The product is formed TRANSPOSED, D[z][o] = sum_x X^T[z][x] Gx^T[x][o], so that the sum over z runs over accumulator
// registers and the four 16-lane groups of a wave (two cross-lane steps per value) instead of over the 16 lanes of a group
// (four steps for each of 32 values: that version spent 80 % of its time in the reduction).  Wave w owns z rows 16w..16w+15
// and all 2nx columns; X streams through the same LDS-DMA chunk ring as xz2d (16 x rows = four k-steps per chunk), Gx^T
// (64 KiB, padded rows) sits in LDS, the eigenvalue planes lamT (z-major, 64 KiB per y-mode) come from L2: planes are
// numbered p-major so that the workgroups of a round share a handful of them.
struct XCArgs {
  const double* in; int64_t in_row, in_plane;   // plane (r, p) at in + r*in_row + p*in_plane
  const double* Mx; int64_t ldmx;               // Gx, PX x NX
  const double* lamT;                           // [planes][NZ][PX]
  double* out; int64_t out_row, out_plane;      // out + r*out_row + p*out_plane + o
  int64_t rows, nplanes;
};

// slot swizzle of the xcorr input chunks: a ds_read_b64 serves lanes 0-31 together = two consecutive x rows (q = 0, 1) of 16
// z columns each; bit 0 of the row goes to bit 3 of the slot XOR, so the two rows land in different 128-byte halves of the
// 256-byte bank row
__device__ __forceinline__ int xswz(int row) { return ((row & 1) << 3) | ((row >> 1) & 7); }

template <int NX, int NZ, int PX, int NW>
__global__ void __launch_bounds__(64 * NW, 1) xcorr_kernel(XCArgs g) {
  static_assert(NZ == 64 && NX % 16 == 0 && PX % 32 == 0 && (NW == 4 || NW == 8), "one 16-row z tile per wave (pair)");
  // NW = 8: two waves per SIMD; waves w and w + 4 share a z tile and split the 2nx output columns
  constexpr int CT = PX / 16 / (NW / 4), NCH = NX / 16, ROWB = NZ * 8, CHB = 16 * ROWB, LPR = ROWB / 16, RPI = 64 / LPR;
  constexpr int ND = CHB / 1024 / NW, GS = PX + 16;
  static_assert(ND >= 1 && NCH >= RING - 1, "chunking");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* const ring = reinterpret_cast<char*>(smem);
  double* const gxt = smem + RING * CHB / 8;           // [NX][GS]: Gx^T, row stride 2nx + 16 (conflict-free b64 fragment reads)
  double* const red = gxt + NX * GS;                   // [2][4][PX]
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wv & 3, ct0 = (wv >> 2) * CT;          // z tile, first output column tile of this wave
  const int lr = lane & 15, q = lane >> 4;
  for (int idx = tid; idx < PX * NX; idx += 64 * NW) {
    const int o = idx / NX, x = idx % NX;
    gxt[x * GS + o] = g.Mx[(int64_t)o * g.ldmx + x];
  }
  const int64_t first = blockIdx.x, pstep = gridDim.x;
  if (first >= g.nplanes) return;
  const int drow = lane / LPR, dpos = lane % LPR;
  auto plane_ptr = [&](int64_t p) { return g.in + (p % g.rows) * g.in_row + (p / g.rows) * g.in_plane; };
  auto stage = [&](const double* plane, int c, int slot) {
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int ii = wv + NW * j;
      const int row = ii * RPI + drow;
      const char* src = reinterpret_cast<const char*>(plane) + (int64_t)(16 * c + row) * ROWB + ((dpos ^ xswz(row)) << 4);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(ring + slot * CHB + ii * 1024), 16, 0, 0);
    }
  };
  const double* cur = plane_ptr(first);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < RING - 1; ++c) stage(cur, c, c);
  int slot0 = 0, it = 0;
  const int col = 16 * w + lr;                         // this lane's z (row i of the transposed product)
  for (int64_t p = first; p < g.nplanes; p += pstep, ++it) {
    const int64_t pn = p + pstep < g.nplanes ? p + pstep : p;
    const double* nxt = plane_ptr(pn);
    v4d acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (v4d){0., 0., 0., 0.};
    // this plane's eigenvalues: issued now, consumed after the MFMAs (their L2 latency hides under the chunk loop)
    const int64_t pl = p / g.rows;
    const double* lp = g.lamT + pl * (int64_t)(NZ * PX) + (int64_t)(16 * w + q) * PX + 16 * ct0 + lr;
    double lam[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) lam[ct][r] = lp[4 * r * PX + 16 * ct];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk c has landed when at most the newer operations are in flight: RING-2 chunks, plus -- for the chunks that were
      // requested during the previous plane -- the 4 CT eigenvalue loads above (conservative for the <= 1 result store)
      if (c <= RING - 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * ND + 4 * CT > 63 ? 63 : (RING - 2) * ND + 4 * CT));
      else __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * ND));
      __builtin_amdgcn_s_barrier();
      {
        const int cn = c + RING - 1;
        if (cn < NCH) stage(cur, cn, (slot0 + cn) % RING);
        else stage(nxt, cn - NCH, (slot0 + cn) % RING);
      }
      const unsigned xs = ring_lds + ((slot0 + c) % RING) * CHB;
      double a[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {                    // A[i = z][k = x]: X[x = 16c + 4s + q][z = col]
        const int row = 4 * s + q;
        const unsigned addr = xs + row * ROWB + ((((col >> 1) ^ xswz(row)) << 4) | ((col & 1) << 3));
        asm volatile("ds_read_b64 %0, %1" : "=v"(a[s]) : "v"(addr));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double* bp = gxt + (16 * c + 4 * s + q) * GS + 16 * ct0 + lr;   // B[k = x][j = o]
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bp[16 * ct], acc[ct], 0, 0, 0);
      }
    }
    // ---- scale by the eigenvalues and sum over z: registers (4 z per lane), then the four 16-lane groups, then the waves --
    double* const rp = red + (it & 1) * (4 * PX) + w * PX + 16 * ct0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      double v = acc[ct][0] * lam[ct][0];
      v = __builtin_fma(acc[ct][1], lam[ct][1], v);
      v = __builtin_fma(acc[ct][2], lam[ct][2], v);
      v = __builtin_fma(acc[ct][3], lam[ct][3], v);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q == 0) rp[16 * ct + lr] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < PX) {
      const double* r4 = red + (it & 1) * (4 * PX) + tid;
      g.out[(p % g.rows) * g.out_row + pl * g.out_plane + tid] = (r4[0] + r4[PX]) + (r4[2 * PX] + r4[3 * PX]);
    }
    slot0 = (slot0 + NCH) % RING;
    cur = nxt;
  }
}

// ---- lattice Gram, y step (geobo_ymul) ----------------------------------------------------------------------------------------
// out[r][ky][c] = sum_y G[ky][y] in[r][y][c]   for every row r and column c < C: the same 128 x 64 matrix from the left of every
// (64 x C) row.  As a batched GEMM (geobo_gemm_batched) every 128 x 128 tile restaged G through LDS and paid a prologue and an
// epilogue for four 16-deep chunks (0.45 ms per 256 rows at 64^3, 3.4 TB/s).  Here G lives in registers as MFMA A fragments
// (wave w owns the 32 output rows 32w .. 32w+31), a persistent workgroup streams (row, 64-column block) tiles of the input
// through a two-stage LDS ring by LDS-DMA, and the only LDS traffic is the B fragments.
struct YMulArgs {
  const double* in; int64_t in_row;     // row r at in + r*in_row, [64][C]
  const double* G; int64_t ldg;         // 128 x 64
  double* out; int64_t out_row;         // row r at out + r*out_row, [128][C]
  int64_t C, R, ntiles;                 // ntiles = R * (C / 64)
};

// FOLD (round 5, geobo_ymul_fold): G is the pair-interleaved basis (row 2b+1 = (-1)^j row 2b, e.g. spectral.forward_matrix with columns
// zeroed): per pair ONE even-input and ONE odd-input sum, out[2b] = E + O, out[2b+1] = E - O -- 64 MFMAs per tile instead of 128 (the
// kernel was on both roofs: 4.2 TB/s of HBM and 65 % MFMA busy); wave w owns the base rows 16 w .. 16 w + 15 = output rows 32 w .. 32 w + 31.
template <bool FOLD>
__global__ void __launch_bounds__(256, 2) ymul_kernel(YMulArgs g) {
  constexpr int NY = 64, CB = 64, TILE = NY * CB;          // doubles per staged tile (32 KiB)
  __shared__ __attribute__((aligned(16))) double xs[2][TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, q = lane >> 4;
  // A fragments: A[i = lane & 15][k = lane >> 4] of M tile mt, k-step t  ->  G[32 w + 16 mt + lr][4 t + q]
  // FOLD: a[0][t] = Fe[b = 16 w + lr][j = 4 t + q] = G[2 b][2 j], a[1][t] = Fo[b][j] = G[2 b][2 j + 1], t < 8
  double a[2][16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if constexpr (FOLD) a[mt][t] = t < 8 ? g.G[(int64_t)(2 * (16 * w + lr)) * g.ldg + 2 * (4 * t + q) + mt] : 0.0;
      else a[mt][t] = g.G[(int64_t)(32 * w + 16 * mt + lr) * g.ldg + 4 * t + q];
    }
  const int64_t cbs = g.C / CB;
  int64_t tile = blockIdx.x;
  if (tile >= g.ntiles) return;
  // one DMA instruction = two y rows of the tile (lanes 0-31 row 2i, lanes 32-63 row 2i+1; 16 bytes per lane).  The four
  // 128-byte column groups of row y are stored at group ^ (y & 3) (the LDS image of a DMA is lane-linear, so the permutation is
  // applied to the SOURCE slot): the four rows 4t .. 4t+3 a B-fragment read touches then sit in four different bank groups
  // (unswizzled: SQ_LDS_BANK_CONFLICT = 50 % of the LDS-active cycles).  (2 pr + h) & 3 = (2 w + h) & 3 for every pr = w + 4 i.
  const int dma_h = lane >> 5;
  const int64_t dma_lane = (int64_t)dma_h * g.C + ((lane & 31) ^ (((2 * w + dma_h) & 3) << 3)) * 2;
  auto stage = [&](int64_t t_, int b) {
    const double* src = g.in + (t_ / cbs) * g.in_row + (t_ % cbs) * CB + dma_lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pr = w + 4 * i;                              // row pair
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (int64_t)(2 * pr) * g.C), (lds_ptr_t)(&xs[b][2 * pr * CB]), 16, 0, 0);
    }
  };
  stage(tile, 0);
  int b = 0;
  // The stores of a tile are issued one iteration LATE, right after the request for the tile after next: both are in flight
  // under a whole tile of MFMAs, so the vmcnt(0) in front of the barrier (loads and stores share the counter and do not retire
  // in order relative to each other: it has to be 0) never waits for a store that was issued a moment ago.
  v4d prev[2][4];
  int64_t prev_tile = -1;
  auto store = [&](const v4d (&acc)[2][4], int64_t t_) {
    double* op = g.out + (t_ / cbs) * g.out_row + (t_ % cbs) * CB + lr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (FOLD)     // acc[0] = E, acc[1] = O of base row 16 w + q + 4 r: rows 2 b (mt = 0) and 2 b + 1 (mt = 1)
            op[(int64_t)(2 * (16 * w + q + 4 * r) + mt) * g.C + 16 * nt] = mt ? acc[0][nt][r] - acc[1][nt][r] : acc[0][nt][r] + acc[1][nt][r];
          else
            op[(int64_t)(32 * w + 16 * mt + q + 4 * r) * g.C + 16 * nt] = acc[mt][nt][r];
        }
  };
  for (; tile < g.ntiles; tile += gridDim.x) {
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));              // this tile has landed
    __builtin_amdgcn_s_barrier();                          // ... for every wave; every wave is done reading the other stage
    const int64_t nxt = tile + gridDim.x;
    if (nxt < g.ntiles) stage(nxt, b ^ 1);
    if (prev_tile >= 0) store(prev, prev_tile);
    v4d acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (v4d){0., 0., 0., 0.};
    // B[k = q][n = lr] of k-step t, column tile nt: xs[b][(4 t + q) * CB + 16 (nt ^ q) + lr].  Inline reads: for LDS reads it can see,
    // the compiler first waits for EVERY outstanding LDS-DMA (vmcnt(0)), i.e. for the tile that was requested a moment ago.
    if constexpr (FOLD) {
      // k-step t: the even rows y = 8 t + 2 q feed E, the odd rows y + 1 feed O (row y is stored with its column groups at nt ^ (y & 3))
      unsigned xe[4], xo[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        xe[nt] = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][(2 * q) * CB + 16 * (nt ^ ((2 * q) & 3)) + lr];
        xo[nt] = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][(2 * q + 1) * CB + 16 * (nt ^ ((2 * q + 1) & 3)) + lr];
      }
      double be[2][4], bo[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        asm volatile("ds_read_b64 %0, %1" : "=v"(be[0][nt]) : "v"(xe[nt]));
        asm volatile("ds_read_b64 %0, %1" : "=v"(bo[0][nt]) : "v"(xo[nt]));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(be[(t + 1) & 1][nt]) : "v"(xe[nt]), "n"(8 * (t + 1) * CB * 8));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(bo[(t + 1) & 1][nt]) : "v"(xo[nt]), "n"(8 * (t + 1) * CB * 8));
          }
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(be[t & 1][0]), "+v"(be[t & 1][1]), "+v"(be[t & 1][2]), "+v"(be[t & 1][3]),
                       "+v"(bo[t & 1][0]), "+v"(bo[t & 1][1]), "+v"(bo[t & 1][2]), "+v"(bo[t & 1][3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(be[t & 1][0]), "+v"(be[t & 1][1]), "+v"(be[t & 1][2]), "+v"(be[t & 1][3]),
                       "+v"(bo[t & 1][0]), "+v"(bo[t & 1][1]), "+v"(bo[t & 1][2]), "+v"(bo[t & 1][3]));
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          acc[0][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][t], be[t & 1][nt], acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][t], bo[t & 1][nt], acc[1][nt], 0, 0, 0);
        }
      }
    } else {
    unsigned xaddr[4];                                     // row q of a k-step, column group nt (stored at group nt ^ q)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) xaddr[nt] = (unsigned)(uintptr_t)(lds_ptr_t)&xs[b][q * CB + 16 * (nt ^ q) + lr];
    double bv[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) asm volatile("ds_read_b64 %0, %1" : "=v"(bv[0][nt]) : "v"(xaddr[nt]));
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + 1 < 16) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(bv[(t + 1) & 1][nt]) : "v"(xaddr[nt]), "n"(4 * (t + 1) * CB * 8));
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bv[t & 1][0]), "+v"(bv[t & 1][1]), "+v"(bv[t & 1][2]), "+v"(bv[t & 1][3]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[t & 1][0]), "+v"(bv[t & 1][1]), "+v"(bv[t & 1][2]), "+v"(bv[t & 1][3]));
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][t], bv[t & 1][nt], acc[mt][nt], 0, 0, 0);
    }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) prev[mt][nt] = acc[mt][nt];
    prev_tile = tile;
    b ^= 1;
  }
  if (prev_tile >= 0) store(prev, prev_tile);
}

}  // namespace

extern "C" int geobo_xz2d(int inverse, int nx, int nz, int64_t rows, int planes_per_row, const double* in, int64_t in_row,
                          int64_t in_plane, const double* Mx, int64_t ldmx, const double* Mz, int64_t ldmz, double* out,
                          int64_t out_row, int64_t out_plane, void* stream) {
  if (!in || !out || !Mx || !Mz) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if ((in_row & 1) || (in_plane & 1) || ((uintptr_t)in & 15)) return GEOBO_E_ALIGN;
  XZArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.out = out; g.out_row = out_row; g.out_plane = out_plane;
  g.Mz = Mz; g.ldmz = ldmz; g.Mx = Mx; g.ldmx = ldmx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row;
  hipStream_t st = (hipStream_t)stream;
  if (nz == 32 && nx == 64) {
    // also the form 32 x 32 planes take: two consecutive planes stacked along x with Mx = diag(Mx32, Mx32) (spectral.py)
    return inverse ? launch<128, 64, 64, 32>(g, st) : launch<64, 32, 128, 64>(g, st);
  }
  if (nz != 64) return GEOBO_E_UNSUPPORTED;
  if (!inverse) {
    if (nx == 64) return launch<64, 64, 128, 128>(g, st);
    if (nx == 48) return launch<48, 64, 96, 128>(g, st);
  } else {
    if (nx == 64) return launch<128, 128, 64, 64>(g, st);
    if (nx == 48) return launch<96, 128, 48, 64>(g, st);
  }
  return GEOBO_E_UNSUPPORTED;
}

extern "C" int geobo_xcorr_reduce(int nx, int nz, int64_t rows, int planes, const double* in, int64_t in_row, int64_t in_plane,
                                  const double* Mx, int64_t ldmx, const double* lamT, double* out, int64_t out_row,
                                  int64_t out_plane, void* stream) {
  if (!in || !out || !Mx || !lamT) return GEOBO_E_ARG;
  if (rows <= 0 || planes <= 0) return GEOBO_OK;
  if ((in_row & 1) || (in_plane & 1) || ((uintptr_t)in & 15)) return GEOBO_E_ALIGN;
  if (nx != 64 || nz != 64) return GEOBO_E_UNSUPPORTED;
  XCArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.Mx = Mx; g.ldmx = ldmx; g.lamT = lamT;
  g.out = out; g.out_row = out_row; g.out_plane = out_plane; g.rows = rows; g.nplanes = rows * planes;
  constexpr int NX = 64, NZ = 64, PX = 128;
  constexpr size_t lds = (size_t)RING * 16 * NZ * 8 + (size_t)NX * (PX + 16) * 8 + 2 * 4 * PX * 8;
  constexpr int NW = 8;
  auto kern = xcorr_kernel<NX, NZ, PX, NW>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds)) return rc;
  const int64_t nwg = g.nplanes < 1024 ? g.nplanes : 1024;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW), lds, (hipStream_t)stream, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_ymul(int m, int k, int64_t C, int64_t rows, const double* G, int64_t ldg, const double* in, int64_t in_row,
                          double* out, int64_t out_row, void* stream) {
  if (!G || !in || !out) return GEOBO_E_ARG;
  if (rows <= 0 || C <= 0) return GEOBO_OK;
  if (m != 128 || k != 64) return GEOBO_E_UNSUPPORTED;
  if (C % 64 || (in_row & 1) || ((uintptr_t)in & 15) || (C & 1)) return GEOBO_E_ALIGN;
  YMulArgs g;
  g.in = in; g.in_row = in_row; g.G = G; g.ldg = ldg; g.out = out; g.out_row = out_row; g.C = C; g.R = rows;
  g.ntiles = rows * (C / 64);
  const int64_t nwg = g.ntiles < 512 ? g.ntiles : 512;   // persistent: two workgroups per CU, 32 tiles each at 64^3
  hipLaunchKernelGGL(ymul_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_ymul_fold(int m, int k, int64_t C, int64_t rows, const double* G, int64_t ldg, const double* in, int64_t in_row,
                               double* out, int64_t out_row, void* stream) {
  if (!G || !in || !out) return GEOBO_E_ARG;
  if (rows <= 0 || C <= 0) return GEOBO_OK;
  if (m != 128 || k != 64) return GEOBO_E_UNSUPPORTED;
  if (C % 64 || (in_row & 1) || ((uintptr_t)in & 15) || (C & 1)) return GEOBO_E_ALIGN;
  YMulArgs g;
  g.in = in; g.in_row = in_row; g.G = G; g.ldg = ldg; g.out = out; g.out_row = out_row; g.C = C; g.R = rows;
  g.ntiles = rows * (C / 64);
  const int64_t nwg = g.ntiles < 512 ? g.ntiles : 512;
  hipLaunchKernelGGL(ymul_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}
