// xz2d_fold.hip -- radix-2 ("folded") form of the fused two-axis real-DFT passes of xz2d.hip: half the MFMAs.
//
// The spectral basis of geobo_amd/spectral.py is stored pair-interleaved: position 2b holds base row g_b, position 2b+1 its
// mirror (-1)^i g_b (the eigenvector of the mirrored frequency).  The two rows of a pair differ only in the sign of the odd
// inputs, so per pair ONE even-input and ONE odd-input partial sum is enough:
//
//   forward  (n -> 2n per axis):  E_b = sum_j Fe[b][j] x[2j],  O_b = sum_j Fo[b][j] x[2j+1];   out[2b] = E_b + O_b,  out[2b+1] = E_b - O_b
//   inverse  (2n -> n per axis):  u_b = s[2b] + s[2b+1],  v_b = s[2b] - s[2b+1];   x[2j] = sum_b Fe[b][j] u_b,   x[2j+1] = sum_b Fo[b][j] v_b
//
// with Fe[b][j] = g_b[2j], Fo[b][j] = g_b[2j+1] (n x n/2 each; passed interleaved as F[b][j][2]).  Both axes of a plane are
// folded: 768 v_mfma_f64_16x16x4 per 64 x 64 <-> 128 x 128 plane instead of 1536, plus ~200 fp64 VALU adds per wave for the
// butterflies.  Everything else follows xz2d.hip: one persistent workgroup carries a plane through both contractions with
// the intermediate in registers (the D fragments of step 1 are the B fragments of step 2), the input streams through an
// LDS-DMA ring of 16-row chunks with hand-counted vmcnt waits, fragment reads are inline ds_read_b128.
//
// What makes the butterflies lane-local:
//   * an MFMA k-step takes its four k values from the four 16-lane groups, and a ds_read_b128 delivers two ADJACENT inputs:
//     the .x MFMA of a read contracts even inputs only, the .y MFMA odd inputs only -> E and O are separate accumulators of
//     the same tile (forward), and a pair (s[2b], s[2b+1]) arrives in one read (inverse);
//   * the rows of a chunk are permuted on their way into LDS (the DMA source row is free to choose) so that D register `reg`
//     of a lane holds row 8 (reg >> 1) + 2 q + (reg & 1) of the chunk: even registers carry even rows, odd registers odd rows,
//     and partner rows (2b, 2b+1) sit in registers (2h, 2h+1) of the same lane.  Step 2's k-step (chunk, reg) then contracts
//     rows of one parity only.
// 4 waves per workgroup (one per base tile of the first axis), 64 / 80 KiB of LDS: two workgroups per CU.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <type_traits>
#include "geobo_hip.h"

// Cache policy of the spectrum streams (A/B, round 6): the forward transform's 8 MB / row of output and the inverse transform's input are
// touched once per launch.  XZF_NT_ST: non-temporal stores of the forward output; XZF_NT_LD: aux = nt on the inverse's LDS-DMA loads.
#ifndef XZF_NT_ST
#define XZF_NT_ST 0
#endif
#ifndef XZF_NT_LD
#define XZF_NT_LD 0
#endif

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct FoldArgs {
  const double* in; int64_t in_row, in_plane;    // plane (r, p) at in + r*in_row + p*in_plane (row-major, dense)
  double* out; int64_t out_row, out_plane;
  int64_t out_rs;                                // inverse: doubles between consecutive rows of an output plane (n: dense planes)
  const double* Fz;                              // [n][n/2][2]: (Fe, Fo) of the contiguous axis ("z")
  const double* Fx;                              // the same for the row axis ("x")
  int ppr; int64_t nplanes;
  // forward only, lattice survey (geobo_xz2d_fold_lattice): the planes of an operator row are windows of the stencil table --
  // row r, plane y at in + row_off[r] + y*in_plane, except the first and the last plane of a row (the 1e6-padded boundary slabs),
  // which come from edge + r*edge_row (+ one plane for the last one).  row_off == nullptr: dense rows as above.
  const int64_t* row_off; const double* edge; int64_t edge_row;
  // inverse with the sum-of-squares reduction (geobo_xz2d_fold_inv_ss): rows r >= r2_first are the SUM of two spectra, plane (r, y)
  // of the second one at in2 + (r - r2_first)*in2_row + y*in_plane (both contractions are linear: the first step accumulates over
  // the terms, the second runs once); nothing is stored per plane -- every workgroup keeps sum_r X_r[y]^2 for its fixed y in
  // registers and adds it to ss + (slot*ppr + y)*n*n at the end (slot = blockIdx.x / ppr; the grid is a multiple of ppr)
  const double* in2; int64_t in2_row, r2_first; double* ss;
  // QUAD form (geobo_xz2d_fold_quad; n = 32 planes on the n = 64 kernels): a 2N x 2N / N x N "plane" of the kernel is a 2 x 2 arrangement
  // of four consecutive planes of half the extent -- [[y, y+1], [y+2, y+3]] -- and the folded matrices are those of diag(G32, G32):
  // four independent transforms per kernel plane (half of the MFMAs meet zero blocks; the passes are HBM bound either way).  Memory
  // rows are half as long; the right half of a kernel row and the bottom half of the rows live in other planes:
  //   byte offset of element (row i, 16-byte slot s) = i * rowB + (i >= rows / 2 ? botB : 0) + (s % hs) * 16 + (s / hs) * halfB
  // with hs = slots per memory row.  Dense planes: rowB = the full row, halfB = rowB / 2, botB = 0.
  int in_rowB, in_halfB, in_botB;        // input side (bytes)
  int out_rowB, out_halfB, out_botB;     // output side (bytes; the inverse's output row stride stays out_rs)
};

constexpr int vmcnt_imm(int v) { return (v & 15) | (7 << 4) | (15 << 8) | ((v >> 4) << 14); }

int ensure_lds_attr(std::atomic<uint64_t>& done, const void* kern, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GEOBO_E_LAUNCH;
  if ((done.load(std::memory_order_acquire) >> dev) & 1) return GEOBO_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GEOBO_E_LAUNCH;
  done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
  return GEOBO_OK;
}

// row i of a chunk in LDS <- row rowperm(i) of the chunk in memory: D register reg = i >> 2 of lane group q = i & 3
__device__ __forceinline__ int rowperm(int i) { return 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1); }

// radix-4 analysis along the row axis (round 5): D register reg of lane group q holds row 4 q + reg of the chunk -- register = residue
// class of the input row mod 4, so that an MFMA k-step (k = q) contracts four rows of ONE class
__device__ __forceinline__ int rowperm4(int i) { return 4 * (i & 3) + (i >> 2); }

// ---- forward: X (N x N) -> O (2N x 2N) -------------------------------------------------------------------------------------
// Register budget (two workgroups per CU = 256 VGPRs per wave): the 16 output accumulator tiles of step 2 (128 VGPRs) stay live for
// the whole plane and every chunk is carried through BOTH steps before the next one is touched -- T never exists beyond one
// chunk (16 VGPRs), the barrier of a chunk is amortised over 48 MFMAs.  All LDS addresses are (one per-lane VGPR) + (an
// instruction immediate): ring slot = chunk index because RT % RING == 0, and the XOR swizzle only touches bits the immediates
// leave alone.  (Left to itself the compiler hoists ~60 address registers out of the plane loop and spills them; a spill reload
// is a vmcnt(0) wait in the middle of the DMA pipeline.)
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

template <int N, int RING>
struct FwdCfg {
  static constexpr int NW = 4, RT = N / 16, KP = N / 8, ROWB = N * 8, CHB = 16 * ROWB, ND = CHB / 1024 / NW;
  static constexpr int LPR = ROWB / 16, RPI = 64 / LPR, MT = N / 16;
  static constexpr size_t LDS = (size_t)RING * CHB + (size_t)N * N * 8;
  static_assert(N == 64 && ND >= 1 && RT >= RING - 1 && RT % RING == 0 && N / 16 == NW, "shape");
};

template <int N, bool QUAD>
__global__ void __launch_bounds__(256, 2) xz_fold_fwd_kernel(FoldArgs g) {
  constexpr int RING = 4;
  // dense planes: compile-time strides (the runtime form costs the n = 64 pipeline ~0.7 % of a 64^3 step: A/B on one box)
  const int in_rowB = QUAD ? g.in_rowB : N * 8, in_halfB = QUAD ? g.in_halfB : N * 4, in_botB = QUAD ? g.in_botB : 0;
  const int out_rowB = QUAD ? g.out_rowB : 2 * N * 8, out_halfB = QUAD ? g.out_halfB : N * 8, out_botB = QUAD ? g.out_botB : 0;
  using K = FwdCfg<N, RING>;
  constexpr int RT = K::RT, KP = K::KP, MT = K::MT, H = N / 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* const ring = reinterpret_cast<char*>(smem);
  double* const mxf = smem + RING * K::CHB / 8;               // [N base rows][N/2 slots of (Fe, Fo)], slots XOR-swizzled by row & 15
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring;
  const unsigned mxf_lds = ring_lds + RING * K::CHB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, q = lane >> 4;

  // Radix-4 analysis along x (dense planes; the quad form's block-diagonal matrices keep radix 2).  The inputs x = 4 j + rho of one
  // residue class need, per frequency w < 16, ONE cosine and ONE sine row of the basis (w = 0: the constant row and the alternating
  // one): image [rho][w][j] of pairs (cos, sin) taken from the folded matrix as handed in, g_b[i] = Fx[b][i >> 1][i & 1] with the base
  // rows b = 4 w, 4 w + 1 of spectral.base_modes -- 16 KiB, 16-byte slots XOR-swizzled by w.  Per chunk and class one fragment read
  // feeds FOUR MFMAs (cos | sin tile x the two columns of the pair): 16 per chunk where the radix-2 form issues 32; the partial sums
  // of the four classes meet in the plane's epilogue (eight outputs per frequency from eight sums, all in one lane).
  constexpr bool R4 = !QUAD;
  if constexpr (R4) {
    for (int idx = tid; idx < 4 * 16 * 16; idx += 64 * K::NW) {
      const int rho = idx >> 8, om = (idx >> 4) & 15, j = idx & 15, i = 4 * j + rho;
      const double c = g.Fx[(((int64_t)(4 * om) * H + (i >> 1)) << 1) + (i & 1)];
      const double sn = g.Fx[(((int64_t)(4 * om + 1) * H + (i >> 1)) << 1) + (i & 1)];
      *reinterpret_cast<v2d*>(mxf + ((((rho * 16 + om) * 16) + (j ^ om)) << 1)) = (v2d){c, om ? sn : ((j & 1) ? -c : c)};
    }
  } else {
  for (int idx = tid; idx < N * H; idx += 64 * K::NW) {
    const int bx = idx / H, j = idx % H;
    *reinterpret_cast<v2d*>(mxf + ((bx * H + (j ^ (bx & 15))) << 1)) = *reinterpret_cast<const v2d*>(g.Fx + ((int64_t)idx << 1));
  }
  }
  // radix 4: fragment of class rho for chunk c (k = q <-> j = 4 c + q) at fx4 ^ (c << 6) + rho * 4096
  const unsigned fx4 = mxf_lds + lr * 256 + (((lr & 12) | (q ^ (lr & 3))) << 4);
  double gE[KP], gO[KP];                                      // B[k = q][j = lr]: F?_z[b = 16 w + lr][j = 4 t + q]
#pragma unroll
  for (int t = 0; t < KP; ++t) {
    const v2d f = *reinterpret_cast<const v2d*>(g.Fz + (((int64_t)(16 * w + lr) * H + 4 * t + q) << 1));
    gE[t] = f.x;
    gO[t] = f.y;
  }
  // per-lane LDS addresses.  Input fragment t of the chunk in ring slot c: raddr[t] + c*CHB.
  unsigned raddr[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) raddr[t] = ring_lds + lr * K::ROWB + (((4 * t + q) ^ lr) << 4);
  // Fx fragment (row tile m, k-step (rt, h)): row 16 m + lr, slot (8 rt + 4 h + q) ^ lr = 16 (rt >> 1) + [(8 (rt & 1) + 4 h + q) ^ lr]
  //   -> faddr[rt & 1][h] + m * 16 * H * 16 + (rt >> 1) * 256
  unsigned faddr[2][2];
#pragma unroll
  for (int r1 = 0; r1 < 2; ++r1)
#pragma unroll
    for (int h = 0; h < 2; ++h) faddr[r1][h] = mxf_lds + lr * (H * 16) + (((8 * r1 + 4 * h + q) ^ lr) << 4);

  const int64_t first = blockIdx.x, pstep = gridDim.x;
  if (first >= g.nplanes) return;
  const int drow = lane / K::LPR, dpos = lane % K::LPR;
  unsigned doff[K::ND];                                       // per-lane byte offset of this lane's DMA source inside a chunk
#pragma unroll
  for (int j = 0; j < K::ND; ++j) {
    const int row = (w + K::NW * j) * K::RPI + drow;          // LDS row of the chunk <- memory row rowperm(row)
    const int sl = dpos ^ (row & 15);                         // 16-byte slot of the kernel row (LPR slots; the right half may be another plane)
    doff[j] = (R4 ? rowperm4(row) : rowperm(row)) * in_rowB + ((sl & (K::LPR / 2 - 1)) << 4) + (sl / (K::LPR / 2)) * in_halfB;
  }
  const int in_chunkB = 16 * in_rowB;
  auto plane_ptr = [&](int64_t p) {
    const int64_t r = p / g.ppr;
    const int y = (int)(p % g.ppr);
    const double* q = g.in + r * g.in_row + y * g.in_plane;
    if (g.row_off) {
      q = g.in + g.row_off[r] + y * g.in_plane;
      if (y == 0 || y == g.ppr - 1) q = g.edge + r * g.edge_row + (y ? N * N : 0);
    }
    return reinterpret_cast<const char*>(q);
  };
  auto stage = [&](const char* plane, int c, int slot) {
#pragma unroll
    for (int j = 0; j < K::ND; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(plane + c * in_chunkB + (c >= K::RT / 2 ? in_botB : 0) + doff[j]),
                                       (lds_ptr_t)(ring + slot * K::CHB + (w + K::NW * j) * 1024), 16, 0, 0);
  };
  const char* cur = plane_ptr(first);
  __syncthreads();                                            // Fx image visible
#pragma unroll
  for (int c = 0; c < RING - 1; ++c) stage(cur, c, c);
  // this lane's byte offset inside an output plane: rows 2 q (+ 1), columns 2 (16 w + lr) (+ 1) -- waves 2, 3 write the right half
  const unsigned soff = (unsigned)(q * 2 * out_rowB + ((2 * (16 * w + lr)) & (N - 1)) * 8 + (w >> 1) * out_halfB);
  bool warm = false;
  for (int64_t p = first; p < g.nplanes; p += pstep) {
    const int64_t pn = p + pstep < g.nplanes ? p + pstep : p;
    const char* nxt = plane_ptr(pn);
    v4d e2[MT][2], o2[MT][2];                                 // [base row tile][column 2b | 2b+1]: even-row / odd-row partial sums
    // (radix 4: e2[rho][v] = cosine tile, o2[rho][v] = sine tile of class rho, column variant v)
#pragma unroll
    for (int m = 0; m < MT; ++m) e2[m][0] = e2[m][1] = o2[m][0] = o2[m][1] = (v4d){0., 0., 0., 0.};
    static_for<0, RT>([&](auto cc) {
      constexpr int c = decltype(cc)::value;                  // chunk index = ring slot (RT % RING == 0)
      if (!(c <= RING - 2 && warm)) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * K::ND));
      __builtin_amdgcn_s_barrier();
      {
        constexpr int cn = c + RING - 1;
        if constexpr (cn < RT) stage(cur, cn, cn % RING);
        else stage(nxt, cn - RT, cn % RING);
      }
      // ---- step 1 on chunk c: E over even inputs, O over odd inputs, two chains each ----------------------------------------
      // (E and O alternate, so one accumulator each is two chains: a dependent fp64 MFMA issues back to back; radix 4 needs the
      // registers a second pair would take)
      v4d e[R4 ? 1 : 2], o[R4 ? 1 : 2];
#pragma unroll
      for (int i = 0; i < (R4 ? 1 : 2); ++i) e[i] = o[i] = (v4d){0., 0., 0., 0.};
      static_for<0, 2>([&](auto hb) {                         // two batches of four fragment reads (16 VGPRs in flight)
        constexpr int t0 = 4 * decltype(hb)::value;
        v2d a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[t]) : "v"(raddr[t0 + t]), "n"(c * K::CHB));
        static_for<0, 4>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[t]) : "n"(3 - t));
          constexpr int ch = R4 ? 0 : (t & 1);
          e[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t].x, gE[t0 + t], e[ch], 0, 0, 0);
          o[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t].y, gO[t0 + t], o[ch], 0, 0, 0);
        });
      });
      v4d es = e[0], os = o[0];
      if constexpr (!R4) { es += e[1]; os += o[1]; }
      const v4d tp = es + os, tm = es - os;                   // T[rows of the chunk][column 2b], [column 2b+1]
      if constexpr (R4) {
        // ---- step 2, radix 4: register rho of T = the chunk's rows of class rho; k = q <-> input 4 c + q of the class ---------------
        v2d f[4];
        const unsigned fa = fx4 ^ (unsigned)(c << 6);
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[rho]) : "v"(fa), "n"(rho * 4096));
        static_for<0, 4>([&](auto rr) {
          constexpr int rho = decltype(rr)::value;
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[rho]) : "n"(3 - rho));
          e2[rho][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].x, tp[rho], e2[rho][0], 0, 0, 0);
          e2[rho][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].x, tm[rho], e2[rho][1], 0, 0, 0);
          o2[rho][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].y, tp[rho], o2[rho][0], 0, 0, 0);
          o2[rho][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].y, tm[rho], o2[rho][1], 0, 0, 0);
        });
      } else
      // ---- step 2, k-steps (c, h): register 2h = even rows -> E2, register 2h+1 = odd rows -> O2 ------------------------------
      static_for<0, 2>([&](auto hh) {
        constexpr int h = decltype(hh)::value;
        v2d f[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[m]) : "v"(faddr[c & 1][h]), "n"(m * 16 * H * 16 + (c >> 1) * 256));
        static_for<0, MT>([&](auto mm) {
          constexpr int m = decltype(mm)::value;
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[m]) : "n"(MT - 1 - m));
          e2[m][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].x, tp[2 * h], e2[m][0], 0, 0, 0);
          e2[m][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].x, tm[2 * h], e2[m][1], 0, 0, 0);
          o2[m][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].y, tp[2 * h + 1], o2[m][0], 0, 0, 0);
          o2[m][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].y, tm[2 * h + 1], o2[m][1], 0, 0, 0);
        });
      });
    });
    // the next plane's first RING-1 chunks were requested during the last chunks of this one: drain them here, so that no later
    // wait has this plane's stores between itself and the chunk it waits for
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    char* const op = reinterpret_cast<char*>(g.out + (p / g.ppr) * g.out_row + (p % g.ppr) * g.out_plane) + soff;
    if constexpr (R4) {
      // register r of a lane = frequency w = q + 4 r: its eight sums (C_rho, S_rho) give the eight spectral rows 8 w .. 8 w + 7
      // (cos w, its mirror, sin w, mirror, cos(n/2 - w), mirror, sin(n/2 - w), mirror); w = 0 (lanes q == 0, r = 0): the rows of
      // frequency 0, n, the middle pair and n/4 from the constant-row sums C and the alternating sums S
      char* const op4 = op + (size_t)(8 * q) * out_rowB - (size_t)(q * 2) * out_rowB;      // (soff carries rows 2 q of the radix-2 layout)
      const bool w0 = q == 0;
      const double r2 = 1.4142135623730951;
      static_for<0, 4>([&](auto rr_) {
        constexpr int r = decltype(rr_)::value;
        // one frequency at a time, stores issued as the rows are formed (left to itself the scheduler interleaves the four
        // frequencies and spills: the 16 accumulator tiles are still live here)
        __builtin_amdgcn_sched_barrier(0);
        v2d u0, u1, v0, v1, ws, wd, z0, z1;                    // [column 2b | 2b+1]
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          u0[v] = e2[0][v][r] + e2[2][v][r]; u1[v] = e2[0][v][r] - e2[2][v][r];
          v0[v] = e2[1][v][r] + e2[3][v][r]; v1[v] = e2[1][v][r] - e2[3][v][r];
          ws[v] = o2[0][v][r] + o2[2][v][r]; wd[v] = o2[0][v][r] - o2[2][v][r];
          z0[v] = o2[1][v][r] + o2[3][v][r]; z1[v] = o2[1][v][r] - o2[3][v][r];
        }
        char* const rowp = op4 + (size_t)(32 * r) * out_rowB;
        auto put = [&](int k, v2d val) {
          if constexpr (XZF_NT_ST) __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(rowp + (size_t)k * out_rowB));
          else *reinterpret_cast<v2d*>(rowp + (size_t)k * out_rowB) = val;
        };
        put(0, u0 + v0);
        put(1, u0 - v0);
        if constexpr (r == 0) {
          const v2d a0 = (v2d){r2 * o2[0][0][r], r2 * o2[0][1][r]}, a2 = (v2d){r2 * o2[2][0][r], r2 * o2[2][1][r]};
          put(2, w0 ? u1 + v1 : ws + z0);
          put(3, w0 ? u1 - v1 : ws - z0);
          put(4, w0 ? a0 + z1 : u1 + z1);
          put(5, w0 ? a0 - z1 : u1 - z1);
          put(6, w0 ? a2 + z0 : v1 - wd);
          put(7, w0 ? a2 - z0 : -wd - v1);
        } else {
          put(2, ws + z0);
          put(3, ws - z0);
          put(4, u1 + z1);
          put(5, u1 - z1);
          put(6, v1 - wd);
          put(7, -wd - v1);
        }
      });
    } else
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // base row bx = 16 m + 4 r + q: output rows 2 bx (E2 + O2) and 2 bx + 1 (E2 - O2), columns (2b, 2b+1)
        const v2d plus = (v2d){e2[m][0][r] + o2[m][0][r], e2[m][1][r] + o2[m][1][r]};
        const v2d minus = (v2d){e2[m][0][r] - o2[m][0][r], e2[m][1][r] - o2[m][1][r]};
        char* const rowp = op + (size_t)(2 * (16 * m + 4 * r)) * out_rowB + (m >= MT / 2 ? out_botB : 0);
        if constexpr (XZF_NT_ST) {
          __builtin_nontemporal_store(plus, reinterpret_cast<v2d*>(rowp));
          __builtin_nontemporal_store(minus, reinterpret_cast<v2d*>(rowp + out_rowB));
        } else {
          *reinterpret_cast<v2d*>(rowp) = plus;
          *reinterpret_cast<v2d*>(rowp + out_rowB) = minus;
        }
      }
    warm = true;
    cur = nxt;
  }
}

// ---- inverse: S (2N x 2N) -> X (N x N) -------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void lds_read_b128(v2d& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void wait_lgkmcnt(v2d& d) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(CNT));
}

template <int KP, int T>
__device__ __forceinline__ void inv_step1(v2d (&a)[KP], const double (&gz)[KP], double sgn, v4d (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[T]) : "n"(KP - 1 - T));
  const double uv = __builtin_fma(sgn, a[T].y, a[T].x);       // u = s[2b] + s[2b+1] (even outputs) or v = s[2b] - s[2b+1] (odd)
  d[T & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv, gz[T], d[T & 3], 0, 0, 0);
  if constexpr (T + 1 < KP) inv_step1<KP, T + 1>(a, gz, sgn, d);
}

// the same over one half of the k-steps (MUL mode: 8 fragment reads in flight instead of 16 -- the registers the staged loads need)
template <int KH, int T, int G0, int KP>
__device__ __forceinline__ void inv_step1h(v2d (&a)[KH], const double (&gz)[KP], double sgn, v4d (&d)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[T]) : "n"(KH - 1 - T));
  const double uv = __builtin_fma(sgn, a[T].y, a[T].x);
  d[T & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv, gz[G0 + T], d[T & 1], 0, 0, 0);   // two chains: a dependent fp64 MFMA issues back to back
  if constexpr (T + 1 < KH) inv_step1h<KH, T + 1, G0, KP>(a, gz, sgn, d);
}

// rolling window (MUL mode): KW fragment reads in flight at all times -- element T + KW is requested into the register pair element T
// has just been taken out of, so only the first KW reads of a chunk are exposed
template <int KW, int T, int KP>
__device__ __forceinline__ void inv_step1r(v2d (&a)[KW], const double (&gz)[KP], double sgn, v4d (&d)[2], unsigned xs, int q, int lr) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[T % KW]) : "n"(T + KW < KP ? KW - 1 : KP - 1 - T));
  const double uv = __builtin_fma(sgn, a[T % KW].y, a[T % KW].x);
  if constexpr (T + KW < KP)
    asm volatile("ds_read_b128 %0, %1" : "=v"(a[T % KW]) : "v"(xs + (((4 * (T + KW) + q) ^ lr) << 4)), "v"(uv));   // (after uv was formed)
  d[T & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv, gz[T], d[T & 1], 0, 0, 0);
  if constexpr (T + 1 < KP) inv_step1r<KW, T + 1, KP>(a, gz, sgn, d, xs, q, lr);
}

// ---- radix-4 synthesis along z (round 5; dense planes) ---------------------------------------------------------------------------
// The basis of spectral.base_modes comes in groups of eight spectral positions 8 w .. 8 w + 7 = the base rows cos w, sin w,
// cos(n/2 - w), sin(n/2 - w) with their mirrors (w = 1 .. 15; w = 0: cos 0, the middle pair, cos 16, sin 16).  On the outputs
// z = 4 j + rho of ONE residue class the rows of frequency n/2 -+ w are +-(cos | sin) of frequency w, so
//     x[4j + rho] = sum_w  g_cos(w)[4j + rho] C'_rho(w) + g_sin(w)[4j + rho] S'_rho(w)
// with two signed sums C', S' of the eight values of a group (for w = 0: the constant row and the alternating one; tests/
// test_spectral_cpu.py::test_radix4_synthesis_identity has the signs).  Wave rho of a workgroup owns class rho: per 16-row chunk it
// reads the same 16 x 16 bytes per lane as the radix-2 form, forms (C', S') for the four frequencies w = t + 4 q of a k-step with
// six multiply-adds, and issues EIGHT MFMAs against the cos / sin rows of its class where the radix-2 form issues sixteen.
// The matrix operand is unchanged (the folded pairs (Fe, Fo) of the base rows: g_b[i] = F[b][i >> 1][i & 1]); only the ORDER of the
// base rows is assumed.  The quad form (four 32 x 32 planes, block-diagonal matrices) keeps radix 2.
struct R4Lane {
  double s1, s2, s3;            // wave-uniform signs of class rho
  double cB, cP, dB, dP, dQ;    // per lane, k-step 0: C' = A + cB B + cP P,  S' = dB B + dP P + dQ Q  (w = 0 in the lanes q == 0)
  int eP, eQ;                   // 16-byte slots (of the group's four) that feed P and Q
};

__device__ __forceinline__ R4Lane r4_lane(int rho, int q) {
  R4Lane c;
  const double r2 = 1.4142135623730951;
  c.s1 = (rho & 1) ? -1.0 : 1.0;
  c.s2 = rho >= 2 ? -1.0 : 1.0;
  c.s3 = (rho == 1 || rho == 2) ? 1.0 : -1.0;
  c.eP = (rho & 1) ? 3 : 2;
  c.eQ = (rho & 1) ? 2 : 3;
  if (q == 0) {                 // the group of w = 0: C' = A + t1 B,  S' = kP P + kQ Q
    c.cB = rho < 2 ? 1.0 : -1.0; c.cP = 0.0; c.dB = 0.0;
    c.dP = rho == 0 ? r2 : rho == 2 ? 0.0 : 1.0;
    c.dQ = rho == 0 ? 0.0 : rho == 1 ? 1.0 : rho == 2 ? r2 : -1.0;
  } else {
    c.cB = 0.0; c.cP = c.s2; c.dB = 1.0; c.dP = 0.0; c.dQ = c.s3;
  }
  return c;
}

// the four 16-byte reads of group T of a row (slots A, B and the class's P, Q); SB: byte offset of the ring slot (an immediate)
template <int T, int SB>
__device__ __forceinline__ void r4_issue(v2d (&oo)[4], unsigned xq, unsigned pP, unsigned pQ) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(oo[0]) : "v"(xq ^ (unsigned)((4 * T) << 4)), "n"(SB));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(oo[1]) : "v"(xq ^ (unsigned)((4 * T + 1) << 4)), "n"(SB));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(oo[2]) : "v"(xq ^ ((unsigned)((4 * T) << 4) | pP)), "n"(SB));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(oo[3]) : "v"(xq ^ ((unsigned)((4 * T) << 4) | pQ)), "n"(SB));
}

// the two signed sums of one group (o[0..3] = the slots A, B, P, Q: pairs (s[2b], s[2b+1]) of the group's four base rows)
template <bool FIRST>
__device__ __forceinline__ void r4_sums(const v2d (&o)[4], const R4Lane& c, double& Cp, double& Sp) {
  const double A_ = __builtin_fma(c.s1, o[0].y, o[0].x), B_ = __builtin_fma(c.s1, o[1].y, o[1].x);
  const double P_ = __builtin_fma(c.s1, o[2].y, o[2].x), Q_ = __builtin_fma(c.s1, o[3].y, o[3].x);
  if constexpr (FIRST) {
    Cp = __builtin_fma(c.cP, P_, __builtin_fma(c.cB, B_, A_));
    Sp = __builtin_fma(c.dQ, Q_, __builtin_fma(c.dP, P_, c.dB * B_));
  } else {
    Cp = __builtin_fma(c.s2, P_, A_);
    Sp = __builtin_fma(c.s3, Q_, B_);
  }
}

template <int N, int RING>
struct InvCfg {
  static constexpr int NW = 4, P = 2 * N, RT = P / 16, KP = N / 4, ROWB = P * 8, CHB = 16 * ROWB, ND = CHB / 1024 / NW;
  static constexpr int LPR = ROWB / 16, RPI = 64 / LPR > 0 ? 64 / LPR : 1, MT = N / 32;
  static constexpr size_t LDS = (size_t)RING * CHB + (size_t)N * N * 8;
  static_assert(N == 64 && ND >= 1 && RT >= RING - 1 && N / 16 == NW && LPR == 64, "shape");
};

// MODE 0: planes stored; 1 (RED): squared and summed over the rows (geobo_xz2d_fold_inv_ss); 2 (MUL): stored, and the input plane
// (r, iz) is the PRODUCT of two cache-resident planes, in[iz] * in2[r] (geobo_xz2d_fold_inv_mul: rows of L^-1 A on a lattice survey,
// W = Lambda[iz] * lhat_r never exists in memory).  MUL stages its chunks through registers instead of LDS-DMA: every thread loads
// its 4 x 16 bytes of both factors two chunks ahead (two register sets), multiplies and writes the ring slot itself; the chunk
// barrier waits for those LDS writes only.
template <int N, int MODE, bool QUAD>
__global__ void __launch_bounds__(256, 2) xz_fold_inv_kernel(FoldArgs g) {
  constexpr bool RED = MODE == 1, MUL = MODE == 2;
  const int in_rowB = QUAD ? g.in_rowB : 2 * N * 8, in_halfB = QUAD ? g.in_halfB : N * 8, in_botB = QUAD ? g.in_botB : 0;
  const int out_halfB = QUAD ? g.out_halfB : N * 4, out_botB = QUAD ? g.out_botB : 0;
  // Radix 4 (dense planes) along BOTH axes: along z wave w owns the outputs z = 4 (lane column) + w; along x (the spectral rows of a
  // plane) the chunks come in pairs -- the 32 rows of a pair are four groups of eight spectral positions, and lane group q holds one
  // whole group in the four D registers of the pair's two chunks (the DMA gathers the rows accordingly: chunk 2 c' + h takes rows
  // 32 c' + 8 q + 4 h + r into LDS row q + 4 r).  The signed sums (C', S') of the four output classes x = 4 j + rho are formed in the
  // lane and contracted against the class's cos / sin rows: 8 MFMAs per pair where the radix-2 form issues 16.  The class image of
  // the x matrix is 16 KiB (the radix-2 image 32): the ring takes a FOURTH slot in the same 80 KiB -- three chunks in flight per
  // workgroup instead of two (the kernel is bound by the loads it can keep outstanding).
  constexpr bool R4 = !QUAD;
  constexpr int RING = R4 ? 4 : 3;
  using K = InvCfg<N, RING>;
  constexpr int RT = K::RT, KP = K::KP, MT = K::MT, H = N / 2;
  static_assert((size_t)RING * K::CHB + (R4 ? 16384 : N * N * 8) <= InvCfg<N, 3>::LDS, "LDS");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* const ring = reinterpret_cast<char*>(smem);
  double* const mxf = smem + RING * K::CHB / 8;               // [N/2 output pairs j][N slots bx of (Fe, Fo)[bx][j]], slots XOR row & 15
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, q = lane >> 4;
  const int par = w >> 1, jt = w & 1;                         // radix 2: outputs z = 2 (16 jt + lane column) + par

  if constexpr (R4) {
    // image [rho][j][w] of pairs (cos, sin): rows of frequency w at the outputs x = 4 j + rho (w = 0: constant and alternating row)
    for (int idx = tid; idx < 4 * 16 * 16; idx += 64 * K::NW) {
      const int rho = idx >> 8, j = (idx >> 4) & 15, om = idx & 15, i = 4 * j + rho;
      const double c = g.Fx[(((int64_t)(4 * om) * H + (i >> 1)) << 1) + (i & 1)];
      const double sn = g.Fx[(((int64_t)(4 * om + 1) * H + (i >> 1)) << 1) + (i & 1)];
      *reinterpret_cast<v2d*>(mxf + ((((rho * 16 + j) * 16) + (om ^ j)) << 1)) = (v2d){c, om ? sn : ((j & 1) ? -c : c)};
    }
  } else {
  for (int idx = tid; idx < N * H; idx += 64 * K::NW) {
    const int bx = idx / H, j = idx % H;
    *reinterpret_cast<v2d*>(mxf + ((j * N + (bx ^ (j & 15))) << 1)) = *reinterpret_cast<const v2d*>(g.Fx + ((int64_t)idx << 1));
  }
  }
  // radix 4: fragment of class rho for the chunk pair c' (k = q <-> frequency 4 c' + q) at fx4 ^ (c' << 6) + rho * 4096
  const unsigned fx4 = (unsigned)(uintptr_t)(lds_ptr_t)mxf + lr * 256 + (((lr & 12) | (q ^ (lr & 3))) << 4);
  double gz[R4 ? 1 : KP];                                     // B[k = q][j = lr]: F{e|o}_z[b = 4 t + q][j = 16 jt + lr]
  double gzc[4], gzs[4];                                      // radix 4: B[k = q][j = lr] = (cos | sin) row of frequency t + 4 q at z = 4 lr + w
  if constexpr (R4) {
    const int i = 4 * lr + w;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int om = t + 4 * q;
      const double c = g.Fz[(((int64_t)(4 * om) * H + (i >> 1)) << 1) + (i & 1)];
      const double sn = g.Fz[(((int64_t)(4 * om + 1) * H + (i >> 1)) << 1) + (i & 1)];
      gzc[t] = c;
      gzs[t] = om ? sn : ((lr & 1) ? -c : c);                // frequency 0: the constant row and the alternating one
    }
    gz[0] = 0.0;
  } else {
#pragma unroll
    for (int t = 0; t < KP; ++t) gz[t] = g.Fz[(((int64_t)(4 * t + q) * H + 16 * jt + lr) << 1) + par];
  }
  const double sgn = par ? -1.0 : 1.0;
  const R4Lane r4 = r4_lane(w, q);
  // radix 4: slot (16 q + c) ^ lr of row lr of ring slot s = (xq0 ^ (c << 4)) + s * CHB
  const unsigned xq0 = ring_lds + lr * K::ROWB + (unsigned)((q << 8) | (lr << 4));
  // step-2 fragments (Fe_x, Fo_x)[bx = 8 rt + 4 h + q][j = 16 m + lr]: row j of the image, slot bx ^ (j & 15) = 16 (rt >> 1) +
  // [(8 (rt & 1) + 4 h + q) ^ lr]  ->  faddr[rt & 1][h] + m * 16 * N * 16 + (rt >> 1) * 256  (one per-lane VGPR + an immediate)
  unsigned faddr[2][2];
#pragma unroll
  for (int r1 = 0; r1 < 2; ++r1)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      faddr[r1][h] = (unsigned)(uintptr_t)(lds_ptr_t)mxf + lr * (N * 16) + (((8 * r1 + 4 * h + q) ^ lr) << 4);

  const int64_t first = blockIdx.x, pstep = gridDim.x;
  if (first >= g.nplanes) return;
  // source planes: (output plane p, term); rows r >= r2_first of the reduction form have a second term
  auto nterms = [&](int64_t p) { return (RED && g.in2 && p / g.ppr >= g.r2_first) ? 2 : 1; };
  auto plane_ptr = [&](int64_t p, int term) {
    const int64_t r = p / g.ppr;
    if (RED && term == 1) return g.in2 + (r - g.r2_first) * g.in2_row + (p % g.ppr) * g.in_plane;
    return g.in + r * g.in_row + (p % g.ppr) * g.in_plane;
  };
  auto stage = [&](const double* plane, int c, int slot) {
#pragma unroll
    for (int j = 0; j < K::ND; ++j) {
      const int row = w + K::NW * j;                          // one 1-KiB row per DMA instruction
      const int sl = lane ^ (row & 15);
      const int mrow = R4 ? 32 * (c >> 1) + 8 * (row & 3) + 4 * (c & 1) + (row >> 2) : 16 * c + rowperm(row);
      const char* src = reinterpret_cast<const char*>(plane) + (int64_t)mrow * in_rowB + (c >= RT / 2 ? in_botB : 0) +
                        ((sl & 31) << 4) + (sl >> 5) * in_halfB;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(ring + slot * K::CHB + row * 1024), 16, 0, XZF_NT_LD ? 2 : 0);
    }
  };
  // MUL: the two factors of plane p; fetch = this thread's 16-byte pieces of chunk c of both (row w + 4 j of the chunk, slot lane),
  // commit = their products into the ring slot, where the DMA of the other modes would have put the chunk
  // MUL: the two factors of plane p through buffer descriptors (uniform base in SGPRs + one 32-bit lane offset per piece + the
  // chunk's scalar offset: no 64-bit address arithmetic per load); fetch = this thread's 16-byte pieces of chunk c of both factors
  // (row w + 4 j of the chunk, slot lane), commit = their products into the ring slot, where the DMA of the other modes would have
  // put the chunk
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  using u32x4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(*static_cast<rsrc_t*>(nullptr), 0, 0, 0));
  auto planeA = [&](int64_t p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.in + (p % g.ppr) * g.in_plane), 0, RT * K::CHB, 0x00020000);
  };
  auto planeB = [&](int64_t p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.in2 + (p / g.ppr) * g.in2_row), 0, RT * K::CHB, 0x00020000);
  };
  v2d fa[2][K::ND], fb[2][K::ND];                             // two sets: chunk k lives in set k % 2 from its fetch (chunk k - 3) to its commit
  unsigned fo[K::ND];                                         // byte offset of this thread's piece j inside a chunk (32-bit, per lane)
#pragma unroll
  for (int j = 0; j < K::ND; ++j) {
    const int row = w + K::NW * j;
    fo[j] = (unsigned)((R4 ? 8 * (row & 3) + (row >> 2) : rowperm(row)) * K::ROWB + ((lane ^ (row & 15)) << 4));
  }
  auto fetch = [&](rsrc_t ra, rsrc_t rb, int c) {
    const int cb = R4 ? (32 * (c >> 1) + 4 * (c & 1)) * K::ROWB : c * K::CHB;     // first row of the chunk's gather
#pragma unroll
    for (int j = 0; j < K::ND; ++j) {
      fa[c & 1][j] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(ra, fo[j], cb, 0));
      fb[c & 1][j] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rb, fo[j], cb, 0));
    }
  };
  auto commit = [&](int c, int slot) {
    // pin the staged registers HERE: left alone, the scheduler forms the products right behind the loads (fewer live registers) and
    // waits for loads it has just issued in the middle of the previous chunk
#pragma unroll
    for (int j = 0; j < K::ND; ++j) asm volatile("" : "+v"(fa[c & 1][j]), "+v"(fb[c & 1][j]));
#pragma unroll
    for (int j = 0; j < K::ND; ++j) {
      const int row = w + K::NW * j;
      *reinterpret_cast<v2d*>(ring + slot * K::CHB + row * 1024 + (lane << 4)) = fa[c & 1][j] * fb[c & 1][j];
    }
  };
  const double* cur = MUL ? nullptr : plane_ptr(first, 0);
  rsrc_t curA = planeA(MUL ? first : 0), curB = planeB(MUL ? first : 0);
  __syncthreads();
  if constexpr (MUL) {
    fetch(curA, curB, 0);
    fetch(curA, curB, 1);
    commit(0, 0);
    fetch(curA, curB, 2);
  } else {
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) stage(cur, c, c);
  }
  int slot0 = 0;
  bool warm = false;
  v4d sse[MT], sso[MT];                                       // RED: sum over this workgroup's planes of the squared outputs
#pragma unroll
  for (int m = 0; m < MT; ++m) sse[m] = sso[m] = (v4d){0., 0., 0., 0.};
  for (int64_t p = first; p < g.nplanes; p += pstep) {
    const int nt = nterms(p);
    // Output tiles of the plane (even / odd output rows x two accumulator chains), carried across the chunks: every 16-row chunk goes
    // through BOTH contractions before the next one is touched (round 4; like the forward kernel).  With step 2 as one burst of 64
    // MFMAs at the end of the plane the chunk stream ran dry behind every plane (only RING - 1 chunks of the next plane are requested
    // during the burst): 55-64 % MFMA busy.  Terms of a two-term row simply accumulate (both contractions are linear).
    v4d xe[MT][2], xo[MT][2];                                 // (radix 4: xe[rho >> 1][rho & 1] = the outputs x = 4 j + rho)
#pragma unroll
    for (int m = 0; m < MT; ++m) xe[m][0] = xe[m][1] = xo[m][0] = xo[m][1] = (v4d){0., 0., 0., 0.};
    v4d tcA = (v4d){0., 0., 0., 0.};                          // radix 4: step-1 result of the pair's first chunk
    for (int term = 0; term < nt; ++term) {
    const bool last_term = term + 1 == nt;
    const int64_t pn = last_term ? (p + pstep < g.nplanes ? p + pstep : p) : p;
    const double* nxt = MUL ? nullptr : plane_ptr(pn, last_term ? 0 : term + 1);
    const rsrc_t nxtA = planeA(MUL ? pn : 0), nxtB = planeB(MUL ? pn : 0);
    static_for<0, RT>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (MUL) {
        commit(c + 1, (slot0 + c + 1) % RING);                // chunk c + 1 (fetched during chunk c - 2): its slot was last read at c - 2
        // the ring writes must have landed, the staged loads of chunk c + 2 must NOT be waited for (__syncthreads drains vmcnt too)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 3 < RT) fetch(curA, curB, c + 3);             // into the set the commit has just freed
        else if (c + 3 < RT + 2) fetch(nxtA, nxtB, c + 3 - RT);
        // (chunk 2 of the next plane is fetched behind the plane's stores: a second set of 16 registers does not fit beside the tiles)
      } else {
      // (reduction form: no stores inside the loop, hence no drain before them -- every chunk takes the counted wait)
      if (RED || !(c <= RING - 2 && warm)) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * K::ND));
      __builtin_amdgcn_s_barrier();
      {
        const int cn = c + RING - 1;
        if (cn < RT) stage(cur, cn, (slot0 + cn) % RING);
        else stage(nxt, cn - RT, (slot0 + cn) % RING);
      }
      }
      const unsigned xs = ring_lds + ((slot0 + c) % RING) * K::CHB + lr * K::ROWB;
      v4d tc;
      if constexpr (R4) {
        // (ring_lds is the start of the dynamic LDS segment and the only __shared__ object: bits 4..9 of a row's address are clear.
        // RT % RING == 0: the ring slot of chunk c is c % RING for every plane -- an immediate; the per-lane part is made opaque per
        // chunk, or the 128 read addresses of a plane are hoisted out of the plane loop as loop invariants and spill)
        static_assert(RT % RING == 0, "ring slot = chunk index");
        unsigned xq = xq0;
        asm volatile("" : "+v"(xq));
        constexpr int SB = (c % RING) * K::CHB;
        const unsigned pP = (unsigned)r4.eP << 4, pQ = (unsigned)r4.eQ << 4;
        v4d d2[2];
        d2[0] = d2[1] = (v4d){0., 0., 0., 0.};
        constexpr int NF = MUL ? 2 : 4;                           // groups in flight (MUL: the staged loads need the registers)
        v2d o[NF][4];
        r4_issue<0, SB>(o[0], xq, pP, pQ);
        r4_issue<1, SB>(o[1], xq, pP, pQ);
        if constexpr (NF == 4) {
          r4_issue<2, SB>(o[2], xq, pP, pQ);
          r4_issue<3, SB>(o[3], xq, pP, pQ);
        }
        static_for<0, 4>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          constexpr int left = 4 * ((t + NF < 4 ? NF : 4 - t) - 1);   // reads still allowed in flight when group t is taken
          v2d (&ot)[4] = o[t % NF];
          asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ot[0]), "+v"(ot[1]), "+v"(ot[2]), "+v"(ot[3]) : "n"(left));
          double Cp, Sp;
          r4_sums<t == 0>(ot, r4, Cp, Sp);
          if constexpr (t + NF < 4) {
            // (behind the sums: the registers of group t are free only now)
            asm volatile("" : "+v"(Cp), "+v"(Sp));
            r4_issue<t + NF, SB>(ot, xq, pP, pQ);
          }
          d2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cp, gzc[t], d2[0], 0, 0, 0);
          d2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Sp, gzs[t], d2[1], 0, 0, 0);
        });
        tc = d2[0] + d2[1];
      } else if constexpr (MUL) {
        constexpr int KH = KP / 2;
        v2d ah[KH];
        v4d d2[2];
        d2[0] = d2[1] = (v4d){0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < KH; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(ah[t]) : "v"(xs + (((4 * t + q) ^ lr) << 4)));
        inv_step1r<KH, 0, KP>(ah, gz, sgn, d2, xs, q, lr);
        tc = d2[0] + d2[1];
      } else {
      v4d d[4];
      d[0] = d[1] = d[2] = d[3] = (v4d){0., 0., 0., 0.};
      v2d a[KP];
#pragma unroll
      for (int t = 0; t < KP; ++t) {
        const unsigned addr = xs + (((4 * t + q) ^ lr) << 4);    // slot b = 4 t + q: the pair (s[2b], s[2b+1]) of row lr
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[t]) : "v"(addr));
      }
      inv_step1<KP, 0>(a, gz, sgn, d);
      tc = (d[0] + d[1]) + (d[2] + d[3]);
      }
      if constexpr (R4) {
        // ---- step 2, radix 4, on the second chunk of a pair: the lane's group of eight = registers 0..3 of both chunks -------------
        if constexpr ((c & 1) == 0) {
          tcA = tc;
        } else {
          constexpr int cp = c >> 1;
          v2d f[4];
          const unsigned fa4 = fx4 ^ (unsigned)(cp << 6);
#pragma unroll
          for (int rho = 0; rho < 4; ++rho) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[rho]) : "v"(fa4), "n"(rho * 4096));
          const double Ap = tcA[0] + tcA[1], Am = tcA[0] - tcA[1], Bp = tcA[2] + tcA[3], Bm = tcA[2] - tcA[3];
          const double Pp = tc[0] + tc[1], Pm = tc[0] - tc[1], Qp = tc[2] + tc[3], Qm = tc[2] - tc[3];
          double Cs[4] = {Ap + Pp, Am + Qm, Ap - Pp, Am - Qm};
          double Ss[4] = {Bp - Qp, Bm + Pm, Bp + Qp, Bm - Pm};
          if constexpr (cp == 0) {
            // frequency 0 (lanes q == 0): C' = A +- B,  S' = sqrt(2) P+ | P- + Q- | sqrt(2) Q+ | Q- - P-
            const bool w0 = q == 0;
            const double r2 = 1.4142135623730951;
            Cs[0] = w0 ? Ap + Bp : Cs[0]; Cs[1] = w0 ? Am + Bm : Cs[1]; Cs[2] = w0 ? Ap - Bp : Cs[2]; Cs[3] = w0 ? Am - Bm : Cs[3];
            Ss[0] = w0 ? r2 * Pp : Ss[0]; Ss[1] = w0 ? Pm + Qm : Ss[1]; Ss[2] = w0 ? r2 * Qp : Ss[2]; Ss[3] = w0 ? Qm - Pm : Ss[3];
          }
          static_for<0, 4>([&](auto rr) {
            constexpr int rho = decltype(rr)::value;
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[rho]) : "n"(3 - rho));
            xe[rho >> 1][rho & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].x, Cs[rho], xe[rho >> 1][rho & 1], 0, 0, 0);
          });
          static_for<0, 4>([&](auto rr) {
            constexpr int rho = decltype(rr)::value;
            xe[rho >> 1][rho & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[rho].y, Ss[rho], xe[rho >> 1][rho & 1], 0, 0, 0);
          });
        }
      } else
      // ---- step 2 on the chunk's rows: row pairs (2 bx, 2 bx + 1) sit in registers (2h, 2h+1): U = sum, V = difference; even output
      //      rows from U with Fe_x, odd ones from V with Fo_x.  The fragment reads are issued behind step 1's (their latency runs under
      //      the tail of its MFMAs) and are inline asm: for LDS reads it can see the compiler drains every outstanding LDS-DMA first.
      static_for<0, 2>([&](auto hh) {
        constexpr int h = decltype(hh)::value;
        v2d f[MT];                                                // (per h: the second pair's reads run under the first pair's MFMAs)
        static_for<0, MT>([&](auto mm) {
          constexpr int m = decltype(mm)::value;
          lds_read_b128<m * 16 * N * 16 + (c >> 1) * 256>(f[m], faddr[c & 1][h]);
        });
        const double u = tc[2 * h] + tc[2 * h + 1], v = tc[2 * h] - tc[2 * h + 1];
        static_for<0, MT>([&](auto mm) {
          constexpr int m = decltype(mm)::value;
          wait_lgkmcnt<MT - 1 - m>(f[m]);
          xe[m][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].x, u, xe[m][h], 0, 0, 0);
          xo[m][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[m].y, v, xo[m][h], 0, 0, 0);
        });
      });
    });
    warm = true;
    slot0 = (slot0 + RT) % RING;
    cur = nxt;
    curA = nxtA;
    curB = nxtB;
    }   // terms
    if constexpr (RED && R4) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        sse[m] += xe[m][0] * xe[m][0];                          // (sse[m] <-> class 2 m, sso[m] <-> class 2 m + 1)
        sso[m] += xe[m][1] * xe[m][1];
      }
    } else if constexpr (RED) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const v4d ev = xe[m][0] + xe[m][1], od = xo[m][0] + xo[m][1];
        sse[m] += ev * ev;
        sso[m] += od * od;
      }
    } else {
      if constexpr (MUL) fetch(curA, curB, 2);                // (curA / curB already name the next plane)
      else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
      // columns 2 (16 jt + lr) + par: wave pair jt = 1 writes the right half (another plane in the quad form); rows >= N / 2 the bottom half
      int64_t ors = g.out_rs;
      asm volatile("" : "+s"(ors));                           // (per plane: keeps the 16 store offsets from being hoisted into 32 live VGPRs)
      double* const op = g.out + (p / g.ppr) * g.out_row + (p % g.ppr) * g.out_plane + 2 * q * ors +
                         (R4 ? 4 * lr + w : 2 * lr + par + jt * (out_halfB >> 3));
      if constexpr (R4) {
        // output row x = 4 (q + 4 r) + rho: the lane part 4 q sits in op4
        double* const op4 = op + (int64_t)(2 * q) * ors;        // (op carries rows 2 q of the radix-2 layout)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
          for (int r = 0; r < 4; ++r) op4[(int64_t)(16 * r + rho) * ors] = xe[rho >> 1][rho & 1][r];
      } else
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const v4d ev = xe[m][0] + xe[m][1], od = xo[m][0] + xo[m][1];
        double* const om = op + (m >= MT / 2 ? (out_botB >> 3) : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j0 = 16 * m + 4 * r;                      // output rows 2 (j0 + q) (even) and + 1 (odd): the lane part sits in op
          om[(int64_t)(2 * j0) * ors] = ev[r];
          om[(int64_t)(2 * j0 + 1) * ors] = od[r];
        }
      }
    }
  }
  if constexpr (RED) {
    // this workgroup's planes all have y = first % ppr (the grid is a multiple of ppr): add its sums to its own partial plane
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    double* const sp = g.ss + ((int64_t)(blockIdx.x / g.ppr) * g.ppr + (first % g.ppr)) * (N * N) + (R4 ? 4 * lr + w : 2 * (16 * jt + lr) + par);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (R4) {
          const int x0 = 4 * (q + 4 * r) + 2 * m;             // classes 2 m (sse) and 2 m + 1 (sso)
          sp[(int64_t)x0 * N] += sse[m][r];
          sp[(int64_t)(x0 + 1) * N] += sso[m][r];
        } else {
        const int j = 16 * m + q + 4 * r;
        sp[(int64_t)(2 * j) * N] += sse[m][r];
        sp[(int64_t)(2 * j + 1) * N] += sso[m][r];
        }
      }
  }
}

// ---- lattice Gram, x step + eigenvalue scaling + channel sum (folded form of xcorr_kernel in xz2d.hip) -------------------------
// For every (row r, y-mode p) plane X (N x N, x rows, z contiguous):   out[r][p][o] = sum_z lamT[p][z][o] * sum_x G_x[o][x] X[x][z].
// The product is formed transposed (D[z][o]: the z sum then runs over accumulator registers and the four 16-lane groups).
struct XCFArgs {
  const double* in; int64_t in_row, in_plane;   // plane (r, p) at in + r*in_row + p*in_plane
  const double* F;                              // [N][N/2][2] folded x matrices
  const double* lamT;                           // [planes][N][2N]
  double* out; int64_t out_row, out_plane;      // out + r*out_row + p*out_plane + o
  int64_t rows, nplanes;
};

// RADIX 4 (round 5; the radix-2 kernel of round 2 -- E / O sums over the parity of x, 64 MFMAs per chunk on eight waves -- is retired).
// The inputs x = 4 j + rho of one residue class need ONE cosine and ONE sine row of the basis per frequency w < 16 (spectral.base_modes:
// the eight spectral positions 8 w .. 8 w + 7 are the orbit of w under a quarter-period shift), so D[z][.] costs 32 MFMAs per 16-row
// chunk instead of 64.  Four waves = the four z tiles; a wave contracts the cosine and the sine rows for the four classes (8 MFMAs per
// chunk; register rho of a lane group = class rho, one 16-byte image read per class feeds both), holds the eight sums of a frequency
// in ONE lane and forms its eight outputs there, scales them by the eight eigenvalues of (z, w) and sums over z (registers, lane
// groups, waves) as before.  256 threads, 56 KiB of LDS: two workgroups per CU with the registers of two waves per SIMD.
template <int N>
__global__ void __launch_bounds__(256, 2) xcorr_fold4_kernel(XCFArgs g) {
  constexpr int RING = 4, NW = 4, PX = 2 * N, H = N / 2, NCH = N / 16, ROWB = N * 8, CHB = 16 * ROWB, LPR = ROWB / 16, RPI = 64 / LPR;
  constexpr int ND = CHB / 1024 / NW;
  static_assert(N == 64 && ND == 2 && NCH >= RING - 1, "shape");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* const ring = reinterpret_cast<char*>(smem);
  double* const img = smem + RING * CHB / 8;           // [rho][j][w] pairs (cosine, sine) of frequency w at the input x = 4 j + rho
  double* const red = img + 4 * 16 * 16 * 2;           // [2][4 waves][PX]
  const unsigned ring_lds = (unsigned)(uintptr_t)(lds_ptr_t)ring;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // z tile
  const int lr = lane & 15, q = lane >> 4;
  for (int idx = tid; idx < 4 * 16 * 16; idx += 64 * NW) {
    const int rho = idx >> 8, j = (idx >> 4) & 15, om = idx & 15, i = 4 * j + rho;
    const double c = g.F[(((int64_t)(4 * om) * H + (i >> 1)) << 1) + (i & 1)];
    const double sn = g.F[(((int64_t)(4 * om + 1) * H + (i >> 1)) << 1) + (i & 1)];
    *reinterpret_cast<v2d*>(img + 2 * idx) = (v2d){c, om ? sn : ((j & 1) ? -c : c)};     // (w = 0: constant and alternating row)
  }
  // Workgroup b keeps ONE y-mode pl = b % planes and walks the rows b / planes, + gridDim.x / planes, ... (the grid is a multiple of the
  // y-mode count): the eigenvalues of its (z, w) stay in registers for the whole launch -- reloaded per plane they were 64 KiB of L2
  // traffic per 32 KiB of input and a fifth of the launch (0.38 -> 0.31 ms with the loads taken out)
  const int64_t nky = g.nplanes / g.rows, pl = blockIdx.x % nky, rstep = gridDim.x / nky;
  const int64_t rfirst = blockIdx.x / nky;
  if (rfirst >= g.rows) return;
  // WAVE-PRIVATE staging: a wave needs only ITS 16 columns of a chunk (the z tiles are independent until the final sum), so it fetches
  // them itself -- 16 rows x 128 bytes = two DMA instructions per chunk into its own quarter of the ring -- and waits on its own vmcnt:
  // no workgroup barrier per chunk (the shared-chunk form had five barriers per plane on four waves: MFMA busy 36 %).  Ring position
  // p = 4 rho + q of a chunk holds its row 4 q + rho: the rows of lane groups q, q + 1 sit in different 128-byte halves of a bank row.
  auto plane_ptr = [&](int64_t row) { return g.in + row * g.in_row + pl * g.in_plane; };
  auto stage = [&](const double* plane, int c, int slot) {
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int pos = 8 * j + (lane >> 3), row = 4 * (pos & 3) + (pos >> 2);
      const char* src = reinterpret_cast<const char*>(plane) + (int64_t)(16 * c + row) * ROWB + w * 128 + ((lane & 7) << 4);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(ring + (w * RING + slot) * (CHB / NW) + j * 1024), 16, 0, 0);
    }
  };
  const double* cur = plane_ptr(rfirst);
  using lrsrc_t = __amdgpu_buffer_rsrc_t;
  const unsigned lofs = (unsigned)(((16 * w + q) * PX + 8 * lr) * 8);
  // eigenvalues lamT[pl][z = 16 w + q + 4 r][8 w' .. 8 w' + 7] of this lane (buffer loads: plane base in SGPRs, one 32-bit lane offset, the
  // row as a scalar offset); requested BEFORE the first DMA, so that every counted vmcnt wait below has them behind it
  const lrsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.lamT + pl * (int64_t)(N * PX)), 0, N * PX * 8, 0x00020000);
  v2d lam[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      lam[r][k] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(lrs, lofs, 4 * r * PX * 8 + 16 * k, 0));
  __syncthreads();
#pragma unroll
  for (int c = 0; c < RING - 1; ++c) stage(cur, c, c);
  int slot0 = 0, it = 0;
  const int col = 16 * w + lr;                         // this lane's z (row i of the transposed product)
  const bool w0 = lr == 0;                             // the lane column of frequency 0
  const double r2 = 1.4142135623730951;
  // A fragments: X[x = 16 c + 4 q + rho][z = col] (the swizzle of row 4 q + rho depends on q alone: the classes are immediates);
  // B fragments: img[rho][j = 4 c + q][w = lr]
  const unsigned aoff = ring_lds + w * RING * (CHB / NW) + q * 128 + lr * 8;
  const unsigned boff = (unsigned)(uintptr_t)(lds_ptr_t)img + q * 256 + lr * 16;
  for (int64_t row = rfirst; row < g.rows; row += rstep, ++it) {
    const int64_t rn = row + rstep < g.rows ? row + rstep : row;
    const double* nxt = plane_ptr(rn);
    v4d ac[4], as[4];                                  // [class]: cosine-row sums C_rho, sine-row sums S_rho; registers r <-> z = 16 w + q + 4 r
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) ac[rho] = as[rho] = (v4d){0., 0., 0., 0.};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk c has landed when at most the newer operations are in flight: RING-2 chunks, plus -- for the chunks that were
      // requested during the previous plane -- the NL eigenvalue loads above (conservative for the <= 1 result store)
      __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * ND));       // (conservative for the <= 1 result store of the previous plane)
      {
        // (slot of chunk c - 1: this wave's own reads of it completed before its MFMAs were issued)
        const int cn = c + RING - 1;
        if (cn < NCH) stage(cur, cn, (slot0 + cn) % RING);
        else stage(nxt, cn - NCH, (slot0 + cn) % RING);
      }
      const unsigned so = ((slot0 + c) % RING) * (CHB / NW);
      double a[4];
      v2d f[4];
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[rho]) : "v"(aoff + so), "n"(rho * 512));
      const unsigned bo = boff + c * 1024;
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[rho]) : "v"(bo), "n"(rho * 4096));
      // (LDS reads return in order: class rho may start once its image pair is there)
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(f[0]));
      ac[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], f[0].x, ac[0], 0, 0, 0);
      as[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], f[0].y, as[0], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[1]));
      ac[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], f[1].x, ac[1], 0, 0, 0);
      as[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], f[1].y, as[1], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[2]));
      ac[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], f[2].x, ac[2], 0, 0, 0);
      as[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], f[2].y, as[2], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[3]));
      ac[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], f[3].x, ac[3], 0, 0, 0);
      as[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], f[3].y, as[3], 0, 0, 0);
    }
    // ---- the eight outputs of frequency w = lr per z, scaled by the eigenvalues and summed over z: registers, lane groups, waves ------
    v2d o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (v2d){0., 0.};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double u0 = ac[0][r] + ac[2][r], u1 = ac[0][r] - ac[2][r], v0 = ac[1][r] + ac[3][r], v1 = ac[1][r] - ac[3][r];
      const double ws = as[0][r] + as[2][r], wd = as[0][r] - as[2][r], z0 = as[1][r] + as[3][r], z1 = as[1][r] - as[3][r];
      // positions (cos w, mirror | sin w, mirror | cos(n/2 - w), mirror | sin(n/2 - w), mirror); frequency 0: (0, n | middle pair |
      // cos n/4, mirror | sin n/4, mirror) from the constant-row sums C and the alternating-row sums S
      const double a0 = r2 * as[0][r], a2 = r2 * as[2][r];
      o[0] += (v2d){u0 + v0, u0 - v0} * lam[r][0];
      o[1] += (w0 ? (v2d){u1 + v1, u1 - v1} : (v2d){ws + z0, ws - z0}) * lam[r][1];
      o[2] += (w0 ? (v2d){a0 + z1, a0 - z1} : (v2d){u1 + z1, u1 - z1}) * lam[r][2];
      o[3] += (w0 ? (v2d){a2 + z0, a2 - z0} : (v2d){v1 - wd, -wd - v1}) * lam[r][3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) { o[k][e] += __shfl_xor(o[k][e], 16); o[k][e] += __shfl_xor(o[k][e], 32); }
    if (q == 0) {
      v2d* const rp = reinterpret_cast<v2d*>(red + (it & 1) * (NW * PX) + w * PX + 8 * lr);
#pragma unroll
      for (int k = 0; k < 4; ++k) rp[k] = o[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < PX) {
      const double* r4 = red + (it & 1) * (NW * PX) + tid;
      g.out[row * g.out_row + pl * g.out_plane + tid] = (r4[0] + r4[PX]) + (r4[2 * PX] + r4[3 * PX]);
    }
    slot0 = (slot0 + NCH) % RING;
    cur = nxt;
  }
}

template <int N, bool QUAD = false>
int launch_fwd(const FoldArgs& g, hipStream_t st) {
  using K = FwdCfg<N, 4>;
  auto kern = xz_fold_fwd_kernel<N, QUAD>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), K::LDS)) return rc;
  const int64_t nwg = g.nplanes < 2048 ? g.nplanes : 2048;   // persistent, two workgroups per CU resident
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), K::LDS, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

template <int N, int MODE, bool QUAD = false>
int launch_inv(const FoldArgs& g, hipStream_t st) {
  using K = InvCfg<N, 3>;
  constexpr bool RED = MODE == 1;
  auto kern = xz_fold_inv_kernel<N, MODE, QUAD>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), K::LDS)) return rc;
  int64_t nwg = g.nplanes < 2048 ? g.nplanes : 2048;
  if (RED) nwg = nwg / g.ppr * g.ppr;                          // every workgroup keeps one y: grid = whole rows of planes
  if (nwg <= 0) return GEOBO_E_ARG;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), K::LDS, st, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

// dense planes: the right half of a row and the bottom rows follow in place
void dense_fwd(FoldArgs& g, int n) { g.in_rowB = n * 8; g.in_halfB = n * 4; g.in_botB = 0; g.out_rowB = 2 * n * 8; g.out_halfB = n * 8; g.out_botB = 0; }
void dense_inv(FoldArgs& g, int n) { g.in_rowB = 2 * n * 8; g.in_halfB = n * 8; g.in_botB = 0; g.out_rowB = 0; g.out_halfB = n * 4; g.out_botB = 0; }

}  // namespace

extern "C" int geobo_xcorr_reduce_fold(int n, int64_t rows, int planes, const double* in, int64_t in_row, int64_t in_plane,
                                       const double* F, const double* lamT, double* out, int64_t out_row, int64_t out_plane,
                                       void* stream) {
  if (!in || !out || !F || !lamT) return GEOBO_E_ARG;
  if (rows <= 0 || planes <= 0) return GEOBO_OK;
  if ((in_row & 1) || (in_plane & 1) || ((uintptr_t)in & 15) || ((uintptr_t)F & 15) || ((uintptr_t)lamT & 15)) return GEOBO_E_ALIGN;
  if (n != 64) return GEOBO_E_UNSUPPORTED;
  XCFArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.F = F; g.lamT = lamT;
  g.out = out; g.out_row = out_row; g.out_plane = out_plane; g.rows = rows; g.nplanes = rows * planes;
  constexpr int N = 64;
  constexpr size_t lds = (size_t)4 * 16 * N * 8 + (size_t)4 * 16 * 16 * 16 + 2 * 4 * (2 * N) * 8;
  auto kern = xcorr_fold4_kernel<N>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds)) return rc;
  if (planes > 2048) return GEOBO_E_UNSUPPORTED;
  const int64_t rper = 2048 / planes;                       // persistent: every workgroup keeps one y-mode (its eigenvalues in registers)
  const int64_t nwg = (int64_t)planes * (rows < rper ? rows : rper);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, (hipStream_t)stream, g);
  return hipGetLastError() == hipSuccess ? GEOBO_OK : GEOBO_E_LAUNCH;
}

extern "C" int geobo_xz2d_fold(int inverse, int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row,
                               int64_t in_plane, const double* Fx, const double* Fz, double* out, int64_t out_row,
                               int64_t out_plane, void* stream) {
  if (!in || !out || !Fx || !Fz) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if ((in_row & 1) || (in_plane & 1) || (out_row & 1) || (out_plane & 1) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) ||
      ((uintptr_t)Fx & 15) || ((uintptr_t)Fz & 15))
    return GEOBO_E_ALIGN;
  if (n != 64) return GEOBO_E_UNSUPPORTED;
  FoldArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.out = out; g.out_row = out_row; g.out_plane = out_plane;
  g.Fz = Fz; g.Fx = Fx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row; g.out_rs = n;
  g.row_off = nullptr; g.edge = nullptr; g.edge_row = 0;
  g.in2 = nullptr; g.in2_row = 0; g.r2_first = 0; g.ss = nullptr;
  if (inverse) dense_inv(g, 64); else dense_fwd(g, 64);
  return inverse ? launch_inv<64, 0>(g, (hipStream_t)stream) : launch_fwd<64>(g, (hipStream_t)stream);
}

extern "C" int geobo_xz2d_fold_quad(int inverse, int n, int64_t rows, int groups_per_row, const double* in, int64_t in_row, int64_t in_plane,
                                    const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane, void* stream) {
  if (!in || !out || !Fx || !Fz) return GEOBO_E_ARG;
  if (rows <= 0 || groups_per_row <= 0) return GEOBO_OK;
  if ((in_row & 1) || (in_plane & 1) || (out_row & 1) || (out_plane & 1) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) ||
      ((uintptr_t)Fx & 15) || ((uintptr_t)Fz & 15))
    return GEOBO_E_ALIGN;
  if (n != 32 || in_plane * 8 >= (1ll << 29) || out_plane * 8 >= (1ll << 29)) return GEOBO_E_UNSUPPORTED;
  // a kernel plane = the planes y .. y+3 of a group as [[y, y+1], [y+2, y+3]]; in_plane / out_plane: stride of ONE small plane
  const int N = 64, h = N / 2;
  FoldArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = 4 * in_plane; g.out = out; g.out_row = out_row; g.out_plane = 4 * out_plane;
  g.Fz = Fz; g.Fx = Fx; g.ppr = groups_per_row; g.nplanes = rows * groups_per_row;
  g.row_off = nullptr; g.edge = nullptr; g.edge_row = 0;
  g.in2 = nullptr; g.in2_row = 0; g.r2_first = 0; g.ss = nullptr;
  if (!inverse) {       // four h x h planes in, four N x N planes out
    g.out_rs = N;
    g.in_rowB = h * 8; g.in_halfB = (int)(in_plane * 8); g.in_botB = (int)(2 * in_plane * 8) - h * h * 8;
    g.out_rowB = N * 8; g.out_halfB = (int)(out_plane * 8); g.out_botB = (int)(2 * out_plane * 8) - N * N * 8;
    return launch_fwd<64, true>(g, (hipStream_t)stream);
  }
  g.out_rs = h;           // four N x N planes in, four h x h planes out
  g.in_rowB = N * 8; g.in_halfB = (int)(in_plane * 8); g.in_botB = (int)(2 * in_plane * 8) - N * N * 8;
  g.out_rowB = 0; g.out_halfB = (int)(out_plane * 8); g.out_botB = (int)(2 * out_plane * 8) - h * h * 8;
  return launch_inv<64, 0, true>(g, (hipStream_t)stream);
}

extern "C" int geobo_xz2d_fold_lattice(int n, int64_t rows, int planes_per_row, const double* Q, const int64_t* row_off,
                                       int64_t q_plane, const double* edge, int64_t edge_row, const double* Fx, const double* Fz,
                                       double* out, int64_t out_row, int64_t out_plane, void* stream) {
  if (!Q || !row_off || !edge || !out || !Fx || !Fz) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if (planes_per_row < 3) return GEOBO_E_ARG;
  if ((q_plane & 1) || (edge_row & 1) || (out_row & 1) || (out_plane & 1) || ((uintptr_t)Q & 15) || ((uintptr_t)edge & 15) ||
      ((uintptr_t)out & 15) || ((uintptr_t)Fx & 15) || ((uintptr_t)Fz & 15))
    return GEOBO_E_ALIGN;
  if (n != 64) return GEOBO_E_UNSUPPORTED;
  FoldArgs g;
  g.in = Q; g.in_row = 0; g.in_plane = q_plane; g.out = out; g.out_row = out_row; g.out_plane = out_plane;
  g.Fz = Fz; g.Fx = Fx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row; g.out_rs = n;
  g.row_off = row_off; g.edge = edge; g.edge_row = edge_row;
  g.in2 = nullptr; g.in2_row = 0; g.r2_first = 0; g.ss = nullptr;
  dense_fwd(g, 64);
  return launch_fwd<64>(g, (hipStream_t)stream);
}

extern "C" int geobo_xz2d_fold_inv_ss_slots(int n, int64_t rows, int planes_per_row) {
  if (n != 64 || rows <= 0 || planes_per_row <= 0) return 0;
  const int64_t np = rows * planes_per_row;
  const int64_t nwg = (np < 2048 ? np : 2048) / planes_per_row * planes_per_row;
  return (int)(nwg / planes_per_row);
}

extern "C" int geobo_xz2d_fold_inv_ss(int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row, int64_t in_plane,
                                      const double* in2, int64_t in2_row, int64_t r2_first, const double* Fx, const double* Fz,
                                      double* ss, void* stream) {
  if (!in || !Fx || !Fz || !ss) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if (in2 && (r2_first < 0 || r2_first > rows)) return GEOBO_E_ARG;
  if ((in_row & 1) || (in_plane & 1) || (in2_row & 1) || ((uintptr_t)in & 15) || ((uintptr_t)in2 & 15) || ((uintptr_t)Fx & 15) ||
      ((uintptr_t)Fz & 15))
    return GEOBO_E_ALIGN;
  if (n != 64 || planes_per_row > 2048) return GEOBO_E_UNSUPPORTED;
  FoldArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.out = nullptr; g.out_row = 0; g.out_plane = 0; g.out_rs = n;
  g.Fz = Fz; g.Fx = Fx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row;
  g.row_off = nullptr; g.edge = nullptr; g.edge_row = 0;
  g.in2 = in2; g.in2_row = in2_row; g.r2_first = in2 ? r2_first : rows; g.ss = ss;
  dense_inv(g, 64);
  return launch_inv<64, 1>(g, (hipStream_t)stream);
}

extern "C" int geobo_xz2d_fold_inv_strided(int n, int64_t rows, int planes_per_row, const double* in, int64_t in_row, int64_t in_plane,
                                           const double* Fx, const double* Fz, double* out, int64_t out_row, int64_t out_plane,
                                           int64_t out_rowstride, void* stream) {
  if (!in || !out || !Fx || !Fz) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if (out_rowstride < n) return GEOBO_E_ARG;
  if ((in_row & 1) || (in_plane & 1) || ((uintptr_t)in & 15) || ((uintptr_t)Fx & 15) || ((uintptr_t)Fz & 15)) return GEOBO_E_ALIGN;
  if (n != 64) return GEOBO_E_UNSUPPORTED;
  FoldArgs g;
  g.in = in; g.in_row = in_row; g.in_plane = in_plane; g.out = out; g.out_row = out_row; g.out_plane = out_plane; g.out_rs = out_rowstride;
  g.Fz = Fz; g.Fx = Fx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row;
  g.row_off = nullptr; g.edge = nullptr; g.edge_row = 0;
  g.in2 = nullptr; g.in2_row = 0; g.r2_first = 0; g.ss = nullptr;
  dense_inv(g, 64);
  return launch_inv<64, 0>(g, (hipStream_t)stream);
}

extern "C" int geobo_xz2d_fold_inv_mul(int n, int64_t rows, int planes_per_row, const double* a, int64_t a_plane, const double* b,
                                       int64_t b_row, const double* Fx, const double* Fz, double* out, int64_t out_row,
                                       int64_t out_plane, int64_t out_rowstride, void* stream) {
  if (!a || !b || !out || !Fx || !Fz) return GEOBO_E_ARG;
  if (rows <= 0 || planes_per_row <= 0) return GEOBO_OK;
  if (out_rowstride < n) return GEOBO_E_ARG;
  if ((a_plane & 1) || (b_row & 1) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)Fx & 15) || ((uintptr_t)Fz & 15))
    return GEOBO_E_ALIGN;
  if (n != 64) return GEOBO_E_UNSUPPORTED;
  FoldArgs g;
  g.in = a; g.in_row = 0; g.in_plane = a_plane; g.out = out; g.out_row = out_row; g.out_plane = out_plane; g.out_rs = out_rowstride;
  g.Fz = Fz; g.Fx = Fx; g.ppr = planes_per_row; g.nplanes = rows * planes_per_row;
  g.row_off = nullptr; g.edge = nullptr; g.edge_row = 0;
  g.in2 = b; g.in2_row = b_row; g.r2_first = 0; g.ss = nullptr;
  dense_inv(g, 64);
  return launch_inv<64, 2>(g, (hipStream_t)stream);
}
