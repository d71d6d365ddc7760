// Stand-alone probe, not part of the library: what a stream reaches on this device when every workgroup touches runs of RUN
// doubles, one run per (row, y), a "plane stride" apart -- the access pattern of toeplitz_y / ymul / the transform kernels, whose
// work buffers are [row][y][Px*Pz = 16384 doubles] (plane stride 128 KiB, a power of two).
//     hipcc --offload-arch=gfx950 -O3 tools/hbm_copy_runs.hip -o /tmp/hbm_copy_runs && /tmp/hbm_copy_runs
// Round 2 measured only the power-of-two stride with 8-byte lane accesses (4.4 - 5.0 TB/s read + write).  Round 3 (VERDICT r02,
// item 3) adds: padded plane strides (+32 / +256 doubles), 16-byte lane accesses, read-only and write-only streams, and a plain
// contiguous copy as the control.  Output is committed as profiles/r03_hbm_copy_runs.txt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0: copy (read + write), 1: read only (the sum is stored once per thread), 2: write only, 3: copy with non-temporal accesses.  T = double (8 B) or v2d (16 B).
// Every thread owns one column position of the run and walks y inside a row, rows strided over gridDim.y.
template <class T, int MODE>
__global__ void runk(const T* in, T* out, long CS, long C, int NY, long R) {   // CS = plane stride, C = valid width (in units of T)
  const long c0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c0 >= C) return;
  T acc = {};
  for (long r = blockIdx.y; r < R; r += gridDim.y)
    for (int y = 0; y < NY; ++y) {
      const long o = (r * NY + y) * CS + c0;
      if (MODE == 0) out[o] = in[o] * 1.0000001;
      if (MODE == 1) acc += in[o];
      if (MODE == 2) out[o] = acc + (double)y;
      if (MODE == 3) __builtin_nontemporal_store(__builtin_nontemporal_load(in + o) * 1.0000001, out + o);
    }
  if (MODE == 1) out[c0 + (long)blockIdx.y * CS] = acc;
}

// Round 5: the access pattern a per-mode MFMA y stage would need -- a workgroup owns MC consecutive modes
// (MC * 8 bytes: 64- or 128-byte segments), all NY planes, a share of the rows; 16-byte lanes, MC / 2 lanes per segment, a 256-thread
// workgroup covers 512 / MC (row, y) segments per pass.  MODE 0 copy, 1 read only, 4 read once + write twice (two property blocks).
template <int MC, int MODE>
__global__ void __launch_bounds__(256) segk(const v2d* in, v2d* out, v2d* out2, long CS2, int NY, long R) {   // CS2 = plane stride in v2d
  constexpr int LPS = MC / 2, SPP = 256 / LPS;                 // lanes per segment, segments per pass
  const int l = threadIdx.x % LPS, slot = threadIdx.x / LPS;
  const long c = (long)blockIdx.x * LPS + l;
  v2d acc = {0., 0.};
  for (long r = blockIdx.y; r < R; r += gridDim.y)
    for (int y0 = 0; y0 < NY; y0 += SPP) {
      const int y = y0 + slot;
      if (y >= NY) continue;
      const long o = (r * NY + y) * CS2 + c;
      const v2d v = in[o];
      if (MODE == 0 || MODE == 4) out[o] = v * 1.0000001;
      if (MODE == 4) out2[o] = v * 0.9999999;
      if (MODE == 1) acc += v;
    }
  if (MODE == 1) out[(long)blockIdx.y * CS2 + c] = acc;
}

template <int MC, int MODE>
static void runseg(const char* what, double* a, double* b, double* b2, long Cd, long padd, int NY, long R, int gy) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((segk<MC, MODE>), dim3((unsigned)(Cd / MC), gy), dim3(256), 0, 0, (const v2d*)a, (v2d*)b, (v2d*)b2, (Cd + padd) / 2, NY, R);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (MODE == 0 ? 2.0 : MODE == 4 ? 3.0 : 1.0) * (double)R * NY * Cd * 8;
  printf("%-12s %3d-byte segments (%2d modes per workgroup)  plane stride %6ld+%-4ld doubles  gridy %3d : %7.3f ms  %5.2f TB/s\n", what, MC * 8, MC, Cd,
         padd, gy, best, bytes / best / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <class T>
__global__ void lineark(const T* in, T* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i] * 1.0000001;
}

template <class T, int MODE>
static void run(const char* what, double* a, double* b, long Cd, long padd, int NY, long R, int run_threads, int gy) {
  const int W = sizeof(T) / 8;
  const long C = Cd / W, CS = (Cd + padd) / W;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((runk<T, MODE>), dim3((unsigned)((C + run_threads - 1) / run_threads), gy), dim3(run_threads), 0, 0,
                       (const T*)a, (T*)b, CS, C, NY, R);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (MODE == 0 || MODE == 3 ? 2.0 : 1.0) * (double)R * NY * Cd * 8;
  printf("%-10s lane %2zu B  plane stride %6ld+%-4ld doubles  run %5d B  gridy %3d : %7.3f ms  %5.2f TB/s\n", what, sizeof(T), Cd, padd,
         run_threads * (int)sizeof(T), gy, best, bytes / best / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  const long C = 16384, R = argc > 1 ? atol(argv[1]) : 128; const int NY = 64;     // R = 128: 2 x 1.07 GB per launch; 1024: 2 x 8.6 GB
  printf("R = %ld rows x NY = %d planes x C = %ld doubles: %.2f GB per direction\n", R, NY, C, (double)R * NY * C * 8 / 1e9);
  const size_t n = (size_t)R * NY * (C + 256) + 4096;
  double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  {   // control: contiguous grid-stride copy of the same volume
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long nn = (long)R * NY * C;
    for (int w = 0; w < 2; ++w) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (w == 0) hipLaunchKernelGGL(lineark<double>, dim3(8192), dim3(256), 0, 0, a, b, nn);
        else hipLaunchKernelGGL(lineark<v2d>, dim3(8192), dim3(256), 0, 0, (const v2d*)a, (v2d*)b, nn / 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("contiguous copy, lane %2d B: %7.3f ms  %5.2f TB/s (read + write)\n", w ? 16 : 8, best, 2.0 * nn * 8 / best / 1e9);
    }
  }
  {   // round 5: per-mode-group segments (the MFMA y stage's pattern); second output buffer for the two-block form
    double* b2; hipMalloc(&b2, n * 8); hipMemset(b2, 0, n * 8);
    printf("---- segments of MC modes per workgroup, 16-byte lanes ----\n");
    for (int gy = 8; gy <= 32; gy *= 2) {
      runseg<16, 0>("seg copy", a, b, b2, C, 0, NY, R, gy);
      runseg<8, 0>("seg copy", a, b, b2, C, 0, NY, R, gy);
    }
    runseg<32, 0>("seg copy", a, b, b2, C, 0, NY, R, 16);
    runseg<16, 0>("seg copy", a, b, b2, C, 256, NY, R, 16);
    runseg<8, 0>("seg copy", a, b, b2, C, 256, NY, R, 16);
    runseg<16, 1>("seg read", a, b, b2, C, 0, NY, R, 16);
    runseg<8, 1>("seg read", a, b, b2, C, 0, NY, R, 16);
    runseg<16, 4>("seg 1r + 2w", a, b, b2, C, 0, NY, R, 16);
    runseg<8, 4>("seg 1r + 2w", a, b, b2, C, 0, NY, R, 16);
    runseg<16, 4>("seg 1r + 2w", a, b, b2, C, 256, NY, R, 16);
    hipFree(b2);
  }
  const long pads[3] = {0, 32, 256};
  for (int pi = 0; pi < 3; ++pi) {
    const long pad = pads[pi];
    printf("---- plane stride %ld + %ld doubles (%ld B) ----\n", C, pad, (C + pad) * 8);
    run<double, 0>("copy", a, b, C, pad, NY, R, 64, 32);
    run<double, 0>("copy", a, b, C, pad, NY, R, 256, 128);
    run<double, 0>("copy", a, b, C, pad, NY, R, 1024, 128);
    run<v2d, 0>("copy", a, b, C, pad, NY, R, 64, 64);
    run<v2d, 0>("copy", a, b, C, pad, NY, R, 256, 128);
    run<v2d, 0>("copy", a, b, C, pad, NY, R, 512, 128);
    run<double, 3>("copy-nt", a, b, C, pad, NY, R, 256, 128);
    run<v2d, 3>("copy-nt", a, b, C, pad, NY, R, 256, 128);
    run<double, 1>("read", a, b, C, pad, NY, R, 256, 128);
    run<v2d, 1>("read", a, b, C, pad, NY, R, 256, 128);
    run<double, 2>("write", a, b, C, pad, NY, R, 256, 128);
    run<v2d, 2>("write", a, b, C, pad, NY, R, 256, 128);
  }
  return 0;
}
