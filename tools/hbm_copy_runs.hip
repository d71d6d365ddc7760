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
