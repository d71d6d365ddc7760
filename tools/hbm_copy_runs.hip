// Stand-alone probe, not part of the library: what a read + write stream reaches on this device when every workgroup touches
// runs of RUN doubles, one run per (row, y), 128 KiB apart -- the access pattern of toeplitz_y / ymul / the transform kernels.
//     hipcc --offload-arch=gfx950 -O3 tools/hbm_copy_runs.hip -o /tmp/hbm_copy_runs && /tmp/hbm_copy_runs
// MI355X, 2 x 1.07 GB per launch: 4.4 - 5.0 TB/s for runs of 512 B to 8 KiB (run length does not matter; too few workgroups do:
// 2.9 TB/s with 8 row groups).  That -- not the 8 TB/s of the data sheet -- is the ceiling the HBM-bound kernels are held against
// in DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int RUN>   // doubles per run; blockDim = RUN threads... each thread 8 B
__global__ void copyk(const double* in, double* out, long C, int NY, long R) {
  const long c0 = (long)blockIdx.x * RUN;
  for (long r = blockIdx.y; r < R; r += gridDim.y)
    for (int y = 0; y < NY; ++y) {
      const long o = (r * NY + y) * C + c0 + threadIdx.x;
      out[o] = in[o] * 1.0000001;
    }
}
int main() {
  const long C = 16384, R = 128; const int NY = 64;
  size_t n = (size_t)R * NY * C;
  double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
#define RUNK(RUN, GY) { hipEventRecord(e0); hipLaunchKernelGGL(copyk<RUN>, dim3(C / RUN, GY), dim3(RUN), 0, 0, a, b, C, NY, R); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("run %4d B x gridy %3d: %.3f ms  %.2f TB/s\n", RUN * 8, GY, ms, 2.0 * n * 8 / ms / 1e9); }
    RUNK(64, 8) RUNK(64, 32) RUNK(128, 16) RUNK(128, 64) RUNK(256, 32) RUNK(256, 128) RUNK(512, 64) RUNK(512, 128) RUNK(1024, 128)
  }
  return 0;
}
