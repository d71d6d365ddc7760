// Which SIMD does wave w of a 512-thread (and a 256-thread) workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
// The two-size wave layout of toeplitz_y2s_kernel assumes wave w -> SIMD w % 4.      hipcc --offload-arch=gfx950 -O2 probe_simd.hip -o probe_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}
int main() {
  for (int threads : {512, 256}) {
    const int nb = 600, nw = threads / 64;
    unsigned* d; hipMalloc(&d, nb * nw * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, d);
    unsigned h[600 * 8]; hipMemcpy(h, d, nb * nw * 4, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < nb; ++b) {
      bool good = true;
      for (int w = 0; w < nw; ++w) good &= ((h[b * nw + w] >> 4) & 3) == (unsigned)((w + ((h[b * nw] >> 4) & 3)) & 3);
      ok += good;
    }
    printf("%d threads: %d of %d workgroups have wave w on SIMD (s0 + w) %% 4\n", threads, ok, nb);
    for (int b : {0, 1, 257, 599}) {
      printf("  block %3d: simd of waves:", b);
      for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
      printf("   cu %u\n", (h[b * nw] >> 8) & 15);
    }
    hipFree(d);
  }
  return 0;
}
