// Probe: does a wave64 VALU instruction on gfx950 cost less when only part of the lanes is enabled (EXEC = low 32 / low 16 lanes)?
// The moist column kernel is one wavefront per SIMD on a 20 000-instruction stream (HISTORY.md 11); if passes over disabled 16-lane groups
// were skipped, a wavefront of 32 columns (two per SIMD) would cost half.  Measured: chains of dependent fp64 fma / div / exp / log with
// 64, 32 and 16 active lanes, one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int OP>
__global__ void k(double *out, int iters, int active) {
  const int lane = threadIdx.x & 63;
  double x = 1.0 + 1e-3 * lane, y = 0.5;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
      if (OP == 0) { x = __builtin_fma(x, 0.999999, y); y = __builtin_fma(y, 1.000001, -1e-7 * x); }
      else if (OP == 1) { x = 1.0 / (x + 1.5); y = y / (x + 2.0); }
      else if (OP == 2) { x = exp(-x * 0.5) + 0.5; }
      else { x = log(x + 2.0); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
template <int OP> static void run(const char *name, double *d) {
  for (int active : {64, 32, 16}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 100, active);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 20000, active);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s %2d active lanes: %.3f ms for 20000 iterations (%.1f ns per iteration)\n", name, active, ms, 1e6 * ms / 20000);
  }
}
int main() {
  double *d; (void)hipMalloc((void **)&d, 256 * 256 * 8);
  run<0>("fma x2", d); run<1>("div x2", d); run<2>("exp", d); run<3>("log", d);
  return 0;
}
