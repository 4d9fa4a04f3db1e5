// Probe: bandwidth of a column-kernel-like access pattern as a function of the level stride.
// Block = 64 columns x 8 wavefronts, wavefront w reads levels [5w, 5w+5) of NA arrays (one 512-byte run per
// level and array) and writes NO arrays.  Usage: stride_probe  (prints GB/s for several strides)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NA, int NO>
__global__ __launch_bounds__(512) void probe(const double *__restrict__ in, double *__restrict__ out, size_t ls, size_t as, int L, int CH) {
  const int tid = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t c2 = (size_t)blockIdx.x * 64 + tid;
  double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = w * CH + i;
    if (k < L) {
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[i] += in[a * as + k * ls + c2];
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = w * CH + i;
    if (k < L) {
#pragma unroll
      for (int o = 0; o < NO; ++o) out[o * as + k * ls + c2] = acc[i] + o;
    }
  }
}

int main() {
  const int L = 40, ncol = 256 * 128, NA = 10, NO = 6, NSET = 6;   // NSET buffer sets (1 GB) so the 256 MB MALL cannot hold them
  const size_t pads[] = {0, 64, 520};
  const size_t ldss[] = {0, 90 * 1024};                            // dynamic LDS: 90 KB forces one block per CU
  for (size_t lds : ldss)
  for (size_t pad : pads) {
    const size_t ls = ncol + pad, as = ls * L + 7 * pad;
    double *in, *out;
    CK(hipMalloc(&in, as * NA * NSET * sizeof(double)));
    CK(hipMalloc(&out, as * NO * NSET * sizeof(double)));
    CK(hipMemset(in, 0, as * NA * NSET * sizeof(double)));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 6; ++it) hipLaunchKernelGGL((probe<NA, NO>), dim3(ncol / 64), dim3(512), lds, 0, in + (it % NSET) * as * NA, out + (it % NSET) * as * NO, ls, as, L, 5);
    (void)hipEventRecord(e0);
    const int reps = 60;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<NA, NO>), dim3(ncol / 64), dim3(512), lds, 0, in + (it % NSET) * as * NA, out + (it % NSET) * as * NO, ls, as, L, 5);
    (void)hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ncol * L * 8 * (NA + NO);
    printf("lds %6zu pad %5zu doubles: %.2f us  %.2f TB/s\n", lds, pad, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    (void)hipFree(in); (void)hipFree(out);
  }
  return 0;
}
