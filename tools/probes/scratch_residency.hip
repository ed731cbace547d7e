// Does a kernel's private segment (scratch) limit how many of its wavefronts are resident at once on the MI355X?
// Two kernels with the same long ALU loop, 64-thread workgroups, 20 KB of LDS each (8 per CU): one keeps a 1 KB/lane
// run-time-indexed array (scratch), the other does not. Time vs grid size shows the resident-wave limit of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCR>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) probe(double* out, int iters, int idx) {
  extern __shared__ double lds[];
  double a[SCR > 0 ? SCR : 1];
  if (SCR > 0) for (int i = 0; i < SCR; i++) a[i] = i * 1e-3;
  double x = threadIdx.x * 1e-3, y = 1.0;
  lds[threadIdx.x] = x;
  for (int i = 0; i < iters; i++) {
    x = fma(x, 0.999999, y * 1e-6);
    y = fma(y, 0.999999, x * 1e-6);
    if (SCR > 0 && (i & 1023) == 0) { a[(idx + i) & (SCR - 1)] += x; x += a[(idx + 2 * i) & (SCR - 1)]; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x + y + lds[63 - threadIdx.x];
}
template <int SCR> void sweep(const char* name, double* d_out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int n : {256, 1024, 2048, 4096, 8192}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<SCR>, dim3(n), dim3(64), 20480, 0, d_out, iters, rep);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s grid %5d: %.2f ms\n", name, n, best);
  }
}
int main() {
  double* d_out; hipMalloc(&d_out, 8192 * 64 * 8);
  sweep<0>("no-scratch", d_out, 2000000);
  sweep<128>("scratch-1KB", d_out, 2000000);
  sweep<512>("scratch-4KB", d_out, 2000000);
  return 0;
}
