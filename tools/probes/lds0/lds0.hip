// probe: a scene image at LDS address 0 reached without the `extern __shared__` symbol (no dynlds table lookup in called functions) -- does a launch with dynamic LDS still get its LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) double smem[];
typedef __attribute__((address_space(3))) double lds_double;
struct L { double a[2048]; int n; double b[2048]; };
#ifdef CONSTBASE
#define SP ((L*)(lds_double*)(unsigned long)0)
#else
#define SP (reinterpret_cast<L*>(smem))
#endif
__device__ __noinline__ void callee(int k) {
  for (int i = threadIdx.x; i < SP->n; i += blockDim.x) SP->b[i] = SP->a[i] * k + blockIdx.x;
}
__global__ void kern(double* out, int k, unsigned* base) {
#ifndef NOSYM
  if (threadIdx.x == 0 && blockIdx.x == 0) *base = (unsigned)(unsigned long)(lds_double*)smem;
#endif
  if (threadIdx.x == 0) SP->n = 2048;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) SP->a[i] = i;
  __syncthreads();
  callee(k);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[blockIdx.x * 2048 + i] = SP->b[i];
}
int main() {
  const int nb = 1024; double* d; unsigned* db; hipMalloc(&d, nb * 2048 * 8); hipMalloc(&db, 4);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(64), sizeof(L), 0, d, 3, db);
  hipError_t e = hipDeviceSynchronize();
  static double h[1024 * 2048]; unsigned hb = 77; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hb, db, 4, hipMemcpyDeviceToHost);
  long bad = 0; for (int b = 0; b < nb; b++) for (int i = 0; i < 2048; i++) if (h[b * 2048 + i] != i * 3.0 + b) bad++;
  printf("%s: sync %s, dynamic LDS base %u, wrong values %ld of %d\n",
#ifdef CONSTBASE
         "constant base",
#else
         "symbol",
#endif
         hipGetErrorString(e), hb, bad, nb * 2048);
  return 0;
}
