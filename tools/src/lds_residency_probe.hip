// How many 64-thread workgroups with a given dynamic-LDS size are resident on the whole GPU at once (MI355X: 256 CUs x 160 KB)?
// Every workgroup bumps a counter, spins 200 us, records the largest value it has seen, and leaves: max = resident workgroups.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/lds_residency_probe tools/src/lds_residency_probe.hip     run (GPU box): tools/lds_residency_probe bytes...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ double smem[];
__global__ void __launch_bounds__(64) probe(int* active, int* peak, int bytes) {
  if (threadIdx.x == 0) {
    smem[bytes / 8 - 1] = 1.0;   // touch the allocation
    int now = atomicAdd(active, 1) + 1;
    atomicMax(peak, now);
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 400000ull) { atomicMax(peak, __hip_atomic_load(active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); __builtin_amdgcn_s_sleep(64); }
    atomicAdd(active, -1);
  }
}
int main(int argc, char** argv) {
  int *d;
  hipMalloc(&d, 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, %zu B LDS per workgroup max\n", p.name, p.multiProcessorCount, (size_t)p.sharedMemPerBlock);
  for (int a = 1; a < argc; a++) {
    int bytes = atoi(argv[a]);
    hipMemset(d, 0, 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(probe, dim3(256 * 24), dim3(64), bytes, 0, d, d + 1, bytes);
    hipError_t e = hipDeviceSynchronize();
    int h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)probe, 64, bytes);
    printf("%6d B: %5d resident workgroups = %.2f per CU   (occupancy API: %d per CU)%s\n", bytes, h[1], h[1] / (double)p.multiProcessorCount, occ, e == hipSuccess ? "" : "  LAUNCH FAILED");
  }
  return 0;
}
