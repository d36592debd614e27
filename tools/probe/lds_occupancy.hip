// lds_occupancy.hip — how many one-wavefront workgroups of a given LDS size does a gfx950 CU hold?
// (the LDS allocation granule decides which Work-struct sizes buy another resident env: DESIGN.md §4)
// build: hipcc --offload-arch=gfx950 -O2 -o lds_occupancy lds_occupancy.hip ; run: ./lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int REGS>
__global__ __launch_bounds__(64) void spin(long long* t, int spin_ticks) {
  extern __shared__ char sm[];
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) sm[0] = 1;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime(); }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s  CUs %d  sharedMemPerMultiprocessor %zu  maxSharedMemoryPerBlock %zu\n", p.name, cus, (size_t)p.sharedMemPerMultiprocessor, (size_t)p.sharedMemPerBlock);
  const int nb = cus * 20;
  long long* d; hipMalloc(&d, nb * 16);
  std::vector<long long> h(2 * nb);
  for (int lds = 10240; lds <= 20480; lds += 256) {
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin<0>, 64, lds);
    hipMemset(d, 0, nb * 16);
    hipLaunchKernelGGL(spin<0>, dim3(nb), dim3(64), lds, 0, d, 20000);   // 200 us at 100 MHz
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    // resident at the time the first block is half way through
    long long tmin = h[0]; for (int i = 0; i < nb; i++) tmin = std::min(tmin, h[2 * i]);
    const long long probe = tmin + 10000;
    int live = 0; for (int i = 0; i < nb; i++) if (h[2 * i] <= probe && h[2 * i + 1] > probe) live++;
    printf("lds %6d B  api %2d blocks/CU  measured %5.2f blocks/CU (%d resident)\n", lds, api, (double)live / cus, live);
  }
  return 0;
}
