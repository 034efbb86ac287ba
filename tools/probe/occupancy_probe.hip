// How many workgroups of a given LDS size / width does the chip really run at once?  Every workgroup stamps the
// 100 MHz clock when it starts and when it ends (after spinning ~30 us); the host counts the workgroups that
// started before the first one ended.   hipcc --offload-arch=gfx950 -O2 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long* st, unsigned long long* en, int spin_ticks) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) { st[blockIdx.x] = t0; en[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); }
}
int main() {
  unsigned long long *st, *en;
  hipMalloc(&st, 4096 * 8); hipMalloc(&en, 4096 * 8);
  struct Case { int threads, lds_kb, grid; };
  const Case cases[] = {{512, 144, 240}, {512, 144, 256}, {512, 144, 270}, {256, 132, 256}, {448, 140, 256}, {256, 48, 512},
                        {576, 36, 240}, {576, 36, 256}};
  for (const Case& c : cases) {
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds_kb * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(spin, dim3(c.grid), dim3(c.threads), c.lds_kb * 1024, 0, st, en, 3000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> s(c.grid), e(c.grid);
    hipMemcpy(s.data(), st, c.grid * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), en, c.grid * 8, hipMemcpyDeviceToHost);
    const unsigned long long first_end = *std::min_element(e.begin(), e.end()), t0 = *std::min_element(s.begin(), s.end());
    int first_round = 0; for (int i = 0; i < c.grid; ++i) first_round += s[i] < first_end;
    const unsigned long long last_end = *std::max_element(e.begin(), e.end());
    printf("threads %3d lds %3d KB grid %3d: %3d workgroups started before the first one ended; whole launch %.1f us (spin 30 us)\n",
           c.threads, c.lds_kb, c.grid, first_round, (last_end - t0) / 100.0);
  }
  return 0;
}
