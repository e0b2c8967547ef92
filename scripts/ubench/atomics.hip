// atomics.hip -- throughput of one histogram atomic per thread on random cells (the grid build / query sort of the 1 M frame:
// 1 M returning 64-bit atomics on a 3.2 M-entry table take 50-67 us there).  Variants: 64 / 32-bit counters, returning or not,
// device or workgroup scope, table sizes.  build: hipcc --offload-arch=gfx950 -O3 atomics.hip -o _bin/atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <class T, int MODE>   // MODE 0: returning, agent | 1: non-returning, agent | 2: returning, workgroup scope (XCD's L2: NOT coherent across XCDs)
__global__ void k(T* tab, const unsigned* cell, int n, unsigned* sink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (MODE == 0) { const T r = __hip_atomic_fetch_add(&tab[cell[i]], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sink[i] = (unsigned)r; }
  if (MODE == 1) { __hip_atomic_fetch_add(&tab[cell[i]], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (MODE == 2) { const T r = __hip_atomic_fetch_add(&tab[cell[i]], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); sink[i] = (unsigned)r; }
}
template <class T, int MODE>
void run(const char* name, size_t cells, int n) {
  T* tab; unsigned *cell, *sink;
  hipMalloc(&tab, cells * sizeof(T)); hipMalloc(&cell, n * 4); hipMalloc(&sink, n * 4);
  hipMemset(tab, 0, cells * sizeof(T));
  std::vector<unsigned> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (unsigned)(s % cells); }
  hipMemcpy(cell, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<T, MODE>), dim3((n + 255) / 256), dim3(256), 0, 0, tab, cell, n, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<T, MODE>), dim3((n + 255) / 256), dim3(256), 0, 0, tab, cell, n, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s cells %8zu n %8d: %7.2f us per launch, %6.2f G atomics/s\n", name, cells, n, ms * 1e3 / reps, n / (ms * 1e-3 / reps) * 1e-9);
  hipFree(tab); hipFree(cell); hipFree(sink);
}
int main() {
  for (size_t cells : {3200000ul, 320000ul, 32000ul}) {
    run<unsigned long long, 0>("u64 returning agent", cells, 1000000);
    run<unsigned, 0>("u32 returning agent", cells, 1000000);
    run<unsigned long long, 1>("u64 non-returning agent", cells, 1000000);
    run<unsigned, 1>("u32 non-returning agent", cells, 1000000);
    run<unsigned long long, 2>("u64 returning workgroup-scope (not coherent)", cells, 1000000);
  }
  run<unsigned long long, 0>("u64 returning agent", 240000, 83500);
  run<unsigned, 0>("u32 returning agent", 240000, 83500);
  run<unsigned long long, 1>("u64 non-returning agent", 240000, 83500);
  return 0;
}
