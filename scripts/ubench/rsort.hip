// rsort.hip -- what a device-wide radix sort of (cell, index) pairs costs on the part, as the alternative to the histogram atomics
// of the grid build / query sort (atomics.hip: one returning atomic per element runs at a flat 21.9 G/s = 45.7 us per 1 M).
// rocPRIM's tuned sort stands in for "the best a hand-written LSD sort with LDS histograms could reach": if IT does not beat
// count + scan + scatter (67 + 26-32 + ~40 us at 1 M targets, 3.2 M cells = 22 key bits), a hand-written one will not either.
// build: hipcc --offload-arch=gfx950 -O3 rsort.hip -o _bin/rsort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <vector>
static void run(int n, unsigned cells, int bits, bool pairs) {
  unsigned *k_in, *k_out, *v_in, *v_out;
  hipMalloc(&k_in, n * 4); hipMalloc(&k_out, n * 4); hipMalloc(&v_in, n * 4); hipMalloc(&v_out, n * 4);
  std::vector<unsigned> h(n), v(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (unsigned)(s % cells); v[i] = i; }
  hipMemcpy(k_in, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v_in, v.data(), n * 4, hipMemcpyHostToDevice);
  size_t tmp_bytes = 0; void* tmp = nullptr;
  if (pairs) rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, (unsigned)bits, 0);
  else rocprim::radix_sort_keys(nullptr, tmp_bytes, k_in, k_out, (size_t)n, 0u, (unsigned)bits, 0);
  hipMalloc(&tmp, tmp_bytes);
  auto once = [&] {
    if (pairs) rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, (unsigned)bits, 0);
    else rocprim::radix_sort_keys(tmp, tmp_bytes, k_in, k_out, (size_t)n, 0u, (unsigned)bits, 0);
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) once();
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) once();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> o(n);
  hipMemcpy(o.data(), k_out, n * 4, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int i = 1; i < n; ++i) ok &= o[i - 1] <= o[i];
  printf("rocprim radix_sort_%-5s n %8d cells %8u bits %2d: %7.2f us per sort (tmp %zu B) %s\n", pairs ? "pairs" : "keys", n, cells, bits,
         ms * 1e3 / reps, tmp_bytes, ok ? "sorted" : "NOT SORTED");
  hipFree(k_in); hipFree(k_out); hipFree(v_in); hipFree(v_out); hipFree(tmp);
}
int main() {
  run(1000000, 3200000u, 22, true);
  run(1000000, 3200000u, 22, false);
  run(1000000, 3200000u, 32, true);
  run(1000000, 65536u, 16, true);    // (two 8-bit passes: what a coarser first key would cost)
  run(1000000, 2048u, 11, true);
  run(83500, 320000u, 19, true);     // the KITTI-size targets
  run(16384, 320000u, 19, true);
  return 0;
}
