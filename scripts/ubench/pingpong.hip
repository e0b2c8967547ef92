// pingpong.hip -- round trip of a flag between two workgroups of one launch (gfx950), for the hand-overs of the one-launch
// Solve: (a) device-scope (sc1) store / load, the blocks on different XCDs or on the same one; (b) blocks on the SAME XCD
// (workgroup ids congruent modulo 8), plain store + L1 invalidate + plain load, i.e. through the XCD's own L2.
// Prints the XCC id of the two blocks and the mean round trip.  build: hipcc --offload-arch=gfx950 -O3 pingpong.hip -o _bin/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int MODE>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(unsigned long long* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 1) { asm volatile("buffer_inv sc0" ::: "memory"); return *(volatile unsigned long long*)p; }
  if (MODE == 2) { asm volatile("buffer_inv sc1" ::: "memory"); return *(volatile unsigned long long*)p; }
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // MODE 3: sc0 load, no invalidate
}
// blocks a and b play; everybody else leaves.  flag[0]: a -> b, flag[16]: b -> a (different lines)
template <int MODE>
__global__ void k(unsigned long long* flag, int a, int b, int rounds, unsigned long long base, unsigned long long* out) {
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  if (threadIdx.x != 0) return;
  const unsigned x = xcc_id();
  unsigned long long t0 = wall_clock64();
  unsigned long long spins = 0;
  if (me == a) {
    for (int r = 1; r <= rounds; ++r) {
      put<MODE>(flag, base + r);
      while (get<MODE>(flag + 16) != base + r) { if (++spins > 50000000ull) break; }
    }
    out[0] = wall_clock64() - t0; out[1] = x; out[4] = spins;
  } else {
    for (int r = 1; r <= rounds; ++r) {
      while (get<MODE>(flag) != base + r) { if (++spins > 50000000ull) break; }
      put<MODE>(flag + 16, base + r);
    }
    out[2] = x; out[5] = spins;
  }
}
int main() {
  unsigned long long *flag, *out;
  hipMalloc(&flag, 4096); hipMemset(flag, 0, 4096);
  hipMallocManaged(&out, 64);
  const int rounds = 2000;
  unsigned long long base = 0;
  struct { int mode, a, b; const char* what; } cases[] = {
    {0, 0, 1, "sc1 store/load, blocks 0 and 1 (neighbouring XCDs)"},
    {0, 0, 8, "sc1 store/load, blocks 0 and 8 (same XCD)"},
    {1, 0, 8, "plain store, buffer_inv sc0 + plain load, blocks 0 and 8"},
    {2, 0, 8, "plain store, buffer_inv sc1 + plain load, blocks 0 and 8"},
    {3, 0, 8, "plain store, sc0 load (no invalidate), blocks 0 and 8"},
    {1, 0, 120, "plain store, buffer_inv sc0 + plain load, blocks 0 and 120"},
    {1, 0, 1, "plain store, buffer_inv sc0 + plain load, blocks 0 and 1 (DIFFERENT XCDs: expected to fail or crawl)"},
  };
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < 8; ++i) out[i] = 0;
      base += 1000000;
      if (c.mode == 0) hipLaunchKernelGGL(k<0>, dim3(128), dim3(64), 0, 0, flag, c.a, c.b, rounds, base, out);
      if (c.mode == 1) hipLaunchKernelGGL(k<1>, dim3(128), dim3(64), 0, 0, flag, c.a, c.b, rounds, base, out);
      if (c.mode == 2) hipLaunchKernelGGL(k<2>, dim3(128), dim3(64), 0, 0, flag, c.a, c.b, rounds, base, out);
      if (c.mode == 3) hipLaunchKernelGGL(k<3>, dim3(128), dim3(64), 0, 0, flag, c.a, c.b, rounds, base, out);
      hipDeviceSynchronize();
      if (rep == 1)
        printf("%-100s xcc %llu / %llu  round trip %.3f us  (spins %llu / %llu)\n", c.what, out[1], out[2],
               (double)out[0] * 0.01 / rounds, out[4], out[5]);
    }
  }
  return 0;
}
