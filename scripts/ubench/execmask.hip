// execmask.hip -- does a wave64 fp64 VALU instruction cost less when only part of the wave is enabled in EXEC?  (gfx950)
// The minimiser's step is wave-uniform: every lane computes the same values.  If the SIMD skipped the 16-lane passes whose
// lanes are all masked off, running the step on lanes 0..15 only would cut its issue time.  One wave; s_memtime around
// unrolled chains inside an `if (lane < n)` region.   build: hipcc --offload-arch=gfx950 -O3 execmask.hip -o _bin/execmask
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 256
template <int LANES>
__device__ void run(double* out, unsigned long long* cyc, const double* in, int slot) {
  const int lane = threadIdx.x;
  double a = in[0], b = in[1], c = in[2];
  double p = in[1], q = in[2], s = in[3], u = in[4];
  unsigned long long d0 = 0, d1 = 0;
  if (lane < LANES) {
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_fma(a, b, c);          // dependent chain
    asm volatile("" :: "v"(a));
    unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { p = __builtin_fma(p, b, c); q = __builtin_fma(q, b, c); s = __builtin_fma(s, b, c); u = __builtin_fma(u, b, c); }
    asm volatile("" :: "v"(p), "v"(q), "v"(s), "v"(u));
    unsigned long long t2 = __builtin_readcyclecounter();
    d0 = t1 - t0; d1 = t2 - t1;
    out[lane] = a + p + q + s + u;
  }
  if (lane == 0) { cyc[slot * 2] = d0; cyc[slot * 2 + 1] = d1; }
}
__global__ void k(double* out, unsigned long long* cyc, const double* in) {
  run<64>(out, cyc, in, 0);
  run<32>(out, cyc, in, 1);
  run<16>(out, cyc, in, 2);
  run<1>(out, cyc, in, 3);
}
int main() {
  double h_in[8] = {1.0000001, 0.9999999, 1e-9, 1.5, 2.5, 3, 4, 5}, *d_in, *d_out;
  unsigned long long* d_cyc, h[8];
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 64 * 8); hipMalloc(&d_cyc, sizeof(h));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_cyc, d_in);
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int lanes[4] = {64, 32, 16, 1};
    for (int i = 0; i < 4; ++i)
      printf("rep %d  EXEC = %2d lanes: %d dependent v_fma_f64 in %llu cycles (%.2f each), %d in 4 independent chains in %llu (%.2f each)\n", rep,
             lanes[i], N, h[2 * i], (double)h[2 * i] / N, N, h[2 * i + 1], (double)h[2 * i + 1] / N);
  }
  return 0;
}
