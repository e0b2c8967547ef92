// lat.hip -- instruction latency / issue micro-benchmark for the one-wave minimiser step (gfx950).
// One wave; s_memtime around unrolled chains.  build: hipcc --offload-arch=gfx950 -O3 lat.hip -o _bin/lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 256
__global__ void k(double* out, unsigned long long* cyc, const double* in, unsigned long long* gmem) {
  double a = in[0], b = in[1], c = in[2], d = in[3], e = in[4];
  unsigned long long t0, t1;
  int r = 0;
  // 1: dependent fma chain
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) a = __builtin_fma(a, b, c);
  asm volatile("" :: "v"(a));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 2: four independent fma chains
  double p = b, q = c, s = d, u = e;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) { p = __builtin_fma(p, b, c); q = __builtin_fma(q, b, c); s = __builtin_fma(s, b, c); u = __builtin_fma(u, b, c); }
  asm volatile("" :: "v"(p), "v"(q), "v"(s), "v"(u));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 3: dependent rsq chain
  double x = d;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rsq(x);
  asm volatile("" :: "v"(x));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 4: independent rsq (4 chains x 16)
  double x1 = d, x2 = e, x3 = b, x4 = c;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) { x1 = __builtin_amdgcn_rsq(x1); x2 = __builtin_amdgcn_rsq(x2); x3 = __builtin_amdgcn_rsq(x3); x4 = __builtin_amdgcn_rsq(x4); }
  asm volatile("" :: "v"(x1), "v"(x2), "v"(x3), "v"(x4));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 5: dependent v_mov_b32 chain (plain 32-bit VALU)
  int m = (int)in[5];
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(m) : "v"(r));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 6: s_memtime overhead
  t0 = __builtin_readcyclecounter();
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 7: sincos + atan (ocml) once each
  double sn, cs;
  t0 = __builtin_readcyclecounter();
  sincos(e, &sn, &cs);
  asm volatile("" :: "v"(sn), "v"(cs));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  double at = atan(e);
  asm volatile("" :: "v"(at));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 9: IEEE division, sqrt
  t0 = __builtin_readcyclecounter();
  double dv = d / e;
  asm volatile("" :: "v"(dv));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  double sq = sqrt(d);
  asm volatile("" :: "v"(sq));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 11: device-scope load round trip (miss), then again (same line)
  t0 = __builtin_readcyclecounter();
  unsigned long long g = __hip_atomic_load(gmem + 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(g));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  g += __hip_atomic_load(gmem + 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(g));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 13: returning device-scope atomic
  t0 = __builtin_readcyclecounter();
  unsigned long long tk = 0;
  if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(gmem, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(tk));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 14: sc1 store + vmcnt(0)
  t0 = __builtin_readcyclecounter();
  __hip_atomic_store(gmem + 2048 + threadIdx.x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 15: plain load (L2/HBM miss) of a fresh line
  t0 = __builtin_readcyclecounter();
  unsigned long long pl = gmem[4096 + threadIdx.x];
  asm volatile("" :: "v"(pl));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 16: LDS write + read back
  __shared__ double sh[64];
  t0 = __builtin_readcyclecounter();
  sh[threadIdx.x] = a;
  double lr = sh[(threadIdx.x + 1) & 63];
  asm volatile("" :: "v"(lr));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  // 17: readlane + use
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) { int lo = __builtin_amdgcn_readlane(m, 1); m += lo; }
  asm volatile("" :: "v"(m));
  t1 = __builtin_readcyclecounter(); cyc[r++] = t1 - t0;
  out[threadIdx.x] = a + p + q + s + u + x + x1 + x2 + x3 + x4 + m + sn + cs + at + dv + sq + (double)g + (double)tk + (double)pl + lr;
}
int main() {
  double *out, *in; unsigned long long *cyc, *gmem;
  hipMalloc(&out, 64 * 8); hipMalloc(&in, 64 * 8); hipMalloc(&cyc, 64 * 8); hipMalloc(&gmem, 8192 * 8);
  double h[8] = {1.0000001, 0.9999999, 1e-9, 2.5, 0.37, 3.0, 0, 0};
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(gmem, 0, 8192 * 8);
  const char* names[] = {"dep fma x256", "4-chain fma x256", "dep rsq x64", "4-chain rsq x64", "dep v_add_u32 x256", "memtime pair", "sincos", "atan",
                         "div", "sqrt", "sc1 load miss", "sc1 load again", "returning atomic", "sc1 store + vmcnt0", "plain load miss", "lds write+read", "readlane+add x16"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, in, gmem);
    hipDeviceSynchronize();
    unsigned long long hc[32]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("rep %d:", rep);
    for (int i = 0; i < 17; ++i) printf(" [%s] %llu", names[i], hc[i]);
    printf("\n");
  }
  return 0;
}
