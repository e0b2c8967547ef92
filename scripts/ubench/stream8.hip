// Ceiling probe for K3's access pattern: N streams of fp64, 16 B per lane per load, persistent waves with
// a 2-deep software pipeline, trivial arithmetic.  Prints achieved GB/s for a 75 MB working set (same size as
// the 1 M-correspondence sweep) re-read back to back (Infinity-Cache resident, like K3's sweeps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NS>
__global__ __launch_bounds__(256, 2) void k_stream(const double* __restrict__ base, size_t stride_elems, int nchunks,
                                                   double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave, W = gridDim.x * 4;
  double acc = 0.0;
  double2 b0[NS], b1[NS];
  auto fetch = [&](int c, double2* b) {
    const unsigned o = ((unsigned)c * 128u + lane * 2u) * 8u;
#pragma unroll
    for (int s = 0; s < NS; ++s) b[s] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base + s * stride_elems) + o);
  };
  auto consume = [&](const double2* b) {
#pragma unroll
    for (int s = 0; s < NS; ++s) acc += b[s].x * b[s].y;
  };
  const int m = (nchunks - gw + W - 1) / W;
  if (m <= 0) return;
  fetch(gw, b0);
  for (int t = 1;; t += 2) {
    if (t >= m) { consume(b0); break; }
    fetch(gw + t * W, b1);
    consume(b0);
    if (t + 1 >= m) { consume(b1); break; }
    fetch(gw + (t + 1) * W, b0);
    consume(b1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int NS = 8;
  const size_t n = 1000000;             // elements per stream
  const size_t stride = (n + 255) / 256 * 256 + 256;
  double *d, *out;
  hipMalloc(&d, sizeof(double) * stride * NS);
  hipMalloc(&out, sizeof(double) * 4096 * 256);
  hipMemset(d, 0, sizeof(double) * stride * NS);
  const int nchunks = (int)((n + 127) / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 384, 489, 512, 768, 1024, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_stream<NS>, dim3(blocks), dim3(256), 0, 0, d, stride, nchunks, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("streams %d blocks %4d: %.2f us/launch  %.0f GB/s (%.1f MB)\n", NS, blocks, ms * 10, n * 8.0 * NS / (ms * 1e-5) / 1e9, n * 8.0 * NS / 1e6);
    }
  }
  return 0;
}
