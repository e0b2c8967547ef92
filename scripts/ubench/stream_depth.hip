// Probe for K3's pipeline shape: 8 streams of fp64 (16 B per lane per load), persistent waves, a D-deep
// software pipeline (D-1 chunks in flight while one is consumed), WPS waves per SIMD and a synthetic
// dependent fp64 chain of FL fused ops per element (K3 spends ~250 VALU ops per 2-correspondence lane-chunk).
// Question answered: does a deeper pipeline at lower occupancy move more bytes than 2-deep at 2 waves/SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NS = 8;
template <int D, int WPS, int FL, bool WR = false>
__global__ __launch_bounds__(256, WPS) void k_stream(const double* __restrict__ base, size_t stride_elems, int nchunks,
                                                     double* __restrict__ out, double* __restrict__ wr = nullptr) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave, W = gridDim.x * 4;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double2 b[D][NS];
  const int m = (nchunks - gw + W - 1) / W;
  if (m <= 0) return;
  auto fetch = [&](int t, double2* bb) {
    const int tt = t < m ? t : m - 1;  // clamped: a drained pipeline re-reads its last chunk (cache hit)
    const unsigned o = ((unsigned)(gw + tt * W) * 128u + lane * 2u) * 8u;
#pragma unroll
    for (int s = 0; s < NS; ++s) bb[s] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base + s * stride_elems) + o);
  };
  auto consume = [&](const double2* bb, int t = 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      double v[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) v[s] = e ? bb[s].y : bb[s].x;
#pragma unroll
      for (int f = 0; f < FL; ++f) acc[f & 7] = fma(v[f % NS], v[(f + 3) % NS], acc[f & 7]);
    }
    if (WR) *reinterpret_cast<double2*>(wr + (size_t)(gw + t * W) * 128u + lane * 2u) = double2{acc[0], acc[1]};
  };
#pragma unroll
  for (int i = 0; i < D - 1; ++i) fetch(i, b[i]);
  for (int t = 0; t < m; t += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (t + u < m) {
        fetch(t + u + D - 1, b[(u + D - 1) % D]);
        consume(b[u], t + u);
      }
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int D, int WPS, int FL, bool WR = false>
void run(const double* d, size_t stride, int nchunks, double* out, size_t n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * WPS;
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_stream<D, WPS, FL, WR>), dim3(blocks), dim3(256), 0, 0, d, stride, nchunks, out, out + 4096 * 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("depth %d waves/SIMD %d flops/elem %3d write %d: %.2f us/launch  %.0f GB/s (reads only)\n", D, WPS, FL, (int)WR, best * 10, n * 8.0 * NS / (best * 1e-5) / 1e9);
}
int main() {
  const size_t n = 1170000;             // elements per stream: 74.9 MB like the 1 M pre-built sweep
  const size_t stride = (n + 255) / 256 * 256 + 256;
  double *d, *out;
  hipMalloc(&d, sizeof(double) * stride * NS);
  hipMalloc(&out, sizeof(double) * (4096 * 256 + stride + 4096));
  hipMemset(d, 0, sizeof(double) * stride * NS);
  const int nchunks = (int)((n + 127) / 128);
  run<2, 2, 8>(d, stride, nchunks, out, n);
  run<2, 2, 256>(d, stride, nchunks, out, n);
  run<2, 2, 128, true>(d, stride, nchunks, out, n);
  run<2, 2, 256, true>(d, stride, nchunks, out, n);
  run<2, 2, 8, true>(d, stride, nchunks, out, n);
  run<2, 2, 128>(d, stride, nchunks, out, n);
  run<3, 2, 128>(d, stride, nchunks, out, n);
  run<2, 1, 128>(d, stride, nchunks, out, n);
  run<3, 1, 128>(d, stride, nchunks, out, n);
  run<4, 1, 128>(d, stride, nchunks, out, n);
  run<5, 1, 128>(d, stride, nchunks, out, n);
  run<6, 1, 128>(d, stride, nchunks, out, n);
  run<4, 1, 8>(d, stride, nchunks, out, n);
  run<4, 2, 8>(d, stride, nchunks, out, n);
  run<3, 3, 8>(d, stride, nchunks, out, n);
  run<2, 4, 8>(d, stride, nchunks, out, n);
  return 0;
}
