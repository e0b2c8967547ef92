// tl_probe.hip -- the on-box ceiling the roofline kernel is held against besides the 8 TB/s data-sheet figure ("also
// report fraction of an on-box measured ... bandwidth").  K3 is a READ stream: 64-80 B in per correspondence, 8 B out.  A
// device-to-device copy (what rounds 1-5 quoted: `measured_copy_GBps`) moves half its bytes as writes and measured BELOW the
// kernel it was meant to bound (K3 at 1.19-1.27x of it) -- it is no ceiling for a read stream.  This probe
// has K3's access pattern and nothing else: eight fp64 streams, one 16-byte load per lane per stream (1 KiB per wave instruction),
// persistent waves two blocks per CU walking 128-element chunks, the next chunk requested while the current one is consumed, one
// multiply-add per 16 bytes, 8 bytes out per thread per launch.
#include "tl_ctx.hpp"

using namespace tl;

namespace {
constexpr int kProbeStreams = 8;
template <bool NT>
__global__ __launch_bounds__(256, 2) void k_read_stream(const double* __restrict__ base, size_t stride_elems, int nchunks,
                                                        double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave, W = gridDim.x * 4;
  typedef double v2d __attribute__((ext_vector_type(2)));
  double acc = 0.0;
  v2d b0[kProbeStreams], b1[kProbeStreams];
  auto fetch = [&](int c, v2d* b) {
    const size_t o = ((size_t)c * 128u + (size_t)lane * 2u) * 8u;
#pragma unroll
    for (int s = 0; s < kProbeStreams; ++s) {
      const v2d* p = reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base + (size_t)s * stride_elems) + o);
      b[s] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  };
  auto consume = [&](const v2d* b) {
#pragma unroll
    for (int s = 0; s < kProbeStreams; ++s) acc = __builtin_fma(b[s].x, b[s].y, acc);
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
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
}  // namespace

extern "C" {

// `launches` back-to-back passes over a buffer of ~`bytes` (eight streams) between one HIP event pair on the context's stream.
// bytes well above the 256 MiB Infinity Cache: every pass comes from HBM; bytes = the sweep's own 75 MB: the passes hit the
// Infinity Cache as the sweeps of a Solve do.  *gbps = bytes read per second / 1e9.
int tloam_time_read_stream(tloam_ctx* c, size_t bytes, int launches, double* gbps) {
  if (!c || !gbps || launches < 1 || bytes < (size_t)kProbeStreams * 128 * 8 || bytes > ((size_t)1 << 34)) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  const size_t n = bytes / 8 / kProbeStreams / 128 * 128;        // elements per stream, whole chunks
  const size_t stride = n + 256;
  double *d = nullptr, *out = nullptr;
  const int blocks = (c->device_cus > 0 ? c->device_cus : 256) * 2;
  HIPC(c, hipMalloc((void**)&d, sizeof(double) * stride * kProbeStreams));
  hipError_t e = hipMalloc((void**)&out, sizeof(double) * (size_t)blocks * 256);
  if (e != hipSuccess) { (void)hipFree(d); HIPC(c, e); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  const int nchunks = (int)(n / 128);
  e = hipMemsetAsync(d, 0, sizeof(double) * stride * kProbeStreams, c->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_read_stream<true>, dim3(blocks), dim3(256), 0, c->stream, d, stride, nchunks, out);
    e = hipEventRecord(e0, c->stream);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_read_stream<true>, dim3(blocks), dim3(256), 0, c->stream, d, stride, nchunks, out);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  (void)hipFree(out);
  HIPC(c, e);
  *gbps = (double)n * 8.0 * kProbeStreams * launches / ((double)ms * 1e-3) / 1e9;
  return TLOAM_OK;
}

}  // extern "C"
