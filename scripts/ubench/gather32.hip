// gather32.hip -- what a gather of 32-byte records costs a CU's L1: (a) every lane fetches both 16-byte halves of ITS record (two
// instructions, 64 distinct lines each) against (b) lane pairs fetch one record per instruction -- lane 2i the low half, lane 2i + 1
// the high half, the halves exchanged by DPP -- two instructions for the pair's two records, 32 distinct lines each.
// build: hipcc --offload-arch=gfx950 -O3 gather32.hip -o _bin/gather32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double double2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dpp_swap1(double v) {   // the value of lane ^ 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(64) void k(const double2v* __restrict__ rec, const int* __restrict__ idx, int per, int n, double* out) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n) return;
  double acc = 0.0;
  const int lane = threadIdx.x;
  for (int c = 0; c < per; c += 2) {
    const int j0 = idx[(size_t)c * n + q], j1 = idx[(size_t)(c + 1) * n + q];
    if (MODE == 0) {
      const double2v a0 = rec[2 * (size_t)j0], a1 = rec[2 * (size_t)j0 + 1];
      const double2v b0 = rec[2 * (size_t)j1], b1 = rec[2 * (size_t)j1 + 1];
      acc += a0.x * a0.y + a1.x + a1.y + b0.x * b0.y + b1.x + b1.y;
    } else {
      // records of the pair: A = this pair's EVEN lane's j0 ... simplest: the pair walks the even lane's (j0, j1) then the odd lane's
      const int ja = __builtin_amdgcn_mov_dpp(j0, 0xA0, 0xf, 0xf, true);   // quad_perm [0,0,2,2]: the even lane's j0
      const int jb = __builtin_amdgcn_mov_dpp(j0, 0xF5, 0xf, 0xf, true);   // quad_perm [1,1,3,3]: the odd lane's j0
      const int jc = __builtin_amdgcn_mov_dpp(j1, 0xA0, 0xf, 0xf, true);
      const int jd = __builtin_amdgcn_mov_dpp(j1, 0xF5, 0xf, 0xf, true);
      const int h = lane & 1;
      const double2v ra = rec[2 * (size_t)ja + h], rb = rec[2 * (size_t)jb + h], rc = rec[2 * (size_t)jc + h], rd = rec[2 * (size_t)jd + h];
      // even lane wants A and C whole, odd lane B and D whole: own half + partner's half
      const double2v mine0 = h ? rb : ra, theirs0 = h ? ra : rb, mine1 = h ? rd : rc, theirs1 = h ? rc : rd;
      const double ox0 = dpp_swap1(theirs0.x), oy0 = dpp_swap1(theirs0.y), ox1 = dpp_swap1(theirs1.x), oy1 = dpp_swap1(theirs1.y);
      // (even lane: mine = low half, other = high half; odd lane the other way round -- the sum below is symmetric in the halves' roles up to the product)
      const double lx0 = h ? ox0 : mine0.x, ly0 = h ? oy0 : mine0.y, hx0 = h ? mine0.x : ox0, hy0 = h ? mine0.y : oy0;
      const double lx1 = h ? ox1 : mine1.x, ly1 = h ? oy1 : mine1.y, hx1 = h ? mine1.x : ox1, hy1 = h ? mine1.y : oy1;
      acc += lx0 * ly0 + hx0 + hy0 + lx1 * ly1 + hx1 + hy1;
    }
  }
  out[q] = acc;
}
int main() {
  const int n = 1000000, per = 24, nrec = 1000000;
  double2v* rec; int* idx; double* out;
  hipMalloc(&rec, (size_t)nrec * 32); hipMalloc(&idx, (size_t)n * per * 4); hipMalloc(&out, n * 8);
  std::vector<double> hr((size_t)nrec * 4);
  for (size_t i = 0; i < hr.size(); ++i) hr[i] = (double)(i % 97) * 0.25;
  hipMemcpy(rec, hr.data(), hr.size() * 8, hipMemcpyHostToDevice);
  for (int local = 0; local < 2; ++local) {
    std::vector<int> hi((size_t)n * per);
    unsigned long long s = 88172645463325252ull;
    for (int q = 0; q < n; ++q)
      for (int c = 0; c < per; ++c) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        // local: candidates near the query's own position in the record array (cell-sorted records, tile-sorted queries); else anywhere
        hi[(size_t)c * n + q] = local ? (int)(((long long)q + (long long)(s % 4096) - 2048 + nrec) % nrec) : (int)(s % nrec);
      }
    hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> o0(n), o1(n);
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 63) / 64), dim3(64), 0, 0, rec, idx, per, n, out);
        else hipLaunchKernelGGL(k<1>, dim3((n + 63) / 64), dim3(64), 0, 0, rec, idx, per, n, out);
      };
      for (int w = 0; w < 3; ++w) launch();
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(mode ? o1.data() : o0.data(), out, n * 8, hipMemcpyDeviceToHost);
      printf("%s %-34s %7.2f us per launch (%d records of 32 B per query, %d queries)\n", local ? "near  " : "random", mode ? "lane pairs, one record per instr" : "every lane both halves", ms * 1e2, per, n);
    }
    int bad = 0;
    for (int q = 0; q < n; ++q) bad += o0[q] != o1[q];
    printf("  results differ in %d of %d\n", bad, n);
  }
  return 0;
}
