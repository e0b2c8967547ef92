// tl_prep.hpp -- the one-launch Solve prepares its own factor set (SolvePrep): caps in index order, compaction or refresh, done
// by every wave for ITS chunk.  Included by tl_gn.hip inside namespace tl, after the K3 helpers (ChunkData, SingleWork,
// single_chunk_of, CorrSeg ...).
#pragma once

// ---- the Solve prepares its own factor set (SolvePrep) ------------------------------------------------------------------
// What k_prepare_small does in a launch of its own -- caps in index order (registration.cpp:448/:538/:592/:735), compaction,
// or the refresh of an unchanged set -- done by every wave for ITS chunk: the kind's flag bytes (at most kFlagbStride, 64 per
// lane) become two 64-bit masks per lane, the cap is one select in the counted mask, a position of the compact set is found by
// a search over the lanes' prefix counts and a select in the valid mask, and the wave fetches its correspondences straight from
// the slot arrays the search wrote -- writing them to the compact arrays on the side, for the finish kernel and the host.
// Same added set, same order, same records as k_prepare_small (the prefix sums are integers): tests/test_gpu_parity.py compares
// the device-driven loop (this path) with the stepwise API (k_prepare_small) bit for bit.
__device__ __forceinline__ unsigned flag_nibble(unsigned x) {   // bit 0 of the four bytes of x -> bits 0..3
  return (((x & 0x01010101u) * 0x01020408u) >> 24) & 0xfu;
}
__device__ __forceinline__ int select64(unsigned long long m, int r) {   // position of the r-th (from 0) set bit; r < popcount(m)
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll((m >> pos) & ((1ull << w) - 1ull));
    if (r >= c) { r -= c; pos += w; }
  }
  return pos;
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  *total = __shfl(incl, 63, 64);
  return incl - v;
}
struct FlagBytes { uint4 q[4]; };
__device__ __forceinline__ void load_flag_bytes(const unsigned char* __restrict__ flagb, int kind, int lane, FlagBytes& f) {
  const uint4* p = reinterpret_cast<const uint4*>(flagb + (size_t)kind * kFlagbStride) + lane * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) f.q[i] = p[i];
}
// The added set of one kind, as the wave sees it: this lane's 64 slots as a mask, the exclusive prefix of the lanes'
// counts and the size of the set.   added(i) <=> valid(i) && #counted before i < maxnum
struct KindSet {
  unsigned vlo, vhi;   // this lane's added slots (64 lane + bit)
  int pv;              // added slots in front of this lane
  int total;
};
__device__ __forceinline__ KindSet kind_set_of(const SolvePrep& P, const CorrView& cv, int kind, int lane, const FlagBytes& f) {
  const int nk = P.sv.slot_off[kind + 1] - P.sv.slot_off[kind];
  unsigned vlo = 0u, vhi = 0u, clo = 0u, chi = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned wd[4] = {f.q[i].x, f.q[i].y, f.q[i].z, f.q[i].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int nib = i * 4 + c;   // slots 4 nib .. 4 nib + 3 of this lane
      const unsigned v = flag_nibble(wd[c]), ct = flag_nibble(wd[c] >> 1);
      if (nib < 8) { vlo |= v << (4 * nib); clo |= ct << (4 * nib); }
      else { vhi |= v << (4 * (nib - 8)); chi |= ct << (4 * (nib - 8)); }
    }
  }
  unsigned long long valid = (unsigned long long)vlo | ((unsigned long long)vhi << 32);
  unsigned long long counted = (unsigned long long)clo | ((unsigned long long)chi << 32);
  {  // bytes past the kind's last slot are whatever an earlier, larger frame left there
    const int mine = nk - lane * 64;
    const unsigned long long keep = mine >= 64 ? ~0ull : (mine <= 0 ? 0ull : ((1ull << mine) - 1ull));
    valid &= keep;
    counted &= keep;
  }
  int ctot;
  const int cc = __popcll(counted);
  const int pc = wave_excl_scan(cc, lane, &ctot);
  const int room = P.maxnum[kind] - pc;
  if (room <= 0) valid = 0ull;
  else if (room <= cc) valid &= (2ull << select64(counted, room - 1)) - 1ull;   // up to and including the room-th counted slot
  KindSet ks;
  const int vc = __popcll(valid);
  ks.pv = wave_excl_scan(vc, lane, &ks.total);
  if (ks.total > cv.k[kind].cap) ks.total = cv.k[kind].cap;   // cannot happen (cap >= min(n, maxnum)); defensive, as in k_prepare_small
  ks.vlo = (unsigned)valid;
  ks.vhi = (unsigned)(valid >> 32);
  return ks;
}
// local slot (within the kind) of compact position p, p < total
__device__ __forceinline__ int slot_of_position(int p, const KindSet& ks) {
  int L = 0;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int pc = __shfl(ks.pv, L + s, 64);   // (L + s <= 63)
    if (pc <= p) L += s;
  }
  const int r = p - __shfl(ks.pv, L, 64);
  const unsigned long long word = (unsigned long long)(unsigned)__shfl((int)ks.vlo, L, 64) |
                                  ((unsigned long long)(unsigned)__shfl((int)ks.vhi, L, 64) << 32);
  return L * 64 + select64(word, r);
}
// one correspondence from the slot arrays into half `H` (0: .x, 1: .y) of the chunk registers + the compact arrays
template <int H>
__device__ __forceinline__ void take_slot(const SlotView& sv, const CorrSeg& seg, int kind, int local, int p, ChunkData& b) {
  const int slot = sv.slot_off[kind] + local;
  const double px = sv.sx[slot], py = sv.sy[slot], pz = sv.sz[slot], w = sv.w_src[slot];
  const double2* q = reinterpret_cast<const double2*>(sv.raw + (size_t)slot * 8);
  const double2 q0 = q[0], q1 = q[1];
  double2 q2 = double2{0.0, 0.0};
  double d = 0.0;
  if (kind == TLOAM_KIND_EDGE) q2 = q[2];
  if (kind <= TLOAM_KIND_GROUND) d = q[3].x;
#define TL_PUT(field, val) if (H == 0) b.field.x = (val); else b.field.y = (val);
  TL_PUT(px, px) TL_PUT(py, py) TL_PUT(pz, pz) TL_PUT(ax, q0.x) TL_PUT(ay, q0.y) TL_PUT(az, q1.x) TL_PUT(w, w)
  TL_PUT(bx, q1.y) TL_PUT(by, q2.x) TL_PUT(bz, q2.y) TL_PUT(d, d)
#undef TL_PUT
  seg.idx[p] = local + sv.src_lo[kind];
  seg.px[p] = px; seg.py[p] = py; seg.pz[p] = pz;
  seg.ax[p] = q0.x; seg.ay[p] = q0.y; seg.az[p] = q1.x;
  if (kind == TLOAM_KIND_EDGE) { seg.bx[p] = q1.y; seg.by[p] = q2.x; seg.bz[p] = q2.y; }
  if (kind <= TLOAM_KIND_GROUND) seg.d[p] = d;
  seg.w[p] = w;        // weight captured by value at construction (registration.hpp:51,76,96)
  seg.cost[p] = 0.0;   // fresh side-channel slot (registration.cpp:1118-1121)
}
// BUILD: returns the size of the wave's kind, fills b with the wave's chunk and slot[] with its correspondences' slots
__device__ __forceinline__ int self_compact(const SolvePrep& P, const CorrView& cv, const SingleWork& wk, int lane, const FlagBytes& f,
                                            ChunkData& b, int slot[2]) {
  const int kind = wk.kind;
  const KindSet ks = kind_set_of(P, cv, kind, lane, f);
  const int total = ks.total;
  const CorrSeg& seg = cv.k[kind];
  const bool two = single_chunk_of(kind) == kChunk;
  const int p0 = wk.j, p1 = wk.j + 1;
  // (every lane takes part in the searches: they exchange through the whole wave)
  const int l0 = total > 0 ? slot_of_position(p0 < total ? p0 : total - 1, ks) : 0;
  if (p0 < total) take_slot<0>(P.sv, seg, kind, l0, p0, b);
  slot[0] = P.sv.slot_off[kind] + l0;
  slot[1] = slot[0];
  if (two) {
    const int l1 = total > 0 ? slot_of_position(p1 < total ? p1 : total - 1, ks) : 0;
    if (p1 < total) take_slot<1>(P.sv, seg, kind, l1, p1, b);
    slot[1] = P.sv.slot_off[kind] + l1;
  }
  return total;
}
// the slots of the wave's correspondences of a set that is kept or refreshed
__device__ __forceinline__ void slots_of_chunk(const SolvePrep& P, const CorrView& cv, const SingleWork& wk, int n, int slot[2]) {
  const int kind = wk.kind;
  const CorrSeg& seg = cv.k[kind];
  const bool two = single_chunk_of(kind) == kChunk;
  const int base = P.sv.slot_off[kind] - P.sv.src_lo[kind];
  slot[0] = wk.j < n ? base + seg.idx[wk.j] : P.sv.slot_off[kind];
  slot[1] = (two && wk.j + 1 < n) ? base + seg.idx[wk.j + 1] : slot[0];
}
// REFRESH: the set of the previous iteration with new captured weights and zeroed slots (k_refresh); b holds the chunk
__device__ __forceinline__ void self_refresh(const SolvePrep& P, const CorrView& cv, const SingleWork& wk, int n, const int slot[2],
                                             ChunkData& b) {
  const CorrSeg& seg = cv.k[wk.kind];
  const bool two = single_chunk_of(wk.kind) == kChunk;
  if (wk.j < n) {
    const double w = P.sv.w_src[slot[0]];
    b.w.x = w; seg.w[wk.j] = w; seg.cost[wk.j] = 0.0;
  }
  if (two && wk.j + 1 < n) {
    const double w = P.sv.w_src[slot[1]];
    b.w.y = w; seg.w[wk.j + 1] = w; seg.cost[wk.j + 1] = 0.0;
  }
}
