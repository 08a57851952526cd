// The forward partition of the q32b layer-1 kernel (gemm_p2.hpp): its cost model and the search for the smallest feasible bound.
// In a header of its own because TWO kernels run it: the GEMM itself (any caller) and -- when the operands are staged by
// stage_fused_kernel -- one workgroup of that launch, which leaves the bound in device memory so that the 256 workgroups of the
// GEMM do not each spend ~15 us of their critical path on the same search.
#pragma once
#include "gemm.hpp"

namespace lirec {

// cost model of the forward partition.  The two wave groups of a workgroup take turns at the matrix pipes (gemm_p2.hpp), so a
// k-step of a tile of g row blocks (32 rows each) costs two barriers plus two multiply phases of 12 g MFMAs: measured
// (tools/micro/p2_bench.hip, cycle stamps) 2 x ~135 + 384 g cycles; a tile's fill and epilogue ~21 000 cycles.  Units of ~190
// cycles: P2_COST_TILE per tile and k-step, P2_COST_RB per row block and k-step, P2_TILE_FIXED per tile.  (Round 3's in-step
// k-loop ran at the LDS-DMA rate -- 8 + g units of 4 KiB per step -- and this model was 8, 1, 48: with it the 24-k-step problems'
// workgroups finished 10 % ahead of the others.)
#define P2_COST_TILE 1
#define P2_COST_RB 2
#ifndef P2_TILE_FIXED
#define P2_TILE_FIXED 110
#endif
// a / b for 0 <= a < 2^24, 0 < b < 2^24: one float reciprocal + fix-up instead of the ~40-instruction integer sequence (the
// partition search below divides ~300 times on every workgroup's critical path)
__host__ __device__ __forceinline__ int p2_div(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
  int q = (int)(__fdividef((float)a, (float)b));
  q -= (q * b > a) ? 1 : 0;
  q += ((q + 1) * b <= a) ? 1 : 0;
  return q;
#else
  return a / b;
#endif
}
// (32-bit arithmetic on purpose -- the bisection below divides ~200 times and a 64-bit division is a ~100-instruction
//  sequence on this machine: the first version spent 70 us in it; rows < 2^21 keep every value below 2^31)
__host__ __device__ __forceinline__ int p2_nt_cost(int g, int ks) {           // a chunk of g row blocks as ceil(g / 8) tiles
  const int nt = (g + 7) >> 3;
  return ks * (P2_COST_TILE * nt + P2_COST_RB * g) + P2_TILE_FIXED * nt;
}
__host__ __device__ __forceinline__ int p2_nt_gmax(int C, int ks, int rb) {    // most row blocks (<= rb) a chunk of cost <= C can hold
  // with nt tiles: g <= 8 nt and ks (T nt + R g) + F nt <= C; the two bounds cross at nt* = C / ((T + 8 R) ks + F)
  int best = 0;
  const int per = (P2_COST_TILE + 8 * P2_COST_RB) * ks + P2_TILE_FIXED;
  const int n0 = p2_div(C, per);
  for (int nt = (n0 > 1 ? n0 : 1); nt <= n0 + 1; ++nt) {
    const int room = C - nt * (P2_COST_TILE * ks + P2_TILE_FIXED);
    if (room <= 0) continue;
    int gq = p2_div(room, P2_COST_RB * ks);
    if (gq > 8 * nt) gq = 8 * nt;
    if (gq > best) best = gq;
  }
  return best > rb ? rb : best;
}

// The smallest bound C for which the chunks of all problems fit `grid` workgroups (each problem's chunks times `nrep` column
// tiles), by four passes of a 64-candidate search -- one candidate per lane of the calling wave.  rbv: row blocks (32 rows) per
// problem, ksv: k-steps per problem, 0 row blocks = problem absent.  Wave-uniform result.
__device__ __forceinline__ int p2_nt_search(const int (&rbv)[LIREC_MAX_PROB], const int (&ksv)[LIREC_MAX_PROB], const int grid,
                                            const int nrep, const int lane) {
  int Clo = 0, Chi = 0;
#pragma unroll
  for (int i = 0; i < LIREC_MAX_PROB; ++i) {
    const int c = rbv[i] > 0 ? p2_nt_cost(rbv[i], ksv[i]) : 0;
    Chi = c > Chi ? c : Chi;
  }
  // (one candidate per lane: the bisection's ~500 dependent integer divisions took 40 us as a scalar loop)
  for (int pass = 0; pass < 4 && Clo < Chi; ++pass) {
    const int span = Chi - Clo;
    const int C = Clo + (int)(((long)span * (lane + 1)) >> 6);          // lane 63 tests Chi (always feasible)
    int W = 0;
#pragma unroll
    for (int i = 0; i < LIREC_MAX_PROB; ++i) {
      if (rbv[i] == 0) continue;
      const int gm = p2_nt_gmax(C, ksv[i], rbv[i]);
      W += gm > 0 ? p2_div(rbv[i] + gm - 1, gm) * nrep : (1 << 20);
    }
    const unsigned long long ok = __ballot(W <= grid);
    const int f = ok ? __builtin_ctzll(ok) : 63;
    const int c_f = Clo + (int)(((long)span * (f + 1)) >> 6);
    const int c_prev = f > 0 ? Clo + (int)(((long)span * f) >> 6) : Clo - 1;
    Chi = __builtin_amdgcn_readfirstlane(c_f);
    Clo = __builtin_amdgcn_readfirstlane(c_prev + 1);
  }
  return Chi;
}

// The same on the host, for launches whose row counts are static (the gate GEMMs): the smallest feasible bound by bisection
// (feasibility is monotone in C), handed to the kernel by value -- no workgroup searches.
inline int p2_nt_bound_host(const int* rbv, const int* ksv, int nprob, int grid, int nrep) {
  auto need = [&](int C) {
    long W = 0;
    for (int i = 0; i < nprob; ++i) {
      if (rbv[i] == 0) continue;
      const int gm = p2_nt_gmax(C, ksv[i], rbv[i]);
      if (gm <= 0) return (long)1 << 40;
      W += (long)((rbv[i] + gm - 1) / gm) * nrep;
    }
    return W;
  };
  int lo = 1, hi = 1;
  for (int i = 0; i < nprob; ++i)
    if (rbv[i] > 0) { const int c = p2_nt_cost(rbv[i], ksv[i]); hi = c > hi ? c : hi; }
  if (need(hi) > grid) return 0;                    // (does not fit at all: the caller falls back)
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (need(mid) <= grid) hi = mid; else lo = mid + 1;
  }
  return hi;
}

}  // namespace lirec
