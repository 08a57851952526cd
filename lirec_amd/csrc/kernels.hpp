// HBM-bound kernels of the hot path: masked-mean pooling over context clips,
// fused loss forward+backward, fused Adam, dtype cast.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm.hpp"
#include "gemm_bf16x3.hpp"
#include "p2_partition.hpp"
#include "lirec_hip.h"

namespace lirec {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Deterministic block sum (fixed tree): every thread gets the total.  `red` >= 8 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// ---------------------------------------------------------------------------
// K3: masked mean over R context clips -> tanh -> dropout   (mlp/model.py:301-327)
// One workgroup per candidate c: the mask row is wave-uniform, so masked-out
// context rows are skipped without being read.  Every thread owns float4
// columns and keeps R independent 16-B loads in flight (HBM-bound pass:
// algorithmic bytes = n*R*W*4 read + n*W*4 written).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ Z2, long ldz,
                                                       const float* __restrict__ mask, int R, int W, int clamp_zero,
                                                       float* __restrict__ Tn, long ldtn, float* __restrict__ E, long lde,
                                                       unsigned seed_lo, unsigned seed_hi, const unsigned long long* __restrict__ seed_dev, unsigned site, unsigned thresh,
                                                       float drop_scale, int plain, float* __restrict__ fout) {
  // plain != 0: Tn <- the masked mean itself (no tanh, E unused), fout[c] <- (sum mask) / div
  apply_seed_offset(seed_lo, seed_hi, seed_dev);
  const int c = blockIdx.x;
  const float* mrow = mask + (long)c * R;
  float div = 0.f;
  for (int r = 0; r < R; ++r) div += mrow[r];
  const float cnt = div;
  if (clamp_zero && div == 0.f) div = 1.f;
  if (plain && fout && threadIdx.x == 0) fout[c] = cnt / div;
  const float* zc = Z2 + (long)c * R * ldz;
  const bool vec = ((W & 3) == 0) && ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(Z2) & 15) == 0) &&
                   ((ldtn & 3) == 0) && ((reinterpret_cast<uintptr_t>(Tn) & 15) == 0) &&
                   (plain || (((lde & 3) == 0) && ((reinterpret_cast<uintptr_t>(E) & 15) == 0)));
  if (vec) {
    for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < R; ++r) {
        const float m = mrow[r];                   // uniform -> scalar branch
        if (m != 0.f) {
          const f32x4 z = *reinterpret_cast<const f32x4*>(zc + (long)r * ldz + 4 * q);
          s.x += z.x * m; s.y += z.y * m; s.z += z.z * m; s.w += z.w * m;
        }
      }
      if (plain) {
        f32x4 o = {s.x / div, s.y / div, s.z / div, s.w / div};
        *reinterpret_cast<f32x4*>(Tn + (long)c * ldtn + 4 * q) = o;
        continue;
      }
      f32x4 t, e;
      float* tp = reinterpret_cast<float*>(&t);
      float* ep = reinterpret_cast<float*>(&e);
      const float* sp = reinterpret_cast<const float*>(&s);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float tv = tanhf(sp[j] / div);
        float ev = tv;
        if (thresh) {
          unsigned rnd[4];
          philox4((unsigned)(4 * q + j), (unsigned)(c >> 2), site, 0u, seed_lo, seed_hi, rnd);
          ev = (rnd[c & 3] >= thresh) ? tv * drop_scale : 0.f;
        }
        tp[j] = tv; ep[j] = ev;
      }
      *reinterpret_cast<f32x4*>(Tn + (long)c * ldtn + 4 * q) = t;
      *reinterpret_cast<f32x4*>(E + (long)c * lde + 4 * q) = e;
    }
  } else {
    for (int col = threadIdx.x; col < W; col += blockDim.x) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) {
        const float m = mrow[r];
        if (m != 0.f) s += zc[(long)r * ldz + col] * m;
      }
      if (plain) { Tn[(long)c * ldtn + col] = s / div; continue; }
      const float tv = tanhf(s / div);
      float ev = tv;
      if (thresh) {
        unsigned rnd[4];
        philox4((unsigned)col, (unsigned)(c >> 2), site, 0u, seed_lo, seed_hi, rnd);
        ev = (rnd[c & 3] >= thresh) ? tv * drop_scale : 0.f;
      }
      Tn[(long)c * ldtn + col] = tv;
      E[(long)c * lde + col] = ev;
    }
  }
}

// dZ2[c,r,:] = dP[c,:] * mask[c,r] / div[c]
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dP, long lddp,
                                                       const float* __restrict__ mask, int R, int W, int clamp_zero,
                                                       float* __restrict__ dZ2, long lddz) {
  const int c = blockIdx.x;
  const float* mrow = mask + (long)c * R;
  float div = 0.f;
  for (int r = 0; r < R; ++r) div += mrow[r];
  if (clamp_zero && div == 0.f) div = 1.f;
  float* zc = dZ2 + (long)c * R * lddz;
  const bool vec = ((W & 3) == 0) && ((lddz & 3) == 0) && ((lddp & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dZ2) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dP) & 15) == 0);
  if (vec) {
    for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dP + (long)c * lddp + 4 * q);
      for (int r = 0; r < R; ++r) {
        const float f = mrow[r] / div;
        f32x4 o = {d.x * f, d.y * f, d.z * f, d.w * f};
        *reinterpret_cast<f32x4*>(zc + (long)r * lddz + 4 * q) = o;
      }
    }
  } else {
    for (int col = threadIdx.x; col < W; col += blockDim.x) {
      const float d = dP[(long)c * lddp + col];
      for (int r = 0; r < R; ++r) zc[(long)r * lddz + col] = d * (mrow[r] / div);
    }
  }
}

// dZ1[c,r,:] = dHbar[c,:] * mask[c,r]/div[c] * [H1[c,r,:] > 0] * scale   (un-pool fused with the
// relu/dropout backward of the first layer; masked-out context rows are written as zeros without
// reading H1)
__global__ __launch_bounds__(256) void unpool_relu_kernel(const float* __restrict__ dHbar, long lddh,
                                                          const float* __restrict__ H1, long ldh,
                                                          const float* __restrict__ mask, int R, int W, int clamp_zero,
                                                          float scale, float* __restrict__ dZ1, long lddz) {
  const int c = blockIdx.x;
  const float* mrow = mask + (long)c * R;
  float div = 0.f;
  for (int r = 0; r < R; ++r) div += mrow[r];
  if (clamp_zero && div == 0.f) div = 1.f;
  const float* hc = H1 + (long)c * R * ldh;
  float* zc = dZ1 + (long)c * R * lddz;
  const bool vec = ((W & 3) == 0) && ((ldh & 3) == 0) && ((lddz & 3) == 0) && ((lddh & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(H1) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dZ1) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dHbar) & 15) == 0);
  if (vec) {
    for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dHbar + (long)c * lddh + 4 * q);
      for (int r = 0; r < R; ++r) {
        const float m = mrow[r];
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (m != 0.f) {
          const float f = m / div * scale;
          const f32x4 h = *reinterpret_cast<const f32x4*>(hc + (long)r * ldh + 4 * q);
          o.x = h.x > 0.f ? d.x * f : 0.f; o.y = h.y > 0.f ? d.y * f : 0.f;
          o.z = h.z > 0.f ? d.z * f : 0.f; o.w = h.w > 0.f ? d.w * f : 0.f;
        }
        *reinterpret_cast<f32x4*>(zc + (long)r * lddz + 4 * q) = o;
      }
    }
  } else {
    for (int col = threadIdx.x; col < W; col += blockDim.x) {
      const float d = dHbar[(long)c * lddh + col];
      for (int r = 0; r < R; ++r) {
        const float m = mrow[r];
        zc[(long)r * lddz + col] = (m != 0.f && hc[(long)r * ldh + col] > 0.f) ? d * (m / div * scale) : 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Row compaction of the context head.  A context row (c, r) with mask[c, r] == 0 has no influence on
// any output (the pooling multiplies it by 0 and its gradient is 0), so layer 1, the pooling pass,
// the un-pooling and the layer-1 weight gradient run on the valid rows only:
//   rowmap[j]  = c*R + r of the j-th valid row (ascending),   cstart[c] = first compact row of candidate c
//   count[0]   = number of valid rows (read by the GEMMs from device memory: no host sync)
// One workgroup, two passes over the n*R mask entries (<= a few 100 k): per-thread counts, scan, ordered write.
// ---------------------------------------------------------------------------
// mask entry e as a float, whatever the loader delivered: dtype 0 fp32, 1 int64 (rels_mask), 2 float64
__device__ __forceinline__ float mask_at(const void* mask, int dtype, long e) {
  if (dtype == 1) return (float)reinterpret_cast<const long long*>(mask)[e];
  if (dtype == 2) return (float)reinterpret_cast<const double*>(mask)[e];
  return reinterpret_cast<const float*>(mask)[e];
}

// Two launches, every candidate a wave (the single-workgroup versions spent 12-30 us on one CU):
//   count:  lane r of the candidate's wave reads mask[c, r]; ballot -> the candidate's valid-row bits and count
//   place:  exclusive prefix of the counts in front of the candidate (every lane sums a strided share of the count
//           array -- n is a few thousand -- then a wave sum), then lane r writes its row id / weight at
//           prefix + (valid rows in front of it); the last candidate's wave also writes cstart[n] and count[0].
// R <= 64 (one lane per context row); larger R takes compact_rows_serial_kernel.
__global__ __launch_bounds__(256) void compact_count_kernel(const void* __restrict__ mask, int dtype, int n, int R,
                                                            int* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n) return;
  const bool valid = lane < R && mask_at(mask, dtype, (long)c * R + lane) != 0.f;
  const unsigned long long bits = __ballot(valid);
  if (lane == 0) counts[c] = __builtin_popcountll(bits);
}

__global__ __launch_bounds__(256) void compact_place_kernel(const void* __restrict__ mask, int dtype, int n, int R,
                                                            const int* __restrict__ counts, int* __restrict__ rowmap,
                                                            int* __restrict__ cstart, int* __restrict__ count,
                                                            float* __restrict__ wts) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n) return;
  int part = 0;
  for (int i = lane; i < c; i += 64) part += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const float v = lane < R ? mask_at(mask, dtype, (long)c * R + lane) : 0.f;
  const unsigned long long bits = __ballot(v != 0.f);
  if (v != 0.f) {
    const int pos = part + __builtin_popcountll(bits & ((1ull << lane) - 1ull));
    rowmap[pos] = c * R + lane;
    if (wts) wts[pos] = v;
  }
  if (lane == 0) {
    cstart[c] = part;
    if (c == n - 1) { const int total = part + __builtin_popcountll(bits); cstart[n] = total; count[0] = total; }
  }
}

__global__ __launch_bounds__(1024) void compact_rows_serial_kernel(const void* __restrict__ mask, int dtype, int n, int R,
                                                            int* __restrict__ rowmap, int* __restrict__ cstart,
                                                            int* __restrict__ count, float* __restrict__ wts, int use_lds) {
  // One workgroup (the output is one ordered list).  With the mask staged in LDS (use_lds: n*R floats + n + 1 ints
  // fit) every global access is coalesced and nothing global is read twice:
  //   A  mask -> LDS as fp32, entry by entry (coalesced, whatever the loader's dtype);
  //   B  one candidate per thread (strided): its number of valid rows;
  //   C  exclusive scan over the candidates (contiguous runs per thread, wave shuffles + one LDS hop) -> cstart;
  //   D  one ENTRY per thread (strided): position = cstart[c] + valid entries before it in its candidate
  //      -> rowmap, wts (neighbouring threads write neighbouring positions).
  // Without LDS staging (very large n*R) the same steps read the mask from global memory.
  __shared__ int wsum[16];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  float* mval = reinterpret_cast<float*>(dyn_lds);                  // [n*R]   (use_lds)
  int* cpos = reinterpret_cast<int*>(mval + (long)n * R);           // [n + 1] (use_lds)
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const long total_e = (long)n * R;
  const bool staged = use_lds != 0;
  auto mv = [&](long e) -> float { return staged ? mval[e] : mask_at(mask, dtype, e); };
  if (staged) {
    // eight loads in flight per thread (a plain loop waited for every load before issuing the next: 18 round trips);
    // the dtype switch is outside the loops
    auto stage = [&](auto load) {
      for (long e0 = tid; e0 < total_e; e0 += 8L * nt) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long e = e0 + (long)u * nt; v[u] = load(e < total_e ? e : 0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long e = e0 + (long)u * nt; if (e < total_e) mval[e] = v[u]; }
      }
    };
    if (dtype == 1) stage([&](long e) { return (float)reinterpret_cast<const long long*>(mask)[e]; });
    else if (dtype == 2) stage([&](long e) { return (float)reinterpret_cast<const double*>(mask)[e]; });
    else stage([&](long e) { return reinterpret_cast<const float*>(mask)[e]; });
    __syncthreads();
  }
  // B + C: thread t owns candidates [c0, c1)
  const int per = (n + nt - 1) / nt;
  const int c0 = min(n, tid * per), c1 = min(n, c0 + per);
  int cnt = 0;
  for (long e = (long)c0 * R; e < (long)c1 * R; ++e) cnt += mv(e) != 0.f;
  int incl = cnt;                                   // inclusive scan inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < (nt >> 6); ++w) {
    const int v = wsum[w];
    if (w < wave) base += v;
    total += v;
  }
  int pos = base + incl - cnt;
  for (int c = c0; c < c1; ++c) {
    cstart[c] = pos;
    if (staged) {
      cpos[c] = pos;
      for (int r = 0; r < R; ++r) pos += mval[(long)c * R + r] != 0.f;
    } else {
      for (int r = 0; r < R; ++r) {
        const long e = (long)c * R + r;
        const float v = mask_at(mask, dtype, e);
        if (v != 0.f) {
          if (wts) wts[pos] = v;
          rowmap[pos++] = c * R + r;
        }
      }
    }
  }
  if (tid == 0) { cstart[n] = total; count[0] = total; }
  if (!staged) return;
  __syncthreads();
  // D
  for (long e = tid; e < total_e; e += nt) {
    const float v = mval[e];
    if (v == 0.f) continue;
    const int c = (int)(e / R), r = (int)(e - (long)c * R);
    int p = cpos[c];
    for (int q = 0; q < r; ++q) p += mval[(long)c * R + q] != 0.f;
    rowmap[p] = (int)e;
    if (wts) wts[p] = v;
  }
}

// ---------------------------------------------------------------------------
// The pooling pass, streaming form (the product path's K3; mlp/model.py:301-324 moved in front of layer 2).
// Work item = (candidate c, 256-column block): one WAVE owns it, lane l holds columns 4l..4l+3 of the block, so a
// row of the block is one 1-KiB wave load.  The candidate's row weights are fetched once (lane r holds the weight
// of row r, R <= 64), the non-zero ones are enumerated from a ballot, and the loads of up to 8 rows are all in
// flight before the first is consumed; 8 waves per SIMD-quad block x 8 blocks per CU keep >= 32 KiB per CU on the
// wire.  (The first version gave a candidate to a 256-thread workgroup that walked its rows one after another with
// two dependent index loads in front of every row: 0.32-0.38 of the HBM peak by the counters.)
// Rows are summed in ascending order, weights multiply before the sum and the quotient comes last: the same
// arithmetic, in the same order, as the per-candidate kernels above.
//   COMPACT: rows j in [cstart[c], cstart[c+1]) of H, weight mask[rowmap[j]];  else rows c*R + r, weight mask[c,r].
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// divider of a candidate from its row weights (lane r holds the weight of row r): summed in row order, as the
// per-candidate kernels do
__device__ __forceinline__ float row_weight_sum(float wr, int nrow) {
  float div = 0.f;
  for (int r = 0; r < nrow; ++r) div += lane_bcast(wr, r);
  return div;
}

// s += sum over the next N set bits r of `nz` (ascending) of w_r * row r; consumes the bits
// `bp` (optional): the SIGN BITS of the rows read -- for row r the 32 bytes bp[r * bstride ..], byte i = the nibbles of lanes
// 2 i (low) and 2 i + 1 (high), bit k of lane l's nibble = [H[r, col0 + 4 l + k] > 0] -- which is all the backward pass needs of
// H1 (unpool_rows_kernel<.., BITS>): 1 / 32 of its bytes, and H1 itself need not be kept from forward to backward.
template <int N, bool BITS = false>
__device__ __forceinline__ void pool_accum(const float* hp, long ldh, float wr, unsigned long long& nz, f32x4& s,
                                           unsigned char* bp = nullptr, int bstride = 0, bool colok = true, int lane = 0) {
  int rr[N];
  f32x4 z[N];
#pragma unroll
  for (int u = 0; u < N; ++u) { rr[u] = (int)__builtin_ctzll(nz); nz &= nz - 1; }
#pragma unroll
  for (int u = 0; u < N; ++u) z[u] = *reinterpret_cast<const f32x4*>(hp + (long)rr[u] * ldh);
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float m = lane_bcast(wr, rr[u]);
    s.x += z[u].x * m; s.y += z[u].y * m; s.z += z[u].z * m; s.w += z[u].w * m;
    if constexpr (BITS) {
      unsigned nib = (z[u].x > 0.f ? 1u : 0u) | (z[u].y > 0.f ? 2u : 0u) | (z[u].z > 0.f ? 4u : 0u) | (z[u].w > 0.f ? 8u : 0u);
      if (!colok) nib = 0u;
      const unsigned other = (unsigned)__shfl_xor((int)nib, 1);
      if (!(lane & 1)) bp[rr[u] * bstride] = (unsigned char)(nib | (other << 4));
    }
  }
}

template <bool COMPACT, bool BITS = false>
__global__ __launch_bounds__(256, 8) void pool_rows_kernel(const float* __restrict__ H, long ldh,
                                                           const float* __restrict__ mask, const int* __restrict__ rowmap,
                                                           const int* __restrict__ cstart, const float* __restrict__ wts,
                                                           int n, int R, int W, int clamp_zero,
                                                           float* __restrict__ Hbar, long ldo, float* __restrict__ fout,
                                                           unsigned char* __restrict__ hbits = nullptr) {
  const int lane = threadIdx.x & 63;
  const int ncb = (W + 255) >> 8;
  const long ntask = (long)n * ncb;
  for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
    const int c = (int)(task / ncb), cb = (int)(task - (long)c * ncb);
    int j0, nrow;
    if (COMPACT) { j0 = cstart[c]; nrow = cstart[c + 1] - j0; }
    else { j0 = c * R; nrow = R; }
    j0 = __builtin_amdgcn_readfirstlane(j0); nrow = __builtin_amdgcn_readfirstlane(nrow);
    float wr = 0.f;
    if (lane < nrow) wr = COMPACT ? (wts ? wts[j0 + lane] : mask[rowmap[j0 + lane]]) : mask[(long)j0 + lane];
    const int col = (cb << 8) + 4 * lane;
    const bool colok = col < W;
    const float* hp = H + (long)j0 * ldh + (colok ? col : 0);
    // (sign bits of H, optional: [row][column block][32 bytes] -- see pool_accum)
    const int bst = 32 * ncb;
    unsigned char* bp = BITS ? hbits + ((long)j0 * ncb + cb) * 32 + (lane >> 1) : nullptr;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // Rows to read.  COMPACT: every compact row has a non-zero weight by construction (lirec_compact_rows), so the
    // list is known as soon as cstart is -- the row loads do not wait for the rowmap -> mask chain that produces the
    // weights.  Dense: the rows with a non-zero weight (masked-out rows are never read).
    const unsigned long long all = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);
    unsigned long long nz = COMPACT ? all : __ballot(wr != 0.f);
    int left = __builtin_popcountll(nz);
    // straight-line groups of 8 / 4 / 2 / 1 rows: every load of a group is issued before the first is consumed,
    // and no load sits behind a per-row branch (a branch per load makes hipcc drain the memory counter each time)
    while (left >= 8) { pool_accum<8, BITS>(hp, ldh, wr, nz, s, bp, bst, colok, lane); left -= 8; }
    if (left & 4) pool_accum<4, BITS>(hp, ldh, wr, nz, s, bp, bst, colok, lane);
    if (left & 2) pool_accum<2, BITS>(hp, ldh, wr, nz, s, bp, bst, colok, lane);
    if (left & 1) pool_accum<1, BITS>(hp, ldh, wr, nz, s, bp, bst, colok, lane);
    float div = row_weight_sum(wr, nrow);
    const float cnt = div;
    if (clamp_zero && div == 0.f) div = 1.f;
    if (cb == 0 && lane == 0 && fout) fout[c] = cnt / div;
    if (colok) {
      const f32x4 o = {s.x / div, s.y / div, s.z / div, s.w / div};
      *reinterpret_cast<f32x4*>(Hbar + (long)c * ldo + col) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// Pre-split bf16 planes (the dZ1 operand of gemm_p2.hpp's weight-gradient kernel).  a = hi + lo with hi = bf16_rne(a), lo = bf16_rne(a - hi): the same
// split the on-the-fly core performs per k-tile (split4), done ONCE per operand and stored as two bf16 arrays.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, uint4& hi, uint4& lo) {
  uint2 h0, l0, h1, l1;
  split4(a, h0, l0);
  split4(b, h1, l1);
  hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
  lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// contiguous fp32 arrays -> planes (weights; small activations).  Every n is a multiple of 8, pointers 16-B aligned.
struct SplitSegs {
  const float* src[8]; unsigned short* hi[8]; unsigned short* lo[8];
  long first[9];                              // prefix sums of n / 8
  int nseg;
};
__device__ __forceinline__ void split_planes(const SplitSegs& q, const int block, const int nblocks) {
  const long total = q.first[q.nseg];
  for (long i = (long)block * blockDim.x + threadIdx.x; i < total; i += (long)nblocks * blockDim.x) {
    int sgi = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) if (j < q.nseg && i >= q.first[j]) sgi = j;
    const long e = (i - q.first[sgi]) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(q.src[sgi] + e), b = *reinterpret_cast<const f32x4*>(q.src[sgi] + e + 4);
    uint4 hi, lo;
    split8(a, b, hi, lo);
    *reinterpret_cast<uint4*>(q.hi[sgi] + e) = hi;
    *reinterpret_cast<uint4*>(q.lo[sgi] + e) = lo;
  }
}
__global__ __launch_bounds__(256) void split_planes_kernel(const SplitSegs q) { split_planes(q, blockIdx.x, gridDim.x); }

// contiguous fp32 matrices [R][C] (R, C multiples of 32) -> q32b (first-layer weights, once per step)
struct SplitQ32b {
  const float* src[8]; unsigned char* dst[8]; int cols[8];
  long first[9];                              // prefix sums of R * C / 8
  int nseg;
  int fmt16c;                                 // the destinations are q16c (single-pass mode: bf16 values, 64-column blocks; C % 64 == 0)
};
__device__ __forceinline__ void split_q32b(const SplitQ32b& q, const int block, const int nblocks);
__global__ __launch_bounds__(256) void split_q32b_kernel(const SplitQ32b q);
__device__ __forceinline__ void split_q32b(const SplitQ32b& q, const int block, const int nblocks) {
  const long total = q.first[q.nseg];
  for (long i = (long)block * blockDim.x + threadIdx.x; i < total; i += (long)nblocks * blockDim.x) {
    int sgi = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) if (j < q.nseg && i >= q.first[j]) sgi = j;
    const long e8 = i - q.first[sgi];
    const int c8n = q.cols[sgi] >> 3;
    const long row = e8 / c8n;
    const int c8 = (int)(e8 - row * c8n);
    const float* src = q.src[sgi] + 8 * e8;
    if (q.fmt16c) p2_store_q16c(q.dst[sgi], row, c8, q.cols[sgi] >> 6, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    else p2_store_q32b(q.dst[sgi], row, c8, q.cols[sgi] >> 5, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
  }
}
// the next N set bits r of `nz` (ascending): dZ1 row r = d * (w_r / div * scale) * [H1 row r > 0]; consumes the bits.
// The divider is formed AFTER the row loads have been issued (it waits for the weights, the loads do not).
// BITS: the decisions [H1 > 0] come from the sign bits the pooling pass left (pool_accum): `hp` then points at this task's first
// byte (row j0, column block cb) and `ldh` is the row stride in bytes.
template <int N, int PLANES, bool BITS>
__device__ __forceinline__ void unpool_rows(const float* hp, long ldh, float* zp, long lddz, float wr, int nrow, int clamp_zero,
                                            float scale, const f32x4 d, unsigned long long& nz, bool colok, long lo_off, int lane,
                                            int zcol = 0) {
  int rr[N];
  f32x4 h[N];
  unsigned hb[N];
#pragma unroll
  for (int u = 0; u < N; ++u) { rr[u] = (int)__builtin_ctzll(nz); nz &= nz - 1; }
  if constexpr (BITS) {
    const unsigned char* bp = reinterpret_cast<const unsigned char*>(hp) + (lane >> 1);
#pragma unroll
    for (int u = 0; u < N; ++u) hb[u] = bp[(long)rr[u] * ldh];
  } else {
#pragma unroll
    for (int u = 0; u < N; ++u) h[u] = *reinterpret_cast<const f32x4*>(hp + (long)rr[u] * ldh);
  }
  float div = row_weight_sum(wr, nrow);
  if (clamp_zero && div == 0.f) div = 1.f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float f = lane_bcast(wr, rr[u]) / div * scale;
    f32x4 o;
    if constexpr (BITS) {
      const unsigned nib = hb[u] >> (4 * (lane & 1));
      o.x = (nib & 1u) ? d.x * f : 0.f; o.y = (nib & 2u) ? d.y * f : 0.f;
      o.z = (nib & 4u) ? d.z * f : 0.f; o.w = (nib & 8u) ? d.w * f : 0.f;
    } else {
      o.x = h[u].x > 0.f ? d.x * f : 0.f; o.y = h[u].y > 0.f ? d.y * f : 0.f;
      o.z = h[u].z > 0.f ? d.z * f : 0.f; o.w = h[u].w > 0.f ? d.w * f : 0.f;
    }
    if constexpr (PLANES == 2) {
      // q32b rows (gemm_p3's weight gradient reads them k-major): zp = the matrix, lddz = its columns, lo_off = compact row j0 of
      // this candidate, `lane`'s four columns start at column 4 (lane + 64 cb') -- passed in as zcol
      uint2 h2, l2;
      split4(o, h2, l2);
      const long row = lo_off + rr[u];
      unsigned char* q = reinterpret_cast<unsigned char*>(zp) + (((row >> 5) * (lddz >> 5) + (zcol >> 5)) * 32 + (row & 31)) * 128 + (zcol & 31) * 2;
      if (colok) { *reinterpret_cast<uint2*>(q) = h2; *reinterpret_cast<uint2*>(q + 64) = l2; }
    } else if constexpr (PLANES == 1) {
      // zp / lddz address the hi plane in bf16 ELEMENTS (zp already at this lane's 4 columns); lo plane at + lo_off
      uint2 h2, l2;
      split4(o, h2, l2);
      unsigned short* zh = reinterpret_cast<unsigned short*>(zp) + (long)rr[u] * lddz;
      // (lo_off = 0: the hi plane alone -- the single-pass weight gradient reads nothing else)
      if (colok) { *reinterpret_cast<uint2*>(zh) = h2; if (lo_off) *reinterpret_cast<uint2*>(zh + lo_off) = l2; }
    } else {
      if (colok) *reinterpret_cast<f32x4*>(zp + (long)rr[u] * lddz) = o;
    }
  }
}

// Un-pooling fused with the relu/dropout backward of layer 1, same work split:
//   dZ1[j,:] = dHbar[c,:] * (m_j / div * scale) * [H1[j,:] > 0]
// COMPACT: rows [cstart[c], cstart[c+1]) (all written).  Dense: rows c*R + r; rows with a zero weight are written
// as zeros without reading H1.
// PLANES: dZ1 is written as pre-split bf16 planes for the weight-gradient GEMM on q32b operands (gemm_p2.hpp): `dZ1` is then
// the hi plane (bf16, lddz in elements), the lo plane lies lo_off elements behind it (0: no lo plane is written), and the rows
// [*count, roundup(*count, 32)) are written as zeros (that GEMM reduces over the rows in whole 32-row k-tiles).
// BITS: `H1` is not read -- it points at the sign bits of H1 written by pool_rows_kernel ([row][column block][32 bytes]).
// PLANES = 2: dZ1 is written as q32b rows (`dZ1` = the matrix, lddz = its columns; for gemm_p3's weight gradient), tail rows zeroed
// likewise; `sq32` / split_blocks then name another head's fp32 dZ1 to be turned into q32b by the first workgroups.
template <bool COMPACT, int PLANES = 0, bool BITS = false>
__global__ __launch_bounds__(256, 8) void unpool_rows_kernel(const float* __restrict__ dHbar, long lddh,
                                                             const float* __restrict__ H1, long ldh,
                                                             const float* __restrict__ mask, const int* __restrict__ rowmap,
                                                             const int* __restrict__ cstart, const float* __restrict__ wts,
                                                             int n, int R, int W,
                                                             int clamp_zero, float scale, float* __restrict__ dZ1, long lddz,
                                                             long lo_off, const int* __restrict__ count,
                                                             const SplitSegs sq = SplitSegs(), const int split_blocks = 0,
                                                             const SplitQ32b sq32 = SplitQ32b()) {
  // (the first `split_blocks` workgroups split another head's fp32 dZ1 into planes / q32b -- a 6 us launch of its own otherwise)
  if ((int)blockIdx.x < split_blocks) {
    if constexpr (PLANES == 2) split_q32b(sq32, blockIdx.x, split_blocks); else split_planes(sq, blockIdx.x, split_blocks);
    return;
  }
  const int block = blockIdx.x - split_blocks, nblocks = gridDim.x - split_blocks;
  const int lane = threadIdx.x & 63;
  const int ncb = (W + 255) >> 8;
  const long ntask = (long)n * ncb;
  if constexpr (PLANES) {
    // zero tail: the first workgroups clear the (at most 31) rows behind the last valid one
    const int valid = count ? *count : n * R;
    const int upto = (valid + 31) & ~31;
    const long ztask = (long)(upto - valid) * ncb;
    for (long zt = (long)block * 4 + (threadIdx.x >> 6); zt < ztask; zt += (long)nblocks * 4) {
      const int row = valid + (int)(zt / ncb), col = ((int)(zt % ncb) << 8) + 4 * lane;
      if (col < W) {
        if constexpr (PLANES == 2) {
          unsigned char* q = reinterpret_cast<unsigned char*>(dZ1) + ((((long)row >> 5) * (lddz >> 5) + (col >> 5)) * 32 + (row & 31)) * 128 + (col & 31) * 2;
          *reinterpret_cast<uint2*>(q) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(q + 64) = make_uint2(0u, 0u);
        } else {
          unsigned short* zh = reinterpret_cast<unsigned short*>(dZ1) + (long)row * lddz + col;
          *reinterpret_cast<uint2*>(zh) = make_uint2(0u, 0u);
          if (lo_off) *reinterpret_cast<uint2*>(zh + lo_off) = make_uint2(0u, 0u);
        }
      }
    }
  }
  for (long task = (long)block * 4 + (threadIdx.x >> 6); task < ntask; task += (long)nblocks * 4) {
    const int c = (int)(task / ncb), cb = (int)(task - (long)c * ncb);
    int j0, nrow;
    if (COMPACT) { j0 = cstart[c]; nrow = cstart[c + 1] - j0; }
    else { j0 = c * R; nrow = R; }
    j0 = __builtin_amdgcn_readfirstlane(j0); nrow = __builtin_amdgcn_readfirstlane(nrow);
    float wr = 0.f;
    if (lane < nrow) wr = COMPACT ? (wts ? wts[j0 + lane] : mask[rowmap[j0 + lane]]) : mask[(long)j0 + lane];
    // (no early exit for the lanes beyond the last column: the ballot below needs every lane of the wave; they read
    //  column 0 instead and their stores are switched off)
    const int col = (cb << 8) + 4 * lane;
    const bool colok = col < W;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dHbar + (long)c * lddh + (colok ? col : 0));
    const float* hp = BITS ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(H1) + ((long)j0 * ncb + cb) * 32)
                           : H1 + (long)j0 * ldh + (colok ? col : 0);
    const long ldq = BITS ? 32L * ncb : ldh;
    float* zp = PLANES == 2 ? dZ1
              : (PLANES == 1 ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(dZ1) + (long)j0 * lddz + col)
                             : dZ1 + (long)j0 * lddz + col);
    const long lo_arg = PLANES == 2 ? (long)j0 : lo_off;
    const unsigned long long all = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);
    unsigned long long nz = COMPACT ? all : __ballot(wr != 0.f);
    unsigned long long zr = ~nz & all;
    int left = __builtin_popcountll(nz);
    while (left >= 8) { unpool_rows<8, PLANES, BITS>(hp, ldq, zp, lddz, wr, nrow, clamp_zero, scale, d, nz, colok, lo_arg, lane, col); left -= 8; }
    if (left & 4) unpool_rows<4, PLANES, BITS>(hp, ldq, zp, lddz, wr, nrow, clamp_zero, scale, d, nz, colok, lo_arg, lane, col);
    if (left & 2) unpool_rows<2, PLANES, BITS>(hp, ldq, zp, lddz, wr, nrow, clamp_zero, scale, d, nz, colok, lo_arg, lane, col);
    if (left & 1) unpool_rows<1, PLANES, BITS>(hp, ldq, zp, lddz, wr, nrow, clamp_zero, scale, d, nz, colok, lo_arg, lane, col);
    while (zr) {                                     // dense form only: masked-out rows are zeros, H1 is not read
      const int r = (int)__builtin_ctzll(zr); zr &= zr - 1;
      if constexpr (PLANES == 2) {
        const long row = (long)j0 + r;
        unsigned char* q = reinterpret_cast<unsigned char*>(dZ1) + (((row >> 5) * (lddz >> 5) + (col >> 5)) * 32 + (row & 31)) * 128 + (col & 31) * 2;
        if (colok) { *reinterpret_cast<uint2*>(q) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(q + 64) = make_uint2(0u, 0u); }
      } else if constexpr (PLANES == 1) {
        unsigned short* zh = reinterpret_cast<unsigned short*>(zp) + (long)r * lddz;
        if (colok) { *reinterpret_cast<uint2*>(zh) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(zh + lo_off) = make_uint2(0u, 0u); }
      } else {
        if (colok) *reinterpret_cast<f32x4*>(zp + (long)r * lddz) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

// masked mean over the COMPACT rows of candidate c: Hbar[c,:] = sum_j m_j H[j,:] / div, f[c] = cnt/div
// (any alignment / width; the aligned case takes pool_rows_kernel)
__global__ __launch_bounds__(256) void pool_compact_kernel(const float* __restrict__ H, long ldh,
                                                           const float* __restrict__ mask, const int* __restrict__ rowmap,
                                                           const int* __restrict__ cstart, const float* __restrict__ wts,
                                                           int R, int W, int clamp_zero,
                                                           float* __restrict__ Hbar, long ldo, float* __restrict__ fout) {
  const int c = blockIdx.x;
  const int j0 = cstart[c], j1 = cstart[c + 1];
  float div = 0.f;
  for (int j = j0; j < j1; ++j) div += (wts ? wts[j] : mask[rowmap[j]]);
  const float cnt = div;
  if (clamp_zero && div == 0.f) div = 1.f;
  if (fout && threadIdx.x == 0) fout[c] = cnt / div;
  const bool vec = ((W & 3) == 0) && ((ldh & 3) == 0) && ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(H) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Hbar) & 15) == 0);
  if (vec) {
    for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int j = j0; j < j1; ++j) {
        const float m = (wts ? wts[j] : mask[rowmap[j]]);
        const f32x4 z = *reinterpret_cast<const f32x4*>(H + (long)j * ldh + 4 * q);
        s.x += z.x * m; s.y += z.y * m; s.z += z.z * m; s.w += z.w * m;
      }
      const f32x4 o = {s.x / div, s.y / div, s.z / div, s.w / div};
      *reinterpret_cast<f32x4*>(Hbar + (long)c * ldo + 4 * q) = o;
    }
  } else {
    for (int col = threadIdx.x; col < W; col += blockDim.x) {
      float s = 0.f;
      for (int j = j0; j < j1; ++j) s += H[(long)j * ldh + col] * (wts ? wts[j] : mask[rowmap[j]]);
      Hbar[(long)c * ldo + col] = s / div;
    }
  }
}

// dZ1[j,:] = dHbar[c,:] * m_j/div * [H1[j,:] > 0] * scale over the compact rows j of candidate c
__global__ __launch_bounds__(256) void unpool_relu_compact_kernel(const float* __restrict__ dHbar, long lddh,
                                                                  const float* __restrict__ H1, long ldh,
                                                                  const float* __restrict__ mask,
                                                                  const int* __restrict__ rowmap, const int* __restrict__ cstart,
                                                                  const float* __restrict__ wts,
                                                                  int W, int clamp_zero, float scale,
                                                                  float* __restrict__ dZ1, long lddz) {
  const int c = blockIdx.x;
  const int j0 = cstart[c], j1 = cstart[c + 1];
  float div = 0.f;
  for (int j = j0; j < j1; ++j) div += (wts ? wts[j] : mask[rowmap[j]]);
  if (clamp_zero && div == 0.f) div = 1.f;
  const bool vec = ((W & 3) == 0) && ((ldh & 3) == 0) && ((lddz & 3) == 0) && ((lddh & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(H1) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dZ1) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dHbar) & 15) == 0);
  if (vec) {
    for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dHbar + (long)c * lddh + 4 * q);
      for (int j = j0; j < j1; ++j) {
        const float f = (wts ? wts[j] : mask[rowmap[j]]) / div * scale;
        const f32x4 h = *reinterpret_cast<const f32x4*>(H1 + (long)j * ldh + 4 * q);
        f32x4 o;
        o.x = h.x > 0.f ? d.x * f : 0.f; o.y = h.y > 0.f ? d.y * f : 0.f;
        o.z = h.z > 0.f ? d.z * f : 0.f; o.w = h.w > 0.f ? d.w * f : 0.f;
        *reinterpret_cast<f32x4*>(dZ1 + (long)j * lddz + 4 * q) = o;
      }
    }
  } else {
    for (int col = threadIdx.x; col < W; col += blockDim.x) {
      const float d = dHbar[(long)c * lddh + col];
      for (int j = j0; j < j1; ++j)
        dZ1[(long)j * lddz + col] = (H1[(long)j * ldh + col] > 0.f) ? d * ((wts ? wts[j] : mask[rowmap[j]]) / div * scale) : 0.f;
    }
  }
}

// q32b staging for the layer-1 kernels (gemm_p2.hpp): the selected feature rows (compact row j = logical
// row rowmap[j] -> physical row of the (B*T, R+1, D) block) as a blocked hi / lo matrix [rows32][D8 * 8 columns].  Rows
// [*count, roundup(*count, 32)) are written as zeros.  One thread per (row, 8 columns): 32 B in, 2 x 16 B out; the 4 threads
// of a row's 32-column block write its whole 128-B line.
// Dropout keep bits of layer 1 (optional, `keep` != NULL): this pass is HBM-bound with idle vector ALUs, the GEMM that
// consumes the bits (gemm_p2.hpp) has one workgroup per CU and nothing to hide its epilogue behind, so the Philox words of
// H1's dropout are produced HERE: task (q, c) = rows 4 q .. 4 q + 3 of the operand (compact), column c of the head's
// `ncol` hidden columns -> one byte, bit j = (word of row 4 q + j, counter = ORIGINAL row id) >= thresh.
struct StageDrop {
  unsigned char* keep; long ld; int ncol;
  unsigned seed_lo, seed_hi; const unsigned long long* seed_dev; unsigned site, thresh;
};
// Source of the rows when they are given as piece tables + index (lirec_embed_fwd_args::pieces) instead of a block: physical
// row r = hstack(clip[index[3 r]], track[index[3 r + 1]], track[index[3 r + 2]]), a negative index = zeros
// (mixed_utils/classification_dataloader.py:336-349, :477-478).  index == nullptr: rows come from X.
// GATHER mode (srow[0] != nullptr): nothing is copied -- the GEMMs fetch their rows from a q32b matrix through a row list (GemmProblem::
// srow) -- and this pass only WRITES the lists, entries [0, roundup32(valid)) with the tail repeating the last valid row:
//   rows of a q32b block (index == nullptr):  srow[0][j] = physical row of compact row j;
//   rows as pieces: srow[0] = clip-table rows, srow[1] / srow[2] = track-table rows of track 1 / 2; an index value v >= 0 names
//   table row v, or rows[v] when a second-level list is given (resident store); v < 0 names the zero row (zero_clip / zero_track).
struct StageSrc {
  const float* clip; const float* track; const int* index; long ld_clip, ld_track; int clip_dim, track_dim, c0;
  int* srow[3]; const int* clip_rows; const int* track_rows; int zero_clip, zero_track;
  int x16;      // the block at X is bf16 (row-major, ldx in elements): its rows are staged as q16b -- ONE plane, 64-byte rows -- and
                // srow[0] gets the identity list the one-plane kernels read them through; 2 = staged as q16c (single-pass mode)
};
// (block = this role's workgroup index, nblocks = how many workgroups the role has: the roles of several row sets and of the
//  weight split share ONE launch, stage_fused_kernel below)
__device__ __forceinline__ void stage_rows_q32b(const float* __restrict__ X, long ldx, int gs, int gstride, int goff,
                                                const int* __restrict__ rowmap, const int* __restrict__ count,
                                                int rows, int D8, unsigned char* __restrict__ dst, const StageDrop& dk,
                                                const int block, const int nblocks, const StageSrc& src = StageSrc()) {
  const int valid = count ? min(*count, rows) : rows;
  const int upto = min((valid + 31) & ~31, (rows + 31) & ~31);
  const long total = (long)upto * D8;
  // The mask tasks and the staging items are dealt to DIFFERENT workgroups of this launch (every third workgroup takes
  // tasks when there are any): the CUs then hold memory-bound and ALU-bound waves side by side -- the staging items wait on
  // HBM, a task is ~300 vector instructions -- and the masks cost what is left of the staging time's idle ALU cycles.
  // A task = four rows x FOUR columns (one 32-bit store, the row ids fetched once).
  const bool masks = dk.keep != nullptr && dk.thresh != 0u;
  const int nc4 = dk.ncol >> 2;
  const int tasks = masks ? ((valid + 3) >> 2) * nc4 : 0;
  const bool split = masks && nblocks >= 3;
  // (GATHER mode: the lists are a few thousand ints -- the first 8 workgroups write them, every other one takes mask tasks)
  const bool gath = src.srow[0] != nullptr && !src.x16;
  const int role_mask = split && (gath ? block >= 8 : block % 3 == 2);
  const int nb_mask = split ? (gath ? nblocks - 8 : nblocks / 3) : 0, nb_stage = nblocks - nb_mask;
  const int bi = split ? (gath ? (role_mask ? block - 8 : block) : (role_mask ? block / 3 : block - block / 3)) : block;
  if (role_mask || (masks && !split)) {
    unsigned key_lo = dk.seed_lo, key_hi = dk.seed_hi;
    apply_seed_offset(key_lo, key_hi, dk.seed_dev);
    const int nb = split ? nb_mask : nblocks;
    for (int tk = bi * blockDim.x + threadIdx.x; tk < tasks; tk += nb * blockDim.x) {
      const int q = tk / nc4, c0 = 4 * (tk - q * nc4);
      unsigned rid[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j < valid ? 4 * q + j : valid - 1;
        rid[j] = (unsigned)(rowmap ? rowmap[r] : r);
      }
      unsigned word = 0u;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        unsigned rnd[4];
        unsigned blk = rid[0] >> 2, bits = 0u;
        philox4((unsigned)(c0 + cc), blk, dk.site, 0u, key_lo, key_hi, rnd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((rid[j] >> 2) != blk) {
            blk = rid[j] >> 2;
            philox4((unsigned)(c0 + cc), blk, dk.site, 0u, key_lo, key_hi, rnd);
          }
          const unsigned k = rid[j] & 3u;
          const unsigned w = k == 0u ? rnd[0] : (k == 1u ? rnd[1] : (k == 2u ? rnd[2] : rnd[3]));
          bits |= (w >= dk.thresh ? 1u : 0u) << j;
        }
        word |= bits << (8 * cc);
      }
      *reinterpret_cast<unsigned*>(dk.keep + (long)q * dk.ld + c0) = word;
    }
    if (role_mask) return;
  }
  if (src.x16) {
    // a bf16 block: the identity list (a few thousand ints, every staging workgroup a share), then the rows as q16b below
    for (int i = bi * blockDim.x + threadIdx.x; i < upto; i += nb_stage * blockDim.x) src.srow[0][i] = i;
  } else if (src.srow[0]) {
    // GATHER mode: the row lists instead of the rows (a few thousand ints)
    const int nlist = src.index ? 3 : 1;
    for (int i = bi * blockDim.x + threadIdx.x; i < upto * nlist; i += nb_stage * blockDim.x) {
      const int part = i / upto, j = i - part * upto;
      const int jj = j < valid ? j : valid - 1;
      int out = 0;
      if (jj >= 0) {
        const int rid = rowmap ? rowmap[jj] : jj;
        long prow = rid;
        if (gs != 0) { const int qd = rid / gs; prow = (long)qd * gstride + (rid - qd * gs) + goff; }
        if (src.index) {
          const int v = src.index[3 * prow + part];
          const int* lst = part == 0 ? src.clip_rows : src.track_rows;
          out = v < 0 ? (part == 0 ? src.zero_clip : src.zero_track) : (lst ? lst[v] : v);
        } else {
          out = (int)prow;
        }
      }
      src.srow[part][j] = out;
    }
    return;
  }
  if (src.x16) {
    for (long i = (long)bi * blockDim.x + threadIdx.x; i < total; i += (long)nb_stage * blockDim.x) {
      const int j = (int)(i / D8), c8 = (int)(i - (long)j * D8);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                           // (eight bf16 values, moved as 16 bytes)
      if (j < valid) {
        const int rid = rowmap ? rowmap[j] : j;
        long prow = rid;
        if (gs != 0) { const int qd = rid / gs; prow = (long)qd * gstride + (rid - qd * gs) + goff; }
        v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(X) + prow * ldx + 8 * c8));
      }
      if (src.x16 == 2) *reinterpret_cast<f32x4*>(dst + p2_q16c_off(j, c8, D8 >> 3)) = v;
      else *reinterpret_cast<f32x4*>(dst + ((((long)(j >> 5)) * (D8 >> 2) + (c8 >> 2)) * 32 + (j & 31)) * 64 + (c8 & 3) * 16) = v;
    }
    return;
  }
  for (long i = (long)bi * blockDim.x + threadIdx.x; i < total; i += (long)nb_stage * blockDim.x) {
    const int j = (int)(i / D8), c8 = (int)(i - (long)j * D8);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (j < valid) {
      const int rid = rowmap ? rowmap[j] : j;
      long prow = rid;
      if (gs != 0) { const int qd = rid / gs; prow = (long)qd * gstride + (rid - qd * gs) + goff; }
      if (src.index) {
        // eight consecutive columns lie in one piece (clip_dim, track_dim are multiples of 8)
        const int col = src.c0 + 8 * c8;
        const int part = col < src.clip_dim ? 0 : (col < src.clip_dim + src.track_dim ? 1 : 2);
        const int pid = src.index[3 * prow + part];
        if (pid >= 0) {
          const float* q = part == 0 ? src.clip + (long)pid * src.ld_clip + col
                                     : src.track + (long)pid * src.ld_track + (col - src.clip_dim - (part - 1) * src.track_dim);
          a = *reinterpret_cast<const f32x4*>(q); b = *reinterpret_cast<const f32x4*>(q + 4);
        }
      } else {
        // (read once: non-temporal, so that the 228 MB of rows do not push the staged copy -- which layer 1 reads next -- out of
        //  the Infinity Cache)
        const float* q = X + prow * ldx + 8 * c8;
        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)); b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + 4));
      }
    }
    p2_store_q32b(dst, j, c8, D8 >> 2, a, b);
  }
}
__global__ __launch_bounds__(256) void stage_rows_q32b_kernel(const float* __restrict__ X, long ldx, int gs, int gstride, int goff,
                                                              const int* __restrict__ rowmap, const int* __restrict__ count,
                                                              int rows, int D8, unsigned char* __restrict__ dst, const StageDrop dk) {
  stage_rows_q32b(X, ldx, gs, gstride, goff, rowmap, count, rows, D8, dst, dk, blockIdx.x, gridDim.x);
}
// fp32 [rows][cols] (row stride ld) -> q32b with the rows padded to rows32 by zero rows (feature storage: piece tables, blocks)
__global__ __launch_bounds__(256) void to_q32b_kernel(const float* __restrict__ src, long ld, long rows, long rows32, int c8n,
                                                      unsigned char* __restrict__ dst) {
  const long total = rows32 * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / c8n;
    const int c8 = (int)(i - row * c8n);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < rows) { const float* q = src + row * ld + 8 * c8; a = *reinterpret_cast<const f32x4*>(q); b = *reinterpret_cast<const f32x4*>(q + 4); }
    p2_store_q32b(dst, row, c8, c8n >> 2, a, b);
  }
}
// fp32 [rows][cols] -> q16b (bf16 round-to-nearest-even of every value, blocked: gemm_bf16x3.hpp), rows padded to rows32 by zeros
// (c64: as q16c -- 64-column blocks, the single-pass mode's storage)
__global__ __launch_bounds__(256) void to_q16b_kernel(const float* __restrict__ src, long ld, long rows, long rows32, int c8n,
                                                      unsigned char* __restrict__ dst, int c64) {
  const long total = rows32 * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / c8n;
    const int c8 = (int)(i - row * c8n);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < rows) { const float* q = src + row * ld + 8 * c8; a = *reinterpret_cast<const f32x4*>(q); b = *reinterpret_cast<const f32x4*>(q + 4); }
    if (c64) p2_store_q16c(dst, row, c8, c8n >> 3, a, b); else p2_store_q16b(dst, row, c8, c8n >> 2, a, b);
  }
}
// Everything layer 1 needs staged, in ONE launch: the feature rows of up to two heads (+ the dropout keep bytes of each) and the
// first-layer weights.  As three launches the two small ones (weights 15 us, the interaction head's 1024 rows 19 us) were pure
// latency in front of the context head's HBM-bound 83 us; as roles of one grid they run in its shadow.  Workgroups are dealt
// head by head (the caller lists the larger head first), the weight split last.
__global__ __launch_bounds__(256) void split_q32b_kernel(const SplitQ32b q) { split_q32b(q, blockIdx.x, gridDim.x); }

// The same, and the q32b form of the TRANSPOSE beside it (the gate GEMMs, gemm_p3.hpp: every operand as k-contiguous rows -- the
// data gradient multiplies by Wg^T, the weight gradient reduces over the rows of dZg and EE).  One workgroup per 32 x 32 block:
// thread (r, c4) holds four consecutive columns of row r -- the normal form goes out from registers, the transposed one through
// a padded LDS tile (thread (c, q4) = output row c, source rows 4 q4 .. 4 q4 + 3).  dstT = NULL: the normal form only.
struct SplitDual {
  const float* src[4]; unsigned char* dst[4]; unsigned char* dstT[4]; int rows[4], cols[4];
  long first[5];                              // prefix sums of the segments' block counts
  int nseg;
  int fmt16c;                                 // single-pass mode: both forms as q16c (bf16 values, 64-column blocks; rows, cols % 64 == 0)
};
__global__ __launch_bounds__(256) void split_q32b_dual_kernel(const SplitDual q) {
  __shared__ float t[32][33];
  const int tid = threadIdx.x, r = tid >> 3, c4 = tid & 7;
  for (long blk = blockIdx.x; blk < q.first[q.nseg]; blk += gridDim.x) {
    int sg = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) if (j < q.nseg && blk >= q.first[j]) sg = j;
    const int cbn = q.cols[sg] >> 5, rbn = q.rows[sg] >> 5;
    const int bi = (int)(blk - q.first[sg]);
    const int rb = bi / cbn, cb = bi - rb * cbn;
    const f32x4 v = *reinterpret_cast<const f32x4*>(q.src[sg] + (long)(32 * rb + r) * q.cols[sg] + 32 * cb + 4 * c4);
    uint2 h, l;
    if (q.fmt16c) {
      // (block (rb, cb) of 32 x 32 = half cb & 1 of the 64-column block cb >> 1: 64 bytes of the row's 128)
      *reinterpret_cast<uint2*>(q.dst[sg] + (((long)rb * (cbn >> 1) + (cb >> 1)) * 32 + r) * 128 + (cb & 1) * 64 + c4 * 8) = hi4(v);
    } else {
    split4(v, h, l);
    unsigned char* d = q.dst[sg] + (((long)rb * cbn + cb) * 32 + r) * 128 + c4 * 8;
    *reinterpret_cast<uint2*>(d) = h;
    *reinterpret_cast<uint2*>(d + 64) = l;
    }
    if (q.dstT[sg]) {
      t[r][4 * c4 + 0] = v.x; t[r][4 * c4 + 1] = v.y; t[r][4 * c4 + 2] = v.z; t[r][4 * c4 + 3] = v.w;
      __syncthreads();
      f32x4 w;
      w.x = t[4 * c4 + 0][r]; w.y = t[4 * c4 + 1][r]; w.z = t[4 * c4 + 2][r]; w.w = t[4 * c4 + 3][r];
      if (q.fmt16c) {
        *reinterpret_cast<uint2*>(q.dstT[sg] + (((long)cb * (rbn >> 1) + (rb >> 1)) * 32 + r) * 128 + (rb & 1) * 64 + c4 * 8) = hi4(w);
      } else {
      split4(w, h, l);
      unsigned char* dt = q.dstT[sg] + (((long)cb * rbn + rb) * 32 + r) * 128 + c4 * 8;      // (here r = the source COLUMN in the block)
      *reinterpret_cast<uint2*>(dt) = h;
      *reinterpret_cast<uint2*>(dt + 64) = l;
      }
      __syncthreads();
    }
  }
}
// diagnostics (lirec_debug_set bit 131072): one wave that does nothing for `ticks` of the 100 MHz clock -- a stream made to lag on
// purpose, in front of the launch it precedes (tests of what a replayed step may leave running on its side stream)
__global__ void spin_kernel(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
struct StageHead {
  const float* X; long ldx; int gs, gstride, goff; const int* rowmap; const int* count; int rows, D8; unsigned char* dst;
  StageDrop dk; int blocks; StageSrc src;
};
// (+ one workgroup that runs the forward GEMM's partition search on the same row counts: p2_partition.hpp)
struct StagePart { int n, grid, nrep; int ks[LIREC_MAX_PROB], rows[LIREC_MAX_PROB]; const int* dyn[LIREC_MAX_PROB]; int* out; };
struct StageFused { StageHead h[2]; int nh; SplitQ32b w; int w_blocks; StagePart part; };
__global__ __launch_bounds__(256) void stage_fused_kernel(const StageFused f) {
  int b = blockIdx.x;
  if (b == 0) {                               // (first, so that its ~15 us of search run beside the copy instead of behind it)
    if (f.part.out && threadIdx.x < 64) {
      int rbv[LIREC_MAX_PROB], ksv[LIREC_MAX_PROB];
#pragma unroll
      for (int i = 0; i < LIREC_MAX_PROB; ++i) {
        int rows = i < f.part.n ? f.part.rows[i] : 0;
        if (i < f.part.n && f.part.dyn[i]) { const int d = *f.part.dyn[i]; rows = d < rows ? d : rows; }
        rbv[i] = (rows + 31) >> 5;
        ksv[i] = i < f.part.n ? f.part.ks[i] : 1;
      }
      const int C = p2_nt_search(rbv, ksv, f.part.grid, f.part.nrep, threadIdx.x);
      if (threadIdx.x == 0) *f.part.out = C;
    }
    return;
  }
  b -= 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < f.nh) {
      const StageHead& h = f.h[i];
      if (b < h.blocks) { stage_rows_q32b(h.X, h.ldx, h.gs, h.gstride, h.goff, h.rowmap, h.count, h.rows, h.D8, h.dst, h.dk, b, h.blocks, h.src); return; }
      b -= h.blocks;
    }
  }
  if (b < f.w_blocks) split_q32b(f.w, b, f.w_blocks);
}

// ---------------------------------------------------------------------------
// Feature assembly from the de-duplicated piece tables (lirec_gather_features).  The reference's loader
// builds every row of the (B, T, R+1, D) block on the host as hstack(clip piece, track-1 piece, track-2 piece)
// (mixed_utils/classification_dataloader.py:336-349, :419, :477-478; mixed_features.py:115-125) and ships the tiled
// float64 block over PCIe; here only the pieces and a (rows, 3) index arrive and the rows are expanded in HBM.
// One thread per (row, 4 columns); a negative index is a zero piece.  F64: the tables are float64 (cast on the fly).
// ---------------------------------------------------------------------------
// round-to-nearest-even fp32 -> bf16 bits (finite inputs; what torch's .to(torch.bfloat16) does)
__device__ __forceinline__ unsigned bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// OUT_BF16: the block is written as bf16 ("bf16 feature storage", BASELINE config 4): half the bytes, read in place by
// layer 1 and its weight gradient
template <bool F64, bool OUT_BF16 = false>
__global__ __launch_bounds__(256) void gather_features_kernel(const void* __restrict__ clip, long ld_clip,
                                                              const void* __restrict__ track, long ld_track,
                                                              const int* __restrict__ index, long rows, int clip_dim,
                                                              int track_dim, float* __restrict__ out, long ld_out) {
  const int D = clip_dim + 2 * track_dim, D4 = D >> 2;
  const long total = rows * D4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / D4;
    const int col = (int)(i - row * D4) << 2;
    const int part = col < clip_dim ? 0 : (col < clip_dim + track_dim ? 1 : 2);
    const int pc = part == 0 ? col : (part == 1 ? col - clip_dim : col - clip_dim - track_dim);
    const int src = index[row * 3 + part];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0) {
      const long off = (long)src * (part == 0 ? ld_clip : ld_track) + pc;
      if constexpr (F64) {
        const double* q = reinterpret_cast<const double*>(part == 0 ? clip : track) + off;
        const double2 a = *reinterpret_cast<const double2*>(q), b = *reinterpret_cast<const double2*>(q + 2);
        v = f32x4{(float)a.x, (float)a.y, (float)b.x, (float)b.y};
      } else {
        v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(part == 0 ? clip : track) + off);
      }
    }
    if constexpr (OUT_BF16) {
      uint2 w;
      w.x = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16);
      w.y = bf16_rne(v[2]) | (bf16_rne(v[3]) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + row * ld_out + col) = w;
    } else {
      *reinterpret_cast<f32x4*>(out + row * ld_out + col) = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Layer 1 on the UNIQUE feature pieces (second half).  A row of the (B, T, R+1, D) block is
// [clip piece | track-1 piece | track-2 piece], every piece shared by many rows; the first Linear of each modality branch
// acts on one piece, so its pre-activation is computed ONCE per piece (four small GEMMs per head over the piece tables:
// zclip [n_clip, 2J] = txt | vis, ztrk [n_track, 2J] = tracks1 | tracks2 weights) and this kernel expands it per row:
//   H1[row, seg*J + c] = dropout(relu(z_seg[index[row, part(seg)], c] + b1_seg[c]))
// with the dropout counters of the dense computation (original row ids), so H1 is bit-identical to layer 1 run on the
// expanded block.  A negative index is a zero piece (pre-activation = bias).  One thread = one column x 4 consecutive
// (compact) rows, the unit one Philox call serves.
// ---------------------------------------------------------------------------
struct GatherActArgs {
  const float* zclip; long ld_zclip;
  const float* ztrk; long ld_ztrk;
  const int* index;                       // [physical rows, 3]
  int gs, gstride, goff; unsigned gs_magic;
  const float* b1[4];
  const int* rowmap; const int* count;    // compact form (context head) or NULL
  float* H1; long ldh;
  int rows, J;
  unsigned seed_lo, seed_hi; const unsigned long long* seed_dev; unsigned site, thresh; float scale;
};

__global__ __launch_bounds__(256) void gather_act_kernel(const GatherActArgs a) {
  const int col = blockIdx.x * 256 + threadIdx.x;            // column of H1, < 4J; a block lies inside one segment
  const int J = a.J, seg = (blockIdx.x * 256) / J, c = col - seg * J;
  const int part = seg < 2 ? 0 : seg - 1;
  const float* ztab = seg < 2 ? a.zclip : a.ztrk;
  const long ldz = seg < 2 ? a.ld_zclip : a.ld_ztrk;
  const int zcol = (seg == 1 || seg == 3) ? J + c : c;
  const float bias = a.b1[seg][c];
  int M = a.rows;
  if (a.count) { const int d = *a.count; M = d < M ? d : M; }
  unsigned key_lo = a.seed_lo, key_hi = a.seed_hi;
  const bool drop = a.thresh != 0u;
  if (drop) apply_seed_offset(key_lo, key_hi, a.seed_dev);
  for (int r0 = 4 * blockIdx.y; r0 < M; r0 += 4 * gridDim.y) {
    unsigned rid[4];
    float z[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int r = r0 + jj < M ? r0 + jj : M - 1;
      rid[jj] = (unsigned)(a.rowmap ? a.rowmap[r] : r);
      const unsigned q = a.gs == 0 ? 0u : (a.gs_magic ? __umulhi(rid[jj], a.gs_magic) : rid[jj] / (unsigned)a.gs);
      const long prow = a.gs == 0 ? (long)rid[jj] : (long)q * a.gstride + (rid[jj] - q * (unsigned)a.gs) + a.goff;
      const int src = a.index[prow * 3 + part];
      z[jj] = src >= 0 ? ztab[(long)src * ldz + zcol] : 0.f;
    }
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (drop) {
      unsigned rnd[4];
      unsigned blk = rid[0] >> 2;
      philox4((unsigned)col, blk, a.site, 0u, key_lo, key_hi, rnd);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if ((rid[jj] >> 2) != blk) {
          blk = rid[jj] >> 2;
          philox4((unsigned)col, blk, a.site, 0u, key_lo, key_hi, rnd);
        }
        const unsigned k = rid[jj] & 3u;
        w[jj] = k == 0u ? rnd[0] : (k == 1u ? rnd[1] : (k == 2u ? rnd[2] : rnd[3]));
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (r0 + jj >= M) break;
      float v = z[jj] + bias * 1.f + 0.f * 0.f;               // the arithmetic of the GEMM epilogue, term for term
      v = relu_f(v);
      v = (!drop || w[jj] >= a.thresh) ? v * a.scale : 0.f;
      a.H1[(long)(r0 + jj) * a.ldh + col] = v;
    }
  }
}

// Backward of the same: the weight gradient of a first layer is  sum over rows of  dZ1[row] (x) piece[index[row]]  =
// sum over PIECES of ( sum of the dZ1 rows that use the piece ) (x) piece.  The inner sums are a product with the 0/1
// incidence matrix of the index, P [rows, pieces] (this kernel writes its ones into a zeroed buffer; a negative index goes to
// one extra "null" column whose piece is a zero row, so that the bias gradient still sees the row), computed by the
// ordinary weight-gradient GEMM  S = P^T dZ1  -- no sort, no atomics, fixed summation order -- followed by  dW1 += S^T pieces.
struct OneHotArgs {
  const int* index; int gs, gstride, goff; unsigned gs_magic;
  const int* rowmap; const int* count;
  float* P; long ldp;
  int rows, n_clip, n_track;              // columns of P: [0, n_clip] clip pieces + null, then two blocks of n_track + 1
};
__global__ __launch_bounds__(256) void onehot_kernel(const OneHotArgs a) {
  int M = a.rows;
  if (a.count) { const int d = *a.count; M = d < M ? d : M; }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 3L * M; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / 3), part = (int)(i - 3L * r);
    const unsigned rid = (unsigned)(a.rowmap ? a.rowmap[r] : r);
    const unsigned q = a.gs == 0 ? 0u : (a.gs_magic ? __umulhi(rid, a.gs_magic) : rid / (unsigned)a.gs);
    const long prow = a.gs == 0 ? (long)rid : (long)q * a.gstride + (rid - q * (unsigned)a.gs) + a.goff;
    const int src = a.index[prow * 3 + part];
    const int npart = part == 0 ? a.n_clip : a.n_track;
    const int base = part == 0 ? 0 : (a.n_clip + 1) + (part - 1) * (a.n_track + 1);
    a.P[(long)r * a.ldp + base + (src >= 0 ? src : npart)] = 1.f;
  }
}

// ---------------------------------------------------------------------------
// Raw feature pooling: what the reference's feature classes compute with numpy when a clip or track
// feature is not in its cache yet --
//   clip-visual : spatial mean of every I3D grid frame in the clip's frame range (visual_features.py:60-103), then the
//                 maximum over those frames (mixed_features.py:54, f_visual = np.max);
//   track       : mean over the person box of the track element's frame (face box blown up to the person box and
//                 scaled to the grid on the host, visual_features.py:105-134), then the maximum over the elements
//                 (mixed_features.py:104-105);
// both are "max over elements of the mean over a box of one frame":
//   out[o, c] = max_{e in [estart[o], estart[o+1])} mean_{y in [y0,y1), x in [x0,x1)} grid[frame_e, c, y, x]
// An element whose frame lies outside the grid is a row of zeros (the reference's `continue` leaves the zero row it
// reserved, :129); an empty box is NaN (np.mean of nothing); no element at all -> zeros.
// The mean reproduces numpy's float32 pairwise summation (8 partial sums up to 128 elements, halves above) so that the
// result is bit-identical to np.mean(..., axis=2) on the float32 grid.
// ---------------------------------------------------------------------------
struct BoxIter {                     // element i of the box in row-major (y, x) order
  const float* base; int w, ld_y;    // base = &grid[frame, c, y0, x0]
  __device__ __forceinline__ float at(int i) const { const int y = i / w; return base[(long)y * ld_y + (i - y * w)]; }
};
__device__ inline float np_pairwise_sum(const BoxIter& b, int i0, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res += b.at(i0 + i);
    return res;
  }
  if (n <= 128) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = b.at(i0 + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += b.at(i0 + i + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += b.at(i0 + i);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(b, i0, n2) + np_pairwise_sum(b, i0 + n2, n - n2);
}

__global__ __launch_bounds__(256) void grid_pool_kernel(const float* __restrict__ grid, int F, int C, int H, int W,
                                                        const int* __restrict__ boxes, const int* __restrict__ estart,
                                                        float* __restrict__ out, long ld_out) {
  const int o = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int e0 = estart[o], e1 = estart[o + 1];
  float best = 0.f;
  bool any = false, nan = false;
  for (int e = e0; e < e1; ++e) {
    const int* bx = boxes + 5 * (long)e;
    const int f = bx[0], y0 = bx[1], y1 = bx[2], x0 = bx[3], x1 = bx[4];
    float v = 0.f;                                   // frame outside the grid: the reserved zero row
    if (f >= 0 && f < F) {
      const int n = (y1 - y0) * (x1 - x0);
      if (n <= 0 || y1 <= y0 || x1 <= x0) {
        nan = true;                                  // np.mean of an empty crop
      } else {
        BoxIter it{grid + (((long)f * C + c) * H + y0) * W + x0, x1 - x0, W};
        v = np_pairwise_sum(it, 0, n) / (float)n;
        if (v != v) nan = true;
      }
    }
    best = any ? fmaxf(best, v) : v;
    any = true;
  }
  out[(long)o * ld_out + c] = nan ? __builtin_nanf("") : best;
}

// out[o, :] = max over rows idx[e], e in [estart[o], estart[o+1]), of src (text tokens in a time range,
// text_features.py:140-165 + mixed_features.py:61 f_text = np.max; rows of zeros when there is no row, :172-177)
__global__ __launch_bounds__(256) void rows_max_kernel(const float* __restrict__ src, long ld, const int* __restrict__ idx,
                                                       const int* __restrict__ estart, int dim, float* __restrict__ out,
                                                       long ld_out) {
  const int o = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dim) return;
  const int e0 = estart[o], e1 = estart[o + 1];
  float best = 0.f;
  bool nan = false;
  for (int e = e0; e < e1; ++e) {
    const float v = src[(long)idx[e] * ld + c];
    if (v != v) nan = true;
    best = (e == e0) ? v : fmaxf(best, v);
  }
  out[(long)o * ld_out + c] = nan ? __builtin_nanf("") : best;
}

// ---------------------------------------------------------------------------
// K5: max-margin losses, forward + d(loss)/d(logits) in one pass.
// One workgroup per clip; the T x C sigmoid table lives in LDS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ctr[i] += inc[i]: the "next step" node of a captured train-step graph (dropout key offset, Adam step)
__global__ void counter_add_kernel(long long* ctr, long long i0, long long i1, long long i2, long long i3, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long inc[4] = {i0, i1, i2, i3};
    for (int i = 0; i < n; ++i) ctr[i] += inc[i];
  }
}

// zero a 16-byte aligned buffer of n16 16-byte words (+ tail bytes) and, in the same launch, ctr[i] += inc[i]: the first
// kernel of a replayed train step (optimizer.zero_grad + "next step" in one launch instead of a memset and a
// one-thread kernel with a dependency gap between them)
__global__ __launch_bounds__(256) void zero_count_kernel(uint4* __restrict__ p, long n16, unsigned char* __restrict__ tail, int ntail,
                                                         long long* ctr, long long i0, long long i1, long long i2, long long i3, int n) {
  if (ctr && threadIdx.x == 0 && blockIdx.x == 0) {
    const long long inc[4] = {i0, i1, i2, i3};
    for (int i = 0; i < n; ++i) ctr[i] += inc[i];
  }
  const uint4 z = {0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) { p[i] = z; p[i + stride] = z; p[i + 2 * stride] = z; p[i + 3 * stride] = z; }
  for (; i < n16; i += stride) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// loader-typed reads: the pointer is declared fp32 / int32 in the ABI; with loader_types it is f64 / i64
__device__ __forceinline__ float ld_f(const float* p, long i, bool f64) {
  return f64 ? (float)reinterpret_cast<const double*>(p)[i] : p[i];
}
__device__ __forceinline__ int ld_i(const int32_t* p, long i, bool i64) {
  return i64 ? (int)reinterpret_cast<const long long*>(p)[i] : p[i];
}

__global__ __launch_bounds__(256) void margin_loss_kernel(const lirec_margin_loss_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int T = a.T, C = a.C, NR = a.NR;
  const bool has_rels = a.rels != nullptr;
  const int NR1 = has_rels ? NR + 1 : 0;
  float* S = sm;                       // [T*C]
  float* Q = S + T * C;                // [T*NR1]
  float* red = Q + T * NR1;            // [16]
  int* ish = reinterpret_cast<int*>(red + 16);   // [4]

  // per-clip masks and labels, staged once (and converted when they arrive in the loader's f64 / i64)
  float* memS = reinterpret_cast<float*>(ish + 4);   // [T]
  float* wS = memS + T;                              // [C]
  int* rS = reinterpret_cast<int*>(wS + C);          // [T]
  const bool lt = a.loader_types != 0;
  for (int t = tid; t < T; t += nt) {
    memS[t] = a.mem ? ld_f(a.mem, (long)b * T + t, lt) : 1.f;
    rS[t] = has_rels ? ld_i(a.r, (long)b * T + t, lt) : 0;
  }
  for (int c = tid; c < C; c += nt) wS[c] = a.w ? ld_f(a.w, (long)b * C + c, lt) : 1.f;
  __syncthreads();
  const int y = ld_i(a.y, (long)b * (a.y_stride > 0 ? a.y_stride : 1), lt);
  const int g0 = a.g ? ld_i(a.g, 2 * b, lt) : 0, g1 = a.g ? ld_i(a.g, 2 * b + 1, lt) : 0;
  const int r0 = has_rels ? rS[g0] : 0, r1 = has_rels ? rS[g1] : 0;
  const float* mem = a.mem ? memS : nullptr;
  const float* w = a.w ? wS : nullptr;
  const float NEG_INF = -__builtin_inff();

  for (int idx = tid; idx < T * C; idx += nt) {
    const int t = idx / C, c = idx - t * C;
    float* xp = a.ints + ((long)b * T + t) * a.ld_ints + c;
    float x = *xp;
    if (mem && mem[t] == 0.f) {
      x = NEG_INF;
      if (a.mask_inplace) *xp = x;
    }
    S[idx] = sigmoidf_(x);
  }
  for (int idx = tid; idx < T * NR1; idx += nt) {
    const int t = idx / NR1, c = idx - t * NR1;
    const bool valid = (!mem || mem[t] != 0.f) && rS[t] != NR && c < NR;
    Q[idx] = valid ? sigmoidf_(a.rels[((long)b * T + t) * a.ld_rels + c]) : 0.f;
  }
  __syncthreads();

  // positive track (wave 0; lane t = track t, T <= 64 -- longer track lists take the serial loop below)
  if (tid < 64) {
    const int lane = tid;
    int k = (a.sel && a.sel[b] >= 0) ? a.sel[b] : -1;
    const bool forced = k >= 0 || a.tr_correct;
    if (k < 0 && a.tr_correct) k = 0;
    if (a.sample && (!forced || a.probs_out)) {
      // categorical distribution over the tracks (mlp/model.py:470, :540-542), on the -inf-masked logits
      float tot_all = 0.f, cum_base = 0.f;
      unsigned rnd[4] = {0u, 0u, 0u, 0u};
      unsigned klo = (unsigned)(a.sample_seed & 0xffffffffull), khi = (unsigned)(a.sample_seed >> 32);
      apply_seed_offset(klo, khi, reinterpret_cast<const unsigned long long*>(a.sample_seed_dev));
      philox4((unsigned)b, 0u, (unsigned)LIREC_SITE_TRACK_SAMPLE, 0u, klo, khi, rnd);
      const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
      // pass 1: maxima;  pass 2: sums;  pass 3: probabilities, their total, the pick -- T <= 64 makes every pass one step
      float mxi = NEG_INF, mxr = NEG_INF;
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        float xi = NEG_INF, xr = NEG_INF;
        if (t < T) {
          xi = (mem && mem[t] == 0.f) ? NEG_INF : a.ints[((long)b * T + t) * a.ld_ints + y];
          if (has_rels) {
            const bool valid = (!mem || mem[t] != 0.f) && rS[t] != NR && r0 < NR;
            xr = valid ? a.rels[((long)b * T + t) * a.ld_rels + r0] : NEG_INF;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { xi = fmaxf(xi, __shfl_xor(xi, o, 64)); xr = fmaxf(xr, __shfl_xor(xr, o, 64)); }
        mxi = fmaxf(mxi, xi); mxr = fmaxf(mxr, xr);
      }
      float si = 0.f, sr = 0.f;
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        float ei = 0.f, er = 0.f;
        if (t < T) {
          const float xi = (mem && mem[t] == 0.f) ? NEG_INF : a.ints[((long)b * T + t) * a.ld_ints + y];
          ei = expf(xi - mxi);                                   // all tracks padded: exp(-inf + inf) = NaN, as torch
          if (has_rels) {
            const bool valid = (!mem || mem[t] != 0.f) && rS[t] != NR && r0 < NR;
            er = expf((valid ? a.rels[((long)b * T + t) * a.ld_rels + r0] : NEG_INF) - mxr);
          }
        }
        si += wave_sum(ei); sr += wave_sum(er);
      }
      int pick = -1, last_pos = -1;
      for (int pass = 0; pass < 2; ++pass) {                    // pass 0: total of p; pass 1: inverse CDF
        cum_base = 0.f;
        for (int t0 = 0; t0 < T; t0 += 64) {
          const int t = t0 + lane;
          float p = 0.f;
          if (t < T) {
            const float xi = (mem && mem[t] == 0.f) ? NEG_INF : a.ints[((long)b * T + t) * a.ld_ints + y];
            p = expf(xi - mxi) / si;
            if (has_rels) {
              const bool valid = (!mem || mem[t] != 0.f) && rS[t] != NR && r0 < NR;
              float q = expf((valid ? a.rels[((long)b * T + t) * a.ld_rels + r0] : NEG_INF) - mxr) / sr;
              if (q != q) q = 0.f;                               // probs_rels[probs_rels != probs_rels] = 0  (:542)
              p = (p + q) / 2.f;
            }
            if (pass == 0 && a.probs_out) a.probs_out[(long)b * T + t] = p;
          }
          if (pass == 0) { tot_all += wave_sum(p); continue; }
          float incl = p;                                        // inclusive prefix sum over the lanes
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const float v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
          }
          incl += cum_base;
          const unsigned long long hit = __ballot(t < T && p > 0.f && incl > u * tot_all);
          const unsigned long long pos = __ballot(t < T && p > 0.f);
          if (pick < 0 && hit) pick = t0 + (int)__builtin_ctzll(hit);
          if (pos) last_pos = t0 + 63 - (int)__builtin_clzll(pos);
          cum_base = __shfl(incl, 63, 64);
        }
      }
      if (pick < 0) pick = last_pos >= 0 ? last_pos : 0;         // u * total rounded past the last step; no mass at all: 0
      if (!forced) k = pick;
    }
    if (k < 0) {
      // argmax_t (S[t,y] + Q[t,r0]) * mem[t], first maximum (:479, :552-553)
      float best = -__builtin_inff();
      int bi = 0x7fffffff;
      for (int t = lane; t < T; t += 64) {
        float v = S[t * C + y] + (has_rels ? Q[t * NR1 + r0] : 0.f);
        v *= mem ? mem[t] : 1.f;
        if (v > best) { best = v; bi = t; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      k = bi == 0x7fffffff ? 0 : bi;
    }
    if (lane == 0) {
      ish[0] = k;
      if (a.sel_out) a.sel_out[b] = k;
    }
  }
  if (a.sample == 2) return;                                     // probabilities / draw only
  // valid-row count for the clip-level multitask loss (mean over rows with label != NR)
  int nvalid = a.B;
  if (has_rels && a.rels_mean_valid) {
    float cnt = 0.f;
    for (int i = tid; i < a.B; i += nt) cnt += (ld_i(a.r, (long)i * T, lt) != NR) ? 1.f : 0.f;
    nvalid = (int)(block_sum(cnt, red) + 0.5f);
  }
  __syncthreads();
  const int k = ish[0];
  const float pos = S[k * C + y];
  const float posr = has_rels ? Q[k * NR1 + r0] : 0.f;
  // denominators of the batch means: this batch's own (the reference, single device), or the data-parallel caller's
  // (lirec_margin_loss_args::batch_divisor: the global batch's, divided by world)
  float div_b = a.divisors_dev ? a.divisors_dev[0] : a.batch_divisor;
  float div_r = a.divisors_dev ? a.divisors_dev[1] : a.rels_divisor;
  if (!(div_b > 0.f)) div_b = (float)a.B;
  if (!(div_r > 0.f)) div_r = (float)nvalid;
  const float coef_i = a.lymbda / div_b;
  const float coef_r = has_rels ? (a.rels_mean_valid ? (nvalid > 0 ? 1.f / div_r : 0.f) : 1.f / div_b) : 0.f;
  const float m = a.margin;

  float li = 0.f, ci = 0.f, lr = 0.f, cr = 0.f;
  if (!a.max_neg) {
    for (int idx = tid; idx < T * C; idx += nt) {
      const int t = idx / C, c = idx - t * C;
      const bool excl = a.tr_correct ? ((t == g0 || t == g1) && c == y) : (c == y);
      const bool mi = (!mem || mem[t] != 0.f) && (!w || w[c] != 0.f) && !excl;
      const float s = S[idx], term = m - pos + s;
      const bool act = mi && term > 0.f;
      if (act) { li += term; ci += 1.f; }
      a.d_ints[((long)b * T + t) * a.ld_dints + c] = act ? coef_i * s * (1.f - s) : 0.f;
    }
    for (int idx = tid; idx < T * NR; idx += nt) {
      const int t = idx / NR, c = idx - t * NR;
      const int rt = rS[t];
      const bool excl = a.tr_correct ? (c == rt) : (c == r0 || c == r1);
      const bool mr = (!mem || mem[t] != 0.f) && rt != NR && !excl;
      const float q = Q[t * NR1 + c], term = m - posr + q;
      const bool act = mr && term > 0.f;
      if (act) { lr += term; cr += 1.f; }
      a.d_rels[((long)b * T + t) * a.ld_drels + c] = act ? coef_r * q * (1.f - q) : 0.f;
    }
  } else {
    // max over classes per track (mlp/model.py:483-486, 557-562): padded tracks still
    // contribute relu(m - pos); the gradient goes to the first arg-max column if unmasked
    for (int idx = tid; idx < T * C; idx += nt) {
      const int t = idx / C, c = idx - t * C;
      a.d_ints[((long)b * T + t) * a.ld_dints + c] = 0.f;
    }
    for (int idx = tid; idx < T * NR; idx += nt) {
      const int t = idx / NR, c = idx - t * NR;
      a.d_rels[((long)b * T + t) * a.ld_drels + c] = 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    for (int t = wv; t < T; t += nw) {
      float best = -1.f; int bi = 0x7fffffff;
      for (int c = lane; c < C; c += 64) {
        const bool excl = a.tr_correct ? ((t == g0 || t == g1) && c == y) : (c == y);
        const bool mi = (!mem || mem[t] != 0.f) && (!w || w[c] != 0.f) && !excl;
        const float v = mi ? S[t * C + c] : 0.f;
        if (v > best) { best = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) {
        const float term = m - pos + best;
        if (term > 0.f) {
          li += term; ci += 1.f;
          const bool excl = a.tr_correct ? ((t == g0 || t == g1) && bi == y) : (bi == y);
          const bool mi = (!mem || mem[t] != 0.f) && (!w || w[bi] != 0.f) && !excl;
          if (mi) a.d_ints[((long)b * T + t) * a.ld_dints + bi] = coef_i * best * (1.f - best);
        }
      }
      if (has_rels) {
        const int rt = rS[t];
        float bq = -1.f; int bqi = 0x7fffffff;
        for (int c = lane; c < NR1; c += 64) {
          const bool excl = a.tr_correct ? (c == rt) : (c == r0 || c == r1);
          const bool mr = (!mem || mem[t] != 0.f) && rt != NR && c < NR && !excl;
          const float v = mr ? Q[t * NR1 + c] : 0.f;
          if (v > bq) { bq = v; bqi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bq, o, 64); const int oi = __shfl_xor(bqi, o, 64);
          if (ov > bq || (ov == bq && oi < bqi)) { bq = ov; bqi = oi; }
        }
        if (lane == 0) {
          const float term = m - posr + bq;
          if (term > 0.f) {
            lr += term; cr += 1.f;
            const bool excl = a.tr_correct ? (bqi == rt) : (bqi == r0 || bqi == r1);
            const bool mr = (!mem || mem[t] != 0.f) && rt != NR && bqi < NR && !excl;
            if (mr) a.d_rels[((long)b * T + t) * a.ld_drels + bqi] = coef_r * bq * (1.f - bq);
          }
        }
      }
    }
  }
  li = block_sum(li, red); ci = block_sum(ci, red);
  if (has_rels) { lr = block_sum(lr, red); cr = block_sum(cr, red); }
  __syncthreads();
  if (tid == 0) {
    // d(loss)/d(pos): every active hinge term carries -1
    a.d_ints[((long)b * T + k) * a.ld_dints + y] += -ci * coef_i * pos * (1.f - pos);
    if (has_rels && r0 < NR) a.d_rels[((long)b * T + k) * a.ld_drels + r0] += -cr * coef_r * posr * (1.f - posr);
    if (!a.arrive) {
      a.partial[2 * b] = li * coef_i;
      a.partial[2 * b + 1] = lr * coef_r;
    }
  }
  if (a.arrive) {
    // In-launch finalize: the workgroup that arrives last adds the per-clip partials in clip order (deterministic).
    // Hand-off by the write-through form (cdna_hip_programming.md, Guideline 16): every partial is ONE agent-scope
    // (sc1) store by one lane, drained before that lane's ticket; the last arriver reads them with agent-scope loads,
    // which bypass its L1 -- no cache is trusted on either side.
    if (tid == 0) {
      __hip_atomic_store(a.partial + 2 * b, li * coef_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.partial + 2 * b + 1, lr * coef_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ish[1] = (ticket == a.B - 1) ? 1 : 0;
    }
    __syncthreads();
    if (ish[1]) {
      float v = 0.f;
      for (int i = tid; i < 2 * a.B; i += nt)
        v += __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = block_sum(v, red);
      if (tid == 0) {
        *a.loss = v;
        __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      }
    }
  }
}

// loss = sum of per-clip partials in a fixed order (deterministic)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s;
}

// MultiTaskCrossEntropyLoss (mlp/model.py:367-378): block i < B -> interaction row i,
// block B+i -> relationship row i (skipped when its label is None).
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ ints, long ld_ints,
                                                      const float* __restrict__ rels, long ld_rels,
                                                      const int* __restrict__ y, const int* __restrict__ r,
                                                      const float* __restrict__ class_w, int B, int C, int NR,
                                                      float* __restrict__ d_ints, long ld_dints,
                                                      float* __restrict__ d_rels, long ld_drels,
                                                      float* __restrict__ partial, float den_ints, float den_rels,
                                                      const float* __restrict__ dens_dev) {
  __shared__ float red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool is_rel = blockIdx.x >= B;
  const int row = is_rel ? blockIdx.x - B : blockIdx.x;
  const float* x = is_rel ? rels + (long)row * ld_rels : ints + (long)row * ld_ints;
  float* dx = is_rel ? d_rels + (long)row * ld_drels : d_ints + (long)row * ld_dints;
  const int n = is_rel ? NR : C;
  const int tgt = is_rel ? r[row] : y[row];
  // denominators: sum of class weights of the targets (ints), count of labelled rows (rels)
  float den = 0.f;
  for (int i = tid; i < B; i += nt) den += is_rel ? ((r[i] != NR) ? 1.f : 0.f) : (class_w ? class_w[y[i]] : 1.f);
  den = block_sum(den, red);
  {   // the data-parallel caller's denominators (lirec_ce_loss: den_ints / den_rels / dens_dev), when given
    const float given = dens_dev ? dens_dev[is_rel ? 1 : 0] : (is_rel ? den_rels : den_ints);
    if (given > 0.f) den = given;
  }
  if (is_rel && tgt == NR) {
    for (int c = tid; c < n; c += nt) dx[c] = 0.f;
    if (tid == 0) partial[blockIdx.x] = 0.f;
    return;
  }
  float mx = -__builtin_inff();
  for (int c = tid; c < n; c += nt) mx = fmaxf(mx, x[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[8 + (tid >> 6)] = mx;
  __syncthreads();
  for (int i = 0; i < (nt >> 6); ++i) mx = fmaxf(mx, red[8 + i]);
  float se = 0.f;
  for (int c = tid; c < n; c += nt) se += expf(x[c] - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  const float wt = (!is_rel && class_w) ? class_w[tgt] : 1.f;
  const float scale = wt / den;
  for (int c = tid; c < n; c += nt) dx[c] = scale * (expf(x[c] - lse) - (c == tgt ? 1.f : 0.f));
  if (tid == 0) partial[blockIdx.x] = scale * (lse - x[tgt]);
}

// ---------------------------------------------------------------------------
// K7: fused Adam over one flat buffer (torch.optim.Adam single-tensor op order)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float step_size, float bc2_sqrt, float beta1, float beta2,
                                                   float eps, float wd, float gscale, float lr,
                                                   const long long* __restrict__ step_dev,
                                                   long long* count_dev = nullptr, int* ticket = nullptr, int advance = 0) {
  // count_dev (lirec_adam_step_counted): a counter of COMPLETED steps owned by this launch's stream -- the step is *count_dev + 1,
  // and, `advance`, the workgroup that finishes last stores it back (every workgroup has read the counter by then: a workgroup
  // takes its ticket behind its last element) -- the one-thread counter launch in front of the side stream's update is gone
  long long t_counted = 0;
  if (count_dev) t_counted = __hip_atomic_load(count_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  if (step_dev || count_dev) {        // step kept on the device (graph replay): same double-precision bias corrections as the host
    const double t = count_dev ? (double)t_counted : (double)*step_dev;
    step_size = (float)((double)lr / (1.0 - pow((double)beta1, t)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
  }
  const AdamFuse ad{p, g, m, v, step_size, bc2_sqrt, beta1, beta2, eps, wd, gscale, lr, step_dev};
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    (void)adam4(ad, step_size, bc2_sqrt, 4 * i, reinterpret_cast<const f32x4*>(g)[i]);
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) (void)adam1(ad, step_size, bc2_sqrt, i, g[i]);
  if (count_dev && advance) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == (int)gridDim.x - 1) {
        __hip_atomic_store(count_dev, t_counted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // ready for the next launch
      }
    }
  }
}

__global__ __launch_bounds__(256) void cast_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long n2 = n >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 d = reinterpret_cast<const double2*>(src)[i];
    reinterpret_cast<float2*>(dst)[i] = make_float2((float)d.x, (float)d.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = (float)src[n - 1];
}

// ---------------------------------------------------------------------------
// Evaluation counters (lirec_eval_max_tracks; utils/evaluation.py:114-176, :179-271).  One workgroup per clip.
// ---------------------------------------------------------------------------
// (value, index) with numpy argmax semantics: the larger value wins, ties go to the lower index; index 0x7fffffff = empty.
// V = float for single probabilities / logits, double for SUMS of two probabilities: the reference adds its float32 sigmoids in
// double (utils/evaluation.py:220 appends a float64 zero column to the relationship probabilities, which promotes them and
// every sum they enter, :221-222,229-231) -- 1 + 4e-8 and 1 + 0 are different numbers there and the same float
// (tests/golden/metrics_ties.npz, case tiny_rels: float sums count 3 joint hits where the reference counts 13).
template <class V>
__device__ __forceinline__ void argmax_take(V& bv, int& bi, V ov, int oi) {
  if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
}
template <class V>
__device__ __forceinline__ void wave_argmax(V& bv, int& bi) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const V ov = __shfl_down(bv, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    argmax_take(bv, bi, ov, oi);
  }
}
// first index of the maximum of f(i), i < n, over the workgroup: wave shuffles, then one LDS slot per wave
// (the first version reduced through LDS with a barrier per halving step: 10 barriers a call, seven calls a clip)
template <class V, class F>
__device__ __forceinline__ int block_argmax(int n, F f, V* vred, int* ired) {
  const int tid = threadIdx.x, nt = blockDim.x;
  V bv = -(V)__builtin_inff();
  int bi = 0x7fffffff;
  for (int i = tid; i < n; i += nt) argmax_take<V>(bv, bi, f(i), i);
  wave_argmax<V>(bv, bi);
  if ((tid & 63) == 0) { vred[tid >> 6] = bv; ired[tid >> 6] = bi; }
  __syncthreads();
  V rv = vred[0];
  int ri = ired[0];
  for (int w = 1; w < (nt >> 6); ++w) argmax_take<V>(rv, ri, vred[w], ired[w]);
  __syncthreads();
  return ri;
}

__global__ __launch_bounds__(256) void eval_max_tracks_kernel(const lirec_eval_args a) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const bool has_rels = a.rels != nullptr;
  const int T = a.T, C = a.C, NR = has_rels ? a.NR : 0;
  const int NR1 = has_rels ? NR + 1 : 0;
  float* L = sm;                         // [T*C]   masked logits
  float* S = L + T * C;                  // [T*C]   sigmoid
  float* RL = S + T * C;                 // [T*NR]  masked relationship logits
  float* Q = RL + T * NR;                // [T*NR1] sigmoid + the "None" column
  // reduction scratch behind the tables (the launcher reserves 512 words for it), 8-byte aligned for the double slots
  double* dred = reinterpret_cast<double*>(sm + ((2 * T * C + T * (NR + NR1) + 1) & ~1));   // [8] per-wave slots
  double* Mt = dred + 8;                 // [<= 64] per-track maxima of the joint score
  float* vred = reinterpret_cast<float*>(Mt + 64);   // [8]
  int* ired = reinterpret_cast<int*>(vred + 8);      // [8]
  const bool lt = a.loader_types != 0;
  const float NEG_INF = -__builtin_inff();
  const int y = ld_i(a.y, b, lt);
  const int g[2] = {ld_i(a.g, 2 * b, lt), ld_i(a.g, 2 * b + 1, lt)};
  const bool keep = !(a.just_zeros && a.just_zeros[b]);
  for (int idx = tid; idx < T * C; idx += nt) {
    const int t = idx / C, c = idx - t * C;
    const bool pad = a.mem && ld_f(a.mem, (long)b * T + t, lt) == 0.f;
    const float x = pad ? NEG_INF : a.ints[((long)b * T + t) * a.ld_ints + c];
    L[idx] = x; S[idx] = sigmoidf_(x);
  }
  if (has_rels) {
    for (int idx = tid; idx < T * NR; idx += nt) {
      const int t = idx / NR, c = idx - t * NR;
      const bool pad = a.mem && ld_f(a.mem, (long)b * T + t, lt) == 0.f;
      const float x = pad ? NEG_INF : a.rels[((long)b * T + t) * a.ld_rels + c];
      RL[idx] = x; Q[t * NR1 + c] = sigmoidf_(x);
    }
    for (int t = tid; t < T; t += nt) Q[t * NR1 + NR] = 0.f;
  }
  __syncthreads();
  const int r0 = has_rels ? ld_i(a.r, (long)b * T, lt) : 0;           // label of the GT pair (candidate 0), :215
  const bool sel = has_rels && r0 != NR;                               // mlp/test.py:57
  // class / relationship prediction given each ground-truth track (raw masked logits)
  int cls_pred[2], rel_pred[2] = {0, 0}, rel_gt[2] = {0, 0};
  for (int i = 0; i < 2; ++i) {
    const float* row = L + g[i] * C;
    cls_pred[i] = block_argmax<float>(C, [&](int c) { return row[c]; }, vred, ired);
    if (sel) {
      const float* rr = RL + g[i] * NR;
      rel_pred[i] = block_argmax<float>(NR, [&](int c) { return rr[c]; }, vred, ired);
      rel_gt[i] = ld_i(a.r, (long)b * T + g[i], lt);
    }
  }
  int pr_track = 0, j_trk = 0, j_cls = 0, j_rel = 0;
  if (keep) {
    if (has_rels) {
      pr_track = block_argmax<double>(T, [&](int t) { return (double)S[t * C + y] + (double)Q[t * NR1 + r0]; }, dred, ired);
      // joint argmax over (t, c, r) of S[t,c] + Q[t,r], the sums in double (utils/evaluation.py:229-235 tiles a
      // (B*T, C, NR+1) tensor for it).  Rounding is monotone, so the maximum of the rounded sums over (c, r) is
      // fl(max_c S + max_r Q): one pass per track instead of T*C*(NR+1) sums.  numpy's argmax returns the FIRST flat index
      // that reaches the maximum, which may be a pair whose exact sum is smaller but rounds to the same double: after the
      // maximum M is known, the first track that reaches it is scanned for the first (c, r) with fl(S + Q) == M.
      if (T <= 64) {
        const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
        for (int t = wave; t < T; t += nw) {
          float sv = -__builtin_inff(), qv = -__builtin_inff();
          int si = 0x7fffffff, qi = 0x7fffffff;
          for (int c = lane; c < C; c += 64) argmax_take<float>(sv, si, S[t * C + c], c);
          for (int r = lane; r < NR1; r += 64) argmax_take<float>(qv, qi, Q[t * NR1 + r], r);
          wave_argmax<float>(sv, si);
          wave_argmax<float>(qv, qi);
          if (lane == 0) Mt[t] = (double)sv + (double)qv;
        }
        __syncthreads();
        j_trk = block_argmax<double>(T, [&](int t) { return Mt[t]; }, dred, ired);
        const double M = Mt[j_trk];
        const int pair = block_argmax<float>(C, [&](int c) {
          // (1 where the class has a matching r: block_argmax then returns the lowest such c -- c-major order)
          const double sc = (double)S[j_trk * C + c];
          for (int r = 0; r < NR1; ++r)
            if (sc + (double)Q[j_trk * NR1 + r] == M) return 1.f;
          return 0.f;
        }, vred, ired);
        j_cls = pair;
        j_rel = 0;
        const double sc = (double)S[j_trk * C + j_cls];
        for (int r = NR1 - 1; r >= 0; --r)
          if (sc + (double)Q[j_trk * NR1 + r] == M) j_rel = r;
      } else {
        const int flat = block_argmax<double>(T * C * NR1, [&](int i) {
          const int t = i / (C * NR1), rem = i - t * (C * NR1);
          const int c = rem / NR1, r = rem - c * NR1;
          return (double)S[t * C + c] + (double)Q[t * NR1 + r];
        }, dred, ired);
        j_trk = flat / (C * NR1);
        const int rem = flat - j_trk * (C * NR1);
        j_cls = rem / NR1; j_rel = rem - j_cls * NR1;
      }
    } else {
      pr_track = block_argmax<float>(T, [&](int t) { return S[t * C + y]; }, vred, ired);
      const int flat = block_argmax<float>(T * C, [&](int i) { return S[i]; }, vred, ired);
      j_trk = flat / C; j_cls = flat - j_trk * C;
    }
  }
  if (tid == 0) {
    long long add[7] = {0, 0, 0, 0, 0, 0, 0};     // total, total_cl, total_rels, top1, trks_top1, cls_top1, rels_top1
    add[1] = 1;
    if (sel) add[2] = 1;
    const bool cls_miss = cls_pred[0] != y;
    add[5] += !cls_miss;
    if (cls_miss && cls_pred[1] == y) add[5] += 1;
    if (sel) {
      const bool rel_miss = rel_pred[0] != rel_gt[0];
      add[6] += !rel_miss;
      if (rel_miss && rel_pred[1] == rel_gt[1]) add[6] += 1;
    }
    if (keep) {
      add[0] = 1;
      const bool jcr = (j_cls == y) && (!has_rels || j_rel == r0);
      // first ground-truth track
      const bool trk0 = pr_track == g[0];
      add[4] += trk0;
      const bool joint0 = jcr && j_trk == g[0];
      add[3] += joint0;
      // second one: only where it exists and the first pass missed (:160-175)
      const bool open_trk = (g[1] != 0) && !trk0;
      if (open_trk && pr_track == g[1]) add[4] += 1;
      if (open_trk && !joint0 && jcr && j_trk == g[1]) add[3] += 1;
    }
    for (int i = 0; i < 7; ++i)
      if (add[i]) atomicAdd(reinterpret_cast<unsigned long long*>(a.counters) + i, (unsigned long long)add[i]);
  }
}

__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, int rows, int cols, unsigned seed_lo, unsigned seed_hi,
                                    const unsigned long long* __restrict__ seed_dev, unsigned site, unsigned thresh) {
  apply_seed_offset(seed_lo, seed_hi, seed_dev);
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols), col = (int)(i - (long)row * cols);
    unsigned rnd[4];
    philox4((unsigned)col, (unsigned)(row >> 2), site, 0u, seed_lo, seed_hi, rnd);
    keep[i] = (thresh == 0u || rnd[row & 3] >= thresh) ? 1 : 0;
  }
}

}  // namespace lirec
