// The gate GEMMs (GatingUnit, mlp/model.py:349-354: 3072 x 3072 weights against ~1024 candidate rows) on q32b operands with
// WAVE-SPECIALISED workgroups (gfx950).
//
// Why another kernel.  tools/micro/l2_lds_intake.hip (profiles/r04_l2_lds_intake.txt): a CU takes 131 GB/s of cache-resident
// operand bytes into LDS when four or more waves issue LDS-DMA back to back -- not the 35 GB/s round 3 assumed; that figure is what
// ONE issuing wave reaches (33.5), because a global_load_lds_dwordx4 holds the issuing wave for 60-140 cycles.  In gemm_p2.hpp
// every wave both issues its share of the DMA and computes: at 32 MF = 256 rows per tile the eight requests per wave and k-step
// hide behind 96 MFMAs, but a 1024-row problem cut for 256 CUs leaves 64-row tiles -- 24 MFMAs per wave and k-step against the
// same eight requests -- and the k-step takes 1.2 us whatever the matrix pipe does (measured: gate forward 117 us on
// gemm_p2_nt at MF = 2 against 101 us on the on-the-fly core).  Here the two jobs belong to different waves:
//   * 128 x 128 x 32 tiles, 512 threads: waves 0-3 COMPUTE (2 x 2, each 64 x 64 outputs = 4 x 4 tiles of
//     v_mfma_f32_16x16x32_bf16, three MFMAs per tile and k-step: hi*lo + lo*hi + hi*hi into one fp32 accumulator), waves 4-7 LOAD
//     (each fills 32 rows of both operands per k-step by LDS-DMA: eight 1-KiB requests).  A workgroup's waves go to the four
//     SIMDs in turn, so every SIMD hosts one compute wave -- which owns its matrix pipe -- and one loader whose issue stalls cost
//     the compute wave nothing;
//   * LDS = a ring of FIVE 32-KiB slots, all 160 KiB (A image 16 KiB | B image 16 KiB per k-step, gemm_p2's swizzled [row][128 B]
//     image); ONE barrier per k-step: in front of barrier t the loaders wait (counted vmcnt) until steps <= t + 1 have landed,
//     behind it they request step t + 4 into the slot step t - 1 has just vacated -- three k-steps (96 KiB per CU) in flight:
//     with four slots (64 KiB in flight) the k-step took 0.73 us, the time a request needs to land divided by two; the compute
//     waves multiply step t from registers + slot t and, at the end of the step, fetch the B fragments of step t + 1, which
//     landed before barrier t (no second barrier, no exposed fragment latency);
//   * one tile per workgroup, tiles dealt to the XCDs by COLUMN: the 24 workgroups of an XCD share three 128-column weight
//     panels (L2 hits), the rows (12.6 MB for 1024 x 3072) come out of the Infinity Cache.
// KIND 0 (NT, forward): p.A rows q32b [M][K], p.B weights q32b [N][K] (k-contiguous), epilogue bias + relu + dropout.
#pragma once
#include "gemm_p2.hpp"

namespace lirec {

struct P3 {
  static constexpr int BM = 128, BN = 128, BK = 32, NTHR = 512;
  static constexpr int SLOT = 32768, BOFF = 16384, NSLOT = 5, LDS_BYTES = NSLOT * SLOT;
};

template <int N> __device__ __forceinline__ void p3_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
__device__ __forceinline__ int p3_next(int s) { return s == P3::NSLOT - 1 ? 0 : s + 1; }

// tile (tm, tn) of problem p: rows [128 tm, +128), columns [128 tn, +128), all of k.
// KIND 0 (NT): p.B = weights q32b [N][K], k-contiguous rows -- the B image is the A image's twin.
// KIND 1 (NN, the gate's data gradient dEE = dZg Wg): p.B = the weights as staged for the FORWARD, q32b [K][N] with the reduced
//   index as the ROW index, at the problem's first column block: B image [32 k][512 B], chunk ch of row k at
//   (ch & ~15) | ((ch & 15) ^ f(k)), fragments by ds_read_b64_tr_b16 (gemm_p2's weight-gradient B image at half the width);
//   epilogue (acc + beta C) * tanh' * dropout factor (EPI_TANH_BWD).
// KIND 2 (TN, the gate's weight gradient dWg = dZg^T EE, p.K = the rows reduced over): BOTH operands k-major -- p.A = q32b rows
//   whose columns are the output's rows (dZg [n][N]), p.B = q32b rows whose columns are the output's columns (EE [n][K]) -- two
//   such images, all fragments by transposed reads; C = beta C + acc; the bias gradient (column sums of p.A) rides along on the
//   matrix pipe in the column-0 tiles (A fragment x ones), dbias (=, dbias_set) or (+=).
// ADAM (KIND 2): the update of the parameters whose gradient this is, folded into the epilogue (AdamFuse, gemm.hpp; the gate's
//   weight: 18.9 M of the 34 M parameters at the bench shape).  Behind the k loop the LDS ring is free: the compute waves leave
//   their accumulators there as a [128][132] fp32 tile and ALL EIGHT waves walk it as float4s along the rows -- gradient = beta C
//   + acc (stored: it stays observable), Adam on the parameters and moments at the same offset of their flat buffers (adam4: the
//   bits of adam_kernel), the new weights' q32b form into p.aux_out (the operand the next forward and data gradient read).  Against
//   the separate pass that saves the gradient's round trip (written here, read back there) and next step's staging of the weights.
template <int KIND, int ABL = 0, bool ADAM = false>
__device__ __forceinline__ void p3_tile(const GemmProblem& p, unsigned char* smem, int tm_, int tn_, int lane, int wave,
                                        const AdamFuse* adp = nullptr) {
  const int tm = __builtin_amdgcn_readfirstlane(tm_), tn = __builtin_amdgcn_readfirstlane(tn_);
  static_assert(!ADAM || KIND == 2, "the fused update belongs to the weight-gradient form");
  constexpr int TLD = 132;                                     // row stride of the fp32 tile (floats): 4 g-groups -> 4 x 16 banks
  auto adam_tail = [&]() {
    if constexpr (ADAM) {
      const AdamFuse& ad = *adp;
      float step_size = ad.step_size, bc2_sqrt = ad.bc2_sqrt;
      if (ad.step_dev) {      // step kept on the device (replays): the same double-precision bias corrections as adam_kernel
        const double t = (double)*ad.step_dev;
        step_size = (float)((double)ad.lr / (1.0 - pow((double)ad.beta1, t)));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)ad.beta2, t));
      }
      __builtin_amdgcn_s_barrier();                             // (the compute waves have left their accumulators in LDS)
      const float* tile = reinterpret_cast<const float*>(smem);
      const int tid = wave * 64 + lane;
      const bool has_beta = p.beta != 0.f;
#pragma unroll 2
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 512 + tid, row = idx >> 5, c4 = idx & 31;
        f32x4 o = *reinterpret_cast<const f32x4*>(tile + row * TLD + 4 * c4);
        const long grow = 128 * tm + row, gcol = 128 * tn + 4 * c4;
        float* cp = p.C + grow * p.ldc + gcol;
        if (has_beta) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = o[j] + p.beta * old[j];
        }
        *reinterpret_cast<f32x4*>(cp) = o;
        const f32x4 pn = adam4(ad, step_size, bc2_sqrt, cp - ad.g, o);
        if (p.aux_out) {
          uint2 h2, l2;
          split4(pn, h2, l2);
          unsigned char* qd = reinterpret_cast<unsigned char*>(p.aux_out) + (((grow >> 5) * (p.ldc >> 5) + (gcol >> 5)) * 32 + (grow & 31)) * 128 + (gcol & 31) * 2;
          *reinterpret_cast<uint2*>(qd) = h2;
          *reinterpret_cast<uint2*>(qd + 64) = l2;
        }
      }
      // the bias: its gradient was stored by the column-0 tile's wave column 0 (below); the same lanes update it
      if (p.dbias != nullptr && tn == 0 && tid < 128) {
        float* bp = p.dbias + 128 * tm + tid;
        (void)adam1(ad, step_size, bc2_sqrt, bp - ad.g, reinterpret_cast<const float*>(smem)[128 * TLD + tid]);
      }
    }
  };
  // (row-compacted context head: the device-side count bounds the rows -- M of the forward, K of the weight gradient, whose
  //  operands are zero from the count up to the next multiple of 32)
  const int Mvalid = KIND == 2 ? p.M : dyn_limit(p, p.M);
  const int nk = KIND == 2 ? (dyn_limit(p, p.K) + 31) >> 5 : p.K >> 5;
  if (128 * tm >= Mvalid || (KIND == 2 && !ADAM && nk == 0 && p.beta != 0.f)) return;
  const unsigned lds0 = p2_lds_addr(smem);
  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    // (four loaders: eight -- two per SIMD -- were measured too and changed nothing, 70 us either way: the loaders wait ~800 cycles
    //  per k-step at the barrier for the compute waves, tools/micro/p3_bench.hip)
    const int lw = wave - 4;
    unsigned off2[2], b_off[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 8 * q + (lane >> 3);
      const int sc = (lane & 7) ^ ((r >> 1) & 7);
      off2[q] = (unsigned)r * 128u + 16u * sc;
    }
    // (KIND 1) request q of a step: image rows k = 8 lw + 2 q + (lane >> 5), LDS chunk lane & 31 <- source chunk sc
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 8 * lw + 2 * q + (lane >> 5), pos = lane & 31;
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      const int sc = (pos & ~15) | ((pos & 15) ^ f);
      b_off[q] = (unsigned)(sc >> 3) * 4096u + (unsigned)k * 128u + (unsigned)(sc & 7) * 16u;
    }
    const unsigned char* a_base = KIND == 2 ? reinterpret_cast<const unsigned char*>(p.A) + 4096L * (4 * tm)
                                            : reinterpret_cast<const unsigned char*>(p.A) + (long)(4 * tm + lw) * (p.lda >> 5) * 4096;
    const long a_step = KIND == 2 ? (long)(p.lda >> 5) * 4096 : 4096L;
    const unsigned char* b_base = KIND == 0 ? reinterpret_cast<const unsigned char*>(p.B) + (long)(4 * tn + lw) * (p.ldb >> 5) * 4096
                                            : reinterpret_cast<const unsigned char*>(p.B) + 4096L * (4 * tn);
    const long b_step = KIND == 0 ? 4096L : (long)(p.ldb >> 5) * 4096;
    const unsigned dstw = lds0 + (KIND == 2 ? (8 * lw) * 512 : (32 * lw) * 128);
    const unsigned dstb = lds0 + P3::BOFF + (KIND == 0 ? (32 * lw) * 128 : (8 * lw) * 512);
    // Rows GATHERED from q32b storage (GemmProblem::srow: the feature rows of layer 1 -- A of the forward, B of the weight
    // gradient -- fetched from a stored block or piece table, no staged copy).  Forward: this lane's four image rows as byte
    // addresses of their k-step-0 chunk (a k-step further = one 4-KiB column block further).  Weight gradient: the eight k-rows
    // this loader requests per step come through the scalar cache when the step is issued.
    const bool gather = p.srow != nullptr && KIND != 1;
    const unsigned char* arow[4] = {nullptr, nullptr, nullptr, nullptr};
    if (KIND == 0 && gather) {
      const int last = ((Mvalid + 31) & ~31) - 1;               // (the list is defined up to the next multiple of 32)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int img = 8 * q + (lane >> 3);
        int j = 128 * tm + 32 * lw + img;
        j = j < last ? j : last;
        const int sc = (lane & 7) ^ ((img >> 1) & 7);
        arow[q] = reinterpret_cast<const unsigned char*>(p.A) + p2_row_off(p.srow[j], p.lda) + 16 * sc;
      }
    }
    // (weight gradient, gathered B: per-lane offset of the request's chunk WITHOUT the row term, which comes from the list)
    unsigned g_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 8 * lw + 2 * q + (lane >> 5), pos = lane & 31;
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      const int sc = (pos & ~15) | ((pos & 15) ^ f);
      g_off[q] = (unsigned)(sc >> 3) * 4096u + (unsigned)(sc & 7) * 16u;
    }
    auto issue = [&](int t, int slot) {
      if constexpr ((ABL & 1) != 0) return;                     // diagnostics: no LDS-DMA at all
      const unsigned so = (unsigned)slot * P3::SLOT;
      if (KIND == 2 && gather) {
        const i32x4v r0 = p2_sload4(p.srow + 32 * t + 8 * lw), r1 = p2_sload4(p.srow + 32 * t + 8 * lw + 4);
        const int rows8[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int sidx = (lane >> 5) ? rows8[2 * q + 1] : rows8[2 * q];
          p2_dma16_v(b_base + p2_row_off(sidx, p.ldb) + g_off[q], dstb + so + q * 1024);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (KIND == 0) p2_dma16(b_base + b_step * t + (q >> 1) * 2048, off2[q & 1], dstb + so + q * 1024);
          else p2_dma16(b_base + b_step * t, b_off[q], dstb + so + q * 1024);
        }
      }
      if (KIND == 0 && gather) {
#pragma unroll
        for (int q = 0; q < 4; ++q) p2_dma16_v(arow[q] + 4096L * t, dstw + so + q * 1024);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (KIND == 2) p2_dma16(a_base + a_step * t, b_off[q], dstw + so + q * 1024);
          else p2_dma16(a_base + 4096L * t + (q >> 1) * 2048, off2[q & 1], dstw + so + q * 1024);
        }
      }
    };
    // steps 0 .. NSLOT - 2 up front; behind barrier t step t + NSLOT - 1 goes into the slot of step t - 1
    int islot = 0;
#pragma unroll
    for (int u = 0; u < P3::NSLOT - 1; ++u) {
      if (u < nk) issue(u, islot);
      islot = p3_next(islot);
    }
    for (int t = 0; t < nk; ++t) {
      // steps <= t + 1 have landed when all but the requests of the steps behind them are done (8 per step and loader)
      const int ahead = nk - 2 - t < P3::NSLOT - 3 ? nk - 2 - t : P3::NSLOT - 3;      // issued steps beyond t + 1
      long long s0 = 0, s1 = 0, s2 = 0;
      if constexpr ((ABL & 4) != 0) s0 = __builtin_readcyclecounter();
      if (ahead >= 2) p3_wait_vm<16>(); else if (ahead == 1) p3_wait_vm<8>(); else p3_wait_vm<0>();
      if constexpr ((ABL & 4) != 0) s1 = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      if constexpr ((ABL & 4) != 0) s2 = __builtin_readcyclecounter();
      if (t + P3::NSLOT - 1 < nk) issue(t + P3::NSLOT - 1, islot);
      islot = p3_next(islot);
      if constexpr ((ABL & 4) != 0) {
        // stamps of workgroup 0, loader wave 4: [t][0..3] = top, data landed, barrier passed, requests issued
        if (blockIdx.x == 0 && lw == 0 && lane == 0 && t < 128) {
          long long* st = reinterpret_cast<long long*>(p.slab) + 8 * t;
          st[0] = s0; st[1] = s1; st[2] = s2; st[3] = __builtin_readcyclecounter();
        }
      }
    }
    if constexpr (ADAM) {
      __builtin_amdgcn_s_barrier();                             // (every compute wave is through with the ring)
      adam_tail();
    }
    return;
  }
  // -------------------------------------------------------------------- compute waves
  const int wr = wave >> 1, wc = wave & 1, g = lane >> 4, l15 = lane & 15;
  const int frag = l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4);
  const int lo_d = 64 - 2 * (frag & 64);                     // lo address = hi address ^ 64
  const int a_frag = frag + (4 * wr) * 2048, b_frag = P3::BOFF + frag + (4 * wc) * 2048;
  // transposed fragment reads from a k-major image (KIND 1: B; KIND 2: A and B): two per fragment, rows 8 g + 4 t + q4
  const int q4 = l15 >> 2, pp = lane & 3;
  int tb[2], tx[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 8 * g + 4 * t + q4;
    const int f = (q4 << 2) | ((2 * g + t) & 3);
    tb[t] = k * 512 + 8 * (pp & 1);
    tx[t] = ((pp >> 1) ^ f) << 4;
  }
  // fragment n (16 columns) of the 64-column group `w` of the k-major image at `img`: hi into h, lo into l
  auto trfrag = [&](const unsigned char* img, int w, int n, bf16x8& h, bf16x8& l) {
    const int cb = ((n >> 1) & 1) * 8 + 2 * (n & 1);
    const unsigned char* q0 = img + w * 256 + tb[0];
    const unsigned char* q1 = img + w * 256 + tb[1];
    {
      const s16x4 x = lds_tr16(q0 + ((cb << 4) ^ tx[0])), y = lds_tr16(q1 + ((cb << 4) ^ tx[1]));
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      h = *reinterpret_cast<const bf16x8*>(&v);
    }
    {
      const s16x4 x = lds_tr16(q0 + (((cb | 4) << 4) ^ tx[0])), y = lds_tr16(q1 + (((cb | 4) << 4) ^ tx[1]));
      const s16x8 v = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      l = *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  // B fragment n of the slot at `sp`
  auto bfrag = [&](const unsigned char* sp, int n, bf16x8& h, bf16x8& l) {
    if constexpr (KIND == 0) {
      h = *reinterpret_cast<const bf16x8*>(sp + b_frag + n * 2048);
      l = *reinterpret_cast<const bf16x8*>(sp + b_frag + n * 2048 + lo_d);
    } else {
      trfrag(sp + P3::BOFF, wc, n, h, l);
    }
  };
  // A fragment i (16 rows) of the slot at `sp`
  auto afrag = [&](const unsigned char* sp, int i, bf16x8& h, bf16x8& l) {
    if constexpr (KIND == 2) {
      trfrag(sp, wr, i, h, l);
    } else {
      h = *reinterpret_cast<const bf16x8*>(sp + a_frag + i * 2048);
      l = *reinterpret_cast<const bf16x8*>(sp + a_frag + i * 2048 + lo_d);
    }
  };
  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // One k-step: multiplies from the B fragments in (bh, bl) and the A fragments 0, 1 in (ah[0..1], al[0..1]) -- all fetched
  // during the PREVIOUS step -- and fetches, behind its MFMA groups, the A fragments 2, 3 of this step and everything the next
  // step starts with (its B fragments into (nh, nl), its A fragments 0, 1): that step's slot landed before this step's barrier,
  // so no read waits behind a barrier.  (tools/micro/p3_bench.hip: with the first A fragments read behind the barrier and the
  // B registers rotated by moves the step took 1290 cycles against 768 of MFMA.)  Behind the last step the fetches read a stale
  // slot and are never used.
  bf16x8 ah[4], al[4];
  // (KIND 2) bias gradient: column sums of A over k = A fragment x ones, in the column-0 tiles, by the wave column 0
  const bool do_db = KIND == 2 && p.dbias != nullptr && tn == 0 && wc == 0;
  f32x4v accb[4];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  auto step = [&](const unsigned char* sp, const unsigned char* sn, const bf16x8 (&bh)[4], const bf16x8 (&bl)[4],
                  bf16x8 (&nh)[4], bf16x8 (&nl)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // reads of this group: one A fragment (this step's 2, 3, then the next step's 0, 1) and one B fragment of the next step
      bf16x8 xh, xl;
      if (i < 2) afrag(sp, i + 2, xh, xl); else afrag(sn, i - 2, xh, xl);
      bfrag(sn, i, nh[i], nl[i]);
      if constexpr ((ABL & 2) == 0) {
        // (ABL bit 8: gemm mode 3, BASELINE config 5's arithmetic -- ONE MFMA per product: bf16(a) bf16(b), fp32 accumulate)
        if constexpr ((ABL & 8) == 0) {
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[n], acc[i][n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[n], acc[i][n], 0, 0, 0);
        if constexpr (KIND == 2) {
          if (do_db) {
            accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], ones, accb[i], 0, 0, 0);
            accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], ones, accb[i], 0, 0, 0);
          }
        }
      } else {
        acc[i][0][0] += (float)ah[i][0] + (float)al[i][0] + (float)bh[i][0] + (float)bl[i][0];     // (keeps the reads alive)
      }
      // fragment i of this step is spent: its registers take the fetched A fragment (i < 2: fragment i + 2 of this step ...
      if (i < 2) { ah[i + 2] = xh; al[i + 2] = xl; } else { ah[i - 2] = xh; al[i - 2] = xl; }     // ... else i - 2 of the next)
      if constexpr (KIND != 2) {
        __builtin_amdgcn_sched_group_barrier(0x100, KIND == 0 ? 4 : 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (ABL & 8) ? 4 : 12, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  bf16x8 b0h[4], b0l[4], b1h[4], b1l[4];
  int cslot = 0;
  for (int t = 0; t < nk; t += 2) {
    long long c0 = 0, c1 = 0;
    if constexpr ((ABL & 4) != 0) c0 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    if constexpr ((ABL & 4) != 0) c1 = __builtin_readcyclecounter();
    const unsigned char* sp = smem + cslot * P3::SLOT;
    cslot = p3_next(cslot);
    const unsigned char* sn = smem + cslot * P3::SLOT;
    if (t == 0) {
      // the first step's fragments (the only reads that wait behind a barrier)
#pragma unroll
      for (int n = 0; n < 4; ++n) bfrag(sp, n, b0h[n], b0l[n]);
#pragma unroll
      for (int i = 0; i < 2; ++i) afrag(sp, i, ah[i], al[i]);
    }
    step(sp, sn, b0h, b0l, b1h, b1l);
    if constexpr ((ABL & 4) != 0) {
      // stamps of workgroup 0, compute wave 0: [t][4..6] = top, barrier passed, step done
      if (blockIdx.x == 0 && wave == 0 && lane == 0 && t < 128) {
        long long* st = reinterpret_cast<long long*>(p.slab) + 8 * t;
        st[4] = c0; st[5] = c1; st[6] = __builtin_readcyclecounter();
      }
    }
    if (t + 1 < nk) {
      __builtin_amdgcn_s_barrier();
      cslot = p3_next(cslot);
      step(sn, smem + cslot * P3::SLOT, b1h, b1l, b0h, b0l);
    }
  }

  // ---- epilogue (compute waves): element (i, n, j) -> row 128 tm + 64 wr + 16 i + 4 g + j, column 128 tn + 64 wc + 16 n + l15
  const bool drop = p.thresh != 0u;
  unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
  if (drop) apply_seed_offset(key_lo, key_hi, p.seed_dev);
  if constexpr (KIND == 0) {
    // dropout: p.aux, when given, holds the launch's KEEP BYTES (one byte per four rows and column, bit j = row 4 q + j kept:
    // written by the staging pass, gemm_p2.hpp); else the Philox words are drawn here, with the ORIGINAL row ids of a
    // row-mapped problem
    const unsigned char* keep = reinterpret_cast<const unsigned char*>(p.aux);
    const bool mapped = drop && !keep && p.rowmap != nullptr;
    float bias_n[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bias_n[n] = p.bias ? p.bias[128 * tn + 64 * wc + 16 * n + l15] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row4 = 128 * tm + 64 * wr + 16 * i + 4 * g;
      if (row4 >= Mvalid) continue;
      unsigned rid[4] = {0u, 0u, 0u, 0u};
      if (mapped) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rid[j] = (unsigned)p.rowmap[row4 + j < Mvalid ? row4 + j : Mvalid - 1];
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = 128 * tn + 64 * wc + 16 * n + l15;
        unsigned kb = 15u;
        if (drop && keep) {
          kb = keep[(long)(row4 >> 2) * p.ldaux + p.drop_col_off + col];
        } else if (drop) {
          unsigned w[4];
          if (!mapped) {
            philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
          } else {
            unsigned rnd[4];
            unsigned blk = rid[0] >> 2;
            philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if ((rid[j] >> 2) != blk) {
                blk = rid[j] >> 2;
                philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
              }
              const unsigned k = rid[j] & 3u;
              w[j] = k == 0u ? rnd[0] : (k == 1u ? rnd[1] : (k == 2u ? rnd[2] : rnd[3]));
            }
          }
          kb = 0u;
#pragma unroll
          for (int j = 0; j < 4; ++j) kb |= (w[j] >= p.thresh ? 1u : 0u) << j;
        }
        float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaxf(acc[i][n][j] + bias_n[n], 0.f);
          if (drop) v = ((kb >> j) & 1u) ? v * p.drop_scale : 0.f;
          if (row4 + j < Mvalid) cp[(long)j * p.ldc] = v;
        }
      }
    }
  } else if constexpr (KIND == 2 && ADAM) {
    __builtin_amdgcn_s_barrier();                               // every compute wave is through with the ring: it becomes the tile
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[(64 * wr + 16 * i + 4 * g + j) * TLD + 64 * wc + 16 * n + l15] = acc[i][n][j];
    if (do_db && l15 == 0) {
      // (the bias gradient: final value to global memory AND behind the tile, for the lanes that update the bias)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 64 * wr + 16 * i + 4 * g + j;
          const float db = p.dbias_set ? accb[i][j] : p.dbias[128 * tm + r] + accb[i][j];
          p.dbias[128 * tm + r] = db;
          tile[128 * TLD + r] = db;
        }
    }
    adam_tail();
  } else if constexpr (KIND == 2) {
    const bool has_beta = p.beta != 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row4 = 128 * tm + 64 * wr + 16 * i + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float* cp = p.C + (long)row4 * p.ldc + 128 * tn + 64 * wc + 16 * n + l15;
        float old[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_beta) {
#pragma unroll
          for (int j = 0; j < 4; ++j) old[j] = cp[(long)j * p.ldc];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cp[(long)j * p.ldc] = acc[i][n][j] + p.beta * old[j];
      }
      if (do_db && l15 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p.dbias[row4 + j] = p.dbias_set ? accb[i][j] : p.dbias[row4 + j] + accb[i][j];
      }
    }
  } else {
    const bool has_beta = p.beta != 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row4 = 128 * tm + 64 * wr + 16 * i + 4 * g;
      if (row4 >= Mvalid) continue;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = 128 * tn + 64 * wc + 16 * n + l15;
        float ax[4], old[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = row4 + j < Mvalid ? row4 + j : Mvalid - 1;
          ax[j] = p.aux[(long)r * p.ldaux + col];
          if (has_beta) old[j] = p.C[(long)r * p.ldc + col];
        }
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (drop) philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
        float* cp = p.C + (long)row4 * p.ldc + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[i][n][j] + p.beta * old[j];
          const bool keep = !drop || w[j] >= p.thresh;
          const float f = 1.f - ax[j] * ax[j];
          v *= keep ? f * p.drop_scale : 0.f;
          if (row4 + j < Mvalid) cp[(long)j * p.ldc] = v;
        }
        __asm__ volatile("" ::: "memory");
      }
    }
  }
}

// grid = tiles of all problems (every problem M % 128 == 0 ... the host guarantees N % 128 == 0, K % 32 == 0 and pads M).
// Tiles are dealt to the XCDs column-major: consecutive logical ids walk the row tiles of one column tile.
template <int KIND, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p3_kernel(const GemmGroup g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P3::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  int L = b;
  if ((G & 7) == 0) L = (b & 7) * (G >> 3) + (b >> 3);
  int first = 0;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProblem& p = g.p[i];
    const int tms = (p.M + 127) >> 7, tns = p.N >> 7;
    if (L >= first && L < first + tms * tns) {
      const int tn = (L - first) / tms, tm = (L - first) - tn * tms;
      p3_tile<KIND, ABL>(p, smem, tm, tn, lane, wave);
      return;
    }
    first += tms * tns;
  }
}

// the gate's weight gradient with the parameters' update in the epilogue (one problem)
template <int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p3_tn_adam_kernel(const GemmGroup g, const AdamFuse ad) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P3::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  int L = b;
  if ((G & 7) == 0) L = (b & 7) * (G >> 3) + (b >> 3);
  const GemmProblem& p = g.p[0];
  const int tms = (p.M + 127) >> 7;
  const int tn = L / tms, tm = L - tn * tms;
  p3_tile<2, ABL, true>(p, smem, tm, tn, lane, wave, &ad);
}

// GROUPED launches (layer 1 and its weight gradient: up to 8 problems of different depths).  grid = every tile of every problem,
// one per workgroup; the hardware hands out workgroups in blockIdx order as CUs fall free, so the order of the tiles IS the
// schedule (longest first = list scheduling), and what an XCD's L2 sees follows from which blockIdx values share an XCD (b % 8):
//   * a UNIT = the g.p3_nc tiles that stream the same rows of the launch's big operand (the feature rows: forward = the column
//     tiles of one row tile; weight gradient = the row tiles of one column tile).  A unit's tiles take consecutive slots of ONE
//     XCD -- they start together and walk k together, so those rows leave HBM once -- and consecutive units go to the eight XCDs in
//     turn, so every XCD gets the same mix of long and short tiles;
//   * NT: units are (row tile, problem), ROW TILE MAJOR: with row compaction the tiles that have rows are a prefix of the grid and
//     the rest leave at once.  The host lists the problems deepest first; problems whose bit is set in g.p3_tall have row tiles
//     beyond g.p3_ta (context head beside interaction head);
//   * TN: units are (problem, column tile), problem major, the host lists the long reductions first.
template <int KIND, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p3g_kernel(const GemmGroup g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[P3::LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, nc = g.p3_nc;
  const int s = b >> 3, u = 8 * (s / nc) + (b & 7), c = s - (s / nc) * nc;
  if constexpr (KIND == 0) {
    const int nall = g.nprob, ntall = __builtin_popcount((unsigned)g.p3_tall);
    int tm, pi;
    if (u < g.p3_ta * nall) { tm = u / nall; pi = u - tm * nall; }
    else {
      if (ntall == 0) return;
      const int v = u - g.p3_ta * nall;
      tm = g.p3_ta + v / ntall;
      int k = v - (v / ntall) * ntall;
      pi = 0;
      for (int i = 0; i < LIREC_MAX_PROB; ++i)
        if ((g.p3_tall >> i) & 1) { if (k == 0) { pi = i; break; } --k; }
    }
    if (pi >= g.nprob || 128 * tm >= g.p[pi].M) return;
    p3_tile<0, ABL>(g.p[pi], smem, tm, c, lane, wave);
  } else {
    int first = 0;
    for (int i = 0; i < g.nprob; ++i) {
      const int tns = g.p[i].N >> 7;
      if (u >= first && u < first + tns) { p3_tile<KIND, ABL>(g.p[i], smem, c, u - first, lane, wave); return; }
      first += tns;
    }
  }
}

}  // namespace lirec
