// Grouped fp32 GEMM for gfx950 on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Every dense feature x weight contraction of the hot path (K1, K2,
// K4, the head GEMMs and all backward GEMMs K6) goes through this one kernel
// family.  f32-input MFMA is exact fp32 (a k-ordered fma chain), which is what
// holds the 1e-4 parity bar against the reference's fp32 CPU path; it runs at
// the fp32 vector rate (157 TF peak), so these kernels are MFMA-bound, not
// HBM-bound (roofline section).
//
// Layouts (reduction index k):
//   NT  C[m,n] = sum_k A[m,k] * B[n,k]     forward  Y = X W^T       (A, B k-contiguous)
//   NN  C[m,n] = sum_k A[m,k] * B[k,n]     backward dX = dY W       (A k-contig, B n-contig)
//   TN  C[m,n] = sum_k A[k,m] * B[k,n]     backward dW = dY^T X     (A m-contig, B n-contig)
// The "X operand" (A in NT, B in TN) may select its rows out of the (B*T, R+1, D)
// feature block through lirec_rowsel, so the feature tensor is read in place
// (no slice/copy, cf. mlp/model.py:279-290).
//
// Tiling: 256 threads = 4 waves (2x2); a wave owns WM x WN MFMA tiles of 32x32;
// block tile = (64*WM) x (64*WN) x 32.  LDS tiles are k-major ([k][m], [k][n]) so
// a fragment read is 32 consecutive dwords per half-wave (conflict-free
// ds_read_b32); k-contiguous operands are transposed on the LDS write (pitch
// = tile+1 makes the 8-lanes-per-row write pattern conflict-free), m/n-contiguous
// operands are copied with ds_write_b128 (pitch = tile+4 keeps 16-B alignment).
// Global loads are register-staged one k-tile ahead (issue before the MFMA
// block, write to the other LDS buffer after it; one barrier per k-tile).
// All edges (M, N, K) are predicated with zero fill, so odd sizes (C=101,
// NR=15, reduced-dim tests) take the same kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lirec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { L_NT = 0, L_NN = 1, L_TN = 2 };
enum { EPI_STORE = 0, EPI_DROP_RELU = 1, EPI_TANH_DROP = 2, EPI_RELU_BWD = 3, EPI_TANH_BWD = 4 };

struct GemmProblem {
  const float* A; const float* B; float* C;
  const float* bias;      // [N], added before the epilogue (NT forward)
  const float* rowscale;  // optional per-row factor: NT/NN bias is added as bias[col] * rowscale[row];
                          // TN bias gradient sums A[k,m] * rowscale[k]   (pooled context head: f = cnt/div)
  const float* aux;       // epilogue operand [M,N] (forward activation for *_BWD)
  float* aux_out;         // EPI_TANH_DROP: tanh before dropout
  float* dbias;           // TN only: dbias[m] += sum_k A[k,m]
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int gs, gstride, goff;  // row selection of the X operand (A rows in NT, B rows in TN)
  unsigned gs_magic;      // floor(2^32 / gs) + 1 when n / gs == umulhi(n, gs_magic) for every row id of the problem, else 0
  int epi; float beta;    // beta: existing C is added (times beta) before the epilogue factor
  int dbias_set;          // TN: 1 = dbias[m] is OVERWRITTEN with the sum (gradients of a step that skips the zeroing pass), 0 = +=
  unsigned seed_lo, seed_hi, site, thresh; float drop_scale; int drop_col_off;
  const unsigned long long* seed_dev;   // optional device counter added to the key (lirec_dropout::seed_dev)
  int x_bf16;                           // the X operand (A in NT, B in TN) is stored as bf16 (lda / ldb in elements)
  int tiles_n, tile_start;
  // split-K (host-chosen): the K range is cut into `ksplit` chunks of `kchunk` (multiple of 32);
  // a workgroup handles one (chunk, tile) and, when ksplit > 1, writes its raw partial tile to
  // slab[chunk][M][N] (and the partial bias gradient to dbias_slab[chunk][M]); splitk_reduce_kernel
  // then applies bias / += C.  Only EPI_STORE problems are split.
  int ksplit, kchunk, tiles_mn;
  float* slab; float* dbias_slab;
  // Row compaction (context head): only the context rows whose mask is non-zero are processed.
  //   rowmap[j] = logical row id of compact row j (ascending); the X operand's row j is then
  //               phys_row(rowmap[j]) and dropout counters use rowmap[j], so every value equals the
  //               uncompacted computation;
  //   dyn       = device pointer to the number of compact rows: it bounds M (NT/NN) or K (TN) at run
  //               time, so the host never has to read the count back (no sync); the grid is sized
  //               for the full row count and surplus workgroups leave at once.
  const int* rowmap; const int* dyn;
  // gemm_p2 kernels, rows GATHERED from a q32b matrix (no staged copy): logical row j of the row operand (A of NT, B of TN)
  // is storage row srow[j] of the q32b matrix at A / B (lda / ldb = its columns); srow has an entry for every row up to
  // the next multiple of 32 (the tail repeats the last valid row).  NULL = the rows are dense.
  const int* srow;
  // Pre-split bf16 planes (gemm_p2.hpp): A / B then point at the HI planes (bf16 elements, lda / ldb in
  // elements) and these at the LO planes (NULL for a bf16-stored operand, whose low half is zero).
  const void* A_lo; const void* B_lo;
  // gemm_p2's forward on rows fetched from the fp32 block (p2_nt_tile, XF): where the q32b form of the rows it splits goes (a
  // q32b matrix of ld_xq columns, at the problem's first column block; compact row j -> its row j), or NULL
  unsigned char* xq_out; long ld_xq;
};

// Adam over four (one) elements at `off` of the flat parameter / moment buffers (torch.optim.Adam's single-tensor op order): the
// ONE definition both adam_kernel and the reduce kernel that folds the first-layer bucket's update in (gemm_p2.hpp) are built
// from, so that the two give the same bits.
struct AdamFuse {
  float* p; const float* g; float* m; float* v;            // flat buffers (same layout; g = the buffer the problems' C point into)
  float step_size, bc2_sqrt, beta1, beta2, eps, wd, gscale, lr;
  const long long* step_dev;
  int wq16c;                                                // the shadow of the new weights (GemmProblem::aux_out) is q16c, not q32b
};
// (no floating-point contraction inside: whether the compiler forms an fma here would otherwise depend on the kernel the
//  function is inlined into, and the two kernels must agree to the bit)
__device__ __forceinline__ float adam1(const AdamFuse& ad, float step_size, float bc2_sqrt, long off, float g) {
#pragma clang fp contract(off)
  const float pp = ad.p[off];
  const float gg = g * ad.gscale + ad.wd * pp;
  const float mm = ad.m[off] + (1.f - ad.beta1) * (gg - ad.m[off]);
  const float vv = ad.v[off] * ad.beta2 + (1.f - ad.beta2) * gg * gg;
  ad.m[off] = mm; ad.v[off] = vv;
  const float pn = pp - step_size * (mm / (sqrtf(vv) / bc2_sqrt + ad.eps));
  ad.p[off] = pn;
  return pn;
}
__device__ __forceinline__ f32x4 adam4(const AdamFuse& ad, float step_size, float bc2_sqrt, long off, const f32x4 gv) {
#pragma clang fp contract(off)
  f32x4 pv = *reinterpret_cast<const f32x4*>(ad.p + off);
  f32x4 mv = *reinterpret_cast<const f32x4*>(ad.m + off);
  f32x4 vv = *reinterpret_cast<const f32x4*>(ad.v + off);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gg = gv[j] * ad.gscale + ad.wd * pv[j];
    mv[j] = mv[j] + (1.f - ad.beta1) * (gg - mv[j]);
    vv[j] = vv[j] * ad.beta2 + (1.f - ad.beta2) * gg * gg;
    const float denom = sqrtf(vv[j]) / bc2_sqrt + ad.eps;
    pv[j] = pv[j] - step_size * (mv[j] / denom);
  }
  *reinterpret_cast<f32x4*>(ad.p + off) = pv;
  *reinterpret_cast<f32x4*>(ad.m + off) = mv;
  *reinterpret_cast<f32x4*>(ad.v + off) = vv;
  return pv;
}

#define LIREC_MAX_PROB 8
// row_tiles > 0: every problem has the same tiles_m and ksplit and the tiles are ordered
// (split, tm, problem, tn) -- `row_tiles` = sum of tiles_n -- so that row panel tm of ALL problems is
// adjacent: with row compaction the valid work is then one contiguous prefix of the grid (dealt evenly
// to the XCDs) instead of a prefix of every problem's own range.
// Two tiers (tier_rows > 0 and row_tiles2 > 0, unsplit problems only): problems [0, first2) have tier_rows row
// panels, problems [first2, nprob) have more; rows tm < tier_rows hold every problem (row_tiles tiles each), the
// rows after them only the taller problems (row_tiles2 tiles each).  This is how the interaction head's layer 1
// (B*T rows) rides in the context head's launch (B*T*R rows).
struct GemmGroup {
  int nprob; int total_tiles; int ablate; int row_tiles;
  const int* nt_bound;    // gemm_p2_nt_kernel: the partition bound, when a staging launch has computed it (p2_partition.hpp)
  int nt_bound_val;       // ... or computed on the host (static row counts: p2_nt_bound_host); 0 = none
  // gemm_p3_kernel (gemm_p3.hpp): the tile space -- p3_tm row tiles x p3_tn column tiles (the problems side by side along the
  // columns) -- and how it is dealt to the XCDs: p3_xm x (8 / p3_xm) rectangular blocks, one per XCD (0 = plain column-major order)
  int p3_tm, p3_tn, p3_xm;
  int nt_ct_major;        // gemm_p2_nt_kernel: workgroup order (column tile, row chunk) instead of (row chunk, column tile): the
                          // workgroups of one XCD then share a WEIGHT panel (few rows, wide weights: the gate GEMMs)
  int onepass;            // bf16 core: single MFMA pass (operands rounded to bf16 once) -- gemm mode 3
  int tier_rows, row_tiles2, first2;
  // pgroup = G > 0 (all problems of a tier have the same tiles_n, G * tiles_n = 32): inside a tier the order is
  // (block of G row panels, problem, panel in block, tn) instead of (panel, problem, tn), and the XCD remap deals
  // 32-tile supertiles: the 32 workgroups an XCD starts together are then G panels x tiles_n column tiles of ONE
  // problem -- same K, so they walk k in step and a k-slice of the weight panel fetched by one is an L2 hit for
  // the other G - 1 -- and the problem order rotates from block to block so that every XCD still gets every
  // problem (the modality segments have different K).  Valid work of a row-compacted launch stays a prefix.
  int pgroup;
  // tm_fast (weight-gradient launches, row_tiles > 0, one tier): order (split, problem, tn, tm) instead of
  // (split, tm, problem, tn).  The tiles_m tiles that read the same k-chunk x column block of the X operand are
  // then adjacent workgroups -- same XCD, same moment -- and the block comes from HBM once, not tiles_m times.
  int tm_fast;
  // Run-time split of a row-compacted weight gradient (see effective_ksplit): tiles of the whole launch per k-chunk,
  // workgroups resident at once, cost of one partial slab (write + read by the reduce) and of the reduce launch, both
  // in units of one k-tile of this tile shape; dyn_is_k: GemmProblem::dyn bounds K (TN launches)
  int tiles_per_split, resident_slots, dyn_is_k;
  int dyn_split;          // host: every problem shares ksplit > 1 and the same `dyn`, tiles ordered split-major across the group
  float slab_cost, reduce_cost;
  GemmProblem p[LIREC_MAX_PROB];
};
struct GemmMeta { int site; int tag; };   // host-side only: profile site, kernel tag

// host: may this problem use the dwordx4 staging path?  (see raw4)
inline bool gemm_problem_is_vec(int layout, const GemmProblem& p) {
  const bool a_kc = layout != 2, b_kc = layout == 0;
  // (a bf16 X operand is staged with 8-byte loads: 8-byte alignment is enough for it)
  const uintptr_t ma = (p.x_bf16 && layout == 0) ? 7 : 15, mb = (p.x_bf16 && layout == 2) ? 7 : 15;
  const bool va = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & ma) == 0) && (((a_kc ? p.K : p.M) & 3) == 0);
  const bool vb = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & mb) == 0) && (((b_kc ? p.K : p.N) & 3) == 0);
  return va && vb;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (same definition as oracle/lirec_oracle.py:philox4x32_10)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                        unsigned k0, unsigned k1, unsigned out[4]) {
  // rolled (two rounds per trip: no register rotation moves) on purpose: the epilogues inline this up to 5 x 32 times
  // per kernel, and fully unrolled copies made the epilogue so large that hipcc stopped unrolling the accumulator
  // loops (accumulators in scratch memory)
#pragma unroll 2
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32): written as mul-high + mul-low hipcc issues two
    // quarter-rate instructions per product, and the epilogue's Philox calls were 18 % of the layer-1 launch
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// key = seed + *seed_dev (64-bit) when a device-resident offset is given
__device__ __forceinline__ void apply_seed_offset(unsigned& lo, unsigned& hi, const unsigned long long* seed_dev) {
  if (seed_dev) {
    const unsigned long long k = (((unsigned long long)hi << 32) | lo) + *seed_dev;
    lo = (unsigned)(k & 0xffffffffull); hi = (unsigned)(k >> 32);
  }
}

__device__ __forceinline__ bool epi_uses_dropout(const GemmProblem& p) {
  return p.thresh != 0u && (p.epi == EPI_DROP_RELU || p.epi == EPI_TANH_DROP || p.epi == EPI_TANH_BWD);
}

// One output element.  `rnd` is the Philox word of (row, col) when dropout is on.
// relu as torch computes it (clamp_min(0): mlp/model.py:281,286,291-292,353): a NaN stays a NaN.  fmaxf(NaN, 0) -- v_max_f32 -- returns 0:
// a non-finite feature in a VALID row, or the 0 / 0 of an all-masked clip of the model without the divider clamp (:175), would be
// silently swallowed at the next relu instead of reaching the loss as it does in the reference (tests/test_gpu_nonfinite.py).
__device__ __forceinline__ float relu_f(float v) { return v > 0.f ? v : (v != v ? v : 0.f); }

__device__ __forceinline__ void epi_store(const GemmProblem& p, int row, int col, float acc, unsigned rnd) {
  float v = acc;
  if (p.bias) v += p.rowscale ? p.bias[col] * p.rowscale[row] : p.bias[col];
  float* cptr = p.C + (long)row * p.ldc + col;
  if (p.beta != 0.f) v += p.beta * (*cptr);
  switch (p.epi) {
    case EPI_DROP_RELU:
      v = relu_f(v);
      if (p.thresh) v = (rnd >= p.thresh) ? v * p.drop_scale : 0.f;
      break;
    case EPI_TANH_DROP: {
      const float t = tanhf(v);
      p.aux_out[(long)row * p.ldaux + col] = t;
      v = t;
      if (p.thresh) v = (rnd >= p.thresh) ? t * p.drop_scale : 0.f;
      break;
    }
    case EPI_RELU_BWD: {
      const float a = p.aux[(long)row * p.ldaux + col];
      v = (a > 0.f) ? v * p.drop_scale : 0.f;
      break;
    }
    case EPI_TANH_BWD: {
      const float t = p.aux[(long)row * p.ldaux + col];
      float f = 1.f - t * t;
      if (p.thresh) f = (rnd >= p.thresh) ? f * p.drop_scale : 0.f;
      v *= f;
      break;
    }
    default: break;
  }
  *cptr = v;
}

__device__ __forceinline__ int dyn_limit(const GemmProblem& p, int full) {
  if (!p.dyn) return full;
  const int d = *p.dyn;
  return d < full ? d : full;
}

// physical row of logical row id n (the row-selector arithmetic alone, no row map)
__device__ __forceinline__ long sel_row(const GemmProblem& p, int n) {
  if (p.gs == 0) return n;
  const unsigned q = p.gs_magic ? __umulhi((unsigned)n, p.gs_magic) : (unsigned)n / (unsigned)p.gs;
  const unsigned r = (unsigned)n - q * (unsigned)p.gs;
  return (long)q * p.gstride + r + p.goff;
}

__device__ __forceinline__ long phys_row(const GemmProblem& p, int n) {
  if (p.rowmap) n = p.rowmap[n];
  return sel_row(p, n);
}

// Staging loads are split in two so that NOTHING consumes a load until the MFMA block that
// follows it has been issued (a per-load "load or zero" select or branch makes hipcc wait
// vmcnt(0) after every load and serialises the burst at full memory latency --
// cdna_hip_programming.md, "Three .s-level traps" (c)):
//   raw4()  issues the load(s) from an always-valid address (out-of-bounds chunks read `safe`);
//   mask4() zeroes the out-of-bounds elements later, when the tile is written to LDS.
// VEC is a property of the whole launch, decided on the host (gemm_problem_is_vec) and compiled
// in: both operands' rows 16-B aligned and their contiguous extents multiples of 4, so a chunk is
// either fully inside or fully outside and one dwordx4 load serves it.  The !VEC build takes any
// alignment with four scalar loads per chunk (only the small odd-sized head GEMMs use it).
template <bool VEC>
__device__ __forceinline__ f32x4 raw4(const float* ptr, int nvalid, const float* safe) {
  if constexpr (VEC) return *reinterpret_cast<const f32x4*>((nvalid >= 4) ? ptr : safe);
  f32x4 v;
  v.x = *((nvalid > 0) ? ptr : safe);
  v.y = *((nvalid > 1) ? ptr + 1 : safe);
  v.z = *((nvalid > 2) ? ptr + 2 : safe);
  v.w = *((nvalid > 3) ? ptr + 3 : safe);
  return v;
}
// four bf16 elements (8 bytes) of a bf16-stored operand, carried in the first two lanes of the chunk register;
// `ebase` + element offset, always-valid fallback `safe` (dwordx2 staging builds only: chunks are all-or-nothing)
__device__ __forceinline__ f32x4 raw4_bf16(const void* ebase, long eoff, bool ok, const void* safe) {
  const char* ptr = ok ? reinterpret_cast<const char*>(ebase) + 2 * eoff : reinterpret_cast<const char*>(safe);
  const uint2 w = *reinterpret_cast<const uint2*>(ptr);
  f32x4 v;
  // (__builtin_bit_cast on a vector ELEMENT miscompiles with this hipcc: element 0 is used for every lane)
  v.x = __uint_as_float(w.x); v.y = __uint_as_float(w.y); v.z = 0.f; v.w = 0.f;
  return v;
}
__device__ __forceinline__ float bf16_at(const void* base, long eoff) {
  const unsigned short h = reinterpret_cast<const unsigned short*>(base)[eoff];
  return __builtin_bit_cast(float, (unsigned)h << 16);
}

__device__ __forceinline__ f32x4 mask4(f32x4 v, int nvalid) {
  v.x = (nvalid > 0) ? v.x : 0.f; v.y = (nvalid > 1) ? v.y : 0.f;
  v.z = (nvalid > 2) ? v.z : 0.f; v.w = (nvalid > 3) ? v.w : 0.f;
  return v;
}

template <int ST = 64>
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // Workgroups are dealt round-robin over the 8 XCDs (bid % 8 = XCD slot, bid / 8 = its
  // sequence number there) and each XCD has its own 4 MiB L2.  Logical tiles are handed out
  // in SUPERTILES of 64 consecutive tiles (with tn fastest: 16 row panels x 4 column tiles):
  // the j-th supertile goes to XCD j % 8.  The 64 tiles of a supertile are what one XCD's 32
  // CUs hold at once (2 workgroups per CU), they start together and walk k in step, so a
  // k-slice of an operand panel fetched by one of them is an L2 hit for the others; and every
  // XCD gets the same mix of problems of a grouped launch (a contiguous range per XCD would
  // hand the short-K problems to some XCDs and the long-K ones to others).  Bijective for any
  // nwg: the last partial group of 8 supertiles is left in launch order.
  const int full = nwg - nwg % (8 * ST);
  const int xcd = bid & 7;
  if (bid < full) {
    const int seq = bid >> 3;
    return ((seq / ST) * 8 + xcd) * ST + seq % ST;
  }
  // the last partial group (or a grid smaller than 8 supertiles, e.g. the split-K weight-gradient
  // launches): one contiguous range of logical tiles per XCD, so neighbours still share an L2.
  // (PMC before this: 25 % L2 hit rate and 2.4x the unique bytes fetched by the dW1 launch, whose
  // 486 tiles all fell into this branch and were being dealt round-robin.)
  const int n = nwg - full, q = n >> 3, r = n & 7;
  const int idx = (bid - full) >> 3;
  return full + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// How many of the host-planned `ksplit` k-chunks a row-compacted weight gradient really uses.  The host sizes the grid
// and the slabs for the STATIC row count (it never reads the device-side count), typically 2-3x the rows that exist;
// with the real K every workgroup re-runs the host's cost rule -- rounds of resident workgroups x k-tiles per chunk +
// the partial slabs' round trip -- and the chunks beyond the chosen split leave at once (they are the tail of the grid:
// split is the slowest tile index).  The reduce kernel makes the same choice and sums only those slabs.
// (At the bench shape: 9 planned chunks of 25 k-tiles in two rounds of workgroups -> 4 chunks of 56 in one round.)
__device__ __forceinline__ int effective_ksplit(const GemmGroup& g, const GemmProblem& p, int K) {
  // (measured null on the bench step, +-0.3 %: the rule stays available as a diagnostic, lirec_debug_set bit 64)
  if (p.ksplit <= 1 || !g.dyn_split || !(g.ablate & 64)) return p.ksplit;
  int best = 1;
  float bestc = 3.0e38f;
  for (int ks = 1; ks <= p.ksplit; ++ks) {
    const int nk = ((K + ks - 1) / ks + 31) / 32;
    if (ks > 1 && nk < 8) break;
    const int rounds = (g.tiles_per_split * ks + g.resident_slots - 1) / g.resident_slots;
    const float c = (float)rounds * (float)nk + (ks > 1 ? g.reduce_cost + g.slab_cost * (float)(ks + 1) : 0.f);
    if (c < bestc) { bestc = c; best = ks; }
  }
  return best;
}

// Logical tile of this workgroup, or -1 when it has nothing to do.  With a run-time split (GemmGroup::dyn_split) only the
// first tiles_per_split x ks workgroups of the grid work, and the XCD remap is taken over THAT count: the working
// workgroups are the first ones dispatched, i.e. dealt round-robin over all eight XCDs (remapping over the full grid
// put the surviving chunks -- a prefix of the logical tiles -- on the first few XCDs only).
__device__ __forceinline__ int launch_tile(const GemmGroup& g) {
  int nwg = gridDim.x;
  if (g.dyn_split && g.dyn_is_k && (g.ablate & 64)) {
    const GemmProblem& p0 = g.p[0];
    const int ks = effective_ksplit(g, p0, dyn_limit(p0, p0.K));
    nwg = g.tiles_per_split * ks;
    if ((int)blockIdx.x >= nwg) return -1;
  }
  return g.pgroup > 0 ? xcd_remap<32>(blockIdx.x, nwg) : xcd_remap<64>(blockIdx.x, nwg);
}

// tile -> (problem, k-chunk, tm, tn); shared by both GEMM cores
struct TileCoord { int pi, split, m0, n0, tn, k_begin, k_end, M; };
// `dyn_is_k`: the run-time row count (GemmProblem::dyn) bounds K (TN) instead of M (NT/NN)
template <int BM, int BN>
__device__ __forceinline__ TileCoord decode_tile(const GemmGroup& g, int tile, bool dyn_is_k) {
  TileCoord c;
  c.pi = 0;
  int tm;
  if (g.pgroup > 0) {
    // (tier, block of G panels, problem (rotated), panel in block, tn); only the last block of a tier can be short
    const int G = g.pgroup, tier1 = g.tier_rows * g.row_tiles;
    const bool t2 = g.row_tiles2 > 0 && tile >= tier1;
    const int first = t2 ? g.first2 : 0, np = t2 ? g.nprob - g.first2 : g.nprob;
    const int rt = t2 ? g.row_tiles2 : g.row_tiles;
    const int rows = t2 ? g.p[first].tiles_mn / g.p[first].tiles_n - g.tier_rows : g.tier_rows;
    const int t = t2 ? tile - tier1 : tile;
    const int tn_each = g.p[first].tiles_n;
    const int blk = t / (G * rt);
    const int within = t - blk * G * rt;
    const int per = min(G, rows - blk * G) * tn_each;
    const int q = within / per, r2 = within - q * per;
    const int rot = (blk * np) / 8 + (t2 ? 1 : 0);
    c.pi = first + (q + rot) % np;
    const int pr = r2 / tn_each;
    c.tn = r2 - pr * tn_each;
    c.split = 0;
    tm = (t2 ? g.tier_rows : 0) + blk * G + pr;
  } else if (g.row_tiles > 0) {
    // interleaved order: tile = (split * tiles_m + tm) * row_tiles + (prefix of tiles_n) + tn;
    // here tile_start holds each problem's offset inside a row of tiles
    const int tier1 = g.tier_rows * g.row_tiles;
    if (g.row_tiles2 > 0 && tile >= tier1) {              // second tier: the taller problems only, never split
      const int t2 = tile - tier1;
      const int row2 = t2 / g.row_tiles2;
      const int rem = t2 - row2 * g.row_tiles2 + g.p[g.first2].tile_start;
      c.pi = g.first2;
#pragma unroll
      for (int i = 1; i < LIREC_MAX_PROB; ++i)
        if (i > g.first2 && i < g.nprob && rem >= g.p[i].tile_start) c.pi = i;
      c.tn = rem - g.p[c.pi].tile_start;
      c.split = 0;
      tm = g.tier_rows + row2;
    } else {
      const int tiles_m = g.p[0].tiles_mn / g.p[0].tiles_n;
      int row, rem;
      if (g.tm_fast) {
        const int per_split = tiles_m * g.row_tiles;
        c.split = tile / per_split;
        const int r = tile - c.split * per_split;
        rem = r / tiles_m;
        tm = r - rem * tiles_m;
      } else {
        row = tile / g.row_tiles;
        rem = tile - row * g.row_tiles;
        c.split = row / tiles_m;
        tm = row - c.split * tiles_m;
      }
#pragma unroll
      for (int i = 1; i < LIREC_MAX_PROB; ++i)
        if (i < g.nprob && rem >= g.p[i].tile_start) c.pi = i;
      c.tn = rem - g.p[c.pi].tile_start;
    }
  } else {
#pragma unroll
    for (int i = 1; i < LIREC_MAX_PROB; ++i)
      if (i < g.nprob && tile >= g.p[i].tile_start) c.pi = i;
    int t = tile - g.p[c.pi].tile_start;
    c.split = t / g.p[c.pi].tiles_mn;
    t -= c.split * g.p[c.pi].tiles_mn;
    tm = t / g.p[c.pi].tiles_n;
    c.tn = t - tm * g.p[c.pi].tiles_n;
  }
  const GemmProblem& p = g.p[c.pi];
  c.m0 = tm * BM; c.n0 = c.tn * BN;
  int K = p.K, kchunk = p.kchunk;
  c.M = p.M;
  if (p.dyn) {
    if (dyn_is_k) {                         // split the rows that exist evenly over the k-chunks that pay
      K = dyn_limit(p, p.K);
      const int ks = effective_ksplit(g, p, K);
      kchunk = ((K + ks - 1) / ks + 31) / 32 * 32;
      if (c.split >= ks) c.M = 0;           // an unused chunk: the workgroup leaves at `m0 >= M`, its slab is never read
    } else {
      c.M = dyn_limit(p, p.M);
    }
  }
  c.k_begin = min(K, c.split * kchunk);
  c.k_end = min(K, c.k_begin + kchunk);
  return c;
}

// accumulator tiles -> memory.  The 32x32 C/D layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)) is the same for the f32 and the bf16 MFMA, so both cores share this.
// One epilogue kind for a wave's WM x WN accumulator tiles.  Per (tile, 4-row group): everything the four
// elements need from memory (old C when beta != 0, the saved activation, the row scale) is loaded up front from
// always-valid addresses, then the arithmetic, then the four stores: one memory wait per group instead of one
// per element.  bias[col] is loaded once per tile column.
// rid_lane (optional, row-mapped dropout only): the ORIGINAL row id of row (m0 + wm0 + lane) of this wave's tile,
// fetched by the caller long before the epilogue; without it the ids are loaded here, one dependent global load
// per 4-row group in front of the Philox call and the stores.
template <int WM, int WN, int EPI>
__device__ __forceinline__ void gemm_epilogue_kind(const GemmProblem& p, const f32x16 (&acc)[WM][WN], int m0, int n0,
                                                   int wm0, int wn0, int lane, int M, bool have_rid = false, int rid_lane = 0) {
  constexpr bool USES_RND = (EPI == EPI_DROP_RELU || EPI == EPI_TANH_DROP || EPI == EPI_TANH_BWD);
  constexpr bool USES_AUX = (EPI == EPI_RELU_BWD || EPI == EPI_TANH_BWD);
  const int l31 = lane & 31, lh = lane >> 5;
  const int N = p.N;
  const bool drop = USES_RND && p.thresh != 0u;
  unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
  if (drop) apply_seed_offset(key_lo, key_hi, p.seed_dev);
  // with a row map the dropout counter is the ORIGINAL row id (rows that share a counter block share a call)
  const bool mapped = drop && (p.rowmap != nullptr);
  const bool has_beta = p.beta != 0.f, has_rs = p.bias != nullptr && p.rowscale != nullptr;
  float bias_j[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = n0 + wn0 + 32 * j + l31;
    bias_j[j] = (p.bias != nullptr && col < N) ? p.bias[col] : 0.f;
  }
  // (r6) Small tiles (one or two 32 x 32 blocks per wave: the 64 x 64 and the 8-wave 128 x 128 configurations, i.e. the
  // hidden-layer gradient and the heads' data gradient of the step's latency-bound middle): the activation values the
  // backward epilogues multiply by (`aux`) are requested for a whole 32 x 32 BLOCK before its first group of four rows is
  // finished -- one exposed round trip per block instead of four (the compiler barrier below keeps the groups apart, and with
  // them their loads).  Sixteen registers: the 8-wave configuration is capped at 128 (a whole wave tile's 32 spilled).
  constexpr bool PRE_AUX = USES_AUX && (WM * WN <= 2);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + wn0 + 32 * j + l31;
      const bool colok = col < N;
      float axp[PRE_AUX ? 16 : 1];
      if constexpr (PRE_AUX) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int row = m0 + wm0 + 32 * i + 8 * q + 4 * lh + jj;
            const bool okk = colok && row < M;
            axp[4 * q + jj] = p.aux[(long)(okk ? row : m0) * p.ldaux + (okk ? col : n0)];
          }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row4 = m0 + wm0 + 32 * i + 8 * q + 4 * lh;   // multiple of 4
        if (row4 >= M) continue;
        bool ok[4];
        float* cptr[4];
        float old[4] = {0.f, 0.f, 0.f, 0.f}, ax[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          ok[jj] = colok && (row4 + jj < M);
          const int r = ok[jj] ? row4 + jj : m0, c = ok[jj] ? col : n0;       // (m0, n0) is always inside
          cptr[jj] = p.C + (long)r * p.ldc + c;
          if (has_beta) old[jj] = *cptr[jj];
          if constexpr (PRE_AUX) ax[jj] = axp[4 * q + jj];
          else if constexpr (USES_AUX) ax[jj] = p.aux[(long)r * p.ldaux + c];
          if (has_rs) rs[jj] = p.rowscale[r];
        }
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (drop) {
          if (!mapped) {
            philox4((unsigned)(p.drop_col_off + col), (unsigned)(row4 >> 2), p.site, 0u, key_lo, key_hi, w);
          } else {
            unsigned rid[4], rnd[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if (have_rid) rid[jj] = (unsigned)__shfl(rid_lane, 32 * i + 8 * q + 4 * lh + jj, 64);
              else rid[jj] = (unsigned)p.rowmap[row4 + jj < M ? row4 + jj : M - 1];
            }
            unsigned blk = rid[0] >> 2;
            philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if ((rid[jj] >> 2) != blk) {
                blk = rid[jj] >> 2;
                philox4((unsigned)(p.drop_col_off + col), blk, p.site, 0u, key_lo, key_hi, rnd);
              }
              const unsigned k = rid[jj] & 3u;
              w[jj] = k == 0u ? rnd[0] : (k == 1u ? rnd[1] : (k == 2u ? rnd[2] : rnd[3]));
            }
          }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          float v = acc[i][j][4 * q + jj] + bias_j[j] * rs[jj] + p.beta * old[jj];
          const bool keep = !drop || w[jj] >= p.thresh;
          if constexpr (EPI == EPI_DROP_RELU) {
            v = relu_f(v);
            v = keep ? v * p.drop_scale : 0.f;
          } else if constexpr (EPI == EPI_TANH_DROP) {
            const float t = tanhf(v);
            if (ok[jj]) p.aux_out[(long)(row4 + jj) * p.ldaux + col] = t;
            v = keep ? t * p.drop_scale : 0.f;
          } else if constexpr (EPI == EPI_RELU_BWD) {
            v = (ax[jj] > 0.f) ? v * p.drop_scale : 0.f;
          } else if constexpr (EPI == EPI_TANH_BWD) {
            const float f = 1.f - ax[jj] * ax[jj];
            v *= keep ? f * p.drop_scale : 0.f;
          }
          if (ok[jj]) *cptr[jj] = v;
        }
        // keep the groups apart: without this hipcc hoists the address arithmetic and loads of ALL groups of the
        // tile above the first store (hundreds of live registers -> spills in the 256x256 kernels)
        __asm__ volatile("" ::: "memory");
      }
    }
  }
}

// Epilogue kinds a layout can be launched with (lirec_hip.hip builds no other combination and launch_layout_
// rejects them): forward GEMMs (NT) store / dropout-relu / tanh-dropout, data gradients (NN) store / relu' / tanh',
// weight gradients (TN) store only.  Keeps each kernel's code small.
__host__ __device__ constexpr bool epi_allowed(int layout, int epi) {
  return epi == EPI_STORE || (layout == L_NT && (epi == EPI_DROP_RELU || epi == EPI_TANH_DROP)) ||
         (layout == L_NN && (epi == EPI_RELU_BWD || epi == EPI_TANH_BWD));
}

template <int WM, int WN, int LAYOUT>
__device__ __forceinline__ void gemm_epilogue(const GemmProblem& p, const f32x16 (&acc)[WM][WN], int m0, int n0,
                                              int wm0, int wn0, int lane, int split, int M_eff, bool have_rid = false, int rid_lane = 0) {
  const int l31 = lane & 31, lh = lane >> 5;
  const int M = M_eff, N = p.N;
  if (p.ksplit > 1) {
    float* slab = p.slab + (long)split * p.M * N;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn0 + 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M && col < N) slab[(long)row * N + col] = acc[i][j][r];
        }
      }
    return;
  }
  // The epilogue kind is a property of the launch: dispatch once, so that the per-element code of a tile is
  // straight-line for its kind (the first version branched on p.epi / p.bias / p.beta for every element
  // and loaded bias[col] per element behind the previous element's store: 20 % of a K1 tile's time).
  if constexpr (LAYOUT == L_NT) {
    if (p.epi == EPI_DROP_RELU) gemm_epilogue_kind<WM, WN, EPI_DROP_RELU>(p, acc, m0, n0, wm0, wn0, lane, M, have_rid, rid_lane);
    else if (p.epi == EPI_TANH_DROP) gemm_epilogue_kind<WM, WN, EPI_TANH_DROP>(p, acc, m0, n0, wm0, wn0, lane, M);
    else gemm_epilogue_kind<WM, WN, EPI_STORE>(p, acc, m0, n0, wm0, wn0, lane, M);
  } else if constexpr (LAYOUT == L_NN) {
    if (p.epi == EPI_RELU_BWD) gemm_epilogue_kind<WM, WN, EPI_RELU_BWD>(p, acc, m0, n0, wm0, wn0, lane, M);
    else if (p.epi == EPI_TANH_BWD) gemm_epilogue_kind<WM, WN, EPI_TANH_BWD>(p, acc, m0, n0, wm0, wn0, lane, M);
    else gemm_epilogue_kind<WM, WN, EPI_STORE>(p, acc, m0, n0, wm0, wn0, lane, M);
  } else {
    gemm_epilogue_kind<WM, WN, EPI_STORE>(p, acc, m0, n0, wm0, wn0, lane, M);
  }
}

// C (op)= bias + beta*C + sum_s slab[s]  and  dbias += sum_s dbias_slab[s]  -- fixed order, deterministic
// (a plain kernel in a header shared by several translation units: internal linkage)
// (A version that dealt the float4 items of all problems to the grid as one index space -- no serial walk over the
//  problems -- measured 10-30 % SLOWER: the per-item problem lookup made the slab loop's trip count a per-lane value.)
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmGroup g) {
  for (int pi = 0; pi < g.nprob; ++pi) {
    const GemmProblem& p = g.p[pi];
    if (p.ksplit <= 1) continue;
    const int ks = (p.dyn && g.dyn_is_k) ? effective_ksplit(g, p, dyn_limit(p, p.K)) : p.ksplit;
    const long mn = (long)p.M * p.N;
    const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    const bool vec4 = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((((size_t)p.C) | ((size_t)p.slab)) % 16 == 0);
    if (p.epi != EPI_STORE) {
      // A split launch WITH an epilogue (the under-filled 3072^2 gate GEMMs on 1024 rows: two k-chunks fill the chip):
      // the sum of the slabs takes the place of the accumulator and the epilogue of gemm_epilogue_kind runs here.
      // One thread = 4 consecutive rows of one column, the unit one Philox call serves (consecutive lanes =
      // consecutive columns: every access is a coalesced row segment).  Never used with a row map.
      const bool drop = p.thresh != 0u && (p.epi == EPI_DROP_RELU || p.epi == EPI_TANH_DROP || p.epi == EPI_TANH_BWD);
      unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
      if (drop) apply_seed_offset(key_lo, key_hi, p.seed_dev);
      const long groups = (long)((p.M + 3) >> 2) * p.N;
      for (long e = gtid; e < groups; e += gsz) {
        const int r4 = (int)(e / p.N), col = (int)(e - (long)r4 * p.N), row0 = r4 << 2;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sidx = 0; sidx < ks; ++sidx) {
          const float* sl = p.slab + (long)sidx * mn + (long)row0 * p.N + col;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) if (row0 + jj < p.M) v[jj] += sl[(long)jj * p.N];
        }
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (drop) philox4((unsigned)(p.drop_col_off + col), (unsigned)r4, p.site, 0u, key_lo, key_hi, w);
        const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int row = row0 + jj;
          if (row >= p.M) continue;
          float* c = p.C + (long)row * p.ldc + col;
          float x = v[jj] + b * ((p.bias && p.rowscale) ? p.rowscale[row] : 1.f) + (p.beta != 0.f ? p.beta * (*c) : 0.f);
          const bool keep = !drop || w[jj] >= p.thresh;
          if (p.epi == EPI_DROP_RELU) {
            x = relu_f(x);
            x = keep ? x * p.drop_scale : 0.f;
          } else if (p.epi == EPI_TANH_DROP) {
            const float t = tanhf(x);
            p.aux_out[(long)row * p.ldaux + col] = t;
            x = keep ? t * p.drop_scale : 0.f;
          } else if (p.epi == EPI_RELU_BWD) {
            x = (p.aux[(long)row * p.ldaux + col] > 0.f) ? x * p.drop_scale : 0.f;
          } else if (p.epi == EPI_TANH_BWD) {
            const float t = p.aux[(long)row * p.ldaux + col];
            const float f = 1.f - t * t;
            x *= keep ? f * p.drop_scale : 0.f;
          }
          *c = x;
        }
      }
    } else if (vec4) {
      const long mn4 = mn >> 2;
      const int n4 = p.N >> 2;
      for (long e = gtid; e < mn4; e += gsz) {
        const int row = (int)(e / n4), col = (int)(e - (long)row * n4) * 4;
        const f32x4* src = (const f32x4*)p.slab + e;
        f32x4* c = (f32x4*)(p.C + (long)row * p.ldc + col);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (p.beta != 0.f) o = *c;
        f32x4 v = src[0];
        // fixed summation order s = 0, 1, 2, ... whatever the load order; four slabs' loads in flight per lane (a
        // one-load-per-trip loop waits for each load in turn and ran at 3.9 TB/s)
        int s = 1;
        for (; s + 3 < ks; s += 4) {
          const f32x4 t0 = src[(long)s * mn4], t1 = src[(long)(s + 1) * mn4], t2 = src[(long)(s + 2) * mn4], t3 = src[(long)(s + 3) * mn4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += t0[j]; v[j] += t1[j]; v[j] += t2[j]; v[j] += t3[j]; }
        }
        for (; s < ks; ++s) { const f32x4 t = src[(long)s * mn4]; v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
        if (p.bias) {
          const float rs = p.rowscale ? p.rowscale[row] : 1.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += p.bias[col + j] * rs;
        }
        if (p.beta != 0.f) { v[0] += p.beta * o[0]; v[1] += p.beta * o[1]; v[2] += p.beta * o[2]; v[3] += p.beta * o[3]; }
        *c = v;
      }
    } else {
      for (long e = gtid; e < mn; e += gsz) {
        const int row = (int)(e / p.N), col = (int)(e - (long)row * p.N);
        float v = 0.f;
        for (int s = 0; s < ks; ++s) v += p.slab[(long)s * mn + e];
        if (p.bias) v += p.rowscale ? p.bias[col] * p.rowscale[row] : p.bias[col];
        float* c = p.C + (long)row * p.ldc + col;
        if (p.beta != 0.f) v += p.beta * (*c);
        *c = v;
      }
    }
    if (p.dbias && p.dbias_slab) {
      for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < ks; ++s) v += p.dbias_slab[(long)s * p.M + m];
        p.dbias[m] = p.dbias_set ? v : p.dbias[m] + v;
      }
    }
  }
}

// The same reduction with one problem per WORKGROUP instead of a serial walk over the problems inside every thread: the
// grid is the concatenation of the problems' item ranges (ReducePlan, built on the host), the problem lookup is uniform
// per workgroup (the earlier flat version looked the problem up per item: a per-lane trip count for the slab loop, 10-30 %
// slower), every thread owns one float4 of one problem: ks + 1 independent loads, one store.  Plain-store problems only.
struct ReducePlan { int n; int start[LIREC_MAX_PROB + 1]; int prob[LIREC_MAX_PROB]; };
static __global__ __launch_bounds__(256) void splitk_reduce_flat_kernel(const GemmGroup g, const ReducePlan rp) {
  int k = 0;
  while (k + 1 < rp.n && (int)blockIdx.x >= rp.start[k + 1]) ++k;
  const GemmProblem& p = g.p[rp.prob[k]];
  const int ks = p.ksplit;
  const long mn = (long)p.M * p.N;
  const long e = (long)((int)blockIdx.x - rp.start[k]) * 256 + threadIdx.x;
  const bool vec4 = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((((size_t)p.C) | ((size_t)p.slab)) % 16 == 0);
  if (vec4) {
    const long mn4 = mn >> 2;
    const int n4 = p.N >> 2;
    if (e < mn4) {
      const int row = (int)(e / n4), col = (int)(e - (long)row * n4) * 4;
      const f32x4* src = (const f32x4*)p.slab + e;
      f32x4* c = (f32x4*)(p.C + (long)row * p.ldc + col);
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (p.beta != 0.f) o = *c;
      f32x4 v = src[0];
      int s = 1;
      for (; s + 3 < ks; s += 4) {
        const f32x4 t0 = src[(long)s * mn4], t1 = src[(long)(s + 1) * mn4], t2 = src[(long)(s + 2) * mn4], t3 = src[(long)(s + 3) * mn4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += t0[j]; v[j] += t1[j]; v[j] += t2[j]; v[j] += t3[j]; }
      }
      for (; s < ks; ++s) { const f32x4 t = src[(long)s * mn4]; v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      if (p.bias) {
        const float rs = p.rowscale ? p.rowscale[row] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += p.bias[col + j] * rs;
      }
      if (p.beta != 0.f) { v[0] += p.beta * o[0]; v[1] += p.beta * o[1]; v[2] += p.beta * o[2]; v[3] += p.beta * o[3]; }
      *c = v;
    }
  } else if (e < mn) {
    const int row = (int)(e / p.N), col = (int)(e - (long)row * p.N);
    // (four slabs in flight, added in slab order: the heads' logits -- N = 101 / 15, no float4 rows -- took one exposed round trip per
    //  slab here)
    float v = 0.f;
    int s = 0;
    for (; s + 3 < ks; s += 4) {
      const float t0 = p.slab[(long)s * mn + e], t1 = p.slab[(long)(s + 1) * mn + e], t2 = p.slab[(long)(s + 2) * mn + e], t3 = p.slab[(long)(s + 3) * mn + e];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; s < ks; ++s) v += p.slab[(long)s * mn + e];
    if (p.bias) v += p.rowscale ? p.bias[col] * p.rowscale[row] : p.bias[col];
    float* c = p.C + (long)row * p.ldc + col;
    if (p.beta != 0.f) v += p.beta * (*c);
    *c = v;
  }
  if (p.dbias && p.dbias_slab && e < p.M) {
    float v = 0.f;
    int s = 0;
    for (; s + 3 < ks; s += 4) {
      const float t0 = p.dbias_slab[(long)s * p.M + e], t1 = p.dbias_slab[(long)(s + 1) * p.M + e], t2 = p.dbias_slab[(long)(s + 2) * p.M + e],
                  t3 = p.dbias_slab[(long)(s + 3) * p.M + e];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; s < ks; ++s) v += p.dbias_slab[(long)s * p.M + e];
    p.dbias[e] = p.dbias_set ? v : p.dbias[e] + v;
  }
}

// TAG has no functional role: it gives the two heavy call sites (1 = embed layer-1
// forward, 2 = embed layer-1 weight gradient) their own kernel symbols so that a
// rocprofv3 kernel trace reports them separately from the small GEMMs.
template <int LAYOUT, int WM, int WN, int TAG, bool VEC>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmGroup g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
  constexpr bool A_KC = (LAYOUT != L_TN);    // A k-contiguous in memory
  constexpr bool B_KC = (LAYOUT == L_NT);    // B k-contiguous in memory
  constexpr int PA = A_KC ? BM + 1 : BM + 4;
  constexpr int PB = B_KC ? BN + 1 : BN + 4;
  constexpr int A_TILE = BK * PA, B_TILE = BK * PB;
  constexpr int CA = BM / 32, CB = BN / 32;  // float4 chunks per thread per k-tile
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
  float* const As = smem;
  float* const Bs = smem + 2 * A_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ltile = launch_tile(g);
  if (ltile < 0) return;
  const TileCoord tc = decode_tile<BM, BN>(g, ltile, LAYOUT == L_TN);
  const GemmProblem& p = g.p[tc.pi];
  const int m0 = tc.m0, n0 = tc.n0, tn = tc.tn;
  const int M = tc.M, N = p.N, K = tc.k_end;     // this workgroup reduces k in [k_begin, K)
  if (m0 >= M) return;                           // row-compacted launch: nothing beyond the valid rows

  // ---- per-thread staging assignment -------------------------------------
  // k-contiguous operand: rows (tid>>3) + 32*i, k-quad tid&7
  // m/n-contiguous operand: k-rows (tid / (cols/4)) + (1024/cols)*i, column quad tid % (cols/4)
  const float* a_rowptr[CA];
  const float* b_rowptr[CB];
  bool a_rowok[CA], b_rowok[CB];
  if constexpr (A_KC) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int m = m0 + (tid >> 3) + 32 * i;
      a_rowok[i] = m < M;
      // NT: A is the X operand (row selection); NN: dense
      a_rowptr[i] = p.A + (LAYOUT == L_NT ? phys_row(p, a_rowok[i] ? m : 0) : (long)(a_rowok[i] ? m : 0)) * p.lda;
    }
  }
  if constexpr (B_KC) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int n = n0 + (tid >> 3) + 32 * i;
      b_rowok[i] = n < N;
      b_rowptr[i] = p.B + (long)(b_rowok[i] ? n : 0) * p.ldb;
    }
  }

  f32x4 ra[CA], rb[CB];

  auto load_tiles = [&](int k0) {
    if constexpr (A_KC) {
      const int k = k0 + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < CA; ++i)
        ra[i] = raw4<VEC>(a_rowptr[i] + k, a_rowok[i] ? K - k : 0, p.A);
    } else {
      constexpr int QPR = BM / 4, KSTEP = 256 / QPR;     // quads per k-row, k-rows per pass
      const int mq = m0 + 4 * (tid % QPR);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int k = k0 + tid / QPR + KSTEP * i;
        const bool ok = k < K;
        ra[i] = raw4<VEC>(p.A + (long)k * p.lda + mq, ok ? M - mq : 0, p.A);
      }
    }
    if constexpr (B_KC) {
      const int k = k0 + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < CB; ++i)
        rb[i] = raw4<VEC>(b_rowptr[i] + k, b_rowok[i] ? K - k : 0, p.B);
    } else {
      constexpr int QPR = BN / 4, KSTEP = 256 / QPR;
      const int nq = n0 + 4 * (tid % QPR);
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int k = k0 + tid / QPR + KSTEP * i;
        const bool ok = k < K;
        // TN: B is the X operand, its rows are the reduction index
        const long row = (LAYOUT == L_TN) ? phys_row(p, ok ? k : 0) : (long)k;
        rb[i] = raw4<VEC>(p.B + row * p.ldb + nq, ok ? N - nq : 0, p.B);
      }
    }
  };

  auto store_tiles = [&](int buf, int k0) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
    if constexpr (A_KC) {
      const int kq = 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const f32x4 v = mask4(ra[i], a_rowok[i] ? K - (k0 + kq) : 0);
        const int r = (tid >> 3) + 32 * i;
        as[(kq + 0) * PA + r] = v.x; as[(kq + 1) * PA + r] = v.y;
        as[(kq + 2) * PA + r] = v.z; as[(kq + 3) * PA + r] = v.w;
      }
    } else {
      constexpr int QPR = BM / 4, KSTEP = 256 / QPR;
      const int mq = m0 + 4 * (tid % QPR);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int kr = tid / QPR + KSTEP * i;
        *reinterpret_cast<f32x4*>(as + kr * PA + 4 * (tid % QPR)) = mask4(ra[i], (k0 + kr < K) ? M - mq : 0);
      }
    }
    if constexpr (B_KC) {
      const int kq = 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const f32x4 v = mask4(rb[i], b_rowok[i] ? K - (k0 + kq) : 0);
        const int r = (tid >> 3) + 32 * i;
        bs[(kq + 0) * PB + r] = v.x; bs[(kq + 1) * PB + r] = v.y;
        bs[(kq + 2) * PB + r] = v.z; bs[(kq + 3) * PB + r] = v.w;
      }
    } else {
      constexpr int QPR = BN / 4, KSTEP = 256 / QPR;
      const int nq = n0 + 4 * (tid % QPR);
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int kr = tid / QPR + KSTEP * i;
        *reinterpret_cast<f32x4*>(bs + kr * PB + 4 * (tid % QPR)) = mask4(rb[i], (k0 + kr < K) ? N - nq : 0);
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const bool do_dbias = (LAYOUT == L_TN) && p.dbias != nullptr && tn == 0 && tid < BM;
  float dbias_acc = 0.f;

  const int kb = tc.k_begin;
  const int nk = (K - kb + BK - 1) / BK;
  if (nk > 0) {
    load_tiles(kb);
    store_tiles(0, kb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kb + (kt + 1) * BK);       // in flight during the MFMA block
    const float* as = As + buf * A_TILE + wm0 + l31;
    const float* bs = Bs + buf * B_TILE + wn0 + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = as[(2 * kk + lh) * PA + 32 * i];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = bs[(2 * kk + lh) * PB + 32 * j];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_dbias) {
      const float* ac = As + buf * A_TILE + tid;
      if (p.rowscale) {
        const int kbase = kb + kt * BK;
        for (int k = 0; k < BK; ++k) dbias_acc += ac[k * PA] * ((kbase + k < K) ? p.rowscale[kbase + k] : 0.f);
      } else {
#pragma unroll 8
        for (int k = 0; k < BK; ++k) dbias_acc += ac[k * PA];
      }
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1, kb + (kt + 1) * BK);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------
  gemm_epilogue<WM, WN, LAYOUT>(p, acc, m0, n0, wm0, wn0, lane, tc.split, M);
  if (do_dbias && m0 + tid < M) {
    if (p.ksplit > 1) p.dbias_slab[(long)tc.split * p.M + m0 + tid] = dbias_acc;
    else p.dbias[m0 + tid] = p.dbias_set ? dbias_acc : p.dbias[m0 + tid] + dbias_acc;
  }
}

// One thread per output element: bring-up cross-check of the MFMA kernels (same
// GemmProblem semantics, still a HIP kernel -- not a CPU fallback).
template <int LAYOUT>
__global__ void gemm_naive_kernel(const GemmProblem p) {
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int row = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int M = (LAYOUT == L_TN) ? p.M : dyn_limit(p, p.M);
  const int K = (LAYOUT == L_TN) ? dyn_limit(p, p.K) : p.K;
  if (row >= M || col >= p.N) return;
  float acc = 0.f;
  if (LAYOUT == L_NT) {
    const long arow = phys_row(p, row) * p.lda;
    const float* a = p.A + arow;
    const float* b = p.B + (long)col * p.ldb;
    if (p.x_bf16) for (int k = 0; k < p.K; ++k) acc = fmaf(bf16_at(p.A, arow + k), b[k], acc);
    else for (int k = 0; k < p.K; ++k) acc = fmaf(a[k], b[k], acc);
  } else if (LAYOUT == L_NN) {
    const float* a = p.A + (long)row * p.lda;
    for (int k = 0; k < p.K; ++k) acc = fmaf(a[k], p.B[(long)k * p.ldb + col], acc);
  } else {
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
      const float a = p.A[(long)k * p.lda + row];
      const long xo = phys_row(p, k) * p.ldb + col;
      acc = fmaf(a, p.x_bf16 ? bf16_at(p.B, xo) : p.B[xo], acc);
      s += p.rowscale ? a * p.rowscale[k] : a;
    }
    if (p.dbias && col == 0) p.dbias[row] = p.dbias_set ? s : p.dbias[row] + s;
  }
  unsigned rnd[4] = {0u, 0u, 0u, 0u};
  const int rid = (p.rowmap && LAYOUT != L_TN) ? p.rowmap[row] : row;     // dropout counters use original row ids
  if (epi_uses_dropout(p)) {
    unsigned key_lo = p.seed_lo, key_hi = p.seed_hi;
    apply_seed_offset(key_lo, key_hi, p.seed_dev);
    philox4((unsigned)(p.drop_col_off + col), (unsigned)(rid >> 2), p.site, 0u, key_lo, key_hi, rnd);
  }
  epi_store(p, row, col, acc, rnd[rid & 3]);
}

}  // namespace lirec
