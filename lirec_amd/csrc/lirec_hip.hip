// C ABI of the MI355X hot-path library (see include/lirec_hip.h for the contract
// and the reference lines each entry point replaces).  Host side: argument
// checks, GEMM problem descriptors, kernel launches on the caller's stream.
// No allocation, no synchronisation, no exceptions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "lirec_hip.h"
#include "gemm.hpp"
#include "gemm_launch.hpp"
#include "kernels.hpp"

using namespace lirec;

thread_local lirec::CmdList* lirec::t_rec = nullptr;       // record.hpp
bool lirec::g_dry = false;                                 // record.hpp: host-side dry run (sanitizer tests, no GPU)
struct lirec_cmdlist { lirec::CmdList list; };

// Library state that used to be process-global (review, round 1): the GEMM core, the split-K scratch and the diagnostic
// switches now live in a CONTEXT.  Every thread has a current context (the default one until lirec_ctx_set_current is
// called); a host that drives two streams -- training and evaluation side by side -- gives each its own context, and
// with it its own scratch, instead of sharing one buffer across streams.
struct lirec_ctx {
  int gemm_mode = 0;
  int ablate = 0;              // diagnostics only (lirec_debug_set)
  int force_cfg = -1;
  // caller-registered scratch for split-K partial tiles (lirec_set_scratch): used by whichever GEMM launch of THIS
  // context runs next, so the launches of one context belong on one stream
  float* scratch = nullptr;
  long scratch_floats = 0;
};
static thread_local bool t_no_split = false;   // a launch whose results must not depend on how K is cut (indexed layer 1)
static lirec_ctx g_default_ctx;
static thread_local lirec_ctx* t_ctx = &g_default_ctx;
#define g_gemm_mode (t_ctx->gemm_mode)
// modes 2 and 3 run the bf16 MFMA core: 2 = three passes (hi/lo split, fp32-grade), 3 = ONE pass on the large GEMMs (layer 1 and its
// weight gradient, the gate's three) with operands rounded to bf16 once -- BASELINE config 5's arithmetic, never the headline's;
// its operands live in q16c (bf16 values, 64-column blocks: gemm_bf16x3.hpp) -- rows, first-layer weights and the gate's staged
// operands alike -- and each mode refuses the other's bf16 layout
#define g_bf_core (t_ctx->gemm_mode == 2 || t_ctx->gemm_mode == 3)
#define g_ablate (t_ctx->ablate)
#define g_force_cfg (t_ctx->force_cfg)
#define g_scratch (t_ctx->scratch)
#define g_scratch_floats (t_ctx->scratch_floats)

// ---------------------------------------------------------------------------
// optional per-call-site timing with HIP events on the launch stream (off by default;
// bench.py turns it on for a separate, un-timed pass to price each kernel against its
// roofline).  Not thread-safe; single-threaded caller.
// ---------------------------------------------------------------------------
enum { PS_EMBED_L1_FWD = 0, PS_EMBED_L2_FWD, PS_EMBED_DW2, PS_EMBED_DZ1, PS_EMBED_DW1, PS_GATE_FWD, PS_GATE_DW,
       PS_GATE_DEE, PS_LINEAR_FWD, PS_LINEAR_DW, PS_LINEAR_DA, PS_POOL_FWD, PS_POOL_BWD, PS_LOSS, PS_ADAM, PS_CAST,
       PS_STAGE, PS_EMBED_DW1_RED, PS_GATE_STAGE, PS_COUNT };
static const char* const g_site_names[PS_COUNT] = {
    "embed_l1_fwd", "embed_l2_fwd", "embed_dW2", "embed_dZ1", "embed_dW1", "gate_fwd", "gate_dW", "gate_dEE",
    "linear_fwd", "linear_dW", "linear_dA", "pool_fwd", "pool_bwd", "loss", "adam", "cast", "stage", "embed_dW1_reduce", "gate_stage"};
#define PROF_CAP 1024
struct ProfRec { hipEvent_t a, b; int site; };
static int g_prof_on = 0, g_nrec = 0, g_nev = 0;
static ProfRec g_recs[PROF_CAP];
static double g_ms[PS_COUNT], g_flops[PS_COUNT], g_bytes[PS_COUNT];
static long g_cnt[PS_COUNT];

static void prof_flush() {
  for (int i = 0; i < g_nrec; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_recs[i].b) == hipSuccess && hipEventElapsedTime(&ms, g_recs[i].a, g_recs[i].b) == hipSuccess)
      g_ms[g_recs[i].site] += ms;
  }
  g_nrec = 0;
}
static inline int prof_start_now(int site, hipStream_t s) {
  if (!g_prof_on || lirec::g_dry) return -1;
  if (g_nrec == PROF_CAP) prof_flush();
  const int i = g_nrec++;
  if (i >= g_nev) { (void)hipEventCreate(&g_recs[i].a); (void)hipEventCreate(&g_recs[i].b); g_nev = i + 1; }
  g_recs[i].site = site;
  (void)hipEventRecord(g_recs[i].a, s);
  return i;
}
static inline void prof_stop_now(int i, hipStream_t s, double flops, double bytes) {
  if (i < 0) return;
  (void)hipEventRecord(g_recs[i].b, s);
  const int site = g_recs[i].site;
  g_cnt[site] += 1; g_flops[site] += flops; g_bytes[site] += bytes;
}
// While a step is being RECORDED (record.hpp) the site brackets go into the command list too: a replay with profiling on then
// prices every site of the recorded step -- the step bench.py times -- exactly as the eager loop's brackets price the eager one
// (one branch per bracket when profiling is off).  Brackets do not nest; the open one of a replay is g_prof_open.
static int g_prof_open = -1;
static inline int prof_start(int site, hipStream_t s) {
  if (lirec::t_rec) lirec::t_rec->push([=]() { g_prof_open = prof_start_now(site, s); }, s, 2);
  return prof_start_now(site, s);
}
static inline void prof_stop(int i, hipStream_t s, double flops, double bytes) {
  if (lirec::t_rec) lirec::t_rec->push([=]() { prof_stop_now(g_prof_open, s, flops, bytes); g_prof_open = -1; }, s, 2);
  prof_stop_now(i, s, flops, bytes);
}

#define LIREC_CHECK_LAUNCH()                      \
  do {                                            \
    hipError_t e__ = lirec::g_dry ? hipSuccess : hipGetLastError(); \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

static inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

static inline unsigned drop_thresh(float p) {
  if (!(p > 0.f)) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (unsigned)t;                     // floor, as oracle: int(p * 2**32)
}

static inline void set_dropout(GemmProblem& q, const lirec_dropout* d, int site, int col_off) {
  const float p = d ? d->p : 0.f;
  q.seed_lo = d ? (unsigned)(d->seed & 0xffffffffull) : 0u;
  q.seed_hi = d ? (unsigned)(d->seed >> 32) : 0u;
  q.seed_dev = d ? (const unsigned long long*)d->seed_dev : nullptr;
  q.site = (unsigned)site;
  q.thresh = drop_thresh(p);
  q.drop_scale = (p > 0.f) ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
  q.drop_col_off = col_off;
}

// exact n / gs by one multiply-high whenever max_rows * gs < 2^32 (see GemmProblem::gs_magic)
static inline unsigned row_magic(int gs, long max_rows) {
  if (gs < 2 || (unsigned long long)max_rows * (unsigned long long)gs >= (1ull << 32)) return 0u;
  return (unsigned)((1ull << 32) / (unsigned)gs) + 1u;
}

// Weight gradients of this process OVERWRITE their buffers instead of accumulating into them (lirec_set_grad_overwrite): a step
// that is issued as a unit -- the recorded command list -- then needs no zeroing pass over the gradient buffer.  One flag for all
// contexts: the weight-gradient launches of one backward run on two of them.
static thread_local int g_grad_overwrite = 0;       // (per host thread, like the recorder and the contexts: a backward on another
                                                    //  thread -- an evaluation loop's, another model's -- never picks it up)
static inline float grad_beta() { return g_grad_overwrite ? 0.f : 1.f; }
// While the mode is on, every weight / bias gradient target that is handed to a launch is noted: a parameter written by TWO
// launches of one step (a tied module, a gradient cut over several launches) would silently lose the first contribution -- the
// caller that switches the mode on asks for the count of such targets afterwards (lirec_grad_overwrite_conflicts) and keeps
// the zeroing pass if there is any.
#include <vector>
static thread_local std::vector<const void*> t_ow_targets;
static thread_local int t_ow_conflicts = 0;
static inline void ow_note(const void* p) {
  if (!g_grad_overwrite || !p) return;
  for (const void* q : t_ow_targets) if (q == p) { ++t_ow_conflicts; return; }
  t_ow_targets.push_back(p);
}
static inline void ow_note_group(const GemmGroup& g) {
  if (!g_grad_overwrite) return;
  for (int i = 0; i < g.nprob; ++i)
    if (g.p[i].M > 0 && g.p[i].N > 0 && g.p[i].beta == 0.f) { ow_note(g.p[i].C); if (g.p[i].dbias_set) ow_note(g.p[i].dbias); }
}

static inline GemmProblem make_problem() {
  GemmProblem q;
  memset(&q, 0, sizeof(q));
  q.drop_scale = 1.f;
  return q;
}

// Per-problem tile counts, split-K chunks and scratch slabs for a bm x bn tiling, then the tile order of the group
// (row-panel-major across problems, in two tiers when the problems have two heights: see GemmGroup).  `ks_want[i]` is
// the wanted split of problem i (clamped to >= 8 k-tiles per chunk, <= 32, and to what fits the registered scratch).
// Returns the grid size.
static int plan_tiles_v(GemmGroup& g, int bm, int bn, const int* ks_want, bool& any_split) {
  int start = 0;
  long scratch_off = 0;
  any_split = false;
  for (int i = 0; i < g.nprob; ++i) {
    GemmProblem& p = g.p[i];
    const int tm = (p.M + bm - 1) / bm, tn = (p.N + bn - 1) / bn;
    p.tiles_n = tn > 0 ? tn : 1;
    p.tiles_mn = tm * tn;
    p.tile_start = start;
    int ks = ks_want[i];
    const int max_ks = p.K / 256 > 0 ? p.K / 256 : 1;            // >= 8 k-tiles per chunk
    if (ks > max_ks) ks = max_ks;
    if (ks > 32) ks = 32;
    if (ks < 1) ks = 1;
    int kchunk = ((p.K + ks - 1) / ks + 31) / 32 * 32;
    ks = (p.K + kchunk - 1) / kchunk;
    const long need = (long)ks * p.M * p.N + (p.dbias ? (long)ks * p.M : 0);
    if (ks > 1 && scratch_off + need <= g_scratch_floats) {
      p.ksplit = ks; p.kchunk = kchunk;
      p.slab = g_scratch + scratch_off;
      p.dbias_slab = p.dbias ? p.slab + (long)ks * p.M * p.N : nullptr;
      scratch_off += need;
      any_split = true;
    } else {
      p.ksplit = 1; p.kchunk = (p.K + 31) / 32 * 32; p.slab = nullptr; p.dbias_slab = nullptr;
    }
    start += (p.M > 0 && p.N > 0) ? tm * tn * p.ksplit : 0;
  }
  g.total_tiles = start;
  g.row_tiles = 0; g.tier_rows = 0; g.row_tiles2 = 0; g.first2 = 0; g.pgroup = 0; g.tm_fast = 0;
  {
    // run-time split of row-compacted weight gradients (effective_ksplit): the numbers the device-side rule needs
    long tiles = 0, mn_total = 0;
    for (int i = 0; i < g.nprob; ++i) { tiles += g.p[i].tiles_mn; mn_total += (long)g.p[i].M * g.p[i].N; }
    const double tk = (bm * bn >= 256 * 256) ? 2.8 : (bm * bn >= 256 * 128 ? 1.6 : 1.2);      // us per k-tile (measured)
    g.tiles_per_split = (int)tiles;
    g.resident_slots = (bm * bn > 128 * 128) ? 256 : 512;
    g.slab_cost = (float)((double)mn_total * 4.0 / 4.0e6 / tk);       // one slab written + read at ~4 TB/s, in k-tiles
    g.reduce_cost = (float)(6.0 / tk);                                // the reduce launch
  }
  if (start == 0) return 0;
  if (g.nprob > 1) {
    // Row-panel-major tile order across the problems of a group (see GemmGroup): same tiles_m and ksplit
    // everywhere -> (split, tm, problem, tn); unsplit problems of exactly two heights, the short ones first ->
    // two tiers.
    auto tiles_m = [&](int i) { return g.p[i].tiles_mn / g.p[i].tiles_n; };
    bool same = true, unsplit = g.p[0].ksplit == 1;
    for (int i = 1; i < g.nprob; ++i) {
      same = same && tiles_m(i) == tiles_m(0) && g.p[i].ksplit == g.p[0].ksplit;
      unsplit = unsplit && g.p[i].ksplit == 1;
    }
    int first2 = 0;
    bool two = !same && unsplit;
    if (two) {
      while (first2 < g.nprob && tiles_m(first2) == tiles_m(0)) ++first2;
      two = first2 < g.nprob && tiles_m(first2) > tiles_m(0);
      for (int i = first2; two && i < g.nprob; ++i) two = tiles_m(i) == tiles_m(first2);
    }
    if (same || two) {
      int off = 0;
      for (int i = 0; i < g.nprob; ++i) { g.p[i].tile_start = off; off += g.p[i].tiles_n; }
      g.row_tiles = off;
      g.tier_rows = tiles_m(0);
      if (two) { g.first2 = first2; g.row_tiles2 = off - g.p[first2].tile_start; }
      // panel-group order (GemmGroup::pgroup): unsplit launches whose problems all have the same number of column tiles
      bool eq = unsplit && !(g_ablate & 256);
      for (int i = 1; i < g.nprob; ++i) eq = eq && g.p[i].tiles_n == g.p[0].tiles_n;
      const int tn = g.p[0].tiles_n;
      if (eq && tn <= 16 && 32 % tn == 0 && g.tier_rows >= 32 / tn) g.pgroup = 32 / tn;
    }
  }
  // run-time split: one split factor and one device-side row count for the whole launch, split-major tile order
  {
    bool ok = g.p[0].ksplit > 1 && g.p[0].dyn != nullptr && (g.nprob == 1 || (g.row_tiles > 0 && g.row_tiles2 == 0));
    for (int i = 1; ok && i < g.nprob; ++i) ok = g.p[i].ksplit == g.p[0].ksplit && g.p[i].dyn == g.p[0].dyn && g.p[i].K == g.p[0].K;
    g.dyn_split = ok ? 1 : 0;
  }
  // weight-gradient launches of a group: the tiles that share an X block are adjacent (GemmGroup::tm_fast)
  g.tm_fast = (g.dyn_is_k && g.row_tiles > 0 && g.row_tiles2 == 0 && !(g_ablate & 512)) ? 1 : 0;
  return start;
}
// the split-K reduce of a launch: one workgroup range per split problem (plain-store problems; diagnostics keep the old kernel)
static void launch_splitk_reduce(const GemmGroup& g, hipStream_t s) {
  ReducePlan rp;
  memset(&rp, 0, sizeof(rp));
  bool plain = !(g_ablate & (64 | 128 | 4096));
  int blocks = 0;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProblem& p = g.p[i];
    if (p.ksplit <= 1) continue;
    plain = plain && p.epi == EPI_STORE && !(p.dyn && g.dyn_split && (g_ablate & 64));
    const bool vec4 = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((((size_t)p.C) | ((size_t)p.slab)) % 16 == 0);
    long items = vec4 ? ((long)p.M * p.N) >> 2 : (long)p.M * p.N;
    if (p.dbias && p.dbias_slab && items < p.M) items = p.M;
    rp.prob[rp.n] = i;
    rp.start[rp.n++] = blocks;
    blocks += (int)((items + 255) / 256);
  }
  rp.start[rp.n] = blocks;
  if (plain && rp.n > 0 && blocks > 0) lirec::launch(splitk_reduce_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, rp);
  else lirec::launch(splitk_reduce_kernel, dim3(1024), dim3(256), 0, s, g);
}

static int plan_tiles(GemmGroup& g, int bm, int bn, int ksplit_want, bool& any_split) {
  int ks[LIREC_MAX_PROB];
  for (int i = 0; i < LIREC_MAX_PROB; ++i) ks[i] = ksplit_want;
  return plan_tiles_v(g, bm, bn, ks, any_split);
}

template <int LAYOUT>
static int launch_layout_(GemmGroup& g, GemmMeta meta, hipStream_t s);

template <int LAYOUT>
static int launch_layout(GemmGroup& g, GemmMeta meta, hipStream_t s) {
  if (g.nprob <= 0) return LIREC_OK;
  double flops = 0.0;
  for (int i = 0; i < g.nprob; ++i) flops += 2.0 * g.p[i].M * (double)g.p[i].N * g.p[i].K;
  const int pi = prof_start(meta.site, s);
  const int rc = launch_layout_<LAYOUT>(g, meta, s);
  prof_stop(pi, s, flops, 0.0);
  return rc;
}

template <int LAYOUT>
static int launch_layout_(GemmGroup& g, GemmMeta meta, hipStream_t s) {
  for (int i = 0; i < g.nprob; ++i)
    if (!epi_allowed(LAYOUT, g.p[i].epi)) return LIREC_EINVAL;   // kernels only carry the epilogues of their layout
  if (g_gemm_mode == 1) {
    for (int i = 0; i < g.nprob; ++i) {
      GemmProblem& p = g.p[i];
      p.ksplit = 1;
      if (p.M <= 0 || p.N <= 0) continue;
      dim3 grid((p.N + 15) / 16, (p.M + 15) / 16);
      if (LAYOUT == L_NT) launch_naive_L0(grid, s, p);
      else if (LAYOUT == L_NN) launch_naive_L1(grid, s, p);
      else launch_naive_L2(grid, s, p);
      LIREC_CHECK_LAUNCH();
    }
    return LIREC_OK;
  }
  long t256 = 0, t128 = 0, t64 = 0, mn_total = 0;
  bool splittable = g_scratch != nullptr && !t_no_split, wide = true, wide256 = true, deep = true, any_epi = false;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProblem& p = g.p[i];
    t256 += (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    t128 += (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    t64 += (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    mn_total += (long)p.M * p.N + (p.dbias ? p.M : 0);
    // (a problem with an epilogue CAN be split -- the reduce kernel then runs the epilogue -- but for the candidates, the
    //  under-filled 3072-deep gate GEMMs, two k-chunks measured SLOWER: gate_fwd 103 -> 107 us, gate_dEE 91 -> 115 us;
    //  the slab round trip and the per-element Philox of the reduce outweigh the better fill.  Diagnostic bit 128.)
    splittable = splittable && p.K >= 512 && (p.epi == EPI_STORE || (p.rowmap == nullptr && p.dyn == nullptr && p.K >= 2048 &&
                                                                     (g_ablate & 128)));
    any_epi = any_epi || p.epi != EPI_STORE;
    wide = wide && p.M >= 96 && p.N >= 96;
    wide256 = wide256 && p.M >= 192 && p.N >= 192;
    deep = deep && p.K >= 4096;
  }
  // Tile choice (bf16x3 core: TileCfg in gemm_bf16x3.hpp; the f32 core only has 64x64 and 128x128).
  // 256x256 (one workgroup per CU) when it fills the chip by itself or the reduction is deep enough to be
  // split; 128x128 when that still gives >= 1.5 workgroups per CU (or split-K can make up the difference);
  // 64x64 otherwise.
  // (measured on the K1 / dW1 shapes: the 256x256 tile wins only for the deep split-K weight gradients;
  //  for the forward GEMMs its 576 tiles on 256 CUs lose more to the partial last round than they gain)
  static const int cfg_bm[5] = {64, 128, 256, 128, 256}, cfg_bn[5] = {64, 128, 256, 128, 128};
  const bool huge = g_bf_core && wide256 && splittable && deep && !any_epi;
  // (NN: the 8-wave 128x128 tile already wins at 192 tiles -- gate dEE 0.095 vs 0.106 ms -- but not at 128 -- dZ1)
  const bool big = !huge && (t128 >= (LAYOUT == L_NN ? 160 : 384) || (splittable && wide));
  int cfg = huge ? 2 : (big ? 3 : 0);
  // (f32-input core, data gradients that cannot be split: fewer 128 x 128 tiles than CUs and a deep reduction -- each tile alone on its
  //  CU pays every k-tile's load latency -- run faster as 64 x 64 tiles, three to a CU: the gate's data gradient 362 -> 245 us; the
  //  hidden-layer gradient, 128 tiles of 48 k-tiles, does not: 166 against 273)
  if (!g_bf_core && LAYOUT == L_NN && !splittable && cfg != 0 && t128 < 256) {
    long kmin = 1L << 40;
    for (int i = 0; i < g.nprob; ++i) kmin = g.p[i].K < kmin ? g.p[i].K : kmin;
    if (kmin >= 2048) cfg = 0;
  }
  if (g_force_cfg >= 0 && g_force_cfg < 5) cfg = g_force_cfg;
  if (cfg == 2 && LAYOUT != L_TN) cfg = 3;
  if (!g_bf_core) cfg = (cfg == 0) ? 0 : 1;
  const int bm = cfg_bm[cfg], bn = cfg_bn[cfg];
  long t0 = 0;
  for (int i = 0; i < g.nprob; ++i) t0 += (long)((g.p[i].M + bm - 1) / bm) * ((g.p[i].N + bn - 1) / bn);
  const long fill = (bm * bn > 128 * 128) ? 256 : 512;            // workgroups resident at once
  int ksplit_want = 1;
  if (splittable && cfg == 0) {
    if (t0 < fill) ksplit_want = (int)((3 * fill / 2 + t0 - 1) / t0);
  } else if (splittable) {
    // Large tiles: pick the split that minimises  rounds x k-tiles per chunk x time per k-tile  +  the cost of the
    // partial tiles (written once, read once by the reduce kernel, plus that kernel's launch).  Measured per
    // k-tile with a CU to itself: 1.0 us for a 128x128 workgroup (1.2 us per workgroup when two share the CU),
    // 1.6 us for 256x128, 2.8 us for 256x256.  (The first rule -- split until 1.5 workgroups per CU -- split the
    // interaction head's dW1 four ways, 46 us + 37 us of reduce, where one round of unsplit tiles takes ~40 us.)
    long kmax = 0, kmin = 1L << 40;
    for (int i = 0; i < g.nprob; ++i) { kmax = g.p[i].K > kmax ? g.p[i].K : kmax; kmin = g.p[i].K < kmin ? g.p[i].K : kmin; }
    // (the f32-input core: a 128 x 128 k-tile is 64 MFMAs of 64 cycles per wave, ~3.4 us with a CU to itself -- the loads of the next
    //  k-tile are not covered -- and ~4 us per workgroup when two share the CU, which its 68 KiB of LDS allow: 512 at once)
    const double tk_alone = !g_bf_core ? 3.4 : ((bm * bn == 256 * 256) ? 2.8 : (bm * bn == 256 * 128 ? 1.6 : 1.0));
    const double tk_shared = !g_bf_core ? 4.0 : ((bm * bn == 128 * 128) ? 1.2 : tk_alone);
    double best = 1e30;
    for (int ks = 1; ks <= 32; ++ks) {
      if (ks > 1 && kmin / ks < 256) break;                       // >= 8 k-tiles per chunk
      if (ks > 1 && (long)ks * mn_total > g_scratch_floats) break;  // all partial tiles must fit the scratch
      const long blocks = t0 * ks;
      // (a launch with an epilogue is one of the under-filled data-path GEMMs: its 128x128 workgroups sit two to a CU,
      //  512 at once; the weight-gradient launches keep the 256 their rule was fitted with)
      const long slots = ((any_epi || !g_bf_core) && bm * bn == 128 * 128) ? 512 : 256;
      const double rounds = (double)((blocks + slots - 1) / slots);
      const double nk = (double)((kmax + ks - 1) / ks + 31) / 32;
      double cost = rounds * nk * (blocks > 256 ? tk_shared : tk_alone);
      if (ks > 1) cost += 6.0 + (double)(ks + 1) * (double)mn_total * 4.0 / 4.0e6;   // us at ~4 TB/s
      if (cost < best) { best = cost; ksplit_want = ks; }
    }
  }
  if (ksplit_want > 1 && mn_total > 0) {                        // all partial tiles must fit the scratch
    const long fit = g_scratch_floats / mn_total;
    if (fit < ksplit_want) ksplit_want = fit > 1 ? (int)fit : 1;
  }
  bool any_split = false;
  const int start = plan_tiles(g, bm, bn, ksplit_want, any_split);
  if (start == 0) return LIREC_OK;
  // tagged symbols exist only where the tag is used: 1 with NT, 2 with TN (and only in the
  // dwordx4-staging build: the heavy call sites are always aligned)
  constexpr int T1 = (LAYOUT == L_NT) ? 1 : (LAYOUT == L_TN ? 2 : 0);
  bool vec = true;
  for (int i = 0; i < g.nprob; ++i) vec = vec && gemm_problem_is_vec(LAYOUT, g.p[i]);
  bool mapped = (LAYOUT == L_TN);
  for (int i = 0; i < g.nprob; ++i) mapped = mapped && g.p[i].rowmap != nullptr;
  bool unmapped = true;
  for (int i = 0; i < g.nprob; ++i) unmapped = unmapped && g.p[i].rowmap == nullptr;
  // the tagged TN symbols fix "row-mapped or not" at compile time; a mixed group takes the generic kernel
  int variant = !vec ? GV_SCALAR : ((T1 != 0 && meta.tag == T1) ? GV_TAGGED : GV_VEC);
  if (LAYOUT == L_TN && variant == GV_TAGGED) variant = mapped ? GV_MAPPED : (unmapped ? GV_TAGGED : GV_VEC);
  // bf16-stored X operand: every problem of the launch or none; tagged dword-aligned launches of the bf16x3 core only
  bool xb_all = true, xb_any = false;
  for (int i = 0; i < g.nprob; ++i) { xb_all = xb_all && g.p[i].x_bf16; xb_any = xb_any || g.p[i].x_bf16; }
  if (xb_any) {
    if (!xb_all || !g_bf_core || LAYOUT == L_NN || (variant != GV_TAGGED && variant != GV_MAPPED)) return LIREC_EINVAL;
    variant = (variant == GV_MAPPED) ? GV_MAPPED_XB : GV_TAGGED_XB;
  }
  const dim3 grid(start);
  typedef void (*bf_fn)(int, dim3, hipStream_t, const GemmGroup&);
  typedef void (*f32_fn)(bool, int, dim3, hipStream_t, const GemmGroup&);
  static const bf_fn bf_table[3][5] = {
      // (the 256 x 256 configuration is built for the weight-gradient layout only: no forward / data-gradient call site ever
      //  selected it, and its NT / NN instantiations carried 576 B of scratch per lane)
      {launch_bf_L0_C0, launch_bf_L0_C1, launch_bf_L0_C3, launch_bf_L0_C3, launch_bf_L0_C4},
      {launch_bf_L1_C0, launch_bf_L1_C1, launch_bf_L1_C3, launch_bf_L1_C3, launch_bf_L1_C4},
      {launch_bf_L2_C0, launch_bf_L2_C1, launch_bf_L2_C2, launch_bf_L2_C3, launch_bf_L2_C4}};
  static const f32_fn f32_table[3] = {launch_f32_L0, launch_f32_L1, launch_f32_L2};
  g.onepass = (g_gemm_mode == 3) && (meta.site == PS_EMBED_L1_FWD || meta.site == PS_EMBED_DW1 || meta.site == PS_GATE_FWD ||
                                     meta.site == PS_GATE_DW || meta.site == PS_GATE_DEE);
  if (g_bf_core) bf_table[LAYOUT][cfg](variant, grid, s, g);
  else f32_table[LAYOUT](cfg != 0, variant, grid, s, g);
  if (any_split) {
    LIREC_CHECK_LAUNCH();
    launch_splitk_reduce(g, s);
  }
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}


// ---------------------------------------------------------------------------
// Layer 1 on q32b operands (gemm_p2.hpp)
// ---------------------------------------------------------------------------
struct PlaneLayout {
  bool gather;                             // rows are not staged: gathered from q32b storage through `srow`
  bool staged16;                           // x_bf16: the rows ARE staged, as q16b (one plane; q16c in the single-pass mode), and read by the
                                           // gathering kernels through an identity list (`gather` is true as well, `xq` holds the staged rows)
  int* srow[3];                            // gather: row lists (block: [0]; pieces: clip, track 1, track 2), rows32 ints each
  unsigned char* xq;                       // feature rows, q32b [rows32][dsum] (NULL when gathered)
  unsigned char* wq[LIREC_MAX_SEG];        // first-layer weights of each segment, q32b [J][in_dim]
  unsigned char* keep;                     // dropout keep bytes of H1 [rows32 / 4][nseg * J]
  int* nt_bound;                           // one int: the forward partition's bound, left by the staging launch (p2_partition.hpp)
  int dsum, c0, rows32;
};

static int p2_grid() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 8) cus = 8;
    cus -= cus % 8;
  }
  return cus;
}
// floats of split-K scratch the weight-gradient launch needs (two partial tiles + bias-gradient rows per workgroup)
static long p2_scratch_floats() { return 2L * p2_grid() * (256L * 256L + 256L); }

// rows given as piece tables + index (forward: the `pieces` field; backward: no X at all -- the q32b rows are in `planes`)
static const lirec_pieces* pieces_of(const lirec_embed_fwd_args* a) { return a->pieces; }
static const lirec_pieces* pieces_of(const lirec_embed_bwd_args* a) { return a->pieces; }
static bool rows_without_x(const lirec_embed_fwd_args* a) { return a->pieces != nullptr; }
static bool rows_without_x(const lirec_embed_bwd_args* a) { return a->X == nullptr && a->planes != nullptr; }
// rows gathered from q32b storage (the block itself stored as q32b, or q32b piece tables): no staged copy
template <class Args>
static bool rows_gathered(const Args* a) {
  const lirec_pieces* pc = pieces_of(a);
  return a->x_q32 != 0 || (pc != nullptr && pc->clip_q != nullptr && pc->track_q != nullptr);
}

// Is the q32b path available for this head, and where do its parts lie in the `planes` workspace?
template <class Args>
static bool plane_layout(const Args* a, PlaneLayout& L) {
  // (default core: fp32 / q32b / q16b rows; the single-pass mode -- gemm mode 3 -- on bf16 values in 64-column blocks only: rows
  //  stored as q16c, or a row-major bf16 block staged as q16c, and the first-layer weights as q16c)
  const bool staged16 = a->x_bf16 != 0;    // a row-major bf16 block: its rows are staged as q16b (mode 3: q16c) for the one-plane kernels
  if (!((g_gemm_mode == 2 && a->x_q32 != 3) || (g_gemm_mode == 3 && (a->x_q32 == 3 || staged16))) || !a->planes || a->rows < 1 || (g_ablate & 8)) return false;
  if (staged16 && (a->x_q32 != 0 || pieces_of(a) || rows_without_x(a) || !a->X || (reinterpret_cast<uintptr_t>(a->X) & 15) != 0 || ((a->ldx * 2) & 15) != 0))
    return false;
  const bool gather = rows_gathered(a);
  if (a->x_q32 >= 2 && pieces_of(a)) return false;             // (q16b / q16c storage: the block form only)
  if (const lirec_pieces* pc = pieces_of(a)) {
    // the four segments must be the pieces' columns: text | clip-visual | track-1 | track-2 from column 0
    if (a->nseg != 4 || a->in_off[0] != 0 || !pc->index) return false;
    if (a->in_dim[0] != pc->text_dim || a->in_dim[1] != pc->visual_dim || a->in_dim[2] != pc->track_dim || a->in_dim[3] != pc->track_dim) return false;
    if (gather) {
      if (((reinterpret_cast<uintptr_t>(pc->clip_q) | reinterpret_cast<uintptr_t>(pc->track_q)) & 255) != 0) return false;
    } else {
      if (!pc->clip || !pc->track) return false;
      if (((reinterpret_cast<uintptr_t>(pc->clip) | reinterpret_cast<uintptr_t>(pc->track)) & 15) != 0 || ((pc->ld_clip | pc->ld_track) & 3) != 0) return false;
    }
  } else if (gather) {
    if (!a->X || (reinterpret_cast<uintptr_t>(a->X) & 255) != 0 || (a->ldx & 31) != 0) return false;
  }
  int dsum = 0;
  for (int i = 0; i < a->nseg; ++i) {
    if (a->in_dim[i] % 256 != 0) return false;
    if (i > 0 && a->in_off[i] != a->in_off[i - 1] + a->in_dim[i - 1]) return false;     // adjacent segments
    dsum += a->in_dim[i];
  }
  if (a->J % 256 != 0 || (a->in_off[0] & 7) != 0 || p2_grid() % (a->J / 256) != 0) return false;
  if (!gather && !staged16 && !rows_without_x(a) && ((reinterpret_cast<uintptr_t>(a->X) & 15) != 0 || ((a->ldx * 4) & 15) != 0)) return false;
  if ((gather || staged16) && (a->in_off[0] & (g_gemm_mode == 3 ? 63 : 31)) != 0) return false;
  if (g_gemm_mode == 3 && a->x_q32 == 3 && (a->ldx & 63) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a->planes) & 255) != 0) return false;
  if (a->planes_bytes < lirec_planes_bytes(a->rows, dsum, a->J, staged16 ? 1 : (gather ? 2 : 0))) return false;
  if (g_scratch_floats < p2_scratch_floats()) return false;
  const int64_t rp = (a->rows + 31) / 32 * 32;
  char* base = reinterpret_cast<char*>(a->planes);
  L.gather = gather || staged16;
  L.staged16 = staged16;
  L.xq = gather ? nullptr : reinterpret_cast<unsigned char*>(base);
  if (staged16) base += align256(rp * dsum * 2);
  else if (!gather) base += 2 * align256(rp * dsum * 2);
  long off = 0;
  for (int i = 0; i < a->nseg; ++i) { L.wq[i] = reinterpret_cast<unsigned char*>(base) + off; off += 4L * a->J * a->in_dim[i]; }
  L.keep = reinterpret_cast<unsigned char*>(base) + 2 * align256((int64_t)a->J * dsum * 2);
  L.nt_bound = reinterpret_cast<int*>(L.keep + align256(rp / 4 * LIREC_MAX_SEG * (int64_t)a->J));
  for (int k = 0; k < 3; ++k) L.srow[k] = L.nt_bound + 64 + k * rp;
  L.dsum = dsum; L.c0 = a->in_off[0]; L.rows32 = (int)rp;
  return true;
}

static int launch_split(const SplitSegs& q, hipStream_t s) {
  if (q.nseg == 0 || q.first[q.nseg] == 0) return LIREC_OK;
  long blocks = (q.first[q.nseg] + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const int pi = prof_start(PS_STAGE, s);
  lirec::launch(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q);
  prof_stop(pi, s, 0.0, 64.0 * (double)q.first[q.nseg]);          // 32 B read + 2 x 16 B written per 8 elements
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}
static bool split_add(SplitSegs& q, const float* src, unsigned short* hi, unsigned short* lo, long n) {
  if (q.nseg >= 8 || (n & 7) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15))
    return false;
  q.src[q.nseg] = src; q.hi[q.nseg] = hi; q.lo[q.nseg] = lo;
  q.first[q.nseg + 1] = q.first[q.nseg] + n / 8;
  ++q.nseg;
  return true;
}
// first-layer weights -> q32b
static bool splitq_add(SplitQ32b& q, const float* src, unsigned char* dst, int rows, int cols) {
  if (q.nseg >= 8 || (rows & 31) || (cols & 31) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return false;
  q.src[q.nseg] = src; q.dst[q.nseg] = dst; q.cols[q.nseg] = cols;
  q.first[q.nseg + 1] = q.first[q.nseg] + (long)rows * cols / 8;
  ++q.nseg;
  return true;
}
static int launch_splitq(const SplitQ32b& q, hipStream_t s) {
  if (q.nseg == 0 || q.first[q.nseg] == 0) return LIREC_OK;
  long blocks = (q.first[q.nseg] + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const int pi = prof_start(PS_STAGE, s);
  lirec::launch(split_q32b_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q);
  prof_stop(pi, s, 0.0, 64.0 * (double)q.first[q.nseg]);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// feature rows of one head -> q32b
template <class Args>
static int launch_stage(const Args* a, const PlaneLayout& L, hipStream_t s, const lirec_dropout* drop = nullptr) {
  const int D8 = L.dsum / 8;
  StageDrop dk;
  memset(&dk, 0, sizeof(dk));
  if (drop && drop->p > 0.f) {
    dk.keep = L.keep; dk.ld = (long)a->nseg * a->J; dk.ncol = a->nseg * a->J;
    dk.seed_lo = (unsigned)(drop->seed & 0xffffffffull); dk.seed_hi = (unsigned)(drop->seed >> 32);
    dk.seed_dev = (const unsigned long long*)drop->seed_dev; dk.site = (unsigned)drop->site; dk.thresh = drop_thresh(drop->p);
  }
  const long total = (long)L.rows32 * D8;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dk.keep) blocks = blocks + blocks / 2;                  // (+ the workgroups that produce the dropout keep bytes)
  const int pi = prof_start(PS_STAGE, s);
  lirec::launch(stage_rows_q32b_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a->X + L.c0, (long)a->ldx, a->sel.group,
                     a->sel.group_stride, a->sel.group_off, a->rowmap, a->count, a->rows, D8, L.xq, dk);
  // static row count (the library does not read the device-side count back): read 4 B, write 4 B per element
  prof_stop(pi, s, 0.0, 8.0 * (double)a->rows * L.dsum);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// the same as a role of the fused staging launch (stage_fused_kernel)
template <class Args>
static void stage_head_fill(StageHead& h, const Args* a, const PlaneLayout& L) {
  memset(&h, 0, sizeof(h));
  h.X = a->X ? (L.staged16 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(a->X) + 2L * L.c0) : a->X + L.c0) : nullptr;
  h.ldx = (long)a->ldx; h.gs = a->sel.group; h.gstride = a->sel.group_stride; h.goff = a->sel.group_off;
  if (const lirec_pieces* pc = pieces_of(a)) {
    h.src.clip = pc->clip; h.src.track = pc->track; h.src.index = pc->index; h.src.ld_clip = pc->ld_clip; h.src.ld_track = pc->ld_track;
    h.src.clip_dim = pc->text_dim + pc->visual_dim; h.src.track_dim = pc->track_dim; h.src.c0 = L.c0;
  }
  h.rowmap = a->rowmap; h.count = a->count; h.rows = a->rows; h.D8 = L.dsum / 8; h.dst = L.xq;
  if (L.staged16) {
    h.src.x16 = g_gemm_mode == 3 ? 2 : 1; h.src.srow[0] = L.srow[0];     // bf16 rows -> q16b (single pass: q16c) rows + the identity list
  } else if (L.gather) {
    for (int k = 0; k < 3; ++k) h.src.srow[k] = L.srow[k];
    if (const lirec_pieces* pc = pieces_of(a)) {
      h.src.index = pc->index; h.src.clip_rows = pc->clip_rows; h.src.track_rows = pc->track_rows;
      h.src.zero_clip = pc->n_clip; h.src.zero_track = pc->n_track;
    }
  }
  const lirec_dropout* drop = &a->drop;
  if (drop->p > 0.f) {
    h.dk.keep = L.keep; h.dk.ld = (long)a->nseg * a->J; h.dk.ncol = a->nseg * a->J;
    h.dk.seed_lo = (unsigned)(drop->seed & 0xffffffffull); h.dk.seed_hi = (unsigned)(drop->seed >> 32);
    h.dk.seed_dev = (const unsigned long long*)drop->seed_dev; h.dk.site = (unsigned)drop->site; h.dk.thresh = drop_thresh(drop->p);
  }
  long blocks = ((long)L.rows32 * h.D8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (h.dk.keep) blocks = blocks + blocks / 2;                // (+ the workgroups that produce the dropout keep bytes)
  if (L.gather && !L.staged16) {
    // eight workgroups for the row lists + the mask tasks' share (one task = 4 rows x 4 columns)
    blocks = 8;
    if (h.dk.keep) { long mb = ((long)(L.rows32 / 4) * (h.dk.ncol / 4) + 255) / 256; blocks += mb > 2048 ? 2048 : (mb < 1 ? 1 : mb); }
  }
  h.blocks = (int)(blocks < 1 ? 1 : blocks);
}
// the row operand of segment i of a head whose rows are gathered: base of the q32b matrix at the segment's first column block,
// its columns, and the row list
template <class Args>
static void gather_operand(const Args* a, const PlaneLayout& L, int i, const float*& base, long& ld, const int*& srow) {
  if (const lirec_pieces* pc = pieces_of(a)) {
    const long cd = pc->text_dim + pc->visual_dim;
    if (i < 2) { base = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(pc->clip_q) + 4096L * (i == 0 ? 0 : pc->text_dim / 32)); ld = cd; srow = L.srow[0]; }
    else { base = reinterpret_cast<const float*>(pc->track_q); ld = pc->track_dim; srow = L.srow[i - 1]; }
  } else if (L.staged16) {
    // (a bf16 block staged as q16b into the workspace: dense rows behind an identity list)
    // (single-pass mode: staged as q16c -- 4-KiB blocks of 64 columns)
    base = reinterpret_cast<const float*>(L.xq + (g_gemm_mode == 3 ? 4096L * ((a->in_off[i] - L.c0) / 64) : 2048L * ((a->in_off[i] - L.c0) / 32)));
    ld = L.dsum; srow = L.srow[0];
  } else {
    // (x_q32 = 2: the block is stored as q16b -- bf16 values, 2-KiB blocks; 3: q16c -- 4-KiB blocks of 64 columns)
    const long boff = a->x_q32 == 3 ? 4096L * (a->in_off[i] / 64) : (a->x_q32 == 2 ? 2048L : 4096L) * (a->in_off[i] / 32);
    base = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a->X) + boff); ld = a->ldx; srow = L.srow[0];
  }
}
// form of the gathered rows of a call: 1 when every head's rows are q16b (stored, or a bf16 block staged), 2 for q32b, 3 for q16c
// (stored, or -- single-pass mode -- a bf16 block staged); 0 = the heads disagree
template <class Args>
static int gather_planes(const Args* const* hs, int nh) {
  int xp = 0;
  for (int h = 0; h < nh; ++h) {
    const int v = hs[h]->x_q32 == 3 ? 3 : (hs[h]->x_bf16 ? (g_gemm_mode == 3 ? 3 : 1) : (hs[h]->x_q32 == 2 ? 1 : 2));
    if (xp && v != xp) return 0;
    xp = v;
  }
  return xp;
}

// lirec_fused_adam -> the kernels' AdamFuse, checked against the weight-gradient problems of the launch: every gradient of the
// call inside the flat buffer, and together exactly the parameters the caller counts on; sets each problem's aux_out to where
// the q32b form of its new weights goes (or NULL)
static int fused_adam_fill(const lirec_fused_adam* adam, GemmGroup& g, AdamFuse& af) {
  if (!adam->p || !adam->g || !adam->m || !adam->v || adam->n < 1 || (adam->step < 1 && !adam->step_dev)) return LIREC_EINVAL;
  if (((reinterpret_cast<uintptr_t>(adam->p) | reinterpret_cast<uintptr_t>(adam->g) | reinterpret_cast<uintptr_t>(adam->m) |
        reinterpret_cast<uintptr_t>(adam->v)) & 15) != 0 || (reinterpret_cast<uintptr_t>(adam->wq) & 255) != 0) return LIREC_EINVAL;
  int64_t covered = 0;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProblem& q = g.p[i];
    if (q.ldc != q.N || q.C < adam->g || q.C + (long)q.M * q.N > adam->g + adam->n || ((q.C - adam->g) & 3) != 0) return LIREC_EINVAL;
    covered += (int64_t)q.M * q.N;
    if (q.dbias) {
      if (q.dbias < adam->g || q.dbias + q.M > adam->g + adam->n) return LIREC_EINVAL;
      covered += q.M;
    }
    g.p[i].aux_out = adam->wq ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(adam->wq) + 4 * ((q.C - adam->g) - adam->wq_first)) : nullptr;
    if (adam->wq && ((q.C - adam->g) < adam->wq_first || (reinterpret_cast<uintptr_t>(g.p[i].aux_out) & 255) != 0 || (q.M & 31) || (q.N & 31)))
      return LIREC_EINVAL;
  }
  if (covered != adam->n_params) return LIREC_EINVAL;
  const int step = adam->step < 1 ? 1 : adam->step;
  af.p = adam->p; af.g = adam->g; af.m = adam->m; af.v = adam->v;
  af.step_size = (float)((double)adam->lr / (1.0 - pow((double)adam->beta1, (double)step)));
  af.bc2_sqrt = (float)sqrt(1.0 - pow((double)adam->beta2, (double)step));
  af.beta1 = adam->beta1; af.beta2 = adam->beta2; af.eps = adam->eps; af.wd = adam->weight_decay; af.gscale = adam->grad_scale;
  af.lr = adam->lr; af.step_dev = (const long long*)adam->step_dev;
  af.wq16c = g_gemm_mode == 3 ? 1 : 0;      // (the single-pass mode keeps its first-layer weights as q16c)
  return LIREC_OK;
}

// persistent launch of the q32b kernels over the problems of `g0` (every problem: the same 256-wide replica count)
template <int LAYOUT>
static int launch_p2(GemmGroup& g0, hipStream_t s, int site, const int* nt_bound = nullptr, int ct_major = 0, bool gather = false,
                     const lirec_fused_adam* adam = nullptr, int xp = 2) {
  if (xp != 2 && ((xp != 1 && xp != 3) || !gather || LAYOUT == L_NN)) return LIREC_EINVAL;      // (one-plane rows: gathered q16b / q16c storage)
  if ((g_gemm_mode == 3) != (xp == 3)) return LIREC_EINVAL;                           // (single pass: q16c operands, and nothing else on them)
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.ablate = g_ablate; g.dyn_is_k = (LAYOUT == L_TN);
  g.nt_bound = nt_bound; g.nt_ct_major = ct_major;
  double flops = 0.0;
  int nrep = 0, tiles = 0;
  for (int i = 0; i < g0.nprob; ++i)
    if (g0.p[i].M > 0 && g0.p[i].N > 0) {
      const int r = (LAYOUT != L_TN ? g0.p[i].N : g0.p[i].M) / 256;
      if (nrep && r != nrep) return LIREC_EINVAL;
      nrep = r;
      tiles += (g0.p[i].N / 256) * (g0.p[i].M / 256);
      g.p[g.nprob++] = g0.p[i];
      flops += 2.0 * g0.p[i].M * (double)g0.p[i].N * g0.p[i].K * ((xp == 3 && LAYOUT == L_NT) ? 2.0 : 1.0);     // (q16c forward: K counts 64-k steps' halves)
    }
  if (adam && LAYOUT != L_TN) return LIREC_EINVAL;
  AdamFuse af;
  memset(&af, 0, sizeof(af));
  if (adam) {
    if (g.nprob != g0.nprob) return LIREC_EINVAL;                // (a problem was dropped as empty: its parameters would miss the update)
    const int rc = fused_adam_fill(adam, g, af);
    if (rc) return rc;
  } else if (LAYOUT == L_TN) {
    for (int i = 0; i < g.nprob; ++i) g.p[i].aux_out = nullptr;
  }
  if (g.nprob == 0) return LIREC_OK;
  if (LAYOUT == L_TN) ow_note_group(g);
  const int G = p2_grid();
  if (LAYOUT != L_TN && !nt_bound) {
    // static row counts only (no problem carries a device-side bound): the partition bound is computed here, once
    bool all_static = true;
    int rbv[LIREC_MAX_PROB], ksv[LIREC_MAX_PROB];
    for (int i = 0; i < g.nprob; ++i) { all_static = all_static && !g.p[i].dyn; rbv[i] = (g.p[i].M + 31) >> 5; ksv[i] = g.p[i].K >> 5; }
    if (all_static) {
      g.nt_bound_val = p2_nt_bound_host(rbv, ksv, g.nprob, G, nrep);
      if (g.nt_bound_val <= 0) return LIREC_EINVAL;
    }
  }
  const int pi = prof_start(site, s);
  if (LAYOUT == L_NT) {
    if (gather && xp == 3) launch_p2_ntg64(dim3(G), s, g, nrep);
    else if (gather && xp == 1) launch_p2_ntg1(dim3(G), s, g, nrep);
    else if (gather) launch_p2_ntg(dim3(G), s, g, nrep);
    else launch_p2_nt(dim3(G), s, g, nrep);
    prof_stop(pi, s, flops, 0.0);
  } else if (LAYOUT == L_NN) {
    launch_p2_nn(dim3(G), s, g, nrep);
    prof_stop(pi, s, flops, 0.0);
  } else {
    g.p[0].slab = g_scratch;
    g.p[0].dbias_slab = g_scratch + 2L * G * 256 * 256;
    if (gather && xp == 3) launch_p2_tng1o(dim3(G), s, g, nrep);
    else if (gather && xp == 1) launch_p2_tng1(dim3(G), s, g, nrep);
    else if (gather) launch_p2_tng(dim3(G), s, g, nrep);
    else launch_p2_tn(dim3(G), s, g, nrep);
    prof_stop(pi, s, flops, 0.0);
    // (a site of its own: the `embed_dW1` figure is then the GEMM kernel's, the one a kernel trace lists under its name)
    // bytes: at most two partial tiles per workgroup read, every output tile written once
    const int pr = prof_start(PS_EMBED_DW1_RED, s);
    launch_p2_tn_reduce(tiles, G, s, g, nrep, adam ? &af : nullptr);
    // (+ the update: p, m, v read and written, the q32b shadow written)
    prof_stop(pr, s, 0.0, 4.0 * 65536.0 * (2.0 * G + tiles) + (adam ? (adam->wq ? 28.0 : 24.0) * (double)adam->n_params : 0.0));
  }
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// Do all `nh` heads of a call qualify for the q32b path?
template <class Args>
static bool planes_for_heads(const Args* const* hs, int nh, PlaneLayout* L) {
  int nseg = 0;
  for (int h = 0; h < nh; ++h) {
    if (!plane_layout(hs[h], L[h]) || hs[h]->J != hs[0]->J || L[h].gather != L[0].gather) return false;
    nseg += hs[h]->nseg;
  }
  return nseg <= LIREC_MAX_PROB;
}

static int launch_gemm(int layout, GemmGroup& g, hipStream_t s, int site, int tag = 0) {
  const GemmMeta meta = {site, tag};
  if (layout == L_TN) ow_note_group(g);
  // drop empty problems (a zero-tile problem must not shadow its successor's tile_start)
  GemmGroup h;
  memset(&h, 0, sizeof(h));
  h.ablate = g_ablate; h.dyn_is_k = (layout == L_TN);
  for (int i = 0; i < g.nprob; ++i)
    if (g.p[i].M > 0 && g.p[i].N > 0) h.p[h.nprob++] = g.p[i];
  switch (layout) {
    case L_NT: return launch_layout<L_NT>(h, meta, s);
    case L_NN: return launch_layout<L_NN>(h, meta, s);
    default: return launch_layout<L_TN>(h, meta, s);
  }
}

extern "C" {

int lirec_version(void) { return LIREC_VERSION; }

int lirec_set_gemm_mode(int mode) {
  if (mode < 0 || mode > 3) return LIREC_EINVAL;
  g_gemm_mode = mode;
  return LIREC_OK;
}

int lirec_set_scratch(void* ptr, int64_t bytes) {
  if (bytes < 0 || (ptr == nullptr && bytes != 0)) return LIREC_EINVAL;
  g_scratch = (float*)ptr;
  g_scratch_floats = ptr ? (long)(bytes / (int64_t)sizeof(float)) : 0;
  return LIREC_OK;
}

/* diagnostics: k-loop ablation mask, forced tile configuration (current context) */
int lirec_debug_set(int ablate, int force_cfg) {
  g_ablate = ablate; g_force_cfg = force_cfg;
  lirec::g_dry = (ablate & 4194304) != 0;        // (process-wide, unlike the other bits: record.hpp)
  return 0;
}

int lirec_get_gemm_mode(void) { return g_gemm_mode; }

int lirec_set_grad_overwrite(int on) {
  g_grad_overwrite = on ? 1 : 0;
  if (on) { t_ow_targets.clear(); t_ow_conflicts = 0; }
  return LIREC_OK;
}
int lirec_grad_overwrite_conflicts(void) { return t_ow_conflicts; }

int lirec_ctx_create(lirec_ctx_t* out) {
  if (!out) return LIREC_EINVAL;
  lirec_ctx* c = new (std::nothrow) lirec_ctx();
  if (!c) return LIREC_EINVAL;
  c->gemm_mode = g_default_ctx.gemm_mode;
  *out = c;
  return LIREC_OK;
}

int lirec_ctx_destroy(lirec_ctx_t ctx) {
  lirec_ctx* c = static_cast<lirec_ctx*>(ctx);
  if (!c || c == &g_default_ctx) return LIREC_EINVAL;
  if (t_ctx == c) t_ctx = &g_default_ctx;
  delete c;
  return LIREC_OK;
}

int lirec_ctx_set_current(lirec_ctx_t ctx) {
  t_ctx = ctx ? static_cast<lirec_ctx*>(ctx) : &g_default_ctx;
  return LIREC_OK;
}

lirec_ctx_t lirec_ctx_get_current(void) { return t_ctx == &g_default_ctx ? nullptr : t_ctx; }

// ---------------------------------------------------------------------------
// command lists (record.hpp)
// ---------------------------------------------------------------------------

int lirec_record_begin(void) {
  if (lirec::t_rec) return LIREC_EINVAL;                     // already recording on this thread
  lirec_cmdlist* l = new (std::nothrow) lirec_cmdlist();
  if (!l) return LIREC_EINVAL;
  lirec::t_rec = &l->list;
  return LIREC_OK;
}

int32_t lirec_record_mark(void) { return lirec::t_rec ? (int32_t)lirec::t_rec->cmds.size() : -1; }

int lirec_record_end(lirec_cmdlist_t* out) {
  if (!lirec::t_rec || !out) return LIREC_EINVAL;
  *out = reinterpret_cast<lirec_cmdlist*>(lirec::t_rec);      // `list` is the first (only) member
  lirec::t_rec = nullptr;
  return LIREC_OK;
}

int32_t lirec_cmdlist_size(lirec_cmdlist_t l) { return l ? (int32_t)l->list.cmds.size() : -1; }

int lirec_cmdlist_replay(lirec_cmdlist_t l, int32_t from, int32_t to) {
  if (!l || lirec::t_rec) return LIREC_EINVAL;                // (a replay is not recorded into another list)
  const int32_t n = (int32_t)l->list.cmds.size();
  if (to < 0 || to > n) to = n;
  if (from < 0 || from > to) return LIREC_EINVAL;
  for (int32_t i = from; i < to; ++i) l->list.cmds[i]();
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// Diagnostics (the dependency fuzzer of tests/test_gpu_recorded_bench_shape.py): the same replay with the stream of command
// `lag_at` held back by `ticks` of the 100 MHz clock in front of that command -- a kernel that starts late, or, for a stream wait,
// a signalling stream that reaches the recorded point late.  Whatever another stream reads of that command's results (or
// overwrites of its inputs) must be ordered by a recorded wait, not by the usual timing: the step's bits may not change.
int lirec_cmdlist_replay_lagged(lirec_cmdlist_t l, int32_t from, int32_t to, int32_t lag_at, int64_t ticks) {
  if (!l || lirec::t_rec) return LIREC_EINVAL;
  const int32_t n = (int32_t)l->list.cmds.size();
  if (to < 0 || to > n) to = n;
  if (from < 0 || from > to) return LIREC_EINVAL;
  for (int32_t i = from; i < to; ++i) {
    if (i == lag_at && ticks > 0 && !lirec::g_dry) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, l->list.streams[i], (long long)ticks);
    l->list.cmds[i]();
  }
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// (stream handle and kind -- 0 launch / memset, 1 stream wait (the signalling stream), 2 profiling bracket -- of command i)
int lirec_cmdlist_command(lirec_cmdlist_t l, int32_t i, lirec_stream_t* stream, int32_t* kind) {
  if (!l || i < 0 || i >= (int32_t)l->list.cmds.size()) return LIREC_EINVAL;
  if (stream) *stream = (lirec_stream_t)l->list.streams[i];
  if (kind) *kind = (int32_t)l->list.kinds[i];
  return LIREC_OK;
}

int lirec_cmdlist_destroy(lirec_cmdlist_t l) {
  if (l && lirec::t_rec == &l->list) lirec::t_rec = nullptr;
  delete l;
  return LIREC_OK;
}

// `waiter` waits for everything enqueued on `signaller` so far (fork / join of the weight-gradient side stream).  Eager calls
// take an event from a small per-thread ring (a wait refers to the record that precedes it, so reuse is safe); a recorded
// wait owns its event.
int lirec_stream_wait_many(const lirec_stream_t* waiters, int32_t n, lirec_stream_t signaller) {
  static thread_local hipEvent_t ring[16];
  static thread_local int ring_n = 0, ring_i = 0;
  if (n < 0 || n > 4 || (n > 0 && !waiters)) return LIREC_EINVAL;
  hipStream_t w[4] = {nullptr, nullptr, nullptr, nullptr};
  int nw = 0;
  for (int i = 0; i < n; ++i)
    if (waiters[i] != signaller) w[nw++] = (hipStream_t)waiters[i];
  if (nw == 0) return LIREC_OK;
  hipEvent_t ev = nullptr;
  if (lirec::g_dry) {                                          // (host-side dry run: the wait is recorded, nothing is issued)
    if (lirec::t_rec) lirec::t_rec->push([]() {}, (hipStream_t)signaller, 1);
    return LIREC_OK;
  }
  if (lirec::t_rec) {
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
    lirec::t_rec->events.push_back(ev);
    hipStream_t g = (hipStream_t)signaller, w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lirec::t_rec->push([=]() {
      (void)hipEventRecord(ev, g);
      (void)hipStreamWaitEvent(w0, ev, 0);
      if (nw > 1) (void)hipStreamWaitEvent(w1, ev, 0);
      if (nw > 2) (void)hipStreamWaitEvent(w2, ev, 0);
      if (nw > 3) (void)hipStreamWaitEvent(w3, ev, 0);
    }, g, 1);
  } else {
    if (ring_n < 16) {
      hipError_t e = hipEventCreateWithFlags(&ring[ring_n], hipEventDisableTiming);
      if (e != hipSuccess) return (int)e;
      ev = ring[ring_n++];
    } else {
      ev = ring[ring_i];
      ring_i = (ring_i + 1) % 16;
    }
  }
  hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
  for (int i = 0; e == hipSuccess && i < nw; ++i) e = hipStreamWaitEvent(w[i], ev, 0);
  return (int)e;
}

int lirec_stream_wait(lirec_stream_t waiter, lirec_stream_t signaller) { return lirec_stream_wait_many(&waiter, 1, signaller); }

int lirec_memset_zero(void* p, int64_t bytes, lirec_stream_t stream) {
  if (bytes < 0 || (!p && bytes > 0)) return LIREC_EINVAL;
  if (bytes == 0) return LIREC_OK;
  return (int)lirec::memset_async(p, 0, (size_t)bytes, (hipStream_t)stream);
}

int lirec_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(lirec_embed_fwd_args);
    case 1: return (int)sizeof(lirec_embed_bwd_args);
    case 2: return (int)sizeof(lirec_margin_loss_args);
    case 3: return (int)sizeof(lirec_dropout);
    case 4: return (int)sizeof(lirec_rowsel);
    case 5: return (int)sizeof(lirec_eval_args);
    case 6: return (int)sizeof(lirec_linear_fwd_args);
    case 7: return (int)sizeof(lirec_linear_bwd_args);
    default: return -1;
  }
}

const char* lirec_error_string(int code) {
  if (code == LIREC_OK) return "ok";
  if (code == LIREC_EINVAL) return "lirec: invalid argument";
  if (code == LIREC_EWORKSPACE) return "lirec: workspace too small";
  return hipGetErrorString((hipError_t)code);
}

int lirec_profile_enable(int on) {
  if (g_nrec) prof_flush();
  if (on) {
    for (int i = 0; i < PS_COUNT; ++i) { g_ms[i] = g_flops[i] = g_bytes[i] = 0.0; g_cnt[i] = 0; }
  }
  g_prof_on = on ? 1 : 0;
  return LIREC_OK;
}

int lirec_profile_sites(void) { return PS_COUNT; }

const char* lirec_profile_site_name(int site) { return (site >= 0 && site < PS_COUNT) ? g_site_names[site] : ""; }

int lirec_profile_read(int site, double* ms, int64_t* launches, double* flops, double* bytes) {
  if (site < 0 || site >= PS_COUNT) return LIREC_EINVAL;
  if (g_nrec) prof_flush();
  if (ms) *ms = g_ms[site];
  if (launches) *launches = g_cnt[site];
  if (flops) *flops = g_flops[site];
  if (bytes) *bytes = g_bytes[site];
  return LIREC_OK;
}

int64_t lirec_workspace_bytes(int32_t rows, int32_t nseg, int32_t J) {
  if (rows < 0 || nseg < 0 || J < 0) return -1;
  return 2 * ((int64_t)((rows + 31) / 32 * 32) + 32) * nseg * J * (int64_t)sizeof(float);
}

int64_t lirec_hbits_bytes(int32_t rows, int32_t W) {
  if (rows < 0 || W < 0) return -1;
  return (int64_t)rows * ((W + 255) / 256) * 32;
}

int64_t lirec_planes_bytes(int32_t rows, int32_t dsum, int32_t J, int32_t x_mode) {
  if (rows < 0 || dsum < 0 || J < 0) return -1;
  const int64_t rp = (rows + 31) / 32 * 32;
  const int64_t xplane = align256(rp * dsum * 2), wplane = align256((int64_t)J * dsum * 2);
  // (+ the dropout keep bytes of H1: one per four rows and hidden column, up to LIREC_MAX_SEG * J columns)
  // (+ 256 B: the forward partition's bound; + three row lists of rp ints: rows gathered from q32b storage)
  return (x_mode == 2 ? 0 : (x_mode == 1 ? 1 : 2)) * xplane + 2 * wplane + align256(rp / 4 * LIREC_MAX_SEG * (int64_t)J) + 256 + align256(12 * rp);
}

int64_t lirec_q32b_bytes(int64_t rows, int64_t cols) {
  if (rows < 0 || cols < 0 || (cols & 31) != 0) return -1;
  return align256((rows + 31) / 32 * 32 * cols * 4);
}

int64_t lirec_q16b_bytes(int64_t rows, int64_t cols) {
  if (rows < 0 || cols < 0 || (cols & 31) != 0) return -1;
  return align256((rows + 31) / 32 * 32 * cols * 2);
}

static int to_q16_impl(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int c64, lirec_stream_t stream);
int lirec_to_q16b(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream) {
  return to_q16_impl(src, ld_src, rows, cols, dst, 0, stream);
}
int lirec_to_q16c(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream) {
  return to_q16_impl(src, ld_src, rows, cols, dst, 1, stream);
}
static int to_q16_impl(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int c64, lirec_stream_t stream) {
  if (!src || !dst || rows < 0 || cols < 32 || (cols & (c64 ? 63 : 31)) != 0 || (ld_src & 3) != 0 || ld_src < cols ||
      (reinterpret_cast<uintptr_t>(src) & 15) != 0 || (reinterpret_cast<uintptr_t>(dst) & 255) != 0)
    return LIREC_EINVAL;
  if (rows == 0) return LIREC_OK;
  const long rows32 = (rows + 31) / 32 * 32, total = rows32 * (cols / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  lirec::launch(to_q16b_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, (long)rows, rows32, (int)(cols / 8),
                reinterpret_cast<unsigned char*>(dst), c64);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_to_q32b(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream) {
  if (!src || !dst || rows < 0 || cols < 32 || (cols & 31) != 0 || (ld_src & 3) != 0 || ld_src < cols ||
      (reinterpret_cast<uintptr_t>(src) & 15) != 0 || (reinterpret_cast<uintptr_t>(dst) & 255) != 0)
    return LIREC_EINVAL;
  if (rows == 0) return LIREC_OK;
  const long rows32 = (rows + 31) / 32 * 32, total = rows32 * (cols / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  lirec::launch(to_q32b_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, (long)rows, rows32, (int)(cols / 8),
                reinterpret_cast<unsigned char*>(dst));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// ---------------------------------------------------------------------------
static int launch_pool(const float* Z, long ldz, const float* mask, int n, int R, int W, int clamp_zero, float* Tn,
                       long ldtn, float* E, long lde, const lirec_dropout* drop, int plain, float* fout, hipStream_t s) {
  const float p = drop ? drop->p : 0.f;
  const uint64_t seed = drop ? drop->seed : 0;
  const int pi = prof_start(PS_POOL_FWD, s);
  lirec::launch(pool_fwd_kernel, dim3(n), dim3(256), 0, s, Z, ldz, mask, R, W, clamp_zero, Tn, ldtn, E, lde,
                     (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32),
                     (const unsigned long long*)(drop ? drop->seed_dev : nullptr), (unsigned)(drop ? drop->site2 : 0),
                     plain ? 0u : drop_thresh(p), (p > 0.f) ? (float)(1.0 / (1.0 - (double)p)) : 1.f, plain, fout);
  // algorithmic bytes of the pooling pass: n*R*W*4 + mask read, n*W*4 (x2 with E) written
  prof_stop(pi, s, 0.0, 4.0 * n * ((double)R * W + R + (plain ? 1.0 : 2.0) * W));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// argument checks + the two GEMM groups of one head's forward (layer 1, layer 2)
static int embed_fwd_build(const lirec_embed_fwd_args* a, GemmGroup& g1, GemmGroup& g2) {
  if (!a || (!a->X && a->parts != 2 && a->parts != 3 && !(a->pieces && a->planes)) || !a->H1 || !a->Z2 || a->nseg < 1 || a->nseg > LIREC_MAX_SEG || a->J < 1 || a->rows < 0)
    return LIREC_EINVAL;
  if (a->epilogue == 1 && !a->Tn) return LIREC_EINVAL;
  const bool pooled = a->mask != nullptr || a->rowmap != nullptr;
  if (pooled && (!a->Hbar || !a->fscale || a->R < 1 || a->rows % a->R != 0)) return LIREC_EINVAL;
  const bool compact = pooled && a->rowmap != nullptr;
  if ((a->rowmap || a->cstart || a->count) && !(a->rowmap && a->cstart && a->count && (a->mask || a->wts))) return LIREC_EINVAL;
  if (a->wts && !compact) return LIREC_EINVAL;
  const int J = a->J, nseg = a->nseg;
  const int n2 = pooled ? a->rows / a->R : a->rows;          // rows of the second layer
  g1.nprob = g2.nprob = nseg;
  int ooff = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!a->W1[i] || !a->W2[i] || a->in_dim[i] < 1 || a->out_dim[i] < 1) return LIREC_EINVAL;
    GemmProblem p = make_problem();
    p.A = a->X + a->in_off[i]; p.lda = a->ldx;
    p.gs = a->sel.group; p.gstride = a->sel.group_stride; p.goff = a->sel.group_off;
    p.gs_magic = row_magic(p.gs, a->rows);
    p.B = a->W1[i]; p.ldb = a->in_dim[i];
    p.bias = a->b1[i];
    p.C = a->H1 + (long)i * J; p.ldc = (long)nseg * J;
    p.M = a->rows; p.N = J; p.K = a->in_dim[i];
    p.epi = EPI_DROP_RELU;
    if (compact) { p.rowmap = a->rowmap; p.dyn = a->count; }
    p.x_bf16 = a->x_bf16 ? 1 : 0;
    if (p.x_bf16) p.A = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a->X) + 2 * (long)a->in_off[i]);
    set_dropout(p, &a->drop, a->drop.site, i * J);
    g1.p[i] = p;

    GemmProblem q = make_problem();
    q.A = (pooled ? a->Hbar : a->H1) + (long)i * J; q.lda = (long)nseg * J;
    q.B = a->W2[i]; q.ldb = J;
    q.bias = a->b2[i];
    q.rowscale = pooled ? a->fscale : nullptr;
    q.C = a->Z2 + ooff; q.ldc = a->ldz2;
    q.M = n2; q.N = a->out_dim[i]; q.K = J;
    if (a->epilogue == 1) {
      q.epi = EPI_TANH_DROP;
      q.aux_out = a->Tn + ooff; q.ldaux = a->ldtn;
      set_dropout(q, &a->drop, a->drop.site2, ooff);
    } else {
      q.epi = EPI_STORE;
    }
    g2.p[i] = q;
    ooff += a->out_dim[i];
  }
  return LIREC_OK;
}

// may the streaming pool / un-pool kernels (one wave per (candidate, 256-column block), float4 accesses) be used?
static inline bool pool_rows_ok(int R, int W, long ld_a, long ld_b, long ld_c, const void* a, const void* b, const void* c) {
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
  return R <= 64 && (W & 3) == 0 && ((ld_a | ld_b | ld_c) & 3) == 0 && (al & 15) == 0;
}
static inline unsigned pool_rows_grid(int n, int W) {
  const long tasks = (long)n * ((W + 255) / 256);
  long blocks = (tasks + 3) / 4;
  if (blocks > 2048) blocks = 2048;                  // 8 workgroups of 4 waves per CU, grid-stride beyond that (1024 / 4096 measured slower)
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

// (pooled form) the masked mean of H1 over each candidate's context rows
static int embed_fwd_pool_only(const lirec_embed_fwd_args* a, hipStream_t s) {
  const bool pooled = a->mask != nullptr || a->rowmap != nullptr, compact = a->rowmap != nullptr;
  if (!pooled) return LIREC_OK;
  const int J = a->J, nseg = a->nseg, n2 = a->rows / a->R, W = nseg * J;
  const long ldh = (long)W;
  // algorithmic bytes of the pass, priced on the STATIC row count n*R (the library never reads the device-side
  // count back); a caller that knows how many rows are valid scales the row term (bench.py does)
  const double bytes = 4.0 * n2 * ((double)a->R * W + a->R + (double)W);
  if (pool_rows_ok(a->R, W, ldh, ldh, ldh, a->H1, a->Hbar, a->Hbar)) {
    const int pi = prof_start(PS_POOL_FWD, s);
    unsigned char* hb = reinterpret_cast<unsigned char*>(a->hbits);
#define LIREC_POOL(COMPACT, ...)                                                                                       \
  do {                                                                                                                 \
    if (hb) lirec::launch(HIP_KERNEL_NAME(pool_rows_kernel<COMPACT, true>), __VA_ARGS__);                              \
    else lirec::launch(HIP_KERNEL_NAME(pool_rows_kernel<COMPACT, false>), __VA_ARGS__);                                \
  } while (0)
    if (compact)
      LIREC_POOL(true, dim3(pool_rows_grid(n2, W)), dim3(256), 0, s, (const float*)a->H1, ldh,
                 a->mask, a->rowmap, a->cstart, a->wts, n2, a->R, W, a->clamp_zero, a->Hbar, ldh, a->fscale, hb);
    else
      LIREC_POOL(false, dim3(pool_rows_grid(n2, W)), dim3(256), 0, s, (const float*)a->H1, ldh,
                 a->mask, (const int*)nullptr, (const int*)nullptr, (const float*)nullptr, n2, a->R, W, a->clamp_zero,
                 a->Hbar, ldh, a->fscale, hb);
#undef LIREC_POOL
    prof_stop(pi, s, 0.0, bytes + (hb ? (double)n2 * a->R * W / 8.0 : 0.0));
    LIREC_CHECK_LAUNCH();
    return LIREC_OK;
  }
  if (a->hbits) return LIREC_EINVAL;           // (only the streaming kernel writes the sign bits)
  if (compact) {
    const int pi = prof_start(PS_POOL_FWD, s);
    lirec::launch(pool_compact_kernel, dim3(n2), dim3(256), 0, s, (const float*)a->H1, ldh, a->mask,
                       a->rowmap, a->cstart, a->wts, a->R, W, a->clamp_zero, a->Hbar, ldh, a->fscale);
    prof_stop(pi, s, 0.0, bytes);
    LIREC_CHECK_LAUNCH();
    return LIREC_OK;
  }
  return launch_pool(a->H1, ldh, a->mask, n2, a->R, W, a->clamp_zero, a->Hbar, ldh, nullptr, 0, nullptr, 1, a->fscale, s);
}

// concatenation of two problem groups when they fit one launch
static bool merge_groups(const GemmGroup& x, const GemmGroup& y, GemmGroup& out) {
  if (x.nprob + y.nprob > LIREC_MAX_PROB) return false;
  out.nprob = x.nprob + y.nprob;
  for (int i = 0; i < x.nprob; ++i) out.p[i] = x.p[i];
  for (int i = 0; i < y.nprob; ++i) out.p[x.nprob + i] = y.p[i];
  return true;
}

// Layer 1 of one or two heads (one grouped launch when they fit), then the pooling pass of the pooled heads.
// stage_mode: 0 = stage everything, then the GEMM; 1 = stage the ROWS only (feature rows / row lists, dropout keep bytes, the
// partition bound) and return -- parts = 4, what a caller runs for the NEXT batch beside this batch's backward; 2 = the rows are
// in `planes` already (such a call was made): stage the weights only, then the GEMM.
static int embed_fwd_layer1_heads(const lirec_embed_fwd_args* const* hs, GemmGroup* g1, int nh, hipStream_t s, int stage_mode = 0) {
  PlaneLayout L[2];
  int rc = LIREC_OK;
  bool planes = planes_for_heads(hs, nh, L);
  if (stage_mode != 0 && !planes) return LIREC_EINVAL;          // (rows staged ahead: the q32b kernels only)
  for (int h = 0; h < nh; ++h)
    if ((hs[h]->pieces || hs[h]->x_q32) && !planes) return LIREC_EINVAL;   // rows given as pieces / stored as q32b: the q32b kernels only
  SplitQ32b q;
  memset(&q, 0, sizeof(q));
  q.fmt16c = g_gemm_mode == 3 ? 1 : 0;       // (single-pass mode: the weights as q16c, like the rows)
  for (int h = 0; planes && h < nh; ++h) {
    // first-layer weights the caller keeps in the q32b form (single-pass mode: q16c) (lirec_embed_fwd_args::W1q): nothing to stage
    int given = 0;
    for (int i = 0; i < hs[h]->nseg; ++i) given += hs[h]->W1q[i] != nullptr;
    if (given != 0 && given != hs[h]->nseg) return LIREC_EINVAL;
    for (int i = 0; planes && i < hs[h]->nseg; ++i) {
      if (given) {
        if ((reinterpret_cast<uintptr_t>(hs[h]->W1q[i]) & 255) != 0) return LIREC_EINVAL;
        L[h].wq[i] = reinterpret_cast<unsigned char*>(const_cast<void*>(hs[h]->W1q[i]));
      } else {
        planes = splitq_add(q, hs[h]->W1[i], L[h].wq[i], hs[h]->J, hs[h]->in_dim[i]);
      }
    }
  }
  for (int h = 0; !planes && h < nh; ++h)
    if (hs[h]->W1q[0]) return LIREC_EINVAL;                       // (the q32b kernels only)
  if (planes) {
    // operands in the q32b form: weights (both heads, one launch), feature rows (one launch per head), then every segment of
    // every head in ONE persistent launch
    GemmGroup m;
    m.nprob = 0;
    for (int h = 0; !rc && h < nh; ++h) {
      for (int i = 0; i < hs[h]->nseg; ++i) {
        GemmProblem p = g1[h].p[i];
        if (hs[h]->drop.p > 0.f) { p.aux = reinterpret_cast<const float*>(L[h].keep); p.ldaux = (long)hs[h]->nseg * hs[h]->J; }
        if (L[h].gather) {
          gather_operand(hs[h], L[h], i, p.A, p.lda, p.srow);
        } else {
          p.A = reinterpret_cast<const float*>(L[h].xq + 4096L * ((hs[h]->in_off[i] - L[h].c0) / 32));
          p.lda = L[h].dsum;
        }
        p.B = reinterpret_cast<const float*>(L[h].wq[i]); p.ldb = hs[h]->in_dim[i];
        if (g_gemm_mode == 3) {
          // q16c operands: byte for byte a q32b problem of half the columns (64 of k per step: gemm_p2_ntg64_kernel)
          p.K >>= 1; p.lda >>= 1; p.ldb >>= 1;
        }
        p.gs = 0; p.gs_magic = 0; p.x_bf16 = 0;             // rows are dense now; rowmap / dyn stay (dropout ids, M bound)
        m.p[m.nprob++] = p;
      }
    }
    // (one launch: the larger head's rows first, then the other head's, then the weights, then the one workgroup that runs
    //  the GEMM's partition search on the device-side row counts -- stage_fused_kernel)
    {
      StageFused f;
      memset(&f, 0, sizeof(f));
      const int first = (nh == 2 && hs[1]->rows > hs[0]->rows) ? 1 : 0;
      double bytes = 0.0;
      for (int k = 0; k < nh && stage_mode != 2; ++k) {
        const int h = (k == 0) ? first : 1 - first;
        stage_head_fill(f.h[k], hs[h], L[h]);
        if (L[h].staged16) bytes += 4.0 * (double)hs[h]->rows * L[h].dsum;
        else if (!L[h].gather) bytes += 8.0 * (double)hs[h]->rows * L[h].dsum;
      }
      f.nh = stage_mode == 2 ? 0 : nh;
      if (stage_mode != 1) {
        f.w = q;
        long wb = (q.first[q.nseg] + 255) / 256;
        f.w_blocks = (int)(wb > 2048 ? 2048 : wb);
        bytes += 64.0 * (double)q.first[q.nseg];
      }
      // the problems exactly as launch_p2 will hand them to the GEMM (empty ones dropped, same order)
      f.part.grid = p2_grid(); f.part.out = stage_mode == 2 ? nullptr : L[0].nt_bound;
      for (int i = 0; i < m.nprob; ++i)
        if (m.p[i].M > 0 && m.p[i].N > 0) {
          const int k = f.part.n++;
          f.part.ks[k] = m.p[i].K >> 5; f.part.rows[k] = m.p[i].M; f.part.dyn[k] = m.p[i].dyn; f.part.nrep = m.p[i].N / 256;
        }
      long grid = f.w_blocks + 1;
      for (int k = 0; k < f.nh; ++k) grid += f.h[k].blocks;
      if (f.nh > 0 || f.w_blocks > 0 || f.part.out) {           // (rows staged ahead and weights kept as q32b: nothing to do)
        // (diagnostics bit 262144: a staging pass run AHEAD of its step -- stage_mode 1, the input pipeline -- starts ~2 ms late)
        if ((g_ablate & 262144) && stage_mode == 1) lirec::launch(spin_kernel, dim3(1), dim3(64), 0, s, 200000LL);
        const int pi = prof_start(PS_STAGE, s);
        lirec::launch(stage_fused_kernel, dim3((unsigned)grid), dim3(256), 0, s, f);
        prof_stop(pi, s, 0.0, bytes);
        LIREC_CHECK_LAUNCH();
      }
    }
    if (stage_mode == 1) return rc;
    if (!rc) rc = launch_p2<L_NT>(m, s, PS_EMBED_L1_FWD, L[0].nt_bound, 0, L[0].gather, nullptr, L[0].gather ? gather_planes(hs, nh) : 2);
  } else {
    GemmGroup m;
    if (nh == 2 && merge_groups(g1[0], g1[1], m)) {
      // both heads in one launch (two tiers of row panels when the heads differ in rows)
      rc = launch_gemm(L_NT, m, s, PS_EMBED_L1_FWD, 1);
    } else {
      for (int h = 0; !rc && h < nh; ++h) rc = launch_gemm(L_NT, g1[h], s, PS_EMBED_L1_FWD, 1);
    }
  }
  for (int h = 0; !rc && h < nh; ++h) rc = embed_fwd_pool_only(hs[h], s);
  return rc;
}

int lirec_embed_fwd(const lirec_embed_fwd_args* a, lirec_stream_t stream) {
  GemmGroup g1, g2;
  int rc = embed_fwd_build(a, g1, g2);
  if (rc) return rc;
  if (a->rows == 0) return LIREC_OK;
  if (a->parts < 0 || a->parts > 4) return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->parts == 4) return embed_fwd_layer1_heads(&a, &g1, 1, s, 1);
  if (a->parts == 3) rc = embed_fwd_pool_only(a, s);
  else if (a->parts != 2) rc = embed_fwd_layer1_heads(&a, &g1, 1, s, a->rows_staged ? 2 : 0);
  if (rc || a->parts == 1) return rc;
  return launch_gemm(L_NT, g2, s, PS_EMBED_L2_FWD);
}

int lirec_embed_fwd2(const lirec_embed_fwd_args* a, const lirec_embed_fwd_args* b, lirec_stream_t stream) {
  GemmGroup g1[2], a2, b2, m2;
  int rc = embed_fwd_build(a, g1[0], a2);
  if (rc) return rc;
  rc = embed_fwd_build(b, g1[1], b2);
  if (rc) return rc;
  if (a->rows == 0 || b->rows == 0) {
    rc = lirec_embed_fwd(a, stream);
    return rc ? rc : lirec_embed_fwd(b, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  if (a->parts < 0 || a->parts > 4 || b->parts != a->parts || (a->rows_staged != 0) != (b->rows_staged != 0)) return LIREC_EINVAL;
  const lirec_embed_fwd_args* hs[2] = {a, b};
  if (a->parts == 4) return embed_fwd_layer1_heads(hs, g1, 2, s, 1);
  if (a->parts == 3) { rc = embed_fwd_pool_only(a, s); if (!rc) rc = embed_fwd_pool_only(b, s); }
  else if (a->parts != 2) rc = embed_fwd_layer1_heads(hs, g1, 2, s, a->rows_staged ? 2 : 0);
  if (rc || a->parts == 1) return rc;
  // the second layers of both heads run on the (pooled) candidate rows: one grouped launch
  if (merge_groups(a2, b2, m2)) return launch_gemm(L_NT, m2, s, PS_EMBED_L2_FWD);
  rc = launch_gemm(L_NT, a2, s, PS_EMBED_L2_FWD);
  return rc ? rc : launch_gemm(L_NT, b2, s, PS_EMBED_L2_FWD);
}

// Layer 1 of `nh` heads on the unique feature pieces (kernels.hpp, gather_act_kernel): the pre-activations of every piece
// (table GEMMs, all heads in one grouped launch, K never split: the sums must equal the dense launch's bit for bit), then
// one expansion pass per head; H1 is what lirec_embed_fwd's layer 1 leaves.  The caller goes on with parts = 3.
int lirec_embed_l1_indexed(const lirec_embed_fwd_args* const* heads, int32_t nh, const lirec_pieces* pc,
                           float* const* zclip, float* const* ztrk, lirec_stream_t stream) {
  if (!heads || !pc || !zclip || !ztrk || nh < 1 || nh > 2 || !pc->clip || !pc->track || !pc->index || pc->n_clip < 1 || pc->n_track < 1)
    return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  for (int h = 0; h < nh; ++h) {
    const lirec_embed_fwd_args* a = heads[h];
    if (!a || !a->H1 || a->nseg != 4 || a->J < 256 || (a->J & 255) || a->rows < 0 || !zclip[h] || !ztrk[h] || a->x_bf16) return LIREC_EINVAL;
    if (a->in_dim[0] != pc->text_dim || a->in_dim[1] != pc->visual_dim || a->in_dim[2] != pc->track_dim || a->in_dim[3] != pc->track_dim)
      return LIREC_EINVAL;
    const int J = a->J;
    for (int i = 0; i < 4; ++i) {
      if (!a->W1[i] || !a->b1[i]) return LIREC_EINVAL;
      GemmProblem p = make_problem();
      if (i < 2) { p.A = pc->clip + (i == 0 ? 0 : pc->text_dim); p.lda = pc->ld_clip; p.M = pc->n_clip; p.C = zclip[h] + (long)i * J; }
      else { p.A = pc->track; p.lda = pc->ld_track; p.M = pc->n_track; p.C = ztrk[h] + (long)(i - 2) * J; }
      p.B = a->W1[i]; p.ldb = a->in_dim[i];
      p.ldc = 2L * J; p.N = J; p.K = a->in_dim[i];
      p.epi = EPI_STORE;
      g.p[g.nprob++] = p;
    }
  }
  t_no_split = true;
  int rc = launch_gemm(L_NT, g, s, PS_EMBED_L1_FWD);
  t_no_split = false;
  for (int h = 0; !rc && h < nh; ++h) {
    const lirec_embed_fwd_args* a = heads[h];
    if (a->rows == 0) continue;
    const bool compact = a->rowmap != nullptr;
    GatherActArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.zclip = zclip[h]; ga.ld_zclip = 2L * a->J; ga.ztrk = ztrk[h]; ga.ld_ztrk = 2L * a->J;
    ga.index = pc->index;
    ga.gs = a->sel.group; ga.gstride = a->sel.group_stride; ga.goff = a->sel.group_off; ga.gs_magic = row_magic(ga.gs, a->rows);
    for (int i = 0; i < 4; ++i) ga.b1[i] = a->b1[i];
    ga.rowmap = compact ? a->rowmap : nullptr; ga.count = compact ? a->count : nullptr;
    ga.H1 = a->H1; ga.ldh = 4L * a->J; ga.rows = a->rows; ga.J = a->J;
    const float p = a->drop.p;
    ga.seed_lo = (unsigned)(a->drop.seed & 0xffffffffull); ga.seed_hi = (unsigned)(a->drop.seed >> 32);
    ga.seed_dev = (const unsigned long long*)a->drop.seed_dev; ga.site = (unsigned)a->drop.site;
    ga.thresh = drop_thresh(p); ga.scale = (p > 0.f) ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
    int gy = (a->rows + 3) / 4;
    if (gy > 256) gy = 256;
    const int pi = prof_start(PS_STAGE, s);
    lirec::launch(gather_act_kernel, dim3((unsigned)(4 * a->J / 256), (unsigned)gy), dim3(256), 0, s, ga);
    prof_stop(pi, s, 0.0, 4.0 * (double)a->rows * 4 * a->J);
    LIREC_CHECK_LAUNCH();
  }
  return rc;
}

// argument checks + the three GEMM groups of one head's backward (dW2, dZ1 / dHbar, dW1)
static int embed_bwd_build(const lirec_embed_bwd_args* a, GemmGroup& gw2, GemmGroup& gdz, GemmGroup& gw1) {
  if (!a || (!a->X && a->parts != 1 && a->parts != 3 && a->parts != 5 && !a->planes) || (!a->H1 && !a->hbits) || !a->dZ2 || a->nseg < 1 || a->nseg > LIREC_MAX_SEG || a->J < 1 || a->rows < 0)
    return LIREC_EINVAL;
  const bool pooled = a->mask != nullptr || a->rowmap != nullptr;
  if (a->hbits && !pooled) return LIREC_EINVAL;
  if (pooled && (!a->Hbar || !a->fscale || a->R < 1 || a->rows % a->R != 0)) return LIREC_EINVAL;
  const bool compact = pooled && a->rowmap != nullptr;
  if ((a->rowmap || a->cstart || a->count) && !(a->rowmap && a->cstart && a->count && (a->mask || a->wts))) return LIREC_EINVAL;
  if (a->wts && !compact) return LIREC_EINVAL;
  const int J = a->J, nseg = a->nseg;
  const int n2 = pooled ? a->rows / a->R : a->rows;
  if (!a->workspace || a->workspace_bytes < lirec_workspace_bytes(a->rows + (pooled ? n2 : 0), nseg, J)) return LIREC_EWORKSPACE;
  float* dZ1 = (float*)a->workspace;                           // [rows, nseg*J] (or its bf16 planes, see embed_bwd_unpool)
  float* dHbar = dZ1 + (long)((a->rows + 31) / 32 * 32) * nseg * J;   // pooled form: [n, nseg*J], behind the 32-row padding
  const long ldh = (long)nseg * J;
  const float scale = (a->drop.p > 0.f) ? (float)(1.0 / (1.0 - (double)a->drop.p)) : 1.f;
  gw2.nprob = gdz.nprob = gw1.nprob = nseg;
  int ooff = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!a->W2[i] || !a->dW1[i] || !a->dW2[i] || !a->db1[i] || !a->db2[i]) return LIREC_EINVAL;
    // dW2_i [out, J] += dZ2_i^T H_i ; db2_i += colsum (f .) dZ2_i       (H = H1, or Hbar when pooled)
    GemmProblem p = make_problem();
    p.A = a->dZ2 + ooff; p.lda = a->lddz2;
    p.B = (pooled ? a->Hbar : a->H1) + (long)i * J; p.ldb = ldh;
    p.C = a->dW2[i]; p.ldc = J;
    p.M = a->out_dim[i]; p.N = J; p.K = n2;
    p.beta = grad_beta(); p.dbias_set = g_grad_overwrite; p.dbias = a->db2[i];
    p.rowscale = pooled ? a->fscale : nullptr;
    gw2.p[i] = p;
    // plain:  dZ1_i [rows, J] = (dZ2_i W2_i) * [H1_i > 0] / (1-p)
    // pooled: dHbar_i [n, J]  =  dZ2_i W2_i          (un-pooled and masked by unpool_relu_kernel)
    GemmProblem q = make_problem();
    q.A = a->dZ2 + ooff; q.lda = a->lddz2;
    q.B = a->W2[i]; q.ldb = J;
    q.C = (pooled ? dHbar : dZ1) + (long)i * J; q.ldc = ldh;
    q.M = n2; q.N = J; q.K = a->out_dim[i];
    if (pooled) {
      q.epi = EPI_STORE;
    } else {
      q.epi = EPI_RELU_BWD; q.aux = a->H1 + (long)i * J; q.ldaux = ldh; q.drop_scale = scale;
    }
    gdz.p[i] = q;
    // dW1_i [J, in] += dZ1_i^T X_i ; db1_i += colsum dZ1_i
    GemmProblem w = make_problem();
    w.A = dZ1 + (long)i * J; w.lda = ldh;
    w.B = a->X + a->in_off[i]; w.ldb = a->ldx;
    w.gs = a->sel.group; w.gstride = a->sel.group_stride; w.goff = a->sel.group_off;
    w.gs_magic = row_magic(w.gs, a->rows);
    w.C = a->dW1[i]; w.ldc = a->in_dim[i];
    w.M = J; w.N = a->in_dim[i]; w.K = a->rows;
    w.beta = grad_beta(); w.dbias_set = g_grad_overwrite; w.dbias = a->db1[i];
    if (compact) { w.rowmap = a->rowmap; w.dyn = a->count; }
    w.x_bf16 = a->x_bf16 ? 1 : 0;
    if (w.x_bf16) w.B = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a->X) + 2 * (long)a->in_off[i]);
    gw1.p[i] = w;
    ooff += a->out_dim[i];
  }
  return LIREC_OK;
}

// (pooled form) dZ1 = un-pooled dHbar with the relu/dropout factor.  `planes`: written as bf16 hi / lo planes
// ([rows32, nseg*J] each, hi first) over the same workspace bytes, for the weight gradient on planes.
static int embed_bwd_unpool(const lirec_embed_bwd_args* a, hipStream_t s, int planes, const SplitSegs* sq = nullptr,
                            const SplitQ32b* sq32 = nullptr) {
  const bool pooled = a->mask != nullptr || a->rowmap != nullptr, compact = a->rowmap != nullptr;
  if (!pooled) return LIREC_OK;
  const int J = a->J, nseg = a->nseg, n2 = a->rows / a->R;
  const long rows32 = (a->rows + 31) / 32 * 32;
  float* dZ1 = (float*)a->workspace;
  float* dHbar = dZ1 + rows32 * nseg * J;
  const long ldh = (long)nseg * J;
  const float scale = (a->drop.p > 0.f) ? (float)(1.0 / (1.0 - (double)a->drop.p)) : 1.f;
  const int W = nseg * J;
  const int pi = prof_start(PS_POOL_BWD, s);
  // (`hbits`: the sign bits of H1 left by the forward pooling pass stand in for H1 -- streaming kernels only)
  const bool bits = a->hbits != nullptr;
  const float* h1 = bits ? reinterpret_cast<const float*>(a->hbits) : a->H1;
  if (bits && (a->R > 64 || (W & 3) != 0)) return LIREC_EINVAL;
#define LIREC_UNPOOL(COMPACT, PLANES, ...)                                                                                        \
  do {                                                                                                                            \
    if (bits) lirec::launch(HIP_KERNEL_NAME(unpool_rows_kernel<COMPACT, PLANES, true>), __VA_ARGS__);                             \
    else lirec::launch(HIP_KERNEL_NAME(unpool_rows_kernel<COMPACT, PLANES, false>), __VA_ARGS__);                                 \
  } while (0)
  SplitQ32b q32z;
  memset(&q32z, 0, sizeof(q32z));
  if (planes == 2) {
    // dZ1 as q32b rows [rows32][W] over the workspace (gemm_p3's weight gradient reads them k-major)
    const SplitQ32b& q = (sq32 && sq32->nseg > 0) ? *sq32 : q32z;
    long sb = (q.first[q.nseg] + 255) / 256;
    if (sb > 512) sb = 512;
    const unsigned grid = pool_rows_grid(n2, W) + (unsigned)sb;
    SplitSegs q0;
    memset(&q0, 0, sizeof(q0));
    if (compact)
      LIREC_UNPOOL(true, 2, dim3(grid), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, a->rowmap, a->cstart, a->wts, n2, a->R, W,
                   a->clamp_zero, scale, dZ1, ldh, 0L, a->count, q0, (int)sb, q);
    else
      LIREC_UNPOOL(false, 2, dim3(grid), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, (const int*)nullptr, (const int*)nullptr,
                   (const float*)nullptr, n2, a->R, W, a->clamp_zero, scale, dZ1, ldh, 0L, (const int*)nullptr, q0, (int)sb, q);
  } else if (planes) {
    // (plane_layout guarantees the alignment the streaming kernel needs: J % 128 == 0)
    // (gemm mode 3: the weight gradient's single-pass form multiplies dZ1's hi halves only -- the lo plane is not written)
    const long lo_off = g_gemm_mode == 3 ? 0L : rows32 * ldh;
    // (`sq`: another head's fp32 dZ1 to be split into planes by the first workgroups of this launch)
    SplitSegs q0;
    memset(&q0, 0, sizeof(q0));
    const SplitSegs& q = (sq && sq->nseg > 0) ? *sq : q0;
    long sb = (q.first[q.nseg] + 255) / 256;
    if (sb > 512) sb = 512;
    const unsigned grid = pool_rows_grid(n2, W) + (unsigned)sb;
    if (compact)
      LIREC_UNPOOL(true, 1, dim3(grid), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, a->rowmap, a->cstart, a->wts, n2, a->R, W,
                   a->clamp_zero, scale, dZ1, ldh, lo_off, a->count, q, (int)sb, q32z);
    else
      LIREC_UNPOOL(false, 1, dim3(grid), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, (const int*)nullptr, (const int*)nullptr,
                   (const float*)nullptr, n2, a->R, W, a->clamp_zero, scale, dZ1, ldh, lo_off, (const int*)nullptr, q, (int)sb, q32z);
  } else if (pool_rows_ok(a->R, W, ldh, ldh, ldh, dHbar, bits ? (const void*)dHbar : (const void*)a->H1, dZ1)) {
    if (compact)
      LIREC_UNPOOL(true, 0, dim3(pool_rows_grid(n2, W)), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, a->rowmap, a->cstart, a->wts, n2, a->R, W,
                   a->clamp_zero, scale, dZ1, ldh, 0L, (const int*)nullptr, SplitSegs(), 0, q32z);
    else
      LIREC_UNPOOL(false, 0, dim3(pool_rows_grid(n2, W)), dim3(256), 0, s,
                   (const float*)dHbar, ldh, h1, ldh, a->mask, (const int*)nullptr, (const int*)nullptr,
                   (const float*)nullptr, n2, a->R, W, a->clamp_zero, scale, dZ1, ldh, 0L, (const int*)nullptr, SplitSegs(), 0, q32z);
  } else if (bits) {
    return LIREC_EINVAL;
  } else if (compact) {
    lirec::launch(unpool_relu_compact_kernel, dim3(n2), dim3(256), 0, s, (const float*)dHbar, ldh, a->H1, ldh,
                       a->mask, a->rowmap, a->cstart, a->wts, W, a->clamp_zero, scale, dZ1, ldh);
  } else {
    lirec::launch(unpool_relu_kernel, dim3(n2), dim3(256), 0, s, (const float*)dHbar, ldh, a->H1, ldh, a->mask,
                       a->R, W, a->clamp_zero, scale, dZ1, ldh);
  }
#undef LIREC_UNPOOL
  // static row count n*R, as for the forward pass (H1 -- or its sign bits -- read + dZ1 written per row, dHbar read per candidate)
  prof_stop(pi, s, 0.0, 4.0 * n2 * ((bits ? 1.0 + 1.0 / 32.0 : 2.0) * a->R * W + a->R + (double)W));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// The rest of the backward of one or two heads once dZ1 / dHbar exist: un-pooling (pooled heads) and the first-layer
// weight gradient -- on planes when the forward ran on planes (the same `planes` buffer), else on the fly.
static int embed_bwd_tail_heads(const lirec_embed_bwd_args* const* hs, GemmGroup* gw1, int nh, hipStream_t s) {
  PlaneLayout L[2];
  const bool planes = planes_for_heads(hs, nh, L);
  int rc = LIREC_OK;
  for (int h = 1; h < nh; ++h) if (hs[h]->adam && hs[h]->adam != hs[0]->adam) return LIREC_EINVAL;
  if (hs[0]->adam && (!planes || (g_ablate & 8192))) return LIREC_EINVAL;      // the fused update: the gemm_p2 reduce kernel only
  if (!planes) {
    for (int h = 0; h < nh; ++h) if (!hs[h]->X || hs[h]->x_q32) return LIREC_EINVAL;   // (no fp32 block and no staged rows: nothing to reduce over)
    for (int h = 0; !rc && h < nh; ++h) rc = embed_bwd_unpool(hs[h], s, 0);
    for (int h = 0; !rc && h < nh; ++h) rc = launch_gemm(L_TN, gw1[h], s, PS_EMBED_DW1, 2);
    return rc;
  }
  GemmGroup m;
  m.nprob = 0;
  SplitSegs q;
  memset(&q, 0, sizeof(q));
  // plain heads first: their fp32 dZ1 (left by the data-gradient GEMM) is split into planes -- by the first workgroups of a
  // pooled head's un-pool launch when there is one, else by a launch of its own
  for (int h = 0; !rc && h < nh; ++h) {
    const lirec_embed_bwd_args* a = hs[h];
    if (a->mask != nullptr || a->rowmap != nullptr) continue;
    const long ldh = (long)a->nseg * a->J, rows32 = (a->rows + 31) / 32 * 32;
    float* dZ1 = reinterpret_cast<float*>(a->workspace);
    unsigned short* zh = reinterpret_cast<unsigned short*>(dZ1 + rows32 * ldh);
    if (!split_add(q, dZ1, zh, zh + rows32 * ldh, (long)a->rows * ldh)) return LIREC_EINVAL;
    if (rows32 > a->rows) {                                     // the k-tail of the weight gradient must be zero
      (void)lirec::memset_async(zh + (long)a->rows * ldh, 0, (size_t)(rows32 - a->rows) * ldh * 2, s);
      (void)lirec::memset_async(zh + rows32 * ldh + (long)a->rows * ldh, 0, (size_t)(rows32 - a->rows) * ldh * 2, s);
    }
  }
  bool split_done = q.nseg == 0;
  for (int h = 0; !rc && h < nh; ++h) {
    const lirec_embed_bwd_args* a = hs[h];
    const bool pooled = a->mask != nullptr || a->rowmap != nullptr;
    const long ldh = (long)a->nseg * a->J, rows32 = (a->rows + 31) / 32 * 32;
    unsigned short *zh, *zl;
    if (pooled) {
      rc = embed_bwd_unpool(a, s, 1, split_done ? nullptr : &q);
      split_done = true;
      zh = reinterpret_cast<unsigned short*>(a->workspace);
    } else {
      zh = reinterpret_cast<unsigned short*>(reinterpret_cast<float*>(a->workspace) + rows32 * ldh);
    }
    zl = zh + rows32 * ldh;
    for (int i = 0; i < a->nseg; ++i) {
      GemmProblem w = gw1[h].p[i];
      w.A = reinterpret_cast<const float*>(zh + (long)i * a->J); w.A_lo = zl + (long)i * a->J; w.lda = ldh;
      if (L[h].gather) {
        gather_operand(a, L[h], i, w.B, w.ldb, w.srow);
      } else {
        w.B = reinterpret_cast<const float*>(L[h].xq + 4096L * ((a->in_off[i] - L[h].c0) / 32)); w.ldb = L[h].dsum;
      }
      w.gs = 0; w.gs_magic = 0; w.rowmap = nullptr; w.x_bf16 = 0;       // dense q32b rows; `dyn` still bounds K
      w.K = (int)rows32 < w.K ? (int)rows32 : w.K;
      m.p[m.nprob++] = w;
    }
  }
  if (!rc && !split_done) rc = launch_split(q, s);
  if (!rc) rc = launch_p2<L_TN>(m, s, PS_EMBED_DW1, nullptr, 0, L[0].gather, hs[0]->adam, L[0].gather ? gather_planes(hs, nh) : 2);
  return rc;
}

int lirec_embed_bwd(const lirec_embed_bwd_args* a, lirec_stream_t stream) {
  GemmGroup gw2, gdz, gw1;
  int rc = embed_bwd_build(a, gw2, gdz, gw1);
  if (rc) return rc;
  if (a->rows == 0) return LIREC_OK;
  hipStream_t s = (hipStream_t)stream;
  const int parts = a->parts;
  if (parts < 0 || parts > 5) return LIREC_EINVAL;
  if (parts == 5) return embed_bwd_unpool(a, s, 0);       // the un-pooling pass of 4 alone (lirec_embed_dw1_indexed follows)
  if (parts == 0 || parts == 1) rc = launch_gemm(L_TN, gw2, s, PS_EMBED_DW2);
  if (rc || parts == 1) return rc;
  if (parts != 4) rc = launch_gemm(L_NN, gdz, s, PS_EMBED_DZ1);
  if (rc || parts == 3) return rc;
  return embed_bwd_tail_heads(&a, &gw1, 1, s);
}

// The first-layer weight gradients of `nh` heads from the unique pieces (kernels.hpp, onehot_kernel): dZ1 is where parts 3
// (+ 5 for the pooled form) left it, in the head's workspace.
int lirec_embed_dw1_indexed(const lirec_embed_bwd_args* const* heads, int32_t nh, const lirec_pieces* pc,
                            float* const* P, float* const* S, lirec_stream_t stream) {
  if (!heads || !pc || !P || !S || nh < 1 || nh > 2 || !pc->clip || !pc->track || !pc->index || pc->n_clip < 1 || pc->n_track < 1)
    return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nc1 = pc->n_clip + 1, nt1 = pc->n_track + 1;
  const long ldp = (long)(nc1 + 2 * nt1 + 3) / 4 * 4;
  GemmGroup gw;
  memset(&gw, 0, sizeof(gw));
  int rc = LIREC_OK;
  for (int h = 0; !rc && h < nh; ++h) {
    const lirec_embed_bwd_args* a = heads[h];
    if (!a || a->nseg != 4 || a->J < 1 || (a->J & 3) || a->rows < 0 || !a->workspace || !P[h] || !S[h] || a->x_bf16) return LIREC_EINVAL;
    if (a->in_dim[0] != pc->text_dim || a->in_dim[1] != pc->visual_dim || a->in_dim[2] != pc->track_dim || a->in_dim[3] != pc->track_dim)
      return LIREC_EINVAL;
    if (a->rows == 0) continue;
    const int J = a->J;
    const bool compact = a->rowmap != nullptr;
    const float* dZ1 = (const float*)a->workspace;
    const long ldh = 4L * J;
    // the incidence matrix of the index (zeroed, then one 1 per row and part)
    rc = (int)lirec::memset_async(P[h], 0, (size_t)a->rows * ldp * sizeof(float), s);
    if (rc) return rc;
    OneHotArgs oh;
    memset(&oh, 0, sizeof(oh));
    oh.index = pc->index; oh.gs = a->sel.group; oh.gstride = a->sel.group_stride; oh.goff = a->sel.group_off;
    oh.gs_magic = row_magic(oh.gs, a->rows);
    oh.rowmap = compact ? a->rowmap : nullptr; oh.count = compact ? a->count : nullptr;
    oh.P = P[h]; oh.ldp = ldp; oh.rows = a->rows; oh.n_clip = pc->n_clip; oh.n_track = pc->n_track;
    long blocks = (3L * a->rows + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    lirec::launch(onehot_kernel, dim3((unsigned)blocks), dim3(256), 0, s, oh);
    LIREC_CHECK_LAUNCH();
    // S = P^T dZ1: per piece, the sum of the dZ1 rows that use it.  S[h]: [nc1, 2J] (txt | vis) then [nt1, 2J] (tracks1 | tracks2)
    float* Sc = S[h];
    float* St = S[h] + (long)nc1 * 2 * J;
    GemmGroup gs;
    memset(&gs, 0, sizeof(gs));
    for (int part = 0; part < 3; ++part) {
      GemmProblem p = make_problem();
      p.A = P[h] + (part == 0 ? 0 : nc1 + (part - 1) * nt1); p.lda = ldp;
      p.B = dZ1 + (part == 0 ? 0 : (long)(part + 1) * J); p.ldb = ldh;
      p.C = part == 0 ? Sc : St + (long)(part - 1) * J; p.ldc = 2L * J;
      p.M = part == 0 ? nc1 : nt1; p.N = part == 0 ? 2 * J : J; p.K = a->rows;
      if (compact) p.dyn = a->count;
      p.epi = EPI_STORE;
      gs.p[gs.nprob++] = p;
    }
    rc = launch_gemm(L_TN, gs, s, PS_EMBED_DW1);
    // dW1_seg += S_seg^T pieces_seg ; db1_seg += column sums of S_seg   (all heads in one launch below)
    for (int i = 0; i < 4; ++i) {
      if (!a->dW1[i] || !a->db1[i]) return LIREC_EINVAL;
      GemmProblem w = make_problem();
      if (i < 2) { w.A = Sc + (long)i * J; w.B = pc->clip + (i == 0 ? 0 : pc->text_dim); w.ldb = pc->ld_clip; w.K = nc1; }
      else { w.A = St + (long)(i - 2) * J; w.B = pc->track; w.ldb = pc->ld_track; w.K = nt1; }
      w.lda = 2L * J;
      w.C = a->dW1[i]; w.ldc = a->in_dim[i];
      w.M = J; w.N = a->in_dim[i];
      w.beta = grad_beta(); w.dbias_set = g_grad_overwrite; w.dbias = a->db1[i];
      gw.p[gw.nprob++] = w;
    }
  }
  if (!rc && gw.nprob > 0) rc = launch_gemm(L_TN, gw, s, PS_EMBED_DW1);
  return rc;
}

int lirec_embed_bwd2(const lirec_embed_bwd_args* a, const lirec_embed_bwd_args* b, lirec_stream_t stream) {
  GemmGroup aw2, adz, bw2, bdz, m, gw1[2];
  int rc = embed_bwd_build(a, aw2, adz, gw1[0]);
  if (rc) return rc;
  rc = embed_bwd_build(b, bw2, bdz, gw1[1]);
  if (rc) return rc;
  if (a->rows == 0 || b->rows == 0) {
    rc = lirec_embed_bwd(a, stream);
    return rc ? rc : lirec_embed_bwd(b, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const int parts = a->parts;
  if (parts < 0 || parts > 4 || b->parts != parts) return LIREC_EINVAL;
  // second-layer weight gradients and the gradients w.r.t. the hidden layer: both heads in one launch each
  if (parts == 0 || parts == 1) {
    if (merge_groups(aw2, bw2, m)) {
      rc = launch_gemm(L_TN, m, s, PS_EMBED_DW2);
    } else {
      rc = launch_gemm(L_TN, aw2, s, PS_EMBED_DW2);
      if (!rc) rc = launch_gemm(L_TN, bw2, s, PS_EMBED_DW2);
    }
  }
  if (rc || parts == 1) return rc;
  if (parts != 4) {
    if (merge_groups(adz, bdz, m)) {
      rc = launch_gemm(L_NN, m, s, PS_EMBED_DZ1);
    } else {
      rc = launch_gemm(L_NN, adz, s, PS_EMBED_DZ1);
      if (!rc) rc = launch_gemm(L_NN, bdz, s, PS_EMBED_DZ1);
    }
  }
  if (rc || parts == 3) return rc;
  const lirec_embed_bwd_args* hs[2] = {a, b};
  return embed_bwd_tail_heads(hs, gw1, 2, s);
}

static int compact_rows_impl(const void* mask, int32_t mask_dtype, int32_t n, int32_t R, int32_t* rowmap, int32_t* cstart,
                             int32_t* count, float* wts, bool scratch_behind_cstart, lirec_stream_t stream) {
  if (!mask || !rowmap || !cstart || !count || n < 0 || R < 1 || mask_dtype < 0 || mask_dtype > 2) return LIREC_EINVAL;
  if (n == 0) {
    (void)lirec::memset_async(count, 0, sizeof(int32_t), (hipStream_t)stream);
    (void)lirec::memset_async(cstart, 0, sizeof(int32_t), (hipStream_t)stream);
    return LIREC_OK;
  }
  if (R <= 64 && scratch_behind_cstart) {
    // one wave per candidate, two launches; the per-candidate counts live behind cstart's n + 1 entries
    int* counts = cstart + n + 1;
    const unsigned blocks = (unsigned)((n + 3) / 4);
    lirec::launch(compact_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, (int)mask_dtype, n, R, counts);
    LIREC_CHECK_LAUNCH();
    lirec::launch(compact_place_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, (int)mask_dtype, n, R,
                       (const int*)counts, rowmap, cstart, count, wts);
    LIREC_CHECK_LAUNCH();
    return LIREC_OK;
  }
  const long entries = (long)n * R;
  // (R > 64) one workgroup; the mask as fp32 + one int per candidate in LDS when that fits the CU's 160 KiB
  const size_t lds = (size_t)entries * 4 + ((size_t)n + 1) * 4;
  const int use_lds = lds <= 150 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(compact_rows_serial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              150 * 1024);
    attr_set = true;
  }
  lirec::launch(compact_rows_serial_kernel, dim3(1), dim3(1024), use_lds ? lds : 0, (hipStream_t)stream, mask,
                     (int)mask_dtype, n, R, rowmap, cstart, count, wts, use_lds);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_compact_rows2(const void* mask, int32_t mask_dtype, int32_t n, int32_t R, int32_t* rowmap, int32_t* cstart,
                        int32_t* count, float* wts, lirec_stream_t stream) {
  return compact_rows_impl(mask, mask_dtype, n, R, rowmap, cstart, count, wts, true, stream);
}

int lirec_compact_rows(const float* mask, int32_t n, int32_t R, int32_t* rowmap, int32_t* cstart, int32_t* count,
                       lirec_stream_t stream) {
  return compact_rows_impl(mask, 0, n, R, rowmap, cstart, count, nullptr, false, stream);     // cstart: n + 1 entries only
}

// ---------------------------------------------------------------------------
int lirec_pool_fwd(const float* Z2, int64_t ldz, const float* mask, int32_t n, int32_t R, int32_t W,
                   int32_t clamp_zero, float* Tn, int64_t ldtn, float* E, int64_t lde,
                   const lirec_dropout* drop, lirec_stream_t stream) {
  if (!Z2 || !mask || !Tn || !E || n < 0 || R < 1 || W < 1) return LIREC_EINVAL;
  if (n == 0) return LIREC_OK;
  return launch_pool(Z2, (long)ldz, mask, n, R, W, clamp_zero, Tn, (long)ldtn, E, (long)lde, drop, 0, nullptr,
                     (hipStream_t)stream);
}

int lirec_pool_bwd(const float* dP, int64_t lddp, const float* mask, int32_t n, int32_t R, int32_t W,
                   int32_t clamp_zero, float* dZ2, int64_t lddz, lirec_stream_t stream) {
  if (!dP || !mask || !dZ2 || n < 0 || R < 1 || W < 1) return LIREC_EINVAL;
  if (n == 0) return LIREC_OK;
  const int pi = prof_start(PS_POOL_BWD, (hipStream_t)stream);
  lirec::launch(pool_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, dP, (long)lddp, mask, R, W,
                     clamp_zero, dZ2, (long)lddz);
  prof_stop(pi, (hipStream_t)stream, 0.0, 4.0 * n * ((double)R * W + R + W));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// ---------------------------------------------------------------------------
int lirec_gate_fwd(const float* EE, int64_t ldee, const float* Wg, const float* bg, int32_t n, int32_t K,
                   int32_t N, float* G, int64_t ldg, const lirec_dropout* drop, lirec_stream_t stream) {
  if (!EE || !Wg || !G || n < 0 || K < 1 || N < 1) return LIREC_EINVAL;
  GemmGroup g; g.nprob = 1;
  GemmProblem p = make_problem();
  p.A = EE; p.lda = ldee; p.B = Wg; p.ldb = K; p.bias = bg; p.C = G; p.ldc = ldg;
  p.M = n; p.N = N; p.K = K; p.epi = EPI_DROP_RELU;
  set_dropout(p, drop, drop ? drop->site : LIREC_SITE_GATE, 0);
  g.p[0] = p;
  return launch_gemm(L_NT, g, (hipStream_t)stream, PS_GATE_FWD);
}

int lirec_gate_bwd(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                   int32_t n, int32_t K, int32_t N, int32_t split,
                   const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                   int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                   lirec_stream_t stream) {
  return lirec_gate_bwd_parts(dZg, lddzg, EE, ldee, Wg, n, K, N, split, Tn, ldtn, dWg, dbg, dEE, lddee, acc_first, drop, site_ctx,
                              site_ints, 0, stream);
}

int lirec_gate_bwd_parts(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                         int32_t n, int32_t K, int32_t N, int32_t split,
                         const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                         int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                         int32_t parts, lirec_stream_t stream) {
  if (!dZg || !EE || !Wg || !Tn || !dWg || !dbg || !dEE || n < 0 || K < 1 || N < 1 || split < 0 || split > K || parts < 0 || parts > 2)
    return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int rc = LIREC_OK;
  if (parts != 2) {
    GemmGroup gw; gw.nprob = 1;
    GemmProblem w = make_problem();
    w.A = dZg; w.lda = lddzg; w.B = EE; w.ldb = ldee; w.C = dWg; w.ldc = K;
    w.M = N; w.N = K; w.K = n; w.beta = grad_beta(); w.dbias_set = g_grad_overwrite; w.dbias = dbg;
    gw.p[0] = w;
    rc = launch_gemm(L_TN, gw, s, PS_GATE_DW);
  }
  if (rc || parts == 1) return rc;
  // dEE = (dZg Wg) * tanh'/dropout factor, two column ranges with their own dropout streams
  GemmGroup gd; gd.nprob = 2;
  for (int h = 0; h < 2; ++h) {
    const int c0 = h == 0 ? 0 : split, nc = h == 0 ? split : K - split;
    GemmProblem p = make_problem();
    p.A = dZg; p.lda = lddzg;
    p.B = Wg + c0; p.ldb = K;
    p.C = dEE + c0; p.ldc = lddee;
    p.M = n; p.N = nc; p.K = N;
    p.epi = EPI_TANH_BWD; p.aux = Tn + c0; p.ldaux = ldtn;
    p.beta = (h == 0 && acc_first) ? 1.f : 0.f;
    set_dropout(p, drop, h == 0 ? site_ctx : site_ints, 0);
    gd.p[h] = p;
  }
  return launch_gemm(L_NN, gd, s, PS_GATE_DEE);
}

// ---- the gate GEMMs on staged q32b operands (gemm_p2.hpp) -----------------------------------------------------------------
// May the persistent q32b kernels serve the gate's forward (and, with the same staged Wg, its data gradient)?
static bool gate_p3_ok(int n, int K, int N, int split);
static bool gate_q32_ok(int n, int K, int N, int64_t ldee, const void* ws, int64_t ws_bytes) {
  // (default core; in the single-pass mode, gemm mode 3, only shapes the wave-specialised kernels take: their ONE forms, on operands
  //  staged as q16c -- nothing else reads that form)
  return (g_gemm_mode == 2 || (g_gemm_mode == 3 && gate_p3_ok(n, K, N, K / 2))) && !(g_ablate & 8) && ws != nullptr &&
         (reinterpret_cast<uintptr_t>(ws) & 255) == 0 && n >= 32 && (n & 31) == 0 &&
         (K & 255) == 0 && (N & 255) == 0 && ldee == K && ws_bytes >= lirec_gate_ws_bytes(n, K, N);
}
// wq, eq, zq: q32b forms of Wg [N][K], EE [n][K], dZg [n][N]; wqT, eqT, zqT: of their transposes (the operands of the data
// gradient -- Wg^T -- and of the weight gradient -- dZg^T, EE^T -- as k-contiguous rows: gemm_p3.hpp)
struct GateWs { unsigned char* wq; unsigned char* eq; unsigned char* zq; unsigned char* wqT; unsigned char* eqT; unsigned char* zqT; };
static GateWs gate_ws(void* ws, int n, int K, int N) {
  const long n32 = (n + 31) / 32 * 32;
  GateWs w;
  w.wq = reinterpret_cast<unsigned char*>(ws);
  w.eq = w.wq + align256(4L * N * K);
  w.zq = w.eq + align256(4L * n32 * K);
  w.wqT = w.zq + align256(4L * n32 * N);
  w.eqT = w.wqT + align256(4L * N * K);
  w.zqT = w.eqT + align256(4L * n32 * K);
  return w;
}
static bool dual_add(SplitDual& q, const float* src, unsigned char* dst, unsigned char* dstT, int rows, int cols) {
  if (q.nseg >= 4 || (rows & 31) || (cols & 31) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dstT)) & 15))
    return false;
  q.src[q.nseg] = src; q.dst[q.nseg] = dst; q.dstT[q.nseg] = dstT; q.rows[q.nseg] = rows; q.cols[q.nseg] = cols;
  q.first[q.nseg + 1] = q.first[q.nseg] + (long)(rows >> 5) * (cols >> 5);
  ++q.nseg;
  return true;
}
static int launch_gate_stage(SplitDual& q, hipStream_t s) {
  long blocks = q.first[q.nseg];
  if (blocks == 0) return LIREC_OK;
  q.fmt16c = g_gemm_mode == 3 ? 1 : 0;      // (single-pass mode: every gate operand as q16c -- gemm_p3.hpp's ONE form reads nothing else)
  double bytes = 0.0;                                           // read once, written once or twice
  for (int i = 0; i < q.nseg; ++i) bytes += 4.0 * q.rows[i] * (double)q.cols[i] * (q.fmt16c ? (q.dstT[i] ? 2.0 : 1.5) : (q.dstT[i] ? 3.0 : 2.0));
  if (blocks > 8192) blocks = 8192;
  const int pi = prof_start(PS_GATE_STAGE, s);
  lirec::launch(split_q32b_dual_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q);
  prof_stop(pi, s, 0.0, bytes);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}
// Shapes the wave-specialised kernel takes (gemm_p3.hpp, 128 x 96 tiles): forward 128 | n, 96 | N; data gradient 96 | split;
// weight gradient 128 | N, 96 | K with at least 2 MI = 8 column tiles (the bias gradient's fragments are dealt to them).
static bool gate_p3_ok(int n, int K, int N, int split) {
  // (single-pass mode: the kernel's k-step covers 64 of the reduced index -- K of the forward, N of the data gradient, n of the
  //  weight gradient)
  if (g_gemm_mode == 3 && ((n | K | N) & 63)) return false;
  return (n & 127) == 0 && (N & 127) == 0 && N % 96 == 0 && K % 96 == 0 && split % 96 == 0 && (K - split) % 96 == 0 && K / 96 >= 8 &&
         !(g_ablate & 16);
}
// 128 x 96 or 128 x 128 tiles for a gemm_p3 launch of M rows x N columns (all problems together)?  The one with fewer
// (rounds of tiles over the CUs) x (tile width): 1024 rows x 3072 -> 256 tiles of 96 columns, one round; 1280 rows (T = 20) -> 320
// such tiles would take two rounds where 240 tiles of 128 columns take one.
static int p3_pick_ni(int M, int N, bool n128_ok) {
  const long cus = p2_grid(), tm = M / 128;
  const long r3 = (tm * (N / 96) + cus - 1) / cus, r4 = (tm * (N / 128) + cus - 1) / cus;
  return (n128_ok && r4 * 4 < r3 * 3) ? 4 : 3;
}
// the tile space of a gemm_p3 launch and its XCD blocks: p3_xm x 8 / p3_xm blocks of (tm / p3_xm) x (tn / (8 / p3_xm)) tiles, the split
// that moves the fewest operand bytes into the XCDs' L2s (each block reads its row panels and its column panels once)
static unsigned p3_setup(GemmGroup& g, int BM, int BN, bool persistent = true) {
  int tn = 0;
  for (int i = 0; i < g.nprob; ++i) tn += g.p[i].N / BN;
  g.p3_tm = g.p[0].M / BM; g.p3_tn = tn; g.p3_xm = 0;
  double best = 0.0;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const int xn = 8 / xm;
    if (g.p3_tm % xm || tn % xn) continue;
    const double cost = (double)(g.p3_tm / xm) * BM + (double)(tn / xn) * BN;
    if (g.p3_xm == 0 || cost < best) { best = cost; g.p3_xm = xm; }
  }
  const long tiles = (long)g.p3_tm * tn;
  const int cus = p2_grid();
  // (persistent: one workgroup per CU walks its tiles, the loaders run ahead across tile boundaries; else one tile per workgroup:
  //  a launch that runs BESIDE the main chain then frees its CUs tile by tile instead of holding every CU to its end)
  const long grid = (tiles < cus || !persistent) ? tiles : cus;
  if (grid & 7) g.p3_xm = 0;                                    // (the block order deals workgroups to XCDs by blockIdx & 7)
  return (unsigned)grid;
}

int64_t lirec_gate_ws_bytes(int32_t n, int32_t K, int32_t N) {
  if (n < 0 || K < 0 || N < 0) return -1;
  const int64_t n32 = (n + 31) / 32 * 32;
  return 2 * (align256(4L * N * K) + align256(4L * n32 * K) + align256(4L * n32 * N));
}

int lirec_gate_stage_weights(const float* Wg, int32_t n, int32_t K, int32_t N, void* ws, int64_t ws_bytes, lirec_stream_t stream) {
  if (!Wg || n < 0 || K < 1 || N < 1) return LIREC_EINVAL;
  if (!gate_q32_ok(n, K, N, K, ws, ws_bytes) || (reinterpret_cast<uintptr_t>(Wg) & 15) != 0) return LIREC_EINVAL;
  const GateWs w = gate_ws(ws, n, K, N);
  SplitDual q;
  memset(&q, 0, sizeof(q));
  if (!dual_add(q, Wg, w.wq, w.wqT, N, K)) return LIREC_EINVAL;
  return launch_gate_stage(q, (hipStream_t)stream);
}

int lirec_gate_fwd_ws(const float* EE, int64_t ldee, const float* Wg, const float* bg, int32_t n, int32_t K,
                      int32_t N, float* G, int64_t ldg, const lirec_dropout* drop, void* ws, int64_t ws_bytes,
                      int32_t weights_staged, lirec_stream_t stream) {
  if (!EE || !Wg || !G || n < 0 || K < 1 || N < 1) return LIREC_EINVAL;
  if (!gate_q32_ok(n, K, N, ldee, ws, ws_bytes) || ((reinterpret_cast<uintptr_t>(EE) | reinterpret_cast<uintptr_t>(Wg)) & 15) != 0)
    return weights_staged ? LIREC_EINVAL : lirec_gate_fwd(EE, ldee, Wg, bg, n, K, N, G, ldg, drop, stream);
  hipStream_t s = (hipStream_t)stream;
  const GateWs w = gate_ws(ws, n, K, N);
  SplitDual q;
  memset(&q, 0, sizeof(q));
  if ((!weights_staged && !dual_add(q, Wg, w.wq, w.wqT, N, K)) || !dual_add(q, EE, w.eq, w.eqT, n, K)) return LIREC_EINVAL;
  int rc = launch_gate_stage(q, s);
  if (rc) return rc;
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.nprob = 1;
  GemmProblem p = make_problem();
  p.A = reinterpret_cast<const float*>(w.eq); p.lda = K;
  p.B = reinterpret_cast<const float*>(w.wq); p.ldb = K;
  p.bias = bg; p.C = G; p.ldc = ldg;
  p.M = n; p.N = N; p.K = K; p.epi = EPI_DROP_RELU;
  set_dropout(p, drop, drop ? drop->site : LIREC_SITE_GATE, 0);
  g.p[0] = p;
  if (gate_p3_ok(n, K, N, K / 2)) {
    // wave-specialised persistent kernel (gemm_p3.hpp), 128 x 96 tiles: 256 of them at the bench shape
    const int ni = p3_pick_ni(n, N, (N & 127) == 0);
    const unsigned grid = p3_setup(g, 128, 32 * ni);
    const int pi = prof_start(PS_GATE_FWD, s);
    g.onepass = g_gemm_mode == 3;
    launch_p3_fwd(ni, dim3(grid), s, g);
    prof_stop(pi, s, 2.0 * n * (double)N * K, 0.0);
    LIREC_CHECK_LAUNCH();
    return LIREC_OK;
  }
  return launch_p2<L_NT>(g, s, PS_GATE_FWD, nullptr, 1);
}

int lirec_gate_bwd_ws(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                      int32_t n, int32_t K, int32_t N, int32_t split,
                      const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                      int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                      int32_t parts, void* ws, int64_t ws_bytes, int32_t rows_staged, lirec_stream_t stream) {
  if (!dZg || !EE || !Wg || !Tn || !dWg || !dbg || !dEE || n < 0 || K < 1 || N < 1 || split < 0 || split > K || parts < 0 ||
      (parts > 2 && parts != 4))
    return LIREC_EINVAL;
  // (the two column ranges of dEE share one launch: they must have the same number of 128-column tiles)
  // (the wave-specialised kernel cuts the two column ranges into 96- or 128-column tiles; the p2 fallback into 256-column ones)
  const bool q32 = gate_q32_ok(n, K, N, ldee, ws, ws_bytes) && lddzg == N && 2 * split == K &&
                   (gate_p3_ok(n, K, N, split) || (split & 255) == 0) && (reinterpret_cast<uintptr_t>(dZg) & 15) == 0;
  if (!q32) {                                                   // (the plain kernels need no staged rows: the flag is moot)
    if (parts == 4) return LIREC_OK;
    return lirec_gate_bwd_parts(dZg, lddzg, EE, ldee, Wg, n, K, N, split, Tn, ldtn, dWg, dbg, dEE, lddee, acc_first, drop, site_ctx,
                                site_ints, parts, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  int rc = LIREC_OK;
  const GateWs w = gate_ws(ws, n, K, N);
  if (!rows_staged) {
    // dZg -> q32b rows (the A operand of the data gradient) and the q32b rows of its transpose (of the weight gradient)
    SplitDual q;
    memset(&q, 0, sizeof(q));
    if (!dual_add(q, dZg, w.zq, w.zqT, n, N)) return LIREC_EINVAL;
    rc = launch_gate_stage(q, s);
    if (rc || parts == 4) return rc;
  } else if (parts == 4) {
    return LIREC_OK;
  }
  const bool p3 = gate_p3_ok(n, K, N, split);
  if (parts != 2) {
    if (p3 && ldee == K) {
      // dWg = dZg^T EE (+ dbg = row sums of dZg^T) from the transposed rows of both (EE^T: staged by the forward call)
      GemmGroup gw;
      memset(&gw, 0, sizeof(gw));
      gw.nprob = 1;
      GemmProblem p = make_problem();
      p.A = reinterpret_cast<const float*>(w.zqT); p.lda = n;
      p.B = reinterpret_cast<const float*>(w.eqT); p.ldb = n;
      p.C = dWg; p.ldc = K;
      p.M = N; p.N = K; p.K = n;
      p.beta = grad_beta(); p.dbias_set = g_grad_overwrite; p.dbias = dbg;
      gw.p[0] = p;
      ow_note_group(gw);
      const unsigned grid = p3_setup(gw, 128, 96, !(g_ablate & 32768));
      const int pi = prof_start(PS_GATE_DW, s);
      gw.onepass = g_gemm_mode == 3;
      launch_p3_wgrad(dim3(grid), s, gw);
      prof_stop(pi, s, 2.0 * n * (double)N * K, 0.0);
      LIREC_CHECK_LAUNCH();
    } else {
      rc = lirec_gate_bwd_parts(dZg, lddzg, EE, ldee, Wg, n, K, N, split, Tn, ldtn, dWg, dbg, dEE, lddee, acc_first, drop, site_ctx,
                                site_ints, 1, stream);
    }
  }
  if (rc || parts == 1) return rc;
  // dEE = (dZg Wg) * tanh'/dropout factor on the staged rows of dZg and the q32b Wg (p3: Wg^T) the FORWARD call staged
  GemmGroup gd;
  memset(&gd, 0, sizeof(gd));
  gd.nprob = 2;
  for (int h = 0; h < 2; ++h) {
    const int c0 = h == 0 ? 0 : split, nc = h == 0 ? split : K - split;
    GemmProblem p = make_problem();
    p.A = reinterpret_cast<const float*>(w.zq); p.lda = N;
    if (p3) { p.B = reinterpret_cast<const float*>(w.wqT + 4096L * (c0 / 32) * (g_gemm_mode == 3 ? N / 64 : N / 32)); p.ldb = N; }    // rows c0 .. of Wg^T [K][N] (single pass: q16c)
    else { p.B = reinterpret_cast<const float*>(w.wq + 4096L * (c0 / 32)); p.ldb = K; }
    p.C = dEE + c0; p.ldc = lddee;
    p.M = n; p.N = nc; p.K = N;
    p.epi = EPI_TANH_BWD; p.aux = Tn + c0; p.ldaux = ldtn;
    p.beta = (h == 0 && acc_first) ? 1.f : 0.f;
    set_dropout(p, drop, h == 0 ? site_ctx : site_ints, 0);
    gd.p[h] = p;
  }
  if (p3) {
    const int ni = p3_pick_ni(n, K, (split & 127) == 0 && ((K - split) & 127) == 0);
    const unsigned grid = p3_setup(gd, 128, 32 * ni);
    const int pi = prof_start(PS_GATE_DEE, s);
    gd.onepass = g_gemm_mode == 3;
    launch_p3_dgrad(ni, dim3(grid), s, gd);
    prof_stop(pi, s, 2.0 * n * (double)N * K, 0.0);
    LIREC_CHECK_LAUNCH();
    return LIREC_OK;
  }
  return launch_p2<L_NN>(gd, s, PS_GATE_DEE, nullptr, 1);
}

// ---------------------------------------------------------------------------
// one head's forward problem
static int linear_fwd_problem(const lirec_linear_fwd_args& v, GemmProblem& p) {
  if (!v.A || !v.W || !v.Y || v.n < 0 || v.K < 1 || v.N < 1) return LIREC_EINVAL;
  p = make_problem();
  p.A = v.A; p.lda = v.lda; p.B = v.W; p.ldb = v.K; p.bias = v.b; p.C = v.Y; p.ldc = v.ldy;
  p.M = v.n; p.N = v.N; p.K = v.K; p.epi = EPI_STORE;
  return LIREC_OK;
}

int lirec_linear_fwd_group(const lirec_linear_fwd_args* v, int32_t count, lirec_stream_t stream) {
  if (!v || count < 1 || count > LIREC_MAX_PROB) return LIREC_EINVAL;
  GemmGroup g; g.nprob = count;
  for (int i = 0; i < count; ++i) {
    const int rc = linear_fwd_problem(v[i], g.p[i]);
    if (rc) return rc;
  }
  return launch_gemm(L_NT, g, (hipStream_t)stream, PS_LINEAR_FWD);
}

int lirec_linear_fwd(const float* A, int64_t lda, const float* W, const float* b, int32_t n, int32_t K,
                     int32_t N, float* Y, int64_t ldy, lirec_stream_t stream) {
  const lirec_linear_fwd_args v = {A, lda, W, b, Y, ldy, n, K, N, 0};
  return lirec_linear_fwd_group(&v, 1, stream);
}

int lirec_linear_bwd_group(const lirec_linear_bwd_args* v, int32_t count, lirec_stream_t stream) {
  if (!v || count < 1 || count > LIREC_MAX_PROB) return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // (diagnostics bit 131072: the weight-gradient launches of this call start ~2 ms late)
  if ((g_ablate & 131072) && v[0].parts != 2) lirec::launch(spin_kernel, dim3(1), dim3(64), 0, s, 200000LL);
  GemmGroup gw, gd;
  gw.nprob = 0; gd.nprob = 0;
  for (int i = 0; i < count; ++i) {
    const lirec_linear_bwd_args& a = v[i];
    if (!a.dY || !a.A || !a.W || !a.dW || !a.db || a.n < 0 || a.K < 1 || a.N < 1 || a.parts < 0 || a.parts > 2) return LIREC_EINVAL;
    if (a.dA && a.mode != 0 && !a.act) return LIREC_EINVAL;
    if (a.parts != 2) {
      GemmProblem w = make_problem();
      w.A = a.dY; w.lda = a.lddy; w.B = a.A; w.ldb = a.lda; w.C = a.dW; w.ldc = a.K;
      w.M = a.N; w.N = a.K; w.K = a.n; w.beta = grad_beta(); w.dbias_set = g_grad_overwrite; w.dbias = a.db;
      gw.p[gw.nprob++] = w;
    }
    if (!a.dA || a.parts == 1) continue;
    GemmProblem p = make_problem();
    p.A = a.dY; p.lda = a.lddy; p.B = a.W; p.ldb = a.K; p.C = a.dA; p.ldc = a.ldda;
    p.M = a.n; p.N = a.K; p.K = a.N;
    p.beta = a.accumulate ? 1.f : 0.f;
    const float pd = a.drop.p;
    if (a.mode == 1) {
      p.epi = EPI_RELU_BWD; p.aux = a.act; p.ldaux = a.ldact;
      p.drop_scale = (pd > 0.f) ? (float)(1.0 / (1.0 - (double)pd)) : 1.f;
    } else if (a.mode == 2) {
      p.epi = EPI_TANH_BWD; p.aux = a.act; p.ldaux = a.ldact;
      set_dropout(p, &a.drop, a.drop.site2, 0);
    } else {
      p.epi = EPI_STORE;
    }
    gd.p[gd.nprob++] = p;
  }
  int rc = launch_gemm(L_TN, gw, s, PS_LINEAR_DW);
  if (rc || gd.nprob == 0) return rc;
  return launch_gemm(L_NN, gd, s, PS_LINEAR_DA);
}

int lirec_linear_bwd(const float* dY, int64_t lddy, const float* A, int64_t lda, const float* W,
                     int32_t n, int32_t K, int32_t N, float* dW, float* db,
                     float* dA, int64_t ldda, int32_t mode, const float* act, int64_t ldact,
                     int32_t accumulate, const lirec_dropout* drop, lirec_stream_t stream) {
  lirec_linear_bwd_args v;
  memset(&v, 0, sizeof(v));
  v.dY = dY; v.lddy = lddy; v.A = A; v.lda = lda; v.W = W; v.n = n; v.K = K; v.N = N; v.dW = dW; v.db = db;
  v.dA = dA; v.ldda = ldda; v.mode = mode; v.act = act; v.ldact = ldact; v.accumulate = accumulate; v.parts = 0;
  if (drop) v.drop = *drop;
  return lirec_linear_bwd_group(&v, 1, stream);
}

// ---------------------------------------------------------------------------
int lirec_margin_loss(const lirec_margin_loss_args* a, lirec_stream_t stream) {
  if (!a || !a->ints || !a->y || a->B < 1 || a->T < 1 || a->C < 1 || a->sample < 0 || a->sample > 2) return LIREC_EINVAL;
  const bool probs_only = a->sample == 2;
  if (probs_only ? !(a->probs_out || a->sel_out) : (!a->d_ints || !a->loss || !a->partial)) return LIREC_EINVAL;
  if (a->rels && (!a->r || (!probs_only && !a->d_rels) || a->NR < 1)) return LIREC_EINVAL;
  if (a->rels && a->rels_mean_valid && a->T != 1) return LIREC_EINVAL;
  if (a->sample && a->tr_correct) return LIREC_EINVAL;             // mlp/model.py:469,539: assert not opt.tr_correct
  if (a->batch_divisor < 0.f || a->rels_divisor < 0.f) return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int NR1 = a->rels ? a->NR + 1 : 0;
  const size_t shm = ((size_t)a->T * a->C + (size_t)a->T * NR1 + 16 + 4 + 2 * (size_t)a->T + a->C) * sizeof(float);
  if (shm > 160 * 1024) return LIREC_EINVAL;
  const int pi = prof_start(PS_LOSS, s);
  lirec::launch(margin_loss_kernel, dim3(a->B), dim3(256), shm, s, *a);
  if (!probs_only && !a->arrive) {
    LIREC_CHECK_LAUNCH();
    lirec::launch(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)a->partial, 2 * a->B, a->loss);
  }
  prof_stop(pi, s, 0.0, 8.0 * a->B * a->T * ((double)a->C + (a->rels ? a->NR : 0)));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_heads_loss_fwd_bwd(const lirec_linear_fwd_args* heads, const lirec_linear_bwd_args* back, int32_t n_heads,
                             const lirec_margin_loss_args* loss, lirec_stream_t stream) {
  if (!heads || n_heads < 1 || n_heads > LIREC_MAX_PROB || (loss && !back)) return LIREC_EINVAL;
  int rc = lirec_linear_fwd_group(heads, n_heads, stream);
  if (rc || !loss) return rc;
  // the loss reads the logits the heads just wrote and leaves d(loss)/d(logits) where `back` expects them
  bool ints_ok = false;
  for (int h = 0; h < n_heads; ++h) ints_ok = ints_ok || heads[h].Y == loss->ints;
  if (!ints_ok) return LIREC_EINVAL;
  rc = lirec_margin_loss(loss, stream);
  if (rc) return rc;
  lirec_linear_bwd_args b[LIREC_MAX_PROB];
  for (int h = 0; h < n_heads; ++h) { b[h] = back[h]; b[h].parts = 2; }
  return lirec_linear_bwd_group(b, n_heads, stream);
}

int lirec_ce_loss(const float* ints, int64_t ld_ints, const float* rels, int64_t ld_rels,
                  const int32_t* y, const int32_t* r, const float* class_w,
                  int32_t B, int32_t C, int32_t NR, float* d_ints, int64_t ld_dints,
                  float* d_rels, int64_t ld_drels, float* loss, float* partial,
                  float den_ints, float den_rels, const float* dens_dev, lirec_stream_t stream) {
  if (!ints || !y || !d_ints || !loss || !partial || B < 1 || C < 1 || den_ints < 0.f || den_rels < 0.f) return LIREC_EINVAL;
  if (rels && (!r || !d_rels || NR < 1)) return LIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = rels ? 2 * B : B;
  lirec::launch(ce_loss_kernel, dim3(nblk), dim3(256), 0, s, ints, (long)ld_ints, rels, (long)ld_rels, y, r,
                     class_w, B, C, NR, d_ints, (long)ld_dints, d_rels, (long)ld_drels, partial, den_ints, den_rels, dens_dev);
  LIREC_CHECK_LAUNCH();
  lirec::launch(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, nblk, loss);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

// ---------------------------------------------------------------------------
int lirec_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, const int64_t* step_dev, lirec_stream_t stream) {
  if (!p || !g || !m || !v || n < 0 || (step < 1 && !step_dev)) return LIREC_EINVAL;
  if (step < 1) step = 1;
  if (n == 0) return LIREC_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const int pi = prof_start(PS_ADAM, (hipStream_t)stream);
  lirec::launch(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                     step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, grad_scale, lr, (const long long*)step_dev,
                     (long long*)nullptr, (int*)nullptr, 0);
  prof_stop(pi, (hipStream_t)stream, 0.0, 28.0 * (double)n);     // read p,g,m,v; write p,m,v
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_adam_step_counted(float* p, const float* g, float* m, float* v, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay,
                            float grad_scale, int64_t* count_dev, int32_t* ticket, int32_t advance, lirec_stream_t stream) {
  if (!p || !g || !m || !v || n < 1 || !count_dev || !ticket) return LIREC_EINVAL;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const int pi = prof_start(PS_ADAM, (hipStream_t)stream);
  lirec::launch(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                     0.f, 0.f, beta1, beta2, eps, weight_decay, grad_scale, lr, (const long long*)nullptr,
                     (long long*)count_dev, (int*)ticket, (int)(advance != 0));
  prof_stop(pi, (hipStream_t)stream, 0.0, 28.0 * (double)n);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_eval_max_tracks(const lirec_eval_args* a, lirec_stream_t stream) {
  if (!a || !a->ints || !a->y || !a->g || !a->counters || a->B < 1 || a->T < 1 || a->C < 1) return LIREC_EINVAL;
  if (a->rels && (!a->r || a->NR < 1)) return LIREC_EINVAL;
  const int NR = a->rels ? a->NR : 0, NR1 = a->rels ? a->NR + 1 : 0;
  const size_t shm = ((size_t)2 * a->T * a->C + (size_t)a->T * (NR + NR1) + 512) * sizeof(float);
  if (shm > 160 * 1024) return LIREC_EINVAL;
  lirec::launch(eval_max_tracks_kernel, dim3(a->B), dim3(256), shm, (hipStream_t)stream, *a);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_counter_add(int64_t* ctr, const int64_t* inc_host, int32_t n, lirec_stream_t stream) {
  if (!ctr || !inc_host || n < 1 || n > 4) return LIREC_EINVAL;
  long long inc[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) inc[i] = inc_host[i];
  lirec::launch(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)ctr, inc[0], inc[1], inc[2],
                     inc[3], (int)n);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_zero_count(void* p, int64_t bytes, int64_t* ctr, const int64_t* inc_host, int32_t n, lirec_stream_t stream) {
  if (bytes < 0 || (!p && bytes > 0) || (reinterpret_cast<uintptr_t>(p) & 15) || n < 0 || n > 4 || (n > 0 && (!ctr || !inc_host)))
    return LIREC_EINVAL;
  long long inc[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) inc[i] = inc_host[i];
  const long n16 = (long)(bytes / 16);
  const int ntail = (int)(bytes - n16 * 16);
  long blocks = (n16 + 4 * 256 - 1) / (4 * 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  lirec::launch(zero_count_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)p, n16,
                (unsigned char*)p + n16 * 16, ntail, (long long*)(n > 0 ? ctr : nullptr), inc[0], inc[1], inc[2], inc[3], (int)n);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_cast_f64_f32(const double* src, float* dst, int64_t n, lirec_stream_t stream) {
  if (!src || !dst || n < 0) return LIREC_EINVAL;
  if (n == 0) return LIREC_OK;
  long blocks = (n / 2 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const int pi = prof_start(PS_CAST, (hipStream_t)stream);
  lirec::launch(cast_f64_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  prof_stop(pi, (hipStream_t)stream, 0.0, 12.0 * (double)n);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

static int gather_features_impl(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                                const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                                float* out, int64_t ld_out, bool out_bf16, lirec_stream_t stream);

int lirec_gather_features(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                          const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                          float* out, int64_t ld_out, lirec_stream_t stream) {
  return gather_features_impl(clip, ld_clip, track, ld_track, table_f64, index, rows, clip_dim, track_dim, out, ld_out, false, stream);
}

int lirec_gather_features_bf16(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                               const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                               void* out, int64_t ld_out, lirec_stream_t stream) {
  return gather_features_impl(clip, ld_clip, track, ld_track, table_f64, index, rows, clip_dim, track_dim, (float*)out, ld_out, true,
                              stream);
}

static int gather_features_impl(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                                const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                                float* out, int64_t ld_out, bool out_bf16, lirec_stream_t stream) {
  if (!clip || !track || !index || !out || rows < 0 || clip_dim < 0 || track_dim < 0) return LIREC_EINVAL;
  const int D = clip_dim + 2 * track_dim;
  if (D < 4 || (clip_dim & 3) || (track_dim & 3) || (ld_out & 3) || (ld_clip & 3) || (ld_track & 3) || ld_out < D) return LIREC_EINVAL;
  if ((reinterpret_cast<uintptr_t>(clip) | reinterpret_cast<uintptr_t>(track) | reinterpret_cast<uintptr_t>(out)) & 15) return LIREC_EINVAL;
  if (rows == 0) return LIREC_OK;
  const long total = (long)rows * (D / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  const int pi = prof_start(PS_STAGE, s);
  if (table_f64 && out_bf16)
    lirec::launch(HIP_KERNEL_NAME(gather_features_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, s, clip, (long)ld_clip, track,
                  (long)ld_track, index, (long)rows, clip_dim, track_dim, out, (long)ld_out);
  else if (table_f64)
    lirec::launch(HIP_KERNEL_NAME(gather_features_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, clip, (long)ld_clip, track,
                  (long)ld_track, index, (long)rows, clip_dim, track_dim, out, (long)ld_out);
  else if (out_bf16)
    lirec::launch(HIP_KERNEL_NAME(gather_features_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, clip, (long)ld_clip, track,
                  (long)ld_track, index, (long)rows, clip_dim, track_dim, out, (long)ld_out);
  else
    lirec::launch(HIP_KERNEL_NAME(gather_features_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, clip, (long)ld_clip, track,
                  (long)ld_track, index, (long)rows, clip_dim, track_dim, out, (long)ld_out);
  prof_stop(pi, s, 0.0, (out_bf16 ? 2.0 : 4.0) * (double)rows * D);   // bytes written (the reads are table hits)
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_grid_pool(const float* grid, int32_t F, int32_t C, int32_t H, int32_t W, const int32_t* boxes,
                    const int32_t* estart, int32_t n_out, float* out, int64_t ld_out, lirec_stream_t stream) {
  if (!grid || !boxes || !estart || !out || F < 1 || C < 1 || H < 1 || W < 1 || n_out < 0 || ld_out < C) return LIREC_EINVAL;
  if (n_out == 0) return LIREC_OK;
  lirec::launch(grid_pool_kernel, dim3((C + 255) / 256, n_out), dim3(256), 0, (hipStream_t)stream, grid, F, C, H, W, boxes,
                     estart, out, (long)ld_out);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_rows_max(const float* src, int64_t ld, const int32_t* idx, const int32_t* estart, int32_t n_out, int32_t dim,
                   float* out, int64_t ld_out, lirec_stream_t stream) {
  if (!src || !idx || !estart || !out || dim < 1 || n_out < 0 || ld_out < dim) return LIREC_EINVAL;
  if (n_out == 0) return LIREC_OK;
  lirec::launch(rows_max_kernel, dim3((dim + 255) / 256, n_out), dim3(256), 0, (hipStream_t)stream, src, (long)ld, idx,
                     estart, dim, out, (long)ld_out);
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

int lirec_dropout_mask(uint8_t* keep, int32_t rows, int32_t cols, const lirec_dropout* drop, int32_t site,
                       lirec_stream_t stream) {
  if (!keep || !drop || rows < 0 || cols < 0) return LIREC_EINVAL;
  if ((long)rows * cols == 0) return LIREC_OK;
  lirec::launch(dropout_mask_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, keep, rows, cols,
                     (unsigned)(drop->seed & 0xffffffffull), (unsigned)(drop->seed >> 32), (const unsigned long long*)drop->seed_dev,
                     (unsigned)site,
                     drop_thresh(drop->p));
  LIREC_CHECK_LAUNCH();
  return LIREC_OK;
}

}  // extern "C"
