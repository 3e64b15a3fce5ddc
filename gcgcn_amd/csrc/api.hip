// C ABI of libgcgcn_hip.so (include/gcgcn.h): host-side orchestration of the HIP kernels for
// each block of the CAGGC/MAGGC path.  Every function only enqueues work on the caller's stream.
#include "../../include/gcgcn.h"

#include <stdarg.h>
#include <string.h>

#include <new>
#include <vector>

#include <stdlib.h>

#include "gat_body.hpp"
#include "gcn_plan.hpp"
#include "gemm.hpp"
#include "rowops.hpp"

namespace gc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// ---- per-kernel timing ---------------------------------------------------------------------------
static struct {
  bool on = false;
  char filter[48] = "";
  std::vector<hipEvent_t> ev;  // pairs: start, stop
  int used = 0;                // pairs recorded
  double work = 0;             // summed work of the recorded launches
} g_prof;

ProfScope::ProfScope(const char* tag, hipStream_t s, double work) : slot(-1), st(s) {
  if (!g_prof.on || strncmp(tag, g_prof.filter, strlen(g_prof.filter)) != 0) return;
  if (2 * (g_prof.used + 1) > (int)g_prof.ev.size()) return;  // ring full: stop recording
  slot = g_prof.used++;
  g_prof.work += work;
  (void)hipEventRecord(g_prof.ev[2 * slot], st);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
}
bool prof_events(const char* tag, double work, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof.on || strncmp(tag, g_prof.filter, strlen(g_prof.filter)) != 0) return false;
  if (2 * (g_prof.used + 1) > (int)g_prof.ev.size()) return false;  // ring full: stop recording
  const int slot = g_prof.used++;
  g_prof.work += work;
  *start = g_prof.ev[2 * slot], *stop = g_prof.ev[2 * slot + 1];
  return true;
}

struct GcnLayout {
  long oWnX, oWe, oWd, oWlin, oblin, total, wd_head;
  int gh;
  long wd_off(int h, int l) const { return oWd + h * wd_head + (long)gh * gh * l * (l - 1) / 2; }
};
static GcnLayout gcn_layout(int D, int L, int H) {
  GcnLayout y;
  y.gh = D / L;
  const long DHD = (long)D * H * D;
  y.wd_head = (long)y.gh * y.gh * L * (L - 1) / 2;
  y.oWnX = 0;
  y.oWe = DHD;
  y.oWd = 2 * DHD;
  y.oWlin = y.oWd + H * y.wd_head;
  y.oblin = y.oWlin + DHD;
  y.total = y.oblin + D;
  return y;
}

static int check_dims(const char* who, int B, int N, int D, int L, int H) {
  GC_REQUIRE(B > 0 && N > 0 && D > 0, "%s: bad shape B=%d N=%d D=%d", who, B, N, D);
  GC_REQUIRE(L > 0 && D % L == 0, "%s: D=%d not divisible by layer_num=%d", who, D, L);
  GC_REQUIRE(H > 0 && D % H == 0, "%s: D=%d not divisible by head_num=%d", who, D, H);
  return 0;
}

// Workspace of one block call: [split-K partials / column-sum partials | partials of a riding column sum]
static long col_ride_elems(int D) { return (long)COL_RIDE_SLICES * D; }
static long gemm_scratch_elems(int B, int N, int D, int H) {
  const long a = colsum_scratch_elems((long)B * N, D, 1);
  const long rows = (long)B * N > D ? (long)B * N : D;
  const long b = gemm_ws_elems(rows, (long)H * D);
  return a > b ? a : b;
}
static long scratch_elems(int B, int N, int D, int H) { return gemm_scratch_elems(B, N, D, H) + col_ride_elems(D); }

// Run-time A/B switches.  option("head_v1", -1) reads gcgcn_set_option's value if one was set, else the environment
// variable GCGCN_HEAD_V1 (read once), else the default.  Every knob a test has to flip lives here, so that one process
// can run both sides of an A/B (function-local statics around getenv could not be switched by the test suite).
struct Opt {
  const char* name;
  int value;
  bool resolved;
};
static Opt g_opts[] = {{"head_v1", 0, false},      {"head_bil3", 0, false},    {"head_bil3_bwd", 0, false}, {"head_dw3", 0, false},
                       {"chain_fuse", 0, false},   {"chain_carry", 0, false},  {"gat_ride", 0, false},      {"chain_t", 0, false},
                       {"mha_ride", 0, false},     {"maggc_fuse", 0, false},   {"carry_spread", 0, false},  {"chain_spread", 0, false},
                       {"carry_cohort", 0, false}, {"chain_cohort", 0, false}, {"carry_spread_min", 0, false},
                       {"chain_spread_min", 0, false}, {"att_in_chain", 0, false}, {"fold_slices", 0, false}, {"head_sum_fold", 0, false},
                       {"head_compact", 0, false}, {"chain_big", 0, false},   {"split_widen", 0, false}};
int option(const char* name, int dflt) {
  for (Opt& o : g_opts) {
    if (strcmp(o.name, name) != 0) continue;
    if (!o.resolved) {
      char env[64] = "GCGCN_";
      size_t n = strlen(env);
      for (const char* c = name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
      env[n] = 0;
      const char* e = getenv(env);
      o.value = e ? atoi(e) : dflt;
      o.resolved = true;
    }
    return o.value;
  }
  return dflt;
}
static bool set_opt(const char* name, int value) {
  for (Opt& o : g_opts)
    if (strcmp(o.name, name) == 0) {
      o.value = value, o.resolved = true;
      return true;
    }
  return false;
}

// GCGCN_NO_CHAIN=1 (or gcgcn_set_option("chain", 0)) runs every per-(doc, head) product as its own batched
// launch instead of inside the chain kernels (A/B testing of chain.hip).
static int g_chain = -1;
static bool use_chain() {
  if (g_chain < 0) {
    const char* e = getenv("GCGCN_NO_CHAIN");
    g_chain = (e && e[0] == '1') ? 0 : 1;
  }
  return g_chain != 0;
}

// Graphs of more than 64 entities have no LDS-resident chain kernel: the generic chain kernels hand every product's tiles through
// L2 inside one workgroup per (document, head) pair, and at that size each product is a full launch of its own anyway (cfg 5:
// 4096 tiles).  Measured at cfg 5 (same session, profiles/r04_ab_chain_big.txt): per-product launches 7.67-7.74 ms, chain
// kernels 7.86-7.90 ms per step; the one chain launch that pays is a FORWARD one with an edge mean riding in it (the chain hides
// under the HBM-bound passenger: 698 us against 705 + 64).  Option chain_big = 1 restores the chain kernels everywhere (A/B, tests).
static bool use_chain_for(int N, bool fwd_with_ride) {
  if (!use_chain()) return false;
  return N <= 64 || fwd_with_ride || option("chain_big", 0) != 0;
}

// GCGCN_NO_MHA_CORE=1 (or gcgcn_set_option("mha_core", 0)) sends small graphs through the generic batched-GEMM +
// row-softmax attention path as well (A/B testing of mha_core.hip).
static int g_mha_core = -1;
static bool use_mha_core() {
  if (g_mha_core < 0) {
    const char* e = getenv("GCGCN_NO_MHA_CORE");
    g_mha_core = (e && e[0] == '1') ? 0 : 1;
  }
  return g_mha_core != 0;
}

// gcgcn_edge_ride -> EdgeRide (kind 1: edge mean forward, 2: its backward); NULL = no passenger
static int make_ride(const char* who, const gcgcn_edge_ride* ride, int kind, EdgeRide& r) {
  memset(&r, 0, sizeof(r));
  if (!ride) return 0;
  GC_REQUIRE(ride->B > 0 && ride->N > 0 && ride->D > 0 && ride->in && ride->out, "%s: bad edge ride B=%d N=%d D=%d", who,
             ride->B, ride->N, ride->D);
  GC_REQUIRE((long)ride->B * ride->N <= 0x3fffffffL, "%s: edge ride too large", who);
  r.kind = kind, r.B = ride->B, r.N = ride->N, r.D = ride->D;
  r.in = ride->in, r.n_valid = ride->n_valid, r.out = ride->out;
  return 0;
}

// Row blocks of a ragged batch (gcgcn_row_blocks) on one GEMM problem: mode 1 = M is the document-row dimension, 2 = K is.
// (both walk the same list of live 16-row blocks: mode 1 four at a time as a 64-row tile, mode 2 two at a time as a k-tile)
static void use_rows(GemmArgs& g, const int* rowblk, int mode, int zero_dead = 0) {
  if (!rowblk) return;
  g.rb = rowblk + ROWBLK_HDR, g.rb_n = rowblk;
  g.rb_mode = mode, g.rb_zero = zero_dead;
}

static GcnCtx make_ctx(int B, int N, int D, int L, int H, const GcnLayout& y, const float* X, const float* A,
                       const float* flat, const int* n_valid, Drop drop) {
  GcnCtx c;
  memset(&c, 0, sizeof(c));
  c.B = B, c.N = N, c.D = D, c.L = L, c.H = H, c.gh = y.gh;
  c.HD = (long)H * D, c.oWd = y.oWd, c.wd_head = y.wd_head;
  c.X = X, c.A = A, c.flat = flat, c.n_valid = n_valid, c.drop = drop;
  return c;
}

}  // namespace gc

using namespace gc;

extern "C" {

int gcgcn_version(void) { return 7; }
const char* gcgcn_last_error(void) { return g_err; }

int gcgcn_set_option(const char* name, int value) {
  GC_REQUIRE(name, "set_option: null name");
  if (strcmp(name, "chain") == 0) {
    g_chain = value ? 1 : 0;
    return 0;
  }
  if (strcmp(name, "mha_core") == 0) {
    g_mha_core = value ? 1 : 0;
    return 0;
  }
  if (set_opt(name, value)) return 0;
  set_error("set_option: unknown option '%s'", name);
  return 1;
}

int64_t gcgcn_row_blocks_ints(int B, int N) { return (B > 0 && N > 0 && N % 16 == 0) ? row_blocks_ints(B, N) : 0; }
int gcgcn_row_blocks(int B, int N, const int32_t* n_valid, int32_t* out, void* stream) {
  return row_blocks(n_valid, B, N, out, (hipStream_t)stream);
}

int gcgcn_prof_start(const char* kernel_prefix, int capacity) {
  GC_REQUIRE(kernel_prefix && capacity > 0, "prof_start: bad arguments");
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.assign(2 * (size_t)capacity, nullptr);
  for (auto& e : g_prof.ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      set_error("prof_start: hipEventCreate failed");
      return 2;
    }
  }
  strncpy(g_prof.filter, kernel_prefix, sizeof(g_prof.filter) - 1);
  g_prof.used = 0;
  g_prof.work = 0;
  g_prof.on = true;
  return 0;
}
int gcgcn_prof_enable(int on) {  // pause / resume recording without touching what was recorded
  g_prof.on = on != 0 && !g_prof.ev.empty();
  return 0;
}
int gcgcn_prof_stop(double* total_ms, int* launches, double* work) {
  GC_REQUIRE(total_ms && launches, "prof_stop: null pointer");
  if (work) *work = g_prof.work;
  g_prof.on = false;
  double tot = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) {
      set_error("prof_stop: event query failed");
      return 2;
    }
    tot += ms;
  }
  *total_ms = tot;
  *launches = g_prof.used;
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.used = 0;
  return 0;
}

int gcgcn_rng_next(void* state, void* snaps, int count, void* stream) {
  GC_REQUIRE(state && snaps && count > 0, "rng_next: bad arguments");
  return rng_next(state, snaps, count, (hipStream_t)stream);
}
int gcgcn_dropout_keep(uint8_t* keep, int64_t n, const void* rng_snap, uint64_t salt, float p, void* stream) {
  GC_REQUIRE(keep && rng_snap, "dropout_keep: null pointer");
  Drop d = make_drop(rng_snap, salt, p);
  d.snap = (const uint64_t*)rng_snap;
  return dropout_keep(keep, n, d, (hipStream_t)stream);
}
int gcgcn_dropout(const float* x, float* y, int64_t n, const void* rng_snap, uint64_t salt, float p, void* stream) {
  GC_REQUIRE(x && y && rng_snap, "dropout: null pointer");
  GC_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%f out of [0,1)", p);
  Drop d = make_drop(rng_snap, salt, p);
  d.snap = (const uint64_t*)rng_snap;
  return dropout(x, y, n, d, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// GATAttention
// ---------------------------------------------------------------------------------------------
int gcgcn_gat_layout(int D, int Dh, int64_t* o) {
  GC_REQUIRE(D > 0 && Dh > 0 && o, "gat_layout: bad arguments");
  const long DD = (long)Dh * D;
  o[0] = 0;              // W_h [Dh, D]
  o[1] = DD;             // b_h
  o[2] = o[1] + Dh;      // W_t
  o[3] = o[2] + DD;      // b_t
  o[4] = o[3] + Dh;      // W_r
  o[5] = o[4] + DD;      // b_r
  o[6] = o[5] + Dh;      // wt
  o[7] = o[6] + 3 * Dh;  // wt bias
  o[8] = o[7] + 1;
  return 0;
}

int gcgcn_gat_fwd(int B, int N, int D, int Dh, const float* X, const float* E, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, float* uvc, float* s, float* P, float* A, float* Ebar, void* rng_state,
                  void* rng_snaps, int rng_count, const uint8_t* mask, int uvc_valid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("gat_fwd", B, N, D, 1, 1));
  GC_REQUIRE(Dh > 0, "gat_fwd: hidden_dim=%d", Dh);
  GC_REQUIRE(X && E && flat && uvc && s && P && Ebar, "gat_fwd: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GAT, p);
  GC_REQUIRE(!drop.snap || A, "gat_fwd: dropout on but A is NULL");
  const long M = (long)B * N;
  GC_REQUIRE(!rng_state || (rng_snaps && rng_count > 0), "gat_fwd: rng_state given without snapshots to fill");
  if (!uvc_valid) {
    GC_TRY(gat_fold_fwd(flat, uvc, D, Dh, st, rng_state, rng_snaps, rng_count));  // + gcgcn_rng_next, if asked to
    GC_TRY(node_score_fwd(X, uvc, s, M, D, st));
  } else {  // the caller kept uvc from an earlier call with the same parameters: the draw rides in the next kernel instead
    GC_TRY(node_score_fwd(X, uvc, s, M, D, st, rng_state, rng_snaps, rng_count));
  }
  GC_TRY(edge_fwd(E, uvc + D, n_valid, Ebar, s, P, A, drop, B, N, D, st, mask));  // + row softmax + dropout
  return 0;
}

int64_t gcgcn_gat_bwd_scratch(int B, int N, int D) {
  const long a = colsum_scratch_elems((long)B * N, D, 1), b = 3L * 64 * (2 * D + 1);
  return a > b ? a : b;
}

int gcgcn_gat_bwd(int B, int N, int D, int Dh, const float* X, const float* E, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, const float* uvc, const float* P, const float* dA, const float* dEbar,
                  const float* dX_in, float* dX, float* dE, float* dflat, float* dlogit, float* ds, float* dvpart,
                  float* duvc, float* scratch, void* defer_queue, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("gat_bwd", B, N, D, 1, 1));
  GC_REQUIRE(Dh > 0, "gat_bwd: hidden_dim=%d", Dh);
  GC_REQUIRE(X && E && flat && uvc && P && dA && dX && dflat && dlogit && ds && dvpart && duvc,
             "gat_bwd: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GAT, p);
  const long M = (long)B * N;
  GC_REQUIRE(scratch, "gat_bwd: scratch is required");
  const bool small = gat_dlogit_ok(N);
  const bool ride = option("gat_ride", 1) != 0;
  GatTail tail{P, dA, uvc, dX_in, ds, dX, drop, B, gat_dlogit_slices(D)};
  if (small && ride) {
    // dlogit, ds and dX = ds u + dX_in ride in the edge pass below: its entity rows take their dlogit row from P and dA
    // themselves, B * slices passenger workgroups produce ds and dX (dlogit is never stored)
  } else if (small) {  // the same in a launch of its own, one workgroup per (document, slice)
    GC_TRY(gat_dlogit(P, dA, uvc, dX_in, dlogit, ds, dX, B, N, D, drop, st));
  } else {
    GC_TRY(softmax_bwd(P, dA, dlogit, M, N, drop, st));
    // ds[b, j] = sum_i dlogit[b, i, j]
    GC_TRY(colsum(dlogit, nullptr, ds, N, N, N, B, (long)N * N, 0, N, 0, nullptr, st));
    GC_TRY(node_score_bwd(ds, uvc, dX_in, dX, M, D, st));
  }
  GC_TRY(edge_bwd(E, uvc + D, n_valid, dlogit, dEbar, dE, dvpart, B, N, D, st, (DeferQueue*)defer_queue,  // + parked weight gradients
                  small && ride ? &tail : nullptr));
  // du = sum_m ds[m] X[m,:],  dv = sum partials,  dc = sum_m ds[m]: row-slice partials in one launch; the fold's
  // backward sums the slices itself (duvc stays unused)
  long part_off[3];
  int ns = 0;
  GC_TRY(colsum3(X, ds, duvc, M, D, D, dvpart, nullptr, duvc + D, M, D, D, ds, nullptr, duvc + 2 * D, M, 1, 1, scratch,
                 st, false, part_off, &ns));
  GC_TRY(gat_fold_bwd(flat, duvc, dflat, D, Dh, st, scratch, part_off, ns));
  return 0;
}

int gcgcn_edge_mean_fwd(int B, int N, int D, const float* E, const int32_t* n_valid, float* Ebar, void* stream) {
  return edge_fwd(E, nullptr, n_valid, Ebar, nullptr, nullptr, nullptr, make_drop(nullptr, 0, 0.f), B, N, D,
                  (hipStream_t)stream);
}
int gcgcn_edge_mean_bwd(int B, int N, int D, const float* dEbar, const int32_t* n_valid, float* dE, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && D > 0, "edge_mean_bwd: bad shape");
  return edge_bcast(dEbar, n_valid, dE, B, N, D, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// MultiHeadAttention
// ---------------------------------------------------------------------------------------------
int gcgcn_mha_layout(int D, int64_t* o) {
  GC_REQUIRE(D > 0 && o, "mha_layout: bad arguments");
  o[0] = 0;
  o[1] = (long)D * D;
  o[2] = o[1] + D;
  return 0;
}

int gcgcn_mha_fwd(int B, int N, int D, int H, const float* X, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, float* Q, float* P, float* A, float* scratch, const int32_t* rowblk, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("mha_fwd", B, N, D, 1, H));
  const long wse = scratch ? gemm_scratch_elems(B, N, D, 1) : 0;
  GC_REQUIRE(X && flat && Q && P, "mha_fwd: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_MHA, p);
  GC_REQUIRE(!drop.snap || A, "mha_fwd: dropout on but A is NULL");
  const long M = (long)B * N;
  const int dh = D / H;
  {  // Q = X Wq^T + bq      (glove:136, all heads at once)
    GemmArgs g;
    g.ws = scratch, g.ws_elems = wse;
    g.A = X, g.lda = D, g.a_kc = 1;
    g.B = flat, g.ldb = D, g.b_kc = 1;
    g.C = Q, g.ldc = D;
    g.M = (int)M, g.N = D, g.K = D;
    g.bias = flat + (long)D * D;
    use_rows(g, (n_valid && N % 16 == 0 && N >= 32) ? rowblk : nullptr, 1, 1);
    GC_TRY(gemm(g, st));
  }
  const float alpha = 1.f / sqrtf((float)dh);
  if (use_mha_core() && mha_core_ok(N, D, H, Q, nullptr))  // small graph: scores stay in LDS
    return mha_core_fwd(Q, n_valid, P, A, B, N, D, H, alpha, drop, st);
  {  // S[b,h] = Q_h Q_h^T / sqrt(dh)   (glove:137-138: keys use the query projection)
    GemmArgs g;
    g.ws = scratch, g.ws_elems = wse;
    g.A = Q, g.lda = D, g.a_kc = 1, g.sA1 = (long)N * D, g.sA2 = dh;
    g.B = Q, g.ldb = D, g.b_kc = 1, g.sB1 = (long)N * D, g.sB2 = dh;
    g.C = P, g.ldc = N, g.sC1 = (long)H * N * N, g.sC2 = (long)N * N;
    g.M = N, g.N = N, g.K = dh;
    g.batch1 = B, g.batch2 = H;
    g.alpha = alpha;
    GC_TRY(gemm(g, st));
  }
  GC_TRY(softmax_fwd(P, nullptr, n_valid, P, A, M * H, N, H, drop, st));
  return 0;
}

int64_t gcgcn_mha_scratch(int B, int N, int D) { return scratch_elems(B, N, D, 1); }

int gcgcn_mha_bwd(int B, int N, int D, int H, const float* X, const float* flat, const void* rng_snap, float p,
                  const float* Q, const float* P, const float* dA, const float* dX_in, float* dX, float* dflat, float* dS,
                  float* dQ, float* scratch, void* defer_queue, int core_done, const int32_t* rowblk, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("mha_bwd", B, N, D, 1, H));
  const long wse = scratch ? gemm_scratch_elems(B, N, D, 1) : 0;
  GC_REQUIRE(X && flat && Q && P && dA && dX && dflat && dS && dQ, "mha_bwd: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_MHA, p);
  const long M = (long)B * N;
  const int dh = D / H;
  const float alpha = 1.f / sqrtf((float)dh);
  if (core_done) {  // dQ arrived with the call (computed as passengers of the convolution's backward, gcgcn_mha_hook)
  } else if (use_mha_core() && mha_core_ok(N, D, H, Q, dQ)) {
    GC_TRY(mha_core_bwd(Q, P, dA, dQ, B, N, D, H, alpha, drop, st));
  } else {
    GC_TRY(softmax_bwd(P, dA, dS, M * H, N, drop, st));
    for (int pass = 0; pass < 2; ++pass) {  // dQ_h = alpha (dS + dS^T) Q_h
      GemmArgs g;
      g.ws = scratch, g.ws_elems = wse;
      g.A = dS, g.lda = N, g.a_kc = (pass == 0), g.sA1 = (long)H * N * N, g.sA2 = (long)N * N;
      g.B = Q, g.ldb = D, g.b_kc = 0, g.sB1 = (long)N * D, g.sB2 = dh;
      g.C = dQ, g.ldc = D, g.sC1 = (long)N * D, g.sC2 = dh;
      g.M = N, g.N = dh, g.K = N;
      g.batch1 = B, g.batch2 = H;
      g.alpha = alpha;
      g.accumulate = pass;
      GC_TRY(gemm(g, st));
    }
  }
  {  // one launch: dX = dQ Wq  and  dWq = dQ^T X
    GemmArgs gs[2];
    gs[0].ws = gs[1].ws = scratch, gs[0].ws_elems = gs[1].ws_elems = wse;
    gs[0].A = dQ, gs[0].lda = D, gs[0].a_kc = 1;
    gs[0].B = flat, gs[0].ldb = D, gs[0].b_kc = 0;
    gs[0].C = dX, gs[0].ldc = D;
    gs[0].M = (int)M, gs[0].N = D, gs[0].K = D;
    gs[0].add = dX_in, gs[0].ldadd = D;  // + the gradient X already collected downstream (NULL = none)
    gs[1].A = dQ, gs[1].lda = D, gs[1].a_kc = 0;
    gs[1].B = X, gs[1].ldb = D, gs[1].b_kc = 0;
    gs[1].C = dflat, gs[1].ldc = D;
    gs[1].M = D, gs[1].N = D, gs[1].K = (int)M;
    if (N % 16 == 0 && N >= 32) use_rows(gs[0], rowblk, 1, 1), use_rows(gs[1], rowblk, 2);   // ragged batch: the rows that exist
    const int ng = gemm_defer((DeferQueue*)defer_queue, gs[1]) ? 1 : 2;  // dWq parked (see gcgcn_gcn_bwd)
    if (scratch) {  // dbq = column sums of dQ ride in the same two launches
      ColRide cr;
      cr.X = dQ, cr.out = dflat + (long)D * D, cr.part = scratch + wse, cr.R = M, cr.ld = D, cr.C = D;
      // with dWq parked the launch has nothing to reduce: the sums' second stage (one workgroup) would be a launch of its own --
      // it is parked too (nobody reads dbq before the end of backward; `scratch` must then outlive the call like dQ and X)
      bool later = false;
      GC_TRY(gemm_group(gs, ng, st, &cr, (ng == 1 && defer_queue) ? &later : nullptr));
      if (later) {
        cr.ready_slices = COL_RIDE_SLICES;
        if (!gemm_defer_col2((DeferQueue*)defer_queue, cr))
          GC_TRY(colsum(cr.part, nullptr, cr.out, cr.ready_slices, cr.C, cr.C, 1, 0, 0, 0, 0, nullptr, st));
      }
    } else {
      GC_TRY(gemm_group(gs, ng, st));
      GC_TRY(colsum(dQ, nullptr, dflat + (long)D * D, M, D, D, 1, 0, 0, 0, 0, scratch, st));
    }
  }
  return 0;
}

void* gcgcn_defer_create(void) { return new (std::nothrow) DeferQueue(); }
void gcgcn_defer_destroy(void* queue) { delete (DeferQueue*)queue; }
int gcgcn_defer_count(const void* queue) { return queue ? ((const DeferQueue*)queue)->n + ((const DeferQueue*)queue)->ncol2 : 0; }
int gcgcn_defer_flush(void* queue, void* stream) { return gemm_flush_deferred((DeferQueue*)queue, (hipStream_t)stream); }

// ---------------------------------------------------------------------------------------------
// trainer loss
// ---------------------------------------------------------------------------------------------
int gcgcn_pair_bce_fwd(int B, int N, int R, const float* logits, const float* labels, const int32_t* n_valid, float* loss,
                       float* part, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && R > 0 && (long)B * N <= 0x7fffffffL && (long)N * R <= 0x7fffffffL,
             "pair_bce_fwd: bad shape B=%d N=%d R=%d", B, N, R);
  GC_REQUIRE(logits && labels && loss && part, "pair_bce_fwd: null pointer");
  return pair_bce_fwd(logits, labels, n_valid, loss, part, B, N, R, (hipStream_t)stream);
}

int gcgcn_pair_bce_bwd(int B, int N, int R, const float* logits, const float* labels, const int32_t* n_valid,
                       const float* dloss, float* dlogits, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && R > 0, "pair_bce_bwd: bad shape B=%d N=%d R=%d", B, N, R);
  GC_REQUIRE(logits && labels && dlogits, "pair_bce_bwd: null pointer");
  return pair_bce_bwd(logits, labels, n_valid, dloss, dlogits, B, N, R, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// GraphConvolution / MultiGraphConvolution
// ---------------------------------------------------------------------------------------------
int gcgcn_gcn_layout(int D, int L, int H, int64_t* o) {
  GC_TRY(check_dims("gcn_layout", 1, 1, D, L, H));
  GC_REQUIRE(o, "gcn_layout: null pointer");
  const GcnLayout y = gcn_layout(D, L, H);
  o[0] = y.oWnX, o[1] = y.oWe, o[2] = y.oWd, o[3] = y.oWlin, o[4] = y.oblin, o[5] = y.total, o[6] = y.wd_head;
  return 0;
}

int gcgcn_gcn_fwd(int B, int N, int D, int L, int H, const float* X, const float* Ebar, const float* A,
                  const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const void* out_rng_snap,
                  float out_p, float* out, float* Pn, float* Y, float* HO, float* rinv, float* G, float* wsum, float* scratch,
                  const gcgcn_edge_ride* ride, const gcgcn_mha_hook* mha, const int32_t* rowblk, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("gcn_fwd", B, N, D, L, H));
  if (mha) {
    GC_REQUIRE(gcgcn_maggc_fusable(N, D, H) && mha->flat_q && mha->Q && mha->P, "gcn_fwd: attention hook on a shape it does not serve");
    GC_REQUIRE(mha_core_ok(N, D, H, mha->Q, nullptr), "gcn_fwd: attention hook: misaligned Q");
    A = mha->A ? mha->A : mha->P;   // what the attention core below writes
  }
  EdgeRide er;
  GC_TRY(make_ride("gcn_fwd", ride, 1, er));
  const long wse = scratch ? gemm_scratch_elems(B, N, D, H) : 0;
  GC_REQUIRE(X && Ebar && A && flat && out && Pn && Y && HO && rinv && G, "gcn_fwd: null pointer");
  const GcnLayout y = gcn_layout(D, L, H);
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GCN, p);
  const long M = (long)B * N;
  const long HD = (long)H * D;
  GC_REQUIRE(M <= 0x7fffffffL, "gcn_fwd: B*N too large");
  // The per-(doc, head) context, with the passenger that will actually ride attached.  Ragged batch with a row-block list:
  // the products around the chain run on the rows that exist; what they leave on the dead row blocks is ZERO (rb_zero), so the
  // chain kernels -- whichever serves the shape -- see exactly what the dense products would have left there.
  const bool chain = use_chain_for(N, er.kind && chain_can_carry(er));
  EdgeRide er_chain = er;
  if (er.kind && !(chain && chain_can_carry(er))) er_chain.kind = 0;
  GcnCtx c = make_ctx(B, N, D, L, H, y, X, A, flat, n_valid, drop);
  c.G = G, c.Pn = Pn, c.Y = Y, c.HO = HO, c.rinv = rinv;
  c.ride = er_chain;
  const int* rows = (rowblk && n_valid && N % 16 == 0 && N >= 32) ? rowblk : nullptr;

  {  // one launch: Pn = X WnX (node term of every (head, sub-layer), X part of the dense input)
     //             G  = Ebar We (edge term, mean commuted with the projection, glove:40-41)
    GemmArgs gs[2];
    for (int q = 0; q < 2; ++q) {
      GemmArgs& g = gs[q];
      g.ws = scratch, g.ws_elems = wse;
      g.A = q ? Ebar : X, g.lda = D, g.a_kc = 1;
      g.B = flat + (q ? y.oWe : y.oWnX), g.ldb = HD, g.b_kc = 0;
      g.C = q ? G : Pn, g.ldc = HD;
      g.M = (int)M, g.N = (int)HD, g.K = D;
      use_rows(g, rows, 1, 1);
    }
    ColRide hs;  // wsum = sum_h Wlin[:, h, :] (a by-product for gcgcn_gcn_bwd) in trailing workgroups of this launch
    if (wsum && H > 1) hs.X = flat + y.oWlin, hs.out = wsum, hs.R = H, hs.ld = D, hs.C = D * D, hs.ready_slices = -1;
    GemmArgs g3[3] = {gs[0], gs[1], GemmArgs()};
    if (mha) {  // Q = X Wq^T + bq (glove:136, all heads at once): one more problem of this launch
      GemmArgs& g = g3[2];
      g.ws = scratch, g.ws_elems = wse;
      g.A = X, g.lda = D, g.a_kc = 1;
      g.B = mha->flat_q, g.ldb = D, g.b_kc = 1;
      g.C = mha->Q, g.ldc = D;
      g.M = (int)M, g.N = D, g.K = D;
      g.bias = mha->flat_q + (long)D * D;
      use_rows(g, rows, 1, 1);   // (the attention core stages all N rows of Q: zeros past the live blocks)
    }
    GC_TRY(gemm_group(g3, mha ? 3 : 2, st, hs.X ? &hs : nullptr));
  }
  {  // the dependent per-(doc, head) sequence: normaliser, then per sub-layer dense connection + aggregation
    if (er.kind && !er_chain.kind) {  // the riding pass as its own launch
      GC_TRY(edge_fwd(er.in, nullptr, er.n_valid, er.out, nullptr, nullptr, nullptr, Drop(), er.B, er.N, er.D, st));
      er.kind = 0;
    }
    if (mha) {  // the attention core: scores in LDS, P / A out (glove:137-140) -- in the chain workgroups' prologue where the
                // shape's chain kernel can do that (chain.hip), as a launch of its own otherwise
      const Drop adrop = make_drop(mha->rng_snap, GCGCN_SALT_MHA, mha->p);
      GC_REQUIRE(!adrop.snap || mha->A, "gcn_fwd: attention dropout on but A is NULL");
      const float alpha = 1.f / sqrtf((float)(D / H));
      if (chain && chain_fwd_computes_attention(c))
        c.mha.Q = mha->Q, c.mha.P = mha->P, c.mha.A = mha->A, c.mha.alpha = alpha, c.mha.drop = adrop, c.mha.dh = D / H,
        c.mha.kchunk = mha_chunk(D / H);
      else
        GC_TRY(mha_core_fwd(mha->Q, n_valid, mha->P, mha->A, B, N, D, H, alpha, adrop, st));
    }
    if (chain) {
      GC_TRY(gcn_chain_fwd(c, st));
    } else {
      GC_TRY(rowsum_inv(A, rinv, (long)B * H * N, N, st));  // glove:47-49
      for (int l = 0; l < L; ++l) {
        if (l > 0) GC_TRY(gemm(plan_fwd_dense(c, l), st, 0, 1));
        GC_TRY(gemm(plan_fwd_agg(c, l), st, 0, 1));
      }
    }
  }
  {  // out = HO Wlin^T + blin   (glove:78 / 118)
    GemmArgs g;
    g.ws = scratch, g.ws_elems = wse;
    g.A = HO, g.lda = HD, g.a_kc = 1;
    g.B = flat + y.oWlin, g.ldb = HD, g.b_kc = 1;
    g.C = out, g.ldc = D;
    g.M = (int)M, g.N = D, g.K = (int)HD;
    g.bias = flat + y.oblin;
    g.n_valid = n_valid, g.nv_rows = N, g.nv_zdoc = 0;
    use_rows(g, rows, 1, 1);   // the block's output: its padding rows are zero
    const Drop odrop = make_drop(out_rng_snap, GCGCN_SALT_GLUE, out_p);
    if (odrop.snap) {  // the hop's output dropout (glove:341) in the same epilogue: out = dropout(linear)
      g.C = G;         // the undropped values go to workspace that is free by now; nobody reads them
      g.C2 = out, g.ldc2 = D;
      g.drop = odrop;
    }
    GC_TRY(gemm(g, st));
  }
  return 0;
}

int64_t gcgcn_gcn_scratch(int B, int N, int D, int H) { return scratch_elems(B, N, D, H); }

int gcgcn_gcn_bwd(int B, int N, int D, int L, int H, const float* X, const float* Ebar, const float* A,
                  const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const void* out_rng_snap,
                  float out_p, const float* Pn, const float* Y, const float* HO, const float* rinv, const float* wsum_fwd,
                  const float* dout, float* dX, float* dEbar,
                  float* dA, float* dflat, float* W1, float* W2, float* W3, float* drow, float* dXres, float* dout_m,
                  float* scratch, const gcgcn_edge_ride* ride, const gcgcn_mha_hook* mha, void* defer_queue, const int32_t* rowblk,
                  void* stream) {
  DeferQueue* dq = (DeferQueue*)defer_queue;
  MhaPass mp;
  if (mha) {
    GC_REQUIRE(gcgcn_maggc_fusable(N, D, H) && mha->Q && mha->P && mha->dQ && mha_core_ok(N, D, H, mha->Q, mha->dQ),
               "gcn_bwd: attention hook on a shape it does not serve");
    mp.Q = mha->Q, mp.P = mha->P, mp.dA = dA, mp.dQ = mha->dQ;
    mp.N = N, mp.D = D, mp.H = H, mp.dh = D / H, mp.kchunk = (D / H < 128) ? (D / H + 31) / 32 * 32 : 128, mp.count = B * H;
    mp.alpha = 1.f / sqrtf((float)(D / H));
    mp.drop = make_drop(mha->rng_snap, GCGCN_SALT_MHA, mha->p);
  }
  const Drop odrop = make_drop(out_rng_snap, GCGCN_SALT_GLUE, out_p);
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(check_dims("gcn_bwd", B, N, D, L, H));
  EdgeRide er;
  GC_TRY(make_ride("gcn_bwd", ride, 2, er));
  const long wse = scratch ? gemm_scratch_elems(B, N, D, H) : 0;
  GC_REQUIRE(X && Ebar && A && flat && Pn && Y && HO && rinv && dout && dX && dEbar && dA && dflat && W1 && W2 && W3 &&
                 drow && dXres,
             "gcn_bwd: null pointer");
  GC_REQUIRE(!(n_valid || odrop.snap) || dout_m, "gcn_bwd: n_valid / output dropout given without dout_m workspace");
  const GcnLayout y = gcn_layout(D, L, H);
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GCN, p);
  const long M = (long)B * N;
  const int gh = y.gh;
  const long HD = (long)H * D;
  float* dYa = W1;  // dHO, then the running gradient of the relu outputs Y (same layout)
  float* dM = W2;   // gradient of M_l = G_l + A_h Pn_l  (== dG)
  float* dP = W3;   // gradient of Pn_l

  // The chain kernel of the default shape computes dHO = dout Wlin and dXres = sum_h dHO_h itself (chain.hip): no launch for
  // that product, no head-sum / dropout kernel, dHO never in HBM.  Decided on the same context the chain will get.
  GcnCtx c = make_ctx(B, N, D, L, H, y, X, A, flat, n_valid, drop);
  c.Pn = const_cast<float*>(Pn), c.Y = const_cast<float*>(Y), c.rinv = const_cast<float*>(rinv);
  c.dYa = dYa, c.dM = dM, c.dP = dP, c.dA = dA, c.drow = drow, c.oWlin = y.oWlin;
  // ragged batch with a row-block list: the products around the chain run on the rows that exist (see gcgcn_gcn_fwd)
  const bool chain = use_chain_for(N, false);
  {
    EdgeRide er_chain = er;
    if (er.kind && !(chain && chain_can_carry(er))) er_chain.kind = 0;
    c.ride = er_chain;
  }
  const int* rows = (rowblk && n_valid && N % 16 == 0 && N >= 32) ? rowblk : nullptr;
  const bool fuse = chain && scratch && chain_bwd_fusable(c) && (((uintptr_t)dXres) & 15) == 0 &&
                    (((uintptr_t)dout) & 15) == 0 && (((uintptr_t)dout_m) & 15) == 0 && (long)M * HD >= (long)D * D;
  // sum_h Wlin_h: from the forward call if it left one, else summed here into dYa's buffer (free when the chain computes dHO)
  const float* wsum = (fuse && H > 1) ? (wsum_fwd ? wsum_fwd : dYa) : nullptr;
  const float* dout_raw = dout;
  if (fuse) {  // the chain masks / un-drops dout while staging it (and writes dout_m back for dWlin); only sum_h Wlin_h is left
    if (wsum && !wsum_fwd) GC_TRY(mask_rows(nullptr, nullptr, M, D, N, nullptr, odrop, st, flat + y.oWlin, dYa, H));
    if (n_valid || odrop.snap) dout = dout_m;
  } else if (n_valid || odrop.snap) {  // gradients arriving on padding rows are ignored; back through the output dropout
    GC_TRY(mask_rows(dout, dout_m, M, D, N, n_valid, odrop, st));
    dout = dout_m;
  }
  ColRide cr;
  bool col_later = false, col_pending = false;
  if (fuse) {  // dWlin = dout^T HO is parked or joins the launch after the chain; dblin's column sums ride there too
    c.dout = dout_raw, c.dXres = dXres, c.Wsum = wsum, c.oWlin = y.oWlin;
    if (dout != dout_raw) c.dout_m = dout_m, c.odrop = odrop;   // c.n_valid is set
    cr.X = dout, cr.out = dflat + y.oblin, cr.part = scratch + wse, cr.R = M, cr.ld = D, cr.C = D, col_pending = true;
    if (2 * B <= COL_RIDE_SLICES) c.colpart = cr.part, cr.ready_slices = 2 * B;  // stage 1 inside the chain (it holds dout_b in LDS)
  } else {
    // Small blocks (launch-bound: cfg 1, the reference's own model) fold the head-sum / dropout-backward kernel into this
    // launch: dY = dropout_bwd(dHO) is the product's own epilogue (the forward mask: same site, same element offsets), and the
    // residual gradient dXres = sum_h dHO_h = dout (sum_h Wlin_h) is one more small product (one head: dHO itself, written
    // beside dY).  At cfg 3 the extra product and the second store cost more than the launch they save (round 2: +8 us).
    const long fold_opt = option("head_sum_fold", 1);   // 0 off, 1 up to 2 M elements of dHO, n > 1: up to n M elements (A/B)
    const bool fold_hs = fold_opt != 0 && scratch && (H == 1 || wsum_fwd) && (long)M * HD <= (fold_opt > 1 ? fold_opt : 2) * (1L << 20) &&
                         (((uintptr_t)dXres) & 15) == 0;
    {  // one launch: dHO = dout Wlin  and  dWlin = dout^T HO
      GemmArgs gs[3];
      gs[0].ws = gs[1].ws = gs[2].ws = scratch, gs[0].ws_elems = gs[1].ws_elems = gs[2].ws_elems = wse;
      gs[0].A = dout, gs[0].lda = D, gs[0].a_kc = 1;
      gs[0].B = flat + y.oWlin, gs[0].ldb = HD, gs[0].b_kc = 0;
      gs[0].C = dYa, gs[0].ldc = HD;
      gs[0].M = (int)M, gs[0].N = (int)HD, gs[0].K = D;
      gs[1].A = dout, gs[1].lda = D, gs[1].a_kc = 0;
      gs[1].B = HO, gs[1].ldb = HD, gs[1].b_kc = 0;
      gs[1].C = dflat + y.oWlin, gs[1].ldc = HD;
      gs[1].M = D, gs[1].N = (int)HD, gs[1].K = (int)M;
      use_rows(gs[0], rows, 1, 1), use_rows(gs[1], rows, 2);
      int np1 = 2;
      if (fold_hs) {
        if (H == 1) {
          gs[0].C = dXres, gs[0].ldc = D;                                  // HD == D: the head sum is dHO
          gs[0].C2 = dYa, gs[0].ldc2 = HD, gs[0].drop = drop, gs[0].drop_base = 0;
        } else {
          if (drop.snap) gs[0].C2 = dYa, gs[0].ldc2 = HD, gs[0].drop = drop, gs[0].drop_base = 0;   // over C: the dropped value stays
          gs[2].A = dout, gs[2].lda = D, gs[2].a_kc = 1;
          gs[2].B = wsum_fwd, gs[2].ldb = D, gs[2].b_kc = 0;
          gs[2].C = dXres, gs[2].ldc = D;
          gs[2].M = (int)M, gs[2].N = D, gs[2].K = D;
          use_rows(gs[2], rows, 1, 1);
          np1 = 3;
        }
      }
      // a weight gradient nobody needs before the end of backward: parked for a later launch with idle matrix pipes
      const bool parked = gemm_defer(dq, gs[1]);
      GemmArgs run[3];
      int nr = 0;
      run[nr++] = gs[0];
      if (!parked) run[nr++] = gs[1];
      if (np1 == 3) run[nr++] = gs[2];
      if (scratch) {  // dblin = column sums of dout ride in this launch (stage 1) and in its reduce or the next kernel (stage 2)
        cr.X = dout, cr.out = dflat + y.oblin, cr.part = scratch + wse, cr.R = M, cr.ld = D, cr.C = D;
        GC_TRY(gemm_group(run, nr, st, &cr, &col_later));
      } else {
        GC_TRY(gemm_group(run, nr, st));
        GC_TRY(colsum(dout, nullptr, dflat + y.oblin, M, D, D, 1, 0, 0, 0, 0, scratch, st));
      }
    }
    if (fold_hs) {  // stage 2 of the bias column sums (if this launch had no reduce pass to do it) joins the launch after the chain
      col_pending = col_later;
      if (col_later) cr.ready_slices = COL_RIDE_SLICES;
    } else {
      GC_TRY(head_sum_drop_bwd(dYa, dYa, dXres, M, H, D, drop, st, col_later ? &cr : nullptr));  // residual + dropout backward
    }
  }

  {  // the dependent per-(doc, head) sequence, last sub-layer first
    if (er.kind && !c.ride.kind) {
      GC_TRY(edge_bcast(er.in, er.n_valid, er.out, er.B, er.N, er.D, st));
      er.kind = 0;
    }
    if (chain) {
      GC_TRY(gcn_chain_bwd(c, st, dq));  // + parked weight gradients of earlier blocks where the chain leaves room
    } else {
      for (int l = L - 1; l >= 0; --l) {
        GC_TRY(relu_norm_bwd(dYa, Y, rinv, dM, drow, M, N, H, L, gh, l, l == L - 1, st));
        GC_TRY(gemm(plan_bwd_dP(c, l), st, 0, 1));
        GC_TRY(gemm(plan_bwd_dA(c, l), st, 0, 1));
        if (l > 0) GC_TRY(gemm(plan_bwd_dY(c, l), st, 0, 1));
      }
    }
  }
  {  // one launch for every product that only needs the finished dPn / dM:
     //   dWnX = X^T dPn, dWe = Ebar^T dM, dX = dPn WnX^T + sum_h dHO_h, dEbar = dM We^T,
     //   dWd_{h,l} = [Y_0 .. Y_{l-1}]_h^T dPn_{h,l}  (l >= 1, batched over heads)
    constexpr int GMAX = 16;
    GemmArgs gs[GMAX];
    int n = 0;
    auto next = [&]() -> GemmArgs& {
      GemmArgs& g = gs[n++];
      g = GemmArgs();
      g.ws = scratch, g.ws_elems = wse;
      return g;
    };
    auto park = [&]() {  // the problem just described is a weight gradient: park it if asked to (and possible)
      if (gemm_defer(dq, gs[n - 1])) --n;
    };
    if (fuse) {  // dWlin = dout^T HO
      GemmArgs& g = next();
      g.A = dout, g.lda = D, g.a_kc = 0;
      g.B = HO, g.ldb = HD, g.b_kc = 0;
      g.C = dflat + y.oWlin, g.ldc = HD;
      g.M = D, g.N = (int)HD, g.K = (int)M;
      use_rows(g, rows, 2);
      park();
    }
    {
      GemmArgs& g = next();
      g.A = X, g.lda = D, g.a_kc = 0;
      g.B = dP, g.ldb = HD, g.b_kc = 0;
      g.C = dflat + y.oWnX, g.ldc = HD;
      g.M = D, g.N = (int)HD, g.K = (int)M;
      use_rows(g, rows, 2);
      park();
    }
    {
      GemmArgs& g = next();
      g.A = Ebar, g.lda = D, g.a_kc = 0;
      g.B = dM, g.ldb = HD, g.b_kc = 0;
      g.C = dflat + y.oWe, g.ldc = HD;
      g.M = D, g.N = (int)HD, g.K = (int)M;
      use_rows(g, rows, 2);
      park();
    }
    {
      GemmArgs& g = next();
      g.A = dP, g.lda = HD, g.a_kc = 1;
      g.B = flat + y.oWnX, g.ldb = HD, g.b_kc = 1;
      g.C = dX, g.ldc = D;
      g.M = (int)M, g.N = D, g.K = (int)HD;
      g.add = dXres, g.ldadd = D;
      use_rows(g, rows, 1, 1);   // the gradients that leave the block: zero on padding rows
    }
    {
      GemmArgs& g = next();
      g.A = dM, g.lda = HD, g.a_kc = 1;
      g.B = flat + y.oWe, g.ldb = HD, g.b_kc = 1;
      g.C = dEbar, g.ldc = D;
      g.M = (int)M, g.N = D, g.K = (int)HD;
      use_rows(g, rows, 1, 1);
    }
    for (int l = 1; l < L; ++l) {
      if (n == GMAX) {  // many sub-layers and nothing parked: launch what has been described so far
        GC_TRY(gemm_group(gs, n, st));
        n = 0;
      }
      GemmArgs& g = next();
      g.A = Y, g.lda = HD, g.a_kc = 0, g.sA2 = (long)L * gh;
      g.B = dP + (long)l * gh, g.ldb = HD, g.b_kc = 0, g.sB2 = (long)L * gh;
      g.C = dflat + y.wd_off(0, l), g.ldc = gh, g.sC2 = y.wd_head;
      g.M = l * gh, g.N = gh, g.K = (int)M;
      g.batch2 = H;
      use_rows(g, rows, 2);
      park();
    }
    // + the attention core's backward: as passenger workgroups of this launch where a (document, head) pair's scratch fits
    // the tile kernel's LDS (head width <= 32), as a launch of its own in front of it otherwise
    const bool mha_rides = mha && gemm_group_can_carry_mha(D / H) && option("mha_ride", 1) != 0;
    if (mha_rides) mp.kchunk = gemm_group_mha_chunk(D / H);
    if (mha && !mha_rides)
      GC_TRY(mha_core_bwd(mp.Q, mp.P, mp.dA, mp.dQ, B, N, D, H, mp.alpha, mp.drop, st));
    GC_TRY(gemm_group(gs, n, st, col_pending ? &cr : nullptr, nullptr, mha_rides ? &mp : nullptr));
  }
  return 0;
}

int gcgcn_maggc_fusable(int N, int D, int H) {
  if (N < 1 || N > 64 || H < 1 || D % H != 0) return 0;
  const int dh = D / H;
  // (the backward core rides in a group launch only for head widths up to 32; wider heads keep that launch, and still gain
  // the query projection inside the forward group launch and the forward core inside the chain workgroups)
  return use_mha_core() && dh % 4 == 0 && D % 4 == 0 && option("maggc_fuse", 1) != 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// GraphConv (the leaf layer, glove:18-50): out = (Ebar We + A X Wn (+ bias)) / rowsum(A)
// ---------------------------------------------------------------------------------------------
int gcgcn_graphconv_fwd(int B, int N, int Din, int De, int Dout, const float* X, const float* Ebar, const float* A,
                        const float* We, const float* Wn, const float* bias, float* out, float* T, float* rinv,
                        float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_REQUIRE(B > 0 && N > 0 && Din > 0 && De > 0 && Dout > 0, "graphconv_fwd: bad shape");
  GC_REQUIRE(X && Ebar && A && We && Wn && out && T && rinv, "graphconv_fwd: null pointer");
  const long M = (long)B * N;
  const long wse = scratch ? scratch_elems(B, N, Dout > Din ? Dout : Din, 1) : 0;
  GC_TRY(rowsum_inv(A, rinv, M, N, st));
  {  // one launch: out <- Ebar We (edge term),  T <- X Wn
    GemmArgs gs[2];
    gs[0].ws = gs[1].ws = scratch, gs[0].ws_elems = gs[1].ws_elems = wse;
    gs[0].A = Ebar, gs[0].lda = De, gs[0].a_kc = 1, gs[0].B = We, gs[0].ldb = Dout, gs[0].b_kc = 0;
    gs[0].C = out, gs[0].ldc = Dout, gs[0].M = (int)M, gs[0].N = Dout, gs[0].K = De;
    gs[1].A = X, gs[1].lda = Din, gs[1].a_kc = 1, gs[1].B = Wn, gs[1].ldb = Dout, gs[1].b_kc = 0;
    gs[1].C = T, gs[1].ldc = Dout, gs[1].M = (int)M, gs[1].N = Dout, gs[1].K = Din;
    GC_TRY(gemm_group(gs, 2, st));
  }
  {  // out = (out + A T + bias) * rinv        (glove:42-50; the bias joins before the division, glove:45-46)
    GemmArgs g;
    g.A = A, g.lda = N, g.a_kc = 1, g.sA1 = (long)N * N;
    g.B = T, g.ldb = Dout, g.b_kc = 0, g.sB1 = (long)N * Dout;
    g.C = out, g.ldc = Dout, g.sC1 = (long)N * Dout;
    g.M = N, g.N = Dout, g.K = N, g.batch1 = B;
    g.add = out, g.ldadd = Dout, g.sAdd1 = (long)N * Dout;
    g.bias = bias;
    g.rowscale = rinv, g.sRs1 = N;
    GC_TRY(gemm(g, st, 0, 1));
  }
  return 0;
}

int gcgcn_graphconv_bwd(int B, int N, int Din, int De, int Dout, const float* X, const float* Ebar, const float* A,
                        const float* We, const float* Wn, const float* out, const float* T, const float* rinv,
                        const float* dout, float* dX, float* dEbar, float* dA, float* dWe, float* dWn, float* dbias,
                        float* dS, float* dT, float* drow, float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_REQUIRE(B > 0 && N > 0 && Din > 0 && De > 0 && Dout > 0, "graphconv_bwd: bad shape");
  GC_REQUIRE(X && Ebar && A && We && Wn && out && T && rinv && dout && dX && dEbar && dA && dWe && dWn && dS && dT &&
                 drow,
             "graphconv_bwd: null pointer");
  const long M = (long)B * N;
  const long wse = scratch ? scratch_elems(B, N, Dout > Din ? Dout : Din, 1) : 0;
  // dS = dout * rinv ;  drow = -rinv * sum_c dout * out      (no relu in the leaf)
  GC_TRY(relu_norm_bwd(dout, out, rinv, dS, drow, M, N, 1, 1, Dout, 0, 1, st, 0));
  if (dbias) GC_TRY(colsum(dS, nullptr, dbias, M, Dout, Dout, 1, 0, 0, 0, 0, scratch, st));
  {  // dT = A^T dS
    GemmArgs g;
    g.A = A, g.lda = N, g.a_kc = 0, g.sA1 = (long)N * N;
    g.B = dS, g.ldb = Dout, g.b_kc = 0, g.sB1 = (long)N * Dout;
    g.C = dT, g.ldc = Dout, g.sC1 = (long)N * Dout;
    g.M = N, g.N = Dout, g.K = N, g.batch1 = B;
    GC_TRY(gemm(g, st, 0, 1));
  }
  GemmArgs gs[5];
  for (auto& g : gs) g.ws = scratch, g.ws_elems = wse;
  // dA = dS T^T + drow (broadcast along j)
  gs[0].A = dS, gs[0].lda = Dout, gs[0].a_kc = 1, gs[0].sA1 = (long)N * Dout;
  gs[0].B = T, gs[0].ldb = Dout, gs[0].b_kc = 1, gs[0].sB1 = (long)N * Dout;
  gs[0].C = dA, gs[0].ldc = N, gs[0].sC1 = (long)N * N;
  gs[0].M = N, gs[0].N = N, gs[0].K = Dout, gs[0].batch1 = B;
  gs[0].rowadd = drow, gs[0].sRa1 = N;
  // dWe = Ebar^T dS
  gs[1].A = Ebar, gs[1].lda = De, gs[1].a_kc = 0, gs[1].B = dS, gs[1].ldb = Dout, gs[1].b_kc = 0;
  gs[1].C = dWe, gs[1].ldc = Dout, gs[1].M = De, gs[1].N = Dout, gs[1].K = (int)M;
  // dEbar = dS We^T
  gs[2].A = dS, gs[2].lda = Dout, gs[2].a_kc = 1, gs[2].B = We, gs[2].ldb = Dout, gs[2].b_kc = 1;
  gs[2].C = dEbar, gs[2].ldc = De, gs[2].M = (int)M, gs[2].N = De, gs[2].K = Dout;
  // dWn = X^T dT
  gs[3].A = X, gs[3].lda = Din, gs[3].a_kc = 0, gs[3].B = dT, gs[3].ldb = Dout, gs[3].b_kc = 0;
  gs[3].C = dWn, gs[3].ldc = Dout, gs[3].M = Din, gs[3].N = Dout, gs[3].K = (int)M;
  // dX = dT Wn^T
  gs[4].A = dT, gs[4].lda = Dout, gs[4].a_kc = 1, gs[4].B = Wn, gs[4].ldb = Dout, gs[4].b_kc = 1;
  gs[4].C = dX, gs[4].ldc = Din, gs[4].M = (int)M, gs[4].N = Din, gs[4].K = Dout;
  GC_TRY(gemm_group(gs, 5, st));
  return 0;
}

// The placement of tile passengers among the rows of a carrying launch (Spread, common.hpp), evaluated on the host with the
// very function the kernels call: for every workgroup index x of a launch of n_tiles + n_others workgroups, kind[x] = 1 and
// ordinal[x] = the tile's number, or kind[x] = 0 and ordinal[x] = the row's number.  Exposed for tests (no GPU needed).
int gcgcn_debug_spread(int64_t n_tiles, int64_t n_others, int64_t cohort, int64_t pct, int32_t* kind, int32_t* ordinal) {
  GC_REQUIRE(n_tiles >= 0 && n_others >= 0 && kind && ordinal, "debug_spread: bad arguments");
  const Spread sp = make_spread(n_tiles, n_others, cohort, pct);
  for (long x = 0; x < n_tiles + n_others; ++x) {
    int idx = -1;
    kind[x] = spread_pick((int)x, sp, idx) ? 1 : 0;
    ordinal[x] = idx;
  }
  return 0;
}

int gcgcn_gemm(int M, int N, int K, const float* A, int64_t lda, int a_kc, const float* B, int64_t ldb, int b_kc,
               float* C, int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, const float* bias,
               int relu, int accumulate, int tile, int splits, float* ws, int64_t ws_elems, void* stream) {
  GemmArgs g;
  g.A = A, g.lda = lda, g.a_kc = a_kc;
  g.B = B, g.ldb = ldb, g.b_kc = b_kc;
  g.C = C, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.batch1 = 1, g.batch2 = batch;
  g.sA2 = sA, g.sB2 = sB, g.sC2 = sC;
  g.alpha = alpha, g.bias = bias, g.relu = relu, g.accumulate = accumulate;
  g.ws = ws, g.ws_elems = ws_elems;
  return gemm(g, (hipStream_t)stream, tile, splits);
}


int gcgcn_gemm_dyn(int M, int N, int K, const float* A, int64_t lda, int a_kc, const float* B, int64_t ldb, int b_kc, float* C,
                   int64_t ldc, const float* bias, int accumulate, const int32_t* count, int dyn, int64_t cap, float* ws,
                   int64_t ws_elems, void* stream) {
  GemmArgs g;
  g.A = A, g.lda = lda, g.a_kc = a_kc;
  g.B = B, g.ldb = ldb, g.b_kc = b_kc;
  g.C = C, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.bias = bias, g.accumulate = accumulate;
  g.ws = ws, g.ws_elems = ws_elems;
  return gemm_dyn(g, count, dyn, cap, (hipStream_t)stream);
}

}  // extern "C"
