// Host side of the column-strip chain kernels: which shapes they serve, how many parked weight-gradient tiles a backward
// launch takes along, and the dispatch to the translation unit that holds a shape's instantiations (chain_t.hpp explains the
// kernels; chain_t_u0 .. u3.hip instantiate them, four units so that the build compiles them side by side).
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int T_LA = 68;   // (chain_t.hpp) row pitch of the adjacency image: part of the forward kernel's LDS size below

// (gh, L) pairs with an instantiation -- the same list as GC_CHAIN_T_SHAPES in chain_t.hpp
#define GC_CHAIN_T_HOST_SHAPES(X) X(32, 2) X(32, 4) X(64, 1) X(64, 2) X(64, 3) X(64, 4) X(128, 1) X(128, 2) X(128, 3) X(128, 4) X(192, 2) X(192, 4) X(256, 1) X(256, 2)

#define GC_T_UNITS(X) X(0) X(1) X(2) X(3)
#define X(u) \
  int chain_t_fwd_unit##u(const GcnCtx& c, dim3 grid, double fl, hipStream_t st, int* rc); \
  int chain_t_bwd_unit##u(const GcnCtx& c, const GemmGroup4& cg, int npw, dim3 grid, double fl, hipStream_t st, int* rc);
GC_T_UNITS(X)
#undef X

static int chain_t_waves(int gh) { return gh / 16; }

// 0 = not served by these kernels (the generic chain kernels take it)
bool chain_t_ok(const GcnCtx& c, bool bwd) {
  const int mode = option("chain_t", 1);
  if (!mode || c.N > 64 || c.N < 1) return false;
  bool shape = false;
#define X(gh_, l_) shape = shape || (c.gh == gh_ && c.L == l_);
  GC_CHAIN_T_HOST_SHAPES(X)
#undef X
  if (!shape) return false;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = al(c.A) && al(c.flat + c.oWd) && c.wd_head % 4 == 0 && c.HD % 4 == 0 && al(c.Pn) && al(c.Y);
  if (bwd) ok = ok && (c.dout ? chain_t_bwd_fusable(c) && al(c.dout) && al(c.dXres) && (!c.dout_m || al(c.dout_m)) : al(c.dYa)) && al(c.dM) && al(c.dP);
  else ok = ok && al(c.G) && al(c.HO) && al(c.X);
  if (c.ride.kind) {   // the passenger bodies use 16-byte accesses and (forward) waves x D floats of the kernel's LDS
    const int lds_fwd = 64 * T_LA + 64 * (c.gh + 4) + 2 * (c.L - 1) * 16 * (c.gh + 4) + 64;
    ok = ok && c.ride.D % 4 == 0 && al(c.ride.in) && al(c.ride.out) && (bwd || (long)chain_t_waves(c.gh) * c.ride.D <= lds_fwd);
  }
  return ok;
}

// The backward kernel can compute the output projection's input gradient itself (chain_t.hpp FUSE): blocks of at most 256
// features whose residual-gradient share divides over the waves (one head: no such product); option chain_fuse = 0: never
bool chain_t_bwd_fusable(const GcnCtx& c) {
  if (option("chain_fuse", 1) == 0 || c.N > 64 || c.D != c.L * c.gh || c.D > 256 || c.D % 32 != 0 || c.gh > 128 || 16 % c.L != 0) return false;
  bool shape = false;
#define X(gh_, l_) shape = shape || (c.gh == gh_ && c.L == l_);
  GC_CHAIN_T_HOST_SHAPES(X)
#undef X
  if (!shape) return false;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (!al(c.flat + c.oWlin) || c.HD % 4 != 0) return false;
  if (c.H == 1) return true;
  const int W = c.gh / 16;
  if (c.D % c.H != 0) return false;
  const int DH = c.D / c.H;
  if (DH % 16 != 0) return false;
  const int ncg = DH / 16;
  if (ncg > W || W % ncg != 0) return false;
  const int kw = W / ncg;
  return c.D % kw == 0 && (c.D / kw) % 16 == 0;
}

// the forward kernel can run the attention core in its prologue: the core's scratch (score tile + one Q chunk) fits the Y image
// and the weight stages
bool chain_t_fwd_att_ok(const GcnCtx& c) {
  const int dh = c.D / c.H;
  const long room = 64L * (c.gh + 4) + 2L * (c.L - 1) * 16 * (c.gh + 4);
  // (the core's body runs on the first four waves of the workgroup: widths of at least 64)
  return c.N <= 64 && c.gh >= 64 && dh % 4 == 0 && (long)(mha_lds_bytes(dh) / sizeof(float)) <= room;
}

int gcn_chain_t_fwd(const GcnCtx& c, dim3 grid, double fl, hipStream_t st) {
  int rc = 0;
#define X(u) \
  if (chain_t_fwd_unit##u(c, grid, fl, st, &rc)) return rc;
  GC_T_UNITS(X)
#undef X
  set_error("gcn_chain_t_fwd: shape gh=%d L=%d not instantiated", c.gh, c.L);
  return 1;
}

// carry: parked weight-gradient products; as many of their tiles ride as fit on the compute units this launch leaves idle
// in one round (a tile's whole K takes about as long as the chain itself), half of them when the dE broadcast rides as well
int gcn_chain_t_bwd(const GcnCtx& c, double fl, hipStream_t st, DeferQueue* carry) {
  GemmGroup4 cg;
  cg.nprob = 0, cg.tile_begin[0] = 0;
  const int nteam = c.gh / 64;
  int npw = 0;
  const long idle = 256 - (long)c.B * c.H;
  if (carry && carry->n > 0 && idle > 0 && nteam > 0 && option("chain_carry", 1) != 0) {   // (nteam == 0: a 128-thread workgroup hosts no tile team)
    bool ok = true;
    int kmax = 0;
    for (int i = 0; i < carry->n; ++i) {
      ok = ok && carry->p[i].K % BK == 0 && carry->p[i].splits <= 1;
      ok = ok && !(carry->p[i].rb && (chain_t_full(c) || carry->p[i].rb_mode != 2));   // row-block products: the ragged instantiations only
      // (a row-block product walks the live blocks only -- 43 % of K at DocRED's entity counts; the host never reads n_valid and
      // prices it at half its K.  At full price no tile fitted beside cfg 2's ragged CAGGC chain, whose 224 idle compute units then
      // waited out the launch while the last edge pass ran all 740 tiles: chain 43.6 -> 51.2 us with 304 of them aboard, edge pass
      // 62.4 -> 48.8; ragged cfg 2 79.4 k -> 80.8 k docs/s, `profiles/r05_ab_chain_t_rb_pct.txt`.)
      const int keff = carry->p[i].rb ? carry->p[i].K / 2 : carry->p[i].K;
      kmax = keff > kmax ? keff : kmax;
    }
    // A tile runs its whole K inside one workgroup (~1.1 us per 32-deep k-step, slower with several teams on the unit): it
    // only pays where the chain itself runs that long.  Chain: MFMAs per wave x 32 cycles x waves per SIMD at ~0.55 of the
    // matrix pipe (cfg 3: 138 us estimated, 132 measured), plus the riding dE broadcast at ~5 TB/s.
    const double mfma = c.L * 128.0 + (c.L * (c.L - 1) / 2) * (double)c.gh;
    double t_chain = mfma * 32.0 * (c.gh / 64) / 2100.0 / 0.55;
    if (c.ride.kind == 2) t_chain += 4.0 * c.ride.B * c.ride.N * (double)c.ride.N * c.ride.D / 5.0e6;
    const double t_tile = 1.1 * (kmax / 32.0) * (1.0 + 0.3 * (nteam - 1));
    ok = ok && t_tile <= 1.25 * t_chain;
    const long wgs = idle;
    if (ok && wgs > 0) {
      gemm_take_deferred_pairs(carry, cg, &fl, wgs * nteam, true);   // whole small problems first
      for (int i = 0; i < cg.nprob; ++i) npw += (cg.tile_take[i] + nteam - 1) / nteam;
    }
  }
  const dim3 grid((unsigned)(c.B * c.H + npw) + (c.ride.kind == 2 ? (unsigned)(c.ride.B * c.ride.N) : 0u));
  GcnCtx cc = c;
  cc.carry = chain_carry_spread(c, npw);
  int rc = 0;
#define X(u) \
  if (chain_t_bwd_unit##u(cc, cg, npw, grid, fl, st, &rc)) return rc;
  GC_T_UNITS(X)
#undef X
  set_error("gcn_chain_t_bwd: shape gh=%d L=%d not instantiated", c.gh, c.L);
  return 1;
}

}  // namespace gc
