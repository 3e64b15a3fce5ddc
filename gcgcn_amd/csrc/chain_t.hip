// LDS-resident chain kernels for every graph of up to 64 entities, any feature width gh in {32, 64, 128, 192, 256} and any
// number of sub-layers the templates are instantiated for -- the generalisation of chain.hip's gcn_chain_s_* (which serve
// one shape: 64 entities, two sub-layers of 128 features) to the reference's own model (hidden 128: gh = 64, L = 2, N <= 42
// ragged; glove:234, 250-251), to cfg 3 (bert-sized: gh = 192, L = 4) and to anything in between.
//
// One workgroup per (document, head), as before, but the work is cut differently:
//
//  * COLUMN STRIPS.  Wave w owns columns [16 w, 16 w + 16) of every gh-wide tensor of its pair, for ALL (up to 64) rows:
//    four 16 x 16 accumulators of v_mfma_f32_16x16x4_f32.  A workgroup is gh / 16 waves (4 ... 16), so every SIMD of the
//    compute unit hosts the same number of waves for every width.
//  * CHAINED PRODUCTS.  The aggregation  A_h Pn_l  contracts over ROWS of Pn_l, and a wave holds all rows of its columns:
//    the accumulator registers of the product that made Pn_l ARE the B operand of the aggregation (accumulator element v of
//    lane (j, g) is row 16 kb + 4 g + v, exactly the k index lane group g supplies at MFMA step v when the A operand is read
//    as one 16-byte LDS word per four steps).  No store, no barrier, no reload between the two products; the same holds for
//    dPn_l = A_h^T dM_l in backward.
//  * PUSH ORDER.  Dense connections are pushed, not pulled: as soon as Y_l exists it is accumulated into the Pn of every
//    later sub-layer (Pn_l' += Y_l Wd_{l',l}; backward: dY_l' += dPn_l Wd_{l,l'}^T for the earlier ones), whose accumulators
//    stay in registers.  Only ONE 64 x gh image (Y_l / dPn_l) has to be in LDS at a time, whatever L is -- the history that
//    does not fit the 160 KB at gh = 192, L = 4 is never needed.
//  * WEIGHTS THROUGH LDS, COALESCED.  Wd is streamed in 16-deep k chunks by all threads (one 16-byte load each per chunk,
//    contiguous in memory), double-buffered, one LDS-only barrier per chunk.  (Per-lane B-operand loads straight from L2
//    run at ~20 GB/s per compute unit -- DESIGN.md section 6 -- a tenth of what this pattern gets.)
//
// Reference: GraphConv.forward glove:36-50 inside the dense loops of GraphConvolution.forward glove:70-76 /
// MultiGraphConvolution.forward glove:102-113, and their autograd.
#include <type_traits>

#include "edge_body.hpp"
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

// Trace build (-DGC_T_TRACE, tools/trace_chain.py): workgroup 0 stamps the 100 MHz wall clock at phase boundaries (forward:
// slots 0..63, backward: 100.. for the MAGGC launch, 164.. for CAGGC's); compiled out of the product build.
#ifdef GC_T_TRACE
__device__ long long gc_trace_t[256];
#define TR(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gc_trace_t[i] = wall_clock64(); } while (0)
#define TRB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gc_trace_t[(i) + (c.H == 1 ? 64 : 0)] = wall_clock64(); } while (0)
extern "C" int gcgcn_debug_trace_t(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_trace_t), sizeof(long long) * 256); }
#else
#define TR(i)
#define TRB(i)
#endif

typedef float t4 __attribute__((ext_vector_type(4)));

constexpr int T_LA = 68;  // row pitch of the 64 x 64 adjacency image (16-byte rows, conflict-free 16-byte reads)

__device__ __forceinline__ void t_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ t4 mfma16(float a, float b, t4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// sum over the 16 lanes of a DPP row (= the 16 columns a lane group holds of one accumulator row); every lane gets it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}

template <int GH, int L>
constexpr int t_fwd_lds() { return 64 * T_LA + 64 * (GH + 4) + 2 * (L - 1) * 16 * (GH + 4) + 64; }
template <int GH>
constexpr int t_bwd_lds() { return 64 * T_LA + 2 * 64 * (GH + 4) + (GH / 16) * 64 + 128; }

// ---------------------------------------------------------------------------------------------------------------------
// forward:  rinv = 1 / rowsum(A_h);  for l:  Y_l = relu((G_l + A_h Pn_l) rinv),  HO_l = dropout(Y_l) + X_l,
//           Pn_l' += Y_l Wd_{l'}[l gh : (l + 1) gh, :]  for l' > l   (Pn_l' starts as X Wn_l'[:D], written by the launch before)
// ---------------------------------------------------------------------------------------------------------------------
// FULL: every document fills all four 16-row blocks (N > 48, no n_valid): the row-block loops carry no branches, so the
// LDS reads of one block overlap the MFMAs of another.  Otherwise blocks beyond ceil(n_valid / 16) are skipped (uniform
// branches): a DocRED batch padded to 42 entities averages 20 real ones, two blocks instead of three.
template <int GH, int L, bool FULL>
__global__ __launch_bounds__(4 * GH) void gcn_chain_t_fwd_kernel(const GcnCtx c) {
  constexpr int W = GH / 16, NT = 4 * GH, P = GH + 4, NC = GH / 16;
  __shared__ __attribute__((aligned(16))) float lds[t_fwd_lds<GH, L>()];
  if (blockIdx.x >= c.B * c.H) {  // passenger workgroup: one entity row of the riding edge mean
    const EdgeRide& r = c.ride;
    edge_fwd_row<4, false, true, W>(r.in, nullptr, r.n_valid, r.out, nullptr, nullptr, nullptr, Drop(), r.N, r.D,
                                    blockIdx.x - c.B * c.H, lds);
    return;
  }
  TR(0);
  float* const As = lds;
  float* const Ys = As + 64 * T_LA;
  float* const Ws = Ys + 64 * P;                    // [2 stages][L - 1 pending sub-layers][16 k][P]
  float* const Rs = Ws + 2 * (L - 1) * 16 * P;
  const int z = blockIdx.x, b = z / c.H, h = z - b * c.H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 15, g = lane >> 4;
  const int N = c.N;
  const int nv = c.n_valid ? min(max(c.n_valid[b], 0), N) : N;
  const int nrb = FULL ? 4 : (nv + 15) >> 4;        // 16-row blocks that hold real entities (padding rows are zero everywhere)
  const unsigned HD = (unsigned)c.HD, D = (unsigned)c.D;
  const long zoff = (long)b * N * c.HD + (long)h * c.D;  // (b, row 0, h, l = 0, k = 0) in [B*N, H, L, gh]
  const float* __restrict__ Ag = c.A + (long)z * N * N;
  const float* __restrict__ Pg = c.Pn + zoff;
  const float* __restrict__ Gg = c.G + zoff;
  const float* __restrict__ Xg = c.X + (long)b * N * c.D;
  float* __restrict__ Yg = c.Y + zoff;
  float* __restrict__ Hg = c.HO + zoff;
  float* __restrict__ Pw = c.Pn + zoff;
  const int col = 16 * w + j;
  const bool dd = c.drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(c.drop) : 0;

  // ---- the adjacency image (zero beyond N) and the row normaliser (glove:47-49) -----------------------------------------
  if (c.mha.Q) {
    // MultiHeadAttention's core for this pair runs here (mha_body.hpp; scratch: the Y image and the weight stages, all free
    // until the first sub-layer): P / A to global memory for backward, the adjacency the chain uses straight into As
    for (int idx = t; idx < (64 - N) * 16; idx += NT) {
      const int row = N + (idx >> 4), c4 = (idx & 15) * 4;
      *reinterpret_cast<t4*>(As + row * T_LA + c4) = t4{0.f, 0.f, 0.f, 0.f};
    }
    mha_core_fwd_body<true>(Ys, z, c.mha.Q, c.n_valid, c.mha.P, c.mha.A, N, c.D, c.H, c.mha.dh, c.mha.kchunk, c.mha.alpha, c.mha.drop, t,
                            t < 256, As, T_LA, W);
    t_barrier();   // (LDS only: the adjacency image is complete; nobody waits for the P / A stores)
    for (int i = w; i < 64; i += W) {
      const float s = wave_sum(i < N ? As[i * T_LA + lane] : 0.f);
      if (lane == 0) {
        const float ri = i < N ? 1.f / (s + (s == 0.f ? 1.f : 0.f)) : 0.f;
        Rs[i] = ri;
        if (i < N) c.rinv[(long)z * N + i] = ri;
      }
    }
  } else {
    const bool v4 = (N & 3) == 0;
    for (int idx = t; idx < 64 * 16; idx += NT) {
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      t4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < N && c4 < N) {
        if (v4) {
          v = *reinterpret_cast<const t4*>(Ag + row * N + c4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c4 + e < N) v[e] = Ag[row * N + c4 + e];
        }
      }
      *reinterpret_cast<t4*>(As + row * T_LA + c4) = v;
    }
    for (int i = w; i < 64; i += W) {
      const float s = wave_sum((i < N && lane < N) ? Ag[i * N + lane] : 0.f);
      if (lane == 0) {
        const float ri = i < N ? 1.f / (s + (s == 0.f ? 1.f : 0.f)) : 0.f;
        Rs[i] = ri;
        if (i < N) c.rinv[(long)z * N + i] = ri;
      }
    }
  }
  TR(1);
  // ---- Pn of every sub-layer: this wave's columns, all rows, in accumulator layout ---------------------------------------
  t4 Pa[L][4];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        // (row blocks past the real entities' are neither read nor computed; everything this kernel leaves there is zero)
        Pa[l][rb][v] = (row < N && rb < nrb) ? Pg[(unsigned)row * HD + (unsigned)(l * GH + col)] : 0.f;
      }
  t_barrier();
  TR(2);

  auto layer = [&](auto lt) __attribute__((always_inline)) {
    constexpr int l = decltype(lt)::value;
    constexpr int NP = L - 1 - l;                     // sub-layers still waiting for this one's output
    TR(10 + 8 * l);
    // requests first: the epilogue's operands and the first two weight chunks land while the aggregation runs
    float gv[4][4], xv[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        gv[rb][v] = (row < N && rb < nrb) ? Gg[(unsigned)row * HD + (unsigned)(l * GH + col)] : 0.f;
        xv[rb][v] = (row < N && rb < nrb) ? Xg[(unsigned)row * D + (unsigned)(l * GH + col)] : 0.f;
      }
    t4 wr[2][NP > 0 ? NP : 1];
    const float* __restrict__ Wb = c.flat + c.oWd + (long)h * c.wd_head + (long)l * GH * GH + 4 * t;   // + wd_off(l') below
    auto gload = [&](const int ch, t4 (&d)[NP > 0 ? NP : 1]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const long off = (long)GH * GH * (l + 1 + p) * (l + p) / 2;   // wd_off(l + 1 + p) - oWd
        d[p] = *reinterpret_cast<const t4*>(Wb + off + (long)ch * 16 * GH);
      }
    };
    auto sstore = [&](const int st, const t4 (&d)[NP > 0 ? NP : 1]) __attribute__((always_inline)) {
      const int kr = t / (GH / 4), n4 = (t - kr * (GH / 4)) * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<t4*>(Ws + ((st * (L - 1) + p) * 16 + kr) * P + n4) = d[p];
    };
    if constexpr (NP > 0) {
      gload(0, wr[0]);
      gload(1, wr[1]);
    }
    // ---- aggregation, chained: B operand = the Pn accumulators themselves ---------------------------------------------
    t4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      acc[ob] = t4{0.f, 0.f, 0.f, 0.f};
      if (ob < nrb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (kb < nrb) {
            const t4 a = *reinterpret_cast<const t4*>(As + (16 * ob + j) * T_LA + 16 * kb + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[ob] = mfma16(a[v], Pa[l][kb][v], acc[ob]);
          }
        }
      }
    }
    TR(11 + 8 * l);
    // ---- Y_l = relu((G_l + A_h Pn_l) rinv);  HO_l = dropout(Y_l) + X_l          (glove:42-50, 71-76) ----------------------
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        const unsigned o = (unsigned)row * HD + (unsigned)(l * GH + col);
        const float y = fmaxf((acc[rb][v] + gv[rb][v]) * Rs[row], 0.f);
        if constexpr (NP > 0) Ys[row * P + col] = y;
        if (row < N) {
          Yg[o] = y;
          float d = y;
          if (dd) d = (rng_u32(key, (uint64_t)(zoff + (long)o)) >= c.drop.thresh) ? y * c.drop.scale : 0.f;
          Hg[o] = d + xv[rb][v];
        }
      }
    if constexpr (NP > 0) {
      // ---- push: Pn_l' += Y_l Wd_l'[l gh : (l + 1) gh, :] for every later sub-layer, 16 k per chunk --------------------
      TR(12 + 8 * l);
      sstore(0, wr[0]);
      t_barrier();   // Y_l's image and the first weight chunk are complete
      TR(13 + 8 * l);
      auto compute = [&](const int ch, const int st) __attribute__((always_inline)) {
        t4 a[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          if (rb < nrb) a[rb] = *reinterpret_cast<const t4*>(Ys + (16 * rb + j) * P + 16 * ch + 4 * g);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float* wb = Ws + ((st * (L - 1) + p) * 16 + 4 * g) * P + col;
          float bv[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) bv[v] = wb[v * P];
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
              if (rb < nrb) Pa[l + 1 + p][rb] = mfma16(a[rb][v], bv[v], Pa[l + 1 + p][rb]);
        }
      };
      for (int ch = 0; ch < NC; ch += 2) {   // two chunks per trip: register sets and stages are compile-time constants
        if (ch + 2 < NC) gload(ch + 2, wr[0]);
        compute(ch, 0);
        sstore(1, wr[1]);
        t_barrier();
        if (ch + 3 < NC) gload(ch + 3, wr[1]);
        compute(ch + 1, 1);
        if (ch + 2 < NC) sstore(0, wr[0]);
        t_barrier();
      }
      TR(14 + 8 * l);
      // Pn_{l+1} is complete: saved for backward
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * rb + 4 * g + v;
          if (row < N) Pw[(unsigned)row * HD + (unsigned)((l + 1) * GH + col)] = Pa[l + 1][rb][v];
        }
    }
  };
  static_assert(L >= 1 && L <= 4, "sub-layers are unrolled by hand");
  layer(std::integral_constant<int, 0>());
  if constexpr (L > 1) layer(std::integral_constant<int, 1>());
  if constexpr (L > 2) layer(std::integral_constant<int, 2>());
  if constexpr (L > 3) layer(std::integral_constant<int, 3>());
  TR(50);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, last sub-layer first:
//   dS = dY_l [Y_l > 0];  dM_l = dS rinv;  drow -= rinv sum_c dS Y_l;  dPn_l = A_h^T dM_l;  dA += dM_l Pn_l^T;
//   dY_l' += dPn_l Wd_l[l' gh : (l' + 1) gh, :]^T for l' < l   (dY_l' starts as dropout_bwd(dHO_l'), read from dYa)
// dA: wave w accumulates rows 16 (w % 4) .. + 15, all 64 columns, over the k range [64 (w / 4), + 64) of every sub-layer in
// registers; the gh / 64 partial sums meet in LDS at the end, in a fixed order (bitwise reproducible).  gh = 32 (two waves: the
// BERT model's width, hidden 128 over four sub-layers, bert:237,247-248): each wave takes two row blocks over the whole k range.
// ---------------------------------------------------------------------------------------------------------------------
// Parked weight-gradient tiles (gemm.hpp DeferQueue) as passengers of a chain launch that leaves compute units idle: a
// passenger workgroup of NTEAM x 256 threads runs NTEAM consecutive 64 x 64 tiles of ONE problem side by side, one per team,
// each over its whole K (all teams of a workgroup pass the same number of barriers: same problem, same K; a team beyond the
// problem's last tile recomputes that tile without storing).  Workgroup pb -> problem by the prefix sums of ceil(take / NTEAM).
constexpr int T_TEAM_LDS = lds_floats<1, 1, true, true>();
template <int NTEAM, bool RB>
__device__ __forceinline__ void t_parked_tiles(const GemmGroup4& cg, int pb, float* __restrict__ lds) {
  int i = 0, w = pb;
  while (i + 1 < cg.nprob && w >= (cg.tile_take[i] + NTEAM - 1) / NTEAM) {
    w -= (cg.tile_take[i] + NTEAM - 1) / NTEAM;
    ++i;
  }
  const int team = threadIdx.x >> 8, t = threadIdx.x & 255;
  int q = NTEAM * w + team;
  const bool live = q < cg.tile_take[i];
  if (!live) q = cg.tile_take[i] - 1;
  q = xcd_remap(q + cg.tile_first[i], cg.tile_count[i]);
  const GemmArgs& g = cg.p[i];
  const int tn = g.N >> 6, tm = g.M >> 6;
  const int zs = q / (tn * tm), r = q - zs * (tn * tm);
  const int bx = (tm < tn) ? r / tm : r % tn, by = (tm < tn) ? r % tm : r / tn;   // same tile list as gemm_group_block
  float* tl = lds + team * T_TEAM_LDS;
  // (weight gradients: K-side problems, whose tile list does not depend on the row blocks)
  if (g.a_kc) {
    if (g.b_kc) gemm_body<1, 1, true, true, true>(g, tl, bx, by, zs, t, live);
    else gemm_body<1, 1, true, false, true>(g, tl, bx, by, zs, t, live);
  } else {
    if (g.b_kc) gemm_body<1, 1, false, true, true>(g, tl, bx, by, zs, t, live);
    else gemm_body<1, 1, false, false, true, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, tl, bx, by, zs, t, live);
  }
}

template <int GH, int L, bool FULL>
__global__ __launch_bounds__(4 * GH) void gcn_chain_t_bwd_kernel(const GcnCtx c, const GemmGroup4 cg, const int npw) {
  constexpr int W = GH / 16, NT = 4 * GH, P = GH + 4, NC = GH / 16, SP = 20;   // SP: row pitch of a [gh][16 k] weight stage
  __shared__ __attribute__((aligned(16))) float lds[t_bwd_lds<GH>()];
  static_assert(2 * GH * SP <= 64 * P, "weight stages live in the Pn image");
  static_assert(W <= 4 || (W / 4 - 1) * 4096 <= 64 * P, "dA exchange lives in the dM image");
  static_assert((GH / 64) * T_TEAM_LDS <= t_bwd_lds<GH>(), "parked tiles use the chain kernel's LDS");
  constexpr int OBW = W >= 4 ? 1 : 4 / W;          // row blocks of dA per wave (fewer than four waves: several each)
  constexpr int KS = GH >= 64 ? 4 : GH / 16;       // 16-deep k steps of a wave's k range (64 features, or all of a narrow sub-layer)
  if (blockIdx.x >= c.B * c.H) {
    int pb;
    if (spread_pick((int)blockIdx.x - c.B * c.H, c.carry, pb)) {  // passenger workgroup: GH / 64 tiles of a parked weight-gradient product
      // (the host hands tiles to 256-thread teams only; a ragged launch -- !FULL -- may carry products on row blocks)
      if constexpr (GH >= 64) t_parked_tiles<GH / 64, !FULL>(cg, pb, lds);
      return;
    }
    const EdgeRide& r = c.ride;  // passenger workgroup: one entity row of the riding dE broadcast
    edge_bcast_row<4, W>(r.in, r.n_valid, r.out, r.N, r.D, 0, pb);
    return;
  }
  TRB(100);
  float* const ATs = lds;                  // A_h transposed: [k = column of A][row of A]
  float* const Ds = ATs + 64 * T_LA;       // dM_l, then dPn_l
  float* const Ps = Ds + 64 * P;           // Pn_l, then the weight stages
  float* const Tp = Ps + 64 * P;           // [W][64] per-wave partial row sums
  float* const Ts = Tp + W * 64;           // gradient of the normaliser's row sums
  float* const Rs = Ts + 64;               // rinv
  const int z = blockIdx.x, b = z / c.H, h = z - b * c.H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 15, g = lane >> 4;
  const int N = c.N;
  const int nv = c.n_valid ? min(max(c.n_valid[b], 0), N) : N;
  const int nrb = FULL ? 4 : (nv + 15) >> 4;
  const unsigned HD = (unsigned)c.HD;
  const long zoff = (long)b * N * c.HD + (long)h * c.D;
  const float* __restrict__ Ag = c.A + (long)z * N * N;
  const float* __restrict__ Pg = c.Pn + zoff;
  const float* __restrict__ Yg = c.Y + zoff;
  const float* __restrict__ Gy = c.dYa + zoff;
  float* __restrict__ Mg = c.dM + zoff;
  float* __restrict__ Qg = c.dP + zoff;
  const int col = 16 * w + j;
  const int ob = w & 3, ks = w >> 2;       // dA: this wave's (first) row block and k range; further blocks: ob + W u

  {  // A_h^T image (zero beyond N), rinv
    const bool v4 = (N & 3) == 0;
    for (int idx = t; idx < 64 * 16; idx += NT) {
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      t4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < N && c4 < N) {
        if (v4) {
          v = *reinterpret_cast<const t4*>(Ag + row * N + c4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c4 + e < N) v[e] = Ag[row * N + c4 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ATs[(c4 + e) * T_LA + row] = v[e];
    }
    if (t < 64) Rs[t] = t < N ? c.rinv[(long)z * N + t] : 0.f, Ts[t] = 0.f;
  }
  t4 Da[L > 1 ? L - 1 : 1][4];             // dY_l' contributions pushed by later sub-layers (l' = 0 .. L - 2)
#pragma unroll
  for (int l = 0; l < L - 1; ++l)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) Da[l][rb] = t4{0.f, 0.f, 0.f, 0.f};
  t4 dacc[OBW][4];
#pragma unroll
  for (int u = 0; u < OBW; ++u)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) dacc[u][jb] = t4{0.f, 0.f, 0.f, 0.f};
  constexpr int PV = (64 * (GH / 4)) / NT;  // 16-byte pieces of a 64 x gh image per thread (= 4)
  static_assert(PV * NT == 64 * (GH / 4), "image load mapping");

  auto layer = [&](auto lt) __attribute__((always_inline)) {
    constexpr int l = decltype(lt)::value;
    TRB(110 + 8 * l);
    // ---- requests: Y_l, dY_l (this wave's strip) and the Pn_l image ------------------------------------------------------
    float yv[4][4], dy[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        const unsigned o = (unsigned)row * HD + (unsigned)(l * GH + col);
        yv[rb][v] = (row < N && rb < nrb) ? Yg[o] : 0.f;      // (dYa's rows past the real entities' blocks may never have been written)
        dy[rb][v] = (row < N && rb < nrb) ? Gy[o] : 0.f;
      }
    t4 pn[PV];
#pragma unroll
    for (int u = 0; u < PV; ++u) {
      const int idx = t + NT * u, row = idx / (GH / 4), c4 = (idx - row * (GH / 4)) * 4;
      pn[u] = (row < N && (row >> 4) < nrb) ? *reinterpret_cast<const t4*>(Pg + (unsigned)row * HD + (unsigned)(l * GH + c4)) : t4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- through Y = relu(S), S = M rinv:  dS = dY [Y > 0];  dM = dS rinv;  drow -= rinv sum_c dS Y --------------------
    t4 dm[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        float gsel = dy[rb][v];
        if constexpr (l < L - 1) gsel += Da[l][rb][v];
        gsel = yv[rb][v] > 0.f ? gsel : 0.f;
        const float m = gsel * Rs[row];
        dm[rb][v] = m;
        Ds[row * P + col] = m;
        if (row < N) Mg[(unsigned)row * HD + (unsigned)(l * GH + col)] = m;
        const float part = row16_sum(gsel * yv[rb][v]);
        // Ragged instantiations: every lane of the row holds the sum and stores it.  A store under `j == 0` is the one lane-divergent
        // region of this kernel, and with it <192, 4, false> -- 190 spilled registers -- returned a different dA on every run for
        // documents of three row blocks once a change elsewhere in the file had moved its register allocation (round 4; the
        // instantiation spills more without the branch and is deterministic).  The FULL instantiations keep the branch (cfg 3:
        // +2 % step time without it); tests/test_hip_parity.py::test_chain_t_is_deterministic scans every instantiation.
        if (!FULL || j == 0) Tp[w * 64 + row] = part;
      }
#pragma unroll
    for (int u = 0; u < PV; ++u) {
      const int idx = t + NT * u, row = idx / (GH / 4), c4 = (idx - row * (GH / 4)) * 4;
      *reinterpret_cast<t4*>(Ps + row * P + c4) = pn[u];
    }
    TRB(111 + 8 * l);
    t_barrier();   // dM_l, Pn_l images and the row-sum partials are complete
    TRB(112 + 8 * l);
    if (t < 64) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) s += Tp[ww * 64 + t];
      Ts[t] -= Rs[t] * s;
    }
    // ---- dPn_l = A_h^T dM_l, chained: B operand = dM_l's registers --------------------------------------------------------
    t4 q[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      q[rb] = t4{0.f, 0.f, 0.f, 0.f};
      if (rb < nrb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (kb < nrb) {
            const t4 a = *reinterpret_cast<const t4*>(ATs + (16 * rb + j) * T_LA + 16 * kb + 4 * g);
#pragma unroll
            for (int v = 0; v < 4; ++v) q[rb] = mfma16(a[v], dm[kb][v], q[rb]);
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * rb + 4 * g + v;
        if (row < N) Qg[(unsigned)row * HD + (unsigned)(l * GH + col)] = q[rb][v];
      }
    }
    TRB(113 + 8 * l);
    // first weight chunks of the push below: requested now, they land while dA's share runs
    constexpr int NQ = l * NC;                         // 16-deep chunks of the push: (l', chunk) flattened
    t4 wr[2];
    const float* __restrict__ Wl = c.flat + c.wd_off(l) + (long)h * c.wd_head;   // Wd_l: [l gh rows (l', n')][gh]
    auto gload = [&](const int qi, t4& d) __attribute__((always_inline)) {
      const int lp = qi / NC, ch = qi - lp * NC;
      d = *reinterpret_cast<const t4*>(Wl + ((long)lp * GH + (t >> 2)) * GH + 16 * ch + 4 * (t & 3));
    };
    auto sstore = [&](const int st, const t4& d) __attribute__((always_inline)) {
      *reinterpret_cast<t4*>(Ps + st * GH * SP + (t >> 2) * SP + 4 * (t & 3)) = d;
    };
    if constexpr (l > 0) {
      gload(0, wr[0]);
      gload(1, wr[1]);
    }
    // ---- dA += dM_l Pn_l^T: rows 16 ob .. + 15, k range [64 ks, 64 ks + 64) ------------------------------------------------
#pragma unroll
    for (int u = 0; u < OBW; ++u) {
      const int obu = ob + W * u;
      if (obu < nrb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const t4 a = *reinterpret_cast<const t4*>(Ds + (16 * obu + j) * P + 64 * ks + 16 * s + 4 * g);
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) {
            if (jb < nrb) {
              const t4 bq = *reinterpret_cast<const t4*>(Ps + (16 * jb + j) * P + 64 * ks + 16 * s + 4 * g);
#pragma unroll
              for (int v = 0; v < 4; ++v) dacc[u][jb] = mfma16(a[v], bq[v], dacc[u][jb]);
            }
          }
        }
      }
    }
    TRB(114 + 8 * l);
    if constexpr (l > 0) {
      t_barrier();   // everybody is done with the dM_l and Pn_l images
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) Ds[(16 * rb + 4 * g + v) * P + col] = q[rb][v];
      sstore(0, wr[0]);
      t_barrier();   // dPn_l's image and the first weight chunk are complete
      TRB(115 + 8 * l);
      // ---- push: dY_l' += dPn_l Wd_l[l' gh + n', k]^T for l' < l ---------------------------------------------------------
      auto compute = [&](const int qi, const int st) __attribute__((always_inline)) {
        const int lp = qi / NC, ch = qi - lp * NC;
        t4 a[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          if (rb < nrb) a[rb] = *reinterpret_cast<const t4*>(Ds + (16 * rb + j) * P + 16 * ch + 4 * g);
        const t4 bq = *reinterpret_cast<const t4*>(Ps + st * GH * SP + col * SP + 4 * g);
#pragma unroll
        for (int p = 0; p < l; ++p) {
          if (p == lp) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
              for (int rb = 0; rb < 4; ++rb)
                if (rb < nrb) Da[p][rb] = mfma16(a[rb][v], bq[v], Da[p][rb]);
          }
        }
      };
      for (int qi = 0; qi < NQ; qi += 2) {
        if (qi + 2 < NQ) gload(qi + 2, wr[0]);
        compute(qi, 0);
        sstore(1, wr[1]);
        t_barrier();
        if (qi + 3 < NQ) gload(qi + 3, wr[1]);
        compute(qi + 1, 1);
        if (qi + 2 < NQ) sstore(0, wr[0]);
        t_barrier();
      }
    }
  };
  t_barrier();
  static_assert(L >= 1 && L <= 4, "sub-layers are unrolled by hand");
  if constexpr (L > 3) layer(std::integral_constant<int, 3>());
  if constexpr (L > 2) layer(std::integral_constant<int, 2>());
  if constexpr (L > 1) layer(std::integral_constant<int, 1>());
  layer(std::integral_constant<int, 0>());
  TRB(150);
  // ---- dA = sum over the k ranges + drow (every column of a row); drow itself ---------------------------------------------
  t_barrier();   // sub-layer 0 is done with the images; Ts is final
  if (ks > 0 && ob < nrb) {      // (only with more than four waves: OBW == 1)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int v = 0; v < 4; ++v) Ds[((((ks - 1) * 4 + ob) * 4 + jb) * 4 + v) * 64 + lane] = dacc[0][jb][v];
  }
  t_barrier();
  float* __restrict__ dAg = c.dA + (long)z * N * N;
#pragma unroll
  for (int u = 0; u < OBW; ++u) {
    const int obu = ob + W * u;
    if (ks == 0 && obu < nrb) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * obu + 4 * g + v, cc = 16 * jb + j;
          float s = dacc[u][jb][v];
#pragma unroll
          for (int k2 = 1; k2 < W / 4; ++k2) s += Ds[((((k2 - 1) * 4 + obu) * 4 + jb) * 4 + v) * 64 + lane];
          if (row < N && cc < N) dAg[row * N + cc] = (jb < nrb ? s : 0.f) + Ts[row];
        }
    }
    // rows of A beyond the real entities' blocks: dA = drow there (no product contributes), written by the wave that would own them
    if (ks == 0 && obu >= nrb && obu < 4) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * obu + 4 * g + v, cc = 16 * jb + j;
          if (row < N && cc < N) dAg[row * N + cc] = Ts[row];
        }
    }
  }
  if (t < N) c.drow[(long)z * N + t] = Ts[t];
  TRB(151);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// Ragged batches at the widest four-sub-layer shape (cfg 3's) run the FULL instantiation too: no row skipping, but 29 spilled
// registers instead of 353 (cfg 3 ragged 1.437 -> 1.417 ms; option chain_t_wide_full = 0: the ragged instantiation, A/B and tests).
// Padding rows hold zeros in every input, as for gcn_chain_s_*.
static bool chain_t_full(const GcnCtx& c) {
  if (c.N > 48 && !c.n_valid) return true;
  return c.N > 48 && c.gh == 192 && c.L == 4 && option("chain_t_wide_full", 1) != 0;
}
template <int GH, int L>
static void launch_fwd(const GcnCtx& c, dim3 grid, double fl, hipStream_t st) {
  if (chain_t_full(c)) GC_LAUNCH_TIMED("gcn_chain_fwd", fl, (gcn_chain_t_fwd_kernel<GH, L, true>), grid, dim3(4 * GH), 0, st, c);
  else GC_LAUNCH_TIMED("gcn_chain_fwd", fl, (gcn_chain_t_fwd_kernel<GH, L, false>), grid, dim3(4 * GH), 0, st, c);
}
template <int GH, int L>
static void launch_bwd(const GcnCtx& c, const GemmGroup4& cg, int npw, dim3 grid, double fl, hipStream_t st) {
  if (chain_t_full(c)) GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, true>), grid, dim3(4 * GH), 0, st, c, cg, npw);
  else GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, false>), grid, dim3(4 * GH), 0, st, c, cg, npw);
}

// (gh, L) pairs the templates are instantiated for: the reference's model (64, 2), cfg 2's width (128, 2), cfg 3 (192, 4),
// and the neighbours a user is most likely to configure
#define GC_CHAIN_T_SHAPES(X) X(32, 2) X(32, 4) X(64, 1) X(64, 2) X(64, 3) X(64, 4) X(128, 1) X(128, 2) X(128, 3) X(128, 4) X(192, 2) X(192, 4) X(256, 1) X(256, 2)

static int chain_t_waves(int gh) { return gh / 16; }

// 0 = not served by these kernels (the generic chain kernels take it)
bool chain_t_ok(const GcnCtx& c, bool bwd) {
  const int mode = option("chain_t", 1);
  if (!mode || c.N > 64 || c.N < 1) return false;
  bool shape = false;
#define X(gh_, l_) shape = shape || (c.gh == gh_ && c.L == l_);
  GC_CHAIN_T_SHAPES(X)
#undef X
  if (!shape) return false;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = al(c.A) && al(c.flat + c.oWd) && c.wd_head % 4 == 0 && c.HD % 4 == 0 && al(c.Pn) && al(c.Y);
  if (bwd) ok = ok && !c.dout && al(c.dYa) && al(c.dM) && al(c.dP);
  else ok = ok && al(c.G) && al(c.HO) && al(c.X);
  if (c.ride.kind) {   // the passenger bodies use 16-byte accesses and (forward) waves x D floats of the kernel's LDS
    const int lds_fwd = 64 * T_LA + 64 * (c.gh + 4) + 2 * (c.L - 1) * 16 * (c.gh + 4) + 64;
    ok = ok && c.ride.D % 4 == 0 && al(c.ride.in) && al(c.ride.out) && (bwd || (long)chain_t_waves(c.gh) * c.ride.D <= lds_fwd);
  }
  return ok;
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
#define X(gh_, l_)                                \
  if (c.gh == gh_ && c.L == l_) {                 \
    launch_fwd<gh_, l_>(c, grid, fl, st);         \
    return check_launch("gcn_chain_t_fwd");       \
  }
  GC_CHAIN_T_SHAPES(X)
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
      kmax = carry->p[i].K > kmax ? carry->p[i].K : kmax;
    }
    // A tile runs its whole K inside one workgroup (~1.1 us per 32-deep k-step, slower with several teams on the unit): it
    // only pays where the chain itself runs that long.  Chain: MFMAs per wave x 32 cycles x waves per SIMD at ~0.55 of the
    // matrix pipe (cfg 3: 138 us estimated, 132 measured), plus the riding dE broadcast at ~5 TB/s.
    const double mfma = c.L * 128.0 + (c.L * (c.L - 1) / 2) * (double)c.gh;
    double t_chain = mfma * 32.0 * (c.gh / 64) / 2100.0 / 0.55;
    if (c.ride.kind == 2) t_chain += 4.0 * c.ride.B * c.ride.N * (double)c.ride.N * c.ride.D / 5.0e6;
    const double t_tile = 1.1 * (kmax / 32.0) * (1.0 + 0.3 * (nteam - 1));
    ok = ok && t_tile <= 1.25 * t_chain;
    long wgs = idle * option("chain_carry_rounds", 1);
    if (c.ride.kind == 2) wgs = wgs * option("chain_carry_pct", 100) / 100;
    if (ok && wgs > 0) {
      gemm_take_deferred_pairs(carry, cg, &fl, wgs * nteam, true);   // whole small problems first
      for (int i = 0; i < cg.nprob; ++i) npw += (cg.tile_take[i] + nteam - 1) / nteam;
    }
  }
  const dim3 grid((unsigned)(c.B * c.H + npw) + (c.ride.kind == 2 ? (unsigned)(c.ride.B * c.ride.N) : 0u));
  GcnCtx cc = c;
  cc.carry = chain_carry_spread(c, npw);
#define X(gh_, l_)                                         \
  if (c.gh == gh_ && c.L == l_) {                          \
    launch_bwd<gh_, l_>(cc, cg, npw, grid, fl, st);        \
    return check_launch("gcn_chain_t_bwd");                \
  }
  GC_CHAIN_T_SHAPES(X)
#undef X
  set_error("gcn_chain_t_bwd: shape gh=%d L=%d not instantiated", c.gh, c.L);
  return 1;
}

}  // namespace gc
