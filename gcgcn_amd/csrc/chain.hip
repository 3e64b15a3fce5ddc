// Chain kernels of the densely connected GraphConv stack.
//
// Inside one GraphConvolution / MultiGraphConvolution every (document b, head h) pair runs its own
// dependent sequence of small products (gcn_plan.hpp): per sub-layer  Pn_l += Y_<l Wd_l,  Y_l =
// relu((G_l + A_h Pn_l) rinv)  forward, and the mirrored four steps backward.  Launched one by
// one these are 2L + 4L latency-bound kernels of 64x64x64-sized work; here ONE persistent
// workgroup per (b, h) runs the whole sequence with the GEMM tile body, handing intermediate
// tiles to itself through global memory (L2-hot) across __syncthreads() -- all waves of a
// workgroup share the CU's vector L1, so no cache maintenance is needed at workgroup scope.
// Pairs never touch each other's slices, so there is no inter-workgroup synchronisation at all
// and every wave reaches the end of the phase list (no spin, no flag).
#include <stdlib.h>

#include "edge_body.hpp"
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

__device__ __forceinline__ bool dev_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// A chain workgroup is 512 threads = two 256-thread tile teams with an LDS image each.  The teams walk the
// 64x64 tiles of a product side by side (team 0 the even ones, team 1 the odd ones), so every SIMD hosts two
// waves whose instruction and memory latencies overlap -- with one workgroup per CU nothing else would.
// Both teams execute the same number of barriers (same K); when the tile count is odd, team 1 recomputes the
// last tile without storing it.
constexpr int TEAM_LDS = lds_floats<1, 1, true, true>();
constexpr int CHAIN_LDS = 2 * TEAM_LDS;
constexpr int XCHG_LDS = 64 * 64;  // passenger tiles of parked products: the two teams' partial sums meet here
constexpr int CW = 8;   // waves per chain workgroup
constexpr int RB = 8;   // rows a wave keeps in flight in the row-wise phases (8 waves x 8 rows: a 64-node graph in one pass)

template <bool AKC, bool BKC, bool ALIGNED, int MASK, int RT>
__device__ __forceinline__ void run_product(GemmArgs g, float* lds, int z) {
  g.ksplit = g.K;
  g.splits = 1;
  if (!ALIGNED) {
    g.vecA = dev_al16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
    g.vecB = dev_al16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
  }
  const int team = threadIdx.x >> 8, t = threadIdx.x & 255;
  float* tl = lds + team * TEAM_LDS;
  const int tn = (g.N + 63) >> 6, tiles = ((g.M + 63) >> 6) * tn;
  for (int p = 0; p < tiles; p += 2) {
    int q = p + team;
    const bool live = q < tiles;
    if (!live) q = tiles - 1;
    gemm_body<1, 1, AKC, BKC, ALIGNED, 16, MASK, RT>(g, tl, q % tn, q / tn, z, t, live);
  }
}

// Two INDEPENDENT products side by side: team 0 runs every tile of g1, team 1 every tile of g2 (dPn_l = A^T dM_l and
// dA += dM_l Pn_l^T both wait only for dM_l).  A tile of the GEMM body passes 1 + ceil(K / 32) workgroup barriers; the
// team with fewer of them in total pads the difference, so both teams arrive at every barrier.
template <bool AKC1, bool BKC1, int MK1, int RTA, bool AKC2, bool BKC2, int MK2, int RTB, bool ALIGNED>
__device__ __forceinline__ void run_two_products(GemmArgs g1, GemmArgs g2, float* lds, int z) {
  auto prep = [](GemmArgs& g) {
    g.ksplit = g.K, g.splits = 1;
    if (!ALIGNED) {
      g.vecA = dev_al16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
      g.vecB = dev_al16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
    }
  };
  prep(g1), prep(g2);
  const int team = threadIdx.x >> 8, t = threadIdx.x & 255;
  float* tl = lds + team * TEAM_LDS;
  const int tn1 = (g1.N + 63) >> 6, tiles1 = ((g1.M + 63) >> 6) * tn1, b1 = tiles1 * (1 + (g1.K + 31) / 32);
  const int tn2 = (g2.N + 63) >> 6, tiles2 = ((g2.M + 63) >> 6) * tn2, b2 = tiles2 * (1 + (g2.K + 31) / 32);
  if (team == 0) {
    for (int q = 0; q < tiles1; ++q) gemm_body<1, 1, AKC1, BKC1, ALIGNED, 16, MK1, RTA>(g1, tl, q % tn1, q / tn1, z, t, true);
    for (int i = b1; i < b2; ++i) __syncthreads();
  } else {
    for (int q = 0; q < tiles2; ++q) gemm_body<1, 1, AKC2, BKC2, ALIGNED, 16, MK2, RTB>(g2, tl, q % tn2, q / tn2, z, t, true);
    for (int i = b2; i < b1; ++i) __syncthreads();
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(64 * CW) void gcn_chain_fwd_kernel(const GcnCtx c) {
  __shared__ __attribute__((aligned(16))) float lds[CHAIN_LDS];
  if (blockIdx.x >= c.B * c.H) {  // passenger workgroup: one entity row of the riding edge mean
    const EdgeRide& r = c.ride;
    edge_fwd_row<4, false, true, CW>(r.in, nullptr, r.n_valid, r.out, nullptr, nullptr, nullptr, Drop(), r.N, r.D,
                                     blockIdx.x - c.B * c.H, lds);
    return;
  }
  const int z = blockIdx.x;  // b * H + h
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // GraphConv row normaliser (glove:47-49): rinv[i] = 1 / (sum_j A[i,j] + [sum == 0])
  {  // RB rows per wave in flight: the loads of a row batch are independent, one round trip per batch
    const float* a = c.A + (long)z * c.N * c.N;
    for (int i0 = wave; i0 < c.N; i0 += CW * RB) {
      float s[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) s[u] = 0.f;
      for (int j = lane; j < c.N; j += 64) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int i = i0 + CW * u;
          if (i < c.N) s[u] += a[(long)i * c.N + j];
        }
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int i = i0 + CW * u;
        const float t = wave_sum(s[u]);
        if (lane == 0 && i < c.N) c.rinv[(long)z * c.N + i] = 1.f / (t + (t == 0.f ? 1.f : 0.f));
      }
    }
  }
  __syncthreads();
  for (int l = 0; l < c.L; ++l) {
    if (l > 0) {
      run_product<true, false, ALIGNED, EPI_ACCUM, 0>(plan_fwd_dense(c, l), lds, z);
      __syncthreads();
    }
    run_product<true, false, ALIGNED, EPI_ADD | EPI_ROWSCALE | EPI_RELU | EPI_C2, 0>(plan_fwd_agg(c, l), lds, z);
    __syncthreads();
  }
}

// cg: parked weight-gradient products (gemm.hpp) carried by a chain launch that leaves most compute units idle -- its
// workgroups follow the chain workgroups in dispatch order, two 64x64 tiles each.
template <bool ALIGNED>
__global__ __launch_bounds__(64 * CW) void gcn_chain_bwd_kernel(const GcnCtx c, const GemmGroup4 cg) {
  __shared__ __attribute__((aligned(16))) float lds[CHAIN_LDS + XCHG_LDS];
  if (blockIdx.x >= c.B * c.H) {
    const int pb = blockIdx.x - c.B * c.H, ng = cg.tile_begin[cg.nprob];
    if (pb < ng) {  // passenger workgroup: one tile of a parked product, K split over the two tile teams
      gemm_group_splitk_block(cg, pb, lds, TEAM_LDS, lds + CHAIN_LDS);
      return;
    }
    const EdgeRide& r = c.ride;  // passenger workgroup: one entity row of the riding dE broadcast
    edge_bcast_row<4, CW>(r.in, r.n_valid, r.out, r.N, r.D, 0, pb - ng);
    return;
  }
  const int z = blockIdx.x;
  const int b = z / c.H, h = z - b * c.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = c.L - 1; l >= 0; --l) {
    // through Y = relu(S), S = M * rinv:  dS = dY [Y > 0];  dM = dS rinv;  drow -= rinv sum_c dS Y
    for (int i0 = wave; i0 < c.N; i0 += CW * RB) {
      float acc[RB], rv[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int i = i0 + CW * u;
        acc[u] = 0.f;
        rv[u] = (i < c.N) ? c.rinv[(long)z * c.N + i] : 0.f;
      }
      for (int k = lane; k < c.gh; k += 64) {
        float y[RB], gy[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int i = i0 + CW * u;
          const long off = ((((long)b * c.N + i) * c.H + h) * c.L + l) * c.gh + k;
          y[u] = 0.f, gy[u] = 0.f;
          if (i < c.N) y[u] = c.Y[off], gy[u] = c.dYa[off];
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int i = i0 + CW * u;
          const long off = ((((long)b * c.N + i) * c.H + h) * c.L + l) * c.gh + k;
          const float g = y[u] > 0.f ? gy[u] : 0.f;
          if (i < c.N) c.dM[off] = g * rv[u];
          acc[u] = fmaf(g, y[u], acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int i = i0 + CW * u;
        const float t = wave_sum(acc[u]);
        if (lane == 0 && i < c.N) {  // row i is always handled by this lane: plain read-modify-write is ordered
          const long ri = (long)z * c.N + i;
          const float d = -rv[u] * t;
          c.drow[ri] = (l == c.L - 1) ? d : c.drow[ri] + d;
        }
      }
    }
    __syncthreads();
    run_two_products<false, false, 0, 0, true, true, 0, EPI_ACCUM | EPI_ROWADD, ALIGNED>(plan_bwd_dP(c, l), plan_bwd_dA(c, l), lds, z);
    __syncthreads();
    if (l > 0) {
      run_product<true, true, ALIGNED, EPI_ACCUM, 0>(plan_bwd_dY(c, l), lds, z);
      __syncthreads();
    }
  }
}

static bool chain_aligned(const GcnCtx& c, bool bwd) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = c.N % 64 == 0 && c.gh % 64 == 0 && al(c.A) && al(c.flat + c.oWd) && al(c.Pn) && al(c.Y);
  if (bwd) ok = ok && al(c.dYa) && al(c.dM) && al(c.dP) && al(c.dA);
  else ok = ok && al(c.G) && al(c.HO) && al(c.X);
  return ok;
}

// The riding pass uses the 16-byte row bodies and the chain kernel's static LDS for its per-wave column sums.
bool chain_can_carry(const EdgeRide& r) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  return r.kind != 0 && r.D % 4 == 0 && al(r.in) && al(r.out) && (long)CW * r.D <= CHAIN_LDS &&
         (long)r.B * r.N <= 0x3fffffffL;
}

// GCGCN_CHAIN_CARRY=0: no parked products in the chain launches (A/B knob)
static bool chain_passengers() {
  static const int v = [] {
    const char* e = getenv("GCGCN_CHAIN_CARRY");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return v != 0;
}

static long carry_budget_pct() {  // GCGCN_CHAIN_CARRY_PCT: tuning knob, default 50
  static const long v = [] {
    const char* e = getenv("GCGCN_CHAIN_CARRY_PCT");
    return e ? atol(e) : 50L;
  }();
  return v;
}

static unsigned chain_grid(const GcnCtx& c, int kind) {
  return (unsigned)(c.B * c.H) + (c.ride.kind == kind ? (unsigned)(c.ride.B * c.ride.N) : 0u);
}

int gcn_chain_fwd(const GcnCtx& c, hipStream_t st) {
  GC_REQUIRE((long)c.B * c.H <= 0x3fffffffL, "gcn_chain_fwd: too many (doc, head) pairs");
  GC_REQUIRE(c.ride.kind == 0 || (c.ride.kind == 1 && chain_can_carry(c.ride)), "gcn_chain_fwd: bad passenger");
  dim3 grid(chain_grid(c, 1)), block(64 * CW);
  double fl = 0;
  for (int l = 0; l < c.L; ++l) fl += 2.0 * c.N * c.gh * (c.N + (double)l * c.gh);
  if (chain_aligned(c, false)) GC_LAUNCH_TIMED("gcn_chain_fwd", fl * c.B * c.H, gcn_chain_fwd_kernel<true>, grid, block, 0, st, c);
  else GC_LAUNCH_TIMED("gcn_chain_fwd", fl * c.B * c.H, gcn_chain_fwd_kernel<false>, grid, block, 0, st, c);
  return check_launch("gcn_chain_fwd");
}

int gcn_chain_bwd(const GcnCtx& c, hipStream_t st, DeferQueue* carry) {
  GC_REQUIRE(c.ride.kind == 0 || (c.ride.kind == 2 && chain_can_carry(c.ride)), "gcn_chain_bwd: bad passenger");
  double fl = 0;
  for (int l = 0; l < c.L; ++l) fl += 4.0 * c.N * c.gh * c.N + 2.0 * c.N * c.gh * (double)l * c.gh;
  fl *= (double)c.B * c.H;
  GemmGroup4 cg;
  cg.nprob = 0, cg.tile_begin[0] = 0;
  // Parked products ride only where the chain leaves the chip mostly empty (few (doc, head) pairs): a passenger workgroup
  // runs ONE unsplit-output tile, its K split over the two tile teams -- matrix-pipe-bound at ~0.47 us per 32-deep k-step
  // of the tile -- and occupies its compute unit alone (register footprint of this kernel).  As many tiles ride as fit
  // beside the chain's own duration (~8 us per dependent product and tile pass, measured at cfg 2 / cfg 3); a problem is
  // split at the budget, the rest of its tiles rides in GATAttention's edge pass.  K % 64 == 0 for equal halves.
  int ng = 0;
  if (carry && carry->n > 0 && (long)c.B * c.H <= 64 && ((long)c.B * c.N) % 64 == 0 && chain_passengers()) {
    const int passes = (((c.N + 63) / 64) * ((c.gh + 63) / 64) + 1) / 2;
    const double t_chain = 8.0 * (4 * c.L - 1) * passes;                  // us
    const double t_tile = 0.47 * ((double)c.B * c.N / 32.0);              // us: weight gradients have K = B N
    const long rounds = t_tile > 0 ? (long)(t_chain / t_tile) : 0;
    bool halves = true;
    for (int i = 0; i < carry->n; ++i) halves = halves && carry->p[i].K % 64 == 0;
    // the riding dE broadcast needs its share of the idle compute units too: with it aboard only half of them take a tile
    long budget = rounds * (256 - (long)c.B * c.H);
    if (c.ride.kind == 2) budget = carry_budget_pct() * budget / 100;
    if (rounds > 0 && halves && budget > 0) ng = gemm_take_deferred_pairs(carry, cg, &fl, budget);
  }
  dim3 grid(chain_grid(c, 2) + (unsigned)ng), block(64 * CW);
  if (chain_aligned(c, true)) GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_bwd_kernel<true>, grid, block, 0, st, c, cg);
  else GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_bwd_kernel<false>, grid, block, 0, st, c, cg);
  return check_launch("gcn_chain_bwd");
}

}  // namespace gc
