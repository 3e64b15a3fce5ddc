// Edge-feature streaming kernels: the HBM-bound half of the CAGGC/MAGGC path.
//
// E is the per-pair edge tensor [B, N, N, D] (fp32, row-major): 4*N*N*D bytes per document,
// far larger than everything else on the path.  Each byte of E is read exactly once in the
// forward pass and once in the backward pass; dE is written exactly once.
//
//   edge_fwd  : one workgroup per entity row (b, i) streams E[b, i, :, :] (N*D contiguous
//               floats, 16 B per lane, a 1-KiB row segment per wave instruction) and produces
//                 Ebar[b, i, :]  = mean_j E[b, i, j, :]          (GraphConv edge term, glove:40-41
//                                                                 after commuting mean and W_e)
//                 logit[b, i, j] = v . E[b, i, j, :]             (GATAttention edge term, glove:161-162
//                                                                 folded: v = W_r^T wt_r)
//   edge_bwd  : re-reads E for dv = sum dlogit * E and writes
//                 dE[b, i, j, :] = dlogit[b, i, j] * v + dEbar[b, i, :] / n
//   edge_bcast: dE[b, i, j, :] = dEbar[b, i, :] / n   (MAGGC hop: E only feeds the mean)
//
// Ragged batches: n_valid[b] <= N entities are real; padding rows/columns are neither read nor
// averaged, and their outputs are zero.
#include <stdlib.h>
#include <string.h>

#include "edge_body.hpp"
#include "gat_body.hpp"
#include "gemm_body.hpp"

namespace gc {

template <int VEC, bool ATT, bool NTL>
__global__ __launch_bounds__(64 * EW) void edge_fwd_kernel(const float* __restrict__ E, const float* __restrict__ v,
                                                           const int* __restrict__ n_valid, float* __restrict__ Ebar,
                                                           const float* __restrict__ coladd, float* __restrict__ P,
                                                           float* __restrict__ Aout, Drop drop, int N, int D,
                                                           const unsigned char* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) float cs[];  // [EW][D] per-wave column sums, then [N] logits
  edge_fwd_row<VEC, ATT, NTL, EW>(E, v, n_valid, Ebar, coladd, P, Aout, drop, N, D, blockIdx.x, cs, mask);
}

// ---------------------------------------------------------------------------------------------
// backward of the CAGGC hop: dE = dlogit (x) v + dEbar / n ;  dv partial per workgroup.
// dynamic LDS: N + EW * D floats.
// ---------------------------------------------------------------------------------------------
template <int VEC, bool NTL>
__device__ __forceinline__ void edge_bwd_row(const float* __restrict__ E, const float* __restrict__ v,
                                             const int* __restrict__ n_valid, const float* __restrict__ dlogit,
                                             const float* __restrict__ dEbar, float* __restrict__ dE,
                                             float* __restrict__ dvpart, int N, int D, int nt, const int bi,
                                             float* __restrict__ sm, const GatTail& gt) {
  float* dl = sm;                    // [N]
  float* cs = sm + ((N + 3) & ~3);   // [EW][D]
  const int b = bi / N, i = bi - b * N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  float* dEr = dE ? dE + (long)bi * N * D : nullptr;
  float* dvp = dvpart + (long)bi * D;
  if (i >= nv) {
    for (int c = t; c < D; c += 64 * EW) dvp[c] = 0.f;
    if (dEr) {
      const long tot = (long)N * D;
      for (long o = t; o < tot; o += 64 * EW) dEr[o] = 0.f;
    }
    return;
  }
  if (gt.P) {  // N <= 64: the row's softmax gradient straight from P and dA (dropout replayed), no dlogit tensor in between
    if (wave == 0) {
      const long o = (long)bi * N + min(lane, N - 1);
      float p = gt.P[o], g = gt.dA[o];
      if (lane >= N) p = 0.f, g = 0.f;
      if (gt.drop.snap) g = (rng_u32(drop_key(gt.drop), (uint64_t)((long)bi * N + lane)) >= gt.drop.thresh) ? g * gt.drop.scale : 0.f;
      const float dot = wave_sum(g * p);
      if (lane < N) dl[lane] = p * (g - dot);
    }
  } else {
    for (int j = t; j < N; j += 64 * EW) dl[j] = dlogit[(long)bi * N + j];
  }
  __syncthreads();
  const float* __restrict__ Er = E + (long)bi * N * D;
  const float inv = 1.f / (float)nv;
  const int nchunk = (D + 64 * VEC - 1) / (64 * VEC);
  for (int q = 0; q < nchunk; ++q) {
    const int c = (q * 64 + lane) * VEC;
    const bool act = c < D;
    float vr[VEC], g[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) vr[e] = 0.f, g[e] = 0.f, acc[e] = 0.f;
    if (act) {
      vload<VEC>(vr, v + c);
      if (dEbar) {
        vload<VEC>(g, dEbar + (long)bi * D + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] *= inv;
      }
    }
    int j = wave;
    for (; j + (EUNR - 1) * EW < nv; j += EUNR * EW) {
      float x[EUNR][VEC];
#pragma unroll
      for (int u = 0; u < EUNR; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[u][e] = 0.f;
        if (act) {
          if (NTL) vload_nt<VEC>(x[u], Er + (long)(j + u * EW) * D + c);
          else vload<VEC>(x[u], Er + (long)(j + u * EW) * D + c);
        }
      }
#pragma unroll
      for (int u = 0; u < EUNR; ++u) {
        const float d = dl[j + u * EW];
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          acc[e] = fmaf(d, x[u][e], acc[e]);
          o[e] = fmaf(d, vr[e], g[e]);
        }
        if (dEr && act) {
          if (nt) vstore_nt<VEC>(dEr + (long)(j + u * EW) * D + c, o);
          else vstore<VEC>(dEr + (long)(j + u * EW) * D + c, o);
        }
      }
    }
    for (; j < nv; j += EW) {
      float x[VEC], o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) x[e] = 0.f;
      if (act) vload<VEC>(x, Er + (long)j * D + c);
      const float d = dl[j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc[e] = fmaf(d, x[e], acc[e]);
        o[e] = fmaf(d, vr[e], g[e]);
      }
      if (dEr && act) {
        if (nt) vstore_nt<VEC>(dEr + (long)j * D + c, o);
        else vstore<VEC>(dEr + (long)j * D + c, o);
      }
    }
    if (dEr && act) {  // padding columns of a real row
      float z[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) z[e] = 0.f;
      for (int jj = nv + wave; jj < N; jj += EW) vstore<VEC>(dEr + (long)jj * D + c, z);
    }
    if (act) vstore<VEC>(cs + wave * D + c, acc);
  }
  __syncthreads();
  for (int c = t; c < D; c += 64 * EW) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < EW; ++w) s += cs[w * D + c];
    dvp[c] = s;
  }
}

template <int VEC, bool NTL>
__global__ __launch_bounds__(64 * EW) void edge_bwd_kernel(const float* __restrict__ E, const float* __restrict__ v,
                                                           const int* __restrict__ n_valid,
                                                           const float* __restrict__ dlogit,
                                                           const float* __restrict__ dEbar, float* __restrict__ dE,
                                                           float* __restrict__ dvpart, int N, int D, int nt,
                                                           const GatTail gt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // the GATAttention backward of the node scores (gat_body.hpp) rides in front: B * slices short workgroups
  const int ngat = gt.P ? gt.B * gt.slices : 0;
  if ((int)blockIdx.x < ngat) {
    const int b = blockIdx.x / gt.slices;
    gat_dlogit_doc(gt.P, gt.dA, gt.uvc, gt.dXin, nullptr, gt.ds, gt.dX, N, D, gt.drop, b, blockIdx.x - b * gt.slices, gt.slices, sm);
    return;
  }
  edge_bwd_row<VEC, NTL>(E, v, n_valid, dlogit, dEbar, dE, dvpart, N, D, nt, blockIdx.x - ngat, sm, gt);
}

// The same pass carrying deferred GEMM problems (gemm.hpp): the first gg.tile_begin[gg.nprob] workgroups each run one
// 64x64 tile of a parked weight-gradient product over its whole K -- matrix-pipe work under an HBM-bound stream --
// the rest are the entity rows.  The tiles go out in cohorts spread through the launch (Spread, common.hpp): every compute
// unit then hosts rows AND a tile for most of the launch, where tiles-first order fills the chip with tiles alone for
// ntile / 1024 rounds before the first row starts (cfg 5: 1.5 rounds, no overlap at all).
template <int VEC, bool NTL, bool RB>   // RB: a carried product runs on the row blocks of a ragged batch (GemmArgs::rb)
__global__ __launch_bounds__(64 * EW) void edge_bwd_carry_kernel(const float* __restrict__ E, const float* __restrict__ v,
                                                                 const int* __restrict__ n_valid,
                                                                 const float* __restrict__ dlogit,
                                                                 const float* __restrict__ dEbar, float* __restrict__ dE,
                                                                 float* __restrict__ dvpart, int N, int D, int nt,
                                                                 const Spread sp, const GatTail gt, const GemmGroup gg,
                                                                 const int col_base) {
  // one LDS image for both kinds of workgroup (the rows need N + EW * D floats of it): a fourth workgroup fits per CU
  __shared__ __attribute__((aligned(16))) float tile_lds[lds_floats<1, 1, true, true>()];
  int r;
  if (col_base > 0 && (int)blockIdx.x >= col_base) {   // behind everything else: the second stage of a parked column sum (gg.col)
    col_ride_stage2_block(gg.col, (int)blockIdx.x - col_base);
    return;
  }
  if (spread_pick((int)blockIdx.x, sp, r)) {
    gemm_group_block<RB>(gg, r, tile_lds);
    return;
  }
  const int ngat = gt.P ? gt.B * gt.slices : 0;
  if (r < ngat) {
    const int b = r / gt.slices;
    gat_dlogit_doc(gt.P, gt.dA, gt.uvc, gt.dXin, nullptr, gt.ds, gt.dX, N, D, gt.drop, b, r - b * gt.slices, gt.slices, tile_lds);
    return;
  }
  edge_bwd_row<VEC, NTL>(E, v, n_valid, dlogit, dEbar, dE, dvpart, N, D, nt, r - ngat, tile_lds, gt);
}

template <int VEC>
__global__ __launch_bounds__(64 * EW) void edge_bcast_kernel(const float* __restrict__ dEbar,
                                                             const int* __restrict__ n_valid, float* __restrict__ dE,
                                                             int N, int D, int nt) {
  edge_bcast_row<VEC, EW>(dEbar, n_valid, dE, N, D, nt, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// E is read once per pass and never again before something else has flushed the caches (a training step feeds new
// documents): all E loads are non-temporal.  GCGCN_NT_E1=0 restores ordinary loads for the attention pass over E1 --
// that was the round-1 setting, tuned on a bench that replayed ONE batch (part of E1 still sat in the Infinity Cache
// from the previous backward); with rotating batches ordinary loads cost 8 us in edge_fwd_att (33.6 -> 25.7 us) and
// 3 us in the backward edge pass.  GCGCN_NT_STORE=1 switches the dE stores to non-temporal too (A/B knob; no gain
// with rotating batches, slower stores with a replayed one: edge_bwd 56 -> 69 us).
static int nt_e1() {
  static const int v = [] {
    const char* e = getenv("GCGCN_NT_E1");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return v;
}
static int nt_store() {
  static const int v = [] {
    const char* e = getenv("GCGCN_NT_STORE");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return v;
}

int edge_fwd(const float* E, const float* v, const int* n_valid, float* Ebar, const float* coladd, float* P, float* A,
             Drop drop, int B, int N, int D, hipStream_t st, const unsigned char* mask) {
  GC_REQUIRE(E && Ebar, "edge_fwd: null pointer");
  GC_REQUIRE(B > 0 && N > 0 && D > 0, "edge_fwd: bad shape B=%d N=%d D=%d", B, N, D);
  const bool att = P != nullptr;
  GC_REQUIRE(!att || (v && coladd), "edge_fwd: attention requested without v / node scores");
  const bool vec = (D % 4 == 0) && al16(E) && al16(Ebar) && (!att || al16(v));
  const size_t lds = ((size_t)EW * D + (att ? (size_t)N : 0)) * sizeof(float);
  GC_REQUIRE(lds <= 160 * 1024, "edge_fwd: N=%d D=%d needs %zu B of LDS", N, D, lds);
  dim3 grid((unsigned)((long)B * N)), block(64 * EW);
  const char* tag = att ? "edge_fwd_att" : "edge_fwd_mean";
  const double bytes = 4.0 * B * N * N * D;
  // the mean-only pass reads E once per step: always non-temporal; the attention pass over E1 too unless GCGCN_NT_E1=0
  const bool ntl = !att || nt_e1();
#define GC_EDGE_FWD(V, AT, NT) \
  GC_LAUNCH_TIMED(tag, bytes, (edge_fwd_kernel<V, AT, NT>), grid, block, lds, st, E, v, n_valid, Ebar, coladd, P, A, drop, N, D, mask)
  if (vec) {
    if (att) { if (ntl) GC_EDGE_FWD(4, true, true); else GC_EDGE_FWD(4, true, false); }
    else GC_EDGE_FWD(4, false, true);
  } else {
    if (att) { if (ntl) GC_EDGE_FWD(1, true, true); else GC_EDGE_FWD(1, true, false); }
    else GC_EDGE_FWD(1, false, true);
  }
#undef GC_EDGE_FWD
  return check_launch("edge_fwd");
}

int edge_bwd(const float* E, const float* v, const int* n_valid, const float* dlogit, const float* dEbar, float* dE,
             float* dvpart, int B, int N, int D, hipStream_t st, DeferQueue* carry, const GatTail* tail) {
  GatTail gt;
  memset(&gt, 0, sizeof gt);
  if (tail) gt = *tail;
  GC_REQUIRE(E && v && (dlogit || gt.P) && dvpart, "edge_bwd: null pointer");
  GC_REQUIRE(!gt.P || (N <= GT && gt.B == B && gt.slices > 0 && gt.dA && gt.uvc && gt.ds && gt.dX), "edge_bwd: bad GAT passenger");
  static_assert(sizeof(GatTail) + sizeof(GemmGroup) + 96 <= 4096, "edge_bwd_carry_kernel: kernel arguments exceed 4 KB");
  static_assert(sizeof(float) * GAT_DOC_LDS <= sizeof(float) * lds_floats<1, 1, true, true>(), "GAT passenger needs more LDS than a tile");
  const int ngat = gt.P ? B * gt.slices : 0;
  const bool vec = (D % 4 == 0) && al16(E) && al16(v) && (!dE || al16(dE)) && (!dEbar || al16(dEbar));
  const size_t lds_row = ((size_t)((N + 3) & ~3) + (size_t)EW * D) * sizeof(float);
  GC_REQUIRE(lds_row <= 160 * 1024, "edge_bwd: N=%d D=%d needs %zu B of LDS", N, D, lds_row);
  const size_t lds = (ngat && lds_row < sizeof(float) * GAT_DOC_LDS) ? sizeof(float) * GAT_DOC_LDS : lds_row;
  dim3 block(64 * EW);
  GemmGroup gg;
  double gflops = 0;
  const int ntile = (carry && carry->n > 0 && vec && lds_row <= sizeof(float) * lds_floats<1, 1, true, true>())
                        ? gemm_take_deferred(carry, gg, &gflops)
                        : 0;
  if (ntile > 0) {  // parked weight-gradient products ride along
    // ... and one parked second stage of a column sum (a bias gradient's 64 partial rows -> its C columns), in trailing workgroups
    const int col_base = gemm_take_deferred_col2(carry, gg.col) ? (int)((long)B * N + ngat + ntile) : 0;
    const int ncolwg = col_base ? cdiv(gg.col.C, 256) : 0;
    dim3 grid((unsigned)((long)B * N + ngat + ntile + ncolwg));
    // options carry_spread: percentage of the launch the tile cohorts are spread over (0: all tiles first, the order until
    // round 3), carry_cohort: tiles per cohort; launches of fewer than carry_spread_min tiles keep them in front
    const Spread sp = make_spread(ntile, (long)B * N + ngat, option("carry_cohort", 256),
                                  ntile >= option("carry_spread_min", 1024) ? option("carry_spread", 90) : 0);
    const double bytes = (dE ? 8.0 : 4.0) * B * N * N * D;
    bool any_rb = false;
    for (int i = 0; i < gg.nprob; ++i) any_rb = any_rb || gg.p[i].rb != nullptr;
#define GC_EDGE_CARRY(NT, RBV)                                                                                                  \
  GC_LAUNCH_TIMED("edge_bwd", bytes, (edge_bwd_carry_kernel<4, NT, RBV>), grid, block, 0, st, E, v, n_valid, dlogit, dEbar, dE, \
                  dvpart, N, D, nt_store(), sp, gt, gg, col_base)
    if (nt_e1()) { if (any_rb) GC_EDGE_CARRY(true, true); else GC_EDGE_CARRY(true, false); }
    else { if (any_rb) GC_EDGE_CARRY(false, true); else GC_EDGE_CARRY(false, false); }
#undef GC_EDGE_CARRY
    return check_launch("edge_bwd_carry");
  }
  dim3 grid((unsigned)((long)B * N + ngat));
  const double bytes = (dE ? 8.0 : 4.0) * B * N * N * D;
#define GC_EDGE_BWD(V, NT)                                                                                                 \
  GC_LAUNCH_TIMED("edge_bwd", bytes, (edge_bwd_kernel<V, NT>), grid, block, lds, st, E, v, n_valid, dlogit, dEbar, dE, dvpart, N, D, \
                  nt_store(), gt)
  if (vec) { if (nt_e1()) GC_EDGE_BWD(4, true); else GC_EDGE_BWD(4, false); }
  else { if (nt_e1()) GC_EDGE_BWD(1, true); else GC_EDGE_BWD(1, false); }
#undef GC_EDGE_BWD
  return check_launch("edge_bwd");
}

int edge_bcast(const float* dEbar, const int* n_valid, float* dE, int B, int N, int D, hipStream_t st) {
  GC_REQUIRE(dEbar && dE, "edge_bcast: null pointer");
  const bool vec = (D % 4 == 0) && al16(dE) && al16(dEbar);
  dim3 grid((unsigned)((long)B * N)), block(64 * EW);
  ProfScope ps("edge_bcast", st, 4.0 * B * N * N * D);
  if (vec) hipLaunchKernelGGL((edge_bcast_kernel<4>), grid, block, 0, st, dEbar, n_valid, dE, N, D, nt_store());
  else hipLaunchKernelGGL((edge_bcast_kernel<1>), grid, block, 0, st, dEbar, n_valid, dE, N, D, nt_store());
  return check_launch("edge_bcast");
}

}  // namespace gc
