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

#include <type_traits>

#include "edge_body.hpp"
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

// Trace build (-DGC_T_TRACE, tools/trace_chain.py): workgroup 0 of the chain_s kernels stamps the 100 MHz wall clock at phase
// boundaries (forward 0..19, backward 20..63; CAGGC's launches 64 slots further); compiled out of the product build.
#ifdef GC_T_TRACE
__device__ long long gc_trace_s[256];
#define TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gc_trace_s[(i) + (c.H == 1 ? 64 : 0)] = wall_clock64(); } while (0)
extern "C" int gcgcn_debug_trace_s(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_trace_s), sizeof(long long) * 256); }
#else
#define TS(i)
#endif

__device__ __forceinline__ bool dev_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// A chain workgroup is 512 threads = two 256-thread tile teams with an LDS image each.  The teams walk the
// 64x64 tiles of a product side by side (team 0 the even ones, team 1 the odd ones), so every SIMD hosts two
// waves whose instruction and memory latencies overlap -- with one workgroup per CU nothing else would.
// Both teams execute the same number of barriers (same K); when the tile count is odd, team 1 recomputes the
// last tile without storing it.
constexpr int TEAM_LDS = lds_floats<1, 1, true, true>();
constexpr int CHAIN_LDS = 2 * TEAM_LDS;
constexpr int XCHG_LDS = 64 * 64;  // passenger tiles of parked products: the two teams' partial sums meet here
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
    int pb;
    if (spread_pick((int)blockIdx.x - c.B * c.H, c.carry, pb)) {  // passenger workgroup: one tile of a parked product, K split over the two tile teams
      gemm_group_splitk_block(cg, pb, lds, TEAM_LDS, lds + CHAIN_LDS);
      return;
    }
    const EdgeRide& r = c.ride;  // passenger workgroup: one entity row of the riding dE broadcast
    edge_bcast_row<4, CW>(r.in, r.n_valid, r.out, r.N, r.D, 0, pb);
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

// ---------------------------------------------------------------------------------------------
// The same two sequences for the shape the model runs at by default -- 64 entities, two sub-layers of 128 features
// per head (hidden 256, glove:254-262) -- with every tile that one product hands to the next kept in LDS.
//
// The generic kernels above hand a product's output to the next product through global memory: stores, a barrier,
// then loads that have to come back from L2 -- ~5 us per dependent product for ~1 us of matrix work.  Here a
// (document, head) pair's whole working set lives in one compute unit's 160 KB of LDS: the adjacency image (17 KB),
// the Pn / Y (forward) or dM / Pn / dPn (backward) operand images (34 KB each).  Eight waves each own one 32x32 block
// of a 64x128 result (or one block and one K half of the 64x64 dA); operands are read from LDS in MFMA order -- the
// operand stored with its reduction index contiguous as one 16-byte read per four MFMAs (the four k of a read go to
// the four MFMAs, lanes 32-63 take the next four: a permutation of the reduction order), the other one as four
// 4-byte reads.  Weights (Wd, shared by all documents, L2-hot) go from global memory straight into the B-operand
// registers, requested a whole product ahead.  Results are stored to global memory as the saved tensors of
// backward / outputs, but nobody waits for those stores.  Barriers: 3 forward, 8 backward (+ 22 in the fused product).
// ---------------------------------------------------------------------------------------------
constexpr int S_LA = 68;    // row pitch of the 64x64 adjacency image (floats; 16-byte rows, conflict-free 16-byte reads)
constexpr int S_LP = 132;   // row pitch of a 64x128 operand image
constexpr int S_GH = 128;
constexpr int S_FWD_LDS = 64 * S_LA + 2 * 64 * S_LP + 64 + S_GH * S_LP;   // + the Wd_1 image [128][132]
constexpr int S_BWD_XS = 6 * 16 * 64;   // dXres: the partial sums of three K quarters (FUSE, H > 1)
constexpr int S_BWD_LDS = 64 * S_LA + 3 * 64 * S_LP + 128 + S_BWD_XS;
static_assert(S_FWD_LDS >= CHAIN_LDS && S_BWD_LDS >= CHAIN_LDS + XCHG_LDS, "passengers use the chain kernels' LDS");
static_assert(S_BWD_LDS * sizeof(float) <= 160 * 1024, "LDS of one compute unit");

// Workgroup barrier for hand-overs through LDS: waits for this wave's LDS operations only.  __syncthreads() also waits
// for every global store to be acknowledged and every outstanding prefetch to land -- microseconds the chain need not spend,
// since nothing one wave stores to global memory is read by another wave of the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// acc += A[32 x K] B[K x 32].  AV: pa points at this lane's A row, 4 * (lane / 32) floats in, k contiguous;
// otherwise at row 4 * (lane / 32) of a [k][row] image, this lane's row.  The same for B with its column.
template <int K, bool AV, bool BV>
__device__ __forceinline__ void mma_lds(f32x16& acc, const float* __restrict__ pa, const int la,
                                        const float* __restrict__ pb, const int lb) {
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += 8) {
    float a[4], b[4];
    if (AV) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pa + k0);
      a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = pa[(k0 + m) * la];
    }
    if (BV) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pb + k0);
      b[0] = v.x, b[1] = v.y, b[2] = v.z, b[3] = v.w;
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) b[m] = pb[(k0 + m) * lb];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], acc, 0, 0, 0);
  }
}
// row of accumulator element q of the lane (its column is lane % 32): rows r0 + {0..3} + 8 * {0..3}, r0 = block row + 4 * (lane / 32)
__device__ __forceinline__ int acc_row(int r0, int q) { return r0 + (q & 3) + 8 * (q >> 2); }

__global__ __launch_bounds__(64 * CW) void gcn_chain_s_fwd_kernel(const GcnCtx c) {
  __shared__ __attribute__((aligned(16))) float lds[S_FWD_LDS];
  if (blockIdx.x >= c.B * c.H) {  // passenger workgroup: one entity row of the riding edge mean
    const EdgeRide& r = c.ride;
    edge_fwd_row<4, false, true, CW>(r.in, nullptr, r.n_valid, r.out, nullptr, nullptr, nullptr, Drop(), r.N, r.D,
                                     blockIdx.x - c.B * c.H, lds);
    return;
  }
  TS(0);
  float* const As = lds;
  float* const Ps = As + 64 * S_LA;
  float* const Ys = Ps + 64 * S_LP;
  float* const Rs = Ys + 64 * S_LP;
  float* const Ws = Rs + 64;                           // Wd_1 [k][n], fetched once per workgroup
  const int z = blockIdx.x, b = z / c.H, h = z - b * c.H;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hf = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;             // this wave's 32x32 block of a 64x128 result
  const unsigned HD = (unsigned)c.HD, D = (unsigned)c.D;
  const long zoff = (long)b * 64 * c.HD + (long)h * 2 * S_GH;  // (b, row 0, h, l = 0, k = 0) in [B*N, H, L, gh]
  const float* __restrict__ Ag = c.A + (long)z * 64 * 64;
  const float* __restrict__ Pg = c.Pn + zoff;
  const float* __restrict__ Gg = c.G + zoff;
  const float* __restrict__ Xg = c.X + (long)b * 64 * c.D;
  float* __restrict__ Yg = c.Y + zoff;
  float* __restrict__ Hg = c.HO + zoff;
  const int r0 = rb * 32 + hf * 4, col = cb * 32 + r;
  const bool dd = c.drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(c.drop) : 0;

  // ---- requests, in the order their data is needed ---------------------------------------------------------------
  const bool own_att = c.mha.Q != nullptr;   // the attention core runs here (uniform over the launch)
  f32x4 a[2], p[4];   // A_h and Pn_0 = X Wn_0 (written by the launch before): the first product's operands
  float s[8];          // A's rows once more for the normaliser (glove:47-49)
  if (!own_att) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 512 * u;
      a[u] = *reinterpret_cast<const f32x4*>(Ag + (idx >> 4) * 64 + (idx & 15) * 4);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = t + 512 * u;
    p[u] = *reinterpret_cast<const f32x4*>(Pg + (unsigned)(idx >> 5) * HD + (idx & 31) * 4);
  }
  if (!own_att) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = Ag[(wave * 8 + u) * 64 + lane];
  }
  float gv[2][16], xv[2][16], pv[16];   // epilogue operands of both sub-layers, Pn_1 = X Wn_1 as the dense product's start
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const unsigned row = (unsigned)acc_row(r0, q);
    gv[0][q] = Gg[row * HD + (unsigned)col];
    xv[0][q] = Xg[row * D + (unsigned)col];
  }
  TS(1);
  if (own_att) {
    // MultiHeadAttention's core for this (document, head) pair (mha_body.hpp; scratch: the Y image, free until the first
    // product's epilogue): P / A to global memory for backward, the adjacency the chain uses into As; its rows again from
    // there for the normaliser
    mha_core_fwd_body<true>(Ys, z, c.mha.Q, c.n_valid, c.mha.P, c.mha.A, 64, c.D, c.H, c.mha.dh, c.mha.kchunk, c.mha.alpha, c.mha.drop, t,
                            t < 256, As, S_LA, CW);
    lds_barrier();   // (LDS only: the adjacency image is complete; nobody waits for the P / A stores)
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = As[(wave * 8 + u) * S_LA + lane];
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 512 * u;
      *reinterpret_cast<f32x4*>(As + (idx >> 4) * S_LA + (idx & 15) * 4) = a[u];
    }
  }
  TS(2);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = t + 512 * u;
    *reinterpret_cast<f32x4*>(Ps + (idx >> 5) * S_LP + (idx & 31) * 4) = p[u];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float tt = wave_sum(s[u]);
    if (lane == 0) {
      const float ri = 1.f / (tt + (tt == 0.f ? 1.f : 0.f));
      Rs[wave * 8 + u] = ri;
      c.rinv[(long)z * 64 + wave * 8 + u] = ri;
    }
  }
  // Wd_1 (the dense connection's weight, [128 x 128] of this head, L2-hot): row by row into its LDS image, once per workgroup
  // (as B-operand registers of every wave it was fetched twice: the two row blocks of a column range)
  {
    const float* __restrict__ W = c.flat + c.wd_off(1) + (long)h * c.wd_head;
    f32x4 wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const f32x4*>(W + (t + 512 * u) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = t + 512 * u;
      *reinterpret_cast<f32x4*>(Ws + (idx >> 5) * S_LP + (idx & 31) * 4) = wv[u];
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) pv[q] = Pg[(unsigned)acc_row(r0, q) * HD + S_GH + (unsigned)col];
  TS(3);
  lds_barrier();
  TS(4);

#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const unsigned lo = (unsigned)l * S_GH + (unsigned)col;
    f32x16 acc;
    if (l == 1) {  // Pn_1 = X Wn_1 (in memory) + Y_0 Wd_1          (dense connection, glove:73 / 110)
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // the second sub-layer's epilogue operands land while this product runs
        const unsigned row = (unsigned)acc_row(r0, q);
        gv[1][q] = Gg[row * HD + S_GH + (unsigned)col];
        xv[1][q] = Xg[row * D + S_GH + (unsigned)col];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = pv[q];
      mma_lds<S_GH, true, false>(acc, Ys + (rb * 32 + r) * S_LP + 4 * hf, 0, Ws + (4 * hf) * S_LP + col, S_LP);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = acc_row(r0, q);
        c.Pn[zoff + (unsigned)row * HD + lo] = acc[q];
        Ps[row * S_LP + col] = acc[q];
      }
      TS(8);
      lds_barrier();
      TS(9);
    }
    // Y_l = relu((G_l + A_h Pn_l) rinv);  HO_l = dropout(Y_l) + X_l          (glove:42-50, 71-76)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    mma_lds<64, true, false>(acc, As + (rb * 32 + r) * S_LA + 4 * hf, 0, Ps + (4 * hf) * S_LP + col, S_LP);
    TS(5 + 5 * l);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = acc_row(r0, q);
      const unsigned o = (unsigned)row * HD + lo;
      const float v = fmaxf((acc[q] + gv[l][q]) * Rs[row], 0.f);
      Yg[o] = v;
      if (l == 0) Ys[row * S_LP + col] = v;
      float w = v;
      if (dd) w = (rng_u32(key, (uint64_t)(zoff + (long)o)) >= c.drop.thresh) ? v * c.drop.scale : 0.f;
      Hg[o] = w + xv[l][q];
    }
    TS(6 + 5 * l);
    if (l == 0) lds_barrier();
    TS(7 + 5 * l);
  }
}

// Backward of the same sequence (last sub-layer first); dA stays in registers over both sub-layers, its reduction
// split over two waves per block.  dYa is read, never written back: nothing after the chain needs it.
// Requests go out in the order their data is needed (row phase of sub-layer 1 first); the barriers wait for LDS only.
// FUSE: the gradient of the block's output projection input is computed here instead of arriving in dYa --
//   dHO_{b,h} = dout_b Wlin[:, h]  (64 x 256 x 256, Wlin streamed from L2 into the B registers),  dY = dropout_bwd(dHO),
//   dXres_b = sum_h dHO_{b,h} = dout_b (sum_h Wlin[:, h]): this workgroup's D / H columns of it (H = 1: dHO itself)
// which removes a GEMM launch, the head-sum / dropout kernel and 2 x B N H D floats of HBM traffic (glove:74-78 / 111-118).
template <bool FUSE>
__global__ __launch_bounds__(64 * CW) void gcn_chain_s_bwd_kernel(const GcnCtx c, const GemmGroup4 cg) {
  __shared__ __attribute__((aligned(16))) float lds[S_BWD_LDS];
  if (blockIdx.x >= c.B * c.H) {
    int pb;
    if (spread_pick((int)blockIdx.x - c.B * c.H, c.carry, pb)) {  // passenger workgroup: one tile of a parked product, K split over the two tile teams
      gemm_group_splitk_block(cg, pb, lds, TEAM_LDS, lds + CHAIN_LDS);
      return;
    }
    const EdgeRide& r = c.ride;  // passenger workgroup: one entity row of the riding dE broadcast
    edge_bcast_row<4, CW>(r.in, r.n_valid, r.out, r.N, r.D, 0, pb);
    return;
  }
  TS(20);
  float* const As = lds;
  float* const Ds = As + 64 * S_LA;     // dM_l; between the sub-layers the updated dY_0
  float* const Ps = Ds + 64 * S_LP;     // Pn_l
  float* const Ws = Ds;                 // Wd_1 [128 x 128] over both images while dY_0 += dPn_1 Wd_1^T runs
  float* const Qs = Ps + 64 * S_LP;     // dPn_1; at the end the K halves of dA meet here
  float* const Rs = Qs + 64 * S_LP;     // rinv
  float* const Ts = Rs + 64;            // gradient of the normaliser's row sums
  float* const Es = Ts + 64;            // FUSE: dXres's K quarters meet here
  // Workgroups go to the eight XCDs round robin (blockIdx % 8).  With z = b H + h and H = 8 an XCD would run ONE head; give
  // each XCD all heads of a few documents instead: the heads of a document share dout_b (and X_b, A's neighbours) in its L2.
  int b, h;
  if (c.H == 8 && (c.B & 7) == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    h = q & 7, b = x + 8 * (q >> 3);
  } else {
    b = blockIdx.x / c.H, h = blockIdx.x - b * c.H;
  }
  const int z = b * c.H + h;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hf = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;             // block of a 64x128 result
  const int ab = wave & 3, kh = wave >> 2;             // block of dA (rows ab & 1, columns ab >> 1) and K half
  const unsigned HD = (unsigned)c.HD;
  const long zoff = (long)b * 64 * c.HD + (long)h * 2 * S_GH;
  const float* __restrict__ Ag = c.A + (long)z * 64 * 64;
  const float* __restrict__ Pg = c.Pn + zoff;
  const float* __restrict__ Yg = c.Y + zoff;
  const float* __restrict__ Gy = c.dYa + zoff;
  const float* __restrict__ Rg = c.rinv + (long)z * 64;
  float* __restrict__ Mg = c.dM + zoff;
  float* __restrict__ Qg = c.dP + zoff;
  const int r0 = rb * 32 + hf * 4, col = cb * 32 + r;
  constexpr int S_LX = 260;             // row pitch of the 64 x 256 dout image (over the Pn and dPn images)
  float* const Xs = Ps;
  static_assert(64 * S_LX <= 2 * 64 * S_LP, "dout image");
  const bool dd = FUSE && c.drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(c.drop) : 0;

  // ---- requests ------------------------------------------------------------------------------------------------
  f32x4 dv[8];   // FUSE: dout_b
  if constexpr (FUSE) {
    const float* __restrict__ Dg = c.dout + (long)b * 64 * 256;
#pragma unroll
    for (int u = 0; u < 8; ++u) dv[u] = *reinterpret_cast<const f32x4*>(Dg + (t + 512 * u) * 4);
    if (c.dout_m) {  // padding rows get no gradient; back through the hop's output dropout (glove:341) -- was a launch of its own
      const int nvb = c.n_valid ? c.n_valid[b] : 64;
      const bool od = c.odrop.snap != nullptr;
      const uint64_t okey = od ? drop_key(c.odrop) : 0;
      float* __restrict__ Dm = c.dout_m + (long)b * 64 * 256;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = (t + 512 * u) * 4;   // element of the document's [64, 256] slice: row e / 256
        const bool keep = (e >> 8) < nvb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = keep ? dv[u][j] : 0.f;
          if (od) v = (rng_u32(okey, (uint64_t)((long)b * 64 * 256 + e + j)) >= c.odrop.thresh) ? v * c.odrop.scale : 0.f;
          dv[u][j] = v;
        }
        if (h == 0) *reinterpret_cast<f32x4*>(Dm + e) = dv[u];
      }
    }
  }
  float y[8][2], gy[8][2], rv[8];   // row phase: rows wave + 8 u, columns lane + 64 kk
  auto request_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = wave + 8 * u;
      rv[u] = Rg[i];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const unsigned o = (unsigned)i * HD + S_GH + (unsigned)(lane + 64 * kk);
        y[u][kk] = Yg[o];
        if constexpr (!FUSE) gy[u][kk] = Gy[o];
      }
    }
  };
  if constexpr (!FUSE) request_rows();
  f32x4 a[2], p1[4], p0[4], wd[8];   // the images of sub-layer 1; Pn_0, Wd_1, Y_0 and dY_0 are requested later
  float y0[8][2], gv[16];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = t + 512 * u;
    a[u] = *reinterpret_cast<const f32x4*>(Ag + (idx >> 4) * 64 + (idx & 15) * 4);
  }
  auto request_pn1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 512 * u;
      p1[u] = *reinterpret_cast<const f32x4*>(Pg + (unsigned)(idx >> 5) * HD + S_GH + (idx & 31) * 4);
    }
  };
  if constexpr (!FUSE) request_pn1();
  const float rs = (t < 64) ? Rg[t] : 0.f;
  f32x16 dacc;     // this wave's K half of its dA block
#pragma unroll
  for (int q = 0; q < 16; ++q) dacc[q] = 0.f;

  if constexpr (FUSE) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = t + 512 * u;
      *reinterpret_cast<f32x4*>(Xs + (idx >> 6) * S_LX + (idx & 63) * 4) = dv[u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 512 * u;
      *reinterpret_cast<f32x4*>(As + (idx >> 4) * S_LA + (idx & 15) * 4) = a[u];
    }
    if (t < 64) Rs[t] = rs;
    // B operands of this wave's share of dXres (Wsum rows of its K quarter, column h * 32 + r): requested now, they land
    // while the workgroup waits for dout at the barrier
    const int rb2 = wave & 1, kq = wave >> 1;
    float wsv[8][4];
    if (c.H != 1) {
      const float* __restrict__ Ws8 = c.Wsum + (unsigned)(kq * 64 + 4 * hf) * 256u + (unsigned)(h * 32 + r);
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int m = 0; m < 4; ++m) wsv[g][m] = Ws8[(8 * g + m) * 256];
    }
    TS(21);
    lds_barrier();
    TS(22);
    if (c.colpart && h == 0) {  // the output bias gradient's column sums of this document: rows 0-31 and 32-63
      const int cc = t & 255, half = t >> 8;
      float sacc = 0.f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) sacc += Xs[(half * 32 + i) * S_LX + cc];
      c.colpart[(long)(2 * b + half) * 256 + cc] = sacc;
    }
    // dHO_b = dout_b Wlin[:, h]: wave = (rows rb, column quarter cq: 64 columns of the head's 256 = half a sub-layer), the whole
    // K.  Two accumulators whose columns interleave: lane r owns columns 2 r, 2 r + 1, accumulator j the columns = j (mod 2), so
    // one 8-byte B read feeds two MFMAs and the results leave as 8-byte stores.  (Round 3 split K over wave pairs instead, four
    // accumulators of 128 columns each: the halves then met in LDS behind two more barriers -- 7.0 us between the product and the
    // first sub-layer, profiles/r04_chain_phase_trace_before.txt.)
    const int cq = wave >> 1, l2 = cq >> 1, cl = (cq & 1) * 64 + 2 * r;   // cl: first of this lane's two columns in its sub-layer
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x16 ho[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) ho[j][q] = 0.f;
    float* __restrict__ Xr = c.dXres + (long)b * 64 * 256;
    f32x16 xs[1];   // H == 8: rows (wave & 1), K quarter (wave >> 1) of this head's 32 columns of dXres
    if (c.H != 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xs[0][q] = 0.f;
      const float* pax = Xs + (rb2 * 32 + r) * S_LX + kq * 64 + 4 * hf;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(pax + 8 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) xs[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], wsv[g][m], xs[0], 0, 0, 0);
      }
      if (kq > 0) {  // the quarters meet in an area of their own: read back behind the product's last barrier
#pragma unroll
        for (int q = 0; q < 16; ++q) Es[(((kq - 1) * 2 + rb2) * 16 + q) * 64 + lane] = xs[0][q];
      }
    }
    {  // Wlin's slice goes through LDS in chunks of 16 k (two stages in the dM image's place), loaded once per workgroup by all
       // 512 threads, coalesced: a compute unit takes in ~20 GB/s from L2, and wave pairs that fetch the same B rows
       // themselves (through an L1 that has long dropped them) make that 512 KB per workgroup instead of 256.
      constexpr int S_LB = 260, BST = 16 * S_LB;
      static_assert(2 * BST <= 64 * S_LP, "B stages live in the dM image");
      float* const Bs = Ds;
      const float* __restrict__ W = c.flat + c.oWlin + (long)h * 256 + (long)(t >> 6) * c.HD + (t & 63) * 4;
      f32x4 br[2][2];
      auto gload = [&](const int ch, f32x4 (&d)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) d[u] = *reinterpret_cast<const f32x4*>(W + (long)(ch * 16 + 8 * u) * c.HD);
      };
      auto sstore = [&](const int st, const f32x4 (&d)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(Bs + st * BST + ((t >> 6) + 8 * u) * S_LB + (t & 63) * 4) = d[u];
      };
      const float* pa = Xs + (rb * 32 + r) * S_LX + 4 * hf;
      const float* pb = Bs + (4 * hf) * S_LB + cq * 64 + 2 * r;
      auto compute = [&](const int ch, const int st) __attribute__((always_inline)) {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {   // the two groups of 8 k of a chunk: the four k of a read go to four MFMAs, lanes 32-63 take the next four
          const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 16 * ch + 8 * g2);
          f32x2 b[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) b[m] = *reinterpret_cast<const f32x2*>(pb + st * BST + (8 * g2 + m) * S_LB);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) ho[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m][j], ho[j], 0, 0, 0);
        }
      };
      TS(23);
      gload(0, br[0]);
      gload(1, br[1]);
      sstore(0, br[0]);
      lds_barrier();
      TS(24);
      for (int ch = 0; ch < 16; ch += 2) {   // two chunks per trip: register sets and stages are compile-time constants
        if (ch + 2 < 16) gload(ch + 2, br[0]);
        compute(ch, 0);
        sstore(1, br[1]);
        lds_barrier();
        if (ch + 3 < 16) gload(ch + 3, br[1]);
        compute(ch + 1, 1);
        if (ch + 2 < 16) sstore(0, br[0]);
        if (ch == 12) {   // for the row phase and the images of sub-layer 1: they land while the last chunks run
          request_rows();
          request_pn1();
        }
        lds_barrier();
      }
    }
    TS(25);
    // (the loop's last barrier: everybody is done with the dout image and the weight stages, and dXres's quarters are in Es)
    if (c.H != 1 && kq == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = xs[0][q];
#pragma unroll
        for (int k = 0; k < 3; ++k) v += Es[((k * 2 + rb2) * 16 + q) * 64 + lane];
        Xr[acc_row(rb2 * 32 + hf * 4, q) * 256 + h * 32 + r] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 512 * u;
      *reinterpret_cast<f32x4*>(Ps + (idx >> 5) * S_LP + (idx & 31) * 4) = p1[u];
    }
    {  // dY = dropout_bwd(dHO): images for the row phase (sub-layer 1) and for dY_0's accumulator start
      float* const img = l2 ? Ds : Qs;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = acc_row(r0, q);
        f32x2 g = {ho[0][q], ho[1][q]};
        if (c.H == 1) *reinterpret_cast<f32x2*>(Xr + row * 256 + l2 * S_GH + cl) = g;
        if (dd) {
          const long o = zoff + (long)((unsigned)row * HD + (unsigned)(l2 * S_GH + cl));
#pragma unroll
          for (int j = 0; j < 2; ++j) g[j] = (rng_u32(key, (uint64_t)(o + j)) >= c.drop.thresh) ? g[j] * c.drop.scale : 0.f;
        }
        *reinterpret_cast<f32x2*>(img + row * S_LP + cl) = g;
      }
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) gv[q] = Qs[acc_row(r0, q) * S_LP + col];
    TS(26);
  }

  auto sublayer = [&](auto lt) __attribute__((always_inline)) {   // l is a compile-time constant: the two passes differ in what they read and hand on
    constexpr int l = decltype(lt)::value;
    const unsigned lo = (unsigned)l * S_GH;
    TS(30 + 10 * l);
    // through Y = relu(S), S = M rinv:  dS = dY [Y > 0];  dM = dS rinv;  drow -= rinv sum_c dS Y
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = wave + 8 * u;
      float acc = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const float yy = (l == 1) ? y[u][kk] : y0[u][kk];
        const float gg = (l == 1 && !FUSE) ? gy[u][kk] : Ds[i * S_LP + lane + 64 * kk];
        const float g = yy > 0.f ? gg : 0.f;
        const float dm = g * rv[u];
        Mg[(unsigned)i * HD + lo + (unsigned)(lane + 64 * kk)] = dm;
        Ds[i * S_LP + lane + 64 * kk] = dm;
        acc = fmaf(g, yy, acc);
      }
      const float tt = wave_sum(acc);
      if (lane == 0) Ts[i] = (l == 1) ? -rv[u] * tt : Ts[i] - rv[u] * tt;
    }
    if constexpr (l == 1 && !FUSE) {  // the images the products of this sub-layer read
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = t + 512 * u;
        *reinterpret_cast<f32x4*>(As + (idx >> 4) * S_LA + (idx & 15) * 4) = a[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = t + 512 * u;
        *reinterpret_cast<f32x4*>(Ps + (idx >> 5) * S_LP + (idx & 31) * 4) = p1[u];
      }
      if (t < 64) Rs[t] = rs;
    }
    TS(31 + 10 * l);
    lds_barrier();
    TS(32 + 10 * l);
    // dPn_l = A_h^T dM_l   and   dA += dM_l Pn_l^T (this wave's K half): both wait only for dM_l
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    mma_lds<64, false, false>(acc, As + (4 * hf) * S_LA + rb * 32 + r, S_LA, Ds + (4 * hf) * S_LP + col, S_LP);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = acc_row(r0, q);
      Qg[(unsigned)row * HD + lo + (unsigned)col] = acc[q];
      if constexpr (l == 1) Qs[row * S_LP + col] = acc[q];
    }
    if constexpr (l == 1) {  // Wd_1 row by row (L2-hot, coalesced), transposed through LDS once the images below are free
      const float* __restrict__ W = c.flat + c.wd_off(1) + (long)h * c.wd_head;
#pragma unroll
      for (int u = 0; u < 8; ++u) wd[u] = *reinterpret_cast<const f32x4*>(W + (t + 512 * u) * 4);
    }
    TS(33 + 10 * l);
    mma_lds<64, true, true>(dacc, Ds + ((ab & 1) * 32 + r) * S_LP + kh * 64 + 4 * hf, 0,
                            Ps + ((ab >> 1) * 32 + r) * S_LP + kh * 64 + 4 * hf, 0);
    TS(34 + 10 * l);
    if constexpr (l == 1) {
      lds_barrier();  // dPn_1 complete in LDS; dM_1 and Pn_1 images free
      TS(35 + 10 * l);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = t + 512 * u;
        *reinterpret_cast<f32x4*>(Ws + (idx >> 5) * S_LP + (idx & 31) * 4) = wd[u];
      }
      // requests for the second sub-layer: they land while the product below runs
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = t + 512 * u;
        p0[u] = *reinterpret_cast<const f32x4*>(Pg + (unsigned)(idx >> 5) * HD + (idx & 31) * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) y0[u][kk] = Yg[(unsigned)(wave + 8 * u) * HD + (unsigned)(lane + 64 * kk)];
      if constexpr (!FUSE) {
#pragma unroll
        for (int q = 0; q < 16; ++q)  // dY_0 as it arrives (dropout_bwd(dHO_0))
          gv[q] = Gy[(unsigned)acc_row(r0, q) * HD + (unsigned)col];
      }
      TS(36 + 10 * l);
      lds_barrier();
      TS(37 + 10 * l);
      // dY_0 += dPn_1 Wd_1^T:  B[k][n] = Wd_1[n][k], this lane's column n is a row of the image
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      mma_lds<S_GH, true, true>(acc, Qs + (rb * 32 + r) * S_LP + 4 * hf, 0, Ws + col * S_LP + 4 * hf, 0);
      TS(38 + 10 * l);
      lds_barrier();  // everybody is done with Wd_1's image
#pragma unroll
      for (int q = 0; q < 16; ++q) Ds[acc_row(r0, q) * S_LP + col] = acc[q] + gv[q];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = t + 512 * u;
        *reinterpret_cast<f32x4*>(Ps + (idx >> 5) * S_LP + (idx & 31) * 4) = p0[u];
      }
      lds_barrier();
    }
  };
  sublayer(std::integral_constant<int, 1>());
  sublayer(std::integral_constant<int, 0>());
  // dA = K half 0 + K half 1 + drow (every column of a row)
  if (kh == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) Qs[(ab * 16 + q) * 64 + lane] = dacc[q];
  }
  lds_barrier();
  if (kh == 0) {
    float* __restrict__ dAg = c.dA + (long)z * 64 * 64;
    const int ar0 = (ab & 1) * 32 + hf * 4, acol = (ab >> 1) * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = acc_row(ar0, q);
      dAg[row * 64 + acol] = dacc[q] + Qs[(ab * 16 + q) * 64 + lane] + Ts[row];
    }
  } else if (t >= 256 && t < 320) {
    c.drow[(long)z * 64 + (t - 256)] = Ts[t - 256];
  }
  TS(55);
}

// the shape the LDS-resident kernels are written for
static bool chain_small_ok(const GcnCtx& c, bool bwd) {
  return c.N == 64 && c.L == 2 && c.gh == S_GH && c.D == 2 * S_GH && (c.flat + c.oWd) && c.wd_head % 4 == 0 &&
         (c.wd_off(1) % 4) == 0;
}

static bool chain_aligned(const GcnCtx& c, bool bwd);
// the dispatch rule of gcn_chain_fwd / _bwd: the column-strip kernels (chain_t.hpp) take the shape instead of the default shape's own kernels
// (option chain_t: 0 never; 1 (default) wherever the default shape's own kernels do not apply, and at that shape too for RAGGED
// batches -- the column-strip kernels skip 16-row blocks per document, gcn_chain_s_* own 32 rows per wave and do not: cfg 2
// ragged 71.2 k -> 72.5 k docs/s at B = 32; dense batches stay on gcn_chain_s_* (0.522 vs 0.542 ms); 2: everywhere)
static bool chain_s_preferred(const GcnCtx& c, bool s_ok) {
  const int mode = option("chain_t", 1);
  return s_ok && (mode < 2) && !(mode == 1 && c.n_valid != nullptr);
}
static bool chain_t_takes(const GcnCtx& c, bool bwd) {
  const bool s_ok = chain_aligned(c, bwd) && chain_small_ok(c, bwd);
  return chain_t_ok(c, bwd) && !chain_s_preferred(c, s_ok);
}
bool chain_bwd_fusable(const GcnCtx& c) {   // asked before c.dout is set
  const bool on = option("chain_fuse", 1) != 0;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  if (on && chain_t_takes(c, true)) return chain_t_bwd_fusable(c);
  return on && chain_small_ok(c, true) && c.N == 64 && (c.H == 1 || c.H == 8) && al(c.A) && al(c.Pn) && al(c.Y) && al(c.dM) &&
         al(c.dP) && al(c.dA) && al(c.flat + c.oWd) && al(c.flat + c.oWlin) && c.HD % 4 == 0;
}

static bool chain_aligned(const GcnCtx& c, bool bwd) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = c.N % 64 == 0 && c.gh % 64 == 0 && al(c.A) && al(c.flat + c.oWd) && al(c.Pn) && al(c.Y);
  if (bwd) ok = ok && (c.dout ? al(c.dout) && al(c.dXres) : al(c.dYa)) && al(c.dM) && al(c.dP) && al(c.dA);
  else ok = ok && al(c.G) && al(c.HO) && al(c.X);
  return ok;
}

bool chain_fwd_computes_attention(const GcnCtx& c) {   // before c.mha is set: a question about the shape and the options
  if (option("att_in_chain", 1) == 0) return false;
  if (chain_t_takes(c, false)) return chain_t_fwd_att_ok(c);
  return chain_aligned(c, false) && chain_small_ok(c, false) && mha_lds_bytes(c.D / c.H) <= sizeof(float) * 64 * S_LP &&
         (c.D / c.H) % 4 == 0;
}

// The riding pass uses the 16-byte row bodies and the chain kernel's static LDS for its per-wave column sums.
bool chain_can_carry(const EdgeRide& r) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  return r.kind != 0 && r.D % 4 == 0 && al(r.in) && al(r.out) && (long)CW * r.D <= CHAIN_LDS &&
         (long)r.B * r.N <= 0x3fffffffL;
}

// GCGCN_CHAIN_CARRY=0: no parked products in the chain launches (A/B knob)
static bool chain_passengers() { return option("chain_carry", 1) != 0; }

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
  // (which kernel generation serves the shape: chain_s_preferred above)
  const bool s_ok = chain_aligned(c, false) && chain_small_ok(c, false);
  if (chain_t_ok(c, false) && !chain_s_preferred(c, s_ok)) return gcn_chain_t_fwd(c, grid, fl * c.B * c.H, st);
  // only the LDS-resident kernels run the attention core in their prologue: a caller that left it to the chain (c.mha) and ends
  // up here would get a convolution over adjacencies nobody computed
  GC_REQUIRE(!c.mha.Q || s_ok, "gcn_chain_fwd: the attention core was left to a chain kernel that does not run it");
  if (s_ok)
    GC_LAUNCH_TIMED("gcn_chain_fwd", fl * c.B * c.H, gcn_chain_s_fwd_kernel, grid, block, 0, st, c);
  else if (chain_aligned(c, false)) GC_LAUNCH_TIMED("gcn_chain_fwd", fl * c.B * c.H, gcn_chain_fwd_kernel<true>, grid, block, 0, st, c);
  else GC_LAUNCH_TIMED("gcn_chain_fwd", fl * c.B * c.H, gcn_chain_fwd_kernel<false>, grid, block, 0, st, c);
  return check_launch("gcn_chain_fwd");
}

static int chain_bwd_launch(const GcnCtx& c, const GemmGroup4& cg, dim3 grid, dim3 block, double fl, hipStream_t st);

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
  const bool s_ok = chain_aligned(c, true) && chain_small_ok(c, true);
  const bool use_t = chain_t_ok(c, true) && !chain_s_preferred(c, s_ok);
  if (use_t) return gcn_chain_t_bwd(c, fl, st, chain_passengers() ? carry : nullptr);
  int ng = 0;
  if (carry && carry->n > 0 && (long)c.B * c.H <= 64 && ((long)c.B * c.N) % 64 == 0 && chain_passengers()) {
    const int passes = (((c.N + 63) / 64) * ((c.gh + 63) / 64) + 1) / 2;
    const double t_chain = 8.0 * (4 * c.L - 1) * passes;                  // us
    const double t_tile = 0.47 * ((double)c.B * c.N / 32.0);              // us: weight gradients have K = B N
    const long rounds = t_tile > 0 ? (long)(t_chain / t_tile) : 0;
    bool halves = true;
    for (int i = 0; i < carry->n; ++i) halves = halves && carry->p[i].K % 64 == 0 && !carry->p[i].rb;   // (row-block products ride elsewhere)
    // the riding dE broadcast needs its share of the idle compute units too: with it aboard only half of them take a tile
    long budget = rounds * (256 - (long)c.B * c.H);
    if (c.ride.kind == 2) budget = carry_budget_pct() * budget / 100;
    if (rounds > 0 && halves && budget > 0) ng = gemm_take_deferred_pairs(carry, cg, &fl, budget);
  }
  dim3 grid(chain_grid(c, 2) + (unsigned)ng), block(64 * CW);
  GcnCtx cc = c;
  cc.carry = chain_carry_spread(c, ng);
  return chain_bwd_launch(cc, cg, grid, block, fl, st);
}

Spread chain_carry_spread(const GcnCtx& c, int ng) {
  // options chain_spread / chain_cohort / chain_spread_min: as carry_spread / carry_cohort / carry_spread_min (edge.hip) for
  // the tile workgroups of a chain launch in which dE-broadcast rows ride as well; these tile workgroups take a compute
  // unit each, so a cohort is half the idle units (cfg 5: 7.89 -> 7.78 ms; neutral at cfg 3; cfg 2 has too few tiles)
  const long rows = c.ride.kind == 2 ? (long)c.ride.B * c.ride.N : 0;
  const bool on = rows > 0 && ng >= option("chain_spread_min", 256);
  return make_spread(ng, rows, option("chain_cohort", 128), on ? option("chain_spread", 85) : 0);
}

static int chain_bwd_launch(const GcnCtx& c, const GemmGroup4& cg, dim3 grid, dim3 block, double fl, hipStream_t st) {
  if (c.dout) {  // the caller asked for the fused output-projection gradient (after chain_bwd_fusable said yes)
    GC_REQUIRE(chain_aligned(c, true) && chain_bwd_fusable(c) && c.dXres && (c.H == 1 || c.Wsum), "gcn_chain_bwd: fused backward not available");
    fl += 2.0 * c.B * c.N * c.D * c.D * (c.H + (c.H > 1 ? 1 : 0));
    GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_s_bwd_kernel<true>, grid, block, 0, st, c, cg);
  } else if (chain_aligned(c, true) && chain_small_ok(c, true))
    GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_s_bwd_kernel<false>, grid, block, 0, st, c, cg);
  else if (chain_aligned(c, true)) GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_bwd_kernel<true>, grid, block, 0, st, c, cg);
  else GC_LAUNCH_TIMED("gcn_chain_bwd", fl, gcn_chain_bwd_kernel<false>, grid, block, 0, st, c, cg);
  return check_launch("gcn_chain_bwd");
}

}  // namespace gc
