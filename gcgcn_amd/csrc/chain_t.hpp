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
//  * ROW BLOCKS AT COMPILE TIME (round 5).  The bodies are templates on NRB = the number of 16-row blocks that hold real
//    entities (0 .. 4).  The kernel picks a body with ONE uniform switch per workgroup; inside a body there is no branch
//    around a matrix instruction and no run-time row-block test.  Before, every MFMA group of a ragged launch sat behind its
//    own `rb < nrb` branch: ~650 EXEC-masked regions per kernel, LDS reads that could not overlap the MFMAs of the next
//    block, 40 more registers than the full-size instantiation, and a compiler hazard-recognizer miss at the branch joins
//    that made one build non-deterministic (tools/isa_mfma_hazard_check.py now checks every build).
//  * BUFFER ADDRESSING, BOUNDS BY THE DESCRIPTOR.  A thread's strip accesses are tensor[(16 rb + 4 g + v) stride + l gh + col]:
//    its own part (4 g stride + col) is ONE 32-bit byte offset (voffset), the row part is uniform (soffset, a scalar
//    register), l gh the instruction's immediate, and the tensor's base sits in a buffer descriptor whose size is the
//    document's N rows: `buffer_load_dword v, v_off, s[desc:desc+3], s_row offen offset:imm`.  Rows past N read as zero and
//    their stores are dropped BY THE HARDWARE's range check -- no `row < N` compare, no EXEC mask, the same code for every
//    N -- and the 64-bit per-row vector addresses this replaces cost ~20 registers.
//
// Reference: GraphConv.forward glove:36-50 inside the dense loops of GraphConvolution.forward glove:70-76 /
// MultiGraphConvolution.forward glove:102-113, and their autograd.
#pragma once
#include <type_traits>

#include "edge_body.hpp"
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

// Trace build (-DGC_T_TRACE, tools/trace_chain.py): workgroup 0 stamps the 100 MHz wall clock at phase boundaries (forward:
// slots 0..63, backward: 100.. for the MAGGC launch, 164.. for CAGGC's); compiled out of the product build.
#ifdef GC_T_TRACE
static __device__ long long gc_trace_t[256];   // (one copy per translation unit: gcgcn_debug_trace_t_<unit> reads its own)
#define TR(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gc_trace_t[i] = wall_clock64(); } while (0)
#define TRB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gc_trace_t[(i) + (c.H == 1 ? 64 : 0)] = wall_clock64(); } while (0)
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
// Buffer descriptor over `rows` rows of a [., stride] fp32 tensor starting at `base` (= this document's row 0, this head's
// column 0), of which the workgroup touches `width` columns: offsets past the last of them are out of range.  Raw buffer
// (stride field 0), 32-bit data format.  Built from blockIdx-derived scalars only: the descriptor stays in scalar registers.
typedef __amdgpu_buffer_rsrc_t t_desc;
__device__ __forceinline__ t_desc t_buffer(const float* base, int rows, unsigned stride, unsigned width) {
  const int bytes = rows > 0 ? (int)(((unsigned)(rows - 1) * stride + width) * 4u) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float t_ld(t_desc d, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ t4 t_ld4(t_desc d, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(t4, __builtin_amdgcn_raw_buffer_load_b128(d, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void t_st(t_desc d, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), d, (int)voff, (int)soff, 0);
}
typedef unsigned t_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void t_st4(t_desc d, unsigned voff, unsigned soff, t4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(t_u4, v), d, (int)voff, (int)soff, 0);
}

// "this value exists now": an empty volatile statement that reads and writes an accumulator, so that the instructions producing
// it can be neither sunk past this point nor hoisted above it
__device__ __forceinline__ void t_pin(t4& acc) { asm volatile("" : "+v"(acc)); }

template <int GH, int L>
constexpr int t_fwd_lds() { return 64 * T_LA + 64 * (GH + 4) + 2 * (L - 1) * 16 * (GH + 4) + 64; }
template <int GH>
constexpr int t_bwd_lds() { return 64 * T_LA + 2 * 64 * (GH + 4) + (GH / 16) * 64 + 128; }
// FUSE (the output projection's input gradient computed by the backward kernel itself): the dout image [64][D + 4] and two
// 16-deep stages of Wlin's slice lie over the dM / Pn images (used only afterwards); behind the row-sum areas the K slices of
// the residual gradient meet
constexpr int t_max(int a, int b) { return a > b ? a : b; }
template <int GH, int L>
constexpr int t_bwd_region() { return t_max(2 * 64 * (GH + 4), 96 * (L * GH + 4)); }
template <int GH, int L>
constexpr int t_bwd_fuse_lds() { return 64 * T_LA + t_bwd_region<GH, L>() + (GH / 16) * 64 + 128 + (GH / 16) * 1024; }
template <int GH, int L>
constexpr bool t_fuse_shape() { return L * GH <= 256 && GH <= 128 && 16 % L == 0; }   // wider blocks: the images do not fit (and the product is a launch's worth); L = 3: the image pieces do not divide over the threads

// ---------------------------------------------------------------------------------------------------------------------
// forward:  rinv = 1 / rowsum(A_h);  for l:  Y_l = relu((G_l + A_h Pn_l) rinv),  HO_l = dropout(Y_l) + X_l,
//           Pn_l' += Y_l Wd_{l'}[l gh : (l + 1) gh, :]  for l' > l   (Pn_l' starts as X Wn_l'[:D], written by the launch before)
// ---------------------------------------------------------------------------------------------------------------------
// NRB: 16-row blocks that hold real entities (padding rows are zero everywhere; blocks past NRB are neither read nor
// computed, everything this kernel leaves there is zero).
template <int GH, int L, int NRB>
__device__ __forceinline__ void chain_t_fwd_body(const GcnCtx& c, float* __restrict__ lds, const int z, const int b, const int h) {
  constexpr int W = GH / 16, NT = 4 * GH, P = GH + 4, NC = GH / 16;
  TR(0);
  float* const As = lds;
  float* const Ys = As + 64 * T_LA;
  float* const Ws = Ys + 64 * P;                    // [2 stages][L - 1 pending sub-layers][16 k][P]
  float* const Rs = Ws + 2 * (L - 1) * 16 * P;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 15, g = lane >> 4;
  const int N = c.N;
  const unsigned HD = (unsigned)c.HD, D = (unsigned)c.D;
  const long zoff = (long)b * N * c.HD + (long)h * c.D;  // (b, row 0, h, l = 0, k = 0) in [B*N, H, L, gh]
  const float* __restrict__ Ag = c.A + (long)z * N * N;
  const t_desc Pd = t_buffer(c.Pn + zoff, N, HD, D), Gd = t_buffer(c.G + zoff, N, HD, D), Xd = t_buffer(c.X + (long)b * N * c.D, N, D, D);
  const t_desc Yd = t_buffer(c.Y + zoff, N, HD, D), Hd = t_buffer(c.HO + zoff, N, HD, D);
  const int col = 16 * w + j;
  const unsigned tHD = ((unsigned)(4 * g) * HD + (unsigned)col) * 4u;   // this thread's byte offset in an [., HD] tensor
  const unsigned tD = ((unsigned)(4 * g) * D + (unsigned)col) * 4u;     //                          in X [., D]
  const bool dd = c.drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(c.drop) : 0;
  // the uniform part of a strip access: row 16 rb + v, sub-layer l (bytes)
  auto rowHD = [&](const int rb, const int v, const int l) { return ((unsigned)(16 * rb + v) * HD + (unsigned)(l * GH)) * 4u; };
  auto rowD = [&](const int rb, const int v, const int l) { return ((unsigned)(16 * rb + v) * D + (unsigned)(l * GH)) * 4u; };
  // LDS the same way: ONE address register per image -- the thread's own (row 4 g, column col) element -- and the rest,
  // (16 rb + v) rows further, in the instruction's immediate.  (Written as image[row * P + col] the compiler keeps a
  // register per row and carries all sixteen from sub-layer to sub-layer: that, not the arithmetic, filled the register file.)
  const t_desc Wdd = t_buffer(c.flat + c.oWd + (long)h * c.wd_head, 1, 0, (unsigned)c.wd_head);   // this head's dense-connection weights
  float* const Ysg = Ys + 4 * g * P + col;
  const float* const Rsg = Rs + 4 * g;
  const float* const Asj = As + j * T_LA + 4 * g;        // A operand: row j of a block, k group g
  const float* const Ysj = Ys + j * P + 4 * g;

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
      for (int v = 0; v < 4; ++v) Pa[l][rb][v] = rb < NRB ? t_ld(Pd, tHD, rowHD(rb, v, l)) : 0.f;
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
        gv[rb][v] = rb < NRB ? t_ld(Gd, tHD, rowHD(rb, v, l)) : 0.f;
        xv[rb][v] = rb < NRB ? t_ld(Xd, tD, rowD(rb, v, l)) : 0.f;
      }
    t4 wr[2][NP > 0 ? NP : 1];
    // chunk ch of Wd_{l'}'s rows [l gh, (l + 1) gh): 16 x gh floats, contiguous -- thread t takes floats 4 t .. 4 t + 3
    auto gload = [&](const int ch, t4 (&d)[NP > 0 ? NP : 1]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned off = (unsigned)(GH * GH * ((l + 1 + p) * (l + p) / 2 + l));   // wd_off(l + 1 + p) - oWd + l gh gh
        d[p] = t_ld4(Wdd, 16u * (unsigned)t, (off + (unsigned)ch * 16u * GH) * 4u);
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
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (ob < NRB && kb < NRB) {
          const t4 a = *reinterpret_cast<const t4*>(Asj + 16 * ob * T_LA + 16 * kb);
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[ob] = mfma16(a[v], Pa[l][kb][v], acc[ob]);
        }
      }
    }
    TR(11 + 8 * l);
    // ---- Y_l = relu((G_l + A_h Pn_l) rinv);  HO_l = dropout(Y_l) + X_l          (glove:42-50, 71-76) ----------------------
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float y = fmaxf((acc[rb][v] + gv[rb][v]) * Rsg[16 * rb + v], 0.f);
        if constexpr (NP > 0) Ysg[(16 * rb + v) * P] = y;
        t_st(Yd, tHD, rowHD(rb, v, l), y);        // (rows past N: dropped by the descriptor's range check)
        float d = y;
        if (dd) {
          const unsigned o = (unsigned)(16 * rb + 4 * g + v) * HD + (unsigned)(l * GH + col);
          d = (rng_u32(key, (uint64_t)(zoff + (long)o)) >= c.drop.thresh) ? y * c.drop.scale : 0.f;
        }
        t_st(Hd, tHD, rowHD(rb, v, l), d + xv[rb][v]);
      }
    if constexpr (NP > 0) {
      // ---- push: Pn_l' += Y_l Wd_l'[l gh : (l + 1) gh, :] for every later sub-layer, 16 k per chunk --------------------
      TR(12 + 8 * l);
      sstore(0, wr[0]);
      t_barrier();   // Y_l's image and the first weight chunk are complete
      TR(13 + 8 * l);
      auto compute = [&](const int ch, const int st) __attribute__((always_inline)) {
        t4 a[NRB > 0 ? NRB : 1];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const t4*>(Ysj + 16 * rb * P + 16 * ch);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float* wb = Ws + ((st * (L - 1) + p) * 16 + 4 * g) * P + col;
          float bv[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) bv[v] = wb[v * P];
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) Pa[l + 1 + p][rb] = mfma16(a[rb][v], bv[v], Pa[l + 1 + p][rb]);
        }
      };
#pragma nounroll
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
        for (int v = 0; v < 4; ++v) t_st(Pd, tHD, rowHD(rb, v, l + 1), Pa[l + 1][rb][v]);
    }
  };
  static_assert(L >= 1 && L <= 4, "sub-layers are unrolled by hand");
  layer(std::integral_constant<int, 0>());
  if constexpr (L > 1) layer(std::integral_constant<int, 1>());
  if constexpr (L > 2) layer(std::integral_constant<int, 2>());
  if constexpr (L > 3) layer(std::integral_constant<int, 3>());
  TR(50);
}

// FULL: N == 64 and no n_valid -- one body, no switch (its code IS the four-block body of the other instantiation).  Otherwise the workgroup picks the body of its document's row-block
// count (a DocRED batch padded to 42 entities averages 20 real ones: two blocks instead of three).
template <int GH, int L, bool FULL>
__global__ __launch_bounds__(4 * GH) void gcn_chain_t_fwd_kernel(const GcnCtx c) {
  constexpr int W = GH / 16;
  __shared__ __attribute__((aligned(16))) float lds[t_fwd_lds<GH, L>()];
  if (blockIdx.x >= c.B * c.H) {  // passenger workgroup: one entity row of the riding edge mean
    const EdgeRide& r = c.ride;
    edge_fwd_row<4, false, true, W>(r.in, nullptr, r.n_valid, r.out, nullptr, nullptr, nullptr, Drop(), r.N, r.D,
                                    blockIdx.x - c.B * c.H, lds);
    return;
  }
  const int z = blockIdx.x, b = z / c.H, h = z - b * c.H;
  if constexpr (FULL) {
    chain_t_fwd_body<GH, L, 4>(c, lds, z, b, h);
  } else {
    const int nv = c.n_valid ? min(max(c.n_valid[b], 0), c.N) : c.N;
    const int nrb = (nv + 15) >> 4;
    switch (nrb) {
      case 0: chain_t_fwd_body<GH, L, 0>(c, lds, z, b, h); break;
      case 1: chain_t_fwd_body<GH, L, 1>(c, lds, z, b, h); break;
      case 2: chain_t_fwd_body<GH, L, 2>(c, lds, z, b, h); break;
      case 3: chain_t_fwd_body<GH, L, 3>(c, lds, z, b, h); break;
      default: chain_t_fwd_body<GH, L, 4>(c, lds, z, b, h); break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, last sub-layer first:
//   dS = dY_l [Y_l > 0];  dM_l = dS rinv;  drow -= rinv sum_c dS Y_l;  dPn_l = A_h^T dM_l;  dA += dM_l Pn_l^T;
//   dY_l' += dPn_l Wd_l[l' gh : (l' + 1) gh, :]^T for l' < l   (dY_l' starts as dropout_bwd(dHO_l'), read from dYa)
// dA: wave w accumulates rows 16 (w % 4) .. + 15, all 64 columns, over the k range [64 (w / 4), + 64) of every sub-layer in
// registers; the gh / 64 partial sums meet in LDS at the end, in a fixed order (bitwise reproducible).  gh = 32 (two waves: the
// BERT model's width, hidden 128 over four sub-layers, bert:237,247-248): each wave takes two row blocks over the whole k range.
// A wave whose row block lies past the real entities' runs its dA products all the same (on zeros: its SIMD's matrix pipe
// has nothing else to do -- row block = wave % 4 = SIMD): no wave-dependent branch around an MFMA anywhere.
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

// FUSE: the gradient of the block's output-projection input is computed here instead of arriving in dYa (what
// gcn_chain_s_bwd<true> does at its one shape, chain.hip):  dHO_{b,h} = dout_b Wlin[:, h]  (N x D x D per pair, Wlin's slice
// streamed through LDS), dY = dropout_bwd(dHO), dXres_b = sum_h dHO_{b,h} = dout_b (sum_h Wlin[:, h]): this workgroup's D / H
// columns of it (one head: dHO itself); dout is masked / un-dropped while it is staged (and written back as dout_m for dWlin),
// its column sums (the output bias gradient, stage 1) leave from the image.  Removes the dHO launch, the mask launch and the
// head-sum / dropout kernel (glove:74-78 / 111-118): five launches of a ragged cfg-2 step's 27, four of cfg 1's 24.
template <int GH, int L, int NRB, bool FUSE>
__device__ __forceinline__ void chain_t_bwd_body(const GcnCtx& c, float* __restrict__ lds, const int z, const int b, const int h, const int nv) {
  constexpr int REG = FUSE ? t_bwd_region<GH, L>() : 2 * 64 * (GH + 4);       // floats of the image region
  constexpr int W = GH / 16, NT = 4 * GH, P = GH + 4, NC = GH / 16, SP = 20;   // SP: row pitch of a [gh][16 k] weight stage
  static_assert(2 * GH * SP <= 64 * P, "weight stages live in the Pn image");
  static_assert(W <= 4 || (W / 4 - 1) * 4096 <= 64 * P, "dA exchange lives in the dM image");
  constexpr int OBW = W >= 4 ? 1 : 4 / W;          // row blocks of dA per wave (fewer than four waves: several each)
  constexpr int KS = GH >= 64 ? 4 : GH / 16;       // 16-deep k steps of a wave's k range (64 features, or all of a narrow sub-layer)
  TRB(100);
  float* const ATs = lds;                  // A_h transposed: [k = column of A][row of A]
  float* const Ds = ATs + 64 * T_LA;       // dM_l, then dPn_l
  float* const Ps = Ds + 64 * P;           // Pn_l, then the weight stages
  float* const Tp = Ds + REG;              // [W][64] per-wave partial row sums
  float* const Ts = Tp + W * 64;           // gradient of the normaliser's row sums
  float* const Rs = Ts + 64;               // rinv
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 15, g = lane >> 4;
  const int N = c.N;
  const unsigned HD = (unsigned)c.HD;
  const long zoff = (long)b * N * c.HD + (long)h * c.D;
  const float* __restrict__ Ag = c.A + (long)z * N * N;
  const unsigned Dw = (unsigned)c.D;
  const t_desc Yd = t_buffer(c.Y + zoff, N, HD, Dw), Gd = t_buffer(c.dYa + zoff, N, HD, Dw);
  const t_desc Md = t_buffer(c.dM + zoff, N, HD, Dw), Qd = t_buffer(c.dP + zoff, N, HD, Dw);
  // the Pn image is fetched by thread index, not by row block: its descriptor ends at the last real entities' block
  // (rows past it may never have been written; they read as zero)
  const t_desc Pd = t_buffer(c.Pn + zoff, min(N, 16 * NRB), HD, Dw);
  const int col = 16 * w + j;
  const int ob = w & 3, ks = w >> 2;       // dA: this wave's (first) row block and k range; further blocks: ob + W u
  const unsigned tHD = ((unsigned)(4 * g) * HD + (unsigned)col) * 4u;   // this thread's byte offset in an [., HD] tensor
  auto rowHD = [&](const int rb, const int v, const int l) { return ((unsigned)(16 * rb + v) * HD + (unsigned)(l * GH)) * 4u; };   // uniform, bytes
  // LDS: one address register per image (the thread's (row 4 g, column col) element, or its operand row j / k group g),
  // everything else in the instructions' immediates (see the forward body)
  const t_desc Wdd = t_buffer(c.flat + c.oWd + (long)h * c.wd_head, 1, 0, (unsigned)c.wd_head);   // this head's dense-connection weights
  const unsigned wvoff = ((unsigned)(t >> 2) * GH + 4u * (unsigned)(t & 3)) * 4u;
  float* const Dsg = Ds + 4 * g * P + col;
  float* const Tpg = Tp + w * 64 + 4 * g;
  const float* const Rsg = Rs + 4 * g;
  const float* const ATj = ATs + j * T_LA + 4 * g;
  const float* const Dsj = Ds + j * P + 4 * g;
  const float* const Psj = Ps + j * P + 4 * g;

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
  // Both start at zero where the LAST sub-layer (the first one processed) first touches them, not here: 64 registers of
  // zeros carried through that sub-layer's first half were what pushed the widest instantiation into scratch.
  t4 Da[L > 1 ? L - 1 : 1][4];             // dY_l' contributions pushed by later sub-layers (l' = 0 .. L - 2)
  t4 dacc[OBW][4];
  constexpr int PV = (64 * (GH / 4)) / NT;  // 16-byte pieces of a 64 x gh image per thread (= 4)
  static_assert(PV * NT == 64 * (GH / 4) && NT == 16 * (GH / 4), "image load mapping");
  t4 dyf[FUSE ? L : 1][4];                  // FUSE: dY_l of every sub-layer (this wave's strip), computed in front of the sub-layers
  const int prow = t / (GH / 4), pc4 = (t - prow * (GH / 4)) * 4;          // a thread's first piece of the Pn image
  const unsigned pvoff = ((unsigned)prow * HD + (unsigned)pc4) * 4u;
  float* const Psp = Ps + prow * P + pc4;

  auto layer = [&](auto lt) __attribute__((always_inline)) {
    constexpr int l = decltype(lt)::value;
    TRB(110 + 8 * l);
    // ---- requests: Y_l, dY_l (this wave's strip) and the Pn_l image ------------------------------------------------------
    float yv[4][4], dy[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        // (dYa's rows past the real entities' blocks may never have been written)
        yv[rb][v] = rb < NRB ? t_ld(Yd, tHD, rowHD(rb, v, l)) : 0.f;
        if constexpr (FUSE) dy[rb][v] = dyf[l][rb][v];
        else dy[rb][v] = rb < NRB ? t_ld(Gd, tHD, rowHD(rb, v, l)) : 0.f;
      }
    t4 pn[PV];   // piece u of a thread: 16 u rows below its first one (NT threads cover 16 rows), same columns
#pragma unroll
    for (int u = 0; u < PV; ++u) pn[u] = t_ld4(Pd, pvoff, ((unsigned)(16 * u) * HD + (unsigned)(l * GH)) * 4u);
    // ---- through Y = relu(S), S = M rinv:  dS = dY [Y > 0];  dM = dS rinv;  drow -= rinv sum_c dS Y --------------------
    t4 dm[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float gsel = dy[rb][v];
        if constexpr (l < L - 1) gsel += Da[l][rb][v];
        gsel = yv[rb][v] > 0.f ? gsel : 0.f;
        const float m = gsel * Rsg[16 * rb + v];
        dm[rb][v] = m;
        Dsg[(16 * rb + v) * P] = m;
        t_st(Md, tHD, rowHD(rb, v, l), m);
        const float part = row16_sum(gsel * yv[rb][v]);
        if (j == 0) Tpg[16 * rb + v] = part;
      }
#pragma unroll
    for (int u = 0; u < PV; ++u) *reinterpret_cast<t4*>(Psp + 16 * u * P) = pn[u];
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
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (rb < NRB && kb < NRB) {
          const t4 a = *reinterpret_cast<const t4*>(ATj + 16 * rb * T_LA + 16 * kb);
#pragma unroll
          for (int v = 0; v < 4; ++v) q[rb] = mfma16(a[v], dm[kb][v], q[rb]);
        }
      }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int v = 0; v < 4; ++v) t_st(Qd, tHD, rowHD(rb, v, l), q[rb][v]);
    TRB(113 + 8 * l);
    // first weight chunks of the push below: requested now, they land while dA's share runs
    constexpr int NQ = l * NC;                         // 16-deep chunks of the push: (l', chunk) flattened
    t4 wr[2];
    // Wd_l: [l gh rows (l', n')][gh]; chunk (l', ch): columns [16 ch, 16 ch + 16) of rows l' gh .. + gh -- thread t takes
    // four of row (t >> 2)'s
    auto gload = [&](const int qi, t4& d) __attribute__((always_inline)) {
      const int lp = qi / NC, ch = qi - lp * NC;
      d = t_ld4(Wdd, wvoff, ((unsigned)(GH * GH * (l * (l - 1) / 2)) + (unsigned)lp * (GH * GH) + 16u * (unsigned)ch) * 4u);
    };
    auto sstore = [&](const int st, const t4& d) __attribute__((always_inline)) {
      *reinterpret_cast<t4*>(Ps + st * GH * SP + (t >> 2) * SP + 4 * (t & 3)) = d;
    };
    if constexpr (l > 0) {
      gload(0, wr[0]);
      gload(1, wr[1]);
    }
    // ---- dA += dM_l Pn_l^T: rows 16 ob .. + 15, k range [64 ks, 64 ks + 64) ------------------------------------------------
    if constexpr (l == L - 1) {
#pragma unroll
      for (int u = 0; u < OBW; ++u)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) dacc[u][jb] = t4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (NRB > 0) {
#pragma unroll
      for (int u = 0; u < OBW; ++u) {
        const int obu = ob + W * u;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const t4 a = *reinterpret_cast<const t4*>(Dsj + 16 * obu * P + 64 * ks + 16 * s);
#pragma unroll
          for (int jb = 0; jb < NRB; ++jb) {
            const t4 bq = *reinterpret_cast<const t4*>(Psj + 64 * ks + 16 * jb * P + 16 * s);
#pragma unroll
            for (int v = 0; v < 4; ++v) dacc[u][jb] = mfma16(a[v], bq[v], dacc[u][jb]);
          }
        }
      }
      // The products end HERE.  Nothing reads dacc before the next sub-layer, and left alone the compiler sinks these MFMAs
      // behind the push loop below -- carrying their twenty LDS operands through it instead of four accumulators (the
      // widest instantiations then ran out of registers: 16 spilled at gh = 256).
#pragma unroll
      for (int u = 0; u < OBW; ++u)
#pragma unroll
        for (int jb = 0; jb < NRB; ++jb) t_pin(dacc[u][jb]);
    }
    TRB(114 + 8 * l);
    if constexpr (l > 0) {
      t_barrier();   // everybody is done with the dM_l and Pn_l images
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) Dsg[(16 * rb + v) * P] = q[rb][v];
      sstore(0, wr[0]);
      t_barrier();   // dPn_l's image and the first weight chunk are complete
      TRB(115 + 8 * l);
      // ---- push: dY_l' += dPn_l Wd_l[l' gh + n', k]^T for l' < l ---------------------------------------------------------
      // One loop per target sub-layer p (its accumulators are named at compile time: no branch selects them), the weight
      // chunks prefetched straight across the seams between the loops (qi = (p, chunk) flattened).
      auto compute = [&](auto pt, const int ch, const int st) __attribute__((always_inline)) {
        constexpr int p = decltype(pt)::value;
        t4 a[NRB > 0 ? NRB : 1];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const t4*>(Dsj + 16 * rb * P + 16 * ch);
        const t4 bq = *reinterpret_cast<const t4*>(Ps + st * GH * SP + col * SP + 4 * g);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) Da[p][rb] = mfma16(a[rb][v], bq[v], Da[p][rb]);
      };
      auto target = [&](auto pt) __attribute__((always_inline)) {
        constexpr int p = decltype(pt)::value;
        if constexpr (l == L - 1) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) Da[p][rb] = t4{0.f, 0.f, 0.f, 0.f};
        }
#pragma nounroll
        for (int ch = 0; ch < NC; ch += 2) {
          const int qi = p * NC + ch;
          if (qi + 2 < NQ) gload(qi + 2, wr[0]);
          compute(pt, ch, 0);
          sstore(1, wr[1]);
          t_barrier();
          if (qi + 3 < NQ) gload(qi + 3, wr[1]);
          compute(pt, ch + 1, 1);
          if (qi + 2 < NQ) sstore(0, wr[0]);
          t_barrier();
        }
      };
      target(std::integral_constant<int, 0>());
      if constexpr (l > 1) target(std::integral_constant<int, 1>());
      if constexpr (l > 2) target(std::integral_constant<int, 2>());
    }
  };
  if constexpr (FUSE) {
    constexpr int D = L * GH, PX = D + 4, DV = (64 * (D / 4)) / NT;   // DV: 16-byte pieces of the dout image per thread (= 4 L)
    static_assert(DV * NT == 64 * (D / 4) && NT % (D / 4) == 0, "dout image load mapping");
    constexpr int DRS = NT / (D / 4);  // rows of the image between a thread's pieces
    float* const Xs = Ds;             // [64][PX]   dout_b, masked / un-dropped
    float* const Bs = Xs + 64 * PX;   // [2][16][PX] Wlin[16 k, this head's D columns]
    float* const Es = Rs + 64;        // [W][64 lanes][16] K slices of this head's share of dXres
    const t_desc Xrd = t_buffer(c.dXres + (long)b * N * D, N, D, D);
    {  // ---- stage dout_b (padding rows get no gradient; back through the hop's output dropout, glove:341) ------------------
      const t_desc Dgd = t_buffer(c.dout + (long)b * N * D, N, D, D);
      const t_desc Dmd = t_buffer(c.dout_m ? c.dout_m + (long)b * N * D : c.dout, c.dout_m ? N : 0, D, D);
      const bool od = c.dout_m && c.odrop.snap != nullptr;
      const uint64_t okey = od ? drop_key(c.odrop) : 0;
      const int drow = t / (D / 4), dc4 = (t - drow * (D / 4)) * 4;
      const unsigned dvoff = ((unsigned)drow * D + (unsigned)dc4) * 4u;
      t4 dv[DV];
#pragma unroll
      for (int u = 0; u < DV; ++u) dv[u] = t_ld4(Dgd, dvoff, (unsigned)(DRS * u) * D * 4u);
#pragma unroll
      for (int u = 0; u < DV; ++u) {
        const int row = drow + DRS * u;
        if (c.dout_m) {
          const bool keep = row < nv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = keep ? dv[u][e] : 0.f;
            if (od) x = (rng_u32(okey, (uint64_t)(((long)b * N + row) * D + dc4 + e)) >= c.odrop.thresh) ? x * c.odrop.scale : 0.f;
            dv[u][e] = x;
          }
          if (h == 0) t_st4(Dmd, dvoff, (unsigned)(DRS * u) * D * 4u, dv[u]);
        }
        *reinterpret_cast<t4*>(Xs + row * PX + dc4) = dv[u];
      }
    }
    // ---- dHO_b = dout_b Wlin[:, h D .. h D + D): every sub-layer's strip of this wave, 16 k per chunk ----------------------
    const t_desc Wld = t_buffer(c.flat + c.oWlin + (long)h * D, D, HD, D);     // Wlin's rows k, this head's D columns
    const int wkr = t / (D / 4), wc4 = (t - wkr * (D / 4)) * 4;                // a thread's piece of a 16 x D chunk
    constexpr int WV = (16 * (D / 4) + NT - 1) / NT;                            // pieces per thread and chunk
    const unsigned wvoff = ((unsigned)wkr * HD + (unsigned)wc4) * 4u;
    t4 wr[2][WV];
    auto gload = [&](const int ch, t4 (&d)[WV]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < WV; ++p) d[p] = t_ld4(Wld, wvoff, (unsigned)(16 * ch + DRS * p) * HD * 4u);
    };
    auto sstore = [&](const int st, const t4 (&d)[WV]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < WV; ++p)
        if (wkr + DRS * p < 16) *reinterpret_cast<t4*>(Bs + (st * 16 + wkr + DRS * p) * PX + wc4) = d[p];
    };
    gload(0, wr[0]);
    gload(1, wr[1]);
    TRB(101);
    t_barrier();   // the dout image (and A^T, rinv) are complete
    TRB(102);
    if (c.colpart && h == 0) {  // the output bias gradient's column sums of this document: rows 0-31 and 32-63
      for (int idx = t; idx < 2 * D; idx += NT) {
        const int half = idx / D, cc = idx - half * D;
        float sacc = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) sacc += Xs[(half * 32 + i) * PX + cc];
        c.colpart[(long)(2 * b + half) * D + cc] = sacc;
      }
    }
    // this head's D / H columns of dXres_b = dout_b Wsum (H > 1): DH / 16 column groups, each K-split over W / (DH / 16) waves
    // (the host only fuses shapes where these divide: chain_t_bwd_fusable); the slices meet in Es behind the product's barriers
    const int DH = D / c.H, ncg = max(DH >> 4, 1), kw = max(W / ncg, 1), cgi = w % ncg, ksl = w / ncg, klen = D / kw;
    t4 xa[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) xa[rb] = t4{0.f, 0.f, 0.f, 0.f};
    if (c.H != 1) {
      const float* __restrict__ Wsp = c.Wsum + (unsigned)(ksl * klen + 4 * g) * (unsigned)D + (unsigned)(h * DH + cgi * 16 + j);
      for (int k0 = 0; k0 < klen; k0 += 16) {
        float bw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) bw[v] = Wsp[(unsigned)(k0 + v) * (unsigned)D];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          const t4 a = *reinterpret_cast<const t4*>(Xs + (16 * rb + j) * PX + ksl * klen + k0 + 4 * g);
#pragma unroll
          for (int v = 0; v < 4; ++v) xa[rb] = mfma16(a[v], bw[v], xa[rb]);
        }
      }
      if (ksl > 0) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int v = 0; v < 4; ++v) Es[w * 1024 + (rb * 4 + v) * 64 + lane] = xa[rb][v];
      }
    }
    t4 acc[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[l][rb] = t4{0.f, 0.f, 0.f, 0.f};
    const float* const Xsj = Xs + j * PX + 4 * g;
    auto compute = [&](const int ch, const int st) __attribute__((always_inline)) {
      t4 a[NRB > 0 ? NRB : 1];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const t4*>(Xsj + 16 * rb * PX + 16 * ch);
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float* wb = Bs + (st * 16 + 4 * g) * PX + l * GH + col;
        float bv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) bv[v] = wb[v * PX];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[l][rb] = mfma16(a[rb][v], bv[v], acc[l][rb]);
      }
    };
    constexpr int NCH = D / 16;
    static_assert(NCH % 2 == 0, "chunks are taken in pairs");
    TRB(103);
    sstore(0, wr[0]);
    t_barrier();
    TRB(104);
#pragma nounroll
    for (int ch = 0; ch < NCH; ch += 2) {   // two chunks per trip: register sets and stages are compile-time constants
      if (ch + 2 < NCH) gload(ch + 2, wr[0]);
      compute(ch, 0);
      sstore(1, wr[1]);
      t_barrier();
      if (ch + 3 < NCH) gload(ch + 3, wr[1]);
      compute(ch + 1, 1);
      if (ch + 2 < NCH) sstore(0, wr[0]);
      t_barrier();
    }
    TRB(105);
    if (c.H != 1 && ksl == 0) {   // (the slices were written before the product's first barrier)
      const unsigned xvoff = ((unsigned)(4 * g) * D + (unsigned)(h * DH + cgi * 16 + j)) * 4u;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float x = xa[rb][v];
          for (int s2 = 1; s2 < kw; ++s2) x += Es[(s2 * ncg + cgi) * 1024 + (rb * 4 + v) * 64 + lane];
          t_st(Xrd, xvoff, (unsigned)(16 * rb + v) * D * 4u, x);
        }
    }
    // dY_l = dropout_bwd(dHO_l) (the forward mask: same site, same element offsets); one head: dXres = dHO itself
    const bool dd = c.drop.snap != nullptr;
    const uint64_t key = dd ? drop_key(c.drop) : 0;
    const unsigned xv1 = ((unsigned)(4 * g) * D + (unsigned)col) * 4u;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        t_pin(acc[l][rb]);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * rb + 4 * g + v;
          float x = acc[l][rb][v];
          if (c.H == 1) t_st(Xrd, xv1, ((unsigned)(16 * rb + v) * D + (unsigned)(l * GH)) * 4u, x);
          if (dd) x = (rng_u32(key, (uint64_t)(zoff + (long)((unsigned)row * HD + (unsigned)(l * GH + col)))) >= c.drop.thresh) ? x * c.drop.scale : 0.f;
          dyf[l][rb][v] = x;
        }
      }
  }
  TRB(106);
  t_barrier();
  static_assert(L >= 1 && L <= 4, "sub-layers are unrolled by hand");
  if constexpr (L > 3) layer(std::integral_constant<int, 3>());
  if constexpr (L > 2) layer(std::integral_constant<int, 2>());
  if constexpr (L > 1) layer(std::integral_constant<int, 1>());
  layer(std::integral_constant<int, 0>());
  TRB(150);
  // ---- dA = sum over the k ranges + drow (every column of a row); drow itself ---------------------------------------------
  t_barrier();   // sub-layer 0 is done with the images; Ts is final
  if constexpr (W > 4) {
    if (ks > 0) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int v = 0; v < 4; ++v) Ds[((((ks - 1) * 4 + ob) * 4 + jb) * 4 + v) * 64 + lane] = dacc[0][jb][v];
    }
    t_barrier();
  }
  float* __restrict__ dAg = c.dA + (long)z * N * N;
  if (ks == 0) {
#pragma unroll
    for (int u = 0; u < OBW; ++u) {
      const int obu = ob + W * u;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * obu + 4 * g + v, cc = 16 * jb + j;
          float s = dacc[u][jb][v];
#pragma unroll
          for (int k2 = 1; k2 < W / 4; ++k2) s += Ds[((((k2 - 1) * 4 + obu) * 4 + jb) * 4 + v) * 64 + lane];
          // (columns and rows past the real entities' blocks: no product contributed, s is an exact zero there)
          if (row < N && cc < N) dAg[row * N + cc] = s + Ts[row];
        }
    }
  }
  if (t < N) c.drow[(long)z * N + t] = Ts[t];
  TRB(151);
}

template <int GH, int L, bool FULL, bool FUSE>
__global__ __launch_bounds__(4 * GH) void gcn_chain_t_bwd_kernel(const GcnCtx c, const GemmGroup4 cg, const int npw) {
  constexpr int W = GH / 16;
  constexpr int LDSF = FUSE ? t_bwd_fuse_lds<GH, L>() : t_bwd_lds<GH>();
  static_assert(LDSF * sizeof(float) <= 160 * 1024, "LDS of one compute unit");
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  static_assert((GH / 64) * T_TEAM_LDS <= LDSF, "parked tiles use the chain kernel's LDS");
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
  const int z = blockIdx.x, b = z / c.H, h = z - b * c.H;
  const int nv = c.n_valid ? min(max(c.n_valid[b], 0), c.N) : c.N;
  if constexpr (FULL) {
    chain_t_bwd_body<GH, L, 4, FUSE>(c, lds, z, b, h, nv);
  } else {
    const int nrb = (nv + 15) >> 4;
    switch (nrb) {
      case 0: chain_t_bwd_body<GH, L, 0, FUSE>(c, lds, z, b, h, nv); break;
      case 1: chain_t_bwd_body<GH, L, 1, FUSE>(c, lds, z, b, h, nv); break;
      case 2: chain_t_bwd_body<GH, L, 2, FUSE>(c, lds, z, b, h, nv); break;
      case 3: chain_t_bwd_body<GH, L, 3, FUSE>(c, lds, z, b, h, nv); break;
      default: chain_t_bwd_body<GH, L, 4, FUSE>(c, lds, z, b, h, nv); break;
    }
  }
}

template <int GH, int L>
static int chain_t_run_fwd(const GcnCtx& c, dim3 grid, double fl, hipStream_t st) {
  if (chain_t_full(c)) GC_LAUNCH_TIMED("gcn_chain_fwd", fl, (gcn_chain_t_fwd_kernel<GH, L, true>), grid, dim3(4 * GH), 0, st, c);
  else GC_LAUNCH_TIMED("gcn_chain_fwd", fl, (gcn_chain_t_fwd_kernel<GH, L, false>), grid, dim3(4 * GH), 0, st, c);
  return check_launch("gcn_chain_t_fwd");
}
template <int GH, int L>
static int chain_t_run_bwd(const GcnCtx& c, const GemmGroup4& cg, int npw, dim3 grid, double fl, hipStream_t st) {
  if constexpr (t_fuse_shape<GH, L>()) {
    if (c.dout) {   // the fused output-projection gradient (chain_t_bwd_fusable said yes)
      if (chain_t_full(c)) GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, true, true>), grid, dim3(4 * GH), 0, st, c, cg, npw);
      else GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, false, true>), grid, dim3(4 * GH), 0, st, c, cg, npw);
      return check_launch("gcn_chain_t_bwd");
    }
  }
  if (chain_t_full(c)) GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, true, false>), grid, dim3(4 * GH), 0, st, c, cg, npw);
  else GC_LAUNCH_TIMED("gcn_chain_bwd", fl, (gcn_chain_t_bwd_kernel<GH, L, false, false>), grid, dim3(4 * GH), 0, st, c, cg, npw);
  return check_launch("gcn_chain_t_bwd");
}

// (gh, L) pairs the templates are instantiated for: the reference's model (64, 2), cfg 2's width (128, 2), cfg 3 (192, 4),
// and the neighbours a user is most likely to configure.  Four translation units (chain_t_{a,b,c,d}.hip) share them so that
// the build runs them side by side; X(gh, L, unit)
#define GC_CHAIN_T_SHAPES(X) \
  X(32, 2, 0) X(32, 4, 0) X(64, 1, 0) X(64, 2, 0) X(64, 3, 0) X(64, 4, 0) \
  X(128, 1, 1) X(128, 2, 1) X(128, 3, 1) X(128, 4, 1) \
  X(192, 2, 2) X(192, 4, 2) \
  X(256, 1, 3) X(256, 2, 3)

// chain_t_u<U>.hip:  GC_CHAIN_T_UNIT(U)  defines the two entry points of unit U (1 = launched, *rc set; 0 = not its shape)
template <int U>
static int chain_t_unit_fwd(const GcnCtx& c, dim3 grid, double fl, hipStream_t st, int* rc) {
#define X(gh_, l_, u_)                                    \
  if constexpr (u_ == U) {                                \
    if (c.gh == gh_ && c.L == l_) {                       \
      *rc = chain_t_run_fwd<gh_, l_>(c, grid, fl, st);    \
      return 1;                                           \
    }                                                     \
  }
  GC_CHAIN_T_SHAPES(X)
#undef X
  return 0;
}
template <int U>
static int chain_t_unit_bwd(const GcnCtx& c, const GemmGroup4& cg, int npw, dim3 grid, double fl, hipStream_t st, int* rc) {
#define X(gh_, l_, u_)                                             \
  if constexpr (u_ == U) {                                         \
    if (c.gh == gh_ && c.L == l_) {                                \
      *rc = chain_t_run_bwd<gh_, l_>(c, cg, npw, grid, fl, st);    \
      return 1;                                                    \
    }                                                              \
  }
  GC_CHAIN_T_SHAPES(X)
#undef X
  return 0;
}
#ifdef GC_T_TRACE
#define GC_CHAIN_T_TRACE_EXPORT(U) \
  extern "C" int gcgcn_debug_trace_t_##U(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc::gc_trace_t), sizeof(long long) * 256); }
#else
#define GC_CHAIN_T_TRACE_EXPORT(U)
#endif
#define GC_CHAIN_T_UNIT(U)                                                                                                          \
  namespace gc {                                                                                                                    \
  int chain_t_fwd_unit##U(const GcnCtx& c, dim3 grid, double fl, hipStream_t st, int* rc) { return chain_t_unit_fwd<U>(c, grid, fl, st, rc); } \
  int chain_t_bwd_unit##U(const GcnCtx& c, const GemmGroup4& cg, int npw, dim3 grid, double fl, hipStream_t st, int* rc) {          \
    return chain_t_unit_bwd<U>(c, cg, npw, grid, fl, st, rc);                                                                       \
  }                                                                                                                                 \
  }                                                                                                                                 \
  GC_CHAIN_T_TRACE_EXPORT(U)

}  // namespace gc
