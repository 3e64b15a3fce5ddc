// The 128 x 128 tile body of the fp32 MFMA GEMM for big interior problems (M % 128 == N % 128 == 0, K % 32 == 0, rows 16-byte
// aligned).  Same contract as gemm_body (gemm_body.hpp: GemmArgs, batch strides, split-K slabs, fused epilogue), different
// inner structure -- what the counters said the 64 x 64 body lacks on big products (cfg 3 / cfg 5 group launches, random data:
// 93-106 TF/s where a library kernel reaches 117-137; the 64 x 64 body is clock-limited there, 28 % faster on all-zero
// operands, i.e. it spends too much energy per flop on LDS and L2 traffic and on address arithmetic):
//   * four waves, each a 64 x 64 quarter as 4 x 4 accumulators of v_mfma_f32_16x16x4_f32 (64 registers): half the LDS and
//     L2 bytes per flop of the 32 x 32-per-wave tiling;
//   * operand fragments by ds_read_b128 only: 8 reads per 64 MFMAs.  An operand whose k index is contiguous in memory sits in
//     LDS as [128 rows][32 k] with its 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free for the four 16-lane
//     groups of ds_read_b128, MI355X_MICROARCH.md LDS table), and a lane's read gives it FOUR CONSECUTIVE k of one row: the
//     k sub-steps of four MFMAs.  An operand whose m / n index is contiguous sits as [32 k][128], unpadded, and a lane's
//     read gives it four consecutive rows / columns at ONE k: the same sub-step of four different accumulator blocks -- block
//     b of a wave is then rows {4 i + b}, not {16 b + i}; the epilogue knows (and stores four consecutive columns at once).
//     Either way lane group kq supplies k = 16 t + 4 kq + s at sub-step s of k group t, for both operands;
//   * staging through ONE register set per operand (4 x 16-byte loads each per k-tile), written to the other LDS stage half way
//     through the current tile's MFMAs and requested again at once: a full tile of flight, no wait at the tile's end but the barrier;
//   * two workgroups per compute unit (64 KB of LDS, <= 256 registers): one's prologue / epilogue under the other's k-loop.
#pragma once
#include "gemm_body.hpp"

namespace gc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BK = 32;
constexpr int GB_IMG = 128 * GB_BK;        // floats of one operand image
constexpr int GB_STAGE = 2 * GB_IMG;       // A image + B image
constexpr int GB_LDS = 2 * GB_STAGE;       // two stages: 64 KB

// float offset of 16-byte chunk c (0..7) of row r in a k-contiguous image
__device__ __forceinline__ int gb_kc_off(int r, int c) { return r * GB_BK + ((c ^ ((r >> 1) & 7)) << 2); }

template <bool KC>
struct GbStage {  // one thread's share of an operand's k-tile: 4 x 16 bytes
  f32x4 v[4];
  // src: the operand's first element of this tile row / column range at k = 0 (batch offsets applied)
  __device__ __forceinline__ void load(const float* __restrict__ src, long ld, int mn0, int k0, int t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = t + 256 * q;
      const float* p = KC ? src + (long)(mn0 + (f >> 3)) * ld + k0 + ((f & 7) << 2)
                          : src + (long)(k0 + (f >> 5)) * ld + mn0 + ((f & 31) << 2);
      v[q] = *reinterpret_cast<const f32x4*>(p);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ img, int t) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = t + 256 * q;
      const int off = KC ? gb_kc_off(f >> 3, f & 7) : (f >> 5) * 128 + ((f & 31) << 2);
      *reinterpret_cast<f32x4*>(img + off) = v[q];
    }
  }
};

// The fragments of one 16-deep k group for a wave's 64 rows (or columns): x[a][b] with the meaning
//   KC : x[block][sub-step]     (one read per block)          !KC : x[sub-step][block]   (one read per sub-step)
template <bool KC>
struct GbFrag {
  f32x4 x[4];
  __device__ __forceinline__ void read(const float* __restrict__ img, int w64, int kgroup, int lane) {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int off = KC ? gb_kc_off(w64 + 16 * u + i, 4 * kgroup + kq) : (16 * kgroup + 4 * kq + u) * 128 + w64 + 4 * i;
      x[u] = *reinterpret_cast<const f32x4*>(img + off);
    }
  }
  __device__ __forceinline__ float at(int block, int s) const { return KC ? x[block][s] : x[s][block]; }
};

template <bool AKC, bool BKC>
__device__ __forceinline__ void gemm_big_body(const GemmArgs& g, float* __restrict__ lds, const int bx, const int by, const int zs) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
  const int z = zs / g.splits, sp = zs - z * g.splits;
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const int m0 = by * 128, n0 = bx * 128;
  const int kbeg = sp * g.ksplit;
  const int kend = min(g.K, kbeg + g.ksplit);
  const int nk = (kend - kbeg) / GB_BK;
  const float* __restrict__ A = g.A + z1 * g.sA1 + z2 * g.sA2;
  const float* __restrict__ B = g.B + z1 * g.sB1 + z2 * g.sB2;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  GbStage<AKC> sa;
  GbStage<BKC> sb;
  auto kof = [&](int kt) { return kbeg + min(kt, nk - 1) * GB_BK; };   // clamped: the extra requests are never stored
  sa.load(A, g.lda, m0, kof(0), t);
  sb.load(B, g.ldb, n0, kof(0), t);
  sa.store(lds, t);
  sb.store(lds + GB_IMG, t);
  __builtin_amdgcn_sched_barrier(0);
  sa.load(A, g.lda, m0, kof(1), t);
  sb.load(B, g.ldb, n0, kof(1), t);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // one k-tile from stage `cur`; half way through, tile kt + 1 goes to the other stage and tile kt + 2 is requested
  auto tile = [&](const int cur, const int kt, const bool stage_next) {
    const float* ai = lds + cur * GB_STAGE;
    const float* bi = ai + GB_IMG;
    GbFrag<AKC> fa0, fa1;
    GbFrag<BKC> fb0, fb1;
    fa0.read(ai, wr, 0, lane);
    fb0.read(bi, wc, 0, lane);
    fa1.read(ai, wr, 1, lane);
    fb1.read(bi, wc, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0.at(a, s), fb0.at(b, s), acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (stage_next) {
      float* ao = lds + (cur ^ 1) * GB_STAGE;
      sa.store(ao, t);
      sb.store(ao + GB_IMG, t);
      __builtin_amdgcn_sched_barrier(0);
      sa.load(A, g.lda, m0, kof(kt + 2), t);
      sb.load(B, g.ldb, n0, kof(kt + 2), t);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1.at(a, s), fb1.at(b, s), acc[a][b], 0, 0, 0);
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    tile(0, kt, true);
    __syncthreads();
    tile(1, kt + 1, true);       // (for the last pair this restages a clamped duplicate nobody reads)
    __syncthreads();
  }
  if (kt < nk) tile(0, kt, false);

  // ---- store ------------------------------------------------------------------------------------------------------
  // acc[a][b][v] of lane (j = lane & 15, gq = lane >> 4) is local row 4 gq + v, local column j of block (a, b)
  const int j = lane & 15, gq = lane >> 4;
  auto row_of = [&](int a, int v) { return m0 + wr + (AKC ? 16 * a + 4 * gq + v : 4 * (4 * gq + v) + a); };
  auto col_of = [&](int b) { return n0 + wc + (BKC ? 16 * b + j : 4 * j + b); };
  if (g.splits > 1) {  // raw partial sums -> workspace [split][batch][M][N]
    float* __restrict__ W = g.ws + ((long)sp * g.batch1 * g.batch2 + z) * g.M * g.N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float* wrow = W + (long)row_of(a, v) * g.N;
        if (BKC) {
#pragma unroll
          for (int b = 0; b < 4; ++b) wrow[col_of(b)] = acc[a][b][v];
        } else {
          *reinterpret_cast<f32x4*>(wrow + col_of(0)) = f32x4{acc[a][0][v], acc[a][1][v], acc[a][2][v], acc[a][3][v]};
        }
      }
    return;
  }
  const Epi e = make_epi(g, z1, z2);
  const bool plain = !(g.add || g.bias || g.rowadd || g.rowscale || g.relu || g.accumulate || g.n_valid || g.C2) && g.alpha == 1.f;
  if (plain && !BKC && (g.ldc & 3) == 0 && ((((uintptr_t)e.C) & 15) == 0)) {  // bare 16-byte stores (weight / data gradients)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<f32x4*>(e.C + (long)row_of(a, v) * g.ldc + col_of(0)) =
            f32x4{acc[a][0][v], acc[a][1][v], acc[a][2][v], acc[a][3][v]};
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = row_of(a, v);
      if (BKC) {
#pragma unroll
        for (int b = 0; b < 4; ++b) epi_store(g, e, row, col_of(b), acc[a][b][v]);
      } else {  // four consecutive columns per lane
        epi_store4(g, e, row, col_of(0), make_float4(acc[a][0][v], acc[a][1][v], acc[a][2][v], acc[a][3][v]));
      }
    }
}

}  // namespace gc
