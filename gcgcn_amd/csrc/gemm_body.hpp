// Device-side tile code of the fp32 MFMA GEMM (shared by gemm.hip and chain.hip).  See gemm.hip for
// the design notes.
#pragma once
#include "gemm.hpp"
#include <type_traits>

namespace gc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
#ifndef GC_GEMM_EG
#define GC_GEMM_EG 4  // epilogue rows gathered at once by the plain kernels (A/B: 4 is the setting until round 3)
#endif
#ifndef GC_GEMM_PF
#define GC_GEMM_PF 2   // LDS read-ahead of the tile body in k-steps (A/B: 1 is the round-2 schedule; 2, 3, 5 measure alike: cfg 3 1.796 / 1.803-1.814 / 1.820 ms)
#endif

template <int BMN, bool KC, bool ALIGNED>
__device__ __forceinline__ void load_tile(float (&r)[BMN / 32][4], const float* __restrict__ src, long ld, int mn0, int k0,
                                          int MN, int Kend, int vec, int t) {
#pragma unroll
  for (int q = 0; q < BMN / 32; ++q) {
    const int f = t + 256 * q;
    int row, col, rlim, clim;  // row indexes the strided dim, col the contiguous one
    if (KC) {
      row = mn0 + (f >> 3);
      col = k0 + ((f & 7) << 2);
      rlim = MN;
      clim = Kend;
    } else {
      row = k0 + f / (BMN / 4);
      col = mn0 + ((f % (BMN / 4)) << 2);
      rlim = Kend;
      clim = MN;
    }
    const float* p = src + (long)row * ld + col;
    if (ALIGNED) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rlim) {
        if (vec && col + 3 < clim) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (col < clim) v.x = p[0];
          if (col + 1 < clim) v.y = p[1];
          if (col + 2 < clim) v.z = p[2];
          if (col + 3 < clim) v.w = p[3];
        }
      }
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  }
}

// (the prefetch sets are plain float arrays: hipcc keeps float4 arrays that are copied whole in scratch)
template <int BMN, bool KC>
__device__ __forceinline__ void store_tile(const float (&r)[BMN / 32][4], float* __restrict__ lds, int t) {
  constexpr int LD = KC ? BMN + 1 : BMN;
#pragma unroll
  for (int q = 0; q < BMN / 32; ++q) {
    const int f = t + 256 * q;
    if (KC) {
      const int m = f >> 3, k = (f & 7) << 2;
      lds[(k + 0) * LD + m] = r[q][0];
      lds[(k + 1) * LD + m] = r[q][1];
      lds[(k + 2) * LD + m] = r[q][2];
      lds[(k + 3) * LD + m] = r[q][3];
    } else {
      const int k = f / (BMN / 4), c = (f % (BMN / 4)) << 2;
      *reinterpret_cast<float4*>(&lds[k * LD + c]) = make_float4(r[q][0], r[q][1], r[q][2], r[q][3]);
    }
  }
}

// Interior tiles of plain operands: each thread's load addresses differ from k-tile to k-tile by a wave-uniform step only.
// Keeping the (BMN / 32) pointers in registers leaves one 64-bit add per load in the k-loop instead of a 64-bit
// multiply-add chain per load (the 64 x 64 kernel issued more address instructions than MFMAs; on random data it is
// clock-limited -- 10-28 % faster on all-zero operands -- and every instruction that is not an MFMA costs clock).
template <int BMN, bool KC>
struct TilePtr {
  const float* p[BMN / 32];
  long kstep;
  __device__ __forceinline__ void init(const float* __restrict__ src, long ld, int mn0, int t) {
#pragma unroll
    for (int q = 0; q < BMN / 32; ++q) {
      const int f = t + 256 * q;
      p[q] = KC ? src + (long)(mn0 + (f >> 3)) * ld + ((f & 7) << 2) : src + (long)(f / (BMN / 4)) * ld + mn0 + ((f % (BMN / 4)) << 2);
    }
    kstep = KC ? 1 : ld;
  }
  __device__ __forceinline__ void load(float (&r)[BMN / 32][4], int k0) const {
    const long o = (long)k0 * kstep;
#pragma unroll
    for (int q = 0; q < BMN / 32; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p[q] + o);
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  }
  // rb_mode 2, [k][.] operand of a 64-wide tile: this thread's piece q lies in rows 16 q .. 16 q + 15 of the k-tile, so a k-tile
  // made of two 16-row blocks ia, ib of the source (any two: the live blocks of a ragged batch, in list order) is two loads
  // at wave-uniform offsets.  A block id < 0 (past the end of the list) gives zeros.
  __device__ __forceinline__ void load_blocks(float (&r)[BMN / 32][4], const int ia, const int ib) const {
    static_assert(!KC && BMN == 64, "row-block k-tiles: [k][64]-wide operands");
    const float4 v0 = *reinterpret_cast<const float4*>(p[0] + (long)(max(ia, 0) * 16) * kstep);
    const float4 v1 = *reinterpret_cast<const float4*>(p[1] + (long)(max(ib, 0) * 16 - 16) * kstep);
    r[0][0] = v0.x, r[0][1] = v0.y, r[0][2] = v0.z, r[0][3] = v0.w;
    r[1][0] = v1.x, r[1][1] = v1.y, r[1][2] = v1.z, r[1][3] = v1.w;
    // (real branches, which the empty asm statements keep: if-converted into selects on the loaded registers, the zeroing made
    // every k-tile wait for its own request at once -- s_waitcnt vmcnt(1) straight behind the loads, edge_bwd_carry 64 -> 75 us)
    if (ia < 0) {
      asm volatile("");
      r[0][0] = r[0][1] = r[0][2] = r[0][3] = 0.f;
    }
    if (ib < 0) {
      asm volatile("");
      r[1][0] = r[1][1] = r[1][2] = r[1][3] = 0.f;
    }
  }
  // ---- ragged batches (GemmArgs::rb): the document rows that exist, through the list of live 16-row blocks -------------
  // rb_mode 1, KC operand [rows][k]: this thread's rows are fixed for the whole k-loop, so the list is read once, here.
  // Blocks of the tile past the live ones (only the last live tile has any) read a live block instead: their results are
  // never stored as computed.
  __device__ __forceinline__ void init_rows(const float* __restrict__ src, long ld, int mn0, int t, const int* __restrict__ rb, int nl) {
    static_assert(KC, "row-gathered operands are stored [rows][k]");
#pragma unroll
    for (int q = 0; q < BMN / 32; ++q) {
      const int f = t + 256 * q, vr = mn0 + (f >> 3);
      const int bid = rb[min(vr >> 4, nl - 1)];
      p[q] = src + ((long)bid * 16 + (vr & 15)) * ld + ((f & 7) << 2);
    }
    kstep = 1;
  }
};

// How a tile body obtains its operands.  PlainOperands reads g.A / g.B; other policies (head.hip: operands that are
// per-row outer products, never materialised) generate the same register tile instead.  Same k-loop for all of them.
struct PlainOperands {
  template <int BMN, bool KC, bool ALIGNED>
  __device__ __forceinline__ void load_a(float (&r)[BMN / 32][4], const GemmArgs& g, const float* __restrict__ A, int m0, int k0,
                                                int kend, int t) const {
    load_tile<BMN, KC, ALIGNED>(r, A, g.lda, m0, k0, g.M, kend, g.vecA, t);
  }
  template <int BMN, bool KC, bool ALIGNED>
  __device__ __forceinline__ void load_b(float (&r)[BMN / 32][4], const GemmArgs& g, const float* __restrict__ B, int n0, int k0,
                                                int kend, int t) const {
    load_tile<BMN, KC, ALIGNED>(r, B, g.ldb, n0, k0, g.N, kend, g.vecB, t);
  }
};

// Compile-time epilogue feature mask: a product that is known not to use a feature does not even compile
// its branch (the chain kernels inline ten tile bodies; with every option live in each, they are
// instruction-fetch bound).  EPI_ALL keeps every feature behind its run-time flag.
enum : int {
  EPI_ADD = 1, EPI_BIAS = 2, EPI_ROWADD = 4, EPI_ROWSCALE = 8, EPI_RELU = 16, EPI_ACCUM = 32, EPI_NVALID = 64,
  EPI_C2 = 128, EPI_ALPHA = 256, EPI_ALL = 511
};

// One output element through the fused epilogue (order documented in gemm.hpp).
struct Epi {
  float* C;
  const float* add;
  const float* rowadd;
  const float* rowscale;
  float* C2;
  const float* add2;
  long offC2;
  uint64_t key;
  bool dodrop;
  int z1;
};
__device__ __forceinline__ Epi make_epi(const GemmArgs& g, int z1, int z2) {
  Epi e;
  e.z1 = z1;
  e.C = g.C + z1 * g.sC1 + z2 * g.sC2;
  e.add = g.add ? g.add + z1 * g.sAdd1 + z2 * g.sAdd2 : nullptr;
  e.rowadd = g.rowadd ? g.rowadd + z1 * g.sRa1 + z2 * g.sRa2 : nullptr;
  e.rowscale = g.rowscale ? g.rowscale + z1 * g.sRs1 + z2 * g.sRs2 : nullptr;
  e.offC2 = z1 * g.sC21 + z2 * g.sC22;
  e.C2 = g.C2 ? g.C2 + e.offC2 : nullptr;
  e.add2 = g.add2 ? g.add2 + z1 * g.sAdd21 + z2 * g.sAdd22 : nullptr;
  e.dodrop = e.C2 && g.drop.snap;
  e.key = e.dodrop ? drop_key(g.drop) : 0;
  return e;
}
__device__ __forceinline__ void epi_store(const GemmArgs& g, const Epi& e, int row, int col, float acc) {
  float v = g.alpha * acc;
  if (e.add) v += e.add[(long)row * g.ldadd + col];
  if (g.bias) v += g.bias[col];
  if (e.rowadd) v += e.rowadd[row];
  if (e.rowscale) v *= e.rowscale[row];
  if (g.relu) v = fmaxf(v, 0.f);
  const long oc = (long)row * g.ldc + col;
  if (g.accumulate) v += e.C[oc];
  bool pad = false;
  if (g.n_valid) {
    const int doc = e.z1 * g.nv_zdoc + row / g.nv_rows;
    pad = (row % g.nv_rows) >= g.n_valid[doc];
  }
  if (pad) v = 0.f;
  e.C[oc] = v;
  if (e.C2) {
    const long o2 = (long)row * g.ldc2 + col;
    float w = v;
    if (e.dodrop)
      w = (rng_u32(e.key, (uint64_t)(g.drop_base + e.offC2 + o2)) >= g.drop.thresh) ? w * g.drop.scale : 0.f;
    if (e.add2) w += e.add2[(long)row * g.ldadd2 + col];
    if (pad) w = 0.f;
    e.C2[o2] = w;
  }
}

// Four consecutive columns of one row through the fused epilogue with 16-byte accesses: the same arithmetic in the same
// order as four epi_store calls (so either path gives the same bits), one round trip per operand instead of four.  The
// caller guarantees col % 4 == 0; operands that are not 16-byte addressable send the element back to the scalar path.
__device__ __forceinline__ void epi_store4(const GemmArgs& g, const Epi& e, int row, int col, const float4 a4) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  const bool vec = ((g.ldc | (e.add ? g.ldadd : 0) | (e.C2 ? g.ldc2 : 0) | (e.add2 ? g.ldadd2 : 0)) & 3) == 0 && al(e.C) &&
                   (!e.add || al(e.add)) && (!g.bias || al(g.bias)) && (!e.C2 || al(e.C2)) && (!e.add2 || al(e.add2));
  if (!vec) {
    epi_store(g, e, row, col, a4.x);
    epi_store(g, e, row, col + 1, a4.y);
    epi_store(g, e, row, col + 2, a4.z);
    epi_store(g, e, row, col + 3, a4.w);
    return;
  }
  const long oc = (long)row * g.ldc + col;
  // requests first
  float4 addv = make_float4(0.f, 0.f, 0.f, 0.f), biasv = addv, cv = addv, a2v = addv;
  if (e.add) addv = *reinterpret_cast<const float4*>(e.add + (long)row * g.ldadd + col);
  if (g.bias) biasv = *reinterpret_cast<const float4*>(g.bias + col);
  if (g.accumulate) cv = *reinterpret_cast<const float4*>(e.C + oc);
  const long o2 = (long)row * g.ldc2 + col;
  if (e.C2 && e.add2) a2v = *reinterpret_cast<const float4*>(e.add2 + (long)row * g.ldadd2 + col);
  const float ra = e.rowadd ? e.rowadd[row] : 0.f;
  const float rs = e.rowscale ? e.rowscale[row] : 1.f;
  bool pad = false;
  if (g.n_valid) {
    const int doc = e.z1 * g.nv_zdoc + row / g.nv_rows;
    pad = (row % g.nv_rows) >= g.n_valid[doc];
  }
  float v[4] = {g.alpha * a4.x, g.alpha * a4.y, g.alpha * a4.z, g.alpha * a4.w};
  const float ad[4] = {addv.x, addv.y, addv.z, addv.w}, bi[4] = {biasv.x, biasv.y, biasv.z, biasv.w};
  const float cc[4] = {cv.x, cv.y, cv.z, cv.w}, a2[4] = {a2v.x, a2v.y, a2v.z, a2v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (e.add) v[i] += ad[i];
    if (g.bias) v[i] += bi[i];
    if (e.rowadd) v[i] += ra;
    if (e.rowscale) v[i] *= rs;
    if (g.relu) v[i] = fmaxf(v[i], 0.f);
    if (g.accumulate) v[i] += cc[i];
    if (pad) v[i] = 0.f;
  }
  *reinterpret_cast<float4*>(e.C + oc) = make_float4(v[0], v[1], v[2], v[3]);
  if (e.C2) {
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = v[i];
      if (e.dodrop) w[i] = (rng_u32(e.key, (uint64_t)(g.drop_base + e.offC2 + o2 + i)) >= g.drop.thresh) ? w[i] * g.drop.scale : 0.f;
      if (e.add2) w[i] += a2[i];
      if (pad) w[i] = 0.f;
    }
    *reinterpret_cast<float4*>(e.C2 + o2) = make_float4(w[0], w[1], w[2], w[3]);
  }
}

// The 16 outputs a lane holds of one 32x32 accumulator tile (one column, 16 rows), EG rows at a
// time: the epilogue operands of a row group are gathered first (all loads in flight together), then
// the arithmetic, then the stores.  Element by element the loads are 16 dependent round trips.
// EG = 4 for the plain GEMM kernels (keeps them at ~75 VGPRs / 4-5 waves per SIMD, other waves hide the
// 4 round trips); EG = 16 for the chain kernels (one workgroup per CU: nothing else hides latency).
// Offsets are 32-bit on purpose: uniform base + 32-bit lane offset addressing keeps the gather to one
// VGPR per load instead of a 64-bit address pair (a [rows x ld] slice of this path is < 2^31 elements).
// MASK: features the caller guarantees are present (no run-time test, no branch: the operand gathers of a row
// group then really are one straight block of loads).  RT: features tested at run time.  Everything else is
// compiled out.
// rb0 / rb1: rows of the tile's first / second 16 rows in the output and in every row-indexed operand (dense: row0, row0 + 16;
// ragged batches: the two row blocks' places in the padded tensors).  dead: bit h set = half h lies past the live row blocks:
// it stores zeros (zero_dead) or nothing, and gathers nothing.
template <bool ALIGNED, int EG, int MASK, int RT>
__device__ __forceinline__ void epi_tile(const GemmArgs& g, const Epi& e, const int rb0, const int rb1, const int lh4, const int col,
                                         const f32x16& acc, const int dead = 0, const bool zero_dead = false) {
  if (!ALIGNED && col >= g.N) return;
  const bool hadd = (MASK & EPI_ADD) || ((RT & EPI_ADD) && e.add);
  const bool hra = (MASK & EPI_ROWADD) || ((RT & EPI_ROWADD) && e.rowadd);
  const bool hrs = (MASK & EPI_ROWSCALE) || ((RT & EPI_ROWSCALE) && e.rowscale);
  const bool hacc = (MASK & EPI_ACCUM) || ((RT & EPI_ACCUM) && g.accumulate);
  const bool hc2 = (MASK & EPI_C2) || ((RT & EPI_C2) && e.C2);
  const bool ha2 = (MASK & EPI_C2) || ((RT & EPI_C2) && e.C2 && e.add2);   // MASK: C2 comes with add2
  const bool hnv = (MASK & EPI_NVALID) || ((RT & EPI_NVALID) && g.n_valid);
  const bool hrelu = (MASK & EPI_RELU) || ((RT & EPI_RELU) && g.relu);
  const float bias = ((MASK & EPI_BIAS) || ((RT & EPI_BIAS) && g.bias)) ? g.bias[col] : 0.f;
  const float alpha = ((MASK | RT) & EPI_ALPHA) ? g.alpha : 1.f;
  const unsigned ldadd = (unsigned)g.ldadd, ldc = (unsigned)g.ldc, ldadd2 = (unsigned)g.ldadd2, ldc2 = (unsigned)g.ldc2;
#pragma unroll
  for (int q0 = 0; q0 < 16; q0 += EG) {
    float addv[EG], accv[EG], a2v[EG], rsv[EG], rav[EG];
    bool ok[EG], pad[EG];
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int r = q0 + u;
      const unsigned row = (unsigned)((r < 8 ? rb0 : rb1) + lh4 + (r & 3) + 8 * ((r >> 2) & 1));
      const bool dd = (dead >> (r >> 3)) & 1;
      ok[u] = (ALIGNED || (int)row < g.M) && (!dd || zero_dead);
      addv[u] = 0.f, accv[u] = 0.f, a2v[u] = 0.f, rsv[u] = 1.f, rav[u] = 0.f, pad[u] = dd;
      if (ok[u] && !dd) {
        if (hadd) addv[u] = e.add[row * ldadd + (unsigned)col];
        if (hra) rav[u] = e.rowadd[row];
        if (hrs) rsv[u] = e.rowscale[row];
        if (hacc) accv[u] = e.C[row * ldc + (unsigned)col];
        if (ha2) a2v[u] = e.add2[row * ldadd2 + (unsigned)col];
        if (hnv) pad[u] = ((int)row % g.nv_rows) >= g.n_valid[e.z1 * g.nv_zdoc + (int)row / g.nv_rows];
      }
    }
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      if (!ok[u]) continue;
      const int r = q0 + u;
      const unsigned row = (unsigned)((r < 8 ? rb0 : rb1) + lh4 + (r & 3) + 8 * ((r >> 2) & 1));
      float v = (alpha * acc[r] + addv[u] + bias + rav[u]) * rsv[u];
      if (hrelu) v = fmaxf(v, 0.f);
      v += accv[u];
      if (pad[u]) v = 0.f;
      e.C[row * ldc + (unsigned)col] = v;
      accv[u] = v;  // reused below as the value feeding the second output
    }
    if (hc2) {  // second output: dropout(v) + add2.  The dropout test is hoisted: one uniform branch per row group.
      if (e.dodrop) {
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          if (!ok[u]) continue;
          const int r = q0 + u;
          const unsigned o2 = (unsigned)((r < 8 ? rb0 : rb1) + lh4 + (r & 3) + 8 * ((r >> 2) & 1)) * ldc2 + (unsigned)col;
          float w = (rng_u32(e.key, (uint64_t)(g.drop_base + e.offC2 + (long)o2)) >= g.drop.thresh) ? accv[u] * g.drop.scale : 0.f;
          w += a2v[u];
          e.C2[o2] = pad[u] ? 0.f : w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          if (!ok[u]) continue;
          const int r = q0 + u;
          const unsigned o2 = (unsigned)((r < 8 ? rb0 : rb1) + lh4 + (r & 3) + 8 * ((r >> 2) & 1)) * ldc2 + (unsigned)col;
          e.C2[o2] = pad[u] ? 0.f : accv[u] + a2v[u];
        }
      }
    }
  }
}

// RB: the row-block paths of a ragged batch (GemmArgs::rb) are compiled in.  A separate instantiation, chosen by the launcher:
// with them in every body the DENSE launches lost 3-9 % (cfg 2 / cfg 1: more registers, a longer prologue) -- A/B of round 4.
// A split problem of a ragged batch whose M is the document-row dimension (rb_mode 1) is launched with the DENSE number of tile
// workgroups (the host never reads n_valid); the tile rows past the live blocks exit at once and leave their compute units idle --
// at DocRED's entity counts 14 of 32 tile rows are live, a launch of 512 workgroups runs 224, one per compute unit, each walking
// its k-tiles at a lone workgroup's latency-bound pace (cfg 2: 46 us for what two resident workgroups per unit do in half that).
// So the device cuts K finer instead: `w` times (a power of two, at most g.widen) as long as the live tiles times w still fit the
// launched tile rows and a slice keeps at least four whole k-tiles.  The same function of the live-block count gives the tile
// workgroups and the reduce pass their slice count: splits * w partial slabs, summed in slice order as always.
__device__ __forceinline__ int split_width(const GemmArgs& g, const int tml, const int tm) {
  int w = 1;
  if (g.splits > 1)
    while (2 * w <= g.widen && 2 * w * tml <= tm && (g.ksplit / (2 * w)) % BK == 0 && g.ksplit / (2 * w) >= 4 * BK) w *= 2;
  return w;
}

// wide: this launch's extra K cut (split_width; 1 everywhere but the split row-block problems of a ragged batch)
template <int TM, int TN, bool AKC, bool BKC, bool ALIGNED, int EG = GC_GEMM_EG, int MASK = 0, int RT = EPI_ALL, class OPS = PlainOperands,
          bool RB = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, float* __restrict__ lds, const int bx, const int by,
                                          const int zs, const int t = threadIdx.x, const bool do_store = true,
                                          const OPS& ops = OPS(), float* __restrict__ xchg = nullptr, const int role = 0,
                                          const int wide = 1) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int PF = GC_GEMM_PF;
  constexpr int LDA = AKC ? BM + 1 : BM;
  constexpr int LDB = BKC ? BN + 1 : BN;
  constexpr int SA = BK * LDA, SB = BK * LDB;  // floats per stage
  // the B image starts 16-byte aligned whatever LDA's parity
  constexpr int OFFB = (2 * SA + 3) & ~3;

  // t: this thread's index inside the 256-thread tile team (a chain workgroup runs two teams side by side);
  // do_store = false: the team only keeps the barriers company (its tile duplicates the other team's)
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int nsplit = RB ? g.splits * wide : g.splits, ksplit = RB ? g.ksplit / wide : g.ksplit;
  const int z = zs / nsplit, sp = zs - z * nsplit;
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const int m0 = by * BM, n0 = bx * BN;
  constexpr bool FASTP = ALIGNED && std::is_same<OPS, PlainOperands>::value;
  // ragged batches (GemmArgs::rb): the live 16-row blocks of the document-row dimension
  const bool rbm = RB && FASTP && AKC && g.rb && g.rb_mode == 1;             // M = document rows: A rows gathered, C rows scattered
  const bool rbk = RB && FASTP && !AKC && !BKC && g.rb && g.rb_mode == 2;    // K = document rows: the live 16-row blocks only
  const int nl = rbm ? __builtin_amdgcn_readfirstlane(*g.rb_n) : 0;
  const int nlk = rbk ? __builtin_amdgcn_readfirstlane(*g.rb_n) : 0;
  const int nkt = (nlk + 1) >> 1;   // k-tiles of two live blocks each (the last one may be half empty)
  int kbeg = sp * ksplit;
  int kend = min(g.K, kbeg + ksplit);
  // rb_mode 2: the ascending list of live 16-row blocks (at most ROWBLK_LIST_MAX = 512, gemm.hip prepare()) sits in four
  // registers, two 16-bit entries to a lane: k-tile e = list entries 2 e and 2 e + 1 in lane e % 64 of register e / 64.  The
  // k-loop picks the two blocks of its next k-tile with one v_readlane, no memory operation, and fetches each operand's two
  // halves at wave-uniform offsets (TilePtr::load_blocks).  (Round 4 first gathered blocks through scalar loads inside the
  // k-loop: twice the time per k-step of the dense loop; then walked live 32-row k-TILES -- tile kt = blocks 2 kt, 2 kt + 1 of
  // the padded tensor, live iff one of them is -- which keeps a document of 1-16 entities, or of 33-48, at a whole extra half
  // tile: 62 % of the dense k-steps at DocRED's entity counts where the blocks are 43 %.)
  int kl[4] = {-1, -1, -1, -1};
  if (rbk) {
    // Every slice runs the SAME number of k-tiles (two tile teams of one workgroup share its barriers): ceil(k-tiles / splits),
    // at least one; list positions past the end contribute zeros, so a slice that reaches past it -- or lies wholly behind it --
    // just adds zeros.
    const int ks = max((nkt + nsplit - 1) / nsplit, 1) * BK;
    kbeg = sp * ks, kend = kbeg + ks;
    // four unconditional 8-byte loads in flight together (entry nlk of an odd-length list exists: the dead blocks follow the
    // live ones and K / 16 is even), the conditions applied afterwards: a tile's prologue waits for ONE round trip
    int2 pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pr[j] = *reinterpret_cast<const int2*>(g.rb + 2 * min(lane + 64 * j, max(nkt - 1, 0)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      kl[j] = e < nkt ? (pr[j].x | ((2 * e + 1 < nlk ? pr[j].y : 0xffff) << 16)) : -1;
    }
  }

  const float* __restrict__ A = g.A + z1 * g.sA1 + z2 * g.sA2;
  const float* __restrict__ B = g.B + z1 * g.sB1 + z2 * g.sB2;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two register sets, each requested TWO k-tiles before its LDS store: tile kt+1 waits in one set while tile kt is
  // multiplied, tile kt+2 is in flight into the other.  A set is requested again right after its store (pinned there with
  // sched_barrier: left alone, hipcc sinks the requests to the end of the NEXT k-tile, just in front of the stores, and then
  // reuses their address registers so that the following k-tile starts with s_waitcnt vmcnt(0) -- every load's L2 round
  // trip exposed once per k-tile and wave, which is what held these kernels at 0.6-0.75 of the matrix pipe).  Tile
  // indices are clamped to the last tile so that all loads/stores are unconditional (a guarded prefetch makes hipcc keep
  // the float4 sets in scratch); the clamped extra loads re-read the last tile from L2 and are never consumed.
  float ra0[BM / 32][4], rb0[BN / 32][4], ra1[BM / 32][4], rb1[BN / 32][4];
  const int nk = (kend - kbeg + BK - 1) / BK;
  // rb_mode 1: a tile whose rows all lie past the live blocks does no arithmetic: it stores zeros where the output leaves
  // the block (rb_zero), nothing otherwise.  (Split problems: the reduce pass does the same per row, gemm.hip.)
  if (rbm && (m0 >> 4) >= nl && !xchg) {
    if (g.rb_zero && do_store && nsplit <= 1) {
      const Epi e = make_epi(g, z1, z2);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int bi = (m0 >> 4) + 2 * (wr * TM + i);
          epi_tile<ALIGNED, EG, MASK, RT>(g, e, g.rb[bi] * 16, g.rb[bi + 1] * 16, 4 * lh, n0 + (wc * TN + j) * 32 + l31, acc[i][j], 3, true);
        }
    }
    return;
  }
  auto kof = [&](int kt) { return kbeg + min(kt, nk - 1) * BK; };
  TilePtr<BM, AKC> tpa;
  TilePtr<BN, BKC> tpb;
  if constexpr (FASTP) {
    if constexpr (AKC) {
      if (rbm) tpa.init_rows(A, g.lda, m0, t, g.rb, max(nl, 1));
      else tpa.init(A, g.lda, m0, t);
      tpb.init(B, g.ldb, n0, t);
    } else {
      tpa.init(A, g.lda, m0, t), tpb.init(B, g.ldb, n0, t);
    }
  }
  // rb_mode 2: the two blocks of a register set's NEXT request (-1: past the list, that half of the registers is zeroed); wave-uniform
  struct BlkPair { int a, b; };
  BlkPair id0 = {0, 0}, id1 = {0, 0};
  auto ids = [&](BlkPair& id, int kt) {
    const int e = __builtin_amdgcn_readfirstlane(kbeg / BK + min(kt, nk - 1));
    const int v = e < 64 ? kl[0] : e < 128 ? kl[1] : e < 192 ? kl[2] : kl[3];
    const int pr = e < 4 * 64 ? __builtin_amdgcn_readlane(v, e & 63) : -1;
    id.a = pr == -1 ? -1 : (pr & 0xffff);                                   // (a live block id is < 0xffff: -1 is "no k-tile" only)
    id.b = ((unsigned)pr >> 16) == 0xffffu ? -1 : (int)((unsigned)pr >> 16);
  };
  auto load_a = [&](float (&r)[BM / 32][4], int k0, const BlkPair id) {
    if constexpr (FASTP) {
      if constexpr (RB && !AKC && !BKC && TM == 1) {
        if (rbk) {
          tpa.load_blocks(r, id.a, id.b);
          return;
        }
      }
      tpa.load(r, k0);
    } else {
      ops.template load_a<BM, AKC, ALIGNED>(r, g, A, m0, k0, kend, t);
    }
  };
  auto load_b = [&](float (&r)[BN / 32][4], int k0, const BlkPair id) {
    if constexpr (FASTP) {
      if constexpr (RB && !AKC && !BKC && TN == 1) {
        if (rbk) {
          tpb.load_blocks(r, id.a, id.b);
          return;
        }
      }
      tpb.load(r, k0);
    } else {
      ops.template load_b<BN, BKC, ALIGNED>(r, g, B, n0, k0, kend, t);
    }
  };
  if (rbk) ids(id0, 0), ids(id1, 1);
  load_a(ra0, kof(0), id0);
  load_b(rb0, kof(0), id0);
  load_a(ra1, kof(1), id1);
  load_b(rb1, kof(1), id1);
  if (rbk) ids(id0, 2), ids(id1, 3);
  __builtin_amdgcn_sched_barrier(0);
  store_tile<BM, AKC>(ra0, lds, t);
  store_tile<BN, BKC>(rb0, lds + OFFB, t);
  __builtin_amdgcn_sched_barrier(0);
  load_a(ra0, kof(2), id0);
  load_b(rb0, kof(2), id0);
  if (rbk) ids(id0, 4);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // One k-tile from LDS stage `cur`.  The operands of k-step kk+2 are read while the MFMAs of step kk
  // issue (register double buffer), so a wave does not serialise LDS latency with its matrix work.
  auto compute = [&](const int cur, auto&& mid) {
    const float* as = lds + cur * SA + wr * 32 * TM + l31 + lh * LDA;
    const float* bs = lds + OFFB + cur * SB + wc * 32 * TN + l31 + lh * LDB;
    // operand reads run PF k-steps (of 2) ahead of the MFMA that consumes them: one step is 64 cycles of matrix pipe per
    // accumulator, less than an LDS round trip when the wave has its SIMD to itself (carried tiles, chain products)
    float a[PF + 1][TM], b[PF + 1][TN];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[p][i] = as[(2 * p) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[p][j] = bs[(2 * p) * LDB + j * 32];
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int c = (kk >> 1) % (PF + 1), n = ((kk >> 1) + PF) % (PF + 1);
      if (kk + 2 * PF < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[n][i] = as[(kk + 2 * PF) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[n][j] = bs[(kk + 2 * PF) * LDB + j * 32];
      }
      // Pin the read-ahead in front of the MFMAs it overlaps with.  Left to itself the compiler folds the register double
      // buffer away and sinks each pair of operand reads next to its use (read, wait, two MFMAs, read, ...): other waves
      // fill those waits at four waves per SIMD, but tiles that ride in other kernels (edge_bwd_carry, the chain launches)
      // and launches of few workgroups run at two or fewer.  Measured: edge_bwd_carry 104 -> 98 us, cfg 3 -1..3 %, cfg 5
      // -1 %, classifier head -0.8 %, no launch slower.
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // half way through the k-tile: the staging work of the OTHER stage (LDS stores of the next tile, requests for the tile
      // after next) goes out in the shadow of this wave's dependent MFMA chain instead of between the last MFMA and the barrier
      if (kk == BK / 2 - 2) {
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto stage1 = [&](int kt) {                     // tile kt + 1 -> stage 1, then request tile kt + 3 into the freed set
    store_tile<BM, AKC>(ra1, lds + SA, t);
    store_tile<BN, BKC>(rb1, lds + OFFB + SB, t);
    __builtin_amdgcn_sched_barrier(0);
    load_a(ra1, kof(kt + 3), id1);
    load_b(rb1, kof(kt + 3), id1);
    if (rbk) ids(id1, kt + 5);
  };
  auto stage0 = [&](int kt) {                     // tile kt + 2 -> stage 0, then request tile kt + 4
    store_tile<BM, AKC>(ra0, lds, t);
    store_tile<BN, BKC>(rb0, lds + OFFB, t);
    __builtin_amdgcn_sched_barrier(0);
    load_a(ra0, kof(kt + 4), id0);
    load_b(rb0, kof(kt + 4), id0);
    if (rbk) ids(id0, kt + 6);
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {                  // whole pairs: no branch inside, the two register sets never meet in a phi
    compute(0, [&] { stage1(kt); });              // tile kt
    __syncthreads();
    compute(1, [&] { stage0(kt); });              // tile kt + 1
    __syncthreads();
  }
  if (kt < nk) {                                  // odd count: the last tile sits in stage 0
    compute(0, [] {});
    __syncthreads();                              // (a chain kernel's next product restages these images at once)
  }

  // ---- two tile teams of one workgroup split K (role 1 gives, role 2 takes): the partial sums meet in LDS --------
  if (xchg) {
    if (role == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xchg[r * 256 + t] = acc[0][0][r];
      __syncthreads();
      return;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += xchg[r * 256 + t];
  }
  // ---- store ------------------------------------------------------------------------------
  if (!do_store) return;
  if (nsplit > 1 && !xchg) {  // raw partial sums -> workspace [split][batch][M][N]
    float* __restrict__ W = g.ws + ((long)sp * g.batch1 * g.batch2 + z) * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (!ALIGNED && row >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + (wc * TN + j) * 32 + l31;
          if (!ALIGNED && col >= g.N) continue;
          W[(long)row * g.N + col] = acc[i][j][r];
        }
      }
    return;
  }
  const Epi e = make_epi(g, z1, z2);
  // rows of this wave's sub-tile(s) in the output: dense = the tile's own rows; rb_mode 1 = the two row blocks' places in
  // the padded tensors (blocks past the live ones: zero-stored or skipped, see epi_tile)
  auto rows_of = [&](const int i, int& r0, int& r1, int& dead) {
    const int ms = m0 + (wr * TM + i) * 32;
    r0 = ms, r1 = ms + 16, dead = 0;
    if (rbm) {
      const int bi = ms >> 4;
      r0 = g.rb[bi] * 16, r1 = g.rb[bi + 1] * 16;
      dead = (bi >= nl ? 1 : 0) | (bi + 1 >= nl ? 2 : 0);
    }
  };
  if (RT == EPI_ALL && !(g.add || g.bias || g.rowadd || g.rowscale || g.relu || g.accumulate || g.n_valid || g.C2) &&
      g.alpha == 1.f) {
    // the common case of the stand-alone kernels (weight / data gradients): a bare store
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int r0, r1, dead;
      rows_of(i, r0, r1, dead);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        epi_tile<ALIGNED, EG, 0, 0>(g, e, r0, r1, 4 * lh, n0 + (wc * TN + j) * 32 + l31, acc[i][j], dead, g.rb_zero != 0);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int r0, r1, dead;
    rows_of(i, r0, r1, dead);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      epi_tile<ALIGNED, EG, MASK, RT>(g, e, r0, r1, 4 * lh, n0 + (wc * TN + j) * 32 + l31, acc[i][j], dead, g.rb_zero != 0);
  }
}

// Stage 2 of a riding column sum whose partials are in memory (ColRide::ready_slices > 0): out[c] = their sum, slices in
// order, 16 independent loads in flight; workgroup idx of 256 threads serves columns [256 idx, 256 idx + 256).
__device__ __forceinline__ void col_ride_stage2_block(const ColRide& cr, const int idx) {
  const int c = idx * 256 + (int)(threadIdx.x & 255);
  if (c < cr.C && threadIdx.x < 256) {
    float s = 0.f;
    for (int q0 = 0; q0 < cr.ready_slices; q0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = cr.part[(long)min(q0 + u, cr.ready_slices - 1) * cr.C + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (q0 + u < cr.ready_slices) ? v[u] : 0.f;
    }
    cr.out[c] = s;
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share an XCD and its 4 MiB L2).
// Give each XCD one CONTIGUOUS run of the row-major tile list instead of every 8th tile, so that the
// tiles resident on an XCD share A row panels / B column panels in its L2 (speed only; any placement
// is correct).  Bijective for every grid size.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Workgroup hb of a multi-problem launch (GemmGroup): find its problem by the prefix sums of (8-aligned) tile ranges,
// remap inside the problem, run one interior 64x64 tile.  Shared by gemm_group_kernel and by kernels that carry
// deferred problems as passengers (edge.hip).
// A problem's tile (bx, by, zs) for position q of its tile list (tn x tm tiles, zs slices; q < tn tm zs): the XCD-contiguous
// order, the short side walked fastest -- the tiles that are neighbours in the list (and therefore on one XCD) then share BOTH
// operand panels (weight gradients are [D x H*D] with K = B*N: 4 x 32 tiles, each reading two 512 KiB panels).
__device__ __forceinline__ void tile_of(int q, const int count, const int tn, const int tm, int& bx, int& by, int& zs) {
  q = xcd_remap(q, count);
  zs = q / (tn * tm);
  const int r = q - zs * (tn * tm);
  bx = (tm < tn) ? r / tm : r % tn, by = (tm < tn) ? r % tm : r / tn;
}
// rb_mode 1 (ragged batch, M = document rows): only the first ceil(live blocks / 4) tile rows hold entities.  The list is
// re-cut on the device: positions [0, live tiles) are the live tiles in the usual XCD-balanced order (leaving them where
// they are would hand the XCDs that own the front of the list all the work), the positions behind them are the dead tile
// rows -- zero-stored where the output leaves the block, left at once otherwise.  False: nothing to do for this position.
__device__ __forceinline__ bool tile_of_rows(const GemmArgs& g, const int q, const int tn, const int tm, int& bx, int& by, int& zs,
                                             int& wide) {
  const int nl = __builtin_amdgcn_readfirstlane(*g.rb_n);
  const int tml = min((nl + 3) >> 2, tm);
  wide = split_width(g, tml, tm);
  const int live = tml * tn * g.splits * wide;
  if (q < live) {
    tile_of(q, live, tn, tml, bx, by, zs);
    return true;
  }
  if (!g.rb_zero || g.splits > 1) return false;     // (split problems: the reduce pass zero-stores the dead rows)
  const int d = q - live;
  by = tml + d / tn, bx = d - (d / tn) * tn, zs = 0;
  return by < tm;
}

template <bool RB = false, class G>
__device__ __forceinline__ void gemm_group_block(const G& gg, int hb, float* __restrict__ lds) {
  int i = 0;
  while (i + 1 < gg.nprob && hb >= gg.tile_begin[i + 1]) ++i;
  int b = hb - gg.tile_begin[i];
  if (b >= gg.tile_take[i]) return;
  const GemmArgs& g = gg.p[i];
  const int tn = g.N >> 6, tm = g.M >> 6;
  int bx, by, zs, wide = 1;
  if (RB && g.rb && g.rb_mode == 1) {
    if (!tile_of_rows(g, b + gg.tile_first[i], tn, tm, bx, by, zs, wide)) return;
  } else {
    tile_of(b + gg.tile_first[i], gg.tile_count[i], tn, tm, bx, by, zs);
  }
  const int t = threadIdx.x;
  const PlainOperands po;
  if (g.a_kc) {
    if (g.b_kc) gemm_body<1, 1, true, true, true, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, bx, by, zs, t, true, po, nullptr, 0, wide);
    else gemm_body<1, 1, true, false, true, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, bx, by, zs, t, true, po, nullptr, 0, wide);
  } else {
    if (g.b_kc) gemm_body<1, 1, false, true, true, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, bx, by, zs, t, true, po, nullptr, 0, wide);
    else gemm_body<1, 1, false, false, true, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, bx, by, zs, t, true, po, nullptr, 0, wide);
  }
}

// The same for a 512-thread workgroup of two tile teams (chain kernels): workgroup hb runs ONE unsplit-output tile of its
// problem, team 0 the first half of K and team 1 the second (K % 64 == 0: equal halves, same number of barriers); team 1
// hands its partial sums to team 0 through xchg (64 * 64 floats of LDS).  Two waves per SIMD on one tile: the compute
// unit's matrix pipe finishes a tile in the time one team would need for it alone, with the second wave hiding latency.
template <class G>
__device__ __forceinline__ void gemm_group_splitk_block(const G& gg, int hb, float* __restrict__ lds, int team_lds,
                                                        float* __restrict__ xchg) {
  int i = 0;
  while (i + 1 < gg.nprob && hb >= gg.tile_begin[i + 1]) ++i;
  int q = hb - gg.tile_begin[i];
  if (q >= gg.tile_take[i]) return;  // padding workgroup (uniform over the workgroup)
  q = xcd_remap(q + gg.tile_first[i], gg.tile_count[i]);
  const int team = threadIdx.x >> 8, t = threadIdx.x & 255;
  GemmArgs g = gg.p[i];
  const int tn = g.N >> 6, tm = g.M >> 6;
  const int z = q / (tn * tm), r = q - z * (tn * tm);
  const int bx = (tm < tn) ? r / tm : r % tn, by = (tm < tn) ? r % tm : r / tn;   // same tile list as gemm_group_block
  g.splits = 2, g.ksplit = g.K >> 1;
  float* tl = lds + team * team_lds;
  const int zs = z * 2 + team, role = team ? 1 : 2;
  if (g.a_kc) {
    if (g.b_kc) gemm_body<1, 1, true, true, true>(g, tl, bx, by, zs, t, true, PlainOperands(), xchg, role);
    else gemm_body<1, 1, true, false, true>(g, tl, bx, by, zs, t, true, PlainOperands(), xchg, role);
  } else {
    if (g.b_kc) gemm_body<1, 1, false, true, true>(g, tl, bx, by, zs, t, true, PlainOperands(), xchg, role);
    else gemm_body<1, 1, false, false, true>(g, tl, bx, by, zs, t, true, PlainOperands(), xchg, role);
  }
}

template <int TM, int TN, bool AKC, bool BKC>
constexpr int lds_floats() {
  return (((2 * BK * (AKC ? 64 * TM + 1 : 64 * TM)) + 3) & ~3) + 2 * BK * (BKC ? 64 * TN + 1 : 64 * TN);
}

}  // namespace gc
