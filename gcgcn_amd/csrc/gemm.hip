// Batched exact-fp32 GEMM for gfx950: v_mfma_f32_32x32x2_f32 tiles, k-major LDS images,
// register-prefetch double buffering, split-K with a deterministic reduce, fused epilogue (gemm.hpp).
//
// Block = 256 threads = 4 waves in a 2x2 arrangement; a wave owns (32*TM) x (32*TN) outputs,
// i.e. TM*TN accumulator tiles of 32x32 (16 VGPRs each).  BK = 32.
//
// LDS images are k-major: As[k][m], Bs[k][n].  The MFMA operand of lane l is
//   a = A[i = l & 31][k = l >> 5],  b = B[k = l >> 5][j = l & 31]
// so each half-wave reads 32 consecutive floats of one k-row: conflict-free ds_read_b32.
// A k-contiguous source (A as [M][K], B as [N][K]) is read 128 B per row (8 lanes x 16 B) and
// transposed on the LDS write; its row stride is BMN + 1 so that the 4-byte stores of the
// 4 rows x 8 k-quads handled by a 32-lane group fall on 32 different banks.
//
// The problems on this path are small (M = B*N ~ 2048, N, K in {64..2048}); an f32 MFMA tile
// takes 64 cycles, so a 64x64x32 step is only ~0.4 us of matrix work against ~1 us of load
// latency.  Throughput therefore comes from residency, not from big tiles: 64x64 blocks at
// 3-5 blocks per CU, and split-K (partials to a workspace + one reduce/epilogue kernel,
// bitwise reproducible) when M*N alone gives fewer than ~3 blocks per CU.
//
// Codegen note (ROCm 7.2): per-lane guarded float4 loads get if-converted into predicated
// scalar loads with a vmcnt(0) at the loop head.  The interior path (ALIGNED) is therefore a
// separate instantiation with unconditional 16-byte loads; ragged shapes take the guarded one.
#include "gemm.hpp"

namespace gc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

template <int BMN, bool KC, bool ALIGNED>
__device__ __forceinline__ void load_tile(float4 (&r)[BMN / 32], const float* __restrict__ src, long ld, int mn0, int k0,
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
      r[q] = *reinterpret_cast<const float4*>(p);
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
      r[q] = v;
    }
  }
}

template <int BMN, bool KC>
__device__ __forceinline__ void store_tile(const float4 (&r)[BMN / 32], float* __restrict__ lds, int t) {
  constexpr int LD = KC ? BMN + 1 : BMN;
#pragma unroll
  for (int q = 0; q < BMN / 32; ++q) {
    const int f = t + 256 * q;
    if (KC) {
      const int m = f >> 3, k = (f & 7) << 2;
      lds[(k + 0) * LD + m] = r[q].x;
      lds[(k + 1) * LD + m] = r[q].y;
      lds[(k + 2) * LD + m] = r[q].z;
      lds[(k + 3) * LD + m] = r[q].w;
    } else {
      const int k = f / (BMN / 4), c = (f % (BMN / 4)) << 2;
      *reinterpret_cast<float4*>(&lds[k * LD + c]) = r[q];
    }
  }
}

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

template <int TM, int TN, bool AKC, bool BKC, bool ALIGNED>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, float* __restrict__ lds, const int bx, const int by,
                                          const int zs) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA = AKC ? BM + 1 : BM;
  constexpr int LDB = BKC ? BN + 1 : BN;
  constexpr int SA = BK * LDA, SB = BK * LDB;  // floats per stage
  // the B image starts 16-byte aligned whatever LDA's parity
  constexpr int OFFB = (2 * SA + 3) & ~3;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int z = zs / g.splits, sp = zs - z * g.splits;
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = sp * g.ksplit;
  const int kend = min(g.K, kbeg + g.ksplit);

  const float* __restrict__ A = g.A + z1 * g.sA1 + z2 * g.sA2;
  const float* __restrict__ B = g.B + z1 * g.sB1 + z2 * g.sB2;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[BM / 32], rb[BN / 32];
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    load_tile<BM, AKC, ALIGNED>(ra, A, g.lda, m0, kbeg, g.M, kend, g.vecA, t);
    load_tile<BN, BKC, ALIGNED>(rb, B, g.ldb, n0, kbeg, g.N, kend, g.vecB, t);
    store_tile<BM, AKC>(ra, lds, t);
    store_tile<BN, BKC>(rb, lds + OFFB, t);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<BM, AKC, ALIGNED>(ra, A, g.lda, m0, kbeg + (kt + 1) * BK, g.M, kend, g.vecA, t);
      load_tile<BN, BKC, ALIGNED>(rb, B, g.ldb, n0, kbeg + (kt + 1) * BK, g.N, kend, g.vecB, t);
    }
    const float* as = lds + cur * SA + wr * 32 * TM + l31;
    const float* bs = lds + OFFB + cur * SB + wc * 32 * TN + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(kk + lh) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[(kk + lh) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<BM, AKC>(ra, lds + (cur ^ 1) * SA, t);
      store_tile<BN, BKC>(rb, lds + OFFB + (cur ^ 1) * SB, t);
    }
    __syncthreads();
  }

  // ---- store ------------------------------------------------------------------------------
  if (g.splits > 1) {  // raw partial sums -> workspace [split][batch][M][N]
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
        epi_store(g, e, row, col, acc[i][j][r]);
      }
    }
}

template <int TM, int TN, bool AKC, bool BKC>
constexpr int lds_floats() {
  return (((2 * BK * (AKC ? 64 * TM + 1 : 64 * TM)) + 3) & ~3) + 2 * BK * (BKC ? 64 * TN + 1 : 64 * TN);
}

template <int TM, int TN, bool AKC, bool BKC, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<TM, TN, AKC, BKC>()];
  gemm_body<TM, TN, AKC, BKC, ALIGNED>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent problems in ONE launch (64x64 tiles, interior shapes only): block -> problem by
// prefix sums of tile counts, layout chosen per problem by a block-uniform branch.
__global__ __launch_bounds__(256) void gemm_group_kernel(const GemmGroup gg) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  int b = blockIdx.x, i = 0;
  while (i + 1 < gg.nprob && b >= gg.tile_begin[i + 1]) ++i;
  b -= gg.tile_begin[i];
  const GemmArgs& g = gg.p[i];
  const int tn = g.N >> 6, tm = g.M >> 6;
  const int bx = b % tn, by = (b / tn) % tm, zs = b / (tn * tm);
  if (g.a_kc) {
    if (g.b_kc) gemm_body<1, 1, true, true, true>(g, lds, bx, by, zs);
    else gemm_body<1, 1, true, false, true>(g, lds, bx, by, zs);
  } else {
    if (g.b_kc) gemm_body<1, 1, false, true, true>(g, lds, bx, by, zs);
    else gemm_body<1, 1, false, false, true>(g, lds, bx, by, zs);
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const GemmGroup gg) {
  int b = blockIdx.x, i = 0;
  while (i + 1 < gg.nprob && b >= gg.red_begin[i + 1]) ++i;
  b -= gg.red_begin[i];
  const GemmArgs& g = gg.p[i];
  if (g.splits <= 1) return;
  const long mn = (long)g.M * g.N;
  const int per = (int)((mn + 255) / 256);
  const int z = b / per;
  const long idx = (long)(b - z * per) * 256 + threadIdx.x;
  if (idx >= mn) return;
  const long nb = (long)g.batch1 * g.batch2;
  const float* w = g.ws + (long)z * mn + idx;
  float acc = 0.f;
  for (int s = 0; s < g.splits; ++s) acc += w[(long)s * nb * mn];
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const Epi e = make_epi(g, z1, z2);
  const int row = (int)(idx / g.N), col = (int)(idx - (long)row * g.N);
  epi_store(g, e, row, col, acc);
}

// Sum the split-K partials in split order (bitwise reproducible) and run the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
  const long mn = (long)g.M * g.N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int z = blockIdx.y;
  if (idx >= mn) return;
  const long nb = (long)g.batch1 * g.batch2;
  const float* w = g.ws + (long)z * mn + idx;
  float acc = 0.f;
  for (int s = 0; s < g.splits; ++s) acc += w[(long)s * nb * mn];
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const Epi e = make_epi(g, z1, z2);
  const int row = (int)(idx / g.N), col = (int)(idx - (long)row * g.N);
  epi_store(g, e, row, col, acc);
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int TM, int TN, bool ALIGNED>
static int launch(const GemmArgs& g, hipStream_t stream) {
  dim3 grid(cdiv(g.N, 64 * TN), cdiv(g.M, 64 * TM), g.batch1 * g.batch2 * g.splits), block(256);
  {
    ProfScope ps(g.tag, stream);
    if (g.a_kc && !g.b_kc)
      hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false, ALIGNED>), grid, block, 0, stream, g);
    else if (g.a_kc && g.b_kc)
      hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true, ALIGNED>), grid, block, 0, stream, g);
    else if (!g.a_kc && !g.b_kc)
      hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false, ALIGNED>), grid, block, 0, stream, g);
    else
      hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true, ALIGNED>), grid, block, 0, stream, g);
  }
  if (int e = check_launch("gemm")) return e;
  if (g.splits > 1) {
    ProfScope ps("gemm_splitk_reduce", stream);
    dim3 rgrid(cdiv((long)g.M * g.N, 256), g.batch1 * g.batch2);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, g);
    return check_launch("gemm_splitk_reduce");
  }
  return 0;
}

// Fill the launcher-owned fields (vec flags, tile, split factor).  Returns the chosen tile (1 or 2) or -1.
static int prepare(GemmArgs& g, int tile, int splits, long resident_tiles_hint) {
  if (!(g.A && g.B && g.C)) { set_error("gemm: null operand"); return -1; }
  if (!(g.M >= 0 && g.N >= 0 && g.K >= 0 && g.batch1 >= 1 && g.batch2 >= 1)) { set_error("gemm: bad shape"); return -1; }
  const long nb = (long)g.batch1 * g.batch2;
  g.vecA = aligned16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
  g.vecB = aligned16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
  const long t64 = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * nb;
  if (tile == 0) tile = (t64 >= 4096) ? 2 : 1;  // 128x128 only when it still leaves >= 4 blocks per CU
  const long tiles = ((tile == 2) ? (long)cdiv(g.M, 128) * cdiv(g.N, 128) * nb : t64) + resident_tiles_hint;
  if (splits == 0) {
    // split K until ~3 blocks per CU are resident, keeping >= 2 k-steps of 32 per split
    splits = 1;
    while (tiles * splits < 768 && splits < 16 && g.K % (splits * 2 * BK) == 0 && g.K / (splits * 2) >= 2 * BK) splits *= 2;
  }
  if (splits > 1 && (!g.ws || (long)splits * nb * g.M * g.N > g.ws_elems || g.K % (splits * BK) != 0)) splits = 1;
  g.splits = splits;
  g.ksplit = (splits > 1) ? g.K / splits : g.K;
  return tile;
}

int gemm(const GemmArgs& g_in, hipStream_t stream, int tile, int splits) {
  GemmArgs g = g_in;
  if (g.M == 0 || g.N == 0) return 0;
  tile = prepare(g, tile, splits, 0);
  if (tile < 0) return 1;
  const long nb = (long)g.batch1 * g.batch2;
  GC_REQUIRE(nb * g.splits <= 65535, "gemm: batch %ld x splits %d exceeds grid.z", nb, g.splits);
  GC_REQUIRE(cdiv(g.M, 64) <= 65535, "gemm: M %d exceeds grid.y", g.M);
  const int bm = (tile == 2) ? 128 : 64;
  const bool al = g.vecA && g.vecB && g.M % bm == 0 && g.N % bm == 0 && g.ksplit % BK == 0;
  if (tile == 2) return al ? launch<2, 2, true>(g, stream) : launch<2, 2, false>(g, stream);
  return al ? launch<1, 1, true>(g, stream) : launch<1, 1, false>(g, stream);
}

// Independent problems in one launch (plus at most one reduce launch).  Problems that are not interior
// 64x64 shapes fall back to their own launches.
int gemm_group(const GemmArgs* probs, int n, hipStream_t stream) {
  GemmGroup gg;
  gg.nprob = 0;
  long total = 0;
  for (int i = 0; i < n; ++i) total += (long)cdiv(probs[i].M, 64) * cdiv(probs[i].N, 64) * probs[i].batch1 * probs[i].batch2;
  int tiles = 0, reds = 0;
  bool any_split = false;
  long ws_used = 0;
  for (int i = 0; i < n; ++i) {
    GemmArgs g = probs[i];
    if (g.M == 0 || g.N == 0) continue;
    const long own = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * g.batch1 * g.batch2;
    // the group shares one workspace: give each problem its own slice
    float* ws0 = g.ws;
    const long wse0 = g.ws_elems;
    if (ws0) g.ws = ws0 + ws_used, g.ws_elems = wse0 - ws_used;
    if (prepare(g, 1, 0, total - own) < 0) return 1;
    const bool al = g.vecA && g.vecB && g.M % 64 == 0 && g.N % 64 == 0 && g.ksplit % BK == 0;
    if (!al || gg.nprob == GemmGroup::MAXP) {
      g.ws = ws0, g.ws_elems = wse0;  // runs before the group launch; the group's slices are written later
      if (int e = gemm(g, stream, 0, 0)) return e;
      continue;
    }
    const long nb = (long)g.batch1 * g.batch2;
    if (g.splits > 1) ws_used += (long)g.splits * nb * g.M * g.N, any_split = true;
    gg.tile_begin[gg.nprob] = tiles;
    gg.red_begin[gg.nprob] = reds;
    tiles += (int)(own * g.splits);
    if (g.splits > 1) reds += (int)(nb * cdiv((long)g.M * g.N, 256));
    gg.p[gg.nprob++] = g;
  }
  if (gg.nprob == 0) return 0;
  gg.tile_begin[gg.nprob] = tiles;
  gg.red_begin[gg.nprob] = reds;
  {
    ProfScope ps("gemm_group", stream);
    hipLaunchKernelGGL(gemm_group_kernel, dim3(tiles), dim3(256), 0, stream, gg);
  }
  if (int e = check_launch("gemm_group")) return e;
  if (any_split) {
    ProfScope ps("gemm_splitk_reduce", stream);
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(reds), dim3(256), 0, stream, gg);
    return check_launch("gemm_group_reduce");
  }
  return 0;
}

}  // namespace gc
