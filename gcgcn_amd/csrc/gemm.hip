// Batched exact-fp32 GEMM for gfx950: v_mfma_f32_32x32x2_f32 tiles, k-major LDS images,
// register-prefetch double buffering, fused epilogue (gemm.hpp).
//
// Block = 256 threads = 4 waves in a 2x2 arrangement; a wave owns (32*TM) x (32*TN) outputs,
// i.e. TM*TN accumulator tiles of 32x32 (16 VGPRs each).  BK = 16.
//
// LDS images are k-major: As[k][m], Bs[k][n].  The MFMA operand of lane l is
//   a = A[i = l & 31][k = l >> 5],  b = B[k = l >> 5][j = l & 31]
// so each half-wave reads 32 consecutive floats of one k-row: conflict free ds_read_b32.
// A k-contiguous source (A as [M][K], B as [N][K]) is transposed on the LDS write; its row
// stride is BMN + 2 so that the four 4-byte stores of the 8 rows x 4 k-quads handled by a
// 32-lane group fall on 32 different banks.  f32 MFMA runs at the f32 vector rate (1/16 of
// bf16), so operand traffic is far from binding: one ds_read_b32 per operand per 64-cycle MFMA.
#include "gemm.hpp"

namespace gc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

// Load the [BMN x BK] tile of an operand into registers.
//   KC = true : memory is [mn][k], k contiguous     -> float4 along k, NV = BMN/64 per thread
//   KC = false: memory is [k][mn], mn contiguous    -> float4 along mn
template <int BMN, bool KC>
__device__ __forceinline__ void load_tile(float4 (&r)[BMN / 64], const float* __restrict__ src, long ld,
                                          int mn0, int k0, int MN, int K, int vec, int t) {
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int f = t + 256 * q;
    int row, col, rlim, clim;  // row indexes the strided dim, col the contiguous one
    if (KC) {
      row = mn0 + (f >> 2);
      col = k0 + ((f & 3) << 2);
      rlim = MN;
      clim = K;
    } else {
      row = k0 + f / (BMN / 4);
      col = mn0 + ((f % (BMN / 4)) << 2);
      rlim = K;
      clim = MN;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rlim) {
      const float* p = src + (long)row * ld + col;
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

template <int BMN, bool KC>
__device__ __forceinline__ void store_tile(const float4 (&r)[BMN / 64], float* __restrict__ lds, int t) {
  constexpr int LD = KC ? BMN + 2 : BMN;
#pragma unroll
  for (int q = 0; q < BMN / 64; ++q) {
    const int f = t + 256 * q;
    if (KC) {
      const int m = f >> 2, k = (f & 3) << 2;
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

template <int TM, int TN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA = AKC ? BM + 2 : BM;
  constexpr int LDB = BKC ? BN + 2 : BN;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (LDA + LDB)];
  auto As = [&](int b) -> float* { return lds + b * (BK * LDA); };
  auto Bs = [&](int b) -> float* { return lds + 2 * BK * LDA + b * (BK * LDB); };

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int z = blockIdx.z;
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const float* __restrict__ A = g.A + z1 * g.sA1 + z2 * g.sA2;
  const float* __restrict__ B = g.B + z1 * g.sB1 + z2 * g.sB2;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[TM], rb[TN];
  const int nk = (g.K + BK - 1) / BK;
  load_tile<BM, AKC>(ra, A, g.lda, m0, 0, g.M, g.K, g.vecA, t);
  load_tile<BN, BKC>(rb, B, g.ldb, n0, 0, g.N, g.K, g.vecB, t);
  store_tile<BM, AKC>(ra, As(0), t);
  store_tile<BN, BKC>(rb, Bs(0), t);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<BM, AKC>(ra, A, g.lda, m0, (kt + 1) * BK, g.M, g.K, g.vecA, t);
      load_tile<BN, BKC>(rb, B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, g.vecB, t);
    }
    const float* as = As(cur) + wr * 32 * TM + l31;
    const float* bs = Bs(cur) + wc * 32 * TN + l31;
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
      store_tile<BM, AKC>(ra, As(cur ^ 1), t);
      store_tile<BN, BKC>(rb, Bs(cur ^ 1), t);
    }
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------
  float* __restrict__ C = g.C + z1 * g.sC1 + z2 * g.sC2;
  const float* add = g.add ? g.add + z1 * g.sAdd1 + z2 * g.sAdd2 : nullptr;
  const float* rowadd = g.rowadd ? g.rowadd + z1 * g.sRa1 + z2 * g.sRa2 : nullptr;
  const float* rowscale = g.rowscale ? g.rowscale + z1 * g.sRs1 + z2 * g.sRs2 : nullptr;
  const long offC2 = z1 * g.sC21 + z2 * g.sC22;
  float* C2 = g.C2 ? g.C2 + offC2 : nullptr;
  const float* add2 = g.add2 ? g.add2 + z1 * g.sAdd21 + z2 * g.sAdd22 : nullptr;
  const bool dodrop = C2 && g.drop.snap;
  const uint64_t key = dodrop ? drop_key(g.drop) : 0;

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= g.M) continue;
      bool pad = false;
      if (g.n_valid) {
        const int doc = z1 * g.nv_zdoc + row / g.nv_rows;
        pad = (row % g.nv_rows) >= g.n_valid[doc];
      }
      const float ra_ = rowadd ? rowadd[row] : 0.f;
      const float rs_ = rowscale ? rowscale[row] : 1.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wc * TN + j) * 32 + l31;
        if (col >= g.N) continue;
        float v = g.alpha * acc[i][j][r];
        if (add) v += add[(long)row * g.ldadd + col];
        if (g.bias) v += g.bias[col];
        v += ra_;
        v *= rs_;
        if (g.relu) v = fmaxf(v, 0.f);
        const long oc = (long)row * g.ldc + col;
        if (g.accumulate) v += C[oc];
        if (pad) v = 0.f;
        C[oc] = v;
        if (C2) {
          const long o2 = (long)row * g.ldc2 + col;
          float w = v;
          if (dodrop) w = (rng_u32(key, (uint64_t)(g.drop_base + offC2 + o2)) >= g.drop.thresh) ? w * g.drop.scale : 0.f;
          if (add2) w += add2[(long)row * g.ldadd2 + col];
          if (pad) w = 0.f;
          C2[o2] = w;
        }
      }
    }
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int TM, int TN>
static int launch(const GemmArgs& g, hipStream_t stream) {
  dim3 grid(cdiv(g.N, 64 * TN), cdiv(g.M, 64 * TM), g.batch1 * g.batch2), block(256);
  ProfScope ps(g.tag, stream);
  if (g.a_kc && !g.b_kc)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false>), grid, block, 0, stream, g);
  else if (g.a_kc && g.b_kc)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true>), grid, block, 0, stream, g);
  else if (!g.a_kc && !g.b_kc)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true>), grid, block, 0, stream, g);
  return check_launch("gemm");
}

int gemm(const GemmArgs& g_in, hipStream_t stream, int tile) {
  GemmArgs g = g_in;
  GC_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
  GC_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0 && g.batch1 >= 1 && g.batch2 >= 1, "gemm: bad shape");
  if (g.M == 0 || g.N == 0) return 0;
  GC_REQUIRE((long)g.batch1 * g.batch2 <= 65535, "gemm: batch %ld exceeds grid.z", (long)g.batch1 * g.batch2);
  GC_REQUIRE(cdiv(g.M, 64) <= 65535, "gemm: M %d exceeds grid.y", g.M);
  g.vecA = aligned16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
  g.vecB = aligned16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
  if (tile == 0) {
    // 128x128 blocks halve operand traffic per flop; use them once they fill the 256 CUs.
    const long big = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * g.batch1 * g.batch2;
    tile = (big >= 192) ? 2 : 1;
  }
  return tile == 2 ? launch<2, 2>(g, stream) : launch<1, 1>(g, stream);
}

}  // namespace gc
