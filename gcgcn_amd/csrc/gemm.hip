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
#include <stdlib.h>

#include "edge_body.hpp"
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "mha_body.hpp"
#include "rowops.hpp"

namespace gc {

template <int TM, int TN, bool AKC, bool BKC, bool ALIGNED, bool RB = false>
__global__ __launch_bounds__(256, (TM * TN == 4 ? 2 : 1)) void gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<TM, TN, AKC, BKC>()];
  const int gx = gridDim.x, gy = gridDim.y;
  const int nwg = gx * gy * gridDim.z;
  const int q = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if (RB && g.rb && g.rb_mode == 1) {   // ragged batch: the live tile rows in XCD-balanced order, then the dead ones (gemm_body.hpp)
    int bx, by, zs, wide;
    if (!tile_of_rows(g, q, gx, gy, bx, by, zs, wide)) return;
    gemm_body<TM, TN, AKC, BKC, ALIGNED, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, bx, by, zs, threadIdx.x, true, PlainOperands(), nullptr, 0, wide);
    return;
  }
  const int b = xcd_remap(q, nwg);
  gemm_body<TM, TN, AKC, BKC, ALIGNED, GC_GEMM_EG, 0, EPI_ALL, PlainOperands, RB>(g, lds, b % gx, (b / gx) % gy, b / (gx * gy));
}

// Stage 1 of a riding column sum: workgroup cb sums one slice of rows for 64 columns (4 waves stride the rows,
// lanes own columns, 8 independent loads in flight per lane).
__device__ __forceinline__ void col_ride_stage1(const ColRide& cr, int cb, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncb = (cr.C + 63) >> 6;
  const int sp = cb / ncb, c = (cb - sp * ncb) * 64 + lane;
  const long rps = (cr.R + COL_RIDE_SLICES - 1) / COL_RIDE_SLICES;
  const long r0 = sp * rps, r1 = min(cr.R, r0 + rps);
  float acc = 0.f;
  if (c < cr.C) {
    long r = r0 + wave;
    for (; r + 28 < r1; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = cr.X[(r + 4 * u) * cr.ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < r1; r += 4) acc += cr.X[r * cr.ld + c];
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && c < cr.C) cr.part[(long)sp * cr.C + c] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
}

// Several independent problems in ONE launch (64x64 tiles, interior shapes only): block -> problem by
// prefix sums of tile counts, layout chosen per problem by a block-uniform branch.
// Problems are laid out longest-K first and every problem starts at a workgroup id that is a multiple of 8, so
// that the XCD-contiguous remap can be applied PER PROBLEM: each XCD receives an equal, contiguous share of every
// problem's tile list.  (One remap over the whole launch would hand XCD 0 the longest problem and XCD 7 the
// shortest.)  The <= 7 padding workgroups per problem exit at once.
// workgroup `idx` past the GEMM tiles (dispatched last, round-robin over the XCDs): the riding column sum
__device__ __forceinline__ void group_tail(const GemmGroup& gg, const int idx, float* __restrict__ lds) {
  const ColRide& cr = gg.col;
  if (cr.ready_slices < 0) {  // sum over heads (sum_h Wlin[:, h, :] for the fused chain backward), eight loads in flight
    const int e = idx * 256 + threadIdx.x;
    if (e < cr.C) {
      const int ld = (int)cr.ld, H = (int)cr.R, k = e / ld, c = e - k * ld;
      float s = 0.f;
      for (int h0 = 0; h0 < H; h0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = cr.X[((long)k * H + min(h0 + u, H - 1)) * ld + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (h0 + u < H) ? v[u] : 0.f;
      }
      cr.out[e] = s;
    }
    return;
  }
  if (cr.ready_slices > 0) {  // partial sums from an earlier launch: out[c] = their sum, slices in order
    col_ride_stage2_block(cr, idx);
    return;
  }
  col_ride_stage1(cr, idx, lds);
}

template <bool RB>
__global__ __launch_bounds__(256, 4) void gemm_group_kernel(const GemmGroup gg) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  const int tiles = gg.tile_begin[gg.nprob];
  if ((int)blockIdx.x >= tiles) {
    group_tail(gg, blockIdx.x - tiles, lds);
    return;
  }
  gemm_group_block<RB>(gg, blockIdx.x, lds);
}

// The same launch with MultiHeadAttention's backward core aboard: mp.count (document, head) pairs spread EVENLY through the
// tile list (behind the tiles they would start when the list has nearly drained and stretch the launch by their own latency).
template <bool RB>
__global__ __launch_bounds__(256, 4) void gemm_group_pass_kernel(const GemmGroup gg, const MhaPass mp) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  const int tiles = gg.tile_begin[gg.nprob], inter = tiles + mp.count;
  const int x = blockIdx.x;
  if (x >= inter) {
    group_tail(gg, x - inter, lds);
    return;
  }
  const int before = (int)((long)x * mp.count / inter);   // pairs among the workgroups [0, x)
  if ((int)((long)(x + 1) * mp.count / inter) > before) {
    mha_core_bwd_body(lds, before, mp.Q, mp.P, mp.dA, mp.dQ, mp.N, mp.D, mp.H, mp.dh, mp.kchunk, mp.alpha, mp.drop);
    return;
  }
  gemm_group_block<RB>(gg, x - before, lds);
}

// Split-K reduce: one thread sums the partials of 4 consecutive outputs (16-byte loads, split order => bitwise
// reproducible) and runs the epilogue.  N % 4 == 0 is guaranteed for split problems (they are interior shapes).
__device__ __forceinline__ void reduce4(const GemmArgs& g, int z, long idx4) {
  const long mn = (long)g.M * g.N;
  const long idx = idx4 * 4;
  if (idx >= mn) return;
  const long nb = (long)g.batch1 * g.batch2;
  int row = (int)(idx / g.N);
  const int col = (int)(idx - (long)row * g.N);
  const int z1 = z / g.batch2, z2 = z - z1 * g.batch2;
  int nsplit = g.splits;
  if (g.rb && g.rb_mode == 1) {  // ragged batch: virtual row -> its place in the padded tensors; rows past the live blocks
    const int bi = row >> 4;     // are zero-stored (outputs that leave the block) or left alone
    const int prow = g.rb[bi] * 16 + (row & 15);
    const int tm = (g.M + 63) >> 6;
    nsplit *= split_width(g, min((*g.rb_n + 3) >> 2, tm), tm);   // the tile workgroups cut K this much finer (gemm_body.hpp)
    if (bi >= *g.rb_n) {
      if (g.rb_zero) {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float* c = g.C + z1 * g.sC1 + z2 * g.sC2 + (long)prow * g.ldc + col;
        c[0] = 0.f, c[1] = 0.f, c[2] = 0.f, c[3] = 0.f;
        if (g.C2) {
          float* c2 = g.C2 + z1 * g.sC21 + z2 * g.sC22 + (long)prow * g.ldc2 + col;
          c2[0] = zero.x, c2[1] = zero.y, c2[2] = zero.z, c2[3] = zero.w;
        }
      }
      return;
    }
    row = prow;
  }
  const float4* w = reinterpret_cast<const float4*>(g.ws + (long)z * mn + idx);
  const long stride4 = nb * mn / 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // split order, eight slabs requested together (one at a time the loop is `splits` dependent round trips: a 64-way split of the
  // classifier head's dynamic-K weight gradients took 20 us on 16 workgroups)
  if (nsplit <= 4) {   // (the block path's factors: the plain loop -- batches with clamped re-reads cost cfg 2 0.4 %)
    for (int s = 0; s < nsplit; ++s) {
      const float4 v = w[(long)s * stride4];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
  } else {
    for (int s0 = 0; s0 < nsplit; s0 += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = w[(long)min(s0 + u, nsplit - 1) * stride4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < nsplit) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
  }
  if (!(g.add || g.bias || g.rowadd || g.rowscale || g.relu || g.accumulate || g.n_valid || g.C2) && g.alpha == 1.f) {
    float* c = g.C + z1 * g.sC1 + z2 * g.sC2 + (long)row * g.ldc + col;
    if ((((uintptr_t)c) & 15) == 0) {
      *reinterpret_cast<float4*>(c) = acc;
    } else {
      c[0] = acc.x, c[1] = acc.y, c[2] = acc.z, c[3] = acc.w;
    }
    return;
  }
  const Epi e = make_epi(g, z1, z2);
  epi_store4(g, e, row, col, acc);
}

__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const GemmGroup gg) {
  int b = blockIdx.x, i = 0;
  if (b >= gg.red_begin[gg.nprob]) {  // stage 2 of the riding column sum, slices in order
    const ColRide& cr = gg.col;
    const int c = (b - gg.red_begin[gg.nprob]) * 256 + threadIdx.x;
    if (c < cr.C) {
      float s = 0.f;
      for (int q = 0; q < COL_RIDE_SLICES; ++q) s += cr.part[(long)q * cr.C + c];
      cr.out[c] = s;
    }
    return;
  }
  while (i + 1 < gg.nprob && b >= gg.red_begin[i + 1]) ++i;
  b -= gg.red_begin[i];
  const GemmArgs& g = gg.p[i];
  if (g.splits <= 1) return;
  const int per = (int)(((long)g.M * g.N / 4 + 255) / 256);
  const int z = b / per;
  reduce4(g, z, (long)(b - z * per) * 256 + threadIdx.x);
}

// Sum the split-K partials in split order (bitwise reproducible) and run the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
  reduce4(g, blockIdx.y, (long)blockIdx.x * 256 + threadIdx.x);
}

int splitk_reduce(const GemmArgs& g, hipStream_t st) {
  ProfScope ps("gemm_splitk_reduce", st);
  dim3 rgrid(cdiv((long)g.M * g.N / 4, 256), g.batch1 * g.batch2);
  hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, st, g);
  return check_launch("gemm_splitk_reduce");
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int TM, int TN, bool ALIGNED>
static int launch(const GemmArgs& g, hipStream_t stream) {
  dim3 grid(cdiv(g.N, 64 * TN), cdiv(g.M, 64 * TM), g.batch1 * g.batch2 * g.splits), block(256);
  const double flops = 2.0 * g.M * g.N * g.K * g.batch1 * g.batch2;
  if constexpr (ALIGNED) {
    if (g.rb) {  // ragged batch: the instantiation with the row-block paths (prepare() kept rb only where they apply)
      if (g.a_kc && !g.b_kc) GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, true, false, true, true>), grid, block, 0, stream, g);
      else if (g.a_kc && g.b_kc) GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, true, true, true, true>), grid, block, 0, stream, g);
      else GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, false, false, true, true>), grid, block, 0, stream, g);
      if (int e = check_launch("gemm")) return e;
      if (g.splits > 1) {
        ProfScope ps("gemm_splitk_reduce", stream);
        dim3 rgrid(cdiv((long)g.M * g.N / 4, 256), g.batch1 * g.batch2);
        hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, g);
        return check_launch("gemm_splitk_reduce");
      }
      return 0;
    }
  }
  if (g.a_kc && !g.b_kc) GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, true, false, ALIGNED>), grid, block, 0, stream, g);
  else if (g.a_kc && g.b_kc) GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, true, true, ALIGNED>), grid, block, 0, stream, g);
  else if (!g.a_kc && !g.b_kc) GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, false, false, ALIGNED>), grid, block, 0, stream, g);
  else GC_LAUNCH_TIMED(g.tag, flops, (gemm_kernel<TM, TN, false, true, ALIGNED>), grid, block, 0, stream, g);
  if (int e = check_launch("gemm")) return e;
  if (g.splits > 1) {
    ProfScope ps("gemm_splitk_reduce", stream);
    dim3 rgrid(cdiv((long)g.M * g.N / 4, 256), g.batch1 * g.batch2);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, g);
    return check_launch("gemm_splitk_reduce");
  }
  return 0;
}

// Split factor of a problem inside a launch whose 64x64 tiles carry `work` tile-k-steps in total: no
// workgroup should run much longer than the average load of one of the 256 x 4 resident slots
// (a K = 2048 tile next to K = 256 tiles would otherwise drain alone), and a launch that cannot fill
// the slots at all is cut until its blocks are ~8 k-steps long.
static int pick_splits(int K, long work) {
  static const int thr_pct = [] {  // GCGCN_SPLIT_PCT: tuning knob (percent of the average slot load), default 250
    const char* e = getenv("GCGCN_SPLIT_PCT");
    return e ? atoi(e) : 250;
  }();
  const long iters = cdiv(K, BK);
  long thr = work / 1024 * thr_pct / 100;
  if (thr < 8) thr = 8;
  int splits = 1;
  while (iters / splits > thr && splits < 16 && K % (splits * 2 * BK) == 0) splits *= 2;
  return splits;
}

// Fill the launcher-owned fields (vec flags, tile, split factor).  Returns the chosen tile (1 or 2) or -1.
static int prepare(GemmArgs& g, int tile, int splits, long group_work) {
  if (!(g.A && g.B && g.C)) { set_error("gemm: null operand"); return -1; }
  if (!(g.M >= 0 && g.N >= 0 && g.K >= 1 && g.batch1 >= 1 && g.batch2 >= 1)) { set_error("gemm: bad shape"); return -1; }
  {  // the epilogue addresses one (batch entry's) [M x ld] slice with 32-bit offsets
    long ldmax = g.ldc;
    if (g.add && g.ldadd > ldmax) ldmax = g.ldadd;
    if (g.C2 && g.ldc2 > ldmax) ldmax = g.ldc2;
    if (g.add2 && g.ldadd2 > ldmax) ldmax = g.ldadd2;
    if ((long)g.M * ldmax >= (1L << 32)) { set_error("gemm: M * ld = %ld exceeds 32-bit epilogue offsets", (long)g.M * ldmax); return -1; }
  }
  const long nb = (long)g.batch1 * g.batch2;
  g.vecA = aligned16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
  g.vecB = aligned16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
  const long t64 = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * nb;
  (void)t64;
  tile = 1;  // measured: 64x64 tiles beat both 128x128 bodies on every product of this path (dropped experiments)
  if (splits == 0) {
    splits = pick_splits(g.K, group_work > 0 ? group_work : t64 * cdiv(g.K, BK));
    // The host never reads n_valid: a long-K problem on the row blocks of a ragged batch that "fills the chip" by its dense tile
    // count (cfg 3's K = 3072 data gradients: 768 tiles, three per compute unit, unsplit) runs with fewer than half of them live,
    // one to a compute unit, at a lone workgroup's pace (336 live tiles of 96 k-tiles: 109 us).  Split it once here; what the launch
    // really finds is dealt with on the device (split_width cuts finer with the workgroups of the dead rows).  Counting every
    // row-block problem at half its dense work instead split the short ones too: cfg 2 ragged 78.0k -> 71.9k docs/s.
    if (splits == 1 && g.rb && g.rb_mode == 1 && cdiv(g.K, BK) >= 48 && t64 <= 1024 && g.K % (2 * BK) == 0 && option("split_widen", 1)) splits = 2;
  }
  if (splits > 1 && (!g.ws || (long)splits * nb * g.M * g.N > g.ws_elems || g.K % (splits * BK) != 0 || g.N % 4 != 0 ||
                     (((uintptr_t)g.ws) & 15) != 0))
    splits = 1;
  g.splits = splits;
  g.ksplit = (splits > 1) ? g.K / splits : g.K;
  if (g.rb) {  // row blocks of a ragged batch: only what the gathering tile body serves, anything else runs dense (equally correct)
    const bool interior = g.vecA && g.vecB && g.M % 64 == 0 && g.N % 64 == 0 && g.K % BK == 0 && g.ksplit % BK == 0 && g.rb_n;
    const bool ok = interior && ((g.rb_mode == 1 && g.a_kc && nb == 1) || (g.rb_mode == 2 && !g.a_kc && !g.b_kc && g.K % 64 == 0 && g.K / 16 <= ROWBLK_LIST_MAX));
    // M-side: a launch that does not fill the chip anyway gains nothing from skipping tiles and pays the list's lookups in
    // its latency-bound prologue (128-tile output projection: 21 vs 16 us measured) -- dense below one tile per compute unit
    // and slot (a problem inside a group launch shares the launch: group_work > 0 keeps it)
    const bool small = g.rb_mode == 1 && group_work <= 0 && (long)cdiv(g.M, 64) * cdiv(g.N, 64) * splits <= 256;
    if (!ok || small) g.rb = nullptr, g.rb_n = nullptr, g.rb_mode = 0, g.rb_zero = 0;
  }
  // a split row-block problem may cut K finer on the device (split_width): as far as the workspace holds the slabs
  g.widen = 1;
  if (g.rb && g.rb_mode == 1 && g.splits > 1 && option("split_widen", 1))
    while (g.widen < 4 && (long)g.splits * 2 * g.widen * nb * g.M * g.N <= g.ws_elems) g.widen *= 2;
  return tile;
}

int gemm(const GemmArgs& g_in, hipStream_t stream, int tile, int splits) {
  GemmArgs g = g_in;
  if (g.M == 0 || g.N == 0) return 0;
  GC_REQUIRE(tile == 0 || tile == 1, "gemm: tile %d (64 x 64 tiles are the only body; 128 x 128 bodies lost every A/B on this path)", tile);
  if (prepare(g, 1, splits, 0) < 0) return 1;
  const long nb = (long)g.batch1 * g.batch2;
  GC_REQUIRE(nb * g.splits <= 65535, "gemm: batch %ld x splits %d exceeds grid.z", nb, g.splits);
  GC_REQUIRE(cdiv(g.M, 64) <= 65535, "gemm: M %d exceeds grid.y", g.M);
  const bool al = g.vecA && g.vecB && g.M % 64 == 0 && g.N % 64 == 0 && g.ksplit % BK == 0;
  static const bool dump = getenv("GCGCN_GROUP_DUMP") != nullptr;   // diagnosis (with the group launches' dump)
  if (dump)
    fprintf(stderr, "gemm: %-12s M %5d N %5d K %5d batch %3ld splits %2d a_kc %d b_kc %d rb_mode %d %s\n", g.tag, g.M, g.N, g.K, nb, g.splits,
            g.a_kc, g.b_kc, g.rb ? g.rb_mode : 0, al ? "interior" : "guarded");
  return al ? launch<1, 1, true>(g, stream) : launch<1, 1, false>(g, stream);
}

// Independent problems in one launch (plus at most one reduce launch).  Problems that are not interior
// 64x64 shapes fall back to their own launches.
// The head-feature chunk with which a (document, head) pair's scratch (score tile + one chunk of Q_h) fits a tile workgroup's
// LDS: the stand-alone kernel's chunk, or 0.  (A shorter chunk would fit wide heads too -- 32 features at a time for cfg 3's 192
// -- but such passengers use two of their four waves and run six dependent passes: the group launch grew by 45 us to save a
// 23 us launch.  Measured, not kept.)
int gemm_group_mha_chunk(int dh) {
  const size_t room = sizeof(float) * lds_floats<1, 1, true, true>();
  const int c = mha_chunk(dh);
  return sizeof(float) * (MT * MS + MT * (c + 1)) <= room ? c : 0;
}
bool gemm_group_can_carry_mha(int dh) { return gemm_group_mha_chunk(dh) > 0; }

int gemm_group(const GemmArgs* probs, int n, hipStream_t stream, const ColRide* col, bool* col_later, const MhaPass* mha) {
  if (col_later) *col_later = false;
  GemmGroup gg;
  gg.nprob = 0;
  long work = 0;  // tile-k-steps of the whole launch
  for (int i = 0; i < n; ++i)
    work += (long)cdiv(probs[i].M, 64) * cdiv(probs[i].N, 64) * probs[i].batch1 * probs[i].batch2 * cdiv(probs[i].K, BK);
  int tiles = 0, reds = 0;
  double flops = 0;
  bool any_split = false;
  long ws_used = 0;
  // longest blocks first: workgroups are dispatched in tile order, and a K = 2048 tile started last would
  // run alone long after the K = 256 tiles of the same launch have drained
  int order[64];
  GC_REQUIRE(n <= 64, "gemm_group: too many problems");
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && probs[order[j]].K > probs[order[j - 1]].K; --j) {
      const int t = order[j];
      order[j] = order[j - 1], order[j - 1] = t;
    }
  // pass 1: fix every problem's split factor and count its workgroups
  GemmArgs prep[64];
  bool groupable[64];
  int ng = 0;
  for (int oi = 0; oi < n; ++oi) {
    GemmArgs& g = prep[oi];
    g = probs[order[oi]];
    groupable[oi] = false;
    if (g.M == 0 || g.N == 0) continue;
    if (prepare(g, 1, 0, work) < 0) return 1;
    const bool al = g.vecA && g.vecB && g.M % 64 == 0 && g.N % 64 == 0 && g.ksplit % BK == 0;
    groupable[oi] = al && ng < GemmGroup::MAXP;
    if (groupable[oi]) ++ng;
  }
  // (Until round 5 a launch that overflowed the 1024 resident slots by <= 128 workgroups had its smallest problem peeled into a
  // launch of its own -- "a whole extra round for a few workgroups".  With the tile order and the kernels of this round the
  // separate launch costs more than the tail: cfg 2's Q projection, 128 tiles, was an 11 us launch in front of a 2048-tile
  // group; inside it the step is 0.5066 against 0.5124 ms, ragged 0.4085 against 0.4142, cfg 3 / cfg 5 -0.2 / -0.4 %, cfg 1 a
  // tie.  The rule is gone.)
  // pass 2: ungroupable problems first (own launches), then the group
  for (int oi = 0; oi < n; ++oi) {
    if (groupable[oi] || probs[order[oi]].M == 0 || probs[order[oi]].N == 0) continue;
    if (int e = gemm(probs[order[oi]], stream)) return e;
  }
  for (int oi = 0; oi < n; ++oi) {
    if (!groupable[oi]) continue;
    GemmArgs g = prep[oi];
    const long nb = (long)g.batch1 * g.batch2;
    const long own = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * nb;
    // the group shares one workspace: give each split problem its own slice
    if (g.splits > 1) {
      while (g.widen > 1 && ws_used + (long)g.splits * g.widen * nb * g.M * g.N > g.ws_elems) g.widen /= 2;
      const long need = (long)g.splits * g.widen * nb * g.M * g.N;
      if (ws_used + need > g.ws_elems) {  // no room left: this problem goes unsplit
        g.splits = 1, g.ksplit = g.K, g.widen = 1;
      } else {
        g.ws = g.ws + ws_used;
        ws_used += need, any_split = true;
      }
    }
    tiles = (tiles + 7) & ~7;
    gg.tile_begin[gg.nprob] = tiles;
    gg.tile_count[gg.nprob] = (int)(own * g.splits);
    gg.tile_first[gg.nprob] = 0, gg.tile_take[gg.nprob] = (int)(own * g.splits);
    gg.red_begin[gg.nprob] = reds;
    tiles += (int)(own * g.splits);
    if (g.splits > 1) reds += (int)(nb * cdiv((long)g.M * g.N / 4, 256));
    flops += 2.0 * g.M * g.N * g.K * nb;
    gg.p[gg.nprob++] = g;
  }
  static const bool dump = getenv("GCGCN_GROUP_DUMP") != nullptr;   // diagnosis: what each group launch is made of
  if (dump) {
    fprintf(stderr, "gemm_group: %d problems, %d tile workgroups, mha pairs %d\n", gg.nprob, tiles, mha ? mha->count : 0);
    for (int i = 0; i < gg.nprob; ++i) {
      const GemmArgs& g = gg.p[i];
      fprintf(stderr, "  %-12s M %5d N %5d K %5d batch %3d splits %2d (widen %d) a_kc %d b_kc %d rb_mode %d tiles %5d\n", g.tag, g.M, g.N, g.K,
              g.batch1 * g.batch2, g.splits, g.widen, g.a_kc, g.b_kc, g.rb ? g.rb_mode : 0, gg.tile_count[i]);
    }
  }
  const bool ride3 = col && col->ready_slices < 0 && col->C > 0;   // head sum: no partials, no second stage
  const bool ride2 = col && col->ready_slices > 0 && col->C > 0;
  const bool ride = !ride2 && !ride3 && col && col->X && col->C > 0 && col->R > 0;
  if (ride || ride2) GC_REQUIRE(col->out && col->part, "gemm_group: column ride without out / part");
  if (ride3) GC_REQUIRE(col->X && col->out && col->R > 0 && col->ld > 0 && col->C == col->ld * col->ld, "gemm_group: bad head sum");
  static_assert(sizeof(GemmGroup) + sizeof(MhaPass) + 32 <= 4096, "gemm_group_pass_kernel: kernel arguments exceed 4 KB");
  if (gg.nprob == 0 && mha && mha->count > 0)   // nothing to ride on: the pairs get their launch
    if (int e = mha_core_bwd(mha->Q, mha->P, mha->dA, mha->dQ, mha->count / mha->H, mha->N, mha->D, mha->H, mha->alpha, mha->drop, stream)) return e;
  if (gg.nprob == 0) {  // nothing to ride on
    if (ride3) return mask_rows(nullptr, nullptr, 0, (int)col->ld, 1, nullptr, make_drop(nullptr, 0, 0.f), stream, col->X, col->out, (int)col->R);
    if (ride2) return colsum(col->part, nullptr, col->out, col->ready_slices, col->C, col->C, 1, 0, 0, 0, 0, nullptr, stream);
    return ride ? colsum(col->X, nullptr, col->out, col->R, col->C, col->ld, 1, 0, 0, 0, 0, col->part, stream) : 0;
  }
  gg.tile_begin[gg.nprob] = tiles;
  gg.red_begin[gg.nprob] = reds;
  int col1 = 0, col2 = 0;
  if (ride) {
    gg.col = *col;
    col1 = cdiv(col->C, 64) * COL_RIDE_SLICES, col2 = cdiv(col->C, 256);
  } else if (ride2 || ride3) {
    gg.col = *col;
    col1 = cdiv(col->C, 256);
  }
  bool any_rb = false;
  for (int i = 0; i < gg.nprob; ++i) any_rb = any_rb || gg.p[i].rb != nullptr;
  if (mha && mha->count > 0) {
    if (any_rb) GC_LAUNCH_TIMED("gemm_group", flops, gemm_group_pass_kernel<true>, dim3(tiles + mha->count + col1), dim3(256), 0, stream, gg, *mha);
    else GC_LAUNCH_TIMED("gemm_group", flops, gemm_group_pass_kernel<false>, dim3(tiles + mha->count + col1), dim3(256), 0, stream, gg, *mha);
  } else {
    if (any_rb) GC_LAUNCH_TIMED("gemm_group", flops, gemm_group_kernel<true>, dim3(tiles + col1), dim3(256), 0, stream, gg);
    else GC_LAUNCH_TIMED("gemm_group", flops, gemm_group_kernel<false>, dim3(tiles + col1), dim3(256), 0, stream, gg);
  }
  if (int e = check_launch("gemm_group")) return e;
  if (ride && !any_split && col_later) {  // nothing to reduce: the caller folds stage 2 into a kernel of its own
    *col_later = true;
    return 0;
  }
  if (any_split || ride) {
    ProfScope ps("gemm_splitk_reduce", stream);
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(reds + col2), dim3(256), 0, stream, gg);
    return check_launch("gemm_group_reduce");
  }
  return 0;
}

// ---- one dimension known only on the device (gemm.hpp) --------------------------------------------------------
// vb / vgrid: this workgroup's index and the number of workgroups walking the problem's tile list (the whole grid, or one
// problem's share of a pair launch)
template <bool AKC, bool BKC, bool ALIGNED>
__device__ __forceinline__ void gemm_dyn_walk(GemmArgs& g, float* __restrict__ lds, const int* __restrict__ cnt, const int dyn,
                                              const int splits, const int vb, const int vgrid) {
  const int c = *cnt;
  const int cr = ALIGNED ? (c + 63) & ~63 : c;
  if (dyn == 1) {
    g.M = cr, g.splits = 1, g.ksplit = g.K;
    const int tn = (g.N + 63) >> 6, tiles = ((cr + 63) >> 6) * tn;
    for (int tile = vb; tile < tiles; tile += vgrid) {
      gemm_body<1, 1, AKC, BKC, ALIGNED>(g, lds, tile % tn, tile / tn, 0);
      __syncthreads();  // the next tile restages LDS
    }
    return;
  }
  g.K = cr, g.splits = splits;
  g.ksplit = ((((cr + splits - 1) / splits) + BK - 1) / BK) * BK;
  if (g.ksplit < BK) g.ksplit = BK;
  const int tn = (g.N + 63) >> 6, tm = (g.M + 63) >> 6, tiles = tm * tn * splits;
  for (int tile = vb; tile < tiles; tile += vgrid) {
    const int sp = tile % splits, r = tile / splits;
    const int bx = r % tn, by = r / tn;
    if (sp * g.ksplit >= cr) {  // this K slice is empty
      if (splits > 1) {         // its partial tile is zero
        float* __restrict__ W = g.ws + (long)sp * g.M * g.N;
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
          const int row = by * 64 + (e >> 6), col = bx * 64 + (e & 63);
          if (row < g.M && col < g.N) W[(long)row * g.N + col] = 0.f;
        }
      }                      // (cr == 0 with one slice: gemm_dyn_zero_kernel defines C)
      continue;
    }
    gemm_body<1, 1, AKC, BKC, ALIGNED>(g, lds, bx, by, sp);
    __syncthreads();
  }
}

template <bool AKC, bool BKC, bool ALIGNED>
__global__ __launch_bounds__(256, 4) void gemm_dyn_kernel(GemmArgs g, const int* __restrict__ cnt, int dyn, int splits) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, AKC, BKC>()];
  gemm_dyn_walk<AKC, BKC, ALIGNED>(g, lds, cnt, dyn, splits, (int)blockIdx.x, (int)gridDim.x);
}

// The two gradients of one Linear layer on a device-side row count in ONE launch: the weight gradient (K = *cnt: operands
// [K][.], split over `splits` slices) on the first gridW workgroups, the data gradient (M = *cnt: dY [M][K] times W [K][N]) on
// the rest.  Interior shapes only (gemm_dyn_pair below checks).
// KIND 0: weight gradient + data gradient of one layer; 1: two weight gradients over the same rows (the classifier head's dW_c
// halves); 2: two data-gradient-shaped products over the same rows (M = *cnt both: the head's dout W_c halves).
template <int KIND>
__global__ __launch_bounds__(256, 4) void gemm_dyn_pair_kernel(GemmArgs gw, GemmArgs gx, const int* __restrict__ cnt, int splits,
                                                               int gridW) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, false>()];
  const int vb2 = (int)blockIdx.x - gridW, vg2 = (int)gridDim.x - gridW;
  if ((int)blockIdx.x < gridW) {
    if (KIND == 2) gemm_dyn_walk<true, false, true>(gw, lds, cnt, 1, 1, (int)blockIdx.x, gridW);
    else gemm_dyn_walk<false, false, true>(gw, lds, cnt, 2, splits, (int)blockIdx.x, gridW);
  } else if (KIND == 1) gemm_dyn_walk<false, false, true>(gx, lds, cnt, 2, splits, vb2, vg2);
  else gemm_dyn_walk<true, false, true>(gx, lds, cnt, 1, 1, vb2, vg2);
}
// the two weight gradients' reduces in one launch (blockIdx.y picks the problem)
__global__ __launch_bounds__(256) void splitk_reduce_pair_kernel(const GemmArgs ga, const GemmArgs gb) {
  reduce4(blockIdx.y ? gb : ga, 0, (long)blockIdx.x * 256 + threadIdx.x);
}

// *cnt == 0 with dyn == 2: every slice is empty; C must still be defined (zeros).
__global__ __launch_bounds__(256) void gemm_dyn_zero_kernel(GemmArgs g, const int* __restrict__ cnt) {
  if (*cnt != 0) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)g.M * g.N) return;
  const int row = (int)(e / g.N), col = (int)(e - (long)row * g.N);
  if (!g.accumulate) g.C[(long)row * g.ldc + col] = 0.f;
}

int gemm_dyn(const GemmArgs& g_in, const int* cnt, int dyn, long cap, hipStream_t st) {
  GemmArgs g = g_in;
  GC_REQUIRE(cnt && (dyn == 1 || dyn == 2) && cap >= 0, "gemm_dyn: bad arguments");
  GC_REQUIRE(g.batch1 == 1 && g.batch2 == 1, "gemm_dyn: batched problems are not supported");
  if (cap == 0) cap = 1;
  if (dyn == 1) g.M = (int)((cap + 63) & ~63L); else g.K = (int)((cap + 63) & ~63L);
  if (prepare(g, 1, 1, 0) < 0) return 1;
  const bool al = g.vecA && g.vecB && g.N % 64 == 0 && (dyn == 1 ? g.K % BK == 0 : g.M % 64 == 0);
  int splits = 1;
  if (dyn == 2) {
    // enough K slices to fill the chip with the few output tiles of a weight gradient; each slice >= 8 k-steps
    const long tiles = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
    long want = (1024 + tiles - 1) / tiles, most = cap / (8 * BK);
    if (most < 1) most = 1;
    splits = (int)(want < most ? want : most);
    if (splits > 64) splits = 64;
    if (splits > 1 && (!g.ws || (long)splits * g.M * g.N > g.ws_elems || g.N % 4 != 0 || (((uintptr_t)g.ws) & 15) != 0)) splits = 1;
    if (splits == 1) {
      dim3 zg(cdiv((long)g.M * g.N, 256));
      hipLaunchKernelGGL(gemm_dyn_zero_kernel, zg, dim3(256), 0, st, g, cnt);
    }
  }
  g.splits = splits;
  const long tiles_cap = dyn == 1 ? (long)cdiv(cap, 64) * cdiv(g.N, 64) : (long)cdiv(g.M, 64) * cdiv(g.N, 64) * splits;
  const unsigned grid = (unsigned)(tiles_cap < 1024 ? (tiles_cap > 0 ? tiles_cap : 1) : 1024);
  const double flops = 0.0;  // data-dependent: not counted by the host-side timer
#define GC_DYN(AK, BKc, AL) GC_LAUNCH_TIMED("gemm_dyn", flops, (gemm_dyn_kernel<AK, BKc, AL>), dim3(grid), dim3(256), 0, st, g, cnt, dyn, splits)
  if (al) {
    if (g.a_kc && !g.b_kc) GC_DYN(true, false, true);
    else if (g.a_kc && g.b_kc) GC_DYN(true, true, true);
    else if (!g.a_kc && !g.b_kc) GC_DYN(false, false, true);
    else GC_DYN(false, true, true);
  } else {
    if (g.a_kc && !g.b_kc) GC_DYN(true, false, false);
    else if (g.a_kc && g.b_kc) GC_DYN(true, true, false);
    else if (!g.a_kc && !g.b_kc) GC_DYN(false, false, false);
    else GC_DYN(false, true, false);
  }
#undef GC_DYN
  if (int e = check_launch("gemm_dyn")) return e;
  if (splits > 1) {  // sums `splits` partial slabs (empty slices wrote zeros) and runs the epilogue
    ProfScope ps("gemm_splitk_reduce", st);
    dim3 rgrid(cdiv((long)g.M * g.N / 4, 256), 1);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, st, g);
    return check_launch("gemm_dyn_reduce");
  }
  return 0;
}

// dW = dY^T X (dyn = 2) and dX (+)= dY W (dyn = 1) of one Linear layer whose row count lives on the device: one launch + the
// weight gradient's reduce instead of two launches + the reduce.  Falls back to two gemm_dyn calls for shapes the pair kernel
// does not serve (guarded tiles, an unsplittable weight gradient).
int gemm_dyn_pair(const GemmArgs& gw_in, const GemmArgs& gx_in, const int* cnt, long cap, hipStream_t st) {
  GemmArgs gw = gw_in, gx = gx_in;
  GC_REQUIRE(cnt && cap >= 0, "gemm_dyn_pair: bad arguments");
  if (cap == 0) cap = 1;
  gw.K = (int)((cap + 63) & ~63L), gx.M = (int)((cap + 63) & ~63L);
  bool ok = gw.batch1 == 1 && gw.batch2 == 1 && gx.batch1 == 1 && gx.batch2 == 1 && !gw.a_kc && !gw.b_kc && gx.a_kc && !gx.b_kc;
  ok = ok && prepare(gw, 1, 1, 0) >= 0 && prepare(gx, 1, 1, 0) >= 0;
  ok = ok && gw.vecA && gw.vecB && gw.N % 64 == 0 && gw.M % 64 == 0 && gx.vecA && gx.vecB && gx.N % 64 == 0 && gx.K % BK == 0;
  int splits = 1;
  if (ok) {   // the split factor of gemm_dyn (dyn = 2)
    const long tiles = (long)cdiv(gw.M, 64) * cdiv(gw.N, 64);
    long want = (1024 + tiles - 1) / tiles, most = cap / (8 * BK);
    if (most < 1) most = 1;
    splits = (int)(want < most ? want : most);
    if (splits > 64) splits = 64;
    ok = splits > 1 && gw.ws && (long)splits * gw.M * gw.N <= gw.ws_elems && gw.N % 4 == 0 && (((uintptr_t)gw.ws) & 15) == 0;
  }
  if (!ok) {
    if (int e = gemm_dyn(gw_in, cnt, 2, cap, st)) return e;
    return gemm_dyn(gx_in, cnt, 1, cap, st);
  }
  gw.splits = splits;
  const long tw = (long)cdiv(gw.M, 64) * cdiv(gw.N, 64) * splits, tx = (long)cdiv(cap, 64) * cdiv(gx.N, 64);
  const int gridW = (int)(tw < 1024 ? tw : 1024), gridX = (int)(tx < 1024 ? (tx > 0 ? tx : 1) : 1024);
  GC_LAUNCH_TIMED("gemm_dyn", 0.0, gemm_dyn_pair_kernel<0>, dim3(gridW + gridX), dim3(256), 0, st, gw, gx, cnt, splits, gridW);
  if (int e = check_launch("gemm_dyn_pair")) return e;
  ProfScope ps("gemm_splitk_reduce", st);
  dim3 rgrid(cdiv((long)gw.M * gw.N / 4, 256), 1);
  hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, st, gw);
  return check_launch("gemm_dyn_reduce");
}

// Two weight gradients over the same device-side row count and of the same shape (dyn = 2 both): one launch + one reduce launch.
int gemm_dyn_pair_ww(const GemmArgs& ga_in, const GemmArgs& gb_in, const int* cnt, long cap, hipStream_t st) {
  GemmArgs ga = ga_in, gb = gb_in;
  GC_REQUIRE(cnt && cap >= 0, "gemm_dyn_pair_ww: bad arguments");
  if (cap == 0) cap = 1;
  ga.K = gb.K = (int)((cap + 63) & ~63L);
  bool ok = ga.batch1 == 1 && ga.batch2 == 1 && gb.batch1 == 1 && gb.batch2 == 1 && !ga.a_kc && !ga.b_kc && !gb.a_kc && !gb.b_kc &&
            ga.M == gb.M && ga.N == gb.N;
  ok = ok && prepare(ga, 1, 1, 0) >= 0 && prepare(gb, 1, 1, 0) >= 0;
  ok = ok && ga.vecA && ga.vecB && gb.vecA && gb.vecB && ga.N % 64 == 0 && ga.M % 64 == 0;
  int splits = 1;
  if (ok) {
    const long tiles = (long)cdiv(ga.M, 64) * cdiv(ga.N, 64);
    long want = (1024 + tiles - 1) / tiles, most = cap / (8 * BK);
    if (most < 1) most = 1;
    splits = (int)(want < most ? want : most);
    if (splits > 64) splits = 64;
    const long slab = (long)splits * ga.M * ga.N;   // each problem's partials: the workspace holds both, one behind the other
    ok = splits > 1 && ga.ws && 2 * slab <= ga.ws_elems && ga.N % 4 == 0 && (((uintptr_t)ga.ws) & 15) == 0 && slab % 4 == 0;
    if (ok) gb.ws = ga.ws + slab;
  }
  if (!ok) {
    if (int e = gemm_dyn(ga_in, cnt, 2, cap, st)) return e;
    return gemm_dyn(gb_in, cnt, 2, cap, st);
  }
  ga.splits = gb.splits = splits;
  const long tw = (long)cdiv(ga.M, 64) * cdiv(ga.N, 64) * splits;
  const int gridW = (int)(tw < 1024 ? tw : 1024);
  GC_LAUNCH_TIMED("gemm_dyn", 0.0, gemm_dyn_pair_kernel<1>, dim3(2 * gridW), dim3(256), 0, st, ga, gb, cnt, splits, gridW);
  if (int e = check_launch("gemm_dyn_pair_ww")) return e;
  ProfScope ps("gemm_splitk_reduce", st);
  dim3 rgrid(cdiv((long)ga.M * ga.N / 4, 256), 2);
  hipLaunchKernelGGL(splitk_reduce_pair_kernel, rgrid, dim3(256), 0, st, ga, gb);
  return check_launch("gemm_dyn_reduce");
}

// Two products whose M is the same device-side row count (dyn = 1 both, A [M][K] row-major, B [K][N]): one launch.
int gemm_dyn_pair_xx(const GemmArgs& ga_in, const GemmArgs& gb_in, const int* cnt, long cap, hipStream_t st) {
  GemmArgs ga = ga_in, gb = gb_in;
  GC_REQUIRE(cnt && cap >= 0, "gemm_dyn_pair_xx: bad arguments");
  if (cap == 0) cap = 1;
  ga.M = gb.M = (int)((cap + 63) & ~63L);
  bool ok = ga.batch1 == 1 && ga.batch2 == 1 && gb.batch1 == 1 && gb.batch2 == 1 && ga.a_kc && !ga.b_kc && gb.a_kc && !gb.b_kc;
  ok = ok && prepare(ga, 1, 1, 0) >= 0 && prepare(gb, 1, 1, 0) >= 0;
  ok = ok && ga.vecA && ga.vecB && gb.vecA && gb.vecB && ga.N % 64 == 0 && gb.N % 64 == 0 && ga.K % BK == 0 && gb.K % BK == 0;
  if (!ok) {
    if (int e = gemm_dyn(ga_in, cnt, 1, cap, st)) return e;
    return gemm_dyn(gb_in, cnt, 1, cap, st);
  }
  const long ta = (long)cdiv(cap, 64) * cdiv(ga.N, 64), tb = (long)cdiv(cap, 64) * cdiv(gb.N, 64);
  const int gridA = (int)(ta < 1024 ? (ta > 0 ? ta : 1) : 1024), gridB = (int)(tb < 1024 ? (tb > 0 ? tb : 1) : 1024);
  GC_LAUNCH_TIMED("gemm_dyn", 0.0, gemm_dyn_pair_kernel<2>, dim3(gridA + gridB), dim3(256), 0, st, ga, gb, cnt, 1, gridA);
  return check_launch("gemm_dyn_pair_xx");
}

// ---- deferred problems ------------------------------------------------------------------------------------------
// One DeferQueue per backward pass (owned by the host side, gcgcn_defer_create / _destroy): no process-wide state, so
// two passes -- other models, other devices, other threads -- never see each other's problems, and a pass that fails
// half-way simply drops its queue.
bool gemm_defer(DeferQueue* q, const GemmArgs& g_in) {
  if (!q) return false;
  GemmArgs g = g_in;
  if (q->n >= DeferQueue::CAP || g.M == 0 || g.N == 0) return false;
  if (prepare(g, 1, 1, 0) < 0) return false;
  if (!(g.vecA && g.vecB && g.M % 64 == 0 && g.N % 64 == 0 && g.K % BK == 0)) return false;
  g.ws = nullptr, g.ws_elems = 0;  // unsplit: the whole K inside one workgroup
  q->done[q->n] = 0;
  q->p[q->n++] = g;
  return true;
}

bool gemm_defer_col2(DeferQueue* q, const ColRide& c) {
  if (!q || q->ncol2 >= DeferQueue::COLCAP || c.ready_slices <= 0 || !c.part || !c.out || c.C <= 0) return false;
  q->col2[q->ncol2++] = c;
  return true;
}
bool gemm_take_deferred_col2(DeferQueue* q, ColRide& out) {
  if (!q || q->ncol2 == 0) return false;
  out = q->col2[--q->ncol2];
  return true;
}

static int tiles_of(const GemmArgs& g);
// longest K first (they run the longest: start them first); small_first: among equal K the problems with the fewest tiles
// lead -- a carrier with a tile budget then completes whole small problems instead of a slice of a big one, which keeps the
// NUMBER of parked problems within what the last carrier's argument block holds (GemmGroup::MAXP)
static void sort_parked(DeferQueue* q, bool small_first = false) {
  auto before = [&](const GemmArgs& a, int da, const GemmArgs& b, int db) {
    if (a.K != b.K) return a.K > b.K;
    return small_first && tiles_of(a) - da < tiles_of(b) - db;
  };
  for (int i = 1; i < q->n; ++i)
    for (int j = i; j > 0 && before(q->p[j], q->done[j], q->p[j - 1], q->done[j - 1]); --j) {
      const GemmArgs t = q->p[j];
      const int d = q->done[j];
      q->p[j] = q->p[j - 1], q->p[j - 1] = t;
      q->done[j] = q->done[j - 1], q->done[j - 1] = d;
    }
}

static double flops_of(const GemmArgs& g) { return 2.0 * g.M * g.N * g.K * g.batch1 * g.batch2; }
static int tiles_of(const GemmArgs& g) { return (g.M >> 6) * (g.N >> 6) * g.batch1 * g.batch2; }

// Move parked work into gg: up to G::MAXP problems, at most max_tiles tiles in all; a problem is split when the budget
// ends inside it (its remaining tiles stay parked).  Returns the number of workgroups (one per tile, ranges 8-aligned).
template <class G>
static int take_parked(DeferQueue* q, G& gg, double* flops, long max_tiles, bool small_first = false) {
  gg.nprob = 0;
  gg.tile_begin[0] = 0;
  if (!q || q->n == 0 || max_tiles <= 0) return 0;
  sort_parked(q, small_first);
  int wgs = 0, np = 0, keep = 0;
  for (int i = 0; i < q->n; ++i) {
    const GemmArgs& g = q->p[i];
    const int total = tiles_of(g), avail = total - q->done[i];
    const long take = (np < G::MAXP && max_tiles > 0) ? (avail < max_tiles ? avail : max_tiles) : 0;
    if (take > 0) {
      wgs = (wgs + 7) & ~7;
      gg.tile_begin[np] = wgs, gg.tile_count[np] = total, gg.tile_first[np] = q->done[i], gg.tile_take[np] = (int)take;
      gg.red_begin[np] = 0;
      gg.p[np++] = g;
      wgs += (int)take;
      max_tiles -= take;
      if (flops) *flops += flops_of(g) * (double)take / total;
    }
    if (q->done[i] + take < total) {  // (part of) the problem stays parked
      q->p[keep] = g, q->done[keep] = q->done[i] + (int)take;
      ++keep;
    }
  }
  q->n = keep;
  for (int i = keep; i < DeferQueue::CAP; ++i) q->done[i] = 0;
  gg.nprob = np;
  gg.tile_begin[np] = wgs, gg.red_begin[np] = 0;
  return wgs;
}

int gemm_take_deferred(DeferQueue* q, GemmGroup& gg, double* flops) { return take_parked(q, gg, flops, 1L << 40); }
int gemm_take_deferred_pairs(DeferQueue* q, GemmGroup4& gg, double* flops, long max_wgs, bool small_first) {
  return take_parked(q, gg, flops, max_wgs, small_first);
}

// tiles of parked problems, unsplit, as a launch of their own (what a carrying kernel would have run as passengers)
template <bool RB>
__global__ __launch_bounds__(256, 4) void gemm_parked_kernel(const GemmGroup gg) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  gemm_group_block<RB>(gg, blockIdx.x, lds);
}

int gemm_flush_deferred(DeferQueue* q, hipStream_t stream) {
  while (q && q->n > 0) {
    bool partial = false;
    for (int i = 0; i < q->n; ++i) partial = partial || q->done[i] > 0;
    if (partial) {  // the rest of a partly carried problem: same unsplit tiles, same tile list
      GemmGroup gg;
      double fl = 0;
      const int wgs = gemm_take_deferred(q, gg, &fl);
      if (wgs > 0) {
        bool any_rb = false;
        for (int i = 0; i < gg.nprob; ++i) any_rb = any_rb || gg.p[i].rb != nullptr;
        if (any_rb) GC_LAUNCH_TIMED("gemm_group", fl, gemm_parked_kernel<true>, dim3(wgs), dim3(256), 0, stream, gg);
        else GC_LAUNCH_TIMED("gemm_group", fl, gemm_parked_kernel<false>, dim3(wgs), dim3(256), 0, stream, gg);
        if (int e = check_launch("gemm_parked")) return e;
      }
      continue;
    }
    GemmArgs probs[GemmGroup::MAXP];
    const int n = q->n < GemmGroup::MAXP ? q->n : GemmGroup::MAXP;
    for (int i = 0; i < n; ++i) probs[i] = q->p[q->n - n + i];
    q->n -= n;
    if (int e = gemm_group(probs, n, stream)) return e;
  }
  ColRide c2;
  while (gemm_take_deferred_col2(q, c2))   // second stages of column sums that met no carrier
    if (int e = colsum(c2.part, nullptr, c2.out, c2.ready_slices, c2.C, c2.C, 1, 0, 0, 0, 0, nullptr, stream)) return e;
  return 0;
}

}  // namespace gc
