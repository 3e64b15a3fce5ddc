// Classifier head: GCGCN_glove.py:306-307 and 344-358.
//
//   feats = cat(node_feats..., ner_emb[node_type])                         [N, F]
//   eh[i, j] = tanh(dense_layer(cat(feats[j], dis_embed[dis_plus + rel[i, j]])))        "head" side: column entity
//   et[i, j] = tanh(dense_layer(cat(feats[i], dis_embed[dis_plus - rel[i, j]])))        "tail" side: row entity
//   logits[i, j, :] = Bilinear(eh, et) + Linear(cat(eh, et))               [N, N, R]
//
// The dense layer is linear in its concatenated input, so it splits into a per-entity term U[n] = W_f feats[n] + b,
// a 7-row table for the entity type and a 21-row table for the relative-position id; eh / et are one gather + tanh per
// pair -- no [N, N, F + 20] tensor.  The bilinear form, 2 N^2 128^2 R flops per document (13 GFLOP at N = 64, more than
// the whole graph path), runs on the exact-fp32 MFMA as ONE product per pass with K = 128 * 128 (+ 256 for the linear
// part): the operand row of pair p is the outer product eh[p] (x) et[p], generated in registers while the tile is staged
// (gemm_body's operand policy) -- the 8.6 GB a [pairs, 16384] operand would take at B = 32 never exists.  Backward: the
// same trick three times (d eh: rows dout[p] (x) et[p] against W viewed [(r, b), a]; d et: dout[p] (x) eh[p] against
// [(r, a), b]; d W: dout^T against the generated [pairs, 16384] operand).
#include <stdlib.h>
#include <string.h>

#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int HW = 128;  // hidden width of eh / et (hidden_size is hard-coded to 128 in the reference, glove:234)

struct HeadOps {
  const float* P;   // [rows, ldp]: the factor indexed by k / 128
  const float* Q;   // [rows, ldq]: the factor indexed by k % 128
  const float* W2;  // forward: classification_layer_01.weight [R, 256] behind the bilinear weight
  long ldp, ldq, ldw2;
  int KB;           // K of the outer-product segment
  int nmax;         // rows of the weight operand that exist (R); beyond: clamped (their outputs are never stored)
  int rows;         // pairs
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// MODE 1 forward  : A gen [pairs, KB + 256] = [eh (x) et | eh | et],   B = W_b [R][16384] | W_c [R][256]   (k-contiguous rows)
// MODE 2 d eh     : A gen [pairs, R * 128]  = dout (x) et,              B[n = a][k = (r, b)] = W_b[r][a][b]
// MODE 3 d et     : A gen [pairs, R * 128]  = dout (x) eh,              B[k = (r, a)][n = b] = W_b[r][a][b]  (natural layout)
// MODE 4 d W_b    : A = doutp [pairs][128] as [K][M],                   B gen [K = pairs][N = (a, b)] = eh (x) et
template <int MODE>
struct HeadOperands {
  HeadOps o;
  template <int BMN, bool KC, bool ALIGNED>
  __device__ __forceinline__ void load_a(float (&r)[BMN / 32][4], const GemmArgs& g, const float* __restrict__ A, int m0, int k0,
                                         int kend, int t) const {
#pragma unroll
    for (int q = 0; q < BMN / 32; ++q) {
      const int f = t + 256 * q;
      float4 v;
      if (MODE != 4) {
        const long row = min(m0 + (f >> 3), o.rows - 1);   // clamped: rows past the end are computed and never stored
        const int kq = k0 + ((f & 7) << 2);
        if (MODE == 1 && k0 >= o.KB) {
          v = (k0 < o.KB + HW) ? ld4(o.P + row * o.ldp + (kq - o.KB)) : ld4(o.Q + row * o.ldq + (kq - o.KB - HW));
        } else {
          const float p = o.P[row * o.ldp + (k0 >> 7)];
          v = ld4(o.Q + row * o.ldq + (kq & (HW - 1)));
          v.x *= p, v.y *= p, v.z *= p, v.w *= p;
        }
      } else {
        const int k = k0 + (f >> 4), col = m0 + ((f & 15) << 2);
        v = ld4(A + (long)min(k, o.rows - 1) * g.lda + col);
        if (k >= o.rows) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  }
  template <int BMN, bool KC, bool ALIGNED>
  __device__ __forceinline__ void load_b(float (&r)[BMN / 32][4], const GemmArgs& g, const float* __restrict__ B, int n0, int k0,
                                         int kend, int t) const {
#pragma unroll
    for (int q = 0; q < BMN / 32; ++q) {
      const int f = t + 256 * q;
      float4 v;
      if (MODE == 1) {         // [N][K] rows of W_b, then of W_c
        const long n = min(n0 + (f >> 3), o.nmax - 1);
        const int kq = k0 + ((f & 7) << 2);
        v = (k0 < o.KB) ? ld4(B + n * (long)o.KB + kq) : ld4(o.W2 + n * o.ldw2 + (kq - o.KB));
      } else if (MODE == 2) {  // [N = a][K = (r, b)]: element at W_b + r * 16384 + a * 128 + b
        const long n = n0 + (f >> 3);
        const int kq = k0 + ((f & 7) << 2);
        v = ld4(B + (long)(k0 >> 7) * (HW * HW) + n * HW + (kq & (HW - 1)));
      } else if (MODE == 3) {  // [K = (r, a)][N = b]: the natural layout
        const long k = k0 + (f >> 4);
        v = ld4(B + k * HW + n0 + ((f & 15) << 2));
      } else {                 // generated [K = pair][N = (a, b)]
        const long k = min(k0 + (f >> 4), o.rows - 1);
        const int n = n0 + ((f & 15) << 2);
        const float p = o.P[k * o.ldp + (n >> 7)];
        v = ld4(o.Q + k * o.ldq + (n & (HW - 1)));
        v.x *= p, v.y *= p, v.z *= p, v.w *= p;
      }
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  }
};

template <int MODE, bool AKC, bool BKC>
__global__ __launch_bounds__(256, 4) void head_gemm_kernel(const GemmArgs g, const HeadOps o) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, AKC, BKC>()];
  const int tn = (g.N + 63) >> 6, tm = (g.M + 63) >> 6;
  const int nwg = tn * tm * g.splits;
  const int b = xcd_remap(blockIdx.x, nwg);
  // the short side fastest: neighbouring workgroups (one XCD) share the weight panel they stream
  const int zs = b / (tn * tm), rr = b - zs * (tn * tm);
  const int bx = (tm < tn) ? rr / tm : rr % tn, by = (tm < tn) ? rr % tm : rr / tn;
  HeadOperands<MODE> ops;
  ops.o = o;
  gemm_body<1, 1, AKC, BKC, false, 4, 0, EPI_ALL, HeadOperands<MODE>>(g, lds, bx, by, zs, threadIdx.x, true, ops);
}

// ---- second generation of the three passes whose A operand is an outer product (MODE 1..3) -------------------------
// The operand never touches LDS at all: with K ordered (b-chunk of 32, then a), the MFMA A operand of lane (row i, k-half)
// is P[i, a] * Q[i, b]: the 16 Q values a lane needs for a whole b-chunk sit in registers (reloaded 4 times per tile),
// P[i, a] is one prefetched dword per k-step, and the operand is ONE v_mul per MFMA pair.  Only the weight panel is staged
// through LDS (k-major image, register double buffering as in gemm_body).  Workgroup tile 64 pairs x 128 columns, four
// waves of 32 x 64 (two accumulators: each generated A value feeds two MFMAs).
//   MODE 1: C[p, r]  = sum_(a,b) eh[p,a] et[p,b] W_b[r,a,b] + sum_k [eh | et][p,k] W_c[r,k] + bias    (P = eh, Q = et)
//   MODE 2: C[p, a]  = sum_(r,b) dout[p,r] et[p,b] W_b[r,a,b]                                         (P = doutp, Q = et)
//   MODE 3: C[p, b]  = sum_(r,a) dout[p,r] eh[p,a] W_b[r,a,b]                                         (P = doutp, Q = eh)
struct Bil2 {
  const float* P; const float* Q; const float* W; const float* W2; const float* bias;
  float* C;
  int rows, nA, ncol, ldc, accumulate_unused;
  // compacted pair rows (ragged batches, head_fwd / head_bwd): the number of rows lives on the device (workgroups past it exit
  // at once), and output row r goes to row crow[r] of C (the pair's slot in the padded [B, N, N, .] tensor); NULL = dense
  const int* rows_dev; const int* crow;
  int sel_lo, sel_hi;   // with rows_dev: this launch runs only when sel_lo <= *rows_dev < sel_hi (two variants are launched for a
                        // device-side count; the one whose range does not hold the count leaves at once)
  // K split over the four b-chunks (gridDim.y = 4; head_bil3_kernel): workgroup (tile, y) runs chunk y of the k sequence (y = 3
  // also the Linear tail of MODE 1) and stores its RAW partial sums to part[y][row][128]; head_bil_combine_kernel sums the
  // four in order, adds the bias and stores / scatters.  For row counts that leave most of the chip idle otherwise.
  float* part;
  int part_rows;        // rows of one partial slab
};

// position in the k-step sequence: for bc in 0..3: aa in 0..per_bc-1 (aa == nA: the Linear's et half, MODE 1), then tail steps
struct BilPos {
  int bc, aa;
  __device__ __forceinline__ void next(int per_bc) {
    if (++aa == per_bc && bc < 4) aa = 0, ++bc;   // bc == 4: the tail (aa keeps counting)
  }
};

template <int MODE>
__global__ __launch_bounds__(256, 3) void head_bil2_kernel(const Bil2 a) {
  constexpr int BN = 128, LDB = (MODE == 3) ? BN : BN + 1, SB = BK * LDB;
  __shared__ __attribute__((aligned(16))) float lds[2 * SB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int nrows = a.rows_dev ? *a.rows_dev : a.rows;
  if (m0 >= nrows || (a.rows_dev && (nrows < a.sel_lo || nrows >= a.sel_hi))) return;
  const long row = min(m0 + wr * 32 + l31, nrows - 1);   // this lane's A row (clamped: rows past the end are never stored)
  const float* __restrict__ prow = a.P + row * HW;
  const float* __restrict__ qrow = a.Q + row * HW;
  const int per_bc = a.nA + (MODE == 1 ? 1 : 0), nsteps = 4 * per_bc + (MODE == 1 ? 4 : 0);
  // per-thread offsets of its four 16-byte pieces of a weight tile (32-bit: the weights are < 2^31 floats), the step adds
  // a uniform offset
  unsigned offW[4], offW2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = t + 256 * q;
    if (MODE == 1) {
      const unsigned nn = (unsigned)min(f >> 3, a.ncol - 1);
      offW[q] = nn * (HW * HW) + ((f & 7) << 2), offW2[q] = nn * (2 * HW) + ((f & 7) << 2);
    } else if (MODE == 2) {
      offW[q] = (unsigned)(f >> 3) * HW + ((f & 7) << 2), offW2[q] = 0;
    } else {
      offW[q] = (unsigned)(f >> 5) * HW + ((f & 31) << 2), offW2[q] = 0;
    }
  }
  auto loadB = [&](float (&r)[4][4], const BilPos& p) {   // uniform: which array, which offset
    const float* base = a.W;
    unsigned so;
    bool second = false;
    if (MODE == 1) {
      if (p.bc >= 4) second = true, so = (unsigned)min(p.aa, 3) * BK;                 // W_c[r][32 j ..]        (tail, clamped)
      else if (p.aa == a.nA) second = true, so = HW + p.bc * BK;                      // W_c[r][128 + 32 bc ..]
      else so = (unsigned)p.aa * HW + p.bc * BK;                                      // W_b[r][a][32 bc ..]
      if (second) base = a.W2;
    } else if (MODE == 2) {
      so = (unsigned)min(p.aa, a.nA - 1) * (HW * HW) + min(p.bc, 3) * BK;             // W_b[r][a = n][32 bc ..]
    } else {
      so = ((unsigned)min(p.aa, a.nA - 1) * HW + min(p.bc, 3) * BK) * HW;             // W_b[r][a = 32 bc + k][b = n]
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = ld4(base + so + (second ? offW2[q] : offW[q]));
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  };
  auto storeB = [&](const float (&r)[4][4], float* __restrict__ dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = t + 256 * q;
      if (MODE == 3) {
        *reinterpret_cast<float4*>(dst + (f >> 5) * LDB + ((f & 31) << 2)) = make_float4(r[q][0], r[q][1], r[q][2], r[q][3]);
      } else {
        float* d = dst + ((f & 7) << 2) * LDB + (f >> 3);
        d[0] = r[q][0], d[LDB] = r[q][1], d[2 * LDB] = r[q][2], d[3 * LDB] = r[q][3];
      }
    }
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  float qv[16];
  auto loadQ = [&](int bc) {   // the lane's 16 Q values of this b-chunk: Q[row, 32 bc + 2 kk + lh]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = ld4(qrow + bc * BK + 4 * j);
      qv[2 * j] = lh ? v.y : v.x, qv[2 * j + 1] = lh ? v.w : v.z;
    }
  };
  auto pval = [&](const BilPos& p) -> float {  // the step's P factor, requested one step ahead
    if (p.bc >= 4) return 0.f;
    return (MODE == 1 && p.aa == a.nA) ? 1.f : prow[min(p.aa, a.nA - 1)];
  };
  const float* bl0 = lds + lh * LDB + wc * 64 + l31;
  auto compute = [&](const float* __restrict__ bl, const BilPos& p, float pa) {
    if (MODE == 1 && p.bc >= 4) {   // Linear, eh half: A = P[row, 32 j + k]
      const float* ph = prow + min(p.aa, 3) * BK;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float av = ph[2 * kk + lh];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 32], acc1, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float av = pa * qv[kk];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 32], acc1, 0, 0, 0);
    }
  };
  float rb0[4][4], rb1[4][4];
  BilPos pc = {0, 0}, pl = {0, 0};   // compute position, load position (two steps ahead)
  loadB(rb0, pl);
  storeB(rb0, lds);
  pl.next(per_bc);
  loadB(rb0, pl);
  pl.next(per_bc);
  BilPos pp = pc;
  float pa = pval(pp);
  pp.next(per_bc);
  float pn = pval(pp);
  loadQ(0);
  __syncthreads();
  for (int s = 0; s < nsteps; s += 2) {
    loadB(rb1, pl);
    pl.next(per_bc);
    if (pc.aa == 0 && pc.bc > 0 && pc.bc < 4) loadQ(pc.bc);
    compute(bl0, pc, pa);
    pc.next(per_bc), pp.next(per_bc);
    pa = pn, pn = pval(pp);
    storeB(rb0, lds + SB);
    __syncthreads();
    if (s + 1 >= nsteps) break;
    loadB(rb0, pl);
    pl.next(per_bc);
    if (pc.aa == 0 && pc.bc > 0 && pc.bc < 4) loadQ(pc.bc);
    compute(bl0 + SB, pc, pa);
    pc.next(per_bc), pp.next(per_bc);
    pa = pn, pn = pval(pp);
    storeB(rb1, lds);
    __syncthreads();
  }
  // epilogue: lane holds column wc*64 + {0, 32} + l31 of rows m0 + wr*32 + (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wc * 64 + j * 32 + l31;
    if (col >= a.ncol) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long rw = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (rw < nrows) a.C[(a.crow ? (long)a.crow[rw] : rw) * a.ldc + col] = (j ? acc1[r] : acc0[r]) + bias;
    }
  }
}

// MODE 1 with the relation count just past a multiple of 32 (R = 97 = 3 x 32 + 1): workgroup tile 128 pairs x 97 columns, a wave
// owns 32 pairs and ALL columns -- three accumulators for columns 0-95 (each generated A value feeds three MFMAs) and column 96
// on the vector ALU: the A values are in registers anyway, the column's weights are one broadcast LDS read per k, so the
// 97th relation costs one v_fma per generated value instead of a fourth 32-column MFMA block (a quarter of the pass).
// MODE 2 / 3 (128 exact columns): the same 128-pair tile with four accumulators per wave -- each generated value feeds four
// MFMAs instead of two.
template <int MODE>
__global__ __launch_bounds__(256, 2) void head_bil3_kernel(const Bil2 a) {
  constexpr int NACC = (MODE == 1) ? 3 : 4;
  constexpr int BN = 128, LDB = (MODE == 3) ? BN : BN + 1, SB = BK * LDB;
  __shared__ __attribute__((aligned(16))) float lds[2 * SB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 128;
  const int nrows = a.rows_dev ? *a.rows_dev : a.rows;
  if (m0 >= nrows || (a.rows_dev && (nrows < a.sel_lo || nrows >= a.sel_hi))) return;
  const long row = min(m0 + wave * 32 + l31, nrows - 1);   // this lane's A row (clamped: rows past the end are never stored)
  const float* __restrict__ prow = a.P + row * HW;
  const float* __restrict__ qrow = a.Q + row * HW;
  const int per_bc = a.nA + (MODE == 1 ? 1 : 0);
  // the k sequence: four b-chunks of per_bc steps, then (MODE 1) four tail steps.  Unsplit: all of it; split (a.part): chunk
  // blockIdx.y, the last one with the tail
  const int ysp = a.part ? (int)blockIdx.y : 0;
  const int nsteps = a.part ? per_bc + ((MODE == 1 && ysp == 3) ? 4 : 0) : 4 * per_bc + (MODE == 1 ? 4 : 0);
  // per-thread offsets of its four 16-byte pieces of a weight tile (32-bit: the weights are < 2^31 floats), the step adds
  // a uniform offset
  unsigned offW[4], offW2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = t + 256 * q;
    if (MODE == 1) {
      const unsigned nn = (unsigned)min(f >> 3, a.ncol - 1);
      offW[q] = nn * (HW * HW) + ((f & 7) << 2), offW2[q] = nn * (2 * HW) + ((f & 7) << 2);
    } else if (MODE == 2) {
      offW[q] = (unsigned)(f >> 3) * HW + ((f & 7) << 2), offW2[q] = 0;
    } else {
      offW[q] = (unsigned)(f >> 5) * HW + ((f & 31) << 2), offW2[q] = 0;
    }
  }
  auto loadB = [&](float (&r)[4][4], const BilPos& p) {   // uniform: which array, which offset
    const float* base = a.W;
    unsigned so;
    bool second = false;
    if (MODE == 1) {
      if (p.bc >= 4) second = true, so = (unsigned)min(p.aa, 3) * BK;                 // W_c[r][32 j ..]        (tail, clamped)
      else if (p.aa == a.nA) second = true, so = HW + p.bc * BK;                      // W_c[r][128 + 32 bc ..]
      else so = (unsigned)p.aa * HW + p.bc * BK;                                      // W_b[r][a][32 bc ..]
      if (second) base = a.W2;
    } else if (MODE == 2) {
      so = (unsigned)min(p.aa, a.nA - 1) * (HW * HW) + min(p.bc, 3) * BK;             // W_b[r][a = n][32 bc ..]
    } else {
      so = ((unsigned)min(p.aa, a.nA - 1) * HW + min(p.bc, 3) * BK) * HW;             // W_b[r][a = 32 bc + k][b = n]
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = ld4(base + so + (second ? offW2[q] : offW[q]));
      r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
    }
  };
  auto storeB = [&](const float (&r)[4][4], float* __restrict__ dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = t + 256 * q;
      if (MODE == 3) {
        *reinterpret_cast<float4*>(dst + (f >> 5) * LDB + ((f & 31) << 2)) = make_float4(r[q][0], r[q][1], r[q][2], r[q][3]);
      } else {
        float* d = dst + ((f & 7) << 2) * LDB + (f >> 3);
        d[0] = r[q][0], d[LDB] = r[q][1], d[2 * LDB] = r[q][2], d[3 * LDB] = r[q][3];
      }
    }
  };
  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f, acc2[r] = 0.f, acc3[r] = 0.f;
  float a96 = 0.f;   // column 96: this lane's k half of its row's dot product
  float qv[16];
  auto loadQ = [&](int bc) {   // the lane's 16 Q values of this b-chunk: Q[row, 32 bc + 2 kk + lh]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = ld4(qrow + bc * BK + 4 * j);
      qv[2 * j] = lh ? v.y : v.x, qv[2 * j + 1] = lh ? v.w : v.z;
    }
  };
  auto pval = [&](const BilPos& p) -> float {  // the step's P factor, requested one step ahead
    if (p.bc >= 4) return 0.f;
    return (MODE == 1 && p.aa == a.nA) ? 1.f : prow[min(p.aa, a.nA - 1)];
  };
  const float* bl0 = lds + lh * LDB + l31;
  auto compute = [&](const float* __restrict__ bl, const BilPos& p, float pa) {
    if (MODE == 1 && p.bc >= 4) {   // Linear, eh half: A = P[row, 32 j + k]
      const float* ph = prow + min(p.aa, 3) * BK;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float av = ph[2 * kk + lh];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 32], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 64], acc2, 0, 0, 0);
        a96 = fmaf(av, bl[2 * kk * LDB + 96 - l31], a96);   // (MODE 1 only reaches this branch)
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float av = pa * qv[kk];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 32], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 64], acc2, 0, 0, 0);
      if (NACC == 4) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bl[2 * kk * LDB + 96], acc3, 0, 0, 0);
      else a96 = fmaf(av, bl[2 * kk * LDB + 96 - l31], a96);
    }
  };
  float rb0[4][4], rb1[4][4];
  BilPos pc = {ysp, 0}, pl = {ysp, 0};   // compute position, load position (two steps ahead)
  loadB(rb0, pl);
  storeB(rb0, lds);
  pl.next(per_bc);
  loadB(rb0, pl);
  pl.next(per_bc);
  BilPos pp = pc;
  float pa = pval(pp);
  pp.next(per_bc);
  float pn = pval(pp);
  loadQ(ysp);
  __syncthreads();
  for (int s = 0; s < nsteps; s += 2) {
    loadB(rb1, pl);
    pl.next(per_bc);
    if (pc.aa == 0 && pc.bc > ysp && pc.bc < 4) loadQ(pc.bc);
    compute(bl0, pc, pa);
    pc.next(per_bc), pp.next(per_bc);
    pa = pn, pn = pval(pp);
    storeB(rb0, lds + SB);
    __syncthreads();
    if (s + 1 >= nsteps) break;
    loadB(rb0, pl);
    pl.next(per_bc);
    if (pc.aa == 0 && pc.bc > ysp && pc.bc < 4) loadQ(pc.bc);
    compute(bl0 + SB, pc, pa);
    pc.next(per_bc), pp.next(per_bc);
    pa = pn, pn = pval(pp);
    storeB(rb1, lds);
    __syncthreads();
  }
  // epilogue: lane holds columns {0, 32, 64} + l31 of rows m0 + wave*32 + (r & 3) + 8 (r >> 2) + 4 lh, and its k half of column 96
  a96 += __shfl_xor(a96, 32);
  if (a.part) {   // K split: raw partial sums of this chunk, [y][row][128]; the combine kernel adds the bias and stores
    float* __restrict__ wp = a.part + (long)ysp * a.part_rows * 128;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const int col = j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long rw = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (rw < nrows) wp[rw * 128 + col] = (j == 0 ? acc0[r] : j == 1 ? acc1[r] : j == 2 ? acc2[r] : acc3[r]);
      }
    }
    if (MODE == 1 && lh == 0) {
      const long rw = m0 + wave * 32 + l31;
      if (rw < nrows) wp[rw * 128 + 96] = a96;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int col = j * 32 + l31;
    if (col >= a.ncol) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long rw = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (rw < nrows) a.C[(a.crow ? (long)a.crow[rw] : rw) * a.ldc + col] = (j == 0 ? acc0[r] : j == 1 ? acc1[r] : j == 2 ? acc2[r] : acc3[r]) + bias;
    }
  }
  if (MODE == 1 && a.ncol > 96 && lh == 0) {
    const long rw = m0 + wave * 32 + l31;
    if (rw < nrows) a.C[(a.crow ? (long)a.crow[rw] : rw) * a.ldc + 96] = a96 + (a.bias ? a.bias[96] : 0.f);
  }
}

// ---- d W_b[r, (a, b)] = sum_p dout[p, r] eh[p, a] et[p, b]  for R just past a multiple of 32 (R = 97) ---------------------
// The GEMM form (M = R rows, N = 16384, K = pairs) in 64 x 64 tiles pads R to 128 rows; here a workgroup owns ALL R rows of 128
// columns (one a, every b), a wave 32 of those columns: three accumulators for rows 0-95, row 96 on the vector ALU (the generated
// B value eh[p, a] et[p, b] is in a register, dout[p, 96] one broadcast LDS read).  The dout rows of a k-step (32 pairs x 128
// floats, coalesced) go through LDS as the A image; et rows are read straight into the B registers (two 128-byte rows per wave
// instruction), eh[p, a] through 32 words of LDS.  K is split over gridDim.y workgroups; the partials are summed in split
// order by the GEMM's reduce kernel.
struct HeadDw {
  const float* doutp; const float* EH; const float* ET;
  float* out;      // [splits][R][16384] partials, or dW_b itself when splits == 1
  long pairs; int R, ksteps_per_split;
  const int* pairs_dev;   // compacted pair rows: K lives on the device (NULL = pairs); the split's share is derived from it
};
__global__ __launch_bounds__(256, 3) void head_dw_kernel(const HeadDw a) {
  constexpr int LA = 132;                        // A image [32 k][132]: dout[p, 0..127]
  __shared__ __attribute__((aligned(16))) float lds[2 * (32 * LA + 32)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int ca = blockIdx.x, sp = blockIdx.y;    // column tile = the index a; K split
  const int bcol = wave * 32 + l31;              // this lane's b
  const long npairs = a.pairs_dev ? (long)*a.pairs_dev : a.pairs;
  const int kps = a.pairs_dev ? (int)((((npairs + 31) >> 5) + gridDim.y - 1) / gridDim.y) : a.ksteps_per_split;
  const long k_begin = (long)sp * kps * 32;
  const long k_end = min(npairs, k_begin + (long)kps * 32);
  const int nk = (int)((k_end - k_begin + 31) / 32);
  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f, acc2[r] = 0.f;
  float a96 = 0.f;
  // staging: thread t loads 4 x 16 bytes of the 32 x 128 dout block (row t >> 3 + ..., coalesced) and, wave 0, eh[p, a]
  float4 ra[4];
  float re = 0.f, rq[16];
  auto gload = [&](const int ks) {
    const long p0 = k_begin + (long)ks * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 256 * u, row = idx >> 5, c4 = idx & 31;
      const long p = min(p0 + row, npairs - 1);
      float4 v = ld4(a.doutp + p * HW + c4 * 4);
      if (p0 + row >= k_end) v = make_float4(0.f, 0.f, 0.f, 0.f);   // past the split's end: contributes nothing
      ra[u] = v;
    }
    if (t < 32) re = a.EH[min(p0 + t, npairs - 1) * HW + ca];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) rq[kk] = a.ET[min(p0 + 2 * kk + lh, npairs - 1) * HW + bcol];
  };
  auto sstore = [&](float* __restrict__ st) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = t + 256 * u;
      *reinterpret_cast<float4*>(st + (idx >> 5) * LA + (idx & 31) * 4) = ra[u];
    }
    if (t < 32) st[32 * LA + t] = re;
  };
  float qv[16];
  auto compute = [&](const float* __restrict__ st) {
    const float* as = st + lh * LA + l31;
    const float* es = st + 32 * LA + lh;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float bv = es[2 * kk] * qv[kk];                    // eh[p, a] et[p, b], p = k0 + 2 kk + lh
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * kk * LA], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * kk * LA + 32], bv, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * kk * LA + 64], bv, acc2, 0, 0, 0);
      a96 = fmaf(as[2 * kk * LA + 96 - l31], bv, a96);
    }
  };
  if (nk > 0) {
    gload(0);
    sstore(lds);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) qv[kk] = rq[kk];
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      float* cur = lds + (ks & 1) * (32 * LA + 32);
      float* nxt = lds + ((ks + 1) & 1) * (32 * LA + 32);
      if (ks + 1 < nk) gload(ks + 1);
      compute(cur);
      if (ks + 1 < nk) {
        sstore(nxt);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) qv[kk] = rq[kk];
      }
      __syncthreads();
    }
  }
  a96 += __shfl_xor(a96, 32);
  float* __restrict__ o = a.out + (long)sp * a.R * (HW * HW) + (long)ca * HW + bcol;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < a.R) o[(long)row * (HW * HW)] = (j == 0 ? acc0[r] : j == 1 ? acc1[r] : acc2[r]);
    }
  if (a.R > 96 && lh == 0) o[96L * (HW * HW)] = a96;
}

// Which generation runs the outer-product passes.  The register-generated one (head_bil2 / head_bil3: 128-pair tiles, two
// workgroups per compute unit) wins once there is a tile per compute unit (B = 32: N = 64 15.8 vs 19.2 ms per step, N = 42 8.1 vs
// 8.3); below that the first one's finer tiles (64 x 64, four per compute unit) fill the chip better.  GCGCN_HEAD_V1=1 / =0 (or
// gcgcn_set_option("head_v1", 1 / 0; -1 = by size) forces one of them (A/B runs and the test suite, which runs both).
static bool head_v1(long pairs) {
  const int v = option("head_v1", -1);
  if (v >= 0) return v != 0;
  return cdiv(pairs, 128) < 256;
}

// the forward pass's 128-pair tile with column 96 on the vector ALU (GCGCN_HEAD_BIL3=0 / set_option("head_bil3", 0): A/B knob)
static bool head_bil3_ok(int ncol) { return option("head_bil3", 1) != 0 && ncol > 64 && ncol <= 97; }

// out[row, col] = sum_y part[y][row][col] + bias[col] for the K-split passes above (y in order: deterministic), row < *rows_dev,
// scattered through crow where the output is the padded logits tensor.
__global__ __launch_bounds__(256) void head_bil_combine_kernel(const float* __restrict__ part, int part_rows, const int* __restrict__ rows_dev,
                                                               int sel_lo, int sel_hi, const float* __restrict__ bias, const int* __restrict__ crow,
                                                               float* __restrict__ C, int ncol, int ldc) {
  const int nrows = *rows_dev;
  if (nrows < sel_lo || nrows >= sel_hi) return;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = e >> 5;
  const int c4 = (int)(e & 31) * 4;
  if (row >= nrows) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const float4 v = ld4(part + ((long)y * part_rows + row) * 128 + c4);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  float* o = C + (crow ? (long)crow[row] : row) * ldc;
  const float vs[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (c4 + u < ncol) o[c4 + u] = vs[u] + (bias ? bias[c4 + u] : 0.f);
}

constexpr int HEAD_TILE128_MIN_ROWS = 32768;   // one 128-pair tile per compute unit (the rule head_v1 applies to host-side counts)
static int head_bil2(int mode, const float* P, const float* Q, const float* W, const float* W2, const float* bias, float* C, long rows,
                     int nA, int ncol, int ldc, hipStream_t st, const int* rows_dev = nullptr, const int* crow = nullptr,
                     float* part = nullptr, long part_rows = 0) {
  Bil2 a;
  a.P = P, a.Q = Q, a.W = W, a.W2 = W2, a.bias = bias, a.C = C, a.rows = (int)rows, a.nA = nA, a.ncol = ncol, a.ldc = ldc, a.accumulate_unused = 0;
  a.rows_dev = rows_dev, a.crow = crow;   // (rows = the capacity the grid is sized for)
  a.sel_lo = 0, a.sel_hi = 0x7fffffff;
  a.part = nullptr, a.part_rows = 0;
  const dim3 grid((unsigned)cdiv(rows, 64)), grid128((unsigned)cdiv(rows, 128)), block(256);
  const double flops = 2.0 * rows * ncol * (double)(nA * HW + (mode == 1 ? 2 * HW : 0));
  const bool wide = option("head_bil3_bwd", 1) != 0;
  const bool tile128 = (mode == 1 && head_bil3_ok(ncol)) || (mode != 1 && wide && ncol == 128);
  auto launch128 = [&](const Bil2& b, const double flops) {   // 128 pairs per workgroup: 3 MFMA column blocks + column 96 on the vector ALU (mode 1), four accumulators (2 / 3)
    if (mode == 1) GC_LAUNCH_TIMED("head_bilinear", flops, head_bil3_kernel<1>, grid128, block, 0, st, b);
    else if (mode == 2) GC_LAUNCH_TIMED("head_bilinear", flops, head_bil3_kernel<2>, grid128, block, 0, st, b);
    else GC_LAUNCH_TIMED("head_bilinear", flops, head_bil3_kernel<3>, grid128, block, 0, st, b);
  };
  auto launch64 = [&](const Bil2& b, const double flops) {
    if (mode == 1) GC_LAUNCH_TIMED("head_bilinear", flops, (head_bil2_kernel<1>), grid, block, 0, st, b);
    else if (mode == 2) GC_LAUNCH_TIMED("head_bilinear", flops, (head_bil2_kernel<2>), grid, block, 0, st, b);
    else GC_LAUNCH_TIMED("head_bilinear", flops, (head_bil2_kernel<3>), grid, block, 0, st, b);
  };
  if (rows_dev && tile128) {
    // The count is on the device and the better launch shape depends on it; the host launches both, the one whose range does
    // not hold the count leaves at once (a few microseconds of a millisecond-sized pass).  One 128-pair tile per compute unit
    // and more (>= 32 768 pairs): the plain tiles.  Fewer (a DocRED batch of 32 documents: 14 k real pairs = 109 tiles on 256
    // compute units, each streaming the whole 6.4 MB weight through its LDS): every tile split four ways over the b-chunks of
    // the k sequence -- four times the workgroups, a quarter of the weight each -- plus a combine pass; without a partial-sum
    // workspace, 64-pair tiles (4.14 -> 3.39 ms per step; the split: see DESIGN.md section 9).
    Bil2 big = a, small = a;
    big.sel_lo = HEAD_TILE128_MIN_ROWS, small.sel_hi = HEAD_TILE128_MIN_ROWS;
    launch128(big, 0.5 * flops);    // (the host-side work counter: one of the two runs, the timer sees both)
    if (int e = check_launch("head_bil3")) return e;
    if (part && part_rows > 0) {
      small.part = part, small.part_rows = (int)part_rows;
      const long prow = rows < part_rows ? rows : part_rows;     // (counts below the threshold never exceed the slab)
      const dim3 grid4((unsigned)cdiv(prow, 128), 4);
      if (mode == 1) GC_LAUNCH_TIMED("head_bilinear", 0.5 * flops, head_bil3_kernel<1>, grid4, block, 0, st, small);
      else if (mode == 2) GC_LAUNCH_TIMED("head_bilinear", 0.5 * flops, head_bil3_kernel<2>, grid4, block, 0, st, small);
      else GC_LAUNCH_TIMED("head_bilinear", 0.5 * flops, head_bil3_kernel<3>, grid4, block, 0, st, small);
      if (int e = check_launch("head_bil3 split")) return e;
      hipLaunchKernelGGL(head_bil_combine_kernel, dim3((unsigned)cdiv(prow * 32, 256)), dim3(256), 0, st, part, (int)part_rows, rows_dev,
                         small.sel_lo, small.sel_hi, bias, crow, C, ncol, ldc);
      return check_launch("head_bil_combine");
    }
    launch64(small, 0.5 * flops);
    return check_launch("head_bil2");
  }
  if (tile128) {
    launch128(a, flops);
    return check_launch("head_bil3");
  }
  launch64(a, flops);
  return check_launch("head_bil2");
}

static int head_gemm(int mode, GemmArgs g, const HeadOps& o, hipStream_t st) {
  const long tiles = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
  int splits = 1;
  if (g.ws && tiles < 2048) {  // long-K weight gradient with few output tiles
    const long iters = cdiv(g.K, BK);
    while (splits < 8 && tiles * splits < 2048 && iters / (splits * 2) >= 64 && (long)(splits * 2) * g.M * g.N <= g.ws_elems) splits *= 2;
  }
  g.splits = splits;
  g.ksplit = splits > 1 ? (int)(((cdiv(g.K, BK) + splits - 1) / splits) * BK) : g.K;
  g.vecA = g.vecB = 1;
  g.batch1 = g.batch2 = 1;
  const dim3 grid((unsigned)(tiles * splits)), block(256);
  const double flops = 2.0 * g.M * g.N * g.K;
  switch (mode) {
    case 1: GC_LAUNCH_TIMED("head_bilinear", flops, (head_gemm_kernel<1, true, true>), grid, block, 0, st, g, o); break;
    case 2: GC_LAUNCH_TIMED("head_bilinear", flops, (head_gemm_kernel<2, true, true>), grid, block, 0, st, g, o); break;
    case 3: GC_LAUNCH_TIMED("head_bilinear", flops, (head_gemm_kernel<3, true, false>), grid, block, 0, st, g, o); break;
    default: GC_LAUNCH_TIMED("head_bilinear", flops, (head_gemm_kernel<4, false, false>), grid, block, 0, st, g, o); break;
  }
  if (int e = check_launch("head_gemm")) return e;
  return splits > 1 ? splitk_reduce(g, st) : 0;
}

// ---- per-entity term + type table:  UT[b, n, :] = U[b, n, :] + Tt[type[b, n], :];  bsum = b_bilinear + b_linear -------
__global__ __launch_bounds__(256) void head_node_kernel(const float* __restrict__ U, const float* __restrict__ Tt,
                                                        const long long* __restrict__ type, float* __restrict__ UT, long BN,
                                                        const float* __restrict__ bb, const float* __restrict__ bc,
                                                        float* __restrict__ bsum, int R) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0)
    for (int r = threadIdx.x; r < R; r += 256) bsum[r] = bb[r] + bc[r];
  if (e >= BN * HW) return;
  const long n = e / HW;
  const int c = (int)(e - n * HW);
  const int ty = min(max((int)type[n], 0), 6);
  UT[e] = U[e] + Tt[ty * HW + c];
}

// ---- compacted pair rows for ragged batches ---------------------------------------------------------------------------
// The reference runs the head on ONE unpadded document (glove:344-358: n x n pairs).  A padded batch has B N^2 pair slots of
// which sum_b n_b^2 exist (a DocRED batch padded to 42 entities averages 20: a quarter); the bilinear passes -- 2 x 128 x 128 x
// 97 flops per pair and pass -- run on the existing pairs only: row q = off[b] + i n_b + j of EH / ET / dout / dEH / dET is pair
// (b, i, j), off = exclusive prefix sums of n_b^2.  idx = [count | - | off[0..B] | prow[capacity]], prow[q] = the pair's slot
// b N^2 + i N + j in the padded tensors.  Everything stays on the device (no host read: capturable); the kernels below size
// their grids for the capacity and leave at once past the count.
__global__ __launch_bounds__(256) void head_index_kernel(const int* __restrict__ n_valid, int B, int N, int* __restrict__ idx) {
  __shared__ int part[256];
  // exclusive scan of n_b^2 over B documents: thread t owns documents [t chunk, (t + 1) chunk)
  const int t = threadIdx.x, chunk = (B + 255) / 256;
  int s = 0;
  for (int b = t * chunk; b < min(B, (t + 1) * chunk); ++b) {
    const int nv = min(max(n_valid[b], 0), N);
    s += nv * nv;
  }
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < 256; ++k) {
      const int v = part[k];
      part[k] = run;
      run += v;
    }
    idx[0] = run, idx[1] = 0;
  }
  __syncthreads();
  int run = part[t];
  for (int b = t * chunk; b < min(B, (t + 1) * chunk); ++b) {
    idx[2 + b] = run;
    const int nv = min(max(n_valid[b], 0), N);
    run += nv * nv;
    if (b == B - 1) idx[2 + B] = run;
  }
}
__global__ __launch_bounds__(256) void head_prow_kernel(const int* __restrict__ n_valid, const int* __restrict__ off,
                                                        int* __restrict__ prow, long pairs, int N) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const long b = p / ((long)N * N);
  const int ij = (int)(p - b * N * N), i = ij / N, j = ij - i * N, nv = min(max(n_valid[b], 0), N);
  if (i < nv && j < nv) prow[off[b] + i * nv + j] = (int)p;
}

// ---- eh[p] = tanh(UT[b, j] + Rt[dis_plus + rel[p]]),  et[p] = tanh(UT[b, i] + Rt[dis_plus - rel[p]])   (one wave per pair)
// cnt / prow != NULL: compacted rows -- row q holds pair prow[q]; rows [count, roundup128(count)) are zero (the tile kernels
// clamp their reads to the last row, the K-side products read up to the next multiple of 64)
__global__ __launch_bounds__(256) void head_feat_fwd_kernel(const float* __restrict__ UT, const float* __restrict__ Rt,
                                                            const long long* __restrict__ rel, float* __restrict__ EH,
                                                            float* __restrict__ ET, long pairs, int N, int dis_plus, int ND,
                                                            const int* __restrict__ cnt, const int* __restrict__ prow) {
  const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  long p = q;
  if (cnt) {
    const long n = *cnt;
    if (q >= ((n + 127) & ~127L)) return;
    if (q >= n) {
      *reinterpret_cast<float2*>(EH + q * HW + 2 * lane) = make_float2(0.f, 0.f);
      *reinterpret_cast<float2*>(ET + q * HW + 2 * lane) = make_float2(0.f, 0.f);
      return;
    }
    p = prow[q];
  } else if (q >= pairs) {
    return;
  }
  const long b = p / ((long)N * N);
  const int ij = (int)(p - b * N * N), i = ij / N, j = ij - i * N;
  const int d = (int)rel[p];
  const int kh = min(max(dis_plus + d, 0), ND - 1), kt = min(max(dis_plus - d, 0), ND - 1);
  const float2 uj = *reinterpret_cast<const float2*>(UT + (b * N + j) * HW + 2 * lane);
  const float2 ui = *reinterpret_cast<const float2*>(UT + (b * N + i) * HW + 2 * lane);
  const float2 rh = *reinterpret_cast<const float2*>(Rt + (long)kh * HW + 2 * lane);
  const float2 rt = *reinterpret_cast<const float2*>(Rt + (long)kt * HW + 2 * lane);
  *reinterpret_cast<float2*>(EH + q * HW + 2 * lane) = make_float2(tanhf(uj.x + rh.x), tanhf(uj.y + rh.y));
  *reinterpret_cast<float2*>(ET + q * HW + 2 * lane) = make_float2(tanhf(ui.x + rt.x), tanhf(ui.y + rt.y));
}

// ---- doutp[p, 0..127] = dout[p, 0..R) (zero padding), pairs of padding entities zeroed ---------------------------------
// cnt / prow != NULL: compacted rows -- doutp[q] = dout[prow[q]] for q < count, zero up to the capacity (`pairs` rows)
__global__ __launch_bounds__(256) void head_pad_kernel(const float* __restrict__ dout, const int* __restrict__ n_valid,
                                                       float* __restrict__ doutp, long pairs, int N, int R,
                                                       const int* __restrict__ cnt, const int* __restrict__ prow) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= pairs * HW) return;
  const long p = e / HW;
  const int c = (int)(e - p * HW);
  float v = 0.f;
  if (cnt) {
    if (c < R && p < *cnt) v = dout[(long)prow[p] * R + c];
  } else if (c < R) {
    bool ok = true;
    if (n_valid) {
      const long b = p / ((long)N * N);
      const int ij = (int)(p - b * N * N), i = ij / N, j = ij - i * N, nv = n_valid[b];
      ok = i < nv && j < nv;
    }
    if (ok) v = dout[p * R + c];
  }
  doutp[e] = v;
}

// ---- through the tanh, in place: dEH <- dEH (1 - eh^2), dET <- dET (1 - et^2) -----------------------------------------
__global__ __launch_bounds__(256) void head_tanh_bwd_kernel(const float* __restrict__ EH, const float* __restrict__ ET,
                                                            float* __restrict__ dEH, float* __restrict__ dET, long n4,
                                                            const int* __restrict__ cnt) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4 || (cnt && e >= (long)*cnt * (HW / 4))) return;
  const float4 h = ld4(EH + 4 * e), t = ld4(ET + 4 * e);
  float4 a = ld4(dEH + 4 * e), b = ld4(dET + 4 * e);
  a.x *= 1.f - h.x * h.x, a.y *= 1.f - h.y * h.y, a.z *= 1.f - h.z * h.z, a.w *= 1.f - h.w * h.w;
  b.x *= 1.f - t.x * t.x, b.y *= 1.f - t.y * t.y, b.z *= 1.f - t.z * t.z, b.w *= 1.f - t.w * t.w;
  *reinterpret_cast<float4*>(dEH + 4 * e) = a;
  *reinterpret_cast<float4*>(dET + 4 * e) = b;
}

// ---- dUT[b, n, :] = sum_i dEH[b, i, n, :] + sum_j dET[b, n, j, :]   (one workgroup per entity, rows in order) ----------
// off != NULL: compacted rows -- pair (b, i, j) is row off[b] + i n_b + j; padding entities get zero
__global__ __launch_bounds__(256) void head_node_bwd_kernel(const float* __restrict__ dEH, const float* __restrict__ dET,
                                                            float* __restrict__ dUT, int N, const int* __restrict__ off,
                                                            const int* __restrict__ n_valid) {
  __shared__ float red[2][HW];
  const long bn = blockIdx.x;
  const long b = bn / N;
  const int n = (int)(bn - b * N);
  const int c = threadIdx.x & (HW - 1), half = threadIdx.x >> 7;   // 2 x 128 threads: half 0 sums the head side, half 1 the tail side
  // one side's len rows of this entity, `step` pair rows apart, in order; eight requests in flight at a time
  auto side_sum = [&](const float* __restrict__ X, const long first, const long step, const int len) {
    float s = 0.f;
    for (int i0 = 0; i0 < len; i0 += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = X[(first + (long)min(i0 + u, len - 1) * step) * HW + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (i0 + u < len) ? x[u] : 0.f;
    }
    return s;
  };
  float a = 0.f;
  if (off) {
    const int nv = min(max(n_valid[b], 0), N);
    const long base = off[b];
    if (n < nv) a = half == 0 ? side_sum(dEH, base + n, nv, nv) : side_sum(dET, base + (long)n * nv, 1, nv);
  } else {
    const long base = b * N * N;
    a = half == 0 ? side_sum(dEH, base + n, N, N) : side_sum(dET, base + (long)n * N, 1, N);
  }
  red[half][c] = a;
  __syncthreads();
  if (half == 0) dUT[bn * HW + c] = red[0][c] + red[1][c];
}

// ---- partial gradient of a gathered table: part[b, k, :] = sum over the items of document b whose id == k --------------
//   rel mode : items = pairs, two contributions per pair (dEH at dis_plus + rel, dET at dis_plus - rel); grid (B, ND)
//   type mode: items = entities (X2 == nullptr), ids = type; grid (B, 7)
//   off / prow != NULL (rel mode): compacted rows -- the document's items are rows [off[b], off[b + 1]), their ids at prow[row]
__global__ __launch_bounds__(256) void head_table_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ X1,
                                                             const float* __restrict__ X2, float* __restrict__ part, long per_doc,
                                                             int dis_plus, int nk, const int* __restrict__ off,
                                                             const int* __restrict__ prow) {
  __shared__ float red[2][HW];
  const int b = blockIdx.x, k = blockIdx.y;
  const int c = threadIdx.x & (HW - 1), half = threadIdx.x >> 7;
  const float* X = half ? X2 : X1;
  float a = 0.f;
  if (X && off) {
    for (long q = off[b]; q < off[b + 1]; ++q) {
      const int d = (int)ids[prow[q]];
      const int id = half ? dis_plus - d : dis_plus + d;
      if (min(max(id, 0), nk - 1) == k) a += X[q * HW + c];
    }
  } else if (X) {
    const long base = (long)b * per_doc;
    for (long q = 0; q < per_doc; ++q) {
      const int d = (int)ids[base + q];
      const int id = X2 ? (half ? dis_plus - d : dis_plus + d) : d;
      if (min(max(id, 0), nk - 1) == k) a += X[(base + q) * HW + c];
    }
  }
  red[half][c] = a;
  __syncthreads();
  if (half == 0) part[((long)b * nk + k) * HW + c] = red[0][c] + red[1][c];
}
// The relative-position table's partial gradient in ONE pass over a document's pairs (the kernel above scans them once per table
// row: 21 workgroups read the same ids to keep 1 / 21 of the rows each).  Workgroup (b, s) takes slice s of document b's items in
// order; a thread owns one column of one side and adds into its own cell of an LDS table [side][id][column] -- no two threads
// share a cell, the order inside a slice is the items' order, the slices meet in head_table_fin_kernel in fixed order: deterministic.
constexpr int HT_SLICES = 8, HT_IDS = 32;
__global__ __launch_bounds__(256) void head_table_rel_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ X1,
                                                                 const float* __restrict__ X2, float* __restrict__ part, long per_doc,
                                                                 int dis_plus, int nk, const int* __restrict__ off,
                                                                 const int* __restrict__ prow) {
  __shared__ float acc[2][HT_IDS][HW];
  const int b = blockIdx.x, sl = blockIdx.y;
  const int c = threadIdx.x & (HW - 1), half = threadIdx.x >> 7;
  const float* __restrict__ X = half ? X2 : X1;
  for (int k = 0; k < nk; ++k) acc[half][k][c] = 0.f;
  const long lo = off ? off[b] : (long)b * per_doc, hi = off ? off[b + 1] : lo + per_doc, len = hi - lo;
  const long q0 = lo + len * sl / HT_SLICES, q1 = lo + len * (sl + 1) / HT_SLICES;
  long q = q0;
  for (; q + 4 <= q1; q += 4) {   // four items requested together; the adds stay in item order
    int id[4];
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = (int)ids[prow ? prow[q + u] : q + u];
      id[u] = min(max(half ? dis_plus - d : dis_plus + d, 0), nk - 1);
      x[u] = X[(q + u) * HW + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[half][id[u]][c] += x[u];
  }
  for (; q < q1; ++q) {
    const int d = (int)ids[prow ? prow[q] : q];
    acc[half][min(max(half ? dis_plus - d : dis_plus + d, 0), nk - 1)][c] += X[q * HW + c];
  }
  __syncthreads();
  for (int k = half; k < nk; k += 2) part[(((long)b * HT_SLICES + sl) * nk + k) * HW + c] = acc[0][k][c] + acc[1][k][c];
}
__global__ __launch_bounds__(256) void head_table_fin_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  // the slices in order, 16 requests in flight at a time (one at a time, the loop was B dependent round trips: 256 slices of the
  // relative-position table took 61 us on 11 workgroups)
  float s = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(long)min(b0 + u, B - 1) * n + e];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (b0 + u < B) ? v[u] : 0.f;
  }
  out[e] = s;
}

// The dense layer's products with the two embedding tables (7 entity types x Pt, ND distance buckets x Pr; glove:241-242, 354):
// a few thousand outputs of 7 .. 128 terms each.  As GEMM launches they were one 5-10 us launch per product (2 forward, 4 +
// a memset backward); here one launch each way, a role per blockIdx.y, plain fp32 dot products in k order.
//   forward   role 0: Tt[r, c] = sum_p ner[r, p] Wt[c, p]          role 1: Rt[r, c] = sum_p dis[r, p] Wr[c, p]
//   backward  role 0: dWt[c, p] = sum_r dTt[r, c] ner[r, p]         role 1: dner[r, p] = sum_c dTt[r, c] Wt[c, p]  (row 0 = padding_idx: 0)
//             role 2: dWr[c, p] = sum_r dRt[r, c] dis[r, p]         role 3: ddis[r, p] = sum_c dRt[r, c] Wr[c, p]
// Wt / Wr: columns [nf Hd, nf Hd + Pt) and [nf Hd + Pt, Fin) of the dense weight W[HW][Fin] (row stride Fin).
__global__ __launch_bounds__(256) void head_tables_fwd_kernel(const float* __restrict__ ner, const float* __restrict__ dis,
                                                              const float* __restrict__ Wt, const float* __restrict__ Wr, long Fin,
                                                              float* __restrict__ Tt, float* __restrict__ Rt, int Pt, int Pr, int ND) {
  const int role = blockIdx.y;
  const float* __restrict__ tab = role ? dis : ner;
  const float* __restrict__ W = role ? Wr : Wt;
  float* __restrict__ out = role ? Rt : Tt;
  const int rows = role ? ND : 7, P = role ? Pr : Pt;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * HW; e += gridDim.x * 256) {
    const int r = e / HW, c = e - r * HW;
    float s = 0.f;
    for (int p_ = 0; p_ < P; ++p_) s = fmaf(tab[r * P + p_], W[(long)c * Fin + p_], s);
    out[e] = s;
  }
}
__global__ __launch_bounds__(256) void head_tables_bwd_kernel(const float* __restrict__ dTt, const float* __restrict__ dRt,
                                                              const float* __restrict__ ner, const float* __restrict__ dis,
                                                              const float* __restrict__ Wt, const float* __restrict__ Wr, long Fin,
                                                              float* __restrict__ dWt, float* __restrict__ dWr, float* __restrict__ dner,
                                                              float* __restrict__ ddis, int Pt, int Pr, int ND) {
  const int role = blockIdx.y, side = role >> 1;          // side 0: the type table, 1: the distance table
  const float* __restrict__ dT = side ? dRt : dTt;
  const float* __restrict__ tab = side ? dis : ner;
  const float* __restrict__ W = side ? Wr : Wt;
  const int rows = side ? ND : 7, P = side ? Pr : Pt;
  if ((role & 1) == 0) {   // weight gradient [HW][P] (row stride Fin)
    float* __restrict__ dW = side ? dWr : dWt;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < HW * P; e += gridDim.x * 256) {
      const int c = e / P, p_ = e - c * P;
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s = fmaf(dT[r * HW + c], tab[r * P + p_], s);
      dW[(long)c * Fin + p_] = s;
    }
  } else {                 // table gradient [rows][P]
    float* __restrict__ dtab = side ? ddis : dner;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * P; e += gridDim.x * 256) {
      const int r = e / P, p_ = e - r * P;
      float s = 0.f;
      if (side || r > 0)   // ner_emb = nn.Embedding(7, 20, padding_idx=0) (glove:241): the padding row never receives a gradient
        for (int c = 0; c < HW; ++c) s = fmaf(dT[r * HW + c], W[(long)c * Fin + p_], s);
      dtab[e] = s;
    }
  }
}

// =====================================================================================================================
struct HeadLayout { long Wd, bd, Wc, bc, bb, Wb, total; int Fin; };
// flat = [dense_layer W [128, Fin] | b | classification_layer_01 W [R, 256] | b | bili b [R] | bili W [R, 128, 128]]
// (the bilinear weight last: the padded linear products read up to 31 rows past W_c, which must stay inside the buffer)
static HeadLayout head_layout(int Hd, int nf, int Pt, int Pr, int R) {
  HeadLayout y;
  y.Fin = Hd * nf + Pt + Pr;
  long o = 0;
  auto take = [&](long n) { const long at = o; o += (n + 3) & ~3L; return at; };
  y.Wd = take((long)HW * y.Fin), y.bd = take(HW), y.Wc = take((long)R * 2 * HW), y.bc = take(R), y.bb = take(R);
  y.Wb = take((long)R * HW * HW);
  y.total = o;
  return y;
}

struct HeadBufs {
  float *U, *Tt, *Rt, *UT, *bsum, *EH, *ET;                                  // forward (EH / ET saved for backward)
  float *doutp, *dEH, *dET, *dUT, *partR, *partT, *dRt, *dTt, *dW, *scratch;  // backward
  long scratch_elems;
  float *partF, *partB;   // [4][part_rows][128] partial sums of the K-split bilinear passes (small compacted row counts)
  long part_rows;
  int* idx;   // compacted pair rows of a ragged batch: [count | - | off[0..B] | prow[B N^2]] (head_index_kernel); NULL = dense
};
static long head_idx_ints(int B, int N) { return ((2 + (long)B + 1 + 3) & ~3L) + (long)B * N * N; }
static const int* idx_cnt(const HeadBufs& w) { return w.idx; }
static const int* idx_off(const HeadBufs& w) { return w.idx ? w.idx + 2 : nullptr; }
static int* idx_prow(const HeadBufs& w, int B) { return w.idx ? w.idx + ((2 + (long)B + 1 + 3) & ~3L) : nullptr; }

// option head_compact (default 1): ragged batches run the pair passes on the pairs that exist (compacted rows).  Served by the
// register-generated kernels with the relation count just past 96 (the reference's 97; head_bil3 / head_dw); anything else keeps
// the dense path, which computes every pair slot of the padded batch.
static bool head_compacts(const int* n_valid, int R) {
  return n_valid && option("head_compact", 1) != 0 && R > 64 && R <= 97;
}

static int small_gemm(const float* A, long lda, int a_kc, const float* B, long ldb, int b_kc, float* C, long ldc, int M, int N, int K,
                      const float* bias, int accumulate, float* ws, long wse, hipStream_t st) {
  GemmArgs g;
  g.A = A, g.lda = lda, g.a_kc = a_kc, g.B = B, g.ldb = ldb, g.b_kc = b_kc, g.C = C, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K, g.bias = bias, g.accumulate = accumulate, g.ws = ws, g.ws_elems = wse;
  g.tag = "head_gemm";
  return gemm(g, st);
}

int head_fwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats, const long long* type,
             const long long* rel, const float* ner_emb, const float* dis_table, const int* n_valid, const float* flat, HeadBufs w,
             float* logits, hipStream_t st) {
  const HeadLayout y = head_layout(Hd, nf, Pt, Pr, R);
  const long BN = (long)B * N, pairs = BN * N;
  GC_REQUIRE(pairs < (1L << 31) / 2 && R >= 1 && R <= HW, "head: %ld pairs / %d relations out of range", pairs, R);
  const bool compact = head_compacts(n_valid, R);
  GC_REQUIRE(!compact || w.idx, "head_fwd: n_valid given without the index workspace");
  for (int k = 0; k < nf; ++k)  // U = sum_k feats_k W_k^T + b           (glove:354-355, the entity part of the dense layer)
    GC_TRY(small_gemm(feats[k], Hd, 1, flat + y.Wd + (long)k * Hd, y.Fin, 1, w.U, HW, (int)BN, HW, Hd, k == 0 ? flat + y.bd : nullptr,
                      k > 0, nullptr, 0, st));
  {
    ProfScope ps("head_gemm", st);
    hipLaunchKernelGGL(head_tables_fwd_kernel, dim3(cdiv((long)(ND > 7 ? ND : 7) * HW, 256), 2), dim3(256), 0, st, ner_emb, dis_table,
                       flat + y.Wd + (long)nf * Hd, flat + y.Wd + (long)nf * Hd + Pt, (long)y.Fin, w.Tt, w.Rt, Pt, Pr, ND);
    GC_TRY(check_launch("head_tables_fwd"));
  }
  {
    ProfScope ps("head_feat", st);
    hipLaunchKernelGGL(head_node_kernel, dim3(cdiv(BN * HW, 256)), dim3(256), 0, st, w.U, w.Tt, type, w.UT, BN, flat + y.bb, flat + y.bc,
                       w.bsum, R);
    GC_TRY(check_launch("head_node"));
    if (compact) {   // which pairs exist, and where their rows go
      hipLaunchKernelGGL(head_index_kernel, dim3(1), dim3(256), 0, st, n_valid, B, N, w.idx);
      GC_TRY(check_launch("head_index"));
      hipLaunchKernelGGL(head_prow_kernel, dim3(cdiv(pairs, 256)), dim3(256), 0, st, n_valid, idx_off(w), idx_prow(w, B), pairs, N);
      GC_TRY(check_launch("head_prow"));
    }
    const long frows = compact ? ((pairs + 127) & ~127L) : pairs;
    hipLaunchKernelGGL(head_feat_fwd_kernel, dim3(cdiv(frows, 4)), dim3(256), 0, st, w.UT, w.Rt, rel, w.EH, w.ET, pairs, N, dis_plus, ND,
                       compact ? idx_cnt(w) : nullptr, compact ? idx_prow(w, B) : nullptr);
    GC_TRY(check_launch("head_feat_fwd"));
  }
  if (compact) {   // the pairs that exist, scattered into the padded tensor; every other slot is zero
    GC_REQUIRE(hipMemsetAsync(logits, 0, sizeof(float) * pairs * R, st) == hipSuccess, "head: memset failed");
    return head_bil2(1, w.EH, w.ET, flat + y.Wb, flat + y.Wc, w.bsum, logits, pairs, HW, R, R, st, idx_cnt(w), idx_prow(w, B), w.partF,
                     w.part_rows);
  }
  GemmArgs g;   // logits = [eh (x) et | eh | et] [W_b ; W_c]^T + (b_b + b_c)                     (glove:358)
  g.A = w.EH, g.B = flat + y.Wb, g.C = logits, g.ldc = R;
  g.M = (int)pairs, g.N = R, g.K = HW * HW + 2 * HW;
  g.bias = w.bsum;
  HeadOps o;
  memset(&o, 0, sizeof(o));
  o.P = w.EH, o.Q = w.ET, o.ldp = o.ldq = HW, o.KB = HW * HW, o.W2 = flat + y.Wc, o.ldw2 = 2 * HW, o.nmax = R, o.rows = (int)pairs;
  if (head_bil3_ok(R) || !head_v1(pairs)) return head_bil2(1, w.EH, w.ET, flat + y.Wb, flat + y.Wc, w.bsum, logits, pairs, HW, R, R, st);
  return head_gemm(1, g, o, st);
}

int head_bwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats, const long long* type,
             const long long* rel, const float* ner_emb, const float* dis_table, const int* n_valid, const float* flat, HeadBufs w,
             const float* dlogits, float* const* dfeats, float* dner_emb, float* ddis_table, float* dflat, hipStream_t st) {
  const HeadLayout y = head_layout(Hd, nf, Pt, Pr, R);
  const long BN = (long)B * N, pairs = BN * N;
  float* ws = w.scratch;
  const long wse = w.scratch_elems;
  // ragged batch: the rows of EH / ET (written by head_fwd), dout, dEH, dET are the pairs that exist, compacted (head_fwd built
  // the index); their count is on the device
  const bool compact = head_compacts(n_valid, R);
  GC_REQUIRE(!compact || w.idx, "head_bwd: n_valid given without the index workspace");
  const int* cnt = compact ? idx_cnt(w) : nullptr;
  const int* prow = compact ? idx_prow(w, B) : nullptr;
  {
    ProfScope ps("head_feat", st);
    const long prows = compact ? ((pairs + 127) & ~127L) : pairs;   // compacted: zero up to the last row any tile or k-step reads
    hipLaunchKernelGGL(head_pad_kernel, dim3(cdiv(prows * HW, 256)), dim3(256), 0, st, dlogits, n_valid, w.doutp, prows, N, R, cnt, prow);
    GC_TRY(check_launch("head_pad"));
  }
  // (products over the pair rows: M or K = the device-side count on the compacted path)
  auto rows_gemm = [&](const float* A, long lda, int a_kc, const float* Bm, long ldb, int b_kc, float* C, long ldc, int M, int Nn, int K,
                       int accumulate, int dyn) -> int {
    if (!compact) return small_gemm(A, lda, a_kc, Bm, ldb, b_kc, C, ldc, M, Nn, K, nullptr, accumulate, dyn == 2 ? ws : nullptr, dyn == 2 ? wse : 0, st);
    GemmArgs g;
    g.A = A, g.lda = lda, g.a_kc = a_kc, g.B = Bm, g.ldb = ldb, g.b_kc = b_kc, g.C = C, g.ldc = ldc;
    g.M = M, g.N = Nn, g.K = K, g.accumulate = accumulate, g.ws = ws, g.ws_elems = wse;
    g.tag = "head_gemm";
    return gemm_dyn(g, cnt, dyn, pairs, st);
  };
  // bias gradients: both biases see the column sums of dlogits
  GC_TRY(colsum(w.doutp, nullptr, dflat + y.bb, pairs, R, HW, 1, 0, 0, 0, 0, ws, st));
  GC_REQUIRE(hipMemcpyAsync(dflat + y.bc, dflat + y.bb, sizeof(float) * R, hipMemcpyDeviceToDevice, st) == hipSuccess, "head: copy failed");
  HeadOps o;
  memset(&o, 0, sizeof(o));
  o.ldp = o.ldq = HW, o.KB = R * HW, o.nmax = R, o.rows = (int)pairs;
  {  // d eh = sum_(r,b) dout[p,r] et[p,b] W_b[r,a,b]  + dout W_c[:, :128]
    GemmArgs g;
    g.A = w.doutp, g.B = flat + y.Wb, g.C = w.dEH, g.ldc = HW, g.M = (int)pairs, g.N = HW, g.K = R * HW;
    o.P = w.doutp, o.Q = w.ET;
    if (compact || !head_v1(pairs))
      GC_TRY(head_bil2(2, w.doutp, w.ET, flat + y.Wb, nullptr, nullptr, w.dEH, pairs, R, HW, HW, st, cnt, nullptr, compact ? w.partB : nullptr, w.part_rows));
    else GC_TRY(head_gemm(2, g, o, st));
    if (!compact) GC_TRY(rows_gemm(w.doutp, HW, 1, flat + y.Wc, 2 * HW, 0, w.dEH, HW, (int)pairs, HW, HW, 1, 1));   // (compacted: with d et's, below)
  }
  {  // d et = sum_(r,a) dout[p,r] eh[p,a] W_b[r,a,b]  + dout W_c[:, 128:]
    GemmArgs g;
    g.A = w.doutp, g.B = flat + y.Wb, g.C = w.dET, g.ldc = HW, g.M = (int)pairs, g.N = HW, g.K = R * HW;
    o.P = w.doutp, o.Q = w.EH;
    if (compact || !head_v1(pairs))
      GC_TRY(head_bil2(3, w.doutp, w.EH, flat + y.Wb, nullptr, nullptr, w.dET, pairs, R, HW, HW, st, cnt, nullptr, compact ? w.partB : nullptr, w.part_rows));
    else GC_TRY(head_gemm(3, g, o, st));
    if (compact) {   // + dout W_c on both sides, over the device-side pair count: one launch for the two halves
      GemmArgs gh[2];
      for (int q = 0; q < 2; ++q) {
        GemmArgs& g2 = gh[q];
        g2.A = w.doutp, g2.lda = HW, g2.a_kc = 1, g2.B = flat + y.Wc + q * HW, g2.ldb = 2 * HW, g2.b_kc = 0;
        g2.C = q ? w.dET : w.dEH, g2.ldc = HW, g2.M = (int)pairs, g2.N = HW, g2.K = HW, g2.accumulate = 1;
        g2.ws = ws, g2.ws_elems = wse;
        g2.tag = "head_gemm";
      }
      GC_TRY(gemm_dyn_pair_xx(gh[0], gh[1], cnt, pairs, st));
    } else {
      GC_TRY(rows_gemm(w.doutp, HW, 1, flat + y.Wc + HW, 2 * HW, 0, w.dET, HW, (int)pairs, HW, HW, 1, 1));
    }
  }
  {  // d W_b[r, (a, b)] = sum_p dout[p, r] eh[p, a] et[p, b]
    GemmArgs g;
    g.A = w.doutp, g.lda = HW, g.B = w.EH, g.C = dflat + y.Wb, g.ldc = HW * HW, g.M = R, g.N = HW * HW, g.K = (int)pairs;
    g.ws = ws, g.ws_elems = wse;
    o.P = w.EH, o.Q = w.ET, o.KB = 0;
    const bool dw3 = option("head_dw3", 1) != 0;
    if (compact || (dw3 && R > 64 && R <= 97 && !head_v1(pairs))) {  // all R rows per workgroup, row 96 on the vector ALU (head_dw_kernel)
      const long ksteps = cdiv(pairs, 32);
      int splits = 1;
      while (splits < 16 && (long)HW * splits < 2048 && ksteps / (splits * 2) >= 64 && (long)(splits * 2) * R * HW * HW <= wse) splits *= 2;
      HeadDw d;
      d.doutp = w.doutp, d.EH = w.EH, d.ET = w.ET, d.pairs = pairs, d.R = R, d.ksteps_per_split = (int)cdiv(ksteps, splits);
      d.pairs_dev = cnt;   // compacted: K = the device-side count (the split factor stays the capacity's)
      d.out = splits > 1 ? ws : dflat + y.Wb;
      GC_LAUNCH_TIMED("head_bilinear", 2.0 * R * HW * HW * (double)pairs, head_dw_kernel, dim3(HW, splits), dim3(256), 0, st, d);
      GC_TRY(check_launch("head_dw"));
      if (splits > 1) {
        g.splits = splits, g.batch1 = g.batch2 = 1;
        GC_TRY(splitk_reduce(g, st));
      }
    } else {
      GC_TRY(head_gemm(4, g, o, st));
    }
  }
  // d W_c = dout^T [eh | et]    (computed 128 rows deep into a workspace, the R real rows copied out)
  if (compact) {   // the two halves over the same device-side pair count: one launch, one reduce
    GemmArgs gh[2];
    for (int q = 0; q < 2; ++q) {
      GemmArgs& g = gh[q];
      g.A = w.doutp, g.lda = HW, g.a_kc = 0, g.B = q ? w.ET : w.EH, g.ldb = HW, g.b_kc = 0;
      g.C = w.dW + q * HW, g.ldc = 2 * HW, g.M = HW, g.N = HW, g.K = (int)pairs;
      g.ws = ws, g.ws_elems = wse;
      g.tag = "head_gemm";
    }
    GC_TRY(gemm_dyn_pair_ww(gh[0], gh[1], cnt, pairs, st));
  } else {
    GC_TRY(rows_gemm(w.doutp, HW, 0, w.EH, HW, 0, w.dW, 2 * HW, HW, HW, (int)pairs, 0, 2));
    GC_TRY(rows_gemm(w.doutp, HW, 0, w.ET, HW, 0, w.dW + HW, 2 * HW, HW, HW, (int)pairs, 0, 2));
  }
  GC_REQUIRE(hipMemcpyAsync(dflat + y.Wc, w.dW, sizeof(float) * R * 2 * HW, hipMemcpyDeviceToDevice, st) == hipSuccess, "head: copy failed");
  {
    ProfScope ps("head_feat", st);
    hipLaunchKernelGGL(head_tanh_bwd_kernel, dim3(cdiv(pairs * HW / 4, 256)), dim3(256), 0, st, w.EH, w.ET, w.dEH, w.dET, pairs * HW / 4, cnt);
    GC_TRY(check_launch("head_tanh_bwd"));
    hipLaunchKernelGGL(head_node_bwd_kernel, dim3((unsigned)BN), dim3(256), 0, st, w.dEH, w.dET, w.dUT, N, compact ? idx_off(w) : nullptr,
                       n_valid);
    GC_TRY(check_launch("head_node_bwd"));
    hipLaunchKernelGGL(head_table_rel_bwd_kernel, dim3(B, HT_SLICES), dim3(256), 0, st, rel, w.dEH, w.dET, w.partR, (long)N * N, dis_plus,
                       ND, compact ? idx_off(w) : nullptr, prow);
    GC_TRY(check_launch("head_table_bwd/rel"));
    hipLaunchKernelGGL(head_table_fin_kernel, dim3(cdiv((long)ND * HW, 256)), dim3(256), 0, st, w.partR, w.dRt, B * HT_SLICES, ND * HW);
    GC_TRY(check_launch("head_table_fin/rel"));
    hipLaunchKernelGGL(head_table_bwd_kernel, dim3(B, 7), dim3(256), 0, st, type, w.dUT, (const float*)nullptr, w.partT, (long)N, 0, 7,
                       (const int*)nullptr, (const int*)nullptr);
    GC_TRY(check_launch("head_table_bwd/type"));
    hipLaunchKernelGGL(head_table_fin_kernel, dim3(cdiv(7L * HW, 256)), dim3(256), 0, st, w.partT, w.dTt, B, 7 * HW);
    GC_TRY(check_launch("head_table_fin/type"));
  }
  // dense layer: entity part, type part, relative-position part.  The entity part's 2 nf products (data and weight gradient of
  // every feature group) are independent of each other: ONE group launch (+ one reduce for the split weight gradients) with
  // the bias gradient's column sums riding in both, where round 4 issued 3 nf + 2 launches of 5-8 us each.
  if (nf >= 1 && 2 * nf <= 8 && ws && wse > (long)COL_RIDE_SLICES * HW + 4) {
    GemmArgs gs[8];
    const long col_elems = ((long)COL_RIDE_SLICES * HW + 3) & ~3L;    // the riding column sum's partials: the workspace's tail
    for (int k = 0; k < nf; ++k) {
      GemmArgs& gx = gs[2 * k];
      gx = GemmArgs();
      gx.A = w.dUT, gx.lda = HW, gx.a_kc = 1, gx.B = flat + y.Wd + (long)k * Hd, gx.ldb = y.Fin, gx.b_kc = 0;
      gx.C = dfeats[k], gx.ldc = Hd, gx.M = (int)BN, gx.N = Hd, gx.K = HW;
      GemmArgs& gw = gs[2 * k + 1];
      gw = GemmArgs();
      gw.A = w.dUT, gw.lda = HW, gw.a_kc = 0, gw.B = feats[k], gw.ldb = Hd, gw.b_kc = 0;
      gw.C = dflat + y.Wd + (long)k * Hd, gw.ldc = y.Fin, gw.M = HW, gw.N = Hd, gw.K = (int)BN;
      gx.ws = gw.ws = ws, gx.ws_elems = gw.ws_elems = wse - col_elems;
      gx.tag = gw.tag = "head_gemm";
    }
    ColRide cr;
    cr.X = w.dUT, cr.out = dflat + y.bd, cr.part = ws + (wse - col_elems), cr.R = BN, cr.ld = HW, cr.C = HW;
    GC_TRY(gemm_group(gs, 2 * nf, st, &cr));
  } else {
    for (int k = 0; k < nf; ++k) {
      GC_TRY(small_gemm(w.dUT, HW, 1, flat + y.Wd + (long)k * Hd, y.Fin, 0, dfeats[k], Hd, (int)BN, Hd, HW, nullptr, 0, ws, wse, st));
      GC_TRY(small_gemm(w.dUT, HW, 0, feats[k], Hd, 0, dflat + y.Wd + (long)k * Hd, y.Fin, HW, Hd, (int)BN, nullptr, 0, ws, wse, st));
    }
    GC_TRY(colsum(w.dUT, nullptr, dflat + y.bd, BN, HW, HW, 1, 0, 0, 0, 0, ws, st));
  }
  {
    ProfScope ps("head_gemm", st);
    const int widest = Pt > Pr ? Pt : Pr;
    hipLaunchKernelGGL(head_tables_bwd_kernel, dim3(cdiv((long)HW * widest, 256), 4), dim3(256), 0, st, w.dTt, w.dRt, ner_emb, dis_table,
                       flat + y.Wd + (long)nf * Hd, flat + y.Wd + (long)nf * Hd + Pt, (long)y.Fin, dflat + y.Wd + (long)nf * Hd,
                       dflat + y.Wd + (long)nf * Hd + Pt, dner_emb, ddis_table, Pt, Pr, ND);
    GC_TRY(check_launch("head_tables_bwd"));
  }
  return 0;
}

static HeadBufs head_bind(float* fwd, float* bwd, int B, int N, int R, int ND, long* n_fwd, long* n_bwd) {
  const long BN = (long)B * N, pairs = (BN * N + 127) & ~127L;   // pair rows: a whole number of 128-row tiles (compacted rows are
                                                                // zero-filled / read that far)
  HeadBufs w;
  memset(&w, 0, sizeof(w));
  float* base = fwd;
  long at = 0;
  auto tf = [&](long n) { float* p = base ? base + at : nullptr; at += (n + 3) & ~3L; return p; };
  w.U = tf(BN * HW), w.Tt = tf(7 * HW), w.Rt = tf((long)ND * HW), w.UT = tf(BN * HW), w.bsum = tf(HW), w.EH = tf(pairs * HW);
  w.ET = tf(pairs * HW);
  w.part_rows = pairs < HEAD_TILE128_MIN_ROWS ? pairs : HEAD_TILE128_MIN_ROWS;
  w.partF = tf(4 * w.part_rows * 128);
  if (n_fwd) *n_fwd = at;
  base = bwd, at = 0;
  w.doutp = tf(pairs * HW), w.dEH = tf(pairs * HW), w.dET = tf(pairs * HW), w.dUT = tf(BN * HW), w.partR = tf((long)B * HT_SLICES * ND * HW);
  w.partT = tf((long)B * 7 * HW), w.dRt = tf((long)ND * HW), w.dTt = tf(7 * HW), w.dW = tf((long)HW * 2 * HW);
  // split-K partials: up to 8 slabs of the [R, 16384] bilinear weight gradient
  w.scratch_elems = 8L * R * HW * HW;
  const long cs = colsum_scratch_elems(pairs, HW, 1);
  if (w.scratch_elems < cs) w.scratch_elems = cs;
  w.scratch = tf(w.scratch_elems);
  w.partB = tf(4 * w.part_rows * 128);
  if (n_bwd) *n_bwd = at;
  return w;
}

}  // namespace gc

#include "../../include/gcgcn.h"
using namespace gc;

extern "C" {

int gcgcn_head_layout(int Hd, int nf, int Pt, int Pr, int R, int64_t* o) {
  GC_REQUIRE(Hd > 0 && nf > 0 && Pt > 0 && Pr > 0 && R > 0 && R <= HW && o, "head_layout: bad arguments");
  const HeadLayout y = head_layout(Hd, nf, Pt, Pr, R);
  o[0] = y.Wd, o[1] = y.bd, o[2] = y.Wc, o[3] = y.bc, o[4] = y.bb, o[5] = y.Wb, o[6] = y.total;
  return 0;
}

int gcgcn_head_sizes(int B, int N, int R, int ND, int64_t* out3) {
  GC_REQUIRE(B > 0 && N > 0 && R > 0 && R <= HW && ND > 0 && out3, "head_sizes: bad arguments");
  long a = 0, b = 0;
  head_bind(nullptr, nullptr, B, N, R, ND, &a, &b);
  out3[0] = a, out3[1] = b, out3[2] = head_idx_ints(B, N);
  return 0;
}

int gcgcn_head_fwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats,
                   const int64_t* node_type, const int64_t* node_relative_pos, const float* ner_emb, const float* dis_table,
                   const int32_t* n_valid, const float* flat, float* fbuf, int32_t* ibuf, float* logits, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && Hd > 0 && nf > 0 && nf <= 8 && Pt > 0 && Pr > 0 && R > 0 && ND > 0, "head_fwd: bad shape");
  GC_REQUIRE(Hd % 4 == 0 && Pt % 4 == 0 && Pr % 4 == 0, "head_fwd: feature widths must be multiples of 4 (16-byte rows)");
  GC_REQUIRE(feats && node_type && node_relative_pos && ner_emb && dis_table && flat && fbuf && logits, "head_fwd: null pointer");
  GC_REQUIRE(!n_valid || ibuf, "head_fwd: n_valid given without ibuf");
  HeadBufs w = head_bind(fbuf, nullptr, B, N, R, ND, nullptr, nullptr);
  w.idx = ibuf;
  return head_fwd(B, N, Hd, nf, Pt, Pr, R, ND, dis_plus, feats, (const long long*)node_type, (const long long*)node_relative_pos, ner_emb,
                  dis_table, n_valid, flat, w, logits, (hipStream_t)stream);
}

int gcgcn_head_bwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats,
                   const int64_t* node_type, const int64_t* node_relative_pos, const float* ner_emb, const float* dis_table,
                   const int32_t* n_valid, const float* flat, float* fbuf, int32_t* ibuf, float* bbuf, const float* dlogits,
                   float* const* dfeats, float* dner_emb, float* ddis_table, float* dflat, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && Hd > 0 && nf > 0 && nf <= 8 && Pt > 0 && Pr > 0 && R > 0 && ND > 0 && ND <= HT_IDS, "head_bwd: bad shape");
  GC_REQUIRE(feats && node_type && node_relative_pos && ner_emb && dis_table && flat && fbuf && bbuf && dlogits && dfeats && dner_emb &&
                 ddis_table && dflat,
             "head_bwd: null pointer");
  GC_REQUIRE(!n_valid || ibuf, "head_bwd: n_valid given without ibuf");
  HeadBufs w = head_bind(fbuf, bbuf, B, N, R, ND, nullptr, nullptr);
  w.idx = ibuf;
  return head_bwd(B, N, Hd, nf, Pt, Pr, R, ND, dis_plus, feats, (const long long*)node_type, (const long long*)node_relative_pos, ner_emb,
                  dis_table, n_valid, flat, w, dlogits, dfeats, dner_emb, ddis_table, dflat, (hipStream_t)stream);
}

}  // extern "C"
