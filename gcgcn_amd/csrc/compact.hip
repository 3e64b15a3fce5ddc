// Consumers of the edge-feature producer's COMPACT rows: the graph blocks of a hop without a dense E tensor.
//
// The reference builds context_sent_att[N, N, hidden] per hop (glove:314-327) and hands it to GATAttention /
// GraphConvolution (hop 0, glove:332-333) or to MultiGraphConvolution (hop >= 1, glove:337).  The producer kernels
// (producer.hip) only ever compute the pairs with a live sentence slot -- Ec[prow, :] for the Q live pairs of the batch, with
// prow = pair_prow[b, i, j] (-1 for a pair without a live slot) -- and every other real pair of E equals the bias vector of
// linear_sentence_att (its sentence sums are exactly zero).  So instead of expanding Ec into E[B, N, N, hidden], streaming E
// through the hop's kernels, writing dE[B, N, N, hidden] and gathering it back:
//
//   mean-only hop (every MAGGC hop):  Ebar[b, i] = (sum_{j live} Ec[prow(b,i,j)] + (n - live_i) bias) / n
//                                     dEc[prow]  = dEbar[b, i] / n,     dbias = sum_{b,i} (n - live_i) / n  dEbar[b, i]
//   attention hop (hop 0):            e_ij = Ec[prow] or bias in the edge pass of GATAttention (logit = v . e_ij, mean_j e_ij),
//                                     dEc[prow] = dlogit_ij v + dEbar_i / n,   dbias = v sum_dead dlogit + sum (n - live_i)/n dEbar_i,
//                                     dv = sum_ij dlogit_ij e_ij  (dead pairs included: e = bias)
//
// E and dE never exist.  One workgroup of 4 waves per entity row (b, i), like the dense edge kernels; every sum has a fixed
// order (bitwise reproducible).  Reference arithmetic: GATAttention.forward glove:154-168 (folded: energy = u.x_j + v.e_ij + c),
// GraphConv's edge term glove:40-41 (mean commuted with the projection).
#include "../../include/gcgcn.h"

#include "gat_body.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int CW4 = 4;       // waves per workgroup
constexpr int CMAXK = 8;     // columns per lane: D <= 64 * CMAXK

struct CmpE {
  const float* Ec;    // [Q, D] compact rows
  const int* prow;    // [B, N, N]: row of Ec, or -1
  const float* bias;  // [D]
};

// dynamic LDS: CW4 * D (+ N when ATT) floats
template <bool ATT>
__global__ __launch_bounds__(64 * CW4) void cmp_edge_fwd_kernel(const CmpE ce, const float* __restrict__ v, const int* __restrict__ n_valid,
                                                                float* __restrict__ Ebar, const float* __restrict__ coladd,
                                                                float* __restrict__ P, float* __restrict__ Aout, const Drop drop, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) float cs[];
  float* lg = cs + (long)CW4 * D;
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  float* eb = Ebar + (long)bi * D;
  if (i >= nv) {  // padding entity: outputs are zero
    for (int c = t; c < D; c += 64 * CW4) eb[c] = 0.f;
    if (ATT)
      for (int j = t; j < N; j += 64 * CW4) {
        P[(long)bi * N + j] = 0.f;
        if (Aout) Aout[(long)bi * N + j] = 0.f;
      }
    return;
  }
  const int* __restrict__ pr = ce.prow + (long)bi * N;
  float acc[CMAXK], vr[CMAXK];
#pragma unroll
  for (int k = 0; k < CMAXK; ++k) {
    acc[k] = 0.f;
    vr[k] = (ATT && lane + 64 * k < D) ? v[lane + 64 * k] : 0.f;
  }
  for (int j = wave; j < nv; j += CW4) {
    const int p = pr[j];
    const float* __restrict__ row = p >= 0 ? ce.Ec + (long)p * D : ce.bias;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CMAXK; ++k) {
      const int c = lane + 64 * k;
      if (c < D) {
        const float x = row[c];
        acc[k] += x;
        dot = fmaf(x, vr[k], dot);
      }
    }
    if (ATT) {
      dot = wave_sum(dot);
      if (lane == 0) lg[j] = dot;
    }
  }
#pragma unroll
  for (int k = 0; k < CMAXK; ++k)
    if (lane + 64 * k < D) cs[wave * D + lane + 64 * k] = acc[k];
  __syncthreads();
  const float inv = 1.f / (float)nv;
  for (int c = t; c < D; c += 64 * CW4) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < CW4; ++w) s += cs[w * D + c];
    eb[c] = s * inv;
  }
  if (ATT && wave == 0) {  // row softmax over the nv real columns (+ the node scores, + dropout): as edge_fwd_row
    const float* ca = coladd + (long)b * N;
    for (int j = lane; j < nv; j += 64) lg[j] += ca[j];
    float m = -INFINITY;
    for (int j = lane; j < nv; j += 64) m = fmaxf(m, lg[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < nv; j += 64) sum += expf(lg[j] - m);
    sum = wave_sum(sum);
    const float isum = 1.f / sum;
    const bool dd = Aout && drop.snap;
    const uint64_t key = dd ? drop_key(drop) : 0;
    for (int j = lane; j < N; j += 64) {
      float pv = 0.f;
      if (j < nv) pv = expf(lg[j] - m) * isum;
      const long o = (long)bi * N + j;
      P[o] = pv;
      if (Aout) {
        if (dd) pv = (rng_u32(key, (uint64_t)o) >= drop.thresh) ? pv * drop.scale : 0.f;
        Aout[o] = pv;
      }
    }
  }
}

// dEc[prow] = dlogit_ij v + dEbar_i / n for the live pairs of row (b, i); dvpart[bi] = sum_j dlogit_ij e_ij;
// sd[bi] = sum over dead pairs of dlogit_ij; cw[bi] = (dead pairs of the row) / n.  dlogit / dEbar / dvpart may be NULL
// (mean-only hop: no attention term).  dynamic LDS: CW4 * (D + 2) floats.
__global__ __launch_bounds__(64 * CW4) void cmp_edge_bwd_kernel(const CmpE ce, const float* __restrict__ v, const int* __restrict__ n_valid,
                                                                const float* __restrict__ dlogit, const float* __restrict__ dEbar,
                                                                float* __restrict__ dEc, float* __restrict__ dvpart,
                                                                float* __restrict__ sd, float* __restrict__ cw, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) float cs[];
  float* ws = cs + (long)CW4 * D;   // [CW4][2]: per-wave dead-pair sums
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  if (i >= nv) {
    if (dvpart)
      for (int c = t; c < D; c += 64 * CW4) dvpart[(long)bi * D + c] = 0.f;
    if (t == 0) sd[bi] = 0.f, cw[bi] = 0.f;
    return;
  }
  const int* __restrict__ pr = ce.prow + (long)bi * N;
  const float inv = 1.f / (float)nv;
  float acc[CMAXK], vr[CMAXK], g[CMAXK], br[CMAXK];
#pragma unroll
  for (int k = 0; k < CMAXK; ++k) {
    const int c = lane + 64 * k;
    acc[k] = 0.f;
    vr[k] = (dlogit && c < D) ? v[c] : 0.f;
    g[k] = (dEbar && c < D) ? dEbar[(long)bi * D + c] * inv : 0.f;
    br[k] = (dlogit && c < D) ? ce.bias[c] : 0.f;
  }
  float sdead = 0.f;
  int ndead = 0;
  for (int j = wave; j < nv; j += CW4) {
    const int p = pr[j];
    const float d = dlogit ? dlogit[(long)bi * N + j] : 0.f;
    if (p >= 0) {
#pragma unroll
      for (int k = 0; k < CMAXK; ++k) {
        const int c = lane + 64 * k;
        if (c < D) {
          if (dlogit) acc[k] = fmaf(d, ce.Ec[(long)p * D + c], acc[k]);
          dEc[(long)p * D + c] = fmaf(d, vr[k], g[k]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < CMAXK; ++k) acc[k] = fmaf(d, br[k], acc[k]);
      sdead += d;
      ++ndead;
    }
  }
  if (dvpart) {
#pragma unroll
    for (int k = 0; k < CMAXK; ++k)
      if (lane + 64 * k < D) cs[wave * D + lane + 64 * k] = acc[k];
  }
  if (lane == 0) ws[2 * wave] = sdead, ws[2 * wave + 1] = (float)ndead;
  __syncthreads();
  if (dvpart)
    for (int c = t; c < D; c += 64 * CW4) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < CW4; ++w) s += cs[w * D + c];
      dvpart[(long)bi * D + c] = s;
    }
  if (t == 0) {
    float s = 0.f, n = 0.f;
#pragma unroll
    for (int w = 0; w < CW4; ++w) s += ws[2 * w], n += ws[2 * w + 1];
    sd[bi] = s;
    cw[bi] = n * inv;
  }
}

// dbias[c] = v[c] sum_rows sd + sum_rows cw[row] dEbar[row, c]   (v / dEbar may be NULL): one workgroup per 64 columns, rows
// strided over 4 waves, fixed order
__global__ __launch_bounds__(256) void cmp_bias_grad_kernel(const float* __restrict__ sd, const float* __restrict__ cw,
                                                            const float* __restrict__ dEbar, const float* __restrict__ v,
                                                            float* __restrict__ dbias, long rows, int D) {
  __shared__ float red[4][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  float a = 0.f, s = 0.f;
  for (long r = wave; r < rows; r += 4) {
    if (dEbar && c < D) a = fmaf(cw[r], dEbar[r * D + c], a);
    s += sd[r];
  }
  red[wave][lane] = a;
  if (lane == 0) red[wave][64] = s;
  __syncthreads();
  if (wave == 0 && c < D) {
    const float at = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const float st = (red[0][64] + red[1][64]) + (red[2][64] + red[3][64]);
    dbias[c] = at + (v ? v[c] * st : 0.f);
  }
}

static int cmp_check(const char* who, int B, int N, int D, const float* Ec, const int* prow, const float* bias) {
  GC_REQUIRE(B > 0 && N > 0 && D > 0 && (long)B * N <= 0x7fffffffL, "%s: bad shape B=%d N=%d D=%d", who, B, N, D);
  GC_REQUIRE(D <= 64 * CMAXK, "%s: hidden width %d (compact rows support up to %d)", who, D, 64 * CMAXK);
  GC_REQUIRE(Ec && prow && bias, "%s: null pointer", who);
  return 0;
}

static int cmp_fwd(const CmpE& ce, const float* v, const int* n_valid, float* Ebar, const float* coladd, float* P, float* A, Drop drop,
                   int B, int N, int D, hipStream_t st) {
  const bool att = P != nullptr;
  const size_t lds = ((size_t)CW4 * D + (att ? N : 0)) * sizeof(float);
  GC_REQUIRE(lds <= 64 * 1024, "compact edge pass: N=%d D=%d needs %zu B of LDS", N, D, lds);
  ProfScope ps(att ? "edge_fwd_att" : "edge_fwd_mean", st, 0.0);
  if (att) hipLaunchKernelGGL(cmp_edge_fwd_kernel<true>, dim3((unsigned)((long)B * N)), dim3(64 * CW4), lds, st, ce, v, n_valid, Ebar, coladd, P, A, drop, N, D);
  else hipLaunchKernelGGL(cmp_edge_fwd_kernel<false>, dim3((unsigned)((long)B * N)), dim3(64 * CW4), lds, st, ce, v, n_valid, Ebar, coladd, P, A, drop, N, D);
  return check_launch("cmp_edge_fwd");
}

static int cmp_bwd(const CmpE& ce, const float* v, const int* n_valid, const float* dlogit, const float* dEbar, float* dEc, float* dvpart,
                   float* rowbuf, float* dbias, int B, int N, int D, hipStream_t st) {
  const long BN = (long)B * N;
  float *sd = rowbuf, *cw = rowbuf + BN;
  const size_t lds = ((size_t)CW4 * D + 2 * CW4) * sizeof(float);
  {
    ProfScope ps("edge_bwd", st, 0.0);
    hipLaunchKernelGGL(cmp_edge_bwd_kernel, dim3((unsigned)BN), dim3(64 * CW4), lds, st, ce, v, n_valid, dlogit, dEbar, dEc, dvpart, sd, cw, N, D);
    GC_TRY(check_launch("cmp_edge_bwd"));
  }
  ProfScope ps("colsum", st, 0.0);
  hipLaunchKernelGGL(cmp_bias_grad_kernel, dim3(cdiv(D, 64)), dim3(256), 0, st, sd, cw, dEbar, dlogit ? v : nullptr, dbias, BN, D);
  return check_launch("cmp_bias_grad");
}

}  // namespace gc

using namespace gc;

extern "C" {

int gcgcn_edge_mean_fwd_compact(int B, int N, int D, const float* Ec, const int32_t* prow, const float* bias, const int32_t* n_valid,
                                float* Ebar, void* stream) {
  GC_TRY(cmp_check("edge_mean_fwd_compact", B, N, D, Ec, prow, bias));
  GC_REQUIRE(Ebar, "edge_mean_fwd_compact: null pointer");
  const CmpE ce{Ec, prow, bias};
  return cmp_fwd(ce, nullptr, n_valid, Ebar, nullptr, nullptr, nullptr, make_drop(nullptr, 0, 0.f), B, N, D, (hipStream_t)stream);
}

int gcgcn_edge_mean_bwd_compact(int B, int N, int D, const int32_t* prow, const int32_t* n_valid, const float* dEbar, float* dEc,
                                float* dbias, float* rowbuf, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && D > 0 && D <= 64 * CMAXK, "edge_mean_bwd_compact: bad shape B=%d N=%d D=%d", B, N, D);
  GC_REQUIRE(prow && dEbar && dEc && dbias && rowbuf, "edge_mean_bwd_compact: null pointer");
  const CmpE ce{nullptr, prow, nullptr};
  return cmp_bwd(ce, nullptr, n_valid, nullptr, dEbar, dEc, nullptr, rowbuf, dbias, B, N, D, (hipStream_t)stream);
}

int gcgcn_gat_fwd_compact(int B, int N, int D, int Dh, const float* X, const float* Ec, const int32_t* prow, const float* bias,
                          const int32_t* n_valid, const float* flat, const void* rng_snap, float p, float* uvc, float* s, float* P,
                          float* A, float* Ebar, void* rng_state, void* rng_snaps, int rng_count, int uvc_valid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(cmp_check("gat_fwd_compact", B, N, D, Ec, prow, bias));
  GC_REQUIRE(Dh > 0 && X && flat && uvc && s && P && Ebar, "gat_fwd_compact: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GAT, p);
  GC_REQUIRE(!drop.snap || A, "gat_fwd_compact: dropout on but A is NULL");
  GC_REQUIRE(!rng_state || (rng_snaps && rng_count > 0), "gat_fwd_compact: rng_state given without snapshots to fill");
  const long M = (long)B * N;
  if (!uvc_valid) {
    GC_TRY(gat_fold_fwd(flat, uvc, D, Dh, st, rng_state, rng_snaps, rng_count));
    GC_TRY(node_score_fwd(X, uvc, s, M, D, st));
  } else {
    GC_TRY(node_score_fwd(X, uvc, s, M, D, st, rng_state, rng_snaps, rng_count));
  }
  const CmpE ce{Ec, prow, bias};
  return cmp_fwd(ce, uvc + D, n_valid, Ebar, s, P, A, drop, B, N, D, st);
}

int64_t gcgcn_gat_bwd_compact_scratch(int B, int N, int D) { return gcgcn_gat_bwd_scratch(B, N, D) + 2L * B * N; }

int gcgcn_gat_bwd_compact(int B, int N, int D, int Dh, const float* X, const float* Ec, const int32_t* prow, const float* bias,
                          const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const float* uvc, const float* P,
                          const float* dA, const float* dEbar, const float* dX_in, float* dX, float* dEc, float* dbias, float* dflat,
                          float* dlogit, float* ds, float* dvpart, float* duvc, float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_TRY(cmp_check("gat_bwd_compact", B, N, D, Ec, prow, bias));
  GC_REQUIRE(Dh > 0 && X && flat && uvc && P && dA && dX && dEc && dbias && dflat && dlogit && ds && dvpart && duvc && scratch,
             "gat_bwd_compact: null pointer");
  const Drop drop = make_drop(rng_snap, GCGCN_SALT_GAT, p);
  const long M = (long)B * N;
  if (gat_dlogit_ok(N)) {
    GC_TRY(gat_dlogit(P, dA, uvc, dX_in, dlogit, ds, dX, B, N, D, drop, st));
  } else {
    GC_TRY(softmax_bwd(P, dA, dlogit, M, N, drop, st));
    GC_TRY(colsum(dlogit, nullptr, ds, N, N, N, B, (long)N * N, 0, N, 0, nullptr, st));
    GC_TRY(node_score_bwd(ds, uvc, dX_in, dX, M, D, st));
  }
  const CmpE ce{Ec, prow, bias};
  float* rowbuf = scratch + gcgcn_gat_bwd_scratch(B, N, D);
  GC_TRY(cmp_bwd(ce, uvc + D, n_valid, dlogit, dEbar, dEc, dvpart, rowbuf, dbias, B, N, D, st));
  long part_off[3];
  int ns = 0;
  GC_TRY(colsum3(X, ds, duvc, M, D, D, dvpart, nullptr, duvc + D, M, D, D, ds, nullptr, duvc + 2 * D, M, 1, 1, scratch, st, false,
                 part_off, &ns));
  return gat_fold_bwd(flat, duvc, dflat, D, Dh, st, scratch, part_off, ns);
}

}  // extern "C"
