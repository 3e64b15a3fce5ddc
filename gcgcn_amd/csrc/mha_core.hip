// Attention core of MultiHeadAttention for small graphs (N <= 64 entities, the DocRED regime).
//
// Per (document b, head h) the reference computes  A = dropout(softmax(Q_h Q_h^T / sqrt(dh)))  (glove:136-140; the
// keys reuse the query projection).  With N <= 64 the whole N x N score matrix of one (b, h) fits in LDS, so one
// workgroup produces P and A straight from Q_h -- scores never touch HBM -- and, backward, turns dA into
// dQ_h = alpha (dS + dS^T) Q_h without materialising dS; both products run on the fp32 MFMA from LDS operands.  Replaces three launches forward (batched score GEMM at
// 64x64x32 per problem, softmax) and three backward (softmax gradient, two batched 64x32x64 GEMMs).
// Larger graphs take the generic GEMM + row-softmax path in api.hip.
#include "mha_body.hpp"
#include "rowops.hpp"

namespace gc {

__global__ __launch_bounds__(256) void mha_core_fwd_kernel(const float* __restrict__ Q, const int* __restrict__ n_valid,
                                                           float* __restrict__ P, float* __restrict__ A, int N, int D, int H,
                                                           int dh, int kchunk, float alpha, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* S = sm;             // [MT][MS]
  float* qs = sm + MT * MS;  // [MT][kchunk + 1]
  const int z = blockIdx.x, b = z / H, h = z - b * H;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const float* q = Q + (long)b * N * D + (long)h * dh;
  // scores: wave (wr, wc) owns the 32 x 32 quadrant S[32 wr .., 32 wc ..] = Q[32 wr ..] Q[32 wc ..]^T
  const int wr = wave >> 1, wc = wave & 1, ld = kchunk + 1;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < dh; k0 += kchunk) {
    const int kc = min(kchunk, dh - k0);
    if (k0) __syncthreads();
    load_q_chunk(qs, q, N, D, k0, kc, (kc + 3) & ~3, ld, t);
    __syncthreads();
    const float* pa = qs + (32 * wr + (lane & 31)) * ld + (lane >> 5);
    const float* pb = qs + (32 * wc + (lane & 31)) * ld + (lane >> 5);
    for (int k = 0; k < kc; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) S[(32 * wr + mfma_row(r, lane)) * MS + 32 * wc + (lane & 31)] = acc[r] * alpha;
  __syncthreads();
  // row softmax over the valid columns, dropout; lane = column
  const bool dd = A && drop.snap;
  const uint64_t key = dd ? drop_key(drop) : 0;
  for (int i = wave; i < N; i += 4) {
    const long r = (long)z * N + i;
    float v = 0.f;
    if (i < nv) {
      const float s = (lane < nv) ? S[i * MS + lane] : -INFINITY;
      const float m = wave_max(s);
      const float e = (lane < nv) ? expf(s - m) : 0.f;
      v = e / wave_sum(e);
    }
    if (lane < N) {
      P[r * N + lane] = v;
      if (A) {
        if (dd) v = (rng_u32(key, (uint64_t)(r * N + lane)) >= drop.thresh) ? v * drop.scale : 0.f;
        A[r * N + lane] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                           const float* __restrict__ dA, float* __restrict__ dQ, int N, int D,
                                                           int H, int dh, int kchunk, float alpha, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  mha_core_bwd_body(sm, blockIdx.x, Q, P, dA, dQ, N, D, H, dh, kchunk, alpha, drop);
}

bool mha_core_ok(int N, int D, int H, const void* Q, const void* dQ) {
  const int dh = D / H;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  return N >= 1 && N <= MT && dh % 4 == 0 && D % 4 == 0 && al(Q) && (!dQ || al(dQ));
}

int mha_core_fwd(const float* Q, const int* n_valid, float* P, float* A, int B, int N, int D, int H, float alpha, Drop drop,
                 hipStream_t st) {
  const int dh = D / H;
  ProfScope ps("mha_core_fwd", st);
  hipLaunchKernelGGL(mha_core_fwd_kernel, dim3(B * H), dim3(256), mha_lds_bytes(dh), st, Q, n_valid, P, A, N, D, H, dh,
                     mha_chunk(dh), alpha, drop);
  return check_launch("mha_core_fwd");
}

int mha_core_bwd(const float* Q, const float* P, const float* dA, float* dQ, int B, int N, int D, int H, float alpha, Drop drop,
                 hipStream_t st) {
  const int dh = D / H;
  ProfScope ps("mha_core_bwd", st);
  hipLaunchKernelGGL(mha_core_bwd_kernel, dim3(B * H), dim3(256), mha_lds_bytes(dh), st, Q, P, dA, dQ, N, D, H, dh,
                     mha_chunk(dh), alpha, drop);
  return check_launch("mha_core_bwd");
}

}  // namespace gc
