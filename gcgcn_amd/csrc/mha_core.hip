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
  mha_core_fwd_body(sm, blockIdx.x, Q, n_valid, P, A, N, D, H, dh, kchunk, alpha, drop, threadIdx.x, true, nullptr, 0, 4);
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
