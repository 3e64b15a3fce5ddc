// The per-(document, head) products of the densely connected GraphConv stack, described once and
// used twice: by the host (one batched launch per product: gcgcn_gcn_fwd/_bwd with GCGCN_NO_CHAIN=1)
// and by the chain kernels (chain.hip: one persistent workgroup per (b, h) runs them back to back).
//
// Tensors are [B*N, H, L, gh] row-major (row stride HD = H*D, D = L*gh); A is [B, H, N, N];
// z1 = document, z2 = head.  Reference: GraphConv.forward glove:36-50 inside the dense loops of
// GraphConvolution.forward glove:70-76 / MultiGraphConvolution.forward glove:102-113.
#pragma once
#include "gemm.hpp"
#include "mha_body.hpp"

namespace gc {

// An edge-tensor streaming pass that depends on nothing the chain computes (the NEXT hop's mean_j E forward, its
// dE = dEbar / n backward).  A chain launch of few (doc, head) pairs leaves most compute units idle, and the chain
// is latency-bound where the pass is HBM-bound: the pass rides in extra workgroups of the same launch.
struct EdgeRide {
  int kind;  // 0 none, 1: out[B,N,D] = mean_j in[B,N,N,D], 2: out[B,N,N,D] = in[B,N,D] / n
  int B, N, D;
  const float* in;
  const int* n_valid;
  float* out;
};

struct GcnCtx {
  int B, N, D, L, H, gh;
  EdgeRide ride;
  // forward with MultiHeadAttention's core computed by the chain workgroup itself (LDS-resident kernels, gcgcn_mha_hook):
  // A_h = dropout(softmax(alpha Q_h Q_h^T)) goes straight into the adjacency image; P and A are still written for backward
  struct MhaFwd {
    const float* Q;   // NULL: the adjacency comes from c.A
    float* P;
    float* A;         // NULL without dropout (the chain then uses P)
    float alpha;
    Drop drop;
    int dh, kchunk;
  } mha;
  Spread carry;      // backward launches: how the tile passengers are placed among the riding rows (common.hpp)
  long HD, oWd, wd_head;
  const float* X;
  const float* A;
  const float* flat;
  const int* n_valid;
  Drop drop;
  // forward: G (edge term) in, Pn/Y/HO/rinv out
  const float* G;
  float* Pn;
  float* Y;
  float* HO;
  float* rinv;
  // backward
  float* dYa;   // running gradient of Y (starts as dropout_bwd(dHO))
  float* dM;    // gradient of G_l + A_h Pn_l
  float* dP;    // gradient of Pn_l
  float* dA;
  float* drow;  // gradient of the normaliser's row sum
  // backward with the output projection's input gradient computed by the chain itself (chain.hip, fused kernels):
  // dHO = dout Wlin per (document, head) instead of a launch of its own, dXres = sum_h dHO_h = dout (sum_h Wlin_h)
  const float* dout;   // [B*N, D]   gradient of the block's output; with dout_m set: as it arrives, the chain zeroes the
                       //            padding rows (n_valid) and undoes the output dropout (odrop) while it stages the rows
  float* dout_m;       // [B*N, D] or NULL: the masked gradient, written back for the products that need it later (dWlin)
  Drop odrop;
  const float* Wsum;   // [D, D]     sum over heads of Wlin's column blocks (H > 1)
  float* dXres;        // [B*N, D]
  float* colpart;      // [2 B, D] or NULL: column sums of dout over each half of a document's rows (bias gradient, stage 1)
  long oWlin;
  __host__ __device__ long wd_off(int l) const { return oWd + (long)gh * gh * l * (l - 1) / 2; }
};

__host__ __device__ inline void plan_common(const GcnCtx& c, GemmArgs& g) {
  g.batch1 = c.B, g.batch2 = c.H;
  g.splits = 1;
}

// Pn_l += [Y_0 .. Y_{l-1}] Wd_l          (dense connection, glove:73 / 110), l >= 1
__host__ __device__ inline GemmArgs plan_fwd_dense(const GcnCtx& c, int l) {
  GemmArgs g;
  plan_common(c, g);
  g.A = c.Y, g.lda = c.HD, g.a_kc = 1, g.sA1 = (long)c.N * c.HD, g.sA2 = (long)c.L * c.gh;
  g.B = c.flat + c.wd_off(l), g.ldb = c.gh, g.b_kc = 0, g.sB1 = 0, g.sB2 = c.wd_head;
  g.C = c.Pn + (long)l * c.gh, g.ldc = c.HD, g.sC1 = (long)c.N * c.HD, g.sC2 = (long)c.L * c.gh;
  g.M = c.N, g.N = c.gh, g.K = l * c.gh;
  g.accumulate = 1;
  g.tag = "gemm_dense";
  return g;
}

// Y_l = relu((G_l + A_h Pn_l) * rinv);  HO_l = dropout(Y_l) + X_l      (glove:42-50, 71-76)
__host__ __device__ inline GemmArgs plan_fwd_agg(const GcnCtx& c, int l) {
  GemmArgs g;
  plan_common(c, g);
  const long off = (long)l * c.gh, sb = (long)c.N * c.HD, sh = (long)c.L * c.gh;
  g.A = c.A, g.lda = c.N, g.a_kc = 1, g.sA1 = (long)c.H * c.N * c.N, g.sA2 = (long)c.N * c.N;
  g.B = c.Pn + off, g.ldb = c.HD, g.b_kc = 0, g.sB1 = sb, g.sB2 = sh;
  g.C = c.Y + off, g.ldc = c.HD, g.sC1 = sb, g.sC2 = sh;
  g.M = c.N, g.N = c.gh, g.K = c.N;
  g.add = c.G + off, g.ldadd = c.HD, g.sAdd1 = sb, g.sAdd2 = sh;
  g.rowscale = c.rinv, g.sRs1 = (long)c.H * c.N, g.sRs2 = c.N;
  g.relu = 1;
  g.C2 = c.HO + off, g.ldc2 = c.HD, g.sC21 = sb, g.sC22 = sh;
  g.add2 = c.X + off, g.ldadd2 = c.D, g.sAdd21 = (long)c.N * c.D, g.sAdd22 = 0;
  g.drop = c.drop, g.drop_base = off;  // dropout index = element offset inside HO
  g.tag = "gemm_agg";
  return g;
}

// dPn_l = A_h^T dM_l
__host__ __device__ inline GemmArgs plan_bwd_dP(const GcnCtx& c, int l) {
  GemmArgs g;
  plan_common(c, g);
  const long off = (long)l * c.gh, sb = (long)c.N * c.HD, sh = (long)c.L * c.gh;
  g.A = c.A, g.lda = c.N, g.a_kc = 0, g.sA1 = (long)c.H * c.N * c.N, g.sA2 = (long)c.N * c.N;
  g.B = c.dM + off, g.ldb = c.HD, g.b_kc = 0, g.sB1 = sb, g.sB2 = sh;
  g.C = c.dP + off, g.ldc = c.HD, g.sC1 = sb, g.sC2 = sh;
  g.M = c.N, g.N = c.gh, g.K = c.N;
  g.tag = "gemm_dP";
  return g;
}

// dA_h (+)= dM_l Pn_l^T ; the normaliser's gradient drow[i] is added to every column on the last pass (l == 0)
__host__ __device__ inline GemmArgs plan_bwd_dA(const GcnCtx& c, int l) {
  GemmArgs g;
  plan_common(c, g);
  const long off = (long)l * c.gh, sb = (long)c.N * c.HD, sh = (long)c.L * c.gh;
  g.A = c.dM + off, g.lda = c.HD, g.a_kc = 1, g.sA1 = sb, g.sA2 = sh;
  g.B = c.Pn + off, g.ldb = c.HD, g.b_kc = 1, g.sB1 = sb, g.sB2 = sh;
  g.C = c.dA, g.ldc = c.N, g.sC1 = (long)c.H * c.N * c.N, g.sC2 = (long)c.N * c.N;
  g.M = c.N, g.N = c.N, g.K = c.gh;
  g.accumulate = (l != c.L - 1);
  if (l == 0) g.rowadd = c.drow, g.sRa1 = (long)c.H * c.N, g.sRa2 = c.N;
  g.tag = "gemm_dA";
  return g;
}

// dY_{0..l-1} += dPn_l Wd_l^T, l >= 1
__host__ __device__ inline GemmArgs plan_bwd_dY(const GcnCtx& c, int l) {
  GemmArgs g;
  plan_common(c, g);
  const long sb = (long)c.N * c.HD, sh = (long)c.L * c.gh;
  g.A = c.dP + (long)l * c.gh, g.lda = c.HD, g.a_kc = 1, g.sA1 = sb, g.sA2 = sh;
  g.B = c.flat + c.wd_off(l), g.ldb = c.gh, g.b_kc = 1, g.sB1 = 0, g.sB2 = c.wd_head;
  g.C = c.dYa, g.ldc = c.HD, g.sC1 = sb, g.sC2 = sh;
  g.M = c.N, g.N = l * c.gh, g.K = c.gh;
  g.accumulate = 1;
  g.tag = "gemm_dY";
  return g;
}

// chain_t.hip (host side) + chain_t.hpp / chain_t_u0..3.hip (kernels): LDS-resident chain kernels for N <= 64 and the instantiated (gh, L) pairs
bool chain_t_ok(const GcnCtx& c, bool bwd);
bool chain_t_bwd_fusable(const GcnCtx& c);   // the column-strip backward computes dHO / dXres itself (c.dout) at this shape
bool chain_t_fwd_att_ok(const GcnCtx& c);
int gcn_chain_t_fwd(const GcnCtx& c, dim3 grid, double flops, hipStream_t st);
int gcn_chain_t_bwd(const GcnCtx& c, double flops, hipStream_t st, DeferQueue* carry);
// the full-size instantiation: every document fills all four 16-row blocks exactly (N == 64, no n_valid) -- one body, no
// switch, no bounds test; everything else runs the instantiation that picks a body per document (chain_t.hpp)
inline bool chain_t_full(const GcnCtx& c) { return c.N == 64 && !c.n_valid; }

// chain.hip
bool chain_fwd_computes_attention(const GcnCtx& c);   // the forward chain kernel of this shape takes c.mha
bool chain_can_carry(const EdgeRide& r);
Spread chain_carry_spread(const GcnCtx& c, int n_tile_workgroups);
int gcn_chain_fwd(const GcnCtx& c, hipStream_t st);
int gcn_chain_bwd(const GcnCtx& c, hipStream_t st, DeferQueue* carry = nullptr);
bool chain_bwd_fusable(const GcnCtx& c);   // c.dout / c.dXres / c.Wsum may be used instead of c.dYa

}  // namespace gc
