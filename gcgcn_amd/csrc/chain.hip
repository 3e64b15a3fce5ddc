// Chain kernels of the densely connected GraphConv stack.
//
// Inside one GraphConvolution / MultiGraphConvolution every (document b, head h) pair runs its own
// dependent sequence of small products (gcn_plan.hpp): per sub-layer  Pn_l += Y_<l Wd_l,  Y_l =
// relu((G_l + A_h Pn_l) rinv)  forward, and the mirrored four steps backward.  Launched one by
// one these are 2L + 4L latency-bound kernels of 64x64x64-sized work; here ONE persistent
// workgroup per (b, h) runs the whole sequence with the GEMM tile body, handing intermediate
// tiles to itself through global memory (L2-hot) across __syncthreads() -- all waves of a
// workgroup share the CU's vector L1, so no cache maintenance is needed at workgroup scope.
// Pairs never touch each other's slices, so there is no inter-workgroup synchronisation at all
// and every wave reaches the end of the phase list (no spin, no flag).
#include "gcn_plan.hpp"
#include "gemm_body.hpp"
#include "rowops.hpp"

namespace gc {

__device__ __forceinline__ bool dev_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Every 64x64 tile of batch entry z of product g, one after the other, by this workgroup.
template <bool AKC, bool BKC, bool ALIGNED>
__device__ __forceinline__ void run_product(GemmArgs g, float* lds, int z) {
  g.ksplit = g.K;
  g.splits = 1;
  if (!ALIGNED) {
    g.vecA = dev_al16(g.A) && g.lda % 4 == 0 && g.sA1 % 4 == 0 && g.sA2 % 4 == 0;
    g.vecB = dev_al16(g.B) && g.ldb % 4 == 0 && g.sB1 % 4 == 0 && g.sB2 % 4 == 0;
  }
  const int tm = (g.M + 63) >> 6, tn = (g.N + 63) >> 6;
  for (int ty = 0; ty < tm; ++ty)
    for (int tx = 0; tx < tn; ++tx) gemm_body<1, 1, AKC, BKC, ALIGNED>(g, lds, tx, ty, z);
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void gcn_chain_fwd_kernel(const GcnCtx c) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  const int z = blockIdx.x;  // b * H + h
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // GraphConv row normaliser (glove:47-49): rinv[i] = 1 / (sum_j A[i,j] + [sum == 0])
  {
    const float* a = c.A + (long)z * c.N * c.N;
    for (int i = wave; i < c.N; i += 4) {
      float s = 0.f;
      for (int j = lane; j < c.N; j += 64) s += a[(long)i * c.N + j];
      s = wave_sum(s);
      if (lane == 0) c.rinv[(long)z * c.N + i] = 1.f / (s + (s == 0.f ? 1.f : 0.f));
    }
  }
  __syncthreads();
  for (int l = 0; l < c.L; ++l) {
    if (l > 0) {
      run_product<true, false, ALIGNED>(plan_fwd_dense(c, l), lds, z);
      __syncthreads();
    }
    run_product<true, false, ALIGNED>(plan_fwd_agg(c, l), lds, z);
    __syncthreads();
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void gcn_chain_bwd_kernel(const GcnCtx c) {
  __shared__ __attribute__((aligned(16))) float lds[lds_floats<1, 1, true, true>()];
  const int z = blockIdx.x;
  const int b = z / c.H, h = z - b * c.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = c.L - 1; l >= 0; --l) {
    // through Y = relu(S), S = M * rinv:  dS = dY [Y > 0];  dM = dS rinv;  drow -= rinv sum_c dS Y
    for (int i = wave; i < c.N; i += 4) {
      const long off = ((((long)b * c.N + i) * c.H + h) * c.L + l) * c.gh;
      const long ri = (long)z * c.N + i;
      const float rv = c.rinv[ri];
      float acc = 0.f;
      for (int k = lane; k < c.gh; k += 64) {
        const float y = c.Y[off + k];
        const float g = y > 0.f ? c.dYa[off + k] : 0.f;
        c.dM[off + k] = g * rv;
        acc = fmaf(g, y, acc);
      }
      acc = wave_sum(acc);
      if (lane == 0) {  // row i is always handled by this lane: plain read-modify-write is ordered
        const float d = -rv * acc;
        c.drow[ri] = (l == c.L - 1) ? d : c.drow[ri] + d;
      }
    }
    __syncthreads();
    run_product<false, false, ALIGNED>(plan_bwd_dP(c, l), lds, z);
    run_product<true, true, ALIGNED>(plan_bwd_dA(c, l), lds, z);
    __syncthreads();
    if (l > 0) {
      run_product<true, true, ALIGNED>(plan_bwd_dY(c, l), lds, z);
      __syncthreads();
    }
  }
}

static bool chain_aligned(const GcnCtx& c, bool bwd) {
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = c.N % 64 == 0 && c.gh % 64 == 0 && al(c.A) && al(c.flat + c.oWd) && al(c.Pn) && al(c.Y);
  if (bwd) ok = ok && al(c.dYa) && al(c.dM) && al(c.dP) && al(c.dA);
  else ok = ok && al(c.G) && al(c.HO) && al(c.X);
  return ok;
}

int gcn_chain_fwd(const GcnCtx& c, hipStream_t st) {
  GC_REQUIRE((long)c.B * c.H <= 0x7fffffffL, "gcn_chain_fwd: too many (doc, head) pairs");
  dim3 grid((unsigned)(c.B * c.H)), block(256);
  ProfScope ps("gcn_chain_fwd", st);
  if (chain_aligned(c, false)) hipLaunchKernelGGL(gcn_chain_fwd_kernel<true>, grid, block, 0, st, c);
  else hipLaunchKernelGGL(gcn_chain_fwd_kernel<false>, grid, block, 0, st, c);
  return check_launch("gcn_chain_fwd");
}

int gcn_chain_bwd(const GcnCtx& c, hipStream_t st) {
  dim3 grid((unsigned)(c.B * c.H)), block(256);
  ProfScope ps("gcn_chain_bwd", st);
  if (chain_aligned(c, true)) hipLaunchKernelGGL(gcn_chain_bwd_kernel<true>, grid, block, 0, st, c);
  else hipLaunchKernelGGL(gcn_chain_bwd_kernel<false>, grid, block, 0, st, c);
  return check_launch("gcn_chain_bwd");
}

}  // namespace gc
