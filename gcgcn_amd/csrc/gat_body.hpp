// GATAttention backward up to the edge pass for small graphs (N <= 64), one workgroup per (document, feature slice):
//   dlogit = softmax_bwd(P, dropout_bwd(dA))      [N x N, kept in LDS; optionally written out]
//   ds[j]  = sum_i dlogit[i, j]                   (gradient of the node score u.x_j + c)
//   dX[j]  = ds[j] u (+ dXin[j])
// A device body so that it can run as its own launch (rowops.hip) or as passenger workgroups of the edge pass (edge.hip),
// whose entity rows compute their own dlogit row and need nothing from here.
#pragma once
#include "common.hpp"

namespace gc {

constexpr int GT = 64;
constexpr int GAT_DOC_LDS = GT * (GT + 1) + GT;   // floats of LDS the body needs

struct GatTail {        // what the passenger workgroups of the edge pass need (P == nullptr: nothing rides)
  const float* P;
  const float* dA;
  const float* uvc;
  const float* dXin;
  float* ds;
  float* dX;
  Drop drop;
  int B, slices;
};

__device__ __forceinline__ void gat_dlogit_doc(const float* __restrict__ P, const float* __restrict__ dA,
                                               const float* __restrict__ uvc, const float* __restrict__ dXin,
                                               float* __restrict__ dlogit, float* __restrict__ ds, float* __restrict__ dX, int N, int D,
                                               const Drop& drop, const int b, const int slice, const int nslices,
                                               float* __restrict__ sm) {
  float (*T)[GT + 1] = reinterpret_cast<float (*)[GT + 1]>(sm);
  float* dss = sm + GT * (GT + 1);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool dd = drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(drop) : 0;
  {  // all 16 rows of a wave are requested before the first one is reduced: one memory round trip, not sixteen
    float p[GT / 4], g[GT / 4];
#pragma unroll
    for (int u = 0; u < GT / 4; ++u) {
      const int i = wave + 4 * u;
      const long oc = ((long)b * N + min(i, N - 1)) * N + min(lane, N - 1);  // clamped: unconditional loads (a load
      const bool ok = i < N && lane < N;                                      // inside a branch is waited for at its end)
      p[u] = P[oc], g[u] = dA[oc];
      if (!ok) p[u] = 0.f, g[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < GT / 4; ++u) {
      const int i = wave + 4 * u;
      const long o = ((long)b * N + i) * N + lane;
      float gg = g[u];
      if (dd) gg = (rng_u32(key, (uint64_t)o) >= drop.thresh) ? gg * drop.scale : 0.f;
      const float dot = wave_sum(gg * p[u]);
      const float v = p[u] * (gg - dot);
      if (dlogit && i < N && lane < N && slice == 0) dlogit[o] = v;
      T[i][lane] = v;
    }
  }
  __syncthreads();
  if (t < GT) {
    float a = 0.f;
#pragma unroll 8
    for (int i = 0; i < GT; ++i) a += T[i][t];
    dss[t] = a;
    if (t < N && slice == 0) ds[(long)b * N + t] = a;
  }
  __syncthreads();
  // dX: this workgroup's slice of the feature columns (the slices share the document; each recomputes the cheap phases
  // above, only slice 0 stores dlogit / ds), four independent elements in flight per thread
  const int cw = (D + nslices - 1) / nslices, c0 = slice * cw;
  const int cn = min(cw, D - c0);
  const long base = (long)b * N * D;
  for (int e0 = t; e0 < N * cn; e0 += 4 * 256) {
    float v[4], xin[4];
    long o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256;
      const bool ok = e < N * cn;
      const int j = ok ? e / cn : 0, c = c0 + (ok ? e - j * cn : 0);
      o[u] = ok ? base + (long)j * D + c : -1;
      v[u] = dss[j] * uvc[c];
      xin[u] = dXin ? dXin[ok ? o[u] : base] : 0.f;   // dXin != NULL is uniform; the address is clamped, not guarded
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (o[u] >= 0) dX[o[u]] = v[u] + xin[u];
  }
}

}  // namespace gc
