// Row bodies of the edge-feature streaming kernels (see edge.hip for the algorithm): one workgroup of EWT waves per
// entity row (b, i).  They live in a header because two kernels instantiate them: the stand-alone launches in edge.hip
// (EWT = 4) and the chain kernels' passenger workgroups (chain.hip, EWT = 8), which stream E on the compute units a
// small batch of (doc, head) chains leaves idle.
#pragma once
#include "common.hpp"

namespace gc {

template <int VEC>
__device__ __forceinline__ void vload(float (&r)[VEC], const float* p) {
  if constexpr (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r[0] = v.x, r[1] = v.y, r[2] = v.z, r[3] = v.w;
  } else {
    r[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    p[0] = r[0];
  }
}

// Streaming variants: dE is written once and never read on this path, E of a mean-only hop is read once per
// step.  Non-temporal accesses keep them from displacing E1 (re-read by the next forward) and the GEMM
// operands in the 256 MiB Infinity Cache.
template <int VEC>
__device__ __forceinline__ void vload_nt(float (&r)[VEC], const float* p) {
  if constexpr (VEC == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    r[0] = v.x, r[1] = v.y, r[2] = v.z, r[3] = v.w;
  } else {
    r[0] = __builtin_nontemporal_load(p);
  }
}
template <int VEC>
__device__ __forceinline__ void vstore_nt(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v;
    v.x = r[0], v.y = r[1], v.z = r[2], v.w = r[3];
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  } else {
    __builtin_nontemporal_store(r[0], p);
  }
}

constexpr int EW = 4;    // waves per workgroup
constexpr int EUNR = 4;  // rows in flight per wave (8 was tried: no faster in the step, 152 VGPRs in edge_bwd)

// ---------------------------------------------------------------------------------------------
// forward: Ebar (always) and raw logits v.e_ij (ATT only).  dynamic LDS: EW * D floats.
// ---------------------------------------------------------------------------------------------
// ATT: logits stay in LDS; the row's softmax (+ coladd[b, j] = u.x_j + c, + dropout) is finished by
// wave 0 in the same launch (GATAttention glove:162-167), so P/A are the only attention outputs.
// EWT waves (the whole workgroup) work on entity row bi; cs = EWT * D (+ N when ATT) floats of LDS.  Control flow is
// uniform over the workgroup, so the body can also ride inside another kernel's spare workgroups (chain.hip).
template <int VEC, bool ATT, bool NTL, int EWT>
__device__ __forceinline__ void edge_fwd_row(const float* __restrict__ E, const float* __restrict__ v,
                                             const int* __restrict__ n_valid, float* __restrict__ Ebar,
                                             const float* __restrict__ coladd, float* __restrict__ P,
                                             float* __restrict__ Aout, const Drop& drop, int N, int D, int bi,
                                             float* __restrict__ cs, const unsigned char* __restrict__ mask = nullptr) {
  const int b = bi / N, i = bi - b * N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  float* eb = Ebar + (long)bi * D;
  float* lg = cs + (long)EWT * D;  // [N] (ATT only)
  if (i >= nv) {  // padding entity: outputs are zero, nothing is read
    for (int c = t; c < D; c += 64 * EWT) eb[c] = 0.f;
    if (ATT)
      for (int j = t; j < N; j += 64 * EWT) {
        P[(long)bi * N + j] = 0.f;
        if (Aout) Aout[(long)bi * N + j] = 0.f;
      }
    return;
  }
  const float* __restrict__ Er = E + (long)bi * N * D;
  const int nchunk = (D + 64 * VEC - 1) / (64 * VEC);
  for (int q = 0; q < nchunk; ++q) {
    const int c = (q * 64 + lane) * VEC;
    const bool act = c < D;
    float vr[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) vr[e] = 0.f, acc[e] = 0.f;
    if (ATT && act) vload<VEC>(vr, v + c);
    int j = wave;
    for (; j + (EUNR - 1) * EWT < nv; j += EUNR * EWT) {
      float x[EUNR][VEC];
#pragma unroll
      for (int u = 0; u < EUNR; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[u][e] = 0.f;
        if (act) {
          if (NTL) vload_nt<VEC>(x[u], Er + (long)(j + u * EWT) * D + c);
          else vload<VEC>(x[u], Er + (long)(j + u * EWT) * D + c);
        }
      }
#pragma unroll
      for (int u = 0; u < EUNR; ++u) {
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          acc[e] += x[u][e];
          dot = fmaf(x[u][e], vr[e], dot);
        }
        if (ATT) {
          dot = wave_sum(dot);
          if (lane == 0) {  // row j is owned by this wave: no race on lg[j]
            if (q == 0) lg[j + u * EWT] = dot;
            else lg[j + u * EWT] += dot;
          }
        }
      }
    }
    for (; j < nv; j += EWT) {
      float x[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) x[e] = 0.f;
      if (act) vload<VEC>(x, Er + (long)j * D + c);
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc[e] += x[e];
        dot = fmaf(x[e], vr[e], dot);
      }
      if (ATT) {
        dot = wave_sum(dot);
        if (lane == 0) {
          if (q == 0) lg[j] = dot;
          else lg[j] += dot;
        }
      }
    }
    if (act) vstore<VEC>(cs + wave * D + c, acc);
  }
  __syncthreads();
  const float inv = 1.f / (float)nv;
  for (int c = t; c < D; c += 64 * EWT) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < EWT; ++w) s += cs[w * D + c];
    eb[c] = s * inv;
  }
  if (ATT && wave == 0) {  // row softmax over the nv real columns
    const float* ca = coladd + (long)b * N;
    // energies; opt-in, paper-faithful mask: energy.masked_fill(mask, -100000.0) (the in-place form glove:163-164 meant)
    for (int j = lane; j < nv; j += 64) lg[j] = (mask && mask[(long)bi * N + j]) ? -100000.0f : lg[j] + ca[j];
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


// ---------------------------------------------------------------------------------------------
// backward of the mean alone: dE[b,i,j,:] = dEbar[b,i,:] / n   (pure streaming store)
// ---------------------------------------------------------------------------------------------
template <int VEC, int EWT>
__device__ __forceinline__ void edge_bcast_row(const float* __restrict__ dEbar, const int* __restrict__ n_valid,
                                               float* __restrict__ dE, int N, int D, int nt, int bi) {
  const int b = bi / N, i = bi - b * N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  float* dEr = dE + (long)bi * N * D;
  const bool rowpad = i >= nv;
  const float inv = rowpad ? 0.f : 1.f / (float)nv;
  const int nchunk = (D + 64 * VEC - 1) / (64 * VEC);
  for (int q = 0; q < nchunk; ++q) {
    const int c = (q * 64 + lane) * VEC;
    if (c >= D) continue;
    float g[VEC], z[VEC];
    vload<VEC>(g, dEbar + (long)bi * D + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] *= inv, z[e] = 0.f;
    for (int j = wave; j < N; j += EWT) {
      const bool live = !rowpad && j < nv;
      if (nt) vstore_nt<VEC>(dEr + (long)j * D + c, live ? g : z);
      else vstore<VEC>(dEr + (long)j * D + c, live ? g : z);
    }
  }
}

}  // namespace gc
