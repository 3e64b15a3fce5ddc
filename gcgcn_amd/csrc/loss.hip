// Per-document training loss of the reference's trainer (SURVEY 8 row f2, config/Config.py:355-366):
//   p = sigmoid(logits[h, t, :]);  loss = sum_{h != t} mean_r BCE(p_r, y_r) / (n^2 - n)
// written there as N^2 - N separate nn.BCELoss launches in a Python loop.  Here: one streaming pass over the
// [B, N, N, R] logits forward (one workgroup per (doc, head entity) row, deterministic two-stage sum) and one
// elementwise pass backward.  Arithmetic follows ATen: both logs clamped at -100 forward;
// dL/dp = (p - y) / max(p (1 - p), 1e-12) backward, times dp/dx = p (1 - p).
// Ragged batches: only the first n_valid[b] entities of a document form pairs; everything else gets zero gradient.
// A document with fewer than two entities has no pairs: its loss is 0 / 0 = NaN, as in the reference.
#include "rowops.hpp"

namespace gc {

__device__ __forceinline__ float bce_term(float x, float y) {
  const float p = 1.f / (1.f + expf(-x));
  return -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
}

// part[b * N + h] = sum_{t != h, t < nv} sum_r bce(logits[b,h,t,r], labels[b,h,t,r])   (0 for h >= nv)
__global__ __launch_bounds__(256) void pair_bce_rows_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                            const int* __restrict__ n_valid, float* __restrict__ part, int N,
                                                            int R) {
  __shared__ float red[4];
  const int bh = blockIdx.x, b = bh / N, h = bh - b * N;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float acc = 0.f;
  if (h < nv) {
    const long base = (long)bh * N * R;
    const int tot = nv * R;  // the first nv tail entities are contiguous
    for (int e0 = t; e0 < tot; e0 += 4 * 256) {  // four independent elements in flight per thread
      float x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = min(e0 + u * 256, tot - 1);
        x[u] = logits[base + e], y[u] = labels[base + e];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256;
        const float v = bce_term(x[u], y[u]);
        acc += (e < tot && e / R != h) ? v : 0.f;
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (t == 0) part[bh] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss[b] = sum_h part[b, h] / (R (nv^2 - nv)), rows in order
__global__ __launch_bounds__(64) void pair_bce_finish_kernel(const float* __restrict__ part, const int* __restrict__ n_valid,
                                                             float* __restrict__ loss, int N, int R) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  float s = 0.f;
  for (int h = 0; h < nv; ++h) s += part[(long)b * N + h];
  loss[b] = s / ((float)R * (float)(nv * nv - nv));
}

template <int VEC>
__global__ __launch_bounds__(256) void pair_bce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                           const int* __restrict__ n_valid, const float* __restrict__ dloss,
                                                           float* __restrict__ dlogits, int N, int R, long total) {
  const long e0 = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (e0 >= total) return;
  float x[VEC], y[VEC], g[VEC];
  if constexpr (VEC == 4) {
    const float4 xv = *reinterpret_cast<const float4*>(logits + e0), yv = *reinterpret_cast<const float4*>(labels + e0);
    x[0] = xv.x, x[1] = xv.y, x[2] = xv.z, x[3] = xv.w, y[0] = yv.x, y[1] = yv.y, y[2] = yv.z, y[3] = yv.w;
  } else {
    x[0] = logits[e0], y[0] = labels[e0];
  }
#pragma unroll
  for (int u = 0; u < VEC; ++u) {  // the four elements of a vector may straddle two pairs (R is odd in the reference)
    const long pair = (e0 + u) / R;
    const int tj = (int)(pair % N), h = (int)((pair / N) % N), b = (int)(pair / ((long)N * N));
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    g[u] = 0.f;
    if (h < nv && tj < nv && h != tj) {
      const float p = 1.f / (1.f + expf(-x[u]));
      const float q = p * (1.f - p);
      const float up = dloss ? dloss[b] : 1.f;
      g[u] = up * (p - y[u]) / fmaxf(q, 1e-12f) * q / ((float)R * (float)(nv * nv - nv));
    }
  }
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(dlogits + e0) = make_float4(g[0], g[1], g[2], g[3]);
  else dlogits[e0] = g[0];
}

int pair_bce_fwd(const float* logits, const float* labels, const int* n_valid, float* loss, float* part, int B, int N, int R,
                 hipStream_t st) {
  {
    ProfScope ps("pair_bce_fwd", st, 8.0 * B * N * N * R);
    hipLaunchKernelGGL(pair_bce_rows_kernel, dim3((unsigned)(B * N)), dim3(256), 0, st, logits, labels, n_valid, part, N, R);
  }
  if (int e = check_launch("pair_bce_rows")) return e;
  hipLaunchKernelGGL(pair_bce_finish_kernel, dim3(B), dim3(64), 0, st, part, n_valid, loss, N, R);
  return check_launch("pair_bce_finish");
}

int pair_bce_bwd(const float* logits, const float* labels, const int* n_valid, const float* dloss, float* dlogits, int B, int N,
                 int R, hipStream_t st) {
  const long total = (long)B * N * N * R;
  auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  ProfScope ps("pair_bce_bwd", st, 12.0 * B * N * N * R);
  if (total % 4 == 0 && al(logits) && al(labels) && al(dlogits))
    hipLaunchKernelGGL(pair_bce_bwd_kernel<4>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, logits, labels, n_valid, dloss,
                       dlogits, N, R, total);
  else
    hipLaunchKernelGGL(pair_bce_bwd_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, st, logits, labels, n_valid, dloss, dlogits,
                       N, R, total);
  return check_launch("pair_bce_bwd");
}

}  // namespace gc
