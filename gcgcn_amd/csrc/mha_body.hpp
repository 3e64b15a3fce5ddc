// Device bodies of the MultiHeadAttention core for small graphs (N <= 64): shared by mha_core.hip (launches of their own) and
// gemm.hip (the pairs of the backward ride as passenger workgroups of a group GEMM launch).  See mha_core.hip for the algorithm.
#pragma once
#include "common.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int MT = 64;       // padded graph size
constexpr int MS = MT + 1;   // row stride of the score tile in LDS
constexpr int MKC = 128;     // head-feature chunk held in LDS at a time

static inline int mha_chunk(int dh) { return dh < MKC ? (dh + 31) / 32 * 32 : MKC; }
static inline size_t mha_lds_bytes(int dh) { return sizeof(float) * (MT * MS + MT * (mha_chunk(dh) + 1)); }

typedef float f16v __attribute__((ext_vector_type(16)));

// Q_h chunk [N x kc] (row stride D in global) -> LDS rows of odd stride ld (conflict-free for both the k-contiguous
// reads of the score product and the column-contiguous reads of the gradient product); rows >= N and columns
// kc .. kpad are zero.
__device__ __forceinline__ void load_q_chunk(float* qs, const float* __restrict__ q, int N, int D, int k0, int kc, int kpad,
                                             int ld, int t) {
  const int k4 = kpad >> 2;
  for (int base = t; base < MT * k4; base += 4 * 256) {  // four independent loads in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256, n = idx / k4, k = (idx - n * k4) << 2;
      // clamped address + select instead of a guarded load: a load inside a branch is waited for at the branch's end,
      // which serialises the four requests
      v[u] = *reinterpret_cast<const float4*>(q + (long)min(n, N - 1) * D + k0 + min(k, kc - 4));
      if (!(idx < MT * k4 && n < N && k < kc)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256, n = idx / k4, k = (idx - n * k4) << 2;
      if (idx < MT * k4) {
        float* d = qs + n * ld + k;
        d[0] = v[u].x, d[1] = v[u].y, d[2] = v[u].z, d[3] = v[u].w;
      }
    }
  }
}

// v_mfma_f32_32x32x2_f32 operand / result mapping (wave of 64 lanes): A[row = lane & 31][k = lane >> 5],
// B[k = lane >> 5][col = lane & 31], D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31].
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// P = softmax(alpha Q_h Q_h^T) over the valid columns, A = dropout(P) for one (document, head) pair z (glove:136-140).
// sm: MT * MS + MT * (kchunk + 1) floats of LDS.  The first 256 threads of the workgroup do the work (`t` their index, `on`
// false for any further waves, which only keep the barriers company): a launch of its own runs it with 256 threads
// (mha_core.hip), the LDS-resident chain kernels run it in their prologue (chain.hip: a_lds != NULL receives the adjacency the
// chain will use -- A, or P when there is no dropout -- as an image of row pitch a_pitch, so that the chain reads it from LDS
// and the attention core needs no launch).  Same arithmetic in both uses, bit for bit.
// LB: the barriers wait for this wave's LDS operations only (the chain kernels: __syncthreads() would also wait for every
// outstanding prefetch of the chain to land and for the P / A stores below to be acknowledged -- nobody in the kernel reads those).
template <bool LB = false>
__device__ __forceinline__ void mha_core_fwd_body(float* __restrict__ sm, const int z, const float* __restrict__ Q,
                                                  const int* __restrict__ n_valid, float* __restrict__ P, float* __restrict__ A,
                                                  int N, int D, int H, int dh, int kchunk, float alpha, const Drop& drop,
                                                  const int t, const bool on, float* __restrict__ a_lds, const int a_pitch,
                                                  const int nwaves) {
  float* S = sm;             // [MT][MS]
  float* qs = sm + MT * MS;  // [MT][kchunk + 1]
  auto barrier = [&]() __attribute__((always_inline)) {
    if constexpr (LB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  const int b = z / H, h = z - b * H;
  const int lane = t & 63, wave = t >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const float* q = Q + (long)b * N * D + (long)h * dh;
  // scores: wave (wr, wc) owns the 32 x 32 quadrant S[32 wr .., 32 wc ..] = Q[32 wr ..] Q[32 wc ..]^T
  const int wr = (wave >> 1) & 1, wc = wave & 1, ld = kchunk + 1;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < dh; k0 += kchunk) {
    const int kc = min(kchunk, dh - k0);
    if (k0) barrier();
    if (on) load_q_chunk(qs, q, N, D, k0, kc, (kc + 3) & ~3, ld, t);
    barrier();
    if (on) {
      const float* pa = qs + (32 * wr + (lane & 31)) * ld + (lane >> 5);
      const float* pb = qs + (32 * wc + (lane & 31)) * ld + (lane >> 5);
      for (int k = 0; k < kc; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
    }
  }
  if (on) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(32 * wr + mfma_row(r, lane)) * MS + 32 * wc + (lane & 31)] = acc[r] * alpha;
  }
  barrier();
  // row softmax over the valid columns, dropout; lane = column; every wave of the workgroup takes rows
  const bool dd = A && drop.snap;
  const uint64_t key = dd ? drop_key(drop) : 0;
  for (int i = wave; i < N; i += nwaves) {
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
    if (a_lds && lane < MT) a_lds[i * a_pitch + lane] = (lane < N) ? v : 0.f;
  }
}

// dQ_h = alpha (dS + dS^T) Q_h,  dS = P (dP - sum_j dP P),  dP = dropout_bwd(dA).   Padding entries have P == 0.
// One (document, head) pair z; sm: MT * MS + MT * (kchunk + 1) floats of LDS; 256 threads.  A device function because two
// kernels run it: mha_core_bwd_kernel (a launch of its own) and the group GEMM launch of the convolution's backward, where
// the pairs ride as passenger workgroups (gemm.hip gemm_group_pass_kernel).
__device__ __forceinline__ void mha_core_bwd_body(float* __restrict__ sm, const int z, const float* __restrict__ Q,
                                                  const float* __restrict__ P, const float* __restrict__ dA,
                                                  float* __restrict__ dQ, int N, int D, int H, int dh, int kchunk, float alpha,
                                                  const Drop& drop) {
  float* T = sm;             // [MT][MS]
  float* qs = sm + MT * MS;  // [MT][kchunk + 1]
  const int b = z / H, h = z - b * H;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool dd = drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(drop) : 0;
  {  // all 16 rows of a wave are requested before the first one is reduced: one memory round trip, not sixteen
    float p[MT / 4], g[MT / 4];
#pragma unroll
    for (int u = 0; u < MT / 4; ++u) {
      const int i = wave + 4 * u;
      const long oc = ((long)z * N + min(i, N - 1)) * N + min(lane, N - 1);  // clamped: unconditional loads
      const bool ok = i < N && lane < N;
      p[u] = P[oc], g[u] = dA[oc];
      if (!ok) p[u] = 0.f, g[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < MT / 4; ++u) {
      const int i = wave + 4 * u;
      float gg = g[u];
      if (dd) gg = (rng_u32(key, (uint64_t)(((long)z * N + i) * N + lane)) >= drop.thresh) ? gg * drop.scale : 0.f;
      const float dot = wave_sum(gg * p[u]);
      T[i * MS + lane] = p[u] * (gg - dot);
    }
  }
  __syncthreads();
  // symmetrise in place: the pair (i, j), (j, i) belongs to one thread
  for (int idx = t; idx < MT * MT; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    if (i < j) {
      const float s = T[i * MS + j] + T[j * MS + i];
      T[i * MS + j] = s, T[j * MS + i] = s;
    } else if (i == j) {
      T[i * MS + i] *= 2.f;
    }
  }
  const float* q = Q + (long)b * N * D + (long)h * dh;
  float* dq = dQ + (long)b * N * D + (long)h * dh;
  const int ld = kchunk + 1;
  for (int k0 = 0; k0 < dh; k0 += kchunk) {
    const int kc = min(kchunk, dh - k0), kpad = (kc + 31) & ~31;
    __syncthreads();  // T symmetrised / previous chunk consumed
    load_q_chunk(qs, q, N, D, k0, kc, kpad, ld, t);
    __syncthreads();
    // 32 x 32 output blocks (row block rb, column block cb) of the chunk, dealt round-robin to the waves
    for (int blk = wave; blk < 2 * (kpad >> 5); blk += 4) {
      const int rb = blk & 1, cb = blk >> 1;
      const float* pa = T + (32 * rb + (lane & 31)) * MS + (lane >> 5);
      const float* pb = qs + (lane >> 5) * ld + 32 * cb + (lane & 31);
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
      for (int j = 0; j < MT; j += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j], pb[j * ld], acc, 0, 0, 0);
      const int c = 32 * cb + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * rb + mfma_row(r, lane);
        if (i < N && c < kc) dq[(long)i * D + k0 + c] = acc[r] * alpha;
      }
    }
  }
}


}  // namespace gc
