// Lab bench for a 128x128 fp32-MFMA GEMM tile (NN: A [M][K], B [K][N], C [M][N], everything a multiple of the tile):
// A from LDS with one 16-byte read per four MFMAs (k-contiguous image, the four k of a read go to four MFMAs, lanes 32-63
// take the next four), B from a k-major image, 2x2 accumulators per wave, register prefetch two k-steps ahead, LDS double
// buffer, barriers that wait for LDS only.  Compared with the product kernel (gcgcn_gemm) by tools/gemm_bench.py numbers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"   // hipError_t results of the harness calls
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LA = BK + 4;    // A image [BM][LA]: 16-byte rows, conflict-free 16-byte reads (pitch = 4 mod 32)
constexpr int LB = BN + 4;    // B image [BK][LB]
constexpr int SA = BM * LA, SB = BK * LB;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void gemm128(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, hf = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tn = N / BN;
  // XCD-contiguous tile order: consecutive workgroups of an XCD walk consecutive tiles
  const int nb = gridDim.x, per = nb / 8;
  const int bid = (nb % 8 == 0) ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int m0 = (bid / tn) * BM, n0 = (bid % tn) * BN;
  const float* __restrict__ Ag = A + (long)m0 * K;
  const float* __restrict__ Bg = B + n0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  f32x4 ra[2][4], rb[2][4];
  auto load = [&](const int kt, f32x4 (&a)[4], f32x4 (&b)[4]) __attribute__((always_inline)) {
    const int k0 = kt * BK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const f32x4*>(Ag + (long)((t >> 3) + 32 * u) * K + k0 + (t & 7) * 4);
      b[u] = *reinterpret_cast<const f32x4*>(Bg + (long)(k0 + (t >> 5) + 8 * u) * N + (t & 31) * 4);
    }
  };
  auto store = [&](const int st, const f32x4 (&a)[4], const f32x4 (&b)[4]) __attribute__((always_inline)) {
    float* as = lds + st * (SA + SB);
    float* bs = as + SA;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<f32x4*>(as + ((t >> 3) + 32 * u) * LA + (t & 7) * 4) = a[u];
      *reinterpret_cast<f32x4*>(bs + ((t >> 5) + 8 * u) * LB + (t & 31) * 4) = b[u];
    }
  };
  auto compute = [&](const int st) __attribute__((always_inline)) {
    const float* as = lds + st * (SA + SB) + (wr * 64 + r) * LA + 4 * hf;
    const float* bs = lds + st * (SA + SB) + SA + (4 * hf) * LB + wc * 64 + r;
    f32x4 av[2][2];
    float bv[2][2][4];
    auto fetch = [&](const int g, f32x4 (&a)[2], float (&b)[2][4]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LA + 8 * g);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][m] = bs[(8 * g + m) * LB + j * 32];
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) fetch(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][m], bv[g & 1][j][m], acc[i][j], 0, 0, 0);
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nk = K / BK;
  load(0, ra[0], rb[0]);
  if (nk > 1) load(1, ra[1], rb[1]);
  store(0, ra[0], rb[0]);
  lds_barrier();
  for (int kt = 0; kt < nk; kt += 2) {   // two k-steps per trip: register sets and LDS stages are compile-time constants
    if (kt + 2 < nk) load(kt + 2, ra[0], rb[0]);        // set 0 was stored before the last barrier
    compute(0);
    if (kt + 1 < nk) store(1, ra[1], rb[1]);
    lds_barrier();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) load(kt + 3, ra[1], rb[1]);
    compute(1);
    if (kt + 2 < nk) store(0, ra[0], rb[0]);
    lds_barrier();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wr * 64 + i * 32 + hf * 4 + (q & 3) + 8 * (q >> 2), col = n0 + wc * 64 + j * 32 + r;
        C[(long)row * N + col] = acc[i][j][q];
      }
}

static float* dalloc(size_t n, std::vector<float>* keep) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  if (keep) *keep = h;
  return d;
}

template <int VARIANT>
static void run(int M, int N, int K) {
  std::vector<float> hA, hB;
  float* A = dalloc((size_t)M * K, &hA);
  float* B = dalloc((size_t)K * N, &hB);
  float* C;
  hipMalloc(&C, (size_t)M * N * 4);
  dim3 grid((M / BM) * (N / BN)), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm128<VARIANT>, grid, block, 0, 0, A, B, C, M, N, K);
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm128<VARIANT>, grid, block, 0, 0, A, B, C, M, N, K);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int s = 0; s < 64; ++s) {
    const int i = rand() % M, j = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)k * N + j];
    const double e = fabs(ref - hC[(size_t)i * N + j]);
    if (e > maxerr) maxerr = e;
  }
  const double us = ms * 1e3 / reps, flop = 2.0 * M * N * K;
  printf("variant %d  M=%5d N=%5d K=%5d : %8.1f us %6.1f TF/s  (max err of 64 samples %.2e)\n", VARIANT, M, N, K, us,
         flop / us / 1e6, maxerr);
  hipFree(A), hipFree(B), hipFree(C);
}

int main() {
  run<0>(4096, 2048, 256);   // two of the path's 2048 x 2048 x 256 problems in one launch (512 tiles: two workgroups per CU)
  run<1>(4096, 2048, 256);
  run<0>(2048, 2048, 256);
  run<1>(2048, 2048, 256);
  run<0>(2048, 2048, 2048);
  run<1>(2048, 2048, 2048);
  run<0>(4096, 4096, 4096);
  run<1>(4096, 4096, 4096);
  return 0;
}
