// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip in launches shaped like ours?
//   variant 0: MFMA only (operands in registers)          variant 1: + 2 ds_read_b32 per MFMA (our inner loop)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[2 * 32 * 65 + 2 * 32 * 64];
  for (int i = threadIdx.x; i < 2 * 32 * 65 + 2 * 32 * 64; i += 256) lds[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = lane * 0.01f, b = 1.f - lane * 0.01f;
  const float* as = lds + (lane & 31), *bs = lds + 2 * 32 * 65 + (lane & 31);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (VAR == 1) {
        a = as[(2 * kk + (lane >> 5)) * 65];
        b = bs[(2 * kk + (lane >> 5)) * 64];
      }
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    if (VAR == 1) __syncthreads();
  }
  float s = 0;
  for (int q = 0; q < NACC; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR, int NACC>
void run(const char* name, int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<VAR, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<VAR, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double flop = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("%-34s blocks=%5d iters=%4d : %8.1f us  %7.1f TF/s\n", name, blocks, iters, us, flop / us / 1e6);
  hipFree(out);
}

int main() {
  run<0, 1>("mfma only, 1 acc", 1024, 8);
  run<0, 1>("mfma only, 1 acc", 1024, 64);
  run<0, 1>("mfma only, 1 acc", 1024, 1024);
  run<0, 4>("mfma only, 4 acc", 256, 8);
  run<0, 4>("mfma only, 4 acc", 256, 256);
  run<1, 1>("mfma + lds reads + barrier, 1 acc", 1024, 8);
  run<1, 1>("mfma + lds reads + barrier, 1 acc", 1024, 64);
  run<1, 1>("mfma + lds reads + barrier, 1 acc", 1024, 1024);
  run<1, 4>("mfma + lds reads + barrier, 4 acc", 256, 8);
  run<1, 4>("mfma + lds reads + barrier, 4 acc", 256, 256);
  run<1, 1>("mfma + lds, 1 acc, 2048 blocks", 2048, 8);
  return 0;
}
