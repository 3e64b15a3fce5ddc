// Standalone timing of the edge streaming kernels through the C ABI (no Python).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "gcgcn.h"

static void run(int B, int N, int D) {
  const size_t ne = (size_t)B * N * N * D, nb = (size_t)B * N * D;
  float *E, *dE, *Ebar;
  hipMalloc(&E, ne * 4), hipMalloc(&dE, ne * 4), hipMalloc(&Ebar, nb * 4);
  hipMemset(E, 0x3c, ne * 4), hipMemset(Ebar, 0x3c, nb * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = 20;
  float ms;
  for (int w = 0; w < 3; ++w) gcgcn_edge_mean_fwd(B, N, D, E, nullptr, Ebar, nullptr);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) gcgcn_edge_mean_fwd(B, N, D, E, nullptr, Ebar, nullptr);
  hipEventRecord(e1, 0), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
  printf("B=%4d N=%3d D=%3d  E=%7.1f MB | mean_fwd %7.1f us %6.0f GB/s", B, N, D, ne * 4 / 1e6, ms * 1e3 / reps,
         ne * 4.0 / (ms * 1e-3 / reps) / 1e9);
  for (int w = 0; w < 3; ++w) gcgcn_edge_mean_bwd(B, N, D, Ebar, nullptr, dE, nullptr);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) gcgcn_edge_mean_bwd(B, N, D, Ebar, nullptr, dE, nullptr);
  hipEventRecord(e1, 0), hipEventSynchronize(e1), hipEventElapsedTime(&ms, e0, e1);
  printf(" | bcast_bwd %7.1f us %6.0f GB/s\n", ms * 1e3 / reps, ne * 4.0 / (ms * 1e-3 / reps) / 1e9);
  hipFree(E), hipFree(dE), hipFree(Ebar);
}

int main() {
  run(32, 64, 256);
  run(128, 64, 256);
  run(256, 64, 256);
  run(32, 64, 768);
  run(8, 256, 512);
  return 0;
}
