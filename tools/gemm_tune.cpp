// Standalone timing of gcgcn_gemm (no Python): hipcc tools/gemm_tune.cpp -Iinclude -Lgcgcn_amd/lib -lgcgcn_hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gcgcn.h"

static float* dalloc(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

static void run(const char* name, int M, int N, int K, int akc, int bkc, int batch, int tile, int splits) {
  float* A = dalloc((size_t)batch * M * K, 1.f);
  float* B = dalloc((size_t)batch * K * N, 1.f);
  float *C, *ws;
  hipMalloc(&C, (size_t)batch * M * N * 4);
  const long wse = 8L << 20;
  hipMalloc(&ws, wse * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto go = [&]() {
    int rc = gcgcn_gemm(M, N, K, A, akc ? K : M, akc, B, bkc ? K : N, bkc, C, N, batch, (long)M * K, (long)K * N,
                        (long)M * N, 1.f, nullptr, 0, 0, tile, splits, ws, wse, nullptr);
    if (rc) { printf("error: %s\n", gcgcn_last_error()); exit(1); }
  };
  for (int i = 0; i < 5; ++i) go();
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) go();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, flop = 2.0 * M * N * K * batch;
  printf("%-22s M=%5d N=%5d K=%5d b=%3d %c%c tile=%d splits=%2d : %7.1f us %6.1f TF/s\n", name, M, N, K, batch,
         akc ? 'N' : 'T', bkc ? 'T' : 'N', tile, splits, us, flop / us / 1e6);
  hipFree(A), hipFree(B), hipFree(C), hipFree(ws);
}

int main(int argc, char** argv) {
  if (argc > 1) {  // layout sweep on a large problem: which operand path limits the steady state?
    run("NN big", 4096, 4096, 4096, 1, 0, 1, 1, 1);
    run("NT big", 4096, 4096, 4096, 1, 1, 1, 1, 1);
    run("TN big", 4096, 4096, 4096, 0, 0, 1, 1, 1);
    run("TT big", 4096, 4096, 4096, 0, 1, 1, 1, 1);
    run("NN big t2", 4096, 4096, 4096, 1, 0, 1, 2, 1);
    run("TN big t2", 4096, 4096, 4096, 0, 0, 1, 2, 1);
    return 0;
  }

  for (int tile = 1; tile <= 2; ++tile) {
    run("NN K=256", 2048, 2048, 256, 1, 0, 1, tile, 1);
    run("NN K=2048", 2048, 2048, 2048, 1, 0, 1, tile, 1);
    run("NT K=2048 n256", 2048, 256, 2048, 1, 1, 1, tile, 1);
    run("NT K=2048 n256", 2048, 256, 2048, 1, 1, 1, tile, 8);
    run("TN K=2048", 256, 2048, 2048, 0, 0, 1, tile, 1);
    run("TN K=2048", 256, 2048, 2048, 0, 0, 1, tile, 4);
    run("TN K=2048", 256, 2048, 2048, 0, 0, 1, tile, 8);
    run("NN big", 4096, 4096, 4096, 1, 0, 1, tile, 1);
  }
  return 0;
}
