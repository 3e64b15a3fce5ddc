// Probe: is a v_mfma_f32_16x16x4_f32 whose destination tuple PARTIALLY overlaps its SrcC tuple safe on gfx950?
//
// Why: the compiler (ROCm 7.2 LLVM) treats 4-register MFMA results as free to overlap SrcC in any way (only results wider
// than four registers get an early-clobber destination), and under register pressure it emits sequences such as
//     v_mfma_f32_16x16x4_f32 v[14:17], v6, v10, v[16:19]
//     v_mfma_f32_16x16x4_f32 v[14:17], v7, v11, v[14:17]
//     v_mfma_f32_16x16x4_f32 v[14:17], v8, v12, v[14:17]
//     v_mfma_f32_16x16x4_f32 v[16:19], v9, v13, v[14:17]
// -- found in the one build of gcn_chain_t_bwd_kernel<192,4,false> that returned a different dA on every run, in exactly the
// accumulator (dacc[0][2]) whose elements 2 and 3 were the ones that varied.
//
// Each variant runs the same four dependent products from the same operands; `tied` keeps destination == SrcC throughout
// and is the expected value.  12 waves per workgroup (3 per SIMD, as in the kernel) so that other waves' MFMAs interleave.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_overlap_probe.hip -o tools/mfma_overlap_probe && tools/mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NOPS "s_nop 15\n\ts_nop 15\n\t"

template <int VAR>
__device__ __forceinline__ void run4(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]);

#define IO                                                                                                                       \
  : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])                                                                            \
  : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) \
  : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

template <>
__device__ __forceinline__ void run4<0>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v22, %6\n\tv_mov_b32 v23, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[20:23], %8, %12, v[20:23]\n\t"
      "v_mfma_f32_16x16x4_f32 v[20:23], %9, %13, v[20:23]\n\t"
      "v_mfma_f32_16x16x4_f32 v[20:23], %10, %14, v[20:23]\n\t"
      "v_mfma_f32_16x16x4_f32 v[20:23], %11, %15, v[20:23]\n\t" NOPS
      "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23\n\t" IO);
}
template <>
__device__ __forceinline__ void run4<1>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v16, %4\n\tv_mov_b32 v17, %5\n\tv_mov_b32 v18, %6\n\tv_mov_b32 v19, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %8, %12, v[16:19]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %9, %13, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %10, %14, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[16:19], %11, %15, v[14:17]\n\t" NOPS
      "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19\n\t" IO);
}
// the same with every product waiting for the one before it to retire (no result forwarding, registers read back)
template <>
__device__ __forceinline__ void run4<2>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v16, %4\n\tv_mov_b32 v17, %5\n\tv_mov_b32 v18, %6\n\tv_mov_b32 v19, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %8, %12, v[16:19]\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %9, %13, v[14:17]\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %10, %14, v[14:17]\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[16:19], %11, %15, v[14:17]\n\t" NOPS
      "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19\n\t" IO);
}
// only the first product shifts (destination = SrcC - 2), the rest tied
template <>
__device__ __forceinline__ void run4<3>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v16, %4\n\tv_mov_b32 v17, %5\n\tv_mov_b32 v18, %6\n\tv_mov_b32 v19, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %8, %12, v[16:19]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %9, %13, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %10, %14, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %11, %15, v[14:17]\n\t" NOPS
      "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\tv_mov_b32 %2, v16\n\tv_mov_b32 %3, v17\n\t" IO);
}
// only the last product shifts (destination = SrcC + 2), the rest tied
template <>
__device__ __forceinline__ void run4<4>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v14, %4\n\tv_mov_b32 v15, %5\n\tv_mov_b32 v16, %6\n\tv_mov_b32 v17, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[14:17], %8, %12, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %9, %13, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[14:17], %10, %14, v[14:17]\n\t"
      "v_mfma_f32_16x16x4_f32 v[16:19], %11, %15, v[14:17]\n\t" NOPS
      "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19\n\t" IO);
}
// destination = SrcC + 2 fed by VALU-written registers (no MFMA in front of it)
template <>
__device__ __forceinline__ void run4<5>(const float (&a)[4], const float (&b)[4], const float (&c)[4], float (&d)[4]) {
  asm volatile(
      "v_mov_b32 v14, %4\n\tv_mov_b32 v15, %5\n\tv_mov_b32 v16, %6\n\tv_mov_b32 v17, %7\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[16:19], %8, %12, v[14:17]\n\t" NOPS
      "v_mfma_f32_16x16x4_f32 v[16:19], %9, %13, v[16:19]\n\t"
      "v_mfma_f32_16x16x4_f32 v[16:19], %10, %14, v[16:19]\n\t"
      "v_mfma_f32_16x16x4_f32 v[16:19], %11, %15, v[16:19]\n\t" NOPS
      "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19\n\t" IO);
}

__device__ __forceinline__ float hashf(unsigned x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return (float)(x & 0xffff) * (1.f / 65536.f) - 0.5f;
}

template <int VAR>
__global__ __launch_bounds__(768) void probe(float* out, int rounds) {
  const unsigned gid = blockIdx.x * 768 + threadIdx.x;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
    float a[4], b[4], c[4], d[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[v] = hashf(gid * 97u + r * 13u + v);
      b[v] = hashf(gid * 89u + r * 17u + v + 1000u);
      c[v] = hashf(gid * 83u + r * 19u + v + 2000u);
    }
    run4<VAR>(a, b, c, d);
#pragma unroll
    for (int v = 0; v < 4; ++v) s[v] += d[v];   // (exact order: the same in every variant)
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) out[(size_t)gid * 4 + v] = s[v];
}

int main(int argc, char** argv) {
  const int blocks = 256, rounds = argc > 1 ? atoi(argv[1]) : 200, reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t n = (size_t)blocks * 768 * 4;
  float* dev;
  if (hipMalloc(&dev, n * 4) != hipSuccess) return 2;
  float* ref = (float*)malloc(n * 4), *got = (float*)malloc(n * 4);
  hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(768), 0, 0, dev, rounds);
  hipMemcpy(ref, dev, n * 4, hipMemcpyDeviceToHost);
  const char* names[] = {"tied (expected value, run again)", "down-shift, tied, tied, up-shift (the compiler's sequence)",
                         "the same, every product retired before the next", "first product: destination = SrcC - 2",
                         "last product: destination = SrcC + 2 (SrcC = the product before it)",
                         "first product: destination = SrcC + 2 (SrcC written by the vector ALU)"};
  int any = 0;
  for (int var = 0; var < 6; ++var) {
    long bad[4] = {0, 0, 0, 0}, runs_bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
      hipMemset(dev, 0xff, n * 4);
      switch (var) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
        case 5: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(768), 0, 0, dev, rounds); break;
      }
      if (hipMemcpy(got, dev, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
      long b0 = 0;
      for (size_t i = 0; i < n; ++i)
        if (memcmp(&got[i], &ref[i], 4) != 0) ++bad[i & 3], ++b0;
      runs_bad += b0 != 0;
    }
    printf("variant %d  %-72s launches differing %ld / %d   differing values by element [0..3]: %ld %ld %ld %ld\n", var, names[var],
           runs_bad, reps, bad[0], bad[1], bad[2], bad[3]);
    any |= (var > 0 && runs_bad);
  }
  printf("%s\n", any ? "PARTIAL OVERLAP IS NOT SAFE in at least one form" : "every form agreed with the tied sequence");
  return 0;
}
