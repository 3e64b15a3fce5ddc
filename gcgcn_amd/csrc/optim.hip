// One-launch Adam over many tensors: the optimiser step of the reference's trainer (config/Config.py:300 `optim.Adam(...,
// lr)` and :372-373 `optimizer.step()`), torch.optim.Adam's arithmetic (no weight decay, no amsgrad):
//   m <- m + (1 - b1) (g - m);  v <- b2 v + (1 - b2) g g;  p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t counted per tensor (a parameter without a gradient in some step is skipped by torch and keeps its own t).
// The model's parameters are a few dozen tensors (one flat tensor per block); launching torch's per-tensor kernels costs
// ~5 launches each.  Here a table in device memory lists {p, g, m, v, numel, first block}; one workgroup handles 1024
// consecutive elements of one tensor (16 bytes per lane where the four pointers allow it).
#include "../../include/gcgcn.h"
#include "common.hpp"

namespace gc {

struct AdamEntry {
  float* p;
  const float* g;
  float* m;
  float* v;
  long numel;
  long block_begin;   // first workgroup of this tensor; entries are sorted by it
  float step_size;    // lr / (1 - b1^t)
  float inv_bc2_sqrt; // 1 / sqrt(1 - b2^t)
};
static_assert(sizeof(AdamEntry) == 56, "table layout is shared with gcgcn_amd/optim.py (7 x 8 bytes)");

__device__ __forceinline__ void adam1(float& p, const float g, float& m, float& v, const float b1c, const float b2, const float b2c,
                                      const float eps, const float ss, const float ib) {
  m = m + b1c * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + b2c * (g * g);            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(v) * ib + eps;
  p = p - ss * (m / denom);              // param.addcdiv_(exp_avg, denom, value = -step_size)
}

// b1c = 1 - beta1 and b2c = 1 - beta2 arrive rounded from double, as torch passes them to lerp_ / addcmul_ (1.f - 0.999f is
// 4.7e-5 away from 0.001f)
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamEntry* __restrict__ tab, int n, float b1c, float b2, float b2c,
                                                         float eps) {
  int lo = 0, hi = n - 1;                // last entry whose block_begin <= blockIdx.x
  const long blk = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block_begin <= blk) lo = mid;
    else hi = mid - 1;
  }
  const AdamEntry e = tab[lo];
  const long base = (blk - e.block_begin) * 1024 + threadIdx.x * 4;
  if (base >= e.numel) return;
  const bool vec = base + 4 <= e.numel && ((((uintptr_t)e.p) | ((uintptr_t)e.g) | ((uintptr_t)e.m) | ((uintptr_t)e.v)) & 15) == 0;
  if (vec) {
    float4 p = *reinterpret_cast<const float4*>(e.p + base), m = *reinterpret_cast<const float4*>(e.m + base);
    float4 v = *reinterpret_cast<const float4*>(e.v + base);
    const float4 g = *reinterpret_cast<const float4*>(e.g + base);
    adam1(p.x, g.x, m.x, v.x, b1c, b2, b2c, eps, e.step_size, e.inv_bc2_sqrt);
    adam1(p.y, g.y, m.y, v.y, b1c, b2, b2c, eps, e.step_size, e.inv_bc2_sqrt);
    adam1(p.z, g.z, m.z, v.z, b1c, b2, b2c, eps, e.step_size, e.inv_bc2_sqrt);
    adam1(p.w, g.w, m.w, v.w, b1c, b2, b2c, eps, e.step_size, e.inv_bc2_sqrt);
    *reinterpret_cast<float4*>(e.p + base) = p;
    *reinterpret_cast<float4*>(e.m + base) = m;
    *reinterpret_cast<float4*>(e.v + base) = v;
  } else {
    for (long i = base; i < base + 4 && i < e.numel; ++i) {
      float p = e.p[i], m = e.m[i], v = e.v[i];
      adam1(p, e.g[i], m, v, b1c, b2, b2c, eps, e.step_size, e.inv_bc2_sqrt);
      e.p[i] = p, e.m[i] = m, e.v[i] = v;
    }
  }
}

}  // namespace gc

using namespace gc;

extern "C" int gcgcn_adam_step(int n_tensors, const void* table, int64_t total_blocks, double beta1, double beta2, double eps,
                               void* stream) {
  GC_REQUIRE(n_tensors >= 0 && total_blocks >= 0, "adam_step: bad arguments");
  if (n_tensors == 0 || total_blocks == 0) return 0;
  GC_REQUIRE(table, "adam_step: null table");
  GC_REQUIRE(total_blocks <= 0x7fffffffL, "adam_step: too many elements for one launch");
  ProfScope ps("adam_step", (hipStream_t)stream);
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table,
                     n_tensors, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
  return check_launch("adam_step");
}
