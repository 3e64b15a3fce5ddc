// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the CAGGC/MAGGC path.
// wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#define GC_WAVE 64

namespace gc {

// ---- error plumbing -------------------------------------------------------------------
// The C ABI returns an int status (0 = ok) and keeps the message in a thread-local buffer
// readable through gcgcn_last_error(); no exceptions cross the boundary.
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define GC_REQUIRE(cond, ...)                      \
  do {                                             \
    if (!(cond)) {                                 \
      gc::set_error(__VA_ARGS__);                  \
      return 1;                                    \
    }                                              \
  } while (0)

#define GC_TRY(expr)                \
  do {                              \
    if (int _e = (expr)) return _e; \
  } while (0)

// ---- wave-level reductions (64 lanes) -------------------------------------------------------------
// DPP lane swizzles instead of ds_bpermute shuffles: four data-parallel-primitive moves fold each row of 16 lanes
// (quad swap, quad-pair swap, half-row mirror, row mirror -- afterwards every lane holds its row's total), then
// the four row totals are read out as scalars.  ~12 short VALU/SALU instructions against six LDS-pipeline round
// trips; every lane receives the result.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ---- counter-based dropout RNG ------------------------------------------------------------
// A dropout site is identified by (seed, counter) -- snapshotted on the device at forward time
// by gcgcn_rng_next, so a hipGraph replay draws fresh masks -- plus a per-site salt.  Element
// `idx` of the site keeps iff u32(key, idx) >= thresh, thresh = p * 2^32.  The backward
// kernels regenerate the same bits from the same snapshot; no mask is ever stored.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rng_key(uint64_t seed, uint64_t ctr, uint64_t salt) {
  return mix64(mix64(seed + 0x9E3779B97F4A7C15ull) ^ mix64(ctr * 0xD1342543DE82EF95ull + salt));
}
// Per-element draw: the 64-bit site key (full mix64 quality, computed once per kernel) seeds a 32-bit
// multiply-xorshift finaliser over the element index -- 4 integer multiplies per element instead of the ~12 a
// 64-bit mix costs on a 32-bit ALU; the draws sit in the epilogue of latency-bound kernels.
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t key, uint64_t idx) {
  uint32_t x = ((uint32_t)idx ^ (uint32_t)(key >> 32)) * 0x9E3779B1u + (uint32_t)key;
  x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

struct Drop {            // by-value kernel argument describing one dropout site
  const uint64_t* snap;  // device int64[2] {seed, counter}; nullptr = dropout off (eval)
  uint64_t salt;
  uint32_t thresh;       // keep iff rng >= thresh
  float scale;           // 1 / (1 - p)
};
__device__ __forceinline__ uint64_t drop_key(const Drop& d) {
  return rng_key(d.snap[0], d.snap[1], d.salt);
}
inline Drop make_drop(const void* snap, uint64_t salt, float p) {
  Drop d;
  d.snap = (p > 0.f) ? (const uint64_t*)snap : nullptr;
  d.salt = salt;
  d.thresh = drop_thresh(p);
  d.scale = (p < 1.f) ? 1.f / (1.f - p) : 0.f;
  return d;
}

// run-time A/B switch `name` (gcgcn_set_option > environment variable GCGCN_<NAME> > dflt); api.hip
int option(const char* name, int dflt);

// Two kinds of workgroup in one launch: `na` long-running ones (matrix tiles) and short ones (rows of a streaming pass).
// The tiles go out in COHORTS of `cohort` consecutive workgroups, one cohort every `stride` indices (stride >= cohort;
// stride == cohort: all tiles first).  A cohort starts together and runs in step, so its tiles find each other's operand
// panels in L2 (tiles started one by one between rows each fetch their own: measured at cfg 5, 2 208 single tiles spread
// through the edge pass took 4.0 ms against 2.6 ms tiles-first); between the cohorts every compute unit hosts rows next to
// its tile.  Returns true and the ordinal among the tiles, or false and the ordinal among the others.
struct Spread {
  int na, cohort, stride;
};
__host__ inline Spread make_spread(long na, long others, long cohort, long pct) {
  Spread s;
  s.na = (int)na, s.cohort = (int)std::max<long>(1, cohort), s.stride = s.cohort;
  const long nc = (na + s.cohort - 1) / s.cohort;
  if (nc > 0 && pct > 0) s.stride = (int)std::max<long>(s.cohort, std::min<long>(na + others, pct * (na + others) / 100) / nc);
  if (s.cohort % 8 == 0) s.stride &= ~7;   // a tile's workgroup index keeps its low bits: xcd_remap (gemm_body.hpp) relies on them
  return s;
}
__host__ __device__ __forceinline__ bool spread_pick(int x, const Spread& s, int& idx) {
  if (s.na <= 0 || s.cohort <= 0) {  // nothing long-running aboard (also a zero-initialised Spread)
    idx = x;
    return false;
  }
  const int nc = (s.na + s.cohort - 1) / s.cohort;
  if (nc == 0 || x >= nc * s.stride) {
    idx = x - s.na;
    return false;
  }
  const int c = x / s.stride, off = x - c * s.stride, left = s.na - c * s.cohort, size = left < s.cohort ? left : s.cohort;
  if (off < size) {
    idx = c * s.cohort + off;
    return true;
  }
  idx = x - (c * s.cohort + size);
  return false;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel timing (gcgcn_prof_start / gcgcn_prof_stop) ------------------------------
// While a filter is active, every launcher whose tag starts with the filter brackets its launch
// with two hipEvents on the launch stream; bench.py reads the summed duration.  Off by default
// (one predictable branch per launch); must stay off while a hipGraph is being captured.
// For the kernels that dominate the step: reserve an event pair for a launch tagged `tag` (false: not being timed) and
// pass it to hipExtLaunchKernelGGL, which stamps the kernel's own begin and end -- the same interval rocprofv3 reports.
// (An event recorded before / after a launch, as ProfScope does, also includes the dispatch gap: +3-4 us per launch.)
bool prof_events(const char* tag, double work, hipEvent_t* start, hipEvent_t* stop);
#define GC_LAUNCH_TIMED(tag, work, kernel, grid, block, lds, st, ...)                                     \
  do {                                                                                                    \
    hipEvent_t _e0, _e1;                                                                                  \
    if (gc::prof_events(tag, work, &_e0, &_e1))                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, _e0, _e1, 0, __VA_ARGS__);                      \
    else                                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                      \
  } while (0)

struct ProfScope {
  ProfScope(const char* tag, hipStream_t st, double work = 0.0);  // work: flops (GEMM) or algorithmic bytes (edge)
  ~ProfScope();
  int slot;
  hipStream_t st;
};

}  // namespace gc
