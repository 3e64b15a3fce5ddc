// Small row-wise kernels of the CAGGC/MAGGC path: softmax (+dropout) over attention rows,
// adjacency row normaliser, relu/normaliser backward, dropout, column reductions, the folded
// GAT projection, and the device-side dropout RNG state.  All are launch-latency sized; the
// big terms live in edge.hip (HBM) and gemm.hip (MFMA).
#include "gat_body.hpp"
#include "gemm.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int RW = 4;  // waves (= rows) per workgroup for the wave-per-row kernels

// ---------------------------------------------------------------------------------------------
// attention rows: P = softmax_j(scale * S[r, j] + coladd[doc, j]),  A = dropout(P)
//   rows r = ((doc * heads) + h) * N + i ; valid columns j < n_valid[doc]; padding rows -> 0
//   GATAttention glove:165-167 (coladd = u.x_j + c) and MultiHeadAttention glove:138-140.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * RW) void softmax_fwd_kernel(const float* __restrict__ S, const float* __restrict__ coladd,
                                                              const int* __restrict__ n_valid, float* __restrict__ P,
                                                              float* __restrict__ A, long rows, int N, int heads,
                                                              Drop drop) {
  const long r = (long)blockIdx.x * RW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const long doc = r / ((long)heads * N);
  const int i = (int)(r % N);
  const int nv = n_valid ? min(max(n_valid[doc], 0), N) : N;
  const float* s = S + r * N;
  float* p = P + r * N;
  float* a = A ? A + r * N : nullptr;
  if (i >= nv) {
    for (int j = lane; j < N; j += 64) {
      p[j] = 0.f;
      if (a) a[j] = 0.f;
    }
    return;
  }
  const float* ca = coladd ? coladd + doc * N : nullptr;
  float m = -INFINITY;
  for (int j = lane; j < nv; j += 64) m = fmaxf(m, s[j] + (ca ? ca[j] : 0.f));
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < nv; j += 64) sum += expf(s[j] + (ca ? ca[j] : 0.f) - m);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const bool dd = a && drop.snap;
  const uint64_t key = dd ? drop_key(drop) : 0;
  for (int j = lane; j < N; j += 64) {
    float v = 0.f;
    if (j < nv) v = expf(s[j] + (ca ? ca[j] : 0.f) - m) * inv;
    p[j] = v;
    if (a) {
      if (dd) v = (rng_u32(key, (uint64_t)(r * N + j)) >= drop.thresh) ? v * drop.scale : 0.f;
      a[j] = v;
    }
  }
}

// dS = P * (dP - sum_j dP * P),  dP = dropout_bwd(dA)
__global__ __launch_bounds__(64 * RW) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dA,
                                                              float* __restrict__ dS, long rows, int N, Drop drop) {
  const long r = (long)blockIdx.x * RW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + r * N;
  const float* da = dA + r * N;
  float* ds = dS + r * N;
  const bool dd = drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(drop) : 0;
  float dot = 0.f;
  for (int j = lane; j < N; j += 64) {
    float g = da[j];
    if (dd) g = (rng_u32(key, (uint64_t)(r * N + j)) >= drop.thresh) ? g * drop.scale : 0.f;
    dot = fmaf(g, p[j], dot);
  }
  dot = wave_sum(dot);
  for (int j = lane; j < N; j += 64) {
    float g = da[j];
    if (dd) g = (rng_u32(key, (uint64_t)(r * N + j)) >= drop.thresh) ? g * drop.scale : 0.f;
    ds[j] = p[j] * (g - dot);  // padding entries have p == 0
  }
}

// ---------------------------------------------------------------------------------------------
// GraphConv row normaliser (glove:47-50): rinv[r] = 1 / (sum_j A[r,j] + [sum == 0])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * RW) void rowsum_inv_kernel(const float* __restrict__ A, float* __restrict__ rinv,
                                                             long rows, int N) {
  const long r = (long)blockIdx.x * RW + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int j = lane; j < N; j += 64) s += A[r * N + j];
  s = wave_sum(s);
  if (lane == 0) rinv[r] = 1.f / (s + (s == 0.f ? 1.f : 0.f));
}

// ---------------------------------------------------------------------------------------------
// backward through  Y = relu(S),  S = M * rinv  for one sub-layer l of the stack:
//   dS = dY * [Y > 0];  dM = dS * rinv;  drow (grad of the row sum r) -= rinv * sum_c dS * Y
// Tensors are [rows_m, H, L, gh]; one wave per (m, h).  rinv / drow are [B, H, N].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * RW) void relu_norm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                const float* __restrict__ rinv, float* __restrict__ dM,
                                                                float* __restrict__ drow, long rows_m, int N, int H,
                                                                int L, int gh, int l, int first, int relu) {
  const long w = (long)blockIdx.x * RW + (threadIdx.x >> 6);
  if (w >= rows_m * H) return;
  const int lane = threadIdx.x & 63;
  const long m = w / H;
  const int h = (int)(w - m * H);
  const long b = m / N;
  const int i = (int)(m - b * N);
  const long off = (w * L + l) * gh;
  const long ri = (b * H + h) * N + i;
  const float rv = rinv[ri];
  float acc = 0.f;
  for (int c = lane; c < gh; c += 64) {
    const float y = Y[off + c];
    const float g = (!relu || y > 0.f) ? dY[off + c] : 0.f;
    dM[off + c] = g * rv;
    acc = fmaf(g, y, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float d = -rv * acc;
    drow[ri] = first ? d : drow[ri] + d;
  }
}

// ---------------------------------------------------------------------------------------------
// backward of  HO[m, h, :] = dropout(Y[m, h, :]) + X[m, :]  (glove:74-76 / 111-113):
//   dY[m, h, c] = dropout_bwd(dHO[m, h, c])   (may alias dHO)
//   dXres[m, c] = sum_h dHO[m, h, c]
// ---------------------------------------------------------------------------------------------
// Trailing workgroups (blockIdx >= main) finish a riding column sum (gemm.hpp ColRide, stage 2): out[c] = sum of the
// COL_RIDE_SLICES row-slice partials, in slice order.
__global__ __launch_bounds__(256) void head_sum_drop_bwd_kernel(const float* dHO, float* dY, float* __restrict__ dXres,
                                                                long M, int H, int D, Drop drop, int main_blocks,
                                                                const float* __restrict__ cpart, float* __restrict__ cout,
                                                                int cC, int vec) {
  if ((int)blockIdx.x >= main_blocks) {
    const int c = (blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (c < cC) {
      float s = 0.f;
      for (int q = 0; q < COL_RIDE_SLICES; ++q) s += cpart[(long)q * cC + c];
      cout[c] = s;
    }
    return;
  }
  const bool dd = drop.snap != nullptr;
  const uint64_t key = dd ? drop_key(drop) : 0;
  if (vec) {  // D % 4 == 0, 16-byte aligned: four columns per thread, the heads' loads in flight together
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= M * D) return;
    const long m = e / D;
    const int c = (int)(e - m * D);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h0 = 0; h0 < H; h0 += 4) {
      float4 g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const float4*>(dHO + (m * H + min(h0 + u, H - 1)) * D + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (h0 + u >= H) continue;
        const long o = (m * H + h0 + u) * D + c;
        s.x += g[u].x, s.y += g[u].y, s.z += g[u].z, s.w += g[u].w;
        if (dd) {
          float4 y;
          y.x = (rng_u32(key, (uint64_t)o) >= drop.thresh) ? g[u].x * drop.scale : 0.f;
          y.y = (rng_u32(key, (uint64_t)(o + 1)) >= drop.thresh) ? g[u].y * drop.scale : 0.f;
          y.z = (rng_u32(key, (uint64_t)(o + 2)) >= drop.thresh) ? g[u].z * drop.scale : 0.f;
          y.w = (rng_u32(key, (uint64_t)(o + 3)) >= drop.thresh) ? g[u].w * drop.scale : 0.f;
          *reinterpret_cast<float4*>(dY + o) = y;
        } else if (dY != dHO) {
          *reinterpret_cast<float4*>(dY + o) = g[u];
        }
      }
    }
    *reinterpret_cast<float4*>(dXres + e) = s;
    return;
  }
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * D) return;
  const long m = e / D;
  const int c = (int)(e - m * D);
  float s = 0.f;
  for (int h = 0; h < H; ++h) {
    const long o = (m * H + h) * D + c;
    const float g = dHO[o];
    s += g;
    if (dd) dY[o] = (rng_u32(key, (uint64_t)o) >= drop.thresh) ? g * drop.scale : 0.f;
    else if (dY != dHO) dY[o] = g;
  }
  dXres[e] = s;
}

// ---------------------------------------------------------------------------------------------
// elementwise dropout (hop glue, glove:341); the same kernel is its own backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, Drop drop) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const uint64_t key = drop_key(drop);
  y[e] = (rng_u32(key, (uint64_t)e) >= drop.thresh) ? x[e] * drop.scale : 0.f;
}

__global__ void dropout_keep_kernel(unsigned char* keep, long n, Drop drop) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  keep[e] = rng_u32(drop_key(drop), (uint64_t)e) >= drop.thresh;
}

// snaps[i] = {seed, counter + i} for i < count; counter += count   (one launch serves `count` dropout sites)
__global__ void rng_next_kernel(uint64_t* state, uint64_t* snaps, int count) {
  const uint64_t seed = state[0], ctr = state[1];
  for (int i = threadIdx.x; i < count; i += 64) {
    snaps[2 * i] = seed;
    snaps[2 * i + 1] = ctr + (uint64_t)i;
  }
  __syncthreads();
  if (threadIdx.x == 0) state[1] = ctr + (uint64_t)count;
}

// ---------------------------------------------------------------------------------------------
// weighted column sums:  out[z, c] (+)= sum_r w[z, r] * X[z, r, c]      (w optional)
// grid (ceil(C/64), nsplit, batch); each block covers a row range, 4 waves stride the rows, lanes
// own columns.  nsplit > 1 writes partials [batch, nsplit, C] that a second call reduces.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                     float* __restrict__ out, long R, int C, long ld, long sXz, long sWz,
                                                     long sOz, long rows_per_split, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int sp = blockIdx.y, z = blockIdx.z;
  const long r0 = sp * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > R) r1 = R;
  const float* x = X + z * sXz;
  const float* ww = w ? w + z * sWz : nullptr;
  // 8 rows in flight per lane: the loads of a batch are independent, one round trip instead of eight
  float acc = 0.f;
  if (c < C) {
    long r = r0 + wave;
    for (; r + 28 < r1; r += 32) {
      float v[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + 4 * u) * ld + c], wv[u] = ww ? ww[r + 4 * u] : 1.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], v[u], acc);
    }
    for (; r < r1; r += 4) acc = fmaf(ww ? ww[r] : 1.f, x[r * ld + c], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) {
    const float s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* o = out + z * sOz + (long)sp * C + c;
    *o = accumulate ? *o + s : s;
  }
}

// Several independent column reductions in one two-stage launch pair (GAT backward: du, dv, dc).
struct ColJobs {
  static constexpr int MAXJ = 4;
  const float* X[MAXJ];
  const float* w[MAXJ];
  float* out[MAXJ];
  long R[MAXJ], ld[MAXJ];
  int C[MAXJ];
  long part_off[MAXJ];  // offset of the job's partials inside scratch
  int n;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(const ColJobs j, float* __restrict__ scratch, int ns, int stage) {
  __shared__ float red[4][64];
  const int job = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = j.C[job];
  const int c = blockIdx.x * 64 + lane;
  if (blockIdx.x * 64 >= C) return;
  float acc = 0.f;
  if (stage == 0) {
    const long R = j.R[job], rps = (R + ns - 1) / ns;
    const long r0 = (long)blockIdx.y * rps;
    const long r1 = r0 + rps < R ? r0 + rps : R;
    const float* x = j.X[job];
    const float* w = j.w[job];
    const long ld = j.ld[job];
    if (c < C) {
      long r = r0 + wave;
      for (; r + 28 < r1; r += 32) {
        float v[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(r + 4 * u) * ld + c], wv[u] = w ? w[r + 4 * u] : 1.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], v[u], acc);
      }
      for (; r < r1; r += 4) acc = fmaf(w ? w[r] : 1.f, x[r * ld + c], acc);
    }
  } else {
    const float* x = scratch + j.part_off[job];
    if (c < C) {
      int r = wave;
      for (; r + 28 < ns; r += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(long)(r + 4 * u) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; r < ns; r += 4) acc += x[(long)r * C + c];
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) {
    const float s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (stage == 0) scratch[j.part_off[job] + (long)blockIdx.y * C + c] = s;
    else j.out[job][c] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// GATAttention folded (glove:156-162 has no non-linearity between the three
// Linear(D,D) and wt):  energy[i,j] = u.x_j + v.e_ij + c
//   u = W_h^T wt_h + W_t^T wt_t,  v = W_r^T wt_r,  c = wt_h.b_h + wt_t.b_t + wt_r.b_r + b
// D = att_input_dim (width of x and e), Dh = hidden_dim (rows of the three Linear layers, glove:148-151).
// flat parameter layout: [W_h Dh*D | b_h Dh | W_t Dh*D | b_t Dh | W_r Dh*D | b_r Dh | wt 3Dh | wtb 1]
// uvc output: [u D | v D | c 1]
// ---------------------------------------------------------------------------------------------
constexpr int FW = 16;  // waves per workgroup of the fold
constexpr int FC = 16;  // columns per workgroup: 64 row lanes per column share the Dh rows (Dh / 64 dependent steps, not Dh / 16)
// rng_state != NULL: workgroup 0 also advances the dropout generator (what rng_next_kernel does), so the first kernel
// of a hop loop serves every dropout site of the step without a launch of its own.
__global__ __launch_bounds__(64 * FW) void gat_fold_fwd_kernel(const float* __restrict__ flat, float* __restrict__ uvc, int D,
                                                               int Dh, uint64_t* rng_state, uint64_t* rng_snaps,
                                                               int rng_count) {
  __shared__ float red[2][64][FC + 1];
  __shared__ float redc[FW];
  if (rng_state && blockIdx.x == 0 && threadIdx.x < 64) {  // one wave: reads of the state precede its update in program order
    const uint64_t seed = rng_state[0], ctr = rng_state[1];
    for (int i = threadIdx.x; i < rng_count; i += 64) rng_snaps[2 * i] = seed, rng_snaps[2 * i + 1] = ctr + (uint64_t)i;
    if (threadIdx.x == 0) rng_state[1] = ctr + (uint64_t)rng_count;
  }
  const long DD = (long)Dh * D;
  const float* Wh = flat;
  const float* bh = Wh + DD;
  const float* Wt = bh + Dh;
  const float* bt = Wt + DD;
  const float* Wr = bt + Dh;
  const float* br = Wr + DD;
  const float* wt = br + Dh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = threadIdx.x & (FC - 1), rl = threadIdx.x >> 4;   // column of this workgroup's FC, row lane 0..63
  const int k = blockIdx.x * FC + cl;
  float au = 0.f, av = 0.f;
  if (k < D) {
#pragma unroll 4
    for (int d = rl; d < Dh; d += 64) {
      au = fmaf(Wh[(long)d * D + k], wt[d], au);
      au = fmaf(Wt[(long)d * D + k], wt[Dh + d], au);
      av = fmaf(Wr[(long)d * D + k], wt[2 * Dh + d], av);
    }
  }
  red[0][rl][cl] = au;
  red[1][rl][cl] = av;
  __syncthreads();
  if (threadIdx.x < 2 * FC) {   // thread (which, column): the 64 row-lane partials in order
    const int which = threadIdx.x / FC, c = threadIdx.x - which * FC, kk = blockIdx.x * FC + c;
    if (kk < D) {
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) sacc += red[which][r][c];
      uvc[which * D + kk] = sacc;
    }
  }
  if (blockIdx.x == 0) {
    float c = 0.f;
    for (int d = threadIdx.x; d < Dh; d += 64 * FW) c += wt[d] * bh[d] + wt[Dh + d] * bt[d] + wt[2 * Dh + d] * br[d];
    c = wave_sum(c);
    if (lane == 0) redc[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = wt[3 * Dh];
      for (int w = 0; w < FW; ++w) tot += redc[w];
      uvc[2 * D] = tot;
    }
  }
}

// backward of the fold; one wave per parameter row d.  duvc = [du D | dv D | dc 1].
// ns > 0: duvc is not final yet -- `part` holds ns row-slice partials of du, dv, dc (colsum3 stage 1, jobs at
// part_off[0..2]); every workgroup first sums them in slice order into LDS, which replaces the second-stage launch.
// (16 waves per workgroup: the slice sums are 32 (2 D + 1) dependent-latency loads per workgroup whatever its size -- at four
// waves a thread ran 24 batches of eight loads one after the other, 20 us at cfg 3; at sixteen, with sixteen loads in flight, four)
__global__ __launch_bounds__(64 * FW) void gat_fold_bwd_kernel(const float* __restrict__ flat, const float* __restrict__ duvc,
                                                               float* __restrict__ dflat, int D, int Dh,
                                                               const float* __restrict__ part, long off_dv, long off_dc,
                                                               int ns) {
  extern __shared__ float sduvc[];  // [2 D + 1] when ns > 0
  if (ns > 0) {
    for (int k = threadIdx.x; k < 2 * D + 1; k += 64 * FW) {
      const float* src = k < D ? part + k : (k < 2 * D ? part + off_dv + (k - D) : part + off_dc);
      const int C = k < 2 * D ? D : 1;
      float a = 0.f;
      int q = 0;
      for (; q + 16 <= ns; q += 16) {  // sixteen independent loads in flight, summed in slice order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(long)(q + u) * C];
#pragma unroll
        for (int u = 0; u < 16; ++u) a += v[u];
      }
      for (; q < ns; ++q) a += src[(long)q * C];
      sduvc[k] = a;
    }
    __syncthreads();
    duvc = sduvc;
  }
  const int d = blockIdx.x * FW + (threadIdx.x >> 6);
  if (d >= Dh) return;
  const int lane = threadIdx.x & 63;
  const long DD = (long)Dh * D;
  const long oWh = 0, obh = DD, oWt = obh + Dh, obt = oWt + DD, oWr = obt + Dh, obr = oWr + DD, owt = obr + Dh;
  const float* du = duvc;
  const float* dv = duvc + D;
  const float dc = duvc[2 * D];
  const float wh = flat[owt + d], wtt = flat[owt + Dh + d], wr = flat[owt + 2 * Dh + d];
  float ah = 0.f, at = 0.f, ar = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float u = du[k], v = dv[k];
    dflat[oWh + (long)d * D + k] = wh * u;
    dflat[oWt + (long)d * D + k] = wtt * u;
    dflat[oWr + (long)d * D + k] = wr * v;
    ah = fmaf(flat[oWh + (long)d * D + k], u, ah);
    at = fmaf(flat[oWt + (long)d * D + k], u, at);
    ar = fmaf(flat[oWr + (long)d * D + k], v, ar);
  }
  ah = wave_sum(ah), at = wave_sum(at), ar = wave_sum(ar);
  if (lane == 0) {
    dflat[obh + d] = wh * dc;
    dflat[obt + d] = wtt * dc;
    dflat[obr + d] = wr * dc;
    dflat[owt + d] = ah + flat[obh + d] * dc;
    dflat[owt + Dh + d] = at + flat[obt + d] * dc;
    dflat[owt + 2 * Dh + d] = ar + flat[obr + d] * dc;
    if (d == 0) dflat[owt + 3 * Dh] = dc;
  }
}

// GATAttention backward up to the edge pass as its own launch (gat_body.hpp): used when the edge pass cannot take the
// documents along as passengers.
__global__ __launch_bounds__(256) void gat_dlogit_kernel(const float* __restrict__ P, const float* __restrict__ dA,
                                                         const float* __restrict__ uvc, const float* __restrict__ dXin,
                                                         float* __restrict__ dlogit, float* __restrict__ ds,
                                                         float* __restrict__ dX, int N, int D, Drop drop) {
  __shared__ float sm[GAT_DOC_LDS];
  gat_dlogit_doc(P, dA, uvc, dXin, dlogit, ds, dX, N, D, drop, blockIdx.x, blockIdx.y, gridDim.y, sm);
}

// s[m] = u . X[m, :] + c      (one wave per node row)
// rng_state != NULL: workgroup 0 also advances the dropout generator (see gat_fold_fwd_kernel: used when the fold is skipped)
__global__ __launch_bounds__(64 * RW) void node_score_fwd_kernel(const float* __restrict__ X, const float* __restrict__ uvc,
                                                                 float* __restrict__ s, long M, int D, uint64_t* rng_state,
                                                                 uint64_t* rng_snaps, int rng_count) {
  if (rng_state && blockIdx.x == 0 && threadIdx.x < 64) {
    const uint64_t seed = rng_state[0], ctr = rng_state[1];
    for (int i = threadIdx.x; i < rng_count; i += 64) rng_snaps[2 * i] = seed, rng_snaps[2 * i + 1] = ctr + (uint64_t)i;
    if (threadIdx.x == 0) rng_state[1] = ctr + (uint64_t)rng_count;
  }
  const long m = (long)blockIdx.x * RW + (threadIdx.x >> 6);
  if (m >= M) return;
  const int lane = threadIdx.x & 63;
  float a = 0.f;
  // (16-byte loads with four row segments in flight were measured: -3 us at cfg 3, -15 us at cfg 5 -- and a different
  // summation order, which moved one relu decision of the N = 256 parity case; not worth a new fixture seed)
  for (int k = lane; k < D; k += 64) a = fmaf(X[m * D + k], uvc[k], a);
  a = wave_sum(a);
  if (lane == 0) s[m] = a + uvc[2 * D];
}

// dX[m, :] = ds[m] * u (+ dXin[m, :], the gradient X already collected downstream)
__global__ __launch_bounds__(256) void node_score_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ uvc,
                                                             const float* __restrict__ dXin, float* __restrict__ dX, long M,
                                                             int D) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * D) return;
  const long m = e / D;
  const float v = ds[m] * uvc[e - m * D];
  dX[e] = dXin ? v + dXin[e] : v;
}

// y[m, :] = dropout_bwd(x[m, :]) for real entities, 0 for padding rows (m = b * N + i, i >= n_valid[b]);
// n_valid == NULL: no padding, drop.snap == NULL: no dropout
// Trailing workgroups (blockIdx >= main_blocks): wsum[k, c] = sum_h wlin[k, h, c] for the fused chain backward (chain.hip).
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long M, int D,
                                                        int N, const int* __restrict__ n_valid, Drop drop, int main_blocks,
                                                        const float* __restrict__ wlin, float* __restrict__ wsum, int H, int vec) {
  if ((int)blockIdx.x >= main_blocks) {
    const int e = (blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (e < D * D) {
      const int k = e / D, c = e - k * D;
      float s = 0.f;
      for (int h0 = 0; h0 < H; h0 += 8) {  // eight heads in flight, summed in head order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = wlin[((long)k * H + min(h0 + u, H - 1)) * D + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (h0 + u < H) ? v[u] : 0.f;
      }
      wsum[e] = s;
    }
    return;
  }
  if (vec) {  // D % 4 == 0, 16-byte aligned: four elements of one row per thread
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= M * D) return;
    const long m = e / D;
    const long b = m / N;
    float4 v = *reinterpret_cast<const float4*>(x + e);
    if (n_valid && (int)(m - b * N) >= n_valid[b]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (drop.snap) {
      const uint64_t key = drop_key(drop);
      v.x = (rng_u32(key, (uint64_t)e) >= drop.thresh) ? v.x * drop.scale : 0.f;
      v.y = (rng_u32(key, (uint64_t)(e + 1)) >= drop.thresh) ? v.y * drop.scale : 0.f;
      v.z = (rng_u32(key, (uint64_t)(e + 2)) >= drop.thresh) ? v.z * drop.scale : 0.f;
      v.w = (rng_u32(key, (uint64_t)(e + 3)) >= drop.thresh) ? v.w * drop.scale : 0.f;
    }
    *reinterpret_cast<float4*>(y + e) = v;
    return;
  }
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * D) return;
  const long m = e / D;
  const long b = m / N;
  float v = (!n_valid || (int)(m - b * N) < n_valid[b]) ? x[e] : 0.f;
  if (drop.snap) v = (rng_u32(drop_key(drop), (uint64_t)e) >= drop.thresh) ? v * drop.scale : 0.f;
  y[e] = v;
}

// =============================================================================================
// launchers
// =============================================================================================
// ---- row blocks of a ragged batch ----------------------------------------------------------------------------------------
// One workgroup: per-document live-block counts, exclusive scan over the documents, then every document writes its blocks
// (live list and dead list, each ascending).  B N / 16 is a few hundred entries; this is launch latency, nothing else.
//   out = { live blocks, 0, 0, 0 | blocks: live ascending, then dead ascending }
// Row-dimension products take the live list four blocks at a time (a 64-row tile), weight gradients two at a time (a k-tile).
__global__ __launch_bounds__(256) void row_blocks_kernel(const int* __restrict__ n_valid, int B, int N, int* __restrict__ out) {
  __shared__ int part[256];
  __shared__ int total;
  const int t = threadIdx.x, chunk = (B + 255) / 256, per = N / 16;
  int s = 0;
  for (int b = t * chunk; b < min(B, (t + 1) * chunk); ++b) s += (min(max(n_valid[b], 0), N) + 15) >> 4;
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < 256; ++k) {
      const int v = part[k];
      part[k] = run;
      run += v;
    }
    total = run;
    out[0] = run, out[1] = 0, out[2] = 0, out[3] = 0;
  }
  __syncthreads();
  const int total_live = total;
  int live_at = part[t];
  for (int b = t * chunk; b < min(B, (t + 1) * chunk); ++b) {
    const int nl = (min(max(n_valid[b], 0), N) + 15) >> 4;
    int dead_at = total_live + b * per - live_at;      // dead blocks of the documents before b: b per - live_at
    for (int r = 0; r < per; ++r) {
      if (r < nl) out[ROWBLK_HDR + live_at++] = b * per + r;
      else out[ROWBLK_HDR + dead_at++] = b * per + r;
    }
  }
}
int row_blocks(const int* n_valid, int B, int N, int* out, hipStream_t st) {
  GC_REQUIRE(n_valid && out && B > 0 && N > 0 && N % 16 == 0, "row_blocks: needs n_valid and N a multiple of 16 (N = %d)", N);
  hipLaunchKernelGGL(row_blocks_kernel, dim3(1), dim3(256), 0, st, n_valid, B, N, out);
  return check_launch("row_blocks");
}

int softmax_fwd(const float* S, const float* coladd, const int* n_valid, float* P, float* A, long rows, int N, int heads,
                Drop drop, hipStream_t st) {
  if (rows == 0) return 0;
  ProfScope ps("softmax_fwd", st);
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, RW)), dim3(64 * RW), 0, st, S, coladd, n_valid, P, A, rows, N,
                     heads, drop);
  return check_launch("softmax_fwd");
}
int softmax_bwd(const float* P, const float* dA, float* dS, long rows, int N, Drop drop, hipStream_t st) {
  if (rows == 0) return 0;
  ProfScope ps("softmax_bwd", st);
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, RW)), dim3(64 * RW), 0, st, P, dA, dS, rows, N, drop);
  return check_launch("softmax_bwd");
}
int rowsum_inv(const float* A, float* rinv, long rows, int N, hipStream_t st) {
  if (rows == 0) return 0;
  ProfScope ps("rowsum_inv", st);
  hipLaunchKernelGGL(rowsum_inv_kernel, dim3(cdiv(rows, RW)), dim3(64 * RW), 0, st, A, rinv, rows, N);
  return check_launch("rowsum_inv");
}
int relu_norm_bwd(const float* dY, const float* Y, const float* rinv, float* dM, float* drow, long rows_m, int N, int H,
                  int L, int gh, int l, int first, hipStream_t st, int relu) {
  ProfScope ps("relu_norm_bwd", st);
  hipLaunchKernelGGL(relu_norm_bwd_kernel, dim3(cdiv(rows_m * H, RW)), dim3(64 * RW), 0, st, dY, Y, rinv, dM, drow,
                     rows_m, N, H, L, gh, l, first, relu);
  return check_launch("relu_norm_bwd");
}
int head_sum_drop_bwd(const float* dHO, float* dY, float* dXres, long M, int H, int D, Drop drop, hipStream_t st,
                      const ColRide* finish) {
  ProfScope ps("head_sum_drop_bwd", st);
  const int vec = D % 4 == 0 && ((((uintptr_t)dHO) | ((uintptr_t)dY) | ((uintptr_t)dXres)) & 15) == 0;
  const int main_blocks = cdiv(vec ? M * D / 4 : M * D, 256), extra = finish ? cdiv(finish->C, 256) : 0;
  hipLaunchKernelGGL(head_sum_drop_bwd_kernel, dim3(main_blocks + extra), dim3(256), 0, st, dHO, dY, dXres, M, H, D, drop,
                     main_blocks, finish ? finish->part : nullptr, finish ? finish->out : nullptr, finish ? finish->C : 0, vec);
  return check_launch("head_sum_drop_bwd");
}
int dropout(const float* x, float* y, long n, Drop drop, hipStream_t st) {
  GC_REQUIRE(drop.snap, "dropout: no rng snapshot");
  if (n == 0) return 0;
  ProfScope ps("dropout", st);
  hipLaunchKernelGGL(dropout_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, y, n, drop);
  return check_launch("dropout");
}
int dropout_keep(unsigned char* keep, long n, Drop drop, hipStream_t st) {
  GC_REQUIRE(drop.snap, "dropout_keep: no rng snapshot");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_keep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, keep, n, drop);
  return check_launch("dropout_keep");
}
int rng_next(void* state, void* snaps, int count, hipStream_t st) {
  hipLaunchKernelGGL(rng_next_kernel, dim3(1), dim3(64), 0, st, (uint64_t*)state, (uint64_t*)snaps, count);
  return check_launch("rng_next");
}

// out[z, :] (+)= sum_r w[z,r] X[z, r, :].  `scratch` must hold batch * COLSUM_SPLITS * C floats when
// R is large enough to be split (see colsum_scratch_elems).
constexpr int COLSUM_MIN_ROWS = 32;
constexpr int COLSUM_MAX_SPLITS = 128;
long colsum_scratch_elems(long R, int C, int batch) {
  if (R < 2 * COLSUM_MIN_ROWS) return 0;
  long ns = R / COLSUM_MIN_ROWS;
  if (ns > COLSUM_MAX_SPLITS) ns = COLSUM_MAX_SPLITS;
  return ns * C * batch;
}
int colsum(const float* X, const float* w, float* out, long R, int C, long ld, int batch, long sXz, long sWz, long sOz,
           int accumulate, float* scratch, hipStream_t st) {
  if (C == 0 || batch == 0) return 0;
  long ns = 1;
  if (R >= 2 * COLSUM_MIN_ROWS && scratch) {
    ns = R / COLSUM_MIN_ROWS;
    if (ns > COLSUM_MAX_SPLITS) ns = COLSUM_MAX_SPLITS;
    // enough row splits to put ~512 workgroups on the chip, no more
    const long want = 512 / ((long)cdiv(C, 64) * batch) + 1;
    if (ns > want) ns = want;
    if (ns < 2) ns = 1;
  }
  ProfScope ps("colsum", st);
  if (ns == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), 1, batch), dim3(256), 0, st, X, w, out, R, C, ld, sXz, sWz, sOz,
                       R > 0 ? R : 1, accumulate);
    return check_launch("colsum");
  }
  const long rps = (R + ns - 1) / ns;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), (unsigned)ns, batch), dim3(256), 0, st, X, w, scratch, R, C, ld, sXz,
                     sWz, ns * C, rps, 0);
  if (int e = check_launch("colsum/1")) return e;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), 1, batch), dim3(256), 0, st, scratch, (const float*)nullptr, out, ns, C,
                     (long)C, ns * C, 0L, sOz, ns, accumulate);
  return check_launch("colsum/2");
}

// Three column sums in two launches; scratch needs 3 * 64 * maxC floats.  stage2 = false: only the row-slice
// partials are produced (job i at scratch + part_off[i], [ns][C_i]); the consumer sums them (gat_fold_bwd).
int colsum3(const float* X0, const float* w0, float* o0, long R0, int C0, long ld0, const float* X1, const float* w1,
            float* o1, long R1, int C1, long ld1, const float* X2, const float* w2, float* o2, long R2, int C2, long ld2,
            float* scratch, hipStream_t st, bool stage2, long* part_off, int* ns_out) {
  ColJobs j;
  j.n = 3;
  const float* Xs[3] = {X0, X1, X2};
  const float* ws[3] = {w0, w1, w2};
  float* os[3] = {o0, o1, o2};
  const long Rs[3] = {R0, R1, R2}, lds[3] = {ld0, ld1, ld2};
  const int Cs[3] = {C0, C1, C2};
  // fewer slices when every consumer workgroup re-sums them (option fold_slices, at most 64: the callers' scratch is sized for that)
  const int ns = stage2 ? 64 : std::min(64, std::max(1, option("fold_slices", 32)));
  long off = 0;
  int maxC = 1;
  for (int i = 0; i < 3; ++i) {
    j.X[i] = Xs[i], j.w[i] = ws[i], j.out[i] = os[i], j.R[i] = Rs[i], j.ld[i] = lds[i], j.C[i] = Cs[i];
    j.part_off[i] = off;
    if (part_off) part_off[i] = off;
    off += (long)ns * Cs[i];
    if (Cs[i] > maxC) maxC = Cs[i];
  }
  if (ns_out) *ns_out = ns;
  {
    ProfScope ps("colsum", st);
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(cdiv(maxC, 64), ns, 3), dim3(256), 0, st, j, scratch, ns, 0);
  }
  if (int e = check_launch("colsum3/1")) return e;
  if (!stage2) return 0;
  ProfScope ps("colsum", st);
  hipLaunchKernelGGL(colsum_multi_kernel, dim3(cdiv(maxC, 64), 1, 3), dim3(256), 0, st, j, scratch, ns, 1);
  return check_launch("colsum3/2");
}

int gat_fold_fwd(const float* flat, float* uvc, int D, int Dh, hipStream_t st, void* rng_state, void* rng_snaps,
                 int rng_count) {
  ProfScope ps("gat_fold_fwd", st);
  hipLaunchKernelGGL(gat_fold_fwd_kernel, dim3(cdiv(D, FC)), dim3(64 * FW), 0, st, flat, uvc, D, Dh, (uint64_t*)rng_state,
                     (uint64_t*)rng_snaps, rng_count);
  return check_launch("gat_fold_fwd");
}
int gat_fold_bwd(const float* flat, const float* duvc, float* dflat, int D, int Dh, hipStream_t st, const float* part,
                 const long* part_off, int ns) {
  ProfScope ps("gat_fold_bwd", st);
  const size_t lds = ns > 0 ? sizeof(float) * (2 * (size_t)D + 1) : 0;
  hipLaunchKernelGGL(gat_fold_bwd_kernel, dim3(cdiv(Dh, FW)), dim3(64 * FW), lds, st, flat, duvc, dflat, D, Dh, part,
                     ns > 0 ? part_off[1] : 0L, ns > 0 ? part_off[2] : 0L, ns);
  return check_launch("gat_fold_bwd");
}
bool gat_dlogit_ok(int N) { return N <= GT; }
int gat_dlogit_slices(int D) { return D >= 256 ? 8 : (D >= 64 ? 4 : 1); }
int gat_dlogit(const float* P, const float* dA, const float* uvc, const float* dXin, float* dlogit, float* ds, float* dX,
               int B, int N, int D, Drop drop, hipStream_t st) {
  ProfScope ps("gat_dlogit", st);
  const int slices = gat_dlogit_slices(D);
  hipLaunchKernelGGL(gat_dlogit_kernel, dim3(B, slices), dim3(256), 0, st, P, dA, uvc, dXin, dlogit, ds, dX, N, D, drop);
  return check_launch("gat_dlogit");
}
int node_score_fwd(const float* X, const float* uvc, float* s, long M, int D, hipStream_t st, void* rng_state, void* rng_snaps,
                   int rng_count) {
  ProfScope ps("node_score_fwd", st);
  hipLaunchKernelGGL(node_score_fwd_kernel, dim3(cdiv(M, RW)), dim3(64 * RW), 0, st, X, uvc, s, M, D, (uint64_t*)rng_state,
                     (uint64_t*)rng_snaps, rng_count);
  return check_launch("node_score_fwd");
}
int node_score_bwd(const float* ds, const float* uvc, const float* dXin, float* dX, long M, int D, hipStream_t st) {
  ProfScope ps("node_score_bwd", st);
  hipLaunchKernelGGL(node_score_bwd_kernel, dim3(cdiv(M * D, 256)), dim3(256), 0, st, ds, uvc, dXin, dX, M, D);
  return check_launch("node_score_bwd");
}
int mask_rows(const float* x, float* y, long M, int D, int N, const int* n_valid, Drop drop, hipStream_t st, const float* wlin,
              float* wsum, int H) {
  ProfScope ps("mask_rows", st);
  const int vec = x && D % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  const int main_blocks = x ? cdiv(vec ? M * D / 4 : M * D, 256) : 0, extra = wsum ? cdiv((long)D * D, 256) : 0;
  if (main_blocks + extra == 0) return 0;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(main_blocks + extra), dim3(256), 0, st, x, y, M, D, N, n_valid, drop, main_blocks,
                     wlin, wsum, H, vec);
  return check_launch("mask_rows");
}

}  // namespace gc
