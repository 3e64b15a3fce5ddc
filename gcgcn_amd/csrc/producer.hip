// Edge-feature producer: WordAttention + SentenceAttention + the two Linear layers that build
// E = context_sent_att[N, N, Hd] for one hop (GCGCN_glove.py:171-214 and the call sequence :300-330).
//
// The reference materialises [N, N, S, T, Hd] tensors (token states broadcast over every entity pair and sentence
// slot): 2.3 GB per document at N = 42, S = 5, T = 512, Hd = 128 -- the reason its batch size is 1.  Nothing of that
// size exists here:
//   * the word score tanh(W_s ctx_t + W_p dis_k) . w_a depends only on (token t, distance id k): a [ND, T] table per
//     document, gathered by the position matrices;
//   * a sentence slot contributes to E iff token 0 belongs to it (sent_att_padding_matrix = ~sen_matrix[..., 0:1],
//     glove:305: every other slot is filled with -100000 and relu'd to exactly zero, glove:208-211, forward and
//     backward) -- only those LIVE slots are computed.  They are compacted into rows (deterministic order); all
//     per-slot products are GEMMs over the compact rows whose count stays on the device (gemm_dyn);
//   * the attended word context of a live slot is a softmax-weighted sum of the token states over the slot's token
//     range only (masked tokens have weight exp(-100000 - max) == 0 exactly).
// The reference's divisor quirk is kept: the sentence-level sum is divided by the number of PADDED slots + 1e-10
// (glove:205, 212).
//
// Per-slot / per-pair arithmetic runs in wave-per-row kernels (lanes over the Hd feature columns, token states read
// from L2).  Backward sums shared by many slots are owner-computed where the owner is cheap to find (token-state and
// score-table gradients: one workgroup per (document, token) scans the document's live rows); the per-entity node-term
// gradient and a few parameter-sized partials still use fp32 atomics (reproducible up to summation order).
#include <string.h>

#include "gemm.hpp"
#include "rowops.hpp"

namespace gc {

constexpr int PW = 4;      // waves per workgroup of the row kernels
constexpr int PGRID = 2048;  // persistent grid of the row kernels (8 workgroups per CU)
constexpr int HCLIM = 8;   // Hd <= 64 * HCLIM

struct ProdIdx {           // compact index of the live slots / pairs (device memory, int32)
  int* doc_counts;         // [B][2] live slots, live pairs per document
  int* pair_bits;          // [B*N*N] bit s set: slot s of the pair is live
  int* pair_row0;          // [B*N*N] first compact row of the pair (its live slots are consecutive rows)
  int* pair_prow;          // [B*N*N] compact pair index, -1: no live slot
  float* pair_div;         // [B*N*N] padded slots + 1e-10 (glove:205, 212); 0 for padding entities
  int* row_slot;           // [cap_rows] flat slot index ((b*N + i)*N + j)*S + s of a compact row
  int* prow_pair;          // [cap_pairs] flat pair index of a compact pair
  int* counts;             // [4] rows, pairs, overflow flag, unused
  int* doc_off;            // [B + 1] first compact row of each document (its live rows are consecutive)
  int* row_rng;            // [cap_rows] token range of the row's slot, t0 | t1 << 16 (written by the forward's word kernel)
  int* tmax;               // [B] one past the last token any live slot of the document covers
};

__device__ __forceinline__ int pos_at(const void* pos, int pos_bytes, long idx) {
  return pos_bytes == 8 ? (int)((const long long*)pos)[idx] : (pos_bytes == 4 ? ((const int*)pos)[idx] : (int)((const unsigned char*)pos)[idx]);
}

// ---- index: pass A, one workgroup per document -------------------------------------------------------------------
// (Round 5: one workgroup per document is 32 workgroups of pure latency -- per 256 pairs it ran S dependent byte loads and a
// 16-barrier Hillis-Steele scan, 28 us per call and two calls per step of the model.  Now the liveness bytes of eight chunks of
// pairs are requested together, and the scan is a wave prefix (shuffles) plus one hand-over between the four waves: two
// barriers per chunk.  Same integers out.)
__global__ __launch_bounds__(256) void prod_index_a_kernel(const unsigned char* __restrict__ sen, const int* __restrict__ n_valid,
                                                           ProdIdx ix, int N, int S, int T) {
  __shared__ int wsum[2][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const int NN = N * N;
  auto wave_incl = [&](int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(v, d, 64);
      if (lane >= d) v += t;
    }
    return v;
  };
  int run_s = 0, run_p = 0;
  constexpr int CH = 8;   // chunks of 256 pairs whose loads are in flight together
  for (int base0 = 0; base0 < NN; base0 += CH * 256) {
    int bitsv[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int p = base0 + ch * 256 + tid;
      int bits = 0;
      if (p < NN) {
        const int i = p / N, j = p - i * N;
        if (i < nv && j < nv) {
          const long s0 = ((long)b * NN + p) * S;
          for (int sb = 0; sb < S; sb += 8) {
            unsigned char v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sen[(s0 + min(sb + u, S - 1)) * T];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (sb + u < S && v[u] != 0) bits |= 1 << (sb + u);
          }
        }
      }
      bitsv[ch] = bits;
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int base = base0 + ch * 256;
      if (base >= NN) break;                       // (uniform)
      const int p = base + tid, bits = bitsv[ch];
      const int c = __popc(bits), live = c > 0;
      const int ic = wave_incl(c), il = wave_incl(live);
      if (lane == 63) wsum[0][wave] = ic, wsum[1][wave] = il;
      __syncthreads();
      int offc = 0, offl = 0, totc = 0, totl = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int a_ = wsum[0][w], q_ = wsum[1][w];
        if (w < wave) offc += a_, offl += q_;
        totc += a_, totl += q_;
      }
      if (p < NN) {
        const long pp = (long)b * NN + p;
        ix.pair_bits[pp] = bits;
        ix.pair_row0[pp] = run_s + offc + ic - c;          // local to the document; pass B adds the document's offset
        ix.pair_prow[pp] = live ? run_p + offl + il - 1 : -1;
      }
      run_s += totc, run_p += totl;
      __syncthreads();
    }
  }
  if (tid == 0) ix.doc_counts[2 * b] = run_s, ix.doc_counts[2 * b + 1] = run_p;
}

// ---- index: pass B, grid (ceil(N*N / 256), B): global rows + inverse maps ------------------------------------------
__global__ __launch_bounds__(256) void prod_index_b_kernel(const int* __restrict__ n_valid, ProdIdx ix, int B, int N, int S,
                                                           int cap_rows, int cap_pairs) {
  __shared__ int off[2], red[2][256];
  const int b = blockIdx.y, tid = threadIdx.x, NN = N * N;
  int a = 0, q = 0, ta = 0, tq = 0;
  for (int d = tid; d < B; d += 256) {
    const int cs = ix.doc_counts[2 * d], cp = ix.doc_counts[2 * d + 1];
    if (d < b) a += cs, q += cp;
    ta += cs, tq += cp;
  }
  red[0][tid] = a, red[1][tid] = q;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) red[0][tid] += red[0][tid + d], red[1][tid] += red[1][tid + d];
    __syncthreads();
  }
  if (tid == 0) off[0] = red[0][0], off[1] = red[1][0];
  __syncthreads();
  red[0][tid] = ta, red[1][tid] = tq;   // totals of the batch (every thread holds a strided share)
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) red[0][tid] += red[0][tid + d], red[1][tid] += red[1][tid + d];
    __syncthreads();
  }
  // Over capacity: nothing is computed at all (every pair counts as dead, E = bias) rather than a partly filled index,
  // and the flag reports it.
  const bool over = red[0][0] > cap_rows || red[1][0] > cap_pairs;
  if (blockIdx.x == 0 && b == 0 && tid == 0)
    ix.counts[0] = over ? 0 : red[0][0], ix.counts[1] = over ? 0 : red[1][0], ix.counts[2] = over, ix.counts[3] = 0;
  if (blockIdx.x == 0 && tid == 0) {
    ix.doc_off[b] = over ? 0 : off[0];
    if (b == B - 1) ix.doc_off[B] = over ? 0 : red[0][0];
  }
  const int p = blockIdx.x * 256 + tid;
  if (p >= NN) return;
  const long pp = (long)b * NN + p;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const int i = p / N, j = p - i * N;
  const int bits = ix.pair_bits[pp];
  const int row0 = off[0] + ix.pair_row0[pp];
  int prow = ix.pair_prow[pp];
  if (prow >= 0) prow += off[1];
  const int n = __popc(bits);
  if (over) {
    ix.pair_bits[pp] = 0, ix.pair_prow[pp] = -1, ix.pair_row0[pp] = 0;
    ix.pair_div[pp] = (i < nv && j < nv) ? (float)S + 1e-10f : 0.f;
    return;
  }
  ix.pair_row0[pp] = row0;
  ix.pair_prow[pp] = prow;
  ix.pair_div[pp] = (i < nv && j < nv) ? (float)(S - n) + 1e-10f : 0.f;
  if (prow >= 0) ix.prow_pair[prow] = (int)pp;
  int k = 0;
  for (int s = 0; s < S; ++s)
    if (bits >> s & 1) ix.row_slot[row0 + k++] = (int)(pp * S + s);
}

// ---- word score table: table[b, k, t] = w_a . tanh(sentF[b, t, :] + disF[k, :]) + b_a   (glove:178-182 folded) ------
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_table_fwd_kernel(const float* __restrict__ sentF, const float* __restrict__ disF,
                                                                 const float* __restrict__ wa,
                                                                 const float* __restrict__ ba, float* __restrict__ table,
                                                                 long BT, int T, int Hd, int ND) {
  const long bt = (long)blockIdx.x * PW + (threadIdx.x >> 6);
  if (bt >= BT) return;
  const int lane = threadIdx.x & 63;
  const long b = bt / T;
  const int t = (int)(bt - b * T);
  float sf[HC], w[HC];
#pragma unroll
  for (int cc = 0; cc < HC; ++cc) {
    const int c = lane + 64 * cc;
    sf[cc] = c < Hd ? sentF[bt * Hd + c] : 0.f;
    w[cc] = c < Hd ? wa[c] : 0.f;
  }
  const float bias = ba[0];
  for (int k = 0; k < ND; ++k) {
    float a = 0.f;
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) {
      const int c = lane + 64 * cc;
      if (c < Hd) a = fmaf(w[cc], tanhf(sf[cc] + disF[(long)k * Hd + c]), a);
    }
    a = wave_sum(a);
    if (lane == 0) table[(b * ND + k) * T + t] = a + bias;
  }
}

// Gathered scores of one (slot, side) for the wave's tokens t = c0 + 64 u + lane, u < 8: the mask byte, the position id and
// the table entry of all eight chunks are requested before the first one is used (clamped addresses, not guarded loads: a
// load inside a branch is waited for at the branch's end).  sc[u] = -inf for masked / out-of-range tokens.
__device__ __forceinline__ void slot_scores8(const unsigned char* __restrict__ m, const void* __restrict__ pos, int pos_bytes,
                                             long slot_base, const float* __restrict__ tab, int T, int ND, int c0, int lane,
                                             float (&sc)[8]) {
  unsigned char mm[8];
  int kk[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int tc = min(c0 + 64 * u + lane, T - 1);
    mm[u] = m[tc];
    kk[u] = pos_at(pos, pos_bytes, slot_base + tc);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int tc = min(c0 + 64 * u + lane, T - 1);
    sc[u] = tab[(long)min(max(kk[u], 0), ND - 1) * T + tc];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (c0 + 64 * u + lane >= T || !mm[u]) sc[u] = -INFINITY;
}

// ---- word attention of the live slots (glove:184-187): CW[row, side*Hd + c] = sum_t softmax_t(score)[t] ctx[b, t, c] -
// dynamic LDS: PW * T floats (the un-normalised weights of the wave's slot)
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_word_fwd_kernel(const float* __restrict__ ctx, const unsigned char* __restrict__ sen,
                                                                const void* __restrict__ pos_h, const void* __restrict__ pos_t,
                                                                int pos_bytes, const float* __restrict__ table, ProdIdx ix,
                                                                float* __restrict__ CW, float* __restrict__ stats, int N, int S,
                                                                int T, int Hd, int ND) {
  extern __shared__ float wsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ev = wsm + (long)wave * T;
  const int nrows = ix.counts[0], total = 2 * nrows;
  const long slots_per_doc = (long)N * N * S;
  for (int r0 = blockIdx.x * PW; r0 < total; r0 += gridDim.x * PW) {
    const int r = r0 + wave;
    const bool on = r < total;
    int t0 = T, t1 = 0;
    float inv = 0.f;
    long b = 0;
    if (on) {
      const int row = r >> 1, side = r & 1;
      const long slot = ix.row_slot[row];
      b = slot / slots_per_doc;
      const unsigned char* m = sen + slot * T;
      const void* pos = side ? pos_t : pos_h;
      const float* tab = table + b * ND * T;
      float mx = -INFINITY;
      for (int c0 = 0; c0 < T; c0 += 512) {
        float sc[8];
        slot_scores8(m, pos, pos_bytes, slot * T, tab, T, ND, c0, lane, sc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = c0 + 64 * u + lane;
          if (t < T) {
            ev[t] = sc[u];
            if (sc[u] != -INFINITY) t0 = min(t0, t), t1 = max(t1, t + 1);
            mx = fmaxf(mx, sc[u]);
          }
        }
      }
      mx = wave_max(mx);
      t0 = -(int)wave_max((float)-t0), t1 = (int)wave_max((float)t1);   // exact for |t| < 2^24
      float sum = 0.f;
      for (int t = lane; t < T; t += 64) {
        const float e = ev[t] == -INFINITY ? 0.f : expf(ev[t] - mx);   // masked tokens: exp(-100000 - max) == 0 in fp32
        ev[t] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      inv = 1.f / sum;
      if (lane == 0) {
        stats[2 * (long)r] = mx, stats[2 * (long)r + 1] = inv;
        if (side == 0) ix.row_rng[row] = t0 | (t1 << 16);   // the slot's token range (T <= 2048): read again by the backward
      }
    }
    __syncthreads();  // ev[] written lane-wise, read by every lane below (uniform trip count: every wave gets here)
    if (on) {
      const int row = r >> 1, side = r & 1;
      const float* cb = ctx + b * T * Hd;
      float acc[HC];
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) acc[cc] = 0.f;
      int t = t0;
      for (; t + 3 < t1; t += 4) {  // four token rows in flight
        float x[4][HC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int cc = 0; cc < HC; ++cc) {
            const int c = lane + 64 * cc;
            x[u][cc] = c < Hd ? cb[(long)(t + u) * Hd + c] : 0.f;
          }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float e = ev[t + u];
#pragma unroll
          for (int cc = 0; cc < HC; ++cc) acc[cc] = fmaf(e, x[u][cc], acc[cc]);
        }
      }
      for (; t < t1; ++t) {
        const float e = ev[t];
#pragma unroll
        for (int cc = 0; cc < HC; ++cc) {
          const int c = lane + 64 * cc;
          if (c < Hd) acc[cc] = fmaf(e, cb[(long)t * Hd + c], acc[cc]);
        }
      }
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        if (c < Hd) CW[(long)row * 2 * Hd + side * Hd + c] = acc[cc] * inv;
      }
    }
    __syncthreads();  // before the next slot overwrites ev[]
  }
  // rows [nrows, roundup64(nrows)) feed the K-dynamic GEMMs as zeros
  if (blockIdx.x == 0) {
    const int hi = (nrows + 63) & ~63;
    for (long e = (long)nrows * 2 * Hd + threadIdx.x; e < (long)hi * 2 * Hd; e += 64 * PW) CW[e] = 0.f;
  }
}

// ---- tmax[b] = one past the last token any live slot of document b covers (bounds the backward's per-token work) -------
__global__ __launch_bounds__(256) void prod_tmax_kernel(ProdIdx ix) {
  __shared__ int red[4];
  const int b = blockIdx.x;
  int m = 0;
  for (int row = ix.doc_off[b] + threadIdx.x; row < ix.doc_off[b + 1]; row += 256) m = max(m, ix.row_rng[row] >> 16);
  m = (int)wave_max((float)m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) ix.tmax[b] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// ---- sentence attention of the live pairs (glove:201-212) ----------------------------------------------------------
//   score_h[s] = w . tanh(sfeat[row] + nterm[b, j]) + c,  score_t[s] uses nterm[b, i];  att = relu(score)
//   CS[prow] = [ sum_s att_h[s] cwa[row] / div | sum_s att_t[s] cwa[row] / div ]
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_sent_fwd_kernel(const float* __restrict__ sfeat, const float* __restrict__ cwa,
                                                                const float* __restrict__ nterm, const float* __restrict__ wsa,
                                                                const float* __restrict__ bsa, ProdIdx ix,
                                                                float* __restrict__ CS, float* __restrict__ score, int N, int Hd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npairs = ix.counts[1];
  const long NN = (long)N * N;
  float w[HC];
#pragma unroll
  for (int cc = 0; cc < HC; ++cc) w[cc] = (lane + 64 * cc) < Hd ? wsa[lane + 64 * cc] : 0.f;
  const float bias = bsa[0];
  for (int prow = blockIdx.x * PW + wave; prow < npairs; prow += gridDim.x * PW) {
    const long pair = ix.prow_pair[prow];
    const long b = pair / NN;
    const int ij = (int)(pair - b * NN), i = ij / N, j = ij - i * N;
    const int row0 = ix.pair_row0[pair], n = __popc(ix.pair_bits[pair]);
    const float div = ix.pair_div[pair];
    const float* nj = nterm + (b * N + j) * Hd;   // head side: node_feat.unsqueeze(0) -> indexed by the column entity
    const float* ni = nterm + (b * N + i) * Hd;   // tail side: node_feat.unsqueeze(1) -> indexed by the row entity
    float nh[HC], nt[HC], ch[HC], ct[HC];
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) {
      const int c = lane + 64 * cc;
      nh[cc] = c < Hd ? nj[c] : 0.f, nt[cc] = c < Hd ? ni[c] : 0.f, ch[cc] = 0.f, ct[cc] = 0.f;
    }
    for (int k = 0; k < n; ++k) {
      const long row = row0 + k;
      float ph = 0.f, pt = 0.f, cw[HC];
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        cw[cc] = 0.f;
        if (c < Hd) {
          const float sf = sfeat[row * Hd + c];
          cw[cc] = cwa[row * Hd + c];
          ph = fmaf(w[cc], tanhf(sf + nh[cc]), ph);
          pt = fmaf(w[cc], tanhf(sf + nt[cc]), pt);
        }
      }
      const float sh = wave_sum(ph) + bias, st = wave_sum(pt) + bias;
      if (lane == 0) score[2 * row] = sh, score[2 * row + 1] = st;
      const float ah = fmaxf(sh, 0.f), at = fmaxf(st, 0.f);
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) ch[cc] = fmaf(ah, cw[cc], ch[cc]), ct[cc] = fmaf(at, cw[cc], ct[cc]);
    }
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) {
      const int c = lane + 64 * cc;
      if (c < Hd) CS[(long)prow * 2 * Hd + c] = ch[cc] / div, CS[(long)prow * 2 * Hd + Hd + c] = ct[cc] / div;
    }
  }
  if (blockIdx.x == 0) {
    const int hi = (npairs + 63) & ~63;
    for (long e = (long)npairs * 2 * Hd + threadIdx.x; e < (long)hi * 2 * Hd; e += 64 * PW) CS[e] = 0.f;
  }
}

// ---- E[b, i, j, :] = Ec[prow] for pairs with a live slot, the bias of linear_sentence_att for the others (their
// sentence sums are exactly zero), 0 for padding entities.  One workgroup per entity row (b, i).
__global__ __launch_bounds__(256) void prod_expand_kernel(const float* __restrict__ Ec, const float* __restrict__ bls,
                                                          const int* __restrict__ n_valid, const int* __restrict__ pair_prow,
                                                          const int* __restrict__ counts, float* __restrict__ E, int N, int Hd) {
  // counts[2]: the caller's capacities were too small for this batch (prod_index_b marked every pair dead).  Under a captured
  // hipGraph nobody can raise, so the result is poisoned: NaN in every real pair instead of a plausible all-bias tensor.
  const bool over = counts[2] != 0;
  const long bi = blockIdx.x;
  const long b = bi / N;
  const int i = (int)(bi - b * N);
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  const long tot = (long)N * Hd;
  for (long e = threadIdx.x; e < tot; e += 256) {
    const int j = (int)(e / Hd), c = (int)(e - (long)j * Hd);
    float v = 0.f;
    if (i < nv && j < nv) {
      const int prow = pair_prow[bi * N + j];
      v = over ? __builtin_nanf("") : (prow >= 0 ? Ec[(long)prow * Hd + c] : bls[c]);
    }
    E[bi * tot + e] = v;
  }
}

// Compact mode, capacities too small (counts[2]): nothing was computed.  Every real pair is pointed at row 0 of Ec and that row
// is NaN, so the consumers' outputs are NaN -- as loud as the dense path's poisoned E, also under a captured hipGraph.
__global__ __launch_bounds__(256) void prod_poison_kernel(ProdIdx ix, float* __restrict__ Ec, long BNN, int Hd) {
  if (ix.counts[2] == 0) return;
  const long pp = (long)blockIdx.x * 256 + threadIdx.x;
  if (pp < BNN && ix.pair_div[pp] > 0.f) ix.pair_prow[pp] = 0;
  if (pp < Hd) Ec[pp] = __builtin_nanf("");
}

// ---- backward: dEc[prow] = dE[pair];  wpair[pair] = 1 for real pairs (weights of the bias-gradient column sum) -----
__global__ __launch_bounds__(64 * PW) void prod_gather_kernel(const float* __restrict__ dE, ProdIdx ix, float* __restrict__ dEc,
                                                              int Hd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npairs = ix.counts[1];
  for (int prow = blockIdx.x * PW + wave; prow < npairs; prow += gridDim.x * PW) {
    const long pair = ix.prow_pair[prow];
    for (int c = lane; c < Hd; c += 64) dEc[(long)prow * Hd + c] = dE[pair * Hd + c];
  }
  if (blockIdx.x == 0) {
    const int hi = (npairs + 63) & ~63;
    for (long e = (long)npairs * Hd + threadIdx.x; e < (long)hi * Hd; e += 64 * PW) dEc[e] = 0.f;
  }
}

// ---- backward of the sentence attention ------------------------------------------------------------------------------
//   in : dCS[prow] = [dcs_h | dcs_t];  out: dcwa[row] (attention-weighted part), dsfeat[row], dnterm += (atomics),
//        dwb[0..Hd) += d w, dwb[Hd] += d c (atomics, once per workgroup).  dynamic LDS: (Hd + 1) floats.
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_sent_bwd_kernel(const float* __restrict__ sfeat, const float* __restrict__ cwa,
                                                                const float* __restrict__ nterm, const float* __restrict__ wsa,
                                                                const float* __restrict__ score, const float* __restrict__ dCS,
                                                                ProdIdx ix, float* __restrict__ dcwa, float* __restrict__ dsfeat,
                                                                float* __restrict__ dnterm, float* __restrict__ dwb, int N, int Hd) {
  extern __shared__ float sacc[];  // [Hd + 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c <= Hd; c += 64 * PW) sacc[c] = 0.f;
  __syncthreads();
  const int npairs = ix.counts[1], nrows = ix.counts[0];
  const long NN = (long)N * N;
  float w[HC], dw[HC];
#pragma unroll
  for (int cc = 0; cc < HC; ++cc) w[cc] = (lane + 64 * cc) < Hd ? wsa[lane + 64 * cc] : 0.f, dw[cc] = 0.f;
  float dbias = 0.f;
  for (int prow = blockIdx.x * PW + wave; prow < npairs; prow += gridDim.x * PW) {
    const long pair = ix.prow_pair[prow];
    const long b = pair / NN;
    const int ij = (int)(pair - b * NN), i = ij / N, j = ij - i * N;
    const int row0 = ix.pair_row0[pair], n = __popc(ix.pair_bits[pair]);
    const float div = ix.pair_div[pair];
    const long oj = (b * N + j) * Hd, oi = (b * N + i) * Hd;
    float nh[HC], nt[HC], gh[HC], gt[HC], dnh[HC], dnt[HC];
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) {
      const int c = lane + 64 * cc;
      nh[cc] = c < Hd ? nterm[oj + c] : 0.f, nt[cc] = c < Hd ? nterm[oi + c] : 0.f;
      gh[cc] = c < Hd ? dCS[(long)prow * 2 * Hd + c] / div : 0.f;          // d(sum_s att_h cwa)
      gt[cc] = c < Hd ? dCS[(long)prow * 2 * Hd + Hd + c] / div : 0.f;
      dnh[cc] = 0.f, dnt[cc] = 0.f;
    }
    for (int k = 0; k < n; ++k) {
      const long row = row0 + k;
      const float sh = score[2 * row], st = score[2 * row + 1];
      const float ah = fmaxf(sh, 0.f), at = fmaxf(st, 0.f);
      float cw[HC], zh[HC], zt[HC], ph = 0.f, pt = 0.f;
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        cw[cc] = 0.f, zh[cc] = 0.f, zt[cc] = 0.f;
        if (c < Hd) {
          const float sf = sfeat[row * Hd + c];
          cw[cc] = cwa[row * Hd + c];
          zh[cc] = tanhf(sf + nh[cc]), zt[cc] = tanhf(sf + nt[cc]);
          ph = fmaf(gh[cc], cw[cc], ph), pt = fmaf(gt[cc], cw[cc], pt);
        }
      }
      // d att -> d score through relu (score <= 0: zero gradient, as torch.relu)
      const float dsh = sh > 0.f ? wave_sum(ph) : 0.f, dst = st > 0.f ? wave_sum(pt) : 0.f;
      dbias += dsh + dst;
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        if (c < Hd) {
          const float dzh = dsh * w[cc] * (1.f - zh[cc] * zh[cc]), dzt = dst * w[cc] * (1.f - zt[cc] * zt[cc]);
          dcwa[row * Hd + c] = ah * gh[cc] + at * gt[cc];
          dsfeat[row * Hd + c] = dzh + dzt;
          dnh[cc] += dzh, dnt[cc] += dzt;
          dw[cc] = fmaf(dsh, zh[cc], fmaf(dst, zt[cc], dw[cc]));
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) {
      const int c = lane + 64 * cc;
      if (c < Hd) {
        atomicAdd(dnterm + oj + c, dnh[cc]);
        atomicAdd(dnterm + oi + c, dnt[cc]);
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < HC; ++cc)
    if ((lane + 64 * cc) < Hd) atomicAdd(&sacc[lane + 64 * cc], dw[cc]);
  if (lane == 0) atomicAdd(&sacc[Hd], dbias);   // every lane holds the same dbias
  __syncthreads();
  for (int c = threadIdx.x; c <= Hd; c += 64 * PW)
    if (sacc[c] != 0.f) atomicAdd(dwb + c, sacc[c]);
  if (blockIdx.x == 0) {
    const int hi = (nrows + 63) & ~63;
    for (long e = (long)nrows * Hd + threadIdx.x; e < (long)hi * Hd; e += 64 * PW) dcwa[e] = 0.f, dsfeat[e] = 0.f;
  }
}

// ---- backward of the word attention, two passes, no atomics -------------------------------------------------------------
//   dcw = dCW[row, side]:  g[t] = dcw . ctx[t];  dscore[t] = att[t] (g[t] - sum_t att g)
//   dtable[b, pos[t], t] += dscore,  dctx[b, t, :] += att[t] dcw     -- sums over MANY (row, side) per (document, token)
// Pass 1 (one wave per (row, side), like the forward): the softmax-weighted sum  dots[r] = sum_t att[t] g[t], the slot's
// token range and the document's last covered token.  Pass 2 (one workgroup per (document, token), the OWNER of
// dctx[b, t, :] and dtable[b, :, t]): scans the document's live rows, and for those whose slot holds the token recomputes
// att and g (a 128-wide dot with the token state it keeps in registers) and accumulates in a fixed order -- deterministic,
// and ~20x faster than scatter-adding with fp32 atomics.
// dynamic LDS pass 1: PW * T floats
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_word_bwd_rows_kernel(const float* __restrict__ ctx, const unsigned char* __restrict__ sen,
                                                                     const void* __restrict__ pos_h, const void* __restrict__ pos_t,
                                                                     int pos_bytes, const float* __restrict__ table,
                                                                     const float* __restrict__ stats, const float* __restrict__ dCW,
                                                                     ProdIdx ix, float* __restrict__ dots, int N, int S, int T, int Hd,
                                                                     int ND) {
  extern __shared__ float wsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* av = wsm + (long)wave * T;  // att[t]
  const int nrows = ix.counts[0], total = 2 * nrows;
  const long slots_per_doc = (long)N * N * S;
  for (int r0 = blockIdx.x * PW; r0 < total; r0 += gridDim.x * PW) {
    const int r = r0 + wave;
    const bool on = r < total;
    int t0 = T, t1 = 0;
    long b = 0;
    if (on) {
      const int row = r >> 1, side = r & 1;
      const long slot = ix.row_slot[row];
      b = slot / slots_per_doc;
      const void* pos = side ? pos_t : pos_h;
      const unsigned char* m = sen + slot * T;
      const float* tab = table + b * ND * T;
      const float mx = stats[2 * (long)r], inv = stats[2 * (long)r + 1];
      for (int c0 = 0; c0 < T; c0 += 512) {
        float sc[8];
        slot_scores8(m, pos, pos_bytes, slot * T, tab, T, ND, c0, lane, sc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = c0 + 64 * u + lane;
          if (t < T) {
            av[t] = sc[u] == -INFINITY ? 0.f : expf(sc[u] - mx) * inv;
            if (sc[u] != -INFINITY) t0 = min(t0, t), t1 = max(t1, t + 1);
          }
        }
      }
      t0 = -(int)wave_max((float)-t0), t1 = (int)wave_max((float)t1);
    }
    float* dl = wsm + (long)PW * T + (long)wave * Hd;   // this (row, side)'s dCW row, broadcast to every lane below
    if (on)
      for (int c = lane; c < Hd; c += 64) dl[c] = dCW[(long)(r >> 1) * 2 * Hd + (r & 1) * Hd + c];
    __syncthreads();
    if (on && (Hd & 3) == 0) {
      // lanes over TOKENS: g[t] = dcw . ctx[t] is a private 128-wide dot per lane (16-byte loads of the lane's own token row,
      // dcw broadcast from LDS) -- one wave reduction per 64 tokens instead of one per token
      const float* cb = ctx + b * T * Hd;
      float dotsum = 0.f;
      for (int tb = t0; tb < t1; tb += 64) {
        const int t = tb + lane;
        const float4* cr = reinterpret_cast<const float4*>(cb + (long)min(t, t1 - 1) * Hd);
        float g0 = 0.f, g1 = 0.f;
        for (int q = 0; q < Hd / 4; q += 2) {
          const float4 x0 = cr[q], x1 = cr[q + 1 < Hd / 4 ? q + 1 : q];
          const float4 d0 = *reinterpret_cast<const float4*>(dl + 4 * q), d1 = *reinterpret_cast<const float4*>(dl + 4 * (q + 1 < Hd / 4 ? q + 1 : q));
          g0 = fmaf(x0.x, d0.x, fmaf(x0.y, d0.y, fmaf(x0.z, d0.z, fmaf(x0.w, d0.w, g0))));
          if (q + 1 < Hd / 4) g1 = fmaf(x1.x, d1.x, fmaf(x1.y, d1.y, fmaf(x1.z, d1.z, fmaf(x1.w, d1.w, g1))));
        }
        if (t < t1) dotsum = fmaf(av[t], g0 + g1, dotsum);
      }
      dotsum = wave_sum(dotsum);
      if (lane == 0) dots[r] = dotsum;
    } else if (on) {
      const int row = r >> 1, side = r & 1;
      const float* cb = ctx + b * T * Hd;
      float dcw[HC], dotsum = 0.f;
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) dcw[cc] = (lane + 64 * cc) < Hd ? dCW[(long)row * 2 * Hd + side * Hd + lane + 64 * cc] : 0.f;
      for (int t = t0; t < t1; t += 4) {  // four tokens at a time: independent loads, interleaved reductions
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = min(t + u, t1 - 1);
#pragma unroll
          for (int cc = 0; cc < HC; ++cc) {
            const int c = lane + 64 * cc;
            if (c < Hd) p[u] = fmaf(dcw[cc], cb[(long)tt * Hd + c], p[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float g = wave_sum(p[u]);
          if (t + u < t1) dotsum = fmaf(av[t + u], g, dotsum);
        }
      }
      if (lane == 0) dots[r] = dotsum;
    }
    __syncthreads();
  }
}

// pass 2: one workgroup per (document, token); LDS: PW * (Hd + ND) floats
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_word_bwd_tok_kernel(const float* __restrict__ ctx, const unsigned char* __restrict__ sen,
                                                                    const void* __restrict__ pos_h, const void* __restrict__ pos_t,
                                                                    int pos_bytes, const float* __restrict__ table,
                                                                    const float* __restrict__ stats, const float* __restrict__ dCW,
                                                                    const float* __restrict__ dots, ProdIdx ix,
                                                                    float* __restrict__ dtable, float* __restrict__ dctx, int T, int Hd,
                                                                    int ND) {
  extern __shared__ float sm[];  // [PW][Hd] partial dctx rows | [PW][ND] partial dtable columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long bt = blockIdx.x;
  const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
  float* accs = sm;
  float* dts = sm + (long)PW * Hd;
  for (int e = threadIdx.x; e < PW * ND; e += 64 * PW) dts[e] = 0.f;
  float acc[HC], cx[HC];
#pragma unroll
  for (int cc = 0; cc < HC; ++cc) acc[cc] = 0.f, cx[cc] = (lane + 64 * cc) < Hd ? ctx[bt * Hd + lane + 64 * cc] : 0.f;
  __syncthreads();
  if (t < ix.tmax[b]) {
    const int r_lo = ix.doc_off[b], r_hi = ix.doc_off[b + 1];
    const float* tab = table + (long)b * ND * T;
    float* dtw = dts + wave * ND;
    for (int base = r_lo + wave * 64; base < r_hi; base += PW * 64) {
      // lane-parallel: everything a hit needs except the dCW rows -- one latency chain per 64 candidate rows, not per hit
      const int row = base + lane;
      bool hit = false;
      float ah = 0.f, at = 0.f, dh = 0.f, dt = 0.f;
      int kh = 0, kt = 0;
      if (row < r_hi) {
        const int rg = ix.row_rng[row];
        if (t >= (rg & 0xffff) && t < (rg >> 16)) {
          const long slot = ix.row_slot[row];
          if (sen[slot * T + t]) {
            hit = true;
            kh = min(max(pos_at(pos_h, pos_bytes, slot * T + t), 0), ND - 1);
            kt = min(max(pos_at(pos_t, pos_bytes, slot * T + t), 0), ND - 1);
            const float4 st = *reinterpret_cast<const float4*>(stats + 4L * row);   // max / 1/sum of the head, of the tail side
            ah = expf(tab[(long)kh * T + t] - st.x) * st.y;
            at = expf(tab[(long)kt * T + t] - st.z) * st.w;
            dh = dots[2L * row], dt = dots[2L * row + 1];
          }
        }
      }
      unsigned long long todo = __ballot(hit);
      float nh[HC], nt[HC];   // dCW rows of the NEXT hit, requested one hit ahead
      int l = todo ? __ffsll((long long)todo) - 1 : 0;
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        nh[cc] = (todo && c < Hd) ? dCW[(long)(base + l) * 2 * Hd + c] : 0.f;
        nt[cc] = (todo && c < Hd) ? dCW[(long)(base + l) * 2 * Hd + Hd + c] : 0.f;
      }
      while (todo) {
        const int cur = l;
        todo &= todo - 1;
        float ch[HC], ct[HC], ph = 0.f, pt = 0.f;
#pragma unroll
        for (int cc = 0; cc < HC; ++cc) ch[cc] = nh[cc], ct[cc] = nt[cc];
        if (todo) {
          l = __ffsll((long long)todo) - 1;
#pragma unroll
          for (int cc = 0; cc < HC; ++cc) {
            const int c = lane + 64 * cc;
            nh[cc] = c < Hd ? dCW[(long)(base + l) * 2 * Hd + c] : 0.f;
            nt[cc] = c < Hd ? dCW[(long)(base + l) * 2 * Hd + Hd + c] : 0.f;
          }
        }
#pragma unroll
        for (int cc = 0; cc < HC; ++cc) ph = fmaf(ch[cc], cx[cc], ph), pt = fmaf(ct[cc], cx[cc], pt);
        const float gh = wave_sum(ph), gt = wave_sum(pt);
        const float a_h = lane_value(ah, cur), a_t = lane_value(at, cur);
        if (lane == 0) {   // one lane, program order: deterministic
          dtw[__builtin_amdgcn_readlane(kh, cur)] += a_h * (gh - lane_value(dh, cur));
          dtw[__builtin_amdgcn_readlane(kt, cur)] += a_t * (gt - lane_value(dt, cur));
        }
#pragma unroll
        for (int cc = 0; cc < HC; ++cc) acc[cc] = fmaf(a_h, ch[cc], fmaf(a_t, ct[cc], acc[cc]));
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < HC; ++cc)
    if ((lane + 64 * cc) < Hd) accs[wave * Hd + lane + 64 * cc] = acc[cc];
  __syncthreads();
  for (int c = threadIdx.x; c < Hd; c += 64 * PW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PW; ++w) v += accs[w * Hd + c];
    dctx[bt * Hd + c] = v;
  }
  for (int k = threadIdx.x; k < ND; k += 64 * PW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PW; ++w) v += dts[w * ND + k];
    dtable[((long)b * ND + k) * T + t] = v;
  }
}

// ---- backward of the score table ------------------------------------------------------------------------------------
//   dsentF[b, t, c] = sum_k dtable[b,k,t] w_a[c] (1 - z^2),  ddisF[k, c] = (the same, summed over b, t),  dwa[c] = sum dtable z,
//   dba = sum dtable          z = tanh(sentF[b,t,c] + disF[k,c]).
// Only tokens below tmax[b] carry a gradient.  Grid (token chunks of TCH, B): a workgroup whose chunk lies beyond tmax[b]
// zeroes its dsentF rows and leaves; the others accumulate per wave in LDS (fixed order) and write ONE partial record
// [ND * Hd | Hd | 1]; prod_table_fin_kernel sums the live records in (document, chunk) order -- deterministic, no atomics.
constexpr int TCH = 16;  // tokens per workgroup (4 per wave)
template <int HC>
__global__ __launch_bounds__(64 * PW) void prod_table_bwd_kernel(const float* __restrict__ sentF, const float* __restrict__ disF,
                                                                 const float* __restrict__ wa, const float* __restrict__ dtable,
                                                                 const int* __restrict__ tmax, float* __restrict__ dsentF,
                                                                 float* __restrict__ part, int T, int Hd, int ND) {
  extern __shared__ float sacc[];  // [PW][ND * Hd + Hd + 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, chunk = blockIdx.x, rec = ND * Hd + Hd + 1;
  const int tlim = min(tmax[b], T);
  const int tb = chunk * TCH + wave * (TCH / PW);
  if (chunk * TCH >= tlim) {  // nothing reaches these tokens (uniform over the workgroup)
    for (int i = 0; i < TCH / PW; ++i) {
      const int t = tb + i;
      if (t < T)
        for (int c = lane; c < Hd; c += 64) dsentF[((long)b * T + t) * Hd + c] = 0.f;
    }
    return;
  }
  float* mine = sacc + (long)wave * rec;
  for (int e = lane; e < rec; e += 64) mine[e] = 0.f;
  float w[HC], dw[HC];
#pragma unroll
  for (int cc = 0; cc < HC; ++cc) w[cc] = (lane + 64 * cc) < Hd ? wa[lane + 64 * cc] : 0.f, dw[cc] = 0.f;
  float dbias = 0.f;
  for (int i = 0; i < TCH / PW; ++i) {
    const int t = tb + i;
    if (t >= T) break;
    const long bt = (long)b * T + t;
    float sf[HC], ds[HC];
#pragma unroll
    for (int cc = 0; cc < HC; ++cc) sf[cc] = (lane + 64 * cc) < Hd ? sentF[bt * Hd + lane + 64 * cc] : 0.f, ds[cc] = 0.f;
    const float gl = lane < ND ? dtable[((long)b * ND + lane) * T + t] : 0.f;   // all ND ids at once
    unsigned long long todo = __ballot(gl != 0.f);
    while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const float g = lane_value(gl, k);
      dbias += g;
#pragma unroll
      for (int cc = 0; cc < HC; ++cc) {
        const int c = lane + 64 * cc;
        if (c < Hd) {
          const float z = tanhf(sf[cc] + disF[(long)k * Hd + c]);
          const float d = g * w[cc] * (1.f - z * z);
          ds[cc] += d;
          mine[k * Hd + c] += d;          // this wave's own LDS record, program order
          dw[cc] = fmaf(g, z, dw[cc]);
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < HC; ++cc)
      if ((lane + 64 * cc) < Hd) dsentF[bt * Hd + lane + 64 * cc] = ds[cc];
  }
#pragma unroll
  for (int cc = 0; cc < HC; ++cc)
    if ((lane + 64 * cc) < Hd) mine[ND * Hd + lane + 64 * cc] = dw[cc];
  if (lane == 0) mine[ND * Hd + Hd] = dbias;
  __syncthreads();
  float* out = part + ((long)b * gridDim.x + chunk) * rec;
  for (int e = threadIdx.x; e < rec; e += 64 * PW) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < PW; ++q) v += sacc[(long)q * rec + e];
    out[e] = v;
  }
}
// ddisF | dwa | dba = sum of the live partial records, documents and chunks in order
__global__ __launch_bounds__(256) void prod_table_fin_kernel(const float* __restrict__ part, const int* __restrict__ tmax, int B, int T,
                                                             int nchunk, int rec, int nd_hd, float* __restrict__ ddisF,
                                                             float* __restrict__ dwab) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rec) return;
  // documents and chunks in order, eight requests in flight at a time (one dependent round trip per record before: 36 us)
  float v = 0.f;
  for (int b = 0; b < B; ++b) {
    const int live = (min(tmax[b], T) + TCH - 1) / TCH;
    for (int c0 = 0; c0 < live; c0 += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = part[((long)b * nchunk + min(c0 + u, live - 1)) * rec + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (c0 + u < live) ? x[u] : 0.f;
    }
  }
  if (e < nd_hd) ddisF[e] = v;
  else dwab[e - nd_hd] = v;
}

// ---- column sums over the compact rows (bias gradients): part[slice][C], then out[c] = sum of the slices in order ----
constexpr int DCS = 64;
__global__ __launch_bounds__(256) void prod_colsum_dyn_kernel(const float* __restrict__ X, const int* __restrict__ cnt, int C,
                                                              float* __restrict__ part) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, sp = blockIdx.y;
  const long R = *cnt, rps = (R + DCS - 1) / DCS;
  const long r0 = sp * rps, r1 = min(R, r0 + rps);
  float acc = 0.f;
  if (c < C && r0 + wave < r1) {   // this wave's rows in order, eight requests in flight at a time
    const long last = r0 + wave + ((r1 - 1 - r0 - wave) / 4) * 4;
    for (long r = r0 + wave; r < r1; r += 32) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = X[min(r + 4 * u, last) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (r + 4 * u < r1) ? x[u] : 0.f;
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) part[(long)sp * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void prod_colsum_fin_kernel(const float* __restrict__ part, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int q = 0; q < DCS; ++q) s += part[(long)q * C + c];
  out[c] = s;
}

// =====================================================================================================================
// host side
// =====================================================================================================================
// kernels are instantiated for HC = ceil(Hd / 64) in {1, 2, 4, 8} column chunks per lane (the reference's Hd = 128: HC = 2)
#define PROD_LAUNCH(kernel, Hd, grid, block, lds, st, ...)                                            \
  do {                                                                                                \
    const int _hc = ((Hd) + 63) / 64;                                                                 \
    if (_hc <= 1) hipLaunchKernelGGL((kernel<1>), grid, block, lds, st, __VA_ARGS__);                 \
    else if (_hc <= 2) hipLaunchKernelGGL((kernel<2>), grid, block, lds, st, __VA_ARGS__);            \
    else if (_hc <= 4) hipLaunchKernelGGL((kernel<4>), grid, block, lds, st, __VA_ARGS__);            \
    else hipLaunchKernelGGL((kernel<8>), grid, block, lds, st, __VA_ARGS__);                          \
  } while (0)

struct ProdLayout {  // offsets (floats) inside the block's flat parameter buffer; reference tensors keep their [out, in] layout
  long Ws, bs, Wp, bp, wa, ba;           // word_attention.{attention_sent, attention_pos, attention_all}
  long Wlw, blw;                         // linear_word_att            [Hd, 2Hd]
  long Wss, bss, Wsp, bsp, wsa, bsa;     // sentence_attention.{attention_sent, attention_pos, attention_all}
  long Wls, bls;                         // linear_sentence_att        [Hd, 2Hd]
  long total;
};
ProdLayout prod_layout(int Hd, int P) {
  ProdLayout y;
  long o = 0;
  auto take = [&](long n) { const long at = o; o += (n + 3) & ~3L; return at; };   // every piece 16-byte aligned (float4 GEMM loads)
  y.Ws = take((long)Hd * Hd), y.bs = take(Hd), y.Wp = take((long)Hd * P), y.bp = take(Hd), y.wa = take(Hd), y.ba = take(1);
  y.Wlw = take(2L * Hd * Hd), y.blw = take(Hd);
  y.Wss = take((long)Hd * Hd), y.bss = take(Hd), y.Wsp = take((long)Hd * Hd), y.bsp = take(Hd), y.wsa = take(Hd), y.bsa = take(1);
  y.Wls = take(2L * Hd * Hd), y.bls = take(Hd);
  y.total = o;
  return y;
}

static long up64(long v) { return (v + 63) & ~63L; }

int prod_index(const unsigned char* sen, const int* n_valid, ProdIdx ix, int B, int N, int S, int T, long cap_rows, long cap_pairs,
               hipStream_t st) {
  GC_REQUIRE(S >= 1 && S <= 31, "producer: %d sentence slots per pair (1..31 supported)", S);
  GC_REQUIRE((long)B * N * N * S < (1L << 31), "producer: B*N*N*S exceeds 32-bit slot indices");
  ProfScope ps("prod_index", st);
  hipLaunchKernelGGL(prod_index_a_kernel, dim3(B), dim3(256), 0, st, sen, n_valid, ix, N, S, T);
  if (int e = check_launch("prod_index_a")) return e;
  hipLaunchKernelGGL(prod_index_b_kernel, dim3(cdiv((long)N * N, 256), B), dim3(256), 0, st, n_valid, ix, B, N, S, (int)cap_rows,
                     (int)cap_pairs);
  return check_launch("prod_index_b");
}

static int linear_fwd(const float* X, long M, int K, const float* W, const float* bias, int Nout, float* Y, float* ws, long wse,
                      hipStream_t st, const int* cnt = nullptr, long cap = 0) {  // Y = X W^T + b, W stored [Nout, K]
  GemmArgs g;
  g.A = X, g.lda = K, g.a_kc = 1;
  g.B = W, g.ldb = K, g.b_kc = 1;
  g.C = Y, g.ldc = Nout;
  g.M = (int)M, g.N = Nout, g.K = K;
  g.bias = bias;
  g.ws = ws, g.ws_elems = wse;
  g.tag = "prod_gemm";
  return cnt ? gemm_dyn(g, cnt, 1, cap, st) : gemm(g, st);
}
// dX (+)= dY W          (W stored [Nout, K]: the "N" form with k = Nout)
static int linear_bwd_x(const float* dY, long M, int Nout, const float* W, int K, float* dX, int accumulate, float* ws, long wse,
                        hipStream_t st, const int* cnt = nullptr, long cap = 0) {
  GemmArgs g;
  g.A = dY, g.lda = Nout, g.a_kc = 1;
  g.B = W, g.ldb = K, g.b_kc = 0;
  g.C = dX, g.ldc = K;
  g.M = (int)M, g.N = K, g.K = Nout;
  g.accumulate = accumulate;
  g.ws = ws, g.ws_elems = wse;
  g.tag = "prod_gemm";
  return cnt ? gemm_dyn(g, cnt, 1, cap, st) : gemm(g, st);
}
// dW[Nout, K] = dY^T X   (rows = the reduction dimension)
static int linear_bwd_w(const float* dY, const float* X, long M, int Nout, int K, float* dW, float* ws, long wse, hipStream_t st,
                        const int* cnt = nullptr, long cap = 0) {
  GemmArgs g;
  g.A = dY, g.lda = Nout, g.a_kc = 0;
  g.B = X, g.ldb = K, g.b_kc = 0;
  g.C = dW, g.ldc = K;
  g.M = Nout, g.N = K, g.K = (int)M;
  g.ws = ws, g.ws_elems = wse;
  g.tag = "prod_gemm";
  return cnt ? gemm_dyn(g, cnt, 2, cap, st) : gemm(g, st);
}

// dW = dY^T X and dX (+)= dY W of one Linear layer whose row count lives on the device: one launch (+ dW's reduce), gemm_dyn_pair
static int linear_bwd_wx_dyn(const float* dY, const float* X, int Nout, int K, const float* W, float* dW, float* dX, int accumulate_x,
                             float* ws, long wse, hipStream_t st, const int* cnt, long cap) {
  GemmArgs gw, gx;
  gw.A = dY, gw.lda = Nout, gw.a_kc = 0, gw.B = X, gw.ldb = K, gw.b_kc = 0, gw.C = dW, gw.ldc = K, gw.M = Nout, gw.N = K, gw.K = 0;
  gx.A = dY, gx.lda = Nout, gx.a_kc = 1, gx.B = W, gx.ldb = K, gx.b_kc = 0, gx.C = dX, gx.ldc = K, gx.M = 0, gx.N = K, gx.K = Nout;
  gx.accumulate = accumulate_x;
  gw.ws = gx.ws = ws, gw.ws_elems = gx.ws_elems = wse;
  gw.tag = gx.tag = "prod_gemm";
  return gemm_dyn_pair(gw, gx, cnt, cap, st);
}

// The three gradients of one Linear layer on a static row count -- dW = dY^T X, db = column sums of dY, dX (+)= dY W -- in ONE
// group launch (+ its reduce): the two products are independent, the column sums ride in both launches (gemm.hpp, ColRide).
// Separately they were a weight-gradient launch, its reduce, one or two column-sum launches and the data-gradient launch.
static int linear_bwd_all(const float* dY, const float* X, long M, int Nout, int K, const float* W, float* dW, float* db, float* dX,
                          int accumulate_x, float* ws, long wse, hipStream_t st) {
  const long col_elems = ((long)COL_RIDE_SLICES * Nout + 3) & ~3L;
  if (!ws || wse <= col_elems + 4) {
    GC_TRY(linear_bwd_w(dY, X, M, Nout, K, dW, ws, wse, st));
    GC_TRY(colsum(dY, nullptr, db, M, Nout, Nout, 1, 0, 0, 0, 0, ws, st));
    return linear_bwd_x(dY, M, Nout, W, K, dX, accumulate_x, ws, wse, st);
  }
  GemmArgs gs[2];
  gs[0].A = dY, gs[0].lda = Nout, gs[0].a_kc = 0, gs[0].B = X, gs[0].ldb = K, gs[0].b_kc = 0;
  gs[0].C = dW, gs[0].ldc = K, gs[0].M = Nout, gs[0].N = K, gs[0].K = (int)M;
  gs[1].A = dY, gs[1].lda = Nout, gs[1].a_kc = 1, gs[1].B = W, gs[1].ldb = K, gs[1].b_kc = 0;
  gs[1].C = dX, gs[1].ldc = K, gs[1].M = (int)M, gs[1].N = K, gs[1].K = Nout, gs[1].accumulate = accumulate_x;
  gs[0].ws = gs[1].ws = ws, gs[0].ws_elems = gs[1].ws_elems = wse - col_elems;
  gs[0].tag = gs[1].tag = "prod_gemm";
  ColRide cr;
  cr.X = dY, cr.out = db, cr.part = ws + (wse - col_elems), cr.R = M, cr.ld = Nout, cr.C = Nout;
  return gemm_group(gs, 2, st, &cr);
}

struct ProdBufs {  // caller-owned device memory (include/gcgcn.h lists the sizes)
  float *sentF, *disF, *table, *CW, *stats, *cwa, *sfeat, *nterm, *score, *CS, *Ec;
};

int prod_fwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const unsigned char* sen, const void* pos_h,
             const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int* n_valid, const float* flat,
             ProdIdx ix, long cap_rows, long cap_pairs, ProdBufs w, float* E, float* ws, long wse, hipStream_t st) {
  GC_REQUIRE(Hd >= 1 && Hd <= 64 * HCLIM, "producer: hidden width %d (1..%d supported)", Hd, 64 * HCLIM);
  GC_REQUIRE(pos_bytes == 8 || pos_bytes == 4 || pos_bytes == 1, "producer: position ids must be int64, int32 or uint8");
  GC_REQUIRE((size_t)PW * 2 * T * sizeof(float) <= 64 * 1024, "producer: T=%d tokens exceed the LDS budget", T);
  const ProdLayout y = prod_layout(Hd, P);
  const long BT = (long)B * T;
  {  // two Linear layers on static inputs in one launch: the token states (glove:178) and the per-entity term of the sentence
     // attention (glove:202), which nothing needs before prod_sent_fwd below
    GemmArgs gs[2];
    for (int q = 0; q < 2; ++q) {
      GemmArgs& g = gs[q];
      g.A = q ? node : ctx, g.lda = Hd, g.a_kc = 1;
      g.B = flat + (q ? y.Wsp : y.Ws), g.ldb = Hd, g.b_kc = 1;
      g.C = q ? w.nterm : w.sentF, g.ldc = Hd;
      g.M = q ? (int)((long)B * N) : (int)BT, g.N = Hd, g.K = Hd;
      g.bias = flat + (q ? y.bsp : y.bs);
      g.ws = ws, g.ws_elems = wse;
      g.tag = "prod_gemm";
    }
    GC_TRY(gemm_group(gs, 2, st));
  }
  GC_TRY(linear_fwd(dis_table, ND, P, flat + y.Wp, flat + y.bp, Hd, w.disF, ws, wse, st));      // glove:179 on the 21 ids
  {
    ProfScope ps("prod_table", st);
    PROD_LAUNCH(prod_table_fwd_kernel, Hd, dim3(cdiv(BT, PW)), dim3(64 * PW), 0, st, w.sentF, w.disF, flat + y.wa,
                       flat + y.ba, w.table, BT, T, Hd, ND);
    GC_TRY(check_launch("prod_table_fwd"));
  }
  GC_TRY(prod_index(sen, n_valid, ix, B, N, S, T, cap_rows, cap_pairs, st));
  {
    ProfScope ps("prod_word", st);
    PROD_LAUNCH(prod_word_fwd_kernel, Hd, dim3(PGRID), dim3(64 * PW), (size_t)PW * T * sizeof(float), st, ctx, sen, pos_h, pos_t,
                       pos_bytes, w.table, ix, w.CW, w.stats, N, S, T, Hd, ND);
    GC_TRY(check_launch("prod_word_fwd"));
    hipLaunchKernelGGL(prod_tmax_kernel, dim3(B), dim3(256), 0, st, ix);
    GC_TRY(check_launch("prod_tmax"));
  }
  GC_TRY(linear_fwd(w.CW, 0, 2 * Hd, flat + y.Wlw, flat + y.blw, Hd, w.cwa, ws, wse, st, ix.counts, cap_rows));       // :320-321
  GC_TRY(linear_fwd(w.cwa, 0, Hd, flat + y.Wss, flat + y.bss, Hd, w.sfeat, ws, wse, st, ix.counts, cap_rows));        // :201
  {
    ProfScope ps("prod_sent", st);
    PROD_LAUNCH(prod_sent_fwd_kernel, Hd, dim3(PGRID), dim3(64 * PW), 0, st, w.sfeat, w.cwa, w.nterm, flat + y.wsa, flat + y.bsa,
                       ix, w.CS, w.score, N, Hd);
    GC_TRY(check_launch("prod_sent_fwd"));
  }
  GC_TRY(linear_fwd(w.CS, 0, 2 * Hd, flat + y.Wls, flat + y.bls, Hd, w.Ec, ws, wse, st, ix.counts + 1, cap_pairs));   // :329-330
  if (E) {
    ProfScope ps("prod_expand", st, 4.0 * B * N * N * Hd);
    hipLaunchKernelGGL(prod_expand_kernel, dim3((unsigned)((long)B * N)), dim3(256), 0, st, w.Ec, flat + y.bls, n_valid, ix.pair_prow,
                       ix.counts, E, N, Hd);
    GC_TRY(check_launch("prod_expand"));
  } else {  // compact consumers (compact.hip) read Ec / pair_prow themselves: E is never written
    GC_REQUIRE(cap_pairs >= 1, "producer: compact rows need a capacity of at least one pair");
    hipLaunchKernelGGL(prod_poison_kernel, dim3(cdiv((long)B * N * N, 256)), dim3(256), 0, st, ix, w.Ec, (long)B * N * N, Hd);
    GC_TRY(check_launch("prod_poison"));
  }
  return 0;
}

struct ProdGrads {  // caller-owned workspace of the backward pass
  float *dEc, *dCS, *dcwa, *dsfeat, *dnterm, *dCW, *dtable, *dsentF, *ddisF, *dwb, *part, *wpair, *dots, *tpart;
};

static int colsum_dyn(const float* X, const int* cnt, int C, float* part, float* out, hipStream_t st) {
  ProfScope ps("prod_colsum", st);
  hipLaunchKernelGGL(prod_colsum_dyn_kernel, dim3(cdiv(C, 64), DCS), dim3(256), 0, st, X, cnt, C, part);
  if (int e = check_launch("prod_colsum_dyn")) return e;
  hipLaunchKernelGGL(prod_colsum_fin_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, part, C, out);
  return check_launch("prod_colsum_fin");
}

int prod_bwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const unsigned char* sen, const void* pos_h,
             const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int* n_valid, const float* flat,
             ProdIdx ix, long cap_rows, long cap_pairs, ProdBufs w, const float* dE, ProdGrads g, float* dctx, float* dnode,
             float* ddis_table, float* dflat, float* ws, long wse, hipStream_t st, const float* dEc_in = nullptr) {
  const ProdLayout y = prod_layout(Hd, P);
  const long BT = (long)B * T, BN = (long)B * N, BNN = BN * N;
  const int* nrows = ix.counts;
  const int* npairs = ix.counts + 1;
  GC_REQUIRE(sizeof(float) * PW * ((size_t)ND * Hd + Hd + 1) <= 150 * 1024, "producer_bwd: hidden width %d exceeds the LDS budget of the table backward", Hd);
  // zero the atomics' targets
  GC_REQUIRE(hipMemsetAsync(g.dnterm, 0, sizeof(float) * BN * Hd, st) == hipSuccess, "producer: memset failed");
  // the two attention vectors' gradients (d w | d b, Hd + 1 floats each) are accumulated / written straight into dflat where the
  // layout keeps each pair adjacent (Hd a multiple of 4: always, for the reference's 128); through g.dwb + four copies otherwise
  const bool adj = y.ba == y.wa + Hd && y.bsa == y.wsa + Hd;
  float* const dws = adj ? dflat + y.wsa : g.dwb;             // sentence attention: accumulated by prod_sent_bwd (zeroed here)
  float* const dww = adj ? dflat + y.wa : g.dwb + Hd + 1;     // word attention: written by prod_table_fin
  GC_REQUIRE(hipMemsetAsync(dws, 0, sizeof(float) * (Hd + 1), st) == hipSuccess, "producer: memset failed");
  // linear_sentence_att: E = CS W_ls^T + b_ls on live pairs, b_ls on every other real pair
  if (dE) {
    {
      ProfScope ps("prod_gather", st);
      hipLaunchKernelGGL(prod_gather_kernel, dim3(PGRID), dim3(64 * PW), 0, st, dE, ix, g.dEc, Hd);
      GC_TRY(check_launch("prod_gather"));
    }
    // d b_ls = sum over real pairs of dE (padding pairs hold a constant 0: their gradient is ignored)
    GC_TRY(colsum(dE, n_valid ? g.wpair : nullptr, dflat + y.bls, BNN, Hd, Hd, 1, 0, 0, 0, 0, ws, st));
  } else {  // compact consumers hand over dEc itself (rows beyond the live pairs are zero)
    g.dEc = const_cast<float*>(dEc_in);
    // d b_ls: the live pairs' share is the column sum of dEc (Ec = CS W_ls^T + b_ls); the share of the pairs that ARE the bias
    // reached the caller through the consumers (compact.hip) and is added by autograd
    GC_TRY(colsum(dEc_in, nullptr, dflat + y.bls, up64(cap_pairs), Hd, Hd, 1, 0, 0, 0, 0, ws, st));
  }
  GC_TRY(linear_bwd_wx_dyn(g.dEc, w.CS, Hd, 2 * Hd, flat + y.Wls, dflat + y.Wls, g.dCS, 0, ws, wse, st, npairs, cap_pairs));
  {
    ProfScope ps("prod_sent", st);
    PROD_LAUNCH(prod_sent_bwd_kernel, Hd, dim3(PGRID), dim3(64 * PW), sizeof(float) * (Hd + 1), st, w.sfeat, w.cwa, w.nterm,
                       flat + y.wsa, w.score, g.dCS, ix, g.dcwa, g.dsfeat, g.dnterm, dws, N, Hd);
    GC_TRY(check_launch("prod_sent_bwd"));
  }
  if (!adj)
    GC_REQUIRE(hipMemcpyAsync(dflat + y.wsa, g.dwb, sizeof(float) * Hd, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                   hipMemcpyAsync(dflat + y.bsa, g.dwb + Hd, sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess,
               "producer: copy failed");
  // sentence_attention.attention_sent / attention_pos
  GC_TRY(colsum_dyn(g.dsfeat, nrows, Hd, g.part, dflat + y.bss, st));
  GC_TRY(linear_bwd_wx_dyn(g.dsfeat, w.cwa, Hd, Hd, flat + y.Wss, dflat + y.Wss, g.dcwa, 1, ws, wse, st, nrows, cap_rows));   // dcwa += dsfeat W_ss
  GC_TRY(linear_bwd_all(g.dnterm, node, BN, Hd, Hd, flat + y.Wsp, dflat + y.Wsp, dflat + y.bsp, dnode, 0, ws, wse, st));
  // linear_word_att
  GC_TRY(colsum_dyn(g.dcwa, nrows, Hd, g.part, dflat + y.blw, st));
  GC_TRY(linear_bwd_wx_dyn(g.dcwa, w.CW, Hd, 2 * Hd, flat + y.Wlw, dflat + y.Wlw, g.dCW, 0, ws, wse, st, nrows, cap_rows));
  // word attention -> score table and token states
  {
    ProfScope ps("prod_word", st);
    PROD_LAUNCH(prod_word_bwd_rows_kernel, Hd, dim3(PGRID), dim3(64 * PW), (size_t)PW * (T + Hd) * sizeof(float), st, ctx, sen, pos_h,
                       pos_t, pos_bytes, w.table, w.stats, g.dCW, ix, g.dots, N, S, T, Hd, ND);
    GC_TRY(check_launch("prod_word_bwd_rows"));
    PROD_LAUNCH(prod_word_bwd_tok_kernel, Hd, dim3((unsigned)BT), dim3(64 * PW), sizeof(float) * PW * (size_t)(Hd + ND), st, ctx, sen,
                       pos_h, pos_t, pos_bytes, w.table, w.stats, g.dCW, g.dots, ix, g.dtable, dctx, T, Hd, ND);
    GC_TRY(check_launch("prod_word_bwd_tok"));
  }
  {
    ProfScope ps("prod_table", st);
    const int nchunk = cdiv(T, TCH), rec = ND * Hd + Hd + 1;
    PROD_LAUNCH(prod_table_bwd_kernel, Hd, dim3(nchunk, B), dim3(64 * PW), sizeof(float) * PW * (size_t)rec, st, w.sentF, w.disF,
                flat + y.wa, g.dtable, ix.tmax, g.dsentF, g.tpart, T, Hd, ND);
    GC_TRY(check_launch("prod_table_bwd"));
    hipLaunchKernelGGL(prod_table_fin_kernel, dim3(cdiv(rec, 256)), dim3(256), 0, st, g.tpart, ix.tmax, B, T, nchunk, rec, ND * Hd, g.ddisF,
                       dww);
    GC_TRY(check_launch("prod_table_fin"));
  }
  if (!adj)
    GC_REQUIRE(hipMemcpyAsync(dflat + y.wa, g.dwb + Hd + 1, sizeof(float) * Hd, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                   hipMemcpyAsync(dflat + y.ba, g.dwb + 2 * Hd + 1, sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess,
               "producer: copy failed");
  // word_attention.attention_sent / attention_pos
  GC_TRY(linear_bwd_all(g.dsentF, ctx, BT, Hd, Hd, flat + y.Ws, dflat + y.Ws, dflat + y.bs, dctx, 1, ws, wse, st));   // dctx += dsentF W_s
  GC_TRY(linear_bwd_w(g.ddisF, dis_table, ND, Hd, P, dflat + y.Wp, ws, wse, st));
  GC_TRY(colsum(g.ddisF, nullptr, dflat + y.bp, ND, Hd, Hd, 1, 0, 0, 0, 0, ws, st));
  GC_TRY(linear_bwd_x(g.ddisF, ND, Hd, flat + y.Wp, P, ddis_table, 0, ws, wse, st));
  return 0;
}

// wpair[pair] = 1 for pairs of real entities, 0 otherwise (ragged batches)
__global__ void prod_wpair_kernel(const int* __restrict__ n_valid, float* __restrict__ wpair, int N, long BNN) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= BNN) return;
  const long b = p / ((long)N * N);
  const int ij = (int)(p - b * N * N), i = ij / N, j = ij - i * N;
  const int nv = min(max(n_valid[b], 0), N);
  wpair[p] = (i < nv && j < nv) ? 1.f : 0.f;
}
int prod_wpair(const int* n_valid, float* wpair, int B, int N, hipStream_t st) {
  const long BNN = (long)B * N * N;
  hipLaunchKernelGGL(prod_wpair_kernel, dim3(cdiv(BNN, 256)), dim3(256), 0, st, n_valid, wpair, N, BNN);
  return check_launch("prod_wpair");
}

}  // namespace gc

// =====================================================================================================================
// C ABI (include/gcgcn.h)
// =====================================================================================================================
#include "../../include/gcgcn.h"

namespace gc {

// live slots / live pairs of a batch (two integer atomics per workgroup: exact, order-independent)
__global__ __launch_bounds__(256) void prod_count_kernel(const unsigned char* __restrict__ sen, const int* __restrict__ n_valid,
                                                         int* __restrict__ counts, int N, int S, int T) {
  __shared__ int red[2][4];
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, NN = N * N;
  const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
  int c = 0;
  if (p < NN) {
    const int i = p / N, j = p - i * N;
    if (i < nv && j < nv) {
      const long s0 = ((long)b * NN + p) * S;
      for (int s = 0; s < S; ++s) c += sen[(s0 + s) * T] != 0;
    }
  }
  const float cs = wave_sum((float)c), cp = wave_sum(c > 0 ? 1.f : 0.f);   // <= 64 * 31: exact in fp32
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = (int)cs, red[1][threadIdx.x >> 6] = (int)cp;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(counts, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(counts + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

struct ProdBound { ProdIdx ix; ProdBufs w; ProdGrads g; float* scratch; long n_int, n_fwd, n_bwd, counts_off, prow_off, ec_off; };
// Carve the caller's three buffers (every piece starts 16-byte aligned); with null bases this only measures them.
static ProdBound prod_bind(int32_t* ibuf, float* fwd, float* bwd, bool want_bwd, int B, int N, int S, int T, int Hd, int P, int ND,
                           long cap_rows, long cap_pairs) {
  const long R = up64(cap_rows), Q = up64(cap_pairs), BNN = (long)B * N * N;
  ProdBound o;
  memset(&o, 0, sizeof(o));
  long at = 0;
  auto ti = [&](long n) { int32_t* p = ibuf ? ibuf + at : nullptr; at += (n + 3) & ~3L; return p; };
  o.ix.doc_counts = ti(2L * B), o.ix.pair_bits = ti(BNN), o.ix.pair_row0 = ti(BNN);
  o.prow_off = at;
  o.ix.pair_prow = ti(BNN);
  o.ix.pair_div = (float*)ti(BNN), o.ix.row_slot = ti(R), o.ix.prow_pair = ti(Q);
  o.counts_off = at;
  o.ix.counts = ti(4);
  o.ix.doc_off = ti(B + 1), o.ix.row_rng = ti(R), o.ix.tmax = ti(B);
  o.n_int = at;
  float* base = fwd;
  at = 0;
  auto tf = [&](long n) { float* p = base ? base + at : nullptr; at += (n + 3) & ~3L; return p; };
  o.w.sentF = tf((long)B * T * Hd), o.w.disF = tf((long)ND * Hd), o.w.table = tf((long)B * ND * T), o.w.CW = tf(R * 2 * Hd);
  o.w.stats = tf(4 * R), o.w.cwa = tf(R * Hd), o.w.sfeat = tf(R * Hd), o.w.nterm = tf((long)B * N * Hd), o.w.score = tf(2 * R);
  o.w.CS = tf(Q * 2 * Hd);
  o.ec_off = at;
  o.w.Ec = tf(Q * Hd);
  o.n_fwd = at;
  if (want_bwd) {
    base = bwd, at = 0;
    o.g.dEc = tf(Q * Hd), o.g.dCS = tf(Q * 2 * Hd), o.g.dcwa = tf(R * Hd), o.g.dsfeat = tf(R * Hd), o.g.dnterm = tf((long)B * N * Hd);
    o.g.dCW = tf(R * 2 * Hd), o.g.dtable = tf((long)B * ND * T), o.g.dsentF = tf((long)B * T * Hd), o.g.ddisF = tf((long)ND * Hd);
    o.g.dwb = tf(2 * (Hd + 1)), o.g.part = tf((long)DCS * 2 * Hd), o.g.wpair = tf(BNN), o.g.dots = tf(2 * R);
    o.g.tpart = tf((long)B * cdiv(T, TCH) * ((long)ND * Hd + Hd + 1));
    o.scratch = base ? base + at : nullptr;
    o.n_bwd = at;
  }
  (void)S, (void)P;
  return o;
}

static long prod_scratch_elems(int B, int N, int T, int Hd) {
  const long rows = (long)B * T > (long)B * N ? (long)B * T : (long)B * N;
  long a = gemm_ws_elems(rows, 2L * Hd), b = 64L * Hd * 2 * Hd + 16;   // split-K partials of the K-dynamic weight gradients
  long c = colsum_scratch_elems((long)B * N * N, Hd, 1);
  long m = a > b ? a : b;
  m = m > c ? m : c;
  return (m + 3) & ~3L;
}

}  // namespace gc

using namespace gc;

extern "C" {

int gcgcn_producer_layout(int Hd, int P, int64_t* o) {
  GC_REQUIRE(Hd > 0 && P > 0 && o, "producer_layout: bad arguments");
  const ProdLayout y = prod_layout(Hd, P);
  const long v[17] = {y.Ws, y.bs, y.Wp, y.bp, y.wa, y.ba, y.Wlw, y.blw, y.Wss, y.bss, y.Wsp, y.bsp, y.wsa, y.bsa, y.Wls, y.bls, y.total};
  for (int i = 0; i < 17; ++i) o[i] = v[i];
  return 0;
}

int gcgcn_producer_count(int B, int N, int S, int T, const uint8_t* sen, const int32_t* n_valid, int32_t* counts2, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_REQUIRE(B > 0 && N > 0 && S >= 1 && S <= 31 && T > 0 && sen && counts2, "producer_count: bad arguments");
  GC_REQUIRE(hipMemsetAsync(counts2, 0, 2 * sizeof(int32_t), st) == hipSuccess, "producer_count: memset failed");
  hipLaunchKernelGGL(prod_count_kernel, dim3(cdiv((long)N * N, 256), B), dim3(256), 0, st, sen, n_valid, counts2, N, S, T);
  return check_launch("prod_count");
}

int gcgcn_producer_sizes(int B, int N, int S, int T, int Hd, int P, int ND, int64_t cap_rows, int64_t cap_pairs, int64_t* out7) {
  int64_t* out4 = out7;
  int64_t* out3 = out4;
  GC_REQUIRE(B > 0 && N > 0 && S > 0 && T > 0 && Hd > 0 && P > 0 && ND > 0 && cap_rows >= 0 && cap_pairs >= 0 && out3,
             "producer_sizes: bad arguments");
  const ProdBound z = prod_bind(nullptr, nullptr, nullptr, true, B, N, S, T, Hd, P, ND, cap_rows, cap_pairs);
  out3[0] = z.n_int, out3[1] = z.n_fwd, out3[2] = z.n_bwd + prod_scratch_elems(B, N, T, Hd);
  out4[3] = z.counts_off;
  out4[4] = z.prow_off, out4[5] = z.ec_off, out4[6] = up64(cap_pairs);   // pair_prow inside ibuf, Ec inside fbuf, its rows   // offset (in int32) of {live rows, live pairs, over-capacity flag, 0} inside ibuf
  return 0;
}

int gcgcn_producer_fwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const uint8_t* sen, const void* pos_h,
                       const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int32_t* n_valid,
                       const float* flat, int64_t cap_rows, int64_t cap_pairs, int32_t* ibuf, float* fbuf, float* scratch,
                       int64_t scratch_elems, float* E, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && T > 0 && Hd > 0 && P > 0 && ND > 0, "producer_fwd: bad shape");
  GC_REQUIRE(ctx && sen && pos_h && pos_t && node && dis_table && flat && ibuf && fbuf, "producer_fwd: null pointer");   // E == NULL: compact
  GC_REQUIRE(cap_rows >= 0 && cap_pairs >= 0 && cap_rows < (1L << 30) && cap_pairs < (1L << 30), "producer_fwd: bad capacities");
  const ProdBound o = prod_bind(ibuf, fbuf, nullptr, false, B, N, S, T, Hd, P, ND, cap_rows, cap_pairs);
  return prod_fwd(B, N, S, T, Hd, P, ND, ctx, sen, pos_h, pos_t, pos_bytes, node, dis_table, n_valid, flat, o.ix, cap_rows, cap_pairs,
                  o.w, E, scratch, scratch ? scratch_elems : 0, (hipStream_t)stream);
}

int gcgcn_producer_bwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const uint8_t* sen, const void* pos_h,
                       const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int32_t* n_valid,
                       const float* flat, int64_t cap_rows, int64_t cap_pairs, int32_t* ibuf, float* fbuf, float* bbuf,
                       const float* dE, const float* dEc, float* dctx, float* dnode, float* ddis_table, float* dflat, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GC_REQUIRE(B > 0 && N > 0 && T > 0 && Hd > 0 && P > 0 && ND > 0, "producer_bwd: bad shape");
  GC_REQUIRE(ctx && sen && pos_h && pos_t && node && dis_table && flat && ibuf && fbuf && bbuf && (dE || dEc) && dctx && dnode &&
                 ddis_table && dflat,
             "producer_bwd: null pointer");
  const ProdBound o = prod_bind(ibuf, fbuf, bbuf, true, B, N, S, T, Hd, P, ND, cap_rows, cap_pairs);
  if (n_valid && dE) GC_TRY(prod_wpair(n_valid, o.g.wpair, B, N, st));
  return prod_bwd(B, N, S, T, Hd, P, ND, ctx, sen, pos_h, pos_t, pos_bytes, node, dis_table, n_valid, flat, o.ix, cap_rows, cap_pairs,
                  o.w, dE, o.g, dctx, dnode, ddis_table, dflat, o.scratch, prod_scratch_elems(B, N, T, Hd), st, dE ? nullptr : dEc);
}

}  // extern "C"
