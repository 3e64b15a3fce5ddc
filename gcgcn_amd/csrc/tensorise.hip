// Device-side tensorisation of packed documents: what Config.from_list_to_tensor (config/Config.py:162-233)
// builds per document with Python loops -- adjacency, sentence masks, distance-id matrices, mention-pooling weights,
// relative entity positions, labels -- written by ONE launch straight from the batch's packed records (gcgcn_amd/data.py).
// All outputs are pre-zeroed by the caller; records of entities / tokens / slots beyond the batch's (N, S, T) are clipped
// exactly as the reference's final slicing [:max_num, :max_length] does.
#include "common.hpp"
#include "../../include/gcgcn.h"

namespace gc {

// config/Config.py:106-116: 0, 1, 2, 2, 3 x4, 4 x8, ... , 10 from 512 on (the table has 1024 entries)
__device__ __forceinline__ int dis2idx(int d) {
  d = min(max(d, 0), 1023);
  return d == 0 ? 0 : min(10, 32 - __clz(d));
}
// Config.py:190-203: signed bucket of the token's distance to a mention span [a, b): left of it negative, right positive,
// inside 0; then + dis_plus
__device__ __forceinline__ int pos_id(int k, int a, int b, int dis_plus) {
  const int dl = k - a, dr = k - b;
  int v = 0;
  if (dl < 0) v = -dis2idx(-dl);
  else if (dr > 0) v = dis2idx(dr);
  return v + dis_plus;
}

struct TensArgs {
  int B, N, S, T, R, dis_plus;
  int n_slots, n_edges, n_labels, n_mentions;
  const int* slots;     // [n_slots][10]  doc, u, v, j, s0, s1, h0, h1, t0, t1
  const int* edges;     // [n_edges][3]   doc, u, v
  const int* labels;    // [n_labels][4]  doc, h, t, r
  const int* mentions;  // [n_mentions][2] start, end
  const int* mnode;     // [n_mentions][4] doc, node, mentions of that node, index of this mention inside the node
  const int* n_valid;   // [B]
  const int* first;     // [B][N] start of each entity's first mention
  float* adj;
  unsigned char *sen, *pos_h, *pos_t;
  float* node_pos;
  long long* rel;
  float* lab;
};

// workgroup ranges: [0, n_slots) one sentence slot each | edges + labels (256 per workgroup) | node_pos rows (one entity
// each) | relative positions (256 pairs per workgroup)
__global__ __launch_bounds__(256) void tensorise_kernel(const TensArgs a, int wg_el, int wg_np, int wg_rel) {
  int wg = blockIdx.x;
  const int t = threadIdx.x;
  if (wg < a.n_slots) {
    const int* s = a.slots + (long)wg * 10;
    const int b = s[0], u = s[1], v = s[2], j = s[3];
    if (u >= a.N || v >= a.N || j >= a.S) return;
    const int s0 = max(s[4], 0), s1 = min(s[5], a.T);
    const long base = ((((long)b * a.N + u) * a.N + v) * a.S + j) * a.T;
    for (int k = s0 + t; k < s1; k += 256) {
      a.sen[base + k] = 1;
      a.pos_h[base + k] = (unsigned char)pos_id(k, s[6], s[7], a.dis_plus);
      a.pos_t[base + k] = (unsigned char)pos_id(k, s[8], s[9], a.dis_plus);
    }
    return;
  }
  wg -= a.n_slots;
  if (wg < wg_el) {
    const int e = wg * 256 + t;
    if (e < a.n_edges) {
      const int* q = a.edges + (long)e * 3;
      if (q[1] < a.N && q[2] < a.N) a.adj[((long)q[0] * a.N + q[1]) * a.N + q[2]] = 1.f;
    }
    if (e < a.n_labels) {
      const int* q = a.labels + (long)e * 4;
      if (q[1] < a.N && q[2] < a.N && q[3] < a.R) a.lab[(((long)q[0] * a.N + q[1]) * a.N + q[2]) * a.R + q[3]] = 1.f;
    }
    return;
  }
  wg -= wg_el;
  if (wg < wg_np) {
    // node_pos (Config.py:170-175): for each mention IN ORDER the span is ASSIGNED 1 / length (a later mention overwrites an
    // overlapping earlier one), then the row is scaled by 1 / (number of mentions) -- in float64, stored as float32.
    // One workgroup per mention would race on overlaps: a workgroup takes one entity and replays its mentions in order.
    // Entities are found through their first mention record.
    const int m0 = wg;   // index into mnode of SOME mention; only first mentions (index 0 inside the node) do the work
    if (m0 >= a.n_mentions) return;
    const int* q = a.mnode + (long)m0 * 4;
    if (q[3] != 0) return;
    const int b = q[0], node = q[1], cnt = q[2];
    if (node >= a.N) return;
    float* row = a.node_pos + ((long)b * a.N + node) * a.T;
    const double scale = 1.0 / (double)cnt;
    for (int k = t; k < a.T; k += 256) {
      double v = 0.0;
      for (int m = 0; m < cnt; ++m) {
        const int p0 = a.mentions[(long)(m0 + m) * 2], p1 = a.mentions[(long)(m0 + m) * 2 + 1];
        if (k >= p0 && k < p1) v = 1.0 / (double)(p1 - p0);
      }
      row[k] = (float)(v * scale);
    }
    return;
  }
  wg -= wg_np;
  if (wg < wg_rel) {
    // node_relative_pos (Config.py:206-215): bucket of (first mention start of h) - (first mention start of t), signed
    const long e = (long)wg * 256 + t;
    const long NN = (long)a.N * a.N;
    if (e >= (long)a.B * NN) return;
    const int b = (int)(e / NN), ij = (int)(e - (long)b * NN), h = ij / a.N, tt = ij - h * a.N;
    const int nv = a.n_valid[b];
    if (h >= nv || tt >= nv || h == tt) return;
    const int d = a.first[(long)b * a.N + h] - a.first[(long)b * a.N + tt];
    a.rel[e] = d < 0 ? -(long long)dis2idx(-d) : (long long)dis2idx(d);
  }
}

}  // namespace gc

using namespace gc;

extern "C" int gcgcn_tensorise(int B, int N, int S, int T, int R, int dis_plus, int n_slots, const int32_t* slots, int n_edges,
                               const int32_t* edges, int n_labels, const int32_t* labels, int n_mentions, const int32_t* mentions,
                               const int32_t* mention_node, const int32_t* n_valid, const int32_t* first_start, float* adj, uint8_t* sen, uint8_t* pos_h,
                               uint8_t* pos_t, float* node_pos, int64_t* node_relative_pos, float* label_matrix, void* stream) {
  GC_REQUIRE(B > 0 && N > 0 && S > 0 && T > 0 && R > 0, "tensorise: bad shape");
  GC_REQUIRE(n_slots >= 0 && n_edges >= 0 && n_labels >= 0 && n_mentions >= 0, "tensorise: negative record count");
  GC_REQUIRE((n_slots == 0 || slots) && (n_edges == 0 || edges) && (n_labels == 0 || labels) && (n_mentions == 0 || (mentions && mention_node)),
             "tensorise: null record array");
  GC_REQUIRE(n_valid && first_start && adj && sen && pos_h && pos_t && node_pos && node_relative_pos && label_matrix, "tensorise: null output");
  TensArgs a;
  a.B = B, a.N = N, a.S = S, a.T = T, a.R = R, a.dis_plus = dis_plus;
  a.n_slots = n_slots, a.n_edges = n_edges, a.n_labels = n_labels, a.n_mentions = n_mentions;
  a.slots = slots, a.edges = edges, a.labels = labels, a.mentions = mentions, a.mnode = mention_node, a.n_valid = n_valid, a.first = first_start;
  a.adj = adj, a.sen = sen, a.pos_h = pos_h, a.pos_t = pos_t, a.node_pos = node_pos, a.rel = (long long*)node_relative_pos;
  a.lab = label_matrix;
  const int wg_el = cdiv(n_edges > n_labels ? n_edges : n_labels, 256), wg_np = n_mentions, wg_rel = cdiv((long)B * N * N, 256);
  const long grid = (long)n_slots + wg_el + wg_np + wg_rel;
  GC_REQUIRE(grid < (1L << 31), "tensorise: too many records for one launch");
  if (grid == 0) return 0;
  hipLaunchKernelGGL(tensorise_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, wg_el, wg_np, wg_rel);
  return check_launch("tensorise");
}
