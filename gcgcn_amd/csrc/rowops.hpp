// Internal launchers of the small row-wise kernels (rowops.hip) and edge kernels (edge.hip).
#pragma once
#include "common.hpp"

namespace gc {

int softmax_fwd(const float* S, const float* coladd, const int* n_valid, float* P, float* A, long rows, int N, int heads,
                Drop drop, hipStream_t st);
int softmax_bwd(const float* P, const float* dA, float* dS, long rows, int N, Drop drop, hipStream_t st);
int rowsum_inv(const float* A, float* rinv, long rows, int N, hipStream_t st);
int relu_norm_bwd(const float* dY, const float* Y, const float* rinv, float* dM, float* drow, long rows_m, int N, int H,
                  int L, int gh, int l, int first, hipStream_t st, int relu = 1);
struct ColRide;
struct DeferQueue;
int head_sum_drop_bwd(const float* dHO, float* dY, float* dXres, long M, int H, int D, Drop drop, hipStream_t st,
                      const ColRide* finish = nullptr);
int dropout(const float* x, float* y, long n, Drop drop, hipStream_t st);
int dropout_keep(unsigned char* keep, long n, Drop drop, hipStream_t st);
int rng_next(void* state, void* snaps, int count, hipStream_t st);
long colsum_scratch_elems(long R, int C, int batch);
int colsum(const float* X, const float* w, float* out, long R, int C, long ld, int batch, long sXz, long sWz, long sOz,
           int accumulate, float* scratch, hipStream_t st);
int colsum3(const float* X0, const float* w0, float* o0, long R0, int C0, long ld0, const float* X1, const float* w1,
            float* o1, long R1, int C1, long ld1, const float* X2, const float* w2, float* o2, long R2, int C2, long ld2,
            float* scratch, hipStream_t st, bool stage2 = true, long* part_off = nullptr, int* ns_out = nullptr);
int gat_fold_fwd(const float* flat, float* uvc, int D, int Dh, hipStream_t st, void* rng_state = nullptr,
                 void* rng_snaps = nullptr, int rng_count = 0);
int gat_fold_bwd(const float* flat, const float* duvc, float* dflat, int D, int Dh, hipStream_t st,
                 const float* part = nullptr, const long* part_off = nullptr, int ns = 0);
bool gat_dlogit_ok(int N);
int gat_dlogit_slices(int D);
struct GatTail;   // gat_body.hpp: the GATAttention node-score backward as passenger of edge_bwd
int gat_dlogit(const float* P, const float* dA, const float* uvc, const float* dXin, float* dlogit, float* ds, float* dX,
               int B, int N, int D, Drop drop, hipStream_t st);
int node_score_fwd(const float* X, const float* uvc, float* s, long M, int D, hipStream_t st, void* rng_state = nullptr,
                   void* rng_snaps = nullptr, int rng_count = 0);
int node_score_bwd(const float* ds, const float* uvc, const float* dXin, float* dX, long M, int D, hipStream_t st);
int mask_rows(const float* x, float* y, long M, int D, int N, const int* n_valid, Drop drop, hipStream_t st,
              const float* wlin = nullptr, float* wsum = nullptr, int H = 0);   // x == NULL: only wsum = sum_h wlin[:, h, :]

int edge_fwd(const float* E, const float* v, const int* n_valid, float* Ebar, const float* coladd, float* P, float* A,
             Drop drop, int B, int N, int D, hipStream_t st, const unsigned char* mask = nullptr);
int edge_bwd(const float* E, const float* v, const int* n_valid, const float* dlogit, const float* dEbar, float* dE,
             float* dvpart, int B, int N, int D, hipStream_t st, DeferQueue* carry = nullptr, const GatTail* tail = nullptr);
int edge_bcast(const float* dEbar, const int* n_valid, float* dE, int B, int N, int D, hipStream_t st);

// Row blocks of a ragged batch (GemmArgs::rb): out = [live count | 0 | 0 | 0 | block list], the list = all B N / 16 blocks of
// 16 entity rows, the LIVE ones first (block r of document b is live iff 16 r < n_valid[b]), each group in ascending order.
int row_blocks(const int* n_valid, int B, int N, int* out, hipStream_t st);
constexpr int ROWBLK_HDR = 4;
constexpr int ROWBLK_LIST_MAX = 512;   // live-block lists a tile body keeps in its lanes (eight registers): longer ones run dense
static inline long row_blocks_ints(int B, int N) { return ROWBLK_HDR + (long)B * N / 16; }

// mha_core.hip: fused attention core of MultiHeadAttention for N <= 64
bool mha_core_ok(int N, int D, int H, const void* Q, const void* dQ);
int mha_core_fwd(const float* Q, const int* n_valid, float* P, float* A, int B, int N, int D, int H, float alpha, Drop drop,
                 hipStream_t st);
int mha_core_bwd(const float* Q, const float* P, const float* dA, float* dQ, int B, int N, int D, int H, float alpha, Drop drop,
                 hipStream_t st);

// loss.hip: the trainer's per-document pair loss
int pair_bce_fwd(const float* logits, const float* labels, const int* n_valid, float* loss, float* part, int B, int N, int R,
                 hipStream_t st);
int pair_bce_bwd(const float* logits, const float* labels, const int* n_valid, const float* dloss, float* dlogits, int B, int N,
                 int R, hipStream_t st);

}  // namespace gc
