// Batched fp32 GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32) with the
// fused epilogues the graph-convolution stack needs.  Internal header (host launcher + args).
#pragma once
#include "common.hpp"

namespace gc {

// C[z] = epilogue( alpha * opA(A[z]) * opB(B[z]) )          z = z1 * batch2 + z2
//   a_kc : A is stored [M][K] (k contiguous, "N" form);  else A is stored [K][M] ("T" form)
//   b_kc : B is stored [N][K] (k contiguous, "T" form);  else B is stored [K][N] ("N" form)
// epilogue, in this order (every pointer optional):
//   v  = alpha * acc
//   v += add[row * ldadd + col]            (batch strided)
//   v += bias[col]
//   v += rowadd[row]                       (batch strided)
//   v *= rowscale[row]                     (batch strided)
//   v  = max(v, 0)           if relu
//   v += C[row, col]         if accumulate
//   v  = 0                   if the row is a padding row (n_valid given)
//   C[row, col] = v
//   C2[row, col] = dropout(v) + add2[row * ldadd2 + col]     if C2 (dropout index = element
//                                                             offset inside the C2 buffer)
struct GemmArgs {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  int M = 0, N = 0, K = 0;
  long lda = 0, ldb = 0, ldc = 0;
  int a_kc = 1, b_kc = 0;
  int batch1 = 1, batch2 = 1;
  long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0, sC1 = 0, sC2 = 0;
  float alpha = 1.f;
  const float* add = nullptr;
  long ldadd = 0, sAdd1 = 0, sAdd2 = 0;
  const float* bias = nullptr;
  const float* rowadd = nullptr;
  long sRa1 = 0, sRa2 = 0;
  const float* rowscale = nullptr;
  long sRs1 = 0, sRs2 = 0;
  int relu = 0;
  int accumulate = 0;
  float* C2 = nullptr;
  long ldc2 = 0, sC21 = 0, sC22 = 0;
  const float* add2 = nullptr;
  long ldadd2 = 0, sAdd21 = 0, sAdd22 = 0;
  Drop drop = {nullptr, 0, 0, 1.f};
  long drop_base = 0;  // added to the C2 element offset to form the dropout index (C2 may be a shifted view)
  // ragged documents: output row r of batch (z1, z2) belongs to document
  // z1 * nv_zdoc + r / nv_rows and is a padding row iff r % nv_rows >= n_valid[doc].
  const int* n_valid = nullptr;
  int nv_rows = 1, nv_zdoc = 0;
  const char* tag = "gemm_single";  // name seen by the per-kernel timer (gemm_kernel launches; groups: "gemm_group")
  // Ragged batches: run on the entity rows that exist.  The [B N]-row tensors stay padded in memory; `rb` lists their 16-row
  // blocks, LIVE blocks first (block r of document b is live iff 16 r < n_valid[b]), then the dead ones; *rb_n = number of
  // live blocks (device).  Interior, 16-byte aligned problems only (the launcher drops rb otherwise: the dense product is
  // equally correct, padding rows hold zeros).
  //   rb_mode 1: M is the document-row dimension -- virtual row m of this problem is row rb[m / 16] * 16 + m % 16 of A
  //              ([M][K], a_kc = 1) and of every row-indexed epilogue operand and output.  Tiles past the live blocks do no
  //              arithmetic; they store zeros when rb_zero is set (outputs that leave the block: their padding rows must be
  //              zero) and nothing otherwise (workspaces whose dead rows nobody reads).
  //   rb_mode 2: K is the document-row dimension (weight gradients; both operands [K][.], a_kc = b_kc = 0): only the live
  //              blocks are summed over.
  const int* rb = nullptr;
  const int* rb_n = nullptr;
  int rb_mode = 0, rb_zero = 0;
  // split-K workspace (optional): partial sums [splits][batch][M][N]
  float* ws = nullptr;
  long ws_elems = 0;
  // filled by the launcher
  short vecA = 0, vecB = 0;
  int splits = 1, ksplit = 0;
  int widen = 1;   // rb_mode 1, splits > 1: a launch of a ragged batch may cut K up to `widen` times finer ON THE DEVICE, with the tile
                   // workgroups its dead rows leave idle (split_width, gemm_body.hpp); the workspace holds splits * widen slabs
};

// Enqueue on `stream`.  tile: 0 or 1 = the 64x64 block body (the only one; the two 128x128 bodies of earlier rounds lost every
// A/B on this path's products and are gone).
// splits: 0 = pick (needs g.ws), 1 = none, n = split K n ways (partials in g.ws, then one
// deterministic reduce + epilogue kernel).
int gemm(const GemmArgs& g, hipStream_t stream, int tile = 0, int splits = 0);

// A column sum riding in a group launch: out[c] = sum_{r < R} X[r * ld + c]  (bias gradients).  Stage 1 (COL_RIDE_SLICES
// row slices -> part[slice][C]) runs in extra workgroups of the GEMM launch, stage 2 in extra workgroups of its
// split-K reduce: the two launches of a stand-alone column sum disappear.  part: COL_RIDE_SLICES * C floats.
constexpr int COL_RIDE_SLICES = 64;
struct ColRide {
  const float* X = nullptr;
  float* out = nullptr;
  float* part = nullptr;
  long R = 0, ld = 0;
  int C = 0;
  int ready_slices = 0;  // > 0: part[ready_slices][C] was filled by an earlier launch (X unused); only stage 2 rides, in the
                         // trailing workgroups of the group launch itself
                         // < 0: not a column sum but a sum over R interleaved blocks (heads) of ld columns, in trailing
                         // workgroups of the launch: out[k * ld + c] = sum_{h < R} X[(k * R + h) * ld + c], k * ld + c < C
};

// Up to NP independent problems carried by one launch.
template <int NP>
struct GemmGroupT {
  static constexpr int MAXP = NP;
  GemmArgs p[NP];
  int tile_begin[NP + 1];  // first workgroup of each problem, a multiple of 8 (see gemm_group_kernel)
  int tile_count[NP];      // 64x64 tiles of each problem (the domain of the XCD remap)
  int tile_first[NP];      // this launch carries tiles [tile_first, tile_first + tile_take) of the problem's (remapped) list:
  int tile_take[NP];       // a parked problem may be split over two carrying launches
  int red_begin[NP + 1];
  int nprob;
  ColRide col;  // col.X == nullptr: nothing rides
};
// MultiHeadAttention's backward core (mha_body.hpp: dA, P, Q -> dQ per (document, head)) riding in a group launch of the
// convolution that follows the attention: `count` = B * H pairs, spread evenly through the launch's tile list.
struct MhaPass {
  const float* Q = nullptr;
  const float* P = nullptr;
  const float* dA = nullptr;
  float* dQ = nullptr;
  int N = 0, D = 0, H = 0, dh = 0, kchunk = 0, count = 0;
  float alpha = 1.f;
  Drop drop = {nullptr, 0, 0, 1.f};
};
using GemmGroup = GemmGroupT<9>;   // gemm_group launches and the edge pass carrying parked problems
using GemmGroup4 = GemmGroupT<4>;  // passengers of a chain launch (kernel arguments stay small)
// col_later (optional): if the launch needs no split-K reduce, stage 2 of the riding column sum is NOT launched on its own;
// *col_later is set and the caller finishes it inside a later kernel of its own (col_ride_stage2_block).
int gemm_group(const GemmArgs* probs, int n, hipStream_t stream, const ColRide* col = nullptr, bool* col_later = nullptr,
               const MhaPass* mha = nullptr);
bool gemm_group_can_carry_mha(int dh);   // the pairs' LDS images fit the group kernel's
int gemm_group_mha_chunk(int dh);        // ... with this head-feature chunk (MhaPass::kchunk)

// ---- deferred problems ------------------------------------------------------------------------------------------
// Weight-gradient products are needed by nobody before the end of backward, while later launches of the same
// backward leave the matrix pipes idle (the edge-tensor stream of GATAttention's backward is HBM-bound).  A block can
// park such a product in the DeferQueue of its backward pass (host memory, owned by the caller: one queue per pass, no
// process-wide state); the next launch that can carry passengers takes them along as extra workgroups, unsplit (their
// full K runs inside one workgroup: no reduce launch).  Whoever parks a problem must keep its operands alive until
// gemm_flush_deferred() or a carrying launch has been enqueued.
// Only interior, 16-byte aligned shapes are parked (gemm_defer returns false otherwise: launch it now).
struct DeferQueue {
  static constexpr int CAP = 32;
  GemmArgs p[CAP];
  int done[CAP] = {};  // tiles of p[i] already carried by an earlier launch
  int n = 0;
  // second stages of riding column sums (bias gradients: ColRide::ready_slices > 0, the partials are in memory) that nobody
  // needs before the end of backward either: the next carrying edge pass sums them in a trailing workgroup, or the flush does
  static constexpr int COLCAP = 4;
  ColRide col2[COLCAP];
  int ncol2 = 0;
};
bool gemm_defer(DeferQueue* q, const GemmArgs& g);  // q == nullptr: never parks
bool gemm_defer_col2(DeferQueue* q, const ColRide& c);          // c.ready_slices > 0; false: run it now
bool gemm_take_deferred_col2(DeferQueue* q, ColRide& out);      // pops one parked second stage
// Move up to MAXP parked problems into gg (longest K first, per-problem XCD-aligned tile ranges); returns the number
// of workgroups (0: nothing parked).  flops (optional) accumulates 2MNK of the taken problems.
int gemm_take_deferred(DeferQueue* q, GemmGroup& gg, double* flops);
// Same for a chain launch: a workgroup of 512 threads runs ONE tile, its two tile teams splitting K (the partial sums meet
// in LDS).  Takes at most `max_wgs` tiles, longest problems first; a problem may be taken partially -- the rest of its
// tiles waits for a later carrier.
int gemm_take_deferred_pairs(DeferQueue* q, GemmGroup4& gg, double* flops, long max_wgs, bool small_first = false);
// Launch whatever is still parked as ordinary group launches (end of backward without a carrying launch).
int gemm_flush_deferred(DeferQueue* q, hipStream_t stream);

// The deterministic reduce + epilogue over g.splits partial slabs in g.ws (for tile kernels defined outside gemm.hip).
int splitk_reduce(const GemmArgs& g, hipStream_t st);

// ---- one dimension known only on the device ------------------------------------------------------------------
// Products over a data-dependent selection of rows (the live sentence slots / entity pairs of the edge-feature producer,
// producer.hip): the row count lives in device memory, the host only knows an upper bound `cap`.  A fixed grid of
// persistent workgroups walks the tile list computed from *cnt, so no host synchronisation and no worst-case grid.
//   dyn = 1: M = *cnt  (rows of A and of C; g.M is ignored)
//   dyn = 2: K = *cnt  (rows of both operands -- weight gradients; g.K is ignored); split-K over a factor chosen from
//            `cap`, partials through g.ws + the deterministic reduce.
// Interior shapes (N, and M or K, multiples of 64 / 32; 16-byte aligned) take the unguarded tile body with the count
// rounded UP to a multiple of 64: operand buffers must be readable that far, C writable that far (dyn = 1), and for
// dyn = 2 at least one operand's rows in [*cnt, roundup64(*cnt)) must be zero.  Other shapes take the guarded body
// with the exact count.
int gemm_dyn(const GemmArgs& g, const int* cnt, int dyn, long cap, hipStream_t st);
// The weight gradient (dyn = 2: gw) and the data gradient (dyn = 1: gx) of one Linear layer on the same device-side row count, in one
// launch (+ the weight gradient's reduce).
int gemm_dyn_pair(const GemmArgs& gw, const GemmArgs& gx, const int* cnt, long cap, hipStream_t st);
// ... and two weight gradients of the same shape over the same rows (dyn = 2 both; one reduce launch for both)
int gemm_dyn_pair_ww(const GemmArgs& ga, const GemmArgs& gb, const int* cnt, long cap, hipStream_t st);
// ... and two products whose M is the same device-side row count (dyn = 1 both)
int gemm_dyn_pair_xx(const GemmArgs& ga, const GemmArgs& gb, const int* cnt, long cap, hipStream_t st);

// Floats of split-K workspace that lets every GEMM of a [rows x cols]-sized problem split freely.
inline long gemm_ws_elems(long rows, long cols) {
  long need = 16 * rows * cols;
  const long cap = 16L << 20;   // 64 MB (r5: two split + device-widened K = 3072 data gradients of cfg 3 need 12.6 M floats)
  return need < cap ? need : cap;
}

}  // namespace gc
