/* gcgcn.h -- C ABI of libgcgcn_hip.so: the CAGGC + MAGGC graph-convolution hot path of GCGCN
 * as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (Huiweizhou/GCGCN) has no FFI: its boundary for this path is the nn.Module
 * protocol of five Python classes in models/GCGCN_glove.py:18-168 (code-identical copies in
 * models/GraphCNN_multihead_bert_gate_cls.py:18-172).  Each entry point below replaces the
 * forward (or autograd backward) of one of those classes and cites it.  gcgcn_amd/functional.py
 * binds these with ctypes; INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (unless typed otherwise), row-major, dense;
 *   - B documents, N entity slots per document, D feature width, L sub-layers, H heads,
 *     gh = D / L, dh = D / H, M = B * N;
 *   - n_valid: optional int32[B]; entities >= n_valid[b] are padding (their rows of X must be
 *     zero; their outputs and gradients are zero).  NULL = every document has N entities;
 *   - the caller owns all memory (inputs, outputs, saved-for-backward, workspace); the library
 *     keeps no device memory and never synchronises: work is enqueued on `stream`
 *     (a hipStream_t) and is hipGraph-capturable;
 *   - return value 0 = ok, otherwise gcgcn_last_error() describes the failure (thread-local);
 *   - dropout: rng_snap is a device int64[2] {seed, counter} written by gcgcn_rng_next at
 *     forward time; NULL (or p == 0) = eval mode.  Backward takes the same snapshot.
 *
 * Parameter buffers ("flat") hold one block's parameters contiguously in the layout the
 * kernels want; gcgcn_*_layout report offsets (in floats).  Gradient buffers use the same
 * layout, so a block's gradient is one contiguous all-reduce bucket.
 */
#ifndef GCGCN_H
#define GCGCN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int gcgcn_version(void);            /* ABI version, currently 7 */
const char* gcgcn_last_error(void); /* message of the last failing call on this thread */

/* Run-time switches for A/B tests.  "chain": 1 (default) = the per-(doc, head) products of a conv run inside the
 * chain kernels, 0 = one batched launch per product.  "mha_core": 1 (default) = graphs of N <= 64 entities take the
 * one-workgroup-per-(doc, head) attention kernels, 0 = batched GEMM + row softmax for every N.  Results are identical
 * up to fp32 summation order.  The other names -- "head_v1" (-1 = by problem size), "head_bil3", "head_bil3_bwd",
 * "head_dw3", "chain_s", "chain_fuse", "chain_carry", "gat_ride", ... -- select kernel generations; each also reads the
 * environment variable GCGCN_<NAME> once when nobody set it (DESIGN.md section 6 lists them). */
int gcgcn_set_option(const char* name, int value);

/* ---- ragged batches: the entity rows that exist (ABI v5; v7: one list serves every product) ------------------------------- */
/* The reference runs one UNPADDED document per call (config/Config.py:339-354): its products have n rows.  A padded batch
 * [B, N, .] with n_valid has sum_b n_b real rows of B N.  gcgcn_row_blocks lists the 16-row blocks of the [B N]-row tensors,
 * the LIVE ones first (block r of document b is live iff 16 r < n_valid[b]; ascending), the dead ones behind them:
 *   out = int32[gcgcn_row_blocks_ints(B, N)] = {live blocks, 0, 0, 0 | B N / 16 blocks};
 * N must be a multiple of 16.  Given as `rowblk` to gcgcn_gcn_fwd / _bwd and gcgcn_mha_fwd / _bwd (NULL = dense), the
 * node-phase products of the block -- X WnX, Ebar We, X Wq, the output projection, every data gradient -- run on the live
 * blocks only (four to a 64-row tile), and every weight gradient (K = the document rows) sums over the live blocks only (two
 * to a k-tile; lists of up to 512 blocks, B N <= 8192 -- longer ones run dense); the tensors keep their padded layout,
 * outputs that leave the block (out, dX, dEbar, Q) are zero on
 * padding rows as always.  Everything stays on the device (no host read; the list is rebuilt by every replay of a captured
 * step).  A block whose shape the column-strip chain kernels do not serve (N > 64, or a width they are not instantiated for)
 * ignores the list and computes every row, as does a forward / backward pair without n_valid.  The SAME list must be given
 * to a block's forward and backward. */
int64_t gcgcn_row_blocks_ints(int B, int N);
int gcgcn_row_blocks(int B, int N, const int32_t* n_valid, int32_t* out, void* stream);

/* ---- per-kernel timing for roofline reports (bench.py) ------------------------------------------- */
/* While active, every kernel launch whose name starts with kernel_prefix ("edge_bwd", "edge_fwd",
 * "edge_bcast", "gemm", "softmax", ...) is bracketed by hipEvents on its launch stream (at most
 * `capacity` launches).  prof_stop synchronises those events and returns their summed duration, their
 * count and (optional) the work they did: executed flops for GEMM kernels, algorithmic HBM bytes for the
 * edge kernels.  Not thread-safe; keep it off while capturing a hipGraph. */
int gcgcn_prof_start(const char* kernel_prefix, int capacity);
int gcgcn_prof_enable(int on); /* pause (0) / resume (1) between prof_start and prof_stop: sample some steps only */
int gcgcn_prof_stop(double* total_ms, int* launches, double* work);

/* ---- dropout RNG state (replaces torch's global CUDA generator used by nn.Dropout) ------- */
/* snaps[i] <- {state.seed, state.counter + i} for i < count; state.counter += count.
 * state: device int64[2] {seed, counter}; snaps: device int64[2 * count].  One launch serves `count`
 * dropout-using forward calls (each takes its own 2-word snapshot). */
int gcgcn_rng_next(void* state, void* snaps, int count, void* stream);
/* keep[i] = 1 iff element i of dropout site (snap, salt, p) is kept.  Test/debug aid. */
int gcgcn_dropout_keep(uint8_t* keep, int64_t n, const void* rng_snap, uint64_t salt, float p, void* stream);
/* y = dropout(x); calling it on a gradient with the same snapshot is the backward.
 * Replaces self.dropout in the hop glue, GCGCN_glove.py:341. */
int gcgcn_dropout(const float* x, float* y, int64_t n, const void* rng_snap, uint64_t salt, float p, void* stream);

/* salts of the dropout sites inside the blocks */
#define GCGCN_SALT_GAT 0x47415431ull
#define GCGCN_SALT_MHA 0x4d484131ull
#define GCGCN_SALT_GCN 0x47434e31ull
#define GCGCN_SALT_GLUE 0x474c5531ull

/* ---- GATAttention (CAGGC adjacency)  GCGCN_glove.py:144-168 -------------------------------- */
/* D = att_input_dim (width of node_feat and edge_feat), Dh = hidden_dim (rows of the three nn.Linear(att_input_dim,
 * hidden_dim), glove:148-150; wt = nn.Linear(3 * hidden_dim, 1), glove:151).
 * flat = [W_h Dh*D | b_h Dh | W_t Dh*D | b_t Dh | W_r Dh*D | b_r Dh | wt 3Dh | wt_bias 1]
 * out[0..7] = offsets of those eight pieces, out[8] = total floats. */
int gcgcn_gat_layout(int D, int Dh, int64_t* out9);

/* forward(node_feat X[B,N,D], edge_feat E[B,N,N,D], mask ignored as in the reference):
 *   P[B,N,N]    softmax_j(u.x_j + v.e_ij + c)            (saved for backward)
 *   A[B,N,N]    dropout(P); may be NULL when rng_snap is NULL (then A == P)
 *   Ebar[B,N,D] mean_j E[b,i,j,:]  -- by-product of the single pass over E, consumed by the
 *               GraphConvolution that follows (GCGCN_glove.py:40-41 commuted)
 *   uvc[2D+1], s[B,N]  folded projection and node scores (saved for backward) */
/* rng_state / rng_snaps / rng_count: optional (NULL, NULL, 0).  When given, the call also performs
 * gcgcn_rng_next(rng_state, rng_snaps, rng_count) inside its first kernel, BEFORE anything reads rng_snap (which may
 * point into rng_snaps): a hop loop draws the snapshots of all its dropout sites without a launch of its own.
 * mask: NULL (default: the reference DISCARDS its masked_fill result, glove:163-164, so its mask is a no-op), or a
 * uint8/bool [B,N,N] whose non-zero entries get energy -100000 before the softmax: the paper-faithful partially
 * connected adjacency, an explicit opt-in (GATAttention(apply_mask=True)).  The backward needs no mask: masked entries
 * have P == 0 exactly, hence a zero logit gradient.
 * uvc_valid != 0: uvc already holds the folded projection of THESE parameters (a function of flat only: the caller may
 * keep it while flat is unchanged -- inference, several documents per optimiser step); the fold kernel is skipped. */
int gcgcn_gat_fwd(int B, int N, int D, int Dh, const float* X, const float* E, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, float* uvc, float* s, float* P, float* A, float* Ebar, void* rng_state,
                  void* rng_snaps, int rng_count, const uint8_t* mask, int uvc_valid, void* stream);

/* backward.  dA[B,N,N], dEbar[B,N,D] (NULL = zero), dX_in[B,N,D] (NULL = zero: gradient node_feat has
 * already collected from its other consumers -- the convolution of the same hop -- added here instead of
 * by a separate kernel) -> dX[B,N,D], dE[B,N,N,D] (NULL = not wanted), dflat.  Workspace: dlogit[B,N,N], ds[B,N], dvpart[B*N*D], duvc[2D+1],
 * scratch[gcgcn_gat_bwd_scratch(B,N,D)]. */
int64_t gcgcn_gat_bwd_scratch(int B, int N, int D);
/* defer_queue: NULL, or the gcgcn_defer queue of this backward pass (below): weight-gradient products parked in it by
 * gcgcn_gcn_bwd / gcgcn_mha_bwd calls that ran earlier in the pass ride in this call's HBM-bound edge pass. */
int gcgcn_gat_bwd(int B, int N, int D, int Dh, const float* X, const float* E, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, const float* uvc, const float* P, const float* dA, const float* dEbar,
                  const float* dX_in, float* dX, float* dE, float* dflat, float* dlogit, float* ds, float* dvpart,
                  float* duvc, float* scratch, void* defer_queue, void* stream);

/* ---- edge mean alone (MAGGC hop: E enters only through GraphConv's mean, glove:40-41) ------ */
int gcgcn_edge_mean_fwd(int B, int N, int D, const float* E, const int32_t* n_valid, float* Ebar, void* stream);
int gcgcn_edge_mean_bwd(int B, int N, int D, const float* dEbar, const int32_t* n_valid, float* dE, void* stream);

/* ---- MultiHeadAttention (MAGGC adjacency)  GCGCN_glove.py:122-142 -------------------------- */
/* flat = [Wq D*D (rows h*dh.. = linears_q.h.weight) | bq D]; linears_k.* never enter (the
 * reference projects keys with linears_q, glove:136-137).  out = {oWq, obq, total}. */
int gcgcn_mha_layout(int D, int64_t* out3);
/* forward(node_feat X[B,N,D]) -> A[B,H,N,N] = dropout(P), P = softmax(Q_h Q_h^T / sqrt(dh));
 * saved: Q[B,N,D], P[B,H,N,N].  A may be NULL when rng_snap is NULL.
 * scratch[gcgcn_mha_scratch(B,N,D)] floats (split-K partials, column-sum partials); may be NULL
 * (then no GEMM is split). */
int64_t gcgcn_mha_scratch(int B, int N, int D);
int gcgcn_mha_fwd(int B, int N, int D, int H, const float* X, const int32_t* n_valid, const float* flat,
                  const void* rng_snap, float p, float* Q, float* P, float* A, float* scratch, const int32_t* rowblk, void* stream);
/* backward.  dX_in[B,N,D] (NULL = zero) as in gcgcn_gat_bwd.  Workspace: dS[B,H,N,N], dQ[B,N,D],
 * scratch[gcgcn_mha_scratch].  defer_queue (may be NULL): dWq is parked as described at gcgcn_gcn_bwd, and so is the second
 * stage of dbq's column sums (its 64 partial rows live in `scratch`: a later gcgcn_gat_bwd given the same queue sums them in a
 * trailing workgroup of its edge pass, or the flush does) -- keep X, dQ, scratch and dflat alive until the flush. */
int gcgcn_mha_bwd(int B, int N, int D, int H, const float* X, const float* flat, const void* rng_snap, float p,
                  const float* Q, const float* P, const float* dA, const float* dX_in, float* dX, float* dflat,
                  float* dS, float* dQ, float* scratch, void* defer_queue, int core_done, const int32_t* rowblk, void* stream);
/* core_done = 1: dQ already holds the attention core's gradient (gcgcn_gcn_bwd with a gcgcn_mha_hook computed it) */

/* ---- GraphConvolution (H = 1) / MultiGraphConvolution  GCGCN_glove.py:52-120 --------------- */
/* flat = [WnX D x H*D | We D x H*D | Wd | Wlin D x H*D | blin D]
 *   WnX[:, (h*L+l)*gh ..] = graphconv.{h*L+l}.weights_node[:D]     (X part of the dense connection)
 *   We [:, (h*L+l)*gh ..] = graphconv.{h*L+l}.weights_edge
 *   Wd: for h, for l = 1..L-1: graphconv.{h*L+l}.weights_node[D:]  ([l*gh, gh], the Y_0..Y_{l-1} part)
 *   Wlin, blin = linear_layer.weight / .bias
 * out = {oWnX, oWe, oWd, oWlin, oblin, total, wd_floats_per_head}. */
int gcgcn_gcn_layout(int D, int L, int H, int64_t* out7);
/* forward(node_feat X[B,N,D], mean edge feature Ebar[B,N,D], adjacency A[B,H,N,N]) -> out[B,N,D].
 * saved for backward: Pn, Y, HO (each [B,N,H*D]) and rinv[B,H,N].  Workspace: G[B,N,H*D],
 * scratch[gcgcn_gcn_scratch(B,N,D,H)] (split-K / column-sum partials; NULL = never split). */
int64_t gcgcn_gcn_scratch(int B, int N, int D, int H);
/* An edge-tensor pass that depends on nothing the block computes may ride along with it: the NEXT hop's
 * edge mean (forward; == gcgcn_edge_mean_fwd(B,N,D,in=E,n_valid,out=Ebar)) and its backward
 * (== gcgcn_edge_mean_bwd(B,N,D,in=dEbar,n_valid,out=dE)).  The dependent per-(doc, head) part of a block
 * is latency-bound and, for few documents x heads, occupies a fraction of the chip; the HBM-bound pass
 * runs in extra workgroups of that same launch (or as its own launch when that is not possible).  The
 * result is complete when the call's work is.  NULL = nothing rides. */
typedef struct gcgcn_edge_ride {
  int32_t B, N, D;
  const float* in;
  const int32_t* n_valid; /* may be NULL */
  float* out;
} gcgcn_edge_ride;
/* out_rng_snap / out_p: the hop's output dropout x <- dropout(block(x)) (glove:341, site GCGCN_SALT_GLUE) applied
 * in the epilogue of the block's last product; NULL / 0 = the plain block output.  gcgcn_gcn_bwd given the same
 * pair takes dout = gradient of the DROPPED output.
 * wsum[D,D] (may be NULL; H > 1 only): sum over heads of linear_layer.weight's column blocks, a function of the parameters
 * alone that rides in this call's first launch; gcgcn_gcn_bwd given the same buffer skips the launch that would sum it. */
/* A whole MAGGC hop in one call pair (glove:336-337: MultiHeadAttention, then MultiGraphConvolution on its adjacencies).
 * The attention's work rides inside the convolution's launches: forward, the query projection Q = X Wq^T + bq is one more
 * problem of the first group launch (node / edge terms), the attention core (Q -> P, A) follows it and the chain reads
 * A (or P when there is no dropout) -- the `A` argument of gcgcn_gcn_fwd is ignored; backward, the attention core
 * (dA, P, Q -> dQ) runs as passenger workgroups of the convolution's last group launch, and gcgcn_mha_bwd is then called
 * with core_done = 1 for the rest (dX = dQ Wq + dX_in, dWq, dbq).  Needs a graph of at most 64 entities and head width
 * D / H a multiple of 4 (gcgcn_maggc_fusable); results equal the separate calls bit for bit.  Where a chain kernel that
 * keeps its pair in LDS serves the shape, the forward core runs in that kernel's prologue instead of a launch; heads wider
 * than 32 features keep the backward core as a launch of its own (its scratch does not fit a tile workgroup's LDS). */
typedef struct gcgcn_mha_hook {
  const float* flat_q;  /* MultiHeadAttention's flat parameters [Wq D*D | bq D] (gcgcn_mha_layout) */
  float* Q;             /* [B,N,D]   forward: out; backward: in */
  float* P;             /* [B,H,N,N] forward: out; backward: in */
  float* A;             /* [B,H,N,N] forward: out (NULL without dropout) */
  float* dQ;            /* [B,N,D]   backward: out */
  const void* rng_snap; /* the attention's dropout snapshot (site GCGCN_SALT_MHA), NULL = eval */
  float p;
} gcgcn_mha_hook;
int gcgcn_maggc_fusable(int N, int D, int H);
int gcgcn_gcn_fwd(int B, int N, int D, int L, int H, const float* X, const float* Ebar, const float* A,
                  const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const void* out_rng_snap,
                  float out_p, float* out, float* Pn, float* Y, float* HO, float* rinv, float* G, float* wsum, float* scratch,
                  const gcgcn_edge_ride* ride, const gcgcn_mha_hook* mha, const int32_t* rowblk, void* stream);
/* backward.  dout[B,N,D] -> dX, dEbar [B,N,D], dA[B,H,N,N], dflat.
 * Workspace: W1, W2, W3 (each [B,N,H*D]), drow[B,H,N], dXres[B,N,D], dout_m[B,N,D] (only used
 * when n_valid != NULL or out_rng_snap != NULL), scratch[gcgcn_gcn_scratch]. */
int gcgcn_gcn_bwd(int B, int N, int D, int L, int H, const float* X, const float* Ebar, const float* A,
                  const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const void* out_rng_snap,
                  float out_p, const float* Pn, const float* Y, const float* HO, const float* rinv, const float* wsum,
                  const float* dout, float* dX, float* dEbar,
                  float* dA, float* dflat, float* W1, float* W2, float* W3, float* drow, float* dXres, float* dout_m,
                  float* scratch, const gcgcn_edge_ride* ride, const gcgcn_mha_hook* mha, void* defer_queue, const int32_t* rowblk,
                  void* stream);
/* defer_queue != NULL: the block's weight-gradient products (dWlin, dWnX, dWe, dWd: nobody needs them before the end
 * of backward) are not launched by this call but parked in that queue -- a small host-side object the caller creates
 * per backward pass (no process-wide state: concurrent passes, models and devices never share one).  A later
 * gcgcn_gat_bwd (or a long gcgcn_gcn_bwd chain launch) given the same queue carries them as extra workgroups of a launch
 * whose matrix pipes are idle.  Until then the caller must keep X, Ebar, Y, HO, dout (dout_m), W2, W3 and dflat of this
 * call alive, and must call gcgcn_defer_flush(queue, stream) at the end of the pass (it launches what is still parked;
 * no-op otherwise).  The parked parts of dflat are complete only after that.  Destroying a queue forgets what is
 * parked in it without launching anything (a pass that failed half-way). */
void* gcgcn_defer_create(void);
void gcgcn_defer_destroy(void* queue);
int gcgcn_defer_count(const void* queue);
int gcgcn_defer_flush(void* queue, void* stream);

/* ---- GraphConv, the leaf layer  GCGCN_glove.py:18-50 ------------------------------------------ */
/* forward(inputs X[B,N,Din], mean edge feature Ebar[B,N,De], adjacency A[B,N,N]):
 *   out[B,N,Dout] = (Ebar We + A X Wn (+ bias)) / (rowsum(A) + [rowsum == 0])
 * We[De,Dout] = weights_edge, Wn[Din,Dout] = weights_node, bias[Dout] or NULL.  Saved: T = X Wn, rinv[B,N].
 * (Inside GraphConvolution / MultiGraphConvolution this layer is fused into gcgcn_gcn_fwd.) */
int gcgcn_graphconv_fwd(int B, int N, int Din, int De, int Dout, const float* X, const float* Ebar, const float* A,
                        const float* We, const float* Wn, const float* bias, float* out, float* T, float* rinv,
                        float* scratch, void* stream);
/* backward.  Workspace: dS, dT [B,N,Dout], drow[B,N], scratch[gcgcn_gcn_scratch(B,N,max(Din,Dout),1)].
 * dbias may be NULL. */
int gcgcn_graphconv_bwd(int B, int N, int Din, int De, int Dout, const float* X, const float* Ebar, const float* A,
                        const float* We, const float* Wn, const float* out, const float* T, const float* rinv,
                        const float* dout, float* dX, float* dEbar, float* dA, float* dWe, float* dWn, float* dbias,
                        float* dS, float* dT, float* drow, float* scratch, void* stream);

/* ---- trainer loss  config/Config.py:302,355-366 ---------------------------- */
/* loss[b] = sum_{h != t < n} mean_r BCE(sigmoid(logits[b,h,t,r]), labels[b,h,t,r]) / (n^2 - n), n = n_valid[b] or N:
 * what the trainer's double Python loop of nn.BCELoss calls computes per document (ATen arithmetic: logs clamped at
 * -100).  logits, labels [B,N,N,R]; workspace part[B*N].  A document with n < 2 yields NaN like the reference. */
int gcgcn_pair_bce_fwd(int B, int N, int R, const float* logits, const float* labels, const int32_t* n_valid, float* loss,
                       float* part, void* stream);
/* dlogits[B,N,N,R] = dloss[b] * d loss[b] / d logits (dloss NULL = ones); zero on the diagonal and on padding. */
int gcgcn_pair_bce_bwd(int B, int N, int R, const float* logits, const float* labels, const int32_t* n_valid,
                       const float* dloss, float* dlogits, void* stream);

/* ---- edge-feature producer  GCGCN_glove.py:171-214 and the call sequence :300-330 ------------ */
/* Builds E = context_sent_att[B,N,N,Hd] -- the edge tensor the blocks above consume -- from the token states:
 * WordAttention (glove:171-190) on the head- and tail-side distance embeddings, Linear(2Hd, Hd) (linear_word_att),
 * SentenceAttention (glove:193-214) on the head- and tail-side node embeddings, Linear(2Hd, Hd) (linear_sentence_att).
 *   ctx[B,T,Hd]        token states (context_output, glove:292)
 *   sen[B,N,N,S,T]     uint8 / bool: token t belongs to sentence slot s of pair (i, j)   (sen_matrix)
 *   pos_h, pos_t       [B,N,N,S,T] distance ids 0..ND-1 (pos_matrix_h / _t), element width pos_bytes in {8, 4, 1}
 *   node[B,N,Hd]       entity features (node_feat);  dis_table[ND,P] = dis_embed.weight
 * Nothing of size [N,N,S,T,Hd] is materialised (the reference's 2.3 GB per document): the word score is a [ND,T] table per
 * document; only LIVE sentence slots -- those containing token 0, the only ones the reference's padding test
 * ~sen_matrix[...,0:1] (glove:305) does not zero out, forward and backward -- are computed, compacted into rows whose
 * count stays on the device.  The reference's divisor (number of PADDED slots + 1e-10, glove:205,212) is reproduced.
 * flat = [word_attention.attention_sent W,b | .attention_pos W,b | .attention_all w,b | linear_word_att W,b |
 *         sentence_attention.attention_sent W,b | .attention_pos W,b | .attention_all w,b | linear_sentence_att W,b],
 * every matrix in the reference's [out, in] layout; out17 = the 16 offsets + total.
 * Capacities: cap_rows >= live slots, cap_pairs >= pairs with a live slot (gcgcn_producer_count reports both; anything
 * beyond a capacity is NOT computed: ibuf[sizes[3] + 2] becomes 1 and every real pair of E is written as NaN, so an
 * undersized capacity is loud even inside a captured hipGraph).  Buffers (caller-owned): ibuf int32[sizes[0]], fbuf
 * float[sizes[1]] (written by forward, read by backward), bbuf float[sizes[2]] (backward workspace) from
 * gcgcn_producer_sizes.  Two small sums of the backward -- the gradient of the per-entity node terms and of the sentence
 * attention's vector / bias -- are scatter-added with fp32 atomics (reproducible up to summation order); every other sum
 * (ctx, the score table, the weight gradients) is computed by its owner in a fixed order. */
int gcgcn_producer_layout(int Hd, int P, int64_t* out17);
int gcgcn_producer_count(int B, int N, int S, int T, const uint8_t* sen, const int32_t* n_valid, int32_t* counts2, void* stream);
int gcgcn_producer_sizes(int B, int N, int S, int T, int Hd, int P, int ND, int64_t cap_rows, int64_t cap_pairs, int64_t* out7);
/* out7 = {ibuf elements, fbuf elements, bbuf elements, offset of int32 {live rows, live pairs, over capacity, 0} in ibuf,
 *         offset of pair_prow int32[B,N,N] in ibuf, offset of the compact rows Ec[rows, Hd] in fbuf, rows of Ec}
 * Compact mode: gcgcn_producer_fwd with E == NULL leaves E unwritten; the hop's graph blocks read Ec / pair_prow / the bias
 * of linear_sentence_att directly (gcgcn_*_compact below) and hand gcgcn_producer_bwd dEc instead of dE (dE == NULL; rows
 * of dEc beyond the live pairs must be zero; the bias gradient is produced by the consumers, not here). */
int gcgcn_producer_fwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const uint8_t* sen, const void* pos_h,
                       const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int32_t* n_valid,
                       const float* flat, int64_t cap_rows, int64_t cap_pairs, int32_t* ibuf, float* fbuf, float* scratch,
                       int64_t scratch_elems, float* E, void* stream);
/* dE[B,N,N,Hd] -> dctx[B,T,Hd], dnode[B,N,Hd], ddis_table[ND,P], dflat (layout of flat). */
int gcgcn_producer_bwd(int B, int N, int S, int T, int Hd, int P, int ND, const float* ctx, const uint8_t* sen, const void* pos_h,
                       const void* pos_t, int pos_bytes, const float* node, const float* dis_table, const int32_t* n_valid,
                       const float* flat, int64_t cap_rows, int64_t cap_pairs, int32_t* ibuf, float* fbuf, float* bbuf,
                       const float* dE, const float* dEc, float* dctx, float* dnode, float* ddis_table, float* dflat, void* stream);

/* ---- consumers of the compact rows: a hop's graph blocks without a dense E (compact.hip) -------------------------------
 * e_ij = Ec[prow[b,i,j]] where prow >= 0, else bias.  Mean-only hop (MultiGraphConvolution's edge term): Ebar = mean_j e_ij;
 * backward dEc[prow] = dEbar_i / n, dbias = sum (dead pairs of row)/n dEbar_i.  Attention hop: gcgcn_gat_fwd / _bwd with the
 * edge pass reading e_ij (no opt-in mask); dEc and dbias instead of dE.  rowbuf: float[2 B N]; scratch of the attention
 * backward: gcgcn_gat_bwd_compact_scratch(B,N,D) floats.  D <= 512.  Every sum in a fixed order. */
int gcgcn_edge_mean_fwd_compact(int B, int N, int D, const float* Ec, const int32_t* prow, const float* bias, const int32_t* n_valid,
                                float* Ebar, void* stream);
int gcgcn_edge_mean_bwd_compact(int B, int N, int D, const int32_t* prow, const int32_t* n_valid, const float* dEbar, float* dEc,
                                float* dbias, float* rowbuf, void* stream);
int gcgcn_gat_fwd_compact(int B, int N, int D, int Dh, const float* X, const float* Ec, const int32_t* prow, const float* bias,
                          const int32_t* n_valid, const float* flat, const void* rng_snap, float p, float* uvc, float* s, float* P,
                          float* A, float* Ebar, void* rng_state, void* rng_snaps, int rng_count, int uvc_valid, void* stream);
int64_t gcgcn_gat_bwd_compact_scratch(int B, int N, int D);
int gcgcn_gat_bwd_compact(int B, int N, int D, int Dh, const float* X, const float* Ec, const int32_t* prow, const float* bias,
                          const int32_t* n_valid, const float* flat, const void* rng_snap, float p, const float* uvc, const float* P,
                          const float* dA, const float* dEbar, const float* dX_in, float* dX, float* dEc, float* dbias, float* dflat,
                          float* dlogit, float* ds, float* dvpart, float* duvc, float* scratch, void* stream);

/* ---- classifier head  GCGCN_glove.py:306-307, 344-358 ------------------------------------------------ */
/* logits[B,N,N,R] = bili_layer_01(eh, et) + classification_layer_01(cat(eh, et)) with
 *   eh[i,j] = tanh(dense_layer(cat(feats[j], ner_emb[type[j]], dis_embed[dis_plus + rel[i,j]])))   (column entity)
 *   et[i,j] = tanh(dense_layer(cat(feats[i], ner_emb[type[i]], dis_embed[dis_plus - rel[i,j]])))   (row entity)
 * feats: host array of nf device pointers, the model's node_feats list (each [B,N,Hd]); node_type int64[B,N] in 0..6;
 * node_relative_pos int64[B,N,N]; ner_emb[7,Pt] (padding_idx 0: its row 0 gets no gradient), dis_table[ND,Pr].
 * The hidden width of eh / et is 128 (hard-coded in the reference, glove:234); R <= 128.
 * flat = [dense_layer W [128, nf Hd + Pt + Pr] | b | classification_layer_01 W [R,256] | b | bili_layer_01 b [R] |
 *         bili_layer_01 W [R,128,128]], pieces 16-byte aligned; out7 = 6 offsets + total.
 * The bilinear form runs on the fp32 MFMA with the per-pair outer product eh (x) et generated in registers: no
 * [pairs, 16384] operand and no [N,N,F] concatenations exist.  Buffers: fbuf float[sizes[0]] (forward, kept for backward),
 * bbuf float[sizes[1]] (backward workspace), ibuf int32[sizes[2]] (pair index of a ragged batch; only touched when n_valid is
 * given) from gcgcn_head_sizes.  Deterministic (no atomics).
 * n_valid (ABI v5): the reference runs these lines on ONE unpadded document (n x n pairs).  With n_valid the pair passes run on
 * the pairs that exist only -- rows of eh / et / their gradients compacted on the device (row off[b] + i n_b + j, off = prefix
 * sums of n_b^2; no host read, capturable), 2 x 128 x 128 x R flops per existing pair and pass instead of per pair slot of the
 * padded batch -- and logits of pairs with a padding entity are exactly zero.  n_valid == NULL: every slot is computed. */
int gcgcn_head_layout(int Hd, int nf, int Pt, int Pr, int R, int64_t* out7);
int gcgcn_head_sizes(int B, int N, int R, int ND, int64_t* out3);
int gcgcn_head_fwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats,
                   const int64_t* node_type, const int64_t* node_relative_pos, const float* ner_emb, const float* dis_table,
                   const int32_t* n_valid, const float* flat, float* fbuf, int32_t* ibuf, float* logits, void* stream);
/* dlogits[B,N,N,R] (entries of padding entities ignored when n_valid is given) -> dfeats (nf pointers, [B,N,Hd] each),
 * dner_emb[7,Pt], ddis_table[ND,Pr], dflat.  n_valid / fbuf / ibuf: the forward call's. */
int gcgcn_head_bwd(int B, int N, int Hd, int nf, int Pt, int Pr, int R, int ND, int dis_plus, const float* const* feats,
                   const int64_t* node_type, const int64_t* node_relative_pos, const float* ner_emb, const float* dis_table,
                   const int32_t* n_valid, const float* flat, float* fbuf, int32_t* ibuf, float* bbuf, const float* dlogits,
                   float* const* dfeats, float* dner_emb, float* ddis_table, float* dflat, void* stream);

/* ---- device-side tensorisation of packed documents  config/Config.py:162-233 -------------------------- */
/* Expands the packed records of a batch (gcgcn_amd/data.py; all int32, device memory) into the dense inputs of the
 * reference's forward, batched and padded to (N entities, S sentence slots, T tokens), in ONE launch.  Outputs must be
 * zero-filled by the caller.
 *   slots[n_slots][10]   doc, u, v, slot j, sentence start, end, head mention start, end, tail mention start, end
 *                        -> sen[B,N,N,S,T] (uint8 0/1), pos_h / pos_t [B,N,N,S,T] (uint8 distance ids: dis_plus +- the
 *                        bucket dis2idx of the distance to the head / tail mention, Config.py:106-116, 190-203)
 *   edges[n_edges][3]    doc, u, v -> adj[B,N,N] = 1                                     (Config.py:183)
 *   labels[n_labels][4]  doc, h, t, r -> label_matrix[B,N,N,R] = 1
 *   mentions[n][2], mention_node[n][4] = (doc, entity, mentions of that entity, index inside the entity), sorted by
 *                        (doc, entity, index) -> node_pos[B,N,T] (each mention span ASSIGNED 1 / length in order, the row
 *                        scaled by 1 / mentions, float64 arithmetic, Config.py:170-175) and node_relative_pos[B,N,N]
 *                        (signed bucket of the difference of the first mention starts, Config.py:206-215)
 *   n_valid[B]           entities per document;  first_start[B,N] start of each entity's first mention */
int gcgcn_tensorise(int B, int N, int S, int T, int R, int dis_plus, int n_slots, const int32_t* slots, int n_edges,
                    const int32_t* edges, int n_labels, const int32_t* labels, int n_mentions, const int32_t* mentions,
                    const int32_t* mention_node, const int32_t* n_valid, const int32_t* first_start, float* adj, uint8_t* sen, uint8_t* pos_h,
                    uint8_t* pos_t, float* node_pos, int64_t* node_relative_pos, float* label_matrix, void* stream);

/* ---- test hook: how tile passengers are placed among the rows of a carrying launch (csrc/common.hpp Spread) ---------
 * kind[x] / ordinal[x] for every workgroup index x < n_tiles + n_others: 1 / tile number or 0 / row number.  Host-only. */
int gcgcn_debug_spread(int64_t n_tiles, int64_t n_others, int64_t cohort, int64_t pct, int32_t* kind, int32_t* ordinal);

/* ---- raw batched GEMM (exposed for unit tests and benchmarks of the MFMA kernel) ----------- */
/* C[z] = alpha * opA(A[z]) opB(B[z]);  a_kc: A stored [M][K] else [K][M];  b_kc: B stored [N][K]
 * else [K][N];  z < batch with element strides sA, sB, sC;  tile: 0 or 1 = 64x64 block tiles (the only body; other values are refused);
 * splits: 0 auto, 1 none, n = split K n ways through ws[ws_elems] (>= n*batch*M*N floats);
 * bias[N] optional, relu/accumulate flags. */
int gcgcn_gemm(int M, int N, int K, const float* A, int64_t lda, int a_kc, const float* B, int64_t ldb, int b_kc,
               float* C, int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, const float* bias,
               int relu, int accumulate, int tile, int splits, float* ws, int64_t ws_elems, void* stream);

/* The same product with ONE dimension read from device memory (exposed for unit tests; used by the edge-feature producer for
 * its compacted rows): dyn = 1: M = *count, dyn = 2: K = *count (the M / K argument is ignored); cap = a host-side upper bound of
 * *count.  Interior shapes run the unguarded tile body on the count rounded up to 64: operand buffers must be readable (C
 * writable, dyn = 1) that far, and for dyn = 2 at least one operand's rows in [*count, roundup64) must be zero. */
int gcgcn_gemm_dyn(int M, int N, int K, const float* A, int64_t lda, int a_kc, const float* B, int64_t ldb, int b_kc, float* C,
                   int64_t ldc, const float* bias, int accumulate, const int32_t* count, int dyn, int64_t cap, float* ws,
                   int64_t ws_elems, void* stream);

/* ---- the trainer's optimiser step (config/Config.py:300, 372-373: torch.optim.Adam, no weight decay) -------------------
 * ONE launch over n_tensors parameter tensors.  table (device memory): n_tensors records of 56 bytes
 *   { float* p; const float* g; float* m; float* v; int64 numel; int64 block_begin; float step_size; float inv_bc2_sqrt; }
 * sorted by block_begin, where a tensor owns ceil(numel / 1024) consecutive workgroups starting at block_begin,
 * step_size = lr / (1 - beta1^t) and inv_bc2_sqrt = 1 / sqrt(1 - beta2^t) with t the tensor's own step count (torch skips a
 * parameter whose .grad is None and does not advance its t).  Arithmetic as torch.optim.Adam's single-tensor path. */
int gcgcn_adam_step(int n_tensors, const void* table, int64_t total_blocks, double beta1, double beta2, double eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCGCN_H */
