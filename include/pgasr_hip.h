/*
 * pgasr_hip.h -- C ABI of libpgasr_hip.so, the MI355X (gfx950) implementation of the
 * acoustic-model policy-gradient training hot path of ana-kuznetsova/Policy-Gradient-ASR.
 *
 * The reference is pure Python on torch.nn (no FFI of its own), so each
 * entry point below names the reference call site whose arithmetic it replaces; the Python
 * host layer (the policy_gradient_asr_amd package) binds these with ctypes and keeps the reference's
 * module-level names and signatures on top.
 *
 * Conventions (all entry points):
 *   - return value: PGASR_OK (0) or a pgasr_status code; no exceptions cross the ABI;
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no sync, no
 *     hipMalloc): workspaces are sized by the *_workspace_bytes queries and caller-allocated;
 *   - no global mutable state: re-entrant per stream;
 *   - tensors are dense, row-major, fp32 unless said otherwise; activations are TIME-MAJOR
 *     (T,B,C) so that one recurrent / CTC step touches one contiguous (B,C) slab;
 *   - blank = pad = index 0 is the reference's convention (CTCdecoder.py:41, data.py:99) and
 *     is passed explicitly as `blank`.
 */
#ifndef PGASR_HIP_H
#define PGASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pgasr_status {
    PGASR_OK = 0,
    PGASR_ERR_INVALID_ARG = 1,   /* null pointer, negative size, shape the kernel cannot take */
    PGASR_ERR_LAUNCH = 2,        /* hipLaunch / runtime error (hipGetLastError != success) */
    PGASR_ERR_WORKSPACE = 3,     /* workspace null or too small */
    PGASR_ERR_UNSUPPORTED = 4,   /* size beyond a compiled-in limit (see each function) */
    PGASR_ERR_TIMEOUT = 5        /* a bounded in-kernel wait gave up (persistent LSTM) */
} pgasr_status;

/* 4 (round 3): pgasr_adam_step(guards, applied), pgasr_lstm_pack_weights(planes), the feed phases, and the streamed order:
 * pgasr_lstm_wgrad_slabs, pgasr_lstm_layer_bwd_streamed, pgasr_lstm_wgrads_streamed(+_workspace_bytes), pgasr_stream_gate_sum.
 * 7 (round 5): pgasr_stream_gate_report, pgasr_lstm_cell_f32, pgasr_gemm_x6w_feed_phase_f32 / _head_items; the sampler's counters for
 * utterances beyond the global batch. */
#define PGASR_ABI_VERSION 7

int pgasr_abi_version(void);
const char* pgasr_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * A5  CTC loss + gradient.  The reference has no call site; the oracle is
 * torch.nn.functional.ctc_loss composed with log_softmax.  Conventions from CTCdecoder.py:41
 * (blank=0) and data.py:99 (targets padded with 0).
 *
 *   log_probs   (T,B,V) fp32 log-softmax outputs
 *   targets     (B,Lmax) int32, first target_lengths[b] entries valid
 *   nll         (B) fp32: -log p(target_b | x_b); +inf when no alignment exists
 *   grad_logits (T,B,V) fp32: d(sum_b utt_scale[b]*nll_b)/d(logits) = utt_scale[b] *
 *               (softmax - posterior occupancy); zero for t >= input_lengths[b] and for
 *               utterances whose nll is +inf.  utt_scale may be NULL (= 1).
 *   If pg_coef and pg_path are non-NULL the REINFORCE term of pgasr_reinforce_grad is
 *   added in the same pass (A12):  + pg_coef[b] * (softmax - onehot(pg_path[t,b])).
 *   Limits: 2*Lmax+1 <= 2048, V <= 64.
 * ---------------------------------------------------------------------------------------- */
size_t pgasr_ctc_workspace_bytes(int T, int B, int V, int Lmax);
int pgasr_ctc_loss_grad(const float* log_probs, const int32_t* targets,
                        const int32_t* input_lengths, const int32_t* target_lengths,
                        int T, int B, int V, int Lmax, int blank,
                        const float* utt_scale, const float* pg_coef, const int32_t* pg_path,
                        float* nll, float* grad_logits,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The gradient pass of pgasr_ctc_loss_grad on its own, over the lattice that an earlier
 * pgasr_ctc_loss_grad(..., grad_logits = NULL, ...) call with the same T, B, V, Lmax left in `workspace`
 * (so the lattice can run on another stream beside the kernels that produce pg_coef).
 * pg_coef_per_frame != 0: pg_coef is (T,B), one coefficient per frame (pgasr_pg_step_coefs), instead of (B,). */
int pgasr_ctc_grad_from_lattice(const float* log_probs, const int32_t* input_lengths,
                                const int32_t* target_lengths, int T, int B, int V, int Lmax, int blank,
                                const float* utt_scale, const float* pg_coef, const int32_t* pg_path,
                                int pg_coef_per_frame, float* grad_logits, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Rewards and gradient coefficients (policy_grad.py:4-16 intent):
 *   dist [2B]: edit distances of the greedy paths, then of the sampled paths (pgasr_edit_distance);
 *   R = -dist / max(L,1);  pg_coef = lam * inv_global_batch * (R_sample - R_greedy);
 *   utt_scale = inv_global_batch / max(L,1).
 * pgasr_pg_loss_value: terms[b] = nll[b]*utt_scale[b] - pg_coef[b] * sum_{t<input_lengths[b]} log_probs[t,b,path[t,b]]
 *   (pg_coef and path both NULL: CTC only); the objective's value is sum_b terms[b].  Deterministic.
 *   pg_coef_per_frame != 0: pg_coef is (T,B) and the second term is sum_t pg_coef[t,b] * log_probs[t,b,path[t,b]].
 * pgasr_pg_step_coefs: per-frame coefficients from the PER-STEP rewards of policy_grad.py:10-15 (the reference computes r_t and
 *   never consumes it; this is the build's use of it, opt-in).  With yhat the collapsed path, d(i) = ED(y, yhat[:i]) and
 *   rho_j = d(j-1) - d(j) the reward of character j (reference: r_1 = rho_1 + rho_2, r_t = rho_{t+1} for t >= 2), the reward-to-go
 *   of frame t is G(t) = d(c(t)) - d(|yhat|), c(t) = characters that start in frames < t, and
 *     coef[t,b] = lam * inv_global_batch * (G_sample(t) - G_greedy(t)) / max(L_b,1)   for t < input_lengths[b], else 0
 *   (baseline = the greedy path's reward-to-go at the same frame; coef[0,b] is pgasr_pg_rewards' pg_coef[b]).
 *   paths (2,T,B) = greedy then sampled frame labels; prefix_dist (2B, prefix_stride) and token_lengths (2B) = the per-prefix
 *   distances and lengths of their collapsed forms (pgasr_edit_distance with prefix_dist, pgasr_ctc_collapse). */
int pgasr_pg_rewards(const int32_t* dist, const int32_t* target_lengths, int B, float lam,
                     float inv_global_batch, float* R_greedy, float* R_sample, float* pg_coef,
                     float* utt_scale, void* stream);
int pgasr_pg_loss_value(const float* log_probs, const int32_t* path, const int32_t* input_lengths,
                        const float* nll, const float* utt_scale, const float* pg_coef,
                        int T, int B, int V, int pg_coef_per_frame, float* terms, void* stream);
int pgasr_pg_step_coefs(const int32_t* paths, const int32_t* input_lengths, const int32_t* prefix_dist, int prefix_stride,
                        const int32_t* token_lengths, const int32_t* target_lengths, int T, int B, int blank,
                        float lam, float inv_global_batch, float* coef, void* stream);

/* ------------------------------------------------------------------------------------------
 * A4  the CTC head as one kernel (model.py:52-55 as the build's Seq2Seq uses it): logits = x W^T + bias, log_probs = log_softmax.
 *   x (rows, K) fp32 with row stride ldx, W (V, K) row-major, bias (V) or NULL; logits / log_probs (rows, V), either may be NULL.
 *   Exact fp32 arithmetic (v_mfma_f32_32x32x2_f32) in every precision mode; one pass over x.  V <= 32, K % 64 == 0, K <= 1024.
 * ---------------------------------------------------------------------------------------- */
int pgasr_head_logsoftmax(const float* x, long long rows, int K, int ldx, const float* W, const float* bias, int V,
                          float* logits, float* log_probs, void* stream);

/* ------------------------------------------------------------------------------------------
 * A9 / A12  per-frame best label and sampled label.
 *   scores (T,B,V) fp32 logits or log-probs (softmax is shift invariant).
 *   greedy_path[t,b] = argmax_v scores[t,b,v], first max wins  (oracle: numpy argmax)
 *   sample_path[t,b] ~ softmax(scores[t,b,:]) by inverse CDF with
 *        u = (philox4x32_10(ctr=(t*ctr_stride+ctr_base+b, offset,0,0), key=(seed_lo,seed_hi)).x >> 8) * 2^-24
 *   ctr_stride = 0 means B (counter t*B+b, the single-process layout); a data-parallel rank passes the GLOBAL batch as
 *   ctr_stride and its first utterance's global index as ctr_base (0 <= ctr_base < ctr_stride), so that N ranks with one
 *   seed draw what one process holding the whole batch draws.  Utterances with ctr_base + b >= ctr_stride lie BEYOND the global batch
 *   (ABI 7: the empty utterances a ragged batch is padded with) and draw from a disjoint domain, ctr = (t*B+b, offset, 1, 0).
 *   Either output may be NULL.  V <= 64.
 * ---------------------------------------------------------------------------------------- */
int pgasr_frame_argmax_sample(const float* scores, int T, int B, int V,
                              uint64_t seed, uint32_t offset, int ctr_stride, int ctr_base,
                              int32_t* greedy_path, int32_t* sample_path, void* stream);

/* ------------------------------------------------------------------------------------------
 * A9  CTC collapse of frame paths: drop repeats, then drop blank, per utterance, for frames
 * t < lengths[b].  paths (P,T,B) int32 (P stacked path sets, e.g. greedy and sampled);
 * tokens (P,B,T) int32, token_lengths (P,B) int32.  Bit-exact against
 * oracle.decode_ref.collapse_path.
 * ---------------------------------------------------------------------------------------- */
/* A0  the collated batch of data.py:107-116 made ready for the kernels in one launch: fmask (B,T) fp32 1/0 -> in_len (B)
 * (model.py:52: lengths = mask.sum(1)); tmask (B,L) int64 1/0 -> tg_len (B) (data.py:101); targets (B,L) int64 -> int32. */
int pgasr_batch_prep(const float* fmask, int B, int T, const long long* tmask, const long long* targets, int L,
                     int32_t* in_len, int32_t* tg_len, int32_t* targets32, void* stream);
int pgasr_ctc_collapse(const int32_t* paths, const int32_t* lengths, int P, int T, int B,
                       int blank, int32_t* tokens, int32_t* token_lengths, void* stream);

/* ------------------------------------------------------------------------------------------
 * A10  batched Levenshtein distance (metrics.py:4-21: sub = ins = del = 1).
 *   ref (N,ref_stride) int32 with ref_len (N); hyp (N,hyp_stride) int32 with hyp_len (N);
 *   dist[n] = ED(ref_n, hyp_n).  If prefix_dist != NULL it is (N,hyp_stride+1) and receives
 *   ED(ref_n, hyp_n[:i]) for i = 0..hyp_len[n] -- the quantity policy_grad.py:11-15
 *   differences to form r_t.  Limit: sequence lengths <= 4095.
 * ---------------------------------------------------------------------------------------- */
int pgasr_edit_distance(const int32_t* ref, const int32_t* ref_len, int ref_stride,
                        const int32_t* hyp, const int32_t* hyp_len, int hyp_stride,
                        int N, int32_t* dist, int32_t* prefix_dist, void* stream);

/* ------------------------------------------------------------------------------------------
 * A12  REINFORCE gradient on the logits:
 *   grad[t,b,v] (+)= coef[b] * (softmax(scores[t,b,:])[v] - (v == path[t,b])),  t < lengths[b]
 *   and 0 (or unchanged when accumulate != 0) for t >= lengths[b].
 * ---------------------------------------------------------------------------------------- */
int pgasr_reinforce_grad(const float* scores, const int32_t* path, const float* coef,
                         const int32_t* lengths, int T, int B, int V, int accumulate,
                         float* grad, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense contractions (A1-A4 and their gradients).  fp32 in, fp32 accumulate on the matrix cores
 * (v_mfma_f32_32x32x2_f32, bit-equal to an fmaf chain).
 *
 *   C[b] (M x N, ldc) = epilogue( alpha * opA(A[b]) * opB(B[b]) )     b = 0..batch-1
 *   opA(A) = A (M x K row-major, lda) or, transA != 0, A stored K x M;  opB likewise (transB != 0:
 *   B stored N x K, i.e. torch's Linear weight layout).
 *   sum_batches != 0: the batch results are summed into the single C (partial slabs in workspace,
 *   reduced in index order); splitk > 1 splits K the same way.  workspace bytes:
 *   pgasr_gemm_workspace_bytes.  epilogue: + bias[n] + bias2[n]; act 1 = leaky_relu(slope)
 *   (model.py:50); dact_y != NULL multiplies by d leaky_relu evaluated on dact_y (same layout as C);
 *   accumulate != 0 adds to C.  norm_operand 1/2 applies (x - shift[b]) * scale[b] to A/B while
 *   the tile is loaded: the per-utterance InstanceNorm2d of model.py:37,48 fused into the affine.
 *   precision 0: exact fp32 MFMA.  precision 1: each operand element is split into bf16 hi+lo while
 *   staged and the product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation
 *   (~1e-6 relative; ~5x the fp32 matrix rate).  precision 2 (ABI 5): the fp32-faithful six-product arithmetic of
 *   pgasr_gemm_x6w_f32 (three bf16 planes per operand) -- only for the weight-gradient shaped products the 256 x 256 TN
 *   kernel takes (transA, !transB, split-K or batch sums, M, N % 256 == 0, K and the K-slabs % 32 == 0), else
 *   PGASR_ERR_UNSUPPORTED.
 *   xcc_busy (NULL = plain launch; precision 1 only): device array of 8 words, e.g. the busy counters a
 *   persistent LSTM sweep keeps in its workspace (pgasr_lstm_busy_offset).  Tiles are then drawn from a
 *   global queue and a workgroup that RUNS on XCD i (HW_REG_XCC_ID) with xcc_busy[i] != 0 takes none; an
 *   unmasked second launch picks up any leftovers.  Keeps a GEMM off the XCDs of a concurrent sweep; the
 *   result is complete and identical for any placement.
 *   workspace: always >= pgasr_gemm_workspace_bytes (>= 256: tile counter + partial slabs).
 * ---------------------------------------------------------------------------------------- */
size_t pgasr_gemm_workspace_bytes(int M, int N, int batch, int splitk, int sum_batches);
int pgasr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                   const float* A, int lda, long long strideA,
                   const float* B, int ldb, long long strideB,
                   float* C, int ldc, long long strideC,
                   int batch, int sum_batches, int splitk,
                   const float* bias, const float* bias2, int act, float slope, int accumulate,
                   const float* dact_y, int norm_operand, const float* shift, const float* scale,
                   int precision, const unsigned* xcc_busy, void* workspace, size_t workspace_bytes, void* stream);

/* column sums of X (rows x cols, leading dim ld) -> out (and out2 if non-NULL): bias gradients. */
/* Activation x weight GEMM with the weight PRE-SPLIT into bf16 hi/lo planes (csrc/gemm_dma.hip):
 *   C[M,N] = A[M,K] * W[N,K]^T (+ bias[n]) (* (dact_y[m,n] > 0 ? 1 : slope))
 * Replaces the same torch calls as pgasr_gemm_f32 for the two big shapes of the path: the LSTM input
 * projections (model.py:39-44) and their input gradients.  Same 3-term bf16 split / fp32 accumulate as
 * pgasr_gemm_f32(precision=1); tiles are moved by LDS-DMA three stages deep.
 * pgasr_split_bf16_planes: src (rows x cols fp32, leading dim ld) -> dense planes hi, lo of
 *   (rows x cols) bf16, or (cols x rows) when transpose != 0; x = hi + lo to ~2^-17 relative.
 * pgasr_gemm_x3w_f32 needs K % 32 == 0, N % 128 == 0, lda % 4 == 0 and 16-byte aligned A / planes, else
 *   PGASR_ERR_UNSUPPORTED (use pgasr_gemm_f32).  N % 256 == 0 takes the 256 x 256 tile (128 x 128 per wave, 4 waves).  dact_y (optional) has C's shape and leading dim. */
int pgasr_split_bf16_planes(const float* src, int rows, int cols, int ld, int transpose,
                            unsigned short* hi, unsigned short* lo, void* stream);
int pgasr_gemm_x3w_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                       const unsigned short* Wlo, float* C, int ldc, const float* bias,
                       const float* dact_y, float slope, void* stream);
/* The same product FEEDING a forward LSTM sweep that is already running (csrc/gemm_dma.hip, FEED kernel): the input
 * projection leaves the step's critical path.  Persistent workgroups on the XCDs the sweep leaves free draw
 * 256 x 256 tiles (256 x 128 when N % 512 != 0) in the order the sweep consumes rows (rows are (t, b) time-major; N = the two directions' column
 * halves: row tile i of direction 0 together with row tile last-i of direction 1), store C write-through and count
 * finished tiles in tiles_done[2][ceil(M/256)] (zeroed by the caller BEFORE the sweep is launched; a word is
 * complete at pgasr_gemm_x3w_feed_col_tiles(N) = the column tiles of one direction half: the consumer's fed_need).  xcc_busy = the sweep's busy counters (pgasr_lstm_busy_offset): workgroups on a busy XCD
 * take no tile, a second unmasked launch picks up any rest.  workspace >= 1024 bytes; with
 * pgasr_gemm_x3w_feed_workspace_bytes() (32 MB) the first 16 tile groups -- the ones the sweep is waiting for -- are
 * computed as four parallel K-quarters and summed in index order by the last arrival (a tile otherwise takes one CU
 * K/32 x 1.7 us).  A tile's result is ONE fp32 accumulation chain over K, except those first tiles of a feed with
 * K >= 1024 and K % 128 == 0, which are ((q0 + q1) + q2) + q3 over K-quarters accumulated from zero.  To get the same
 * bits from the sequential order of the same product, call THIS function before the (plain) sweep with xcc_busy = NULL:
 * the decomposition depends on the arguments only, never on which workgroup computes what.  (On the 256 x 128 tile
 * -- N % 512 != 0 -- every x3w product with such a K, pgasr_gemm_x3w_f32's too, is the quarter sum.)
 * Call order on the host:
 * zero tiles_done -> pgasr_lstm_layer_fwd_fed (stream S) -> pgasr_stream_gate + this call (another stream that
 * waits for the zeroing).  order 0: rows in the order a forward sweep consumes them; 1: a backward sweep's (the
 * product is then the input gradient of the layer above, feeding pgasr_lstm_layer_bwd_fed).  Needs N % 256 == 0 on top of pgasr_gemm_x3w_f32's conditions and M*ldc*4 < 2^31. */
/* phase 0: the whole feed in one call (queue zeroed here, two persistent launches).  phase 1 (HEAD) + phase 2 (REST) split it:
 * phase 1 zeroes the queue and computes the first head_groups tile groups -- pgasr_gemm_x3w_feed_head_items(N, K, head_groups)
 * work items, one per workgroup, no XCD mask -- and is meant for the SWEEP'S OWN stream, in front of the sweep launch, so
 * that the rows of the sweep's first steps are in memory when it starts (they are what a fed sweep otherwise waits ~50-70 us
 * for: stream gate + launch + first tiles); phase 2, on the feeding stream and ordered behind phase 1, continues the SAME
 * queue (same workspace, not zeroed again) with the persistent launches.  The decomposition -- and so the bits -- is the
 * same as phase 0's.  head_items == 0: no head for this shape (use phase 0). */
size_t pgasr_gemm_x3w_feed_workspace_bytes(void);
int pgasr_gemm_x3w_feed_col_tiles(int N);
int pgasr_gemm_x3w_feed_head_items(int N, int K, int groups);
int pgasr_gemm_x3w_feed_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                            const unsigned short* Wlo, float* C, int ldc, const float* bias,
                            const unsigned* xcc_busy, unsigned* tiles_done, int order, int phase, int head_groups,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The two weight-gradient products of one BLSTM layer (model.py:39-44; torch autograd's dW_ih, dW_hh) in ONE queue-mode
 * launch of the 256 x 256 TN kernel:
 *   dwih_perm (2*4H x in_dim) = dgates^T x                   (K = T*B rows of the (t, b)-major tensors)
 *   dwhh_perm (2 x 4H x H)    = dgates[d]^T h_prev(d)        (K = (T-1)*B rows; prev = t-1 for d = 0, t+1 for d = 1; h = out)
 * bf16x3 arithmetic (as pgasr_gemm_f32 precision 1); every 256 x 256 tile is the sum of its partial products over the time
 * slabs of pgasr_lstm_wgrad_slabs(T) in the order a backward sweep completes them (direction 0: [h_j, h_j+1) from the top
 * down; direction 1: the mirror image) -- a function of the shapes only, so every mode gives the same bits.
 * slab_done != NULL: dgates belongs to a backward sweep that is STILL RUNNING (pgasr_lstm_layer_bwd_streamed on another
 * stream, launched before this call); the work items then wait for the slab_done words that cover their rows and read
 * dgates with agent-scope loads.  slab_done == NULL: dgates is complete (the sequential order, same bits).
 * xcc_busy (optional): the sweep's busy counters (workgroups on its XCDs take no item; a second unmasked launch picks up
 * any rest).  err_word (optional): set to 1 when a wait gives up after 3 s -- pass the sweep workspace's error word.
 * Needs in_dim % 256 == 0, B % 32 == 0, T >= 2 and 16-byte aligned tensors, else PGASR_ERR_UNSUPPORTED (use pgasr_gemm_f32
 * behind the sweep). */
/* planes (ABI 5): 2 = the bf16x3 arithmetic above; 3 = the fp32-faithful six-product arithmetic of pgasr_gemm_x6w_f32 (csrc/gemm_x6.hip:
 * 16-deep steps, three planes per operand), same slabs, same order, same gate. */
size_t pgasr_lstm_wgrads_workspace_bytes(int T, int in_dim);
int pgasr_lstm_wgrads_streamed(const float* dgates, const float* x, const float* out, int T, int B, int in_dim,
                               float* dwih_perm, float* dwhh_perm, const unsigned* xcc_busy,
                               const unsigned* slab_done, int* err_word, int planes,
                               void* workspace, size_t workspace_bytes, void* stream);

/* The same two roles of W_ih (input projection, input gradient: model.py:39-44) in the REFERENCE'S arithmetic -- torch fp32 -- at
 * bf16 MFMA rate (csrc/gemm_x6.hip; the host layer's "f32" precision mode): every fp32 operand as THREE bf16 planes
 * (x = hi + mid + lo to 2^-24) and every product as the SIX terms hh + hm + mh + hl + lh + mm (all terms >= 2^-24 of the product),
 * fp32 accumulate.  256 x 256 tile, 16-deep steps.
 * pgasr_split_bf16_planes3: like pgasr_split_bf16_planes with the third plane.
 * pgasr_gemm_x6w_f32: C[M,N] = A[M,K] * W[N,K]^T (+ bias[n]) (* (dact_y[m,n] > 0 ? 1 : slope)); needs K % 16 == 0, K >= 64,
 *   N % 256 == 0, lda % 4 == 0, 16-byte aligned A / planes, M*ldc*4 < 2^32 and M*lda*4 < 2^32, else PGASR_ERR_UNSUPPORTED
 *   (use pgasr_gemm_f32 precision 0).  One fp32 accumulation chain over K per tile.
 * pgasr_gemm_x6w_feed_f32: the product FEEDING a sweep that is already running -- contract, call order, tiles_done / xcc_busy /
 *   order / workspace exactly as pgasr_gemm_x3w_feed_f32 (phase 0; there is no head launch), N % 512 == 0, M*ldc*4 < 2^31;
 *   fed_need of the consumer = pgasr_gemm_x6w_feed_col_tiles(N) = N / 512.  With pgasr_gemm_x6w_feed_workspace_bytes() the first
 *   tiles of a K >= 1024 (K % 64 == 0) feed are fixed-order sums ((q0 + q1) + q2) + q3 of K-quarters computed in parallel; called
 *   with xcc_busy = NULL before a plain sweep it is the sequential order of the same product with the same bits. */
/* pgasr_pack_x6w_planes: the same three planes PACKED tile by tile in the kernel's own LDS-image order,
 *   pack[N/256][K/16][plane 3][8 KB]  (N*K*6 bytes; N = rows, K = cols of the weight as the GEMM sees it; transpose as above),
 * so that a W piece of a step is 1 KB of consecutive bytes (8 full cache lines per LDS-DMA instruction instead of 32 quarter lines of a
 * row-major plane).  Needs N % 256 == 0, K % 16 == 0.  pgasr_gemm_x6w_f32 / _feed_f32 take a pack as Whi with Wmid = Wlo = NULL. */
int pgasr_pack_x6w_planes(const float* src, int rows, int cols, int ld, int transpose, void* pack, void* stream);
int pgasr_split_bf16_planes3(const float* src, int rows, int cols, int ld, int transpose,
                             unsigned short* hi, unsigned short* mid, unsigned short* lo, void* stream);
int pgasr_gemm_x6w_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                       const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc, const float* bias,
                       const float* dact_y, float slope, void* stream);
size_t pgasr_gemm_x6w_feed_workspace_bytes(void);
int pgasr_gemm_x6w_feed_col_tiles(int N);
int pgasr_gemm_x6w_feed_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                            const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc, const float* bias,
                            const unsigned* xcc_busy, unsigned* tiles_done, int order,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same feed in two launches (round 5).  phase 0: as pgasr_gemm_x6w_feed_f32.  phase 1, the HEAD: the queue head is zeroed and the
 * pgasr_gemm_x6w_feed_head_items(M, N, K) K-split items at the front of the queue (the tile groups the consumer takes first) are computed
 * one per workgroup, without an XCD mask -- to be issued on the feeding stream right behind the PREVIOUS sweep, before the consuming sweep
 * has registered its XCDs (its launch, registration, the gate and the memset are then off the first rows' path).  phase 2, the rest: no
 * memset, the persistent passes continue the same queue (same workspace, same arguments).  A phased feed needs the full
 * pgasr_gemm_x6w_feed_workspace_bytes(); head_items == 0 (no split head for this shape): PGASR_ERR_UNSUPPORTED for phases 1 and 2.
 * ctrl (optional): 256 words zeroed by the CALLER for this feed -- the queue and arrival counters live there instead of the workspace's
 * first KB and no phase issues a memset, so phases 1 and 2 may be issued on two streams (the head right behind the previous sweep, the rest
 * behind the consuming sweep's registration) and draw from the one queue side by side.
 * The decomposition -- hence every bit of C -- is that of phase 0. */
int pgasr_gemm_x6w_feed_head_items(int M, int N, int K);
int pgasr_gemm_x6w_feed_phase_f32(int M, int N, int K, const float* A, int lda, const unsigned short* Whi,
                                  const unsigned short* Wmid, const unsigned short* Wlo, float* C, int ldc, const float* bias,
                                  const unsigned* xcc_busy, unsigned* tiles_done, int order, int phase, unsigned* ctrl,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* The reference's attention context (model.py:58-94, Attention.forward, AS EXECUTED -- csrc/attention.hip):
 *   ctx[q,k] = sum_i e[b,i,k] * (sum_r exp(d[q,r] e[b,i,k])) / (sum_k' exp(d[q,k] e[b,i,k'])),   b = q % B
 * dec (NQ x H) decoder states -- NQ = L * B rows in (step, utterance) order, or just B rows for one step --, enc (B x T x H) encoder
 * outputs, ctx (NQ x H).  The denominator is the reference's own: model.py:73 divides the (H,H) outer-product matrix by its row
 * sums lined up with the LAST axis (entry [r,k] by row k's sum).  Every encoder frame counts (no mask), H_dec == H_enc.  fp32,
 * exponent maxima taken out exactly.  A defined function for parity, not a hot kernel (2 H exponentials per output and frame). */
int pgasr_attention_ctx(const float* dec, const float* enc, int NQ, int B, int T, int H, float* ctx, void* stream);

/* One step of the decoder's one-layer LSTM (model.py:104,111, Decoder's nn.LSTM(128 -> hidden); gate order i, f, g, o):
 *   g = gh + xp;  c <- sigmoid(g_f) c + sigmoid(g_i) tanh(g_g);  h <- sigmoid(g_o) tanh(c);  h_out <- h (may be NULL)
 * gh (B x 4H) = h_{t-1} W_hh^T (pgasr_gemm_f32, exact mode), xp (B x 4H) = x_t W_ih^T + b_ih + b_hh, c and h (B x H) in place.
 * Forward only, like the reference's decoder. */
int pgasr_lstm_cell_f32(const float* gh, const float* xp, float* c, float* h, float* h_out, int B, int H, void* stream);

size_t pgasr_colsum_workspace_bytes(int rows, int cols);
int pgasr_colsum_f32(const float* X, int rows, int cols, int ld, float* out, float* out2, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);

/* A1: per-utterance mean and 1/sqrt(var+eps) over all F*T values of x (B,F,T), biased variance,
 * padding included (model.py:37,48; nn.InstanceNorm2d on (B,1,F,T)). */
int pgasr_instnorm_stats(const float* x, int B, int F, int T, float eps, float* mean, float* rstd,
                         void* stream);

/* A4: row-wise log_softmax of logits (rows, V), V <= 64 (the commented head of model.py:166-167;
 * consumer contract model.py:323). */
int pgasr_log_softmax_rows(const float* logits, long long rows, int V, float* log_probs, void* stream);

/* ------------------------------------------------------------------------------------------
 * A3  bidirectional LSTM layer, H = 256 per direction (model.py:39-44), packed-sequence semantics
 * of model.py:52-55: per-utterance lengths, the reverse direction starts at each utterance's own
 * last frame, outputs are zero past the length.
 *
 * pgasr_lstm_pack_weights: torch-layout parameters of the two directions (weight_ih (4H,in),
 *   weight_hh (4H,H), bias_ih, bias_hh (4H); gate rows i|f|g|o) ->
 *     wih_perm (2*4H, in) and bias_perm (2*4H) in column order dir*4H + unit*4 + gate
 *     (bias_perm = bias_ih + bias_hh), and the two register-resident bf16 W_hh packs
 *     (pgasr_lstm_pack_bytes(which, planes) each) the sweeps load once.  planes = 2: every fp32 weight as bf16 hi + lo
 *     (products hi*hi + hi*lo + lo*hi: ~16 operand bits, the default); planes = 3: hi + mid + lo for the fp32-faithful
 *     sweeps (flags bit 1 below; the reference's nn.LSTM is fp32, model.py:39-44): six products, every term down to
 *     2^-24 of the product.
 * pgasr_lstm_layer_fwd: gates (T,B,2,H,4) holds xproj = X * wih_perm^T + bias_perm on entry and
 *   the activations (i,f,g,o) on exit; out (T,B,2H) = h; cbuf (T,B,2,H) = c.
 * pgasr_lstm_layer_bwd: dout (T,B,2H) -> gates overwritten in place by d(pre-activation gates);
 *   the caller forms dX = dgates * wih_perm, dW_ih = dgates^T X, dW_hh = dgates^T h_prev with
 *   pgasr_gemm_f32 and maps them back with pgasr_lstm_unpack_grads.  dbias_part (optional, 16-byte aligned,
 *   ceil(B/16) x 2*4H floats): per 16-utterance group, the sum over t of dgates in gates' column order -- the bias
 *   gradient is the sum of its rows (saves a 262 MB column-sum pass over dgates at B=32,T=1000).
 * The sweeps are persistent kernels: 16 compute workgroups (+ 4 helper workgroups that stage the coming steps'
 * HBM rows in an L2-resident ring inside the workspace) per (direction, 16-utterance group) that hand
 * h_t / partial dh sums to each other through global memory every step (self-validating words; plain
 * stores when the cluster is verified to share an XCD, write-through stores otherwise).
 * flags bit 0: force the write-through protocol (testing the placement-independent path); bit 1: three-plane
 * (fp32-faithful) arithmetic -- the W_hh pack must have been made with planes = 3; bit 2: no helper
 * workgroups (the loaders read HBM themselves: slower, but two processes can then share one GPU's XCDs).
 * All workgroups of a sweep wait for each other and must be co-resident: with helpers a sweep takes 20 of an XCD's 32
 * CUs per cluster, so run ONE sweep at a time per GPU (other kernels beside it are fine: they finish on their own) or
 * set bit 2.  Every wait is bounded (3 s): a sweep that cannot make progress sets the error word and returns.
 * Limit: B <= 128.  workspace (pgasr_lstm_workspace_bytes) holds exchange buffers and an
 * error word (offset: pgasr_lstm_error_offset) that is set when a bounded wait times out.
 * ---------------------------------------------------------------------------------------- */
size_t pgasr_lstm_pack_bytes(int which, int planes);
int pgasr_lstm_pack_weights(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                            const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                            int in_dim, float* wih_perm, float* bias_perm,
                            void* whh_pack_fwd, void* whh_pack_bwd, int planes, void* stream);
int pgasr_lstm_unpack_grads(const float* dwih_perm, const float* dbias_perm, const float* dwhh_perm, int in_dim,
                            float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                            float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                            int accumulate, void* stream);
size_t pgasr_lstm_workspace_bytes(int T, int B, int backward);
/* Byte offset of the workspace's error word: set to 1 when a bounded wait inside a sweep gives up (results invalid).
 * The word is STICKY -- no launch clears it: zero the first 16 bytes of a workspace once after allocating it. */
int pgasr_lstm_error_offset(int B, int backward, size_t* offset);
/* How a time-out reaches the caller: the launch calls themselves are asynchronous and return PGASR_OK; the sweep sets
 * the sticky word.  pgasr_lstm_status synchronises `stream`, reads the word and returns PGASR_ERR_TIMEOUT when it is
 * set (PGASR_OK otherwise); pgasr_adam_step takes the words' device addresses as guards and skips the update while
 * one is set, so invalid gradients never reach the parameters between two host-side checks. */
int pgasr_lstm_status(const void* workspace, size_t workspace_bytes, int B, int backward, void* stream);
/* out[0] = 1.0f if *word0 or *word1 (sweep error words; either may be NULL) is non-zero, else 0.0f -- the flag a data-parallel rank writes
 * into word 0 of its gradient buffer before the last all-reduce, so that every replica skips the Adam update together (one launch). */
int pgasr_error_flag(const int32_t* word0, const int32_t* word1, float* out, void* stream);
int pgasr_lstm_busy_offset(int B, int backward, size_t* offset);   /* 8 per-XCD busy counters (hint for pgasr_gemm_f32) */
/* Holds `stream` until any of words[0..count) is non-zero or timeout_us (<= 100000) has passed: put in front of
 * GEMMs that are to run BESIDE a sweep, so that the sweep's workgroups are dispatched first (a large grid
 * enqueued ahead of the sweep delays it by the whole GEMM).  A hint only: never affects results. */
int pgasr_stream_gate(const unsigned* words, int count, int timeout_us, void* stream);
/* The same for consumers that WAIT for the sweep (pgasr_lstm_wgrads_streamed): holds `stream` until the counters add up to
 * `need` = the sweep's clusters, 2 * ceil(B/16) -- all of its workgroups are then resident; workgroups that poll for the
 * sweep's publications must not take CUs the sweep still needs.  running (optional): the sweep's slab_done words -- a
 * non-zero first word also opens the gate (the sweep has published, i.e. it is under way or already OVER: a gate that
 * comes late would otherwise sit out its whole time-out on counters that have gone back to zero). */
int pgasr_stream_gate_sum(const unsigned* words, int count, int need, const unsigned* running, int timeout_us, void* stream);
/* pgasr_stream_gate_sum that also says how it left (ABI 7): report[0] = 1 opened on the busy counters, 2 opened on a publication
 * (`running`), 3 timed out; report[1] = microseconds the gate held the stream.  report may be NULL. */
int pgasr_stream_gate_report(const unsigned* words, int count, int need, const unsigned* running, int timeout_us,
                             unsigned* report, void* stream);
/* One wave on `stream` waits (at most timeout_us <= 1e6) for words[0] != 0 and then writes words[1] = 1 if it saw it, else
 * 0.  Set words[0] from ANOTHER stream after this call: words[1] tells whether the two streams really run concurrently
 * (they do not under kernel-serialising profilers / launch-blocking modes / a single hardware queue).  The fed sweeps
 * below REQUIRE that concurrency; callers probe once per stream pair and otherwise use the sequential order. */
int pgasr_stream_probe(unsigned* words, int timeout_us, void* stream);
/* out_drop (optional, with drop_p in (0,1)): the sweep also writes dropout(out) there -- nn.LSTM's inter-layer dropout
 * (model.py:42), i.e. the NEXT layer's input, with exactly pgasr_dropout(out, drop_p, drop_seed, drop_offset)'s mask;
 * out itself stays un-dropped (the backward pass needs h_t).  NULL / 0: no second output. */
int pgasr_lstm_layer_fwd(float* gates, float* out, float* cbuf, const void* whh_pack_fwd,
                         const int32_t* lengths, int T, int B, int flags,
                         float* out_drop, float drop_p, uint64_t drop_seed, uint32_t drop_offset,
                         void* workspace, size_t workspace_bytes, void* stream);
/* pgasr_lstm_layer_fwd whose gates rows are produced WHILE it runs by pgasr_gemm_x3w_feed_f32 (launched after this
 * call on another stream): fed = that call's tiles_done, fed_need = 8H/256 column tiles per direction.  The helper
 * workgroups wait for a step's row tile(s) before they stage its rows and read them with agent-scope loads.
 * PGASR_ERR_UNSUPPORTED when no XCD would be left for the GEMM (B > 32) or helpers are off (flags bit 2):
 * pgasr_lstm_fed_ok(T, B, flags) != 0 says beforehand whether this call is possible. */
int pgasr_lstm_layer_fwd_fed(float* gates, float* out, float* cbuf, const void* whh_pack_fwd,
                             const int32_t* lengths, int T, int B, int flags, const unsigned* fed, int fed_need,
                             float* out_drop, float drop_p, uint64_t drop_seed, uint32_t drop_offset,
                             void* workspace, size_t workspace_bytes, void* stream);
int pgasr_lstm_fed_ok(int T, int B, int flags);
/* Backward counterpart: dout (= the input gradient of the layer above) is produced while the sweep runs by
 * pgasr_gemm_x3w_feed_f32(order = 1); fed_need = 2H/256.  drop_p != 0: dout arrives WITHOUT the inter-layer dropout
 * mask (model.py:42) and the helper workgroups apply pgasr_dropout(drop_p, drop_seed, drop_offset)'s mask to the rows
 * they stage -- the separate dropout pass between the layers disappears. */
int pgasr_lstm_layer_bwd_fed(float* gates, const float* out, const float* cbuf, const float* dout,
                             const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                             float* dbias_part, const unsigned* fed, int fed_need, float drop_p, uint64_t drop_seed,
                             uint32_t drop_offset, void* workspace, size_t workspace_bytes, void* stream);
int pgasr_lstm_layer_bwd(float* gates, const float* out, const float* cbuf, const float* dout,
                         const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                         float* dbias_part, void* workspace, size_t workspace_bytes, void* stream);
/* STREAMED backward sweep (round 3): the sweep publishes its progress so that the SAME layer's weight-gradient products
 * (pgasr_lstm_wgrads_streamed, on another stream, launched after this call) consume its d(pre-activation) rows while it
 * runs, instead of waiting for its end.  The products are sums over n = pgasr_lstm_wgrad_slabs(T, edges, max) TIME slabs
 * 0 = h_0 < h_1 < .. < h_n = T (sizes 16, 24, 32, 40, 48, 64, 80, 104, 128, 128 .. frames: small where a sweep ends); sweep step s is
 * frame T-1-s for direction 0 and frame s for direction 1, so for both directions the sweep steps < T - h_(n-k) complete
 * the first k slabs of the direction's order.  slab_done[c], c = 2 * (16-utterance group) + direction (2 * ceil(B/16)
 * words, zeroed by the caller BEFORE this launch), counts those publications k = 1..n: the rows are then in memory and
 * readable with agent-scope loads from any XCD.  The rows leave the storer waves as ordinary stores; one extra
 * ("flusher") workgroup per cluster waits until every member's stores of a slab are acknowledged, writes the XCD's L2
 * back (agent-scope release) and then publishes the count -- nothing on the sweep's dependent chain (measured: 1.4 us
 * of sweep time per publication).  fed == NULL: dout is complete (as pgasr_lstm_layer_bwd); else as
 * pgasr_lstm_layer_bwd_fed.  Conditions as for the fed sweeps (pgasr_lstm_fed_ok), else PGASR_ERR_UNSUPPORTED. */
int pgasr_lstm_wgrad_slabs(int T, int* edges, int max_edges);
int pgasr_lstm_layer_bwd_streamed(float* gates, const float* out, const float* cbuf, const float* dout,
                                  const void* whh_pack_bwd, const int32_t* lengths, int T, int B, int flags,
                                  float* dbias_part, const unsigned* fed, int fed_need, float drop_p, uint64_t drop_seed,
                                  uint32_t drop_offset, unsigned* slab_done,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A7  CTC prefix beam search (CTCdecoder.py:41-116), one workgroup per utterance.
 *   log_probs: natural-log probabilities, element (t,b,v) at log_probs[t*stride_t + b*stride_b + v],
 *              fp32 (is_f64 = 0: fp64 score carries, fp32 exp/log on differences, ~1e-7 relative) or
 *              fp64 (is_f64 != 0: exact fp64 math; the drop-in CTCDecoder.decode passes
 *              numpy's log of its probability matrix, like CTCdecoder.py:55);
 *   lengths (B) frames per utterance (NULL = T); beam <= 128, V <= 64, beam*V LDS-limited.
 *   out_tokens (B,T) best prefix, out_len (B), out_score (B) = -logsumexp(p_blank, p_nonblank)
 *   of that prefix (CTCdecoder.py:115-116).  Candidate order, prefix merging and the stable
 *   descending sort (ties -> first insertion) follow the reference exactly; scores are fp64.
 *   flags bit 0: out_tokens / out_len are given AFTER collapse_fn (adjacent duplicate symbols removed,
 *   CTCdecoder.py:119-131) -- the string policy_grad.py:8 and model.py:326 score; bit 1: never take the
 *   single-wave kernel.  fp32 input with beam <= 16, V <= 64 (8 symbols per lane up to 32, 16 beyond) and T * beam <= 24576 (the reward hypothesis inside
 *   the train step) runs as ONE WAVE per utterance with the prefix trie in LDS (no workspace traffic); the scores
 *   of the two fp32 kernels agree to ~1e-7 relative, their hypotheses wherever no two candidates are closer than that.
 * ---------------------------------------------------------------------------------------- */
size_t pgasr_beam_workspace_bytes(int T, int B, int V, int beam);
int pgasr_ctc_beam_search(const void* log_probs, int is_f64, long long stride_t, long long stride_b,
                          const int32_t* lengths, int T, int B, int V, int beam, int blank, int flags,
                          int32_t* out_tokens, int32_t* out_len, double* out_score,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise pieces of the train step.
 * pgasr_dropout: inverted dropout y = keep ? x/(1-p) : 0 (nn.Dropout, model.py:45,51 p=0.5; LSTM
 *   inter-layer dropout model.py:42 p=0.3).  keep is a pure function of (seed, offset, element
 *   index) (Philox4x32-10 on counter (index/4, offset), word index%4 >= p*2^32), so the backward
 *   pass re-applies the SAME call to the gradient instead of storing a mask.  x == y allowed.
 *   dact_y (optional, n elements): the result is also multiplied by (dact_y > 0 ? 1 : slope) -- the backward
 *   of F.leaky_relu (model.py:50) fused into the backward of the dropout that follows it (model.py:51).
 * pgasr_adam_step: torch.optim.Adam update (model.py:207, lr=5e-4) on flat fp32 buffers;
 *   step is the 1-based count of CALLS.  guard0 / guard1 (optional, device int32 words, e.g. the sweep workspaces'
 *   error words, or -- data parallel -- the word of the gradient buffer that carried every rank's error flag through
 *   the all-reduce, so that all replicas skip together): the update is skipped -- parameters and moments untouched --
 *   while either is != 0.  applied (optional, two device int32 words, zeroed once by the owner): the number of updates
 *   really applied, kept on the device in ping-pong fashion (word (step-1)&1 is read, word step&1 written); the bias
 *   correction then uses that count + 1 instead of `step`, so a skipped update does not shift it.  NULL: `step` is used.
 * ---------------------------------------------------------------------------------------- */
int pgasr_dropout(const float* x, float* y, unsigned long long n, float p, uint64_t seed, uint32_t offset,
                  const float* dact_y, float slope, void* stream);
/* A0 batch hand-over (model.py:227-230, `.to(device)`): copies `bytes` from src to dst (both 16-byte aligned) with
 * `workgroups` (<= 1024) streaming workgroups on `stream`.  src may be pinned host memory that is mapped into the
 * device's address space (hipHostMalloc, PyTorch's pinned tensors): the launch never blocks the host, and on a stream of
 * its own the copy runs beside the train step instead of in front of it. */
int pgasr_stream_copy(const void* src, void* dst, unsigned long long bytes, int workgroups, void* stream);
int pgasr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, unsigned long long n,
                    int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                    const int32_t* guard0, const int32_t* guard1, int32_t* applied, void* stream);

/* ------------------------------------------------------------------------------------------
 * N3  feature front end (data.py:44-79): MFCC(40) + delta + delta-delta of torchaudio's defaults
 *   (16 kHz, n_fft 400, hop 200, periodic Hann, reflect-centred frames, power spectrum, 128 HTK mel bands,
 *   10*log10 floored at the utterance's max - top_db, orthonormal DCT-II; 5-tap delta filter with replicate
 *   padding).  The DFT / mel / DCT contractions are pgasr_gemm_f32 calls made by the host between these:
 *   pgasr_feat_frames:  wave (B rows of wave_stride samples), n_samples (B), n_frames (B) = 1 + n_samples/200
 *                       -> frames (B*Tmax, 400), windowed, zero rows past an utterance's frames
 *   pgasr_feat_power:   spec (rows, 402) = [re(201) | im(201)] -> power (rows, 201)
 *   pgasr_feat_db:      in place on mel (B, Tmax, n_mels)
 *   pgasr_feat_deltas_stack: mfcc (B, Tmax, n_mfcc) -> feat (B, 3*n_mfcc, Tmax) zero padded (the reference's
 *                       batch layout, data.py:71-79), fmask (B, 1, Tmax) (optional)
 * ---------------------------------------------------------------------------------------- */
int pgasr_feat_frames(const float* wave, const int32_t* n_samples, const int32_t* n_frames, int B,
                      long long wave_stride, int Tmax, float* frames, void* stream);
int pgasr_feat_power(const float* spec, long long rows, float* power, void* stream);
int pgasr_feat_db(float* mel, const int32_t* n_frames, int B, int Tmax, int n_mels, float top_db, void* stream);
int pgasr_feat_deltas_stack(const float* mfcc, const int32_t* n_frames, int B, int Tmax, int n_mfcc,
                            float* feat, float* fmask, void* stream);
/* x (B x Tmax x C) -> feat (B x C x Tmax), 0 past n_frames[b], and fmask (B x 1 x Tmax): the layout of data.py:64-79 for a front
 * end without deltas -- the 80-band log-mel features (F = 80 of the benchmark; features.LogMel). */
int pgasr_feat_stack(const float* x, const int32_t* n_frames, int B, int Tmax, int C, float* feat, float* fmask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PGASR_HIP_H */
