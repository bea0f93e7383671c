/*
 * icap - caption-decoding engine for AMD Instinct MI355X (gfx950).  C ABI.
 *
 * This header is the drop-in boundary for the batched show-attend-tell inference
 * path of king821221/image-caption.  The reference has no FFI of its own (it is
 * pure Python on TensorFlow eager), so every entry point below names the Python
 * call site whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a CALLER-OWNED DEVICE pointer (HIP global memory), row-major
 *     and contiguous unless a leading dimension is given; weights use the Keras
 *     layouts: Dense.kernel [in,out], LSTMCell.kernel [C+E,4H] /
 *     recurrent_kernel [H,4H] / bias [4H] with gate order i,f,g,o.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls only
 *     enqueue work; they never synchronise the stream and never allocate, except
 *     icap_engine_create / _destroy / _refresh_weights (and their icap_tengine_ twins),
 *     icap_engine_set_timing (creates the event pool) and icap_engine_kernel_times (waits
 *     for the recorded events).
 *   - return value: ICAP_OK (0) or a negative ICAP_ERR_* code; the message is kept
 *     per host thread and read with icap_last_error().  Nothing throws across the ABI.
 *   - bool tensors are one byte per element (0/1), ids and lengths are int32.
 */
#ifndef ICAP_H_
#define ICAP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICAP_OK 0
#define ICAP_ERR_INVALID (-1)     /* bad argument: null pointer, shape, alignment */
#define ICAP_ERR_HIP (-2)         /* a HIP runtime call failed */
#define ICAP_ERR_UNSUPPORTED (-3) /* valid request outside the built limits */

#define ICAP_ACT_NONE 0
#define ICAP_ACT_RELU 1

#define ICAP_TILE_REFERENCE 0 /* row r uses image r % B  (tf.tile, recurrent_example.py:486) */
#define ICAP_TILE_REPEAT 1    /* row r uses image r / W  (self-consistent layout) */

#define ICAP_PRECISION_FP32 0 /* exact-f32 MFMA everywhere */
#define ICAP_PRECISION_F32_SPLIT 2 /* fp32-accurate products on the bf16 matrix pipe: every fp32
                                    * operand split exactly into three bf16 pieces, the six
                                    * partial products of weight >= 2^-16 summed in fp32 */
#define ICAP_PRECISION_BF16 1 /* bf16 operands on MFMA for every dense contraction, f32 accumulate */

#define ICAP_CELL_LSTM 0 /* decoder/lstm_recurrent_decoder.py: state [h, c], query and beta gate from c */
#define ICAP_CELL_GRU 1  /* decoder/recurrent_decoder.py: state [h], query and beta gate from h */

#define ICAP_MAX_BEAM_WIDTH 16

typedef void* icap_stream_t;

/* Library version (major*10000 + minor*100 + patch). */
int icap_version(void);

/* Copies the calling thread's last error message (NUL terminated, truncated to
 * `len`) and returns the full message length. */
size_t icap_last_error(char* buf, size_t len);

/* ---- stateless operators ------------------------------------------------------ */

/* y[M,N] = act(x[M,K] . w[K,N] + bias[N]).  bias may be NULL.
 * Replaces every tf.keras.layers.Dense on the path: encoder/mlp_encoder.py:17-18
 * (act = RELU), attention/attention.py:48 (W1, W2), decoder/init_state.py:34,
 * decoder/lstm_recurrent_decoder.py:106 (fc1) and :112 (fc2). */
int icap_dense_f32(const float* x, const float* w, const float* bias, float* y,
                   int M, int K, int N, int act, icap_stream_t stream);

/* pooled[G,D] = mean over L of features[G,L,D]          (mask == NULL)
 *             = sum over L of features * mask[G,L], 0 where the mask row sums to 0.
 * Replaces the pooling of FeatureProjState.__call__, decoder/init_state.py:15-32. */
int icap_pool_features_f32(const float* features, const float* mask, float* pooled,
                           int G, int L, int D, icap_stream_t stream);

/* dst[N,H] rows <- scale * src[B,H] rows, N = B*W; row r reads image r % B
 * (ICAP_TILE_REFERENCE) or r / W (ICAP_TILE_REPEAT).  Replaces tf.tile at
 * recurrent_example.py:486 applied to per-image state, and the
 * `reset_state_from_feat *` scaling at decoder/lstm_recurrent_decoder.py:129. */
int icap_expand_rows_f32(const float* src, float* dst, int B, int W, int H,
                         float scale, int tile_mode, icap_stream_t stream);

/* out[N,E] = table[ids[n], :]; ids outside [0,V) give a zero row.
 * Replaces EmbeddingLayer.call, decoder/embedding.py:9-10. */
int icap_embedding_f32(const float* table, const int32_t* ids, float* out,
                       int N, int V, int E, icap_stream_t stream);

/* Additive attention over L locations, one feature block shared by a group of rows.
 *   score[n,l] = sum_u v[u] * tanh(pf[g,l,u] + q[n,u]) + bv (+ score_bias[n,l])
 *   weights[n,:] = softmax_l(score[n,:]);  ctx[n,:] = sum_l weights[n,l] * features[g,l,:]
 *   if gate_state != NULL: ctx[n,:] *= sigmoid(gate_state[n,:] . gate_w + gate_b)
 * pf = W1(features) + b1 [G,L,U] (hoisted, step-invariant), q = W2(query) + b2 [N,U].
 * Group g owns rows n = g*row_stride_g + k*row_stride_k, k in [0, rows_per_group).
 * Replaces BahdanauAttention.call, attention/attention.py:45-63, and the beta gate
 * at decoder/lstm_recurrent_decoder.py:93-96.  Requires U % 4 == 0, D % 4 == 0. */
int icap_attention_f32(const float* pf, const float* features, const float* q,
                       const float* v, float bv, const float* score_bias,
                       const float* gate_state, const float* gate_w, float gate_b,
                       float* ctx, float* weights,
                       int G, int rows_per_group, int row_stride_g, int row_stride_k,
                       int L, int U, int D, int H, icap_stream_t stream);
/* The same with another activation inside the score than tanh (BahdanauAttention's
 * `activation` argument, attention/attention.py:31,47); these serve rows_per_group = 1. */
#define ICAP_ATT_ACT_TANH 0
#define ICAP_ATT_ACT_LINEAR 1
#define ICAP_ATT_ACT_RELU 2
#define ICAP_ATT_ACT_SIGMOID 3
int icap_attention_act_f32(const float* pf, const float* features, const float* q,
                           const float* v, float bv, const float* score_bias,
                           const float* gate_state, const float* gate_w, float gate_b,
                           float* ctx, float* weights,
                           int G, int rows_per_group, int row_stride_g, int row_stride_k,
                           int L, int U, int D, int H, int activation, icap_stream_t stream);

/* One Keras LSTMCell step with the input concatenation and embedding gather fused:
 *   x = [ctx[n,:C] | emb_n],  emb_n = table[ids[n]] (ids != NULL) or emb[n] (ids == NULL)
 *   z = x . kernel + h . recurrent_kernel + bias;  i,f,g,o = split(z, 4)
 *   c' = sig(f)*c + sig(i)*tanh(g);  h' = sig(o)*tanh(c')
 * emb_proj (optional, with ids): [V,4H] = table . kernel[C:C+E] computed once with
 * icap_dense_f32(table, kernel + C*4H, NULL, emb_proj, V, E, 4H, ...).  The token-only
 * part of z then is the row emb_proj[ids[n]] and the product runs over [ctx | h] only
 * (a quarter less work at C = E = H/2); `table` may be NULL in that case.
 * c_out may be the same buffer as c (each element is read and written by one thread);
 * h_out must be a different buffer from h (every workgroup reads all of h).
 * Replaces tf.concat + tf.keras.layers.LSTMCell at
 * decoder/lstm_recurrent_decoder.py:100-102 (and embedding.py:10 when ids != NULL). */
int icap_lstm_step_f32(const float* ctx, const float* emb, const float* table,
                       const int32_t* ids, const float* emb_proj, const float* h,
                       const float* c, const float* kernel, const float* recurrent_kernel,
                       const float* bias, float* h_out, float* c_out,
                       int N, int C, int E, int H, int V, icap_stream_t stream);

/* One Keras GRUCell step (TF 2.3 defaults: reset_after=True, gates z, r, h) with the same
 * input handling as icap_lstm_step_f32:
 *   x = [ctx | emb_n];  mx = x . kernel + bias[0];  mh = h . recurrent_kernel + bias[1]
 *   z = sig(mx_z + mh_z);  r = sig(mx_r + mh_r);  hh = tanh(mx_h + r*mh_h)
 *   h' = z*h + (1-z)*hh
 * The product runs on the LSTM gate kernels over operands re-laid as four columns per unit
 * (z, r, x_h, rec_h; zero rows where a column has no input or no recurrent part):
 * icap_gru_pack_f32 builds kernel4 [CE,4H], recurrent4 [H,4H], bias4 [4H] from the Keras
 * variables kernel [CE,3H], recurrent_kernel [H,3H], bias [2,3H] (CE = C+E), once per
 * weight version.  emb_proj (optional, with ids) = table . kernel4[C:C+E], [V,4H].
 * h_out must be a different buffer from h.
 * Replaces tf.concat + tf.keras.layers.GRUCell at decoder/recurrent_decoder.py:97-99. */
int icap_gru_pack_f32(const float* kernel, const float* recurrent_kernel, const float* bias,
                      float* kernel4, float* recurrent4, float* bias4, int CE, int H,
                      icap_stream_t stream);
int icap_gru_step_f32(const float* ctx, const float* emb, const float* table,
                      const int32_t* ids, const float* emb_proj, const float* h,
                      const float* kernel4, const float* recurrent4, const float* bias4,
                      float* h_out, int N, int C, int E, int H, int V, icap_stream_t stream);

/* ---- bf16-operand mode (ICAP_PRECISION_BF16) -------------------------------------
 * The dense contractions take bf16 operands (round-to-nearest-even) and accumulate in
 * fp32; everything else (attention scores, softmax, cell update, beam search) is fp32.
 * Weights are packed once: bf16, transposed to [N_pad][K_total] with k contiguous and
 * N_pad = N rounded up to 128.  icap_pack_weight_bf16 writes source rows [0,K_src) of
 * w[K_src,N] (Keras [in,out] layout) to k positions [k_offset, k_offset+K_src), so a
 * product over concatenated inputs is packed with one call per source matrix.
 * gate_units = H > 0 (N == 4H, H % 32 == 0) interleaves the LSTM gates in groups of 32
 * units as icap_lstm_step_bf16 expects; 0 keeps the column order for icap_dense_bf16.
 * K must be a multiple of 64.  Same reference call sites as icap_dense_f32 /
 * icap_lstm_step_f32 (Dense: encoder/mlp_encoder.py:12, attention/attention.py:34-35,
 * decoder/lstm_recurrent_decoder.py:39-50; LSTMCell: :33-37,102). */
size_t icap_packed_weight_bytes(int K, int N);
int icap_pack_weight_bf16(const float* w, void* packed, int K_src, int N, int k_offset,
                          int K_total, int gate_units, icap_stream_t stream);
int icap_dense_bf16(const float* x, const void* packed_w, const float* bias, float* y,
                    int M, int K, int N, int act, icap_stream_t stream);
/* packed_w: [ctx | h] rows (kernel[0:C] then recurrent_kernel), gates interleaved.
 * emb_proj/ids: optional token-only gate term (see icap_lstm_step_f32); bias in the
 * original [4H] order.  c_out may alias c, h_out must not alias h. */
int icap_lstm_step_bf16(const float* ctx, const float* h, const float* c,
                        const void* packed_w, const float* bias, const float* emb_proj,
                        const int32_t* ids, float* h_out, float* c_out,
                        int N, int C, int H, int V, icap_stream_t stream);
/* GRU step with bf16 operands: packed_w = icap_pack_weight_bf16 of the four-column operands
 * of icap_gru_pack_f32 (kernel4 rows [0,C), then recurrent4, gate_units = H), bias4 [4H].
 * Same cell arithmetic as icap_gru_step_f32 (decoder/recurrent_decoder.py:97-99). */
int icap_gru_step_bf16(const float* ctx, const float* h, const void* packed_w, const float* bias4,
                       const float* emb_proj, const int32_t* ids, float* h_out,
                       int N, int C, int H, int V, icap_stream_t stream);

/* ---- split mode (ICAP_PRECISION_F32_SPLIT) -----------------------------------------
 * Same contractions with fp32 accuracy at 6/16 of the exact-f32 matrix-pipe time: an fp32
 * number is exactly hi + mid + lo with three bf16 pieces, the product of two bf16 numbers is
 * exact in fp32, and hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid differs from a*b by
 * at most 2^-23 |a*b| (the dropped pieces have weight 2^-24 and 2^-32).  Sums are fp32.
 * Weights are split once into three planes, each packed like icap_pack_weight_bf16 packs
 * its one: [3][N_pad][K_total] bf16; arguments as for the bf16 entry points, K % 32 == 0.
 * icap_dense_topk_split writes the records of icap_dense_topk_f32. */
size_t icap_packed_split_bytes(int K, int N);
int icap_pack_weight_split(const float* w, void* packed, int K_src, int N, int k_offset,
                           int K_total, int gate_units, icap_stream_t stream);
int icap_dense_split(const float* x, const void* packed_w, const float* bias, float* y,
                     int M, int K, int N, int act, icap_stream_t stream);
int icap_dense_topk_split(const float* x, const void* packed_w, const float* bias, int M,
                          int K, int N, int eos_token, int top_k, float* records,
                          float* eos_logit, icap_stream_t stream);
int icap_lstm_step_split(const float* ctx, const float* h, const float* c,
                         const void* packed_w, const float* bias, const float* emb_proj,
                         const int32_t* ids, float* h_out, float* c_out,
                         int N, int C, int H, int V, icap_stream_t stream);

/* Fused beam-search step for B images x W beams x V words: log-softmax, finished
 * masking, accumulated log-prob, length penalty, top-W over the flattened W*V
 * candidates (ties: lower flat index first), gather of log-probs / finished /
 * lengths by parent beam.  Reads logits once (rows up to 65536 words).
 *   in : logits[B,W,V]; log_probs[B,W] f32; finished[B,W] u8; lengths[B,W] i32
 *   out: scores[B,W] f32; predicted_ids[B,W] i32; parent_ids[B,W] i32;
 *        next_log_probs[B,W]; next_finished[B,W]; next_lengths[B,W]
 * The next_* buffers may be the same buffers as the inputs (in-place update).
 * time == 0 scores only beam `start_offset` and gathers from beam 0, as the
 * reference does.  input_is_log_prob != 0 skips the log-softmax (custom prob_fn).
 * Replaces beam_search_step_batch, beam_search/beam_search.py:360-543 (with
 * mask_probs :165-197, hyp_score :131-141, choose_top_k :144-148, batch_gather
 * :347-358).  Requires 1 <= W <= ICAP_MAX_BEAM_WIDTH and W <= V.
 * `workspace` is caller-owned device scratch of at least
 * icap_beam_step_workspace_bytes(B, W) bytes, 16-byte aligned; its contents need not
 * be preserved between calls, but two calls that overlap in time need their own. */
size_t icap_beam_step_workspace_bytes(int B, int W);
int icap_beam_step_f32(const float* logits, const float* log_probs,
                       const uint8_t* finished, const int32_t* lengths,
                       int time, int start_offset, int eos_token,
                       double length_penalty_weight, int input_is_log_prob,
                       float* scores, int32_t* predicted_ids, int32_t* parent_ids,
                       float* next_log_probs, uint8_t* next_finished,
                       int32_t* next_lengths, int B, int W, int V,
                       void* workspace, size_t workspace_bytes, icap_stream_t stream);
/* The same step with R input rows (beams) per image and W successors out: logits[B,R,V],
 * log_probs / finished / lengths [B,R]; outputs [B,W]; parent_ids in [0,R); start_offset in
 * [0,R); workspace of icap_beam_step_workspace_bytes(B, R) bytes; R, W <=
 * ICAP_MAX_BEAM_WIDTH.  beam_search_step_batch takes any number of rows
 * (beam_search.py:375-455 reshapes by vocab size only); the beam-search-optimisation trainer
 * calls it with the gold row in front of the search rows, R = W + 1 and start_offset = 1
 * (utils/train_utils.py:275-279). */
int icap_beam_step_rows_f32(const float* logits, const float* log_probs,
                            const uint8_t* finished, const int32_t* lengths,
                            int time, int start_offset, int eos_token,
                            double length_penalty_weight, int input_is_log_prob,
                            float* scores, int32_t* predicted_ids, int32_t* parent_ids,
                            float* next_log_probs, uint8_t* next_finished,
                            int32_t* next_lengths, int B, int R, int W, int V,
                            void* workspace, size_t workspace_bytes, icap_stream_t stream);

/* icap_beam_step_f32 for callers that cannot allocate device memory or hand over a
 * workspace: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference.  Same arguments, arithmetic and results as
 * icap_beam_step_f32 without the workspace pair: the candidate workspace is a fixed array
 * in device memory (ICAP_BEAM_STEP_STATIC_MAX_ROWS rows of 200 bytes = 3.2 MB, held by every
 * GPU the library runs on until the process ends).  B*W above that limit returns
 * ICAP_ERR_UNSUPPORTED.  The call allocates nothing.  The FIRST call on a device looks the
 * array's device address up (one runtime query, kept per device) and must therefore be made
 * outside stream capture, with the stream's device current - any valid call serves as that
 * warm-up; a first call under capture returns ICAP_ERR_INVALID.  Every later call on that
 * device only enqueues and may be captured.  All calls on one GPU share the array: calls that
 * overlap in time on one GPU (two streams) race - order them or use icap_beam_step_f32. */
#define ICAP_BEAM_STEP_STATIC_MAX_ROWS 16384
int icap_beam_step_static_f32(const float* logits, const float* log_probs,
                              const uint8_t* finished, const int32_t* lengths,
                              int time, int start_offset, int eos_token,
                              double length_penalty_weight, int input_is_log_prob,
                              float* scores, int32_t* predicted_ids, int32_t* parent_ids,
                              float* next_log_probs, uint8_t* next_finished,
                              int32_t* next_lengths, int B, int W, int V, icap_stream_t stream);

/* The beam-search step in two halves, for a caller-supplied successor choice
 * (BeamSearchConfig.choose_successors_fn other than choose_top_k).
 * icap_beam_scores_f32 materialises, for every (image, beam, word),
 *   totals[B,W,V] = log_probs + masked log-softmax(logits)      beam_search.py:396-409
 *   scores[B,W,V] = totals / length_penalty(lengths')           beam_search.py:416-445
 * (a live beam grows by one except at eos; a finished beam keeps its length and has
 * all its mass on eos).  The caller picks K successors per image as flat indices into
 * [W*V] (at time 0: into the V words of one beam) and icap_beam_gather_f32 does the
 * rest of the step, beam_search.py:475-529: word = idx mod V, parent = idx div V,
 * next_log_probs = totals[b, idx], finished / lengths gathered by parent and updated.
 * Indices must lie in [0, W*V); the kernel clamps, the Python front end raises. */
int icap_beam_scores_f32(const float* logits, const float* log_probs,
                         const uint8_t* finished, const int32_t* lengths, int eos_token,
                         double length_penalty_weight, int input_is_log_prob,
                         float* scores, float* totals, int B, int W, int V,
                         icap_stream_t stream);
int icap_beam_gather_f32(const float* totals, const uint8_t* finished,
                         const int32_t* lengths, const int32_t* word_indices,
                         int eos_token, int32_t* predicted_ids, int32_t* parent_ids,
                         float* next_log_probs, uint8_t* next_finished,
                         int32_t* next_lengths, int B, int W, int V, int K,
                         icap_stream_t stream);

/* Stand-alone helpers with the reference's semantics.
 * icap_mask_probs_f32: rows of finished beams become [f32.min .. 0@eos .. f32.min],
 *   computed as (1-fm)*row + fm*probs like beam_search/beam_search.py:165-197.
 * icap_hyp_score_f32: out[i] = log_probs[i] / ((5+len[i])^w / 6^w), beam_search.py:131-141;
 *   with log_probs == NULL it returns the penalty itself (length_penalty, :115-128).
 * icap_batch_gather_f32: out[b,y] = a[b, idx[b,y]], beam_search.py:347-358 (the reference
 *   round-trips through float32; so does this).  Indices must lie in [0, X). */
int icap_mask_probs_f32(const float* probs, const uint8_t* finished, float* out,
                        int eos_token, int N, int V, icap_stream_t stream);
int icap_hyp_score_f32(const float* log_probs, const int32_t* lengths,
                       double length_penalty_weight, float* out, long long n,
                       icap_stream_t stream);
int icap_batch_gather_f32(const float* a, const int32_t* idx, float* out, int B, int X,
                          int Y, icap_stream_t stream);

/* Fused form of the vocab projection and the beam step (the engine's default from about
 * three 128x128 projection tiles per compute unit upwards): the [M,N] logits are never
 * written to memory.
 * icap_dense_topk_f32 computes x[M,K] . w[K,N] + bias like icap_dense_f32 (same kernel,
 * same sums) but its epilogue keeps, for every row and every 64-column half tile, one
 * record of icap_dense_topk_record_floats(top_k) = 2 + 2*top_k floats:
 *   [max, sum exp(x - max), the top_k largest logits, their columns (int32 bits,
 *    0x7fffffff = none; equal logits: lower column first)]
 * records is [M][icap_dense_topk_half_tiles(N)][2 + 2*top_k], 16-byte aligned;
 * eos_logit[M] receives the logit of column eos_token (untouched if it is outside [0,N)).
 * Needs top_k in {1,3,5}, K % 32 == 0, N % 4 == 0, 16-byte aligned x and w.
 * icap_beam_step_from_records_f32 is icap_beam_step_f32 (start_offset 0, log-softmax
 * input) reading those records (M = B*W rows, W <= top_k).  The two together replace fc2
 * at decoder/lstm_recurrent_decoder.py:112 and beam_search_step_batch,
 * beam_search/beam_search.py:360-543. */
int icap_dense_topk_half_tiles(int N);
int icap_dense_topk_record_floats(int top_k);
int icap_dense_topk_f32(const float* x, const float* w, const float* bias, int M, int K,
                        int N, int eos_token, int top_k, float* records, float* eos_logit,
                        icap_stream_t stream);
int icap_beam_step_from_records_f32(const float* records, const float* eos_logit,
                                    int half_tiles, int top_k, const float* log_probs,
                                    const uint8_t* finished, const int32_t* lengths,
                                    int time, int eos_token, double length_penalty_weight,
                                    float* scores, int32_t* predicted_ids,
                                    int32_t* parent_ids, float* next_log_probs,
                                    uint8_t* next_finished, int32_t* next_lengths,
                                    int B, int W, int V, icap_stream_t stream);

/* The same step for beam widths up to ICAP_MAX_BEAM_WIDTH on records that hold fewer logits
 * per half tile than the beam is wide (top_k = 5 only).  A half tile may then hide a
 * candidate; the kernel proves per image that none of the hidden ones can be among the W
 * successors (the last logit a record holds scores below a lower bound on the W-th best
 * score) and sets *overflow = 1 (device int32, zero on entry) where it cannot.  With
 * *overflow == 0 afterwards the outputs are those of icap_beam_step_f32 on the full logits;
 * otherwise the caller repeats the step there (the engine enqueues that repetition behind
 * this call, switched off on the device while the flag is down).  The next_* buffers must
 * not alias the state inputs.  compact_first != 0: time 0 on one record row per image
 * (only beam 0 was projected).  Same reference call sites as
 * icap_beam_step_from_records_f32. */
int icap_beam_step_from_records_wide_f32(const float* records, const float* eos_logit,
                                         int half_tiles, int top_k, const float* log_probs,
                                         const uint8_t* finished, const int32_t* lengths,
                                         int time, int eos_token, double length_penalty_weight,
                                         float* scores, int32_t* predicted_ids,
                                         int32_t* parent_ids, float* next_log_probs,
                                         uint8_t* next_finished, int32_t* next_lengths,
                                         int B, int W, int V, int32_t* overflow,
                                         int compact_first, icap_stream_t stream);

/* Back-tracking over T recorded steps: pick the best final beam per image
 * (unfinished beams get -1e9 when any beam of the image is finished, first max
 * wins), then follow parents backwards.  ids/parents are [T,B,W]; tokens is [B,T].
 * best_beam[B] and best_score[B] may be NULL.
 * Replaces loc_optimal_beam_path_batch, beam_search/beam_search.py:545-613. */
int icap_backtrack(const int32_t* ids, const int32_t* parents, const float* last_scores,
                   const uint8_t* last_finished, int32_t* tokens, int32_t* best_beam,
                   float* best_score, int T, int B, int W, icap_stream_t stream);

/* ---- fused whole-sequence engine ---------------------------------------------- */

typedef struct icap_config {
  int32_t B, W, L, F, D, E, H, U, V, T; /* D = encoder/context dim, U = attention units */
  int32_t start_id, eos_id;
  int32_t tile_mode;            /* ICAP_TILE_* */
  int32_t precision;            /* ICAP_PRECISION_* */
  double length_penalty_weight; /* a double: 6^w is folded in double, beam_search.py:127 */
  float reset_state_from_feat;  /* decoder/lstm_recurrent_decoder.py:129 */
  int32_t cell;                 /* ICAP_CELL_* (0 = LSTM, the field was reserved-as-zero before) */
} icap_config;

typedef struct icap_weights { /* device pointers; must outlive the engine */
  const float* enc_kernel;  const float* enc_bias;        /* [F,D], [D]   */
  const float* embedding;                                 /* [V,E]        */
  const float* init_h_kernel; const float* init_h_bias;   /* [D,H], [H]   */
  const float* init_c_kernel; const float* init_c_bias;   /* [D,H], [H]   (GRU: unused, may be NULL) */
  const float* att_w1_kernel; const float* att_w1_bias;   /* [D,U], [U]   */
  const float* att_w2_kernel; const float* att_w2_bias;   /* [H,U], [U]   */
  const float* att_v_kernel;  const float* att_v_bias;    /* [U,1], [1]   */
  const float* fc3_kernel;    const float* fc3_bias;      /* [H,1], [1]   */
  const float* lstm_kernel;                               /* [D+E,4H]   (GRU: [D+E,3H]) */
  const float* lstm_recurrent_kernel;                     /* [H,4H]     (GRU: [H,3H])   */
  const float* lstm_bias;                                 /* [4H]       (GRU: [2,3H])   */
  const float* fc1_kernel;    const float* fc1_bias;      /* [H,H], [H]   */
  const float* fc2_kernel;    const float* fc2_bias;      /* [H,V], [V]   */
} icap_weights;

typedef struct icap_engine icap_engine;

/* Allocates the workspace for one (device, config) pair on the current device and
 * reads the scalar biases (att_v_bias, fc3_bias) back once.  One engine serves one
 * stream at a time; distinct engines are independent.  The workspace includes a [B,L,U] table
 * exp(2 (W1 features + b1)), rebuilt in every decode, from which the attention scores
 * (attention/attention.py:47-50) take tanh(a + b) = 1 - 2 / (1 + exp(2a) exp(2b)) with one
 * reciprocal per element; pre-activations of magnitude 40 and more are scored with tanh itself
 * (ICAP_ATT_TABLE=0 in the environment at creation: no table). */
int icap_engine_create(const icap_config* config, const icap_weights* weights,
                       icap_engine** out);
int icap_engine_destroy(icap_engine* engine);
/* Recomputes everything the engine derives from the weights: the [V,4H] token part of the
 * LSTM gates, the GRU re-layout, the packed bf16 / split copies, and the two scalar biases
 * (att_v_bias, fc3_bias) that travel to the kernels by value.  icap_engine_create does this
 * once on the null stream; call it again after changing ANY weight value in place.  It
 * enqueues on `stream` behind the caller's writes and waits for the stream (the scalars are
 * read back to the host): the one engine call besides create / destroy that synchronises. */
int icap_engine_refresh_weights(icap_engine* engine, icap_stream_t stream);
/* Bytes of device workspace the engine holds. */
size_t icap_engine_workspace_bytes(const icap_engine* engine);

/* Whole-sequence decode: encoder -> tile -> reset_state -> T x (decoder step, beam
 * step) -> back-track, all enqueued on `stream` without host synchronisation.
 *   img[B,L,F] f32 -> tokens[B,T] i32 (+ best_score[B] f32, may be NULL)
 * Replaces predict_image_caption_sequence, recurrent_example.py:477-562. */
int icap_decode(icap_engine* engine, const float* img, int32_t* tokens,
                float* best_score, icap_stream_t stream);

/* The loop of the baseline beam searcher (SCST / actor-critic baselines) on the same engine:
 *   features[B,L,D] f32 (already encoded) -> tokens[B,T] i32, T = cfg.T = max_length - 1
 * tile -> reset_state -> for i = 1..T: decoder step, logits += start-token mask, with
 * seq_len also the forced-eos mask where i+1 == seq_len[image] and the forced-pad mask where
 * i+1 > seq_len[image], beam step at time i (never the time-0 form, so the first step
 * already ranks all W rows) -> back-track.  seq_len[B] i32 = number of non-pad tokens of the
 * target sequences (NULL: the max_length > 0 form without forced tokens); the caller
 * prepends the start-token column.  length_penalty_weight comes from the engine's config
 * (the searcher uses 1.0).  Replaces BaselineSequeneBeamSearcher.__call__,
 * utils/agent_utils.py:152-262. */
int icap_search_baseline(icap_engine* engine, const float* features, const int32_t* seq_len,
                         int pad_token, int32_t* tokens, float* best_score,
                         icap_stream_t stream);

/* The searcher's additive masks on a logits buffer [N,V], in place: float32.min on column
 * start_token (< 0: none); with row_len[N] (may be NULL) float32.min on every column but
 * eos_token where step_plus_1 == row_len[n], on every column but pad_token where
 * step_plus_1 > row_len[n].  Replaces utils/agent_utils.py:188-237 (and the three masks
 * of SequenceSampler, :81-107, with start = each of pad/start/eos in turn). */
int icap_logit_masks_f32(float* logits, int N, int V, int start_token, const int32_t* row_len,
                         int step_plus_1, int eos_token, int pad_token, icap_stream_t stream);

/* Copies the per-step records of the last icap_decode into caller buffers (device
 * pointers, any may be NULL): ids, parents [T,B,W] i32; scores, log_probs [T,B,W] f32
 * (BeamSearchStepOutput / BeamSearchState.log_probs of every step).  Enqueued on
 * `stream`, which must be the stream of the decode or ordered after it. */
int icap_engine_trace(const icap_engine* engine, int32_t* ids, int32_t* parents,
                      float* scores, float* log_probs, icap_stream_t stream);

/* Per-kernel timing with HIP events on the decode stream.  Launches of every kernel
 * class whose bit is set in `class_mask` (bit ICAP_K_*) are bracketed by events;
 * 0 disables timing.  icap_engine_kernel_times synchronises those events and returns,
 * per kernel class, the summed duration in milliseconds and the launch count since
 * the mask was last set.  icap_engine_set_timing creates the event pool (enough for
 * ICAP_TIMING_POOL_DECODES bracketed decodes between two reads; launches beyond it go
 * unbracketed), so that icap_decode itself never creates an event. */
#define ICAP_TIMING_POOL_DECODES 64
#define ICAP_K_ENCODE 0
#define ICAP_K_FEAT_PROJ 1
#define ICAP_K_INIT_STATE 2
#define ICAP_K_ATTN_QUERY 3
#define ICAP_K_ATTENTION 4
#define ICAP_K_LSTM 5
#define ICAP_K_FC1 6
#define ICAP_K_VOCAB 7
#define ICAP_K_BEAM 8
#define ICAP_K_BACKTRACK 9
#define ICAP_K_COUNT 10
int icap_engine_set_timing(icap_engine* engine, unsigned class_mask);
/* Brackets the selected classes in every n-th decode call only (n >= 1, default 1; the count
 * restarts when the mask is set): an event pair costs the stream a few microseconds per
 * launch, so a sample keeps a timed run close to an untimed one.  No counterpart in the
 * reference. */
int icap_engine_set_timing_every(icap_engine* engine, int n);
/* By default the attention of step t+1 runs on an engine-owned side stream beside the vocab
 * projection and beam step of step t (it only depends on the LSTM of step t).  on == 0
 * keeps every launch on the caller's stream, e.g. to time kernels one by one; results are
 * identical either way.  No counterpart in the reference (its loop is sequential,
 * recurrent_example.py:517-558). */
int icap_engine_set_overlap(icap_engine* engine, int on);
/* fc1 and fc2 of the decoder have no activation between them
 * (decoder/lstm_recurrent_decoder.py:41-42,106-112; the dropout between them is the identity at
 * inference), so logits = h . (W1 . W2) + (b1 . W2 + b2).  In fp32 and split mode the engine
 * forms the folded kernel and bias once (icap_engine_create / icap_engine_refresh_weights;
 * sums in double precision, rounded to float32 once) and projects the cell output directly,
 * which removes one launch per decode step; on == 0 keeps the reference's two products
 * (default: folded in fp32 mode, the two products in split mode; ICAP_FOLD_FC=0|1 in the
 * environment when the engine is created overrides).  The logits of the two forms differ by float32 rounding only (both
 * within the 1e-5 bar of the oracle).  ICAP_ERR_UNSUPPORTED in bf16 mode with on != 0. */
int icap_engine_set_fold_fc(icap_engine* engine, int on);
/* 1 when the engine carries the next step's query projection and attention as workgroups of
 * the vocab projection's launch (fp32 fused path with the folded output kernel, every beam
 * width up to ICAP_MAX_BEAM_WIDTH the attention shape fits; ICAP_MERGED=0|1 in the environment
 * at creation overrides, and =1 also selects the bf16 form of the launch in bf16 mode), 0 when
 * they run on the side stream. */
int icap_engine_merged_launch(const icap_engine* engine);
/* The current choices behind icap_engine_set_fold_fc (1: the folded fc1 . fc2 kernel projects
 * the cell output) and ICAP_ATT_TABLE (1: the attention scores use the exp(2 PF) table).  No
 * counterpart in the reference; they exist so that a benchmark line can say what it ran. */
int icap_engine_fold_fc(const icap_engine* engine);
int icap_engine_att_table(const icap_engine* engine);
/* 1 when greedy decodes of this engine (beam width 1, no masks, fp32, LSTM, a few dozen rows)
 * run their whole step loop (/root/reference/recurrent_example.py:517-558) as ONE
 * launch instead of five launches per step.  Opt-in: ICAP_SMALL_LOOP=1 in the environment when
 * the engine is created, and only for shapes the launch supports; the launch-per-kernel loop
 * is the default because it is the faster one on MI355X.  Same tokens and history either
 * way; the waits between its phases are bounded and a wait that runs out is reported through
 * icap_engine_decode_status like the merged launch's. */
int icap_engine_small_loop(const icap_engine* engine);
/* Validity of the last decode enqueued on this engine (icap_decode / icap_search_baseline).
 * The merged launch (icap_engine_merged_launch) hands the query projection q over INSIDE one
 * launch: its query tiles store q through to device scope and count themselves on a device
 * counter, and the attention workgroups of the same launch poll that counter before they read
 * q.  That is deadlock-free only because workgroups are dispatched in grid order and the query
 * tiles come first in the grid - true of this hardware's dispatcher, not an architectural
 * promise - so the poll is bounded (2^16 polls, a fraction of a second): a workgroup whose
 * wait runs out raises a device word and leaves its rows untouched, and the back-tracking
 * kernel then writes -1 into every token and NaN into every score instead of a wrong caption.
 * (The accesses are relaxed device-scope atomics with an explicit wait for the stores'
 * acknowledgement in front of the count: release / acquire fences at device scope write back
 * and invalidate a whole L2 under the running tiles and were measured at 232 instead of
 * 156 microseconds per launch.)  The call returns ICAP_OK either way, as every enqueueing call does;
 * this function copies the word back behind the decode on `stream` (the decode's stream or one
 * ordered after it), WAITS for the stream and sets *invalid to 1 when a wait ran out, else 0
 * (always 0 for decodes without such a hand-over).  A caller that hands tokens on should either
 * call it once per decode at its next synchronisation point or test the tokens for -1.
 * ICAP_MERGED_FORCE_TIMEOUT=1 in the environment at engine creation makes every wait
 * unmeetable (test knob).  No counterpart in the reference (its loop has no such hand-over,
 * recurrent_example.py:517-558). */
int icap_engine_decode_status(icap_engine* engine, int32_t* invalid, icap_stream_t stream);
/* Diagnostics of a decode that does not finish: hipStreamQuery of `stream` and of the engine's
 * side stream, hipEventQuery of the two events that join them (0 = complete, 600 = not ready),
 * without waiting for anything.  No counterpart in the reference. */
int icap_engine_debug_sync_state(const icap_engine* engine, icap_stream_t stream, int32_t state[4]);
/* The engine runs the vocab projection and the beam step fused (icap_dense_topk_f32 +
 * icap_beam_step_from_records_f32; in bf16 mode the bf16 projection writes the same
 * records) whenever its shape is served: for fp32 V % 4 == 0, H % 32 == 0 (beams wider than 5
 * through icap_beam_step_from_records_wide_f32 and its repetition on full logits); beam
 * width <= 5 in the bf16 and split modes.  on == 0 selects the unfused kernels (a [B*W,V] logits buffer,
 * icap_dense_f32 + icap_beam_step_f32); on != 0 returns ICAP_ERR_UNSUPPORTED for a shape
 * that is not served.  icap_engine_fused_vocab reports the current choice.  No
 * counterpart in the reference. */
int icap_engine_set_fused_vocab(icap_engine* engine, int on);
int icap_engine_fused_vocab(const icap_engine* engine);
int icap_engine_kernel_times(icap_engine* engine, float* ms /*[ICAP_K_COUNT]*/,
                             int32_t* launches /*[ICAP_K_COUNT]*/);

/* One sampling step of the sequence sampler of the policy-gradient trainers
 * (SequenceSampler.__call__, utils/agent_utils.py:81-125): f32.min is added to the start, eos
 * and pad columns of the logits (negative token: no mask), samples[n] is drawn from the
 * softmax of row n, and with row_len the reference's forcing applies: eos where
 * step + 1 == row_len[n], pad where step >= row_len[n] (row_len as sequence_length_fn
 * gives it).  The reference draws with tf.random.categorical, whose random stream is TF's own;
 * here the caller supplies one uniform number u[n] in [0,1) per row and the draw is the
 * inverse of the cumulative distribution at u[n], so the distribution and the forcing are
 * what can be checked, not the individual draws.  logits[N,V], u[N] f32, row_len[N] i32 or
 * NULL, samples[N] i32. */
int icap_sample_categorical_f32(const float* logits, const float* u, const int32_t* row_len,
                                int step, int start_token, int eos_token, int pad_token,
                                int32_t* samples, int N, int V, icap_stream_t stream);

/* ---- transformer decoder path ------------------------------------------------------
 * decoder/transform_decoder.py + attention/attention.py:160-562 + decoder/ffn_layer.py +
 * embedding/position_embedding.py, run incrementally: one pass of the layer stack per new
 * target position with per-row key / value caches (the reference re-runs the whole prefix
 * every step, transform_decoder.py:177; the results per position are the same).  No driver
 * in the reference constructs this decoder; the loop is predict_image_caption_sequence
 * (recurrent_example.py:477-562) with the decoder swapped. */

/* Self-attention mask handling.  AS_WRITTEN: the additive causal bias (0 / -1e9) is passed
 * as `attention_mask` (transform_decoder.py:139) and goes through (1 - mask) * -1e9
 * (attention.py:471-473): visible positions get -1e9 added to their fp32 score, hidden ones
 * about -1e18.  CAUSAL: the conventional additive bias. */
#define ICAP_TMASK_AS_WRITTEN 0
#define ICAP_TMASK_CAUSAL 1

/* [length, hidden] sinusoid table of RelativePositionEmbedding.call
 * (embedding/position_embedding.py:67-102), fp32, timescales 1 .. 1e4. */
int icap_position_table_f32(float* table, int length, int hidden, icap_stream_t stream);
/* x[n,:] = table[ids[n],:] + pos_row[:]  (ids NULL: token 0).  Embedding + position
 * encoding of one target position, transform_decoder.py:206-210.  D % 4 == 0. */
int icap_embed_pos_f32(const float* table, const int32_t* ids, const float* pos_row, float* x,
                       int N, int V, int D, icap_stream_t stream);
/* x[i] += o[i % period] in place, i < n: the wrapper's residual add x + y
 * (transform_decoder.py:42; period = n) and the position-encoding add broadcast over the
 * batch (transform_decoder.py:210; period = length * hidden). */
int icap_add_f32(float* x, const float* o, long long n, long long period, icap_stream_t stream);
/* Pre-norm wrapper arithmetic (transform_decoder.py:35,42): x += o when o != NULL (in
 * place), then y = LayerNormalization(x) * gamma + beta over the last axis with the biased
 * variance and epsilon `eps` (the stack uses 1e-6).  D % 4 == 0, D <= 4096. */
int icap_residual_layernorm_f32(float* x, const float* o, const float* gamma, const float* beta,
                                float* y, int N, int D, float eps, icap_stream_t stream);
/* Self-attention of the newest position `pos` of every row over its cached positions
 * 0..pos: q[N,heads*key_dim] (unscaled; 1/sqrt(key_dim) is applied to the query first,
 * attention.py:501); kv_cache[pos'][N][2*heads*key_dim] holds keys then values of position
 * pos' (the caller writes position `pos` before the call); out[N,heads*key_dim] is the
 * combined value, the input of the output projection.  Replaces
 * MultiHeadAttention._compute_attention, attention.py:476-524, for one query position.
 * key_dim % 4 == 0 and key_dim / 8 (or / 4) a power of two. */
int icap_mha_self_step_f32(const float* q, const float* kv_cache, float* out, int N, int heads,
                           int key_dim, int pos, int mask_mode, icap_stream_t stream);
/* Multi-head attention of rows_per_group query rows per group over the group's L key / value
 * rows: kv[G][L][2*heads*key_dim] holds keys then values (in the engine: of the L encoder
 * outputs of image g, rows grouped per image as in icap_attention_f32; for a whole
 * [B,T,dim] query: G = B, rows_per_group = T, row_stride_g = T, row_stride_k = 1).
 * mask (optional): attention_mask values [G][rows_per_group][L] (mask_per_group != 0) or
 * one [rows_per_group][L] block for every group, applied as (1 - mask) * -1e9 on the fp32
 * scores like MultiHeadAttention._masked_softmax (attention.py:461-474); the engine's
 * encoder-decoder attention passes none (attention_bias = None). */
int icap_mha_cross_step_f32(const float* q, const float* kv, float* out, int G, int rows_per_group,
                            int row_stride_g, int row_stride_k, int L, int heads, int key_dim,
                            const float* mask, int mask_per_group, icap_stream_t stream);

/* The general form of MultiHeadAttention._compute_attention (attention.py:476-524), for the
 * call forms the step kernels do not serve: key != value, value_dim != key_dim, several
 * attention axes, return_attention_scores (attention.py:259-270,526-562).
 * q [Bf][T][heads][key_dim], k [Bf][S][heads][key_dim] (projected, unscaled; the query is
 * multiplied by 1/sqrt(key_dim) first, `:501`), mask NULL or [Bf][heads][T][S] (1 = attend),
 * applied as (1 - mask) * -1e9 (`:461-474`); probs [Bf][heads][T][S].  The soft-max runs over
 * every run of softmax_segment consecutive key positions: S for one attention axis, the size of
 * the LAST key attention axis when several axes were flattened into S (the reference builds
 * Softmax(axis=-1), `:457-458`).  S <= 4096. */
int icap_mha_scores_f32(const float* q, const float* k, const float* mask, float* probs, int Bf, int T,
                        int S, int heads, int key_dim, int softmax_segment, icap_stream_t stream);
/* out[Bf][T][heads][value_dim] = sum_s probs[Bf][heads][T][s] * v[Bf][s][heads][value_dim]
 * (the combine einsum, attention.py:522-523). */
int icap_mha_combine_f32(const float* probs, const float* v, float* out, int Bf, int T, int S,
                         int heads, int value_dim, icap_stream_t stream);

typedef struct icap_tconfig {
  int32_t B, W, L, F, D;        /* images, beam width, locations, CNN feature dim, encoder dim */
  int32_t hidden, heads, key_dim, filter, layers; /* as written: heads = hidden_size, key_dim =
                                  * params["num_heads"] (transform_decoder.py:65-72 binds
                                  * positionally to attention.py:259-262) */
  int32_t V, T;
  int32_t start_id, eos_id;
  int32_t tile_mode;            /* ICAP_TILE_* */
  int32_t mask_mode;            /* ICAP_TMASK_* */
  int32_t kv_from_normed;       /* 0 as written: self-attention keys / values from the raw
                                 * layer input, only the query from LayerNorm(x)
                                 * (transform_decoder.py:35-38,136-140) */
  int32_t precision;            /* ICAP_PRECISION_FP32 */
  double length_penalty_weight;
} icap_tconfig;

typedef struct icap_tlayer { /* device pointers; EinsumDense kernels flattened to 2-D */
  const float* self_ln_gamma;  const float* self_ln_beta;           /* [hidden] */
  const float* self_q_kernel;  const float* self_q_bias;            /* [hidden,HD], [HD], HD = heads*key_dim */
  const float* self_k_kernel;  const float* self_k_bias;
  const float* self_v_kernel;  const float* self_v_bias;
  const float* self_o_kernel;  const float* self_o_bias;            /* [HD,hidden], [hidden] */
  const float* cross_ln_gamma; const float* cross_ln_beta;
  const float* cross_q_kernel; const float* cross_q_bias;           /* [hidden,HD] */
  const float* cross_k_kernel; const float* cross_k_bias;           /* [D,HD] */
  const float* cross_v_kernel; const float* cross_v_bias;
  const float* cross_o_kernel; const float* cross_o_bias;
  const float* ffn_ln_gamma;   const float* ffn_ln_beta;
  const float* ffn_filter_kernel; const float* ffn_filter_bias;     /* [hidden,filter], [filter] */
  const float* ffn_output_kernel; const float* ffn_output_bias;     /* [filter,hidden], [hidden] */
} icap_tlayer;

typedef struct icap_tweights {
  const float* enc_kernel; const float* enc_bias;       /* [F,D], [D] */
  const float* embedding;                               /* [V,hidden] */
  const float* out_ln_gamma; const float* out_ln_beta;  /* [hidden] */
  const float* fc2_kernel; const float* fc2_bias;       /* [hidden,V], [V] */
  const icap_tlayer* layers;                            /* host array of cfg.layers entries (copied) */
} icap_tweights;

typedef struct icap_tengine icap_tengine;

int icap_tengine_create(const icap_tconfig* config, const icap_tweights* weights,
                        icap_tengine** out);
int icap_tengine_destroy(icap_tengine* engine);
/* Re-packs what the engine derives from the weights (key|value kernels side by side, the
 * position table). */
int icap_tengine_refresh_weights(icap_tengine* engine, icap_stream_t stream);
size_t icap_tengine_workspace_bytes(const icap_tengine* engine);
/* img[B,L,F] -> tokens[B,T] (+ best_score[B], may be NULL): encoder -> tile -> prefix [0] ->
 * T x (TransformDecoder.call on [prefix | token], beam step) -> back-track. */
int icap_tdecode(icap_tengine* engine, const float* img, int32_t* tokens, float* best_score,
                 icap_stream_t stream);
/* Per-step beam records of the last icap_tdecode, as icap_engine_trace. */
int icap_tengine_trace(const icap_tengine* engine, int32_t* ids, int32_t* parents, float* scores,
                       float* log_probs, icap_stream_t stream);
/* The stack's normalised output of the newest position, [B*W,hidden] (the fourth value
 * TransformDecoder.decode returns, transform_decoder.py:231) of the last step. */
int icap_tengine_last_output(const icap_tengine* engine, float* out, icap_stream_t stream);
#define ICAP_TK_ENCODE 0     /* encoder + encoder-side keys / values of every layer */
#define ICAP_TK_EMBED 1
#define ICAP_TK_LAYERNORM 2
#define ICAP_TK_PROJ 3       /* q / k|v / output projections */
#define ICAP_TK_SELF_ATT 4
#define ICAP_TK_CROSS_ATT 5
#define ICAP_TK_FFN 6
#define ICAP_TK_VOCAB 7
#define ICAP_TK_BEAM 8
#define ICAP_TK_BACKTRACK 9
#define ICAP_TK_COUNT 10
/* As icap_engine_set_timing: creates the event pool (ICAP_TTIMING_POOL_DECODES bracketed
 * decodes between two reads), so that icap_tdecode creates no event. */
#define ICAP_TTIMING_POOL_DECODES 24
int icap_tengine_set_timing(icap_tengine* engine, unsigned class_mask);
int icap_tengine_kernel_times(icap_tengine* engine, float* ms /*[ICAP_TK_COUNT]*/,
                              int32_t* launches /*[ICAP_TK_COUNT]*/);

#ifdef __cplusplus
}
#endif
#endif /* ICAP_H_ */
