/* libasr_mi355x.so - C ABI of the MI355X-native (gfx950) ASR training hot path.
 *
 * Drop-in scope: the arithmetic that cosmoquester/speech-recognition's `run.train` delegates to
 * TensorFlow ops.  The reference has no FFI of its own (it is pure
 * Python on TF), so every entry point cites the reference call site (file:line under
 * /root/reference/speech_recognition) whose TF op it replaces.
 *
 * Conventions
 *   - plain C, no HIP/torch types: device pointers are raw pointers, `stream` is a hipStream_t
 *     passed as void* (NULL = default stream).
 *   - the caller allocates and owns every buffer (including workspaces); the library keeps no
 *     pointer after a call returns and holds no mutable global state besides the last-error string.
 *   - every call is asynchronous on `stream`, safe to capture into a hipGraph (no allocation, no
 *     synchronisation, no host read-back inside).
 *   - return value: ASR_OK (0) or a negative asr_status; asr_last_error() describes the failure.
 *   - all matrices are row-major f32 unless stated; masks are uint8 (0/1); token ids int32.
 *   - randomness (dropout, SpecAugment) comes from the stateless hash RNG documented in
 *     oracle/rng.py: r(seed, stream, idx); `seed` is read from DEVICE memory (so a captured graph
 *     sees a fresh seed every replay), `stream_id` names the draw site.
 */
#ifndef ASR_MI355X_H
#define ASR_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum asr_status {
  ASR_OK = 0,
  ASR_ERR_ARG = -1,         /* null pointer / inconsistent flags        */
  ASR_ERR_SHAPE = -2,       /* shape the kernels do not support         */
  ASR_ERR_UNSUPPORTED = -3, /* e.g. unknown rnn_type (las.py:17)        */
  ASR_ERR_HIP = -4          /* HIP runtime / launch failure             */
} asr_status;

const char* asr_last_error(void);
int asr_version(void);
/* sizeof(struct <name>) as compiled into the library (-1 = unknown): lets a binding verify its mirror */
long asr_struct_size(const char* name);
/* Once per process, after loading: returns and clears the HIP runtime's sticky error left by earlier calls of this thread (a probe
 * made before the device was initialised leaves hipErrorNoDevice behind).  The entry points never clear it themselves. */
int asr_runtime_init(void);

/* ------------------------------------------------------------------------------------------
 * Front end: log-mel + SpecAugment + delta/delta-delta + zero padding, one fused kernel.
 * Replaces data.py:169-187 (tf.signal.stft, tf.abs, tf.square, linear_to_mel_weight_matrix,
 * tf.matmul, tf.math.log), data.py:282-301 (SpecAugment frequency/time masks),
 * data.py:319-324 (delta_accelerate) and the zero padding of run/train.py:189-197.
 * ------------------------------------------------------------------------------------------ */
typedef struct asr_logmel_cfg {
  int sample_rate, frame_length, frame_step, fft_length, num_mel_bins;
  float lower_edge_hertz, upper_edge_hertz, epsilon;
  int use_delta;                       /* 1: output [..., 3] (x, delta, delta-delta); 0: [..., 1] */
  int sa_enable, sa_F, sa_mF, sa_T, sa_mT; /* SpecAugment masks; the time warp (W) is asr_time_warp */
  float sa_p;
  int feature_type;                    /* data_config.py:77-101: 0 "log-mel-spectrogram" (data.py:145-189), 1 "spectrogram"
                                          (data.py:122-142: |STFT|, fft_length/2+1 features, the mel fields are unused),
                                          2 "mfcc" (data.py:192-241: DCT-II of the log-mel, first num_mfcc coefficients)  */
  int num_mfcc;                        /* feature_type 2 only, <= num_mel_bins                                      */
} asr_logmel_cfg;

/* sizes (in elements) of the three constant tables the kernel reads */
int asr_logmel_table_sizes(const asr_logmel_cfg* cfg, long* n_twiddle_f32, long* n_melw_f32, long* n_melrange_i32);
/* fills HOST buffers (computed in double, rounded to f32); the caller uploads them once */
int asr_logmel_build_tables(const asr_logmel_cfg* cfg, float* twiddle, float* melw, int32_t* melrange);
/* audio [B, n_max] f32 (device), n_samples [B] int32 (device) -> out [B, T_out, v, C] f32 with v = num_mel_bins,
 * fft_length/2+1 or num_mfcc by feature_type (SpecAugment masks act on these v features, data.py:282-301).
 * Frames t >= frames(n_samples[b]) are written as exact 0.0.  seed may be NULL when !sa_enable. */
int asr_logmel_features(const asr_logmel_cfg* cfg, const float* audio, const int32_t* n_samples, int B, int n_max,
                        const float* twiddle, const float* melw, const int32_t* melrange, const uint32_t* seed,
                        float* out, int T_out, void* stream);

/* The same SpecAugment and delta steps as stand-alone kernels on feature tensors, for batches that
 * arrive as stored features (run/train.py:70-74 --use-tfrecord) instead of raw audio.
 * asr_spec_augment (data.py:282-301): x [B, T, v, C] is masked IN PLACE with 0.0; n_frames [B] int32
 * (device, NULL = T for every clip) is the reference's num_time of each clip; the draws are those of
 * asr_logmel_features for the same seed, so both routes zero the same bands.  Only the sa_* fields
 * and num_mel_bins (= v) of cfg are read.
 * asr_delta_accelerate (data.py:310-328): x [B, T, v] -> out [B, T, v, 3] = (x, delta, delta-delta)
 * with x[-1] = 0; frames t >= n_frames[b] are written as exact 0.0 (padding stays padding). */
int asr_spec_augment(const asr_logmel_cfg* cfg, float* x, const int32_t* n_frames, int B, int T, int C,
                     const uint32_t* seed, void* stream);
int asr_delta_accelerate(const float* x, const int32_t* n_frames, int B, int T, int v, float* out, void* stream);
/* asr_time_warp (data.py:275-280, SpecAugment's time warping): per clip tfa.image.sparse_image_warp of the
 * [T_b, v, C] feature image with ONE control point moved along time, (src, v/2) -> (dst, v/2),
 * src = W + U{0 .. T_b-2W-1}, dst = src - W + U{0 .. 2W-1} (RNG stream 5, indices 2b and 2b+1),
 * num_boundary_points = 3 (12 zero-flow points on the image border), polyharmonic spline of order 2,
 * bilinear resampling.  x [B, T, v, C] -> out [B, T, v, C] (must not alias x); frames t >= n_frames[b]
 * (NULL = T) and clips with T_b <= 2W (where the reference's draw range is empty) are copied unchanged.
 * coef: [B, 32] floats of scratch (the spline of each clip). */
int asr_time_warp(const float* x, const int32_t* n_frames, int B, int T, int v, int C, int W, const uint32_t* seed,
                  float* coef, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense contraction (tf.matmul / Dense / the batched halves of LSTM, attention and vocab
 * projections: las.py:43-59,169,193,196-202,264; deepspeech2.py:177; and every weight/input
 * gradient of those).  C[z] (+)= alpha * op(A[z]) op(B[z]) (+ bias), exact f32 on the MFMA.
 * ------------------------------------------------------------------------------------------ */
typedef struct asr_gemm_desc {
  int trans_a, trans_b;        /* 0: A is [M,K] / B is [K,N];  1: A is [K,M] / B is [N,K]          */
  int M, N, K;
  int batch;                   /* z extent (>= 1)                                                   */
  int split_k;                 /* > 1: partition K over that many workgroups (atomic accumulation)  */
  long lda, ldb, ldc;          /* row strides of the stored matrices, in elements                   */
  long stride_a, stride_b, stride_c, stride_a_scale; /* per-z strides; stride_c == 0 with batch > 1 = split-K (atomic) */
  float alpha;
  int accumulate;              /* 0: C = ..., 1: C += ..., 2: atomicAdd                             */
  int relu;
  const float* bias;           /* [N] or NULL                                                       */
  const float* a_scale;        /* optional multiplier on stored A: a_scale[(row / a_rpg) * cols + col] */
  int a_rpg;
  const float* c_scale;        /* optional multiplier on C: c_scale[(row / c_rpg) * N + col]        */
  int c_rpg;
  int compute;                 /* 0: f32 operands on the f32 MFMA (exact products); 1: mixed precision (train.py:62-66
                                  --mixed-precision): operands rounded to bf16 (RNE) into the bf16 MFMA, f32
                                  accumulation, f32 storage and epilogue; 2: f32 operands, every product a * b evaluated
                                  as the nine bf16 pair products of the exact three-way splits a = a1 + a2 + a3,
                                  b = b1 + b2 + b3 on the bf16 MFMA (2^-32 relative per product: tighter than an f32
                                  FMA), f32 accumulation - the f32 arithmetic of the reference at ~2x the f32-MFMA
                                  rate; 3: the same without the three pairs of weight <= 2^-24 (<= 3 * 2^-24 |a b|) */
} asr_gemm_desc;
int asr_gemm_f32(const asr_gemm_desc* d, const float* A, const float* B, float* C, void* stream);
/* Process-wide choice of how the CONVOLUTIONS (asr_conv2d_*) evaluate their f32 products: 0 the f32 MFMA, 2 / 3 nine / six bf16 pair
 * products of exact three-way operand splits on the bf16 MFMA (asr_gemm_desc.compute has the definitions; a GEMM carries its own
 * choice per call).  Returns the previous mode (0, 2 or 3), or a negative asr_status for a bad argument.  Default: the environment variable
 * ASR_GEMM_F32 = mfma | split9 | split6 (split6 when unset). */
int asr_set_f32_product_mode(int mode);

/* ------------------------------------------------------------------------------------------
 * Conv2D, padding VALID, NHWC activations, HWIO kernel, linear (las.py:163-164 + 183-184,
 * deepspeech2.py:47-50 + 57-59) and its two gradients, as implicit GEMMs on the MFMA.
 * x [B,H,W,C], w [kh,kw,C,O], y [B,Ho,Wo,O] with Ho = (H-kh)/sh + 1, Wo = (W-kw)/sw + 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct asr_conv_desc {
  int B, H, W, C;   /* input  */
  int kh, kw, sh, sw, O;
} asr_conv_desc;
int asr_conv2d_out_dims(const asr_conv_desc* d, int* Ho, int* Wo);
/* y = conv(x, w) + bias, then optional Keras Dropout (las.py:183-184): y *= mult(stream, flat index) */
int asr_conv2d_fwd(const asr_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                   const uint32_t* drop_seed, uint32_t drop_stream, float drop_rate, void* stream);
/* dw += im2col(x)^T dy   (atomic accumulation: dw must hold zeros or a running sum) */
int asr_conv2d_bwd_filter(const asr_conv_desc* d, const float* x, const float* dy, float* dw, void* stream);
/* dx = full correlation of dy with w (overwrites dx) */
int asr_conv2d_bwd_data(const asr_conv_desc* d, const float* dy, const float* w, float* dx, void* stream);
/* The same forward pass / input gradient for kernels that slide with stride 1 along W over 32-channel multiples (deepspeech2.py:47-50, conv2 and
 * conv3): every input row is staged in LDS once per kernel row and shared by the kw taps (csrc/conv_halo.hip).  asr_conv2d_halo_workspace returns
 * the bytes of device scratch the call needs for the re-ordered kernel (which: 0 forward, 1 input gradient), or 0 when the geometry or the
 * product mode (asr_set_f32_product_mode 0) takes the general entry points above; the *_halo calls then return ASR_ERR_UNSUPPORTED.  Same
 * results as asr_conv2d_fwd / asr_conv2d_bwd_data up to the order of the f32 accumulation.  No dropout epilogue. */
long asr_conv2d_halo_workspace(const asr_conv_desc* d, int which);
/* The row-staged kernels run one long workgroup per compute unit, so a geometry whose workgroups leave the last round of 256 mostly empty keeps the
 * general kernels (workspace 0).  asr_conv2d_halo_force(1) lifts that gate (tests, tuning), (0) restores it, (-1) only reads; returns the old value. */
int asr_conv2d_halo_force(int on);
int asr_conv2d_fwd_halo(const asr_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, long ws_bytes, void* stream);
int asr_conv2d_bwd_data_halo(const asr_conv_desc* d, const float* dy, const float* w, float* dx, void* ws, long ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory-bound layer kernels
 * ------------------------------------------------------------------------------------------ */
int asr_fill_f32(float* p, long n, float value, void* stream);
/* Listener._audio_mask (las.py:205-217) / Convolution._audio_mask (deepspeech2.py:68-78):
 * out[b][j] = any(x[b, j*group : (j+1)*group, :] != 0.0); x [B,T,FC], out [B,Tout] uint8 */
int asr_frame_mask(const float* x, int B, int T, int FC, int group, int Tout, uint8_t* out, void* stream);
/* out[c] += sum_r A[r][c]  (bias gradients; atomic) */
int asr_colsum(const float* A, int M, int N, long lda, float* out, void* stream);
/* Mixed precision with bf16 operands IN MEMORY (the wide models, BASELINE configs[4]): C[M,N] (+)= alpha A[M,K] B[N,K]^T (+ bias) from bf16 images
 * of both operands, k-contiguous (desc->trans_a = 0, trans_b = 1; lda / ldb / K multiples of 8), f32 accumulation and C, the epilogue options
 * of asr_gemm_f32 except a_scale (fold it into the image).  asr_f32_to_bf16_image makes the images: dst[r][c] (transpose: dst[c][r]) =
 * bf16(src[r][c] * scale[r / rows_per_group][c]); source row r = batch r / rows_per_batch (stride batch_stride) x row r % rows_per_batch
 * (rows_per_batch 0 = one batch), so that the shifted [B, T-1, H] views of the recurrent-kernel gradient flatten into one product. */
int asr_gemm_bf16_nt(const asr_gemm_desc* d, const void* A16, const void* B16, float* C, void* stream);
int asr_debug_sweep_trace(unsigned long long* out, int n);   /* timing aid: stage stamps of one BPTT-sweep workgroup (ASR_SWEEP_DBG bit 128), 8 per step */
int asr_debug_decoder_trace(unsigned long long* out, int n); /* timing aid: stage stamps of two forward decoder-sweep workgroups (ASR_DECODER_SWEEP_TRACE=1), [2][128 steps][16] */
int asr_gemm_bf16_config(int cfg);   /* tile configuration of asr_gemm_bf16_nt (tuning and tests; -1 = query); returns the previous one */
int asr_f32_to_bf16_image(const float* src, long ld_src, int rows, int cols, int rows_per_batch, long batch_stride, const float* scale,
                          int rows_per_group, int transpose, void* dst, long ld_dst, int dst_rows_per_batch, int dst_shift, void* stream);
/* (dst_rows_per_batch, dst_shift; transposed image only, 0 0 = off: source row r lands in destination column
 *  (r / rows_per_batch) * dst_rows_per_batch + r % rows_per_batch + dst_shift - h[b, t] beside ds[b, t + 1] for the recurrent-kernel gradient,
 *  which then shares the transposed ds image of the input-kernel gradient; the columns left out must already be zero.) */
/* The one-column products of the hoisted attention (las.py:46-59: query_weight bias): out[c] += sum_r w[r] A[r][c] (d bq = K^T ds0),
 * y[r] = A[r][:] . x (s0 = K bq), C[r][c] += u[r] v[c] (the ds0 (x) bq term of dK) - memory-bound kernels instead of N = 1 / K = 1 GEMMs. */
int asr_colsum_weighted(const float* A, int M, int N, long lda, const float* w, float* out, void* stream);
int asr_rowdot(const float* A, int M, int K, long lda, const float* x, float* y, void* stream);
int asr_rank1_add(float* C, int M, int N, long ldc, const float* u, const float* v, void* stream);
/* BatchNormalization(axis=-1) (+ fused ReLU) over x [M, C] (las.py:170,193; deepspeech2.py:112,118).
 * training: biased batch statistics over all M rows, mean/rstd saved, moving stats updated with
 * `momentum`; inference: moving statistics.  stats_ws: 2*C doubles of scratch. */
int asr_bn_fwd(const float* x, int M, int C, long ld, const float* gamma, const float* beta, float eps, float momentum,
               int relu, int training, float* y, long ldy, float* mean_out, float* rstd_out, float* moving_mean,
               float* moving_var, double* stats_ws, void* stream);
/* dx overwritten; dgamma/dbeta accumulated (+=).  y is the forward output (needed when relu). */
int asr_bn_bwd(const float* x, const float* y, const float* dy, int M, int C, long ld, long ldy, long lddy,
               const float* mean, const float* rstd, const float* gamma, int relu, float* dx, long lddx, float* dgamma,
               float* dbeta, double* sums_ws, void* stream);

/* Dropout sites addressed per row.  Rows are (batch b, step i) pairs: period > 0 means batch-major
 * rows (r = b*period + i), period < 0 step-major rows (r = i*|period| + b).  Element (r, k) uses RNG
 * stream stream0 + stream_step * i and index b * idx_ld + idx_off + k; rate <= 0 disables the site. */
typedef struct asr_rowdrop {
  uint32_t stream0, stream_step;
  int period;
  long idx_ld;
  int idx_off;
  float rate;
} asr_rowdrop;
int asr_dropout_rows(const float* x, long ldx, float* y, long ldy, int R, int K, const uint32_t* seed, uint32_t stream0,
                     uint32_t stream_step, int period, long idx_ld, int idx_off, float rate, void* stream);
/* x[i] *= mult(stream_id, i) in place (gradient of a whole-tensor Keras Dropout) */
int asr_dropout_flat(float* x, long n, const uint32_t* seed, uint32_t stream_id, float rate, void* stream);
/* out[i] = mult(stream_id, i): the [B, D] table of a Keras RNN input dropout (las.py:94,102) */
int asr_dropout_table(float* out, long n, const uint32_t* seed, uint32_t stream_id, float rate, void* stream);
/* several such tables in one launch (all BiRNN layers of a step draw from the same seed) */
int asr_dropout_tables(int ntables, float* const* outs, const long* ns, const uint32_t* stream_ids, const float* rates, const uint32_t* seed,
                       void* stream);
/* Embedding (las.py:258,278): backward == 0: x[r,:] = E[tok[r],:] * drop1 * drop2;
 * backward != 0: dE[tok[r],:] += dx[r,:] * drop1 * drop2 (atomic).  R rows, Hd columns. */
int asr_embedding(int backward, float* E_or_dE, const int32_t* tok, int R, int Hd, int V, float* x_or_dx, long ld,
                  const uint32_t* seed, const asr_rowdrop* drop1, const asr_rowdrop* drop2, void* stream);
/* out[i*out_stride] = (tok[i] != pad)  (decoder step mask, las.py:276) */
int asr_token_mask(const int32_t* tok, long n, int pad, uint8_t* out, long out_stride, void* stream);
/* out[r] = argmax_c x[r][c], lowest index on ties (tf.argmax, las.py:372) */
int asr_argmax_rows(const float* x, long ld, int R, int N, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder attention, one step (las.py:43-59 as called from las.py:282) with the key projections
 * hoisted out of the step loop: Kq = (enc Wk + bk) Wq^T, s0 = (enc Wk + bk) bq.
 * ------------------------------------------------------------------------------------------ */
int asr_attn_step_fwd(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask, const float* enc, int B,
                      int T, int Hd, int D, float* e, float* p, float* ctx, long ldctx, void* stream);
int asr_attn_step_bwd(const float* dctx, long lddctx, const float* p, const float* Kq, const float* enc, int B, int T, int Hd,
                      int D, float* dp, float* ds, float* dh, long lddh, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked sparse softmax cross-entropy + accuracy (measure.py:4-21, 45-69), forward and gradient in
 * one pass.  logits [R, V] are overwritten with grad_scale * d(mean NLL)/d logits when write_grad.
 * stats (3 floats, zeroed by the caller): [0] loss (mean NLL over kept tokens), [1] #correct, [2] #kept.
 * ------------------------------------------------------------------------------------------ */
int asr_softmax_xent(float* logits, long ld, const int32_t* labels, int R, int V, int ignore_index, float* stats, int write_grad,
                     float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (run/train.py:159-168, Keras semantics) fused with LRScheduler (utils.py:11-35) over one flat
 * parameter buffer.  `state` (device int32[4]): [0] iterations, [1] dropout seed.
 * ------------------------------------------------------------------------------------------ */
typedef struct asr_lr_schedule {
  float increasing_delta, decreasing_delta, max_learning_rate, min_learning_rate;
  int warmup_steps, offset_steps;
} asr_lr_schedule;
int asr_lr_schedule_init(asr_lr_schedule* s, long total_steps, double max_learning_rate, double min_learning_rate,
                         double warmup_rate, long warmup_steps, long offset_steps);
int asr_adam_step(float* params, const float* grads, float* m, float* v, long n, const int32_t* state,
                  const asr_lr_schedule* lr, float beta1, float beta2, float eps, float grad_scale, const float* skip_flag,
                  void* stream);
/* skip_flag (optional device float): when non-zero the step is invalid (a hand-off of a one-launch sweep timed out): Adam leaves
 * parameters and moments untouched, asr_advance_state sets the sticky error word state[2] instead of counting the step. */
int asr_advance_state(int32_t* state, const float* skip_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * CTC (measure.py:24-42 CTCLoss on tf.nn.ctc_loss): labels [B, L] int32 zero(pad)-padded dense rows,
 * label_length = count(label != pad), logit_length = T for every row, blank_index as configured
 * (deepspeech.yml: 14).  logits [B*T, V] batch-major (row stride ld) are overwritten with
 * grad_scale * d(mean_b nll_b / len_b) / d logits when write_grad.  stats[0] += loss (zero it first);
 * per_sample [B] receives nll_b / len_b.  ws: asr_ctc_workspace_floats(B, T, L) floats of scratch.
 * ------------------------------------------------------------------------------------------ */
long asr_ctc_workspace_floats(int B, int T, int L);
int asr_ctc_loss(float* logits, long ld, const int32_t* labels, int B, int T, int V, int L, int blank, int pad, float* ws,
                 float* per_sample, float* stats, int write_grad, float grad_scale, void* stream);
/* out[r, :] = x[r, :] * (mask[r] != 0)   (deepspeech2.py:176 `* mask[:, :, None]` and its gradient) */
int asr_mask_rows(const float* x, long ldx, const uint8_t* mask, int R, int C, float* out, long ldo, void* stream);


/* ------------------------------------------------------------------------------------------
 * Recurrent layers: Keras LSTM / GRU(reset_after=True) / SimpleRNN with K.rnn mask semantics
 * (las.py:10-17 get_rnn_cls, las.py:62-126 BiRNN, las.py:259-262 + 285-288 decoder cells,
 * deepspeech2.py:109-119).  rnn_type: 0 = "lstm", 1 = "gru", 2 = "rnn".
 * One kernel launch per time step; a BiRNN's two directions share each launch.
 * ------------------------------------------------------------------------------------------ */
#define ASR_RNN_LSTM 0
#define ASR_RNN_GRU 1
#define ASR_RNN_RNN 2
#define ASR_RNN_MAXSEG 3

typedef struct asr_rnn_geom {
  int Q;                       /* unit groups of 4: ceil(H / 4)                                   */
  int KSt;                     /* packed 16-wide K blocks: sum over segments of ceil(K_i / 16)    */
  int ks0[ASR_RNN_MAXSEG];     /* first block of each segment                                     */
  long wp_floats;              /* size of the packed forward weight image                         */
} asr_rnn_geom;
int asr_rnn_geometry(int rnn_type, int H, int nseg, const int* K, asr_rnn_geom* g);

/* Pack the weights multiplying the concatenated cell input [seg0 | seg1 | ...] (each W[i] is
 * [K_i, G*H] Keras layout, row stride ldw[i]; is_rec marks the recurrent segment) into MFMA
 * fragment order.  Must be re-run whenever the weights change (once per optimizer step). */
int asr_rnn_pack(int rnn_type, int H, int nseg, const float* const* W, const long* ldw, const int* K, const int* is_rec,
                 float* Wp, void* stream);
/* The same for every cell of a model in one launch (the images of a LAS / DeepSpeech2 are refreshed together, once per step). */
typedef struct asr_rnn_pack_desc {
  int rnn_type, H, nseg;
  const float* W[ASR_RNN_MAXSEG]; long ldw[ASR_RNN_MAXSEG]; int K[ASR_RNN_MAXSEG]; int is_rec[ASR_RNN_MAXSEG];
  float* Wp;
} asr_rnn_pack_desc;
int asr_rnn_pack_many(int ncells, const asr_rnn_pack_desc* cells, void* stream);

/* One cell step for one direction; row b of every matrix is at ptr + b * ld. */
typedef struct asr_rnn_step_fwd {
  int nseg, KSt;
  const float* Wp;
  const void* Wp16;                         /* optional bf16 image of Wp (asr_f32_to_bf16), same order: mixed precision  */
  const float* seg_x[ASR_RNN_MAXSEG];       /* inputs multiplying the packed weights              */
  long seg_ld[ASR_RNN_MAXSEG];
  int seg_K[ASR_RNN_MAXSEG], seg_ks0[ASR_RNN_MAXSEG];
  float seg_drop_rate[ASR_RNN_MAXSEG];      /* > 0: inverted dropout on that input, element index  */
  uint32_t seg_drop_stream[ASR_RNN_MAXSEG]; /*      = b * seg_drop_ld + seg_drop_off + k           */
  long seg_drop_ld[ASR_RNN_MAXSEG];
  int seg_drop_off[ASR_RNN_MAXSEG];
  const float* pre; long pre_ld;            /* pre-computed input projection [B, G*H] or NULL     */
  const float* bias;                        /* [G*H] added to pre (NULL if pre already has it)     */
  const float* bias_rec;                    /* GRU recurrent bias [3H] or NULL                     */
  const float* h_prev; long h_prev_ld;
  const float* c_prev; long c_prev_ld;      /* LSTM                                                */
  const float* y_prev; long y_prev_ld;      /* previously emitted output, NULL = zeros             */
  const uint8_t* mask; long mask_ld;        /* mask[b * mask_ld], NULL = all valid                 */
  float* h_out; long h_out_ld;
  float* c_out; long c_out_ld;
  float* y_out; long y_out_ld;
  float* saved; long saved_ld;              /* activations for backward [B, NS*H], NS = 4,4,1      */
} asr_rnn_step_fwd;
int asr_rnn_cell_fwd(int rnn_type, int B, int H, int ndir, const asr_rnn_step_fwd* steps, const uint32_t* seed,
                     void* stream);
/* dst[i] = bf16(src[i]), round to nearest even: the bf16 weight images above (train.py:62-66 --mixed-precision:
 * with them the wide step kernels multiply bf16 weights by bf16-rounded states on the bf16 MFMA, f32 accumulation) */
int asr_f32_to_bf16(const float* src, void* dst, long n, void* stream);
/* Transposed bf16 image with TIME-MAJOR columns - the layout asr_rnn_sweep_wide_bwd writes ds16T in: src is [nbatch][rows][cols] f32
 * (batch stride batch_stride, row stride ld_src), dst[c][t * nbatch + b + dst_shift] = bf16(src[b][t][c] * scale[b][c]); scale
 * (optional) is the Keras RNN input-dropout table [nbatch][cols].  dst_shift = nbatch pairs h[b, t - 1] with ds[b, t]. */
int asr_f32_to_bf16_image_tb(const float* src, long ld_src, int nbatch, int rows, int cols, long batch_stride, const float* scale, void* dst,
                             long ld_dst, long dst_shift, void* stream);
/* dst[i] = float(src[i]) for a bf16 array (gradient buckets that were all-reduced as bf16) */
int asr_bf16_to_f32(const void* src, float* dst, long n, void* stream);
/* Diagnostic: `blocks` workgroups of `threads` threads that do nothing for `microseconds` (<= 2 s, bounded by the real-time
 * counter): stands in for a foreign kernel (an RCCL channel) holding compute units while the one-launch sweeps run. */
int asr_debug_occupy(int blocks, int threads, int microseconds, void* stream);
/* Diagnostic: `blocks` workgroups of 256 threads that copy the first half of buf[0, bytes) onto the second half, again and again,
 * for `microseconds` (<= 2 s): a memory-streaming co-tenant (the weight-gradient products and RCCL reductions that run beside the
 * BPTT sweeps under data parallelism, utils.py:142-153) - the regression condition of the two sweep races of DESIGN.md 4.2. */
int asr_debug_stream_memory(float* buf, long bytes, int blocks, int microseconds, void* stream);

/* Backward of one cell step.  The gradient handed from step to step is ds, the gradient wrt the gate
 * sums ([B, NS*H], written over the saved activations).  A source describes one consumer of this
 * cell's h (srcA, -> state gradient) or emitted output (srcB, -> output gradient, optionally through
 * that consumer's input dropout): dh[b, j] = sum_c D[b, d_col0 + c] * W[j * ldw + w_col0 + c] over the
 * column segments (W rows = Keras kernel rows: recurrent kernel rows are units, input kernel rows are
 * input features).  out != NULL selects the linear mode: out = sum_A + addA + direct + sum_B + addB. */
typedef struct asr_rnn_back_src {
  const float* D; long ldd;                 /* consumer's ds rows (NULL: source unused)             */
  const float* W; long ldw;
  const void* W16;                          /* optional bf16 image of W (same layout and ldw): mixed precision */
  int nseg; int d_col0[2], w_col0[2], len[2];
  float drop_rate; uint32_t drop_stream; long drop_ld; int drop_off; /* srcB only                  */
} asr_rnn_back_src;
typedef struct asr_rnn_step_bwd {
  int n_units;                              /* H (cell mode) / output width (linear mode)          */
  asr_rnn_back_src srcA, srcB;
  const float* addA; long addA_ld;          /* dense addend of the state gradient                  */
  const float* addB; long addB_ld;          /* dense addend of the output gradient                 */
  float* direct; long direct_ld;            /* [B,H] carried part of dh (masked rows, GRU z*dh): read then rewritten, or NULL */
  float* out; long out_ld;                  /* linear mode output                                  */
  float* dc; long dc_ld;                    /* LSTM cell-state gradient, updated in place          */
  float* dy_carry; long dy_carry_ld;        /* pending output gradient across masked steps or NULL */
  const uint8_t* mask; long mask_ld;
  const float* saved; long saved_ld;
  const float* h_prev; long h_prev_ld;
  const float* c_prev; long c_prev_ld;
  const float* c_out; long c_out_ld;
  float* dslots; long dslots_ld;            /* out: ds [B, NS*H] (may alias saved)                 */
} asr_rnn_step_bwd;
int asr_rnn_cell_bwd(int rnn_type, int B, int ndir, const asr_rnn_step_bwd* steps, const uint32_t* seed, void* stream);


/* A whole (Bi)RNN layer over time.  Tensors are batch-major: pre [B,T,G*H] (input projection incl.
 * bias), hseq/cseq [B,T,H] (states after each step, time order), y [B,T,y_ld] with direction d
 * writing columns [y_col[d], y_col[d]+H) (so forward|backward concatenation and the re-reversal
 * of las.py:125 are free), saved [B,T,NS*H] (LSTM/RNN: may alias pre), mask [B,T] uint8 or NULL. */
typedef struct asr_rnn_seq {
  int rnn_type, B, T, H, ndir;
  int reverse[2];                           /* 1: go_backwards                                      */
  const float* pre[2];
  const float* Wp[2];                       /* packed recurrent kernels (asr_rnn_pack)              */
  const float* U[2]; long ldu[2];           /* recurrent kernels [H, G*H] in the Keras layout (backward) */
  const void* Wp16[2]; const void* U16[2];  /* optional bf16 images of Wp / U (mixed precision; used by the wide step kernels) */
  const float* bias_rec[2];                 /* GRU                                                   */
  const float* h0[2]; long h0_ld[2];        /* initial states (NULL = zeros)                         */
  const float* c0[2]; long c0_ld[2];
  const float* rec_mult[2];                 /* recurrent-dropout multipliers (constant over time, deepspeech2.py:95-107) or NULL: ONE [B,H] TABLE
                                               PER GATE, back to back ([4][B,H] LSTM i,f,c,o; [3][B,H] GRU z,r,h; [1][B,H] SimpleRNN) - tf.keras
                                               switches its LSTM / GRU cells to implementation 1 whenever recurrent_dropout != 0, which masks
                                               h_tm1 separately per gate and carries the GRU state unmasked.  Per-step kernels only - the
                                               one-launch sweeps reject it */
  const uint8_t* mask;
  float* hseq[2]; float* cseq[2];
  float* y; long y_ld; int y_col[2];
  float* saved[2];
  float* coef[2];                           /* one-launch sweeps only, or NULL: [B,T,H,CW] (CW = 8; SimpleRNN 4) the element-wise
                                               backward of every (row, step, unit) folded into coefficients by the forward sweep -
                                               LSTM {A, f, Co, m | Ci, Cf, Cg, 0}: dc' = dc + dh A, ds_i,f,g = dc' (Ci, Cf, Cg), ds_o = dh Co,
                                               dc_prev = dc' f; GRU {Cz, Cr, E, E r | z, m}: ds = dh (.) (...), carried dh = dh z; m = step mask.
                                               asr_rnn_sweep_fwd writes it (when non-NULL), asr_rnn_sweep_bwd reads it instead of
                                               saved / cseq / hseq / mask: two 16-byte loads per unit and step instead of seven scalar ones */
} asr_rnn_seq;
int asr_rnn_seq_fwd(const asr_rnn_seq* s, void* stream);
typedef struct asr_rnn_seq_grad {
  const float* dy; long dy_ld;              /* gradient wrt y, same layout as y                      */
  const float* dh_last[2]; long dh_last_ld[2]; /* gradient wrt the final h state or NULL            */
  float* dc[2];                             /* [B,H]: in = gradient wrt final c, out = wrt initial c */
  float* dy_carry[2];                       /* [B,H] scratch, zeroed by the caller (masked runs)     */
  float* direct[2];                         /* scratch [B,H] per direction (zeroed by the call)      */
  float* dh0[2]; long dh0_ld[2];            /* out: gradient wrt initial h or NULL                   */
  float* ds[2];                             /* asr_rnn_sweep_bwd only: out, the gate-sum gradients [B,T,NS*H], a buffer of its own */
  float* db[2]; float* db_rec[2];           /* asr_rnn_sweep_bwd only, optional: bias gradients accumulated (+=) by the sweep itself - db [G*H] the
                                               column sums of the input-side slots of ds (Keras bias, GRU: bias[0]), db_rec [3H] GRU's
                                               recurrent bias (bias[1]) - instead of a separate pass over ds (asr_rnn_sweep_wide_bwd: db too)  */
  void* ds16[2];                            /* asr_rnn_sweep_wide_bwd only, optional: the gate-sum gradients as a bf16 image [B T, 4H] ...   */
  void* ds16T[2]; long ds16T_ld;            /* ... and transposed, bf16 [4H][ds16T_ld], column t * B + b (time-major; ds16T_ld % 4 == 0, >= B T):
                                               the operands of the layer's dX / dW / dU products, written by the sweep itself; ds may then be NULL */
} asr_rnn_seq_grad;
/* After the call saved[d] holds the gate-sum gradients [B,T,NS*H] for the batched dW/dU/dX GEMMs (written in place over the
 * activations, one time step per launch).  asr_rnn_sweep_bwd writes them to g->ds[d] instead: its resident workgroups read the
 * saved activations of a step from several compute units at their own pace, so it must not overwrite them. */
int asr_rnn_seq_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* g, void* stream);
/* The same layer forward / backward-through-time as ONE launch each (rnn_sweep.hip / rnn_sweep_bwd.hip): workgroups stay
 * resident over all T steps, keep their slice of the recurrent kernel in registers and hand the recurrent quantity to each
 * other through global memory as self-validating 16-byte pieces (a sentinel NaN pattern marks "not written yet": no tags, no
 * flags, no fences).  Forward: every workgroup owns 4 (or 8) hidden units x all gates and gathers h_{t-1} of its 16 batch rows;
 * backward: a square of workgroups per (direction, batch tile) splits ds x U^T over both its axes, so a workgroup publishes one
 * [16 x KU] block of partial dh and gathers 16 x H floats.  Same contracts as asr_rnn_seq_fwd / asr_rnn_seq_bwd (g->direct and
 * g->dy_carry are unused; results equal to fp32 rounding, the forward bit for bit), except for how the element-wise operands
 * travel: asr_rnn_sweep_bwd reads s->coef (written by asr_rnn_sweep_fwd of the same layer when non-NULL) instead of saved /
 * cseq / hseq / mask, and writes the gate-sum gradients to g->ds.  Supported when H % 16 == 0, H <= 256,
 * T >= 2 and the grid fits the chip (asr_rnn_sweep[_bwd]_supported); every spin is bounded.  ws: asr_rnn_sweep[_bwd]_ws_floats() floats, re-armed by every call; the uint32 at
 * ws[ws_floats - 32] is non-zero after the call if a hand-off timed out (results invalid); err_flag: optional device float
 * that is set to 1.0f in that case and never cleared by the library (TrainStep's sticky error cell).
 * asr_rnn_sweep_set_spin_limit: polls before a hand-off gives up (default 2^20, about 1.2 s: a deadlock detector, not a latency bound; tests force time-outs with 0). */
long asr_rnn_sweep_ws_floats(int B, int H, int ndir);
int asr_rnn_sweep_supported(int rnn_type, int B, int T, int H, int ndir);
int asr_rnn_sweep_fwd(const asr_rnn_seq* s, float* ws, float* err_flag, void* stream);
long asr_rnn_sweep_bwd_ws_floats(int B, int H, int ndir);
int asr_rnn_sweep_bwd_supported(int rnn_type, int B, int T, int H, int ndir);
int asr_rnn_sweep_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* g, float* ws, float* err_flag, void* stream);
void asr_rnn_sweep_set_spin_limit(int polls);
int asr_rnn_sweep_spin_limit(void);
/* Every sweep keeps 32 diagnosis words behind its exchange buffer (the last 32 floats of an asr_rnn_sweep[_bwd|_wide]_ws_floats()
 * workspace; 288 floats from the end of a decoder-sweep workspace; csrc/sweep_common.h): word 0 the error word, word 4 the
 * workgroups that have started, word 5 the workgroups expected, words 8-15 what the first workgroup that gave up saw (its
 * abort word, block, XCD, arrivals at that moment, publish mode, wave, clock).  "arrivals < expected" in that record means a
 * workgroup was never resident (another tenant held its compute unit); equal means a hand-off was lost with everybody there.
 * asr_sweep_gate: one wave on `stream` that returns once the sweep owning `diag_words` has all its workgroups resident, or after
 * max_microseconds: launched in front of side-stream work that is to run BESIDE that sweep (the weight-gradient GEMMs of the
 * previous layer next to a BPTT sweep - the overlap of north_star's "all-reduce ... overlapped with the backward RNN sweep"
 * applied to the step's own off-critical-path products), so that it cannot take the sweep's compute units first.  No
 * reference counterpart (TensorFlow schedules its own streams). */
int asr_sweep_gate(const float* diag_words, int max_microseconds, void* stream);

/* Wide layers under mixed precision (las_large: H = 1024, B = 64): one launch per layer with the recurrent kernel resident as
 * bf16 MFMA operands (64 KB per workgroup, one workgroup of 8 hidden units per compute unit and direction), h_t exchanged as bf16
 * pieces that are the next step's A operands; same contract as asr_rnn_seq_fwd with the state operand of the recurrent product
 * rounded to bf16 (as the wide step kernels do with bf16 weights).  LSTM, 256 < H <= 1024, H % 128 == 0 (H / 128 in {4, 6, 8}),
 * B <= 64, H / 8 * ndir <= compute units.  ws: asr_rnn_sweep_wide_ws_floats() floats; error word / err_flag as asr_rnn_sweep_fwd. */
int asr_rnn_sweep_wide_supported(int rnn_type, int B, int T, int H, int ndir);
long asr_rnn_sweep_wide_ws_floats(int B, int H, int ndir);
int asr_rnn_sweep_wide_fwd(const asr_rnn_seq* s, float* ws, float* err_flag, void* stream);
/* The same layer's backward-through-time in one launch (models/las.py:90-106 differentiated; csrc/rnn_sweep_wide_bwd.hip): a 32 x 4 grid of
 * workgroups per direction with resident bf16 blocks of the recurrent kernel exchanges bf16 partial sums of dh.  Contract of
 * asr_rnn_sweep_bwd, except: reads s->saved / s->cseq (not coefficient packs), writes g->ds out of place (must not alias s->saved),
 * ignores g->db.  LSTM, H = 1024, B <= 64, 128 * ndir <= compute units.  ws: asr_rnn_sweep_wide_bwd_ws_floats() floats (48 MB exchange
 * for two directions + the 32 diagnosis words). */
int asr_rnn_sweep_wide_bwd_supported(int rnn_type, int B, int T, int H, int ndir);
long asr_rnn_sweep_wide_bwd_ws_floats(int B, int H, int ndir);
int asr_rnn_sweep_wide_bwd(const asr_rnn_seq* s, const asr_rnn_seq_grad* g, float* ws, float* err_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * One-launch forward sweep of the LAS decoder under teacher forcing (las.py:267-292 looped by las.py:368-377): all U steps of
 * {attention, decoder LSTM 0, decoder LSTM 1} in one kernel (decoder_sweep.hip) - replaces U x {asr_attn_step_fwd,
 * asr_rnn_cell_fwd x 2}.  256 resident workgroups keep the attention operands (Kq, enc: one (batch row, 1/8 of the frames)
 * slice each, in LDS) and the packed cell weights (registers) on chip; a step is four sentinel hand-offs (see rnn_sweep.hip).
 * Step-major tensors as in the per-step path: pre0 [U,B,4Hd] (embedding half of layer 0's input projection incl. bias),
 * tokmask [U,B]; outputs p [U,B,T2], ctx [U,B,D], hin/cin [U+1,B,Hd] (row i+1 = last layer's state after step i; row 0 in),
 * per layer y [U,B,Hd], saved gate activations [U,B,4Hd], and layer 0's state h0/c0 [U,B,Hd].  Dropout as in the per-step
 * kernels: layer j's input dropout uses stream drop_stream0 + drop_stream_step * i + 2 + j.
 * Supported (asr_decoder_sweep_supported): LSTM, 2 layers, B <= 64 (more than 32 rows run as two launches of <= 32), Hd % 16 == 0 <= 256,
 * D % 32 == 0 <= 512, T2 <= 512 (chunks of up to 64 frames: 32 resident in LDS, the rest streamed from L2 every step), a
 * device with >= 256 compute units.  ws: asr_decoder_sweep_ws_floats() floats; error word / err_flag as for asr_rnn_sweep_fwd.
 * ------------------------------------------------------------------------------------------ */
typedef struct asr_decoder_sweep {
  int B, U, T2, Hd, D;
  const float* Kq; const float* enc; const float* s0; const uint8_t* mask;   /* [B,T2,Hd], [B,T2,D], [B,T2] or NULL, [B,T2] */
  const float* h_init; const float* c_init;                                   /* [B,Hd] contiguous                           */
  const float* Wp0; int KSt0, ks0_ctx, ks0_h;     /* asr_rnn_pack image of layer 0: segments {context rows of the kernel, recurrent kernel} */
  const float* Wp1; int KSt1, ks1_x, ks1_h;       /* layer 1: segments {kernel, recurrent kernel}                                        */
  const float* pre0; const float* bias1;
  const uint8_t* tokmask;
  const uint32_t* seed; float drop_rate; uint32_t drop_stream0, drop_stream_step;
  float* p; float* ctx; float* hin; float* cin;
  float* y0; float* saved0; float* h0; float* c0;
  float* y1; float* saved1;
} asr_decoder_sweep;
int asr_decoder_sweep_supported(int rnn_type, int num_layers, int B, int U, int T2, int Hd, int D);
long asr_decoder_sweep_ws_floats(int Hd, int D);
int asr_decoder_sweep_fwd(const asr_decoder_sweep* s, float* ws, float* err_flag, void* stream);

/* The mirror for the backward pass (the decoder loop of las.py:368-377 differentiated; replaces U x {two cell-backward launches,
 * the context-gradient launch, the attention-backward launches} of asr_rnn_cell_bwd / asr_attn_step_bwd): from the forward
 * sweep's saved tensors and dy1 = gradient wrt the last layer's outputs [U*B, Hd] it writes
 *   ds0 / ds1 [U,B,4Hd]  gradients wrt the gate sums of the two layers (OUT OF PLACE: must not alias saved0 / saved1),
 *   de [U,B,T2]          gradient wrt the attention scores,   dctx [U,B,D]  gradient wrt the context (after layer 0's input dropout),
 *   dh_init / dc_init [B,Hd]  gradients wrt the decoder's initial state.
 * U1/W1/U0/W0: recurrent_kernel and kernel of layers 1 and 0, row-major [rows, 4Hd] (W0 has Hd embedding rows, then D context rows).
 * Pad-token rows (tokmask 0) carry the state gradients through unchanged.  Supported (asr_decoder_sweep_bwd_supported): LSTM,
 * 2 layers, B <= 64 (passes of 32 rows), T2 <= 512 (as the forward sweep), Hd in {16..128 step 16, 160..256 step 32}, D/4 a power of two <= 128 with D % (16 G) == 0 and D/G <= 64
 * (G = Hd/16 or Hd/32), T2 <= 256, >= 256 compute units.  ws: asr_decoder_sweep_bwd_ws_floats() floats; error words as forward. */
typedef struct asr_decoder_sweep_grad {
  int B, U, T2, Hd, D;
  const float* Kq; const float* enc;                 /* [B,T2,Hd], [B,T2,D]                               */
  const float* p; const float* ctx;                  /* forward outputs [U,B,T2], [U,B,D]                 */
  const float* saved0; const float* saved1;          /* gate activations [U,B,4Hd]                        */
  const float* cin; const float* c0;                 /* [U+1,B,Hd] (row i = last layer's c before step i), [U,B,Hd] */
  const uint8_t* tokmask;                            /* [U,B]                                             */
  const float* dy1; long dy1_ld;
  const float* U1; const float* W1; const float* U0; const float* W0;
  const uint32_t* seed; float drop_rate; uint32_t drop_stream0, drop_stream_step;
  float* ds0; float* ds1; float* de; float* dctx; float* dh_init; float* dc_init;
  float* de_sum;                                  /* optional [B,T2]: sum over the steps of de - the gradient wrt the loop-invariant score term
                                                     s0 = K bq of the hoisted attention (las.py:46-54), otherwise a [B,U,T2] x ones product */
} asr_decoder_sweep_grad;
int asr_decoder_sweep_bwd_supported(int rnn_type, int num_layers, int B, int U, int T2, int Hd, int D);
long asr_decoder_sweep_bwd_ws_floats(int Hd, int D);
int asr_decoder_sweep_bwd(const asr_decoder_sweep_grad* s, float* ws, float* err_flag, void* stream);

/* The same two steps with the two streamed operands given as bf16 images (asr_f32_to_bf16 of Kq and enc, made once per
 * training step): --mixed-precision.  The streams are what these kernels move, so the images halve their time; every
 * product and sum stays f32.  Needs Hd % 8 == 0, D % 8 == 0, 16-byte aligned h / dctx rows. */
int asr_attn_step_fwd_bf16(const float* h, long ldh, const void* Kq16, const float* s0, const uint8_t* mask, const void* enc16, int B,
                           int T, int Hd, int D, float* e, float* p, float* ctx, long ldctx, void* stream);
int asr_attn_step_bwd_bf16(const float* dctx, long lddctx, const float* p, const void* Kq16, const void* enc16, int B, int T, int Hd,
                           int D, float* dp, float* ds, float* dh, long lddh, int accumulate, void* stream);

/* The same two attention steps as ONE launch each (asr_attn_step_fwd / asr_attn_step_bwd are two): the T axis
 * is cut into chunks, every (batch row, chunk) workgroup publishes chunk-local softmax statistics and partial
 * sums, and the last one to arrive at an agent-scope ticket combines them.  Results equal to fp32 rounding
 * (the softmax is evaluated per chunk and rescaled).  scratch: asr_attn_fused_ws_floats() floats; tickets:
 * B uint32 words that are zero when first used and are only ever passed to these two functions (each call adds
 * the same constant per row, so they never need resetting - safe to replay from a hipGraph).  Supported when
 * asr_attn_fused_supported(): Hd % 4 == 0, D % 4 == 0, T <= 512, 16-byte aligned rows. */
long asr_attn_fused_ws_floats(int B, int Hd, int D);
int asr_attn_fused_supported(int T, int Hd, int D);
int asr_attn_fused_fwd(const float* h, long ldh, const float* Kq, const float* s0, const uint8_t* mask, const float* enc, int B,
                       int T, int Hd, int D, float* scratch, uint32_t* tickets, float* p, float* ctx, long ldctx, void* stream);
int asr_attn_fused_bwd(const float* dctx, long lddctx, const float* p, const float* Kq, const float* enc, int B, int T, int Hd,
                       int D, float* scratch, uint32_t* tickets, float* ds, float* dh, long lddh, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Greedy decoding (search.py).  All state stays on the device: a decode needs no host round trip per step.
 * asr_greedy_update - one step of LAS_Searcher.greedy_search (search.py:41-55) on logits [B, V]:
 *   (lp, tok) = top-1 of log_softmax (ties -> lowest index); log_ppl += lp unless ended; tok = pad if ended;
 *   ended |= tok == eos; seq_len = cur_len + 1 where tok == eos; next_tok = tok.  cur_len = tokens decoded so
 *   far including BOS (tf.shape(decoder_input)[1]).
 * asr_ctc_greedy - DeepSpeechSearcher.greedy_search (search.py:223-252) on logits [B*T, V]: the blank competes
 *   as the LAST class (ties go to non-blank), log-probabilities are log_softmax over the V classes, repeats are
 *   merged and blanks dropped ([TF-sem] tf.nn.ctc_greedy_decoder): tokens [B, T] zero padded, lengths [B],
 *   neg_sum_logits [B] = -sum_t max log-prob.  best [B*T] int32 / best_lp [B*T] f32 are scratch that also
 *   hold the per-frame alignment (class V = blank) and its log-probability on return.
 * ------------------------------------------------------------------------------------------ */
int asr_greedy_update(const float* logits, long ld, int B, int V, int cur_len, int eos, int pad, int32_t* next_tok,
                      uint8_t* ended, float* log_ppl, int32_t* seq_len, void* stream);
int asr_ctc_greedy(const float* logits, long ld, int B, int T, int V, int blank, int32_t* best, float* best_lp,
                   int32_t* tokens, int32_t* lengths, float* neg_sum_logits, void* stream);

/* ------------------------------------------------------------------------------------------
 * Beam search (search.py:83-209 LAS, search.py:254-285 DeepSpeech2).
 * asr_beam_topk - rows [R, V] of logits -> the k best classes of log_softmax per row in tf.math.top_k order
 *   (descending, ties -> lowest index): lp [R, k], tok [R, k]  (search.py:128-131).  k <= 32.
 * asr_beam_select - one step of the LAS beam update for B utterances x `beam` hypotheses (rows b*beam + j),
 *   reading the `in` state and writing the `out` state (they must not alias):
 *     candidate (j, m): log_prob = (ended[j] ? 0 : lp[j, m]) + log_ppl[j]            (search.py:136-138)
 *     length = first EOS + 1 of the extended row, penalty = ((1 + length) / (1 + beta)) ^ alpha in float64
 *     cast to float32, the `beam` best of log_prob * penalty in stable top_k order     (search.py:159-162)
 *     hist_out / log_ppl_out = gathered parents (+ the new token)                     (search.py:165-175)
 *   cur_len = tokens per row so far (BOS included).  At cur_len == 1 the k best classes of row b*beam become the
 *   beam (search.py:141-153; the caller keeps the beam rows of one utterance identical until then).
 *   hist [R, ld_hist] int32, log_ppl [R] f32, ended [R] u8 (row holds an EOS), slen [R] int32 (first EOS + 1
 *   where ended).  next_tok [R] receives the appended tokens, parent [R] the source rows, *final_len the row
 *   length after this step.  When every row had ended before the step (the reference has left its loop,
 *   search.py:122-125) the state is forwarded unchanged and *final_len is not touched.
 *   NOTE the reference does not re-order the decoder states by `parent` (search.py:169 returns them as they
 *   are); a caller that wants the hypotheses' own states gathers them with `parent`.
 * asr_ctc_log_softmax - search.py:268-272 on rows [R, V]: out [R, V + 1] = log_softmax of the row with the
 *   blank class appended last and its old slot masked with -1e9.
 * asr_ctc_beam_search - HOST function ([TF-sem] tf.nn.ctc_beam_search_decoder, merge_repeated = false, the op
 *   TensorFlow also runs on the CPU): log_probs host [B, T, C] with the blank as class C - 1, seq_len [B] or
 *   NULL (= T).  tokens [B, top_paths, T] zero padded, lengths [B, top_paths], log_prob [B, top_paths].
 *   `threads` host threads share the utterances.
 * ------------------------------------------------------------------------------------------ */
int asr_beam_topk(const float* logits, long ld, int R, int V, int k, float* lp, int32_t* tok, void* stream);
int asr_beam_select(const float* lp, const int32_t* tok, int B, int beam, int cur_len, int ld_hist, int eos, double alpha,
                    double beta, const int32_t* hist_in, const float* ppl_in, const uint8_t* ended_in, const int32_t* slen_in,
                    int32_t* hist_out, float* ppl_out, uint8_t* ended_out, int32_t* slen_out, int32_t* next_tok, int32_t* parent,
                    int32_t* final_len, void* stream);
int asr_ctc_log_softmax(const float* logits, long ld, long R, int V, int blank, float* out, void* stream);
int asr_ctc_beam_search(const float* log_probs, int B, int T, int C, const int32_t* seq_len, int beam_width, int top_paths,
                        int32_t* tokens, int32_t* lengths, float* log_prob, int threads);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (replaces the implicit all-reduce of tf.distribute.MirroredStrategy that
 * utils.py:142-153 sets up and model.fit drives, run/train.py:203-217): RCCL over xGMI, one process per GPU.
 *   asr_comm_available  - 1 when librccl.so.1 can be resolved in this process (it is bound at run time).
 *   asr_comm_unique_id  - rank 0 fills 128 bytes (ncclUniqueId); the host side hands them to every rank.
 *   asr_comm_init       - collective over all ranks: this process becomes `rank` of `nranks` (its current HIP device).
 *   asr_allreduce_bucket- bucket[0,n) <- sum over ranks, in place, asynchronously on `stream` (capturable into a hipGraph);
 *                         wire_bf16 = NULL: f32 on the wire; else n bf16 of device staging: the values cross the fabric as bf16
 *                         (--mixed-precision, run/train.py:62-66) and the sum lands back in the f32 bucket.
 *   Replicas scale their loss gradient by 1 / nranks, so SUM yields the replica mean ([TF-sem] SUM_OVER_BATCH_SIZE under
 *   MirroredStrategy).
 * ------------------------------------------------------------------------------------------ */
int asr_comm_available(void);
int asr_comm_unique_id(void* id128);
int asr_comm_init(const void* id128, int nranks, int rank, void** comm);
int asr_comm_destroy(void* comm);
int asr_allreduce_bucket(void* comm, float* bucket, long n, void* wire_bf16, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host-side input decoding (no GPU work, thread-safe, re-entrant): what tensorflow-io does for
 * data.py:94-117 and what TFRecord framing needs (data.py:75, run/make_tfrecord.py:47).
 * ------------------------------------------------------------------------------------------ */
typedef enum asr_audio_format { ASR_AUDIO_WAV = 0, ASR_AUDIO_FLAC = 1, ASR_AUDIO_PCM16 = 2 } asr_audio_format;
typedef struct asr_audio_info_t {
  int sample_rate;       /* 0 for headerless PCM                              */
  int channels;
  int bits_per_sample;
  long frames;           /* samples per channel                               */
} asr_audio_info_t;
/* file: the whole audio file in host memory */
int asr_audio_info(const uint8_t* file, long nbytes, int format, asr_audio_info_t* info);
/* Decodes to mono float32 as data.py:97-117 does: int16 / 32768 per channel, then the mean over
 * channels.  out: host buffer of `capacity` floats (>= info.frames); *n_out = samples written.
 * 16-bit sources only (the reference opens files as tf.int16); FLAC: every subframe type, both Rice
 * codings, all stereo decorrelation modes, header CRC-8 and frame CRC-16 verified. */
int asr_audio_decode(const uint8_t* file, long nbytes, int format, float* out, long capacity, long* n_out);
/* CRC-32C (Castagnoli) of n bytes continuing from `crc` (0 to start) */
uint32_t asr_crc32c(const void* data, long n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* ASR_MI355X_H */
