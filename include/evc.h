/*
 * evc.h - C ABI of libevc_hip.so: the MI355X (gfx950) kernels behind the
 * teacher/student frame-level aggregation + distillation hot path.
 *
 * The reference (shwetabhardwaj44/EfficientVideoClassification_Youtube8M) has
 * no native code and no FFI: its hot path is Python graph wiring over
 * TensorFlow-1.x primitive ops.  Each entry point below therefore replaces the
 * TensorFlow op(s) invoked at the cited reference call site (paths relative to
 * code_student_uniform/, "cs/").  The Python host
 * (efficientvideoclassification_youtube8m_amd/) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; the library never
 *    allocates, frees or synchronises.  Scratch is caller-provided.
 *  - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*).
 *  - return value: 0 = EVC_OK, negative = error; text via evc_last_error().
 *  - bf16 tensors are raw uint16 bit patterns; "ld" = leading dimension in
 *    ELEMENTS.  All bf16 GEMM operands are K-contiguous ("NT" form) with
 *    16-byte aligned rows (ld % 8 == 0, pointer % 16 == 0) and K % 64 == 0.
 *  - time-major activations: [T][M][width]; "M" = rows (videos x chunks).
 */
#ifndef EVC_H_
#define EVC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_OK 0
#define EVC_ERR_BAD_SHAPE (-1)
#define EVC_ERR_BAD_ALIGN (-2)
#define EVC_ERR_UNSUPPORTED_ARCH (-3)
#define EVC_ERR_HIP (-4)
#define EVC_ERR_BAD_ARG (-5)

#define EVC_VERSION 107   /* 107 (round 6, late): evc_lstm_level2_fwd_high (the two-layer L1 level of the "high" mode in T + 1 two-tile launches); 106 (round 6): evc_gemm_nt_sqnorm, evc_l2norm_chunk_int + the x_row_scale / x_col_const / b8_gap arguments of evc_lstm_layer_fwd_f16_fp8lo (integer-frame layer 0: the uint8 input exact), evc_lstm_layer_fwd_f16_fp8lo / evc_lstm_stack2_fwd_f16_fp8lo gained h_lo (low-order half of h corrected: 4H-byte h rows), evc_cast_f32_to_fp8_lohi, evc_lstm_adam_fused gained fp8_hi_tail; evc_absmax_partials, evc_cast_f32_to_f16_fp8x_dyn, evc_gemm_nt_f16_fp8_dyn (dynamic e4m3 range of the MoE head's input state); evc_l2norm_chunk_fwd accepts out1 == NULL (student-only graphs read the sub-sampled frames only), evc_clip_adam_small limited to 2^15 elements per tensor; 105 (round 5, second session): evc_cast_f32_to_f16_dither, evc_lstm_layer_fwd_f16_dith (time-dithered f16 weight images: an L1 layer of the "high" mode without stages for its weights' low-order halves), evc_gemm_tn2_rows (weight-gradient products that skip the dead rows of a row-planned level's time slabs); 104 (round 5): evc_lstm_level2_fwd, evc_ce_loss_ordered, evc_rep_loss_ordered, evc_gemm_tn2_slabs, evc_sum_slabs, evc_clip_adam_small; evc_moe_grad_update* accept p_bf16 == NULL (forward shadow not written), evc_lstm_stack2_bwd runs M <= 512 stacks on the skinny pair launches, evc_dbof_cluster_pool_fwd walks tiles (EVC_DBOF_WALK), EVC_DETERMINISTIC parsed as "set, not empty, not 0"; 103 (round 4): evc_sqnorm2_partials, evc_lstm_adam_fused, evc_gram_slabs, evc_moe_grad_norms, evc_moe_grad_update_apply, evc_adam2d_fused, evc_colsum_bf16_det, evc_sample_sequence_gather, evc_relu6_fwd/bwd, evc_framepool_mean_fwd/bwd, evc_stream_create_cu_mask / evc_stream_destroy; EVC_DETERMINISTIC=1 read by the library; 102: evc_lstm_layer_fwd_f16_fp8lo, evc_lstm_stack2_fwd_f16_fp8lo, evc_gemm_nt_f16_fp8, evc_cast_f32_to_fp8_lo, evc_cast_f32_to_f16_fp8x, aux_mode 5, evc_moe_grad_update_wide; 101 (round 3): evc_l2norm_chunk_fwd gained aux_mode; evc_lstm_layer_fwd_hp takes wide split operands; f16 / wide-split entries added */

typedef uint16_t evc_bf16;
typedef uint16_t evc_f16;   /* raw IEEE binary16 bits (the "high" precision forward operands of the L1 levels) */

int evc_version(void);
const char* evc_last_error(void);
/* 0 if device `dev` is gfx950, EVC_ERR_UNSUPPORTED_ARCH otherwise. */
int evc_check_device(int dev);

/* ---- a1 + a2: input preparation -------------------------------------------
 * tf.nn.l2_normalize(model_input_raw, 2)            cs/train.py:253-256
 * + every_n frame sub-sampling (transpose/gather)   cs/train.py:262-272
 * + bf16 cast + re-layout to the time-major chunked layout the LSTM consumes.
 *
 * x_raw [B, T, F] f32.  Teacher output out1: frame s -> chunk = s / (T/C1),
 * t = s % (T/C1), row m = chunk*B + b; out1[t][m][F] bf16.  If out2 != NULL the
 * frames s*every_n (s < S2 = T/every_n, integer division) are also written to
 * out2 with C2 chunks of S2/C2 frames (the student's view).  F % 4 == 0.
 * If x_u8 != NULL the input is the on-disk uint8 quantisation and is
 * dequantised first as cs/utils.py:22-25 (q*4/255 + 4/512 - 2) with rows
 * t >= num_frames[b] forced to zero (cs/readers.py:170-173); x_raw is ignored.
 * normalize = 0 skips the l2-normalisation (input already normalised by the
 * caller, as create_model() receives it at cs/train.py:282-284): cast + re-layout only.
 * Row plans (evc_sort_rows_by_len): with row_pos1 != NULL the teacher image is [T/C1][rows1][F] and chunk row
 * m goes to slot row_pos1[m]; rows whose slot is >= rows1 (length-0 rows) are neither read nor written.
 * row_pos2 / rows2: the same for the student image.
 * out1 == NULL (needs out2; out1_lo must be NULL too): a student-only graph (cs/train_finetune.py:243-318) - only the frames s*every_n
 * of x_raw are read, nothing else of the tensor is touched.
 */
int evc_l2norm_chunk_fwd(const float* x_raw, const uint8_t* x_u8, const int32_t* num_frames,
                         int B, int T, int F,
                         int C1, evc_bf16* out1,
                         int every_n, int C2, evc_bf16* out2, int normalize,
                         evc_bf16* out1_lo, evc_bf16* out2_lo /* second image of each view or NULL: see aux_f16 */,
                         int aux_mode /* 0: the split-bf16 low halves bf16(x - bf16(x)), rows of F;
                                         1..3: IEEE f16 images with rows of aux_mode*F = the first aux_mode of the segments
                                         [f16(x) | (x - f16(x))*64 | f16(x)/64] - against kernel rows [f16(W) | f16(W)/64 |
                                         (W - f16(W))*64] (evc_cast_f32_to_f16_wide) a plain f16 contraction over the wide rows
                                         adds the x_lo.W_hi (2) and x_hi.W_lo (3) corrections; the 64s keep both low-order
                                         factors in f16's normal range;
                                         4: wide split-bf16 images, rows of 2F = [bf16(x - bf16(x)) | bf16(x)] - the [lo | hi]
                                         operand of evc_lstm_layer_fwd_hp;
                                         5: rows of 4F bytes = [f16(x) (F halfwords) | e4m3(x 2^7) (F bytes) | e4m3((x - f16(x)) 2^18) (F bytes)] -
                                         the x rows of evc_lstm_layer_fwd_f16_fp8lo (F % 32 == 0; out*_lo then hold 2F halfwords per row) */,
                         const int32_t* row_pos1, int rows1, const int32_t* row_pos2, int rows2, void* stream);

/* Row plan of an LSTM stack: stable counting sort of its M rows by sequence length, longest first
 * (0 <= len <= max_len <= 63, M <= 65535).  pos[m] = slot of row m, inv[slot] = row, len_sorted[slot] =
 * len[inv[slot]].  In slot order the rows active at step t are the prefix [0, #{len > t}): the step kernels
 * below take that count per step (rows_per_step, a HOST array the caller derives from the same lengths) and
 * never touch the padding rows that tf.nn.dynamic_rnn masks (cs/frame_level_models.py:232-235: ~28% of the
 * chunk rows of a YouTube-8M batch have length 0). */
int evc_sort_rows_by_len(const int32_t* len, int M, int max_len, int32_t* pos, int32_t* inv, int32_t* len_sorted,
                         void* stream);

/* a2 (integer part, bit-exact): num_frames_student = int64(float64(n)/300*S)
 * cs/train.py:263-264; and the per-chunk L1 lengths / L2 length of
 * cs/frame_level_models.py:238-240,256 (teacher) and :308-310,327 (student).
 * len_l1 [C*B] int32 with row m = chunk*B + b; len_l2 [B] int32.
 * n_out [B] int64 receives the (possibly sub-sampled) frame count used.
 * subsampled != 0: the student input of cs/train.py:263-264 - the float64 formula is applied for every every_n,
 * including 1 (where it gives n-1 for n = 55, 79, 97, 110, ...); 0: the teacher input, n as it is. */
int evc_frame_counts(const int32_t* num_frames, int B, int every_n, int subsampled, int max_frames_before_sampling,
                     int num_chunks, int chunk_len, int64_t* n_out, int32_t* len_l1, int32_t* len_l2,
                     void* stream);

/* ---- generic bf16 MFMA GEMM (NT): C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N]) --
 * Replaces tf.matmul / slim.fully_connected's MatMul+BiasAdd
 * (cs/video_level_models.py:423-435, cs/frame_level_models.py:148,172,80-82).
 * out_bf16: 0 -> C is f32, 1 -> C is bf16.  accumulate: C += (f32 only).
 * K % 64 == 0.  bias may be NULL. */
int evc_gemm_nt(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, void* C, int64_t ldc,
                int M, int N, int K, const float* bias, int out_bf16, int accumulate, void* stream);
/* evc_gemm_nt (plain f32 output, no bias) + evc_grad_sqnorm in one pass (round 6): sums[0] += sum of (C + l2_coeff * P)^2 and sums[1] += sum of P^2 over the
 * product's elements (what evc_grad_sqnorm leaves), from the tiles' stores (P [M][N] f32 laid out as C; NULL with l2_coeff 0).  The MoE weight gradient that is materialised at 1024 rows
 * (cs/train.py:329-334 clip_by_norm of `MatMul(transpose_a=True)`, cs/video_level_models.py:423-435) needs no separate norm pass.  M > 512,
 * N % 256 == 0, K < 8192; the launch ADDS to sums (zeroed by the caller); part_ws: >= 16 ceil(M/128) ceil(N/128) floats of scratch - one slot per wave,
 * summed in index order by a finishing launch (no atomics: bit-identical runs). */
int evc_gemm_nt_sqnorm(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                       const float* P, float l2_coeff, float* sums, float* part_ws, int64_t part_ws_floats, void* stream);

/* "TN" product for the weight gradients: C[M,N] (+)= A^T . B with A [K][lda] and B [K][ldb] bf16
 * (the contraction index is the ROW of both operands - dz, x and h are all [T*M rows][width]),
 * f32 C.  Uses ds_read_b64_tr_b16 transpose reads, so no transposed copies are needed; long-K
 * products are split along K and joined with f32 atomics.  M % 8 == 0, N % 8 == 0, K % 32 == 0.
 * row_interleave_H > 0 (M == 4*H): A's columns are gate-interleaved (dz4); product row u*4+g is
 * stored at row g*H+u (TF gate order). */
int evc_gemm_tn(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* C, int64_t ldc,
                int M, int N, int K, int row_interleave_H, int accumulate, void* stream);

/* ---- a3/a4: one BasicLSTMCell layer over T steps with sequence lengths -----
 * tf.nn.dynamic_rnn(MultiRNNCell[BasicLSTMCell(H, forget_bias=1.0)], x,
 * sequence_length=len)  cs/frame_level_models.py:221-235,247-257,291-305,318-328
 *
 * x   [T][M][Kin] bf16 time-major layer input (for layer l>0: hbuf+M*H of layer l-1)
 * wT  [4H][Kin+H] bf16 = kernel^T, gate row blocks in TF order i,j,f,o
 * bias[4H] f32 (forget_bias 1.0 is added inside, as TF does at run time)
 * len [M] int32: state copied through for t >= len[m]
 * hoist: 0 = each step contracts [x_t, h_{t-1}] in one GEMM (K = Kin+H);
 *        1 = x.Wx for all T hoisted into one GEMM into zx_ws [T*M][4H] f32.
 * hbuf [(T+1)][M][H] bf16: slab 0 is zero-filled here, slab t+1 = h_t (0 where t >= len)
 * c_state/h_state: f32 final state columns, row stride ld_state (so they can
 *        point into the [M, 2*L*H] state tensor concat([c0,h0,c1,h1]));
 *        zero for len == 0.  c_state doubles as the running f32 cell state: it is read and rewritten in
 *        place by every active step (a row stops updating at t = len, which leaves the returned state);
 *        h_state is written once, at t = len-1.
 * gates [T][M][H] 8-byte records (bf16 post-activation i, j, f, o) and
 * c_all [(T+1)][M][H] bf16 (slab t+1 = cell state after step t, rounded; slab 0 is never read) are the
 *        history kept for the backward pass (12 bytes per element and step); both NULL = inference.
 * Epilogue layout: the MFMA is issued with the weight fragment first, so a lane holds 4
 * consecutive units of one row and every state / tape / h access is an 8-16 byte vector access
 * (c_state, h_state, bias 16-byte aligned; ld_state % 4 == 0).
 * Row plan (both NULL = none): rows_per_step [T] HOST int32, non-increasing: step t runs on rows
 *        [0, rows_per_step[t]) only (rows sorted by length, len = len_sorted); row_map [M] device int32 =
 *        the row of c_state / h_state that slot m writes (inv of evc_sort_rows_by_len).  Rows beyond
 *        rows_per_step[0] are not touched at all: the caller zeroes their final state.
 */
int evc_lstm_layer_fwd(const evc_bf16* x, const evc_bf16* wT, const float* bias, const int32_t* len,
                       int T, int M, int Kin, int H, int hoist, float* zx_ws,
                       evc_bf16* hbuf, float* c_state, float* h_state, int64_t ld_state,
                       void* gates, evc_bf16* c_all, const int32_t* row_map, const int32_t* rows_per_step, void* stream);

/* "High" precision layer for stacks with M ~ batch rows (the L2 level): split-bf16 operands - x = hi + lo to ~2^-16, products
 * lo.hi + hi.lo + hi.hi (the lo.lo term is below f32 noise) - contracted as K-EXTENSIONS of the plain bf16 loops (see
 * evc_gemm_nt_split): f32-operand accuracy at 3x the MFMA depth, same kernels, same epilogues.
 * x_lohi  [T][M][2Kin]  wide input image, rows [lo | hi] (evc_cast_f32_to_bf16_wide, lo_first = 1);
 * wx_hilo [4H] rows [hi(Kin) | lo(Kin)] of the kernel's x-part, row stride ldwx; wh_hilo [4H] rows [hi(H) | lo(H)] of its h-part;
 * zx_ws   [T][M][4H] f32: the x-projection of all T steps, hoisted into ONE split product;
 * hbuf    [(T+1)][M][H] bf16: h_t rounded (= the hi half), what the backward products read;
 * hbuf_lohi [(T+1)][M][2H]: the wide image [lo | hi] of every h_t - the next step's operand and the next layer's x_lohi.
 * Everything else as evc_lstm_layer_fwd (cs/frame_level_models.py:252-257: dynamic_rnn over the L1 final states); no row plan. */
int evc_lstm_layer_fwd_hp(const evc_bf16* x_lohi, const evc_bf16* wx_hilo, int64_t ldwx, const evc_bf16* wh_hilo, int64_t ldwh,
                          const float* bias, const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                          evc_bf16* hbuf, evc_bf16* hbuf_lohi, float* c_state, float* h_state, int64_t ld_state,
                          void* gates, evc_bf16* c_all, void* stream);
/* evc_lstm_layer_fwd on IEEE f16 operands: x [T][M][Kin], wT [4H][Kin+H] and hbuf [(T+1)][M][H] hold f16 and each step issues
 * ONE v_mfma_f32_16x16x32_f16 product per depth (the cost of the bf16 step; operand rounding 2^-12 instead of 2^-9).  Always the
 * fused [x_t | h] form.  An x-part that needs more than 2^-12 (layer 0 of the L1 level: scripts/precision_budget.py) is handed
 * over as a K-extension: x rows of nseg*F from evc_l2norm_chunk_fwd (aux_mode = nseg), kernel rows from
 * evc_cast_f32_to_f16_wide(nseg), Kin = nseg*F here.  hbuf_bf16 [(T+1)][M][H] receives the bf16 copy of every h_t: the operand of the backward products
 * (whose gradients need bf16's range).  Everything else as evc_lstm_layer_fwd (same dynamic_rnn semantics,
 * cs/frame_level_models.py:221-250). */
int evc_lstm_layer_fwd_f16(const evc_f16* x, int64_t ldx /* row stride of x (>= Kin): a wide h image of the layer below has 2H */,
                           const evc_f16* wT, const float* bias, const int32_t* len,
                           int T, int M, int Kin, int H, evc_f16* hbuf,
                           int h_wide /* 1: hbuf rows are [f16(h) | f16(h)/64] (2H) and the kernel's h-part is [f16(Wh) | (Wh - f16(Wh))*64]:
                                         the recurrent weights K-extended by their low-order halves (wT rows: Kin + 2H) */,
                           evc_bf16* hbuf_bf16,
                           float* c_state, float* h_state, int64_t ld_state, void* gates, evc_bf16* c_all,
                           const int32_t* row_map, const int32_t* rows_per_step, void* stream);
/* evc_lstm_layer_fwd_f16 with every weight of the layer exact to ~2^-15: per step
 *   z = [x16 | h16] . [W16x | W16h]^T (IEEE f16)  +  2^-(7 + w8_scale_exp) [x8 | h8] . [W8x | W8h]^T (OCP e4m3 on
 *   v_mfma_scale_f32_16x16x128_f8f6f4: per K element twice the MFMA rate of the 16-bit products)
 * with W8 = e4m3((W - f16(W)) 2^w8_scale_exp) from evc_cast_f32_to_fp8_lo, x8 = e4m3(x 2^7), h8 = e4m3(h_t 2^7).  The f16 rounding of a
 * WEIGHT is the same at every time step and in every chunk and enters the integrating cell states coherently (the term
 * that decides whether 1e-3 on the logits holds on trained weights); K-extending the weights by f16 low-order halves (h_wide = 1 above)
 * costs a second f16 product per depth, the e4m3 term half of that.  x rows (row stride ldx halfwords): kx16 halfwords at the row start -
 * the f16 operand against the first kx16 columns of wT16 - and kx8 e4m3 bytes at byte offset x8_off (layer 0: the rows of
 * evc_l2norm_chunk_fwd's aux_mode 5, kx16 = F, x8_off = 2F, kx8 = 2F = [e4m3(x 2^7) | e4m3((x - f16(x)) 2^18)] against wT8 columns
 * [lo(Wx) | e4m3(Wx 2^6)] - the second pair is the rounding of the INPUT, the one activation term f16 does not cover; upper
 * layers: the hbuf rows of the layer below, kx16 = H, x8_off = 2H, kx8 = H).  hbuf [(T+1)][M] rows of 3H bytes: [f16(h_t) (H halfwords) | e4m3(h_t 2^7) (H bytes)]; wT16 [4H][kx16 + H] f16,
 * wT8 [4H][kx8 + H] bytes.  H % 128 == 0, kx8 % 128 == 0, kx8 >= 384, kx16 % 64 == 0.  Everything else as evc_lstm_layer_fwd_f16
 * (cs/frame_level_models.py:221-250). */
/* h_lo = 1 (round 6, ABI 106): the low-order half of h is corrected as well - the uncorrected f16 rounding of the activations is what left 1e-3 .. 2e-3
 * on the logits of towers trained for 512 steps (profiles/r06_budget_worst_draw.txt).  hbuf rows are then 4H bytes [f16(h) | e4m3(h 2^7) |
 * e4m3((h - f16(h)) 2^18)], the h-part of wT8's rows is [lo(Wh) | hi(Wh)] (evc_cast_f32_to_fp8_lohi): wT8 [4H][kx8 + 2H]; a layer above reads those
 * rows as its x with kx16 = H, x8_off = 2H, kx8 = 2H against [lo(Wx) | hi(Wx)].  h_lo = 0: the rows / images described above. */
int evc_lstm_layer_fwd_f16_fp8lo(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16,
                                 const uint8_t* wT8, int w8_scale_exp, int h_lo, const float* bias, const int32_t* len,
                                 int T, int M, int H, evc_f16* hbuf, evc_bf16* hbuf_bf16, float* c_state, float* h_state,
                                 int64_t ld_state, void* gates, evc_bf16* c_all, const int32_t* row_map,
                                 const int32_t* rows_per_step, const float* x_row_scale, const float* x_col_const, int b8_gap, void* stream);
/* (x_row_scale / x_col_const != NULL, round 6: the INTEGER-FRAME form of the layer that reads the reader's uint8 frames - x rows from
 * evc_l2norm_chunk_int: kx16 = F exact integers c = 2q - 255, kx8 = F bytes e4m3(x_hat 2^7) at x8_off = 2F; x_row_scale [T][M] f32 = rs of every frame
 * row, x_col_const [4H] f32 = (255/256) sum_k f16(Wx[j][k]) (- 1 in the forget-gate block: the step kernels add forget_bias to their initial
 * accumulators).  Behind the x-part of the f16 stages the accumulators become acc * rs[row] + bias: z = rs (c + 255/256) . f16(Wx) + ... with the input
 * EXACT; its low-order e4m3 stages do not exist (b8_gap = the bytes of wT8's rows between the x-part's lo(Wx) block and the h-part: the hi(Wx) block
 * of an image laid out for the f32-input form).  NULL / 0: the forms above.) */
/* evc_lstm_layer_fwd_f16 on TIME-DITHERED weight images (DESIGN.md 7 "dither"): step t contracts
 *   z = [x16 | h16] . W16_t^T (IEEE f16; W16_t = the [4H][kx16 + H] image at wT16 + t * w16_step_stride halfwords, evc_cast_f32_to_f16_dither)
 *       + 2^-scale8_exp x8 . W8^T (OCP e4m3 stages behind the f16 ones; kx8 = 0: none)
 * - the same BasicLSTMCell step (cs/frame_level_models.py:221-250).  The f16 rounding error of a weight is the same at every step and is
 * integrated coherently by the cell state (what evc_lstm_layer_fwd_f16_fp8lo spends 8-9 e4m3 stages per weight block on); image t rounds
 * each element DOWN or UP so that over any run of steps the share of round-ups equals the element's position between its two f16
 * neighbours: the errors cancel over the steps of a chunk and the weights need no correction stages.  The e4m3 stages that remain are the
 * input frames' low-order half (layer 0: x rows of evc_l2norm_chunk_fwd aux_mode 5, kx16 = F, x8_off = 3F, kx8 = F against the
 * e4m3(Wx 2^6) block of evc_cast_f32_to_fp8_lo's rows: wT8 = that block's first byte, ldb8 = the row stride, scale8_exp = 18 + 6).
 * hbuf [(T+1)][M][H] PLAIN f16 rows (an upper layer's x: kx16 = H, kx8 = 0, wT8 NULL), hbuf_bf16 the bf16 copy; w16_step_stride = 0:
 * one image for every step.  kx16 % 64 == 0; with kx8 > 0: H % 128 == 0, kx8 % 128 == 0, kx8 >= 384, ldb8 % 16 == 0. */
int evc_lstm_layer_fwd_f16_dith(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16,
                                int64_t w16_step_stride, const uint8_t* wT8, int64_t ldb8, int scale8_exp, const float* bias,
                                const int32_t* len, int T, int M, int H, evc_f16* hbuf, evc_bf16* hbuf_bf16, float* c_state,
                                float* h_state, int64_t ld_state, void* gates, evc_bf16* c_all, const int32_t* row_map,
                                const int32_t* rows_per_step, void* stream);
/* The two-layer L1 level of the "high" mode in T + 1 launches (round 6; the reference's two stacked BasicLSTMCells under dynamic_rnn,
 * cs/frame_level_models.py:221-250): layer 0 as evc_lstm_layer_fwd_f16_fp8lo (x / wT16_0 / wT8_0 / w8_scale_exp / h_lo / bias0 / x_row_scale / x_col_const /
 * b8_gap: that entry's arguments), layer 1 as evc_lstm_layer_fwd_f16_dith with kx8 = 0 (wT16_1 = [4H][2H] f16 images, image t at + t * w16_step_stride
 * halfwords, contracting layer 0's rows [f16(h) | ...] and its own plain f16 rows).  One launch = layer 0's step s + layer 1's step s-1, every workgroup
 * walking both tiles (the bf16 form: evc_lstm_level2_fwd).  hbuf0 [(T+1)][M] rows of 4H bytes (h_lo) or 3H bytes, hbuf1 [(T+1)][M][H] plain f16,
 * hbuf*_bf16 [(T+1)][M][H] the bf16 copies of the backward pass.  Results: those of the two layer entries, bit for bit. */
int evc_lstm_level2_fwd_high(const evc_f16* x, int64_t ldx, int kx16, int64_t x8_off, int kx8, const evc_f16* wT16_0, const uint8_t* wT8_0,
                             int w8_scale_exp, int h_lo, const float* bias0, const float* x_row_scale, const float* x_col_const, int b8_gap,
                             const evc_f16* wT16_1, int64_t w16_step_stride, const float* bias1, const int32_t* len, int T, int M, int H,
                             evc_f16* hbuf0, evc_bf16* hbuf0_bf16, evc_f16* hbuf1, evc_bf16* hbuf1_bf16, float* c_state0, float* h_state0,
                             float* c_state1, float* h_state1, int64_t ld_state, void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1,
                             const int32_t* row_map, const int32_t* rows_per_step, void* stream);
/* evc_lstm_stack2_fwd (below) on IEEE f16 operands, the "high" precision form of the L2 level: layer 0 plain f16 (x-projection
 * hoisted into one f16 product), layer 1 with its kernel K-extended by the weights' low-order halves - wT1_wlo [4H] rows
 * [f16(Wx) | (Wx - f16(Wx))*64 | f16(Wh) | (Wh - f16(Wh))*64] (evc_cast_f32_to_f16_wlo) against activation rows [h | h/64] - because
 * the rounding of the upper layer's weights is the one error of this level that f16 does not cover (the same error at every step
 * into an integrating cell state; scripts/precision_budget.py).  Layer 0 may be K-extended too: its input (the L1 states) in
 * x_segments segments [f16(x) | (x - f16(x))*64 | f16(x)/64] against [Wx | Wx/64 | Wx_lo*64] (the hoisted product, K = x_segments*Kin)
 * and, with h0_ext, its recurrent weights [Wh | Wh_lo*64] against the whole wide h row - free inside the pair launches, whose time
 * the upper layer's K = 4H sets.  x [T][M][x_segments*Kin] f16; h0_wide / h1_wide
 * [(T+1)][M][2H] f16 = [f16(h_t) | f16(h_t)/64] per row, hbuf0 / hbuf1 [(T+1)][M][H] bf16 = the copies the backward products read.
 * Same wavefront, math and outputs as evc_lstm_stack2_fwd (cs/frame_level_models.py:252-257). */
int evc_lstm_stack2_fwd_f16(const evc_f16* x, int x_segments /* x rows: x_segments*Kin, evc_cast_f32_to_f16_segs */,
                            const evc_f16* wT0, int h0_ext /* wT0 rows: evc_cast_f32_to_f16_wide(x_segments, h0_ext) */, const float* bias0,
                            const evc_f16* wT1_wlo, const float* bias1,
                            const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                            evc_f16* h0_wide, evc_f16* h1_wide, evc_bf16* hbuf0, evc_bf16* hbuf1,
                            float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                            void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1, void* stream);
/* evc_lstm_stack2_fwd_f16 with the low-order halves of layer 0's recurrent weights and of all of layer 1's weights contracted as e4m3
 * operands behind the f16 stages of the same (pair) launches instead of f16 K-extensions (see evc_lstm_layer_fwd_f16_fp8lo): layer 1 walks
 * 32 f16 + 16 e4m3 stages instead of 64 f16 ones at H = 1024 - the M ~ batch steps are bound by their chain of dependent stages.
 * x [T][M][x_segments Kin] f16 (K-extended input of the hoisted product); wT0 [4H][x_segments Kin + H] f16 = [Wx segments | f16(Wh)],
 * wT0_8 [4H][H] = e4m3((Wh - f16(Wh)) 2^w8_scale_exp); wT1 [4H][2H] f16, wT1_8 [4H][2H]; h0_rows / h1_rows [(T+1)][M] rows of 3H bytes =
 * [f16(h) | e4m3(h 2^7)].  H % 128 == 0, H >= 512. */
/* h_lo = 1 (round 6): h rows of 4H bytes as in evc_lstm_layer_fwd_f16_fp8lo, wT0_8 [4H][2H] = [lo(Wh0) | hi(Wh0)], wT1_8 [4H][4H] = [lo(Wx1) | hi(Wx1) |
 * lo(Wh1) | hi(Wh1)]: both activation operands of layer 1 and the recurrent one of layer 0 are corrected (layer 1: 32 f16 + 32 e4m3 stages). */
int evc_lstm_stack2_fwd_f16_fp8lo(const evc_f16* x, int x_segments, const evc_f16* wT0, const uint8_t* wT0_8, const float* bias0,
                                  const evc_f16* wT1, const uint8_t* wT1_8, int w8_scale_exp, int h_lo, const float* bias1,
                                  const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                                  evc_f16* h0_rows, evc_f16* h1_rows, evc_bf16* hbuf0, evc_bf16* hbuf1,
                                  float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                                  void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1, void* stream);
/* A TWO-layer stack with M ~ batch rows (the L2 level: M = videos) in wavefront order: after layer 0's hoisted
 * x-projection (one GEMM into zx_ws [T][M][4H] f32), launch s runs layer 0's step s and layer 1's step s-1 side by
 * side (they are independent, and each is latency-bound at this size), so the chain of dependent launches is T+1
 * long instead of 2T.  Layer 1 reads layer 0's output slab as its x_t in the fused [x_t | h_{t-1}] form.  Same math
 * and outputs as two evc_lstm_layer_fwd calls (cs/frame_level_models.py:229-235,254-257 and :299-305,325-328, a 2-layer MultiRNNCell under
 * dynamic_rnn); no row plan (rows are videos).  Kin % 64 == 0, H % 64 == 0. */
int evc_lstm_stack2_fwd(const evc_bf16* x, const evc_bf16* wT0, const float* bias0, const evc_bf16* wT1, const float* bias1,
                        const int32_t* len, int T, int M, int Kin, int H, float* zx_ws,
                        evc_bf16* hbuf0, evc_bf16* hbuf1, float* c_state0, float* h_state0, float* c_state1,
                        float* h_state1, int64_t ld_state, void* gates0, evc_bf16* c_all0, void* gates1,
                        evc_bf16* c_all1, void* stream);


/* BPTT of the above (what tf.gradients builds inside
 * slim.learning.create_train_op, cs/train.py:329-334,413-418).
 * w_il [Kin+H][4H] bf16 = kernel in TF layout with the 4H axis GATE-INTERLEAVED
 *        (column u*4+g holds TF column g*H+u; evc_transpose_to_bf16(..., interleave_H=H))
 * dS_c/dS_h: f32 gradient wrt the final c/h state, row stride ld_dS
 * dh_above [T][M][H] bf16 or NULL: gradient arriving at h_t from the layer above (its dX = dz . Wx^T,
 *        evc_gemm_nt with bf16 output)
 * dc_ws [M][H] f32 scratch.
 * dz4  [T][M][H][4] bf16 out: gate pre-activation gradients, gate-interleaved (0 where
 *        t >= len); viewed as [T*M][4H] it is the A operand of dz . W^T with w_il.
 *        (evc_transpose_to_bf16(dz4, ..., interleave_H=-H) gives dz^T in TF gate order for the
 *        weight-gradient GEMM; writing it from this kernel's epilogue was measured 2.4x slower.)
 * db   [4H] f32 or NULL: bias gradient (TF gate order), ACCUMULATED into (caller zeroes it) with f32 atomics from
 *        the unrounded gate gradients of every active (row, step) - BiasAddGrad without another pass over dz.
 * Row plan (see evc_lstm_layer_fwd): rows_per_step [T] host counts, row_map = the row of dS_c / dS_h of each
 *        slot.  Every dz4 row is still written (zeros beyond the active prefix), because the
 *        weight-gradient products contract over all T*M rows.
 * dz_above [T][M][H][4] bf16 + w_above [H][4H] bf16 (both or neither; instead of dh_above): the gate gradients of the
 *        layer above and the x-rows of ITS kernel in the backward layout - the gradient arriving at h_t from above is
 *        then contracted inside this layer's step, [dz_{t+1} | dz_above_t] . [Wh ; Wx_above]^T with K = 8H, kept in the
 *        f32 accumulator (no hoisted dX product, no bf16 round trip).  Needs H % 128 == 0.
 */
int evc_lstm_layer_bwd(const evc_bf16* w_il, const int32_t* len, int T, int M, int Kin, int H,
                       const void* gates, const evc_bf16* c_all, const float* dS_c, const float* dS_h, int64_t ld_dS,
                       const evc_bf16* dh_above, float* dc_ws, evc_bf16* dz4, float* db,
                       const int32_t* row_map, const int32_t* rows_per_step, const evc_bf16* dz_above,
                       const evc_bf16* w_above, void* stream);

/* ---- a5 + a8 + a9 fused: MoE weight update without materialising the gradient ----------------
 * d(loss)/dW of slim.fully_connected (cs/video_level_models.py:423-435) is dlogits^T . x over the batch rows:
 * rank = batch.  For W [V][K] (stored transposed, the forward B operand) this entry point recomputes the
 * gradient tile from the factors twice: pass 1 accumulates sum (g + l2 W)^2 and sum W^2 into sums[0..1]
 * (fixed summation order), pass 2 applies tf.clip_by_norm(clip_norm) and the TF-Adam step in the GEMM
 * epilogue, writing W, m, v, the bf16 forward shadow p_bf16 [V][K] and the bf16 transposed shadow
 * pT_bf16 [K][ldT] (30 bytes of HBM traffic per parameter instead of 46 with a materialised gradient).
 *   dlogits [rows][ld_dlogits] bf16 (columns >= V zero), x [rows][ldx] bf16, rows % 32 == 0 (zero rows pad);
 *   under data parallelism rows = world x batch: the all-gathered factors replace the gradient all-reduce.
 *   partial_ws: 2 * ceil(V/128) * ceil(K/128) floats of scratch.  V % 4 == 0, K % 8 == 0.
 *   p_bf16 may be NULL (here and in the _wide / _phase / _apply forms): the forward shadow is then not written - a caller whose forward
 *   reads other operand images of W (the f16 + e4m3 images of _wide) saves 2 of the 34 bytes per parameter. */
int evc_moe_grad_update(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                        int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                        float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                        float beta1, float beta2, float eps, void* stream);
/* The same in two halves, for a weight matrix whose rows are SHARDED over data-parallel ranks (every pointer
 * already offset to the rank's row slab [v0, v0 + V): dlogits + v0, p/m/v/p_bf16 + v0*K, pT_bf16 + v0):
 *   phase 1: sums[0..1] += this slab's share of sum (g + l2 p)^2 and sum p^2   (then all-reduce sums: 8 bytes)
 *   phase 2: clip (by the norm of the WHOLE tensor in sums[0]) + Adam + shadows of the slab
 *   phase 0: both back to back (= evc_moe_grad_update).
 * The slab's updated bf16 rows are then all-gathered; f32 p / m / v stay sharded (ZeRO-1 for 2/3 of the
 * parameters: each rank streams 1/world of the Adam state, nothing contracts over more rows than it must). */
int evc_moe_grad_update_phase(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                              int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                              float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                              float beta1, float beta2, float eps, int phase, void* stream);
/* The clip norm of a MoE weight gradient without a pass over the weights (one process; replaces phase 1): with the batch-row factors
 * A = dlogits [R][V] and X [R][K] (g = A^T X), |g + l2 W|^2 = <A A^T, X X^T>_F + 2 l2 <A, logits - bias>_F + l2^2 |W|^2
 * (csrc/evc_moe_norms.hip; per-tensor clip_by_norm of slim.learning.create_train_op, cs/train.py:329-334, on the gradient that
 * includes the l2 regulariser's, cs/video_level_models.py:428,434).
 *   evc_gram_slabs: slabs[s][i][j] = sum over K slab s of A[i][k] A[j][k] (S slabs of R x R f32, stored plainly, summed later in
 *     slab order: run-to-run identical).  R % 32 == 0, Kc % 32 == 0 columns (zero padding allowed), S <= Kc / 32.
 *   evc_moe_grad_norms: sums[0] += the norm above, sums[1] += wsq[0]; gram_a / gram_x from evc_gram_slabs of dlogits / x; logits
 *     [B][V] f32 = the forward's X W^T (+ bias: pass it, or NULL for the gates); wsq[0] = |W|^2 of the current weights (kept by
 *     evc_moe_grad_update_apply); part_ws: 256 + 4 B floats of scratch; B <= 512.
 *   evc_moe_grad_update_apply: phase 2 alone with the optional operand images of evc_moe_grad_update_wide (any may be NULL) and
 *     wsq_out[0] = sum of the NEW weights squared, wsq_out[1] = 0 (both through partial_ws in a fixed order). */
int evc_gram_slabs(const evc_bf16* A, int64_t lda, int R, int Kc, int S, float* slabs, void* stream);
int evc_moe_grad_norms(const float* gram_a, int SA, const float* gram_x, int SX, int R, const evc_bf16* dlogits, int64_t ld_dlogits,
                       const float* logits, int64_t ld_logits, const float* bias, int B, int V, float l2_coeff, const float* wsq,
                       float* part_ws, float* sums, void* stream);
int evc_moe_grad_update_apply(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                              int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                              evc_bf16* p_wide_hilo, evc_f16* p_f16, uint8_t* p_fp8, int fp8_lo_exp, int fp8_hi_exp,
                              float l2_coeff, const float* sums, float* partial_ws, float clip_norm, float lr_t,
                              float beta1, float beta2, float eps, float* wsq_out, void* stream);
/* evc_moe_grad_update that ALSO writes, from the update's epilogue, the forward operand images of the new weights the non-bf16 precision
 * modes contract - instead of separate passes over the f32 weights after every update: p_wide_hilo [V][2K] = [bf16(W) | bf16(W - bf16(W))]
 * (the B operand of evc_gemm_nt_split: "split" mode) and / or p_f16 [V][K] = f16(W) with p_fp8 [V][2K] = [e4m3((W - f16(W)) 2^fp8_lo_exp) |
 * e4m3(W 2^fp8_hi_exp)] (the B16 / B8 operands of evc_gemm_nt_f16_fp8: "high" mode).  Either may be NULL, not both. */
int evc_moe_grad_update_wide(const evc_bf16* dlogits, int64_t ld_dlogits, const evc_bf16* x, int64_t ldx, int rows,
                             int V, int K, float* p, float* m, float* v, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT,
                             evc_bf16* p_wide_hilo, evc_f16* p_f16, uint8_t* p_fp8, int fp8_lo_exp, int fp8_hi_exp,
                             float l2_coeff, float* sums, float* partial_ws, float clip_norm, float lr_t,
                             float beta1, float beta2, float eps, void* stream);

/* ---- layout helpers --------------------------------------------------------
 * out[c][r] = in[r][c], r < R, c < C; out has ld_out >= Rpad columns and
 * columns [R, Rpad) are zero-filled (Rpad % 64 == 0 keeps GEMM K aligned).
 * in_f32: 1 -> `in` is f32 (cast to bf16 on the way), 0 -> bf16.
 * interleave_H > 0 (needs R == 4*H): input row g*H+u is written to output column
 * u*4+g instead of g*H+u - the gate-interleaved K order of evc_lstm_layer_bwd.
 * interleave_H = -H (needs C == 4*H): the INPUT columns are gate-interleaved (dz4 of
 * evc_lstm_layer_bwd); input column u*4+g is written to output row g*H+u (TF order). */
int evc_transpose_to_bf16(const void* in, int in_f32, int64_t ld_in, int R, int C,
                          evc_bf16* out, int64_t ld_out, int Rpad, int interleave_H, void* stream);
/* out_bf16[i] = bf16(in_f32[i]) for a [R, C] matrix (ld_in, ld_out). */
int evc_cast_f32_to_bf16(const float* in, int64_t ld_in, int R, int C, evc_bf16* out, int64_t ld_out, void* stream);
/* lo(w) = e4m3(clamp((w - f16(w)) * 2^lo_exp, +-448)) (OCP e4m3fn, round to nearest even) for a [R][C] f32 matrix: the low-order halves of a
 * weight matrix next to its f16 image, the wT8 operand of evc_lstm_layer_fwd_f16_fp8lo (lo_exp 17: |w| < 4 never clamps).  hi_cols > 0: out
 * rows are [lo(W[:, :hi_cols]) | hi(W[:, :hi_cols]) | lo(W[:, hi_cols:])] with hi(w) = e4m3(clamp(w * 2^hi_exp)) - the columns the INPUT's
 * low-order half e4m3((x - f16(x)) 2^18) (evc_l2norm_chunk_fwd aux_mode 5) is contracted against (hi_exp 6: the same 2^24 as 7 + 17). */
int evc_cast_f32_to_fp8_lo(const float* in, int64_t ld_in, int R, int C, int lo_exp, int hi_cols, int hi_exp, uint8_t* out, int64_t ld_out,
                           void* stream);
/* ... with the full-value image of EVERY column (round 6): out rows [lo(W[:, :hi_cols]) | hi(W[:, :hi_cols]) | lo(W[:, hi_cols:]) | hi(W[:, hi_cols:])],
 * 2C bytes (hi_cols = 0: [lo(W) | hi(W)]) - what activation rows [a8 | a_lo8 | b8 | b_lo8] are contracted against (h_lo = 1 entries). */
int evc_cast_f32_to_fp8_lohi(const float* in, int64_t ld_in, int R, int C, int lo_exp, int hi_cols, int hi_exp, uint8_t* out, int64_t ld_out,
                             void* stream);
/* out_f16[i] = f16(in_f32[i]), round to nearest even (the f16 weight shadows of evc_lstm_layer_fwd_f16). */
int evc_cast_f32_to_f16(const float* in, int64_t ld_in, int R, int C, evc_f16* out, int64_t ld_out, void* stream);
/* T time-dithered f16 images of n f32 values (n % 4 == 0, n < 2^32; image t at out + t * img_stride halfwords, same element order as in):
 * image t of element i = up if (uint32)(fmix32(i ^ seed * 0x9E3779B9) + t * 0x9E3779B9) < frac * 2^32 else dn, with dn <= in[i] <= up its
 * two f16 neighbours (equal when in[i] is an f16 value), frac = (in[i] - dn) / (up - dn), fmix32 = murmur3's finaliser: over any run of r
 * images an element is rounded up r * frac times +- a few (2.03 measured over every run inside 20 steps) (golden-ratio rotation of a per-element phase).  The operand of
 * evc_lstm_layer_fwd_f16_dith; restated bit for bit by oracle/lowprec.py::f16_dither_images.  row_len > 0: the values are rows of row_len elements and
 * only the columns from col0 on are dithered - the others hold their round-to-nearest f16 value in every image (a kernel whose input block keeps
 * its e4m3 correction while its recurrent block is dithered); 0, 0: every element. */
int evc_cast_f32_to_f16_dither(const float* in, int64_t n, int T, int64_t img_stride, uint32_t seed, evc_f16* out, int64_t row_len, int64_t col0,
                               void* stream);
/* f16 image of an LSTM kernel [R][Kin+H] (f32, row stride ld_in) for a K-extended x-part: out [R][nseg*Kin + H] =
 * [f16(Wx) | f16(Wx)/64 | (Wx - f16(Wx))*64 | f16(Wh) | (Wh - f16(Wh))*64] keeping the first nseg (1..3) x blocks and, with
 * h_ext = 1, the low-order block of the h-part (evc_lstm_layer_fwd_f16 with h_wide = 1). */
int evc_cast_f32_to_f16_wide(const float* in, int64_t ld_in, int R, int Kin, int H, int nseg, int h_ext, evc_f16* out, void* stream);
/* K-extended f16 image of an activation matrix [R][C] f32: out [R][nseg*C] = [f16(x) | (x - f16(x))*64 | f16(x)/64] (first nseg). */
int evc_cast_f32_to_f16_segs(const float* in, int64_t ld_in, int R, int C, int nseg, evc_f16* out, void* stream);
/* f16 image of an LSTM kernel [R][Kin+H] with both parts K-extended by the weights' low-order halves: out [R][2Kin + 2H] =
 * [f16(Wx) | (Wx - f16(Wx))*64 | f16(Wh) | (Wh - f16(Wh))*64] (the upper layer of evc_lstm_stack2_fwd_f16). */
int evc_cast_f32_to_f16_wlo(const float* in, int64_t ld_in, int R, int Kin, int H, evc_f16* out, void* stream);
/* wide split-bf16 image of a [R][C] f32 matrix: out rows [lo | hi] (lo_first = 1: the A operand of evc_gemm_nt_split) or
 * [hi | lo] (0: its B operand); hi = bf16(x), lo = bf16(x - hi); C % 4 == 0, ld_out >= 2C. */
int evc_cast_f32_to_bf16_wide(const float* in, int64_t ld_in, int R, int C, evc_bf16* out, int64_t ld_out, int lo_first, void* stream);
/* C[M,N] = (A_hi + A_lo) . (B_hi + B_lo)^T (+ bias), f32 out, to ~2^-16 relative, as ONE K-extended launch of the plain NT
 * loop: A_lohi rows [lo(K) | hi(K)] (lda >= 2K), B_hilo rows [hi(K) | lo(K)] (ldb >= 2K); K % 64 == 0.  The "high" precision
 * form of the MoE head (cs/video_level_models.py:423-435) and of the L2 level's hoisted input projection. */
int evc_gemm_nt_split(const evc_bf16* A_lohi, int64_t lda, const evc_bf16* B_hilo, int64_t ldb, float* C, int64_t ldc,
                      int M, int N, int K, const float* bias, void* stream);
/* C [M][N] f32 = A16 . B16^T (IEEE f16, K16 deep) + 2^scale_exp A8 . B8^T (OCP e4m3 bytes, K8 deep) + bias in ONE launch: a product with its
 * low-order corrections behind its f16 stages on v_mfma_scale_f32_16x16x128_f8f6f4 (per K element twice the MFMA rate, half the operand
 * bytes).  The "high" precision MoE head (cs/video_level_models.py:423-448 at f32-operand accuracy): A rows from
 * evc_cast_f32_to_f16_fp8x(hi_exp 6, lo_exp 17) = [f16(x) | e4m3(x 2^6) | e4m3((x - f16(x)) 2^17)], B16 = f16(W), B8 = [e4m3((W - f16(W)) 2^18) |
 * e4m3(W 2^7)] (evc_cast_f32_to_fp8_lo with hi_cols = K, or the epilogue of evc_moe_grad_update_wide), scale_exp = -24: 2.5e-5 on logits of
 * magnitude 8 (f16 alone 8e-4, bf16 4e-3).  lda / ldb in halfwords, lda8 / ldb8 in bytes (multiples of 16); K16 % 64 == 0, K8 % 128 == 0, K8 >= 512. */
int evc_gemm_nt_f16_fp8(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                        const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                        const float* bias, void* stream);
/* out rows of 4C bytes = [f16(x) (C halfwords) | e4m3(x 2^hi_exp) (C bytes) | e4m3((x - f16(x)) 2^lo_exp) (C bytes)] for an f32 matrix [R][C]
 * (C % 32 == 0): the A16 / A8 operands of evc_gemm_nt_f16_fp8 (A16 = out, lda = 2C halfwords; A8 = (uint8_t*)out + 2C, lda8 = 4C bytes). */
int evc_cast_f32_to_f16_fp8x(const float* in, int64_t ld_in, int R, int C, int hi_exp, int lo_exp, evc_f16* out, void* stream);
/* Dynamic e4m3 range of an ACTIVATION operand (round 6; the MoE head's input is the L2 state [c0|h0|c1|h1] of cs/frame_level_models.py:255-263 and
 * its cell-state half is unbounded: |c| ~ 16 on towers trained for 512 steps, where the fixed e4m3(x 2^6) saturates at 7).
 *   evc_absmax_partials           ws[64] f32 = partial maxima of |x| over an f32 matrix [R][C] (plain stores: no atomics, nothing to zero)
 *   evc_cast_f32_to_f16_fp8x_dyn  evc_cast_f32_to_f16_fp8x with both e4m3 images scaled by 2^(hi_exp - d) / 2^(lo_exp - d), d >= 0 the fewest bits
 *                                 with max|x| 2^(hi_exp - d) <= 448
 *   evc_gemm_nt_f16_fp8_dyn       evc_gemm_nt_f16_fp8 whose e4m3 products are scaled by 2^(scale_exp + d), the same d from the same ws / a8_hi_exp
 * d = 0 (max|x| <= 448 2^-hi_exp) gives the bits of the fixed-scale entries. */
int evc_absmax_partials(const float* in, int64_t ld_in, int R, int C, float* ws, void* stream);
int evc_cast_f32_to_f16_fp8x_dyn(const float* in, int64_t ld_in, int R, int C, int hi_exp, int lo_exp, const float* amax_ws, evc_f16* out, void* stream);
int evc_gemm_nt_f16_fp8_dyn(const evc_f16* A16, int64_t lda, const uint8_t* A8, int64_t lda8, const evc_f16* B16, int64_t ldb,
                            const uint8_t* B8, int64_t ldb8, float* C, int64_t ldc, int M, int N, int K16, int K8, int scale_exp,
                            const float* amax_ws, int a8_hi_exp, const float* bias, void* stream);
/* split-bf16 cast: hi = bf16(x), lo = bf16(x - hi).  Three NT products (hi.hi + hi.lo + lo.hi, via
 * evc_gemm_nt with accumulate) then reproduce an f32-operand GEMM to ~2^-16 relative: the
 * "high" precision forward mode for models whose activations are O(1) (DBoF after batch-norm). */
int evc_cast_f32_to_bf16_split(const float* in, int64_t ld_in, int R, int C, evc_bf16* hi, evc_bf16* lo,
                               int64_t ld_out, void* stream);
/* out[r] = sum_c in[r][c] (bf16 in, f32 out): bias gradients from dz^T. */
int evc_rowsum_bf16(const evc_bf16* in, int64_t ld_in, int R, int C, float* out, void* stream);
/* out[c'] = sum_r in[r][c] (bf16 in, f32 out; out is zeroed inside): bias gradients straight from
 * dz4; deinterleave_H > 0 (C == 4*H) maps gate-interleaved column u*4+g to c' = g*H+u. */
int evc_colsum_bf16(const evc_bf16* in, int64_t ld_in, int R, int C, int deinterleave_H, float* out, void* stream);

/* ---- a5: MoeModel tail -----------------------------------------------------
 * cs/video_level_models.py:437-448: softmax over M+1 gate logits, sigmoid over
 * M expert logits, p = sum_{m<M} g_m e_m.   gate_logits [B][V*(M+1)],
 * expert_logits [B][V*M] (bias already added), class-major columns.
 * pred [B][V] f32; rowsum[B] = sum_c pred (used by L_PRED).  M <= 4. */
int evc_moe_tail_fwd(const float* gate_logits, const float* expert_logits, int B, int V, int M,
                     float* pred, float* rowsum, void* stream);
/* dL/dlogits from dL/dpred; writes bf16 (GEMM operand) transposed-ready row-major. */
int evc_moe_tail_bwd(const float* gate_logits, const float* expert_logits, const float* dpred,
                     int B, int V, int M, evc_bf16* dgate, int64_t ld_dgate,
                     evc_bf16* dexpert, int64_t ld_dexpert, void* stream);

/* ---- a6 + a7: losses ---------------------------------------------------------
 * CrossEntropyLoss (cs/losses.py:90-97), L_PRED = sum_b KL(pT/sum pT || pS/sum pS)
 * (cs/train.py:398-402), L_REP = mean_b sum_d (sT-sS)^2 (cs/train.py:359-362).
 * Each call ACCUMULATES its scalar into *loss (f32, device; zero it first) and
 * writes/accumulates the gradient.
 */
/* loss += scale_loss * mean_b CE ; dpred (=|+=) grad_scale * dCE/dpred. labels uint8 [B][V]. */
int evc_ce_loss(const float* pred, const uint8_t* labels, int B, int V, float grad_scale,
                float* loss, float* dpred, int accumulate_grad, void* stream);
/* The same with the loss scalar summed in a FIXED order (EVC_DETERMINISTIC=1 callers, round 5): every block leaves its partial sum in
 * partials[block] (256 floats of caller scratch) and a one-thread pass adds them in block order - the full grid still computes the gradient. */
int evc_ce_loss_ordered(const float* pred, const uint8_t* labels, int B, int V, float grad_scale,
                        float* loss, float* dpred, int accumulate_grad, float* partials, void* stream);
/* loss += KL sum ; dpred_student (=|+=) grad_scale * dKL/dpS. */
int evc_kl_pred_loss(const float* pred_t, const float* rowsum_t, const float* pred_s, const float* rowsum_s,
                     int B, int V, float grad_scale, float* loss, float* dpred_s, int accumulate_grad,
                     void* stream);
/* loss += mean_b sum_d (sT-sS)^2 ; dstate_s (=|+=) grad_scale * d/dsS. */
int evc_rep_loss(const float* state_t, const float* state_s, int B, int D, float grad_scale,
                 float* loss, float* dstate_s, int accumulate_grad, void* stream);
int evc_rep_loss_ordered(const float* state_t, const float* state_s, int B, int D, float grad_scale,
                         float* loss, float* dstate_s, int accumulate_grad, float* partials /* 256 floats of scratch: fixed-order sum, as evc_ce_loss_ordered */,
                         void* stream);

/* ---- a8 + a9: regulariser, per-tensor clip, TF-Adam ---------------------------
 * slim.l2_regularizer (cs/video_level_models.py:428,434) folded into the
 * gradient: g_eff = g + l2_coeff * p  (l2_coeff = regularization_penalty*1e-8).
 * Pass 1: sums[0] += sum(g_eff^2), sums[1] += sum(p^2)   (f32 device, zero first); p == NULL (allowed when
 *   l2_coeff == 0: a tensor without regulariser) reads the gradient only and leaves sums[1] alone.
 * Pass 2: clip_by_norm per tensor (slim create_train_op, cs/train.py:329-334)
 *   scale = clip / max(sqrt(sums[0]), clip) (clip <= 0: no clipping), then
 *   tf.train.AdamOptimizer update with lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
 *   p -= lr_t * m / (sqrt(v) + eps); optionally refreshes a bf16 copy of p.
 */
int evc_grad_sqnorm(const float* g, const float* p, float l2_coeff, int64_t n, float* sums, void* stream);
int evc_clip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float l2_coeff,
                       const float* sums, float clip_norm, float lr_t, float beta1, float beta2, float eps,
                       evc_bf16* p_bf16, void* stream);
/* The same two graph nodes (per-tensor clip_by_norm + Adam; cs/train.py:241-242,329-334) for up to 16 SMALL tensors without an l2 term - biases,
 * batch-norm scales / offsets - in one launch: workgroup i computes tensor i's squared gradient norm (block sum: a fixed order), leaves
 * sums[i] = {|g|^2, 0} and applies the update (round 5: the DBoF step spent 14 launches on 12 k parameters).  p, g, m, v, n, sums: HOST arrays of
 * `count` device pointers / sizes (copied into the launch's arguments); n[i] <= 2^15 (one workgroup walks a tensor). */
int evc_clip_adam_small(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                        float* const* sums, float clip_norm, float lr_t, float beta1, float beta2, float eps, void* stream);

/* The reader's uint8 frames as EXACT f16 integers (round 6; cs/readers.py:146-174 delivers uint8, cs/utils.py:22-25 dequantises x = (2/255)(2q - 255)
 * + 1/128): second images with rows of 3F bytes [f16(2q - 255) (F halfwords) | e4m3(x_hat 2^7) (F bytes)] and rs1 / rs2 [steps][rows] f32 =
 * (2/255) / |x| per frame row (0 for padded frames), x_hat = rs (c + 255/256).  out1 / out2: the usual bf16 images; row plans as
 * evc_l2norm_chunk_fwd; out1 == NULL: student-only. */
int evc_l2norm_chunk_int(const uint8_t* x_u8, const int32_t* num_frames, int B, int T, int F, int C1, evc_bf16* out1,
                         int every_n, int C2, evc_bf16* out2, evc_f16* out1_int, evc_f16* out2_int, float* rs1, float* rs2,
                         const int32_t* row_pos1, int rows1, const int32_t* row_pos2, int rows2, void* stream);

/* ---- a11: FrameLevelLogisticModel pooling ------------------------------------
 * cs/frame_level_models.py:72-78: sum over ALL T (padded) frames / true n.
 * x [B][T][F] f32, or x_u8 [B][T][F] as the reader delivers it (Dequantize cs/utils.py:22-25 fused; frames >=
 * num_frames are padding = 0) - exactly one non-NULL -> avg [B][F] f32 (required) and bf16 (GEMM operand, optional).
 * normalize=1 fuses tf.nn.l2_normalize of every frame (cs/train.py:256) into the pooling pass. */
int evc_meanpool_fwd(const float* x, const uint8_t* x_u8, const int32_t* num_frames, int B, int T, int F, int normalize,
                     float* avg_f32, evc_bf16* avg_bf16, void* stream);
/* elementwise sigmoid fwd (in place on f32 [n]) and dz = dp * p * (1-p) -> bf16 */
int evc_sigmoid_fwd(float* z, int64_t n, void* stream);
int evc_sigmoid_bwd(const float* p, const float* dp, int64_t n, evc_bf16* dz, void* stream);

/* ---- a10: DbofModel pieces ----------------------------------------------------
 * SampleRandomFrames (cs/model_utils.py:39-58): idx = int32(u * float32(n));
 * gathers x[b, idx[b,s], :] -> out [B*S][F] f32.  u [B][S] f32 supplied by caller.
 * normalize=1 l2-normalises each gathered frame (cs/train.py:256: same values as
 * normalising all 300 frames first, 10x less work).  x f32 or x_u8 (the reader's uint8 tensor: Dequantize fused,
 * frames >= num_frames are padding) - exactly one non-NULL. */
int evc_sample_frames_gather(const float* x, const uint8_t* x_u8, const float* u, const int32_t* num_frames, int B, int T, int F,
                             int S, int normalize, float* out, int32_t* idx_out, void* stream);
/* slim.batch_norm training statistics over rows: mean[C], var[C] (biased, f64 accumulation). x [R][C] f32. */
int evc_bn_stats(const float* x, int R, int C, double* ws /* 2*C, zeroed inside */, float* mean, float* var,
                 void* stream);
/* y = relu6?(gamma*(x-mean)*rsqrt(var+1e-3)+beta); writes f32 y and/or bf16 y. */
int evc_bn_apply(const float* x, int R, int C, const float* mean, const float* var, const float* gamma,
                 const float* beta, int relu6, float* y_f32, evc_bf16* y_bf16, void* stream);
/* The two halves of evc_bn_stats, exposed so data-parallel ranks can all-reduce
 * the f64 partial sums ws[2*C] in between (SyncBN == single-device batch statistics). */
int evc_bn_stats_partial(const float* x, int R, int C, double* ws, void* stream);
int evc_bn_stats_finalize(const double* ws, int R_total, int C, float* mean, float* var, void* stream);
/* slim.batch_norm UPDATE_OPS: moving -= (1-decay)*(moving - batch_value), decay 0.999. */
int evc_ema_update(float* moving, const float* batch_value, float decay, int n, void* stream);
/* backward of relu6?(bn(x)) given dy [R][C]: dx f32/bf16, dgamma, dbeta.  If argmax != NULL,
 * dy is the POOLED gradient [R/S][C] and row r=(b,s) receives it only where argmax[b][c]==s
 * (gradient of FramePooling 'max' fused in).  partial/finalize are split for SyncBN. */
int evc_bn_bwd_partial(const float* x, const float* dy, int R, int C, const float* mean, const float* var,
                       const float* gamma, const float* beta, int relu6, const int32_t* argmax, int S,
                       double* ws /* 2*C, zeroed inside */, void* stream);
int evc_bn_bwd_finalize(const float* x, const float* dy, int R, int R_total, int C, const float* mean,
                        const float* var, const float* gamma, const float* beta, int relu6,
                        const int32_t* argmax, int S, const double* ws, float* dx_f32, evc_bf16* dx_bf16,
                        float* dgamma, float* dbeta, void* stream);
int evc_bn_relu6_bwd(const float* x, const float* dy, int R, int C, const float* mean, const float* var,
                     const float* gamma, const float* beta, int relu6, const int32_t* argmax, int S,
                     double* ws /* 2*C, zeroed inside */,
                     float* dx_f32, evc_bf16* dx_bf16, float* dgamma, float* dbeta, void* stream);
/* cluster_bn + relu6 + FramePooling('max') in one pass over act [B][S][C] (cs/frame_level_models.py:149-167). */
int evc_bn_relu6_framepool_fwd(const float* act, int B, int S, int C, const float* mean, const float* var,
                               const float* gamma, const float* beta, float* pooled_f32, evc_bf16* pooled_bf16,
                               int32_t* argmax, void* stream);
/* FramePooling 'max' (cs/model_utils.py:77-78): y [B][S][C] f32 -> pooled [B][C] + argmax. */
int evc_framepool_max_fwd(const float* y, int B, int S, int C, float* pooled_f32, evc_bf16* pooled_bf16,
                          int32_t* argmax, void* stream);
int evc_framepool_max_bwd(const float* dpooled, const int32_t* argmax, int B, int S, int C, float* dy, void* stream);

/* Two-layer L1 level (many rows, row plans), bf16 - HierarchicalLstmModel's MultiRNNCell of two BasicLSTMCells over one chunk of frames
 * (cs/frame_level_models.py:221-257, tf.nn.dynamic_rnn :247,255): what two evc_lstm_layer_fwd calls compute (layer 1 reading layer 0's output
 * slabs), as T + 1 launches - layer 0's step s and layer 1's step s-1 are independent, every workgroup walks both tiles and the second tile's
 * first ring stages are issued under the first tile's gate tail.  Bit-identical results.  x [T][M][Kin], hbuf0 / hbuf1 [(T+1)][M][H] (slab 0 is
 * zeroed here), states as in evc_lstm_layer_fwd with the two layers' c / h columns, gates / c_all per layer or all NULL (evaluation),
 * row_map / rows_per_step as in evc_lstm_layer_fwd (NULL: every row at every step).  Kin % 64 == 0, H % 64 == 0. */
int evc_lstm_level2_fwd(const evc_bf16* x, const evc_bf16* wT0, const float* bias0, const evc_bf16* wT1, const float* bias1,
                        const int32_t* len, int T, int M, int Kin, int H, evc_bf16* hbuf0, evc_bf16* hbuf1,
                        float* c_state0, float* h_state0, float* c_state1, float* h_state1, int64_t ld_state,
                        void* gates0, evc_bf16* c_all0, void* gates1, evc_bf16* c_all1,
                        const int32_t* row_map, const int32_t* rows_per_step, void* stream);

/* Two-layer stack, BPTT in wavefront order: layer 0's step t+1 and layer 1's step t share a launch (T+1 dependent
 * launches instead of 2T) and the gradient arriving at layer 0 from layer 1 is contracted inside layer 0's step
 * ([dz0 | dz1] . [Wh0 ; Wx1]^T, K = 8H) instead of a hoisted dX product.  No reference counterpart: tf.gradients of
 * cs/frame_level_models.py:221-257.  w_il0 [Kin0+H][4H], w_il1 [2H][4H] (backward layout, 4H gate-interleaved);
 * dS [M][4H] f32 = [c0 | h0 | c1 | h1]; dz0/dz1 [T][M][H][4] bf16 out; dc_ws0/1 [M][H] f32 scratch; db0/db1 [4H]
 * accumulated (zero them first); row_map / rows_per_step as in evc_lstm_layer_bwd.  H % 128 == 0.
 * M <= 512 (the M ~ batch stacks of the L2 levels; round 5): the same wavefront on the skinny kernel's pair launches (32 x 32 tiles, K split
 * over four waves, 64 KB of LDS: two workgroups per CU); M > 512: 128 x 128 ring tiles. */
int evc_lstm_stack2_bwd(const evc_bf16* w_il0, const evc_bf16* w_il1, const int32_t* len, int T, int M, int Kin0, int H,
                        const void* gates0, const evc_bf16* c_all0, const void* gates1, const evc_bf16* c_all1,
                        const float* dS, int64_t ld_dS, float* dc_ws0, float* dc_ws1, evc_bf16* dz0, evc_bf16* dz1,
                        float* db0, float* db1, const int32_t* row_map, const int32_t* rows_per_step, void* stream);

/* ---- a10 fused: the [frames x clusters] activation never leaves the chip in f32 ---------------------------
 * (cs/frame_level_models.py:126-167: reshape -> input_bn -> matmul(cluster_weights) -> cluster_bn -> relu6 ->
 *  FramePooling 'max'; cs/model_utils.py:39-58,77-78).  Padded frame layout: every video owns 32 frame slots
 * (iterations <= 32), 4 videos form a 128-row block, frame s of video b is row
 * (b>>2)*128 + (s>>2)*16 + (b&3)*4 + (s&3); all [Mp][..] operands below use it (Mp = padded_rows). */
/* sizes of the caller-provided buffers: padded_rows = ceil(B/4)*128; gather_part_rows / gemm_part_rows = rows of the
 * [rows][2][width] f32 column-partial-sum buffers of evc_dbof_gather / evc_dbof_cluster_pool_fwd. */
int evc_dbof_workspace(int B, int S, int32_t* padded_rows, int32_t* gather_part_rows, int32_t* gemm_part_rows);
/* SampleRandomFrames + tf.nn.l2_normalize of the gathered frames into r [Mp][F] f32 (only the sampled slots are
 * written), from x_f32 [B][T][F] or x_u8 [B][T][F] (Dequantize cs/utils.py:22-25, frames >= num_frames are padding);
 * exactly one of the two is non-NULL.  idx_out [B][S] int32 (optional) = the int32-truncated indices.
 * part (optional) [gather_part_rows][2][F]: per-workgroup column sums of r and r^2 for the input batch-norm. */
int evc_dbof_gather(const float* x_f32, const uint8_t* x_u8, const float* u, const int32_t* num_frames, int B, int T,
                    int F, int S, int normalize, float* r, int32_t* idx_out, float* part, void* stream);
/* [P][2][C] f32 partial sums -> ws f64 [2C] (sum x | sum x^2), rows added in index order (deterministic). */
int evc_bn_partials_reduce(const float* part, int P, int C, double* ws, void* stream);
/* mean / biased variance from ws over R_total rows + slim.batch_norm's moving averages (optional), one launch. */
int evc_bn_finalize_ema(const double* ws, int R_total, int C, float* mean, float* var, float* moving_mean,
                        float* moving_var, float decay, void* stream);
/* input_bn applied: r -> r_bn = gamma*xhat+beta (bf16, + low half r_bn_lo for the split-bf16 mode, optional) and
 * xhat (bf16, optional: the operand of the weight-gradient product); empty frame slots are written as zeros. */
int evc_dbof_input_bn_apply(const float* r, int B, int S, int F, const float* mean, const float* var, const float* gamma,
                            const float* beta, evc_bf16* r_bn, evc_bf16* r_bn_lo, evc_bf16* xhat, void* stream);
/* act = r_bn . wT^T  (wT [C][F] bf16 = cluster_weights transposed) on 256x256 MFMA tiles with the epilogue
 *   part [gemm_part_rows][2][C]  column sums of act, act^2 over each 128-row half tile (NULL: evaluation)
 *   xsel [B][C], arg [B][C]      per (video, cluster): the max over the sampled frames of sign(gamma)*act, stored as
 *                                the selected act itself, and its frame slot (first maximum wins)
 *   act [Mp][C] bf16             the activation for the backward pass (NULL: not kept).
 * r_bn_lo / wT_lo non-NULL: split-bf16 operands (hi.hi + hi.lo + lo.hi).
 * Plain operands and >= 512 output tiles (round 5): one workgroup per CU walks the row tiles of one column panel of wT (the next tile's first
 * ring stages are issued under the current tile's epilogue; a panel is fetched by one XCD's L2) - the same arithmetic in the same order as one
 * tile per workgroup, bit-identical results; EVC_DBOF_WALK=0 switches it off, 2 forces it at any tile count (tests). */
int evc_dbof_cluster_pool_fwd(const evc_bf16* r_bn, const evc_bf16* r_bn_lo, const evc_bf16* wT, const evc_bf16* wT_lo,
                              int B, int S, int F, int C, const float* gamma, evc_bf16* act, float* part, float* xsel,
                              uint8_t* arg, void* stream);
/* The "high" precision forward of the two entries above on f16 + e4m3 operands (both operands' roundings corrected as e4m3 stages behind the f16
 * stages of the same launch: evc_gemm_nt_f16_fp8's arithmetic, 2x the MFMA time of the bf16 product instead of the split-bf16 form's 3x):
 * evc_dbof_input_bn_apply_f16fp8 writes r_rows [Mp] rows of 4F bytes = [f16(y) | e4m3(y 2^hi_exp) | e4m3((y - f16(y)) 2^lo_exp)], y = the
 * batch-normalised frame (and xhat as above); evc_dbof_cluster_pool_fwd_f16fp8 contracts them against wT16 [C][F] = f16(W) and wT8 [C][2F] =
 * [e4m3((W - f16(W)) 2^w_lo_exp) | e4m3(W 2^w_hi_exp)] (evc_cast_f32_to_fp8_lo with hi_cols = F), scale_exp = -(hi_exp + w_lo_exp) =
 * -(lo_exp + w_hi_exp); epilogue and outputs as evc_dbof_cluster_pool_fwd.  F % 128 == 0 (cs/frame_level_models.py:149-160). */
int evc_dbof_input_bn_apply_f16fp8(const float* r, int B, int S, int F, const float* mean, const float* var, const float* gamma,
                                   const float* beta, evc_f16* r_rows, int hi_exp, int lo_exp, evc_bf16* xhat, void* stream);
int evc_dbof_cluster_pool_fwd_f16fp8(const evc_f16* r_rows, const evc_f16* wT16, const uint8_t* wT8, int scale_exp,
                                     int B, int S, int F, int C, const float* gamma, evc_bf16* act, float* part, float* xsel,
                                     uint8_t* arg, void* stream);
/* pooled = relu6(gamma*(xsel-mean)*rsqrt(var+1e-3)+beta): f32, bf16 (optional), bf16 low half (optional). */
int evc_dbof_pool_finish(const float* xsel, int B, int C, const float* mean, const float* var, const float* gamma,
                         const float* beta, float* pooled_f32, evc_bf16* pooled_bf16, evc_bf16* pooled_lo, void* stream);
/* backward of max-pool + relu6 + cluster_bn, in place on act [Mp][C] bf16 (-> dact).  ws f64 [2C] = sum d, sum d*xhat
 * over the [B][C] selected entries (evc_bn_bwd_partial on xsel / dpooled with R = B), all-reduced under data
 * parallelism; R_total = sampled frames of the global batch.  dgamma / dbeta [C] (optional) receive cluster_bn's own
 * gradients (= the two sums). */
int evc_dbof_dact(evc_bf16* act, const float* dpooled, const float* pooled, const uint8_t* arg, const float* mean,
                  const float* var, const float* gamma, const double* ws, int R_total, int B, int S, int C, float* dgamma,
                  float* dbeta, void* stream);
/* evc_gemm_tn with B given as two column segments: C[:, 0:N1] from B1 [K][ldb1], C[:, c_col2:c_col2+N2] from B2 [K][ldb2] - the
 * x- and h-part of a layer's weight gradient dW^T = dz^T . [x | h_prev] in ONE launch (dz is read once, twice the tiles per
 * launch: 4096 x 2048 x 56 640 runs 1.0 ms against 2 x 0.6).  N1 % 256 == 0 (a workgroup's columns lie in one segment);
 * c_col2 >= N1 is where the second segment starts in C (c_col2 > N1, a gap the caller fills with another product, needs
 * accumulate = 1 on a zeroed C). */
int evc_gemm_tn2(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                 int N2, int c_col2, float* C, int64_t ldc, int M, int K, int row_interleave_H, int accumulate, void* stream);
/* evc_gemm_tn / evc_gemm_tn2 over TIME SLABS with a live prefix (ABI 105): K = nslabs * slab_rows rows, slab t holding rows_per_slab[t] (HOST
 * array) live rows at its start - the layout of a row-planned LSTM level (evc_sort_rows_by_len: rows sorted by length), whose weight-gradient
 * products dW^T = dz^T . [x | h_prev] (the MatMul(transpose_a) gradient nodes of BasicLSTMCell's kernel, cs/frame_level_models.py:221-250) contract
 * over every row of the [T][P] images although the rows beyond a step's active prefix carry zero gate gradients.  The K walk covers
 * ceil(rows_per_slab[t] / 32) steps of slab t and jumps over the rest: the same sums (the skipped rows of A are zeros) for 5 % fewer MFMAs on
 * the teacher's L1 level.  slab_rows % 32 == 0; B2 == NULL: one column segment (evc_gemm_tn); more than 16 non-empty slabs, nothing to skip, or
 * EVC_DETERMINISTIC: the plain product over all K rows. */
int evc_gemm_tn2_rows(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                      int N2, int c_col2, float* C, int64_t ldc, int M, int slab_rows, int nslabs, const int32_t* rows_per_slab,
                      int row_interleave_H, int accumulate, void* stream);
/* evc_gemm_tn with the K range cut into nslab partial products stored plainly at slabs + s*M*N (no atomics). */
int evc_gemm_tn_slabs(const evc_bf16* A, int64_t lda, const evc_bf16* B, int64_t ldb, float* slabs, int M, int N, int K,
                      int nslab, void* stream);
/* evc_gemm_tn2's product with the K split stored as nslab plain partial images (slab s at slabs + s*slab_stride, each laid out like C: row stride ldc,
 * rows de-interleaved, second segment at column c_col2; B2 == NULL: one segment) and their fixed-order sum: the weight gradients of
 * EVC_DETERMINISTIC=1 without atomics and without giving up the K split (round 5; tf.gradients' MatMul(transpose_a=True), cs/frame_level_models.py:221-257). */
int evc_gemm_tn2_slabs(const evc_bf16* A, int64_t lda, const evc_bf16* B1, int64_t ldb1, int N1, const evc_bf16* B2, int64_t ldb2,
                       int N2, int c_col2, float* slabs, int64_t ldc, int64_t slab_stride, int M, int K, int row_interleave_H,
                       int nslab, void* stream);
int evc_sum_slabs(const float* slabs, int64_t slab_stride, int nslab, int M, int N, int64_t ld, float* C, int64_t ldc, int accumulate,
                  void* stream);
/* G = sum of the slabs [nslab][C][F] (= dact^T . xhat): dW[c][f] = gamma_in[f]*G, dgamma_in[f] = sum_c W[c][f]*G,
 * dbeta_in = 0 (the batch-norm backward's output sums to zero over the batch).  part_ws: [ceil(C/8)][F] f32 scratch
 * (per-block column sums, added in block order: run-to-run identical). */
int evc_dbof_wgrad_finish(const float* slabs, int nslab, int C, int F, const float* W, const float* gamma_in, float* dW,
                          float* dgamma_in, float* dbeta_in, float* part_ws, void* stream);

/* ---- NetVLAD aggregation (EXTENSION: the reference's NetVLADModel is an empty stub, cs/frame_level_models.py:341-347; the
 * math is oracle/model_math.py::netvlad_fwd).  Sampled frames row-major [B*S][..]; V / dV / Y are [B][K][F] (cluster-major).
 * The GEMM-shaped parts of the tower use evc_gemm_nt / evc_gemm_tn. ------------------------------------------------------ */
/* a = softmax over the K clusters of cluster_bn(act), per sampled frame; and dz = a * (da - sum_k a*da). */
int evc_netvlad_softmax_fwd(const float* act, int R, int K, const float* mean, const float* var, const float* gamma,
                            const float* beta, float* a, void* stream);
int evc_netvlad_softmax_bwd(const float* a, const float* da, int R, int K, float* dz, void* stream);
/* V[b][k][:] = sum_s a[b,s,k] * x_bn[b,s,:] - asum[b][k] * c2[k][:], x_bn = input_bn(r) recomputed from r [B*S][F] f32 and
 * the statistics; asum [B][K] = sum_s a.  Backward: da [B*S][K], dx_bn [B*S][F] from dV. */
int evc_netvlad_aggregate_fwd(const float* a, const float* r, int B, int S, int K, int F, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* c2, float* V, float* asum, void* stream);
int evc_netvlad_aggregate_bwd(const float* a, const float* r, int B, int S, int K, int F, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* c2, const float* dV, float* da, float* dx,
                              void* stream);
/* dc2[k][:] = - sum_b asum[b][k] * dV[b][k][:] */
int evc_netvlad_dcenters(const float* asum, const float* dV, int B, int K, int F, float* dc2, void* stream);
/* U_k = V_k / |V_k| per cluster, Y = U / |U| (tf.nn.l2_normalize, epsilon 1e-12 on the squared norms); n1 [B][K], n2 [B]
 * are kept for the backward pass, which recomputes U and Y from V. */
int evc_netvlad_normalize_fwd(const float* V, int B, int K, int F, float* n1, float* n2, float* Y_f32, evc_bf16* Y_bf16,
                              void* stream);
int evc_netvlad_normalize_bwd(const float* V, const float* n1, const float* n2, const float* dY, int B, int K, int F, float* dV,
                              void* stream);

/* ---- one LSTM layer's train op in two launches (csrc/evc_optim.hip; a9: cs/train.py:329-334,413-418 + AdamOptimizer :241-242) ----
 * evc_sqnorm2_partials: part[0..1023] = per-block sums of squares of ga [na], part[1024] = sum of squares of gb [nb] (gb may be
 *   NULL with nb = 0); plain stores, fixed order.  part: 1025 floats.
 * evc_lstm_adam_fused: per-tensor clip_by_norm(clip_norm) from those partials + TF-Adam (arithmetic of evc_clip_adam_step, no l2
 *   term) of the kernel p [4H][C] and the bias pb [4H], and from the same registers the operand images of the new kernel:
 *   p_bf16 [4H][C]; pT_bf16 [C][ldT] with column u*4+g <- row g*H+u (the gate-interleaved layout of evc_lstm_layer_bwd);
 *   p_f16 (or NULL) rows [f16(Wx) | f16(Wx)/64 | (Wx - f16(Wx))*64 (the first nseg of these blocks over the nin input columns) |
 *   f16(Wh)] = evc_cast_f32_to_f16 (nseg 1) / evc_cast_f32_to_f16_wide(h_ext 0); p_fp8 (or NULL) = evc_cast_f32_to_fp8_lo of the
 *   columns from fp8_col0 on (hi_cols, exponents as there).  sums_w[0] / sums_b[0] receive the two squared gradient norms. */
int evc_sqnorm2_partials(const float* ga, int64_t na, const float* gb, int64_t nb, float* part, void* stream);
int evc_lstm_adam_fused(float* p, const float* g, float* m, float* v, float* pb, const float* gb, float* mb, float* vb, int H, int C,
                        const float* part, float* sums_w, float* sums_b, float clip_norm, float lr_t, float beta1, float beta2, float eps,
                        evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT, evc_f16* p_f16, int64_t ld16, int nin, int nseg,
                        uint8_t* p_fp8, int64_t ld8, int fp8_col0, int fp8_hi_cols, int fp8_lo_exp, int fp8_hi_exp, int fp8_hi_tail, void* stream);
/* (fp8_hi_tail = 1, round 6: the e4m3 image in evc_cast_f32_to_fp8_lohi's layout, ld8 >= 2 (C - fp8_col0)) */

/* ---- a10, the branches no launcher of the reference selects (cs/frame_level_models.py:126-187; towers.DbofGenericTower) ----
 * evc_sample_sequence_gather: SampleRandomSequence (cs/model_utils.py:11-36): S consecutive frames from start = int32(u[b] *
 *   float32(max(n - S, 0) + 1)), index min(start + s, n - 1); arguments as evc_sample_frames_gather, u [B].
 * evc_relu6_fwd / _bwd: tf.nn.relu6 on a pre-activation that already carries its bias (--dbof_add_batch_norm False) and its gradient
 *   dy * [0 < x < 6]; f32 and / or bf16 outputs.
 * evc_framepool_mean_fwd / _bwd: FramePooling 'average' (cs/model_utils.py:75-76): mean over the S frames of y [B][S][C]; dy = dpooled / S. */
int evc_sample_sequence_gather(const float* x, const uint8_t* x_u8, const float* u, const int32_t* num_frames, int B, int T, int F,
                               int S, int normalize, float* out, int32_t* idx_out, void* stream);
int evc_relu6_fwd(const float* x, int64_t n, float* y_f32, evc_bf16* y_bf16, void* stream);
int evc_relu6_bwd(const float* x, const float* dy, int64_t n, float* dx_f32, evc_bf16* dx_bf16, void* stream);
int evc_framepool_mean_fwd(const float* y, int B, int S, int C, float* pooled_f32, evc_bf16* pooled_bf16, void* stream);
int evc_framepool_mean_bwd(const float* dpooled, int B, int S, int C, float* dy, void* stream);
/* evc_colsum_bf16 without atomics: row block y (of min(R / 64, ws_rows)) leaves its partial sums in ws[y][C] (plain stores), a
 * second launch adds them in index order - the bias gradient from dz under EVC_DETERMINISTIC=1. */
int evc_colsum_bf16_det(const evc_bf16* in, int64_t ld_in, int R, int C, int deinterleave_H, float* out, float* ws, int ws_rows, void* stream);
/* evc_lstm_adam_fused's pass for a plain 2-D weight p [R][C] without a bias (DBoF cluster / hidden weights, the logistic model's matrix): partials from
 * evc_sqnorm2_partials(g, R*C, NULL, 0); pT_bf16 [C][ldT] = the plain transpose, pad columns R..round_up(R, 64) written as zeros; p_f16 [R][ld16] = f16(W)
 * and p_fp8 [R][ld8] = evc_cast_f32_to_fp8_lo(W, hi_cols) optional. */
int evc_adam2d_fused(float* p, const float* g, float* m, float* v, int R, int C, const float* part, float* sums_w, float clip_norm, float lr_t,
                     float beta1, float beta2, float eps, evc_bf16* p_bf16, evc_bf16* pT_bf16, int64_t ldT, evc_f16* p_f16, int64_t ld16,
                     uint8_t* p_fp8, int64_t ld8, int fp8_hi_cols, int fp8_lo_exp, int fp8_hi_exp, void* stream);

/* utility: out[i] = value for n floats (avoids torch for tiny fills inside C loops) */
int evc_fill_f32(float* p, int64_t n, float value, void* stream);
/* Measurement aid, not part of the path: `blocks` workgroups of `threads` threads with `lds_bytes` of LDS each stay resident for
 * `microseconds` - the footprint of a collective's kernel on the compute units, for the one-GPU stand-in runs of DESIGN.md 6.1. */
int evc_debug_occupy(int blocks, int threads, int lds_bytes, double microseconds, void* stream);
/* Scheduling aid: a HIP stream whose kernels may only run on the compute units set in `mask` (hipExtStreamCreateWithCUMask;
 * `words` 32-bit words, bit i of the mask = CU i in the driver's XCD-interleaved enumeration, so the low 8 n bits are n CUs of
 * each XCD).  The caller owns the stream (evc_stream_destroy).  Used by streams.cu_masked_stream for the optimizer side stream
 * (DESIGN.md 5); the path itself never creates streams. */
int evc_stream_create_cu_mask(const unsigned* mask, int words, void** stream_out);
int evc_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVC_H_ */
