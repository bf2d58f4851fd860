/*
 * C ABI of libicrl_a2c.so: hand-written gfx950 (MI355X) HIP kernels for the
 * Advantage-Actor-Critic caption-training step.
 *
 * The reference (pratikpv/image-captioning-through-rl) is pure Python/PyTorch and
 * has no FFI of its own; each entry point below replaces the PyTorch calls at the
 * cited reference file:line.  Conventions:
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - buffers, including workspaces, are owned by the caller; nothing here
 *     allocates or frees device memory;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - return value: 0 on success, negative ICRL_ERR_* otherwise, with a message
 *     available from icrl_last_error() (per host thread).  No exceptions cross
 *     the boundary;
 *   - one host thread per GPU; matrices are row-major float32, token ids int64;
 *   - H = 512 is the hidden width of every recurrent layer (models.py:41, 94, 189).
 */
#ifndef ICRL_A2C_H
#define ICRL_A2C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* icrl_stream_t;

#define ICRL_OK 0
#define ICRL_ERR_ARG (-1)     /* bad shape, alignment or null pointer */
#define ICRL_ERR_HIP (-2)     /* a HIP runtime call failed */
#define ICRL_ERR_TIMEOUT (-3) /* a chain kernel gave up waiting for a peer workgroup */

#define ICRL_ABI_VERSION 4

int icrl_abi_version(void);
const char* icrl_last_error(void);

/* ---- dense contraction -------------------------------------------------------
 * C[M,N] = (accumulate ? C : 0) + A[M,K] * B[K,N] + bias[n] + bias2[n]
 * on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).  Replaces every nn.Linear and
 * the LSTM's gate products on the path: models.py:75 (cnn2linear), :80 (lstm),
 * :82 (linear2vocab), :177-178 (value head), :259-260 (reward embeds) and their
 * autograd transposes (trainers.py:479).
 *   a_kmajor = 0: A stored [M][lda] (k contiguous); 1: stored [K][lda] (m contiguous)
 *   b_kmajor = 0: B stored [N][ldb] (k contiguous, an nn.Linear weight); 1: [K][ldb]
 *   a_rows: optional int64[M] row gather for A (a_kmajor = 0 only), i.e. the
 *           nn.Embedding lookup of models.py:73 fused into the product.
 *   bias, bias2: optional float[N].
 *   tile: output tile per workgroup, 0 = chosen by shape, 1 = 128x128, 2 = 128x64, 3 = 64x64
 *         (a performance choice only; the caller plans it per shape and per what else runs). */
int icrl_gemm_f32(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                  const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                  int M, int N, int K, const float* bias, const float* bias2,
                  int accumulate, int tile, icrl_stream_t stream);

/* Split-K form for products with a long k and a small output (the weight gradients
 * summed over ~10^5 chain positions; the per-step products of the segment-parallel chain):
 * split z in [0, splits) writes its partial product over its k-range to
 * C_slabs + z*slab_stride; the consumer adds the slabs (fixed order, so results are
 * reproducible run to run, unlike float atomics). */
int icrl_gemm_f32_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                         int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M,
                         int N, int K, int splits, int tile, icrl_stream_t stream);

/* The same two products with bf16 arithmetic: operands and result stay float32 in memory, every operand tile is
 * rounded to bf16 (nearest even) on its way into LDS, products run on v_mfma_f32_32x32x16_bf16 and accumulate
 * in float32.  An opt-in for the policy network's products (models.py:75, 80, 82 and their autograd
 * transposes); the recurrent chains and everything the default path computes stay exact float32.
 * tile: 0 = by shape, 1 = 128x128, 3 = 64x64. */
int icrl_gemm_bf16(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                   const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                   int M, int N, int K, const float* bias, const float* bias2,
                   int accumulate, int tile, icrl_stream_t stream);
int icrl_gemm_bf16_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                          int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M,
                          int N, int K, int splits, int tile, icrl_stream_t stream);

/* The same products once more, at close to f32 accuracy on the bf16 matrix instruction: every element is cut
 * into three bf16 pieces (hi + mid + lo = the f32 value exactly) on its way into LDS and a product is the six
 * piece products of weight 2^-16 and above, summed in f32 (the dropped ones are below 2^-23 of the product).
 * Agrees with icrl_gemm_f32 to f32 rounding error, not bit for bit; about twice its rate on long products.
 * 128x128 tiles (tile 0 or 1; 256 threads, 36 KB of LDS: four workgroups per CU, two beside a persistent chain
 * kernel); no row gather, no second bias.  For the backward
 * pass's products over all (step, row) positions: the two vocabulary-sized ones and the recurrent layers' weight
 * and input gradients (autograd of models.py:80-82, 133 at trainers.py:479). */
int icrl_gemm_f32x3(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                    const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                    int M, int N, int K, const float* bias, const float* bias2,
                    int accumulate, int tile, icrl_stream_t stream);
int icrl_gemm_f32x3_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                           int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M,
                           int N, int K, int splits, int tile, icrl_stream_t stream);

/* C = A . B^T (+ bias) for operands cut once and multiplied many times: the rollout's per-round product of the
 * newest h with [linear2vocab.weight ; lstm.weight_hh] (models.py:80-82 run one column at a time, trainers.py:443-458),
 * whose weights are constant over the minibatch's 19 rounds.  icrl_cut_f16x2 scales each row of x [rows, K] (row
 * stride ld floats) by a power of two - its own, so that the row's largest entry lands in [2^10, 2^11), or
 * fixed_scale > 0 for rows known to lie within [-1, 1] (fixed_scale 4096) - and cuts it into two f16 pieces
 * (x s = hi + mid to 2^-22 relative), stored in the matrix instruction's fragment order: rows in blocks of 32, k in
 * blocks of 16, block (rb, kb) piece p lane l = r % 32 + 32 (k % 16 / 8) at (((rb K/16 + kb) 2 + p) 64 + l) 8 + k % 8
 * halves; descale[r] = 1 / s.  `pieces` holds icrl_f16x2_piece_halves(rows, K) halves (rows rounded up to 32: the rows
 * of the padding are never written and must be zero, so zero the buffer once).  icrl_gemm_f16x2 computes
 * C[m, n] = (sum_k hi.hi + hi.mid + mid.hi) a_descale[m] b_descale[n] + bias[n] on v_mfma_f32_32x32x16_f16 with
 * float32 sums: each element product to about 2^-21, i.e. float32 rounding level for these 512-term sums; not bit
 * for bit the f32 kernel's result.  K % 16 == 0.  128 x 192 tiles, fragments loaded straight from memory (no LDS). */
int64_t icrl_f16x2_piece_halves(int rows, int K);
int icrl_cut_f16x2(const float* x, int64_t ld, int rows, int K, float fixed_scale, void* pieces, float* descale,
                   icrl_stream_t stream);
int icrl_gemm_f16x2(const void* a_pieces, const float* a_descale, const void* b_pieces, const float* b_descale,
                    float* C, int64_t ldc, int M, int N, int K, const float* bias, icrl_stream_t stream);

/* icrl_gemm_f32x3's products once more on two f16 pieces per element: the caller supplies a power-of-two scale for
 * every row of A and of B as the product sees them (a_scale[M], b_scale[N]: constant along k), which brings the row's
 * largest magnitude into [2^14, 2^15); an element is cut on chip into hi + mid of its scaled value (2^-22 relative
 * wherever both are normal f16 numbers), a product is hi.hi + hi.mid + mid.hi on v_mfma_f32_32x32x16_f16 with float32
 * sums, and the epilogue divides the scales out (exactly).  Half the matrix instructions and two thirds of the LDS
 * traffic of the three-piece kernel, 1.45-1.6 times its rate on the backward pass's shapes; accuracy of the same
 * class (2^-21 per element product).  Same layouts, tiles and split-K slabs as icrl_gemm_f32x3; no row gather, one
 * bias.  For the same products (autograd of models.py:80-82, 133 at trainers.py:479).
 * icrl_absmax_scales computes such scales for x [rows, cols] (row stride ld): row_scale[rows] and / or
 * col_scale[cols] (either may be null); `work` holds rows + cols 32-bit words.  icrl_g0_row_scales gives them for the
 * rows of G0 = onehot(token) - softmax from the rows' log-probabilities alone (the largest magnitude of such a row is
 * 1 - p_token). */
int icrl_gemm_f16x2s(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B, int64_t ldb,
                     int b_kmajor, const float* b_scale, float* C, int64_t ldc, int M, int N, int K,
                     const float* bias, int accumulate, icrl_stream_t stream);
int icrl_gemm_f16x2s_splitk(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B,
                            int64_t ldb, int b_kmajor, const float* b_scale, float* C_slabs, int64_t ldc,
                            int64_t slab_stride, int M, int N, int K, int splits, icrl_stream_t stream);
/* icrl_gemm_f16x2s_splitk with the tile shape chosen by the caller: tile 0 or 1 = 128 x 128 on 256 threads, 2 = 256 x 256
   on 512 threads (one workgroup per CU; per element product half the operand bytes and half the cutting: the two
   vocabulary-sized products of the backward pass, trainers.py:479 through models.py:82).  Same results to float32
   rounding (the k order within a split is the same; bit-equal to the 128 x 128 tiles for the same splits). */
int icrl_gemm_f16x2s_splitk_tile(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B,
                                 int64_t ldb, int b_kmajor, const float* b_scale, float* C_slabs, int64_t ldc,
                                 int64_t slab_stride, int M, int N, int K, int splits, int tile, icrl_stream_t stream);
int icrl_absmax_scales(const float* x, int64_t ld, int rows, int cols, float* row_scale, float* col_scale,
                       unsigned* work, icrl_stream_t stream);
int icrl_g0_row_scales(const float* log_prob, int count, float* scale, icrl_stream_t stream);

/* out[i] = (accumulate ? out[i] : 0) + sum_z slabs[z*slab_stride + i], i < count (count % 4 == 0) */
int icrl_sum_slabs(const float* slabs, int64_t slab_stride, int splits, int64_t count, float* out,
                   int accumulate, icrl_stream_t stream);

/* out[r, :] = table[idx[r*idx_stride], :]   (nn.Embedding, models.py:73, 132, 225) */
int icrl_gather_rows(const float* table, int width, const int64_t* idx, int64_t idx_stride,
                     int rows, float* out, icrl_stream_t stream);

/* grad_table[idx[r*idx_stride], :] += rows_in[r, :]  (Embedding backward, float atomics) */
int icrl_scatter_add_rows(float* grad_table, int width, const int64_t* idx, int64_t idx_stride,
                          int rows, const float* rows_in, icrl_stream_t stream);

/* The same as an overwrite, reproducible bit for bit (the reference's CPU index_add is; float atomics are
 * not), in two calls.  icrl_token_order lists, for every token v in [0, vocab), the rows r with
 * idx[r*idx_stride] == v in ascending r - exact integer work (per-token counts, their exclusive scan,
 * per-token row lists) kept in `work`, icrl_embedding_grad_work_ints(vocab, rows) int32.  It depends on the
 * tokens only, so one call serves every embedding table looked up with the same tokens, and it can run as
 * soon as they exist.  icrl_embedding_grad then writes grad_table[v, :] = the sum of rows_in[r, :] over v's
 * rows in that order (zeros for tokens no row holds).  Embedding backward of models.py:73, 132 through
 * trainers.py:479. */
int64_t icrl_embedding_grad_work_ints(int vocab, int rows);
int icrl_token_order(const int64_t* idx, int64_t idx_stride, int rows, int vocab, int32_t* work,
                     icrl_stream_t stream);
int icrl_embedding_grad(float* grad_table, int vocab, int width, const float* rows_in, const int32_t* work,
                        icrl_stream_t stream);

/* out[n] (+)= sum_r row_scale[r] * x[r, n]   (bias gradients; row_scale may be NULL = ones).  scratch (may be
 * NULL): icrl_colsum_scratch_floats(rows, cols) floats; with it the row blocks' partial sums are added in a
 * fixed order (reproducible bit for bit), without it by float atomics. */
int64_t icrl_colsum_scratch_floats(int rows, int cols);
int icrl_colsum(const float* x, int64_t ldx, int rows, int cols, float* out, int accumulate,
                const float* row_scale, float* scratch, icrl_stream_t stream);

/* out[r, :] = coef[r] * x[r, :]: the policy-gradient coefficient dLoss/dlog_prob applied to rows whose
 * products with the vocabulary matrix were formed before the loss was known (cols % 4 == 0). */
int icrl_scale_rows(const float* x, int64_t ldx, const float* coef, float* out, int64_t ldo, int rows, int cols,
                    icrl_stream_t stream);

/* ---- batched LSTM cell (policy network, models.py:80) -----------------------
 * gates_pre[N,4H] holds x*W_ih^T + h*W_hh^T + b_ih + b_hh, PyTorch order i,f,g,o.
 * fwd: c = sig(f)*c_prev + sig(i)*tanh(g); h = sig(o)*tanh(c); the activated
 *      gates are written back over gates_pre for the backward pass.
 * bwd: given dh, dc (dc may be null = 0) and the saved activations, writes
 *      dgates[N,4H] (w.r.t. the pre-activations) and dc_prev. */
int icrl_lstm_cell_fwd(float* gates, const float* c_prev, float* c, float* h, int N,
                       icrl_stream_t stream);
/* Pointwise part of nn.GRU for N rows at once (models.py:226 run with one recurrent state per row):
 * gx = x W_ih^T + b_ih and gh = h_prev W_hh^T + b_hh, both [N, 3*512] in gate order r, z, n;
 * h = (1 - z) n + z h_prev with r = sigmoid(xr + hr), z = sigmoid(xz + hz), n = tanh(xn + r hn). */
int icrl_gru_cell_fwd(const float* gx, const float* gh, const float* h_prev, float* h, int N,
                      icrl_stream_t stream);
int icrl_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c,
                       const float* dh, const float* dc, float* dgates, float* dc_prev, int N,
                       icrl_stream_t stream);

/* The policy LSTM's backward-through-time over the T steps of a rollout (autograd of models.py:80,
 * trainers.py:479) as one call: for k = T-1 .. 0, icrl_lstm_cell_bwd on step k (gates_act [T,N,4H], c_all
 * [T+1,N,H] = the cell state before step 0 and after every step, dH [T,N,H] = the gradient that reaches each
 * step's output from the vocabulary projection), then dH[k-1] += dgates[k] . W_hh, and for k = 0
 * dh0 = dgates[0] . W_hh (the gradient of the state the image features initialise).  dgates [T,N,4H] is
 * written for the weight-gradient products; dc0 / dc1 [N,H] carry the cell-state gradient.  The products run
 * as icrl_gemm_f32_splitk + icrl_sum_slabs (splits > 1; scratch >= splits*N*512 floats) or icrl_gemm_f32, the
 * bf16 variants if bf16 != 0: the same launches in the same order as a caller issuing them one by one, queued
 * without a Python host's per-launch cost. */
int icrl_policy_lstm_bptt(const float* gates_act, const float* c_all, float* dH, const float* w_hh,
                          float* dgates, float* dc0, float* dc1, float* dh0, float* scratch,
                          int64_t scratch_floats, int N, int T, int splits, int tile, int bf16,
                          icrl_stream_t stream);
/* The same backward-through-time (autograd of models.py:80, trainers.py:479) as ONE persistent launch: the N rows
 * are N independent chains of T steps with a gradient injected at every step, which is the per-row chain
 * geometry of icrl_lstm_chain_bwd_rows with every column the last column of a call of its own; the launch walks
 * T + 1 steps, dgates handed between its workgroups as in icrl_lstm_chain_bwd_seg_persist (exact float32
 * products).  d_h [T,N,H] is read only (the running dH of step k lives inside the launch); dgates [T,N,4H] and
 * dh0 [N,H] as for icrl_policy_lstm_bptt.  work: icrl_chain_seg_workspace_floats(N) floats.  *status != 0
 * afterwards: the launch gave up waiting for a workgroup (results unusable).  Needs icrl_chain_pair_supported(N). */
int icrl_policy_lstm_bptt_persist(const float* gates_act, const float* c_all, const float* d_h, const float* w_hh,
                                  int N, int T, float* dgates, float* dh0, float* work, int* status,
                                  icrl_stream_t stream);
/* The backward chains above as an owner-computes launch on the f16 matrix instruction: dh = dgates . W_hh sums over
 * all 2048 gate rows, and a workgroup multiplies only the 128 gate gradients it produced itself by its rows of
 * W_hh (partial[32 segments][512 units]), hands every workgroup of its row group the 32 x 32 tile of that one's
 * units and adds up the 16 tiles it is sent: half the exchanged bytes of the gathering launch, and the gate
 * gradients never leave the workgroup, so they are cut into two f16 pieces with a power-of-two scale per segment
 * row (W_hh's slice: one scale per wave, cut once, kept in registers); products hi.hi + hi.mid + mid.hi with
 * float32 sums, 2^-21 per element product, the exchanged partial sums float32.  Same arguments, outputs and seam
 * errors as icrl_lstm_chain_bwd_seg_persist / icrl_policy_lstm_bptt_persist (trainers.py:479's autograd through
 * models.py:133 / :80); work: icrl_chain_bwd_f16x2_workspace_floats(segments or N) floats.  Results agree with the
 * float32 launches to float32 rounding, not bit for bit.  icrl_chain_bwd_f16x2_supported: whether it fits the device. */
int64_t icrl_chain_bwd_f16x2_workspace_floats(int segments);
int icrl_chain_bwd_f16x2_supported(int segments);
int icrl_lstm_chain_bwd_seg_f16x2(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                  int n_calls, int full_calls, int segments, int warmup, float* dgates, float* work,
                                  float* seam_err, icrl_stream_t stream);
int icrl_policy_lstm_bptt_f16x2(const float* gates_act, const float* c_all, const float* d_h, const float* w_hh,
                                int N, int T, float* dgates, float* dh0, float* work, int* status,
                                icrl_stream_t stream);


/* One whole step of the policy LSTM in a single launch (models.py:73 + :80 for one position):
 * x = emb[tokens[i]], gates = x*W_ih^T + h_cur*W_hh^T + b_ih + b_hh, then the cell.  Writes the
 * activated gates [N,4H] (for the backward pass), c_next, h_next and, if x_out is given, the
 * gathered embedding rows [N,wordvec_dim].  wordvec_dim must be a multiple of 64. */
int icrl_policy_lstm_step(const float* emb, int wordvec_dim, const int64_t* tokens, const float* w_ih,
                          const float* w_hh, const float* b_ih, const float* b_hh, const float* h_cur,
                          const float* c_cur, float* gates_act, float* c_next, float* h_next,
                          float* x_out, int N, icrl_stream_t stream);

/* ---- batched autoregressive sampler (trainers.py:444-458, :69) ---------------
 * One workgroup per row of logits[N,V]:
 *   mode ICRL_SAMPLE: softmax in f32, float64 CDF, token = #{k : cdf[k]/cdf[V-1] <= u}
 *        with ONE uniform double per row - numpy.random.choice(V, p=...) given the
 *        same draw; log_prob = log(p[token]).
 *   mode ICRL_GREEDY: token = argmax logits, first index wins (torch.argmax).
 *   mode ICRL_FORCED: token is read from tokens_out; only log_prob is produced.
 * tokens_out[i*token_stride] receives / holds the int64 token.  probs (optional,
 * [N,V]) receives the softmax; entropy (optional, [N]) the diagnostic
 * -sum p log p, which the reference's loss does NOT contain.
 * mode | ICRL_EMIT_DLOGITS (with ICRL_SAMPLE or ICRL_FORCED): probs receives onehot(token) - softmax
 * instead, i.e. d log_prob / d logits (autograd of trainers.py:458), what icrl_policy_dlogits would make of the
 * softmax with coef = 1, bit for bit - written by the launch that has the row on chip anyway. */
#define ICRL_SAMPLE 0
#define ICRL_GREEDY 1
#define ICRL_FORCED 2
#define ICRL_EMIT_DLOGITS 16
int icrl_sample_tokens(const float* logits, int64_t ld_logits, int N, int V, int mode,
                       const double* uniforms, int64_t* tokens_out, int64_t token_stride,
                       float* log_prob, int64_t log_prob_stride, float* probs, float* entropy,
                       icrl_stream_t stream);

/* icrl_sample_tokens (modes ICRL_SAMPLE and ICRL_FORCED) followed, in the same launch, by the policy LSTM's
 * NEXT cell update for every row (models.py:73, 80 for the token just fixed): the workgroup that fixed row i's
 * token w forms gates = pre[i] + table[w], applies the cell to c_prev[i] and writes the activated gates
 * (i, f, g, o; [N,2048]), c_out and h_out ([N,512]).  pre [N, >= 2048] (row stride ld_pre) = h . W_hh^T of the
 * state the logits were computed from, without bias; table [V,2048] = W_ih emb[w] + b_ih + b_hh for every
 * word w.  One launch less, and no matrix product, between a token and the next step's logits.  h_pieces
 * (optional) receives the new h once more as the operand of icrl_gemm_f16x2: icrl_cut_f16x2's layout for [N, 512]
 * with fixed_scale 4096 (the descale vector is then 2^-12 in every entry). */
int icrl_sample_tokens_lstm_cell(const float* logits, int64_t ld_logits, int N, int V, int mode,
                                 const double* uniforms, int64_t* tokens_out, int64_t token_stride,
                                 float* log_prob, int64_t log_prob_stride, float* probs, float* entropy,
                                 const float* pre, int64_t ld_pre, const float* table, float* gates_out,
                                 const float* c_prev, float* c_out, float* h_out, void* h_pieces,
                                 icrl_stream_t stream);

/* `rounds` consecutive rollout rounds (models.py:73-82 one column at a time; trainers.py:443-458) queued by one call:
 * round r multiplies the newest h (in h_pieces, where the round before left it; the caller cuts the first one) by
 * the pieces of [linear2vocab.weight (rows padded to pre_col) ; lstm.weight_hh] with icrl_gemm_f16x2 into
 * ext [N, pre_col + 2048] (row stride ld_ext), then runs icrl_sample_tokens_lstm_cell on it: token, log-probability,
 * optional probabilities / entropy, and the next cell update.  Arrays advance by one round per round: uniforms,
 * tokens, log_prob, entropy by N; probs by N V; gates (receives the next cell's activated gates) by N 2048; c_state
 * holds the cell state the round starts from at r N 512 and receives the next at (r + 1) N 512; h_state receives the
 * next h at r N 512.  Exactly the launches the caller would queue one by one, without a host call per launch (at 25
 * + 20 us per round the host, not the device, was what paced the rollout). */
int icrl_policy_rollout_rounds(const void* w_pieces, const float* w_descale, const float* bias, void* h_pieces,
                               const float* h_descale, float* ext, int64_t ld_ext, int N, int V, int pre_col,
                               int rounds, int mode, const double* uniforms, int64_t* tokens, float* log_prob,
                               float* probs, float* entropy, const float* table, float* gates,
                               const float* c_state, float* h_state, icrl_stream_t stream);

/* values[r, 0..k) / indices[r, 0..k) = the k largest entries of logits[r, 0..V) in descending order,
 * ties by lowest index: torch.topk(output[:, -1:, :], beamSize) of the look-ahead decoder,
 * trainers.py:96.  k <= 64, V <= 16384. */
int icrl_topk_rows(const float* logits, int64_t ld_logits, int N, int V, int k, float* values,
                   int64_t* indices, icrl_stream_t stream);

/* d logits for the policy-gradient term from the saved softmax (dlogits may alias probs):
 * g[r, k] = coef[r] * ((k == token[r]) - p[r, k]),  coef = dLoss/dlog_prob; coef NULL = ones (the
 * coefficient is then applied later, by icrl_scale_rows / icrl_colsum's row_scale). */
int icrl_policy_dlogits(const float* probs, float* dlogits, int rows, int V, const int64_t* tokens,
                        int64_t token_stride, const float* coef, icrl_stream_t stream);

/* ---- batch-1 recurrent chains (models.py:130-135, 166-169; :223-228, 253-255) -
 * The reference feeds (N,1,E) to an nn.LSTM / nn.GRU built without batch_first,
 * so the N rows of a minibatch are consecutive time steps of ONE recurrence whose
 * state also carries across columns and across successive forward() calls.
 * Position order: for call s in [0, n_calls): for column t in [0, p0+s):
 * for row i in [0, N); p0 is the number of caption columns the FIRST call sees (the
 * reward network also sees the newly sampled token, trainers.py:454-459, so its caller
 * passes one more than the value network's).  xproj[(t*N + i), G*H] holds
 * W_ih*emb[token[i,t]] + b_ih
 * (+ b_hh for the LSTM), G = 4 (LSTM, gates i,f,g,o) or 3 (GRU, gates r,z,n).
 * The kernel is persistent: ICRL_CHAIN_WGS workgroups each keep a slice of W_hh in
 * registers and exchange the 512-float state once per step through tagged 8-byte
 * granules in `sync_ws` (icrl_chain_sync_bytes(), zeroed by the call itself).
 * h_emit[s, i, :] receives the hidden output at the last column of call s.
 * For the LSTM, `saved` (optional) receives per position {i,f,g,o,c,h} = 6*H floats
 * for the backward pass; for the GRU {r, z, n, W_hn.h_prev + b_hn, h} = 5*H floats.  state_in (optional, {h, c} = 2*H floats, GRU: h only) is the
 * carried hidden state (models.py:127-128, 133: hidden_cell survives across forward()
 * calls until init_hidden()); null = zeros.  state_out (optional) receives the final
 * state.  status: one int32 the kernel sets non-zero on time-out. */
#define ICRL_CHAIN_WGS 64
int64_t icrl_chain_sync_bytes(void);
int icrl_lstm_chain_fwd(const float* xproj, const float* w_hh, int N, int p0, int n_calls,
                        float* h_emit, float* saved, const float* state_in, float* state_out,
                        void* sync_ws, int32_t* status, icrl_stream_t stream);
int icrl_gru_chain_fwd(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                       int n_calls, float* h_emit, float* saved, const float* state_in, float* state_out,
                       void* sync_ws, int32_t* status, icrl_stream_t stream);
/* Backward of icrl_lstm_chain_fwd (autograd through models.py:133, trainers.py:479):
 * walks the positions in reverse; d_emit[s,i,:] is dLoss/dh_emit; writes
 * dgates[pos, 4H] (gradient w.r.t. every position's gate pre-activations).  The
 * forward pass must have started from a zero state (as every A2C minibatch does,
 * trainers.py:495-496). */
int icrl_lstm_chain_bwd(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                        int n_calls, float* dgates, void* sync_ws, int32_t* status,
                        icrl_stream_t stream);
/* Backward of icrl_gru_chain_fwd (autograd through models.py:226 in the reward pre-training loop,
 * trainers.py:288-300): dgates_x[pos, 3H] = gradient w.r.t. the input-side pre-activations (r, z, n),
 * dgates_h[pos, 3H] = w.r.t. the hidden-side ones (they differ in the n slot: n = tanh(x_n + r * h_n)).
 * The forward pass must have started from a zero state. */
int icrl_gru_chain_bwd(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                       int n_calls, float* dgates_x, float* dgates_h, void* sync_ws, int32_t* status,
                       icrl_stream_t stream);
/* ---- the same chains, segment-parallel with a seam check ------------------------
 * A gated recurrence forgets its initial state, so the chain is cut into `segments`
 * pieces that run side by side: each piece first runs `warmup` extra steps over the
 * positions before its start (after its end, for the backward pass) from a zero state.
 * Every step of all pieces is one [segments,512] x W_hh product on the matrix cores plus
 * a fused cell kernel.  Nothing is assumed about the forgetting rate: the state each
 * piece reached after its warm-up is compared with the state its neighbour really ended
 * in, and the worst mismatch is reported in seam_err (forward: seam_err[0] = max
 * |a-b|/max(1,|b|) over h and c; backward: seam_err[0] = max |a-b|, seam_err[1] = max |b|
 * over dh and dc).  The caller accepts the result only when that is at rounding level and
 * otherwise reruns with a longer warm-up or the exact kernels above.  Same inputs,
 * outputs and layouts as icrl_lstm_chain_fwd / icrl_gru_chain_fwd / icrl_lstm_chain_bwd
 * (zero initial state; icrl_lstm_chain_fwd_seg also takes state_in = {h, c}, 2*512 floats, or NULL: the
 * state the chain starts from, as ValueNetworkRNN.hidden_cell carries it from call to call,
 * models.py:133-134); work = icrl_chain_seg_workspace_floats(segments) floats.
 *
 * full_calls (0 = n_calls = every call in full).  Call s of the reference walks caption columns
 * 0 .. p0+s-1 (models.py:168-169, 254-255), so call s+1 begins by REPLAYING every input of call s, from
 * the state call s ended in, before it reads its one new column.  A recurrence that forgets ends that
 * replay in the state it started from, and the replay can then be left out: with full_calls = f the chain
 * is calls 0 .. f-1 in full followed by the LAST column only of calls f .. n_calls-1, i.e.
 * N * (f*p0 + f(f-1)/2 + n_calls - f) positions instead of N * (n_calls*p0 + n_calls(n_calls-1)/2);
 * `saved` and `dgates` are indexed by the positions of that shorter chain, h_emit / d_emit are unchanged.
 * The hidden outputs agree with the full chain's, and the gate gradients with the full chain's summed
 * over the calls that contain a column (icrl_chain_fold_calls), to the accuracy with which the replays
 * return to their start state - which the caller must check with icrl_chain_pair_replay_check, and the
 * backward pass by requiring its accepted warm-up to be shorter than the shortest replay left out. */
int64_t icrl_chain_seg_workspace_floats(int segments);
int icrl_lstm_chain_fwd_seg(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                            int segments, int warmup, float* h_emit, float* saved, const float* state_in,
                            float* work, float* seam_err, icrl_stream_t stream);
/* icrl_lstm_chain_fwd_seg for callers that cannot allocate or carry a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * Same arguments minus `work`, same kernels and the same results bit for bit; the segment workspace is a
 * static device array inside the library (4 MiB per GPU, held until the process ends), so segments is
 * limited to 128 and larger values are rejected with ICRL_ERR_ARG.  The call makes no hipMalloc / hipFree
 * (its first call on a device looks the array's address up once, which may load the code object: make one
 * call on each device before starting a stream capture there).  The array used is the CURRENT device's, and
 * `stream` must belong to that device (ICRL_ERR_ARG otherwise).  Every call on one GPU shares that
 * array: calls that overlap on one GPU (two streams, two host threads) race; calls in stream order are
 * safe, whatever their sizes. */
int icrl_lstm_chain_fwd_seg_static(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                                   int segments, int warmup, float* h_emit, float* saved,
                                   const float* state_in, float* seam_err, icrl_stream_t stream);
/* saved (may be NULL): the GRU chain's activations {r, z, n, hn, h}, 5*512 floats per position, as
 * icrl_gru_chain_fwd keeps them for icrl_gru_chain_bwd. */
int icrl_gru_chain_fwd_seg(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                           int n_calls, int full_calls, int segments, int warmup, float* h_emit, float* saved,
                           float* work, float* seam_err, icrl_stream_t stream);
/* icrl_gru_chain_bwd (the BPTT through RewardNetworkRNN's recurrence, trainers.py:288-300) evaluated
 * segment-parallel: same inputs, outputs and layouts; seam_err as for icrl_lstm_chain_bwd_seg
 * (seam_err[0] = max |a-b|, seam_err[1] = max |b| over the product W_hh^T dgates_h and the carried
 * dh*z at the seams). */
int icrl_gru_chain_bwd_seg(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                           int n_calls, int segments, int warmup, float* dgates_x, float* dgates_h,
                           float* work, float* seam_err, icrl_stream_t stream);
int icrl_lstm_chain_bwd_seg(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                            int n_calls, int full_calls, int segments, int warmup, float* dgates, float* work,
                            float* seam_err, icrl_stream_t stream);

/* Both forward chains of one minibatch (value LSTM: models.py:133, 140-143; reward GRU:
 * models.py:226, 231-233) in ONE persistent launch: same results as icrl_lstm_chain_fwd_seg
 * followed by icrl_gru_chain_fwd_seg with the same segments and warmup_v / warmup_r (each chain
 * has its own warm-up length: the two networks forget at different rates), bit for bit, without a
 * kernel boundary per step.  Its workgroups (16 per 32 segments) hand the hidden state to each
 * other inside the launch and must all be resident: icrl_chain_pair_supported(segments) says
 * whether the current device has enough CUs, and the call fails otherwise.  p0_v / p0_r =
 * columns of the first call of each chain (the reward chain sees one more column).  If a
 * workgroup gives up waiting for a peer (20 ms), both seam errors are set to +inf so that
 * the caller rejects the result like any failed seam check.  Do not run two of these
 * launches at once on one device. */
int icrl_chain_pair_supported(int segments);
int icrl_chain_pair_fwd_seg(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                            const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                            int n_calls, int full_calls, int segments, int warmup_v, int warmup_r, float* h_emit_v,
                            float* saved_v, float* h_emit_r, float* saved_r, float* work_v, float* work_r,
                            float* seam_err_v, float* seam_err_r, icrl_stream_t stream);

/* icrl_chain_pair_fwd_seg with the first cheap_v / cheap_r steps of each chain's warm-up (0 <= cheap < warmup)
 * on the bf16 matrix instruction with both operands cut in two bf16 pieces (16 significant bits; the products
 * hi.hi + hi.mid + mid.hi), a launch of its own before the float32 launch, which continues from the state it
 * leaves.  A warm-up's results are discarded and the seam errors measure the state it ends in, exactly as
 * without this; the float32 steps that remain (warmup - cheap; 16 or more in practice) wash the 2^-16 out.
 * No counterpart in the reference: it shortens work that the segment-parallel evaluation of
 * models.py:133, 226 adds.  The 16-bit launch holds every CU's registers while it runs, so a caller with work
 * on other streams may want it to start only afterwards: phase 1 queues everything up to and including that
 * launch, phase 2 (same arguments, same workspaces) the float32 launch and the seam checks; phase 0 both.
 * icrl_chain_warm16_supported(segments): whether the 16-bit launch fits the device. */
int icrl_chain_warm16_supported(int segments);
int icrl_chain_pair_fwd_seg_warm16(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                   const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                                   int n_calls, int full_calls, int segments, int warmup_v, int warmup_r,
                                   int cheap_v, int cheap_r, int phase, float* h_emit_v, float* saved_v,
                                   float* h_emit_r, float* saved_r, float* work_v, float* work_r,
                                   float* seam_err_v, float* seam_err_r, icrl_stream_t stream);
/* icrl_chain_pair_fwd_seg (both forward chains of models.py:133, 226, every step, one persistent launch) with the
 * recurrent products on the f16 matrix instruction at float32-level accuracy: every operand is scaled by a power
 * of two (h by 2^12, each row of W_hh by the one that brings its largest entry into [2^10, 2^11)) and cut into two
 * f16 pieces, x s = hi + mid to 2^-22, and a product is hi.hi + hi.mid + mid.hi with float32 sums - what is
 * dropped is below 2^-21 of each element product, the size of float32's rounding of the 512-term sum.  W_hh stays
 * in registers for the whole launch, so a step moves only the exchanged h.  Same arguments, outputs, seam errors
 * and workspaces as icrl_chain_pair_fwd_seg; results agree with it to float32 rounding, not bit for bit.
 * icrl_chain_f16x2_supported(segments): whether the launch fits the device. */
int icrl_chain_f16x2_supported(int segments);
int icrl_chain_pair_fwd_seg_f16x2(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                  const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r, int n_calls,
                                  int full_calls, int segments, int warmup_v, int warmup_r, float* h_emit_v,
                                  float* saved_v, float* h_emit_r, float* saved_r, float* work_v, float* work_r,
                                  float* seam_err_v, float* seam_err_r, icrl_stream_t stream);


/* The check that goes with full_calls < n_calls (no counterpart in the reference, which always replays):
 * for every call s >= full_calls, run the first `steps` positions of the replay that was left out, from the
 * state the chain had at the end of call s-1 (taken from saved_v / saved_r, the activations the forward
 * launch kept: LSTM {i,f,g,o,c,h}, GRU {r,z,n,hn,h} per position of the shortened chain), all calls side by
 * side in one persistent launch, and compare the state reached with the state the last fully walked call
 * had after the same inputs.  err[0] (value LSTM, h and c) and err[1] (reward GRU, h) receive
 * max |a-b| / max(1,|b|); +inf if the launch gave up waiting.  At rounding level the replays may be left
 * out; otherwise rerun the chains with a larger full_calls.  steps <= N * (p0 + full_calls - 1), the
 * shortest replay left out.  work_v / work_r: icrl_chain_seg_workspace_floats(n_calls - full_calls)
 * floats each, not shared with a launch that may run at the same time. */
int icrl_chain_pair_replay_check(const float* xproj_v, const float* w_hh_v, const float* saved_v,
                                 const float* xproj_r, const float* w_hh_r, const float* b_hh_r,
                                 const float* saved_r, int N, int p0_v, int p0_r, int n_calls, int full_calls,
                                 int steps, float* work_v, float* work_r, float* err, icrl_stream_t stream);
/* The same check with the replays walked by the f16 launch of icrl_chain_pair_fwd_seg_f16x2 (two scaled f16 pieces
 * per operand, float32-level accuracy; about half the time per step): for chains that were walked with it. */
int icrl_chain_pair_replay_check_f16x2(const float* xproj_v, const float* w_hh_v, const float* saved_v,
                                       const float* xproj_r, const float* w_hh_r, const float* b_hh_r,
                                       const float* saved_r, int N, int p0_v, int p0_r, int n_calls, int full_calls,
                                       int steps, float* work_v, float* work_r, float* err, icrl_stream_t stream);

/* The give-up path of the persistent launches, on its own (for the test-suite; no counterpart in the reference).
 * Every wait inside a persistent launch is bounded by 20 ms of wall clock; the first wave whose wait runs out raises
 * the launch's abort word, which releases every other waiter, and records next to it who it is (abort_words[1]:
 * workgroup id), the step it waited at ([2]) and which producers of its row group had not reached that step ([3],
 * bit i: producer i).  The kernel that follows every persistent launch turns a raised word into infinite seam errors
 * (or a status word) and prints one line with those three numbers.  This entry point runs one such wait (16 flags,
 * or 32 with `wide`) for `epoch` on caller-supplied flags - which never reach it unless the caller has set them - and
 * then that follow-up kernel: err[0:2] = +inf if the wait ran out.  flags: 32 words; abort_words: 4 words, zeroed by
 * the caller. */
int icrl_chain_give_up_selftest(unsigned* flags, unsigned* abort_words, unsigned epoch, int wide, float* err,
                                icrl_stream_t stream);

/* One forward chain alone in the same persistent launch (the look-ahead decoder's value calls,
 * trainers.py:73-105, and the pre-training loops, trainers.py:125-199, run one chain at a time):
 * icrl_lstm_chain_fwd_seg / icrl_gru_chain_fwd_seg (models.py:133, 140-143 / 226, 231-233) with the same
 * arguments and the same results bit for bit, without a kernel boundary per step; state_in needs
 * warmup >= 1.  Needs icrl_chain_pair_supported(segments); a
 * hand-off time-out sets seam_err[0] = +inf. */
int icrl_lstm_chain_fwd_seg_persist(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                                    int segments, int warmup, float* h_emit, float* saved,
                                    const float* state_in, float* work, float* seam_err,
                                    icrl_stream_t stream);
int icrl_gru_chain_fwd_seg_persist(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                                   int n_calls, int full_calls, int segments, int warmup, float* h_emit, float* saved,
                                   float* work, float* seam_err, icrl_stream_t stream);

/* icrl_lstm_chain_bwd_seg (the BPTT of models.py:140-143's recurrence) as ONE persistent
 * launch: same arguments, same results bit for bit, no kernel boundary per step.  Needs
 * icrl_chain_pair_supported(segments); a hand-off time-out sets seam_err[0] = +inf. */
int icrl_lstm_chain_bwd_seg_persist(const float* w_hh, const float* saved, const float* d_emit, int N,
                                    int p0, int n_calls, int full_calls, int segments, int warmup, float* dgates,
                                    float* work, float* seam_err, icrl_stream_t stream);

/* icrl_lstm_chain_bwd_seg_persist with the first `cheap` steps of the warm-up (0 <= cheap <= warmup) on the
 * bf16 matrix instruction, both operands of dh = dgates . W_hh cut in two bf16 pieces (16 significant bits; the
 * products hi.hi + hi.mid + mid.hi), in a launch of its own before the float32 launch, which continues from the
 * dgates and dc it leaves.  The backward warm-up's results are discarded like the forward one's and seam_err
 * measures the (dh, dc) it ends in, exactly as without this.  No counterpart in the reference: it shortens work
 * that the segment-parallel evaluation of models.py:133's BPTT adds.  The exchanged dgates travel already cut
 * in two (bf16 planes in the bytes of the float32 exchange buffers), so segments must be a multiple of 32;
 * icrl_chain_bwd_warm16_supported(segments) says whether that holds and the 16-bit launch fits the device. */
int icrl_chain_bwd_warm16_supported(int segments);
int icrl_lstm_chain_bwd_seg_warm16(const float* w_hh, const float* saved, const float* d_emit, int N,
                                   int p0, int n_calls, int full_calls, int segments, int warmup, int cheap,
                                   float* dgates, float* work, float* seam_err, icrl_stream_t stream);

/* out[(t*N+i), :] = sum over calls s that contain column t of dgates[pos(s,t,i), :] (positions of the
 * chain with `full_calls` calls in full; with full_calls = 1 and p0 = 1 every column occurs once) */
int icrl_chain_fold_calls(const float* dgates, int N, int p0, int n_calls, int full_calls, int width, float* out,
                          icrl_stream_t stream);
/* h_prev[pos, :] = h of position pos-stride (zeros for pos < stride), taken from `saved` (gru = 0:
 * the LSTM chain's 6 slots per position; 1: the GRU chain's 5).  stride 1: the batch-1 chains;
 * stride N: the per-row chains below. */
int icrl_chain_hprev(const float* saved, int64_t positions, int gru, int64_t stride, float* h_prev,
                     icrl_stream_t stream);

/* ---- per-row chains -----------------------------------------------------------
 * ValueNetworkRNN / RewardNetworkRNN (models.py:130-135, 223-228) as the reference runs them at
 * batch size 1, for N images at once: row i walks its own columns, call after call (the state is
 * kept across the calls of a minibatch, trainers.py:495-496), from a zero state and independently
 * of the other rows.  An explicit change of semantics for batches of more than one row, where the
 * reference's recurrence runs through all rows (see icrl_lstm_chain_fwd); never a default.
 * Arguments, per-position layouts (saved, h_emit, dgates) and workspace
 * (icrl_chain_seg_workspace_floats(N) floats each) as for the segmented entry points with one
 * segment per row; N <= 4096.  persistent = 1: all steps in one launch (needs
 * icrl_chain_pair_supported(N)), 0: one launch per step; same results bit for bit.  status[0] is
 * set to 0, or to 1 when a persistent launch gave up waiting for a peer workgroup (results are
 * then invalid); check it after synchronising. */
int icrl_chain_pair_fwd_rows(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                             const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                             int n_calls, int persistent, float* h_emit_v, float* saved_v,
                             float* h_emit_r, float* work_v, float* work_r, int* status,
                             icrl_stream_t stream);
/* BPTT of the per-row value chains: dgates[pos, 4H] from d_emit [n_calls, N, H] and `saved`. */
int icrl_lstm_chain_bwd_rows(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                             int n_calls, int persistent, float* dgates, float* work, int* status,
                             icrl_stream_t stream);

/* ---- value head (models.py:175-178) ------------------------------------------
 * y[r,:] = yh[r,:] + u[r % N,:]  (u = W1[:, :F]*feat + b1, yh = W1[:, F:]*h_emit);
 * value[r] = y[r,:].w2 + b2.  No non-linearity between the two layers. */
int icrl_value_head_fwd(float* yh_inout, const float* u, const float* w2, const float* b2,
                        int rows, int N, float* values, icrl_stream_t stream);

/* ---- embedding-cosine reward (trainers.py:116-120) ---------------------------
 * reward[r] = <ve[r % N]/max(|ve|,1e-12), se[r]/max(|se|,1e-12)>, width 512. */
int icrl_cosine_reward(const float* ve, const float* se, int rows, int N, float* reward,
                       icrl_stream_t stream);

/* ---- advantage / policy-gradient loss (trainers.py:471-476) -------------------
 * Inputs are [steps*N] in (step,row) order.  advantage = value - reward;
 * loss = mean(-logp*adv) + 0.5*mean(adv^2).  scalars[0..4] = loss, actor, critic,
 * mean reward, mean advantage.  d_value and d_logp receive dLoss/dvalue and
 * dLoss/dlog_prob (the advantage is not detached in the reference). */
int icrl_a2c_loss(const float* values, const float* rewards, const float* log_probs, int count,
                  float* scalars, float* d_value, float* d_logp, icrl_stream_t stream);

/* VisualSemanticEmbeddingLoss (trainers.py:21-54) from S = visuals . semantics^T [N, N]:
 * loss[0] = (sum_ij relu(S_ij - S_ii + m_ij) + sum_ij relu(S_ji - S_ii + m_ij)) / N with
 * m_ij = (beta / N) * (i != j); dS (optional) = dLoss/dS.  row_scratch: N floats. */
int icrl_vse_loss(const float* S, int N, float beta, float* row_scratch, float* loss, float* dS,
                  icrl_stream_t stream);

/* Length-weighted teacher-forced cross entropy of the policy pre-training loop
 * (trainers.py:238-242): loss[0] = sum_i row_weight[i] * sum_{t < valid_steps[i]} -log_probs[t*N+i];
 * d_logp (optional) [steps*N] = -row_weight[i] inside the mask, 0 outside. */
int icrl_masked_nll(const float* log_probs, const int* valid_steps, const float* row_weight, int steps, int N,
                    float* loss, float* d_logp, icrl_stream_t stream);

/* nn.MSELoss() of the value pre-training loop (trainers.py:155, 178): loss[0] = mean (v - t)^2,
 * d_value (optional) = 2 (v - t) / count. */
int icrl_mse_loss(const float* values, const float* targets, int count, float* loss, float* d_value,
                  icrl_stream_t stream);

/* ---- Adam (trainers.py:378, 480): torch.optim.Adam defaults, bias-corrected ---- */
int icrl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   double lr, double beta1, double beta2, double eps, int step,
                   icrl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ICRL_A2C_H */
