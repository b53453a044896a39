/*
 * aa_hip.h -- C ABI of libaa_hip.so, the MI355X (gfx950) drop-in for the DPO/PPO inner loop of
 * PKU-Alignment/align-anything.
 *
 * The reference has no FFI: its hot path reaches accelerated code only through Python APIs
 * (torch ops, HF transformers modules, DeepSpeed FusedAdam).  Each entry point below replaces one
 * of those calls; the citation names the reference (or, prefixed hf:, the HuggingFace transformers)
 * site it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 * `stream` is a hipStream_t; bf16 tensors are raw 16-bit words (torch.bfloat16 layout), row-major with
 * an explicit leading dimension in ELEMENTS; functions return 0 on success, <0 on error, and
 * aa_last_error() then returns a thread-local message (the Python side raises RuntimeError).
 * No function synchronises the device or allocates memory.
 */
#ifndef AA_HIP_H
#define AA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* activation codes (GEMM epilogue, aa_act_*) */
#define AA_ACT_NONE 0
#define AA_ACT_GELU 1
#define AA_ACT_QUICK_GELU 2
#define AA_ACT_RELU 3
#define AA_ACT_SILU 4
/* aa_gemm_bf16 flags */
#define AA_GEMM_A_T 1     /* A stored [K][M] */
#define AA_GEMM_B_N 2     /* B stored [K][N] */
#define AA_GEMM_OUT_F32 4 /* C is fp32 */
#define AA_GEMM_ACCUM 8   /* C += A*B */

/* ---- runtime ------------------------------------------------------------------------------- */
const char* aa_last_error(void);
int aa_version(void);
int aa_device_info(int* cu_count, int* lds_per_cu, int* wave_size, char* arch, int arch_len);
/* Library contexts (SURVEY.md section 8(b) `aa_ctx*`): all state behind the aa_*_set_* switches, the SwiGLU-backward plan records
   (aa_gemm_glu_bwd_probe / _plan) and the communicator (aa_comm_init, one per context) belongs to the calling THREAD's current context.
   A thread that never calls aa_ctx_set_current uses the process-wide default context -- the behaviour of every earlier version.  A host
   that drives several devices or models from one process makes one context per (thread, device / model): aa_ctx_create, then
   aa_ctx_set_current on the thread that issues that device's calls (per thread, like hipSetDevice); aa_ctx_set_current(NULL) returns to
   the default.  aa_ctx_destroy also destroys the context's communicator.  Contexts are not locked: one host thread per context at a time
   (the reference's model: one process per GPU, a single thread driving its streams -- trainers/base/supervised_trainer.py:234-271). */
int aa_ctx_create(void** ctx);
int aa_ctx_set_current(void* ctx);
int aa_ctx_get_current(void** ctx);      /* *ctx = NULL while the thread is on the default context */
int aa_ctx_destroy(void* ctx);
int aa_event_create(void** ev);
int aa_event_record(void* ev, void* stream);
int aa_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int aa_event_destroy(void* ev);
/* A one-GPU MODEL of a resident collective beside the compute stream (tools/dp_shadow.py -> profiles/r05_dp_shadow.json; the reference overlaps DeepSpeed's
 * gradient reduction with backward, base/supervised_trainer.py:258-264): a stream restricted to the compute units whose bit is set in `mask`
 * (`words` x 32 bits; hipExtStreamCreateWithCUMask), and a traffic kernel of `workgroups` long-lived workgroups streaming `bytes` src -> dst `passes` times.
 * Measurement utilities: no trainer calls them. */
int aa_stream_create_cu_mask(const unsigned int* mask, int words, void** stream_out);
int aa_stream_destroy(void* stream);
int aa_shadow_traffic(const void* src, void* dst, long bytes, int workgroups, int passes, void* stream);
int aa_probe_mfma(float* out256, int row_sel, void* stream);
int aa_probe_tr16(const int* addr64, float* out256, void* stream);

/* ---- RLHF scalar math ----------------------------------------------------------------------- */
/* align_anything/utils/tools.py:402-413 gather_log_probabilities = log_softmax + gather.
 * logits [rows, V] (dtype 0 = bf16, 1 = f32), labels int64[rows] -> logp f32[rows], lse f32[rows].
 * round_bf16 != 0 rounds logp through bf16 (what the reference returns for bf16 logits). */
int aa_logprob_gather_fwd(const void* logits, long ld, const int64_t* labels, float* logp, float* lse,
                          int rows, int V, int dtype, int round_bf16, void* stream);
/* autograd backward of the above; dlogits may alias logits */
int aa_logprob_gather_bwd(const void* logits, long ld, const int64_t* labels, const float* lse,
                          const float* dlogp, void* dlogits, long ldd, int rows, int V, int dtype,
                          void* stream);
/* The same log-prob with the lm_head inside and no [rows, V] logits buffer (utils/tools.py:402-413 applied to `model(**batch).logits`,
 * trainers/text_to_text/dpo.py:128-138): the vocabulary is walked `chunk` columns (a multiple of 2048) at a time through the caller's scratch
 * `ws` (size from aa_lmhead_logprob_ws_bytes).  hidden [rows, h] is the final-norm output, W [V, h] the lm_head weight, dtype 0 = bf16 / 1 = fp32.
 * Forward is bit-identical to aa_gemm_* + aa_logprob_gather_fwd; backward recomputes every chunk, accumulates d_hidden in fp32 across chunks
 * and writes (dw_accumulate: adds) dW [V, h] (bf16, or fp32 when dw_f32) unless dW is NULL. */
int aa_lmhead_logprob_ws_bytes(int rows, int chunk, int h, int dtype, int backward, long* bytes_out);
int aa_lmhead_logprob_fwd(const void* hidden, long ldh, const void* W, long ldw, const int64_t* labels, float* logp, float* lse,
                          void* ws, long ws_bytes, int rows, int V, int h, int chunk, int dtype, int round_bf16, void* stream);
int aa_lmhead_logprob_bwd(const void* hidden, long ldh, const void* W, long ldw, const int64_t* labels, const float* lse,
                          const float* dlogp, void* d_hidden, long lddh, void* dW, long lddw, int dw_f32, int dw_accumulate,
                          void* ws, long ws_bytes, int rows, int V, int h, int chunk, int dtype, void* stream);
/* align_anything/trainers/text_to_text/dpo.py:131-137: labels = strip_pad(ids[n])[-R_n:][1:], written at
 * labels[row_off[n] .. row_off[n] + R_n - 1); bit-exact integer path. ids int64[N,T]. */
int aa_window_labels(const int64_t* ids, int N, int T, int64_t pad_id, const int* resp_len,
                     const int* row_off, int64_t* labels, void* stream);
/* align_anything/trainers/text_to_text/dpo.py:144-203 DPOTrainer.loss (forward + d loss/d logp).
 * sequences [0,B) better, [B,2B) worse; per-token log-probs flat, sequence s = rows
 * [seq_off[s], seq_off[s+1]).  out6 = loss, reward_accuracy, mean reward, mean better, mean worse,
 * mean margin; per_sample4B = better_reward[B], worse_reward[B], reward[B], margin[B].
 * keep (uint8 [B], NULL = all): pairs with keep == 0 are skipped and every mean runs over the kept pairs -- the audio / image
 * trainers drop pairs whose chosen and rejected rows are identical (trainers/text_audio_to_text/dpo.py:139-140). */
int aa_dpo_loss_fwd_bwd(const float* pol_logp, const float* ref_logp, const int* seq_off, int B,
                        float beta, float* out6, float* per_sample4B, float* dlogp, const uint8_t* keep, void* stream);
/* trainers/text_to_text/rm.py:97-132 reward-model pairwise loss (+ L2 regularisation) and its gradient */
int aa_rm_loss_fwd_bwd(const float* end_scores, int B, float regularization, float* out2, float* dscores,
                       void* stream);
/* trainers/text_to_text/ppo.py:528-547 add_kl_divergence_regularization */
int aa_kl_reward(const float* reward, const float* logp, const float* ref_logp, const uint8_t* mask,
                 int B, int L, float kl_coeff, float clip, float* rewards_out, int* end_index_out,
                 void* stream);
/* trainers/text_to_text/ppo.py:487-508 get_advantages_and_returns (adv/ret are [B, L-start]) */
int aa_gae(const float* values, const float* rewards, const uint8_t* mask, int B, int L, int start,
           float gamma, float lam, float* adv, float* ret, void* stream);
/* trainers/text_to_text/ppo.py:291-307 actor_loss_fn (+ utils/tools.py:460-467 masked_mean) */
int aa_ppo_actor_loss(const float* logp, const float* old_logp, const float* adv, const uint8_t* mask,
                      int B, int L, float clip_ratio, float* row_scratch, float* loss_out, float* dlogp,
                      void* stream);
/* trainers/text_to_text/ppo.py:510-526 critic_loss_fn */
int aa_ppo_critic_loss(const float* values, const float* old_values, const float* returns,
                       const uint8_t* mask, int B, int L, float clip_value, float* row_scratch,
                       float* loss_out, float* dvalues, void* stream);

/* trainers/text_to_text/grpo.py:257-329: group-normalised advantage (unbiased std + 1e-4), first-EOS completion
 * mask, and the masked GRPO loss  sum(mask * -(A - beta * k3KL)) / sum(mask)  with its gradient w.r.t. logp.
 * row_scratch2 = fp32 [2 * rows]. */
int aa_group_advantage(const float* rewards, int B, int G, float* adv, void* stream);
int aa_completion_mask(const int64_t* tokens, long ld, int rows, int L, int64_t eos, uint8_t* mask, void* stream);
int aa_grpo_loss_fwd_bwd(const float* logp, const float* ref_logp, const float* adv, const uint8_t* mask, int rows,
                         int L, float beta, float* row_scratch2, float* loss_out, float* dlogp, void* stream);

/* Sibling preference losses on the same response-window log-probs: SimPO (trainers/text_to_text/simpo.py:41-108),
 * ORPO (orpo.py:41-112), KTO (kto.py:83-160).  aa_pair_slice_index reproduces the reference's slice of the padded
 * window tensor by ABSOLUTE positions [diverge_index, end_index + 1) as flat-row ranges lo/hi (int[2B]), the row lengths
 * len = end_index + 1 and keep[i] = chosen/rejected rows differ (identical pairs are skipped, :64-65).
 * aa_pref_loss_fwd_bwd: kind 0 SimPO (p1 = gamma), 1 ORPO, 2 KTO (p1 = scale_better, p2 = scale_worse, p3 = kl; needs
 * ref_logp); out7 = loss, reward_accuracy, mean reward / better / worse / margin, #kept pairs; dlogp fp32[total_rows].
 * aa_window_kl: kto.py:74-81, max(mean over the padded [2B, W] tensor of (logp - ref_logp), 0), denom = 2B * W. */
int aa_pair_slice_index(const int64_t* ids, const int64_t* mask, int B, int T, const int* seq_off, int* lo, int* hi,
                        int* len, uint8_t* keep, void* stream);
int aa_pref_loss_fwd_bwd(int kind, const float* pol_logp, const float* ref_logp, const int* lo, const int* hi,
                         const int* len, const uint8_t* keep, int B, int total_rows, float scale_coeff, float p1,
                         float p2, float p3, float* out7, float* per_sample4B, float* dlogp, void* stream);
int aa_window_kl(const float* pol_logp, const float* ref_logp, int rows, float denom, float* out, void* stream);
/* trainers/text_to_text/sft.py:94-97 (hf ForCausalLMLoss): loss = -mean(logp) over the label rows of the window, dlogp = -1/rows (pad rows 0) */
int aa_sft_loss_fwd_bwd(const float* logp, int rows, int rows_pad, float* loss_out, float* dlogp, void* stream);

/* head_dim-128 attention implementation: bit 0 = forward, bit 1 = backward on the one-wave-per-SIMD v_mfma_f32_32x32x16_bf16 kernels (csrc/attn128.inc;
 * default 3, env AA_ATTN128), 0 = the 16x16x32 kernels every other head_dim uses.  Process-wide; for same-box A/B runs and tests of both paths. */
int aa_attn_set_impl(int impl);

/* ---- transformer blocks (what model(**batch).logits executes, dpo.py:128) -------------------- */
/* torch nn.Linear / its backward: C[M,N] (+)= op(A) op(B), fp32 accumulate, fused bias/act/residual.
 * K % 64 == 0 (zero-pad), N % 4 == 0. */
int aa_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb,
                 long ldc, const void* bias, const void* residual, long ldr, int act, int flags,
                 void* stream);
/* Split-K form of aa_gemm_bf16 for few-row launches (M ~ 1000 rows against a 7B weight matrix: a PPO rollout's scoring forwards and its rl_step,
 * align_anything/trainers/text_to_text/ppo.py:224-289): the contraction is cut into S chunks that run side by side (blockIdx.y), their fp32 partial products
 * land in `ws` (S x M x N floats, caller-owned) and one pass sums them in order and applies the same epilogue.  Deterministic; equal to aa_gemm_bf16 up to the
 * fp32 association of the accumulator. */
int aa_gemm_splitk_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, const void* bias,
                        const void* residual, long ldr, int act, int flags, float* ws, int S, void* stream);
int aa_gemm_set_tile(int tile);
int aa_gemm_set_group(int gm);   /* tile-group height of the L2-aware tile order (0 = heuristic: 4, or 3 for NN with wide N) */
/* hf:models/llama/modeling_llama.py:62-67 LlamaRMSNorm */
int aa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int h, float eps,
                   void* stream);
/* y = rope(rmsnorm(x)) over rows = tokens x heads of width hd: Qwen3's per-head q_norm / k_norm followed by apply_rotary_pos_emb
   (hf:models/qwen3_moe/modeling_qwen3_moe.py attention forward, reached from align_anything/models/qwen3_moe.py:28-60) in ONE pass; pos[token] indexes the
   [., hd / 2] cos / sin tables; x may be a column slice of the fused q | k | v projection output (ldx = elements per token); bit-identical to aa_rmsnorm_fwd
   followed by aa_rope_inplace */
int aa_rmsnorm_rope_fwd(const void* x, long ldx, const void* w, void* y, float* rstd, long rows, int hd, float eps, const int* pos, const void* cos_t,
                        const void* sin_t, int heads, void* stream);
/* backward of the per-head norm when x (the saved projection output) and dx are column slices of the fused q | k | v buffers: row = (token, head) at token * ld + head * hd; dy dense; dw [hd] fp32 accumulated through ws [ws_rows, hd] */
int aa_rmsnorm_heads_bwd(const void* dy, const void* x, long ldx, const void* w, const float* rstd, void* dx, long lddx, float* dw, float* ws, int ws_rows,
                         long rows, int hd, int heads, void* stream);
/* dw (fp32 [h], accumulated) needs ws = fp32 [ws_rows, h] scratch (per-workgroup partial rows) */
int aa_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, float* dw,
                   float* ws, int ws_rows, int rows, int h, int add_to_dx, void* stream);
/* torch F.layer_norm (CLIP, OPT) */
int aa_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                     int rows, int h, float eps, void* stream);
int aa_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                     void* dx, float* dw, float* db, float* ws /* [2, ws_rows, h] */, int ws_rows, int rows,
                     int h, int add_to_dx, void* stream);
/* hf:models/llama/modeling_llama.py:113-160 rotary embedding, in place, inverse = backward.  head_stride = columns
 * between consecutive heads (0 -> hd; > hd for zero-padded heads); precise = 1: fp32 tables (float*) and fp32 arithmetic
 * with one rounding, hf:models/qwen2_vl/modeling_qwen2_vl.py:225-236 (vision rotary) */
int aa_rope_inplace(void* buf, long ld, int col0, int nheads, int hd, const int* pos, const void* cos_t,
                    const void* sin_t, long rows, int inverse, int head_stride, int precise, void* stream);
/* hf:models/qwen2_vl/modeling_qwen2_vl.py:156-222 multimodal RoPE: per-token cos/sin rows [rows, half] from the 3-D position
 * ids pos3 [3, rows]; frequency f uses component 0 / 1 / 2 for f < sec0 / < sec0 + sec1 / else */
int aa_mrope_tables(const int* pos3, long rows, const float* inv_freq, int half, int sec0, int sec1, void* cos_t,
                    void* sin_t, void* stream);
/* GEMMs with the element-wise neighbour of the HF graph folded into the epilogue of the one-wave-per-SIMD kernel (csrc/gemm4.hip).
 * Each runs fused when the shape qualifies (M and N multiples of 256, K of 128, 16-byte aligned rows; rotary: head_dim 128) and as the
 * unfused pair of kernels otherwise -- same rounding points, bit-identical results (tests/test_gemm_gpu.py).
 *  - aa_gemm_qkv_rope_bf16: hf:models/llama/modeling_llama.py:228-246 (q/k/v projections of the fused [q|k|v] weight) +
 *    apply_rotary_pos_emb :130-160 on the heads in columns [0, rope_cols); pos[M] int32, cos_t / sin_t [max_pos, hd/2] bf16
 *  - aa_gemm_glu_fwd_bf16: LlamaMLP :163-176: GU[M, 2F] = A [Wgate; Wup]^T (kept for the backward), ACT[M, F] = silu(gate) * up
 *  - aa_gemm_glu_bwd_bf16: its backward: dGU[M, 2F] from d_act = dY[M, K] Wdown[K, F] (never stored when fused; the unfused path
 *    needs the [M, F] workspace dact_ws) and the saved GU
 *    needs the [M, F] workspace dact_ws) and the saved GU.  Which of the two runs is a per-(M, F, K) plan: aa_gemm_glu_bwd_plan
 *    writes *plan = 1 (fused), 0 (unfused pair: dact_ws required) or 2 (fusable, not yet measured: fused); aa_gemm_glu_bwd_probe times both
 *    variants on the caller's buffers (`reps` launches each, HIP events on `stream`, synchronises it), records the faster one for the
 *    shape and leaves dGU computed -- boxes with a longer memory round trip run the fused epilogue 1.6 x slower than the pair
 *    (profiles/r02_glu_bwd_latency.txt), which no kernel can see from the inside; aa_gemm_glu_bwd_set_mode(-1 / 0 / 1) = follow the
 *    record / always unfused / always fused (env AA_GLU_BWD), aa_gemm_glu_bwd_forget() drops the records
 *  - aa_gemm_set_fuse(0): always the unfused kernels (A/B and parity runs; env AA_GEMM_FUSE=0 does the same) */
int aa_gemm_qkv_rope_bf16(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc, const int* pos,
                          const void* cos_t, const void* sin_t, int rope_cols, int hd, void* stream);
int aa_gemm_glu_fwd_bf16(const void* A, const void* Wgu, void* GU, void* ACT, int M, int F, int K, long lda, long ldw, long ldgu,
                         long ldact, void* stream);
int aa_gemm_glu_bwd_bf16(const void* dY, const void* Wdown, const void* GU, void* dGU, void* dact_ws, int M, int F, int K, long ldy,
                         long ldw, long ldgu, long lddgu, void* stream);
int aa_gemm_glu_bwd_plan(const void* dY, const void* Wdown, const void* GU, const void* dGU, int M, int F, int K, long ldy, long ldw,
                         long ldgu, long lddgu, int* plan);
int aa_gemm_glu_bwd_probe(const void* dY, const void* Wdown, const void* GU, void* dGU, void* dact_ws, int M, int F, int K, long ldy,
                          long ldw, long ldgu, long lddgu, int reps, float* ms_fused, float* ms_unfused, void* stream);
int aa_gemm_glu_bwd_set_mode(int mode);
int aa_gemm_glu_bwd_forget(void);
int aa_gemm_set_fuse(int on);
/* hf:models/llama/modeling_llama.py:163-176 LlamaMLP gate: silu(gate)*up on [M, 2F] -> [M, F] */
int aa_swiglu_fwd(const void* gate_up, void* out, long M, int F, void* stream);
int aa_swiglu_bwd(const void* gate_up, const void* dact, void* dgate_up, long M, int F, void* stream);
int aa_act_fwd(const void* x, void* y, long n, int act, void* stream);
int aa_act_bwd(const void* pre, const void* dy, void* dx, long n, int act, void* stream);
int aa_add(const void* a, const void* b, void* y, long n, void* stream);
/* hf:models/llava/modeling_llava.py:234-248 embed_tokens + masked_scatter of image features */
int aa_image_slot_index(const int64_t* ids, int n, int64_t image_token_id, int* slot, int* count,
                        void* stream);
int aa_embed_fwd(const int64_t* ids, const int* slot, const void* E, const void* feat, const int* pos,
                 const void* P, void* out, long n, int h, int vocab, void* stream);
int aa_embed_bwd(const int64_t* ids, const int* slot, const int* pos, const void* dx, float* dE,
                 void* dfeat, float* dP, long n, int h, int vocab, void* stream);
int aa_transpose_bf16(const void* in, long ldi, void* out, long ldo, int R, int C, void* stream);
int aa_colsum_bf16(const void* in, long ld, long R, int C, float* out, void* stream);
/* score head Linear(h->1, no bias) of the reward / critic models: align_anything/models/opt.py:59-60,
 * models/llava.py:60.  out f32[rows] = float(bf16(x . w)); backward gives dx and accumulates dw (fp32). */
int aa_rowdot_fwd(const void* x, const void* w, float* out, long rows, int h, void* stream);
int aa_rowdot_bwd(const float* dy, const void* x, const void* w, void* dx, float* dw, float* ws, int ws_rows,
                  long rows, int h, void* stream);
/* Whisper / Qwen2-Audio front-end (hf:models/qwen2_audio/modeling_qwen2_audio.py:315-316, 372-393): Conv1d(k = 3, padding 1,
 * stride 1 | 2) as im2col + GEMM on the HF weight as stored, its input gradient (col2im), and AvgPool1d(2) forward / backward.
 * x element (b, ci, tin) at b*sb + ci*sc + tin*st (channels-first features or token-major activations); x_dtype 0 bf16, 1 f32 */
int aa_conv1d_im2col(const void* x, int x_dtype, long sb, long sc, long st, void* col, int B, int C, int Tin, int Tout,
                     int stride, void* stream);
int aa_conv1d_col2im(const void* dcol, void* dx, int B, int C, int Tin, int Tout, int stride, void* stream);
int aa_avgpool2(const void* x, void* y, long rows_out, int C, int backward, void* stream);
/* Qwen3-MoE sparse block (hf:models/qwen3_moe/modeling_qwen3_moe.py:210-283): router softmax (fp32) + top-k (k <= 8, ties -> lower
 * expert) + optional renormalisation, weights cast to the activation dtype; its backward from d weights; expert-major token copy
 * (src_row < 0 -> zero pad row); weighted combine as a gather over each token's k expert rows (ascending expert order, like
 * hf's index_add; weights NULL = 1; optional residual) and its backward (dYp rows + d weights). */
int aa_moe_route(const void* logits, long ld, long rows, int E, int k, int norm_topk, float* probs, int* idx, void* weights,
                 void* stream);
int aa_moe_route_bwd(const float* probs, const int* idx, const float* dweights, long rows, int E, int k, int norm_topk,
                     void* dlogits, long ld, void* stream);
int aa_moe_gather(const void* x, const int* src_row, void* out, long rows_out, int h, void* stream);
/* out[r] = x[row_a[r]] + x[row_b[r]] (an index of -1 contributes a zero row; one rounding of the sum): the pack-reduce of shared-prompt packing -- the
 * gradient of a token row that the reference layout holds twice (the common prompt of a preference pair, rows [0, B) and [B, 2B) of
 * trainers/text_image_to_text/dpo.py:85-105) is the sum over its copies.  Bit-identical to aa_moe_gather x 2 + aa_add. */
int aa_gather2_add(const void* x, const int* row_a, const int* row_b, void* out, long rows_out, int h, void* stream);
int aa_moe_combine(const void* yp, const int* pos, const void* weights, const void* residual, void* out, long rows, int k, int h,
                   void* stream);
/* dyp rows no (token, slot) pair maps to carry no gradient: src_row (aa_moe_plan's row -> token table, -1 = pad) + cap_rows make the launch zero them;
 * src_row == NULL: the caller has zeroed dyp. */
int aa_moe_combine_bwd(const void* dout, const void* yp, const int* pos, const void* weights, void* dyp, float* dweights,
                       long rows, int k, int h, const int* src_row, long cap_rows, void* stream);
/* Expert-major layout of a routing decision, computed on the device (no host read): counts[E], segment offsets off[E+1] aligned to
 * `align` rows (128 or 256: the row tiles of aa_gemm_grouped_*), pos[rows*k] (row of every (token, slot) pair, stable in token order),
 * src[cap_rows] (token of every row, -1 = pad) and tile_expert[cap_rows / tg] (-1 beyond the rows in use; tg = 128 rows per entry when align is a
 * multiple of 128, else align).
 * cap_rows >= rows*k + E*(align-1) rounded to `align`: an upper bound known without looking at the routing. */
int aa_moe_plan(const int* idx, long rows, int k, int E, int align, long cap_rows, int* counts, int* off, int* pos, int* src,
                int* tile_expert, void* stream);
/* Grouped GEMM over the expert-major buffer, all experts in ONE launch, sizes read from the device tables of aa_moe_plan:
 *   mode 1: C[M, N] = A[M, K] * op(B_e) where the 128-row tile t of A / C uses expert tile_expert[t]'s matrix B + e * stride
 *           (AA_GEMM_B_N selects the [K][N] layout: forward NT and input-gradient NN of the expert MLPs); tiles with -1 exit (are zero-filled on
 *           the 256-row tile);
 *   mode 3: mode 1 with the promise that the segments are aligned to 256 rows (aa_moe_plan align 256): runs on the 256 x 256 one-wave-per-SIMD tile
 *           when M % 256 == 0, N % 256 == 0 and K % 128 == 0 (env AA_MOE_GEMM4=0: never), otherwise like mode 1;
 *   mode 2: C_e[M, N] (+)= A[r0:r1, :M]^T * B[r0:r1, :N] with [r0, r1) = seg_off[e], seg_off[e+1] and C_e = C + e * stride
 *           (TN: the per-expert weight gradients; an expert without tokens gets zeros). */
int aa_gemm_grouped_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int flags,
                         int mode, const int* tile_expert, const int* seg_off, long stride, int E, void* stream);
/* hf:models/clip/modeling_clip.py:138-218 CLIPVisionEmbeddings (patch conv as im2col + GEMM) */
int aa_patch_im2col(const void* pixels, int pix_dtype, void* out, int n_img, int channels,
                    int image_size, int patch, int Kp, void* stream);
int aa_clip_embed(const void* patch, const void* cls, const void* pos, void* out, int n_img, int G2,
                  int h, void* stream);
int aa_f32_to_bf16(const float* in, void* out, long n, void* stream);
/* torch SDPA in hf:models/llama/modeling_llama.py:243-281 / clip :289 / opt attention.
 * start[n] = first valid key of left-padded sequence n (or NULL); kv_len[n] = number of valid keys of a right-padded
 * sequence (keys >= kv_len[n] masked; NULL = T).  lse f32[N,H,T]. */
int aa_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, const int* start, const int* kv_len,
                long ldq, long ldk, long ldv, long ldo, int N, int T, int H, int Hkv, int hd, int causal,
                float scale, void* stream);
int aa_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                const float* lse, float* delta, void* dQ, void* dK, void* dV, const int* start, const int* kv_len, long ldq,
                long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, int N, int T,
                int H, int Hkv, int hd, int causal, float scale, void* stream);
/* aa_attn_bwd + the backward of the rotary embedding (aa_rope_inplace with inverse = 1 on dQ and dK; hf apply_rotary_pos_emb, modeling_llama.py:130-160) in the
 * epilogues of the dQ and dK/dV kernels: pos[N * T] int32 rotary position of every token row, cos_t / sin_t [., hd / 2] bf16 (the tables of the forward).
 * Bit-identical to the two launches; saves a pass over d[q | k]. */
int aa_attn_bwd_rope(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                     const float* lse, float* delta, void* dQ, void* dK, void* dV,
                     const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                     long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                     int causal, float scale, const int* pos, const void* cos_t, const void* sin_t, void* stream);

/* aa_attn_fwd / aa_attn_bwd(_rope) with q_skip[N] (shared-prompt packing, trainers/common.py::build_pack_plan): query rows below q_skip[n] of sequence n have
 * no consumer -- the rejected row's copy of the pair's common prefix, whose outputs the packed layout takes from the chosen row.  Whole query blocks below it
 * are not computed (O / lse rows stay unwritten), get dQ = 0 and are left out of the dK / dV accumulation; the caller hands dO = 0 for those rows.  pos / cos_t /
 * sin_t of the backward: all NULL (aa_attn_bwd) or all given (aa_attn_bwd_rope).  bf16 only.  Same reference call site as aa_attn_fwd. */
int aa_attn_fwd_qskip(const void* Q, const void* K, const void* V, void* O, float* lse, const int* start, const int* kv_len,
                      long ldq, long ldk, long ldv, long ldo, int N, int T, int H, int Hkv, int hd, int causal, float scale,
                      const int* q_skip, void* stream);
int aa_attn_bwd_qskip(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                      const float* lse, float* delta, void* dQ, void* dK, void* dV,
                      const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                      long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                      int causal, float scale, const int* pos, const void* cos_t, const void* sin_t, const int* q_skip, void* stream);

/* ---- data-parallel exchange for non-Python hosts (csrc/comm.hip; the Python host side uses torch.distributed for the same three
 * operations).  RCCL is bound at run time (dlopen librccl.so); one communicator per process = per GPU.
 * reference: DeepSpeed's gradient all-reduce behind engine.backward / step (trainers/text_to_text/dpo.py:212-213) and
 * utils/multi_process.py:74-89 get_all_reduce_mean / get_all_reduce_max on the logged scalars. */
int aa_comm_unique_id(void* id128);                               /* rank 0: the 128-byte id to hand to the other ranks */
int aa_comm_init(const void* id128, int rank, int world);         /* after hipSetDevice(local rank) */
int aa_comm_world(int* rank, int* world);
int aa_comm_destroy(void);
/* SUM all-reduce in place of one slice of a flat gradient buffer (dtype 0 = bf16, 1 = fp32) on `stream` (a side stream overlaps it
 * with the backward of the layers below); 1/world is folded into aa_grad_sumsq / aa_adamw_flat */
int aa_grad_allreduce_bucket(void* grads, long count, int dtype, void* stream);
int aa_metrics_allreduce(float* vals, int n, int op /* 0 mean, 1 max */, void* stream);
int aa_broadcast(void* buf, long bytes, int root, void* stream);

/* ---- autoregressive decode (replaces HF generate in the PPO rollout, trainers/text_to_text/ppo.py:209-222) ---- */
/* out[M<=16, N] = x[M,K] W[N,K]^T (+bias) (+residual, HF rounding); K % 32 == 0; HBM-streaming skinny GEMM */
int aa_gemm_skinny_bf16(const void* x, const void* W, void* out, int M, int N, int K, long ldx, long ldw, long ldo,
                        const void* bias, const void* residual, long ldr, void* stream);
/* the same weight stream with the element-wise kernel that precedes it in a decode step folded in:
 * prologue 1: out = RMSNorm(x; norm_w, eps) W^T (hf LlamaRMSNorm, models/llama/modeling_llama.py:50-68; the per-row rstd is applied
 *             to the finished dot products, so rounding differs from the unfused chain by bf16 noise -- rollout sampling only);
 * prologue 2: x = [gate | up] rows [M, 2K], out = (silu(gate) * up) W^T (LlamaMLP :154-158), same values as aa_swiglu_fwd. */
int aa_gemm_skinny_fused_bf16(const void* x, const void* W, void* out, int M, int N, int K, long ldx, long ldw, long ldo,
                              const void* bias, const void* residual, long ldr, int prologue, const void* norm_w, float eps,
                              void* stream);
/* rollout-only weight layout: out[((n/16)*(K/32) + k/32)*512 + ((n%16) + 16*((k%32)/8))*8 + k%8] = W[n, k] (rows beyond N zero; out holds
 * ceil(N/16)*16*K elements) -- every fragment load of aa_gemm_skinny_swz_bf16 is then 1 KB contiguous.  `generate` builds it once per call
 * (the weights are frozen for the whole rollout); results are bit-identical to aa_gemm_skinny_bf16 on the row-major matrix. */
int aa_swizzle_weights_bf16(const void* W, long ld, void* out, int N, int K, void* stream);
int aa_gemm_skinny_swz_bf16(const void* x, const void* Wswz, void* out, int M, int N, int K, long ldx, long ldo, const void* bias,
                            const void* residual, long ldr, void* stream);
/* The strip kernel with the element-wise kernel that FOLLOWS it in a decode position folded into its epilogue -- possible because the row order
 * inside a strip of the rollout-only weight copy is free (aa_swizzle_weights_perm_bf16): mode 1 puts gate column c and the up value of the same
 * column into one strip of the fused [gate; up] weight (aa_gemm_skinny_swz_glu_bf16 = GEMV + aa_swiglu_fwd, hf LlamaMLP :163-176), mode 2 both
 * members d / d + 64 of every rotation pair of a head_dim-128 head of the fused [q | k | v] weight (aa_gemm_skinny_swz_rope_cache_bf16 = GEMV +
 * aa_decode_rope_cache: q rotated into q_out, k rotated into / v copied into the KV-cache slot of the position).  Bit-identical to the unfused
 * pairs (tests/test_decode_gpu.py); two launches and two small round trips less per layer and position. */
int aa_swizzle_weights_perm_bf16(const void* W, long ld, void* out, int N, int K, int mode, void* stream);
int aa_gemm_skinny_swz_glu_bf16(const void* x, const void* Wswz, void* act, int M, int F, int K, long ldx, long ldo, void* stream);
int aa_gemm_skinny_swz_rope_cache_bf16(const void* x, const void* Wswz, void* q_out, int M, int H, int Hkv, int K, long ldx, long ldq,
                                       const void* bias, const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc, int Tmax,
                                       const int64_t* slot, void* stream);
/* The RMSNorm that PRECEDES a projection of a decode position folded into the strip kernel (hf LlamaRMSNorm + q/k/v, gate/up or lm_head:
 * hf:models/llama/modeling_llama.py:62-67, :243-281, :163-176, :413).  rmsnorm(x; w, eps) W^T = rstd(x) * x (W diag(w))^T: the rollout-only copy is
 * made of W diag(w) (aa_swizzle_weights_scaled_bf16, mode 0 / 1 / 2 = the plain / [gate; up] / head_dim-128 row orders above, kscale = w [K]), the
 * strip kernel accumulates sum(x^2) from the fragments it feeds the MFMA and scales the finished dot products by rstd.  x is the UN-normalised
 * residual stream.  Rounding differs from the separate kernel (bf16(W w) instead of w * bf16(x rstd)): bf16 noise, rollout sampling only. */
int aa_swizzle_weights_scaled_bf16(const void* W, long ld, void* out, int N, int K, int mode, const void* kscale, void* stream);
int aa_gemm_skinny_swz_norm_bf16(const void* x, const void* Wswz, void* out, int M, int N, int K, long ldx, long ldo, const void* bias,
                                 const void* residual, long ldr, float eps, void* stream);
int aa_gemm_skinny_swz_norm_glu_bf16(const void* x, const void* Wswz, void* act, int M, int F, int K, long ldx, long ldo, float eps, void* stream);
int aa_gemm_skinny_swz_norm_rope_cache_bf16(const void* x, const void* Wswz, void* q_out, int M, int H, int Hkv, int K, long ldx, long ldq,
                                            const void* bias, const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc, int Tmax,
                                            const int64_t* slot, float eps, void* stream);
/* new token of every sequence: rotate the q heads of the fused [q|k|v] row in place (aa_rope_inplace rounding), rotate the k heads
 * into cache[(n*Tmax + slot[n]), 0:Hkv*hd] and copy the v heads to [.., Hkv*hd:2*Hkv*hd] (HF DynamicCache.update) */
int aa_decode_rope_cache(void* qkv, long ld, int N, int H, int Hkv, int hd, const int* pos, const void* cos_t, const void* sin_t,
                         void* cache, long ldc, int Tmax, const int64_t* slot, void* stream);
/* sparse-MoE block at one token per sequence (hf:models/qwen3_moe/modeling_qwen3_moe.py:210-283 during generate):
 * out[r, :] = x[r / x_div, :] W3[row_expert[r]]^T for R routed rows r = (token, choice); W3 [E, N, K], expert stride strideE */
int aa_moe_gemv_bf16(const void* x, const void* W3, void* out, int R, int N, int K, long ldx, long ldw, long ldo,
                     const int* row_expert, long strideE, int x_div, void* stream);
/* launch rules of the decode kernels (a bit mask; process-wide; the previous mask is returned through *old when given; initial value: env AA_DECODE_R6, default 3).
 * bit 0: a narrow deep strip launch with more strips than compute units gets 8 waves per strip (one round instead of two), and aa_attn_decode keeps four key
 *        steps in flight per wave when H * N < 128;  bit 1: the deep 16-wave strips (K >= 8192: the down projection) run software-pipelined trips (same sums,
 *        bit for bit);  bit 2: four key steps in flight whenever H * N < 512.  0 = the round-5 rules (same-box A/B; generate() has no reference-side counterpart: hf generate, ppo.py:209-222) */
int aa_decode_set_rules(int mask, int* old);
/* one query per sequence against the token-major KV cache [N, Tmax, Hkv*hd] (row stride ldc); keys [start[n], len[n]) */
int aa_attn_decode(const void* q, long ldq, const void* Kc, const void* Vc, long ldc, int Tmax, const int* start,
                   const int* len, void* o, long ldo, int N, int H, int Hkv, int hd, float scale, void* stream);
/* greedy token (first index of the max, torch.argmax rule) and temperature/top-p sampling (HF logits warpers) */
/* seen (uint8 [rows, ld_seen], NULL = off) + repetition_penalty: hf RepetitionPenaltyLogitsProcessor on the fp32 scores
 * before the warpers; aa_mark_seen sets seen[row, ids[row, j]] = 1 (prompt ids incl. pads, then each new token) */
int aa_mark_seen(const int64_t* ids, long ld, int rows, int L, uint8_t* seen, long ld_seen, int V, void* stream);
int aa_argmax_rows(const void* logits, long ld, int rows, int V, const uint8_t* seen, long ld_seen,
                   float repetition_penalty, int64_t* out, void* stream);
int aa_sample_top_p(const void* logits, long ld, int rows, int V, float temperature, float top_p,
                    const float* uniform, const uint8_t* seen, long ld_seen, float repetition_penalty, int64_t* out,
                    void* stream);
/* the same with HF's TopKLogitsWarper between temperature and top-p (warper order of hf:generation/utils.py): top_k > 0 keeps every score >= the
   k-th largest (ties kept), 0 = no cut.  The reference's GenerationConfig (trainers/text_to_text/ppo.py:161-170) inherits HF's default top_k
   (50 under transformers 4.x, None under 5.x): generation.py passes the installed default unless train_cfgs.top_k says otherwise */
int aa_sample_top_k_top_p(const void* logits, long ld, int rows, int V, float temperature, int top_k, float top_p,
                          const float* uniform, const uint8_t* seen, long ld_seen, float repetition_penalty, int64_t* out, void* stream);

/* trainers/text_image_to_text/ppo.py:56-86 move_padding_left on the generated sequences (circular shift per row, bit-exact) */
int aa_move_padding_left(const int64_t* in, long ldi, int64_t* out, long ldo, int rows, int L, int64_t pad, void* stream);
/* The decode loop's per-position bookkeeping (hf GenerationMixin._sample: `next_tokens * unfinished + pad * (1 - unfinished)`, the scatter into the
 * output, EosTokenCriteria; then the cache slot / position / length counters) in two launches instead of ~11 one-element torch kernels:
 * aa_decode_record after the selection kernel (tok[n] = unfinished[n] ? selected[n] : pad; out[n, tslot[n]] = tok[n]; nact += any(unfinished);
 * unfinished[n] &= tok[n] != eos, eos < 0 = none; unfinished = one byte per row), aa_decode_tick after the decode pass (tslot, pos, length, step += 1). */
int aa_decode_record(const int64_t* selected, uint8_t* unfinished, int64_t* out, long ldo, const int64_t* tslot, int64_t* tok, int64_t* nact, int N,
                     int64_t pad, int64_t eos, void* stream);
int aa_decode_tick(int64_t* tslot, int* pos, int* length, int64_t* step, int N, void* stream);

/* ---- optimizer (DeepSpeed FusedAdam + gradient_clipping, supervised_trainer.py:245-249) ------ */
/* *out_accum += sum((g*scale)^2); deterministic (no float atomics): ws = caller-owned scratch of AA_SUMSQ_WS floats, so the
   clip coefficient is bit-identical on every data-parallel rank */
#define AA_SUMSQ_WS 2048
int aa_grad_sumsq(const void* g, int g_dtype, long n, float scale, float* out_accum, float* ws, void* stream);
/* out[i] = sum_r in[r * chunk + i] (i < chunk; fp32 accumulation in rank order, one rounding to dtype 0 = bf16 / 1 = f32): the reduce step between the
   all-to-all and the all-gather of the direct (w - 1 link) gradient exchange, engine.GradReducer mode 'direct' -- DeepSpeed's gradient reduction behind
   engine.backward / engine.step (trainers/text_to_text/dpo.py:212-213), SURVEY.md section 8(e).  chunk % 8 == 0, 16-byte aligned buffers */
int aa_chunk_sum(const void* in, void* out, int dtype, long chunk, int world, void* stream);
/* coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)), norm = sqrt(sumsq).  *sumsq == -inf is the "skip this update" sentinel (the all-reduced
   capacity-overflow flag of the expert-parallel exchange): coef = norm = -1, and aa_adamw_flat returns without writing on a negative coefficient */
int aa_clip_coef(const float* sumsq, float max_norm, float* coef_out, float* norm_out, void* stream);
/* 1: use the <= 16-VGPR update kernel that can be co-resident with the 256x256 GEMM tiles (overlapped optimizer), 0: default */
int aa_adamw_set_thin(int on);
int aa_adamw_flat(float* master, float* m, float* v, void* p16, const void* g, int g_dtype, long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float gscale, const float* clip_coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AA_HIP_H */
