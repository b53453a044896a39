/* aa_hip_f32.h -- fp32 PARITY MODE twins of the transformer-block entry points of aa_hip.h.
 *
 * Same signatures, argument meaning and error behaviour as the bf16 functions they mirror; every activation, weight
 * and gradient pointer is fp32 instead of bf16, the bf16 rounding points of the HF graph vanish, GEMMs run on the
 * exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32).  The production path is bf16 (aa_hip.h); this set exists so
 * the native DPO/PPO step can be compared against the reference's fp32 CPU trainer
 * (align_anything/trainers/text_to_text/dpo.py:122-237 on HF fp32 modules) at 1e-4 on the loss curve, the tolerance
 * BASELINE.json states.  The dtype-independent entry points (RL math, optimizer with p16 = NULL, aa_logprob_gather_*
 * with logits_dtype = 1, aa_image_slot_index, events) are shared with aa_hip.h.  No decode twins: rollouts stay bf16.
 */
#ifndef AA_HIP_F32_H
#define AA_HIP_F32_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* fp32 twin of aa_gemm_bf16 */
int aa_gemm_f32(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                const void* bias, const void* residual, long ldr, int act, int flags, void* stream);
/* fp32 twin of aa_rmsnorm_fwd */
int aa_rmsnorm_fwd_f32(const void* x, const void* w, void* y, float* rstd, int rows, int h, float eps,
                       void* stream);
/* fp32 twin of aa_rmsnorm_rope_fwd */
int aa_rmsnorm_rope_fwd_f32(const void* x, long ldx, const void* w, void* y, float* rstd, long rows, int hd, float eps, const int* pos, const void* cos_t,
                        const void* sin_t, int heads, void* stream);
/* fp32 twin of aa_rmsnorm_heads_bwd */
int aa_rmsnorm_heads_bwd_f32(const void* dy, const void* x, long ldx, const void* w, const float* rstd, void* dx, long lddx, float* dw, float* ws, int ws_rows,
                         long rows, int hd, int heads, void* stream);
/* fp32 twin of aa_rmsnorm_bwd */
int aa_rmsnorm_bwd_f32(const void* dy, const void* x, const void* w, const float* rstd, void* dx, float* dw,
                       float* ws, int ws_rows, int rows, int h, int add_to_dx, void* stream);
/* fp32 twin of aa_layernorm_fwd */
int aa_layernorm_fwd_f32(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int rows,
                         int h, float eps, void* stream);
/* fp32 twin of aa_layernorm_bwd */
int aa_layernorm_bwd_f32(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                         void* dx, float* dw, float* db, float* ws, int ws_rows, int rows, int h, int add_to_dx,
                         void* stream);
/* fp32 twin of aa_rope_inplace */
int aa_rope_inplace_f32(void* buf, long ld, int col0, int nheads, int hd, const int* pos, const void* cos_t,
                        const void* sin_t, long rows, int inverse, int head_stride, int precise, void* stream);
/* fp32 twin of aa_mrope_tables */
int aa_mrope_tables_f32(const int* pos3, long rows, const float* inv_freq, int half, int sec0, int sec1, void* cos_t,
                        void* sin_t, void* stream);
/* fp32 twin of aa_swiglu_fwd */
int aa_swiglu_fwd_f32(const void* gate_up, void* out, long M, int F, void* stream);
/* fp32 twin of aa_swiglu_bwd */
int aa_swiglu_bwd_f32(const void* gate_up, const void* dact, void* dgate_up, long M, int F, void* stream);
/* fp32 twin of aa_act_fwd */
int aa_act_fwd_f32(const void* x, void* y, long n, int act, void* stream);
/* fp32 twin of aa_act_bwd */
int aa_act_bwd_f32(const void* pre, const void* dy, void* dx, long n, int act, void* stream);
/* fp32 twin of aa_add */
int aa_add_f32(const void* a, const void* b, void* y, long n, void* stream);
/* fp32 twin of aa_embed_fwd */
int aa_embed_fwd_f32(const int64_t* ids, const int* slot, const void* E, const void* feat, const int* pos,
                     const void* P, void* out, long n, int h, int vocab, void* stream);
/* fp32 twin of aa_embed_bwd */
int aa_embed_bwd_f32(const int64_t* ids, const int* slot, const int* pos, const void* dx, float* dE, void* dfeat,
                     float* dP, long n, int h, int vocab, void* stream);
/* fp32 twin of aa_transpose_bf16 */
int aa_transpose_f32(const void* in, long ldi, void* out, long ldo, int R, int C, void* stream);
/* fp32 twin of aa_colsum_bf16 */
int aa_colsum_f32(const void* in, long ld, long R, int C, float* out, void* stream);
/* fp32 twin of aa_rowdot_fwd */
int aa_rowdot_fwd_f32(const void* x, const void* w, float* out, long rows, int h, void* stream);
/* fp32 twin of aa_rowdot_bwd */
int aa_rowdot_bwd_f32(const float* dy, const void* x, const void* w, void* dx, float* dw, float* ws, int ws_rows,
                      long rows, int h, void* stream);
/* fp32 twin of aa_patch_im2col */
int aa_patch_im2col_f32(const void* pixels, int pix_dtype, void* out, int n_img, int channels, int image_size,
                        int patch, int Kp, void* stream);
/* fp32 twin of aa_clip_embed */
int aa_clip_embed_f32(const void* patch, const void* cls, const void* pos, void* out, int n_img, int G2, int h,
                      void* stream);
/* fp32 twin of aa_attn_fwd */
int aa_attn_fwd_f32(const void* Q, const void* K, const void* V, void* O, float* lse, const int* start, const int* kv_len, long ldq,
                    long ldk, long ldv, long ldo, int N, int T, int H, int Hkv, int hd, int causal, float scale,
                    void* stream);
/* fp32 twin of aa_attn_bwd */
int aa_attn_bwd_f32(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                    float* delta, void* dQ, void* dK, void* dV, const int* start, const int* kv_len, long ldq, long ldk, long ldv,
                    long ldo, long lddo, long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                    int causal, float scale, void* stream);

/* fp32 twins: Whisper / Qwen2-Audio front-end (hf:models/qwen2_audio/modeling_qwen2_audio.py:315-316, 372-393): Conv1d(k = 3, padding 1,
 * stride 1 | 2) as im2col + GEMM on the HF weight as stored, its input gradient (col2im), and AvgPool1d(2) forward / backward.
 * x element (b, ci, tin) at b*sb + ci*sc + tin*st (channels-first features or token-major activations); x_dtype 0 bf16, 1 f32 */
int aa_conv1d_im2col_f32(const void* x, int x_dtype, long sb, long sc, long st, void* col, int B, int C, int Tin, int Tout,
                     int stride, void* stream);
int aa_conv1d_col2im_f32(const void* dcol, void* dx, int B, int C, int Tin, int Tout, int stride, void* stream);
int aa_avgpool2_f32(const void* x, void* y, long rows_out, int C, int backward, void* stream);

/* fp32 twins: Qwen3-MoE sparse block (see aa_hip.h) */
int aa_moe_route_f32(const void* logits, long ld, long rows, int E, int k, int norm_topk, float* probs, int* idx, void* weights,
                     void* stream);
int aa_moe_route_bwd_f32(const float* probs, const int* idx, const float* dweights, long rows, int E, int k, int norm_topk,
                         void* dlogits, long ld, void* stream);
int aa_moe_gather_f32(const void* x, const int* src_row, void* out, long rows_out, int h, void* stream);
int aa_gather2_add_f32(const void* x, const int* row_a, const int* row_b, void* out, long rows_out, int h, void* stream);
int aa_moe_combine_f32(const void* yp, const int* pos, const void* weights, const void* residual, void* out, long rows, int k, int h,
                       void* stream);
int aa_moe_combine_bwd_f32(const void* dout, const void* yp, const int* pos, const void* weights, void* dyp, float* dweights,
                           long rows, int k, int h, const int* src_row, long cap_rows, void* stream);

/* fp32 twin of aa_gemm_grouped_bf16 */
int aa_gemm_grouped_f32(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int flags,
                        int mode, const int* tile_expert, const int* seg_off, long stride, int E, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AA_HIP_F32_H */
