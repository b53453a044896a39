// Sibling preference losses of the DPO trainer: SimPO, ORPO, KTO
// (align_anything/trainers/text_to_text/simpo.py:41-108, orpo.py:41-112, kto.py:83-160).
//
// All three slice the RESPONSE-WINDOW log-prob tensor of compute_log_probs ([2B, max(R)-1], right-padded with 0) with
// ABSOLUTE sequence positions: row i contributes sum(window[i, diverge_index : end_index + 1]) where end_index is the
// last attended position of the row and diverge_index the first position where chosen and rejected ids differ.  The
// slice is reproduced as the reference executes it (python slicing clamps at the tensor width; window entries beyond
// the row's own R-1 values are the 0.0 padding), pairs whose two rows are identical are skipped, and every mean runs
// over the kept pairs only.  Integer part (aa_pair_slice_index) is bit-exact; the loss part emits loss, the five
// metrics and d loss / d logp in one launch, like aa_dpo_loss_fwd_bwd.
#include "aa_common.h"

// One workgroup per pair.  Rows [0,B) chosen, [B,2B) rejected.  For row s with window rows [seq_off[s], seq_off[s+1]):
//   lo[s], hi[s] = flat-row range of the slice (possibly empty), len[s] = end_index + 1, keep[i] = rows differ.
__global__ __launch_bounds__(256) void pair_slice_index_kernel(const int64_t* __restrict__ ids,
                                                               const int64_t* __restrict__ mask, int B, int T,
                                                               const int* __restrict__ seq_off, int* __restrict__ lo,
                                                               int* __restrict__ hi, int* __restrict__ len,
                                                               uint8_t* __restrict__ keep) {
    __shared__ int red[3][4];
    const int i = blockIdx.x;
    const int64_t* a = ids + (long)i * T;
    const int64_t* b = ids + (long)(i + B) * T;
    const int64_t* ma = mask + (long)i * T;
    const int64_t* mb = mask + (long)(i + B) * T;
    int div = T, ea = -1, eb = -1;
    for (int t = threadIdx.x; t < T; t += 256) {
        if (a[t] != b[t]) div = min(div, t);
        if (ma[t] != 0) ea = max(ea, t);
        if (mb[t] != 0) eb = max(eb, t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        div = min(div, __shfl_xor(div, o, 64));
        ea = max(ea, __shfl_xor(ea, o, 64));
        eb = max(eb, __shfl_xor(eb, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = div; red[1][w] = ea; red[2][w] = eb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { div = min(div, red[0][k]); ea = max(ea, red[1][k]); eb = max(eb, red[2][k]); }
        keep[i] = div < T ? 1 : 0;
        const int e[2] = {ea, eb};
        for (int h = 0; h < 2; ++h) {
            const int s = i + h * B;
            const int n = seq_off[s + 1] - seq_off[s];          // the row's own R-1 window entries
            lo[s] = seq_off[s] + min(div, n);
            hi[s] = seq_off[s] + max(min(e[h] + 1, n), min(div, n));
            len[s] = e[h] + 1;
        }
    }
}

extern "C" int aa_pair_slice_index(const int64_t* ids, const int64_t* mask, int B, int T, const int* seq_off,
                                   int* lo, int* hi, int* len, uint8_t* keep, void* stream) {
    AA_REQUIRE(B > 0 && T > 0, "aa_pair_slice_index: bad shape B=%d T=%d", B, T);
    hipLaunchKernelGGL(pair_slice_index_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, mask, B, T, seq_off,
                       lo, hi, len, keep);
    AA_CHECK_LAUNCH("aa_pair_slice_index");
    return AA_OK;
}

#define AA_PREF_SIMPO 0
#define AA_PREF_ORPO 1
#define AA_PREF_KTO 2

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// out[0]=loss out[1]=reward_accuracy out[2]=mean reward out[3]=mean better out[4]=mean worse out[5]=mean margin
// out[6]=number of kept pairs ; per_sample [4,B] (skipped pairs: 0) ; dlogp[total_rows] fully written.
__global__ __launch_bounds__(256) void pref_loss_kernel(int kind, const float* __restrict__ pol,
                                                        const float* __restrict__ ref, const int* __restrict__ lo,
                                                        const int* __restrict__ hi, const int* __restrict__ len,
                                                        const uint8_t* __restrict__ keep, int B, int total_rows,
                                                        float beta, float p1, float p2, float p3,
                                                        float* __restrict__ out, float* __restrict__ per_sample,
                                                        float* __restrict__ dlogp) {
    __shared__ float red[8];
    if (dlogp)
        for (int t = threadIdx.x; t < total_rows; t += 256) dlogp[t] = 0.f;
    int kept = 0;
    for (int i = 0; i < B; ++i) kept += keep[i] ? 1 : 0;
    const float invk = kept > 0 ? 1.f / (float)kept : 0.f;
    __syncthreads();
    float loss = 0.f, acc = 0.f, rsum = 0.f, bsum = 0.f, wsum = 0.f, msum = 0.f;
    for (int i = 0; i < B; ++i) {
        if (!keep[i]) {
            if (threadIdx.x == 0 && per_sample)
                per_sample[i] = per_sample[B + i] = per_sample[2 * B + i] = per_sample[3 * B + i] = 0.f;
            continue;
        }
        float sum[2];
        for (int h = 0; h < 2; ++h) {
            const int s = i + h * B;
            float p = 0.f;
            for (int t = lo[s] + threadIdx.x; t < hi[s]; t += 256) p += (kind == AA_PREF_KTO) ? pol[t] - ref[t] : pol[t];
            sum[h] = block_sum<256>(p, red);
        }
        float br, wr, li, gb, gw;   // per-pair "log ratios", loss and d loss / d (window sum) of each row
        if (kind == AA_PREF_SIMPO) {            // simpo.py:79-88: p1 = gamma
            const float lb = (float)len[i], lw = (float)len[i + B];
            br = sum[0] / lb; wr = sum[1] / lw;
            const float z = beta * (br - wr) - p1;
            li = softplusf(-z);
            const float g = -beta * sigmoidf_(-z);
            gb = g / lb; gw = -g / lw;
        } else if (kind == AA_PREF_ORPO) {      // orpo.py:82-93
            const float lb = (float)len[i], lw = (float)len[i + B];
            br = sum[0] / lb; wr = sum[1] / lw;
            const float eb = expf(br), ew = expf(wr);
            const float log_odds = (br - wr) - (log1pf(-eb) - log1pf(-ew));
            li = -br + beta * softplusf(-log_odds);
            const float g = -beta * sigmoidf_(-log_odds);      // d/d log_odds of beta * softplus(-log_odds)
            gb = (-1.f + g / (1.f - eb)) / lb;
            gw = (-g / (1.f - ew)) / lw;
        } else {                                // kto.py:128-137: p1 = scale_better, p2 = scale_worse, p3 = kl
            br = sum[0]; wr = sum[1];
            const float sb = sigmoidf_(beta * (br - p3)), sw = sigmoidf_(beta * (p3 - wr));
            li = p1 * (1.f - sb) - p2 * (1.f - sw);
            gb = -p1 * beta * sb * (1.f - sb);
            gw = -p2 * beta * sw * (1.f - sw);
        }
        if (dlogp) {
            for (int t = lo[i] + threadIdx.x; t < hi[i]; t += 256) dlogp[t] = gb * invk;
            for (int t = lo[i + B] + threadIdx.x; t < hi[i + B]; t += 256) dlogp[t] = gw * invk;
        }
        const float rb = beta * br, rw = beta * wr;
        if (threadIdx.x == 0 && per_sample) {
            per_sample[i] = rb; per_sample[B + i] = rw; per_sample[2 * B + i] = rb + rw; per_sample[3 * B + i] = rb - rw;
        }
        loss += li; acc += (rb > rw) ? 1.f : 0.f; rsum += rb + rw; bsum += rb; wsum += rw; msum += rb - rw;
    }
    if (threadIdx.x == 0) {
        out[0] = loss * invk; out[1] = acc * invk; out[2] = rsum * invk; out[3] = bsum * invk; out[4] = wsum * invk;
        out[5] = msum * invk; out[6] = (float)kept;
    }
}

extern "C" int aa_pref_loss_fwd_bwd(int kind, const float* pol_logp, const float* ref_logp, const int* lo,
                                    const int* hi, const int* len, const uint8_t* keep, int B, int total_rows,
                                    float scale_coeff, float p1, float p2, float p3, float* out7,
                                    float* per_sample4B, float* dlogp, void* stream) {
    AA_REQUIRE(kind >= 0 && kind <= 2, "aa_pref_loss_fwd_bwd: kind %d (0 SimPO, 1 ORPO, 2 KTO)", kind);
    AA_REQUIRE(B > 0 && total_rows >= 0, "aa_pref_loss_fwd_bwd: bad shape B=%d rows=%d", B, total_rows);
    AA_REQUIRE(kind != AA_PREF_KTO || ref_logp != nullptr, "aa_pref_loss_fwd_bwd: KTO needs the reference log-probs");
    hipLaunchKernelGGL(pref_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kind, pol_logp, ref_logp, lo, hi,
                       len, keep, B, total_rows, scale_coeff, p1, p2, p3, out7, per_sample4B, dlogp);
    AA_CHECK_LAUNCH("aa_pref_loss_fwd_bwd");
    return AA_OK;
}

// kto.py:74-81 compute_kl: kl = (log_probs - ref_log_probs).mean() over the PADDED [2B, W] tensors (the zero padding
// counts in the denominator), clamped at 0.  denom = 2B * W.
__global__ __launch_bounds__(256) void window_kl_kernel(const float* __restrict__ pol, const float* __restrict__ ref,
                                                        int rows, float denom, float* __restrict__ out) {
    __shared__ float red[8];
    float p = 0.f;
    for (int t = threadIdx.x; t < rows; t += 256) p += pol[t] - ref[t];
    p = block_sum<256>(p, red);
    if (threadIdx.x == 0) out[0] = fmaxf(p / denom, 0.f);
}
extern "C" int aa_window_kl(const float* pol_logp, const float* ref_logp, int rows, float denom, float* out,
                            void* stream) {
    AA_REQUIRE(rows >= 0 && denom > 0.f, "aa_window_kl: bad shape rows=%d denom=%f", rows, denom);
    hipLaunchKernelGGL(window_kl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pol_logp, ref_logp, rows, denom, out);
    AA_CHECK_LAUNCH("aa_window_kl");
    return AA_OK;
}

// Supervised fine-tuning loss (trainers/text_to_text/sft.py:94-97 -> hf:loss/loss_utils.py ForCausalLMLoss): mean negative
// log-likelihood over the label positions (labels != -100 after the shift; the host-side window holds exactly those rows).
// loss = -sum(logp[0:rows]) / rows;  dlogp[r] = -1/rows for the real rows, 0 for the padding rows up to rows_pad.
__global__ __launch_bounds__(256) void sft_loss_kernel(const float* __restrict__ logp, int rows, int rows_pad,
                                                       float* __restrict__ loss, float* __restrict__ dlogp) {
    __shared__ float red[8];
    float p = 0.f;
    for (int t = threadIdx.x; t < rows; t += 256) p += logp[t];
    p = block_sum<256>(p, red);
    if (threadIdx.x == 0) loss[0] = -p / (float)rows;
    if (dlogp) {
        const float g = -1.f / (float)rows;
        for (int t = threadIdx.x; t < rows_pad; t += 256) dlogp[t] = t < rows ? g : 0.f;
    }
}
extern "C" int aa_sft_loss_fwd_bwd(const float* logp, int rows, int rows_pad, float* loss_out, float* dlogp, void* stream) {
    AA_REQUIRE(rows > 0 && rows_pad >= rows, "aa_sft_loss_fwd_bwd: rows=%d rows_pad=%d (a batch without label positions has no loss)", rows, rows_pad);
    hipLaunchKernelGGL(sft_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, rows, rows_pad, loss_out, dlogp);
    AA_CHECK_LAUNCH("aa_sft_loss_fwd_bwd");
    return AA_OK;
}
