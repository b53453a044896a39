// RLHF scalar math of the DPO/PPO inner loop, hand-written for gfx950.
//   aa_logprob_gather_{fwd,bwd}   <- align_anything/utils/tools.py:402-413 (log_softmax + gather)
//   aa_dpo_loss_fwd_bwd           <- align_anything/trainers/text_to_text/dpo.py:144-203
//   aa_kl_reward                  <- trainers/text_to_text/ppo.py:528-547
//   aa_gae                        <- trainers/text_to_text/ppo.py:487-508
//   aa_ppo_actor_loss             <- trainers/text_to_text/ppo.py:291-307 (+ utils/tools.py:460-467 masked_mean)
//   aa_ppo_critic_loss            <- trainers/text_to_text/ppo.py:510-526
// All HBM-bound: one wide coalesced pass per operand, fp32 arithmetic, reductions in LDS/wave shuffles.
#include "aa_common.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// ------------------------------------------------------------------ log-prob gather
// One 256-thread workgroup per logits row. Single streaming read of the row (16 B / lane),
// online (max, sum-exp) per lane, then a workgroup reduction. The label pick is an indexed read of
// the same row (bit-exact index path).
template <typename T> struct RowLoad;
template <> struct RowLoad<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
        u16x8 r = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(r[i]);
    }
    __device__ static __forceinline__ float one(const bf16_t* p) { return bf2f(*p); }
};
template <> struct RowLoad<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void load(const float* p, float* v) {
        f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    __device__ static __forceinline__ float one(const float* p) { return *p; }
};

template <typename T>
__global__ __launch_bounds__(256) void logprob_fwd_kernel(const T* __restrict__ logits, long ld,
                                                          const int64_t* __restrict__ labels,
                                                          float* __restrict__ logp,
                                                          float* __restrict__ lse_out, int V,
                                                          int round_bf16) {
    __shared__ float red[8];
    constexpr int VEC = RowLoad<T>::VEC;
    const long row = blockIdx.x;
    const T* x = logits + row * ld;
    float m = -INFINITY, s = 0.f;
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int nvec = aligned ? V / VEC : 0;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC];
        RowLoad<T>::load(x + (long)i * VEC, v);
        float vm = v[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) vm = fmaxf(vm, v[j]);
        const float mn = fmaxf(m, vm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += exp2f((v[j] - mn) * LOG2E);
        s = s * exp2f((m - mn) * LOG2E) + acc;
        m = mn;
    }
    for (int i = nvec * VEC + threadIdx.x; i < V; i += 256) {  // tail / unaligned rows
        const float v = RowLoad<T>::one(x + i);
        const float mn = fmaxf(m, v);
        s = s * exp2f((m - mn) * LOG2E) + exp2f((v - mn) * LOG2E);
        m = mn;
    }
    const float gm = block_max<256>(m, red);
    const float sc = (m == -INFINITY) ? 0.f : s * exp2f((m - gm) * LOG2E);
    const float gs = block_sum<256>(sc, red);
    if (threadIdx.x == 0) {
        const float lse = gm + logf(gs);
        const int64_t lab = labels[row];
        float out;
        if (lab < 0 || lab >= V) {
            out = __builtin_nanf("");  // torch.gather would raise; surface it as NaN
        } else {
            out = RowLoad<T>::one(x + lab) - lse;
            if (round_bf16) out = rbf(out);  // reference returns bf16 log-probs for bf16 logits
        }
        logp[row] = out;
        if (lse_out) lse_out[row] = lse;
    }
}

// dlogits[r, v] = dlogp[r] * (1[v == label_r] - exp(logits[r, v] - lse_r)); may alias logits (in place)
template <typename T>
__global__ __launch_bounds__(256) void logprob_bwd_kernel(const T* logits, long ld,
                                                          const int64_t* __restrict__ labels,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ dlogp,
                                                          T* dlogits, long ldd, int V) {
    constexpr int VEC = RowLoad<T>::VEC;
    const long row = blockIdx.x;
    const T* x = logits + row * ld;
    T* dx = dlogits + row * ldd;
    const float g = dlogp[row];
    const float l = lse[row];
    const int lab = (int)labels[row];
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    const int nvec = aligned ? V / VEC : 0;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC];
        RowLoad<T>::load(x + (long)i * VEC, v);
        const int base = i * VEC;
        if constexpr (sizeof(T) == 2) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float p = exp2f((v[j] - l) * LOG2E);
                o[j] = f2bf(g * ((base + j == lab ? 1.f : 0.f) - p));
            }
            *reinterpret_cast<u16x8*>(dx + base) = o;
        } else {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float p = exp2f((v[j] - l) * LOG2E);
                o[j] = g * ((base + j == lab ? 1.f : 0.f) - p);
            }
            *reinterpret_cast<f32x4*>(dx + base) = o;
        }
    }
    for (int i = nvec * VEC + threadIdx.x; i < V; i += 256) {
        const float p = exp2f((RowLoad<T>::one(x + i) - l) * LOG2E);
        const float o = g * ((i == lab ? 1.f : 0.f) - p);
        if constexpr (sizeof(T) == 2) dx[i] = f2bf(o); else dx[i] = o;
    }
}

extern "C" int aa_logprob_gather_fwd(const void* logits, long ld, const int64_t* labels, float* logp,
                                     float* lse, int rows, int V, int dtype, int round_bf16,
                                     void* stream) {
    AA_REQUIRE(rows >= 0 && V > 0 && ld >= V, "aa_logprob_gather_fwd: bad shape rows=%d V=%d ld=%ld", rows, V, ld);
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_logprob_gather_fwd: dtype must be 0 (bf16) or 1 (f32)");
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(logprob_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, st,
                           (const bf16_t*)logits, ld, labels, logp, lse, V, round_bf16);
    else
        hipLaunchKernelGGL(logprob_fwd_kernel<float>, dim3(rows), dim3(256), 0, st,
                           (const float*)logits, ld, labels, logp, lse, V, round_bf16);
    AA_CHECK_LAUNCH("aa_logprob_gather_fwd");
    return AA_OK;
}

extern "C" int aa_logprob_gather_bwd(const void* logits, long ld, const int64_t* labels,
                                     const float* lse, const float* dlogp, void* dlogits, long ldd,
                                     int rows, int V, int dtype, void* stream) {
    AA_REQUIRE(rows >= 0 && V > 0 && ld >= V && ldd >= V, "aa_logprob_gather_bwd: bad shape");
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_logprob_gather_bwd: dtype must be 0 (bf16) or 1 (f32)");
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(logprob_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, st,
                           (const bf16_t*)logits, ld, labels, lse, dlogp, (bf16_t*)dlogits, ldd, V);
    else
        hipLaunchKernelGGL(logprob_bwd_kernel<float>, dim3(rows), dim3(256), 0, st,
                           (const float*)logits, ld, labels, lse, dlogp, (float*)dlogits, ldd, V);
    AA_CHECK_LAUNCH("aa_logprob_gather_bwd");
    return AA_OK;
}

// ------------------------------------------------------------------ DPO loss (fwd + bwd fused)
// Sequences [0,B) are "better", [B,2B) "worse" (PreferenceCollator layout). Per-token log-probs are
// stored flat (no padding): sequence s owns rows [seq_off[s], seq_off[s+1]).
// out[0]=loss  out[1]=reward_accuracy  out[2]=mean reward  out[3]=mean better  out[4]=mean worse
// out[5]=mean margin ; per_sample [4,B] = better_reward, worse_reward, reward, margin
// dlogp[row] = d loss / d pol_logp[row]  (constant within a sequence).
__global__ __launch_bounds__(256) void dpo_loss_kernel(const float* __restrict__ pol,
                                                       const float* __restrict__ ref,
                                                       const int* __restrict__ seq_off, int B,
                                                       float beta, float* __restrict__ out,
                                                       float* __restrict__ per_sample,
                                                       float* __restrict__ dlogp,
                                                       const uint8_t* __restrict__ keep) {
    __shared__ float red[8];
    __shared__ float sums[2];
    float loss = 0.f, acc = 0.f, rsum = 0.f, bsum = 0.f, wsum = 0.f, msum = 0.f;
    int kept = B;
    if (keep) { kept = 0; for (int i = 0; i < B; ++i) kept += keep[i] ? 1 : 0; }
    const float invk = kept > 0 ? 1.f / (float)kept : 0.f;
    for (int i = 0; i < B; ++i) {
        if (keep && !keep[i]) {   // trainers/text_audio_to_text/dpo.py:139-140: identical chosen / rejected rows are skipped
            if (dlogp) {
                for (int t = seq_off[i] + threadIdx.x; t < seq_off[i + 1]; t += 256) dlogp[t] = 0.f;
                for (int t = seq_off[i + B] + threadIdx.x; t < seq_off[i + B + 1]; t += 256) dlogp[t] = 0.f;
            }
            if (threadIdx.x == 0 && per_sample) per_sample[i] = per_sample[B + i] = per_sample[2 * B + i] = per_sample[3 * B + i] = 0.f;
            continue;
        }
        float lr[2];
        for (int h = 0; h < 2; ++h) {
            const int s = i + h * B;
            float p = 0.f;
            for (int t = seq_off[s] + threadIdx.x; t < seq_off[s + 1]; t += 256) p += pol[t] - ref[t];
            lr[h] = block_sum<256>(p, red);  // log-ratio pi - ref summed over the response window
        }
        const float z = beta * (lr[0] - lr[1]);
        // -logsigmoid(z) = softplus(-z), stable form
        const float li = fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));
        const float sg = 1.f / (1.f + expf(z));  // sigmoid(-z)
        const float gb = -beta * sg * invk;
        if (dlogp) {
            for (int t = seq_off[i] + threadIdx.x; t < seq_off[i + 1]; t += 256) dlogp[t] = gb;
            for (int t = seq_off[i + B] + threadIdx.x; t < seq_off[i + B + 1]; t += 256) dlogp[t] = -gb;
        }
        const float br = beta * lr[0], wr = beta * lr[1];
        if (threadIdx.x == 0 && per_sample) {
            per_sample[i] = br;
            per_sample[B + i] = wr;
            per_sample[2 * B + i] = br + wr;
            per_sample[3 * B + i] = br - wr;
        }
        loss += li; acc += (br > wr) ? 1.f : 0.f; rsum += br + wr; bsum += br; wsum += wr; msum += br - wr;
    }
    if (threadIdx.x == 0) {
        const float inv = invk;
        out[0] = loss * inv; out[1] = acc * inv; out[2] = rsum * inv;
        out[3] = bsum * inv; out[4] = wsum * inv; out[5] = msum * inv;
    }
    (void)sums;
}

extern "C" int aa_dpo_loss_fwd_bwd(const float* pol_logp, const float* ref_logp, const int* seq_off,
                                   int B, float beta, float* out6, float* per_sample4B,
                                   float* dlogp, const uint8_t* keep, void* stream) {
    AA_REQUIRE(B > 0, "aa_dpo_loss_fwd_bwd: B must be > 0 (got %d)", B);
    hipLaunchKernelGGL(dpo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pol_logp, ref_logp,
                       seq_off, B, beta, out6, per_sample4B, dlogp, keep);
    AA_CHECK_LAUNCH("aa_dpo_loss_fwd_bwd");
    return AA_OK;
}

// ------------------------------------------------------------------ PPO: KL-shaped reward
// rewards[b,t] = clamp(-kl_coeff*(logp-ref)[b,t] + (t == end_b ? reward_b : 0), +-clip)
// end_b = last index with mask != 0 (ppo.py:536). One wave per row.
__global__ __launch_bounds__(64) void kl_reward_kernel(const float* __restrict__ reward,
                                                       const float* __restrict__ logp,
                                                       const float* __restrict__ ref,
                                                       const uint8_t* __restrict__ mask, int L,
                                                       float kl_coeff, float clip,
                                                       float* __restrict__ out, int* __restrict__ end_out) {
    const int b = blockIdx.x, l = threadIdx.x;
    int last = -1;
    for (int t = l; t < L; t += 64) if (mask[(long)b * L + t]) last = t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (l == 0 && end_out) end_out[b] = last;
    const float r = reward[b];
    for (int t = l; t < L; t += 64) {
        const long i = (long)b * L + t;
        float v = -kl_coeff * (logp[i] - ref[i]);
        if (t == last) v += r;
        out[i] = fminf(fmaxf(v, -clip), clip);
    }
}

extern "C" int aa_kl_reward(const float* reward, const float* logp, const float* ref_logp,
                            const uint8_t* mask, int B, int L, float kl_coeff, float clip,
                            float* rewards_out, int* end_index_out, void* stream) {
    AA_REQUIRE(B > 0 && L > 0, "aa_kl_reward: bad shape B=%d L=%d", B, L);
    hipLaunchKernelGGL(kl_reward_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, reward, logp,
                       ref_logp, mask, L, kl_coeff, clip, rewards_out, end_index_out);
    AA_CHECK_LAUNCH("aa_kl_reward");
    return AA_OK;
}

// ------------------------------------------------------------------ PPO: GAE reverse scan
// values/rewards pre-multiplied by mask; delta_t = r_t + gamma*v_{t+1} - v_t;
// A_t = delta_t + gamma*lambda*A_{t+1}; ret = A + v[:, start:].  One lane per row (sequential in t,
// parallel in B) -- the scan is 2 FMA per step, latency-bound, L <= 2048.
__global__ __launch_bounds__(64) void gae_kernel(const float* __restrict__ values,
                                                 const float* __restrict__ rewards,
                                                 const uint8_t* __restrict__ mask, int B, int L,
                                                 int start, float gamma, float lam,
                                                 float* __restrict__ adv, float* __restrict__ ret) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const long base = (long)b * L;
    const int W = L - start;
    float last = 0.f, nextv = 0.f;
    for (int t = L - 1; t >= start; --t) {
        const float mk = mask[base + t] ? 1.f : 0.f;
        const float v = values[base + t] * mk;
        const float r = rewards[base + t] * mk;
        const float delta = r + gamma * nextv - v;
        last = delta + gamma * lam * last;
        adv[(long)b * W + (t - start)] = last;
        ret[(long)b * W + (t - start)] = last + v;
        nextv = v;
    }
}

extern "C" int aa_gae(const float* values, const float* rewards, const uint8_t* mask, int B, int L,
                      int start, float gamma, float lam, float* adv, float* ret, void* stream) {
    AA_REQUIRE(B > 0 && L > 0 && start >= 0 && start < L, "aa_gae: bad shape B=%d L=%d start=%d", B, L, start);
    hipLaunchKernelGGL(gae_kernel, dim3(aa_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, values,
                       rewards, mask, B, L, start, gamma, lam, adv, ret);
    AA_CHECK_LAUNCH("aa_gae");
    return AA_OK;
}

// ------------------------------------------------------------------ PPO losses (fwd + bwd)
// masked_mean(x, m) = mean_b( sum_t x*m / sum_t m )   (utils/tools.py:460-467)
// actor : loss = -masked_mean(min(A*rho, A*clip(rho,1-e,1+e))), rho = exp(logp-old)
// critic: loss = 0.5*masked_mean(max((v-ret)^2, (clip(v,old-c,old+c)-ret)^2))
// One wave per row; per-row partial written to row_out[b]; loss = sum_b row_out[b]/B (second tiny
// kernel) so the result is deterministic.
__global__ __launch_bounds__(64) void actor_loss_kernel(const float* __restrict__ logp,
                                                        const float* __restrict__ old,
                                                        const float* __restrict__ adv,
                                                        const uint8_t* __restrict__ mask, int B,
                                                        int L, float eps, float* __restrict__ row_out,
                                                        float* __restrict__ dlogp) {
    const int b = blockIdx.x, l = threadIdx.x;
    const long base = (long)b * L;
    float num = 0.f, cnt = 0.f;
    for (int t = l; t < L; t += 64) {
        const float mk = mask[base + t] ? 1.f : 0.f;
        const float rho = expf(logp[base + t] - old[base + t]);
        const float a = adv[base + t];
        const float s1 = a * rho, s2 = a * fminf(fmaxf(rho, 1.f - eps), 1.f + eps);
        num += fminf(s1, s2) * mk;
        cnt += mk;
    }
    num = wave_sum(num); cnt = wave_sum(cnt);
    if (l == 0) row_out[b] = -(num / cnt) / (float)B;
    if (dlogp) {
        const float sc = -1.f / (cnt * (float)B);
        for (int t = l; t < L; t += 64) {
            const float mk = mask[base + t] ? 1.f : 0.f;
            const float rho = expf(logp[base + t] - old[base + t]);
            const float a = adv[base + t];
            const float s1 = a * rho, s2 = a * fminf(fmaxf(rho, 1.f - eps), 1.f + eps);
            // torch.minimum passes the gradient to s1 when s1 <= s2 (ties: both get 0.5 in torch;
            // a tie here means rho unclipped so d s2 = d s1 and the result is identical)
            float g;
            if (s1 <= s2) g = a * rho;
            else g = (rho > 1.f - eps && rho < 1.f + eps) ? a * rho : 0.f;
            dlogp[base + t] = sc * g * mk;
        }
    }
}

__global__ __launch_bounds__(64) void critic_loss_kernel(const float* __restrict__ values,
                                                         const float* __restrict__ old,
                                                         const float* __restrict__ ret,
                                                         const uint8_t* __restrict__ mask, int B,
                                                         int L, float clipv,
                                                         float* __restrict__ row_out,
                                                         float* __restrict__ dvalues) {
    const int b = blockIdx.x, l = threadIdx.x;
    const long base = (long)b * L;
    float num = 0.f, cnt = 0.f;
    for (int t = l; t < L; t += 64) {
        const float mk = mask[base + t] ? 1.f : 0.f;
        const float v = values[base + t], o = old[base + t], r = ret[base + t];
        const float vc = fminf(fmaxf(v, o - clipv), o + clipv);
        const float l1 = (v - r) * (v - r), l2 = (vc - r) * (vc - r);
        num += fmaxf(l1, l2) * mk;
        cnt += mk;
    }
    num = wave_sum(num); cnt = wave_sum(cnt);
    if (l == 0) row_out[b] = 0.5f * (num / cnt) / (float)B;
    if (dvalues) {
        const float sc = 0.5f / (cnt * (float)B);
        for (int t = l; t < L; t += 64) {
            const float mk = mask[base + t] ? 1.f : 0.f;
            const float v = values[base + t], o = old[base + t], r = ret[base + t];
            const float vc = fminf(fmaxf(v, o - clipv), o + clipv);
            const float l1 = (v - r) * (v - r), l2 = (vc - r) * (vc - r);
            float g;
            if (l1 >= l2) g = 2.f * (v - r);
            else g = (v > o - clipv && v < o + clipv) ? 2.f * (vc - r) : 0.f;
            dvalues[base + t] = sc * g * mk;
        }
    }
}

__global__ __launch_bounds__(64) void row_total_kernel(const float* __restrict__ row, int B,
                                                       float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += row[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

extern "C" int aa_ppo_actor_loss(const float* logp, const float* old_logp, const float* adv,
                                 const uint8_t* mask, int B, int L, float clip_ratio,
                                 float* row_scratch, float* loss_out, float* dlogp, void* stream) {
    AA_REQUIRE(B > 0 && L > 0, "aa_ppo_actor_loss: bad shape B=%d L=%d", B, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(actor_loss_kernel, dim3(B), dim3(64), 0, st, logp, old_logp, adv, mask, B, L,
                       clip_ratio, row_scratch, dlogp);
    hipLaunchKernelGGL(row_total_kernel, dim3(1), dim3(64), 0, st, row_scratch, B, loss_out);
    AA_CHECK_LAUNCH("aa_ppo_actor_loss");
    return AA_OK;
}

extern "C" int aa_ppo_critic_loss(const float* values, const float* old_values, const float* returns,
                                  const uint8_t* mask, int B, int L, float clip_value,
                                  float* row_scratch, float* loss_out, float* dvalues, void* stream) {
    AA_REQUIRE(B > 0 && L > 0, "aa_ppo_critic_loss: bad shape B=%d L=%d", B, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(critic_loss_kernel, dim3(B), dim3(64), 0, st, values, old_values, returns, mask,
                       B, L, clip_value, row_scratch, dvalues);
    hipLaunchKernelGGL(row_total_kernel, dim3(1), dim3(64), 0, st, row_scratch, B, loss_out);
    AA_CHECK_LAUNCH("aa_ppo_critic_loss");
    return AA_OK;
}

// ------------------------------------------------------------------ response-window labels
// align_anything/trainers/text_to_text/dpo.py:131-137: raw = strip_pad(ids[n]); labels = raw[-R:][1:].
// Pure integer work, reproduced exactly (including pad ids inside the text being dropped): a non-pad
// token whose inclusive non-pad suffix count is `rank` (1 = last token) is element R - rank of raw[-R:],
// i.e. label slot R - 1 - rank for 1 <= rank <= R - 1.  One workgroup per sequence, scanning from the end.
__global__ __launch_bounds__(256) void window_labels_kernel(const int64_t* __restrict__ ids, int T,
                                                            int64_t pad_id,
                                                            const int* __restrict__ resp_len,
                                                            const int* __restrict__ row_off,
                                                            int64_t* __restrict__ labels) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int n = blockIdx.x;
    const int R = resp_len[n];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 256) {
        const int k = base + threadIdx.x;  // distance from the end
        const int p = T - 1 - k;
        const int64_t tok = (p >= 0) ? ids[(long)n * T + p] : pad_id;
        const int f = (p >= 0 && tok != pad_id) ? 1 : 0;
        const unsigned long long bal = __ballot(f);
        const int incl = __popcll(bal & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
        if (lane == 0) wsum[wid] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int c = carry;
        const int rank = c + woff + incl;
        if (f && rank >= 1 && rank <= R - 1) labels[row_off[n] + (R - 1 - rank)] = tok;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
        if (carry >= R) break;  // uniform: carry is shared
    }
}

extern "C" int aa_window_labels(const int64_t* ids, int N, int T, int64_t pad_id, const int* resp_len,
                                const int* row_off, int64_t* labels, void* stream) {
    AA_REQUIRE(N > 0 && T > 0, "aa_window_labels: bad shape N=%d T=%d", N, T);
    hipLaunchKernelGGL(window_labels_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ids, T, pad_id,
                       resp_len, row_off, labels);
    AA_CHECK_LAUNCH("aa_window_labels");
    return AA_OK;
}

// ------------------------------------------------------------------ reward-model pairwise loss (fwd + bwd)
// align_anything/trainers/text_to_text/rm.py:97-132: end scores [2B] (higher = [0,B), lower = [B,2B)):
//   loss = mean(-logsigmoid(h - l)) + reg * mean(concat(l, h)^2) ; accuracy = mean(h > l)
// out[0] = loss, out[1] = accuracy ; dscores[2B] = d loss / d end score.
__global__ __launch_bounds__(256) void rm_loss_kernel(const float* __restrict__ end_scores, int B, float reg,
                                                      float* __restrict__ out, float* __restrict__ dscores) {
    __shared__ float red[8];
    float l = 0.f, a = 0.f, sq = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float h = end_scores[i], w = end_scores[B + i];
        const float z = h - w;
        l += fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));
        a += (h > w) ? 1.f : 0.f;
        sq += h * h + w * w;
        if (dscores) {
            const float sg = 1.f / (1.f + expf(z));  // sigmoid(-z)
            dscores[i] = -sg / (float)B + reg * 2.f * h / (float)(2 * B);
            dscores[B + i] = sg / (float)B + reg * 2.f * w / (float)(2 * B);
        }
    }
    l = block_sum<256>(l, red);
    a = block_sum<256>(a, red);
    sq = block_sum<256>(sq, red);
    if (threadIdx.x == 0) {
        out[0] = l / (float)B + reg * sq / (float)(2 * B);
        out[1] = a / (float)B;
    }
}
extern "C" int aa_rm_loss_fwd_bwd(const float* end_scores, int B, float regularization, float* out2,
                                  float* dscores, void* stream) {
    AA_REQUIRE(B > 0, "aa_rm_loss_fwd_bwd: B must be > 0 (got %d)", B);
    hipLaunchKernelGGL(rm_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, end_scores, B, regularization,
                       out2, dscores);
    AA_CHECK_LAUNCH("aa_rm_loss_fwd_bwd");
    return AA_OK;
}

// ------------------------------------------------------------------ GRPO (trainers/text_to_text/grpo.py:257-329)
// group-normalised advantage: rewards [B, G] -> (r - mean_g) / (std_g + 1e-4), std unbiased (torch.std, grpo.py:272-276)
__global__ __launch_bounds__(64) void group_advantage_kernel(const float* __restrict__ rewards, int G,
                                                             float* __restrict__ adv) {
    const int b = blockIdx.x, l = threadIdx.x;
    float s = 0.f;
    for (int i = l; i < G; i += 64) s += rewards[b * G + i];
    const float mean = wave_sum(s) / (float)G;
    float ss = 0.f;
    for (int i = l; i < G; i += 64) { const float d = rewards[b * G + i] - mean; ss += d * d; }
    const float sd = sqrtf(wave_sum(ss) / (float)(G - 1)) + 1e-4f;
    for (int i = l; i < G; i += 64) adv[b * G + i] = (rewards[b * G + i] - mean) / sd;
}
extern "C" int aa_group_advantage(const float* rewards, int B, int G, float* adv, void* stream) {
    AA_REQUIRE(B > 0 && G > 1, "aa_group_advantage: need B > 0 and G > 1 (got B=%d G=%d)", B, G);
    hipLaunchKernelGGL(group_advantage_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, rewards, G, adv);
    AA_CHECK_LAUNCH("aa_group_advantage");
    return AA_OK;
}

// completion mask: 1 up to and including the first eos of each row, 0 after (grpo.py:305-313)
__global__ __launch_bounds__(64) void completion_mask_kernel(const int64_t* __restrict__ tok, long ld, int L,
                                                             int64_t eos, uint8_t* __restrict__ mask) {
    const int r = blockIdx.x, l = threadIdx.x;
    int first = L;  // no eos -> everything counts
    for (int j = l; j < L; j += 64) if (tok[(long)r * ld + j] == eos) first = min(first, j);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    for (int j = l; j < L; j += 64) mask[(long)r * L + j] = (j <= first) ? 1 : 0;
}
extern "C" int aa_completion_mask(const int64_t* tokens, long ld, int rows, int L, int64_t eos, uint8_t* mask,
                                  void* stream) {
    AA_REQUIRE(rows > 0 && L > 0 && ld >= L, "aa_completion_mask: bad shape rows=%d L=%d", rows, L);
    hipLaunchKernelGGL(completion_mask_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, tokens, ld, L, eos, mask);
    AA_CHECK_LAUNCH("aa_completion_mask");
    return AA_OK;
}

// loss = sum(mask * -(A - beta*KL)) / sum(mask), KL = exp(ref-logp) - (ref-logp) - 1   (value of
// exp(logp - stopgrad(logp)) is 1; its gradient carries A).  d loss/d logp = mask*(-A + beta*(1 - exp(ref-logp)))/sum(mask)
__global__ __launch_bounds__(64) void grpo_rows_kernel(const float* __restrict__ logp, const float* __restrict__ ref,
                                                       const float* __restrict__ adv, const uint8_t* __restrict__ mask,
                                                       int L, float beta, float* __restrict__ row_num,
                                                       float* __restrict__ row_cnt) {
    const int r = blockIdx.x, l = threadIdx.x;
    float num = 0.f, cnt = 0.f;
    const float a = adv[r];
    for (int j = l; j < L; j += 64) {
        const long i = (long)r * L + j;
        const float mk = mask[i] ? 1.f : 0.f;
        const float d = ref[i] - logp[i];
        num += mk * (-(a - beta * (expf(d) - d - 1.f)));
        cnt += mk;
    }
    num = wave_sum(num); cnt = wave_sum(cnt);
    if (l == 0) { row_num[r] = num; row_cnt[r] = cnt; }
}
__global__ __launch_bounds__(256) void grpo_finish_kernel(const float* __restrict__ logp, const float* __restrict__ ref,
                                                          const float* __restrict__ adv, const uint8_t* __restrict__ mask,
                                                          int rows, int L, float beta, const float* __restrict__ row_num,
                                                          const float* __restrict__ row_cnt, float* __restrict__ loss,
                                                          float* __restrict__ dlogp) {
    __shared__ float red[8];
    float n = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) { n += row_num[i]; c += row_cnt[i]; }
    n = block_sum<256>(n, red);
    c = block_sum<256>(c, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) loss[0] = n / c;
    if (dlogp) {
        const float inv = 1.f / c;
        const long total = (long)rows * L;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const float mk = mask[i] ? 1.f : 0.f;
            const float a = adv[i / L];
            dlogp[i] = mk * (-a + beta * (1.f - expf(ref[i] - logp[i]))) * inv;
        }
    }
}
extern "C" int aa_grpo_loss_fwd_bwd(const float* logp, const float* ref_logp, const float* adv, const uint8_t* mask,
                                    int rows, int L, float beta, float* row_scratch2, float* loss_out, float* dlogp,
                                    void* stream) {
    AA_REQUIRE(rows > 0 && L > 0, "aa_grpo_loss_fwd_bwd: bad shape rows=%d L=%d", rows, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grpo_rows_kernel, dim3(rows), dim3(64), 0, st, logp, ref_logp, adv, mask, L, beta, row_scratch2,
                       row_scratch2 + rows);
    const long total = (long)rows * L;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(grpo_finish_kernel, dim3(grid), dim3(256), 0, st, logp, ref_logp, adv, mask, rows, L, beta,
                       row_scratch2, row_scratch2 + rows, loss_out, dlogp);
    AA_CHECK_LAUNCH("aa_grpo_loss_fwd_bwd");
    return AA_OK;
}
