// Autoregressive decode kernels for the PPO rollout (replaces HF `generate`, called at
// align_anything/trainers/text_to_text/ppo.py:209-222 with GenerationConfig(max_length, temperature, top_p,
// repetition_penalty=1.0, do_sample=True)).  Decode is HBM-bound (every step streams all weights and the
// KV cache once), so these kernels are built for streaming, not for MFMA peak:
//   aa_gemm_skinny_bf16 : out[M<=16, N] = x[M,K] W[N,K]^T (+bias, +residual) -- one 16x16 MFMA column strip per
//                         workgroup, K split over its 4 waves, 16-B loads straight to registers (no LDS round
//                         trip for a stream that is read once), cross-wave reduction in LDS
//   aa_attn_decode      : one query per sequence against the token-major KV cache, online softmax, fp32
//   aa_argmax_rows / aa_sample_top_p : greedy and temperature / nucleus sampling on the logits rows
#include "aa_common.h"

#define LOG2E_D 1.4426950408889634f

// ------------------------------------------------------------------ skinny GEMM (M <= 16)
// PRO selects what the kernel does to its activation fragments on the way in -- the decode step's element-wise kernels folded
// into the weight stream they precede (x is a few KB, L2-resident; the VALU work hides under the HBM stream):
//   0: x as is
//   1: RMSNorm.  rmsnorm(x)[k] = x[k] * rstd * w[k] with a per-row scalar rstd, so the kernel multiplies bf16(x[k] * w[k]) into the
//      dot products, accumulates sum(x^2) from the very fragments it loads, and scales the finished dot product by rstd in the
//      epilogue.  Rounding points differ from hf LlamaRMSNorm (w * bf16(x * rstd)) by bf16 noise; rollout sampling only.
//   2: SwiGLU.  x = [gate | up] rows of width 2K; the fragment is bf16(bf16(silu(gate)) * up), exactly aa_swiglu_fwd's value.
//   3: x as is, W pre-arranged by aa_swizzle_weights_bf16 so that every fragment load of a wave is 1 KB contiguous (below).
//   4: RMSNorm on strip-major weights whose COLUMNS were multiplied by the norm weight when the copy was made (aa_swizzle_weights_scaled_bf16;
//      the weights are frozen during a rollout): rmsnorm(x) W^T = rstd * (x (W diag(w))^T), so the prologue only accumulates sum(x^2) from the
//      fragments it feeds to the MFMA unchanged (no norm-weight loads, no extra registers: 8 k-steps stay in flight) and the epilogue scales the
//      finished dot product by rstd.  Removes the RMSNorm launch in front of the q/k/v, gate/up and lm_head projections: 57 of the 211 launches
//      of a Qwen2-VL-7B decode position, ~4.8 us each whatever their size (profiles/r04_decode_trace_summary.txt).  Rounding: bf16(W w) in
//      place of w * bf16(x rstd) -- bf16 noise on the logits, rollout sampling only (AA_DECODE_NORM_FOLD=0 restores the separate kernel).
template <int PRO>
__device__ __forceinline__ bf16x8 skinny_x(const bf16_t* __restrict__ xp, const bf16_t* __restrict__ nwp, int k, int K, float& ss) {
    const u16x8 v = *reinterpret_cast<const u16x8*>(xp + k);
    if constexpr (PRO == 1) {
        const u16x8 nw = *reinterpret_cast<const u16x8*>(nwp + k);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            ss += f * f;
            o[j] = f2bf(f * bf2f(nw[j]));
        }
        return __builtin_bit_cast(bf16x8, o);
    } else if constexpr (PRO == 2) {
        const u16x8 u = *reinterpret_cast<const u16x8*>(xp + K + k);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = bf2f(v[j]);
            o[j] = f2bf(rbf(gf * aa_sigmoid<false>(gf)) * bf2f(u[j]));
        }
        return __builtin_bit_cast(bf16x8, o);
    } else if constexpr (PRO == 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            ss += f * f;
        }
        return __builtin_bit_cast(bf16x8, v);
    } else {
        return __builtin_bit_cast(bf16x8, v);
    }
}

// S k-steps of 32: all 2S fragment loads are issued before the first MFMA consumes one
template <int PRO, int S>
__device__ __forceinline__ void skinny_trip(const bf16_t* __restrict__ wp, const bf16_t* __restrict__ xp, const bf16_t* __restrict__ nwp,
                                            int k, int K, float& ss, f32x4& acc0, f32x4& acc1) {
    bf16x8 wf[S], xf[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        wf[s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((PRO == 3 || PRO == 4) ? (long)(k + s * 32) * 16 : (long)(k + s * 32))));
        xf[s] = skinny_x<PRO>(xp, nwp, k + s * 32, K, ss);
    }
#pragma unroll
    for (int s = 0; s < S; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], xf[s + 1], acc1, 0, 0, 0);
    }
}

// The same k-steps in groups of four, software-pipelined (PIPE): the next group's fragments are requested BEFORE the MFMAs of the group that has arrived,
// so a wave never drains its load queue between trips (a deep strip -- the down projection, 1184 k per wave -- is nine such trips).  Same k order and the
// same accumulator per step parity as skinny_trip: bit-identical sums.
template <int PRO>
__device__ __forceinline__ void skinny_load4(const bf16_t* __restrict__ wp, const bf16_t* __restrict__ xp, const bf16_t* __restrict__ nwp,
                                             int k, int K, float& ss, bf16x8 (&wf)[4], bf16x8 (&xf)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        wf[s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((PRO == 3 || PRO == 4) ? (long)(k + s * 32) * 16 : (long)(k + s * 32))));
        xf[s] = skinny_x<PRO>(xp, nwp, k + s * 32, K, ss);
    }
}
__device__ __forceinline__ void skinny_mfma4(const bf16x8 (&wf)[4], const bf16x8 (&xf)[4], f32x4& acc0, f32x4& acc1) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], xf[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], xf[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], xf[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3], xf[3], acc1, 0, 0, 0);
}

// EPI: what the strip's 16 finished columns turn into (strip-major weights only, whose row ORDER inside a strip is free):
//   0: out[m, n0 + 0..15]  (+ bias, + residual)
//   1: SwiGLU.  The strip holds gate columns 8 s .. 8 s + 7 and the up values of the SAME columns (aa_swizzle_weights_perm_bf16 mode 1 of
//      the fused [gate; up] weight), so the thread that owns (m, c) has both and writes act[m, 8 s + c] = bf16(bf16(silu(bf16 gate)) *
//      bf16 up): aa_swiglu_fwd's value bit for bit, without the [M, 2F] round trip and without that kernel's launch.
//   2: rotary embedding + KV-cache write of the new token (head_dim 128).  The strip holds d = 8 s' .. + 7 and d + 64 of one head of the
//      fused [q | k | v] projection (mode 2), i.e. both members of every rotation pair: q heads are rotated and written to out[m, head *
//      128 + d], k heads rotated and written to the cache slot of this position, v heads copied there -- aa_decode_rope_cache's
//      arithmetic (bf16(bf16(x cos) + bf16(rotate_half(x) sin)) on the bf16-rounded projection) without its launch.
struct SkinnyEpi {
    const int* pos; const bf16_t* cos_t; const bf16_t* sin_t; bf16_t* cache; long ldc; int Tmax; const int64_t* slot; int H, Hkv;
};

template <int NWAVE, int PRO, int EPI, bool PIPE = false>
__global__ __launch_bounds__(NWAVE * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                  const bf16_t* __restrict__ W, long ldw,
                                                                  bf16_t* __restrict__ out, long ldo,
                                                                  const bf16_t* __restrict__ bias,
                                                                  const bf16_t* __restrict__ residual, long ldr,
                                                                  int M, int N, int K,
                                                                  const int* __restrict__ row_expert, long strideE, int x_div,
                                                                  const bf16_t* __restrict__ norm_w, float eps, const SkinnyEpi epi) {
    __shared__ float red[NWAVE][16][17];
    __shared__ float ssred[NWAVE][16];
    if (row_expert) {   // mixture-of-experts decode: blockIdx.y = routed row r = (token, choice); its own weight matrix, M = 1
        const int r = blockIdx.y;
        W += (long)row_expert[r] * strideE;
        x += (long)(r / x_div) * ldx;
        out += (long)r * ldo;
    }
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int nrow = min(n0 + l15, N - 1);
    const int mrow = min(l15, M - 1);
    // PRO 3 = strip-major swizzled weights: [N/16][K/32][lane = n%16 + 16*(k%32/8)][8] -- a wave's fragment load is 1 KB contiguous
    constexpr bool SWZ = PRO == 3 || PRO == 4, NORM = PRO == 1 || PRO == 4;
    const bf16_t* wp = SWZ ? W + (long)blockIdx.x * 16 * K + lane * 8 : W + (long)nrow * ldw + g * 8;
    const bf16_t* xp = x + (long)mrow * ldx + g * 8;
    const bf16_t* nwp = PRO == 1 ? norm_w + g * 8 : nullptr;
    float ss = 0.f;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // wave w owns k in [w*kq, (w+1)*kq), kq = K/NWAVE rounded up to 32; 8 MFMA k-steps (256 k) per trip keep
    // 16 x 16-B loads per lane in flight (the weight stream is read exactly once: no LDS round trip)
    const int kq = ((K / NWAVE + 31) / 32) * 32;
    const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
    int k = k_lo;
    // fragments in flight per trip: 8 k-steps (256 k); 4 with the RMSNorm prologue (its raw x, norm weight and product
    // fragments would otherwise push the kernel past 128 VGPRs and halve the waves that keep the HBM queue full).
    // (16 in flight -- 512 k per trip -- measured no different: 4.19 vs 4.20 ms per position, tools/gpu_skinny_ab.sh.)
    constexpr int S = PRO == 1 ? 4 : 8;
    if constexpr (PIPE) {
        if (k + 128 <= k_hi) {
            bf16x8 wa[4], xa[4], wb[4], xb[4];
            skinny_load4<PRO>(wp, xp, nwp, k, K, ss, wa, xa);
            k += 128;
            // the scheduling fences keep hipcc from sinking a group's loads below the MFMAs of the other one (it otherwise re-uses the fragment registers
            // and drains the load queue to vmcnt(0) in the middle of the loop)
            for (; k + 256 <= k_hi; k += 256) {
                skinny_load4<PRO>(wp, xp, nwp, k, K, ss, wb, xb);
                __builtin_amdgcn_sched_barrier(0);
                skinny_mfma4(wa, xa, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
                skinny_load4<PRO>(wp, xp, nwp, k + 128, K, ss, wa, xa);
                __builtin_amdgcn_sched_barrier(0);
                skinny_mfma4(wb, xb, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k + 128 <= k_hi) {
                skinny_load4<PRO>(wp, xp, nwp, k, K, ss, wb, xb);
                __builtin_amdgcn_sched_barrier(0);
                skinny_mfma4(wa, xa, acc0, acc1);
                skinny_mfma4(wb, xb, acc0, acc1);
                k += 128;
            } else {
                skinny_mfma4(wa, xa, acc0, acc1);
            }
        }
    } else {
        for (; k + S * 32 <= k_hi; k += S * 32) skinny_trip<PRO, S>(wp, xp, nwp, k, K, ss, acc0, acc1);
        if constexpr (S > 4) { for (; k + 128 <= k_hi; k += 128) skinny_trip<PRO, 4>(wp, xp, nwp, k, K, ss, acc0, acc1); }   // a 16-wave strip's 224-k share: 4 + 2 + 1 steps
    }
    for (; k + 64 <= k_hi; k += 64) skinny_trip<PRO, 2>(wp, xp, nwp, k, K, ss, acc0, acc1);
    if (k < k_hi) {
        const bf16x8 wf = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (SWZ ? (long)k * 16 : (long)k)));
        const bf16x8 xf = skinny_x<PRO>(xp, nwp, k, K, ss);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc0, 0, 0, 0);
    }
    // D[i = n (4g + r)][j = m (l15)]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][l15] = acc0[r] + acc1[r];
    if constexpr (NORM) {          // lane (l15, g) holds row l15's sum of squares over its k-chunks: fold the 4 g groups
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) ssred[wave][l15] = ss;
    }
    __syncthreads();
    if constexpr (EPI != 0) {
        // thread t < 128 -> (m = t / 8, c = t % 8): columns c and c + 8 of the strip are a (gate, up) / (d, d + 64) pair
        const int m = threadIdx.x >> 3, c = threadIdx.x & 7;
        if (threadIdx.x < 128 && m < M) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) { v1 += red[w][c][m]; v2 += red[w][c + 8][m]; }
            if constexpr (NORM) {
                float q = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) q += ssred[w][m];
                const float rs = rsqrtf(q / (float)K + eps);
                v1 *= rs; v2 *= rs;
            }
            if constexpr (EPI == 1) {
                const float gf = rbf(v1), uf = rbf(v2);
                out[(long)m * ldo + blockIdx.x * 8 + c] = f2bf(rbf(gf * aa_sigmoid<false>(gf)) * uf);
            } else {
                const int head = blockIdx.x >> 3, d = (blockIdx.x & 7) * 8 + c;          // head_dim 128: 8 strips per head
                const int col = head * 128 + d;
                if (bias) { v1 += bf2f(bias[col]); v2 += bf2f(bias[col + 64]); }
                const float a = rbf(v1), b = rbf(v2);
                bf16_t* crow = epi.cache + ((long)m * epi.Tmax + epi.slot[m]) * epi.ldc;
                if (head >= epi.H + epi.Hkv) {                                            // value head: copy into the cache
                    bf16_t* dst = crow + (long)epi.Hkv * 128 + (long)(head - epi.H - epi.Hkv) * 128 + d;
                    dst[0] = f2bf(a);
                    dst[64] = f2bf(b);
                } else {
                    const long tb = (long)epi.pos[m] * 64 + d;
                    const float cc = bf2f(epi.cos_t[tb]), ss = bf2f(epi.sin_t[tb]);
                    const bf16_t o1 = f2bf(rbf(a * cc) + rbf(-b * ss)), o2 = f2bf(rbf(b * cc) + rbf(a * ss));
                    bf16_t* dst = head < epi.H ? out + (long)m * ldo + col : crow + (long)(head - epi.H) * 128 + d;
                    dst[0] = o1;
                    dst[64] = o2;
                }
            }
        }
        return;
    }
    // thread t < 256 -> (m = t / 16, n = t % 16)
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
    const int n = n0 + nn;
    if (threadIdx.x < 256 && m < M && n < N) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) v += red[w][nn][m];
        if constexpr (NORM) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) q += ssred[w][m];
            v *= rsqrtf(q / (float)K + eps);
        }
        if (bias) v += bf2f(bias[n]);
        if (residual) v = rbf(v) + bf2f(residual[(long)m * ldr + n]);
        out[(long)m * ldo + n] = f2bf(v);
    }
}

// Launch rules (aa_decode_set_rules; env AA_DECODE_R6 sets the initial mask).  0: the round-5 rules (16 waves for every narrow deep strip launch, two key
// steps in flight in the cache attention) -- same-box A/B.  bit 0: the round-6 wave rule + four key steps at a handful of sequences; bit 1: software-pipelined
// trips in the deep 16-wave strips (the down projection); bit 2: four key steps in flight for every launch of fewer than 512 workgroups (eight steps at a
// handful of sequences measured neutral: profiles/r06_decode_rules.txt).
static int g_decode_rules = -1;
static int decode_r6() {
    if (g_decode_rules < 0) { const char* e = getenv("AA_DECODE_R6"); g_decode_rules = e ? atoi(e) : 3; }
    return g_decode_rules;
}
extern "C" int aa_decode_set_rules(int mask, int* old) {
    AA_REQUIRE(mask >= 0 && mask <= 7, "aa_decode_set_rules: mask %d (bits 0 - 2)", mask);
    if (old) *old = decode_r6();
    g_decode_rules = mask;
    return AA_OK;
}

// compute units of the current device (cached per device)
static int skinny_cus() {
    static int cus_of[AA_MAX_DEVICES] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AA_MAX_DEVICES) return 256;
    if (cus_of[dev] == 0) {
        int cus = 0;
        cus_of[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    return cus_of[dev];
}

template <int PRO, int EPI = 0>
static void launch_skinny(const void* x, const void* W, void* out, int M, int N, int K, long ldx, long ldw, long ldo,
                          const void* bias, const void* residual, long ldr, const void* norm_w, float eps, hipStream_t st,
                          const SkinnyEpi epi = SkinnyEpi{}) {
    // few column strips (N/16 < ~3 per CU): split K over 8 waves so enough loads are in flight per CU
    const bool wide = (N / 16) >= 768 || K < 2048;
    // narrow AND deep (the o / down / qkv projections of a 7B decoder: 224-288 strips on 256 CUs, K >= 3584): 16 waves per strip -- the launch is
    // bound by its fixed costs, and twice the waves halve each wave's share of the stream (same-box PPO A/B: 3.181 -> 3.118 ms per position).
    // Only while every strip finds a CU at once: a 16-wave workgroup of 66 - 107 VGPRs has a CU to itself (4 - 7 waves per SIMD), so a launch of more
    // strips than CUs runs in TWO rounds -- the q/k/v projection of Qwen2-VL-7B (288 strips) took 12.3 us against 5.7 us for the o projection's 224
    // strips of the same depth (profiles/r05_ppo_kernel_stats.csv).  Such launches get 8 waves per strip (two workgroups per CU: one round).
    if (!wide && K >= 3584 && (aa_cdiv(N, 16) <= skinny_cus() || !(decode_r6() & 1))) {
        if (K >= 8192 && (decode_r6() & 2))      // deep: software-pipelined trips (bit-identical sums)
            hipLaunchKernelGGL((gemm_skinny_kernel<16, PRO, EPI, true>), dim3(aa_cdiv(N, 16)), dim3(1024), 0, st,
                               (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                               (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1, (const bf16_t*)norm_w, eps, epi);
        else
            hipLaunchKernelGGL((gemm_skinny_kernel<16, PRO, EPI>), dim3(aa_cdiv(N, 16)), dim3(1024), 0, st,
                               (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                               (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1, (const bf16_t*)norm_w, eps, epi);
        return;
    }
    if (wide)
        hipLaunchKernelGGL((gemm_skinny_kernel<4, PRO, EPI>), dim3(aa_cdiv(N, 16)), dim3(256), 0, st,
                           (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                           (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1, (const bf16_t*)norm_w, eps, epi);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<8, PRO, EPI>), dim3(aa_cdiv(N, 16)), dim3(512), 0, st,
                           (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                           (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1, (const bf16_t*)norm_w, eps, epi);
}

extern "C" int aa_gemm_skinny_bf16(const void* x, const void* W, void* out, int M, int N, int K, long ldx,
                                   long ldw, long ldo, const void* bias, const void* residual, long ldr,
                                   void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_bf16: M=%d must be in [1, 16] (use aa_gemm_bf16 beyond)", M);
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0, "aa_gemm_skinny_bf16: K=%d must be a multiple of 32", K);
    AA_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0, "aa_gemm_skinny_bf16: ldx/ldw must be multiples of 8");
    launch_skinny<0>(x, W, out, M, N, K, ldx, ldw, ldo, bias, residual, ldr, nullptr, 0.f, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_bf16");
    return AA_OK;
}

// The same weight stream with the preceding element-wise kernel of the decode step folded in (see PRO above):
// prologue 1 = RMSNorm(x; norm_w, eps) -> out = rmsnorm(x) W^T;  prologue 2 = SwiGLU, x = [gate | up] [M, 2K] -> out = swiglu(x) W^T.
extern "C" int aa_gemm_skinny_fused_bf16(const void* x, const void* W, void* out, int M, int N, int K, long ldx, long ldw,
                                         long ldo, const void* bias, const void* residual, long ldr, int prologue,
                                         const void* norm_w, float eps, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_fused_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0, "aa_gemm_skinny_fused_bf16: K=%d must be a multiple of 32", K);
    AA_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0, "aa_gemm_skinny_fused_bf16: ldx/ldw must be multiples of 8");
    AA_REQUIRE(prologue == 1 || prologue == 2, "aa_gemm_skinny_fused_bf16: prologue %d (1 = RMSNorm, 2 = SwiGLU)", prologue);
    AA_REQUIRE(prologue != 1 || norm_w != nullptr, "aa_gemm_skinny_fused_bf16: the RMSNorm prologue needs norm_w");
    AA_REQUIRE(prologue != 2 || ldx >= 2L * K, "aa_gemm_skinny_fused_bf16: the SwiGLU prologue reads [gate | up] rows of width 2K = %d", 2 * K);
    if (prologue == 1) launch_skinny<1>(x, W, out, M, N, K, ldx, ldw, ldo, bias, residual, ldr, norm_w, eps, (hipStream_t)stream);
    else launch_skinny<2>(x, W, out, M, N, K, ldx, ldw, ldo, bias, residual, ldr, nullptr, 0.f, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_fused_bf16");
    return AA_OK;
}

// Strip-major weight layout for the rollout.  In the row-major [N, K] matrix a wave's MFMA fragment load touches 16 rows x 64 B
// (half a cache line per row per instruction); measured, that access pattern -- not bytes in flight -- is what holds the strip
// kernel at 4.5 of 8 TB/s.  Weights do not change during a rollout (hundreds of positions per optimizer step), so `generate`
// re-arranges them once per call into [N/16 strips][K/32 blocks][lane = n%16 + 16*(k%32/8)][8 elements]: the very order the
// lanes consume, 1 KB contiguous per wave load.  Same operands, same MFMA order -> bit-identical results to the row-major kernel.
// mode: 0 = strip s holds rows 16 s .. 16 s + 15; 1 = rows of a fused [gate; up] weight (N = 2 F): strip s = gate rows 8 s .. + 7 then the up
// rows F + 8 s .. + 7 (EPI 1 of the strip kernel); 2 = head_dim-128 heads: strip 8 h + s' = rows 128 h + 8 s' .. + 7 then the rows 64 further
// (the rotation partners, EPI 2)
__global__ __launch_bounds__(256) void swizzle_weights_kernel(const bf16_t* __restrict__ W, long ld, bf16_t* __restrict__ out, int N, int K, int mode,
                                                              const bf16_t* __restrict__ kscale) {
    const long kblocks = K >> 5;
    const long total = (long)((N + 15) >> 4) * kblocks * 64;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const long t = idx >> 6;
        const long kb = t % kblocks, strip = t / kblocks;
        const int c = lane & 15;
        long n = strip * 16 + c;
        if (mode == 1) n = (c < 8 ? 0 : (long)(N >> 1)) + strip * 8 + (c & 7);
        else if (mode == 2) n = (strip >> 3) * 128 + (strip & 7) * 8 + (c & 7) + (c < 8 ? 0 : 64);
        const long k = kb * 32 + (lane >> 4) * 8;
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < N) v = *reinterpret_cast<const u16x8*>(W + n * ld + k);
        if (kscale != nullptr) {       // column scaling (PRO 4 of the strip kernel): W[n, k] * w_norm[k], one rounding
            const u16x8 sc = *reinterpret_cast<const u16x8*>(kscale + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * bf2f(sc[j]));
        }
        *reinterpret_cast<u16x8*>(out + idx * 8) = v;
    }
}
extern "C" int aa_swizzle_weights_bf16(const void* W, long ld, void* out, int N, int K, void* stream) {
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ld % 8 == 0, "aa_swizzle_weights_bf16: N=%d K=%d ld=%ld (K %% 32 == 0, ld %% 8 == 0)", N, K, ld);
    const long total = (long)((N + 15) >> 4) * (K >> 5) * 64;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(swizzle_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, N, K, 0, (const bf16_t*)nullptr);
    AA_CHECK_LAUNCH("aa_swizzle_weights_bf16");
    return AA_OK;
}
// The same re-arrangement with the rows of a strip permuted for a fused epilogue of the strip kernel: mode 1 = [gate; up] weight of a SwiGLU
// MLP (N = 2 F, F a multiple of 8) for aa_gemm_skinny_swz_glu_bf16, mode 2 = fused [q | k | v] projection with head_dim 128 (N a multiple of
// 128) for aa_gemm_skinny_swz_rope_cache_bf16.
extern "C" int aa_swizzle_weights_perm_bf16(const void* W, long ld, void* out, int N, int K, int mode, void* stream) {
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ld % 8 == 0, "aa_swizzle_weights_perm_bf16: N=%d K=%d ld=%ld (K %% 32 == 0, ld %% 8 == 0)", N, K, ld);
    AA_REQUIRE((mode == 1 && N % 16 == 0) || (mode == 2 && N % 128 == 0), "aa_swizzle_weights_perm_bf16: mode %d needs N=%d a multiple of %d", mode, N, mode == 1 ? 16 : 128);
    const long total = (long)(N >> 4) * (K >> 5) * 64;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(swizzle_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, N, K, mode, (const bf16_t*)nullptr);
    AA_CHECK_LAUNCH("aa_swizzle_weights_perm_bf16");
    return AA_OK;
}
extern "C" int aa_gemm_skinny_swz_bf16(const void* x, const void* Wswz, void* out, int M, int N, int K, long ldx, long ldo,
                                       const void* bias, const void* residual, long ldr, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_bf16: N=%d K=%d (K %% 32 == 0), ldx %% 8 == 0", N, K);
    launch_skinny<3>(x, Wswz, out, M, N, K, ldx, K, ldo, bias, residual, ldr, nullptr, 0.f, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_bf16");
    return AA_OK;
}

// hf LlamaMLP front half of a decode position in one launch: act[M, F] = silu(x Wg^T) * (x Wu^T) from the mode-1 strip-major copy of the
// fused [gate; up] weight -- bit-identical to aa_gemm_skinny_swz_bf16 followed by aa_swiglu_fwd.
extern "C" int aa_gemm_skinny_swz_glu_bf16(const void* x, const void* Wswz, void* act, int M, int F, int K, long ldx, long ldo, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_glu_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(F > 0 && F % 8 == 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_glu_bf16: F=%d (multiple of 8) K=%d (multiple of 32)", F, K);
    launch_skinny<3, 1>(x, Wswz, act, M, 2 * F, K, ldx, K, ldo, nullptr, nullptr, 0, nullptr, 0.f, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_glu_bf16");
    return AA_OK;
}
// q/k/v projection of a decode position with aa_decode_rope_cache in its epilogue (head_dim 128): q[M, H * 128] rotated, k rotated into and v
// copied into cache slot `slot[m]` of sequence m (cache rows [M * Tmax, ldc] = keys | values), from the mode-2 strip-major copy of the
// fused weight; bias = the fused [q | k | v] bias or null.  Bit-identical to aa_gemm_skinny_swz_bf16 + aa_decode_rope_cache.
extern "C" int aa_gemm_skinny_swz_rope_cache_bf16(const void* x, const void* Wswz, void* q_out, int M, int H, int Hkv, int K, long ldx, long ldq,
                                                  const void* bias, const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc,
                                                  int Tmax, const int64_t* slot, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_rope_cache_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(H > 0 && Hkv > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_rope_cache_bf16: H=%d Hkv=%d K=%d (multiple of 32)", H, Hkv, K);
    AA_REQUIRE(ldc >= 2L * Hkv * 128 && Tmax > 0 && ldq >= (long)H * 128, "aa_gemm_skinny_swz_rope_cache_bf16: ldc >= 2 * Hkv * 128, ldq >= H * 128");
    SkinnyEpi e{pos, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)cache, ldc, Tmax, slot, H, Hkv};
    launch_skinny<3, 2>(x, Wswz, q_out, M, (H + 2 * Hkv) * 128, K, ldx, K, ldq, bias, nullptr, 0, nullptr, 0.f, (hipStream_t)stream, e);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_rope_cache_bf16");
    return AA_OK;
}

// ---- RMSNorm folded into the strip kernel (PRO 4).  aa_swizzle_weights_scaled_bf16 makes the strip-major copy (mode 0 / 1 / 2 as above) of
// W diag(kscale), kscale = the weight [K] of the RMSNorm in front of the projection; the three entry points below are their namesakes without
// `_norm` applied to the UN-normalised residual stream x: out = rmsnorm(x; kscale, eps) W^T (+ bias, + residual / SwiGLU / rotary + cache write),
// hf LlamaRMSNorm + the projection of hf LlamaDecoderLayer (hf:models/llama/modeling_llama.py:62-67) up to bf16 rounding of W kscale.
extern "C" int aa_swizzle_weights_scaled_bf16(const void* W, long ld, void* out, int N, int K, int mode, const void* kscale, void* stream) {
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ld % 8 == 0 && kscale != nullptr, "aa_swizzle_weights_scaled_bf16: N=%d K=%d ld=%ld (K %% 32 == 0, ld %% 8 == 0), kscale required", N, K, ld);
    AA_REQUIRE(mode == 0 || (mode == 1 && N % 16 == 0) || (mode == 2 && N % 128 == 0), "aa_swizzle_weights_scaled_bf16: mode %d does not fit N=%d", mode, N);
    const long total = (long)((N + 15) >> 4) * (K >> 5) * 64;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(swizzle_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ld, (bf16_t*)out, N, K, mode, (const bf16_t*)kscale);
    AA_CHECK_LAUNCH("aa_swizzle_weights_scaled_bf16");
    return AA_OK;
}
extern "C" int aa_gemm_skinny_swz_norm_bf16(const void* x, const void* Wswz, void* out, int M, int N, int K, long ldx, long ldo,
                                            const void* bias, const void* residual, long ldr, float eps, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_norm_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_norm_bf16: N=%d K=%d (K %% 32 == 0), ldx %% 8 == 0", N, K);
    launch_skinny<4>(x, Wswz, out, M, N, K, ldx, K, ldo, bias, residual, ldr, nullptr, eps, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_norm_bf16");
    return AA_OK;
}
extern "C" int aa_gemm_skinny_swz_norm_glu_bf16(const void* x, const void* Wswz, void* act, int M, int F, int K, long ldx, long ldo, float eps, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_norm_glu_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(F > 0 && F % 8 == 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_norm_glu_bf16: F=%d (multiple of 8) K=%d (multiple of 32)", F, K);
    launch_skinny<4, 1>(x, Wswz, act, M, 2 * F, K, ldx, K, ldo, nullptr, nullptr, 0, nullptr, eps, (hipStream_t)stream);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_norm_glu_bf16");
    return AA_OK;
}
extern "C" int aa_gemm_skinny_swz_norm_rope_cache_bf16(const void* x, const void* Wswz, void* q_out, int M, int H, int Hkv, int K, long ldx, long ldq,
                                                       const void* bias, const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc,
                                                       int Tmax, const int64_t* slot, float eps, void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_swz_norm_rope_cache_bf16: M=%d must be in [1, 16]", M);
    AA_REQUIRE(H > 0 && Hkv > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0, "aa_gemm_skinny_swz_norm_rope_cache_bf16: H=%d Hkv=%d K=%d (multiple of 32)", H, Hkv, K);
    AA_REQUIRE(ldc >= 2L * Hkv * 128 && Tmax > 0 && ldq >= (long)H * 128, "aa_gemm_skinny_swz_norm_rope_cache_bf16: ldc >= 2 * Hkv * 128, ldq >= H * 128");
    SkinnyEpi e{pos, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)cache, ldc, Tmax, slot, H, Hkv};
    launch_skinny<4, 2>(x, Wswz, q_out, M, (H + 2 * Hkv) * 128, K, ldx, K, ldq, bias, nullptr, 0, nullptr, eps, (hipStream_t)stream, e);
    AA_CHECK_LAUNCH("aa_gemm_skinny_swz_norm_rope_cache_bf16");
    return AA_OK;
}

// Mixture-of-experts decode: out[r, :] = x[r / x_div, :] W3[row_expert[r]]^T for R routed rows (a handful of tokens x top-k):
// every routed row streams its own expert matrix once (the floor for a row-private weight), no padding to the 128-row
// tiles of the grouped training GEMM.  hf:models/qwen3_moe/modeling_qwen3_moe.py:210-283 at one token per sequence.
extern "C" int aa_moe_gemv_bf16(const void* x, const void* W3, void* out, int R, int N, int K, long ldx, long ldw, long ldo,
                                const int* row_expert, long strideE, int x_div, void* stream) {
    AA_REQUIRE(R >= 1 && R <= 65535 && N > 0 && K > 0 && K % 32 == 0, "aa_moe_gemv_bf16: R=%d N=%d K=%d (K must be a multiple of 32)", R, N, K);
    AA_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && strideE % 8 == 0 && x_div >= 1 && row_expert != nullptr, "aa_moe_gemv_bf16: ldx/ldw/strideE must be multiples of 8");
    hipLaunchKernelGGL((gemm_skinny_kernel<4, 0, 0>), dim3(aa_cdiv(N, 16), R), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (const bf16_t*)W3, ldw, (bf16_t*)out, ldo, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, 0, 1, N, K, row_expert, strideE, x_div, (const bf16_t*)nullptr, 0.f, SkinnyEpi{});
    AA_CHECK_LAUNCH("aa_moe_gemv_bf16");
    return AA_OK;
}

// ------------------------------------------------------------------ RoPE + KV-cache write of the new token
// One pass over the fused [q | k | v] row of every sequence: q heads are rotated in place, k heads are rotated and written to
// the cache slot of this position, v heads are copied there -- aa_rope_inplace + an index_put, without the round trip.
// Rounding as aa_rope_inplace (hf apply_rotary_pos_emb in bf16): bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)).
__global__ __launch_bounds__(256) void decode_rope_cache_kernel(bf16_t* __restrict__ qkv, long ld, int H, int Hkv, int hd,
                                                                const int* __restrict__ pos, const bf16_t* __restrict__ cos_t,
                                                                const bf16_t* __restrict__ sin_t, bf16_t* __restrict__ cache,
                                                                long ldc, int Tmax, const int64_t* __restrict__ slot, int N) {
    const int half = hd >> 1, vph = half >> 3, heads = H + 2 * Hkv, kw = Hkv * hd;
    const long total = (long)N * heads * vph;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int v = (int)(idx % vph);
        const long t = idx / vph;
        const int head = (int)(t % heads);
        const int row = (int)(t / heads);
        bf16_t* base = qkv + (long)row * ld + (long)head * hd + v * 8;
        const u16x8 x1 = *reinterpret_cast<const u16x8*>(base);
        const u16x8 x2 = *reinterpret_cast<const u16x8*>(base + half);
        bf16_t* crow = cache + ((long)row * Tmax + slot[row]) * ldc;
        if (head >= H + Hkv) {                 // value head: copy
            bf16_t* dst = crow + kw + (long)(head - H - Hkv) * hd + v * 8;
            *reinterpret_cast<u16x8*>(dst) = x1;
            *reinterpret_cast<u16x8*>(dst + half) = x2;
            continue;
        }
        const int p = pos[row];
        const u16x8 c = *reinterpret_cast<const u16x8*>(cos_t + (long)p * half + v * 8);
        const u16x8 s = *reinterpret_cast<const u16x8*>(sin_t + (long)p * half + v * 8);
        u16x8 o1, o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = bf2f(x1[j]), b = bf2f(x2[j]), cc = bf2f(c[j]), ss = bf2f(s[j]);
            o1[j] = f2bf(rbf(a * cc) + rbf(-b * ss));
            o2[j] = f2bf(rbf(b * cc) + rbf(a * ss));
        }
        bf16_t* dst = head < H ? base : crow + (long)(head - H) * hd + v * 8;
        *reinterpret_cast<u16x8*>(dst) = o1;
        *reinterpret_cast<u16x8*>(dst + half) = o2;
    }
}
extern "C" int aa_decode_rope_cache(void* qkv, long ld, int N, int H, int Hkv, int hd, const int* pos, const void* cos_t,
                                    const void* sin_t, void* cache, long ldc, int Tmax, const int64_t* slot, void* stream) {
    AA_REQUIRE(N > 0 && H > 0 && Hkv > 0 && hd >= 16 && (hd & 15) == 0, "aa_decode_rope_cache: N=%d H=%d Hkv=%d head_dim %d (multiple of 16)", N, H, Hkv, hd);
    AA_REQUIRE((ld & 7) == 0 && (ldc & 7) == 0 && ldc >= 2L * Hkv * hd && Tmax > 0, "aa_decode_rope_cache: ld / ldc must be multiples of 8, ldc >= 2*Hkv*hd");
    const long total = (long)N * (H + 2 * Hkv) * (hd >> 4);
    hipLaunchKernelGGL(decode_rope_cache_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv, ld, H, Hkv,
                       hd, pos, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)cache, ldc, Tmax, slot, N);
    AA_CHECK_LAUNCH("aa_decode_rope_cache");
    return AA_OK;
}

// ------------------------------------------------------------------ decode attention
// q [N, H*HD] (one new token per sequence), caches Kc/Vc token-major [N, Tmax, Hkv*HD] (row stride ldc),
// valid keys of sequence n: [start[n], len[n]).  One workgroup per (head, sequence); HD/8 lanes per key,
// 64/(HD/8) keys per wave step, NW waves stride the keys two steps at a time (both steps' K and V rows are requested before
// either is used: the loop is a chain of dependent online-softmax updates, so memory parallelism has to come from the
// loads); partial (m, l, acc) merged through LDS.  NW = 8 when H*N alone would leave CUs idle and waves scarce
// (a 4-sequence rollout of a 32-head model is 128 workgroups on 256 CUs), 4 otherwise.
template <int HD, int NW, int U>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const bf16_t* __restrict__ q, long ldq,
                                                              const bf16_t* __restrict__ Kc,
                                                              const bf16_t* __restrict__ Vc, long ldc, int Tmax,
                                                              const int* __restrict__ start,
                                                              const int* __restrict__ len, bf16_t* __restrict__ o,
                                                              long ldo, int H, int Hkv, float scale) {
    constexpr int LPK = HD / 8;        // lanes per key
    constexpr int KPW = 64 / LPK;      // keys per wave step
    constexpr int STRIDE = NW * KPW;   // keys per workgroup step
    __shared__ float sm_m[NW][KPW], sm_l[NW][KPW];
    __shared__ float sm_acc[NW][KPW][HD];
    const int h = blockIdx.x, n = blockIdx.y, hk = h / (H / Hkv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPK, kg = lane / LPK;
    const int s0 = start ? start[n] : 0, s1 = len[n];
    float qv[8];
    {
        const u16x8 t = *reinterpret_cast<const u16x8*>(q + (long)n * ldq + h * HD + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = bf2f(t[j]) * scale * LOG2E_D;
    }
    float m = -INFINITY, l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16_t* kb = Kc + (long)n * Tmax * ldc + hk * HD + sub * 8;
    const bf16_t* vb = Vc + (long)n * Tmax * ldc + hk * HD + sub * 8;
    // U wave steps per trip (2, or 4 when the launch is a handful of workgroups: one sequence of a 28-head model is 28 workgroups walking ~800 keys in
    // trips of dependent loads -- the trip count, not the bytes, is its time): all 2 U rows are requested before any is used
    for (int j0 = s0 + wave * KPW; j0 < s1; j0 += U * STRIDE) {
        bool ok[U];
        u16x8 kf[U], vf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + kg + u * STRIDE;
            ok[u] = j < s1;
            const long r = ok[u] ? j : s1 - 1;
            kf[u] = *reinterpret_cast<const u16x8*>(kb + r * ldc);
            vf[u] = *reinterpret_cast<const u16x8*>(vb + r * ldc);
        }
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += qv[e] * bf2f(kf[u][e]);
            sc[u] = t;
        }
#pragma unroll
        for (int off = LPK / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) sc[u] += __shfl_xor(sc[u], off, 64);
        }
        float smax = -INFINITY;
#pragma unroll
        for (int u = U - 1; u >= 0; --u) { sc[u] = ok[u] ? sc[u] : -INFINITY; smax = fmaxf(sc[u], smax); }
        const float mn = fmaxf(m, smax);
        const float ms = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = exp2f(m - ms);
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = exp2f(sc[u] - ms);
        m = mn;
        float lt = l * alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) lt += p[u];
        l = lt;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = acc[e] * alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) t += p[u] * bf2f(vf[u][e]);
            acc[e] = t;
        }
    }
    if (sub == 0) { sm_m[wave][kg] = m; sm_l[wave][kg] = l; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_acc[wave][kg][sub * 8 + e] = acc[e];
    __syncthreads();
    // merge the NW*KPW partial states; thread d < HD owns output dim d
    if (threadIdx.x < HD) {
        const int d = threadIdx.x;
        float gm = -INFINITY;
        for (int w = 0; w < NW; ++w)
            for (int c = 0; c < KPW; ++c) gm = fmaxf(gm, sm_m[w][c]);
        float tl = 0.f, ta = 0.f;
        if (gm > -INFINITY) {
            for (int w = 0; w < NW; ++w)
                for (int c = 0; c < KPW; ++c) {
                    const float f = exp2f(sm_m[w][c] - gm);
                    tl += sm_l[w][c] * f;
                    ta += sm_acc[w][c][d] * f;
                }
        }
        o[(long)n * ldo + h * HD + d] = f2bf(tl > 0.f ? ta / tl : 0.f);
    }
}

extern "C" int aa_attn_decode(const void* q, long ldq, const void* Kc, const void* Vc, long ldc, int Tmax,
                              const int* start, const int* len, void* o, long ldo, int N, int H, int Hkv,
                              int hd, float scale, void* stream) {
    AA_REQUIRE(N > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Tmax > 0, "aa_attn_decode: bad shape N=%d H=%d Hkv=%d Tmax=%d", N, H, Hkv, Tmax);
    AA_REQUIRE(hd == 64 || hd == 128, "aa_attn_decode: head_dim %d not built (64 and 128 are)", hd);
    AA_REQUIRE(len != nullptr, "aa_attn_decode: len (keys per sequence) is required");
    AA_REQUIRE((ldq | ldc | ldo) % 8 == 0, "aa_attn_decode: leading dims must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    const bool few = (long)H * N < 512;      // fewer than two workgroups per CU: give each one 8 waves
    const bool handful = (long)H * N < 128 && (decode_r6() & 1);   // one or two sequences: four key steps in flight per wave (half the dependent trips)
#define AA_LAUNCH_ATTN_DECODE(HD_, NW_, U_)                                                                                   \
    hipLaunchKernelGGL((attn_decode_kernel<HD_, NW_, U_>), dim3(H, N), dim3(NW_ * 64), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)Kc, \
                       (const bf16_t*)Vc, ldc, Tmax, start, len, (bf16_t*)o, ldo, H, Hkv, scale)
    const bool four = handful || (few && (decode_r6() & 4));  // bit 2: four steps for every launch of fewer than 512 workgroups (a GRPO rollout's 10 sequences x 28 heads)
    if (hd == 128) { if (four) AA_LAUNCH_ATTN_DECODE(128, 8, 4); else if (few) AA_LAUNCH_ATTN_DECODE(128, 8, 2); else AA_LAUNCH_ATTN_DECODE(128, 4, 2); }
    else { if (four) AA_LAUNCH_ATTN_DECODE(64, 8, 4); else if (few) AA_LAUNCH_ATTN_DECODE(64, 8, 2); else AA_LAUNCH_ATTN_DECODE(64, 4, 2); }
#undef AA_LAUNCH_ATTN_DECODE
    AA_CHECK_LAUNCH("aa_attn_decode");
    return AA_OK;
}

// ------------------------------------------------------------------ token selection
// greedy: first index of the row maximum (torch.argmax tie rule)
// hf:generation/logits_process.py RepetitionPenaltyLogitsProcessor (applied to the fp32 scores before the warpers):
// tokens already in the row's sequence (`seen` bitmap, prompt incl. its pad ids + everything generated) get
// score < 0 ? score * penalty : score / penalty.
__device__ __forceinline__ float penalised(const bf16_t* __restrict__ x, const uint8_t* __restrict__ seen, int i, float pen) {
    float v = bf2f(x[i]);
    if (seen && seen[i]) v = v < 0.f ? v * pen : v / pen;
    return v;
}

// seen[row, ids[row, j]] = 1 for every j (ids outside [0, V) are ignored)
__global__ __launch_bounds__(256) void mark_seen_kernel(const int64_t* __restrict__ ids, long ld, int rows, int L,
                                                        uint8_t* __restrict__ seen, long ld_seen, int V) {
    const long total = (long)rows * L;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / L;
        const int64_t id = ids[r * ld + (t % L)];
        if (id >= 0 && id < V) seen[r * ld_seen + id] = 1;
    }
}
extern "C" int aa_mark_seen(const int64_t* ids, long ld, int rows, int L, uint8_t* seen, long ld_seen, int V,
                            void* stream) {
    AA_REQUIRE(rows >= 0 && L >= 0 && V > 0, "aa_mark_seen: bad shape rows=%d L=%d V=%d", rows, L, V);
    if (rows == 0 || L == 0) return AA_OK;
    const long total = (long)rows * L;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(mark_seen_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, ld, rows, L, seen, ld_seen, V);
    AA_CHECK_LAUNCH("aa_mark_seen");
    return AA_OK;
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const bf16_t* __restrict__ logits, long ld, int V,
                                                          const uint8_t* __restrict__ seen_all, long ld_seen, float pen,
                                                          int64_t* __restrict__ out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const bf16_t* x = logits + (long)blockIdx.x * ld;
    const uint8_t* seen = seen_all ? seen_all + (long)blockIdx.x * ld_seen : nullptr;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += 256) {
        const float v = penalised(x, seen, i, pen);
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = bv[threadIdx.x + s]; const int i = bi[threadIdx.x + s];
            if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && i < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}
extern "C" int aa_argmax_rows(const void* logits, long ld, int rows, int V, const uint8_t* seen, long ld_seen,
                              float repetition_penalty, int64_t* out, void* stream) {
    AA_REQUIRE(rows > 0 && V > 0, "aa_argmax_rows: bad shape rows=%d V=%d", rows, V);
    AA_REQUIRE(repetition_penalty > 0.f, "aa_argmax_rows: repetition_penalty must be > 0");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                       seen, ld_seen, repetition_penalty, out);
    AA_CHECK_LAUNCH("aa_argmax_rows");
    return AA_OK;
}

// temperature + top-k + nucleus (top-p) sampling, HF semantics and warper order (TemperatureLogitsWarper, TopKLogitsWarper,
// TopPLogitsWarper with min_tokens_to_keep = 1; hf:generation/logits_process.py -- the reference's GenerationConfig(temperature, top_p,
// repetition_penalty, do_sample=True), trainers/text_to_text/ppo.py:161-170, inherits HF's default top_k: 50 under transformers 4.x):
// top-k keeps every score >= the k-th largest (ties stay, as `scores < topk(scores, k)[..., -1]` removes only strictly smaller ones) --
// found EXACTLY by an 8-way search over the order-preserving integer keys of the scores (counts, not masses: <= 12 passes); then keep the
// smallest set of highest-probability tokens of the renormalised rest whose mass reaches top_p, renormalise, draw with the caller's
// uniform u[row].  The kept set is found by searching the probability threshold (no sort): 10 rounds
// of an 8-way split (7 candidate thresholds per pass over the row = the 2^-30 resolution of 30 bisections in a third of the
// passes); the draw walks the vocabulary in index order.
// One 512-thread workgroup per row; thread t owns the contiguous slice [t*per, (t+1)*per) and re-reads it from L2 with 16-B
// loads in each of the 13 passes (a register-resident copy spills at 1024 x 40 and at 512 x 80 values -- measured, not kept).
// floor: scores below it are removed (-inf), the top-k cut of TopKLogitsWarper; -inf = keep everything
__device__ __forceinline__ void load_scores8(const bf16_t* __restrict__ x, const uint8_t* __restrict__ seen, int i0, int e, float pen,
                                             float inv_temp, float (&v)[8], float floor = -INFINITY) {
    if (i0 + 8 <= e && ((reinterpret_cast<uintptr_t>(x + i0) & 15) == 0)) {
        const u16x8 t = *reinterpret_cast<const u16x8*>(x + i0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(t[j]);
        if (seen) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (seen[i0 + j]) v[j] = v[j] < 0.f ? v[j] * pen : v[j] / pen;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv_temp;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (i0 + j < e) ? penalised(x, seen, i0 + j, pen) * inv_temp : -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] >= floor ? v[j] : -INFINITY;
}

// order-preserving map float -> uint32 (larger float = larger key; -inf is the smallest key of any score)
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
}

// exp on the hardware's exp2 (v_exp_f32, ~1 ulp): the sampler evaluates it V times in each of its >= 3 passes on ONE compute unit -- with libm's expf (range reduction +
// polynomial, ~20 VALU instructions) that was ~60 of the kernel's 100 us at V = 152064 (round 5: coalescing the loads instead changed nothing, it is not the memory system).
// exp(-inf) = 0 and exp(0) = 1 exactly, as before.
__device__ __forceinline__ float sexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__global__ __launch_bounds__(512) void sample_top_p_kernel(const bf16_t* __restrict__ logits, long ld, int V,
                                                           float inv_temp, float top_p, int top_k,
                                                           const float* __restrict__ u,
                                                           const uint8_t* __restrict__ seen_all, long ld_seen, float pen,
                                                           int64_t* __restrict__ out) {
    constexpr int NT = 512, NW = NT / 64, KS = 7;
    __shared__ float red[NW];
    __shared__ float red7[NW][KS];
    __shared__ float wtot[NW];
    __shared__ int sel[2];
    const bf16_t* x = logits + (long)blockIdx.x * ld;
    const uint8_t* seen = seen_all ? seen_all + (long)blockIdx.x * ld_seen : nullptr;
    const int per = (((V + NT - 1) / NT) + 7) & ~7;
    const int b = min(V, (int)threadIdx.x * per), e = min(V, b + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- pass 1: maximum of the scores (penalty, temperature applied)
    float mx = -INFINITY;
#pragma unroll 2
    for (int i = b; i < e; i += 8) {
        float v[8];
        load_scores8(x, seen, i, e, pen, inv_temp, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[j]);
    }
    mx = block_max<NT>(mx, red);
    // ---- top-k: floor = the k-th largest score (exact).  Invariant: count(score >= lo) >= k > count(score >= hi), on integer keys.
    float floor = -INFINITY;
    if (top_k > 0 && top_k < V) {
        __shared__ int cnt7[NW][KS];
        unsigned long long lo = (unsigned long long)float_key(-INFINITY) + 1ull, hi = (unsigned long long)float_key(mx) + 1ull;
        while (hi - lo > 1ull) {
            const unsigned long long width = hi - lo;
            int cs[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) cs[k] = 0;
#pragma unroll 2
            for (int i = b; i < e; i += 8) {
                float v[8];
                load_scores8(x, seen, i, e, pen, inv_temp, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned long long kj = float_key(v[j]);
#pragma unroll
                    for (int k = 0; k < KS; ++k) cs[k] += (kj >= lo + width * (unsigned long long)(k + 1) / 8ull) ? 1 : 0;
                }
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cs[k] += __shfl_xor(cs[k], o, 64);
            }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < KS; ++k) cnt7[wave][k] = cs[k];
            }
            __syncthreads();
            unsigned long long nlo = lo, nhi = lo + width / 8ull;
            if (nhi <= lo) nhi = lo + 1ull;           // width < 8: the first candidate equals lo (whose count is >= k by the invariant)
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                int t = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += cnt7[w][k];
                const unsigned long long ck = lo + width * (unsigned long long)(k + 1) / 8ull;
                if (t >= top_k && ck > nlo) { nlo = ck; nhi = (k + 1 < KS) ? lo + width * (unsigned long long)(k + 2) / 8ull : hi; }
            }
            if (nhi <= nlo) nhi = nlo + 1ull;
            lo = nlo; hi = nhi;
        }
        floor = key_float((uint32_t)lo);
    }
    // ---- pass 2: partition function
    float z = 0.f;
#pragma unroll 2
    for (int i = b; i < e; i += 8) {
        float v[8];
        load_scores8(x, seen, i, e, pen, inv_temp, v, floor);
#pragma unroll
        for (int j = 0; j < 8; ++j) z += sexp(v[j] - mx);          // exp(-inf) = 0 for the slots beyond e
    }
    z = block_sum<NT>(z, red);
    const float invz = 1.f / z;
    // ---- threshold search: largest tau with mass(p >= tau) >= top_p   (p in (0, 1], p_max = 1/z)
    float lo = 0.f, hi = invz, kept = 1.f;     // mass(p >= 0) = 1 >= top_p
    if (top_p < 1.f) {
        for (int it = 0; it < 10; ++it) {
            const float step = (hi - lo) * 0.125f;
            float ms[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) ms[k] = 0.f;
#pragma unroll 2
            for (int i = b; i < e; i += 8) {
                float v[8];
                load_scores8(x, seen, i, e, pen, inv_temp, v, floor);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pj = sexp(v[j] - mx) * invz;
#pragma unroll
                    for (int k = 0; k < KS; ++k) ms[k] += (pj >= lo + step * (float)(k + 1)) ? pj : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) ms[k] = wave_sum(ms[k]);
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < KS; ++k) red7[wave][k] = ms[k];
            }
            __syncthreads();
            float nlo = lo, nhi = lo + step, nkept = kept;      // no candidate keeps enough mass -> the answer is below the first one
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += red7[w][k];
                if (t >= top_p) { nlo = lo + step * (float)(k + 1); nhi = (k + 1 < KS) ? lo + step * (float)(k + 2) : hi; nkept = t; }
            }
            lo = nlo; hi = nhi; kept = nkept;
        }
    }
    const float tau = lo;
    // ---- the draw: first index (vocabulary order) whose running kept mass exceeds u * kept
    float mine = 0.f;
#pragma unroll 2
    for (int i = b; i < e; i += 8) {
        float v[8];
        load_scores8(x, seen, i, e, pen, inv_temp, v, floor);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float pj = sexp(v[j] - mx) * invz; mine += (pj >= tau) ? pj : 0.f; }
    }
    if (top_p >= 1.f) kept = block_sum<NT>(mine, red);          // (the search did not run: kept = total mass as summed here)
    const float target = u[blockIdx.x] * kept;
    float incl = mine;                                           // inclusive scan over the threads: wave level, then wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (threadIdx.x == 0) { sel[0] = NT; sel[1] = -1; }
    __syncthreads();
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    incl += base;
    if (incl > target && mine > 0.f) atomicMin(&sel[0], (int)threadIdx.x);
    if (mine > 0.f) atomicMax(&sel[1], (int)threadIdx.x);
    __syncthreads();
    // u ~ 1 and rounding can leave the total a hair under the target: then the last kept token is drawn
    const int owner = sel[0] < NT ? sel[0] : sel[1];
    if ((int)threadIdx.x == owner) {
        float c = incl - mine;
        int pick = -1, last_kept = -1;
        for (int i = b; i < e && pick < 0; i += 8) {
            float v[8];
            load_scores8(x, seen, i, e, pen, inv_temp, v, floor);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pj = sexp(v[j] - mx) * invz;
                if (pick < 0 && i + j < e && pj >= tau) { last_kept = i + j; c += pj; if (c > target) pick = i + j; }
            }
        }
        out[blockIdx.x] = pick >= 0 ? pick : (last_kept >= 0 ? last_kept : b);
    }
    if (owner < 0 && threadIdx.x == 0) out[blockIdx.x] = 0;      // unreachable: p_max >= tau always keeps one token
}
extern "C" int aa_sample_top_k_top_p(const void* logits, long ld, int rows, int V, float temperature, int top_k, float top_p,
                                     const float* uniform, const uint8_t* seen, long ld_seen, float repetition_penalty,
                                     int64_t* out, void* stream) {
    AA_REQUIRE(rows > 0 && V > 0, "aa_sample_top_k_top_p: bad shape rows=%d V=%d", rows, V);
    AA_REQUIRE(repetition_penalty > 0.f, "aa_sample_top_k_top_p: repetition_penalty must be > 0");
    AA_REQUIRE(temperature > 0.f && top_p > 0.f && top_p <= 1.f, "aa_sample_top_k_top_p: temperature=%f / top_p=%f out of range", temperature, top_p);
    AA_REQUIRE(top_k >= 0, "aa_sample_top_k_top_p: top_k %d (0 = no cut)", top_k);
    hipLaunchKernelGGL(sample_top_p_kernel, dim3(rows), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                       1.f / temperature, top_p, top_k, uniform, seen, ld_seen, repetition_penalty, out);
    AA_CHECK_LAUNCH("aa_sample_top_k_top_p");
    return AA_OK;
}
extern "C" int aa_sample_top_p(const void* logits, long ld, int rows, int V, float temperature, float top_p,
                               const float* uniform, const uint8_t* seen, long ld_seen, float repetition_penalty,
                               int64_t* out, void* stream) {
    return aa_sample_top_k_top_p(logits, ld, rows, V, temperature, 0, top_p, uniform, seen, ld_seen, repetition_penalty, out, stream);
}

// align_anything/trainers/text_image_to_text/ppo.py:56-86 move_padding_left: every row of the generated sequences is rotated
// right by (L - #non-pad - #leading-pad) so that the padding appended after EOS joins the left padding.  Same integer
// arithmetic as the reference (a circular shift; pad ids inside the text are NOT compacted) -- bit-exact.
__global__ __launch_bounds__(256) void move_padding_left_kernel(const int64_t* __restrict__ in, long ldi,
                                                                int64_t* __restrict__ out, long ldo, int L, int64_t pad) {
    __shared__ int red[2][4];
    const int64_t* x = in + (long)blockIdx.x * ldi;
    int nonpad = 0, first = L;                       // first non-pad index = length of the leading pad run
    for (int t = threadIdx.x; t < L; t += 256) {
        if (x[t] != pad) { ++nonpad; first = min(first, t); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nonpad += __shfl_xor(nonpad, o, 64); first = min(first, __shfl_xor(first, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nonpad; red[1][threadIdx.x >> 6] = first; }
    __syncthreads();
    nonpad = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    first = min(min(red[1][0], red[1][1]), min(red[1][2], red[1][3]));
    const int shift = L - nonpad - first;
    int64_t* y = out + (long)blockIdx.x * ldo;
    for (int t = threadIdx.x; t < L; t += 256) y[t] = x[((t - shift) % L + L) % L];
}
extern "C" int aa_move_padding_left(const int64_t* in, long ldi, int64_t* out, long ldo, int rows, int L, int64_t pad,
                                    void* stream) {
    AA_REQUIRE(rows >= 0 && L > 0 && in != out, "aa_move_padding_left: bad arguments (rows=%d L=%d, in-place not supported)", rows, L);
    if (rows == 0) return AA_OK;
    hipLaunchKernelGGL(move_padding_left_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo, L, pad);
    AA_CHECK_LAUNCH("aa_move_padding_left");
    return AA_OK;
}

// ------------------------------------------------------------------ the decode loop's bookkeeping in two launches
// `generate` keeps everything a step needs in device buffers (one fixed launch sequence per position, hipGraph-capturable).  Written with torch
// ops that bookkeeping was ~11 one-element kernels per position, each at the ~4.8 us launch floor (profiles/r04_decode_trace_summary_before_norm_fold.txt).
//   aa_decode_record : after the selection kernel.  nact += any(unfinished) (the columns hf's stopping criteria keep); tok[n] = unfinished[n] ?
//                      selected[n] : pad; out[n, tslot[n]] = tok[n]; unfinished[n] &= tok[n] != eos (eos < 0: no EOS) -- hf GenerationMixin._sample's
//                      `next_tokens * unfinished + pad * (1 - unfinished)` and EosTokenCriteria, per row.
//   aa_decode_tick   : after the decode pass.  tslot / pos / length += 1 per row, step += 1.
__global__ __launch_bounds__(256) void decode_record_kernel(const int64_t* __restrict__ selected, uint8_t* __restrict__ unfinished, int64_t* __restrict__ out,
                                                            long ldo, const int64_t* __restrict__ tslot, int64_t* __restrict__ tok, int64_t* __restrict__ nact,
                                                            int N, int64_t pad, int64_t eos) {
    __shared__ int any_s;
    if (threadIdx.x == 0) any_s = 0;
    __syncthreads();
    int mine = 0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const bool u = unfinished[n] != 0;
        mine |= u ? 1 : 0;
        const int64_t t = u ? selected[n] : pad;
        tok[n] = t;
        out[(long)n * ldo + tslot[n]] = t;
        if (eos >= 0 && u && t == eos) unfinished[n] = 0;
    }
    if (mine) atomicOr(&any_s, 1);
    __syncthreads();
    if (threadIdx.x == 0 && any_s) *nact += 1;
}
extern "C" int aa_decode_record(const int64_t* selected, uint8_t* unfinished, int64_t* out, long ldo, const int64_t* tslot, int64_t* tok, int64_t* nact,
                                int N, int64_t pad, int64_t eos, void* stream) {
    AA_REQUIRE(N >= 0 && ldo > 0, "aa_decode_record: bad arguments (N=%d ldo=%ld)", N, ldo);
    if (N == 0) return AA_OK;
    hipLaunchKernelGGL(decode_record_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, selected, unfinished, out, ldo, tslot, tok, nact, N, pad, eos);
    AA_CHECK_LAUNCH("aa_decode_record");
    return AA_OK;
}
__global__ __launch_bounds__(256) void decode_tick_kernel(int64_t* __restrict__ tslot, int* __restrict__ pos, int* __restrict__ length, int64_t* __restrict__ step, int N) {
    for (int n = threadIdx.x; n < N; n += 256) { tslot[n] += 1; pos[n] += 1; length[n] += 1; }
    if (threadIdx.x == 0) *step += 1;
}
extern "C" int aa_decode_tick(int64_t* tslot, int* pos, int* length, int64_t* step, int N, void* stream) {
    AA_REQUIRE(N >= 0, "aa_decode_tick: N=%d", N);
    hipLaunchKernelGGL(decode_tick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tslot, pos, length, step, N);
    AA_CHECK_LAUNCH("aa_decode_tick");
    return AA_OK;
}
