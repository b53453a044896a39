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
template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                  const bf16_t* __restrict__ W, long ldw,
                                                                  bf16_t* __restrict__ out, long ldo,
                                                                  const bf16_t* __restrict__ bias,
                                                                  const bf16_t* __restrict__ residual, long ldr,
                                                                  int M, int N, int K,
                                                                  const int* __restrict__ row_expert, long strideE, int x_div) {
    __shared__ float red[NWAVE][16][17];
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
    const bf16_t* wp = W + (long)nrow * ldw + g * 8;
    const bf16_t* xp = x + (long)mrow * ldx + g * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // wave w owns k in [w*kq, (w+1)*kq), kq = K/NWAVE rounded up to 32; 8 MFMA k-steps (256 k) per trip keep
    // 16 x 16-B loads per lane in flight (the weight stream is read exactly once: no LDS round trip)
    const int kq = ((K / NWAVE + 31) / 32) * 32;
    const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
    int k = k_lo;
    for (; k + 256 <= k_hi; k += 256) {
        bf16x8 wf[8], xf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            wf[s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + s * 32));
            xf[s] = *reinterpret_cast<const bf16x8*>(xp + k + s * 32);
        }
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], xf[s + 1], acc1, 0, 0, 0);
        }
    }
    for (; k < k_hi; k += 32) {
        const bf16x8 wf = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k));
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + k);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc0, 0, 0, 0);
    }
    // D[i = n (4g + r)][j = m (l15)]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][l15] = acc0[r] + acc1[r];
    __syncthreads();
    // thread t < 256 -> (m = t / 16, n = t % 16)
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
    const int n = n0 + nn;
    if (threadIdx.x < 256 && m < M && n < N) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) v += red[w][nn][m];
        if (bias) v += bf2f(bias[n]);
        if (residual) v = rbf(v) + bf2f(residual[(long)m * ldr + n]);
        out[(long)m * ldo + n] = f2bf(v);
    }
}

extern "C" int aa_gemm_skinny_bf16(const void* x, const void* W, void* out, int M, int N, int K, long ldx,
                                   long ldw, long ldo, const void* bias, const void* residual, long ldr,
                                   void* stream) {
    AA_REQUIRE(M >= 1 && M <= 16, "aa_gemm_skinny_bf16: M=%d must be in [1, 16] (use aa_gemm_bf16 beyond)", M);
    AA_REQUIRE(N > 0 && K > 0 && K % 32 == 0, "aa_gemm_skinny_bf16: K=%d must be a multiple of 32", K);
    AA_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0, "aa_gemm_skinny_bf16: ldx/ldw must be multiples of 8");
    // few column strips (N/16 < ~3 per CU): split K over 8 waves so enough loads are in flight per CU
    const bool wide = (N / 16) >= 768 || K < 2048;
    if (wide)
        hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(aa_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                           (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3(aa_cdiv(N, 16)), dim3(512), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, (const bf16_t*)bias,
                           (const bf16_t*)residual, ldr, M, N, K, nullptr, 0, 1);
    AA_CHECK_LAUNCH("aa_gemm_skinny_bf16");
    return AA_OK;
}

// Mixture-of-experts decode: out[r, :] = x[r / x_div, :] W3[row_expert[r]]^T for R routed rows (a handful of tokens x top-k):
// every routed row streams its own expert matrix once (the floor for a row-private weight), no padding to the 128-row
// tiles of the grouped training GEMM.  hf:models/qwen3_moe/modeling_qwen3_moe.py:210-283 at one token per sequence.
extern "C" int aa_moe_gemv_bf16(const void* x, const void* W3, void* out, int R, int N, int K, long ldx, long ldw, long ldo,
                                const int* row_expert, long strideE, int x_div, void* stream) {
    AA_REQUIRE(R >= 1 && R <= 65535 && N > 0 && K > 0 && K % 32 == 0, "aa_moe_gemv_bf16: R=%d N=%d K=%d (K must be a multiple of 32)", R, N, K);
    AA_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && strideE % 8 == 0 && x_div >= 1 && row_expert != nullptr, "aa_moe_gemv_bf16: ldx/ldw/strideE must be multiples of 8");
    hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(aa_cdiv(N, 16), R), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (const bf16_t*)W3, ldw, (bf16_t*)out, ldo, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, 0, 1, N, K, row_expert, strideE, x_div);
    AA_CHECK_LAUNCH("aa_moe_gemv_bf16");
    return AA_OK;
}

// ------------------------------------------------------------------ decode attention
// q [N, H*HD] (one new token per sequence), caches Kc/Vc token-major [N, Tmax, Hkv*HD] (row stride ldc),
// valid keys of sequence n: [start[n], len[n]).  One workgroup per (head, sequence); HD/8 lanes per key,
// 64/(HD/8) keys per wave step, 4 waves stride the keys; partial (m, l, acc) merged through LDS.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, long ldq,
                                                          const bf16_t* __restrict__ Kc,
                                                          const bf16_t* __restrict__ Vc, long ldc, int Tmax,
                                                          const int* __restrict__ start,
                                                          const int* __restrict__ len, bf16_t* __restrict__ o,
                                                          long ldo, int H, int Hkv, float scale) {
    constexpr int LPK = HD / 8;        // lanes per key
    constexpr int KPW = 64 / LPK;      // keys per wave step
    __shared__ float sm_m[4][KPW], sm_l[4][KPW];
    __shared__ float sm_acc[4][KPW][HD];
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
    for (int j0 = s0 + wave * KPW; j0 < s1; j0 += 4 * KPW) {
        const int j = j0 + kg;
        const bool ok = j < s1;
        const int jr = ok ? j : s1 - 1;
        const u16x8 kk = *reinterpret_cast<const u16x8*>(kb + (long)jr * ldc);
        const u16x8 vv = *reinterpret_cast<const u16x8*>(vb + (long)jr * ldc);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qv[e] * bf2f(kk[e]);
#pragma unroll
        for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        s = ok ? s : -INFINITY;
        const float mn = fmaxf(m, s);
        const float ms = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = exp2f(m - ms), p = exp2f(s - ms);
        m = mn;
        l = l * alpha + p;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + p * bf2f(vv[e]);
    }
    if (sub == 0) { sm_m[wave][kg] = m; sm_l[wave][kg] = l; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_acc[wave][kg][sub * 8 + e] = acc[e];
    __syncthreads();
    // merge the 4*KPW partial states; thread d < HD owns output dim d
    if (threadIdx.x < HD) {
        const int d = threadIdx.x;
        float gm = -INFINITY;
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < KPW; ++c) gm = fmaxf(gm, sm_m[w][c]);
        float tl = 0.f, ta = 0.f;
        if (gm > -INFINITY) {
            for (int w = 0; w < 4; ++w)
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
    if (hd == 128)
        hipLaunchKernelGGL(attn_decode_kernel<128>, dim3(H, N), dim3(256), 0, st, (const bf16_t*)q, ldq,
                           (const bf16_t*)Kc, (const bf16_t*)Vc, ldc, Tmax, start, len, (bf16_t*)o, ldo, H, Hkv, scale);
    else
        hipLaunchKernelGGL(attn_decode_kernel<64>, dim3(H, N), dim3(256), 0, st, (const bf16_t*)q, ldq,
                           (const bf16_t*)Kc, (const bf16_t*)Vc, ldc, Tmax, start, len, (bf16_t*)o, ldo, H, Hkv, scale);
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

// temperature + nucleus (top-p) sampling, HF semantics (TemperatureLogitsWarper, TopPLogitsWarper with
// min_tokens_to_keep = 1): keep the smallest set of highest-probability tokens whose mass reaches top_p, renormalise,
// draw with the caller's uniform u[row].  The kept set is found by bisection on the probability threshold (no sort);
// the draw walks the vocabulary in index order.
__global__ __launch_bounds__(256) void sample_top_p_kernel(const bf16_t* __restrict__ logits, long ld, int V,
                                                           float inv_temp, float top_p,
                                                           const float* __restrict__ u,
                                                           const uint8_t* __restrict__ seen_all, long ld_seen, float pen,
                                                           int64_t* __restrict__ out) {
    __shared__ float red[8];
    __shared__ float part[256];
    const bf16_t* x = logits + (long)blockIdx.x * ld;
    const uint8_t* seen = seen_all ? seen_all + (long)blockIdx.x * ld_seen : nullptr;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, penalised(x, seen, i, pen) * inv_temp);
    mx = block_max<256>(mx, red);
    float z = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) z += expf(penalised(x, seen, i, pen) * inv_temp - mx);
    z = block_sum<256>(z, red);
    const float invz = 1.f / z;
    // bisection: largest tau with mass(p >= tau) >= top_p   (p in (0, 1], p_max = 1/z * 1)
    float lo = 0.f, hi = invz;  // mass(p >= lo) = 1 >= top_p ; hi = p_max
    if (top_p < 1.f) {
        for (int it = 0; it < 30; ++it) {
            const float tau = 0.5f * (lo + hi);
            float ms = 0.f;
            for (int i = threadIdx.x; i < V; i += 256) {
                const float p = expf(penalised(x, seen, i, pen) * inv_temp - mx) * invz;
                ms += (p >= tau) ? p : 0.f;
            }
            ms = block_sum<256>(ms, red);
            if (ms >= top_p) lo = tau; else hi = tau;
        }
    }
    const float tau = lo;
    float kept = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) {
        const float p = expf(penalised(x, seen, i, pen) * inv_temp - mx) * invz;
        kept += (p >= tau) ? p : 0.f;
    }
    kept = block_sum<256>(kept, red);
    const float target = u[blockIdx.x] * kept;
    // each thread owns a contiguous slice so the walk is in index order
    const int per = (V + 255) / 256;
    const int b = threadIdx.x * per, e = min(V, b + per);
    float mine = 0.f;
    for (int i = b; i < e; ++i) {
        const float p = expf(penalised(x, seen, i, pen) * inv_temp - mx) * invz;
        mine += (p >= tau) ? p : 0.f;
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f; int t = 0;
        for (; t < 255; ++t) { if (c + part[t] > target) break; c += part[t]; }
        part[0] = c; red[0] = __int_as_float(t);
    }
    __syncthreads();
    const int owner = __float_as_int(red[0]);
    if (threadIdx.x == owner) {
        float c = part[0];
        int pick = -1, last_kept = -1;
        for (int i = b; i < e; ++i) {
            const float p = expf(penalised(x, seen, i, pen) * inv_temp - mx) * invz;
            if (p >= tau) { last_kept = i; c += p; if (c > target) { pick = i; break; } }
        }
        if (pick < 0) pick = last_kept >= 0 ? last_kept : (e > b ? b : V - 1);
        out[blockIdx.x] = pick;
    }
}
extern "C" int aa_sample_top_p(const void* logits, long ld, int rows, int V, float temperature, float top_p,
                               const float* uniform, const uint8_t* seen, long ld_seen, float repetition_penalty,
                               int64_t* out, void* stream) {
    AA_REQUIRE(rows > 0 && V > 0, "aa_sample_top_p: bad shape rows=%d V=%d", rows, V);
    AA_REQUIRE(repetition_penalty > 0.f, "aa_sample_top_p: repetition_penalty must be > 0");
    AA_REQUIRE(temperature > 0.f && top_p > 0.f && top_p <= 1.f, "aa_sample_top_p: temperature=%f / top_p=%f out of range", temperature, top_p);
    hipLaunchKernelGGL(sample_top_p_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                       1.f / temperature, top_p, uniform, seen, ld_seen, repetition_penalty, out);
    AA_CHECK_LAUNCH("aa_sample_top_p");
    return AA_OK;
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
