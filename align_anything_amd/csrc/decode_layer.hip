// One decoder layer of a decode position in ONE launch (hf LlamaDecoderLayer at one new token per sequence,
// hf:models/llama/modeling_llama.py:284-325; called per generated position from generation.py for the rollout of
// align_anything/trainers/text_to_text/ppo.py:209-222).
//
// STATUS: built and cross-compiled, NOT YET RUN ON HARDWARE (written after round 4's GPU budget was spent).  Off by default
// (AA_DECODE_PERSISTENT=1 turns it on); tests/test_zzz_unvalidated_gpu.py holds its parity tests.
//
// Why: a kernel trace of the decode window (profiles/r04_decode_trace_summary_before_norm_fold.txt) shows that no launch of a decode position
// costs less than ~4.8 us whatever its size, and a layer is five launches (q/k/v + rope + cache, attention, o-projection, gate/up + SwiGLU,
// down-projection) of which three stream 25-34 MB (4-5 us of HBM time).  Here the five steps are phases of one persistent kernel -- one
// workgroup of 16 waves per CU, all co-resident -- separated by four device-side grid barriers (~2-3 us each instead of a launch boundary).
// The strips, their K split over the 16 waves, the MFMA order and the epilogues are those of csrc/decode.hip's strip kernel (PRO 3 / 4 on
// strip-major copies, EPI 0 / 1 / 2); that file is left untouched so that its validated kernels stay byte-identical.
//
// A barrier that does not complete (a workgroup that was not co-resident: G > what the device holds) can never hang the GPU: the spin is
// bounded by the shader clock, sets *status = 1 and lets the kernel run to its end with garbage results; the host checks `status` after the
// first position of a rollout and falls back to the per-step launches.
#include "aa_common.h"

#define LOG2E_DL 1.4426950408889634f

namespace {

struct LayerEpi {
    const int* pos; const bf16_t* cos_t; const bf16_t* sin_t; bf16_t* cache; long ldc; int Tmax; const int64_t* slot; int H, Hkv;
};

struct DecodeLayerParams {
    const bf16_t* x_in;      // [M, h] residual stream entering the layer (un-normalised)
    bf16_t* x_mid;           // [M, h] after the attention block
    bf16_t* x_out;           // [M, h] after the MLP block
    bf16_t* q;               // [M, H * 128] rotated queries
    bf16_t* attn;            // [M, H * 128] attention output
    bf16_t* act;             // [M, F] silu(gate) * up
    const bf16_t* Wqkv;      // strip-major, mode 2 (rotation pairs), columns scaled by the input norm weight
    const bf16_t* Wo;        // strip-major, mode 0
    const bf16_t* Wgu;       // strip-major, mode 1 ([gate; up] pairs), columns scaled by the post-attention norm weight
    const bf16_t* Wdown;     // strip-major, mode 0
    const bf16_t* bqkv;      // fused q | k | v bias or null
    int M, h, H, Hkv, F;
    float eps, scale;
    LayerEpi epi;
    const int* start;        // first valid key per sequence or null
    const int* len;          // keys per sequence (the new token included)
    unsigned int* bar;       // [2]: arrivals, generation (sense-reversing grid barrier; zero-initialised once)
    int* status;             // set to 1 by a barrier that timed out
};

// ---- fragments (csrc/decode.hip skinny_x / skinny_trip for the strip-major layouts)
template <int NORM>
__device__ __forceinline__ bf16x8 dl_x(const bf16_t* __restrict__ xp, int k, float& ss) {
    const u16x8 v = *reinterpret_cast<const u16x8*>(xp + k);
    if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            ss += f * f;
        }
    }
    return __builtin_bit_cast(bf16x8, v);
}

template <int NORM, int S>
__device__ __forceinline__ void dl_trip(const bf16_t* __restrict__ wp, const bf16_t* __restrict__ xp, int k, float& ss, f32x4& acc0, f32x4& acc1) {
    bf16x8 wf[S], xf[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        wf[s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)(k + s * 32) * 16));
        xf[s] = dl_x<NORM>(xp, k + s * 32, ss);
    }
#pragma unroll
    for (int s = 0; s < S; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], xf[s + 1], acc1, 0, 0, 0);
    }
}

constexpr int DL_NWAVE = 16;

// One 16-column strip of out = (rstd *) x W^T by the whole workgroup (16 waves split K).  NORM: W's columns carry the norm weight, the kernel
// accumulates sum(x^2) and scales by rstd.  EPI 0: (+ bias, + residual) -> out; 1: SwiGLU of the strip's (gate, up) pairs -> out[m, 8 strip + c];
// 2: rotary embedding + KV-cache write (head_dim 128).  The caller separates two strips by a __syncthreads() (red / ssred are reused).
template <int NORM, int EPI>
__device__ __forceinline__ void dl_strip(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ W, bf16_t* __restrict__ out, long ldo,
                                         const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, long ldr, int M, int N, int K,
                                         float eps, const LayerEpi& epi, const int strip, float (*red)[16][17], float (*ssred)[16]) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int n0 = strip * 16;
    const int mrow = min(l15, M - 1);
    const bf16_t* wp = W + (long)strip * 16 * K + lane * 8;
    const bf16_t* xp = x + (long)mrow * ldx + g * 8;
    float ss = 0.f;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int kq = ((K / DL_NWAVE + 31) / 32) * 32;
    const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
    int k = k_lo;
    for (; k + 256 <= k_hi; k += 256) dl_trip<NORM, 8>(wp, xp, k, ss, acc0, acc1);
    for (; k + 128 <= k_hi; k += 128) dl_trip<NORM, 4>(wp, xp, k, ss, acc0, acc1);
    for (; k + 64 <= k_hi; k += 64) dl_trip<NORM, 2>(wp, xp, k, ss, acc0, acc1);
    if (k < k_hi) {
        const bf16x8 wf = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (long)k * 16));
        const bf16x8 xf = dl_x<NORM>(xp, k, ss);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc0, 0, 0, 0);
    }
    // D[i = n (4g + r)][j = m (l15)]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][l15] = acc0[r] + acc1[r];
    if constexpr (NORM) {
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) ssred[wave][l15] = ss;
    }
    __syncthreads();
    if constexpr (EPI != 0) {
        const int m = threadIdx.x >> 3, c = threadIdx.x & 7;
        if (threadIdx.x < 128 && m < M) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < DL_NWAVE; ++w) { v1 += red[w][c][m]; v2 += red[w][c + 8][m]; }
            if constexpr (NORM) {
                float q = 0.f;
#pragma unroll
                for (int w = 0; w < DL_NWAVE; ++w) q += ssred[w][m];
                const float rs = rsqrtf(q / (float)K + eps);
                v1 *= rs; v2 *= rs;
            }
            if constexpr (EPI == 1) {
                const float gf = rbf(v1), uf = rbf(v2);
                out[(long)m * ldo + strip * 8 + c] = f2bf(rbf(gf * aa_sigmoid<false>(gf)) * uf);
            } else {
                const int head = strip >> 3, d = (strip & 7) * 8 + c;
                const int col = head * 128 + d;
                if (bias) { v1 += bf2f(bias[col]); v2 += bf2f(bias[col + 64]); }
                const float a = rbf(v1), b = rbf(v2);
                bf16_t* crow = epi.cache + ((long)m * epi.Tmax + epi.slot[m]) * epi.ldc;
                if (head >= epi.H + epi.Hkv) {
                    bf16_t* dst = crow + (long)epi.Hkv * 128 + (long)(head - epi.H - epi.Hkv) * 128 + d;
                    dst[0] = f2bf(a);
                    dst[64] = f2bf(b);
                } else {
                    const long tb = (long)epi.pos[m] * 64 + d;
                    const float cc = bf2f(epi.cos_t[tb]), sn = bf2f(epi.sin_t[tb]);
                    const bf16_t o1 = f2bf(rbf(a * cc) + rbf(-b * sn)), o2 = f2bf(rbf(b * cc) + rbf(a * sn));
                    bf16_t* dst = head < epi.H ? out + (long)m * ldo + col : crow + (long)(head - epi.H) * 128 + d;
                    dst[0] = o1;
                    dst[64] = o2;
                }
            }
        }
        return;
    }
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
    const int n = n0 + nn;
    if (threadIdx.x < 256 && m < M && n < N) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < DL_NWAVE; ++w) v += red[w][nn][m];
        if constexpr (NORM) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < DL_NWAVE; ++w) q += ssred[w][m];
            v *= rsqrtf(q / (float)K + eps);
        }
        if (bias) v += bf2f(bias[n]);
        if (residual) v = rbf(v) + bf2f(residual[(long)m * ldr + n]);
        out[(long)m * ldo + n] = f2bf(v);
    }
}

// One (head h, sequence n) of the decode attention by the whole workgroup (csrc/decode.hip attn_decode_kernel<128, NW> with NW = 16).
__device__ __forceinline__ void dl_attention(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vc, long ldc,
                                             int Tmax, const int* __restrict__ start, const int* __restrict__ len, bf16_t* __restrict__ o, long ldo,
                                             int H, int Hkv, float scale, const int h, const int n, float* smem) {
    constexpr int HD = 128, NW = DL_NWAVE, LPK = HD / 8, KPW = 64 / LPK, STRIDE = NW * KPW;
    float (*sm_m)[KPW] = reinterpret_cast<float (*)[KPW]>(smem);
    float (*sm_l)[KPW] = reinterpret_cast<float (*)[KPW]>(smem + NW * KPW);
    float (*sm_acc)[KPW][HD] = reinterpret_cast<float (*)[KPW][HD]>(smem + 2 * NW * KPW);
    const int hk = h / (H / Hkv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPK, kg = lane / LPK;
    const int s0 = start ? start[n] : 0, s1 = len[n];
    float qv[8];
    {
        const u16x8 t = *reinterpret_cast<const u16x8*>(q + (long)n * ldq + h * HD + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = bf2f(t[j]) * scale * LOG2E_DL;
    }
    float m = -INFINITY, l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16_t* kb = Kc + (long)n * Tmax * ldc + hk * HD + sub * 8;
    const bf16_t* vb = Vc + (long)n * Tmax * ldc + hk * HD + sub * 8;
    for (int j0 = s0 + wave * KPW; j0 < s1; j0 += 2 * STRIDE) {
        const int ja = j0 + kg, jb = ja + STRIDE;
        const bool oka = ja < s1, okb = jb < s1;
        const long ra = oka ? ja : s1 - 1, rb = okb ? jb : s1 - 1;
        const u16x8 ka = *reinterpret_cast<const u16x8*>(kb + ra * ldc);
        const u16x8 kb2 = *reinterpret_cast<const u16x8*>(kb + rb * ldc);
        const u16x8 va = *reinterpret_cast<const u16x8*>(vb + ra * ldc);
        const u16x8 vb2 = *reinterpret_cast<const u16x8*>(vb + rb * ldc);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa += qv[e] * bf2f(ka[e]); sb += qv[e] * bf2f(kb2[e]); }
#pragma unroll
        for (int off = LPK / 2; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sb += __shfl_xor(sb, off, 64); }
        sa = oka ? sa : -INFINITY;
        sb = okb ? sb : -INFINITY;
        const float mn = fmaxf(m, fmaxf(sa, sb));
        const float ms = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = exp2f(m - ms), pa = exp2f(sa - ms), pb = exp2f(sb - ms);
        m = mn;
        l = l * alpha + pa + pb;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + pa * bf2f(va[e]) + pb * bf2f(vb2[e]);
    }
    if (sub == 0) { sm_m[wave][kg] = m; sm_l[wave][kg] = l; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_acc[wave][kg][sub * 8 + e] = acc[e];
    __syncthreads();
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

// Sense-reversing grid barrier over all workgroups of the launch.  Every thread publishes its stores (agent-scope fence: on a multi-XCD part
// that writes the XCD's L2 back), thread 0 arrives; the last arrival resets the count and flips the generation; everyone else spins on the
// generation with a bounded wait.  After the barrier every thread fences again (acquire: stale L1 / L2 lines of the other XCDs' data are dropped).
__device__ __forceinline__ void dl_grid_barrier(unsigned int* bar, int* status) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int gen = __hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int arrived = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {      // the acquire is the fence after the barrier
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000LL) {          // 0.2 s of the 100 MHz wall clock: a workgroup never became resident
                    atomicExch(status, 1);
                    break;
                }
            }
        }
    }
    __syncthreads();
    __threadfence();
}

__global__ __launch_bounds__(DL_NWAVE * 64) void decode_layer_kernel(const DecodeLayerParams p) {
    // reduction scratch of the strips (16 x 16 x 17 + 16 x 16 floats) and of the attention merge (2 x 64 + 16 x 4 x 128 floats) share the space
    __shared__ float smem[2 * DL_NWAVE * 4 + DL_NWAVE * 4 * 128];
    float (*red)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);
    float (*ssred)[16] = reinterpret_cast<float (*)[16]>(smem + DL_NWAVE * 16 * 17);
    const int G = gridDim.x;
    const int qw = p.H * 128, kw = p.Hkv * 128;
    // ---- A: rstd(x) * x (Wqkv diag(w1))^T (+ bias), rotary embedding, q out, k / v into the cache slot of the position
    for (int s = blockIdx.x; s < (p.H + 2 * p.Hkv) * 8; s += G) {
        dl_strip<1, 2>(p.x_in, p.h, p.Wqkv, p.q, qw, p.bqkv, nullptr, 0, p.M, (p.H + 2 * p.Hkv) * 128, p.h, p.eps, p.epi, s, red, ssred);
        __syncthreads();
    }
    dl_grid_barrier(p.bar, p.status);
    // ---- B: one query per sequence against the cache (the new token included)
    for (int it = blockIdx.x; it < p.H * p.M; it += G) {
        dl_attention(p.q, qw, p.epi.cache, p.epi.cache + kw, p.epi.ldc, p.epi.Tmax, p.start, p.len, p.attn, qw, p.H, p.Hkv, p.scale, it % p.H, it / p.H, smem);
        __syncthreads();
    }
    dl_grid_barrier(p.bar, p.status);
    // ---- C: x_mid = x + attn Wo^T
    for (int s = blockIdx.x; s < (p.h + 15) / 16; s += G) {
        dl_strip<0, 0>(p.attn, qw, p.Wo, p.x_mid, p.h, nullptr, p.x_in, p.h, p.M, p.h, qw, 0.f, p.epi, s, red, ssred);
        __syncthreads();
    }
    dl_grid_barrier(p.bar, p.status);
    // ---- D: act = silu(gate) * up of rstd(x_mid) * x_mid (Wgu diag(w2))^T
    for (int s = blockIdx.x; s < p.F / 8; s += G) {
        dl_strip<1, 1>(p.x_mid, p.h, p.Wgu, p.act, p.F, nullptr, nullptr, 0, p.M, 2 * p.F, p.h, p.eps, p.epi, s, red, ssred);
        __syncthreads();
    }
    dl_grid_barrier(p.bar, p.status);
    // ---- E: x_out = x_mid + act Wdown^T
    for (int s = blockIdx.x; s < (p.h + 15) / 16; s += G) {
        dl_strip<0, 0>(p.act, p.F, p.Wdown, p.x_out, p.h, nullptr, p.x_mid, p.h, p.M, p.h, p.F, 0.f, p.epi, s, red, ssred);
        __syncthreads();
    }
}

// All layers of a decode position in one launch: the per-layer parameter blocks sit in a device array (aa_decode_layers_pack), a fifth barrier
// separates layer l's down-projection from layer l + 1's q/k/v projection.
__global__ __launch_bounds__(DL_NWAVE * 64) void decode_layers_kernel(const DecodeLayerParams* __restrict__ layers, const int L) {
    __shared__ float smem[2 * DL_NWAVE * 4 + DL_NWAVE * 4 * 128];
    float (*red)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);
    float (*ssred)[16] = reinterpret_cast<float (*)[16]>(smem + DL_NWAVE * 16 * 17);
    const int G = gridDim.x;
    for (int li = 0; li < L; ++li) {
        const DecodeLayerParams& p = layers[li];      // fields are read where they are used (uniform scalar loads), not held across the phases
        const int qw = p.H * 128, kw = p.Hkv * 128;
        for (int s = blockIdx.x; s < (p.H + 2 * p.Hkv) * 8; s += G) {
            dl_strip<1, 2>(p.x_in, p.h, p.Wqkv, p.q, qw, p.bqkv, nullptr, 0, p.M, (p.H + 2 * p.Hkv) * 128, p.h, p.eps, p.epi, s, red, ssred);
            __syncthreads();
        }
        dl_grid_barrier(p.bar, p.status);
        for (int it = blockIdx.x; it < p.H * p.M; it += G) {
            dl_attention(p.q, qw, p.epi.cache, p.epi.cache + kw, p.epi.ldc, p.epi.Tmax, p.start, p.len, p.attn, qw, p.H, p.Hkv, p.scale, it % p.H, it / p.H, smem);
            __syncthreads();
        }
        dl_grid_barrier(p.bar, p.status);
        for (int s = blockIdx.x; s < (p.h + 15) / 16; s += G) {
            dl_strip<0, 0>(p.attn, qw, p.Wo, p.x_mid, p.h, nullptr, p.x_in, p.h, p.M, p.h, qw, 0.f, p.epi, s, red, ssred);
            __syncthreads();
        }
        dl_grid_barrier(p.bar, p.status);
        for (int s = blockIdx.x; s < p.F / 8; s += G) {
            dl_strip<1, 1>(p.x_mid, p.h, p.Wgu, p.act, p.F, nullptr, nullptr, 0, p.M, 2 * p.F, p.h, p.eps, p.epi, s, red, ssred);
            __syncthreads();
        }
        dl_grid_barrier(p.bar, p.status);
        for (int s = blockIdx.x; s < (p.h + 15) / 16; s += G) {
            dl_strip<0, 0>(p.act, p.F, p.Wdown, p.x_out, p.h, nullptr, p.x_mid, p.h, p.M, p.h, p.F, 0.f, p.epi, s, red, ssred);
            __syncthreads();
        }
        if (li + 1 < L) dl_grid_barrier(p.bar, p.status);
    }
}

static int dl_fill(DecodeLayerParams& p, const void* x_in, void* x_mid, void* x_out, void* q, void* attn, void* act, const void* Wqkv, const void* Wo,
                   const void* Wgu, const void* Wdown, const void* bqkv, int M, int h, int H, int Hkv, int F, float eps, float scale, const int* pos,
                   const void* cos_t, const void* sin_t, void* cache, long ldc, int Tmax, const int64_t* slot, const int* start, const int* len, void* bar,
                   int* status, const char* who) {
    AA_REQUIRE(M >= 1 && M <= 16, "%s: M=%d must be in [1, 16]", who, M);
    AA_REQUIRE(h > 0 && h % 32 == 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && F > 0 && F % 32 == 0, "%s: h=%d (multiple of 32) H=%d Hkv=%d F=%d (multiple of 32)", who, h, H, Hkv, F);
    AA_REQUIRE(ldc >= 2L * Hkv * 128 && ldc % 8 == 0 && Tmax > 0, "%s: ldc >= 2 * Hkv * 128 (multiple of 8), Tmax > 0", who);
    AA_REQUIRE(len != nullptr && bar != nullptr && status != nullptr, "%s: len, bar and status are required", who);
    p.x_in = (const bf16_t*)x_in; p.x_mid = (bf16_t*)x_mid; p.x_out = (bf16_t*)x_out; p.q = (bf16_t*)q; p.attn = (bf16_t*)attn; p.act = (bf16_t*)act;
    p.Wqkv = (const bf16_t*)Wqkv; p.Wo = (const bf16_t*)Wo; p.Wgu = (const bf16_t*)Wgu; p.Wdown = (const bf16_t*)Wdown; p.bqkv = (const bf16_t*)bqkv;
    p.M = M; p.h = h; p.H = H; p.Hkv = Hkv; p.F = F; p.eps = eps; p.scale = scale;
    p.epi = LayerEpi{pos, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)cache, ldc, Tmax, slot, H, Hkv};
    p.start = start; p.len = len; p.bar = (unsigned int*)bar; p.status = status;
    return AA_OK;
}

}  // namespace

// How many workgroups the persistent kernel may use on the current device: one per compute unit, provided the device can hold that many at once
// (hipOccupancyMaxActiveBlocksPerMultiprocessor >= 1); 0 when it cannot.
extern "C" int aa_decode_layer_grid(int* grid) {
    AA_REQUIRE(grid != nullptr, "aa_decode_layer_grid: grid is null");
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_layer_kernel, DL_NWAVE * 64, 0);
    if (e != hipSuccess) {
        aa_set_error("aa_decode_layer_grid: %s", hipGetErrorString(e));
        return AA_ERR_LAUNCH;
    }
    *grid = per_cu >= 1 ? cus : 0;
    return AA_OK;
}

// One decoder layer of a decode position (M <= 16 sequences, head_dim 128, SwiGLU MLP) from the rollout's strip-major weight copies:
//   Wqkv: aa_swizzle_weights_scaled_bf16 mode 2 with the input RMSNorm weight, Wgu: mode 1 with the post-attention RMSNorm weight,
//   Wo / Wdown: aa_swizzle_weights_bf16.
// x_in [M, h] -> x_out [M, h]; x_mid / q / attn / act are workspaces the caller owns ([M, h], [M, H*128], [M, H*128], [M, F]); the KV cache
// (rows [M * Tmax, ldc] = keys | values) gets the new token at slot[m].  bar: 2 uint32 zeroed ONCE by the caller and then left to the kernel;
// status: int32, becomes 1 if a grid barrier timed out (results are then garbage -- fall back to the per-step launches).
// grid: aa_decode_layer_grid()'s value.
extern "C" int aa_decode_layer_bf16(const void* x_in, void* x_mid, void* x_out, void* q, void* attn, void* act, const void* Wqkv, const void* Wo,
                                    const void* Wgu, const void* Wdown, const void* bqkv, int M, int h, int H, int Hkv, int F, float eps, float scale,
                                    const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc, int Tmax, const int64_t* slot,
                                    const int* start, const int* len, void* bar, int* status, int grid, void* stream) {
    AA_REQUIRE(grid > 0, "aa_decode_layer_bf16: a grid from aa_decode_layer_grid is required");
    DecodeLayerParams p;
    const int rc = dl_fill(p, x_in, x_mid, x_out, q, attn, act, Wqkv, Wo, Wgu, Wdown, bqkv, M, h, H, Hkv, F, eps, scale, pos, cos_t, sin_t, cache, ldc, Tmax, slot,
                           start, len, bar, status, "aa_decode_layer_bf16");
    if (rc != AA_OK) return rc;
    hipLaunchKernelGGL(decode_layer_kernel, dim3(grid), dim3(DL_NWAVE * 64), 0, (hipStream_t)stream, p);
    AA_CHECK_LAUNCH("aa_decode_layer_bf16");
    return AA_OK;
}

// All L layers of a decode position in ONE launch.  The per-layer arguments of aa_decode_layer_bf16 are packed once per rollout into a device array
// (`blocks`: L x aa_decode_layers_block_bytes bytes; layer l's x_out is layer l + 1's x_in: two alternating buffers), then every position is
// aa_decode_layers_bf16(blocks, L, grid, stream).  The copy of a block is stream-ordered (hipMemcpyAsync from a host struct that is consumed before the
// call returns: pageable memory, so the runtime stages it).
extern "C" int aa_decode_layers_block_bytes(int* bytes) {
    AA_REQUIRE(bytes != nullptr, "aa_decode_layers_block_bytes: bytes is null");
    *bytes = (int)sizeof(DecodeLayerParams);
    return AA_OK;
}
extern "C" int aa_decode_layers_pack(void* blocks, int layer, const void* x_in, void* x_mid, void* x_out, void* q, void* attn, void* act, const void* Wqkv,
                                     const void* Wo, const void* Wgu, const void* Wdown, const void* bqkv, int M, int h, int H, int Hkv, int F, float eps,
                                     float scale, const int* pos, const void* cos_t, const void* sin_t, void* cache, long ldc, int Tmax, const int64_t* slot,
                                     const int* start, const int* len, void* bar, int* status, void* stream) {
    AA_REQUIRE(blocks != nullptr && layer >= 0, "aa_decode_layers_pack: blocks / layer");
    DecodeLayerParams p;
    const int rc = dl_fill(p, x_in, x_mid, x_out, q, attn, act, Wqkv, Wo, Wgu, Wdown, bqkv, M, h, H, Hkv, F, eps, scale, pos, cos_t, sin_t, cache, ldc, Tmax, slot,
                           start, len, bar, status, "aa_decode_layers_pack");
    if (rc != AA_OK) return rc;
    const hipError_t e = hipMemcpyAsync((char*)blocks + (size_t)layer * sizeof(DecodeLayerParams), &p, sizeof(DecodeLayerParams), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
        aa_set_error("aa_decode_layers_pack: %s", hipGetErrorString(e));
        return AA_ERR_LAUNCH;
    }
    return AA_OK;
}
extern "C" int aa_decode_layers_bf16(const void* blocks, int L, int grid, void* stream) {
    AA_REQUIRE(blocks != nullptr && L >= 1 && grid > 0, "aa_decode_layers_bf16: blocks, L=%d >= 1 and a grid from aa_decode_layer_grid are required", L);
    hipLaunchKernelGGL(decode_layers_kernel, dim3(grid), dim3(DL_NWAVE * 64), 0, (hipStream_t)stream, (const DecodeLayerParams*)blocks, L);
    AA_CHECK_LAUNCH("aa_decode_layers_bf16");
    return AA_OK;
}
