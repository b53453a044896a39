// Shared between gemm.hip (8-wave 16x16x32 MFMA kernels) and gemm4.hip (one wave per SIMD, 16x16x32; its 32x32x16 sibling is a recorded negative: tools/lab/gemm5).
#pragma once
#include "aa_common.h"

#define AA_ACT_NONE 0
#define AA_ACT_GELU 1
#define AA_ACT_QUICK_GELU 2
#define AA_ACT_RELU 3
#define AA_ACT_SILU 4

// flags
#define AA_GEMM_A_T 1         // A stored [K][M] (M contiguous) instead of [M][K]
#define AA_GEMM_B_N 2         // B stored [K][N] (N contiguous) instead of [N][K]
#define AA_GEMM_OUT_F32 4     // C is fp32 (default bf16)
#define AA_GEMM_ACCUM 8       // C += result

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gemm_act(float x, int act) {
    switch (act) {
        case AA_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case AA_ACT_QUICK_GELU: return x / (1.f + expf(-1.702f * x));
        case AA_ACT_RELU: return x > 0.f ? x : 0.f;
        case AA_ACT_SILU: return x / (1.f + expf(-x));
        default: return x;
    }
}

struct GemmParams {
    const bf16_t* A; const bf16_t* B; void* C;
    const bf16_t* bias;      // [N] or null
    const bf16_t* residual;  // [M, ldr] or null (added after bf16 rounding, like HF's `residual + x`)
    int M, N, K;
    long lda, ldb, ldc, ldr;
    int act, flags;
    int tiles_m, tiles_n;
    int gm;                  // tile-group height of the grouped tile order
    // grouped (mixture-of-experts) launches, aa_gemm_grouped_bf16; null / 0 for ordinary GEMMs
    const int* grp_tile_expert;  // mode 1: expert of every BM-row tile of the expert-major A / C (-1 = unused tile)
    const int* grp_off;          // mode 2: row offsets [E + 1] of the expert segments (the contraction range of expert e)
    long grp_strideB, grp_strideC;  // elements between consecutive experts' B (mode 1) / C (mode 2) matrices
    // fused epilogues of the one-wave-per-SIMD kernel (gemm4.hip), selected by `fuse`; 0 / null for ordinary GEMMs
    int fuse;                    // AA_FUSE_*
    const int* rope_pos;         // ROPE: position of every row of C
    const bf16_t* rope_cos;      //       [max_pos, 64] tables (head_dim 128)
    const bf16_t* rope_sin;
    int rope_cols;               //       columns [0, rope_cols) hold rotary heads (q and k of a fused [q|k|v] projection)
    void* aux;                   // GLU_FWD: act [M, F] output;  GLU_BWD: d[gate|up] [M, 2F] output (C is not written)
    const bf16_t* aux_in;        // GLU_BWD: the saved [gate|up] [M, 2F]
    long ldaux, ldaux_in;
    int glu_f;                   // GLU_*: F (C / B of GLU_FWD have 2F columns / rows: gate block then up block)
};
#define AA_FUSE_NONE 0
#define AA_FUSE_ROPE 1       // C = rope(A W^T) on the q / k heads of a fused qkv projection (hf apply_rotary_pos_emb, bf16 rounding points)
#define AA_FUSE_GLU_FWD 2    // C = [gate | up] = A [Wg; Wu]^T and aux = silu(gate) * up (LlamaMLP), one tile owning both halves of its columns
#define AA_FUSE_GLU_BWD 3    // aux = d[gate | up] from d_act = A W (never stored) and the saved [gate | up]

constexpr int BK = 64;


// Epilogue of one lane-owned group of 4 consecutive output columns C[m][n..n+3] (HF rounding points: bias,
// activation on the bf16-rounded value, residual added after bf16 rounding, optional accumulate / fp32 out).
__device__ __forceinline__ void gemm_store4(const GemmParams& p, int m, int n, float v0, float v1, float v2, float v3) {
    float v[4] = {v0, v1, v2, v3};
    if (p.bias) {
        const u16x4 b = *reinterpret_cast<const u16x4*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
    }
    if (p.act != AA_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gemm_act(rbf(v[e]), p.act);
    }
    if (p.residual) {
        const u16x4 r = *reinterpret_cast<const u16x4*>(p.residual + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) + bf2f(r[e]);
    }
    if (p.flags & AA_GEMM_OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if (p.flags & AA_GEMM_ACCUM) { const f32x4 old = *reinterpret_cast<const f32x4*>(c); o += old; }
        *reinterpret_cast<f32x4*>(c) = o;
    } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
        if (p.flags & AA_GEMM_ACCUM) {
            const u16x4 old = *reinterpret_cast<const u16x4*>(c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(old[e]);
        }
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<u16x4*>(c) = o;
    }
}

// one-wave-per-SIMD 256x256 kernel, accumulators in the accumulator file (gemm4.hip)
bool aa_gemm4_supports(int K);      // the 4-slot ring walks K in trips of 128
int aa_gemm4_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st);
// fused-epilogue launches of the same kernel (p.fuse); 1 = shape does not qualify, run the unfused kernels
int aa_gemm4_fused(GemmParams& p, hipStream_t st);
int aa_gemm4_grouped(GemmParams& p, bool b_n, hipStream_t st);   // MoE rows on the 256 x 256 one-wave-per-SIMD tile (segments aligned to 256 rows); 1 = shape does not fit
