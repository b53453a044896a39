// Shared between gemm.hip (16x16x32 MFMA kernels) and gemm32.hip (32x32x16 MFMA kernel).
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
};

constexpr int BK = 64;


// 32x32x16-MFMA 256x256 kernel (gemm32.hip)
int aa_gemm32_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st);
