// Shared device/host helpers for the MI355X (gfx950) DPO/PPO hot-path kernels.
// gfx950 only: wave = 64 lanes, bf16 MFMA 16x16x32, LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "aa_ctx.h"

typedef unsigned short bf16_t;  // raw bf16 storage (torch.bfloat16 bit pattern)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

#define AA_WAVE 64
// Inline-asm statements that contain `s_add_u32` write SCC, and the compiler must be told: AA_SCC is that clobber.  Rounds 2 - 5 shipped the LDS-DMA
// statements of gemm4.hip / attn128.inc without it; round 6 found hipcc scheduling 64-bit address adds ACROSS them (s_add_u32 lo ... asm ... s_addc_u32 hi:
// eight sites in five gemm4 kernels, the carry of the low word replaced by the asm's) -- wrong only when an operand's first stages straddle a 4 GB boundary,
// i.e. silently and almost never -- and a compare / select pair in a lab kernel (profiles/r06_scc_clobber.txt).  tests/test_host_logic.py scans the
// generated ISA for the pattern.  -DAA_NO_DECLARE_SCC rebuilds the old behaviour for the same-box A/B.
#ifdef AA_NO_DECLARE_SCC
#define AA_SCC
#else
#define AA_SCC , "scc"
#endif
#define AA_MAX_DEVICES 16     // per-device caches of launch parameters (one node: 8 GPUs)

__device__ __forceinline__ float bf2f(bf16_t u) {
    return __builtin_bit_cast(float, (uint32_t)u << 16);
}
// round-to-nearest-even, lowers to v_cvt_pk_bf16_f32 on gfx950 (same rounding as torch)
__device__ __forceinline__ bf16_t f2bf(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
// round a float through bf16 (emulates a bf16 intermediate of the HF graph)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// sigmoid(x) of the bf16 PRODUCTION kernels (SwiGLU forward / backward: element-wise kernels, the gemm4 epilogues, the decode strip kernel): the
// hardware's v_exp_f32 and v_rcp_f32 (1 ulp each) instead of OCML's expf (range reduction, ~15 instructions) and an IEEE division (~10): every
// caller rounds the result to bf16 (2^-9), four orders of magnitude coarser.  Inside a one-workgroup-per-CU GEMM tile the precise form was ~60
// VALU instructions per output element of the SwiGLU epilogues -- 5-8 % of the whole gate_up GEMM, 17 % of the down-projection dX GEMM.
// The fp32 parity-mode instantiations (elementwise_f32.hip) keep expf and the division: aa_sigmoid<true>.
template <bool PRECISE>
__device__ __forceinline__ float aa_sigmoid(float x) {
    if constexpr (PRECISE) return 1.f / (1.f + expf(-x));
    else return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// ---------------------------------------------------------------- element type of the block kernels
// elementwise.hip is written against elem_t / ev8 / ev4 / e2f / f2e / ernd and compiled twice: as is (bf16
// activations, HF rounding points kept: the production path) and through elementwise_f32.hip with AA_ELEM_F32
// (fp32 activations, rounding points vanish: the fp32 parity mode that tracks the reference's fp32 CPU trainer).
// AA_FN() appends _f32 to the C-ABI names of the second instantiation.
typedef __attribute__((ext_vector_type(8))) float f32x8;
#ifdef AA_ELEM_F32
typedef float elem_t;
typedef f32x8 ev8;
typedef f32x4 ev4;
__device__ __forceinline__ float e2f(float x) { return x; }
__device__ __forceinline__ float f2e(float x) { return x; }
__device__ __forceinline__ float ernd(float x) { return x; }
#define AA_FN(name) name##_f32
#define AA_FN2(name, name_f32) name_f32
#define AA_ELEM_NS aa_elem_f32
#define AA_ELEM_PRECISE true
#else
typedef bf16_t elem_t;
typedef u16x8 ev8;
typedef u16x4 ev4;
__device__ __forceinline__ float e2f(bf16_t x) { return bf2f(x); }
__device__ __forceinline__ bf16_t f2e(float x) { return f2bf(x); }
__device__ __forceinline__ float ernd(float x) { return rbf(x); }
#define AA_FN(name) name
#define AA_FN2(name, name_f32) name
#define AA_ELEM_NS aa_elem_bf16
#define AA_ELEM_PRECISE false
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64, <= 1024). `red` is >= NT/64 floats of LDS.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---------------------------------------------------------------- host side
extern "C" const char* aa_last_error(void);
void aa_set_error(const char* fmt, ...);

#define AA_OK 0
#define AA_ERR_ARG (-1)
#define AA_ERR_LAUNCH (-2)

#define AA_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            aa_set_error(__VA_ARGS__);             \
            return AA_ERR_ARG;                     \
        }                                          \
    } while (0)

#define AA_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            aa_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return AA_ERR_LAUNCH;                                               \
        }                                                                       \
    } while (0)

static inline int aa_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
