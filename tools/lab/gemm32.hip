// bf16 GEMM on v_mfma_f32_32x32x16_bf16 (the higher-throughput MFMA shape of gfx950: 2382 vs 2075 TF in the
// micro-benchmarks of /opt/skills/guides/MI355X_MICROARCH.md).  Same contract, tile (256x256x64, 8 waves as
// 2(M) x 4(N), per-wave 128x64), LDS images, DMA staging and software-pipelined K loop as gemm.hip; only the
// fragment geometry differs:
//   operand fragment: lane l holds row (l & 31), 8 consecutive k at 16*k16 + 8*(l >> 5)
//   D (operands swapped: D = Wfrag x Xfrag): lane l holds m = l & 31, n = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
// so each lane still owns groups of 4 consecutive output columns (8-byte stores).
#include "gemm_params.h"

constexpr int BK32 = 64;
__device__ __forceinline__ int tr_swz32(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

template <bool A_T, bool B_N>
__global__ __launch_bounds__(512, 2) void gemm32_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, WN = 4, NW = 8;
    constexpr int TM = 128, TN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * BK32 * 2, B_BYTES = BN * BK32 * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = (A_BYTES / 1024) / NW, B_IT = (B_BYTES / 1024) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (wg % per_group) % gsz;
    const int tn = (wg % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* srcA[A_IT];
    const bf16_t* srcB[B_IT];
    long stepA, stepB;
    if constexpr (!A_T) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            srcA[j] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + ks * 8;
        }
        stepA = BK32;
    } else {
        constexpr int RPI = 1024 / (BM * 2), SPR = BM * 2 / 16;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz32(kr);
            srcA[j] = p.A + (long)kr * p.lda + min(m0 + unit * 16 + (s & 1) * 8, p.M - 8);
        }
        stepA = (long)BK32 * p.lda;
    }
    if constexpr (!B_N) {
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            srcB[j] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + ks * 8;
        }
        stepB = BK32;
    } else {
        constexpr int RPI = 1024 / (BN * 2), SPR = BN * 2 / 16;
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz32(kr);
            srcB[j] = p.B + (long)kr * p.ldb + min(n0 + unit * 16 + (s & 1) * 8, p.N - 8);
        }
        stepB = (long)BK32 * p.ldb;
    }
    auto stage = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)srcA[j], (lptr_t)(base + (wave + j * NW) * 1024), 16, 0, 0);
            srcA[j] += stepA;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)srcB[j], (lptr_t)(base + A_BYTES + (wave + j * NW) * 1024), 16, 0, 0);
            srcB[j] += stepB;
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, g4 = lane >> 4;
    // K-contiguous fragment: row = base32 + l31, 16-B slot = (2*k16 + hi) ^ ((l31 >> 1) & 7)
    int offK[4];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) offK[k16] = l31 * 128 + (((2 * k16 + hi) ^ ((l31 >> 1) & 7)) << 4);

    // one "half" = two 16-deep k-steps (k16 = 2*half, 2*half+1): 12 fragment loads, 16 MFMAs
    auto load_frags = [&](int buf, int half, bf16x8 (&af)[2][FM], bf16x8 (&bfr)[2][FN]) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k16 = half * 2 + s;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if constexpr (!A_T) {
                    af[s][i] = *reinterpret_cast<const bf16x8*>(sa + (wm * TM + i * 32) * 128 + offK[k16]);
                } else {
                    bf16x4 h[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int kr = k16 * 16 + (g4 >> 1) * 8 + hh * 4 + (l15 >> 2);
                        const int colb = wm * TM + i * 32 + (g4 & 1) * 16;
                        const int unit = (colb >> 4) ^ tr_swz32(kr);
                        const char* a = sa + kr * (BM * 2) + unit * 32 + (l15 & 3) * 8;
                        h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a);
                    }
                    af[s][i] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (!B_N) {
                    bfr[s][j] = *reinterpret_cast<const bf16x8*>(sb + (wn * TN + j * 32) * 128 + offK[k16]);
                } else {
                    bf16x4 h[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int kr = k16 * 16 + (g4 >> 1) * 8 + hh * 4 + (l15 >> 2);
                        const int colb = wn * TN + j * 32 + (g4 & 1) * 16;
                        const int unit = (colb >> 4) ^ tr_swz32(kr);
                        const char* a = sb + kr * (BN * 2) + unit * 32 + (l15 & 3) * 8;
                        h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a);
                    }
                    bfr[s][j] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
    };
    auto mfma_half = [&](const bf16x8 (&af)[2][FM], const bf16x8 (&bfr)[2][FN]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
    };

    const int nt = p.K / BK32;
    bf16x8 a0[2][FM], b0[2][FN], a1[2][FM], b1[2][FN];
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(0, 0, a0, b0);
    if (nt > 1) stage(1);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        load_frags(cur, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) load_frags(cur ^ 1, 0, a0, b0);
        if (t + 2 < nt) stage(cur);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: lane owns C[m][n .. n+3] for q = 0..3: n = nb + 8q + 4*hi
    const bool out_f32 = p.flags & AA_GEMM_OUT_F32;
    const bool accum = p.flags & AA_GEMM_ACCUM;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * TN + j * 32 + q * 8 + hi * 4;
                if (n >= p.N) continue;
                float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
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
                if (out_f32) {
                    float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    if (accum) { const f32x4 old = *reinterpret_cast<const f32x4*>(c); o += old; }
                    *reinterpret_cast<f32x4*>(c) = o;
                } else {
                    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
                    if (accum) {
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
        }
    }
}

template <bool A_T, bool B_N>
static int launch32(GemmParams& p, hipStream_t st) {
    p.tiles_m = aa_cdiv(p.M, 256);
    p.tiles_n = aa_cdiv(p.N, 256);
    constexpr int lds = 2 * (256 + 256) * BK32 * 2;
    auto kern = gemm32_kernel<A_T, B_N>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_bf16(32x32): cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_bf16(32x32)");
    return AA_OK;
}

int aa_gemm32_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if (!a_t && !b_n) return launch32<false, false>(p, st);
    if (!a_t && b_n) return launch32<false, true>(p, st);
    return launch32<true, true>(p, st);
}
