// bf16 GEMM, 256x256x64 tile, 8 waves -- "split-K ring" schedule.
//
// Same contract / MFMA mapping / epilogue as gemm.hip; what changes is the LDS ring and the synchronisation:
//   * every K-tile is staged as TWO 32-deep sub-tiles (kk = 0, 1), each with its own 32 KiB slot
//     (A part 16 KiB + B part 16 KiB); the ring holds 2 K-tiles = 4 slots = 128 KiB (as before);
//   * a sub-tile slot is free as soon as ITS fragment reads are done, so the DMA of sub-tile (kk, t+2) is issued
//     half an iteration earlier than in the whole-tile ring: every DMA has 1.5 iterations (3 phases of 32 MFMAs)
//     to land instead of 1.0;
//   * the loop never drains the VMEM queue: each phase boundary is  s_waitcnt lgkmcnt(0) ; s_waitcnt vmcnt(8) ;
//     s_barrier  (raw barrier, 8 = the two younger sub-tile DMAs of 4 loads each that may stay in flight).
// Phase structure of iteration t (F0/F1 = the two named fragment sets):
//   A(t): F1 <- (kk=1, t)     | MFMA(F0) | DMA (kk=0, t+2) -> slot of (kk=0, t)   [its reads finished in C(t-1)]
//   B1  : wait (kk=0, t+1) landed, barrier
//   C(t): F0 <- (kk=0, t+1)   | MFMA(F1) | DMA (kk=1, t+2) -> slot of (kk=1, t)   [its reads finished in A(t)]
//   B2  : wait (kk=1, t+1) landed, barrier
// LDS images: K-contiguous operand part = [256 rows][32 k] (64-B rows, 16-B slot ^= {0,2,3,1}[(row>>2)&3], conflict
// free for ds_read_b128); row-contiguous operand part = [32 k][256 cols] (the halves of the gemm.hip image).
#include "gemm_params.h"

constexpr int RBK = 64;
__device__ __forceinline__ int ring_kswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
__device__ __forceinline__ int ring_trswz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

#define RING_SYNC(N)  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")

template <bool A_T, bool B_N>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, WN = 4, NW = 8;
    constexpr int TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int PART = 16 * 1024;          // one operand part of a sub-tile
    constexpr int SLOT = 2 * PART;           // sub-tile slot: A part | B part
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 slots: (kt&1)*2 + kk

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g = lane >> 4;

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

    // ---- per-lane DMA sources for sub-tile (kk = 0) of K-tile 0; two 1-KiB pieces per operand per sub-tile
    const bf16_t* srcA[2];
    const bf16_t* srcB[2];
    long stepA, stepB, halfA, halfB;
    if constexpr (!A_T) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wave + j * NW;               // piece = 16 rows x 64 B
            const int r = c * 16 + (lane >> 2);
            const int ks = (lane & 3) ^ ring_kswz(r);
            srcA[j] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + ks * 8;
        }
        stepA = RBK; halfA = 32;
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wave + j * NW;               // piece = 2 k-rows x 512 B
            const int kr = c * 2 + (lane >> 5);
            const int s = lane & 31;
            const int unit = (s >> 1) ^ ring_trswz(kr);
            srcA[j] = p.A + (long)kr * p.lda + min(m0 + unit * 16 + (s & 1) * 8, p.M - 8);
        }
        stepA = (long)RBK * p.lda; halfA = 32 * p.lda;
    }
    if constexpr (!B_N) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wave + j * NW;
            const int r = c * 16 + (lane >> 2);
            const int ks = (lane & 3) ^ ring_kswz(r);
            srcB[j] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + ks * 8;
        }
        stepB = RBK; halfB = 32;
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wave + j * NW;
            const int kr = c * 2 + (lane >> 5);
            const int s = lane & 31;
            const int unit = (s >> 1) ^ ring_trswz(kr);
            srcB[j] = p.B + (long)kr * p.ldb + min(n0 + unit * 16 + (s & 1) * 8, p.N - 8);
        }
        stepB = (long)RBK * p.ldb; halfB = 32 * p.ldb;
    }
    // issue the 4 DMA pieces of sub-tile kk of the K-tile the pointers currently address, into ring slot `slot`
    auto dma = [&](int slot, int kk) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + (kk ? halfA : 0)), (lptr_t)(base + (wave + j * NW) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + (kk ? halfB : 0)), (lptr_t)(base + PART + (wave + j * NW) * 1024), 16, 0, 0);
    };
    auto advance = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) { srcA[j] += stepA; srcB[j] += stepB; }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int offK = l15 * 64 + ((g ^ ring_kswz(l15)) << 4);   // K-contiguous: row = base16 + l15, 16-B slot g
    auto load_frags = [&](int slot, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
        const char* sa = smem + slot * SLOT;
        const char* sb = sa + PART;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (!A_T) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * TM + i * 16) * 64 + offK);
            } else {
                bf16x4 h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kr = g * 8 + hh * 4 + (l15 >> 2);
                    const int unit = ((wm * TM + i * 16) >> 4) ^ ring_trswz(kr);
                    const char* a = sa + kr * (BM * 2) + unit * 32 + (l15 & 3) * 8;
                    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a);
                }
                af[i] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (!B_N) {
                bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (wn * TN + j * 16) * 64 + offK);
            } else {
                bf16x4 h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kr = g * 8 + hh * 4 + (l15 >> 2);
                    const int unit = ((wn * TN + j * 16) >> 4) ^ ring_trswz(kr);
                    const char* a = sb + kr * (BN * 2) + unit * 32 + (l15 & 3) * 8;
                    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a);
                }
                bfr[j] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto mfma_step = [&](const bf16x8 (&af)[FM], const bf16x8 (&bfr)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };
    constexpr int NREADS = (A_T ? 2 * FM : FM) + (B_N ? 2 * FN : FN);
    constexpr int RATIO = ((FM * FN) / NREADS) > 0 ? ((FM * FN) / NREADS) : 1;
#define RING_ILV()                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NREADS; ++i_) {                     \
        __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
    }

    const int nt = p.K / RBK;   // host guarantees nt >= 4
    bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
    // ---- prologue: K-tiles 0 and 1 entirely (4 sub-tiles, 16 loads per wave)
    dma(0, 0); dma(1, 1); advance();
    dma(2, 0); dma(3, 1); advance();            // pointers now address K-tile 2
    RING_SYNC(8);                               // (kk0,0) and (kk1,0) landed for everyone
    load_frags(0, a0, b0);
    // slot 0 is re-used by DMA (kk0, 2) in A(0): every wave must be past its reads of (kk0, 0) first
    RING_SYNC(8);
    int t = 0;
    for (; t + 2 < nt; ++t) {
        const int s0 = (t & 1) * 2;             // slots of K-tile t (and of t+2); K-tile t+1 lives in s0 ^ 2
        // ---- A(t)
        load_frags(s0 + 1, a1, b1);
        mfma_step(a0, b0);
        RING_ILV();
        __builtin_amdgcn_sched_barrier(0);
        dma(s0, 0);                             // (kk0, t+2) -> slot of (kk0, t)
        RING_SYNC(8);                           // B1: (kk0, t+1) landed; younger: (kk1,t+1), (kk0,t+2)
        // ---- C(t)
        load_frags((s0 ^ 2), a0, b0);
        mfma_step(a1, b1);
        RING_ILV();
        __builtin_amdgcn_sched_barrier(0);
        dma(s0 + 1, 1); advance();              // (kk1, t+2) -> slot of (kk1, t)
        RING_SYNC(8);                           // B2: (kk1, t+1) landed; younger: (kk0,t+2), (kk1,t+2)
    }
    // ---- the last two K-tiles: nothing left to stage, drain conservatively
    for (; t < nt; ++t) {
        const int s0 = (t & 1) * 2;
        load_frags(s0 + 1, a1, b1);
        mfma_step(a0, b0);
        RING_ILV();
        __builtin_amdgcn_sched_barrier(0);
        RING_SYNC(0);
        load_frags((s0 ^ 2), a0, b0);           // (last iteration: stale slot, result unused)
        mfma_step(a1, b1);
        RING_ILV();
        __builtin_amdgcn_sched_barrier(0);
        RING_SYNC(0);
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = .. + l15, n = .. + g*4
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * TN + j * 16 + g * 4;
            if (n >= p.N) continue;
            gemm_store4(p, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

template <bool A_T, bool B_N>
static int launch_ring(GemmParams& p, hipStream_t st) {
    p.tiles_m = aa_cdiv(p.M, 256);
    p.tiles_n = aa_cdiv(p.N, 256);
    constexpr int lds = 4 * 32 * 1024;
    auto kern = gemm_ring_kernel<A_T, B_N>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_bf16(ring): cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_bf16(ring)");
    return AA_OK;
}

int aa_gemm_ring_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if (!a_t && !b_n) return launch_ring<false, false>(p, st);
    if (!a_t && b_n) return launch_ring<false, true>(p, st);
    return launch_ring<true, true>(p, st);
}
