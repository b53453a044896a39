// bf16 MFMA GEMM for gfx950 (CDNA4):  C[M,N] (+)= op(A)[M,K] * op(B)[K,N]  with fp32 accumulation.
//
// Replaces the torch/rocBLAS `nn.Linear` calls that `model(**batch)` executes in the reference hot path
// (align_anything/trainers/text_to_text/dpo.py:128 -> hf:models/llama/modeling_llama.py q/k/v/o/gate/up/down,
//  hf:models/llava/modeling_llava.py:87-106 projector, :361 lm_head) and their autograd backward.
//
// Design (MI355X-first):
//  * 64-lane waves, v_mfma_f32_16x16x32_bf16, operands swapped (D = Wfrag x Afrag) so each lane owns 4
//    consecutive output columns -> 8-byte bf16x4 epilogue stores.
//  * Operand tiles go HBM -> LDS with direct-to-LDS DMA (global_load_lds_dwordx4, 1 KiB per wave
//    instruction), double-buffered over BK = 64, one barrier per K-tile.
//  * LDS images are lane-linear (DMA constraint); bank-conflict-free reads come from an XOR swizzle
//    applied to the per-lane *source* address and mirrored on the ds_read side.
//  * Layouts: an operand is either K-contiguous (row-major [rows][K], read with ds_read_b128) or
//    row-contiguous ([K][rows], read with the gfx950 transpose read ds_read_b64_tr_b16), which gives
//    NT (forward), NN (dX = dY*W) and TN (dW = dY^T*X) without materialising transposes.
//  * 1-D grid with a bijective XCD-aware remap + grouped tile order so neighbouring tiles share an L2.
#include "aa_common.h"

#include "gemm_params.h"

// K-contiguous tile: [R rows][64 k] bf16, row = 128 B = 8 slots of 16 B; slot ^= (row>>1)&7.
// row-contiguous tile: [64 k rows][R cols] bf16; 32-B unit ^= (krow&3) | ((krow>>3)&1)<<2.
__device__ __forceinline__ int tr_swz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// GRP (mixture-of-experts grouped launches, aa_gemm_grouped_bf16):
//   1: the rows of A / C are an expert-major token buffer whose BM-row tiles each belong to one expert (grp_tile_expert);
//      that expert's weight matrix B + e * grp_strideB is the other operand (forward NT and dX NN of the expert MLPs)
//   2: blockIdx.y = expert; the contraction runs over that expert's row segment [grp_off[e], grp_off[e+1]) of A and B
//      (both row-contiguous, TN) and the result goes to C + e * grp_strideC (the per-expert weight gradients)
template <int BM, int BN, int WM, int WN, bool A_T, bool B_N, bool PIPE, int ILV, int GRP = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8) ? 2 : 1)
void gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = (A_BYTES / 1024) / NW, B_IT = (B_BYTES / 1024) / NW;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the wave count");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware bijective remap, then grouped (GM rows of tiles) order
    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int GM = p.gm & 0xff;   // tile-group height (tiles of one group share panels in an XCD's L2); see pick_group()
    int tm, tn;
    if (!(p.gm & 0x100)) {        // groups of GM tile rows, swept column by column
        const int per_group = GM * p.tiles_n;
        const int group = wg / per_group;
        const int first_m = group * GM;
        const int gsz = min(p.tiles_m - first_m, GM);
        tm = first_m + (wg % per_group) % gsz;
        tn = (wg % per_group) / gsz;
    } else {                      // groups of GM tile columns, swept row by row
        const int per_group = GM * p.tiles_m;
        const int group = wg / per_group;
        const int first_n = group * GM;
        const int gsz = min(p.tiles_n - first_n, GM);
        tn = first_n + (wg % per_group) % gsz;
        tm = (wg % per_group) / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* Ap = p.A;
    const bf16_t* Bp = p.B;
    void* Cp = p.C;
    int Kp = p.K;
    if constexpr (GRP == 1) {
        const int e = p.grp_tile_expert[tm];
        if (e < 0) {                                        // tile beyond the rows in use (uniform per block): defined output (zeros) without a memset
            if (!(p.flags & AA_GEMM_ACCUM)) {               // of the whole buffer before every launch (round 3: 5 ms of fills per 12-layer step)
                for (int i = threadIdx.x; i < BM * (BN / 4); i += NW * 64) {
                    const int m = m0 + i / (BN / 4), n = n0 + (i % (BN / 4)) * 4;
                    if (m < p.M && n < p.N) {
                        if (p.flags & AA_GEMM_OUT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cp) + (long)m * p.ldc + n) = f32x4{0.f, 0.f, 0.f, 0.f};
                        else *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_t*>(Cp) + (long)m * p.ldc + n) = u16x4{0, 0, 0, 0};
                    }
                }
            }
            return;
        }
        Bp += (long)e * p.grp_strideB;
    } else if constexpr (GRP == 2) {
        const int e = blockIdx.y;
        const int r0 = p.grp_off[e], r1 = p.grp_off[e + 1];
        Ap += (long)r0 * p.lda;
        Bp += (long)r0 * p.ldb;
        Kp = r1 - r0;
        const bool f32o = p.flags & AA_GEMM_OUT_F32;
        Cp = f32o ? (void*)(reinterpret_cast<float*>(p.C) + (long)e * p.grp_strideC)
                  : (void*)(reinterpret_cast<bf16_t*>(p.C) + (long)e * p.grp_strideC);
        if (Kp == 0) {                                      // expert without tokens: its gradient tile is zero
            if (!(p.flags & AA_GEMM_ACCUM)) {
                for (int i = threadIdx.x; i < BM * (BN / 4); i += NW * 64) {
                    const int m = m0 + i / (BN / 4), n = n0 + (i % (BN / 4)) * 4;
                    if (m < p.M && n < p.N) {
                        if (f32o) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cp) + (long)m * p.ldc + n) = f32x4{0.f, 0.f, 0.f, 0.f};
                        else *reinterpret_cast<u16x4*>(reinterpret_cast<bf16_t*>(Cp) + (long)m * p.ldc + n) = u16x4{0, 0, 0, 0};
                    }
                }
            }
            return;
        }
    } else if constexpr (GRP == 3) {
        // split-K (aa_gemm_splitk_bf16): blockIdx.y = chunk of the contraction, [k0, k0 + grp_strideB); the fp32 partial product goes to C + chunk * grp_strideC
        // and splitk_reduce_kernel sums the chunks in order and applies the epilogue.  For few-row launches (a rollout's scoring forwards, M ~ 1000) whose
        // weight matrix would otherwise stream through a quarter of the compute units.
        const int k0 = blockIdx.y * (int)p.grp_strideB;
        Kp = min((int)p.grp_strideB, p.K - k0);
        Ap += A_T ? (long)k0 * p.lda : (long)k0;
        Bp += B_N ? (long)k0 * p.ldb : (long)k0;
        Cp = (void*)(reinterpret_cast<float*>(p.C) + (long)blockIdx.y * p.grp_strideC);
    }

    // ---- per-lane DMA source pointers (advance by one K-tile per stage)
    const bf16_t* srcA[A_IT];
    const bf16_t* srcB[B_IT];
    long stepA, stepB;
    if constexpr (!A_T) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            const int gr = min(m0 + r, p.M - 1);
            srcA[j] = Ap + (long)gr * p.lda + ks * 8;
        }
        stepA = BK;
    } else {
        // tile [64 k][BM]: one DMA = 1 KiB = RPI k-rows of BM*2 bytes
        constexpr int RPI = 1024 / (BM * 2);
        constexpr int SPR = BM * 2 / 16;  // 16-B slots per k-row
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz(kr);
            int col = m0 + unit * 16 + (s & 1) * 8;
            col = min(col, p.M - 8);  // M % 8 == 0 enforced on host for transposed operands
            srcA[j] = Ap + (long)kr * p.lda + col;
        }
        stepA = (long)BK * p.lda;
    }
    if constexpr (!B_N) {
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            const int gr = min(n0 + r, p.N - 1);
            srcB[j] = Bp + (long)gr * p.ldb + ks * 8;
        }
        stepB = BK;
    } else {
        constexpr int RPI = 1024 / (BN * 2);
        constexpr int SPR = BN * 2 / 16;
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz(kr);
            int col = n0 + unit * 16 + (s & 1) * 8;
            col = min(col, p.N - 8);
            srcB[j] = Bp + (long)kr * p.ldb + col;
        }
        stepB = (long)BK * p.ldb;
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

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane LDS read offsets
    const int l15 = lane & 15, g = lane >> 4;
    // K-contiguous: row = base16 + l15 ; slot = (kk*4+g) ^ ((l15>>1)&7)
    int offK[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) offK[kk] = l15 * 128 + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) << 4);
    // row-contiguous (transpose read): k-row = kk*32 + 8g + 4*hh + (l15>>2), col = base16 + (l15&3)*4

    // fragment loads for one 32-deep k-step (kk) of buffer `buf` into a named register set
    auto load_frags = [&](int buf, int kk, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (!A_T) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * TM + i * 16) * 128 + offK[kk]);
            } else {
                bf16x4 h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kr = kk * 32 + g * 8 + hh * 4 + (l15 >> 2);
                    const int colb = wm * TM + i * 16;  // multiple of 16 -> unit index
                    const int unit = (colb >> 4) ^ tr_swz(kr);
                    const char* a = sa + kr * (BM * 2) + unit * 32 + (l15 & 3) * 8;
                    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a);
                }
                af[i] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (!B_N) {
                bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (wn * TN + j * 16) * 128 + offK[kk]);
            } else {
                bf16x4 h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kr = kk * 32 + g * 8 + hh * 4 + (l15 >> 2);
                    const int colb = wn * TN + j * 16;
                    const int unit = (colb >> 4) ^ tr_swz(kr);
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

    const int nt = Kp / BK;
    if constexpr (!PIPE) {
        // simple schedule: one barrier per K-tile, fragment reads interleaved by the compiler
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) stage(cur ^ 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[FM], bfr[FN];
                load_frags(cur, kk, af, bfr);
                mfma_step(af, bfr);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // software-pipelined schedule: two named fragment sets; every batch of ds_reads is issued BEFORE a
        // 32-MFMA batch that does not depend on it, and the barrier sits between the two batches of a tile:
        //   A: F1 <- tile t (kk=1)      | MFMA(F0)          B: wait DMA(t+1), barrier
        //   C: F0 <- tile t+1 (kk=0), DMA tile t+2 -> freed buffer | MFMA(F1)
        bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
        // ds_read instructions per fragment set and the MFMA:ds_read interleave ratio used when ILV is on
        constexpr int NREADS = (A_T ? 2 * FM : FM) + (B_N ? 2 * FN : FN);
        constexpr int NMFMA = FM * FN;
        constexpr int RATIO = (NMFMA / NREADS) > 0 ? (NMFMA / NREADS) : 1;
        if constexpr (ILV == 0) {
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            load_frags(0, 0, a0, b0);
            if (nt > 1) stage(1);
            for (int t = 0; t < nt; ++t) {
                const int cur = t & 1;
                load_frags(cur, 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // tile t+1 landed for everyone; every wave is done reading buf[cur]
                if (t + 1 < nt) load_frags(cur ^ 1, 0, a0, b0);
                if (t + 2 < nt) stage(cur);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (ILV == 2) {
            // Peeled variant: the main loop body is branch-free (tile t+2 is always staged), so phase C is one
            // scheduling region too: LDS reads of the next fragment set trickle between the first MFMAs and the
            // DMA issues between the last ones; the two final iterations (nothing left to stage) are peeled.
            constexpr int NDMA = A_IT + B_IT;
            constexpr int RATIO_C = ((NMFMA - NDMA) / NREADS) > 0 ? ((NMFMA - NDMA) / NREADS) : 1;
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            load_frags(0, 0, a0, b0);
            if (nt > 1) stage(1);
            int t = 0;
            for (; t + 2 < nt; ++t) {
                const int cur = t & 1;
                load_frags(cur, 1, a1, b1);
                mfma_step(a0, b0);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                load_frags(cur ^ 1, 0, a0, b0);
                stage(cur);
                mfma_step(a1, b1);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO_C, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NDMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; t < nt; ++t) {
                const int cur = t & 1;
                load_frags(cur, 1, a1, b1);
                mfma_step(a0, b0);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                load_frags(cur ^ 1, 0, a0, b0);  // (last iteration: stale buffer, result unused)
                mfma_step(a1, b1);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // Measured best (profiles/r01_gemm_schedule_ab.json): trickle the LDS reads of the next fragment set
            // between the MFMAs of phase A (the matrix pipe starts at once after the barrier-free boundary), keep
            // phase C as reads -> DMA issue -> MFMAs.  A fully branch-free body with the DMA issues interleaved
            // as well measured 3-6 % slower on the NT shapes.
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            load_frags(0, 0, a0, b0);
            if (nt > 1) stage(1);
            for (int t = 0; t < nt; ++t) {
                const int cur = t & 1;
                load_frags(cur, 1, a1, b1);
                mfma_step(a0, b0);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // tile t+1 landed for everyone; every wave is done reading buf[cur]
                load_frags(cur ^ 1, 0, a0, b0);  // (last iteration: stale buffer, result unused)
                if (t + 2 < nt) stage(cur);
                mfma_step(a1, b1);
#pragma unroll
                for (int i = 0; i < NREADS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = .. + l15, n = .. + g*4
    const bool out_f32 = p.flags & AA_GEMM_OUT_F32;
    const bool accum = p.flags & AA_GEMM_ACCUM;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * TN + j * 16 + g * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
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
                float* c = reinterpret_cast<float*>(Cp) + (long)m * p.ldc + n;
                f32x4 o = {v[0], v[1], v[2], v[3]};
                if (accum) { const f32x4 old = *reinterpret_cast<const f32x4*>(c); o += old; }
                *reinterpret_cast<f32x4*>(c) = o;
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(Cp) + (long)m * p.ldc + n;
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

// tile-group height of the grouped tile order: aa_ctx::gm, 0 = heuristic (AA_GEMM_GM / aa_gemm_set_group override)


// Height (in tiles) of the groups of the L2-aware tile order.  Same-process sweep on the 7B shapes at M = 16384
// (tools/bench_gemm_gm.py, profiles/r01_gemm_group_height.txt): 4 beats the former 8 by 0-7 % (NN most), 3 is best for NN with a wide
// N (down-projection dX, +8 %), 16 / 32 lose 3-15 %.
// Second sweep (same file): with few tile columns (N <= 4096) and a long contraction (K >= 8192) grouping tile COLUMNS and
// sweeping the rows is 3-6.5 % faster (down-projection forward, qkv / gate_up dX and dW); otherwise rows stay grouped.
static int pick_group(bool a_t, bool b_n, int tiles_n, int K) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("AA_GEMM_GM"); env = e ? atoi(e) : 0; }
    if (aa_ctx_cur()->gm > 0) return aa_ctx_cur()->gm;
    if (env > 0) return env;
    if (tiles_n <= 16 && K >= 8192) return 0x100 | 4;
    if (!a_t && b_n && tiles_n > 32) return 3;
    return 4;
}

template <int BM, int BN, int WM, int WN, bool A_T, bool B_N, bool PIPE, int ILV>
static int launch_cfg2(GemmParams& p, hipStream_t st) {
    p.tiles_m = aa_cdiv(p.M, BM);
    p.tiles_n = aa_cdiv(p.N, BN);
    p.gm = pick_group(A_T, B_N, p.tiles_n, p.K);
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_kernel<BM, BN, WM, WN, A_T, B_N, PIPE, ILV>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_bf16: cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_bf16");
    return AA_OK;
}

template <int BM, int BN, int WM, int WN, bool A_T, bool B_N>
static int launch_cfg(GemmParams& p, hipStream_t st) {
    if constexpr (BM == 256 && BN == 256 && WM * WN == 4) {
        return launch_cfg2<BM, BN, WM, WN, A_T, B_N, true, 1>(p, st);     // one-wave-per-SIMD tile: the phase-A interleave only
    } else if constexpr (BM == 256 && BN == 256) {
        // the schedule each layout measured best with (profiles/r01_gemm_*_ab.json): NN = fully interleaved peeled loop (2), NT / TN = phase-A
        // interleave (1).  The other schedules (simple pipeline, no interleave) were A/B variants of rounds 1-2 and are no longer instantiated.
        if constexpr (!A_T && B_N) return launch_cfg2<BM, BN, WM, WN, A_T, B_N, true, 2>(p, st);
        else return launch_cfg2<BM, BN, WM, WN, A_T, B_N, true, 1>(p, st);
    } else {
        return launch_cfg2<BM, BN, WM, WN, A_T, B_N, true, 0>(p, st);
    }
}

template <bool A_T, bool B_N>
static int launch_layout(GemmParams& p, int tile, hipStream_t st) {
    switch (tile) {
        case 0: return launch_cfg<256, 256, 2, 4, A_T, B_N>(p, st);
        case 1: return launch_cfg<128, 128, 2, 2, A_T, B_N>(p, st);
        case 2: return launch_cfg<256, 128, 4, 2, A_T, B_N>(p, st);
        case 3: return launch_cfg<128, 256, 2, 4, A_T, B_N>(p, st);
        default: aa_set_error("aa_gemm_bf16: unknown tile config %d", tile); return AA_ERR_ARG;
    }
}

// forced tile config: aa_ctx::force_tile (-2: read env once; -1: heuristic)

// waves-quantisation heuristic over the 256-CU chip
static int pick_tile(int M, int N, int K) {
    if (aa_ctx_cur()->force_tile == -2) {
        const char* e = getenv("AA_GEMM_TILE");
        aa_ctx_cur()->force_tile = e ? atoi(e) : -1;
    }
    if (aa_ctx_cur()->force_tile >= 0) return aa_ctx_cur()->force_tile;
    static int g4 = -1;      // AA_GEMM_G4=0 keeps the 8-wave kernel for the 256x256 tile (A/B runs)
    if (g4 < 0) { const char* e = getenv("AA_GEMM_G4"); g4 = e ? atoi(e) : 1; }
    struct Cfg { int bm, bn, slots; float eff; };
    // slots = concurrently resident tiles on the chip; eff = relative per-flop efficiency of the config
    // (round 6: the 8-wave tiles are rated against gemm4, which runs the 256 x 256 tile since round 2, not against the 8-wave kernel of that tile: measured
    // 1035 TFLOP/s on 256 x 128 against ~1400 on gemm4 at M = 10240, N = 4096 (profiles/r06_dpo7b_packed_kernel_stats.csv) -- with 0.86 the model sent the
    // 640-tile shapes of the packed step to five half rounds of the 8-wave kernel instead of three rounds of gemm4.  No shape of the unpacked step changes.)
    const Cfg cfgs[4] = {{256, 256, 256, 1.00f}, {128, 128, 512, g4 ? 0.62f : 0.72f}, {256, 128, 256, g4 ? 0.76f : 0.86f}, {128, 256, 256, g4 ? 0.76f : 0.86f}};
    // Short contractions (K < 2048: the CLIP / ViT / Whisper towers, OPT-125m): a 256 x 256 tile spends its time in the pipeline fill and in the epilogue of
    // 65536 outputs, not in its 16 k-steps -- the tower's fc1 (M 2308, N 4096, K 1024, bias + quick-GELU) ran 292 us on the one-wave-per-SIMD kernel and
    // 137 us on the 8-wave 256 x 256 one against 59 us on 256 x 128 although that needs two rounds (tools/bench_clip_gemms.py, profiles/r04_clip_gemms.json):
    // 5.3 ms of the DPO step.  The model below therefore halves the 256 x 256 tile's efficiency there.  AA_GEMM_SMALLK=0: the round-3 choice (A/B).
    static int smallk = -1;
    if (smallk < 0) { const char* e = getenv("AA_GEMM_SMALLK"); smallk = e ? atoi(e) : 1; }
    int best = 1; float best_t = 1e30f;
    for (int c = 0; c < 4; ++c) {
        const long tiles = (long)aa_cdiv(M, cfgs[c].bm) * aa_cdiv(N, cfgs[c].bn);
        const long rounds = (tiles + cfgs[c].slots - 1) / cfgs[c].slots;
        // time ~ rounds * (tile flops / (eff * per-slot rate)); per-slot rate halves when 2 tiles share a CU
        const float eff = (c == 0 && smallk && K < 2048) ? cfgs[c].eff * 0.45f : cfgs[c].eff;
        const float per = (float)cfgs[c].bm * cfgs[c].bn / eff * (cfgs[c].slots / 256.f);
        const float t = rounds * per;
        if (t < best_t) { best_t = t; best = c; }
    }
    // the 256x256 tile runs on the one-wave-per-SIMD kernel (gemm4.hip): +1..17 % on every hot 7B shape and layout
    // (profiles/r02_gemm_lab_g4.json); the 8-wave kernel stays reachable as tile 0
    if (best == 0 && g4) return 5;
    return best;
}

extern "C" int aa_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda,
                            long ldb, long ldc, const void* bias, const void* residual, long ldr,
                            int act, int flags, void* stream) {
    AA_REQUIRE(M > 0 && N > 0 && K > 0, "aa_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    AA_REQUIRE(K % BK == 0, "aa_gemm_bf16: K=%d must be a multiple of %d (zero-pad the operands)", K, BK);
    AA_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "aa_gemm_bf16: N=%d and ldc=%ld must be multiples of 4", N, ldc);
    AA_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "aa_gemm_bf16: lda=%ld / ldb=%ld must be multiples of 8", lda, ldb);
    AA_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
               "aa_gemm_bf16: operands must be 16-byte aligned");
    const bool a_t = flags & AA_GEMM_A_T, b_n = flags & AA_GEMM_B_N;
    if (a_t) AA_REQUIRE(M % 8 == 0, "aa_gemm_bf16: transposed A needs M %% 8 == 0 (got %d)", M);
    if (b_n) AA_REQUIRE(N % 8 == 0, "aa_gemm_bf16: N-contiguous B needs N %% 8 == 0 (got %d)", N);
    if (residual) AA_REQUIRE(ldr % 4 == 0, "aa_gemm_bf16: ldr=%ld must be a multiple of 4", ldr);
    GemmParams p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = (const bf16_t*)bias; p.residual = (const bf16_t*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.act = act; p.flags = flags;
    int tile = pick_tile(M, N, K);
    hipStream_t st = (hipStream_t)stream;
    if (tile == 5 && !aa_gemm4_supports(K)) tile = 0;      // K not a multiple of 128: the 8-wave kernel of the same tile
    if (tile == 5) {   // one-wave-per-SIMD 256x256 tile with accumulator-file MFMAs (gemm4.hip)
        p.tiles_m = aa_cdiv(p.M, 256);
        p.tiles_n = aa_cdiv(p.N, 256);
        p.gm = pick_group(a_t, b_n, p.tiles_n, p.K);
        return aa_gemm4_dispatch(p, a_t, b_n, st);
    }
    if (!a_t && !b_n) return launch_layout<false, false>(p, tile, st);
    if (!a_t && b_n) return launch_layout<false, true>(p, tile, st);
    if (a_t && b_n) return launch_layout<true, true>(p, tile, st);
    aa_set_error("aa_gemm_bf16: layout A^T with K-contiguous B is not built (unused by the hot path)");
    return AA_ERR_ARG;
}

// ---- grouped (mixture-of-experts) GEMM: the 128x256 tile, so expert segments are aligned to AA_MOE_ALIGN = 128 rows
// BM = 128 for the row-grouped modes (the expert segments are 128-row aligned); the per-expert weight gradients (mode 2) tile the OUTPUT [N_out, K_in] and may
// take the 256 x 256 tile (half the LDS traffic and half the epilogues per flop of their short, segment-long contraction): AA_MOE_DW_TILE, A/B in DESIGN section 5
template <bool A_T, bool B_N, int GRP, int BM = 128, int ILV = 0>
static int launch_grouped(GemmParams& p, int E, hipStream_t st) {
    constexpr int BN = 256, WM = 2, WN = 4;
    p.tiles_m = aa_cdiv(p.M, BM);
    p.tiles_n = aa_cdiv(p.N, BN);
    p.gm = 4;      // grouped (MoE) launches: rows are expert segments, keep the row-grouped order
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_kernel<BM, BN, WM, WN, A_T, B_N, true, ILV, GRP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { aa_set_error("aa_gemm_grouped_bf16: cannot reserve %d B LDS: %s", lds, hipGetErrorString(e)); return AA_ERR_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, GRP == 2 ? E : 1), dim3(WM * WN * 64), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_grouped_bf16");
    return AA_OK;
}

// ---- split-K for few-row launches.  out = epilogue(sum_s partial_s), the chunks summed in order (deterministic; fp32 association differs from the one-launch
// kernel's single k-ordered chain, so results agree to fp32 rounding of the accumulator, not bit for bit).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long strideS, const GemmParams p) {
    const long n4 = p.N >> 2, total = (long)p.M * n4;
    const bool out_f32 = p.flags & AA_GEMM_OUT_F32, accum = p.flags & AA_GEMM_ACCUM;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / n4;
        const int n = (int)(i % n4) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + m * p.N + n);
        for (int s2 = 1; s2 < S; ++s2) a += *reinterpret_cast<const f32x4*>(ws + (long)s2 * strideS + m * p.N + n);
        float v[4] = {a[0], a[1], a[2], a[3]};
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
            const u16x4 r = *reinterpret_cast<const u16x4*>(p.residual + m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) + bf2f(r[e]);
        }
        if (out_f32) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if (accum) o += *reinterpret_cast<const f32x4*>(c);
            *reinterpret_cast<f32x4*>(c) = o;
        } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n;
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

template <bool A_T, bool B_N>
static int launch_splitk(GemmParams& q, int S, hipStream_t st) {
    constexpr int BM = 256, BN = 128, WM = 4, WN = 2;
    q.tiles_m = aa_cdiv(q.M, BM);
    q.tiles_n = aa_cdiv(q.N, BN);
    q.gm = pick_group(A_T, B_N, q.tiles_n, q.K);
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_kernel<BM, BN, WM, WN, A_T, B_N, true, 0, 3>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { aa_set_error("aa_gemm_splitk_bf16: cannot reserve %d B LDS: %s", lds, hipGetErrorString(e)); return AA_ERR_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.tiles_m * q.tiles_n, S), dim3(WM * WN * 64), lds, st, q);
    AA_CHECK_LAUNCH("aa_gemm_splitk_bf16");
    return AA_OK;
}

extern "C" int aa_gemm_splitk_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, const void* bias,
                                   const void* residual, long ldr, int act, int flags, float* ws, int S, void* stream) {
    AA_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0 && ldc % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0,
               "aa_gemm_splitk_bf16: bad shape M=%d N=%d K=%d lda=%ld ldb=%ld ldc=%ld", M, N, K, lda, ldb, ldc);
    AA_REQUIRE(S >= 2 && S <= 64 && ws != nullptr, "aa_gemm_splitk_bf16: %d chunks need an fp32 workspace of chunks x M x N", S);
    AA_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)ws & 15) == 0, "aa_gemm_splitk_bf16: operands must be 16-byte aligned");
    AA_REQUIRE(residual == nullptr || (ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0), "aa_gemm_splitk_bf16: residual rows must be 8-byte aligned (ldr = %ld)", ldr);
    AA_REQUIRE(((uintptr_t)bias & 7) == 0, "aa_gemm_splitk_bf16: bias must be 8-byte aligned");
    const bool a_t = flags & AA_GEMM_A_T, b_n = flags & AA_GEMM_B_N;
    AA_REQUIRE(!a_t || b_n, "aa_gemm_splitk_bf16: layout A^T with K-contiguous B is not built");
    if (a_t) AA_REQUIRE(M % 8 == 0, "aa_gemm_splitk_bf16: transposed A needs M %% 8 == 0 (got %d)", M);
    int kc = aa_cdiv(aa_cdiv(K, S), BK) * BK;            // chunk length: a multiple of the k-tile
    S = aa_cdiv(K, kc);
    hipStream_t st = (hipStream_t)stream;
    GemmParams q{};
    q.A = (const bf16_t*)A; q.B = (const bf16_t*)B; q.C = ws;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = N;
    q.act = AA_ACT_NONE; q.flags = (flags & (AA_GEMM_A_T | AA_GEMM_B_N)) | AA_GEMM_OUT_F32;
    q.grp_strideB = kc; q.grp_strideC = (long)M * N;
    int rc;
    if (!a_t && !b_n) rc = launch_splitk<false, false>(q, S, st);
    else if (!a_t && b_n) rc = launch_splitk<false, true>(q, S, st);
    else rc = launch_splitk<true, true>(q, S, st);
    if (rc != AA_OK) return rc;
    GemmParams p{};
    p.C = C; p.bias = (const bf16_t*)bias; p.residual = (const bf16_t*)residual;
    p.M = M; p.N = N; p.K = K; p.ldc = ldc; p.ldr = ldr; p.act = act; p.flags = flags;
    const long total = (long)M * (N >> 2);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)ws, S, (long)M * N, p);
    AA_CHECK_LAUNCH("aa_gemm_splitk_bf16");
    return AA_OK;
}

extern "C" int aa_gemm_grouped_bf16(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                                    int flags, int mode, const int* tile_expert, const int* seg_off, long stride, int E,
                                    void* stream) {
    AA_REQUIRE(mode >= 1 && mode <= 3, "aa_gemm_grouped_bf16: mode %d (1 = rows grouped, 2 = contraction grouped, 3 = rows grouped in 256-aligned segments)", mode);
    AA_REQUIRE(M > 0 && N > 0 && E > 0 && N % 8 == 0 && ldc % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0,
               "aa_gemm_grouped_bf16: bad shape M=%d N=%d E=%d", M, N, E);
    const bool a_t = flags & AA_GEMM_A_T, b_n = flags & AA_GEMM_B_N;
    GemmParams p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.flags = flags; p.act = AA_ACT_NONE;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1 || mode == 3) {
        AA_REQUIRE(tile_expert != nullptr && !a_t && K > 0 && K % BK == 0 && M % 128 == 0,
                   "aa_gemm_grouped_bf16: mode 1 needs tile_expert, A row-major, K %% 64 == 0, M %% 128 == 0 (K=%d M=%d)", K, M);
        p.grp_tile_expert = tile_expert; p.grp_strideB = stride;
        // segments aligned to 256 rows (ops.MOE_ALIGN) run on gemm4's 256 x 256 one-wave-per-SIMD tile when the shape fits it: 742 / 660 / 683 TFLOP/s on the routed rows
        // of the Qwen3-30B-A3B layer (gate_up, down, down dX) against 563 / 463 / 502 for the 128 x 256 8-wave kernel below (profiles/r04_moe_gemm4_grouped.txt).
        // AA_MOE_GEMM4=0 keeps the 8-wave kernel (same-box A/B).  Mode 3 is the caller's promise that no 256-row tile straddles two experts.
        static int g4 = -1;
        if (g4 < 0) { const char* e = getenv("AA_MOE_GEMM4"); g4 = e ? atoi(e) : 1; }
        if (g4 && mode == 3 && M % 256 == 0) {
            const int rc = aa_gemm4_grouped(p, b_n, st);
            if (rc != 1) return rc;
        }
        return b_n ? launch_grouped<false, true, 1>(p, E, st) : launch_grouped<false, false, 1>(p, E, st);
    }
    AA_REQUIRE(seg_off != nullptr && a_t && b_n && M % 8 == 0, "aa_gemm_grouped_bf16: mode 2 needs seg_off and the TN layout");
    p.grp_off = seg_off; p.grp_strideC = stride;
    static int dw_tile = -1;
    if (dw_tile < 0) { const char* e = getenv("AA_MOE_DW_TILE"); dw_tile = e ? atoi(e) : 256; }      // same-box A/B, 12-layer Qwen3-30B-A3B-geometry step: 157.1 / 157.4 ms vs 159.9 / 160.7 ms at 128
    if (dw_tile == 256) return launch_grouped<true, true, 2, 256, 1>(p, E, st);
    return launch_grouped<true, true, 2>(p, E, st);
}

// ---- GEMMs with the element-wise neighbour of the HF graph folded into the epilogue (gemm4.hip); each entry point runs the fused
// kernel when the shape qualifies and the unfused pair of kernels otherwise, with identical rounding points (bit-identical results)
extern "C" int aa_rope_inplace(void* buf, long ld, int col0, int nheads, int hd, const int* pos, const void* cos_t, const void* sin_t,
                               long rows, int inverse, int head_stride, int precise, void* stream);
extern "C" int aa_swiglu_fwd(const void* gate_up, void* out, long M, int F, void* stream);
extern "C" int aa_swiglu_bwd(const void* gate_up, const void* dact, void* dgate_up, long M, int F, void* stream);

// aa_ctx::fuse: AA_GEMM_FUSE=0 = always the unfused kernels (A/B runs, parity tests)
static bool fuse_enabled() {
    if (aa_ctx_cur()->fuse < 0) { const char* e = getenv("AA_GEMM_FUSE"); aa_ctx_cur()->fuse = e ? atoi(e) : 1; }
    return aa_ctx_cur()->fuse != 0 && aa_ctx_cur()->force_tile != 0;
}
extern "C" int aa_gemm_set_fuse(int on) { aa_ctx_cur()->fuse = on ? 1 : 0; return AA_OK; }

// hf:models/llama/modeling_llama.py:228-246 q/k/v projections + apply_rotary_pos_emb (:130-160): C[M, N] = A W^T with the rotary
// embedding applied to the heads in columns [0, rope_cols) (q and k of the fused [q|k|v] weight); pos[M], cos / sin [max_pos, hd/2] bf16
extern "C" int aa_gemm_qkv_rope_bf16(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                                     const int* pos, const void* cos_t, const void* sin_t, int rope_cols, int hd, void* stream) {
    AA_REQUIRE(hd > 0 && rope_cols % hd == 0 && rope_cols <= N, "aa_gemm_qkv_rope_bf16: rope_cols %d must be whole heads of %d within N=%d", rope_cols, hd, N);
    if (fuse_enabled() && hd == 128) {
        GemmParams p{};
        p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = C;
        p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc;
        p.gm = pick_group(false, false, aa_cdiv(N, 256), K);
        p.fuse = AA_FUSE_ROPE; p.rope_pos = pos; p.rope_cos = (const bf16_t*)cos_t; p.rope_sin = (const bf16_t*)sin_t; p.rope_cols = rope_cols;
        if ((lda & 7) == 0 && (ldw & 7) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0) {
            const int rc = aa_gemm4_fused(p, (hipStream_t)stream);
            if (rc != 1) return rc;
        }
    }
    const int rc = aa_gemm_bf16(A, W, C, M, N, K, lda, ldw, ldc, nullptr, nullptr, 0, AA_ACT_NONE, 0, stream);
    if (rc != AA_OK || rope_cols == 0) return rc;
    return aa_rope_inplace(C, ldc, 0, rope_cols / hd, hd, pos, cos_t, sin_t, M, 0, 0, 0, stream);
}

// hf:models/llama/modeling_llama.py:163-176 LlamaMLP: GU[M, 2F] = A [Wg; Wu]^T (saved for the backward) and ACT[M, F] = silu(gate) * up
extern "C" int aa_gemm_glu_fwd_bf16(const void* A, const void* Wgu, void* GU, void* ACT, int M, int F, int K, long lda, long ldw,
                                    long ldgu, long ldact, void* stream) {
    AA_REQUIRE(F > 0 && (F & 7) == 0, "aa_gemm_glu_fwd_bf16: ffn %d must be a multiple of 8", F);
    if (fuse_enabled()) {
        GemmParams p{};
        p.A = (const bf16_t*)A; p.B = (const bf16_t*)Wgu; p.C = GU;
        p.M = M; p.N = 2 * F; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldgu;
        p.gm = pick_group(false, false, aa_cdiv(2 * F, 256), K);
        p.fuse = AA_FUSE_GLU_FWD; p.aux = ACT; p.ldaux = ldact; p.glu_f = F;
        if ((lda & 7) == 0 && (ldw & 7) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)Wgu & 15) == 0 && ((uintptr_t)GU & 15) == 0) {
            const int rc = aa_gemm4_fused(p, (hipStream_t)stream);
            if (rc != 1) return rc;
        }
    }
    const int rc = aa_gemm_bf16(A, Wgu, GU, M, 2 * F, K, lda, ldw, ldgu, nullptr, nullptr, 0, AA_ACT_NONE, 0, stream);
    if (rc != AA_OK) return rc;
    AA_REQUIRE(ldgu == 2L * F && ldact == F, "aa_gemm_glu_fwd_bf16: the unfused path needs dense [M, 2F] / [M, F] buffers");
    return aa_swiglu_fwd(GU, ACT, M, F, stream);
}

// backward of the same block: dGU[M, 2F] = swiglu'(GU) (.) (dY[M, h] W_down[h, F]); the fused kernel never stores d_act, the unfused
// path needs `dact_ws` [M, F].
//
// Which of the two runs is a per-shape PLAN.  The fused epilogue moves 4 x M x F x 2 B from inside a one-workgroup-per-CU GEMM tile
// (its loads are dependent round trips the MFMA work of no other tile can hide), the unfused pair streams d_act once more but from a
// full-chip element-wise kernel.  On most MI355X boxes the two are within 5 % of each other; on boxes with a longer memory round trip the
// fused kernel was measured at 2.2-2.3 ms against 1.03 + 0.36 ms for the pair (BENCH_r02, profiles/r02_glu_bwd_latency.txt).  A box cannot
// be told apart from inside a kernel, so the caller measures: aa_gemm_glu_bwd_probe() times both variants on the caller's own buffers
// (both produce bit-identical dGU, tests/test_gemm_gpu.py) and records the faster one for that (M, F, K); aa_gemm_glu_bwd_bf16 follows
// the record, and fuses when there is none.
static bool glu_bwd_fusable(const void* dY, const void* Wdown, int M, int F, int K, long ldy, long ldw, long ldgu, long lddgu,
                            const void* GU, const void* dGU) {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return fuse_enabled() && M % 256 == 0 && F % 256 == 0 && aa_gemm4_supports(K) && (ldy & 7) == 0 && (ldw & 7) == 0 && (ldgu & 7) == 0 &&
           (lddgu & 7) == 0 && (F & 7) == 0 && al16(dY) && al16(Wdown) && al16(GU) && al16(dGU);
}

namespace {
// Records are keyed on (F, K, bucket of M): the winner is a property of the box's memory round trip and of the layer geometry, not of the exact row
// count, and variable sequence lengths make M take many values -- one record per power-of-two bucket of M / 256 keeps re-probes (8 GEMM launches +
// stream syncs each) bounded; 64 slots, least-recently-used eviction.
// (the records live in the library context: aa_ctx::glu_plans, csrc/aa_ctx.h)
int glu_mode() {
    aa_ctx* c = aa_ctx_cur();
    if (!c->glu_mode_read) { const char* e = getenv("AA_GLU_BWD"); if (e) c->glu_mode = atoi(e); c->glu_mode_read = true; }
    return c->glu_mode;
}
int glu_m_bucket(int M) {            // 0: M <= 256, 1: <= 512, 2: <= 1024, ...
    int b = 0;
    for (long cap = 256; cap < M; cap <<= 1) ++b;
    return b;
}
int glu_plan_lookup(int M, int F, int K) {
    aa_ctx* c = aa_ctx_cur();
    const int mb = glu_m_bucket(M);
    for (int i = 0; i < c->glu_nplans; ++i)
        if (c->glu_plans[i].mb == mb && c->glu_plans[i].F == F && c->glu_plans[i].K == K) { c->glu_plans[i].stamp = ++c->glu_clock; return c->glu_plans[i].fused; }
    return -1;
}
void glu_plan_store(int M, int F, int K, int fused) {
    aa_ctx* c = aa_ctx_cur();
    const int mb = glu_m_bucket(M);
    int slot = -1;
    for (int i = 0; i < c->glu_nplans; ++i)
        if (c->glu_plans[i].mb == mb && c->glu_plans[i].F == F && c->glu_plans[i].K == K) slot = i;
    if (slot < 0) {
        if (c->glu_nplans < AA_GLU_PLANS) slot = c->glu_nplans++;
        else { slot = 0; for (int i = 1; i < AA_GLU_PLANS; ++i) if (c->glu_plans[i].stamp < c->glu_plans[slot].stamp) slot = i; }
    }
    c->glu_plans[slot] = AaGluPlan{mb, F, K, fused, ++c->glu_clock};
}
int glu_bwd_run(bool fused, const void* dY, const void* Wdown, const void* GU, void* dGU, void* dact_ws, int M, int F, int K, long ldy,
                long ldw, long ldgu, long lddgu, void* stream) {
    if (fused) {
        GemmParams p{};
        p.A = (const bf16_t*)dY; p.B = (const bf16_t*)Wdown; p.C = nullptr;
        p.M = M; p.N = F; p.K = K; p.lda = ldy; p.ldb = ldw; p.ldc = 8; p.flags = AA_GEMM_B_N;
        p.gm = pick_group(false, true, aa_cdiv(F, 256), K);
        p.fuse = AA_FUSE_GLU_BWD; p.aux = dGU; p.ldaux = lddgu; p.aux_in = (const bf16_t*)GU; p.ldaux_in = ldgu; p.glu_f = F;
        const int rc = aa_gemm4_fused(p, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    AA_REQUIRE(dact_ws != nullptr && ldgu == 2L * F && lddgu == 2L * F, "aa_gemm_glu_bwd_bf16: the unfused path needs a d_act workspace and dense [M, 2F] buffers");
    const int rc = aa_gemm_bf16(dY, Wdown, dact_ws, M, F, K, ldy, ldw, F, nullptr, nullptr, 0, AA_ACT_NONE, AA_GEMM_B_N, stream);
    if (rc != AA_OK) return rc;
    return aa_swiglu_bwd(GU, dact_ws, dGU, M, F, stream);
}
}  // namespace

// *plan = 1: the fused kernel will run for these arguments, 0 = the unfused pair (the caller must pass a d_act workspace), 2 = fusable but no
// record yet for (M, F, K): fused unless aa_gemm_glu_bwd_probe decides otherwise.  The single source of truth for "is dact_ws needed".
static int glu_bwd_plan(const void* dY, const void* Wdown, const void* GU, const void* dGU, int M, int F, int K, long ldy, long ldw,
                        long ldgu, long lddgu) {
    if (!glu_bwd_fusable(dY, Wdown, M, F, K, ldy, ldw, ldgu, lddgu, GU, dGU) || glu_mode() == 0) return 0;
    if (glu_mode() == 1) return 1;
    const int rec = glu_plan_lookup(M, F, K);
    return rec < 0 ? 2 : rec;
}
extern "C" int aa_gemm_glu_bwd_plan(const void* dY, const void* Wdown, const void* GU, const void* dGU, int M, int F, int K, long ldy,
                                    long ldw, long ldgu, long lddgu, int* plan) {
    AA_REQUIRE(plan != nullptr, "aa_gemm_glu_bwd_plan: plan is null");
    *plan = glu_bwd_plan(dY, Wdown, GU, dGU, M, F, K, ldy, ldw, ldgu, lddgu);
    return AA_OK;
}

extern "C" int aa_gemm_glu_bwd_set_mode(int mode) { aa_ctx_cur()->glu_mode = mode; aa_ctx_cur()->glu_mode_read = true; return AA_OK; }
extern "C" int aa_gemm_glu_bwd_forget(void) { aa_ctx_cur()->glu_nplans = 0; return AA_OK; }

// Times `reps` launches of each variant on the caller's buffers (after one untimed launch each) with HIP events on `stream`, records the
// faster one for (M, F, K) and leaves dGU computed.  Synchronises the stream: call it once per shape, outside any timed region.
extern "C" int aa_gemm_glu_bwd_probe(const void* dY, const void* Wdown, const void* GU, void* dGU, void* dact_ws, int M, int F, int K,
                                     long ldy, long ldw, long ldgu, long lddgu, int reps, float* ms_fused, float* ms_unfused, void* stream) {
    AA_REQUIRE(dact_ws != nullptr && reps > 0, "aa_gemm_glu_bwd_probe: needs a d_act workspace and reps > 0");
    if (!glu_bwd_fusable(dY, Wdown, M, F, K, ldy, ldw, ldgu, lddgu, GU, dGU)) {
        if (ms_fused) *ms_fused = -1.f;
        if (ms_unfused) *ms_unfused = -1.f;
        return glu_bwd_run(false, dY, Wdown, GU, dGU, dact_ws, M, F, K, ldy, ldw, ldgu, lddgu, stream);
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { aa_set_error("aa_gemm_glu_bwd_probe: hipEventCreate failed"); return AA_ERR_LAUNCH; }
    float ms[2] = {0.f, 0.f};
    int rc = AA_OK;
    for (int variant = 0; variant < 2 && rc == AA_OK; ++variant) {
        const bool fused = variant == 0;
        rc = glu_bwd_run(fused, dY, Wdown, GU, dGU, dact_ws, M, F, K, ldy, ldw, ldgu, lddgu, stream);
        (void)hipEventRecord(e0, (hipStream_t)stream);
        for (int r = 0; r < reps && rc == AA_OK; ++r) rc = glu_bwd_run(fused, dY, Wdown, GU, dGU, dact_ws, M, F, K, ldy, ldw, ldgu, lddgu, stream);
        (void)hipEventRecord(e1, (hipStream_t)stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms[variant], e0, e1) != hipSuccess) {
            aa_set_error("aa_gemm_glu_bwd_probe: event timing failed");
            rc = AA_ERR_LAUNCH;
        }
        ms[variant] /= (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != AA_OK) return rc;
    if (ms_fused) *ms_fused = ms[0];
    if (ms_unfused) *ms_unfused = ms[1];
    glu_plan_store(M, F, K, ms[0] <= ms[1] ? 1 : 0);
    return AA_OK;
}

extern "C" int aa_gemm_glu_bwd_bf16(const void* dY, const void* Wdown, const void* GU, void* dGU, void* dact_ws, int M, int F, int K,
                                    long ldy, long ldw, long ldgu, long lddgu, void* stream) {
    AA_REQUIRE(F > 0 && (F & 7) == 0, "aa_gemm_glu_bwd_bf16: ffn %d must be a multiple of 8", F);
    const bool fused = glu_bwd_plan(dY, Wdown, GU, dGU, M, F, K, ldy, ldw, ldgu, lddgu) != 0;
    return glu_bwd_run(fused, dY, Wdown, GU, dGU, dact_ws, M, F, K, ldy, ldw, ldgu, lddgu, stream);
}

// test hook: force a tile config (-1 = heuristic)
extern "C" int aa_gemm_set_tile(int tile) { aa_ctx_cur()->force_tile = tile; return AA_OK; }
// test/bench hook: 1 = software-pipelined K loop (default), 0 = simple one-barrier schedule
extern "C" int aa_gemm_set_group(int gm) { aa_ctx_cur()->gm = gm < 0 ? 0 : gm; return AA_OK; }   // 0 = heuristic; +256 = group tile columns instead of rows
