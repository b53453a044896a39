// bf16 MFMA GEMM, one-wave-per-SIMD variant of gemm.hip for the big (>= one round of 256 tiles) shapes of the DPO step.
//
// Same contract, LDS images, swizzles and epilogue as gemm_kernel (gemm.hip); what changes is the decomposition:
//   * 256 x 256 x 64 tile, 4 waves (2 x 2), each wave owns 128 x 128 of the output = 64 accumulator tiles of
//     v_mfma_f32_16x16x32_bf16 = 256 registers.  A 128 x 128 wave tile reads (128 + 128) rows of fragments per 32-deep
//     k-step for 64 MFMAs: 1/3 fewer LDS bytes and LDS instructions per flop than the 128 x 64 wave tile of the 8-wave
//     kernel, and half the waves to keep in step at the K-tile barrier.
//   * the accumulators live in the ACCUMULATOR register file (a0..a255) and never move: the MFMAs are inline asm with
//     "+a" operands.  (Compiled from the builtin, the same tile makes hipcc shuttle half of the accumulators between
//     the two files -- two v_accvgpr moves per MFMA, profiles/r02_gemm_lab_w4.txt -- because 256 accumulators + the
//     fragments exceed what its allocator places cleanly.)  The arch VGPRs hold two fragment sets (128), the DMA lane
//     offsets and the LDS read offsets.
//   * operand DMA addresses are a wave-uniform base (SGPR pair, advanced by one K-tile per stage) + a per-lane 32-bit
//     byte offset that never changes: no 64-bit VALU pointer arithmetic in the loop.
//   * schedule = the two-fragment-set pipeline of gemm.hip with the LDS reads of the next set placed by hand between
//     the MFMAs of the current one (2 reads per 8 MFMAs, pinned with sched_barrier), one barrier per K-tile.
// hipBLASLt's own kernel for these shapes has the same decomposition (MT256x256x64, 256 threads, 1 wave / SIMD) and keeps the
// matrix pipe 83 % busy where the 8-wave kernel reaches 62 % (profiles/r02_gemm_vs_hipblaslt_pmc.txt); it is a yardstick only.
#include "aa_common.h"

#include <type_traits>

#include "gemm_params.h"

namespace {

constexpr int BM = 256, BN = 256, NW = 4, WN = 2, TM = 128, TNW = 128, FM = 8, FN = 8;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
constexpr int A_IT = (A_BYTES / 1024) / NW, B_IT = (B_BYTES / 1024) / NW;

__device__ __forceinline__ int tr_swz4(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// D = B-fragment x A-fragment (operands swapped like gemm.hip: a lane owns 4 consecutive output columns); the accumulator
// is read and written in place in the accumulator file
#define AA_MFMA_ACC(ACC, BF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(BF), "v"(AF))


// One LDS-DMA piece: K < 8 = A chunk K, else B chunk K - 8.  M0 (the LDS destination) is a running pointer: G4_M0_SET points it at
// this wave's first A chunk of the target buffer well ahead of piece 0, and every piece advances it for the NEXT one right after its
// request (4 KB to the wave's next chunk; from the last A chunk to the first B chunk; nothing after the last piece) -- the one
// wait state the M0 write needs before a DMA reads it is then covered by the MFMA in between, no s_nop in the stream.  saddr form:
// wave-uniform 64-bit base + per-lane 32-bit byte offset.
#define G4_M0_SET(LDSW) asm volatile("s_mov_b32 m0, %0" ::"s"(LDSW) : "memory")
template <int K>
__device__ __forceinline__ void G4_DMA_PIECE(const unsigned (&offA)[A_IT], const unsigned (&offB)[B_IT], const char* srcA, const char* srcB) {
    if constexpr (K < A_IT - 1) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(offA[K]), "s"(srcA), "i"(NW * 1024) : "memory");
    } else if constexpr (K == A_IT - 1) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(offA[K]), "s"(srcA), "i"(65536 - (A_IT - 1) * NW * 1024) : "memory");
    } else if constexpr (K < A_IT + B_IT - 1) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(offB[K - A_IT]), "s"(srcB), "i"(NW * 1024) : "memory");
    } else {
        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(offB[K - A_IT]), "s"(srcB) : "memory");
    }
}

// PLAIN: bf16 C = A * B with no bias / activation / residual / accumulate and M, N multiples of the tile (every forward, dX and
// dW GEMM of the 7B decoder stack): the epilogue is straight-line 16-byte stores, and -- its own instantiation -- shares no
// registers with the general epilogue, whose 256-value fan-out would otherwise make the compiler spill accumulators
template <bool A_T, bool B_N, bool PLAIN>
__global__ __launch_bounds__(NW * 64, 1)
void gemm4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware bijective remap, then grouped tile order (identical to gemm_kernel)
    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int GM = p.gm & 0xff;
    int tm, tn;
    if (!(p.gm & 0x100)) {
        const int per_group = GM * p.tiles_n;
        const int group = wg / per_group;
        const int first_m = group * GM;
        const int gsz = min(p.tiles_m - first_m, GM);
        tm = first_m + (wg % per_group) % gsz;
        tn = (wg % per_group) / gsz;
    } else {
        const int per_group = GM * p.tiles_m;
        const int group = wg / per_group;
        const int first_n = group * GM;
        const int gsz = min(p.tiles_n - first_n, GM);
        tn = first_n + (wg % per_group) % gsz;
        tm = (wg % per_group) / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA sources: uniform base (advanced per K-tile) + per-lane byte offset inside the tile's row / column block
    const char* baseA;
    const char* baseB;
    unsigned offA[A_IT], offB[B_IT];
    long stepA, stepB;
    if constexpr (!A_T) {
        baseA = reinterpret_cast<const char*>(p.A + (long)m0 * p.lda);
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            const int gr = min(m0 + r, p.M - 1) - m0;
            offA[j] = (unsigned)((gr * p.lda + ks * 8) * 2);
        }
        stepA = BK * 2;
    } else {
        constexpr int RPI = 1024 / (BM * 2), SPR = BM * 2 / 16;
        baseA = reinterpret_cast<const char*>(p.A + m0);
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz4(kr);
            const int col = min(m0 + unit * 16 + (s & 1) * 8, p.M - 8) - m0;
            offA[j] = (unsigned)((kr * p.lda + col) * 2);
        }
        stepA = (long)BK * p.lda * 2;
    }
    if constexpr (!B_N) {
        baseB = reinterpret_cast<const char*>(p.B + (long)n0 * p.ldb);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int r = c * 8 + (lane >> 3);
            const int ks = (lane & 7) ^ ((r >> 1) & 7);
            const int gr = min(n0 + r, p.N - 1) - n0;
            offB[j] = (unsigned)((gr * p.ldb + ks * 8) * 2);
        }
        stepB = BK * 2;
    } else {
        constexpr int RPI = 1024 / (BN * 2), SPR = BN * 2 / 16;
        baseB = reinterpret_cast<const char*>(p.B + n0);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz4(kr);
            const int col = min(n0 + unit * 16 + (s & 1) * 8, p.N - 8) - n0;
            offB[j] = (unsigned)((kr * p.ldb + col) * 2);
        }
        stepB = (long)BK * p.ldb * 2;
    }

    // ---- LDS: A tile of buffer b at b * 32 KB, B tile at 64 KB + b * 32 KB (every read offset then fits the 16-bit ds immediate)
    const int lds0 = (int)(uintptr_t)smem;                                  // wave-uniform LDS byte address of the dynamic segment
    const int ldsw = lds0 + wave * 1024;                                    // this wave's first DMA chunk
    // DMA piece k (0..7 = A chunks, 8..15 = B chunks of this wave) of the K-tile whose operand pointers are (srcA, srcB) into
    // buffer offset `cb` (0 / 32768).  asm: saddr form (uniform 64-bit base + per-lane 32-bit offset), M0 = LDS destination.
#define G4_DMA(K) G4_DMA_PIECE<K>(offA, offB, srcA, srcB)

    // ---- per-lane LDS read addresses (same swizzled images as gemm.hip)
    const int l15 = lane & 15, g = lane >> 4;
    int vak[2], vbk[2], ta[FM], tb[FN];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int offK = l15 * 128 + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) << 4);
        vak[kk] = lds0 + offK + wm * TM * 128;
        vbk[kk] = lds0 + 65536 + offK + wn * TNW * 128;
    }
    {
        const int swz = ((l15 >> 2) & 3) | ((g & 1) << 2);                  // tr_swz4(k-row): independent of k-step and half
        const int lanepart = (g * 8 + (l15 >> 2)) * (BM * 2) + (l15 & 3) * 8;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            ta[i] = lds0 + lanepart + (wm * 8 + (i ^ swz)) * 32;
            tb[i] = lds0 + 65536 + lanepart + (wn * 8 + (i ^ swz)) * 32;
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment sets: whole 128-bit fragments for K-contiguous images, two 64-bit halves for transposed reads
    bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
    bf16x4 a0h[FM][2], b0h[FN][2], a1h[FM][2], b1h[FN][2];

#define G4_MFMA(ACC, BF, AF) AA_MFMA_ACC(ACC, BF, AF)
#define G4_RDK(DST, VADDR, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(VADDR), "i"(IMM))
#define G4_RDT(DST, VADDR, IMM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(VADDR), "i"(IMM))
#define G4_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N))
#define G4_PIN __builtin_amdgcn_sched_barrier(0)
#define G4_SYNC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_mov_b32 m0, %0" ::"s"(ldsw_c) : "memory")
#define G4_JOIN(LO, HI) __builtin_shufflevector(LO, HI, 0, 1, 2, 3, 4, 5, 6, 7)

    const int nt = p.K / BK;
    // ---- prologue: K-tile 0 -> buffer 0, wait, first fragment set, K-tile 1 -> buffer 1
    {
        const char* srcA = baseA;
        const char* srcB = baseB;
        G4_M0_SET(ldsw);
        asm volatile("s_nop 0");
        G4_DMA(0); G4_DMA(1); G4_DMA(2); G4_DMA(3); G4_DMA(4); G4_DMA(5); G4_DMA(6); G4_DMA(7);
        G4_DMA(8); G4_DMA(9); G4_DMA(10); G4_DMA(11); G4_DMA(12); G4_DMA(13); G4_DMA(14); G4_DMA(15);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    {
        // same order as the in-loop reads (b0..b7, a0..a7): the phase-A wait counts rely on it
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (!B_N) { G4_RDK(b0[j], vbk[0], j * 2048); }
            else { G4_RDT(b0h[j][0], tb[j], 0); G4_RDT(b0h[j][1], tb[j], 2048); }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (!A_T) { G4_RDK(a0[i], vak[0], i * 2048); }
            else { G4_RDT(a0h[i][0], ta[i], 0); G4_RDT(a0h[i][1], ta[i], 2048); }
        }
        const int t1 = min(1, nt - 1);
        const char* srcA = baseA + (long)t1 * stepA;
        const char* srcB = baseB + (long)t1 * stepB;
        G4_M0_SET(ldsw + 32768);
        asm volatile("s_nop 0");
        G4_DMA(0); G4_DMA(1); G4_DMA(2); G4_DMA(3); G4_DMA(4); G4_DMA(5); G4_DMA(6); G4_DMA(7);
        G4_DMA(8); G4_DMA(9); G4_DMA(10); G4_DMA(11); G4_DMA(12); G4_DMA(13); G4_DMA(14); G4_DMA(15);
    }
    // ---- main loop: one K-tile per iteration, branch-free (the K-tile index of the request is clamped: the last two
    // iterations re-request the last tile into a buffer nobody reads any more)
    for (int t = 0; t < nt; ++t) {
        const int cbc = (t & 1) * 32768, cbn = cbc ^ 32768;                  // buffer offsets: current / next K-tile
        const int t2 = min(t + 2, nt - 1);
        const char* srcA = baseA + (long)t2 * stepA;
        const char* srcB = baseB + (long)t2 * stepB;
        const int ldsw_c = ldsw + cbc;
        [[maybe_unused]] const int vak1_cur = vak[1] + cbc, vbk1_cur = vbk[1] + cbc, vak0_nxt = vak[0] + cbn, vbk0_nxt = vbk[0] + cbn;
        if constexpr (!A_T && !B_N) {
#define G4_FRAG_A(S, I) a##S[I]
#define G4_FRAG_B(S, J) b##S[J]
#include "gemm4_sched_nt.inc"
#undef G4_FRAG_A
#undef G4_FRAG_B
        } else if constexpr (!A_T && B_N) {
#define G4_FRAG_A(S, I) a##S[I]
#define G4_FRAG_B(S, J) G4_JOIN(b##S##h[J][0], b##S##h[J][1])
#include "gemm4_sched_nn.inc"
#undef G4_FRAG_A
#undef G4_FRAG_B
        } else {
#define G4_FRAG_A(S, I) G4_JOIN(a##S##h[I][0], a##S##h[I][1])
#define G4_FRAG_B(S, J) G4_JOIN(b##S##h[J][0], b##S##h[J][1])
#include "gemm4_sched_tn.inc"
#undef G4_FRAG_A
#undef G4_FRAG_B
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped re-requests of the last K-tile
    // the last MFMAs are still in the matrix pipe: the compiler does not know the asm statements wrote the accumulators late
    // (nothing may read the last row's accumulators above this statement: they are its operands)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[FM - 1][0]), "+a"(acc[FM - 1][1]), "+a"(acc[FM - 1][2]), "+a"(acc[FM - 1][3]), "+a"(acc[FM - 1][4]),
                   "+a"(acc[FM - 1][5]), "+a"(acc[FM - 1][6]), "+a"(acc[FM - 1][7])
                 :: "memory");

    // ---- epilogue: lane owns C[m][n..n+3], m = .. + l15, n = .. + g*4
    if constexpr (PLAIN) {
        // 16-byte stores: lanes g and g^1 (16 lanes apart) exchange halves with v_permlane16_swap, so that a lane ends up with 8
        // consecutive columns of ONE fragment (even g: fragment j, odd g: fragment j+1) -- half the store instructions of the
        // 4-columns-per-lane layout the MFMA leaves (the tile's store tail is issue-bound)
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (long)(m0 + wm * TM + l15) * p.ldc + n0 + wn * TNW + (g & 1) * 16 + (g >> 1) * 8;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                unsigned w[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        w[f][h] = (unsigned)f2bf(acc[i][j + f][2 * h]) | ((unsigned)f2bf(acc[i][j + f][2 * h + 1]) << 16);
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
                *reinterpret_cast<u32x4*>(crow + (long)i * 16 * p.ldc + j * 16) = u32x4{lo[0], hi[0], lo[1], hi[1]};
            }
            __builtin_amdgcn_sched_barrier(0);      // one accumulator row at a time: no 256-register fan-out of the reads
        }
    } else {
    // general path (gemm_store4: the rounding points of gemm.hip)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * TNW + j * 16 + g * 4;
            if (n >= p.N) continue;
            gemm_store4(p, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }
}

template <bool A_T, bool B_N, bool PLAIN>
int launch4(GemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * STAGE;
    auto kern = gemm4_kernel<A_T, B_N, PLAIN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_bf16 (4-wave tile): cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(NW * 64), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_bf16");
    return AA_OK;
}

template <bool PLAIN>
int launch4_layout(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if (!a_t && !b_n) return launch4<false, false, PLAIN>(p, st);
    if (!a_t && b_n) return launch4<false, true, PLAIN>(p, st);
    if (a_t && b_n) return launch4<true, true, PLAIN>(p, st);
    aa_set_error("aa_gemm_bf16: layout A^T with K-contiguous B is not built (unused by the hot path)");
    return AA_ERR_ARG;
}

}  // namespace

// p.tiles_m / tiles_n / gm are set by the caller (gemm.hip)
int aa_gemm4_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    const bool plain = p.flags == (p.flags & (AA_GEMM_A_T | AA_GEMM_B_N)) && !p.bias && !p.residual && p.act == AA_ACT_NONE &&
                       p.M % BM == 0 && p.N % BN == 0 && (p.ldc & 7) == 0;
    return plain ? launch4_layout<true>(p, a_t, b_n, st) : launch4_layout<false>(p, a_t, b_n, st);
}
