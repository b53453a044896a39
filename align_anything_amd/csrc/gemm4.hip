// bf16 MFMA GEMM, one-wave-per-SIMD variant of gemm.hip for the big (>= one round of 256 tiles) shapes of the DPO step.
//
// Same contract and epilogue as gemm_kernel (gemm.hip); what changes is the decomposition and the operand pipeline:
//   * 256 x 256 tile, 4 waves (2 x 2), each wave owns 128 x 128 of the output = 64 accumulator tiles of
//     v_mfma_f32_16x16x32_bf16 = 256 registers.  A 128 x 128 wave tile reads (128 + 128) rows of fragments per 32-deep
//     k-step for 64 MFMAs: 1/3 fewer LDS bytes and LDS instructions per flop than the 128 x 64 wave tile of the 8-wave
//     kernel, and half the waves to keep in step at the barrier.
//   * the accumulators live in the ACCUMULATOR register file (a0..a255) and never move: the MFMAs are inline asm with
//     "+a" operands.  (Compiled from the builtin, the same tile makes hipcc shuttle half of the accumulators between
//     the two files -- two v_accvgpr moves per MFMA, profiles/r02_gemm_lab_w4.txt.)  The arch VGPRs hold two fragment
//     sets (128), the DMA lane offsets and the LDS read bases.
//   * operands stream HBM -> LDS by DMA through a RING of four 32 KB slots, one slot = one 32-deep stage ([A 256 x 32 | B 256 x 32]
//     = one MFMA k-step).  Step s multiplies the fragments of stage s (registers), reads the fragments of stage s+1 from its slot
//     and requests stage s+4 into the slot stage s came from: 8 DMA pieces per wave per step, spread over the step, three to four
//     steps of flight.  In-kernel clocks of the earlier two-buffer version (a 64-deep K-tile requested inside ONE 64-MFMA phase)
//     showed that phase at 1400-2000 cycles against 1080 for the DMA-free one (ideal 1024): 64 DMA instructions per workgroup in
//     one phase saturate the CU's address path (16 cycles per 1-KB piece) and the waves stall at ISSUE, not on the data --
//     profiles/r02_gemm4_phase_clocks.txt.
//   * operand DMA addresses are a wave-uniform base (buffer descriptor advanced by SALU per stage) + per-lane 32-bit byte offsets
//     that never change: no 64-bit VALU pointer arithmetic in the loop.
//   * schedule (tools/gen_gemm4_sched.py): per step 64 MFMAs with at most one LDS read or DMA piece behind each, one barrier.
// LDS stage images: K-contiguous operand: row r = 64 bytes at r * 64, 16-byte unit u of the row stored at unit u ^ (2 * bit3(r))
// (conflict-free for ds_read_b128's 16-lane groups); row-contiguous operand ([k][256] as it lies in memory): k-row = 512 bytes,
// 32-byte column blocks XOR-swizzled with tr_swz4(k-row) for ds_read_b64_tr_b16.
// hipBLASLt's own kernel for these shapes has the same decomposition (MT256x256x64, 256 threads, 1 wave / SIMD) and keeps the
// matrix pipe 83 % busy where the 8-wave kernel reaches 62 % (profiles/r02_gemm_vs_hipblaslt_pmc.txt); it is a yardstick only.
#include "aa_common.h"

#include <type_traits>

#include "gemm_params.h"

namespace {

constexpr int BM = 256, BN = 256, NW = 4, WN = 2, TM = 128, TNW = 128, FM = 8, FN = 8;
constexpr int SK = 32;                                        // contraction depth of one ring stage = one MFMA k-step
constexpr int NSLOT = 4, SLOT = 32768, PART = 16384;          // LDS ring: slot = [A part | B part]
constexpr int NP = PART / 1024 / NW;                          // DMA pieces (1 KB) per operand per stage per wave

__device__ __forceinline__ int tr_swz4(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// D = B-fragment x A-fragment (operands swapped like gemm.hip: a lane owns 4 consecutive output columns); the accumulator
// is read and written in place in the accumulator file
#define AA_MFMA_ACC(ACC, BF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(BF), "v"(AF))

// One LDS-DMA piece of a stage: J < NP = A piece J, else B piece J - NP.  M0 (the LDS destination) is a running pointer: the step's
// begin points it at this wave's first A chunk of the target slot, and every piece advances it by the 4 KB to the wave's next chunk
// (the B part follows the A part, so the distance is the same across the operand boundary) -- the one wait state the M0 write needs
// before a DMA reads it is covered by the MFMA in between, no s_nop in the stream.
#define G4_M0_SET(LDSW) asm volatile("s_mov_b32 m0, %0" ::"s"(LDSW) : "memory")
// PLAIN kernels (no edge clamping): the lanes' source pattern is the same for every piece of an operand up to a uniform row
// stride (K-contiguous image) or alternates between two patterns with the piece's parity (row-contiguous image, whose swizzle takes
// bit 3 of the k-row), so ONE or TWO offset VGPRs per operand serve all pieces: buffer_load ... lds with the piece's byte offset in
// the scalar offset operand.  `srd` = buffer descriptor of the operand at the stage being requested.
typedef __attribute__((ext_vector_type(4))) int g4_srd_t;
__device__ __forceinline__ g4_srd_t g4_make_srd(const char* base) {
    const unsigned long long a = (unsigned long long)base;
    return g4_srd_t{(int)(unsigned)a, (int)(unsigned)(a >> 32), 0x7fffffff, 0x00020000};     // stride 0, raw 32-bit format
}
template <int J, bool A_T, bool B_N>
__device__ __forceinline__ void G4_DMA_PIECE_BUF(unsigned vA0, unsigned vA1, unsigned vB0, unsigned vB1, const int (&soA)[NP],
                                                 const int (&soB)[NP], g4_srd_t srdA, g4_srd_t srdB) {
    constexpr int j = J < NP ? J : J - NP;
    const unsigned v = J < NP ? ((A_T && (j & 1)) ? vA1 : vA0) : ((B_N && (j & 1)) ? vB1 : vB0);
    const int so = J < NP ? soA[j] : soB[j];
    if constexpr (J < 2 * NP - 1) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" ::"v"(v), "s"(J < NP ? srdA : srdB), "s"(so), "i"(NW * 1024) : "memory" AA_SCC);
    } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(v), "s"(srdB), "s"(so) : "memory");
    }
}

// general kernel (ragged M / N: per-lane clamped rows): saddr form, wave-uniform 64-bit base + per-lane 32-bit byte offset per piece
template <int J>
__device__ __forceinline__ void G4_DMA_PIECE(const unsigned (&offA)[NP], const unsigned (&offB)[NP], const char* srcA, const char* srcB) {
    if constexpr (J < NP) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(offA[J]), "s"(srcA), "i"(NW * 1024) : "memory" AA_SCC);
    } else if constexpr (J < 2 * NP - 1) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(offB[J - NP]), "s"(srcB), "i"(NW * 1024) : "memory" AA_SCC);
    } else {
        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(offB[J - NP]), "s"(srcB) : "memory");
    }
}

// GRP (mixture-of-experts rows, gemm.hip GRP 1 on this tile): expert segments aligned to BM = 256 rows (aa_moe_plan align = 256; its tile table has one
// entry per 128 rows, both halves of a tile agree); a tile without an expert is zero-filled
__device__ __forceinline__ void g4_zero_tile(const GemmParams& p, int m0, int n0) {
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
    for (int i = threadIdx.x; i < BM * (BN / 8); i += NW * 64) {
        const int r = i / (BN / 8), c = (i % (BN / 8)) * 8;
        *reinterpret_cast<f32x4*>(C + (long)(m0 + r) * p.ldc + n0 + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// XCD-aware bijective remap of the dispatch position, then grouped tile order (identical to gemm_kernel)
__device__ __forceinline__ void g4_map_tile(const GemmParams& p, int& m0, int& n0) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
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
    m0 = tm * BM;
    n0 = tn * BN;
}

#ifdef AA_G4_TIMING
// timing build only (AA_HIPCC_EXTRA=-DAA_G4_TIMING python -m align_anything_amd.build; tools/gemm4_timing2.py): wave 0 of every
// workgroup stamps the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) at four points of its tile, and sums the
// cycles its steps spend in the begin-of-step wait and in the barrier
__device__ unsigned long long g4_timing[8 * 16384];
#define G4_STAMP(K)                                                                                          \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                        \
            g4_timing[blockIdx.x * 8 + (K)] = __builtin_amdgcn_s_memtime();                                  \
            g4_timing[blockIdx.x * 8 + 4 + (K)] = __builtin_amdgcn_s_memrealtime();                          \
        }                                                                                                    \
    } while (0)
#else
#define G4_STAMP(K)
#endif

// PLAIN (EPI != 0): bf16 C = A * B with no bias / activation / accumulate and M, N multiples of the tile (every forward, dX and
// dW GEMM of the 7B decoder stack): the epilogue is straight-line 16-byte stores, and -- its own instantiation -- shares no
// registers with the general epilogue, whose 256-value fan-out would otherwise make the compiler spill accumulators.
template <bool A_T, bool B_N, int EPI, bool GRP = false>
__global__ __launch_bounds__(NW * 64, 1)
void gemm4_kernel(const GemmParams p) {
    // EPI: 0 = general epilogue; 1 = plain bf16 store; 2 = + residual add (o / down projections); 3 = rotary embedding on the q / k
    // heads of a fused qkv projection (head_dim 128 = one wave's columns); 4 = SwiGLU forward (tile = 128 gate + the matching 128 up
    // columns, writes [gate | up] and silu(gate) * up); 5 = SwiGLU backward on the down-projection's dX (writes d[gate | up] only)
    constexpr bool PLAIN = EPI != 0;
    constexpr bool GLU_FWD = EPI == 4;
    static_assert(A_T || B_N, "both operands K-contiguous: gemm4nt_kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G4_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int m0, n0;
    g4_map_tile(p, m0, n0);

    // ---- DMA sources: uniform base (advanced per stage) + per-lane byte offset inside the tile's row / column block.
    // K-contiguous operand: piece c = wave + 4 j covers tile rows 16 c .. 16 c + 15, lane -> row lane >> 2, stored unit lane & 3
    // holds k-unit (lane & 3) ^ 2 * bit3(row) = (lane & 3) ^ (2 * bit5(lane)).  Row-contiguous operand: piece c covers k-rows 2 c, 2 c + 1.
    const char* baseA = reinterpret_cast<const char*>(A_T ? p.A + m0 : p.A + (long)m0 * p.lda);
    const char* baseB;
    if constexpr (GLU_FWD) baseB = reinterpret_cast<const char*>(p.B + (long)(n0 >> 1) * p.ldb);
    else baseB = reinterpret_cast<const char*>(B_N ? p.B + n0 : p.B + (long)n0 * p.ldb);
    if constexpr (GRP) {              // this tile's expert (aa_moe_plan's table, one entry per 128 rows; uniform per workgroup -> scalar load) selects the weight matrix
        const int e = p.grp_tile_expert[m0 / 128];
        if (e < 0) { g4_zero_tile(p, m0, n0); return; }
        baseB += (long)e * p.grp_strideB * 2;
    }
    unsigned offA[NP], offB[NP];
    long stepA, stepB;
    const int kcu = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    if constexpr (!A_T) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int r = (wave + j * NW) * 16 + (lane >> 2);
            const int gr = min(m0 + r, p.M - 1) - m0;
            offA[j] = (unsigned)((gr * p.lda + kcu * 8) * 2);
        }
        stepA = SK * 2;
    } else {
        constexpr int RPI = 1024 / (BM * 2), SPR = BM * 2 / 16;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz4(kr);
            const int col = min(m0 + unit * 16 + (s & 1) * 8, p.M - 8) - m0;
            offA[j] = (unsigned)((kr * p.lda + col) * 2);
        }
        stepA = (long)SK * p.lda * 2;
    }
    if constexpr (!B_N) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int r = (wave + j * NW) * 16 + (lane >> 2);
            const int gr = min(n0 + r, p.N - 1) - n0;
            offB[j] = (unsigned)((gr * p.ldb + kcu * 8) * 2);
        }
        stepB = SK * 2;
    } else {
        constexpr int RPI = 1024 / (BN * 2), SPR = BN * 2 / 16;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = wave + j * NW;
            const int kr = c * RPI + lane / SPR;
            const int s = lane % SPR;
            const int unit = (s >> 1) ^ tr_swz4(kr);
            const int col = min(n0 + unit * 16 + (s & 1) * 8, p.N - 8) - n0;
            offB[j] = (unsigned)((kr * p.ldb + col) * 2);
        }
        stepB = (long)SK * p.ldb * 2;
    }

    // ---- LDS ring: slot sg at sg * 32 KB = [A part 16 KB | B part 16 KB]
    const int lds0 = (int)(uintptr_t)smem;                                  // wave-uniform LDS byte address of the dynamic segment
    const int ldsw = lds0 + wave * 1024;                                    // this wave's first DMA chunk of slot 0
    // PLAIN: per-operand lane patterns (even / odd piece) and the uniform byte distance between consecutive pieces of this wave
    const int pieceA = (A_T ? NW * (1024 / (BM * 2)) : NW * 16) * (int)p.lda * 2;       // rows (k-rows) per piece step x row bytes
    const int pieceB = (B_N ? NW * (1024 / (BN * 2)) : NW * 16) * (int)p.ldb * 2;
    const unsigned vA0 = offA[0], vA1 = A_T ? offA[1] - (unsigned)pieceA : 0u;
    const unsigned vB0 = offB[0], vB1 = B_N ? offB[1] - (unsigned)pieceB : 0u;
    // scalar byte offset of piece j relative to piece 0.  GLU_FWD: the B tile's 256 rows are, per N-wave, 64 gate rows followed by
    // the 64 up rows of the same output columns (F rows further down the fused [gate; up] weight), so that one lane ends up with the
    // gate and the up value of a column: piece j (tile rows 16 wave + 64 j ..) starts at a remapped weight row
    int soA[NP], soB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) soA[j] = j * pieceA;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if constexpr (GLU_FWD) {
            const int r0 = 16 * wave + 64 * j, wq = r0 >> 7, q = r0 & 127;
            const int row = q < 64 ? wq * 64 + q : p.glu_f + wq * 64 + (q - 64);
            soB[j] = (row - 16 * wave) * (int)p.ldb * 2;
        } else {
            soB[j] = j * pieceB;
        }
    }
#define G4_DMA(J)                                                                                                   \
    do {                                                                                                            \
        if constexpr (PLAIN) G4_DMA_PIECE_BUF<J, A_T, B_N>(vA0, vA1, vB0, vB1, soA, soB, g4_make_srd(srcA), g4_make_srd(srcB)); \
        else G4_DMA_PIECE<J>(offA, offB, srcA, srcB);                                                               \
    } while (0)
#define G4_DMA_STAGE() do { G4_DMA(0); G4_DMA(1); G4_DMA(2); G4_DMA(3); G4_DMA(4); G4_DMA(5); G4_DMA(6); G4_DMA(7); } while (0)
    static_assert(NP == 4, "G4_DMA_STAGE and the generated schedule issue 8 pieces per stage");

    // ---- per-lane LDS read addresses.  K-contiguous image: fragment i = rows 16 i .. of the wave's 128 = 1 KB further (immediate);
    // L / H = the register for slots 0-1 / 2-3 (the ds immediate is 16 bits)
    const int l15 = lane & 15, g = lane >> 4;
    [[maybe_unused]] int vakL, vakH, vbkL, vbkH, taL[FM], taH[FM], tbL[FN], tbH[FN];
    {
        const int offK = l15 * 64 + ((g ^ (((l15 >> 3) & 1) << 1)) << 4);
        vakL = lds0 + offK + wm * TM * 64;
        vbkL = lds0 + PART + offK + wn * TNW * 64;
        vakH = vakL + 2 * SLOT;
        vbkH = vbkL + 2 * SLOT;
        const int swz = ((l15 >> 2) & 3) | ((g & 1) << 2);                  // tr_swz4(k-row): independent of the half
        const int lanepart = (g * 8 + (l15 >> 2)) * (BM * 2) + (l15 & 3) * 8;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            taL[i] = lds0 + lanepart + (wm * 8 + (i ^ swz)) * 32;
            tbL[i] = lds0 + PART + lanepart + (wn * 8 + (i ^ swz)) * 32;
            taH[i] = taL[i] + 2 * SLOT;
            tbH[i] = tbL[i] + 2 * SLOT;
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
#define G4_PIN __builtin_amdgcn_sched_barrier(0)
#define G4_JOIN(LO, HI) __builtin_shufflevector(LO, HI, 0, 1, 2, 3, 4, 5, 6, 7)
    // begin of step (slot SG): stage s+1 has landed (the two younger stages, 16 pieces, may still be in flight), this wave's reads of
    // slot SG are complete; after the barrier that holds for every wave, so slot SG may be overwritten and slot SG+1 read.  M0 = this
    // wave's first chunk of slot SG for the step's DMA pieces.
#ifdef AA_G4_TIMING
    unsigned long long g4_wait_cyc = 0, g4_bar_cyc = 0;
#define G4_SYNC(SG)                                                                                                              \
    do {                                                                                                                         \
        unsigned long long ta_, tb_, tc_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_memtime %1\n\ts_barrier\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     "s_mov_b32 m0, %3" : "=&s"(ta_), "=&s"(tb_), "=&s"(tc_) : "s"(ldsw + (SG) * SLOT) : "memory");                \
        g4_wait_cyc += tb_ - ta_;                                                                                                \
        g4_bar_cyc += tc_ - tb_;                                                                                                 \
    } while (0)
#else
#define G4_SYNC(SG) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier\n\ts_mov_b32 m0, %0" ::"s"(ldsw + (SG) * SLOT) : "memory")
#endif
    // the stage requested during step s0 + SG is s0 + SG + 4, clamped to the last one (re-requested into a slot nobody reads any more)
#define G4_STEP_BEGIN(SG)                                                   \
    do {                                                                    \
        const int q_ = min(s0 + (SG) + NSLOT, nsteps - 1);                  \
        srcA = baseA + (long)q_ * stepA;                                    \
        srcB = baseB + (long)q_ * stepB;                                    \
        G4_SYNC(SG);                                                        \
    } while (0)

    const int nsteps = p.K / SK;                                            // a multiple of 4 (dispatch)
    const char* srcA;
    const char* srcB;
    // ---- prologue: stages 0..3 -> slots 0..3, wait for stage 0, first fragment set
#pragma unroll
    for (int sg = 0; sg < NSLOT; ++sg) {
        srcA = baseA + (long)sg * stepA;
        srcB = baseB + (long)sg * stepB;
        G4_M0_SET(ldsw + sg * SLOT);
        asm volatile("s_nop 0");
        G4_DMA_STAGE();
    }
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    {
        // same order as the in-loop reads (b0..b7, a0..a7)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (!B_N) { G4_RDK(b0[j], vbkL, j * 1024); }
            else { G4_RDT(b0h[j][0], tbL[j], 0); G4_RDT(b0h[j][1], tbL[j], 2048); }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (!A_T) { G4_RDK(a0[i], vakL, i * 1024); }
            else { G4_RDT(a0h[i][0], taL[i], 0); G4_RDT(a0h[i][1], taL[i], 2048); }
        }
    }
    G4_STAMP(1);
    // ---- K loop: four ring steps per trip, branch-free
    for (int s0 = 0; s0 < nsteps; s0 += NSLOT) {
        if constexpr (!A_T && B_N) {
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
    G4_STAMP(2);
#ifdef AA_G4_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 16384) {      // (overwrites the two wall-clock stamps nobody reads)
        g4_timing[blockIdx.x * 8 + 5] = g4_wait_cyc;
        g4_timing[blockIdx.x * 8 + 6] = g4_bar_cyc;
    }
#endif
    // the clamped re-requests of the last stage and the (unused) fragment reads of the last step: nothing of this tile may still be
    // writing LDS or registers when compiler-scheduled code runs
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the last MFMAs are still in the matrix pipe: the compiler does not know the asm statements wrote the accumulators late
    // (nothing may read the last row's accumulators above this statement: they are its operands)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[FM - 1][0]), "+a"(acc[FM - 1][1]), "+a"(acc[FM - 1][2]), "+a"(acc[FM - 1][3]), "+a"(acc[FM - 1][4]),
                   "+a"(acc[FM - 1][5]), "+a"(acc[FM - 1][6]), "+a"(acc[FM - 1][7])
                 :: "memory");

    {
#include "gemm4_epilogue.inc"
    }
#ifdef AA_G4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have been issued to memory
    G4_STAMP(3);
#endif
}


// ================================================================================================================================
// NT layout (both operands K-contiguous: every forward GEMM).  A K-contiguous row delivers 128 bytes = 64 k per cache line, so the
// DMA unit stays the 64-deep K-tile (piece = 8 rows x 128 B: eight full lines per request; 64-byte row segments -- 32-deep stages,
// or 64-deep tiles shifted by half a tile -- were measured 10-50 % slower under load, profiles/r02_gemm4_phase_clocks.txt).  Two
// operands with two buffers each would both free a buffer at the same step and both need their next tile requested within that
// one step (the two-buffer version did exactly that and stalled at instruction issue).  Here A has THREE 32 KB buffers and B two
// (160 KB, the whole LDS of the CU, one workgroup per CU as before): the buffer of A tile t+2 is free a tile early, so EVEN steps
// request the next A tile (two steps of flight beyond its own) and ODD steps the next B tile (into the buffer that step's begin
// frees; one step of flight beyond its own) -- 8 full-line pieces per wave in every step.
// LDS image of a tile: row r = 128 bytes at r * 128, 16-byte unit u stored at u ^ ((r >> 1) & 7).
constexpr int NP8 = 32768 / 1024 / NW;                        // DMA pieces per operand per 64-deep tile per wave
constexpr int NT_B0 = 3 * 32768, NT_LDS = 5 * 32768;          // B buffers behind the three A buffers

template <int J>
__device__ __forceinline__ void G4NT_DMA_PIECE_BUF(unsigned v, const int (&so)[NP8], g4_srd_t srd) {
    if constexpr (J < NP8 - 1) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" ::"v"(v), "s"(srd), "s"(so[J]), "i"(NW * 1024) : "memory" AA_SCC);
    } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(v), "s"(srd), "s"(so[J]) : "memory");
    }
}
template <int J>
__device__ __forceinline__ void G4NT_DMA_PIECE(unsigned v, const char* src) {
    if constexpr (J < NP8 - 1) {
        asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, %2" ::"v"(v), "s"(src), "i"(NW * 1024) : "memory" AA_SCC);
    } else {
        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(v), "s"(src) : "memory");
    }
}

template <int EPI, bool GRP = false>
__global__ __launch_bounds__(NW * 64, 1)
void gemm4nt_kernel(const GemmParams p) {
    constexpr bool PLAIN = EPI != 0;
    constexpr bool GLU_FWD = EPI == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G4_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    g4_map_tile(p, m0, n0);

    // ---- DMA sources.  Piece c = wave + 4 j covers tile rows 8 c .. 8 c + 7; lane -> row lane >> 3, stored unit lane & 7 holds
    // k-unit (lane & 7) ^ ((row >> 1) & 7) of the tile's 64 k
    const char* baseA = reinterpret_cast<const char*>(p.A + (long)m0 * p.lda);
    const char* baseB;
    if constexpr (GLU_FWD) baseB = reinterpret_cast<const char*>(p.B + (long)(n0 >> 1) * p.ldb);
    else baseB = reinterpret_cast<const char*>(p.B + (long)n0 * p.ldb);
    if constexpr (GRP) {
        const int e = p.grp_tile_expert[m0 / 128];
        if (e < 0) { g4_zero_tile(p, m0, n0); return; }
        baseB += (long)e * p.grp_strideB * 2;
    }
    unsigned offA[NP8], offB[NP8];
    {
        const int rr = wave * 8 + (lane >> 3);                 // row of piece 0; piece j is 32 rows further: (row >> 1) & 7 unchanged
        const int ks = (lane & 7) ^ ((rr >> 1) & 7);
#pragma unroll
        for (int j = 0; j < NP8; ++j) {
            const int r = rr + j * NW * 8;
            offA[j] = (unsigned)(((min(m0 + r, p.M - 1) - m0) * p.lda + ks * 8) * 2);
            offB[j] = (unsigned)(((min(n0 + r, p.N - 1) - n0) * p.ldb + ks * 8) * 2);
        }
    }
    const int lds0 = (int)(uintptr_t)smem;
    const int ldsw = lds0 + wave * 1024;
    const int pieceA = NW * 8 * (int)p.lda * 2, pieceB = NW * 8 * (int)p.ldb * 2;
    int soA[NP8], soB[NP8];
#pragma unroll
    for (int j = 0; j < NP8; ++j) {
        soA[j] = j * pieceA;
        if constexpr (GLU_FWD) {       // B tile rows: per N-wave 64 gate rows then the 64 up rows of the same columns (F rows further down)
            const int r0 = 8 * wave + 32 * j, wq = r0 >> 7, q = r0 & 127;
            const int row = q < 64 ? wq * 64 + q : p.glu_f + wq * 64 + (q - 64);
            soB[j] = (row - 8 * wave) * (int)p.ldb * 2;
        } else {
            soB[j] = j * pieceB;
        }
    }
    // one request group = the 8 pieces of one operand tile
#define G4NT_DMA_A(J)                                                                \
    do {                                                                             \
        if constexpr (PLAIN) G4NT_DMA_PIECE_BUF<J>(offA[0], soA, g4_make_srd(srcA)); \
        else G4NT_DMA_PIECE<J>(offA[J], srcA);                                       \
    } while (0)
#define G4NT_DMA_B(J)                                                                \
    do {                                                                             \
        if constexpr (PLAIN) G4NT_DMA_PIECE_BUF<J>(offB[0], soB, g4_make_srd(srcB)); \
        else G4NT_DMA_PIECE<J>(offB[J], srcB);                                       \
    } while (0)
#define G4NT_GROUP_A() do { G4NT_DMA_A(0); G4NT_DMA_A(1); G4NT_DMA_A(2); G4NT_DMA_A(3); G4NT_DMA_A(4); G4NT_DMA_A(5); G4NT_DMA_A(6); G4NT_DMA_A(7); } while (0)
#define G4NT_GROUP_B() do { G4NT_DMA_B(0); G4NT_DMA_B(1); G4NT_DMA_B(2); G4NT_DMA_B(3); G4NT_DMA_B(4); G4NT_DMA_B(5); G4NT_DMA_B(6); G4NT_DMA_B(7); } while (0)
    static_assert(NP8 == 8, "request groups are written out for 8 pieces");

    // ---- LDS: A tile t in buffer t % 3 at (t % 3) * 32 KB, B tile u at 96 KB + (u & 1) * 32 KB; fragment i = 16 rows = 2 KB further
    // (immediate).  The A read bases move with the tile (two adds per tile), B's parity is static in the unrolled trip.
    const int l15 = lane & 15, g = lane >> 4;
    int vak[2], vbk[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int offK = l15 * 128 + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) << 4);
        vak[kk] = lds0 + offK + wm * TM * 128;
        vbk[kk] = lds0 + NT_B0 + offK + wn * TNW * 128;
    }
    const int vbk0 = vbk[0], vbk1 = vbk[1];
    int vak0 = vak[0], vak1 = vak[1];                          // bases of the A tile whose fragments the current step reads

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];

    // begin of a step: the tiles the step reads have landed (even steps: the last two request groups may be in flight, odd steps: the
    // last one), this wave's fragment reads are complete; after the barrier that holds for every wave.  M0 = this wave's first chunk
    // of the buffer the step's request group fills.
#ifdef AA_G4_TIMING
    unsigned long long g4_wait_cyc = 0, g4_bar_cyc = 0;
#define G4NT_SYNC(VM, DST)                                                                                                       \
    do {                                                                                                                         \
        unsigned long long ta_, tb_, tc_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(" #VM ") lgkmcnt(0)\n\ts_memtime %1\n\ts_barrier\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     "s_mov_b32 m0, %3" : "=&s"(ta_), "=&s"(tb_), "=&s"(tc_) : "s"(DST) : "memory");                               \
        g4_wait_cyc += tb_ - ta_;                                                                                                \
        g4_bar_cyc += tc_ - tb_;                                                                                                 \
    } while (0)
#else
#define G4NT_SYNC(VM, DST) asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)\n\ts_barrier\n\ts_mov_b32 m0, %0" ::"s"(DST) : "memory")
#endif
    const int nsteps = p.K / SK;                               // a multiple of 4 (dispatch)
    const int last = nsteps / 2 - 1;                           // last 64-deep tile
    // even step s = 2 t (SG 0 / 2): request A tile t + 2 into buffer (t + 2) % 3, and the step reads the UPPER half of A tile t;
    // odd step s = 2 u + 1: request B tile u + 2 into buffer u & 1, and the step reads the LOWER half of tile u + 1: the A read
    // bases move on to that tile.  Tiles past the last one are re-requests of the last (into buffers nobody reads any more).
#define G4NT_STEP_BEGIN(SG)                                                                     \
    do {                                                                                        \
        if constexpr (((SG) & 1) == 0) {                                                        \
            srcA = baseA + (long)min((s0 + (SG)) / 2 + 2, last) * 128;                          \
            G4NT_SYNC(16, ldsw + a_wr);                                                         \
            a_wr = a_wr == 65536 ? 0 : a_wr + 32768;                                            \
        } else {                                                                                \
            srcB = baseB + (long)min((s0 + (SG)) / 2 + 2, last) * 128;                          \
            a_rd = a_rd == 65536 ? 0 : a_rd + 32768;                                            \
            vak0 = vak[0] + a_rd;                                                               \
            vak1 = vak[1] + a_rd;                                                               \
            G4NT_SYNC(8, ldsw + NT_B0 + ((SG) >> 1) * 32768);                                   \
        }                                                                                       \
    } while (0)

    const char* srcA;
    const char* srcB;
    int a_wr = 65536, a_rd = 0;                                // byte offsets of the A buffer requested next / read now
    // ---- prologue: A tile 0, B tile 0, A tile 1, B tile 1; the first two have landed -> first fragment set
    srcA = baseA;
    G4_M0_SET(ldsw); asm volatile("s_nop 0");
    G4NT_GROUP_A();
    srcB = baseB;
    G4_M0_SET(ldsw + NT_B0); asm volatile("s_nop 0");
    G4NT_GROUP_B();
    srcA = baseA + 128;
    G4_M0_SET(ldsw + 32768); asm volatile("s_nop 0");
    G4NT_GROUP_A();
    srcB = baseB + 128;
    G4_M0_SET(ldsw + NT_B0 + 32768); asm volatile("s_nop 0");
    G4NT_GROUP_B();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    {
        // stage 0 = lower halves of A tile 0 and B tile 0; same order as the in-loop reads (b0..b7, a0..a7)
#pragma unroll
        for (int j = 0; j < FN; ++j) G4_RDK(b0[j], vbk0, j * 2048);
#pragma unroll
        for (int i = 0; i < FM; ++i) G4_RDK(a0[i], vak0, i * 2048);
    }
    G4_STAMP(1);
    for (int s0 = 0; s0 < nsteps; s0 += 4) {
#define G4_FRAG_A(S, I) a##S[I]
#define G4_FRAG_B(S, J) b##S[J]
#include "gemm4_sched_nt.inc"
#undef G4_FRAG_A
#undef G4_FRAG_B
    }
    G4_STAMP(2);
#ifdef AA_G4_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g4_timing[blockIdx.x * 8 + 5] = g4_wait_cyc;
        g4_timing[blockIdx.x * 8 + 6] = g4_bar_cyc;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[FM - 1][0]), "+a"(acc[FM - 1][1]), "+a"(acc[FM - 1][2]), "+a"(acc[FM - 1][3]), "+a"(acc[FM - 1][4]),
                   "+a"(acc[FM - 1][5]), "+a"(acc[FM - 1][6]), "+a"(acc[FM - 1][7])
                 :: "memory");
    {
#include "gemm4_epilogue.inc"
    }
#ifdef AA_G4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G4_STAMP(3);
#endif
}

template <bool A_T, bool B_N, int EPI>
int launch4(GemmParams& p, hipStream_t st) {
    constexpr int lds = (!A_T && !B_N) ? NT_LDS : NSLOT * SLOT;
    void (*kern)(const GemmParams);
    if constexpr (!A_T && !B_N) kern = gemm4nt_kernel<EPI>;
    else kern = gemm4_kernel<A_T, B_N, EPI>;
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

// mixture-of-experts rows on the plain epilogue: forward (NT) and dX (NN)
template <bool B_N>
int launch4_grouped(GemmParams& p, hipStream_t st) {
    constexpr int lds = !B_N ? NT_LDS : NSLOT * SLOT;
    void (*kern)(const GemmParams);
    if constexpr (!B_N) kern = gemm4nt_kernel<1, true>;
    else kern = gemm4_kernel<false, true, 1, true>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_grouped_bf16 (4-wave tile): cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(NW * 64), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_grouped_bf16");
    return AA_OK;
}

template <int EPI>
int launch4_layout(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if constexpr (EPI == 5) {          // SwiGLU backward rides on the dX (NN) GEMM of the down projection
        return launch4<false, true, EPI>(p, st);
    } else {
        if (!a_t && !b_n) return launch4<false, false, EPI>(p, st);
        if constexpr (EPI <= 1) {      // the residual / rotary / SwiGLU-forward epilogues exist for the forward (NT) layout only
            if (!a_t && b_n) return launch4<false, true, EPI>(p, st);
            if (a_t && b_n) return launch4<true, true, EPI>(p, st);
        }
        aa_set_error("aa_gemm_bf16: layout not built for this epilogue (A^T with K-contiguous B is unused by the hot path)");
        return AA_ERR_ARG;
    }
}

}  // namespace

#ifdef AA_G4_TIMING
extern "C" int aa_gemm4_timing_dump(unsigned long long* out, int n_words) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g4_timing), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? AA_OK : AA_ERR_LAUNCH;
}
#endif

// The ring walks the contraction four 32-deep stages per trip: K must be a multiple of 128 (every 7B shape is).
bool aa_gemm4_supports(int K) { return K >= NSLOT * SK && K % (NSLOT * SK) == 0; }

// p.tiles_m / tiles_n / gm are set by the caller (gemm.hip), which has checked aa_gemm4_supports(p.K).
int aa_gemm4_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if (!aa_gemm4_supports(p.K)) {
        aa_set_error("aa_gemm_bf16 (4-wave tile): K=%d must be a multiple of %d", p.K, NSLOT * SK);
        return AA_ERR_ARG;
    }
    const bool shape_ok = p.flags == (p.flags & (AA_GEMM_A_T | AA_GEMM_B_N)) && !p.bias && p.act == AA_ACT_NONE && p.M % BM == 0 &&
                          p.N % BN == 0 && (p.ldc & 7) == 0;
    const bool plain = shape_ok && !p.residual;
    const bool resid = shape_ok && p.residual && !a_t && !b_n && (p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0;
    if (plain) return launch4_layout<1>(p, a_t, b_n, st);
    if (resid) return launch4_layout<2>(p, a_t, b_n, st);
    return launch4_layout<0>(p, a_t, b_n, st);
}

// Grouped (mixture-of-experts) rows: C[cap, N] = A[cap, K] op(B[e]) with e = p.grp_tile_expert[row / 256], expert segments aligned to 256 rows (the caller's
// promise: aa_moe_plan with align 256).  Returns 1 when the shape does not fit the tile (the caller then runs the 128 x 256 8-wave kernel).
int aa_gemm4_grouped(GemmParams& p, bool b_n, hipStream_t st) {
    if (p.M % BM || p.N % BN || !aa_gemm4_supports(p.K) || (p.ldc & 7) || ((uintptr_t)p.C & 15) || p.bias || p.residual || p.act != AA_ACT_NONE ||
        (p.flags & ~(AA_GEMM_B_N)) != 0 || !p.grp_tile_expert)
        return 1;
    p.tiles_m = p.M / BM;
    p.tiles_n = p.N / BN;
    p.gm = 4;
    return b_n ? launch4_grouped<true>(p, st) : launch4_grouped<false>(p, st);
}

// Fused epilogues (p.fuse = AA_FUSE_*).  Returns 1 when the shape does not qualify (the caller then runs the unfused pair of kernels):
// M, N multiples of the 256 tile, K of 128, 16-byte aligned rows everywhere, and per mode: ROPE head_dim 128 with rope_cols a multiple
// of 128; GLU_FWD F a multiple of 128 (N = 2F); GLU_BWD N = F.
int aa_gemm4_fused(GemmParams& p, hipStream_t st) {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (p.M % BM || p.N % BN || !aa_gemm4_supports(p.K) || (p.ldc & 7) || p.bias || p.residual || p.act != AA_ACT_NONE) return 1;
    p.tiles_m = p.M / BM;
    p.tiles_n = p.N / BN;
    if (p.fuse == AA_FUSE_ROPE) {
        if (!p.rope_pos || !al16(p.rope_cos) || !al16(p.rope_sin) || p.rope_cols % 128 || p.rope_cols < 0 || p.rope_cols > p.N) return 1;
        return launch4_layout<3>(p, false, false, st);
    }
    if (p.fuse == AA_FUSE_GLU_FWD) {
        if (p.glu_f % 128 || p.N != 2 * p.glu_f || !p.aux || !al16(p.aux) || (p.ldaux & 7) || (p.glu_f & 7)) return 1;
        return launch4_layout<4>(p, false, false, st);
    }
    if (p.fuse == AA_FUSE_GLU_BWD) {
        if (p.N != p.glu_f || !p.aux || !p.aux_in || !al16(p.aux) || !al16(p.aux_in) || (p.ldaux & 7) || (p.ldaux_in & 7) || (p.glu_f & 7)) return 1;
        return launch4_layout<5>(p, false, true, st);
    }
    return 1;
}
