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
// PLAIN kernels (no edge clamping): the lanes' source pattern is the same for every piece of an operand up to a uniform row
// stride (K-contiguous image) or alternates between two patterns with the piece's parity (row-contiguous image, whose swizzle takes
// bit 3 of the k-row), so ONE or TWO offset VGPRs per operand serve all 8 pieces: buffer_load ... lds with the piece's byte offset in
// the scalar offset operand.  `srd` = buffer descriptor of the operand at the K-tile being requested (base advanced by SALU).
typedef __attribute__((ext_vector_type(4))) int g4_srd_t;
__device__ __forceinline__ g4_srd_t g4_make_srd(const char* base) {
    const unsigned long long a = (unsigned long long)base;
    return g4_srd_t{(int)(unsigned)a, (int)(unsigned)(a >> 32), 0x7fffffff, 0x00020000};     // stride 0, raw 32-bit format
}
template <int K, bool A_T, bool B_N>
__device__ __forceinline__ void G4_DMA_PIECE_BUF(unsigned vA0, unsigned vA1, unsigned vB0, unsigned vB1, const int (&soA)[A_IT],
                                                 const int (&soB)[B_IT], g4_srd_t srdA, g4_srd_t srdB) {
    constexpr int j = K < A_IT ? K : K - A_IT;
    constexpr int adv = K == A_IT - 1 ? 65536 - (A_IT - 1) * NW * 1024 : NW * 1024;
    const unsigned v = K < A_IT ? ((A_T && (j & 1)) ? vA1 : vA0) : ((B_N && (j & 1)) ? vB1 : vB0);
    const int so = K < A_IT ? soA[j] : soB[j];
    if constexpr (K < A_IT + B_IT - 1) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" ::"v"(v), "s"(K < A_IT ? srdA : srdB), "s"(so), "i"(adv) : "memory");
    } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(v), "s"(srdB), "s"(so) : "memory");
    }
}

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
// PERSIST (PLAIN only, K >= 2 K-tiles): at most one workgroup per CU walks the tile list with stride gridDim.x and treats its
// tiles as ONE stream of K-tiles -- the requests of the next tile's first two K-tiles ride in the last two iterations of the current
// tile, so the epilogue's stores overlap their flight and no tile after the first pays a prologue (or a workgroup launch).
template <bool A_T, bool B_N, int EPI, bool PERSIST>
__global__ __launch_bounds__(NW * 64, 1)
void gemm4_kernel(const GemmParams p) {
    // EPI: 0 = general epilogue; 1 = plain bf16 store; 2 = + residual add (o / down projections); 3 = rotary embedding on the q / k
    // heads of a fused qkv projection (head_dim 128 = one wave's columns); 4 = SwiGLU forward (tile = 128 gate + the matching 128 up
    // columns, writes [gate | up] and silu(gate) * up); 5 = SwiGLU backward on the down-projection's dX (writes d[gate | up] only)
    constexpr bool PLAIN = EPI != 0;
    constexpr bool GLU_FWD = EPI == 4;
    static_assert(PLAIN || !PERSIST, "the persistent walk relies on tile-independent DMA lane offsets (no edge clamping)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware bijective remap, then grouped tile order (identical to gemm_kernel); `bid` = position in dispatch order
    const int nwg = p.tiles_m * p.tiles_n;
    auto map_tile = [&](int bid, int& m0_, int& n0_) {
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
        m0_ = tm * BM;
        n0_ = tn * BN;
    };
    int m0, n0;
    map_tile(blockIdx.x, m0, n0);
    auto tile_base_a = [&](int m0_) { return reinterpret_cast<const char*>(A_T ? p.A + m0_ : p.A + (long)m0_ * p.lda); };
    auto tile_base_b = [&](int n0_) {
        if constexpr (GLU_FWD) return reinterpret_cast<const char*>(p.B + (long)(n0_ >> 1) * p.ldb);
        return reinterpret_cast<const char*>(B_N ? p.B + n0_ : p.B + (long)n0_ * p.ldb);
    };

    // ---- DMA sources: uniform base (advanced per K-tile) + per-lane byte offset inside the tile's row / column block
    const char* baseA;
    const char* baseB;
    unsigned offA[A_IT], offB[B_IT];
    long stepA, stepB;
    if constexpr (!A_T) {
        baseA = tile_base_a(m0);
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
        baseA = tile_base_a(m0);
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
        baseB = tile_base_b(n0);
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
        baseB = tile_base_b(n0);
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
    // PLAIN: per-operand lane patterns (even / odd piece) and the uniform byte distance between consecutive pieces of this wave
    const unsigned vA0 = offA[0], vA1 = A_T ? offA[1] - (unsigned)(NW * (1024 / (BM * 2)) * p.lda * 2) : 0u;
    const unsigned vB0 = offB[0], vB1 = B_N ? offB[1] - (unsigned)(NW * (1024 / (BN * 2)) * p.ldb * 2) : 0u;
    const int pieceA = (A_T ? NW * (1024 / (BM * 2)) : NW * 8) * (int)p.lda * 2;       // rows (k-rows) per piece step x row bytes
    const int pieceB = (B_N ? NW * (1024 / (BN * 2)) : NW * 8) * (int)p.ldb * 2;
    // scalar byte offset of piece j relative to piece 0.  GLU_FWD: the B tile's 256 rows are, per N-wave, 64 gate rows followed by
    // the 64 up rows of the same output columns (F rows further down the fused [gate; up] weight), so that one lane ends up with the
    // gate and the up value of a column: piece j (tile rows 8 wave + 32 j ..) starts at a remapped weight row
    int soA[A_IT], soB[B_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) soA[j] = j * pieceA;
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        if constexpr (GLU_FWD) {
            const int r0 = 8 * wave + 32 * j, wq = r0 >> 7, q = r0 & 127;
            const int row = q < 64 ? wq * 64 + q : p.glu_f + wq * 64 + (q - 64);
            soB[j] = (row - 8 * wave) * (int)p.ldb * 2;
        } else {
            soB[j] = j * pieceB;
        }
    }
#define G4_DMA(K)                                                                                                   \
    do {                                                                                                            \
        if constexpr (PLAIN) G4_DMA_PIECE_BUF<K, A_T, B_N>(vA0, vA1, vB0, vB1, soA, soB, g4_make_srd(srcA), g4_make_srd(srcB)); \
        else G4_DMA_PIECE<K>(offA, offB, srcA, srcB);                                                               \
    } while (0)

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
    // ---- tile walk (one tile unless PERSIST).  `u` counts the K-tiles of the whole walk: buffer parity = u & 1.
    int u = 0;
    for (int bid = blockIdx.x;;) {
        const int bid_next = bid + (int)gridDim.x;
        const bool has_next = PERSIST && bid_next < nwg;
        // operands of the tile after this one (PERSIST): its first two K-tiles are requested by this tile's last two iterations
        const char* nextA = baseA;
        const char* nextB = baseB;
        int m0n = m0, n0n = n0;
        if (has_next) {
            map_tile(bid_next, m0n, n0n);
            nextA = tile_base_a(m0n);
            nextB = tile_base_b(n0n);
        }
        // ---- K loop: one K-tile per iteration, branch-free.  The request two K-tiles ahead wraps into the next tile; without one it
        // is clamped to the last K-tile (re-requested into a buffer nobody reads any more).
        for (int t = 0; t < nt; ++t, ++u) {
            const int cbc = (u & 1) * 32768, cbn = cbc ^ 32768;                  // buffer offsets: current / next K-tile
            const bool wrap = has_next && t + 2 >= nt;
            const int t2 = wrap ? t + 2 - nt : min(t + 2, nt - 1);
            const char* srcA = (wrap ? nextA : baseA) + (long)t2 * stepA;
            const char* srcB = (wrap ? nextB : baseB) + (long)t2 * stepB;
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
        if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped re-requests of the last K-tile
        // PERSIST: the next tile's first fragment set was read (asm, not tracked by the compiler) during the last phase; let it land
        // before compiler-scheduled code runs, so that whatever the register allocator does with those registers here is safe
        if constexpr (PERSIST) G4_WAIT_LGKM(0);
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
            // (the lane id is laundered through an empty asm: the store addresses are then recomputed per tile -- a handful of VALU --
            // instead of being hoisted out of the tile walk as loop invariants that sit in registers across the whole K loop)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int e15 = ln & 15, eg = ln >> 4;
            const long erow = m0 + wm * TM + e15;                           // + 16 i
            const int ecol = (eg & 1) * 16 + (eg >> 1) * 8;                 // + 16 j: first of this lane's 8 columns inside the wave's 128
            // packed bf16 of the fragment pair (j, j+1), redistributed so that this lane holds 8 consecutive columns of fragment j + (g & 1)
            auto pack_pair = [&](int i, int j) -> u32x4 {
                asm volatile("" : "+a"(acc[i][j]), "+a"(acc[i][j + 1])::"memory");
                unsigned w[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        w[f][h] = (unsigned)f2bf(acc[i][j + f][2 * h]) | ((unsigned)f2bf(acc[i][j + f][2 * h + 1]) << 16);
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
                return u32x4{lo[0], hi[0], lo[1], hi[1]};
            };
            auto lo16 = [](unsigned x) { return bf2f((bf16_t)(x & 0xffff)); };
            auto hi16 = [](unsigned x) { return bf2f((bf16_t)(x >> 16)); };
            auto pk = [](float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); };
            if (EPI == 1 && (p.grp_strideC & 1)) {
                // ABLATION (AA_GEMM_ABLATE=1, timing only): no conversion, no stores -- what does the epilogue cost?
            } else if constexpr (EPI == 1 || EPI == 2) {
                bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + erow * p.ldc + n0 + wn * TNW + ecol;
                [[maybe_unused]] const bf16_t* rrow = p.residual + erow * p.ldr + n0 + wn * TNW + ecol;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; j += 2) {
                        u32x4 o = pack_pair(i, j);
                        if constexpr (EPI == 2) {
                            // HF: `residual + linear(x)`: the projection's bf16 output (what `o` holds) plus the bf16 residual, rounded once more
                            const u32x4 r = *reinterpret_cast<const u32x4*>(rrow + (long)i * 16 * p.ldr + j * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = pk(lo16(o[e]) + lo16(r[e]), hi16(o[e]) + hi16(r[e]));
                        }
                        *reinterpret_cast<u32x4*>(crow + (long)i * 16 * p.ldc + j * 16) = o;
                        __builtin_amdgcn_sched_barrier(0);  // one fragment pair at a time: a dozen live registers, not a 256-value fan-out
                    }
                }
            } else if constexpr (EPI == 3) {
                // rotary embedding (hf:models/llama/modeling_llama.py:130-160 on the bf16 projection output, rounding points of
                // aa_rope_inplace): the wave's 128 columns are one head; fragments j < 4 hold d < 64, fragment j + 4 holds d + 64
                const int colw = n0 + wn * TNW;
                const bool rot = colw < p.rope_cols;                        // q / k head (v heads are stored as they are)
                bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + erow * p.ldc + colw + ecol;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const long tb = (long)p.rope_pos[erow + i * 16] * 64 + ecol;
#pragma unroll
                    for (int j = 0; j < FN / 2; j += 2) {
                        u32x4 x1 = pack_pair(i, j), x2 = pack_pair(i, j + 4);
                        if (rot) {
                            const u32x4 c = *reinterpret_cast<const u32x4*>(p.rope_cos + tb + j * 16);
                            const u32x4 sn = *reinterpret_cast<const u32x4*>(p.rope_sin + tb + j * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a0 = lo16(x1[e]), a1 = hi16(x1[e]), b0 = lo16(x2[e]), b1 = hi16(x2[e]);
                                const float c0 = lo16(c[e]), c1 = hi16(c[e]), s0 = lo16(sn[e]), s1 = hi16(sn[e]);
                                x1[e] = pk(rbf(a0 * c0) + rbf(-b0 * s0), rbf(a1 * c1) + rbf(-b1 * s1));
                                x2[e] = pk(rbf(b0 * c0) + rbf(a0 * s0), rbf(b1 * c1) + rbf(a1 * s1));
                            }
                        }
                        *reinterpret_cast<u32x4*>(crow + (long)i * 16 * p.ldc + j * 16) = x1;
                        *reinterpret_cast<u32x4*>(crow + (long)i * 16 * p.ldc + j * 16 + 64) = x2;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if constexpr (EPI == 4) {
                // SwiGLU forward (hf LlamaMLP: act_fn(gate_proj(x)) * up_proj(x), every factor bf16; rounding points of aa_swiglu_fwd):
                // fragments j < 4 = gate columns, j + 4 = the up values of the same columns
                const int gcol = (n0 >> 1) + wn * 64 + ecol;
                bf16_t* gu = reinterpret_cast<bf16_t*>(p.C) + erow * p.ldc + gcol;
                bf16_t* act = reinterpret_cast<bf16_t*>(p.aux) + erow * p.ldaux + gcol;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN / 2; j += 2) {
                        const u32x4 gt = pack_pair(i, j), up = pack_pair(i, j + 4);
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g0 = lo16(gt[e]), g1 = hi16(gt[e]);
                            o[e] = pk(rbf(g0 / (1.f + expf(-g0))) * lo16(up[e]), rbf(g1 / (1.f + expf(-g1))) * hi16(up[e]));
                        }
                        *reinterpret_cast<u32x4*>(gu + (long)i * 16 * p.ldc + j * 16) = gt;
                        *reinterpret_cast<u32x4*>(gu + (long)i * 16 * p.ldc + j * 16 + p.glu_f) = up;
                        *reinterpret_cast<u32x4*>(act + (long)i * 16 * p.ldaux + j * 16) = o;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // SwiGLU backward (EPI 5): this tile of d_act = dY W_down (bf16-rounded like the stored tensor it replaces) with the saved
                // [gate | up] -> d[gate | up] (aa_swiglu_bwd arithmetic); d_act itself is never written
                const int col = n0 + wn * TNW + ecol;
                const bf16_t* gu = p.aux_in + erow * p.ldaux_in + col;
                bf16_t* dgu = reinterpret_cast<bf16_t*>(p.aux) + erow * p.ldaux + col;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; j += 2) {
                        const u32x4 d = pack_pair(i, j);
                        const u32x4 gt = *reinterpret_cast<const u32x4*>(gu + (long)i * 16 * p.ldaux_in + j * 16);
                        const u32x4 up = *reinterpret_cast<const u32x4*>(gu + (long)i * 16 * p.ldaux_in + j * 16 + p.glu_f);
                        u32x4 og, ou;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float r[2][2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float gf = h ? hi16(gt[e]) : lo16(gt[e]), uf = h ? hi16(up[e]) : lo16(up[e]), df = h ? hi16(d[e]) : lo16(d[e]);
                                const float sg = 1.f / (1.f + expf(-gf));
                                r[0][h] = df * uf * sg * (1.f + gf * (1.f - sg));
                                r[1][h] = df * gf * sg;
                            }
                            og[e] = pk(r[0][0], r[0][1]);
                            ou[e] = pk(r[1][0], r[1][1]);
                        }
                        *reinterpret_cast<u32x4*>(dgu + (long)i * 16 * p.ldaux + j * 16) = og;
                        *reinterpret_cast<u32x4*>(dgu + (long)i * 16 * p.ldaux + j * 16 + p.glu_f) = ou;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
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
        if (!has_next) break;
        // ---- next tile of the walk: fresh accumulators; its first fragment set and K-tile requests are already under way
        bid = bid_next;
        m0 = m0n; n0 = n0n;
        baseA = nextA; baseB = nextB;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (the zeroing of row i sits above this statement, and the two wait states an accumulator write needs before an MFMA
            // reads it are inside it: the compiler pads nothing around asm)
            asm volatile("s_nop 1"
                         : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]),
                           "+a"(acc[i][6]), "+a"(acc[i][7]));
        }
    }
}

template <bool A_T, bool B_N, int EPI, bool PERSIST>
int launch4(GemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * STAGE;
    auto kern = gemm4_kernel<A_T, B_N, EPI, PERSIST>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            aa_set_error("aa_gemm_bf16 (4-wave tile): cannot reserve %d B LDS: %s", lds, hipGetErrorString(e));
            return AA_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    static int cus = 0;                  // persistent walk: one workgroup per CU (128 KB of LDS each: a CU holds exactly one)
    if (PERSIST && !cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        cus &= ~7;                       // multiple of the 8 XCDs: block b and block b + grid run on the same XCD
        if (cus < 8) cus = 8;
    }
    const int grid = PERSIST ? (tiles < cus ? tiles : cus) : tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p);
    AA_CHECK_LAUNCH("aa_gemm_bf16");
    return AA_OK;
}

template <int EPI, bool PERSIST>
int launch4_layout(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    if constexpr (EPI == 5) {          // SwiGLU backward rides on the dX (NN) GEMM of the down projection
        return launch4<false, true, EPI, PERSIST>(p, st);
    } else {
        if (!a_t && !b_n) return launch4<false, false, EPI, PERSIST>(p, st);
        if constexpr (EPI <= 1) {      // the residual / rotary / SwiGLU-forward epilogues exist for the forward (NT) layout only
            if (!a_t && b_n) return launch4<false, true, EPI, PERSIST>(p, st);
            if (a_t && b_n) return launch4<true, true, EPI, PERSIST>(p, st);
        }
        aa_set_error("aa_gemm_bf16: layout not built for this epilogue (A^T with K-contiguous B is unused by the hot path)");
        return AA_ERR_ARG;
    }
}

int g4_persist() {
    static int persist = -1;
    // default off: the walk measured +-1 % against one workgroup per tile on every hot shape and on the whole step (804.96 vs 804.98 ms,
    // profiles/r02_gemm_vs_hipblaslt_pmc.txt) -- the per-tile cost it removes (launch, prologue) is not where the tile's fixed ~8 us go
    if (persist < 0) { const char* e = getenv("AA_GEMM_PERSIST"); persist = e ? atoi(e) : 0; }
    return persist;
}

}  // namespace

// p.tiles_m / tiles_n / gm are set by the caller (gemm.hip).  AA_GEMM_PERSIST=0: one workgroup per tile for the plain kernels too.
int aa_gemm4_dispatch(GemmParams& p, bool a_t, bool b_n, hipStream_t st) {
    const bool shape_ok = p.flags == (p.flags & (AA_GEMM_A_T | AA_GEMM_B_N)) && !p.bias && p.act == AA_ACT_NONE && p.M % BM == 0 &&
                          p.N % BN == 0 && (p.ldc & 7) == 0;
    const bool plain = shape_ok && !p.residual;
    const bool resid = shape_ok && p.residual && !a_t && !b_n && (p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0;
    const bool pers = g4_persist() && p.K >= 2 * BK;
    { static long abl = -1; if (abl < 0) { const char* e = getenv("AA_GEMM_ABLATE"); abl = e ? atol(e) : 0; } p.grp_strideC = abl; }   // timing experiments only
    if (plain) return pers ? launch4_layout<1, true>(p, a_t, b_n, st) : launch4_layout<1, false>(p, a_t, b_n, st);
    if (resid) return pers ? launch4_layout<2, true>(p, a_t, b_n, st) : launch4_layout<2, false>(p, a_t, b_n, st);
    return launch4_layout<0, false>(p, a_t, b_n, st);
}

// Fused epilogues (p.fuse = AA_FUSE_*).  Returns 1 when the shape does not qualify (the caller then runs the unfused pair of kernels):
// M, N multiples of the 256 tile, 16-byte aligned rows everywhere, and per mode: ROPE head_dim 128 with rope_cols a multiple of 128;
// GLU_FWD F a multiple of 128 (N = 2F); GLU_BWD N = F.
int aa_gemm4_fused(GemmParams& p, hipStream_t st) {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (p.M % BM || p.N % BN || p.K % BK || p.K < BK || (p.ldc & 7) || p.bias || p.residual || p.act != AA_ACT_NONE) return 1;
    p.tiles_m = p.M / BM;
    p.tiles_n = p.N / BN;
    const bool pers = g4_persist() && p.K >= 2 * BK;
    if (p.fuse == AA_FUSE_ROPE) {
        if (!p.rope_pos || !al16(p.rope_cos) || !al16(p.rope_sin) || p.rope_cols % 128 || p.rope_cols < 0 || p.rope_cols > p.N) return 1;
        return pers ? launch4_layout<3, true>(p, false, false, st) : launch4_layout<3, false>(p, false, false, st);
    }
    if (p.fuse == AA_FUSE_GLU_FWD) {
        if (p.glu_f % 128 || p.N != 2 * p.glu_f || !p.aux || !al16(p.aux) || (p.ldaux & 7) || (p.glu_f & 7)) return 1;
        return pers ? launch4_layout<4, true>(p, false, false, st) : launch4_layout<4, false>(p, false, false, st);
    }
    if (p.fuse == AA_FUSE_GLU_BWD) {
        if (p.N != p.glu_f || !p.aux || !p.aux_in || !al16(p.aux) || !al16(p.aux_in) || (p.ldaux & 7) || (p.ldaux_in & 7) || (p.glu_f & 7)) return 1;
        return pers ? launch4_layout<5, true>(p, false, true, st) : launch4_layout<5, false>(p, false, true, st);
    }
    return 1;
}
