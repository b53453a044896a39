// Flash-style fused attention (forward + backward) for gfx950, bf16 I/O, fp32 softmax/accumulation.
//
// Replaces torch SDPA inside hf:models/llama/modeling_llama.py:243-281 (causal, additive padding mask
// for LEFT-padded rows), hf:models/clip/modeling_clip.py:289 (non-causal, 577 tokens, head_dim 64) and
// hf:models/opt/modeling_opt.py attention, which `model(**batch).logits` executes in the reference
// (align_anything/trainers/text_to_text/dpo.py:128).
//
// Layout: token-major activations [N*T, ld] (the fused qkv GEMM output is consumed in place),
// head h of a row lives at columns [h*HD, (h+1)*HD).  Key j of sequence n is valid iff
// j >= start[n] (left padding) and j < T; causal additionally j <= query index.  Fully masked
// query rows (pad positions) produce 0 and lse = -inf; they never influence valid rows.
//
// MFMA mapping (64-lane waves, v_mfma_f32_16x16x32_bf16), chosen so softmax state is lane-local:
//   S^T[kv][q] = K * Q^T      -> lane owns query q = lane&15, kv = 16*kb + 4*(lane>>4) + r
//   O^T[d][q] += V^T * P^T    -> the bf16-packed S^T accumulators ARE the B fragment (contraction index
//                                permuted consistently on the V side via ds_read_b64_tr_b16)
// K/V (fwd, dQ) and Q/dO (dK/dV) tiles are streamed HBM->LDS by global_load_lds DMA, double buffered,
// with a 32-byte-unit XOR swizzle that is conflict-free for both ds_read_b128 and the transpose read.
//
// Scheduling notes (round 2, from the device assembly): hipcc treats the transpose-read builtin as a reader of ANY LDS byte,
// so it put `s_waitcnt vmcnt(0)` -- a full drain of the next tile's DMA -- in front of the first transpose read of every tile,
// i.e. the prefetch only overlapped half a tile.  The transpose reads are therefore issued as inline asm (TrPipe: groups of
// fragments one group ahead of the MFMAs that consume them, own lgkmcnt wait tied to the fragment registers), which the
// compiler does not order against the DMA; the tile's own data is complete since the barrier that ended the previous tile.
// Mask handling is a template parameter of the tile body (two instantiations, chosen per tile by a wave-uniform branch) instead
// of a branch per score, and the dK/dV kernel's per-row statistics arrive by DMA as well instead of through registers.
#include "aa_common.h"

#include <type_traits>
#include <utility>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_p;

#define LOG2E_F 1.4426950408889634f
#ifndef TR_NBUF
#define TR_NBUF 4      // transposed-read groups per wave: TR_NBUF - 1 in flight ahead of the MFMAs (2 reads -> 2 MFMAs per group)
#endif
#ifndef AA_ATTN_XCD_LOCAL
#define AA_ATTN_XCD_LOCAL 1   // 0 (lab builds): kv heads fastest, the round-1 order
#endif
#ifndef AA_ATTN_XCD_LOCAL_FWD
#define AA_ATTN_XCD_LOCAL_FWD 0   // measured: the forward is 5 % SLOWER with the XCD-local order (load balance), the two backward kernels 2 % faster
#endif
#ifndef AA_ATTN_PK
#define AA_ATTN_PK 0
#endif
#ifndef TR_NBUF_KV
#define TR_NBUF_KV 4   // the same for the dK/dV kernel (4 reads -> 2 MFMAs per group)
#endif
#define LN2_F 0.6931471805599453f
// Lab builds of the backward kernels (tools/build_bwd_lab.sh: -DAA_BWD_LAB=<mask>; TIMING ONLY, results are wrong): bit 0 no softmax VALU,
// bit 1 no in-loop DMA (stale tiles), bit 2 no S / dP MFMAs, bit 3 no dV / dK (dQ) MFMAs, bit 4 no per-tile wait + barrier.  0 = the shipped kernels
// (every switch is `if constexpr`: the default build's ISA does not change).  AA_BWD_LAB_ONLY: 1 = launch only the dQ kernel, 2 = only dK/dV.
// What they measured: profiles/r04_attn_bwd_anatomy.txt.
#ifndef AA_BWD_LAB
#define AA_BWD_LAB 0
#endif
// delta = rowsum(dO * O): 1 = computed by the dQ kernel's prologue from the dO fragments it holds anyway (+ one read of its O rows) and written for the dK/dV
// kernel that follows on the stream -- same partial sums in the same association as attn_delta_kernel (bit-identical), one launch and one pass over dO less
// per backward; 0 = the separate kernel (same-box A/B)
#ifndef AA_ATTN_DELTA_IN_DQ
#define AA_ATTN_DELTA_IN_DQ 1
#endif
#ifndef AA_BWD_LAB_ONLY
#define AA_BWD_LAB_ONLY 0
#endif
// Wave priority inside a tile (round 3, same-box A/B in profiles/r03_attention_lab.txt; outputs bit-identical): a forward wave raises its priority for the
// softmax (VALU) segment between the two MFMA clusters, so the SIMD's other wave -- mid-way through ITS MFMA cluster -- cannot starve the exps and
// the wave gets back to feeding the matrix pipe sooner: forward 402 -> 382 us (-4.8 %) at level 3 (-3 % at level 1; around the MFMA clusters instead
// -3.4 %).  The backward kernels do not gain from either placement (+1 %): off there.
#ifndef AA_ATTN_SETPRIO
#define AA_ATTN_SETPRIO 2     // forward: 0 = off, 1 = s_setprio around the MFMA clusters of a tile, 2 = around the softmax VALU (default)
#endif
#ifndef AA_ATTN_SETPRIO_BWD
#define AA_ATTN_SETPRIO_BWD 0 // the same for the dQ and dK/dV kernels
#endif
#ifndef AA_ATTN_PRIO_LEVEL
#define AA_ATTN_PRIO_LEVEL 3
#endif
#define AT_STR2(X) #X
#define AT_STR(X) AT_STR2(X)
#define AT_SETPRIO_ON() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_setprio " AT_STR(AA_ATTN_PRIO_LEVEL) ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define AT_SETPRIO_OFF() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_setprio 0" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define AT_PRIO(MODE, WHICH, ON) do { if constexpr ((MODE) == (WHICH)) { if constexpr (ON) AT_SETPRIO_ON(); else AT_SETPRIO_OFF(); } } while (0)
// forward
#define AT_PRIO_MFMA(X) AT_PRIO(AA_ATTN_SETPRIO, 1, X)
#define AT_PRIO_VALU(X) AT_PRIO(AA_ATTN_SETPRIO, 2, X)
// backward kernels
#define AT_PRIO_MFMA_B(X) AT_PRIO(AA_ATTN_SETPRIO_BWD, 1, X)
#define AT_PRIO_VALU_B(X) AT_PRIO(AA_ATTN_SETPRIO_BWD, 2, X)

template <int HD> __device__ __forceinline__ int unit_swz(int row) {
    if constexpr (HD == 128) return row & 7; else return (row >> 1) & 3;
}

// DMA a [ROWS][HD] bf16 tile (rows row0.. of one sequence/head) into LDS, 1 KiB per wave-instruction.  A lane's (row in tile,
// column) pair depends only on the lane, so its byte offset from the tile's first row is computed ONCE per tensor (DmaOff) and a
// piece is issued as `global_load_lds_dwordx4 voff, s[tile base]` with M0 = its LDS chunk: no vector arithmetic per piece (the
// 64-bit per-lane addresses cost 6 VALU a piece, two of them quarter-rate integer multiplies: 53 of the ~300 VALU instructions
// of a forward tile).  Tiles that reach past the sequence's last row (ragged T) take the clamped per-lane path.
template <int HD, int ROWS, int NW>
struct DmaLane {
    static constexpr int ROW_B = HD * 2, RPI = 1024 / ROW_B, SPR = ROW_B / 16;
    static constexpr int IT = (ROWS / RPI) / NW;
    static_assert(IT >= 1, "tile too small");
    struct Off { int v[IT]; };
    int wave, lane;
    __device__ __forceinline__ void init(int wave_, int lane_) { wave = wave_; lane = lane_; }
    __device__ __forceinline__ int row_of(int j) const { return (wave + j * NW) * RPI + lane / SPR; }
    __device__ __forceinline__ int col_of(int j) const {
        const int s = lane % SPR;
        return (((s >> 1) ^ unit_swz<HD>(row_of(j))) << 4) + (s & 1) * 8;
    }
    // lane-constant byte offsets of this lane's IT pieces inside a tile of a tensor with leading dimension ld (elements)
    __device__ __forceinline__ Off offsets(long ld) const {
        Off o;
#pragma unroll
        for (int j = 0; j < IT; ++j) o.v[j] = (row_of(j) * (int)ld + col_of(j)) * 2;
        return o;
    }
    // lds: LDS byte address of the tile (wave-uniform)
    __device__ __forceinline__ void issue(const bf16_t* gbase, long ld, const Off& off, int row0, int max_row, int lds) const {
        if (row0 + ROWS <= max_row) {
            const bf16_t* tile = gbase + (long)row0 * ld;          // wave-uniform: lives in an SGPR pair
#pragma unroll
            for (int j = 0; j < IT; ++j)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(lds + (wave + j * NW) * 1024), "v"(off.v[j]), "s"(tile) : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int gr = min(row0 + row_of(j), max_row - 1);
                const bf16_t* src = gbase + (long)gr * ld + col_of(j);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(lds + (wave + j * NW) * 1024), "v"(src) : "memory");
            }
        }
    }
};

// 16-byte fragment read: tile row `row`, 16-B slot index `slot16` (8 bf16 along HD)
template <int HD>
__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int slot16) {
    const int unit = (slot16 >> 1) ^ unit_swz<HD>(row);
    return *reinterpret_cast<const bf16x8*>(tile + row * (HD * 2) + unit * 32 + (slot16 & 1) * 16);
}
// transpose read of the 4x16 block rows rbase..rbase+3, cols 16*db..: lane (i=lane&15) gets
// tile[rbase + j][16*db + i], j = 0..3
template <int HD>
__device__ __forceinline__ bf16x4 lds_tr(const char* tile, int rbase, int db, int l15) {
    const int row = rbase + (l15 >> 2);
    const int unit = db ^ unit_swz<HD>(row);
    const char* a = tile + row * (HD * 2) + unit * 32 + (l15 & 3) * 8;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_p)a);
}
// two transpose reads -> 8 contraction slots (g,e): e<4 -> row 16*a + 4g + e ; e>=4 -> row 16*b + 4g + e-4
template <int HD>
__device__ __forceinline__ bf16x8 lds_tr_pair(const char* tile, int a16, int b16, int db, int g, int l15) {
    const bf16x4 lo = lds_tr<HD>(tile, a16 + 4 * g, db, l15);
    const bf16x4 hi = lds_tr<HD>(tile, b16 + 4 * g, db, l15);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- compile-time loops (inline-asm immediates must be constant expressions)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- transpose reads as inline asm
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int IMM>
__device__ __forceinline__ i32x2 at_rdt(int vaddr) {
    i32x2 d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(vaddr), "i"(IMM));
    return d;
}
#define AT_PIN __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ bf16x8 at_join(const i32x2& lo, const i32x2& hi) {
    const i32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    return __builtin_bit_cast(bf16x8, v);
}
// wait for every outstanding LDS read of this wave; the fragments pass through the statement so no consumer can be moved above it
__device__ __forceinline__ void at_wait(i32x2& a, i32x2& b, i32x2& c, i32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// "this register value is needed HERE": makes hipcc put the wait for a prologue load in front of the tile loop.  Left to itself it waits
// at the first use INSIDE the loop -- `s_waitcnt vmcnt(3..0)` in every iteration, which (the counter being shared) drains the DMA
// prefetch of the next tile a quarter into the current one.
__device__ __forceinline__ void landed(const bf16x8& v) {
    const i32x4 t = __builtin_bit_cast(i32x4, v);
    asm volatile("" ::"v"(t));
}
__device__ __forceinline__ void landed(float v) { asm volatile("" ::"v"(v)); }
template <int NR> struct TrBuf { i32x2 f[NR]; };
// counted form: returns once at most LEFT LDS operations of this wave are outstanding (LDS returns in order, so everything
// issued before the youngest LEFT reads has landed); b's registers pass through so their consumers stay below
template <int LEFT>
__device__ __forceinline__ void at_wait_buf(TrBuf<4>& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b.f[0]), "+v"(b.f[1]), "+v"(b.f[2]), "+v"(b.f[3]) : "i"(LEFT));
}
template <int LEFT>
__device__ __forceinline__ void at_wait_buf(TrBuf<8>& b) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(b.f[0]), "+v"(b.f[1]), "+v"(b.f[2]), "+v"(b.f[3]), "+v"(b.f[4]), "+v"(b.f[5]), "+v"(b.f[6]), "+v"(b.f[7])
                 : "i"(LEFT));
}
template <int LEFT>
__device__ __forceinline__ void at_wait_buf(TrBuf<2>& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b.f[0]), "+v"(b.f[1]) : "i"(LEFT));
}
// Lane part of a transpose-read address inside a [64][HD] tile: row 4g + (l15 >> 2) (further multiples of 16 rows and the tile's
// offset go into the instruction's immediate), 8-byte slot l15 & 3, and the row's swizzle already XORed into the 32-byte-unit bits
// (5.. of the byte offset, zero in every other term), so that d block db is reached with one more XOR:
//     address(db, rows16) = (tile + tr_lane_base) ^ (db << 5)  + rows16 * 16 * HD * 2        -- the bytes lds_tr() addresses
template <int HD>
__device__ __forceinline__ int tr_lane_base(int g, int l15) {
    const int row = 4 * g + (l15 >> 2);
    return row * (HD * 2) + (unit_swz<HD>(row) << 5) + (l15 & 3) * 8;
}
// Streams every transposed fragment of one or two (NT) [64][HD] tiles to `consume(S, D, frag0[, frag1])`: S = 32-row half (the
// contraction slots 32S .. 32S+31 of the tile), D = 16-wide d block, frag = the bf16x8 MFMA operand lds_tr_pair() would return.
// The reads are inline asm, NBUF - 1 groups (one (S, D) pair = 2 NT reads) ahead of their consumers; OFF0 / OFF1 = byte offsets
// of the tiles from `base` (= LDS address of the stage + tr_lane_base).
template <int HD, int NT, int OFF0, int OFF1, int NBUF, typename F>
__device__ __forceinline__ void tr_stream(const int base, F&& consume) {
    constexpr int ROWB = HD * 2, DB = HD / 16, NG = 2 * DB, NR = 2 * NT;
    static_assert(NR * (NBUF - 2) <= 15 && NBUF >= 2 && NBUF - 1 <= NG, "lgkmcnt is a 4-bit counter");
    TrBuf<NR> tb[NBUF];
    auto issue = [&](auto gi) {
        constexpr int G = decltype(gi)::value, S = G / DB, D = G % DB;
        const int a = base ^ (D << 5);
        TrBuf<NR>& b = tb[G % NBUF];
        b.f[0] = at_rdt<OFF0 + S * 32 * ROWB>(a);
        b.f[1] = at_rdt<OFF0 + (S * 32 + 16) * ROWB>(a);
        if constexpr (NT == 2) {
            b.f[2] = at_rdt<OFF1 + S * 32 * ROWB>(a);
            b.f[3] = at_rdt<OFF1 + (S * 32 + 16) * ROWB>(a);
        }
    };
    static_for<NBUF - 1>(issue);
    static_for<NG>([&](auto gi) {
        constexpr int G = decltype(gi)::value, S = G / DB, D = G % DB;
        constexpr int LEFT = (NBUF - 2 < NG - 1 - G ? NBUF - 2 : NG - 1 - G) * NR;   // reads of younger groups that may stay in flight
        TrBuf<NR>& b = tb[G % NBUF];
        at_wait_buf<LEFT>(b);
        if constexpr (G + NBUF - 1 < NG) issue(std::integral_constant<int, G + NBUF - 1>{});
        if constexpr (NT == 2)
            consume(std::integral_constant<int, S>{}, std::integral_constant<int, D>{}, at_join(b.f[0], b.f[1]), at_join(b.f[2], b.f[3]));
        else
            consume(std::integral_constant<int, S>{}, std::integral_constant<int, D>{}, at_join(b.f[0], b.f[1]));
    });
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x4& a, const f32x4& b) {
    // 2-wide converts lower to v_cvt_pk_bf16_f32 (one instruction per pair)
    const bf16x2 p0 = __builtin_convertvector(f32x2{a[0], a[1]}, bf16x2);
    const bf16x2 p1 = __builtin_convertvector(f32x2{a[2], a[3]}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector(f32x2{b[0], b[1]}, bf16x2);
    const bf16x2 p3 = __builtin_convertvector(f32x2{b[2], b[3]}, bf16x2);
    const bf16x4 lo = __builtin_shufflevector(p0, p1, 0, 1, 2, 3);
    const bf16x4 hi = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Epilogue store of a wave's [16 rows (lane & 15)][16 DB columns] fp32 block (lane holds columns 16 db + 4 (lane >> 4) + 0..3 of its row) as bf16 with
// 16-BYTE stores: lanes g and g ^ 1 exchange halves with v_permlane16_swap (the gemm4 epilogue's trick), so a lane ends up with 8 consecutive columns
// of one 16-column block -- half the store instructions of the 4-columns-per-lane form (the store tail of a workgroup is issue-bound), one packed
// convert per dword.  Every lane must call it (the swaps are wave-wide); `ok` masks the store only.
// Same-box A/B (profiles/r03_attention_lab.txt, run 6): the backward pair gains 2.2 % (1211 -> 1185 us), the FORWARD kernel loses 6 % (388 -> 413 us) with
// the wide form -- it sits at its register limit -- so WIDE is a template parameter: dQ and dK/dV use it, the forward keeps the 8-byte stores.
template <int DB, bool WIDE>
__device__ __forceinline__ void at_store_rows(bf16_t* row, const f32x4 (&acc)[DB], float scale, bool ok, int g) {
    if constexpr (WIDE) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
    auto pk2 = [](float a, float b) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    };
    const int ecol = (g & 1) * 16 + (g >> 1) * 8;
#pragma unroll
    for (int db = 0; db < DB; db += 2) {
        unsigned w[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) w[f][h] = pk2(acc[db + f][2 * h] * scale, acc[db + f][2 * h + 1] * scale);
        const auto lo = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
        if (ok) *reinterpret_cast<u32x4_*>(row + db * 16 + ecol) = u32x4_{lo[0], hi[0], lo[1], hi[1]};
    }
    } else {
    if (!ok) return;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        u16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[db][r] * scale);
        *reinterpret_cast<u16x4*>(row + db * 16 + g * 4) = o;
    }
    }
}

// at_store_rows<DB, true> with aa_rope_inplace(inverse = 1) applied on the way out (backward of hf apply_rotary_pos_emb on the bf16 gradient, the rounding
// points of elementwise.hip::rope_kernel: o1 = bf16(bf16(a c) + bf16(b s)), o2 = bf16(bf16(b c) + bf16(-a s)) on the bf16-ROUNDED dQ / dK values): the lane
// that owns 8 columns d .. d + 7 of block pair (db, db + 1) also owns their partners d + HD / 2 in blocks (db + DB / 2, ...), so the rotation is
// register-local; `tab` = byte-free offset rope_pos[row] * (HD / 2) into the cos / sin tables.  Removes the separate rope launch over d[q | k].
template <int DB>
__device__ __forceinline__ void at_store_rows_rope(bf16_t* row, const f32x4 (&acc)[DB], bool ok, int g, const bf16_t* cos_t, const bf16_t* sin_t, long tab) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
    auto pk2 = [](float a, float b) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); };
    auto lo16 = [](unsigned x) { return __builtin_bit_cast(float, x << 16); };
    auto hi16 = [](unsigned x) { return __builtin_bit_cast(float, x & 0xffff0000u); };
    auto r16 = [](float x) { return bf2f(f2bf(x)); };
    const int ecol = (g & 1) * 16 + (g >> 1) * 8;
    auto packed = [&](int db) -> u32x4_ {
        unsigned w[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) w[f][h] = pk2(acc[db + f][2 * h], acc[db + f][2 * h + 1]);
        const auto lo = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
        return u32x4_{lo[0], hi[0], lo[1], hi[1]};
    };
#pragma unroll
    for (int db = 0; db < DB / 2; db += 2) {
        u32x4_ x1 = packed(db), x2 = packed(db + DB / 2);
        if (ok) {
            const u32x4_ c = *reinterpret_cast<const u32x4_*>(cos_t + tab + db * 16 + ecol);
            const u32x4_ sn = *reinterpret_cast<const u32x4_*>(sin_t + tab + db * 16 + ecol);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = lo16(x1[e]), a1 = hi16(x1[e]), b0 = lo16(x2[e]), b1 = hi16(x2[e]);
                const float c0 = lo16(c[e]), c1 = hi16(c[e]), s0 = lo16(sn[e]), s1 = hi16(sn[e]);
                x1[e] = pk2(r16(a0 * c0) + r16(b0 * s0), r16(a1 * c1) + r16(b1 * s1));
                x2[e] = pk2(r16(b0 * c0) + r16(-a0 * s0), r16(b1 * c1) + r16(-a1 * s1));
            }
            *reinterpret_cast<u32x4_*>(row + db * 16 + ecol) = x1;
            *reinterpret_cast<u32x4_*>(row + (db + DB / 2) * 16 + ecol) = x2;
        }
    }
}

// Reductions over the four 16-lane rows of a wave (the lanes that share l15): v_permlane16_swap / v_permlane32_swap exchange
// rows (halves) between two registers, so with both operands = v the two results hold the row pair's members in every lane and
// one max / add finishes the step -- 2 VALU per step instead of __shfl_xor's index arithmetic + ds_bpermute round trip
// (~7 VALU and an LDS wait per step).  Row step first, half step second: the association of the sums __shfl_xor(16), (32) gave.
__device__ __forceinline__ float row4_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
}
__device__ __forceinline__ float row4_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    // the adds are asm as well: written in C++, hipcc (ROCm 7.2) emitted r[0] + r[0] for a swap of a value with itself
    float m;
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
}
// the same with the half step FIRST: the association of __shfl_xor(32) followed by __shfl_xor(16), which is what attn_delta_kernel's butterfly does to the four
// 8-column slots a row's four lanes hold (the dQ kernel's in-prologue delta must reproduce that kernel's bits)
__device__ __forceinline__ float row4_sum_half_first(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    float m;
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
}
// v_exp_f32 without the denormal-range fix-up of exp2f(): arguments here are <= 0 and results that would be
// denormal contribute nothing to a softmax sum
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnParams {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    const bf16_t* dO; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    float* lse;          // [N, H, T] natural-log LSE of the scaled scores
    float* delta;        // [N, H, T] rowsum(dO * O)
    const int* start;    // [N] first valid key (left padding) or null
    const int* kvlen;    // [N] number of valid keys (right padding: keys >= kvlen[n] are masked) or null
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int N, T, H, Hkv, causal;
    float scale;
    // backward only: the transpose rotary rotation of dQ / dK in the kernels' epilogues (aa_attn_bwd_rope): position of token row r = rope_pos[r],
    // cos / sin tables [., HD / 2] bf16; null = plain dQ / dK (aa_attn_bwd)
    const int* rope_pos; const bf16_t* rope_cos; const bf16_t* rope_sin;
    // optional (shared-prompt packing): query rows below qskip[n] of sequence n have no consumer -- the caller never reads their O / lse and hands the backward
    // dO = 0 for them.  Whole query blocks below it are not computed (forward), get dQ = 0 without being visited (dQ kernel) and are left out of the dK / dV loop
    const int* qskip;
    int pair;            // a128 forward only: a workgroup runs the query blocks nqb - 1 - r and r (causal load balance, see attn128.inc)
};

// ------------------------------------------------------------------ workgroup -> (sequence, head, block)
// Block b runs on XCD b % 8 and every XCD has its own 4 MB L2.  All blocks that stream the SAME K/V (forward, dQ: the query
// blocks of the query heads sharing one kv head) or the same Q/dO (dK/dV: the key blocks of one kv head) are therefore given to
// ONE XCD, back to back in its dispatch sequence (b / 8), so that they run at the same time and the head's rows are fetched from
// HBM once and served to the others by that L2; with heads fastest (round 1) the 64 blocks resident on an XCD belonged to 64
// different heads and every one of them streamed its K/V from memory alone: the PMC counters showed 2.3 GB of L2-miss reads per
// forward launch for 0.4 GB of operands, i.e. the kernels were bound by HBM/MALL bandwidth, not by the MFMAs.  Heaviest blocks of
// a head first (causal); heads are dealt to the XCDs round-robin, which needs Hkv * N % 8 == 0 -- otherwise kv heads fastest.
// Measured (profiles/r02_attention_lab.txt): L2-miss reads drop 5-6x (forward 2.3 GB -> 0.4 GB per launch) but the kernels are
// NOT bound by that traffic -- the backward pair gains 2 %, the forward loses 5 % to the coarser load balance -- so the
// forward keeps kv heads fastest and only the backward kernels use the XCD-local order (less HBM power next to the GEMMs).
template <bool XCD_LOCAL>
__device__ __forceinline__ void q_block_of(const AttnParams& p, int nqb, int& n, int& h, int& hk, int& qb) {
    const int HkN = p.Hkv * p.N, group = p.H / p.Hkv, per = nqb * group;
    const int b = blockIdx.x;
    int hkn, r;
    if (XCD_LOCAL && AA_ATTN_XCD_LOCAL && (HkN & 7) == 0) { hkn = (b & 7) + 8 * ((b >> 3) / per); r = (b >> 3) % per; }
    else                { hkn = b % HkN; r = b / HkN; }
    n = hkn / p.Hkv; hk = hkn % p.Hkv;
    h = hk * group + r % group;
    qb = nqb - 1 - r / group;
}
// a128 forward with p.pair: block b -> (sequence, query head, pair index r in [0, npair)); all npair * group workgroups that stream one kv head's K / V sit
// back to back in ONE XCD's dispatch sequence when the kv heads can be dealt to the 8 XCDs evenly, else kv heads fastest (any order is balanced: equal pairs)
__device__ __forceinline__ void pair_block_of(const AttnParams& p, int npair, int& n, int& h, int& hk, int& r) {
    const int HkN = p.Hkv * p.N, group = p.H / p.Hkv, per = npair * group;
    const int b = blockIdx.x;
    int hkn, rr;
    if ((HkN & 7) == 0) { hkn = (b & 7) + 8 * ((b >> 3) / per); rr = (b >> 3) % per; }
    else                { hkn = b % HkN; rr = b / HkN; }
    n = hkn / p.Hkv; hk = hkn % p.Hkv;
    h = hk * group + rr % group;
    r = rr / group;
}
__device__ __forceinline__ void kv_block_of(const AttnParams& p, int nkvb, int& n, int& hk, int& kvb) {
    const int HkN = p.Hkv * p.N;
    const int b = blockIdx.x;
    int hkn;
    if (AA_ATTN_XCD_LOCAL && (HkN & 7) == 0) { hkn = (b & 7) + 8 * ((b >> 3) / nkvb); kvb = (b >> 3) % nkvb; }
    else                { hkn = b % HkN; kvb = b / HkN; }
    n = hkn / p.Hkv; hk = hkn % p.Hkv;
}

// ================================================================== forward
// 1-D grid (q_block_of).  256 threads: wave w owns queries q0 + 16 QI w .. + 16 QI - 1 (QI blocks of 16 query rows; AA_ATTN_QI, lab: 4 = 64 rows per wave,
// one workgroup of 256 query rows per CU, one wave per SIMD -- half the LDS fragment bytes and half the tile steps per query row; row results do not depend on QI)
#ifndef AA_ATTN_QI
#define AA_ATTN_QI 2
#endif
template <int HD>
__global__ __launch_bounds__(256, AA_ATTN_QI == 2 ? 2 : 1) void attn_fwd_kernel(const AttnParams p) {
    constexpr int QI = AA_ATTN_QI, QROWS = 64 * QI;
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (p.T + QROWS - 1) / QROWS;
    int n, h, hk, qb;
    q_block_of<AA_ATTN_XCD_LOCAL_FWD>(p, nqb, n, h, hk, qb);
    const int q0 = qb * QROWS, qw = q0 + wave * 16 * QI;
    if (p.qskip && q0 + QROWS <= p.qskip[n]) return;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Qb = p.Q + (long)n * T * p.ldq + h * HD;
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    DmaLane<HD, 64, 4> dma;
    dma.init(wave, lane);
    const auto koff = dma.offsets(p.ldk), voff = dma.offsets(p.ldv);
    const int lds0 = (int)(uintptr_t)smem;                   // LDS byte address of the dynamic segment
    const int trl = tr_lane_base<HD>(g, l15);                // lane part of the transposed-read addresses

    // Q fragments (B operand of S^T): lane -> query l15, d = ks*32 + g*8 ..+7
    bf16x8 qf[QI][KS];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const int qr = min(qw + qi * 16 + l15, T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qi][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * p.ldq + ks * 32 + g * 8);
    }
    f32x4 oacc[QI][DB];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi)
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[qi][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m2[QI], lsum[QI];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) { m2[qi] = -INFINITY; lsum[qi] = 0.f; }

    const int kv_begin = (start / 64) * 64;
    const int kv_end = p.causal ? min(T, q0 + QROWS) : T;
    const int ntile = (kv_end - kv_begin + 63) / 64;
    if (ntile > 0) {
        dma.issue(Kb, p.ldk, koff, kv_begin, T, lds0);
        dma.issue(Vb, p.ldv, voff, kv_begin, T, lds0 + TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qi = 0; qi < QI; ++qi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) landed(qf[qi][ks]);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int kv0 = kv_begin + t * 64;
        if (t + 1 < ntile) {
            dma.issue(Kb, p.ldk, koff, kv0 + 64, T, lds0 + (cur ^ 1) * 2 * TILE_B);
            dma.issue(Vb, p.ldv, voff, kv0 + 64, T, lds0 + (cur ^ 1) * 2 * TILE_B + TILE_B);
        }
        const char* kt = smem + cur * 2 * TILE_B;
        const char* vt = kt + TILE_B;
        // wave-uniform skip: every key of this tile is after every query of this wave (causal)
        const bool wave_active = !(p.causal && kv0 > qw + 16 * QI - 1) && (qw < T);
        if (wave_active) {
            f32x4 sacc[QI][4];
#pragma unroll
            for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) sacc[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
            AT_PRIO_MFMA(1);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = lds_frag<HD>(kt, kb * 16 + l15, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < QI; ++qi)
                        sacc[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], sacc[qi][kb], 0, 0, 0);
                }
            AT_PRIO_MFMA(0);
            AT_PRIO_VALU(1);
            // masks only where a mask can bite: diagonal tile, left-pad boundary, ragged end (wave-uniform)
            const bool need_mask = (p.causal && kv0 + 63 > qw) || kv0 < start || kv0 + 64 > KT;
            bf16x8 pf[QI][2];
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                float mx = -INFINITY;
                if (need_mask) {
                    const int qg = qw + qi * 16 + l15;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kv = kv0 + kb * 16 + g * 4 + r;
                            const bool ok = kv >= start && kv < KT && (!p.causal || kv <= qg);
                            const float s = ok ? sacc[qi][kb][r] * c2 : -INFINITY;
                            sacc[qi][kb][r] = s;
                            mx = fmaxf(mx, s);
                        }
                } else {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = sacc[qi][kb][r] * c2;
                            sacc[qi][kb][r] = s;
                            mx = fmaxf(mx, s);
                        }
                }
                mx = row4_max(mx);
                const float mn = fmaxf(m2[qi], mx);
                const float ms = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = fast_exp2(m2[qi] - ms);
                m2[qi] = mn;
                float ps = 0.f;
#if AA_ATTN_PK
                // packed form (lab): the 16 subtractions and the 16 additions as 8 + 8 v_pk_add_f32; the row sum becomes
                // (even lanes' sum) + (odd lanes' sum), i.e. a different association than the sequential sum below
                {
                    const f32x2 ms2 = {ms, ms};
                    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const f32x2 x = f32x2{sacc[qi][kb][2 * hh], sacc[qi][kb][2 * hh + 1]} - ms2;
                            const f32x2 e = {fast_exp2(x[0]), fast_exp2(x[1])};
                            sacc[qi][kb][2 * hh] = e[0];
                            sacc[qi][kb][2 * hh + 1] = e[1];
                            acc2 += e;
                        }
                    ps = acc2[0] + acc2[1];
                }
#else
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = fast_exp2(sacc[qi][kb][r] - ms);
                        sacc[qi][kb][r] = pe;
                        ps += pe;
                    }
#endif
                lsum[qi] = lsum[qi] * alpha + ps;
                // exact skip of the O rescale while the running max is unchanged for the whole wave
                if (!__all(alpha == 1.f)) {
#pragma unroll
                    for (int db = 0; db < DB; ++db) oacc[qi][db] *= alpha;
                }
                pf[qi][0] = pack_bf16x8(sacc[qi][0], sacc[qi][1]);
                pf[qi][1] = pack_bf16x8(sacc[qi][2], sacc[qi][3]);
            }
            // O^T += V^T P^T: the transposed V fragments come by inline asm (tr_stream), a few groups ahead of their MFMAs
            AT_PRIO_VALU(0);
            AT_PRIO_MFMA(1);
            tr_stream<HD, 1, TILE_B, 0, TR_NBUF>(lds0 + cur * 2 * TILE_B + trl, [&](auto si, auto di, const bf16x8 vf) {
                constexpr int S = decltype(si)::value, D = decltype(di)::value;
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    oacc[qi][D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qi][S], oacc[qi][D], 0, 0, 0);
            });
            AT_PRIO_MFMA(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: O[q][16db + 4g + r] = oacc / l
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const float l = row4_sum(lsum[qi]);
        const int qg = qw + qi * 16 + l15;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        at_store_rows<DB, false>(p.O + ((long)n * T + min(qg, T - 1)) * p.ldo + h * HD, oacc[qi], inv, qg < T, g);
        if (qg >= T) continue;
        if (g == 0 && p.lse)
            p.lse[((long)n * p.H + h) * T + qg] = l > 0.f ? (m2[qi] + log2f(l)) * LN2_F : -INFINITY;
    }
}

// ================================================================== delta = rowsum(dO * O)
// one (HD/8)-lane group per (token row, head)
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnParams p) {
    constexpr int LPG = HD / 8;  // lanes per (row, head)
    const long total = (long)p.N * p.T * p.H;
    const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / LPG;
    const int sub = threadIdx.x % LPG;
    if (gid >= total) return;
    const int h = (int)(gid % p.H);
    const long row = gid / p.H;  // n*T + t
    const u16x8 a = *reinterpret_cast<const u16x8*>(p.dO + row * p.lddo + h * HD + sub * 8);
    const u16x8 b = *reinterpret_cast<const u16x8*>(p.O + row * p.ldo + h * HD + sub * 8);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(b[j]);
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) {
        const long n = row / p.T, t = row % p.T;
        p.delta[(n * p.H + h) * p.T + t] = s;
    }
}

// ================================================================== backward: dQ
// same structure / block order as forward; dQ^T[d][q] += K^T[d][kv] * dS^T[kv][q]
// NW waves of 32 query rows share one K / V tile stream (4 shipped; 8 was measured, see attn_bwd_impl).
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2, QROWS = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (p.T + QROWS - 1) / QROWS;
    int n, h, hk, qb;
    q_block_of<true>(p, nqb, n, h, hk, qb);
    const int q0 = qb * QROWS, qw = q0 + wave * 32;
    const int T = p.T;
    if (p.qskip && q0 + QROWS <= p.qskip[n]) {      // nobody consumes these query rows (dO = 0): dQ = 0, no visit
        for (int i = threadIdx.x; i < QROWS * (HD / 8); i += 64 * NW) {
            const int r = q0 + i / (HD / 8);
            if (r < T) *reinterpret_cast<f32x4*>(p.dQ + ((long)n * T + r) * p.lddq + h * HD + (i % (HD / 8)) * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Qb = p.Q + (long)n * T * p.ldq + h * HD;
    const bf16_t* dOb = p.dO + (long)n * T * p.lddo + h * HD;
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    DmaLane<HD, 64, NW> dma;
    dma.init(wave, lane);
    const auto koff = dma.offsets(p.ldk), voff = dma.offsets(p.ldv);
    const int lds0 = (int)(uintptr_t)smem;
    const int trl = tr_lane_base<HD>(g, l15);

    bf16x8 qf[2][KS], dof[2][KS];
    float lse2[2], dl[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qr = min(qw + qi * 16 + l15, T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qi][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * p.ldq + ks * 32 + g * 8);
            dof[qi][ks] = *reinterpret_cast<const bf16x8*>(dOb + (long)qr * p.lddo + ks * 32 + g * 8);
        }
        lse2[qi] = p.lse[((long)n * p.H + h) * T + qr] * LOG2E_F;
        if constexpr (AA_ATTN_DELTA_IN_DQ) {
            // attn_delta_kernel's arithmetic: 8 products per 8-column slot `sub` = 4 ks + g, then the butterfly over the slots (xor 8, 4 inside the lane; xor 2, 1 =
            // lanes 32 and 16 apart)
            const bf16_t* Ob = p.O + (long)n * T * p.ldo + h * HD;
            float sp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u16x8 ov = *reinterpret_cast<const u16x8*>(Ob + (long)qr * p.ldo + ks * 32 + g * 8);
                const u16x8 dv = __builtin_bit_cast(u16x8, dof[qi][ks]);
                float sacc_ = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sacc_ += bf2f(dv[j]) * bf2f(ov[j]);
                sp[ks] = sacc_;
            }
            float dsum;
            if constexpr (KS == 4) dsum = (sp[0] + sp[2]) + (sp[1] + sp[3]); else dsum = sp[0] + sp[1];
            dsum = row4_sum_half_first(dsum);
            dl[qi] = dsum;
            if (g == 0 && qw + qi * 16 + l15 < T) p.delta[((long)n * p.H + h) * T + qr] = dsum;
        } else {
            dl[qi] = p.delta[((long)n * p.H + h) * T + qr];
        }
    }
    f32x4 dqacc[2][DB];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int db = 0; db < DB; ++db) dqacc[qi][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kv_begin = (start / 64) * 64;
    const int kv_end = p.causal ? min(T, q0 + QROWS) : T;
    const int ntile = (kv_end - kv_begin + 63) / 64;
    if (ntile > 0) {
        dma.issue(Kb, p.ldk, koff, kv_begin, T, lds0);
        dma.issue(Vb, p.ldv, voff, kv_begin, T, lds0 + TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { landed(qf[qi][ks]); landed(dof[qi][ks]); }
        landed(lse2[qi]); landed(dl[qi]);
    }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int kv0 = kv_begin + t * 64;
        if (t + 1 < ntile && !(AA_BWD_LAB & 2)) {
            dma.issue(Kb, p.ldk, koff, kv0 + 64, T, lds0 + (cur ^ 1) * 2 * TILE_B);
            dma.issue(Vb, p.ldv, voff, kv0 + 64, T, lds0 + (cur ^ 1) * 2 * TILE_B + TILE_B);
        }
        const char* kt = smem + cur * 2 * TILE_B;
        const char* vt = kt + TILE_B;
        const bool wave_active = !(p.causal && kv0 > qw + 31) && (qw < T);
        if (wave_active) {
            const bool need_mask = (p.causal && kv0 + 63 > qw) || kv0 < start || kv0 + 64 > KT || qw + 32 > T;
            bf16x8 dsf[2][2];
            // S, dP and dS = P * (dP - delta) * scale, one 32-key half of the tile at a time: only 2 x 2 score blocks per operand
            // are live at once (the register file holds dQ, Q and dO fragments as well).  The masked instantiation runs only on
            // the diagonal / padding / ragged tiles.
            auto half = [&](auto hi, auto masked) {
                constexpr int S = decltype(hi)::value;
                f32x4 sacc[2][2], dpacc[2][2];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        sacc[qi][kk] = f32x4{0.f, 0.f, 0.f, 0.f};
                        dpacc[qi][kk] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf16x8 kf = lds_frag<HD>(kt, (2 * S + kk) * 16 + l15, ks * 4 + g);
                        const bf16x8 vf = lds_frag<HD>(vt, (2 * S + kk) * 16 + l15, ks * 4 + g);
#pragma unroll
                        for (int qi = 0; qi < 2; ++qi) {
                            if constexpr (AA_BWD_LAB & 4) { landed(kf); landed(vf); continue; }
                            sacc[qi][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], sacc[qi][kk], 0, 0, 0);
                            dpacc[qi][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qi][ks], dpacc[qi][kk], 0, 0, 0);
                        }
                    }
                AT_PRIO_VALU_B(1);
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const int qg = qw + qi * 16 + l15;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (AA_BWD_LAB & 1) { sacc[qi][kk][r] += dpacc[qi][kk][r]; continue; }
                            float pe = fast_exp2(sacc[qi][kk][r] * c2 - lse2[qi]);
                            if constexpr (decltype(masked)::value) {
                                const int kv = kv0 + (2 * S + kk) * 16 + g * 4 + r;
                                const bool ok = kv >= start && kv < KT && (!p.causal || kv <= qg) && qg < T;
                                pe = ok ? pe : 0.f;
                            }
                            sacc[qi][kk][r] = pe * (dpacc[qi][kk][r] - dl[qi]) * p.scale;
                        }
                    dsf[qi][S] = pack_bf16x8(sacc[qi][0], sacc[qi][1]);
                }
                AT_PRIO_VALU_B(0);
            };
            if (need_mask) {
                half(std::integral_constant<int, 0>{}, std::true_type{});
                AT_PIN;
                half(std::integral_constant<int, 1>{}, std::true_type{});
            } else {
                half(std::integral_constant<int, 0>{}, std::false_type{});
                AT_PIN;
                half(std::integral_constant<int, 1>{}, std::false_type{});
            }
            // dQ^T += K^T dS^T: transposed K fragments by inline asm (see the forward's PV step)
            AT_PRIO_MFMA_B(1);
            tr_stream<HD, 1, 0, 0, TR_NBUF>(lds0 + cur * 2 * TILE_B + trl, [&](auto si, auto di, const bf16x8 ktf) {
                constexpr int S = decltype(si)::value, D = decltype(di)::value;
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    if constexpr (AA_BWD_LAB & 8) { landed(ktf); landed(dsf[qi][S]); continue; }
                    dqacc[qi][D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qi][S], dqacc[qi][D], 0, 0, 0);
                }
            });
            AT_PRIO_MFMA_B(0);
        }
        if constexpr (!(AA_BWD_LAB & 16)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qg = qw + qi * 16 + l15;
        const long qrow = (long)n * T + min(qg, T - 1);
        if (p.rope_pos) at_store_rows_rope<DB>(p.dQ + qrow * p.lddq + h * HD, dqacc[qi], qg < T, g, p.rope_cos, p.rope_sin, (long)p.rope_pos[qrow] * (HD / 2));
        else at_store_rows<DB, true>(p.dQ + qrow * p.lddq + h * HD, dqacc[qi], 1.f, qg < T, g);
    }
}

// ================================================================== backward: dK, dV
// 1-D grid (kv_block_of): the key blocks of a kv head ascending (block 0 sees every query tile under a causal mask: heaviest
// first).  Wave w owns keys kv0 + 16w .. +15 and loops over 64-query tiles (and over the H/Hkv query heads sharing this kv head).
//   S[q][kv] = Q K^T, dP[q][kv] = dO V^T          (lane: kv = lane&15, q = 16qb + 4g + r)
//   dV^T[d][kv] += dO^T[d][q] P[q][kv] ; dK^T[d][kv] += Q^T[d][q] dS[q][kv]
// NW waves of 16 keys share one Q / dO tile stream.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2, KVB = 16 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][Q tile | dO tile] + [2][64 lse | 64 delta]
    float* stat = reinterpret_cast<float*>(smem + 4 * TILE_B);
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = p.H / p.Hkv;
    int n, hk, kvb;
    kv_block_of(p, (p.T + KVB - 1) / KVB, n, hk, kvb);
    const int kv0 = kvb * KVB, kvw = kv0 + wave * 16;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    const int kvg = kvw + l15;
    DmaLane<HD, 64, NW> dma;
    dma.init(wave, lane);
    const auto qoff = dma.offsets(p.ldq), dooff = dma.offsets(p.lddo);
    const int lds0 = (int)(uintptr_t)smem;
    const int trl = tr_lane_base<HD>(g, l15);

    bf16x8 kf[KS], vf[KS];
    {
        const int kr = min(kvg, T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(Kb + (long)kr * p.ldk + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(Vb + (long)kr * p.ldv + ks * 32 + g * 8);
        }
    }
    f32x4 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { dkacc[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    int q_begin = p.causal ? (kv0 / 64) * 64 : 0;
    if (p.qskip) q_begin = max(q_begin, (p.qskip[n] / 64) * 64);      // query tiles wholly below qskip carry dO = 0 (and unspecified lse): left out
    const int ntq = max(0, (T - q_begin + 63) / 64);
    const int total = ntq * group;  // iteration = (head in group, q tile)
    const bool kv_valid_block = kv0 + KVB - 1 >= start;  // some key of this block can be attended

    // The tile's 64 LSE and 64 delta values travel by DMA as well (wave 0 / wave 1, one dword per lane): a register round trip
    // would put `s_waitcnt vmcnt(0)` -- the whole prefetch -- in front of the ds_write at the top of every iteration.
    auto issue = [&](int it, int buf) {
        const int hh = hk * group + it / ntq;
        const int qt0 = q_begin + (it % ntq) * 64;
        const bf16_t* Qb = p.Q + (long)n * T * p.ldq + hh * HD;
        const bf16_t* dOb = p.dO + (long)n * T * p.lddo + hh * HD;
        dma.issue(Qb, p.ldq, qoff, qt0, T, lds0 + buf * 2 * TILE_B);
        dma.issue(dOb, p.lddo, dooff, qt0, T, lds0 + buf * 2 * TILE_B + TILE_B);
        if (wave < 2) {
            const long idx = ((long)n * p.H + hh) * T + min(qt0 + lane, T - 1);
            const float* src = (wave == 0 ? p.lse : p.delta) + idx;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                         :: "s"(lds0 + 4 * TILE_B + buf * 512 + wave * 256), "v"(src) : "memory");
        }
    };

    if (total > 0 && kv_valid_block) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { landed(kf[ks]); landed(vf[ks]); }
    __syncthreads();
    for (int it = 0; it < total && kv_valid_block; ++it) {
        const int cur = it & 1;
        if (it + 1 < total && !(AA_BWD_LAB & 2)) issue(it + 1, cur ^ 1);
        const int qt0 = q_begin + (it % ntq) * 64;
        const char* qt = smem + cur * 2 * TILE_B;
        const char* dot = qt + TILE_B;
        const float* st = stat + cur * 128;
        const bool wave_active = kvw < T && !(p.causal && kvw > qt0 + 63);
        if (wave_active) {
            f32x4 sacc[4], dpacc[4];
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) { sacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            AT_PRIO_MFMA_B(1);
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = lds_frag<HD>(qt, qb * 16 + l15, ks * 4 + g);
                    const bf16x8 da = lds_frag<HD>(dot, qb * 16 + l15, ks * 4 + g);
                    if constexpr (AA_BWD_LAB & 4) { landed(qa); landed(da); continue; }
                    sacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], sacc[qb], 0, 0, 0);
                    dpacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dpacc[qb], 0, 0, 0);
                }
            AT_PRIO_MFMA_B(0);
            AT_PRIO_VALU_B(1);
            const bool need_mask = (p.causal && qt0 < kvw + 15) || kvw < start || kvw + 16 > KT || qt0 + 64 > T;
            bf16x8 pfr[2], dsfr[2];
            // P and dS of the 64 x 16 block; the rows' statistics come as 16-byte LDS reads (q = 16 qb + 4 g + r)
            auto softmax_bwd = [&](auto masked) {
                f32x4 pv[4], dsv[4];
#pragma unroll
                for (int qb = 0; qb < 4; ++qb) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(st + qb * 16 + g * 4);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(st + 64 + qb * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (AA_BWD_LAB & 1) { pv[qb][r] = sacc[qb][r] + l4[r]; dsv[qb][r] = dpacc[qb][r] + d4[r]; continue; }
                        float l2;       // lse * log2(e) rounded on its own (as when it was pre-scaled into LDS), never contracted into the fma below
                        asm("v_mul_f32 %0, %1, %2" : "=v"(l2) : "v"(l4[r]), "v"(LOG2E_F));
                        float pe = fast_exp2(sacc[qb][r] * c2 - l2);
                        if constexpr (decltype(masked)::value) {
                            const int qg = qt0 + qb * 16 + g * 4 + r;
                            const bool ok = kvg >= start && kvg < KT && qg < T && (!p.causal || kvg <= qg);
                            pe = ok ? pe : 0.f;
                        }
                        pv[qb][r] = pe;
                        dsv[qb][r] = pe * (dpacc[qb][r] - d4[r]) * p.scale;
                    }
                }
                pfr[0] = pack_bf16x8(pv[0], pv[1]); pfr[1] = pack_bf16x8(pv[2], pv[3]);
                dsfr[0] = pack_bf16x8(dsv[0], dsv[1]); dsfr[1] = pack_bf16x8(dsv[2], dsv[3]);
            };
            if (need_mask) softmax_bwd(std::true_type{}); else softmax_bwd(std::false_type{});
            // dV^T += dO^T P, dK^T += Q^T dS: the transposed Q / dO fragments by inline asm (tr_stream)
            AT_PRIO_VALU_B(0);
            AT_PRIO_MFMA_B(1);
            tr_stream<HD, 2, 0, TILE_B, TR_NBUF_KV>(lds0 + cur * 2 * TILE_B + trl, [&](auto si, auto di, const bf16x8 qt_f, const bf16x8 dot_f) {
                constexpr int S = decltype(si)::value, D = decltype(di)::value;
                if constexpr (AA_BWD_LAB & 8) { landed(dot_f); landed(qt_f); landed(pfr[S]); landed(dsfr[S]); return; }
                dvacc[D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pfr[S], dvacc[D], 0, 0, 0);
                dkacc[D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsfr[S], dkacc[D], 0, 0, 0);
            });
            AT_PRIO_MFMA_B(0);
        }
        if constexpr (!(AA_BWD_LAB & 16)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const long krow_i = (long)n * T + min(kvg, T - 1);
    if (p.rope_pos) at_store_rows_rope<DB>(p.dK + krow_i * p.lddk + hk * HD, dkacc, kvg < T, g, p.rope_cos, p.rope_sin, (long)p.rope_pos[krow_i] * (HD / 2));
    else at_store_rows<DB, true>(p.dK + krow_i * p.lddk + hk * HD, dkacc, 1.f, kvg < T, g);
    at_store_rows<DB, true>(p.dV + ((long)n * T + min(kvg, T - 1)) * p.lddv + hk * HD, dvacc, 1.f, kvg < T, g);
}

#include "attn128.inc"

// ================================================================== C ABI
static int check_common(const char* fn, int N, int T, int H, int Hkv, int hd) {
    if (!(N > 0 && T > 0 && H > 0 && Hkv > 0 && H % Hkv == 0)) {
        aa_set_error("%s: bad shape N=%d T=%d H=%d Hkv=%d", fn, N, T, H, Hkv);
        return AA_ERR_ARG;
    }
    if (hd != 64 && hd != 128) {
        aa_set_error("%s: head_dim %d not built (64 and 128 are)", fn, hd);
        return AA_ERR_ARG;
    }
    return AA_OK;
}

template <typename K>
static int set_lds(K kern, int bytes, const char* name) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        aa_set_error("%s: cannot reserve %d B LDS: %s", name, bytes, hipGetErrorString(e));
        return AA_ERR_LAUNCH;
    }
    return AA_OK;
}

// head_dim 128 runs on the one-wave-per-SIMD 32x32x16 kernels of attn128.inc; AA_ATTN128=0 / aa_attn_set_impl(0) keeps the 16x16x32 kernels above
// (same-box A/B, bisecting; both stay tested).  Bit 0: forward, bit 1: backward, bit 2 (round 5, default on): the forward's workgroups run PAIRS of query blocks
// in the XCD-local order (attn128.inc; bit-identical outputs; in the step 405.8 -> 386.8 us and 1.34 -> 0.59 GB fetched per launch, profiles/r05_attn_fwd_pair.txt).
// aa_ctx::attn_impl (-1: read AA_ATTN128 once)
static int attn_impl() {
    if (aa_ctx_cur()->attn_impl < 0) { const char* e = getenv("AA_ATTN128"); aa_ctx_cur()->attn_impl = e ? atoi(e) : 7; }
    return aa_ctx_cur()->attn_impl;
}
extern "C" int aa_attn_set_impl(int impl) {
    AA_REQUIRE(impl >= 0 && impl <= 7, "aa_attn_set_impl: %d (bit 0 = forward, bit 1 = backward on the 32x32x16 kernels, bit 2 = paired query blocks)", impl);
    aa_ctx_cur()->attn_impl = impl;
    return AA_OK;
}
static bool attn128_enabled() { return (attn_impl() & 1) != 0; }

static int attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, float* lse,
                         const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, int N, int T,
                         int H, int Hkv, int hd, int causal, float scale, const int* q_skip, void* stream) {
    int rc = check_common("aa_attn_fwd", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE((ldq | ldk | ldv | ldo) % 8 == 0, "aa_attn_fwd: leading dims must be multiples of 8");
    AttnParams p{};
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.lse = lse; p.start = start; p.kvlen = kv_len; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal; p.scale = scale; p.qskip = q_skip;
    dim3 grid(aa_cdiv(T, 64 * AA_ATTN_QI) * H * N);
    const int lds = 4 * 64 * hd * 2;
    if (hd == 128 && attn128_enabled()) {
        if ((rc = set_lds(a128::attn128_fwd_kernel<128>, a128::LDS_FWD, "aa_attn_fwd"))) return rc;
        const int nqb = aa_cdiv(T, 256);
        p.pair = (causal && (nqb & 1) == 0 && (attn_impl() & 4)) ? 1 : 0;      // bit 2 of AA_ATTN128 / aa_attn_set_impl: paired query blocks, XCD-local
        hipLaunchKernelGGL(a128::attn128_fwd_kernel<128>, dim3((p.pair ? nqb / 2 : nqb) * H * N), dim3(256), a128::LDS_FWD, (hipStream_t)stream, p);
    } else if (hd == 128) {
        if ((rc = set_lds(attn_fwd_kernel<128>, lds, "aa_attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        if ((rc = set_lds(attn_fwd_kernel<64>, lds, "aa_attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    AA_CHECK_LAUNCH("aa_attn_fwd");
    return AA_OK;
}

extern "C" int aa_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse,
                           const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, int N, int T,
                           int H, int Hkv, int hd, int causal, float scale, void* stream) {
    return attn_fwd_impl(Q, K, V, O, lse, start, kv_len, ldq, ldk, ldv, ldo, N, T, H, Hkv, hd, causal, scale, nullptr, stream);
}
// aa_attn_fwd with q_skip[N]: query rows below q_skip[n] have no consumer (shared-prompt packing: the rejected row's copy of the pair's common prefix) -- whole
// query blocks below it are not computed and their O / lse rows stay unwritten.
extern "C" int aa_attn_fwd_qskip(const void* Q, const void* K, const void* V, void* O, float* lse,
                                 const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, int N, int T,
                                 int H, int Hkv, int hd, int causal, float scale, const int* q_skip, void* stream) {
    AA_REQUIRE(q_skip != nullptr, "aa_attn_fwd_qskip: q_skip is required (aa_attn_fwd is the form without it)");
    return attn_fwd_impl(Q, K, V, O, lse, start, kv_len, ldq, ldk, ldv, ldo, N, T, H, Hkv, hd, causal, scale, q_skip, stream);
}

static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                         const float* lse, float* delta, void* dQ, void* dK, void* dV,
                         const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                         long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                         int causal, float scale, const int* rope_pos, const void* rope_cos, const void* rope_sin, void* stream, const int* q_skip = nullptr) {
    int rc = check_common("aa_attn_bwd", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 8 == 0,
               "aa_attn_bwd: leading dims must be multiples of 8");
    AttnParams p{};
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.dO = (const bf16_t*)dO; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV;
    p.lse = const_cast<float*>(lse); p.delta = delta; p.start = start; p.kvlen = kv_len;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    p.rope_pos = rope_pos; p.rope_cos = (const bf16_t*)rope_cos; p.rope_sin = (const bf16_t*)rope_sin; p.qskip = q_skip;
    hipStream_t st = (hipStream_t)stream;
    const long groups = (long)N * T * H;
    const int lds = 4 * 64 * hd * 2;
    if (hd == 128) {
        // NW = 8 (one 8-wave workgroup per CU sharing the tile stream) measured 1356 us on the bench block against ~1200-1340 for two independent 4-wave
        // workgroups per CU: what it saves in LDS-DMA pieces it loses in overlap across the per-tile barrier (profiles/r04_attn128_anatomy.txt, section 6)
        const dim3 gq(aa_cdiv(T, 128) * H * N), gkv(aa_cdiv(T, 64) * Hkv * N);
        if (!AA_ATTN_DELTA_IN_DQ || AA_BWD_LAB_ONLY == 2) hipLaunchKernelGGL(attn_delta_kernel<128>, dim3(aa_cdiv(groups * 16, 256)), dim3(256), 0, st, p);
        if ((rc = set_lds(attn_bwd_dq_kernel<128, 4>, lds, "aa_attn_bwd"))) return rc;
        if (AA_BWD_LAB_ONLY != 2) hipLaunchKernelGGL((attn_bwd_dq_kernel<128, 4>), gq, dim3(256), lds, st, p);
        if (AA_BWD_LAB_ONLY == 1) { AA_CHECK_LAUNCH("aa_attn_bwd"); return AA_OK; }
        if ((rc = set_lds(attn_bwd_dkv_kernel<128, 4>, lds + 1024, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 4>), gkv, dim3(256), lds + 1024, st, p);
    } else {
        const dim3 gq(aa_cdiv(T, 128) * H * N), gkv(aa_cdiv(T, 64) * Hkv * N);
        if (!AA_ATTN_DELTA_IN_DQ) hipLaunchKernelGGL(attn_delta_kernel<64>, dim3(aa_cdiv(groups * 8, 256)), dim3(256), 0, st, p);
        if ((rc = set_lds(attn_bwd_dq_kernel<64, 4>, lds, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 4>), gq, dim3(256), lds, st, p);
        if ((rc = set_lds(attn_bwd_dkv_kernel<64, 4>, lds + 1024, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, 4>), gkv, dim3(256), lds + 1024, st, p);
    }
    AA_CHECK_LAUNCH("aa_attn_bwd");
    return AA_OK;
}
extern "C" int aa_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                           const float* lse, float* delta, void* dQ, void* dK, void* dV,
                           const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                           long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                           int causal, float scale, void* stream) {
    return attn_bwd_impl(Q, K, V, O, dO, lse, delta, dQ, dK, dV, start, kv_len, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, N, T, H, Hkv, hd, causal, scale,
                         nullptr, nullptr, nullptr, stream);
}
// aa_attn_bwd followed by aa_rope_inplace(inverse = 1) on dQ and dK (the backward of the rotary embedding the forward applied to q / k), in the kernels'
// epilogues: pos[N * T] = rotary position of every token row, cos_t / sin_t [., hd / 2] bf16.  Bit-identical to the two-kernel form.
extern "C" int aa_attn_bwd_rope(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                const float* lse, float* delta, void* dQ, void* dK, void* dV,
                                const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                                long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                                int causal, float scale, const int* pos, const void* cos_t, const void* sin_t, void* stream) {
    AA_REQUIRE(pos != nullptr && cos_t != nullptr && sin_t != nullptr, "aa_attn_bwd_rope: pos / cos_t / sin_t are required");
    AA_REQUIRE(((uintptr_t)cos_t & 15) == 0 && ((uintptr_t)sin_t & 15) == 0, "aa_attn_bwd_rope: tables must be 16-byte aligned");
    return attn_bwd_impl(Q, K, V, O, dO, lse, delta, dQ, dK, dV, start, kv_len, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, N, T, H, Hkv, hd, causal, scale,
                         pos, cos_t, sin_t, stream);
}
// aa_attn_bwd / aa_attn_bwd_rope (pos / cos_t / sin_t all NULL or all given) with q_skip[N] as in aa_attn_fwd_qskip: the caller guarantees dO = 0 for the
// query rows below q_skip[n]; whole query blocks below it get dQ = 0 without being visited and are left out of the dK / dV accumulation.
extern "C" int aa_attn_bwd_qskip(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                 const float* lse, float* delta, void* dQ, void* dK, void* dV,
                                 const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                                 long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                                 int causal, float scale, const int* pos, const void* cos_t, const void* sin_t, const int* q_skip, void* stream) {
    AA_REQUIRE(q_skip != nullptr, "aa_attn_bwd_qskip: q_skip is required");
    AA_REQUIRE((pos == nullptr) == (cos_t == nullptr) && (pos == nullptr) == (sin_t == nullptr), "aa_attn_bwd_qskip: pos / cos_t / sin_t together or not at all");
    AA_REQUIRE(pos == nullptr || ((((uintptr_t)cos_t | (uintptr_t)sin_t) & 15) == 0), "aa_attn_bwd_qskip: tables must be 16-byte aligned");
    AA_REQUIRE(AA_ATTN_DELTA_IN_DQ, "aa_attn_bwd_qskip: built without the in-kernel delta");
    return attn_bwd_impl(Q, K, V, O, dO, lse, delta, dQ, dK, dV, start, kv_len, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, N, T, H, Hkv, hd, causal, scale,
                         pos, cos_t, sin_t, stream, q_skip);
}
