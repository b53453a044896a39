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
#include "aa_common.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_p;

#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

template <int HD> __device__ __forceinline__ int unit_swz(int row) {
    if constexpr (HD == 128) return row & 7; else return (row >> 1) & 3;
}

// DMA a [ROWS][HD] bf16 tile (rows row0.. of one sequence/head, clamped to max_row-1) into LDS.
// The per-lane (row-in-tile, column) pairs depend only on the lane: computed once (DmaLane), so a tile
// issue costs one min + one 64-bit mad per 1-KiB piece.
template <int HD, int ROWS, int NW>
struct DmaLane {
    static constexpr int ROW_B = HD * 2, RPI = 1024 / ROW_B, SPR = ROW_B / 16;
    static constexpr int IT = (ROWS / RPI) / NW;
    static_assert(IT >= 1, "tile too small");
    int r[IT], col[IT], chunk[IT];
    __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int c = wave + j * NW;
            r[j] = c * RPI + lane / SPR;
            const int s = lane % SPR;
            col[j] = (((s >> 1) ^ unit_swz<HD>(r[j])) << 4) + (s & 1) * 8;
            chunk[j] = c * 1024;
        }
    }
    __device__ __forceinline__ void issue(const bf16_t* gbase, long ld, int row0, int max_row, char* lds) const {
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int gr = min(row0 + r[j], max_row - 1);
            const bf16_t* src = gbase + (long)gr * ld + col[j];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + chunk[j]), 16, 0, 0);
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
};

// ================================================================== forward
// 1-D grid, heaviest (latest) query blocks first and heads fastest, so the 8 XCDs (block b -> XCD b % 8) get
// equal causal work and short blocks fill the tail.  256 threads: wave w owns queries q0 + 32w .. +31.
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HN = p.H * p.N;
    const int nqb = (p.T + 127) / 128;
    const int hn = blockIdx.x % HN;
    const int qb = nqb - 1 - blockIdx.x / HN;
    const int n = hn / p.H, h = hn % p.H, hk = h / (p.H / p.Hkv);
    const int q0 = qb * 128, qw = q0 + wave * 32;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Qb = p.Q + (long)n * T * p.ldq + h * HD;
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    DmaLane<HD, 64, 4> dma;
    dma.init(wave, lane);

    // Q fragments (B operand of S^T): lane -> query l15, d = ks*32 + g*8 ..+7
    bf16x8 qf[2][KS];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qr = min(qw + qi * 16 + l15, T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qi][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * p.ldq + ks * 32 + g * 8);
    }
    f32x4 oacc[2][DB];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[qi][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m2[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};

    const int kv_begin = (start / 64) * 64;
    const int kv_end = p.causal ? min(T, q0 + 128) : T;
    const int ntile = (kv_end - kv_begin + 63) / 64;
    if (ntile > 0) {
        dma.issue(Kb, p.ldk, kv_begin, T, smem);
        dma.issue(Vb, p.ldv, kv_begin, T, smem + TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int kv0 = kv_begin + t * 64;
        if (t + 1 < ntile) {
            dma.issue(Kb, p.ldk, kv0 + 64, T, smem + (cur ^ 1) * 2 * TILE_B);
            dma.issue(Vb, p.ldv, kv0 + 64, T, smem + (cur ^ 1) * 2 * TILE_B + TILE_B);
        }
        const char* kt = smem + cur * 2 * TILE_B;
        const char* vt = kt + TILE_B;
        // wave-uniform skip: every key of this tile is after every query of this wave (causal)
        const bool wave_active = !(p.causal && kv0 > qw + 31) && (qw < T);
        if (wave_active) {
            f32x4 sacc[2][4];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) sacc[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = lds_frag<HD>(kt, kb * 16 + l15, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi)
                        sacc[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], sacc[qi][kb], 0, 0, 0);
                }
            // masks only where a mask can bite: diagonal tile, left-pad boundary, ragged end (wave-uniform)
            const bool need_mask = (p.causal && kv0 + 63 > qw) || kv0 < start || kv0 + 64 > KT;
            bf16x8 pf[2][2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
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
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m2[qi], mx);
                const float ms = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = fast_exp2(m2[qi] - ms);
                m2[qi] = mn;
                float ps = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = fast_exp2(sacc[qi][kb][r] - ms);
                        sacc[qi][kb][r] = pe;
                        ps += pe;
                    }
                lsum[qi] = lsum[qi] * alpha + ps;
                // exact skip of the O rescale while the running max is unchanged for the whole wave
                if (!__all(alpha == 1.f)) {
#pragma unroll
                    for (int db = 0; db < DB; ++db) oacc[qi][db] *= alpha;
                }
                pf[qi][0] = pack_bf16x8(sacc[qi][0], sacc[qi][1]);
                pf[qi][1] = pack_bf16x8(sacc[qi][2], sacc[qi][3]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const bf16x8 vf = lds_tr_pair<HD>(vt, s * 32, s * 32 + 16, db, g, l15);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi)
                        oacc[qi][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qi][s], oacc[qi][db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: O[q][16db + 4g + r] = oacc / l
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        float l = lsum[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qg = qw + qi * 16 + l15;
        if (qg >= T) continue;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16_t* orow = p.O + ((long)n * T + qg) * p.ldo + h * HD;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(oacc[qi][db][r] * inv);
            *reinterpret_cast<u16x4*>(orow + db * 16 + g * 4) = o;
        }
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
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HN = p.H * p.N;
    const int nqb = (p.T + 127) / 128;
    const int hn = blockIdx.x % HN;
    const int qb = nqb - 1 - blockIdx.x / HN;
    const int n = hn / p.H, h = hn % p.H, hk = h / (p.H / p.Hkv);
    const int q0 = qb * 128, qw = q0 + wave * 32;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Qb = p.Q + (long)n * T * p.ldq + h * HD;
    const bf16_t* dOb = p.dO + (long)n * T * p.lddo + h * HD;
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    DmaLane<HD, 64, 4> dma;
    dma.init(wave, lane);

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
        dl[qi] = p.delta[((long)n * p.H + h) * T + qr];
    }
    f32x4 dqacc[2][DB];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int db = 0; db < DB; ++db) dqacc[qi][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kv_begin = (start / 64) * 64;
    const int kv_end = p.causal ? min(T, q0 + 128) : T;
    const int ntile = (kv_end - kv_begin + 63) / 64;
    if (ntile > 0) {
        dma.issue(Kb, p.ldk, kv_begin, T, smem);
        dma.issue(Vb, p.ldv, kv_begin, T, smem + TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const int kv0 = kv_begin + t * 64;
        if (t + 1 < ntile) {
            dma.issue(Kb, p.ldk, kv0 + 64, T, smem + (cur ^ 1) * 2 * TILE_B);
            dma.issue(Vb, p.ldv, kv0 + 64, T, smem + (cur ^ 1) * 2 * TILE_B + TILE_B);
        }
        const char* kt = smem + cur * 2 * TILE_B;
        const char* vt = kt + TILE_B;
        const bool wave_active = !(p.causal && kv0 > qw + 31) && (qw < T);
        if (wave_active) {
            f32x4 sacc[2][4], dpacc[2][4];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    sacc[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dpacc[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = lds_frag<HD>(kt, kb * 16 + l15, ks * 4 + g);
                    const bf16x8 vf = lds_frag<HD>(vt, kb * 16 + l15, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) {
                        sacc[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], sacc[qi][kb], 0, 0, 0);
                        dpacc[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qi][ks], dpacc[qi][kb], 0, 0, 0);
                    }
                }
            const bool need_mask = (p.causal && kv0 + 63 > qw) || kv0 < start || kv0 + 64 > KT || qw + 32 > T;
            bf16x8 dsf[2][2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                const int qg = qw + qi * 16 + l15;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pe = fast_exp2(sacc[qi][kb][r] * c2 - lse2[qi]);
                        if (need_mask) {
                            const int kv = kv0 + kb * 16 + g * 4 + r;
                            const bool ok = kv >= start && kv < KT && (!p.causal || kv <= qg) && qg < T;
                            pe = ok ? pe : 0.f;
                        }
                        sacc[qi][kb][r] = pe * (dpacc[qi][kb][r] - dl[qi]) * p.scale;
                    }
                dsf[qi][0] = pack_bf16x8(sacc[qi][0], sacc[qi][1]);
                dsf[qi][1] = pack_bf16x8(sacc[qi][2], sacc[qi][3]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const bf16x8 ktf = lds_tr_pair<HD>(kt, s * 32, s * 32 + 16, db, g, l15);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi)
                        dqacc[qi][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qi][s], dqacc[qi][db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qg = qw + qi * 16 + l15;
        if (qg >= T) continue;
        bf16_t* row = p.dQ + ((long)n * T + qg) * p.lddq + h * HD;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(dqacc[qi][db][r]);
            *reinterpret_cast<u16x4*>(row + db * 16 + g * 4) = o;
        }
    }
}

// ================================================================== backward: dK, dV
// 1-D grid: kv block = blockIdx / (Hkv*N) ascending (block 0 sees every query tile under a causal mask: heaviest
// first), kv heads fastest.  Wave w owns keys kv0 + 16w .. +15 and loops over 64-query tiles (and over the
// H/Hkv query heads sharing this kv head).
//   S[q][kv] = Q K^T, dP[q][kv] = dO V^T          (lane: kv = lane&15, q = 16qb + 4g + r)
//   dV^T[d][kv] += dO^T[d][q] P[q][kv] ; dK^T[d][kv] += Q^T[d][q] dS[q][kv]
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DB = HD / 16;
    constexpr int TILE_B = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][Q tile | dO tile] + [2][64 lse | 64 delta]
    float* stat = reinterpret_cast<float*>(smem + 4 * TILE_B);
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HkN = p.Hkv * p.N;
    const int hn = blockIdx.x % HkN;
    const int n = hn / p.Hkv, hk = hn % p.Hkv;
    const int group = p.H / p.Hkv;
    const int kv0 = (blockIdx.x / HkN) * 64, kvw = kv0 + wave * 16;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    const int kvg = kvw + l15;
    DmaLane<HD, 64, 4> dma;
    dma.init(wave, lane);

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

    const int q_begin = p.causal ? (kv0 / 64) * 64 : 0;
    const int ntq = (T - q_begin + 63) / 64;
    const int total = ntq * group;  // iteration = (head in group, q tile)
    const bool kv_valid_block = kv0 + 63 >= start;  // some key of this block can be attended

    auto issue = [&](int it, int buf) {
        const int hh = hk * group + it / ntq;
        const int qt0 = q_begin + (it % ntq) * 64;
        const bf16_t* Qb = p.Q + (long)n * T * p.ldq + hh * HD;
        const bf16_t* dOb = p.dO + (long)n * T * p.lddo + hh * HD;
        dma.issue(Qb, p.ldq, qt0, T, smem + buf * 2 * TILE_B);
        dma.issue(dOb, p.lddo, qt0, T, smem + buf * 2 * TILE_B + TILE_B);
        if (threadIdx.x < 128) {
            const int i = threadIdx.x & 63;
            const int qr = min(qt0 + i, T - 1);
            const long idx = ((long)n * p.H + hh) * T + qr;
            stat[buf * 128 + threadIdx.x] = (threadIdx.x < 64) ? p.lse[idx] * LOG2E_F : p.delta[idx];
        }
    };

    if (total > 0 && kv_valid_block) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < total && kv_valid_block; ++it) {
        const int cur = it & 1;
        if (it + 1 < total) issue(it + 1, cur ^ 1);
        const int qt0 = q_begin + (it % ntq) * 64;
        const char* qt = smem + cur * 2 * TILE_B;
        const char* dot = qt + TILE_B;
        const float* st = stat + cur * 128;
        const bool wave_active = kvw < T && !(p.causal && kvw > qt0 + 63);
        if (wave_active) {
            f32x4 sacc[4], dpacc[4];
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) { sacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qa = lds_frag<HD>(qt, qb * 16 + l15, ks * 4 + g);
                    const bf16x8 da = lds_frag<HD>(dot, qb * 16 + l15, ks * 4 + g);
                    sacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], sacc[qb], 0, 0, 0);
                    dpacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dpacc[qb], 0, 0, 0);
                }
            const bool need_mask = (p.causal && qt0 < kvw + 15) || kvw < start || kvw + 16 > KT || qt0 + 64 > T;
            bf16x8 pfr[2], dsfr[2];
            f32x4 pv[4], dsv[4];
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = qb * 16 + g * 4 + r;
                    float pe = fast_exp2(sacc[qb][r] * c2 - st[ql]);
                    if (need_mask) {
                        const int qg = qt0 + ql;
                        const bool ok = kvg >= start && kvg < KT && qg < T && (!p.causal || kvg <= qg);
                        pe = ok ? pe : 0.f;
                    }
                    pv[qb][r] = pe;
                    dsv[qb][r] = pe * (dpacc[qb][r] - st[64 + ql]) * p.scale;
                }
            pfr[0] = pack_bf16x8(pv[0], pv[1]); pfr[1] = pack_bf16x8(pv[2], pv[3]);
            dsfr[0] = pack_bf16x8(dsv[0], dsv[1]); dsfr[1] = pack_bf16x8(dsv[2], dsv[3]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const bf16x8 dot_f = lds_tr_pair<HD>(dot, s * 32, s * 32 + 16, db, g, l15);
                    const bf16x8 qt_f = lds_tr_pair<HD>(qt, s * 32, s * 32 + 16, db, g, l15);
                    dvacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pfr[s], dvacc[db], 0, 0, 0);
                    dkacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsfr[s], dkacc[db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (kvg < T) {
        bf16_t* krow = p.dK + ((long)n * T + kvg) * p.lddk + hk * HD;
        bf16_t* vrow = p.dV + ((long)n * T + kvg) * p.lddv + hk * HD;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u16x4 a, b;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = f2bf(dkacc[db][r]); b[r] = f2bf(dvacc[db][r]); }
            *reinterpret_cast<u16x4*>(krow + db * 16 + g * 4) = a;
            *reinterpret_cast<u16x4*>(vrow + db * 16 + g * 4) = b;
        }
    }
}

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

extern "C" int aa_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse,
                           const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, int N, int T,
                           int H, int Hkv, int hd, int causal, float scale, void* stream) {
    int rc = check_common("aa_attn_fwd", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE((ldq | ldk | ldv | ldo) % 8 == 0, "aa_attn_fwd: leading dims must be multiples of 8");
    AttnParams p{};
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.lse = lse; p.start = start; p.kvlen = kv_len; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    dim3 grid(aa_cdiv(T, 128) * H * N);
    const int lds = 4 * 64 * hd * 2;
    if (hd == 128) {
        if ((rc = set_lds(attn_fwd_kernel<128>, lds, "aa_attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        if ((rc = set_lds(attn_fwd_kernel<64>, lds, "aa_attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    AA_CHECK_LAUNCH("aa_attn_fwd");
    return AA_OK;
}

extern "C" int aa_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                           const float* lse, float* delta, void* dQ, void* dK, void* dV,
                           const int* start, const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo,
                           long lddq, long lddk, long lddv, int N, int T, int H, int Hkv, int hd,
                           int causal, float scale, void* stream) {
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
    hipStream_t st = (hipStream_t)stream;
    const long groups = (long)N * T * H;
    const int lds = 4 * 64 * hd * 2;
    const dim3 gq(aa_cdiv(T, 128) * H * N), gkv(aa_cdiv(T, 64) * Hkv * N);
    if (hd == 128) {
        hipLaunchKernelGGL(attn_delta_kernel<128>, dim3(aa_cdiv(groups * 16, 256)), dim3(256), 0, st, p);
        if ((rc = set_lds(attn_bwd_dq_kernel<128>, lds, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), lds, st, p);
        if ((rc = set_lds(attn_bwd_dkv_kernel<128>, lds + 1024, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, gkv, dim3(256), lds + 1024, st, p);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<64>, dim3(aa_cdiv(groups * 8, 256)), dim3(256), 0, st, p);
        if ((rc = set_lds(attn_bwd_dq_kernel<64>, lds, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), lds, st, p);
        if ((rc = set_lds(attn_bwd_dkv_kernel<64>, lds + 1024, "aa_attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, gkv, dim3(256), lds + 1024, st, p);
    }
    AA_CHECK_LAUNCH("aa_attn_bwd");
    return AA_OK;
}
