// LAB (not in the library, never run on hardware at the time of writing): attention forward on v_mfma_f32_32x32x16_bf16 with 64 query rows per wave, ONE wave per
// SIMD -- DESIGN.md section 8 item 1 / tools/lab/README_attention_next.md item 4.  Built by tools/lab/attn_v2/build.sh INTO a lab copy of the library (this file
// includes csrc/attention.hip, so the object replaces attention.o); entry point aa_attn_fwd_v2 = aa_attn_fwd's signature; tools/lab/attn_v2/check.py compares it with
// aa_attn_fwd (tolerance: the 32 x 32 tiles add in another order) and times both.  head_dim 128 only.
//
// Layouts (lane l: l31 = l & 31, hi = l >> 5; 16-lane group of a wave = (hi, dh) with dh = (l >> 4) & 1):
//   S^T[kv][q] = K Q^T, one 32 x 32 tile per (kvb, qb2), 8 k-steps of 16 d:
//       A = K   : lane -> key row 32 kvb + l31, d = 16 ks + 8 hi .. + 7   (one ds_read_b128 of the swizzled K image, slot16 = 2 ks + hi)
//       B = Q^T : lane -> query 32 qb2 + l31, the same 8 d                (registers, loaded once from global)
//       D       : lane -> query l31; register r -> key 32 kvb + crow(r, hi), crow = (r & 3) + 8 (r >> 2) + 4 hi
//   so a query's 64 scores of a tile sit in 2 x 16 registers of lane l and of lane l ^ 32: row max / sum = 31 lane-local ops + ONE v_permlane32_swap.
//   O^T[d][q] += V^T P^T, one 32 x 32 tile per (dblk, qb2), 4 k-steps (kvb, kq) of 16 keys.  Contraction slot (hi, j) of k-step (kvb, kq) is DEFINED as key
//       32 kvb + 16 kq + 4 hi + j (j < 4)  /  32 kvb + 16 kq + 8 + 4 hi + (j - 4) (j >= 4):
//       B = P^T : exactly the lane's accumulator registers r = 8 kq .. 8 kq + 7 of S^T tile (kvb, qb2), packed to bf16 -- no cross-lane move at all
//       A = V^T : lane -> d = 32 dblk + 16 dh + l15, slots j < 4 / j >= 4 = two ds_read_b64_tr_b16 of the 4 x 16 blocks at key rows base / base + 8,
//                 base = 32 kvb + 16 kq + 4 hi, 16-column block 2 dblk + dh (the transposed read works per 16-lane group)
//       D       : lane -> query l31; register r -> d = 32 dblk + crow(r, hi)
// Per 64-key tile and wave: 32 + 32 MFMAs of 32 cycles (2048 matrix cycles), 16 ds_read_b128 + 32 ds_read_b64_tr_b16 (32 KB of LDS reads for 64 rows: half the
// bytes per row of the 32-row waves), 64 exponentials per lane.  Registers: O 128 + S 64 accumulators, Q 64, P 32 (+ addressing): one wave per SIMD (512 per lane).
#include "attention.hip"

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float at_pair32_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));      // {x[l31], x[l31 + 32]} in every lane
}
__device__ __forceinline__ float at_pair32_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ bf16x8 pack8_bf16(const f32x16& a, int r0) {      // registers r0 .. r0 + 7 (r0 compile-time after unrolling)
    const bf16x2 p0 = __builtin_convertvector(f32x2{a[r0 + 0], a[r0 + 1]}, bf16x2);
    const bf16x2 p1 = __builtin_convertvector(f32x2{a[r0 + 2], a[r0 + 3]}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector(f32x2{a[r0 + 4], a[r0 + 5]}, bf16x2);
    const bf16x2 p3 = __builtin_convertvector(f32x2{a[r0 + 6], a[r0 + 7]}, bf16x2);
    const bf16x4 lo = __builtin_shufflevector(p0, p1, 0, 1, 2, 3);
    const bf16x4 hi = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int HD>
__global__ __launch_bounds__(256, 1) void attn_fwd_v2_kernel(const AttnParams p) {
    static_assert(HD == 128, "lab kernel: head_dim 128");
    constexpr int KS = HD / 16, DBLK = HD / 32;
    constexpr int TILE_B = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, dh = (lane >> 4) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (p.T + 255) / 256;
    int n, h, hk, qb;
    q_block_of<false>(p, nqb, n, h, hk, qb);
    const int q0 = qb * 256, qw = q0 + wave * 64;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;
    const bf16_t* Qb = p.Q + (long)n * T * p.ldq + h * HD;
    const bf16_t* Kb = p.K + (long)n * T * p.ldk + hk * HD;
    const bf16_t* Vb = p.V + (long)n * T * p.ldv + hk * HD;
    const float c2 = p.scale * LOG2E_F;
    DmaLane<HD, 64, 4> dma;
    dma.init(wave, lane);
    const auto koff = dma.offsets(p.ldk), voff = dma.offsets(p.ldv);
    const int lds0 = (int)(uintptr_t)smem;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
        const int qr = min(qw + qb2 * 32 + l31, T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qb2][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * p.ldq + ks * 16 + hi * 8);
    }
    f32x16 oacc[DBLK][2];
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][qb2][r] = 0.f;
    float m2[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};      // lsum: this lane's 32 of a tile's 64 keys; the two halves meet in the epilogue

    const int kv_begin = (start / 64) * 64;
    const int kv_end = p.causal ? min(T, q0 + 256) : T;
    const int ntile = (kv_end - kv_begin + 63) / 64;
    if (ntile > 0) {
        dma.issue(Kb, p.ldk, koff, kv_begin, T, lds0);
        dma.issue(Vb, p.ldv, voff, kv_begin, T, lds0 + TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) landed(qf[qb2][ks]);
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
        const bool wave_active = !(p.causal && kv0 > qw + 63) && (qw < T);
        if (wave_active) {
            f32x16 sacc[2][2];      // [kvb][qb2]
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                for (int qb2 = 0; qb2 < 2; ++qb2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kvb][qb2][r] = 0.f;
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = lds_frag<HD>(kt, kvb * 32 + l31, ks * 2 + hi);
#pragma unroll
                    for (int qb2 = 0; qb2 < 2; ++qb2)
                        sacc[kvb][qb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb2][ks], sacc[kvb][qb2], 0, 0, 0);
                }
            const bool need_mask = (p.causal && kv0 + 63 > qw) || kv0 < start || kv0 + 64 > KT;
            bf16x8 pf[2][2][2];     // [qb2][kvb][kq]
#pragma unroll
            for (int qb2 = 0; qb2 < 2; ++qb2) {
                float mx = -INFINITY;
                const int qg = qw + qb2 * 32 + l31;
#pragma unroll
                for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float sv = sacc[kvb][qb2][r] * c2;
                        if (need_mask) {
                            const int kv = kv0 + kvb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const bool ok = kv >= start && kv < KT && (!p.causal || kv <= qg);
                            sv = ok ? sv : -INFINITY;
                        }
                        sacc[kvb][qb2][r] = sv;
                        mx = fmaxf(mx, sv);
                    }
                mx = at_pair32_max(mx);
                const float mn = fmaxf(m2[qb2], mx);
                const float ms = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = fast_exp2(m2[qb2] - ms);
                m2[qb2] = mn;
                float ps = 0.f;
#pragma unroll
                for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pe = fast_exp2(sacc[kvb][qb2][r] - ms);
                        sacc[kvb][qb2][r] = pe;
                        ps += pe;
                    }
                lsum[qb2] = lsum[qb2] * alpha + ps;
                if (!__all(alpha == 1.f)) {
#pragma unroll
                    for (int d = 0; d < DBLK; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[d][qb2][r] *= alpha;
                }
#pragma unroll
                for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                    for (int kq = 0; kq < 2; ++kq) pf[qb2][kvb][kq] = pack8_bf16(sacc[kvb][qb2], 8 * kq);
            }
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                    for (int d = 0; d < DBLK; ++d) {
                        const int base = kvb * 32 + kq * 16 + 4 * hi;
                        const bf16x4 lo = lds_tr<HD>(vt, base, 2 * d + dh, l15);
                        const bf16x4 up = lds_tr<HD>(vt, base + 8, 2 * d + dh, l15);
                        const bf16x8 vf = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int qb2 = 0; qb2 < 2; ++qb2)
                            oacc[d][qb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb2][kvb][kq], oacc[d][qb2], 0, 0, 0);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: O[q][32 d + 8 (r >> 2) + 4 hi + (r & 3)] = oacc / l
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
        const float l = at_pair32_sum(lsum[qb2]);
        const int qg = qw + qb2 * 32 + l31;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        if (qg < T) {
            bf16_t* row = p.O + ((long)n * T + qg) * p.ldo + h * HD;
#pragma unroll
            for (int d = 0; d < DBLK; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    u16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(oacc[d][qb2][rq * 4 + j] * inv);
                    *reinterpret_cast<u16x4*>(row + d * 32 + rq * 8 + hi * 4) = o;
                }
            if (hi == 0 && p.lse)
                p.lse[((long)n * p.H + h) * T + qg] = l > 0.f ? (m2[qb2] + log2f(l)) * LN2_F : -INFINITY;
        }
    }
}

extern "C" int aa_attn_fwd_v2(const void* Q, const void* K, const void* V, void* O, float* lse, const int* start, const int* kv_len, long ldq, long ldk,
                              long ldv, long ldo, int N, int T, int H, int Hkv, int hd, int causal, float scale, void* stream) {
    int rc = check_common("aa_attn_fwd_v2", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE(hd == 128, "aa_attn_fwd_v2: lab kernel, head_dim 128 only");
    AA_REQUIRE((ldq | ldk | ldv | ldo) % 8 == 0, "aa_attn_fwd_v2: leading dims must be multiples of 8");
    AttnParams p{};
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.lse = lse; p.start = start; p.kvlen = kv_len; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    dim3 grid(aa_cdiv(T, 256) * H * N);
    const int lds = 4 * 64 * hd * 2;
    if ((rc = set_lds(attn_fwd_v2_kernel<128>, lds, "aa_attn_fwd_v2"))) return rc;
    hipLaunchKernelGGL(attn_fwd_v2_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, p);
    AA_CHECK_LAUNCH("aa_attn_fwd_v2");
    return AA_OK;
}
