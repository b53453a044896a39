// fp32 GEMM for the parity mode: C[M,N] (+)= op(A) op(B) with exact-fp32 inputs on the matrix cores
// (v_mfma_f32_32x32x2_f32, fp32 accumulate), fused bias / activation / residual like aa_gemm_bf16 but without any
// bf16 rounding point.  Same layouts and flags as aa_gemm_bf16 (NT default, AA_GEMM_B_N, AA_GEMM_A_T, AA_GEMM_ACCUM);
// every operand is fp32.  128x128 tile, BK = 16, 4 waves each owning a 64x64 quadrant (2x2 MFMA blocks); operands
// are staged k-major in LDS ([k][m], row pitch 132 floats) so a fragment read is 32 consecutive floats per
// half-wave (conflict-free) whatever the global layout.  This path exists to track the reference's fp32 CPU trainer
// to 1e-4 on the loss; throughput is secondary (fp32 MFMA peak is 157 TFLOP/s).
#include "gemm_params.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int FBM = 128, FBN = 128, FBK = 16, FLD = 132;

struct GemmF32Params {
    const float* A; const float* B; float* C;
    const float* bias; const float* residual;
    int M, N, K;
    long lda, ldb, ldc, ldr;
    int act, flags;
    const int* grp_tile_expert; const int* grp_off; long grp_stride;   // grouped (mixture-of-experts) launches, see gemm.hip
};

// Stage one operand tile (128 x 16) into LDS as [k][r].  `kmajor_global` = the operand's contiguous dimension is
// its row index r (A stored [K][M] / B stored [K][N]); otherwise K is contiguous.
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int rmax, int k0, f32x4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!KMAJOR) {      // P[r][k], 4 consecutive k per thread
            const int r = r0 + (t >> 2) + i * 64, kq = (t & 3) * 4;
            if (r < rmax) v = *reinterpret_cast<const f32x4*>(P + (long)r * ld + k0 + kq);
        } else {                      // P[k][r], 4 consecutive r per thread
            const int k = (t >> 5) + i * 8, rq = r0 + (t & 31) * 4;
            const float* src = P + (long)(k0 + k) * ld + rq;
            if (rq + 3 < rmax) v = *reinterpret_cast<const f32x4*>(src);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (rq + e < rmax) v[e] = src[e];
            }
        }
        reg[i] = v;
    }
}
template <bool KMAJOR>
__device__ __forceinline__ void store_tile(float (*S)[FLD], const f32x4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (!KMAJOR) {
            const int r = (t >> 2) + i * 64, kq = (t & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) S[kq + e][r] = reg[i][e];
        } else {
            const int k = (t >> 5) + i * 8, rq = (t & 31) * 4;
            *reinterpret_cast<f32x4*>(&S[k][rq]) = reg[i];
        }
    }
}

template <bool A_T, bool B_N, int GRP = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Params p) {
    __shared__ __attribute__((aligned(16))) float As[FBK][FLD];
    __shared__ __attribute__((aligned(16))) float Bs[FBK][FLD];
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
    if constexpr (GRP == 1) {          // rows grouped by expert: this 128-row tile uses expert e's B
        const int e = p.grp_tile_expert[blockIdx.y];
        if (e < 0) return;
        p.B += (long)e * p.grp_stride;
    } else if constexpr (GRP == 2) {   // contraction grouped: expert = blockIdx.z, rows [off[e], off[e+1]) of A and B
        const int e = blockIdx.z, r0 = p.grp_off[e];
        p.A += (long)r0 * p.lda; p.B += (long)r0 * p.ldb; p.C += (long)e * p.grp_stride;
        p.K = p.grp_off[e + 1] - r0;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 ra[2], rb[2];
    if (p.K > 0) {                    // (a grouped launch may hand an expert an empty contraction range)
        load_tile<A_T>(p.A, p.lda, m0, p.M, 0, ra);
        load_tile<B_N>(p.B, p.ldb, n0, p.N, 0, rb);
    }
    const int l31 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
        __syncthreads();              // previous tile's fragment reads are done
        store_tile<A_T>(As, ra);
        store_tile<B_N>(Bs, rb);
        __syncthreads();
        if (k0 + FBK < p.K) {         // register prefetch of the next tile overlaps the MFMAs below
            load_tile<A_T>(p.A, p.lda, m0, p.M, k0 + FBK, ra);
            load_tile<B_N>(p.B, p.ldb, n0, p.N, k0 + FBK, rb);
        }
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk + kh][wm + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[kk + kh][wn + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout of the 32x32 MFMAs: lane -> column lane&31, rows 8*(e/4) + 4*(lane>>5) + e%4
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + 8 * (e >> 2) + 4 * kh + (e & 3);
                if (m >= p.M) continue;
                float v = acc[i][j][e] + bv;
                if (p.act != AA_ACT_NONE) v = gemm_act(v, p.act);
                if (p.residual) v += p.residual[(long)m * p.ldr + n];
                float* c = p.C + (long)m * p.ldc + n;
                if (p.flags & AA_GEMM_ACCUM) v += *c;
                *c = v;
            }
        }
}

}  // namespace

extern "C" int aa_gemm_grouped_f32(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                                   int flags, int mode, const int* tile_expert, const int* seg_off, long stride, int E,
                                   void* stream) {
    AA_REQUIRE(mode >= 1 && mode <= 3, "aa_gemm_grouped_f32: mode %d (1 = rows grouped, 2 = contraction grouped, 3 = rows grouped in 256-aligned segments)", mode);
    AA_REQUIRE(M > 0 && N > 0 && E > 0 && (lda & 3) == 0 && (ldb & 3) == 0, "aa_gemm_grouped_f32: bad shape M=%d N=%d E=%d", M, N, E);
    const bool a_t = flags & AA_GEMM_A_T, b_n = flags & AA_GEMM_B_N;
    GemmF32Params p{(const float*)A, (const float*)B, (float*)C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, AA_ACT_NONE, flags,
                    tile_expert, seg_off, stride};
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1 || mode == 3) {
        AA_REQUIRE(tile_expert != nullptr && !a_t && K % FBK == 0 && M % FBM == 0, "aa_gemm_grouped_f32: mode 1 needs tile_expert, A row-major, M %% 128 == 0");
        const dim3 grid(aa_cdiv(N, FBN), M / FBM);
        if (b_n) hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false, 1>), grid, dim3(256), 0, st, p);
    } else {
        AA_REQUIRE(seg_off != nullptr && a_t && b_n, "aa_gemm_grouped_f32: mode 2 needs seg_off and the TN layout");
        hipLaunchKernelGGL((gemm_f32_kernel<true, true, 2>), dim3(aa_cdiv(N, FBN), aa_cdiv(M, FBM), E), dim3(256), 0, st, p);
    }
    AA_CHECK_LAUNCH("aa_gemm_grouped_f32");
    return AA_OK;
}

extern "C" int aa_gemm_f32(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb,
                           long ldc, const void* bias, const void* residual, long ldr, int act, int flags,
                           void* stream) {
    AA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && (K % FBK) == 0, "aa_gemm_f32: K=%d must be a multiple of %d", K, FBK);
    AA_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0, "aa_gemm_f32: lda=%ld / ldb=%ld must be multiples of 4", lda, ldb);
    if (M == 0 || N == 0) return AA_OK;
    GemmF32Params p{(const float*)A, (const float*)B, (float*)C, (const float*)bias, (const float*)residual,
                    M, N, K, lda, ldb, ldc, ldr, act, flags, nullptr, nullptr, 0};
    const dim3 grid(aa_cdiv(N, FBN), aa_cdiv(M, FBM));
    hipStream_t st = (hipStream_t)stream;
    const bool a_t = flags & AA_GEMM_A_T, b_n = flags & AA_GEMM_B_N;
    if (!a_t && !b_n) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, p);
    else if (!a_t && b_n) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (a_t && !b_n) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, p);
    AA_CHECK_LAUNCH("aa_gemm_f32");
    return AA_OK;
}
