// fp32 attention (forward + backward) for the parity mode: same contract as aa_attn_fwd / aa_attn_bwd (token-major
// [N*T, ld] activations, head h at columns [h*HD, (h+1)*HD), key j valid iff start[n] <= j < T and, when causal,
// j <= query index; fully masked query rows give O = 0 and lse = -inf; GQA via Hkv), but every tensor is fp32 and
// all arithmetic is fp32 FMA with expf -- no bf16 rounding anywhere.  Flash-style (online softmax, nothing of size
// T x T is materialised): one query (or key) row is owned by 4 adjacent lanes, each holding HD/4 of the head
// dimension in registers; the other operand streams through LDS in 32-row tiles and is read as wave-broadcast
// float4s.  Exists to track the reference's fp32 CPU trainer to 1e-4 on the loss; throughput is secondary.
#include "aa_common.h"

namespace {

struct AttnF32Params {
    const float* Q; const float* K; const float* V; float* O;
    const float* dO; float* dQ; float* dK; float* dV;
    float* lse; float* delta; const int* start; const int* kvlen;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int N, T, H, Hkv, causal;
    float scale;
};

constexpr int TILE = 32;   // rows of the streamed operand per LDS tile
constexpr int ROWS = 64;   // rows owned by one 256-thread block (4 lanes per row)

__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

// cooperative load of `TILE` rows x HD floats (rows r0.. of sequence n, head column col) into LDS, zero beyond T
template <int HD>
__device__ __forceinline__ void stage_tile(float (*S)[HD], const float* __restrict__ P, long ld, long seq_row0,
                                           int r0, int T, int col) {
    constexpr int V4 = HD / 4;
    for (int i = threadIdx.x; i < TILE * V4; i += 256) {
        const int r = i / V4, c = (i % V4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < T) v = *reinterpret_cast<const f32x4*>(P + (seq_row0 + r0 + r) * ld + col + c);
        *reinterpret_cast<f32x4*>(&S[r][c]) = v;
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const AttnF32Params p) {
    constexpr int DPL = HD / 4;
    __shared__ __attribute__((aligned(16))) float Ks[TILE][HD];
    __shared__ __attribute__((aligned(16))) float Vs[TILE][HD];
    const int qblocks = (p.T + ROWS - 1) / ROWS;
    const int qb = blockIdx.x % qblocks;
    const int h = (blockIdx.x / qblocks) % p.H;
    const int n = blockIdx.x / (qblocks * p.H);
    const int hk = h / (p.H / p.Hkv);
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const int qi = qb * ROWS + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    const long seq0 = (long)n * T;
    float q[DPL], o[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (qi < T) {
        const float* qr = p.Q + (seq0 + qi) * p.ldq + h * HD + part * DPL;
#pragma unroll
        for (int d = 0; d < DPL; d += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qr + d);
            q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        }
    }
    float m = -INFINITY, l = 0.f;
    const int kv_end = p.causal ? min(T, qb * ROWS + ROWS) : T;
    for (int k0 = (start / TILE) * TILE; k0 < kv_end; k0 += TILE) {
        __syncthreads();
        stage_tile<HD>(Ks, p.K, p.ldk, seq0, k0, T, hk * HD);
        stage_tile<HD>(Vs, p.V, p.ldv, seq0, k0, T, hk * HD);
        __syncthreads();
        float s[TILE];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TILE; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < DPL; d += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(&Ks[j][part * DPL + d]);
                acc += q[d] * kv[0] + q[d + 1] * kv[1] + q[d + 2] * kv[2] + q[d + 3] * kv[3];
            }
            acc = quad_sum(acc) * p.scale;
            const int kj = k0 + j;
            const bool ok = kj >= start && kj < KT && (!p.causal || kj <= qi) && qi < T;
            s[j] = ok ? acc : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        if (mx == -INFINITY) continue;          // nothing attendable in this tile for this row (uniform per quad)
        const float mn = fmaxf(m, mx);
        const float corr = expf(m - mn);        // m = -inf -> 0
        l *= corr;
#pragma unroll
        for (int d = 0; d < DPL; ++d) o[d] *= corr;
#pragma unroll
        for (int j = 0; j < TILE; ++j) {
            const float pj = expf(s[j] - mn);   // masked -> exp(-inf) = 0
            l += pj;
#pragma unroll
            for (int d = 0; d < DPL; d += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[j][part * DPL + d]);
                o[d] += pj * vv[0]; o[d + 1] += pj * vv[1]; o[d + 2] += pj * vv[2]; o[d + 3] += pj * vv[3];
            }
        }
        m = mn;
    }
    if (qi < T) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        float* orow = p.O + (seq0 + qi) * p.ldo + h * HD + part * DPL;
#pragma unroll
        for (int d = 0; d < DPL; d += 4)
            *reinterpret_cast<f32x4*>(orow + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
        if (part == 0 && p.lse) p.lse[((long)n * p.H + h) * T + qi] = l > 0.f ? m + logf(l) : -INFINITY;
    }
}

// dQ (and delta = rowsum(dO * O), stored for the dK/dV kernel): one query row per lane quad, keys stream through LDS
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_bwd_dq_kernel(const AttnF32Params p) {
    constexpr int DPL = HD / 4;
    __shared__ __attribute__((aligned(16))) float Ks[TILE][HD];
    __shared__ __attribute__((aligned(16))) float Vs[TILE][HD];
    const int qblocks = (p.T + ROWS - 1) / ROWS;
    const int qb = blockIdx.x % qblocks;
    const int h = (blockIdx.x / qblocks) % p.H;
    const int n = blockIdx.x / (qblocks * p.H);
    const int hk = h / (p.H / p.Hkv);
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const int qi = qb * ROWS + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    const long seq0 = (long)n * T;
    float q[DPL], go[DPL], dq[DPL];
    float lse = -INFINITY, delta = 0.f;
#pragma unroll
    for (int d = 0; d < DPL; ++d) { q[d] = 0.f; go[d] = 0.f; dq[d] = 0.f; }
    if (qi < T) {
        const float* qr = p.Q + (seq0 + qi) * p.ldq + h * HD + part * DPL;
        const float* gr = p.dO + (seq0 + qi) * p.lddo + h * HD + part * DPL;
        const float* orow = p.O + (seq0 + qi) * p.ldo + h * HD + part * DPL;
        float dl = 0.f;
#pragma unroll
        for (int d = 0; d < DPL; ++d) { q[d] = qr[d]; go[d] = gr[d]; dl += gr[d] * orow[d]; }
        delta = quad_sum(dl);
        lse = p.lse[((long)n * p.H + h) * T + qi];
        if (part == 0) p.delta[((long)n * p.H + h) * T + qi] = delta;
    } else {
        quad_sum(0.f);
    }
    const int kv_end = p.causal ? min(T, qb * ROWS + ROWS) : T;
    for (int k0 = (start / TILE) * TILE; k0 < kv_end; k0 += TILE) {
        __syncthreads();
        stage_tile<HD>(Ks, p.K, p.ldk, seq0, k0, T, hk * HD);
        stage_tile<HD>(Vs, p.V, p.ldv, seq0, k0, T, hk * HD);
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < TILE; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int d = 0; d < DPL; ++d) {
                a += q[d] * Ks[j][part * DPL + d];
                b += go[d] * Vs[j][part * DPL + d];
            }
            a = quad_sum(a) * p.scale;
            b = quad_sum(b);
            const int kj = k0 + j;
            const bool ok = kj >= start && kj < KT && (!p.causal || kj <= qi) && qi < T && lse != -INFINITY;
            const float pj = ok ? expf(a - lse) : 0.f;
            const float ds = pj * (b - delta) * p.scale;
#pragma unroll
            for (int d = 0; d < DPL; ++d) dq[d] += ds * Ks[j][part * DPL + d];
        }
    }
    if (qi < T) {
        float* dr = p.dQ + (seq0 + qi) * p.lddq + h * HD + part * DPL;
#pragma unroll
        for (int d = 0; d < DPL; ++d) dr[d] = dq[d];
    }
}

// dK / dV: one key row per lane quad; the queries (all heads of the KV group) stream through LDS with their dO,
// lse and delta.  Deterministic: every dK/dV element is produced by exactly one lane, no atomics.
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_bwd_dkv_kernel(const AttnF32Params p) {
    constexpr int DPL = HD / 4;
    __shared__ __attribute__((aligned(16))) float Qs[TILE][HD];
    __shared__ __attribute__((aligned(16))) float Gs[TILE][HD];
    __shared__ float Ls[TILE], Ds[TILE];
    const int kblocks = (p.T + ROWS - 1) / ROWS;
    const int kb = blockIdx.x % kblocks;
    const int hk = (blockIdx.x / kblocks) % p.Hkv;
    const int n = blockIdx.x / (kblocks * p.Hkv);
    const int group = p.H / p.Hkv;
    const int T = p.T;
    const int start = p.start ? p.start[n] : 0;
    const int KT = p.kvlen ? min(p.kvlen[n], p.T) : p.T;   // keys [start, KT) are attendable
    const int kj = kb * ROWS + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    const long seq0 = (long)n * T;
    float k[DPL], v[DPL], dk[DPL], dv[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    const bool key_ok = kj < KT && kj >= start;
    if (kj < T) {
        const float* kr = p.K + (seq0 + kj) * p.ldk + hk * HD + part * DPL;
        const float* vr = p.V + (seq0 + kj) * p.ldv + hk * HD + part * DPL;
#pragma unroll
        for (int d = 0; d < DPL; ++d) { k[d] = kr[d]; v[d] = vr[d]; }
    }
    const int q_begin = p.causal ? (kb * ROWS / TILE) * TILE : 0;
    for (int g = 0; g < group; ++g) {
        const int h = hk * group + g;
        for (int q0 = q_begin; q0 < T; q0 += TILE) {
            __syncthreads();
            stage_tile<HD>(Qs, p.Q, p.ldq, seq0, q0, T, h * HD);
            stage_tile<HD>(Gs, p.dO, p.lddo, seq0, q0, T, h * HD);
            if (threadIdx.x < TILE) {
                const int qi = q0 + threadIdx.x;
                Ls[threadIdx.x] = qi < T ? p.lse[((long)n * p.H + h) * T + qi] : -INFINITY;
                Ds[threadIdx.x] = qi < T ? p.delta[((long)n * p.H + h) * T + qi] : 0.f;
            }
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < TILE; ++i) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int d = 0; d < DPL; ++d) {
                    a += Qs[i][part * DPL + d] * k[d];
                    b += Gs[i][part * DPL + d] * v[d];
                }
                a = quad_sum(a) * p.scale;
                b = quad_sum(b);
                const int qi = q0 + i;
                const float lse = Ls[i];
                const bool ok = key_ok && qi < T && (!p.causal || kj <= qi) && lse != -INFINITY;
                const float pj = ok ? expf(a - lse) : 0.f;
                const float ds = pj * (b - Ds[i]) * p.scale;
#pragma unroll
                for (int d = 0; d < DPL; ++d) {
                    dv[d] += pj * Gs[i][part * DPL + d];
                    dk[d] += ds * Qs[i][part * DPL + d];
                }
            }
        }
    }
    if (kj < T) {
        float* dkr = p.dK + (seq0 + kj) * p.lddk + hk * HD + part * DPL;
        float* dvr = p.dV + (seq0 + kj) * p.lddv + hk * HD + part * DPL;
#pragma unroll
        for (int d = 0; d < DPL; ++d) { dkr[d] = dk[d]; dvr[d] = dv[d]; }
    }
}

int check_f32(const char* who, int N, int T, int H, int Hkv, int hd) {
    AA_REQUIRE(N >= 0 && T > 0 && H > 0 && Hkv > 0 && H % Hkv == 0, "%s: bad geometry N=%d T=%d H=%d Hkv=%d", who, N, T, H, Hkv);
    AA_REQUIRE(hd == 64 || hd == 128, "%s: head_dim %d not supported (64, 128)", who, hd);
    return AA_OK;
}

}  // namespace

extern "C" int aa_attn_fwd_f32(const void* Q, const void* K, const void* V, void* O, float* lse, const int* start,
                               const int* kv_len, long ldq, long ldk, long ldv, long ldo, int N, int T, int H, int Hkv, int hd,
                               int causal, float scale, void* stream) {
    int rc = check_f32("aa_attn_fwd_f32", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE((ldq | ldk | ldv | ldo) % 4 == 0, "aa_attn_fwd_f32: leading dims must be multiples of 4");
    if (N == 0) return AA_OK;
    AttnF32Params p{};
    p.Q = (const float*)Q; p.K = (const float*)K; p.V = (const float*)V; p.O = (float*)O; p.lse = lse; p.start = start; p.kvlen = kv_len;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal;
    p.scale = scale;
    const dim3 grid(aa_cdiv(T, ROWS) * H * N);
    if (hd == 128) hipLaunchKernelGGL(attn_f32_fwd_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_f32_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AA_CHECK_LAUNCH("aa_attn_fwd_f32");
    return AA_OK;
}

extern "C" int aa_attn_bwd_f32(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                               const float* lse, float* delta, void* dQ, void* dK, void* dV, const int* start,
                               const int* kv_len, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv,
                               int N, int T, int H, int Hkv, int hd, int causal, float scale, void* stream) {
    int rc = check_f32("aa_attn_bwd_f32", N, T, H, Hkv, hd);
    if (rc) return rc;
    AA_REQUIRE((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 4 == 0, "aa_attn_bwd_f32: leading dims must be multiples of 4");
    if (N == 0) return AA_OK;
    AttnF32Params p{};
    p.Q = (const float*)Q; p.K = (const float*)K; p.V = (const float*)V; p.O = (float*)O; p.dO = (const float*)dO;
    p.dQ = (float*)dQ; p.dK = (float*)dK; p.dV = (float*)dV; p.lse = const_cast<float*>(lse); p.delta = delta; p.start = start; p.kvlen = kv_len;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.N = N; p.T = T; p.H = H; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gq(aa_cdiv(T, ROWS) * H * N), gkv(aa_cdiv(T, ROWS) * Hkv * N);
    if (hd == 128) {
        hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<128>, gq, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<128>, gkv, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<64>, gq, dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<64>, gkv, dim3(256), 0, st, p);
    }
    AA_CHECK_LAUNCH("aa_attn_bwd_f32");
    return AA_OK;
}
