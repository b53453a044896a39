// lm_head x log-prob without the [rows, V] logits buffer (gfx950).
//   aa_lmhead_logprob_fwd / _bwd  <- align_anything/utils/tools.py:402-413 (log_softmax + gather) applied to
//                                    `model(**batch).logits` (trainers/text_to_text/dpo.py:128-138), lm_head included
// The vocabulary is walked in chunks of `chunk` columns: each chunk's logits come out of the lm_head GEMM into a
// [rows, chunk] scratch, an online (max, sum-exp) pass folds them into per-lane running state, and the scratch is
// reused for the next chunk.  Backward recomputes each chunk, turns it into dlogits in place and feeds the two
// gradient GEMMs (d_hidden accumulated in fp32 across chunks, dW written chunk by chunk).
//
// Rounding points are those of the unfused pair (aa_gemm_* then aa_logprob_gather_*): the GEMM rounds the logits to the
// compute dtype exactly as the reference's lm_head does, and lane t of the chunk pass owns the same 16-byte vectors of a
// row, in the same order, as lane t of logprob_fwd_kernel (chunk starts are multiples of 256 vectors), so logp / lse are
// bit-identical to the one-pass kernel over a materialised buffer.  HBM-bound glue around MFMA GEMMs: one coalesced
// read of the scratch per chunk in forward, one read + one write in backward.
#include "aa_common.h"
#include "gemm_params.h"

#define LOG2E 1.4426950408889634f

extern "C" int aa_gemm_bf16(const void*, const void*, void*, int, int, int, long, long, long, const void*, const void*,
                            long, int, int, void*);
extern "C" int aa_gemm_f32(const void*, const void*, void*, int, int, int, long, long, long, const void*, const void*,
                           long, int, int, void*);

namespace {

template <typename T> struct ChunkLoad;
template <> struct ChunkLoad<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
        u16x8 r = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(r[i]);
    }
    __device__ static __forceinline__ float one(const bf16_t* p) { return bf2f(*p); }
};
template <> struct ChunkLoad<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void load(const float* p, float* v) {
        f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    __device__ static __forceinline__ float one(const float* p) { return *p; }
};

// One 256-thread workgroup per row.  state[row][lane] = (m, s) of the vectors lane has seen in earlier chunks;
// picked[row] = the label's logit once its chunk has passed.  The last chunk also folds the sub-vector tail of the row
// (V % VEC elements) and reduces across the workgroup, as logprob_fwd_kernel does.
template <typename T>
__global__ __launch_bounds__(256) void lmhead_lse_chunk_kernel(const T* __restrict__ x_all, long ld,
                                                               const int64_t* __restrict__ labels,
                                                               float2* __restrict__ state, float* __restrict__ picked,
                                                               float* __restrict__ logp, float* __restrict__ lse_out,
                                                               int c0, int vc, int V, int first, int last,
                                                               int round_bf16) {
    __shared__ float red[8];
    constexpr int VEC = ChunkLoad<T>::VEC;
    const long row = blockIdx.x;
    const T* x = x_all + row * ld;
    float m = -INFINITY, s = 0.f;
    if (!first) {
        const float2 ms = state[row * 256 + threadIdx.x];
        m = ms.x; s = ms.y;
    }
    const int nvec_total = V / VEC;
    int nvec = nvec_total - c0 / VEC;
    nvec = nvec < 0 ? 0 : (nvec > vc / VEC ? vc / VEC : nvec);
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC];
        ChunkLoad<T>::load(x + (long)i * VEC, v);
        float vm = v[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) vm = fmaxf(vm, v[j]);
        const float mn = fmaxf(m, vm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += exp2f((v[j] - mn) * LOG2E);
        s = s * exp2f((m - mn) * LOG2E) + acc;
        m = mn;
    }
    const int64_t lab = labels[row];
    if (threadIdx.x == 0 && lab >= c0 && lab < c0 + vc) picked[row] = ChunkLoad<T>::one(x + (lab - c0));
    if (!last) {
        state[row * 256 + threadIdx.x] = make_float2(m, s);
        return;
    }
    for (int i = nvec_total * VEC + threadIdx.x; i < V; i += 256) {
        const float v = ChunkLoad<T>::one(x + (i - c0));
        const float mn = fmaxf(m, v);
        s = s * exp2f((m - mn) * LOG2E) + exp2f((v - mn) * LOG2E);
        m = mn;
    }
    const float gm = block_max<256>(m, red);
    const float sc = (m == -INFINITY) ? 0.f : s * exp2f((m - gm) * LOG2E);
    const float gs = block_sum<256>(sc, red);
    if (threadIdx.x == 0) {
        const float lse = gm + logf(gs);
        float out;
        if (lab < 0 || lab >= V) {
            out = __builtin_nanf("");  // torch.gather would raise; surface it as NaN (aa_logprob_gather_fwd does the same)
        } else {
            out = picked[row] - lse;
            if (round_bf16) out = rbf(out);
        }
        logp[row] = out;
        lse_out[row] = lse;
    }
}

// chunk of logits -> chunk of dlogits, in place: dlogp[r] * (1[c0 + v == label_r] - exp(x - lse_r))
template <typename T>
__global__ __launch_bounds__(256) void lmhead_dlogits_chunk_kernel(T* x_all, long ld, const int64_t* __restrict__ labels,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ dlogp, int c0, int vc) {
    constexpr int VEC = ChunkLoad<T>::VEC;
    const long row = blockIdx.x;
    T* x = x_all + row * ld;
    const float g = dlogp[row];
    const float l = lse[row];
    const long labl = labels[row] - c0;
    const int lab = (labl >= 0 && labl < vc) ? (int)labl : -1;
    const int nvec = vc / VEC;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC];
        ChunkLoad<T>::load(x + (long)i * VEC, v);
        const int base = i * VEC;
        if constexpr (sizeof(T) == 2) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = f2bf(g * ((base + j == lab ? 1.f : 0.f) - exp2f((v[j] - l) * LOG2E)));
            *reinterpret_cast<u16x8*>(x + base) = o;
        } else {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = g * ((base + j == lab ? 1.f : 0.f) - exp2f((v[j] - l) * LOG2E));
            *reinterpret_cast<f32x4*>(x + base) = o;
        }
    }
    for (int i = nvec * VEC + threadIdx.x; i < vc; i += 256) {
        const float o = g * ((i == lab ? 1.f : 0.f) - exp2f((ChunkLoad<T>::one(x + i) - l) * LOG2E));
        if constexpr (sizeof(T) == 2) x[i] = f2bf(o); else x[i] = o;
    }
}

__global__ __launch_bounds__(256) void lmhead_cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long ldd,
                                                               int h) {
    const long row = blockIdx.x;
    for (int c = threadIdx.x * 4; c < h; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + row * h + c);
        u16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
        *reinterpret_cast<u16x4*>(dst + row * ldd + c) = o;
    }
}

inline int gemm_any(int dtype, const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                    int flags, void* stream) {
    return dtype == 0 ? aa_gemm_bf16(A, B, C, M, N, K, lda, ldb, ldc, nullptr, nullptr, 0, 0, flags, stream)
                      : aa_gemm_f32(A, B, C, M, N, K, lda, ldb, ldc, nullptr, nullptr, 0, 0, flags, stream);
}

}  // namespace

// bytes of scratch the two entry points need for (rows, chunk, h): [rows, chunk] logits chunk + per-lane LSE state +
// picked logits (+ the fp32 d_hidden accumulator in backward)
static long ws_bytes_for(int rows, int chunk, int h, int dtype, int backward) {
    const long el = dtype == 0 ? 2 : 4;
    long b = (long)rows * chunk * el;
    b = (b + 255) / 256 * 256;
    // backward: fp32 d_hidden accumulator + a zero-padded copy of the last (V mod K-granule) weight rows (ragged vocabularies)
    b += backward ? (long)rows * h * 4 + 64L * h * 4 : (long)rows * 256 * 8 + (long)rows * 4;
    return b;
}
extern "C" int aa_lmhead_logprob_ws_bytes(int rows, int chunk, int h, int dtype, int backward, long* bytes_out) {
    AA_REQUIRE(bytes_out && rows >= 0 && chunk > 0 && h > 0 && (dtype == 0 || dtype == 1), "aa_lmhead_logprob_ws_bytes: bad arguments");
    *bytes_out = ws_bytes_for(rows, chunk, h, dtype, backward);
    return AA_OK;
}

extern "C" int aa_lmhead_logprob_fwd(const void* hidden, long ldh, const void* W, long ldw, const int64_t* labels,
                                     float* logp, float* lse, void* ws, long ws_bytes, int rows, int V, int h,
                                     int chunk, int dtype, int round_bf16, void* stream) {
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_lmhead_logprob_fwd: dtype must be 0 (bf16) or 1 (f32)");
    AA_REQUIRE(rows >= 0 && V > 0 && h > 0 && ldh >= h && ldw >= h, "aa_lmhead_logprob_fwd: bad shape rows=%d V=%d h=%d", rows, V, h);
    AA_REQUIRE(chunk > 0 && chunk % 2048 == 0, "aa_lmhead_logprob_fwd: chunk=%d must be a positive multiple of 2048 (256 lanes x 16-byte vectors)", chunk);
    AA_REQUIRE(V % 4 == 0, "aa_lmhead_logprob_fwd: V=%d must be a multiple of 4", V);
    AA_REQUIRE(ws && ws_bytes >= ws_bytes_for(rows, chunk, h, dtype, 0),
               "aa_lmhead_logprob_fwd: scratch too small (%ld bytes, need %ld)", ws_bytes, ws_bytes_for(rows, chunk, h, dtype, 0));
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
    const long el = dtype == 0 ? 2 : 4;
    char* base = (char*)ws;
    const long chunk_bytes = ((long)rows * chunk * el + 255) / 256 * 256;
    float2* state = (float2*)(base + chunk_bytes);
    float* picked = (float*)(base + chunk_bytes + (long)rows * 256 * 8);
    for (int c0 = 0; c0 < V; c0 += chunk) {
        const int vc = V - c0 < chunk ? V - c0 : chunk;
        const int first = c0 == 0, last = c0 + vc >= V;
        const int rc = gemm_any(dtype, hidden, (const char*)W + (long)c0 * ldw * el, ws, rows, vc, h, ldh, ldw, chunk,
                                dtype == 1 ? AA_GEMM_OUT_F32 : 0, stream);
        if (rc != AA_OK) return rc;
        if (dtype == 0)
            hipLaunchKernelGGL(lmhead_lse_chunk_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, (const bf16_t*)ws, (long)chunk,
                               labels, state, picked, logp, lse, c0, vc, V, first, last, round_bf16);
        else
            hipLaunchKernelGGL(lmhead_lse_chunk_kernel<float>, dim3(rows), dim3(256), 0, st, (const float*)ws, (long)chunk,
                               labels, state, picked, logp, lse, c0, vc, V, first, last, round_bf16);
        AA_CHECK_LAUNCH("aa_lmhead_logprob_fwd");
    }
    return AA_OK;
}

// d_hidden [rows, h] (compute dtype) = sum over chunks of dlogits_chunk @ W_chunk (fp32 across chunks, rounded once);
// dW [V, h] (+)= dlogits^T @ hidden when dW != NULL (dw_f32: fp32 gradient buffer; dw_accumulate: add to it).
extern "C" int aa_lmhead_logprob_bwd(const void* hidden, long ldh, const void* W, long ldw, const int64_t* labels,
                                     const float* lse, const float* dlogp, void* d_hidden, long lddh, void* dW,
                                     long lddw, int dw_f32, int dw_accumulate, void* ws, long ws_bytes, int rows,
                                     int V, int h, int chunk, int dtype, void* stream) {
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_lmhead_logprob_bwd: dtype must be 0 (bf16) or 1 (f32)");
    AA_REQUIRE(rows >= 0 && V > 0 && h > 0 && ldh >= h && ldw >= h && lddh >= h, "aa_lmhead_logprob_bwd: bad shape rows=%d V=%d h=%d", rows, V, h);
    AA_REQUIRE(chunk > 0 && chunk % 2048 == 0, "aa_lmhead_logprob_bwd: chunk=%d must be a positive multiple of 2048", chunk);
    AA_REQUIRE(V % 4 == 0 && h % 4 == 0, "aa_lmhead_logprob_bwd: V=%d and h=%d must be multiples of 4", V, h);
    AA_REQUIRE(dtype == 0 || !dW || dw_f32, "aa_lmhead_logprob_bwd: fp32 operands need an fp32 dW");
    AA_REQUIRE(ws && ws_bytes >= ws_bytes_for(rows, chunk, h, dtype, 1),
               "aa_lmhead_logprob_bwd: scratch too small (%ld bytes, need %ld)", ws_bytes, ws_bytes_for(rows, chunk, h, dtype, 1));
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
    const long el = dtype == 0 ? 2 : 4;
    const long chunk_bytes = ((long)rows * chunk * el + 255) / 256 * 256;
    float* dh32 = (float*)((char*)ws + chunk_bytes);
    char* wtail = (char*)ws + chunk_bytes + (long)rows * h * 4;      // [KG, h] zero-padded tail rows of W
    const int KG = dtype == 0 ? 64 : 16;                              // contraction granule of aa_gemm_bf16 / aa_gemm_f32
    if (hipMemsetAsync(dh32, 0, (size_t)rows * h * 4, st) != hipSuccess) {
        aa_set_error("aa_lmhead_logprob_bwd: hipMemsetAsync failed");
        return AA_ERR_LAUNCH;
    }
    for (int c0 = 0; c0 < V; c0 += chunk) {
        const int vc = V - c0 < chunk ? V - c0 : chunk;
        const char* Wc = (const char*)W + (long)c0 * ldw * el;
        int rc = gemm_any(dtype, hidden, Wc, ws, rows, vc, h, ldh, ldw, chunk, dtype == 1 ? AA_GEMM_OUT_F32 : 0, stream);
        if (rc != AA_OK) return rc;
        if (dtype == 0)
            hipLaunchKernelGGL(lmhead_dlogits_chunk_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, (bf16_t*)ws, (long)chunk, labels,
                               lse, dlogp, c0, vc);
        else
            hipLaunchKernelGGL(lmhead_dlogits_chunk_kernel<float>, dim3(rows), dim3(256), 0, st, (float*)ws, (long)chunk, labels,
                               lse, dlogp, c0, vc);
        AA_CHECK_LAUNCH("aa_lmhead_logprob_bwd");
        // d_hidden += dlogits_chunk @ W_chunk: contraction over the chunk's vocabulary columns.  A ragged vocabulary (OPT: 50272 =
        // 64 x 785 + 32) leaves a last piece shorter than the GEMM's K granule: it runs as one granule against a zero-padded copy
        // of those weight rows, with the dlogits columns past the vocabulary zeroed in the scratch chunk.
        const int tail = vc % KG, vc_main = vc - tail;
        if (vc_main) {
            rc = gemm_any(dtype, ws, Wc, dh32, rows, h, vc_main, chunk, ldw, h, AA_GEMM_B_N | AA_GEMM_OUT_F32 | AA_GEMM_ACCUM, stream);
            if (rc != AA_OK) return rc;
        }
        if (tail) {
            hipError_t e = hipMemset2DAsync((char*)ws + (long)vc * el, (size_t)chunk * el, 0, (size_t)(KG - tail) * el, rows, st);
            if (e == hipSuccess) e = hipMemsetAsync(wtail, 0, (size_t)KG * h * el, st);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(wtail, (size_t)h * el, Wc + (long)vc_main * ldw * el, (size_t)ldw * el, (size_t)h * el, tail,
                                     hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) {
                aa_set_error("aa_lmhead_logprob_bwd: staging the ragged vocabulary tail failed: %s", hipGetErrorString(e));
                return AA_ERR_LAUNCH;
            }
            rc = gemm_any(dtype, (char*)ws + (long)vc_main * el, wtail, dh32, rows, h, KG, chunk, h, h,
                          AA_GEMM_B_N | AA_GEMM_OUT_F32 | AA_GEMM_ACCUM, stream);
            if (rc != AA_OK) return rc;
        }
        if (dW) {
            const long del = dw_f32 ? 4 : 2;
            rc = gemm_any(dtype, ws, hidden, (char*)dW + (long)c0 * lddw * del, vc, h, rows, chunk, ldh, lddw,
                          AA_GEMM_A_T | AA_GEMM_B_N | (dw_f32 ? AA_GEMM_OUT_F32 : 0) | (dw_accumulate ? AA_GEMM_ACCUM : 0), stream);
            if (rc != AA_OK) return rc;
        }
    }
    if (dtype == 0) {
        hipLaunchKernelGGL(lmhead_cast_bf16_kernel, dim3(rows), dim3(256), 0, st, dh32, (bf16_t*)d_hidden, lddh, h);
        AA_CHECK_LAUNCH("aa_lmhead_logprob_bwd");
    } else if (hipMemcpy2DAsync(d_hidden, (size_t)lddh * 4, dh32, (size_t)h * 4, (size_t)h * 4, rows, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        aa_set_error("aa_lmhead_logprob_bwd: copy of d_hidden failed");
        return AA_ERR_LAUNCH;
    }
    return AA_OK;
}
