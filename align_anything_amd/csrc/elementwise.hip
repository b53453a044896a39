// HBM-bound block kernels of the transformer forward/backward for gfx950: RMSNorm, LayerNorm, RoPE,
// SwiGLU, GELU/quick-GELU/ReLU (+backward), embedding gather / image-feature scatter, transposes,
// column reductions (bias gradients), CLIP patch im2col.  All loads/stores are 16 B per lane;
// reductions use wave shuffles + LDS.  Numerics follow the installed HF transformers graph that the
// reference executes (SURVEY.md §8 a'): fp32 internal math, bf16 rounding at the same points.
#include "aa_common.h"

namespace AA_ELEM_NS {

// ================================================================== RMSNorm
// hf:models/llama/modeling_llama.py:62-67 : y = w * bf16(x_f32 * rsqrt(mean(x^2)+eps))
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const elem_t* __restrict__ x,
                                                          const elem_t* __restrict__ w,
                                                          elem_t* __restrict__ y,
                                                          float* __restrict__ rstd_out, int rows,
                                                          int h, float eps) {
    __shared__ float red[8];
    const int nv = h >> 3;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const elem_t* xr = x + row * h;
        float ss = 0.f;
        for (int i = threadIdx.x; i < nv; i += 256) {
            ev8 v = *reinterpret_cast<const ev8*>(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = e2f(v[j]); ss += f * f; }
        }
        ss = block_sum<256>(ss, red);
        const float rstd = rsqrtf(ss / (float)h + eps);
        if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
        elem_t* yr = y + row * h;
        for (int i = threadIdx.x; i < nv; i += 256) {
            ev8 v = *reinterpret_cast<const ev8*>(xr + i * 8);
            ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
            ev8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(wv[j]) * ernd(e2f(v[j]) * rstd));
            *reinterpret_cast<ev8*>(yr + i * 8) = o;
        }
    }
}

// Two block reductions behind ONE barrier pair, each in block_sum's order (wave butterfly, then the waves' sums in wave order): bit-identical to two block_sum calls.
template <int NT>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) { red[w] = a; red[NT / 64 + w] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { ta += red[i]; tb += red[NT / 64 + i]; }
    a = ta; b = tb;
}

// h <= 2048 MAXV: a block walks TWO rows per iteration and keeps them in registers between the sum and the output pass -- twice the bytes in flight per
// barrier pair and x read once (round 4's form read every row twice, the second time through L2: 5.2 TB/s in the step).  Per row the
// arithmetic and its order are those of rmsnorm_fwd_kernel: results are bit-identical.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd2_kernel(const elem_t* __restrict__ x, const elem_t* __restrict__ w, elem_t* __restrict__ y,
                                                           float* __restrict__ rstd_out, int rows, int h, float eps) {
    __shared__ float red[8];
    const int nv = h >> 3;
    for (long row0 = 2L * blockIdx.x; row0 < rows; row0 += 2L * gridDim.x) {
        const bool two = row0 + 1 < rows;
        ev8 v[2][MAXV];
        float ss[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv && (u == 0 || two)) v[u][a] = *reinterpret_cast<const ev8*>(x + (row0 + u) * h + i * 8);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv && (u == 0 || two)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = e2f(v[u][a][j]); ss[u] += f * f; }
                }
            }
        block_sum2<256>(ss[0], ss[1], red);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const float rstd = rsqrtf(ss[u] / (float)h + eps);
            if (threadIdx.x == 0 && rstd_out) rstd_out[row0 + u] = rstd;
            elem_t* yr = y + (row0 + u) * h;
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv) {
                    const ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
                    ev8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(wv[j]) * ernd(e2f(v[u][a][j]) * rstd));
                    *reinterpret_cast<ev8*>(yr + i * 8) = o;
                }
            }
        }
    }
}

// h <= 512 (per-head q / k norms of Qwen3: head_dim 128 over tokens x heads "rows").  LPR = lanes per row (the power of two >= h / 8): a wave holds 64 / LPR rows
// at once and every group walks TWO rows per iteration, so all 64 lanes load 16 bytes and a wave keeps 2 KB in flight (round 3's one-wave-per-row form had 16
// of 64 lanes active and one 256-byte load in flight per wave at head_dim 128: 391 us for 270 MB = latency-bound at 0.7 TB/s).  The row sum
// is the same butterfly over the row's lanes as before (the idle lanes contributed zeros), so results are bit-identical.
// ROPE (h == 8 LPR, Qwen3's per-head q / k norm followed by the rotary embedding, hf:models/qwen3_moe/modeling_qwen3_moe.py q_norm -> apply_rotary_pos_emb):
// the normalised row is rotated before it is stored -- element d pairs with d + h / 2, which lives LPR / 2 lanes away in the same lane group, so the partner
// arrives by one cross-lane read per element; rounding points exactly those of rmsnorm followed by aa_rope_inplace (bit-identical to the pair), one launch
// and one pass over q / k instead of two.  Row r belongs to token r / heads; pos[token] indexes the [., h / 2] tables.
template <int LPR, bool ROPE = false>
__global__ __launch_bounds__(256) void rmsnorm_fwd_small_kernel(const elem_t* __restrict__ x, const elem_t* __restrict__ w,
                                                                elem_t* __restrict__ y, float* __restrict__ rstd_out, long rows,
                                                                int h, float eps, const int* __restrict__ pos = nullptr,
                                                                const elem_t* __restrict__ cos_t = nullptr, const elem_t* __restrict__ sin_t = nullptr,
                                                                int heads = 1, long ldx = 0) {
    constexpr int GPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool act = sub < (h >> 3);
    ev8 wv;
    if (act) wv = *reinterpret_cast<const ev8*>(w + sub * 8);
    for (long row0 = (long)blockIdx.x * (2 * GPB) + grp; row0 < rows; row0 += (long)gridDim.x * (2 * GPB)) {
        ev8 v[2];
        float ss[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = row0 + u * GPB;
            if (act && row < rows) {
                // ROPE form: x may be a column slice of a wider buffer (the fused q | k | v projection): token row / heads starts at x + token * ldx
                const elem_t* xr = ROPE ? x + (row / heads) * ldx + (row % heads) * h : x + row * h;
                v[u] = *reinterpret_cast<const ev8*>(xr + sub * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = e2f(v[u][j]); ss[u] += f * f; }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { ss[0] += __shfl_xor(ss[0], o, 64); ss[1] += __shfl_xor(ss[1], o, 64); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = row0 + u * GPB;
            if (row >= rows) continue;
            const float rstd = rsqrtf(ss[u] / (float)h + eps);
            if (sub == 0 && rstd_out) rstd_out[row] = rstd;
            if (act) {
                ev8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(wv[j]) * ernd(e2f(v[u][j]) * rstd));
                if constexpr (ROPE) {       // every lane of the group is active here (h == 8 LPR) and both rows of a pair of groups exist or not together
                    const int half_l = LPR / 2, cs = (sub % half_l) * 8;
                    const long tab = (long)pos[row / heads] * (h >> 1) + cs;
                    const ev8 c = *reinterpret_cast<const ev8*>(cos_t + tab), sn = *reinterpret_cast<const ev8*>(sin_t + tab);
                    const bool lower = sub < half_l;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float mine = e2f(o[j]), other = __shfl_xor(mine, half_l, 64);
                        const float cc = e2f(c[j]), ss = e2f(sn[j]);
                        // lower half: o1 = rnd(a c) + rnd(-b s) with a = mine, b = other; upper half: o2 = rnd(b c) + rnd(a s) with b = mine, a = other
                        o[j] = lower ? f2e(ernd(mine * cc) + ernd(-other * ss)) : f2e(ernd(mine * cc) + ernd(other * ss));
                    }
                }
                *reinterpret_cast<ev8*>(y + row * h + sub * 8) = o;
            }
        }
    }
}
// backward for h <= 512, same row -> lane mapping; every lane keeps the dw partial of its 8 columns over the rows of its group, the groups of a block meet in LDS
template <int LPR>
__global__ __launch_bounds__(256) void rmsnorm_bwd_small_kernel(const elem_t* __restrict__ dy, const elem_t* __restrict__ x,
                                                                const elem_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                                elem_t* __restrict__ dx, float* __restrict__ dw_part, long rows, int h,
                                                                int add_to_dx, int heads = 0, long ldx = 0, long lddx = 0) {
    // heads > 0: x and dx are column slices of wider buffers (the fused q | k | v projection output and its gradient): row r = (token r / heads, head
    // r % heads) lives at token * ld + head * h; dy stays dense
    constexpr int GPB = 256 / LPR;
    __shared__ float part[GPB][LPR * 8];
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool act = sub < (h >> 3);
    float dwacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ev8 wv;
    if (act) wv = *reinterpret_cast<const ev8*>(w + sub * 8);
    for (long row0 = (long)blockIdx.x * (2 * GPB) + grp; row0 < rows; row0 += (long)gridDim.x * (2 * GPB)) {
        ev8 xv[2], gv[2], ov[2];
        float dot[2] = {0.f, 0.f}, rs[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = row0 + u * GPB;
            if (act && row < rows) {
                rs[u] = rstd_in[row];
                const long xo = heads > 0 ? (row / heads) * ldx + (row % heads) * h : row * h;
                const long dxo = heads > 0 ? (row / heads) * lddx + (row % heads) * h : row * h;
                xv[u] = *reinterpret_cast<const ev8*>(x + xo + sub * 8);
                gv[u] = *reinterpret_cast<const ev8*>(dy + row * h + sub * 8);
                if (add_to_dx) ov[u] = *reinterpret_cast<const ev8*>(dx + dxo + sub * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = row0 + u * GPB;
            if (act && row < rows) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = e2f(xv[u][j]) * rs[u], g = e2f(gv[u][j]);
                    dot[u] += g * e2f(wv[j]) * xh;
                    dwacc[j] += g * ernd(xh);
                }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { dot[0] += __shfl_xor(dot[0], o, 64); dot[1] += __shfl_xor(dot[1], o, 64); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = row0 + u * GPB;
            if (act && row < rows) {
                const float dm = dot[u] / (float)h;
                ev8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = e2f(xv[u][j]) * rs[u];
                    float d = rs[u] * (e2f(gv[u][j]) * e2f(wv[j]) - xh * dm);
                    if (add_to_dx) d += e2f(ov[u][j]);
                    o[j] = f2e(d);
                }
                const long dxo = heads > 0 ? (row / heads) * lddx + (row % heads) * h : row * h;
                *reinterpret_cast<ev8*>(dx + dxo + sub * 8) = o;
            }
        }
    }
    if (dw_part) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[grp][sub * 8 + j] = act ? dwacc[j] : 0.f;
        __syncthreads();
        for (int c = threadIdx.x; c < h; c += 256) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < GPB; ++g) t += part[g][c];
            dw_part[(long)blockIdx.x * h + c] = t;
        }
    }
}

// dx = rstd * (dy*w - xhat * mean(dy*w*xhat)),  dw[c] += sum_rows dy*bf16(xhat)
// dw partials are kept per thread in registers across the block's rows and written as one partial row per block (reduce_rows_kernel sums them).
// A block walks TWO rows per iteration (round 5): their x / dy / residual vectors are all requested before the first is used and the two row
// reductions share one barrier pair -- twice the bytes in flight per block and half the barriers.  Per row, and per column of dw (row order kept), the
// arithmetic is that of the one-row form: dx is bit-identical to round 4's kernel (dw: equal up to the fp32 summation order of the per-block partials, which the grid fixes).  Bytes per launch with the residual gradient added
// (add_to_dx, every call of the decoder stack): read x + dy + dx, write dx = 8 h bytes per row.
template <int MAXV>  // max 16-byte vectors per thread (h <= 256*8*MAXV)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const elem_t* __restrict__ dy,
                                                          const elem_t* __restrict__ x,
                                                          const elem_t* __restrict__ w,
                                                          const float* __restrict__ rstd_in,
                                                          elem_t* __restrict__ dx,
                                                          float* __restrict__ dw_part, int rows, int h,
                                                          int add_to_dx) {
    __shared__ float red[8];
    const int nv = h >> 3;
    float dwacc[MAXV][8];
#pragma unroll
    for (int a = 0; a < MAXV; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[a][j] = 0.f;
    // rows of a block: blockIdx.x, + gridDim.x, + 2 gridDim.x ... (as before), taken two at a time
    for (long row0 = blockIdx.x; row0 < rows; row0 += 2L * gridDim.x) {
        const long row1 = row0 + gridDim.x;
        const bool two = row1 < rows;
        const long rws[2] = {row0, row1};
        float rstd[2], dot[2] = {0.f, 0.f};
        rstd[0] = rstd_in[row0];
        rstd[1] = two ? rstd_in[row1] : 0.f;
        ev8 xk[2][MAXV], gk[2][MAXV], ok[2][MAXV];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv && (u == 0 || two)) {
                    xk[u][a] = *reinterpret_cast<const ev8*>(x + rws[u] * h + i * 8);
                    gk[u][a] = *reinterpret_cast<const ev8*>(dy + rws[u] * h + i * 8);
                    if (add_to_dx) ok[u][a] = *reinterpret_cast<const ev8*>(dx + rws[u] * h + i * 8);
                }
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv && (u == 0 || two)) {
                    const ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = e2f(xk[u][a][j]) * rstd[u];
                        const float g = e2f(gk[u][a][j]);
                        dot[u] += g * e2f(wv[j]) * xh;
                        dwacc[a][j] += g * ernd(xh);
                    }
                }
            }
        block_sum2<256>(dot[0], dot[1], red);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const float dm = dot[u] / (float)h;
            elem_t* dxr = dx + rws[u] * h;
#pragma unroll
            for (int a = 0; a < MAXV; ++a) {
                const int i = threadIdx.x + a * 256;
                if (i < nv) {
                    const ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
                    ev8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = e2f(xk[u][a][j]) * rstd[u];
                        float d = rstd[u] * (e2f(gk[u][a][j]) * e2f(wv[j]) - xh * dm);
                        if (add_to_dx) d += e2f(ok[u][a][j]);
                        o[j] = f2e(d);
                    }
                    *reinterpret_cast<ev8*>(dxr + i * 8) = o;
                }
            }
        }
    }
    if (dw_part) {  // per-block partial row, reduced by reduce_rows_kernel (no atomics on the hot path)
        float* pr = dw_part + (long)blockIdx.x * h;
#pragma unroll
        for (int a = 0; a < MAXV; ++a) {
            const int i = threadIdx.x + a * 256;
            if (i < nv) {
                *reinterpret_cast<f32x4*>(pr + i * 8) = f32x4{dwacc[a][0], dwacc[a][1], dwacc[a][2], dwacc[a][3]};
                *reinterpret_cast<f32x4*>(pr + i * 8 + 4) = f32x4{dwacc[a][4], dwacc[a][5], dwacc[a][6], dwacc[a][7]};
            }
        }
    }
}

// out[c] += sum_r part[r, c]; grid = (ceil(h/64), RS); block = 256 (64 columns x 4 row lanes)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int nrows,
                                                          int h, float* __restrict__ out) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < h)
        for (int r = blockIdx.y * 4 + rl; r < nrows; r += gridDim.y * 4) acc += part[(long)r * h + c];
    sm[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < h) atomicAdd(out + c, sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
static void launch_reduce_rows(const float* part, int nrows, int h, float* out, hipStream_t st) {
    int rs = nrows / 16; if (rs < 1) rs = 1; if (rs > 32) rs = 32;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(aa_cdiv(h, 64), rs), dim3(256), 0, st, part, nrows, h, out);
}

extern "C" int AA_FN(aa_rmsnorm_fwd)(const void* x, const void* w, void* y, float* rstd, int rows, int h,
                              float eps, void* stream) {
    AA_REQUIRE(rows >= 0 && h > 0 && (h & 7) == 0, "aa_rmsnorm_fwd: hidden %d must be a multiple of 8", h);
    if (rows == 0) return AA_OK;
    if (h <= 512) {
        hipStream_t st = (hipStream_t)stream;
#define LAUNCH_RMSF_SMALL(LPR)                                                                                                      \
    do {                                                                                                                            \
        const long nb = ((long)rows + 2 * (256 / LPR) - 1) / (2 * (256 / LPR));                                                     \
        hipLaunchKernelGGL(rmsnorm_fwd_small_kernel<LPR>, dim3((int)(nb < 16384 ? nb : 16384)), dim3(256), 0, st, (const elem_t*)x, \
                           (const elem_t*)w, (elem_t*)y, rstd, (long)rows, h, eps);                                                 \
    } while (0)
        if (h <= 64) LAUNCH_RMSF_SMALL(8); else if (h <= 128) LAUNCH_RMSF_SMALL(16); else if (h <= 256) LAUNCH_RMSF_SMALL(32); else LAUNCH_RMSF_SMALL(64);
#undef LAUNCH_RMSF_SMALL
        AA_CHECK_LAUNCH("aa_rmsnorm_fwd");
        return AA_OK;
    }
    if (h <= 8192) {
        const int pairs = (rows + 1) / 2, grid2 = pairs < 4096 ? pairs : 4096;
#define LAUNCH_RMSF2(MV)                                                                                                     \
    hipLaunchKernelGGL(rmsnorm_fwd2_kernel<MV>, dim3(grid2), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (const elem_t*)w, \
                       (elem_t*)y, rstd, rows, h, eps)
        if (h <= 2048) LAUNCH_RMSF2(1); else if (h <= 4096) LAUNCH_RMSF2(2); else LAUNCH_RMSF2(4);
#undef LAUNCH_RMSF2
        AA_CHECK_LAUNCH("aa_rmsnorm_fwd");
        return AA_OK;
    }
    const int grid = rows < 4096 ? rows : 4096;
    hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)x, (const elem_t*)w, (elem_t*)y, rstd, rows, h, eps);
    AA_CHECK_LAUNCH("aa_rmsnorm_fwd");
    return AA_OK;
}

// y = rope(rmsnorm(x)) on [rows = tokens x heads, hd] rows (Qwen3 q_norm / k_norm + apply_rotary_pos_emb in one pass); hd in {64, 128, 256, 512}
// ldx: elements between the first head of consecutive tokens in x (heads * hd when x is dense; larger when x is a column slice of the fused q | k | v
// projection output); y is dense [rows, hd]
extern "C" int AA_FN(aa_rmsnorm_rope_fwd)(const void* x, long ldx, const void* w, void* y, float* rstd, long rows, int hd, float eps, const int* pos,
                                          const void* cos_t, const void* sin_t, int heads, void* stream) {
    AA_REQUIRE(hd == 64 || hd == 128 || hd == 256 || hd == 512, "aa_rmsnorm_rope_fwd: head_dim %d (64, 128, 256 or 512)", hd);
    AA_REQUIRE(heads > 0 && rows % heads == 0 && pos && cos_t && sin_t, "aa_rmsnorm_rope_fwd: rows %ld must be tokens x heads (%d), tables required", rows, heads);
    AA_REQUIRE(ldx >= (long)heads * hd && (ldx & 7) == 0, "aa_rmsnorm_rope_fwd: ldx %ld must be >= heads * head_dim and a multiple of 8", ldx);
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_RMSR(LPR)                                                                                                                       \
    do {                                                                                                                                       \
        const long nb = (rows + 2 * (256 / LPR) - 1) / (2 * (256 / LPR));                                                                      \
        hipLaunchKernelGGL((rmsnorm_fwd_small_kernel<LPR, true>), dim3((int)(nb < 16384 ? nb : 16384)), dim3(256), 0, st, (const elem_t*)x,    \
                           (const elem_t*)w, (elem_t*)y, rstd, rows, hd, eps, pos, (const elem_t*)cos_t, (const elem_t*)sin_t, heads, ldx);    \
    } while (0)
    if (hd == 64) LAUNCH_RMSR(8); else if (hd == 128) LAUNCH_RMSR(16); else if (hd == 256) LAUNCH_RMSR(32); else LAUNCH_RMSR(64);
#undef LAUNCH_RMSR
    AA_CHECK_LAUNCH("aa_rmsnorm_rope_fwd");
    return AA_OK;
}

// backward of the per-head norm with x and dx as column slices of the fused q | k | v buffers (see aa_rmsnorm_rope_fwd): dy dense [rows, hd]
extern "C" int AA_FN(aa_rmsnorm_heads_bwd)(const void* dy, const void* x, long ldx, const void* w, const float* rstd, void* dx, long lddx, float* dw, float* ws,
                                           int ws_rows, long rows, int hd, int heads, void* stream) {
    AA_REQUIRE(hd == 64 || hd == 128 || hd == 256 || hd == 512, "aa_rmsnorm_heads_bwd: head_dim %d (64, 128, 256 or 512)", hd);
    AA_REQUIRE(heads > 0 && rows % heads == 0 && ldx >= (long)heads * hd && lddx >= (long)heads * hd && ((ldx | lddx) & 7) == 0,
               "aa_rmsnorm_heads_bwd: rows %ld = tokens x heads (%d); ldx %ld / lddx %ld >= heads * head_dim, multiples of 8", rows, heads, ldx, lddx);
    AA_REQUIRE(dw == nullptr || (ws != nullptr && ws_rows > 0), "aa_rmsnorm_heads_bwd: dw needs a [ws_rows, hd] fp32 workspace");
    if (rows == 0) return AA_OK;
    hipStream_t st = (hipStream_t)stream;
    float* part = dw ? ws : nullptr;
    int g2 = 0;
#define LAUNCH_RMSHB(LPR)                                                                                                                 \
    do {                                                                                                                                  \
        const long nb = (rows + 2 * (256 / LPR) - 1) / (2 * (256 / LPR));                                                                 \
        g2 = (int)(nb < 2048 ? nb : 2048);                                                                                                \
        if (dw && g2 > ws_rows) g2 = ws_rows;                                                                                             \
        hipLaunchKernelGGL(rmsnorm_bwd_small_kernel<LPR>, dim3(g2), dim3(256), 0, st, (const elem_t*)dy, (const elem_t*)x, (const elem_t*)w, rstd, \
                           (elem_t*)dx, part, rows, hd, 0, heads, ldx, lddx);                                                             \
    } while (0)
    if (hd == 64) LAUNCH_RMSHB(8); else if (hd == 128) LAUNCH_RMSHB(16); else if (hd == 256) LAUNCH_RMSHB(32); else LAUNCH_RMSHB(64);
#undef LAUNCH_RMSHB
    if (dw) launch_reduce_rows(part, g2, hd, dw, st);
    AA_CHECK_LAUNCH("aa_rmsnorm_heads_bwd");
    return AA_OK;
}

extern "C" int AA_FN(aa_rmsnorm_bwd)(const void* dy, const void* x, const void* w, const float* rstd,
                              void* dx, float* dw, float* ws, int ws_rows, int rows, int h,
                              int add_to_dx, void* stream) {
    AA_REQUIRE(rows >= 0 && h > 0 && (h & 7) == 0 && h <= 8 * 256 * 8,
               "aa_rmsnorm_bwd: hidden %d must be a multiple of 8 and <= 16384", h);
    AA_REQUIRE(dw == nullptr || (ws != nullptr && ws_rows > 0), "aa_rmsnorm_bwd: dw needs a [ws_rows, h] fp32 workspace");
    if (rows == 0) return AA_OK;
    int grid = rows < 1024 ? rows : 1024;
    if (dw && grid > ws_rows) grid = ws_rows;
    float* part = dw ? ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (h <= 512) {
        int g2 = 0;
#define LAUNCH_RMSB_SMALL(LPR)                                                                                                          \
    do {                                                                                                                                \
        const long nb = ((long)rows + 2 * (256 / LPR) - 1) / (2 * (256 / LPR));                                                         \
        g2 = (int)(nb < 2048 ? nb : 2048);                                                                                              \
        if (dw && g2 > ws_rows) g2 = ws_rows;                                                                                           \
        hipLaunchKernelGGL(rmsnorm_bwd_small_kernel<LPR>, dim3(g2), dim3(256), 0, st, (const elem_t*)dy, (const elem_t*)x, (const elem_t*)w, rstd, \
                           (elem_t*)dx, part, (long)rows, h, add_to_dx);                                                                \
    } while (0)
        if (h <= 64) LAUNCH_RMSB_SMALL(8); else if (h <= 128) LAUNCH_RMSB_SMALL(16); else if (h <= 256) LAUNCH_RMSB_SMALL(32); else LAUNCH_RMSB_SMALL(64);
#undef LAUNCH_RMSB_SMALL
        if (dw) launch_reduce_rows(part, g2, h, dw, st);
        AA_CHECK_LAUNCH("aa_rmsnorm_bwd");
        return AA_OK;
    }
    // one round of co-resident blocks: the two-row kernel holds 2 x 3 row vectors per thread (168 VGPRs at h = 4096: three blocks per CU), and a grid
    // larger than what is resident would run a part-empty second round; every block walks rows / grid rows, so a smaller grid costs nothing
    // (the slot count is cached per DEVICE -- ADVICE r5: one process-wide static took the first caller's device for every later one; the race of two host
    // threads on a first call is benign, both compute the same value.  dx is bit-identical whatever the grid; the dw partial sums are grouped by the grid,
    // so dw is equal up to fp32 summation order across grids.)
#define LAUNCH_RMSB(MV)                                                                             \
    do {                                                                                            \
        static int slots_of[AA_MAX_DEVICES] = {0};                                                  \
        int dev = 0;                                                                                \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AA_MAX_DEVICES) dev = -1;         \
        int slots = dev >= 0 ? slots_of[dev] : 0;                                                   \
        if (slots == 0) {                                                                           \
            int per_cu = 0, cus = 0;                                                                \
            if (dev < 0 || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmsnorm_bwd_kernel<MV>, 256, 0) != hipSuccess ||                        \
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0)                         \
                slots = 1024;                                                                       \
            else                                                                                    \
                slots = per_cu * cus;                                                               \
            if (dev >= 0) slots_of[dev] = slots;                                                    \
        }                                                                                           \
        if (grid > slots) grid = slots;                                                             \
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<MV>, dim3(grid), dim3(256), 0, st, (const elem_t*)dy, \
                           (const elem_t*)x, (const elem_t*)w, rstd, (elem_t*)dx, part, rows, h, add_to_dx); \
    } while (0)
    if (h <= 2048) LAUNCH_RMSB(1);
    else if (h <= 4096) LAUNCH_RMSB(2);
    else if (h <= 8192) LAUNCH_RMSB(4);
    else LAUNCH_RMSB(8);
#undef LAUNCH_RMSB
    if (dw) launch_reduce_rows(part, grid, h, dw, st);
    AA_CHECK_LAUNCH("aa_rmsnorm_bwd");
    return AA_OK;
}

// ================================================================== LayerNorm
// torch F.layer_norm on bf16: fp32 stats, y = bf16((x-mean)*rstd*w + b)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const elem_t* __restrict__ x,
                                                            const elem_t* __restrict__ w,
                                                            const elem_t* __restrict__ b,
                                                            elem_t* __restrict__ y,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows,
                                                            int h, float eps) {
    __shared__ float red[8];
    const int nv = h >> 3;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const elem_t* xr = x + row * h;
        float s = 0.f;
        for (int i = threadIdx.x; i < nv; i += 256) {
            ev8 v = *reinterpret_cast<const ev8*>(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += e2f(v[j]);
        }
        const float mean = block_sum<256>(s, red) / (float)h;
        float ss = 0.f;
        for (int i = threadIdx.x; i < nv; i += 256) {
            ev8 v = *reinterpret_cast<const ev8*>(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = e2f(v[j]) - mean; ss += d * d; }
        }
        const float rstd = rsqrtf(block_sum<256>(ss, red) / (float)h + eps);
        if (threadIdx.x == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        elem_t* yr = y + row * h;
        for (int i = threadIdx.x; i < nv; i += 256) {
            ev8 v = *reinterpret_cast<const ev8*>(xr + i * 8);
            ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
            ev8 bv = *reinterpret_cast<const ev8*>(b + i * 8);
            ev8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = f2e((e2f(v[j]) - mean) * rstd * e2f(wv[j]) + e2f(bv[j]));
            *reinterpret_cast<ev8*>(yr + i * 8) = o;
        }
    }
}

// dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)), dxhat = dy*w ; dw += dy*xhat ; db += dy
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const elem_t* __restrict__ dy,
                                                            const elem_t* __restrict__ x,
                                                            const elem_t* __restrict__ w,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            elem_t* __restrict__ dx,
                                                            float* __restrict__ dw_part,
                                                            float* __restrict__ db_part, int rows, int h,
                                                            int add_to_dx) {
    __shared__ float red[8];
    const int nv = h >> 3;
    float dwacc[MAXV][8], dbacc[MAXV][8];
#pragma unroll
    for (int a = 0; a < MAXV; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwacc[a][j] = 0.f; dbacc[a][j] = 0.f; }
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const elem_t* xr = x + row * h;
        const elem_t* gr = dy + row * h;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < MAXV; ++a) {
            const int i = threadIdx.x + a * 256;
            if (i < nv) {
                ev8 xv = *reinterpret_cast<const ev8*>(xr + i * 8);
                ev8 gv = *reinterpret_cast<const ev8*>(gr + i * 8);
                ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (e2f(xv[j]) - mean) * rstd;
                    const float g = e2f(gv[j]);
                    const float dxh = g * e2f(wv[j]);
                    s1 += dxh; s2 += dxh * xh;
                    dwacc[a][j] += g * xh; dbacc[a][j] += g;
                }
            }
        }
        s1 = block_sum<256>(s1, red) / (float)h;
        s2 = block_sum<256>(s2, red) / (float)h;
        elem_t* dxr = dx + row * h;
#pragma unroll
        for (int a = 0; a < MAXV; ++a) {
            const int i = threadIdx.x + a * 256;
            if (i < nv) {
                ev8 xv = *reinterpret_cast<const ev8*>(xr + i * 8);
                ev8 gv = *reinterpret_cast<const ev8*>(gr + i * 8);
                ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
                ev8 o;
                if (add_to_dx) o = *reinterpret_cast<const ev8*>(dxr + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (e2f(xv[j]) - mean) * rstd;
                    float d = rstd * (e2f(gv[j]) * e2f(wv[j]) - s1 - xh * s2);
                    if (add_to_dx) d += e2f(o[j]);
                    o[j] = f2e(d);
                }
                *reinterpret_cast<ev8*>(dxr + i * 8) = o;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < MAXV; ++a) {
        const int i = threadIdx.x + a * 256;
        if (i < nv) {
            if (dw_part) {
                float* pr = dw_part + (long)blockIdx.x * h + i * 8;
                *reinterpret_cast<f32x4*>(pr) = f32x4{dwacc[a][0], dwacc[a][1], dwacc[a][2], dwacc[a][3]};
                *reinterpret_cast<f32x4*>(pr + 4) = f32x4{dwacc[a][4], dwacc[a][5], dwacc[a][6], dwacc[a][7]};
            }
            if (db_part) {
                float* pr = db_part + (long)blockIdx.x * h + i * 8;
                *reinterpret_cast<f32x4*>(pr) = f32x4{dbacc[a][0], dbacc[a][1], dbacc[a][2], dbacc[a][3]};
                *reinterpret_cast<f32x4*>(pr + 4) = f32x4{dbacc[a][4], dbacc[a][5], dbacc[a][6], dbacc[a][7]};
            }
        }
    }
}

extern "C" int AA_FN(aa_layernorm_fwd)(const void* x, const void* w, const void* b, void* y, float* mean,
                                float* rstd, int rows, int h, float eps, void* stream) {
    AA_REQUIRE(rows >= 0 && h > 0 && (h & 7) == 0, "aa_layernorm_fwd: hidden %d must be a multiple of 8", h);
    if (rows == 0) return AA_OK;
    const int grid = rows < 4096 ? rows : 4096;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)x, (const elem_t*)w, (const elem_t*)b, (elem_t*)y, mean, rstd,
                       rows, h, eps);
    AA_CHECK_LAUNCH("aa_layernorm_fwd");
    return AA_OK;
}

extern "C" int AA_FN(aa_layernorm_bwd)(const void* dy, const void* x, const void* w, const float* mean,
                                const float* rstd, void* dx, float* dw, float* db, float* ws,
                                int ws_rows, int rows, int h, int add_to_dx, void* stream) {
    AA_REQUIRE(rows >= 0 && h > 0 && (h & 7) == 0 && h <= 8 * 256 * 4,
               "aa_layernorm_bwd: hidden %d must be a multiple of 8 and <= 8192", h);
    AA_REQUIRE((dw == nullptr && db == nullptr) || (ws != nullptr && ws_rows > 0),
               "aa_layernorm_bwd: dw/db need a [2, ws_rows, h] fp32 workspace");
    if (rows == 0) return AA_OK;
    int grid = rows < 1024 ? rows : 1024;
    if ((dw || db) && grid > ws_rows) grid = ws_rows;
    float* dwp = dw ? ws : nullptr;
    float* dbp = db ? ws + (long)ws_rows * h : nullptr;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_LNB(MV)                                                                              \
    hipLaunchKernelGGL(layernorm_bwd_kernel<MV>, dim3(grid), dim3(256), 0, st, (const elem_t*)dy,   \
                       (const elem_t*)x, (const elem_t*)w, mean, rstd, (elem_t*)dx, dwp, dbp, rows, h, \
                       add_to_dx)
    if (h <= 2048) LAUNCH_LNB(1);
    else if (h <= 4096) LAUNCH_LNB(2);
    else LAUNCH_LNB(4);
#undef LAUNCH_LNB
    if (dw) launch_reduce_rows(dwp, grid, h, dw, st);
    if (db) launch_reduce_rows(dbp, grid, h, db, st);
    AA_CHECK_LAUNCH("aa_layernorm_bwd");
    return AA_OK;
}

// ================================================================== RoPE (in place on a [M, ld] buffer)
// hf:models/llama/modeling_llama.py:113-160: cos/sin are tables [maxpos, hd/2] in the activation dtype;
// q' = bf16(bf16(q*cos) + bf16(rotate_half(q)*sin)), half-split layout.  Applied to `nheads` heads of rotary width hd
// starting at column col0 of every row, consecutive heads `head_stride` columns apart (== hd unless the heads are
// zero-padded, e.g. Qwen2-VL's 80-wide vision heads stored 128 wide); position of row r is pos[r].
// inverse != 0 applies the transpose rotation (backward).
// precise != 0: hf:models/qwen2_vl/modeling_qwen2_vl.py:225-236 apply_rotary_pos_emb_vision -- fp32 tables (cos_t / sin_t
// are float*), fp32 arithmetic, one rounding of the result.
__global__ __launch_bounds__(256) void rope_kernel(elem_t* __restrict__ buf, long ld, int col0,
                                                   int nheads, int hd, int head_stride,
                                                   const int* __restrict__ pos,
                                                   const void* __restrict__ cos_v,
                                                   const void* __restrict__ sin_v, long rows,
                                                   int inverse, int precise) {
    const int half = hd >> 1;
    const int vec_per_head = half >> 3;  // 8 elements per lane from each half
    const long total = rows * nheads * vec_per_head;
    const elem_t* cos_t = reinterpret_cast<const elem_t*>(cos_v);
    const elem_t* sin_t = reinterpret_cast<const elem_t*>(sin_v);
    const float* cos_f = reinterpret_cast<const float*>(cos_v);
    const float* sin_f = reinterpret_cast<const float*>(sin_v);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int v = (int)(idx % vec_per_head);
        const long t = idx / vec_per_head;
        const int head = (int)(t % nheads);
        const long row = t / nheads;
        const int p = pos[row];
        elem_t* base = buf + row * ld + col0 + (long)head * head_stride + v * 8;
        ev8 x1 = *reinterpret_cast<const ev8*>(base);
        ev8 x2 = *reinterpret_cast<const ev8*>(base + half);
        ev8 o1, o2;
        if (precise) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = e2f(x1[j]), b = e2f(x2[j]);
                const float cc = cos_f[(long)p * half + v * 8 + j];
                const float ss = inverse ? -sin_f[(long)p * half + v * 8 + j] : sin_f[(long)p * half + v * 8 + j];
                o1[j] = f2e(a * cc - b * ss);
                o2[j] = f2e(b * cc + a * ss);
            }
        } else {
            ev8 c = *reinterpret_cast<const ev8*>(cos_t + (long)p * half + v * 8);
            ev8 s = *reinterpret_cast<const ev8*>(sin_t + (long)p * half + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = e2f(x1[j]), b = e2f(x2[j]), cc = e2f(c[j]);
                const float ss = inverse ? -e2f(s[j]) : e2f(s[j]);
                o1[j] = f2e(ernd(a * cc) + ernd(-b * ss));
                o2[j] = f2e(ernd(b * cc) + ernd(a * ss));
            }
        }
        *reinterpret_cast<ev8*>(base) = o1;
        *reinterpret_cast<ev8*>(base + half) = o2;
    }
}

extern "C" int AA_FN(aa_rope_inplace)(void* buf, long ld, int col0, int nheads, int hd, const int* pos,
                               const void* cos_t, const void* sin_t, long rows, int inverse, int head_stride,
                               int precise, void* stream) {
    AA_REQUIRE(hd > 0 && (hd % 16) == 0 && nheads > 0, "aa_rope_inplace: head_dim %d must be a multiple of 16", hd);
    AA_REQUIRE((ld & 7) == 0 && (col0 & 7) == 0, "aa_rope_inplace: ld/col0 must be multiples of 8");
    if (head_stride <= 0) head_stride = hd;
    AA_REQUIRE(head_stride >= hd && (head_stride & 7) == 0, "aa_rope_inplace: head_stride %d must be >= head_dim and a multiple of 8", head_stride);
    if (rows == 0) return AA_OK;
    const long total = rows * nheads * (hd >> 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (elem_t*)buf, ld,
                       col0, nheads, hd, head_stride, pos, cos_t, sin_t, rows, inverse, precise);
    AA_CHECK_LAUNCH("aa_rope_inplace");
    return AA_OK;
}

// hf:models/qwen2_vl/modeling_qwen2_vl.py:156-222 (Qwen2VLRotaryEmbedding + apply_multimodal_rotary_pos_emb): per-token
// cos/sin rows for multimodal RoPE.  pos3 = [3, rows] (temporal, height, width position of every token); frequency f of
// the half-dim uses component 0 for f < sec0, 1 for f < sec0 + sec1, else 2.  Tables are [rows, half] in the activation
// dtype (fp32 angle, fp32 cos/sin, one rounding) and are consumed by aa_rope_inplace with pos[row] = row.
__global__ __launch_bounds__(256) void mrope_tables_kernel(const int* __restrict__ pos3, long rows,
                                                           const float* __restrict__ inv_freq, int half, int sec0,
                                                           int sec1, elem_t* __restrict__ cos_t,
                                                           elem_t* __restrict__ sin_t) {
    const long total = rows * half;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int f = (int)(idx % half);
        const long r = idx / half;
        const int c = f < sec0 ? 0 : (f < sec0 + sec1 ? 1 : 2);
        const float ang = inv_freq[f] * (float)pos3[(long)c * rows + r];
        cos_t[idx] = f2e(cosf(ang));
        sin_t[idx] = f2e(sinf(ang));
    }
}
extern "C" int AA_FN(aa_mrope_tables)(const int* pos3, long rows, const float* inv_freq, int half, int sec0, int sec1,
                                      void* cos_t, void* sin_t, void* stream) {
    AA_REQUIRE(half > 0 && sec0 >= 0 && sec1 >= 0 && sec0 + sec1 <= half, "aa_mrope_tables: bad sections %d/%d of %d", sec0, sec1, half);
    if (rows == 0) return AA_OK;
    const long total = rows * half;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(mrope_tables_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pos3, rows, inv_freq, half,
                       sec0, sec1, (elem_t*)cos_t, (elem_t*)sin_t);
    AA_CHECK_LAUNCH("aa_mrope_tables");
    return AA_OK;
}

// ================================================================== gated / pointwise activations
// act codes shared with the GEMM epilogue
#define AA_ACT_NONE 0
#define AA_ACT_GELU 1        // erf GELU (LLaVA projector)
#define AA_ACT_QUICK_GELU 2  // x*sigmoid(1.702x) (CLIP)
#define AA_ACT_RELU 3        // OPT
#define AA_ACT_SILU 4

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case AA_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case AA_ACT_QUICK_GELU: return x / (1.f + expf(-1.702f * x));
        case AA_ACT_RELU: return x > 0.f ? x : 0.f;
        case AA_ACT_SILU: return x / (1.f + expf(-x));
        default: return x;
    }
}
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case AA_ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case AA_ACT_QUICK_GELU: {
            const float s = 1.f / (1.f + expf(-1.702f * x));
            return s + 1.702f * x * s * (1.f - s);
        }
        case AA_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case AA_ACT_SILU: {
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        default: return 1.f;
    }
}

// SwiGLU: gu = [gate | up] as [M, 2F] ; act = bf16(bf16(silu(gate)) * up)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const elem_t* __restrict__ gu,
                                                         elem_t* __restrict__ out, long M, int F) {
    const int nv = F >> 3;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / nv;
        const int v = (int)(idx % nv);
        ev8 g = *reinterpret_cast<const ev8*>(gu + row * 2 * F + v * 8);
        ev8 u = *reinterpret_cast<const ev8*>(gu + row * 2 * F + F + v * 8);
        ev8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = e2f(g[j]);
            o[j] = f2e(ernd(gf * aa_sigmoid<AA_ELEM_PRECISE>(gf)) * e2f(u[j]));
        }
        *reinterpret_cast<ev8*>(out + row * F + v * 8) = o;
    }
}
// dgu = [dact*up*silu'(gate) | dact*silu(gate)]
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const elem_t* __restrict__ gu,
                                                         const elem_t* __restrict__ dact,
                                                         elem_t* __restrict__ dgu, long M, int F) {
    const int nv = F >> 3;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / nv;
        const int v = (int)(idx % nv);
        ev8 g = *reinterpret_cast<const ev8*>(gu + row * 2 * F + v * 8);
        ev8 u = *reinterpret_cast<const ev8*>(gu + row * 2 * F + F + v * 8);
        ev8 d = *reinterpret_cast<const ev8*>(dact + row * F + v * 8);
        ev8 og, ou;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = e2f(g[j]), uf = e2f(u[j]), df = e2f(d[j]);
            const float s = aa_sigmoid<AA_ELEM_PRECISE>(gf);
            og[j] = f2e(df * uf * s * (1.f + gf * (1.f - s)));
            ou[j] = f2e(df * gf * s);
        }
        *reinterpret_cast<ev8*>(dgu + row * 2 * F + v * 8) = og;
        *reinterpret_cast<ev8*>(dgu + row * 2 * F + F + v * 8) = ou;
    }
}

extern "C" int AA_FN(aa_swiglu_fwd)(const void* gate_up, void* out, long M, int F, void* stream) {
    AA_REQUIRE(F > 0 && (F & 7) == 0, "aa_swiglu_fwd: ffn %d must be a multiple of 8", F);
    if (M == 0) return AA_OK;
    const long total = M * (F >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)gate_up, (elem_t*)out, M, F);
    AA_CHECK_LAUNCH("aa_swiglu_fwd");
    return AA_OK;
}
extern "C" int AA_FN(aa_swiglu_bwd)(const void* gate_up, const void* dact, void* dgate_up, long M, int F,
                             void* stream) {
    AA_REQUIRE(F > 0 && (F & 7) == 0, "aa_swiglu_bwd: ffn %d must be a multiple of 8", F);
    if (M == 0) return AA_OK;
    const long total = M * (F >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)gate_up, (const elem_t*)dact, (elem_t*)dgate_up, M, F);
    AA_CHECK_LAUNCH("aa_swiglu_bwd");
    return AA_OK;
}

// pointwise activation backward: dx = dy * act'(pre)   (pre = saved pre-activation, bf16)
__global__ __launch_bounds__(256) void act_bwd_kernel(const elem_t* __restrict__ pre,
                                                      const elem_t* __restrict__ dy,
                                                      elem_t* __restrict__ dx, long n8, int act) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (long)gridDim.x * 256) {
        ev8 p = *reinterpret_cast<const ev8*>(pre + idx * 8);
        ev8 d = *reinterpret_cast<const ev8*>(dy + idx * 8);
        ev8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(d[j]) * act_grad(e2f(p[j]), act));
        *reinterpret_cast<ev8*>(dx + idx * 8) = o;
    }
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const elem_t* __restrict__ x,
                                                      elem_t* __restrict__ y, long n8, int act) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (long)gridDim.x * 256) {
        ev8 p = *reinterpret_cast<const ev8*>(x + idx * 8);
        ev8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2e(act_fwd(e2f(p[j]), act));
        *reinterpret_cast<ev8*>(y + idx * 8) = o;
    }
}
extern "C" int AA_FN(aa_act_fwd)(const void* x, void* y, long n, int act, void* stream) {
    AA_REQUIRE((n & 7) == 0, "aa_act_fwd: element count %ld must be a multiple of 8", n);
    if (n == 0) return AA_OK;
    const long n8 = n >> 3;
    const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x,
                       (elem_t*)y, n8, act);
    AA_CHECK_LAUNCH("aa_act_fwd");
    return AA_OK;
}
extern "C" int AA_FN(aa_act_bwd)(const void* pre, const void* dy, void* dx, long n, int act, void* stream) {
    AA_REQUIRE((n & 7) == 0, "aa_act_bwd: element count %ld must be a multiple of 8", n);
    if (n == 0) return AA_OK;
    const long n8 = n >> 3;
    const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)pre, (const elem_t*)dy, (elem_t*)dx, n8, act);
    AA_CHECK_LAUNCH("aa_act_bwd");
    return AA_OK;
}

// y = a + b (bf16, rounded once) -- residual adds that are not fused into a GEMM epilogue
__global__ __launch_bounds__(256) void add_kernel(const elem_t* __restrict__ a,
                                                  const elem_t* __restrict__ b,
                                                  elem_t* __restrict__ y, long n8) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (long)gridDim.x * 256) {
        ev8 p = *reinterpret_cast<const ev8*>(a + idx * 8);
        ev8 q = *reinterpret_cast<const ev8*>(b + idx * 8);
        ev8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(p[j]) + e2f(q[j]));
        *reinterpret_cast<ev8*>(y + idx * 8) = o;
    }
}
extern "C" int AA_FN(aa_add)(const void* a, const void* b, void* y, long n, void* stream) {
    AA_REQUIRE((n & 7) == 0, "aa_add: element count %ld must be a multiple of 8", n);
    if (n == 0) return AA_OK;
    const long n8 = n >> 3;
    const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipLaunchKernelGGL(add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)a,
                       (const elem_t*)b, (elem_t*)y, n8);
    AA_CHECK_LAUNCH("aa_add");
    return AA_OK;
}

// ================================================================== embedding gather + image scatter
// hf:models/llava/modeling_llava.py:234-248 : inputs_embeds = embed(ids); rows where ids ==
// image_token_id are replaced, in row-major order of occurrence, by the projector outputs.
// slot[t] = running index among image tokens, or -1.  Single-workgroup scan (n <= a few 10k).
__global__ __launch_bounds__(1024) void image_slot_kernel(const int64_t* __restrict__ ids, int n,
                                                          int64_t image_token_id,
                                                          int* __restrict__ slot,
                                                          int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int f = (i < n && ids[i] == image_token_id) ? 1 : 0;
        const unsigned long long bal = __ballot(f);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int c = carry;
        if (i < n) slot[i] = f ? (c + woff + prefix) : -1;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            carry = c + tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && count) *count = carry;
}

#ifndef AA_ELEM_F32   // integer work: one instantiation
extern "C" int aa_image_slot_index(const int64_t* ids, int n, int64_t image_token_id, int* slot,
                                   int* count, void* stream) {
    AA_REQUIRE(n >= 0, "aa_image_slot_index: n must be >= 0");
    hipLaunchKernelGGL(image_slot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, n,
                       image_token_id, slot, count);
    AA_CHECK_LAUNCH("aa_image_slot_index");
    return AA_OK;
}
#endif

// out[t,:] = slot[t] >= 0 ? feat[slot[t],:] : E[ids[t],:] (+ pos_emb[pos[t],:] when given, OPT)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids,
                                                        const int* __restrict__ slot,
                                                        const elem_t* __restrict__ E,
                                                        const elem_t* __restrict__ feat,
                                                        const int* __restrict__ pos,
                                                        const elem_t* __restrict__ P,
                                                        elem_t* __restrict__ out, long n, int h,
                                                        int vocab) {
    const int nv = h >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n * nv; idx += (long)gridDim.x * 256) {
        const long t = idx / nv;
        const int v = (int)(idx % nv);
        const int s = slot ? slot[t] : -1;
        ev8 r;
        if (s >= 0) {
            r = *reinterpret_cast<const ev8*>(feat + (long)s * h + v * 8);
        } else {
            long id = ids[t];
            if (id < 0 || id >= vocab) id = 0;  // host validates; never fault
            r = *reinterpret_cast<const ev8*>(E + id * h + v * 8);
        }
        if (P) {
            ev8 p = *reinterpret_cast<const ev8*>(P + (long)pos[t] * h + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = f2e(e2f(r[j]) + e2f(p[j]));
        }
        *reinterpret_cast<ev8*>(out + t * h + v * 8) = r;
    }
}
// backward: dfeat[slot] = dx (gather, unique rows) ; dE[ids] += dx (fp32 atomics) ; dP[pos] += dx
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids,
                                                        const int* __restrict__ slot,
                                                        const int* __restrict__ pos,
                                                        const elem_t* __restrict__ dx,
                                                        float* __restrict__ dE,
                                                        elem_t* __restrict__ dfeat,
                                                        float* __restrict__ dP, long n, int h,
                                                        int vocab) {
    const int nv = h >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n * nv; idx += (long)gridDim.x * 256) {
        const long t = idx / nv;
        const int v = (int)(idx % nv);
        const int s = slot ? slot[t] : -1;
        ev8 r = *reinterpret_cast<const ev8*>(dx + t * h + v * 8);
        if (s >= 0) {
            if (dfeat) *reinterpret_cast<ev8*>(dfeat + (long)s * h + v * 8) = r;
        } else if (dE) {
            const long id = ids[t];
            if (id >= 0 && id < vocab) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(dE + id * h + v * 8 + j, e2f(r[j]));
            }
        }
        if (dP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(dP + (long)pos[t] * h + v * 8 + j, e2f(r[j]));
        }
    }
}

extern "C" int AA_FN(aa_embed_fwd)(const int64_t* ids, const int* slot, const void* E, const void* feat,
                            const int* pos, const void* P, void* out, long n, int h, int vocab,
                            void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_embed_fwd: hidden %d must be a multiple of 8", h);
    AA_REQUIRE((slot == nullptr) || (feat != nullptr), "aa_embed_fwd: slot given without features");
    if (n == 0) return AA_OK;
    const long total = n * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, slot,
                       (const elem_t*)E, (const elem_t*)feat, pos, (const elem_t*)P, (elem_t*)out, n, h,
                       vocab);
    AA_CHECK_LAUNCH("aa_embed_fwd");
    return AA_OK;
}
extern "C" int AA_FN(aa_embed_bwd)(const int64_t* ids, const int* slot, const int* pos, const void* dx,
                            float* dE, void* dfeat, float* dP, long n, int h, int vocab,
                            void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_embed_bwd: hidden %d must be a multiple of 8", h);
    if (n == 0) return AA_OK;
    const long total = n * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, slot, pos,
                       (const elem_t*)dx, dE, (elem_t*)dfeat, dP, n, h, vocab);
    AA_CHECK_LAUNCH("aa_embed_bwd");
    return AA_OK;
}

// ================================================================== bf16 2-D transpose  out[C,R] = in[R,C]^T
// 64x64 tile through LDS (padded rows): coalesced 128-B row segments on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const elem_t* __restrict__ in, long ldi,
                                                        elem_t* __restrict__ out, long ldo, int R,
                                                        int C) {
    __shared__ elem_t tile[64][66];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(long)r * ldi + c] : (elem_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(long)c * ldo + r] = tile[tx][i];
    }
}
extern "C" int AA_FN2(aa_transpose_bf16, aa_transpose_f32)(const void* in, long ldi, void* out, long ldo, int R, int C,
                                 void* stream) {
    AA_REQUIRE(R >= 0 && C >= 0 && ldi >= C && ldo >= R, "aa_transpose_bf16: bad shape R=%d C=%d", R, C);
    if (R == 0 || C == 0) return AA_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3(aa_cdiv(C, 64), aa_cdiv(R, 64)), dim3(256), 0,
                       (hipStream_t)stream, (const elem_t*)in, ldi, (elem_t*)out, ldo, R, C);
    AA_CHECK_LAUNCH("aa_transpose_bf16");
    return AA_OK;
}

// ================================================================== column sum (bias gradient)
// out[c] += sum_r in[r, c]   (fp32 atomics; each block reduces a 256-row slab of 64*8 columns)
__global__ __launch_bounds__(256) void colsum_kernel(const elem_t* __restrict__ in, long ld, long R,
                                                     int C, float* __restrict__ out) {
    // thread layout: 64 lanes x 8 columns wide (512 columns per block), 4 row groups
    __shared__ float part[4][512];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    const long r0 = (long)blockIdx.y * 256;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        for (long r = r0 + rg; r < R && r < r0 + 256; r += 4) {
            ev8 v = *reinterpret_cast<const ev8*>(in + r * ld + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += e2f(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[rg][lane * 8 + j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int cc = blockIdx.x * 512 + i;
        if (cc < C) atomicAdd(out + cc, part[0][i] + part[1][i] + part[2][i] + part[3][i]);
    }
}
extern "C" int AA_FN2(aa_colsum_bf16, aa_colsum_f32)(const void* in, long ld, long R, int C, float* out, void* stream) {
    AA_REQUIRE(C > 0 && (C & 7) == 0 && (ld & 7) == 0, "aa_colsum_bf16: C=%d / ld must be multiples of 8", C);
    if (R == 0) return AA_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3(aa_cdiv(C, 512), aa_cdiv(R, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const elem_t*)in, ld, R, C, out);
    AA_CHECK_LAUNCH("aa_colsum_bf16");
    return AA_OK;
}

// ================================================================== CLIP patch im2col
// hf:models/clip/modeling_clip.py:138-218: Conv2d(3->h, k=s=P, no bias) == GEMM over
// patches[n*G*G + gy*G + gx, c*P*P + py*P + px] = pixel[n, c, gy*P+py, gx*P+px]; K padded to Kp with 0.
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_kernel(const TIN* __restrict__ pix,
                                                     elem_t* __restrict__ out, int n_img, int Cc,
                                                     int H, int P, int Kp) {
    const int G = H / P;
    const int K = Cc * P * P;
    const long total = (long)n_img * G * G * Kp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % Kp);
        const long prow = idx / Kp;
        float v = 0.f;
        if (k < K) {
            const int px = k % P, py = (k / P) % P, c = k / (P * P);
            const int gx = (int)(prow % G), gy = (int)((prow / G) % G);
            const long n = prow / ((long)G * G);
            const long src = ((n * Cc + c) * H + (gy * P + py)) * H + gx * P + px;
            if constexpr (sizeof(TIN) == 2) v = bf2f(pix[src]); else v = pix[src];
        }
        out[idx] = f2e(v);
    }
}
extern "C" int AA_FN(aa_patch_im2col)(const void* pixels, int pix_dtype, void* out, int n_img, int channels,
                               int image_size, int patch, int Kp, void* stream) {
    AA_REQUIRE(image_size % patch == 0 && Kp >= channels * patch * patch,
               "aa_patch_im2col: image %d / patch %d / Kp %d mismatch", image_size, patch, Kp);
    if (n_img == 0) return AA_OK;
    const int G = image_size / patch;
    const long total = (long)n_img * G * G * Kp;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (pix_dtype == 0)
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)pixels, (elem_t*)out, n_img, channels, image_size, patch, Kp);
    else
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)pixels, (elem_t*)out, n_img, channels, image_size, patch, Kp);
    AA_CHECK_LAUNCH("aa_patch_im2col");
    return AA_OK;
}

#ifndef AA_ELEM_F32
// ================================================================== casts / fills
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in,
                                                          elem_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = f2e(in[i]);
}
extern "C" int aa_f32_to_bf16(const float* in, void* out, long n, void* stream) {
    if (n == 0) return AA_OK;
    const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in,
                       (elem_t*)out, n);
    AA_CHECK_LAUNCH("aa_f32_to_bf16");
    return AA_OK;
}
#endif

// CLIP embeddings: x[n, 0, :] = cls + pos[0]; x[n, 1+p, :] = patch[n*G2+p, :] + pos[1+p]
__global__ __launch_bounds__(256) void clip_embed_kernel(const elem_t* __restrict__ patch,
                                                         const elem_t* __restrict__ cls,
                                                         const elem_t* __restrict__ pos,
                                                         elem_t* __restrict__ out, int n_img, int G2,
                                                         int h) {
    const int nv = h >> 3;
    const long total = (long)n_img * (G2 + 1) * nv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int v = (int)(idx % nv);
        const long t = idx / nv;
        const int p = (int)(t % (G2 + 1));
        const long n = t / (G2 + 1);
        ev8 a = (p == 0) ? *reinterpret_cast<const ev8*>(cls + v * 8)
                           : *reinterpret_cast<const ev8*>(patch + (n * G2 + p - 1) * h + v * 8);
        ev8 b = *reinterpret_cast<const ev8*>(pos + (long)p * h + v * 8);
        ev8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(a[j]) + e2f(b[j]));
        *reinterpret_cast<ev8*>(out + t * h + v * 8) = o;
    }
}
extern "C" int AA_FN(aa_clip_embed)(const void* patch, const void* cls, const void* pos, void* out,
                             int n_img, int G2, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_clip_embed: hidden %d must be a multiple of 8", h);
    if (n_img == 0) return AA_OK;
    const long total = (long)n_img * (G2 + 1) * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(clip_embed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)patch, (const elem_t*)cls, (const elem_t*)pos, (elem_t*)out, n_img,
                       G2, h);
    AA_CHECK_LAUNCH("aa_clip_embed");
    return AA_OK;
}

// ================================================================== Conv1d (k = 3) as im2col + GEMM, AvgPool1d(2)
// hf:models/qwen2_audio/modeling_qwen2_audio.py:315-316, 372-373 (Whisper front-end): conv1 = Conv1d(mel, d, 3, padding 1),
// conv2 = Conv1d(d, d, 3, stride 2, padding 1).  col[(b, t), ci * 3 + k] = x[b, ci, t * stride + k - 1] (0 outside), so
// the HF weight [co, ci, 3] is used as stored ([co, ci * 3 + k]).  x element (b, ci, tin) lives at b*sb + ci*sc + tin*st:
// channels-first input_features (sb = C*Tin, sc = Tin, st = 1) or token-major activations (sb = Tin*C, sc = 1, st = C).
template <typename TIN>
__global__ __launch_bounds__(256) void conv1d_im2col_kernel(const TIN* __restrict__ x, long sb, long sc, long st,
                                                            elem_t* __restrict__ col, int B, int C, int Tin, int Tout,
                                                            int stride) {
    const long total = (long)B * Tout * C * 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % 3);
        const int ci = (int)((idx / 3) % C);
        const long row = idx / (3L * C);
        const int t = (int)(row % Tout);
        const long b = row / Tout;
        const int tin = t * stride + k - 1;
        float v = 0.f;
        if (tin >= 0 && tin < Tin) {
            const TIN r = x[b * sb + ci * sc + (long)tin * st];
            if constexpr (sizeof(TIN) == 2) v = bf2f(r); else v = r;
        }
        col[idx] = f2e(v);
    }
}
extern "C" int AA_FN(aa_conv1d_im2col)(const void* x, int x_dtype, long sb, long sc, long st, void* col, int B, int C,
                                       int Tin, int Tout, int stride, void* stream) {
    AA_REQUIRE(B >= 0 && C > 0 && Tin > 0 && Tout > 0 && stride > 0 && (Tin + 2 - 3) / stride + 1 == Tout,
               "aa_conv1d_im2col: Tin=%d stride=%d does not give Tout=%d (k=3, padding=1)", Tin, stride, Tout);
    if (B == 0) return AA_OK;
    const long total = (long)B * Tout * C * 3;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (x_dtype == 0)
        hipLaunchKernelGGL(conv1d_im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, sb, sc, st,
                           (elem_t*)col, B, C, Tin, Tout, stride);
    else
        hipLaunchKernelGGL(conv1d_im2col_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, sb, sc, st,
                           (elem_t*)col, B, C, Tin, Tout, stride);
    AA_CHECK_LAUNCH("aa_conv1d_im2col");
    return AA_OK;
}
// backward to a token-major input: dx[(b, tin), ci] = sum_k dcol[(b, t), ci * 3 + k] over t * stride + k - 1 == tin
__global__ __launch_bounds__(256) void conv1d_col2im_kernel(const elem_t* __restrict__ dcol, elem_t* __restrict__ dx, int B,
                                                            int C, int Tin, int Tout, int stride) {
    const long total = (long)B * Tin * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ci = (int)(idx % C);
        const long r = idx / C;
        const int tin = (int)(r % Tin);
        const long b = r / Tin;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int num = tin + 1 - k;
            if (num >= 0 && num % stride == 0) {
                const int t = num / stride;
                if (t < Tout) acc += e2f(dcol[((b * Tout + t) * C + ci) * 3 + k]);
            }
        }
        dx[idx] = f2e(acc);
    }
}
extern "C" int AA_FN(aa_conv1d_col2im)(const void* dcol, void* dx, int B, int C, int Tin, int Tout, int stride, void* stream) {
    AA_REQUIRE(B >= 0 && C > 0 && Tin > 0 && Tout > 0 && stride > 0, "aa_conv1d_col2im: bad shape");
    if (B == 0) return AA_OK;
    const long total = (long)B * Tin * C;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(conv1d_col2im_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)dcol, (elem_t*)dx, B, C,
                       Tin, Tout, stride);
    AA_CHECK_LAUNCH("aa_conv1d_col2im");
    return AA_OK;
}
// nn.AvgPool1d(2, stride 2) over time of token-major rows (T even): out[j] = (x[2j] + x[2j+1]) / 2 ; backward dx[2j] = dx[2j+1] = dy[j] / 2
__global__ __launch_bounds__(256) void avgpool2_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ y, long rows_out, int C,
                                                       int backward) {
    const int nv = C >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < rows_out * nv; idx += (long)gridDim.x * 256) {
        const long r = idx / nv;
        const int v = (int)(idx % nv);
        if (!backward) {
            ev8 a = *reinterpret_cast<const ev8*>(x + (2 * r) * C + v * 8);
            ev8 b = *reinterpret_cast<const ev8*>(x + (2 * r + 1) * C + v * 8);
            ev8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(0.5f * (e2f(a[j]) + e2f(b[j])));
            *reinterpret_cast<ev8*>(y + r * C + v * 8) = o;
        } else {      // x = dy [rows_out, C], y = dx [2 * rows_out, C]
            ev8 a = *reinterpret_cast<const ev8*>(x + r * C + v * 8);
            ev8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(0.5f * e2f(a[j]));
            *reinterpret_cast<ev8*>(y + (2 * r) * C + v * 8) = o;
            *reinterpret_cast<ev8*>(y + (2 * r + 1) * C + v * 8) = o;
        }
    }
}
extern "C" int AA_FN(aa_avgpool2)(const void* x, void* y, long rows_out, int C, int backward, void* stream) {
    AA_REQUIRE(rows_out >= 0 && C > 0 && (C & 7) == 0, "aa_avgpool2: C=%d must be a multiple of 8", C);
    if (rows_out == 0) return AA_OK;
    const long total = rows_out * (C >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (elem_t*)y, rows_out, C, backward);
    AA_CHECK_LAUNCH("aa_avgpool2");
    return AA_OK;
}

// ================================================================== score head  (Linear(h -> 1, bias=False))
// align_anything/models/opt.py:59-60 / models/llava.py:60: scores = score_head(last_hidden_state).float()
// out[r] = float(bf16(sum_c x[r,c] * w[c]))   (the bf16 Linear output, upcast)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const elem_t* __restrict__ x,
                                                         const elem_t* __restrict__ w,
                                                         float* __restrict__ out, long rows, int h) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = h >> 3;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        const elem_t* xr = x + row * h;
        float acc = 0.f;
        for (int i = lane; i < nv; i += 64) {
            ev8 a = *reinterpret_cast<const ev8*>(xr + i * 8);
            ev8 b = *reinterpret_cast<const ev8*>(w + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += e2f(a[j]) * e2f(b[j]);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[row] = ernd(acc);
    }
}
// dx[r,c] = dy[r] * w[c] ; dw_part[block, c] = sum over the block's rows of dy[r] * x[r,c]
template <int MAXV>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy,
                                                         const elem_t* __restrict__ x,
                                                         const elem_t* __restrict__ w,
                                                         elem_t* __restrict__ dx,
                                                         float* __restrict__ dw_part, long rows, int h) {
    const int nv = h >> 3;
    float dwacc[MAXV][8];
#pragma unroll
    for (int a = 0; a < MAXV; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[a][j] = 0.f;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float g = dy[row];
#pragma unroll
        for (int a = 0; a < MAXV; ++a) {
            const int i = threadIdx.x + a * 256;
            if (i < nv) {
                ev8 xv = *reinterpret_cast<const ev8*>(x + row * h + i * 8);
                ev8 wv = *reinterpret_cast<const ev8*>(w + i * 8);
                ev8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dwacc[a][j] += g * e2f(xv[j]);
                    o[j] = f2e(g * e2f(wv[j]));
                }
                *reinterpret_cast<ev8*>(dx + row * h + i * 8) = o;
            }
        }
    }
    if (dw_part) {
        float* pr = dw_part + (long)blockIdx.x * h;
#pragma unroll
        for (int a = 0; a < MAXV; ++a) {
            const int i = threadIdx.x + a * 256;
            if (i < nv) {
                *reinterpret_cast<f32x4*>(pr + i * 8) = f32x4{dwacc[a][0], dwacc[a][1], dwacc[a][2], dwacc[a][3]};
                *reinterpret_cast<f32x4*>(pr + i * 8 + 4) = f32x4{dwacc[a][4], dwacc[a][5], dwacc[a][6], dwacc[a][7]};
            }
        }
    }
}

extern "C" int AA_FN(aa_rowdot_fwd)(const void* x, const void* w, float* out, long rows, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_rowdot_fwd: hidden %d must be a multiple of 8", h);
    if (rows == 0) return AA_OK;
    const long nb = (rows + 3) / 4;
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((int)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)x, (const elem_t*)w, out, rows, h);
    AA_CHECK_LAUNCH("aa_rowdot_fwd");
    return AA_OK;
}
extern "C" int AA_FN(aa_rowdot_bwd)(const float* dy, const void* x, const void* w, void* dx, float* dw, float* ws,
                             int ws_rows, long rows, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0 && h <= 8 * 256 * 8, "aa_rowdot_bwd: hidden %d must be a multiple of 8 and <= 16384", h);
    AA_REQUIRE(dw == nullptr || (ws != nullptr && ws_rows > 0), "aa_rowdot_bwd: dw needs a [ws_rows, h] fp32 workspace");
    if (rows == 0) return AA_OK;
    int grid = (int)(rows < 1024 ? rows : 1024);
    if (dw && grid > ws_rows) grid = ws_rows;
    float* part = dw ? ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_RDB(MV)                                                                                \
    hipLaunchKernelGGL(rowdot_bwd_kernel<MV>, dim3(grid), dim3(256), 0, st, dy, (const elem_t*)x,      \
                       (const elem_t*)w, (elem_t*)dx, part, rows, h)
    if (h <= 2048) LAUNCH_RDB(1);
    else if (h <= 4096) LAUNCH_RDB(2);
    else if (h <= 8192) LAUNCH_RDB(4);
    else LAUNCH_RDB(8);
#undef LAUNCH_RDB
    if (dw) launch_reduce_rows(part, grid, h, dw, st);
    AA_CHECK_LAUNCH("aa_rowdot_bwd");
    return AA_OK;
}

}  // namespace AA_ELEM_NS
