// Mixture-of-experts routing and token movement for Qwen3-MoE (hf:models/qwen3_moe/modeling_qwen3_moe.py:210-283):
//   router:  probs = softmax_fp32(x Wg^T); top-k; optional renormalisation over the k; weights cast to the activation dtype
//   experts: tokens grouped by expert (64-row aligned segments so a segment can be the contraction dim of a dW GEMM),
//            per-expert SwiGLU MLP through the ordinary GEMM, weighted combine back to token order.
// The combine is a GATHER over each token's k expert rows (no atomics: deterministic).  Written against elem_t and
// compiled twice (bf16 production / fp32 parity mode, moe_f32.hip), like elementwise.hip.
#include "aa_common.h"

namespace AA_ELEM_NS {

// one wave per token row; E <= 1024.  probs fp32 [rows, E] is kept for the backward.
__global__ __launch_bounds__(256) void moe_route_kernel(const elem_t* __restrict__ logits, long ld, long rows, int E, int k,
                                                        int norm, float* __restrict__ probs, int* __restrict__ idx,
                                                        elem_t* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const elem_t* x = logits + row * ld;
    float mx = -INFINITY;
    for (int e = lane; e < E; e += 64) mx = fmaxf(mx, e2f(x[e]));
    mx = wave_max(mx);
    float z = 0.f;
    for (int e = lane; e < E; e += 64) z += expf(e2f(x[e]) - mx);
    z = wave_sum(z);
    float* pr = probs + row * E;
    for (int e = lane; e < E; e += 64) pr[e] = expf(e2f(x[e]) - mx) / z;
    __threadfence_block();     // other lanes of this wave read the probabilities back below
    // k rounds of wave-wide argmax (ties -> lower index), previous winners excluded
    float sum = 0.f;
    float val[8]; int sel[8];
    for (int j = 0; j < k; ++j) {
        float best = -1.f; int bi = 0x7fffffff;
        for (int e = lane; e < E; e += 64) {
            bool taken = false;
            for (int q = 0; q < j; ++q) taken |= (sel[q] == e);
            const float p = taken ? -1.f : pr[e];
            if (p > best || (p == best && e < bi)) { best = p; bi = e; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        val[j] = best; sel[j] = bi; sum += best;
    }
    if (lane == 0) {
        for (int j = 0; j < k; ++j) {
            idx[row * k + j] = sel[j];
            w[row * k + j] = f2e(norm ? val[j] / sum : val[j]);
        }
    }
}

extern "C" int AA_FN(aa_moe_route)(const void* logits, long ld, long rows, int E, int k, int norm_topk, float* probs, int* idx,
                                   void* weights, void* stream) {
    AA_REQUIRE(rows >= 0 && E > 0 && k > 0 && k <= 8 && k <= E, "aa_moe_route: bad shape rows=%ld E=%d k=%d (k <= 8)", rows, E, k);
    if (rows == 0) return AA_OK;
    hipLaunchKernelGGL(moe_route_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)logits, ld, rows,
                       E, k, norm_topk, probs, idx, (elem_t*)weights);
    AA_CHECK_LAUNCH("aa_moe_route");
    return AA_OK;
}

// d logits from d weights:  w_j = p_j / s (s = sum of the selected p, or 1 when not normalised);
//   dp_i = (dw_i - [norm] sum_j dw_j w_j) / s for the selected i, 0 otherwise;  dlogit = p * (dp - sum_i p_i dp_i)
__global__ __launch_bounds__(256) void moe_route_bwd_kernel(const float* __restrict__ probs, const int* __restrict__ idx,
                                                            const float* __restrict__ dw, long rows, int E, int k, int norm,
                                                            elem_t* __restrict__ dlogits, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = probs + row * E;
    float s = 1.f, dot_w = 0.f;
    if (norm) {
        s = 0.f;
        for (int j = 0; j < k; ++j) s += pr[idx[row * k + j]];
        for (int j = 0; j < k; ++j) dot_w += dw[row * k + j] * (pr[idx[row * k + j]] / s);
    }
    float pdp = 0.f;       // sum_i p_i dp_i (only selected i contribute)
    for (int j = 0; j < k; ++j) {
        const float p = pr[idx[row * k + j]];
        pdp += p * (dw[row * k + j] - dot_w) / s;
    }
    for (int e = lane; e < E; e += 64) {
        float dp = 0.f;
        for (int j = 0; j < k; ++j) if (idx[row * k + j] == e) dp = (dw[row * k + j] - dot_w) / s;
        dlogits[row * ld + e] = f2e(pr[e] * (dp - pdp));
    }
}
extern "C" int AA_FN(aa_moe_route_bwd)(const float* probs, const int* idx, const float* dweights, long rows, int E, int k,
                                       int norm_topk, void* dlogits, long ld, void* stream) {
    AA_REQUIRE(rows >= 0 && E > 0 && k > 0 && k <= 8, "aa_moe_route_bwd: bad shape rows=%ld E=%d k=%d", rows, E, k);
    if (rows == 0) return AA_OK;
    hipLaunchKernelGGL(moe_route_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, probs, idx, dweights, rows, E,
                       k, norm_topk, (elem_t*)dlogits, ld);
    AA_CHECK_LAUNCH("aa_moe_route_bwd");
    return AA_OK;
}

// Xp[r, :] = src[r] >= 0 ? x[src[r], :] : 0      (expert-major token copy; pad rows of a segment are zero)
__global__ __launch_bounds__(256) void moe_gather_kernel(const elem_t* __restrict__ x, const int* __restrict__ src,
                                                         elem_t* __restrict__ out, long rows_out, int h) {
    const int nv = h >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows_out * nv; i += (long)gridDim.x * 256) {
        const long r = i / nv;
        const int v = (int)(i % nv);
        const int s = src[r];
        ev8 o;
        if (s >= 0) o = *reinterpret_cast<const ev8*>(x + (long)s * h + v * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(0.f);
        }
        *reinterpret_cast<ev8*>(out + r * h + v * 8) = o;
    }
}
extern "C" int AA_FN(aa_moe_gather)(const void* x, const int* src_row, void* out, long rows_out, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_moe_gather: hidden %d must be a multiple of 8", h);
    if (rows_out == 0) return AA_OK;
    const long total = rows_out * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(moe_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, src_row, (elem_t*)out, rows_out, h);
    AA_CHECK_LAUNCH("aa_moe_gather");
    return AA_OK;
}

// out[r, :] = (a[r] >= 0 ? x[a[r], :] : 0) + (b[r] >= 0 ? x[b[r], :] : 0), the sum rounded once to the activation dtype: the pack-reduce of shared-prompt
// packing (trainers/common.py::build_pack_plan) -- a packed row's q | k | v gradient is the sum over the (one or two) slots of the reference layout that hold a
// copy of it.  Bit-identical to aa_moe_gather twice + aa_add (the gathers are exact, the add rounds once), one pass instead of three.
__global__ __launch_bounds__(256) void gather2_add_kernel(const elem_t* __restrict__ x, const int* __restrict__ a, const int* __restrict__ b,
                                                          elem_t* __restrict__ out, long rows_out, int h) {
    const int nv = h >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows_out * nv; i += (long)gridDim.x * 256) {
        const long r = i / nv;
        const int v = (int)(i % nv);
        const int sa = a[r], sb = b[r];
        ev8 o;
        if (sa >= 0 && sb >= 0) {
            const ev8 p = *reinterpret_cast<const ev8*>(x + (long)sa * h + v * 8);
            const ev8 q = *reinterpret_cast<const ev8*>(x + (long)sb * h + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(p[j]) + e2f(q[j]));
        } else if (sa >= 0 || sb >= 0) {
            const ev8 p = *reinterpret_cast<const ev8*>(x + (long)(sa >= 0 ? sa : sb) * h + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(e2f(p[j]) + 0.f);          // (+ 0: what the separate add does to the gathered zero row; -0 -> +0)
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2e(0.f);
        }
        *reinterpret_cast<ev8*>(out + r * h + v * 8) = o;
    }
}
extern "C" int AA_FN(aa_gather2_add)(const void* x, const int* row_a, const int* row_b, void* out, long rows_out, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0, "aa_gather2_add: hidden %d must be a multiple of 8", h);
    AA_REQUIRE(row_a != nullptr && row_b != nullptr, "aa_gather2_add: both index vectors are required (-1 = no row)");
    if (rows_out == 0) return AA_OK;
    const long total = rows_out * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(gather2_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, row_a, row_b, (elem_t*)out, rows_out, h);
    AA_CHECK_LAUNCH("aa_gather2_add");
    return AA_OK;
}

// out[t, :] = (residual ? residual[t, :] : 0) + sum_j w[t, j] * Yp[pos[t, j], :]   (w == NULL -> unit weights; pos < 0 -> the slot is skipped)
// hf :244-246: each expert output is multiplied by the (activation-dtype) weight, rounded, then index_add'ed in expert order.
// Round 3: every slot's row is requested before the first one is used (the rows are 4 KB apart in HBM; the round-1 form loaded and accumulated slot by
// slot, one dependent round trip each: 2.4 TB/s at k = 8), and the ascending-row order comes from each slot's RANK among the token's positions (k <= 8
// compares per slot, no indexed private array).  Same arithmetic, same order: bit-identical.
template <int KMAX>
__global__ __launch_bounds__(256) void moe_combine_kernel(const elem_t* __restrict__ yp, const int* __restrict__ pos,
                                                          const elem_t* __restrict__ w, const elem_t* __restrict__ residual,
                                                          elem_t* __restrict__ out, long rows, int k, int h) {
    const int nv = h >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * nv; i += (long)gridDim.x * 256) {
        const long t = i / nv;
        const int v = (int)(i % nv);
        int pj[KMAX];
        float wj[KMAX];
        ev8 y[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            pj[j] = j < k ? pos[t * k + j] : -1;                   // < 0: no row behind this slot (pad rows of the capacity-padded exchange): adds nothing
            wj[j] = (j < k && w) ? e2f(w[t * k + j]) : 1.f;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (pj[j] >= 0) y[j] = *reinterpret_cast<const ev8*>(yp + (long)pj[j] * h + v * 8);
        // hf index_add's the expert outputs in ascending expert order = ascending row of the expert-major buffer
        int rk[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            int r = 0;
#pragma unroll
            for (int q = 0; q < KMAX; ++q) r += (pj[q] >= 0 && pj[q] < pj[j]) ? 1 : 0;      // a token's rows are distinct
            rk[j] = pj[j] >= 0 ? r : -1;
        }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < KMAX; ++r) {
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                if (rk[j] == r) {                                  // uniform over the lanes that share the token
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] = ernd(acc[q] + ernd(e2f(y[j][q]) * wj[j]));
                }
            }
        }
        ev8 o;
        if (residual) {
            const ev8 r = *reinterpret_cast<const ev8*>(residual + t * h + v * 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = f2e(acc[q] + e2f(r[q]));
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = f2e(acc[q]);
        }
        *reinterpret_cast<ev8*>(out + t * h + v * 8) = o;
    }
}
extern "C" int AA_FN(aa_moe_combine)(const void* yp, const int* pos, const void* weights, const void* residual, void* out, long rows,
                                     int k, int h, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0 && k > 0 && k <= 8, "aa_moe_combine: hidden %d must be a multiple of 8 (k <= 8)", h);
    if (rows == 0) return AA_OK;
    const long total = rows * (h >> 3);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
#define AA_COMBINE(KM) hipLaunchKernelGGL(moe_combine_kernel<KM>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const elem_t*)yp, pos, \
                                         (const elem_t*)weights, (const elem_t*)residual, (elem_t*)out, rows, k, h)
    if (k == 1) AA_COMBINE(1); else if (k == 2) AA_COMBINE(2); else if (k <= 4) AA_COMBINE(4); else AA_COMBINE(8);
#undef AA_COMBINE
    AA_CHECK_LAUNCH("aa_moe_combine");
    return AA_OK;
}

// backward of the weighted combine: dYp[pos[t, j], :] = w[t, j] * dout[t, :] ;  dw[t, j] = <dout[t, :], Yp[pos[t, j], :]>
// one wave per (token, slot).  Rows of dYp that no pair writes (the pad rows of the expert segments) carry no gradient: with `src_row` (the plan's row ->
// token table, -1 = pad) the launch has one more wave per layout row that zeroes those itself -- round 3 had a torch memset of the WHOLE [cap, h] buffer in
// front of every call (at::native::FillFunctor<bf16> in the product path); with src_row == null the caller must have zeroed them.
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const elem_t* __restrict__ dout, const elem_t* __restrict__ yp,
                                                              const int* __restrict__ pos, const elem_t* __restrict__ w,
                                                              elem_t* __restrict__ dyp, float* __restrict__ dw, long rows, int k,
                                                              int h, const int* __restrict__ src_row, long cap_rows) {
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= rows * k) {
        const long r = pair - rows * k;
        if (src_row != nullptr && r < cap_rows && src_row[r] < 0) {
            ev8 z;
#pragma unroll
            for (int q = 0; q < 8; ++q) z[q] = f2e(0.f);
            for (int c = lane * 8; c < h; c += 512) *reinterpret_cast<ev8*>(dyp + r * h + c) = z;
        }
        return;
    }
    const long t = pair / k;
    const long r = pos[pair];
    if (r < 0) {                                   // slot without a row (see moe_combine_kernel): no gradient row, zero weight gradient
        if (lane == 0) dw[pair] = 0.f;
        return;
    }
    const float wj = e2f(w[pair]);
    float dot = 0.f;
    for (int c = lane * 8; c < h; c += 512) {
        const ev8 g = *reinterpret_cast<const ev8*>(dout + t * h + c);
        const ev8 y = *reinterpret_cast<const ev8*>(yp + r * h + c);
        ev8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) { dot += e2f(g[q]) * e2f(y[q]); o[q] = f2e(e2f(g[q]) * wj); }
        *reinterpret_cast<ev8*>(dyp + r * h + c) = o;
    }
    dot = wave_sum(dot);
    if (lane == 0) dw[pair] = dot;
}
extern "C" int AA_FN(aa_moe_combine_bwd)(const void* dout, const void* yp, const int* pos, const void* weights, void* dyp,
                                         float* dweights, long rows, int k, int h, const int* src_row, long cap_rows, void* stream) {
    AA_REQUIRE(h > 0 && (h & 7) == 0 && k > 0, "aa_moe_combine_bwd: hidden %d must be a multiple of 8", h);
    AA_REQUIRE(src_row == nullptr || cap_rows >= 0, "aa_moe_combine_bwd: cap_rows %ld", cap_rows);
    const long waves = rows * k + (src_row ? cap_rows : 0);
    if (waves == 0) return AA_OK;
    hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3((int)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)dout,
                       (const elem_t*)yp, pos, (const elem_t*)weights, (elem_t*)dyp, dweights, rows, k, h, src_row, cap_rows);
    AA_CHECK_LAUNCH("aa_moe_combine_bwd");
    return AA_OK;
}

#ifndef AA_ELEM_F32   // integer work: one instantiation
// Expert-major layout of the (token, slot) pairs, entirely on the device (no host read):
//   counts[e]; off[e+1] = off[e] + align_up(counts[e], align) (align = 128 = the row tile of the grouped GEMM);
//   pos[pair] = off[e] + rank of the pair among expert e's pairs in (token, slot) order (stable); pairs whose idx is outside [0, E) are in no
//   segment and their pos is left as the caller initialised it (-1 for the rows of the capacity-padded exchange that carry no token);
//   src[row] = token of the pair stored at that row, -1 for pad rows / rows beyond off[E];
//   tile_expert[t] = expert owning rows [t*tg, (t+1)*tg), -1 beyond off[E]; tg = 128 when align is a multiple of 128 (256: the gemm4 tile), else align.
// One workgroup per expert counts, then scans all pairs with ballots (E x rows*k reads; E <= 1024) and writes its rows.
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ __launch_bounds__(256) void moe_count_kernel(const int* __restrict__ idx, long npairs, int* __restrict__ counts) {
    __shared__ int wsum[4];
    const int e = blockIdx.x;
    int c = 0;
    const long nv = npairs >> 2;                                     // 16-byte loads (idx comes from the allocator: 16-byte aligned)
    const i32x4* v4 = reinterpret_cast<const i32x4*>(idx);
    for (long i = threadIdx.x; i < nv; i += 256) {
        const i32x4 q = v4[i];
        c += (q[0] == e) + (q[1] == e) + (q[2] == e) + (q[3] == e);
    }
    for (long i = (nv << 2) + threadIdx.x; i < npairs; i += 256) c += (idx[i] == e) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[e] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void moe_plan_kernel(const int* __restrict__ idx, long npairs, int k, int E, int align, long cap_rows,
                                                       const int* __restrict__ counts, int* __restrict__ off, int* __restrict__ pos,
                                                       int* __restrict__ src, int* __restrict__ tile_expert) {
    __shared__ int wsum[2][4];
    __shared__ int my_off;
    const int e = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int before = 0;
        for (int ee = 0; ee < e; ++ee) before += (counts[ee] + align - 1) / align * align;
        my_off = before; off[e] = before;
    }
    __syncthreads();
    const int o0 = my_off, my_cnt = counts[e];
    const int seg = (my_cnt + align - 1) / align * align;
    if (e == E - 1 && threadIdx.x == 0) off[E] = o0 + seg;
    // stable ranks: scan the pairs in order, 256 threads x VPT consecutive pairs at a time (round 3: was 256 pairs and four barriers per trip --
    // 140 us per plan at 65 k pairs x 128 experts; one barrier per 4096 pairs now).  A thread's matches form a bit mask; the block-wide exclusive
    // prefix of the match counts (wave scan + the four wave totals, double-buffered so one barrier per trip is enough) places them.
    constexpr int VPT = 16;
    int base = 0;                                                   // matches in earlier trips (every thread keeps the same value)
    int trip = 0;
    for (long b0 = 0; b0 < npairs; b0 += 256 * VPT, ++trip) {
        const long i0 = b0 + (long)threadIdx.x * VPT;
        unsigned m = 0;
        if (i0 + VPT <= npairs) {
            const i32x4* v4 = reinterpret_cast<const i32x4*>(idx + i0);
#pragma unroll
            for (int q = 0; q < VPT / 4; ++q) {
                const i32x4 x = v4[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) m |= (x[j] == e ? 1u : 0u) << (4 * q + j);
            }
        } else {
            for (int j = 0; j < VPT; ++j)
                if (i0 + j < npairs && idx[i0 + j] == e) m |= 1u << j;
        }
        const int c = __popc(m);
        int inc = c;                                                // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[trip & 1][wid] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int sw = wsum[trip & 1][w];
            woff += w < wid ? sw : 0;
            tot += sw;
        }
        int r = o0 + base + woff + inc - c;
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1;
            pos[i0 + j] = r;
            src[r] = (int)((i0 + j) / k);
            ++r;
        }
        base += tot;
    }
    for (int r = my_cnt + threadIdx.x; r < seg; r += 256) src[o0 + r] = -1;           // pad rows of the segment
    const int tg = (align % 128 == 0) ? 128 : align;     // rows per table entry: 128 for every alignment the grouped GEMMs take (their tiles are 128 or 256 rows)
    for (int t = threadIdx.x; t < seg / tg; t += 256) tile_expert[o0 / tg + t] = e;
    if (e == E - 1) {                                                                   // everything beyond the last segment
        for (long r = o0 + seg + threadIdx.x; r < cap_rows; r += 256) src[r] = -1;
        for (long t = (o0 + seg) / tg + threadIdx.x; t < cap_rows / tg; t += 256) tile_expert[t] = -1;
    }
}
extern "C" int aa_moe_plan(const int* idx, long rows, int k, int E, int align, long cap_rows, int* counts, int* off, int* pos, int* src,
                           int* tile_expert, void* stream) {
    AA_REQUIRE(((uintptr_t)idx & 15) == 0, "aa_moe_plan: idx must be 16-byte aligned");
    AA_REQUIRE(rows >= 0 && k > 0 && E > 0 && align > 0 && cap_rows % align == 0 && cap_rows >= rows * k + (long)E * (align - 1) / align * align,
               "aa_moe_plan: cap_rows=%ld too small / unaligned for rows=%ld k=%d E=%d align=%d", cap_rows, rows, k, E, align);
    hipLaunchKernelGGL(moe_count_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, idx, rows * k, counts);
    hipLaunchKernelGGL(moe_plan_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, idx, rows * k, k, E, align, cap_rows, counts, off, pos, src,
                       tile_expert);
    AA_CHECK_LAUNCH("aa_moe_plan");
    return AA_OK;
}
#endif

}  // namespace AA_ELEM_NS
